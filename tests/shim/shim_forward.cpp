// The whole encrypted Linformer-d128 forward pass driven through include/FHEController.h in C++: the call sequence of
// the reference's drivers (encoder1 -> pooler -> classifier; reference src/main.cpp:145-475 for the CLS-query variant,
// src/main_2.cpp:145-430 for full attention), reading the reference's text weight files (format of
// src/python/extract_parameters_numeric.py:28) through every read_* packing layout and fitting the Chebyshev
// coefficients in the shim's C++ code.  tests/test_shim_gpu.py writes synthetic weights/tokens, runs this binary on the
// GPU and compares the dumped slot vectors with oracle/circuit_sim.py.
//
// usage: shim_forward <root> <main|main_2> [generate] [batch=B]
//   batch=B: B samples through ONE controller in one pass (include/FHEControllerBatch.h): sample x reads <root>/input_b<x>, <root>/tokens_b<x>
//   and dumps to <root>/out/b<x>; the driver code below is the same template for one sample and for a batch.
//   <root>/weights-20NG, <root>/input (XE_i / XF_i), <root>/tokens (input_i.txt), <root>/keys, <root>/checkpoint;
//   the process runs in <root>/build so that the reference's relative paths ("../weights-20NG/...") resolve.
//   "generate": write ../keys first (what `FHE-Linformer --generate_keys` does, main.cpp:81-86); otherwise the keys
//   directory written by an earlier run — of this driver or of the reference's own binary — is loaded (:88-89).
#include "FHEControllerBatch.h"
#include <filesystem>
#include <unistd.h>

FHEController controller;

namespace {
const string W = "../weights-20NG/";
const string L0 = W + "linformer_transformerLayers_transformer0_";
int n_bootstraps = 0;

int n_samples = 0;   // 0: one sample through FHEController; B > 0: a batch through FHEControllerBatch
void dump_to(const string& path, const vector<double>& v) {
    ofstream f(path);
    f << setprecision(17);
    for (size_t i = 0; i < v.size(); i++) f << v[i] << (i + 1 == v.size() ? "\n" : ",");
}
void dump(const string& name, const Ctxt& c) { dump_to("../out/" + name + ".out", controller.decrypt_tovector(c, 16384)); }
void dump(const string& name, const CtxtBatch& c) {
    for (size_t x = 0; x < c->size(); x++) dump_to("../out/b" + to_string(x) + "/" + name + ".out", controller.decrypt_tovector(c->sample(x), 16384));
}
// the two controllers behind one driver: inputs are per sample, plaintexts are shared
struct One {
    typedef Ctxt H;
    FHEController& ctl() { return controller; }
    Ctxt input(const string& dir, const string& file) { return controller.read_expanded_input("../" + dir + "/" + file); }
    Ctxt shared_input(const string& path) { return controller.read_expanded_input(path); }
};
struct Many {
    typedef CtxtBatch H;
    FHEControllerBatch b;
    explicit Many(int B) : b(controller, B) {}
    FHEControllerBatch& ctl() { return b; }
    CtxtBatch input(const string& dir, const string& file) {
        vector<string> files;
        for (int x = 0; x < b.samples(); x++) files.push_back("../" + dir + "_b" + to_string(x) + "/" + file);
        return b.read_expanded_input(files);
    }
    CtxtBatch shared_input(const string& path) { return b.read_expanded_input(vector<string>(b.samples(), path)); }
};
double scalar_file(const string& path) {
    vector<double> v = read_values_from_file(path);
    if (v.empty()) {
        cerr << "no value in " << path << endl;
        exit(1);
    }
    return v[0];
}
template <class C, class H> H boot(C& ctl, const H& c) {
    ++n_bootstraps;
    return ctl.bootstrap(c);
}
template <class T> vector<T> slice(const vector<T>& v, size_t lo, size_t hi) {
    return vector<T>(v.begin() + std::min(lo, v.size()), v.begin() + std::min(hi, v.size()));
}
template <class T> vector<T> joined(vector<T> a, const vector<T>& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}

struct AffineParams {
    double c0, c1, c2;
    string a_file, b_file;
    double factor(size_t S) const { return c0 + c1 / sqrt((double)S) + c2 / (double)S; }
};
AffineParams affine(int which) {
    const string p = L0 + "ffn_affine" + to_string(which) + "_";
    return {scalar_file(p + "c0.txt"), scalar_file(p + "c1.txt"), scalar_file(p + "c2.txt"), p + "a.txt", p + "b.txt"};
}

// ---- encoder layer: self-attention, W_O + residual, affine-1, FFN with GELU, residual, affine-2
void save_checkpoint(const Ctxt& c) { controller.save(c, "../checkpoint/encodered.bin"); }
void save_checkpoint(const CtxtBatch&) {}   // the batch driver keeps no checkpoint
size_t level_of(const Ctxt& c) { return c->GetLevel(); }
size_t level_of(const CtxtBatch& c) { return c->GetLevel(); }

template <class Wr> typename Wr::H encoder(Wr& w_, bool full_attention) {
    typedef typename Wr::H H;
    auto& ctl = w_.ctl();
    int n_tokens = 0;
    for (auto& e : std::filesystem::directory_iterator(n_samples ? "../tokens_b0" : "../tokens")) n_tokens += e.is_regular_file() ? 1 : 0;
    vector<H> xe, xf, x;
    for (int i = 0; i < 32; i++) xe.push_back(w_.input("input", "XE_" + to_string(i) + ".txt"));
    for (int i = 0; i < 32; i++) xf.push_back(w_.input("input", "XF_" + to_string(i) + ".txt"));
    x.push_back(w_.shared_input(W + "cls_token.txt"));
    for (int i = 0; i < n_tokens; i++) x.push_back(w_.input("tokens", "input_" + to_string(i) + ".txt"));
    const size_t S = x.size();
    cout << S << " rows (CLS + " << n_tokens << " tokens)" << endl;

    Ptxt wq = controller.read_plain_input(L0 + "selfAttn_WQ_weight_T.txt"), bq = controller.read_plain_repeated_input(L0 + "selfAttn_WQ_bias.txt");
    Ptxt wk = controller.read_plain_input(L0 + "selfAttn_WK_weight_T.txt"), bk = controller.read_plain_repeated_input(L0 + "selfAttn_WK_bias.txt");
    vector<H> Q = ctl.matmulRE(x, wq, bq);
    H K = ctl.wrapUpRepeated(ctl.matmulRE(xe, wk, bk));
    Ptxt wv = controller.read_plain_input(L0 + "selfAttn_WV_weight_T.txt"), bv = controller.read_plain_repeated_input(L0 + "selfAttn_WV_bias.txt");

    vector<H> attn;
    if (!full_attention) {
        // main.cpp:196-224: the CLS query only; every other token's attention output is an encryption of zero
        H scores = ctl.matmulScores(Q[0], K);
        dump("scores", scores);
        scores = ctl.eval_exp(scores, 32);
        dump("exp", scores);
        H denom = ctl.eval_inverse_naive(ctl.rotsum(scores, 32, 128), -1, 128);
        scores = ctl.mult(scores, denom);
        vector<H> un = ctl.unwrapExpanded(scores, 1);
        H V = ctl.wrapUpRepeated(ctl.matmulRE(xf, wv, bv));
        H cls = ctl.matmulRE(un, V, 128, 128)[0];
        attn.push_back(cls);
        H zero = ctl.encrypt_ptxt(controller.encode(0, (int)cls->GetLevel(), 0));
        for (size_t i = 1; i < S; i++) attn.push_back(zero->Clone());
    } else {
        // main_2.cpp:187-229: all queries, wrapped 128 at a time
        vector<H> q1 = slice(Q, 0, 128), q2 = slice(Q, 128, S);
        H s1 = ctl.matmulScores(q1, K), s2 = ctl.matmulScores(q2, K);
        s1 = ctl.eval_exp(s1, (int)q1.size());
        s2 = ctl.eval_exp(s2, (int)q2.size());
        dump("scores", s1);   // trace names as in fhe-linformer_amd/linformer.py: both halves after eval_exp
        dump("exp", s2);
        H d1 = ctl.eval_inverse_naive(ctl.rotsum(s1, 32, 128), -1, 190000);
        H d2 = ctl.eval_inverse_naive(ctl.rotsum(s2, 32, 128), -1, 190000);
        s1 = ctl.mult(s1, d1);
        s2 = ctl.mult(s2, d2);
        vector<H> un = joined(ctl.unwrapExpanded(s1, 128), ctl.unwrapExpanded(s2, (int)S - 128));
        H V = ctl.wrapUpRepeated(ctl.matmulRE(xf, wv, bv));
        attn = ctl.matmulRE(un, V, 128, 128);
    }
    dump("self_attention", attn[0]);

    Ptxt wo = controller.read_plain_input(L0 + "selfAttn_WO_weight.txt", (int)attn[0]->GetLevel());
    Ptxt bo = controller.read_plain_expanded_input(L0 + "selfAttn_WO_bias.txt", (int)attn[0]->GetLevel() + 1);
    vector<H> h;
    if (full_attention) {
        h = ctl.matmulCR(attn, wo, bo);
    } else {
        Ptxt none = nullptr;
        h = ctl.matmulCR(attn, wo, none);
        h[0] = ctl.add(h[0], bo);
    }
    for (size_t i = 0; i < S; i++) h[i] = ctl.add(h[i], x[i]);

    const AffineParams A1 = affine(1);
    H w0 = ctl.wrapUpExpanded(slice(h, 0, 128)), w1 = ctl.wrapUpExpanded(slice(h, 128, S));
    Ptxt a1 = controller.read_plain_repeated_input(A1.a_file, (int)w0->GetLevel(), A1.factor(S));
    Ptxt b1 = controller.read_plain_repeated_input(A1.b_file, (int)w0->GetLevel() + 1, A1.factor(S));
    w0 = ctl.add(ctl.mult(w0, a1), b1);
    w1 = ctl.add(ctl.mult(w1, a1), b1);
    dump("affine1_0", w0);
    w0 = boot(ctl, w0);
    w1 = boot(ctl, w1);
    H keep0 = w0->Clone(), keep1 = w1->Clone();
    vector<H> t0 = ctl.unwrapExpanded(w0, 128), t1 = ctl.unwrapExpanded(w1, (int)S - 128);

    const double gelu_scale = 1.0 / 8.0;
    vector<Ptxt> ffn_in;
    for (int k = 0; k < 4; k++)
        ffn_in.push_back(controller.read_plain_input(W + "ffn_W0_transposed_block_" + to_string(k) + ".txt", (int)w0->GetLevel(), gelu_scale));
    Ptxt ffn_in_b = controller.read_plain_input(L0 + "ffn_Wffn_0_bias.txt", (int)w0->GetLevel() + 1, gelu_scale);
    t0 = ctl.matmulRElarge(t0, ffn_in, ffn_in_b);
    t1 = ctl.matmulRElarge(t1, ffn_in, ffn_in_b);
    vector<H> cont = ctl.generate_containers(joined(t0, t1), nullptr);
    for (auto& c : cont) {
        c = ctl.eval_gelu_function(c, -1, 1, gelu_scale, 119);
        c = boot(ctl, c);
    }
    vector<vector<H>> hidden = ctl.unwrapRepeatedLarge(cont, (int)S);

    const int lv = (int)hidden[0][0]->GetLevel();
    vector<Ptxt> ffn_out;
    for (int k = 0; k < 4; k++) ffn_out.push_back(controller.read_plain_input(W + "ffn_W2_block_" + to_string(k) + ".txt", lv));
    Ptxt ffn_out_b = controller.read_plain_expanded_input(L0 + "ffn_Wffn_2_bias.txt", lv + 1);
    vector<H> y = ctl.matmulCRlarge(hidden, ffn_out, ffn_out_b);
    H w2 = ctl.add(ctl.wrapUpExpanded(slice(y, 0, 128)), keep0);
    H w3 = ctl.add(ctl.wrapUpExpanded(slice(y, 128, S)), keep1);
    const AffineParams A2 = affine(2);
    Ptxt a2 = controller.read_plain_repeated_input(A2.a_file, (int)w2->GetLevel(), A2.factor(y.size()));
    Ptxt b2 = controller.read_plain_repeated_input(A2.b_file, (int)w3->GetLevel() + 1, A2.factor(y.size()));
    w2 = ctl.add(ctl.mult(w2, a2), b2);
    w3 = ctl.add(ctl.mult(w3, a2), b2);
    vector<H> out = ctl.unwrapExpanded(w2, 128);
    (void)ctl.unwrapExpanded(w3, (int)S - 128);   // the reference expands these rows too and drops them (quirk Q7)
    dump("encoder_out", out[0]);
    save_checkpoint(out[0]);   // main.cpp:422
    return out[0];
}

template <class Wr> typename Wr::H pool(Wr& w_, const typename Wr::H& in, double tanh_scale) {
    typedef typename Wr::H H;
    auto& ctl = w_.ctl();
    Ptxt w = controller.read_plain_input(W + "pooler_dense_weight_T.txt", (int)in->GetLevel(), tanh_scale);
    Ptxt b = controller.read_plain_repeated_input(W + "pooler_dense_bias.txt", (int)in->GetLevel() + 1, tanh_scale);
    H o = ctl.add(ctl.rotsum(ctl.mult(in, w), 128, 128), b);
    o = boot(ctl, o);
    o = ctl.eval_tanh_function(o, -1, 1, tanh_scale, 300);
    dump("pooled", o);
    return o;
}

template <class Wr> typename Wr::H classify(Wr& w_, const typename Wr::H& in, bool encrypted_mask) {
    typedef typename Wr::H H;
    auto& ctl = w_.ctl();
    Ptxt w = controller.read_plain_input(W + "fcLinear_0_weight.txt", (int)in->GetLevel());
    Ptxt b = controller.read_plain_expanded_input(W + "fcLinear_0_bias.txt", (int)in->GetLevel());
    H o = ctl.add(ctl.rotsum(ctl.mult(in, w), 128, 1), b);
    vector<double> m(controller.num_slots, 0.0);
    for (int i = 0; i < 20; i++) m[i * 128] = 1;
    if (encrypted_mask) return ctl.mult(o, ctl.encrypt(m, (int)o->GetLevel()));   // main.cpp:472
    return ctl.mult(o, controller.encode(m, (int)o->GetLevel(), 0));                     // main_2.cpp:427
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        cerr << "usage: shim_forward <root> <main|main_2> [generate]" << endl;
        return 2;
    }
    const string root = argv[1], variant = argv[2];
    bool generate = false;
    for (int i = 3; i < argc; i++) {
        const string a = argv[i];
        if (a == "generate") generate = true;
        if (a.rfind("batch=", 0) == 0) n_samples = atoi(a.c_str() + 6);
    }
    const bool full = variant == "main_2";
    for (const char* d : {"build", "keys", "checkpoint", "out"}) std::filesystem::create_directories(root + "/" + d);
    for (int x = 0; x < n_samples; x++) std::filesystem::create_directories(root + "/out/b" + to_string(x));
    if (chdir((root + "/build").c_str()) != 0) return 2;
    const vector<int> rotations = {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, -1, -2, -4, -8, -16, -32, -64};
    if (generate) {
        controller.generate_context(true, false);
        controller.generate_bootstrapping_and_rotation_keys(rotations, 16384, true, "rotation_keys.txt");
    } else {
        controller.load_context(false);
        controller.load_bootstrapping_and_rotation_keys("rotation_keys.txt", 16384, false);
    }
    auto t0 = start_time();
    if (n_samples > 0) {
        Many w(n_samples);
        CtxtBatch enc = encoder(w, full);
        CtxtBatch pooled = pool(w, enc, full ? 1.0 / 18 : 1.0 / 50);
        CtxtBatch logits = classify(w, pooled, !full);
        print_duration(t0, "forward pass of " + to_string(n_samples) + " samples");
        dump("logits", logits);
        ofstream("../out/meta.out") << n_bootstraps << "," << logits->GetLevel() << "," << enc->GetLevel() << "\n";
        return 0;
    }
    One w;
    Ctxt enc = encoder(w, full);
    Ctxt pooled = pool(w, enc, full ? 1.0 / 18 : 1.0 / 50);
    Ctxt logits = classify(w, pooled, !full);
    print_duration(t0, "forward pass");
    dump("logits", logits);
    ofstream("../out/meta.out") << n_bootstraps << "," << logits->GetLevel() << "," << enc->GetLevel() << "\n";
    return 0;
}
