/*
 * FHEControllerBatch.h — the reference's FHEController surface for a BATCH OF SAMPLES through one engine.
 *
 * The reference's driver (src/main.cpp:145-475, src/main_2.cpp:145-430) processes one sample per run; samples are independent
 * circuits ("the batch of independent input ciphertexts", BASELINE config 4).  `FHEControllerBatch` carries B samples through the
 * SAME call sequence with the same method names and argument meaning as `FHEController` (include/FHEController.h, reference
 * src/FHEController.h:34-152): where the driver holds one `Ctxt` it holds a `CtxtBatch` (one ciphertext per sample; `->GetLevel()`,
 * `->Clone()` like `Ctxt`), where it holds a `vector<Ctxt>` of rows a `vector<CtxtBatch>`.  Every method hands the rows of ALL samples
 * to one entry point of the C ABI - the row loops concatenated into fhelin_fc_matmul_pt / _matmulRElarge / _matmulCRlarge /
 * fhelin_mult_batch ..., the calls that fold a row set into one ciphertext through fhelin_fcb_* (include/fhelin.h) - so that one key
 * set, one plaintext cache and one launch set serve the whole batch.  Sample x ends in exactly the residues its own pass through
 * `FHEController` gives (tests/test_batched_forward_gpu.py compares them for the Python twin of this class).
 * Plaintexts (`Ptxt`) are shared by the samples: read / encode them through the wrapped `FHEController` (`one()`).
 */
#ifndef FHELIN_FHECONTROLLER_BATCH_H
#define FHELIN_FHECONTROLLER_BATCH_H

#include "FHEController.h"

namespace fhelin_shim {
class CiphertextBatchImpl;
}
using CtxtBatch = std::shared_ptr<fhelin_shim::CiphertextBatchImpl>;

namespace fhelin_shim {
class CiphertextBatchImpl {
public:
    vector<Ctxt> v;   // sample order
    explicit CiphertextBatchImpl(vector<Ctxt> cts) : v(std::move(cts)) {}
    size_t GetLevel() const { return v.at(0)->GetLevel(); }
    uint32_t GetSlots() const { return v.at(0)->GetSlots(); }
    size_t size() const { return v.size(); }
    const Ctxt& sample(size_t x) const { return v.at(x); }
    CtxtBatch Clone() const {
        vector<Ctxt> c;
        for (auto& h : v) c.push_back(h->Clone());
        return std::make_shared<CiphertextBatchImpl>(c);
    }
};
}  // namespace fhelin_shim

class FHEControllerBatch {
public:
    int num_slots;

    FHEControllerBatch(FHEController& single, int samples) : num_slots(single.num_slots), c_(single), B_(samples) {
        if (samples < 1) throw std::runtime_error("FHEControllerBatch: need at least one sample");
    }
    FHEController& one() { return c_; }
    int samples() const { return B_; }
    static CtxtBatch make(vector<Ctxt> cts) { return std::make_shared<fhelin_shim::CiphertextBatchImpl>(std::move(cts)); }

    /* inputs: one file per sample (reference read_expanded_input :623-650 once per sample) */
    CtxtBatch read_expanded_input(const vector<string>& files, double scale = 1) {
        need_B(files.size());
        vector<Ctxt> v;
        for (auto& f : files) v.push_back(c_.read_expanded_input(f, scale));
        return make(v);
    }
    /* the server-side encryptions of the drivers (src/main.cpp:220, :472): one fresh encryption per sample */
    CtxtBatch encrypt(const vector<double>& vec, int level = 0, int plaintext_num_slots = 0) {
        vector<Ctxt> v;
        for (int x = 0; x < B_; x++) v.push_back(c_.encrypt(vec, level, plaintext_num_slots));
        return make(v);
    }
    CtxtBatch encrypt_ptxt(const Ptxt& p) {
        vector<Ctxt> v;
        for (int x = 0; x < B_; x++) v.push_back(c_.encrypt_ptxt(p));
        return make(v);
    }
    vector<vector<double>> decrypt_tovector(const CtxtBatch& c, int slots) {
        vector<vector<double>> out;
        for (auto& h : c->v) out.push_back(c_.decrypt_tovector(h, slots));
        return out;
    }

    /* leaf operations (reference :409-436) */
    CtxtBatch add(const CtxtBatch& a, const CtxtBatch& b) {
        // one EvalAdd per sample through the single-handle entry point: the C ABI defers these and runs everything pending - the
        // driver's loop over its S residual additions (src/main.cpp:237-239) times B samples - as ONE batched adjustment + addition
        vector<Ctxt> v;
        for (int x = 0; x < B_; x++) v.push_back(c_.add(a->v.at(x), b->v.at(x)));
        return make(v);
    }
    CtxtBatch add(const CtxtBatch& a, const Ptxt& p) {
        return rows1(a, [&](const fhelin_ct* const* hs, int32_t n, fhelin_ct** outs) { return fhelin_add_plain_batch(ctx(), hs, n, p->h, outs); }, "EvalAdd");
    }
    CtxtBatch mult(const CtxtBatch& a, const CtxtBatch& b) {
        auto ha = handles(a), hb = handles(b);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(fhelin_mult_batch(ctx(), ha.data(), hb.data(), B_, outs.data()), "EvalMult");
        return wrap(outs);
    }
    CtxtBatch mult(const CtxtBatch& a, const Ptxt& p) {
        return rows1(a, [&](const fhelin_ct* const* hs, int32_t n, fhelin_ct** outs) { return fhelin_mult_plain_batch(ctx(), hs, n, p->h, outs); }, "EvalMult");
    }
    CtxtBatch rotate(const CtxtBatch& a, int index) {
        return rows1(a, [&](const fhelin_ct* const* hs, int32_t n, fhelin_ct** outs) { return fhelin_rotate_batch(ctx(), hs, n, index, outs); }, "EvalRotate");
    }
    CtxtBatch rotsum(const CtxtBatch& a, int slots, int padding) {
        return rows1(a, [&](const fhelin_ct* const* hs, int32_t n, fhelin_ct** outs) { return fhelin_fc_rotsum_batch(ctx(), hs, n, slots, padding, 0, outs); }, "rotsum");
    }
    CtxtBatch bootstrap(const CtxtBatch& a, bool timing = false) {
        // per-handle calls: deferred by the C ABI and evaluated as one batch (the driver's loop over containers x B samples)
        vector<Ctxt> v;
        for (auto& h : a->v) v.push_back(c_.bootstrap(h, timing));
        return make(v);
    }

    /* matmuls (reference :869-1058) */
    vector<CtxtBatch> matmulRE(const vector<CtxtBatch>& rows, const Ptxt& weight, const Ptxt& bias) { return mm_pt(rows, weight, bias, 128, 128); }
    vector<CtxtBatch> matmulRE(const vector<CtxtBatch>& rows, const Ptxt& weight, const Ptxt& bias, int row_size, int padding) {
        return mm_pt(rows, weight, bias, row_size, padding);
    }
    vector<CtxtBatch> matmulRE(const vector<CtxtBatch>& rows, const CtxtBatch& weight, int row_size, int padding) {
        return mm_ct(rows, weight, row_size, padding);
    }
    vector<CtxtBatch> matmulCR(const vector<CtxtBatch>& rows, const Ptxt& weight, const Ptxt& bias) { return mm_pt(rows, weight, bias, 128, 1); }
    vector<CtxtBatch> matmulCR(const vector<CtxtBatch>& rows, const CtxtBatch& matrix) { return mm_ct(rows, matrix, 64, 1); }
    vector<CtxtBatch> matmulRElarge(vector<CtxtBatch>& rows, const vector<Ptxt>& weight, const Ptxt& bias, double mask_value = 1) {
        auto hs = flat(rows);
        vector<const fhelin_pt*> ws;
        for (auto& w : weight) ws.push_back(w->h);
        vector<fhelin_ct*> outs(hs.size());
        fhelin_shim::check(fhelin_fc_matmulRElarge(ctx(), hs.data(), (int32_t)hs.size(), ws.data(), (int32_t)ws.size(), bias ? bias->h : nullptr,
                                                   mask_value, outs.data()), "matmulRElarge");
        return unflat(outs, rows.size());
    }
    vector<CtxtBatch> matmulCRlarge(const vector<vector<CtxtBatch>>& rows, const vector<Ptxt>& weights, const Ptxt& bias) {
        vector<const fhelin_ct*> hs;
        for (int x = 0; x < B_; x++)
            for (auto& r : rows)
                for (int j = 0; j < 4; j++) hs.push_back(r.at(j)->v.at(x)->h);
        vector<const fhelin_pt*> ws;
        for (auto& w : weights) ws.push_back(w->h);
        if (ws.size() < 4) throw std::runtime_error("matmulCRlarge: need 4 weight blocks");
        vector<fhelin_ct*> outs(rows.size() * B_);
        fhelin_shim::check(fhelin_fc_matmulCRlarge(ctx(), hs.data(), (int32_t)(rows.size() * B_), ws.data(), bias ? bias->h : nullptr, outs.data()),
                           "matmulCRlarge");
        return unflat(outs, rows.size());
    }
    CtxtBatch matmulScores(const vector<CtxtBatch>& queries, const CtxtBatch& key) {
        auto hs = flat(queries);
        auto hk = handles(key);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(fhelin_fcb_matmulScores(ctx(), hs.data(), (int32_t)queries.size(), hk.data(), B_, outs.data()), "matmulScores");
        return wrap(outs);
    }
    CtxtBatch matmulScores(const CtxtBatch& query, const CtxtBatch& key) { return matmulScores(vector<CtxtBatch>{query}, key); }

    /* layout shuffles (reference :1060-1205) */
    CtxtBatch wrapUpRepeated(const vector<CtxtBatch>& vectors) {
        auto hs = flat(vectors);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(fhelin_fcb_wrapUpRepeated(ctx(), hs.data(), (int32_t)vectors.size(), B_, outs.data()), "wrapUpRepeated");
        return wrap(outs);
    }
    CtxtBatch wrapUpExpanded(const vector<CtxtBatch>& vectors) {
        auto hs = flat(vectors);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(fhelin_fcb_wrapUpExpanded(ctx(), hs.data(), (int32_t)vectors.size(), B_, outs.data()), "wrapUpExpanded");
        return wrap(outs);
    }
    vector<CtxtBatch> unwrapExpanded(const CtxtBatch& c, int inputs_num) {
        auto hs = handles(c);
        vector<fhelin_ct*> outs((size_t)B_ * inputs_num);
        fhelin_shim::check(fhelin_fcb_unwrapExpanded(ctx(), hs.data(), B_, inputs_num, outs.data()), "unwrapExpanded");
        return unflat(outs, inputs_num);
    }
    vector<vector<CtxtBatch>> unwrapRepeatedLarge(const vector<CtxtBatch>& c, int input_number) {
        auto hs = flat(c);
        vector<fhelin_ct*> outs((size_t)B_ * input_number * 4);
        fhelin_shim::check(fhelin_fcb_unwrapRepeatedLarge(ctx(), hs.data(), (int32_t)c.size(), B_, input_number, outs.data()), "unwrapRepeatedLarge");
        vector<vector<CtxtBatch>> res;
        for (int i = 0; i < input_number; i++) {
            vector<CtxtBatch> four;
            for (int k = 0; k < 4; k++) {
                vector<Ctxt> per;
                for (int x = 0; x < B_; x++) per.push_back(one_ct(outs[((size_t)x * input_number + i) * 4 + k]));
                four.push_back(make(per));
            }
            res.push_back(four);
        }
        return res;
    }
    vector<CtxtBatch> generate_containers(const vector<CtxtBatch>& inputs, const Ptxt& bias) {
        auto hs = flat(inputs);
        const size_t per = (inputs.size() + 31) / 32;
        vector<fhelin_ct*> outs(per * B_);
        int32_t n = 0;
        fhelin_shim::check(fhelin_fcb_generate_containers(ctx(), hs.data(), (int32_t)inputs.size(), B_, bias ? bias->h : nullptr, outs.data(), &n),
                           "generate_containers");
        outs.resize((size_t)n * B_);
        return unflat(outs, (size_t)n);
    }

    /* activations (reference :1289-1336) */
    CtxtBatch eval_exp(const CtxtBatch& c, int inputs_number) {
        const double coeffs[7] = {1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0};
        auto hs = handles(c);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(fhelin_eval_poly_batch(ctx(), hs.data(), B_, coeffs, 7, outs.data()), "EvalPoly");
        CtxtBatch res = wrap(outs);
        vector<const fhelin_ct*> eight;
        for (auto& h : res->v)
            for (int k = 0; k < 8; k++) eight.push_back(h->h);
        fhelin_shim::check(fhelin_mult_many_batch(ctx(), eight.data(), 8, B_, outs.data()), "EvalMultMany");
        res = wrap(outs);
        vector<double> m;
        for (int i = 0; i < num_slots; i++) m.push_back((i % 128 < inputs_number && i < (128 * inputs_number)) ? 0 : -1);
        return add(res, c_.encode(m, (int)res->GetLevel(), num_slots));
    }
    CtxtBatch eval_inverse_naive(const CtxtBatch& c, double min, double max) { return each(c, [&](const Ctxt& h) { return c_.eval_inverse_naive(h, min, max); }); }
    CtxtBatch eval_gelu_function(const CtxtBatch& c, double min, double max, double mult, int degree) {
        return each(c, [&](const Ctxt& h) { return c_.eval_gelu_function(h, min, max, mult, degree); });
    }
    CtxtBatch eval_tanh_function(const CtxtBatch& c, double min, double max, double mult, int degree) {
        return each(c, [&](const Ctxt& h) { return c_.eval_tanh_function(h, min, max, mult, degree); });
    }

private:
    FHEController& c_;
    int B_;
    fhelin_ctx* ctx() { return c_.engine(); }
    void need_B(size_t n) const {
        if ((int)n != B_) throw std::runtime_error("FHEControllerBatch: one item per sample expected");
    }
    Ctxt one_ct(fhelin_ct* h) { return std::make_shared<fhelin_shim::CiphertextImpl>(ctx(), h); }
    CtxtBatch wrap(const vector<fhelin_ct*>& outs) {
        vector<Ctxt> v;
        for (auto* h : outs) v.push_back(one_ct(h));
        return make(v);
    }
    vector<const fhelin_ct*> handles(const CtxtBatch& b) const {
        need_B(b->v.size());
        vector<const fhelin_ct*> hs;
        for (auto& h : b->v) hs.push_back(h->h);
        return hs;
    }
    // rows x samples -> the sample-major handle list the C ABI takes: v[x * n + i] = row i of sample x
    vector<const fhelin_ct*> flat(const vector<CtxtBatch>& rows) const {
        vector<const fhelin_ct*> hs;
        for (int x = 0; x < B_; x++)
            for (auto& r : rows) {
                need_B(r->v.size());
                hs.push_back(r->v[x]->h);
            }
        return hs;
    }
    vector<CtxtBatch> unflat(const vector<fhelin_ct*>& outs, size_t n) {
        vector<CtxtBatch> res;
        for (size_t i = 0; i < n; i++) {
            vector<Ctxt> per;
            for (int x = 0; x < B_; x++) per.push_back(one_ct(outs[(size_t)x * n + i]));
            res.push_back(make(per));
        }
        return res;
    }
    template <class F> CtxtBatch rows1(const CtxtBatch& a, F f, const char* what) {
        auto hs = handles(a);
        vector<fhelin_ct*> outs(B_);
        fhelin_shim::check(f(hs.data(), (int32_t)B_, outs.data()), what);
        return wrap(outs);
    }
    // per-handle calls whose evaluation the C ABI defers and batches (bootstraps, Chebyshev evaluations)
    template <class F> CtxtBatch each(const CtxtBatch& c, F f) {
        vector<Ctxt> v;
        for (auto& h : c->v) v.push_back(f(h));
        return make(v);
    }
    vector<CtxtBatch> mm_pt(const vector<CtxtBatch>& rows, const Ptxt& w, const Ptxt& bias, int slots, int padding) {
        auto hs = flat(rows);
        vector<fhelin_ct*> outs(hs.size());
        fhelin_shim::check(fhelin_fc_matmul_pt(ctx(), hs.data(), (int32_t)hs.size(), w->h, bias ? bias->h : nullptr, slots, padding, outs.data()), "matmul");
        return unflat(outs, rows.size());
    }
    vector<CtxtBatch> mm_ct(const vector<CtxtBatch>& rows, const CtxtBatch& w, int slots, int padding) {
        auto hs = flat(rows);
        vector<const fhelin_ct*> ws;
        for (int x = 0; x < B_; x++)
            for (size_t i = 0; i < rows.size(); i++) ws.push_back(w->v.at(x)->h);
        vector<fhelin_ct*> outs(hs.size());
        fhelin_shim::check(fhelin_fcb_matmul_ct(ctx(), hs.data(), ws.data(), (int32_t)hs.size(), slots, padding, outs.data()), "matmul");
        return unflat(outs, rows.size());
    }
};

#endif /* FHELIN_FHECONTROLLER_BATCH_H */
