// Exercises include/FHEController.h the way the reference's main.cpp does (same calls, reference parameter
// set N=2^15 / 28+7 limbs): text-file inputs -> read_* packing -> matmulRE / wrapUpRepeated / matmulScores /
// eval_exp / rotsum -> decrypt.  tests/test_shim_gpu.py writes the inputs, runs this binary on the GPU and
// compares the dumped slots with oracle/slotsim.py.
#include "FHEController.h"

FHEController controller;

static void dump(const string& path, const vector<double>& v) {
    ofstream f(path);
    f << setprecision(17);
    for (size_t i = 0; i < v.size(); i++) f << v[i] << (i + 1 == v.size() ? "\n" : ",");
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const string dir = argv[1];
    controller.parameters_folder = "keys";
    controller.generate_context(false, false);
    controller.generate_rotation_keys({1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, -1, -2, -4, -8 - 16, -32, -64});

    vector<Ctxt> inputs;
    for (int i = 0; i < 3; i++) inputs.push_back(controller.read_expanded_input(dir + "/input_" + to_string(i) + ".txt"));
    Ptxt w = controller.read_plain_input(dir + "/W_T.txt");
    Ptxt b = controller.read_plain_repeated_input(dir + "/bias.txt");
    vector<Ctxt> Q = controller.matmulRE(inputs, w, b);                 // main.cpp:183
    for (int i = 0; i < 3; i++) dump(dir + "/Q_" + to_string(i) + ".out", controller.decrypt_tovector(Q[i], 16384));

    Ctxt K_wrapped = controller.wrapUpRepeated(Q);                       // main.cpp:186
    dump(dir + "/K_wrapped.out", controller.decrypt_tovector(K_wrapped, 16384));
    Ctxt scores = controller.matmulScores(Q[0], K_wrapped);              // main.cpp:196
    dump(dir + "/scores.out", controller.decrypt_tovector(scores, 16384));
    Ctxt e = controller.eval_exp(scores, 3);                             // main.cpp:197
    dump(dir + "/exp.out", controller.decrypt_tovector(e, 16384));
    Ctxt s = controller.rotsum(e, 32, 128);                              // main.cpp:201
    dump(dir + "/sum.out", controller.decrypt_tovector(s, 16384));

    // handle semantics main.cpp relies on: GetLevel, Clone, nullptr plaintexts, save/load
    Ptxt null_bias = nullptr;
    vector<Ctxt> cr = controller.matmulCR({Q[0]}, w, null_bias);         // main.cpp:234-235
    Ctxt copy = cr[0]->Clone();
    controller.save(copy, dir + "/ct.bin");
    Ctxt back = controller.load_ciphertext(dir + "/ct.bin");
    dump(dir + "/cr.out", controller.decrypt_tovector(back, 16384));
    Ctxt missing = controller.load_ciphertext(dir + "/does_not_exist.bin");   // prints a message, returns a null handle
    ofstream(dir + "/levels.out") << Q[0]->GetLevel() << "," << scores->GetLevel() << "," << e->GetLevel() << "," << (missing == nullptr) << "\n";
    return 0;
}
