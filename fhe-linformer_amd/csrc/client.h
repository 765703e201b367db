// Client-side CKKS: key generation, encode/decode (special FFT, fp64), public-key encryption, decryption.
// Mirrors reference FHEController::generate_context / generate_*_keys / encode / encrypt / decrypt
// (src/FHEController.cpp:47-49, :237-273, :348-404).  Sampling and the fp64 FFT run on the host (as in the
// reference); every residue-polynomial operation (NTT, dyadic products) runs on the GPU kernels.
#pragma once
#include <vector>
#include "evaluator.h"
#include "kernels_client.h"

namespace fhelin {

// Cryptographic pseudo-random generator: the ChaCha20 stream (RFC 8439 block function, 20 rounds) keyed with a 256-bit
// secret seed; 64-bit block counter (state words 12-13), 64-bit stream id (words 14-15).  Every secret the client samples
// (secret key, encryption randomness, key-switching-key noise and masks) comes from this stream.
struct Prng {
    static constexpr int LANES = 8;                // blocks produced per refill
    explicit Prng(const uint8_t seed[32], u64 stream = 0);
    u64 next();
    u64 uniform(u64 q);        // unbiased in [0, q)
    double normal();           // standard normal (Box-Muller)
    // one 64-byte block for the given (counter, stream): known-answer tests against RFC 8439
    static void block(const uint8_t seed[32], u64 counter, u64 stream, uint8_t out[64]);
private:
    u32 key_[8];
    u64 counter_ = 0, stream_ = 0;
    u32 buf_[16 * LANES];
    int pos_ = 16 * LANES;     // u32 words consumed of buf_
    void refill();
    bool have_spare = false;
    double spare = 0;
};

class Client {
public:
    Client(Evaluator& ev, const uint8_t seed[32]);
    ~Client();
    void keygen();                       // secret (sparse ternary) + public key
    bool has_keys() const { return s_all != nullptr; }
    void gen_relin_key();                // EvalMultKeyGen
    void gen_rotation_key(int index);    // EvalRotateKeyGen for one index
    void gen_conj_key();
    KeyPtr make_switch_key(const u64* s_from_all, const u64* s_to_all);  // device [L+1+k][N] NTT form

    PtPtr encode(const double* vals, int n, int level, int slots);
    CtPtr encrypt(const PtPtr& p, int drop = 0);   // drop: limbs left out below the plaintext's level (level plan)
    // n_vec real vectors of n_per values each (row-major) -> n_vec fresh ciphertexts at `level`: encoding (special FFT, scaling,
    // rounding), sampling of (u, e0, e1) and the dyadic combination all on the GPU, in batched launches
    std::vector<CtPtr> encrypt_batch(const double* vals, int n_vec, int n_per, int level, int slots);
    std::vector<CtPtr> ingest_sample(const double* emb, const int* tokens, const double* table, int vocab, int S, const double* cls,
                                     const double* pos, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                                     int w_cols, int level, const std::vector<int>& drop, std::vector<double>* proj_out = nullptr);
    // test hook: the sampler's raw output, n_poly polynomials of N centred coefficients (kind 0 Gaussian, 1 ternary)
    std::vector<long> debug_sample(int kind, int n_poly);
    std::vector<double> decrypt(const CtPtr& c, int slots);
    CtPtr phase(const CtPtr& c, int nlimbs);   // c0 + c1 s (+ c2 s^2) on the first nlimbs limbs, NTT form, 1 component

    // raw import/export of key material (parity tests feed identical arrays to the oracle)
    void export_secret(u64* out);        // [L+1+k][N]
    void import_secret(const u64* in);   // replaces the secret (NTT form) — tests only

private:
    Evaluator& ev_;
    Context& c_;
    Prng rng_;
    u64* s_all = nullptr;   // secret, NTT form over Q and P limbs [L+1+k][N]
    u64* pk = nullptr;      // [2][L+1][N]: b = -a s + e, a
    void sample_small_to_ntt(u64* dst, int nlimbs_q, bool with_p, int kind);  // kind 0 gaussian, 1 ternary (host sampler: key generation)
    // device sampler (encryption randomness): dst [n_poly][ell][N] coefficient form; a fresh ChaCha20 key per call
    void sample_small_device(u64* dst, int n_poly, int ell, int kind);
    // c0 = b u + e0 + m, c1 = a u + e1 for n_vec encodings enc [n_vec][ell][N] (enc_stride words apart; 0 = one shared encoding)
    void encrypt_encoded(const u64* enc, size_t enc_stride, int n_vec, int ell, long double scale, int slots, std::vector<CtPtr>& out);
    u64 sample_calls_ = 0;
};

// special FFT helpers (shared by encode/decode); slots must be a power of two
void ckks_fft_special(std::vector<std::pair<double, double>>& v, bool inverse);
// tables of the special FFT for `slots` slots: rot[j] = 5^j mod 4*slots, ksi[k] = exp(2 pi i k / (4*slots))
void ckks_fft_tables(int slots, std::vector<u32>& rot, std::vector<std::pair<double, double>>& ksi);
std::shared_ptr<Encoding> encode_to_device(Context& c, const std::vector<double>& values, const std::vector<double>& imag, int slots,
                                           int ell, long double scale);
// device encoder for n_vec vectors: re / im [n_vec][n_per] (im may be null) -> dst [n_vec][ell][N] NTT form
void encode_batch_device(Context& c, u64* dst, const double* re, const double* im, int n_vec, int n_per, int slots, int ell, long double scale);
void encode_complex_on_device(Context& c, u64* dst, double* dv, int n_vec, int slots, int ell, long double scale);

}  // namespace fhelin
