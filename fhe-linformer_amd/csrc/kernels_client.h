// Launchers of the client-side kernels (kernels_client.hip): batched CKKS encoding and public-key encryption.
#pragma once
#include "kernels.h"

namespace fhelin {

struct SamplerKey {
    u32 w[8];  // ChaCha20 key words of one sampling call (drawn from the client's generator)
};

// data [n_vec][slots] complex (re, im): all stages of the inverse special FFT (without the bit reversal and the 1/n scaling,
// which launch_encode_round_reduce applies while reading).  rot [slots] = 5^j mod 4 slots, ksi [4 slots + 1] = e^{2 pi i k / (4 slots)}
void launch_fft_special_inv(double* data, const u32* rot, const double* ksi, int slots, int n_vec, hipStream_t s);
// out [n_vec][ell][N] (coefficient form) <- round(v * scale) mod q_l with scale = scale_mant * 2^scale_exp (64-bit significand)
void launch_encode_round_reduce(const DeviceTables& t, u64* out, const double* fftdata, int slots, int ell, u64 scale_mant, int scale_exp,
                                int n_vec, hipStream_t s);
// out [n_poly][ell][N] (coefficient form): kind 0 rounded Gaussian sigma 3.19, kind 1 uniform ternary; polynomial p uses
// ChaCha20 stream stream_base + p
void launch_sample_small(const DeviceTables& t, u64* out, const SamplerKey& key, u64 stream_base, int kind, int ell, int n_poly, hipStream_t s);
// ct [n_vec][2][ell][N] <- (pk_b u + e0 + m, pk_a u + e1); u, e0, e1 [n_vec][ell][N]; m at m + b * m_stride; pk [2][L1][N]
void launch_encrypt_combine(const DeviceTables& t, u64* ct, const u64* pk, const u64* u, const u64* e0, const u64* e1, const u64* m, int ell,
                            int L1, size_t m_stride, int n_vec, hipStream_t s);


// ---- client-side ingestion of one sample (reference src/python/dimReduce.py:141-160 and the read_expanded_input packing,
// src/FHEController.cpp:623-650), all in fp64 with the operation order of the NumPy statement and FMA contraction off:
//   x_in[0] = cls, x_in[t] = emb[t-1] + pos[t-1] / 3   (emb row = table[token[t-1]] when a token-id list is given)
//   X[i]    = (...((W[i][0] x_in[0]) + W[i][1] x_in[1]) + ...) + b[i]      for the 32 rows of E and of F (sequential sums)
//   slot j*128 + i of vector v = v[j]  (i < 128; "expanded" layout), written as complex doubles for the encoder's inverse FFT
// x_in [S1][128]; proj [64][128] (E rows then F rows); out [(64 + S1)][slots][2] in the order E rows, F rows, tokens.
void launch_ingest_xin(double* x_in, const double* emb, const int* tokens, const double* table, const double* cls, const double* pos,
                       int S, hipStream_t s);
void launch_ingest_project(double* proj, const double* x_in, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                           int w_cols, int S1, hipStream_t s);
void launch_ingest_expand(double* out, const double* proj, const double* x_in, int S1, int slots, hipStream_t s);
}  // namespace fhelin
