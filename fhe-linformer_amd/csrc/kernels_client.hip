// Client-side work on the GPU: CKKS encoding and public-key encryption of whole batches.
//
// Reference side: FHEController::encode / encrypt / read_expanded_input (reference src/FHEController.cpp:348-385,
// :623-650) run one MakeCKKSPackedPlaintext + Encrypt per input on the host; a sample of the Linformer driver has 194
// inputs (src/main.cpp:159-173).  Here the inverse special FFT (fp64), the scaling / rounding to integers, the sampling of
// the encryption randomness and the final dyadic combination run as batched kernels over [vectors][...] arrays.
//
//  * fft_special_inv_stage_kernel : one radix-2 stage of the inverse special FFT (canonical embedding restricted to <5>),
//    bit-identical to the host loop in client.cpp ckks_fft_special(inverse): IEEE double operations in the same order,
//    contraction to FMA switched off.
//  * encode_round_reduce_kernel   : bit-reversal, 1/n scaling, (long double)v * Delta exactly as the x87 host code rounds
//    it (64-bit significand, round-to-nearest-even), round-half-away to an integer, residues modulo every limb.
//  * sample_small_kernel          : ChaCha20 (RFC 8439 block function) keyed per call; ternary {-1,0,1} or rounded
//    Gaussian (sigma 3.19, Box-Muller) coefficients written as residues of every limb.
//  * encrypt_combine_kernel       : c0 = b*u + e0 + m, c1 = a*u + e1 (NTT form) for a batch.
// All HBM-bound or trivially small; no MFMA.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "kernels_client.h"

namespace fhelin {
namespace {

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ Barrett load_barrett(const DeviceTables& t, int limb) {
    Barrett b;
    b.q = t.moduli[limb];
    b.r0 = t.barrett[2 * limb];
    b.r1 = t.barrett[2 * limb + 1];
    return b;
}

// ---- inverse special FFT, one stage.  data [n_vec][size] (re, im) pairs; grid (size/2 / 256, n_vec)
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void fft_special_inv_stage_kernel(double2* data, const u32* rot, const double2* ksi, int size, int len) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= size / 2) return;
    double2* v = data + (size_t)blockIdx.y * size;
    const int lenh = len >> 1, lenq = len << 2, gap = (4 * size) / lenq;
    const int j = t % lenh, i = (t / lenh) * len;
    const int idx = (lenq - (int)(rot[j] % (u32)lenq)) * gap;
    const double2 a = v[i + j], b = v[i + j + lenh], k = ksi[idx];
    const double dr = a.x - b.x, di = a.y - b.y;
    double2 u, w;
    u.x = a.x + b.x;
    u.y = a.y + b.y;
    w.x = dr * k.x - di * k.y;   // (dr + i di)(k.x + i k.y), each product and sum rounded separately like the host code
    w.y = dr * k.y + di * k.x;
    v[i + j] = u;
    v[i + j + lenh] = w;
}

// |v| * s with v a double and s = ms * 2^es a positive long double (64-bit significand): the product rounded to a 64-bit
// significand (round to nearest even), then rounded half away from zero to an integer -> magnitude as (lo, hi)
__device__ __forceinline__ void x87_mul_round(double v, u64 ms, int es, u64& lo, u64& hi, bool& neg) {
    const u64 bits = (u64)__double_as_longlong(v);
    neg = (bits >> 63) != 0;
    const int ex = (int)((bits >> 52) & 0x7FF);
    u64 mv = bits & 0xFFFFFFFFFFFFFull;
    lo = hi = 0;
    if (ex == 0 && mv == 0) return;             // +-0
    int ev;
    if (ex == 0) ev = -1074;                     // subnormal
    else {
        mv |= 1ull << 52;
        ev = ex - 1075;
    }
    // P = mv * ms < 2^117
    u64 plo = mv * ms, phi = __umul64hi(mv, ms);
    // significant bits of P
    const int nb = phi ? 128 - __clzll((long long)phi) : 64 - __clzll((long long)plo);
    int E = ev + es;                             // value = P * 2^E
    u64 q;                                       // 64-bit significand after the first rounding
    if (nb > 64) {
        const int sh = nb - 64;                  // 1..53
        q = (plo >> sh) | (phi << (64 - sh));
        const u64 rem = plo & ((1ull << sh) - 1), half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) {
            ++q;
            if (q == 0) {                        // carried out of 64 bits
                q = 1ull << 63;
                ++E;
            }
        }
        E += sh;
    } else {
        q = plo;                                 // exact in 64 bits
    }
    // round q * 2^E half away from zero to an integer (|result| < 2^127 for every scale this library uses)
    if (E >= 0) {
        if (E >= 64) {
            hi = q << (E - 64);
            lo = 0;
        } else if (E == 0) {
            lo = q;
        } else {
            lo = q << E;
            hi = q >> (64 - E);
        }
    } else {
        const int t = -E;
        if (t > 64) return;                      // below 1/2
        if (t == 64) {
            lo = q >> 63;                        // >= 1/2 rounds to 1
            return;
        }
        lo = q >> t;
        if ((q >> (t - 1)) & 1) {
            ++lo;
            if (lo == 0) hi = 1;
        }
    }
}

// grid (N/256, n_vec): thread = coefficient position n of vector b.  fftdata [n_vec][slots] is the output of the stage
// kernels (before bit reversal and 1/n scaling); out [n_vec][ell][N] residues (coefficient form)
__global__ __launch_bounds__(256) void encode_round_reduce_kernel(DeviceTables t, u64* out, const double2* fftdata, int slots, int log_slots,
                                                                  int ell, u64 ms, int es, double inv_n) {
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t gapc = (N / 2) / slots;
    u64* o = out + (size_t)blockIdx.y * ell * N + n;
    const size_t half = N / 2;
    const size_t pos = n < half ? n : n - half;
    if (pos % gapc != 0) {
        for (int l = 0; l < ell; ++l) o[(size_t)l * N] = 0;
        return;
    }
    const u32 i = (u32)(pos / gapc);
    const u32 src = __brev(i) >> (32 - log_slots);     // bit_reverse(v): element i comes from position bitrev(i)
    const double2 c = fftdata[(size_t)blockIdx.y * slots + src];
    double val;
    {
#pragma clang fp contract(off)
        val = (n < half ? c.x : c.y) * inv_n;
    }
    u64 lo, hi;
    bool neg;
    x87_mul_round(val, ms, es, lo, hi, neg);
    for (int l = 0; l < ell; ++l) {
        const Barrett br = load_barrett(t, l);
        const u64 h = barrett_reduce128(hi, 0, br);
        u64 r = barrett_reduce128(lo, h, br);
        if (neg) r = neg_mod(r, br.q);
        o[(size_t)l * N] = r;
    }
}

// ---- ChaCha20
__device__ __forceinline__ u32 rotl32(u32 x, int k) { return (x << k) | (x >> (32 - k)); }
#define FHELIN_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
__device__ __forceinline__ void chacha20_block(const SamplerKey& k, u64 counter, u64 stream, u64 (&out)[8]) {
    u32 x[16], in[16];
    in[0] = 0x61707865u; in[1] = 0x3320646eu; in[2] = 0x79622d32u; in[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) in[4 + i] = k.w[i];
    in[12] = (u32)counter; in[13] = (u32)(counter >> 32); in[14] = (u32)stream; in[15] = (u32)(stream >> 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = in[i];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        FHELIN_QR(x[0], x[4], x[8], x[12]) FHELIN_QR(x[1], x[5], x[9], x[13]) FHELIN_QR(x[2], x[6], x[10], x[14]) FHELIN_QR(x[3], x[7], x[11], x[15])
        FHELIN_QR(x[0], x[5], x[10], x[15]) FHELIN_QR(x[1], x[6], x[11], x[12]) FHELIN_QR(x[2], x[7], x[8], x[13]) FHELIN_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (u64)(x[2 * i] + in[2 * i]) | ((u64)(x[2 * i + 1] + in[2 * i + 1]) << 32);
}

// grid (N/8/256, n_poly): one ChaCha20 block = 8 coefficients per thread.  out [n_poly][ell][N] residues (coefficient form)
// kind 0: rounded Gaussian sigma 3.19 (Box-Muller on 53-bit uniforms); kind 1: uniform ternary
__global__ __launch_bounds__(256) void sample_small_kernel(DeviceTables t, u64* out, SamplerKey key, u64 stream_base, int kind, int ell) {
    const size_t N = (size_t)1 << t.log_n;
    const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= N / 8) return;
    u64 r[8];
    chacha20_block(key, blk, stream_base + blockIdx.y, r);
    int v[8];
    if (kind == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (int)__umul64hi(r[i], 3) - 1;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double u1 = ((double)(r[2 * i] >> 11) + 1.0) * (1.0 / 9007199254740992.0);   // (0, 1]
            const double u2 = (double)(r[2 * i + 1] >> 11) * (1.0 / 9007199254740992.0);       // [0, 1)
            const double rad = sqrt(-2.0 * log(u1)) * 3.19, th = 6.283185307179586476925 * u2;
            double sn, cs;
            sincos(th, &sn, &cs);
            v[2 * i] = (int)llrint(rad * cs);
            v[2 * i + 1] = (int)llrint(rad * sn);
        }
    }
    u64* o = out + (size_t)blockIdx.y * ell * N + blk * 8;
    for (int l = 0; l < ell; ++l) {
        const u64 q = t.moduli[l];
        u64x2* dst = reinterpret_cast<u64x2*>(o + (size_t)l * N);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u64x2 w;
            w.x = v[2 * i] < 0 ? q - (u64)(-v[2 * i]) : (u64)v[2 * i];
            w.y = v[2 * i + 1] < 0 ? q - (u64)(-v[2 * i + 1]) : (u64)v[2 * i + 1];
            dst[i] = w;
        }
    }
}

// grid (N/512, n_vec * ell): ct [n_vec][2][ell][N] <- (pk_b * u + e0 + m, pk_a * u + e1); pk [2][L1][N]
__global__ __launch_bounds__(256) void encrypt_combine_kernel(DeviceTables t, u64* ct, const u64* pk, const u64* u, const u64* e0, const u64* e1,
                                                              const u64* m, int ell, int L1, size_t m_stride) {
    const int b = blockIdx.y / ell, l = blockIdx.y % ell;
    const Barrett br = load_barrett(t, l);
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t in = ((size_t)b * ell + l) * row + n2;
    const u64x2 uu = reinterpret_cast<const u64x2*>(u)[in];
    const u64x2 a0 = reinterpret_cast<const u64x2*>(e0)[in], a1 = reinterpret_cast<const u64x2*>(e1)[in];
    const u64x2 mm = reinterpret_cast<const u64x2*>(m + (size_t)b * m_stride)[(size_t)l * row + n2];
    const u64x2 kb = reinterpret_cast<const u64x2*>(pk)[(size_t)l * row + n2];
    const u64x2 ka = reinterpret_cast<const u64x2*>(pk)[((size_t)L1 + l) * row + n2];
    u64x2 c0, c1;
    c0.x = add_mod(add_mod(mul_mod(kb.x, uu.x, br), a0.x, br.q), mm.x, br.q);
    c0.y = add_mod(add_mod(mul_mod(kb.y, uu.y, br), a0.y, br.q), mm.y, br.q);
    c1.x = add_mod(mul_mod(ka.x, uu.x, br), a1.x, br.q);
    c1.y = add_mod(mul_mod(ka.y, uu.y, br), a1.y, br.q);
    u64x2* C = reinterpret_cast<u64x2*>(ct);
    C[((size_t)(2 * b) * ell + l) * row + n2] = c0;
    C[((size_t)(2 * b + 1) * ell + l) * row + n2] = c1;
}

}  // namespace

void launch_fft_special_inv(double* data, const u32* rot, const double* ksi, int slots, int n_vec, hipStream_t s) {
    const dim3 g((unsigned)((slots / 2 + 255) / 256), (unsigned)n_vec);
    for (int len = slots; len >= 2; len >>= 1)
        hipLaunchKernelGGL(fft_special_inv_stage_kernel, g, dim3(256), 0, s, reinterpret_cast<double2*>(data), rot,
                           reinterpret_cast<const double2*>(ksi), slots, len);
}
void launch_encode_round_reduce(const DeviceTables& t, u64* out, const double* fftdata, int slots, int ell, u64 scale_mant, int scale_exp,
                                int n_vec, hipStream_t s) {
    int log_slots = 0;
    while ((1 << log_slots) < slots) ++log_slots;
    hipLaunchKernelGGL(encode_round_reduce_kernel, dim3((1u << t.log_n) / 256, (unsigned)n_vec), dim3(256), 0, s, t, out,
                       reinterpret_cast<const double2*>(fftdata), slots, log_slots, ell, scale_mant, scale_exp, 1.0 / slots);
}
void launch_sample_small(const DeviceTables& t, u64* out, const SamplerKey& key, u64 stream_base, int kind, int ell, int n_poly, hipStream_t s) {
    const unsigned bx = (unsigned)(((1u << t.log_n) / 8 + 255) / 256);
    hipLaunchKernelGGL(sample_small_kernel, dim3(bx, (unsigned)n_poly), dim3(256), 0, s, t, out, key, stream_base, kind, ell);
}
void launch_encrypt_combine(const DeviceTables& t, u64* ct, const u64* pk, const u64* u, const u64* e0, const u64* e1, const u64* m, int ell,
                            int L1, size_t m_stride, int n_vec, hipStream_t s) {
    hipLaunchKernelGGL(encrypt_combine_kernel, dim3((1u << t.log_n) / 512, (unsigned)(n_vec * ell)), dim3(256), 0, s, t, ct, pk, u, e0, e1, m,
                       ell, L1, m_stride);
}


// ---- sample ingestion (kernels_client.h) ------------------------------------------------------------------------------
namespace {
__global__ void ingest_xin_kernel(double* __restrict__ x_in, const double* __restrict__ emb, const int* __restrict__ tokens,
                                  const double* __restrict__ table, const double* __restrict__ cls, const double* __restrict__ pos, int S) {
    const int t = blockIdx.x, j = threadIdx.x;           // row t of x_in (0 = CLS), feature j < 128
    if (t == 0) {
        x_in[j] = cls[j];
        return;
    }
    const double e = tokens ? table[(size_t)tokens[t - 1] * 128 + j] : emb[(size_t)(t - 1) * 128 + j];
    {
#pragma clang fp contract(off)
        const double p3 = pos[(size_t)(t - 1) * 128 + j] / 3.0;
        x_in[(size_t)t * 128 + j] = e + p3;
    }
}
__global__ void ingest_project_kernel(double* __restrict__ proj, const double* __restrict__ x_in, const double* __restrict__ E_w,
                                      const double* __restrict__ E_b, const double* __restrict__ F_w, const double* __restrict__ F_b,
                                      int w_cols, int S1) {
    const int r = blockIdx.x, j = threadIdx.x;           // r < 64: rows of E then rows of F
    const double* w = (r < 32 ? E_w : F_w) + (size_t)(r & 31) * w_cols;
    // every product and every sum rounded on its own, in this order (the header's __dmul_rn / __dadd_rn are plain operators that
    // the compiler is free to contract into FMAs; the pragma below is what forbids it)
    {
#pragma clang fp contract(off)
        double acc = w[0] * x_in[j];
        for (int t = 1; t < S1; ++t) {
            const double p = w[t] * x_in[(size_t)t * 128 + j];
            acc = acc + p;
        }
        proj[(size_t)r * 128 + j] = acc + (r < 32 ? E_b : F_b)[r & 31];
    }
}
__global__ void ingest_expand_kernel(double* __restrict__ out, const double* __restrict__ proj, const double* __restrict__ x_in, int S1, int slots) {
    const int v = blockIdx.y;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= slots) return;
    const double* src = v < 64 ? proj + (size_t)v * 128 : x_in + (size_t)(v - 64) * 128;
    const int j = slot >> 7;
    double2 o;
    o.x = j < 128 ? src[j] : 0.0;
    o.y = 0.0;
    reinterpret_cast<double2*>(out)[(size_t)v * slots + slot] = o;
}
}  // namespace
void launch_ingest_xin(double* x_in, const double* emb, const int* tokens, const double* table, const double* cls, const double* pos,
                       int S, hipStream_t s) {
    hipLaunchKernelGGL(ingest_xin_kernel, dim3(S + 1), dim3(128), 0, s, x_in, emb, tokens, table, cls, pos, S);
}
void launch_ingest_project(double* proj, const double* x_in, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                           int w_cols, int S1, hipStream_t s) {
    hipLaunchKernelGGL(ingest_project_kernel, dim3(64), dim3(128), 0, s, proj, x_in, E_w, E_b, F_w, F_b, w_cols, S1);
}
void launch_ingest_expand(double* out, const double* proj, const double* x_in, int S1, int slots, hipStream_t s) {
    hipLaunchKernelGGL(ingest_expand_kernel, dim3((slots + 255) / 256, 64 + S1), dim3(256), 0, s, out, proj, x_in, S1, slots);
}

}  // namespace fhelin
