// K1 — batched negacyclic NTT / INTT over RNS limbs, hand-written for gfx950 (CDNA4).
//
// Replaces (reference side): every DCRTPoly::SetFormat(EVALUATION/COEFFICIENT) executed inside
// OpenFHE underneath context->EvalMult / EvalRotate / rescale (reference call sites
// src/FHEController.cpp:423-435, :833, :843).  SURVEY.md §8(a) row K1.
//
// Function computed (bit-exact contract, same as oracle/fhe_oracle.c orc_ntt_forward):
//   forward:  out[j] = sum_i a[i] * psi^{(2*bitrev(j)+1) * i}  mod q        (natural in, bit-reversed out)
//   inverse:  the inverse map, including the N^{-1} factor               (bit-reversed in, natural out)
//
// Decomposition: N = 2^A x 256.  One transform = two kernels ("passes"), each a coalesced tile read,
// up to nine radix-2 stages done in registers/LDS, and a coalesced tile write:
//   forward : pass A (column transforms over the high A index bits, all 256 columns share 2^A-1 twiddles)
//             pass B (row transforms over the low 8 index bits, one 256-entry twiddle slice per row)
//   inverse : pass B' then pass A' (Gentleman-Sande order), N^{-1} folded into the last stage.
// A workgroup (256 threads = 4 wavefronts) owns a tile of 4096 residues; each thread keeps 16 residues
// in VGPRs and runs 4 stages on them between LDS exchanges ("rounds").  Butterflies use Shoup twiddles with an
// approximate quotient (product in [0,5q), 11 VALU instructions) and carry NO conditional subtraction for primes
// below 2^53 (value bounds proven per stage, one reduction before the final store), one mask-form conditional
// subtraction per butterfly for the 55/60-bit primes.  This is 64-bit modular-integer work: no MFMA.
//
// Memory behaviour: every global access of a wave instruction is a run of whole 128-byte lines (column pass:
// 4 row segments of 128 B; row pass: 512 B contiguous, reached through one extra LDS exchange).  The first pass can
// read out of place and strided (LimbBatch::src / src_group), which is how key switching and rescale avoid staging
// copies.  The row pass maps workgroups to (limb, tile, repetition) in XCD-contiguous order so that the 4 KiB
// twiddle slice of a (limb, tile) is shared by the vectors of a batch in one L2.
// Measured (profiles/): 1.95 M limb-NTT/s at N=2^16, bound by VALU issue slots (88 % busy; 15 VALU
// instructions per lazy butterfly, 9 of them 32x32 multiplies), 2.07x algorithmic HBM traffic (two passes + twiddles).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "kernels.h"

namespace fhelin {

namespace {

struct NttArgs {
    u64* data;
    const u64* src;     // input of this pass (== data when in place)
    const u64* tw;      // [n_limbs][2N]
    const u64* tw_rows; // [n_limbs][N/4096][15][256][2]  row-pass layout of the per-thread stages
    const u64* moduli;  // [n_limbs]
    const u64* ninv;    // [n_limbs][8]: N^-1 constants (4), lazy-reduction shift and ratio (2), spare (2)
    const int* limb_tab;
    int tab_len;
    int limb_first;
    int limb_count;
    int log_n;
    int nvec;
    int period;  // > 0: vectors v and v + period use the same limb (twiddles); used to co-schedule them
    int lazy_out;             // forward: lazy-path limbs below 2^53 skip the final reduction (LimbBatch::lazy_out)
    int src_group;            // > 0: strided first-pass input (see LimbBatch)
    size_t src_group_stride;
    int src_group2;
    size_t src_group2_stride;
    const u64* lift_qlm;      // rescale: first-pass input = centred lift of src[vec / limb_count] (LimbBatch::lift_qlm)
    int lift_limb;
    int vec0;                 // first vector of this launch: a large transform runs as chunks of vectors, both passes per chunk (launch_ntt_impl)
};

__device__ __forceinline__ size_t src_offset(const NttArgs& a, int vec, int log_n) {
    if (a.src_group > 0) {
        const int g = vec / a.src_group;
        const size_t in = (size_t)(vec % a.src_group) << log_n;
        if (a.src_group2 > 0) return (size_t)(g / a.src_group2) * a.src_group2_stride + (size_t)(g % a.src_group2) * a.src_group_stride + in;
        return (size_t)g * a.src_group_stride + in;
    }
    return (size_t)vec << log_n;
}

constexpr int TILE = 4096;
constexpr int LDS_WORDS = TILE + TILE / 16;

// tile-local element index held by thread tau in register slot k when the 4-bit register window
// sits at bit p of the index
__device__ __forceinline__ int tile_index(int tau, int k, int p) {
    return ((tau >> p) << (p + 4)) | (k << p) | (tau & ((1 << p) - 1));
}
// one pad word per 16 residues keeps every exchange pattern used below bank-conflict free
__device__ __forceinline__ int lds_slot(int e) { return e + (e >> 4); }

__device__ __forceinline__ int limb_of(const NttArgs& a, int vec) {
    return a.limb_tab ? a.limb_tab[a.tab_len ? vec % a.tab_len : vec] : a.limb_first + (vec % a.limb_count);
}

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// Per-limb constants of a block.
struct LimbConst {
    u64 q;
    u64 nq;   // 2^64 - q
    u64 q5;   // 5q   (lazy path: upper bound of mul_shoup_lazy5)
    u32 sh;   // lazy path: reduce_lazy_2q shift and ratio
    u32 rr;
};

// Lazy path (q < 2^53).  No conditional subtraction inside the transform: the approximate-quotient Shoup product
// returns T in [0,5q) for any 64-bit input, a forward butterfly maps (X, Y) -> (X + T, X + 5q - T), so the value bound
// grows by 5q per stage: canonical input -> < (1 + 5*17)q = 86q < 2^60 after the 17 stages of N = 2^17.  The column
// pass stores these lazy values; the row pass reduces once before its final store.
//
// Forward (Cooley-Tukey) stages for register-index bits KB_HI-1 .. KB_LO (descending).
//   gbase : global index bit that register bit 0 of the window corresponds to
//   hi    : the thread's global index bits above the window (index >> (gbase+4))
// The (up to) 15 twiddles of one round, fetched as a block so that a kernel can issue the loads of its NEXT round before
// the LDS exchange (their L2 latency then hides behind the barrier instead of sitting on the critical path).
// Stage KB uses 8 >> KB twiddles, stored at t[(8 >> KB) - 1 ...].
struct RoundTw {
    u64x2 t[15];
};
template <int KB_LO, int KB_HI>
__device__ __forceinline__ void load_round_tw(RoundTw& r, const u64x2* __restrict__ tw, int log_n, int gbase, int hi) {
#pragma unroll
    for (int kb = KB_LO; kb < KB_HI; ++kb) {
        const int m = 1 << (log_n - (gbase + kb) - 1);
        const int tbase = m + (hi << (3 - kb));
#pragma unroll
        for (int j = 0; j < (8 >> kb); ++j) r.t[(8 >> kb) - 1 + j] = tw[tbase + j];
    }
}

// the per-thread twiddles of the row pass (global stage bits 0..3) from the transposed table: lanes read consecutive pairs
__device__ __forceinline__ void load_round_tw_rows(RoundTw& r, const u64x2* __restrict__ trows, int tau) {
#pragma unroll
    for (int s = 0; s < 15; ++s) r.t[s] = trows[s * 256 + tau];
}

template <int KB, bool LAZY>
__device__ __forceinline__ void fwd_stage(u64 (&x)[16], const RoundTw& tw, const LimbConst& c) {
#pragma unroll
    for (int k0 = 0; k0 < 16; ++k0) {
        if (k0 & (1 << KB)) continue;
        const int k1 = k0 | (1 << KB);
        const u64x2 w = tw.t[(8 >> KB) - 1 + (k0 >> (KB + 1))];
        const u64 X = LAZY ? x[k0] : csub_mask(x[k0], c.q5);
#if defined(FHELIN_BFLY15)  // A/B measurement builds only (tools/build_variant.sh): the 15-instruction form
        const u64 T = mul_shoup_lazy5(x[k1], w.x, w.y, c.nq);
        x[k0] = X + T;
        x[k1] = X + c.q5 - T;
#else
        // A = X + T comes out of the multiply-add chain itself; B = X + 5q - T = (2X + 5q) - A
        const u64 A = mul_shoup_lazy5_add(x[k1], w.x, w.y, c.nq, X);
        x[k0] = A;
        x[k1] = shl1_add(X, c.q5) - A;
#endif
    }
}
template <int KB_LO, int KB_HI, bool LAZY>
__device__ __forceinline__ void fwd_round(u64 (&x)[16], const RoundTw& tw, const LimbConst& c) {
    if constexpr (KB_LO < KB_HI) {
        fwd_stage<KB_HI - 1, LAZY>(x, tw, c);
        fwd_round<KB_LO, KB_HI - 1, LAZY>(x, tw, c);
    }
}

// Lazy inverse stages: (X, Y) -> (X + Y, (X + B_s - Y) * w) with the product in [0,5q) and no conditional subtraction.
// With inputs below c_s*q at stage s of a kernel, sums stay below 2*c_s*q, so c_{s+1} = max(2*c_s, 5) and B_s = c_s*q.
// A kernel runs at most 9 stages from c_0 <= 2: c_8 = 640, every intermediate < 1280q < 2^64 for q < 2^53.
constexpr int lazy_coef(int s, int c0) { return s == 0 ? c0 : 5 << (s - 1); }  // c0 <= 2

// Inverse (Gentleman-Sande) stages for register-index bits KB_LO .. KB_HI-1 (ascending).
// If LAST, the stage at bit KB_HI-1 is the final stage of the whole transform and carries N^{-1}.
// Lazy path: S0 = stages of this kernel already done, C0 = bound coefficient of the kernel's input (see lazy_coef).
template <int KB, bool FINAL, bool LAZY, int S, int C0>
__device__ __forceinline__ void inv_stage(u64 (&x)[16], const RoundTw& tw, const LimbConst& c, const u64* __restrict__ ninv) {
    const u64 B = c.q * (u64)lazy_coef(S, C0);  // lazy path only
    if constexpr (FINAL) {
        const u64 ni = ninv[0], nis = ninv[1], wn = ninv[2], wns = ninv[3];
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) {
            if (k0 & (1 << KB)) continue;
            const int k1 = k0 | (1 << KB);
            const u64 X = x[k0], Y = x[k1];
            if (LAZY) {
                x[k0] = csub_mask(reduce_lazy_2q(mul_shoup_lazy5(X + Y, ni, nis, c.nq), c.q, c.sh, c.rr), c.q);
                x[k1] = csub_mask(reduce_lazy_2q(mul_shoup_lazy5(X + B - Y, wn, wns, c.nq), c.q, c.sh, c.rr), c.q);
            } else {
                x[k0] = csub_mask(reduce_lazy_2q(mul_shoup_lazy5(X + Y, ni, nis, c.nq), c.q, c.sh, c.rr), c.q);
                x[k1] = csub_mask(reduce_lazy_2q(mul_shoup_lazy5(X + c.q5 - Y, wn, wns, c.nq), c.q, c.sh, c.rr), c.q);
            }
        }
    } else {
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) {
            if (k0 & (1 << KB)) continue;
            const int k1 = k0 | (1 << KB);
            const u64x2 w = tw.t[(8 >> KB) - 1 + (k0 >> (KB + 1))];
            const u64 X = x[k0], Y = x[k1];
            if (LAZY) {
                x[k0] = X + Y;
                x[k1] = mul_shoup_lazy5(X + B - Y, w.x, w.y, c.nq);
            } else {
                x[k0] = csub_mask(X + Y, c.q5);
                x[k1] = mul_shoup_lazy5(X + c.q5 - Y, w.x, w.y, c.nq);
            }
        }
    }
}
template <int KB_LO, int KB_HI, bool LAST, bool LAZY, int S0 = 0, int C0 = 1>
__device__ __forceinline__ void inv_round(u64 (&x)[16], const RoundTw& tw, const LimbConst& c, const u64* __restrict__ ninv) {
    if constexpr (KB_LO < KB_HI) {
        inv_stage<KB_LO, (LAST && KB_LO == KB_HI - 1), LAZY, S0, C0>(x, tw, c, ninv);
        inv_round<KB_LO + 1, KB_HI, LAST, LAZY, S0 + 1, C0>(x, tw, c, ninv);
    }
}

__device__ __forceinline__ void exchange(u64 (&x)[16], u64* lds, int tau, int p_from, int p_to, bool sync_first) {
    if (sync_first) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[lds_slot(tile_index(tau, k, p_from))] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[lds_slot(tile_index(tau, k, p_to))];
}

__device__ __forceinline__ LimbConst limb_const(const NttArgs& a, int limb) {
    LimbConst c;
    c.q = a.moduli[limb];
    c.nq = 0 - c.q;
    c.q5 = 5 * c.q;
    c.sh = (u32)a.ninv[8 * limb + 4];
    c.rr = (u32)a.ninv[8 * limb + 5];
    return c;
}
// the 52-bit scaling primes (most limbs) take the lazy path, the 60-bit special primes the semi-lazy path (one
// conditional subtraction per butterfly), the 55-bit first prime lazy forward / semi-lazy inverse; uniform per workgroup
#if defined(FHELIN_NTT_FORCE_PATH)  // ISA inspection builds only (tools/isa_count.py): 1 = lazy, 0 = classic
template <bool INVERSE> __device__ __forceinline__ bool lazy_prime(u64) { return FHELIN_NTT_FORCE_PATH; }
#else
// forward: bound 86q must fit 64 bits -> q < 2^57 (covers the 55-bit first prime too); inverse: 1280q -> q < 2^53
template <bool INVERSE> __device__ __forceinline__ bool lazy_prime(u64 q) { return q < (1ull << (INVERSE ? 53 : 57)); }
#endif

// ------------------------------------------------------------------------------------------------
// pass A: column transforms over the high A bits of the index.  Tile = 2^A rows x CW columns.
// tile-local index e = (row << LOGCW) | col ; global index j = (row << 8) | (tile*CW + col).
// ------------------------------------------------------------------------------------------------
template <int A, bool INVERSE, bool LAZY, bool LIFT>
__device__ __forceinline__ void cols_body(const NttArgs& a, u64* lds, int vec, int tile, int limb, const LimbConst& c) {
    constexpr int LOGCW = 12 - A;
    constexpr int CW = 1 << LOGCW;
    constexpr int LOGN = A + 8;
    constexpr int NFULL = A / 4;
    constexpr int REM = A % 4;
    const u64x2* tw = reinterpret_cast<const u64x2*>(a.tw) + ((size_t)limb << LOGN);
    u64* base = a.data + ((size_t)vec << LOGN) + tile * CW;
    const u64* sbase = (LIFT ? a.src + ((size_t)(vec / a.limb_count) << LOGN) : a.src + src_offset(a, vec, LOGN)) + tile * CW;
    const int tau = threadIdx.x;
    u64 x[16];

    // global offset (relative to base) of tile-local index e
    auto goff = [](int e) { return ((e >> LOGCW) << 8) | (e & (CW - 1)); };

    if (!INVERSE) {
        // windows (in tile-index bits), top down: LOGCW+A-4, LOGCW+A-8, ..., then the remainder at LOGCW
        constexpr int P0 = LOGCW + A - 4;
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[goff(tile_index(tau, k, P0))];
        if constexpr (LIFT) {
            // rescale: x is a coefficient modulo q_lift < 2 q (checked by the host): x mod q is one conditional subtraction,
            // the centred representative subtracts q_lift mod q where x > q_lift / 2  (== rescale_lift_kernel)
            const u64 half = a.moduli[a.lift_limb] >> 1, qlm = a.lift_qlm[limb];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const u64 r = csub_mask(x[k], c.q);
                x[k] = x[k] > half ? sub_mod(r, qlm, c.q) : r;
            }
        }
        RoundTw rt, rn;
        load_round_tw<0, 4>(rt, tw, LOGN, 8 + P0 - LOGCW, tau >> P0);
        fwd_round<0, 4, LAZY>(x, rt, c);
        int p_prev = P0;
        if constexpr (NFULL >= 2) {
            constexpr int P1 = LOGCW + A - 8;
            load_round_tw<0, 4>(rn, tw, LOGN, 8 + P1 - LOGCW, tau >> P1);
            exchange(x, lds, tau, p_prev, P1, false);
            fwd_round<0, 4, LAZY>(x, rn, c);
            p_prev = P1;
        }
        if constexpr (REM > 0) {
            constexpr int PR = LOGCW;
            load_round_tw<0, REM>(rt, tw, LOGN, 8, tau >> PR);
            exchange(x, lds, tau, p_prev, PR, NFULL >= 2);
            fwd_round<0, REM, LAZY>(x, rt, c);
            p_prev = PR;
        }
        // lazy path: values up to (1 + 5A)q go to memory as they are; the row pass bounds and reduces them
        constexpr int PL = REM > 0 ? LOGCW : (NFULL >= 2 ? LOGCW + A - 8 : P0);
        (void)p_prev;
#pragma unroll
        for (int k = 0; k < 16; ++k) base[goff(tile_index(tau, k, PL))] = x[k];
    } else {
        // inverse: bottom up.  full rounds at LOGCW, LOGCW+4, ...; remainder = top REM bits of the window at 8.
        // lazy path: the row pass hands over values below 2q (C0 = 2)
        const u64* ninv = a.ninv + 8 * limb;
        constexpr int P0 = LOGCW;
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[goff(tile_index(tau, k, P0))];
        RoundTw rt, rn;
        load_round_tw<0, 4>(rt, tw, LOGN, 8, tau >> P0);
        inv_round<0, 4, (NFULL == 1 && REM == 0), LAZY, 0, 2>(x, rt, c, ninv);
        int p_prev = P0;
        if constexpr (NFULL >= 2) {
            constexpr int P1 = LOGCW + 4;
            load_round_tw<0, 4>(rn, tw, LOGN, 8 + 4, tau >> P1);
            exchange(x, lds, tau, p_prev, P1, false);
            inv_round<0, 4, (REM == 0), LAZY, 4, 2>(x, rn, c, ninv);
            p_prev = P1;
        }
        if constexpr (REM > 0) {
            constexpr int PR = 8;  // window covers tile bits 8..11; its top REM bits are still to do
            load_round_tw<4 - REM, 4>(rt, tw, LOGN, 8 + PR - LOGCW, tau >> PR);
            exchange(x, lds, tau, p_prev, PR, NFULL >= 2);
            inv_round<4 - REM, 4, true, LAZY, 4 * NFULL, 2>(x, rt, c, ninv);
            p_prev = PR;
        }
        constexpr int PL = REM > 0 ? 8 : (NFULL >= 2 ? LOGCW + 4 : P0);
        (void)p_prev;
#pragma unroll
        for (int k = 0; k < 16; ++k) base[goff(tile_index(tau, k, PL))] = x[k];
    }
}

template <int A, bool INVERSE, bool LIFT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LIFT ? 3 : 4))) void ntt_cols_kernel(NttArgs a) {
    constexpr int LOGTILES = A - 4;
    __shared__ u64 lds[LDS_WORDS];
    const int vec = a.vec0 + (blockIdx.x >> LOGTILES);
    const int tile = blockIdx.x & ((1 << LOGTILES) - 1);
    const int limb = limb_of(a, vec);
    if (limb < 0) return;
    const LimbConst c = limb_const(a, limb);
    if (lazy_prime<INVERSE>(c.q))
        cols_body<A, INVERSE, true, LIFT>(a, lds, vec, tile, limb, c);
    else
        cols_body<A, INVERSE, false, LIFT>(a, lds, vec, tile, limb, c);
}

// ------------------------------------------------------------------------------------------------
// pass B: row transforms over the low 8 bits of the index.  Tile = 16 consecutive rows of 256
// (4096 contiguous residues); tile-local index e == global index - tile*4096.
// ------------------------------------------------------------------------------------------------
template <bool INVERSE, bool LAZY, bool MODDOWN>
__device__ __forceinline__ void rows_body(const NttArgs& a, u64* lds, int vec, int tile, int limb, const LimbConst& c,
                                          const NttModDown* md) {
    const int log_n = a.log_n;
    const u64x2* tw = reinterpret_cast<const u64x2*>(a.tw) + ((size_t)limb << log_n);
    const u64x2* trows = reinterpret_cast<const u64x2*>(a.tw_rows) + (((size_t)limb << (log_n - 12)) + tile) * (15 * 256);
    u64* base = a.data + ((size_t)vec << log_n) + ((size_t)tile << 12);
    const u64* sbase = a.src + src_offset(a, vec, log_n) + ((size_t)tile << 12);
    const int tau = threadIdx.x;
    u64 x[16];

    if (!INVERSE) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[tile_index(tau, k, 4)];
        RoundTw rt, rn;
        load_round_tw<0, 4>(rt, tw, log_n, 4, (tile << 4) | (tau >> 4));
        fwd_round<0, 4, LAZY>(x, rt, c);
        load_round_tw_rows(rn, trows, tau);                          // per-thread twiddles of the last four stages
        exchange(x, lds, tau, 4, 0, false);
        fwd_round<0, 4, LAZY>(x, rn, c);
        if (!(LAZY && !MODDOWN && a.lazy_out && c.q < (1ull << 53))) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                x[k] = csub_mask(reduce_lazy_2q(x[k], c.q, c.sh, c.rr), c.q);   // from < 86q (lazy) or < 10q (semi-lazy)
        }
#if defined(FHELIN_ROWS_DIRECT_STORE)   // A/B builds only: 16-byte stores straight from the bit-0 window (128-byte lane stride), no exchange
        if constexpr (!MODDOWN) {
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                u64x2 v;
                v.x = x[k];
                v.y = x[k + 1];
                reinterpret_cast<u64x2*>(base + 16 * tau)[k >> 1] = v;
            }
            return;
        }
#endif
        // window at bit 0 leaves 16 consecutive residues per thread (128-byte lane stride); one more LDS exchange
        // to the bit-8 window makes every store instruction a contiguous 512-byte wave access
        exchange(x, lds, tau, 0, 8, true);
        if constexpr (MODDOWN) {
            const int ell = md->ell;
            const int bi = vec / (2 * ell), comp = (vec / ell) & 1, t = vec % ell;
            const size_t n = (size_t)1 << log_n;
            const u64 pw = md->pinv[2 * t], pws = md->pinv[2 * t + 1];
            const u64* aq = md->accQ + ((size_t)vec << log_n) + ((size_t)tile << 12);
            const u64* add = comp == 0 ? md->add0 : md->add1;
            if (add) add += (size_t)bi * md->add_stride + (size_t)t * n + ((size_t)tile << 12);
            const u32* im = md->per_row ? md->invmap_row[bi] : md->invmap;
            const u64* post = md->post ? md->post + (size_t)bi * md->post_stride + (size_t)(comp * ell + t) * n : nullptr;
            u64* out = md->out + (size_t)bi * md->out_stride + (size_t)(comp * ell + t) * n;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = tile_index(tau, k, 8);
                u64 r = mul_shoup(sub_mod(aq[e], x[k], c.q), pw, pws, c.q);
                if (add) r = add_mod(r, add[e], c.q);
                const size_t m = ((size_t)tile << 12) + e;
                const size_t j = im ? (size_t)im[m] : m;
                if (post) r = add_mod(r, post[j], c.q);
                out[j] = r;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) base[tile_index(tau, k, 8)] = x[k];
        }
    } else {
        // coalesced load through the bit-8 window, then an LDS exchange to the bit-0 window of the first GS round
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[tile_index(tau, k, 8)];
        RoundTw rt, rn;
        load_round_tw_rows(rt, trows, tau);                          // per-thread twiddles of the first four stages
        exchange(x, lds, tau, 8, 0, false);
        inv_round<0, 4, false, LAZY, 0, 1>(x, rt, c, nullptr);
        load_round_tw<0, 4>(rn, tw, log_n, 4, (tile << 4) | (tau >> 4));
        exchange(x, lds, tau, 0, 4, true);
        inv_round<0, 4, false, LAZY, 4, 1>(x, rn, c, nullptr);
        if (LAZY) {  // sums have grown to < 640q: hand values below 2q to the column pass
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = reduce_lazy_2q(x[k], c.q, c.sh, c.rr);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) base[tile_index(tau, k, 4)] = x[k];
    }
}

template <bool INVERSE, bool MODDOWN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void ntt_rows_kernel(NttArgs a, NttModDown md) {
    __shared__ u64 lds[LDS_WORDS];
    const int logtiles = a.log_n - 12;
    // Block -> (vector, tile) map.  A row tile needs its own 4 KiB twiddle slice per (limb, tile); vectors of the same
    // limb (the two polys of a ciphertext, the rows of a batch) reuse it.  Blocks b and b+8 share an XCD (L2), so
    // each XCD gets a contiguous range of logical ids, ordered (limb, tile, repetition): the slice is fetched once
    // per XCD instead of once per block.  Placement affects speed only.
    int lid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    int vec, tile;
    if (a.period > 0) {
        const int nper = a.nvec / a.period;
        const int rep = lid % nper, r = lid / nper;
        tile = r & ((1 << logtiles) - 1);
        vec = a.vec0 + rep * a.period + (r >> logtiles);
    } else {
        vec = a.vec0 + (lid >> logtiles);
        tile = lid & ((1 << logtiles) - 1);
    }
    const int limb = limb_of(a, vec);
    if (limb < 0) return;
    const LimbConst c = limb_const(a, limb);
    if (lazy_prime<INVERSE>(c.q))
        rows_body<INVERSE, true, MODDOWN>(a, lds, vec, tile, limb, c, &md);
    else
        rows_body<INVERSE, false, MODDOWN>(a, lds, vec, tile, limb, c, &md);
}

template <int A>
void launch_cols(const NttArgs& a, bool inverse, int blocks, hipStream_t s) {
    if (inverse)
        hipLaunchKernelGGL((ntt_cols_kernel<A, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.lift_qlm)
        hipLaunchKernelGGL((ntt_cols_kernel<A, false, true>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((ntt_cols_kernel<A, false>), dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace

static void launch_ntt_impl(const DeviceTables& t, const LimbBatch& b, bool inverse, const NttModDown* md, hipStream_t s);

void launch_ntt(const DeviceTables& t, const LimbBatch& b, bool inverse, hipStream_t s) { launch_ntt_impl(t, b, inverse, nullptr, s); }
void launch_ntt_moddown(const DeviceTables& t, const LimbBatch& b, const NttModDown& md, hipStream_t s) {
    launch_ntt_impl(t, b, false, &md, s);
}

static void launch_ntt_impl(const DeviceTables& t, const LimbBatch& b, bool inverse, const NttModDown* md, hipStream_t s) {
    if (b.nvec <= 0) return;
    NttArgs a;
    a.data = b.data;
    a.src = b.src ? b.src : b.data;
    a.src_group = b.src ? b.src_group : 0;
    a.src_group_stride = b.src_group_stride;
    a.src_group2 = (b.src && b.src_group > 0) ? b.src_group2 : 0;
    a.src_group2_stride = b.src_group2_stride;
    a.lazy_out = (!inverse && b.lazy_out) ? 1 : 0;
    a.lift_qlm = (!inverse && b.src && b.lift_limb >= 0) ? b.lift_qlm : nullptr;
    a.lift_limb = b.lift_limb;
    a.tw = inverse ? t.tw_inv : t.tw_fwd;
    a.tw_rows = inverse ? t.tw_rows_inv : t.tw_rows_fwd;
    a.moduli = t.moduli;
    a.ninv = t.ninv;
    a.limb_tab = b.limb_tab;
    a.tab_len = b.tab_len;
    a.limb_first = b.limb_first;
    a.limb_count = b.limb_count > 0 ? b.limb_count : 1;
    a.log_n = t.log_n;
    a.nvec = b.nvec;
    const int per = b.limb_tab ? b.tab_len : a.limb_count;
    a.period = (per > 0 && per < b.nvec && b.nvec % per == 0) ? per : 0;
    const int A = t.log_n - 8;
    a.vec0 = 0;
    // FHELIN_NTT_CHUNK_MB=<m> (experiment, default off): a large transform as CHUNKS of about m MiB of vectors, both tile passes per chunk,
    // so that the intermediate a chunk's first pass writes is still in the 256 MiB Infinity Cache when its second pass reads it.
    // Measured in round 4 (profiles/r04_q_*): SLOWER - 274 -> 284 / 293 / 316 ms per pass at 96 / 64 / 32 MiB, micro-benchmark 1.97 -> 1.69 M
    // limb-NTT/s: the extra launch boundaries (every launch drains before the next starts) cost more than HBM round trips of the
    // intermediate do, i.e. the tile passes are not waiting on HBM.  Chunks are multiples of the twiddle period (vectors v and
    // v + period share a limb) so that the block order of the row pass keeps working inside a chunk.
    static const int chunk_mb = [] {
        const char* e = std::getenv("FHELIN_NTT_CHUNK_MB");
        return e ? std::atoi(e) : 0;
    }();
    int chunk = b.nvec;
    if (chunk_mb > 0 && !md) {
        const size_t vec_bytes = (size_t)8 << t.log_n;
        const int want = (int)std::max<size_t>(1, ((size_t)chunk_mb << 20) / vec_bytes);
        if (b.nvec > want + want / 2) {
            const int per = a.period > 0 ? a.period : 1;
            chunk = std::max(per, want / per * per);
        }
    }
    const int nvec_all = b.nvec, period_all = a.period;
    const u64* src_all = a.src;
    const int sg = a.src_group, sg2 = a.src_group2;
    for (int v0 = 0; v0 < nvec_all; v0 += chunk) {
        a.vec0 = v0;
        a.nvec = std::min(chunk, nvec_all - v0);
        a.period = (period_all > 0 && a.nvec % period_all == 0 && a.nvec > period_all) ? period_all : 0;
        a.src = src_all;
        a.src_group = sg;
        a.src_group2 = sg2;
        const int blocks = a.nvec << (t.log_n - 12);
        auto cols = [&]() {
            switch (A) {
                case 4: launch_cols<4>(a, inverse, blocks, s); break;
                case 5: launch_cols<5>(a, inverse, blocks, s); break;
                case 6: launch_cols<6>(a, inverse, blocks, s); break;
                case 7: launch_cols<7>(a, inverse, blocks, s); break;
                case 8: launch_cols<8>(a, inverse, blocks, s); break;
                case 9: launch_cols<9>(a, inverse, blocks, s); break;
                default: break;
            }
        };
        // the first pass may read out of place; the second always works in place on `data`
        if (!inverse) {
            cols();
            a.src = a.data;
            a.src_group = 0;
            a.src_group2 = 0;
            if (md)
                hipLaunchKernelGGL((ntt_rows_kernel<false, true>), dim3(blocks), dim3(256), 0, s, a, *md);
            else
                hipLaunchKernelGGL((ntt_rows_kernel<false, false>), dim3(blocks), dim3(256), 0, s, a, NttModDown());
        } else {
            hipLaunchKernelGGL((ntt_rows_kernel<true, false>), dim3(blocks), dim3(256), 0, s, a, NttModDown());
            a.src = a.data;
            a.src_group = 0;
            a.src_group2 = 0;
            cols();
        }
    }
}

}  // namespace fhelin
