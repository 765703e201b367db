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
// in VGPRs and runs 4 stages on them between LDS exchanges ("rounds").  Butterflies are Harvey lazy
// butterflies (values in [0,4q) forward / [0,2q) inverse) with Shoup twiddles, canonicalised only on
// the final store.  This is 64-bit modular-integer work: no MFMA.
//
// Memory behaviour: every global access of a wave instruction is a run of whole 128-byte lines (column pass:
// 4 row segments of 128 B; row pass: 512 B contiguous, reached through one extra LDS exchange).  The first pass can
// read out of place and strided (LimbBatch::src / src_group), which is how key switching and rescale avoid staging
// copies.  The row pass maps workgroups to (limb, tile, repetition) in XCD-contiguous order so that the 4 KiB
// twiddle slice of a (limb, tile) is shared by the vectors of a batch in one L2.
// Measured (profiles/): 1.6 M limb-NTT/s at N=2^16, VALU-bound (~87 % VALU busy, ~32 VALU
// instructions per butterfly of which 10 are half-rate 32x32 multiplies), 1.9x algorithmic HBM traffic.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace fhelin {

namespace {

struct NttArgs {
    u64* data;
    const u64* src;     // input of this pass (== data when in place)
    const u64* tw;      // [n_limbs][2N]
    const u64* moduli;  // [n_limbs]
    const u64* ninv;    // [n_limbs][4]
    const int* limb_tab;
    int tab_len;
    int limb_first;
    int limb_count;
    int log_n;
    int nvec;
    int period;  // > 0: vectors v and v + period use the same limb (twiddles); used to co-schedule them
    int src_group;            // > 0: strided first-pass input (see LimbBatch)
    size_t src_group_stride;
};

__device__ __forceinline__ size_t src_offset(const NttArgs& a, int vec, int log_n) {
    if (a.src_group > 0) return (size_t)(vec / a.src_group) * a.src_group_stride + ((size_t)(vec % a.src_group) << log_n);
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

// Forward (Cooley-Tukey) stages for register-index bits KB_HI-1 .. KB_LO (descending).
//   gbase : global index bit that register bit 0 of the window corresponds to
//   hi    : the thread's global index bits above the window (index >> (gbase+4))
template <int KB_LO, int KB_HI>
__device__ __forceinline__ void fwd_round(u64 (&x)[16], const u64x2* __restrict__ tw, u64 q, u64 q2, int log_n,
                                          int gbase, int hi) {
#pragma unroll
    for (int kb = KB_HI - 1; kb >= KB_LO; --kb) {
        const int g = gbase + kb;
        const int m = 1 << (log_n - g - 1);
        const int tbase = m + (hi << (3 - kb));
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) {
            if (k0 & (1 << kb)) continue;
            const int k1 = k0 | (1 << kb);
            const u64x2 w = tw[tbase + (k0 >> (kb + 1))];
            u64 X = x[k0];
            const u64 Y = x[k1];
            X = csub(X, q2);
            const u64 T = mul_shoup_lazy(Y, w.x, w.y, q);
            x[k0] = X + T;
            x[k1] = X - T + q2;
        }
    }
}

// Inverse (Gentleman-Sande) stages for register-index bits KB_LO .. KB_HI-1 (ascending).
// If LAST, the stage at bit KB_HI-1 is the final stage of the whole transform and carries N^{-1}.
template <int KB_LO, int KB_HI, bool LAST>
__device__ __forceinline__ void inv_round(u64 (&x)[16], const u64x2* __restrict__ tw, u64 q, u64 q2, int log_n,
                                          int gbase, int hi, const u64* __restrict__ ninv) {
#pragma unroll
    for (int kb = KB_LO; kb < KB_HI; ++kb) {
        const int g = gbase + kb;
        const int m = 1 << (log_n - g - 1);
        const int tbase = m + (hi << (3 - kb));
        if (LAST && kb == KB_HI - 1) {
            const u64 ni = ninv[0], nis = ninv[1], wn = ninv[2], wns = ninv[3];
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) {
                if (k0 & (1 << kb)) continue;
                const int k1 = k0 | (1 << kb);
                const u64 X = x[k0], Y = x[k1];
                x[k0] = csub(mul_shoup_lazy(X + Y, ni, nis, q), q);
                x[k1] = csub(mul_shoup_lazy(X - Y + q2, wn, wns, q), q);
            }
        } else {
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) {
                if (k0 & (1 << kb)) continue;
                const int k1 = k0 | (1 << kb);
                const u64x2 w = tw[tbase + (k0 >> (kb + 1))];
                const u64 X = x[k0], Y = x[k1];
                x[k0] = csub(X + Y, q2);
                x[k1] = mul_shoup_lazy(X - Y + q2, w.x, w.y, q);
            }
        }
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

// ------------------------------------------------------------------------------------------------
// pass A: column transforms over the high A bits of the index.  Tile = 2^A rows x CW columns.
// tile-local index e = (row << LOGCW) | col ; global index j = (row << 8) | (tile*CW + col).
// ------------------------------------------------------------------------------------------------
template <int A, bool INVERSE>
__global__ __launch_bounds__(256) void ntt_cols_kernel(NttArgs a) {
    constexpr int LOGCW = 12 - A;
    constexpr int CW = 1 << LOGCW;
    constexpr int LOGTILES = A - 4;
    constexpr int LOGN = A + 8;
    constexpr int NFULL = A / 4;
    constexpr int REM = A % 4;
    __shared__ u64 lds[LDS_WORDS];

    const int vec = blockIdx.x >> LOGTILES;
    const int tile = blockIdx.x & ((1 << LOGTILES) - 1);
    const int limb = limb_of(a, vec);
    if (limb < 0) return;
    const u64 q = a.moduli[limb];
    const u64 q2 = q << 1;
    const u64x2* tw = reinterpret_cast<const u64x2*>(a.tw) + ((size_t)limb << LOGN);
    u64* base = a.data + ((size_t)vec << LOGN) + tile * CW;
    const u64* sbase = a.src + src_offset(a, vec, LOGN) + tile * CW;
    const int tau = threadIdx.x;
    u64 x[16];

    // global offset (relative to base) of tile-local index e
    auto goff = [](int e) { return ((e >> LOGCW) << 8) | (e & (CW - 1)); };

    if (!INVERSE) {
        // windows (in tile-index bits), top down: LOGCW+A-4, LOGCW+A-8, ..., then the remainder at LOGCW
        constexpr int P0 = LOGCW + A - 4;
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[goff(tile_index(tau, k, P0))];
        fwd_round<0, 4>(x, tw, q, q2, LOGN, 8 + P0 - LOGCW, tau >> P0);
        int p_prev = P0;
        if constexpr (NFULL >= 2) {
            constexpr int P1 = LOGCW + A - 8;
            exchange(x, lds, tau, p_prev, P1, false);
            fwd_round<0, 4>(x, tw, q, q2, LOGN, 8 + P1 - LOGCW, tau >> P1);
            p_prev = P1;
        }
        if constexpr (REM > 0) {
            constexpr int PR = LOGCW;
            exchange(x, lds, tau, p_prev, PR, NFULL >= 2);
            fwd_round<0, REM>(x, tw, q, q2, LOGN, 8, tau >> PR);
            p_prev = PR;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) base[goff(tile_index(tau, k, p_prev))] = x[k];
    } else {
        // inverse: bottom up.  full rounds at LOGCW, LOGCW+4, ...; remainder = top REM bits of the window at 8
        const u64* ninv = a.ninv + 4 * limb;
        constexpr int P0 = LOGCW;
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[goff(tile_index(tau, k, P0))];
        inv_round<0, 4, (NFULL == 1 && REM == 0)>(x, tw, q, q2, LOGN, 8, tau >> P0, ninv);
        int p_prev = P0;
        if constexpr (NFULL >= 2) {
            constexpr int P1 = LOGCW + 4;
            exchange(x, lds, tau, p_prev, P1, false);
            inv_round<0, 4, (REM == 0)>(x, tw, q, q2, LOGN, 8 + 4, tau >> P1, ninv);
            p_prev = P1;
        }
        if constexpr (REM > 0) {
            constexpr int PR = 8;  // window covers tile bits 8..11; its top REM bits are still to do
            exchange(x, lds, tau, p_prev, PR, NFULL >= 2);
            inv_round<4 - REM, 4, true>(x, tw, q, q2, LOGN, 8 + PR - LOGCW, tau >> PR, ninv);
            p_prev = PR;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) base[goff(tile_index(tau, k, p_prev))] = x[k];
    }
}

// ------------------------------------------------------------------------------------------------
// pass B: row transforms over the low 8 bits of the index.  Tile = 16 consecutive rows of 256
// (4096 contiguous residues); tile-local index e == global index - tile*4096.
// ------------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(256) void ntt_rows_kernel(NttArgs a) {
    __shared__ u64 lds[LDS_WORDS];
    const int log_n = a.log_n;
    const int logtiles = log_n - 12;
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
        vec = rep * a.period + (r >> logtiles);
    } else {
        vec = lid >> logtiles;
        tile = lid & ((1 << logtiles) - 1);
    }
    const int limb = limb_of(a, vec);
    if (limb < 0) return;
    const u64 q = a.moduli[limb];
    const u64 q2 = q << 1;
    const u64x2* tw = reinterpret_cast<const u64x2*>(a.tw) + ((size_t)limb << log_n);
    u64* base = a.data + ((size_t)vec << log_n) + ((size_t)tile << 12);
    const u64* sbase = a.src + src_offset(a, vec, log_n) + ((size_t)tile << 12);
    const int tau = threadIdx.x;
    u64 x[16];

    if (!INVERSE) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[tile_index(tau, k, 4)];
        fwd_round<0, 4>(x, tw, q, q2, log_n, 4, (tile << 4) | (tau >> 4));
        exchange(x, lds, tau, 4, 0, false);
        fwd_round<0, 4>(x, tw, q, q2, log_n, 0, (tile << 8) | tau);
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = csub(csub(x[k], q2), q);
        // window at bit 0 leaves 16 consecutive residues per thread (128-byte lane stride); one more LDS exchange
        // to the bit-8 window makes every store instruction a contiguous 512-byte wave access
        exchange(x, lds, tau, 0, 8, true);
#pragma unroll
        for (int k = 0; k < 16; ++k) base[tile_index(tau, k, 8)] = x[k];
    } else {
        // coalesced load through the bit-8 window, then an LDS exchange to the bit-0 window of the first GS round
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = sbase[tile_index(tau, k, 8)];
        exchange(x, lds, tau, 8, 0, false);
        inv_round<0, 4, false>(x, tw, q, q2, log_n, 0, (tile << 8) | tau, nullptr);
        exchange(x, lds, tau, 0, 4, true);
        inv_round<0, 4, false>(x, tw, q, q2, log_n, 4, (tile << 4) | (tau >> 4), nullptr);
#pragma unroll
        for (int k = 0; k < 16; ++k) base[tile_index(tau, k, 4)] = x[k];
    }
}

template <int A>
void launch_cols(const NttArgs& a, bool inverse, int blocks, hipStream_t s) {
    if (inverse)
        hipLaunchKernelGGL((ntt_cols_kernel<A, true>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((ntt_cols_kernel<A, false>), dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace

void launch_ntt(const DeviceTables& t, const LimbBatch& b, bool inverse, hipStream_t s) {
    if (b.nvec <= 0) return;
    NttArgs a;
    a.data = b.data;
    a.src = b.src ? b.src : b.data;
    a.src_group = b.src ? b.src_group : 0;
    a.src_group_stride = b.src_group_stride;
    a.tw = inverse ? t.tw_inv : t.tw_fwd;
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
    const int blocks = b.nvec << (t.log_n - 12);
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
        hipLaunchKernelGGL((ntt_rows_kernel<false>), dim3(blocks), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((ntt_rows_kernel<true>), dim3(blocks), dim3(256), 0, s, a);
        a.src = a.data;
        a.src_group = 0;
        cols();
    }
}

}  // namespace fhelin
