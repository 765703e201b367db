// K2/K3/K4/K5/K9 — element-wise residue kernels for gfx950: dyadic multiply / tensor, add / sub / negate /
// per-limb scalar multiply, NTT-domain automorphism gather, rescale lift/finish, int128 -> RNS reduction.
//
// Reference side: the DCRTPoly operator*=, +=, -=, AutomorphismTransform and DropLastElementAndScale loops
// that run inside OpenFHE under context->EvalMult / EvalAdd / EvalRotate (reference src/FHEController.cpp
// :410,:414,:423-435).  SURVEY.md §8(a) rows K2-K5, K9.
//
// All of these are HBM-bound streaming kernels: limb-major [vec][N] u64 arrays, 16-byte accesses per lane,
// grid = (N / 512, vectors); per-limb constants (q, Barrett ratio, Shoup scalars) are wave-uniform and come
// in through scalar loads.  No MFMA (64-bit modular integers).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "kernels_elem.h"

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

// out[v] = a[v] (op) b[v % b_mod]
template <int OP>  // 0 mul, 1 add, 2 sub
__global__ __launch_bounds__(256) void ew_binary_kernel(DeviceTables t, u64* out, const u64* a, const u64* b, int b_mod,
                                                        int limb_first, int limb_count) {
    const int v = blockIdx.y;
    const int limb = limb_first + v % limb_count;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;  // index in u64x2 units
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(a)[(size_t)v * row + n2];
    const u64x2 y = reinterpret_cast<const u64x2*>(b)[(size_t)(v % b_mod) * row + n2];
    u64x2 r;
    if (OP == 0) {
        const Barrett br = load_barrett(t, limb);
        r.x = mul_mod(x.x, y.x, br);
        r.y = mul_mod(x.y, y.y, br);
    } else if (OP == 1) {
        const u64 q = t.moduli[limb];
        r.x = add_mod(x.x, y.x, q);
        r.y = add_mod(x.y, y.y, q);
    } else {
        const u64 q = t.moduli[limb];
        r.x = sub_mod(x.x, y.x, q);
        r.y = sub_mod(x.y, y.y, q);
    }
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

template <int OP>
__global__ __launch_bounds__(256) void ew_items_kernel(DeviceTables t, EwItems it, int limb_count) {
    const int item = blockIdx.y / it.vecs, v = blockIdx.y % it.vecs;
    const int limb = v % limb_count;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(it.a[item])[(size_t)v * row + n2];
    u64x2 r = x;
    if (OP != 4 && (OP != 3 || v < it.b_vecs)) {
        const u64x2 y = reinterpret_cast<const u64x2*>(it.b[item])[(size_t)(v % it.b_vecs) * row + n2];
        if (OP == 0) {
            const Barrett br = load_barrett(t, limb);
            r.x = mul_mod(x.x, y.x, br);
            r.y = mul_mod(x.y, y.y, br);
        } else if (OP == 2) {
            const u64 q = t.moduli[limb];
            r.x = sub_mod(x.x, y.x, q);
            r.y = sub_mod(x.y, y.y, q);
        } else {
            const u64 q = t.moduli[limb];
            r.x = add_mod(x.x, y.x, q);
            r.y = add_mod(x.y, y.y, q);
        }
    }
    reinterpret_cast<u64x2*>(it.out[item])[(size_t)v * row + n2] = r;
}

// out[v] = sum_i a_i[v] * b_i[v % b_vecs]; grid (N/512, vecs)
__global__ __launch_bounds__(256) void ew_dot_kernel(DeviceTables t, u64* out, EwItems it, int limb_count) {
    const int v = blockIdx.y;
    const Barrett br = load_barrett(t, v % limb_count);
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    Acc128 ax = {0, 0}, ay = {0, 0};
    for (int i = 0; i < it.n; ++i) {
        const u64x2 x = reinterpret_cast<const u64x2*>(it.a[i])[(size_t)v * row + n2];
        const u64x2 y = reinterpret_cast<const u64x2*>(it.b[i])[(size_t)(v % it.b_vecs) * row + n2];
        acc_mac(ax, x.x, y.x);
        acc_mac(ay, x.y, y.y);
    }
    u64x2 r;
    r.x = barrett_reduce128(ax.lo, ax.hi, br);
    r.y = barrett_reduce128(ay.lo, ay.hi, br);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// out_g[c][tt] = sum_b a_b[c][tt] * p_{g,b}[tt] for every g; grid (N/512, 2 ell): one component c of one limb tt per block, the
// NA ciphertext pairs of a thread held in registers across the loop over g.  The two blocks of a (limb, tile) that differ
// only in c read the same plaintext tile: in XCD-aware order (logical id = (tt, tile, c), c fastest, contiguous id ranges per
// XCD) they run back to back on one XCD and the second finds the tile in L2.
template <int NA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NA <= 8 ? 4 : 3))) void ew_dot_groups_kernel(DeviceTables t, EwDotGroups d) {
    const unsigned nx = gridDim.x, nblk = gridDim.x * gridDim.y;
    unsigned id = blockIdx.y * nx + blockIdx.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
    const int comp = (int)(id & 1);
    const unsigned rest = id >> 1;
    const int xb = (int)(rest % (unsigned)d.nbatch);                 // batch element: fastest after the component
    const int bx = (int)((rest / (unsigned)d.nbatch) % nx), tt = (int)((rest / (unsigned)d.nbatch) / nx);
    const Barrett br = load_barrett(t, tt);
    const size_t n2 = (size_t)bx * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t op = (size_t)tt * row + n2, oc = (size_t)(comp * d.ell + tt) * row + n2;
    // Every pointer of the struct is readable (launch_ew_dot_groups pads absent terms and the columns up to NA; pmask says which count): the NA
    // plaintext loads of a group go out back to back, and for NA <= 8 the next group's are in flight while this group's products are formed.
    // With a null test per term the compiler put a full wait behind every plaintext load - NA serial round trips per group.
    u64x2 a[NA];
#pragma unroll
    for (int b = 0; b < NA; ++b) a[b] = reinterpret_cast<const u64x2*>(d.a[b] + (size_t)xb * d.a_stride[b])[oc];
    if constexpr (NA <= 8) {
        u64x2 w[NA], wn[NA];
#pragma unroll
        for (int b = 0; b < NA; ++b) w[b] = reinterpret_cast<const u64x2*>(d.p[0][b])[op];
        for (int g = 0; g < d.ng; ++g) {
            const int gn = g + 1 < d.ng ? g + 1 : g;
#pragma unroll
            for (int b = 0; b < NA; ++b) wn[b] = reinterpret_cast<const u64x2*>(d.p[gn][b])[op];
            Acc128 x = {0, 0}, y = {0, 0};
            const u32 m = d.pmask[g];
#pragma unroll
            for (int b = 0; b < NA; ++b)
                if (m >> b & 1u) {
                    acc_mac(x, a[b].x, w[b].x);
                    acc_mac(y, a[b].y, w[b].y);
                }
            u64x2 r;
            r.x = barrett_reduce128(x.lo, x.hi, br);
            r.y = barrett_reduce128(y.lo, y.hi, br);
            reinterpret_cast<u64x2*>(d.out[g] + (size_t)xb * d.out_stride[g])[oc] = r;
#pragma unroll
            for (int b = 0; b < NA; ++b) w[b] = wn[b];
        }
    } else {   // 16 columns: the plaintexts of a group in two batches of eight (the 16 ciphertext pairs already hold 64 registers)
        for (int g = 0; g < d.ng; ++g) {
            Acc128 x = {0, 0}, y = {0, 0};
            const u32 m = d.pmask[g];
#pragma unroll
            for (int b0 = 0; b0 < NA; b0 += 8) {
                u64x2 w[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) w[b] = reinterpret_cast<const u64x2*>(d.p[g][b0 + b])[op];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (m >> (b0 + b) & 1u) {
                        acc_mac(x, a[b0 + b].x, w[b].x);
                        acc_mac(y, a[b0 + b].y, w[b].y);
                    }
            }
            u64x2 r;
            r.x = barrett_reduce128(x.lo, x.hi, br);
            r.y = barrett_reduce128(y.lo, y.hi, br);
            reinterpret_cast<u64x2*>(d.out[g] + (size_t)xb * d.out_stride[g])[oc] = r;
        }
    }
}

// grid (N/256, ell, 2 components): one coefficient per thread.  The plaintext values sit in LDS (a dynamically indexed register file:
// every thread reads its own column), the two ciphertext windows in registers, all pre-split in 30-bit halves.  The outputs are
// walked in four segments of eight (template parameter O0): inside a segment the operand of column j is cur_j for j < O0 and prev_j
// for j >= O0 + 8 whatever the output, and the LDS row is a fixed distance from the output's own - only the eight columns of the
// segment itself need the per-output select and the wrapped index.  (A fully unrolled register-only form - 1024 static
// multiply-accumulates - measured slower: 48 KB of straight-line code per kernel.)
// SH = bits of the low half of the pre-split operands: 30 for the 60-bit limbs (eight products per column before a flush, a fold
// every 16), 27 for limbs below 2^53 (every product below 2^54: all 32 of an output in one accumulator, one flush, one reduction)
template <int SH>
__device__ __forceinline__ u64 pack_sh(u64 x) { return (x & ((1ull << SH) - 1)) | ((x >> SH) << 32); }
template <int SH>
__device__ __forceinline__ void acc_flush_sh(const Acc30& a, u64& lo, u64& hi) {
    u64 t = lo + a.s0;
    hi += (t < lo);
    lo = t;
    t = lo + (a.s1 << SH);
    hi += (a.s1 >> (64 - SH)) + (t < lo);
    lo = t;
    t = lo + (a.s2 << (2 * SH));
    hi += (a.s2 >> (64 - 2 * SH)) + (t < lo);
    lo = t;
}

// G outputs of a segment per trip (FHELIN_WINDOW_GROUP, default 2; 4 spills): the outputs o .. o + G - 1 meet the mask m_(o - j) on the columns j, j + 1, ..,
// j + G - 1 - ONE LDS read feeds G multiply-accumulates.  With one output per trip (rounds 3) every product had its own 8-byte LDS read: 2 MB per
// workgroup at 128 bytes per clock is as long as the products themselves take on the four SIMDs, and the two do not overlap well at two waves per SIMD.
#ifndef WSPLIT
#define WSPLIT 1
#endif
// compile-time loop: f(integral_constant<int, I>) for I in [I0, I1) - every index inside is a constant whatever the unroller's budget says
// (a loop the unroller gives up on turns the register arrays below into scratch memory: measured 1.5 ms per launch instead of 0.9)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// The 32 cyclic sums out_k = sum_i a_i * m_((i + k) mod 32): grid (N/256, ell, 2 components), one coefficient per thread, the plaintext
// values in LDS, the ciphertext values in registers, both pre-split (SH as above).  Two outputs per trip: output k meets the mask of row i + k on
// column i, output k + 1 on column i - 1 - ONE LDS read feeds both products; the prologue's 64 loads go out back to back (every a[] entry is a
// readable pointer: launch_ew_cyclic_dot points the ones beyond n at the first, their values are dropped by a wave-uniform select).
template <int SH>
__device__ __forceinline__ void cyclic_body(const DeviceTables& t, const EwCyclic& d, const Barrett& br, u64 (*ml)[256]) {
    constexpr int P = EwCyclic::PERIOD, G = 2;
    const int tt = blockIdx.y;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t oc = (size_t)(blockIdx.z * d.ell + tt) * N + n;
    // byte offsets below 4 GB (2 ell N words): loads take the scalar base + 32-bit lane offset form - no 64-bit address pair per load
    const u32 bm = (u32)(((size_t)tt * N + n) * 8), bc = (u32)(oc * 8);
    auto ld = [](const u64* p, u32 byte_off) -> u64 { return *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(p) + byte_off); };
    {
        u64 mv[P];
#pragma unroll
        for (int j = 0; j < P; ++j) mv[j] = ld(d.m[j], bm);
#pragma unroll
        for (int j = 0; j < P; ++j) ml[j][threadIdx.x] = pack_sh<SH>(mv[j]);
    }
    __builtin_amdgcn_sched_barrier(0);   // the masks' registers are free before the ciphertext values arrive
    u64 a[P];
#pragma unroll
    for (int i = 0; i < P; ++i) a[i] = ld(d.a[i], bc);
#pragma unroll
    for (int i = 0; i < P; ++i) a[i] = i < d.n ? pack_sh<SH>(a[i]) : 0;
    // every thread reads only its own column of ml: no barrier needed (a thread sees its own LDS writes in program order)
#pragma unroll 1
    for (int k = 0; k < P; k += G) {
        u64 lo[G], hi[G];
        Acc30 x[G];
        static_for<0, G>([&](auto I) __attribute__((always_inline)) {
            lo[decltype(I)::value] = hi[decltype(I)::value] = 0;
            x[decltype(I)::value] = Acc30{0, 0, 0};
        });
        static_for<0, P / 8>([&](auto I0) __attribute__((always_inline)) {
            constexpr int i0 = 8 * decltype(I0)::value;
            static_for<i0, i0 + 8>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const u64 w = ml[(i + k) & (P - 1)][threadIdx.x];
                const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
                static_for<0, G>([&](auto Gi) __attribute__((always_inline)) {
                    constexpr int g = decltype(Gi)::value;
                    constexpr int ii = (i - g) & (P - 1);          // (ii + k + g) mod 32 = (i + k) mod 32
                    mac30(x[g], (u32)a[ii], (u32)(a[ii] >> 32), w0, w1);
                });
            });
            __builtin_amdgcn_sched_barrier(0);   // LDS reads of later chunks stay where they are (hoisted, they cost the registers two outputs need)
            if constexpr (SH == 30) {
                static_for<0, G>([&](auto Gi) __attribute__((always_inline)) {
                    constexpr int g = decltype(Gi)::value;
                    acc_flush_sh<30>(x[g], lo[g], hi[g]);
                    x[g] = Acc30{0, 0, 0};
                    if constexpr (i0 == 8) {   // 16 products so far: fold, so that the final 128-bit value stays below q * 2^64 for 60-bit limbs too
                        lo[g] = barrett_reduce128(lo[g], hi[g], br);
                        hi[g] = 0;
                    }
                });
            }
        });
        static_for<0, G>([&](auto Gi) __attribute__((always_inline)) {
            constexpr int g = decltype(Gi)::value;
            if constexpr (SH != 30) acc_flush_sh<SH>(x[g], lo[g], hi[g]);   // all products below 2^54: one accumulator, one flush
            d.out[k + g][oc] = barrett_reduce128(lo[g], hi[g], br);
        });
    }
}

__global__ __launch_bounds__(256, 2) void ew_cyclic_dot_kernel(DeviceTables t, EwCyclic d) {
    __shared__ u64 ml[EwCyclic::PERIOD][256];        // the plaintext values of this workgroup's 256 coefficients, pre-split
    const Barrett br = load_barrett(t, blockIdx.y);
    if ((br.q >> 53) == 0)                           // wave-uniform: the limb of the block
        cyclic_body<27>(t, d, br, ml);
    else
        cyclic_body<30>(t, d, br, ml);
}

// `pre`: the sums of earlier tap chunks for the NEXT trip's outputs, fetched one trip ahead.  On gfx9 one counter orders loads AND stores: a load
// issued after a trip's stores is only known complete once those stores are acknowledged, so fetching the running sums at the top of their own
// trip made every trip wait for a store round trip and a load round trip (SQ_WAIT_ANY: 65 % of a wave's life at two waves per SIMD).
template <int O0, int SH, int G>
__device__ __forceinline__ void window_segment(const EwWindow& d, const Barrett& br, const u64 (&cur)[EwWindow::W], const u64 (&prv)[EwWindow::W],
                                               const u64 (*ml)[256], size_t oc, u64 (&pre)[G]) {
    constexpr int P = EwWindow::W;
#pragma unroll 1
    for (int oo = 0; oo < 8; oo += G) {
        u64 lo[G], hi[G];
        Acc30 x[G][WSPLIT];   // WSPLIT independent accumulation chains per output (consecutive columns alternate)
        static_for<0, G>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            lo[i] = pre[i];
            hi[i] = 0;
            static_for<0, WSPLIT>([&](auto U) __attribute__((always_inline)) { x[i][decltype(U)::value] = Acc30{0, 0, 0}; });
        });
        if (d.accumulate && O0 + oo + G < P) {
            static_for<0, G>([&](auto I) __attribute__((always_inline)) { pre[decltype(I)::value] = d.out[O0 + oo + G + decltype(I)::value][oc]; });
        }
        static_for<0, P / 8>([&](auto J0) __attribute__((always_inline)) {
            constexpr int j0 = 8 * decltype(J0)::value;
            static_for<j0, j0 + 8>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                // the mask of the diagonal o - j' = O0 + oo - j (mod 32), the same for the G outputs: row oo + c, c = O0 - j
                constexpr int c = O0 - j;
                u64 w;
                if constexpr (c >= 0)                  // oo + c <= (8 - G) + 24: no wrap
                    w = ml[oo + c][threadIdx.x];
                else if constexpr (c + (8 - G) < 0)    // below zero for every trip: wrapped
                    w = ml[oo + c + P][threadIdx.x];
                else
                    w = ml[(oo + c) & (P - 1)][threadIdx.x];
                const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
                static_for<0, G>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    constexpr int jj = (j + i) & (P - 1);      // the column output o = O0 + oo + i meets this mask on
                    constexpr int e = jj - O0 - i;             // cur_jj iff jj <= o iff e <= oo
                    u64 a;
                    if constexpr (jj < O0 || e <= 0)
                        a = cur[jj];
                    else if constexpr (jj >= O0 + 8 || e > 8 - G)
                        a = prv[jj];
                    else
                        a = oo >= e ? cur[jj] : prv[jj];       // wave-uniform
                    mac30(x[i][j % WSPLIT], (u32)a, (u32)(a >> 32), w0, w1);
                });
            });
            if constexpr (SH == 30) {
                static_for<0, G>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    static_for<0, WSPLIT>([&](auto U) __attribute__((always_inline)) {
                        acc_flush_sh<30>(x[i][decltype(U)::value], lo[i], hi[i]);
                        x[i][decltype(U)::value] = Acc30{0, 0, 0};
                    });
                    if constexpr (j0 == 8 || j0 == 24) {   // <= 16 products (+ one carried residue) per fold: below q * 2^64 for the 60-bit limbs too
                        lo[i] = barrett_reduce128(lo[i], hi[i], br);
                        hi[i] = 0;
                    }
                });
            }
        });
        static_for<0, G>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (SH != 30) {
                static_for<0, WSPLIT>([&](auto U) __attribute__((always_inline)) { acc_flush_sh<SH>(x[i][decltype(U)::value], lo[i], hi[i]); });
                lo[i] = barrett_reduce128(lo[i], hi[i], br);
            }
            d.out[O0 + oo + i][oc] = lo[i];
        });
    }
}

template <int SH, int G>
__device__ __forceinline__ void window_body(const DeviceTables& t, const EwWindow& d, const Barrett& br, u64 (*ml)[256]) {
    constexpr int P = EwWindow::W;
    const int tt = blockIdx.y;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t om = (size_t)tt * N + n, oc = (size_t)(blockIdx.z * d.ell + tt) * N + n;
    // The prologue's 96 loads go out back to back (every entry of the struct is a valid pointer - the host points absent ones at the first
    // output - and the masks say which values count): with a null test per entry the compiler put a branch and a full wait between two loads,
    // 96 serial round trips per thread - two thirds of a wave's life (SQ_WAIT_ANY, profiles/r04_bc_*).
    {
        u64 mv[P];
#pragma unroll
        for (int k = 0; k < P; ++k) mv[k] = d.m[k][om];
#pragma unroll
        for (int k = 0; k < P; ++k) ml[k][threadIdx.x] = pack_sh<SH>(mv[k]);
    }
    // Segment O0 reads cur_j for j < O0 + 8 and prev_j for j >= O0: 40 of the 64 window values are live at a time when the current window
    // arrives segment by segment (one segment ahead of its use) and the previous one drains - 80 VGPRs instead of 128, which is what lets G
    // outputs travel together without spills.
    u64 cur[P], prv[P];
    auto load_cur = [&](int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 8; ++j) cur[j] = d.cur[j][oc];
#pragma unroll
        for (int j = j0; j < j0 + 8; ++j) cur[j] = (d.cur_mask >> j & 1u) ? pack_sh<SH>(cur[j]) : 0;
    };
#pragma unroll
    for (int j = 0; j < P; ++j) prv[j] = d.prev[j][oc];
#pragma unroll
    for (int j = 0; j < P; ++j) prv[j] = (d.prev_mask >> j & 1u) ? pack_sh<SH>(prv[j]) : 0;
    load_cur(0);
    load_cur(8);
    u64 pre[G];
#pragma unroll
    for (int i = 0; i < G; ++i) pre[i] = d.accumulate ? d.out[i][oc] : 0;
    // every thread reads only its own column of ml: no barrier needed (a thread sees its own LDS writes in program order)
    window_segment<0, SH, G>(d, br, cur, prv, ml, oc, pre);
    load_cur(16);
    window_segment<8, SH, G>(d, br, cur, prv, ml, oc, pre);
    load_cur(24);
    window_segment<16, SH, G>(d, br, cur, prv, ml, oc, pre);
    window_segment<24, SH, G>(d, br, cur, prv, ml, oc, pre);
}

template <int G>
__global__ __launch_bounds__(256, 2) void ew_window_dot_kernel(DeviceTables t, EwWindow d) {
    __shared__ u64 ml[EwWindow::W][256];             // the plaintext values of this workgroup's 256 coefficients, pre-split
    const Barrett br = load_barrett(t, blockIdx.y);
    if ((br.q >> 53) == 0)                           // wave-uniform: the limb of the block
        window_body<27, G>(t, d, br, ml);
    else
        window_body<30, G>(t, d, br, ml);
}

// out[v] = acc[v] + a[v] * b[v % b_mod]
__global__ __launch_bounds__(256) void ew_muladd_kernel(DeviceTables t, u64* out, const u64* acc, const u64* a, const u64* b,
                                                        int b_mod, int limb_first, int limb_count) {
    const int v = blockIdx.y;
    const int limb = limb_first + v % limb_count;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const Barrett br = load_barrett(t, limb);
    const u64x2 x = reinterpret_cast<const u64x2*>(a)[(size_t)v * row + n2];
    const u64x2 y = reinterpret_cast<const u64x2*>(b)[(size_t)(v % b_mod) * row + n2];
    const u64x2 c = reinterpret_cast<const u64x2*>(acc)[(size_t)v * row + n2];
    u64x2 r;
    r.x = add_mod(c.x, mul_mod(x.x, y.x, br), br.q);
    r.y = add_mod(c.y, mul_mod(x.y, y.y, br), br.q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

__global__ __launch_bounds__(256) void ew_neg_kernel(DeviceTables t, u64* out, const u64* a, int limb_first, int limb_count) {
    const int v = blockIdx.y;
    const u64 q = t.moduli[limb_first + v % limb_count];
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(a)[(size_t)v * row + n2];
    u64x2 r;
    r.x = neg_mod(x.x, q);
    r.y = neg_mod(x.y, q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// out[v] = a[v] * s[limb]   (s given with Shoup companion: sc[2*i], sc[2*i+1] for i = v % limb_count)
// in_limbs > 0: the input polynomials have in_limbs limbs each and only their first limb_count are read (a product that is
// level-reduced at once: the limbs nobody will read are neither multiplied nor copied)
__global__ __launch_bounds__(256) void ew_scalar_kernel(DeviceTables t, u64* out, const u64* a, ScalarSet sc, int limb_first,
                                                        int limb_count, int in_limbs) {
    const int v = blockIdx.y;
    const int li = v % limb_count;
    const u64 q = t.moduli[limb_first + li];
    const u64 w = sc.v[2 * li], ws = sc.v[2 * li + 1];
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t vin = in_limbs > 0 ? (size_t)(v / limb_count) * in_limbs + li : (size_t)v;
    const u64x2 x = reinterpret_cast<const u64x2*>(a)[vin * row + n2];
    u64x2 r;
    r.x = mul_shoup(x.x, w, ws, q);
    r.y = mul_shoup(x.y, w, ws, q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// out[v] = a[v] + s[limb]   (adds a constant to every NTT slot == adds the constant polynomial)
// add_vecs: only the first add_vecs vectors (component 0 of a ciphertext) get the constant, the others are copied through
__global__ __launch_bounds__(256) void ew_addscalar_kernel(DeviceTables t, u64* out, const u64* a, ScalarSet sc, int limb_first,
                                                           int limb_count, int add_vecs) {
    const int v = blockIdx.y;
    const int li = v % limb_count;
    const u64 q = t.moduli[limb_first + li];
    const u64 w = v < add_vecs ? sc.v[2 * li] : 0;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(a)[(size_t)v * row + n2];
    u64x2 r;
    r.x = add_mod(x.x, w, q);
    r.y = add_mod(x.y, w, q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// out[v] = sum_k a_k[v] * s_k[limb] + s_n[limb]; grid (N/512, vecs).  The products are summed in 128 bits and reduced once:
// the canonical residue of the sum, whatever the order (32 q^2 < q 2^64 for the Q limbs, q < 2^59).
__global__ __launch_bounds__(256) void ew_lincomb_kernel(DeviceTables t, u64* out, LinComb lc, const u64* __restrict__ scal, int ell,
                                                         int in_limbs) {
    const int v = blockIdx.y;
    const int l = v % ell;
    const Barrett br = load_barrett(t, l);
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t vin = in_limbs > 0 ? (size_t)(v / ell) * in_limbs + l : (size_t)v;   // inputs with more limbs: the first ell are read
    Acc128 ax = {0, 0}, ay = {0, 0};
    for (int k = 0; k < lc.n; ++k) {
        const u64 sk = scal[(size_t)k * ell + l];
        const u64x2 x = reinterpret_cast<const u64x2*>(lc.a[k])[vin * row + n2];
        acc_mac(ax, x.x, sk);
        acc_mac(ay, x.y, sk);
    }
    const u64 c0 = v < ell ? scal[(size_t)lc.n * ell + l] : 0;   // the constant is added to component 0 only
    u64x2 r;
    r.x = add_mod(barrett_reduce128(ax.lo, ax.hi, br), c0, br.q);
    r.y = add_mod(barrett_reduce128(ay.lo, ay.hi, br), c0, br.q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// tensor product of two 2-component ciphertexts a, b [2][ell][N] -> d [3][ell][N]
__global__ __launch_bounds__(256) void tensor_kernel(DeviceTables t, u64* d, const u64* a, const u64* b, int ell) {
    const int l = blockIdx.y;
    const Barrett br = load_barrett(t, l);
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t poly = (size_t)ell * row;
    const u64x2* A = reinterpret_cast<const u64x2*>(a);
    const u64x2* B = reinterpret_cast<const u64x2*>(b);
    const u64x2 a0 = A[l * row + n2], a1 = A[poly + l * row + n2];
    const u64x2 b0 = B[l * row + n2], b1 = B[poly + l * row + n2];
    u64x2 d0, d1, d2;
    d0.x = mul_mod(a0.x, b0.x, br);
    d0.y = mul_mod(a0.y, b0.y, br);
    d2.x = mul_mod(a1.x, b1.x, br);
    d2.y = mul_mod(a1.y, b1.y, br);
    d1.x = add_mod(mul_mod(a0.x, b1.x, br), mul_mod(a1.x, b0.x, br), br.q);
    d1.y = add_mod(mul_mod(a0.y, b1.y, br), mul_mod(a1.y, b0.y, br), br.q);
    u64x2* D = reinterpret_cast<u64x2*>(d);
    D[l * row + n2] = d0;
    D[poly + l * row + n2] = d1;
    D[2 * poly + l * row + n2] = d2;
}

// the same for up to EwItems::MAX_ITEMS independent pairs in one launch (the products of a batched multiplication): grid.z = pair
__global__ __launch_bounds__(256) void tensor_items_kernel(DeviceTables t, EwItems it, int ell) {
    const int l = blockIdx.y, k = blockIdx.z;
    const Barrett br = load_barrett(t, l);
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t poly = (size_t)ell * row;
    const u64x2* A = reinterpret_cast<const u64x2*>(it.a[k]);
    const u64x2* B = reinterpret_cast<const u64x2*>(it.b[k]);
    const u64x2 a0 = A[l * row + n2], a1 = A[poly + l * row + n2];
    const u64x2 b0 = B[l * row + n2], b1 = B[poly + l * row + n2];
    u64x2 d0, d1, d2;
    d0.x = mul_mod(a0.x, b0.x, br);
    d0.y = mul_mod(a0.y, b0.y, br);
    d2.x = mul_mod(a1.x, b1.x, br);
    d2.y = mul_mod(a1.y, b1.y, br);
    d1.x = add_mod(mul_mod(a0.x, b1.x, br), mul_mod(a1.x, b0.x, br), br.q);
    d1.y = add_mod(mul_mod(a0.y, b1.y, br), mul_mod(a1.y, b0.y, br), br.q);
    u64x2* D = reinterpret_cast<u64x2*>(it.out[k]);
    D[l * row + n2] = d0;
    D[poly + l * row + n2] = d1;
    D[2 * poly + l * row + n2] = d2;
}

// K4: out[v][j] = in[v][map[j]]
__global__ __launch_bounds__(256) void automorph_kernel(int log_n, u64* out, const u64* in, const u32* map) {
    const int v = blockIdx.y;
    const size_t n = (size_t)1 << log_n;
    const size_t j = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const u32 m0 = map[j], m1 = map[j + 1];
    u64x2 r;
    r.x = in[v * n + m0];
    r.y = in[v * n + m1];
    reinterpret_cast<u64x2*>(out)[(v * n + j) >> 1] = r;
}

// the same gather with every word stored pre-split in 30-bit halves (pack30): the merged inner product multiplies key words
// in halves anyway, so a key copy made for it (EvalKey::d_perm) carries them ready-made
// ... and times 2^64 (mod the vector's limb, v mod n_limbs): the merged inner product finishes its sums with redc128
__global__ __launch_bounds__(256) void automorph_pack30_kernel(DeviceTables t, u64* out, const u64* in, const u32* map) {
    const int v = blockIdx.y;
    const int limb = v % t.n_limbs;
    const u64 q = t.moduli[limb], R = t.mont[2 * limb], Rs = t.mont[2 * limb + 1];
    const size_t n = (size_t)1 << t.log_n;
    const size_t j = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const u32 m0 = map[j], m1 = map[j + 1];
    u64x2 r;
    r.x = pack30(mul_shoup(in[v * n + m0], R, Rs, q));
    r.y = pack30(mul_shoup(in[v * n + m1], R, Rs, q));
    reinterpret_cast<u64x2*>(out)[(v * n + j) >> 1] = r;
}

// K5 step 2: centred lift of the dropped limb (coefficient form, modulus q_l) into every remaining limb.
// last [npoly][N] -> lifted [npoly][ell-1][N]
__global__ __launch_bounds__(256) void rescale_lift_kernel(DeviceTables t, u64* lifted, const u64* last, int ell1 /* = ell-1 */,
                                                           const u64* qlmod_row) {
    const int v = blockIdx.y;  // p * ell1 + tq
    const int p = v / ell1, tq = v % ell1;
    const Barrett br = load_barrett(t, tq);
    const u64 ql = t.moduli[ell1];
    const u64 half = ql >> 1;
    const u64 qlm = qlmod_row[tq];
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(last)[(size_t)p * row + n2];
    u64x2 r;
    r.x = barrett_reduce128(x.x, 0, br);
    r.y = barrett_reduce128(x.y, 0, br);
    if (x.x > half) r.x = sub_mod(r.x, qlm, br.q);
    if (x.y > half) r.y = sub_mod(r.y, qlm, br.q);
    reinterpret_cast<u64x2*>(lifted)[(size_t)v * row + n2] = r;
}

// K9 ModRaise: centred lift of a single-limb polynomial (coefficient form, modulus q_src) into nl limbs.
// src [npoly][N] -> out [npoly][nl][N]
__global__ __launch_bounds__(256) void modraise_kernel(DeviceTables t, u64* out, const u64* src, int src_limb, int nl) {
    const int v = blockIdx.y;
    const int p = v / nl, tq = v % nl;
    const Barrett br = load_barrett(t, tq);
    const u64 qs = t.moduli[src_limb];
    const u64 half = qs >> 1;
    const u64 qsm = barrett_reduce128(qs, 0, br);
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(src)[(size_t)p * row + n2];
    u64x2 r;
    r.x = barrett_reduce128(x.x, 0, br);
    r.y = barrett_reduce128(x.y, 0, br);
    if (x.x > half) r.x = sub_mod(r.x, qsm, br.q);
    if (x.y > half) r.y = sub_mod(r.y, qsm, br.q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// K5 step 4: out[p][t] = (c[p][t] - lifted[p][t]) * q_l^{-1}   (c has ell limbs per poly, out ell-1)
__global__ __launch_bounds__(256) void rescale_finish_kernel(DeviceTables t, u64* out, const u64* c, const u64* lifted, int ell1,
                                                             const u64* qlinv_row) {
    const int v = blockIdx.y;
    const int p = v / ell1, tq = v % ell1;
    const u64 q = t.moduli[tq];
    const u64 w = qlinv_row[2 * tq], ws = qlinv_row[2 * tq + 1];
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const u64x2 x = reinterpret_cast<const u64x2*>(c)[((size_t)p * (ell1 + 1) + tq) * row + n2];
    const u64x2 y = reinterpret_cast<const u64x2*>(lifted)[(size_t)v * row + n2];
    u64x2 r;
    r.x = mul_shoup(sub_mod(x.x, y.x, q), w, ws, q);
    r.y = mul_shoup(sub_mod(x.y, y.y, q), w, ws, q);
    reinterpret_cast<u64x2*>(out)[(size_t)v * row + n2] = r;
}

// encode helper: signed 128-bit integer coefficients (lo, hi two's complement) -> residues of nlimbs limbs
__global__ __launch_bounds__(256) void reduce_i128_kernel(DeviceTables t, u64* out, const u64* coeffs, int limb_first) {
    const int v = blockIdx.y;
    const Barrett br = load_barrett(t, limb_first + v);
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t N = (size_t)1 << t.log_n;
    u64 lo = coeffs[2 * n], hi = coeffs[2 * n + 1];
    const bool neg = (hi >> 63) != 0;
    if (neg) {  // magnitude = -(hi:lo)
        lo = ~lo + 1;
        hi = ~hi + (lo == 0);
    }
    // |x| < 2^126: reduce the high word first so that the 128-bit Barrett input is < q * 2^64
    const u64 h = barrett_reduce128(hi, 0, br);
    u64 r = barrett_reduce128(lo, h, br);
    if (neg) r = neg_mod(r, br.q);
    out[(size_t)v * N + n] = r;
}

inline dim3 grid2(int log_n, int nvec) { return dim3((1u << log_n) / 512, (unsigned)nvec); }

}  // namespace

void launch_ew_items(const DeviceTables& t, const EwItems& it, int op, int limb_count, hipStream_t s) {
    if (it.n <= 0 || it.vecs <= 0) return;
    const dim3 g = grid2(t.log_n, it.n * it.vecs);
    switch (op) {
        case 0: hipLaunchKernelGGL((ew_items_kernel<0>), g, dim3(256), 0, s, t, it, limb_count); break;
        case 1: hipLaunchKernelGGL((ew_items_kernel<1>), g, dim3(256), 0, s, t, it, limb_count); break;
        case 2: hipLaunchKernelGGL((ew_items_kernel<2>), g, dim3(256), 0, s, t, it, limb_count); break;
        case 4: hipLaunchKernelGGL((ew_items_kernel<4>), g, dim3(256), 0, s, t, it, limb_count); break;
        default: hipLaunchKernelGGL((ew_items_kernel<3>), g, dim3(256), 0, s, t, it, limb_count); break;
    }
}
void launch_ew_dot(const DeviceTables& t, u64* out, const EwItems& it, int limb_count, hipStream_t s) {
    if (it.n <= 0 || it.vecs <= 0) return;
    hipLaunchKernelGGL(ew_dot_kernel, grid2(t.log_n, it.vecs), dim3(256), 0, s, t, out, it, limb_count);
}
void launch_ew_dot_groups(const DeviceTables& t, const EwDotGroups& din, hipStream_t s) {
    if (din.na <= 0 || din.ng <= 0 || din.ell <= 0) return;
    // the kernel reads through every pointer of the columns it is instantiated for: absent terms (null) and the columns beyond na point at
    // readable memory of the right shape and are left out through pmask
    EwDotGroups d = din;
    const int width = d.na <= 8 ? 8 : EwDotGroups::MAX_A;
    const u64* any = d.a[0];                       // [2][ell][N]: readable at every plaintext offset
    for (int g = 0; g < d.ng && any == d.a[0]; ++g)
        for (int b = 0; b < d.na; ++b)
            if (d.p[g][b]) {
                any = d.p[g][b];
                break;
            }
    for (int b = d.na; b < width; ++b) {
        d.a[b] = d.a[0];
        d.a_stride[b] = d.a_stride[0];
    }
    for (int g = 0; g < d.ng; ++g) {
        d.pmask[g] = 0;
        for (int b = 0; b < width; ++b) {
            if (b < d.na && d.p[g][b])
                d.pmask[g] |= 1u << b;
            else
                d.p[g][b] = any;
        }
    }
    const int nb = d.nbatch > 0 ? d.nbatch : 1;
    if (d.na <= 8)
        hipLaunchKernelGGL((ew_dot_groups_kernel<8>), grid2(t.log_n, 2 * d.ell * nb), dim3(256), 0, s, t, d);
    else
        hipLaunchKernelGGL((ew_dot_groups_kernel<16>), grid2(t.log_n, 2 * d.ell * nb), dim3(256), 0, s, t, d);
}
void launch_ew_cyclic_dot(const DeviceTables& t, const EwCyclic& din, hipStream_t s) {
    if (din.n <= 0 || din.ell <= 0) return;
    EwCyclic d = din;
    for (int i = d.n; i < EwCyclic::PERIOD; ++i) d.a[i] = d.a[0];   // the kernel loads through all 32 entries and drops the ones beyond n
    hipLaunchKernelGGL(ew_cyclic_dot_kernel, dim3((1u << t.log_n) / 256, (unsigned)d.ell, 2), dim3(256), 0, s, t, d);
}
void launch_ew_window_dot(const DeviceTables& t, const EwWindow& d, hipStream_t s) {
    if (d.ell <= 0) return;
    static const int group = [] { const char* e = std::getenv("FHELIN_WINDOW_GROUP"); return e ? std::atoi(e) : 2; }();
    const dim3 g((1u << t.log_n) / 256, (unsigned)d.ell, 2);
    switch (group) {
        case 1: hipLaunchKernelGGL(ew_window_dot_kernel<1>, g, dim3(256), 0, s, t, d); break;
        case 4: hipLaunchKernelGGL(ew_window_dot_kernel<4>, g, dim3(256), 0, s, t, d); break;
        default: hipLaunchKernelGGL(ew_window_dot_kernel<2>, g, dim3(256), 0, s, t, d); break;
    }
}
void launch_ew_mul(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL((ew_binary_kernel<0>), grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, b, b_mod, limb_first, limb_count);
}
void launch_ew_add(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL((ew_binary_kernel<1>), grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, b, b_mod, limb_first, limb_count);
}
void launch_ew_sub(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL((ew_binary_kernel<2>), grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, b, b_mod, limb_first, limb_count);
}
void launch_ew_muladd(const DeviceTables& t, u64* out, const u64* acc, const u64* a, const u64* b, int nvec, int b_mod, int limb_first,
                      int limb_count, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(ew_muladd_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t, out, acc, a, b, b_mod, limb_first, limb_count);
}
void launch_ew_neg(const DeviceTables& t, u64* out, const u64* a, int nvec, int limb_first, int limb_count, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(ew_neg_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, limb_first, limb_count);
}
void launch_ew_scalar(const DeviceTables& t, u64* out, const u64* a, const ScalarSet& sc, int nvec, int limb_first, int limb_count, hipStream_t s,
                      int in_limbs) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(ew_scalar_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, sc, limb_first, limb_count, in_limbs);
}
void launch_ew_addscalar(const DeviceTables& t, u64* out, const u64* a, const ScalarSet& sc, int nvec, int limb_first, int limb_count, hipStream_t s,
                         int add_vecs) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(ew_addscalar_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t, out, a, sc, limb_first, limb_count, add_vecs < 0 ? nvec : add_vecs);
}
void launch_ew_lincomb(const DeviceTables& t, u64* out, const LinComb& lc, const u64* scal, int ell, hipStream_t s, int in_limbs) {
    if (lc.n <= 0 || lc.vecs <= 0) return;
    hipLaunchKernelGGL(ew_lincomb_kernel, grid2(t.log_n, lc.vecs), dim3(256), 0, s, t, out, lc, scal, ell, in_limbs);
}
void launch_tensor(const DeviceTables& t, u64* d, const u64* a, const u64* b, int ell, hipStream_t s) {
    hipLaunchKernelGGL(tensor_kernel, grid2(t.log_n, ell), dim3(256), 0, s, t, d, a, b, ell);
}
void launch_tensor_items(const DeviceTables& t, const EwItems& it, int ell, hipStream_t s) {
    if (it.n <= 0) return;
    hipLaunchKernelGGL(tensor_items_kernel, dim3((1u << t.log_n) / 512, (unsigned)ell, (unsigned)it.n), dim3(256), 0, s, t, it, ell);
}
void launch_automorph(const DeviceTables& t, u64* out, const u64* in, const u32* map, int nvec, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(automorph_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t.log_n, out, in, map);
}
void launch_automorph_pack30(const DeviceTables& t, u64* out, const u64* in, const u32* map, int nvec, hipStream_t s) {
    if (nvec <= 0) return;
    hipLaunchKernelGGL(automorph_pack30_kernel, grid2(t.log_n, nvec), dim3(256), 0, s, t, out, in, map);
}
void launch_rescale_lift(const DeviceTables& t, u64* lifted, const u64* last, int npoly, int ell, const u64* qlmod_row, hipStream_t s) {
    hipLaunchKernelGGL(rescale_lift_kernel, grid2(t.log_n, npoly * (ell - 1)), dim3(256), 0, s, t, lifted, last, ell - 1, qlmod_row);
}
void launch_rescale_finish(const DeviceTables& t, u64* out, const u64* c, const u64* lifted, int npoly, int ell, const u64* qlinv_row,
                           hipStream_t s) {
    hipLaunchKernelGGL(rescale_finish_kernel, grid2(t.log_n, npoly * (ell - 1)), dim3(256), 0, s, t, out, c, lifted, ell - 1, qlinv_row);
}
void launch_modraise(const DeviceTables& t, u64* out, const u64* src, int npoly, int src_limb, int nl, hipStream_t s) {
    hipLaunchKernelGGL(modraise_kernel, grid2(t.log_n, npoly * nl), dim3(256), 0, s, t, out, src, src_limb, nl);
}
void launch_reduce_i128(const DeviceTables& t, u64* out, const u64* coeffs, int limb_first, int nlimbs, hipStream_t s) {
    hipLaunchKernelGGL(reduce_i128_kernel, dim3((1u << t.log_n) / 256, (unsigned)nlimbs), dim3(256), 0, s, t, out, coeffs, limb_first);
}

}  // namespace fhelin
