// K6/K7/K8 — hybrid (dnum-digit) key switching on gfx950: ModUp fast basis extension, evaluation-key inner
// product, ModDown.  Together with K1 these implement what OpenFHE's KeySwitchHYBRID does underneath
// context->EvalRotate / context->EvalMult(ct,ct) (reference src/FHEController.cpp:431,:435,:833,:843; HYBRID
// with SetNumLargeDigits(4), :11).  SURVEY.md §8(a) rows K6-K8.
//
// Exact functions computed (shared with oracle/fhe_oracle.c orc_keyswitch):
//   ModUp   d_j[t] = sum_{i in D_j} [c_i * (Q_j/q_i)^{-1}]_{q_i} * [(Q_j/q_i)]_t   mod t      (t outside digit j)
//   inner   acc_c[t] = sum_j d_j[t] * evk_j,c[t]                                    mod t
//   ModDown out_c[t] = (acc_c[t] - sum_p [acc_c[p] * (P/p)^{-1}]_p * [(P/p)]_t) * P^{-1}  mod q_t
// The sums are accumulated in 128 bits and reduced once (Barrett), so the results are canonical residues and
// independent of evaluation order.  Limb-major arrays, one thread per coefficient column, per-limb constants
// through scalar loads; no MFMA (u64 modular integers).
#include <hip/hip_runtime.h>
#include <cstdlib>
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

constexpr int TCH = 16; // targets per block: every z-chunk re-reads the digit's source limbs, so fewer/larger chunks cut traffic

// grid (N/256, beta, ceil((ell+k)/TCH)).  MAXA = digit size alpha (exact for alpha <= 8): full digits run the FULL body, whose
// loops carry no per-source conditions (the scalar branches otherwise cost about as much as the multiply-accumulates
// they guard); the shorter last digit takes the guarded body.
template <int MAXA, bool FULL>
__device__ __forceinline__ void modup_body(const DeviceTables& t, const KsShape& sh, u64* __restrict__ ext, const u64* __restrict__ cc,
                                           const u64* __restrict__ hatinv, const u64* __restrict__ hatmod, int j, int lo, int cnt, size_t n) {
    const size_t N = (size_t)1 << t.log_n;
    const int nt = sh.ell + sh.k;
    u32 y0[MAXA], y1[MAXA];  // y_i = [c_i * (Q_j/q_i)^{-1}]_{q_i}, split in 30-bit halves
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        if (FULL || i < cnt) {
            const int li = lo + i;
            split30(mul_shoup(cc[(size_t)li * N + n], hatinv[2 * li], hatinv[2 * li + 1], t.moduli[li]), y0[i], y1[i]);
        } else {
            y0[i] = y1[i] = 0;
        }
    }
    u64* dst = ext + (size_t)j * nt * N + n;
    const int t0 = blockIdx.z * sh.tch, t1 = min(nt, t0 + sh.tch);
    for (int tt = t0; tt < t1; ++tt) {
        if (tt >= lo && tt < lo + cnt) continue;  // own-digit slots: the inner product reads c (NTT form) directly
        const int limb = tt < sh.ell ? tt : sh.L1 + (tt - sh.ell);
        const u64 qt = t.moduli[limb], qti = t.qinv[limb];
        u64 slo = 0, shi = 0;
#pragma unroll
        for (int i0 = 0; i0 < MAXA; i0 += 8) {
            Acc30 acc = {0, 0, 0};
#pragma unroll
            for (int i = i0; i < i0 + 8 && i < MAXA; ++i)
                if (FULL || i < cnt) {
                    const u64 h = hatmod[(size_t)(lo + i) * nt + tt];  // pre-split on the host (pack30)
                    mac30(acc, y0[i], y1[i], (u32)h, (u32)(h >> 32));
                }
            acc30_flush(acc, slo, shi);
        }
        dst[(size_t)tt * N] = redc128(slo, shi, qt, qti);   // hatmod holds the constants times 2^64: the canonical residue of the plain sum
    }
}

template <int MAXA>
__global__ __launch_bounds__(256) void modup_conv_kernel(DeviceTables t, KsShape sh, u64* __restrict__ ext, const u64* __restrict__ cc, const u64* __restrict__ c_ntt,
                                                         const u64* __restrict__ hatinv, const u64* __restrict__ hatmod) {
    const int bi = blockIdx.y / sh.beta, j = blockIdx.y % sh.beta;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lo = j * sh.alpha;
    const int cnt = min(sh.alpha, sh.ell - lo);
    const int nt = sh.ell + sh.k;
    cc += (size_t)bi * sh.ell * N;
    ext += (size_t)bi * sh.beta * nt * N;
    (void)c_ntt;
    if (cnt == MAXA)
        modup_body<MAXA, true>(t, sh, ext, cc, hatinv, hatmod, j, lo, cnt, n);
    else
        modup_body<MAXA, false>(t, sh, ext, cc, hatinv, hatmod, j, lo, cnt, n);
}

// Block -> (coefficient tile, target limb, batch row).  The workgroups of a launch are dealt round-robin to the 8 XCDs (each
// with its own L2) in linear block order; taking the linear id modulo 8 as the XCD and ordering each XCD's share as
// (limb, tile, row) with the row fastest makes the B rows that share one key tile run back to back on ONE XCD (the tile is
// fetched once instead of once per row and XCD), and keeps all gathers of one (limb, row) plane on one XCD.  Placement only.
struct KsBlock {
    int bx, bi, tt;
};
__device__ __forceinline__ KsBlock ks_block(const KsShape& sh, int rows = 0) {
    const unsigned nx = gridDim.x, nblk = gridDim.x * gridDim.y;
    const unsigned nb = rows ? (unsigned)rows : (unsigned)sh.batch;   // batch rows (or row pairs) of the launch
    unsigned b = blockIdx.y * nx + blockIdx.x;
    if ((nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
    KsBlock k;
    k.bi = (int)(b % nb);
    k.bx = (int)((b / nb) % nx);
    k.tt = (int)(b / (nb * nx));
    return k;
}

// grid (N/512, ell + k)
__global__ __launch_bounds__(256) void ks_inner_kernel(DeviceTables t, KsShape sh, u64* __restrict__ accQ, u64* __restrict__ accP, const u64* __restrict__ ext,
                                                       const u64* __restrict__ evk, const u64* __restrict__ c_ntt) {
    const int nt = sh.ell + sh.k;
    const KsBlock kb_ = ks_block(sh);
    const int bi = kb_.bi, tt = kb_.tt;
    const int limb = tt < sh.ell ? tt : sh.L1 + (tt - sh.ell);
    if (sh.row_mod) {
        const int in = bi / sh.row_mod;
        ext += (size_t)in * sh.ext_batch_stride;
        c_ntt += (size_t)in * sh.c_stride;
        evk = sh.evk_row[bi % sh.row_mod];
    } else {
        if (!sh.shared_input) {
            ext += (size_t)bi * sh.beta * nt * ((size_t)1 << t.log_n);
            c_ntt += (size_t)bi * sh.c_stride;
        }
        if (sh.per_row) evk = sh.evk_row[bi];
    }
    const int own = tt < sh.ell ? tt / sh.alpha : -1;  // the digit that contains target limb tt (its slot in ext is unused)
    accQ += (size_t)bi * 2 * sh.ell * ((size_t)1 << t.log_n);
    accP += (size_t)bi * 2 * sh.k * ((size_t)1 << t.log_n);
    const Barrett br = load_barrett(t, limb);
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t n2 = (size_t)kb_.bx * 256 + threadIdx.x;
    const size_t kstride = (size_t)(sh.L1 + sh.k) * row;  // one evk component, in u64x2 units
    const u64x2* E = reinterpret_cast<const u64x2*>(ext);
    const u64x2* K = reinterpret_cast<const u64x2*>(evk);
    u64 lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};  // b.x, b.y, a.x, a.y
    const u64 monR = t.mont[2 * limb], monRs = t.mont[2 * limb + 1];
    for (int j0 = 0; j0 < sh.beta; j0 += 8) {
        Acc30 b0 = {0, 0, 0}, b1 = {0, 0, 0}, a0 = {0, 0, 0}, a1 = {0, 0, 0};
        const int j1 = min(sh.beta, j0 + 8);
        for (int j = j0; j < j1; ++j) {
            u64x2 d = j == own ? reinterpret_cast<const u64x2*>(c_ntt)[(size_t)tt * row + n2] : E[((size_t)j * nt + tt) * row + n2];
            if (j == own) {   // the digits of this kernel's ModUp come times 2^64 (LevelTables::up_hatmod_r2); the ciphertext's own limb gets the factor here
                d.x = mul_shoup(d.x, monR, monRs, br.q);
                d.y = mul_shoup(d.y, monR, monRs, br.q);
            }
            const u64x2 kb = K[(size_t)(2 * j) * kstride + (size_t)limb * row + n2];
            const u64x2 ka = K[(size_t)(2 * j + 1) * kstride + (size_t)limb * row + n2];
            u32 dx0, dx1, dy0, dy1, k0, k1;
            split30(d.x, dx0, dx1);
            split30(d.y, dy0, dy1);
            split30(kb.x, k0, k1);
            mac30(b0, dx0, dx1, k0, k1);
            split30(kb.y, k0, k1);
            mac30(b1, dy0, dy1, k0, k1);
            split30(ka.x, k0, k1);
            mac30(a0, dx0, dx1, k0, k1);
            split30(ka.y, k0, k1);
            mac30(a1, dy0, dy1, k0, k1);
        }
        acc30_flush(b0, lo[0], hi[0]);
        acc30_flush(b1, lo[1], hi[1]);
        acc30_flush(a0, lo[2], hi[2]);
        acc30_flush(a1, lo[3], hi[3]);
    }
    u64x2 rb, ra;
    const u64 qi = t.qinv[limb];       // every term carries 2^64 through its digit: the Montgomery reduction returns the plain sum's residue
    rb.x = redc128(lo[0], hi[0], br.q, qi);
    rb.y = redc128(lo[1], hi[1], br.q, qi);
    ra.x = redc128(lo[2], hi[2], br.q, qi);
    ra.y = redc128(lo[3], hi[3], br.q, qi);
    if (tt < sh.ell) {
        u64x2* O = reinterpret_cast<u64x2*>(accQ);
        O[(size_t)tt * row + n2] = rb;
        O[(size_t)(sh.ell + tt) * row + n2] = ra;
    } else {
        u64x2* O = reinterpret_cast<u64x2*>(accP);
        const int pj = tt - sh.ell;
        O[(size_t)pj * row + n2] = rb;
        O[(size_t)(sh.k + pj) * row + n2] = ra;
    }
}

// r = mask_bit(lane) ? b : a with the lane mask in SGPRs (the VCC form of v_cndmask issues ~5x slower on gfx950)
__device__ __forceinline__ u32 sel_mask(u32 a, u32 b, unsigned long long mask) {
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}

// grid (N/512, batch*(ell + k)): the coefficient pair (2m, 2m+1) per thread.  In the bit-reversed evaluation order an
// automorphism map satisfies map[2m+1] = map[2m] ^ 1 (the two slots differ by psi^N = -1 in the evaluation point, which
// multiplication of the exponent by the odd g preserves): a pair is gathered with ONE 16-byte load and swapped in
// registers when map[2m] is odd.  The keys come pre-permuted (KsShape::evk_rot): their loads are contiguous.
// STAGE (KsShape::lds_digits): every rotation of the launch maps a 512-coefficient tile onto ITSELF (Galois element = 1 mod
// N/256: rotations by multiples of 64 slots at N = 2^16 - the 128 s and 512 s units of the matmul trees) and all rotations read
// the same digits: the workgroup's beta digit tiles are loaded ONCE, coalesced, into LDS and every rotation gathers from there.
// Without it each rotation re-gathers the tile through L2, where the key stream (112 KB per tile) keeps evicting it.
// ROWS = 2: a thread carries TWO batch rows (bi, bi + 1) of the same (limb, coefficient pair): the rows share the rotation keys, so every
// key word is loaded once for two multiply-accumulate sets (the loop was paced by its three loads per four products); an odd
// last row runs with its partner switched off (computed on the row's own data, not stored).
template <bool STAGE, int ROWS>
__global__ __launch_bounds__(256) void ks_inner_multi_kernel(DeviceTables t, KsShape sh, u64* __restrict__ accQ, u64* __restrict__ accP,
                                                             const u64* __restrict__ ext, const u64* __restrict__ c_ntt) {
    __shared__ u64x2 dl[STAGE ? ROWS : 1][STAGE ? 4 : 1][256];
    const int nt = sh.ell + sh.k;
    const KsBlock kb_ = ks_block(sh, (sh.batch + ROWS - 1) / ROWS);
    const int bi = kb_.bi * ROWS, tt = kb_.tt;
    const int limb = tt < sh.ell ? tt : sh.L1 + (tt - sh.ell);
    const size_t N = (size_t)1 << t.log_n, row = N >> 1;
    const size_t ext_stride = sh.ext_batch_stride ? sh.ext_batch_stride : (size_t)sh.beta * nt * N;
    const int own = tt < sh.ell ? tt / sh.alpha : -1;
    const Barrett br = load_barrett(t, limb);
    const size_t n2 = (size_t)kb_.bx * 256 + threadIdx.x;
    const size_t kstride = (size_t)(sh.L1 + sh.k) * row;  // one evk component, in u64x2 units
    bool live[ROWS];
    const u64* extr[ROWS];
    const u64* cr_[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        live[q] = bi + q < sh.batch;
        const int b = live[q] ? bi + q : bi;
        extr[q] = ext + (size_t)b * ext_stride;
        cr_[q] = c_ntt + (size_t)b * sh.c_stride;
    }
    u64 lo[ROWS][4], hi[ROWS][4];      // b.x, b.y, a.x, a.y
    Acc30 acc[ROWS][4];
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[q][i] = hi[q][i] = 0;
            acc[q][i] = Acc30{0, 0, 0};
        }
    int pending = 0, folded = 0;
    if constexpr (STAGE) {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const u64x2* __restrict__ e0 = reinterpret_cast<const u64x2*>(extr[q]);
            const u64x2* __restrict__ c0 = reinterpret_cast<const u64x2*>(cr_[q]);
            for (int j = 0; j < sh.beta; ++j) dl[q][j][threadIdx.x] = j == own ? c0[(size_t)tt * row + n2] : e0[((size_t)j * nt + tt) * row + n2];
        }
        __syncthreads();
    }
    // The (rotation, digit) iterations form ONE software-pipelined loop: the loads of iteration i+1 (a 16-byte digit gather per row and two
    // key words) are issued before the multiply-accumulates of iteration i, and the map entry of rotation r+1 one rotation ahead.  A wave
    // then keeps six loads in flight instead of three; with three the kernel ran at the pace of one HBM round trip per iteration
    // (152 waves per SIMD in rounds of 8 x 14 iterations x ~1.4 us = the 387 us measured in round 2).
    struct Operands {
        u64x2 d[ROWS], kb, ka;
    };
    auto map_of = [&](int r) -> u32 { return sh.map_rot[r] ? sh.map_rot[r][2 * n2] : (u32)(2 * n2); };
    auto issue = [&](int r, int j, u32 m0) {
        const size_t mp = m0 >> 1;
        const u64x2* __restrict__ K = reinterpret_cast<const u64x2*>(sh.evk_rot[r]) + (size_t)limb * row + n2;
        Operands o;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if constexpr (STAGE) {
                o.d[q] = dl[q][j][mp & 255];
            } else {
                const u64x2* __restrict__ er = reinterpret_cast<const u64x2*>(extr[q] + (size_t)r * sh.rot_ext_stride);
                const u64x2* __restrict__ cr = reinterpret_cast<const u64x2*>(cr_[q] + (size_t)r * sh.rot_input_stride);
                o.d[q] = j == own ? cr[(size_t)tt * row + mp] : er[((size_t)j * nt + tt) * row + mp];
            }
        }
        o.kb = K[(size_t)(2 * j) * kstride];
        o.ka = K[(size_t)(2 * j + 1) * kstride];
        return o;
    };
    const int total = sh.n_rot * sh.beta;
    int r = 0, j = 0;
    u32 m_cur = map_of(0), m_nxt = sh.n_rot > 1 ? map_of(1) : 0;
    Operands cur = issue(0, 0, m_cur);
    for (int it = 0; it < total; ++it) {
        int rn = r, jn = j + 1;
        u32 mn = m_cur;
        if (jn == sh.beta) {
            jn = 0;
            rn = r + 1;
            mn = m_nxt;
        }
        Operands nxt = cur;
        if (it + 1 < total) {
            nxt = issue(rn, jn, mn);
            if (jn == 0 && rn + 1 < sh.n_rot) m_nxt = map_of(rn + 1);
        }
        {
            const unsigned long long swp = __ballot((m_cur & 1) != 0);
            const u64x2 kb = cur.kb, ka = cur.ka;
            // the key words come pre-split (pack30: low half in the low dword, high half in the high dword)
            const u32 kbx0 = (u32)kb.x, kbx1 = (u32)(kb.x >> 32), kby0 = (u32)kb.y, kby1 = (u32)(kb.y >> 32);
            const u32 kax0 = (u32)ka.x, kax1 = (u32)(ka.x >> 32), kay0 = (u32)ka.y, kay1 = (u32)(ka.y >> 32);
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                const u64x2 d = cur.d[q];
                u32 p0, p1, q0, q1;
                split30(d.x, p0, p1);
                split30(d.y, q0, q1);
                // element 2m of the rotated digit is d.x, or d.y when the map sends 2m to an odd position
                const u32 dx0 = sel_mask(p0, q0, swp), dx1 = sel_mask(p1, q1, swp), dy0 = sel_mask(q0, p0, swp), dy1 = sel_mask(q1, p1, swp);
                mac30(acc[q][0], dx0, dx1, kbx0, kbx1);
                mac30(acc[q][1], dy0, dy1, kby0, kby1);
                mac30(acc[q][2], dx0, dx1, kax0, kax1);
                mac30(acc[q][3], dy0, dy1, kay0, kay1);
            }
            if (++pending == 8) {  // at most 8 products of 60-bit halves fit the 64-bit columns
                pending = 0;
                // barrett_reduce128 needs a sum below q * 2^64: 16 products of operands below 2^60 (special limbs:
                // canonical digits and keys, 16 p^2 < p * 2^64; scaling limbs: digits below 86q, 16 * 86 q^2 < q * 2^64
                // for q < 2^53) plus one carried residue.  R * beta can reach 28 (giant steps): fold every 16.
                const bool fold = ++folded == 2;
                if (fold) folded = 0;
#pragma unroll
                for (int q = 0; q < ROWS; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc30_flush(acc[q][i], lo[q][i], hi[q][i]);
                        acc[q][i] = Acc30{0, 0, 0};
                        if (fold) {
                            lo[q][i] = barrett_reduce128(lo[q][i], hi[q][i], br);
                            hi[q][i] = 0;
                        }
                    }
            }
        }
        cur = nxt;
        r = rn;
        j = jn;
        m_cur = mn;
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        if (!live[q]) continue;
        u64x2 rb, ra;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc30_flush(acc[q][i], lo[q][i], hi[q][i]);
        // the keys of this kernel (EvalKey::d_perm, folded keys) are stored times 2^64: the Montgomery reduction returns the canonical residue
        // of the plain sum at a third of a Barrett reduction's instructions (the folds above keep the factor: they reduce, they do not divide)
        const u64 qi = t.qinv[limb];
        rb.x = redc128(lo[q][0], hi[q][0], br.q, qi);
        rb.y = redc128(lo[q][1], hi[q][1], br.q, qi);
        ra.x = redc128(lo[q][2], hi[q][2], br.q, qi);
        ra.y = redc128(lo[q][3], hi[q][3], br.q, qi);
        if (tt < sh.ell) {
            u64x2* O = reinterpret_cast<u64x2*>(accQ + (size_t)(bi + q) * 2 * sh.ell * N);
            O[(size_t)tt * row + n2] = rb;
            O[(size_t)(sh.ell + tt) * row + n2] = ra;
        } else {
            u64x2* O = reinterpret_cast<u64x2*>(accP + (size_t)(bi + q) * 2 * sh.k * N);
            const int pj = tt - sh.ell;
            O[(size_t)pj * row + n2] = rb;
            O[(size_t)(sh.k + pj) * row + n2] = ra;
        }
    }
}

// grid (N/256, batch*ell)
__global__ __launch_bounds__(256) void gather_sum_kernel(DeviceTables t, KsShape sh, u64* __restrict__ out, const u64* __restrict__ in,
                                                         size_t in_stride) {
    const int bi = blockIdx.y / sh.ell, tt = blockIdx.y % sh.ell;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64 q = t.moduli[tt];
    const u64* src = in + (size_t)bi * in_stride + (size_t)tt * N;
    u64 acc = 0;
    for (int r = 0; r < sh.n_rot; ++r) {
        const u64* sr = src + (size_t)r * sh.rot_input_stride;
        acc = add_mod(acc, sr[sh.map_rot[r] ? (size_t)sh.map_rot[r][n] : n], q);
    }
    out[((size_t)bi * sh.ell + tt) * N + n] = acc;
}

// grid (N/256, 2, ceil(ell/TCH)).  MAXK = number of special limbs (exact for k <= 8, then FULL: no per-source conditions)
template <int MAXK, bool FULL>
__device__ __forceinline__ void moddown_body(const DeviceTables& t, const KsShape& sh, u64* __restrict__ conv, const u64* __restrict__ accP,
                                             const u64* __restrict__ phatinv, const u64* __restrict__ phatmod, int c, size_t n) {
    const size_t N = (size_t)1 << t.log_n;
    u32 z0[MAXK], z1[MAXK];
#pragma unroll
    for (int p = 0; p < MAXK; ++p) {
        if (FULL || p < sh.k) {
            split30(mul_shoup(accP[((size_t)c * sh.k + p) * N + n], phatinv[2 * p], phatinv[2 * p + 1], t.moduli[sh.L1 + p]), z0[p], z1[p]);
        } else {
            z0[p] = z1[p] = 0;
        }
    }
    u64* dst = conv + (size_t)c * sh.ell * N + n;
    const int t0 = blockIdx.z * sh.tch, t1 = min(sh.ell, t0 + sh.tch);
    for (int tt = t0; tt < t1; ++tt) {
        const u64 qt = t.moduli[tt], qti = t.qinv[tt];
        u64 slo = 0, shi = 0;
#pragma unroll
        for (int p0 = 0; p0 < MAXK; p0 += 8) {
            Acc30 acc = {0, 0, 0};
#pragma unroll
            for (int p = p0; p < p0 + 8 && p < MAXK; ++p)
                if (FULL || p < sh.k) {
                    const u64 h = phatmod[(size_t)p * sh.L1 + tt];  // pre-split on the host (pack30)
                    mac30(acc, z0[p], z1[p], (u32)h, (u32)(h >> 32));
                }
            acc30_flush(acc, slo, shi);
        }
        dst[(size_t)tt * N] = redc128(slo, shi, qt, qti);   // phatmod holds the constants times 2^64
    }
}

template <int MAXK>
__global__ __launch_bounds__(256) void moddown_conv_kernel(DeviceTables t, KsShape sh, u64* __restrict__ conv, const u64* __restrict__ accP, const u64* __restrict__ phatinv,
                                                           const u64* __restrict__ phatmod) {
    const int bi = blockIdx.y >> 1, c = blockIdx.y & 1;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    accP += (size_t)bi * 2 * sh.k * N;
    conv += (size_t)bi * 2 * sh.ell * N;
    if (sh.k == MAXK)
        moddown_body<MAXK, true>(t, sh, conv, accP, phatinv, phatmod, c, n);
    else
        moddown_body<MAXK, false>(t, sh, conv, accP, phatinv, phatmod, c, n);
}

// grid (N/512, 2*ell)
__global__ __launch_bounds__(256) void moddown_finish_kernel(DeviceTables t, KsShape sh, u64* __restrict__ out, const u64* __restrict__ accQ, const u64* __restrict__ conv,
                                                             const u64* __restrict__ pinv, const u64* __restrict__ add0, const u64* __restrict__ add1, const u32* __restrict__ map,
                                                             const u64* __restrict__ post) {
    const int bi = blockIdx.y / (2 * sh.ell), v = blockIdx.y % (2 * sh.ell);
    const int c = v / sh.ell, tt = v % sh.ell;
    const u64 q = t.moduli[tt];
    const u64 w = pinv[2 * tt], ws = pinv[2 * tt + 1];
    const size_t N = (size_t)1 << t.log_n;
    const size_t j = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    accQ += (size_t)bi * 2 * sh.ell * N;
    conv += (size_t)bi * 2 * sh.ell * N;
    out += (size_t)bi * sh.out_stride;
    const u64* add = c == 0 ? add0 : add1;
    if (add) add += (size_t)(sh.row_mod ? bi / sh.row_mod : bi) * sh.add_stride;
    if (post) post += (size_t)bi * sh.post_stride;
    if (sh.per_row) map = sh.map_row[sh.row_mod ? bi % sh.row_mod : bi];
    u64x2 r;
    if (map) {
        const u32 m0 = map[j], m1 = map[j + 1];
        r.x = mul_shoup(sub_mod(accQ[(size_t)v * N + m0], conv[(size_t)v * N + m0], q), w, ws, q);
        r.y = mul_shoup(sub_mod(accQ[(size_t)v * N + m1], conv[(size_t)v * N + m1], q), w, ws, q);
        if (add) {
            r.x = add_mod(r.x, add[(size_t)tt * N + m0], q);
            r.y = add_mod(r.y, add[(size_t)tt * N + m1], q);
        }
    } else {
        const u64x2 a = reinterpret_cast<const u64x2*>(accQ)[((size_t)v * N + j) >> 1];
        const u64x2 b = reinterpret_cast<const u64x2*>(conv)[((size_t)v * N + j) >> 1];
        r.x = mul_shoup(sub_mod(a.x, b.x, q), w, ws, q);
        r.y = mul_shoup(sub_mod(a.y, b.y, q), w, ws, q);
        if (add) {
            const u64x2 d = reinterpret_cast<const u64x2*>(add)[((size_t)tt * N + j) >> 1];
            r.x = add_mod(r.x, d.x, q);
            r.y = add_mod(r.y, d.y, q);
        }
    }
    if (post) {
        const u64x2 p = reinterpret_cast<const u64x2*>(post)[((size_t)v * N + j) >> 1];
        r.x = add_mod(r.x, p.x, q);
        r.y = add_mod(r.y, p.y, q);
    }
    if (sh.gsrc && c == 0) {  // merged rotation sum: + sum_r sigma_r(c0), gathered here
        const u64* __restrict__ g = sh.gsrc + (size_t)bi * sh.gsrc_stride + (size_t)tt * N;
        for (int rr = 0; rr < sh.n_rot; ++rr) {
            const u64* __restrict__ gr = g + (size_t)rr * sh.rot_input_stride;
            const u32* __restrict__ mr = sh.map_rot[rr];
            const u32 m0 = mr ? mr[j] : (u32)j;                          // map[j + 1] = map[j] ^ 1: one 16-byte load per pair
            const u64x2 gp = reinterpret_cast<const u64x2*>(gr)[m0 >> 1];
            r.x = add_mod(r.x, (m0 & 1) ? gp.y : gp.x, q);
            r.y = add_mod(r.y, (m0 & 1) ? gp.x : gp.y, q);
        }
    }
    reinterpret_cast<u64x2*>(out)[((size_t)v * N + j) >> 1] = r;
}

// Plaintext-folded rotation key (launch_fold_key): grid (N/512, digits * 2 * (L1 + k)); vector v belongs to limb v % (L1 + k)
__global__ __launch_bounds__(256) void fold_key_kernel(DeviceTables t, u64* __restrict__ out, const u64* __restrict__ key, const u32* __restrict__ map,
                                                       const u64* __restrict__ V) {
    const size_t v = blockIdx.y;
    const int limb = (int)(v % (size_t)t.n_limbs);
    const Barrett br = load_barrett(t, limb);
    const size_t N = (size_t)1 << t.log_n;
    const size_t j = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const u32 m0 = map[j], m1 = map[j + 1];
    const u64x2 pv = reinterpret_cast<const u64x2*>(V)[((size_t)limb * N + j) >> 1];
    u64x2 r;
    const u64 R = t.mont[2 * limb], Rs = t.mont[2 * limb + 1];   // times 2^64, like EvalKey::d_perm: ks_inner_multi ends in redc128
    r.x = pack30(mul_shoup(mul_mod(pv.x, key[v * N + m0], br), R, Rs, br.q));
    r.y = pack30(mul_shoup(mul_mod(pv.y, key[v * N + m1], br), R, Rs, br.q));
    reinterpret_cast<u64x2*>(out)[(v * N + j) >> 1] = r;
}

// Addends of a hoisted plaintext-rotation sum (launch_hoist_addends): grid (N/512, batch * 2 * ell)
__global__ __launch_bounds__(256) void hoist_addends_kernel(DeviceTables t, KsShape sh, HoistAdd h, u64* __restrict__ pre, const u64* __restrict__ ct) {
    const int bi = blockIdx.y / (2 * sh.ell), v = blockIdx.y % (2 * sh.ell);
    const int c = v / sh.ell, tt = v % sh.ell;
    const Barrett br = load_barrett(t, tt);
    const size_t N = (size_t)1 << t.log_n, row = N >> 1;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64x2* __restrict__ src = reinterpret_cast<const u64x2*>(ct + (size_t)bi * sh.c_stride + (size_t)v * N);
    const u64x2 x = src[n2];
    const u64x2 p0 = reinterpret_cast<const u64x2*>(h.v[0])[(size_t)tt * row + n2];
    Acc128 a0 = {0, 0}, a1 = {0, 0};     // <= 8 products of canonical residues below 2^60: far below q * 2^64
    acc_mac(a0, x.x, p0.x);
    acc_mac(a1, x.y, p0.y);
    if (c == 0)
        for (int r = 0; r < h.n_rot; ++r) {
            const u32 m0 = h.map[r][2 * n2];                           // map[2m + 1] = map[2m] ^ 1: one 16-byte gather per pair
            const u64x2 g = src[m0 >> 1];
            const u64x2 pr = reinterpret_cast<const u64x2*>(h.v[r + 1])[(size_t)tt * row + n2];
            acc_mac(a0, (m0 & 1) ? g.y : g.x, pr.x);
            acc_mac(a1, (m0 & 1) ? g.x : g.y, pr.y);
        }
    u64x2 o;
    o.x = barrett_reduce128(a0.lo, a0.hi, br);
    o.y = barrett_reduce128(a1.lo, a1.hi, br);
    if (h.acc) {   // into the key switch's accumulator, multiplied by P
        u64x2* A = reinterpret_cast<u64x2*>(h.acc + (size_t)bi * 2 * sh.ell * N) + (size_t)v * row + n2;
        const u64 w = h.pmod[2 * tt], ws = h.pmod[2 * tt + 1];
        u64x2 a = *A;
        a.x = add_mod(a.x, mul_shoup(o.x, w, ws, br.q), br.q);
        a.y = add_mod(a.y, mul_shoup(o.y, w, ws, br.q), br.q);
        *A = a;
        return;
    }
    reinterpret_cast<u64x2*>(pre + (size_t)bi * 2 * sh.ell * N)[(size_t)v * row + n2] = o;
}

// grid (N/512, batch * 2 * ell)
__global__ __launch_bounds__(256) void affine_acc_kernel(DeviceTables t, KsShape sh, u64* __restrict__ accQ, const u64* __restrict__ d,
                                                         const u64* __restrict__ sub, ScalarSet cst, int has_cst, int f, const u64* __restrict__ pmod) {
    const int bi = blockIdx.y / (2 * sh.ell), v = blockIdx.y % (2 * sh.ell);
    const int c = v / sh.ell, tt = v % sh.ell;
    const u64 q = t.moduli[tt];
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    u64x2 x = reinterpret_cast<const u64x2*>(d)[((size_t)bi * 3 * sh.ell + v) * row + n2];
    if (f == 2) {
        x.x = add_mod(x.x, x.x, q);
        x.y = add_mod(x.y, x.y, q);
    }
    if (has_cst && c == 0) {
        x.x = add_mod(x.x, cst.v[2 * tt], q);
        x.y = add_mod(x.y, cst.v[2 * tt], q);
    }
    if (sub) {
        const u64x2 sv = reinterpret_cast<const u64x2*>(sub)[((size_t)bi * 2 * sh.ell + v) * row + n2];
        x.x = sub_mod(x.x, sv.x, q);
        x.y = sub_mod(x.y, sv.y, q);
    }
    u64x2* A = reinterpret_cast<u64x2*>(accQ) + ((size_t)bi * 2 * sh.ell + v) * row + n2;
    u64x2 a = *A;
    if (f == 2) {
        a.x = add_mod(a.x, a.x, q);
        a.y = add_mod(a.y, a.y, q);
    }
    const u64 w = pmod[2 * tt], ws = pmod[2 * tt + 1];
    a.x = add_mod(a.x, mul_shoup(x.x, w, ws, q), q);
    a.y = add_mod(a.y, mul_shoup(x.y, w, ws, q), q);
    *A = a;
}

// ModDown + rescale, conversion (launch_moddown_rescale_conv): grid (N/256, batch * 2, ceil((ell-1)/TCH)); sources = the k special
// limbs of accP and the top Q limb (coefficient form), targets t < ell - 1
__global__ __launch_bounds__(256) void moddown_rescale_conv_kernel(DeviceTables t, KsShape sh, u64* __restrict__ conv, const u64* __restrict__ accP,
                                                                   const u64* __restrict__ top, const u64* __restrict__ hatinv,
                                                                   const u64* __restrict__ hatmod, const u64* __restrict__ mmod) {
    constexpr int MAXS = 16;
    const int bi = blockIdx.y >> 1, c = blockIdx.y & 1;
    const size_t N = (size_t)1 << t.log_n;
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int e1 = sh.ell - 1, ns = sh.k + 1;
    u32 z0[MAXS], z1[MAXS];
    u32 neg = 0;        // sources taken as y_b - b (centred conversion)
#pragma unroll
    for (int p = 0; p < MAXS; ++p) {
        if (p < ns) {
            const u64 x = p < sh.k ? accP[(((size_t)bi * 2 + c) * sh.k + p) * N + n] : top[((size_t)bi * 2 + c) * N + n];
            const u64 m = t.moduli[p < sh.k ? sh.L1 + p : sh.ell - 1];
            const u64 y = mul_shoup(x, hatinv[2 * p], hatinv[2 * p + 1], m);
            neg += y > (m >> 1);
            split30(y, z0[p], z1[p]);
        } else {
            z0[p] = z1[p] = 0;
        }
    }
    u64* dst = conv + ((size_t)bi * 2 + c) * e1 * N + n;
    const int t0 = blockIdx.z * sh.tch, t1 = min(e1, t0 + sh.tch);
    for (int tt = t0; tt < t1; ++tt) {
        const Barrett br = load_barrett(t, tt);
        u64 slo = 0, shi = 0;
#pragma unroll
        for (int p0 = 0; p0 < MAXS; p0 += 8) {
            Acc30 acc = {0, 0, 0};
#pragma unroll
            for (int p = p0; p < p0 + 8; ++p)
                if (p < ns) {
                    const u64 hm = hatmod[(size_t)p * e1 + tt];  // pre-split on the host (pack30)
                    mac30(acc, z0[p], z1[p], (u32)hm, (u32)(hm >> 32));
                }
            acc30_flush(acc, slo, shi);
        }
        dst[(size_t)tt * N] = sub_mod(barrett_reduce128(slo, shi, br), barrett_reduce128((u64)neg * mmod[tt], 0, br), br.q);
    }
}

// grid (N/512, batch * 2 * (ell-1))
__global__ __launch_bounds__(256) void moddown_rescale_finish_kernel(DeviceTables t, KsShape sh, u64* __restrict__ out, const u64* __restrict__ accQ,
                                                                     const u64* __restrict__ conv, const u64* __restrict__ minv) {
    const int e1 = sh.ell - 1;
    const int bi = blockIdx.y / (2 * e1), v = blockIdx.y % (2 * e1);
    const int c = v / e1, tt = v % e1;
    const u64 q = t.moduli[tt];
    const u64 w = minv[2 * tt], ws = minv[2 * tt + 1];
    const size_t row = ((size_t)1 << t.log_n) >> 1;
    const size_t n2 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64x2 a = reinterpret_cast<const u64x2*>(accQ)[(((size_t)bi * 2 + c) * sh.ell + tt) * row + n2];
    const u64x2 b = reinterpret_cast<const u64x2*>(conv)[(((size_t)bi * 2 + c) * e1 + tt) * row + n2];
    u64x2 r;
    r.x = mul_shoup(sub_mod(a.x, b.x, q), w, ws, q);
    r.y = mul_shoup(sub_mod(a.y, b.y, q), w, ws, q);
    reinterpret_cast<u64x2*>(out + (size_t)bi * sh.out_stride)[(size_t)v * row + n2] = r;
}

}  // namespace

void launch_fold_key(const DeviceTables& t, u64* out, const u64* key, const u32* map, const u64* V, int nvec, hipStream_t s) {
    hipLaunchKernelGGL(fold_key_kernel, dim3((1u << t.log_n) / 512, (unsigned)nvec), dim3(256), 0, s, t, out, key, map, V);
}
void launch_hoist_addends(const DeviceTables& t, const KsShape& sh, const HoistAdd& h, u64* pre, const u64* ct, hipStream_t s) {
    hipLaunchKernelGGL(hoist_addends_kernel, dim3((1u << t.log_n) / 512, (unsigned)(sh.batch * 2 * sh.ell)), dim3(256), 0, s, t, sh, h, pre, ct);
}

void launch_affine_acc(const DeviceTables& t, const KsShape& sh, u64* accQ, const u64* d, const u64* sub, const ScalarSet& cst, int has_cst, int f,
                       const u64* pmod, hipStream_t s) {
    hipLaunchKernelGGL(affine_acc_kernel, dim3((1u << t.log_n) / 512, (unsigned)(sh.batch * 2 * sh.ell)), dim3(256), 0, s, t, sh, accQ, d, sub, cst,
                       has_cst, f, pmod);
}
// targets per block of a conversion launch with `planes` (coefficient tile x row x ...) blocks per chunk: all targets in one block when
// the launch fills the GPU that way (the sources are then read once), chunks of TCH otherwise.  FHELIN_CONV_CHUNKS=1: always chunks (A/B).
static int conv_tch(unsigned planes, int targets) {
    static const int force = [] {
        const char* e = std::getenv("FHELIN_CONV_CHUNKS");
        return e ? std::atoi(e) : 0;
    }();
    if (force || planes < 2048u) return TCH;
    return targets > TCH ? targets : TCH;
}
void launch_moddown_rescale_conv(const DeviceTables& t, const KsShape& sh_in, u64* conv, const u64* accP, const u64* top, const u64* hatinv,
                                 const u64* hatmod, const u64* mmod, hipStream_t s) {
    KsShape sh = sh_in;
    const unsigned nx = (1u << t.log_n) / 256;
    sh.tch = conv_tch(nx * (unsigned)(sh.batch * 2), sh.ell - 1);
    dim3 g(nx, (unsigned)(sh.batch * 2), (unsigned)((sh.ell - 1 + sh.tch - 1) / sh.tch));
    hipLaunchKernelGGL(moddown_rescale_conv_kernel, g, dim3(256), 0, s, t, sh, conv, accP, top, hatinv, hatmod, mmod);
}
void launch_moddown_rescale_finish(const DeviceTables& t, const KsShape& sh, u64* out, const u64* accQ, const u64* conv, const u64* minv,
                                   hipStream_t s) {
    dim3 g((1u << t.log_n) / 512, (unsigned)(sh.batch * 2 * (sh.ell - 1)));
    hipLaunchKernelGGL(moddown_rescale_finish_kernel, g, dim3(256), 0, s, t, sh, out, accQ, conv, minv);
}

void launch_modup_conv(const DeviceTables& t, const KsShape& sh_in, u64* ext, const u64* cc, const u64* c_ntt, const u64* hatinv,
                       const u64* hatmod, hipStream_t s) {
    KsShape sh = sh_in;
    const unsigned nx = (1u << t.log_n) / 256;
    sh.tch = conv_tch(nx * (unsigned)(sh.batch * sh.beta), sh.ell + sh.k);
    dim3 g(nx, (unsigned)(sh.batch * sh.beta), (unsigned)((sh.ell + sh.k + sh.tch - 1) / sh.tch));
#define FHELIN_MODUP_CASE(A) case A: hipLaunchKernelGGL((modup_conv_kernel<A>), g, dim3(256), 0, s, t, sh, ext, cc, c_ntt, hatinv, hatmod); break;
    switch (sh.alpha) {
        FHELIN_MODUP_CASE(1) FHELIN_MODUP_CASE(2) FHELIN_MODUP_CASE(3) FHELIN_MODUP_CASE(4)
        FHELIN_MODUP_CASE(5) FHELIN_MODUP_CASE(6) FHELIN_MODUP_CASE(7) FHELIN_MODUP_CASE(8)
        default: hipLaunchKernelGGL((modup_conv_kernel<16>), g, dim3(256), 0, s, t, sh, ext, cc, c_ntt, hatinv, hatmod); break;
    }
#undef FHELIN_MODUP_CASE
}
void launch_ks_inner(const DeviceTables& t, const KsShape& sh, u64* accQ, u64* accP, const u64* ext, const u64* evk, const u64* c_ntt,
                     hipStream_t s) {
    dim3 g((1u << t.log_n) / 512, (unsigned)(sh.batch * (sh.ell + sh.k)));
    hipLaunchKernelGGL(ks_inner_kernel, g, dim3(256), 0, s, t, sh, accQ, accP, ext, evk, c_ntt);
}
void launch_ks_inner_multi(const DeviceTables& t, const KsShape& sh, u64* accQ, u64* accP, const u64* ext, const u64* c_ntt,
                           hipStream_t s) {
    static const int pair_rows = [] { const char* e = std::getenv("FHELIN_KS_ROW_PAIRS"); return e ? std::atoi(e) : 1; }();
    const bool stage = sh.lds_digits && sh.rot_ext_stride == 0 && sh.beta <= 4;
    if (pair_rows && sh.batch >= 2) {       // two rows per thread: the rows' key words are loaded once
        dim3 g((1u << t.log_n) / 512, (unsigned)(((sh.batch + 1) / 2) * (sh.ell + sh.k)));
        if (stage)
            hipLaunchKernelGGL((ks_inner_multi_kernel<true, 2>), g, dim3(256), 0, s, t, sh, accQ, accP, ext, c_ntt);
        else
            hipLaunchKernelGGL((ks_inner_multi_kernel<false, 2>), g, dim3(256), 0, s, t, sh, accQ, accP, ext, c_ntt);
        return;
    }
    dim3 g((1u << t.log_n) / 512, (unsigned)(sh.batch * (sh.ell + sh.k)));
    if (stage)
        hipLaunchKernelGGL((ks_inner_multi_kernel<true, 1>), g, dim3(256), 0, s, t, sh, accQ, accP, ext, c_ntt);
    else
        hipLaunchKernelGGL((ks_inner_multi_kernel<false, 1>), g, dim3(256), 0, s, t, sh, accQ, accP, ext, c_ntt);
}
void launch_gather_sum(const DeviceTables& t, const KsShape& sh, u64* out, const u64* in, size_t in_stride, hipStream_t s) {
    dim3 g((1u << t.log_n) / 256, (unsigned)(sh.batch * sh.ell));
    hipLaunchKernelGGL(gather_sum_kernel, g, dim3(256), 0, s, t, sh, out, in, in_stride);
}
void launch_moddown_conv(const DeviceTables& t, const KsShape& sh_in, u64* conv, const u64* accP, const u64* phatinv, const u64* phatmod,
                         hipStream_t s) {
    KsShape sh = sh_in;
    const unsigned nx = (1u << t.log_n) / 256;
    sh.tch = conv_tch(nx * (unsigned)(2 * sh.batch), sh.ell);
    dim3 g(nx, (unsigned)(2 * sh.batch), (unsigned)((sh.ell + sh.tch - 1) / sh.tch));
#define FHELIN_MODDOWN_CASE(K) case K: hipLaunchKernelGGL((moddown_conv_kernel<K>), g, dim3(256), 0, s, t, sh, conv, accP, phatinv, phatmod); break;
    switch (sh.k) {
        FHELIN_MODDOWN_CASE(1) FHELIN_MODDOWN_CASE(2) FHELIN_MODDOWN_CASE(3) FHELIN_MODDOWN_CASE(4)
        FHELIN_MODDOWN_CASE(5) FHELIN_MODDOWN_CASE(6) FHELIN_MODDOWN_CASE(7) FHELIN_MODDOWN_CASE(8)
        default: hipLaunchKernelGGL((moddown_conv_kernel<16>), g, dim3(256), 0, s, t, sh, conv, accP, phatinv, phatmod); break;
    }
#undef FHELIN_MODDOWN_CASE
}
void launch_moddown_finish(const DeviceTables& t, const KsShape& sh, u64* out, const u64* accQ, const u64* conv, const u64* pinv,
                           const u64* add0, const u64* add1, const u32* map, const u64* post, hipStream_t s) {
    dim3 g((1u << t.log_n) / 512, (unsigned)(sh.batch * 2 * sh.ell));
    hipLaunchKernelGGL(moddown_finish_kernel, g, dim3(256), 0, s, t, sh, out, accQ, conv, pinv, add0, add1, map, post);
}

}  // namespace fhelin
