// Launch interface of the hand-written gfx950 kernels (implementation: kernels_*.hip).
// Host code (evaluator.cpp, capi.cpp) sees only these plain-C++ launchers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime_api.h>
#include "modarith.h"

namespace fhelin {

// A batch of independent length-N residue vectors ("limb vectors"), contiguous: data[v][N].
// Vector v belongs to RNS limb  limb_tab ? limb_tab[v] : limb_first + (v % limb_count);
// a negative table entry skips the vector.  Limb ids index the context-wide arrays
// (moduli, twiddles): Q limbs 0..L, then special limbs L+1..L+k.
struct LimbBatch {
    u64* data;
    int nvec;
    const int* limb_tab;  // device pointer or nullptr
    int limb_first;
    int limb_count;
    const u64* src = nullptr;  // optional: read the input from src[v][N] (out of place), results land in data
    int tab_len = 0;           // > 0: the table describes one batch element and repeats: limb_tab[v % tab_len]
    // strided out-of-place input: vector v is read from src + (v / src_group) * src_group_stride + (v % src_group) * N
    int src_group = 0;         // 0 = src is dense like data
    size_t src_group_stride = 0;
    // second level: groups come in rows of src_group2 groups; row r starts at src + r * src_group2_stride and its groups follow
    // src_group_stride apart (the rotated terms of a batch of rows that each skip their unrotated term)
    int src_group2 = 0;
    size_t src_group2_stride = 0;
    // forward only: limbs on the lazy butterfly path (q < 2^53) may leave the transform unreduced, in [0, 86q) < 2^60,
    // when the consumer takes any value below 2^60 (the evaluation-key inner product K7 splits its operands in 30-bit
    // halves and reduces the 128-bit sums once).  Saves the final reduction of the row pass.
    bool lazy_out = false;
    // forward only, rescale (K5): the input of vector v is the CENTRED LIFT of the single coefficient-form limb
    // src[v / limb_count][N] (modulus q_lift_limb) into limb v's modulus, formed in the first pass's load instead of by a
    // separate kernel: r = x mod q_j, minus (q_lift mod q_j) where x > q_lift / 2.  lift_qlm[j] = q_lift mod q_j.
    const u64* lift_qlm = nullptr;
    int lift_limb = -1;
};

// Device-resident per-context tables.
struct DeviceTables {
    int log_n;
    int n_limbs;           // L+1+k
    const u64* moduli;     // [n_limbs]
    const u64* barrett;    // [n_limbs][2]  (r0, r1) of floor(2^128/q)
    const u64* qinv;       // [n_limbs]  q^-1 mod 2^64 (modarith.h redc128)
    const u64* mont;       // [n_limbs][2]  2^64 mod q and its Shoup companion: what a key copy made for redc128 sums is multiplied by
    const u64* tw_fwd;     // [n_limbs][2N]  (w, w') pairs, bit-reversed powers of psi
    const u64* tw_inv;     // [n_limbs][2N]  same for psi^{-1}
    // the per-thread twiddles of the row pass's last (forward) / first (inverse) four stages, transposed so that one load
    // instruction of a wave reads 64 consecutive 16-byte pairs: [n_limbs][N/4096 tiles][15 slots][256 threads][2]
    const u64* tw_rows_fwd;
    const u64* tw_rows_inv;
    const u64* ninv;       // [n_limbs][8]   N^{-1}, shoup, ipsi_br[1]*N^{-1}, shoup, lazy shift, lazy ratio, 0, 0
};

// Optional epilogue of the FORWARD transform's row pass: instead of storing NTT(conv), finish the ModDown of hybrid key
// switching in registers (K8b, see kernels_elem.h launch_moddown_finish):
//   out[bi][c][t][j] = (accQ[v][m] - NTT(conv)[v][m]) * P^-1 + add_c[bi][t][m] (+ post[bi][c][t][j]),   j = invmap[m]
// for vector v = (bi*2 + c)*ell + t.  invmap is the INVERSE automorphism map (position m lands at j), so a wave still
// writes whole 128-byte lines: the map permutes lines and the 16 residues inside a line.
struct NttModDown {
    static constexpr int MAX_ROWS = 16;
    const u64* accQ = nullptr;
    u64* out = nullptr;
    const u64* add0 = nullptr;
    const u64* add1 = nullptr;
    const u64* post = nullptr;
    const u32* invmap = nullptr;
    const u64* pinv = nullptr;   // [L+1][2] P^-1 mod q_t, shoup
    int ell = 0;
    size_t out_stride = 0, add_stride = 0, post_stride = 0;
    int per_row = 0;
    const u32* invmap_row[MAX_ROWS] = {};
};

// K1: negacyclic NTT (natural -> bit-reversed) / INTT (bit-reversed -> natural, scaled by N^{-1}).
// In place, canonical [0,q) in and out.
void launch_ntt(const DeviceTables& t, const LimbBatch& b, bool inverse, hipStream_t s);
// forward NTT of b (conv, [batch][2][ell][N]) whose row pass ends in the ModDown epilogue; b.data is scratch afterwards
void launch_ntt_moddown(const DeviceTables& t, const LimbBatch& b, const NttModDown& md, hipStream_t s);


}  // namespace fhelin
