// Launchers of the element-wise, rescale and key-switch kernels (kernels_elem.hip, kernels_ks.hip).
#pragma once
#include "kernels.h"

namespace fhelin {

// per-limb Shoup scalars passed by value in the kernarg segment: v[2*i] = s mod q_i, v[2*i+1] = shoup
struct ScalarSet {
    u64 v[128];
};

// out[v] = a[v] (op) b[v % b_mod] for v < nvec; vector v uses limb limb_first + v % limb_count
void launch_ew_mul(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s);
void launch_ew_add(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s);
void launch_ew_sub(const DeviceTables& t, u64* out, const u64* a, const u64* b, int nvec, int b_mod, int limb_first, int limb_count, hipStream_t s);
void launch_ew_muladd(const DeviceTables& t, u64* out, const u64* acc, const u64* a, const u64* b, int nvec, int b_mod, int limb_first,
                      int limb_count, hipStream_t s);
void launch_ew_neg(const DeviceTables& t, u64* out, const u64* a, int nvec, int limb_first, int limb_count, hipStream_t s);
void launch_ew_scalar(const DeviceTables& t, u64* out, const u64* a, const ScalarSet& sc, int nvec, int limb_first, int limb_count, hipStream_t s,
                      int in_limbs = 0);   // in_limbs > 0: inputs have in_limbs limbs per polynomial, the first limb_count are used
void launch_ew_addscalar(const DeviceTables& t, u64* out, const u64* a, const ScalarSet& sc, int nvec, int limb_first, int limb_count, hipStream_t s,
                         int add_vecs = -1);   // add_vecs >= 0: the constant goes to the first add_vecs vectors, the rest are copied
// out[v] = sum_k a_k[v] * scal[k][limb] + scal[n][limb]  (limb = v % ell): a linear combination with real constants of up to
// MAX_TERMS ciphertexts of identical shape in ONE pass — the base case of a Chebyshev / power-basis evaluation, which as
// single ops is one scalar-multiply launch and one add launch per term.  scal: device array [(n + 1)][ell] of residues.
struct LinComb {
    static constexpr int MAX_TERMS = 32;
    int n = 0;
    int vecs = 0;  // limb vectors per operand (npoly * ell)
    const u64* a[MAX_TERMS];
};
void launch_ew_lincomb(const DeviceTables& t, u64* out, const LinComb& lc, const u64* scal, int ell, hipStream_t s, int in_limbs = 0);
void launch_tensor(const DeviceTables& t, u64* d, const u64* a, const u64* b, int ell, hipStream_t s);
struct EwItems;
// it.out[k] [3][ell][N] = tensor(it.a[k], it.b[k]) for the it.n pairs of one batched multiplication, in ONE launch
void launch_tensor_items(const DeviceTables& t, const EwItems& it, int ell, hipStream_t s);
void launch_automorph(const DeviceTables& t, u64* out, const u64* in, const u32* map, int nvec, hipStream_t s);
// the same gather, every word stored as pack30(word) (low 30 bits | next 30 bits << 32): operands of launch_ks_inner_multi
void launch_automorph_pack30(const DeviceTables& t, u64* out, const u64* in, const u32* map, int nvec, hipStream_t s);
void launch_rescale_lift(const DeviceTables& t, u64* lifted, const u64* last, int npoly, int ell, const u64* qlmod_row, hipStream_t s);
void launch_rescale_finish(const DeviceTables& t, u64* out, const u64* c, const u64* lifted, int npoly, int ell, const u64* qlinv_row,
                           hipStream_t s);
void launch_modraise(const DeviceTables& t, u64* out, const u64* src, int npoly, int src_limb, int nl, hipStream_t s);
void launch_reduce_i128(const DeviceTables& t, u64* out, const u64* coeffs, int limb_first, int nlimbs, hipStream_t s);

// ---- hybrid key switching (K6-K8)
// Element-wise op over up to MAX_ITEMS independent operands of identical shape in ONE launch (grid.y = items x vectors):
//   out_i[v] = a_i[v] (op) b_i[v % b_vecs]     op 0 mul, 1 add, 2 sub;   op 3: add for v < b_vecs, copy otherwise
// (op 3 = ciphertext + plaintext: only component 0 changes); op 4: out_i = a_i (scattered ciphertexts gathered into one block by ONE
// launch instead of a copy per ciphertext, Evaluator::make_contiguous).  Pointers travel in the kernel arguments.
struct EwItems {
    static constexpr int MAX_ITEMS = 32;
    int n = 0;
    int vecs = 0;
    int b_vecs = 0;
    u64* out[MAX_ITEMS];
    const u64* a[MAX_ITEMS];
    const u64* b[MAX_ITEMS];
};
void launch_ew_items(const DeviceTables& t, const EwItems& it, int op, int limb_count, hipStream_t s);
// out[v] = sum_i a_i[v] * b_i[v % b_vecs] over the n items (n <= MAX_ITEMS): an inner product of ciphertexts with plaintexts in
// one pass (the diagonal sums of the bootstrapping linear transforms, wrapUpRepeated, matmulCRlarge) instead of n product
// launches and a tree of additions.  128-bit accumulation, one reduction: the canonical residue of the sum.
void launch_ew_dot(const DeviceTables& t, u64* out, const EwItems& it, int limb_count, hipStream_t s);
// Several such inner products over the SAME ciphertexts in one pass: out_g = sum_b a_b * p_{g,b} for g < ng (the inner sums of
// all giant steps of a baby-step/giant-step linear transform: every rotated ciphertext is read once instead of once per
// giant step).  a_b: [2][ell][N]; p_{g,b}: [ell][N] or nullptr (term absent); out_g: [2][ell][N].  Same 128-bit accumulation and
// single reduction per output as launch_ew_dot: identical residues.
struct EwDotGroups {
    static constexpr int MAX_A = 16, MAX_G = 8;
    int na = 0, ng = 0, ell = 0;
    const u64* a[MAX_A];
    const u64* p[MAX_G][MAX_A];
    u64* out[MAX_G];
    // nbatch > 1: the same plaintexts applied to nbatch sets of ciphertexts in ONE launch (the inner sums of a bootstrapping stage
    // over a batch of bootstraps): set x reads a[b] + x * a_stride[b] and writes out[g] + x * out_stride[g].  The sets of one
    // (limb, tile) run back to back on one XCD: every plaintext tile comes from HBM once for the whole batch.
    int nbatch = 1;
    size_t a_stride[MAX_A] = {};
    size_t out_stride[MAX_G] = {};
    u32 pmask[MAX_G] = {};       // filled by launch_ew_dot_groups: bit b of pmask[g] = term (g, b) present
};
void launch_ew_dot_groups(const DeviceTables& t, const EwDotGroups& d, hipStream_t s);
// The 32 CYCLIC plaintext-weighted sums over n <= 32 ciphertexts of one shape (Composite::relarge_container):
//     out_k = sum_{i<n} a_i * m_{(i + k) mod 32},   k < 32        a_i, out_k: [2][ell][N];  m_j: [ell][N]
// in ONE pass: a workgroup keeps the 32 plaintext values of its 256 coefficients in LDS (a dynamically indexed register file: every
// thread reads its own column) and the n ciphertext values of both components in registers, pre-split in 30-bit halves; every
// operand is read from memory once.  Exact sums (128-bit accumulation, folded every 16 products): the canonical residues of the sum.
struct EwCyclic {
    static constexpr int PERIOD = 32;
    int n = 0, ell = 0;
    const u64* a[PERIOD];
    const u64* m[PERIOD];
    u64* out[PERIOD];
};
void launch_ew_cyclic_dot(const DeviceTables& t, const EwCyclic& d, hipStream_t s);
// One 32 x 32 block of a plaintext-weighted SLIDING-WINDOW sum  x_i = sum_k m_k * D_{i-k}  (Composite::unwrapExpanded_bulk):
//     out_t (+)= sum_{j<32} (j <= t ? cur_j : prev_j) * m_{(t - j) mod 32},   t < 32
// cur_j = D_{p+j}, prev_j = D_{p-32+j} for the block's window position p: the pairs (i, k) = (32 g + t, 32 c + k') of one output
// block g and one tap chunk c.  Same structure as launch_ew_cyclic_dot (plaintext values in LDS, ciphertext values in registers,
// both pre-split); an entry whose mask bit is clear counts as zero; accumulate: out_t already holds the sum of earlier tap chunks.
struct EwWindow {
    static constexpr int W = 32;
    int ell = 0, accumulate = 0;
    u32 cur_mask = 0, prev_mask = 0;   // bit j: cur[j] / prev[j] is present; an absent entry still points at readable memory of the operands' shape
    const u64* cur[W];
    const u64* prev[W];
    const u64* m[W];
    u64* out[W];
};
void launch_ew_window_dot(const DeviceTables& t, const EwWindow& d, hipStream_t s);

struct KsShape {
    int ell;     // live Q limbs
    int k;       // special limbs
    int alpha;   // limbs per digit
    int beta;    // digits at this level
    int L1;      // total Q limbs (evk limb stride is L1 + k)
    // batching over independent ciphertexts (rows of a matmul): element b of the batch lives at base + b*stride
    int batch = 1;
    size_t c_stride = 0;     // input polynomial c (NTT form)
    size_t out_stride = 0;   // output [2][ell][N]
    size_t add_stride = 0;   // add0 / add1
    size_t post_stride = 0;  // post-add operand
    // Rows that rotate by different amounts: per-row evaluation keys and automorphism maps (per_row = 1), and rows
    // that all extend the SAME input polynomial (shared_input = 1: hoisted rotations — ModUp runs once, the digits
    // are reused by every row's inner product).
    static constexpr int MAX_ROWS = 16;
    int per_row = 0;
    int shared_input = 0;
    // row_mod = R > 0 (with per_row): batch row bi is rotation bi % R (key evk_row[bi % R], map map_row[bi % R]) of INPUT bi / R - the
    // digits at ext + (bi / R) * ext_batch_stride, the polynomial at c + (bi / R) * c_stride, the K8b addend at add + (bi / R) *
    // add_stride: the hoisted rotations of MANY inputs in one launch (Evaluator::rotate_many_batch).  In the XCD-aware block order
    // the rows of one (tile, limb) run back to back on one XCD: every key tile is fetched once for all inputs, every digit tile
    // once for all rotations.
    int row_mod = 0;
    const u64* evk_row[MAX_ROWS] = {};
    const u32* map_row[MAX_ROWS] = {};
    // merged rotations (launch_ks_inner_multi): n_rot <= MAX_ROT rotations of every row are accumulated before ONE ModDown
    static constexpr int MAX_ROT = 7;
    int n_rot = 0;
    const u64* evk_rot[MAX_ROT] = {};   // the keys in PERMUTED, PRE-SPLIT layout (EvalKey::d_perm): evk_rot[r][v][n] = pack30(key_r[v][map_rot[r][n]])
    const u32* map_rot[MAX_ROT] = {};
    // rot_input_stride > 0: rotation r acts on its OWN input (digits at ext + r * rot_ext_stride, polynomial at c + r *
    // rot_input_stride): a sum of rotations of different ciphertexts (giant steps) shares the one ModDown
    size_t rot_ext_stride = 0;
    size_t rot_input_stride = 0;
    int lds_digits = 0;           // launch_ks_inner_multi: every map_rot keeps each 512-coefficient tile in place (see the kernel)
    size_t ext_batch_stride = 0;  // digits of batch row b at ext + b * ext_batch_stride (0: beta * (ell + k) * N, one input per row)
    // K8b epilogue of a merged rotation sum: component 0 additionally receives sum_r gsrc[bi][tt][map_rot[r][n]] (the rotated
    // c0 parts, gathered in place of a separate gather-and-sum pass); rotation r reads gsrc + r * rot_input_stride
    const u64* gsrc = nullptr;
    size_t gsrc_stride = 0;   // per batch row
    // basis-conversion kernels: targets per block (set by the launchers).  Every chunk of targets re-reads the conversion's source limbs,
    // so a launch that fills the GPU anyway (many rows) takes ALL targets in one block - sources read once - and only small launches
    // are cut into chunks of 16 for parallelism
    int tch = 16;
};
// K6: cc [ell][N] coefficient form, c_ntt [ell][N] NTT form -> ext [beta][ell+k][N]
//     (own-digit slots stay unused — K7 reads c_ntt there; the others get the fast-basis-extended values, coefficient form)
void launch_modup_conv(const DeviceTables& t, const KsShape& sh, u64* ext, const u64* cc, const u64* c_ntt, const u64* hatinv,
                       const u64* hatmod, hipStream_t s);
// K7: accQ [2][ell][N], accP [2][k][N] <- sum_j ext[j][t] * evk[j][comp][limb(t)]
void launch_ks_inner(const DeviceTables& t, const KsShape& sh, u64* accQ, u64* accP, const u64* ext, const u64* evk, const u64* c_ntt,
                     hipStream_t s);
// K7 for a sum of rotations: acc_c[t][n] = sum_r sum_j d_j[t][m_r(n)] * evk_{r,j,c}[t][m_r(n)],  m_r = map_rot[r]: the
// automorphism of every term is applied while gathering (the maps move whole 128-byte lines), all terms are
// accumulated in 128 bits and reduced once; ONE ModDown then serves the whole sum.
void launch_ks_inner_multi(const DeviceTables& t, const KsShape& sh, u64* accQ, u64* accP, const u64* ext, const u64* c_ntt,
                           hipStream_t s);
// ---- double hoisting with plaintext-folded keys: sum_r rot(x, i_r) * V_r with ONE ModUp and ONE ModDown, the plaintext products taken
// in the extended basis.  sigma_r(d * evk_r) * V_r = sigma_r(d) * (sigma_r(evk_r) * V_r): with the folded key
//     fold_r[v][n] = pack30( V_r[limb(v)][n] * key_r[v][map_r[n]] mod m_limb(v) ),   limb(v) = v % (L1 + k),
// (V_r encoded over the FULL key basis q_0..q_L, p_0..p_{k-1}: [L1 + k][N], NTT form) launch_ks_inner_multi computes the whole sum
// as it stands (evk_rot[r] = fold_r).  The folded key is a canonical key in the layout of EvalKey::d_perm.
void launch_fold_key(const DeviceTables& t, u64* out, const u64* key, const u32* map, const u64* V, int nvec, hipStream_t s);
// what does not pass through the key switch: pre[b][0][t][n] = V_0[t][n] c0_b[t][n] + sum_r V_{r+1}[t][n] c0_b[t][map_r[n]],
// pre[b][1][t][n] = V_0[t][n] c1_b[t][n]   (ct: batch row b at ct + b * sh.c_stride, [2][ell][N]; pre dense [batch][2][ell][N];
// v[i]: [>= ell][N]).  128-bit sums, one reduction.
// acc != nullptr: instead of storing pre, the kernel adds P * pre to the Q part of the key switch's accumulator (acc [batch][2][ell][N]
// += pre * pmod[t]; pmod [L+1][2] = P mod q_t, shoup): the sum then passes through the ModDown with everything else, which is what a
// ModDown merged with the rescale needs (launch_moddown_rescale_conv) - pre itself is not written.
struct HoistAdd {
    int n_rot = 0;
    const u64* v[KsShape::MAX_ROT + 1] = {};
    const u32* map[KsShape::MAX_ROT] = {};
    u64* acc = nullptr;
    const u64* pmod = nullptr;
};
void launch_hoist_addends(const DeviceTables& t, const KsShape& sh, const HoistAdd& h, u64* pre, const u64* ct, hipStream_t s);
// out[v][n] = sum_r in[v][map_rot[r][n]]  for v in [0, nvec) (the c0 parts of the rotated copies), per batch row
// (FHELIN_FUSE_GATHER=0 only: by default the sum rides in launch_moddown_finish, KsShape::gsrc)
void launch_gather_sum(const DeviceTables& t, const KsShape& sh, u64* out, const u64* in, size_t in_stride, hipStream_t s);
// K8a: accP coefficient form [2][k][N] -> conv [2][ell][N] (coefficient form)
void launch_moddown_conv(const DeviceTables& t, const KsShape& sh, u64* conv, const u64* accP, const u64* phatinv, const u64* phatmod,
                         hipStream_t s);
// ModDown and rescale as ONE basis conversion: with B = (p_0..p_{k-1}, q_{ell-1}), M = P q_{ell-1} and X the accumulator over
// (q_0..q_{ell-1}, p_0..p_{k-1}):      out_t = (X_t - sum_{b in B} [X_b (M/b)^{-1}]_b [(M/b)]_t) M^{-1}  mod q_t,   t < ell - 1
// with the sources y_b = [X_b (M/b)^{-1}]_b taken CENTRED (y_b - b above b/2; mmod[t] = M mod q_t comes off once per such source): the
// conversion then misses round(X / M) by an integer of mean zero and |u| <= (k+1)/2 - without centring every coefficient would carry
// a common offset of ~ -(k+1)/2 at the result's own scale, which the slots next to the root of unity 1 see N/pi-fold (K8's own
// non-centred error is divided away by the rescale that follows it; here nothing follows).  A key switch whose result is rescaled right away then costs 2(k+1) inverse and 2(ell-1) forward transforms
// where ModDown + rescale cost 2k + 2 inverse and 2 ell + 2(ell-1) forward ones.  Addends that do not pass through the key switch must
// be in X already (multiplied by P: HoistAdd::acc).
//   conv:   accP [batch][2][k][N] and top [batch][2][N] (= X mod q_{ell-1}) in coefficient form -> conv [batch][2][ell-1][N] (coefficient form)
//   finish: out [batch][2][ell-1][N] = (accQ[.][.][t] - NTT(conv)) * minv_t      (accQ [batch][2][ell][N]; sh.out_stride per batch row)
// A relinearised product that is rescaled right away (Evaluator::mult_affine_rescale_batch): what does not pass through the key switch
// enters the accumulator's Q part times P, so that ModDown and rescale can be one conversion:
//     accQ[b][c][t] = f * accQ[b][c][t] + pmod_t * (f * d[b][c][t] + (c == 0 ? cst_t : 0) - sub[b][c][t])        f = 1 or 2
// d: the tensor block [batch][3][ell][N] (components 0, 1 are read); sub: [batch][2][ell][N] or null; cst: per-limb constants
// (ScalarSet::v[2 t]) or has_cst = 0.  The special limbs of the accumulator are scaled by f separately (launch_ew_add).
void launch_affine_acc(const DeviceTables& t, const KsShape& sh, u64* accQ, const u64* d, const u64* sub, const ScalarSet& cst, int has_cst, int f,
                       const u64* pmod, hipStream_t s);
void launch_moddown_rescale_conv(const DeviceTables& t, const KsShape& sh, u64* conv, const u64* accP, const u64* top, const u64* hatinv,
                                 const u64* hatmod, const u64* mmod, hipStream_t s);
void launch_moddown_rescale_finish(const DeviceTables& t, const KsShape& sh, u64* out, const u64* accQ, const u64* conv, const u64* minv,
                                   hipStream_t s);
// K8b: out[c][t][j] = (accQ[c][t][m] - conv[c][t][m]) * P^{-1} + add_c[t][m] (+ post[c][t][j]),  m = map ? map[j] : j
void launch_moddown_finish(const DeviceTables& t, const KsShape& sh, u64* out, const u64* accQ, const u64* conv, const u64* pinv,
                           const u64* add0, const u64* add1, const u32* map, const u64* post, hipStream_t s);

}  // namespace fhelin
