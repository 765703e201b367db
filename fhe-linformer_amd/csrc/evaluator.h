// RNS-CKKS evaluator on top of the gfx950 kernels: ciphertext / plaintext / key objects and the leveled
// operations the reference reaches through OpenFHE's CryptoContext (reference src/FHEController.cpp
// :409-436 add/mult/rotate, FLEXIBLEAUTO level + scale bookkeeping implied by :18-24).
#pragma once
#include <map>
#include <memory>
#include <vector>
#include "context.h"
#include "kernels_elem.h"

namespace fhelin {

// One device allocation shared by the ciphertexts of a batch (rows of a matmul processed together).
struct DevBlock {
    Context* ctx = nullptr;
    u64* d = nullptr;
    ~DevBlock();
};

// Device-resident ciphertext: d[npoly][ell][N], NTT (evaluation) form.
struct Ciphertext {
    Context* ctx = nullptr;
    u64* d = nullptr;
    std::shared_ptr<DevBlock> block;  // set when d is a view into a batch allocation (then d is not owned)
    int npoly = 2;
    int ell = 0;          // live Q limbs; level (OpenFHE GetLevel) = L+1-ell
    int deg = 1;          // noiseScaleDeg: 1 after rescale / fresh, 2 after a multiplication
    long double scale = 0;
    int slots = 0;
    // set when the ciphertext was produced by a heavy op running asynchronously on a worker lane (capi_internal.h):
    // its data is valid on other streams only after async_ev
    bool async_pending = false;
    int async_lane = 0;
    u64 async_seq = 0;
    hipEvent_t async_ev = nullptr;
    ~Ciphertext();
    int level() const { return ctx->L + 1 - ell; }
    size_t words() const { return (size_t)npoly * ell * ctx->N; }
};
typedef std::shared_ptr<Ciphertext> CtPtr;

// One device encoding of a plaintext vector at a given (limb count, scale).
struct Encoding {
    Context* ctx = nullptr;
    u64* d = nullptr;  // [ell][N] NTT form
    int ell = 0;
    long double scale = 0;
    // made under lane `made_lane` (stream order covers its later uses there); another lane's stream waits for `ready` before its first
    // use (Plaintext::at) - a device-side wait, the host never blocks
    hipEvent_t ready = nullptr;
    int made_lane = 0;
    unsigned lanes_ordered = 0;   // bit k: lane k's stream is already ordered behind `ready`
    ~Encoding();
};

// Plaintext handle: keeps the slot values and lazily (re-)encodes them at whatever (level, scale) an
// operation needs — the effect OpenFHE gets by adjusting plaintext levels inside EvalMult/EvalAdd.
struct Plaintext {
    Context* ctx = nullptr;
    std::vector<double> values;  // real slot values, size == slots
    std::vector<double> imag;    // optional imaginary parts (bootstrapping's DFT diagonals); empty = real vector
    int slots = 0;
    int level = 0;               // level requested at encode time (reference encode(vec, level, slots))
    static constexpr size_t MAX_ENCODINGS = 16;
    std::vector<std::shared_ptr<Encoding>> cache;   // most recently used first (Plaintext::at)
    std::shared_ptr<Encoding> at(int ell, long double scale);
};
typedef std::shared_ptr<Plaintext> PtPtr;

struct EvalKey {
    Context* ctx = nullptr;
    u64* d = nullptr;  // [dnum][2][L+1+k][N] NTT form; component 0 = b, 1 = a
    // the same key with every limb vector gathered through the key's own automorphism map (d_perm[v][n] = d[v][map_g[n]]):
    // the merged rotation sums apply sigma_g to the inner product d * evk BEFORE the shared ModDown, so with this copy the
    // key streams of ks_inner_multi are contiguous and only the digits are gathered.  Built on first use (Evaluator::permuted).
    mutable u64* d_perm = nullptr;
    int digits = 0;
    ~EvalKey();
    size_t words() const { return (size_t)digits * 2 * (ctx->L + 1 + ctx->K) * ctx->N; }
};
typedef std::shared_ptr<EvalKey> KeyPtr;

// round(v) (half away from zero, 80-bit v) modulo the first ell limbs, with Shoup companions (polyeval.cpp)
void real_to_scalars(const Context& c, long double v, int ell, ScalarSet& sc);

class Evaluator {
public:
    explicit Evaluator(Context& c) : c_(c) {}
    Context& ctx() { return c_; }

    // ---- allocation / import / export
    CtPtr new_ct(int npoly, int ell, int deg, long double scale, int slots);
    std::vector<CtPtr> new_ct_batch(int count, int npoly, int ell, int deg, long double scale, int slots);
    // base pointer of `v` if its members are consecutive views of one block (else nullptr)
    static u64* contiguous_base(const std::vector<CtPtr>& v);
    std::vector<CtPtr> make_contiguous(const std::vector<CtPtr>& v, int site = -1);
    u64 gather_copies[8] = {};   // ciphertext copies made by make_contiguous, per call site (diagnostics)
    bool dot_groups = true;    // inner sums of all giant steps of a linear stage in one pass (dot_plain_groups); FHELIN_DOT_GROUPS=0: one pass each
    bool cheb_leaf_at_product = true;   // r-leaves of the Paterson-Stockmeyer tree born at their product's (limbs, scale); FHELIN_CHEB_LEAF_AT=0: level-adjusted afterwards
    bool cheb_leaf_classes = true;   // a Chebyshev leaf's babies aligned to the deepest power IT uses (one level saved: OpenFHE's depth); FHELIN_CHEB_LEAF_CLASSES=0: all babies at one level
    bool cheb_rounds = true;   // Paterson-Stockmeyer products in rounds (polyeval.cpp cheb_recurse); FHELIN_CHEB_ROUNDS=0: one at a time
    int batch_limit = 32;   // rows processed per batched key switch (FHELIN_BATCH overrides; 16 / 24 / 32 / 48 / 64 re-measured at the end of round 2: DESIGN.md)
    CtPtr clone(const CtPtr& a);
    KeyPtr new_key();

    // ---- keys
    std::map<u64, KeyPtr> rot_keys;  // by galois element
    KeyPtr relin_key;
    KeyPtr conj_key;

    // ---- K6-K8 composite: out[2][ell][N] = KeySwitch(c) (+ add0/add1, gathered through map if given)
    void keyswitch(const u64* c_ntt, int ell, const EvalKey& key, u64* out, const u64* add0, const u64* add1, const u32* map,
                   const u64* post = nullptr);
    // the same for `batch` independent polynomials laid out with the given element strides
    void keyswitch_batch(int batch, const u64* c_ntt, size_t c_stride, int ell, const EvalKey& key, u64* out, size_t out_stride,
                         const u64* add0, const u64* add1, size_t add_stride, const u32* map, const u64* post, size_t post_stride);
    // rows with their own evaluation key / automorphism map; shared_input: all rows key-switch the SAME polynomial
    // (c_stride ignored, ModUp done once).  At most KsShape::MAX_ROWS rows.
    struct KsRows {
        std::vector<const EvalKey*> keys;
        std::vector<const u32*> maps;
        bool shared_input = false;
    };
    void keyswitch_rows(const KsRows& rows, const u64* c_ntt, size_t c_stride, int ell, u64* out, size_t out_stride, const u64* add0,
                        size_t add_stride);
    // v[i] + sum_r rot(v[i], indices[r]) with ONE ModUp and ONE ModDown per row: the rotated terms are accumulated in the
    // extended basis (kernels_elem.h launch_ks_inner_multi).  Two steps of a rotate-and-sum tree, x += rot(x, s);
    // x += rot(x, 2s), are the call {s, 2s, 3s}.  All keys must exist (have_rotation_keys).
    std::vector<CtPtr> rotate_sum_batch(const std::vector<CtPtr>& v, const std::vector<int>& indices);
    bool have_rotation_keys(const std::vector<int>& indices, int slots) const;
    const u64* permuted(const EvalKey& key, const u32* map);   // key.d_perm, built once
    // sum_i rot(v[i], indices[i]) for up to 7 ciphertexts of identical shape (index 0 = unrotated term allowed): every
    // term has its own ModUp and key, the inner products are accumulated in QP and share ONE ModDown (giant steps)
    CtPtr rotate_each_sum(const std::vector<CtPtr>& v, const std::vector<int>& indices);
    // the same for many independent rows that share one index list: out[b] = sum_r rot(rows[b][r], indices[r]), with the key
    // switches of a chunk of rows batched (one ModUp over rows x terms, one inner-product launch, one ModDown over rows).
    // Same residues as rotate_each_sum row by row (<= 7 rotated terms per row).
    std::vector<CtPtr> rotate_each_sum_rows(const std::vector<std::vector<CtPtr>>& rows, const std::vector<int>& indices);
    // double hoisting: out[b] = xs[b] * pts[0] + sum_r rot(xs[b], indices[r]) * pts[r + 1]  (== sum_r rot(xs[b] * rot(pts[r + 1], -indices[r]),
    // indices[r]) in slot values) with ONE ModUp and ONE ModDown per row: the rotations share the row's digits and the plaintext
    // products are taken in the extended basis, through rotation keys with the plaintext folded in (folded_key; kernels_elem.h
    // launch_fold_key).  Meant for FEW plaintexts shared by MANY rows (the re-arranged weights of matmulRElarge): a folded key is a
    // full key copy.  Output: noise degree + 1, scale x the level's plaintext scale; 1 <= R <= 7 rotations.
    // rescale_out: the result is wanted rescaled - ModDown and rescale run as ONE basis conversion (drop P and the top limb together:
    // kernels_elem.h launch_moddown_rescale_conv); output one limb and one noise degree lower, scale / q_top.  A different integer
    // function from hoisted_dot_rows + rescale (one rounding instead of two), same value up to rounding noise.
    std::vector<CtPtr> hoisted_dot_rows(const std::vector<CtPtr>& xs, const std::vector<PtPtr>& pts, const std::vector<int>& indices,
                                        bool rescale_out = false);
    // the power steps of a Chebyshev evaluation and EvalMod's double angle through mult_affine_rescale_batch.  OFF by default
    // (FHELIN_MERGED_PRODUCTS=1 turns it on): measured at no gain (282.0 vs 281.2 ms per pass: the launches it saves are replaced by its
    // own) and at 3 x the logit error (0.017 vs 0.006: the merged conversion's rounding has three times the standard deviation of a
    // centred rescale, and these chains are where rounding noise is amplified) - DESIGN.md 6c
    bool merged_products = false;
    bool merged_rescale = true; // results that are rescaled right after their key switch drop P and the top limb in one conversion; FHELIN_MERGED_RESCALE=0: ModDown, then rescale
    bool double_hoist = true;   // Composite::matmulRElarge takes its first step through hoisted_dot_rows; FHELIN_DOUBLE_HOIST=0: rotate_each_sum_rows
    struct FoldedKey {
        PtPtr pt;
        KeyPtr key;
        int index = 0;
        long double scale = 0;
        std::shared_ptr<Encoding> enc;   // the plaintext over the full key basis [L+1+k][N]
        std::shared_ptr<DevBlock> d;     // [digits][2][L+1+k][N], layout of EvalKey::d_perm
    };
    std::vector<FoldedKey> folded_keys;  // bounded (MAX_FOLDED), oldest first
    static constexpr size_t MAX_FOLDED = 12;
    FoldedKey folded_key(const PtPtr& p, int index, long double scale);   // by value: the cache may evict the entry on a later call
    // hoisted rotations: rot(a, i) for every i in `indices` with ONE ModUp of a (results identical to rotate(a, i))
    std::vector<CtPtr> rotate_many(const CtPtr& a, const std::vector<int>& indices);
    // the same for several ciphertexts of one shape and ONE index list (the baby steps of a batch of bootstraps): one ModUp over
    // all inputs, per input the inner products of all indices, one ModDown over all inputs x indices.  out[i][k] holds exactly
    // the residues of rotate(xs[i], indices[k]).
    std::vector<std::vector<CtPtr>> rotate_many_batch(const std::vector<CtPtr>& xs, const std::vector<int>& indices);
    // rows of identical shape through one batched key switch with an explicit Galois element and key (conjugation, the SubSum
    // rotations by multiples of the slot count); accumulate: v_i + sigma_g(v_i)
    std::vector<CtPtr> rotate_galois_batch(const std::vector<CtPtr>& v, u64 galois, const EvalKey& key, bool accumulate);
    std::vector<CtPtr> conjugate_batch(const std::vector<CtPtr>& v);
    std::vector<CtPtr> raw_modraise_batch(const std::vector<CtPtr>& v, int new_ell);   // raw_modraise over many one-limb ciphertexts
    // rot(v[i], indices[i]) for ciphertexts of identical shape, one batched key switch per chunk of rows
    std::vector<CtPtr> rotate_each(const std::vector<CtPtr>& v, const std::vector<int>& indices);
    // batched leveled ops over independent ciphertexts of identical (level, degree, scale): one launch set per op
    std::vector<CtPtr> rotate_add_batch(const std::vector<CtPtr>& v, int index);   // v_i + rot(v_i, index)
    std::vector<CtPtr> rotate_batch(const std::vector<CtPtr>& v, int index);
    std::vector<CtPtr> rotate_batch_impl(const std::vector<CtPtr>& v, int index, bool accumulate);
    std::vector<CtPtr> mult_plain_batch(const std::vector<CtPtr>& v, const PtPtr& p);
    // element-wise ops over many independent ciphertexts, one launch per 32 operands of identical shape; the
    // bookkeeping (rescale before a product, level/scale matching before a sum) is exactly that of the single ops
    std::vector<CtPtr> mult_plain_each(const std::vector<CtPtr>& v, const std::vector<PtPtr>& p);  // v[i] * p[i]
    // sum_i v[i] * p[i]: the residues of add(...add(mult_plain(v0,p0), mult_plain(v1,p1))...) in one pass per 32 terms when the
    // operands share one shape after the usual pre-rescale (else that chain itself)
    // out_g = sum_b cts[b] * pts[g][b] (null = term absent) for all g in ONE pass over the ciphertexts, written to dest[g]
    // (deg+1, scale x plaintext scale); false if the operands do not fit the kernel (caller falls back to dot_plain per g)
    bool dot_plain_groups(const std::vector<CtPtr>& cts, const std::vector<std::vector<PtPtr>>& pts, long double pt_scale,
                          const std::vector<CtPtr>& dest);
    // the 32 cyclic sums out_k = sum_i cts[i] * pts[(i + k) mod 32] over <= 32 ciphertexts of one shape (degree 1) in one pass
    // (kernels_elem.h launch_ew_cyclic_dot), written to dest[k]; the residues of dot_plain per k.  false: operands do not fit
    bool dot_plain_cyclic(const std::vector<CtPtr>& cts, const std::vector<PtPtr>& pts, const std::vector<CtPtr>& dest);
    // one 32 x 32 block of a sliding-window sum (kernels_elem.h launch_ew_window_dot):
    //     dest[t] (+)= sum_{j<32} (j <= t ? cur[j] : prev[j]) * pts[(t - j) mod 32]
    // cur / prev: 32 entries each, null = zero; all present ciphertexts of one shape, degree 1; accumulate: dest holds earlier blocks' sums
    bool dot_plain_window(const std::vector<CtPtr>& cur, const std::vector<CtPtr>& prev, const std::vector<PtPtr>& pts,
                          const std::vector<CtPtr>& dest, bool accumulate);
    // dot_plain_groups for SEVERAL sets of ciphertexts and ONE set of plaintexts in one launch (a batch of bootstraps: the stage's
    // diagonals are fetched once for the batch): out[x][g] = sum_b cts[x][b] * pts[g][b].  Needs every cts[.][b] and every dest[.][g]
    // equally spaced over x (views of one block); false otherwise (caller: dot_plain_groups per set).  Same residues.
    bool dot_plain_groups_batch(const std::vector<std::vector<CtPtr>>& cts, const std::vector<std::vector<PtPtr>>& pts, long double pt_scale,
                                const std::vector<std::vector<CtPtr>>& dest);
    CtPtr dot_plain(const std::vector<CtPtr>& v, const std::vector<PtPtr>& p, long double pt_scale = 0,
                    const CtPtr& dest = CtPtr());   // pt_scale > 0: encode the plaintexts at this scale instead of the level's own;
                                                    // dest: write the sum there (a slice of a caller's batch block)
    std::vector<CtPtr> add_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b);        // a[i] + b[i]
    std::vector<CtPtr> add_plain_batch(const std::vector<CtPtr>& v, const PtPtr& p);
    std::vector<CtPtr> sub_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b);        // a[i] - b[i]
    // a[i] * b[i] with relinearisation: one batched key switch (shared relin key) per chunk of rows
    std::vector<CtPtr> mult_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b);
    // rescale(f * a[i] * b[i] + cadd - sub[i]) (f = 1 or 2; sub empty = none): relinearised products that are rescaled right away - the
    // power steps T_2k = 2 T_k^2 - 1, T_(j+k) = 2 T_j T_k - T_(j-k) of a Chebyshev evaluation, EvalMod's double angle.  Operands as in
    // mult_batch; the constant and the subtrahend (adjusted to the product's limbs / degree 2 / scale) enter the key switch's
    // accumulator times P, and ModDown and rescale run as ONE basis conversion: 2 ell fewer transforms and five fewer launches per
    // product than mult_batch, add_batch, add_real / sub_batch, rescale_batch.  One rounding where those have two (oracle:
    // orc_mult_affine_rescale).  Output: degree 1, one limb fewer, scale a.scale * b.scale / q_top.
    std::vector<CtPtr> mult_affine_rescale_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, int f, double cadd,
                                                 const std::vector<CtPtr>& sub);
    // the same Chebyshev series on several ciphertexts at once: every product of the evaluation runs as mult_batch over
    // all ciphertexts AND all independent nodes of the same depth (baby powers of one doubling round)
    std::vector<CtPtr> eval_chebyshev_many(const std::vector<CtPtr>& xs, const std::vector<double>& coeffs, double a, double b);
    std::vector<CtPtr> rescale_batch(const std::vector<CtPtr>& v);

    // ---- leveled ops (functional: inputs are never modified)
    CtPtr add(const CtPtr& a, const CtPtr& b);
    CtPtr sub(const CtPtr& a, const CtPtr& b);
    CtPtr negate(const CtPtr& a);
    CtPtr add_plain(const CtPtr& a, const PtPtr& p);
    CtPtr mult_plain(const CtPtr& a, const PtPtr& p);
    CtPtr mult(const CtPtr& a, const CtPtr& b);             // tensor + relinearise (auto-rescale inputs of deg 2)
    CtPtr mult_no_relin(const CtPtr& a, const CtPtr& b);    // 3-component result
    CtPtr relinearize(const CtPtr& a);
    CtPtr mult_int(const CtPtr& a, u64 k, bool raise_deg, long double new_scale, int keep_ell = 0);
    std::vector<CtPtr> adjust_deg1_batch(const std::vector<CtPtr>& v, int ell, long double scale);  // by an integer constant
    CtPtr mult_real(const CtPtr& a, double c);              // by a real constant: per-limb scalar round(c * Delta_level)
    CtPtr add_real(const CtPtr& a, double c);               // add a real constant to every slot
    // sum_k coef[k] * terms[k] + c0 for ciphertexts of identical (level, degree 1, scale): the residues of the chain
    // add(mult_real(t_1, c_1), mult_real(t_2, c_2), ...) + add_real(c0), in one kernel pass (falls back to that chain when
    // the shapes differ).  Terms with coef 0 are skipped.
    CtPtr lincomb(const std::vector<CtPtr>& terms, const std::vector<double>& coef, double c0);
    CtPtr lincomb_at(const std::vector<CtPtr>& terms, const std::vector<double>& coef, double c0, long double want_scale, int keep_ell);
    CtPtr rotate(const CtPtr& a, int index);
    CtPtr conjugate(const CtPtr& a);
    // polynomial evaluation (EvalPoly :1291, EvalMultMany :1297, EvalChebyshevFunction :1319-1335)
    CtPtr eval_poly(const CtPtr& x, const std::vector<double>& coeffs);                 // power basis
    CtPtr eval_chebyshev(const CtPtr& x, const std::vector<double>& coeffs, double a, double b);  // sum' c_k T_k(u), u = affine map of [a,b] to [-1,1]
    CtPtr mult_many(const std::vector<CtPtr>& v);
    // the same on several inputs / operand lists at once: every multiplication round is ONE batched relinearisation over all of them
    std::vector<CtPtr> eval_poly_many(const std::vector<CtPtr>& xs, const std::vector<double>& coeffs);
    std::vector<CtPtr> mult_many_rows(const std::vector<std::vector<CtPtr>>& vs);
    CtPtr rescale(const CtPtr& a);                           // drop one limb, divide the scale by it
    CtPtr level_reduce(const CtPtr& a, int new_ell);         // drop limbs without scaling
    // bring `a` to (ell, deg) with scale `scale` following the FLEXIBLEAUTO rules
    CtPtr adjust(const CtPtr& a, int ell, int deg, long double scale);

    // raw, no bookkeeping (parity tests): exactly the residue functions of the oracle
    CtPtr raw_rescale(const CtPtr& a);
    void lift_and_ntt(u64* lifted, const u64* last, int P, int ell);
    CtPtr raw_rotate(const CtPtr& a, u64 galois, const EvalKey& key, bool accumulate = false);
    CtPtr rotate_add(const CtPtr& a, int index);            // a + rot(a, index), one fused key switch (rotsum step :833)
    CtPtr raw_mult_relin(const CtPtr& a, const CtPtr& b, const EvalKey& key);
    // K9 ModRaise (EvalBootstrap's first step): a ciphertext with ONE limb (modulus q0) -> new_ell limbs, every coefficient's
    // centred representative read modulo each q_t.  Scale / degree / slots are copied; the caller sets what they mean.
    CtPtr raw_modraise(const CtPtr& a, int new_ell);

private:
    Context& c_;
    void keyswitch_impl(int batch, const KsRows* rows, const u64* c_ntt, size_t c_stride, int ell, const EvalKey* key, u64* out,
                        size_t out_stride, const u64* add0, const u64* add1, size_t add_stride, const u32* map, const u64* post,
                        size_t post_stride);
    typedef std::vector<CtPtr> CtRow;  // one value per input ciphertext
    CtRow cheb_recurse(const std::vector<double>& c, const std::vector<CtRow>& T, const std::map<int, CtRow>& G, int baby);
    std::vector<CtPtr> add_sub_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, int op);
    void match(const CtPtr& a, const CtPtr& b, CtPtr& ao, CtPtr& bo);
    void match_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, std::vector<CtPtr>& x, std::vector<CtPtr>& y);
};

}  // namespace fhelin
