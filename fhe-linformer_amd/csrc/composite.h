// Composite circuit operations of the reference's FHEController, re-expressed on the GPU evaluator:
// rotate-and-sum reductions, the rotation-based 128x128 matmuls, slot masks, wrap/unwrap layout shuffles.
// Each method cites the reference function it mirrors (all in reference src/FHEController.cpp).
// These are the CALLERS of the hot path; they fix the kernel mix.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "client.h"
#include "evaluator.h"

namespace fhelin {

typedef std::vector<CtPtr> CtVec;

class Composite {
public:
    Composite(Evaluator& ev, Client& cl);
    int num_slots() const { return 1 << ev_.ctx().prm.log_slots; }

    // leaf helpers
    PtPtr encode_vec(const std::vector<double>& v, int level);
    PtPtr encode_const(double val, int level);                          // encode(double,...)        :358-371
    CtPtr mult_const(const CtPtr& c, double d);                         // mult(ct,double)           :421-424

    // masks (plaintexts cached per pattern; the Plaintext re-encodes per level lazily)      :1207-1286
    CtPtr mask_block(const CtPtr& c, int from, int to, double v);
    CtPtr mask_heads(const CtPtr& c, double v);
    CtPtr mask_heads_128(const CtPtr& c, double v);
    PtPtr heads_128_mask(double v);
    CtPtr mask_mod_n(const CtPtr& c, int n, int padding);
    CtPtr mask_first_n(const CtPtr& c, int n, double v);
    PtPtr first_n_mask(int n, double v);   // the cached mask plaintexts themselves (batched callers)
    PtPtr block_mask(int from, int to, double v);
    PtPtr mod_n_mask(int n, int padding);
    PtPtr mod_range_mask(int period, int from, int to, int span_from = 0, int span_to = -1);   // slot mod period in [from, to), slot in [span_from, span_to)

    // log-tree reductions                                                                     :829-867
    CtPtr rotsum(const CtPtr& in, int slots, int padding);
    CtPtr rotsum_padded(const CtPtr& in, int slots);
    CtPtr repeat(const CtPtr& in, int slots, int padding);              // repeat(in,slots) == padding 1
    CtPtr add_many(const CtVec& v);                                     // EvalAddMany               :417-419
    // the same reductions over independent rows, step i for ALL rows before step i+1 (one batched key switch per
    // step: the evaluation key is read once per batch and every launch fills the GPU) — results are bit-identical
    CtVec rotsum_batch(const CtVec& in, int slots, int padding);
    CtVec repeat_batch(const CtVec& in, int slots, int padding);
    CtVec tree_batch(const CtVec& in, int slots, int step_sign, int padding);
    CtVec tree_steps(CtVec r, int n, int unit);   // n doubling steps x += rot(x, unit * 2^i), merged in pairs when the 3s key exists

    // matmuls                                                                                 :869-1058
    CtVec matmul_pt(const CtVec& rows, const PtPtr& w, const PtPtr& bias, int slots, int padding);   // RE / CR with plaintext weight
    CtVec matmul_ct(const CtVec& rows, const CtPtr& w, int slots, int padding);                      // RE / CR with ciphertext weight
    CtVec matmulRElarge(const CtVec& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias, double mask_val);
    CtVec matmulCRlarge(const std::vector<CtVec>& rows, const std::vector<PtPtr>& weights, const PtPtr& bias);
    CtPtr matmulScores(const CtVec& queries, const CtPtr& key);

    // ---- the same calls on SEVERAL SAMPLES at once (BASELINE config 4's per-GPU unit: a batch of independent inputs through one
    // engine, src/main.cpp:145-475 once per sample).  Group x of every argument belongs to sample x; all groups of a call have one
    // size.  The single-sample methods above are these with one group, so a sample's results hold the same residues either way
    // (rows / groups of a batched key switch are independent): what changes is that the small launches of a single pass - one
    // query row, one container tail, one wrapped ciphertext - now carry every sample's rows.
    CtVec matmul_ct_each(const CtVec& rows, const CtVec& ws, int slots, int padding);                     // row i against weight ws[i]
    CtVec matmulScores_multi(const std::vector<CtVec>& queries, const CtVec& keys);
    CtVec shift_sum_multi(const std::vector<CtVec>& groups, int step);
    std::vector<CtVec> shift_fan_rows_multi(const CtVec& cs, int n, int step, const std::vector<int>& idx);
    CtVec wrapUpExpanded_multi(const std::vector<CtVec>& groups);
    std::vector<CtVec> unwrapExpanded_rows_multi(const CtVec& cs, int inputs_num, const std::vector<int>& idx);
    std::vector<CtVec> unwrapExpanded_bulk_multi(const CtVec& cs, int n, const std::vector<int>& idx);
    std::vector<std::vector<CtVec>> unwrapRepeatedLarge_multi(const std::vector<CtVec>& containers, int input_number, int first = 0, int count = -1);
    CtVec wrap_containers_multi(const std::vector<CtVec>& cs, int inputs_number);
    std::vector<CtVec> generate_containers_multi(const std::vector<CtVec>& inputs, const PtPtr& bias);
    std::vector<CtVec> relarge_containers_multi(const std::vector<CtVec>& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias,
                                                double mask_val, const PtPtr& cbias);

    // shifted sums / fans shared by the layout shuffles (log-depth forms of the reference's rotate-by-one chains)
    CtPtr shift_sum(const CtVec& terms, int step);          // sum_i rot(terms[i], step * i)
    CtVec shift_fan(const CtPtr& c, int n, int step);       // rot(c, step * i), i = 0..n-1
    CtVec shift_fan_rows(const CtPtr& c, int n, int step, const std::vector<int>& idx);   // the listed rows only

    // layout shuffles                                                                         :1060-1205
    CtPtr wrapUpRepeated(const CtVec& v);
    CtPtr wrapUpExpanded(const CtVec& v);
    CtVec unwrapExpanded(CtPtr c, int inputs_num);
    // the rows `idx` of unwrapExpanded(c, inputs_num) only, each with the residues the full call gives (the rotation
    // chain of row i is that of shift_fan: the set bits of i in ascending order)
    CtVec unwrapExpanded_rows(CtPtr c, int inputs_num, const std::vector<int>& idx);
    CtVec unwrapScoresExpanded(CtPtr c, int inputs_num);
    CtVec unwrap_512_in_4_128(const CtPtr& c, int index);
    // tokens [first, first + count) only (count < 0: all)
    std::vector<CtVec> unwrapRepeatedLarge(const CtVec& containers, int input_number, int first = 0, int count = -1);
    CtVec generate_containers(const CtVec& inputs, const PtPtr& bias);
    // generate_containers(matmulRElarge(inputs, weights, bias, mask_val), cbias) in one go, for rows of matmulRElarge nobody has read
    // yet (the C ABI defers them: capi_composite.cpp): the 5-step tree of every row and the container sum collapse into ONE shift sum
    // per group of 32 rows (relarge_container).  Same slot values, a different integer function (oracle/residue_controller.py
    // restates it); groups of fewer than RELARGE_FUSE_MIN rows and contexts without the shared form take the two calls as they stand.
    CtVec relarge_containers(const CtVec& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias, double mask_val, const PtPtr& cbias);
    static constexpr int RELARGE_FUSE_MIN = 8;
    bool fuse_relarge = true;   // FHELIN_FUSE_RELARGE=0: never
    // unwrapExpanded for MANY rows of one call at once (>= UNWRAP_BULK_MIN rows read together): row i = repeat(mask_0 * rot(c, i), 128, 1)
    // = sum_{k<128} mask_k * rot(c, i - k)  (mask_k: slots = k mod 128): the rotations rot(c, j), -127 <= j < n, are hoisted fans shared
    // by all rows, and every row is a plaintext-weighted sliding-window sum of them (Evaluator::dot_plain_window) - no key switch per row
    // where the tree form runs three.  Same slot values, a different integer function (oracle/residue_eval.py unwrapExpanded_bulk).
    CtVec unwrapExpanded_bulk(CtPtr c, int n, const std::vector<int>& idx);
    static constexpr int UNWRAP_BULK_MIN = 64;
    bool bulk_unwrap = true;    // FHELIN_BULK_UNWRAP=0: the tree form for every read
    CtPtr wrap_containers(const CtVec& c, int inputs_number);

private:
    Evaluator& ev_;
    Client& cl_;
    std::map<std::string, PtPtr> mask_cache_;
    // matmulRElarge: the four 128x128 weight blocks re-arranged block-wise (W''_t, t = 0..3), cached per weight set
    std::map<std::string, std::vector<PtPtr>> relarge_cache_;
    std::map<std::string, std::vector<std::vector<double>>> relarge_src_;   // the slot values an entry was made from: a hash hit is compared with them
    std::vector<PtPtr> relarge_weights(const std::vector<PtPtr>& weights, bool rotated);
    bool relarge_shared(const CtVec& inputs, const std::vector<PtPtr>& weights);
    CtVec relarge_u(const CtVec& inputs, const std::vector<PtPtr>& weights);        // the shared form's first step: U per row
    CtVec relarge_tail(const CtVec& u, const PtPtr& bias, double mask_val);          // its 5-step tree, mask and bias
    CtPtr relarge_container(const CtVec& u, const PtPtr& bias, double mask_val);     // tree + container sum of one group of rows
    CtVec relarge_w(const CtVec& u, double mask_val);                                // its 32 cyclic plaintext-weighted sums
    PtPtr relarge_tiled_bias(const PtPtr& bias, int q);
    bool early_rescale_ = true;   // FHELIN_EARLY_RESCALE: rescale a fresh product before its rotation tree
    bool merge_rot_ = true;       // FHELIN_MERGE_ROT: two tree steps as one merged key switch when the 3s key exists
    bool row_lanes_ = false;      // FHELIN_ROW_LANES: row chunks of a tree on separate streams (off: measured slower)
    PtPtr mask_plain(const std::string& key, const std::vector<double>& v);
};

}  // namespace fhelin
