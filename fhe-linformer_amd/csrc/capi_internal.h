// Shared plumbing for the extern "C" translation units.
#pragma once
#include <exception>
#include <string>
#include "context.h"
#include "evaluator.h"
#include "client.h"
#include "composite.h"
#include "bootstrap.h"

namespace fhelin {
// Profile-guided level planning (include/fhelin.h fhelin_level_plan_*).
// The drivers are straight-line programs: which ciphertext meets which, and how many limbs each operation consumes, does
// not depend on the data.  A RECORDING pass notes, for every handle the boundary gives out, the handles the producing call
// read and the result's effective limb count (limbs minus a pending rescale).  From that the planner derives, back to
// front, the fewest limbs each value may have without any terminal (a bootstrap input, a decryption, an export) running
// short, and so by how many limbs every SOURCE - a fresh encryption, a bootstrap output - can start lower.  An APPLYING pass
// of the same program then encrypts at that level and raises bootstrap inputs to fewer limbs, so that limbs nobody would
// ever use are not carried through the key switches in between.  Sources are identified by their order in the pass.
struct LevelPlan {
    int mode = 0;  // 0 off, 1 record, 2 apply
    struct Node {
        std::vector<int> in;
        int eff = -1;       // limbs - (deg >= 2): -1 while a deferred row is unevaluated
        int ordinal = -1;   // >= 0: a source
        int need = -1;      // fewest effective limbs any consumer chain requires (-1: reaches no terminal)
    };
    std::vector<Node> nodes;
    int epoch = 0;              // recordings are numbered: a handle's node id only means something in the recording that made it
    int next_ordinal = 0;
    std::vector<int> target;    // the plan, per source ordinal: effective limbs the source should start with (-1: as asked)

    static int eff_of(const Ciphertext& c) { return c.ell - (c.deg >= 2 ? 1 : 0); }
    int add_node(const std::vector<int>& in, int eff) {
        nodes.push_back(Node{in, eff, -1, -1});
        return (int)nodes.size() - 1;
    }
    void terminal(int node, int need) {
        if (mode == 1 && node >= 0) nodes[node].need = std::max(nodes[node].need, need);
    }
    // applying: a terminal that is short of limbs means the pass is not the recorded program
    void check_terminal(const Ciphertext& c, int need) const {
        if (mode == 2 && eff_of(c) < need)
            throw Error(FHELIN_ERR_STATE, "level plan: a value reached a decryption / export with fewer limbs than it needs - this pass does not follow the recorded program");
    }
    // the next source of the pass would start with `ell` limbs: how many of them to leave out (0 unless applying)
    int next_drop(int ell) {
        const int k = next_ordinal++;
        if (mode != 2 || k >= (int)target.size() || target[k] < 1) return 0;
        return std::max(0, ell - target[k]);
    }
    void begin(int m) {
        mode = m;
        next_ordinal = 0;
        nodes.clear();
        ++epoch;
    }
    bool live(int node, int node_epoch) const { return mode == 1 && node >= 0 && node_epoch == epoch && node < (int)nodes.size(); }
    void finish();  // record mode: derive `target` from the recording (capi.cpp)
};
// handles read by the C-ABI call in progress on this thread (node ids of the recording context)
std::vector<int>& plan_inputs();
}  // namespace fhelin

// Heavy single-ciphertext operations (bootstrap, Chebyshev evaluation) whose evaluation is DEFERRED until the result is read:
// the reference's drivers issue them in loops over independent ciphertexts (the GELU containers, src/main.cpp:354-358: eval_gelu
// then bootstrap per container; the two halves of affine-1, :313-314) and read the results only after the loop.  When the first
// result is read, everything pending is evaluated in dependency order with the operations of one kind, one parameter set and
// one input shape BATCHED (Evaluator::eval_chebyshev_many, Bootstrapper::bootstrap_batch): every launch then carries all of
// them and the switching keys / plaintext diagonals are read once.  Each result holds exactly the residues the single call gives.
// EvalAdd(ct, ct) is deferred the same way: the drivers' residual additions are loops of 130 single calls (src/main.cpp:237-239,
// output[i] = add(output[i], inputs[i])), each with a FLEXIBLEAUTO level adjustment of its own; read together they are ONE batched
// adjustment (integer multiply + rescale over all rows) and one batched addition.
// FHELIN_LAZY_HEAVY=0 (or a recording level-plan pass) evaluates at the call, on alternating worker lanes (run_heavy).
namespace fhelin {
struct LazyHeavy {
    enum Kind { Cheb, Boot, Add } kind = Boot;
    CtPtr in;                               // the input, once it exists ...
    std::shared_ptr<LazyHeavy> in_heavy;    // ... or the deferred operation that will produce it
    CtPtr in2;                              // Add: the second operand, likewise
    std::shared_ptr<LazyHeavy> in2_heavy;
    std::vector<double> coeffs;             // Cheb
    double a = 0, b = 0;
    int drop = 0;                           // Boot: limbs left out under a level plan
    CtPtr result;
    bool done = false, failed = false;
    int err_code = 0;                       // failed: what the batched call that evaluated it threw (reported when the handle is read)
    std::string err_msg;
    int lane = 0;                           // the lane (stream) the call was made under: it is evaluated there (fhelin_ctx_set_lane)
};
}
// opaque handle behind include/fhelin.h's `fhelin_ctx`
struct fhelin_ctx {
    fhelin::Context ctx;
    fhelin::Evaluator ev;
    fhelin::Client cl;
    fhelin::Composite comp;
    fhelin::Bootstrapper boot;
    bool lazy_rows = true;      // FHELIN_LAZY_ROWS
    bool lazy_heavy = true;     // FHELIN_LAZY_HEAVY
    // deferred operations, per lane: a lane's pending operations are evaluated on that lane's stream (fhelin_ctx_set_lane)
    std::vector<std::shared_ptr<fhelin::LazyHeavy>> pending_heavy[fhelin::DevicePool::MAX_LANES];
    bool any_pending() const {
        for (const auto& v : pending_heavy)
            if (!v.empty()) return true;
        return false;
    }
    int user_lane = 0;          // fhelin_ctx_set_lane: 0 = the context's main stream
    hipEvent_t lane_mark[fhelin::DevicePool::MAX_LANES] = {};   // fhelin_ctx_lane_mark: a point in a lane's stream others can wait for
    bool lane_marked[fhelin::DevicePool::MAX_LANES] = {};
    fhelin::LevelPlan plan;
    explicit fhelin_ctx(const fhelin::Params& p);
};
// Rows of a batched composite whose evaluation is DEFERRED until a row is consumed (fhelin_fc_matmul_pt,
// fhelin_fc_unwrapExpanded): the reference's drivers compute whole row sets of which later code reads a single row
// (src/main.cpp:183,:196 — all S query projections, only Q[0] used; :416-424 — all S tokens expanded, only output_2[0]
// used).  A row is evaluated the first time a handle to it is read — together with every other
// deferred row of the same call that the consuming operation also takes — and never if nobody reads it.  Rows are
// independent, so a forced row holds exactly the residues eager evaluation gives (bit-exact tests run with deferral
// on).  FHELIN_LAZY_ROWS=0 evaluates eagerly.
// fhelin_fc_matmulRElarge defers its rows too, for another reason: the drivers hand ALL of them to generate_containers next
// (src/main.cpp:341-352), and the two calls together are one shift sum per 32 rows (Composite::relarge_containers) where each alone
// is a tree per row and then the container sum.  generate_containers takes the fused form when every input is a row of
// matmulRElarge that nobody has read (same weights, bias and mask); a row somebody reads first is evaluated as matmulRElarge
// itself would.  The fused form is a different integer function with the same slot values: it is part of the default path the
// residue-level oracle restates (oracle/residue_controller.py).
namespace fhelin {
struct LazyRows {
    enum Kind { MatmulPt, UnwrapExpanded, RElarge } kind = MatmulPt;
    CtVec rows;                 // MatmulPt, RElarge: the input rows (kept alive)
    PtPtr w, bias;
    std::vector<PtPtr> weights; // RElarge: the four weight blocks; mask value
    double mask_val = 1.0;
    int slots = 0, padding = 0;
    CtPtr src;                  // UnwrapExpanded: the wrapped ciphertext
    int n = 0;
    std::vector<CtPtr> done;    // per row: null until evaluated
    int partial_reads = 0;      // reads so far that asked for only some of the rows (force_group)
    std::vector<int> node;      // level-plan recording: the rows' node ids ...
    int node_epoch = -1;        // ... of this recording (LevelPlan::epoch)
};
}
struct fhelin_ct {
    mutable fhelin::CtPtr p;                          // null while the row is still deferred
    mutable std::shared_ptr<fhelin::LazyRows> lazy;
    mutable std::shared_ptr<fhelin::LazyHeavy> heavy;   // a deferred bootstrap / Chebyshev evaluation
    int lazy_idx = 0;
    fhelin_ctx* owner = nullptr;
    int node = -1;                                    // level-plan recording: this value's node ...
    int node_epoch = -1;                              // ... in this recording (a handle may outlive the pass that made it)
};
struct fhelin_pt {
    fhelin::PtPtr p;
};

namespace fhelin {
int capi_fail(int code, const std::string& msg);
// evaluate the listed rows of a deferred group in ONE batched call (capi_composite.cpp)
void force_rows(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx);
// the same for several groups at once; groups that are the same call on different inputs (one call per sample of a batch) share
// their batched key switches
void force_rows_multi(fhelin_ctx* c, std::vector<std::pair<LazyRows*, std::vector<int>>>& reqs);
std::vector<int> rows_for_read(LazyRows& g, const std::vector<int>& idx);
// A consumer asks for rows `idx` of a deferred group.  The first partial read evaluates just those rows (a driver that
// uses Q[0] only); a second one means the driver is walking over the rows (for (i...) output[i] = add(output[i],
// inputs[i]), src/main.cpp:237-239): everything that is left is evaluated in one batched call instead of row by row.
void force_group(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx);
// evaluate every pending deferred heavy operation, batched (capi_composite.cpp)
// report: throw the first failure of this flush (fhelin_sync / fhelin_ctx_trim: "evaluate everything pending"); a read of ONE result
// (force) does not - it reports that result's own failure, if any, and leaves the others' to their readers
void flush_heavy(fhelin_ctx* c, bool report = false);          // the CURRENT lane's pending operations
void flush_heavy_all(fhelin_ctx* c, bool report = false);      // every lane's, each on its own stream
// a value produced under lane `lane` is about to be read under the current one: order the current stream behind that lane's work
void wait_for_lane(fhelin_ctx* c, int lane);
fhelin_ct* defer_add(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b);
bool defer_allowed(fhelin_ctx* c);
inline void force(fhelin_ctx* c, const fhelin_ct* h) {
    if (!h->p && h->heavy) {
        if (!h->heavy->done) {
            const int cur = c->ctx.pool.cur_lane, own = h->heavy->lane;
            if (own == cur) {
                flush_heavy(c);
            } else {                 // issued under another lane: evaluated there, then this lane's stream goes behind it
                c->ctx.stream = own == 0 ? c->ctx.main_stream : c->ctx.lane_stream[own];
                c->ctx.pool.cur_lane = own;
                try {
                    flush_heavy(c);
                } catch (...) {
                    c->ctx.stream = cur == 0 ? c->ctx.main_stream : c->ctx.lane_stream[cur];
                    c->ctx.pool.cur_lane = cur;
                    throw;
                }
                c->ctx.stream = cur == 0 ? c->ctx.main_stream : c->ctx.lane_stream[cur];
                c->ctx.pool.cur_lane = cur;
                wait_for_lane(c, own);
            }
        }
        if (h->heavy->failed || !h->heavy->result)
            throw Error(h->heavy->err_code ? h->heavy->err_code : FHELIN_ERR_STATE,
                        "deferred operation failed: " + (h->heavy->err_msg.empty() ? std::string("no result") : h->heavy->err_msg));
        h->p = h->heavy->result;
        h->heavy.reset();
        return;
    }
    if (h->p || !h->lazy) return;
    LazyRows& g = *h->lazy;
    if (!g.done[h->lazy_idx]) force_group(c, g, std::vector<int>{h->lazy_idx});
    h->p = g.done[h->lazy_idx];
    h->lazy.reset();
}
// the deferred rows among v[0..n) are evaluated together, one batched call per group
void force_many(fhelin_ctx* c, const fhelin_ct* const* v, int n);

// The main stream is about to consume `h`: if a worker lane is still producing it, order the main stream behind the
// producing op (no host wait) and drop the holds that op needed.
inline void note_input(fhelin_ctx* c, const fhelin_ct* h) {
    if (c->plan.live(h->node, h->node_epoch)) plan_inputs().push_back(h->node);
}
// a new handle for a result of the call in progress
inline fhelin_ct* wrap(fhelin_ctx* c, const CtPtr& p) {
    auto* h = new fhelin_ct;
    h->p = p;
    if (c->plan.mode == 1) {
        h->node = c->plan.add_node(plan_inputs(), LevelPlan::eff_of(*p));
        h->node_epoch = c->plan.epoch;
    }
    return h;
}
inline const CtPtr& ct_in(fhelin_ctx* c, const fhelin_ct* h) {
    note_input(c, h);
    force(c, h);
    Ciphertext& ct = *h->p;
    if (ct.async_pending) {
        Context& x = c->ctx;
        hip_check(hipStreamWaitEvent(x.main_stream, ct.async_ev, 0), "hipStreamWaitEvent(async result)");
        ct.async_pending = false;
        x.release_holds(ct.async_lane, ct.async_seq);
    }
    return h->p;
}

// Run a heavy single-ciphertext op (bootstrap, polynomial evaluation) on a worker lane without joining it.  Ops issued
// back to back on independent ciphertexts (the reference's per-container loops, src/main.cpp:354-358, :313-314)
// then overlap on the GPU; a chain on the same ciphertext stays on its lane.  Falls back to the main stream when
// lanes are off.
template <class F>
CtPtr run_heavy(fhelin_ctx* c, const fhelin_ct* in, F&& f) {
    Context& x = c->ctx;
    note_input(c, in);
    force(c, in);               // a deferred input is evaluated on the main stream, before the lane is entered
    if (x.n_lanes < 2 || !x.async_lanes || x.stream != x.main_stream) return f(ct_in(c, in));
    Ciphertext& ci = *in->p;
    const int k = ci.async_pending ? ci.async_lane : 1 + (x.async_rr++ % x.n_lanes);
    // the lane starts behind everything the main stream has been given so far: the op's input, and every earlier
    // consumer of blocks that this lane's pool may hand out again
    hip_check(hipEventRecord(x.fork_event, x.main_stream), "hipEventRecord(async fork)");
    hip_check(hipStreamWaitEvent(x.lane_stream[k], x.fork_event, 0), "hipStreamWaitEvent(async fork)");
    CtPtr out;
    {
        Context::LaneScope scope(x, k);
        out = f(in->p);
    }
    const u64 seq = ++x.lane_seq[k];
    if (!out->async_ev) hip_check(hipEventCreateWithFlags(&out->async_ev, hipEventDisableTiming), "hipEventCreate(async)");
    hip_check(hipEventRecord(out->async_ev, x.lane_stream[k]), "hipEventRecord(async)");
    out->async_pending = true;
    out->async_lane = k;
    out->async_seq = seq;
    x.lane_hold[k].emplace_back(seq, std::static_pointer_cast<void>(in->p));
    return out;
}
}

#define FHELIN_TRY try { fhelin::plan_inputs().clear();
#define FHELIN_CATCH                                                          \
    return FHELIN_OK;                                                         \
    }                                                                         \
    catch (const fhelin::Error& e) { return fhelin::capi_fail(e.code, e.what()); }          \
    catch (const std::bad_alloc&) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, "host out of memory"); } \
    catch (const std::exception& e) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, e.what()); }
