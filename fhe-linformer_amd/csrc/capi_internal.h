// Shared plumbing for the extern "C" translation units.
#pragma once
#include <exception>
#include <string>
#include "context.h"
#include "evaluator.h"
#include "client.h"
#include "composite.h"
#include "bootstrap.h"

// opaque handle behind include/fhelin.h's `fhelin_ctx`
struct fhelin_ctx {
    fhelin::Context ctx;
    fhelin::Evaluator ev;
    fhelin::Client cl;
    fhelin::Composite comp;
    fhelin::Bootstrapper boot;
    bool lazy_rows = true;      // FHELIN_LAZY_ROWS
    explicit fhelin_ctx(const fhelin::Params& p);
};
// Rows of a batched composite whose evaluation is DEFERRED until a row is consumed (fhelin_fc_matmul_pt,
// fhelin_fc_unwrapExpanded): the reference's drivers compute whole row sets of which later code reads a single row
// (src/main.cpp:183,:196 — all S query projections, only Q[0] used; :416-424 — all S tokens expanded, only output_2[0]
// used).  A row is evaluated the first time a handle to it is read — together with every other
// deferred row of the same call that the consuming operation also takes — and never if nobody reads it.  Rows are
// independent, so a forced row holds exactly the residues eager evaluation gives (bit-exact tests run with deferral
// on).  FHELIN_LAZY_ROWS=0 evaluates eagerly.
namespace fhelin {
struct LazyRows {
    enum Kind { MatmulPt, UnwrapExpanded } kind = MatmulPt;
    CtVec rows;                 // MatmulPt: the input rows (kept alive)
    PtPtr w, bias;
    int slots = 0, padding = 0;
    CtPtr src;                  // UnwrapExpanded: the wrapped ciphertext
    int n = 0;
    std::vector<CtPtr> done;    // per row: null until evaluated
    int partial_reads = 0;      // reads so far that asked for only some of the rows (force_group)
};
}
struct fhelin_ct {
    mutable fhelin::CtPtr p;                          // null while the row is still deferred
    mutable std::shared_ptr<fhelin::LazyRows> lazy;
    int lazy_idx = 0;
    fhelin_ctx* owner = nullptr;
};
struct fhelin_pt {
    fhelin::PtPtr p;
};

namespace fhelin {
int capi_fail(int code, const std::string& msg);
// evaluate the listed rows of a deferred group in ONE batched call (capi_composite.cpp)
void force_rows(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx);
// A consumer asks for rows `idx` of a deferred group.  The first partial read evaluates just those rows (a driver that
// uses Q[0] only); a second one means the driver is walking over the rows (for (i...) output[i] = add(output[i],
// inputs[i]), src/main.cpp:237-239): everything that is left is evaluated in one batched call instead of row by row.
void force_group(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx);
inline void force(fhelin_ctx* c, const fhelin_ct* h) {
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
inline const CtPtr& ct_in(fhelin_ctx* c, const fhelin_ct* h) {
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

#define FHELIN_TRY try {
#define FHELIN_CATCH                                                          \
    return FHELIN_OK;                                                         \
    }                                                                         \
    catch (const fhelin::Error& e) { return fhelin::capi_fail(e.code, e.what()); }          \
    catch (const std::bad_alloc&) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, "host out of memory"); } \
    catch (const std::exception& e) { return fhelin::capi_fail(FHELIN_ERR_INTERNAL, e.what()); }
