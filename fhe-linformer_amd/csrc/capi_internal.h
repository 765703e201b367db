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
    explicit fhelin_ctx(const fhelin::Params& p) : ctx(p), ev(ctx), cl(ev, ctx.prm.seed_bytes), comp(ev, cl), boot(ev, cl) {}
};
struct fhelin_ct {
    fhelin::CtPtr p;
};
struct fhelin_pt {
    fhelin::PtPtr p;
};

namespace fhelin {
int capi_fail(int code, const std::string& msg);

// The main stream is about to consume `h`: if a worker lane is still producing it, order the main stream behind the
// producing op (no host wait) and drop the holds that op needed.
inline const CtPtr& ct_in(fhelin_ctx* c, const fhelin_ct* h) {
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
