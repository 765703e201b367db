// extern "C" boundary, part 3: one entry point per FHEController composite method.
#include "../../include/fhelin.h"
#include <algorithm>
#include <functional>
#include "capi_internal.h"

using namespace fhelin;

#define NEED(x) if (!(x)) return capi_fail(FHELIN_ERR_ARG, "null argument")

static CtVec vec_of(fhelin_ctx* c, const fhelin_ct* const* v, int n) {
    for (int i = 0; i < n; ++i)
        if (!v[i]) throw Error(FHELIN_ERR_ARG, "null ciphertext handle in array");
    force_many(c, v, n);
    CtVec out;
    for (int i = 0; i < n; ++i) out.push_back(ct_in(c, v[i]));
    return out;
}
// handles to the rows of a deferred group
static void emit_lazy(fhelin_ctx* c, const std::shared_ptr<LazyRows>& g, int n, fhelin_ct** outs) {
    g->done.assign(n, CtPtr());
    for (int i = 0; i < n; ++i) {
        auto* h = new fhelin_ct;
        h->lazy = g;
        h->lazy_idx = i;
        h->owner = c;
        if (c->plan.mode == 1) {
            h->node = c->plan.add_node(plan_inputs(), -1);   // level known once the row is evaluated (force_rows)
            h->node_epoch = g->node_epoch = c->plan.epoch;
            g->node.push_back(h->node);
        }
        outs[i] = h;
    }
}

namespace fhelin {
// Evaluate the listed rows of several deferred groups.  Groups that are the SAME call on different inputs (one driver call per
// sample of a batch: same weights / same row list) are evaluated TOGETHER - their rows share the batched key switches - and every
// row still holds the residues its own group's evaluation gives (rows are independent).
void force_rows_multi(fhelin_ctx* c, std::vector<std::pair<LazyRows*, std::vector<int>>>& reqs) {
    struct Todo { LazyRows* g; std::vector<int> rows; };
    std::vector<Todo> todo;
    for (auto& e : reqs) {
        Todo t{e.first, {}};
        for (int i : e.second)
            if (!t.g->done[i] && std::find(t.rows.begin(), t.rows.end(), i) == t.rows.end()) t.rows.push_back(i);
        if (!t.rows.empty()) todo.push_back(t);
    }
    std::vector<char> taken(todo.size(), 0);
    for (size_t first = 0; first < todo.size(); ++first) {
        if (taken[first]) continue;
        const LazyRows& f = *todo[first].g;
        std::vector<size_t> grp;
        for (size_t k = first; k < todo.size(); ++k) {
            if (taken[k]) continue;
            const LazyRows& g = *todo[k].g;
            bool same = g.kind == f.kind;
            if (same && g.kind == LazyRows::MatmulPt) same = g.w == f.w && g.bias == f.bias && g.slots == f.slots && g.padding == f.padding;
            if (same && g.kind == LazyRows::RElarge) same = g.weights == f.weights && g.bias == f.bias && g.mask_val == f.mask_val;
            if (same && g.kind == LazyRows::UnwrapExpanded) same = g.n == f.n && todo[k].rows == todo[first].rows;
            if (same) {
                grp.push_back(k);
                taken[k] = 1;
            }
        }
        std::vector<CtVec> res(grp.size());
        if (f.kind == LazyRows::UnwrapExpanded) {
            CtVec srcs;
            for (size_t k : grp) srcs.push_back(todo[k].g->src);
            res = c->comp.unwrapExpanded_rows_multi(srcs, f.n, todo[first].rows);
        } else {
            CtVec sub;
            for (size_t k : grp)
                for (int i : todo[k].rows) sub.push_back(todo[k].g->rows[i]);
            const CtVec r = f.kind == LazyRows::MatmulPt ? c->comp.matmul_pt(sub, f.w, f.bias, f.slots, f.padding)
                                                          : c->comp.matmulRElarge(sub, f.weights, f.bias, f.mask_val);
            size_t p = 0;
            for (size_t j = 0; j < grp.size(); ++j) {
                const size_t cnt = todo[grp[j]].rows.size();
                res[j].assign(r.begin() + p, r.begin() + p + cnt);
                p += cnt;
            }
        }
        for (size_t j = 0; j < grp.size(); ++j) {
            LazyRows& g = *todo[grp[j]].g;
            const std::vector<int>& rows = todo[grp[j]].rows;
            for (size_t k = 0; k < rows.size(); ++k) {
                g.done[rows[k]] = res[j][k];
                if (!g.node.empty() && c->plan.live(g.node[rows[k]], g.node_epoch)) c->plan.nodes[g.node[rows[k]]].eff = LevelPlan::eff_of(*res[j][k]);
            }
            bool all = true;
            for (const CtPtr& d : g.done) all = all && d;
            if (all) {  // nothing left to evaluate: release the inputs
                g.rows.clear();
                g.src.reset();
            }
        }
    }
}
void force_rows(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx) {
    std::vector<std::pair<LazyRows*, std::vector<int>>> one{{&g, idx}};
    force_rows_multi(c, one);
}
// Deferred heavy operations: everything pending, in dependency order; per round the ready operations of one kind, one
// parameter set and one input shape go through ONE batched call.
// A batched call that throws fails ONLY its own group (and, transitively, the operations that read its results); every other ready
// group is still evaluated.  The failure stays with the operation: reading its handle reports it (force), fhelin_sync reports the
// first one of the flush; reading an unrelated result succeeds.
static void fail_op(LazyHeavy& h, int code, const std::string& msg) {
    h.done = h.failed = true;
    h.err_code = code;
    h.err_msg = msg;
    h.in.reset();
    h.in_heavy.reset();
    h.in2.reset();
    h.in2_heavy.reset();
}
void flush_heavy(fhelin_ctx* c, bool report) {
    std::vector<std::shared_ptr<LazyHeavy>> pend;
    pend.swap(c->pending_heavy[c->ctx.pool.cur_lane]);
    int first_code = 0;
    std::string first_msg;
    auto note_failure = [&](int code, const std::string& msg) {
        if (!first_code) {
            first_code = code;
            first_msg = msg;
        }
    };
    // run one batched call over `grp`; a throw fails exactly these operations
    auto guarded = [&](const std::vector<LazyHeavy*>& grp, const std::function<void()>& run) {
        try {
            run();
        } catch (const Error& e) {
            for (LazyHeavy* g : grp) fail_op(*g, e.code, e.what());
            note_failure(e.code, e.what());
        } catch (const std::exception& e) {
            for (LazyHeavy* g : grp) fail_op(*g, FHELIN_ERR_INTERNAL, e.what());
            note_failure(FHELIN_ERR_INTERNAL, e.what());
        }
    };
    for (;;) {
        std::vector<LazyHeavy*> ready;
        bool waiting = false;
        for (auto& sp : pend) {
            LazyHeavy& h = *sp;
            if (h.done) continue;
            if (sp.use_count() == 1) {   // every handle to the result is gone and nothing pending reads it: never evaluated
                fail_op(h, FHELIN_ERR_STATE, "never evaluated: no handle to the result was left");
                continue;
            }
            bool dep_failed = false;
            std::string why;
            for (auto* dep : {&h.in_heavy, &h.in2_heavy})
                if (*dep && (*dep)->done) {
                    if ((*dep)->failed || !(*dep)->result) {
                        dep_failed = true;
                        why = (*dep)->err_msg;
                    } else {
                        (dep == &h.in_heavy ? h.in : h.in2) = (*dep)->result;
                        dep->reset();
                    }
                }
            if (dep_failed) {
                fail_op(h, FHELIN_ERR_STATE, "a deferred operation this one reads failed earlier: " + why);
                continue;
            }
            if (h.in && (h.kind != LazyHeavy::Add || h.in2)) ready.push_back(&h);
            else waiting = true;
        }
        if (ready.empty()) {
            if (waiting) {   // cannot happen (an operation only ever reads earlier ones); fail what is left rather than spin
                for (auto& sp : pend)
                    if (!sp->done) fail_op(*sp, FHELIN_ERR_INTERNAL, "deferred heavy operations: dependency cycle");
                note_failure(FHELIN_ERR_INTERNAL, "deferred heavy operations: dependency cycle");
            }
            break;
        }
        std::vector<char> taken(ready.size(), 0);
        {   // every ready addition of the round in ONE batched call (Evaluator::add_batch aligns levels per group of rows); additions whose
            // operands cannot be added (component counts differ) fail on their own, before the batch
            CtVec as, bs;
            std::vector<LazyHeavy*> adds;
            for (size_t k = 0; k < ready.size(); ++k)
                if (ready[k]->kind == LazyHeavy::Add) {
                    taken[k] = 1;
                    if (ready[k]->in->npoly != ready[k]->in2->npoly) {
                        fail_op(*ready[k], FHELIN_ERR_STATE, "add: component count mismatch");
                        note_failure(FHELIN_ERR_STATE, "add: component count mismatch");
                        continue;
                    }
                    adds.push_back(ready[k]);
                    as.push_back(ready[k]->in);
                    bs.push_back(ready[k]->in2);
                }
            if (!adds.empty())
                guarded(adds, [&] {
                    CtVec out = adds.size() == 1 ? CtVec{c->ev.add(as[0], bs[0])} : c->ev.add_batch(as, bs);
                    for (size_t k = 0; k < adds.size(); ++k) {
                        adds[k]->result = out[k];
                        adds[k]->done = true;
                        adds[k]->in.reset();
                        adds[k]->in2.reset();
                    }
                });
        }
        for (size_t first = 0; first < ready.size(); ++first) {
            if (taken[first]) continue;
            LazyHeavy& f = *ready[first];
            std::vector<LazyHeavy*> grp;
            for (size_t k = first; k < ready.size(); ++k) {
                LazyHeavy& g = *ready[k];
                const bool same = !taken[k] && g.kind == f.kind && g.in->ell == f.in->ell && g.in->deg == f.in->deg &&
                                  g.in->npoly == f.in->npoly &&
                                  (f.kind == LazyHeavy::Boot ? g.drop == f.drop : (g.a == f.a && g.b == f.b && g.coeffs == f.coeffs));
                if (same) {
                    grp.push_back(&g);
                    taken[k] = 1;
                }
            }
            guarded(grp, [&] {
                CtVec in;
                for (LazyHeavy* g : grp) in.push_back(g->in);
                CtVec out = f.kind == LazyHeavy::Boot ? c->boot.bootstrap_batch(in, f.drop) : c->ev.eval_chebyshev_many(in, f.coeffs, f.a, f.b);
                for (size_t k = 0; k < grp.size(); ++k) {
                    grp[k]->result = out[k];
                    grp[k]->done = true;
                    grp[k]->in.reset();
                }
            });
        }
    }
    if (report && first_code) throw Error(first_code, "a deferred operation failed: " + first_msg);
}
// a handle to a deferred heavy operation on `a` (itself possibly deferred)
static fhelin_ct* defer_heavy(fhelin_ctx* c, const fhelin_ct* a, const std::shared_ptr<LazyHeavy>& op) {
    if (!a->p && a->heavy && !a->heavy->done) {
        op->in_heavy = a->heavy;
    } else {
        op->in = ct_in(c, a);
        // what can be checked now is checked now: the offending call reports it, not a later read
        if (op->in->npoly != 2) throw Error(FHELIN_ERR_STATE, "bootstrap / polynomial evaluation: the ciphertext must have 2 components (relinearise first)");
        if (op->kind == LazyHeavy::Cheb && op->in->ell - (op->in->deg >= 2 ? 1 : 0) < 2)
            throw Error(FHELIN_ERR_STATE, "polynomial evaluation: no limb left for a multiplication");
    }
    op->lane = c->ctx.pool.cur_lane;
    c->pending_heavy[op->lane].push_back(op);
    auto* h = new fhelin_ct;
    h->heavy = op;
    h->owner = c;
    return h;
}
// a handle to the deferred sum a + b (either operand possibly deferred itself); capi_eval.cpp's fhelin_add
fhelin_ct* defer_add(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b) {
    auto op = std::make_shared<LazyHeavy>();
    op->kind = LazyHeavy::Add;
    if (!a->p && a->heavy && !a->heavy->done) op->in_heavy = a->heavy;
    else op->in = ct_in(c, a);
    if (!b->p && b->heavy && !b->heavy->done) op->in2_heavy = b->heavy;
    else op->in2 = ct_in(c, b);
    if (op->in && op->in2 && op->in->npoly != op->in2->npoly) throw Error(FHELIN_ERR_STATE, "add: component count mismatch");
    op->lane = c->ctx.pool.cur_lane;
    c->pending_heavy[op->lane].push_back(op);
    auto* h = new fhelin_ct;
    h->heavy = op;
    h->owner = c;
    return h;
}
// deferral on the context's main stream or on a lane the CALLER selected (fhelin_ctx_set_lane) - not inside the library's own lane scopes
bool defer_allowed(fhelin_ctx* c) {
    return c->lazy_heavy && c->plan.mode != 1 && c->ctx.pool.cur_lane == c->user_lane &&
           c->ctx.stream == (c->user_lane ? c->ctx.lane_stream[c->user_lane] : c->ctx.main_stream);
}
static bool defer_ok(fhelin_ctx* c) { return defer_allowed(c); }
void flush_heavy_all(fhelin_ctx* c, bool report) {
    const int cur = c->ctx.pool.cur_lane;
    hipStream_t cur_stream = c->ctx.stream;
    int code = 0;
    std::string msg;
    for (int k = 0; k < DevicePool::MAX_LANES; ++k) {
        if (c->pending_heavy[k].empty()) continue;
        c->ctx.stream = k == 0 ? c->ctx.main_stream : c->ctx.lane_stream[k];
        c->ctx.pool.cur_lane = k;
        try {
            flush_heavy(c, report);
        } catch (const Error& e) {
            if (!code) {
                code = e.code;
                msg = e.what();
            }
        }
    }
    c->ctx.stream = cur_stream;
    c->ctx.pool.cur_lane = cur;
    if (code) throw Error(code, msg);
}
void wait_for_lane(fhelin_ctx* c, int lane) {
    Context& x = c->ctx;
    if (lane == x.pool.cur_lane) return;
    hipStream_t src = lane == 0 ? x.main_stream : x.lane_stream[lane];
    hip_check(hipEventRecord(x.fork_event, src), "hipEventRecord(lane hand-over)");
    hip_check(hipStreamWaitEvent(x.stream, x.fork_event, 0), "hipStreamWaitEvent(lane hand-over)");
}

void force_many(fhelin_ctx* c, const fhelin_ct* const* v, int n) {
    for (int i = 0; i < n; ++i)
        if (v[i] && !v[i]->p && v[i]->heavy) force(c, v[i]);      // the first one evaluates everything pending
    std::vector<std::pair<LazyRows*, std::vector<int>>> groups;
    for (int i = 0; i < n; ++i) {
        const fhelin_ct* h = v[i];
        if (!h || h->p || !h->lazy) continue;
        LazyRows* g = h->lazy.get();
        auto it = std::find_if(groups.begin(), groups.end(), [&](const auto& e) { return e.first == g; });
        if (it == groups.end()) it = groups.insert(groups.end(), {g, {}});
        if (std::find(it->second.begin(), it->second.end(), h->lazy_idx) == it->second.end()) it->second.push_back(h->lazy_idx);
    }
    // per group the rows its read pattern asks for (force_group's rule), then ONE merged evaluation over the groups
    for (auto& e : groups) e.second = rows_for_read(*e.first, e.second);
    force_rows_multi(c, groups);
}
// A consumer asks for rows `idx` of a deferred group: which rows to evaluate now.  The first partial read evaluates just those rows
// (a driver that uses Q[0] only); a second one means the driver is walking over the rows: everything that is left is evaluated in
// one batched call instead of row by row.
std::vector<int> rows_for_read(LazyRows& g, const std::vector<int>& idx) {
    std::vector<int> rest;
    for (int i = 0; i < (int)g.done.size(); ++i)
        if (!g.done[i]) rest.push_back(i);
    bool wanted = false, partial = false;
    for (int i : idx) wanted |= !g.done[i];
    if (!wanted) return {};
    for (int i : rest) partial |= std::find(idx.begin(), idx.end(), i) == idx.end();
    if (partial && g.partial_reads++ == 0) return idx;
    return rest;
}
void force_group(fhelin_ctx* c, LazyRows& g, const std::vector<int>& idx) { force_rows(c, g, rows_for_read(g, idx)); }
}  // namespace fhelin
static void emit(fhelin_ctx* c, const CtVec& v, fhelin_ct** outs) {
    for (size_t i = 0; i < v.size(); ++i) outs[i] = wrap(c, v[i]);
}
static PtPtr opt(const fhelin_pt* p) { return p ? p->p : PtPtr(); }

extern "C" {

int fhelin_ct_force(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n) {
    NEED(c && (v || n == 0) && n >= 0);
    FHELIN_TRY
    std::vector<std::pair<LazyRows*, std::vector<int>>> groups;
    for (int i = 0; i < n; ++i) {
        const fhelin_ct* h = v[i];
        if (!h) throw Error(FHELIN_ERR_ARG, "null ciphertext handle in array");
        if (!h->p && h->heavy) force(c, h);
        if (h->p || !h->lazy) continue;
        LazyRows* g = h->lazy.get();
        auto it = std::find_if(groups.begin(), groups.end(), [&](const auto& e) { return e.first == g; });
        if (it == groups.end()) it = groups.insert(groups.end(), {g, {}});
        if (std::find(it->second.begin(), it->second.end(), h->lazy_idx) == it->second.end()) it->second.push_back(h->lazy_idx);
    }
    force_rows_multi(c, groups);   // exactly these rows (force_group would widen a second partial read); same calls on several samples merged
    FHELIN_CATCH
}
int fhelin_rotate_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t index, fhelin_ct** outs) {
    NEED(c && v && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.rotate_batch(vec_of(c, v, n), index), outs);
    FHELIN_CATCH
}
int fhelin_rescale_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** outs) {
    NEED(c && v && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.rescale_batch(vec_of(c, v, n)), outs);
    FHELIN_CATCH
}
int fhelin_mult_plain_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* p, fhelin_ct** outs) {
    NEED(c && v && p && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.mult_plain_batch(vec_of(c, v, n), p->p), outs);
    FHELIN_CATCH
}
int fhelin_mult_batch(fhelin_ctx* c, const fhelin_ct* const* a, const fhelin_ct* const* b, int32_t n, fhelin_ct** outs) {
    NEED(c && a && b && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.mult_batch(vec_of(c, a, n), vec_of(c, b, n)), outs);
    FHELIN_CATCH
}
int fhelin_add_batch(fhelin_ctx* c, const fhelin_ct* const* a, const fhelin_ct* const* b, int32_t n, fhelin_ct** outs) {
    NEED(c && a && b && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.add_batch(vec_of(c, a, n), vec_of(c, b, n)), outs);
    FHELIN_CATCH
}
int fhelin_fc_mult_const(fhelin_ctx* c, const fhelin_ct* a, double d, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.mult_const(ct_in(c, a), d));
    FHELIN_CATCH
}
int fhelin_fc_mask(fhelin_ctx* c, const fhelin_ct* a, int32_t kind, int32_t x, int32_t y, double v, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    switch (kind) {
        case 0: *out = wrap(c, c->comp.mask_block(ct_in(c, a), x, y, v)); break;
        case 1: *out = wrap(c, c->comp.mask_heads(ct_in(c, a), v)); break;
        case 2: *out = wrap(c, c->comp.mask_heads_128(ct_in(c, a), v)); break;
        case 3: *out = wrap(c, c->comp.mask_mod_n(ct_in(c, a), x, y)); break;
        case 4: *out = wrap(c, c->comp.mask_first_n(ct_in(c, a), x, v)); break;
        default: throw Error(FHELIN_ERR_ARG, "unknown mask kind");
    }
    FHELIN_CATCH
}
int fhelin_fc_rotsum(fhelin_ctx* c, const fhelin_ct* a, int32_t slots, int32_t padding, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.rotsum(ct_in(c, a), slots, padding));
    FHELIN_CATCH
}
int fhelin_fc_repeat(fhelin_ctx* c, const fhelin_ct* a, int32_t slots, int32_t padding, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.repeat(ct_in(c, a), slots, padding));
    FHELIN_CATCH
}
int fhelin_fc_add_many(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out) {
    NEED(c && v && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.add_many(vec_of(c, v, n)));
    FHELIN_CATCH
}
int fhelin_fc_matmul_pt(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* w, const fhelin_pt* bias,
                        int32_t slots, int32_t padding, fhelin_ct** outs) {
    NEED(c && rows && w && outs);
    FHELIN_TRY
    if (c->lazy_rows && n > 1) {
        auto g = std::make_shared<LazyRows>();
        g->kind = LazyRows::MatmulPt;
        g->rows = vec_of(c, rows, n);
        g->w = w->p;
        g->bias = opt(bias);
        g->slots = slots;
        g->padding = padding;
        emit_lazy(c, g, n, outs);
    } else {
        emit(c, c->comp.matmul_pt(vec_of(c, rows, n), w->p, opt(bias), slots, padding), outs);
    }
    FHELIN_CATCH
}
int fhelin_fc_matmul_ct(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_ct* w, int32_t slots,
                        int32_t padding, fhelin_ct** outs) {
    NEED(c && rows && w && outs);
    FHELIN_TRY
    emit(c, c->comp.matmul_ct(vec_of(c, rows, n), ct_in(c, w), slots, padding), outs);
    FHELIN_CATCH
}
int fhelin_fc_matmulRElarge(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* const* weights, int32_t nw,
                            const fhelin_pt* bias, double mask_val, fhelin_ct** outs) {
    NEED(c && rows && weights && outs);
    FHELIN_TRY
    std::vector<PtPtr> w;
    for (int i = 0; i < nw; ++i) {
        if (!weights[i]) throw Error(FHELIN_ERR_ARG, "null weight");
        w.push_back(weights[i]->p);
    }
    if (c->lazy_rows && n >= 1) {      // rows that generate_containers reads next are never evaluated on their own (capi_internal.h)
        auto g = std::make_shared<LazyRows>();
        g->kind = LazyRows::RElarge;
        g->rows = vec_of(c, rows, n);
        g->weights = w;
        g->bias = opt(bias);
        g->mask_val = mask_val;
        emit_lazy(c, g, n, outs);
    } else {
        emit(c, c->comp.matmulRElarge(vec_of(c, rows, n), w, opt(bias), mask_val), outs);
    }
    FHELIN_CATCH
}
int fhelin_fc_matmulCRlarge(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* const* weights,
                            const fhelin_pt* bias, fhelin_ct** outs) {
    NEED(c && rows && weights && outs);
    FHELIN_TRY
    std::vector<PtPtr> w;
    for (int i = 0; i < 4; ++i) {
        if (!weights[i]) throw Error(FHELIN_ERR_ARG, "null weight");
        w.push_back(weights[i]->p);
    }
    std::vector<CtVec> r;
    for (int i = 0; i < n; ++i) r.push_back(vec_of(c, rows + 4 * i, 4));
    emit(c, c->comp.matmulCRlarge(r, w, opt(bias)), outs);
    FHELIN_CATCH
}
int fhelin_fc_matmulScores(fhelin_ctx* c, const fhelin_ct* const* queries, int32_t n, const fhelin_ct* key, fhelin_ct** out) {
    NEED(c && queries && key && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.matmulScores(vec_of(c, queries, n), ct_in(c, key)));
    FHELIN_CATCH
}
int fhelin_fc_wrapUpRepeated(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out) {
    NEED(c && v && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.wrapUpRepeated(vec_of(c, v, n)));
    FHELIN_CATCH
}
int fhelin_fc_wrapUpExpanded(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out) {
    NEED(c && v && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.wrapUpExpanded(vec_of(c, v, n)));
    FHELIN_CATCH
}
int fhelin_fc_unwrapExpanded(fhelin_ctx* c, const fhelin_ct* a, int32_t n, fhelin_ct** outs) {
    NEED(c && a && outs);
    FHELIN_TRY
    if (c->lazy_rows && n > 1) {
        auto g = std::make_shared<LazyRows>();
        g->kind = LazyRows::UnwrapExpanded;
        g->src = ct_in(c, a);
        g->n = n;
        emit_lazy(c, g, n, outs);
    } else {
        emit(c, c->comp.unwrapExpanded(ct_in(c, a), n), outs);
    }
    FHELIN_CATCH
}
int fhelin_fc_unwrapScoresExpanded(fhelin_ctx* c, const fhelin_ct* a, int32_t n, fhelin_ct** outs) {
    NEED(c && a && outs);
    FHELIN_TRY
    emit(c, c->comp.unwrapScoresExpanded(ct_in(c, a), n), outs);
    FHELIN_CATCH
}
int fhelin_fc_unwrap_512_in_4_128(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** outs4) {
    NEED(c && a && outs4);
    FHELIN_TRY
    emit(c, c->comp.unwrap_512_in_4_128(ct_in(c, a), index), outs4);
    FHELIN_CATCH
}
int fhelin_fc_unwrapRepeatedLarge(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t input_number,
                                  fhelin_ct** outs) {
    NEED(c && containers && outs);
    FHELIN_TRY
    auto r = c->comp.unwrapRepeatedLarge(vec_of(c, containers, nc), input_number);
    for (size_t i = 0; i < r.size(); ++i) emit(c, r[i], outs + 4 * i);
    FHELIN_CATCH
}
int fhelin_fc_unwrapRepeatedLarge_range(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t input_number,
                                        int32_t first, int32_t count, fhelin_ct** outs) {
    NEED(c && containers && outs);
    FHELIN_TRY
    auto r = c->comp.unwrapRepeatedLarge(vec_of(c, containers, nc), input_number, first, count);
    for (size_t i = 0; i < r.size(); ++i) emit(c, r[i], outs + 4 * i);
    FHELIN_CATCH
}
int fhelin_fc_generate_containers(fhelin_ctx* c, const fhelin_ct* const* inputs, int32_t n, const fhelin_pt* bias,
                                  fhelin_ct** outs, int32_t* n_out) {
    NEED(c && inputs && outs);
    FHELIN_TRY
    // every input an unread row of matmulRElarge with one set of weights: the two calls as one (Composite::relarge_containers)
    bool fused = n > 0 && c->comp.fuse_relarge;
    const LazyRows* g0 = nullptr;
    for (int i = 0; fused && i < n; ++i) {
        const fhelin_ct* h = inputs[i];
        if (!h) throw Error(FHELIN_ERR_ARG, "null ciphertext handle in array");
        const LazyRows* g = (!h->p && !h->heavy && h->lazy) ? h->lazy.get() : nullptr;
        fused = g && g->kind == LazyRows::RElarge && !g->done[h->lazy_idx] && g->rows.size() == g->done.size();
        if (!fused) break;
        if (!g0) g0 = g;
        fused = g->weights == g0->weights && g->bias == g0->bias && g->mask_val == g0->mask_val;
    }
    CtVec r;
    if (fused) {
        CtVec x;
        for (int i = 0; i < n; ++i) {
            const fhelin_ct* h = inputs[i];
            x.push_back(h->lazy->rows[h->lazy_idx]);
            // level plan: the containers consume what the rows of matmulRElarge would have consumed
            if (c->plan.live(h->node, h->node_epoch))
                for (int in : c->plan.nodes[h->node].in) plan_inputs().push_back(in);
        }
        r = c->comp.relarge_containers(x, g0->weights, g0->bias, g0->mask_val, opt(bias));
    } else {
        r = c->comp.generate_containers(vec_of(c, inputs, n), opt(bias));
    }
    emit(c, r, outs);
    if (n_out) *n_out = (int)r.size();
    FHELIN_CATCH
}
int fhelin_fc_wrap_containers(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t inputs_number, fhelin_ct** out) {
    NEED(c && v && out);
    FHELIN_TRY
    *out = wrap(c, c->comp.wrap_containers(vec_of(c, v, n), inputs_number));
    FHELIN_CATCH
}

/* ---- the composite calls on a BATCH OF SAMPLES (include/fhelin.h: fhelin_fcb_*) ---- */
static std::vector<CtVec> groups_of(fhelin_ctx* c, const fhelin_ct* const* v, int n, int B) {
    if (n < 0 || B < 0) throw Error(FHELIN_ERR_ARG, "negative count");
    const CtVec flat = vec_of(c, v, n * B);     // the deferred rows of ALL samples are evaluated together (force_many)
    std::vector<CtVec> g(B);
    for (int x = 0; x < B; ++x) g[x].assign(flat.begin() + (size_t)x * n, flat.begin() + (size_t)(x + 1) * n);
    return g;
}
int fhelin_fcb_matmulScores(fhelin_ctx* c, const fhelin_ct* const* queries, int32_t n, const fhelin_ct* const* keys, int32_t B, fhelin_ct** outs) {
    NEED(c && queries && keys && outs);
    FHELIN_TRY
    emit(c, c->comp.matmulScores_multi(groups_of(c, queries, n, B), vec_of(c, keys, B)), outs);
    FHELIN_CATCH
}
int fhelin_fcb_matmul_ct(fhelin_ctx* c, const fhelin_ct* const* rows, const fhelin_ct* const* ws, int32_t n, int32_t slots, int32_t padding,
                         fhelin_ct** outs) {
    NEED(c && rows && ws && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->comp.matmul_ct_each(vec_of(c, rows, n), vec_of(c, ws, n), slots, padding), outs);
    FHELIN_CATCH
}
int fhelin_fcb_wrapUpRepeated(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs) {
    NEED(c && v && outs);
    FHELIN_TRY
    CtVec r;
    for (const CtVec& g : groups_of(c, v, n, B)) r.push_back(c->comp.wrapUpRepeated(g));   // one inner-product pass per sample fills the GPU
    emit(c, r, outs);
    FHELIN_CATCH
}
int fhelin_fcb_wrapUpExpanded(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs) {
    NEED(c && v && outs);
    FHELIN_TRY
    emit(c, c->comp.wrapUpExpanded_multi(groups_of(c, v, n, B)), outs);
    FHELIN_CATCH
}
int fhelin_fcb_unwrapExpanded(fhelin_ctx* c, const fhelin_ct* const* cs, int32_t B, int32_t n, fhelin_ct** outs) {
    NEED(c && cs && outs && B >= 0 && n >= 0);
    FHELIN_TRY
    if (c->lazy_rows && n > 1) {
        // one deferred group per sample, as B calls of fhelin_fc_unwrapExpanded make them: each keeps the read pattern of its own sample
        // (which rows, how many together - that decides the form), and groups read together are evaluated together (force_rows_multi)
        for (int x = 0; x < B; ++x) {
            if (!cs[x]) throw Error(FHELIN_ERR_ARG, "null ciphertext handle in array");
            auto g = std::make_shared<LazyRows>();
            g->kind = LazyRows::UnwrapExpanded;
            plan_inputs().clear();
            g->src = ct_in(c, cs[x]);
            g->n = n;
            emit_lazy(c, g, n, outs + (size_t)x * n);
        }
    } else {
        std::vector<int> all(n);
        for (int i = 0; i < n; ++i) all[i] = i;
        const std::vector<CtVec> r = c->comp.unwrapExpanded_rows_multi(vec_of(c, cs, B), n, all);
        for (int x = 0; x < B; ++x) emit(c, r[x], outs + (size_t)x * n);
    }
    FHELIN_CATCH
}
int fhelin_fcb_unwrapRepeatedLarge(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t B, int32_t input_number,
                                   fhelin_ct** outs) {
    NEED(c && containers && outs);
    FHELIN_TRY
    const auto r = c->comp.unwrapRepeatedLarge_multi(groups_of(c, containers, nc, B), input_number);
    for (int x = 0; x < B; ++x)
        for (size_t i = 0; i < r[x].size(); ++i) emit(c, r[x][i], outs + ((size_t)x * input_number + i) * 4);
    FHELIN_CATCH
}
int fhelin_fcb_generate_containers(fhelin_ctx* c, const fhelin_ct* const* inputs, int32_t n, int32_t B, const fhelin_pt* bias,
                                   fhelin_ct** outs, int32_t* n_out) {
    NEED(c && inputs && outs && n >= 0 && B >= 0);
    FHELIN_TRY
    // every input an unread row of matmulRElarge with one set of weights: the two calls as one (Composite::relarge_containers_multi)
    const int total = n * B;
    bool fused = total > 0 && c->comp.fuse_relarge;
    const LazyRows* g0 = nullptr;
    for (int i = 0; fused && i < total; ++i) {
        const fhelin_ct* h = inputs[i];
        if (!h) throw Error(FHELIN_ERR_ARG, "null ciphertext handle in array");
        const LazyRows* g = (!h->p && !h->heavy && h->lazy) ? h->lazy.get() : nullptr;
        fused = g && g->kind == LazyRows::RElarge && !g->done[h->lazy_idx] && g->rows.size() == g->done.size();
        if (!fused) break;
        if (!g0) g0 = g;
        fused = g->weights == g0->weights && g->bias == g0->bias && g->mask_val == g0->mask_val;
    }
    std::vector<CtVec> r;
    if (fused) {
        std::vector<CtVec> x(B);
        for (int i = 0; i < total; ++i) {
            const fhelin_ct* h = inputs[i];
            x[i / n].push_back(h->lazy->rows[h->lazy_idx]);
            // level plan: the containers consume what the rows of matmulRElarge would have consumed
            if (c->plan.live(h->node, h->node_epoch))
                for (int in : c->plan.nodes[h->node].in) plan_inputs().push_back(in);
        }
        r = c->comp.relarge_containers_multi(x, g0->weights, g0->bias, g0->mask_val, opt(bias));
    } else {
        r = c->comp.generate_containers_multi(groups_of(c, inputs, n, B), opt(bias));
    }
    const size_t per = r.empty() ? 0 : r[0].size();
    for (int x = 0; x < B; ++x) emit(c, r[x], outs + (size_t)x * per);
    if (n_out) *n_out = (int)per;
    FHELIN_CATCH
}
int fhelin_fc_rotsum_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t slots, int32_t padding, int32_t repeat, fhelin_ct** outs) {
    NEED(c && v && outs && n >= 0);
    FHELIN_TRY
    emit(c, repeat ? c->comp.repeat_batch(vec_of(c, v, n), slots, padding) : c->comp.rotsum_batch(vec_of(c, v, n), slots, padding), outs);
    FHELIN_CATCH
}
int fhelin_add_plain_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* p, fhelin_ct** outs) {
    NEED(c && v && p && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.add_plain_batch(vec_of(c, v, n), p->p), outs);
    FHELIN_CATCH
}
int fhelin_eval_poly_batch(fhelin_ctx* c, const fhelin_ct* const* xs, int32_t n, const double* coeffs, int32_t n_coeffs, fhelin_ct** outs) {
    NEED(c && xs && coeffs && outs && n >= 0);
    FHELIN_TRY
    emit(c, c->ev.eval_poly_many(vec_of(c, xs, n), std::vector<double>(coeffs, coeffs + n_coeffs)), outs);
    FHELIN_CATCH
}
int fhelin_mult_many_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs) {
    NEED(c && v && outs);
    FHELIN_TRY
    // handles that are the SAME object repeat as the same ciphertext (EvalMultMany({r, r, ...}) squares)
    emit(c, c->ev.mult_many_rows(groups_of(c, v, n, B)), outs);
    FHELIN_CATCH
}

int fhelin_mult_real(fhelin_ctx* c, const fhelin_ct* a, double k, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.mult_real(ct_in(c, a), k));
    FHELIN_CATCH
}
int fhelin_add_real(fhelin_ctx* c, const fhelin_ct* a, double k, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.add_real(ct_in(c, a), k));
    FHELIN_CATCH
}
int fhelin_mult_many(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out) {
    NEED(c && v && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.mult_many(vec_of(c, v, n)));
    FHELIN_CATCH
}
int fhelin_lincomb(fhelin_ctx* c, const fhelin_ct* const* v, const double* coeffs, int32_t n, double c0, fhelin_ct** out) {
    NEED(c && v && coeffs && out && n >= 1);
    FHELIN_TRY
    *out = wrap(c, c->ev.lincomb(vec_of(c, v, n), std::vector<double>(coeffs, coeffs + n), c0));
    FHELIN_CATCH
}
int fhelin_eval_poly(fhelin_ctx* c, const fhelin_ct* x, const double* coeffs, int32_t n, fhelin_ct** out) {
    NEED(c && x && coeffs && out);
    FHELIN_TRY
    const std::vector<double> cf(coeffs, coeffs + n);
    *out = wrap(c, run_heavy(c, x, [&](const CtPtr& in) { return c->ev.eval_poly(in, cf); }));
    FHELIN_CATCH
}
int fhelin_eval_chebyshev(fhelin_ctx* c, const fhelin_ct* x, const double* coeffs, int32_t n, double a, double b, fhelin_ct** out) {
    NEED(c && x && coeffs && out);
    FHELIN_TRY
    const std::vector<double> cf(coeffs, coeffs + n);
    if (n < 2) throw Error(FHELIN_ERR_ARG, "eval_chebyshev: need degree >= 1");
    if (defer_ok(c)) {
        auto op = std::make_shared<LazyHeavy>();
        op->kind = LazyHeavy::Cheb;
        op->coeffs = cf;
        op->a = a;
        op->b = b;
        *out = defer_heavy(c, x, op);
    } else {
        *out = wrap(c, run_heavy(c, x, [&](const CtPtr& in) { return c->ev.eval_chebyshev(in, cf, a, b); }));
    }
    FHELIN_CATCH
}
int fhelin_eval_chebyshev_batch(fhelin_ctx* c, const fhelin_ct* const* xs, int32_t n, const double* coeffs, int32_t n_coeffs, double a,
                                double b, fhelin_ct** outs) {
    NEED(c && xs && coeffs && outs && n >= 0);
    FHELIN_TRY
    const std::vector<double> cf(coeffs, coeffs + n_coeffs);
    emit(c, c->ev.eval_chebyshev_many(vec_of(c, xs, n), cf, a, b), outs);
    FHELIN_CATCH
}
int fhelin_bootstrap_setup(fhelin_ctx* c, int32_t budget_enc, int32_t budget_dec, int32_t slots) {
    NEED(c);
    FHELIN_TRY
    c->boot.setup(budget_enc, budget_dec, slots);
    FHELIN_CATCH
}
int fhelin_bootstrap(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    // a bootstrap is a terminal for its input (two limbs are all it reads) and a source of the level plan for its output
    const int drop = c->plan.next_drop(c->boot.out_ell());
    if (c->plan.live(a->node, a->node_epoch)) c->plan.terminal(a->node, 2);
    if (defer_ok(c)) {
        if (!c->boot.ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
        auto op = std::make_shared<LazyHeavy>();
        op->kind = LazyHeavy::Boot;
        op->drop = drop;
        *out = defer_heavy(c, a, op);
        return FHELIN_OK;
    }
    *out = wrap(c, run_heavy(c, a, [&](const CtPtr& in) { return c->boot.bootstrap(in, drop); }));
    if (c->plan.live((*out)->node, (*out)->node_epoch)) {
        LevelPlan::Node& nd = c->plan.nodes[(*out)->node];
        nd.in.clear();
        nd.ordinal = c->plan.next_ordinal - 1;
    }
    FHELIN_CATCH
}
int fhelin_bootstrap_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** outs) {
    NEED(c && v && outs && n >= 0);
    FHELIN_TRY
    // every ciphertext is a source of the level plan of its own (call order); those with the same planned drop share a batch
    CtVec in = vec_of(c, v, n);
    std::vector<int> drop(n);
    for (int i = 0; i < n; ++i) {
        drop[i] = c->plan.next_drop(c->boot.out_ell());
        if (c->plan.live(v[i]->node, v[i]->node_epoch)) c->plan.terminal(v[i]->node, 2);
    }
    const int first_ordinal = c->plan.next_ordinal - n;
    CtVec r(n);
    std::vector<char> done(n, 0);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        std::vector<int> pick;
        CtVec sub;
        for (int k = i; k < n; ++k)
            if (!done[k] && drop[k] == drop[i]) {
                pick.push_back(k);
                sub.push_back(in[k]);
                done[k] = 1;
            }
        CtVec o = c->boot.bootstrap_batch(sub, drop[i]);
        for (size_t k = 0; k < pick.size(); ++k) r[pick[k]] = o[k];
    }
    plan_inputs().clear();
    for (int i = 0; i < n; ++i) {
        outs[i] = wrap(c, r[i]);
        if (c->plan.live(outs[i]->node, outs[i]->node_epoch)) c->plan.nodes[outs[i]->node].ordinal = first_ordinal + i;
    }
    FHELIN_CATCH
}
int fhelin_bootstrap_partial(fhelin_ctx* c, const fhelin_ct* a, int32_t stage, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->boot.partial(ct_in(c, a), stage));
    FHELIN_CATCH
}
int fhelin_bootstrap_drop(fhelin_ctx* c, const fhelin_ct* a, int32_t drop, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->boot.bootstrap(ct_in(c, a), drop));
    FHELIN_CATCH
}
int fhelin_bootstrap_describe(fhelin_ctx* c, int32_t* out, int32_t cap, int32_t* n) {
    NEED(c && n && (out || cap == 0));
    FHELIN_TRY
    if (!c->boot.ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
    const Bootstrapper& b = c->boot;
    std::vector<int32_t> d = {b.packed() ? 1 : 0, b.slots(), b.K, b.R, b.cheb_degree, b.correction, b.depth(),
                              (int32_t)b.stages(false).size(), (int32_t)b.stages(true).size()};
    for (int which = 0; which < 2; ++which)
        for (const LinStage& st : b.stages(which != 0)) {
            d.push_back(st.terms.empty() ? 0 : st.terms[0].diag->slots);
            d.push_back((int32_t)st.terms.size());
            for (const auto& t : st.terms) {
                d.push_back(t.giant);
                d.push_back(t.baby);
            }
        }
    *n = (int32_t)d.size();
    for (int i = 0; i < std::min(cap, *n); ++i) out[i] = d[i];
    FHELIN_CATCH
}
int fhelin_bootstrap_diag(fhelin_ctx* c, int32_t which, int32_t stage, int32_t term, fhelin_pt** out) {
    NEED(c && out);
    FHELIN_TRY
    if (!c->boot.ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
    const auto& sts = c->boot.stages(which != 0);
    if (which < 0 || which > 1 || stage < 0 || stage >= (int)sts.size() || term < 0 || term >= (int)sts[stage].terms.size())
        throw Error(FHELIN_ERR_ARG, "bootstrap_diag: no such term");
    auto* h = new fhelin_pt;
    h->p = sts[stage].terms[term].diag;
    *out = h;
    FHELIN_CATCH
}
int fhelin_bootstrap_cheb(fhelin_ctx* c, double* out, int32_t cap, int32_t* n) {
    NEED(c && n && (out || cap == 0));
    FHELIN_TRY
    if (!c->boot.ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
    const std::vector<double>& cf = c->boot.cheb();
    *n = (int32_t)cf.size();
    for (int i = 0; i < std::min(cap, *n); ++i) out[i] = cf[i];
    FHELIN_CATCH
}
int fhelin_bootstrap_config(fhelin_ctx* c, int32_t K, int32_t R, int32_t cheb_degree, int32_t correction) {
    NEED(c);
    FHELIN_TRY
    if (K < 1 || R < 0 || R > 8 || cheb_degree < 3 || cheb_degree > 255 || correction < 0 || correction > 20)
        throw Error(FHELIN_ERR_ARG, "bootstrap_config: parameter out of range");
    c->boot.K = K;
    c->boot.R = R;
    c->boot.cheb_degree = cheb_degree;
    c->boot.correction = correction;
    FHELIN_CATCH
}

}  // extern "C"
