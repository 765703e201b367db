// extern "C" boundary: translates C++ exceptions into status codes, owns nothing but handles.
#include "../../include/fhelin.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "context.h"
#include <map>
#include "capi_internal.h"

using namespace fhelin;

static thread_local std::string g_last_error;

namespace fhelin {
int capi_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace fhelin

fhelin_ctx::fhelin_ctx(const fhelin::Params& p) : ctx(p), ev(ctx), cl(ev, ctx.prm.seed_bytes), comp(ev, cl), boot(ev, cl) {
    if (const char* e = std::getenv("FHELIN_LAZY_ROWS")) lazy_rows = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_LAZY_HEAVY")) lazy_heavy = std::atoi(e) != 0;
}

#define NEED(x) if (!(x)) return capi_fail(FHELIN_ERR_ARG, "null argument")

namespace fhelin {
std::vector<int>& plan_inputs() {
    static thread_local std::vector<int> v;
    return v;
}
// Back to front over the recording: a value that some chain of consumers needs with `need` effective limbs asks of each of its
// producers' inputs need + (limbs the producing call consumed); an input that sat above the lowest one at recording time
// keeps one limb more than that, so that the level adjustment it went through (integer scalar x rescale) still has its limb.
void LevelPlan::finish() {
    target.assign(next_ordinal, -1);
    for (int n = (int)nodes.size() - 1; n >= 0; --n) {
        Node& nd = nodes[n];
        if (nd.need < 0 || nd.eff < 0) continue;
        if (nd.ordinal >= 0) {
            if (nd.ordinal < (int)target.size()) target[nd.ordinal] = std::min(nd.eff, std::max(nd.need, 1));
            continue;
        }
        int lo = 1 << 30;
        for (int i : nd.in)
            if (nodes[i].eff >= 0) lo = std::min(lo, nodes[i].eff);
        if (lo == 1 << 30) continue;
        const int used = std::max(0, lo - nd.eff);
        for (int i : nd.in) {
            Node& src = nodes[i];
            if (src.eff < 0) continue;
            src.need = std::max(src.need, std::min(src.eff, nd.need + used + (src.eff > lo ? 1 : 0)));
        }
    }
}
}  // namespace fhelin

extern "C" {

const char* fhelin_last_error(void) { return g_last_error.c_str(); }
const char* fhelin_version(void) { return "fhelin_amd 0.1 (gfx950)"; }

static int ctx_create(const fhelin_params* p, const uint8_t* seed32, fhelin_ctx** out) {
    if (!p || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    Params q;
    if (seed32) {
        std::memcpy(q.seed_bytes, seed32, 32);
        q.have_seed_bytes = true;
    }
    q.log_n = p->log_n;
    q.n_q = p->n_q;
    q.first_bits = p->first_bits;
    q.scale_bits = p->scale_bits;
    q.n_p = p->n_p;
    q.special_bits = p->special_bits;
    q.dnum = p->dnum;
    q.log_slots = p->log_slots;
    q.hamming = p->hamming;
    q.device = p->device;
    q.seed = p->seed;
    auto* c = new fhelin_ctx(q);
    *out = c;
    FHELIN_CATCH
}
int fhelin_ctx_create(const fhelin_params* p, fhelin_ctx** out) { return ctx_create(p, nullptr, out); }
int fhelin_ctx_create_seeded(const fhelin_params* p, const uint8_t* seed32, fhelin_ctx** out) {
    if (!seed32) return capi_fail(FHELIN_ERR_ARG, "null seed");
    return ctx_create(p, seed32, out);
}
int fhelin_ctx_secret_seed(const fhelin_ctx* c, uint8_t* out32) {
    if (!c || !out32) return capi_fail(FHELIN_ERR_ARG, "null argument");
    std::memcpy(out32, c->ctx.prm.seed_bytes, 32);
    return FHELIN_OK;
}
int fhelin_prng_block(const uint8_t* seed32, uint64_t counter, uint64_t stream, uint8_t* out64) {
    if (!seed32 || !out64) return capi_fail(FHELIN_ERR_ARG, "null argument");
    Prng::block(seed32, counter, stream, out64);
    return FHELIN_OK;
}

void fhelin_ctx_destroy(fhelin_ctx* c) {
    if (c && std::getenv("FHELIN_COPY_STATS")) {
        // diagnostics: ciphertext copies made to line operands up for a batched key switch, per call site of make_contiguous
        // (0 rotate_sum_batch, 1 rotate_each_sum, 2 rotate_each_sum_rows, 3 rotate_each, 4 rotate_batch, 5 rescale_batch, 6 hoisted batches, 7 modraise)
        std::fprintf(stderr, "fhelin gather copies:");
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %llu", (unsigned long long)c->ev.gather_copies[i]);
        std::fprintf(stderr, "\n");
    }
    if (c)
        for (hipEvent_t e : c->lane_mark)
            if (e) (void)hipEventDestroy(e);
    delete c;
}

int fhelin_ctx_info(const fhelin_ctx* c, fhelin_params* out, int32_t* alpha, int32_t* has_device) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    if (out) {
        const Params& q = c->ctx.prm;
        out->log_n = q.log_n;
        out->n_q = q.n_q;
        out->first_bits = q.first_bits;
        out->scale_bits = q.scale_bits;
        out->n_p = q.n_p;
        out->special_bits = q.special_bits;
        out->dnum = q.dnum;
        out->log_slots = q.log_slots;
        out->hamming = q.hamming;
        out->device = q.device;
        out->seed = q.seed;
    }
    if (alpha) *alpha = c->ctx.alpha;
    if (has_device) *has_device = c->ctx.has_device ? 1 : 0;
    return FHELIN_OK;
}

int fhelin_ctx_moduli(const fhelin_ctx* c, uint64_t* out, int32_t cap) {
    if (!c || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    if (cap < (int)c->ctx.moduli.size()) return capi_fail(FHELIN_ERR_ARG, "buffer too small");
    std::memcpy(out, c->ctx.moduli.data(), c->ctx.moduli.size() * sizeof(uint64_t));
    return FHELIN_OK;
}

int fhelin_ctx_roots(const fhelin_ctx* c, uint64_t* out, int32_t cap) {
    if (!c || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    if (cap < (int)c->ctx.tw.size()) return capi_fail(FHELIN_ERR_ARG, "buffer too small");
    for (size_t i = 0; i < c->ctx.tw.size(); ++i) out[i] = c->ctx.tw[i].psi;
    return FHELIN_OK;
}

int fhelin_ctx_scaling_factors(const fhelin_ctx* c, double* out, int32_t cap) {
    if (!c || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    if (cap < c->ctx.prm.n_q) return capi_fail(FHELIN_ERR_ARG, "buffer too small");
    for (int i = 0; i < c->ctx.prm.n_q; ++i) out[i] = (double)c->ctx.sf_real[i];
    return FHELIN_OK;
}

int fhelin_ctx_set_stream(fhelin_ctx* c, void* hip_stream) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    c->ctx.sync();
    if (c->ctx.own_stream && c->ctx.main_stream) hip_check(hipStreamDestroy(c->ctx.main_stream), "hipStreamDestroy");
    c->ctx.stream = c->ctx.main_stream = (hipStream_t)hip_stream;
    c->ctx.pool.lane_stream[0] = c->ctx.main_stream;
    c->ctx.own_stream = false;
    FHELIN_CATCH
}

int fhelin_ctx_set_lazy_rows(fhelin_ctx* c, int32_t on) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    c->lazy_rows = on != 0;
    return FHELIN_OK;
}

int fhelin_level_plan_begin(fhelin_ctx* c, int32_t mode) {
    NEED(c);
    if (mode < 0 || mode > 2) return capi_fail(FHELIN_ERR_ARG, "level plan: mode is 0 (off), 1 (record) or 2 (apply)");
    if (mode == 2 && c->plan.target.empty()) return capi_fail(FHELIN_ERR_STATE, "level plan: nothing recorded or loaded to apply");
    c->plan.begin(mode);
    return FHELIN_OK;
}
int fhelin_level_plan_seek(fhelin_ctx* c, int32_t source) {
    NEED(c);
    if (source < 0) return capi_fail(FHELIN_ERR_ARG, "level plan: negative source index");
    if (c->plan.mode == 1) return capi_fail(FHELIN_ERR_STATE, "level plan: a recording pass runs from its first source");
    c->plan.next_ordinal = source;
    return FHELIN_OK;
}
int fhelin_level_plan_tell(fhelin_ctx* c, int32_t* mode, int32_t* source) {
    NEED(c);
    if (mode) *mode = c->plan.mode;
    if (source) *source = c->plan.next_ordinal;
    return FHELIN_OK;
}
int fhelin_level_plan_end(fhelin_ctx* c, int32_t* n_sources) {
    NEED(c);
    FHELIN_TRY
    if (c->plan.mode == 1) c->plan.finish();
    if (n_sources) *n_sources = (int32_t)c->plan.target.size();
    c->plan.begin(0);
    FHELIN_CATCH
}
int fhelin_level_plan_get(fhelin_ctx* c, int32_t* target, int32_t cap, int32_t* n) {
    NEED(c && n && (target || cap <= 0));
    *n = (int32_t)c->plan.target.size();
    for (int32_t i = 0; i < cap && i < *n; ++i) target[i] = c->plan.target[i];
    return FHELIN_OK;
}
int fhelin_level_plan_set(fhelin_ctx* c, const int32_t* target, int32_t n) {
    NEED(c && (target || n == 0));
    if (n < 0) return capi_fail(FHELIN_ERR_ARG, "level plan: negative length");
    c->plan.target.assign(target, target + n);
    return FHELIN_OK;
}
int fhelin_sync(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    // deferred bootstraps / polynomial evaluations count as issued work; the first failure among them is this call's error
    // (everything that could be evaluated has been: the device is synchronised either way)
    if (c->any_pending()) {
        try {
            flush_heavy_all(c, true);
        } catch (...) {
            c->ctx.sync();
            throw;
        }
    }
    c->ctx.sync();
    FHELIN_CATCH
}

int fhelin_ctx_set_lane(fhelin_ctx* c, int32_t lane) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (lane < 0 || lane > c->ctx.n_lanes) throw Error(FHELIN_ERR_ARG, "set_lane: the context has lanes 0.." + std::to_string(c->ctx.n_lanes));
    c->user_lane = lane;
    c->ctx.stream = lane == 0 ? c->ctx.main_stream : c->ctx.lane_stream[lane];
    c->ctx.pool.cur_lane = lane;
    FHELIN_CATCH
}
int fhelin_ctx_lane_wait(fhelin_ctx* c, int32_t from_lane) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (from_lane < 0 || from_lane > c->ctx.n_lanes) throw Error(FHELIN_ERR_ARG, "lane_wait: no such lane");
    if (from_lane != c->ctx.pool.cur_lane) {
        // what was DEFERRED under that lane is issued there first, then this lane's stream goes behind everything queued on it
        if (!c->pending_heavy[from_lane].empty()) {
            const int cur = c->ctx.pool.cur_lane;
            hipStream_t cur_stream = c->ctx.stream;
            c->ctx.stream = from_lane == 0 ? c->ctx.main_stream : c->ctx.lane_stream[from_lane];
            c->ctx.pool.cur_lane = from_lane;
            try {
                flush_heavy(c);
            } catch (...) {
                c->ctx.stream = cur_stream;
                c->ctx.pool.cur_lane = cur;
                throw;
            }
            c->ctx.stream = cur_stream;
            c->ctx.pool.cur_lane = cur;
        }
        wait_for_lane(c, from_lane);
    }
    FHELIN_CATCH
}
int fhelin_ctx_lane_mark(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    const int k = c->ctx.pool.cur_lane;
    if (!c->lane_mark[k]) hip_check(hipEventCreateWithFlags(&c->lane_mark[k], hipEventDisableTiming), "hipEventCreate(lane mark)");
    if (!c->pending_heavy[k].empty()) flush_heavy(c);
    hip_check(hipEventRecord(c->lane_mark[k], c->ctx.stream), "hipEventRecord(lane mark)");
    c->lane_marked[k] = true;
    FHELIN_CATCH
}
int fhelin_ctx_lane_wait_mark(fhelin_ctx* c, int32_t from_lane) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (from_lane < 0 || from_lane > c->ctx.n_lanes) throw Error(FHELIN_ERR_ARG, "lane_wait_mark: no such lane");
    if (from_lane != c->ctx.pool.cur_lane && c->lane_marked[from_lane])
        hip_check(hipStreamWaitEvent(c->ctx.stream, c->lane_mark[from_lane], 0), "hipStreamWaitEvent(lane mark)");
    FHELIN_CATCH
}
int fhelin_ctx_lanes_fork(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (c->user_lane != 0) throw Error(FHELIN_ERR_STATE, "lanes_fork: call it under lane 0");
    if (!c->pending_heavy[0].empty()) flush_heavy(c);     // what the lanes are about to read exists
    c->ctx.fork_lanes();
    c->ctx.pool.n_user_lanes = c->ctx.n_lanes;
    c->ctx.pool.conservative_foreign_free = true;
    for (bool& m : c->lane_marked) m = false;
    FHELIN_CATCH
}
int fhelin_ctx_lanes_join(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (c->user_lane != 0) throw Error(FHELIN_ERR_STATE, "lanes_join: call it under lane 0");
    if (c->any_pending()) flush_heavy_all(c);
    c->ctx.join_lanes();
    c->ctx.pool.conservative_foreign_free = false;
    FHELIN_CATCH
}

int fhelin_ctx_trim(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    if (c->any_pending()) flush_heavy_all(c, true);
    c->ctx.sync();
    c->ctx.pool.trim();
    FHELIN_CATCH
}

int fhelin_timer_start(fhelin_ctx* c) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    hip_check(hipEventRecord(c->ctx.ev_start, c->ctx.stream), "hipEventRecord");
    FHELIN_CATCH
}

int fhelin_timer_stop(fhelin_ctx* c, float* ms) {
    if (!c || !ms) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    c->ctx.require_device();
    if (c->any_pending()) flush_heavy_all(c);   // deferred operations issued inside the timed region belong to it
    hip_check(hipEventRecord(c->ctx.ev_stop, c->ctx.stream), "hipEventRecord");
    hip_check(hipEventSynchronize(c->ctx.ev_stop), "hipEventSynchronize");
    hip_check(hipEventElapsedTime(ms, c->ctx.ev_start, c->ctx.ev_stop), "hipEventElapsedTime");
    FHELIN_CATCH
}

int fhelin_dev_alloc(fhelin_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    c->ctx.require_device();
    *out = c->ctx.pool.alloc(bytes);
    FHELIN_CATCH
}

int fhelin_dev_free(fhelin_ctx* c, void* p) {
    if (!c) return capi_fail(FHELIN_ERR_ARG, "null context");
    FHELIN_TRY
    c->ctx.require_device();
    c->ctx.pool.free(p);
    FHELIN_CATCH
}

int fhelin_dev_upload(fhelin_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    c->ctx.require_device();
    hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->ctx.stream), "hipMemcpyAsync H2D");
    hip_check(hipStreamSynchronize(c->ctx.stream), "hipStreamSynchronize");
    FHELIN_CATCH
}

int fhelin_dev_download(fhelin_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    c->ctx.require_device();
    hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->ctx.stream), "hipMemcpyAsync D2H");
    hip_check(hipStreamSynchronize(c->ctx.stream), "hipStreamSynchronize");
    FHELIN_CATCH
}

int fhelin_ntt(fhelin_ctx* c, uint64_t* d_data, int32_t nvec, int32_t limb_first, int32_t limb_count, int32_t inverse) {
    if (!c || !d_data) return capi_fail(FHELIN_ERR_ARG, "null argument");
    FHELIN_TRY
    Context& x = c->ctx;
    x.require_device();
    if (nvec < 0 || limb_count < 1 || limb_first < 0 || limb_first + limb_count > (int)x.moduli.size())
        throw Error(FHELIN_ERR_ARG, "fhelin_ntt: limb range outside the context's moduli");
    LimbBatch b{d_data, nvec, nullptr, limb_first, limb_count};
    x.ntt(b, inverse != 0);
    hip_check(hipGetLastError(), "launch_ntt");
    FHELIN_CATCH
}

int fhelin_stats(fhelin_ctx* c, uint64_t* out, int32_t cap, int32_t reset) {
    if (!c || !out) return capi_fail(FHELIN_ERR_ARG, "null argument");
    if (cap < 7) return capi_fail(FHELIN_ERR_ARG, "need room for 7 counters");
    OpStats& s = c->ctx.stats;
    out[0] = s.limb_ntt;
    out[1] = s.keyswitch;
    out[2] = s.keyswitch_limbs;
    out[3] = s.rescale;
    out[4] = s.ct_pt_mult;
    out[5] = s.bootstrap;
    out[6] = s.encode;
    if (cap >= 9) {
        out[7] = s.rescale_limbs;
        out[8] = s.ct_pt_limbs;
    }
    if (cap >= 12) {   // device-pool growth: blocks obtained from hipMalloc, their bytes, host nanoseconds inside hipMalloc
        DevicePool& p = c->ctx.pool;
        out[9] = p.malloc_calls;
        out[10] = p.malloc_bytes;
        out[11] = p.malloc_ns;
        if (cap >= 16) {   // bytes the pool holds now, high-water marks of bytes in use / held, out-of-memory trims
            out[12] = p.bytes_reserved();
            out[13] = p.live_peak;
            out[14] = p.reserved_peak;
            out[15] = p.trims;
            if (reset) {   // the marks restart from the present state
                p.live_peak = p.bytes_live();
                p.reserved_peak = p.bytes_reserved();
                p.trims = 0;
            }
        }
        if (reset) p.malloc_calls = p.malloc_bytes = p.malloc_ns = 0;
    }
    if (reset) s = OpStats();
    return FHELIN_OK;
}


// ---- host-side self-test of the device arena's logic (no GPU: a mock backend hands out address ranges of a pretend device) -----------
namespace {
size_t g_mock_cap = 0, g_mock_used = 0, g_mock_cursor = 0;
std::map<uintptr_t, size_t> g_mock_blocks;
void* mock_malloc(size_t bytes) {
    if (g_mock_used + bytes > g_mock_cap) return nullptr;
    g_mock_used += bytes;
    const uintptr_t p = (uintptr_t(1) << 40) + g_mock_cursor;     // never dereferenced
    g_mock_cursor += bytes + (size_t(1) << 21);
    g_mock_blocks[p] = bytes;
    return reinterpret_cast<void*>(p);
}
void mock_free(void* p) {
    auto it = g_mock_blocks.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_mock_blocks.end()) throw fhelin::Error(FHELIN_ERR_INTERNAL, "mock backend: free of unknown slab");
    g_mock_used -= it->second;
    g_mock_blocks.erase(it);
}
}  // namespace
int fhelin_debug_pool_selftest(uint64_t seed, int32_t n_ops, uint64_t device_bytes, uint64_t* out, int32_t cap) {
    if (!out || cap < 6 || n_ops < 1) return capi_fail(FHELIN_ERR_ARG, "pool selftest: need room for 6 results");
    FHELIN_TRY
    g_mock_cap = device_bytes;
    g_mock_used = g_mock_cursor = 0;
    g_mock_blocks.clear();
    DevicePool::Backend be;
    be.malloc_fn = mock_malloc;
    be.free_fn = mock_free;
    {
        DevicePool pool(be);
        std::map<uintptr_t, size_t> live;       // what the test believes is allocated: address -> bytes asked
        std::vector<uintptr_t> order;
        u64 x = seed * 0x9E3779B97F4A7C15ull + 1;
        auto rnd = [&]() {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            return x;
        };
        size_t want_live = 0, peak_live = 0, oom = 0;
        for (int op = 0; op < n_ops; ++op) {
            const bool do_alloc = live.empty() || (rnd() % 100) < 55;
            if (do_alloc) {
                // the engine's mix: limb vectors (512 KiB), ciphertexts (tens of MiB), batch blocks (GBs), odd small tables
                size_t bytes;
                switch (rnd() % 6) {
                    case 0: bytes = 256 + rnd() % 8192; break;
                    case 1: bytes = (size_t(512) << 10) * (1 + rnd() % 8); break;
                    case 2: bytes = (size_t(1) << 20) * (2 + rnd() % 56); break;
                    case 3: bytes = (size_t(1) << 20) * (64 + rnd() % 400); break;
                    case 4: bytes = (size_t(1) << 20) * (147); break;
                    default: bytes = (size_t(1) << 20) * (1024 + rnd() % 5000); break;
                }
                void* p = nullptr;
                try {
                    p = pool.alloc(bytes);
                } catch (const Error&) {
                    ++oom;          // the pretend device is full: legitimate when what is live plus this request exceeds it
                    if (want_live + bytes + (size_t(64) << 20) < device_bytes / 2) throw Error(FHELIN_ERR_INTERNAL, "pool selftest: out of memory at less than half the device");
                    continue;
                }
                const uintptr_t a = reinterpret_cast<uintptr_t>(p);
                auto nx = live.lower_bound(a);
                if (nx != live.end() && a + bytes > nx->first) throw Error(FHELIN_ERR_INTERNAL, "pool selftest: block overlaps its successor");
                if (nx != live.begin()) {
                    auto pv = std::prev(nx);
                    if (pv->first + pv->second > a) throw Error(FHELIN_ERR_INTERNAL, "pool selftest: block overlaps its predecessor");
                }
                bool inside = false;
                for (const auto& b : g_mock_blocks) inside = inside || (a >= b.first && a + bytes <= b.first + b.second);
                if (!inside) throw Error(FHELIN_ERR_INTERNAL, "pool selftest: block outside every slab");
                live[a] = bytes;
                order.push_back(a);
                want_live += bytes;
                peak_live = std::max(peak_live, want_live);
            } else {
                const size_t k = rnd() % order.size();
                const uintptr_t a = order[k];
                order[k] = order.back();
                order.pop_back();
                want_live -= live[a];
                live.erase(a);
                pool.free(reinterpret_cast<void*>(a));
            }
        }
        out[0] = peak_live;
        out[1] = pool.reserved_peak;
        out[2] = pool.malloc_calls;
        out[3] = oom;
        for (uintptr_t a : order) pool.free(reinterpret_cast<void*>(a));
        // everything free: every slab is ONE free range again (coalescing), and a trim hands all of it back
        out[4] = pool.free_ranges() == pool.slabs() && pool.bytes_live() == 0 ? 1 : 0;
        pool.trim();
        out[5] = pool.bytes_reserved() == 0 && g_mock_used == 0 ? 1 : 0;
    }
    FHELIN_CATCH
}

}  // extern "C"
