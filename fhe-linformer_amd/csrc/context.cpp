#include "context.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/random.h>
#include <time.h>
#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>

namespace fhelin {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(FHELIN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------- DevicePool
void* DevicePool::raw_malloc(size_t bytes) {
    if (backend_.malloc_fn) return backend_.malloc_fn(bytes);
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();  // the failure is handled by the caller; do not leave it for the next launch check
        return nullptr;
    }
    return p;
}
void DevicePool::raw_free(void* p) {
    if (backend_.free_fn) backend_.free_fn(p);
    else (void)hipFree(p);
}
DevicePool::~DevicePool() {
    for (auto& ln : lane_) {
        for (auto& pk : ln.parked)
            for (int k = 0; k < pk.n_ev; ++k) (void)hipEventDestroy(pk.ev[k]);
        for (auto& sl : ln.slabs)
            if (sl.base) raw_free(sl.base);
    }
    for (hipEvent_t e : spare_events_) (void)hipEventDestroy(e);
}
size_t DevicePool::slabs() const {
    size_t n = 0;
    for (const auto& ln : lane_)
        for (const auto& sl : ln.slabs) n += sl.base != nullptr;
    return n;
}
size_t DevicePool::free_ranges() const {
    size_t n = 0;
    for (const auto& ln : lane_) n += ln.by_size.size();
    return n;
}
void DevicePool::erase_size_entry(Lane& ln, int slab, size_t off, size_t len) {
    auto r = ln.by_size.equal_range(len);
    for (auto it = r.first; it != r.second; ++it)
        if (it->second.first == slab && it->second.second == off) {
            ln.by_size.erase(it);
            return;
        }
    throw Error(FHELIN_ERR_INTERNAL, "DevicePool: free list out of step");
}
// a range goes back into its slab's free list, merged with the free ranges it touches
void DevicePool::insert_free(Lane& ln, int slab, size_t off, size_t len) {
    Slab& sl = ln.slabs[slab];
    auto nxt = sl.free_at.lower_bound(off);
    if (nxt != sl.free_at.begin()) {
        auto prv = std::prev(nxt);
        if (prv->first + prv->second == off) {
            off = prv->first;
            len += prv->second;
            erase_size_entry(ln, slab, prv->first, prv->second);
            sl.free_at.erase(prv);
        }
    }
    if (nxt != sl.free_at.end() && off + len == nxt->first) {
        len += nxt->second;
        erase_size_entry(ln, slab, nxt->first, nxt->second);
        sl.free_at.erase(nxt);
    }
    sl.free_at[off] = len;
    ln.by_size.emplace(len, std::make_pair(slab, off));
}
// one more slab for this lane, large enough for `bytes`: geometric growth, so that a small context stays small and a large one
// reaches the slab size in a few steps
bool DevicePool::grow(Lane& ln, size_t bytes) {
    size_t want = std::max(bytes, std::min(SLAB_MAX, std::max(SLAB_MIN, reserved_)));
    want = (want + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    void* p = raw_malloc(want);
    if (!p && want > bytes) {   // not that much left: exactly what is asked for
        want = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
        p = raw_malloc(want);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!p) return false;
    malloc_calls += 1;
    malloc_bytes += want;
    malloc_ns += (u64)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
    reserved_ += want;
    if (reserved_ > reserved_peak) reserved_peak = reserved_;
    int idx = -1;
    for (size_t i = 0; i < ln.slabs.size(); ++i)
        if (!ln.slabs[i].base) idx = (int)i;      // a slot a trim has emptied
    if (idx < 0) {
        ln.slabs.emplace_back();
        idx = (int)ln.slabs.size() - 1;
    }
    Slab& sl = ln.slabs[idx];
    sl.base = static_cast<char*>(p);
    sl.size = want;
    sl.used = 0;
    sl.free_at.clear();
    insert_free(ln, idx, 0, want);
    return true;
}
void* DevicePool::alloc(size_t bytes) {
    if (bytes == 0) bytes = 256;
    // 256-byte granules for small blocks, 4 KiB above 64 KiB (fewer distinct fragment sizes)
    const size_t gran = bytes > (size_t(64) << 10) ? 4096 : 256;
    bytes = (bytes + gran - 1) & ~(gran - 1);
    Lane& ln = lane_[cur_lane];
    // ranges that were freed under another lane: this lane's stream goes behind that use (no host wait), then they are ordinary
    // free memory of this lane
    if (!ln.parked.empty()) {
        for (auto& pk : ln.parked) {
            for (int k = 0; k < pk.n_ev; ++k) {
                hip_check(hipStreamWaitEvent(lane_stream[cur_lane], pk.ev[k], 0), "hipStreamWaitEvent(pool reuse)");
                spare_events_.push_back(pk.ev[k]);
            }
            insert_free(ln, pk.slab, pk.off, pk.len);
        }
        ln.parked.clear();
    }
    auto it = ln.by_size.lower_bound(bytes);
    if (it == ln.by_size.end()) {
        if (!grow(ln, bytes)) {
            trim();                                            // out of memory: give back what holds nothing, then once more
            if (!grow(ln, bytes)) throw Error(FHELIN_ERR_HIP, "hipMalloc: out of device memory (" + std::to_string(bytes >> 20) + " MiB asked, " +
                                                                  std::to_string(reserved_ >> 20) + " MiB held, " + std::to_string(live_bytes_ >> 20) + " MiB in use)");
        }
        it = ln.by_size.lower_bound(bytes);
    }
    const int slab = it->second.first;
    const size_t off = it->second.second, len = it->first;
    Slab& sl = ln.slabs[slab];
    ln.by_size.erase(it);
    sl.free_at.erase(off);
    if (len > bytes) {
        sl.free_at[off + bytes] = len - bytes;
        ln.by_size.emplace(len - bytes, std::make_pair(slab, off + bytes));
    }
    sl.used += bytes;
    void* p = sl.base + off;
    live_[p] = Live{bytes, cur_lane, slab, off};
    live_bytes_ += bytes;
    if (live_bytes_ > live_peak) live_peak = live_bytes_;
    return p;
}
void DevicePool::free(void* p, unsigned used_by_lanes) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) throw Error(FHELIN_ERR_STATE, "DevicePool::free of unknown pointer");
    const Live lv = it->second;
    live_.erase(it);
    live_bytes_ -= lv.bytes;
    Lane& ln = lane_[lv.lane];
    ln.slabs[lv.slab].used -= lv.bytes;
    // the lanes whose streams may still have work queued that reads the block, other than the owner (whose own stream order covers it)
    unsigned wait = used_by_lanes;
    if (have_streams && lv.lane != cur_lane) {
        wait |= 1u << cur_lane;
        if (conservative_foreign_free)
            for (int k = 0; k <= n_user_lanes; ++k) wait |= 1u << k;
    }
    wait &= ~(1u << lv.lane);
    if (!have_streams || !wait) {
        insert_free(ln, lv.slab, lv.off, lv.bytes);
        return;
    }
    Parked pk{lv.slab, lv.off, lv.bytes, {}, 0};
    for (int k = 0; k < MAX_LANES; ++k) {
        if (!(wait & (1u << k)) || !lane_stream[k]) continue;
        hipEvent_t ev = nullptr;
        if (!spare_events_.empty()) {
            ev = spare_events_.back();
            spare_events_.pop_back();
        } else {
            hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate(pool)");
        }
        hip_check(hipEventRecord(ev, lane_stream[k]), "hipEventRecord(pool free)");
        pk.ev[pk.n_ev++] = ev;
    }
    foreign_frees += 1;
    ln.parked.push_back(pk);
}
void DevicePool::trim() {
    trims += 1;
    if (!backend_.malloc_fn) (void)hipDeviceSynchronize();
    for (auto& ln : lane_) {
        for (auto& pk : ln.parked) {            // the device is idle: parked ranges are plain free memory
            for (int k = 0; k < pk.n_ev; ++k) spare_events_.push_back(pk.ev[k]);
            insert_free(ln, pk.slab, pk.off, pk.len);
        }
        ln.parked.clear();
        for (size_t i = 0; i < ln.slabs.size(); ++i) {
            Slab& sl = ln.slabs[i];
            if (!sl.base || sl.used != 0) continue;
            erase_size_entry(ln, (int)i, 0, sl.size);   // an unused slab is one free range
            raw_free(sl.base);
            reserved_ -= sl.size;
            sl = Slab();
        }
    }
}

// ---------------------------------------------------------------- Context
template <class T>
const T* Context::upload_table(const std::vector<T>& v) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hip_check(hipMalloc(&p, bytes), "hipMalloc(table)");
    if (!v.empty()) hip_check(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(table)");
    table_allocs.push_back(p);
    return static_cast<const T*>(p);
}
template const u64* Context::upload_table<u64>(const std::vector<u64>&);
template const int* Context::upload_table<int>(const std::vector<int>&);
template const u32* Context::upload_table<u32>(const std::vector<u32>&);
template const double* Context::upload_table<double>(const std::vector<double>&);

static u64 mont_r(u64 m) { return (u64)((((u128)1) << 64) % m); }   // 2^64 mod m

static u64 prod_mod(const std::vector<u64>& ms, int skip, u64 t) {
    u64 r = 1 % t;
    for (int i = 0; i < (int)ms.size(); ++i)
        if (i != skip) r = h_mulmod(r, ms[i] % t, t);
    return r;
}

// n_p < 0: the number of special primes OpenFHE's HYBRID parameter generation picks — enough primes of special_bits
// to cover the widest digit:  sizeP = ceil(max_j bits(prod of digit j) / auxBits)  (29 limbs, dnum 4: 55 + 7*52 = 419 -> 7;
// the reference's 28 limbs: 55 + 6*52 = 367 -> 7)
static void resolve_seed(Params& p) {
    if (p.have_seed_bytes) return;
    if (p.seed != 0) {  // deterministic test seed: little-endian in the first 8 key bytes
        std::memset(p.seed_bytes, 0, sizeof p.seed_bytes);
        for (int i = 0; i < 8; ++i) p.seed_bytes[i] = (uint8_t)(p.seed >> (8 * i));
    } else {
        size_t got = 0;
        while (got < sizeof p.seed_bytes) {
            const ssize_t n = getrandom(p.seed_bytes + got, sizeof p.seed_bytes - got, 0);
            if (n <= 0) throw Error(FHELIN_ERR_INTERNAL, "getrandom failed: no entropy for key generation");
            got += (size_t)n;
        }
    }
    p.have_seed_bytes = true;
}

static Params resolve_params(Params p) {
    resolve_seed(p);
    if (p.n_p < 0 && p.n_q >= 1 && p.dnum >= 1 && p.special_bits >= 1) {
        const int a = (p.n_q + p.dnum - 1) / p.dnum;
        const int first = p.first_bits + (std::min(a, p.n_q) - 1) * p.scale_bits;   // the digit that holds q0
        const int other = p.n_q > a ? std::min(a, p.n_q - a) * p.scale_bits : 0;
        p.n_p = (std::max(first, other) + p.special_bits - 1) / p.special_bits;
    }
    return p;
}

Context::Context(const Params& p_in) : prm(resolve_params(p_in)) {
    const Params& p = prm;
    if (p.log_n < 12 || p.log_n > 17) throw Error(FHELIN_ERR_ARG, "log_n must be in [12,17]");
    if (p.n_q < 1 || p.n_q > 64 || p.n_p < 0 || p.n_p > 16) throw Error(FHELIN_ERR_ARG, "bad limb counts");
    if (p.dnum < 1) throw Error(FHELIN_ERR_ARG, "dnum must be >= 1");
    if (p.first_bits > 60 || p.scale_bits > 60 || p.special_bits > 60) throw Error(FHELIN_ERR_ARG, "primes must be <= 60 bits");
    if (p.first_bits < 20 || p.scale_bits < 20 || (p.n_p > 0 && p.special_bits < 20)) throw Error(FHELIN_ERR_ARG, "primes must be >= 20 bits");
    if (p.log_slots < 1 || p.log_slots > p.log_n - 1) throw Error(FHELIN_ERR_ARG, "log_slots out of range");
    N = 1 << p.log_n;
    L = p.n_q - 1;
    K = p.n_p;
    alpha = (p.n_q + p.dnum - 1) / p.dnum;
    if (alpha > 16) throw Error(FHELIN_ERR_ARG, "digit size > 16 limbs not supported");
    // the evaluation-key inner product sums beta <= dnum products in 128 bits before ONE Barrett reduction, which needs
    // beta * p^2 < p * 2^64 for the 60-bit special primes (kernels_ks.hip ks_inner_kernel)
    if ((p.n_q + alpha - 1) / alpha > 16) throw Error(FHELIN_ERR_ARG, "more than 16 key-switching digits not supported");
    try {
        chain = make_prime_chain(p.log_n, p.n_q, p.first_bits, p.scale_bits, p.n_p, p.special_bits);
    } catch (const std::exception& e) {
        throw Error(FHELIN_ERR_ARG, std::string("prime chain: ") + e.what());
    }
    moduli = chain.q;
    moduli.insert(moduli.end(), chain.p.begin(), chain.p.end());
    for (u64 q : moduli) {
        barrett.push_back(h_barrett(q));
        tw.push_back(make_twiddles(q, p.log_n));
    }
    sf_real.assign(p.n_q + 1, 0.0L);
    sf_real[0] = (long double)chain.q[L];
    for (int k = 0; k < L; ++k) sf_real[k + 1] = sf_real[k] * sf_real[k] / (long double)chain.q[L - k];

    if (p.device < 0) return;  // host-only: parameter layer only

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(FHELIN_ERR_NO_DEVICE, "no HIP device visible: the fhe-linformer_amd engine has no CPU fallback");
    if (p.device >= ndev) throw Error(FHELIN_ERR_ARG, "device index out of range");
    hip_check(hipSetDevice(p.device), "hipSetDevice");
    hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    main_stream = stream;
    own_stream = true;
    {
        int want = 2;
        if (const char* e = std::getenv("FHELIN_LANES")) want = std::atoi(e);
        n_lanes = std::max(0, std::min(want, DevicePool::MAX_LANES - 1));
        for (int k = 1; k <= n_lanes; ++k) {
            hip_check(hipStreamCreateWithFlags(&lane_stream[k], hipStreamNonBlocking), "hipStreamCreate(lane)");
            hip_check(hipEventCreateWithFlags(&lane_event[k], hipEventDisableTiming), "hipEventCreate(lane)");
        }
        hip_check(hipEventCreateWithFlags(&fork_event, hipEventDisableTiming), "hipEventCreate(fork)");
        pool.lane_stream[0] = main_stream;
        for (int k = 1; k <= n_lanes; ++k) pool.lane_stream[k] = lane_stream[k];
        pool.have_streams = true;
        if (const char* e = std::getenv("FHELIN_ASYNC")) async_lanes = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_FUSE_MODDOWN")) fuse_moddown = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_HOST_ENCODE")) host_encode = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_FUSE_GATHER")) fuse_gather = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_FUSE_LIFT")) fuse_lift = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_LDS_DIGITS")) lds_digits = std::atoi(e) != 0;
        if (const char* e = std::getenv("FHELIN_NTT_TRACE")) trace_small_ntt = std::atoi(e);
        if (const char* e = std::getenv("FHELIN_HOST_WAITS")) trace_waits = std::atoi(e) != 0;
    }
    hip_check(hipEventCreate(&ev_start), "hipEventCreate");
    stage_words = (size_t)2 << p.log_n;
    for (int i = 0; i < STAGE_SLOTS; ++i) {
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&stage_buf[i]), stage_words * sizeof(u64), hipHostMallocDefault), "hipHostMalloc(stage)");
        hip_check(hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming), "hipEventCreate(stage)");
    }
    hip_check(hipHostMalloc(reinterpret_cast<void**>(&small_buf), (size_t)SMALL_SLOTS * SMALL_WORDS * sizeof(u64), hipHostMallocDefault), "hipHostMalloc(stage, small)");
    for (int i = 0; i < SMALL_SLOTS; ++i) hip_check(hipEventCreateWithFlags(&small_ev[i], hipEventDisableTiming), "hipEventCreate(stage, small)");
    hip_check(hipEventCreate(&ev_stop), "hipEventCreate");
    has_device = true;

    const int nl = (int)moduli.size();
    {
        std::vector<u64> bar(2 * nl), ninv(8 * nl, 0);
        for (int i = 0; i < nl; ++i) {
            bar[2 * i] = barrett[i].r0;
            bar[2 * i + 1] = barrett[i].r1;
            ninv[8 * i] = tw[i].n_inv;
            ninv[8 * i + 1] = tw[i].n_inv_s;
            ninv[8 * i + 2] = tw[i].w1_n_inv;
            ninv[8 * i + 3] = tw[i].w1_n_inv_s;
            // NTT final reduction (modarith.h reduce_lazy_2q): shift = bits(q) - 10, ratio = floor(2^(bits(q)+22) / q)
            int bits = 0;
            while (bits < 64 && (moduli[i] >> bits)) ++bits;
            if (bits >= 12 && bits <= 60) {
                ninv[8 * i + 4] = (u64)(bits - 10);
                ninv[8 * i + 5] = (u64)((((u128)1) << (bits + 22)) / moduli[i]);
            }
        }
        dt.log_n = p.log_n;
        dt.n_limbs = nl;
        dt.moduli = upload_table(moduli);
        dt.barrett = upload_table(bar);
        {
            std::vector<u64> qi(nl);
            for (int i = 0; i < nl; ++i) {
                u64 x = moduli[i];                       // Newton: x <- x (2 - q x), correct bits double (q odd: q * q = 1 mod 8)
                for (int it = 0; it < 6; ++it) x *= 2 - moduli[i] * x;
                qi[i] = x;
            }
            dt.qinv = upload_table(qi);
            std::vector<u64> mr(2 * (size_t)nl);
            for (int i = 0; i < nl; ++i) {
                mr[2 * i] = mont_r(moduli[i]);
                mr[2 * i + 1] = h_shoup(mr[2 * i], moduli[i]);
            }
            dt.mont = upload_table(mr);
        }
        dt.ninv = upload_table(ninv);
        std::vector<u64> f((size_t)nl * 2 * N), g((size_t)nl * 2 * N);
        for (int i = 0; i < nl; ++i) {
            std::memcpy(&f[(size_t)i * 2 * N], tw[i].fwd.data(), sizeof(u64) * 2 * N);
            std::memcpy(&g[(size_t)i * 2 * N], tw[i].inv.data(), sizeof(u64) * 2 * N);
        }
        dt.tw_fwd = upload_table(f);
        dt.tw_inv = upload_table(g);
        // row-pass layout (kernels_ntt.hip load_round_tw_rows): slot s = (8 >> kb) - 1 + j of thread tau in tile T is
        // tw[m_kb + (((T << 8) | tau) << (3 - kb)) + j],  m_kb = N >> (kb + 1),  kb = 0..3 (global stage bit)
        const size_t tiles = N >> 12;
        const size_t per_limb = tiles * 15 * 256 * 2;
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<u64>& src = dir ? g : f;
            std::vector<u64> r((size_t)nl * per_limb);
            for (int i = 0; i < nl; ++i)
                for (size_t T = 0; T < tiles; ++T)
                    for (int kb = 0; kb < 4; ++kb)
                        for (int j = 0; j < (8 >> kb); ++j)
                            for (size_t tau = 0; tau < 256; ++tau) {
                                const size_t idx = (N >> (kb + 1)) + ((((T << 8) | tau)) << (3 - kb)) + j;
                                const size_t s_ = (size_t)(8 >> kb) - 1 + j;
                                const size_t o = (size_t)i * per_limb + ((T * 15 + s_) * 256 + tau) * 2;
                                r[o] = src[(size_t)i * 2 * N + 2 * idx];
                                r[o + 1] = src[(size_t)i * 2 * N + 2 * idx + 1];
                            }
            (dir ? dt.tw_rows_inv : dt.tw_rows_fwd) = upload_table(r);
        }
    }
    // ---- ModDown / rescale constants
    {
        std::vector<u64> phatinv(2 * std::max(K, 1)), phatmod((size_t)std::max(K, 1) * (L + 1)), pinv(2 * (L + 1));
        for (int j = 0; j < K; ++j) {
            u64 pj = chain.p[j];
            u64 hat = prod_mod(chain.p, j, pj);
            u64 inv = h_invmod(hat, pj);
            phatinv[2 * j] = inv;
            phatinv[2 * j + 1] = h_shoup(inv, pj);
            // times 2^64 mod q_t: the conversion kernels finish their sums with a Montgomery reduction (modarith.h redc128)
            for (int t = 0; t <= L; ++t) phatmod[(size_t)j * (L + 1) + t] = pack30(h_mulmod(prod_mod(chain.p, j, chain.q[t]), mont_r(chain.q[t]), chain.q[t]));  // 30-bit halves
        }
        for (int t = 0; t <= L; ++t) {
            u64 qt = chain.q[t];
            u64 pm = prod_mod(chain.p, -1, qt);
            u64 inv = K > 0 ? h_invmod(pm, qt) : 1;
            pinv[2 * t] = inv;
            pinv[2 * t + 1] = h_shoup(inv, qt);
        }
        d_phatinv = upload_table(phatinv);
        d_phatmod = upload_table(phatmod);
        d_pinv = upload_table(pinv);
        std::vector<u64> pmod(2 * (L + 1));
        for (int t = 0; t <= L; ++t) {
            pmod[2 * t] = prod_mod(chain.p, -1, chain.q[t]);
            pmod[2 * t + 1] = h_shoup(pmod[2 * t], chain.q[t]);
        }
        d_pmod = upload_table(pmod);
        std::vector<u64> qlinv((size_t)(L + 1) * (L + 1) * 2, 0), qlmod((size_t)(L + 1) * (L + 1), 0);
        for (int l = 0; l <= L; ++l)
            for (int t = 0; t < l; ++t) {
                u64 qt = chain.q[t];
                u64 r = chain.q[l] % qt;
                u64 inv = h_invmod(r, qt);
                qlmod[(size_t)l * (L + 1) + t] = r;
                qlinv[((size_t)l * (L + 1) + t) * 2] = inv;
                qlinv[((size_t)l * (L + 1) + t) * 2 + 1] = h_shoup(inv, qt);
            }
        d_qlinv = upload_table(qlinv);
        d_qlmod = upload_table(qlmod);
    }
    // ---- ModUp constants per level
    lvl.resize(L + 2);
    for (int ell = 1; ell <= L + 1; ++ell) {
        LevelTables& lt = lvl[ell];
        lt.ell = ell;
        lt.beta = digits_at(ell);
        const int nt = ell + K;
        std::vector<u64> hatinv(2 * ell), hatmod((size_t)ell * nt, 0), hatmod2((size_t)ell * nt, 0);
        std::vector<int> tab((size_t)lt.beta * nt, -1);
        for (int j = 0; j < lt.beta; ++j) {
            const int lo = j * alpha, hi = std::min((j + 1) * alpha, ell);
            std::vector<u64> dig(chain.q.begin() + lo, chain.q.begin() + hi);
            for (int i = lo; i < hi; ++i) {
                u64 qi = chain.q[i];
                u64 inv = h_invmod(prod_mod(dig, i - lo, qi), qi);
                hatinv[2 * i] = inv;
                hatinv[2 * i + 1] = h_shoup(inv, qi);
                for (int t = 0; t < nt; ++t) {
                    u64 mt = t < ell ? chain.q[t] : chain.p[t - ell];
                    const u64 hr = h_mulmod(prod_mod(dig, i - lo, mt), mont_r(mt), mt);
                    hatmod[(size_t)i * nt + t] = pack30(hr);  // times 2^64 mod m_t (redc128); 30-bit halves for mac30
                    hatmod2[(size_t)i * nt + t] = pack30(h_mulmod(hr, mont_r(mt), mt));
                }
            }
            for (int t = 0; t < nt; ++t) {
                bool own = t >= lo && t < hi;
                tab[(size_t)j * nt + t] = own ? -1 : (t < ell ? limb_id_q(t) : limb_id_p(t - ell));
            }
        }
        lt.up_hatinv = upload_table(hatinv);
        lt.up_hatmod = upload_table(hatmod);
        lt.up_hatmod_r2 = upload_table(hatmod2);
        lt.ext_limb_tab = upload_table(tab);
        if (ell >= 2 && K >= 1) {   // ModDown + rescale in one conversion: drop B = (p_0..p_{k-1}, q_{ell-1})
            const u64 ql = chain.q[ell - 1];
            const int e1 = ell - 1;
            std::vector<u64> mhi(2 * (K + 1)), mhm((size_t)(K + 1) * e1), minv(2 * e1), mmod(e1);
            for (int j = 0; j <= K; ++j) {
                const u64 b = j < K ? chain.p[j] : ql;
                const u64 hat_b = j < K ? h_mulmod(prod_mod(chain.p, j, b), ql % b, b) : prod_mod(chain.p, -1, b);
                const u64 inv = h_invmod(hat_b, b);
                mhi[2 * j] = inv;
                mhi[2 * j + 1] = h_shoup(inv, b);
                for (int t = 0; t < e1; ++t) {
                    const u64 qt = chain.q[t];
                    mhm[(size_t)j * e1 + t] = pack30(j < K ? h_mulmod(prod_mod(chain.p, j, qt), ql % qt, qt) : prod_mod(chain.p, -1, qt));
                }
            }
            for (int t = 0; t < e1; ++t) {
                const u64 qt = chain.q[t];
                mmod[t] = h_mulmod(prod_mod(chain.p, -1, qt), ql % qt, qt);
                const u64 inv = h_invmod(mmod[t], qt);
                minv[2 * t] = inv;
                minv[2 * t + 1] = h_shoup(inv, qt);
            }
            lt.md_hatinv = upload_table(mhi);
            lt.md_hatmod = upload_table(mhm);
            lt.md_minv = upload_table(minv);
            lt.md_mmod = upload_table(mmod);
        }
    }
}

void Context::upload_async(u64* dst, const u64* src, size_t words) {
    if (words <= SMALL_WORDS && small_buf) {
        const int i = small_next;
        small_next = (small_next + 1) % SMALL_SLOTS;
        if (small_used[i]) {
            timespec a, b;
            if (trace_waits) clock_gettime(CLOCK_MONOTONIC, &a);
            hip_check(hipEventSynchronize(small_ev[i]), "hipEventSynchronize(stage, small)");
            if (trace_waits) {
                clock_gettime(CLOCK_MONOTONIC, &b);
                note_wait("upload ring (small)", (u64)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)));
            }
        }
        u64* slot = small_buf + (size_t)i * SMALL_WORDS;
        std::memcpy(slot, src, words * sizeof(u64));
        hip_check(hipMemcpyAsync(dst, slot, words * sizeof(u64), hipMemcpyHostToDevice, stream), "hipMemcpyAsync(stage, small)");
        hip_check(hipEventRecord(small_ev[i], stream), "hipEventRecord(stage, small)");
        small_used[i] = true;
        return;
    }
    if (words > stage_words) throw Error(FHELIN_ERR_INTERNAL, "upload_async: block larger than a staging slot");
    const int i = stage_next;
    stage_next = (stage_next + 1) % STAGE_SLOTS;
    if (stage_used[i]) {
        timespec a, b;
        if (trace_waits) clock_gettime(CLOCK_MONOTONIC, &a);
        hip_check(hipEventSynchronize(stage_ev[i]), "hipEventSynchronize(stage)");
        if (trace_waits) {
            clock_gettime(CLOCK_MONOTONIC, &b);
            note_wait("upload ring (large)", (u64)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)));
        }
    }
    std::memcpy(stage_buf[i], src, words * sizeof(u64));
    hip_check(hipMemcpyAsync(dst, stage_buf[i], words * sizeof(u64), hipMemcpyHostToDevice, stream), "hipMemcpyAsync(stage)");
    hip_check(hipEventRecord(stage_ev[i], stream), "hipEventRecord(stage)");
    stage_used[i] = true;
}

Context::~Context() {
    if (trace_waits) {
        fprintf(stderr, "[fhelin] host time blocked inside the library, by site (waits, ms):\n");
        for (const auto& e : wait_sites) fprintf(stderr, "  %8llu %10.1f  %s\n", (unsigned long long)e.second.first, e.second.second / 1e6, e.first.c_str());
    }
    if (trace_small_ntt && !small_ntt_sites.empty()) {
        std::vector<std::pair<std::string, std::pair<u64, u64>>> v(small_ntt_sites.begin(), small_ntt_sites.end());
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.first > b.second.first; });
        fprintf(stderr, "[fhelin] transforms of <= %d limb vectors by call stack (launch pairs, limb vectors):\n", trace_small_ntt);
        for (size_t i = 0; i < v.size() && i < 40; ++i)
            fprintf(stderr, "  %8llu %9llu  %s\n", (unsigned long long)v[i].second.first, (unsigned long long)v[i].second.second, v[i].first.c_str());
    }
    if (has_device) {
        (void)hipSetDevice(prm.device);
        (void)hipDeviceSynchronize();
        for (void* p : table_allocs) (void)hipFree(p);
        for (int k = 1; k <= n_lanes; ++k) {
            if (lane_event[k]) (void)hipEventDestroy(lane_event[k]);
            if (lane_stream[k]) (void)hipStreamDestroy(lane_stream[k]);
        }
        if (fork_event) (void)hipEventDestroy(fork_event);
        for (int i = 0; i < STAGE_SLOTS; ++i) {
            if (stage_ev[i]) (void)hipEventDestroy(stage_ev[i]);
            if (stage_buf[i]) (void)hipHostFree(stage_buf[i]);
        }
        for (int i = 0; i < SMALL_SLOTS; ++i)
            if (small_ev[i]) (void)hipEventDestroy(small_ev[i]);
        if (small_buf) (void)hipHostFree(small_buf);
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_stop) (void)hipEventDestroy(ev_stop);
        if (own_stream && main_stream) (void)hipStreamDestroy(main_stream);
    }
}

void Context::require_device() const {
    if (!has_device)
        throw Error(FHELIN_ERR_NO_DEVICE, "operation needs the HIP device path (host-only context): no CPU fallback exists");
}

void Context::sync() {
    require_device();
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (stream == main_stream) {  // asynchronous lane work counts as issued work of the context
        for (int k = 1; k <= n_lanes; ++k) {
            hip_check(hipStreamSynchronize(lane_stream[k]), "hipStreamSynchronize(lane)");
            lane_hold[k].clear();
        }
    }
}

void Context::note_small_ntt(int nvec) {
    void* fr[12];
    const int n = backtrace(fr, 12);
    std::string key;
    for (int i = 2; i < n && i < 9; ++i) {
        Dl_info info;
        std::string name = "?";
        if (dladdr(fr[i], &info) && info.dli_sname) {
            int st = 0;
            char* d = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &st);
            name = (st == 0 && d) ? d : info.dli_sname;
            if (d) std::free(d);
            const size_t par = name.find('(');
            if (par != std::string::npos) name.resize(par);
        }
        key += (i > 2 ? " < " : "") + name;
    }
    auto& e = small_ntt_sites[key];
    e.first += 1;
    e.second += (u64)nvec;
}

void Context::release_holds(int lane, u64 up_to_seq) {
    auto& h = lane_hold[lane];
    h.erase(std::remove_if(h.begin(), h.end(), [&](const std::pair<u64, std::shared_ptr<void>>& e) { return e.first <= up_to_seq; }),
            h.end());
}

void Context::fork_lanes() {
    hip_check(hipEventRecord(fork_event, main_stream), "hipEventRecord(fork)");
    for (int k = 1; k <= n_lanes; ++k) hip_check(hipStreamWaitEvent(lane_stream[k], fork_event, 0), "hipStreamWaitEvent(fork)");
}

void Context::join_lanes() {
    for (int k = 1; k <= n_lanes; ++k) {
        hip_check(hipEventRecord(lane_event[k], lane_stream[k]), "hipEventRecord(lane)");
        hip_check(hipStreamWaitEvent(main_stream, lane_event[k], 0), "hipStreamWaitEvent(join)");
    }
}

u64 Context::galois_element(int r) const {
    const u64 M = 2ull * N;
    const u64 order = N / 2;  // order of 5 in Z_{2N}^*
    long rr = r % (long)order;
    if (rr < 0) rr += order;
    u64 g = 1, b = 5;
    u64 e = (u64)rr;
    while (e) {
        if (e & 1) g = (g * b) % M;
        b = (b * b) % M;
        e >>= 1;
    }
    return g;
}

static std::vector<u32> build_automorph_map(u64 g, size_t N, int ln) {
    const u64 M = 2ull * N;
    std::vector<u32> m(N);
    for (u32 j = 0; j < (u32)N; ++j) {
        u64 e = (2ull * bitrev32(j, ln) + 1) * g % M;  // odd exponent of the evaluation point that feeds slot j
        m[j] = bitrev32((u32)((e - 1) >> 1), ln);
    }
    return m;
}

const u32* Context::automorph_map(u64 g) {
    require_device();
    auto it = automorph_maps.find(g);
    if (it != automorph_maps.end()) return it->second;
    const u64 M = 2ull * N;
    u64 gi = g;  // g^-1 mod 2N (g odd, 2N a power of two): Newton steps double the number of correct bits
    for (int i = 0; i < 6; ++i) gi = gi * (2 - g * gi % M + M) % M;
    const u32* d = upload_table(build_automorph_map(g, N, prm.log_n));
    automorph_maps[g] = d;
    const u32* di = d;
    if (gi != g) {
        auto iit = automorph_maps.find(gi);
        di = iit != automorph_maps.end() ? iit->second : upload_table(build_automorph_map(gi, N, prm.log_n));
        automorph_maps[gi] = di;
    }
    automorph_inverse[d] = di;
    automorph_inverse[di] = d;
    return d;
}

const u32* Context::automorph_inverse_of(const u32* map) {
    if (!map) return nullptr;
    auto it = automorph_inverse.find(map);
    if (it == automorph_inverse.end()) throw Error(FHELIN_ERR_INTERNAL, "automorphism map without a registered inverse");
    return it->second;
}

}  // namespace fhelin
