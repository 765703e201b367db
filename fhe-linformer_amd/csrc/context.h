// Engine context: CKKS parameter set -> prime chain -> host/device tables, HIP stream, device pool.
// Mirrors what reference FHEController::generate_context builds through OpenFHE's GenCryptoContext
// (reference src/FHEController.cpp:3-49): ring dimension, 55/52-bit Q chain, HYBRID key switching with
// dnum large digits and 60-bit special primes, SPARSE_TERNARY secret, FLEXIBLEAUTO real scaling factors.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "../../include/fhelin.h"  // status codes
#include "hostmath.h"
#include "kernels.h"

namespace fhelin {

struct Params {
    int log_n = 16;
    int n_q = 24;          // L+1 Q limbs
    int first_bits = 55;
    int scale_bits = 52;
    int n_p = 6;           // special limbs k
    int special_bits = 60;
    int dnum = 4;
    int log_slots = 14;
    int hamming = 192;     // sparse ternary secret weight
    // secret seed of the client-side generator (client.h Prng).  seed == 0: 32 bytes of OS entropy (getrandom) — the
    // default and the only secure choice; seed != 0: a deterministic 64-bit TEST seed (tests, reproducible benchmarks);
    // seed_bytes set explicitly (have_seed_bytes): a stored 256-bit seed (load_context regenerating a client's keys)
    uint64_t seed = 0;
    uint8_t seed_bytes[32] = {};
    bool have_seed_bytes = false;
    int device = 0;        // < 0: host-only context (parameter tables only; every device op fails)
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char* what);

// Device memory arena (sized for the 288 GB of an MI355X: the key set, the resident inputs and the intermediates of a batched pass
// share it).  Memory comes from the driver in SLABS (few, large hipMalloc calls: geometric growth up to 8 GiB a slab, or the
// request itself when that is larger) and is handed out by BEST FIT from an address-ordered free list with coalescing: a freed
// block merges with its free neighbours, so the ciphertext blocks of one stage (say 6 GB of unwrapped rows) are carved up again by
// the next one, whatever its sizes.  (Rounds 1-3 cached freed blocks by exact size class: with many samples per pass the idle
// lists held 1.6 x the bytes in use and the device ran out of memory - every time at the price of a device-wide synchronisation.)
// Work is stream-ordered, so a freed block can be handed out again immediately - to the SAME stream.  With several lanes (streams)
// every lane has its own slabs and free list; a block that is freed while ANOTHER lane is current (a worker lane's result consumed
// and released on the main stream, or the reverse) may still be read by work queued on that other lane's stream: the free records
// an event there, and the owner lane's stream waits for it before the range goes back into the owner's free list.
// The backend (hipMalloc / hipFree) is replaceable: the allocator's logic is unit-tested on the host (fhelin_debug_pool_selftest).
class DevicePool {
public:
    static constexpr int MAX_LANES = 5;  // lane 0 = main stream
    static constexpr size_t SLAB_MAX = size_t(8) << 30, SLAB_MIN = size_t(32) << 20;
    struct Backend {
        void* (*malloc_fn)(size_t) = nullptr;   // null: hipMalloc / hipFree
        void (*free_fn)(void*) = nullptr;
    };
    DevicePool() = default;
    explicit DevicePool(const Backend& b) : backend_(b) {}
    ~DevicePool();
    void* alloc(size_t bytes);
    // used_by_lanes: lanes (bit k) whose queued work may still read the block besides the current one (a plaintext encoding shared by
    // the lanes of a pass): the owner lane reuses the range only behind all of them
    void free(void* p, unsigned used_by_lanes = 0);
    // between fhelin_ctx_lanes_fork and _join the caller's lanes run side by side and a handle may be dropped (a host language's garbage
    // collection) while ANY lane is current: a free under a foreign lane then waits for every lane, not only the current one
    bool conservative_foreign_free = false;
    int n_user_lanes = 0;
    void trim();                                // hand slabs that hold nothing back to the driver (synchronises the device first)
    size_t bytes_reserved() const { return reserved_; }
    size_t bytes_live() const { return live_bytes_; }
    size_t slabs() const;
    size_t free_ranges() const;
    int cur_lane = 0;
    hipStream_t lane_stream[MAX_LANES] = {};   // set by the Context: the stream each lane launches on (lane 0 = main stream)
    bool have_streams = false;
    // growth diagnostics (fhelin_stats slots 9..11): slabs obtained from hipMalloc, their bytes, host time spent inside hipMalloc
    u64 malloc_calls = 0, malloc_bytes = 0, malloc_ns = 0;
    u64 foreign_frees = 0;
    u64 trims = 0;                              // out-of-memory events: empty slabs handed back (a device-wide synchronisation each)
    size_t live_peak = 0, reserved_peak = 0;    // high-water marks: bytes in use, bytes held from the driver
private:
    struct Slab {
        char* base = nullptr;
        size_t size = 0, used = 0;
        std::map<size_t, size_t> free_at;       // offset -> length of the free ranges, address-ordered
    };
    struct Live { size_t bytes; int lane; int slab; size_t off; };
    struct Parked { int slab; size_t off, len; hipEvent_t ev[MAX_LANES]; int n_ev; };   // freed while other lanes may read it: waits for their events
    struct Lane {
        std::vector<Slab> slabs;
        std::multimap<size_t, std::pair<int, size_t>> by_size;      // length -> (slab, offset): best fit
        std::vector<Parked> parked;
    };
    Backend backend_;
    Lane lane_[MAX_LANES];
    std::unordered_map<void*, Live> live_;
    std::vector<hipEvent_t> spare_events_;
    size_t reserved_ = 0, live_bytes_ = 0;
    void insert_free(Lane& ln, int slab, size_t off, size_t len);
    void erase_size_entry(Lane& ln, int slab, size_t off, size_t len);
    bool grow(Lane& ln, size_t bytes);
    void* raw_malloc(size_t bytes);
    void raw_free(void* p);
};

// Per-level constants of hybrid key switching / rescale, resident on the device.
struct LevelTables {
    int ell = 0;        // live Q limbs
    int beta = 0;       // digits present at this level
    // ModUp: per source limb i < ell: (Qhat_i^{-1} mod q_i, shoup); and [ell][ell+k] Qhat_i mod t
    const u64* up_hatinv = nullptr;   // [ell][2]
    const u64* up_hatmod = nullptr;   // [ell][ell+k]  (Q_j/q_i) mod m_t times 2^64 (the conversion ends in redc128), pre-split (pack30)
    const u64* up_hatmod_r2 = nullptr;   // the same times 2^128: digits that come out times 2^64, for the inner product over the keys as they are
                                         // stored (ks_inner: it ends in redc128 too and has no scaled key copy to read)
    const int* ext_limb_tab = nullptr;  // [beta*(ell+k)] limb id for the NTT after ModUp, -1 on own-digit slots
    // ModDown and rescale as ONE basis conversion (kernels_elem.h launch_moddown_rescale_conv; ell >= 2): the dropped basis is
    // B = (p_0..p_{k-1}, q_{ell-1}) with product M = P q_{ell-1}
    const u64* md_hatinv = nullptr;   // [k+1][2]      (M/b)^{-1} mod b, shoup
    const u64* md_hatmod = nullptr;   // [k+1][ell-1]  (M/b) mod q_t, pre-split (pack30)
    const u64* md_minv = nullptr;     // [ell-1][2]    M^{-1} mod q_t, shoup
    const u64* md_mmod = nullptr;     // [ell-1]       M mod q_t (the centred conversion takes one off per source above b/2)
};

// host-side operation counters (bench.py scales the CPU baseline sample with these)
struct OpStats {
    u64 limb_ntt = 0;      // limb vectors transformed (forward + inverse)
    u64 keyswitch = 0;     // hybrid key switches (rotations + relinearisations)
    u64 keyswitch_limbs = 0;  // sum of live Q limbs over those key switches
    u64 rescale = 0;
    u64 ct_pt_mult = 0;
    u64 bootstrap = 0;
    u64 encode = 0;
    u64 rescale_limbs = 0;    // sum of live Q limbs (before the drop) over those rescales
    u64 ct_pt_limbs = 0;      // sum of live Q limbs over those products
};

struct Context {
    Params prm;
    OpStats stats;
    int N = 0;
    int L = 0;       // n_q - 1
    int K = 0;       // n_p
    int alpha = 0;   // limbs per digit
    PrimeChain chain;
    std::vector<u64> moduli;          // Q then P
    std::vector<Barrett> barrett;     // same order
    std::vector<TwiddleTable> tw;     // same order (host copies)
    std::vector<long double> sf_real; // FLEXIBLEAUTO real scaling factor per level (0 = fresh)

    bool has_device = false;
    hipStream_t stream = nullptr;   // the CURRENT stream: main stream, or a lane's stream inside a LaneScope
    hipStream_t main_stream = nullptr;
    bool own_stream = false;
    // worker lanes: independent chunks of a batched op (rows of a matmul) run on different streams so that the
    // VALU-bound NTT passes of one chunk overlap the bandwidth-bound conversion / inner-product kernels of another
    int n_lanes = 0;
    hipStream_t lane_stream[DevicePool::MAX_LANES] = {};
    hipEvent_t lane_event[DevicePool::MAX_LANES] = {};
    hipEvent_t fork_event = nullptr;
    // Heavy single-ciphertext ops (bootstrap, polynomial evaluation) issued back to back by the caller run on
    // alternating lanes WITHOUT joining: the result carries an event, the main stream waits for it only when the result
    // is consumed.  An op's input is kept alive until the main stream has passed the op's event (lane_hold).
    bool async_lanes = true;        // FHELIN_ASYNC=0 disables
    int async_rr = 0;
    u64 lane_seq[DevicePool::MAX_LANES] = {};
    std::vector<std::pair<u64, std::shared_ptr<void>>> lane_hold[DevicePool::MAX_LANES];
    void release_holds(int lane, u64 up_to_seq);
    void fork_lanes();              // lanes wait for everything enqueued on the main stream so far
    void join_lanes();              // the main stream waits for every lane
    struct LaneScope {              // route launches and allocations to lane k (1-based) until destruction
        Context& c;
        LaneScope(Context& ctx, int k) : c(ctx) {
            c.stream = c.lane_stream[k];
            c.pool.cur_lane = k;
        }
        ~LaneScope() {
            c.stream = c.main_stream;
            c.pool.cur_lane = 0;
        }
    };
    DevicePool pool;
    DeviceTables dt{};
    std::vector<void*> table_allocs;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;

    // Pinned staging ring for host->device uploads on the hot path (plaintext encodings, encryption randomness):
    // the copy is queued on the current stream and the host does NOT wait for the stream to drain.  A slot is reused
    // only after the event recorded behind its copy has completed.
    static constexpr int STAGE_SLOTS = 64;   // 64 x 2N words pinned (64 MiB at N=2^16): with 8 the host ran at most eight plaintext encodings ahead of the GPU
    u64* stage_buf[STAGE_SLOTS] = {};
    hipEvent_t stage_ev[STAGE_SLOTS] = {};
    bool stage_used[STAGE_SLOTS] = {};
    int stage_next = 0;
    size_t stage_words = 0;
    // small payloads (the scalar tables of a linear combination: a few KB, hundreds per pass) have a ring of their own with MANY slots:
    // with the eight large slots alone the host could run at most eight uploads ahead of the GPU - harmless on one stream, but a host
    // that waits for lane 1's old upload cannot feed lane 2 (fhelin_ctx_set_lane)
    static constexpr int SMALL_SLOTS = 256;
    static constexpr size_t SMALL_WORDS = 8192;      // 64 KiB
    u64* small_buf = nullptr;                        // SMALL_SLOTS x SMALL_WORDS, pinned
    hipEvent_t small_ev[SMALL_SLOTS] = {};
    bool small_used[SMALL_SLOTS] = {};
    int small_next = 0;
    void upload_async(u64* dst, const u64* src, size_t words);  // words <= 2N

    // ModDown / rescale constants (level independent)
    const u64* d_phatinv = nullptr;  // [k][2]   (P/p)^{-1} mod p, shoup
    const u64* d_phatmod = nullptr;  // [k][L+1] (P/p) mod q_t
    const u64* d_pinv = nullptr;     // [L+1][2] P^{-1} mod q_t, shoup
    const u64* d_pmod = nullptr;     // [L+1][2] P mod q_t, shoup
    const u64* d_qlinv = nullptr;    // [L+1][L+1][2]  q_l^{-1} mod q_t, shoup   (row l, col t<l)
    const u64* d_qlmod = nullptr;    // [L+1][L+1]     q_l mod q_t
    std::vector<LevelTables> lvl;    // index by ell (1..L+1)
    std::map<u64, const u32*> automorph_maps;  // galois element -> device map [N]

    explicit Context(const Params& p);
    ~Context();
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;

    void require_device() const;
    template <class T> T* dalloc(size_t count) { return static_cast<T*>(pool.alloc(count * sizeof(T))); }
    template <class T> const T* upload_table(const std::vector<T>& v);
    int limb_id_q(int i) const { return i; }
    int limb_id_p(int j) const { return L + 1 + j; }
    int digits_at(int ell) const { return (ell + alpha - 1) / alpha; }
    u64 galois_element(int rot_index) const;       // 5^r mod 2N (r may be negative)
    const u32* automorph_map(u64 galois);          // device map for the NTT-domain permutation
    const u32* automorph_inverse_of(const u32* map);  // the map of the inverse automorphism (built together with `map`)
    std::map<const u32*, const u32*> automorph_inverse;
    // FHELIN_FUSE_MODDOWN=1: K8b as an epilogue of NTT(conv)'s row pass instead of its own kernel.  Bit-identical, one
    // memory pass fewer — and measured SLOWER end to end (1615 vs 1574 ms/sample): the epilogue's ~25 instructions per
    // residue land in the VALU-bound NTT, while the separate kernel is bandwidth-bound and overlaps with it.  Off.
    bool fuse_moddown = false;
    // FHELIN_HOST_ENCODE=1 / fhelin_ctx_set_host_encode: the special FFT of CKKS encoding on the host (the original path, kept
    // as the reference the device encoder is compared with bit for bit); default: on the GPU (kernels_client.hip)
    bool host_encode = false;
    // FHELIN_FUSE_GATHER=0: the rotated c0 parts of a merged rotation sum go through their own gather-and-sum kernel instead of
    // the ModDown epilogue (bit-identical; kept for A/B measurements)
    bool fuse_gather = true;
    bool fuse_lift = true;      // rescale: centred lift formed in the NTT's first-pass load (FHELIN_FUSE_LIFT)
    bool lds_digits = true;     // merged rotate-and-sum: digit tiles staged once in LDS when every rotation keeps tiles in place (FHELIN_LDS_DIGITS)
    struct FftDev {
        const u32* rot = nullptr;     // [slots]      5^j mod 4 slots
        const double* ksi = nullptr;  // [4 slots + 1][2]
    };
    std::map<int, FftDev> fft_dev;    // by slot count
    void sync();
    // K1 launch + accounting (active = vectors actually transformed, for tables with skipped entries)
    void ntt(const LimbBatch& b, bool inverse, int active = -1) {
        stats.limb_ntt += (u64)(active >= 0 ? active : b.nvec);
        if (trace_small_ntt && b.nvec <= trace_small_ntt) note_small_ntt(b.nvec);
        launch_ntt(dt, b, inverse, stream);
    }
    // FHELIN_NTT_TRACE=<n>: transforms of at most n limb vectors (launches that cannot fill the GPU) are counted by call stack and the
    // table is printed when the context goes away (a development aid: which caller still issues single-ciphertext launches)
    // FHELIN_HOST_WAITS=1: host time spent blocked inside the library, by site (printed when the context goes away)
    bool trace_waits = false;
    std::map<std::string, std::pair<u64, u64>> wait_sites;   // site -> (calls that waited, nanoseconds)
    void note_wait(const char* site, u64 ns) {
        auto& e = wait_sites[site];
        e.first += 1;
        e.second += ns;
    }
    int trace_small_ntt = 0;
    std::map<std::string, std::pair<u64, u64>> small_ntt_sites;   // call stack -> (launches, limb vectors)
    void note_small_ntt(int nvec);
    void ntt_moddown(const LimbBatch& b, const NttModDown& md) {
        stats.limb_ntt += (u64)b.nvec;
        launch_ntt_moddown(dt, b, md, stream);
    }
};

}  // namespace fhelin
