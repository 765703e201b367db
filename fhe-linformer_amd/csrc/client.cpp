#include "client.h"
#include <time.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <map>
#include <mutex>

namespace fhelin {

// ------------------------------------------------------------------------------------------------ PRNG
// ChaCha20 over GCC vector types: lane j of every state word belongs to block (counter + j), so one pass of the
// 20 rounds yields LANES blocks (lowered to SSE2/AVX2 by the compiler; no intrinsics).
namespace {
typedef u32 vw __attribute__((vector_size(4 * Prng::LANES)));
#define FHELIN_ROTL(x, k) (((x) << (k)) | ((x) >> (32 - (k))))
inline void quarter(vw& a, vw& b, vw& c, vw& d) {  // vectors travel by reference only (no vector ABI involved)
    a += b; d ^= a; d = FHELIN_ROTL(d, 16);
    c += d; b ^= c; b = FHELIN_ROTL(b, 12);
    a += b; d ^= a; d = FHELIN_ROTL(d, 8);
    c += d; b ^= c; b = FHELIN_ROTL(b, 7);
}
void chacha20_blocks(const u32 key[8], u64 counter, u64 stream, u32* out /* [LANES][16] */) {
    static const u32 sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};  // "expand 32-byte k"
    vw init[16], x[16];
    for (int i = 0; i < 4; ++i) init[i] = vw{} + sigma[i];
    for (int i = 0; i < 8; ++i) init[4 + i] = vw{} + key[i];
    for (int j = 0; j < Prng::LANES; ++j) {
        const u64 c = counter + (u64)j;
        init[12][j] = (u32)c;
        init[13][j] = (u32)(c >> 32);
    }
    init[14] = vw{} + (u32)stream;
    init[15] = vw{} + (u32)(stream >> 32);
    for (int i = 0; i < 16; ++i) x[i] = init[i];
    for (int r = 0; r < 10; ++r) {
        quarter(x[0], x[4], x[8], x[12]);
        quarter(x[1], x[5], x[9], x[13]);
        quarter(x[2], x[6], x[10], x[14]);
        quarter(x[3], x[7], x[11], x[15]);
        quarter(x[0], x[5], x[10], x[15]);
        quarter(x[1], x[6], x[11], x[12]);
        quarter(x[2], x[7], x[8], x[13]);
        quarter(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) {
        x[i] += init[i];
        for (int j = 0; j < Prng::LANES; ++j) out[j * 16 + i] = x[i][j];
    }
}
void load_key(const uint8_t seed[32], u32 key[8]) {
    for (int i = 0; i < 8; ++i)
        key[i] = (u32)seed[4 * i] | ((u32)seed[4 * i + 1] << 8) | ((u32)seed[4 * i + 2] << 16) | ((u32)seed[4 * i + 3] << 24);
}
}  // namespace

Prng::Prng(const uint8_t seed[32], u64 stream) : stream_(stream) { load_key(seed, key_); }

void Prng::refill() {
    chacha20_blocks(key_, counter_, stream_, buf_);
    counter_ += LANES;
    pos_ = 0;
}
void Prng::block(const uint8_t seed[32], u64 counter, u64 stream, uint8_t out[64]) {
    u32 key[8], buf[16 * LANES];
    load_key(seed, key);
    chacha20_blocks(key, counter, stream, buf);
    for (int i = 0; i < 16; ++i)
        for (int b = 0; b < 4; ++b) out[4 * i + b] = (uint8_t)(buf[i] >> (8 * b));
}
u64 Prng::next() {
    if (pos_ + 2 > 16 * LANES) refill();
    const u64 r = (u64)buf_[pos_] | ((u64)buf_[pos_ + 1] << 32);
    pos_ += 2;
    return r;
}
u64 Prng::uniform(u64 q) {  // multiply-shift with rejection of the biased low range (exactly uniform)
    const u64 thresh = (0 - q) % q;  // 2^64 mod q
    for (;;) {
        const u128 m = (u128)next() * q;
        if ((u64)m >= thresh) return (u64)(m >> 64);
    }
}
double Prng::normal() {
    if (have_spare) {
        have_spare = false;
        return spare;
    }
    double u1, u2;
    do {
        u1 = (next() >> 11) * (1.0 / 9007199254740992.0);
    } while (u1 <= 0.0);
    u2 = (next() >> 11) * (1.0 / 9007199254740992.0);
    const double r = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925 * u2;
    spare = r * std::sin(th);
    have_spare = true;
    return r * std::cos(th);
}

// ------------------------------------------------------------------------------------------------ special FFT
namespace {
struct FftTables {
    std::vector<u32> rot;                         // 5^j mod 4*slots
    std::vector<std::complex<double>> ksi;        // exp(2 pi i k / (4*slots)), k in [0, 4*slots]
};
std::map<int, FftTables> g_fft;
std::mutex g_fft_mu;
const FftTables& fft_tables(int slots) {
    std::lock_guard<std::mutex> lk(g_fft_mu);
    auto it = g_fft.find(slots);
    if (it != g_fft.end()) return it->second;
    FftTables t;
    const u32 m = 4u * slots;
    t.rot.resize(slots);
    u32 p = 1;
    for (int j = 0; j < slots; ++j) {
        t.rot[j] = p;
        p = (u32)(((u64)p * 5) % m);
    }
    t.ksi.resize(m + 1);
    for (u32 k = 0; k <= m; ++k) {
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)m;
        t.ksi[k] = {(double)cosl(a), (double)sinl(a)};
    }
    return g_fft.emplace(slots, std::move(t)).first->second;
}
void bit_reverse(std::vector<std::complex<double>>& v) {
    const size_t n = v.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j >= bit; bit >>= 1) j -= bit;
        j += bit;
        if (i < j) std::swap(v[i], v[j]);
    }
}
}  // namespace

// canonical embedding restricted to the rotation group <5>: forward = decode direction
void ckks_fft_special(std::vector<std::pair<double, double>>& pv, bool inverse) {
    const int size = (int)pv.size();
    const FftTables& t = fft_tables(size);
    const int m = 4 * size;
    std::vector<std::complex<double>> v(size);
    for (int i = 0; i < size; ++i) v[i] = {pv[i].first, pv[i].second};
    if (!inverse) {
        bit_reverse(v);
        for (int len = 2; len <= size; len <<= 1) {
            const int lenh = len >> 1, lenq = len << 2, gap = m / lenq;
            for (int i = 0; i < size; i += len)
                for (int j = 0; j < lenh; ++j) {
                    const int idx = (t.rot[j] % lenq) * gap;
                    const auto u = v[i + j], w = v[i + j + lenh] * t.ksi[idx];
                    v[i + j] = u + w;
                    v[i + j + lenh] = u - w;
                }
        }
    } else {
        for (int len = size; len >= 2; len >>= 1) {
            const int lenh = len >> 1, lenq = len << 2, gap = m / lenq;
            for (int i = 0; i < size; i += len)
                for (int j = 0; j < lenh; ++j) {
                    const int idx = (lenq - (t.rot[j] % lenq)) * gap;
                    const auto u = v[i + j] + v[i + j + lenh];
                    const auto w = (v[i + j] - v[i + j + lenh]) * t.ksi[idx];
                    v[i + j] = u;
                    v[i + j + lenh] = w;
                }
        }
        bit_reverse(v);
        const double inv = 1.0 / size;
        for (auto& x : v) x *= inv;
    }
    for (int i = 0; i < size; ++i) pv[i] = {v[i].real(), v[i].imag()};
}

void ckks_fft_tables(int slots, std::vector<u32>& rot, std::vector<std::pair<double, double>>& ksi) {
    const FftTables& t = fft_tables(slots);
    rot = t.rot;
    ksi.resize(t.ksi.size());
    for (size_t i = 0; i < t.ksi.size(); ++i) ksi[i] = {t.ksi[i].real(), t.ksi[i].imag()};
}

static void ld_to_i128(long double v, u64& lo, u64& hi) {
    if (v > -9.0e18L && v < 9.0e18L) {  // fits 64 bits: same rounding (half away from zero), no 128-bit split
        const long long r = llroundl(v);
        lo = (u64)r;
        hi = r < 0 ? ~0ull : 0;
        return;
    }
    const bool neg = v < 0;
    long double mag = roundl(fabsl(v));
    const long double two64 = 18446744073709551616.0L;
    u64 h = (u64)floorl(mag / two64);
    u64 l = (u64)(mag - (long double)h * two64);
    if (neg) {
        l = ~l + 1;
        h = ~h + (l == 0);
    }
    lo = l;
    hi = h;
}

static const Context::FftDev& fft_dev_tables(Context& c, int slots) {
    auto it = c.fft_dev.find(slots);
    if (it != c.fft_dev.end()) return it->second;
    const FftTables& t = fft_tables(slots);
    std::vector<double> k(2 * t.ksi.size());
    for (size_t i = 0; i < t.ksi.size(); ++i) {
        k[2 * i] = t.ksi[i].real();
        k[2 * i + 1] = t.ksi[i].imag();
    }
    Context::FftDev d;
    d.rot = c.upload_table(t.rot);
    d.ksi = c.upload_table(k);
    return c.fft_dev.emplace(slots, d).first->second;
}

void encode_batch_device(Context& c, u64* dst, const double* re, const double* im, int n_vec, int n_per, int slots, int ell, long double scale) {
    c.require_device();
    if (slots < 2 || (slots & (slots - 1)) || slots > c.N / 2) throw Error(FHELIN_ERR_ARG, "encode: slots must be a power of two in [2, N/2]");
    // ell = L + 1 + k: the encoding over the FULL key basis (limb ids 0..L+k: Q limbs, then the special limbs), for plaintexts
    // folded into rotation keys (Evaluator::folded_key); the leveled operations only ever ask for ell <= L + 1
    if (ell < 1 || ell > c.L + 1 + c.K) throw Error(FHELIN_ERR_ARG, "encode: level out of range");
    if (n_vec < 1) return;
    const size_t words = (size_t)2 * slots;                 // one complex vector, in doubles
    double* dv = reinterpret_cast<double*>(c.dalloc<u64>(words * n_vec));
    std::vector<u64> host(words);
    for (int b = 0; b < n_vec; ++b) {                       // through the pinned staging ring: no stream drain
        double* h = reinterpret_cast<double*>(host.data());
        for (int i = 0; i < slots; ++i) {
            h[2 * i] = i < n_per ? re[(size_t)b * n_per + i] : 0.0;
            h[2 * i + 1] = (im && i < n_per) ? im[(size_t)b * n_per + i] : 0.0;
        }
        c.upload_async(reinterpret_cast<u64*>(dv) + words * b, host.data(), words);
    }
    encode_complex_on_device(c, dst, dv, n_vec, slots, ell, scale);
    c.pool.free(dv);
}

// dv [n_vec][slots][2] complex slot values already on the device (overwritten) -> dst [n_vec][ell][N] encodings, NTT form
void encode_complex_on_device(Context& c, u64* dst, double* dv, int n_vec, int slots, int ell, long double scale) {
    const Context::FftDev& tab = fft_dev_tables(c, slots);
    launch_fft_special_inv(dv, tab.rot, tab.ksi, slots, n_vec, c.stream);
    // scale = mant * 2^exp with a 64-bit significand, exactly (the host code multiplies in x87 extended precision)
    int e2 = 0;
    const long double m = frexpl(scale, &e2);               // scale = m * 2^e2, m in [0.5, 1)
    const u64 mant = (u64)ldexpl(m, 64);
    launch_encode_round_reduce(c.dt, dst, dv, slots, ell, mant, e2 - 64, n_vec, c.stream);
    c.stats.encode += (u64)n_vec;
    c.ntt(LimbBatch{dst, n_vec * ell, nullptr, 0, ell}, false);
    hip_check(hipGetLastError(), "encode kernels (device)");
}

std::shared_ptr<Encoding> encode_to_device(Context& c, const std::vector<double>& values, const std::vector<double>& imag, int slots,
                                           int ell, long double scale) {
    c.require_device();
    if (slots < 1 || (slots & (slots - 1)) || slots > c.N / 2) throw Error(FHELIN_ERR_ARG, "encode: slots must be a power of two <= N/2");
    if (ell < 1 || ell > c.L + 1 + c.K) throw Error(FHELIN_ERR_ARG, "encode: level out of range");   // L + 1 + k: full key basis (see above)
    if (!c.host_encode && slots >= 2) {
        auto e = std::make_shared<Encoding>();
        e->ctx = &c;
        e->ell = ell;
        e->scale = scale;
        e->d = c.dalloc<u64>((size_t)ell * c.N);
        std::vector<double> re(slots, 0.0), im(slots, 0.0);
        for (int i = 0; i < slots && i < (int)values.size(); ++i) re[i] = values[i];
        for (int i = 0; i < slots && i < (int)imag.size(); ++i) im[i] = imag[i];
        encode_batch_device(c, e->d, re.data(), imag.empty() ? nullptr : im.data(), 1, slots, slots, ell, scale);
        return e;
    }
    std::vector<std::pair<double, double>> v(slots, {0.0, 0.0});
    for (int i = 0; i < slots && i < (int)values.size(); ++i) v[i].first = values[i];
    for (int i = 0; i < slots && i < (int)imag.size(); ++i) v[i].second = imag[i];
    ckks_fft_special(v, true);
    const size_t N = c.N;
    const size_t gap = (N / 2) / slots;
    std::vector<u64> coeffs(2 * N, 0);
    for (int i = 0; i < slots; ++i) {
        ld_to_i128((long double)v[i].first * scale, coeffs[2 * (i * gap)], coeffs[2 * (i * gap) + 1]);
        ld_to_i128((long double)v[i].second * scale, coeffs[2 * (i * gap + N / 2)], coeffs[2 * (i * gap + N / 2) + 1]);
    }
    u64* dco = c.dalloc<u64>(2 * N);
    c.upload_async(dco, coeffs.data(), 2 * N);  // pinned staging: no stream drain (the GPU keeps its queue)
    auto e = std::make_shared<Encoding>();
    e->ctx = &c;
    e->ell = ell;
    e->scale = scale;
    e->d = c.dalloc<u64>((size_t)ell * N);
    launch_reduce_i128(c.dt, e->d, dco, 0, ell, c.stream);
    c.stats.encode += 1;
    c.ntt(LimbBatch{e->d, ell, nullptr, 0, ell}, false);
    hip_check(hipGetLastError(), "encode kernels");
    c.pool.free(dco);
    return e;
}

std::shared_ptr<Encoding> Plaintext::at(int ell, long double scale) {
    // an encoding is made on the stream that first asks for it; a DIFFERENT lane that uses it later orders its stream behind the event
    // recorded after the encode (no host wait: with two lanes fed by one host thread a host-side synchronisation of one lane starves the
    // other - 2.9 s of host time per 16 samples before this, tools: FHELIN_HOST_WAITS=1)
    auto order = [&](const std::shared_ptr<Encoding>& e) {
        const int lane = ctx->pool.cur_lane;
        if (e->ready && lane != e->made_lane && !(e->lanes_ordered & (1u << lane))) {
            hip_check(hipStreamWaitEvent(ctx->stream, e->ready, 0), "hipStreamWaitEvent(encoding)");
            e->lanes_ordered |= 1u << lane;
        }
    };
    for (size_t i = 0; i < cache.size(); ++i)
        if (cache[i]->ell == ell && fabsl(cache[i]->scale / scale - 1.0L) < 1e-12L) {
            if (i) std::rotate(cache.begin(), cache.begin() + i, cache.begin() + i + 1);   // most recently used first
            order(cache[0]);
            return cache[0];
        }
    auto e = encode_to_device(*ctx, values, imag, slots, ell, scale);
    e->made_lane = ctx->pool.cur_lane;
    e->lanes_ordered = 1u << e->made_lane;
    if (ctx->n_lanes > 0) {
        hip_check(hipEventCreateWithFlags(&e->ready, hipEventDisableTiming), "hipEventCreate(encoding)");
        hip_check(hipEventRecord(e->ready, ctx->stream), "hipEventRecord(encoding)");
    }
    cache.insert(cache.begin(), e);
    // one entry per (limb count, scale) the plaintext has been used at: bounded by the chain length in principle, capped here
    // so that a long-lived mask or bootstrap diagonal cannot pin more than MAX_ENCODINGS device copies (least recently used
    // goes; callers hold their own reference while they enqueue work on it)
    if (cache.size() > MAX_ENCODINGS) {
        // the evicted copy may still be read by work in flight on any stream: its block must not go back to a pool before that
        // work is done (rare: more than MAX_ENCODINGS distinct levels for one plaintext)
        hip_check(hipStreamSynchronize(ctx->main_stream), "encoding eviction sync");
        for (int k = 1; k <= ctx->n_lanes; ++k) hip_check(hipStreamSynchronize(ctx->lane_stream[k]), "encoding eviction sync (lane)");
        cache.pop_back();
    }
    return e;
}

// ------------------------------------------------------------------------------------------------ Client
Client::Client(Evaluator& ev, const uint8_t seed[32]) : ev_(ev), c_(ev.ctx()), rng_(seed) {}
Client::~Client() {
    try {
        if (s_all) c_.pool.free(s_all);
        if (pk) c_.pool.free(pk);
    } catch (...) {
    }
}

// sample a small polynomial on the host, reduce it into `nl` limbs on the GPU and transform to NTT form
void Client::sample_small_to_ntt(u64* dst, int nlimbs_q, bool with_p, int kind) {
    const size_t N = c_.N;
    std::vector<u64> co(2 * N);
    for (size_t i = 0; i < N; ++i) {
        long v;
        if (kind == 0) v = lround(rng_.normal() * 3.19);
        else v = (long)(rng_.next() % 3) - 1;
        co[2 * i] = (u64)v;
        co[2 * i + 1] = v < 0 ? ~0ull : 0;
    }
    u64* dco = c_.dalloc<u64>(2 * N);
    c_.upload_async(dco, co.data(), 2 * N);
    launch_reduce_i128(c_.dt, dst, dco, 0, nlimbs_q, c_.stream);
    launch_ntt(c_.dt, LimbBatch{dst, nlimbs_q, nullptr, 0, nlimbs_q}, false, c_.stream);
    if (with_p && c_.K > 0) {
        u64* dp = dst + (size_t)nlimbs_q * N;
        launch_reduce_i128(c_.dt, dp, dco, c_.L + 1, c_.K, c_.stream);
        launch_ntt(c_.dt, LimbBatch{dp, c_.K, nullptr, c_.L + 1, c_.K}, false, c_.stream);
    }
    c_.pool.free(dco);
}

void Client::keygen() {
    c_.require_device();
    const size_t N = c_.N;
    const int L1 = c_.L + 1, nl = L1 + c_.K;
    // sparse ternary secret of Hamming weight h (reference SetSecretKeyDist(SPARSE_TERNARY), :8)
    std::vector<u64> co(2 * N, 0);
    int h = std::min<int>(c_.prm.hamming, (int)N);
    int placed = 0;
    while (placed < h) {
        size_t pos = rng_.uniform(N);
        if (co[2 * pos] | co[2 * pos + 1]) continue;
        if (rng_.next() & 1) {
            co[2 * pos] = 1;
        } else {
            co[2 * pos] = ~0ull;
            co[2 * pos + 1] = ~0ull;
        }
        ++placed;
    }
    if (!s_all) s_all = c_.dalloc<u64>((size_t)nl * N);
    u64* dco = c_.dalloc<u64>(2 * N);
    hip_check(hipMemcpyAsync(dco, co.data(), 2 * N * 8, hipMemcpyHostToDevice, c_.stream), "secret upload");
    hip_check(hipStreamSynchronize(c_.stream), "secret sync");
    launch_reduce_i128(c_.dt, s_all, dco, 0, nl, c_.stream);
    launch_ntt(c_.dt, LimbBatch{s_all, nl, nullptr, 0, nl}, false, c_.stream);
    c_.pool.free(dco);
    // public key over Q: a uniform (sampled directly in NTT form), b = e - a s
    if (!pk) pk = c_.dalloc<u64>((size_t)2 * L1 * N);
    std::vector<u64> a((size_t)L1 * N);
    for (int l = 0; l < L1; ++l)
        for (size_t i = 0; i < N; ++i) a[(size_t)l * N + i] = rng_.uniform(c_.chain.q[l]);
    u64* pa = pk + (size_t)L1 * N;
    hip_check(hipMemcpyAsync(pa, a.data(), a.size() * 8, hipMemcpyHostToDevice, c_.stream), "pk upload");
    hip_check(hipStreamSynchronize(c_.stream), "pk sync");
    u64* e = c_.dalloc<u64>((size_t)L1 * N);
    sample_small_to_ntt(e, L1, false, 0);
    launch_ew_mul(c_.dt, pk, pa, s_all, L1, L1, 0, L1, c_.stream);
    launch_ew_sub(c_.dt, pk, e, pk, L1, L1, 0, L1, c_.stream);
    hip_check(hipGetLastError(), "keygen kernels");
    c_.pool.free(e);
}

KeyPtr Client::make_switch_key(const u64* s_from_all, const u64* s_to_all) {
    c_.require_device();
    if (c_.K < 1) throw Error(FHELIN_ERR_STATE, "key switching keys need special primes (n_p >= 1)");
    const size_t N = c_.N;
    const int L1 = c_.L + 1, nl = L1 + c_.K;
    KeyPtr key = ev_.new_key();
    std::vector<u64> a((size_t)nl * N);
    u64* e = c_.dalloc<u64>((size_t)nl * N);
    u64* tmp = c_.dalloc<u64>((size_t)nl * N);
    for (int j = 0; j < key->digits; ++j) {
        u64* kb = key->d + (size_t)(2 * j) * nl * N;
        u64* ka = key->d + (size_t)(2 * j + 1) * nl * N;
        for (int l = 0; l < nl; ++l) {
            const u64 m = c_.moduli[l];
            for (size_t i = 0; i < N; ++i) a[(size_t)l * N + i] = rng_.uniform(m);
        }
        hip_check(hipMemcpyAsync(ka, a.data(), a.size() * 8, hipMemcpyHostToDevice, c_.stream), "evk upload");
        hip_check(hipStreamSynchronize(c_.stream), "evk sync");
        sample_small_to_ntt(e, L1, true, 0);
        launch_ew_mul(c_.dt, tmp, ka, s_to_all, nl, nl, 0, nl, c_.stream);
        launch_ew_sub(c_.dt, kb, e, tmp, nl, nl, 0, nl, c_.stream);
        // + P * (Q/Q_j) * [(Q/Q_j)^{-1}]_{Q_j} * s_from  ==  (P mod q_t) * s_from on the limbs of digit j, 0 elsewhere
        const int lo = j * c_.alpha, hi = std::min((j + 1) * c_.alpha, L1);
        ScalarSet sc;
        for (int t = lo; t < hi; ++t) {
            const u64 qt = c_.chain.q[t];
            u64 pm = 1;
            for (u64 p : c_.chain.p) pm = h_mulmod(pm, p % qt, qt);
            sc.v[2 * (t - lo)] = pm;
            sc.v[2 * (t - lo) + 1] = h_shoup(pm, qt);
        }
        launch_ew_scalar(c_.dt, tmp, s_from_all + (size_t)lo * N, sc, hi - lo, lo, hi - lo, c_.stream);
        launch_ew_add(c_.dt, kb + (size_t)lo * N, kb + (size_t)lo * N, tmp, hi - lo, hi - lo, lo, hi - lo, c_.stream);
    }
    hip_check(hipGetLastError(), "make_switch_key kernels");
    c_.pool.free(e);
    c_.pool.free(tmp);
    return key;
}

void Client::gen_relin_key() {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const int nl = c_.L + 1 + c_.K;
    u64* s2 = c_.dalloc<u64>((size_t)nl * c_.N);
    launch_ew_mul(c_.dt, s2, s_all, s_all, nl, nl, 0, nl, c_.stream);
    ev_.relin_key = make_switch_key(s2, s_all);
    c_.pool.free(s2);
}

void Client::gen_rotation_key(int index) {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const u64 g = c_.galois_element(index);
    if (ev_.rot_keys.count(g)) return;
    const int nl = c_.L + 1 + c_.K;
    // key switches from s to sigma_{g^-1}(s); applying sigma_g afterwards restores s (oracle orc_rotate)
    const u64 ginv = c_.galois_element(-index);
    u64* sp = c_.dalloc<u64>((size_t)nl * c_.N);
    launch_automorph(c_.dt, sp, s_all, c_.automorph_map(ginv), nl, c_.stream);
    ev_.rot_keys[g] = make_switch_key(s_all, sp);
    c_.pool.free(sp);
}

void Client::gen_conj_key() {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const u64 g = 2ull * c_.N - 1;  // X -> X^{-1}; its own inverse
    const int nl = c_.L + 1 + c_.K;
    u64* sp = c_.dalloc<u64>((size_t)nl * c_.N);
    launch_automorph(c_.dt, sp, s_all, c_.automorph_map(g), nl, c_.stream);
    ev_.conj_key = make_switch_key(s_all, sp);
    ev_.rot_keys[g] = ev_.conj_key;
    c_.pool.free(sp);
}

PtPtr Client::encode(const double* vals, int n, int level, int slots) {
    if (slots <= 0) slots = 1 << c_.prm.log_slots;
    if (slots & (slots - 1)) throw Error(FHELIN_ERR_ARG, "encode: slots must be a power of two");
    if (level < 0 || level > c_.L) throw Error(FHELIN_ERR_ARG, "encode: level out of range");
    auto p = std::make_shared<Plaintext>();
    p->ctx = &c_;
    p->slots = slots;
    p->level = level;
    p->values.assign(slots, 0.0);
    for (int i = 0; i < n && i < slots; ++i) p->values[i] = vals[i];
    return p;
}

void Client::sample_small_device(u64* dst, int n_poly, int ell, int kind) {
    SamplerKey key;   // a fresh ChaCha20 key per call, drawn from the client's own (secret-seeded) stream
    for (int i = 0; i < 4; ++i) {
        const u64 w = rng_.next();
        key.w[2 * i] = (u32)w;
        key.w[2 * i + 1] = (u32)(w >> 32);
    }
    launch_sample_small(c_.dt, dst, key, (sample_calls_++) << 32, kind, ell, n_poly, c_.stream);
}

std::vector<long> Client::debug_sample(int kind, int n_poly) {
    c_.require_device();
    const size_t N = c_.N;
    u64* d = c_.dalloc<u64>((size_t)n_poly * N);
    sample_small_device(d, n_poly, 1, kind);
    std::vector<u64> h((size_t)n_poly * N);
    hip_check(hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c_.stream), "sample download");
    hip_check(hipStreamSynchronize(c_.stream), "sample sync");
    c_.pool.free(d);
    const u64 q0 = c_.chain.q[0];
    std::vector<long> out(h.size());
    for (size_t i = 0; i < h.size(); ++i) out[i] = h[i] > q0 / 2 ? -(long)(q0 - h[i]) : (long)h[i];
    return out;
}

// c0 = b u + e0 + m, c1 = a u + e1 for n_vec encodings enc [n_vec][ell][N] (stride enc_stride words; 0 = one shared encoding)
void Client::encrypt_encoded(const u64* enc, size_t enc_stride, int n_vec, int ell, long double scale, int slots, std::vector<CtPtr>& out) {
    Context& c = c_;
    const size_t N = c.N, pn = (size_t)ell * N;
    const int L1 = c.L + 1;
    std::vector<CtPtr> cts = ev_.new_ct_batch(n_vec, 2, ell, 1, scale, slots);
    u64* rnd = c.dalloc<u64>((size_t)3 * n_vec * pn);       // u | e0 | e1, each [n_vec][ell][N]
    sample_small_device(rnd, n_vec, ell, 1);
    sample_small_device(rnd + (size_t)n_vec * pn, 2 * n_vec, ell, 0);
    c.ntt(LimbBatch{rnd, 3 * n_vec * ell, nullptr, 0, ell}, false);
    launch_encrypt_combine(c.dt, cts[0]->d, pk, rnd, rnd + (size_t)n_vec * pn, rnd + (size_t)2 * n_vec * pn, enc, ell, L1, enc_stride, n_vec,
                           c.stream);
    hip_check(hipGetLastError(), "encrypt kernels");
    // the encryption randomness (u, e0, e1) does not stay behind in a recycled pool block
    hip_check(hipMemsetAsync(rnd, 0, (size_t)3 * n_vec * pn * sizeof(u64), c.stream), "hipMemsetAsync(encryption randomness)");
    c.pool.free(rnd);
    for (auto& ct : cts) out.push_back(ct);
}

CtPtr Client::encrypt(const PtPtr& p, int drop) {
    if (!pk) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const int level = std::min(c_.L, p->level + std::max(0, drop));   // level plan: start `drop` limbs lower
    const int ell = c_.L + 1 - level;
    auto enc = p->at(ell, c_.sf_real[level]);
    std::vector<CtPtr> out;
    encrypt_encoded(enc->d, 0, 1, ell, enc->scale, p->slots, out);
    return out[0];
}

std::vector<CtPtr> Client::encrypt_batch(const double* vals, int n_vec, int n_per, int level, int slots) {
    if (!pk) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    if (slots <= 0) slots = 1 << c_.prm.log_slots;
    if (level < 0 || level > c_.L) throw Error(FHELIN_ERR_ARG, "encrypt: level out of range");
    if (n_vec < 0 || n_per < 0) throw Error(FHELIN_ERR_ARG, "encrypt_batch: negative count");
    const int ell = c_.L + 1 - level;
    const size_t pn = (size_t)ell * c_.N;
    const long double scale = c_.sf_real[level];
    std::vector<CtPtr> out;
    const int CHUNK = 32;                                    // bounds the temporaries (4 polynomials per vector in flight)
    for (int lo = 0; lo < n_vec; lo += CHUNK) {
        const int n = std::min(CHUNK, n_vec - lo);
        u64* enc = c_.dalloc<u64>((size_t)n * pn);
        encode_batch_device(c_, enc, vals + (size_t)lo * n_per, nullptr, n, n_per, slots, ell, scale);
        encrypt_encoded(enc, pn, n, ell, scale, slots, out);
        c_.pool.free(enc);
    }
    return out;
}

// One sample's client side on the device (kernels_client.h "sample ingestion"): embedding rows (given, or gathered from a table by
// token id) + positional embedding, the two Linformer projections, the expanded packing, encoding and encryption.  drop[v]:
// limbs vector v starts lower by (level plan); vectors of one level share the batched encryptor.
std::vector<CtPtr> Client::ingest_sample(const double* emb, const int* tokens, const double* table, int vocab, int S, const double* cls,
                                         const double* pos, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                                         int w_cols, int level, const std::vector<int>& drop, std::vector<double>* proj_out) {
    if (!pk) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const int slots = 1 << c_.prm.log_slots, S1 = S + 1, n_vec = 64 + S1;
    if (slots != 16384) throw Error(FHELIN_ERR_ARG, "ingest: the expanded layout needs 16384 slots (128 x 128)");
    if (S < 1 || S1 > w_cols || (!emb && !(tokens && table && vocab > 0))) throw Error(FHELIN_ERR_ARG, "ingest: bad token count / inputs");
    if (level < 0 || level > c_.L || (int)drop.size() != n_vec) throw Error(FHELIN_ERR_ARG, "ingest: level out of range");
    hipStream_t s = c_.stream;
    // every temporary of the sample - the embeddings, x_in, the projections, the expanded packing: the client's PLAINTEXT - is wiped
    // before its block goes back to the recycled pool, and freed on every path out of this function (an exception included)
    struct Temps {
        Context& c;
        hipStream_t s;
        std::vector<std::pair<void*, size_t>> blocks;
        void* get(size_t bytes) {
            void* d = c.pool.alloc(bytes);
            blocks.emplace_back(d, bytes);
            return d;
        }
        ~Temps() {
            for (auto& b : blocks) {
                (void)hipMemsetAsync(b.first, 0, b.second, s);
                try { c.pool.free(b.first); } catch (...) {}
            }
        }
    } tmp{c_, s, {}};
    auto up = [&](const void* h, size_t bytes) {
        void* d = tmp.get(bytes);
        hip_check(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s), "ingest upload");
        return d;
    };
    if (tokens)
        for (int t = 0; t < S; ++t)
            if (tokens[t] < 0 || tokens[t] >= vocab) throw Error(FHELIN_ERR_ARG, "ingest: token id outside the embedding table");
    double* d_emb = emb ? (double*)up(emb, (size_t)S * 128 * 8) : nullptr;
    int* d_tok = tokens ? (int*)up(tokens, (size_t)S * 4) : nullptr;
    double* d_tab = tokens ? (double*)up(table, (size_t)vocab * 128 * 8) : nullptr;
    double* d_cls = (double*)up(cls, 128 * 8);
    double* d_pos = (double*)up(pos, (size_t)S * 128 * 8);
    double* d_Ew = (double*)up(E_w, (size_t)32 * w_cols * 8);
    double* d_Fw = (double*)up(F_w, (size_t)32 * w_cols * 8);
    double* d_Eb = (double*)up(E_b, 32 * 8);
    double* d_Fb = (double*)up(F_b, 32 * 8);
    double* x_in = (double*)tmp.get((size_t)S1 * 128 * 8);
    double* proj = (double*)tmp.get((size_t)64 * 128 * 8);
    launch_ingest_xin(x_in, d_emb, d_tok, d_tab, d_cls, d_pos, S, s);
    launch_ingest_project(proj, x_in, d_Ew, d_Eb, d_Fw, d_Fb, w_cols, S1, s);
    double* dv = (double*)tmp.get((size_t)n_vec * slots * 16);
    launch_ingest_expand(dv, proj, x_in, S1, slots, s);
    hip_check(hipGetLastError(), "ingest kernels");
    if (proj_out) {   // test hook: x_in rows then the 64 projected rows, as computed on the device
        proj_out->resize((size_t)(S1 + 64) * 128);
        hip_check(hipMemcpyAsync(proj_out->data(), x_in, (size_t)S1 * 128 * 8, hipMemcpyDeviceToHost, s), "ingest download");
        hip_check(hipMemcpyAsync(proj_out->data() + (size_t)S1 * 128, proj, (size_t)64 * 128 * 8, hipMemcpyDeviceToHost, s), "ingest download");
    }
    hip_check(hipStreamSynchronize(s), "ingest sync");   // the host buffers are the caller's: done with them
    std::vector<CtPtr> out(n_vec);
    std::vector<char> seen(n_vec, 0);
    for (int i = 0; i < n_vec; ++i) {
        if (seen[i]) continue;
        const int lvl = std::min(c_.L, level + std::max(0, drop[i]));
        const int ell = c_.L + 1 - lvl;
        const size_t pn = (size_t)ell * c_.N;
        const long double scale = c_.sf_real[lvl];
        // runs of consecutive vectors that start at this level, in chunks of 32 (bounds the temporaries)
        int j = i;
        while (j < n_vec) {
            if (seen[j] || std::min(c_.L, level + std::max(0, drop[j])) != lvl) {
                ++j;
                continue;
            }
            int hi = j;
            while (hi < n_vec && hi - j < 32 && !seen[hi] && std::min(c_.L, level + std::max(0, drop[hi])) == lvl) ++hi;
            const int n = hi - j;
            u64* enc = (u64*)tmp.get((size_t)n * pn * sizeof(u64));   // the encoded plaintext: wiped and freed with the other temporaries
            encode_complex_on_device(c_, enc, dv + (size_t)j * slots * 2, n, slots, ell, scale);
            std::vector<CtPtr> part;
            encrypt_encoded(enc, pn, n, ell, scale, slots, part);
            for (int k = 0; k < n; ++k) {
                out[j + k] = part[k];
                seen[j + k] = 1;
            }
            j = hi;
        }
    }
    return out;   // ~Temps wipes and frees
}

CtPtr Client::phase(const CtPtr& ct, int nl) {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    if (nl < 1 || nl > ct->ell || ct->npoly < 2) throw Error(FHELIN_ERR_ARG, "phase: bad limb count / component count");
    const size_t pn = (size_t)ct->ell * c_.N;
    CtPtr o = ev_.new_ct(1, nl, ct->deg, ct->scale, ct->slots);
    u64* m = o->d;
    // m = c0 + c1 s (+ c2 s^2) on the first nl limbs
    launch_ew_muladd(c_.dt, m, ct->d, ct->d + pn, s_all, nl, nl, 0, nl, c_.stream);
    if (ct->npoly == 3) {
        u64* s2 = c_.dalloc<u64>((size_t)nl * c_.N);
        launch_ew_mul(c_.dt, s2, s_all, s_all, nl, nl, 0, nl, c_.stream);
        launch_ew_muladd(c_.dt, m, m, ct->d + 2 * pn, s2, nl, nl, 0, nl, c_.stream);
        c_.pool.free(s2);
    }
    hip_check(hipGetLastError(), "phase kernels");
    return o;
}

std::vector<double> Client::decrypt(const CtPtr& cin, int slots) {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    CtPtr ct = cin;
    while (ct->deg > 1 && ct->ell > 2) ct = ev_.rescale(ct);
    if (slots <= 0) slots = ct->slots > 0 ? ct->slots : (1 << c_.prm.log_slots);
    const size_t N = c_.N;
    const int ell = ct->ell, nl = std::min(ell, 2);
    CtPtr ph = phase(ct, nl);
    u64* m = ph->d;
    c_.ntt(LimbBatch{m, nl, nullptr, 0, nl}, true);
    std::vector<u64> h((size_t)nl * N);
    hip_check(hipMemcpyAsync(h.data(), m, h.size() * 8, hipMemcpyDeviceToHost, c_.stream), "decrypt download");
    hip_check(hipStreamSynchronize(c_.stream), "decrypt sync");
    ph.reset();
    const u64 q0 = c_.chain.q[0];
    const size_t gap = (N / 2) / slots;
    std::vector<std::pair<double, double>> v(slots);
    const u64 q1 = nl > 1 ? c_.chain.q[1] : 1;
    const u64 inv = nl > 1 ? h_invmod(q0 % q1, q1) : 0;     // q0^-1 mod q1, once (not per coefficient)
    const u128 Q = (u128)q0 * q1;
    auto lift = [&](size_t idx) -> long double {
        if (nl == 1) {
            u64 x = h[idx];
            return x > q0 / 2 ? -(long double)(q0 - x) : (long double)x;
        }
        const u64 x0 = h[idx], x1 = h[N + idx];
        const u64 d = h_mulmod(sub_mod(x1, x0 % q1, q1), inv, q1);
        const u128 x = (u128)x0 + (u128)q0 * d;
        if (x > Q / 2) {
            const u128 mag = Q - x;
            return -((long double)(u64)(mag >> 64) * 18446744073709551616.0L + (long double)(u64)mag);
        }
        return (long double)(u64)(x >> 64) * 18446744073709551616.0L + (long double)(u64)x;
    };
    for (int i = 0; i < slots; ++i) {
        v[i].first = (double)(lift(i * gap) / ct->scale);
        v[i].second = (double)(lift(i * gap + N / 2) / ct->scale);
    }
    ckks_fft_special(v, false);
    std::vector<double> out(slots);
    for (int i = 0; i < slots; ++i) out[i] = v[i].first;
    return out;
}

void Client::export_secret(u64* out) {
    if (!s_all) throw Error(FHELIN_ERR_KEY, "keygen() has not been called");
    const size_t n = (size_t)(c_.L + 1 + c_.K) * c_.N;
    hip_check(hipMemcpyAsync(out, s_all, n * 8, hipMemcpyDeviceToHost, c_.stream), "export secret");
    hip_check(hipStreamSynchronize(c_.stream), "sync");
}

void Client::import_secret(const u64* in) {
    c_.require_device();
    const size_t n = (size_t)(c_.L + 1 + c_.K) * c_.N;
    if (!s_all) s_all = c_.dalloc<u64>(n);
    hip_check(hipMemcpyAsync(s_all, in, n * 8, hipMemcpyHostToDevice, c_.stream), "import secret");
    hip_check(hipStreamSynchronize(c_.stream), "sync");
}

}  // namespace fhelin
