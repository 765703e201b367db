#include "bootstrap.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <set>

namespace fhelin {

namespace {

const double PI = 3.14159265358979323846264338327950288;

void add_diag(DiagMap& m, int r, int n, int pos, cplx v) {
    r = ((r % n) + n) % n;
    auto& d = m[r];
    if (d.empty()) d.assign(n, cplx(0, 0));
    d[pos] += v;
}

// one radix-2 stage of the special FFT as a 3-diagonal map (see ckks_fft_special in client.cpp)
DiagMap stage_map(int n, int len, bool inverse, const std::vector<u32>& rot, const std::vector<std::pair<double, double>>& ksi) {
    DiagMap m;
    const int lenh = len >> 1, lenq = len << 2, gap = (4 * n) / lenq;
    for (int p = 0; p < n; ++p) {
        const int jj = p % len;
        if (inverse) {
            if (jj < lenh) {
                add_diag(m, 0, n, p, 1.0);
                add_diag(m, lenh, n, p, 1.0);
            } else {
                const int j = jj - lenh;
                const auto& k = ksi[(lenq - (rot[j] % lenq)) * gap];
                const cplx w(k.first, k.second);
                add_diag(m, -lenh, n, p, w);
                add_diag(m, 0, n, p, -w);
            }
        } else {
            if (jj < lenh) {
                const auto& k = ksi[(rot[jj] % lenq) * gap];
                add_diag(m, 0, n, p, 1.0);
                add_diag(m, lenh, n, p, cplx(k.first, k.second));
            } else {
                const int j = jj - lenh;
                const auto& k = ksi[(rot[j] % lenq) * gap];
                add_diag(m, -lenh, n, p, 1.0);
                add_diag(m, 0, n, p, -cplx(k.first, k.second));
            }
        }
    }
    return m;
}

// (B after A): out = sum_s e_s * rot_s( sum_r d_r * rot_r(x) ) = sum_{r,s} e_s * rot_s(d_r) * rot_{r+s}(x)
DiagMap compose(const DiagMap& B, const DiagMap& A, int n) {
    DiagMap C;
    for (const auto& be : B)
        for (const auto& ae : A) {
            const int s = be.first, r = ae.first;
            auto& c = C[(r + s) % n];
            if (c.empty()) c.assign(n, cplx(0, 0));
            for (int j = 0; j < n; ++j) c[j] += be.second[j] * ae.second[(j + s) % n];
        }
    // drop numerically empty diagonals
    for (auto it = C.begin(); it != C.end();) {
        double mx = 0;
        for (const auto& v : it->second) mx = std::max(mx, std::abs(v));
        if (mx < 1e-300) it = C.erase(it);
        else ++it;
    }
    return C;
}

// Sparse packing (2 slots <= N/2): a ciphertext with n slots has room for 2n.  The LAST CoeffsToSlots stage written over 2n
// slots with the diagonals [d_r | -i d_r] leaves w on the left half and -i w on the right one, so that after adding the
// conjugate the left half holds the real parts (t_k) and the right half the imaginary parts (t_{k+n}): ONE ciphertext goes
// through the modular reduction instead of two.
DiagMap pack_last_c2s(const DiagMap& m, int n) {
    DiagMap out;
    for (const auto& e : m) {
        std::vector<cplx> d(2 * n);
        for (int j = 0; j < n; ++j) {
            d[j] = e.second[j];
            d[j + n] = cplx(0, -1) * e.second[j];
        }
        out[e.first] = d;
    }
    return out;
}
// ... and the FIRST SlotsToCoeffs stage takes u = [L | R] (2n real slots) in place of the n-slot vector z = L + i R it is
// defined on: out_j = sum_r e_r[j mod n] z_{(j + r) mod n} for all 2n positions j (the result has period n again), with
// z_q = u_q + i u_{q+n}: every diagonal r becomes two, at offsets r and r + n of the 2n-slot rotation group.
DiagMap unpack_first_s2c(const DiagMap& m, int n) {
    DiagMap out;
    for (const auto& e : m) {
        const int r = ((e.first % n) + n) % n;
        std::vector<cplx> a(2 * n), b(2 * n);
        for (int j = 0; j < 2 * n; ++j) {
            const int jj = j % n;
            const bool wraps = jj + r >= n;             // (jj + r) mod n came around: the halves of u trade places
            const bool left = (j < n) != wraps;         // rot_r(u)_j is the L entry (else rot_{r+n}(u)_j is)
            const cplx v = e.second[jj];
            a[j] = left ? v : cplx(0, 1) * v;           // coefficient of rot_r(u)_j
            b[j] = left ? cplx(0, 1) * v : v;           // coefficient of rot_{r+n}(u)_j
        }
        out[r] = a;
        out[r + n] = b;
    }
    return out;
}

void scale_map(DiagMap& m, double f) {
    for (auto& e : m)
        for (auto& v : e.second) v *= f;
}

}  // namespace

Bootstrapper::~Bootstrapper() {
    if (mono_i_) {
        try { ev_.ctx().pool.free(mono_i_); } catch (...) {}
    }
}

LinStage Bootstrapper::prepare(const DiagMap& m, int n) {
    LinStage st;
    int g = n;
    for (const auto& e : m)
        if (e.first) g = std::gcd(g, e.first);
    st.g0 = g;
    const int span = n / g;  // number of possible multiples
    const int count = (int)m.size();
    int bsz = 1;
    while (bsz * bsz < count) bsz <<= 1;
    st.bsz = std::min(bsz, span);
    for (const auto& e : m) {
        const int k = e.first / g;
        const int G = k / st.bsz, B = k % st.bsz;
        const int sG = G * st.bsz * g;
        auto p = std::make_shared<Plaintext>();
        p->ctx = &ev_.ctx();
        p->slots = n;
        p->level = 0;
        p->values.resize(n);
        p->imag.resize(n);
        for (int j = 0; j < n; ++j) {
            const cplx v = e.second[((j - sG) % n + n) % n];  // rot_{-sG}(d)
            p->values[j] = v.real();
            p->imag[j] = v.imag();
        }
        st.terms.push_back({sG, B * g, p});
    }
    return st;
}

void Bootstrapper::setup(int budget_enc, int budget_dec, int slots) {
    Context& c = ev_.ctx();
    c.require_device();
    if (!cl_.has_keys()) throw Error(FHELIN_ERR_KEY, "bootstrap setup needs the secret key (keygen first)");
    if (slots <= 0) slots = 1 << c.prm.log_slots;
    if (slots & (slots - 1) || slots > c.N / 2 || slots < 4) throw Error(FHELIN_ERR_ARG, "bootstrap: slots must be a power of two in [4, N/2]");
    if (budget_enc < 1 || budget_dec < 1) throw Error(FHELIN_ERR_ARG, "bootstrap: level budget must be >= 1");
    slots_ = slots;
    const int n = slots;
    int logn = 0;
    while ((1 << logn) < n) ++logn;
    budget_enc = std::min(budget_enc, logn);
    budget_dec = std::min(budget_dec, logn);
    std::vector<u32> rot;
    std::vector<std::pair<double, double>> ksi;
    ckks_fft_tables(n, rot, ksi);

    const int gapN = (c.N / 2) / n;
    packed_ = gapN >= 2;
    if (const char* e = std::getenv("FHELIN_BOOT_PACKED")) packed_ = packed_ && std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_BOOT_STAGES_LEGACY")) stage_order_legacy_ = std::atoi(e) != 0;
    const double q0 = (double)c.chain.q[0];
    // CoeffsToSlots: inverse-FFT stages (len = n .. 2), total constant Delta_r / (gapN q0 K 2n) spread over the levels
    auto build = [&](bool inverse, int budget, double total_factor, std::vector<LinStage>& out) {
        std::vector<int> lens;
        if (inverse) for (int len = n; len >= 2; len >>= 1) lens.push_back(len);
        else for (int len = 2; len <= n; len <<= 1) lens.push_back(len);
        const double f = std::pow(total_factor, 1.0 / budget);
        size_t pos = 0;
        for (int g = 0; g < budget; ++g) {
            // radix levels per stage: the remainder goes to the LATER stages.  The first CoeffsToSlots stage runs at the top of
            // the chain (the dearest key switches) and the first SlotsToCoeffs stage has its diagonals doubled under sparse
            // packing: both should be the small ones (14 levels, budget 3: 4 + 5 + 5 rather than 5 + 5 + 4)
            int cnt = logn / budget + (budget - 1 - g < logn % budget ? 1 : 0);
            if (stage_order_legacy_) cnt = logn / budget + (g < logn % budget ? 1 : 0);
            DiagMap m = stage_map(n, lens[pos], inverse, rot, ksi);
            for (int i = 1; i < cnt; ++i) m = compose(stage_map(n, lens[pos + i], inverse, rot, ksi), m, n);
            pos += cnt;
            scale_map(m, f);
            if (packed_ && inverse && g == budget - 1) out.push_back(prepare(pack_last_c2s(m, n), 2 * n));
            else if (packed_ && !inverse && g == 0) out.push_back(prepare(unpack_first_s2c(m, n), 2 * n));
            else out.push_back(prepare(m, n));
        }
    };
    c2s_.clear();
    s2c_.clear();
    const double delta_r = (double)c.sf_real[0];
    build(true, budget_enc, delta_r / ((double)gapN * q0 * K * 2.0 * n), c2s_);
    build(false, budget_dec, 1.0 / (2.0 * PI), s2c_);

    // rotation keys: baby/giant shifts of every stage, SubSum shifts, conjugation
    std::set<int> idx;
    for (const auto* v : {&c2s_, &s2c_})
        for (const auto& st : *v)
            for (const auto& t : st.terms) {
                if (t.giant) idx.insert(t.giant);
                if (t.baby) idx.insert(t.baby);
            }
    for (int r : idx) cl_.gen_rotation_key(r);
    for (int j = 1; j < gapN; j <<= 1) cl_.gen_rotation_key(n * j);
    if (!ev_.conj_key) cl_.gen_conj_key();

    // cosine fit: f(y) = cos((2 pi K y - pi/2) / 2^R) on [-1,1]; R double-angle steps give sin(2 pi K y)
    const int d = cheb_degree, nn = d + 1;
    cheb_.assign(nn, 0.0);
    std::vector<double> fx(nn);
    const double sc = 1.0 / (double)(1 << R);
    for (int j = 0; j < nn; ++j) fx[j] = std::cos((2.0 * PI * K * std::cos(PI * (j + 0.5) / nn) - PI / 2) * sc);
    for (int k = 0; k < nn; ++k) {
        double s = 0;
        for (int j = 0; j < nn; ++j) s += fx[j] * std::cos(PI * k * (j + 0.5) / nn);
        cheb_[k] = 2.0 * s / nn;
    }

    // multiplication by i == multiplication by the monomial X^{N/2}
    const int L1 = c.L + 1;
    const size_t N = c.N;
    if (!mono_i_) mono_i_ = c.dalloc<u64>((size_t)L1 * N);
    std::vector<u64> h((size_t)L1 * N, 0);
    for (int l = 0; l < L1; ++l) h[(size_t)l * N + N / 2] = 1;
    hip_check(hipMemcpyAsync(mono_i_, h.data(), h.size() * 8, hipMemcpyHostToDevice, c.stream), "monomial upload");
    hip_check(hipStreamSynchronize(c.stream), "monomial sync");
    launch_ntt(c.dt, LimbBatch{mono_i_, L1, nullptr, 0, L1}, false, c.stream);
    hip_check(hipGetLastError(), "monomial ntt");
    depth_ = budget_enc + budget_dec + R + 6;
}

CtPtr Bootstrapper::mult_i(const CtPtr& x) {
    Context& c = ev_.ctx();
    CtPtr o = ev_.new_ct(x->npoly, x->ell, x->deg, x->scale, x->slots);
    launch_ew_mul(c.dt, o->d, x->d, mono_i_, x->npoly * x->ell, x->ell, 0, x->ell, c.stream);
    hip_check(hipGetLastError(), "mult_i");
    return o;
}

CtPtr Bootstrapper::apply(const LinStage& st, const CtPtr& xin) {
    CtPtr x = xin->deg >= 2 ? ev_.rescale(xin) : xin;
    // baby steps: every rotation of x in ONE hoisted key switch (one ModUp of x, per-index inner product + ModDown)
    std::vector<int> bidx;
    for (const auto& t : st.terms)
        if (std::find(bidx.begin(), bidx.end(), t.baby) == bidx.end()) bidx.push_back(t.baby);
    std::vector<CtPtr> brot = ev_.rotate_many(x, bidx);
    std::map<int, CtPtr> babies;
    for (size_t i = 0; i < bidx.size(); ++i) babies[bidx[i]] = brot[i];
    // per giant step: sum_b diag_{g,b} * rot_b(x) as ONE inner-product pass (Evaluator::dot_plain) instead of a product
    // launch per term and a tree of additions
    std::map<int, std::pair<std::vector<CtPtr>, std::vector<PtPtr>>> groups;
    for (const auto& t : st.terms) {
        groups[t.giant].first.push_back(babies[t.baby]);
        groups[t.giant].second.push_back(t.diag);
    }
    // an input whose scale is not its level's own (the first stage after a ModRaise to fewer limbs than the chain has:
    // run(drop > 0)) takes the diagonals at the scale that makes the product land on the next level's scale again
    Context& c = ev_.ctx();
    long double pt_scale = 0;
    const int lvl = x->level();
    if (x->ell >= 2 && fabsl(x->scale / c.sf_real[lvl] - 1.0L) > 1e-12L)
        pt_scale = c.sf_real[lvl + 1] * (long double)c.chain.q[x->ell - 1] / x->scale;
    // the inner sums land in ONE block, in giant-step order (the shared-ModDown key switch below takes them as they stand), and
    // all of them come out of ONE pass over the rotated ciphertexts (Evaluator::dot_plain_groups); per-group passes otherwise
    std::map<int, CtPtr> inner;
    {
        CtPtr f = groups.begin()->second.first[0];
        const int ell_in = f->deg >= 2 ? f->ell - 1 : f->ell;
        std::vector<CtPtr> slab = ev_.new_ct_batch((int)groups.size(), 2, ell_in, 2, 0, f->slots);
        std::vector<int> order;   // unrotated group first, then the rotated ones in ascending order (= rotate_each_sum's order)
        for (auto& g : groups)
            if (g.first % f->slots == 0) order.push_back(g.first);
        for (auto& g : groups)
            if (g.first % f->slots != 0) order.push_back(g.first);
        std::vector<CtPtr> cts;
        for (int b : bidx) cts.push_back(babies[b]);
        std::vector<std::vector<PtPtr>> pts(order.size(), std::vector<PtPtr>(bidx.size()));
        for (const auto& t : st.terms) {
            const size_t gi = std::find(order.begin(), order.end(), t.giant) - order.begin();
            const size_t bi = std::find(bidx.begin(), bidx.end(), t.baby) - bidx.begin();
            pts[gi][bi] = t.diag;
        }
        if (ev_.dot_plain_groups(cts, pts, pt_scale, slab)) {
            for (size_t k = 0; k < order.size(); ++k) inner[order[k]] = slab[k];
        } else {
            size_t k = 0;
            for (int g : order) {
                auto& grp = groups[g];
                inner[g] = ev_.dot_plain(grp.first, grp.second, pt_scale, slab[k++]);
            }
        }
    }
    // giant steps: different inputs, different keys, same shape
    std::vector<CtPtr> gin;
    std::vector<int> gidx;
    for (auto& g : inner) {
        gin.push_back(g.second);
        gidx.push_back(g.first);
    }
    // sum_g rot(inner_g, g): own ModUp per term, ONE shared ModDown (Evaluator::rotate_each_sum)
    return ev_.rotate_each_sum(gin, gidx);
}

CtPtr Bootstrapper::mod_raise(const CtPtr& ct, long double& rho, int top_ell) {
    Context& c = ev_.ctx();
    CtPtr x = ct->deg >= 2 ? ev_.rescale(ct) : ct;
    if (x->npoly != 2) throw Error(FHELIN_ERR_STATE, "bootstrap: ciphertext must be relinearised");
    if (x->ell < 2) throw Error(FHELIN_ERR_STATE, "bootstrap: need at least two limbs to set the message scale (bootstrap one level earlier)");
    if (x->ell > 2) x = ev_.level_reduce(x, 2);
    const long double q0 = (long double)c.chain.q[0], q1 = (long double)c.chain.q[1];
    const long double target = q0 / (long double)(1ull << correction);
    const u64 k0 = (u64)llroundl(target * q1 / x->scale);
    if (k0 < 2) throw Error(FHELIN_ERR_STATE, "bootstrap: ciphertext scale too large for the correction factor");
    x = ev_.mult_int(x, k0, true, x->scale * (long double)k0);
    x = ev_.rescale(x);  // one limb (q0), scale ~ q0 / 2^correction
    rho = x->scale * (long double)(1ull << correction) / q0;

    c.stats.bootstrap += 1;
    CtPtr up = ev_.raw_modraise(x, top_ell);
    up->deg = 1;
    up->scale = c.sf_real[0];   // the scale the CoeffsToSlots constants were built for (setup), whatever the start level
    up->slots = slots_;
    // SubSum: project onto the subring of X^{N/(2 slots)} (sparse packing)
    const int gapN = (c.N / 2) / slots_;
    for (int j = 1; j < gapN; j <<= 1) {
        const u64 g = c.galois_element(slots_ * j);
        auto it = ev_.rot_keys.find(g);
        if (it == ev_.rot_keys.end()) throw Error(FHELIN_ERR_KEY, "bootstrap: SubSum rotation key missing");
        up = ev_.add(up, ev_.raw_rotate(up, g, *it->second));
    }
    return up;
}

// EvalMod of the real and the imaginary half together: both go through the same Chebyshev fit and double-angle steps,
// so every multiplication is one batched relinearisation over the pair
std::vector<CtPtr> Bootstrapper::eval_mod(const std::vector<CtPtr>& xs) {
    std::vector<CtPtr> u = ev_.eval_chebyshev_many(xs, cheb_, -1.0, 1.0);
    for (int i = 0; i < R; ++i) {
        if (ev_.merged_products) {     // rescale(2 u^2 - 1) in one go: the next step (or the first SlotsToCoeffs stage) rescales it anyway
            u = ev_.mult_affine_rescale_batch(u, u, 2, -1.0, {});
            continue;
        }
        std::vector<CtPtr> t = ev_.mult_batch(u, u);
        t = ev_.add_batch(t, t);
        for (size_t k = 0; k < t.size(); ++k) u[k] = ev_.add_real(t[k], -1.0);
    }
    return u;
}

CtPtr Bootstrapper::run(const CtPtr& ct, int stop_after, int drop) {
    if (!ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
    Context& c = ev_.ctx();
    if (drop < 0 || c.L + 1 - drop - depth_ < 1) throw Error(FHELIN_ERR_ARG, "bootstrap: level plan leaves no limb for the result");
    long double rho = 1;
    CtPtr w = mod_raise(ct, rho, c.L + 1 - drop);
    if (stop_after == 1) return w;
    for (const auto& st : c2s_) w = apply(st, w);
    if (packed_) w->slots = 2 * slots_;          // the last stage wrote [w | -i w] over 2n slots
    CtPtr wc = ev_.conjugate(w);
    CtPtr a = ev_.add(w, wc);                    // real parts  t_k / (q0 K)  (1/2 folded into the DFT constants); packed: [t_k | t_{k+n}]
    if (stop_after == 2) return a;
    CtPtr v;
    if (packed_) {
        v = eval_mod({a})[0];                    // one modular reduction for both halves
        if (stop_after == 3) return v;
    } else {
        CtPtr b = mult_i(ev_.sub(wc, w));        // imaginary   t_{k+n}   / (q0 K)
        if (stop_after == 3) return eval_mod({a})[0];
        std::vector<CtPtr> ab = eval_mod({a, b});
        v = ev_.add(ab[0], mult_i(ab[1]));
    }
    for (size_t i = 0; i < s2c_.size(); ++i) {
        v = apply(s2c_[i], v);
        if (packed_ && i == 0) v->slots = slots_;   // the first stage reads [L | R] over 2n slots and leaves an n-periodic vector
    }
    // slots now hold m * rho / 2^correction: undo the correction exactly and absorb rho in the scale
    v = ev_.mult_int(v, 1ull << correction, false, v->scale);
    v->scale = v->scale * rho;
    if (v->deg >= 2) v = ev_.rescale(v);
    v->slots = ct->slots > 0 ? ct->slots : slots_;
    return v;
}

// ---- the same pipeline over a batch of independent ciphertexts ----------------------------------------------------------
// Every step below is the batched form of the step of run() / apply() above it stands for; each ciphertext goes through
// exactly the same residue operations (the batched evaluator entry points return the residues of the single ones).
std::vector<CtPtr> Bootstrapper::apply_batch(const LinStage& st, const std::vector<CtPtr>& xin) {
    Context& c = ev_.ctx();
    const size_t B = xin.size();
    std::vector<CtPtr> xs = xin[0]->deg >= 2 ? ev_.rescale_batch(xin) : xin;
    std::vector<int> bidx;
    for (const auto& t : st.terms)
        if (std::find(bidx.begin(), bidx.end(), t.baby) == bidx.end()) bidx.push_back(t.baby);
    // baby steps: ONE ModUp over the batch, per ciphertext the inner products of all indices, ONE ModDown over everything
    std::vector<std::vector<CtPtr>> brot = ev_.rotate_many_batch(xs, bidx);
    const CtPtr& x = xs[0];
    long double pt_scale = 0;
    const int lvl = x->level();
    if (x->ell >= 2 && fabsl(x->scale / c.sf_real[lvl] - 1.0L) > 1e-12L)
        pt_scale = c.sf_real[lvl + 1] * (long double)c.chain.q[x->ell - 1] / x->scale;
    // giant-step groups: the unrotated one first, then ascending (= rotate_each_sum's order)
    std::vector<int> order;
    {
        std::set<int> gs;
        for (const auto& t : st.terms) gs.insert(t.giant);
        for (int g : gs)
            if (g % x->slots == 0) order.push_back(g);
        for (int g : gs)
            if (g % x->slots != 0) order.push_back(g);
    }
    const size_t G = order.size();
    size_t n_plain = 0;
    for (int g : order) n_plain += g % x->slots == 0 ? 1 : 0;
    std::vector<std::vector<PtPtr>> pts(G, std::vector<PtPtr>(bidx.size()));
    for (const auto& t : st.terms) {
        const size_t gi = std::find(order.begin(), order.end(), t.giant) - order.begin();
        const size_t bi = std::find(bidx.begin(), bidx.end(), t.baby) - bidx.begin();
        pts[gi][bi] = t.diag;
    }
    // inner sums: the rotated groups of ALL ciphertexts in one block (the batched giant-step key switch takes them as they
    // stand), the unrotated groups in another
    std::vector<CtPtr> slab_plain, slab_rot;
    if (n_plain) slab_plain = ev_.new_ct_batch((int)(B * n_plain), 2, x->ell, 2, 0, x->slots);
    if (G > n_plain) slab_rot = ev_.new_ct_batch((int)(B * (G - n_plain)), 2, x->ell, 2, 0, x->slots);
    std::vector<std::vector<CtPtr>> rows(B), dests(B);
    for (size_t i = 0; i < B; ++i)
        for (size_t k = 0; k < G; ++k)
            dests[i].push_back(k < n_plain ? slab_plain[i * n_plain + k] : slab_rot[i * (G - n_plain) + (k - n_plain)]);
    // ONE pass for the whole batch where the operands are equally spaced (they are: the babies come out of one block per index,
    // the inputs out of one rescale): every diagonal is fetched once for all ciphertexts
    const bool batched = G <= (size_t)EwDotGroups::MAX_G && ev_.dot_plain_groups_batch(brot, pts, pt_scale, dests);
    for (size_t i = 0; i < B; ++i) {
        std::vector<CtPtr>& dest = dests[i];
        if (!batched && !ev_.dot_plain_groups(brot[i], pts, pt_scale, dest)) {
            for (size_t k = 0; k < G; ++k) {
                std::vector<CtPtr> cts;
                std::vector<PtPtr> ps;
                for (size_t b = 0; b < bidx.size(); ++b)
                    if (pts[k][b]) {
                        cts.push_back(brot[i][b]);
                        ps.push_back(pts[k][b]);
                    }
                dest[k] = ev_.dot_plain(cts, ps, pt_scale, dest[k]);
            }
        }
        // ascending giant shift, as apply() hands them to rotate_each_sum
        std::vector<std::pair<int, CtPtr>> byg;
        for (size_t k = 0; k < G; ++k) byg.push_back({order[k], dest[k]});
        std::sort(byg.begin(), byg.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (auto& e : byg) rows[i].push_back(e.second);
    }
    std::vector<int> gidx = order;
    std::sort(gidx.begin(), gidx.end());
    return ev_.rotate_each_sum_rows(rows, gidx);
}

std::vector<CtPtr> Bootstrapper::run_batch(const std::vector<CtPtr>& cts, int drop) {
    if (!ready()) throw Error(FHELIN_ERR_STATE, "EvalBootstrapSetup has not been called");
    Context& c = ev_.ctx();
    if (drop < 0 || c.L + 1 - drop - depth_ < 1) throw Error(FHELIN_ERR_ARG, "bootstrap: level plan leaves no limb for the result");
    const size_t B = cts.size();
    const int top_ell = c.L + 1 - drop;
    // ---- ModRaise (mod_raise() per ciphertext: the integer constant depends on each one's scale)
    std::vector<CtPtr> xs(B);
    {
        std::vector<CtPtr> need;
        std::vector<size_t> pos;
        for (size_t i = 0; i < B; ++i) {
            if (cts[i]->deg >= 2) {
                need.push_back(cts[i]);
                pos.push_back(i);
            } else {
                xs[i] = cts[i];
            }
        }
        if (!need.empty()) {
            std::vector<CtPtr> r = ev_.rescale_batch(need);
            for (size_t k = 0; k < pos.size(); ++k) xs[pos[k]] = r[k];
        }
    }
    std::vector<long double> rho(B);
    const long double q0 = (long double)c.chain.q[0], q1 = (long double)c.chain.q[1];
    const long double target = q0 / (long double)(1ull << correction);
    for (size_t i = 0; i < B; ++i) {
        CtPtr x = xs[i];
        if (x->npoly != 2) throw Error(FHELIN_ERR_STATE, "bootstrap: ciphertext must be relinearised");
        if (x->ell < 2) throw Error(FHELIN_ERR_STATE, "bootstrap: need at least two limbs to set the message scale (bootstrap one level earlier)");
        if (x->ell > 2) x = ev_.level_reduce(x, 2);
        const u64 k0 = (u64)llroundl(target * q1 / x->scale);
        if (k0 < 2) throw Error(FHELIN_ERR_STATE, "bootstrap: ciphertext scale too large for the correction factor");
        xs[i] = ev_.mult_int(x, k0, true, x->scale * (long double)k0);
    }
    xs = ev_.rescale_batch(xs);
    for (size_t i = 0; i < B; ++i) rho[i] = xs[i]->scale * (long double)(1ull << correction) / q0;
    c.stats.bootstrap += B;
    std::vector<CtPtr> w = ev_.raw_modraise_batch(xs, top_ell);
    for (CtPtr& u : w) {
        u->deg = 1;
        u->scale = c.sf_real[0];
        u->slots = slots_;
    }
    const int gapN = (c.N / 2) / slots_;
    for (int j = 1; j < gapN; j <<= 1) {
        const u64 g = c.galois_element(slots_ * j);
        auto it = ev_.rot_keys.find(g);
        if (it == ev_.rot_keys.end()) throw Error(FHELIN_ERR_KEY, "bootstrap: SubSum rotation key missing");
        w = ev_.rotate_galois_batch(w, g, *it->second, true);     // u + sigma_g(u), the sum in the key switch's epilogue
    }
    // ---- CoeffsToSlots, conjugation, EvalMod, SlotsToCoeffs
    for (const auto& st : c2s_) w = apply_batch(st, w);
    if (packed_)
        for (CtPtr& u : w) u->slots = 2 * slots_;
    std::vector<CtPtr> wc = ev_.conjugate_batch(w);
    std::vector<CtPtr> a = ev_.add_batch(w, wc);
    std::vector<CtPtr> v;
    if (packed_) {
        v = eval_mod(a);
    } else {
        std::vector<CtPtr> d = ev_.sub_batch(wc, w), all = a;
        for (const CtPtr& t : d) all.push_back(mult_i(t));
        std::vector<CtPtr> ab = eval_mod(all);
        std::vector<CtPtr> im;
        for (size_t i = 0; i < B; ++i) im.push_back(mult_i(ab[B + i]));
        v = ev_.add_batch(std::vector<CtPtr>(ab.begin(), ab.begin() + B), im);
    }
    for (size_t k = 0; k < s2c_.size(); ++k) {
        v = apply_batch(s2c_[k], v);
        if (packed_ && k == 0)
            for (CtPtr& u : v) u->slots = slots_;
    }
    std::vector<CtPtr> out(B);
    for (size_t i = 0; i < B; ++i) {
        out[i] = ev_.mult_int(v[i], 1ull << correction, false, v[i]->scale);
        out[i]->scale = out[i]->scale * rho[i];
    }
    if (out[0]->deg >= 2) out = ev_.rescale_batch(out);
    for (size_t i = 0; i < B; ++i) out[i]->slots = cts[i]->slots > 0 ? cts[i]->slots : slots_;
    return out;
}

std::vector<CtPtr> Bootstrapper::bootstrap_batch(const std::vector<CtPtr>& cts, int drop) {
    if (cts.empty()) return {};
    if (cts.size() == 1) return {run(cts[0], 0, drop)};
    return run_batch(cts, drop);
}

CtPtr Bootstrapper::bootstrap(const CtPtr& ct, int drop) { return run(ct, 0, drop); }
CtPtr Bootstrapper::partial(const CtPtr& ct, int stage) { return run(ct, stage); }

}  // namespace fhelin
