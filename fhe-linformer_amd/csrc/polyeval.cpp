// Polynomial evaluation on ciphertexts: real-constant multiply/add, power-basis EvalPoly, EvalMultMany and
// Chebyshev-series evaluation (baby-step/giant-step with the identity T_i = 2 T_m T_{i-m} - T_{2m-i}).
// Mirrors what the reference obtains from OpenFHE's EvalPoly / EvalMultMany / EvalChebyshevFunction
// (reference src/FHEController.cpp:1289-1336).  All arithmetic runs through the GPU evaluator ops.
#include <cmath>
#include <map>
#include "evaluator.h"

namespace fhelin {

void real_to_scalars(const Context& c, long double v, int ell, ScalarSet& sc) {
    const bool neg = v < 0;
    const long double mag = roundl(fabsl(v));
    const long double two64 = 18446744073709551616.0L;
    const u64 hi = (u64)floorl(mag / two64);
    const u64 lo = (u64)(mag - (long double)hi * two64);
    for (int i = 0; i < ell; ++i) {
        const u64 q = c.chain.q[i];
        u64 r = (u64)((((u128)(hi % q) << 64) | lo) % q);
        if (neg) r = neg_mod(r, q);
        sc.v[2 * i] = r;
        sc.v[2 * i + 1] = h_shoup(r, q);
    }
}

CtPtr Evaluator::mult_real(const CtPtr& a, double cst) {
    CtPtr x = a->deg >= 2 ? rescale(a) : a;
    const long double sf = c_.sf_real[x->level()];
    ScalarSet sc;
    real_to_scalars(c_, (long double)cst * sf, x->ell, sc);
    CtPtr o = new_ct(x->npoly, x->ell, x->deg + 1, x->scale * sf, x->slots);
    launch_ew_scalar(c_.dt, o->d, x->d, sc, x->npoly * x->ell, 0, x->ell, c_.stream);
    hip_check(hipGetLastError(), "mult_real");
    return o;
}

CtPtr Evaluator::add_real(const CtPtr& a, double cst) {
    ScalarSet sc;
    real_to_scalars(c_, (long double)cst * a->scale, a->ell, sc);
    // one pass: component 0 gets the constant, the other components are copied through (no separate copy of the ciphertext)
    CtPtr o = new_ct(a->npoly, a->ell, a->deg, a->scale, a->slots);
    launch_ew_addscalar(c_.dt, o->d, a->d, sc, a->npoly * a->ell, 0, a->ell, c_.stream, a->ell);
    hip_check(hipGetLastError(), "add_real");
    return o;
}

CtPtr Evaluator::lincomb(const std::vector<CtPtr>& terms_in, const std::vector<double>& coef_in, double c0) {
    return lincomb_at(terms_in, coef_in, c0, 0, 0);
}

// out_scale > 0: the coefficients are scaled so that the result has exactly this scale (instead of the level's own factor),
// keep_ell > 0: on the first keep_ell limbs only.  A summand that is going to be added to a product of known (limbs, scale) is
// born there and needs no level adjustment (scalar multiply + rescale) afterwards.  Null if the operands do not fit the kernel.
CtPtr Evaluator::lincomb_at(const std::vector<CtPtr>& terms_in, const std::vector<double>& coef_in, double c0, long double want_scale,
                            int keep_ell) {
    std::vector<CtPtr> terms;
    std::vector<double> coef;
    for (size_t i = 0; i < terms_in.size(); ++i)
        if (coef_in[i] != 0.0) {
            terms.push_back(terms_in[i]);
            coef.push_back(coef_in[i]);
        }
    if (terms.empty()) throw Error(FHELIN_ERR_ARG, "lincomb: all coefficients are zero");
    bool uniform = (int)terms.size() <= LinComb::MAX_TERMS && terms.size() >= 2;
    for (const CtPtr& t : terms)
        uniform = uniform && t->deg == 1 && t->npoly == terms[0]->npoly && t->ell == terms[0]->ell &&
                  fabsl(t->scale / terms[0]->scale - 1.0L) < 1e-9L;
    if (!uniform) {  // the chain of single operations (also the definition of the result)
        if (want_scale > 0 || keep_ell > 0) return CtPtr();
        CtPtr acc;
        for (size_t i = 0; i < terms.size(); ++i) {
            CtPtr t = mult_real(terms[i], coef[i]);
            acc = acc ? add(acc, t) : t;
        }
        return c0 != 0.0 ? add_real(acc, c0) : acc;
    }
    const CtPtr& x = terms[0];
    const int ell = keep_ell > 0 ? std::min(keep_ell, x->ell) : x->ell, n = (int)terms.size();
    const long double sf = want_scale > 0 ? want_scale / x->scale : c_.sf_real[x->level()];
    const long double out_scale = x->scale * sf;
    std::vector<u64> sc((size_t)(n + 1) * ell, 0);
    ScalarSet one;
    for (int k = 0; k < n; ++k) {
        real_to_scalars(c_, (long double)coef[k] * sf, ell, one);
        for (int l = 0; l < ell; ++l) sc[(size_t)k * ell + l] = one.v[2 * l];
    }
    if (c0 != 0.0) {
        real_to_scalars(c_, (long double)c0 * out_scale, ell, one);
        for (int l = 0; l < ell; ++l) sc[(size_t)n * ell + l] = one.v[2 * l];
    }
    u64* dsc = c_.dalloc<u64>(sc.size());
    c_.upload_async(dsc, sc.data(), sc.size());
    LinComb lc;
    lc.n = n;
    lc.vecs = x->npoly * ell;
    for (int k = 0; k < n; ++k) lc.a[k] = terms[k]->d;
    CtPtr o = new_ct(x->npoly, ell, 2, out_scale, x->slots);
    launch_ew_lincomb(c_.dt, o->d, lc, dsc, ell, c_.stream, ell < x->ell ? x->ell : 0);
    hip_check(hipGetLastError(), "lincomb");
    c_.pool.free(dsc);
    return o;
}

CtPtr Evaluator::conjugate(const CtPtr& a) {
    if (!conj_key) throw Error(FHELIN_ERR_KEY, "no conjugation key");
    return raw_rotate(a, 2ull * c_.N - 1, *conj_key);
}

CtPtr Evaluator::mult_many(const std::vector<CtPtr>& v) {
    if (v.empty()) throw Error(FHELIN_ERR_ARG, "mult_many: empty vector");
    return mult_many_rows(std::vector<std::vector<CtPtr>>{v})[0];
}

// EvalMultMany on several operand lists of ONE length at once (the same call on several samples): every level of the product
// tree is one batched relinearisation over all lists.  A list's result holds the residues of the tree evaluated alone.
std::vector<CtPtr> Evaluator::mult_many_rows(const std::vector<std::vector<CtPtr>>& vs) {
    if (vs.empty()) return {};
    const size_t X = vs.size();
    for (const auto& v : vs)
        if (v.empty() || v.size() != vs[0].size()) throw Error(FHELIN_ERR_ARG, "mult_many: lists must be non-empty and of one length");
    std::vector<std::vector<CtPtr>> cur = vs;
    while (cur[0].size() > 1) {
        const size_t n = cur[0].size();
        // identical operand pairs give identical products: reuse (EvalMultMany({r,r,...}) squares repeatedly) - when EVERY list repeats there
        std::vector<char> reuse(n / 2, 0);
        for (size_t i = 2; i + 1 < n; i += 2) {
            bool all = true;
            for (size_t x = 0; x < X; ++x) all = all && cur[x][i] == cur[x][i - 2] && cur[x][i + 1] == cur[x][i - 1];
            reuse[i / 2] = all ? 1 : 0;
        }
        std::vector<CtPtr> lhs, rhs;
        for (size_t i = 0; i + 1 < n; i += 2)
            if (!reuse[i / 2])
                for (size_t x = 0; x < X; ++x) {
                    lhs.push_back(cur[x][i]);
                    rhs.push_back(cur[x][i + 1]);
                }
        const std::vector<CtPtr> prod = lhs.size() == 1 ? std::vector<CtPtr>{mult(lhs[0], rhs[0])} : mult_batch(lhs, rhs);
        std::vector<std::vector<CtPtr>> nxt(X);
        size_t p = 0;
        for (size_t i = 0; i + 1 < n; i += 2) {
            if (reuse[i / 2]) {
                for (size_t x = 0; x < X; ++x) nxt[x].push_back(nxt[x].back());
            } else {
                for (size_t x = 0; x < X; ++x) nxt[x].push_back(prod[p * X + x]);
                ++p;
            }
        }
        if (n & 1)
            for (size_t x = 0; x < X; ++x) nxt[x].push_back(cur[x].back());
        cur.swap(nxt);
    }
    std::vector<CtPtr> out(X);
    for (size_t x = 0; x < X; ++x) out[x] = cur[x][0];
    return out;
}

// bring a set of ciphertexts to one common (level, degree 1)
static void align_deg1(Evaluator& ev, std::vector<CtPtr>& v, int from) {
    int ell = 1 << 30;
    for (size_t i = from; i < v.size(); ++i) {
        if (v[i]->deg >= 2) v[i] = ev.rescale(v[i]);
        ell = std::min(ell, v[i]->ell);
    }
    const long double sf = ev.ctx().sf_real[ev.ctx().L + 1 - ell];
    // all the powers that sit above the common level go down together: one batched rescale (Evaluator::adjust_deg1_batch)
    std::vector<CtPtr> r = ev.adjust_deg1_batch(std::vector<CtPtr>(v.begin() + from, v.end()), ell, sf);
    for (size_t i = from; i < v.size(); ++i) v[i] = r[i - from];
}
// the same for several such sets (one per input of a batched evaluation): each set to ITS OWN common level - and the sets that end
// at one level (all of them, for the inputs of one driver call) through one batched rescale and one batched adjustment.  A set's
// members hold the residues align_deg1 gives for that set alone.
static void align_deg1_cols(Evaluator& ev, std::vector<std::vector<CtPtr>>& cols, int from) {
    if (cols.size() == 1) {
        align_deg1(ev, cols[0], from);
        return;
    }
    {
        std::vector<CtPtr> need;
        std::vector<std::pair<size_t, size_t>> pos;
        for (size_t c = 0; c < cols.size(); ++c)
            for (size_t i = from; i < cols[c].size(); ++i)
                if (cols[c][i]->deg >= 2) {
                    need.push_back(cols[c][i]);
                    pos.push_back({c, i});
                }
        if (!need.empty()) {
            const std::vector<CtPtr> r = ev.rescale_batch(need);
            for (size_t k = 0; k < pos.size(); ++k) cols[pos[k].first][pos[k].second] = r[k];
        }
    }
    std::map<int, std::vector<size_t>> by_ell;
    for (size_t c = 0; c < cols.size(); ++c) {
        int ell = 1 << 30;
        for (size_t i = from; i < cols[c].size(); ++i) ell = std::min(ell, cols[c][i]->ell);
        by_ell[ell].push_back(c);
    }
    for (const auto& g : by_ell) {
        const long double sf = ev.ctx().sf_real[ev.ctx().L + 1 - g.first];
        std::vector<CtPtr> all;
        for (size_t c : g.second) all.insert(all.end(), cols[c].begin() + from, cols[c].end());
        const std::vector<CtPtr> r = ev.adjust_deg1_batch(all, g.first, sf);
        size_t p = 0;
        for (size_t c : g.second)
            for (size_t i = from; i < cols[c].size(); ++i) cols[c][i] = r[p++];
    }
}

CtPtr Evaluator::eval_poly(const CtPtr& x, const std::vector<double>& coeffs) { return eval_poly_many(std::vector<CtPtr>{x}, coeffs)[0]; }

// EvalPoly on several ciphertexts at once: the powers h < i <= 2h only need powers <= h, so each doubling round is ONE batched
// multiplication over all its i and all inputs.  Every input's result holds the residues of the power tree evaluated alone.
std::vector<CtPtr> Evaluator::eval_poly_many(const std::vector<CtPtr>& xs, const std::vector<double>& coeffs) {
    int n = (int)coeffs.size() - 1;
    while (n > 0 && coeffs[n] == 0.0) --n;
    if (n < 1) throw Error(FHELIN_ERR_ARG, "eval_poly: need degree >= 1");
    const size_t X = xs.size();
    if (!X) return {};
    std::vector<CtRow> pw(n + 1, CtRow(X));
    {
        std::vector<CtPtr> need;
        for (size_t x = 0; x < X; ++x)
            if (xs[x]->deg >= 2) need.push_back(xs[x]);
        const std::vector<CtPtr> r = need.empty() ? std::vector<CtPtr>() : rescale_batch(need);
        size_t p = 0;
        for (size_t x = 0; x < X; ++x) pw[1][x] = xs[x]->deg >= 2 ? r[p++] : xs[x];
    }
    for (int h = 1; h < n; h *= 2) {
        const int i_hi = std::min(2 * h, n);
        CtRow lhs, rhs;
        for (int i = h + 1; i <= i_hi; ++i)
            for (size_t x = 0; x < X; ++x) {
                lhs.push_back(pw[h][x]);                                  // x^i = x^h * x^(i-h), h the largest power of two <= i
                rhs.push_back(i == 2 * h ? pw[h][x] : pw[i - h][x]);
            }
        const CtRow prod = lhs.size() == 1 ? CtRow{mult(lhs[0], rhs[0])} : mult_batch(lhs, rhs);
        size_t p = 0;
        for (int i = h + 1; i <= i_hi; ++i)
            for (size_t x = 0; x < X; ++x) pw[i][x] = prod[p++];
    }
    std::vector<CtPtr> out(X);
    const std::vector<double> cf(coeffs.begin() + 1, coeffs.begin() + n + 1);
    std::vector<std::vector<CtPtr>> cols(X, std::vector<CtPtr>(n + 1));
    for (size_t x = 0; x < X; ++x)
        for (int i = 1; i <= n; ++i) cols[x][i] = pw[i][x];
    align_deg1_cols(*this, cols, 1);
    for (size_t x = 0; x < X; ++x) out[x] = lincomb(std::vector<CtPtr>(cols[x].begin() + 1, cols[x].begin() + n + 1), cf, coeffs[0]);
    return out;
}

// Chebyshev evaluation over a ROW of ciphertexts (one value per input): every op below acts on all inputs at once, the
// products through mult_batch (one batched relinearisation).
//
// The Paterson-Stockmeyer recursion p = q * T_m + r is a tree whose products are independent of one another except along
// a q-chain: the tree is built first and then evaluated in ROUNDS - all products whose q operand is ready go through ONE
// mult_batch (degree 47: 2 rounds instead of 5 dependent relinearisations, degree 119: 3 instead of 7, degree 300: 5 instead
// of 18), then the sums that have become ready.  Each node computes exactly what the recursive form computed
// (mult(q, T_m), then add(., r)), so the residues do not depend on the order.
namespace {
struct ChebNode {
    std::vector<double> c;
    int m = 0;           // 0: a leaf (degree < baby), evaluated as one linear combination of the baby powers
    int q = -1, r = -1;
    std::vector<CtPtr> val, prod;
    bool has_prod = false, done = false;
};
int cheb_build(std::vector<ChebNode>& nodes, const std::vector<double>& c_in, int baby) {
    std::vector<double> c = c_in;
    int n = (int)c.size() - 1;
    while (n > 0 && c[n] == 0.0) --n;
    c.resize(n + 1);
    const int id = (int)nodes.size();
    nodes.emplace_back();
    nodes[id].c = c;
    if (n < baby) return id;
    int m = baby;
    while (m * 2 <= n) m *= 2;
    std::vector<double> q(n - m + 1, 0.0), r(m, 0.0);
    for (int i = 0; i < m; ++i) r[i] = c[i];
    q[0] = c[m];
    for (int i = m + 1; i <= n; ++i) {
        q[i - m] = 2 * c[i];
        r[2 * m - i] -= c[i];
    }
    nodes[id].m = m;
    const int qi = cheb_build(nodes, q, baby);
    nodes[id].q = qi;
    bool r_zero = true;
    for (double v : r) r_zero = r_zero && v == 0.0;
    if (!r_zero) {
        const int ri = cheb_build(nodes, r, baby);
        nodes[id].r = ri;
    }
    return id;
}
}  // namespace

Evaluator::CtRow Evaluator::cheb_recurse(const std::vector<double>& c, const std::vector<CtRow>& Traw, const std::map<int, CtRow>& G,
                                         int baby) {
    const size_t rows = Traw[1].size();
    std::vector<ChebNode> nodes;
    const int root = cheb_build(nodes, c, baby);
    // A leaf sum_{k<=d} c_k T_k only needs the babies T_1..T_d at ONE level - that of the deepest power among them, depth ceil(log2 d) -
    // not at the level of T_(baby-1): the leaves at the bottom of the quotient chain p = q T_m + r (degree < baby / 2) then sit one level
    // higher, and the whole evaluation takes ceil(log2(degree)) levels for the powers + 1 for the constants - OpenFHE's Paterson-Stockmeyer
    // depth (7 for degree 119, 9 for 300; all babies at one level costs one more).  The babies are aligned per CLASS j (degrees in
    // (2^(j-1), 2^j] use T_1..T_min(2^j, baby-1) brought to their common level), lazily, all inputs of the batch together.
    std::map<int, std::vector<CtRow>> aligned;   // class -> [k][input], k = 1..prefix
    auto class_of = [](int d) {
        int j = 0;
        while ((1 << j) < d) ++j;
        return j;
    };
    auto babies_for = [&](int d) -> const std::vector<CtRow>& {
        const int j = cheb_leaf_classes ? class_of(std::max(d, 1)) : 30;
        const int prefix = std::min(j >= 30 ? baby - 1 : (1 << j), baby - 1);
        auto it = aligned.find(prefix);
        if (it != aligned.end()) return it->second;
        std::vector<std::vector<CtPtr>> cols(rows, std::vector<CtPtr>(prefix + 1));
        for (size_t i = 0; i < rows; ++i)
            for (int k = 1; k <= prefix; ++k) cols[i][k] = Traw[k][i];
        align_deg1_cols(*this, cols, 1);
        std::vector<CtRow> out(prefix + 1, CtRow(rows));
        for (size_t i = 0; i < rows; ++i)
            for (int k = 1; k <= prefix; ++k) out[k][i] = cols[i][k];
        return aligned.emplace(prefix, std::move(out)).first->second;
    };
    // a leaf that is the r of p = q T_m + r is only ever ADDED to the product q T_m: it is evaluated when that product is known,
    // with its coefficients scaled so that it is born with the product's limbs and scale (no level adjustment = no rescale)
    std::vector<char> is_r_leaf(nodes.size(), 0);
    for (const ChebNode& nd : nodes)
        if (nd.m && nd.r >= 0 && !nodes[nd.r].m) {
            bool any = false;
            for (size_t k = 1; k < nodes[nd.r].c.size(); ++k) any = any || nodes[nd.r].c[k] != 0.0;
            is_r_leaf[nd.r] = any && cheb_leaf_at_product ? 1 : 0;
        }
    auto leaf_terms = [&](const ChebNode& nd, size_t i) {
        const std::vector<CtRow>& T = babies_for((int)nd.c.size() - 1);
        std::vector<CtPtr> terms;
        for (size_t k = 1; k < nd.c.size(); ++k) terms.push_back(T[k][i]);
        return terms;
    };
    // leaves: sum_k c_k T_k + c_0 in one pass per input (the babies share one level: align_deg1)
    for (size_t id = 0; id < nodes.size(); ++id) {
        ChebNode& nd = nodes[id];
        if (nd.m || is_r_leaf[id]) continue;
        const int n = (int)nd.c.size() - 1;
        bool any = false;
        for (int k = 1; k <= n; ++k) any = any || nd.c[k] != 0.0;
        nd.val.resize(rows);
        const std::vector<CtRow>& T = babies_for(any ? n : baby - 1);   // a constant leaf hangs on T_1 at the level of all babies
        for (size_t i = 0; i < rows; ++i) {
            if (!any) {  // constant polynomial: c0 as an encryption-free shift of 0 * T_1
                nd.val[i] = mult_real(T[1][i], 0.0);
                if (nd.c[0] != 0.0) nd.val[i] = add_real(nd.val[i], nd.c[0]);
                continue;
            }
            std::vector<CtPtr> terms;
            for (int k = 1; k <= n; ++k) terms.push_back(T[k][i]);
            nd.val[i] = lincomb(terms, std::vector<double>(nd.c.begin() + 1, nd.c.begin() + n + 1), nd.c[0]);
        }
        nd.done = true;
    }
    while (!nodes[root].done) {
        // one batched product for every node whose q operand is ready
        std::vector<int> ready;
        CtRow lhs, rhs;
        for (int id = 0; id < (int)nodes.size(); ++id) {
            ChebNode& nd = nodes[id];
            if (!nd.m || nd.has_prod || !nodes[nd.q].done) continue;
            if (!cheb_rounds && !ready.empty()) break;   // the dependent, one-product-at-a-time order (A/B and parity test)
            ready.push_back(id);
            const CtRow& g = G.at(nd.m);
            for (size_t i = 0; i < rows; ++i) {
                lhs.push_back(nodes[nd.q].val[i]);
                rhs.push_back(g[i]);
            }
        }
        if (ready.empty()) throw Error(FHELIN_ERR_INTERNAL, "chebyshev evaluation: no product is ready");
        CtRow prod = mult_batch(lhs, rhs);
        for (size_t k = 0; k < ready.size(); ++k) {
            ChebNode& nd = nodes[ready[k]];
            nd.prod.assign(prod.begin() + k * rows, prod.begin() + (k + 1) * rows);
            nd.has_prod = true;
            nodes[nd.q].val.clear();
        }
        // the sums that are ready now (a finished sum can complete its parent's: repeat until nothing moves)
        for (bool moved = true; moved;) {
            moved = false;
            for (ChebNode& nd : nodes) {
                if (nd.done || !nd.has_prod) continue;
                if (nd.r >= 0 && is_r_leaf[nd.r] && !nodes[nd.r].done) {
                    ChebNode& lf = nodes[nd.r];
                    const std::vector<double> cf(lf.c.begin() + 1, lf.c.end());
                    lf.val.resize(rows);
                    for (size_t i = 0; i < rows; ++i) {
                        const CtPtr& pr = nd.prod[i];
                        CtPtr v = pr->deg == 2 ? lincomb_at(leaf_terms(lf, i), cf, lf.c[0], pr->scale, pr->ell) : CtPtr();
                        lf.val[i] = v ? v : lincomb(leaf_terms(lf, i), cf, lf.c[0]);
                    }
                    lf.done = true;
                }
                if (nd.r >= 0 && !nodes[nd.r].done) continue;
                nd.val = nd.r >= 0 ? add_batch(nd.prod, nodes[nd.r].val) : nd.prod;
                if (nd.r >= 0) nodes[nd.r].val.clear();
                nd.prod.clear();
                nd.done = true;
                moved = true;
            }
        }
    }
    return nodes[root].val;
}

CtPtr Evaluator::eval_chebyshev(const CtPtr& x, const std::vector<double>& coeffs_in, double a, double b) {
    return eval_chebyshev_many(std::vector<CtPtr>{x}, coeffs_in, a, b)[0];
}

std::vector<CtPtr> Evaluator::eval_chebyshev_many(const std::vector<CtPtr>& xs, const std::vector<double>& coeffs_in, double a, double b) {
    if (xs.empty()) return {};
    std::vector<double> c = coeffs_in;
    int n = (int)c.size() - 1;
    while (n > 0 && c[n] == 0.0) --n;
    c.resize(n + 1);
    if (n < 1) throw Error(FHELIN_ERR_ARG, "eval_chebyshev: need degree >= 1");
    c[0] *= 0.5;  // series convention: c0/2 + sum_{k>=1} c_k T_k  (OpenFHE EvalChebyshevSeries)
    const size_t rows = xs.size();
    // affine map of [a,b] onto [-1,1]
    CtRow u(rows);
    for (size_t i = 0; i < rows; ++i) {
        u[i] = xs[i];
        if (!(a == -1.0 && b == 1.0)) {
            u[i] = mult_real(xs[i], 2.0 / (b - a));
            u[i] = add_real(u[i], -(a + b) / (b - a));
        }
    }
    {
        std::vector<CtPtr> need;
        for (size_t i = 0; i < rows; ++i)
            if (u[i]->deg >= 2) need.push_back(u[i]);
        if (!need.empty()) {
            const std::vector<CtPtr> r = rescale_batch(need);   // (a single one goes through rescale() itself)
            size_t p = 0;
            for (size_t i = 0; i < rows; ++i)
                if (u[i]->deg >= 2) u[i] = r[p++];
        }
    }
    int l = 0;
    while ((1 << (2 * l)) < n + 1) ++l;  // baby = 2^ceil(log2(n+1)/2)
    const int baby = std::max(2, 1 << l);
    std::vector<CtRow> T(baby + 1);
    T[1] = u;
    // T_k = 2 T_floor(k/2) T_ceil(k/2) - T_(k mod 2): the powers h < k <= 2h only need powers <= h, so each doubling round
    // is ONE batched multiplication over all its k and all inputs
    for (int h = 1; h < baby; h *= 2) {
        const int k_hi = std::min(2 * h, baby);
        CtRow lhs, rhs;
        for (int k = h + 1; k <= k_hi; ++k)
            for (size_t i = 0; i < rows; ++i) {
                lhs.push_back(T[k / 2][i]);
                rhs.push_back(T[k - k / 2][i]);
            }
        CtRow t;
        size_t p = 0;
        if (merged_products) {
            // rescale(2 T_j T_k - 1) / rescale(2 T_j T_k - T_1) in one go each (mult_affine_rescale_batch): the even and the odd powers of
            // the round as one batched call each
            CtRow ea, eb, oa, ob, osub;
            std::vector<size_t> epos, opos;
            for (int k = h + 1; k <= k_hi; ++k)
                for (size_t i = 0; i < rows; ++i, ++p) {
                    if (k % 2 == 0) {
                        ea.push_back(lhs[p]);
                        eb.push_back(rhs[p]);
                        epos.push_back(p);
                    } else {
                        oa.push_back(lhs[p]);
                        ob.push_back(rhs[p]);
                        osub.push_back(T[1][i]);
                        opos.push_back(p);
                    }
                }
            t.resize(lhs.size());
            if (!ea.empty()) {
                CtRow r = mult_affine_rescale_batch(ea, eb, 2, -1.0, {});
                for (size_t j = 0; j < epos.size(); ++j) t[epos[j]] = r[j];
            }
            if (!oa.empty()) {
                CtRow r = mult_affine_rescale_batch(oa, ob, 2, 0.0, osub);
                for (size_t j = 0; j < opos.size(); ++j) t[opos[j]] = r[j];
            }
        } else {
            t = mult_batch(lhs, rhs);
            t = add_batch(t, t);
            CtRow odd_a, odd_b;
            std::vector<size_t> odd_pos;
            for (int k = h + 1; k <= k_hi; ++k)
                for (size_t i = 0; i < rows; ++i, ++p) {
                    if (k % 2 == 0) {
                        t[p] = add_real(t[p], -1.0);
                    } else {
                        odd_a.push_back(t[p]);
                        odd_b.push_back(T[1][i]);
                        odd_pos.push_back(p);
                    }
                }
            if (!odd_a.empty()) {
                CtRow d = sub_batch(odd_a, odd_b);
                for (size_t j = 0; j < odd_pos.size(); ++j) t[odd_pos[j]] = d[j];
            }
            t = rescale_batch(t);
        }
        p = 0;
        for (int k = h + 1; k <= k_hi; ++k) {
            T[k].resize(rows);
            for (size_t i = 0; i < rows; ++i, ++p) T[k][i] = t[p];
        }
    }
    std::map<int, CtRow> G;
    G[baby] = T[baby];
    for (int m = baby; m * 2 <= n; m *= 2) {
        if (merged_products) {
            G[2 * m] = mult_affine_rescale_batch(G[m], G[m], 2, -1.0, {});
            continue;
        }
        CtRow t = mult_batch(G[m], G[m]);
        t = add_batch(t, t);
        for (size_t i = 0; i < rows; ++i) t[i] = add_real(t[i], -1.0);
        G[2 * m] = rescale_batch(t);
    }
    // the baby powers as they are: cheb_recurse brings the ones a leaf needs to that leaf's level
    std::vector<CtRow> babies(T.begin(), T.begin() + baby);   // index 0 unused
    return cheb_recurse(c, babies, G, baby);
}

}  // namespace fhelin
