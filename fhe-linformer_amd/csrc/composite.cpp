#include "composite.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace fhelin {

static bool env_flag(const char* name) {  // unset or non-zero = on
    const char* e = std::getenv(name);
    return !e || std::atoi(e) != 0;
}

Composite::Composite(Evaluator& ev, Client& cl) : ev_(ev), cl_(cl) {
    early_rescale_ = env_flag("FHELIN_EARLY_RESCALE");  // read per context, so that one process can hold both kinds
    merge_rot_ = env_flag("FHELIN_MERGE_ROT");
    if (const char* e = std::getenv("FHELIN_ROW_LANES")) row_lanes_ = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_CHEB_ROUNDS")) ev_.cheb_rounds = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_CHEB_LEAF_CLASSES")) ev_.cheb_leaf_classes = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_DOT_GROUPS")) ev_.dot_groups = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_DOUBLE_HOIST")) ev_.double_hoist = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_FUSE_RELARGE")) fuse_relarge = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_BULK_UNWRAP")) bulk_unwrap = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_MERGED_RESCALE")) ev_.merged_rescale = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_MERGED_PRODUCTS")) ev_.merged_products = std::atoi(e) != 0;
    if (const char* e = std::getenv("FHELIN_CHEB_LEAF_AT")) ev_.cheb_leaf_at_product = std::atoi(e) != 0;
    if (const char* b = std::getenv("FHELIN_BATCH")) {
        int v = std::atoi(b);
        if (v >= 1 && v <= 256) ev_.batch_limit = v;
    }
}

PtPtr Composite::encode_vec(const std::vector<double>& v, int level) {
    return cl_.encode(v.data(), (int)v.size(), level, num_slots());
}

PtPtr Composite::encode_const(double val, int level) {
    std::vector<double> v(num_slots(), val);
    return encode_vec(v, level);
}

CtPtr Composite::mult_const(const CtPtr& c, double d) {
    char key[64];
    uint64_t bits;
    std::memcpy(&bits, &d, 8);
    snprintf(key, sizeof key, "const:%016llx", (unsigned long long)bits);
    auto it = mask_cache_.find(key);
    PtPtr p = it != mask_cache_.end() ? it->second : (mask_cache_[key] = encode_const(d, c->level()));
    return ev_.mult_plain(c, p);
}

PtPtr Composite::mask_plain(const std::string& key, const std::vector<double>& v) {
    auto it = mask_cache_.find(key);
    if (it != mask_cache_.end()) return it->second;
    PtPtr p = encode_vec(v, 0);
    mask_cache_[key] = p;
    return p;
}

static std::string mkey(const char* kind, long a, long b, double v) {
    char key[96];
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    snprintf(key, sizeof key, "%s:%ld:%ld:%016llx", kind, a, b, (unsigned long long)bits);
    return key;
}

PtPtr Composite::block_mask(int from, int to, double val) {
    const std::string key = mkey("block", from, to, val);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = std::max(from, 0); i < to && i < num_slots(); ++i) m[i] = val;
        mask_plain(key, m);
    }
    return mask_cache_[key];
}
CtPtr Composite::mask_block(const CtPtr& c, int from, int to, double val) { return ev_.mult_plain(c, block_mask(from, to, val)); }

CtPtr Composite::mask_heads(const CtPtr& c, double val) {
    const std::string key = mkey("mod", 64, 0, val);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = 0; i < num_slots(); i += 64) m[i] = val;
        mask_plain(key, m);
    }
    return ev_.mult_plain(c, mask_cache_[key]);
}

PtPtr Composite::heads_128_mask(double val) {
    const std::string key = mkey("mod", 128, 0, val);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = 0; i < num_slots(); i += 128) m[i] = val;
        mask_plain(key, m);
    }
    return mask_cache_[key];
}
CtPtr Composite::mask_heads_128(const CtPtr& c, double val) { return ev_.mult_plain(c, heads_128_mask(val)); }

PtPtr Composite::mod_n_mask(int n, int padding) {
    if (n <= 0) throw Error(FHELIN_ERR_ARG, "mask_mod_n: n must be positive");
    const std::string key = mkey("mod", n, padding, 1.0);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = 0; i < num_slots(); ++i)
            if (i % n == padding) m[i] = 1.0;
        mask_plain(key, m);
    }
    return mask_cache_[key];
}
CtPtr Composite::mask_mod_n(const CtPtr& c, int n, int padding) { return ev_.mult_plain(c, mod_n_mask(n, padding)); }

PtPtr Composite::mod_range_mask(int period, int from, int to, int span_from, int span_to) {
    if (span_to < 0) span_to = num_slots();
    const std::string key = mkey("modrange", period, (long)from * 100000 + to, 1.0) + "/" + std::to_string(span_from) + "-" + std::to_string(span_to);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = std::max(span_from, 0); i < span_to && i < num_slots(); ++i)
            if (i % period >= from && i % period < to) m[i] = 1.0;
        mask_plain(key, m);
    }
    return mask_cache_[key];
}

PtPtr Composite::first_n_mask(int n, double val) {
    const std::string key = mkey("first", n, 0, val);
    if (!mask_cache_.count(key)) {
        std::vector<double> m(num_slots(), 0.0);
        for (int i = 0; i < n && i < num_slots(); ++i) m[i] = val;
        mask_plain(key, m);
    }
    return mask_cache_[key];
}

CtPtr Composite::mask_first_n(const CtPtr& c, int n, double val) { return ev_.mult_plain(c, first_n_mask(n, val)); }

// the reference loops `for (int i = 0; i < log2(slots); i++)` with a floating-point bound (:832,:842,:852,:862)
static int log_steps(int slots) {
    int n = 0;
    while ((double)n < std::log2((double)slots)) ++n;
    return n;
}

// A degree-2 input (a fresh product) is rescaled BEFORE the rotations instead of lazily before its next multiplication:
// the same single rescale, but the key switches of the tree then run with one limb fewer (EvalRotate and ModReduce
// commute up to rounding noise; FHELIN_EARLY_RESCALE=0 restores the reference's order).


// Two consecutive steps of a rotate-and-sum tree, x += rot(x, s); x += rot(x, 2s), equal x + rot(x,s) + rot(x,2s) +
// rot(x,3s): with the key for 3s they run as ONE merged key switch (one ModUp, three gathered inner products
// accumulated in the extended basis, one ModDown: half the NTTs of two key switches).  The key set a client generates
// through this library contains the 3*2^i rotations for that purpose; without them the tree falls back to single steps.
// FHELIN_MERGE_ROT=0 disables.

CtVec Composite::tree_steps(CtVec r, int n, int unit) {
    int i = 0;
    while (i < n) {
        const int s = unit * (1 << i);
        // three steps x += rot(x,s); x += rot(x,2s); x += rot(x,4s) equal x + sum_{m=1..7} rot(x, m s): ONE merged key switch
        // with the keys of s..7s (measured at N=2^16, 28+7 limbs: a 7-step tree as 3+3+1 takes 0.85x the time of 2+2+2+1 at 12
        // limbs, 0.96x at 27: profiles/r02_m_radix_probe.txt); two steps with the keys of s, 2s, 3s; else one plain step
        if (i + 2 < n && merge_rot_ && ev_.have_rotation_keys({s, 2 * s, 3 * s, 4 * s, 5 * s, 6 * s, 7 * s}, r[0]->slots)) {
            r = ev_.rotate_sum_batch(r, {s, 2 * s, 3 * s, 4 * s, 5 * s, 6 * s, 7 * s});
            i += 3;
        } else if (i + 1 < n && merge_rot_ && ev_.have_rotation_keys({s, 2 * s, 3 * s}, r[0]->slots)) {
            r = ev_.rotate_sum_batch(r, {s, 2 * s, 3 * s});
            i += 2;
        } else {
            r = ev_.rotate_add_batch(r, s);
            i += 1;
        }
    }
    return r;
}

CtPtr Composite::rotsum(const CtPtr& in, int slots, int padding) {
    const int n = log_steps(slots);
    CtPtr r = (n && early_rescale_ && in->deg >= 2 && in->ell >= 2) ? ev_.rescale(in) : in;  // immutable handles: no Clone()
    return n ? tree_steps(CtVec{r}, n, padding)[0] : ev_.clone(in);
}

CtPtr Composite::rotsum_padded(const CtPtr& in, int slots) { return rotsum(in, slots, slots); }

CtPtr Composite::repeat(const CtPtr& in, int slots, int padding) {
    const int n = log_steps(slots);
    CtPtr r = (n && early_rescale_ && in->deg >= 2 && in->ell >= 2) ? ev_.rescale(in) : in;
    return n ? tree_steps(CtVec{r}, n, -padding)[0] : ev_.clone(in);
}

// log-tree rotate-and-add over independent rows.  Rows are cut into chunks of `batch_limit`; every chunk runs its
// whole chain (one batched key switch per step) on its own lane, so chunks overlap on the GPU.
CtVec Composite::tree_batch(const CtVec& in_raw, int slots, int step_sign, int padding) {
    const int n = log_steps(slots);
    CtVec in = in_raw;
    if (n && early_rescale_) {  // see rotsum(): one limb fewer for every key switch of the tree
        CtVec need;
        std::vector<size_t> pos;
        for (size_t i = 0; i < in.size(); ++i)
            if (in[i]->deg >= 2 && in[i]->ell >= 2) {
                need.push_back(in[i]);
                pos.push_back(i);
            }
        if (!need.empty()) {
            CtVec r = ev_.rescale_batch(need);
            for (size_t k = 0; k < pos.size(); ++k) in[pos[k]] = r[k];
        }
    }
    CtVec out(in.size());
    if (!n) {
        for (size_t i = 0; i < in.size(); ++i) out[i] = ev_.clone(in[i]);
        return out;
    }
    Context& c = ev_.ctx();
    const size_t B = (size_t)std::max(1, ev_.batch_limit);
    const bool lanes = row_lanes_ && c.n_lanes > 0 && in.size() > B;
    if (lanes) c.fork_lanes();
    int k = 0;
    for (size_t lo = 0; lo < in.size(); lo += B, ++k) {
        const size_t hi = std::min(in.size(), lo + B);
        CtVec r(in.begin() + lo, in.begin() + hi);
        if (lanes) {
            Context::LaneScope scope(c, 1 + k % c.n_lanes);
            r = tree_steps(r, n, padding * step_sign);
        } else {
            r = tree_steps(r, n, padding * step_sign);
        }
        std::copy(r.begin(), r.end(), out.begin() + lo);
    }
    if (lanes) c.join_lanes();
    return out;
}

CtVec Composite::rotsum_batch(const CtVec& in, int slots, int padding) { return tree_batch(in, slots, +1, padding); }
CtVec Composite::repeat_batch(const CtVec& in, int slots, int padding) { return tree_batch(in, slots, -1, padding); }

CtPtr Composite::add_many(const CtVec& v) {
    if (v.empty()) throw Error(FHELIN_ERR_ARG, "add_many: empty vector");
    CtVec cur = v;
    while (cur.size() > 1) {  // binary tree like EvalAddMany
        CtVec nxt;
        CtVec lhs, rhs;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) {
            lhs.push_back(cur[i]);
            rhs.push_back(cur[i + 1]);
        }
        nxt = ev_.add_batch(lhs, rhs);  // one launch per 32 pairs
        if (cur.size() & 1) nxt.push_back(cur.back());
        cur.swap(nxt);
    }
    return cur[0];
}

CtVec Composite::matmul_pt(const CtVec& rows_in, const PtPtr& w, const PtPtr& bias, int slots, int padding) {
    // rows that are the SAME ciphertext object (handles made by Clone(): the CLS-only driver feeds 129 clones of one
    // encryption of zero through W_O, src/main.cpp:220-235) give the same result: evaluate each distinct row once
    CtVec rows;
    std::vector<size_t> pos(rows_in.size());
    {
        std::map<const Ciphertext*, size_t> seen;
        for (size_t i = 0; i < rows_in.size(); ++i) {
            auto it = seen.find(rows_in[i].get());
            if (it == seen.end()) {
                it = seen.emplace(rows_in[i].get(), rows.size()).first;
                rows.push_back(rows_in[i]);
            }
            pos[i] = it->second;
        }
    }
    // rows are independent (reference loop :872,:888,:985): the same operation sequence, interchanged so that
    // every rotate-and-sum step runs over all rows at once
    CtVec out = rotsum_batch(ev_.mult_plain_batch(rows, w), slots, padding);
    if (bias) out = ev_.add_plain_batch(out, bias);
    if (rows.size() == rows_in.size()) return out;
    CtVec full(rows_in.size());
    for (size_t i = 0; i < rows_in.size(); ++i) full[i] = out[pos[i]];
    return full;
}

CtVec Composite::matmul_ct(const CtVec& rows, const CtPtr& w, int slots, int padding) {
    return matmul_ct_each(rows, CtVec(rows.size(), w), slots, padding);
}

// row i against its own ciphertext weight (the same ciphertext-weight matmul on several samples: each sample's rows meet that
// sample's weight): the products of all rows through ONE batched relinearisation, then the trees of all rows together
CtVec Composite::matmul_ct_each(const CtVec& rows, const CtVec& ws, int slots, int padding) {
    if (rows.size() != ws.size()) throw Error(FHELIN_ERR_ARG, "matmul_ct: one weight per row");
    if (rows.empty()) return {};
    CtVec prod = rows.size() == 1 ? CtVec{ev_.mult(rows[0], ws[0])} : ev_.mult_batch(rows, ws);
    return rotsum_batch(prod, slots, padding);
}

static bool same_plain_set(const std::vector<std::vector<double>>& kept, const std::vector<PtPtr>& now) {
    if (kept.size() != now.size()) return false;
    for (size_t i = 0; i < now.size(); ++i)
        if (kept[i] != now[i]->values) return false;
    return true;
}
static uint64_t hash_plain(const PtPtr& w, uint64_t h) {   // FNV-1a over the slot values and the level
    for (double v : w->values) {
        uint64_t bits;
        std::memcpy(&bits, &v, sizeof bits);
        h = (h ^ bits) * 1099511628211ull;
    }
    return (h ^ (uint64_t)w->values.size() ^ ((uint64_t)w->level << 32) ^ ((uint64_t)w->slots << 48)) * 1099511628211ull;
}

// W''_t (t = 0..3): block b (128 slots) of W''_t is block b of W_j with j = (b - t) mod 4 — see matmulRElarge.
// rotated: V_t = rot(W''_t, 128 t) instead (slot s of V_t = slot s + 128 t of W''_t): rot(x * W''_t, 128 t) = rot(x, 128 t) * V_t
std::vector<PtPtr> Composite::relarge_weights(const std::vector<PtPtr>& weights, bool rotated) {
    // keyed by CONTENT: a driver that reads its weights again for every sample (src/main.cpp does, per run) gets the same re-arranged
    // plaintexts back, and with them the rotation keys they are folded into (Evaluator::folded_key is keyed by plaintext handle)
    uint64_t h = 1469598103934665603ull;
    for (const PtPtr& w : weights) h = hash_plain(w, h);
    char key[96];
    snprintf(key, sizeof key, "relarge:%016llx:%d", (unsigned long long)h, rotated ? 1 : 0);
    auto it = relarge_cache_.find(key);
    if (it != relarge_cache_.end() && same_plain_set(relarge_src_[key], weights)) return it->second;   // a hash hit is CHECKED against the values
    const int ns = num_slots();
    std::vector<PtPtr> out;
    for (int t = 0; t < 4; ++t) {
        std::vector<double> v(ns, 0.0);
        for (int s = 0; s < ns; ++s) {
            const int src_slot = rotated ? (s + 128 * t) % ns : s;
            const int b = src_slot / 128, j = ((b - t) % 4 + 4) % 4;
            const auto& src = weights[j]->values;
            v[s] = src_slot < (int)src.size() ? src[src_slot] : 0.0;
        }
        out.push_back(encode_vec(v, weights[0]->level));
    }
    if (relarge_cache_.size() > 16) {   // bounded: a driver uses one or two weight sets
        relarge_cache_.clear();
        relarge_src_.clear();
    }
    relarge_cache_[key] = out;
    std::vector<std::vector<double>> src;
    for (const PtPtr& w : weights) src.push_back(w->values);
    relarge_src_[key] = src;
    return out;
}

bool Composite::relarge_shared(const CtVec& inputs, const std::vector<PtPtr>& weights) {
    return merge_rot_ && weights.size() == 4 && num_slots() == 16384 && !inputs.empty() &&
           ev_.have_rotation_keys({128, 256, 384, 512, 1024, 2048, 4096, 8192}, inputs[0]->slots);
}

CtVec Composite::relarge_u(const CtVec& inputs, const std::vector<PtPtr>& weights) {
    // rescale degree-2 inputs once (each product below would otherwise do it again)
    CtVec x = inputs;
    {
        CtVec need;
        std::vector<size_t> pos;
        for (size_t i = 0; i < x.size(); ++i)
            if (x[i]->deg >= 2) {
                need.push_back(x[i]);
                pos.push_back(i);
            }
        if (!need.empty()) {
            CtVec r = ev_.rescale_batch(need);
            for (size_t k = 0; k < pos.size(); ++k) x[pos[k]] = r[k];
        }
    }
    if (ev_.double_hoist) {
        // U = x * V_0 + sum_{t=1..3} rot(x, 128 t) * V_t with V_t = rot(W''_t, 128 t): the three rotations share ONE ModUp of the row
        // and the plaintext products are taken in the extended basis (rotation keys with V_t folded in: Evaluator::hoisted_dot_rows),
        // so a row costs one ModUp and one ModDown where the products-then-rotations form costs three ModUps and one ModDown.
        // U is wanted rescaled (the tree below runs on degree-1 rows): ModDown and rescale are one basis conversion
        return ev_.hoisted_dot_rows(x, relarge_weights(weights, true), {128, 256, 384}, ev_.merged_rescale);
    }
    const std::vector<PtPtr> w2 = relarge_weights(weights, false);
    // the four products stay unrescaled through the key switch that sums them: ONE rescale of U (inside rotsum_batch) instead
    // of four, at the price of running that one key switch a limb higher
    // the three products of a row that get rotated are produced next to one another (one block, [row][t]): the key switch
    // takes them where they are
    CtVec xflat;
    std::vector<PtPtr> wflat;
    for (size_t i = 0; i < x.size(); ++i)
        for (int t = 1; t < 4; ++t) {
            xflat.push_back(x[i]);
            wflat.push_back(w2[t]);
        }
    const CtVec rotated = ev_.mult_plain_each(xflat, wflat);
    const CtVec y0 = ev_.mult_plain_batch(x, w2[0]);
    std::vector<CtVec> rows(x.size(), CtVec(4));
    for (size_t i = 0; i < x.size(); ++i) {
        rows[i][0] = y0[i];
        for (int t = 1; t < 4; ++t) rows[i][t] = rotated[3 * i + (t - 1)];
    }
    return ev_.rotate_each_sum_rows(rows, {0, 128, 256, 384});
}

CtVec Composite::relarge_tail(const CtVec& u, const PtPtr& bias, double mask_val) {
    CtVec z = rotsum_batch(u, 32, 512);
    CtVec res = ev_.mult_plain_batch(z, block_mask(0, 512, mask_val));
    if (bias) res = ev_.add_plain_batch(res, bias);
    return res;
}

// One container of generate_containers(matmulRElarge(.)) from the U rows of its q <= 32 tokens:
//   C = sum_{i<q} rot(Z_i * mask_[0,512) + bias, -512 i),   Z_i = sum_{k<32} rot(U_i, 512 k).
// Z_i has period 512, so rot(Z_i * mask_[0,512), -512 i) = Z_i * mask_i with mask_i = mask_val on [512 i, 512 i + 512), and
// mask_i * rot(U_i, 512 k) = rot(mask_{(i + k) mod 32} * U_i, 512 k):
//   C = sum_{k<32} rot(W_k, 512 k) + sum_{i<q} rot(bias, -512 i),      W_k = sum_{i<q} mask_{(i + k) mod 32} * U_i.
// 32 plaintext-weighted sums over the q rows (the mask product the tree form has too: same depth) and ONE shift sum of 32 terms -
// 31 ModUps and 5 ModDowns for the group, where the tree form runs 2 key switches per row and then the container's own shift sum.
CtVec Composite::relarge_w(const CtVec& uin, double mask_val) {
    const int q = (int)uin.size();
    CtVec u = uin;
    {
        bool any2 = false;
        for (const CtPtr& c : u) any2 = any2 || c->deg >= 2;
        if (any2) u = ev_.rescale_batch(u);
    }
    std::vector<PtPtr> mask(32);
    for (int j = 0; j < 32; ++j) mask[j] = block_mask(512 * j, 512 * (j + 1), mask_val);
    const CtPtr& f = u[0];
    CtVec w = ev_.new_ct_batch(32, 2, f->ell, f->deg + 1, f->scale, f->slots);
    // all 32 sums in one pass over the rows (every operand read once), landing in one block in order
    if (!ev_.dot_plain_cyclic(u, mask, w)) {
        for (int k = 0; k < 32; ++k) {
            std::vector<PtPtr> pk(q);
            for (int b = 0; b < q; ++b) pk[b] = mask[(b + k) % 32];
            w[k] = ev_.dot_plain(u, pk);
        }
    }
    return w;
}

// sum_{i<q} rot(bias, -512 i) as one plaintext (cached per (bias, q))
PtPtr Composite::relarge_tiled_bias(const PtPtr& bias, int q) {
    const int ns = num_slots();
    char key[96];
    snprintf(key, sizeof key, "relarge_bias:%016llx:%d", (unsigned long long)hash_plain(bias, 1469598103934665603ull), q);
    auto it = relarge_cache_.find(key);
    if (it != relarge_cache_.end() && same_plain_set(relarge_src_[key], std::vector<PtPtr>{bias})) return it->second[0];
    std::vector<double> v(ns, 0.0);
    for (int i = 0; i < q; ++i)
        for (int s = 0; s < ns; ++s) {
            const double b = s < (int)bias->values.size() ? bias->values[s] : 0.0;
            if (b != 0.0) v[(s + 512 * i) % ns] += b;       // rot(bias, -512 i): slot s lands at s + 512 i
        }
    PtPtr tiled = encode_vec(v, bias->level);
    if (relarge_cache_.size() > 16) {
        relarge_cache_.clear();
        relarge_src_.clear();
    }
    relarge_cache_[key] = {tiled};
    relarge_src_[key] = {bias->values};
    return tiled;
}

CtPtr Composite::relarge_container(const CtVec& uin, const PtPtr& bias, double mask_val) {
    CtPtr c = shift_sum(relarge_w(uin, mask_val), 512);
    return bias ? ev_.add_plain(c, relarge_tiled_bias(bias, (int)uin.size())) : c;
}

CtVec Composite::relarge_containers(const CtVec& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias, double mask_val, const PtPtr& cbias) {
    return relarge_containers_multi(std::vector<CtVec>{inputs}, weights, bias, mask_val, cbias)[0];
}

// the fused form for several samples (one input count): the first step U of ALL rows in one batched call, the 32 cyclic sums per
// (sample, group), then ONE shift sum over the groups of one size of all samples.  Every container holds the residues the
// single-sample call gives (rows / groups of a batched key switch are independent).
std::vector<CtVec> Composite::relarge_containers_multi(const std::vector<CtVec>& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias,
                                                       double mask_val, const PtPtr& cbias) {
    const size_t X = inputs.size();
    if (!X) return {};
    CtVec flat_in;
    for (const CtVec& in : inputs) {
        if (in.size() != inputs[0].size()) throw Error(FHELIN_ERR_ARG, "generate_containers: samples must have one input count");
        flat_in.insert(flat_in.end(), in.begin(), in.end());
    }
    if (!fuse_relarge || !relarge_shared(flat_in, weights)) {
        std::vector<CtVec> rows(X);
        const CtVec r = matmulRElarge(flat_in, weights, bias, mask_val);
        const size_t n = inputs[0].size();
        for (size_t x = 0; x < X; ++x) rows[x].assign(r.begin() + x * n, r.begin() + (x + 1) * n);
        return generate_containers_multi(rows, cbias);
    }
    const CtVec u = relarge_u(flat_in, weights);
    const int total = (int)inputs[0].size();
    const int n_cont = (total + 31) / 32;
    std::vector<CtVec> containers(X, CtVec(n_cont));
    // the groups of generate_containers (:1164-1191): 32 rows each, a shorter last one
    std::map<int, std::vector<std::pair<size_t, int>>> fused, tree;   // rows in the group -> (sample, container)
    std::map<int, std::vector<CtVec>> fused_w, tree_rows;
    for (size_t x = 0; x < X; ++x)
        for (int i = 0, lo = 0; lo < total; lo += 32, ++i) {
            const int q = std::min(32, total - lo);
            CtVec ug(u.begin() + x * total + lo, u.begin() + x * total + lo + q);
            bool uniform = q >= RELARGE_FUSE_MIN;
            for (const CtPtr& c : ug)
                uniform = uniform && c->npoly == 2 && c->ell == ug[0]->ell && c->deg == ug[0]->deg && fabsl(c->scale / ug[0]->scale - 1.0L) < 1e-9L;
            if (uniform) {
                fused[q].push_back({x, i});
                fused_w[q].push_back(relarge_w(ug, mask_val));
            } else {
                tree[q].push_back({x, i});
                tree_rows[q].push_back(ug);
            }
        }
    for (const auto& e : fused) {
        CtVec c = shift_sum_multi(fused_w[e.first], 512);
        if (bias) c = ev_.add_plain_batch(c, relarge_tiled_bias(bias, e.first));
        for (size_t k = 0; k < e.second.size(); ++k) containers[e.second[k].first][e.second[k].second] = c[k];
    }
    for (const auto& e : tree) {
        // short groups (the driver's last tokens): the 5-step trees of all their rows together, then the container sums together
        const int q = e.first;
        CtVec flat;
        for (const CtVec& ug : tree_rows[q]) flat.insert(flat.end(), ug.begin(), ug.end());
        const CtVec rows = relarge_tail(flat, bias, mask_val);
        std::vector<CtVec> groups;
        for (size_t k = 0; k < e.second.size(); ++k) {
            CtVec g(rows.begin() + k * q, rows.begin() + (k + 1) * q);
            std::reverse(g.begin(), g.end());
            groups.push_back(g);
        }
        const CtVec c = wrap_containers_multi(groups, q);
        for (size_t k = 0; k < e.second.size(); ++k) containers[e.second[k].first][e.second[k].second] = c[k];
    }
    if (cbias) {
        CtVec flat;
        for (const CtVec& c : containers) flat.insert(flat.end(), c.begin(), c.end());
        flat = ev_.add_plain_batch(flat, cbias);
        for (size_t x = 0; x < X; ++x) containers[x].assign(flat.begin() + x * n_cont, flat.begin() + (x + 1) * n_cont);
    }
    return containers;
}

CtVec Composite::matmulRElarge(const CtVec& inputs, const std::vector<PtPtr>& weights, const PtPtr& bias, double mask_val) {
    // per input the reference computes (:915-944) res = sum_j shift_j(mask_first_128(rotsum(x * W_j, 128, 128))) + bias: four
    // 7-step rotate-and-sum trees per input.  What it needs of tree j is ONE block: block j of out = the sum of all 128 blocks
    // of P_j = x * W_j.  Re-associated (same slot values):
    //   U = sum_{t<4} rot(x * W''_t, 128 t),  W''_t = block-wise re-arrangement of the four weights (relarge_weights):
    //       block b of U holds sum_{t<4} P_j[block b + t] for j = b mod 4  (the tree steps 128, 256 of all four trees, with the
    //       "keep blocks = j mod 4" masks folded into the plaintext weights: no extra level);
    //   Z = U + its rotations by 512, 1024, ..., 8192 (5 doubling steps): block j of Z = sum of all blocks of P_j;
    //   out = Z * (mask_val on slots [0, 512)) + bias.
    // One key switch over three rotations (relarge_u) and a 5-step tree instead of four 7-step trees per input.
    if (relarge_shared(inputs, weights)) return relarge_tail(relarge_u(inputs, weights), bias, mask_val);
    // the reference's formulation: one tree per weight block; a rotsum(., 128, 128) output repeats with period 128, so masking
    // block j directly selects what "mask the first block, then shift it" does (FHELIN_MERGE_ROT=0: the reference's shifts)
    CtVec res(inputs.size());
    const bool direct = merge_rot_;
    for (int j = (int)weights.size() - 1; j >= 0; --j) {
        CtVec outs = rotsum_batch(ev_.mult_plain_batch(inputs, weights[j]), 128, 128);
        CtVec masked = ev_.mult_plain_batch(outs, direct ? block_mask(128 * j, 128 * (j + 1), mask_val) : first_n_mask(128, mask_val));
        if (j == (int)weights.size() - 1) {
            res = masked;
        } else {
            if (!direct) {
                res = ev_.rotate_batch(res, -64);
                res = ev_.rotate_batch(res, -64);
            }
            res = ev_.add_batch(res, masked);
        }
    }
    if (bias) res = ev_.add_plain_batch(res, bias);
    return res;
}

CtVec Composite::matmulCRlarge(const std::vector<CtVec>& rows, const std::vector<PtPtr>& weights, const PtPtr& bias) {
    // per row (:1005-1014) sum_j r[j] * W_j: the 4 products of every row in one batched launch sequence, then the
    // same pairwise sums as EvalAddMany ((p0 + p1) + (p2 + p3)) over all rows at once
    CtVec flat;
    std::vector<PtPtr> wflat;
    for (const CtVec& r : rows) {
        if (r.size() < 4 || weights.size() < 4) throw Error(FHELIN_ERR_ARG, "matmulCRlarge: need 4 blocks");
        for (int j = 0; j < 4; ++j) {
            flat.push_back(r[j]);
            wflat.push_back(weights[j]);
        }
    }
    CtVec prod = ev_.mult_plain_each(flat, wflat);
    CtVec l0, r0, l1, r1;
    for (size_t i = 0; i + 3 < prod.size(); i += 4) {
        l0.push_back(prod[i]);
        r0.push_back(prod[i + 1]);
        l1.push_back(prod[i + 2]);
        r1.push_back(prod[i + 3]);
    }
    CtVec sums = ev_.add_batch(ev_.add_batch(l0, r0), ev_.add_batch(l1, r1));
    CtVec out = rotsum_batch(sums, 128, 1);
    if (bias) out = ev_.add_plain_batch(out, bias);
    return out;
}

// sum_i rot(terms[i], step * i).  The reference builds these sums as a Horner chain (rotate the running sum by `step`,
// add the next term: n-1 DEPENDENT key switches).  Default form (round 3): radix-8 levels with a SHARED ModDown -
//   i = 8 m + k:  sum_i rot(t_i, step i) = sum_m rot( sum_{k<8} rot(t_{8m+k}, step k), 8 step m )
// every inner sum of eight terms is ONE key switch of seven rotated terms accumulated in the extended basis plus the unrotated one
// (Evaluator::rotate_each_sum_rows: a ModUp per term, one ModDown per sum), all inner sums of a level batched; the next level does
// the same with the unit 8 step.  128 terms: 127 ModUps as before but 16 + 2 + 1 = 19 ModDowns instead of 127, and every term passes
// through 3 key switches instead of 7.  Needs the keys of unit * {1..7} per level (the client's key set has them for the units
// 1, 8, 512; the last level of <= 128 terms needs 64 only); otherwise - or with FHELIN_MERGE_ROT=0 - the binary tree of round 1
// (ceil(log2 n) dependent levels, one batched key switch with one key per level).
CtPtr Composite::shift_sum(const CtVec& terms, int step) {
    if (terms.empty()) throw Error(FHELIN_ERR_ARG, "shift_sum: empty vector");
    return shift_sum_multi(std::vector<CtVec>{terms}, step)[0];
}

// the same sum for several independent groups of ONE size (the same call of a driver on several samples): every level's key
// switches of all groups share their launches.  A group's result holds the residues shift_sum gives for that group alone
// (rows of a batched key switch are independent).
CtVec Composite::shift_sum_multi(const std::vector<CtVec>& groups, int step) {
    if (groups.empty()) return {};
    const size_t G = groups.size();
    for (const CtVec& g : groups)
        if (g.empty() || g.size() != groups[0].size()) throw Error(FHELIN_ERR_ARG, "shift_sum: groups must be non-empty and of one size");
    std::vector<CtVec> cur = groups;
    int unit = step;
    while (cur[0].size() > 1 && merge_rot_) {
        const size_t n = cur[0].size();
        const int width = (int)std::min<size_t>(8, n);
        std::vector<int> need;
        for (int k = 1; k < width; ++k) need.push_back(unit * k);
        if (!ev_.have_rotation_keys(need, cur[0][0]->slots)) break;
        // full groups of eight go through one batched call, a shorter last group through its own
        const size_t full = n / 8;
        std::vector<CtVec> nxt(G);
        if (full) {
            std::vector<CtVec> rows;
            for (size_t g = 0; g < G; ++g)
                for (size_t m = 0; m < full; ++m) rows.emplace_back(cur[g].begin() + 8 * m, cur[g].begin() + 8 * m + 8);
            std::vector<int> idx;
            for (int k = 0; k < 8; ++k) idx.push_back(unit * k);
            const CtVec r = ev_.rotate_each_sum_rows(rows, idx);
            for (size_t g = 0; g < G; ++g) nxt[g].assign(r.begin() + g * full, r.begin() + (g + 1) * full);
        }
        if (n % 8) {
            const size_t tn = n - 8 * full;
            if (tn == 1) {
                for (size_t g = 0; g < G; ++g) nxt[g].push_back(cur[g].back());
            } else {
                std::vector<CtVec> tails;
                for (size_t g = 0; g < G; ++g) tails.emplace_back(cur[g].begin() + 8 * full, cur[g].end());
                std::vector<int> idx;
                for (size_t k = 0; k < tn; ++k) idx.push_back(unit * (int)k);
                const CtVec r = ev_.rotate_each_sum_rows(tails, idx);
                for (size_t g = 0; g < G; ++g) nxt[g].push_back(r[g]);
            }
        }
        cur.swap(nxt);
        unit *= 8;
    }
    // binary tree over what is left (all of it when the radix-8 keys are missing), rotations by unit * 2^level
    for (int level = 0; cur[0].size() > 1; ++level) {
        const size_t n = cur[0].size();
        CtVec odd, even;
        for (size_t g = 0; g < G; ++g) {
            for (size_t j = 1; j < n; j += 2) odd.push_back(cur[g][j]);
            for (size_t j = 0; j + 1 < n; j += 2) even.push_back(cur[g][j]);
        }
        const CtVec sum = ev_.add_batch(even, ev_.rotate_batch(odd, unit * (1 << level)));
        const size_t half = n / 2;
        std::vector<CtVec> nxt(G);
        for (size_t g = 0; g < G; ++g) {
            nxt[g].assign(sum.begin() + g * half, sum.begin() + (g + 1) * half);
            if (n & 1) nxt[g].push_back(cur[g].back());
        }
        cur.swap(nxt);
    }
    CtVec out(G);
    for (size_t g = 0; g < G; ++g) out[g] = cur[g][0];
    return out;
}

// rot(c, step * i) for the listed i (all below n).  Default form (round 3): i = 64 a + 8 b + k, three levels of HOISTED rotations -
// rot(c, 64 step a) from c, rot(., 8 step b) from each of those, rot(., step k) from each of those: every level is one ModUp per
// input for all its indices (Evaluator::rotate_many / rotate_many_batch), 1 + 2 + 16 = 19 ModUps for 128 rows instead of 127, and
// every row passes through at most 3 key switches.  Row i is the same composition whichever rows are asked for.  Needs the keys of
// step {1..7}, 8 step {1..7}, 64 step a; otherwise the doubling fan of round 1 (rot(c, j) = rot(rot(c, j - 2^h), 2^h)).
CtVec Composite::shift_fan_rows(const CtPtr& c, int n, int step, const std::vector<int>& idx) {
    CtVec out(idx.size());
    if (idx.empty()) return out;
    for (int i : idx)
        if (i < 0 || i >= n) throw Error(FHELIN_ERR_ARG, "shift_fan: row out of range");
    const int top = n - 1;   // the form (hoisted levels or doubling chain) is a property of the call, not of the rows that are read
    std::vector<int> need;
    for (int k = 1; k < 8 && k <= top; ++k) need.push_back(step * k);
    for (int b = 1; b < 8 && 8 * b <= top; ++b) need.push_back(8 * step * b);
    for (int a = 1; 64 * a <= top; ++a) need.push_back(64 * step * a);
    const bool hoisted = merge_rot_ && (need.empty() || ev_.have_rotation_keys(need, c->slots));
    if (hoisted) return shift_fan_rows_multi(CtVec{c}, n, step, idx)[0];
    {
        std::map<int, CtPtr> fan;
        fan[0] = c;
        std::vector<char> want(std::max(n, 1), 0);
        for (int i : idx)
            for (int j = i; j > 0;) {
                want[j] = 1;
                int h = 0;
                while ((2 << h) <= j) ++h;
                j -= 1 << h;
            }
        for (int have = 1; have < n; have *= 2) {
            CtVec src;
            std::vector<int> dst;
            for (int j = have; j < std::min(2 * have, n); ++j)
                if (want[j]) {
                    src.push_back(fan.at(j - have));
                    dst.push_back(j);
                }
            if (src.empty()) continue;
            CtVec rot = ev_.rotate_batch(src, step * have);
            for (size_t k = 0; k < dst.size(); ++k) fan[dst[k]] = rot[k];
        }
        for (size_t k = 0; k < idx.size(); ++k) out[k] = fan.at(idx[k]);
        return out;
    }
}

// the hoisted form for several source ciphertexts of ONE shape and one row list (the same call on several samples): every level
// is one batched hoisted key switch over all sources.  out[x] holds the residues shift_fan_rows(cs[x], ...) gives.
std::vector<CtVec> Composite::shift_fan_rows_multi(const CtVec& cs, int n, int step, const std::vector<int>& idx) {
    std::vector<CtVec> out(cs.size(), CtVec(idx.size()));
    if (idx.empty() || cs.empty()) return out;
    for (int i : idx)
        if (i < 0 || i >= n) throw Error(FHELIN_ERR_ARG, "shift_fan: row out of range");
    {
        const int top = n - 1;
        std::vector<int> need;
        for (int k = 1; k < 8 && k <= top; ++k) need.push_back(step * k);
        for (int b = 1; b < 8 && 8 * b <= top; ++b) need.push_back(8 * step * b);
        for (int a = 1; 64 * a <= top; ++a) need.push_back(64 * step * a);
        if (!(merge_rot_ && (need.empty() || ev_.have_rotation_keys(need, cs[0]->slots)))) {
            for (size_t x = 0; x < cs.size(); ++x) out[x] = shift_fan_rows(cs[x], n, step, idx);
            return out;
        }
    }
    const size_t X = cs.size();
    // level 1: the 64-blocks that hold a wanted row
    std::vector<int> as;
    for (int i : idx)
        if (std::find(as.begin(), as.end(), i / 64) == as.end()) as.push_back(i / 64);
    std::sort(as.begin(), as.end());
    std::vector<int> a_idx;
    for (int a : as) a_idx.push_back(64 * step * a);
    const std::vector<CtVec> A = ev_.rotate_many_batch(cs, a_idx);      // [x][a]; index 0 -> the source itself
    // level 2: per 64-block the 8-blocks that hold a wanted row (the union over the blocks: one index list for the batch)
    std::vector<int> bs;
    for (int i : idx)
        if (std::find(bs.begin(), bs.end(), (i % 64) / 8) == bs.end()) bs.push_back((i % 64) / 8);
    std::sort(bs.begin(), bs.end());
    std::vector<int> b_idx;
    for (int b : bs) b_idx.push_back(8 * step * b);
    CtVec Aflat;
    for (size_t x = 0; x < X; ++x) Aflat.insert(Aflat.end(), A[x].begin(), A[x].end());
    const std::vector<CtVec> Bm = ev_.rotate_many_batch(Aflat, b_idx);  // [x * |as| + a][b]
    // level 3: the (a, b) pairs that hold a wanted row, all their k at once
    std::vector<std::pair<int, int>> ab;
    for (int i : idx) {
        const std::pair<int, int> key{i / 64, (i % 64) / 8};
        if (std::find(ab.begin(), ab.end(), key) == ab.end()) ab.push_back(key);
    }
    std::vector<int> ks;
    for (int i : idx)
        if (std::find(ks.begin(), ks.end(), i % 8) == ks.end()) ks.push_back(i % 8);
    std::sort(ks.begin(), ks.end());
    std::vector<int> k_idx;
    for (int k : ks) k_idx.push_back(step * k);
    CtVec Bsel;
    for (size_t x = 0; x < X; ++x)
        for (const auto& e : ab) {
            const size_t ai = std::find(as.begin(), as.end(), e.first) - as.begin();
            const size_t bi = std::find(bs.begin(), bs.end(), e.second) - bs.begin();
            Bsel.push_back(Bm[x * as.size() + ai][bi]);
        }
    const std::vector<CtVec> Km = ev_.rotate_many_batch(Bsel, k_idx);   // [x * |ab| + (a, b)][k]
    for (size_t x = 0; x < X; ++x)
        for (size_t r = 0; r < idx.size(); ++r) {
            const int i = idx[r];
            const size_t abi = std::find(ab.begin(), ab.end(), std::pair<int, int>{i / 64, (i % 64) / 8}) - ab.begin();
            const size_t ki = std::find(ks.begin(), ks.end(), i % 8) - ks.begin();
            out[x][r] = Km[x * ab.size() + abi][ki];
        }
    return out;
}

CtVec Composite::shift_fan(const CtPtr& c, int n, int step) {
    std::vector<int> all(std::max(n, 0));
    for (int i = 0; i < n; ++i) all[i] = i;
    return shift_fan_rows(c, n, step, all);
}

CtPtr Composite::matmulScores(const CtVec& queries, const CtPtr& key) {
    return matmulScores_multi(std::vector<CtVec>{queries}, CtVec{key})[0];
}

// matmulScores for several samples at once (queries[x] against keys[x], one group size): one batched relinearisation, one set of
// trees and one shift sum per level for all of them
CtVec Composite::matmulScores_multi(const std::vector<CtVec>& queries, const CtVec& keys) {
    if (queries.empty() || queries.size() != keys.size()) throw Error(FHELIN_ERR_ARG, "matmulScores: one key per group of queries");
    CtVec rows, ws;
    for (size_t x = 0; x < queries.size(); ++x) {
        if (queries[x].empty() || queries[x].size() != queries[0].size()) throw Error(FHELIN_ERR_ARG, "matmulScores: no queries");
        for (const CtPtr& q : queries[x]) {
            rows.push_back(q);
            ws.push_back(keys[x]);
        }
    }
    const CtVec scores = matmul_ct_each(rows, ws, 128, 1);
    const double r = 1 / 8.0;  // "later corrected with e^(x/r)"  (:1031)
    // :1036-1044 rotate-by(-1)-and-add chain == sum_i rot(masked_i, -i)
    const CtVec masked = ev_.mult_plain_batch(scores, heads_128_mask(1 / 8.0 * r));
    const size_t nq = queries[0].size();
    std::vector<CtVec> groups;
    for (size_t x = 0; x < queries.size(); ++x) groups.emplace_back(masked.begin() + x * nq, masked.begin() + (x + 1) * nq);
    return shift_sum_multi(groups, -1);
}

CtPtr Composite::wrapUpRepeated(const CtVec& v) {
    std::vector<PtPtr> masks;
    for (size_t i = 0; i < v.size(); ++i) masks.push_back(block_mask(128 * (int)i, 128 * ((int)i + 1), 1));
    return ev_.dot_plain(v, masks);   // sum_i v_i * mask_i in one inner-product pass (EvalAddMany of the masked rows, :1060-1068)
}

CtPtr Composite::wrapUpExpanded(const CtVec& v) {
    if (v.empty()) throw Error(FHELIN_ERR_ARG, "wrapUpExpanded: empty vector");
    return wrapUpExpanded_multi(std::vector<CtVec>{v})[0];
}

CtVec Composite::wrapUpExpanded_multi(const std::vector<CtVec>& groups) {
    // :1072-1084 rotate-by(-1)-and-add chain == sum_i rot(mask(v_i), -i); several samples' groups (one size) share every launch
    CtVec flat;
    for (const CtVec& g : groups) {
        if (g.empty() || g.size() != groups[0].size()) throw Error(FHELIN_ERR_ARG, "wrapUpExpanded: groups must be non-empty and of one size");
        flat.insert(flat.end(), g.begin(), g.end());
    }
    const CtVec masked = ev_.mult_plain_batch(flat, mod_n_mask(128, 0));
    const size_t n = groups[0].size();
    std::vector<CtVec> mg;
    for (size_t x = 0; x < groups.size(); ++x) mg.emplace_back(masked.begin() + x * n, masked.begin() + (x + 1) * n);
    return shift_sum_multi(mg, -1);
}

CtVec Composite::unwrapExpanded(CtPtr c, int n) {
    // :1089-1097 masks rot(c, i) for i = 0..n-1 (there: n-1 dependent rotations by 1); the `repeat` of every extracted
    // token is independent
    std::vector<int> all(std::max(n, 0));
    for (int i = 0; i < n; ++i) all[i] = i;
    return unwrapExpanded_rows(c, n, all);
}

CtVec Composite::unwrapExpanded_rows(CtPtr c, int n, const std::vector<int>& idx) {
    return unwrapExpanded_rows_multi(CtVec{c}, n, idx)[0];
}

// the rows `idx` of unwrapExpanded(cs[x], n) for several sources of one shape (the same call on several samples): out[x]
std::vector<CtVec> Composite::unwrapExpanded_rows_multi(const CtVec& cs, int n, const std::vector<int>& idx) {
    if (cs.empty()) return {};
    if (bulk_unwrap && merge_rot_ && (int)idx.size() >= UNWRAP_BULK_MIN && n <= 128 && num_slots() % 128 == 0) {
        std::vector<int> need;
        for (int k = 1; k < 8; ++k) need.push_back(-k), need.push_back(-8 * k);
        need.push_back(-64);
        if (ev_.have_rotation_keys(need, cs[0]->slots)) return unwrapExpanded_bulk_multi(cs, n, idx);
    }
    // row i of the fan is the same composition whichever rows are asked for (shift_fan_rows)
    const std::vector<CtVec> fans = shift_fan_rows_multi(cs, n, 1, idx);
    CtVec flat;
    for (const CtVec& f : fans) flat.insert(flat.end(), f.begin(), f.end());
    const CtVec rep = repeat_batch(ev_.mult_plain_batch(flat, mod_n_mask(128, 0)), 128, 1);
    std::vector<CtVec> out;
    for (size_t x = 0; x < cs.size(); ++x) out.emplace_back(rep.begin() + x * idx.size(), rep.begin() + (x + 1) * idx.size());
    return out;
}

CtVec Composite::unwrapExpanded_bulk(CtPtr c, int n, const std::vector<int>& idx) { return unwrapExpanded_bulk_multi(CtVec{c}, n, idx)[0]; }

std::vector<CtVec> Composite::unwrapExpanded_bulk_multi(const CtVec& cs_in, int n, const std::vector<int>& idx) {
    for (int i : idx)
        if (i < 0 || i >= n) throw Error(FHELIN_ERR_ARG, "unwrapExpanded: row out of range");
    CtVec cs = cs_in;
    {
        CtVec need;
        std::vector<size_t> pos;
        for (size_t x = 0; x < cs.size(); ++x)
            if (cs[x]->deg >= 2) {
                need.push_back(cs[x]);
                pos.push_back(x);
            }
        if (!need.empty()) {
            const CtVec r = ev_.rescale_batch(need);   // a single one goes through rescale() itself
            for (size_t k = 0; k < pos.size(); ++k) cs[pos[k]] = r[k];
        }
    }
    const int imin = *std::min_element(idx.begin(), idx.end()), imax = *std::max_element(idx.begin(), idx.end());
    // D_j = rot(c, j) for j in [imin - 127, imax]: the fan of the call for j >= 0, the fan by -1 (128 rows) for j < 0
    std::vector<int> pj, nj;
    for (int j = 0; j <= imax; ++j) pj.push_back(j);
    for (int j = 1; j <= 127 - imin; ++j) nj.push_back(j);
    const std::vector<CtVec> pos_all = shift_fan_rows_multi(cs, n, 1, pj);
    const std::vector<CtVec> neg_all = nj.empty() ? std::vector<CtVec>(cs.size()) : shift_fan_rows_multi(cs, 128, -1, nj);
    std::vector<PtPtr> mask(128);
    for (int k = 0; k < 128; ++k) mask[k] = mod_n_mask(128, k);
    std::vector<CtVec> out_all;
    for (size_t x = 0; x < cs.size(); ++x) {
        const CtPtr& c = cs[x];
        const CtVec &pos = pos_all[x], &neg = neg_all[x];
        auto D = [&](int d) -> CtPtr {
            if (d >= 0) return d <= imax ? pos[d] : CtPtr();
            return -d <= 127 - imin ? neg[-d - 1] : CtPtr();
        };
        CtVec out(idx.size());
        std::vector<char> block_wanted(4, 0);
        for (int i : idx) block_wanted[i / 32] = 1;
        for (int g = 0; g < 4; ++g) {
            if (!block_wanted[g]) continue;
            // x_i = sum_{k<128} mask_k * D_{i-k} for the 32 rows i = 32 g + t: four tap chunks of 32, accumulated in place
            CtVec dest = ev_.new_ct_batch(32, 2, c->ell, c->deg + 1, c->scale, c->slots);
            for (int ch = 0; ch < 4; ++ch) {
                const int p = 32 * g - 32 * ch;
                CtVec cur(32), prev(32);
                for (int j = 0; j < 32; ++j) {
                    cur[j] = D(p + j);
                    prev[j] = p - 32 + j >= -127 ? D(p - 32 + j) : CtPtr();
                }
                const std::vector<PtPtr> m(mask.begin() + 32 * ch, mask.begin() + 32 * ch + 32);
                if (!ev_.dot_plain_window(cur, prev, m, dest, ch > 0)) throw Error(FHELIN_ERR_INTERNAL, "unwrapExpanded_bulk: window operands");
            }
            for (size_t r = 0; r < idx.size(); ++r)
                if (idx[r] / 32 == g) out[r] = dest[idx[r] % 32];
        }
        out_all.push_back(out);
    }
    return out_all;
}

CtVec Composite::unwrapScoresExpanded(CtPtr c, int n) {
    const CtVec fan = shift_fan(c, n, 1);
    CtVec a = repeat_batch(ev_.mult_plain_batch(fan, mod_n_mask(128, 0)), 64, 1);
    CtVec b = repeat_batch(ev_.mult_plain_batch(fan, mod_n_mask(128, 64)), 64, 1);
    return ev_.add_batch(a, b);
}

CtVec Composite::unwrap_512_in_4_128(const CtPtr& c, int index) {
    CtVec result;
    const int shift = index * 512;
    for (int k = 0; k < 4; ++k) {
        CtPtr s = mask_block(c, shift + 128 * k, shift + 128 * (k + 1), 1);
        result.push_back(repeat(s, 128, -128));
    }
    return result;
}

std::vector<CtVec> Composite::unwrapRepeatedLarge(const CtVec& containers, int input_number, int first, int count) {
    return unwrapRepeatedLarge_multi(std::vector<CtVec>{containers}, input_number, first, count)[0];
}

// unwrapRepeatedLarge for several samples (one container count, one token count): every stage's key switches of all samples share
// their launches.  out[x][token][k]; sample x's outputs hold the residues the single-sample call gives.
std::vector<std::vector<CtVec>> Composite::unwrapRepeatedLarge_multi(const std::vector<CtVec>& containers, int input_number, int first, int count) {
    if (count < 0) count = input_number - first;
    if (first < 0 || first + count > input_number) throw Error(FHELIN_ERR_ARG, "unwrapRepeatedLarge: token range out of bounds");
    const size_t X = containers.size();
    std::vector<std::vector<CtVec>> out(X);
    if (!X) return out;
    for (const CtVec& cv : containers)
        if (cv.size() != containers[0].size()) throw Error(FHELIN_ERR_ARG, "unwrapRepeatedLarge: samples must have one container count");
    const size_t nc = containers[0].size();
    std::vector<int> quantities;
    for (int i = 0; i < input_number / 32.0; ++i) {
        int q = 32;
        if ((i + 1) * 32 > input_number) q = input_number - i * 32;
        quantities.push_back(q);
    }
    const int ns = num_slots();
    const bool shared = merge_rot_ && ns == 16384 &&
                        ev_.have_rotation_keys({128, 256, 384, -128, -256, -384, 512, 1024, 1536, 2048, 2560, 3072, 3584, 4096, 8192, 12288},
                                               nc == 0 ? 0 : containers[0][0]->slots);
    if (!shared) {
        // unwrap_512_in_4_128 (:1142-1162) for every (container, token): mask the four 128-slot blocks, then one batched
        // repeat(128, -128) over all of them
        CtVec src;
        std::vector<PtPtr> masks;
        for (size_t x = 0; x < X; ++x)
            for (size_t i = 0; i < nc && i < quantities.size(); ++i)
                for (int j = 0; j < quantities[i]; ++j) {
                    const int token = (int)i * 32 + j;
                    if (token < first || token >= first + count) continue;
                    for (int k = 0; k < 4; ++k) {
                        src.push_back(containers[x][i]);
                        masks.push_back(block_mask(j * 512 + 128 * k, j * 512 + 128 * (k + 1), 1));
                    }
                }
        CtVec rep = repeat_batch(ev_.mult_plain_each(src, masks), 128, -128);
        const size_t per = rep.size() / X;
        for (size_t x = 0; x < X; ++x)
            for (size_t i = 0; i + 3 < per; i += 4) out[x].push_back(CtVec(rep.begin() + x * per + i, rep.begin() + x * per + i + 4));
        return out;
    }
    // The same slot values with four of the seven doubling steps SHARED between tokens.  The reference masks block k of
    // token j and replicates it 128 times: 7 steps per (token, block).  Here, per container, block k and range a (the 8 tokens
    // 8a..8a+7 = slots [4096a, 4096a+4096)):
    //   A   = container * (slot mod 512 in [128k, 128k+128) and slot in range a)                        one mask level
    //   B   = A + its rotations by 128(k - m), m != k: block k of each of the 8 tokens copied over that token's own 512 slots
    //         (ONE merged key switch of three rotations)
    //   D   = B replicated over the four ranges (rotations by 4096, 8192, 12288: one merged key switch): 4096-periodic
    //   out_{8a+b, k} = (D * (slot mod 4096 in [512b, 512b+512))) replicated by 512, 1024, ..., 3584      one mask level, ONE
    //         merged key switch of seven rotations per output (the 32 shifts 512 i = the 4 x 8 shifts 4096 a' + 512 i')
    // One key switch per output instead of three, 5 shared ones per (container, k, a); the same two mask levels as the
    // one-stage shared form this replaces (which needed two key switches per output).
    // the (sample, container, range) triples that hold a wanted token
    struct Grp { size_t x; int i, a; };
    std::vector<Grp> groups;
    for (size_t x = 0; x < X; ++x)
        for (size_t i = 0; i < nc && i < quantities.size(); ++i)
            for (int a = 0; a < 4; ++a) {
                bool wanted = false;
                for (int j = 8 * a; j < 8 * a + 8 && j < quantities[i]; ++j) {
                    const int token = (int)i * 32 + j;
                    wanted = wanted || (token >= first && token < first + count);
                }
                if (wanted) groups.push_back({x, (int)i, a});
            }
    // stage 1, batched over the groups: per block k one masked product, one rescale, one merged key switch; then ONE
    // replication over the four ranges for all of them
    CtVec Ball;
    for (int k = 0; k < 4; ++k) {
        CtVec from;
        std::vector<PtPtr> mk;
        for (const Grp& g : groups) {
            from.push_back(containers[g.x][g.i]);
            mk.push_back(mod_range_mask(512, 128 * k, 128 * (k + 1), 4096 * g.a, 4096 * (g.a + 1)));
        }
        CtVec A = ev_.rescale_batch(ev_.mult_plain_each(from, mk));
        std::vector<int> idx;
        for (int m = 0; m < 4; ++m)
            if (m != k) idx.push_back(128 * (k - m));
        CtVec B = ev_.rotate_sum_batch(A, idx);
        Ball.insert(Ball.end(), B.begin(), B.end());
    }
    const CtVec Dall = repeat_batch(Ball, 4, -4096);   // [k][group]
    // stage 2 (round 3 form): out_{8a+b, k} = sum_{m<8} rot(D * mask_b, 512 m) with mask_b = (slot mod 4096 in [512b, 512b+512)).  A rotation
    // commutes with the mask: rot(D * mask_b, 512 m) = rot(D, 512 m) * mask_{(b-m) mod 8}, so the eight rotations of D - ONE hoisted key
    // switch per (container, range, block): one ModUp, seven inner products - serve all eight tokens of the range, and every output is an
    // inner sum of eight ciphertext x plaintext products (Evaluator::dot_plain_groups: the rotated ciphertexts are read once for all
    // tokens).  Before: one merged key switch of seven rotations PER OUTPUT (520 ModUps + 520 ModDowns + 3640 gathered inner products
    // per pass); now 80 ModUps, 560 ModDowns, 560 inner products.  The outputs leave with noise degree 2 (rescaled by their consumer).
    std::vector<size_t> d_of;                       // indices into Dall that hold a wanted token, and per such input the wanted b's
    std::vector<std::vector<int>> bs_of;
    for (int k = 0; k < 4; ++k)
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const int i = groups[gi].i, a = groups[gi].a;
            std::vector<int> bs;
            for (int j = 8 * a; j < 8 * a + 8 && j < quantities[i]; ++j) {
                const int token = i * 32 + j;
                if (token >= first && token < first + count) bs.push_back(j - 8 * a);
            }
            if (bs.empty()) continue;
            d_of.push_back((size_t)k * groups.size() + gi);
            bs_of.push_back(bs);
        }
    CtVec dsel;
    for (size_t x : d_of) dsel.push_back(Dall[x]);
    std::vector<int> ridx;
    for (int m = 0; m < 8; ++m) ridx.push_back(512 * m);
    const std::vector<CtVec> rot = ev_.rotate_many_batch(dsel, ridx);      // [input][m], m = 0: the input itself
    // a sample's outputs in ONE block in the order their consumer reads them ([token][k]: matmulCRlarge rescales and multiplies them in that
    // order, so its batched rescale takes them as they stand - no gather copies)
    std::vector<CtVec> all_out(X);
    for (size_t x = 0; x < X && !rot.empty(); ++x) all_out[x] = ev_.new_ct_batch(count * 4, 2, rot[0][0]->ell, 2, 0, rot[0][0]->slots);
    std::vector<std::map<std::pair<int, int>, CtPtr>> made(X);             // per sample: (token, k) -> output
    for (size_t z = 0; z < d_of.size(); ++z) {
        const int k = (int)(d_of[z] / groups.size());
        const Grp& g = groups[d_of[z] % groups.size()];
        const int i = g.i, a = g.a;
        const std::vector<int>& bs = bs_of[z];
        std::vector<std::vector<PtPtr>> pts(bs.size(), std::vector<PtPtr>(8));
        for (size_t q = 0; q < bs.size(); ++q)
            for (int m = 0; m < 8; ++m) {
                const int bb = ((bs[q] - m) % 8 + 8) % 8;
                pts[q][m] = mod_range_mask(4096, 512 * bb, 512 * (bb + 1));
            }
        CtVec dest;
        for (int b : bs) dest.push_back(all_out[g.x][(size_t)(i * 32 + 8 * a + b - first) * 4 + k]);
        if (!ev_.dot_plain_groups(rot[z], pts, 0, dest))
            for (size_t q = 0; q < bs.size(); ++q) dest[q] = ev_.dot_plain(rot[z], pts[q], 0, dest[q]);
        for (size_t q = 0; q < bs.size(); ++q) made[g.x][{i * 32 + 8 * a + bs[q], k}] = dest[q];
    }
    for (size_t x = 0; x < X; ++x)
        for (int token = first; token < first + count; ++token) {
            CtVec four;
            for (int k = 0; k < 4; ++k) four.push_back(made[x].at({token, k}));
            out[x].push_back(four);
        }
    return out;
}

CtPtr Composite::wrap_containers(const CtVec& c, int n) { return wrap_containers_multi(std::vector<CtVec>{c}, n)[0]; }

CtVec Composite::wrap_containers_multi(const std::vector<CtVec>& cs, int n) {
    // :1186-1193 result = rot(result, -512) + c[i]  ==  sum_i rot(c[n-1-i], -512 i)
    std::vector<CtVec> groups;
    for (const CtVec& c : cs) {
        if (c.empty() || n > (int)c.size()) throw Error(FHELIN_ERR_ARG, "wrap_containers: bad input count");
        CtVec terms(c.begin(), c.begin() + n);
        std::reverse(terms.begin(), terms.end());
        groups.push_back(terms);
    }
    return shift_sum_multi(groups, -512);
}

CtVec Composite::generate_containers(const CtVec& inputs, const PtPtr& bias) {
    return generate_containers_multi(std::vector<CtVec>{inputs}, bias)[0];
}

// generate_containers for several samples (one input count): container i of every sample is built in the same launches
std::vector<CtVec> Composite::generate_containers_multi(const std::vector<CtVec>& inputs, const PtPtr& bias) {
    const size_t X = inputs.size();
    std::vector<CtVec> containers(X);
    if (!X) return containers;
    const int total = (int)inputs[0].size();
    for (const CtVec& in : inputs)
        if ((int)in.size() != total) throw Error(FHELIN_ERR_ARG, "generate_containers: samples must have one input count");
    // all containers of one quantity (the full ones of every sample; the shorter last ones) go through ONE shift sum: their
    // key switches share launches.  A container's residues do not depend on which others it shares a launch with.
    std::map<int, std::vector<std::pair<size_t, int>>> by_q;   // quantity -> (sample, container)
    int n_cont = 0;
    for (int i = 0; i < total / 32.0; ++i, ++n_cont) {
        int quantity = 32;
        if ((i + 1) * 32 > total) quantity = total - i * 32;
        for (size_t x = 0; x < X; ++x) by_q[quantity].push_back({x, i});
    }
    for (size_t x = 0; x < X; ++x) containers[x].resize(n_cont);
    for (const auto& e : by_q) {
        std::vector<CtVec> sliced;
        for (const auto& xi : e.second) {
            // slicing(inputs, 32 i, 32 (i+1))  (:1338-1357): returns the whole vector when it has <= 32 entries
            const CtVec& in = inputs[xi.first];
            const int lo = xi.second * 32, hi = (xi.second + 1) * 32;
            CtVec sl = hi - lo >= total ? in : CtVec(in.begin() + lo, in.begin() + std::min(hi, total));
            std::reverse(sl.begin(), sl.end());
            sliced.push_back(sl);
        }
        const CtVec part = wrap_containers_multi(sliced, e.first);
        for (size_t k = 0; k < e.second.size(); ++k) containers[e.second[k].first][e.second[k].second] = part[k];
    }
    if (bias) {
        CtVec flat;
        for (const CtVec& c : containers) flat.insert(flat.end(), c.begin(), c.end());
        flat = ev_.add_plain_batch(flat, bias);
        const size_t per = containers[0].size();
        for (size_t x = 0; x < X; ++x) containers[x].assign(flat.begin() + x * per, flat.begin() + (x + 1) * per);
    }
    return containers;
}

}  // namespace fhelin
