// extern "C" boundary, part 2: keys, plaintexts, ciphertexts, leveled evaluation.
#include "../../include/fhelin.h"
#include <hip/hip_runtime.h>
#include <cstring>
#include "capi_internal.h"

using namespace fhelin;

#define NEED(x) if (!(x)) return capi_fail(FHELIN_ERR_ARG, "null argument")


static KeyPtr& key_slot(fhelin_ctx* c, int kind, int index) {
    if (kind == 0) return c->ev.relin_key;
    if (kind == 1) return c->ev.rot_keys[c->ctx.galois_element(index)];
    if (kind == 2) return c->ev.conj_key;
    throw Error(FHELIN_ERR_ARG, "unknown key kind");
}

extern "C" {

int fhelin_keygen(fhelin_ctx* c) {
    NEED(c);
    FHELIN_TRY
    c->cl.keygen();
    FHELIN_CATCH
}
int fhelin_gen_relin_key(fhelin_ctx* c) {
    NEED(c);
    FHELIN_TRY
    c->cl.gen_relin_key();
    FHELIN_CATCH
}
int fhelin_gen_rotation_keys(fhelin_ctx* c, const int32_t* idx, int32_t n) {
    NEED(c && (idx || n == 0));
    FHELIN_TRY
    for (int i = 0; i < n; ++i) c->cl.gen_rotation_key(idx[i]);
    FHELIN_CATCH
}
int fhelin_gen_conj_key(fhelin_ctx* c) {
    NEED(c);
    FHELIN_TRY
    c->cl.gen_conj_key();
    FHELIN_CATCH
}
int fhelin_secret_export(fhelin_ctx* c, uint64_t* out, size_t cap) {
    NEED(c && out);
    FHELIN_TRY
    if (cap < (size_t)(c->ctx.L + 1 + c->ctx.K) * c->ctx.N) throw Error(FHELIN_ERR_ARG, "buffer too small");
    c->cl.export_secret(out);
    FHELIN_CATCH
}
int fhelin_secret_import(fhelin_ctx* c, const uint64_t* in, size_t words) {
    NEED(c && in);
    FHELIN_TRY
    if (words != (size_t)(c->ctx.L + 1 + c->ctx.K) * c->ctx.N) throw Error(FHELIN_ERR_ARG, "secret must be [L+1+k][N]");
    c->cl.import_secret(in);
    FHELIN_CATCH
}
int fhelin_key_export(fhelin_ctx* c, int32_t kind, int32_t index, uint64_t* out, size_t cap) {
    NEED(c && out);
    FHELIN_TRY
    c->ctx.require_device();
    KeyPtr k = key_slot(c, kind, index);
    if (!k) throw Error(FHELIN_ERR_KEY, "key not present");
    if (cap < k->words()) throw Error(FHELIN_ERR_ARG, "buffer too small");
    hip_check(hipMemcpyAsync(out, k->d, k->words() * 8, hipMemcpyDeviceToHost, c->ctx.stream), "key export");
    c->ctx.sync();
    FHELIN_CATCH
}
int fhelin_key_import(fhelin_ctx* c, int32_t kind, int32_t index, const uint64_t* in, size_t words) {
    NEED(c && in);
    FHELIN_TRY
    c->ctx.require_device();
    KeyPtr k = c->ev.new_key();
    if (words != k->words()) throw Error(FHELIN_ERR_ARG, "key must be [dnum][2][L+1+k][N]");
    hip_check(hipMemcpyAsync(k->d, in, words * 8, hipMemcpyHostToDevice, c->ctx.stream), "key import");
    c->ctx.sync();
    key_slot(c, kind, index) = k;
    FHELIN_CATCH
}

int fhelin_encode(fhelin_ctx* c, const double* vals, int32_t n, int32_t level, int32_t slots, fhelin_pt** out) {
    NEED(c && out && (vals || n == 0));
    FHELIN_TRY
    auto* h = new fhelin_pt;
    h->p = c->cl.encode(vals, n, level, slots);
    *out = h;
    FHELIN_CATCH
}
void fhelin_pt_free(fhelin_pt* p) { delete p; }
int fhelin_pt_export(fhelin_ctx* c, const fhelin_pt* p, int32_t ell, double scale_hi, double scale_lo, uint64_t* out, size_t cap) {
    NEED(c && p && out);
    FHELIN_TRY
    Context& x = c->ctx;
    x.require_device();
    const bool full = ell == x.L + 1 + x.K;      // the encoding over the full key basis (Q limbs, then the special limbs)
    if (ell < 1 || (ell > x.L + 1 && !full)) throw Error(FHELIN_ERR_ARG, "pt_export: bad limb count");
    if (cap < (size_t)ell * x.N) throw Error(FHELIN_ERR_ARG, "buffer too small");
    if (full && !(scale_hi > 0)) throw Error(FHELIN_ERR_ARG, "pt_export: the full-basis encoding needs an explicit scale");
    auto enc = p->p->at(ell, scale_hi > 0 ? (long double)scale_hi + (long double)scale_lo : x.sf_real[x.L + 1 - ell]);
    hip_check(hipMemcpyAsync(out, enc->d, (size_t)ell * x.N * 8, hipMemcpyDeviceToHost, x.stream), "pt export");
    x.sync();
    FHELIN_CATCH
}

int fhelin_encrypt(fhelin_ctx* c, const fhelin_pt* p, fhelin_ct** out) {
    NEED(c && p && out);
    FHELIN_TRY
    *out = wrap(c, c->cl.encrypt(p->p, c->plan.next_drop(c->ctx.L + 1 - p->p->level)));
    if (c->plan.live((*out)->node, (*out)->node_epoch)) c->plan.nodes[(*out)->node].ordinal = c->plan.next_ordinal - 1;
    FHELIN_CATCH
}
int fhelin_encrypt_batch(fhelin_ctx* c, const double* vals, int32_t n_vec, int32_t n_per, int32_t level, int32_t slots, fhelin_ct** outs) {
    NEED(c && (vals || n_vec == 0) && outs);
    FHELIN_TRY
    // every vector is a source of the level plan of its own: vectors that the plan starts at the same level go through
    // the batched encryptor together
    if (n_vec < 0 || n_per < 0) throw Error(FHELIN_ERR_ARG, "encrypt_batch: negative count");
    if (level < 0 || level > c->ctx.L) throw Error(FHELIN_ERR_ARG, "encrypt_batch: level out of range");
    std::vector<int> drop(n_vec);
    for (int i = 0; i < n_vec; ++i)
        drop[i] = std::max(0, std::min(c->ctx.L - level, c->plan.next_drop(c->ctx.L + 1 - level)));
    const int first_ordinal = c->plan.next_ordinal - n_vec;
    std::vector<CtPtr> r(n_vec);
    std::vector<char> seen(n_vec, 0);
    for (int i = 0; i < n_vec; ++i) {
        if (seen[i]) continue;
        std::vector<int> pick;
        for (int j = i; j < n_vec; ++j)
            if (!seen[j] && drop[j] == drop[i]) {
                pick.push_back(j);
                seen[j] = 1;
            }
        if ((int)pick.size() == n_vec) {
            r = c->cl.encrypt_batch(vals, n_vec, n_per, level + drop[i], slots);
            break;
        }
        std::vector<double> sub((size_t)pick.size() * n_per);
        for (size_t k = 0; k < pick.size(); ++k)
            std::memcpy(sub.data() + k * n_per, vals + (size_t)pick[k] * n_per, (size_t)n_per * sizeof(double));
        std::vector<CtPtr> part = c->cl.encrypt_batch(sub.data(), (int)pick.size(), n_per, level + drop[i], slots);
        for (size_t k = 0; k < pick.size(); ++k) r[pick[k]] = part[k];
    }
    for (int i = 0; i < n_vec; ++i) {
        outs[i] = wrap(c, r[i]);
        if (c->plan.live(outs[i]->node, outs[i]->node_epoch)) c->plan.nodes[outs[i]->node].ordinal = first_ordinal + i;
    }
    FHELIN_CATCH
}
int fhelin_client_ingest(fhelin_ctx* c, const double* emb, const int32_t* tokens, const double* table, int32_t vocab, int32_t S,
                         const double* cls, const double* pos, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                         int32_t w_cols, int32_t level, fhelin_ct** outs, double* proj_out) {
    NEED(c && (emb || (tokens && table)) && cls && pos && E_w && E_b && F_w && F_b && outs);
    FHELIN_TRY
    if (S < 1) throw Error(FHELIN_ERR_ARG, "ingest: need at least one token");
    if (level < 0 || level > c->ctx.L) throw Error(FHELIN_ERR_ARG, "ingest: level out of range");
    const int n_vec = 64 + S + 1;
    std::vector<int> drop(n_vec);
    for (int i = 0; i < n_vec; ++i) drop[i] = std::max(0, std::min(c->ctx.L - level, c->plan.next_drop(c->ctx.L + 1 - level)));
    const int first_ordinal = c->plan.next_ordinal - n_vec;
    std::vector<double> po;
    std::vector<CtPtr> r = c->cl.ingest_sample(emb, tokens, table, vocab, S, cls, pos, E_w, E_b, F_w, F_b, w_cols, level, drop,
                                               proj_out ? &po : nullptr);
    if (proj_out) std::memcpy(proj_out, po.data(), po.size() * sizeof(double));
    for (int i = 0; i < n_vec; ++i) {
        outs[i] = wrap(c, r[i]);
        if (c->plan.live(outs[i]->node, outs[i]->node_epoch)) c->plan.nodes[outs[i]->node].ordinal = first_ordinal + i;
    }
    FHELIN_CATCH
}
int fhelin_ctx_set_host_encode(fhelin_ctx* c, int32_t on) {
    NEED(c);
    c->ctx.host_encode = on != 0;
    return FHELIN_OK;
}
int fhelin_debug_sample(fhelin_ctx* c, int32_t kind, int32_t n_poly, int64_t* out, size_t cap) {
    NEED(c && out);
    FHELIN_TRY
    if (n_poly < 1 || cap < (size_t)n_poly * c->ctx.N) throw Error(FHELIN_ERR_ARG, "debug_sample: buffer too small");
    std::vector<long> v = c->cl.debug_sample(kind, n_poly);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    FHELIN_CATCH
}
int fhelin_decrypt(fhelin_ctx* c, const fhelin_ct* ct, double* out, int32_t slots) {
    NEED(c && ct && out);
    FHELIN_TRY
    if (c->plan.live(ct->node, ct->node_epoch)) c->plan.terminal(ct->node, 2);
    c->plan.check_terminal(*ct_in(c, ct), 2);
    auto v = c->cl.decrypt(ct_in(c, ct), slots);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    FHELIN_CATCH
}
int fhelin_ct_import(fhelin_ctx* c, const uint64_t* limbs, int32_t npoly, int32_t ell, int32_t deg, double scale, int32_t slots,
                     fhelin_ct** out) {
    NEED(c && limbs && out);
    FHELIN_TRY
    CtPtr p = c->ev.new_ct(npoly, ell, deg, scale, slots);
    hip_check(hipMemcpyAsync(p->d, limbs, p->words() * 8, hipMemcpyHostToDevice, c->ctx.stream), "ct import");
    c->ctx.sync();
    *out = wrap(c, p);
    FHELIN_CATCH
}
int fhelin_ct_export(fhelin_ctx* c, const fhelin_ct* ct, uint64_t* out, size_t cap) {
    NEED(c && ct && out);
    FHELIN_TRY
    if (c->plan.live(ct->node, ct->node_epoch)) c->plan.terminal(ct->node, 2);
    const CtPtr& p = ct_in(c, ct);
    c->plan.check_terminal(*p, 2);
    if (cap < p->words()) throw Error(FHELIN_ERR_ARG, "buffer too small");
    hip_check(hipMemcpyAsync(out, p->d, p->words() * 8, hipMemcpyDeviceToHost, c->ctx.stream), "ct export");
    c->ctx.sync();
    FHELIN_CATCH
}
int fhelin_ct_export_device(fhelin_ctx* c, const fhelin_ct* ct, uint64_t* d_out, size_t cap) {
    NEED(c && ct && d_out);
    FHELIN_TRY
    if (c->plan.live(ct->node, ct->node_epoch)) c->plan.terminal(ct->node, 2);
    const CtPtr& p = ct_in(c, ct);
    c->plan.check_terminal(*p, 2);
    if (cap < p->words()) throw Error(FHELIN_ERR_ARG, "buffer too small");
    hip_check(hipMemcpyAsync(d_out, p->d, p->words() * 8, hipMemcpyDeviceToDevice, c->ctx.stream), "ct export (device)");
    c->ctx.sync();   // the caller's framework reads the buffer on its own stream next
    FHELIN_CATCH
}
int fhelin_ct_import_device(fhelin_ctx* c, const uint64_t* d_limbs, int32_t npoly, int32_t ell, int32_t deg, double scale_hi,
                            double scale_lo, int32_t slots, fhelin_ct** out) {
    NEED(c && d_limbs && out);
    FHELIN_TRY
    CtPtr p = c->ev.new_ct(npoly, ell, deg, (long double)scale_hi + (long double)scale_lo, slots);
    hip_check(hipMemcpyAsync(p->d, d_limbs, p->words() * 8, hipMemcpyDeviceToDevice, c->ctx.stream), "ct import (device)");
    c->ctx.sync();
    *out = wrap(c, p);
    FHELIN_CATCH
}
int fhelin_ct_scale(const fhelin_ct* ct, double* scale_hi, double* scale_lo) {
    NEED(ct && scale_hi && scale_lo);
    if (!ct->p && (ct->lazy || ct->heavy)) {
        try {
            force(ct->owner, ct);
        } catch (const std::exception& e) {
            return capi_fail(FHELIN_ERR_INTERNAL, e.what());
        }
    }
    const long double s = ct->p->scale;
    *scale_hi = (double)s;
    *scale_lo = (double)(s - (long double)*scale_hi);
    return FHELIN_OK;
}
int fhelin_ct_info(const fhelin_ct* ct, int32_t* npoly, int32_t* ell, int32_t* level, int32_t* deg, double* scale, int32_t* slots) {
    NEED(ct);
    if (!ct->p && (ct->lazy || ct->heavy)) {  // a deferred row / heavy operation: its shape is known only once evaluated
        try {
            force(ct->owner, ct);
        } catch (const fhelin::Error& e) {
            return capi_fail(e.code, e.what());
        } catch (const std::exception& e) {
            return capi_fail(FHELIN_ERR_INTERNAL, e.what());
        }
    }
    if (npoly) *npoly = ct->p->npoly;
    if (ell) *ell = ct->p->ell;
    if (level) *level = ct->p->level();
    if (deg) *deg = ct->p->deg;
    if (scale) *scale = (double)ct->p->scale;
    if (slots) *slots = ct->p->slots;
    return FHELIN_OK;
}
int fhelin_ct_clone(fhelin_ctx* c, const fhelin_ct* ct, fhelin_ct** out) {
    NEED(c && ct && out);
    FHELIN_TRY
    // ciphertext objects are immutable (every operation returns a new one), so Clone() is a new HANDLE to the same
    // residues: no copy — and rows that are clones of one ciphertext can be recognised as identical (Composite::matmul_pt)
    *out = wrap(c, ct_in(c, ct));
    FHELIN_CATCH
}
void fhelin_ct_free(fhelin_ct* ct) { delete ct; }

#define BINOP(name, expr)                                                                   \
    int name(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out) {      \
        NEED(c && a && b && out);                                                           \
        FHELIN_TRY                                                                          \
        const fhelin_ct* both[2] = {a, b};                                                  \
        force_many(c, both, 2); /* deferred operands of one group: one batched call */      \
        *out = wrap(c, expr);                                                                  \
        FHELIN_CATCH                                                                        \
    }
int fhelin_add(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out) {
    NEED(c && a && b && out);
    FHELIN_TRY
    const fhelin_ct* both[2] = {a, b};
    // deferred rows among the operands are evaluated now (their own batching rules apply); the SUM is deferred: additions issued
    // in a loop over independent rows are aligned and added as one batch when a result is first read (capi_internal.h LazyHeavy)
    for (const fhelin_ct* h : both)
        if (!h->p && h->lazy) force(c, h);
    if (defer_allowed(c)) {
        *out = defer_add(c, a, b);
        return FHELIN_OK;
    }
    force_many(c, both, 2);
    *out = wrap(c, c->ev.add(ct_in(c, a), ct_in(c, b)));
    FHELIN_CATCH
}
BINOP(fhelin_sub, c->ev.sub(ct_in(c, a), ct_in(c, b)))
BINOP(fhelin_mult, c->ev.mult(ct_in(c, a), ct_in(c, b)))
BINOP(fhelin_raw_mult_relin, (c->ev.relin_key ? c->ev.raw_mult_relin(ct_in(c, a), ct_in(c, b), *c->ev.relin_key)
                                              : throw Error(FHELIN_ERR_KEY, "no relinearisation key")))

int fhelin_negate(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.negate(ct_in(c, a)));
    FHELIN_CATCH
}
int fhelin_add_plain(fhelin_ctx* c, const fhelin_ct* a, const fhelin_pt* p, fhelin_ct** out) {
    NEED(c && a && p && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.add_plain(ct_in(c, a), p->p));
    FHELIN_CATCH
}
int fhelin_mult_plain(fhelin_ctx* c, const fhelin_ct* a, const fhelin_pt* p, fhelin_ct** out) {
    NEED(c && a && p && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.mult_plain(ct_in(c, a), p->p));
    FHELIN_CATCH
}
int fhelin_rotate(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.rotate(ct_in(c, a), index));
    FHELIN_CATCH
}
int fhelin_rotate_many(fhelin_ctx* c, const fhelin_ct* a, const int32_t* indices, int32_t n, fhelin_ct** outs) {
    NEED(c && a && indices && outs && n >= 0);
    FHELIN_TRY
    std::vector<CtPtr> r = c->ev.rotate_many(ct_in(c, a), std::vector<int>(indices, indices + n));
    for (int i = 0; i < n; ++i) outs[i] = wrap(c, r[i]);
    FHELIN_CATCH
}
int fhelin_rotate_each(fhelin_ctx* c, const fhelin_ct* const* v, const int32_t* indices, int32_t n, fhelin_ct** outs) {
    NEED(c && v && indices && outs && n >= 0);
    FHELIN_TRY
    std::vector<CtPtr> in;
    for (int i = 0; i < n; ++i)
        if (!v[i]) throw Error(FHELIN_ERR_ARG, "null ciphertext in array");
    force_many(c, v, n);
    for (int i = 0; i < n; ++i) in.push_back(ct_in(c, v[i]));
    std::vector<CtPtr> r = c->ev.rotate_each(in, std::vector<int>(indices, indices + n));
    for (int i = 0; i < n; ++i) outs[i] = wrap(c, r[i]);
    FHELIN_CATCH
}
int fhelin_hoisted_dot(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* const* pts, const int32_t* indices, int32_t n_rot,
                       int32_t rescale, fhelin_ct** outs) {
    NEED(c && v && pts && indices && outs && n >= 0 && n_rot >= 1);
    FHELIN_TRY
    std::vector<CtPtr> in;
    std::vector<PtPtr> p;
    for (int i = 0; i < n; ++i)
        if (!v[i]) throw Error(FHELIN_ERR_ARG, "null ciphertext in array");
    for (int i = 0; i <= n_rot; ++i) {
        if (!pts[i]) throw Error(FHELIN_ERR_ARG, "null plaintext in array");
        p.push_back(pts[i]->p);
    }
    force_many(c, v, n);
    for (int i = 0; i < n; ++i) in.push_back(ct_in(c, v[i]));
    std::vector<CtPtr> r = c->ev.hoisted_dot_rows(in, p, std::vector<int>(indices, indices + n_rot), rescale != 0);
    for (int i = 0; i < n; ++i) outs[i] = wrap(c, r[i]);
    FHELIN_CATCH
}
int fhelin_rotate_sum(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const int32_t* indices, int32_t n_rot, fhelin_ct** outs) {
    NEED(c && v && indices && outs && n >= 0 && n_rot >= 1);
    FHELIN_TRY
    std::vector<CtPtr> in;
    for (int i = 0; i < n; ++i)
        if (!v[i]) throw Error(FHELIN_ERR_ARG, "null ciphertext in array");
    force_many(c, v, n);
    for (int i = 0; i < n; ++i) in.push_back(ct_in(c, v[i]));
    std::vector<CtPtr> r = c->ev.rotate_sum_batch(in, std::vector<int>(indices, indices + n_rot));
    for (int i = 0; i < n; ++i) outs[i] = wrap(c, r[i]);
    FHELIN_CATCH
}
int fhelin_rotate_each_sum(fhelin_ctx* c, const fhelin_ct* const* v, const int32_t* indices, int32_t n, fhelin_ct** out) {
    NEED(c && v && indices && out && n >= 1);
    FHELIN_TRY
    std::vector<CtPtr> in;
    for (int i = 0; i < n; ++i)
        if (!v[i]) throw Error(FHELIN_ERR_ARG, "null ciphertext in array");
    force_many(c, v, n);
    for (int i = 0; i < n; ++i) in.push_back(ct_in(c, v[i]));
    *out = wrap(c, c->ev.rotate_each_sum(in, std::vector<int>(indices, indices + n)));
    FHELIN_CATCH
}
int fhelin_raw_modraise(fhelin_ctx* c, const fhelin_ct* a, int32_t new_ell, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.raw_modraise(ct_in(c, a), new_ell));
    FHELIN_CATCH
}
int fhelin_raw_phase(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    const CtPtr& p = ct_in(c, a);
    *out = wrap(c, c->cl.phase(p, p->ell));
    FHELIN_CATCH
}
int fhelin_rescale(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.rescale(ct_in(c, a)));
    FHELIN_CATCH
}
int fhelin_level_reduce(fhelin_ctx* c, const fhelin_ct* a, int32_t new_ell, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.level_reduce(ct_in(c, a), new_ell));
    FHELIN_CATCH
}
int fhelin_raw_rescale(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    *out = wrap(c, c->ev.raw_rescale(ct_in(c, a)));
    FHELIN_CATCH
}
int fhelin_raw_rotate(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** out) {
    NEED(c && a && out);
    FHELIN_TRY
    const u64 g = c->ctx.galois_element(index);
    auto it = c->ev.rot_keys.find(g);
    if (it == c->ev.rot_keys.end() || !it->second) throw Error(FHELIN_ERR_KEY, "no rotation key for this index");
    *out = wrap(c, c->ev.raw_rotate(ct_in(c, a), g, *it->second));
    FHELIN_CATCH
}

}  // extern "C"
