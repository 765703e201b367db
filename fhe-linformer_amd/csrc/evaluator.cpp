#include "evaluator.h"
#include <map>
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>

namespace fhelin {

DevBlock::~DevBlock() {
    if (d && ctx) {
        try { ctx->pool.free(d); } catch (...) {}
    }
}
Ciphertext::~Ciphertext() {
    if (async_ev) (void)hipEventDestroy(async_ev);
    if (d && ctx && !block) {
        try { ctx->pool.free(d); } catch (...) {}
    }
}
Encoding::~Encoding() {
    if (ready) (void)hipEventDestroy(ready);
    if (d && ctx) {
        try { ctx->pool.free(d, lanes_ordered); } catch (...) {}   // every lane that used it may still have reads queued
    }
}
EvalKey::~EvalKey() {
    if (d && ctx) {
        try { ctx->pool.free(d); } catch (...) {}
    }
    if (d_perm && ctx) {
        try { ctx->pool.free(d_perm); } catch (...) {}
    }
}

static void launch_ok(const char* what) { hip_check(hipGetLastError(), what); }

CtPtr Evaluator::new_ct(int npoly, int ell, int deg, long double scale, int slots) {
    c_.require_device();
    if (ell < 1 || ell > c_.L + 1 || npoly < 1 || npoly > 3) throw Error(FHELIN_ERR_ARG, "new_ct: bad shape");
    auto ct = std::make_shared<Ciphertext>();
    ct->ctx = &c_;
    ct->npoly = npoly;
    ct->ell = ell;
    ct->deg = deg;
    ct->scale = scale;
    ct->slots = slots;
    ct->d = c_.dalloc<u64>(ct->words());
    return ct;
}

CtPtr Evaluator::clone(const CtPtr& a) {
    CtPtr o = new_ct(a->npoly, a->ell, a->deg, a->scale, a->slots);
    hip_check(hipMemcpyAsync(o->d, a->d, a->words() * 8, hipMemcpyDeviceToDevice, c_.stream), "clone");
    return o;
}

std::vector<CtPtr> Evaluator::new_ct_batch(int count, int npoly, int ell, int deg, long double scale, int slots) {
    c_.require_device();
    if (count < 1 || ell < 1 || ell > c_.L + 1 || npoly < 1 || npoly > 3) throw Error(FHELIN_ERR_ARG, "new_ct_batch: bad shape");
    auto blk = std::make_shared<DevBlock>();
    blk->ctx = &c_;
    const size_t words = (size_t)npoly * ell * c_.N;
    blk->d = c_.dalloc<u64>(words * count);
    std::vector<CtPtr> v;
    for (int i = 0; i < count; ++i) {
        auto ct = std::make_shared<Ciphertext>();
        ct->ctx = &c_;
        ct->npoly = npoly;
        ct->ell = ell;
        ct->deg = deg;
        ct->scale = scale;
        ct->slots = slots;
        ct->block = blk;
        ct->d = blk->d + words * i;
        v.push_back(ct);
    }
    return v;
}

u64* Evaluator::contiguous_base(const std::vector<CtPtr>& v) {
    if (v.empty() || !v[0]->block) return nullptr;
    const size_t words = v[0]->words();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i]->block != v[0]->block || v[i]->d != v[0]->d + words * i || v[i]->words() != words) return nullptr;
    return v[0]->d;
}

std::vector<CtPtr> Evaluator::make_contiguous(const std::vector<CtPtr>& v, int site) {
    if (v.size() <= 1 || contiguous_base(v)) return v;
    if (site >= 0 && site < 8) gather_copies[site] += v.size();   // FHELIN_COPY_STATS=1 prints these when the context goes away
    std::vector<CtPtr> o = new_ct_batch((int)v.size(), v[0]->npoly, v[0]->ell, v[0]->deg, v[0]->scale, v[0]->slots);
    for (size_t lo = 0; lo < v.size(); lo += EwItems::MAX_ITEMS) {   // one launch per 32 ciphertexts (a copy each costs 6.5 us in the stream)
        EwItems it;
        it.n = (int)std::min<size_t>(EwItems::MAX_ITEMS, v.size() - lo);
        it.vecs = v[0]->npoly * v[0]->ell;
        it.b_vecs = 0;
        for (int i = 0; i < it.n; ++i) {
            if (v[lo + i]->words() != v[0]->words()) throw Error(FHELIN_ERR_INTERNAL, "make_contiguous: operands of different shapes");
            it.out[i] = o[lo + i]->d;
            it.a[i] = v[lo + i]->d;
            it.b[i] = nullptr;
        }
        launch_ew_items(c_.dt, it, 4, v[0]->ell, c_.stream);
    }
    hip_check(hipGetLastError(), "batch gather");
    return o;
}

KeyPtr Evaluator::new_key() {
    c_.require_device();
    auto k = std::make_shared<EvalKey>();
    k->ctx = &c_;
    k->digits = c_.digits_at(c_.L + 1);
    k->d = c_.dalloc<u64>(k->words());
    return k;
}

// ------------------------------------------------------------------------------------------------
void Evaluator::keyswitch(const u64* c_ntt, int ell, const EvalKey& key, u64* out, const u64* add0, const u64* add1,
                          const u32* map, const u64* post) {
    keyswitch_batch(1, c_ntt, 0, ell, key, out, 0, add0, add1, 0, map, post, 0);
}

void Evaluator::keyswitch_batch(int B, const u64* c_ntt, size_t c_stride, int ell, const EvalKey& key, u64* out, size_t out_stride,
                                const u64* add0, const u64* add1, size_t add_stride, const u32* map, const u64* post,
                                size_t post_stride) {
    keyswitch_impl(B, nullptr, c_ntt, c_stride, ell, &key, out, out_stride, add0, add1, add_stride, map, post, post_stride);
}

void Evaluator::keyswitch_rows(const KsRows& rows, const u64* c_ntt, size_t c_stride, int ell, u64* out, size_t out_stride,
                               const u64* add0, size_t add_stride) {
    const int B = (int)rows.keys.size();
    if (B < 1) return;
    if (B > KsShape::MAX_ROWS || rows.maps.size() != rows.keys.size()) throw Error(FHELIN_ERR_ARG, "keyswitch_rows: bad row count");
    keyswitch_impl(B, &rows, c_ntt, c_stride, ell, nullptr, out, out_stride, add0, nullptr, add_stride, nullptr, nullptr, 0);
}

void Evaluator::keyswitch_impl(int B, const KsRows* rows, const u64* c_ntt, size_t c_stride, int ell, const EvalKey* key, u64* out,
                               size_t out_stride, const u64* add0, const u64* add1, size_t add_stride, const u32* map, const u64* post,
                               size_t post_stride) {
    c_.require_device();
    if (c_.K < 1) throw Error(FHELIN_ERR_STATE, "hybrid key switching needs at least one special prime");
    if (B < 1) return;
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1;
    const LevelTables& lt = c_.lvl[ell];
    const int nt = ell + K;
    KsShape sh{ell, K, c_.alpha, lt.beta, L1, B, c_stride, out_stride, add_stride, post_stride};
    const bool shared = rows && rows->shared_input;
    if (rows) {
        sh.per_row = 1;
        sh.shared_input = shared ? 1 : 0;
        for (int b = 0; b < B; ++b) {
            sh.evk_row[b] = rows->keys[b]->d;
            sh.map_row[b] = rows->maps[b];
        }
    }
    const int Bu = shared ? 1 : B;  // polynomials that go through ModUp
    KsShape shu = sh;
    shu.batch = Bu;
    hipStream_t s = c_.stream;
    u64* cc = c_.dalloc<u64>((size_t)Bu * ell * N);
    c_.stats.keyswitch += (u64)B;
    c_.stats.keyswitch_limbs += (u64)B * ell;
    {
        // out of place: cc = INTT(c); the inputs of a batch are strided (c1 of consecutive ciphertexts)
        LimbBatch ib{cc, Bu * ell, nullptr, 0, ell, c_ntt};
        if (Bu > 1 && c_stride != (size_t)ell * N) {
            ib.src_group = ell;
            ib.src_group_stride = c_stride;
        }
        c_.ntt(ib, true);
    }
    u64* ext = c_.dalloc<u64>((size_t)Bu * lt.beta * nt * N);
    launch_modup_conv(c_.dt, shu, ext, cc, c_ntt, lt.up_hatinv, lt.up_hatmod_r2, s);   // digits times 2^64: launch_ks_inner ends in redc128
    LimbBatch eb{ext, Bu * lt.beta * nt, lt.ext_limb_tab, 0, 1};
    eb.tab_len = lt.beta * nt;
    eb.lazy_out = true;  // only K7 reads the digits: it takes any residue below 2^60
    c_.ntt(eb, false, Bu * (lt.beta * nt - ell));
    u64* accQ = c_.dalloc<u64>((size_t)B * 2 * ell * N);
    u64* accP = c_.dalloc<u64>((size_t)B * 2 * K * N);
    launch_ks_inner(c_.dt, sh, accQ, accP, ext, key ? key->d : nullptr, c_ntt, s);
    c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
    u64* conv = c_.dalloc<u64>((size_t)B * 2 * ell * N);
    launch_moddown_conv(c_.dt, sh, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
    if (c_.fuse_moddown) {
        // K8b rides in the row pass of NTT(conv): (accQ - NTT(conv)) * P^-1 + add (+ post) is formed in registers and
        // written through the inverse automorphism map — NTT(conv) never goes to memory
        NttModDown md;
        md.accQ = accQ;
        md.out = out;
        md.add0 = add0;
        md.add1 = add1;
        md.post = post;
        md.pinv = c_.d_pinv;
        md.ell = ell;
        md.out_stride = out_stride;
        md.add_stride = add_stride;
        md.post_stride = post_stride;
        if (rows) {
            md.per_row = 1;
            for (int b = 0; b < B; ++b) md.invmap_row[b] = c_.automorph_inverse_of(rows->maps[b]);
        } else {
            md.invmap = c_.automorph_inverse_of(map);
        }
        c_.ntt_moddown(LimbBatch{conv, B * 2 * ell, nullptr, 0, ell}, md);
    } else {
        c_.ntt(LimbBatch{conv, B * 2 * ell, nullptr, 0, ell}, false);
        launch_moddown_finish(c_.dt, sh, out, accQ, conv, c_.d_pinv, add0, add1, map, post, s);
    }
    launch_ok("keyswitch");
    c_.pool.free(cc);
    c_.pool.free(ext);
    c_.pool.free(accQ);
    c_.pool.free(accP);
    c_.pool.free(conv);
}

bool Evaluator::have_rotation_keys(const std::vector<int>& indices, int slots) const {
    const int ns = slots > 0 ? slots : (1 << c_.prm.log_slots);
    for (int r : indices) {
        if (r % ns == 0) return false;
        if (!rot_keys.count(c_.galois_element(r))) return false;
    }
    return !indices.empty();
}

const u64* Evaluator::permuted(const EvalKey& key, const u32* map) {
    if (key.d_perm) return key.d_perm;
    const int nvec = key.digits * 2 * (c_.L + 1 + c_.K);
    key.d_perm = c_.dalloc<u64>(key.words());
    launch_automorph_pack30(c_.dt, key.d_perm, key.d, map, nvec, c_.stream);
    launch_ok("key permutation");
    // built once, read from every stream afterwards: finish it before anyone else can see the pointer
    hip_check(hipStreamSynchronize(c_.stream), "key permutation sync");
    return key.d_perm;
}

std::vector<CtPtr> Evaluator::rotate_sum_batch(const std::vector<CtPtr>& vin, const std::vector<int>& indices) {
    if (vin.empty()) return {};
    const int R = (int)indices.size();
    if (R < 1 || R > KsShape::MAX_ROT) throw Error(FHELIN_ERR_ARG, "rotate_sum_batch: 1..7 rotations");
    if (!have_rotation_keys(indices, vin[0]->slots)) throw Error(FHELIN_ERR_KEY, "rotate_sum_batch: missing rotation key");
    if (c_.K < 1) throw Error(FHELIN_ERR_STATE, "hybrid key switching needs at least one special prime");
    std::vector<const EvalKey*> keys;
    std::vector<const u32*> maps;
    for (int r : indices) {
        const u64 g = c_.galois_element(r);
        keys.push_back(rot_keys.at(g).get());
        maps.push_back(c_.automorph_map(g));
    }
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1;
    hipStream_t s = c_.stream;
    std::vector<CtPtr> out(vin.size());
    std::vector<char> done(vin.size(), 0);
    for (size_t first = 0; first < vin.size(); ++first) {
        if (done[first]) continue;
        if (vin[first]->npoly != 2) throw Error(FHELIN_ERR_STATE, "rotate: ciphertext must have 2 components");
        std::vector<size_t> idx;
        for (size_t i = first; i < vin.size() && (int)idx.size() < batch_limit; ++i) {
            const CtPtr &a = vin[first], &b = vin[i];
            if (!done[i] && b->npoly == 2 && b->ell == a->ell && b->deg == a->deg && fabsl(b->scale / a->scale - 1.0L) < 1e-9L)
                idx.push_back(i);
        }
        std::vector<CtPtr> chunk;
        for (size_t i : idx) chunk.push_back(vin[i]);
        chunk = make_contiguous(chunk, 0);
        const int B = (int)chunk.size(), ell = chunk[0]->ell;
        const size_t pn = (size_t)ell * N, ctw = 2 * pn;
        const LevelTables& lt = c_.lvl[ell];
        const int nt = ell + K;
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell, chunk[0]->deg, chunk[0]->scale, chunk[0]->slots);
        const u64* base = chunk[0]->d;
        KsShape sh{ell, K, c_.alpha, lt.beta, L1, B, ctw, ctw, pn, ctw};
        sh.n_rot = R;
        sh.lds_digits = c_.lds_digits ? 1 : 0;
        for (int r = 0; r < R; ++r) {
            sh.evk_rot[r] = permuted(*keys[r], maps[r]);
            sh.map_rot[r] = maps[r];
            // the automorphism of g keeps every 512-coefficient tile in place iff g = 1 mod N/256 (index bits above the tile
            // correspond to the low bits of the odd exponent 2 bitrev(j) + 1, which multiplication by such a g leaves alone)
            if (c_.galois_element(indices[r]) % (u64)(c_.N / 256) != 1) sh.lds_digits = 0;
        }
        // accounting in units of the reference's rotations: R = 2^k - 1 merged terms stand for k tree steps
        int steps = 0;
        while ((1 << steps) < R + 1) ++steps;
        c_.stats.keyswitch += (u64)B * steps;
        c_.stats.keyswitch_limbs += (u64)B * steps * ell;
        // ModUp of c1, once for all rotations
        u64* cc = c_.dalloc<u64>((size_t)B * ell * N);
        {
            LimbBatch ib{cc, B * ell, nullptr, 0, ell, base + pn};
            if (B > 1) {
                ib.src_group = ell;
                ib.src_group_stride = ctw;
            }
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)B * lt.beta * nt * N);
        launch_modup_conv(c_.dt, sh, ext, cc, base + pn, lt.up_hatinv, lt.up_hatmod, s);
        LimbBatch eb{ext, B * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, B * (lt.beta * nt - ell));
        // all rotated inner products, gathered and accumulated in the extended basis; the c0 parts likewise
        u64* accQ = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)B * 2 * K * N);
        launch_ks_inner_multi(c_.dt, sh, accQ, accP, ext, base + pn, s);
        u64* c0sum = nullptr;
        if (c_.fuse_gather) {
            sh.gsrc = base;      // the epilogue gathers the rotated c0 parts itself
            sh.gsrc_stride = ctw;
        } else {
            c0sum = c_.dalloc<u64>((size_t)B * ell * N);
            launch_gather_sum(c_.dt, sh, c0sum, base, ctw, s);
        }
        // one ModDown; the epilogue adds the gathered c0 parts and the unrotated input
        c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
        u64* conv = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        launch_moddown_conv(c_.dt, sh, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
        c_.ntt(LimbBatch{conv, B * 2 * ell, nullptr, 0, ell}, false);
        launch_moddown_finish(c_.dt, sh, o[0]->d, accQ, conv, c_.d_pinv, c0sum, nullptr, nullptr, base, s);
        launch_ok("rotate_sum_batch");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        if (c0sum) c_.pool.free(c0sum);
        c_.pool.free(conv);
        for (int b = 0; b < B; ++b) {
            o[b]->scale = vin[idx[b]]->scale;
            out[idx[b]] = o[b];
            done[idx[b]] = 1;
        }
    }
    return out;
}

Evaluator::FoldedKey Evaluator::folded_key(const PtPtr& p, int index, long double scale) {
    const u64 g = c_.galois_element(index);
    auto kit = rot_keys.find(g);
    if (kit == rot_keys.end()) throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(index) + " (EvalRotateKeyGen list)");
    for (size_t i = 0; i < folded_keys.size(); ++i) {
        const FoldedKey& f = folded_keys[i];
        if (f.pt.get() == p.get() && f.key.get() == kit->second.get() && f.index == index && fabsl(f.scale / scale - 1.0L) < 1e-12L) {
            FoldedKey hit = f;
            if (i + 1 != folded_keys.size()) {   // least recently used goes first when the cache is full
                folded_keys.erase(folded_keys.begin() + i);
                folded_keys.push_back(hit);
            }
            return hit;
        }
    }
    FoldedKey f;
    f.pt = p;
    f.key = kit->second;
    f.index = index;
    f.scale = scale;
    const int nl = c_.L + 1 + c_.K;
    f.enc = p->at(nl, scale);
    f.d = std::make_shared<DevBlock>();
    f.d->ctx = &c_;
    f.d->d = c_.dalloc<u64>(f.key->words());
    launch_fold_key(c_.dt, f.d->d, f.key->d, c_.automorph_map(g), f.enc->d, f.key->digits * 2 * nl, c_.stream);
    launch_ok("fold_key");
    // built once, read from every stream afterwards: finish it before anyone else can see the pointer
    hip_check(hipStreamSynchronize(c_.stream), "fold_key sync");
    if (folded_keys.size() >= MAX_FOLDED) {
        // the evicted copy may still be read by work in flight on any stream
        hip_check(hipStreamSynchronize(c_.main_stream), "folded key eviction sync");
        for (int k = 1; k <= c_.n_lanes; ++k) hip_check(hipStreamSynchronize(c_.lane_stream[k]), "folded key eviction sync (lane)");
        folded_keys.erase(folded_keys.begin());
    }
    folded_keys.push_back(f);
    return f;
}

std::vector<CtPtr> Evaluator::hoisted_dot_rows(const std::vector<CtPtr>& xin, const std::vector<PtPtr>& pts, const std::vector<int>& indices,
                                              bool rescale_out) {
    if (xin.empty()) return {};
    const int R = (int)indices.size();
    if (R < 1 || R > KsShape::MAX_ROT || (int)pts.size() != R + 1)
        throw Error(FHELIN_ERR_ARG, "hoisted_dot_rows: 1..7 rotations, one plaintext per rotation + the unrotated term's");
    if (!have_rotation_keys(indices, xin[0]->slots)) throw Error(FHELIN_ERR_KEY, "hoisted_dot_rows: missing rotation key");
    if (c_.K < 1) throw Error(FHELIN_ERR_STATE, "hybrid key switching needs at least one special prime");
    const int ns = xin[0]->slots > 0 ? xin[0]->slots : (1 << c_.prm.log_slots);
    for (int r : indices)
        if (r % ns == 0) throw Error(FHELIN_ERR_ARG, "hoisted_dot_rows: a rotation by 0 is the unrotated term (pts[0])");
    std::vector<CtPtr> x = xin;
    {   // degree-2 operands are rescaled first (as before any product with a plaintext)
        std::vector<CtPtr> need;
        std::vector<size_t> pos;
        for (size_t i = 0; i < x.size(); ++i)
            if (x[i]->deg >= 2) {
                need.push_back(x[i]);
                pos.push_back(i);
            }
        if (!need.empty()) {
            std::vector<CtPtr> r = rescale_batch(need);
            for (size_t k = 0; k < pos.size(); ++k) x[pos[k]] = r[k];
        }
    }
    std::vector<const u32*> maps;
    for (int r : indices) maps.push_back(c_.automorph_map(c_.galois_element(r)));
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1;
    hipStream_t s = c_.stream;
    std::vector<CtPtr> out(x.size());
    std::vector<char> done(x.size(), 0);
    for (size_t first = 0; first < x.size(); ++first) {
        if (done[first]) continue;
        if (x[first]->npoly != 2) throw Error(FHELIN_ERR_STATE, "hoisted_dot_rows: ciphertext must have 2 components");
        std::vector<size_t> idx;
        for (size_t i = first; i < x.size() && (int)idx.size() < batch_limit; ++i) {
            const CtPtr &a = x[first], &b = x[i];
            if (!done[i] && b->npoly == 2 && b->ell == a->ell && b->deg == a->deg && fabsl(b->scale / a->scale - 1.0L) < 1e-9L) idx.push_back(i);
        }
        std::vector<CtPtr> chunk;
        for (size_t i : idx) chunk.push_back(x[i]);
        chunk = make_contiguous(chunk, 0);
        const int B = (int)chunk.size(), ell = chunk[0]->ell;
        const size_t pn = (size_t)ell * N, ctw = 2 * pn;
        const LevelTables& lt = c_.lvl[ell];
        const int nt = ell + K;
        const long double sf = c_.sf_real[chunk[0]->level()];
        const bool merged = rescale_out && ell >= 2 && K + 1 <= 16;
        if (rescale_out && !merged) throw Error(FHELIN_ERR_STATE, "hoisted_dot_rows: no limb left to drop");
        const int oell = merged ? ell - 1 : ell;
        std::vector<CtPtr> o = new_ct_batch(B, 2, oell, merged ? chunk[0]->deg : chunk[0]->deg + 1, chunk[0]->scale * sf, chunk[0]->slots);
        const u64* base = chunk[0]->d;
        KsShape sh{ell, K, c_.alpha, lt.beta, L1, B, ctw, (size_t)2 * oell * N, pn, ctw};
        sh.n_rot = R;
        sh.lds_digits = c_.lds_digits ? 1 : 0;
        HoistAdd h;
        h.n_rot = R;
        std::vector<FoldedKey> hold;            // the folded keys and encodings of this launch set stay alive across a cache eviction
        const std::shared_ptr<Encoding> e0 = pts[0]->at(ell, sf);
        h.v[0] = e0->d;
        for (int r = 0; r < R; ++r) {
            hold.push_back(folded_key(pts[r + 1], indices[r], sf));
            sh.evk_rot[r] = hold.back().d->d;
            sh.map_rot[r] = maps[r];
            h.v[r + 1] = hold.back().enc->d;
            h.map[r] = maps[r];
            if (c_.galois_element(indices[r]) % (u64)(c_.N / 256) != 1) sh.lds_digits = 0;
        }
        c_.stats.keyswitch += (u64)B * R;
        c_.stats.keyswitch_limbs += (u64)B * R * ell;
        c_.stats.ct_pt_mult += (u64)B * (R + 1);
        c_.stats.ct_pt_limbs += (u64)B * (R + 1) * ell;
        if (merged) {
            c_.stats.rescale += (u64)B;
            c_.stats.rescale_limbs += (u64)B * ell;
        }
        // ModUp of c1, once for all rotations
        u64* cc = c_.dalloc<u64>((size_t)B * ell * N);
        {
            LimbBatch ib{cc, B * ell, nullptr, 0, ell, base + pn};
            if (B > 1) {
                ib.src_group = ell;
                ib.src_group_stride = ctw;
            }
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)B * lt.beta * nt * N);
        launch_modup_conv(c_.dt, sh, ext, cc, base + pn, lt.up_hatinv, lt.up_hatmod, s);
        LimbBatch eb{ext, B * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, B * (lt.beta * nt - ell));
        // sum_r V_r . sigma_r(d * evk_r) in the extended basis: the merged inner product with the folded keys
        u64* accQ = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)B * 2 * K * N);
        launch_ks_inner_multi(c_.dt, sh, accQ, accP, ext, base + pn, s);
        u64* pre = nullptr;
        u64* conv = nullptr;
        u64* top = nullptr;
        if (!merged) {
            // what does not pass through the key switch: V_0 (c0, c1) + sum_r V_r sigma_r(c0)
            pre = c_.dalloc<u64>((size_t)B * ctw);
            launch_hoist_addends(c_.dt, sh, h, pre, base, s);
            // one ModDown
            c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
            conv = c_.dalloc<u64>((size_t)B * 2 * ell * N);
            launch_moddown_conv(c_.dt, sh, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
            c_.ntt(LimbBatch{conv, B * 2 * ell, nullptr, 0, ell}, false);
            launch_moddown_finish(c_.dt, sh, o[0]->d, accQ, conv, c_.d_pinv, nullptr, nullptr, nullptr, pre, s);
        } else {
            // the same addends times P into the accumulator's Q part, then P and the top limb are dropped together
            h.acc = accQ;
            h.pmod = c_.d_pmod;
            launch_hoist_addends(c_.dt, sh, h, nullptr, base, s);
            c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
            top = c_.dalloc<u64>((size_t)B * 2 * N);
            {
                LimbBatch tb{top, B * 2, nullptr, ell - 1, 1, accQ + (size_t)(ell - 1) * N};
                tb.src_group = 1;
                tb.src_group_stride = pn;
                c_.ntt(tb, true);
            }
            conv = c_.dalloc<u64>((size_t)B * 2 * (ell - 1) * N);
            launch_moddown_rescale_conv(c_.dt, sh, conv, accP, top, lt.md_hatinv, lt.md_hatmod, lt.md_mmod, s);
            c_.ntt(LimbBatch{conv, B * 2 * (ell - 1), nullptr, 0, ell - 1}, false);
            launch_moddown_rescale_finish(c_.dt, sh, o[0]->d, accQ, conv, lt.md_minv, s);
        }
        launch_ok("hoisted_dot_rows");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        if (pre) c_.pool.free(pre);
        if (top) c_.pool.free(top);
        c_.pool.free(conv);
        for (int b = 0; b < B; ++b) {
            o[b]->scale = x[idx[b]]->scale * sf;
            if (merged) o[b]->scale = o[b]->scale / (long double)c_.chain.q[ell - 1];
            out[idx[b]] = o[b];
            done[idx[b]] = 1;
        }
    }
    return out;
}

CtPtr Evaluator::rotate_each_sum(const std::vector<CtPtr>& vin, const std::vector<int>& indices) {
    if (vin.empty() || vin.size() != indices.size()) throw Error(FHELIN_ERR_ARG, "rotate_each_sum: one index per ciphertext");
    const int ns = vin[0]->slots > 0 ? vin[0]->slots : (1 << c_.prm.log_slots);
    // unrotated terms are plain additions; the rest go through the merged key switch in groups of <= 7
    std::vector<CtPtr> rot;
    std::vector<int> ridx;
    CtPtr acc;
    for (size_t i = 0; i < vin.size(); ++i) {
        if (indices[i] % ns == 0) acc = acc ? add(acc, vin[i]) : vin[i];
        else {
            rot.push_back(vin[i]);
            ridx.push_back(indices[i]);
        }
    }
    for (size_t first = 0; first < rot.size(); first += KsShape::MAX_ROT) {
        const int R = (int)std::min(rot.size() - first, (size_t)KsShape::MAX_ROT);
        std::vector<CtPtr> chunk(rot.begin() + first, rot.begin() + first + R);
        bool uniform = have_rotation_keys(std::vector<int>(ridx.begin() + first, ridx.begin() + first + R), ns) && c_.K >= 1;
        for (const CtPtr& c : chunk)
            uniform = uniform && c->npoly == 2 && c->ell == chunk[0]->ell && c->deg == chunk[0]->deg &&
                      fabsl(c->scale / chunk[0]->scale - 1.0L) < 1e-9L;
        if (!uniform || R < 2) {  // fall back to separate rotations
            for (int r = 0; r < R; ++r) {
                CtPtr t = rotate(chunk[r], ridx[first + r]);
                acc = acc ? add(acc, t) : t;
            }
            continue;
        }
        chunk = make_contiguous(chunk, 1);
        const size_t N = c_.N;
        const int K = c_.K, L1 = c_.L + 1, ell = chunk[0]->ell;
        const size_t pn = (size_t)ell * N, ctw = 2 * pn;
        const LevelTables& lt = c_.lvl[ell];
        const int nt = ell + K;
        hipStream_t s = c_.stream;
        const u64* base = chunk[0]->d;
        // ModUp of the R inputs as one batch
        KsShape up{ell, K, c_.alpha, lt.beta, L1, R, ctw, 0, 0, 0};
        u64* cc = c_.dalloc<u64>((size_t)R * ell * N);
        {
            LimbBatch ib{cc, R * ell, nullptr, 0, ell, base + pn};
            ib.src_group = ell;
            ib.src_group_stride = ctw;
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)R * lt.beta * nt * N);
        launch_modup_conv(c_.dt, up, ext, cc, base + pn, lt.up_hatinv, lt.up_hatmod, s);
        LimbBatch eb{ext, R * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, R * (lt.beta * nt - ell));
        // one accumulator for all R rotated inner products, one ModDown
        KsShape sh{ell, K, c_.alpha, lt.beta, L1, 1, 0, ctw, pn, 0};
        sh.n_rot = R;
        sh.rot_ext_stride = (size_t)lt.beta * nt * N;
        sh.rot_input_stride = ctw;
        for (int r = 0; r < R; ++r) {
            const u64 g = c_.galois_element(ridx[first + r]);
            sh.map_rot[r] = c_.automorph_map(g);
            sh.evk_rot[r] = permuted(*rot_keys.at(g), sh.map_rot[r]);
        }
        c_.stats.keyswitch += (u64)R;
        c_.stats.keyswitch_limbs += (u64)R * ell;
        u64* accQ = c_.dalloc<u64>((size_t)2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)2 * K * N);
        launch_ks_inner_multi(c_.dt, sh, accQ, accP, ext, base + pn, s);
        u64* c0sum = nullptr;
        if (c_.fuse_gather) {
            sh.gsrc = base;
            sh.gsrc_stride = 0;
        } else {
            c0sum = c_.dalloc<u64>((size_t)ell * N);
            launch_gather_sum(c_.dt, sh, c0sum, base, 0, s);
        }
        c_.ntt(LimbBatch{accP, 2 * K, nullptr, L1, K}, true);
        u64* conv = c_.dalloc<u64>((size_t)2 * ell * N);
        launch_moddown_conv(c_.dt, sh, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
        c_.ntt(LimbBatch{conv, 2 * ell, nullptr, 0, ell}, false);
        CtPtr o = new_ct(2, ell, chunk[0]->deg, chunk[0]->scale, chunk[0]->slots);
        launch_moddown_finish(c_.dt, sh, o->d, accQ, conv, c_.d_pinv, c0sum, nullptr, nullptr, nullptr, s);
        launch_ok("rotate_each_sum");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        if (c0sum) c_.pool.free(c0sum);
        c_.pool.free(conv);
        acc = acc ? add(acc, o) : o;
    }
    return acc;
}

std::vector<CtPtr> Evaluator::rotate_each_sum_rows(const std::vector<std::vector<CtPtr>>& rows, const std::vector<int>& indices) {
    if (rows.empty()) return {};
    const int ns = rows[0][0]->slots > 0 ? rows[0][0]->slots : (1 << c_.prm.log_slots);
    std::vector<int> rot_pos, plain_pos, ridx;
    for (size_t r = 0; r < indices.size(); ++r) {
        if (indices[r] % ns == 0) plain_pos.push_back((int)r);
        else {
            rot_pos.push_back((int)r);
            ridx.push_back(indices[r]);
        }
    }
    const int R = (int)rot_pos.size();
    bool uniform = R >= 2 && R <= KsShape::MAX_ROT && have_rotation_keys(ridx, ns) && c_.K >= 1;
    const CtPtr& f = rows[0][0];
    for (const auto& row : rows) {
        uniform = uniform && row.size() == indices.size();
        for (const CtPtr& c : row)
            uniform = uniform && c->npoly == 2 && c->ell == f->ell && c->deg == f->deg && fabsl(c->scale / f->scale - 1.0L) < 1e-9L;
    }
    std::vector<CtPtr> out(rows.size());
    if (!uniform) {
        for (size_t b = 0; b < rows.size(); ++b) out[b] = rotate_each_sum(rows[b], indices);
        return out;
    }
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1, ell = f->ell;
    const size_t pn = (size_t)ell * N, ctw = 2 * pn;
    const LevelTables& lt = c_.lvl[ell];
    const int nt = ell + K;
    hipStream_t s = c_.stream;
    const size_t chunk_rows = (size_t)std::max(1, batch_limit / 2);   // rows x R polynomials go through one ModUp
    for (size_t lo = 0; lo < rows.size(); lo += chunk_rows) {
        const size_t hi = std::min(rows.size(), lo + chunk_rows);
        const int B = (int)(hi - lo);
        std::vector<CtPtr> flat;
        for (size_t b = lo; b < hi; ++b)
            for (int r : rot_pos) flat.push_back(rows[b][r]);
        // the rotated terms where they stand when every row holds them consecutively and the rows are equally spaced (the groups of
        // a shift sum: seven rotated terms, then the next group's unrotated one): the transforms read them with a row stride;
        // a gather copy otherwise
        size_t row_stride = (size_t)R * ctw;
        bool regular = B >= 1 && flat[0]->block != nullptr;
        for (int b = 0; b < B && regular; ++b) {
            for (int r = 0; r < R && regular; ++r) {
                const CtPtr& c = flat[(size_t)b * R + r];
                regular = c->block == flat[0]->block && c->words() == (size_t)ctw &&
                          c->d == flat[(size_t)b * R]->d + (size_t)r * ctw;
            }
            if (regular && b == 1) {
                regular = flat[R]->d > flat[0]->d;
                if (regular) row_stride = (size_t)(flat[R]->d - flat[0]->d);
            }
            if (regular && b >= 1) regular = flat[(size_t)b * R]->d == flat[0]->d + (size_t)b * row_stride;
        }
        if (!regular) {
            flat = make_contiguous(flat, 2);
            row_stride = (size_t)R * ctw;
        }
        const u64* base = flat[0]->d;
        KsShape up{ell, K, c_.alpha, lt.beta, L1, B * R, ctw, 0, 0, 0};
        u64* cc = c_.dalloc<u64>((size_t)B * R * ell * N);
        {
            LimbBatch ib{cc, B * R * ell, nullptr, 0, ell, base + pn};
            ib.src_group = ell;
            ib.src_group_stride = ctw;
            if (row_stride != (size_t)R * ctw) {
                ib.src_group2 = R;
                ib.src_group2_stride = row_stride;
            }
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)B * R * lt.beta * nt * N);
        launch_modup_conv(c_.dt, up, ext, cc, base + pn, lt.up_hatinv, lt.up_hatmod, s);
        LimbBatch eb{ext, B * R * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, B * R * (lt.beta * nt - ell));
        KsShape sh{ell, K, c_.alpha, lt.beta, L1, B, row_stride, ctw, pn, 0};
        sh.n_rot = R;
        sh.rot_ext_stride = (size_t)lt.beta * nt * N;
        sh.rot_input_stride = ctw;
        sh.ext_batch_stride = (size_t)R * lt.beta * nt * N;
        for (int r = 0; r < R; ++r) {
            const u64 g = c_.galois_element(ridx[r]);
            sh.map_rot[r] = c_.automorph_map(g);
            sh.evk_rot[r] = permuted(*rot_keys.at(g), sh.map_rot[r]);
        }
        sh.gsrc = base;
        sh.gsrc_stride = row_stride;
        c_.stats.keyswitch += (u64)B * R;
        c_.stats.keyswitch_limbs += (u64)B * R * ell;
        u64* accQ = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)B * 2 * K * N);
        launch_ks_inner_multi(c_.dt, sh, accQ, accP, ext, base + pn, s);
        c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
        u64* conv = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        launch_moddown_conv(c_.dt, sh, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
        c_.ntt(LimbBatch{conv, B * 2 * ell, nullptr, 0, ell}, false);
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell, f->deg, f->scale, f->slots);
        launch_moddown_finish(c_.dt, sh, o[0]->d, accQ, conv, c_.d_pinv, nullptr, nullptr, nullptr, nullptr, s);
        launch_ok("rotate_each_sum_rows");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        c_.pool.free(conv);
        for (int b = 0; b < B; ++b) out[lo + b] = o[b];
    }
    // unrotated terms are plain additions, in the order rotate_each_sum adds them: plain terms first, then the rotated sum
    if (!plain_pos.empty()) {
        std::vector<CtPtr> acc(rows.size());
        for (size_t b = 0; b < rows.size(); ++b) acc[b] = rows[b][plain_pos[0]];
        for (size_t k = 1; k < plain_pos.size(); ++k) {
            std::vector<CtPtr> t(rows.size());
            for (size_t b = 0; b < rows.size(); ++b) t[b] = rows[b][plain_pos[k]];
            acc = add_batch(acc, t);
        }
        out = add_batch(acc, out);
    }
    return out;
}

// hoisted rotations of one ciphertext.  A rotation here is KeySwitch_{s -> sigma^-1(s)}(c1) + c0 followed by the NTT-domain
// automorphism gather in the ModDown epilogue, so the ModUp of c1 does not depend on the rotation index: it is computed
// once and every index runs only its own inner product + ModDown.  Bit-identical to rotate(a, i).
std::vector<CtPtr> Evaluator::rotate_many(const CtPtr& a, const std::vector<int>& indices) {
    if (a->npoly != 2) throw Error(FHELIN_ERR_STATE, "rotate: ciphertext must have 2 components");
    const int ns = a->slots > 0 ? a->slots : (1 << c_.prm.log_slots);
    std::vector<CtPtr> out(indices.size());
    std::vector<size_t> todo;
    for (size_t i = 0; i < indices.size(); ++i) {
        if (indices[i] % ns == 0) out[i] = a;
        else todo.push_back(i);
    }
    const size_t pn = (size_t)a->ell * c_.N, ctw = 2 * pn;
    for (size_t first = 0; first < todo.size(); first += KsShape::MAX_ROWS) {
        const int B = (int)std::min(todo.size() - first, (size_t)KsShape::MAX_ROWS);
        if (B == 1) {
            out[todo[first]] = rotate(a, indices[todo[first]]);
            continue;
        }
        KsRows rows;
        rows.shared_input = true;
        for (int b = 0; b < B; ++b) {
            const int index = indices[todo[first + b]];
            const u64 g = c_.galois_element(index);
            auto it = rot_keys.find(g);
            if (it == rot_keys.end())
                throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(index) + " (EvalRotateKeyGen list)");
            rows.keys.push_back(it->second.get());
            rows.maps.push_back(c_.automorph_map(g));
        }
        std::vector<CtPtr> o = new_ct_batch(B, 2, a->ell, a->deg, a->scale, a->slots);
        keyswitch_rows(rows, a->d + pn, 0, a->ell, o[0]->d, ctw, a->d, 0);
        for (int b = 0; b < B; ++b) out[todo[first + b]] = o[b];
    }
    return out;
}

std::vector<CtPtr> Evaluator::rotate_each(const std::vector<CtPtr>& vin, const std::vector<int>& indices) {
    if (vin.size() != indices.size()) throw Error(FHELIN_ERR_ARG, "rotate_each: one index per ciphertext");
    std::vector<CtPtr> out(vin.size());
    std::vector<char> done(vin.size(), 0);
    for (size_t first = 0; first < vin.size(); ++first) {
        if (done[first]) continue;
        const CtPtr& a = vin[first];
        if (a->npoly != 2) throw Error(FHELIN_ERR_STATE, "rotate: ciphertext must have 2 components");
        const int ns = a->slots > 0 ? a->slots : (1 << c_.prm.log_slots);
        if (indices[first] % ns == 0) {
            out[first] = clone(a);
            done[first] = 1;
            continue;
        }
        std::vector<size_t> idx;
        for (size_t i = first; i < vin.size() && (int)idx.size() < std::min(batch_limit, (int)KsShape::MAX_ROWS); ++i) {
            const CtPtr& b = vin[i];
            if (!done[i] && indices[i] % ns != 0 && b->npoly == 2 && b->ell == a->ell && b->deg == a->deg &&
                fabsl(b->scale / a->scale - 1.0L) < 1e-9L)
                idx.push_back(i);
        }
        if (idx.size() < 2) {
            out[first] = rotate(a, indices[first]);
            done[first] = 1;
            continue;
        }
        std::vector<CtPtr> chunk;
        KsRows rows;
        for (size_t i : idx) {
            chunk.push_back(vin[i]);
            const u64 g = c_.galois_element(indices[i]);
            auto it = rot_keys.find(g);
            if (it == rot_keys.end())
                throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(indices[i]) + " (EvalRotateKeyGen list)");
            rows.keys.push_back(it->second.get());
            rows.maps.push_back(c_.automorph_map(g));
        }
        chunk = make_contiguous(chunk, 3);
        const int B = (int)chunk.size();
        const size_t pn = (size_t)a->ell * c_.N, ctw = 2 * pn;
        std::vector<CtPtr> o = new_ct_batch(B, 2, a->ell, a->deg, a->scale, a->slots);
        keyswitch_rows(rows, chunk[0]->d + pn, ctw, a->ell, o[0]->d, ctw, chunk[0]->d, ctw);
        for (int b = 0; b < B; ++b) {
            o[b]->scale = vin[idx[b]]->scale;
            out[idx[b]] = o[b];
            done[idx[b]] = 1;
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ raw ops
// K5 steps 2+3: NTT of the centred lift of `last` ([P][N], coefficients modulo q_{ell-1}) into the ell-1 remaining limbs of
// each polynomial -> lifted [P][ell-1][N].  The lift rides in the load of the transform's first pass (LimbBatch::lift_qlm)
// when the dropped modulus is below twice every remaining one (x mod q_j is then one conditional subtraction); otherwise,
// or with FHELIN_FUSE_LIFT=0, the separate lift kernel runs first.  Same residues either way.
void Evaluator::lift_and_ntt(u64* lifted, const u64* last, int P, int ell) {
    const u64* qlm = c_.d_qlmod + (size_t)(ell - 1) * (c_.L + 1);
    bool fuse = c_.fuse_lift;
    for (int j = 0; j + 1 < ell && fuse; ++j) fuse = c_.chain.q[ell - 1] < 2 * c_.chain.q[j];
    if (fuse) {
        LimbBatch fb{lifted, P * (ell - 1), nullptr, 0, ell - 1, last};
        fb.lift_qlm = qlm;
        fb.lift_limb = ell - 1;
        c_.ntt(fb, false);
        return;
    }
    launch_rescale_lift(c_.dt, lifted, last, P, ell, qlm, c_.stream);
    c_.ntt(LimbBatch{lifted, P * (ell - 1), nullptr, 0, ell - 1}, false);
}

CtPtr Evaluator::raw_rescale(const CtPtr& a) {
    const int ell = a->ell, P = a->npoly;
    if (ell < 2) throw Error(FHELIN_ERR_STATE, "rescale: no limb left to drop");
    const size_t N = c_.N;
    hipStream_t s = c_.stream;
    u64* last = c_.dalloc<u64>((size_t)P * N);
    c_.stats.rescale += 1;
    c_.stats.rescale_limbs += (u64)ell;
    {
        // INTT of the last limb of every polynomial, read in place (stride = one polynomial), written densely
        LimbBatch lb{last, P, nullptr, ell - 1, 1, a->d + (size_t)(ell - 1) * N};
        lb.src_group = 1;
        lb.src_group_stride = (size_t)ell * N;
        c_.ntt(lb, true);
    }
    u64* lifted = c_.dalloc<u64>((size_t)P * (ell - 1) * N);
    lift_and_ntt(lifted, last, P, ell);
    CtPtr o = new_ct(P, ell - 1, a->deg, a->scale, a->slots);
    launch_rescale_finish(c_.dt, o->d, a->d, lifted, P, ell, c_.d_qlinv + (size_t)(ell - 1) * (c_.L + 1) * 2, s);
    launch_ok("rescale");
    c_.pool.free(last);
    c_.pool.free(lifted);
    return o;
}

CtPtr Evaluator::raw_rotate(const CtPtr& a, u64 g, const EvalKey& key, bool accumulate) {
    if (a->npoly != 2) throw Error(FHELIN_ERR_STATE, "rotate: ciphertext must have 2 components");
    const size_t pn = (size_t)a->ell * c_.N;
    CtPtr o = new_ct(2, a->ell, a->deg, a->scale, a->slots);
    // accumulate: out = a + rot(a) — the addition of the rotate-and-sum step rides in the ModDown epilogue
    keyswitch(a->d + pn, a->ell, key, o->d, a->d, nullptr, c_.automorph_map(g), accumulate ? a->d : nullptr);
    return o;
}

std::vector<CtPtr> Evaluator::rotate_batch(const std::vector<CtPtr>& vin, int index) { return rotate_batch_impl(vin, index, false); }
std::vector<CtPtr> Evaluator::rotate_add_batch(const std::vector<CtPtr>& vin, int index) { return rotate_batch_impl(vin, index, true); }

std::vector<CtPtr> Evaluator::rotate_batch_impl(const std::vector<CtPtr>& vin, int index, bool accumulate) {
    if (vin.empty()) return {};
    const int ns = vin[0]->slots > 0 ? vin[0]->slots : (1 << c_.prm.log_slots);
    std::vector<CtPtr> out(vin.size());
    if (vin.size() == 1 || index % ns == 0) {
        for (size_t i = 0; i < vin.size(); ++i) out[i] = accumulate ? rotate_add(vin[i], index) : rotate(vin[i], index);
        return out;
    }
    const u64 g = c_.galois_element(index);
    auto it = rot_keys.find(g);
    if (it == rot_keys.end())
        throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(index) + " (EvalRotateKeyGen list)");
    return rotate_galois_batch(vin, g, *it->second, accumulate);
}

std::vector<CtPtr> Evaluator::conjugate_batch(const std::vector<CtPtr>& v) {
    if (!conj_key) throw Error(FHELIN_ERR_KEY, "no conjugation key");
    return rotate_galois_batch(v, 2ull * c_.N - 1, *conj_key, false);
}

std::vector<CtPtr> Evaluator::rotate_galois_batch(const std::vector<CtPtr>& vin, u64 g, const EvalKey& key, bool accumulate) {
    std::vector<CtPtr> out(vin.size());
    if (vin.size() == 1) {
        out[0] = raw_rotate(vin[0], g, key, accumulate);
        return out;
    }
    const u32* map = c_.automorph_map(g);
    // rows that share (level, degree, scale) go through one batched key switch; others form their own groups
    std::vector<char> done(vin.size(), 0);
    for (size_t first = 0; first < vin.size(); ++first) {
        if (done[first]) continue;
        if (vin[first]->npoly != 2) throw Error(FHELIN_ERR_STATE, "rotate: ciphertext must have 2 components");
        std::vector<size_t> idx;
        for (size_t i = first; i < vin.size() && (int)idx.size() < batch_limit; ++i) {
            const CtPtr &a = vin[first], &b = vin[i];
            if (!done[i] && b->npoly == 2 && b->ell == a->ell && b->deg == a->deg && fabsl(b->scale / a->scale - 1.0L) < 1e-9L)
                idx.push_back(i);
        }
        std::vector<CtPtr> chunk;
        for (size_t i : idx) chunk.push_back(vin[i]);
        chunk = make_contiguous(chunk, 4);
        const int B = (int)chunk.size();
        const int ell = chunk[0]->ell;
        const size_t pn = (size_t)ell * c_.N, ctw = 2 * pn;
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell, chunk[0]->deg, chunk[0]->scale, chunk[0]->slots);
        const u64* base = chunk[0]->d;  // contiguous by construction (or a single ciphertext)
        keyswitch_batch(B, base + pn, ctw, ell, key, o[0]->d, ctw, base, nullptr, ctw, map, accumulate ? base : nullptr, ctw);
        for (int b = 0; b < B; ++b) {
            o[b]->scale = vin[idx[b]]->scale;
            out[idx[b]] = o[b];
            done[idx[b]] = 1;
        }
    }
    return out;
}

std::vector<std::vector<CtPtr>> Evaluator::rotate_many_batch(const std::vector<CtPtr>& xs, const std::vector<int>& indices) {
    std::vector<std::vector<CtPtr>> out(xs.size());
    if (xs.empty()) return out;
    bool uniform = xs.size() >= 2 && !c_.fuse_moddown && c_.K >= 1;
    for (const CtPtr& x : xs)
        uniform = uniform && x->npoly == 2 && x->ell == xs[0]->ell && x->deg == xs[0]->deg && x->slots == xs[0]->slots;
    if (!uniform) {
        for (size_t i = 0; i < xs.size(); ++i) out[i] = rotate_many(xs[i], indices);
        return out;
    }
    const int ns = xs[0]->slots > 0 ? xs[0]->slots : (1 << c_.prm.log_slots);
    std::vector<size_t> todo;
    for (size_t i = 0; i < xs.size(); ++i) out[i].resize(indices.size());
    for (size_t k = 0; k < indices.size(); ++k) {
        if (indices[k] % ns == 0)
            for (size_t i = 0; i < xs.size(); ++i) out[i][k] = xs[i];
        else
            todo.push_back(k);
    }
    const int R = (int)todo.size();
    if (R == 0) return out;
    std::vector<const EvalKey*> keys;
    std::vector<const u32*> maps;
    for (size_t k : todo) {
        const u64 g = c_.galois_element(indices[k]);
        auto it = rot_keys.find(g);
        if (it == rot_keys.end())
            throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(indices[k]) + " (EvalRotateKeyGen list)");
        keys.push_back(it->second.get());
        maps.push_back(c_.automorph_map(g));
    }
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1, ell = xs[0]->ell;
    const size_t pn = (size_t)ell * N, ctw = 2 * pn;
    const LevelTables& lt = c_.lvl[ell];
    const int nt = ell + K;
    hipStream_t s = c_.stream;
    // inputs per pass: the ModDown of a pass works on (inputs x R) rows
    const size_t per_pass = (size_t)std::max(1, 160 / R);
    for (size_t lo = 0; lo < xs.size(); lo += per_pass) {
        const size_t hi = std::min(xs.size(), lo + per_pass);
        const int B = (int)(hi - lo);
        std::vector<CtPtr> in = make_contiguous(std::vector<CtPtr>(xs.begin() + lo, xs.begin() + hi), 6);
        const u64* base = in[0]->d;
        const int rows = B * R;
        c_.stats.keyswitch += (u64)rows;
        c_.stats.keyswitch_limbs += (u64)rows * ell;
        // ModUp of the B inputs' c1, once
        KsShape up{ell, K, c_.alpha, lt.beta, L1, B, ctw, 0, 0, 0};
        u64* cc = c_.dalloc<u64>((size_t)B * ell * N);
        {
            LimbBatch ib{cc, B * ell, nullptr, 0, ell, base + pn};
            if (B > 1) {
                ib.src_group = ell;
                ib.src_group_stride = ctw;
            }
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)B * lt.beta * nt * N);
        launch_modup_conv(c_.dt, up, ext, cc, base + pn, lt.up_hatinv, lt.up_hatmod_r2, s);   // digits times 2^64: launch_ks_inner ends in redc128
        LimbBatch eb{ext, B * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, B * (lt.beta * nt - ell));
        // inner products: per input, rows of <= MAX_ROWS indices with their own keys, all reading that input's digits
        u64* accQ = c_.dalloc<u64>((size_t)rows * 2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)rows * 2 * K * N);
        // ONE launch per chunk of <= MAX_ROWS indices over ALL inputs (KsShape::row_mod): row (i, r) = rotation r of input i; in the
        // XCD-aware block order a key tile is fetched once for all inputs and a digit tile once for all indices
        auto row_shape = [&](int first, int cnt) {
            KsShape sh{ell, K, c_.alpha, lt.beta, L1, B * cnt, ctw, ctw, ctw, 0};
            sh.per_row = 1;
            sh.shared_input = 1;
            sh.row_mod = cnt;
            sh.ext_batch_stride = (size_t)lt.beta * nt * N;
            for (int b = 0; b < cnt; ++b) {
                sh.evk_row[b] = keys[first + b]->d;
                sh.map_row[b] = maps[first + b];
            }
            return sh;
        };
        const bool one_chunk = R <= (int)KsShape::MAX_ROWS;     // rows then lie [input][index] as the outputs do
        if (one_chunk) launch_ks_inner(c_.dt, row_shape(0, R), accQ, accP, ext, nullptr, base + pn, s);
        else
            for (int i = 0; i < B; ++i)
                for (int first = 0; first < R; first += KsShape::MAX_ROWS) {
                    const int cnt = std::min(R - first, (int)KsShape::MAX_ROWS);
                    const size_t row0 = (size_t)i * R + first;
                    KsShape sh1{ell, K, c_.alpha, lt.beta, L1, cnt, 0, ctw, 0, 0};
                    sh1.per_row = 1;
                    sh1.shared_input = 1;
                    for (int b = 0; b < cnt; ++b) {
                        sh1.evk_row[b] = keys[first + b]->d;
                        sh1.map_row[b] = maps[first + b];
                    }
                    launch_ks_inner(c_.dt, sh1, accQ + row0 * 2 * ell * N, accP + row0 * 2 * K * N, ext + (size_t)i * lt.beta * nt * N, nullptr,
                                    base + pn + (size_t)i * ctw, s);
                }
        // ONE ModDown over all rows
        KsShape dn{ell, K, c_.alpha, lt.beta, L1, rows, 0, ctw, 0, 0};
        c_.ntt(LimbBatch{accP, rows * 2 * K, nullptr, L1, K}, true);
        u64* conv = c_.dalloc<u64>((size_t)rows * 2 * ell * N);
        launch_moddown_conv(c_.dt, dn, conv, accP, c_.d_phatinv, c_.d_phatmod, s);
        c_.ntt(LimbBatch{conv, rows * 2 * ell, nullptr, 0, ell}, false);
        std::vector<CtPtr> o = new_ct_batch(rows, 2, ell, xs[lo]->deg, xs[lo]->scale, xs[lo]->slots);
        // the epilogue adds the input's c0 (gathered through each row's map)
        if (one_chunk) {
            launch_moddown_finish(c_.dt, row_shape(0, R), o[0]->d, accQ, conv, c_.d_pinv, base, nullptr, nullptr, nullptr, s);
        } else {
            for (int i = 0; i < B; ++i)
                for (int first = 0; first < R; first += KsShape::MAX_ROWS) {
                    const int cnt = std::min(R - first, (int)KsShape::MAX_ROWS);
                    const size_t row0 = (size_t)i * R + first;
                    KsShape sh1{ell, K, c_.alpha, lt.beta, L1, cnt, 0, ctw, 0, 0};
                    sh1.per_row = 1;
                    sh1.shared_input = 1;
                    for (int b = 0; b < cnt; ++b) {
                        sh1.evk_row[b] = keys[first + b]->d;
                        sh1.map_row[b] = maps[first + b];
                    }
                    // add_stride 0 = the same c0 for every row
                    launch_moddown_finish(c_.dt, sh1, o[row0]->d, accQ + row0 * 2 * ell * N, conv + row0 * 2 * ell * N, c_.d_pinv,
                                          base + (size_t)i * ctw, nullptr, nullptr, nullptr, s);
                }
        }
        launch_ok("rotate_many_batch");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        c_.pool.free(conv);
        for (int i = 0; i < B; ++i)
            for (int r = 0; r < R; ++r) {
                o[(size_t)i * R + r]->scale = xs[lo + i]->scale;
                out[lo + i][todo[r]] = o[(size_t)i * R + r];
            }
    }
    return out;
}

// rescale of many ciphertexts of identical shape as ONE polynomial batch [2B][ell][N] (K5 kernels are per polynomial)
std::vector<CtPtr> Evaluator::rescale_batch(const std::vector<CtPtr>& vin) {
    std::vector<CtPtr> out(vin.size());
    std::vector<char> done(vin.size(), 0);
    for (size_t first = 0; first < vin.size(); ++first) {
        if (done[first]) continue;
        std::vector<size_t> idx;
        for (size_t i = first; i < vin.size() && (int)idx.size() < batch_limit; ++i)
            if (!done[i] && vin[i]->npoly == 2 && vin[first]->npoly == 2 && vin[i]->ell == vin[first]->ell) idx.push_back(i);
        if (idx.size() < 2) {
            out[first] = rescale(vin[first]);
            done[first] = 1;
            continue;
        }
        std::vector<CtPtr> chunk;
        for (size_t i : idx) chunk.push_back(vin[i]);
        chunk = make_contiguous(chunk, 5);
        const int B = (int)chunk.size(), ell = chunk[0]->ell, P = 2 * B;
        if (ell < 2) throw Error(FHELIN_ERR_STATE, "rescale: no limb left to drop");
        const size_t N = c_.N;
        hipStream_t s = c_.stream;
        const u64* base = chunk[0]->d;
        u64* last = c_.dalloc<u64>((size_t)P * N);
        c_.stats.rescale += (u64)B;
        c_.stats.rescale_limbs += (u64)B * ell;
        LimbBatch lb{last, P, nullptr, ell - 1, 1, base + (size_t)(ell - 1) * N};
        lb.src_group = 1;
        lb.src_group_stride = (size_t)ell * N;
        c_.ntt(lb, true);
        u64* lifted = c_.dalloc<u64>((size_t)P * (ell - 1) * N);
        lift_and_ntt(lifted, last, P, ell);
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell - 1, 1, 0, chunk[0]->slots);
        launch_rescale_finish(c_.dt, o[0]->d, base, lifted, P, ell, c_.d_qlinv + (size_t)(ell - 1) * (c_.L + 1) * 2, s);
        launch_ok("rescale_batch");
        c_.pool.free(last);
        c_.pool.free(lifted);
        for (int b = 0; b < B; ++b) {
            const CtPtr& a = vin[idx[b]];
            o[b]->scale = a->scale / (long double)c_.chain.q[ell - 1];
            o[b]->deg = a->deg > 1 ? a->deg - 1 : 1;
            out[idx[b]] = o[b];
            done[idx[b]] = 1;
        }
    }
    return out;
}

std::vector<CtPtr> Evaluator::mult_plain_batch(const std::vector<CtPtr>& vin, const PtPtr& p) {
    return mult_plain_each(vin, std::vector<PtPtr>(vin.size(), p));
}

namespace {
// consecutive runs of operands with the same (components, limbs): one output block and one launch per <= 32 of them
template <class SameShape, class Emit>
void for_runs(size_t n, SameShape same, Emit emit) {
    size_t lo = 0;
    while (lo < n) {
        size_t hi = lo + 1;
        while (hi < n && hi - lo < (size_t)EwItems::MAX_ITEMS && same(lo, hi)) ++hi;
        emit(lo, hi);
        lo = hi;
    }
}
}  // namespace

std::vector<CtPtr> Evaluator::mult_plain_each(const std::vector<CtPtr>& vin, const std::vector<PtPtr>& p) {
    if (vin.size() != p.size()) throw Error(FHELIN_ERR_ARG, "mult_plain_each: one plaintext per ciphertext");
    if (vin.empty()) return {};
    std::vector<CtPtr> x = vin;
    {
        // degree-2 operands are rescaled first (as in mult_plain); the same ciphertext may occur many times (one
        // container masked 128 ways): rescale every distinct one once
        std::vector<CtPtr> need;
        std::map<const Ciphertext*, size_t> slot;
        for (size_t i = 0; i < vin.size(); ++i)
            if (vin[i]->deg >= 2 && !slot.count(vin[i].get())) {
                slot[vin[i].get()] = need.size();
                need.push_back(vin[i]);
            }
        if (!need.empty()) {
            std::vector<CtPtr> r = rescale_batch(need);
            for (size_t i = 0; i < vin.size(); ++i)
                if (vin[i]->deg >= 2) x[i] = r[slot[vin[i].get()]];
        }
    }
    std::vector<CtPtr> out(x.size());
    // operands of one shape: ONE output block for all products, so that any sub-range a later batched key switch takes is
    // contiguous as it stands (no gather copies); the launches still go in runs of MAX_ITEMS
    bool one_shape = x.size() > 1;
    for (const CtPtr& c : x) one_shape = one_shape && c->npoly == x[0]->npoly && c->ell == x[0]->ell && c->deg == x[0]->deg;
    std::vector<CtPtr> all;
    if (one_shape) all = new_ct_batch((int)x.size(), x[0]->npoly, x[0]->ell, x[0]->deg + 1, x[0]->scale, x[0]->slots);
    for_runs(x.size(), [&](size_t a, size_t b) { return x[a]->npoly == x[b]->npoly && x[a]->ell == x[b]->ell && x[a]->deg == x[b]->deg; },
             [&](size_t lo, size_t hi) {
                 const CtPtr& f = x[lo];
                 std::vector<CtPtr> o = one_shape ? std::vector<CtPtr>(all.begin() + lo, all.begin() + hi)
                                                  : new_ct_batch((int)(hi - lo), f->npoly, f->ell, f->deg + 1, f->scale, f->slots);
                 EwItems it;
                 it.n = (int)(hi - lo);
                 it.vecs = f->npoly * f->ell;
                 it.b_vecs = f->ell;
                 for (size_t i = lo; i < hi; ++i) {
                     auto enc = p[i]->at(x[i]->ell, c_.sf_real[x[i]->level()]);
                     o[i - lo]->scale = x[i]->scale * enc->scale;
                     it.out[i - lo] = o[i - lo]->d;
                     it.a[i - lo] = x[i]->d;
                     it.b[i - lo] = enc->d;
                     out[i] = o[i - lo];
                 }
                 launch_ew_items(c_.dt, it, 0, f->ell, c_.stream);
                 c_.stats.ct_pt_mult += (u64)(hi - lo);
                 c_.stats.ct_pt_limbs += (u64)(hi - lo) * f->ell;
             });
    launch_ok("mult_plain_each");
    return out;
}

CtPtr Evaluator::dot_plain(const std::vector<CtPtr>& vin, const std::vector<PtPtr>& p, long double pt_scale, const CtPtr& dest) {
    if (vin.size() != p.size() || vin.empty()) throw Error(FHELIN_ERR_ARG, "dot_plain: one plaintext per ciphertext, at least one term");
    // degree-2 operands are rescaled first, every distinct ciphertext once (as mult_plain does)
    std::vector<CtPtr> x = vin;
    {
        std::vector<CtPtr> need;
        std::map<const Ciphertext*, size_t> slot;
        for (size_t i = 0; i < vin.size(); ++i)
            if (vin[i]->deg >= 2 && !slot.count(vin[i].get())) {
                slot[vin[i].get()] = need.size();
                need.push_back(vin[i]);
            }
        if (!need.empty()) {
            std::vector<CtPtr> r = rescale_batch(need);
            for (size_t i = 0; i < vin.size(); ++i)
                if (vin[i]->deg >= 2) x[i] = r[slot[vin[i].get()]];
        }
    }
    bool uniform = x.size() >= 2;
    for (const CtPtr& c : x)
        uniform = uniform && c->npoly == x[0]->npoly && c->ell == x[0]->ell && c->deg == x[0]->deg && fabsl(c->scale / x[0]->scale - 1.0L) < 1e-9L;
    if (!uniform) {
        if (pt_scale > 0) throw Error(FHELIN_ERR_ARG, "dot_plain: an explicit plaintext scale needs uniform operands");
        std::vector<CtPtr> prod = mult_plain_each(x, p);
        CtPtr acc = prod[0];
        for (size_t i = 1; i < prod.size(); ++i) acc = add(acc, prod[i]);
        return acc;
    }
    const CtPtr& f = x[0];
    const long double sf = pt_scale > 0 ? pt_scale : c_.sf_real[f->level()];
    CtPtr acc;
    for (size_t lo = 0; lo < x.size(); lo += EwItems::MAX_ITEMS) {
        const size_t hi = std::min(x.size(), lo + (size_t)EwItems::MAX_ITEMS);
        EwItems it;
        it.n = (int)(hi - lo);
        it.vecs = f->npoly * f->ell;
        it.b_vecs = f->ell;
        for (size_t i = lo; i < hi; ++i) {
            it.a[i - lo] = x[i]->d;
            it.b[i - lo] = p[i]->at(f->ell, sf)->d;
        }
        // a caller that will hand several such sums to one batched key switch provides their common block (dest)
        const bool into_dest = dest && x.size() <= (size_t)EwItems::MAX_ITEMS && dest->npoly == f->npoly && dest->ell == f->ell;
        CtPtr o = into_dest ? dest : new_ct(f->npoly, f->ell, f->deg + 1, f->scale * sf, f->slots);
        if (into_dest) {
            o->deg = f->deg + 1;
            o->scale = f->scale * sf;
            o->slots = f->slots;
        }
        launch_ew_dot(c_.dt, o->d, it, f->ell, c_.stream);
        c_.stats.ct_pt_mult += (u64)(hi - lo);
        c_.stats.ct_pt_limbs += (u64)(hi - lo) * f->ell;
        acc = acc ? add(acc, o) : o;
    }
    launch_ok("dot_plain");
    return acc;
}

bool Evaluator::dot_plain_groups(const std::vector<CtPtr>& cts, const std::vector<std::vector<PtPtr>>& pts, long double pt_scale,
                                 const std::vector<CtPtr>& dest) {
    if (!dot_groups || cts.empty() || cts.size() > (size_t)EwDotGroups::MAX_A || pts.empty() || pts.size() > (size_t)EwDotGroups::MAX_G ||
        dest.size() != pts.size())
        return false;
    const CtPtr& f = cts[0];
    for (const CtPtr& c : cts)
        if (c->npoly != 2 || c->deg != 1 || c->ell != f->ell || fabsl(c->scale / f->scale - 1.0L) > 1e-9L) return false;
    for (const CtPtr& o : dest)
        if (!o || o->npoly != 2 || o->ell != f->ell) return false;
    const long double sf = pt_scale > 0 ? pt_scale : c_.sf_real[f->level()];
    EwDotGroups d;
    d.na = (int)cts.size();
    d.ng = (int)pts.size();
    d.ell = f->ell;
    for (int b = 0; b < d.na; ++b) d.a[b] = cts[b]->d;
    u64 terms = 0;
    for (int g = 0; g < d.ng; ++g) {
        if (pts[g].size() != cts.size()) return false;
        for (int b = 0; b < d.na; ++b) {
            d.p[g][b] = pts[g][b] ? pts[g][b]->at(f->ell, sf)->d : nullptr;
            terms += pts[g][b] ? 1 : 0;
        }
        d.out[g] = dest[g]->d;
        dest[g]->deg = f->deg + 1;
        dest[g]->scale = f->scale * sf;
        dest[g]->slots = f->slots;
    }
    launch_ew_dot_groups(c_.dt, d, c_.stream);
    launch_ok("dot_plain_groups");
    c_.stats.ct_pt_mult += terms;
    c_.stats.ct_pt_limbs += terms * (u64)f->ell;
    return true;
}

bool Evaluator::dot_plain_groups_batch(const std::vector<std::vector<CtPtr>>& cts, const std::vector<std::vector<PtPtr>>& pts, long double pt_scale,
                                       const std::vector<std::vector<CtPtr>>& dest) {
    const size_t nb = cts.size();
    if (!dot_groups || nb < 2 || dest.size() != nb || cts[0].empty() || cts[0].size() > (size_t)EwDotGroups::MAX_A || pts.empty() ||
        pts.size() > (size_t)EwDotGroups::MAX_G)
        return false;
    const size_t na = cts[0].size(), ng = pts.size();
    const CtPtr& f = cts[0][0];
    for (size_t x = 0; x < nb; ++x) {
        if (cts[x].size() != na || dest[x].size() != ng) return false;
        for (const CtPtr& c : cts[x])
            if (c->npoly != 2 || c->deg != 1 || c->ell != f->ell || fabsl(c->scale / f->scale - 1.0L) > 1e-9L) return false;
        for (const CtPtr& o : dest[x])
            if (!o || o->npoly != 2 || o->ell != f->ell) return false;
    }
    EwDotGroups d;
    d.na = (int)na;
    d.ng = (int)ng;
    d.ell = f->ell;
    d.nbatch = (int)nb;
    // equally spaced over the batch, ascending addresses
    for (size_t b = 0; b < na; ++b) {
        if (cts[1][b]->d <= cts[0][b]->d) return false;
        const size_t st = (size_t)(cts[1][b]->d - cts[0][b]->d);
        for (size_t x = 0; x < nb; ++x)
            if (cts[x][b]->d != cts[0][b]->d + x * st) return false;
        d.a[b] = cts[0][b]->d;
        d.a_stride[b] = st;
    }
    for (size_t g = 0; g < ng; ++g) {
        if (dest[1][g]->d <= dest[0][g]->d) return false;
        const size_t st = (size_t)(dest[1][g]->d - dest[0][g]->d);
        for (size_t x = 0; x < nb; ++x)
            if (dest[x][g]->d != dest[0][g]->d + x * st) return false;
        d.out[g] = dest[0][g]->d;
        d.out_stride[g] = st;
    }
    const long double sf = pt_scale > 0 ? pt_scale : c_.sf_real[f->level()];
    u64 terms = 0;
    std::vector<std::shared_ptr<Encoding>> hold;
    for (size_t g = 0; g < ng; ++g) {
        if (pts[g].size() != na) return false;
        for (size_t b = 0; b < na; ++b) {
            if (pts[g][b]) {
                hold.push_back(pts[g][b]->at(f->ell, sf));
                d.p[g][b] = hold.back()->d;
                ++terms;
            } else {
                d.p[g][b] = nullptr;
            }
        }
        for (size_t x = 0; x < nb; ++x) {
            dest[x][g]->deg = f->deg + 1;
            dest[x][g]->scale = cts[x][0]->scale * sf;
            dest[x][g]->slots = f->slots;
        }
    }
    launch_ew_dot_groups(c_.dt, d, c_.stream);
    launch_ok("dot_plain_groups_batch");
    c_.stats.ct_pt_mult += terms * nb;
    c_.stats.ct_pt_limbs += terms * nb * (u64)f->ell;
    return true;
}

bool Evaluator::dot_plain_cyclic(const std::vector<CtPtr>& cts, const std::vector<PtPtr>& pts, const std::vector<CtPtr>& dest) {
    constexpr int P = EwCyclic::PERIOD;
    if (cts.empty() || (int)cts.size() > P || (int)pts.size() != P || (int)dest.size() != P) return false;
    const CtPtr& f = cts[0];
    for (const CtPtr& c : cts)
        if (c->npoly != 2 || c->deg != 1 || c->ell != f->ell || fabsl(c->scale / f->scale - 1.0L) > 1e-9L) return false;
    for (const CtPtr& o : dest)
        if (!o || o->npoly != 2 || o->ell != f->ell) return false;
    const long double sf = c_.sf_real[f->level()];
    EwCyclic d;
    d.n = (int)cts.size();
    d.ell = f->ell;
    std::vector<std::shared_ptr<Encoding>> hold;
    for (int j = 0; j < P; ++j) {
        hold.push_back(pts[j]->at(f->ell, sf));
        d.m[j] = hold.back()->d;
        d.a[j] = j < d.n ? cts[j]->d : nullptr;
        d.out[j] = dest[j]->d;
        dest[j]->deg = f->deg + 1;
        dest[j]->scale = f->scale * sf;
        dest[j]->slots = f->slots;
    }
    launch_ew_cyclic_dot(c_.dt, d, c_.stream);
    launch_ok("dot_plain_cyclic");
    c_.stats.ct_pt_mult += (u64)P * d.n;
    c_.stats.ct_pt_limbs += (u64)P * d.n * (u64)f->ell;
    return true;
}

bool Evaluator::dot_plain_window(const std::vector<CtPtr>& cur, const std::vector<CtPtr>& prev, const std::vector<PtPtr>& pts,
                                 const std::vector<CtPtr>& dest, bool accumulate) {
    constexpr int W = EwWindow::W;
    if ((int)cur.size() != W || (int)prev.size() != W || (int)pts.size() != W || (int)dest.size() != W) return false;
    CtPtr f;
    for (const auto* v : {&cur, &prev})
        for (const CtPtr& c : *v)
            if (c && !f) f = c;
    if (!f) return false;
    u64 terms = 0;
    for (const auto* v : {&cur, &prev})
        for (const CtPtr& c : *v)
            if (c) {
                if (c->npoly != 2 || c->deg != 1 || c->ell != f->ell || fabsl(c->scale / f->scale - 1.0L) > 1e-9L) return false;
                ++terms;
            }
    for (const CtPtr& o : dest)
        if (!o || o->npoly != 2 || o->ell != f->ell) return false;
    const long double sf = c_.sf_real[f->level()];
    EwWindow d;
    d.ell = f->ell;
    d.accumulate = accumulate ? 1 : 0;
    std::vector<std::shared_ptr<Encoding>> hold;
    for (int j = 0; j < W; ++j) {
        hold.push_back(pts[j]->at(f->ell, sf));
        d.m[j] = hold.back()->d;
        d.cur[j] = cur[j] ? cur[j]->d : dest[0]->d;      // absent: any readable block of the operands' shape, dropped by the mask
        d.prev[j] = prev[j] ? prev[j]->d : dest[0]->d;
        if (cur[j]) d.cur_mask |= 1u << j;
        if (prev[j]) d.prev_mask |= 1u << j;
        d.out[j] = dest[j]->d;
        dest[j]->deg = f->deg + 1;
        dest[j]->scale = f->scale * sf;
        dest[j]->slots = f->slots;
    }
    launch_ew_window_dot(c_.dt, d, c_.stream);
    launch_ok("dot_plain_window");
    c_.stats.ct_pt_mult += terms * (u64)W / 2;          // every operand meets half of the block's plaintexts on average
    c_.stats.ct_pt_limbs += terms * (u64)W / 2 * (u64)f->ell;
    return true;
}

std::vector<CtPtr> Evaluator::add_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b) { return add_sub_batch(a, b, 1); }
std::vector<CtPtr> Evaluator::sub_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b) { return add_sub_batch(a, b, 2); }

// FLEXIBLEAUTO alignment of many operand pairs at once: x[i], y[i] = what match(a[i], b[i]) gives.  The common case of a row loop or
// of a round of a polynomial evaluation - a degree-1 operand with more limbs meets a degree-1 operand with fewer - goes through ONE
// batched integer multiply + rescale per (target limbs, target scale) instead of one per pair (adjust_deg1_batch: the residues of
// adjust(); the same ciphertext brought to the same target for several pairs is adjusted once); everything else through match().
void Evaluator::match_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, std::vector<CtPtr>& x, std::vector<CtPtr>& y) {
    const size_t n = a.size();
    x.assign(n, CtPtr());
    y.assign(n, CtPtr());
    std::vector<char> aligned(n, 0);
    for (int side = 0; side < 2; ++side) {
        // side 0: b is brought down to a; side 1: a is brought down to b
        std::map<std::pair<int, long long>, std::vector<size_t>> groups;   // (target ell, bits of the target scale) -> pairs
        for (size_t i = 0; i < n; ++i) {
            if (aligned[i]) continue;
            const CtPtr& lo = side == 0 ? a[i] : b[i];
            const CtPtr& hi = side == 0 ? b[i] : a[i];
            if (lo->deg == 1 && hi->deg == 1 && lo->ell < hi->ell) {
                long long bits;
                const double sd = (double)lo->scale;
                std::memcpy(&bits, &sd, sizeof bits);
                groups[{lo->ell, bits}].push_back(i);
            }
        }
        for (auto& g : groups) {
            if (g.second.size() < 2) continue;
            // the 80-bit scales of a group must agree exactly (they do for rows of one call); else leave the pairs to match()
            const long double sc = (side == 0 ? a[g.second[0]] : b[g.second[0]])->scale;
            bool same = true;
            for (size_t i : g.second) same = same && (side == 0 ? a[i] : b[i])->scale == sc;
            if (!same) continue;
            std::vector<CtPtr> hi;
            std::map<const Ciphertext*, size_t> slot;                       // distinct ciphertexts to adjust
            std::vector<size_t> which(g.second.size());
            for (size_t k = 0; k < g.second.size(); ++k) {
                const CtPtr& h = side == 0 ? b[g.second[k]] : a[g.second[k]];
                auto it = slot.find(h.get());
                if (it == slot.end()) {
                    it = slot.emplace(h.get(), hi.size()).first;
                    hi.push_back(h);
                }
                which[k] = it->second;
            }
            std::vector<CtPtr> adj = adjust_deg1_batch(hi, g.first.first, sc);
            for (size_t k = 0; k < g.second.size(); ++k) {
                const size_t i = g.second[k];
                x[i] = side == 0 ? a[i] : adj[which[k]];
                y[i] = side == 0 ? adj[which[k]] : b[i];
                aligned[i] = 1;
            }
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (!aligned[i]) match(a[i], b[i], x[i], y[i]);
}

std::vector<CtPtr> Evaluator::add_sub_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, int op) {
    if (a.size() != b.size()) throw Error(FHELIN_ERR_ARG, "add_batch: operand count mismatch");
    std::vector<CtPtr> x, y, out(a.size());
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i]->npoly != b[i]->npoly) throw Error(FHELIN_ERR_STATE, "add: component count mismatch");
    match_batch(a, b, x, y);   // the driver's residual additions output[i] + inputs[i] (src/main.cpp:237-239): one batched adjustment
    for_runs(x.size(), [&](size_t p, size_t q) { return x[p]->npoly == x[q]->npoly && x[p]->ell == x[q]->ell; },
             [&](size_t lo, size_t hi) {
                 const CtPtr& f = x[lo];
                 std::vector<CtPtr> o = new_ct_batch((int)(hi - lo), f->npoly, f->ell, f->deg, f->scale, f->slots);
                 EwItems it;
                 it.n = (int)(hi - lo);
                 it.vecs = it.b_vecs = f->npoly * f->ell;
                 for (size_t i = lo; i < hi; ++i) {
                     o[i - lo]->deg = x[i]->deg;
                     o[i - lo]->scale = x[i]->scale;
                     it.out[i - lo] = o[i - lo]->d;
                     it.a[i - lo] = x[i]->d;
                     it.b[i - lo] = y[i]->d;
                     out[i] = o[i - lo];
                 }
                 launch_ew_items(c_.dt, it, op, f->ell, c_.stream);
             });
    launch_ok("add_batch");
    return out;
}

std::vector<CtPtr> Evaluator::mult_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b) {
    if (a.size() != b.size()) throw Error(FHELIN_ERR_ARG, "mult_batch: operand count mismatch");
    if (!relin_key) throw Error(FHELIN_ERR_KEY, "no relinearisation key (EvalMultKeyGen not called)");
    const size_t n = a.size();
    // operands of degree 2 are rescaled first (every distinct ciphertext once), then brought to a common level per pair
    std::vector<CtPtr> in;
    std::map<const Ciphertext*, size_t> slot;
    for (const auto* side : {&a, &b})
        for (const CtPtr& c : *side) {
            if (c->npoly != 2) throw Error(FHELIN_ERR_STATE, "mult: operands must have 2 components");
            if (c->deg >= 2 && !slot.count(c.get())) {
                slot[c.get()] = in.size();
                in.push_back(c);
            }
        }
    std::vector<CtPtr> resc = in.empty() ? std::vector<CtPtr>() : rescale_batch(in);
    auto ready = [&](const CtPtr& c) { return c->deg >= 2 ? resc[slot[c.get()]] : c; };
    std::vector<CtPtr> x, y, out(n), ra(n), rb(n);
    for (size_t i = 0; i < n; ++i) {
        ra[i] = ready(a[i]);
        rb[i] = ready(b[i]);
    }
    match_batch(ra, rb, x, y);   // the level adjustments of all pairs of a round together (one batched rescale per target level)
    std::vector<char> done(n, 0);
    for (size_t first = 0; first < n; ++first) {
        if (done[first]) continue;
        std::vector<size_t> idx;
        for (size_t i = first; i < n && (int)idx.size() < batch_limit; ++i)
            if (!done[i] && x[i]->ell == x[first]->ell) idx.push_back(i);
        const int B = (int)idx.size(), ell = x[first]->ell;
        const size_t pn = (size_t)ell * c_.N;
        std::vector<CtPtr> d = new_ct_batch(B, 3, ell, 2, 0, x[first]->slots);   // tensor products, contiguous [B][3][ell][N]
        for (int k0 = 0; k0 < B; k0 += EwItems::MAX_ITEMS) {   // all tensor products of the batch in one launch per 32 pairs
            EwItems it;
            it.n = std::min(B - k0, (int)EwItems::MAX_ITEMS);
            for (int k = 0; k < it.n; ++k) {
                it.out[k] = d[k0 + k]->d;
                it.a[k] = x[idx[k0 + k]]->d;
                it.b[k] = y[idx[k0 + k]]->d;
            }
            launch_tensor_items(c_.dt, it, ell, c_.stream);
        }
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell, 2, 0, x[first]->slots);
        keyswitch_batch(B, d[0]->d + 2 * pn, 3 * pn, ell, *relin_key, o[0]->d, 2 * pn, d[0]->d, d[0]->d + pn, 3 * pn, nullptr, nullptr, 0);
        for (int k = 0; k < B; ++k) {
            const size_t i = idx[k];
            o[k]->deg = x[i]->deg + y[i]->deg;
            o[k]->scale = x[i]->scale * y[i]->scale;
            out[i] = o[k];
            done[i] = 1;
        }
    }
    launch_ok("mult_batch");
    return out;
}

std::vector<CtPtr> Evaluator::mult_affine_rescale_batch(const std::vector<CtPtr>& a, const std::vector<CtPtr>& b, int f, double cadd,
                                                        const std::vector<CtPtr>& sub) {
    if (a.size() != b.size() || (!sub.empty() && sub.size() != a.size())) throw Error(FHELIN_ERR_ARG, "mult_affine_rescale_batch: operand count mismatch");
    if (f != 1 && f != 2) throw Error(FHELIN_ERR_ARG, "mult_affine_rescale_batch: factor 1 or 2");
    if (!relin_key) throw Error(FHELIN_ERR_KEY, "no relinearisation key (EvalMultKeyGen not called)");
    if (c_.K < 1 || c_.K + 1 > 16) throw Error(FHELIN_ERR_STATE, "mult_affine_rescale_batch: 1..15 special primes");
    const size_t n = a.size();
    // operands exactly as mult_batch takes them: degree 2 rescaled first (every distinct ciphertext once), pairs brought to one level
    std::vector<CtPtr> in;
    std::map<const Ciphertext*, size_t> slot;
    for (const auto* side : {&a, &b})
        for (const CtPtr& c : *side) {
            if (c->npoly != 2) throw Error(FHELIN_ERR_STATE, "mult: operands must have 2 components");
            if (c->deg >= 2 && !slot.count(c.get())) {
                slot[c.get()] = in.size();
                in.push_back(c);
            }
        }
    std::vector<CtPtr> resc = in.empty() ? std::vector<CtPtr>() : rescale_batch(in);
    auto ready = [&](const CtPtr& c) { return c->deg >= 2 ? resc[slot[c.get()]] : c; };
    std::vector<CtPtr> x(n), y(n), out(n);
    for (size_t i = 0; i < n; ++i) match(ready(a[i]), ready(b[i]), x[i], y[i]);
    const size_t N = c_.N;
    const int K = c_.K, L1 = c_.L + 1;
    hipStream_t s = c_.stream;
    std::vector<char> done(n, 0);
    for (size_t first = 0; first < n; ++first) {
        if (done[first]) continue;
        std::vector<size_t> idx;
        for (size_t i = first; i < n && (int)idx.size() < batch_limit; ++i)
            if (!done[i] && x[i]->ell == x[first]->ell && fabsl(x[i]->scale * y[i]->scale / (x[first]->scale * y[first]->scale) - 1.0L) < 1e-12L)
                idx.push_back(i);
        const int B = (int)idx.size(), ell = x[first]->ell;
        if (ell < 2) throw Error(FHELIN_ERR_STATE, "mult_affine_rescale_batch: no limb left to drop");
        const size_t pn = (size_t)ell * N;
        const long double sc = x[first]->scale * y[first]->scale;
        const LevelTables& lt = c_.lvl[ell];
        const int nt = ell + K;
        std::vector<CtPtr> d = new_ct_batch(B, 3, ell, 2, 0, x[first]->slots);   // tensor products, contiguous [B][3][ell][N]
        for (int k0 = 0; k0 < B; k0 += EwItems::MAX_ITEMS) {
            EwItems it;
            it.n = std::min(B - k0, (int)EwItems::MAX_ITEMS);
            for (int k = 0; k < it.n; ++k) {
                it.out[k] = d[k0 + k]->d;
                it.a[k] = x[idx[k0 + k]]->d;
                it.b[k] = y[idx[k0 + k]]->d;
            }
            launch_tensor_items(c_.dt, it, ell, s);
        }
        // the subtrahends at the products' (limbs, degree 2, scale), in one block
        std::vector<CtPtr> sadj;
        if (!sub.empty()) {
            for (int k = 0; k < B; ++k) sadj.push_back(adjust(sub[idx[k]], ell, 2, sc));
            sadj = make_contiguous(sadj, 7);
        }
        ScalarSet cst;
        if (cadd != 0.0) real_to_scalars(c_, (long double)cadd * sc, ell, cst);
        // key switch of d2 up to the accumulator
        KsShape sh{ell, K, c_.alpha, lt.beta, L1, B, 3 * pn, (size_t)2 * (ell - 1) * N, pn, 0};
        c_.stats.keyswitch += (u64)B;
        c_.stats.keyswitch_limbs += (u64)B * ell;
        c_.stats.rescale += (u64)B;
        c_.stats.rescale_limbs += (u64)B * ell;
        const u64* c_ntt = d[0]->d + 2 * pn;
        u64* cc = c_.dalloc<u64>((size_t)B * ell * N);
        {
            LimbBatch ib{cc, B * ell, nullptr, 0, ell, c_ntt};
            if (B > 1) {
                ib.src_group = ell;
                ib.src_group_stride = 3 * pn;
            }
            c_.ntt(ib, true);
        }
        u64* ext = c_.dalloc<u64>((size_t)B * lt.beta * nt * N);
        launch_modup_conv(c_.dt, sh, ext, cc, c_ntt, lt.up_hatinv, lt.up_hatmod_r2, s);   // digits times 2^64: launch_ks_inner ends in redc128
        LimbBatch eb{ext, B * lt.beta * nt, lt.ext_limb_tab, 0, 1};
        eb.tab_len = lt.beta * nt;
        eb.lazy_out = true;
        c_.ntt(eb, false, B * (lt.beta * nt - ell));
        u64* accQ = c_.dalloc<u64>((size_t)B * 2 * ell * N);
        u64* accP = c_.dalloc<u64>((size_t)B * 2 * K * N);
        launch_ks_inner(c_.dt, sh, accQ, accP, ext, relin_key->d, c_ntt, s);
        // X_Q = f acc_Q + P (f (d0, d1) + constant - subtrahend),  X_P = f acc_P
        launch_affine_acc(c_.dt, sh, accQ, d[0]->d, sadj.empty() ? nullptr : sadj[0]->d, cst, cadd != 0.0 ? 1 : 0, f, c_.d_pmod, s);
        if (f == 2) launch_ew_add(c_.dt, accP, accP, accP, B * 2 * K, B * 2 * K, L1, K, s);
        // P and the top limb dropped together
        c_.ntt(LimbBatch{accP, B * 2 * K, nullptr, L1, K}, true);
        u64* top = c_.dalloc<u64>((size_t)B * 2 * N);
        {
            LimbBatch tb{top, B * 2, nullptr, ell - 1, 1, accQ + (size_t)(ell - 1) * N};
            tb.src_group = 1;
            tb.src_group_stride = pn;
            c_.ntt(tb, true);
        }
        u64* conv = c_.dalloc<u64>((size_t)B * 2 * (ell - 1) * N);
        launch_moddown_rescale_conv(c_.dt, sh, conv, accP, top, lt.md_hatinv, lt.md_hatmod, lt.md_mmod, s);
        c_.ntt(LimbBatch{conv, B * 2 * (ell - 1), nullptr, 0, ell - 1}, false);
        std::vector<CtPtr> o = new_ct_batch(B, 2, ell - 1, 1, 0, x[first]->slots);
        launch_moddown_rescale_finish(c_.dt, sh, o[0]->d, accQ, conv, lt.md_minv, s);
        launch_ok("mult_affine_rescale_batch");
        c_.pool.free(cc);
        c_.pool.free(ext);
        c_.pool.free(accQ);
        c_.pool.free(accP);
        c_.pool.free(top);
        c_.pool.free(conv);
        for (int k = 0; k < B; ++k) {
            const size_t i = idx[k];
            o[k]->scale = x[i]->scale * y[i]->scale / (long double)c_.chain.q[ell - 1];
            out[i] = o[k];
            done[i] = 1;
        }
    }
    return out;
}

std::vector<CtPtr> Evaluator::add_plain_batch(const std::vector<CtPtr>& v, const PtPtr& p) {
    std::vector<CtPtr> out(v.size());
    for_runs(v.size(), [&](size_t a, size_t b) { return v[a]->npoly == v[b]->npoly && v[a]->ell == v[b]->ell; },
             [&](size_t lo, size_t hi) {
                 const CtPtr& f = v[lo];
                 std::vector<CtPtr> o = new_ct_batch((int)(hi - lo), f->npoly, f->ell, f->deg, f->scale, f->slots);
                 EwItems it;
                 it.n = (int)(hi - lo);
                 it.vecs = f->npoly * f->ell;
                 it.b_vecs = f->ell;
                 for (size_t i = lo; i < hi; ++i) {
                     auto enc = p->at(v[i]->ell, v[i]->scale);
                     o[i - lo]->deg = v[i]->deg;
                     o[i - lo]->scale = v[i]->scale;
                     it.out[i - lo] = o[i - lo]->d;
                     it.a[i - lo] = v[i]->d;
                     it.b[i - lo] = enc->d;
                     out[i] = o[i - lo];
                 }
                 launch_ew_items(c_.dt, it, 3, f->ell, c_.stream);
             });
    launch_ok("add_plain_batch");
    return out;
}

CtPtr Evaluator::rotate_add(const CtPtr& a, int index) {
    const int ns = a->slots > 0 ? a->slots : (1 << c_.prm.log_slots);
    if (index % ns == 0) return add(a, a);
    const u64 g = c_.galois_element(index);
    auto it = rot_keys.find(g);
    if (it == rot_keys.end())
        throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(index) + " (EvalRotateKeyGen list)");
    return raw_rotate(a, g, *it->second, true);
}

CtPtr Evaluator::raw_mult_relin(const CtPtr& a, const CtPtr& b, const EvalKey& key) {
    if (a->npoly != 2 || b->npoly != 2 || a->ell != b->ell) throw Error(FHELIN_ERR_STATE, "mult: operands must be 2-component, same level");
    const int ell = a->ell;
    const size_t pn = (size_t)ell * c_.N;
    u64* d = c_.dalloc<u64>(3 * pn);
    launch_tensor(c_.dt, d, a->d, b->d, ell, c_.stream);
    CtPtr o = new_ct(2, ell, a->deg + b->deg, a->scale * b->scale, a->slots);
    keyswitch(d + 2 * pn, ell, key, o->d, d, d + pn, nullptr);
    c_.pool.free(d);
    return o;
}

CtPtr Evaluator::raw_modraise(const CtPtr& a, int new_ell) {
    if (a->ell != 1) throw Error(FHELIN_ERR_STATE, "modraise: the input must have exactly one limb (q0)");
    if (new_ell < 1 || new_ell > c_.L + 1) throw Error(FHELIN_ERR_ARG, "modraise: bad target limb count");
    const size_t N = c_.N;
    const int P = a->npoly;
    hipStream_t s = c_.stream;
    u64* coef = c_.dalloc<u64>((size_t)P * N);
    // INTT of the q0 limb of every polynomial, out of place (the input is immutable)
    LimbBatch ib{coef, P, nullptr, 0, 1, a->d};
    c_.ntt(ib, true);
    CtPtr up = new_ct(P, new_ell, a->deg, a->scale, a->slots);
    launch_modraise(c_.dt, up->d, coef, P, 0, new_ell, s);
    c_.ntt(LimbBatch{up->d, P * new_ell, nullptr, 0, new_ell}, false);
    launch_ok("modraise");
    c_.pool.free(coef);
    return up;
}

std::vector<CtPtr> Evaluator::raw_modraise_batch(const std::vector<CtPtr>& vin, int new_ell) {
    if (vin.size() <= 1) {
        std::vector<CtPtr> out;
        for (const CtPtr& a : vin) out.push_back(raw_modraise(a, new_ell));
        return out;
    }
    for (const CtPtr& a : vin)
        if (a->ell != 1 || a->npoly != vin[0]->npoly) throw Error(FHELIN_ERR_STATE, "modraise: the inputs must have exactly one limb (q0) and one shape");
    if (new_ell < 1 || new_ell > c_.L + 1) throw Error(FHELIN_ERR_ARG, "modraise: bad target limb count");
    std::vector<CtPtr> in = make_contiguous(vin, 7);
    const size_t N = c_.N;
    const int B = (int)in.size(), P = in[0]->npoly * B;
    hipStream_t s = c_.stream;
    u64* coef = c_.dalloc<u64>((size_t)P * N);
    c_.ntt(LimbBatch{coef, P, nullptr, 0, 1, in[0]->d}, true);
    std::vector<CtPtr> up = new_ct_batch(B, in[0]->npoly, new_ell, 1, 0, in[0]->slots);
    launch_modraise(c_.dt, up[0]->d, coef, P, 0, new_ell, s);
    c_.ntt(LimbBatch{up[0]->d, P * new_ell, nullptr, 0, new_ell}, false);
    launch_ok("modraise_batch");
    c_.pool.free(coef);
    for (int b = 0; b < B; ++b) {
        up[b]->deg = vin[b]->deg;
        up[b]->scale = vin[b]->scale;
        up[b]->slots = vin[b]->slots;
    }
    return up;
}

// ------------------------------------------------------------------------------------------------ leveled ops
CtPtr Evaluator::rescale(const CtPtr& a) {
    CtPtr o = raw_rescale(a);
    o->scale = a->scale / (long double)c_.chain.q[a->ell - 1];
    o->deg = a->deg > 1 ? a->deg - 1 : 1;
    return o;
}

CtPtr Evaluator::level_reduce(const CtPtr& a, int new_ell) {
    if (new_ell == a->ell) return a;
    if (new_ell < 1 || new_ell > a->ell) throw Error(FHELIN_ERR_ARG, "level_reduce: bad target");
    CtPtr o = new_ct(a->npoly, new_ell, a->deg, a->scale, a->slots);
    const size_t N = c_.N;
    hip_check(hipMemcpy2DAsync(o->d, (size_t)new_ell * N * 8, a->d, (size_t)a->ell * N * 8, (size_t)new_ell * N * 8, a->npoly,
                               hipMemcpyDeviceToDevice, c_.stream), "level_reduce");
    return o;
}

CtPtr Evaluator::mult_int(const CtPtr& a, u64 k, bool raise_deg, long double new_scale, int keep_ell) {
    // keep_ell > 0: the product on the first keep_ell limbs only (== level_reduce(mult_int(a), keep_ell), without touching the
    // limbs that would be dropped)
    const int ell = keep_ell > 0 ? std::min(keep_ell, a->ell) : a->ell;
    ScalarSet sc;
    for (int i = 0; i < ell; ++i) {
        u64 q = c_.chain.q[i];
        u64 v = k % q;
        sc.v[2 * i] = v;
        sc.v[2 * i + 1] = h_shoup(v, q);
    }
    CtPtr o = new_ct(a->npoly, ell, raise_deg ? a->deg + 1 : a->deg, new_scale, a->slots);
    launch_ew_scalar(c_.dt, o->d, a->d, sc, a->npoly * ell, 0, ell, c_.stream, ell < a->ell ? a->ell : 0);
    launch_ok("mult_int");
    return o;
}

// FLEXIBLEAUTO level adjustment of several degree-1 ciphertexts to ONE (limb count, scale): integer scalar x, the limbs above
// ell + 1 left out, and ONE batched rescale for all of them.  Same residues as adjust() one by one.
std::vector<CtPtr> Evaluator::adjust_deg1_batch(const std::vector<CtPtr>& v, int ell, long double scale) {
    std::vector<CtPtr> out(v.size()), pending, src;
    std::vector<size_t> pos;
    for (size_t i = 0; i < v.size(); ++i) {
        CtPtr cur = v[i];
        if (cur->ell < ell) throw Error(FHELIN_ERR_STATE, "adjust: cannot raise a ciphertext to a lower level");
        if (cur->deg == 2) cur = rescale(cur);
        if (cur->ell == ell) {
            out[i] = cur;
            continue;
        }
        src.push_back(cur);
        pos.push_back(i);
    }
    if (!src.empty()) {
        // the integer products land in ONE block (ell + 1 limbs each: only what the rescale reads), so that the batched rescale takes
        // them as they stand
        bool same = true;
        for (const CtPtr& c : src) same = same && c->npoly == src[0]->npoly;
        std::vector<CtPtr> blk = same ? new_ct_batch((int)src.size(), src[0]->npoly, ell + 1, 2, 0, src[0]->slots) : std::vector<CtPtr>();
        for (size_t j = 0; j < src.size(); ++j) {
            const CtPtr& cur = src[j];
            const long double qdrop = (long double)c_.chain.q[ell];
            const u64 k = (u64)llroundl(scale * qdrop / cur->scale);
            if (!same) {
                pending.push_back(mult_int(cur, k, true, cur->scale * (long double)k, ell + 1));
                continue;
            }
            ScalarSet sc;
            for (int l = 0; l <= ell; ++l) {
                const u64 q = c_.chain.q[l], r = k % q;
                sc.v[2 * l] = r;
                sc.v[2 * l + 1] = h_shoup(r, q);
            }
            CtPtr o = blk[j];
            o->deg = cur->deg + 1;
            o->scale = cur->scale * (long double)k;
            o->slots = cur->slots;
            launch_ew_scalar(c_.dt, o->d, cur->d, sc, cur->npoly * (ell + 1), 0, ell + 1, c_.stream, ell + 1 < cur->ell ? cur->ell : 0);
            pending.push_back(o);
        }
        launch_ok("adjust_deg1_batch");
    }
    if (!pending.empty()) {
        std::vector<CtPtr> r = rescale_batch(pending);
        for (size_t k = 0; k < pos.size(); ++k) {
            r[k]->scale = scale;
            out[pos[k]] = r[k];
        }
    }
    return out;
}

CtPtr Evaluator::adjust(const CtPtr& a, int ell, int deg, long double scale) {
    CtPtr cur = a;
    if (cur->ell < ell) throw Error(FHELIN_ERR_STATE, "adjust: cannot raise a ciphertext to a lower level");
    if (cur->ell == ell) {
        if (cur->deg == deg) return cur;
        if (cur->deg == 1 && deg == 2) {
            u64 k = (u64)llroundl(scale / cur->scale);
            return mult_int(cur, k, true, scale);
        }
        throw Error(FHELIN_ERR_STATE, "adjust: cannot lower the scale degree without dropping a limb");
    }
    if (deg == 1) {
        if (cur->deg == 2) cur = rescale(cur);
        if (cur->ell == ell) return cur;
        const long double qdrop = (long double)c_.chain.q[ell];
        u64 k = (u64)llroundl(scale * qdrop / cur->scale);
        cur = mult_int(cur, k, true, cur->scale * (long double)k, ell + 1);   // only the limbs the rescale reads
        cur = rescale(cur);
        cur->scale = scale;
        return cur;
    }
    if (cur->deg == 2) cur = rescale(cur);
    u64 k = (u64)llroundl(scale / cur->scale);
    return mult_int(cur, k, true, scale, ell);   // the product on the limbs that are kept only
}

void Evaluator::match(const CtPtr& a, const CtPtr& b, CtPtr& ao, CtPtr& bo) {
    if (a->ell == b->ell && a->deg == b->deg) {
        ao = a;
        bo = b;
        return;
    }
    // the operand with fewer limbs fixes the level; at equal level the degree-2 operand fixes the degree
    const bool a_rules = a->ell < b->ell || (a->ell == b->ell && a->deg >= b->deg);
    if (a_rules) {
        ao = a;
        bo = adjust(b, a->ell, a->deg, a->scale);
    } else {
        bo = b;
        ao = adjust(a, b->ell, b->deg, b->scale);
    }
}

CtPtr Evaluator::add(const CtPtr& a, const CtPtr& b) {
    if (a->npoly != b->npoly) throw Error(FHELIN_ERR_STATE, "add: component count mismatch");
    CtPtr x, y;
    match(a, b, x, y);
    CtPtr o = new_ct(x->npoly, x->ell, x->deg, x->scale, x->slots);
    launch_ew_add(c_.dt, o->d, x->d, y->d, x->npoly * x->ell, x->npoly * x->ell, 0, x->ell, c_.stream);
    launch_ok("add");
    return o;
}

CtPtr Evaluator::sub(const CtPtr& a, const CtPtr& b) {
    if (a->npoly != b->npoly) throw Error(FHELIN_ERR_STATE, "sub: component count mismatch");
    CtPtr x, y;
    match(a, b, x, y);
    CtPtr o = new_ct(x->npoly, x->ell, x->deg, x->scale, x->slots);
    launch_ew_sub(c_.dt, o->d, x->d, y->d, x->npoly * x->ell, x->npoly * x->ell, 0, x->ell, c_.stream);
    launch_ok("sub");
    return o;
}

CtPtr Evaluator::negate(const CtPtr& a) {
    CtPtr o = new_ct(a->npoly, a->ell, a->deg, a->scale, a->slots);
    launch_ew_neg(c_.dt, o->d, a->d, a->npoly * a->ell, 0, a->ell, c_.stream);
    launch_ok("negate");
    return o;
}

CtPtr Evaluator::add_plain(const CtPtr& a, const PtPtr& p) {
    auto enc = p->at(a->ell, a->scale);
    // one pass (op 3 of ew_items: add on component 0, copy the others) instead of a copy of the ciphertext + an add
    CtPtr o = new_ct(a->npoly, a->ell, a->deg, a->scale, a->slots);
    EwItems it;
    it.n = 1;
    it.vecs = a->npoly * a->ell;
    it.b_vecs = a->ell;
    it.out[0] = o->d;
    it.a[0] = a->d;
    it.b[0] = enc->d;
    launch_ew_items(c_.dt, it, 3, a->ell, c_.stream);
    launch_ok("add_plain");
    return o;
}

CtPtr Evaluator::mult_plain(const CtPtr& a, const PtPtr& p) {
    CtPtr x = a->deg >= 2 ? rescale(a) : a;
    auto enc = p->at(x->ell, c_.sf_real[x->level()]);
    CtPtr o = new_ct(x->npoly, x->ell, x->deg + 1, x->scale * enc->scale, x->slots);
    launch_ew_mul(c_.dt, o->d, x->d, enc->d, x->npoly * x->ell, x->ell, 0, x->ell, c_.stream);
    c_.stats.ct_pt_mult += 1;
    c_.stats.ct_pt_limbs += (u64)x->ell;
    launch_ok("mult_plain");
    return o;
}

CtPtr Evaluator::mult_no_relin(const CtPtr& a, const CtPtr& b) {
    if (a->npoly != 2 || b->npoly != 2) throw Error(FHELIN_ERR_STATE, "mult: operands must have 2 components");
    CtPtr x = a->deg >= 2 ? rescale(a) : a;
    CtPtr y = b->deg >= 2 ? rescale(b) : b;
    CtPtr xa, ya;
    match(x, y, xa, ya);
    CtPtr o = new_ct(3, xa->ell, xa->deg + ya->deg, xa->scale * ya->scale, xa->slots);
    launch_tensor(c_.dt, o->d, xa->d, ya->d, xa->ell, c_.stream);
    launch_ok("tensor");
    return o;
}

CtPtr Evaluator::relinearize(const CtPtr& a) {
    if (a->npoly == 2) return a;
    if (!relin_key) throw Error(FHELIN_ERR_KEY, "no relinearisation key (EvalMultKeyGen not called)");
    const size_t pn = (size_t)a->ell * c_.N;
    CtPtr o = new_ct(2, a->ell, a->deg, a->scale, a->slots);
    keyswitch(a->d + 2 * pn, a->ell, *relin_key, o->d, a->d, a->d + pn, nullptr);
    return o;
}

CtPtr Evaluator::mult(const CtPtr& a, const CtPtr& b) { return relinearize(mult_no_relin(a, b)); }

CtPtr Evaluator::rotate(const CtPtr& a, int index) {
    const int ns = a->slots > 0 ? a->slots : (1 << c_.prm.log_slots);
    int r = index % ns;
    if (r == 0) return clone(a);
    const u64 g = c_.galois_element(index);
    auto it = rot_keys.find(g);
    if (it == rot_keys.end())
        throw Error(FHELIN_ERR_KEY, "no rotation key for index " + std::to_string(index) + " (EvalRotateKeyGen list)");
    return raw_rotate(a, g, *it->second);
}

}  // namespace fhelin
