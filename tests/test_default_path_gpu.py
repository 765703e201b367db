"""The path the product runs BY DEFAULT, residue for residue against the oracle (through the C-ABI):
merged rotate-and-sum key switches (ks_inner_multi + gather_sum, one ModDown), the shared-ModDown giant steps,
ct x pt / scalar / ct + pt element-wise kernels, the FLEXIBLEAUTO level adjustment, ModRaise, the decryption phase,
and the composites with every default-on knob (FHELIN_MERGE_ROT, FHELIN_EARLY_RESCALE, log-depth shift trees, block
masks in place of matmulRElarge's -128 shifts, its double-hoisted first step) left ON.  All comparisons are bit-exact (integer functions); plaintext operands enter the
oracle as the residues the library's encoder produced (fhelin_pt_export; the encoder is the client-side row a16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LD = np.longdouble


def _ct(orc, eng, seed, ell, npoly=2):
    return np.stack([orc.uniform_residues(seed + 1000 * p, eng.q[:ell], eng.N) for p in range(npoly)])


def _evk(orc, eng, seed):
    d = eng.dnum_digits
    k = np.stack([orc.uniform_residues(seed + 50 * j, eng.moduli, eng.N) for j in range(2 * d)])
    return k.reshape(d, 2, eng.n_limbs, eng.N)


def _keys(orc, eng, indices, seed=9000):
    """uniform 'keys' for the given rotation indices, imported into the engine (parity of the residue functions does not
    need real keys)"""
    keys = {}
    for r in indices:
        keys[r] = _evk(orc, eng, seed + 17 * (r % 100003))
        eng.key_import(1, r, keys[r])
    return keys


def _rev(orc, eng, keys):
    from oracle.residue_eval import ResidueEvaluator
    return ResidueEvaluator(eng.q, eng.p, eng.psi_q, eng.psi_p, eng.alpha, eng.log_n, keys, eng.params.log_slots)


def _imp(eng, rev, x, deg=1, level_scale=None):
    """the same ciphertext on both sides: engine handle + oracle-side RCt (scale = the level's Delta as a double)"""
    from oracle.residue_eval import RCt
    ell = x.shape[1]
    sc = float(rev.sf[len(eng.q) - ell]) if level_scale is None else float(level_scale)
    if deg == 2:
        sc = float(LD(sc) * LD(sc))
    return eng.ct_import(x, deg=deg, scale=sc), RCt(x, deg, LD(sc))


def _same(ct, r, what=""):
    inf = ct.info()
    assert (inf["npoly"], inf["ell"], inf["deg"]) == (r.npoly, r.ell, r.deg), (what, inf)
    assert np.array_equal(ct.export(), r.d), what


FORWARD16 = dict(log_n=16, n_q=28, n_p=-1)      # bench.py's forward chain: N=2^16, 28+7 limbs, alpha=7


@pytest.mark.parametrize("preset,over,ells,idx,rows", [
    ("toy", {}, [6, 5, 3, 1], [1, 2, 3], 2),                 # dnum=3, alpha=2: full, partial and single digit
    ("toy13", {}, [7, 4], [64, 128, 192], 3),
    ("toy13", {}, [6], [1, 2, 3, 4, 5, 6, 7], 2),            # the kernel's maximum of 7 merged rotations
    ("toy13", {}, [5], [-2], 1),                             # a single merged term
    ("reference", {}, [28, 9], [128, 256, 384], 2),          # reference parameters: N=2^15, 28+7 limbs, alpha=7
    ("bench", {}, [24, 13], [1, 2, 3], 2),                   # BASELINE size: N=2^16, 24+6 limbs
    ("bench", FORWARD16, [28], [128, 256, 384], 1),          # the headline chain
])
def test_rotate_sum_bit_exact(engine_factory, orc, preset, over, ells, idx, rows):
    """fhelin_rotate_sum (Evaluator::rotate_sum_batch -> ks_inner_multi_kernel + gather_sum_kernel, ONE ModDown) ==
    orc_rotate_sum on every residue, for a batch of rows"""
    eng = engine_factory(preset, **over)
    keys = _keys(orc, eng, idx)
    evks = np.stack([keys[r] for r in idx])
    gs = [orc.galois(eng.log_n, r) for r in idx]
    for ell in ells:
        xs = [_ct(orc, eng, 300 + 7 * i + ell, ell) for i in range(rows)]
        got = eng.rotate_sum([eng.ct_import(x) for x in xs], idx)
        for x, g in zip(xs, got):
            want = orc.rotate_sum(x, evks, gs, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
            assert np.array_equal(g.export(), want), (preset, ell)
    del keys, evks


@pytest.mark.parametrize("preset,over,ells,idx,rows,deg", [
    ("toy", {}, [6, 5, 3, 2], [1, 2, 3], 2, 1),              # dnum=3, alpha=2: full, partial and single digit
    ("toy13", {}, [7, 4], [64, 128, 192], 3, 1),             # three rotations, three rows
    ("toy13", {}, [6], [1, 2, 3, 4, 5, 6, 7], 2, 2),         # the kernel's maximum of 7 rotations; degree-2 rows are rescaled first
    ("toy13", {}, [5], [-2], 1, 1),                          # a single rotated term
    ("reference", {}, [13, 9], [128, 256, 384], 2, 1),       # matmulRElarge's first step at the reference's ring (N=2^15, 28+7 limbs)
    ("bench", FORWARD16, [13], [128, 256, 384], 3, 1),       # ... and at the headline ring: the LDS-staged digit tiles (rotations by 64 k slots)
])
def test_hoisted_dot_bit_exact(engine_factory, orc, preset, over, ells, idx, rows, deg):
    """fhelin_hoisted_dot (Evaluator::hoisted_dot_rows: one ModUp, ks_inner_multi_kernel over rotation keys with the plaintexts folded
    in by fold_key_kernel, hoist_addends_kernel, ONE ModDown) == orc_hoisted_dot on every residue, for a batch of rows.  The plaintext
    encodings over the full key basis enter the oracle as exported (fhelin_pt_export at n_q + n_p limbs)."""
    eng = engine_factory(preset, **over)
    keys = _keys(orc, eng, idx, seed=6100)
    rev = _rev(orc, eng, keys)
    rng = np.random.default_rng(77)
    ns = 1 << eng.params.log_slots
    pts = [eng.encode(rng.uniform(-1, 1, ns)) for _ in range(len(idx) + 1)]
    encs = [(lambda p: (lambda ell, sc: eng.pt_export(p, ell, sc)))(p) for p in pts]
    for ell in ells:
        pairs = [_imp(eng, rev, _ct(orc, eng, 500 + 7 * i + ell, ell), deg=deg) for i in range(rows)]
        for rescale in (False, True):            # True: ModDown and rescale as one basis conversion (moddown_rescale_*_kernel)
            if rescale and (ell if deg == 1 else ell - 1) < 2:
                continue
            got = eng.hoisted_dot([p[0] for p in pairs], pts, idx, rescale=rescale)
            for (c, r), g in zip(pairs, got):
                want = rev.hoisted_dot(r, encs, idx, rescale=rescale)
                _same(g, want, (preset, ell, rescale))
                hi, lo = g.scale_parts()
                assert LD(hi) + LD(lo) == want.scale
    del keys


def test_hoisted_dot_is_the_sum_of_rotated_products(fa):
    """semantics with real keys: decrypt(hoisted_dot(x, V, idx)) = x * V_0 + sum_r rot(x, idx_r) * V_{r+1}, and it agrees with the
    products-then-rotations form the library used before (mult_plain + rotate + add) to key-switching noise"""
    eng = fa.Engine("toy13", seed=21)
    try:
        idx = [128, 256, 384]
        eng.keygen()
        eng.gen_rotation_keys(idx)
        ns = 1 << eng.params.log_slots
        rng = np.random.default_rng(5)
        xs = [rng.uniform(-1, 1, ns) for _ in range(3)]
        vs = [rng.uniform(-1, 1, ns) for _ in range(4)]
        cts = [eng.encrypt(x) for x in xs]
        outs = eng.hoisted_dot(cts, [eng.encode(v) for v in vs], idx)
        for x, o in zip(xs, outs):
            want = x * vs[0] + sum(np.roll(x, -r) * vs[k + 1] for k, r in enumerate(idx))
            assert o.info()["deg"] == 2
            assert np.max(np.abs(eng.decrypt(o)[:ns] - want)) < 1e-6
        for x, o in zip(xs, eng.hoisted_dot(cts, [eng.encode(v) for v in vs], idx, rescale=True)):     # ModDown + rescale in one conversion
            want = x * vs[0] + sum(np.roll(x, -r) * vs[k + 1] for k, r in enumerate(idx))
            assert o.info()["deg"] == 1 and o.info()["ell"] == cts[0].info()["ell"] - 1
            assert np.max(np.abs(eng.decrypt(o)[:ns] - want)) < 1e-6
    finally:
        eng.close()


@pytest.mark.parametrize("preset,over,ell,idx", [
    ("toy13", {}, 7, [1, 2, 0, 4, 8, 16, 32, 64]),           # an unrotated addend + exactly 7 rotated terms
    ("toy13", {}, 3, [1, 2, 4, 8, 16, 32, 64, 128, 256]),    # 9 terms: a group of 7 and a group of 2 (partial digit)
    ("toy13", {}, 5, [1, 2, 4, 8, 16, 32, 64, 128]),         # 8 terms: a group of 7 and one plain rotation
    ("bench", {}, 24, [16, 32, 48, 64, 80, 96, 112]),        # R * beta = 28 products per coefficient at N=2^16
    ("reference", {}, 28, [1, 2, 3]),
])
def test_rotate_each_sum_bit_exact(engine_factory, orc, preset, over, ell, idx):
    """fhelin_rotate_each_sum (the giant steps of the bootstrapping linear transforms: one ModUp per term, shared
    ModDown) == the oracle's composition of orc_rotate_each_sum / orc_rotate / orc_add"""
    eng = engine_factory(preset, **over)
    keys = _keys(orc, eng, [r for r in idx if r], seed=4000)
    rev = _rev(orc, eng, keys)
    pairs = [_imp(eng, rev, _ct(orc, eng, 800 + 3 * i, ell)) for i in range(len(idx))]
    got = eng.rotate_each_sum([p[0] for p in pairs], idx)
    _same(got, rev.rotate_each_sum([p[1] for p in pairs], idx), (preset, ell))
    del keys


@pytest.mark.parametrize("preset,ell,R", [("bench", 24, 7), ("reference", 28, 7), ("toy", 6, 7)])
def test_merged_inner_product_extreme_operands(engine_factory, orc, preset, ell, R):
    """all-maximal residues (q-1) in the ciphertexts and in every key limb: the worst case for the 128-bit
    accumulation of R * beta = 28 products before a Barrett reduction (the kernel folds every 16)"""
    eng = engine_factory(preset)
    idx = list(range(1, R + 1))
    mx_key = np.stack([np.full(eng.N, int(m) - 1, dtype=np.uint64) for m in eng.moduli])
    mx_key = np.ascontiguousarray(np.broadcast_to(mx_key, (eng.dnum_digits, 2) + mx_key.shape))
    for r in idx:
        eng.key_import(1, r, mx_key)
    keys = {r: mx_key for r in idx}
    rev = _rev(orc, eng, keys)
    mx = np.stack([np.stack([np.full(eng.N, int(m) - 1, dtype=np.uint64) for m in eng.q[:ell]])] * 2)
    pairs = [_imp(eng, rev, mx) for _ in idx]
    _same(eng.rotate_each_sum([p[0] for p in pairs], idx), rev.rotate_each_sum([p[1] for p in pairs], idx), "each_sum max")
    got = eng.rotate_sum([pairs[0][0]], idx)[0]
    _same(got, rev.rotate_sum(pairs[0][1], idx), "rotate_sum max")


@pytest.mark.parametrize("preset,over,ells", [("toy13", {}, [7, 3, 1]), ("reference", {}, [28]), ("bench", {}, [24, 8]),
                                              ("bench", FORWARD16, [28])])
def test_ct_pt_products_and_sums_bit_exact(engine_factory, orc, preset, over, ells):
    """EvalMult(ct,pt) / EvalAdd(ct,pt) (reference :427,:414): the single-operand kernels (ew_binary) and the batched
    ones the forward pass uses (ew_items<0> via mult_plain_batch, ew_items<3> via add_plain_batch), also on a
    degree-2 operand (rescaled first)"""
    eng = engine_factory(preset, **over)
    rev = _rev(orc, eng, {})
    ns = 1 << eng.params.log_slots
    rng = np.random.default_rng(5)
    w = eng.encode(rng.uniform(-1, 1, ns))
    b = eng.encode(rng.uniform(-1, 1, ns))
    wenc = lambda ell, sc: eng.pt_export(w, ell, sc)
    benc = lambda ell, sc: eng.pt_export(b, ell, sc)
    for ell in ells:
        c, r = _imp(eng, rev, _ct(orc, eng, 40 + ell, ell))
        prod, rprod = eng.mult(c, w), rev.mult_plain(r, wenc)
        _same(prod, rprod, ("mult_plain", preset, ell))
        _same(eng.add(prod, b), rev.add_plain(rprod, benc), ("add_plain deg2", preset, ell))
        _same(eng.add(c, b), rev.add_plain(r, benc), ("add_plain", preset, ell))
        if ell >= 2:
            _same(eng.mult(prod, b), rev.mult_plain(rprod, benc), ("mult_plain of a degree-2 operand", preset, ell))
        # batched rows: matmul with a rotate-and-sum over 1 slot = rows[i] * w + bias, no rotations
        rows = [_imp(eng, rev, _ct(orc, eng, 90 + ell + 5 * i, ell)) for i in range(3)]
        outs = eng.matmul_pt([x[0] for x in rows], w, b, 1, 1)
        for o, (_, rr) in zip(outs, rows):
            _same(o, rev.add_plain(rev.mult_plain(rr, wenc), benc), ("matmul_pt(slots=1)", preset, ell))


def _real_scalars(q, v):
    """round(|v|) with the sign restored, modulo each q (polyeval.cpp real_to_scalars)"""
    from oracle.residue_eval import _llround
    k = _llround(v)
    return np.array([k % int(m) for m in q], dtype=np.uint64)


@pytest.mark.parametrize("preset,ells", [("toy13", [7, 2]), ("bench", [24])])
def test_real_constant_ops_bit_exact(engine_factory, orc, preset, ells):
    """EvalMult(ct, double) / EvalAdd(ct, double) as per-limb scalars (ew_scalar_kernel / ew_addscalar_kernel), the
    form every Chebyshev / power-basis evaluation step uses"""
    eng = engine_factory(preset)
    rev = _rev(orc, eng, {})
    for ell in ells:
        x = _ct(orc, eng, 60 + ell, ell)
        c, r = _imp(eng, rev, x)
        ql = eng.q[:ell]
        for cst in (0.7310585786300049, -1.25e-3, 3.0):
            sf = rev.sf[len(eng.q) - ell]
            s = _real_scalars(ql, LD(cst) * sf)
            want = np.stack([orc.mul_scalar(x[p], s, ql) for p in range(2)])
            got = eng.mult_real(c, cst)
            assert got.info()["deg"] == 2 and np.array_equal(got.export(), want), (preset, ell, cst)
            s = _real_scalars(ql, LD(cst) * r.scale)
            want = x.copy()
            want[0] = orc.add_scalar(x[0], s, ql)
            assert np.array_equal(eng.add_real(c, cst).export(), want), (preset, ell, cst)


@pytest.mark.parametrize("preset,over", [("toy13", {}), ("bench", {}), ("bench", FORWARD16)])
def test_level_and_degree_adjustment_bit_exact(engine_factory, orc, preset, over):
    """EvalAdd / EvalSub of operands at different (level, noiseScaleDeg) under FLEXIBLEAUTO: the operand with more limbs
    is multiplied by an integer, level-reduced and rescaled (Evaluator::adjust -> ew_scalar_kernel + rescale)"""
    eng = engine_factory(preset, **over)
    rev = _rev(orc, eng, {})
    L1 = len(eng.q)
    hi, mid, lo = L1, L1 - 2, max(2, L1 - 4)
    a, ra = _imp(eng, rev, _ct(orc, eng, 1, hi))
    b, rb = _imp(eng, rev, _ct(orc, eng, 2, mid))
    c2, rc2 = _imp(eng, rev, _ct(orc, eng, 3, lo), deg=2)
    d2, rd2 = _imp(eng, rev, _ct(orc, eng, 4, hi), deg=2)
    _same(eng.add(a, b), rev.add(ra, rb), "deg1 + deg1, two levels apart")
    _same(eng.sub(b, a), rev.sub(rb, ra), "sub, operands swapped")
    _same(eng.add(a, c2), rev.add(ra, rc2), "deg1 raised to a deeper deg2 operand")
    _same(eng.add(d2, b), rev.add(rd2, rb), "deg2 operand with more limbs: rescale, scale, reduce")
    _same(eng.add(a, d2), rev.add(ra, rd2), "same level, deg1 -> deg2 (integer multiply only)")


@pytest.mark.parametrize("preset,new_ell", [("toy13", 7), ("reference", 28), ("bench", 24), ("deep", 30), ("toy13", 3)])
def test_modraise_bit_exact(engine_factory, orc, preset, new_ell):
    """ModRaise (first step of EvalBootstrap, :445): INTT of the q0 limb, centred lift into every limb, NTT"""
    eng = engine_factory(preset)
    x = _ct(orc, eng, 77, 1)
    q0 = int(eng.q[0])
    co = np.stack([orc.ntt_inverse(x[p, 0], q0, eng.psi_q[0]) for p in range(2)])
    co[0, :4] = [0, q0 - 1, q0 // 2, q0 // 2 + 1]                 # the boundary of the centring
    x = np.stack([orc.ntt_forward(co[p], q0, eng.psi_q[0]) for p in range(2)])[:, None, :]
    got = eng.raw_modraise(eng.ct_import(x), new_ell).export()
    want = orc.modraise(x[:, 0], new_ell, eng.q[:new_ell], eng.psi_q[:new_ell])
    assert got.shape == want.shape == (2, new_ell, eng.N)
    assert np.array_equal(got, want)
    with pytest.raises(Exception):
        eng.raw_modraise(eng.ct_import(_ct(orc, eng, 5, 2)), new_ell)   # more than one limb: refused


@pytest.mark.parametrize("preset,ell,npoly", [("toy13", 7, 2), ("toy13", 4, 3), ("bench", 24, 2)])
def test_decryption_phase_bit_exact(fa, orc, preset, ell, npoly):
    """c0 + c1 s (+ c2 s^2): the multiply-accumulate kernel under context->Decrypt (ew_muladd_kernel)"""
    eng = fa.Engine(preset, seed=5)
    try:
        eng.keygen()
        s = eng.secret_export()[:ell]
        x = _ct(orc, eng, 11, ell, npoly)
        ql = eng.q[:ell]
        want = orc.muladd(x[0], x[1], s, ql)
        if npoly == 3:
            want = orc.muladd(want, x[2], orc.mul(s, s, ql), ql)
        got = eng.raw_phase(eng.ct_import(x)).export()
        assert got.shape == (1, ell, eng.N) and np.array_equal(got[0], want)
    finally:
        eng.close()


@pytest.mark.parametrize("preset,over,ell", [("bench", FORWARD16, 16), ("reference", {}, 28), ("toy13", {}, 5)])
def test_rotsum_repeat_default_knobs_bit_exact(engine_factory, orc, preset, over, ell):
    """rotsum / repeat exactly as the forward pass runs them (FHELIN_MERGE_ROT and FHELIN_EARLY_RESCALE at their
    defaults): a degree-2 product is rescaled first, tree steps run in merged pairs {s,2s,3s}, an odd last step as
    rotate+add"""
    eng = engine_factory(preset, **over)
    need = [128, 256, 384, 512, -1, -2, -3, -4, -8, -12, -16]
    keys = _keys(orc, eng, need, seed=600)
    rev = _rev(orc, eng, keys)
    c1, r1 = _imp(eng, rev, _ct(orc, eng, 7, ell))
    c2, r2 = _imp(eng, rev, _ct(orc, eng, 8, ell), deg=2)
    _same(eng.rotsum(c1, 8, 128), rev.rotsum(r1, 8, 128), "rotsum: merged pair + single step")
    _same(eng.rotsum(c2, 4, 128), rev.rotsum(r2, 4, 128), "rotsum of a product: early rescale + one merged pair")
    _same(eng.repeat(c2, 32, 1), rev.repeat(r2, 32, 1), "repeat: two merged pairs + single step, negative steps")
    _same(eng.rotsum(c1, 1, 128), r1, "zero steps: a copy")
    # with the keys of s..7s three steps run as one merged key switch: 5 steps = a triple + a pair
    more = _keys(orc, eng, [m * 128 for m in (5, 6, 7)] + [1024, 2048, 3072], seed=777)
    keys.update(more)
    rev = _rev(orc, eng, keys)
    _same(eng.rotsum(c2, 32, 128), rev.rotsum(r2, 32, 128), "rotsum: merged triple {s..7s} + merged pair")
    _same(eng.rotsum(c1, 8, 128), rev.rotsum(r1, 8, 128), "rotsum: exactly one merged triple")
    del keys, more


def test_layout_shuffles_and_large_matmul_default_path_bit_exact(fa, orc):
    """the log-depth forms of the reference's rotate-by-one chains (shift_sum in wrapUpExpanded / wrap_containers,
    shift_fan in unwrapExpanded) and matmulRElarge with block masks in place of its -128 shifts, composed on the oracle side in the
    same tree order, at the reference's ring (N=2^15, 16384 slots) on a short chain"""
    eng = fa.Engine("reference", seed=3, n_q=6, n_p=2, dnum=3)
    try:
        need = set()
        for u in (128, 512, 2048):
            need.update((u, 2 * u, 3 * u))
        need.update((8192, 1024, 4096, 12288))                      # rotsum(128,128); matmulRElarge's 5-step tree by 512 (pairs: no 5s/7s keys here)
        need.update((1536, 3072, 6144, 2560, 3584))                 # ... and the radix-8 shift sum by 512 of the fused containers
        for u in (-1, -4, -16):
            need.update((u, 2 * u, 3 * u))
        need.update((-64, 1, 2, -512, -1024, -2048))                # repeat(128,1), fans, wrap_containers
        need.update((-1536, -2560, -3072, -3584))                   # ... of 9 degree-2 rows: the radix-8 shift sum by -512, as in the forward pass
        keys = _keys(orc, eng, sorted(need), seed=100)
        keys[-4096] = keys[12288]                                   # one Galois element (16384 slots): the engine holds one key for both
        rev = _rev(orc, eng, keys)
        ns, rng = 16384, np.random.default_rng(9)
        pt = lambda v: eng.encode(v)
        enc_of = lambda p: (lambda ell, sc: eng.pt_export(p, ell, sc))
        ell = 6
        # wrap_containers: sum_i rot(c[n-1-i], -512 i)
        cts = [_imp(eng, rev, _ct(orc, eng, 200 + i, ell)) for i in range(5)]
        _same(eng.wrap_containers([c[0] for c in cts], 5), rev.wrap_containers([c[1] for c in cts], 5), "wrap_containers")
        # wrapUpExpanded / unwrapExpanded with the (i % 128 == 0) mask
        m128 = np.zeros(ns)
        m128[::128] = 1.0
        menc = enc_of(pt(m128))
        w, rw = eng.wrapUpExpanded([c[0] for c in cts[:3]]), rev.wrapUpExpanded([c[1] for c in cts[:3]], menc)
        _same(w, rw, "wrapUpExpanded")
        for o, r in zip(eng.unwrapExpanded(w, 3), rev.unwrapExpanded(rw, 3, menc)):
            _same(o, r, "unwrapExpanded")
        # matmulRElarge (four weight blocks, shared form): W''_t block-wise re-arranged weights, mask value 0.5 on [0, 512), bias
        wv = [rng.uniform(-1, 1, ns) / 8 for _ in range(4)]
        ws = [pt(v) for v in wv]
        bias_values = rng.uniform(-1, 1, ns)
        bias = pt(bias_values)
        w2 = []
        for t in range(4):
            v = np.zeros(ns)
            for b in range(128):
                v[128 * b:128 * (b + 1)] = wv[(b - t) % 4][128 * b:128 * (b + 1)]
            w2.append(enc_of(pt(np.roll(v, -128 * t))))                 # V_t = rot(W''_t, 128 t): the first step is double-hoisted
        m512 = np.zeros(ns)
        m512[:512] = 0.5
        rows = cts[:2]
        outs = eng.matmulRElarge([c[0] for c in rows], ws, bias, 0.5)
        want = rev.matmulRElarge([c[1] for c in rows], w2, enc_of(bias), enc_of(pt(m512)))
        for o, r in zip(outs, want):
            _same(o, r, "matmulRElarge")
        # generate_containers over UNREAD rows of matmulRElarge: the trees of the rows and the container sum as one shift sum of 32
        # cyclic plaintext-weighted sums (Composite::relarge_container, ew_cyclic_dot_kernel); 9 rows = one fused group
        rows9 = [_imp(eng, rev, _ct(orc, eng, 900 + i, ell)) for i in range(9)]
        lazy = eng.matmulRElarge([c[0] for c in rows9], ws, bias, 0.5)
        got = eng.generate_containers(lazy)
        assert len(got) == 1
        masks = []
        for j in range(32):
            m = np.zeros(ns)
            m[512 * j:512 * (j + 1)] = 0.5
            masks.append(enc_of(pt(m)))
        tiled = np.zeros(ns)
        for i in range(9):
            tiled += np.roll(bias_values, 512 * i)
        us = [rev.relarge_u(c[1], w2) for c in rows9]
        _same(got[0], rev.relarge_container(us, masks, enc_of(pt(tiled))), "fused containers")
        # a row that IS read first is matmulRElarge's own (and the container then takes the evaluated rows the old way)
        lazy2 = eng.matmulRElarge([c[0] for c in rows9], ws, bias, 0.5)
        _same(lazy2[3], rev.matmulRElarge([rows9[3][1]], w2, enc_of(bias), enc_of(pt(m512)))[0], "a read row")
        full = rev.matmulRElarge([c[1] for c in rows9], w2, enc_of(bias), enc_of(pt(m512)))
        for i in (0, 8):
            _same(lazy2[i], full[i], ("row read after another", i))
        _same(eng.generate_containers(lazy2)[0], rev.wrap_containers(list(reversed(full)), 9), "containers of evaluated rows")
    finally:
        eng.close()


def test_unwrap_expanded_bulk_bit_exact(fa, orc):
    """unwrapExpanded with 64 or more rows read together (Composite::unwrapExpanded_bulk: the fans rot(c, j), -127 <= j < n, and every
    row as a plaintext-weighted sliding-window sum over them, ew_window_dot_kernel) == ResidueEvaluator.unwrapExpanded_bulk; a single
    row read on its own keeps the tree form (mask, repeat)"""
    eng = fa.Engine("reference", seed=5, n_q=4, n_p=2, dnum=2)
    try:
        need = set()
        for s in (1, -1):
            need.update(s * k for k in range(1, 8))
            need.update(8 * s * k for k in range(1, 8))
            need.add(64 * s)
        keys = _keys(orc, eng, sorted(need), seed=300)
        rev = _rev(orc, eng, keys)
        ns, n, ell = 16384, 70, 4
        enc_of = lambda p: (lambda l, sc: eng.pt_export(p, l, sc))
        masks = []
        for k in range(128):
            m = np.zeros(ns)
            m[k::128] = 1.0
            masks.append(enc_of(eng.encode(m)))
        c, r = _imp(eng, rev, _ct(orc, eng, 77, ell))
        rows = eng.unwrapExpanded(c, n)
        idx = list(range(n))
        eng.force(rows)                                            # all 70 rows read together: the bulk form
        want = rev.unwrapExpanded_bulk(r, n, idx, masks)
        for i in (0, 1, 31, 32, 63, 64, 69):
            _same(rows[i], want[i], ("bulk row", i))
        one = eng.unwrapExpanded(c, n)[5]                          # one row read on its own: the tree form
        _same(one, rev.unwrapExpanded(r, n, masks[0])[5], "single row")
    finally:
        eng.close()


@pytest.mark.parametrize("preset,ell", [("toy13", 6), ("bench", 24)])
def test_batched_leaf_ops_bit_exact(engine_factory, orc, preset, ell):
    """the batched leaf entry points (fhelin_rotate_batch / rescale_batch / mult_plain_batch / mult_batch / add_batch:
    rows of a matmul in one launch set; bench.py's op section times them) return the residues of the single ops"""
    eng = engine_factory(preset)
    keys = _keys(orc, eng, [5], seed=1500)
    relin = _evk(orc, eng, 31)
    eng.key_import(0, 0, relin)
    rev = _rev(orc, eng, keys)
    rows = [_imp(eng, rev, _ct(orc, eng, 900 + 11 * i, ell)) for i in range(3)]
    cs, rs = [r[0] for r in rows], [r[1] for r in rows]
    w = eng.encode(np.random.default_rng(1).uniform(-1, 1, 1 << eng.params.log_slots))
    wenc = lambda e, sc: eng.pt_export(w, e, sc)
    for got, r in zip(eng.rotate_batch(cs, 5), rs):
        _same(got, rev.rotate(r, 5), "rotate_batch")
    prods = eng.mult_plain_batch(cs, w)
    rprods = [rev.mult_plain(r, wenc) for r in rs]
    for got, r in zip(prods, rprods):
        _same(got, r, "mult_plain_batch")
    for got, r in zip(eng.rescale_batch(prods), rprods):
        _same(got, rev.rescale(r), "rescale_batch")
    for got, a, b in zip(eng.add_batch(cs, cs[1:] + cs[:1]), rs, rs[1:] + rs[:1]):
        _same(got, rev.add(a, b), "add_batch")
    for got, a, b in zip(eng.mult_batch(cs, cs[1:] + cs[:1]), rs, rs[1:] + rs[:1]):
        want = orc.mult_relin(a.d, b.d, relin, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
        assert got.info()["deg"] == 2 and np.array_equal(got.export(), want), "mult_batch"


def test_deferred_rows_match_eager_and_skip_dead_rows(fa, orc):
    """fhelin_fc_matmul_pt / fhelin_fc_unwrapExpanded return deferred rows: whatever subset is read, in whatever order,
    every row holds the residues of the eager call; rows nobody reads are never evaluated (no key switch is counted)"""
    eng = fa.Engine("reference", seed=3, n_q=6, n_p=2, dnum=3)
    try:
        need = set()
        for u in (128, 512, 2048, -1, -4, -16):
            need.update((u, 2 * u, 3 * u))
        need.update((8192, -64, 1, 2, 4))
        _keys(orc, eng, sorted(need), seed=100)
        rng = np.random.default_rng(4)
        ns = 16384
        w, b = eng.encode(rng.uniform(-1, 1, ns) / 8), eng.encode(rng.uniform(-1, 1, ns))
        rows = [eng.ct_import(_ct(orc, eng, 500 + i, 6)) for i in range(6)]
        eng.set_lazy_rows(False)
        eager = [o.export() for o in eng.matmulRE(rows, w, b)]
        src = eng.mult(rows[0], w)
        eager_un = [o.export() for o in eng.unwrapExpanded(src, 6)]
        eng.set_lazy_rows(True)
        eng.sync()
        eng.stats(reset=True)
        lazy = eng.matmulRE(rows, w, b)
        assert eng.stats()["keyswitch"] == 0                           # nothing evaluated yet
        assert np.array_equal(lazy[4].export(), eager[4])              # one row alone
        assert eng.stats()["keyswitch"] == 7                           # one rotate-and-sum tree (7 steps), not six
        s = eng.add(lazy[1], lazy[2])                                  # a consumer of two rows of the group
        want = eng.add(eng.ct_import(eager[1]), eng.ct_import(eager[2])).export()
        assert np.array_equal(s.export(), want)
        assert eng.stats()["keyswitch"] == 42                          # second partial read: the five rows left, one batch
        assert lazy[0].info()["ell"] == 5                              # GetLevel() on a deferred row evaluates it
        for i in (0, 3, 5):
            assert np.array_equal(lazy[i].export(), eager[i])
        un = eng.unwrapExpanded(src, 6)
        for i in (5, 0, 3):                                            # odd order; row 5 needs rot(rot(c,1),4)
            assert np.array_equal(un[i].export(), eager_un[i]), i
        before = eng.stats()["keyswitch"]
        dead = eng.matmulRE(rows, w, b)                                # never read
        del dead
        eng.sync()
        assert eng.stats()["keyswitch"] == before
    finally:
        eng.close()


@pytest.mark.parametrize("preset,ell,n", [("toy13", 6, 5), ("bench", 24, 31), ("toy13", 3, 1)])
def test_linear_combination_bit_exact(engine_factory, orc, preset, ell, n):
    """fhelin_lincomb (EvalLinearWSum; the base case of every Chebyshev / power-basis evaluation) in one kernel pass ==
    the chain EvalMult(ct,double) -> EvalAdd -> EvalAdd(ct,double) on the oracle side"""
    from oracle.residue_eval import RCt
    eng = engine_factory(preset)
    rev = _rev(orc, eng, {})
    rng = np.random.default_rng(8)
    pairs = [_imp(eng, rev, _ct(orc, eng, 700 + i, ell)) for i in range(n)]
    coef = rng.uniform(-2, 2, n)
    if n > 3:
        coef[2] = 0.0                                                # a skipped term
    c0 = 0.37
    ql = eng.q[:ell]
    sf = rev.sf[len(eng.q) - ell]
    acc = None
    for (_, r), ck in zip(pairs, coef):
        if ck == 0.0:
            continue
        s = _real_scalars(ql, LD(ck) * sf)
        t = np.stack([orc.mul_scalar(r.d[p], s, ql) for p in range(2)])
        acc = t if acc is None else np.stack([orc.add(acc[p], t[p], ql) for p in range(2)])
    out_scale = pairs[0][1].scale * sf
    acc[0] = orc.add_scalar(acc[0], _real_scalars(ql, LD(c0) * out_scale), ql)
    got = eng.lincomb([p[0] for p in pairs], coef, c0)
    _same(got, RCt(acc, 2, out_scale), ("lincomb", preset, n))


@pytest.mark.parametrize("preset,ell,n,deg", [("bench", 24, 3, 1), ("reference", 9, 33, 2)])
def test_wrapUpRepeated_inner_product_bit_exact(engine_factory, orc, preset, ell, n, deg):
    """wrapUpRepeated (:1060-1068) = sum_i v_i * block_mask_i: the fused inner-product kernel (ew_dot, also the diagonal sums
    of the bootstrapping stages) == products and additions composed on the oracle side; 33 terms cross the 32-term chunk;
    degree-2 inputs are rescaled first"""
    eng = engine_factory(preset)
    rev = _rev(orc, eng, {})
    ns = 1 << eng.params.log_slots
    pairs = [_imp(eng, rev, _ct(orc, eng, 40 + i, ell), deg=deg) for i in range(n)]
    got = eng.wrapUpRepeated([p[0] for p in pairs])
    acc = None
    for i, (_, r) in enumerate(pairs):
        m = np.zeros(ns)
        m[128 * i:128 * (i + 1)] = 1.0
        pt = eng.encode(m)
        t = rev.mult_plain(r, lambda e, sc: eng.pt_export(pt, e, sc))
        acc = t if acc is None else rev.add(acc, t)
    _same(got, acc, ("wrapUpRepeated", preset, n))


def test_unwrapRepeatedLarge_shared_steps_bit_exact(fa, orc):
    """unwrapRepeatedLarge (:1102-1123) in its two-stage shared form (per range of 8 tokens: block k spread over each token's
    512 slots by one merged key switch and replicated over the four ranges; per token a mask and repeat(., 8, -512)) == the
    same composition on the oracle; 10 tokens = two ranges, the second one ragged"""
    eng = fa.Engine("reference", seed=3, n_q=6, n_p=2, dnum=3)
    try:
        need = [128, 256, 384, -128, -256, -384] + [512 * m for m in range(1, 8)] + [4096, 8192, 12288]
        keys = _keys(orc, eng, need, seed=300)
        rev = _rev(orc, eng, keys)
        pts = {}

        def enc_of_values(v):
            key = v.tobytes()
            if key not in pts:
                pts[key] = eng.encode(v)
            p = pts[key]
            return lambda ell, sc: eng.pt_export(p, ell, sc)

        c, r = _imp(eng, rev, _ct(orc, eng, 61, 6))
        got = eng.unwrapRepeatedLarge([c], 10)                    # one container holding 10 tokens
        want = rev.unwrapRepeatedLarge([r], 10, enc_of_values)
        assert len(got) == 10
        for t in range(10):
            for k in range(4):
                _same(got[t][k], want[t][k], ("unwrapRepeatedLarge", t, k))
        part = eng.unwrapRepeatedLarge_range([c], 10, 7, 3)      # the token range form used by row sharding (spans both ranges)
        for t in (7, 8, 9):
            for k in range(4):
                _same(part[t - 7][k], want[t][k], ("range", t, k))
    finally:
        eng.close()
