"""Bootstrapping and polynomial evaluation, residue for residue against the oracle.

The reference calls context->EvalBootstrap (src/FHEController.cpp:445), EvalPoly / EvalMultMany (:1291,:1297) and
EvalChebyshevFunction (:1319-1335).  The library runs them with every knob at its default (hoisted baby steps, grouped inner sums,
shared-ModDown giant steps, one EvalMod ciphertext under sparse packing, Paterson-Stockmeyer products in batched rounds,
remainder leaves born at their product's scale, bootstrap to fewer limbs under a level plan); oracle/residue_eval.py and
oracle/residue_boot.py compose the same integer functions from oracle/fhe_oracle.c.  Exported residues must be EQUAL.
Plaintext diagonals enter the oracle as the residues the library's encoder produced (fhelin_pt_export), keys as exported."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LD = np.longdouble


def _uniform_ct(orc, eng, seed, ell):
    return np.stack([orc.uniform_residues(seed + 1000 * p, eng.q[:ell], eng.N) for p in range(2)])


def _uniform_key(orc, eng, seed):
    d = eng.dnum_digits
    k = np.stack([orc.uniform_residues(seed + 50 * j, eng.moduli, eng.N) for j in range(2 * d)])
    return k.reshape(d, 2, eng.n_limbs, eng.N)


def _rev(eng, keys):
    from oracle.residue_eval import ResidueEvaluator
    return ResidueEvaluator(eng.q, eng.p, eng.psi_q, eng.psi_p, eng.alpha, eng.log_n, keys, eng.params.log_slots)


def _pair(eng, rev, x, deg=1):
    from oracle.residue_eval import RCt
    sc = float(rev.sf[len(eng.q) - x.shape[1]])
    if deg == 2:
        sc = float(LD(sc) * LD(sc))
    return eng.ct_import(x, deg=deg, scale=sc), RCt(x, deg, LD(sc))


def _same(ct, r, what=""):
    inf = ct.info()
    assert (inf["npoly"], inf["ell"], inf["deg"]) == (r.npoly, r.ell, r.deg), (what, inf, r.ell, r.deg)
    hi, lo = ct.scale_parts()
    assert LD(hi) + LD(lo) == r.scale, (what, "scale")
    assert np.array_equal(ct.export(), r.d), what


def _cheb_fit(f, a, b, degree):
    n = degree + 1
    j = np.arange(n)
    nodes = np.cos(np.pi * (j + 0.5) / n)
    fx = np.array([f(0.5 * (b - a) * t + 0.5 * (b + a)) for t in nodes])
    return [float(2.0 / n * np.sum(fx * np.cos(np.pi * k * (j + 0.5) / n))) for k in range(n)]


@pytest.fixture(scope="module")
def poly_eng(fa, orc):
    """N=2^12, 22+6 limbs; a uniform 'relinearisation key' (parity of integer functions does not need a real one)"""
    e = fa.Engine("boot12", seed=9)
    relin = _uniform_key(orc, e, 31)
    e.key_import(0, 0, relin)
    yield e, _rev(e, {"relin": relin})
    e.close()


@pytest.mark.parametrize("degree,a,b,ell,deg_in", [
    (5, -1.0, 1.0, 8, 1),             # baby = 4: one recursion node
    (31, -1.0, 1.0, 10, 2),           # a degree-2 input is rescaled first
    (47, -1.0, 1.0, 12, 1),           # the cosine fit of EvalMod
    (119, -1.0, 128.0, 14, 1),        # eval_inverse_naive on [-1, 128] (src/main.cpp:203): affine map + 7 levels
    (119, -1.0, 1.0, 22, 1),          # eval_gelu_function (src/main.cpp:356) from the top of the chain
    (300, -1.0, 1.0, 16, 2),          # eval_tanh_function (src/main.cpp:441): giants up to T_256
])
def test_chebyshev_evaluation_bit_exact(poly_eng, orc, degree, a, b, ell, deg_in):
    eng, rev = poly_eng
    f = {5: np.tanh, 31: np.sin, 47: np.cos, 119: (lambda x: 1.0 / (x + 130.0)), 300: (lambda x: np.tanh(3 * x))}[degree]
    coeffs = _cheb_fit(f, a, b, degree)
    c, r = _pair(eng, rev, _uniform_ct(orc, eng, 40 + degree, ell), deg_in)
    want = rev.eval_chebyshev(r, coeffs, a, b)
    got = eng.eval_chebyshev(c, coeffs, a, b)
    _same(got, want, ("chebyshev", degree, a, b))
    # levels consumed = OpenFHE's Paterson-Stockmeyer depth for the degree (GetMultiplicativeDepthByCoeffVector's table: 3 / 4 / 5 / 6 / 7 / 8 / 9
    # up to degree 5 / 13 / 27 / 59 / 119 / 247 / 495; reference src/FHEController.cpp:1319-1335 budgets with it), + 1 for the affine map
    eff = lambda ct: ct.info()["ell"] - (1 if ct.info()["deg"] >= 2 else 0)
    table = next(v for k, v in ((5, 3), (13, 4), (27, 5), (59, 6), (119, 7), (247, 8), (495, 9)) if degree <= k)
    assert eff(c) - eff(got) == table + (0 if (a, b) == (-1.0, 1.0) else 1), (degree, eff(c), eff(got))
    if degree == 47:      # the comparison is sensitive to the order of roundings: the same series with one knob off differs
        other = rev.eval_chebyshev(r, coeffs, a, b, leaf_at_product=False)
        assert other.d.shape != want.d.shape or not np.array_equal(other.d, want.d)


def test_chebyshev_with_merged_products_bit_exact(fa, orc, monkeypatch):
    """FHELIN_MERGED_PRODUCTS=1 (off by default: no gain, 3 x the rounding noise - DESIGN.md 6c): the power steps rescale(2 T_j T_k - 1),
    rescale(2 T_j T_k - T_1) through Evaluator::mult_affine_rescale_batch (constant / level-adjusted subtrahend into the key switch's
    accumulator times P, ModDown and rescale as one centred conversion) == orc_mult_affine_rescale, composed the same way"""
    monkeypatch.setenv("FHELIN_MERGED_PRODUCTS", "1")
    e = fa.Engine("boot12", seed=9)
    try:
        relin = _uniform_key(orc, e, 31)
        e.key_import(0, 0, relin)
        rev = _rev(e, {"relin": relin})
        rev.merged_products = True
        for degree, ell, deg_in in ((31, 10, 2), (47, 12, 1)):
            coeffs = _cheb_fit(np.sin if degree == 31 else np.cos, -1.0, 1.0, degree)
            c, r = _pair(e, rev, _uniform_ct(orc, e, 140 + degree, ell), deg_in)
            _same(e.eval_chebyshev(c, coeffs, -1.0, 1.0), rev.eval_chebyshev(r, coeffs, -1.0, 1.0), ("merged products", degree))
    finally:
        e.close()


def test_chebyshev_batch_and_sparse_coefficients_bit_exact(poly_eng, orc):
    """fhelin_eval_chebyshev_batch (the GELU containers of one sample, src/main.cpp:354-358) == the single evaluation per row; an odd
    function's fit has zero even coefficients: remainders that are constants / empty take their own paths in the recursion"""
    eng, rev = poly_eng
    coeffs = _cheb_fit(np.tanh, -1.0, 1.0, 59)
    for k in range(0, 60, 2):
        coeffs[k] = 0.0
    rows = [_pair(eng, rev, _uniform_ct(orc, eng, 500 + i, 12)) for i in range(3)]
    got = eng.eval_chebyshev_batch([p[0] for p in rows], coeffs, -1.0, 1.0)
    for g, (_, r) in zip(got, rows):
        _same(g, rev.eval_chebyshev(r, coeffs, -1.0, 1.0), "chebyshev batch")


def test_eval_exp_polynomial_and_product_tree_bit_exact(poly_eng, orc):
    """eval_exp (:1289-1311): EvalPoly Taylor-6 then EvalMultMany of 8 copies"""
    eng, rev = poly_eng
    coeffs = [1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0]
    c, r = _pair(eng, rev, _uniform_ct(orc, eng, 77, 10), 2)
    p, rp = eng.eval_poly(c, coeffs), rev.eval_poly(r, coeffs)
    _same(p, rp, "eval_poly")
    _same(eng.mult_many([p] * 8), rev.mult_many([rp] * 8), "mult_many")


def _boot_setup(fa, preset, log_slots, budget=(3, 3), **over):
    """engine with real keys + bootstrapping set up; returns (engine, oracle-side bootstrapper)"""
    from oracle.residue_boot import ResidueBootstrapper
    eng = fa.Engine(preset, seed=77, log_slots=log_slots, **over)
    eng.keygen()
    eng.gen_relin_key()
    eng.bootstrap_setup(budget[0], budget[1], 1 << log_slots)
    desc = eng.bootstrap_describe()
    keys = {"relin": eng.key_export(0), "conj": eng.key_export(2)}
    need = set()
    for st in desc["c2s"] + desc["s2c"]:
        for (g, b, _) in st["terms"]:
            need.update((g, b))
    n = desc["slots"]
    j = 1
    while j < (eng.N // 2) // n:
        need.add(n * j)
        j <<= 1
    for r in sorted(need):
        if r:
            keys[r] = eng.key_export(1, r)
    rev = _rev(eng, keys)
    boot = ResidueBootstrapper(rev, desc, lambda pt: (lambda ell, sc: eng.pt_export(pt, ell, sc)))
    return eng, boot


def _boot_input(eng, boot, ell_left=3, seed=3):
    from oracle.residue_eval import RCt
    n = boot.n
    m = np.random.default_rng(seed).uniform(-1, 1, n)
    ct = eng.encrypt(m, level=eng.n_q - ell_left)
    hi, lo = ct.scale_parts()
    inf = ct.info()
    return m, ct, RCt(ct.export(), inf["deg"], LD(hi) + LD(lo))


@pytest.mark.parametrize("log_slots", [10, 11])     # 10: sparse packing (SubSum, ONE EvalMod ciphertext over 2n slots); 11: full
def test_bootstrap_bit_exact_stage_by_stage(fa, orc, log_slots):
    """EvalBootstrap (:445) with real keys at N=2^12, 22+6 limbs: after ModRaise(+SubSum), after CoeffsToSlots + conjugation,
    after EvalMod, and the complete bootstrap — equal residues, equal 80-bit scale"""
    eng, boot = _boot_setup(fa, "boot12", log_slots)
    try:
        assert boot.desc["packed"] == (log_slots == 10)
        m, ct, r = _boot_input(eng, boot)
        for stage in (1, 2, 3):
            got = eng.bootstrap_partial(ct, stage)
            want = boot.run(r, stop_after=stage)
            assert np.array_equal(got.export(), want.d), ("stage", stage)
        out, want = eng.bootstrap(ct), boot.run(r)
        _same(out, want, "bootstrap")
        assert np.max(np.abs(eng.decrypt(out) - m)) < 2e-4      # ... and it is a bootstrap
    finally:
        eng.close()


def test_bootstrap_to_fewer_limbs_bit_exact(fa, orc):
    """a level plan makes a bootstrap raise to fewer limbs (Bootstrapper::run(ct, stop, drop): the first CoeffsToSlots stage
    takes its diagonals at another scale): explicit drop and the applied plan give the oracle's residues"""
    eng, boot = _boot_setup(fa, "boot12", 10)
    try:
        m, ct, r = _boot_input(eng, boot, ell_left=2, seed=5)
        full = eng.n_q - boot.desc["depth"]
        for drop in (2, 4):
            want = boot.run(r, drop=drop)
            assert want.ell == full - drop
            _same(eng.bootstrap_drop(ct, drop), want, ("drop", drop))
        eng.set_level_plan([full - 4])                        # source 0 of the 'program' is this bootstrap
        eng.level_plan_begin("apply")
        out = eng.bootstrap(ct)
        eng.level_plan_begin("off")
        _same(out, want, "planned bootstrap")
        assert np.max(np.abs(eng.decrypt(out) - m)) < 2e-4
    finally:
        eng.close()


def test_bootstrap_bit_exact_reference_ring(fa, orc):
    """the reference's literal ring and packing (N=2^15, 16384 slots = full packing: two EvalMod ciphertexts, level budget {3,3}
    = stages of 4+5+5 radix levels, src/FHEController.cpp:6-16,:238) on a chain just long enough for one bootstrap"""
    eng, boot = _boot_setup(fa, "reference", 14, n_q=17, n_p=-1)
    try:
        assert not boot.desc["packed"] and len(boot.desc["c2s"]) == len(boot.desc["s2c"]) == 3
        m, ct, r = _boot_input(eng, boot)
        out, want = eng.bootstrap(ct), boot.run(r)
        _same(out, want, "bootstrap, reference ring")
        assert np.max(np.abs(eng.decrypt(out) - m)) < 2e-4
    finally:
        eng.close()


def test_bootstrap_bit_exact_headline_ring_under_a_level_plan(fa, orc):
    """the chain bench.py's forward pass runs (N=2^16, 28+7 limbs, alpha=7, 16384 slots = sparse packing: SubSum, one EvalMod
    ciphertext, first SlotsToCoeffs stage over 2n slots) with the GELU bootstraps' planned drop of 4 limbs:
    ~85 real switching keys of 147 MB each go to the oracle"""
    eng, boot = _boot_setup(fa, "bench", 14, n_q=28, n_p=-1)
    try:
        assert boot.desc["packed"] and eng.alpha == 7 and eng.n_p == 7
        m, ct, r = _boot_input(eng, boot, ell_left=2)
        out, want = eng.bootstrap_drop(ct, 4), boot.run(r, drop=4)
        assert want.ell == 28 - boot.desc["depth"] - 4 == 9
        _same(out, want, "bootstrap, headline ring, drop 4")
        assert np.max(np.abs(eng.decrypt(out) - m)) < 2e-4
    finally:
        eng.close()
