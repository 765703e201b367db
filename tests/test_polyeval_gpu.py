"""EvalPoly / EvalMultMany / Chebyshev evaluation on the GPU, checked by
decryption against numpy.  Tolerances: polynomial approximation error is excluded by comparing with the SAME
polynomial evaluated in float64; what remains is CKKS noise growth over <= 9 levels (< 1e-4 stated)."""
import math

import numpy as np
import pytest
from numpy.polynomial import chebyshev as Ch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(fa):
    e = fa.Engine("toy13", seed=5, n_q=14, n_p=4, dnum=4)
    e.keygen()
    e.gen_relin_key()
    yield e
    e.close()


def cheb_coeffs(f, a, b, degree):
    """EvalChebyshevCoefficients convention: c_k = 2/n sum_j f(x_j) cos(pi k (j+1/2)/n), n = degree+1."""
    n = degree + 1
    j = np.arange(n)
    nodes = np.cos(np.pi * (j + 0.5) / n)
    fx = np.array([f(0.5 * (b - a) * t + 0.5 * (b + a)) for t in nodes])
    return np.array([2.0 / n * np.sum(fx * np.cos(np.pi * k * (j + 0.5) / n)) for k in range(n)])


def cheb_eval(c, x, a, b):
    u = (2 * x - (a + b)) / (b - a)
    cc = np.array(c, dtype=float).copy()
    cc[0] *= 0.5
    return Ch.chebval(u, cc)


def _x(eng, seed, lo, hi):
    return np.random.default_rng(seed).uniform(lo, hi, 1 << eng.params.log_slots)


def test_real_constants(eng):
    x = _x(eng, 1, -1, 1)
    c = eng.encrypt(x)
    assert np.max(np.abs(eng.decrypt(eng.mult_real(c, -2.5)) - (-2.5 * x))) < 1e-8
    assert np.max(np.abs(eng.decrypt(eng.add_real(c, 0.75)) - (x + 0.75))) < 1e-8
    m = eng.mult(c, c)                      # degree-2 ciphertext: constant is added at scale Delta^2
    assert np.max(np.abs(eng.decrypt(eng.add_real(m, -1.0)) - (x * x - 1))) < 1e-7


def test_eval_poly_taylor_exp(eng):
    """the reference's eval_exp polynomial (src/FHEController.cpp:1291) and its EvalMultMany(8 copies)"""
    coeffs = [1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0]
    x = _x(eng, 2, -1, 0.25)
    c = eng.encrypt(x)
    p = eng.eval_poly(c, coeffs)
    want = np.polyval(coeffs[::-1], x)
    assert np.max(np.abs(eng.decrypt(p) - want)) < 1e-6
    p8 = eng.mult_many([p] * 8)
    assert np.max(np.abs(eng.decrypt(p8) - want ** 8)) < 1e-5


@pytest.mark.parametrize("degree", [5, 31, 119])
def test_chebyshev_series_matches_numpy(eng, degree):
    rng = np.random.default_rng(degree)
    c = rng.uniform(-1, 1, degree + 1) / np.arange(1, degree + 2)
    x = _x(eng, 3, -1, 1)
    got = eng.decrypt(eng.eval_chebyshev(eng.encrypt(x), c))
    assert np.max(np.abs(got - cheb_eval(c, x, -1, 1))) < 1e-5


def test_chebyshev_functions_of_the_reference(eng):
    # GELU (erf form), degree 119 on [-1,1] with mult 1/8   (:1330-1332, main.cpp: eval_gelu_function(.., -1, 1, 1/8, 119))
    mult = 1 / 8.0
    gelu = lambda v: 0.5 * (v / mult) * (1 + math.erf((v / mult) / 1.41421356237))
    cg = cheb_coeffs(gelu, -1, 1, 119)
    x = _x(eng, 4, -1, 1)
    got = eng.decrypt(eng.eval_chebyshev(eng.encrypt(x), cg, -1, 1))
    assert np.max(np.abs(got - cheb_eval(cg, x, -1, 1))) < 1e-4
    assert np.max(np.abs(got - np.array([gelu(v) for v in x]))) < 5e-2      # approximation error of the degree-119 fit
    # 1/x, degree 119 on [-1,128] is what main.cpp asks for (:1322-1324); on a benign interval [1, 16] the fit is accurate
    inv = lambda v: 1.0 / v
    ci = cheb_coeffs(inv, 1.0, 16.0, 119)
    y = _x(eng, 5, 1.0, 16.0)
    got = eng.decrypt(eng.eval_chebyshev(eng.encrypt(y), ci, 1.0, 16.0))
    assert np.max(np.abs(got - 1.0 / y)) < 1e-4


@pytest.mark.parametrize("degree", [47, 119, 300])
def test_paterson_stockmeyer_rounds_give_the_residues_of_the_dependent_order(fa, monkeypatch, degree):
    """polyeval.cpp cheb_recurse evaluates the products of the recursion p = q T_m + r in rounds (all products whose q operand
    is ready in ONE batched relinearisation).  Every node still computes mult(q, T_m) then add(., r): the exported residues
    must equal those of the one-product-at-a-time order (FHELIN_CHEB_ROUNDS=0) bit for bit — degrees of the bootstrap's
    cosine fit (47), GELU / 1/x (119) and tanh (300)."""
    rng = np.random.default_rng(degree)
    c = rng.uniform(-1, 1, degree + 1) / np.arange(1, degree + 2)
    outs = []
    for rounds in ("1", "0"):
        monkeypatch.setenv("FHELIN_CHEB_ROUNDS", rounds)
        e = fa.Engine("toy13", seed=5, n_q=14, n_p=4, dnum=4)
        try:
            e.keygen()
            e.gen_relin_key()
            x = np.random.default_rng(3).uniform(-1, 1, 1 << e.params.log_slots)
            ct = e.encrypt(x)
            e.stats(reset=True)
            out = e.eval_chebyshev(ct, c)
            outs.append((out.export(), out.info(), e.stats()["keyswitch"]))
            if rounds == "1":
                assert np.max(np.abs(e.decrypt(out) - cheb_eval(c, x, -1, 1))) < 1e-4
        finally:
            e.close()
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][0], outs[1][0])


def test_chebyshev_batch_entry_point_matches_single_calls(eng):
    """fhelin_eval_chebyshev_batch: n ciphertexts through one series together == n single calls, residue for residue"""
    rng = np.random.default_rng(8)
    c = rng.uniform(-1, 1, 32) / np.arange(1, 33)
    cts = [eng.encrypt(_x(eng, 20 + i, -1, 1)) for i in range(3)]
    single = [eng.eval_chebyshev(ct, c).export() for ct in cts]
    batch = eng.eval_chebyshev_batch(cts, c)
    for s, b in zip(single, batch):
        assert np.array_equal(s, b.export())
