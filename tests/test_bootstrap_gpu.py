"""CKKS bootstrapping on the GPU, checked by decryption.
Stated tolerance: 2e-4 absolute for |m| <= 1 — the sine approximation of the modular reduction contributes
(2 pi 2^-c |mu|)^2 / 6 relative (c = 10: 6e-6), the degree-47 cosine fit ~1e-10 * 4^R * 2^c = 7e-6, CKKS noise
of ~25 multiplicative levels the rest."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(fa, log_slots):
    e = fa.Engine("boot12", seed=77, log_slots=log_slots)
    e.keygen()
    e.gen_relin_key()
    e.bootstrap_setup(3, 3, 1 << log_slots)
    return e


def _coeffs(m):
    """unscaled plaintext coefficients (mu_lo + i mu_hi) of the slot vector m: w = U^H m / n, U_jk = zeta_j^k"""
    n = len(m)
    j = np.arange(n)
    e5 = np.array([pow(5, int(t), 4 * n) for t in j])
    U = np.exp(2j * np.pi * np.outer(e5, np.arange(n)) / (4 * n))
    return U.conj().T @ m / n


@pytest.mark.parametrize("log_slots", [10, 11])     # 10: sparse packing (N/4 slots, SubSum path); 11: full packing
def test_bootstrap_refreshes_levels_and_keeps_message(fa, log_slots):
    eng = _engine(fa, log_slots)
    try:
        n = 1 << log_slots
        m = np.random.default_rng(3).uniform(-1, 1, n)
        ct = eng.encrypt(m, level=eng.n_q - 3)               # 3 limbs left: nearly exhausted
        assert ct.info()["ell"] == 3
        # stage 2: slots hold t_k / (q0 K) (bit-reversed order) with t = Delta' mu + q0 I
        a = eng.decrypt(eng.bootstrap_partial(ct, 2))
        frac = a * 28 - np.round(a * 28)
        w = _coeffs(m)
        if len(a) == 2 * n:                                   # sparse packing: [t_k | t_{k+n}] share one ciphertext (2n slots)
            assert np.max(np.abs(np.sort(frac[:n] * 2 ** 10) - np.sort(w.real))) < 1e-3
            assert np.max(np.abs(np.sort(frac[n:] * 2 ** 10) - np.sort(w.imag))) < 1e-3
        else:
            assert len(a) == n
            assert np.max(np.abs(np.sort(frac * 2 ** 10) - np.sort(w.real))) < 1e-3
        out = eng.bootstrap(ct)
        info = out.info()
        assert info["ell"] >= 4, info                         # strictly more limbs than before
        err = np.max(np.abs(eng.decrypt(out) - m))
        assert err < 2e-4, err
        # the refreshed ciphertext is usable: one more multiplication
        sq = eng.mult(out, out)
        assert np.max(np.abs(eng.decrypt(sq) - m * m)) < 1e-3
    finally:
        eng.close()


def test_bootstrap_requires_setup_and_two_limbs(fa):
    e = fa.Engine("boot12", seed=1)
    e.keygen()
    ct = e.encrypt(np.zeros(4), level=e.n_q - 2)
    with pytest.raises(fa.FhelinError) as ei:
        e.bootstrap(ct)
    assert ei.value.code == 4                                 # EvalBootstrapSetup has not been called
    e.close()


def test_async_heavy_ops_match_the_synchronous_path(fa, monkeypatch):
    """Heavy ops issued back to back run on alternating worker lanes and are joined when their results are used
    (capi_internal.h run_heavy).  The residues must not depend on that: the same call sequence with FHELIN_ASYNC=0
    (everything on the main stream) has to export bit-identical ciphertexts — chains on one ciphertext (polynomial then
    bootstrap), independent ciphertexts interleaved, results consumed in the opposite order, and an input freed by the
    host while its consumer may still be running."""
    cheb = [0.0, 1.0, 0.0, -0.25, 0.0, 0.05, 0.0, -0.01]

    def run(async_on):
        monkeypatch.setenv("FHELIN_ASYNC", "1" if async_on else "0")
        eng = _engine(fa, 10)
        try:
            rng = np.random.default_rng(11)
            ms = [rng.uniform(-0.5, 0.5, 1 << 10) for _ in range(3)]
            outs = []
            for m in ms:
                ct = eng.encrypt(m, level=eng.n_q - 6)
                p = eng.eval_chebyshev(ct, cheb, -1.0, 1.0)     # lane k
                del ct                                           # the host drops the input while lane k may still read it
                outs.append(eng.bootstrap(p))                   # same lane (chain), next ciphertext -> other lane
            s = eng.add(outs[2], outs[0])                       # consume in a different order than produced
            s = eng.add(s, outs[1])
            return [o.export() for o in outs] + [s.export()], [eng.decrypt(o) for o in outs], ms
        finally:
            eng.close()

    a, dec, ms = run(True)
    b, _, _ = run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    u = [np.polynomial.chebyshev.chebval(m, [cheb[0] / 2] + cheb[1:]) for m in ms]
    for d, w in zip(dec, u):
        assert np.max(np.abs(d - w)) < 5e-4


def test_packed_and_two_ciphertext_modular_reduction_agree(fa, monkeypatch):
    """sparse packing: the real and imaginary coefficient halves go through EvalMod in ONE ciphertext (last CoeffsToSlots stage
    written over 2n slots as [d | -i d], first SlotsToCoeffs stage reading [L | R]); FHELIN_BOOT_PACKED=0 keeps the two-ciphertext
    form.  Same message out of both, same level, and fewer key switches in the packed form's EvalMod."""
    res = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("FHELIN_BOOT_PACKED", packed)
        eng = _engine(fa, 10)
        try:
            m = np.random.default_rng(3).uniform(-1, 1, 1 << 10)
            ct = eng.encrypt(m, level=eng.n_q - 3)
            out = eng.bootstrap(ct)
            res[packed] = (eng.decrypt(out), out.info()["ell"], out.info()["slots"])
        finally:
            eng.close()
    for packed in ("1", "0"):
        assert np.max(np.abs(res[packed][0] - m)) < 2e-4, packed
    assert res["1"][1:] == res["0"][1:]


def test_grouped_inner_sums_give_the_same_residues(fa, monkeypatch):
    """the inner sums of all giant steps of a linear stage come out of one pass over the rotated ciphertexts
    (Evaluator::dot_plain_groups, kernel ew_dot_groups): exact integer sums, so the bootstrapped ciphertext must equal the one
    computed with one pass per giant step (FHELIN_DOT_GROUPS=0) residue for residue"""
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("FHELIN_DOT_GROUPS", on)
        eng = _engine(fa, 10)
        try:
            m = np.random.default_rng(3).uniform(-1, 1, 1 << 10)
            ct = eng.encrypt(m, level=eng.n_q - 3)               # same seed, same call sequence: the same keys and ciphertext
            outs.append(eng.bootstrap(ct).export())
        finally:
            eng.close()
    assert np.array_equal(outs[0], outs[1])
