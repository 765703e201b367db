"""K1 parity: HIP negacyclic NTT/INTT vs the CPU oracle, bit-exact, through the C-ABI (fhelin_ntt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_limbs(orc, eng, seed, npoly=2, with_p=False):
    mods = list(eng.q) + (list(eng.p) if with_p else [])
    return np.stack([orc.uniform_residues(seed + 1000 * p, mods, eng.N) for p in range(npoly)])


@pytest.mark.parametrize("preset", ["toy", "toy13", "reference", "bench", "deep"])
def test_ntt_forward_inverse_bit_exact(engine_factory, orc, preset):
    eng = engine_factory(preset)
    x = _rand_limbs(orc, eng, 0x5EED0001)            # [2][n_q][N] — one ciphertext's worth
    buf = eng.upload(x)
    eng.ntt(buf, 2 * eng.n_q)
    got = buf.download(x.shape)
    want = orc.ntt_batch(x, eng.q, eng.psi_q)
    assert np.array_equal(got, want)
    eng.ntt(buf, 2 * eng.n_q, inverse=True)
    back = buf.download(x.shape)
    assert np.array_equal(back, x)                    # INTT(NTT(x)) = x, size-independent property
    buf.free()


@pytest.mark.parametrize("preset", ["toy", "bench"])
def test_ntt_special_limbs_and_inverse_alone(engine_factory, orc, preset):
    eng = engine_factory(preset)
    xp = np.stack([orc.uniform_residues(77 + i, eng.p, eng.N) for i in range(3)])   # [3][k][N] over P
    buf = eng.upload(xp)
    eng.ntt(buf, 3 * eng.n_p, limb_first=eng.n_q, limb_count=eng.n_p, inverse=True)
    got = buf.download(xp.shape)
    assert np.array_equal(got, orc.ntt_batch(xp, eng.p, eng.psi_p, inverse=True))
    buf.free()


def test_ntt_edge_inputs(engine_factory, orc):
    eng = engine_factory("toy13")
    n, nq = eng.N, eng.n_q
    x = np.zeros((4, nq, n), dtype=np.uint64)
    x[1] = (eng.q - np.uint64(1))[:, None]            # all residues maximal
    x[2, :, 0] = 1                                    # the constant 1
    x[3, :, 1] = 1                                    # the monomial X
    buf = eng.upload(x)
    eng.ntt(buf, 4 * nq)
    got = buf.download(x.shape)
    assert not got[0].any()
    assert np.array_equal(got, orc.ntt_batch(x, eng.q, eng.psi_q))
    assert np.all(got[2] == 1)
    assert np.array_equal(got[3, :, 0], eng.psi_q)    # X evaluates to psi at slot 0
    buf.free()


def test_ntt_single_vector_and_ragged_batch(engine_factory, orc):
    """nvec not a multiple of the limb count (ragged): vector v still uses limb v % count."""
    eng = engine_factory("toy")
    nv = eng.n_q + 2
    mods = [eng.q[v % eng.n_q] for v in range(nv)]
    x = orc.uniform_residues(5, mods, eng.N)
    buf = eng.upload(x)
    eng.ntt(buf, nv)
    assert np.array_equal(buf.download(x.shape), orc.ntt_batch(x, eng.q, eng.psi_q))
    eng.ntt(buf, 0)                                   # empty batch is a no-op
    buf.free()
    one = orc.uniform_residues(6, [eng.q[2]], eng.N)
    b1 = eng.upload(one)
    eng.ntt(b1, 1, limb_first=2, limb_count=1)
    assert np.array_equal(b1.download(one.shape)[0], orc.ntt_forward(one[0], eng.q[2], eng.psi_q[2]))
    b1.free()


def test_ntt_linearity_full_size(engine_factory, orc):
    """NTT(a + b) = NTT(a) + NTT(b) at BASELINE size (N=2^16, 24 limbs), checked without the oracle's NTT."""
    eng = engine_factory("bench")
    a = _rand_limbs(orc, eng, 11, npoly=1)[0]
    b = _rand_limbs(orc, eng, 12, npoly=1)[0]
    s = orc.add(a, b, eng.q)
    bufs = [eng.upload(v) for v in (a, b, s)]
    for bf in bufs:
        eng.ntt(bf, eng.n_q)
    fa_, fb, fs = [bf.download(a.shape) for bf in bufs]
    assert np.array_equal(orc.add(fa_, fb, eng.q), fs)
    for bf in bufs:
        bf.free()


@pytest.mark.parametrize("preset", ["bench", "deep"])
def test_ntt_extreme_patterns_full_size(engine_factory, orc, preset):
    """Inputs that drive the lazy (no conditional subtraction) butterflies towards their value bounds: all residues
    maximal, and maximal/zero alternations at the strides of the first, the tile-boundary and the last stage.
    Forward and inverse, every limb kind (52-bit lazy path, 55-bit and 60-bit classic path), bit-exact vs the oracle."""
    eng = engine_factory(preset)
    n = eng.N
    idx = np.arange(n)
    pats = [np.ones(n, dtype=bool)] + [((idx // s) & 1).astype(bool) for s in (1, 16, 256, 4096, n // 2)]
    for mods, psis, first in ((eng.q, eng.psi_q, 0), (eng.p, eng.psi_p, eng.n_q)):
        mods = np.asarray(mods, dtype=np.uint64)
        x = np.zeros((len(pats), len(mods), n), dtype=np.uint64)
        for i, pat in enumerate(pats):
            x[i] = np.where(pat[None, :], (mods - np.uint64(1))[:, None], np.uint64(0))
        for inverse in (False, True):
            buf = eng.upload(x)
            eng.ntt(buf, x.shape[0] * x.shape[1], limb_first=first, limb_count=len(mods), inverse=inverse)
            got = buf.download(x.shape)
            want = orc.ntt_batch(x, mods, psis, inverse=inverse)
            assert np.array_equal(got, want), f"first={first} inverse={inverse}"
            buf.free()
