"""K2-K8 parity: dyadic ops, automorphism, rescale, hybrid key switching (rotate, mult+relin) on the GPU
vs the CPU oracle, bit-exact on identical (seeded, uniform) residues and key material, through the C-ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ct(orc, eng, seed, ell, npoly=2):
    return np.stack([orc.uniform_residues(seed + 1000 * p, eng.q[:ell], eng.N) for p in range(npoly)])


def _evk(orc, eng, seed):
    """uniform 'key' [dnum][2][L+1+k][N]: parity of the residue functions does not need a real key"""
    d = eng.dnum_digits
    k = np.stack([orc.uniform_residues(seed + 50 * j, eng.moduli, eng.N) for j in range(2 * d)])
    return k.reshape(d, 2, eng.n_limbs, eng.N)


@pytest.mark.parametrize("preset,ells", [("toy", [6, 5, 3, 2]), ("toy13", [7, 4]), ("bench", [24, 13]), ("reference", [28])])
def test_rescale_bit_exact(engine_factory, orc, preset, ells):
    eng = engine_factory(preset)
    for ell in ells:
        x = _ct(orc, eng, 100 + ell, ell)
        got = eng.raw_rescale(eng.ct_import(x)).export()
        want = orc.rescale(x, eng.q[:ell], eng.psi_q[:ell])
        assert got.shape == want.shape == (2, ell - 1, eng.N)
        assert np.array_equal(got, want)


def test_rescale_three_components_and_last_limb(engine_factory, orc):
    eng = engine_factory("toy")
    x = _ct(orc, eng, 7, 4, npoly=3)
    got = eng.raw_rescale(eng.ct_import(x)).export()
    assert np.array_equal(got, orc.rescale(x, eng.q[:4], eng.psi_q[:4]))
    one = eng.ct_import(_ct(orc, eng, 8, 1))
    with pytest.raises(Exception):
        eng.raw_rescale(one)          # nothing left to drop: must fail, not wrap around


@pytest.mark.parametrize("preset,ells,rots", [
    ("toy", [6, 5, 4, 2, 1], [1, -1, 5]),          # dnum=3, alpha=2: full, partial and single digit levels
    ("toy13", [7, 3], [128, -64]),
    ("bench", [24, 8], [1, 128]),                  # BASELINE size: N=2^16, 24 limbs, k=6, alpha=6
    ("reference", [28], [-1]),                     # reference parameters: N=2^15, 28+7 limbs, alpha=7
    ("deep", [30, 9], [128]),                      # BASELINE config 5: N=2^17, 30+8 limbs, alpha=8
])
def test_rotate_bit_exact(engine_factory, orc, preset, ells, rots):
    eng = engine_factory(preset)
    for r in rots:
        evk = _evk(orc, eng, 900 + abs(r))
        eng.key_import(1, r, evk)
        g = orc.galois(eng.log_n, r)
        for ell in ells:
            x = _ct(orc, eng, 300 + ell, ell)
            got = eng.raw_rotate(eng.ct_import(x), r).export()
            want = orc.rotate(x, evk, g, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
            assert np.array_equal(got, want), (preset, r, ell)


@pytest.mark.parametrize("preset,ells", [("toy", [6, 3, 1]), ("toy13", [5]), ("bench", [24])])
def test_mult_relin_bit_exact(engine_factory, orc, preset, ells):
    eng = engine_factory(preset)
    evk = _evk(orc, eng, 4242)
    eng.key_import(0, 0, evk)
    for ell in ells:
        a, b = _ct(orc, eng, 500 + ell, ell), _ct(orc, eng, 600 + ell, ell)
        got = eng.raw_mult_relin(eng.ct_import(a), eng.ct_import(b)).export()
        want = orc.mult_relin(a, b, evk, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
        assert np.array_equal(got, want)


def test_key_roundtrip_and_missing_key(engine_factory, orc, fa):
    eng = engine_factory("toy")
    evk = _evk(orc, eng, 31337)
    eng.key_import(1, 7, evk)
    assert np.array_equal(eng.key_export(1, 7), evk)
    x = eng.ct_import(_ct(orc, eng, 1, 3))
    with pytest.raises(fa.FhelinError) as ei:
        eng.raw_rotate(x, 11)                         # no key for this index
    assert ei.value.code == 5


def test_add_sub_dyadic_bit_exact(engine_factory, orc):
    eng = engine_factory("toy13")
    ell = 5
    a, b = _ct(orc, eng, 21, ell), _ct(orc, eng, 22, ell)
    ca, cb = eng.ct_import(a), eng.ct_import(b)
    q = eng.q[:ell]
    assert np.array_equal(eng.add(ca, cb).export(), np.stack([orc.add(a[p], b[p], q) for p in range(2)]))
    assert np.array_equal(eng.sub(ca, cb).export(), np.stack([orc.sub(a[p], b[p], q) for p in range(2)]))
    zero = np.zeros_like(a)
    assert np.array_equal(eng.negate(ca).export(), np.stack([orc.sub(zero[p], a[p], q) for p in range(2)]))


def test_rotation_composition_full_size(engine_factory, orc):
    """size-independent property at BASELINE size: with the SAME key material for r, rotating a ciphertext
    whose c1 = 0 is exactly the automorphism of c0 (no key-switch noise enters): rot_a(rot_b(x)) = rot_{a+b}(x)."""
    eng = engine_factory("bench")
    ell = 12
    x = _ct(orc, eng, 77, ell)
    x[1] = 0
    for r in (1, 2, 3):
        eng.key_import(1, r, _evk(orc, eng, 5000 + r))
    c = eng.ct_import(x)
    r12 = eng.raw_rotate(eng.raw_rotate(c, 1), 2).export()
    r3 = eng.raw_rotate(c, 3).export()
    assert np.array_equal(r12, r3)
    assert not r3[1].any()


@pytest.mark.parametrize("preset,ell,rots", [("toy13", 5, [1, -2, 3, 0, 64]), ("bench", 17, list(range(1, 19)))])
def test_hoisted_and_per_row_rotations_bit_exact(engine_factory, orc, preset, ell, rots):
    """rotate_many (one ModUp shared by all indices) and rotate_each (different inputs, different keys, one batched key
    switch) against the oracle's plain rotation: hoisting must not change a single residue.  18 indices at N=2^16 also
    cross the 16-row chunk limit of the kernel arguments."""
    eng = engine_factory(preset)
    evks = {}
    for r in rots:
        if r % (eng.N // 2) == 0:
            continue
        evks[r] = _evk(orc, eng, 7000 + 13 * r)
        eng.key_import(1, r, evks[r])
    x = _ct(orc, eng, 4242, ell)
    cx = eng.ct_import(x)
    want = {r: (x if r not in evks else orc.rotate(x, evks[r], orc.galois(eng.log_n, r), eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p))
            for r in rots}
    got = eng.rotate_many(cx, rots)
    for r, g in zip(rots, got):
        assert np.array_equal(g.export(), want[r]), ("rotate_many", preset, r)
    ys = [_ct(orc, eng, 4300 + i, ell) for i in range(len(rots))]
    got = eng.rotate_each([eng.ct_import(y) for y in ys], rots)
    for r, y, g in zip(rots, ys, got):
        w = y if r not in evks else orc.rotate(y, evks[r], orc.galois(eng.log_n, r), eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
        assert np.array_equal(g.export(), w), ("rotate_each", preset, r)
