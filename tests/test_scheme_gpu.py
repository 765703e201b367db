"""Scheme-level correctness by decryption: real keys from the engine's keygen,
GPU evaluation, decrypt + decode, compared with the plaintext computation to CKKS precision.
Tolerances are stated per test: fresh encryptions carry ~2^-40 relative noise at Delta = 2^52."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROT = [1, 2, 3, -1, 128, -64]


@pytest.fixture(scope="module")
def keyed(fa):
    out = {}
    for preset in ("toy13", "bench"):
        e = fa.Engine(preset, seed=1234)
        e.keygen()
        e.gen_relin_key()
        e.gen_rotation_keys(ROT)
        out[preset] = e
    yield out
    for e in out.values():
        e.close()


def _vec(eng, seed, lo=-1.0, hi=1.0):
    n = 1 << eng.params.log_slots
    return np.random.default_rng(seed).uniform(lo, hi, size=n)


@pytest.mark.parametrize("preset", ["toy13", "bench"])
def test_encrypt_decrypt_roundtrip(keyed, preset):
    eng = keyed[preset]
    x = _vec(eng, 1)
    for level in (0, 2, eng.n_q - 2, eng.n_q - 1):
        ct = eng.encrypt(x, level=level)
        assert ct.level == level and ct.info()["ell"] == eng.n_q - level
        y = eng.decrypt(ct)
        assert np.max(np.abs(y - x)) < 1e-8, (preset, level)


def test_encode_short_vector_and_sparse_slots(keyed):
    eng = keyed["toy13"]
    x = np.array([1.5, -2.25, 3.0])
    y = eng.decrypt(eng.encrypt(x))            # padded with zeros up to the batch size
    assert np.max(np.abs(y[:3] - x)) < 1e-8 and np.max(np.abs(y[3:])) < 1e-8
    s = 64                                      # sparse packing: 64 slots in a 2^13 ring
    z = np.linspace(-1, 1, s)
    ct = eng.encrypt(z, slots=s)
    assert np.max(np.abs(eng.decrypt(ct, s) - z)) < 1e-8


@pytest.mark.parametrize("preset", ["toy13", "bench"])
def test_rotation_semantics(keyed, preset):
    """EvalRotate(c, +i) is a LEFT shift of the slots: out[s] = in[s+i]."""
    eng = keyed[preset]
    x = _vec(eng, 2)
    ct = eng.encrypt(x)
    for r in ROT:
        y = eng.decrypt(eng.rotate(ct, r))
        assert np.max(np.abs(y - np.roll(x, -r))) < 1e-7, (preset, r)
    assert np.max(np.abs(eng.decrypt(eng.rotate(ct, 0)) - x)) < 1e-8


@pytest.mark.parametrize("preset", ["toy13", "bench"])
def test_mult_plain_and_cipher(keyed, preset):
    eng = keyed[preset]
    x, y = _vec(eng, 3), _vec(eng, 4)
    cx, cy = eng.encrypt(x), eng.encrypt(y)
    p = eng.mult(cx, eng.encode(y))
    assert p.info()["deg"] == 2
    assert np.max(np.abs(eng.decrypt(p) - x * y)) < 1e-7
    m = eng.mult(cx, cy)
    assert m.info()["deg"] == 2 and m.info()["npoly"] == 2
    assert np.max(np.abs(eng.decrypt(m) - x * y)) < 1e-7
    r = eng.rescale(m)
    assert r.level == 1 and r.info()["deg"] == 1
    assert np.max(np.abs(eng.decrypt(r) - x * y)) < 1e-7


def test_flexibleauto_depth_chain(keyed):
    """x^(2^d) by repeated squaring: each EvalMult first rescales a degree-2 input (FLEXIBLEAUTO)."""
    eng = keyed["toy13"]
    x = _vec(eng, 5, 0.5, 1.0)
    c = eng.encrypt(x)
    want = x.copy()
    for d in range(4):
        c = eng.mult(c, c)
        want = want * want
        assert c.level == d and c.info()["deg"] == 2      # rescale is lazy: level rises on the NEXT mult
        assert np.max(np.abs(eng.decrypt(c) - want)) < 1e-6
    # the real scaling factor tracked per level stays within 2^-18 of 2^52 (prime chain property)
    assert abs(eng.rescale(c).info()["scale"] / 2.0 ** 52 - 1) < 2.0 ** -18


def test_add_across_levels_and_degrees(keyed):
    eng = keyed["toy13"]
    x, y, z = _vec(eng, 6), _vec(eng, 7), _vec(eng, 8)
    cx, cy, cz = eng.encrypt(x), eng.encrypt(y), eng.encrypt(z)
    xy = eng.mult(cx, cy)                        # level 0, degree 2
    s1 = eng.add(xy, cz)                         # degree 1 operand is lifted to degree 2
    assert s1.info()["deg"] == 2 and np.max(np.abs(eng.decrypt(s1) - (x * y + z))) < 1e-7
    xyz = eng.mult(xy, cz)                       # rescales xy -> level 1, cz adjusted to level 1
    assert xyz.level == 1
    s2 = eng.add(xyz, cx)                        # level 0 / degree 1 operand brought to level 1 / degree 2
    assert np.max(np.abs(eng.decrypt(s2) - (x * y * z + x))) < 1e-6
    s3 = eng.add(eng.rescale(xyz), cy)           # level 2 / degree 1  +  level 0 / degree 1
    assert s3.level == 2 and np.max(np.abs(eng.decrypt(s3) - (x * y * z + y))) < 1e-6
    d = eng.sub(cx, cy)
    assert np.max(np.abs(eng.decrypt(d) - (x - y))) < 1e-8
    b = eng.add(xy, eng.encode(z))               # plaintext encoded at the ciphertext's own scale (degree 2)
    assert np.max(np.abs(eng.decrypt(b) - (x * y + z))) < 1e-7


def test_rotsum_pattern(keyed):
    """The reference's hot loop (src/FHEController.cpp:829-837): r <- r + rot(r, 128*2^i) — here 128, 256, ... on
    a 4096-slot toy ring: after log2(32) steps slot j holds the sum over j mod 128."""
    eng = keyed["toy13"]
    n = 1 << eng.params.log_slots
    x = _vec(eng, 9)
    steps = [128 << i for i in range(int(np.log2(n // 128)))]
    eng.gen_rotation_keys(steps)
    c = eng.encrypt(x)
    for s in steps:
        c = eng.add(c, eng.rotate(c, s))
    want = np.tile(x.reshape(-1, 128).sum(axis=0), n // 128)
    assert np.max(np.abs(eng.decrypt(c) - want)) < 1e-6


@pytest.mark.parametrize("preset", ["toy13", "bench"])
def test_merged_rotation_sum(keyed, preset):
    """x + rot(x, 1) + rot(x, 2) + rot(x, 3) with one ModUp and one ModDown (rotated inner products accumulated in the
    extended basis) — two steps of a rotate-and-sum tree — for a batch of rows, at a full and a partial-digit level,
    also on a degree-2 input.  Tolerance: CKKS noise, 1e-7."""
    eng = keyed[preset]
    ns = 1 << eng.params.log_slots
    for level in (0, eng.n_q - 3):
        xs = [_vec(eng, 50 + i) for i in range(3)]
        cts = [eng.encrypt(x, level=level) for x in xs]
        outs = eng.rotate_sum(cts, [1, 2, 3])
        for x, o in zip(xs, outs):
            want = x + np.roll(x, -1) + np.roll(x, -2) + np.roll(x, -3)
            assert o.info()["ell"] == eng.n_q - level
            assert np.max(np.abs(eng.decrypt(o) - want)) < 1e-7, (preset, level)
    x = _vec(eng, 77)
    m = np.zeros(ns)
    m[::2] = 0.5
    prod = eng.mult(eng.encrypt(x), eng.encode(m, 0))          # degree 2
    got = eng.decrypt(eng.rotate_sum([prod], [128, -64])[0])
    y = x * m
    assert np.max(np.abs(got - (y + np.roll(y, -128) + np.roll(y, 64)))) < 1e-7
    with pytest.raises(Exception):
        eng.rotate_sum([prod], [5])                            # no key for rotation 5
