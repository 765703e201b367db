"""Known-answer tests of the encoder oracle itself (oracle/encode_oracle.py: CKKS encoding by definition, mpmath) - CPU.
The GPU encoder is compared with it in tests/test_encode_oracle_gpu.py."""
import numpy as np

from oracle import encode_oracle as eo


def test_constant_and_monomial_vectors_encode_to_the_obvious_polynomials():
    n, N, D = 8, 64, 1 << 30                                   # sparse packing: gap N/2n = 4
    allj = list(range(N))
    c = eo.exact_coefficients([1.0] * n, n, N, D, allj)        # z = (1, ..., 1)  ->  m = Delta
    assert c[0] == D and all(v == 0 for j, v in c.items() if j)
    rot = eo.rotation_group(n)
    w = [np.exp(2j * np.pi * g / (4 * n)) for g in rot]         # z_k = omega^(5^k)  ->  m = Delta * Y,  Y = X^gap
    c = eo.exact_coefficients(w, n, N, D, allj)
    assert c[4] == D and all(abs(v) <= 1 for j, v in c.items() if j != 4)
    c = eo.exact_coefficients([1j] * n, n, N, D, allj)          # z = (i, ..., i)  ->  m = Delta * X^(N/2)
    assert c[N // 2] == D and all(v == 0 for j, v in c.items() if j != N // 2)


def test_embedding_inverts_the_inverse_embedding():
    rng = np.random.default_rng(5)
    for n, N in ((16, 32), (16, 128), (64, 512)):               # full and sparse packing
        z = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
        D = 1 << 45
        c = eo.exact_coefficients(z, n, N, D, list(range(N)))
        coeffs = [c[j] for j in range(N)]
        gap = (N // 2) // n
        assert all(v == 0 for j, v in c.items() if (j % (N // 2)) % gap)          # the subring
        back = eo.evaluate_slots(coeffs, n, N, list(range(n)))
        err = max(abs(complex(b) / D - zz) for b, zz in zip(back, z))
        assert err < 2 * n / D, (n, N, err)                                       # 2n roundings of <= 1/2


def test_slot_rotation_is_the_galois_automorphism_of_the_evaluator_oracle(orc):
    """encode(rot(z, r)) = sigma_(5^r)(encode(z)) EXACTLY (rounding commutes with a signed permutation of the coefficients): ties the
    slot order of the encoder oracle to the rotation convention of the residue oracle (orc.galois / orc.automorph_coeff: index r <-> 5^r,
    a LEFT shift of the slots)"""
    log_n = 7
    N, n, D = 1 << log_n, 1 << (log_n - 1), 1 << 40
    rng = np.random.default_rng(9)
    z = rng.uniform(-1, 1, n)
    qq = (1 << 61) - 1                                          # any odd modulus > 2 Delta max|coef| does for a signed permutation
    base = eo.exact_coefficients(z, n, N, D, list(range(N)))
    a = np.array([[base[j] % qq for j in range(N)]], dtype=np.uint64)
    for r in (1, 3, -2):
        zr = np.roll(z, -r)                                     # rot(z, r): out[s] = z[s + r]
        want = eo.exact_coefficients(zr, n, N, D, list(range(N)))
        got = orc.automorph_coeff(a, orc.galois(log_n, r), qq)
        assert [int(v) for v in got[0]] == [want[j] % qq for j in range(N)], r
