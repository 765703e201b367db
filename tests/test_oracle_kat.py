"""Known-answer tests that pin the CPU oracle without any library: the oracle is 'parity unpinned'
against OpenFHE, so it is anchored on mathematics instead."""
import numpy as np
import pytest

Q60 = None


def _chain(orc, log_n=12, n_q=4, n_p=2):
    q, p = orc.prime_chain(log_n, n_q, 55, 52, n_p, 60)
    psi_q = np.array([orc.min_root(x, 2 << log_n) for x in q], dtype=np.uint64)
    psi_p = np.array([orc.min_root(x, 2 << log_n) for x in p], dtype=np.uint64)
    return q, p, psi_q, psi_p


def test_prime_chain_properties(orc):
    for log_n, n_q, n_p in [(12, 6, 2), (15, 28, 7), (16, 24, 6), (17, 30, 8)]:
        q, p = orc.prime_chain(log_n, n_q, 55, 52, n_p, 60)
        allp = [int(x) for x in q] + [int(x) for x in p]
        assert len(set(allp)) == len(allp)
        m = 2 << log_n
        for x in allp:
            assert orc.is_prime(x) and x % m == 1
        assert int(q[0]).bit_length() == 55
        assert all(int(x).bit_length() in (52, 53) for x in q[1:])
        assert all(int(x).bit_length() == 60 for x in p)
        # FLEXIBLEAUTO property: the real scaling factor stays within 2^-20 relative of 2^52 at every level
        sf = float(q[-1])
        for k in range(n_q - 2):
            sf = sf * sf / float(q[n_q - 1 - k])
            assert abs(sf / 2.0 ** 52 - 1.0) < 2.0 ** -18


def test_known_small_primes(orc):
    assert orc.is_prime(2 ** 61 - 1) and not orc.is_prime(2 ** 61 + 1)
    assert orc.is_prime(0xFFFFFFFF00000001)  # Goldilocks
    assert not orc.is_prime(3215031751)  # strong pseudoprime to bases 2,3,5,7


def test_min_root_is_primitive_and_minimal(orc):
    q, _, _, _ = _chain(orc, 12, 3, 1)
    for x in q:
        x = int(x)
        r = orc.min_root(x, 8192)
        assert pow(r, 4096, x) == x - 1
        # no smaller primitive root: brute force over odd powers
        roots = set()
        cur, r2 = r, r * r % x
        for _ in range(4096):
            roots.add(cur)
            cur = cur * r2 % x
        assert min(roots) == r and len(roots) == 4096


def test_ntt_matches_definition(orc):
    q, _, psi, _ = _chain(orc, 12, 3, 1)
    rng = np.random.default_rng(1)
    for ql, ps in zip(q, psi):
        a = rng.integers(0, int(ql), size=4096, dtype=np.uint64)
        assert np.array_equal(orc.ntt_forward(a, ql, ps), orc.ntt_naive(a, ql, ps))


def test_ntt_roundtrip_all_ring_sizes(orc):
    rng = np.random.default_rng(2)
    for log_n in (12, 13, 15, 16, 17):
        q, p = orc.prime_chain(log_n, 3, 55, 52, 1, 60)
        for ql in list(q) + list(p):
            ps = orc.min_root(ql, 2 << log_n)
            a = rng.integers(0, int(ql), size=1 << log_n, dtype=np.uint64)
            f = orc.ntt_forward(a, ql, ps)
            assert f.max() < ql
            assert np.array_equal(orc.ntt_inverse(f, ql, ps), a)


def test_ntt_edge_vectors(orc):
    q, _, psi, _ = _chain(orc, 12, 2, 1)
    ql, ps = int(q[1]), int(psi[1])
    zero = np.zeros(4096, dtype=np.uint64)
    assert not orc.ntt_forward(zero, ql, ps).any()
    one = zero.copy(); one[0] = 1
    assert np.all(orc.ntt_forward(one, ql, ps) == 1)  # constant polynomial evaluates to 1 everywhere
    mx = np.full(4096, ql - 1, dtype=np.uint64)
    assert np.array_equal(orc.ntt_inverse(orc.ntt_forward(mx, ql, ps), ql, ps), mx)
    # X evaluates to psi^(2 br(j)+1): slot 0 holds psi itself
    x = zero.copy(); x[1] = 1
    assert int(orc.ntt_forward(x, ql, ps)[0]) == ps


def test_convolution_theorem(orc):
    q, _, psi, _ = _chain(orc, 12, 2, 1)
    rng = np.random.default_rng(3)
    ql, ps = q[0], psi[0]
    a = rng.integers(0, int(ql), size=4096, dtype=np.uint64)
    b = rng.integers(0, int(ql), size=4096, dtype=np.uint64)
    fa_, fb = orc.ntt_forward(a, ql, ps), orc.ntt_forward(b, ql, ps)
    prod = orc.mul(fa_[None], fb[None], [ql])[0]
    assert np.array_equal(orc.ntt_inverse(prod, ql, ps), orc.negacyclic_mul_naive(a, b, ql))


def test_automorphism_domains_agree(orc):
    q, _, psi, _ = _chain(orc, 12, 2, 1)
    rng = np.random.default_rng(4)
    ql, ps = q[1], psi[1]
    a = rng.integers(0, int(ql), size=4096, dtype=np.uint64)
    for r in (1, 2, 128, -1, -64, 1023):
        g = orc.galois(12, r)
        assert g % 2 == 1
        lhs = orc.automorph_ntt(orc.ntt_forward(a, ql, ps), g)
        rhs = orc.ntt_forward(orc.automorph_coeff(a, g, ql), ql, ps)
        assert np.array_equal(lhs, rhs)
    assert orc.galois(12, 1) * orc.galois(12, -1) % 8192 == 1


def _crt2(x0, x1, q0, q1):
    """centred CRT lift of residues mod q0, q1 -> python ints"""
    inv = pow(q0, -1, q1)
    out = []
    Q = q0 * q1
    for a, b in zip(x0.tolist(), x1.tolist()):
        v = a + q0 * (((b - a) * inv) % q1)
        out.append(v - Q if v > Q // 2 else v)
    return out


def test_rescale_divides_by_last_prime(orc):
    log_n, ell = 12, 3
    q, _, psi, _ = _chain(orc, log_n, ell, 1)
    rng = np.random.default_rng(5)
    qs = [int(x) for x in q]
    # a polynomial with small (|c| < 2^100) integer coefficients, known exactly
    coeffs = [int(rng.integers(-2 ** 62, 2 ** 62)) * int(rng.integers(1, 2 ** 38)) for _ in range(1 << log_n)]
    limbs = np.array([[c % m for c in coeffs] for m in qs], dtype=np.uint64)
    ct = orc.ntt_batch(limbs, q, psi)[None]
    out = orc.rescale(ct, q, psi)[0]
    back = orc.ntt_batch(out, q[:-1], psi[:-1], inverse=True)
    got = _crt2(back[0], back[1], qs[0], qs[1])
    for c, g in zip(coeffs, got):
        # exact division of (c - [c]_ql centred) by ql: |g - c/ql| <= 1/2
        assert abs(g * qs[-1] - c) <= qs[-1] // 2 + 1


def _toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s_from, s_to, rng):
    """Hybrid key-switching key from secret s_from to s_to (both small ternary coefficient vectors)."""
    n = 1 << log_n
    L1, k = len(q), len(p)
    dnum = -(-L1 // alpha)
    mods = [int(x) for x in q] + [int(x) for x in p]
    psis = list(psi_q) + list(psi_p)
    P = 1
    for x in p:
        P *= int(x)
    Q = 1
    for x in q:
        Q *= int(x)

    def to_ntt(coeffs):
        limbs = np.array([[c % m for c in coeffs] for m in mods], dtype=np.uint64)
        return orc.ntt_batch(limbs, mods, psis)

    s_to_ntt = to_ntt(s_to)
    s_from_ntt = to_ntt(s_from)
    evk = np.zeros((dnum, 2, L1 + k, n), dtype=np.uint64)
    for j in range(dnum):
        Qj = 1
        for x in q[j * alpha:(j + 1) * alpha]:
            Qj *= int(x)
        Qhat = Q // Qj
        factor = P * Qhat * pow(Qhat, -1, Qj)
        a = np.array([rng.integers(0, m, size=n, dtype=np.uint64) for m in mods])
        e = to_ntt([int(v) for v in rng.integers(-3, 4, size=n)])
        fs = orc.mul_scalar(s_from_ntt, np.array([factor % m for m in mods], dtype=np.uint64), mods)
        b = orc.add(orc.sub(e, orc.mul(a, s_to_ntt, mods), mods), fs, mods)
        evk[j, 0], evk[j, 1] = b, a
    return evk, s_to_ntt


@pytest.mark.parametrize("ell", [6, 4, 3])
def test_keyswitch_correct_by_decryption(orc, ell):
    """KeySwitch(c, evk(s'->s)) decrypts under s to c*s' up to small noise (hybrid, dnum=3, partial digits)."""
    log_n, L1, k, alpha = 12, 6, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(6)
    n = 1 << log_n
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    s2 = [int(v) for v in rng.integers(-1, 2, size=n)]
    evk, s_ntt_all = _toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s2, s, rng)
    ql = q[:ell]
    c = np.array([rng.integers(0, int(m), size=n, dtype=np.uint64) for m in ql])
    ks = orc.keyswitch(c, evk, alpha, q, p, psi_q, psi_p)
    s_ntt = s_ntt_all[:ell]
    s2_ntt = orc.ntt_batch(np.array([[v % int(m) for v in s2] for m in ql], dtype=np.uint64), ql, psi_q[:ell])
    lhs = orc.add(ks[0], orc.mul(ks[1], s_ntt, ql), ql)          # <ks, (1, s)>
    rhs = orc.mul(c, s2_ntt, ql)                                 # c * s'
    diff = orc.ntt_batch(orc.sub(lhs, rhs, ql), ql, psi_q[:ell], inverse=True)
    err = _crt2(diff[0], diff[1], int(ql[0]), int(ql[1]))
    assert max(abs(e) for e in err) < 2 ** 40   # noise << q (2^52): ~ N * dnum * q_digit * e / P + rounding


def test_rotate_matches_definition(orc):
    """orc_rotate == automorphism of (c0 + ks0, ks1) and decrypts to the rotated message polynomial."""
    log_n, L1, k, alpha, ell = 12, 4, 2, 2, 4
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(7)
    n = 1 << log_n
    g = orc.galois(log_n, 3)
    ginv = pow(g, -1, 2 * n)
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    # key from sigma_{g^-1}... we need KS from s to sigma_g^{-1}(s) so that after applying sigma_g the secret is s
    s_coeff = np.array([v % int(q[0]) for v in s], dtype=np.uint64)
    s_perm = orc.automorph_coeff(s_coeff, ginv, int(q[0]))
    s_perm_int = [int(v) if v < int(q[0]) // 2 else int(v) - int(q[0]) for v in s_perm]
    evk, _ = _toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, s_perm_int, rng)
    ct = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)])
    out = orc.rotate(ct, evk, g, alpha, q, p, psi_q, psi_p)
    s_ntt = orc.ntt_batch(np.array([[v % int(m) for v in s] for m in q], dtype=np.uint64), q, psi_q)
    dec_in = orc.add(ct[0], orc.mul(ct[1], s_ntt, q), q)
    dec_out = orc.add(out[0], orc.mul(out[1], s_ntt, q), q)
    expect = np.array([orc.automorph_ntt(x, g) for x in dec_in])
    diff = orc.ntt_batch(orc.sub(dec_out, expect, q), q, psi_q, inverse=True)
    err = _crt2(diff[0], diff[1], int(q[0]), int(q[1]))
    assert max(abs(e) for e in err) < 2 ** 40


def test_mult_relin_by_decryption(orc):
    log_n, L1, k, alpha = 12, 4, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(8)
    n = 1 << log_n
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    s_q = np.array([[v % int(m) for v in s] for m in q], dtype=np.uint64)
    s_ntt = orc.ntt_batch(s_q, q, psi_q)
    s2_ntt = orc.mul(s_ntt, s_ntt, q)
    s2_coeff = orc.ntt_batch(s2_ntt, q, psi_q, inverse=True)
    s2 = _crt2(s2_coeff[0], s2_coeff[1], int(q[0]), int(q[1]))
    evk, _ = _toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s2, s, rng)
    a = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)])
    b = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)])
    out = orc.mult_relin(a, b, evk, alpha, q, p, psi_q, psi_p)
    da = orc.add(a[0], orc.mul(a[1], s_ntt, q), q)
    db = orc.add(b[0], orc.mul(b[1], s_ntt, q), q)
    dout = orc.add(out[0], orc.mul(out[1], s_ntt, q), q)
    diff = orc.ntt_batch(orc.sub(dout, orc.mul(da, db, q), q), q, psi_q, inverse=True)
    err = _crt2(diff[0], diff[1], int(q[0]), int(q[1]))
    assert max(abs(e) for e in err) < 2 ** 40


@pytest.mark.parametrize("f,with_add", [(2, True), (1, False)])
def test_mult_affine_rescale_is_rescale_of_the_product(orc, f, with_add):
    """rescale(f * mult_relin(a, b) + add) with ModDown and rescale as ONE conversion (orc_mult_affine_rescale) against the same
    thing step by step (orc_mult_relin, doubling, addition, orc_rescale): equal up to the roundings of the two conversions, and
    not the same integer function"""
    log_n, L1, k, alpha = 12, 4, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(18)
    n = 1 << log_n
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    s_q = np.array([[v % int(m) for v in s] for m in q], dtype=np.uint64)
    s_ntt = orc.ntt_batch(s_q, q, psi_q)
    s2_coeff = orc.ntt_batch(orc.mul(s_ntt, s_ntt, q), q, psi_q, inverse=True)
    s2 = _crt2(s2_coeff[0], s2_coeff[1], int(q[0]), int(q[1]))
    evk, _ = _toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s2, s, rng)
    a = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)])
    b = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)])
    add = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)]) if with_add else None
    one = orc.mult_affine_rescale(a, b, evk, f, add, alpha, q, p, psi_q, psi_p)
    t = orc.mult_relin(a, b, evk, alpha, q, p, psi_q, psi_p)
    if f == 2:
        t = np.stack([orc.add(t[c], t[c], q) for c in range(2)])
    if with_add:
        t = np.stack([orc.add(t[c], add[c], q) for c in range(2)])
    two = orc.rescale(t, q, psi_q)
    assert one.shape == two.shape == (2, L1 - 1, n)
    q1, p1, s1 = q[:L1 - 1], psi_q[:L1 - 1], s_ntt[:L1 - 1]
    d = orc.ntt_batch(orc.sub(_phase(orc, one, s1, q1), _phase(orc, two, s1, q1), q1), q1, p1, inverse=True)
    assert max(abs(e) for e in _crt2(d[0], d[1], int(q1[0]), int(q1[1]))) < 2 ** 20
    assert not np.array_equal(one, two)
    # coefficient by coefficient the two differ by a few units of MEAN ZERO (centred conversion): a common offset would be seen
    # N/pi-fold by the slots next to the root of unity 1
    for c in range(2):
        dc = orc.ntt_batch(orc.sub(one[c], two[c], q1), q1, p1, inverse=True)
        e = np.array(_crt2(dc[0], dc[1], int(q1[0]), int(q1[1])), dtype=np.float64)
        assert np.max(np.abs(e)) <= (k + 1) // 2 + 2 and abs(np.mean(e)) < 0.1, (np.max(np.abs(e)), np.mean(e))


def _rot_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, rots, rng):
    """rotation keys (s -> sigma_{g^-1}(s)) for every index in rots, stacked [R][dnum][2][L1+k][N]; galois elements"""
    n = 1 << log_n
    q0 = int(q[0])
    s_coeff = np.array([v % q0 for v in s], dtype=np.uint64)
    evks, gs = [], []
    for r in rots:
        g = orc.galois(log_n, r)
        s_perm = orc.automorph_coeff(s_coeff, pow(g, -1, 2 * n), q0)
        s_perm_int = [int(v) if v < q0 // 2 else int(v) - q0 for v in s_perm]
        evks.append(_toy_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, s_perm_int, rng)[0])
        gs.append(g)
    return np.stack(evks), gs


def _phase(orc, ct, s_ntt, q):
    return orc.add(ct[0], orc.mul(ct[1], s_ntt, q), q)


@pytest.mark.parametrize("ell,rots", [(4, [1, 2, 3]), (3, [128, -64]), (4, [5])])
def test_rotate_sum_by_decryption(orc, ell, rots):
    """merged rotate-and-sum (one ModUp, inner products summed in QP, one ModDown): decrypts to
    m + sum_r sigma_r(m) up to key-switching noise, at a full and a partial-digit level"""
    log_n, L1, k, alpha = 12, 4, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(11)
    n = 1 << log_n
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    evks, gs = _rot_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, rots, rng)
    ql, pl = q[:ell], psi_q[:ell]
    ct = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in ql] for _ in range(2)])
    out = orc.rotate_sum(ct, evks, gs, alpha, q, p, psi_q, psi_p)
    s_ntt = orc.ntt_batch(np.array([[v % int(m) for v in s] for m in ql], dtype=np.uint64), ql, pl)
    din = _phase(orc, ct, s_ntt, ql)
    expect = din
    for g in gs:
        expect = orc.add(expect, np.array([orc.automorph_ntt(x, g) for x in din]), ql)
    diff = orc.ntt_batch(orc.sub(_phase(orc, out, s_ntt, ql), expect, ql), ql, pl, inverse=True)
    err = _crt2(diff[0], diff[1], int(ql[0]), int(ql[1]))
    assert max(abs(e) for e in err) < 2 ** 40
    # the merged form differs from separate rotations only by ModDown rounding: tiny, but not zero
    sep = ct
    acc = ct
    for r, g in zip(range(len(gs)), gs):
        acc = np.stack([orc.add(acc[c], orc.rotate(ct, evks[r], g, alpha, q, p, psi_q, psi_p)[c], ql) for c in range(2)])
    d2 = orc.ntt_batch(orc.sub(_phase(orc, out, s_ntt, ql), _phase(orc, acc, s_ntt, ql), ql), ql, pl, inverse=True)
    assert max(abs(e) for e in _crt2(d2[0], d2[1], int(ql[0]), int(ql[1]))) < 2 ** 40
    del sep


def test_rotate_each_sum_by_decryption(orc):
    """giant-step form: sum_r rot(ct_r, r) with one ModUp per term and ONE shared ModDown"""
    log_n, L1, k, alpha, ell = 12, 4, 2, 2, 4
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(12)
    n = 1 << log_n
    rots = [4, 8, -16]
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    evks, gs = _rot_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, rots, rng)
    cts = np.array([[[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in q] for _ in range(2)] for _ in rots])
    out = orc.rotate_each_sum(cts, evks, gs, alpha, q, p, psi_q, psi_p)
    s_ntt = orc.ntt_batch(np.array([[v % int(m) for v in s] for m in q], dtype=np.uint64), q, psi_q)
    expect = None
    for ct, g in zip(cts, gs):
        t = np.array([orc.automorph_ntt(x, g) for x in _phase(orc, ct, s_ntt, q)])
        expect = t if expect is None else orc.add(expect, t, q)
    diff = orc.ntt_batch(orc.sub(_phase(orc, out, s_ntt, q), expect, q), q, psi_q, inverse=True)
    assert max(abs(e) for e in _crt2(diff[0], diff[1], int(q[0]), int(q[1]))) < 2 ** 40
    # a single term is the plain rotation minus nothing: ModDown(sigma(x)) vs sigma(ModDown(x)) differ by rounding only
    one = orc.rotate_each_sum(cts[:1], evks[:1], gs[:1], alpha, q, p, psi_q, psi_p)
    ref = orc.rotate(cts[0], evks[0], gs[0], alpha, q, p, psi_q, psi_p)
    d1 = orc.ntt_batch(orc.sub(_phase(orc, one, s_ntt, q), _phase(orc, ref, s_ntt, q), q), q, psi_q, inverse=True)
    assert max(abs(e) for e in _crt2(d1[0], d1[1], int(q[0]), int(q[1]))) < 2 ** 40


@pytest.mark.parametrize("ell", [4, 3])
def test_hoisted_dot_by_decryption(orc, ell):
    """double hoisting (plaintext products in the extended basis, one ModUp, one ModDown): decrypts to V_0 m + sum_r V_{r+1} sigma_r(m)
    up to key-switching noise times the plaintext size; and the variant that drops P and the top limb in ONE conversion equals
    rescale(that) up to the roundings of the two conversions"""
    log_n, L1, k, alpha = 12, 4, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    rng = np.random.default_rng(13)
    n = 1 << log_n
    rots = [1, 128, -64]
    s = [int(v) for v in rng.integers(-1, 2, size=n)]
    evks, gs = _rot_keys(orc, log_n, q, p, psi_q, psi_p, alpha, s, rots, rng)
    mods, psis = list(q) + list(p), list(psi_q) + list(psi_p)
    # plaintext polynomials with small integer coefficients, over the full key basis, NTT form
    vco = [[int(v) for v in rng.integers(-2 ** 10, 2 ** 10, size=n)] for _ in range(len(rots) + 1)]
    pts = np.stack([orc.ntt_batch(np.array([[c % int(m) for c in v] for m in mods], dtype=np.uint64), mods, psis) for v in vco])
    ql, pl = q[:ell], psi_q[:ell]
    ct = np.array([[rng.integers(0, int(m), size=n, dtype=np.uint64) for m in ql] for _ in range(2)])
    out = orc.hoisted_dot(ct, evks, gs, pts, alpha, q, p, psi_q, psi_p)
    s_ntt = orc.ntt_batch(np.array([[v % int(m) for v in s] for m in ql], dtype=np.uint64), ql, pl)
    din = _phase(orc, ct, s_ntt, ql)
    expect = orc.mul(din, pts[0][:ell], ql)
    for r, g in enumerate(gs):
        expect = orc.add(expect, orc.mul(np.array([orc.automorph_ntt(x, g) for x in din]), pts[r + 1][:ell], ql), ql)
    diff = orc.ntt_batch(orc.sub(_phase(orc, out, s_ntt, ql), expect, ql), ql, pl, inverse=True)
    assert max(abs(e) for e in _crt2(diff[0], diff[1], int(ql[0]), int(ql[1]))) < 2 ** 62      # noise 2^40 x plaintext 2^10 x sqrt(n) with room
    # ModDown and rescale as one conversion vs ModDown, then rescale
    one = orc.hoisted_dot(ct, evks, gs, pts, alpha, q, p, psi_q, psi_p, drop=True)
    two = orc.rescale(out, ql, pl)
    assert one.shape == two.shape == (2, ell - 1, n)
    q1, p1 = ql[:ell - 1], pl[:ell - 1]
    s1 = s_ntt[:ell - 1]
    d = orc.ntt_batch(orc.sub(_phase(orc, one, s1, q1), _phase(orc, two, s1, q1), q1), q1, p1, inverse=True)
    assert max(abs(e) for e in _crt2(d[0], d[1], int(q1[0]), int(q1[1]))) < 2 ** 20           # a few units per coefficient x |s|_1
    assert not np.array_equal(one, two)                                                      # a different integer function


def test_modraise_is_the_centred_lift(orc):
    """ModRaise: every coefficient's centred representative modulo q0, read modulo each q_t (python big ints)"""
    log_n, nl = 12, 4
    q, _, psi, _ = _chain(orc, log_n, nl, 1)
    rng = np.random.default_rng(13)
    n = 1 << log_n
    q0 = int(q[0])
    co = rng.integers(0, q0, size=(2, n), dtype=np.uint64)
    co[0, :4] = [0, q0 - 1, q0 // 2, q0 // 2 + 1]          # boundary values of the centring
    src = np.stack([orc.ntt_forward(c, q0, psi[0]) for c in co])
    out = orc.modraise(src, nl, q, psi)
    assert out.shape == (2, nl, n)
    for pidx in range(2):
        back = orc.ntt_batch(out[pidx], q, psi, inverse=True)
        for t in range(nl):
            qt = int(q[t])
            want = np.array([(int(v) - q0 if int(v) > q0 // 2 else int(v)) % qt for v in co[pidx]], dtype=np.uint64)
            assert np.array_equal(back[t], want), t


def test_fast_build_equals_definition_build(orc):
    """libfhe_oracle_fast.so (Barrett reductions; the timed cpu_baseline leg) returns the residues of the
    by-definition build (`%`) for every residue function, including all-maximal operands"""
    log_n, L1, k, alpha = 12, 5, 2, 2
    q, p, psi_q, psi_p = _chain(orc, log_n, L1, k)
    n = 1 << log_n
    rng = np.random.default_rng(14)
    allm = np.concatenate([q, p])
    d = -(-L1 // alpha)
    evks = np.stack([np.stack([orc.uniform_residues(900 + 50 * j + 7 * r, allm, n) for j in range(2 * d)]).reshape(d, 2, L1 + k, n)
                     for r in range(2)])
    gs = [orc.galois(log_n, 1), orc.galois(log_n, -2)]
    res = {}
    for fast in (False, True):
        orc.use_fast(fast)
        try:
            assert bool(orc.lib().orc_is_fast_build()) == fast
            for ell in (5, 3):
                ql = q[:ell]
                a = np.stack([orc.uniform_residues(1 + i, ql, n) for i in range(2)])
                b = np.stack([orc.uniform_residues(5 + i, ql, n) for i in range(2)])
                mx = np.stack([np.stack([np.full(n, int(m) - 1, dtype=np.uint64) for m in ql])] * 2)
                sc = np.array([int(m) - 2 for m in ql], dtype=np.uint64)
                for tag, x in (("rnd", a), ("max", mx)):
                    out = [orc.mul(x[0], b[1], ql), orc.mul_scalar(x[0], sc, ql), orc.muladd(x[0], x[1], b[0], ql),
                           orc.rescale(x, ql, psi_q[:ell]) if ell > 1 else x,
                           orc.rotate(x, evks[0], gs[0], alpha, q, p, psi_q, psi_p),
                           orc.rotate_sum(x, evks, gs, alpha, q, p, psi_q, psi_p),
                           orc.rotate_each_sum(np.stack([x, b]), evks, gs, alpha, q, p, psi_q, psi_p),
                           orc.mult_relin(x, b, evks[1], alpha, q, p, psi_q, psi_p),
                           orc.modraise(x[:, 0], ell, ql, psi_q[:ell])]
                    key = (ell, tag)
                    if fast:
                        for u, v in zip(res[key], out):
                            assert np.array_equal(u, v), key
                    else:
                        res[key] = out
        finally:
            orc.use_fast(False)
    del rng
