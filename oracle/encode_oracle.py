"""CKKS encoding BY DEFINITION, in multi-precision arithmetic.  TEST INFRASTRUCTURE (part of oracle/; never imported by the product).

What the reference obtains from OpenFHE's `MakeCKKSPackedPlaintext(vec, 1, level, nullptr, slots)` (reference
src/FHEController.cpp:348-371; OpenFHE is absent from the image: "parity unpinned") is, by the published definition of the scheme
(Cheon-Kim-Kim-Song 2017, section 3.2; the HEAAN / OpenFHE slot order): the integer polynomial m(X) in Z[X]/(X^N + 1) whose values at
the roots zeta^(5^k), zeta = exp(2 pi i / 2N), are Delta * z_k for the n slots k = 0..n-1 - and their conjugates at zeta^(-5^k).
For n < N/2 slots the polynomial lives in the subring Z[Y], Y = X^(N/2n): with omega = zeta^(N/2n) = exp(2 pi i / 4n)

    m(X) = sum_{i<n} ( Re u_i + Im u_i X^(N/2) ) Y^i ,      u_i = (1/n) sum_{k<n} z_k omega^(-(5^k mod 4n) i)        (complex, i < n)

(X^(N/2) evaluates to i at every zeta^(5^k) because 5^k = 1 mod 4; the n exponents 5^k mod 4n are the residues = 1 mod 4, so the
sum over k is an inverse DFT on a coset and the u_i are unique), every coefficient rounded to the nearest integer after the
multiplication by Delta.  Nothing here shares code or an algorithm with the library's encoder (csrc/client.cpp: an in-place
radix-2 "special FFT" in fp64, host and device): this file evaluates the two definitions directly - O(n) per coefficient,
O(n) per slot - in mpmath at 256 bits, for the coefficients / slots a test asks for.

    exact_coefficients(z, n, N, scale, idx)   coefficient idx[j] of m(X) as exact integers (the inverse embedding)
    evaluate_slots(coeffs, n, N, ks)          m(zeta^(5^k)) for k in ks from an integer coefficient vector (the embedding)
"""
import mpmath as mp

PREC = 256


def _omega_table(n):
    """omega^t, t < 4n, omega = exp(2 pi i / 4n): one quadrant by cos / sin, the rest by symmetry (exact to working precision)"""
    m = 4 * n
    tab = [None] * m
    for t in range(n + 1):
        a = 2 * mp.pi * t / m
        tab[t] = mp.mpc(mp.cos(a), mp.sin(a))
    for t in range(n + 1, m):
        q, r = divmod(t, n)
        b = tab[r]
        tab[t] = (b * 1j) if q == 1 else (-b if q == 2 else (b * -1j))
    return tab


def rotation_group(n):
    """5^k mod 4n, k < n"""
    g, out = 1, []
    for _ in range(n):
        out.append(g)
        g = g * 5 % (4 * n)
    return out


def exact_coefficients(z, n, N, scale, idx):
    """{j: coefficient j of the encoding of the slot vector z (length <= n, complex or real; missing slots are 0) at scale `scale`}
    for j in idx.  Coefficients that are structurally zero (j not a multiple of the gap N/2n modulo N/2) come back as 0."""
    with mp.workprec(PREC):
        gap = (N // 2) // n
        tab = _omega_table(n)
        rot = rotation_group(n)
        zs = [mp.mpc(complex(v).real, complex(v).imag) for v in z] + [mp.mpc(0)] * (n - len(z))
        sc = mp.mpf(scale) if not isinstance(scale, tuple) else mp.mpf(scale[0]) + mp.mpf(scale[1])
        out = {}
        cache = {}
        for j in idx:
            jj = j % (N // 2)
            if jj % gap:
                out[j] = 0
                continue
            i = jj // gap
            if i not in cache:
                acc = mp.mpc(0)
                for k in range(n):
                    if zs[k] != 0:
                        acc += zs[k] * tab[(-rot[k] * i) % (4 * n)]
                cache[i] = acc / n
            u = cache[i]
            v = (u.real if j < N // 2 else u.imag) * sc
            out[j] = int(mp.floor(v + mp.mpf(1) / 2)) if v >= 0 else -int(mp.floor(-v + mp.mpf(1) / 2))     # half away from zero
        return out


def evaluate_slots(coeffs, n, N, ks):
    """m(zeta^(5^k)) for k in ks, as Python complex pairs of mp values: coeffs = the N integer coefficients (only multiples of the
    gap may be non-zero: checked)"""
    with mp.workprec(PREC):
        gap = (N // 2) // n
        tab = _omega_table(n)
        rot = rotation_group(n)
        nz = [(j, int(c)) for j, c in enumerate(coeffs) if int(c) != 0]
        for j, _ in nz:
            if (j % (N // 2)) % gap:
                raise ValueError("coefficient %d is outside the subring of %d slots" % (j, n))
        out = []
        for k in ks:
            acc = mp.mpc(0)
            for j, c in nz:
                i = (j % (N // 2)) // gap
                term = tab[(rot[k] * i) % (4 * n)] * c
                acc += term * 1j if j >= N // 2 else term
            out.append(acc)
        return out
