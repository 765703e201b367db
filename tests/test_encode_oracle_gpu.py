"""The library's CKKS encoder and encryptor against INDEPENDENT statements (row a16 of SURVEY.md 8(a); reference
src/FHEController.cpp:348-385: MakeCKKSPackedPlaintext / Encrypt).

Encoder: `fhelin_pt_export` (the residues every ct x pt / ct + pt operation and every residue-level test takes as "the plaintext")
is transformed back with the oracle's INTT, lifted to integers by the CRT, and compared coefficient for coefficient with
oracle/encode_oracle.py - the encoding BY DEFINITION (inverse canonical embedding in 256-bit mpmath arithmetic; no code or algorithm
shared with csrc/client.cpp's fp64 special FFT).  Tolerance, stated: the library computes the inverse FFT in fp64 (as OpenFHE does), so
a coefficient may miss the exactly rounded one by 1 + scale x 2^-50 x max|z| (5 units at the 2^52 scale; measured: 1 unit at 2^52, 2^-52 relative at 2^104); a convention error (slot order, conjugation, 1/n, the gap of sparse packing, the X^(N/2) half) is off by ~scale.
The reverse direction - the exported polynomial evaluated at zeta^(5^k) - must give back Delta z_k.

Encryptor: for a fresh ciphertext, c0 + c1 s - m (oracle arithmetic on the exported secret and the exported encoding) must be a SMALL
polynomial with the variance public-key encryption predicts: sigma^2 (2N/3 + h + 1) for v = e u + e0 + e1 s (sigma = 3.19, u
uniform ternary, s ternary of weight h) - a statement about fhelin_encrypt's kernels that does not go through the library's decryption."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _crt(res, mods):
    """centred integer with the given residues"""
    x, M = 0, 1
    for r, q in zip(res, mods):
        r, q = int(r), int(q)
        t = ((r - x) * pow(M, -1, q)) % q
        x, M = x + M * t, M * q
    return x - M if x > M // 2 else x


@pytest.mark.parametrize("preset", ["toy13", "reference", "bench"])
def test_exported_encodings_equal_the_definition(fa, orc, preset):
    from oracle import encode_oracle as eo
    eng = fa.Engine(preset, seed=3)
    try:
        N, n = eng.N, 1 << eng.params.log_slots
        gap = (N // 2) // n
        rng = np.random.default_rng(17)
        z = rng.uniform(-1, 1, n)
        pt = eng.encode(z)
        sf = eng.scaling_factors
        nl = eng.n_q + eng.n_p
        dl = lambda ell: float(sf[eng.n_q - ell])       # the level's Delta rounded to a double: given explicitly, so both sides use the same number
        cases = [(eng.n_q, dl(eng.n_q)), (max(2, eng.n_q // 2), dl(max(2, eng.n_q // 2))), (1, dl(1)),   # level 0, mid, last
                 (eng.n_q, float(sf[0]) * float(sf[0])),                                    # a degree-2 scale (~2^104)
                 (nl, float(sf[0]))]                                                        # the full key basis Q u P (folded keys)
        pick = sorted(set([0, gap, (n - 1) * gap, N // 2, N // 2 + gap] + [int(i) * gap for i in rng.integers(0, n, 20)] +
                          [N // 2 + int(i) * gap for i in rng.integers(0, n, 20)]))
        worst = 0.0
        for ell, sc in cases:
            res = eng.pt_export(pt, ell, sc)                                                # [ell][N], NTT form
            mods = eng.moduli[:ell]
            psi = eng.roots[:ell]
            co = orc.ntt_batch(res, mods, psi, inverse=True)                                # coefficient residues
            scale = sc
            # the subring: only multiples of the gap (mod N/2) are non-zero
            if gap > 1:
                off = np.ones(N, dtype=bool)
                off[np.arange(0, N // 2, gap)] = False
                off[N // 2 + np.arange(0, N // 2, gap)] = False
                assert not co[:, off].any(), (preset, ell, "coefficients outside the subring")
            want = eo.exact_coefficients(z, n, N, scale, pick)
            tol = 1 + scale * 2.0 ** -50
            for j in pick:
                got = _crt(co[:, j], mods)
                assert all(got % int(q) == int(r) for r, q in zip(co[:, j], mods))          # one integer behind all limbs
                d = abs(got - want[j])
                worst = max(worst, d / tol)
                assert d <= tol, (preset, ell, sc, j, got, want[j])
        # the embedding itself: the exported polynomial at zeta^(5^k) is Delta z_k
        res = eng.pt_export(pt, 2, dl(2))
        co = orc.ntt_batch(res, eng.moduli[:2], eng.roots[:2], inverse=True)
        nzj = np.nonzero(co[0])[0]
        coeffs = [0] * N
        for j in nzj:
            coeffs[j] = _crt(co[:, j], eng.moduli[:2])
        ks = [0, 1, n - 1] + [int(k) for k in rng.integers(0, n, 5)]
        back = eo.evaluate_slots(coeffs, n, N, ks)
        D = float(sf[eng.n_q - 2])
        for k, b in zip(ks, back):
            assert abs(complex(b) / D - z[k]) < 2.0 ** -40, (preset, k)
        print(f"{preset}: worst coefficient deviation {worst:.3f} of the stated tolerance")
    finally:
        eng.close()


@pytest.mark.parametrize("preset", ["toy13", "bench"])
def test_fresh_encryption_is_the_encoding_plus_small_noise(fa, orc, preset):
    eng = fa.Engine(preset, seed=21)
    try:
        eng.keygen()
        N, n = eng.N, 1 << eng.params.log_slots
        rng = np.random.default_rng(4)
        z = rng.uniform(-1, 1, n)
        pt = eng.encode(z)
        s = eng.secret_export()                                                            # [n_q + n_p][N], NTT form
        h = eng.params.hamming
        for ct in (eng.encrypt(z), eng.encrypt_batch(np.stack([z, -z]), 0, n)[0]):           # the one-input and the batched path
            c = ct.export()
            ell = c.shape[1]
            q, psi = eng.q[:ell], eng.psi_q[:ell]
            phase = orc.muladd(c[0], c[1], s[:ell], q)                                       # c0 + c1 s
            hi, lo = ct.scale_parts()
            m = eng.pt_export(pt, ell, np.longdouble(hi) + np.longdouble(lo))
            noise = orc.ntt_batch(orc.sub(phase, m, q), q, psi, inverse=True)
            v = noise[0].astype(np.int64)
            v = np.where(v > int(q[0]) // 2, v - int(q[0]), v).astype(np.float64)            # centred; the same polynomial on every limb:
            for t in range(1, ell):
                vt = noise[t].astype(np.int64)
                assert np.array_equal(np.where(vt > int(q[t]) // 2, vt - int(q[t]), vt), v.astype(np.int64))
            std = np.sqrt(3.19 ** 2 * (2 * N / 3 + h + 1))
            assert abs(v.std() / std - 1) < 0.1 and np.abs(v).max() < 6.5 * std and abs(v.mean()) < 0.1 * std, (v.std(), std)
    finally:
        eng.close()
