"""oracle — CPU restatement of the reference's RNS-CKKS hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (fhe-linformer_amd/) never does.  PARITY STATUS: "parity unpinned" (see fhe_oracle.c header):
the arithmetic of this path lives in OpenFHE, an un-pinned, un-vendored dependency of the reference, and
the reference holds no golden vectors; the oracle is pinned by library-independent known-answer tests.

numpy front-end over libfhe_oracle.so (plain C, oracle/fhe_oracle.c).  On top of it (pure Python, exact integer functions):
residue_eval.py (the evaluator's bookkeeping, composites, polynomial evaluation), residue_boot.py (bootstrapping),
residue_controller.py (the FHEController surface: the whole driver on residues).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def use_fast(fast=True):
    """Select the build every wrapper below calls: the by-definition build (default; the parity checker) or the
    -DORC_FAST build (Barrett reductions instead of `%`), which bench.py times as `cpu_baseline`."""
    global _LIB
    _LIB = _load("libfhe_oracle_fast.so" if fast else "libfhe_oracle.so")
    return _LIB


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load("libfhe_oracle.so")
    return _LIB


def _load(name):
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u64, i32, vp, lng = C.c_uint64, C.c_int, C.c_void_p, C.c_long
        sig = {
            "orc_is_prime": (i32, [u64]),
            "orc_min_root": (u64, [u64, u64]),
            "orc_prime_chain": (i32, [i32, i32, i32, i32, i32, i32, vp, vp]),
            "orc_ntt_naive": (None, [vp, vp, i32, u64, u64]),
            "orc_negacyclic_mul_naive": (None, [vp, vp, vp, i32, u64]),
            "orc_ntt_forward": (None, [vp, i32, u64, u64]),
            "orc_ntt_inverse": (None, [vp, i32, u64, u64]),
            "orc_ntt_batch": (None, [vp, i32, i32, vp, vp, i32, i32]),
            "orc_mul": (None, [vp, vp, vp, i32, i32, vp]),
            "orc_add": (None, [vp, vp, vp, i32, i32, vp]),
            "orc_sub": (None, [vp, vp, vp, i32, i32, vp]),
            "orc_mul_scalar": (None, [vp, vp, vp, i32, i32, vp]),
            "orc_add_scalar": (None, [vp, vp, vp, i32, i32, vp]),
            "orc_muladd": (None, [vp, vp, vp, vp, i32, i32, vp]),
            "orc_modraise": (None, [vp, vp, i32, i32, i32, vp, vp]),
            "orc_rotate_sum": (None, [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_rotate_each_sum": (None, [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_dot": (None, [vp, vp, i32, vp, i32, i32, vp]),
            "orc_mult_affine_rescale": (None, [vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_hoisted_dot": (None, [vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_is_fast_build": (i32, []),
            "orc_automorph_coeff": (None, [vp, vp, i32, u64, u64]),
            "orc_automorph_ntt": (None, [vp, vp, i32, u64]),
            "orc_galois": (u64, [i32, lng]),
            "orc_rescale": (None, [vp, vp, i32, i32, i32, vp, vp]),
            "orc_keyswitch": (None, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_rotate": (None, [vp, vp, vp, u64, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_mult_relin": (None, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
            "orc_num_threads": (i32, []),
            "orc_set_threads": (None, [i32]),
        }
        for n, (r, a) in sig.items():
            f = getattr(L, n)
            f.restype, f.argtypes = r, a
        _LIBS[name] = L
    return _LIBS[name]


def _u(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def is_prime(n):
    return bool(lib().orc_is_prime(int(n)))


def min_root(q, two_n):
    return int(lib().orc_min_root(int(q), int(two_n)))


def prime_chain(log_n, n_q, first_bits, scale_bits, n_p, special_bits):
    q = np.zeros(n_q, dtype=np.uint64)
    p = np.zeros(max(n_p, 1), dtype=np.uint64)
    rc = lib().orc_prime_chain(log_n, n_q, first_bits, scale_bits, n_p, special_bits, _p(q), _p(p))
    if rc:
        raise RuntimeError("prime chain search failed")
    return q, p[:n_p]


def ntt_naive(a, q, psi):
    a = _u(a)
    out = np.empty_like(a)
    lib().orc_ntt_naive(_p(a), _p(out), int(np.log2(a.size)), int(q), int(psi))
    return out


def negacyclic_mul_naive(a, b, q):
    a, b = _u(a), _u(b)
    c = np.empty_like(a)
    lib().orc_negacyclic_mul_naive(_p(a), _p(b), _p(c), int(np.log2(a.size)), int(q))
    return c


def ntt_forward(a, q, psi):
    a = _u(a).copy()
    lib().orc_ntt_forward(_p(a), int(np.log2(a.size)), int(q), int(psi))
    return a


def ntt_inverse(a, q, psi):
    a = _u(a).copy()
    lib().orc_ntt_inverse(_p(a), int(np.log2(a.size)), int(q), int(psi))
    return a


def ntt_batch(data, q, psi, inverse=False, inplace=False):
    """data[..., N]; flattened vector v uses q[v % len(q)]."""
    d = _u(data) if inplace else _u(data).copy()
    n = d.shape[-1]
    nvec = d.size // n
    q, psi = _u(q), _u(psi)
    lib().orc_ntt_batch(_p(d), nvec, len(q), _p(q), _p(psi), int(np.log2(n)), 1 if inverse else 0)
    return d


def _dy(fn, a, b, q):
    a, b, q = _u(a), _u(b), _u(q)
    c = np.empty_like(a)
    nl, n = a.shape[-2], a.shape[-1]
    assert a.ndim == 2 and len(q) == nl
    fn(_p(a), _p(b), _p(c), nl, int(np.log2(n)), _p(q))
    return c


def mul(a, b, q):
    return _dy(lib().orc_mul, a, b, q)


def add(a, b, q):
    return _dy(lib().orc_add, a, b, q)


def sub(a, b, q):
    return _dy(lib().orc_sub, a, b, q)


def mul_scalar(a, s, q):
    a, s, q = _u(a), _u(s), _u(q)
    c = np.empty_like(a)
    lib().orc_mul_scalar(_p(a), _p(s), _p(c), a.shape[0], int(np.log2(a.shape[1])), _p(q))
    return c


def add_scalar(a, s, q):
    a, s, q = _u(a), _u(s), _u(q)
    c = np.empty_like(a)
    lib().orc_add_scalar(_p(a), _p(s), _p(c), a.shape[0], int(np.log2(a.shape[1])), _p(q))
    return c


def dot(a_list, b_list, q):
    """sum_i a_list[i] (.) b_list[i] over [nlimbs][N] arrays (exact sums of dyadic products)"""
    import ctypes as C
    a_list = [_u(a) for a in a_list]
    b_list = [_u(b) for b in b_list]
    q = _u(q)
    nl, n = a_list[0].shape
    assert len(a_list) == len(b_list) and all(a.shape == (nl, n) for a in a_list) and all(b.shape == (nl, n) for b in b_list)
    pa = (C.c_void_p * len(a_list))(*[a.ctypes.data for a in a_list])
    pb = (C.c_void_p * len(b_list))(*[b.ctypes.data for b in b_list])
    out = np.empty((nl, n), dtype=np.uint64)
    lib().orc_dot(pa, pb, len(a_list), _p(out), nl, int(np.log2(n)), _p(q))
    return out


def muladd(acc, a, b, q):
    """acc + a * b per limb ([nlimbs][N] each)"""
    acc, a, b, q = _u(acc), _u(a), _u(b), _u(q)
    c = np.empty_like(a)
    lib().orc_muladd(_p(acc), _p(a), _p(b), _p(c), a.shape[0], int(np.log2(a.shape[1])), _p(q))
    return c


def modraise(src, nl, q, psi):
    """src [npoly][N] (one limb per polynomial, NTT form mod q[0]) -> [npoly][nl][N] NTT form over q[:nl]"""
    src, q, psi = _u(src), _u(q), _u(psi)
    npoly, n = src.shape
    out = np.empty((npoly, nl, n), dtype=np.uint64)
    lib().orc_modraise(_p(src), _p(out), npoly, nl, int(np.log2(n)), _p(q), _p(psi))
    return out


def galois(log_n, r):
    return int(lib().orc_galois(log_n, int(r)))


def automorph_coeff(a, g, q):
    a = _u(a)
    out = np.empty_like(a)
    lib().orc_automorph_coeff(_p(a), _p(out), int(np.log2(a.size)), int(g), int(q))
    return out


def automorph_ntt(a, g):
    a = _u(a)
    out = np.empty_like(a)
    lib().orc_automorph_ntt(_p(a), _p(out), int(np.log2(a.size)), int(g))
    return out


def rescale(ct, q, psi):
    """ct [npoly][ell][N] -> [npoly][ell-1][N]"""
    ct, q, psi = _u(ct), _u(q), _u(psi)
    npoly, ell, n = ct.shape
    out = np.empty((npoly, ell - 1, n), dtype=np.uint64)
    lib().orc_rescale(_p(ct), _p(out), npoly, ell, int(np.log2(n)), _p(q), _p(psi))
    return out


def keyswitch(c, evk, alpha, q, p, psi_q, psi_p):
    """c [ell][N]; evk [digits][2][L1+k][N]; q, psi_q: full Q chain (L1 entries)."""
    c, evk, q, p, psi_q, psi_p = _u(c), _u(evk), _u(q), _u(p), _u(psi_q), _u(psi_p)
    ell, n = c.shape
    out = np.empty((2, ell, n), dtype=np.uint64)
    lib().orc_keyswitch(_p(c), _p(evk), _p(out), ell, len(q), len(p), alpha, int(np.log2(n)), _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def rotate(ct, evk, g, alpha, q, p, psi_q, psi_p):
    ct, evk, q, p, psi_q, psi_p = _u(ct), _u(evk), _u(q), _u(p), _u(psi_q), _u(psi_p)
    _, ell, n = ct.shape
    out = np.empty((2, ell, n), dtype=np.uint64)
    lib().orc_rotate(_p(ct), _p(evk), _p(out), int(g), ell, len(q), len(p), alpha, int(np.log2(n)), _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def rotate_sum(ct, evks, gs, alpha, q, p, psi_q, psi_p):
    """merged rotate-and-sum: ct + sum_r rot(ct, r) with ONE ModUp and ONE ModDown; evks [R][digits][2][L1+k][N]"""
    ct, evks, q, p, psi_q, psi_p = _u(ct), _u(evks), _u(q), _u(p), _u(psi_q), _u(psi_p)
    gs = _u(gs)
    _, ell, n = ct.shape
    assert evks.shape[0] == len(gs) and evks.shape[1:] == (-(-len(q) // alpha), 2, len(q) + len(p), n)
    out = np.empty((2, ell, n), dtype=np.uint64)
    lib().orc_rotate_sum(_p(ct), _p(evks), _p(gs), len(gs), _p(out), ell, len(q), len(p), alpha, int(np.log2(n)),
                         _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def rotate_each_sum(cts, evks, gs, alpha, q, p, psi_q, psi_p):
    """sum_r rot(cts[r], r): own ModUp per term, inner products summed in QP, ONE ModDown; cts [R][2][ell][N]"""
    cts, evks, q, p, psi_q, psi_p = _u(cts), _u(evks), _u(q), _u(p), _u(psi_q), _u(psi_p)
    gs = _u(gs)
    R, _, ell, n = cts.shape
    assert R == len(gs) == evks.shape[0] and evks.shape[1:] == (-(-len(q) // alpha), 2, len(q) + len(p), n)
    out = np.empty((2, ell, n), dtype=np.uint64)
    lib().orc_rotate_each_sum(_p(cts), _p(evks), _p(gs), R, _p(out), ell, len(q), len(p), alpha, int(np.log2(n)),
                              _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def hoisted_dot(ct, evks, gs, pts, alpha, q, p, psi_q, psi_p, drop=False):
    """double hoisting: ct * V_0 + sum_r rot(ct, r) * V_{r+1} with ONE ModUp and ONE ModDown, the plaintext products taken in QP;
    pts [R+1][L1+k][N] = the plaintext encodings over the full key basis.  drop: the result rescaled, ModDown and rescale as one
    basis conversion -> [2][ell-1][N]"""
    ct, evks, pts, q, p, psi_q, psi_p = _u(ct), _u(evks), _u(pts), _u(q), _u(p), _u(psi_q), _u(psi_p)
    gs = _u(gs)
    _, ell, n = ct.shape
    R = len(gs)
    assert evks.shape == (R, -(-len(q) // alpha), 2, len(q) + len(p), n) and pts.shape == (R + 1, len(q) + len(p), n)
    out = np.empty((2, ell - 1 if drop else ell, n), dtype=np.uint64)
    lib().orc_hoisted_dot(_p(ct), _p(evks), _p(gs), R, _p(pts), _p(out), int(bool(drop)), ell, len(q), len(p), alpha, int(np.log2(n)),
                          _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def mult_affine_rescale(a, b, evk, f, addq, alpha, q, p, psi_q, psi_p):
    """rescale(f * EvalMult(a, b) + addq) with ModDown and rescale as one basis conversion; a, b [2][ell][N]; addq [2][ell][N] or None
    -> [2][ell-1][N]"""
    a, b, evk, q, p, psi_q, psi_p = _u(a), _u(b), _u(evk), _u(q), _u(p), _u(psi_q), _u(psi_p)
    _, ell, n = a.shape
    assert b.shape == a.shape and f in (1, 2) and ell >= 2
    if addq is not None:
        addq = _u(addq)
        assert addq.shape == a.shape
    out = np.empty((2, ell - 1, n), dtype=np.uint64)
    lib().orc_mult_affine_rescale(_p(a), _p(b), _p(evk), int(f), _p(addq) if addq is not None else None, _p(out), ell, len(q), len(p), alpha,
                                  int(np.log2(n)), _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def mult_relin(a, b, evk, alpha, q, p, psi_q, psi_p):
    a, b, evk, q, p, psi_q, psi_p = _u(a), _u(b), _u(evk), _u(q), _u(p), _u(psi_q), _u(psi_p)
    _, ell, n = a.shape
    out = np.empty((2, ell, n), dtype=np.uint64)
    lib().orc_mult_relin(_p(a), _p(b), _p(evk), _p(out), ell, len(q), len(p), alpha, int(np.log2(n)), _p(q), _p(p), _p(psi_q), _p(psi_p))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def set_threads(n):
    lib().orc_set_threads(int(n))


def splitmix64(seed, n):
    """n uniform 64-bit words from SplitMix64(seed) — the synthetic-input generator of SURVEY.md §8(d)."""
    out = np.empty(n, dtype=np.uint64)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out[:] = z ^ (z >> np.uint64(31))
    return out


def uniform_residues(seed, q, n):
    """[len(q)][n] residues uniform-ish in [0, q_l) from SplitMix64(seed + l) (128-bit multiply-shift)."""
    q = _u(q)
    out = np.empty((len(q), n), dtype=np.uint64)
    for l, ql in enumerate(q):
        w = splitmix64(int(seed) + l, n)
        # floor(w * q / 2^64) via Python ints on 32-bit halves (exact)
        hi, lo = w >> np.uint64(32), w & np.uint64(0xFFFFFFFF)
        qh, qlo = int(ql) >> 32, int(ql) & 0xFFFFFFFF
        with np.errstate(over="ignore"):
            t = (lo * np.uint64(qlo)) >> np.uint64(32)
            m1 = hi * np.uint64(qlo) + t
            m2 = lo * np.uint64(qh) + (m1 & np.uint64(0xFFFFFFFF))
            out[l] = hi * np.uint64(qh) + (m1 >> np.uint64(32)) + (m2 >> np.uint64(32))
    return out
