"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fhelin.h declares, builds the
parameter layer identically to the oracle's independent restatement, and refuses to evaluate without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fhelin.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fhelin_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(fa):
    lib = fa.load_library()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libfhelin_amd.so does not export {n}"
    assert b"gfx950" in lib.fhelin_version()


@pytest.mark.parametrize("preset", ["toy", "reference", "bench", "deep"])
def test_parameter_layer_matches_oracle(fa, orc, preset):
    e = fa.Engine(preset, device=-1)
    try:
        cfg = fa.PRESETS[preset]
        q, p = orc.prime_chain(cfg["log_n"], cfg["n_q"], cfg["first_bits"], cfg["scale_bits"], cfg["n_p"], cfg["special_bits"])
        assert np.array_equal(e.q, q) and np.array_equal(e.p, p)
        two_n = 2 << cfg["log_n"]
        for m, r in list(zip(e.moduli, e.roots))[:: max(1, len(e.moduli) // 6)]:
            assert int(r) == orc.min_root(int(m), two_n)
        assert e.alpha == -(-cfg["n_q"] // cfg["dnum"])
        # FLEXIBLEAUTO real scaling factors: Delta_0 = q_L, Delta_{k+1} = Delta_k^2 / q_{L-k}
        sf = e.scaling_factors
        assert sf[0] == float(q[-1])
        for k in range(len(sf) - 1):
            assert abs(sf[k + 1] - sf[k] * sf[k] / float(q[len(q) - 1 - k])) <= 2.0 ** -40 * sf[k + 1]
    finally:
        e.close()


def test_reference_parameter_shape(fa):
    """reference src/FHEController.cpp:6-35: N=2^15, depth 27 -> 28 limbs, dnum 4 -> 7 limbs per digit."""
    e = fa.Engine("reference", device=-1)
    assert (e.N, e.n_q, e.n_p, e.alpha) == (1 << 15, 28, 7, 7)
    e.close()


def test_host_only_context_refuses_to_evaluate(fa):
    """No CPU fallback: evaluation entry points on a device-less context fail with FHELIN_ERR_NO_DEVICE."""
    e = fa.Engine("toy", device=-1)
    assert not e.has_device
    with pytest.raises(fa.FhelinError) as ei:
        e.buf(1024)
    assert ei.value.code == 2
    with pytest.raises(fa.FhelinError) as ei:
        e.sync()
    assert ei.value.code == 2
    dummy = type("B", (), {"ptr": C.c_void_p(16)})()
    with pytest.raises(fa.FhelinError) as ei:
        e.ntt(dummy, 1)
    assert ei.value.code == 2
    e.close()


def test_bad_parameters_are_rejected(fa):
    for kw in (dict(log_n=11), dict(log_n=18), dict(n_q=0), dict(dnum=0), dict(special_bits=62)):
        with pytest.raises(fa.FhelinError) as ei:
            fa.Engine("toy", device=-1, **kw)
        assert ei.value.code == 1


@pytest.mark.parametrize("n_q,dnum,want", [(29, 4, 7), (28, 4, 7), (30, 4, 7), (24, 4, 6), (6, 3, 2)])
def test_special_prime_count_follows_openfhe_rule(fa, n_q, dnum, want):
    """n_p < 0: ceil(bits of the widest digit / special_bits) special primes, the count OpenFHE's HYBRID parameter
    generation picks (reference parameters: 28 limbs, dnum 4 -> 55 + 6*52 = 367 bits -> 7; this engine's 29-limb chain:
    419 bits -> 7).  The product of the special primes must cover every digit."""
    e = fa.Engine("reference", device=-1, n_q=n_q, dnum=dnum, n_p=-1)
    try:
        assert e.n_p == want and len(e.p) == want
        alpha = -(-n_q // dnum)
        digits = [e.q[i:i + alpha] for i in range(0, n_q, alpha)]
        widest = max(sum(np.log2(d.astype(np.float64))) for d in digits)
        assert sum(np.log2(e.p.astype(np.float64))) >= widest
    finally:
        e.close()


def test_client_generator_is_chacha20_rfc8439(fa):
    """the client-side generator (secret keys, encryption randomness, key-switching noise) is the ChaCha20 stream:
    RFC 8439 section 2.3.2 block-function test vector (key 00..1f, counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00)"""
    lib = fa.load_library()
    out = (C.c_uint8 * 64)()
    key = (C.c_uint8 * 32)(*range(32))
    assert lib.fhelin_prng_block(key, (0x09000000 << 32) | 1, 0x4A000000, out) == 0
    want = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert bytes(out) == want
    assert lib.fhelin_prng_block(key, 2, 0, out) == 0 and bytes(out) != want


def test_default_seed_is_os_entropy_and_test_seed_is_deterministic(fa):
    """seed 0 (the default) keys the generator with 256 bits of OS entropy: two contexts never share it; a non-zero seed is
    the explicit deterministic test opt-in; an exported seed re-creates the same context (load_context path)"""
    a, b = fa.Engine("toy", device=-1), fa.Engine("toy", device=-1)
    sa, sb = a.secret_seed(), b.secret_seed()
    assert len(sa) == 32 and sa != sb and sa != bytes(32)
    c, d = fa.Engine("toy", device=-1, seed=7), fa.Engine("toy", device=-1, seed=7)
    assert c.secret_seed() == d.secret_seed() == (7).to_bytes(8, "little") + bytes(24)
    e = fa.Engine("toy", device=-1, seed_bytes=sa)
    assert e.secret_seed() == sa
    for x in (a, b, c, d, e):
        x.close()


def test_level_plan_bookkeeping_on_a_host_context(fa):
    """fhelin_level_plan_* (include/fhelin.h): the plan is plain data on the context — it can be set, read back and a pass
    can be opened on it without a device; applying with nothing loaded, bad modes and seeking inside a recording are refused"""
    e = fa.Engine("toy", device=-1)
    assert e.level_plan() == []
    with pytest.raises(fa.FhelinError) as ei:
        e.level_plan_begin("apply")
    assert ei.value.code == 4
    assert e.lib.fhelin_level_plan_begin(e.h, 3) == 1
    e.set_level_plan([4, -1, 9, 2])
    assert e.level_plan() == [4, -1, 9, 2]
    e.level_plan_begin("apply", first_source=2)
    assert e.level_plan_end() == [4, -1, 9, 2]               # ending an applying pass keeps the plan
    e.level_plan_begin("record")
    assert e.lib.fhelin_level_plan_seek(e.h, 1) == 4         # a recording runs from its first source
    assert e.level_plan_end() == []                          # nothing was recorded: an empty plan replaces the old one
    e.close()


def test_device_arena_logic_on_a_pretend_device(fa):
    """csrc/context.cpp DevicePool (slabs from the driver, best fit, address-ordered free list with coalescing) against a mock backend:
    blocks never overlap and lie inside a slab, the pool follows the bytes in use closely (the exact-size caching pool it replaces
    held 1.6 x), freeing everything leaves one free range per slab, and a trim hands every slab back."""
    import ctypes as C
    lib = fa.load_library()
    for seed, dev in ((1, 288 << 30), (2, 288 << 30), (3, 16 << 30)):
        out = np.zeros(6, dtype=np.uint64)
        rc = lib.fhelin_debug_pool_selftest(seed, 6000, dev, out.ctypes.data_as(C.POINTER(C.c_uint64)), 6)
        assert rc == 0, lib.fhelin_last_error()
        peak_live, peak_held, slabs, oom, coalesced, trimmed = (int(v) for v in out)
        assert coalesced == 1 and trimmed == 1
        assert peak_held <= dev and slabs >= 1
        assert peak_held <= 1.25 * peak_live + (8 << 30), (peak_held / 2**30, peak_live / 2**30)
