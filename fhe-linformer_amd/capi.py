"""ctypes binding of include/fhelin.h (test / bench harness only; the product is the .so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FhelinError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fhelin error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("log_n", C.c_int32), ("n_q", C.c_int32), ("first_bits", C.c_int32), ("scale_bits", C.c_int32),
        ("n_p", C.c_int32), ("special_bits", C.c_int32), ("dnum", C.c_int32), ("log_slots", C.c_int32),
        ("hamming", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64),
    ]


# Parameter presets.  "reference" = literal values of reference src/FHEController.cpp:6-35 (N=2^15, depth 27
# -> 28 Q limbs, dnum 4 -> alpha 7, 7 special primes); "bench" = BASELINE.json synthetic config
# (N=2^16, 24 limbs, k=6); "deep" = config 5 (N=2^17, 30 limbs, k=8); "toy*" = small rings for fast tests.
PRESETS = {
    "bench": dict(log_n=16, n_q=24, first_bits=55, scale_bits=52, n_p=6, special_bits=60, dnum=4, log_slots=14, hamming=192),
    "reference": dict(log_n=15, n_q=28, first_bits=55, scale_bits=52, n_p=7, special_bits=60, dnum=4, log_slots=14, hamming=192),
    "deep": dict(log_n=17, n_q=30, first_bits=55, scale_bits=52, n_p=8, special_bits=60, dnum=4, log_slots=14, hamming=192),
    "toy": dict(log_n=12, n_q=6, first_bits=55, scale_bits=52, n_p=2, special_bits=60, dnum=3, log_slots=11, hamming=64),
    "boot12": dict(log_n=12, n_q=22, first_bits=55, scale_bits=52, n_p=6, special_bits=60, dnum=4, log_slots=10, hamming=64),
    "toy13": dict(log_n=13, n_q=7, first_bits=55, scale_bits=52, n_p=3, special_bits=60, dnum=3, log_slots=12, hamming=64),
}


def circuit_rotation_indices():
    """The rotation keys a client generates for the Linformer circuit: the +-2^i set the reference's composites rotate by
    (the shim's generate_rotation_keys extends main.cpp's list to it, quirk Q3), plus +-3*2^i, with which two steps of a
    rotate-and-sum tree run as one merged key switch (Evaluator::rotate_sum_batch), plus +-5s and +-7s for the tree units
    s in {1, 8, 128, 512, 1024}, with which three steps do."""
    r = set()
    for i in range(14):
        r.update((1 << i, -(1 << i)))
    for i in range(13):
        r.update((3 << i, -(3 << i)))
    for s in (1, 8, 128, 512, 1024):      # three tree steps as one merged key switch: the keys of s..7s (6s = 3*2s is above)
        r.update((5 * s, -5 * s, 7 * s, -7 * s))
    return sorted(r)


def library_path():
    # FHELIN_LIB: an alternative build of the same library (A/B measurements of kernel variants: tools/build_variant.sh)
    return os.environ.get("FHELIN_LIB") or os.path.join(_HERE, "libfhelin_amd.so")


def load_library():
    """Load libfhelin_amd.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise FhelinError(-1, f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i32, u64p, f32p = C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    sigs = {
        "fhelin_last_error": (C.c_char_p, []),
        "fhelin_version": (C.c_char_p, []),
        "fhelin_ctx_create": (i32, [C.POINTER(Params), C.POINTER(vp)]),
        "fhelin_ctx_create_seeded": (i32, [C.POINTER(Params), vp, C.POINTER(vp)]),
        "fhelin_ctx_secret_seed": (i32, [vp, vp]),
        "fhelin_prng_block": (i32, [vp, C.c_uint64, C.c_uint64, vp]),
        "fhelin_ctx_destroy": (None, [vp]),
        "fhelin_ctx_info": (i32, [vp, C.POINTER(Params), C.POINTER(i32), C.POINTER(i32)]),
        "fhelin_ctx_moduli": (i32, [vp, u64p, i32]),
        "fhelin_ctx_roots": (i32, [vp, u64p, i32]),
        "fhelin_ctx_scaling_factors": (i32, [vp, C.POINTER(C.c_double), i32]),
        "fhelin_ctx_set_stream": (i32, [vp, vp]),
        "fhelin_ctx_set_lazy_rows": (i32, [vp, i32]),
        "fhelin_ct_force": (i32, [vp, C.POINTER(vp), i32]),
        "fhelin_level_plan_begin": (i32, [vp, i32]),
        "fhelin_level_plan_seek": (i32, [vp, i32]),
        "fhelin_level_plan_tell": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "fhelin_level_plan_end": (i32, [vp, C.POINTER(i32)]),
        "fhelin_level_plan_get": (i32, [vp, C.POINTER(i32), i32, C.POINTER(i32)]),
        "fhelin_level_plan_set": (i32, [vp, C.POINTER(i32), i32]),
        "fhelin_sync": (i32, [vp]),
        "fhelin_ctx_set_lane": (i32, [vp, i32]),
        "fhelin_ctx_lane_wait": (i32, [vp, i32]),
        "fhelin_ctx_lane_mark": (i32, [vp]),
        "fhelin_ctx_lane_wait_mark": (i32, [vp, i32]),
        "fhelin_ctx_lanes_fork": (i32, [vp]),
        "fhelin_ctx_lanes_join": (i32, [vp]),
        "fhelin_ctx_trim": (i32, [vp]),
        "fhelin_timer_start": (i32, [vp]),
        "fhelin_timer_stop": (i32, [vp, f32p]),
        "fhelin_dev_alloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
        "fhelin_dev_free": (i32, [vp, vp]),
        "fhelin_dev_upload": (i32, [vp, vp, vp, C.c_size_t]),
        "fhelin_dev_download": (i32, [vp, vp, vp, C.c_size_t]),
        "fhelin_ntt": (i32, [vp, vp, i32, i32, i32, i32]),
        "fhelin_stats": (i32, [vp, u64p, i32, i32]),
        "fhelin_debug_pool_selftest": (i32, [C.c_uint64, i32, C.c_uint64, u64p, i32]),
        "fhelin_keygen": (i32, [vp]),
        "fhelin_gen_relin_key": (i32, [vp]),
        "fhelin_gen_rotation_keys": (i32, [vp, C.POINTER(i32), i32]),
        "fhelin_gen_conj_key": (i32, [vp]),
        "fhelin_secret_export": (i32, [vp, vp, C.c_size_t]),
        "fhelin_secret_import": (i32, [vp, vp, C.c_size_t]),
        "fhelin_key_export": (i32, [vp, i32, i32, vp, C.c_size_t]),
        "fhelin_key_import": (i32, [vp, i32, i32, vp, C.c_size_t]),
        "fhelin_encode": (i32, [vp, C.POINTER(C.c_double), i32, i32, i32, C.POINTER(vp)]),
        "fhelin_pt_free": (None, [vp]),
        "fhelin_pt_export": (i32, [vp, vp, i32, C.c_double, C.c_double, vp, C.c_size_t]),
        "fhelin_rotate_each_sum": (i32, [vp, C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(vp)]),
        "fhelin_raw_modraise": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_raw_phase": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_encrypt": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_encrypt_batch": (i32, [vp, C.POINTER(C.c_double), i32, i32, i32, i32, C.POINTER(vp)]),
        "fhelin_ctx_set_host_encode": (i32, [vp, i32]),
        "fhelin_client_ingest": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(vp), vp]),
        "fhelin_debug_sample": (i32, [vp, i32, i32, vp, C.c_size_t]),
        "fhelin_decrypt": (i32, [vp, vp, C.POINTER(C.c_double), i32]),
        "fhelin_ct_import": (i32, [vp, vp, i32, i32, i32, C.c_double, i32, C.POINTER(vp)]),
        "fhelin_ct_export": (i32, [vp, vp, vp, C.c_size_t]),
        "fhelin_ct_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(i32)]),
        "fhelin_ct_export_device": (i32, [vp, vp, vp, C.c_size_t]),
        "fhelin_ct_import_device": (i32, [vp, vp, i32, i32, i32, C.c_double, C.c_double, i32, C.POINTER(vp)]),
        "fhelin_ct_scale": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "fhelin_fc_unwrapRepeatedLarge_range": (i32, [vp, C.POINTER(vp), i32, i32, i32, i32, C.POINTER(vp)]),
        "fhelin_ct_clone": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_ct_free": (None, [vp]),
        "fhelin_add": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_sub": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_negate": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_add_plain": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_mult_plain": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_mult": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_rotate": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_rotate_many": (i32, [vp, vp, C.POINTER(i32), i32, C.POINTER(vp)]),
        "fhelin_rotate_each": (i32, [vp, C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(vp)]),
        "fhelin_rotate_sum": (i32, [vp, C.POINTER(vp), i32, C.POINTER(i32), i32, C.POINTER(vp)]),
        "fhelin_hoisted_dot": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(i32), i32, i32, C.POINTER(vp)]),
        "fhelin_rescale": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_rotate_batch": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_rescale_batch": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_mult_plain_batch": (i32, [vp, C.POINTER(vp), i32, vp, C.POINTER(vp)]),
        "fhelin_mult_batch": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_add_batch": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_level_reduce": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_raw_rescale": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_raw_rotate": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_raw_mult_relin": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "fhelin_fc_mult_const": (i32, [vp, vp, C.c_double, C.POINTER(vp)]),
        "fhelin_fc_mask": (i32, [vp, vp, i32, i32, i32, C.c_double, C.POINTER(vp)]),
        "fhelin_fc_rotsum": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "fhelin_fc_repeat": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "fhelin_fc_add_many": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_fc_matmul_pt": (i32, [vp, C.POINTER(vp), i32, vp, vp, i32, i32, C.POINTER(vp)]),
        "fhelin_fc_matmul_ct": (i32, [vp, C.POINTER(vp), i32, vp, i32, i32, C.POINTER(vp)]),
        "fhelin_fc_matmulRElarge": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp), i32, vp, C.c_double, C.POINTER(vp)]),
        "fhelin_fc_matmulCRlarge": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp), vp, C.POINTER(vp)]),
        "fhelin_fc_matmulScores": (i32, [vp, C.POINTER(vp), i32, vp, C.POINTER(vp)]),
        "fhelin_fc_wrapUpRepeated": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_fc_wrapUpExpanded": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_fc_unwrapExpanded": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_fc_unwrapScoresExpanded": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_fc_unwrap_512_in_4_128": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_fc_unwrapRepeatedLarge": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_fc_generate_containers": (i32, [vp, C.POINTER(vp), i32, vp, C.POINTER(vp), C.POINTER(i32)]),
        "fhelin_fc_wrap_containers": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_mult_real": (i32, [vp, vp, C.c_double, C.POINTER(vp)]),
        "fhelin_add_real": (i32, [vp, vp, C.c_double, C.POINTER(vp)]),
        "fhelin_mult_many": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_lincomb": (i32, [vp, C.POINTER(vp), C.POINTER(C.c_double), i32, C.c_double, C.POINTER(vp)]),
        "fhelin_eval_poly": (i32, [vp, vp, C.POINTER(C.c_double), i32, C.POINTER(vp)]),
        "fhelin_eval_chebyshev": (i32, [vp, vp, C.POINTER(C.c_double), i32, C.c_double, C.c_double, C.POINTER(vp)]),
        "fhelin_eval_chebyshev_batch": (i32, [vp, C.POINTER(vp), i32, C.POINTER(C.c_double), i32, C.c_double, C.c_double, C.POINTER(vp)]),
        "fhelin_bootstrap_setup": (i32, [vp, i32, i32, i32]),
        "fhelin_bootstrap": (i32, [vp, vp, C.POINTER(vp)]),
        "fhelin_bootstrap_config": (i32, [vp, i32, i32, i32, i32]),
        "fhelin_bootstrap_partial": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_bootstrap_drop": (i32, [vp, vp, i32, C.POINTER(vp)]),
        "fhelin_bootstrap_batch": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_bootstrap_describe": (i32, [vp, C.POINTER(i32), i32, C.POINTER(i32)]),
        "fhelin_bootstrap_diag": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "fhelin_bootstrap_cheb": (i32, [vp, C.POINTER(C.c_double), i32, C.POINTER(i32)]),
        "fhelin_fcb_matmulScores": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp), i32, C.POINTER(vp)]),
        "fhelin_fcb_matmul_ct": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, i32, i32, C.POINTER(vp)]),
        "fhelin_fcb_wrapUpRepeated": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_fcb_wrapUpExpanded": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_fcb_unwrapExpanded": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
        "fhelin_fcb_unwrapRepeatedLarge": (i32, [vp, C.POINTER(vp), i32, i32, i32, C.POINTER(vp)]),
        "fhelin_fcb_generate_containers": (i32, [vp, C.POINTER(vp), i32, i32, vp, C.POINTER(vp), C.POINTER(i32)]),
        "fhelin_fc_rotsum_batch": (i32, [vp, C.POINTER(vp), i32, i32, i32, i32, C.POINTER(vp)]),
        "fhelin_add_plain_batch": (i32, [vp, C.POINTER(vp), i32, vp, C.POINTER(vp)]),
        "fhelin_eval_poly_batch": (i32, [vp, C.POINTER(vp), i32, C.POINTER(C.c_double), i32, C.POINTER(vp)]),
        "fhelin_mult_many_batch": (i32, [vp, C.POINTER(vp), i32, i32, C.POINTER(vp)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _np_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.c_void_p)


class DevBuf:
    """A raw device allocation owned by an Engine."""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, int(nbytes)
        p = C.c_void_p()
        eng._ck(eng.lib.fhelin_dev_alloc(eng.h, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.eng._ck(self.eng.lib.fhelin_dev_upload(self.eng.h, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.eng._ck(self.eng.lib.fhelin_dev_download(self.eng.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.eng._ck(self.eng.lib.fhelin_dev_free(self.eng.h, self.ptr))
            self.ptr = None


class Engine:
    """One fhelin context (one GPU, one stream)."""

    def __init__(self, preset="bench", device=0, seed=0, seed_bytes=None, **overrides):
        """seed=0 (default): keys from 256 bits of OS entropy.  seed != 0: deterministic TEST seed.  seed_bytes: an explicit
        32-byte secret seed (re-creating a client's keys)."""
        self.lib = load_library()
        cfg = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
        cfg.update(overrides)
        self.params = Params(device=device, seed=seed, **cfg)
        h = C.c_void_p()
        if seed_bytes is not None:
            assert len(seed_bytes) == 32
            sb = (C.c_uint8 * 32)(*bytes(seed_bytes))
            rc = self.lib.fhelin_ctx_create_seeded(C.byref(self.params), sb, C.byref(h))
        else:
            rc = self.lib.fhelin_ctx_create(C.byref(self.params), C.byref(h))
        if rc != 0:
            raise FhelinError(rc, self.lib.fhelin_last_error().decode())
        self.h = h
        if self.params.n_p < 0:      # derived by the library (OpenFHE's sizeP rule): read the resolved parameters back
            rc = self.lib.fhelin_ctx_info(self.h, C.byref(self.params), None, None)
            if rc != 0:
                raise FhelinError(rc, self.lib.fhelin_last_error().decode())
        self.log_n, self.N = self.params.log_n, 1 << self.params.log_n
        self.n_q, self.n_p = self.params.n_q, self.params.n_p
        nl = self.n_q + self.n_p
        m = np.zeros(nl, dtype=np.uint64)
        self._ck(self.lib.fhelin_ctx_moduli(self.h, m.ctypes.data_as(C.POINTER(C.c_uint64)), nl))
        r = np.zeros(nl, dtype=np.uint64)
        self._ck(self.lib.fhelin_ctx_roots(self.h, r.ctypes.data_as(C.POINTER(C.c_uint64)), nl))
        self.moduli, self.roots = m, r
        self.q, self.p = m[: self.n_q], m[self.n_q:]
        self.psi_q, self.psi_p = r[: self.n_q], r[self.n_q:]
        sf = np.zeros(self.n_q, dtype=np.float64)
        self._ck(self.lib.fhelin_ctx_scaling_factors(self.h, sf.ctypes.data_as(C.POINTER(C.c_double)), self.n_q))
        self.scaling_factors = sf
        a, d = C.c_int32(), C.c_int32()
        self._ck(self.lib.fhelin_ctx_info(self.h, None, C.byref(a), C.byref(d)))
        self.alpha, self.has_device = a.value, bool(d.value)
        self.lazy_heavy = os.environ.get("FHELIN_LAZY_HEAVY", "1") != "0"
        self.lazy_rows_on = os.environ.get("FHELIN_LAZY_ROWS", "1") != "0"   # the library reads the same knob (capi.cpp)

    def _ck(self, rc):
        if rc != 0:
            raise FhelinError(rc, self.lib.fhelin_last_error().decode())

    def set_lazy_rows(self, on):
        """deferred evaluation of the rows of matmul_pt / unwrapExpanded (default on)"""
        self._ck(self.lib.fhelin_ctx_set_lazy_rows(self.h, 1 if on else 0))
        self.lazy_rows_on = bool(on)

    # level plan (include/fhelin.h fhelin_level_plan_*): record one pass of a straight-line driver, apply to later ones
    def force(self, cts):
        """evaluate exactly the deferred rows among `cts` now, one batched call per producing call"""
        cts = list(cts)
        if cts:
            self._ck(self.lib.fhelin_ct_force(self.h, self._harr(cts), len(cts)))

    def level_plan_begin(self, mode, first_source=0):
        """mode: "record", "apply" or "off"; the pass starts at the program's first source unless first_source says otherwise
        (apply only: a server pass that starts after the client's encryptions)"""
        self._ck(self.lib.fhelin_level_plan_begin(self.h, {"off": 0, "record": 1, "apply": 2}[mode]))
        if first_source:
            self._ck(self.lib.fhelin_level_plan_seek(self.h, int(first_source)))

    def level_plan_tell(self):
        """(mode, index of the next source call): mode "off" / "record" / "apply" """
        m, k = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.fhelin_level_plan_tell(self.h, C.byref(m), C.byref(k)))
        return ("off", "record", "apply")[m.value], k.value

    def level_plan_seek(self, source):
        self._ck(self.lib.fhelin_level_plan_seek(self.h, int(source)))

    def level_plan_end(self):
        """ends the pass (a recording pass derives the plan); returns the plan: limbs per source, -1 = as asked"""
        n = C.c_int32(0)
        self._ck(self.lib.fhelin_level_plan_end(self.h, C.byref(n)))
        return self.level_plan()

    def level_plan(self):
        n = C.c_int32(0)
        self._ck(self.lib.fhelin_level_plan_get(self.h, None, 0, C.byref(n)))
        buf = (C.c_int32 * max(1, n.value))()
        self._ck(self.lib.fhelin_level_plan_get(self.h, buf, n.value, C.byref(n)))
        return list(buf[:n.value])

    def set_level_plan(self, target):
        arr = (C.c_int32 * len(target))(*[int(t) for t in target])
        self._ck(self.lib.fhelin_level_plan_set(self.h, arr, len(target)))

    def secret_seed(self):
        out = (C.c_uint8 * 32)()
        self._ck(self.lib.fhelin_ctx_secret_seed(self.h, out))
        return bytes(out)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fhelin_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- raw buffers / kernels
    def buf(self, nbytes):
        return DevBuf(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, arr.nbytes).upload(arr)

    def sync(self):
        self._ck(self.lib.fhelin_sync(self.h))

    def set_lane(self, k):
        """subsequent calls launch on lane k's stream (0 = the main stream) and allocate from its arena"""
        self._ck(self.lib.fhelin_ctx_set_lane(self.h, int(k)))

    def lane_wait(self, from_lane):
        """the current lane's stream waits for everything issued under from_lane so far"""
        self._ck(self.lib.fhelin_ctx_lane_wait(self.h, int(from_lane)))

    def lane_mark(self):
        self._ck(self.lib.fhelin_ctx_lane_mark(self.h))

    def lane_wait_mark(self, from_lane):
        self._ck(self.lib.fhelin_ctx_lane_wait_mark(self.h, int(from_lane)))

    def lanes_fork(self):
        self._ck(self.lib.fhelin_ctx_lanes_fork(self.h))

    def lanes_join(self):
        self._ck(self.lib.fhelin_ctx_lanes_join(self.h))

    def trim(self):
        """release the device memory the caching pool holds but does not use"""
        self._ck(self.lib.fhelin_ctx_trim(self.h))

    def timer_start(self):
        self._ck(self.lib.fhelin_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.fhelin_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def ntt(self, buf, nvec, limb_first=0, limb_count=None, inverse=False):
        if limb_count is None:
            limb_count = self.n_q
        self._ck(self.lib.fhelin_ntt(self.h, buf.ptr, nvec, limb_first, limb_count, 1 if inverse else 0))

    def stats(self, reset=False):
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.lib.fhelin_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 16, 1 if reset else 0))
        keys = ["limb_ntt", "keyswitch", "keyswitch_limbs", "rescale", "ct_pt_mult", "bootstrap", "encode", "rescale_limbs", "ct_pt_limbs",
                "pool_malloc_calls", "pool_malloc_bytes", "pool_malloc_ns", "pool_reserved_bytes", "pool_live_peak_bytes",
                "pool_reserved_peak_bytes", "pool_trims"]
        return {k: int(v) for k, v in zip(keys, out)}

    # ---- keys
    def keygen(self):
        self._ck(self.lib.fhelin_keygen(self.h))

    def gen_relin_key(self):
        self._ck(self.lib.fhelin_gen_relin_key(self.h))

    def gen_rotation_keys(self, indices):
        arr = (C.c_int32 * len(indices))(*indices)
        self._ck(self.lib.fhelin_gen_rotation_keys(self.h, arr, len(indices)))

    @property
    def n_limbs(self):
        return self.n_q + self.n_p

    @property
    def dnum_digits(self):
        return -(-self.n_q // self.alpha)

    def secret_export(self):
        out = np.empty((self.n_limbs, self.N), dtype=np.uint64)
        self._ck(self.lib.fhelin_secret_export(self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def secret_import(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self._ck(self.lib.fhelin_secret_import(self.h, arr.ctypes.data_as(C.c_void_p), arr.size))

    def key_export(self, kind, index=0):
        out = np.empty((self.dnum_digits, 2, self.n_limbs, self.N), dtype=np.uint64)
        self._ck(self.lib.fhelin_key_export(self.h, kind, index, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def key_import(self, kind, index, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        assert arr.shape == (self.dnum_digits, 2, self.n_limbs, self.N)
        self._ck(self.lib.fhelin_key_import(self.h, kind, index, arr.ctypes.data_as(C.c_void_p), arr.size))

    # ---- plaintexts / ciphertexts
    def encode(self, vals, level=0, slots=0):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        h = C.c_void_p()
        self._ck(self.lib.fhelin_encode(self.h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, level, slots, C.byref(h)))
        return Pt(self, h)

    def encrypt(self, vals, level=0, slots=0):
        pt = vals if isinstance(vals, Pt) else self.encode(vals, level, slots)
        h = C.c_void_p()
        self._ck(self.lib.fhelin_encrypt(self.h, pt.h, C.byref(h)))
        return Ct(self, h)

    def encrypt_batch(self, rows, level=0, slots=0):
        """rows [n_vec][n_per] -> n_vec fresh ciphertexts: encode + sample + combine as batched GPU kernels"""
        a = np.ascontiguousarray(rows, dtype=np.float64)
        assert a.ndim == 2
        outs = self._outs(a.shape[0])
        self._ck(self.lib.fhelin_encrypt_batch(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1], level, slots, outs))
        return self._cts(outs, a.shape[0])

    def client_ingest(self, cls, pos, E_w, E_b, F_w, F_b, emb=None, tokens=None, table=None, level=0, want_proj=False):
        """one sample's client side on the device: (embedding gather,) positional embedding, Linformer projections, expanded packing,
        encode + encrypt.  Returns {"inputs_E": 32 cts, "inputs_F": 32 cts, "inputs": S+1 cts} (+ x_in, proj if want_proj)"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        cls, pos, E_w, E_b, F_w, F_b = f(cls), f(pos), f(E_w), f(E_b), f(F_w), f(F_b)
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        if emb is not None:
            emb = f(emb)
            S, tok, tab, vocab = emb.shape[0], None, None, 0
        else:
            tok = np.ascontiguousarray(tokens, dtype=np.int32)
            tab = f(table)
            S, vocab = tok.shape[0], tab.shape[0]
        if pos.shape[0] < S or pos.shape[1] != 128 or E_w.shape != F_w.shape or E_w.shape[0] != 32 or E_w.shape[1] < S + 1 or cls.size != 128 \
                or E_b.size != 32 or F_b.size != 32:
            raise FhelinError(1, "client_ingest: pos needs >= S rows of 128, E_w / F_w [32][>= S + 1], E_b / F_b [32], cls [128]")
        n = 64 + S + 1
        outs = self._outs(n)
        proj = np.empty((S + 1 + 64, 128)) if want_proj else None
        self._ck(self.lib.fhelin_client_ingest(self.h, p(emb), p(tok), p(tab), vocab, S, p(cls), p(pos), p(E_w), p(E_b), p(F_w), p(F_b),
                                               E_w.shape[1], level, outs, p(proj)))
        cts = self._cts(outs, n)
        res = {"inputs_E": cts[:32], "inputs_F": cts[32:64], "inputs": cts[64:]}
        if want_proj:
            res["x_in"], res["proj"] = proj[:S + 1], proj[S + 1:]
        return res

    def set_host_encode(self, on):
        self._ck(self.lib.fhelin_ctx_set_host_encode(self.h, 1 if on else 0))

    def debug_sample(self, kind, n_poly=1):
        out = np.empty((n_poly, self.N), dtype=np.int64)
        self._ck(self.lib.fhelin_debug_sample(self.h, kind, n_poly, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def decrypt(self, ct, slots=0):
        n = slots or ct.slots or (1 << self.params.log_slots)
        out = np.empty(n, dtype=np.float64)
        self._ck(self.lib.fhelin_decrypt(self.h, ct.h, out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def ct_import(self, limbs, deg=1, scale=None, slots=0):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        npoly, ell, n = limbs.shape
        assert n == self.N
        if scale is None:
            scale = float(self.scaling_factors[self.n_q - ell])
        h = C.c_void_p()
        self._ck(self.lib.fhelin_ct_import(self.h, limbs.ctypes.data_as(C.c_void_p), npoly, ell, deg, scale, slots, C.byref(h)))
        return Ct(self, h)

    def _un(self, fn, a, *extra):
        h = C.c_void_p()
        self._ck(fn(self.h, a.h, *extra, C.byref(h)))
        return Ct(self, h)

    def add(self, a, b):
        return self._un(self.lib.fhelin_add_plain if isinstance(b, Pt) else self.lib.fhelin_add, a, b.h)

    def sub(self, a, b):
        return self._un(self.lib.fhelin_sub, a, b.h)

    def negate(self, a):
        return self._un(self.lib.fhelin_negate, a)

    def mult(self, a, b):
        return self._un(self.lib.fhelin_mult_plain if isinstance(b, Pt) else self.lib.fhelin_mult, a, b.h)

    def rotate(self, a, index):
        return self._un(self.lib.fhelin_rotate, a, index)

    def rotate_many(self, a, indices):
        """hoisted rotations of one ciphertext (one ModUp, bit-identical to rotate(a, i) for every i)"""
        idx = (C.c_int32 * len(indices))(*indices)
        outs = self._outs(len(indices))
        self._ck(self.lib.fhelin_rotate_many(self.h, a.h, idx, len(indices), outs))
        return self._cts(outs, len(indices))

    def rotate_each(self, v, indices):
        idx = (C.c_int32 * len(indices))(*indices)
        outs = self._outs(len(v))
        self._ck(self.lib.fhelin_rotate_each(self.h, self._harr(v), idx, len(v), outs))
        return self._cts(outs, len(v))

    def rotate_sum(self, v, indices):
        """v[i] + sum_r rot(v[i], indices[r]): one ModUp and one ModDown per row"""
        idx = (C.c_int32 * len(indices))(*indices)
        outs = self._outs(len(v))
        self._ck(self.lib.fhelin_rotate_sum(self.h, self._harr(v), len(v), idx, len(indices), outs))
        return self._cts(outs, len(v))

    def hoisted_dot(self, v, pts, indices, rescale=False):
        """v[i] * pts[0] + sum_r rot(v[i], indices[r]) * pts[r + 1]: one ModUp and one ModDown per row, plaintext products in the
        extended basis (plaintext-folded rotation keys); rescale: ModDown and rescale as one basis conversion"""
        assert len(pts) == len(indices) + 1
        idx = (C.c_int32 * len(indices))(*indices)
        parr = (C.c_void_p * len(pts))(*[p.h for p in pts])
        outs = self._outs(len(v))
        self._ck(self.lib.fhelin_hoisted_dot(self.h, self._harr(v), len(v), parr, idx, len(indices), int(bool(rescale)), outs))
        return self._cts(outs, len(v))

    def rotate_each_sum(self, v, indices):
        """sum_i rot(v[i], indices[i]): own ModUp per term, one shared ModDown per <= 7 terms"""
        idx = (C.c_int32 * len(indices))(*indices)
        h = C.c_void_p()
        self._ck(self.lib.fhelin_rotate_each_sum(self.h, self._harr(v), idx, len(v), C.byref(h)))
        return Ct(self, h)

    def raw_modraise(self, a, new_ell):
        return self._un(self.lib.fhelin_raw_modraise, a, new_ell)

    def raw_phase(self, a):
        return self._un(self.lib.fhelin_raw_phase, a)

    def pt_export(self, pt, ell, scale=0.0):
        """[ell][N] residues of the plaintext's encoding at (ell limbs, scale); scale 0 = Delta of that level.  `scale`
        may be a numpy longdouble: it travels exactly as hi + lo doubles."""
        out = np.empty((ell, self.N), dtype=np.uint64)     # ell = n_q + n_p (explicit scale): the encoding over the full key basis
        hi = float(scale)
        lo = float(np.longdouble(scale) - np.longdouble(hi))
        self._ck(self.lib.fhelin_pt_export(self.h, pt.h, ell, hi, lo, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def rescale(self, a):
        return self._un(self.lib.fhelin_rescale, a)

    # ---- leaf ops over independent rows (one launch set per chunk of rows of equal shape)
    def _rows(self, fn, v, *extra):
        outs = self._outs(len(v))
        self._ck(fn(self.h, self._harr(v), *extra, outs))
        return self._cts(outs, len(v))

    def rotate_batch(self, v, index):
        return self._rows(self.lib.fhelin_rotate_batch, v, len(v), index)

    def rescale_batch(self, v):
        return self._rows(self.lib.fhelin_rescale_batch, v, len(v))

    def mult_plain_batch(self, v, pt):
        return self._rows(self.lib.fhelin_mult_plain_batch, v, len(v), pt.h)

    def mult_batch(self, a, b):
        outs = self._outs(len(a))
        self._ck(self.lib.fhelin_mult_batch(self.h, self._harr(a), self._harr(b), len(a), outs))
        return self._cts(outs, len(a))

    def add_batch(self, a, b):
        outs = self._outs(len(a))
        self._ck(self.lib.fhelin_add_batch(self.h, self._harr(a), self._harr(b), len(a), outs))
        return self._cts(outs, len(a))

    def level_reduce(self, a, new_ell):
        return self._un(self.lib.fhelin_level_reduce, a, new_ell)

    def raw_rescale(self, a):
        return self._un(self.lib.fhelin_raw_rescale, a)

    def raw_rotate(self, a, index):
        return self._un(self.lib.fhelin_raw_rotate, a, index)

    def raw_mult_relin(self, a, b):
        return self._un(self.lib.fhelin_raw_mult_relin, a, b.h)

    # ---- FHEController composites (names follow the reference methods)
    @staticmethod
    def _harr(objs):
        return (C.c_void_p * len(objs))(*[o.h for o in objs])

    def _outs(self, n):
        return (C.c_void_p * n)()

    def _cts(self, arr, n):
        return [Ct(self, C.c_void_p(arr[i])) for i in range(n)]

    def mult_real(self, a, k):
        return self._un(self.lib.fhelin_mult_real, a, float(k))

    def add_real(self, a, k):
        return self._un(self.lib.fhelin_add_real, a, float(k))

    def mult_many(self, v):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_mult_many(self.h, self._harr(v), len(v), C.byref(h)))
        return Ct(self, h)

    def lincomb(self, v, coeffs, c0=0.0):
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        h = C.c_void_p()
        self._ck(self.lib.fhelin_lincomb(self.h, self._harr(v), cf.ctypes.data_as(C.POINTER(C.c_double)), len(v), float(c0), C.byref(h)))
        return Ct(self, h)

    def eval_poly(self, x, coeffs):
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        return self._un(self.lib.fhelin_eval_poly, x, cf.ctypes.data_as(C.POINTER(C.c_double)), cf.size)

    def eval_chebyshev(self, x, coeffs, a=-1.0, b=1.0):
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        return self._un(self.lib.fhelin_eval_chebyshev, x, cf.ctypes.data_as(C.POINTER(C.c_double)), cf.size, float(a), float(b))

    def eval_chebyshev_batch(self, xs, coeffs, a=-1.0, b=1.0):
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        outs = self._outs(len(xs))
        self._ck(self.lib.fhelin_eval_chebyshev_batch(self.h, self._harr(xs), len(xs), cf.ctypes.data_as(C.POINTER(C.c_double)), cf.size,
                                                      float(a), float(b), outs))
        return self._cts(outs, len(xs))

    def bootstrap(self, a):
        return self._un(self.lib.fhelin_bootstrap, a)

    def bootstrap_setup(self, budget_enc=3, budget_dec=3, slots=0):
        self._ck(self.lib.fhelin_bootstrap_setup(self.h, budget_enc, budget_dec, slots))

    def bootstrap_config(self, K=28, R=3, cheb_degree=47, correction=10):
        self._ck(self.lib.fhelin_bootstrap_config(self.h, K, R, cheb_degree, correction))

    def bootstrap_batch(self, v):
        """EvalBootstrap on independent ciphertexts in one batched pipeline (same residues as bootstrap() one by one)"""
        outs = self._outs(len(v))
        self._ck(self.lib.fhelin_bootstrap_batch(self.h, self._harr(v), len(v), outs))
        return self._cts(outs, len(v))

    def bootstrap_drop(self, a, drop):
        """bootstrap raising to L+1-drop limbs only (what a level plan asks of a bootstrap)"""
        return self._un(self.lib.fhelin_bootstrap_drop, a, int(drop))

    def bootstrap_describe(self):
        """the bootstrapping set-up as the residue-level oracle needs it: parameters + per linear stage the stage's slot count
        and its (giant, baby, diagonal plaintext) terms in evaluation order"""
        n = C.c_int32()
        self._ck(self.lib.fhelin_bootstrap_describe(self.h, None, 0, C.byref(n)))
        d = (C.c_int32 * n.value)()
        self._ck(self.lib.fhelin_bootstrap_describe(self.h, d, n.value, C.byref(n)))
        d = list(d)
        out = dict(zip(("packed", "slots", "K", "R", "cheb_degree", "correction", "depth"), d[:7]))
        out["packed"] = bool(out["packed"])
        n_c2s, n_s2c = d[7], d[8]
        pos, stages = 9, []
        for k in range(n_c2s + n_s2c):
            which, idx = (0, k) if k < n_c2s else (1, k - n_c2s)
            st_slots, nt = d[pos], d[pos + 1]
            pos += 2
            terms = []
            for t in range(nt):
                h = C.c_void_p()
                self._ck(self.lib.fhelin_bootstrap_diag(self.h, which, idx, t, C.byref(h)))
                terms.append((d[pos], d[pos + 1], Pt(self, h)))
                pos += 2
            stages.append(dict(slots=st_slots, terms=terms))
        out["c2s"], out["s2c"] = stages[:n_c2s], stages[n_c2s:]
        cn = C.c_int32()
        self._ck(self.lib.fhelin_bootstrap_cheb(self.h, None, 0, C.byref(cn)))
        cf = (C.c_double * cn.value)()
        self._ck(self.lib.fhelin_bootstrap_cheb(self.h, cf, cn.value, C.byref(cn)))
        out["cheb"] = list(cf)
        return out

    def bootstrap_partial(self, a, stage):
        return self._un(self.lib.fhelin_bootstrap_partial, a, stage)

    def mult_const(self, a, d):
        return self._un(self.lib.fhelin_fc_mult_const, a, float(d))

    def mask_block(self, a, frm, to, v=1.0):
        return self._un(self.lib.fhelin_fc_mask, a, 0, frm, to, float(v))

    def mask_heads(self, a, v=1.0):
        return self._un(self.lib.fhelin_fc_mask, a, 1, 0, 0, float(v))

    def mask_heads_128(self, a, v=1.0):
        return self._un(self.lib.fhelin_fc_mask, a, 2, 0, 0, float(v))

    def mask_mod_n(self, a, n, padding=0):
        return self._un(self.lib.fhelin_fc_mask, a, 3, n, padding, 1.0)

    def mask_first_n(self, a, n, v=1.0):
        return self._un(self.lib.fhelin_fc_mask, a, 4, n, 0, float(v))

    def rotsum(self, a, slots, padding):
        return self._un(self.lib.fhelin_fc_rotsum, a, slots, padding)

    def repeat(self, a, slots, padding=1):
        return self._un(self.lib.fhelin_fc_repeat, a, slots, padding)

    def add_many(self, v):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_fc_add_many(self.h, self._harr(v), len(v), C.byref(h)))
        return Ct(self, h)

    def matmul_pt(self, rows, w, bias, slots, padding):
        outs = self._outs(len(rows))
        self._ck(self.lib.fhelin_fc_matmul_pt(self.h, self._harr(rows), len(rows), w.h, bias.h if bias else None, slots, padding, outs))
        return self._cts(outs, len(rows))

    def matmulRE(self, rows, w, bias=None, row_size=128, padding=128):
        if isinstance(w, Ct):
            return self.matmul_ct(rows, w, row_size, padding)
        return self.matmul_pt(rows, w, bias, row_size, padding)

    def matmulCR(self, rows, w, bias=None):
        if isinstance(w, Ct):
            return self.matmul_ct(rows, w, 64, 1)
        return self.matmul_pt(rows, w, bias, 128, 1)

    def matmulCR_128(self, rows, w):
        return self.matmul_ct(rows, w, 128, 1)

    def matmul_ct(self, rows, w, slots, padding):
        outs = self._outs(len(rows))
        self._ck(self.lib.fhelin_fc_matmul_ct(self.h, self._harr(rows), len(rows), w.h, slots, padding, outs))
        return self._cts(outs, len(rows))

    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        outs = self._outs(len(rows))
        self._ck(self.lib.fhelin_fc_matmulRElarge(self.h, self._harr(rows), len(rows), self._harr(weights), len(weights),
                                                  bias.h if bias else None, float(mask_val), outs))
        return self._cts(outs, len(rows))

    def matmulCRlarge(self, rows, weights, bias):
        flat = [c for r in rows for c in r]
        outs = self._outs(len(rows))
        self._ck(self.lib.fhelin_fc_matmulCRlarge(self.h, self._harr(flat), len(rows), self._harr(weights), bias.h if bias else None, outs))
        return self._cts(outs, len(rows))

    def matmulScores(self, queries, key):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_fc_matmulScores(self.h, self._harr(queries), len(queries), key.h, C.byref(h)))
        return Ct(self, h)

    def wrapUpRepeated(self, v):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_fc_wrapUpRepeated(self.h, self._harr(v), len(v), C.byref(h)))
        return Ct(self, h)

    def wrapUpExpanded(self, v):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_fc_wrapUpExpanded(self.h, self._harr(v), len(v), C.byref(h)))
        return Ct(self, h)

    def unwrapExpanded(self, c, n):
        outs = self._outs(n)
        self._ck(self.lib.fhelin_fc_unwrapExpanded(self.h, c.h, n, outs))
        return self._cts(outs, n)

    def unwrapScoresExpanded(self, c, n):
        outs = self._outs(n)
        self._ck(self.lib.fhelin_fc_unwrapScoresExpanded(self.h, c.h, n, outs))
        return self._cts(outs, n)

    def unwrap_512_in_4_128(self, c, index):
        outs = self._outs(4)
        self._ck(self.lib.fhelin_fc_unwrap_512_in_4_128(self.h, c.h, index, outs))
        return self._cts(outs, 4)

    def unwrapRepeatedLarge(self, containers, input_number):
        outs = self._outs(4 * input_number)
        self._ck(self.lib.fhelin_fc_unwrapRepeatedLarge(self.h, self._harr(containers), len(containers), input_number, outs))
        flat = self._cts(outs, 4 * input_number)
        return [flat[4 * i: 4 * i + 4] for i in range(input_number)]

    def unwrapRepeatedLarge_range(self, containers, input_number, first, count):
        outs = self._outs(4 * count)
        self._ck(self.lib.fhelin_fc_unwrapRepeatedLarge_range(self.h, self._harr(containers), len(containers), input_number, first, count, outs))
        flat = self._cts(outs, 4 * count)
        return [flat[4 * i: 4 * i + 4] for i in range(count)]

    def ct_import_device(self, dptr, npoly, ell, deg, scale_hi, scale_lo, slots):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_ct_import_device(self.h, C.c_void_p(dptr), npoly, ell, deg, scale_hi, scale_lo, slots, C.byref(h)))
        return Ct(self, h)

    def generate_containers(self, inputs, bias=None):
        cap = (len(inputs) + 31) // 32
        outs = self._outs(cap)
        n = C.c_int32()
        self._ck(self.lib.fhelin_fc_generate_containers(self.h, self._harr(inputs), len(inputs), bias.h if bias else None, outs, C.byref(n)))
        return self._cts(outs, n.value)

    def wrap_containers(self, v, inputs_number):
        h = C.c_void_p()
        self._ck(self.lib.fhelin_fc_wrap_containers(self.h, self._harr(v), len(v), inputs_number, C.byref(h)))
        return Ct(self, h)


    # ---- the composite calls on a batch of samples (include/fhelin.h fhelin_fcb_*): flat handle lists are sample-major
    def fcb_matmulScores(self, queries, n, keys):
        """queries: B*n handles (sample-major), keys: B handles -> B handles"""
        outs = self._outs(len(keys))
        self._ck(self.lib.fhelin_fcb_matmulScores(self.h, self._harr(queries), n, self._harr(keys), len(keys), outs))
        return self._cts(outs, len(keys))

    def fcb_matmul_ct(self, rows, ws, slots, padding):
        outs = self._outs(len(rows))
        self._ck(self.lib.fhelin_fcb_matmul_ct(self.h, self._harr(rows), self._harr(ws), len(rows), slots, padding, outs))
        return self._cts(outs, len(rows))

    def fcb_wrapUpRepeated(self, v, n, B):
        outs = self._outs(B)
        self._ck(self.lib.fhelin_fcb_wrapUpRepeated(self.h, self._harr(v), n, B, outs))
        return self._cts(outs, B)

    def fcb_wrapUpExpanded(self, v, n, B):
        outs = self._outs(B)
        self._ck(self.lib.fhelin_fcb_wrapUpExpanded(self.h, self._harr(v), n, B, outs))
        return self._cts(outs, B)

    def fcb_unwrapExpanded(self, cs, n):
        outs = self._outs(len(cs) * n)
        self._ck(self.lib.fhelin_fcb_unwrapExpanded(self.h, self._harr(cs), len(cs), n, outs))
        return self._cts(outs, len(cs) * n)

    def fcb_unwrapRepeatedLarge(self, containers, nc, B, input_number):
        outs = self._outs(B * input_number * 4)
        self._ck(self.lib.fhelin_fcb_unwrapRepeatedLarge(self.h, self._harr(containers), nc, B, input_number, outs))
        return self._cts(outs, B * input_number * 4)

    def fcb_generate_containers(self, inputs, n, B, bias=None):
        per = (n + 31) // 32
        outs = self._outs(B * per)
        k = C.c_int32()
        self._ck(self.lib.fhelin_fcb_generate_containers(self.h, self._harr(inputs), n, B, bias.h if bias else None, outs, C.byref(k)))
        return self._cts(outs, B * k.value), k.value

    def rotsum_batch(self, v, slots, padding, repeat=False):
        return self._rows(self.lib.fhelin_fc_rotsum_batch, v, len(v), slots, padding, 1 if repeat else 0)

    def add_plain_batch(self, v, pt):
        return self._rows(self.lib.fhelin_add_plain_batch, v, len(v), pt.h)

    def eval_poly_batch(self, xs, coeffs):
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        return self._rows(self.lib.fhelin_eval_poly_batch, xs, len(xs), cf.ctypes.data_as(C.POINTER(C.c_double)), cf.size)

    def mult_many_batch(self, v, n, B):
        outs = self._outs(B)
        self._ck(self.lib.fhelin_mult_many_batch(self.h, self._harr(v), n, B, outs))
        return self._cts(outs, B)


class Pt:
    def __init__(self, eng, h):
        self.eng, self.h = eng, h

    def __del__(self):
        try:
            if self.h and self.eng.h:
                self.eng.lib.fhelin_pt_free(self.h)
        except Exception:
            pass
        self.h = None


class Ct:
    def __init__(self, eng, h):
        self.eng, self.h = eng, h

    def info(self):
        i = [C.c_int32() for _ in range(5)]
        sc = C.c_double()
        self.eng._ck(self.eng.lib.fhelin_ct_info(self.h, C.byref(i[0]), C.byref(i[1]), C.byref(i[2]), C.byref(i[3]), C.byref(sc), C.byref(i[4])))
        return dict(npoly=i[0].value, ell=i[1].value, level=i[2].value, deg=i[3].value, scale=sc.value, slots=i[4].value)

    @property
    def slots(self):
        return self.info()["slots"]

    @property
    def level(self):
        return self.info()["level"]

    def export(self):
        inf = self.info()
        out = np.empty((inf["npoly"], inf["ell"], self.eng.N), dtype=np.uint64)
        self.eng._ck(self.eng.lib.fhelin_ct_export(self.eng.h, self.h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def scale_parts(self):
        hi, lo = C.c_double(), C.c_double()
        self.eng._ck(self.eng.lib.fhelin_ct_scale(self.h, C.byref(hi), C.byref(lo)))
        return hi.value, lo.value

    def export_device(self, dptr, cap_words):
        self.eng._ck(self.eng.lib.fhelin_ct_export_device(self.eng.h, self.h, C.c_void_p(dptr), cap_words))

    def clone(self):
        return self.eng._un(self.eng.lib.fhelin_ct_clone, self)

    def free(self):
        if self.h and self.eng.h:
            self.eng.lib.fhelin_ct_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
