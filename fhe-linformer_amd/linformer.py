"""Host-side mirror of the reference's encrypted Linformer driver (reference src/main.cpp / src/main_2.cpp):
`encoder1()`, `pooler()`, `classifier()` with the SAME call sequence against a controller object that exposes
the reference's FHEController method names.  Two controllers exist: `GpuController` (this file; every operation
is a C-ABI call into libfhelin_amd.so) and `oracle.circuit_sim.SlotSimController` (plaintext slot vectors, test
infrastructure).  Text-file I/O of the reference (`read_*`, src/FHEController.cpp:501-698) is replaced by
in-memory arrays with the same three packing layouts (plain / repeated / expanded).

Nothing here computes on residues: this module only sequences operations (like main.cpp does)."""
import math

import numpy as np

SLOTS = 16384


# ---- offline weight helpers (reference src/python/split_ffn_w1.py:24-37, split_ffn_w2_cols.py:22-29) -------
def split_transposed_blocks(W0, cols_per_block=128):
    WT = np.asarray(W0).T                                    # [128, 512]
    return [WT[:, i:i + cols_per_block] for i in range(0, WT.shape[1], cols_per_block)]


def split_col_blocks(W2, cols_per_block=128):
    W2 = np.asarray(W2)                                       # [128, 512]
    return [W2[:, i:i + cols_per_block] for i in range(0, W2.shape[1], cols_per_block)]


def expanded(v, num_inputs=128):                             # read_expanded_input / read_plain_expanded_input :623-698
    out = np.zeros(SLOTS)
    v = np.asarray(v, dtype=np.float64)
    for j in range(128):
        out[j * 128: j * 128 + num_inputs] = v[j]
    return out


def repeated(v):                                             # read_plain_repeated_input :582-603
    return np.tile(np.asarray(v, dtype=np.float64)[:128], 128)


def cheb_coeffs(f, a, b, degree):
    """EvalChebyshevCoefficients (what include/FHEController.h::chebyshev does in C++)"""
    n = degree + 1
    j = np.arange(n)
    nodes = np.cos(np.pi * (j + 0.5) / n)
    fx = np.array([f(0.5 * (b - a) * t + 0.5 * (b + a)) for t in nodes])
    return np.array([2.0 / n * np.sum(fx * np.cos(np.pi * k * (j + 0.5) / n)) for k in range(n)])


class GpuController:
    """The reference's FHEController surface on top of fhe_linformer_amd.Engine."""

    def __init__(self, eng, verbose=False):
        self.e, self.verbose = eng, verbose
        self.num_slots = SLOTS
        self.n_boot = 0

    # handles
    def level(self, c):
        return c.level

    def clone(self, c):
        return c.clone()

    # encode / encrypt
    def encode(self, v, level=0):
        if np.isscalar(v):
            v = np.full(SLOTS, float(v))
        return self.e.encode(np.asarray(v, dtype=np.float64), level, SLOTS)

    def encrypt(self, v, level=0):
        return self.e.encrypt(np.asarray(v, dtype=np.float64), level, SLOTS)

    def decrypt(self, c):
        return self.e.decrypt(c, SLOTS)

    def read_expanded_input(self, v, scale=1.0):
        return self.encrypt(expanded(np.asarray(v) * scale), 0)

    def read_expanded_inputs(self, rows, scale=1.0):
        """all inputs of a sample in one batched client call (encode + sample + encrypt on the GPU)"""
        return self.e.encrypt_batch(np.stack([expanded(np.asarray(v) * scale) for v in rows]), 0, SLOTS)

    def client_ingest(self, w, x_emb):
        return self.e.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], emb=x_emb)

    def read_plain_input(self, m, level=0, scale=1.0):
        return self.encode(np.asarray(m, dtype=np.float64).reshape(-1) * scale, level)

    def read_plain_repeated_input(self, v, level=0, scale=1.0):
        return self.encode(repeated(v) * scale, level)

    def read_plain_expanded_input(self, v, level=0, scale=1.0):
        return self.encode(expanded(np.asarray(v) * scale), level)

    # leaf ops
    def add(self, a, b):
        return self.e.add(a, b)

    def mult(self, a, b):
        return self.e.mult_const(a, b) if np.isscalar(b) else self.e.mult(a, b)

    def rotate(self, a, i):
        return self.e.rotate(a, i)

    def bootstrap(self, a):
        self.n_boot += 1
        return self.e.bootstrap(a)

    # composites: same names as the reference
    def rotsum(self, a, slots, padding):
        return self.e.rotsum(a, slots, padding)

    def matmulRE(self, rows, w, bias=None, row_size=128, padding=128):
        return self.e.matmulRE(rows, w, bias, row_size, padding)

    def matmulCR(self, rows, w, bias=None):
        return self.e.matmulCR(rows, w, bias)

    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        return self.e.matmulRElarge(rows, weights, bias, mask_val)

    def matmulCRlarge(self, rows, weights, bias):
        return self.e.matmulCRlarge(rows, weights, bias)

    def matmulScores(self, queries, key):
        return self.e.matmulScores(queries if isinstance(queries, list) else [queries], key)

    def wrapUpRepeated(self, v):
        return self.e.wrapUpRepeated(v)

    def wrapUpExpanded(self, v):
        return self.e.wrapUpExpanded(v)

    def unwrapExpanded(self, c, n):
        return self.e.unwrapExpanded(c, n)

    def unwrapRepeatedLarge(self, cs, n):
        return self.e.unwrapRepeatedLarge(cs, n)

    def unwrapRepeatedLarge_range(self, cs, n, first, count):
        return self.e.unwrapRepeatedLarge_range(cs, n, first, count)

    def generate_containers(self, inputs, bias=None):
        return self.e.generate_containers(inputs, bias)

    # activations (reference src/FHEController.cpp:1289-1336)
    def eval_exp(self, c, inputs_number):
        res = self.e.eval_poly(c, [1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0])
        res = self.e.mult_many([res] * 8)
        i = np.arange(SLOTS)
        mask = np.where((i % 128 < inputs_number) & (i < 128 * inputs_number), 0.0, -1.0)
        return self.e.add(res, self.encode(mask, res.level))

    _fits = {}   # Chebyshev fits are data-independent: one fit per (function, interval, degree), as a C++ driver's static table

    def _cheb(self, key, f, c, a, b, degree):
        key = key + (a, b, degree)
        if key not in GpuController._fits:
            GpuController._fits[key] = cheb_coeffs(f, a, b, degree)
        return self.e.eval_chebyshev(c, GpuController._fits[key], a, b)

    def eval_inverse_naive(self, c, lo, hi):
        return self._cheb(("inv",), lambda x: 1.0 / x, c, lo, hi, 119)

    def eval_gelu_function(self, c, lo, hi, mult, degree):
        return self._cheb(("gelu", mult), lambda x: 0.5 * (x / mult) * (1 + math.erf((x / mult) / 1.41421356237)), c, lo, hi, degree)

    def eval_tanh_function(self, c, lo, hi, mult, degree):
        return self._cheb(("tanh", mult), lambda x: math.tanh(x / mult), c, lo, hi, degree)


class DataflowController:
    """ONE sample, its independent branches on different lanes.  The reference's driver is a straight-line program, but not a chain: the
    value projection (src/main.cpp:212-213: 32 rows of 7-step trees, large launches) does not depend on the scores chain that the driver
    writes in front of it (:196-207: single-ciphertext steps, each a handful of launches that cannot fill the GPU).  This wrapper puts
    every call on the lane (HIP stream + memory arena of the one context, include/fhelin.h fhelin_ctx_set_lane) of the ciphertexts it
    reads: a call whose inputs carry no lane yet - fresh encryptions, rows nobody has evaluated - starts a new branch on the next lane; a
    call that reads values of two branches runs on the first one's lane after fhelin_ctx_lane_wait on the other.  The host issues both
    branches back to back (3.4 us per launch, it never waits for the GPU), the GPU runs them side by side.  No knowledge of the driver: the
    rule is data flow.  Scheduling only - the residues of the plain single pass (tests/test_batched_forward_gpu.py).
    Use: ctl.begin() ... forward_encrypted(ctl, ...) ... ctl.end()."""

    # calls whose outputs are deferred rows when they return more than one (capi_composite.cpp): evaluated later, under the lane that reads them
    _DEFERRED = ("matmulRE", "matmulCR", "unwrapExpanded", "matmulRElarge")
    BULK_ROWS = 8        # a call that reads this many ciphertexts is a row loop: its launches fill the GPU
    _PLAIN = ("level", "encode", "read_plain_input", "read_plain_repeated_input", "read_plain_expanded_input", "decrypt")

    def __init__(self, inner, lanes=2):
        self.c, self.e, self.lanes = inner, inner.e, lanes
        self.num_slots = inner.num_slots
        self._next = 0
        self.branches = 0

    @property
    def n_boot(self):
        return self.c.n_boot

    def begin(self):
        self.e.set_lane(0)
        self.e.lanes_fork()
        self._next = 0

    def end(self):
        self.e.set_lane(0)
        self.e.lanes_join()

    def _cts(self, x, out):
        if hasattr(x, "info") and hasattr(x, "export"):
            out.append(x)
        elif isinstance(x, (list, tuple)):
            for v in x:
                self._cts(v, out)

    def __getattr__(self, name):
        fn = getattr(self.c, name)
        if name in DataflowController._PLAIN or not callable(fn):
            if name == "decrypt":
                def dec(c):
                    self.e.set_lane(0)
                    return fn(c)
                return dec
            if name == "level":
                def lev(c):            # reading the level may evaluate a deferred value: under its own lane
                    if getattr(c, "lane", None) is not None:
                        self.e.set_lane(c.lane)
                    return fn(c)
                return lev
            return fn

        def call(*a, **kw):
            ins = []
            self._cts(a, ins)
            lanes = []
            for h in ins:
                l = getattr(h, "lane", None)
                if l is not None and l not in lanes:
                    lanes.append(l)
            if lanes:
                lane = lanes[0]
                self.e.set_lane(lane)
            else:                                   # a new branch: nothing it reads has been computed under a lane
                lane = 1 + self._next % self.lanes
                self._next += 1
                self.branches += 1
                self.e.set_lane(lane)
                # ... held back until the other lanes' last BULK call has drained: beside a row loop that fills the GPU by itself a
                # second one gains nothing; beside the single-ciphertext steps that FOLLOW it there, it fills the CUs they leave idle
                for l in range(1, self.lanes + 1):
                    if l != lane:
                        self.e.lane_wait_mark(l)
            for l in lanes[1:]:
                self.e.lane_wait(l)
            r = fn(*a, **kw)
            if len(ins) >= DataflowController.BULK_ROWS:
                self.e.lane_mark()
            outs = []
            self._cts(r, outs)
            deferred = name in DataflowController._DEFERRED and len(outs) > 1 and not lanes and not any(_is_ct(v) for v in a[1:2])
            for h in outs:
                if getattr(h, "lane", None) is None:
                    h.lane = None if deferred else lane
            return r
        return call


def _is_ct(x):
    return hasattr(x, "info") and hasattr(x, "export")


class Batch(list):
    """one value of the driver for each of B samples (B ciphertext handles, sample order)"""


class BatchedController:
    """B samples through ONE engine in lock step (BASELINE config 4's per-GPU unit: "the batch of independent input ciphertexts").

    The driver below (`encoder1` / `pooler` / `classifier` = reference src/main.cpp:145-475 once per sample; samples never meet) runs
    UNCHANGED: where it holds one ciphertext it now holds a `Batch` of B, where it holds a vector of rows it holds a list of Batches,
    and every method hands the samples' rows TOGETHER to one entry point of the C ABI - the row loops concatenated into the batched
    calls that exist for them (fhelin_fc_matmul_pt, _matmulRElarge, _matmulCRlarge, fhelin_bootstrap_batch, fhelin_eval_chebyshev_batch,
    fhelin_mult_batch, ...), the calls that fold a row set into one ciphertext through their sample-batched forms (include/fhelin.h
    fhelin_fcb_*).  One key set, one plaintext / mask cache (weights are read once, not per sample), one level plan, one launch set:
    the single-ciphertext chains of a pass (scores, Taylor exp, 1/x, pooler, tanh, classifier) are B rows wide, and the 8 bootstraps
    of a sample run in batches of 2B, 5B and B.  Sample x ends in EXACTLY the residues its own single pass gives
    (tests/test_batched_forward_gpu.py).  All samples of a batch have one token count."""

    def __init__(self, eng, B, verbose=False):
        self.e, self.B, self.verbose = eng, int(B), verbose
        self.num_slots = SLOTS
        self.n_boot = 0

    # ---- layout helpers: a list of n Batches <-> the flat sample-major handle list the C ABI takes
    def _flat(self, rows):
        return [r[x] for x in range(self.B) for r in rows]

    def _unflat(self, flat, n):
        return [Batch(flat[x * n + i] for x in range(self.B)) for i in range(n)]

    # handles
    def level(self, c):
        return c[0].level

    def clone(self, c):
        return Batch(h.clone() for h in c)

    # encode / encrypt: plaintexts are shared by all samples
    def encode(self, v, level=0):
        if np.isscalar(v):
            v = np.full(SLOTS, float(v))
        return self.e.encode(np.asarray(v, dtype=np.float64), level, SLOTS)

    def encrypt(self, v, level=0):
        """the server-side encryptions of the driver (src/main.cpp:220, :472): one fresh encryption per sample"""
        v = np.asarray(v, dtype=np.float64)
        return Batch(self.e.encrypt_batch(np.stack([v] * self.B), level, SLOTS))

    def decrypt(self, c):
        return [self.e.decrypt(h, SLOTS) for h in c]

    def read_expanded_inputs_batch(self, rows_per_sample, scale=1.0):
        """per sample the rows of read_expanded_inputs -> list of Batches (row-major)"""
        cts = [self.e.encrypt_batch(np.stack([expanded(np.asarray(v) * scale) for v in rows]), 0, SLOTS) for rows in rows_per_sample]
        return [Batch(cts[x][i] for x in range(self.B)) for i in range(len(cts[0]))]

    def client_ingest_batch(self, w, x_embs):
        encs = [self.e.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], emb=x) for x in x_embs]
        return batch_inputs(encs)

    read_plain_input = GpuController.read_plain_input
    read_plain_repeated_input = GpuController.read_plain_repeated_input
    read_plain_expanded_input = GpuController.read_plain_expanded_input

    # leaf ops
    def add(self, a, b):
        if isinstance(b, Batch):
            # one fhelin_add per handle, as the single pass makes them: the C ABI defers EvalAdd(ct, ct) and runs everything pending as ONE
            # batched level adjustment + addition when a result is first read - the driver's loop over the S residual additions
            # (src/main.cpp:237-239) times B samples in one go, instead of S calls of B pairs
            if self.e.lazy_heavy:
                return Batch(self.e.add(x, y) for x, y in zip(a, b))
            return Batch(self.e.add_batch(list(a), list(b)))
        return Batch(self.e.add_plain_batch(list(a), b))

    def mult(self, a, b):
        if np.isscalar(b):
            return Batch(self.e.mult_const(h, b) for h in a)
        if isinstance(b, Batch):
            return Batch(self.e.mult_batch(list(a), list(b)))
        return Batch(self.e.mult_plain_batch(list(a), b))

    def rotate(self, a, i):
        return Batch(self.e.rotate_batch(list(a), i))

    def bootstrap(self, a):
        # one call per handle, as the single pass makes them: the C ABI defers bootstraps / Chebyshev evaluations and runs everything
        # pending as ONE batch when a result is first read (fhelin_bootstrap_batch) - here the driver's loop over containers
        # (src/main.cpp:354-358) times B samples: 5B wide.  With deferral off (FHELIN_LAZY_HEAVY=0) the B samples still share a batch.
        self.n_boot += 1
        if self.e.lazy_heavy:
            return Batch(self.e.bootstrap(h) for h in a)
        return Batch(self.e.bootstrap_batch(list(a)))

    # composites: same names as the reference
    def rotsum(self, a, slots, padding):
        return Batch(self.e.rotsum_batch(list(a), slots, padding))

    def matmulRE(self, rows, w, bias=None, row_size=128, padding=128):
        if isinstance(w, Batch):      # ciphertext weight: sample x's rows against sample x's weight
            flat = self._flat(rows)
            ws = [w[x] for x in range(self.B) for _ in rows]
            return self._unflat(self.e.fcb_matmul_ct(flat, ws, row_size, padding), len(rows))
        return self._unflat(self.e.matmul_pt(self._flat(rows), w, bias, row_size, padding), len(rows))

    def matmulCR(self, rows, w, bias=None):
        if isinstance(w, Batch):
            flat = self._flat(rows)
            return self._unflat(self.e.fcb_matmul_ct(flat, [w[x] for x in range(self.B) for _ in rows], 64, 1), len(rows))
        return self._unflat(self.e.matmul_pt(self._flat(rows), w, bias, 128, 1), len(rows))

    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        return self._unflat(self.e.matmulRElarge(self._flat(rows), weights, bias, mask_val), len(rows))

    def matmulCRlarge(self, rows, weights, bias):
        flat = [[r[k][x] for k in range(4)] for x in range(self.B) for r in rows]
        return self._unflat(self.e.matmulCRlarge(flat, weights, bias), len(rows))

    def matmulScores(self, queries, key):
        q = queries if not isinstance(queries, Batch) else [queries]
        return Batch(self.e.fcb_matmulScores(self._flat(q), len(q), list(key)))

    def wrapUpRepeated(self, v):
        return Batch(self.e.fcb_wrapUpRepeated(self._flat(v), len(v), self.B))

    def wrapUpExpanded(self, v):
        return Batch(self.e.fcb_wrapUpExpanded(self._flat(v), len(v), self.B))

    def unwrapExpanded(self, c, n):
        return self._unflat(self.e.fcb_unwrapExpanded(list(c), n), n)

    def unwrapRepeatedLarge(self, cs, n):
        flat = self.e.fcb_unwrapRepeatedLarge(self._flat(cs), len(cs), self.B, n)
        return [[Batch(flat[(x * n + i) * 4 + k] for x in range(self.B)) for k in range(4)] for i in range(n)]

    def generate_containers(self, inputs, bias=None):
        flat, per = self.e.fcb_generate_containers(self._flat(inputs), len(inputs), self.B, bias)
        return self._unflat(flat, per)

    # activations (reference src/FHEController.cpp:1289-1336)
    def eval_exp(self, c, inputs_number):
        res = self.e.eval_poly_batch(list(c), [1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0])
        res = self.e.mult_many_batch([r for r in res for _ in range(8)], 8, self.B)
        i = np.arange(SLOTS)
        mask = np.where((i % 128 < inputs_number) & (i < 128 * inputs_number), 0.0, -1.0)
        return Batch(self.e.add_plain_batch(res, self.encode(mask, res[0].level)))

    def _cheb(self, key, f, c, a, b, degree):
        key = key + (a, b, degree)
        if key not in GpuController._fits:
            GpuController._fits[key] = cheb_coeffs(f, a, b, degree)
        if self.e.lazy_heavy:
            return Batch(self.e.eval_chebyshev(h, GpuController._fits[key], a, b) for h in c)
        return Batch(self.e.eval_chebyshev_batch(list(c), GpuController._fits[key], a, b))

    eval_inverse_naive = GpuController.eval_inverse_naive
    eval_gelu_function = GpuController.eval_gelu_function
    eval_tanh_function = GpuController.eval_tanh_function


class LanedBatchedController:
    """B samples through ONE engine as `lanes` sub-batches that run CONCURRENTLY on the GPU (include/fhelin.h fhelin_ctx_set_lane): every
    call of the driver is made once per sub-batch, each under its own lane (HIP stream + memory arena) of the one context, from the
    one host thread.  Same keys, same plaintext cache, same level plan as `BatchedController` - and the same residues per sample
    (scheduling only) - but while one lane's launch drains, the other lane's fills the CUs: what two engines "in flight" gained in
    round 3 at the price of a second key set.  Sub-batch k holds samples [k B/lanes, (k+1) B/lanes).
    Use: ctl.begin() (lanes wait for the inputs) ... forward_encrypted(ctl, ...) ... ctl.end() (main stream waits for the lanes)."""

    _ONCE = ("level", "encode", "read_plain_input", "read_plain_repeated_input", "read_plain_expanded_input")

    def __init__(self, eng, B, lanes=2):
        assert B % lanes == 0 and lanes >= 1
        self.e, self.B, self.lanes, self.h = eng, B, lanes, B // lanes
        self.sub = [BatchedController(eng, self.h) for _ in range(lanes)]
        self.num_slots = SLOTS

    @property
    def n_boot(self):
        return self.sub[0].n_boot

    def begin(self):
        self.e.set_lane(0)
        self.e.lanes_fork()

    def end(self):
        self.e.set_lane(0)
        self.e.lanes_join()

    def _split(self, x, k):
        if isinstance(x, Batch):
            return Batch(x[k * self.h:(k + 1) * self.h])
        if isinstance(x, (list, tuple)):
            return [self._split(v, k) for v in x]
        return x

    def _merge(self, parts):
        r0 = parts[0]
        if isinstance(r0, Batch):
            return Batch(h for p in parts for h in p)
        if isinstance(r0, list):
            return [self._merge([p[i] for p in parts]) for i in range(len(r0))]
        return r0

    def __getattr__(self, name):
        if name in LanedBatchedController._ONCE:          # plaintexts / levels: the same for every sample, made once
            def once(*a, **kw):
                self.e.set_lane(1)
                return getattr(self.sub[0], name)(*[self._split(v, 0) for v in a], **kw)
            return once

        def call(*a, **kw):
            parts = []
            for k in range(self.lanes):
                self.e.set_lane(k + 1)
                parts.append(getattr(self.sub[k], name)(*[self._split(v, k) for v in a], **kw))
            return self._merge(parts)
        return call

    def decrypt(self, c):
        self.e.set_lane(0)
        return [self.e.decrypt(h, SLOTS) for h in c]


def batched_level_plan(plan, B, n_client):
    """the level plan of a batched pass from the plan recorded on ONE sample (Engine.level_plan_end): the sources of a batched pass
    are the B samples' client encryptions (sample-major, n_client each) and then, call by call, B sources per source of the single pass"""
    return [t for _ in range(B) for t in plan[:n_client]] + [t for t in plan[n_client:] for _ in range(B)]


def batch_inputs(encs):
    """B per-sample input dicts (encrypt_inputs / ingest_sample) -> one dict of lists of Batches"""
    B = len(encs)
    return {k: [Batch(encs[x][k][i] for x in range(B)) for i in range(len(encs[0][k]))] for k in ("inputs_E", "inputs_F", "inputs")}


# ---- the circuit: reference src/main.cpp:145-475 (CLS-query variant, as built) -----------------------------
def encrypt_inputs(ctl, x_in, X_E, X_F):
    """client side (main.cpp:159-173): x_in [S_total,128] (row 0 = CLS token), X_E / X_F [32,128] (Linformer projections)"""
    rows = [X_E[i] for i in range(32)] + [X_F[i] for i in range(32)] + [x_in[i] for i in range(x_in.shape[0])]
    if hasattr(ctl, "read_expanded_inputs"):
        cts = ctl.read_expanded_inputs(rows)            # one batched call instead of 194 read_expanded_input calls
    else:
        cts = [ctl.read_expanded_input(r) for r in rows]
    return {"inputs_E": cts[:32],                                                           # main.cpp:159-162
            "inputs_F": cts[32:64],                                                         # :164-167
            "inputs": cts[64:]}                                                             # :169-173


def ingest_sample(ctl, w, x_emb):
    """client side of one sample from its token embeddings x_emb [S,128] (dimReduce.py:141-160 + main.cpp:159-173).  A controller
    with `client_ingest` (the GPU engine: fhelin_client_ingest) does positional embedding, both Linformer projections, packing,
    encoding and encryption on the device; any other controller gets the NumPy statement of the same lines."""
    if hasattr(ctl, "client_ingest"):
        return ctl.client_ingest(w, x_emb)
    S = x_emb.shape[0]
    x_in = np.vstack([np.asarray(w["cls_token"]).reshape(1, -1), x_emb + w["posEmb"][:S] / 3.0])
    St = x_in.shape[0]
    X_E = w["E_w"][:, :St] @ x_in + w["E_b"].reshape(-1, 1)
    X_F = w["F_w"][:, :St] @ x_in + w["F_b"].reshape(-1, 1)
    return encrypt_inputs(ctl, x_in, X_E, X_F)


def encoder1(ctl, w, enc, trace=None, full_attention=False):
    t = trace if trace is not None else {}
    inputs_E, inputs_F, inputs = enc["inputs_E"], enc["inputs_F"], enc["inputs"]
    S = len(inputs)

    query_w = ctl.read_plain_input(w["WQ"].T)                                               # ..._WQ_weight_T.txt :177
    query_b = ctl.read_plain_repeated_input(w["BQ"])
    key_w = ctl.read_plain_input(w["WK"].T)
    key_b = ctl.read_plain_repeated_input(w["BK"])
    Q = ctl.matmulRE(inputs, query_w, query_b)                                              # :183
    K = ctl.matmulRE(inputs_E, key_w, key_b)                                                # :184
    K_wrapped = ctl.wrapUpRepeated(K)                                                       # :186
    t["K_wrapped"] = K_wrapped
    value_w = ctl.read_plain_input(w["WV"].T)
    value_b = ctl.read_plain_repeated_input(w["BV"])
    if not full_attention:
        # src/main.cpp:196-224 — attention for the CLS query only; the other tokens' attention output is an encrypted zero
        q0 = t["Q0"] = Q[0]                                 # (the trace keeps handles for the tests; unread handles cost nothing)
        scores = ctl.matmulScores(q0, K_wrapped)                                            # :196
        t["scores"] = scores
        scores = ctl.eval_exp(scores, 32)                                                   # :197
        t["exp"] = scores
        scores_sum = ctl.rotsum(scores, 32, 128)                                            # :201
        scores_denominator = ctl.eval_inverse_naive(scores_sum, -1, 128)                    # :203
        scores = ctl.mult(scores, scores_denominator)                                       # :205
        unwrapped_scores = ctl.unwrapExpanded(scores, 1)                                    # :207
        V = ctl.matmulRE(inputs_F, value_w, value_b)                                        # :212
        V_wrapped = ctl.wrapUpRepeated(V)
        cls_output = ctl.matmulRE(unwrapped_scores, V_wrapped, None, 128, 128)[0]           # :215
        t["self_attention"] = cls_output
        output = [cls_output]
        zero_c = ctl.encrypt(np.zeros(SLOTS), ctl.level(cls_output))                        # :220-221
        for _ in range(1, S):
            output.append(ctl.clone(zero_c))
        dense_bias_in_matmul = False
    else:
        # src/main_2.cpp:187-229 — attention for every token, queries wrapped 128 at a time
        Q_1, Q_2 = Q[:128], Q[128:]
        scores_1 = ctl.eval_exp(ctl.matmulScores(Q_1, K_wrapped), len(Q_1))                 # main_2.cpp:196-197
        scores_2 = ctl.eval_exp(ctl.matmulScores(Q_2, K_wrapped), len(Q_2))                 # :199-200
        t["scores"], t["exp"] = scores_1, scores_2
        den_1 = ctl.eval_inverse_naive(ctl.rotsum(scores_1, 32, 128), -1, 190000)           # :202-212
        den_2 = ctl.eval_inverse_naive(ctl.rotsum(scores_2, 32, 128), -1, 190000)
        scores_1 = ctl.mult(scores_1, den_1)
        scores_2 = ctl.mult(scores_2, den_2)
        unwrapped_scores = ctl.unwrapExpanded(scores_1, 128) + ctl.unwrapExpanded(scores_2, S - 128)   # :216-221
        V = ctl.matmulRE(inputs_F, value_w, value_b)                                        # :226
        V_wrapped = ctl.wrapUpRepeated(V)
        output = ctl.matmulRE(unwrapped_scores, V_wrapped, None, 128, 128)                  # :229
        t["self_attention"] = output[0]
        dense_bias_in_matmul = True

    dense_w = ctl.read_plain_input(w["WO"], ctl.level(output[0]))                           # ..._WO_weight.txt :231
    dense_b = ctl.read_plain_expanded_input(w["BO"], ctl.level(output[0]) + 1)
    if dense_bias_in_matmul:
        output = ctl.matmulCR(output, dense_w, dense_b)                                     # main_2.cpp:241
    else:
        output = ctl.matmulCR(output, dense_w, None)                                        # main.cpp:235
        output[0] = ctl.add(output[0], dense_b)
    output = [ctl.add(output[i], inputs[i]) for i in range(S)]                              # :237-239

    fL1 = w["c10"] + w["c11"] / math.sqrt(S) + w["c12"] / S                                 # :290-291
    output_0, output_1 = output[:128], output[128:]                                         # :293-300
    wrapped_0 = ctl.wrapUpExpanded(output_0)
    wrapped_1 = ctl.wrapUpExpanded(output_1)
    a1 = ctl.read_plain_repeated_input(w["a1"], ctl.level(wrapped_0), fL1)
    b1 = ctl.read_plain_repeated_input(w["b1"], ctl.level(wrapped_0) + 1, fL1)
    wrapped_0 = ctl.add(ctl.mult(wrapped_0, a1), b1)                                        # :308-311
    wrapped_1 = ctl.add(ctl.mult(wrapped_1, a1), b1)
    t["affine1_0"] = wrapped_0
    wrapped_0 = ctl.bootstrap(wrapped_0)                                                    # :313-314
    wrapped_1 = ctl.bootstrap(wrapped_1)
    copy_0, copy_1 = ctl.clone(wrapped_0), ctl.clone(wrapped_1)
    output_0 = ctl.unwrapExpanded(wrapped_0, 128)                                           # :319-320
    output_1 = ctl.unwrapExpanded(wrapped_1, S - 128)

    gelu_scale = 1.0 / 8.0                                                                  # :329
    blocks = split_transposed_blocks(w["Wffn0"])                                            # ffn_W0_transposed_block_k.txt
    dense_weights = [ctl.read_plain_input(b, ctl.level(wrapped_0), gelu_scale) for b in blocks]
    inter_bias = ctl.read_plain_input(w["Bffn0"], ctl.level(wrapped_0) + 1, gelu_scale)
    output_0 = ctl.matmulRElarge(output_0, dense_weights, inter_bias)                       # :341-342
    output_1 = ctl.matmulRElarge(output_1, dense_weights, inter_bias)
    outputs_raw = output_0 + output_1
    outputs = ctl.generate_containers(outputs_raw, None)                                    # :352
    for i in range(len(outputs)):                                                           # :354-358
        outputs[i] = ctl.eval_gelu_function(outputs[i], -1, 1, gelu_scale, 119)
        outputs[i] = ctl.bootstrap(outputs[i])
    unwrapped_large = ctl.unwrapRepeatedLarge(outputs, S)                                   # :360

    lvl = ctl.level(unwrapped_large[0][0])
    out_w = [ctl.read_plain_input(b, lvl) for b in split_col_blocks(w["Wffn2"])]            # ffn_W2_block_k.txt :367-370
    out_bias = ctl.read_plain_expanded_input(w["Bffn2"], lvl + 1)
    output = ctl.matmulCRlarge(unwrapped_large, out_w, out_bias)                            # :374
    output_2, output_3 = output[:128], output[128:]
    wrapped_2 = ctl.add(ctl.wrapUpExpanded(output_2), copy_0)                               # :386-390
    wrapped_3 = ctl.add(ctl.wrapUpExpanded(output_3), copy_1)
    S1 = len(output)
    fL2 = w["c20"] + w["c21"] / math.sqrt(S1) + w["c22"] / S1
    a2 = ctl.read_plain_repeated_input(w["a2"], ctl.level(wrapped_2), fL2)
    b2 = ctl.read_plain_repeated_input(w["b2"], ctl.level(wrapped_3) + 1, fL2)
    wrapped_2 = ctl.add(ctl.mult(wrapped_2, a2), b2)                                        # :404-408
    wrapped_3 = ctl.add(ctl.mult(wrapped_3, a2), b2)
    output_2 = ctl.unwrapExpanded(wrapped_2, 128)                                           # :410-411
    output_3 = ctl.unwrapExpanded(wrapped_3, S - 128)
    t["encoder_out"] = output_2[0]
    return output_2[0]


def pooler(ctl, w, x, trace=None, tanh_scale=1.0 / 50):                                     # main.cpp:427-451 (main_2.cpp:385: 1/18)
    weight = ctl.read_plain_input(w["Wp"].T, ctl.level(x), tanh_scale)                      # pooler_dense_weight_T.txt
    bias = ctl.read_plain_repeated_input(w["bp"], ctl.level(x) + 1, tanh_scale)
    out = ctl.mult(x, weight)
    out = ctl.rotsum(out, 128, 128)
    out = ctl.add(out, bias)
    out = ctl.bootstrap(out)
    out = ctl.eval_tanh_function(out, -1, 1, tanh_scale, 300)
    if trace is not None:
        trace["pooled"] = out
    return out


def classifier(ctl, w, x, encrypted_mask=True):                                             # main.cpp:453-475
    fc = np.zeros((128, 128))
    fc[:20] = w["fc_w"]                                                                     # fcLinear_0_weight.txt: 20 rows
    weight = ctl.read_plain_input(fc, ctl.level(x))
    bias = ctl.read_plain_expanded_input(np.concatenate([w["fc_b"], np.zeros(108)]), ctl.level(x))
    out = ctl.mult(x, weight)
    out = ctl.rotsum(out, 128, 1)
    out = ctl.add(out, bias)
    mask = np.zeros(SLOTS)
    mask[np.arange(20) * 128] = 1
    if encrypted_mask:
        return ctl.mult(out, ctl.encrypt(mask, ctl.level(out)))                             # encrypted mask (quirk Q8)
    return ctl.mult(out, ctl.encode(mask, ctl.level(out)))                                  # main_2.cpp:427: plaintext mask


def forward_encrypted(ctl, w, enc, trace=None, variant="main"):
    """server side of one sample: encoder1 -> pooler -> classifier -> logits at slots {0,128,...,19*128} (main.cpp:105-123).
    variant "main" = src/main.cpp as built (CLS-query attention); "main_2" = src/main_2.cpp (full attention,
    tanh scale 1/18, plaintext output mask)."""
    full = variant == "main_2"
    out = encoder1(ctl, w, enc, trace, full_attention=full)
    pooled = pooler(ctl, w, out, trace, tanh_scale=1.0 / 18 if full else 1.0 / 50)
    return classifier(ctl, w, pooled, encrypted_mask=not full)


def forward(ctl, w, x_in, X_E, X_F, trace=None, variant="main"):
    return forward_encrypted(ctl, w, encrypt_inputs(ctl, x_in, X_E, X_F), trace, variant)


def logits_from_slots(v):
    return np.asarray(v)[np.arange(20) * 128]
