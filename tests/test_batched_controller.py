"""Host logic of the sample-batched driver (CPU): `BatchedController` carries B samples through the UNCHANGED driver
(fhe-linformer_amd/linformer.py = reference src/main.cpp:145-475 / src/main_2.cpp once per sample) and hands the samples' rows together
to the engine - here a stand-in engine on clear slot vectors (oracle/slotsim.py), so that every flattening / regrouping of the
sample-major handle lists is checked without a GPU: sample x of the batch must end in what its own single pass gives.
(The residues of the real engine are compared in tests/test_batched_forward_gpu.py.)"""
import numpy as np
import pytest

from oracle import circuit_sim as cs, plain_forward as pf, slotsim as S


class _H:
    """a ciphertext handle of the stand-in engine: a slot vector"""

    def __init__(self, v):
        self.v = np.asarray(v, dtype=np.float64)

    level = 0

    def clone(self):
        return _H(self.v.copy())


class SlotEngine:
    """the Engine methods BatchedController calls, on slot vectors; flat lists are sample-major exactly as the C ABI takes them"""
    lazy_heavy = True

    def __init__(self):
        self.sim = cs.SlotSimController()
        self.calls = []

    def _note(self, name, n):
        self.calls.append((name, n))

    def set_lane(self, k):
        self.lane_log = getattr(self, "lane_log", [])
        self.lane_log.append(k)

    def lanes_fork(self):
        pass

    def lanes_join(self):
        pass

    def encode(self, v, level=0, slots=0):
        return self.sim.encode(v)

    def encrypt_batch(self, rows, level=0, slots=0):
        return [_H(self.sim.encode(r)) for r in rows]

    def decrypt(self, h, slots=0):
        return h.v

    def add_batch(self, a, b):
        return [_H(x.v + y.v) for x, y in zip(a, b)]

    def add(self, a, b):
        return _H(a.v + b.v)

    def add_plain_batch(self, a, p):
        return [_H(x.v + p) for x in a]

    def mult_batch(self, a, b):
        self._note("mult_batch", len(a))
        return [_H(x.v * y.v) for x, y in zip(a, b)]

    def mult_plain_batch(self, a, p):
        return [_H(x.v * p) for x in a]

    def rotate_batch(self, a, i):
        return [_H(S.rot(x.v, i)) for x in a]

    def bootstrap(self, h):
        return _H(h.v)

    def rotsum_batch(self, a, slots, padding, repeat=False):
        self._note("rotsum_batch", len(a))
        return [_H(S.rotsum(x.v, slots, padding)) for x in a]

    def matmul_pt(self, rows, w, bias, slots, padding):
        self._note("matmul_pt", len(rows))
        return [_H(v) for v in S.matmul([r.v for r in rows], w, bias, slots, padding)]

    def fcb_matmul_ct(self, rows, ws, slots, padding):
        self._note("matmul_ct", len(rows))
        return [_H(S.matmul([r.v], w.v, None, slots, padding)[0]) for r, w in zip(rows, ws)]

    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        self._note("matmulRElarge", len(rows))
        return [_H(v) for v in S.matmulRElarge([r.v for r in rows], weights, bias, mask_val)]

    def matmulCRlarge(self, rows, weights, bias):
        self._note("matmulCRlarge", len(rows))
        return [_H(v) for v in S.matmulCRlarge([[c.v for c in r] for r in rows], weights, bias)]

    def fcb_matmulScores(self, queries, n, keys):
        B = len(keys)
        assert len(queries) == n * B
        return [_H(S.matmulScores([q.v for q in queries[x * n:(x + 1) * n]], keys[x].v)) for x in range(B)]

    def fcb_wrapUpRepeated(self, v, n, B):
        assert len(v) == n * B
        return [_H(S.wrapUpRepeated([h.v for h in v[x * n:(x + 1) * n]])) for x in range(B)]

    def fcb_wrapUpExpanded(self, v, n, B):
        assert len(v) == n * B
        return [_H(S.wrapUpExpanded([h.v for h in v[x * n:(x + 1) * n]])) for x in range(B)]

    def fcb_unwrapExpanded(self, cs_, n):
        return [_H(v) for c in cs_ for v in S.unwrapExpanded(c.v, n)]

    def fcb_unwrapRepeatedLarge(self, containers, nc, B, n):
        assert len(containers) == nc * B
        out = []
        for x in range(B):
            for four in S.unwrapRepeatedLarge([c.v for c in containers[x * nc:(x + 1) * nc]], n):
                out.extend(_H(v) for v in four)
        return out

    def fcb_generate_containers(self, inputs, n, B, bias=None):
        assert len(inputs) == n * B
        out, per = [], 0
        for x in range(B):
            c = S.generate_containers([h.v for h in inputs[x * n:(x + 1) * n]], bias)
            per = len(c)
            out.extend(_H(v) for v in c)
        return out, per

    def eval_poly_batch(self, xs, coeffs):
        return [_H(sum(c * x.v ** k for k, c in enumerate(coeffs))) for x in xs]

    def mult_many_batch(self, v, n, B):
        assert len(v) == n * B
        return [_H(np.prod([h.v for h in v[x * n:(x + 1) * n]], axis=0)) for x in range(B)]

    def eval_chebyshev(self, h, coeffs, a, b):
        return _H(pf.cheb_apply(np.asarray(coeffs), h.v, a, b))


@pytest.mark.parametrize("variant,S_tok,B", [("main", 129, 3), ("main_2", 130, 2)])
def test_batched_driver_gives_every_sample_its_own_single_pass(variant, S_tok, B):
    from fhe_linformer_amd import linformer as lf
    w = pf.synthetic_model(1234)
    samples = [pf.client_inputs(w, pf.synthetic_tokens(S_tok, 4321 + 7 * x)) for x in range(B)]
    eng = SlotEngine()
    ctl = lf.BatchedController(eng, B)
    encs = []
    for (x_in, X_E, X_F) in samples:
        rows = [X_E[i] for i in range(32)] + [X_F[i] for i in range(32)] + [x_in[i] for i in range(x_in.shape[0])]
        cts = [_H(lf.expanded(r)) for r in rows]
        encs.append({"inputs_E": cts[:32], "inputs_F": cts[32:64], "inputs": cts[64:]})
    out = lf.forward_encrypted(ctl, w, lf.batch_inputs(encs), None, variant)
    assert isinstance(out, lf.Batch) and len(out) == B
    got = [lf.logits_from_slots(v) for v in ctl.decrypt(out)]
    for x, smp in enumerate(samples):
        want = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *smp, None, variant))
        assert np.allclose(got[x], want, rtol=0, atol=1e-9), (x, np.max(np.abs(got[x] - want)))
    assert not np.allclose(got[0], got[1])          # the samples really differ
    # the row loops saw every sample's rows in ONE call (S tokens -> B * S rows)
    St = S_tok + 1
    assert ("matmul_pt", B * St) in eng.calls and ("matmulCRlarge", B * St) in eng.calls and ("matmul_pt", B * 32) in eng.calls


def test_laned_driver_splits_and_merges_the_sub_batches():
    """LanedBatchedController: every driver call once per sub-batch under its own lane, results merged in sample order"""
    from fhe_linformer_amd import linformer as lf
    B, S_tok = 4, 129
    w = pf.synthetic_model(1234)
    samples = [pf.client_inputs(w, pf.synthetic_tokens(S_tok, 500 + 3 * x)) for x in range(B)]
    eng = SlotEngine()
    ctl = lf.LanedBatchedController(eng, B, 2)
    encs = []
    for (x_in, X_E, X_F) in samples:
        rows = [X_E[i] for i in range(32)] + [X_F[i] for i in range(32)] + [x_in[i] for i in range(x_in.shape[0])]
        cts = [_H(lf.expanded(r)) for r in rows]
        encs.append({"inputs_E": cts[:32], "inputs_F": cts[32:64], "inputs": cts[64:]})
    ctl.begin()
    out = lf.forward_encrypted(ctl, w, lf.batch_inputs(encs))
    ctl.end()
    assert isinstance(out, lf.Batch) and len(out) == B
    got = [lf.logits_from_slots(v) for v in ctl.decrypt(out)]
    for x, smp in enumerate(samples):
        want = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *smp))
        assert np.allclose(got[x], want, rtol=0, atol=1e-9), x
    assert set(eng.lane_log) == {0, 1, 2} and eng.lane_log[0] == 0 and eng.lane_log[-1] == 0
    assert ("matmul_pt", 2 * (S_tok + 1)) in eng.calls          # a sub-batch of two samples per call


def test_batched_level_plan_from_a_single_sample_plan():
    from fhe_linformer_amd import linformer as lf
    single = [28] * 3 + [9] * 2 + [13, 7]          # 5 client sources, 2 later ones
    assert lf.batched_level_plan(single, 2, 5) == [28, 28, 28, 9, 9] * 2 + [13, 13, 7, 7]
