"""B samples through ONE engine (BASELINE config 4's per-GPU unit), residue for residue.

`BatchedController` (fhe-linformer_amd/linformer.py) runs the unchanged driver (reference src/main.cpp:145-475 once per sample; samples
never meet) with every value B ciphertexts wide: the row loops of all samples concatenated into the batched entry points, the
single-ciphertext chains B rows wide, the bootstraps in batches of 2B / 5B / B.  Rows of a batched key switch are independent, so
sample x must end in EXACTLY the residues its own single pass gives:
  * B = 4 batched - in one launch set (BatchedController) and as two sub-batches on two lanes of the context that run concurrently
    (LanedBatchedController) - == four single passes on the same input ciphertexts (the server-side fresh encryptions - the encrypted zero,
    src/main.cpp:220, and the encrypted mask, :472 - replayed), final ciphertext and six traced intermediates, N=2^15;
  * B = 2 batched, one of the two samples against the CPU residue oracle (oracle/residue_controller.py) - the check of
    tests/test_forward_residue_gpu.py made on a sample that travelled in a batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LD = np.longdouble
TRACED = ("scores", "exp", "self_attention", "affine1_0", "encoder_out", "pooled")


def _engine(fa, preset):
    eng = fa.Engine(preset, seed=11, n_q=28, n_p=-1)
    eng.keygen()
    eng.gen_relin_key()
    eng.gen_rotation_keys(fa.circuit_rotation_indices())
    eng.bootstrap_setup(3, 3, 16384)
    return eng


def _same(a, b, what):
    ia, ib = a.info(), b.info()
    assert (ia["npoly"], ia["ell"], ia["deg"]) == (ib["npoly"], ib["ell"], ib["deg"]), (what, ia, ib)
    assert a.scale_parts() == b.scale_parts(), (what, "scale")
    assert np.array_equal(a.export(), b.export()), what


@pytest.mark.parametrize("variant,S", [("main", 129), ("main_2", 130)])
def test_batched_pass_gives_the_residues_of_the_single_passes(fa, variant, S):
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    B = 4
    w = pf.synthetic_model(1234)
    samples = [pf.client_inputs(w, pf.synthetic_tokens(S, 4321 + 13 * x)) for x in range(B)]
    eng = _engine(fa, "reference")
    try:
        single = lf.GpuController(eng)
        encs = [lf.encrypt_inputs(single, *smp) for smp in samples]

        class Recording(lf.GpuController):          # keeps the server-side fresh encryptions of a single pass, in call order
            def __init__(self, e):
                super().__init__(e)
                self.fresh = []

            def encrypt(self, v, level=0):
                c = super().encrypt(v, level)
                self.fresh.append(c)
                return c

        outs, traces, fresh = [], [], []
        eng.stats(reset=True)
        for x in range(B):
            ctl, tr = Recording(eng), {}
            outs.append(lf.forward_encrypted(ctl, w, encs[x], tr, variant))
            eng.sync()
            traces.append(tr)
            fresh.append(ctl.fresh)
        st_single = eng.stats(reset=True)

        class Replaying(lf.BatchedController):      # the batched pass meets the same fresh encryptions, sample by sample
            def __init__(self, e, B):
                super().__init__(e, B)
                self.k = 0

            def encrypt(self, v, level=0):
                b = lf.Batch(fresh[x][self.k] for x in range(self.B))
                self.k += 1
                assert b[0].level == level
                return b

        bctl, btr = Replaying(eng, B), {}
        bout = lf.forward_encrypted(bctl, w, lf.batch_inputs(encs), btr, variant)
        eng.sync()
        st_batch = eng.stats(reset=True)
        assert bctl.k == len(fresh[0]) and bctl.n_boot == 8
        for x in range(B):
            for k in TRACED:
                _same(btr[k][x], traces[x][k], (variant, x, k))
            _same(bout[x], outs[x], (variant, x, "out"))
        # the same work (key switches, transforms, bootstraps), just in wider launches
        for k in ("keyswitch", "keyswitch_limbs", "bootstrap"):
            assert st_batch[k] == st_single[k], (k, st_batch[k], st_single[k])
        # (level adjustments of B additions made in one call share their rescales' bookkeeping: the counter may differ by a few)
        assert abs(st_batch["rescale"] - st_single["rescale"]) <= 0.01 * st_single["rescale"]
        assert st_batch["limb_ntt"] <= st_single["limb_ntt"]      # (the single passes also encoded + encrypted their zero and mask)
        # the same batch as two sub-batches on two lanes (streams) of the one context: scheduling only, the same residues
        class ReplayingLanes(lf.LanedBatchedController):
            def __init__(self, e, B):
                super().__init__(e, B, 2)
                self.k = 0

            def encrypt(self, v, level=0):
                b = lf.Batch(fresh[x][self.k] for x in range(self.B))
                self.k += 1
                return b

        lctl, ltr = ReplayingLanes(eng, B), {}
        lctl.begin()
        lout = lf.forward_encrypted(lctl, w, lf.batch_inputs(encs), ltr, variant)
        lctl.end()
        eng.sync()
        st_lane = eng.stats(reset=True)
        for x in range(B):
            for k in TRACED:
                _same(ltr[k][x], traces[x][k], (variant, "lanes", x, k))
            _same(lout[x], outs[x], (variant, "lanes", x, "out"))
        assert st_lane["keyswitch"] == st_single["keyswitch"] and st_lane["bootstrap"] == st_single["bootstrap"]
        # ONE sample with its independent branches on different lanes (DataflowController: the value projection beside the scores chain):
        # scheduling only, the residues of the plain pass
        class ReplayingFlow(lf.GpuController):
            def __init__(self, e, x):
                super().__init__(e)
                self.k, self.x = 0, x

            def encrypt(self, v, level=0):
                c = fresh[self.x][self.k]
                self.k += 1
                return c

        for x in (0, B - 1):
            fctl, ftr = lf.DataflowController(ReplayingFlow(eng, x)), {}
            fctl.begin()
            fout = lf.forward_encrypted(fctl, w, encs[x], ftr, variant)
            fctl.end()
            eng.sync()
            assert fctl.branches >= 2                       # at least the K and the V projection started branches of their own
            for k in TRACED:
                _same(ftr[k], traces[x][k], (variant, "dataflow", x, k))
            _same(fout, outs[x], (variant, "dataflow", x, "out"))
        eng.stats(reset=True)
        # ... and it is the forward pass of every sample
        for x in range(B):
            lg = lf.logits_from_slots(eng.decrypt(bout[x]))
            ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *samples[x], None, variant))
            assert np.max(np.abs(lg - ref)) < 2e-2 and int(np.argmax(lg)) == int(np.argmax(ref)), (x, np.max(np.abs(lg - ref)))
    finally:
        eng.close()


def test_a_sample_of_a_batch_bit_exact_vs_residue_oracle(fa, orc):
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf
    from oracle.residue_eval import ResidueEvaluator, RCt
    from oracle.residue_boot import ResidueBootstrapper
    from oracle.residue_controller import ResidueController, GaloisKeys
    B, S, CHECK, variant = 2, 129, 1, "main"
    w = pf.synthetic_model(1234)
    samples = [pf.client_inputs(w, pf.synthetic_tokens(S, 4321 + 13 * x)) for x in range(B)]
    eng = _engine(fa, "reference")
    try:
        def rct(ct):
            hi, lo = ct.scale_parts()
            return RCt(ct.export(), ct.info()["deg"], LD(hi) + LD(lo))

        single = lf.GpuController(eng)
        encs = [lf.encrypt_inputs(single, *smp) for smp in samples]
        fresh = [rct(c) for k in ("inputs_E", "inputs_F", "inputs") for c in encs[CHECK][k]]      # the checked sample's client encryptions

        class Recording(lf.BatchedController):      # ... and its server-side ones, in call order
            def encrypt(self, v, level=0):
                b = super().encrypt(v, level)
                fresh.append(rct(b[CHECK]))
                return b

        tr = {}
        out = lf.forward_encrypted(Recording(eng, B), w, lf.batch_inputs(encs), tr, variant)
        got = {k: rct(v[CHECK]) for k, v in tr.items()}
        got["out"] = rct(out[CHECK])

        desc = eng.bootstrap_describe()
        keys = GaloisKeys(eng.log_n)
        keys["relin"], keys["conj"] = eng.key_export(0), eng.key_export(2)
        idx = set(fa.circuit_rotation_indices())
        for st in desc["c2s"] + desc["s2c"]:
            for (g, b, _) in st["terms"]:
                idx.update((g, b))
        j = 1
        while j < (eng.N // 2) // desc["slots"]:
            idx.add(desc["slots"] * j)
            j <<= 1
        for r in sorted(idx):
            if r % (eng.N // 2) and r not in keys:
                keys[r] = eng.key_export(1, r)
        rev = ResidueEvaluator(eng.q, eng.p, eng.psi_q, eng.psi_p, eng.alpha, eng.log_n, keys, eng.params.log_slots)
        boot = ResidueBootstrapper(rev, desc, lambda pt: (lambda ell, sc: eng.pt_export(pt, ell, sc)))
        ctl = ResidueController(eng, rev, boot, fresh, [])
        orc.use_fast(True)
        try:
            tw = {}
            want = lf.forward_encrypted(ctl, w, lf.encrypt_inputs(ctl, *samples[CHECK]), tw, variant)
        finally:
            orc.use_fast(False)
        assert ctl.n_boot == 8 and not ctl.fresh
        tw["out"] = want
        for k in TRACED + ("out",):
            g, r = got[k], tw[k]
            assert (g.npoly, g.ell, g.deg) == (r.npoly, r.ell, r.deg), (k, g.ell, g.deg, r.ell, r.deg)
            assert g.scale == r.scale, (k, "scale")
            assert np.array_equal(g.d, r.d), k
    finally:
        eng.close()
