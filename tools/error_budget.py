#!/usr/bin/env python3
"""Per-knob error budget of the forward pass: max |logit - circuit oracle| over the bench's samples for
packed EvalMod on/off x level plan on/off, and the default configuration with each re-association knob off in turn, at the headline ring.  Writes one JSON line per configuration.
Usage: python tools/error_budget.py [n_samples] > profiles/rNN_error_budget.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


KNOBS = ("FHELIN_MERGED_RESCALE", "FHELIN_DOUBLE_HOIST", "FHELIN_CHEB_LEAF_CLASSES", "FHELIN_FUSE_RELARGE", "FHELIN_BULK_UNWRAP")


def run(packed, plan_on, n_samples, S=129, off=None):
    os.environ["FHELIN_BOOT_PACKED"] = "1" if packed else "0"
    for k in KNOBS:                      # every re-association on, except the one under test
        os.environ.pop(k, None)
    if off:
        os.environ[off] = "0"
    import fhe_linformer_amd as fa
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    eng = fa.Engine("bench", seed=2024, n_q=28, n_p=-1)
    try:
        eng.keygen()
        eng.gen_relin_key()
        eng.gen_rotation_keys(fa.circuit_rotation_indices())
        eng.bootstrap_setup(3, 3, 16384)
        w = pf.synthetic_model(1234)
        ctl = lf.GpuController(eng)
        n_src = 0
        if plan_on:
            eng.level_plan_begin("record")
            enc = lf.encrypt_inputs(ctl, *pf.client_inputs(w, pf.synthetic_tokens(S, 999)))
            n_src = sum(len(v) for v in enc.values())
            eng.decrypt(lf.forward_encrypted(ctl, w, enc))
            eng.level_plan_end()
            del enc
        errs, boot_errs = [], []
        for i in range(n_samples):
            x = pf.synthetic_tokens(S, 4321 + i)
            ins = pf.client_inputs(w, x)
            if plan_on:
                eng.level_plan_begin("apply")
            enc = lf.encrypt_inputs(ctl, *ins)
            if plan_on:
                eng.level_plan_begin("apply", first_source=n_src)
            lg = lf.logits_from_slots(eng.decrypt(lf.forward_encrypted(ctl, w, enc)))
            ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *ins))
            errs.append(float(np.max(np.abs(lg - ref))))
            del enc
        eng.level_plan_begin("off")
        # one bootstrap alone on a uniform message, for scale
        for s in range(4):
            m = np.random.default_rng(s).uniform(-1, 1, 16384)
            ct = eng.encrypt(m, level=eng.n_q - 3)
            boot_errs.append(float(np.max(np.abs(eng.decrypt(eng.bootstrap(ct)) - m))))
        return {"packed_evalmod": packed, "level_plan": plan_on, "knob_off": off, "samples": n_samples, "max_logit_err": max(errs),
                "mean_logit_err": float(np.mean(errs)), "per_sample": [round(e, 6) for e in errs],
                "single_bootstrap_max_err_4_seeds": [float("%.3g" % e) for e in boot_errs]}
    finally:
        eng.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for packed in (True, False):
        for plan_on in (True, False):
            print(json.dumps(run(packed, plan_on, n)), flush=True)
    # which re-association costs precision: the default configuration with ONE knob off at a time
    for off in KNOBS:
        print(json.dumps(run(True, True, n, off=off)), flush=True)
