"""Probe: level plan on the full forward pass (GPU box): record one pass, apply to the next ones, compare logits and time."""
import sys, time, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf, circuit_sim as cs

S = int(sys.argv[1]) if len(sys.argv) > 1 else 129
variant = sys.argv[2] if len(sys.argv) > 2 else "main"
w = pf.synthetic_model(1234)
e = fa.Engine("bench", seed=11, n_q=28, n_p=-1)
e.keygen(); e.gen_relin_key()
e.gen_rotation_keys(fa.circuit_rotation_indices())
e.bootstrap_setup(3, 3, 16384)
ctl = lf.GpuController(e)


def one(mode, seed):
    x = pf.synthetic_tokens(S, seed)
    ins = pf.client_inputs(w, x)
    if mode:
        e.level_plan_begin(mode)
    enc = lf.encrypt_inputs(ctl, *ins)
    e.sync(); t0 = time.time(); s0 = e.stats()
    out = lf.forward_encrypted(ctl, w, enc, variant=variant)
    lg = lf.logits_from_slots(e.decrypt(out))
    e.sync(); dt = (time.time() - t0) * 1e3; s1 = e.stats()
    plan = e.level_plan_end() if mode else None
    ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *ins, variant=variant))
    return dt, float(np.max(np.abs(lg - ref))), s1["limb_ntt"] - s0["limb_ntt"], plan, out.info()


for rep in range(2):
    dt, err, ntt, _, info = one(None, 100 + rep)
    print(f"plain   pass: {dt:7.1f} ms  err {err:.2e}  limb-NTT {ntt}  out ell {info['ell']}")
dt, err, ntt, plan, info = one("record", 200)
print(f"record  pass: {dt:7.1f} ms  err {err:.2e}  limb-NTT {ntt}")
print("plan:", collections.Counter(plan[:194]).most_common(6), "server sources:", plan[194:])
for rep in range(3):
    dt, err, ntt, _, info = one("apply", 300 + rep)
    print(f"applied pass: {dt:7.1f} ms  err {err:.2e}  limb-NTT {ntt}  out ell {info['ell']}")
