"""Probe: one encrypted Linformer forward pass on the GPU vs the plaintext circuit simulation (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf, circuit_sim as cs

preset = sys.argv[1] if len(sys.argv) > 1 else "reference"
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 28
S = int(sys.argv[3]) if len(sys.argv) > 3 else 129
variant = sys.argv[4] if len(sys.argv) > 4 else "main"
w = pf.synthetic_model(1234); x = pf.synthetic_tokens(S, 4321)
x_in, X_E, X_F = pf.client_inputs(w, x)
sim = cs.SlotSimController(); st = {}
ref = lf.forward(sim, w, x_in, X_E, X_F, st, variant)
n_p = -1   # library rule (OpenFHE sizeP): 7 special limbs for 29-30 Q limbs
e = fa.Engine(preset, seed=11, n_q=n_q, n_p=n_p)
t0 = time.time(); e.keygen(); e.gen_relin_key()
e.gen_rotation_keys(fa.circuit_rotation_indices())
e.bootstrap_setup(3, 3, 16384); print("keys+setup s", round(time.time() - t0, 2), flush=True)


class Tracing(lf.GpuController):
    def bootstrap(self, a):
        print("  bootstrap at", a.info(), flush=True)
        return super().bootstrap(a)


ctl = Tracing(e); tr = {}
e.sync(); t0 = time.time()
try:
    out = lf.forward(ctl, w, x_in, X_E, X_F, tr, variant)
    e.sync(); print("forward s", round(time.time() - t0, 2), "bootstraps", ctl.n_boot)
except Exception as ex:
    print("FAILED:", ex)
    out = None
for k, c in tr.items():
    got = e.decrypt(c)
    print(f"{k:16s} info={c.info()} max|sim|={np.max(np.abs(st[k])):.3g} max err={np.max(np.abs(got - st[k])):.3e}")
if out is not None:
    lg, lr = lf.logits_from_slots(e.decrypt(out)), lf.logits_from_slots(ref)
    print("out info", out.info()); print("gpu logits", np.round(lg[:8], 4)); print("sim logits", np.round(lr[:8], 4))
    print("max logit err", np.max(np.abs(lg - lr)), "pred gpu/sim", int(np.argmax(lg)), int(np.argmax(lr)))
