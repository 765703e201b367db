"""Probe: per controller method, how many key switches / limb-NTTs / wall ms one forward pass spends (GPU box only).
Each method call is bracketed by a stream sync, so the wall times add up to more than the pipelined run."""
import sys, time, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf

S = int(sys.argv[1]) if len(sys.argv) > 1 else 129
w = pf.synthetic_model(1234); x = pf.synthetic_tokens(S, 4321)
x_in, X_E, X_F = pf.client_inputs(w, x)
e = fa.Engine("bench", seed=11, n_q=28, n_p=-1)
e.keygen(); e.gen_relin_key()
e.gen_rotation_keys(fa.circuit_rotation_indices())
e.bootstrap_setup(3, 3, 16384)
agg = collections.defaultdict(lambda: [0, 0, 0, 0.0])
depth = [0]


def wrap(name, fn):
    def inner(*a, **k):
        if depth[0] > 0:
            return fn(*a, **k)
        depth[0] += 1
        e.sync(); s0 = e.stats(); t0 = time.time()
        try:
            return fn(*a, **k)
        finally:
            e.sync(); s1 = e.stats(); depth[0] -= 1
            r = agg[name]
            r[0] += 1; r[1] += s1["keyswitch"] - s0["keyswitch"]; r[2] += s1["limb_ntt"] - s0["limb_ntt"]; r[3] += (time.time() - t0) * 1e3
    return inner


ctl = lf.GpuController(e)
for name in dir(ctl):
    if name.startswith("_") or name in ("level", "clone", "e", "n_boot", "verbose"):
        continue
    fn = getattr(ctl, name)
    if callable(fn):
        setattr(ctl, name, wrap(name, fn))
plan = os.environ.get("LEVEL_PLAN", "1") != "0"
for it in range(3):
    agg.clear()
    if plan:
        e.level_plan_begin("record" if it == 0 else "apply")
    enc = lf.encrypt_inputs(ctl, x_in, X_E, X_F)
    e.sync(); t0 = time.time()
    out = lf.forward_encrypted(ctl, w, enc)
    e.decrypt(out)
    e.sync(); total = (time.time() - t0) * 1e3
    if plan:
        e.level_plan_end()
print(f"forward (synchronised per call) {total:.0f} ms")
print(f"{'method':28s} {'calls':>6s} {'keyswitch':>10s} {'limb-NTT':>10s} {'ms':>8s} {'us/KS':>8s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    print(f"{k:28s} {v[0]:6d} {v[1]:10d} {v[2]:10d} {v[3]:8.1f} {v[3]*1e3/max(v[1],1):8.1f}")
