"""Probe: GPU time per controller method of one forward pass (each call synchronised; deferred heavy operations - bootstraps and
Chebyshev evaluations, which run batched when their results are first read - are flushed and timed as their own bucket before
the next other call).  Usage: python tools/method_breakdown.py [log_n] [tokens] [plan 0/1]"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 129
use_plan = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
w = pf.synthetic_model(1234)
e = fa.Engine("bench", seed=2024, log_n=log_n, n_q=28, n_p=-1)
e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices()); e.bootstrap_setup(3, 3, 16384)
HEAVY = {"bootstrap", "eval_gelu_function", "eval_tanh_function", "eval_inverse_naive"}
acc = collections.OrderedDict()


def note(name, dt, d):
    a = acc.setdefault(name, [0, 0.0, 0, 0, 0])
    a[0] += 1; a[1] += dt; a[2] += d["keyswitch"]; a[3] += d["limb_ntt"]; a[4] += d["keyswitch_limbs"]


class Timed(lf.GpuController):
    pending = []

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if name.startswith("_") or not callable(attr) or name in ("level", "encode", "encrypt", "decrypt", "clone"):
            return attr
        eng = object.__getattribute__(self, "e")

        def call(*a, **k):
            if name not in HEAVY and Timed.pending:
                s0 = eng.stats(); t0 = time.perf_counter(); eng.sync(); dt = time.perf_counter() - t0
                s1 = eng.stats()
                note("deferred batch: " + "+".join(f"{n}x{Timed.pending.count(n)}" for n in dict.fromkeys(Timed.pending)), dt,
                     {k2: s1[k2] - s0[k2] for k2 in s0})
                Timed.pending = []
            s0 = eng.stats(); t0 = time.perf_counter()
            r = attr(*a, **k)
            if name in HEAVY:
                Timed.pending.append(name)
                return r
            eng.sync(); dt = time.perf_counter() - t0
            s1 = eng.stats()
            note(name, dt, {k2: s1[k2] - s0[k2] for k2 in s0})
            return r
        return call


ctl0 = lf.GpuController(e)
ins = pf.client_inputs(w, pf.synthetic_tokens(S, 1))
if use_plan:
    e.level_plan_begin("record"); enc = lf.encrypt_inputs(ctl0, *ins); n_src = sum(len(v) for v in enc.values())
    e.decrypt(lf.forward_encrypted(ctl0, w, enc)); e.level_plan_end()
for rep in range(2):
    if use_plan:
        e.level_plan_begin("apply")
    enc = lf.encrypt_inputs(ctl0, *ins); e.sync()
    if use_plan:
        e.level_plan_begin("apply", first_source=n_src)
    acc.clear(); Timed.pending = []
    ctl = Timed(e)
    t0 = time.perf_counter()
    out = lf.forward_encrypted(ctl, w, enc); e.sync()
    total = time.perf_counter() - t0
print(f"forward (synchronised per call) {total*1e3:.0f} ms, N=2^{log_n}, S={S}, level plan {'on' if use_plan else 'off'}")
print(f"{'method':58s} {'calls':>5s} {'keyswitch':>10s} {'limbs/ks':>8s} {'limb-NTT':>10s} {'ms':>8s}")
for k, (n, dt, ks, ntt, ksl) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:58s} {n:5d} {ks:10d} {ksl / max(ks, 1):8.1f} {ntt:10d} {dt*1e3:8.1f}")
