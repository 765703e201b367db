"""Probe: host-side time per controller / engine call of one forward pass (cProfile; GPU box)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf

S = 129
w = pf.synthetic_model(1234)
e = fa.Engine("bench", seed=11, n_q=28, n_p=-1)
e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices()); e.bootstrap_setup(3, 3, 16384)
ctl = lf.GpuController(e)
ins = pf.client_inputs(w, pf.synthetic_tokens(S, 1))
e.level_plan_begin("record"); enc = lf.encrypt_inputs(ctl, *ins); e.decrypt(lf.forward_encrypted(ctl, w, enc)); e.level_plan_end()
for rep in range(2):
    e.level_plan_begin("apply"); enc = lf.encrypt_inputs(ctl, *ins); e.sync()
    pr = cProfile.Profile()
    pr.enable()
    out = lf.forward_encrypted(ctl, w, enc); v = e.decrypt(out)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
