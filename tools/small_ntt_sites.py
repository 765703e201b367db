"""Which callers still issue transforms that cannot fill the GPU: one forward pass with FHELIN_NTT_TRACE (csrc/context.cpp
note_small_ntt) - the table is printed by the library when the context is closed.  Usage: FHELIN_NTT_TRACE=8 python tools/small_ntt_sites.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FHELIN_NTT_TRACE", "8")
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = pf.synthetic_model(1234)
e = fa.Engine("bench", seed=2024, n_q=28, n_p=-1)
e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices()); e.bootstrap_setup(3, 3, 16384)
single = lf.GpuController(e)
e.level_plan_begin("record")
enc = lf.encrypt_inputs(single, *pf.client_inputs(w, pf.synthetic_tokens(129, 999)))
e.decrypt(lf.forward_encrypted(single, w, enc))
plan = e.level_plan_end()
encs = []
for x in range(B):
    e.level_plan_begin("apply")
    encs.append(lf.ingest_sample(single, w, pf.synthetic_tokens(129, 5 + x)))
e.set_level_plan(lf.batched_level_plan(plan, B, 194))
ctl = lf.BatchedController(e, B)
for rep in range(2):
    if rep == 1:
        e.lib.fhelin_sync(e.h)
    e.level_plan_begin("apply", first_source=194 * B)
    out = lf.forward_encrypted(ctl, w, lf.batch_inputs(encs))
    [e.decrypt(o) for o in out]
print("passes done; the table below covers set-up + 2 passes of", B, "sample(s)")
e.close()
