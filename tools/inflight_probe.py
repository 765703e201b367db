"""Probe: forward-pass throughput with K samples in flight on ONE GPU (K engines = K contexts with their own HIP streams and the
same replicated keys, one host thread each): the single-ciphertext chains of one sample (Chebyshev evaluations, the pooler's
bootstrap) overlap with the batched row loops of another.  Usage: python tools/inflight_probe.py [K] [passes_per_engine]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = 129
w = pf.synthetic_model(1234)
engines = []
for k in range(K):
    e = fa.Engine("bench", seed=2024, n_q=28, n_p=-1)
    e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices()); e.bootstrap_setup(3, 3, 16384)
    ctl = lf.GpuController(e)
    e.level_plan_begin("record")
    enc = lf.encrypt_inputs(ctl, *pf.client_inputs(w, pf.synthetic_tokens(S, 999)))
    n_src = sum(len(v) for v in enc.values())
    e.decrypt(lf.forward_encrypted(ctl, w, enc)); e.level_plan_end()
    samples = []
    for i in range(P + 1):
        e.level_plan_begin("apply")
        samples.append(lf.encrypt_inputs(ctl, *pf.client_inputs(w, pf.synthetic_tokens(S, 4321 + 100 * k + i))))
    e.sync()
    engines.append((e, ctl, n_src, samples))


def run(k, lo, hi, out):
    e, ctl, n_src, samples = engines[k]
    for i in range(lo, hi):
        e.level_plan_begin("apply", first_source=n_src)
        out.append(lf.logits_from_slots(e.decrypt(lf.forward_encrypted(ctl, w, samples[i]))))
    e.sync()


for k in range(K):
    run(k, 0, 1, [])           # warm-up pass per engine
for mode in ("one at a time", f"{K} in flight"):
    outs = [[] for _ in range(K)]
    t0 = time.perf_counter()
    if mode == "one at a time":
        for k in range(K):
            run(k, 1, P + 1, outs[k])
    else:
        th = [threading.Thread(target=run, args=(k, 1, P + 1, outs[k])) for k in range(K)]
        for t in th: t.start()
        for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{mode}: {K * P} samples in {dt*1e3:.0f} ms = {dt*1e3/(K*P):.1f} ms/sample")
