"""Probe: how fast ONE host thread issues launches through the C ABI (is the host the limit when two lanes are fed from one thread?).
Tiny transforms (1 limb vector = 16 workgroups per tile pass, ~8 us on the GPU) back to back; time until the loop returns vs until the GPU is done."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

e = fa.Engine("bench", seed=3)
buf = e.upload(np.zeros((4, e.N), dtype=np.uint64))
for nvec, reps in ((1, 3000), (4, 3000)):
    e.ntt(buf, nvec, 0, 1); e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.ntt(buf, nvec, 0, 1)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print(f"nvec {nvec}: {reps} transforms = {2 * reps} launches: loop returned after {1e3 * (t1 - t0):.1f} ms "
          f"({1e6 * (t1 - t0) / (2 * reps):.1f} us per launch on the host), GPU done after {1e3 * (t2 - t0):.1f} ms "
          f"({1e6 * (t2 - t0) / (2 * reps):.1f} us per launch)")
e.close()
