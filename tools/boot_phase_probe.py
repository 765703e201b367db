"""Probe for rocprofv3: N runs of bootstrap_partial(ct, STAGE) at N=2^16, 28+7 limbs (kernel stats of one bootstrap phase =
difference of two such profiles).  usage: boot_phase_probe.py <stage> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

stage = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
e = fa.Engine("bench", seed=5, n_q=28, n_p=7)
e.keygen(); e.gen_relin_key(); e.bootstrap_setup(3, 3, 1 << 14)
m = np.random.default_rng(1).uniform(-1, 1, 1 << 14)
ct = e.encrypt(m, level=e.n_q - 3)
for _ in range(reps):
    e.bootstrap_partial(ct, stage) if stage else e.bootstrap(ct)
e.sync()
