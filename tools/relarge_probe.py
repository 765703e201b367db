"""Probe for rocprofv3: matmulRElarge on 128 rows at the level the forward pass runs it at (N=2^16, 28+7 limbs, 12 limbs in)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ell = int(sys.argv[2]) if len(sys.argv) > 2 else 12
e = fa.Engine("bench", seed=5, n_q=28, n_p=7)
e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices())
rng = np.random.default_rng(1)
rows = e.encrypt_batch(rng.uniform(-1, 1, (128, 16384)), level=e.n_q - ell)
ws = [e.encode(rng.uniform(-1, 1, 16384)) for _ in range(4)]
bias = e.encode(rng.uniform(-1, 1, 16384))
for it in range(reps):
    e.sync(); s0 = e.stats(); t0 = time.time()
    out = e.matmulRElarge(rows, ws, bias, 1.0)
    e.sync(); s1 = e.stats()
    print(f"matmulRElarge 128 rows: {(time.time()-t0)*1e3:.1f} ms, key switches {s1['keyswitch']-s0['keyswitch']}, limb-NTT {s1['limb_ntt']-s0['limb_ntt']}, out ell {out[0].info()['ell']}")
