"""Fixed-shape probe of the merged key switch at the forward pass's typical shape (rocprofv3 kernel stats): N=2^16, 28+7 limbs,
16 rows at ell limbs, merged rotate-sum of 7 rotations {512..3584} (a tree triple) and of 3 {512,1024,1536} (a pair)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

ell = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 16
e = fa.Engine("bench", seed=5, n_q=28, n_p=7)
e.keygen()
idx7 = [512 * k for k in range(1, 8)]
e.gen_rotation_keys(idx7)
ns = 1 << e.params.log_slots
rng = np.random.default_rng(1)
xs = e.encrypt_batch(rng.uniform(-1, 1, (B, ns)), level=e.n_q - ell)
for _ in range(2):
    e.rotate_sum(xs, idx7); e.rotate_sum(xs, idx7[:3])
e.sync()
e.timer_start()
for _ in range(reps):
    e.rotate_sum(xs, idx7)
t7 = e.timer_stop() / reps
e.timer_start()
for _ in range(reps):
    e.rotate_sum(xs, idx7[:3])
t3 = e.timer_stop() / reps
print(json.dumps({"N": e.N, "ell": ell, "k": e.n_p, "alpha": e.alpha, "beta": -(-ell // e.alpha), "batch": B, "reps": reps,
                  "ms_rotate_sum7_batch": round(t7, 4), "ms_rotate_sum3_batch": round(t3, 4)}))
