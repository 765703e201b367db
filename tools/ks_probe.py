"""Fixed-shape probe of the hybrid key-switch kernels for rocprofv3 (kernel stats / PMC passes): N=2^16, 24+6 limbs
(alpha 6, beta 4), batch of 8 rows at ell=24: `reps` plain rotations (modup_conv, ks_inner, moddown_conv, moddown_finish
+ the NTT passes) and `reps` merged rotate-sums {128,256,384} (ks_inner_multi, gather_sum).  Prints the shape so that the
algorithmic bytes per launch can be computed (tools/pmc_summary.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

ell = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 8
e = fa.Engine("bench", seed=5)
e.keygen()
e.gen_rotation_keys([128, 256, 384])
ns = 1 << e.params.log_slots
rng = np.random.default_rng(1)
xs = e.encrypt_batch(rng.uniform(-1, 1, (B, ns)), level=e.n_q - ell)
for _ in range(2):
    e.rotate_batch(xs, 128); e.rotate_sum(xs, [128, 256, 384])
e.sync()
e.timer_start()
for _ in range(reps):
    e.rotate_batch(xs, 128)
t_rot = e.timer_stop() / reps
e.timer_start()
for _ in range(reps):
    e.rotate_sum(xs, [128, 256, 384])
t_sum = e.timer_stop() / reps
print(json.dumps({"N": e.N, "ell": ell, "k": e.n_p, "alpha": e.alpha, "beta": -(-ell // e.alpha), "batch": B, "reps": reps,
                  "ms_rotate_batch": round(t_rot, 4), "ms_rotate_sum_batch": round(t_sum, 4)}))
