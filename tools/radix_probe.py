"""Probe: a 7-step rotate-and-sum tree as 3 merged pairs + 1 single step (today) vs 2 merged triples {s..7s} + 1 single."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

e = fa.Engine("bench", seed=5, n_q=28, n_p=-1)
e.keygen()
u = 128
idx = [u * k for k in range(1, 8)] + [8 * u * k for k in range(1, 8)] + [64 * u, 2 * u * 2, ]
e.gen_rotation_keys(sorted(set(idx + [4 * u, 8 * u, 12 * u, 16 * u, 32 * u, 48 * u])))
ns = 1 << e.params.log_slots
rng = np.random.default_rng(1)
for ell in (27, 12, 6):
    xs = e.encrypt_batch(rng.uniform(-1, 1, (8, ns)), level=e.n_q - ell)

    def radix4(v):
        v = e.rotate_sum(v, [u, 2 * u, 3 * u]); v = e.rotate_sum(v, [4 * u, 8 * u, 12 * u]); v = e.rotate_sum(v, [16 * u, 32 * u, 48 * u])
        return e.rotate_sum(v, [64 * u])

    def radix8(v):
        v = e.rotate_sum(v, [u * k for k in range(1, 8)]); v = e.rotate_sum(v, [8 * u * k for k in range(1, 8)])
        return e.rotate_sum(v, [64 * u])

    res = {}
    for name, fn in (("radix4", radix4), ("radix8", radix8), ("radix4b", radix4), ("radix8b", radix8)):
        for _ in range(2):
            fn(xs)
        e.sync(); e.timer_start()
        for _ in range(10):
            fn(xs)
        res[name] = round(e.timer_stop() / 10, 3)
    a, b = e.decrypt(radix4(xs)[0]), e.decrypt(radix8(xs)[0])
    res["max_diff"] = float(np.max(np.abs(a - b)))
    print(json.dumps({"ell": ell, "ms_per_tree_of_8_rows": res}))
