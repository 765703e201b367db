"""Probe for rocprofv3: the re-associated FFN stretch at the forward pass's shape (N=2^16, 28+7 limbs): unwrapExpanded of 128 rows read
together (two hoisted fans + window sums), matmulRElarge's first step (double hoisting, merged rescale) and the fused containers
(cyclic sums + one shift sum per 32 rows).  Usage: python tools/ffn_probe.py [reps] [ell]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ell = int(sys.argv[2]) if len(sys.argv) > 2 else 13
e = fa.Engine("bench", seed=5, n_q=28, n_p=7)
e.keygen(); e.gen_relin_key(); e.gen_rotation_keys(fa.circuit_rotation_indices())
rng = np.random.default_rng(1)
wrapped = e.encrypt(rng.uniform(-1, 1, 16384), level=e.n_q - ell)
ws = [e.encode(rng.uniform(-1, 1, 16384) / 8) for _ in range(4)]
bias = e.encode(rng.uniform(-1, 1, 16384))
for it in range(reps):
    e.sync(); s0 = e.stats(); t0 = time.time()
    rows = e.unwrapExpanded(wrapped, 128)
    e.force(rows)
    e.sync(); t1 = time.time(); s1 = e.stats()
    lazy = e.matmulRElarge(rows, ws, bias, 1.0)
    cont = e.generate_containers(lazy)
    e.sync(); t2 = time.time(); s2 = e.stats()
    print(f"unwrapExpanded 128 rows (bulk): {(t1-t0)*1e3:.1f} ms, key switches {s1['keyswitch']-s0['keyswitch']}, limb-NTT {s1['limb_ntt']-s0['limb_ntt']}; "
          f"generate_containers(matmulRElarge) fused: {(t2-t1)*1e3:.1f} ms, key switches {s2['keyswitch']-s1['keyswitch']}, "
          f"limb-NTT {s2['limb_ntt']-s1['limb_ntt']}, {len(cont)} containers at ell {cont[0].info()['ell']}")
