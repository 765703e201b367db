"""Probe: wall time of fhelin_decrypt (phase on the GPU, download, CRT + special FFT on the host) at N=2^16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa
e = fa.Engine("bench", seed=5, n_q=28, n_p=7)
e.keygen()
ct = e.encrypt(np.random.default_rng(1).uniform(-1, 1, 16384), level=e.n_q - 3)
for _ in range(3):
    e.sync(); t0 = time.time(); v = e.decrypt(ct); dt = (time.time() - t0) * 1e3
    print(f"decrypt at 3 limbs: {dt:.2f} ms")
x = np.random.default_rng(2).uniform(-1, 1, 16384)
for _ in range(3):
    e.sync(); t0 = time.time(); p = e.encode(x); e.sync(); dt = (time.time() - t0) * 1e3
    print(f"encode (handle only): {dt:.2f} ms")
