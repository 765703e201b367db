"""Probe: where the bootstrapping error comes from (boot12 preset, n = 1024 slots, sparse): error of the CoeffsToSlots output against
the known plaintext coefficients, and of the final message, packed vs two-ciphertext EvalMod."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

def coeffs(m):
    n = len(m); j = np.arange(n)
    e5 = np.array([pow(5, int(t), 4 * n) for t in j])
    U = np.exp(2j * np.pi * np.outer(e5, np.arange(n)) / (4 * n))
    return U.conj().T @ m / n

for seed in (77, 78, 79):
    e = fa.Engine("boot12", seed=seed, log_slots=10)
    e.keygen(); e.gen_relin_key(); e.bootstrap_setup(3, 3, 1 << 10)
    n = 1 << 10
    m = np.random.default_rng(seed).uniform(-1, 1, n)
    ct = e.encrypt(m, level=e.n_q - 3)
    a = e.decrypt(e.bootstrap_partial(ct, 2))
    w = coeffs(m)
    fr = a * 28 - np.round(a * 28)
    if len(a) == 2 * n:
        eL = np.sort(fr[:n] * 2 ** 10) - np.sort(w.real); eR = np.sort(fr[n:] * 2 ** 10) - np.sort(w.imag)
        print(f"seed {seed} packed: C2S err left {np.max(np.abs(eL)):.2e} right {np.max(np.abs(eR)):.2e}", end="  ")
    else:
        eL = np.sort(fr * 2 ** 10) - np.sort(w.real)
        print(f"seed {seed} two-ct: C2S err real {np.max(np.abs(eL)):.2e}", end="  ")
    s3 = e.decrypt(e.bootstrap_partial(ct, 3))
    out = e.bootstrap(ct)
    err = e.decrypt(out) - m
    print(f"final max {np.max(np.abs(err)):.2e} rms {np.sqrt(np.mean(err**2)):.2e}")
    e.close()
