"""Probe: bootstrapping at a given preset — precision, levels, wall time (not a test; used to fill DESIGN.md)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_linformer_amd as fa

preset = sys.argv[1] if len(sys.argv) > 1 else "bench"
kw = {}
if len(sys.argv) > 2:
    kw["n_q"] = int(sys.argv[2])
if len(sys.argv) > 3:
    kw["n_p"] = int(sys.argv[3])
e = fa.Engine(preset, seed=int(os.environ.get("BOOT_SEED", "5")), **kw)
t0 = time.time(); e.keygen(); e.gen_relin_key(); print("keygen s", round(time.time() - t0, 2))
t0 = time.time(); _bud = [int(v) for v in os.environ.get("BOOT_BUDGET", "3,3").split(",")]
t0 = time.time(); e.bootstrap_setup(_bud[0], _bud[1], 1 << e.params.log_slots); print("setup s", round(time.time() - t0, 2))
n = 1 << e.params.log_slots
m = np.random.default_rng(1).uniform(-1, 1, n)
ct = e.encrypt(m, level=e.n_q - 3)
for it in range(int(os.environ.get("BOOT_REPS", "3"))):
    e.sync(); t0 = time.time()
    out = e.bootstrap_drop(ct, 0)      # evaluated at the call (fhelin_bootstrap itself defers until the result is read)
    e.sync(); dt = time.time() - t0
    err = np.max(np.abs(e.decrypt(out) - m))
    print(f"bootstrap {it}: {dt*1e3:.1f} ms, out {out.info()}, max err {err:.3e}")
# time per phase (each partial run repeats the earlier phases): ModRaise+SubSum | + CoeffsToSlots | + EvalMod (real half only)
if os.environ.get("BOOT_PHASES"):
    for stage in (1, 2, 3):
        ts = []
        for it in range(4):
            e.sync(); t0 = time.time()
            e.bootstrap_partial(ct, stage)
            e.sync(); ts.append((time.time() - t0) * 1e3)
        print(f"partial stage {stage}: {min(ts):.2f} ms")

# batched bootstrapping: B independent ciphertexts through one pipeline (BOOT_BATCH=5, BOOT_DROP=4: the GELU containers under the level plan)
if os.environ.get("BOOT_BATCH"):
    B, drop = int(os.environ["BOOT_BATCH"]), int(os.environ.get("BOOT_DROP", "0"))
    cts = [e.encrypt(np.random.default_rng(10 + i).uniform(-1, 1, n), level=e.n_q - 2) for i in range(B)]
    if drop:
        e.set_level_plan([e.n_q - 15 - drop] * B)
    for it in range(4):
        if drop:
            e.level_plan_begin("apply")
        e.sync(); t0 = time.time()
        outs = e.bootstrap_batch(cts)
        e.sync(); dt = time.time() - t0
        print(f"batch of {B} (drop {drop}) {it}: {dt*1e3:.1f} ms = {dt*1e3/B:.2f} ms per bootstrap, out ell {outs[0].info()['ell']}")
    e.level_plan_begin("off")
    ts = []
    for it in range(3):
        e.sync(); t0 = time.time()
        for c in cts:
            e.bootstrap_drop(c, drop)
        e.sync(); ts.append((time.time() - t0) * 1e3)
    print(f"one by one (drop {drop}): {min(ts):.1f} ms = {min(ts)/B:.2f} ms per bootstrap")
