#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native RNS-CKKS engine (BASELINE.json metric:
"encrypted Linformer-d128 forward ms/sample; NTT/s at N=2^16 (1/2/4/8 GPU)").

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0; for N>1 the
driver launches it under torch.distributed.run, one rank per GPU (RCCL over xGMI).

Workload "forward" (default).  A *step* = one encrypted Linformer-d128 forward pass (the call sequence of
reference src/main.cpp:145-475: Q/K/V matmuls, scores, Taylor^8 exp, Chebyshev 1/x, attention, W_O, affine-1,
bootstrap, FFN 128->512 + Chebyshev GELU + bootstrap, FFN 512->128, affine-2, pooler with Chebyshev tanh,
classifier; 8 bootstraps, ~14.8k key switches) over one sample of S=129 tokens + CLS, on synthetic
weights/tokens (oracle/plain_forward.py, seeds 1234/4321).  Input ciphertexts are encrypted before the timed
region (resident in HBM); the timed region is pure server-side evaluation plus the final decrypt of the logits.
Ring: N=2^16, 16384 slots, dnum 4, 28+7 limbs (55-bit q0, 52-bit scaling, 60-bit special) — the reference's own
chain: depth 27 = 28 Q limbs, and the 7 special limbs OpenFHE's HYBRID rule gives for it.
Sharding: independent samples, one per GPU per step (weak scaling), keys replicated, no data-path collective;
one RCCL all_gather of the logits at the end.

The same run also measures the second headline metric, limb-NTT/s at N=2^16 (fwd+inv NTT of 8 ciphertexts x
2 x 24 limbs), which is where `roofline` (dominant kernel = the NTT tile passes) comes from.
`--workload ntt` runs only that part (value = limb-NTT/s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["forward", "ntt"], default="forward")
    ap.add_argument("--tokens", type=int, default=129, help="tokens per sample (S); S+1 rows incl. CLS, 128 < S+1 <= 256")
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--n-q", type=int, default=28)
    ap.add_argument("--n-p", type=int, default=0, help="special limbs (0: OpenFHE's rule ceil(widest digit bits / 60) = 7 for 28 limbs)")
    ap.add_argument("--ntt-batch", type=int, default=8, help="ciphertexts per NTT step per GPU")
    ap.add_argument("--ntt-steps", type=int, default=30)
    ap.add_argument("--micro", action="store_true", help="instruction-rate probes instead of the benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: ranks may share GPU 0)")
    return ap.parse_args()


def host_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except Exception:
            pass
    return min(n, 16)   # the GPU box grants 16 CPUs per GPU


def micro(fa):
    eng = fa.Engine("toy", device=0)
    names = ["v_mul_lo_u32 x8", "v_mul_hi_u32 x8", "v_mad_u64_u32 x8", "harvey butterfly x4", "v_fma_f64 x64",
             "add_u64 x8", "mulhi64 x8", "mullo64 x8", "lazy butterfly (approx quotient, no csub) x4", "harvey butterfly (variable twiddle) x4"]
    per_iter = [8, 8, 8, 4, 64, 8, 8, 8, 4, 4]
    blocks, iters = 256 * 8, 2048
    out = {}
    for v, (nm, k) in enumerate(zip(names, per_iter)):
        ms = min(eng.microbench(v, iters, blocks) for _ in range(3))
        ops = blocks * 256 * iters * k
        out[nm] = {"ms": round(ms, 4), "Gop_s": round(ops / ms / 1e6, 1),
                   "ops_per_clk_per_CU@2.4GHz": round(ops / (ms * 1e-3) / 2.4e9 / 256, 2)}
    # pure issue rates (asm, 16 independent instructions per trip); "slots" = 4-cycle SIMD issue slots per instruction
    kinds = ["v_mov_b32", "v_add_u32", "v_lshl_add_u64", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32",
             "v_sub_co+s_nop1+v_subb_co (pair)", "v_cndmask_b32", "v_add3_u32", "v_xor_b32", "v_lshrrev_b64",
             "v_mad_u32_u24", "v_mul_hi_u32/v_mad_u64_u32 alternating", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp_gt_u64_e64",
             "v_ashrrev_i32", "v_and_b32", "v_min_u32", "v_sub_co_u32 (sgpr carry out)", "v_add_co_u32 (vcc)", "v_addc_co_u32 (vcc chain)",
             "v_cndmask_b32_e32 (vcc, no clobber)", "v_bfi_b32", "v_max_u32"]
    per = [16, 16, 16, 16, 16, 16, 8] + [16] * 17
    issue = {}
    for k, (nm, n) in enumerate(zip(kinds, per)):
        ms = min(eng.microbench(100 + k, iters, blocks) for _ in range(3))
        ops = blocks * 256 * iters * n
        issue[nm] = {"ms": round(ms, 4), "ops_per_clk_per_CU@2.4GHz": round(ops / (ms * 1e-3) / 2.4e9 / 256, 2)}
    eng.close()
    print(json.dumps({"micro": out, "issue": issue}))


def ntt_section(eng, orc, np, batch, steps, warmup=3):
    """fwd+inv NTT of `batch` ciphertexts x 2 polys x 24 limbs; returns (limb-NTTs done, kernel ms, one-ct host array)"""
    nq = min(24, eng.n_q)
    nvec = batch * 2 * nq
    one = np.stack([orc.uniform_residues(0x5EED0001 + 1000 * p, eng.q[:nq], eng.N) for p in range(2)])  # [2][nq][N]
    host = np.ascontiguousarray(np.broadcast_to(one, (batch,) + one.shape))
    buf = eng.upload(host)
    for _ in range(warmup):
        eng.ntt(buf, nvec, 0, nq)
        eng.ntt(buf, nvec, 0, nq, inverse=True)
    eng.sync()
    eng.timer_start()
    for _ in range(steps):
        eng.ntt(buf, nvec, 0, nq)
        eng.ntt(buf, nvec, 0, nq, inverse=True)
    ms = eng.timer_stop()          # HIP events on the engine's stream
    back = buf.download(host.shape)
    assert np.array_equal(back, host), "INTT(NTT(x)) != x after the timed NTT region"
    buf.free()
    return 2 * nvec * steps, ms, one, nq


def cpu_ntt_baseline(eng, orc, np, one, nq, seconds, cores):
    d = np.ascontiguousarray(one.reshape(-1, eng.N)).copy()
    q, psi = eng.q[:nq], eng.psi_q[:nq]
    orc.ntt_batch(d, q, psi, inplace=True)
    orc.ntt_batch(d, q, psi, inverse=True, inplace=True)  # warm tables
    t0, n = time.perf_counter(), 0
    while True:
        orc.ntt_batch(d, q, psi, inplace=True)
        orc.ntt_batch(d, q, psi, inverse=True, inplace=True)
        n += 2 * d.shape[0]
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return n / dt, n, dt


def cpu_forward_baseline(eng, orc, np, stats, seconds, cores):
    """CPU port (oracle, OpenMP over limbs) of the op that dominates the forward pass — a hybrid key-switched
    rotation at the mean level of the GPU run — timed for ~`seconds`, then scaled by the GPU run's op count."""
    ell = max(2, int(round(stats["keyswitch_limbs"] / max(1, stats["keyswitch"]))))
    rng = np.random.default_rng(7)
    mods = [int(m) for m in eng.moduli]
    evk = np.stack([rng.integers(0, m, size=eng.N, dtype=np.uint64) for _ in range(2 * eng.dnum_digits) for m in mods])
    evk = evk.reshape(eng.dnum_digits, 2, eng.n_limbs, eng.N)
    ct = np.stack([orc.uniform_residues(5 + 1000 * p, eng.q[:ell], eng.N) for p in range(2)])
    g = orc.galois(eng.log_n, 128)
    orc.rotate(ct, evk, g, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)     # warm tables
    t0, n = time.perf_counter(), 0
    while True:
        orc.rotate(ct, evk, g, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    per_rot = dt / n
    return per_rot * stats["keyswitch"] * 1e3, ell, n, dt


def main():
    args = parse()
    import torch  # first: the engine then shares torch's HIP runtime (same libamdhip64 soname)
    import numpy as np
    import fhe_linformer_amd as fa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:  # rehearsal on a box with fewer GPUs than ranks: every rank computes on GPU 0, collectives over gloo
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.micro:
        return micro(fa)

    import oracle as orc                                    # checker + cpu_baseline leg only
    from oracle import plain_forward as pf, circuit_sim as cs
    from fhe_linformer_amd import linformer as lf, shard

    n_q = args.n_q if args.workload == "forward" else 24
    n_p = args.n_p if args.n_p > 0 else -1
    eng = fa.Engine("bench", device=local_rank, seed=2024 + rank, log_n=args.log_n, n_q=n_q, n_p=n_p)
    cores = host_cpus()
    orc.set_threads(cores)
    line = {}

    fwd = None
    if args.workload == "forward":
        eng.keygen()
        eng.gen_relin_key()
        eng.gen_rotation_keys(fa.circuit_rotation_indices())   # +-2^i and +-3*2^i (merged tree steps)
        eng.bootstrap_setup(3, 3, 16384)
        w = pf.synthetic_model(1234)
        S = args.tokens
        ctl = lf.GpuController(eng)
        samples = []
        for i in range(args.warmup + args.steps):           # every step gets its own sample (seeded per rank)
            x = pf.synthetic_tokens(S, 4321 + 1000 * rank + (0 if i < args.warmup else i - args.warmup))
            x_in, X_E, X_F = pf.client_inputs(w, x)
            samples.append((x, lf.encrypt_inputs(ctl, x_in, X_E, X_F)))   # client side: resident in HBM before timing
        for i in range(args.warmup):
            lf.forward_encrypted(ctl, w, samples[i][1])
        eng.sync()
        torch.cuda.synchronize()
        eng.stats(reset=True)
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        logits = []
        for i in range(args.warmup, args.warmup + args.steps):
            out = lf.forward_encrypted(ctl, w, samples[i][1])
            logits.append(lf.logits_from_slots(eng.decrypt(out)))
        eng.sync()
        torch.cuda.synchronize()
        if dist:
            # the path's only collective: gather of the per-sample logits over RCCL/xGMI (a few hundred bytes)
            gathered = shard.gather_results(dist, np.array(logits), max_rows=args.steps)
            assert sum(len(g) for g in gathered) == args.steps * world
            torch.cuda.synchronize()
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            elapsed = shard.max_over_ranks(dist, elapsed)
        stats = eng.stats()
        for k in stats:
            stats[k] = stats[k] // max(1, args.steps)       # per sample
        # parity of the timed path: last sample vs the same op sequence in the clear (oracle/circuit_sim.py)
        x_last = samples[-1][0]
        ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *pf.client_inputs(w, x_last)))
        err = float(np.max(np.abs(logits[-1] - ref)))
        top2 = np.sort(ref)[-2:]
        decided = (top2[1] - top2[0]) > 4e-2           # arg-max is only meaningful when the oracle's margin exceeds the tolerance
        assert err < 2e-2, f"encrypted logits differ from the circuit oracle ({err})"
        assert not decided or int(np.argmax(logits[-1])) == int(np.argmax(ref)), "encrypted prediction differs from the circuit oracle"
        fwd = {"elapsed": elapsed, "stats": stats, "logit_err_vs_circuit_oracle": err, "pred": int(np.argmax(logits[-1]))}
        for _, enc in samples:
            del enc
        samples = None

    # ---- NTT section (second headline metric + roofline of the dominant kernel) -------------------------
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    n_ntt, ntt_ms, one, nq = ntt_section(eng, orc, np, args.ntt_batch, args.ntt_steps)
    torch.cuda.synchronize()
    ntt_elapsed = time.perf_counter() - t0
    if dist:
        ntt_ms = shard.max_over_ranks(dist, ntt_ms)
    ntt_rate = n_ntt * world / (ntt_ms * 1e-3)

    if rank == 0:
        alg_bytes = 16.0 * eng.N                            # read N + write N u64 per limb-NTT
        achieved = n_ntt * alg_bytes / (ntt_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("ntt_bytes_per_limb_transform")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "ntt_cols_kernel + ntt_rows_kernel (one limb-NTT = one tile pass of each)",
                    "limb_ntt_per_s_per_gpu": round(n_ntt / (ntt_ms * 1e-3), 1),
                    "algorithmic_bytes_per_limb_ntt": alg_bytes,
                    "note": "64-bit modular-integer butterflies: bound by VALU issue slots (88 % busy, 15 instr/butterfly), ceiling ~2.3 TB/s algorithmic at the sustained clock (DESIGN.md §6)"}
        if args.workload == "forward":
            value = fwd["elapsed"] * 1e3 / (args.steps * world)
            line = {
                "metric": "encrypted Linformer-d128 forward ms/sample", "value": round(value, 2), "unit": "ms/sample",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(fwd["elapsed"] * 1e3 / args.steps, 2), "higher_is_better": False, "scaling": "weak",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"forward: 1 sample/GPU/step, S={args.tokens}+CLS tokens, d=128, k=32, FFN 512, 20 classes, "
                                       f"N=2^{eng.log_n}, 16384 slots, {eng.n_q}+{eng.n_p} limbs, dnum 4, 8 bootstraps",
                           "ops_per_sample": fwd["stats"], "parallelism": f"independent samples x{world}",
                           "logit_err_vs_circuit_oracle": round(fwd["logit_err_vs_circuit_oracle"], 5)},
                "ntt": {"metric": "NTT/s at N=2^16", "value": round(ntt_rate, 1), "unit": "limb-NTT/s",
                        "workload": f"fwd+inv NTT of {args.ntt_batch} ciphertexts x 2 polys x {nq} limbs per GPU"},
                "roofline": roofline,
            }
        else:
            line = {
                "metric": "NTT/s at N=2^16 (limb-NTTs per second)", "value": round(ntt_rate, 1), "unit": "limb-NTT/s",
                "n_gpus": world, "steps": args.ntt_steps, "warmup": 3, "ms_per_step": round(ntt_ms / args.ntt_steps, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"ntt: fwd+inv negacyclic NTT of {args.ntt_batch} ciphertexts x 2 polys x {nq} limbs, N=2^{eng.log_n} per GPU",
                           "parallelism": f"independent ciphertexts x{world}"},
                "roofline": roofline,
            }
        if not args.no_cpu_baseline and world == 1:
            if args.workload == "forward":
                ms, ell, n, dt = cpu_forward_baseline(eng, orc, np, fwd["stats"], args.cpu_seconds, cores)
                line["cpu_baseline"] = {
                    "value": round(ms, 1), "unit": "ms/sample", "cores": cores, "kind": "port",
                    "sample": f"oracle/fhe_oracle.c orc_rotate (hybrid key switch + automorphism, OpenMP over limbs) at the GPU run's mean "
                              f"level ell={ell}, N=2^{eng.log_n}: {n} rotations in {dt:.1f}s, scaled by the {fwd['stats']['keyswitch']} key switches "
                              f"of one forward pass (lower bound: rescales, ct x pt products and encodes not counted)"}
            else:
                rate, n, dt = cpu_ntt_baseline(eng, orc, np, one, nq, args.cpu_seconds, cores)
                line["cpu_baseline"] = {"value": round(rate, 1), "unit": "limb-NTT/s", "cores": cores, "kind": "port",
                                        "sample": f"oracle/fhe_oracle.c orc_ntt_batch: fwd+inv NTT of 1 ciphertext (2x{nq} limbs, N=2^{eng.log_n}), "
                                                  f"OpenMP over limbs, {n} limb-NTTs in {dt:.1f}s"}
        print(json.dumps(line))
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
