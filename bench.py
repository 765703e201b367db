#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native RNS-CKKS engine (BASELINE.json metric).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N>1 the driver launches it under torch.distributed.run, one rank per GPU (RCCL).

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  workload "ntt"    : forward NTT then inverse NTT of B ciphertexts x 2 polys x 24 limbs at N=2^16
                      (SURVEY.md §8(d) NTT micro-benchmark; SplitMix64(0x5EED0001 + limb) residues)
The path shards embarrassingly (independent ciphertexts): every rank processes its own B ciphertexts
(weak scaling), no data-path collective; a final RCCL all_gather of per-rank checksums stands in for the
"gather of results" and doubles as a cross-rank bit-exactness check.
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
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="ciphertexts per step per GPU")
    ap.add_argument("--preset", default="bench")
    ap.add_argument("--micro", action="store_true", help="instruction-rate probes instead of the benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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
    return n


def micro(fa):
    eng = fa.Engine("toy", device=0)
    names = ["v_mul_lo_u32 x8", "v_mul_hi_u32 x8", "v_mad_u64_u32 x8", "harvey butterfly x4", "v_fma_f64 x64",
             "add_u64 x8", "mulhi64 x8", "mullo64 x8"]
    per_iter = [8, 8, 8, 4, 64, 8, 8, 8]
    blocks, iters = 256 * 8, 2048
    out = {}
    for v, (nm, k) in enumerate(zip(names, per_iter)):
        ms = min(eng.microbench(v, iters, blocks) for _ in range(3))
        ops = blocks * 256 * iters * k
        out[nm] = {"ms": round(ms, 4), "Gop_s": round(ops / ms / 1e6, 1),
                   "ops_per_clk_per_CU@2.4GHz": round(ops / (ms * 1e-3) / 2.4e9 / 256, 2)}
    eng.close()
    print(json.dumps({"micro": out}))


def cpu_baseline(eng, orc, x_one_ct, seconds):
    """Oracle (CPU port) timed on this host: forward+inverse NTT of ONE ciphertext (48 limb vectors),
    OpenMP over limb vectors on all host cores, repeated for ~`seconds`."""
    import numpy as np
    cores = host_cpus()
    orc.set_threads(cores)
    d = np.ascontiguousarray(x_one_ct.reshape(-1, eng.N)).copy()
    orc.ntt_batch(d, eng.q, eng.psi_q, inplace=True)
    orc.ntt_batch(d, eng.q, eng.psi_q, inverse=True, inplace=True)  # warm tables
    t0, n = time.perf_counter(), 0
    while True:
        orc.ntt_batch(d, eng.q, eng.psi_q, inplace=True)
        orc.ntt_batch(d, eng.q, eng.psi_q, inverse=True, inplace=True)
        n += 2 * d.shape[0]
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": round(n / dt, 1), "unit": "limb-NTT/s", "cores": cores, "kind": "port",
            "sample": f"oracle/fhe_oracle.c orc_ntt_batch: fwd+inv NTT of 1 ciphertext (2x{eng.n_q} limbs, N=2^{eng.log_n}), "
                      f"OpenMP over limbs, {n} limb-NTTs in {dt:.1f}s"}


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
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.micro:
        return micro(fa)

    import oracle as orc  # checker + cpu_baseline leg only
    eng = fa.Engine(args.preset, device=local_rank)
    B, nq, N = args.batch, eng.n_q, eng.N
    nvec = B * 2 * nq
    one = np.stack([orc.uniform_residues(0x5EED0001 + 1000 * p, eng.q, N) for p in range(2)])  # [2][nq][N]
    host = np.ascontiguousarray(np.broadcast_to(one, (B,) + one.shape))
    buf = eng.upload(host)

    def step():
        eng.ntt(buf, nvec)
        eng.ntt(buf, nvec, inverse=True)

    for _ in range(args.warmup):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms = eng.timer_stop()   # HIP events on the engine's stream (also drains it)
    torch.cuda.synchronize()
    chk = int(np.bitwise_xor.reduce(buf.download((8,), np.uint64)))
    if dist:
        t = torch.tensor([chk & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device="cuda")
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allc, t)   # the path's only collective: gather of per-rank results over RCCL/xGMI
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        assert all(int(c.item()) == int(allc[0].item()) for c in allc), "ranks disagree on the result checksum"

    # parity of the timed data path: after K x (NTT, INTT) the buffer must equal the input bit for bit
    back = buf.download(host.shape)
    assert np.array_equal(back, host), "INTT(NTT(x)) != x after the timed region"

    if rank == 0:
        transforms = 2 * nvec * args.steps                       # limb-NTTs per rank
        value = transforms * world / elapsed
        alg_bytes = 16.0 * N                                     # read N + write N u64 per limb-NTT
        achieved = transforms * alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("ntt_bytes_per_limb_transform")
            except Exception:
                traffic = None
        line = {
            "metric": "NTT/s at N=2^16 (limb-NTTs per second; one of BASELINE.json's two headline metrics)",
            "value": round(value, 1), "unit": "limb-NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"ntt: fwd+inv negacyclic NTT of {B} ciphertexts x 2 polys x {nq} limbs, N=2^{eng.log_n} per GPU",
                       "preset": args.preset, "batch_ciphertexts_per_gpu": B, "parallelism": f"independent ciphertexts x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "ntt_cols_kernel + ntt_rows_kernel (one limb-NTT = one tile pass of each)",
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 4),
                         "algorithmic_bytes_per_limb_ntt": alg_bytes},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(eng, orc, one, args.cpu_seconds)
        print(json.dumps(line))
    buf.free()
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
