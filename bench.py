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
`--workload ops` (and, in short form, every default run under the key "ops") times the leaf operations of SURVEY.md
§8(d) at fixed shapes (N=2^16, 24+6 limbs, batch 8, ell in {24,16,8}): ct x pt, rescale, rotate, mult+relin, the merged
rotate-sum, rotsum(128,128), one matmulRE row and 128 batched rows — HIP-event time, algorithmic bytes by the §8(d)
formulas and the resulting fraction of the HBM roofline per op.

Multi-GPU: every rank creates the SAME keys (one key seed, checked by an all-gather of a key checksum) — keys are
replicated, samples are sharded.  Default: one sample per GPU per step (weak scaling).  `--batch B`: a step is a batch of
B samples split over the ranks (BASELINE config 4: B=64 over 8 GPUs; strong scaling in the batch).
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
    ap.add_argument("--workload", choices=["forward", "ntt", "ops"], default="forward")
    ap.add_argument("--batch", type=int, default=0, help="samples per step over ALL ranks (0: one per rank per step); a rank's share of a "
                    "step goes through ONE engine as ONE batched pass (linformer.BatchedController: BASELINE config 4's per-GPU unit)")
    ap.add_argument("--pass-width", type=int, default=4, help="with --batch: at most this many samples travel in ONE pass (a pass keeps ~17 GB of "
                    "intermediates per sample at N=2^16 next to ~50 GB of keys: 4 fit 288 GB with room for the resident inputs, 8 do not); "
                    "a rank's share of a step runs as ceil(share / width) passes")
    ap.add_argument("--lanes", type=int, default=2, help="with --batch: a pass runs as this many sub-batches on as many lanes (HIP streams) of the ONE "
                    "context, concurrently on the GPU (linformer.LanedBatchedController); 1: one launch set for the whole pass")
    ap.add_argument("--dataflow", action="store_true", help="one sample per pass: the independent branches of the driver's circuit on different lanes "
                    "of the context (linformer.DataflowController; measured at no gain in round 4, DESIGN.md 6d: off by default)")
    ap.add_argument("--batch-loop", action="store_true", help="with --batch: a rank runs its samples one after another instead (A/B)")
    ap.add_argument("--resident-gb", type=float, default=75.0, help="with --batch: device memory for the resident input sets of the "
                    "timed region; steps cycle through the sets that fit (every pass still does all of its work)")
    ap.add_argument("--shard-rows", action="store_true", help="batch-1 latency mode: ONE sample per step, the rows inside its matmul / "
                    "unwrap loops split over the ranks (all-gather of ciphertext rows over RCCL/xGMI per row loop)")
    ap.add_argument("--no-ops", action="store_true", help="skip the short leaf-op section of a forward run")
    ap.add_argument("--no-level-plan", action="store_true", help="run every pass at the levels the driver asks for (no recorded level plan)")
    ap.add_argument("--throughput-batch", type=int, default=4, help="after the timed region (one sample per pass: `value`), also measure passes that "
                    "carry this many samples through the SAME engine (one key set: linformer.BatchedController); 0/1: skip")
    ap.add_argument("--no-reference-ring", dest="reference_ring", action="store_false", help="skip the extra passes at N=2^15, the ring the reference "
                    "is built with (reported as ms_per_sample_at_the_reference_ring_n15; `value` stays BASELINE's N=2^16)")
    ap.add_argument("--inflight", type=int, default=0, help="after the timed region (one sample in flight: `value`), also measure the "
                    "throughput with this many samples in flight on the GPU, one engine (context, stream, host thread) each; 0/1: skip")
    ap.add_argument("--forward-only", action="store_true", help="profiling runs: only the timed forward passes (no eager comparison, "
                    "no NTT / op sections, no CPU leg); prints a reduced line without `roofline`")
    ap.add_argument("--key-seed", type=int, default=2024, help="deterministic key seed, the same on every rank (replicated keys)")
    ap.add_argument("--tokens", type=int, default=129, help="tokens per sample (S); S+1 rows incl. CLS, 128 < S+1 <= 256")
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--variant", choices=["main", "main_2"], default="main", help="the driver: src/main.cpp as built (attention for the CLS query "
                    "only) or src/main_2.cpp (attention for every token; BASELINE config 5 names it with --log-n 17 --n-q 30)")
    ap.add_argument("--n-q", type=int, default=28)
    ap.add_argument("--n-p", type=int, default=0, help="special limbs (0: OpenFHE's rule ceil(widest digit bits / 60) = 7 for 28 limbs)")
    ap.add_argument("--special-bits", type=int, default=60, help="size of the special primes of hybrid key switching (60 = OpenFHE's, the reference's; "
                    "an experiment knob: primes below 2^53 take the fully lazy NTT path in both directions)")
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
    # the probes live in their own library (tools/probe/, built on demand): they are measurement code, not part of the product .so
    import ctypes as C
    import subprocess
    pdir = os.path.join(ROOT, "tools", "probe")
    subprocess.check_call(["make", "-s", "-C", pdir])
    plib = C.CDLL(os.path.join(pdir, "libfhelin_probe.so"))
    plib.fhelin_probe_microbench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]

    class _Probe:
        def microbench(self, variant, iters, blocks):
            ms = C.c_float()
            rc = plib.fhelin_probe_microbench(0, variant, iters, blocks, C.byref(ms))
            if rc:
                raise SystemExit(f"probe failed ({rc})")
            return ms.value

        def close(self):
            pass
    eng = _Probe()
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


def _timed(fn, budget_s, min_reps=1):
    fn()                                                     # warm tables / caches
    t0, n = time.perf_counter(), 0
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= min_reps:
            return dt / n, n


def cpu_forward_baseline(eng, orc, np, stats, seconds, cores):
    """CPU port of the forward pass's residue work, EXTRAPOLATED from timed leaf operations x the GPU run's operation
    counts (a full CPU pass takes minutes): hybrid key-switched rotation, rescale and ct x pt product, each timed at the
    mean level the GPU run executed it at, with the oracle's Barrett build (libfhe_oracle_fast.so — no hardware division;
    bit-identical to the by-definition build) and OpenMP over limbs; then the same ops single-threaded."""
    orc.use_fast(True)
    try:
        mean = lambda tot, cnt: max(2, int(round(tot / max(1, cnt))))
        ell_ks, ell_rs, ell_pt = mean(stats["keyswitch_limbs"], stats["keyswitch"]), mean(stats["rescale_limbs"], stats["rescale"]), \
            mean(stats["ct_pt_limbs"], stats["ct_pt_mult"])
        rng = np.random.default_rng(7)
        mods = [int(m) for m in eng.moduli]
        evk = np.stack([rng.integers(0, m, size=eng.N, dtype=np.uint64) for _ in range(2 * eng.dnum_digits) for m in mods])
        evk = evk.reshape(eng.dnum_digits, 2, eng.n_limbs, eng.N)
        g = orc.galois(eng.log_n, 128)
        ct = lambda ell: np.stack([orc.uniform_residues(5 + 1000 * p, eng.q[:ell], eng.N) for p in range(2)])
        c_ks, c_rs, c_pt = ct(ell_ks), ct(ell_rs), ct(ell_pt)
        ops = {
            "rotate": (lambda: orc.rotate(c_ks, evk, g, eng.alpha, eng.q, eng.p, eng.psi_q, eng.psi_p), stats["keyswitch"], ell_ks),
            "rescale": (lambda: orc.rescale(c_rs, eng.q[:ell_rs], eng.psi_q[:ell_rs]), stats["rescale"], ell_rs),
            "ct_x_pt": (lambda: [orc.mul(c_pt[p], c_pt[0], eng.q[:ell_pt]) for p in range(2)], stats["ct_pt_mult"], ell_pt),
        }
        out = {}
        # one REAL composite stage timed end to end on the CPU port (not extrapolated): a matmulRE row as the library runs it by
        # default - ct x pt, rescale, rotsum(128,128) as merged key switches {s..7s},{8s..56s},{64s}, + bias
        # (reference src/FHEController.cpp:869-883) - through oracle/residue_eval.py on the exported rotation keys
        try:
            from oracle.residue_eval import ResidueEvaluator, RCt
            idx = [128 * m for m in range(1, 8)] + [1024 * m for m in range(1, 8)] + [8192]
            keys = {r: eng.key_export(1, r) for r in idx}
            rev = ResidueEvaluator(eng.q, eng.p, eng.psi_q, eng.psi_p, eng.alpha, eng.log_n, keys, eng.params.log_slots)
            ns = 1 << eng.params.log_slots
            wpt, bpt = eng.encode(rng.uniform(-1, 1, ns) / 8), eng.encode(rng.uniform(-1, 1, ns))
            cache = {}

            def enc_of(pt):
                def f(ell, sc):
                    k = (id(pt), ell, float(sc))
                    if k not in cache:
                        cache[k] = eng.pt_export(pt, ell, sc)
                    return cache[k]
                return f
            ell_row = min(eng.n_q, ell_ks + 1)
            row = RCt(ct(ell_row), 1, rev.sf[eng.n_q - ell_row])
            orc.set_threads(cores)
            fn = lambda: rev.matmul_pt([row], enc_of(wpt), enc_of(bpt), 128, 128)
            per, n = _timed(fn, 0.15 * seconds)
            out["composite"] = {"stage": "matmulRE row: ct x pt, rescale, rotsum(128,128) as three merged key switches, + bias "
                                         "(src/FHEController.cpp:869-883), timed end to end on the CPU port",
                                "ell": ell_row, "cores": cores, "ms_per_row": round(per * 1e3, 2), "rows_timed": n,
                                "ms_for_the_130_rows_of_one_call": round(per * 1e3 * 130, 1)}
            del keys, rev, cache
        except Exception as ex:           # a diagnostic leg: never fail the bench line over it
            out["composite"] = {"error": repr(ex)}
        for label, threads, budget in (("all", cores, 0.5 * seconds), ("single", 1, 0.35 * seconds)):
            orc.set_threads(threads)
            total, detail = 0.0, {}
            for name, (fn, count, ell) in ops.items():
                per, n = _timed(fn, budget * (0.8 if name == "rotate" else 0.1))
                total += per * count
                detail[name] = {"ms": round(per * 1e3, 3), "ell": ell, "count_per_sample": count, "timed_reps": n}
            out[label] = (total * 1e3, detail)
        orc.set_threads(cores)
        return out
    finally:
        orc.use_fast(False)


# ---- leaf-operation benchmarks ------------------------------------------------------------------
def ops_section(eng, np, ells=(24, 16, 8), batch=8, reps=10, big_rows=128, short=False):
    """eng: the "bench" preset (N=2^16, 24+6 limbs, alpha 6) with keys.  Returns a list of per-op records.
    Algorithmic bytes per ciphertext (beta = ceil(ell/alpha) digits, k special limbs, one limb = 8N bytes):
      ct x pt 5 ell | ct + ct 6 ell | rescale 4 ell - 2 | rotation / key switch 3 ell + 2 beta (ell + k)
      mult + relin 6 ell + 2 beta (ell + k)  (4 ell in, 2 ell out, the relinearisation key)
      rotate-sum {s,2s,3s} = two steps of the reference's rotsum loop = 2 rotations + 2 ct + ct
      rotsum(128,128) = 7 rotations + 7 ct + ct ; matmulRE row = ct x pt + rotsum(128,128) + ct + pt (3 ell)
    `frac` = those bytes / HIP-event time / 8 TB/s: the share of the HBM roofline the OPERATION (as the reference
    formulates it) reaches; batching lets rows share one evaluation-key read, so it is not a per-kernel traffic figure."""
    N, k, alpha = eng.N, eng.n_p, eng.alpha
    limb = 8.0 * N
    L1 = eng.n_q
    rng = np.random.default_rng(3)
    ns = 1 << eng.params.log_slots
    w = eng.encode(rng.uniform(-1, 1, ns) / 8)
    bias = eng.encode(rng.uniform(-1, 1, ns))
    recs = []

    def timed(fn, n_ct, reps_):
        for _ in range(2):
            fn()
        eng.sync()
        eng.timer_start()
        for _ in range(reps_):
            fn()
        return eng.timer_stop() / reps_ / n_ct               # ms per ciphertext

    def rec(op, ell, n_ct, ms_per_ct, limbs):
        b = limbs * limb
        recs.append({"op": op, "ell": ell, "batch": n_ct, "us_per_ct": round(ms_per_ct * 1e3, 2), "alg_MB_per_ct": round(b / 1e6, 2),
                     "GBps": round(b / (ms_per_ct * 1e-3) / 1e9, 1), "frac": round(b / (ms_per_ct * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})

    for ell in ells:
        beta = -(-ell // alpha)
        ks = 3 * ell + 2 * beta * (ell + k)
        xs = [eng.encrypt(rng.uniform(-1, 1, ns), level=L1 - ell) for _ in range(batch)]
        ys = [eng.encrypt(rng.uniform(-1, 1, ns), level=L1 - ell) for _ in range(batch)]
        prods = eng.mult_plain_batch(xs, w)                  # degree 2: what rescale consumes
        rec("ct_x_pt", ell, batch, timed(lambda: eng.mult_plain_batch(xs, w), batch, reps), 5 * ell)
        rec("ct_plus_ct", ell, batch, timed(lambda: eng.add_batch(xs, ys), batch, reps), 6 * ell)
        rec("rescale", ell, batch, timed(lambda: eng.rescale_batch(prods), batch, reps), 4 * ell - 2)
        for r in ((1,) if short else (1, 128, -1)):
            rec(f"rotate({r})", ell, batch, timed(lambda: eng.rotate_batch(xs, r), batch, reps), ks)
        rec("mult_relin", ell, batch, timed(lambda: eng.mult_batch(xs, ys), batch, reps), 6 * ell + 2 * beta * (ell + k))
        rec("rotate_sum{128,256,384}", ell, batch, timed(lambda: eng.rotate_sum(xs, [128, 256, 384]), batch, reps), 2 * ks + 2 * 6 * ell)
        rotsum_l = 7 * (ks + 6 * ell)
        rec("rotsum(128,128)", ell, 1, timed(lambda: eng.rotsum(xs[0], 128, 128), 1, reps), rotsum_l)
        row_l = 5 * ell + rotsum_l + 3 * ell
        rec("matmulRE_row", ell, 1, timed(lambda: eng.matmulRE(xs[:1], w, bias), 1, reps), row_l)
        if not short or ell == ells[0]:
            # distinct ciphertexts (identical rows of one call are evaluated once) and eager evaluation (deferred rows would
            # return handles without doing the work): every one of the big_rows trees is computed inside the timed region
            rows = eng.encrypt_batch(rng.uniform(-1, 1, (big_rows, ns)), level=L1 - ell)
            eng.set_lazy_rows(False)
            try:
                rec("matmulRE_rows", ell, big_rows, timed(lambda: eng.matmulRE(rows, w, bias), big_rows, max(2, reps // 4)), row_l)
            finally:
                eng.set_lazy_rows(True)
            del rows
        del xs, ys, prods
    return recs


def inflight_section(fa, lf, pf, np, eng, ctl, w, S, args, use_plan, n_src, plan, passes=3):
    """K samples in flight on ONE GPU: K engines (contexts with their own HIP stream, the same replicated keys), one host
    thread each, every engine running the same driver on its own samples.  The single-ciphertext chains of one sample
    (Chebyshev evaluations, the pooler's bootstrap: launches that fill a fraction of the GPU) overlap with the batched row
    loops of another.  Returns ms per sample over K x passes samples, and the same engines one after the other."""
    import threading
    K = args.inflight
    eng.trim()                                   # the first engine's pool holds the timed region's samples: hand the memory back
    engines = [(eng, ctl)]
    for k in range(1, K):
        e = fa.Engine("bench", device=eng.params.device, seed=args.key_seed, log_n=args.log_n, n_q=eng.n_q, n_p=eng.n_p)
        e.keygen()
        e.gen_relin_key()
        e.gen_rotation_keys(fa.circuit_rotation_indices())
        e.bootstrap_setup(3, 3, 16384)
        if use_plan:
            e.set_level_plan(plan)               # the plan is a property of the driver program, not of the engine
        engines.append((e, lf.GpuController(e)))
    work = []
    for k, (e, c) in enumerate(engines):
        enc = []
        for i in range(passes + 1):
            if use_plan:
                e.level_plan_begin("apply")
            enc.append(lf.encrypt_inputs(c, *pf.client_inputs(w, pf.synthetic_tokens(S, 8000 + 100 * k + i))))
        e.sync()
        work.append(enc)

    def run(k, lo, hi):
        e, c = engines[k]
        for i in range(lo, hi):
            if use_plan:
                e.level_plan_begin("apply", first_source=n_src)
            e.decrypt(lf.forward_encrypted(c, w, work[k][i]))
        e.sync()

    for k in range(K):
        run(k, 0, 1)
    t0 = time.perf_counter()
    for k in range(K):
        run(k, 1, passes + 1)
    serial = (time.perf_counter() - t0) * 1e3 / (K * passes)
    th = [threading.Thread(target=run, args=(k, 1, passes + 1)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    conc = (time.perf_counter() - t0) * 1e3 / (K * passes)
    work = None
    for e, _ in engines[1:]:
        e.close()
    return {"samples_in_flight": K, "ms_per_sample": round(conc, 2), "same_engines_one_at_a_time_ms_per_sample": round(serial, 2),
            "samples": K * passes,
            "note": "throughput figure: K engines (context + HIP stream + host thread each, replicated keys) on ONE GPU; per-sample latency is "
                    "about K x this.  `value` is the one-sample-at-a-time pass."}


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
    if args.workload == "ops":
        n_p = 6
    # replicated keys: the SAME deterministic key seed on every rank (a deployment ships one client's key set to all GPUs)
    eng = fa.Engine("bench", device=local_rank, seed=args.key_seed, log_n=args.log_n, n_q=n_q, n_p=n_p, special_bits=args.special_bits)
    cores = host_cpus()
    orc.set_threads(cores)
    line = {}
    per_rank = 1
    if args.batch > 0:
        if args.batch % world:
            raise SystemExit("--batch must be a multiple of the number of ranks")
        per_rank = args.batch // world

    fwd = None
    if args.workload == "forward":
        eng.keygen()
        eng.gen_relin_key()
        eng.gen_rotation_keys(fa.circuit_rotation_indices())   # +-2^i and +-3*2^i (merged tree steps)
        eng.bootstrap_setup(3, 3, 16384)
        w = pf.synthetic_model(1234)
        S = args.tokens
        ctl = lf.GpuController(eng)
        ctl0 = ctl                                               # the unsharded controller: client-side ingestion is per rank
        row_mode = bool(args.shard_rows and dist)
        if row_mode:
            ctl = shard.RowShardedController(ctl, dist, shard.EngineTransport(eng, device=args.dist_backend == "nccl"))
        if dist:
            # keys replicated: every rank must hold the same secret (same seed -> same ChaCha20 stream -> same keys)
            sk = np.frombuffer(eng.secret_seed(), dtype=np.uint8).astype(np.float64).reshape(1, -1)
            allsk = shard.gather_results(dist, sk, max_rows=1)
            assert all(np.array_equal(a, allsk[0]) for a in allsk), "ranks hold different keys"
        samples = []
        n_samples = (args.warmup + args.steps) * per_rank
        batched = args.batch > 0 and not row_mode and not args.batch_loop and per_rank >= 1
        # a rank's share of a step as passes of <= pass_width samples (equal widths: 8 -> 4 + 4, 6 -> 3 + 3)
        n_pass = -(-per_rank // max(1, args.pass_width)) if batched else 1
        while batched and per_rank % n_pass:
            n_pass += 1
        width = per_rank // n_pass if batched else 1
        laned = batched and args.lanes > 1 and width % args.lanes == 0
        bctl = (lf.LanedBatchedController(eng, width, args.lanes) if laned else lf.BatchedController(eng, width)) if batched else None
        n_sets = args.warmup + args.steps                    # distinct input sets (one set = a rank's samples of one step)
        if batched:
            n_sets = max(1, min(n_sets, int(args.resident_gb / (4.5 * per_rank * (eng.N / 65536.0)))))
            n_samples = n_sets * per_rank
        # level plan (include/fhelin.h fhelin_level_plan_*): ONE untimed pass of the same driver is recorded; every later pass
        # - the client's encryptions and the server's evaluation - starts each fresh encryption / bootstrap output with the
        # limbs the recording shows its consumers read.  Row-sharded runs exchange ciphertexts between ranks: not planned.
        # Row-sharded runs record the plan on the UNSHARDED driver (the plan is a property of the program: its sources - fresh
        # encryptions and bootstraps - are replicated calls, in the same order on every rank) and apply it to the sharded passes
        use_plan = not args.no_level_plan
        n_client_sources = 0
        plan = []
        if use_plan:
            x = pf.synthetic_tokens(S, 999)
            eng.level_plan_begin("record")
            enc_rec = lf.encrypt_inputs(ctl0, *pf.client_inputs(w, x))
            n_client_sources = sum(len(v) for v in enc_rec.values())      # the client's encryptions are the pass's first sources
            eng.decrypt(lf.forward_encrypted(ctl0, w, enc_rec, None, args.variant))
            plan = eng.level_plan_end()
            del enc_rec

        bplan = lf.batched_level_plan(plan, width, n_client_sources) if (use_plan and batched) else []

        flow = lf.DataflowController(ctl) if (not row_mode and args.dataflow and os.environ.get("FHELIN_LANES", "2") != "0") else None

        def server_pass(enc):
            if use_plan:
                eng.level_plan_begin("apply", first_source=n_client_sources)
            if flow is not None and eng.lazy_rows_on:
                flow.begin()
                r = lf.forward_encrypted(flow, w, enc, None, args.variant)
                flow.end()
                return r
            return lf.forward_encrypted(ctl, w, enc, None, args.variant)

        def server_pass_batched(encs):
            # passes of the driver with every value `width` ciphertexts wide (this rank's share of the step: n_pass passes)
            outs = []
            for lo in range(0, len(encs), width):
                if use_plan:
                    eng.level_plan_begin("apply", first_source=n_client_sources * width)
                if laned:
                    bctl.begin()
                outs.extend(lf.forward_encrypted(bctl, w, lf.batch_inputs(encs[lo:lo + width]), None, args.variant))
                if laned:
                    bctl.end()
            return outs

        eng.sync()
        eng.stats(reset=True)
        t_client = time.perf_counter()
        t_prep = 0.0
        for i in range(n_samples):                           # every step gets its own samples (seeded per rank)
            timed_idx = i if batched else i - args.warmup * per_rank
            tp = time.perf_counter()
            x = pf.synthetic_tokens(S, 4321 + (0 if row_mode else 100000 * rank) + max(0, timed_idx))   # row mode: every rank, same sample
            t_prep += time.perf_counter() - tp
            if use_plan:
                eng.level_plan_begin("apply")
            # client side (fhelin_client_ingest: positional embedding, Linformer projections, packing, encode, encrypt - on the device):
            # resident in HBM before timing
            samples.append((x, lf.ingest_sample(ctl0, w, x)))
        eng.sync()
        # plaintext prep + encode + encrypt of the sample's inputs, INCLUDING the growth of the device pool: every sample stays
        # resident for the timed region, so each one's ciphertexts are fresh hipMalloc blocks (counted below)
        client_first_ms = (time.perf_counter() - t_client) * 1e3 / max(1, n_samples)
        st = eng.stats()
        client_pool = {"hipMalloc_calls_per_sample": st["pool_malloc_calls"] // max(1, n_samples),
                       "hipMalloc_GB_per_sample": round(st["pool_malloc_bytes"] / 1e9 / max(1, n_samples), 2),
                       "hipMalloc_ms_per_sample": round(st["pool_malloc_ns"] / 1e6 / max(1, n_samples), 2),
                       "synthetic_token_generation_ms_per_sample": round(t_prep * 1e3 / max(1, n_samples), 2)}
        # the same ingestion into memory the pool already owns (a server that releases a sample's inputs when it is done):
        # one extra sample encrypted and dropped, then timed three times
        def ingest_once():
            xs = pf.synthetic_tokens(S, 777)
            if use_plan:
                eng.level_plan_begin("apply")
            e = lf.ingest_sample(ctl0, w, xs)
            eng.sync()
            del e
        ingest_once()
        t_client = time.perf_counter()
        for _ in range(3):
            ingest_once()
        client_ms = (time.perf_counter() - t_client) * 1e3 / 3
        def step_set(step):                                   # the input set a step runs on (batched: sets are cycled)
            return [samples[(step % n_sets) * per_rank + x] for x in range(per_rank)]
        if batched and use_plan:
            eng.set_level_plan(bplan)
        if batched:
            for st_ in range(args.warmup):
                server_pass_batched([e for _, e in step_set(st_)])
        else:
            for i in range(args.warmup * per_rank):
                server_pass(samples[i][1])
        eng.sync()
        torch.cuda.synchronize()
        eng.stats(reset=True)
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        logits = []
        timed_inputs = []
        host_enqueue = 0.0
        if batched:
            for st_ in range(args.warmup, args.warmup + args.steps):
                cur = step_set(st_)
                th = time.perf_counter()
                outs_b = server_pass_batched([e for _, e in cur])
                host_enqueue += time.perf_counter() - th
                for x in range(per_rank):
                    logits.append(lf.logits_from_slots(eng.decrypt(outs_b[x])))
                    timed_inputs.append(cur[x][0])
        else:
            for i in range(args.warmup * per_rank, n_samples):
                th = time.perf_counter()
                out = server_pass(samples[i][1])
                host_enqueue += time.perf_counter() - th           # host time to ISSUE the pass (the GPU runs behind asynchronously)
                logits.append(lf.logits_from_slots(eng.decrypt(out)))
                timed_inputs.append(samples[i][0])
        eng.sync()
        torch.cuda.synchronize()
        if batched and use_plan:
            eng.set_level_plan(plan)
        n_timed = args.steps * per_rank
        if dist:
            # the path's only collective: gather of the per-sample logits over RCCL/xGMI (a few hundred bytes)
            gathered = shard.gather_results(dist, np.array(logits), max_rows=n_timed)
            assert sum(len(g) for g in gathered) == n_timed * world
            torch.cuda.synchronize()
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            elapsed = shard.max_over_ranks(dist, elapsed)
        stats = eng.stats()
        pool_now = {"held_GB": round(stats.pop("pool_reserved_bytes") / 1e9, 1), "in_use_peak_GB": round(stats.pop("pool_live_peak_bytes") / 1e9, 1),
                    "held_peak_GB": round(stats.pop("pool_reserved_peak_bytes") / 1e9, 1), "out_of_memory_trims_in_the_timed_region": stats.pop("pool_trims")}
        for k in stats:
            stats[k] = stats[k] // max(1, n_timed)          # per sample
        # parity of the timed path: EVERY timed sample vs the same op sequence in the clear (oracle/circuit_sim.py),
        # after the timed region
        err, err_sum = 0.0, 0.0
        assert len(timed_inputs) == len(logits) == n_timed
        oracle_cache = {}
        for x_i, lg in zip(timed_inputs, logits):
            if id(x_i) not in oracle_cache:                    # a set that the steps cycled through more than once: one oracle pass
                oracle_cache[id(x_i)] = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *pf.client_inputs(w, x_i), None, args.variant))
            ref = oracle_cache[id(x_i)]
            e_i = float(np.max(np.abs(lg - ref)))
            top2 = np.sort(ref)[-2:]
            decided = (top2[1] - top2[0]) > 4e-2       # arg-max is only meaningful when the oracle's margin exceeds the tolerance
            assert e_i < 2e-2, f"encrypted logits differ from the circuit oracle ({e_i})"
            assert not decided or int(np.argmax(lg)) == int(np.argmax(ref)), "encrypted prediction differs from the circuit oracle"
            err = max(err, e_i)
            err_sum += e_i
        # the same pass with deferred rows OFF (every row of every batched call evaluated, read or not), outside the timed
        # region, for the record: the reference's CLS-only driver computes 129 query projections and 129 final token
        # expansions that nothing reads (src/main.cpp:183,:196,:416-424)
        eager_ms = float("nan")
        unplanned_ms = float("nan")
        literal_ms = float("nan")
        if not args.forward_only:
            leg_pool = {}

            def timed_passes(fn, n=2, name=""):
                for _ in range(3):                             # untimed: plaintext caches of this variant, and the device arena grown to
                    m0 = eng.stats()["pool_malloc_calls"]      # the variant's working set (a slab from the driver costs ~60 ms)
                    eng.decrypt(fn())
                    eng.sync()
                    if eng.stats()["pool_malloc_calls"] == m0:
                        break
                s0 = eng.stats()
                t1 = time.perf_counter()
                for _ in range(n):
                    eng.decrypt(fn())
                eng.sync()
                ms = (time.perf_counter() - t1) * 1e3 / n
                s1 = eng.stats()
                leg_pool[name] = {"hipMalloc_ms": round((s1["pool_malloc_ns"] - s0["pool_malloc_ns"]) / 1e6, 1),
                                  "out_of_memory_trims": s1["pool_trims"] - s0["pool_trims"], "held_GB": round(s1["pool_reserved_bytes"] / 1e9, 1),
                                  "limb_ntt_per_pass": (s1["limb_ntt"] - s0["limb_ntt"]) // n}
                return ms
            eng.set_lazy_rows(False)
            eager_ms = timed_passes(lambda: server_pass(samples[-1][1]), name="rows_eager_plan_on")
            eng.set_lazy_rows(True)
            # and the pass at the levels the driver asks for (no plan), on inputs encrypted at level 0 as the driver does
            if use_plan:
                eng.level_plan_begin("off")
                enc_full = lf.encrypt_inputs(ctl0, *pf.client_inputs(w, samples[-1][0]))
                unplanned_ms = timed_passes(lambda: lf.forward_encrypted(ctl, w, enc_full, None, args.variant), name="rows_deferred_plan_off")
                # ... with every row evaluated as well: the reference's literal operation sequence at the reference's own levels
                eng.set_lazy_rows(False)
                literal_ms = timed_passes(lambda: lf.forward_encrypted(ctl, w, enc_full, None, args.variant), name="literal")
                eng.set_lazy_rows(True)
                del enc_full
            else:
                unplanned_ms, literal_ms = elapsed * 1e3 / (n_timed * (1 if row_mode else world)), eager_ms
        else:
            if rank == 0:
                print(json.dumps({"metric": "encrypted Linformer-d128 forward ms/sample (profiling run)",
                                  "value": round(elapsed * 1e3 / (n_timed * world), 2), "unit": "ms/sample", "n_gpus": world,
                                  "steps": args.steps, "warmup": args.warmup, "log_n": args.log_n, "n_q": eng.n_q, "n_p": eng.n_p, "special_bits": args.special_bits, "variant": args.variant,
                                  "host_issue_ms_per_sample": round(host_enqueue * 1e3 / n_timed, 2), "ops_per_sample": stats, "device_pool": pool_now,
                                  "samples_per_pass": width, "passes_per_step": n_pass if batched else per_rank, "lanes": args.lanes if (batched and laned) else 1, "distinct_input_sets": n_sets,
                                  "logit_err_vs_circuit_oracle": round(err, 5), "level_plan": bool(plan)}))
            eng.close()
            if dist:
                dist.destroy_process_group()
            return
        fwd = {"elapsed": elapsed, "stats": stats, "logit_err_vs_circuit_oracle": err, "logit_err_mean": err_sum / max(1, len(logits)),
               "batched": batched, "n_sets": n_sets, "pool": pool_now, "leg_pool": leg_pool if not args.forward_only else {}, "width": width, "lanes": args.lanes if (batched and laned) else 1, "n_pass": n_pass if batched else per_rank, "host_enqueue_ms": host_enqueue * 1e3 / max(1, n_timed), "pred": int(np.argmax(logits[-1])),
               "samples_checked": len(logits), "eager_ms": eager_ms, "client_ms": client_ms, "unplanned_ms": unplanned_ms, "literal_ms": literal_ms,
               "client_first_ms": client_first_ms, "client_pool": client_pool,
               "plan": plan, "n_client_sources": n_client_sources}
        for _, enc in samples:
            del enc
        samples = None
        # ---- secondary figure: B samples per pass through the SAME engine (BASELINE config 4's per-GPU unit; `value` above stays one sample per pass)
        fwd["tb"] = None
        Bt = args.throughput_batch
        if Bt > 1 and world == 1 and not row_mode and not batched:
            eng.trim()
            t_lanes = 2 if Bt % 2 == 0 else 1                 # two sub-batches on two lanes (streams) of the one context
            tctl = lf.LanedBatchedController(eng, Bt, t_lanes) if t_lanes > 1 else lf.BatchedController(eng, Bt)
            xs = [pf.synthetic_tokens(S, 9100 + i) for i in range(Bt)]
            if use_plan:
                eng.set_level_plan(plan)
            encs_t = []
            for x in xs:
                if use_plan:
                    eng.level_plan_begin("apply")
                encs_t.append(lf.ingest_sample(ctl0, w, x))
            if use_plan:
                eng.set_level_plan(lf.batched_level_plan(plan, Bt, n_client_sources))

            def tpass():
                if use_plan:
                    eng.level_plan_begin("apply", first_source=n_client_sources * Bt)
                if t_lanes > 1:
                    tctl.begin()
                r = lf.forward_encrypted(tctl, w, lf.batch_inputs(encs_t), None, args.variant)
                if t_lanes > 1:
                    tctl.end()
                return r
            outs_t = tpass()
            lg_t = [lf.logits_from_slots(eng.decrypt(o)) for o in outs_t]
            eng.sync()
            eng.stats(reset=True)
            t1 = time.perf_counter()
            n_tp = 2
            for _ in range(n_tp):
                for o in tpass():
                    eng.decrypt(o)
            eng.sync()
            tb_ms = (time.perf_counter() - t1) * 1e3 / (n_tp * Bt)
            st_t = eng.stats()
            err_t = 0.0
            for x, lg in zip(xs, lg_t):
                ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *pf.client_inputs(w, x), None, args.variant))
                err_t = max(err_t, float(np.max(np.abs(lg - ref))))
            assert err_t < 2e-2, f"batched pass: encrypted logits differ from the circuit oracle ({err_t})"
            fwd["tb"] = {"samples_per_pass": Bt, "lanes": t_lanes, "ms_per_sample": round(tb_ms, 2), "passes_timed": n_tp, "logit_err_vs_circuit_oracle": round(err_t, 5),
                         "device_pool_in_use_peak_GB": round(st_t["pool_live_peak_bytes"] / 1e9, 1),
                         "device_pool_held_GB": round(st_t["pool_reserved_bytes"] / 1e9, 1),
                         "out_of_memory_trims": st_t["pool_trims"],
                         "note": "ONE engine, one key set, one plaintext cache: every call of the driver carries the rows of all samples of the pass "
                                 "(linformer.BatchedController), as two sub-batches on two HIP streams of the one context (LanedBatchedController); each sample ends in the residues of its own single pass (tests/test_batched_forward_gpu.py); "
                                 "per-sample latency is the whole pass.  `value` is the one-sample-per-pass figure."}
            if use_plan:
                eng.set_level_plan(plan)
            del encs_t, outs_t
        # ---- the same pass at the ring the reference is BUILT with (src/FHEController.cpp:12-13: SetRingDim(1 << 15), 16384 slots = full packing;
        # BASELINE.json quotes N=2^16, which is `value`): a second engine, its own keys and level plan, a few passes after the timed region
        fwd["n15"] = None
        if args.reference_ring and world == 1 and not row_mode and not batched and args.log_n == 16:
            eng.trim()
            e15 = fa.Engine("bench", device=local_rank, seed=args.key_seed, log_n=15, n_q=n_q, n_p=n_p, special_bits=args.special_bits)
            e15.keygen()
            e15.gen_relin_key()
            e15.gen_rotation_keys(fa.circuit_rotation_indices())
            e15.bootstrap_setup(3, 3, 16384)
            c15 = lf.GpuController(e15)
            x15 = pf.synthetic_tokens(S, 4321)
            n_src15 = 0
            if use_plan:
                e15.level_plan_begin("record")
                enc15 = lf.encrypt_inputs(c15, *pf.client_inputs(w, pf.synthetic_tokens(S, 999)))
                n_src15 = sum(len(v) for v in enc15.values())
                e15.decrypt(lf.forward_encrypted(c15, w, enc15, None, args.variant))
                e15.level_plan_end()
                del enc15
                e15.level_plan_begin("apply")
            enc15 = lf.ingest_sample(c15, w, x15)

            def pass15():
                if use_plan:
                    e15.level_plan_begin("apply", first_source=n_src15)
                return lf.forward_encrypted(c15, w, enc15, None, args.variant)
            lg15 = None
            for _ in range(3):                                 # untimed: plaintext caches, the arena
                m0 = e15.stats()["pool_malloc_calls"]
                lg15 = lf.logits_from_slots(e15.decrypt(pass15()))
                e15.sync()
                if e15.stats()["pool_malloc_calls"] == m0:
                    break
            e15.stats(reset=True)
            t1 = time.perf_counter()
            for _ in range(3):
                e15.decrypt(pass15())
            e15.sync()
            ms15 = (time.perf_counter() - t1) * 1e3 / 3
            st15 = e15.stats()
            ref15 = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, *pf.client_inputs(w, x15), None, args.variant))
            err15 = float(np.max(np.abs(lg15 - ref15)))
            assert err15 < 2e-2, f"N=2^15 pass: encrypted logits differ from the circuit oracle ({err15})"
            fwd["n15"] = {"ms_per_sample": round(ms15, 2), "limb_ntt_per_pass": st15["limb_ntt"] // 3, "logit_err_vs_circuit_oracle": round(err15, 5),
                          "ring": "N=2^15, 16384 slots (full packing), %d+%d limbs" % (e15.n_q, e15.n_p)}
            del enc15
            e15.close()
        # ---- K samples in flight on the one GPU, one engine EACH (replicated keys; off by default since round 4: the batched pass above shares one key set)
        fwd["inflight"] = None
        if args.inflight > 1 and world == 1 and not row_mode:
            fwd["inflight"] = inflight_section(fa, lf, pf, np, eng, ctl, w, S, args, use_plan, n_client_sources, plan)

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

    ops = None
    if rank == 0 and (args.workload == "ops" or (args.workload == "forward" and not args.no_ops and args.log_n == 16)):
        full = args.workload == "ops"
        oe = eng
        if args.workload != "ops":
            eng.sync()
            oe = fa.Engine("bench", device=local_rank, seed=args.key_seed)     # N=2^16, 24+6 limbs: SURVEY §8(d)'s op shapes
        oe.keygen()
        oe.gen_relin_key()
        oe.gen_rotation_keys([1, -1] + [128 << i for i in range(7)] + [384 << i for i in range(6)])
        ops = ops_section(oe, np, ells=(24, 16, 8) if full else (24, 8), reps=10 if full else 4, short=not full)
        if oe is not eng:
            oe.close()

    by_batch = None
    if rank == 0 and args.ntt_batch == 8:
        # SURVEY 8(d)'s other two batch sizes (one ciphertext; 64), outside the timed regions: the same section, fewer steps
        by_batch = {"8": round(n_ntt / (ntt_ms * 1e-3), 1)}
        for bsz, st in ((1, 30), (64, 4)):
            nb, msb, _, _ = ntt_section(eng, orc, np, bsz, st)
            by_batch[str(bsz)] = round(nb / (msb * 1e-3), 1)

    if rank == 0:
        alg_bytes = 16.0 * eng.N                            # read N + write N u64 per limb-NTT
        achieved = n_ntt * alg_bytes / (ntt_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")   # PMC passes of this round's kernels (tools/profile_kernels.sh)
        by_counters = None
        if os.path.exists(tpath) and eng.log_n == 16:     # the counters were collected at N=2^16
            try:
                traffic = json.load(open(tpath)).get("ntt_bytes_per_limb_transform")
                # what the HBM counters say the transforms move per second at the rate measured in THIS run (two tile passes, each reading
                # and writing its vectors once: twice the algorithmic bytes), as a fraction of the peak
                by_counters = round(traffic * n_ntt / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "frac_of_peak_by_counter_traffic": by_counters,
                    "kernel": "ntt_cols_kernel + ntt_rows_kernel (one limb-NTT = one tile pass of each)",
                    "limb_ntt_per_s_per_gpu": round(n_ntt / (ntt_ms * 1e-3), 1),
                    "limb_ntt_per_s_by_batch_of_ciphertexts": by_batch,
                    "algorithmic_bytes_per_limb_ntt": alg_bytes,
                    "traffic_source": "profiles/r04_pmc_traffic.json: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch / 768 limb vectors, measured in "
                                      "round 4 on these kernels in separate --pmc passes (a static file: the bench run itself does not collect counters)",
                    "note": "64-bit modular-integer butterflies: bound by VALU issue slots (92 % busy; 14 instr per forward / 15 per inverse butterfly, "
                            "9 of them 32x32 multiplies); 100 % VALU utilisation at the sustained 1.6 GHz would be 2.5 M limb-NTT/s = 0.31 of the HBM roofline "
                            "(instruction-count bound in DESIGN.md 6b)"}
        if args.workload == "forward":
            value = fwd["elapsed"] * 1e3 / (args.steps * per_rank * (1 if row_mode else world))
            line = {
                "metric": "encrypted Linformer-d128 forward ms/sample", "value": round(value, 2), "unit": "ms/sample",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(fwd["elapsed"] * 1e3 / args.steps, 2), "higher_is_better": False,
                "scaling": "strong" if (args.batch > 0 or row_mode) else "weak",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"forward ({'src/main.cpp' if args.variant == 'main' else 'src/main_2.cpp'} call sequence): {per_rank} sample(s)/GPU/step, S={args.tokens}+CLS tokens, d=128, k=32, FFN 512, 20 classes, "
                                       f"N=2^{eng.log_n}, 16384 slots, {eng.n_q}+{eng.n_p} limbs, dnum 4, {fwd['stats']['bootstrap']} bootstraps; "
                                       f"deferred rows on, level plan {'on' if fwd['plan'] else 'off'} (ms_per_sample_cells has the other three combinations)",
                           "ops_per_sample": fwd["stats"],
                           "parallelism": (f"ONE sample per step, rows of its matmul / unwrap loops over {world} ranks, keys replicated"
                                           if row_mode else f"independent samples x{world}, keys replicated (one key seed)"),
                           "samples_per_pass": fwd["width"],
                           "lanes_per_pass": fwd["lanes"],
                           "passes_per_step_per_gpu": fwd["n_pass"],
                           "samples_per_pass_note": ("ONE pass of the driver carries this many samples through one engine (linformer.BatchedController: one key set, "
                                                     "one plaintext cache, one launch set; every sample ends in the residues of its own single pass, "
                                                     "tests/test_batched_forward_gpu.py)" if fwd["batched"] else "one sample per pass"),
                           "distinct_input_sets_in_the_timed_region": fwd["n_sets"],
                           "device_pool": fwd["pool"],
                           "device_pool_in_the_other_cells": fwd["leg_pool"],
                           "host_issue_ms_per_sample": round(fwd["host_enqueue_ms"], 2),
                           "logit_err_vs_circuit_oracle": round(fwd["logit_err_vs_circuit_oracle"], 5),
                           "logit_err_vs_circuit_oracle_mean": round(fwd["logit_err_mean"], 5),
                           "logit_err_tolerance": 2e-2,
                           # flat copies of ms_per_sample_cells (nested objects do not survive into the driver's parsed record)
                           "ms_per_sample_rows_eager_plan_on": round(fwd["eager_ms"], 2),
                           "ms_per_sample_rows_deferred_plan_off": round(fwd["unplanned_ms"], 2),
                           "ms_per_sample_literal": round(fwd["literal_ms"], 2),
                           "samples_checked_vs_circuit_oracle": fwd["samples_checked"],
                           "deferred_rows": "on: rows of matmulRE / unwrapExpanded / matmulRElarge are evaluated when read - rows no later call "
                                            "reads never; rows of matmulRElarge that generate_containers takes unread go through the fused "
                                            "container sum (DESIGN.md 7g: same slot values, checked bit for bit against the oracle's restatement); "
                                            "ops_per_sample counts what was executed",
                           "ms_per_sample_with_every_row_evaluated": round(fwd["eager_ms"], 2),
                           "level_plan": ("off" if not fwd["plan"] else
                                          "on: one untimed pass of the same driver was recorded; fresh encryptions and bootstrap outputs start with "
                                          "the limbs their consumers read (values unchanged up to noise; every timed sample is checked)"),
                           "level_plan_limbs_per_source": ({} if not fwd["plan"] else {
                               "client_encryptions": {str(k): fwd["plan"][:fwd["n_client_sources"]].count(k)
                                                      for k in sorted(set(fwd["plan"][:fwd["n_client_sources"]]))},
                               "server_sources_in_call_order": fwd["plan"][fwd["n_client_sources"]:]}),
                           "ms_per_sample_without_level_plan": round(fwd["unplanned_ms"], 2),
                           # the 2 x 2 of the two driver-level optimisations; `value` is the first cell
                           "ms_per_sample_cells": {
                               "value_is": "deferred_rows=on, level_plan=" + ("on" if fwd["plan"] else "off"),
                               "deferred_rows=on,level_plan=on": round(value, 2) if fwd["plan"] else None,
                               "deferred_rows=off,level_plan=on": round(fwd["eager_ms"], 2) if fwd["plan"] else None,
                               "deferred_rows=on,level_plan=off": round(fwd["unplanned_ms"], 2),
                               "deferred_rows=off,level_plan=off (every call of the reference's driver evaluated in full where it is made, at the driver's own levels)": round(fwd["literal_ms"], 2),
                               "how": "value: the timed region (steps x passes, max over ranks); the other cells: 2 passes each after one untimed "
                                      "pass, same build, same process, after the timed region"},
                           "ms_per_sample_with_4_samples_per_pass": (fwd["tb"] or {}).get("ms_per_sample") if (fwd["tb"] or {}).get("samples_per_pass") == 4 else None,
                           "throughput_with_several_samples_per_pass": fwd["tb"],
                           "ms_per_sample_at_the_reference_ring_n15": (fwd["n15"] or {}).get("ms_per_sample"),
                           "pass_at_the_reference_ring_n15": fwd["n15"],
                           "throughput_with_samples_in_flight": fwd["inflight"],
                           "client_ingest_ms_per_sample": round(fwd["client_ms"], 2),
                           "client_ingest_note": "fhelin_client_ingest: positional embedding, the two Linformer projections, packing, encoding and encryption of the sample's "
                                                 "194 inputs on the device, into pool-owned memory (3 timed repeats); "
                                                 "first-touch figure below = the same while the pool grows through hipMalloc because "
                                                 "every sample of the run stays resident",
                           "client_ingest_first_touch_ms_per_sample": round(fwd["client_first_ms"], 2),
                           "client_ingest_first_touch_pool_growth": fwd["client_pool"]},
                "ntt": {"metric": f"NTT/s at N=2^{eng.log_n}", "value": round(ntt_rate, 1), "unit": "limb-NTT/s",
                        "workload": f"fwd+inv NTT of {args.ntt_batch} ciphertexts x 2 polys x {nq} limbs per GPU"},
                "roofline": roofline,
            }
        elif args.workload == "ops":
            best = max(ops, key=lambda r: r["frac"])
            line = {"metric": "leaf operations at N=2^16, 24+6 limbs (SURVEY 8(d))", "value": best["GBps"], "unit": "GB/s (best op, algorithmic)",
                    "n_gpus": world, "steps": 10, "warmup": 2, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                    "config": {"workload": "ops: ct x pt, ct + ct, rescale, rotate, mult+relin, merged rotate-sum, rotsum(128,128), matmulRE row / 128 rows; batch 8; ell 24/16/8"},
                    "roofline": roofline}
        else:
            line = {
                "metric": f"NTT/s at N=2^{eng.log_n} (limb-NTTs per second)", "value": round(ntt_rate, 1), "unit": "limb-NTT/s",
                "n_gpus": world, "steps": args.ntt_steps, "warmup": 3, "ms_per_step": round(ntt_ms / args.ntt_steps, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"ntt: fwd+inv negacyclic NTT of {args.ntt_batch} ciphertexts x 2 polys x {nq} limbs, N=2^{eng.log_n} per GPU",
                           "parallelism": f"independent ciphertexts x{world}"},
                "roofline": roofline,
            }
        if not args.no_cpu_baseline and world == 1 and args.workload != "ops":
            if args.workload == "forward":
                res = cpu_forward_baseline(eng, orc, np, fwd["stats"], args.cpu_seconds, cores)
                ms_all, detail = res["all"]
                ms_one, detail_one = res["single"]
                line["cpu_baseline"] = {
                    "value": round(ms_all, 1), "unit": "ms/sample", "cores": cores, "kind": "port",
                    "single_thread_value": round(ms_one, 1),
                    "sample": "EXTRAPOLATED, not a timed CPU forward pass: oracle/fhe_oracle.c built with Barrett reductions "
                              "(libfhe_oracle_fast.so, bit-identical to the by-definition build), OpenMP over limbs; hybrid key-switched "
                              f"rotation, rescale and ct x pt product timed for {args.cpu_seconds:.0f}s in all at N=2^{eng.log_n} at the mean level "
                              "the GPU run executed each at, multiplied by the GPU run's per-sample operation counts (merged rotations counted "
                              "in the reference's units); plaintext encodes, additions and host orchestration not counted (lower bound)",
                    "ops": detail, "ops_single_thread": detail_one, "timed_composite_stage": res.get("composite")}
                # the WHOLE pass timed on the CPU port (tools/cpu_forward_pass.py, ~2 minutes: too long for this run) - a static record
                rec = os.path.join(ROOT, "profiles", "r03_be_cpu_forward_pass.json")
                if os.path.exists(rec):
                    try:
                        full = json.load(open(rec))
                        line["cpu_baseline"]["timed_full_pass_record"] = {
                            "source": "profiles/r03_be_cpu_forward_pass.json (tools/cpu_forward_pass.py on a GPU box's host cores, round 3; not re-run here)",
                            "cpu_port_s": full.get("cpu_port_s"), "cpu_threads": full.get("cpu_threads"), "gpu_s_same_run": full.get("gpu_s"),
                            "ring": full.get("ring"), "same_residues_as_the_gpu_pass": full.get("same_residues_as_the_gpu_pass")}
                    except Exception:
                        pass
            else:
                rate, n, dt = cpu_ntt_baseline(eng, orc, np, one, nq, args.cpu_seconds, cores)
                line["cpu_baseline"] = {"value": round(rate, 1), "unit": "limb-NTT/s", "cores": cores, "kind": "port",
                                        "sample": f"oracle/fhe_oracle.c orc_ntt_batch: fwd+inv NTT of 1 ciphertext (2x{nq} limbs, N=2^{eng.log_n}), "
                                                  f"OpenMP over limbs, {n} limb-NTTs in {dt:.1f}s"}
        if ops is not None:
            line["ops"] = ops
        print(json.dumps(line))
    eng.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
