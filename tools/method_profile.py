#!/usr/bin/env python3
"""Where one forward pass spends its time, per FHEController method (GPU box):
    python3 tools/method_profile.py [--variant main|main_2] [--log-n 16] [--no-plan] [--eager]
Every controller method the driver calls (reference src/FHEController.h:73-139 names) is wrapped: the device is drained after the call
(`fhelin_sync` also evaluates the deferred heavy operations of the call), and wall time, limb-NTTs, key switches and rescales of the call are
added to the method's row.  Deferred ROWS are evaluated where their reader forces them, so a producer's rows appear under the method
that reads them unless --eager is given (FHELIN_LAZY_ROWS=0: every row where it is produced).  The drains cost the overlap between calls: the sum is a
few per cent above the pass of bench.py; the shares are what this is for.  One JSON line per method, largest first, then the total."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Timed:
    def __init__(self, inner, eng, rows):
        self._inner, self._eng, self._rows, self._depth = inner, eng, rows, 0

    def __getattr__(self, name):
        f = getattr(self._inner, name)
        if not callable(f) or name in ("level", "clone", "decrypt", "encode"):
            return f

        def call(*a, **kw):
            if self._depth:                              # a method calling another of the wrapped object: the outer row owns it
                return f(*a, **kw)
            self._depth += 1
            try:
                s0 = self._eng.stats()
                t0 = time.perf_counter()
                r = f(*a, **kw)
                self._eng.sync()
                dt = time.perf_counter() - t0
                s1 = self._eng.stats()
            finally:
                self._depth -= 1
            row = self._rows.setdefault(name, {"calls": 0, "ms": 0.0, "limb_ntt": 0, "keyswitch": 0, "rescale": 0, "bootstrap": 0})
            row["calls"] += 1
            row["ms"] += dt * 1e3
            for k in ("limb_ntt", "keyswitch", "rescale", "bootstrap"):
                row[k] += s1[k] - s0[k]
            return r
        return call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="main")
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--n-q", type=int, default=28)
    ap.add_argument("--tokens", type=int, default=129)
    ap.add_argument("--no-plan", action="store_true")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--passes", type=int, default=2)
    args = ap.parse_args()
    if args.eager:
        os.environ["FHELIN_LAZY_ROWS"] = "0"
    import torch  # noqa: F401  (the engine shares torch's HIP runtime)
    import fhe_linformer_amd as fa
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf             # synthetic weights / tokens only (tools/, not the product)

    eng = fa.Engine("bench", device=0, seed=7, log_n=args.log_n, n_q=args.n_q)
    eng.keygen()
    eng.gen_relin_key()
    eng.gen_rotation_keys(fa.circuit_rotation_indices())
    eng.bootstrap_setup(3, 3, 16384)
    w = pf.synthetic_model(1234)
    ctl = lf.GpuController(eng)
    n_client = 0
    if not args.no_plan:
        eng.level_plan_begin("record")
        enc = lf.encrypt_inputs(ctl, *pf.client_inputs(w, pf.synthetic_tokens(args.tokens, 999)))
        n_client = sum(len(v) for v in enc.values())
        eng.decrypt(lf.forward_encrypted(ctl, w, enc, None, args.variant))
        eng.level_plan_end()
        del enc
        eng.level_plan_begin("apply")
    enc = lf.ingest_sample(ctl, w, pf.synthetic_tokens(args.tokens, 4321))
    rows = {}
    for p in range(1 + args.passes):                     # pass 0 untimed: plaintext caches, the arena
        rows.clear() if p == 1 else None
        if not args.no_plan:
            eng.level_plan_begin("apply", first_source=n_client)
        t0 = time.perf_counter()
        out = lf.forward_encrypted(Timed(ctl, eng, rows), w, enc, None, args.variant)
        eng.decrypt(out)
        eng.sync()
        last = (time.perf_counter() - t0) * 1e3
    tot = {"calls": 0, "ms": 0.0, "limb_ntt": 0, "keyswitch": 0, "rescale": 0, "bootstrap": 0}
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        o = {"method": name}
        for k, v in r.items():
            o[k] = round(v / args.passes, 2) if k == "ms" else v // args.passes
            tot[k] += o[k]
        print(json.dumps(o))
    tot["ms"] = round(tot["ms"], 2)
    print(json.dumps({"method": "TOTAL of the rows", **tot, "last_pass_wall_ms_with_the_drains": round(last, 1), "variant": args.variant,
                      "log_n": args.log_n, "level_plan": not args.no_plan, "rows_eager": args.eager}))


if __name__ == "__main__":
    main()
