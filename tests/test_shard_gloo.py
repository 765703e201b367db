"""The N>1 path of bench.py on CPU: two gloo ranks shard a batch of independent samples, gather the per-sample
results and agree on the max step time (no GPU, no engine — only the distributed plumbing is exercised)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fhe_linformer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.sample_ids(total, world, rank)
    local = np.array([[i * 100.0 + c for c in range(20)] for i in ids]).reshape(len(ids), 20)   # stand-in logits
    parts = shard.gather_results(dist, local, max_rows=-(-total // world))
    tmax = shard.max_over_ranks(dist, 1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((ids, [p.tolist() for p in parts], tmax))
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather():
    world, total = 2, 5                                    # ragged: 3 + 2 samples
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    ids0, parts, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ids0 == [0, 1, 2]
    allrows = np.vstack([np.array(p) for p in parts])
    assert allrows.shape == (total, 20)
    assert np.array_equal(allrows[:, 0], np.arange(total) * 100.0)      # every sample exactly once, in order
    assert tmax == 2.0                                                  # max over ranks


def test_sample_partition_properties():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fhe_linformer_amd import shard
    for total in (0, 1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            got = [i for r in range(world) for i in shard.sample_ids(total, world, r)]
            assert got == list(range(total))
            sizes = [len(shard.sample_ids(total, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _row_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fhe_linformer_amd import shard, linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = pf.synthetic_model(1234)
    x_in, X_E, X_F = pf.client_inputs(w, pf.synthetic_tokens(129, 4321))
    calls = {}

    class Counting(cs.SlotSimController):                   # which heavy single-ciphertext calls THIS rank makes
        def bootstrap(self, c):
            calls["bootstrap"] = calls.get("bootstrap", 0) + 1
            return super().bootstrap(c)

        def eval_gelu_function(self, c, *a, **kw):
            calls["gelu"] = calls.get("gelu", 0) + 1
            return super().eval_gelu_function(c, *a, **kw)

    ctl = shard.RowShardedController(Counting(), dist, shard.SlotTransport())
    out = {}
    for variant in ("main", "main_2"):
        ctl.gathers = 0
        ctl.gather_rows = []
        calls.clear()
        out[variant] = (lf.logits_from_slots(lf.forward(ctl, w, x_in, X_E, X_F, None, variant)).tolist(), ctl.gathers, list(ctl.gather_rows), dict(calls))
    # a chain somebody still reads after a longer chain took its place is evaluated by itself; three independent chains are split 2 + 1
    plain = cs.SlotSimController()
    vals = [np.linspace(-0.5, 0.5, 64) * (k + 1) for k in range(3)]
    b = [ctl.bootstrap(v) for v in vals]
    g = [ctl.eval_gelu_function(x, -1, 1, 0.125, 119) for x in b]
    want_g = [plain.eval_gelu_function(plain.bootstrap(v), -1, 1, 0.125, 119) for v in vals]
    got_g = [np.asarray(ctl._res(x)) for x in g]
    got_b = [np.asarray(ctl._res(x)) for x in b]               # superseded handles, read afterwards
    chains_ok = all(np.array_equal(a, w_) for a, w_ in zip(got_g, want_g)) and all(np.array_equal(a, plain.bootstrap(v)) for a, v in zip(got_b, vals))
    out["chains"] = (bool(chains_ok), 3 in ctl.gather_rows)
    # ragged row counts through the gather itself: 5 rows over 2 ranks, 4-row groups of 3 tokens
    rows = {i: np.full(16, 10.0 * i) for i in shard.sample_ids(5, world, rank)}
    got = shard.all_gather_rows(dist, shard.SlotTransport(), rows, 5, world)
    blk = {4 * g + k: np.full(8, 100.0 * g + k) for g in shard.sample_ids(3, world, rank) for k in range(4)}
    gotb = shard.all_gather_rows_blocked(dist, shard.SlotTransport(), blk, 3, world, 4)
    dist.barrier()
    q.put((rank, out, [float(g[0]) for g in got], [float(g[0]) for g in gotb]))
    dist.destroy_process_group()


def test_rows_of_one_sample_shard_over_two_ranks():
    """the second sharding axis: the row loops inside matmul* / unwrap* of ONE sample are split over the
    ranks and re-assembled by an all-gather per call; with the clear-text controller the logits must equal the unsharded
    run exactly, on every rank, for both driver variants"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = pf.synthetic_model(1234)
    x_in, X_E, X_F = pf.client_inputs(w, pf.synthetic_tokens(129, 4321))
    for variant in ("main", "main_2"):
        ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, x_in, X_E, X_F, None, variant))
        for rank, out, _, _ in res:
            logits, gathers, gather_rows, calls = out[variant]
            assert np.array_equal(np.array(logits), ref), (variant, rank)
            assert gathers >= 7                     # K/V (Q in main_2), W_O, the unwraps, the FFN matmuls: every row loop was split
            # the rows of matmulRElarge are never gathered: generate_containers takes them unread and the five GROUPS of 32 rows are
            # split over the ranks (one gather of five containers)
            assert 5 in gather_rows, gather_rows
            # ... and they stay with their owners through the driver's GELU + bootstrap loop (src/main.cpp:354-358): 3 + 2 chains, gathered
            # once afterwards; the two bootstraps after affine-1 (:313-314) run one per rank (a gather of 2); the pooler's single
            # bootstrap is replicated
            assert 2 in gather_rows, gather_rows
            assert calls == ({"bootstrap": 1 + 3 + 1, "gelu": 3} if rank == 0 else {"bootstrap": 1 + 2 + 1, "gelu": 2}), (rank, calls)
    for _, out, rows, blk in res:
        assert out["chains"] == (True, True), out["chains"]
        assert rows == [0.0, 10.0, 20.0, 30.0, 40.0]
        assert blk == [100.0 * g + k for g in range(3) for k in range(4)]
