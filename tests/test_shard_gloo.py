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
