"""Multi-GPU plumbing of the benchmark / batch driver: the path shards over independent samples (ciphertexts),
one process per GPU, keys replicated, no data-path collective — only a final gather of the per-sample results
(20 logits each) and a max-reduction of the step time.  Works with the `nccl` backend (RCCL over xGMI, CUDA
tensors) and with `gloo` (CPU tensors; used by the world_size-2 test)."""
import numpy as np


def sample_ids(total, world, rank):
    """contiguous, balanced partition of `total` independent samples over `world` ranks"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_results(dist, local, max_rows):
    """all-gather per-rank result rows [n_local, width] (n_local <= max_rows) -> list of arrays ordered by rank"""
    import torch
    local = np.asarray(local, dtype=np.float64).reshape(len(local), -1)
    width = local.shape[1] if local.size else 0
    dev = _device(dist)
    meta = torch.tensor([local.shape[0], width], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(dist.get_world_size())]
    dist.all_gather(metas, meta)
    width = max(int(m[1]) for m in metas)
    pad = np.zeros((max_rows, width))
    pad[: local.shape[0], : local.shape[1]] = local
    t = torch.tensor(pad, dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy()[: int(m[0])] for o, m in zip(out, metas)]


def max_over_ranks(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
