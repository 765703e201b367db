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


def _payload_ready(dev):
    """the engine imports gathered residues on ITS OWN HIP stream: the collective's output must be complete on torch's
    stream first (fhelin_ct_import_device / _export_device are host-synchronous on the library stream only)"""
    if dev.type == "cuda":
        import torch
        torch.cuda.current_stream().synchronize()


def max_over_ranks(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- second sharding axis: the rows inside one sample's matmul / unwrap loops (batch-1 latency on several GPUs) ---------
class SlotTransport:
    """ciphertexts are plain slot vectors (oracle.circuit_sim.SlotSimController): what the gloo tests exchange"""

    def pack(self, ct):
        a = np.asarray(ct, dtype=np.float64).reshape(-1)
        return a.view(np.uint8), np.zeros(8)

    def unpack(self, payload, meta):
        return payload.view(np.float64).copy()


class EngineTransport:
    """ciphertexts of fhe_linformer_amd.Engine: residues + (npoly, ell, deg, slots, scale hi/lo).  With the nccl backend the
    residues go GPU -> GPU through device buffers (fhelin_ct_export_device / fhelin_ct_import_device, RCCL over xGMI);
    with gloo (rehearsal, tests) through host arrays."""

    def __init__(self, eng, device=False):
        self.eng, self.device = eng, device

    def force(self, cts):
        """the rows this rank owns are evaluated in ONE batched call, and only they (deferred rows of the engine)"""
        self.eng.force(cts)

    def pack(self, ct):
        inf = ct.info()
        hi, lo = ct.scale_parts()
        meta = np.array([inf["npoly"], inf["ell"], inf["deg"], inf["slots"], hi, lo, 0, 0], dtype=np.float64)
        if self.device:
            import torch
            words = inf["npoly"] * inf["ell"] * self.eng.N
            t = torch.empty(words * 8, dtype=torch.uint8, device="cuda")
            ct.export_device(t.data_ptr(), words)
            return t, meta
        return ct.export().reshape(-1).view(np.uint8), meta

    def unpack(self, payload, meta):
        npoly, ell, deg, slots = (int(meta[i]) for i in range(4))
        words = npoly * ell * self.eng.N
        if self.device:
            return self.eng.ct_import_device(payload.data_ptr(), npoly, ell, deg, float(meta[4]), float(meta[5]), slots)
        limbs = np.ascontiguousarray(payload[: words * 8]).view(np.uint64).reshape(npoly, ell, self.eng.N)
        return self.eng.ct_import(limbs, deg=deg, scale=float(meta[4]), slots=slots)


def _all_gather_indexed(dist, transport, local, owned):
    """owned[r] = the indices rank r holds (local[i] on that rank) -> {index: ciphertext} of all of them on every rank (the owner
    re-imports its own too, so that every replica continues from identical bytes).  One size all-reduce, one metadata all-gather and
    one payload all-gather (RCCL over xGMI with nccl)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = owned[rank]
    per = max(1, max(len(o) for o in owned))
    if hasattr(transport, "force"):
        transport.force([local[i] for i in mine])
    packed = [transport.pack(local[i]) for i in mine]
    dev = _device(dist)
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    for p, _ in packed:
        size[0] = max(int(size[0]), int(p.numel() if hasattr(p, "numel") else p.size))
    dist.all_reduce(size, op=dist.ReduceOp.MAX)
    meta = torch.zeros((per, 8), dtype=torch.float64, device=dev)
    buf = torch.zeros((per, int(size[0])), dtype=torch.uint8, device=dev)
    for k, (p, m) in enumerate(packed):
        meta[k] = torch.as_tensor(m, dtype=torch.float64)
        t = p if hasattr(p, "numel") else torch.from_numpy(np.ascontiguousarray(p))
        buf[k, : t.numel()] = t.to(dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(metas, meta)
    dist.all_gather(bufs, buf)
    _payload_ready(dev)
    out = {}
    for r in range(world):
        for k, i in enumerate(owned[r]):
            payload = bufs[r][k] if getattr(transport, "device", False) else bufs[r][k].cpu().numpy()
            out[i] = transport.unpack(payload, metas[r][k].cpu().numpy())
    return out


def all_gather_rows(dist, transport, local, n_rows, world):
    """local: {row index: ciphertext} of this rank's rows (sample_ids partition) -> list of all n_rows ciphertexts, the
    same objects on every rank"""
    got = _all_gather_indexed(dist, transport, local, [sample_ids(n_rows, world, r) for r in range(world)])
    return [got[i] for i in range(n_rows)]


def all_gather_owned(dist, transport, local, owners):
    """owners[i] = the rank that holds value i (local[i] there) -> list of all values on every rank"""
    world = dist.get_world_size()
    got = _all_gather_indexed(dist, transport, local, [[i for i, o in enumerate(owners) if o == r] for r in range(world)])
    return [got[i] for i in range(len(owners))]


class _ShardedRows(list):
    """Rows of a row loop whose evaluation is deferred by the engine (plaintext-weight matmulRE / matmulCR, unwrapExpanded): the
    sharded evaluation + all-gather happens only when the driver turns out to read the rows.  The reference's CLS-only driver
    computes all S query projections and all S final token expansions and reads ONE of each (src/main.cpp:183,:196,:416-424):
    a single row read is evaluated locally (every rank runs the same driver, so every rank evaluates that one row), nothing is
    split, nothing is gathered.  The second distinct access — another index, a slice, an iteration, a hand-over to another
    call — means the driver walks the rows: each rank then evaluates the rows it owns (one batched call) and ONE all-gather puts
    all of them on every rank."""

    def __init__(self, ctl, rows):
        super().__init__(rows)
        self._ctl, self._single, self._done = ctl, None, False

    def _materialize(self):
        if not self._done:
            self._done = True
            n = super().__len__()
            ids = self._ctl._mine(n)
            got = self._ctl._gather({i: list.__getitem__(self, i) for i in ids}, n)
            for i in range(n):
                list.__setitem__(self, i, got[i])

    def __getitem__(self, i):
        if not self._done:
            if isinstance(i, int) and self._single in (None, i % super().__len__()):
                self._single = i % super().__len__()
                return list.__getitem__(self, i)
            self._materialize()
        return list.__getitem__(self, i)

    def __iter__(self):
        self._materialize()
        return list.__iter__(self)

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


class _PendingRelarge(list):
    """Rows of matmulRElarge that nobody has read yet.  The drivers hand all of them to generate_containers next
    (src/main.cpp:341-352), where the engine evaluates a whole group of 32 rows as ONE shift sum (Composite::relarge_containers): the
    sharded controller then splits the GROUPS over the ranks and gathers five containers instead of 130 rows.  Any other access
    evaluates the rows the usual sharded way (each rank its own rows, one all-gather)."""

    def __init__(self, ctl, parts):
        super().__init__([None] * sum(len(p[0]) for p in parts))
        self._ctl, self.parts, self.rows_done = ctl, parts, False

    def _materialize(self):
        if not self.rows_done:
            self.rows_done = True
            out = []
            for rows, weights, bias, mask_val in self.parts:
                out += list(self._ctl._relarge_rows(rows, weights, bias, mask_val))
            for i, v in enumerate(out):
                list.__setitem__(self, i, v)

    def __getitem__(self, i):
        self._materialize()
        return list.__getitem__(self, i)

    def __iter__(self):
        self._materialize()
        return list.__iter__(self)

    def __add__(self, other):
        if isinstance(other, _PendingRelarge) and not self.rows_done and not other.rows_done:
            return _PendingRelarge(self._ctl, self.parts + other.parts)
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


class _Chain:
    """A ciphertext that is the result of single-ciphertext heavy calls (bootstrap, eval_gelu_function) the sharded controller has not
    made yet.  The reference's drivers run such calls in loops over independent ciphertexts and read the results afterwards
    (src/main.cpp:313-314: the two halves of affine-1; :354-358: GELU then bootstrap per container): the controller collects the
    calls, and when another call reads one of the results every chain is evaluated by ONE rank - the rank that holds its input
    (a container of generate_containers' split) or, for replicated inputs, the least loaded one - and one all-gather puts the
    results on every rank.  `ops`: (method, args, kwargs, level-plan source index or None)."""
    __slots__ = ("src", "ops", "owner", "value")

    def __init__(self, src, ops, owner):
        self.src, self.ops, self.owner, self.value = src, ops, owner, None


def _resolved(fn):
    def call(self, *a, **kw):
        return fn(self, *[self._res(x) for x in a], **{k: self._res(v) for k, v in kw.items()})
    call.__name__, call.__doc__ = fn.__name__, fn.__doc__
    return call


class RowShardedController:
    """Batch-1 latency on `world` GPUs: every rank holds the same keys and runs the same driver; the row loops of the
    reference's matmul* / unwrap* methods (src/FHEController.cpp:872,888,904,918,949,963,985,1001,1089,1115) are split
    over the ranks with `sample_ids`, each rank evaluates only its rows, and one all-gather per call puts all rows back on
    every rank (the driver's next call — a wrapUp*, a bootstrap, the next matmul — is replicated).  With the engine's
    deferred rows a rank evaluates exactly the rows it owns (fhelin_ct_force: one batched call) before it packs them; calls
    whose rows are deferred return _ShardedRows: nothing is split or gathered until the driver reads more than one row.
    Calls with fewer rows than `min_rows` stay replicated."""

    def __init__(self, inner, dist, transport, min_rows=8, counter=None):
        self.c, self.dist, self.t = inner, dist, transport
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.min_rows = min_rows
        self.gathers = 0
        self.gather_rows = []      # rows per gather, in call order
        # counter() -> a running operation count (e.g. the engine's key switches): row_ops accumulates what this rank spent
        # evaluating the rows it owns, so that a test can check the split against an unsharded pass
        self.counter, self.row_ops = counter, 0
        self._pending = []         # _Chain values nobody has read yet, in call order
        self.shard_chains = True   # False: bootstrap / eval_gelu_function replicated on every rank (rounds 2-3)
        eng = getattr(transport, "eng", None)
        self._plan_eng = eng if hasattr(eng, "level_plan_tell") else None

    def _counted(self, fn):
        if self.counter is None:
            return fn()
        before = self.counter()
        r = fn()
        self.row_ops += self.counter() - before
        return r

    def __getattr__(self, name):            # everything that is not a row loop: replicated, as the inner controller does it
        f = getattr(self.c, name)
        if not callable(f):
            return f

        def call(*a, **kw):                 # ... on the VALUES of pending chains
            return f(*[self._res(x) for x in a], **{k: self._res(v) for k, v in kw.items()})
        return call

    # ---- single-ciphertext heavy calls in loops over independent ciphertexts: one rank each -------------------------------------
    def _plan_pos(self):
        return self._plan_eng.level_plan_tell() if self._plan_eng is not None else ("off", 0)

    def _chain(self, name, x, a, kw, source):
        mode, pos = self._plan_pos()
        if self.world == 1 or not self.shard_chains or mode == "record":   # a recording pass runs the whole program on every rank
            return getattr(self.c, name)(self._res(x), *a, **kw)
        ordinal = None
        if source and mode == "apply":       # the call is a source of the level plan: it keeps its place in the program's order
            ordinal = pos
            self._plan_eng.level_plan_seek(pos + 1)
        op = (name, a, kw, ordinal)
        if isinstance(x, _Chain) and x.value is None:
            if x in self._pending:
                self._pending.remove(x)      # superseded by the longer chain (evaluated by itself if somebody still reads it)
            ch = _Chain(x.src, x.ops + [op], x.owner)
        else:
            ch = _Chain(x.value if isinstance(x, _Chain) else x, [op], None)
        self._pending.append(ch)
        return ch

    def bootstrap(self, x):
        return self._chain("bootstrap", x, (), {}, True)

    def eval_gelu_function(self, x, *a, **kw):
        return self._chain("eval_gelu_function", x, a, kw, False)

    def _res(self, x):
        """x with every pending chain in it (lists / tuples searched) replaced by its value; x itself when there is none"""
        if isinstance(x, _Chain):
            if x.value is None:
                if x not in self._pending:
                    self._pending.append(x)
                self._flush()
            return x.value
        if type(x) in (list, tuple):
            if not any(isinstance(e, _Chain) or type(e) in (list, tuple) for e in x):
                return x
            y = [self._res(e) for e in x]
            if all(p is q for p, q in zip(x, y)):
                return x
            return y if type(x) is list else tuple(y)
        return x

    def _flush(self):
        chains, self._pending = self._pending, []
        if not chains:
            return
        if len(chains) == 1 and chains[0].owner is None:        # nothing to split: replicated, no gather
            ch = chains[0]
            _, pos = self._plan_pos()
            ch.value = self._run_chains([ch])[0]
            if self._plan_eng is not None:
                self._plan_eng.level_plan_seek(pos)
            return
        load = [sum(1 for c in chains if c.owner == r) for r in range(self.world)]
        for c in chains:                                        # the same assignment on every rank
            if c.owner is None:
                c.owner = load.index(min(load))
                load[c.owner] += 1
        mine = [c for c in chains if c.owner == self.rank]
        _, pos = self._plan_pos()
        vals = self._counted(lambda: self._run_chains(mine))
        if self._plan_eng is not None:
            self._plan_eng.level_plan_seek(pos)
        local = {k: vals[mine.index(c)] for k, c in enumerate(chains) if c.owner == self.rank}
        self.gathers += 1
        self.gather_rows.append(len(chains))
        got = all_gather_owned(self.dist, self.t, local, [c.owner for c in chains])
        for c, v in zip(chains, got):
            c.value = v

    def _run_chains(self, chains):
        """step by step over all chains, so that the engine's own deferral sees the calls of one kind next to each other and
        evaluates them as ONE batched call (three Chebyshev evaluations, then three bootstraps)"""
        vals = [c.src for c in chains]
        for s in range(max((len(c.ops) for c in chains), default=0)):
            for k, c in enumerate(chains):
                if s < len(c.ops):
                    name, a, kw, ordinal = c.ops[s]
                    if ordinal is not None:
                        self._plan_eng.level_plan_seek(ordinal)
                    vals[k] = getattr(self.c, name)(vals[k], *a, **kw)
        return vals

    def _mine(self, n):
        return sample_ids(n, self.world, self.rank)

    def _gather(self, local, n):
        self.gathers += 1
        self.gather_rows.append(n)
        if hasattr(self.t, "force"):
            self._counted(lambda: self.t.force(list(local.values())))
        return all_gather_rows(self.dist, self.t, local, n, self.world)

    def _rows(self, n, all_rows_fn, subset_fn=None):
        """all_rows_fn() -> n row handles of which only the owned ones are read (deferred rows), or subset_fn(ids) -> the
        owned rows only"""
        if self.world == 1 or n < self.min_rows:
            return all_rows_fn()
        ids = self._mine(n)
        if subset_fn is not None:
            got = self._counted(lambda: subset_fn(ids))
            return self._gather(dict(zip(ids, got)), n)
        rows = all_rows_fn()
        if hasattr(self.t, "force"):            # the engine defers these rows: split and gather only if the driver reads them
            return _ShardedRows(self, rows)
        return self._gather({i: rows[i] for i in ids}, n)

    @_resolved
    def matmulRE(self, rows, w, bias=None, row_size=128, padding=128):
        if not _is_ct(w):       # plaintext weight: deferred rows, a rank reads (= evaluates) only the rows it owns
            return self._rows(len(rows), lambda: self.c.matmulRE(rows, w, bias, row_size, padding))
        return self._rows(len(rows), lambda: self.c.matmulRE(rows, w, bias, row_size, padding),
                          lambda ids: self.c.matmulRE([rows[i] for i in ids], w, bias, row_size, padding))

    @_resolved
    def matmulCR(self, rows, w, bias=None):
        return self._rows(len(rows), lambda: self.c.matmulCR(rows, w, bias))

    @_resolved
    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        if self.world == 1:
            return self.c.matmulRElarge(rows, weights, bias, mask_val)
        return _PendingRelarge(self, [(rows, weights, bias, mask_val)])

    def _relarge_rows(self, rows, weights, bias, mask_val):
        return self._rows(len(rows), lambda: self.c.matmulRElarge(rows, weights, bias, mask_val),
                          lambda ids: self.c.matmulRElarge([rows[i] for i in ids], weights, bias, mask_val))

    @_resolved
    def generate_containers(self, inputs, bias=None):
        """containers of unread matmulRElarge rows: the groups of 32 rows are split over the ranks, every rank runs the engine's two
        calls on its groups (fused there) and ONE all-gather puts the containers on every rank - the same residues as the unsharded
        call (a group's container does not depend on the other groups)"""
        p = inputs.parts if isinstance(inputs, _PendingRelarge) and not inputs.rows_done else None
        if p and all(q[1] is p[0][1] and q[2] is p[0][2] and q[3] == p[0][3] for q in p):
            rows = [r for q in p for r in q[0]]
            _, weights, rbias, mask_val = p[0]
            n_groups = (len(rows) + 31) // 32
            if n_groups >= 2:
                local = {}

                def run():
                    for g in self._mine(n_groups):
                        outs = self.c.matmulRElarge(rows[32 * g:32 * g + 32], weights, rbias, mask_val)
                        local[g] = self.c.generate_containers(outs, bias)[0]
                self._counted(run)
                if self.shard_chains:
                    # the containers stay where they were computed: the driver's next calls on them (GELU, bootstrap) run on the
                    # owner, and the gather moves the refreshed ciphertexts once (the first other reader flushes)
                    owners = [r for r in range(self.world) for _ in sample_ids(n_groups, self.world, r)]
                    chains = [_Chain(local.get(g), [], owners[g]) for g in range(n_groups)]
                    self._pending += chains
                    return chains
                return self._gather(local, n_groups)
        return self.c.generate_containers(list(inputs), bias)

    @_resolved
    def matmulCRlarge(self, rows, weights, bias):
        return self._rows(len(rows), lambda: self.c.matmulCRlarge(rows, weights, bias),
                          lambda ids: self.c.matmulCRlarge([rows[i] for i in ids], weights, bias))

    @_resolved
    def unwrapExpanded(self, c, n):
        return self._rows(n, lambda: self.c.unwrapExpanded(c, n))

    @_resolved
    def unwrapRepeatedLarge(self, cs, n):
        if self.world == 1 or n < self.min_rows:
            return self.c.unwrapRepeatedLarge(cs, n)
        ids = self._mine(n)
        if hasattr(self.c, "unwrapRepeatedLarge_range"):
            mine = self._counted(lambda: self.c.unwrapRepeatedLarge_range(cs, n, ids[0], len(ids))) if ids else []
        else:
            allr = self.c.unwrapRepeatedLarge(cs, n)
            mine = [allr[i] for i in ids]
        flat = {4 * i + k: mine[j][k] for j, i in enumerate(ids) for k in range(4)}
        # 4 ciphertexts per token: gather them as 4n rows owned in blocks of 4
        out = all_gather_rows_blocked(self.dist, self.t, flat, n, self.world, 4)
        self.gathers += 1
        self.gather_rows.append(4 * n)
        return [out[4 * i: 4 * i + 4] for i in range(n)]


def _is_ct(x):
    return x.__class__.__name__ == "Ct"


def all_gather_rows_blocked(dist, transport, local, n_groups, world, group):
    """like all_gather_rows for n_groups groups of `group` consecutive rows, ownership by group"""
    owned = [[g * group + k for g in sample_ids(n_groups, world, r) for k in range(group)] for r in range(world)]
    got = _all_gather_indexed(dist, transport, local, owned)
    return [got[i] for i in range(n_groups * group)]
