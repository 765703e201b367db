"""The reference's FHEController surface on RESIDUES, on the CPU.  TEST INFRASTRUCTURE (part of oracle/).

`ResidueController` runs the driver fhe-linformer_amd/linformer.py (the call sequence of reference src/main.cpp:145-475 /
src/main_2.cpp) unchanged, every method composed from oracle/fhe_oracle.c through oracle/residue_eval.py and
oracle/residue_boot.py in the order fhe-linformer_amd/csrc/composite.cpp runs it BY DEFAULT (products rescaled before their trees,
tree steps merged in triples / pairs, log-depth shift trees, the re-associated matmulRElarge with its double-hoisted first step, the
two-stage unwrapRepeatedLarge).
The complete forward pass of the GPU library must end in the same residues, bit for bit (tests/test_forward_residue_gpu.py).

What enters from the library, as for every other residue test: the switching keys (exported), the plaintext encodings
(fhelin_pt_export of the same slot values at the (limbs, scale) an operation asks for) and the FRESH ENCRYPTIONS of the run that
is being checked (encryption is randomised: the GPU run records the ciphertexts it made, the oracle replays them in call order).
Reference methods restated here beyond residue_eval.py (all src/FHEController.cpp): matmulRE with a ciphertext weight :901-913,
matmulCRlarge :998-1026, matmulScores :1028-1048, wrapUpRepeated :1060-1068, generate_containers :1164-1191, eval_exp :1289-1311,
the mask_* constants :1207-1286."""
import math

import numpy as np

import oracle as orc
from oracle.residue_eval import LD, RCt

SLOTS = 16384


class GaloisKeys:
    """switching keys by Galois element, addressed by rotation index like the library's table (indices that differ by the order
    of 5 share a key); "relin" and "conj" pass through"""

    def __init__(self, log_n):
        self.log_n, self.k = log_n, {}

    def _g(self, r):
        return r if isinstance(r, str) else orc.galois(self.log_n, int(r))

    def __setitem__(self, r, v):
        self.k[self._g(r)] = v

    def __getitem__(self, r):
        return self.k[self._g(r)]

    def __contains__(self, r):
        return self._g(r) in self.k


class RPt:
    """a plaintext operand: its slot values + the library handle that yields its encodings"""

    def __init__(self, eng, values, level):
        self.values = np.asarray(values, dtype=np.float64)
        self.level = level
        self.pt = eng.encode(self.values, level, SLOTS)
        self.eng, self._cache = eng, {}

    def enc(self, ell, scale):
        k = (ell, float(scale), float(LD(scale) - LD(float(scale))))
        if k not in self._cache:
            if len(self._cache) > 4:
                self._cache.clear()
            self._cache[k] = self.eng.pt_export(self.pt, ell, scale)
        return self._cache[k]


class _LazyRows(list):
    """rows evaluated when read (the library defers the same calls: rows nobody reads cost nothing on either side).  bulk: how MANY
    rows read together are evaluated (the library takes another form of the same rows when >= bulk_min of them are forced at once:
    Composite::unwrapExpanded_bulk)"""

    def __init__(self, n, fn, bulk=None, bulk_min=64):
        super().__init__([None] * n)
        self._fn, self._bulk, self._bulk_min = fn, bulk, bulk_min

    def _get(self, i):
        v = list.__getitem__(self, i)
        if v is None:
            v = self._fn(i)
            list.__setitem__(self, i, v)
        return v

    def _get_many(self, ids):
        todo = [i for i in ids if list.__getitem__(self, i) is None]
        if self._bulk is not None and len(todo) >= self._bulk_min:
            for i, v in zip(todo, self._bulk(todo)):
                list.__setitem__(self, i, v)
        return [self._get(i) for i in ids]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._get_many(list(range(*i.indices(len(self)))))
        return self._get(i if i >= 0 else i + len(self))

    def __iter__(self):
        return iter(self._get_many(list(range(len(self)))))

    def __add__(self, other):
        if isinstance(other, (_LazyRows, _LazyConcat)):
            return _LazyConcat([self] + (other.parts if isinstance(other, _LazyConcat) else [other]))   # still nobody has read a row
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


class _LazyConcat(list):
    """rows of several deferred calls in a row, unread (linformer.py: outputs_raw = output_0 + output_1)"""

    def __init__(self, parts):
        super().__init__()
        self.parts = parts
        if all(getattr(p, "relarge", None) is not None for p in parts):
            self.relarge_lists = parts
        for p in parts:
            list.extend(self, [None] * len(p))

    def _locate(self, i):
        for p in self.parts:
            if i < len(p):
                return p, i
            i -= len(p)
        raise IndexError(i)

    def _get_many(self, ids):
        by_part = {}
        for i in ids:
            p, k = self._locate(i)
            by_part.setdefault(id(p), (p, []))[1].append(k)
        for p, ks in by_part.values():        # rows of one call that are read together are evaluated together
            p._get_many(ks)
        out = []
        for i in ids:
            p, k = self._locate(i)
            out.append(p._get(k))
        return out

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._get_many(list(range(*i.indices(len(self)))))
        return self._get_many([i if i >= 0 else i + len(self)])[0]

    def __iter__(self):
        return iter(self._get_many(list(range(len(self)))))

    def __add__(self, other):
        if isinstance(other, _LazyRows):
            return _LazyConcat(self.parts + [other])
        if isinstance(other, _LazyConcat):
            return _LazyConcat(self.parts + other.parts)
        return list(self) + list(other)


class ResidueController:
    num_slots = SLOTS

    def __init__(self, eng, rev, boot, fresh, drops=None):
        """rev: ResidueEvaluator over all keys of the run; boot: ResidueBootstrapper; fresh: the run's fresh encryptions (RCt) in call
        order; drops: limbs each bootstrap of the run left out (level plan), in call order"""
        self.eng, self.rev, self.boot = eng, rev, boot
        self.fresh, self.drops = list(fresh), list(drops or [])
        self.n_boot = 0
        self._masks = {}

    # ---- handles, plaintexts, fresh encryptions
    def level(self, c):
        return len(self.rev.q) - c.ell

    def clone(self, c):
        return c

    def encode(self, v, level=0):
        if np.isscalar(v):
            v = np.full(SLOTS, float(v))
        out = np.zeros(SLOTS)
        v = np.asarray(v, dtype=np.float64).reshape(-1)
        out[: len(v)] = v[:SLOTS]
        return RPt(self.eng, out, level)

    def encrypt(self, v, level=0):
        return self.fresh.pop(0)

    def read_expanded_inputs(self, rows, scale=1.0):
        out, self.fresh = self.fresh[:len(rows)], self.fresh[len(rows):]
        return out

    def read_expanded_input(self, v, scale=1.0):
        return self.fresh.pop(0)

    def read_plain_input(self, m, level=0, scale=1.0):
        return self.encode(np.asarray(m, dtype=np.float64).reshape(-1) * scale, level)

    def read_plain_repeated_input(self, v, level=0, scale=1.0):
        return self.encode(np.tile(np.asarray(v, dtype=np.float64)[:128], 128) * scale, level)

    def read_plain_expanded_input(self, v, level=0, scale=1.0):
        out = np.zeros(SLOTS)
        vv = np.asarray(v, dtype=np.float64) * scale
        for j in range(128):
            out[j * 128: j * 128 + 128] = vv[j]
        return self.encode(out, level)

    def _mask(self, values):
        """composite.cpp's cached mask plaintexts (encoded at level 0, re-encoded per use)"""
        key = values.tobytes()
        if key not in self._masks:
            self._masks[key] = RPt(self.eng, values, 0)
        return self._masks[key]

    def _block_mask(self, frm, to, val):
        m = np.zeros(SLOTS)
        m[max(frm, 0):min(to, SLOTS)] = val
        return self._mask(m)

    def _mod_mask(self, n, padding, val=1.0):
        m = np.zeros(SLOTS)
        m[padding::n] = val
        return self._mask(m)

    # ---- leaf operations
    def add(self, a, b):
        if isinstance(b, RPt):
            return self.rev.add_plain(a, b.enc)
        return self.rev.add(a, b)

    def mult(self, a, b):
        if np.isscalar(b):
            return self.rev.mult_plain(a, self._mask(np.full(SLOTS, float(b))).enc)       # Composite::mult_const
        if isinstance(b, RPt):
            return self.rev.mult_plain(a, b.enc)
        return self.rev.mult(a, b)

    def rotate(self, a, i):
        return self.rev.rotate(a, i)

    def bootstrap(self, a):
        drop = self.drops[self.n_boot] if self.n_boot < len(self.drops) else 0
        self.n_boot += 1
        return self.boot.run(a, drop=drop)

    # ---- composites
    def rotsum(self, a, slots, padding):
        return self.rev.rotsum(a, slots, padding)

    def _matmul_pt(self, rows, w, bias, slots, padding):
        rev = self.rev
        rows = list(rows)          # the library reads its input rows at the call (deferred input rows are evaluated together there)
        return _LazyRows(len(rows), lambda i: rev.matmul_pt([rows[i]], w.enc, bias.enc if bias is not None else None, slots, padding)[0])

    def matmulRE(self, rows, w, bias=None, row_size=128, padding=128):
        if isinstance(w, RPt):
            return self._matmul_pt(rows, w, bias, row_size, padding)
        assert bias is None
        return [self.rev.rotsum(self.rev.mult(r, w), row_size, padding) for r in rows]      # Composite::matmul_ct (:901-913)

    def matmulCR(self, rows, w, bias=None):
        return self._matmul_pt(rows, w, bias, 128, 1)

    def _relarge_v(self, weights):
        if getattr(self, "_relarge_for", None) is not weights[0]:       # a call's four V_t, kept while the same weights come back
            w2 = []
            for t in range(4):                                 # Composite::relarge_weights: block b of W''_t = block b of W_((b - t) mod 4),
                v = np.zeros(SLOTS)                            # V_t = rot(W''_t, 128 t)
                for b in range(128):
                    v[128 * b:128 * (b + 1)] = weights[(b - t) % 4].values[128 * b:128 * (b + 1)]
                w2.append(RPt(self.eng, np.roll(v, -128 * t), weights[0].level).enc)
            self._relarge_for, self._relarge_vt = weights[0], w2
        return self._relarge_vt

    def matmulRElarge(self, rows, weights, bias, mask_val=1.0):
        """rows that nobody has read when generate_containers takes them are never evaluated on their own (the C ABI defers them the
        same way: csrc/capi_internal.h); a row that is read is Composite::matmulRElarge"""
        v = self._relarge_v(weights)
        m512 = self._block_mask(0, 512, mask_val).enc
        rows = list(rows)
        rev, benc = self.rev, (bias.enc if bias is not None else None)
        out = _LazyRows(len(rows), lambda i: rev.matmulRElarge([rows[i]], v, benc, m512)[0])
        out.relarge = (rows, weights, bias, mask_val)
        return out

    def matmulCRlarge(self, rows, weights, bias):
        """:998-1026: per row sum_j r[j] * W_j as (p0 + p1) + (p2 + p3), one rotsum(128, 1), + bias"""
        rev, out = self.rev, []
        for r in rows:
            p = [rev.mult_plain(r[j], weights[j].enc) for j in range(4)]
            s = rev.add(rev.add(p[0], p[1]), rev.add(p[2], p[3]))
            o = rev.rotsum(s, 128, 1)
            out.append(rev.add_plain(o, bias.enc) if bias is not None else o)
        return out

    def matmulScores(self, queries, key):
        """:1028-1048: scores_i = rotsum(q_i * K, 128, 1), mask the heads with 1/64, sum_i rot(masked_i, -i)"""
        rev = self.rev
        queries = queries if isinstance(queries, list) else [queries]
        mask = self._mod_mask(128, 0, (1 / 8.0) * (1 / 8.0)).enc
        masked = [rev.mult_plain(rev.rotsum(rev.mult(q, key), 128, 1), mask) for q in queries]
        return rev.shift_sum(masked, -1)

    def wrapUpRepeated(self, v):
        """:1060-1068: sum_i v_i * (block i mask); exact sums of products (Evaluator::dot_plain)"""
        acc = None
        for i, c in enumerate(v):
            t = self.rev.mult_plain(c, self._block_mask(128 * i, 128 * (i + 1), 1.0).enc)
            acc = t if acc is None else self.rev.add(acc, t)
        return acc

    def wrapUpExpanded(self, v):
        return self.rev.wrapUpExpanded(list(v), self._mod_mask(128, 0).enc)

    def unwrapExpanded(self, c, n):
        """:1086-1100 with the three-level fan of Composite::shift_fan_rows (ResidueEvaluator.fan_row); rows evaluated when read -
        64 or more of them read together as sliding-window sums of the two fans (ResidueEvaluator.unwrapExpanded_bulk)"""
        rev, mask, memo = self.rev, self._mod_mask(128, 0).enc, {}
        bulk = lambda ids: rev.unwrapExpanded_bulk(c, n, ids, [self._mod_mask(128, k).enc for k in range(128)])
        return _LazyRows(n, lambda i: rev.repeat(rev.mult_plain(rev.fan_row(c, i, 1, memo, n - 1), mask), 128, 1), bulk if n <= 128 else None)

    def unwrapRepeatedLarge(self, cs, n):
        return self.rev.unwrapRepeatedLarge(list(cs), n, lambda v: self._mask(v).enc)

    def generate_containers(self, inputs, bias=None):
        """:1164-1191: groups of 32 inputs, sum_i rot(group[i], -512 i) each.  Inputs that are unread rows of matmulRElarge (one set of
        weights) take Composite::relarge_containers: per group of >= 8 rows the trees of the rows and the container sum are one
        shift sum (ResidueEvaluator.relarge_container)"""
        lists = getattr(inputs, "relarge_lists", None)
        if lists is None and getattr(inputs, "relarge", None) is not None:
            lists = [inputs]
        parts = [lz.relarge for lz in lists] if lists else None
        same = parts is not None and all(p[1] is parts[0][1] and p[2] is parts[0][2] and p[3] == parts[0][3] for p in parts)
        if same and all(list.__getitem__(lz, i) is None for lz in lists for i in range(len(lz))) and sum(len(lz) for lz in lists) > 0:
            rows = [r for p in parts for r in p[0]]
            _, weights, rbias, mask_val = parts[0]
            v = self._relarge_v(weights)
            us = [self.rev.relarge_u(r, v) for r in rows]
            masks = [self._block_mask(512 * j, 512 * (j + 1), mask_val).enc for j in range(32)]
            m512 = self._block_mask(0, 512, mask_val).enc
            out = []
            for lo in range(0, len(us), 32):
                ug = us[lo:lo + 32]
                q = len(ug)
                if q >= 8:
                    tiled = None
                    if rbias is not None:
                        tv = np.zeros(SLOTS)
                        for i in range(q):
                            tv += np.roll(rbias.values, 512 * i)
                        tiled = RPt(self.eng, tv, rbias.level).enc
                    part = self.rev.relarge_container(ug, masks, tiled)
                else:
                    tail = []
                    for u in ug:
                        o = self.rev.mult_plain(self.rev.rotsum(u, 32, 512), m512)
                        tail.append(self.rev.add_plain(o, rbias.enc) if rbias is not None else o)
                    part = self.rev.wrap_containers(list(reversed(tail)), q)
                out.append(self.rev.add_plain(part, bias.enc) if bias is not None else part)
            return out
        inputs = list(inputs)
        total, out = len(inputs), []
        i = 0
        while i < total / 32.0:
            q = 32 if (i + 1) * 32 <= total else total - i * 32
            chunk = inputs if total <= 32 else inputs[i * 32:min((i + 1) * 32, total)]
            part = self.rev.wrap_containers(list(reversed(chunk)), q)
            out.append(self.rev.add_plain(part, bias.enc) if bias is not None else part)
            i += 1
        return out

    # ---- activations (:1289-1336), with the coefficient fit of the driver (linformer.cheb_coeffs)
    def eval_exp(self, c, inputs_number):
        res = self.rev.eval_poly(c, [1, 1, 1 / 2.0, 1 / 6.0, 1 / 24.0, 1 / 120.0, 1 / 720.0])
        res = self.rev.mult_many([res] * 8)
        i = np.arange(SLOTS)
        mask = np.where((i % 128 < inputs_number) & (i < 128 * inputs_number), 0.0, -1.0)
        return self.add(res, self.encode(mask, self.level(res)))

    def _cheb(self, f, c, a, b, degree):
        from fhe_linformer_amd.linformer import cheb_coeffs
        return self.rev.eval_chebyshev(c, [float(v) for v in cheb_coeffs(f, a, b, degree)], float(a), float(b))

    def eval_inverse_naive(self, c, lo, hi):
        return self._cheb(lambda x: 1.0 / x, c, lo, hi, 119)

    def eval_gelu_function(self, c, lo, hi, mult, degree):
        return self._cheb(lambda x: 0.5 * (x / mult) * (1 + math.erf((x / mult) / 1.41421356237)), c, lo, hi, degree)

    def eval_tanh_function(self, c, lo, hi, mult, degree):
        return self._cheb(lambda x: math.tanh(x / mult), c, lo, hi, degree)
