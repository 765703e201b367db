"""Residue-level restatement of the evaluator's LEVELED operations and of the composite circuit ops as the
product runs them BY DEFAULT (merged rotate-and-sum steps, products rescaled before their rotation trees,
log-depth shift trees, block masks in place of the -128 shifts of matmulRElarge).  TEST INFRASTRUCTURE (part of oracle/).

Every function is an exact integer function of its inputs, composed from oracle/fhe_oracle.c (orc_mul, orc_add,
orc_mul_scalar, orc_rescale, orc_rotate, orc_rotate_sum, ...): the GPU library must return the same residues bit for
bit (tests/test_default_path_gpu.py).  Reference functions restated (all src/FHEController.cpp):
  EvalMult(ct,pt) :427  EvalAdd :410,:414  EvalRotate :435,:833  FLEXIBLEAUTO level/scale alignment (implicit in
  :410,:431)  rotsum :829-837  repeat :849-867  matmulRE :869-899  matmulRElarge :915-944  wrapUpExpanded :1070-1084
  unwrapExpanded :1086-1100  wrap_containers :1193-1205.
The order of operations inside a tree (which pairs are merged, which operands are rotated) follows
fhe-linformer_amd/csrc/composite.cpp; it is part of the function (each key switch rounds once).

Scales are numpy longdouble (x87 80-bit, what `long double` is in the library's host code), so the integer
constants of the level adjustment round identically.  Plaintext operands arrive as residue arrays (the encoder is the
client-side row a16, not restated here): `enc(ell, scale)` callbacks return them — tests pass
`lambda ell, scale: eng.pt_export(pt, ell, scale)`."""
import numpy as np

import oracle as orc

LD = np.longdouble


def _llround(x):
    """C llroundl: round half away from zero"""
    x = LD(x)
    r = np.floor(np.abs(x) + LD(0.5))
    return int(r) if x >= 0 else -int(r)


class RCt:
    """ciphertext as the library holds it: residues [npoly][ell][N], noiseScaleDeg, real scale"""

    def __init__(self, d, deg, scale):
        self.d = np.ascontiguousarray(d, dtype=np.uint64)
        self.deg = int(deg)
        self.scale = LD(scale)

    @property
    def ell(self):
        return self.d.shape[1]

    @property
    def npoly(self):
        return self.d.shape[0]


class ResidueEvaluator:
    """mirror of fhe-linformer_amd/csrc/evaluator.cpp bookkeeping over oracle residue functions"""

    def __init__(self, q, p, psi_q, psi_p, alpha, log_n, keys, log_slots=14):
        self.q, self.p = np.asarray(q, dtype=np.uint64), np.asarray(p, dtype=np.uint64)
        self.psi_q, self.psi_p = np.asarray(psi_q, dtype=np.uint64), np.asarray(psi_p, dtype=np.uint64)
        self.alpha, self.log_n = alpha, log_n
        self.keys = keys                      # rotation index -> evk [digits][2][L1+k][N]
        self.slots = 1 << log_slots
        L = len(self.q) - 1
        sf = [LD(int(self.q[L]))]
        for k in range(L):
            sf.append(sf[k] * sf[k] / LD(int(self.q[L - k])))
        self.sf = sf                          # real Delta per level (level = L + 1 - ell)

    # ---- leaf ops
    def level(self, a):
        return len(self.q) - a.ell

    def _each(self, fn, a, *others):
        return np.stack([fn(a.d[c], *[o[c] for o in others]) for c in range(a.npoly)])

    def rescale(self, a):
        ell = a.ell
        d = orc.rescale(a.d, self.q[:ell], self.psi_q[:ell])
        return RCt(d, a.deg - 1 if a.deg > 1 else 1, a.scale / LD(int(self.q[ell - 1])))

    def level_reduce(self, a, new_ell):
        return RCt(a.d[:, :new_ell], a.deg, a.scale)

    def mult_int(self, a, k, raise_deg, new_scale):
        ql = self.q[:a.ell]
        s = np.array([k % int(m) for m in ql], dtype=np.uint64)
        return RCt(self._each(lambda x: orc.mul_scalar(x, s, ql), a), a.deg + 1 if raise_deg else a.deg, new_scale)

    def adjust(self, a, ell, deg, scale):
        """FLEXIBLEAUTO alignment of `a` to (ell, deg, scale): evaluator.cpp Evaluator::adjust"""
        cur = a
        assert cur.ell >= ell
        if cur.ell == ell:
            if cur.deg == deg:
                return cur
            assert cur.deg == 1 and deg == 2
            return self.mult_int(cur, _llround(scale / cur.scale), True, scale)
        if deg == 1:
            if cur.deg == 2:
                cur = self.rescale(cur)
            if cur.ell == ell:
                return cur
            qdrop = LD(int(self.q[ell]))
            k = _llround(scale * qdrop / cur.scale)
            cur = self.mult_int(cur, k, True, cur.scale * LD(k))
            cur = self.level_reduce(cur, ell + 1)
            cur = self.rescale(cur)
            cur.scale = LD(scale)
            return cur
        if cur.deg == 2:
            cur = self.rescale(cur)
        cur = self.mult_int(cur, _llround(scale / cur.scale), True, scale)
        return self.level_reduce(cur, ell)

    def match(self, a, b):
        if a.ell == b.ell and a.deg == b.deg:
            return a, b
        a_rules = a.ell < b.ell or (a.ell == b.ell and a.deg >= b.deg)
        if a_rules:
            return a, self.adjust(b, a.ell, a.deg, a.scale)
        return self.adjust(a, b.ell, b.deg, b.scale), b

    def add(self, a, b):
        x, y = self.match(a, b)
        ql = self.q[:x.ell]
        return RCt(self._each(lambda u, v: orc.add(u, v, ql), x, y.d), x.deg, x.scale)

    def sub(self, a, b):
        x, y = self.match(a, b)
        ql = self.q[:x.ell]
        return RCt(self._each(lambda u, v: orc.sub(u, v, ql), x, y.d), x.deg, x.scale)

    def mult_plain(self, a, enc):
        """EvalMult(ct, pt): a degree-2 operand is rescaled first; the plaintext is encoded at the level's Delta"""
        x = self.rescale(a) if a.deg >= 2 else a
        sf = self.sf[self.level(x)]
        e = enc(x.ell, sf)
        ql = self.q[:x.ell]
        return RCt(self._each(lambda u: orc.mul(u, e, ql), x), x.deg + 1, x.scale * sf)

    def add_plain(self, a, enc):
        e = enc(a.ell, a.scale)
        ql = self.q[:a.ell]
        d = a.d.copy()
        d[0] = orc.add(a.d[0], e, ql)
        return RCt(d, a.deg, a.scale)

    def _g(self, index):
        return orc.galois(self.log_n, index)

    def rotate(self, a, index):
        if index % self.slots == 0:
            return a
        d = orc.rotate(a.d, self.keys[index], self._g(index), self.alpha, self.q, self.p, self.psi_q, self.psi_p)
        return RCt(d, a.deg, a.scale)

    def rotate_add(self, a, index):
        """a + EvalRotate(a, index): one step of the reference's rotsum loop (:833)"""
        if index % self.slots == 0:
            return self.add(a, a)
        return self.add(self.rotate(a, index), a)

    def _stack(self, indices):
        """the keys of a merged key switch as one array; cached per index set (a whole-pass run asks for the same dozen sets
        thousands of times, and one stack of seven 73 MB keys is a 0.5 GB copy)"""
        if not hasattr(self, "_stacks"):
            self._stacks = {}
        k = tuple(int(r) for r in indices)
        if k not in self._stacks:
            self._stacks[k] = np.stack([self.keys[r] for r in indices])
        return self._stacks[k]

    def rotate_sum(self, a, indices):
        """merged tree steps: a + sum_r rot(a, r), one ModUp / one ModDown (Evaluator::rotate_sum_batch)"""
        evks = self._stack(indices)
        gs = [self._g(r) for r in indices]
        d = orc.rotate_sum(a.d, evks, gs, self.alpha, self.q, self.p, self.psi_q, self.psi_p)
        return RCt(d, a.deg, a.scale)

    def rotate_each_sum(self, cts, indices):
        """sum_i rot(cts[i], indices[i]) (Evaluator::rotate_each_sum): index-0 terms are plain addends, the others go
        through the shared-ModDown key switch in groups of <= 7, single leftovers through a plain rotation"""
        acc, rot, ridx = None, [], []
        for c, i in zip(cts, indices):
            if i % self.slots == 0:
                acc = c if acc is None else self.add(acc, c)
            else:
                rot.append(c)
                ridx.append(i)
        for lo in range(0, len(rot), 7):
            chunk, idx = rot[lo:lo + 7], ridx[lo:lo + 7]
            if len(chunk) < 2:
                t = self.rotate(chunk[0], idx[0])
            else:
                evks = self._stack(idx)
                d = orc.rotate_each_sum(np.stack([c.d for c in chunk]), evks, [self._g(r) for r in idx], self.alpha,
                                        self.q, self.p, self.psi_q, self.psi_p)
                t = RCt(d, chunk[0].deg, chunk[0].scale)
            acc = t if acc is None else self.add(acc, t)
        return acc

    def hoisted_dot(self, a, encs, indices, rescale=False):
        """a * V_0 + sum_r rot(a, indices[r]) * V_{r+1} (Evaluator::hoisted_dot_rows, one row): a degree-2 operand is rescaled
        first, the plaintexts are encoded at the level's Delta over the FULL key basis (enc(n_q + n_p, scale)), one ModUp and
        one ModDown, the plaintext products taken in QP (orc_hoisted_dot).  rescale: the result rescaled, ModDown and rescale as
        ONE basis conversion (P and the top limb dropped together)"""
        x = self.rescale(a) if a.deg >= 2 else a
        sf = self.sf[self.level(x)]
        nl = len(self.q) + len(self.p)
        if not hasattr(self, "_full"):
            self._full = {}
        k = (tuple(id(e) for e in encs), float(sf), float(sf - LD(float(sf))))
        if k not in self._full:           # the same few plaintexts serve every row of a call
            self._full.clear()
            self._full[k] = np.stack([e(nl, sf) for e in encs])
        d = orc.hoisted_dot(x.d, self._stack(indices), [self._g(r) for r in indices], self._full[k], self.alpha, self.q, self.p,
                            self.psi_q, self.psi_p, drop=rescale)
        if rescale:
            return RCt(d, x.deg, x.scale * sf / LD(int(self.q[x.ell - 1])))
        return RCt(d, x.deg + 1, x.scale * sf)

    # ---- ct x ct, real constants, polynomial evaluation (reference :431, :1289-1336; order of csrc/polyeval.cpp)
    def mult(self, a, b):
        """EvalMult(ct, ct) (:431): degree-2 operands are rescaled first, the pair is level-adjusted, tensor + relinearisation
        (keys[0] is the relinearisation key).  Evaluator::mult and one pair of Evaluator::mult_batch."""
        x = self.rescale(a) if a.deg >= 2 else a
        y = x if b is a else (self.rescale(b) if b.deg >= 2 else b)
        x, y = self.match(x, y)
        d = orc.mult_relin(x.d, y.d, self.keys["relin"], self.alpha, self.q, self.p, self.psi_q, self.psi_p)
        return RCt(d, x.deg + y.deg, x.scale * y.scale)

    cheb_leaf_classes = True    # Evaluator::cheb_leaf_classes (FHELIN_CHEB_LEAF_CLASSES=0 restores all babies at one level)
    merged_products = False     # Evaluator::merged_products (FHELIN_MERGED_PRODUCTS=1, off by default): the power steps of a Chebyshev
                                # evaluation / EvalMod's double angle through mult_affine_rescale

    def mult_affine_rescale(self, a, b, f, cadd=0.0, sub=None):
        """rescale(f * mult(a, b) + cadd - sub) as Evaluator::mult_affine_rescale_batch runs it: operands as in mult(); the constant is
        round(cadd * scale) on component 0 and `sub` is adjusted to the product's (limbs, degree 2, scale), both at the product's
        scale a.scale * b.scale; they enter the key switch's accumulator times P and ModDown + rescale are ONE conversion
        (orc_mult_affine_rescale).  One rounding where mult / add / rescale have two."""
        x = self.rescale(a) if a.deg >= 2 else a
        y = x if b is a else (self.rescale(b) if b.deg >= 2 else b)
        x, y = self.match(x, y)
        sc = x.scale * y.scale
        ql = self.q[:x.ell]
        addq = None
        if sub is not None:
            s_adj = self.adjust(sub, x.ell, 2, sc)
            addq = np.stack([orc.sub(np.zeros_like(s_adj.d[c]), s_adj.d[c], ql) for c in range(2)])
        if cadd != 0.0:
            if addq is None:
                addq = np.zeros_like(x.d)
            addq[0] = orc.add_scalar(addq[0], self.real_scalars(LD(cadd) * sc, x.ell), ql)
        d = orc.mult_affine_rescale(x.d, y.d, self.keys["relin"], f, addq, self.alpha, self.q, self.p, self.psi_q, self.psi_p)
        return RCt(d, 1, sc / LD(int(self.q[x.ell - 1])))

    def real_scalars(self, v, ell):
        """polyeval.cpp real_to_scalars: round(|v|) (half away from zero, 80-bit v) with the sign restored, modulo each limb"""
        k = _llround(v)
        return np.array([k % int(m) for m in self.q[:ell]], dtype=np.uint64)

    def mult_real(self, a, cst):
        """EvalMult(ct, double): a constant vector encodes to a constant polynomial -> per-limb scalar round(cst * Delta_level)"""
        x = self.rescale(a) if a.deg >= 2 else a
        sf = self.sf[self.level(x)]
        s = self.real_scalars(LD(cst) * sf, x.ell)
        ql = self.q[:x.ell]
        return RCt(self._each(lambda u: orc.mul_scalar(u, s, ql), x), x.deg + 1, x.scale * sf)

    def add_real(self, a, cst):
        """EvalAdd(ct, double): round(cst * scale) added to component 0"""
        ql = self.q[:a.ell]
        d = a.d.copy()
        d[0] = orc.add_scalar(a.d[0], self.real_scalars(LD(cst) * a.scale, a.ell), ql)
        return RCt(d, a.deg, a.scale)

    MAX_LINCOMB = 32       # LinComb::MAX_TERMS (kernels_elem.h): above it the library runs the chain of single operations

    def lincomb(self, terms, coef, c0):
        return self.lincomb_at(terms, coef, c0, 0, 0)

    def lincomb_at(self, terms_in, coef_in, c0, want_scale, keep_ell):
        """sum_k coef_k T_k + c0 (EvalLinearWSum; Evaluator::lincomb_at).  Uniform operands: every coefficient becomes the scalar
        round(coef * f) with f = the level's Delta, or want_scale / scale when the sum has to be born at `want_scale` on the first
        keep_ell limbs (a summand of a Paterson-Stockmeyer product).  Otherwise the chain mult_real / add / add_real; None
        when a (want_scale, keep_ell) request cannot be met that way."""
        pairs = [(t, c) for t, c in zip(terms_in, coef_in) if c != 0.0]
        assert pairs, "lincomb: all coefficients are zero"
        terms, coef = [t for t, _ in pairs], [c for _, c in pairs]
        x = terms[0]
        uniform = 2 <= len(terms) <= self.MAX_LINCOMB and all(
            t.deg == 1 and t.npoly == x.npoly and t.ell == x.ell and abs(t.scale / x.scale - LD(1)) < LD(1e-9) for t in terms)
        if not uniform:
            if want_scale > 0 or keep_ell > 0:
                return None
            acc = None
            for t, c in zip(terms, coef):
                m = self.mult_real(t, c)
                acc = m if acc is None else self.add(acc, m)
            return self.add_real(acc, c0) if c0 != 0.0 else acc
        ell = min(keep_ell, x.ell) if keep_ell > 0 else x.ell
        sf = LD(want_scale) / x.scale if want_scale > 0 else self.sf[self.level(x)]
        out_scale = x.scale * sf
        ql = self.q[:ell]
        acc = None
        for t, c in zip(terms, coef):
            s = self.real_scalars(LD(c) * sf, ell)
            m = np.stack([orc.mul_scalar(t.d[k, :ell], s, ql) for k in range(t.npoly)])
            acc = m if acc is None else np.stack([orc.add(acc[k], m[k], ql) for k in range(x.npoly)])
        if c0 != 0.0:
            acc[0] = orc.add_scalar(acc[0], self.real_scalars(LD(c0) * out_scale, ell), ql)
        return RCt(acc, 2, out_scale)

    def align_deg1(self, v):
        """polyeval.cpp align_deg1: a set of powers to one common (fewest limbs, degree 1, that level's Delta)"""
        v = [self.rescale(t) if t.deg >= 2 else t for t in v]
        ell = min(t.ell for t in v)
        sf = self.sf[len(self.q) - ell]
        return [self.adjust(t, ell, 1, sf) for t in v]

    def mult_many(self, v):
        """EvalMultMany (:1297): pairwise product tree; identical operand pairs give one product"""
        cur = list(v)
        while len(cur) > 1:
            nxt = []
            for i in range(0, len(cur) - 1, 2):
                if i >= 2 and cur[i] is cur[i - 2] and cur[i + 1] is cur[i - 1]:
                    nxt.append(nxt[-1])
                else:
                    nxt.append(self.mult(cur[i], cur[i + 1]))
            if len(cur) & 1:
                nxt.append(cur[-1])
            cur = nxt
        return cur[0]

    def eval_poly(self, x, coeffs):
        """EvalPoly (:1291), power basis: x^i by the binary power tree, powers aligned, one linear combination"""
        n = len(coeffs) - 1
        while n > 0 and coeffs[n] == 0.0:
            n -= 1
        pw = [None] * (n + 1)
        pw[1] = self.rescale(x) if x.deg >= 2 else x
        for i in range(2, n + 1):
            hi = 1
            while hi * 2 <= i:
                hi *= 2
            pw[i] = self.mult(pw[i // 2], pw[i // 2]) if hi == i else self.mult(pw[hi], pw[i - hi])
        terms = self.align_deg1(pw[1:])
        return self.lincomb(terms, list(coeffs[1:n + 1]), coeffs[0])

    def eval_chebyshev(self, x, coeffs_in, a=-1.0, b=1.0, leaf_at_product=True):
        """EvalChebyshevFunction's series evaluation (:1319-1335) as polyeval.cpp runs it: c0/2 + sum c_k T_k(u); babies
        T_1..T_{b-1} (b = 2^ceil(log2(n+1)/2)) by T_k = 2 T_floor(k/2) T_ceil(k/2) - T_(k mod 2), giants T_b, T_2b, ... by
        squaring, Paterson-Stockmeyer recursion p = q T_m + r.  The library evaluates the recursion's products in batched
        rounds; every node computes mult(q, T_m) then add(., r), so the order does not enter the residues."""
        a, b = float(a), float(b)
        c = [float(v) for v in coeffs_in]
        n = len(c) - 1
        while n > 0 and c[n] == 0.0:
            n -= 1
        c = c[:n + 1]
        assert n >= 1
        c[0] *= 0.5
        u = x
        if not (a == -1.0 and b == 1.0):
            u = self.mult_real(x, 2.0 / (b - a))
            u = self.add_real(u, -(a + b) / (b - a))
        if u.deg >= 2:
            u = self.rescale(u)
        l = 0
        while (1 << (2 * l)) < n + 1:
            l += 1
        baby = max(2, 1 << l)
        T = [None] * (baby + 1)
        T[1] = u
        h = 1
        while h < baby:
            for k in range(h + 1, min(2 * h, baby) + 1):
                if self.merged_products:
                    T[k] = self.mult_affine_rescale(T[k // 2], T[k - k // 2], 2, -1.0 if k % 2 == 0 else 0.0, None if k % 2 == 0 else T[1])
                    continue
                t = self.mult(T[k // 2], T[k - k // 2])
                t = self.add(t, t)
                t = self.add_real(t, -1.0) if k % 2 == 0 else self.sub(t, T[1])
                T[k] = self.rescale(t)
            h *= 2
        G = {baby: T[baby]}
        m = baby
        while m * 2 <= n:
            if self.merged_products:
                G[2 * m] = self.mult_affine_rescale(G[m], G[m], 2, -1.0)
                m *= 2
                continue
            t = self.mult(G[m], G[m])
            t = self.add_real(self.add(t, t), -1.0)
            G[2 * m] = self.rescale(t)
            m *= 2
        # a leaf's babies are aligned to the deepest power THE LEAF uses (polyeval.cpp cheb_recurse babies_for): degrees in
        # (2^(j-1), 2^j] take T_1..T_min(2^j, baby-1) at their common level - one level saved against all babies at one level
        aligned = {}

        def babies_for(d):
            j = 0
            while (1 << j) < max(d, 1):
                j += 1
            prefix = min(1 << j, baby - 1) if self.cheb_leaf_classes else baby - 1
            if prefix not in aligned:
                aligned[prefix] = [None] + self.align_deg1(T[1:prefix + 1])
            return aligned[prefix]

        def strip(p):
            k = len(p) - 1
            while k > 0 and p[k] == 0.0:
                k -= 1
            return p[:k + 1]

        def leaf(p, at=None):
            """sum_{k>=1} p_k T_k + p_0 over the aligned babies; at = the product it will be added to"""
            if not any(v != 0.0 for v in p[1:]):
                z = self.mult_real(babies_for(baby - 1)[1], 0.0)
                return self.add_real(z, p[0]) if p[0] != 0.0 else z
            terms, cf = babies_for(len(p) - 1)[1:len(p)], p[1:]
            if at is not None and at.deg == 2:
                v = self.lincomb_at(terms, cf, p[0], at.scale, at.ell)
                if v is not None:
                    return v
            return self.lincomb(terms, cf, p[0])

        def node(p, at=None):
            p = strip(p)
            k = len(p) - 1
            if k < baby:
                return leaf(p, at)
            mm = baby
            while mm * 2 <= k:
                mm *= 2
            qc, rc = [0.0] * (k - mm + 1), list(p[:mm])
            qc[0] = p[mm]
            for i in range(mm + 1, k + 1):
                qc[i - mm] = 2 * p[i]
                rc[2 * mm - i] -= p[i]
            prod = self.mult(node(qc), G[mm])
            if all(v == 0.0 for v in rc):
                return prod
            rs = strip(rc)
            r_is_leaf = len(rs) - 1 < baby
            if r_is_leaf and leaf_at_product and any(v != 0.0 for v in rs[1:]):
                return self.add(prod, leaf(rs, prod))
            return self.add(prod, node(rc))

        return node(c)

    # ---- composites as composite.cpp runs them by default
    @staticmethod
    def log_steps(slots):
        n = 0
        while n < np.log2(slots):
            n += 1
        return n

    def have(self, indices):
        return all(r % self.slots != 0 and r in self.keys for r in indices)

    def tree_steps(self, r, n, unit, merge=True):
        i = 0
        while i < n:
            s = unit * (1 << i)
            if i + 2 < n and merge and self.have([m * s for m in range(1, 8)]):
                r = self.rotate_sum(r, [m * s for m in range(1, 8)])
                i += 3
            elif i + 1 < n and merge and self.have([s, 2 * s, 3 * s]):
                r = self.rotate_sum(r, [s, 2 * s, 3 * s])
                i += 2
            else:
                r = self.rotate_add(r, s)
                i += 1
        return r

    def rotsum(self, a, slots, padding, early_rescale=True, merge=True):
        n = self.log_steps(slots)
        r = self.rescale(a) if (n and early_rescale and a.deg >= 2 and a.ell >= 2) else a
        return self.tree_steps(r, n, padding, merge) if n else a

    def repeat(self, a, slots, padding=1, early_rescale=True, merge=True):
        return self.rotsum(a, slots, -padding, early_rescale, merge)

    def shift_sum(self, terms, step, merge=True):
        """sum_i rot(terms[i], step * i) as Composite::shift_sum builds it: radix-8 levels, every group of eight consecutive terms one
        shared-ModDown key switch (seven rotated terms + the unrotated one), then the same with the unit 8 * step; the binary tree
        of round 1 over what is left when the keys of a level are missing"""
        cur, unit = list(terms), step
        while len(cur) > 1 and merge:
            width = min(8, len(cur))
            if not self.have([unit * k for k in range(1, width)]):
                break
            full, nxt = len(cur) // 8, []
            for m in range(full):
                nxt.append(self.rotate_each_sum(cur[8 * m:8 * m + 8], [unit * k for k in range(8)]))
            if len(cur) % 8:
                tail = cur[8 * full:]
                nxt.append(tail[0] if len(tail) == 1 else self.rotate_each_sum(tail, [unit * k for k in range(len(tail))]))
            cur, unit = nxt, unit * 8
        level = 0
        while len(cur) > 1:
            odd = [self.rotate(c, unit * (1 << level)) for c in cur[1::2]]
            even = cur[0:len(cur) - 1:2]
            nxt = [self.add(e, o) for e, o in zip(even, odd)]
            if len(cur) & 1:
                nxt.append(cur[-1])
            cur = nxt
            level += 1
        return cur[0]

    def fan_row(self, c, i, step, memo, top=None, merge=True):
        """rot(c, step * i) as Composite::shift_fan_rows composes it: i = 64 a + 8 b + k, rot(rot(rot(c, 64 step a), 8 step b), step k)
        (three hoisted levels in the library; a hoisted rotation equals the separate one); the doubling chain of round 1 when the
        keys are missing.  memo: shared between the rows of one fan."""
        top = i if top is None else top
        need = [step * k for k in range(1, 8) if k <= top] + [8 * step * b for b in range(1, 8) if 8 * b <= top] + \
               [64 * step * a for a in range(1, top // 64 + 1)]
        if not (merge and (not need or self.have(need))):
            if i not in memo:
                h = 1 << (i.bit_length() - 1) if i else 0
                memo[i] = c if i == 0 else self.rotate(self.fan_row(c, i - h, step, memo, top, merge), step * h)
            return memo[i]
        a, b, k = i // 64, (i % 64) // 8, i % 8
        if ("a", a) not in memo:
            memo[("a", a)] = self.rotate(c, 64 * step * a) if a else c
        if ("b", a, b) not in memo:
            memo[("b", a, b)] = self.rotate(memo[("a", a)], 8 * step * b) if b else memo[("a", a)]
        if ("k", i) not in memo:       # a row is asked for many times when the rows are window sums over the fan (unwrapExpanded_bulk)
            memo[("k", i)] = self.rotate(memo[("b", a, b)], step * k) if k else memo[("b", a, b)]
        return memo[("k", i)]

    def shift_fan(self, c, n, step):
        """rot(c, step * i), i < n (Composite::shift_fan)"""
        memo = {}
        return [self.fan_row(c, i, step, memo, n - 1) for i in range(n)]

    def matmul_pt(self, rows, w_enc, bias_enc, slots, padding):
        out = [self.rotsum(self.mult_plain(r, w_enc), slots, padding) for r in rows]
        if bias_enc is not None:
            out = [self.add_plain(o, bias_enc) for o in out]
        return out

    def relarge_u(self, row, v_encs):
        """composite.cpp relarge_u: U = x * V_0 + sum_{t=1..3} rot(x, 128 t) * V_t (double hoisting: one ModUp, the plaintext products in
        the extended basis, ModDown and rescale as one conversion); v_encs[t] encodes V_t = rot(W''_t, 128 t) (block b of W''_t =
        block b of W_((b - t) mod 4)); a degree-2 input is rescaled first"""
        return self.hoisted_dot(row, v_encs, [128, 256, 384], rescale=True)

    def matmulRElarge(self, rows, v_encs, bias_enc, mask512_enc):
        """composite.cpp matmulRElarge, shared form: U (relarge_u), Z = rotsum(U, 32, 512), out = Z * mask[0,512) + bias"""
        out = []
        for r in rows:
            z = self.rotsum(self.relarge_u(r, v_encs), 32, 512)
            o = self.mult_plain(z, mask512_enc)
            out.append(self.add_plain(o, bias_enc) if bias_enc is not None else o)
        return out

    def relarge_container(self, us, mask_encs, bias_tiled_enc):
        """composite.cpp relarge_container: one container of generate_containers(matmulRElarge(.)) from the U rows of its q <= 32
        tokens: W_k = sum_i mask_((i + k) mod 32) * U_i for k < 32 (exact sums of dyadic products, noise degree 2), C = sum_k
        rot(W_k, 512 k) (shift_sum), + the bias of the q tokens tiled at their 512-slot blocks.  mask_encs[j] encodes the mask
        value on slots [512 j, 512 j + 512)."""
        x0 = us[0]
        sf = self.sf[self.level(x0)]
        ql = self.q[:x0.ell]
        encs = [m(x0.ell, sf) for m in mask_encs]
        w = []
        for k in range(32):
            ms = [encs[(i + k) % 32] for i in range(len(us))]
            acc = np.stack([orc.dot([u.d[c] for u in us], ms, ql) for c in range(2)])
            w.append(RCt(acc, x0.deg + 1, x0.scale * sf))
        c = self.shift_sum(w, 512)
        return self.add_plain(c, bias_tiled_enc) if bias_tiled_enc is not None else c

    def unwrapRepeatedLarge(self, containers, n_tokens, enc_of_values):
        """composite.cpp unwrapRepeatedLarge (two-stage shared form): per container, block k and range a (8 tokens = 4096
        slots) one mask (slot mod 512 in block k, slot in range a), one merged key switch copying block k of every token of
        the range over the token's own 512 slots, and repeat(., 4, -4096); per (token, k) the inner sum of the eight rotations of that
        result with the token's shifted masks.  enc_of_values(vector) -> enc callback for that vector."""
        ns = self.slots
        idx = np.arange(ns)
        D = {}
        for t in range(n_tokens):
            i, j = divmod(t, 32)
            D.setdefault((i, j // 8), None)
        for (i, a) in D:
            row = []
            for k in range(4):
                m = ((idx % 512 >= 128 * k) & (idx % 512 < 128 * (k + 1)) & (idx >= 4096 * a) & (idx < 4096 * (a + 1))).astype(np.float64)
                A = self.rescale(self.mult_plain(containers[i], enc_of_values(m)))
                B = self.rotate_sum(A, [128 * (k - mm) for mm in range(4) if mm != k])
                row.append(self.repeat(B, 4, -4096))
            D[(i, a)] = row
        # stage 2: out_{8a+b, k} = sum_{m<8} rot(D_k, 512 m) * mask_{(b - m) mod 8}  (== sum_m rot(D_k * mask_b, 512 m): the rotation
        # commutes with the mask), the eight rotations of D_k shared by the tokens of the range; exact sums of dyadic products, noise degree 2
        masks = [((idx % 4096 >= 512 * bb) & (idx % 4096 < 512 * (bb + 1))).astype(np.float64) for bb in range(8)]
        rot = {}
        out = []
        for t in range(n_tokens):
            i, j = divmod(t, 32)
            a, b = divmod(j, 8)
            four = []
            for k in range(4):
                if (i, a, k) not in rot:
                    rot[(i, a, k)] = [D[(i, a)][k]] + [self.rotate(D[(i, a)][k], 512 * m) for m in range(1, 8)]
                acc = None
                for m in range(8):
                    term = self.mult_plain(rot[(i, a, k)][m], enc_of_values(masks[(b - m) % 8]))
                    acc = term if acc is None else self.add(acc, term)
                four.append(acc)
            out.append(four)
        return out

    def wrap_containers(self, cts, n):
        terms = list(cts[:n])[::-1]
        return self.shift_sum(terms, -512)

    def wrapUpExpanded(self, cts, mask_enc):
        return self.shift_sum([self.mult_plain(c, mask_enc) for c in cts], -1)

    def unwrapExpanded(self, c, n, mask_enc):
        return [self.repeat(self.mult_plain(x, mask_enc), 128, 1) for x in self.shift_fan(c, n, 1)]

    def unwrapExpanded_bulk(self, c, n, idx, mask_encs):
        """Composite::unwrapExpanded_bulk (many rows of one call read together): row i = sum_{k<128} mask_k * rot(c, i - k), mask_k =
        slots = k mod 128 - exact sums of dyadic products (noise degree 2); the rotations rot(c, j) are the hoisted fans of
        shift_fan_rows: the call's own fan for j >= 0, the fan by -1 over 128 rows for j < 0; a degree-2 input is rescaled first.
        mask_encs[k] encodes mask_k."""
        x = self.rescale(c) if c.deg >= 2 else c
        mp, mn = {}, {}
        sf = self.sf[self.level(x)]
        ql = self.q[:x.ell]
        encs = [m(x.ell, sf) for m in mask_encs]
        out = []
        for i in idx:
            ds = [self.fan_row(x, i - k, 1, mp, n - 1) if i - k >= 0 else self.fan_row(x, k - i, -1, mn, 127) for k in range(128)]
            d = np.stack([orc.dot([t.d[cmp] for t in ds], encs, ql) for cmp in range(2)])
            out.append(RCt(d, x.deg + 1, x.scale * sf))
        return out
