"""Residue-level restatement of CKKS bootstrapping as the product runs it BY DEFAULT.  TEST INFRASTRUCTURE (part of oracle/).

Reference call: context->EvalBootstrap(c) (src/FHEController.cpp:438-449, set up at :237-240 with level budget {3,3});
the algorithm itself lives in OpenFHE (absent: "parity unpinned", oracle/__init__.py).  What is restated here is the order
of residue operations of fhe-linformer_amd/csrc/bootstrap.cpp with every knob at its default:

  message to q0 / 2^correction (integer multiply + rescale) -> ModRaise (orc_modraise) -> SubSum rotations (sparse packing)
  -> CoeffsToSlots: per linear stage ONE hoisted set of baby rotations (== separate orc_rotate calls), inner sums
     sum_b rot_b(x) (.) diag_{g,b} (exact sums of dyadic products), giant steps through the shared-ModDown key switch
     (orc_rotate_each_sum)
  -> conjugation (a rotation by the Galois element 2N-1), real / imaginary split (one ciphertext under sparse packing)
  -> EvalMod: Chebyshev cosine fit (ResidueEvaluator.eval_chebyshev) + R double-angle steps
  -> SlotsToCoeffs stages -> exact multiplication by 2^correction.

Every function is an exact integer function of (input residues, keys, exported plaintext diagonals, Chebyshev coefficients):
the GPU library must return the same residues bit for bit (tests/test_boot_residue_gpu.py).  The diagonals enter as the
residues the library's encoder produced (fhelin_pt_export), as every plaintext operand of oracle/residue_eval.py does; the
coefficients of the cosine fit enter as the doubles the library computed (fhelin_bootstrap_cheb)."""
import numpy as np

import oracle as orc
from oracle.residue_eval import LD, RCt, _llround


class ResidueBootstrapper:
    """mirror of csrc/bootstrap.cpp Bootstrapper::run over a ResidueEvaluator `rev` whose key table holds the rotation keys
    by index, keys["relin"] and keys["conj"].  `desc` = Engine.bootstrap_describe(); `enc(pt)` -> callback (ell, scale) ->
    residues [ell][N] of that diagonal."""

    def __init__(self, rev, desc, enc):
        self.rev, self.desc, self.enc = rev, desc, enc
        self.n = desc["slots"]
        self.N = 1 << rev.log_n

    # ---- helpers that take the slot count of the ciphertext explicitly (the library keeps it on the handle)
    def _rot(self, a, index, ns):
        rev = self.rev
        if index % ns == 0:
            return a
        d = orc.rotate(a.d, rev.keys[index], orc.galois(rev.log_n, index), rev.alpha, rev.q, rev.p, rev.psi_q, rev.psi_p)
        return RCt(d, a.deg, a.scale)

    def _rotate_each_sum(self, cts, indices, ns):
        """Evaluator::rotate_each_sum: index-0 terms are plain addends, the rest go through the shared-ModDown key switch in
        groups of <= 7 (a single leftover through a plain rotation)"""
        rev = self.rev
        acc, rot, ridx = None, [], []
        for c, i in zip(cts, indices):
            if i % ns == 0:
                acc = c if acc is None else rev.add(acc, c)
            else:
                rot.append(c)
                ridx.append(i)
        for lo in range(0, len(rot), 7):
            chunk, idx = rot[lo:lo + 7], ridx[lo:lo + 7]
            if len(chunk) < 2:
                t = self._rot(chunk[0], idx[0], ns)
            else:
                evks = rev._stack(idx)
                d = orc.rotate_each_sum(np.stack([c.d for c in chunk]), evks, [orc.galois(rev.log_n, r) for r in idx], rev.alpha,
                                        rev.q, rev.p, rev.psi_q, rev.psi_p)
                t = RCt(d, chunk[0].deg, chunk[0].scale)
            acc = t if acc is None else rev.add(acc, t)
        return acc

    def apply(self, stage, xin, ns):
        """one linear stage (Bootstrapper::apply): out = sum_g rot_g( sum_b diag_{g,b} (.) rot_b(x) )"""
        rev = self.rev
        x = rev.rescale(xin) if xin.deg >= 2 else xin
        terms = stage["terms"]
        bidx = []
        for (_, b, _) in terms:
            if b not in bidx:
                bidx.append(b)
        babies = {b: self._rot(x, b, ns) for b in bidx}
        lvl = rev.level(x)
        pt_scale = LD(0)
        if x.ell >= 2 and abs(x.scale / rev.sf[lvl] - LD(1)) > LD(1e-12):
            # the first stage after a ModRaise to fewer limbs than the chain has: the diagonals are taken at the scale that
            # lands the product on the next level's own scale
            pt_scale = rev.sf[lvl + 1] * LD(int(rev.q[x.ell - 1])) / x.scale
        sf = pt_scale if pt_scale > 0 else rev.sf[lvl]
        ql = rev.q[:x.ell]
        inner = {}
        for (g, b, pt) in terms:
            e = self.enc(pt)(x.ell, sf)
            prod = np.stack([orc.mul(babies[b].d[k], e, ql) for k in range(2)])
            inner[g] = prod if g not in inner else np.stack([orc.add(inner[g][k], prod[k], ql) for k in range(2)])
        gs = sorted(inner)                                  # std::map order: ascending giant shift, the unrotated one first
        gin = [RCt(inner[g], x.deg + 1, x.scale * sf) for g in gs]
        return self._rotate_each_sum(gin, gs, ns)

    def mod_raise(self, ct, top_ell):
        rev = self.rev
        corr = self.desc["correction"]
        x = rev.rescale(ct) if ct.deg >= 2 else ct
        assert x.npoly == 2 and x.ell >= 2
        if x.ell > 2:
            x = rev.level_reduce(x, 2)
        q0, q1 = LD(int(rev.q[0])), LD(int(rev.q[1]))
        target = q0 / LD(1 << corr)
        k0 = _llround(target * q1 / x.scale)
        assert k0 >= 2
        x = rev.mult_int(x, k0, True, x.scale * LD(k0))
        x = rev.rescale(x)
        rho = x.scale * LD(1 << corr) / q0
        up = RCt(orc.modraise(x.d[:, 0], top_ell, rev.q[:top_ell], rev.psi_q[:top_ell]), 1, rev.sf[0])
        gap = (self.N // 2) // self.n
        j = 1
        while j < gap:                                       # SubSum: rotations by multiples of the slot count
            idx = self.n * j
            r = orc.rotate(up.d, rev.keys[idx], orc.galois(rev.log_n, idx), rev.alpha, rev.q, rev.p, rev.psi_q, rev.psi_p)
            up = rev.add(up, RCt(r, up.deg, up.scale))
            j <<= 1
        return up, rho

    def eval_mod(self, x):
        rev = self.rev
        u = rev.eval_chebyshev(x, self.desc["cheb"], -1.0, 1.0)
        for _ in range(self.desc["R"]):
            if rev.merged_products:          # 2 u^2 - 1 rescaled at once (Evaluator::mult_affine_rescale_batch)
                u = rev.mult_affine_rescale(u, u, 2, -1.0)
                continue
            t = rev.mult(u, u)
            t = rev.add(t, t)
            u = rev.add_real(t, -1.0)
        return u

    def conjugate(self, a):
        rev = self.rev
        d = orc.rotate(a.d, rev.keys["conj"], 2 * self.N - 1, rev.alpha, rev.q, rev.p, rev.psi_q, rev.psi_p)
        return RCt(d, a.deg, a.scale)

    def mult_i(self, x):
        """multiplication by i = by the monomial X^{N/2} (two-ciphertext form of EvalMod, full packing)"""
        rev = self.rev
        mono = np.zeros((x.ell, self.N), dtype=np.uint64)
        mono[:, self.N // 2] = 1
        mono = orc.ntt_batch(mono, rev.q[:x.ell], rev.psi_q[:x.ell])
        ql = rev.q[:x.ell]
        return RCt(np.stack([orc.mul(x.d[k], mono, ql) for k in range(x.npoly)]), x.deg, x.scale)

    def run(self, ct, stop_after=0, drop=0):
        rev, n = self.rev, self.n
        packed = self.desc["packed"]
        L1 = len(rev.q)
        w, rho = self.mod_raise(ct, L1 - drop)
        if stop_after == 1:
            return w
        for st in self.desc["c2s"]:
            w = self.apply(st, w, n)
        ns = 2 * n if packed else n                          # the last stage wrote [w | -i w] over 2n slots
        wc = self.conjugate(w)
        a = rev.add(w, wc)
        if stop_after == 2:
            return a
        if packed:
            v = self.eval_mod(a)
            if stop_after == 3:
                return v
        else:
            b = self.mult_i(rev.sub(wc, w))
            if stop_after == 3:
                return self.eval_mod(a)
            va, vb = self.eval_mod(a), self.eval_mod(b)
            v = rev.add(va, self.mult_i(vb))
        for i, st in enumerate(self.desc["s2c"]):
            v = self.apply(st, v, ns)
            if packed and i == 0:
                ns = n
        corr = self.desc["correction"]
        v = rev.mult_int(v, 1 << corr, False, v.scale)
        v.scale = v.scale * rho
        if v.deg >= 2:
            v = rev.rescale(v)
        return v
