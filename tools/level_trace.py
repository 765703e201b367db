"""Probe: limb count (ell) of the ciphertexts entering and leaving every controller call of one forward pass (GPU box only;
eager rows).  Shows where the chain has slack before a bootstrap."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fhe_linformer_amd as fa
from fhe_linformer_amd import linformer as lf
from oracle import plain_forward as pf
from fhe_linformer_amd import capi

S = int(sys.argv[1]) if len(sys.argv) > 1 else 129
w = pf.synthetic_model(1234); x = pf.synthetic_tokens(S, 4321)
x_in, X_E, X_F = pf.client_inputs(w, x)
e = fa.Engine("bench", seed=11, n_q=28, n_p=-1)
e.set_lazy_rows(False)
e.keygen(); e.gen_relin_key()
e.gen_rotation_keys(fa.circuit_rotation_indices())
e.bootstrap_setup(3, 3, 16384)
depth = [0]


def ells(v):
    if isinstance(v, capi.Ct):
        return [v.info()["ell"]]
    if isinstance(v, (list, tuple)):
        out = []
        for x in v:
            out += ells(x)
        return out
    return []


def brief(l):
    if not l:
        return "-"
    return f"{min(l)}" if min(l) == max(l) else f"{min(l)}..{max(l)}"


def wrap(name, fn):
    def inner(*a, **k):
        if depth[0] > 0:
            return fn(*a, **k)
        depth[0] += 1
        try:
            r = fn(*a, **k)
        finally:
            depth[0] -= 1
        i, o = ells(list(a)), ells(r)
        if i or o:
            print(f"{name:28s} in x{len(i):<4d} ell {brief(i):8s} -> out x{len(o):<4d} ell {brief(o)}")
        return r
    return inner


ctl = lf.GpuController(e)
for name in dir(ctl):
    if name.startswith("_") or name in ("level", "clone", "e", "n_boot", "verbose"):
        continue
    fn = getattr(ctl, name)
    if callable(fn):
        setattr(ctl, name, wrap(name, fn))
enc = lf.encrypt_inputs(ctl, x_in, X_E, X_F)
out = lf.forward_encrypted(ctl, w, enc)
