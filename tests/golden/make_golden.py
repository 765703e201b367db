"""Generates tests/golden/plain_forward_s129.json by running the REFERENCE's own plaintext model
(/root/reference/src/python/compute_simple.py main()) on synthetic seeded weights written in the text format of
extract_parameters_numeric.py:28.  Run in the build container only (the reference does not travel):
    python tests/golden/make_golden.py
Inputs are regenerated from the seeds at test time (oracle/plain_forward.py synthetic_model / synthetic_tokens);
the fixture holds the seeds and the reference's printed intermediates / prediction."""
import builtins
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/python")
from oracle import plain_forward as pf  # noqa: E402

S, W_SEED, T_SEED = 129, 1234, 4321


def main():
    w = pf.synthetic_model(W_SEED)
    x_emb = pf.synthetic_tokens(S, T_SEED)
    with tempfile.TemporaryDirectory() as d:
        wd, td = os.path.join(d, "weights"), os.path.join(d, "tokens")
        os.makedirs(wd), os.makedirs(td)
        for name, arr in pf.weight_files(w).items():
            a = np.asarray(arr, dtype=np.float64)
            rows = a if a.ndim == 2 else a.reshape(1, -1)          # 1-D vectors: one comma-separated line
            np.savetxt(os.path.join(wd, name), rows, delimiter=",", fmt="%.18e")
        for i in range(S):
            np.savetxt(os.path.join(td, f"input_{i}.txt"), x_emb[i].reshape(1, -1), delimiter=" ", fmt="%.18e")
        import compute_simple  # the reference module (numpy + torch only)
        captured = []
        real_print = builtins.print
        builtins.print = lambda *a, **k: captured.append(a)
        argv = sys.argv
        sys.argv = ["compute_simple.py", "--tokens_dir", td, "--weights_dir", wd]
        try:
            compute_simple.main()
        finally:
            builtins.print = real_print
            sys.argv = argv
    out = {"S": S, "weights_seed": W_SEED, "tokens_seed": T_SEED, "source": "reference src/python/compute_simple.py main()"}
    for a in captured:
        if len(a) == 2 and isinstance(a[0], str) and isinstance(a[1], np.ndarray):
            key = a[0].rstrip(":").strip()
            out.setdefault(key, np.asarray(a[1], dtype=np.float64).reshape(-1).tolist())
        elif len(a) == 1 and isinstance(a[0], np.ndarray):
            out["y_prob"] = np.asarray(a[0], dtype=np.float64).tolist()
        elif len(a) == 4 and a[0] == "Pred:":
            out["Pred"], out["Prob"] = int(a[1]), float(a[3])
    if "K" in out:
        out["K"] = out["K"][:256]           # first two rows are enough to pin the layout
    path = os.path.join(ROOT, "tests", "golden", "plain_forward_s129.json")
    with open(path, "w") as f:
        json.dump(out, f)
    real_print("wrote", path, {k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})


def helpers():
    """pin the reference's offline weight splitters (split_ffn_w1.py:24-37, split_ffn_w2_cols.py:22-29) on seeded matrices"""
    import hashlib
    import split_ffn_w1
    import split_ffn_w2_cols
    rng = np.random.default_rng(77)
    W0 = rng.normal(size=(512, 128))
    W2 = rng.normal(size=(128, 512))
    b0 = split_ffn_w1.split_transposed_blocks(W0, 128)
    b2 = split_ffn_w2_cols.split_col_blocks(W2, 128)
    dig = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()
    out = {"seed": 77, "source": "reference split_ffn_w1.split_transposed_blocks / split_ffn_w2_cols.split_col_blocks",
           "w1_blocks_sha256": [dig(b) for b in b0], "w2_blocks_sha256": [dig(b) for b in b2],
           "w1_block0_row0_head": np.asarray(b0[0], dtype=np.float64)[0, :4].tolist(),
           "w2_block3_row5_head": np.asarray(b2[3], dtype=np.float64)[5, :4].tolist()}
    path = os.path.join(ROOT, "tests", "golden", "weight_split_helpers.json")
    json.dump(out, open(path, "w"))
    print("wrote", path)


if __name__ == "__main__":
    main()
    helpers()
