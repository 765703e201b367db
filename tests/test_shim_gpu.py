"""The C++ shim include/FHEController.h driven like the reference's main.cpp (tests/shim/shim_driver.cpp),
at the reference's literal parameter set, compared with oracle/slotsim.py by decryption (tolerance 1e-4:
eval_exp raises a degree-6 polynomial to the 8th power, ~2^-30 relative noise)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "shim", "shim_driver")


def _write(path, arr):
    with open(path, "w") as f:
        for row in np.atleast_2d(arr):
            f.write(",".join("%.18e" % v for v in row) + "\n")


def _read(path):
    return np.array([float(v) for v in open(path).read().strip().split(",")])


@pytest.mark.parametrize("early_rescale", ["1", "0"])
def test_shim_driver_matches_slot_model(tmp_path, early_rescale):
    """early_rescale=0 keeps OpenFHE's lazy order (a product stays at its level until its next multiplication), so
    GetLevel() reads exactly what the reference would print; the default rescales a product before its rotation tree
    (same values, one limb fewer per key switch) and GetLevel() of such a product reads one higher."""
    from oracle import slotsim as sim
    assert os.path.exists(BIN), "tests/shim/shim_driver missing: run __graft_entry__.build()"
    rng = np.random.default_rng(2024)
    xs = [rng.normal(0, 0.3, 128) for _ in range(3)]
    W = rng.normal(0, 0.05, (128, 128))
    bias = rng.normal(0, 0.05, 128)
    for i, x in enumerate(xs):
        _write(tmp_path / f"input_{i}.txt", x)
    _write(tmp_path / "W_T.txt", W)          # one row per line, like extract_parameters_numeric.py:28
    _write(tmp_path / "bias.txt", bias)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fhe-linformer_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               FHELIN_EARLY_RESCALE=early_rescale)
    r = subprocess.run([BIN, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Could not find" in r.stderr          # load_ciphertext of a missing file: message + null handle
    wv, bv = W.reshape(-1), np.tile(bias, 128)
    Q = sim.matmul([np.repeat(x, 128) for x in xs], wv, bv, 128, 128)
    for i in range(3):
        assert np.max(np.abs(_read(tmp_path / f"Q_{i}.out") - Q[i])) < 1e-6
        assert np.max(np.abs(Q[i][:128] - (xs[i] @ W + bias))) < 1e-12
    K = sim.wrapUpRepeated(Q)
    assert np.max(np.abs(_read(tmp_path / "K_wrapped.out") - K)) < 1e-6
    scores = sim.matmulScores([Q[0]], K)
    assert np.max(np.abs(_read(tmp_path / "scores.out") - scores)) < 1e-6
    taylor = np.polyval([1 / 720, 1 / 120, 1 / 24, 1 / 6, 1 / 2, 1, 1], scores) ** 8
    idx = np.arange(16384)
    expv = taylor + np.where((idx % 128 < 3) & (idx < 128 * 3), 0.0, -1.0)
    assert np.max(np.abs(_read(tmp_path / "exp.out") - expv)) < 1e-4
    assert np.max(np.abs(_read(tmp_path / "sum.out") - sim.rotsum(expv, 32, 128))) < 1e-3
    cr = sim.matmul([Q[0]], wv, None, 128, 1)[0]
    assert np.max(np.abs(_read(tmp_path / "cr.out") - cr)) < 1e-6
    lv = open(tmp_path / "levels.out").read().strip().split(",")
    assert lv[0] == ("1" if early_rescale == "1" else "0") and lv[3] == "1" and int(lv[1]) >= 1 and int(lv[2]) > int(lv[1])
