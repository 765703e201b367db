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


REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "ref_main")
FWD = os.path.join(ROOT, "tests", "shim", "shim_forward")


def _write_model(root, w, x_in, X_E, X_F):
    """synthetic weights / tokens in the files and formats the reference's drivers read (src/main.cpp:159-173,
    :177-470; written like src/python/extract_parameters_numeric.py:28, split like split_ffn_w1.py / split_ffn_w2_cols.py)"""
    from oracle.plain_forward import PFX
    from fhe_linformer_amd import linformer as lf
    wd = os.path.join(root, "weights-20NG")
    for d in ("weights-20NG", "input", "tokens", "build"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    files = {
        "cls_token.txt": x_in[0],
        PFX + "selfAttn_WQ_weight_T.txt": w["WQ"].T, PFX + "selfAttn_WQ_bias.txt": w["BQ"],
        PFX + "selfAttn_WK_weight_T.txt": w["WK"].T, PFX + "selfAttn_WK_bias.txt": w["BK"],
        PFX + "selfAttn_WV_weight_T.txt": w["WV"].T, PFX + "selfAttn_WV_bias.txt": w["BV"],
        PFX + "selfAttn_WO_weight.txt": w["WO"], PFX + "selfAttn_WO_bias.txt": w["BO"],
        PFX + "ffn_affine1_a.txt": w["a1"], PFX + "ffn_affine1_b.txt": w["b1"],
        PFX + "ffn_affine2_a.txt": w["a2"], PFX + "ffn_affine2_b.txt": w["b2"],
        PFX + "ffn_Wffn_0_bias.txt": w["Bffn0"], PFX + "ffn_Wffn_2_bias.txt": w["Bffn2"],
        "pooler_dense_weight_T.txt": w["Wp"].T, "pooler_dense_bias.txt": w["bp"],
        "fcLinear_0_weight.txt": w["fc_w"],
        "fcLinear_0_bias.txt": np.concatenate([w["fc_b"], np.zeros(108)]),   # the driver indexes 128 values (quirk Q12)
    }
    for i in (1, 2):
        for j in range(3):
            files[PFX + f"ffn_affine{i}_c{j}.txt"] = np.array([w[f"c{i}{j}"]])
    for k, blk in enumerate(lf.split_transposed_blocks(w["Wffn0"])):
        files[f"ffn_W0_transposed_block_{k}.txt"] = blk
    for k, blk in enumerate(lf.split_col_blocks(w["Wffn2"])):
        files[f"ffn_W2_block_{k}.txt"] = blk
    for name, arr in files.items():
        _write(os.path.join(wd, name), arr)
    for i in range(32):
        _write(os.path.join(root, "input", f"XE_{i}.txt"), X_E[i])
        _write(os.path.join(root, "input", f"XF_{i}.txt"), X_F[i])
    for i in range(1, x_in.shape[0]):
        _write(os.path.join(root, "tokens", f"input_{i - 1}.txt"), x_in[i])


def _run(cmd, cwd, timeout=900, extra_env=None):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fhe-linformer_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("FHELIN_SEED", None)            # keys from OS entropy, persisted through ../keys/secret-key.txt
    env.pop("FHELIN_LEVEL_PLAN", None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (cmd, r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("variant", ["main", "main_2"])
def test_full_forward_through_cpp_shim_and_reference_binary(tmp_path, variant):
    """The whole encoder1 -> pooler -> classifier sequence through include/FHEController.h in C++ (tests/shim/shim_forward.cpp:
    every read_* layout, the shim's own Chebyshev fit, 8 bootstraps) at the reference's parameters, against the clear-text
    circuit (oracle/circuit_sim.py, tolerances of tests/test_forward_gpu.py).  Where the reference tree was available at
    build time, the reference's OWN main.cpp — compiled unchanged against the shim (oracle/_ref/ref_main) — generates the
    keys (`--generate_keys`) and afterwards resumes from the encoder checkpoint this driver saved (main.cpp:105-110) to run
    pooler + classifier itself: its printed softmax must match."""
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    assert os.path.exists(FWD), "tests/shim/shim_forward missing: run __graft_entry__.build()"
    S = 129
    w = pf.synthetic_model(1234)
    x_in, X_E, X_F = pf.client_inputs(w, pf.synthetic_tokens(S, 4321))
    root = str(tmp_path)
    _write_model(root, w, x_in, X_E, X_F)
    build = os.path.join(root, "build")
    have_ref = os.path.exists(REF_MAIN) and variant == "main"
    if have_ref:
        _run([REF_MAIN, "--generate_keys"], build)                       # the reference binary writes ../keys
        _run([FWD, root, variant], build)
    else:
        _run([FWD, root, variant, "generate"], build)
    assert open(os.path.join(root, "keys", "crypto-context.txt")).read().startswith("fhelin-context 2")
    assert "seed" not in open(os.path.join(root, "keys", "crypto-context.txt")).read()
    sim, st = cs.SlotSimController(), {}
    ref = lf.forward(sim, w, x_in, X_E, X_F, st, variant)
    tol = {"scores": 5e-8, "exp": 5e-8, "self_attention": 5e-8, "affine1_0": 5e-8, "encoder_out": 1e-4, "pooled": 5e-3}
    for k, t in tol.items():
        err = np.max(np.abs(_read(os.path.join(root, "out", k + ".out")) - st[k]))
        assert err < t, (k, err)
    lg, lr = lf.logits_from_slots(_read(os.path.join(root, "out", "logits.out"))), lf.logits_from_slots(ref)
    assert np.max(np.abs(lg - lr)) < 2e-2 and int(np.argmax(lg)) == int(np.argmax(lr))
    meta = open(os.path.join(root, "out", "meta.out")).read().strip().split(",")
    assert int(meta[0]) == 8                                             # bootstraps: 2 + 5 + 1
    if have_ref:
        out = _run([REF_MAIN, "--verbose"], build)                       # load keys + checkpoint, pooler, classifier
        probs = np.array([float(l.split(":")[1]) for l in out.splitlines() if l.startswith("Softmax Prob:")])
        pred = [int(l.split(":")[1]) for l in out.splitlines() if l.startswith("Pred:")]
        e = np.exp(lg - lg.max())
        assert probs.shape == (20,) and np.max(np.abs(probs - e / e.sum())) < 5e-3
        assert pred == [int(np.argmax(lg))]


def test_cpp_driver_records_a_level_plan_and_later_runs_apply_it(tmp_path):
    """FHELIN_LEVEL_PLAN=<file> (include/FHEController.h start_level_plan / save_level_plan): the first run of the C++ driver
    records its pass and leaves the plan next to the keys, the second run of the same driver loads and applies it — same
    logits (up to noise), the V-projection inputs and the GELU bootstraps start on fewer limbs."""
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    assert os.path.exists(FWD), "tests/shim/shim_forward missing: run __graft_entry__.build()"
    w = pf.synthetic_model(1234)
    x_in, X_E, X_F = pf.client_inputs(w, pf.synthetic_tokens(129, 4321))
    root = str(tmp_path)
    _write_model(root, w, x_in, X_E, X_F)
    build = os.path.join(root, "build")
    plan_file = os.path.join(root, "keys", "level-plan.txt")
    out1 = _run([FWD, root, "main", "generate"], build, extra_env={"FHELIN_LEVEL_PLAN": plan_file})
    assert "Level plan recorded" in out1
    head, body = open(plan_file).read().split("\n", 1)
    assert head.split()[0] == "fhelin-level-plan"
    plan = [int(t) for t in body.split()]
    assert len(plan) == int(head.split()[1]) and min(t for t in plan if t > 0) < 28
    lg1 = lf.logits_from_slots(_read(os.path.join(root, "out", "logits.out")))
    out2 = _run([FWD, root, "main"], build, extra_env={"FHELIN_LEVEL_PLAN": plan_file})
    assert "applied" in out2 and "Level plan recorded" not in out2
    lg2 = lf.logits_from_slots(_read(os.path.join(root, "out", "logits.out")))
    ref = lf.logits_from_slots(lf.forward(cs.SlotSimController(), w, x_in, X_E, X_F, {}, "main"))
    assert np.max(np.abs(lg1 - ref)) < 2e-2 and np.max(np.abs(lg2 - ref)) < 2e-2
    assert int(np.argmax(lg2)) == int(np.argmax(ref))


def test_batch_of_samples_through_the_cpp_batch_controller(tmp_path):
    """include/FHEControllerBatch.h: B samples through ONE FHEController in one pass of the same C++ driver template
    (tests/shim/shim_forward.cpp batch=2; sample x reads input_b<x> / tokens_b<x>): every sample's traced intermediates and logits
    against its own clear-text circuit - the C++ twin of tests/test_batched_forward_gpu.py, at the boundary the reference binds."""
    from fhe_linformer_amd import linformer as lf
    from oracle import plain_forward as pf, circuit_sim as cs
    assert os.path.exists(FWD), "tests/shim/shim_forward missing: run __graft_entry__.build()"
    B, S = 2, 129
    w = pf.synthetic_model(1234)
    samples = [pf.client_inputs(w, pf.synthetic_tokens(S, 4321 + 17 * x)) for x in range(B)]
    root = str(tmp_path)
    _write_model(root, w, *samples[0])                       # weights + the plain single-sample layout (unused by the batch run)
    for x, (x_in, X_E, X_F) in enumerate(samples):
        os.makedirs(os.path.join(root, f"input_b{x}"), exist_ok=True)
        os.makedirs(os.path.join(root, f"tokens_b{x}"), exist_ok=True)
        for i in range(32):
            _write(os.path.join(root, f"input_b{x}", f"XE_{i}.txt"), X_E[i])
            _write(os.path.join(root, f"input_b{x}", f"XF_{i}.txt"), X_F[i])
        for i in range(1, x_in.shape[0]):
            _write(os.path.join(root, f"tokens_b{x}", f"input_{i - 1}.txt"), x_in[i])
    out = _run([FWD, root, "main", "generate", f"batch={B}"], os.path.join(root, "build"))
    assert f"forward pass of {B} samples" in out
    tol = {"scores": 5e-8, "exp": 5e-8, "self_attention": 5e-8, "affine1_0": 5e-8, "encoder_out": 1e-4, "pooled": 5e-3}
    logits = []
    for x, smp in enumerate(samples):
        st = {}
        ref = lf.forward(cs.SlotSimController(), w, *smp, st, "main")
        for k, t in tol.items():
            err = np.max(np.abs(_read(os.path.join(root, "out", f"b{x}", k + ".out")) - st[k]))
            assert err < t, (x, k, err)
        lg, lr = lf.logits_from_slots(_read(os.path.join(root, "out", f"b{x}", "logits.out"))), lf.logits_from_slots(ref)
        assert np.max(np.abs(lg - lr)) < 2e-2 and int(np.argmax(lg)) == int(np.argmax(lr)), x
        logits.append(lg)
    assert np.max(np.abs(logits[0] - logits[1])) > 1e-6      # two different samples
    meta = open(os.path.join(root, "out", "meta.out")).read().strip().split(",")
    assert int(meta[0]) == 8                                 # the driver's 8 bootstrap CALLS, each B ciphertexts wide
