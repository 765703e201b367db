"""Client-side work on the GPU: the device CKKS encoder must produce the residues of the
host encoder bit for bit (special FFT in fp64 with the same operation order, the x87 scaling product emulated exactly);
the device sampler must be ternary / rounded Gaussian with the right moments; batched encryption must decrypt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,over,ells", [
    ("bench", {}, [24, 3]),                       # sparse packing: 16384 slots in N=2^16
    ("reference", {}, [28, 1]),                   # full packing: 16384 slots in N=2^15
    ("toy13", {}, [7]),
    ("toy13", dict(log_slots=3), [5]),            # 8 slots
])
def test_device_encoder_equals_host_encoder(fa, preset, over, ells):
    eng = fa.Engine(preset, seed=2, **over)
    try:
        ns = 1 << eng.params.log_slots
        rng = np.random.default_rng(1)
        vecs = [rng.uniform(-1, 1, ns), rng.normal(0, 1e-6, ns), rng.uniform(-300, 300, ns), np.zeros(ns),
                np.where(np.arange(ns) % 128 == 0, 1.0, 0.0), np.full(ns, -0.5)]
        sf = eng.scaling_factors
        for ell in ells:
            lvl = eng.n_q - ell
            scales = [0.0, float(sf[lvl]) * float(sf[lvl]), float(sf[lvl]) * 0.7310585786300049]   # Delta, ~2^104, odd factor
            for v in vecs:
                for sc in scales:
                    eng.set_host_encode(True)
                    want = eng.pt_export(eng.encode(v), ell, sc)
                    eng.set_host_encode(False)
                    got = eng.pt_export(eng.encode(v), ell, sc)
                    assert np.array_equal(got, want), (preset, ell, sc)
    finally:
        eng.close()


def test_device_sampler_distributions(fa):
    eng = fa.Engine("bench", seed=0)              # OS-entropy key: the moments must hold for any key
    try:
        t = eng.debug_sample(1, 4)
        assert set(np.unique(t)) == {-1, 0, 1}
        for p in t:
            c = np.bincount(p + 1, minlength=3) / p.size
            assert np.all(np.abs(c - 1 / 3) < 0.01)
        assert not np.array_equal(t[0], t[1])                      # streams differ per polynomial
        g = eng.debug_sample(0, 4).astype(np.float64)
        assert abs(g.mean()) < 0.03 and abs(g.std() - np.sqrt(3.19 ** 2 + 1 / 12)) < 0.03 and np.abs(g).max() < 30
        g2 = eng.debug_sample(0, 1).astype(np.float64)
        assert not np.array_equal(g[0], g2[0])                     # a fresh key per call
    finally:
        eng.close()


@pytest.mark.parametrize("preset", ["bench", "reference"])
def test_encrypt_batch_decrypts(fa, preset):
    eng = fa.Engine(preset, seed=9)
    try:
        eng.keygen()
        ns = 1 << eng.params.log_slots
        rows = np.random.default_rng(3).uniform(-1, 1, (37, 200))    # ragged: 200 values per row, 37 rows (> one chunk of 32)
        cts = eng.encrypt_batch(rows, level=0)
        assert len(cts) == 37
        for i in (0, 17, 36):
            got = eng.decrypt(cts[i])
            assert np.max(np.abs(got[:200] - rows[i])) < 1e-9 and np.max(np.abs(got[200:])) < 1e-9
            assert cts[i].info()["ell"] == eng.n_q and cts[i].info()["slots"] == ns
        a, b = cts[0].export(), cts[1].export()
        assert not np.array_equal(a[1], b[1])                       # independent randomness per ciphertext
        low = eng.encrypt_batch(rows[:2], level=eng.n_q - 2)
        assert low[0].info()["ell"] == 2 and np.max(np.abs(eng.decrypt(low[1])[:200] - rows[1])) < 1e-8
        single = eng.encrypt(rows[5])                               # the one-input path uses the same device kernels
        assert np.max(np.abs(eng.decrypt(single)[:200] - rows[5])) < 1e-9
    finally:
        eng.close()


def test_device_ingestion_of_a_sample_matches_the_numpy_statement(fa):
    """fhelin_client_ingest: embedding gather + positional embedding + the Linformer projections X_E = E x + b_E,
    X_F = F x + b_F (reference src/python/dimReduce.py:141-160) + the driver's 64 + S + 1 read_expanded_input calls
    (src/main.cpp:159-173) on the device.  Floating point (fp64): the device's sums run in the stated order (t = 0..S, no FMA), so
    x_in and the projections are BIT-IDENTICAL to the same order in NumPy and within 1e-13 of NumPy's BLAS product (stated
    tolerance: summation order only); the ciphertexts decrypt to the expanded packing of those rows to encoder precision (1e-9);
    token ids into a table give the same ciphertext values as the gathered embeddings."""
    from oracle import plain_forward as pf
    from fhe_linformer_amd import linformer as lf
    w = pf.synthetic_model(1234)
    S = 37
    x = pf.synthetic_tokens(S, 4321)
    eng = fa.Engine("reference", seed=5, n_q=4, n_p=2, dnum=2)
    try:
        eng.keygen()
        got = eng.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], emb=x, level=1, want_proj=True)
        x_in, X_E, X_F = pf.client_inputs(w, x)                               # the reference statement (BLAS matmul)
        assert np.array_equal(got["x_in"], x_in)                             # element-wise: bit-identical
        seq = np.zeros((64, 128))
        for r in range(64):
            W, b = (w["E_w"], w["E_b"]) if r < 32 else (w["F_w"], w["F_b"])
            acc = W[r % 32, 0] * x_in[0]
            for t in range(1, S + 1):
                acc = acc + W[r % 32, t] * x_in[t]
            seq[r] = acc + b[r % 32]
        assert np.array_equal(got["proj"], seq)                              # the stated summation order: bit-identical
        assert np.max(np.abs(got["proj"] - np.vstack([X_E, X_F]))) < 1e-13   # vs the BLAS product
        assert len(got["inputs_E"]) == len(got["inputs_F"]) == 32 and len(got["inputs"]) == S + 1
        for ct, row in ((got["inputs_E"][3], X_E[3]), (got["inputs_F"][31], X_F[31]), (got["inputs"][0], x_in[0]), (got["inputs"][S], x_in[S])):
            assert ct.info()["ell"] == 3 and ct.info()["slots"] == 16384
            assert np.max(np.abs(eng.decrypt(ct) - lf.expanded(row))) < 1e-9
        # token ids into an embedding table: the same rows
        table = np.random.default_rng(2).normal(0, 0.3, (50, 128))
        tok = np.random.default_rng(3).integers(0, 50, S)
        a = eng.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], tokens=tok, table=table, want_proj=True)
        b = eng.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], emb=table[tok], want_proj=True)
        assert np.array_equal(a["x_in"], b["x_in"]) and np.array_equal(a["proj"], b["proj"])
        with pytest.raises(fa.FhelinError):
            eng.client_ingest(w["cls_token"], w["posEmb"], w["E_w"], w["E_b"], w["F_w"], w["F_b"], tokens=[0, 50], table=table)
    finally:
        eng.close()
