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
