"""Deferred heavy operations (fhelin_add / fhelin_bootstrap / fhelin_eval_chebyshev return handles that are evaluated, batched, when a
result is first read - csrc/capi_internal.h LazyHeavy): a batched call that throws fails ONLY its own group and the operations that read
its results; independent operations of the same flush are evaluated and read normally; the failure is reported by the read of the
failed handle (and by fhelin_sync for a flush nobody read), with the message of the call that threw.  Argument errors that can be seen
when the call is made are reported by the call.  fhelin_timer_stop evaluates what is pending (a timed region that ends on a deferred
operation includes it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_a_failing_deferred_operation_does_not_poison_independent_ones(fa):
    eng = fa.Engine("toy13", seed=4)
    try:
        eng.keygen()
        eng.gen_relin_key()
        ns = 1 << eng.params.log_slots
        v = np.random.default_rng(1).uniform(-0.9, 0.9, ns)
        cf = np.array([0.3, 0.5, -0.2, 0.1, 0.05, -0.03, 0.02, 0.01])           # degree 7: 3 + 1 levels
        x_ok = eng.encrypt(v, level=0)                                              # 7 limbs
        x_low = eng.encrypt(v, level=eng.n_q - 2)                                   # 2 limbs: passes the call-time check, runs out during evaluation
        good = eng.eval_chebyshev(x_ok, cf)
        bad = eng.eval_chebyshev(x_low, cf)
        other = eng.add(x_ok, x_ok)
        dependent = eng.add(bad, bad)                                               # reads the failing result
        # the independent results of the same flush are there
        want = np.polynomial.chebyshev.chebval(v, np.concatenate([[cf[0] / 2], cf[1:]]))
        assert np.max(np.abs(eng.decrypt(good) - want)) < 1e-6
        assert np.max(np.abs(eng.decrypt(other) - 2 * v)) < 1e-8
        with pytest.raises(fa.FhelinError) as e1:
            eng.decrypt(bad)
        assert "deferred operation failed" in str(e1.value) and "limb" in str(e1.value)
        with pytest.raises(fa.FhelinError) as e2:
            eng.decrypt(dependent)
        assert "failed earlier" in str(e2.value)
        eng.sync()                                                                  # nothing pending any more: no stale error
        # a flush nobody read: fhelin_sync reports the first failure, the good result of the same flush is still usable
        good2 = eng.eval_chebyshev(x_ok, cf)
        bad2 = eng.eval_chebyshev(x_low, cf)
        with pytest.raises(fa.FhelinError):
            eng.sync()
        assert np.max(np.abs(eng.decrypt(good2) - want)) < 1e-6
        del bad2
        # errors that are visible at the call are reported by the call
        x1 = eng.encrypt(v, level=eng.n_q - 1)                                      # one limb
        with pytest.raises(fa.FhelinError):
            eng.eval_chebyshev(x1, cf)
        with pytest.raises(fa.FhelinError):
            eng.eval_chebyshev(x_ok, cf[:1])
        # a timed region that ends on a deferred operation includes it
        eng.sync()
        eng.stats(reset=True)
        eng.timer_start()
        h = eng.eval_chebyshev(x_ok, cf)
        ms = eng.timer_stop()
        assert eng.stats()["keyswitch"] > 0 and ms > 0.0                            # evaluated inside the region, before any read
        assert np.max(np.abs(eng.decrypt(h) - want)) < 1e-6
    finally:
        eng.close()
