"""Level plan (include/fhelin.h fhelin_level_plan_*): a recorded pass of a straight-line program yields, for
every source (fresh encryption, bootstrap output), the fewest limbs it may start with; an applying pass of the same
program then runs on those.  Checked here: the derived plan on a circuit small enough to do by hand, that the applied
pass computes the same values on fewer limbs, that bootstrapping to fewer limbs keeps its precision, and that a pass
which does not follow the plan fails loudly instead of returning garbage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _circuit(eng, x, y, z):
    a = eng.encrypt(x)                       # source 0
    b = eng.encrypt(y)                       # source 1
    e = eng.encrypt(z)                       # source 2: only ever added to a value that sits far lower
    c = eng.mult(a, b)                       # one level
    d = eng.mult(c, c)                       # one more
    f = eng.add(d, e)
    return (a, b, e, f), eng.decrypt(f)      # terminal: two limbs


def test_plan_of_a_hand_checked_circuit_and_its_application(fa):
    eng = fa.Engine("toy13", seed=5)         # 7 limbs
    try:
        eng.keygen()
        eng.gen_relin_key()
        rng = np.random.default_rng(2)
        n = 1 << eng.params.log_slots
        x, y, z = (rng.uniform(-1, 1, n) for _ in range(3))
        want = (x * y) ** 2 + z
        eng.level_plan_begin("record")
        cts, got = _circuit(eng, x, y, z)
        plan = eng.level_plan_end()
        assert [c.info()["ell"] for c in cts[:3]] == [7, 7, 7]
        assert np.max(np.abs(got - want)) < 1e-6
        # decrypt needs 2 effective limbs of f = d + e; d = c*c pends a rescale (3 limbs, one pending), c likewise (4), so
        # a and b start with 4; e met d from above, so it keeps one limb for its level adjustment: 3
        assert plan == [4, 4, 3], plan
        eng.level_plan_begin("apply")
        cts2, got2 = _circuit(eng, x, y, z)
        eng.level_plan_end()
        assert [c.info()["ell"] for c in cts2[:3]] == [4, 4, 3]
        assert cts2[3].info()["ell"] - (1 if cts2[3].info()["deg"] == 2 else 0) == 2
        assert np.max(np.abs(got2 - want)) < 1e-6
        # plans travel: get -> set on the same program
        eng.set_level_plan([5, 5, -1])
        assert eng.level_plan() == [5, 5, -1]
        eng.level_plan_begin("apply")
        cts3, got3 = _circuit(eng, x, y, z)
        eng.level_plan_end()
        assert [c.info()["ell"] for c in cts3[:3]] == [5, 5, 7]
        assert np.max(np.abs(got3 - want)) < 1e-6
        # with no plan mode set, nothing changes
        cts4, _ = _circuit(eng, x, y, z)
        assert [c.info()["ell"] for c in cts4[:3]] == [7, 7, 7]
    finally:
        eng.close()


def test_plan_that_does_not_fit_the_program_fails_loudly(fa):
    eng = fa.Engine("toy13", seed=6)
    try:
        eng.keygen()
        eng.gen_relin_key()
        with pytest.raises(fa.FhelinError) as ei:
            eng.level_plan_begin("apply")                      # nothing recorded or loaded
        assert ei.value.code == 4
        n = 1 << eng.params.log_slots
        x = np.full(n, 0.5)
        eng.set_level_plan([2, 2, 2])                          # a plan of some other program: too few limbs for this one
        eng.level_plan_begin("apply")
        with pytest.raises(fa.FhelinError):
            _circuit(eng, x, x, x)
        eng.level_plan_end()
    finally:
        eng.close()


def test_deferred_rows_are_planned_through(fa):
    """rows of a deferred matmul are nodes of the recording whose level is known once they are forced (a call's results
    all depend on all of its inputs as far as the planner is concerned: the rows of one call share their level anyway)"""
    eng = fa.Engine("toy13", seed=7)
    try:
        eng.keygen()
        eng.gen_relin_key()
        eng.gen_rotation_keys(fa.circuit_rotation_indices())
        rng = np.random.default_rng(4)
        n = 1 << eng.params.log_slots
        rows = rng.uniform(-1, 1, (4, n))
        w = eng.encode(rng.uniform(-1, 1, n))

        def prog():
            cts = eng.encrypt_batch(rows)                       # sources 0..3
            out = eng.matmulRE(cts, w, None)                    # deferred rows
            return cts, eng.decrypt(out[2])                     # only row 2 is ever read

        eng.level_plan_begin("record")
        cts, want = prog()
        plan = eng.level_plan_end()
        used = cts[2].info()["ell"] - plan[2]
        assert used >= 3, plan                                  # 7 limbs were far more than this needs
        assert plan[0] == plan[1] == plan[3] == plan[2]
        eng.level_plan_begin("apply")
        cts2, got = prog()
        eng.level_plan_end()
        assert [c.info()["ell"] for c in cts2] == [plan[2]] * 4
        assert np.max(np.abs(got - want)) < 1e-5
    finally:
        eng.close()


def test_bootstrap_to_fewer_limbs_keeps_its_precision(fa):
    eng = fa.Engine("boot12", seed=77, log_slots=10)
    try:
        eng.keygen()
        eng.gen_relin_key()
        eng.bootstrap_setup(3, 3, 1 << 10)
        n = 1 << 10
        m = np.random.default_rng(3).uniform(-1, 1, n)

        def prog():
            ct = eng.encrypt(m, level=eng.n_q - 3)             # source 0
            return eng.bootstrap(ct)                            # source 1

        full = prog()
        ell_full = full.info()["ell"]
        err_full = np.max(np.abs(eng.decrypt(full) - m))
        eng.set_level_plan([-1, ell_full - 2])                  # the circuit after the bootstrap is known to leave two limbs unused
        eng.level_plan_begin("apply")
        low = prog()
        eng.level_plan_end()
        assert low.info()["ell"] == ell_full - 2
        err_low = np.max(np.abs(eng.decrypt(low) - m))
        assert err_full < 2e-4 and err_low < 2e-4, (err_full, err_low)
        sq = eng.mult(low, low)
        assert np.max(np.abs(eng.decrypt(sq) - m * m)) < 1e-3
    finally:
        eng.close()


def test_handles_from_an_earlier_recording_are_plain_inputs_of_a_later_one(fa):
    """a handle may outlive the recording pass that made it (a driver keeping a ciphertext between passes, deferred rows
    forced later): in a later recording it is an input with no history, not an index into a recording that is gone"""
    eng = fa.Engine("toy13", seed=9)
    try:
        eng.keygen()
        eng.gen_relin_key()
        eng.gen_rotation_keys(fa.circuit_rotation_indices())
        rng = np.random.default_rng(6)
        n = 1 << eng.params.log_slots
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        w = eng.encode(rng.uniform(-1, 1, n))
        eng.level_plan_begin("record")
        old = eng.encrypt(x)                                    # node 0 of the first recording
        for _ in range(40):                                     # ... which grows far beyond the second one
            old2 = eng.add(old, old)
        lazy = eng.matmulRE(eng.encrypt_batch(rng.uniform(-1, 1, (3, n))), w, None)   # deferred rows, not forced yet
        eng.level_plan_end()
        eng.level_plan_begin("record")
        b = eng.encrypt(y)                                      # source 0 of THIS recording
        c = eng.mult(old2, b)                                   # old2: a stale node id (41) in a recording of 2 nodes
        got = eng.decrypt(c)
        eng.decrypt(lazy[1])                                    # rows recorded in the first pass are forced in the second
        plan = eng.level_plan_end()
        assert np.max(np.abs(got - 2 * x * y)) < 1e-6
        assert plan == [3]                                      # mult (1 level) + decrypt (2 limbs); `old2` puts no constraint
    finally:
        eng.close()
