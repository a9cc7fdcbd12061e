"""SURVEY.md §8(e) with PRODUCT code on real ranks: 2, 3, 4 and 8 processes, one rank each, sharded by point
(ceres-solver_amd/partition.py), camera-space sums combined by the library's one-shot peer-to-peer all-reduce over
hipIpc-mapped buffers.  The GPU box has one MI355X, so all ranks run on device 0 — the same code path (hipIpc
mapping, system-scope flags, rank-ordered sums) that runs over xGMI between the GPUs of a node.  Rung (5) of the
parity ladder: multi-GPU vs 1-GPU / oracle, same tolerances (sums are re-associated again)."""
import multiprocessing as mp

import numpy as np
import pytest

from step_check import assert_lm_style_step
from test_gpu_operators import rel

pytestmark = pytest.mark.gpu

WORLDS = (2, 3, 4, 8)   # the peer-to-peer all-reduce indexes [parity][source rank][cap] slots for world <= 8 (kernels_cg.hip)


def run_ranks(scenario, WORLD=2, timeout=420):
    from multirank_worker import run_rank
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(WORLD)]
    procs = [ctx.Process(target=run_rank, args=(r, WORLD, pipes[r][1], 0, scenario)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [None] * WORLD
    try:
        while any(r is None for r in results):
            msgs = []
            for r in range(WORLD):
                if results[r] is not None:
                    continue
                if not pipes[r][0].poll(timeout):
                    raise TimeoutError(f"rank {r} sent nothing for {timeout} s")
                msgs.append((r, pipes[r][0].recv()))
            kinds = {m[1][0] for m in msgs}
            errors = [(r, payload) for r, (kind, payload) in msgs if kind == "error"]
            if errors:   # every rank's own story: the first to report is often the one that waited for a peer that had failed
                raise AssertionError("\n".join(f"rank {r} failed:\n{payload}" for r, payload in errors))
            assert len(kinds) == 1, f"ranks out of step: {kinds}"
            if kinds == {"exchange"}:
                handles = [m[1][1] for m in sorted(msgs)]
                for r, _ in msgs:
                    pipes[r][0].send(handles)
            else:
                for r, (_, payload) in msgs:
                    results[r] = payload
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()  # the exact process we started
    return results


def assemble(partition_mod, recs, num_cols, key):
    x = np.full(num_cols, np.nan)
    for rec in recs:
        x[rec["col_index"][: rec["n_e"]]] = rec[key][0][: rec["n_e"]]
    x[recs[0]["col_index"][recs[0]["n_e"]:]] = recs[0][key][0][recs[0]["n_e"]:]
    return x


def make_bal_case(hip, oracle, problems, nc, npts, nobs, structured=None, solvers=None):
    """A problem of the sharded BAL tests and everything the oracle says about it (computed once for all world sizes)."""
    kw = dict(kind="bal", seed=31, nc=nc, np=npts, no=nobs, skew=0.5,
              solvers=solvers or [(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)])
    if structured:
        kw["structured"] = structured
        p = problems.synthetic_structured(nc, npts, nobs, seed=31, skew=0.5, **structured)
    else:
        p = problems.synthetic_bal(None, layout="schur", seed=31, skew=0.5, num_cameras=nc, num_points=npts, num_observations=nobs)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    m0 = oracle.Matrix(p.bs, 0)
    ref = {}
    for solver_type, pre in kw["solvers"]:
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        ref[(solver_type, "solve")] = lambda lo, hi, q, r, fn=fn, pre=pre: fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r)
        ref[(solver_type, "converged")] = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=400, q_tol=-1.0, r_tol=1e-12)
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    xf = np.random.default_rng(5).standard_normal(m.num_cols_f)
    ref["rhs"], ref["sx"], ref["precond"] = isc.rhs().copy(), isc.sx(xf).copy(), m.schur_jacobi(p.values, p.D)[0]
    xx = np.random.default_rng(6).standard_normal(p.bs.num_cols)
    ref["jtjx"] = m0.left_multiply(p.values, m0.right_multiply(p.values, xx)) + p.D ** 2 * xx
    ref["jtb"] = m0.left_multiply(p.values, p.b)
    return kw, p, m0, ref


@pytest.fixture(scope="module")
def bal_case(hip, oracle, problems):
    return make_bal_case(hip, oracle, problems, 37, 6000, 26000)


@pytest.fixture(scope="module")
def many_camera_case(hip, oracle, problems):
    # 2600 cameras: more than LDS holds — every rank builds a hybrid plan for its shard (popular cameras + windows, the rest spilled),
    # CGNR runs on internally numbered points, and the step's merged all-reduce is 99 x 2600 doubles = 126 chunks of the one-shot kernel
    return make_bal_case(hip, oracle, problems, 2600, 20000, 90000)


@pytest.fixture(scope="module")
def quaternion_case(hip, oracle, problems):
    # <2,3,10> cameras (bundle_adjuster --use_quaternions): both solvers sharded on the fused path of that shape
    return make_bal_case(hip, oracle, problems, 37, 6000, 26000, structured=dict(camera_width=10))


@pytest.fixture(scope="module")
def libmv_like_case(hip, oracle, problems):
    # shared intrinsics + 6-wide poses + a constant camera, sharded by point: the shared block's preconditioner block and the strip's sums
    # are all-reduced with the cameras' (ITERATIVE_SCHUR; sharded CGNR needs points-then-cameras columns, which a shared block breaks)
    return make_bal_case(hip, oracle, problems, 37, 6000, 26000, structured=dict(camera_width=6, shared_widths=(8,), locked_cameras=(0,), shared_first=False),
                         solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)])


@pytest.fixture(scope="module")
def homogeneous_points_case(hip, oracle, problems):
    # (2,4,9): 4-wide point blocks (round 5), ITERATIVE_SCHUR sharded by point on the fused path of that shape
    return make_bal_case(hip, oracle, problems, 37, 6000, 26000, structured=dict(camera_width=9, point_width=4), solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)])


@pytest.fixture(scope="module")
def three_residual_rows_case(hip, oracle, problems):
    # (3,3,3): rows of three residuals (round 5)
    return make_bal_case(hip, oracle, problems, 37, 6000, 26000, structured=dict(camera_width=3, point_width=3, row_height=3), solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)])


def test_sharded_homogeneous_points_against_the_oracle(hip, homogeneous_points_case):
    check_sharded_case(hip, homogeneous_points_case, 2)


def test_sharded_rows_of_three_residuals_against_the_oracle(hip, three_residual_rows_case):
    check_sharded_case(hip, three_residual_rows_case, 3)


@pytest.mark.parametrize("world", (2, 4))
def test_sharded_quaternion_cameras_against_the_oracle(hip, quaternion_case, world):
    check_sharded_case(hip, quaternion_case, world)


@pytest.mark.parametrize("world", (2, 4))
def test_sharded_libmv_structure_against_the_oracle(hip, libmv_like_case, world):
    check_sharded_case(hip, libmv_like_case, world)


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_bal_both_solvers_against_the_oracle(hip, bal_case, world):
    check_sharded_case(hip, bal_case, world)


@pytest.mark.parametrize("world", (2, 4))
def test_sharded_many_camera_regime_against_the_oracle(hip, many_camera_case, world):
    check_sharded_case(hip, many_camera_case, world)


def check_sharded_case(hip, case, world):
    kw, p, m0, ref = case
    res = run_ranks([("bal", kw)], world)
    for solver_type, pre in kw["solvers"]:
        recs = [res[r][("bal", solver_type, pre)] for r in range(world)]
        assert all(rec["path"] == hip.PATH_BAL and rec["world"] == world for rec in recs) and [rec["rank"] for rec in recs] == list(range(world))
        # converged solve
        xo, so = ref[(solver_type, "converged")]
        assert all(rec["converged"][1] == hip.SUCCESS for rec in recs)
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-8
        # the camera part is REPLICATED: identical bits on every rank (the all-reduce sums in rank order everywhere)
        for key in ("converged", "lm_style", "lm_step", "retry"):
            a = recs[0][key][0][recs[0]["n_e"]:]
            for rec in recs[1:]:
                assert np.array_equal(a, rec[key][0][rec["n_e"]:]), key
                assert recs[0][key][2] == rec[key][2]   # same iteration count on every rank
                assert recs[0][key][4] == rec[key][4]   # and the same message (zeta, |r|): the replicated CG state is bit-identical
        # LM-style call (eta = 0.1): unconditional comparison with the oracle's CG sequence (tests/step_check.py)
        class S:  # the summary every rank reported
            termination_type, num_iterations, message = recs[0]["lm_style"][1], recs[0]["lm_style"][2], recs[0]["lm_style"][4]
        xo, so = assert_lm_style_step(assemble(None, recs, p.bs.num_cols, "lm_style"), S, ref[(solver_type, "solve")], 0.1, hip.SUCCESS)
        # the LM step on the device: step = -(solve), model cost change summed over ranks and equal on all of them
        step = assemble(None, recs, p.bs.num_cols, "lm_step")
        assert all(rec["lm_step"][3] == recs[0]["lm_step"][3] for rec in recs) and recs[0]["lm_step"][3] > 0
        S.termination_type, S.num_iterations, S.message = recs[0]["lm_step"][1], recs[0]["lm_step"][2], recs[0]["lm_step"][4]
        assert_lm_style_step(-step, S, ref[(solver_type, "solve")], 0.1, hip.SUCCESS)
        Jx = m0.right_multiply(p.values, step)   # the model cost change belongs to the step the ranks produced, whatever its index
        assert abs(recs[0]["lm_step"][3] - (-(Jx @ (p.b + Jx / 2)))) <= 1e-9 * abs(recs[0]["lm_step"][3])
        # collectives of that step (info.collectives_last_step): the merged per-step sum, the operator's camera vector per CG iteration
        # (CGNR: + the scalars' all-reduce), {finite flag, model cost}; the same on every rank
        its, per_it = recs[0]["lm_step"][2], (1 if solver_type == hip.ITERATIVE_SCHUR else 2)
        assert all(rec["collectives"] == recs[0]["collectives"] for rec in recs)
        assert per_it * its + 2 <= recs[0]["collectives"] <= per_it * (its + 1) + 6, (recs[0]["collectives"], its)
        if solver_type == hip.ITERATIVE_SCHUR:
            for rec in recs:
                assert rel(rec["rhs"], ref["rhs"]) <= 1e-12 and rel(rec["sx"], ref["sx"]) <= 1e-12 and rel(rec["precond"], ref["precond"]) <= 1e-11
        else:
            for rec in recs:
                ci = rec["col_index"]
                assert rel(rec["jtjx"], ref["jtjx"][ci]) <= 1e-12 and rel(rec["jtb"], ref["jtb"][ci]) <= 1e-12


@pytest.mark.parametrize("WORLD", (2, 3))
def test_sharded_generic_structure(hip, oracle, problems, WORLD):
    kw = dict(kind="general", seed=6, ne=40, nf=7, solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)], max_it=3000)
    res = run_ranks([("general", kw)], WORLD)
    p = problems.random_schur_problem(num_e_blocks=40, num_f_blocks=7, num_no_e_rows=2, seed=6)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    m0 = oracle.Matrix(p.bs, 0)
    for solver_type, pre in kw["solvers"]:
        recs = [res[r][("general", solver_type, pre)] for r in range(WORLD)]
        assert all(rec["path"] == hip.PATH_GENERIC for rec in recs)
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=3000, q_tol=-1.0, r_tol=1e-12)
        assert all(rec["converged"][1] == hip.SUCCESS for rec in recs), [rec["converged"][1:] for rec in recs]
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-7


@pytest.mark.parametrize("WORLD", (2, 4))
def test_sharded_cgnr_fused_iteration_equals_the_six_kernel_iteration(hip, problems, WORLD):
    """Sharded CGNR on the <2,3,9> path: the two-launch iteration (the shard's p.q share travels with the camera vector through
    the operator's all-reduce, r.z / Q1 / |r|^2 of the shard in one 4-double all-reduce) against the six-kernel sequence with
    its four scalar all-reduces per iteration: same iteration counts, same termination, solutions equal to rounding — for a
    converged solve (incl. residual resets every 10th iteration) and for the LM step."""
    base = dict(kind="bal", seed=77, nc=45, np=5000, no=23000, skew=0.4, solvers=[(hip.CGNR, hip.JACOBI)], max_it=400)
    res = run_ranks([("fused", dict(base, cg_fused="1")), ("unfused", dict(base, cg_fused="0"))], WORLD)
    for r in range(WORLD):
        a, b = res[r][("fused", hip.CGNR, hip.JACOBI)], res[r][("unfused", hip.CGNR, hip.JACOBI)]
        for key in ("converged", "lm_style", "lm_step"):
            assert a[key][1] == b[key][1] == hip.SUCCESS and a[key][2] == b[key][2], (key, a[key][1:3], b[key][1:3])
            assert rel(a[key][0], b[key][0]) <= 1e-10, key
        assert a["converged"][2] > 12   # long enough to contain a residual reset
        assert abs(a["lm_step"][3] - b["lm_step"][3]) <= 1e-10 * abs(b["lm_step"][3])
    # and the ranks agree with each other on the replicated part, bit for bit, in the fused mode too
    a0 = res[0][("fused", hip.CGNR, hip.JACOBI)]
    for r in range(1, WORLD):
        a1 = res[r][("fused", hip.CGNR, hip.JACOBI)]
        assert np.array_equal(a0["converged"][0][a0["n_e"]:], a1["converged"][0][a1["n_e"]:])


@pytest.mark.parametrize("WORLD", (2, 3))
def test_sharded_bal_with_leftover_rows(hip, oracle, problems, WORLD):
    """Rows without a point cell (camera priors) in a sharded run of the fused path: partition.py gives them to the last rank, whose
    generic kernels add their sums BEFORE the camera-space all-reduce (csrc/solver.hip: add_remainder)."""
    kw = dict(kind="bal", seed=41, nc=29, np=3000, no=14000, skew=0.5, camera_rows=40,
              solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)])
    res = run_ranks([("bal", kw)], WORLD)
    p = problems.add_camera_rows(problems.synthetic_bal(None, layout="schur", seed=41, skew=0.5, num_cameras=29, num_points=3000, num_observations=14000),
                                 40, seed=41, pair_fraction=0.3)
    m, m0 = oracle.Matrix(p.bs, p.num_eliminate_blocks), oracle.Matrix(p.bs, 0)
    for solver_type, pre in kw["solvers"]:
        recs = [res[r][("bal", solver_type, pre)] for r in range(WORLD)]
        assert all(rec["path"] == hip.PATH_BAL for rec in recs)
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        solve = lambda lo, hi, q, r: fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r)
        xo, so = solve(0, 400, -1.0, 1e-12)
        assert all(rec["converged"][1] == hip.SUCCESS for rec in recs)
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-8
        # the camera part and the CG scalars are REPLICATED also when one rank holds rows the others do not: identical bits, counts and
        # messages on every rank (round 6: the rank with the leftover rows summed p . q in another kernel than its peers — a last-bit
        # difference in a replicated scalar lets one rank leave CG an iteration before the others; tools/fuzz_multirank.py)
        for key in ("converged", "lm_style", "lm_step"):
            a = recs[0][key][0][recs[0]["n_e"]:]
            for rec in recs[1:]:
                assert np.array_equal(a, rec[key][0][rec["n_e"]:]), key
                assert recs[0][key][2] == rec[key][2] and recs[0][key][4] == rec[key][4], (key, recs[0][key][1:], rec[key][1:])

        class S:
            termination_type, num_iterations, message = recs[0]["lm_style"][1], recs[0]["lm_style"][2], recs[0]["lm_style"][4]
        assert_lm_style_step(assemble(None, recs, p.bs.num_cols, "lm_style"), S, solve, 0.1, hip.SUCCESS)
        step = assemble(None, recs, p.bs.num_cols, "lm_step")
        Jx = m0.right_multiply(p.values, step)
        assert abs(recs[0]["lm_step"][3] - (-(Jx @ (p.b + Jx / 2)))) <= 1e-9 * abs(recs[0]["lm_step"][3])
        if solver_type == hip.ITERATIVE_SCHUR:
            isc = oracle.ImplicitSchurComplement(m)
            isc.init(p.values, p.D, p.b)
            xf = np.random.default_rng(5).standard_normal(m.num_cols_f)
            inv, _ = m.schur_jacobi(p.values, p.D)
            for rec in recs:
                assert rel(rec["rhs"], isc.rhs()) <= 1e-12 and rel(rec["sx"], isc.sx(xf)) <= 1e-12 and rel(rec["precond"], inv) <= 1e-10
        else:
            xx = np.random.default_rng(6).standard_normal(p.bs.num_cols)
            want = m0.left_multiply(p.values, m0.right_multiply(p.values, xx)) + p.D ** 2 * xx
            g = m0.left_multiply(p.values, p.b)
            for rec in recs:
                ci = rec["col_index"]
                assert rel(rec["jtjx"], want[ci]) <= 1e-12 and rel(rec["jtb"], g[ci]) <= 1e-12


def test_a_rank_that_leaves_mid_run_is_an_error_not_a_hang(hip):
    """One of three ranks goes away after the communicator is connected; the other two enter an LM step.  Their all-reduces wait
    CERES_HIP_P2P_TIMEOUT seconds for the missing rank ONCE, poison their output with NaN, and the step returns CERES_HIP_E_COMM
    at its next poll — within a few timeouts, never a hung GPU (the collective is a spin-waiting kernel)."""
    kw = dict(kind="bal", seed=31, nc=20, np=1500, no=7000, skew=0.5, solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)],
              drop_rank=2, p2p_timeout=2)
    res = run_ranks([("drop", kw)], 3, timeout=180)
    for solver in kw["solvers"]:
        assert res[2][("drop",) + solver]["dropped"]
        for r in (0, 1):
            rec = res[r][("drop",) + solver]
            assert rec["p2p_enabled"] == 1
            assert rec["error"] is not None and "timed out" in rec["error"] and f"error {hip.E_COMM}" in rec["error"], rec
            assert rec["seconds"] < 30.0, rec


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("what", ["nan", "singular_point"])
@pytest.mark.parametrize("WORLD", [2, 4])
def test_a_shard_that_cannot_be_solved_ends_the_call_alike_on_every_rank(hip, oracle, problems, WORLD, what, generic):
    """One rank's shard holds a NaN (or a point whose block is singular: zeroed E cells, no D).  The ranks of one call must agree about
    how it ended — one rank reporting FAILURE while its peers return a "successful" step computed without that rank's sums would send
    the ranks of the caller's loop different ways — and nobody may wait for a peer that has already left the solve.  Afterwards the same
    instances solve the unpoisoned problem, against the oracle.  (On the generic kernels the verdict on the point blocks was a read-back
    of the rank's own flag: the rank with the singular block left with "E^T E + D^2 is not positive definite", its peer waited out an
    all-reduce, and the exchange epochs of the two stayed apart — every later call failed.  tools/probes/poison_ranks.py.)"""
    kw = dict(kind="bal", seed=31, nc=20, np=1500, no=7000, skew=0.5, solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)],
              poison=(WORLD - 1 if what == "nan" else 0, what), p2p_timeout=5, force_generic=generic)
    res = run_ranks([("poison", kw)], WORLD, timeout=180)
    p = problems.synthetic_bal(None, layout="schur", seed=31, skew=0.5, num_cameras=20, num_points=1500, num_observations=7000)
    for solver in kw["solvers"]:
        recs = [res[r][("poison",) + solver] for r in range(WORLD)]
        for key in ("bad_solve", "bad_step"):
            kinds = {rec[key][0] for rec in recs}
            assert len(kinds) == 1, (solver, key, [rec[key][:3] for rec in recs])
            assert kinds != {"error"}, (solver, key, recs[0][key])
            assert len({rec[key][1] for rec in recs}) == 1, (solver, key, [rec[key][:3] for rec in recs])
            if generic:   # the early returns of the generic path: every rank leaves at the same check
                assert len({rec[key][2] for rec in recs}) == 1, (solver, key, [rec[key][:3] for rec in recs])
            if what == "nan":   # a NaN cannot end in a finite "successful" step anywhere
                assert not any(rec[key][0] == hip.SUCCESS and rec[key][3] for rec in recs), (solver, key, [rec[key][:3] for rec in recs])
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks if solver[0] == hip.ITERATIVE_SCHUR else 0)
        fn = m.iterative_schur_solve if solver[0] == hip.ITERATIVE_SCHUR else m.cgnr_solve
        solve = lambda lo, hi, q, r: fn(p.values, p.b, p.D, preconditioner=solver[1], min_it=lo, max_it=hi, q_tol=q, r_tol=r)
        x = assemble(None, recs, p.bs.num_cols, "good_solve")
        summ = type("S", (), dict(termination_type=recs[0]["good_solve"][1], num_iterations=recs[0]["good_solve"][2], message=recs[0]["good_solve"][4]))
        assert_lm_style_step(x, summ, solve, 0.1, hip.SUCCESS, 1e-9)


def _variants(hip):
    K = dict(min_num_iterations=3, max_num_iterations=3)
    out = []
    for generic in (False, True):
        g = dict(force_generic_path=generic)
        out += [dict(type=hip.DENSE_SCHUR, max_num_iterations=1, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, use_explicit_schur_complement=True, **K, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_POWER_SERIES_EXPANSION, max_num_spse_iterations=5, spse_tolerance=0.1, **K, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, use_spse_initialization=True, max_num_spse_iterations=5, spse_tolerance=0.1, **K, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.JACOBI, **K, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.IDENTITY, **K, **g),
                dict(type=hip.CGNR, preconditioner_type=hip.IDENTITY, **K, **g),
                dict(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, residual_reset_period=1, min_num_iterations=0, max_num_iterations=50, r_tolerance=1e-6, **g)]
    return out


@pytest.mark.parametrize("WORLD", [2, 3])
def test_sharded_solver_options_equal_the_single_rank_instance(hip, problems, WORLD):
    """DENSE_SCHUR, the explicit Schur complement, SCHUR_POWER_SERIES_EXPANSION as preconditioner and as initialisation, JACOBI and
    IDENTITY, a residual reset every iteration — sharded by point, each against ONE instance on the whole problem with the same options
    (which tests/test_gpu_explicit_schur.py, test_gpu_spse.py and the option campaign of tools/fuzz_parity.py hold against the oracle)."""
    variants = _variants(hip)
    kw = dict(kind="bal", seed=33, nc=20, np=1500, no=7000, skew=0.5, solvers=[], variants=variants, p2p_timeout=10)
    res = run_ranks([("opts", kw)], WORLD, timeout=240)
    p = problems.synthetic_bal(None, layout="schur", seed=33, skew=0.5, num_cameras=20, num_points=1500, num_observations=7000)
    for vi, var in enumerate(variants):
        var = dict(var)
        q_tol, r_tol = var.pop("q_tolerance", -1.0), var.pop("r_tolerance", -1.0)
        recs = [res[r][("opts", "variant", vi)] for r in range(WORLD)]
        one = hip.HipLinearSolver(hip.LinearSolverOptions(elimination_groups=[p.num_eliminate_blocks], **var))
        one.set_structure(p.bs)
        xo, so = one.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
        one.close()
        errors = [rec.get("error") for rec in recs]
        assert not any(errors), (var, errors)
        assert all((rec["x"][1], rec["x"][2]) == (so.termination_type, so.num_iterations) for rec in recs), (var, [rec["x"][1:] for rec in recs], so)
        x = np.full(p.bs.num_cols, np.nan)
        for rec in recs:
            x[rec["col_index"][: rec["n_e"]]] = rec["x"][0][: rec["n_e"]]
        x[recs[0]["col_index"][recs[0]["n_e"]:]] = recs[0]["x"][0][recs[0]["n_e"]:]
        for rec in recs[1:]:   # the camera part is replicated
            assert np.array_equal(recs[0]["x"][0][recs[0]["n_e"]:], rec["x"][0][rec["n_e"]:]), var
        nx, no = np.isnan(x), np.isnan(xo)
        assert np.array_equal(nx, no), (var, int(nx.sum()), int(no.sum()))   # (NO_CONVERGENCE at the cap: no back-substitution, the point part stays NaN)
        assert rel(x[~nx], xo[~no]) <= 1e-9, (var, rel(x[~nx], xo[~no]))


@pytest.mark.parametrize("seed,WORLD,pk", [
    (52, 3, dict(num_e_blocks=40, num_f_blocks=2, max_rows_per_e=1, num_no_e_rows=3, static_sizes=(2, 3, 6), seed=52)),
    (83, 8, dict(num_e_blocks=8, num_f_blocks=2, max_rows_per_e=1, num_no_e_rows=0, static_sizes=None, seed=83))])
def test_ranks_agree_on_the_kernel_path(hip, oracle, problems, seed, WORLD, pk):
    """A shard of a GENERAL structure can look like bundle adjustment — each of its rows one point cell and one camera cell — while its
    neighbour's does not; the two kernel paths issue different sequences of exchanges, and ranks that chose for themselves waited for
    each other until the time-out (tools/fuzz_multirank.py --generic, seeds 52 and 83 among nine of 149).  ceres_hip_set_structure now
    agrees on the path (and on the fused shape) over the ranks."""
    solvers = [(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)]
    kw = dict(kind="general_fuzz", problem=pk, solvers=solvers, radius=1.0, max_it=3000, p2p_timeout=8)
    res = run_ranks([("g", kw)], WORLD, timeout=120)
    p = problems.random_schur_problem(**pk)
    m, m0 = oracle.Matrix(p.bs, p.num_eliminate_blocks), oracle.Matrix(p.bs, 0)
    for solver_type, pre in solvers:
        recs = [res[r][("g", solver_type, pre)] for r in range(WORLD)]
        assert len({rec["path"] for rec in recs}) == 1, [rec["path"] for rec in recs]
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=3000, q_tol=-1.0, r_tol=1e-12)
        assert all(rec["converged"][1] == so.termination_type for rec in recs), [rec["converged"][1:] for rec in recs]
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-7
