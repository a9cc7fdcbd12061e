"""the neighbours of Solve inside one trust-region step, kept on the device
(LM diagonal from diag(J^T J), solve, finite check + negation, model cost change, ScaleColumns),
against the oracle and the formulas of levenberg_marquardt_strategy.cc:84-132 and
trust_region_minimizer.cc:420-438."""
import numpy as np
import pytest

from step_check import assert_lm_style_step
from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


def reference_step(oracle, hip, p, solver_type, pre, radius, eta, diag=None, max_it=500):
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks if solver_type == hip.ITERATIVE_SCHUR else 0)
    if diag is None:
        diag = np.clip(m.squared_column_norm(p.values), 1e-6, 1e32)
    D = np.sqrt(diag / radius)
    fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m.cgnr_solve
    x, s = fn(p.values, p.b, D, preconditioner=pre, min_it=0, max_it=max_it, q_tol=eta, r_tol=-1.0)
    step = -x
    model = oracle.Matrix(p.bs, 0).right_multiply(p.values, step)
    return step, s, -model @ (p.b + model / 2.0), D, diag


def check_step(oracle, hip, p, solver_type, pre, D, step, summ, mcc, eta, tol=1e-9):
    """Unconditional (tests/step_check.py): the step is the negated oracle iterate of the product's own iteration count (counts within
    one of each other), and the model cost change is the one of that iterate."""
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks if solver_type == hip.ITERATIVE_SCHUR else 0)
    fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m.cgnr_solve
    solve = lambda lo, hi, q, r: fn(p.values, p.b, D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r)
    xo, so = assert_lm_style_step(-step, summ, solve, eta, hip.SUCCESS, tol)
    if so.num_iterations != summ.num_iterations:
        xo, _ = solve(summ.num_iterations, summ.num_iterations, -1.0, -1.0)
    model = oracle.Matrix(p.bs, 0).right_multiply(p.values, -xo)
    want = -model @ (p.b + model / 2.0)
    assert abs(mcc - want) <= max(tol, 1e-9) * abs(want), (mcc, want)


CASES = [("bal_schur", 5, 2), ("bal_schur", 6, 1), ("bal_cgnr", 6, 1), ("general", 5, 2), ("general", 6, 1),
         ("bal_many_cameras", 5, 2), ("bal_many_cameras", 6, 1)]


@pytest.mark.parametrize("kind,solver_type,pre", CASES)
def test_lm_compute_step_matches_reference(hip, oracle, problems, kind, solver_type, pre):
    if kind == "bal_schur":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=30, num_points=2000, num_observations=9000, seed=41, skew=0.5)
    elif kind == "bal_many_cameras":  # > 2261 cameras: the camera accumulators leave LDS (per-slot F^T z + camera-major pass)
        p = problems.synthetic_bal(None, layout="schur", num_cameras=2500, num_points=12000, num_observations=62000, seed=43, skew=0.4)
    elif kind == "bal_cgnr":
        p = problems.synthetic_bal(None, layout="cgnr", num_cameras=30, num_points=2000, num_observations=9000, seed=41, skew=0.5)
    else:
        p = problems.random_schur_problem(num_e_blocks=50, num_f_blocks=8, num_no_e_rows=3, seed=42)
    if solver_type == hip.CGNR and kind == "bal_cgnr":
        p.num_eliminate_blocks = 0
    s = make_solver(hip, p, solver_type, pre, max_it=500)
    assert s.info().kernel_path == (hip.PATH_GENERIC if kind == "general" else hip.PATH_BAL)
    if kind != "general":
        assert s.info().camera_accum_in_lds == (0 if kind == "bal_many_cameras" else 1)
    radius, eta = 1e4, 0.1
    step, summ, model_cost_change = s.lm_compute_step(p.values, p.b, radius, eta)
    ref_step, ref_summ, ref_mcc, ref_D, diag = reference_step(oracle, hip, p, solver_type, pre, radius, eta)
    assert rel(s.lm_diagonal(), ref_D) <= 1e-13
    check_step(oracle, hip, p, solver_type, pre, ref_D, step, summ, model_cost_change, eta)
    assert model_cost_change > 0  # a valid LM step decreases the model
    # rejected step: radius halves, the diagonal is reused (StepRejected, :171-175) even if J changed
    p2 = type(p)(p.bs, p.values * 1.5, p.b, None, p.num_eliminate_blocks)
    step2, summ2, mcc2 = s.lm_compute_step(p2.values, p2.b, radius / 2, eta, reuse_diagonal=True)
    ref2 = reference_step(oracle, hip, p2, solver_type, pre, radius / 2, eta, diag=diag)
    assert rel(s.lm_diagonal(), ref2[3]) <= 1e-13
    check_step(oracle, hip, p2, solver_type, pre, ref2[3], step2, summ2, mcc2, eta)
    # and a fresh diagonal again
    step3, summ3, _ = s.lm_compute_step(p2.values, p2.b, radius, eta)
    ref3 = reference_step(oracle, hip, p2, solver_type, pre, radius, eta)
    assert rel(s.lm_diagonal(), ref3[3]) <= 1e-13
    s.close()


def test_camera_major_mo_records_experiment(hip, oracle, problems, monkeypatch):
    """CERES_HIP_MO_CAMERA_MAJOR=1 (design/11 §11.8: measured, net zero, off by default): kInit scatters the 2 x 2 M_o records into
    camera-major order and the SCHUR_JACOBI pass reads them back to back.  Same step, same preconditioner blocks as the oracle's."""
    monkeypatch.setenv("CERES_HIP_MO_CAMERA_MAJOR", "1")
    p = problems.synthetic_bal(None, layout="schur", num_cameras=30, num_points=2000, num_observations=9000, seed=41, skew=0.5)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
    assert s.info().kernel_path == hip.PATH_BAL
    step, summ, mcc = s.lm_compute_step(p.values, p.b, 1e4, 0.1)
    ref_step, ref_summ, ref_mcc, ref_D, diag = reference_step(oracle, hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, 1e4, 0.1)
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, ref_D, step, summ, mcc, 0.1)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    assert rel(s.preconditioner_blocks(not_inverted=True), m.schur_jacobi(p.values, ref_D)[1]) <= 1e-11
    s.close()


@pytest.mark.parametrize("kind,solver_type,pre", [("bal_schur", 5, 2), ("bal_cgnr", 6, 1), ("general", 5, 2), ("bal_many_cameras", 5, 2)])
def test_retry_after_rejection_keeps_the_resident_jacobian(hip, oracle, problems, kind, solver_type, pre):
    """ceres_hip_lm_options::values_unchanged: after a rejected step the minimizer calls ComputeStep on the SAME Jacobian with a smaller
    radius (trust_region_minimizer.cc:832-837 evaluates only in HandleSuccessfulStep; levenberg_marquardt_strategy.cc:134,164,170).  The
    retry must not need the host arrays (None is passed), must give what a full re-submission gives, and must match the oracle."""
    import torch
    if kind == "bal_schur":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=30, num_points=2000, num_observations=9000, seed=47, skew=0.5)
    elif kind == "bal_many_cameras":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=2500, num_points=12000, num_observations=62000, seed=48, skew=0.4)
    elif kind == "bal_cgnr":
        p = problems.synthetic_bal(None, layout="cgnr", num_cameras=30, num_points=2000, num_observations=9000, seed=47, skew=0.5)
        p.num_eliminate_blocks = 0
    else:
        p = problems.random_schur_problem(num_e_blocks=50, num_f_blocks=8, num_no_e_rows=3, seed=49)
    radius, eta = 1e4, 0.1
    s = make_solver(hip, p, solver_type, pre, max_it=500)
    ref = make_solver(hip, p, solver_type, pre, max_it=500)
    s.lm_compute_step(p.values, p.b, radius, eta)
    ref.lm_compute_step(p.values, p.b, radius, eta)
    diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
    for k in (1, 2):   # two rejections in a row: radius / 2, then / 8 (decrease_factor doubles)
        r_k = radius / (2.0 if k == 1 else 8.0)
        step, summ, mcc = s.lm_compute_step(None, None, r_k, eta, reuse_diagonal=True, values_unchanged=True)
        step_f, summ_f, mcc_f = ref.lm_compute_step(p.values, p.b, r_k, eta, reuse_diagonal=True)
        assert summ.num_iterations == summ_f.num_iterations and summ.termination_type == summ_f.termination_type
        assert rel(step, step_f) <= 1e-12 and abs(mcc - mcc_f) <= 1e-12 * abs(mcc_f)
        check_step(oracle, hip, p, solver_type, pre, np.sqrt(diag / r_k), step, summ, mcc, eta)
    # an accepted step later: new values go up as usual
    p2v = p.values * 1.1
    step3, summ3, mcc3 = s.lm_compute_step(p2v, p.b, radius, eta)
    step3f, summ3f, mcc3f = ref.lm_compute_step(p2v, p.b, radius, eta)
    assert rel(step3, step3f) <= 1e-12
    # the device entry point: same pointers, untouched contents
    dev = torch.device("cuda:0")
    tv, tb = torch.from_numpy(p2v).to(dev), torch.from_numpy(p.b).to(dev)
    tx = torch.full((p.num_cols,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    s.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), radius, eta)
    summ_d, mcc_d, finite = s.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), radius / 2, eta, reuse_diagonal=True, values_unchanged=True)
    step_rf, summ_rf, mcc_rf = ref.lm_compute_step(p2v, p.b, radius / 2, eta, reuse_diagonal=True)
    assert finite and rel(tx.cpu().numpy(), step_rf) <= 1e-12 and abs(mcc_d - mcc_rf) <= 1e-12 * abs(mcc_rf)
    # other pointers with values_unchanged = 1: refused, not silently wrong
    other = tv.clone()
    with pytest.raises(Exception):
        s.lm_compute_step_device(other.data_ptr(), tb.data_ptr(), tx.data_ptr(), radius / 2, eta, reuse_diagonal=True, values_unchanged=True)
    # the LinearSolver::Solve form of the retry: only D goes up
    D1, D2 = np.sqrt(diag / radius), np.sqrt(diag / (radius / 2))
    x1, s1 = s.solve(p.values, p.b, hip.PerSolveOptions(D=D1, q_tolerance=eta, r_tolerance=-1.0))
    x2, s2 = s.solve_unchanged_values(hip.PerSolveOptions(D=D2, q_tolerance=eta, r_tolerance=-1.0))
    x2f, s2f = ref.solve(p.values, p.b, hip.PerSolveOptions(D=D2, q_tolerance=eta, r_tolerance=-1.0))
    assert s2.termination_type == s2f.termination_type == hip.SUCCESS and s2.num_iterations == s2f.num_iterations
    assert rel(x2, x2f) <= 1e-12
    s.close()
    ref.close()


def test_lm_step_device_pointers_and_long_tracks(hip, oracle, problems):
    import torch
    p = problems.synthetic_bal(None, num_cameras=230, num_points=350, num_observations=11000, seed=43)  # tracks > 64
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
    dev = torch.device("cuda:0")
    tv, tb = torch.from_numpy(p.values).to(dev), torch.from_numpy(p.b).to(dev)
    tx = torch.full((p.num_cols,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    summ, mcc, finite = s.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), 1e4, 0.1)
    ref_step, ref_summ, ref_mcc, ref_D, _ = reference_step(oracle, hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, 1e4, 0.1)
    assert finite and summ.termination_type == hip.SUCCESS
    assert rel(s.lm_diagonal(), ref_D) <= 1e-13
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, ref_D, tx.cpu().numpy(), summ, mcc, 0.1)
    s.close()


@pytest.mark.parametrize("kind", ["bal", "general"])
def test_scale_columns_and_column_norms(hip, oracle, problems, kind):
    p = (problems.synthetic_bal(None, num_cameras=20, num_points=900, num_observations=4000, seed=44) if kind == "bal"
         else problems.random_schur_problem(num_e_blocks=30, num_f_blocks=7, seed=45))
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    s.load(p.values, p.b, p.D)
    assert rel(s.squared_column_norm(), m.squared_column_norm(p.values)) <= 1e-13
    # jacobi scaling as TrustRegionMinimizer applies it (trust_region_minimizer.cc:263-279)
    scale = 1.0 / (1.0 + np.sqrt(m.squared_column_norm(p.values)))
    scaled = s.scale_columns(scale)
    want = m.scale_columns(p.values, scale)
    np.testing.assert_allclose(scaled[: len(want)], want, rtol=1e-15)
    assert rel(s.squared_column_norm(), m.squared_column_norm(want)) <= 1e-13  # operates on the scaled copy now
    s.schur_init()
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(want, p.D, p.b)
    assert rel(s.schur_rhs(), isc.rhs()) <= 1e-12
    s.close()


@pytest.mark.parametrize("solver_type,pre", [(5, 2), (6, 1)])
def test_lm_step_sharded_code_paths_in_loopback(hip, problems, solver_type, pre):
    # world > 1 branches of the LM step (fused LM diagonal from all-reduced sums, model cost and
    # finite flag summed over ranks) through a 1-rank RCCL communicator: must equal the unsharded step
    p = problems.synthetic_bal(None, num_cameras=25, num_points=1500, num_observations=7000, seed=46)
    o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=300,
                                elimination_groups=[p.num_eliminate_blocks])
    ref, loop = hip.HipLinearSolver(o), hip.HipLinearSolver(o, loopback_world=8)
    ref.set_structure(p.bs)
    loop.set_structure(p.bs)
    a = ref.lm_compute_step(p.values, p.b, 1e4, 0.1)
    b = loop.lm_compute_step(p.values, p.b, 1e4, 0.1)
    assert rel(loop.lm_diagonal(), ref.lm_diagonal()) <= 1e-13
    assert a[1].termination_type == b[1].termination_type == hip.SUCCESS and a[1].num_iterations == b[1].num_iterations
    assert rel(b[0], a[0]) <= 1e-10 and abs(a[2] - b[2]) <= 1e-10 * abs(a[2])
    # a rejected step (diagonal reused, radius halved) and a fresh one again: fused and unfused LM diagonals alternate
    for radius, reuse in ((5e3, True), (2e4, False)):
        a = ref.lm_compute_step(p.values * 1.25, p.b, radius, 0.1, reuse_diagonal=reuse)
        b = loop.lm_compute_step(p.values * 1.25, p.b, radius, 0.1, reuse_diagonal=reuse)
        assert rel(loop.lm_diagonal(), ref.lm_diagonal()) <= 1e-13
        assert a[1].num_iterations == b[1].num_iterations
        assert rel(b[0], a[0]) <= 1e-10 and abs(a[2] - b[2]) <= 1e-10 * abs(a[2])
    ref.close()
    loop.close()


@pytest.mark.parametrize("solver_type,pre", [(5, 2), (6, 1)])
@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("nc", [25, 1500])
def test_lm_step_with_ghost_peers_equals_the_unsharded_step(hip, problems, monkeypatch, solver_type, pre, fuse, nc):
    """ceres_hip_debug_comm_ghost_peers (bench.py: extra.shard_ceiling): rank 0 of eight ranks whose peers are local dummy buffers that
    contribute zeros — every exchange of the sharded step runs (inside the producing kernels, or with CERES_HIP_P2P_FUSE=0 as stand-alone
    all-reduces), so with the WHOLE problem on this rank the step must be the unsharded one.  1500 cameras: the four-cameras-per-wavefront
    exchange of the per-camera sums."""
    monkeypatch.setenv("CERES_HIP_P2P_FUSE", fuse)
    p = problems.synthetic_bal(None, num_cameras=nc, num_points=1500 if nc == 25 else 4000, num_observations=7000 if nc == 25 else 14000, seed=46)
    o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=300,
                                elimination_groups=[p.num_eliminate_blocks])
    ref = hip.HipLinearSolver(o)
    ghost = hip.HipLinearSolver(o, ghost_world=8, p2p_max_elements=99 * nc + 2)
    ref.set_structure(p.bs)
    ghost.set_structure(p.bs)
    assert ghost.info().world_size == 8 and ghost.info().p2p_enabled
    a = ref.lm_compute_step(p.values, p.b, 1e4, 0.1)
    b = ghost.lm_compute_step(p.values, p.b, 1e4, 0.1)
    assert rel(ghost.lm_diagonal(), ref.lm_diagonal()) <= 1e-13
    assert a[1].termination_type == b[1].termination_type == hip.SUCCESS and a[1].num_iterations == b[1].num_iterations
    assert rel(b[0], a[0]) <= 1e-10 and abs(a[2] - b[2]) <= 1e-10 * abs(a[2])
    assert ghost.info().collectives_last_step >= 4
    for radius, reuse in ((5e3, True), (2e4, False)):   # a rejected step, then a fresh one
        a = ref.lm_compute_step(p.values * 1.25, p.b, radius, 0.1, reuse_diagonal=reuse)
        b = ghost.lm_compute_step(p.values * 1.25, p.b, radius, 0.1, reuse_diagonal=reuse)
        assert a[1].num_iterations == b[1].num_iterations
        assert rel(b[0], a[0]) <= 1e-10 and abs(a[2] - b[2]) <= 1e-10 * abs(a[2])
    ref.close()
    ghost.close()


@pytest.mark.parametrize("solver_type,pre", [(5, 2), (6, 1)])
def test_speculative_tail_equals_the_two_synchronisation_sequence(hip, oracle, problems, monkeypatch, solver_type, pre):
    """The LM step's tail (back-substitution / model cost, negation, finite check, read-back) is enqueued in front of every poll
    of the CG status word and gated by it on the device.  Every class of CG ending must give what the plain
    sequence (CERES_HIP_SPECULATE=0) gives: convergence inside the first batch of iterations, convergence after several polls
    (the gated kernels ran as no-ops first), NO_CONVERGENCE at the iteration cap (the step is still produced), a non-finite
    Jacobian (FAILURE: nothing is produced)."""
    layout = "schur" if solver_type == hip.ITERATIVE_SCHUR else "cgnr"
    p = problems.synthetic_bal(None, layout=layout, num_cameras=35, num_points=2500, num_observations=12000, seed=61, skew=0.6)
    if solver_type == hip.CGNR:
        p.num_eliminate_blocks = 0
    # badly scaled camera columns: CG needs a few dozen iterations at a small eta
    cams = np.flatnonzero(np.asarray(p.bs.col_block_size) == 9)
    scale = np.ones(p.bs.num_cols)
    for j in cams[::3]:
        scale[p.bs.col_block_pos[j]:p.bs.col_block_pos[j] + 9] = 30.0
    hard = type(p)(p.bs, oracle.Matrix(p.bs, 0).scale_columns(p.values, scale), p.b, None, p.num_eliminate_blocks)
    bad = type(p)(p.bs, p.values.copy(), p.b, None, p.num_eliminate_blocks)
    bad.values[::997] = np.nan
    runs = {}
    for spec in ("1", "0"):
        monkeypatch.setenv("CERES_HIP_SPECULATE", spec)
        out = []
        for prob, eta, max_it in ((p, 0.1, 500), (hard, 1e-8, 500), (hard, 1e-8, 3), (bad, 0.1, 500)):
            s = make_solver(hip, prob, solver_type, pre, max_it=max_it)
            step, summ, mcc = s.lm_compute_step(prob.values, prob.b, 1e4, eta)
            out.append((step, summ.termination_type, summ.num_iterations, mcc))
            s.close()
        runs[spec] = out
    a, b = runs["1"], runs["0"]
    assert [r[1] for r in a] == [r[1] for r in b] == [hip.SUCCESS, hip.SUCCESS, hip.NO_CONVERGENCE, hip.FAILURE]
    assert [r[2] for r in a][:3] == [r[2] for r in b][:3]
    assert a[0][2] <= 2 and a[1][2] > 4 and a[2][2] == 3   # inside the first batch / after several polls / at the cap
    for k in range(3):
        assert rel(a[k][0], b[k][0]) <= 1e-12 and abs(a[k][3] - b[k][3]) <= 1e-12 * abs(b[k][3]), k
    # the converged cases also against the oracle
    # (eta = 1e-8 ends on a zeta of rounding size: the index may differ — compare with the oracle's iterate of the PRODUCT's index)
    ref_step, ref_summ, ref_mcc, ref_D, _ = reference_step(oracle, hip, hard, solver_type, pre, 1e4, 1e-8)
    assert abs(ref_summ.num_iterations - a[1][2]) <= 2, (ref_summ, a[1][2])
    m = oracle.Matrix(hard.bs, hard.num_eliminate_blocks if solver_type == hip.ITERATIVE_SCHUR else 0)
    fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m.cgnr_solve
    xk, sk = fn(hard.values, hard.b, ref_D, preconditioner=pre, min_it=a[1][2], max_it=a[1][2], q_tol=-1.0, r_tol=-1.0)
    assert sk.num_iterations == a[1][2] and rel(a[1][0], -xk) <= 1e-8, (sk, rel(a[1][0], -xk))


@pytest.mark.parametrize("solver_type,pre", [(5, 2), (6, 1)])
def test_min_lm_diagonal_zero_is_a_valid_option(hip, oracle, problems, solver_type, pre):
    """Solver::Options::IsValid accepts min_lm_diagonal = 0 (internal/ceres/solver.cc:414: OPTION_GE): the LM diagonal is then the
    unclipped column norm.  (The boundary used to refuse it: found by the sharded failure-path probe, tools/probes/poison_ranks.py.)"""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=30, num_points=2000, num_observations=9000, seed=41, skew=0.5)
    if solver_type == hip.CGNR:
        p.num_eliminate_blocks = 0
    s = make_solver(hip, p, solver_type, pre, max_it=500)
    radius = 3.0
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1, min_diagonal=0.0, max_diagonal=1e32)
    diag = oracle.Matrix(p.bs, 0).squared_column_norm(p.values)
    check_step(oracle, hip, p, solver_type, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1)
    for bad in (dict(min_diagonal=-1.0), dict(min_diagonal=2.0, max_diagonal=1.0), dict(min_diagonal=float("nan")), dict(max_diagonal=-1.0, min_diagonal=0.0)):
        with pytest.raises(hip.HipError, match="bad LM options"):
            s.lm_compute_step(p.values, p.b, radius, 0.1, **bad)
    s.close()
