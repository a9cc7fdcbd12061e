"""Edge cases through the C ABI on the GPU: degenerate structures, call-order and argument
errors (no exceptions cross the boundary: error codes + ceres_hip_last_error)."""
import numpy as np
import pytest

from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


def bal_from_lists(problems, BlockStructure, n_cams, obs):
    """obs: list of (point, camera) sorted by point.  Schur layout, N(0,1) values."""
    n_pts = max(p for p, _ in obs) + 1
    n_o = len(obs)
    rows = [(2, [(p, 6 * r), (n_pts + c, 6 * n_o + 18 * r)]) for r, (p, c) in enumerate(obs)]
    bs = BlockStructure.from_rows([3] * n_pts + [9] * n_cams, rows)
    rng = np.random.default_rng(7)
    return problems.LinearProblem(bs, rng.standard_normal(24 * n_o), rng.standard_normal(2 * n_o), 0.5 + rng.random(bs.num_cols), n_pts)


def check_against_oracle(hip, oracle, p, expect_path):
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for solver_type, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        s = make_solver(hip, p, solver_type, pre, max_it=300)
        assert s.info().kernel_path == expect_path
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else oracle.Matrix(p.bs, 0).cgnr_solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=300, q_tol=-1.0, r_tol=1e-12)
        assert summ.termination_type == so.termination_type == hip.SUCCESS, (summ, so)
        assert rel(x, xo) <= 1e-8
        s.close()


def test_cameras_without_observations_and_single_observation_points(hip, oracle, problems):
    from ceres_solver_amd import BlockStructure
    # cameras 2 and 5 are never observed; point 3 has a single observation (allowed: >= 1 residual per
    # eliminated block, internal/ceres/reorder_program.cc:313-317)
    obs = [(0, 0), (0, 1), (1, 1), (1, 3), (1, 4), (2, 0), (2, 4), (3, 3), (4, 0), (4, 1), (4, 3), (4, 4)]
    p = bal_from_lists(problems, BlockStructure, 6, obs)
    check_against_oracle(hip, oracle, p, hip.PATH_BAL)


def test_single_point_and_exactly_one_tile(hip, oracle, problems):
    from ceres_solver_amd import BlockStructure
    check_against_oracle(hip, oracle, bal_from_lists(problems, BlockStructure, 3, [(0, 0), (0, 1), (0, 2)]), hip.PATH_BAL)
    # 64 observations of ONE point (fills a tile exactly), then 65 (becomes a long point of two tiles)
    for n in (64, 65, 129):
        p = bal_from_lists(problems, BlockStructure, n, [(0, c) for c in range(n)] + [(1, 0), (1, 1)])
        check_against_oracle(hip, oracle, p, hip.PATH_BAL)


@pytest.mark.parametrize("layout", ["schur", "cgnr"])
def test_tiles_full_of_single_observation_points(hip, oracle, problems, layout):
    """Runs of one-observation points (allowed: >= 1 residual per eliminated block): a tile would hold up to 64
    of them = 192 point-space scalars, but the cooperative point-space path of the fused JtJx covers 128
    per tile, so the plan caps a tile at 42 points (csrc/plan.cc).  Operators AND solves against the oracle,
    both solvers, in the contiguous (schur) layout that selects the pipelined kernels and in the cgnr layout."""
    tracks = [1] * 700 + [2] * 60 + [1] * 130 + [3, 1, 1, 70, 1, 1] + [1] * 64 + [5] * 30
    p = problems.bal_from_tracks(tracks, 90, layout=layout, seed=12)
    rng = np.random.default_rng(1)
    p.D = 0.5 + rng.random(p.bs.num_cols)   # one-observation points need D: their E^T E alone is singular
    nelim = p.num_eliminate_blocks
    m0 = oracle.Matrix(p.bs, 0)
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, max_it=400)
    assert s.info().kernel_path == hip.PATH_BAL
    s.load(p.values, p.b, p.D)
    x = rng.standard_normal(p.bs.num_cols)
    Jx = m0.right_multiply(p.values, x)
    want = m0.left_multiply(p.values, Jx) + p.D ** 2 * x
    got = s.jtjx(x)
    assert np.isfinite(got).all() and rel(got, want) <= 1e-12
    y = rng.standard_normal(p.bs.num_cols)   # symmetry: the defect made the operator non-symmetric
    assert abs(y @ got - x @ s.jtjx(y)) <= 1e-11 * abs(y @ got)
    assert rel(s.jtb(), m0.left_multiply(p.values, p.b)) <= 1e-12
    xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
    xo, so = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=0, max_it=400, q_tol=-1.0, r_tol=1e-12)
    assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(xs, xo) <= 1e-8
    s.close()
    if layout == "schur":
        m = oracle.Matrix(p.bs, nelim)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=400)
        s.load(p.values, p.b, p.D)
        isc = oracle.ImplicitSchurComplement(m)
        isc.init(p.values, p.D, p.b)
        s.schur_init()
        xf = rng.standard_normal(m.num_cols_f)
        assert rel(s.schur_rhs(), isc.rhs()) <= 1e-12 and rel(s.schur_sx(xf), isc.sx(xf)) <= 1e-12
        assert rel(s.back_substitute(xf), isc.back_substitute(xf)) <= 1e-12
        xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
        xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=0, max_it=400, q_tol=-1.0, r_tol=1e-12)
        assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(xs, xo) <= 1e-8
        s.close()


def test_generic_structures_with_odd_rows(hip, oracle, problems):
    from ceres_solver_amd import BlockStructure
    # a row block with no cells at all, 1-wide and 16-wide blocks, an E block seen by one row only
    col = [2, 1, 16, 3]
    rows = [(1, [(0, 0), (2, 2)]), (3, [(0, 18), (3, 24)]), (2, [(1, 33), (2, 35)]), (2, []), (1, [(3, 67)])]
    bs = BlockStructure.from_rows(col, rows)
    rng = np.random.default_rng(3)
    p = problems.LinearProblem(bs, rng.standard_normal(70), rng.standard_normal(bs.num_rows), 0.5 + rng.random(bs.num_cols), 2)
    check_against_oracle(hip, oracle, p, hip.PATH_GENERIC)


def test_errors_are_codes_not_crashes(hip, problems):
    p = problems.synthetic_bal(None, num_cameras=5, num_points=30, num_observations=100, seed=1)
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, max_num_iterations=10,
                                elimination_groups=[p.num_eliminate_blocks])
    s = hip.HipLinearSolver(o)
    # solve before set_structure: FATAL_ERROR summary, like a LinearSolver that could not run
    s._info = type("I", (), dict(num_rows=p.num_rows, num_cols=p.num_cols))()
    s.bs = p.bs
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))
    assert summ.termination_type == hip.FATAL_ERROR and "set_structure" in summ.message and np.isnan(x).all()
    s.set_structure(p.bs)
    with pytest.raises(hip.HipError):  # one instance sees one sparsity (internal/ceres/linear_solver.h:137-142)
        s.set_structure(p.bs)
    with pytest.raises(hip.HipError):
        s.schur_sx(np.zeros(s.info().num_cols_f))  # nothing loaded yet
    with pytest.raises(hip.HipError):
        s.jtjx(np.zeros(p.num_cols))               # wrong solver kind, even after a load
    s.close()
    # ITERATIVE_SCHUR with no eliminated blocks is the host's job to turn into CGNR
    with pytest.raises(hip.HipError) as e:
        bad = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, max_num_iterations=5, elimination_groups=[0]))
        bad.set_structure(p.bs)
    assert "num_eliminate_blocks" in str(e.value)
    via_create = hip.create_linear_solver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI,
                                                                  max_num_iterations=50, min_num_iterations=0, elimination_groups=[0]), p.bs)
    assert via_create.options.type == hip.CGNR and via_create.options.preconditioner_type == hip.JACOBI
    x, summ = via_create.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert summ.termination_type == hip.SUCCESS and np.isfinite(x).all()
    via_create.close()
    # rows not grouped by E block are rejected for a Schur solver
    perm = np.arange(p.bs.num_row_blocks)
    perm[[0, -1]] = perm[[-1, 0]]
    from ceres_solver_amd import BlockStructure
    ptr = p.bs.row_cell_ptr
    cells = np.concatenate([np.arange(ptr[i], ptr[i + 1]) for i in perm])
    shuffled = BlockStructure(p.bs.row_block_size[perm], p.bs.row_block_pos, p.bs.col_block_size, p.bs.col_block_pos,
                              np.concatenate([[0], np.cumsum(np.diff(ptr)[perm])]), p.bs.cell_col_block[cells], p.bs.cell_value_pos[cells])
    s2 = hip.HipLinearSolver(o)
    with pytest.raises(hip.HipError) as e:
        s2.set_structure(shuffled)
    assert "grouped by E block" in str(e.value) or "not ordered" in str(e.value)
    s2.close()
    # a block wider than the generic kernels take
    big = BlockStructure.from_rows([20, 3], [(2, [(0, 0), (1, 40)])])
    s3 = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.CGNR, max_num_iterations=5))
    with pytest.raises(hip.HipError) as e:
        s3.set_structure(big)
    assert "exceeds" in str(e.value)
    s3.close()


def test_indefinite_and_failure_paths(hip, problems):
    # E^T E singular without D: the Cholesky of the point block fails -> FAILURE, x untouched (NaN-poisoned)
    from ceres_solver_amd import BlockStructure
    rows = [(1, [(0, 0), (1, 2)]), (1, [(0, 4), (1, 6)])]
    bs = BlockStructure.from_rows([2, 2], rows)
    vals = np.array([1.0, 2.0, 1.0, 0.0, 2.0, 4.0, 0.0, 1.0])  # E rows (1,2) and (2,4): rank one
    p = problems.LinearProblem(bs, vals, np.array([1.0, 2.0]), None, 1)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI, max_it=10)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=None, q_tolerance=0.0, r_tolerance=1e-10))
    assert summ.termination_type == hip.FAILURE and np.isnan(x).all(), summ
    s.close()


@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("solver_type,pre", [(6, 1), (5, 2)])
def test_singular_preconditioner_block_is_a_failure(hip, problems, solver_type, pre, force_generic):
    # Preconditioner::Update returning false ends the solve with FAILURE
    # (iterative_schur_complement_solver.cc:113-121, cgnr_solver.cc:176-183).  One camera with an all-zero
    # Jacobian and no regularisation: its JACOBI / SCHUR_JACOBI block is singular.  On the fused path the
    # flag is consumed on the device by the CG init kernel (no host round trip before CG).
    p = problems.synthetic_bal(None, layout="schur", num_cameras=9, num_points=200, num_observations=900, seed=5)
    cam0 = int(p.camera_of_row.min())
    rows = np.nonzero(p.camera_of_row == cam0)[0]
    fpos = p.bs.cell_value_pos[1::2][rows].astype(np.int64)
    vals = p.values.copy()
    vals[(fpos[:, None] + np.arange(18)[None, :]).reshape(-1)] = 0.0
    s = make_solver(hip, p, solver_type, pre, force_generic, max_it=50)
    x, summ = s.solve(vals, p.b, hip.PerSolveOptions(D=None, q_tolerance=0.0, r_tolerance=1e-10))
    assert summ.termination_type == hip.FAILURE, summ
    assert "Preconditioner update failed" in summ.message
    # the instance stays usable
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-10))
    assert summ.termination_type == hip.SUCCESS
    s.close()
