"""The fused <2,3,9> path with LEFTOVER ROWS: trailing row blocks without a point cell (priors / regularisers on cameras) next to
the BAL rows.  The reference handles them inside the same eliminator and the same partitioned view —
SchurEliminator::NoEBlockRowsUpdate (internal/ceres/schur_eliminator_impl.h:574-666), PartitionedMatrixView's loops over the rows
behind num_row_blocks_e (internal/ceres/partitioned_matrix_view_impl.h:171-190, 617-658) — and so does the product: the tiles
cover the BAL rows, small generic kernels add the remainder's sums (csrc/solver.hip: add_remainder).  Everything is checked
against the oracle on the WHOLE problem, and against the generic path of the product."""
import numpy as np
import pytest

from step_check import assert_lm_style_step
from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu


def problem(problems, layout, seed=11, nc=37, npts=4000, nobs=18000, rows=60, row_size=9, pairs=0.3):
    p = problems.synthetic_bal(None, layout=layout, num_cameras=nc, num_points=npts, num_observations=nobs, seed=seed, skew=0.5)
    return problems.add_camera_rows(p, rows, seed=seed, row_size=row_size, pair_fraction=pairs)


def test_plan_takes_the_fused_path_with_trailing_camera_rows(hip, problems):
    p = problem(problems, "schur")
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    info = s.info()
    assert info.kernel_path == hip.PATH_BAL and info.num_observations == 18000 and info.num_row_blocks_e == 18000
    s.close()


@pytest.mark.parametrize("row_size,pairs", [(9, 0.3), (3, 0.0), (1, 1.0)])
def test_schur_operators_with_leftover_rows_against_the_oracle(hip, oracle, problems, row_size, pairs):
    p = problem(problems, "schur", row_size=row_size, pairs=pairs)
    rng = np.random.default_rng(0)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=200)
    g = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, True, max_it=200)
    assert s.info().kernel_path == hip.PATH_BAL and g.info().kernel_path == hip.PATH_GENERIC
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    xf = rng.standard_normal(m.num_cols_f)
    inv, raw = m.schur_jacobi(p.values, p.D)
    for name, slv in (("fused", s), ("generic", g)):
        slv.load(p.values, p.b, p.D)
        slv.schur_init()
        errs = {"rhs": rel(slv.schur_rhs(), isc.rhs()), "sx": rel(slv.schur_sx(xf), isc.sx(xf)),
                "back_substitute": rel(slv.back_substitute(xf), isc.back_substitute(xf)),
                "squared_column_norm": rel(slv.squared_column_norm(), m.squared_column_norm(p.values))}
        slv.schur_jacobi_update()
        mine = slv.preconditioner_blocks(not_inverted=True).reshape(-1, 9, 9)
        errs["schur_jacobi_raw"] = np.abs(np.triu(mine) - np.triu(raw.reshape(-1, 9, 9))).max() / np.abs(raw).max()
        slv.schur_jacobi_update()
        errs["schur_jacobi_inv"] = rel(slv.preconditioner_blocks(), inv)
        for k, v in errs.items():
            assert v <= (1e-10 if k.endswith("_inv") else 1e-12), (name, k, v)
    # ITERATIVE_SCHUR + JACOBI: blockdiag(F^T F + D^2)^-1 includes the leftover rows' F^T F
    j = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI)
    j.load(p.values, p.b, p.D)
    j.block_jacobi_update()
    ftf = m.block_diagonal_ftf(p.values).reshape(-1, 9, 9) + np.stack([np.diag(d ** 2) for d in p.D[m.num_cols_e:].reshape(-1, 9)])
    assert rel(j.preconditioner_blocks(), np.linalg.inv(ftf).reshape(-1)) <= 1e-10
    j.close()
    # solves: converged (rung 3) and the call LM makes (rung 4, unconditional)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
    xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=0, max_it=200, q_tol=-1.0, r_tol=1e-12)
    assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(x, xo) <= 1e-8
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, q, r: m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=lo, max_it=hi, q_tol=q, r_tol=r),
                         0.1, hip.SUCCESS)
    s.close()
    g.close()


@pytest.mark.parametrize("layout", ["schur", "cgnr"])
def test_cgnr_operators_with_leftover_rows_against_the_oracle(hip, oracle, problems, layout):
    p = problem(problems, layout, seed=12)
    if layout == "cgnr":
        p.num_eliminate_blocks = 0
    rng = np.random.default_rng(1)
    m0 = oracle.Matrix(p.bs, 0)
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, max_it=300)
    assert s.info().kernel_path == hip.PATH_BAL
    s.load(p.values, p.b, p.D)
    x = rng.standard_normal(p.bs.num_cols)
    want = m0.left_multiply(p.values, m0.right_multiply(p.values, x)) + p.D ** 2 * x
    errs = {"jtjx": rel(s.jtjx(x), want), "jtb": rel(s.jtb(), m0.left_multiply(p.values, p.b)),
            "squared_column_norm": rel(s.squared_column_norm(), m0.squared_column_norm(p.values))}
    s.block_jacobi_update()
    inv, _ = m0.block_jacobi(p.values, p.D)
    errs["block_jacobi_inv"] = rel(s.preconditioner_blocks(), inv)
    for k, v in errs.items():
        assert v <= (1e-10 if k.endswith("_inv") else 1e-12), (k, v)
    xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
    xo, so = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=0, max_it=300, q_tol=-1.0, r_tol=1e-12)
    assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(xs, xo) <= 1e-8
    xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(xs, summ, lambda lo, hi, q, r: m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=lo, max_it=hi, q_tol=q, r_tol=r),
                         0.1, hip.SUCCESS)
    s.close()


@pytest.mark.parametrize("solver,pre", [("schur", 2), ("cgnr", 1)])
def test_lm_step_with_leftover_rows(hip, oracle, problems, solver, pre):
    """LevenbergMarquardtStrategy::ComputeStep + the model cost change with the fused LM diagonal: the leftover rows join the camera
    columns' norms (hence D), the preconditioner, the operator, the right-hand side and the model cost."""
    p = problem(problems, "schur", seed=13)
    m0 = oracle.Matrix(p.bs, 0)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    typ = hip.ITERATIVE_SCHUR if solver == "schur" else hip.CGNR
    s = make_solver(hip, p, typ, hip.SCHUR_JACOBI if solver == "schur" else hip.JACOBI, max_it=300)
    assert s.info().kernel_path == hip.PATH_BAL
    radius = 3e3
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
    D = np.sqrt(np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32) / radius)
    assert rel(s.lm_diagonal(), D) <= 1e-12
    fn = m.iterative_schur_solve if solver == "schur" else m0.cgnr_solve
    assert_lm_style_step(-step, summ, lambda lo, hi, q, r: fn(p.values, p.b, D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    Jx = m0.right_multiply(p.values, step)
    want = -(Jx @ (p.b + Jx / 2))
    assert abs(mcc - want) <= 1e-9 * abs(want), (mcc, want)
    # a rejected step: same Jacobian, smaller radius, stored diagonal
    step2, summ2, mcc2 = s.lm_compute_step(p.values, p.b, radius / 2, 0.1, reuse_diagonal=True)
    D2 = D * np.sqrt(2.0)
    assert_lm_style_step(-step2, summ2, lambda lo, hi, q, r: fn(p.values, p.b, D2, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    s.close()


# ---- round 5: leftover rows next to every compiled shape (the remainder kernels are templated on the camera width) ----
OTHER_SHAPES = {
    "f9": dict(camera_width=9),   # (the BAL shape too: CGNR WITHOUT elimination groups on Schur-ordered columns — cameras back to back behind the points)
    "f10_quaternion_cameras": dict(camera_width=10), "f6": dict(camera_width=6), "f3": dict(camera_width=3), "f4": dict(camera_width=4),
    "f8": dict(camera_width=8), "e4_f9": dict(point_width=4, camera_width=9), "e2_f2": dict(point_width=2, camera_width=2),
    # round 6: the widths added to the fused path this round had no remainder kernels (hipErrorInvalidValue from every pass over the
    # leftover rows; found by tools/fuzz_parity.py)
    "f5": dict(camera_width=5), "f7": dict(camera_width=7), "f2": dict(camera_width=2), "e4_f7": dict(point_width=4, camera_width=7),
    "r3_e3_f3": dict(row_height=3, point_width=3, camera_width=3), "r4_e4_f4": dict(row_height=4, point_width=4, camera_width=4),
}


@pytest.mark.parametrize("name", list(OTHER_SHAPES))
def test_leftover_rows_next_to_every_compiled_shape(hip, oracle, problems, name):
    """Priors on cameras that are not 9 wide (e.g. on quaternion cameras, examples/snavely_reprojection_error.h:164): the tiles of the
    shape's own kernels plus the remainder kernels of the camera's width — every Schur operator, both block preconditioners, the
    LM-style solve and the LM step against the oracle on the WHOLE problem; CGNR where it can tell cameras from points."""
    from test_gpu_operators import assert_errs, check_cgnr_operators, check_schur_operators
    from test_gpu_lm_step import check_step
    kw = OTHER_SHAPES[name]
    w = kw["camera_width"]
    base = problems.synthetic_structured(37, 3000, 13000, seed=21, skew=0.5, **kw)
    # rows as high as the camera is wide (a prior on the whole block), 2 high, and pairs coupling two cameras
    p = problems.add_camera_rows(base, 50, seed=3, row_size=w, pair_fraction=0.3, camera_width=w)
    p = problems.add_camera_rows(p, 30, seed=4, row_size=2, pair_fraction=0.0, camera_width=w)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for pre in (hip.SCHUR_JACOBI, hip.JACOBI):
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, pre, max_it=500)
        assert s.info().kernel_path == hip.PATH_BAL
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
        assert_lm_style_step(x, summ, lambda lo, hi, q, r: m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r),
                             0.1, hip.SUCCESS)
        radius = 1e4
        step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
        diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
        assert rel(s.lm_diagonal(), np.sqrt(diag / radius)) <= 1e-13
        check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1)
        s.close()
    # CGNR knows no elimination order: points are whatever is 3 wide (tests/test_gpu_shapes.py) — fused where that reading exists
    if kw.get("point_width", 3) == 3 and kw.get("row_height", 2) == 2 and w != 3:
        assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))


@pytest.mark.parametrize("shape", ["f9", "f10_quaternion_cameras", "f6"])
def test_cgnr_with_camera_rows_anywhere_among_the_observations(hip, oracle, problems, shape):
    """Without an elimination order the Jacobian's rows come in the order the residual blocks were added
    (internal/ceres/block_jacobian_writer.cc:198-263): priors on cameras may sit anywhere among the observations.  The tiles take the
    observation rows, the remainder kernels the others (their residuals gathered into a compact row space) — every CGNR operator, the
    converged solve, the LM-style solve and the LM step with its model cost against the oracle on the whole problem."""
    from test_gpu_operators import assert_errs, check_cgnr_operators
    from test_gpu_lm_step import check_step
    w = {"f9": 9, "f10_quaternion_cameras": 10, "f6": 6}[shape]
    base = problems.synthetic_structured(37, 3000, 13000, seed=31, skew=0.5, camera_width=w, layout="cgnr")
    base = type(base)(base.bs, base.values, base.b, base.D, 0)
    q = problems.add_camera_rows(base, 60, seed=5, row_size=w, pair_fraction=0.3, camera_width=w)
    q = problems.add_camera_rows(q, 25, seed=6, row_size=3, camera_width=w)
    rng = np.random.default_rng(17)
    n_obs = base.bs.num_row_blocks
    keys = np.concatenate([np.arange(n_obs, dtype=np.float64), rng.uniform(-1, n_obs, 85)])
    p = problems.permute_rows(q, np.argsort(keys, kind="stable"))
    assert p.bs.row_cell_ptr[1] - p.bs.row_cell_ptr[0] >= 1
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))
    m0 = oracle.Matrix(p.bs, 0)
    s = make_solver(hip, p, hip.CGNR, hip.JACOBI, max_it=400)
    assert s.info().kernel_path == hip.PATH_BAL and s.info().num_observations == n_obs
    xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-12))
    xo, so = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=0, max_it=400, q_tol=-1.0, r_tol=1e-12)
    assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(xs, xo) <= 1e-8
    xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(xs, summ, lambda lo, hi, qt, r: m0.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=lo, max_it=hi, q_tol=qt, r_tol=r),
                         0.1, hip.SUCCESS)
    radius = 3e3
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
    diag = np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32)
    assert rel(s.lm_diagonal(), np.sqrt(diag / radius)) <= 1e-12
    check_step(oracle, hip, p, hip.CGNR, hip.JACOBI, np.sqrt(diag / radius), step, summ, mcc, 0.1)
    s.close()
