"""The fused path beyond <2,3,9>: camera blocks of the widths the reference specialises for a 3-wide E block
(internal/ceres/generate_template_specializations.py:55-75: 3, 4, 6, 9; 8 as in (2,4,8); 10 = bundle_adjuster --use_quaternions,
examples/snavely_reprojection_error.h:164), rows with extra cells on SHARED blocks and rows without a camera cell
(examples/libmv_bundle_adjuster.cc:697-728: <2, 8, 6, 3>, the first camera constant) — every operator against the oracle at 1e-12, the
solvers on rungs (2) and (4) of the parity ladder, all on the FUSED kernels (kernel_path is asserted)."""
import numpy as np
import pytest

from step_check import assert_lm_style_step
from test_gpu_operators import assert_errs, check_cgnr_operators, check_schur_operators, make_solver, rel
from test_gpu_lm_step import check_step

pytestmark = pytest.mark.gpu

SHAPES = {
    "f3": dict(camera_width=3), "f4": dict(camera_width=4), "f6": dict(camera_width=6), "f8": dict(camera_width=8),
    # round 6: the widths the reference reaches through (2,3,d) — 2, 5 (a camera without distortion terms), 7 (quaternion + translation)
    "f2": dict(camera_width=2), "f5": dict(camera_width=5), "f7": dict(camera_width=7),
    "f10_quaternion_cameras": dict(camera_width=10),
    "f6_s8_libmv_like": dict(camera_width=6, shared_widths=(8,), locked_cameras=(0,)),
    "f6_s3_subset_manifold": dict(camera_width=6, shared_widths=(3,), locked_cameras=(0, 5)),
    "f9_s3_shared_last": dict(camera_width=9, shared_widths=(3,), shared_first=False),
    "f9_s8_two_shared_blocks": dict(camera_width=9, shared_widths=(5, 3)),
}


def cgnr_ambiguous(name):
    return SHAPES[name]["camera_width"] == 3 or 3 in SHAPES[name].get("shared_widths", ())


def shaped(problems, name, layout="schur", seed=5, nc=40, npts=2500, nobs=11000):
    return problems.synthetic_structured(nc, npts, nobs, layout=layout, seed=seed, skew=0.5, **SHAPES[name])


@pytest.mark.parametrize("name", list(SHAPES))
def test_operators_of_every_shape_on_the_fused_path(hip, oracle, problems, name):
    p = shaped(problems, name)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    # CGNR sees no elimination order: points are the 3-wide blocks, so a 3-wide camera or shared block cannot be told from a point
    # (generic path)
    cgnr_path = hip.PATH_GENERIC if cgnr_ambiguous(name) else hip.PATH_BAL
    q = shaped(problems, name, layout="cgnr")
    assert_errs(check_cgnr_operators(hip, oracle, q, False, cgnr_path))
    # and CGNR on the Schur-ordered Jacobian (what a sharded run uses)
    assert_errs(check_cgnr_operators(hip, oracle, p, False, cgnr_path))


@pytest.mark.parametrize("name", ["f10_quaternion_cameras", "f6", "f6_s8_libmv_like", "f9_s8_two_shared_blocks", "f3", "f8"])
@pytest.mark.parametrize("solver_type,pre", [(5, 2), (5, 1), (6, 1)])
def test_solvers_of_every_shape(hip, oracle, problems, name, solver_type, pre):
    if solver_type == hip.CGNR and cgnr_ambiguous(name):
        pytest.skip("CGNR cannot tell 3-wide cameras / shared blocks from points")
    p = shaped(problems, name, seed=6)
    if solver_type == hip.CGNR:
        p = type(p)(p.bs, p.values, p.b, p.D, 0)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m.cgnr_solve
    for k in (1, 7, 25):   # rung (2): fixed iteration counts (25 crosses two residual resets)
        s = make_solver(hip, p, solver_type, pre, min_it=k, max_it=k)
        assert s.info().kernel_path == hip.PATH_BAL
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=0.0))
        s.close()
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=k, max_it=k, q_tol=-1.0, r_tol=0.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        assert rel(x, xo) <= 1e-9, (k, rel(x, xo))
    # rung (4): the call LevenbergMarquardtStrategy issues
    s = make_solver(hip, p, solver_type, pre, max_it=500)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, q, r: fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    # the whole LM step on the device (diag(J'J) by the fused column-norm pass incl. the strip's columns), and the retry after a rejection
    radius = 1e4
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
    diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
    assert rel(s.lm_diagonal(), np.sqrt(diag / radius)) <= 1e-13
    check_step(oracle, hip, p, solver_type, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1)
    step, summ, mcc = s.lm_compute_step(None, None, radius / 2, 0.1, reuse_diagonal=True, values_unchanged=True)
    check_step(oracle, hip, p, solver_type, pre, np.sqrt(diag / (radius / 2)), step, summ, mcc, 0.1)
    s.close()


def test_libmv_structure_on_the_real_visibility_graph(hip, oracle, problems):
    """examples/libmv_bundle_adjuster.cc on data/libmv-ba-problems/problem_02.bin's visibility: shared intrinsics (8) + 6-wide pose +
    point, first camera constant; every point has far more than 64 observations (long points: rounds) — on the fused path."""
    p = problems.libmv_structured(2, 1)
    errs = check_schur_operators(hip, oracle, p, False, hip.PATH_BAL)
    assert_errs(errs)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
    info = s.info()
    assert info.kernel_path == hip.PATH_BAL and (info.row_block_size, info.e_block_size, info.f_block_size) == (2, 3, -1)   # DetectStructure: dynamic F
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, q, r: m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    s.close()
    # a SubsetManifold leaves a narrower intrinsics block (:754-771): tangent size 3 (focal length, k1, k2)
    p3 = problems.libmv_structured(2, 1, intrinsics_width=3)
    assert_errs(check_schur_operators(hip, oracle, p3, False, hip.PATH_BAL))
    # the CGNR solver on the same Jacobian
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))


def test_narrow_cameras_beyond_lds(hip, oracle, problems):
    # 30 000 6-wide cameras: the accumulators do not fit in LDS (hybrid plan: popular cameras + windows, the rest spilled to the ring)
    p = problems.synthetic_structured(30000, 60000, 200000, camera_width=6, seed=8, skew=0.4)
    errs = check_schur_operators(hip, oracle, p, False, hip.PATH_BAL)
    raw = errs.pop("schur_jacobi_raw")   # max-norm over 30 000 blocks of cameras with a handful of observations: 1.3e-12 on the worst one
    assert raw <= 1e-11, raw
    assert_errs(errs)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().camera_accum_in_lds == 0
    s.close()
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))


def test_strip_and_locked_cameras_beyond_lds(hip, oracle, problems):
    """More cameras than LDS rows TOGETHER with a shared block and rows without a camera cell: the strip's sums go through the global
    accumulator (one atomic per wave and scalar), a slot without a camera cell is neither summed nor spilled."""
    p = problems.synthetic_structured(30000, 60000, 200000, camera_width=6, shared_widths=(8,), locked_cameras=(0, 7, 29999), seed=10, skew=0.4)
    errs = check_schur_operators(hip, oracle, p, False, hip.PATH_BAL)
    raw = errs.pop("schur_jacobi_raw")
    assert raw <= 1e-11, raw
    assert_errs(errs)
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))
    # the libmv structure on the replicated real visibility: 8 800 cameras x 6 + the intrinsics, every point a long one (rounds, hybrid groups)
    q = problems.libmv_structured(2, 20)
    s = make_solver(hip, q, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
    info = s.info()
    assert info.kernel_path == hip.PATH_BAL and info.camera_accum_in_lds == 0
    m = oracle.Matrix(q.bs, q.num_eliminate_blocks)
    x, summ = s.solve(q.values, q.b, hip.PerSolveOptions(D=q.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, qq, r: m.iterative_schur_solve(q.values, q.b, q.D, preconditioner=2, min_it=lo, max_it=hi, q_tol=qq, r_tol=r), 0.1, hip.SUCCESS)
    s.close()
    errs = check_schur_operators(hip, oracle, q, False, hip.PATH_BAL)
    errs.pop("schur_jacobi_raw")
    assert_errs(errs, 1e-11)


def test_unsupported_widths_fall_back_to_the_generic_path(hip, oracle, problems):
    p = problems.synthetic_structured(12, 200, 900, camera_width=11, seed=9)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_GENERIC))


# ---- point blocks that are not 3 wide (round 5): the reference's (2,2,*) and (2,4,*) specialisations
# (internal/ceres/generate_template_specializations.py:55-75) on the fused path — every (E, F) pair of that list
# (round 6: + the camera widths of (2,2,d) / (2,4,d) that are compiled statically: 6 and 9 next to 2-wide points, 2, 5, 7, 10 next to 4-wide ones)
POINT_SHAPES = {f"e{ne}_f{nf}": dict(point_width=ne, camera_width=nf) for ne, nf in ((2, 2), (2, 3), (2, 4), (4, 3), (4, 4), (4, 6), (4, 8), (4, 9),
                                                                                   (2, 6), (2, 9), (4, 2), (4, 5), (4, 7), (4, 10))}


@pytest.mark.parametrize("name", list(POINT_SHAPES))
def test_operators_with_point_blocks_of_2_and_4(hip, oracle, problems, name):
    """ImplicitSchurComplement / SchurEliminator / PartitionedMatrixView operators of the Schur solvers on the fused kernels at 1e-12;
    DetectStructure reports the reference's static triple.  CGNR (no Schur solver): the generic kernels."""
    kw = POINT_SHAPES[name]
    p = problems.synthetic_structured(40, 2500, 11000, seed=5, skew=0.5, **kw)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    info = s.info()
    assert (info.row_block_size, info.e_block_size, info.f_block_size) == (2, kw["point_width"], kw["camera_width"])
    s.close()
    # CGNR knows no elimination order: "points" are whatever is 3 wide.  (2,4,3) / (2,2,3): the 3-wide cameras take the points' role and the
    # 4- / 2-wide points the cameras' — a <2,3,4> / <2,3,2> plan, fused (<2,3,2> since round 6); everything else here has no such reading
    # and runs on the generic kernels.
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL if kw["camera_width"] == 3 else hip.PATH_GENERIC))


@pytest.mark.parametrize("name", ["e4_f9", "e4_f6", "e2_f3", "e2_f2", "e4_f3"])
@pytest.mark.parametrize("pre", [2, 1])
def test_schur_solver_with_point_blocks_of_2_and_4(hip, oracle, problems, name, pre):
    """ITERATIVE_SCHUR (SCHUR_JACOBI / JACOBI) on rungs (2) and (4) of the parity ladder, the device LM step and its retry."""
    p = problems.synthetic_structured(40, 2500, 11000, seed=6, skew=0.5, **POINT_SHAPES[name])
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    fn = m.iterative_schur_solve
    for k in (1, 7, 25):
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, pre, min_it=k, max_it=k)
        assert s.info().kernel_path == hip.PATH_BAL
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=0.0))
        s.close()
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=k, max_it=k, q_tol=-1.0, r_tol=0.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        assert rel(x, xo) <= 1e-9, (k, rel(x, xo))
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, pre, max_it=500)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, q, r: fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    radius = 1e4
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
    diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
    assert rel(s.lm_diagonal(), np.sqrt(diag / radius)) <= 1e-13
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1)
    step, summ, mcc = s.lm_compute_step(None, None, radius / 2, 0.1, reuse_diagonal=True, values_unchanged=True)
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, pre, np.sqrt(diag / (radius / 2)), step, summ, mcc, 0.1)
    s.close()


def test_homogeneous_points_with_long_tracks_and_many_cameras(hip, oracle, problems):
    """(2,4,9) where the plain tile walk is not enough: points of more than 64 observations (whole tiles, rounds) on the libmv visibility
    graph, and more cameras than LDS rows (hybrid plan, spilled rows)."""
    n_c, n_p, cam_of, pt_of = problems.libmv_visibility(2)
    order = np.lexsort((cam_of, pt_of))
    p = problems.structured_bal(n_c, n_p, pt_of[order], cam_of[order], 9, point_width=4, seed=3)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    # (tracks of at least THREE observations: two rows of a 4-wide point make its E square, E^T E + D^2 is then inverted at a condition
    # number of 1e4 .. 1e6 and M_o = I - E (E^T E + D^2)^-1 E^T is a difference of nearly equal numbers — the blocks of a camera that sees
    # only such points agree with the oracle to 3e-10 instead of 1e-12; a point needs three views to be determined up to scale anyway)
    rng = np.random.default_rng(8)
    k = 3 + np.minimum(rng.geometric(0.75, size=60000) - 1, 20)
    pt_of = np.repeat(np.arange(60000, dtype=np.int64), k)
    w = np.arange(1, 30001, dtype=np.float64) ** -0.4
    cam_of = problems._distinct_cameras(rng, 30000, pt_of, w / w.sum())
    order = np.lexsort((cam_of, pt_of))
    q = problems.structured_bal(30000, 60000, pt_of[order], cam_of[order], 6, point_width=4, seed=9)
    errs = check_schur_operators(hip, oracle, q, False, hip.PATH_BAL)
    raw = errs.pop("schur_jacobi_raw")
    assert_errs(errs)
    assert raw <= 1e-11, raw
    s = make_solver(hip, q, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().camera_accum_in_lds == 0
    s.close()


def test_dense_schur_with_homogeneous_points(hip, oracle, problems):
    """DENSE_SCHUR on a (2,4,6) problem: the fused set-up pass's packed 4 x 4 inverses, expanded into the eliminator's dense E-block
    store, feed SchurEliminator::Eliminate; the solve is the damped least-squares solution."""
    p = problems.synthetic_structured(24, 900, 4000, camera_width=6, point_width=4, seed=12, skew=0.3)
    A = p.bs.to_dense(p.values)
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1))
    s.set_structure(p.bs)
    assert s.info().kernel_path == hip.PATH_BAL
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))
    s.close()
    assert summ.termination_type == hip.SUCCESS and rel(x, ref) <= 1e-9, (summ, rel(x, ref))


# ---- row blocks that are not 2 high (round 5): the reference's remaining static specialisations (3,3,3), (4,4,2), (4,4,3), (4,4,4)
ROW_SHAPES = {f"r{nr}_e{ne}_f{nf}": dict(row_height=nr, point_width=ne, camera_width=nf) for nr, ne, nf in ((3, 3, 3), (4, 4, 2), (4, 4, 3), (4, 4, 4))}


@pytest.mark.parametrize("name", list(ROW_SHAPES))
def test_operators_with_rows_of_3_and_4_residuals(hip, oracle, problems, name):
    kw = ROW_SHAPES[name]
    p = problems.synthetic_structured(40, 2500, 11000, seed=5, skew=0.5, **kw)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    info = s.info()
    assert (info.row_block_size, info.e_block_size, info.f_block_size) == (kw["row_height"], kw["point_width"], kw["camera_width"])
    s.close()
    # a row with a locked camera, tracks of more than 64 observations (whole tiles, rounds)
    n_c, n_p, cam_of, pt_of = problems.libmv_visibility(2)
    order = np.lexsort((cam_of, pt_of))
    q = problems.structured_bal(n_c, n_p, pt_of[order], cam_of[order], kw["camera_width"], locked_cameras=(0,), point_width=kw["point_width"],
                                row_height=kw["row_height"], seed=3)
    assert_errs(check_schur_operators(hip, oracle, q, False, hip.PATH_BAL))


@pytest.mark.parametrize("name", list(ROW_SHAPES))
@pytest.mark.parametrize("pre", [2, 1])
def test_schur_solver_with_rows_of_3_and_4_residuals(hip, oracle, problems, name, pre):
    p = problems.synthetic_structured(40, 2500, 11000, seed=6, skew=0.5, **ROW_SHAPES[name])
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    fn = m.iterative_schur_solve
    for k in (1, 7, 25):
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, pre, min_it=k, max_it=k)
        assert s.info().kernel_path == hip.PATH_BAL
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=0.0))
        s.close()
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=k, max_it=k, q_tol=-1.0, r_tol=0.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        assert rel(x, xo) <= 1e-9, (k, rel(x, xo))
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, pre, max_it=500)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
    assert_lm_style_step(x, summ, lambda lo, hi, q, r: fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r), 0.1, hip.SUCCESS)
    radius = 1e4
    step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
    diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
    assert rel(s.lm_diagonal(), np.sqrt(diag / radius)) <= 1e-13
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1)
    step, summ, mcc = s.lm_compute_step(None, None, radius / 2, 0.1, reuse_diagonal=True, values_unchanged=True)
    check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, pre, np.sqrt(diag / (radius / 2)), step, summ, mcc, 0.1)
    s.close()


def test_rows_of_4_residuals_with_many_cameras_and_dense_schur(hip, oracle, problems):
    """(4,4,3): more cameras than LDS rows (hybrid plan, spilled rows) and DENSE_SCHUR on the fused set-up passes."""
    q = problems.synthetic_structured(60000, 60000, 200000, camera_width=3, point_width=4, row_height=4, seed=8, skew=0.4)
    errs = check_schur_operators(hip, oracle, q, False, hip.PATH_BAL)
    raw = errs.pop("schur_jacobi_raw")
    assert_errs(errs)
    assert raw <= 1e-11, raw
    s = make_solver(hip, q, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().camera_accum_in_lds == 0
    s.close()
    p = problems.synthetic_structured(24, 900, 4000, camera_width=4, point_width=4, row_height=4, seed=12, skew=0.3)
    A = p.bs.to_dense(p.values)
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1))
    s.set_structure(p.bs)
    assert s.info().kernel_path == hip.PATH_BAL
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))
    s.close()
    assert summ.termination_type == hip.SUCCESS and rel(x, ref) <= 1e-9, (summ, rel(x, ref))


# ---- round 5: the LDS copies of x (the popular cameras' part in the streaming kernels, all of it in back-substitution / the model-cost
# pass) switch on from about a hundred tiles per workgroup; the shape tests above are far smaller — here every kind of shape once at 1.8 M
# observations, so that those paths run for camera widths other than 9, with a shared strip, and for the shapes whose S.x runs on
# bal_fused_kernel (point blocks 4 wide, rows 4 high)
BIG_SHAPES = {"f10_quaternion_cameras": dict(camera_width=10), "f6_s8_libmv_like": dict(camera_width=6, shared_widths=(8,), locked_cameras=(0,)),
              "f3": dict(camera_width=3), "e4_f9": dict(point_width=4, camera_width=9), "r4_e4_f4": dict(row_height=4, point_width=4, camera_width=4)}


@pytest.mark.parametrize("name", list(BIG_SHAPES))
def test_shapes_at_a_size_where_x_is_read_from_lds(hip, oracle, problems, name):
    import os
    oracle.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        kw = BIG_SHAPES[name]
        p = problems.synthetic_structured(900, 420000, 1800000, seed=41, skew=0.7, **kw)
        assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
        assert s.info().camera_accum_in_lds == 1
        radius = 1e4
        step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)   # back-substitution + model cost inside the step
        diag = np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 1e-6, 1e32)
        check_step(oracle, hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, np.sqrt(diag / radius), step, summ, mcc, 0.1)
        s.close()
        if name == "f10_quaternion_cameras":
            assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))
    finally:
        oracle.set_num_threads(1)
