"""REAL visibility on the GPU: the three bundle-adjustment problems the reference ships (data/libmv-ba-problems/problem_0{1,2,3}.bin, read by
examples/libmv_bundle_adjuster.cc; committed as tests/golden/libmv_problems.npz) — 26-71 tracks followed through 333-500 consecutive frames,
so EVERY point has more than 64 observations (it owns whole tiles: the two-sweep long-point path of the fused kernels) and neighbouring
cameras see the same points.  Values are N(0,1) like the synthetic workloads; every operator of both solvers and the LM-style solves are
checked against the oracle (tests/test_gpu_fullsize.py's checkers).  Replicated side by side the cameras outgrow LDS: the hybrid plan then
keeps (nearly) whole tracks in one workgroup's window — windows of consecutive camera ids that overlap."""
import numpy as np
import pytest

from test_gpu_fullsize import check_cgnr_side, check_schur_side, oracle_threads  # noqa: F401  (module fixture)
from test_gpu_operators import make_solver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("problem", [1, 2, 3])
def test_libmv_problem_against_the_oracle(hip, oracle, problems, problem):
    p = problems.libmv_bal(problem, 1)
    track = np.bincount(p.point_of_row)
    assert (track > 64).mean() > 0.75 and track.max() >= 333    # (nearly) all points are long ones
    check_schur_side(hip, oracle, p, True)
    check_cgnr_side(hip, oracle, p, True)


@pytest.mark.parametrize("problem,copies", [(2, 6), (3, 8)])
def test_replicated_libmv_problem_in_the_hybrid_regime(hip, oracle, problems, problem, copies):
    """2640 / 4000 cameras: more than LDS holds.  No camera is popular here (every row of a workgroup goes to its window), windows
    overlap, and a copy's cameras fit one window: (nearly) every observation is summed in LDS — against 56 % on random visibility."""
    p = problems.libmv_bal(problem, copies)
    for typ, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        s = make_solver(hip, p, typ, pre, max_it=500)
        info = s.info()
        assert info.kernel_path == hip.PATH_BAL and info.camera_accum_in_lds == 0 and info.camera_accum_hybrid == 1
        assert info.points_renumbered == 1 and info.hybrid_popular_rows == 0
        assert info.num_observations_in_lds >= 0.9 * info.num_observations, (info.num_observations_in_lds, info.num_observations)
        s.close()
    check_schur_side(hip, oracle, p, False)
    check_cgnr_side(hip, oracle, p, False)


def test_many_camera_regime_is_hybrid_with_popular_cameras(hip, problems):
    """The synthetic many-camera shape: skewed popularity -> the popular cameras take most rows, windows do not overlap; more than half of
    the observations stay in LDS (the parity checks of this shape are test_gpu_fullsize.py::test_many_camera_regime_against_the_oracle)."""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=50000, num_points=400000, num_observations=1200000, seed=38401, skew=0.6,
                               with_values=False)
    for typ, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        s = hip.HipLinearSolver(hip.LinearSolverOptions(type=typ, preconditioner_type=pre, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=10))
        s.set_structure(p.bs)
        info = s.info()
        assert info.camera_accum_hybrid == 1 and info.hybrid_popular_rows > 1000
        assert 0.5 * info.num_observations < info.num_observations_in_lds < 0.7 * info.num_observations
        s.close()


def test_preconditioner_without_residuals_in_the_hybrid_regime(hip, oracle, problems):
    """Preconditioner::Update needs no residual vector: Init without b scatters nothing, but its tile pass still walks the hybrid groups
    (one workgroup per group — a grid sized for the non-scattering passes once indexed past the group table here)."""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=3000, num_points=12000, num_observations=60000, seed=4, skew=0.6)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().camera_accum_hybrid == 1
    s.load(p.values, None, p.D)
    s.schur_jacobi_update()
    inv, raw = m.schur_jacobi(p.values, p.D)
    assert np.linalg.norm(s.preconditioner_blocks() - inv) <= 1e-11 * np.linalg.norm(inv)
    s.close()


MIXED_TRACKS = ([3, 70, 2, 2, 129, 64, 65, 5, 513, 1, 300, 512, 7, 449, 200, 100, 66] + [4] * 50 + [90, 1000, 3] + [2, 9, 130] * 40 +
                [640, 65, 448, 384, 1, 1, 63, 64, 65])


@pytest.mark.parametrize("cameras,expect_lds", [(1100, True), (2600, False)])
def test_every_track_length_against_the_oracle(hip, oracle, problems, cameras, expect_lds):
    """Points of 1 .. 1000 observations side by side: normal tiles (pipelined), points of 2 .. 8 tiles (taken in ROUNDS: one tile per
    wave of a workgroup, tile sums exchanged through LDS; rounds with idle waves among them) and points of more than 8 tiles (rounds of
    their own: sum, then apply) in one plan — with every camera's accumulator in LDS, and in the hybrid regime (rounds per group)."""
    p = problems.bal_from_tracks(MIXED_TRACKS, cameras, seed=11)
    r = hip.debug_long_rounds(p.bs, p.num_eliminate_blocks, True)
    kinds = np.bincount(r["tile_kind"], minlength=4)
    assert kinds[3] >= 50 and kinds[1] == 0 and (r["round_word"] == 0xFFFFFFFF).any() and (np.diff(r["seq_ptr"]) > 1).sum() == 3
    check_schur_side(hip, oracle, p, expect_lds)
    check_cgnr_side(hip, oracle, p, expect_lds)
    # the same problem with its columns in CGNR's caller order (points not renumbered: the long points' tiles still move behind)
    check_cgnr_side(hip, oracle, problems.bal_from_tracks(MIXED_TRACKS, cameras, layout="cgnr", seed=11), expect_lds)


def test_every_track_length_with_fp32_tiles(hip, oracle, problems):
    """The same mix of track lengths with the tiles rounded to fp32 (jacobian_storage = 1; round 5: the pipelined kernels take fp32 tiles too — CERES_HIP_F32_PIPELINE=0 for the unpipelined ones, whose long points
    run the rounds of `fused_long_rounds` — also for S.x and JtJx).  An accuracy mode: exact against the oracle on the fp32-rounded
    Jacobian, ~1e-7 against the fp64 one."""
    p = problems.bal_from_tracks(MIXED_TRACKS, 1100, seed=11)
    rounded = type(p)(p.bs, p.values.astype(np.float32).astype(np.float64), p.b, p.D, p.num_eliminate_blocks)
    rng = np.random.default_rng(2)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for solver_type, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=200,
                                    elimination_groups=[p.num_eliminate_blocks], jacobian_storage=1)
        s = hip.HipLinearSolver(o)
        s.set_structure(p.bs)
        assert s.info().kernel_path == hip.PATH_BAL
        s.load(p.values, p.b, p.D)
        if solver_type == hip.ITERATIVE_SCHUR:
            s.schur_init()
            x = rng.standard_normal(m.num_cols_f)
            got, rhs, back = s.schur_sx(x), s.schur_rhs(), s.back_substitute(x)
            for prob, tol in ((p, 2e-6), (rounded, 1e-11)):
                isc = oracle.ImplicitSchurComplement(m)
                isc.init(prob.values, prob.D, prob.b)
                assert rel(got, isc.sx(x)) <= tol and rel(rhs, isc.rhs()) <= tol and rel(back, isc.back_substitute(x)) <= tol
        else:
            x = rng.standard_normal(m.num_cols)
            got, jtb = s.jtjx(x), s.jtb()
            for prob, tol in ((p, 2e-6), (rounded, 1e-11)):
                assert rel(got, m.left_multiply(prob.values, m.right_multiply(prob.values, x)) + p.D ** 2 * x) <= tol
                assert rel(jtb, m.left_multiply(prob.values, prob.b)) <= tol
        s.close()
