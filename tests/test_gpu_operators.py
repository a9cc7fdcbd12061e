"""Parity of every operator of SURVEY.md §8(a) against the oracle, through the C ABI, on a real
MI355X.  Rung (1) of the parity ladder: operator outputs rel-l2 <= 1e-12 (the reference's own
bar for re-associated sums, internal/ceres/block_sparse_matrix_test.cc:246)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def make_solver(hip, p, solver_type, pre, force_generic=False, **kw):
    nelim = p.num_eliminate_blocks if solver_type == hip.ITERATIVE_SCHUR else p.num_eliminate_blocks
    o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, max_num_iterations=kw.pop("max_it", 50),
                                min_num_iterations=kw.pop("min_it", 0), elimination_groups=[nelim],
                                force_generic_path=force_generic, **kw)
    s = hip.HipLinearSolver(o)
    s.set_structure(p.bs)
    return s


def check_schur_operators(hip, oracle, p, force_generic, expect_path):
    rng = np.random.default_rng(0)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, force_generic)
    assert s.info().kernel_path == expect_path
    s.load(p.values, p.b, p.D)
    errs = {}
    x, yr = rng.standard_normal(m.num_cols), rng.standard_normal(m.num_rows)
    y0 = rng.standard_normal(m.num_rows)
    errs["right_multiply"] = rel(s.right_multiply(x, y0), m.right_multiply(p.values, x, y0))
    c0 = rng.standard_normal(m.num_cols)
    errs["left_multiply"] = rel(s.left_multiply(yr, c0), m.left_multiply(p.values, yr, c0))
    errs["squared_column_norm"] = rel(s.squared_column_norm(), m.squared_column_norm(p.values))
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    s.schur_init()
    errs["schur_rhs"] = rel(s.schur_rhs(), isc.rhs())
    errs["ete_inverse"] = rel(s.ete_inverse(), isc.ete_inverse())
    xf = rng.standard_normal(m.num_cols_f)
    errs["sx"] = rel(s.schur_sx(xf), isc.sx(xf))
    errs["back_substitute"] = rel(s.back_substitute(xf), isc.back_substitute(xf))
    s.schur_jacobi_update()
    inv, raw = m.schur_jacobi(p.values, p.D)
    mine_raw = s.preconditioner_blocks(not_inverted=True)
    # only the upper triangle of each block is authoritative in the reference
    sizes = p.bs.col_block_size[p.num_eliminate_blocks:]
    off, e_up = 0, []
    for n in sizes:
        a, b = mine_raw[off:off + n * n].reshape(n, n), raw[off:off + n * n].reshape(n, n)
        e_up.append(np.abs(np.triu(a) - np.triu(b)).max() / max(np.abs(b).max(), 1e-300))
        off += n * n
    errs["schur_jacobi_raw"] = max(e_up)
    s.schur_jacobi_update()
    errs["schur_jacobi_inv"] = rel(s.preconditioner_blocks(), inv)
    errs["precond_apply"] = rel(s.precond_apply(xf, xf.copy()), oracle.block_diagonal_apply(sizes, inv, xf, xf.copy()))
    s.close()
    # JACOBI for ITERATIVE_SCHUR = blockdiag(F^T F + D_f^2)^-1
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI, force_generic)
    s.load(p.values, p.b, p.D)
    s.block_jacobi_update()
    ftf = m.block_diagonal_ftf(p.values)
    off = 0
    want = []
    Df = p.D[m.num_cols_e:] if p.D is not None else None
    pos = 0
    for n in sizes:
        blk = ftf[off:off + n * n].reshape(n, n).copy()
        if Df is not None:
            blk += np.diag(Df[pos:pos + n] ** 2)
        want.append(np.linalg.inv(blk).reshape(-1))
        off += n * n
        pos += n
    errs["ftf_jacobi_inv"] = rel(s.preconditioner_blocks(), np.concatenate(want))
    s.close()
    return errs


def check_cgnr_operators(hip, oracle, p, force_generic, expect_path):
    rng = np.random.default_rng(1)
    m = oracle.Matrix(p.bs, 0)
    q = type(p)(p.bs, p.values, p.b, p.D, 0)
    s = make_solver(hip, q, hip.CGNR, hip.JACOBI, force_generic)
    assert s.info().kernel_path == expect_path
    s.load(p.values, p.b, p.D)
    errs = {}
    x = rng.standard_normal(m.num_cols)
    want = m.left_multiply(p.values, m.right_multiply(p.values, x)) + (p.D ** 2 * x if p.D is not None else 0)
    errs["jtjx"] = rel(s.jtjx(x), want)
    errs["jtb"] = rel(s.jtb(), m.left_multiply(p.values, p.b))
    s.block_jacobi_update()
    inv, raw = m.block_jacobi(p.values, p.D)
    errs["block_jacobi_inv"] = rel(s.preconditioner_blocks(), inv)
    errs["block_jacobi_raw"] = rel(s.preconditioner_blocks(not_inverted=True), raw)
    s.block_jacobi_update()
    errs["precond_apply"] = rel(s.precond_apply(x), oracle.block_diagonal_apply(p.bs.col_block_size, inv, x))
    a, b = rng.standard_normal(m.num_cols), rng.standard_normal(m.num_cols)
    errs["dot"] = abs(s.dot(a, b) - a @ b) / np.linalg.norm(a) / np.linalg.norm(b)
    errs["axpby"] = rel(s.axpby(1.5, a, -0.25, b), 1.5 * a - 0.25 * b)
    s.close()
    return errs


def assert_errs(errs, tol=TOL):
    bad = {k: v for k, v in errs.items() if not (v <= tol)}
    assert not bad, f"operators above {tol:g}: {bad}   (all: {errs})"


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_known_answer_problems_generic_path(hip, oracle, problems, pid):
    p = problems.linear_least_squares_problem(pid)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_GENERIC))
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_GENERIC))


@pytest.mark.parametrize("case", [dict(seed=1, static_sizes=None), dict(seed=2, static_sizes=(2, 3, 6)),
                                  dict(seed=3, static_sizes=(1, 1, 1)), dict(seed=4, static_sizes=None, num_e_blocks=60, num_f_blocks=11)])
def test_random_structures_generic_path(hip, oracle, problems, case):
    p = problems.random_schur_problem(**case)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_GENERIC))
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_GENERIC))


def test_dense_schur_elimination(hip, oracle, problems):
    # SchurEliminator::Eliminate into a dense lhs + BackSubstitute (schur_eliminator_test.cc:125-222)
    for seed, ss in ((5, None), (6, (2, 3, 9)), (7, (2, 2, 4))):
        p = problems.random_schur_problem(seed=seed, static_sizes=ss, num_e_blocks=12, num_f_blocks=6)
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI, True)
        s.load(p.values, p.b, p.D)
        lhs, rhs = s.schur_eliminate_dense()
        want_lhs, want_rhs = m.schur_eliminate(p.values, p.b, p.D)
        assert np.abs(lhs - want_lhs).max() <= 1e-12 * np.abs(want_lhs).max()
        assert rel(rhs, want_rhs) <= 1e-12
        S = np.triu(want_lhs) + np.triu(want_lhs, 1).T
        z = np.linalg.solve(S, want_rhs)
        assert rel(s.eliminator_back_substitute(z), m.schur_back_substitute(p.values, p.b, p.D, z)) <= 1e-12
        s.close()


BAL_CASES = [
    dict(num_cameras=16, num_points=2000, num_observations=7600, seed=1),               # dubrovnik-like, few cameras
    dict(num_cameras=300, num_points=4000, num_observations=21000, seed=2, skew=0.8),   # skewed camera popularity
    dict(num_cameras=230, num_points=350, num_observations=11000, seed=3),              # long tracks (> 64 obs/point)
]


@pytest.mark.parametrize("case", BAL_CASES)
def test_bal_fused_kernels_schur_layout(hip, oracle, problems, case):
    p = problems.synthetic_bal(None, layout="schur", **case)
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL))
    # the same operators through the generic kernels must agree too (cross-check of both paths)
    assert_errs(check_schur_operators(hip, oracle, p, True, hip.PATH_GENERIC))


@pytest.mark.parametrize("case", BAL_CASES[:2])
@pytest.mark.parametrize("layout", ["cgnr", "schur"])
def test_bal_fused_kernels_cgnr(hip, oracle, problems, case, layout):
    # cgnr layout: cameras and points interleaved in column order, cells sorted by column block
    p = problems.synthetic_bal(None, layout=layout, **case)
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL))
    assert_errs(check_cgnr_operators(hip, oracle, p, True, hip.PATH_GENERIC))


def test_bal_without_regulariser_and_global_accumulators(hip, oracle, problems):
    # D == NULL, and enough cameras (> 2261) that the camera accumulators leave LDS
    p = problems.synthetic_bal(None, num_cameras=2600, num_points=14000, num_observations=70000, seed=4)
    p.D = None
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().kernel_path == hip.PATH_BAL and s.info().camera_accum_in_lds == 0
    s.close()
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL), tol=1e-11)
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL), tol=1e-11)


@pytest.mark.parametrize("chunk_mib", ["1", "0"])
def test_chunked_camera_pass_many_chunks_shared_units_and_long_points(hip, oracle, problems, monkeypatch, chunk_mib):
    """Cameras not in LDS: the tile pass and the camera-major pass run chunk by chunk (csrc/plan.cc, solver.hip::bal_scatter).
    1 MiB chunks = 227 tiles each: a dozen chunks here, points of > 64 observations (whole-tile points, never split by a chunk
    boundary), and popular cameras with more than 64 observations inside one chunk (units combined by atomics).  "0" = one
    chunk for everything (the unchunked reference behaviour of the same kernels)."""
    monkeypatch.setenv("CERES_HIP_Z_CHUNK_MIB", chunk_mib)
    p = problems.synthetic_bal(None, num_cameras=2600, num_points=3000, num_observations=200000, seed=14, skew=1.2)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().kernel_path == hip.PATH_BAL and s.info().camera_accum_in_lds == 0
    s.close()
    assert_errs(check_schur_operators(hip, oracle, p, False, hip.PATH_BAL), tol=1e-11)
    assert_errs(check_cgnr_operators(hip, oracle, p, False, hip.PATH_BAL), tol=1e-11)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for solver_type, pre, fn in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, m.iterative_schur_solve), (hip.CGNR, hip.JACOBI, oracle.Matrix(p.bs, 0).cgnr_solve)):
        s = make_solver(hip, p, solver_type, pre, max_it=300)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=1e-11))
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=300, q_tol=-1.0, r_tol=1e-11)
        assert summ.termination_type == so.termination_type == hip.SUCCESS and rel(x, xo) <= 1e-8
        # the LM step (fused diagonal, model cost) through the same chunked kernels
        step, summ, mcc = s.lm_compute_step(p.values, p.b, 1e4, 0.1)
        m0 = oracle.Matrix(p.bs, 0)
        Dl = np.sqrt(np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32) / 1e4)
        xo, so = fn(p.values, p.b, Dl, preconditioner=pre, min_it=0, max_it=300, q_tol=0.1, r_tol=-1.0)
        assert abs(summ.num_iterations - so.num_iterations) <= 1
        if summ.num_iterations == so.num_iterations:
            assert rel(step, -xo) <= 1e-9
        s.close()


def test_fp32_tile_storage_is_accurate_not_exact(hip, oracle, problems):
    """jacobian_storage = 1: the tiles hold J rounded to fp32, arithmetic stays fp64.  This is an
    ACCURACY mode, never reported as parity: operators agree with the fp64
    oracle to ~1e-7 relative, and exactly with the oracle run on the fp32-rounded Jacobian."""
    p = problems.synthetic_bal(None, num_cameras=60, num_points=4000, num_observations=19000, seed=51, skew=0.5)
    rounded = type(p)(p.bs, p.values.astype(np.float32).astype(np.float64), p.b, p.D, p.num_eliminate_blocks)
    rng = np.random.default_rng(2)
    for solver_type, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=200,
                                    elimination_groups=[p.num_eliminate_blocks], jacobian_storage=1)
        s = hip.HipLinearSolver(o)
        s.set_structure(p.bs)
        assert s.info().kernel_path == hip.PATH_BAL
        s.load(p.values, p.b, p.D)
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
        if solver_type == hip.ITERATIVE_SCHUR:
            s.schur_init()
            x = rng.standard_normal(m.num_cols_f)
            got = s.schur_sx(x)
            for prob, tol in ((p, 5e-7), (rounded, 1e-12)):
                isc = oracle.ImplicitSchurComplement(m)
                isc.init(prob.values, prob.D, prob.b)
                assert rel(got, isc.sx(x)) <= tol
                assert rel(s.schur_rhs(), isc.rhs()) <= tol
        else:
            x = rng.standard_normal(m.num_cols)
            got = s.jtjx(x)
            for prob, tol in ((p, 5e-7), (rounded, 1e-12)):
                want = m.left_multiply(prob.values, m.right_multiply(prob.values, x)) + p.D ** 2 * x
                assert rel(got, want) <= tol
        xs, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=0.1, r_tolerance=-1.0))
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else oracle.Matrix(p.bs, 0).cgnr_solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=200, q_tol=0.1, r_tol=-1.0)
        assert summ.termination_type == hip.SUCCESS and abs(summ.num_iterations - so.num_iterations) <= 1
        if summ.num_iterations == so.num_iterations:
            assert rel(xs, xo) <= 1e-5
        s.close()


@pytest.mark.parametrize("n_cams,n_pts,n_obs,what", [(30000, 60000, 200000, "more cameras than LDS rows: hybrid plan, spilled rows"),
                                                      (1500, 420000, 1800000, "popular cameras' x staged in LDS (a hundred tiles per workgroup)")])
def test_fp32_tiles_through_the_pipelined_kernels(hip, oracle, problems, n_cams, n_pts, n_obs, what):
    """Round 5: fp32 tiles run on the software-pipelined kernels too (CERES_HIP_F32_PIPELINE=0: the unpipelined ones).  Both regimes
    of the camera sums, S.x and JtJx exact (1e-12) against the oracle on the fp32-ROUNDED Jacobian — the arithmetic is fp64 — and
    within 5e-7 of the oracle on the fp64 one (an accuracy mode, never parity)."""
    p = problems.synthetic_bal(None, num_cameras=n_cams, num_points=n_pts, num_observations=n_obs, seed=53, skew=0.5)
    rounded = p.values.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(3)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for solver_type, pre in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)):
        o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=200,
                                    elimination_groups=[p.num_eliminate_blocks], jacobian_storage=1)
        s = hip.HipLinearSolver(o)
        s.set_structure(p.bs)
        info = s.info()
        assert info.kernel_path == hip.PATH_BAL and info.camera_accum_in_lds == int(n_cams <= 2000)
        s.load(p.values, p.b, p.D)
        if solver_type == hip.ITERATIVE_SCHUR:
            s.schur_init()
            x = rng.standard_normal(m.num_cols_f)
            got = s.schur_sx(x)
            for vals, tol in ((p.values, 5e-7), (rounded, 1e-12)):
                isc = oracle.ImplicitSchurComplement(m)
                isc.init(vals, p.D, p.b)
                assert rel(got, isc.sx(x)) <= tol, (what, tol)
        else:
            x = rng.standard_normal(m.num_cols)
            got = s.jtjx(x)
            for vals, tol in ((p.values, 5e-7), (rounded, 1e-12)):
                want = m.left_multiply(vals, m.right_multiply(vals, x)) + p.D ** 2 * x
                assert rel(got, want) <= tol, (what, tol)
        s.close()
