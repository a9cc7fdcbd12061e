"""the explicit Schur complement solvers.

ITERATIVE_SCHUR with use_explicit_schur_complement — S formed by elimination into the block-sparse storage of
BlockRandomAccessSparseMatrix with the block pairs of SparseSchurComplementSolver::InitStorage (dense storage when
sharded), SCHUR_JACOBI from its diagonal blocks, CG with SymmetricRightMultiplyAndAccumulate, back-substitution on SUCCESS
(internal/ceres/schur_complement_solver.cc:100-158, 224-290, 337-408; block_random_access_sparse_matrix.cc:51-163) — and
DENSE_SCHUR (DenseSchurComplementSolver, :163-222).  Checked against the oracle's SchurEliminator + its CG on the dense
reduced system, dense algebra, and the implicit solver."""
import numpy as np
import pytest

from test_gpu_operators import rel

pytestmark = pytest.mark.gpu


def explicit_options(hip, nelim, **kw):
    return hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[nelim],
                                   use_explicit_schur_complement=True, min_num_iterations=kw.pop("min_it", 0),
                                   max_num_iterations=kw.pop("max_it", 200), **kw)


def oracle_explicit(oracle, p, min_it, max_it, q_tol, r_tol):
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    lhs, rhs = m.schur_eliminate(p.values, p.b, p.D)
    nf = rhs.shape[0]
    S = np.triu(lhs.reshape(nf, nf))
    S = S + np.triu(S, 1).T
    sizes = p.bs.col_block_size[p.num_eliminate_blocks:]
    Minv = np.zeros_like(S)
    o = 0
    for n in sizes:
        Minv[o:o + n, o:o + n] = np.linalg.inv(S[o:o + n, o:o + n])
        o += n
    z, summ = oracle.cg_dense(S, rhs, Minv=Minv, min_it=min_it, max_it=max_it, q_tol=q_tol, r_tol=r_tol)
    x = m.schur_back_substitute(p.values, p.b, p.D, z)
    return x, summ.num_iterations, summ.termination_type


def solve(hip, p, o, q_tol, r_tol, **kw):
    s = hip.HipLinearSolver(o, **kw)
    s.set_structure(p.bs)
    if o.use_explicit_schur_complement:
        assert s.info().kernel_path == hip.PATH_GENERIC
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
    s.close()
    return x, summ


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_known_answer_problems(hip, oracle, problems, pid):
    p = problems.linear_least_squares_problem(pid)
    A = p.bs.to_dense(p.values)
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=p.num_cols), -1.0, 1e-12)
    assert s.termination_type == hip.SUCCESS
    assert np.linalg.norm(x - ref) < 1e-12 * max(1.0, np.linalg.norm(ref))


@pytest.mark.parametrize("kind", ["bal", "general"])
def test_fixed_iterations_match_oracle_explicit_solver(hip, oracle, problems, kind):
    if kind == "bal":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=14, num_points=500, num_observations=2400, seed=31)
    else:
        p = problems.random_schur_problem(num_e_blocks=60, num_f_blocks=9, seed=4)
    K = 5
    xo, its, _ = oracle_explicit(oracle, p, K, K, -1.0, -1.0)
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, min_it=K, max_it=K), -1.0, -1.0)
    assert s.num_iterations == K == its
    # NO_CONVERGENCE after max iterations: no back-substitution (schur_complement_solver.cc:150-154), x = [0; reduced solution]
    assert s.termination_type == hip.NO_CONVERGENCE
    ne = p.bs.col_block_pos[p.num_eliminate_blocks]
    assert np.all(x[:ne] == 0.0)
    assert rel(x[ne:], xo[ne:]) <= 1e-9


def test_converged_explicit_equals_implicit_and_oracle(hip, oracle, problems):
    p = problems.synthetic_bal(None, layout="schur", num_cameras=14, num_points=500, num_observations=2400, seed=32)
    xo, its, term = oracle_explicit(oracle, p, 0, 500, -1.0, 1e-13)
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=500), -1.0, 1e-13)
    assert s.termination_type == hip.SUCCESS and abs(s.num_iterations - its) <= 1
    assert rel(x, xo) <= 1e-9
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[p.num_eliminate_blocks],
                                min_num_iterations=0, max_num_iterations=500)
    xi, si = solve(hip, p, o, -1.0, 1e-13)
    assert rel(x, xi) <= 1e-9 and abs(si.num_iterations - s.num_iterations) <= 1


def test_sharded_elimination_in_loopback(hip, oracle, problems):
    # world > 1 branches (D_f^2 on one rank only, all-reduce of S) through the 1-rank communicator
    p = problems.synthetic_bal(None, layout="schur", num_cameras=10, num_points=300, num_observations=1400, seed=33)
    x1, s1 = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=300), -1.0, 1e-13)
    x2, s2 = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=300), -1.0, 1e-13, loopback_world=2)
    assert s1.termination_type == s2.termination_type == hip.SUCCESS
    assert rel(x2, x1) <= 1e-12


def test_option_validation(hip, problems):
    with pytest.raises(hip.HipError):   # "Only SCHUR_JACOBI is supported"
        hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.JACOBI, max_num_iterations=5,
                                                    use_explicit_schur_complement=True))
    with pytest.raises(hip.HipError):
        hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.CGNR, preconditioner_type=hip.JACOBI, max_num_iterations=5,
                                                    use_explicit_schur_complement=True))
    p = problems.synthetic_bal(None, layout="schur", num_cameras=1000, num_points=3000, num_observations=12000, seed=1, with_values=False)
    s = hip.HipLinearSolver(explicit_options(hip, p.num_eliminate_blocks), loopback_world=2)
    with pytest.raises(hip.HipError):   # sharded runs store S densely: 9000 reduced columns > 8192
        s.set_structure(p.bs)
    s.close()
    s = hip.HipLinearSolver(explicit_options(hip, p.num_eliminate_blocks))   # one rank: block-sparse storage, no such limit
    s.set_structure(p.bs)
    s.close()
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1))
    with pytest.raises(hip.HipError):   # DENSE_SCHUR is dense by definition
        s.set_structure(p.bs)
    s.close()


def reference_block_pairs(bs, nelim):
    """SparseSchurComplementSolver::InitStorage (schur_complement_solver.cc:224-290), restated on the flattened structure."""
    nf = bs.num_col_blocks - nelim
    pairs = {(i, i) for i in range(nf)}
    ptr, col = bs.row_cell_ptr, bs.cell_col_block
    r = 0
    n = bs.num_row_blocks
    while r < n:
        if ptr[r] == ptr[r + 1] or col[ptr[r]] >= nelim:
            break
        e = col[ptr[r]]
        f = set()
        while r < n and ptr[r] < ptr[r + 1] and col[ptr[r]] == e:
            f.update(int(c) - nelim for c in col[ptr[r] + 1:ptr[r + 1]])
            r += 1
        f = sorted(f)
        pairs.update((f[a], f[b]) for a in range(len(f)) for b in range(a + 1, len(f)))
    for r in range(r, n):
        cs = [int(c) - nelim for c in col[ptr[r]:ptr[r + 1]]]
        pairs.update((a, b) for a in cs for b in cs if a <= b)
    return sorted(pairs)


@pytest.mark.parametrize("kind", ["lsq2", "lsq4", "lsq6", "general", "bal"])
def test_block_sparse_storage_eliminate_and_symmetric_multiply(hip, oracle, problems, kind):
    if kind.startswith("lsq"):
        p = problems.linear_least_squares_problem(int(kind[3]))
    elif kind == "general":
        p = problems.random_schur_problem(num_e_blocks=40, num_f_blocks=12, num_no_e_rows=4, seed=8)
    else:
        p = problems.synthetic_bal(None, layout="schur", num_cameras=60, num_points=400, num_observations=1500, seed=34)
    nelim = p.num_eliminate_blocks
    s = hip.HipLinearSolver(explicit_options(hip, nelim))
    s.set_structure(p.bs)
    s.load(p.values, p.b, p.D)
    pi, pj, off, vals = s.schur_eliminate_sparse()
    # the structure is InitStorage's
    assert list(zip(pi.tolist(), pj.tolist())) == reference_block_pairs(p.bs, nelim)
    # the values are SchurEliminator::Eliminate's (oracle, dense lhs: upper block triangle), and the oracle has nothing outside the pairs
    m = oracle.Matrix(p.bs, nelim)
    lhs, _ = m.schur_eliminate(p.values, p.b, p.D)
    sizes = p.bs.col_block_size[nelim:].astype(int)
    pos = np.concatenate([[0], np.cumsum(sizes)])
    scale = np.abs(lhs).max()
    covered = np.zeros_like(lhs, dtype=bool)
    S = np.zeros_like(lhs)
    for i, j, o in zip(pi, pj, off):
        ni, nj = sizes[i], sizes[j]
        blk = vals[o:o + ni * nj].reshape(ni, nj)
        want = lhs[pos[i]:pos[i] + ni, pos[j]:pos[j] + nj]
        if i == j:   # only the upper triangle of a diagonal cell is authoritative in the reference
            assert np.abs(np.triu(blk) - np.triu(want)).max() <= 1e-12 * scale, (i, j)
            assert np.abs(blk - blk.T).max() <= 1e-12 * scale
        else:
            assert np.abs(blk - want).max() <= 1e-12 * scale, (i, j)
            S[pos[j]:pos[j] + nj, pos[i]:pos[i] + ni] = blk.T
        S[pos[i]:pos[i] + ni, pos[j]:pos[j] + nj] = blk
        covered[pos[i]:pos[i] + ni, pos[j]:pos[j] + nj] = True
    upper = np.triu(np.ones_like(lhs, dtype=bool))
    assert np.abs(lhs[upper & ~covered]).max(initial=0.0) == 0.0
    # SymmetricRightMultiplyAndAccumulate against the dense symmetric matrix (block_random_access_sparse_matrix_test.cc pattern)
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(S.shape[0]), rng.standard_normal(S.shape[0])
    assert rel(s.schur_symmetric_multiply(x, y0), y0 + S @ x) <= 1e-13
    s.close()


def dense_reference(p):
    A = p.bs.to_dense(p.values)
    D = p.D if p.D is not None else np.zeros(p.num_cols)
    return np.linalg.lstsq(np.vstack([A, np.diag(D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]


@pytest.mark.parametrize("kind", ["lsq2", "lsq4", "lsq5", "lsq6", "general", "bal", "bal_wide"])
def test_dense_schur_solver(hip, oracle, problems, kind):
    """DenseSchurComplementSolver: the reference tests it against the DENSE_QR solution of the same regularised problem
    (schur_complement_solver_test.cc: ComputeAndCompareSolutions, 1e-10 there)."""
    if kind.startswith("lsq"):
        p = problems.linear_least_squares_problem(int(kind[3]))
    elif kind == "general":
        p = problems.random_schur_problem(num_e_blocks=50, num_f_blocks=11, num_no_e_rows=3, seed=9)
    elif kind == "bal":
        p = problems.synthetic_bal(None, layout="schur", num_cameras=9, num_points=300, num_observations=1300, seed=35)
    else:   # 120 cameras: 1080 reduced columns, 34 panels of the blocked factorisation incl. a ragged last one
        p = problems.synthetic_bal(None, layout="schur", num_cameras=120, num_points=2000, num_observations=9000, seed=36)
    o = hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1)
    s = hip.HipLinearSolver(o)
    s.set_structure(p.bs)
    # <2,3,9> problems: Init / rhs / back-substitution on the fused tile passes; the elimination is a gather per block either way
    assert s.info().kernel_path == (hip.PATH_BAL if kind.startswith("bal") else hip.PATH_GENERIC)
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))
    assert summ.termination_type == hip.SUCCESS and summ.num_iterations == 1, summ
    ref = dense_reference(p) if p.num_cols <= 4000 else None
    if ref is not None:
        assert np.linalg.norm(x - ref) <= 1e-10 * max(1.0, np.linalg.norm(ref))
    # and against the converged implicit solver (any size)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, min_it=0, max_it=2000, q_tol=-1.0, r_tol=1e-14)
    assert rel(x, xo) <= 1e-9
    # LM-style call through the f1 entry point: D from the radius, step negated, model cost change
    step, summ, mcc = s.lm_compute_step(p.values, p.b, 1e4)
    assert summ.termination_type == hip.SUCCESS and mcc > 0 and np.isfinite(step).all()
    s.close()


def test_dense_schur_reports_factorization_failure(hip, problems):
    # S singular: one camera with an all-zero Jacobian and no regularisation
    p = problems.synthetic_bal(None, layout="schur", num_cameras=9, num_points=200, num_observations=900, seed=5)
    cam0 = int(p.camera_of_row.min())
    rows = np.nonzero(p.camera_of_row == cam0)[0]
    fpos = p.bs.cell_value_pos[1::2][rows].astype(np.int64)
    vals = p.values.copy()
    vals[(fpos[:, None] + np.arange(18)[None, :]).reshape(-1)] = 0.0
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1))
    s.set_structure(p.bs)
    x, summ = s.solve(vals, p.b, hip.PerSolveOptions(D=None))
    assert summ.termination_type == hip.FAILURE and "Cholesky" in summ.message, summ
    x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))   # the instance stays usable
    assert summ.termination_type == hip.SUCCESS
    s.close()


def test_explicit_sparse_on_a_wide_problem_matches_implicit(hip, oracle, problems):
    # 1000 cameras = 9000 reduced columns: beyond what the dense storage takes, fine for the block-sparse one
    p = problems.synthetic_bal(None, layout="schur", num_cameras=1000, num_points=3000, num_observations=12000, seed=1)
    x, s = solve(hip, p, explicit_options(hip, p.num_eliminate_blocks, max_it=2000), -1.0, 1e-12)
    assert s.termination_type == hip.SUCCESS
    o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[p.num_eliminate_blocks],
                                min_num_iterations=0, max_num_iterations=2000)
    xi, si = solve(hip, p, o, -1.0, 1e-12)
    assert rel(x, xi) <= 1e-8 and abs(si.num_iterations - s.num_iterations) <= 2


@pytest.mark.parametrize("n", [1, 17, 128, 129, 300, 1000])
def test_dense_cholesky_on_the_matrix_pipe(hip, n):
    """The blocked factorisation behind DENSE_SCHUR (128-wide panels: diagonal block in LDS, substitution below it, trailing update on
    v_mfma_f64_16x16x4_f64) against numpy on random SPD matrices whose LOWER triangle is garbage (the upper one is authoritative, as
    DenseCholesky's callers leave it): sizes around the panel and tile edges."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 3))
    A = M @ M.T + 0.5 * np.eye(n)
    b = rng.standard_normal(n)
    Au = np.triu(A) + np.tril(rng.standard_normal((n, n)), -1)
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[1], max_num_iterations=1))
    x, ms, failed = s.dense_cholesky_solve(Au, b, repeats=1)
    assert not failed
    ref = np.linalg.solve(A, b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref), np.linalg.norm(x - ref) / np.linalg.norm(ref)
    Au[n // 2, n // 2] = -1.0   # not positive definite
    _, _, failed = s.dense_cholesky_solve(Au, b, repeats=1)
    assert failed
    s.close()
