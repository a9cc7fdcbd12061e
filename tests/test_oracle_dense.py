"""KA-7: the dense-algebra constructions of the reference's unit tests re-expressed with
numpy on random inputs (schur_eliminator_test.cc:80-222, implicit_schur_complement_test.cc:126-217,
partitioned_matrix_view_test.cc:103-262, block_jacobi_preconditioner_test.cc:45-95,
block_sparse_matrix_test.cc:196-247).  CPU only."""
import numpy as np
import pytest


def dense_schur(A, b, D, ne):
    """H = A'A + D^2, eliminate the leading ne columns with dense algebra."""
    H = A.T @ A + (np.diag(D ** 2) if D is not None else 0)
    g = A.T @ b
    P, Q, R = H[:ne, :ne], H[:ne, ne:], H[ne:, ne:]
    # P is block diagonal in the E blocks, so inv(P) is what the eliminator applies blockwise
    Pinv = np.linalg.inv(P)
    S = R - Q.T @ Pinv @ Q
    r = g[ne:] - Q.T @ Pinv @ g[:ne]
    return H, g, S, r


CASES = [dict(seed=s, static_sizes=ss, with_D=wd) for s, ss, wd in
         [(1, None, True), (2, None, False), (3, (2, 3, 9), True), (4, (2, 3, 6), True), (5, (1, 1, 1), True), (6, (2, 3, 9), False)]]


@pytest.mark.parametrize("case", CASES)
def test_operators_against_dense(oracle, problems, case):
    p = problems.random_schur_problem(num_e_blocks=9, num_f_blocks=4, **case)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    A = p.bs.to_dense(p.values)
    rng = np.random.default_rng(0)
    x, yr = rng.standard_normal(m.num_cols), rng.standard_normal(m.num_rows)
    ne = m.num_cols_e
    tol = dict(rtol=0, atol=1e-13 * np.abs(A).max() * A.shape[1])
    np.testing.assert_allclose(m.right_multiply(p.values, x), A @ x, **tol)
    np.testing.assert_allclose(m.left_multiply(p.values, yr), A.T @ yr, **tol)
    np.testing.assert_allclose(m.squared_column_norm(p.values), (A * A).sum(0), **tol)
    np.testing.assert_allclose(m.right_multiply_e(p.values, x[:ne]), A[:, :ne] @ x[:ne], **tol)
    np.testing.assert_allclose(m.right_multiply_f(p.values, x[ne:]), A[:, ne:] @ x[ne:], **tol)
    np.testing.assert_allclose(m.left_multiply_e(p.values, yr), A[:, :ne].T @ yr, **tol)
    np.testing.assert_allclose(m.left_multiply_f(p.values, yr), A[:, ne:].T @ yr, **tol)
    scale = 0.5 + rng.random(m.num_cols)
    np.testing.assert_allclose(p.bs.to_dense(m.scale_columns(p.values, scale)), A * scale, rtol=1e-15)
    # accumulate semantics: y += A x
    y0 = rng.standard_normal(m.num_rows)
    np.testing.assert_allclose(m.right_multiply(p.values, x, y0), y0 + A @ x, **tol)
    # block diagonals
    ete = m.block_diagonal_ete(p.values)
    ftf = m.block_diagonal_ftf(p.values)
    H = A.T @ A
    off = 0
    for j in range(p.bs.num_col_blocks):
        s, c = int(p.bs.col_block_size[j]), int(p.bs.col_block_pos[j])
        src = ete if j < p.num_eliminate_blocks else ftf
        if j == p.num_eliminate_blocks:
            off = 0
        np.testing.assert_allclose(src[off:off + s * s].reshape(s, s), H[c:c + s, c:c + s], **tol)
        off += s * s


@pytest.mark.parametrize("case", CASES)
def test_eliminator_and_implicit_complement(oracle, problems, case):
    p = problems.random_schur_problem(num_e_blocks=8, num_f_blocks=5, **case)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    A = p.bs.to_dense(p.values)
    D = p.D if p.D is not None else None
    if D is None:  # keep E'E invertible without regularisation: only use cases where it is
        if np.linalg.matrix_rank(A[:, :m.num_cols_e]) < m.num_cols_e:
            pytest.skip("rank-deficient E without D")
    H, g, S, r = dense_schur(A, p.b, D, m.num_cols_e)
    lhs, rhs = m.schur_eliminate(p.values, p.b, D)
    Sfull = np.triu(lhs) + np.triu(lhs, 1).T
    # only cells with block1 <= block2 are written; the strictly-lower blocks stay zero, but inside a
    # diagonal block both triangles are written
    scale = np.abs(S).max()
    bpos = p.bs.col_block_pos[p.num_eliminate_blocks:] - m.num_cols_e
    bsz = p.bs.col_block_size[p.num_eliminate_blocks:]
    for i in range(len(bsz)):
        for j in range(len(bsz)):
            blk = lhs[bpos[i]:bpos[i] + bsz[i], bpos[j]:bpos[j] + bsz[j]]
            ref = S[bpos[i]:bpos[i] + bsz[i], bpos[j]:bpos[j] + bsz[j]]
            if i <= j:
                np.testing.assert_allclose(blk, ref, rtol=0, atol=1e-13 * scale)
            else:
                assert not blk.any()
    np.testing.assert_allclose(rhs, r, rtol=0, atol=1e-13 * max(1, np.abs(r).max()) * 10)
    # diagonal-only lhs (SCHUR_JACOBI) = the diagonal blocks of S
    dl, _ = m.schur_eliminate(p.values, None, D, diagonal_only=True, want_rhs=False)
    off = 0
    for i in range(len(bsz)):
        n = int(bsz[i])
        np.testing.assert_allclose(dl[off:off + n * n].reshape(n, n), S[bpos[i]:bpos[i] + n, bpos[i]:bpos[i] + n],
                                   rtol=0, atol=1e-13 * scale)
        off += n * n
    # back substitution reproduces the dense solve
    sol = np.linalg.solve(H, g)
    z = np.linalg.solve(S, r)
    np.testing.assert_allclose(m.schur_back_substitute(p.values, p.b, D, z), sol, rtol=0, atol=1e-11 * max(1, np.abs(sol).max()))
    # implicit complement: every column of S, rhs, back substitution (kEpsilon = 1e-14 in the reference)
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, D, p.b)
    cols = np.stack([isc.sx(e) for e in np.eye(m.num_cols_f)], axis=1)
    np.testing.assert_allclose(cols, S, rtol=0, atol=1e-13 * scale)
    np.testing.assert_allclose(isc.rhs(), r, rtol=0, atol=1e-12 * max(1, np.abs(r).max()))
    np.testing.assert_allclose(isc.back_substitute(z), sol, rtol=0, atol=1e-11 * max(1, np.abs(sol).max()))
    # preconditioners: M^-1 * block = I
    inv, raw = m.schur_jacobi(p.values, D)
    off = 0
    for i in range(len(bsz)):
        n = int(bsz[i])
        Sb = S[bpos[i]:bpos[i] + n, bpos[i]:bpos[i] + n]
        np.testing.assert_allclose(inv[off:off + n * n].reshape(n, n) @ Sb, np.eye(n), atol=1e-10)
        np.testing.assert_allclose(np.triu(raw[off:off + n * n].reshape(n, n)), np.triu(Sb), atol=1e-13 * scale)
        off += n * n
    inv, raw = m.block_jacobi(p.values, D)
    off = 0
    for j in range(p.bs.num_col_blocks):
        n, c = int(p.bs.col_block_size[j]), int(p.bs.col_block_pos[j])
        np.testing.assert_allclose(inv[off:off + n * n].reshape(n, n) @ H[c:c + n, c:c + n], np.eye(n), atol=1e-10)
        off += n * n


def test_threads_agree(oracle, problems):
    # reference: parallel right-multiply bit-exact, parallel left-multiply 1e-12
    # (internal/ceres/block_sparse_matrix_test.cc:208-247)
    p = problems.synthetic_bal(None, num_cameras=12, num_points=400, num_observations=1700, seed=3)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(m.num_cols), rng.standard_normal(m.num_rows)
    oracle.set_num_threads(1)
    r1, l1 = m.right_multiply(p.values, x), m.left_multiply(p.values, y)
    s1, sum1 = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, max_it=30, r_tol=1e-12)
    oracle.set_num_threads(4)
    try:
        r4, l4 = m.right_multiply(p.values, x), m.left_multiply(p.values, y)
        s4, sum4 = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, max_it=30, r_tol=1e-12)
    finally:
        oracle.set_num_threads(1)
    np.testing.assert_array_equal(r1, r4)
    np.testing.assert_allclose(l1, l4, rtol=0, atol=1e-12 * np.abs(l1).max())
    np.testing.assert_allclose(s1, s4, rtol=0, atol=1e-9 * np.abs(s1).max())


@pytest.mark.parametrize("layout", ["schur", "cgnr"])
def test_bal_shaped_solvers_match_dense(oracle, problems, layout):
    p = problems.synthetic_bal(None, layout=layout, num_cameras=6, num_points=40, num_observations=150, seed=11)
    nelim = p.num_eliminate_blocks
    m = oracle.Matrix(p.bs, nelim)
    assert m.detect_structure() == ((2, 3, 9) if layout == "schur" else (0, 0, 0))
    A = p.bs.to_dense(p.values)
    ref = np.linalg.solve(A.T @ A + np.diag(p.D ** 2), A.T @ p.b)
    if layout == "schur":
        x, s = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=2, max_it=200, r_tol=1e-13)
        assert s.termination_type == 0
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    # CGNR works on the normal equations: error ~ cond(J'J + D^2) * r_tol, cond ~ 1e4..1e6 here
    x, s = m.cgnr_solve(p.values, p.b, p.D, preconditioner=1, max_it=500, r_tol=1e-13)
    assert s.termination_type == 0
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-6 * np.abs(ref).max())


@pytest.mark.parametrize("with_D", [True, False])
def test_power_series_expansion_against_dense(oracle, problems, with_D):
    """implicit_schur_complement_test.cc:185-217: the operator Z = (F'F)^-1 F'E (E'E)^-1 E'F column by column
    against dense algebra; S^-1 = sum_k Z^k (F'F)^-1, so the SPSE estimate converges to S^-1 x; and
    iterative_schur_complement_solver_test.cc's SPSE variants: preconditioner and initialisation vs a dense solve."""
    p = problems.synthetic_bal(None, num_cameras=5, num_points=60, num_observations=260, seed=61)
    if not with_D:
        p.D = None
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    A = p.bs.to_dense(p.values)
    ne = m.num_cols_e
    H = A.T @ A + (np.diag(p.D ** 2) if with_D else 0)
    P, Q, R = H[:ne, :ne], H[:ne, ne:], H[ne:, ne:]
    Rbd = np.zeros_like(R)
    for c in range(5):
        Rbd[9 * c:9 * c + 9, 9 * c:9 * c + 9] = R[9 * c:9 * c + 9, 9 * c:9 * c + 9]
    Z = np.linalg.solve(Rbd, Q.T @ np.linalg.solve(P, Q))
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    isc.compute_ftf_inverse()
    cols = np.stack([isc.power_series_operator(e) for e in np.eye(m.num_cols_f)], axis=1)
    np.testing.assert_allclose(cols, Z, rtol=0, atol=1e-13 * max(1.0, np.abs(Z).max()))
    x = np.random.default_rng(0).standard_normal(m.num_cols_f)
    want = np.linalg.solve(Rbd, x)
    acc = want.copy()
    term = want.copy()
    for _ in range(4):
        term = Z @ term
        acc += term
    np.testing.assert_allclose(isc.spse_apply(x, 4, 0.0), acc, rtol=0, atol=1e-12 * np.abs(acc).max())
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]) if with_D else A, np.concatenate([p.b, np.zeros(p.num_cols)]) if with_D else p.b, rcond=None)[0]
    for pre, init in ((3, False), (2, True), (3, True)):
        xs, s = oracle.iterative_schur_solve_spse(m, p.values, p.b, p.D, preconditioner=pre, max_it=300, r_tol=1e-12,
                                                  use_spse_initialization=init, max_num_spse_iterations=5, spse_tolerance=0.1)
        assert s.termination_type == 0, s
        assert np.linalg.norm(xs - ref) <= 1e-8 * np.linalg.norm(ref)


@pytest.mark.parametrize("shape", [dict(camera_width=9), dict(camera_width=10), dict(camera_width=6, point_width=4)])
def test_generators_of_leftover_rows_against_dense(oracle, problems, shape):
    """The problem generators the round-5 GPU tests lean on, pinned on the CPU: camera-only rows appended to a structured problem
    (problems.add_camera_rows) and the rows of a problem in another order (problems.permute_rows) are what they claim — the dense
    matrix of the result is the dense matrix of the parts, stacked / permuted — and the oracle's operators on them equal dense algebra.
    (Size-independent property: J'J, J'b and D do not depend on the order of the rows.)"""
    w = shape["camera_width"]
    base = problems.synthetic_structured(7, 60, 260, seed=3, skew=0.4, layout="cgnr", **shape)
    base = type(base)(base.bs, base.values, base.b, base.D, 0)
    q = problems.add_camera_rows(base, 9, seed=5, row_size=w, pair_fraction=0.4, camera_width=w)
    q = problems.add_camera_rows(q, 4, seed=6, row_size=2, camera_width=w)
    A0, A1 = base.bs.to_dense(base.values), q.bs.to_dense(q.values)
    n0 = base.bs.num_rows
    assert A1.shape == (n0 + 9 * w + 8, A0.shape[1]) and np.array_equal(A1[:n0], A0)
    extra = A1[n0:]
    pw = shape.get("point_width", 3)
    point_cols = np.concatenate([np.arange(p, p + s) for p, s in zip(base.bs.col_block_pos, base.bs.col_block_size) if s == pw and pw != w] or [np.zeros(0, int)]).astype(int)
    assert not extra[:, point_cols].any() and np.abs(extra).sum() > 0             # camera-only rows
    np.testing.assert_allclose(q.D ** 2 * 1e4, np.clip((A1 * A1).sum(0), 1e-6, 1e32), rtol=1e-12)   # D follows LM's formula on the whole matrix
    rng = np.random.default_rng(1)
    order = np.argsort(np.concatenate([np.arange(base.bs.num_row_blocks, dtype=float), rng.uniform(-1, base.bs.num_row_blocks, 13)]), kind="stable")
    p = problems.permute_rows(q, order)
    A2 = p.bs.to_dense(p.values)
    # same rows, another order: the row blocks of A2 are those of A1 in `order`
    pos1 = q.bs.row_block_pos.astype(int); sz1 = q.bs.row_block_size.astype(int)
    rows = np.concatenate([np.arange(pos1[r], pos1[r] + sz1[r]) for r in order])
    assert np.array_equal(A2, A1[rows]) and np.array_equal(p.b, q.b[rows])
    m = oracle.Matrix(p.bs, 0)
    x = rng.standard_normal(A2.shape[1])
    tol = dict(rtol=0, atol=1e-12 * np.abs(A2).max() ** 2 * A2.shape[0])
    np.testing.assert_allclose(m.left_multiply(p.values, m.right_multiply(p.values, x)), A1.T @ (A1 @ x), **tol)
    np.testing.assert_allclose(m.left_multiply(p.values, p.b), A1.T @ q.b, **tol)
    np.testing.assert_allclose(m.squared_column_norm(p.values), (A1 * A1).sum(0), **tol)
    xs, so = m.cgnr_solve(p.values, p.b, p.D, preconditioner=1, min_it=0, max_it=500, q_tol=-1.0, r_tol=1e-13)
    want = np.linalg.solve(A1.T @ A1 + np.diag(q.D ** 2), A1.T @ q.b)
    assert so.termination_type == 0 and np.linalg.norm(xs - want) <= 1e-9 * np.linalg.norm(want)
