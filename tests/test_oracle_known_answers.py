"""Pin the oracle against the reference's own hand-computed known answers
(SURVEY.md §8c KA-1..KA-6).  CPU only."""
import numpy as np
import pytest


def test_problem0_dense_solution(oracle, problems):
    # KA-3: internal/ceres/linear_least_squares_problems.cc:64-132 — x = [2,3], x_D with D = [1,2]
    p = problems.linear_least_squares_problem(0)
    m = oracle.Matrix(p.bs, 0)
    x, s = m.cgnr_solve(p.values, p.b, None, preconditioner=0, max_it=10, r_tol=1e-14)
    assert s.termination_type == 0
    np.testing.assert_allclose(x, p.known["x"], rtol=0, atol=1e-10)
    xd, s = m.cgnr_solve(p.values, p.b, p.D, preconditioner=1, max_it=10, r_tol=1e-14)
    np.testing.assert_allclose(xd, p.known["x_D"], rtol=0, atol=1e-8)  # the reference truncates to 8 decimals


def test_problem2_normal_equations(oracle, problems):
    # KA-1: A'A and A'b from the comment at :135-185
    p = problems.linear_least_squares_problem(2)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    A = m.to_dense(p.values)
    np.testing.assert_array_equal(A.T @ A, p.known["AtA"])
    np.testing.assert_array_equal(m.left_multiply(p.values, p.b), p.known["Atb"])
    np.testing.assert_array_equal(A, p.bs.to_dense(p.values))


@pytest.mark.parametrize("pid", [2, 5])
def test_schur_complement_known_answers(oracle, problems, pid):
    # KA-1 / KA-2: S, r, S\r, A\b (4 digits in the reference's comments; D = 0 there)
    p = problems.linear_least_squares_problem(pid)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    lhs, rhs = m.schur_eliminate(p.values, p.b, None)
    S = np.triu(lhs) + np.triu(lhs, 1).T  # the eliminator fills block1 <= block2 only
    np.testing.assert_allclose(S, p.known["S"], atol=6e-5)
    np.testing.assert_allclose(rhs, p.known["r"], atol=6e-5)
    z = np.linalg.solve(S, rhs)
    np.testing.assert_allclose(z, p.known["S_solve_r"], atol=6e-5)
    x = m.schur_back_substitute(p.values, p.b, None, z)
    np.testing.assert_allclose(x, p.known["x"], atol=1.1e-4)
    # the implicit operator reproduces the same S column by column, and the same rhs
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, None, p.b)
    cols = np.stack([isc.sx(e) for e in np.eye(m.num_cols_f)], axis=1)
    np.testing.assert_allclose(cols, S, atol=1e-13)
    np.testing.assert_allclose(isc.rhs(), rhs, atol=1e-13)
    # both solvers land on A\b
    for solve in (m.iterative_schur_solve, m.cgnr_solve):
        xs, s = solve(p.values, p.b, None, preconditioner=1, max_it=50, r_tol=1e-14)
        assert s.termination_type == 0, s
        np.testing.assert_allclose(xs, p.known["x"], atol=1.1e-4)


def test_conjugate_gradients_known_answers(oracle):
    # KA-5: internal/ceres/conjugate_gradients_solver_test.cc:57-140
    x, s = oracle.cg_dense(np.eye(3), np.array([1., 2., 3.]), max_it=10, r_tol=1e-9)
    assert s.termination_type == 0 and s.num_iterations == 1
    np.testing.assert_allclose(x, [1, 2, 3], atol=1e-14)
    A = np.array([[2., -1, 0], [-1, 2, -1], [0, -1, 2]])
    b = np.array([-1., 0, 3])
    x, s = oracle.cg_dense(A, b, max_it=10, r_tol=1e-9)
    assert s.termination_type == 0
    np.testing.assert_allclose(x, [0, 1, 2], atol=1e-9)
    # |b| = 0 short-circuit and max-iterations message
    x, s = oracle.cg_dense(A, np.zeros(3), max_it=10)
    assert s.termination_type == 0 and "|b| = 0" in s.message and not x.any()
    x, s = oracle.cg_dense(A, b, max_it=1, r_tol=1e-30)
    assert s.termination_type == 1 and s.num_iterations == 1 and "Maximum number of iterations" in s.message


def test_structure_edge_cases(oracle, problems):
    # KA-4: problem 4 / 6 — the non-E row has an F block size that no E row has;
    # problem 3 has no F blocks at all (num_schur_complement_blocks == 0 branch).
    p4 = problems.linear_least_squares_problem(4)
    m4 = oracle.Matrix(p4.bs, 1)
    assert m4.detect_structure() == (2, 2, 2)
    assert m4.num_row_blocks_e == 1 and (m4.num_cols_e, m4.num_cols_f) == (2, 5)
    p6 = problems.linear_least_squares_problem(6)
    m6 = oracle.Matrix(p6.bs, 1)
    assert m6.detect_structure() == (2, 2, 2) and m6.num_row_blocks_e == 2
    p2 = problems.linear_least_squares_problem(2)
    assert oracle.Matrix(p2.bs, 2).detect_structure() == (1, 1, 1)
    p3 = problems.linear_least_squares_problem(3)
    m3 = oracle.Matrix(p3.bs, 2)
    x, s = m3.iterative_schur_solve(p3.values, p3.b, p3.D, preconditioner=2, max_it=5)
    assert s.termination_type == 0 and s.num_iterations == 0
    A = m3.to_dense(p3.values)
    expect = np.linalg.solve(A.T @ A + np.diag(p3.D ** 2), A.T @ p3.b)
    np.testing.assert_allclose(x, expect, rtol=1e-14)


@pytest.mark.parametrize("pid", [2, 4, 5, 6])
def test_solvers_match_dense_qr(oracle, problems, pid):
    # internal/ceres/iterative_schur_complement_solver_test.cc:75-117: vs a dense solve of the
    # regularised problem, r_tolerance = 1e-12, max iterations = num_cols
    p = problems.linear_least_squares_problem(pid)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    A = m.to_dense(p.values)
    ref = np.linalg.lstsq(np.vstack([A, np.diag(p.D)]), np.concatenate([p.b, np.zeros(p.num_cols)]), rcond=None)[0]
    for pre in (0, 1, 2):
        x, s = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=pre, max_it=p.num_cols, r_tol=1e-12)
        assert np.linalg.norm(x - ref) < 1e-12 * max(1.0, np.linalg.norm(ref)), (pre, s)
    for pre in (0, 1):
        x, s = m.cgnr_solve(p.values, p.b, p.D, preconditioner=pre, max_it=4 * p.num_cols, r_tol=1e-14)
        assert np.linalg.norm(x - ref) < 1e-10 * max(1.0, np.linalg.norm(ref)), (pre, s)


@pytest.mark.parametrize("matrix_id", [0, 1, 2])
def test_block_sparse_matrix_layouts_ka6(oracle, problems, matrix_id):
    """KA-6: the reference's hand-written BlockSparseMatrix test matrices (block_sparse_matrix_test.cc:50-146) — any
    cell.position order, incl. the E|F-split ordering of id 2 — reproduce the dense matrices its comments give, through
    ToDenseMatrix, Right/LeftMultiplyAndAccumulate (tested there against the dense product, :225-263) and SquaredColumnNorm."""
    bs, values, dense = problems.block_sparse_test_matrix(matrix_id)
    assert (bs.num_rows, bs.num_cols, bs.num_nonzeros) == (dense.shape[0], dense.shape[1], values.shape[0])
    m = oracle.Matrix(bs, 0)
    np.testing.assert_array_equal(m.to_dense(values), dense)
    np.testing.assert_array_equal(bs.to_dense(values), dense)
    rng = np.random.default_rng(matrix_id)
    x, y0 = rng.standard_normal(bs.num_cols), rng.standard_normal(bs.num_rows)
    np.testing.assert_allclose(m.right_multiply(values, x, y0), y0 + dense @ x, rtol=0, atol=1e-14)
    z, c0 = rng.standard_normal(bs.num_rows), rng.standard_normal(bs.num_cols)
    np.testing.assert_allclose(m.left_multiply(values, z, c0), c0 + dense.T @ z, rtol=0, atol=1e-14)
    np.testing.assert_allclose(m.squared_column_norm(values), (dense * dense).sum(0), rtol=0, atol=1e-14)
