"""SURVEY.md §8 rows a7 and a24 on the GPU, through the C ABI.

a7 — the four PartitionedMatrixView products and the two block diagonals as DIRECT entry points,
     tested the way the reference tests them (internal/ceres/partitioned_matrix_view_test.cc:103-230):
     E x / F x against the full product with the other half of x zeroed, E^T x | F^T x against the
     full transpose product, blockdiag(E^T E) / blockdiag(F^T F) against the dense matrix — on
     LinearLeastSquaresProblem 2, 4, 6 (the reference's parameters, :247-251), on 5 and on a
     BAL-shaped matrix; plus the oracle's own PMV restatement.
a24 — DetectStructure as the PRODUCT reports it (ceres_hip_info.{row,e,f}_block_size) against the
     values the reference's detect_structure_test.cc expects and against the oracle's restatement
     (internal/ceres/detect_structure.cc:39-121).
"""
import numpy as np
import pytest

from test_gpu_operators import make_solver, rel

pytestmark = pytest.mark.gpu

K_EPS = 1e-14  # the reference's kEpsilon (partitioned_matrix_view_test.cc:54); relative here: values are O(1..10)


def _problems(problems):
    out = [(f"lsq{i}", problems.linear_least_squares_problem(i)) for i in (2, 4, 5, 6)]
    out.append(("random_schur", problems.random_schur_problem(seed=5)))
    out.append(("bal", problems.synthetic_bal(None, layout="schur", num_cameras=7, num_points=300, num_observations=1500, seed=3)))
    return out


def test_partitioned_matrix_view_products(hip, oracle, problems):
    rng = np.random.default_rng(11)
    for name, p in _problems(problems):
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI)
        s.load(p.values, p.b, p.D)
        i = s.info()
        assert (i.num_e_blocks, i.num_f_blocks) == (p.num_eliminate_blocks, p.bs.num_col_blocks - p.num_eliminate_blocks)
        assert (i.num_cols, i.num_rows) == (p.bs.num_cols, p.bs.num_rows)
        nce, ncf = i.num_cols_e, i.num_cols_f
        # RightMultiplyAndAccumulateE / F (:103-143): against the full product with the other half zero
        x1 = rng.random(nce)
        x2 = np.concatenate([x1, np.zeros(ncf)])
        expected = s.right_multiply(x2)
        scale = max(np.abs(expected).max(), 1.0)
        assert np.abs(s.right_multiply_e(x1) - expected).max() <= K_EPS * scale * 10, name
        assert rel(s.right_multiply_e(x1), m.right_multiply_e(p.values, x1)) <= 1e-13, name
        x1 = rng.random(ncf)
        x2 = np.concatenate([np.zeros(nce), x1])
        expected = s.right_multiply(x2)
        scale = max(np.abs(expected).max(), 1.0)
        assert np.abs(s.right_multiply_f(x1) - expected).max() <= K_EPS * scale * 10, name
        assert rel(s.right_multiply_f(x1), m.right_multiply_f(p.values, x1)) <= 1e-13, name
        # ... and they ACCUMULATE
        y0 = rng.standard_normal(i.num_rows)
        assert rel(s.right_multiply_f(x1, y0), m.right_multiply_f(p.values, x1, y0)) <= 1e-13, name
        # LeftMultiplyAndAccumulate (:145-165)
        x = rng.random(i.num_rows)
        expected = s.left_multiply(x)
        e_actual, f_actual = s.left_multiply_e(x), s.left_multiply_f(x)
        scale = max(np.abs(expected).max(), 1.0)
        assert np.abs(np.concatenate([e_actual, f_actual]) - expected).max() <= K_EPS * scale * 10, name
        assert rel(e_actual, m.left_multiply_e(p.values, x)) <= 1e-13 and rel(f_actual, m.left_multiply_f(p.values, x)) <= 1e-13, name
        c0 = rng.standard_normal(ncf)
        assert rel(s.left_multiply_f(x, c0), m.left_multiply_f(p.values, x, c0)) <= 1e-13, name
        # BlockDiagonalEtE / FtF (:167-230): against the dense matrix
        A = m.to_dense(p.values)
        E, F = A[:, :nce], A[:, nce:]
        ete, ftf = s.block_diagonal_ete(), s.block_diagonal_ftf()
        sizes = p.bs.col_block_size
        off, pos = 0, 0
        EtE = E.T @ E
        for n in sizes[:p.num_eliminate_blocks]:
            assert np.abs(ete[off:off + n * n].reshape(n, n) - EtE[pos:pos + n, pos:pos + n]).max() <= 1e-13 * max(np.abs(EtE).max(), 1), name
            off += n * n
            pos += n
        # E^T E is block diagonal (one E cell per row): nothing outside the blocks
        mask = np.zeros_like(EtE, dtype=bool)
        pos = 0
        for n in sizes[:p.num_eliminate_blocks]:
            mask[pos:pos + n, pos:pos + n] = True
            pos += n
        assert np.abs(EtE[~mask]).max(initial=0.0) == 0.0, name
        off, pos = 0, 0
        FtF = F.T @ F
        for n in sizes[p.num_eliminate_blocks:]:
            assert np.abs(ftf[off:off + n * n].reshape(n, n) - FtF[pos:pos + n, pos:pos + n]).max() <= 1e-13 * max(np.abs(FtF).max(), 1), name
            off += n * n
            pos += n
        assert rel(ete, m.block_diagonal_ete(p.values)) <= 1e-13 and rel(ftf, m.block_diagonal_ftf(p.values)) <= 1e-13, name
        s.close()


def test_partitioned_products_on_the_fused_path_instance(hip, oracle, problems):
    """The entry points also work on an instance that selected the fused <2,3,9> kernels (they read
    the caller's layout through cell.position, like everything generic)."""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=40, num_points=5000, num_observations=22000, seed=9, skew=0.5)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    assert s.info().kernel_path == hip.PATH_BAL
    s.load(p.values, p.b, p.D)
    rng = np.random.default_rng(2)
    xe, xf, xr = rng.standard_normal(m.num_cols_e), rng.standard_normal(m.num_cols_f), rng.standard_normal(m.num_rows)
    assert rel(s.right_multiply_e(xe), m.right_multiply_e(p.values, xe)) <= 1e-13
    assert rel(s.right_multiply_f(xf), m.right_multiply_f(p.values, xf)) <= 1e-13
    assert rel(s.left_multiply_e(xr), m.left_multiply_e(p.values, xr)) <= 1e-12
    assert rel(s.left_multiply_f(xr), m.left_multiply_f(p.values, xr)) <= 1e-12
    s.close()


# (row, e, f) the reference's DetectStructure returns: detect_structure_test.cc builds these cases;
# problems 2/5 are scalar (1,1,1), problem 4 votes row 2 / e 2 / f {2} from its single E row -> (2,2,2)
# [the second row has no E cell and does not vote], problem 6 likewise (2,2,2), BAL (2,3,9).
EXPECTED_STRUCTURE = {2: (1, 1, 1), 4: (2, 2, 2), 5: (1, 1, 1), 6: (2, 2, 2)}


def test_detect_structure_of_the_product(hip, oracle, problems):
    for pid, want in EXPECTED_STRUCTURE.items():
        p = problems.linear_least_squares_problem(pid)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.JACOBI)
        i = s.info()
        got = (i.row_block_size, i.e_block_size, i.f_block_size)
        assert got == want, (pid, got)
        assert got == oracle.Matrix(p.bs, p.num_eliminate_blocks).detect_structure(), pid
        s.close()
    p = problems.synthetic_bal(None, layout="schur", num_cameras=5, num_points=60, num_observations=200, seed=1)
    s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    i = s.info()
    assert (i.row_block_size, i.e_block_size, i.f_block_size) == (2, 3, 9) and i.kernel_path == hip.PATH_BAL
    s.close()
    # dynamic sizes (-1 = Eigen::Dynamic): rows / E blocks / F blocks of different sizes among the E rows
    # (detect_structure_test.cc: DynamicRow, DynamicFBlockDifferentRows, DynamicEBlock)
    from ceres_solver_amd import BlockStructure
    cases = {
        "dynamic_row": (BlockStructure.from_rows([3, 3], [(2, [(0, 0), (1, 6)]), (1, [(0, 12), (1, 15)])]), 1, (-1, 3, 3)),
        "dynamic_f_different_rows": (BlockStructure.from_rows([3, 3, 4], [(2, [(0, 0), (1, 6)]), (2, [(0, 12), (2, 18)])]), 1, (2, 3, -1)),
        "dynamic_e": (BlockStructure.from_rows([3, 4, 3], [(2, [(0, 0), (2, 6)]), (2, [(1, 12), (2, 20)])]), 2, (2, -1, 3)),
    }
    for name, (bs, nelim, want) in cases.items():
        o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.JACOBI, max_num_iterations=5,
                                    elimination_groups=[nelim])
        s = hip.HipLinearSolver(o)
        s.set_structure(bs)
        i = s.info()
        got = (i.row_block_size, i.e_block_size, i.f_block_size)
        assert got == want, (name, got)
        assert got == oracle.Matrix(bs, nelim).detect_structure(), name
        s.close()


@pytest.mark.parametrize("matrix_id", [0, 1, 2])
def test_block_sparse_matrix_layouts_ka6_on_the_device(hip, problems, matrix_id):
    """The reference's hand-written BlockSparseMatrix matrices (block_sparse_matrix_test.cc:50-146; dense forms from its
    comments) through the PRODUCT's operators: y += A x, y += A^T x, |A_j|^2 and (J^T J + D^2) x."""
    bs, values, dense = problems.block_sparse_test_matrix(matrix_id)
    o = hip.LinearSolverOptions(type=hip.CGNR, preconditioner_type=hip.JACOBI, max_num_iterations=20)
    s = hip.HipLinearSolver(o)
    s.set_structure(bs)
    rng = np.random.default_rng(matrix_id)
    D = 0.5 + rng.random(bs.num_cols)
    s.load(values, rng.standard_normal(bs.num_rows), D)
    x, y0 = rng.standard_normal(bs.num_cols), rng.standard_normal(bs.num_rows)
    np.testing.assert_allclose(s.right_multiply(x, y0), y0 + dense @ x, rtol=0, atol=1e-13)
    z, c0 = rng.standard_normal(bs.num_rows), rng.standard_normal(bs.num_cols)
    np.testing.assert_allclose(s.left_multiply(z, c0), c0 + dense.T @ z, rtol=0, atol=1e-13)
    np.testing.assert_allclose(s.squared_column_norm(), (dense * dense).sum(0), rtol=0, atol=1e-13)
    np.testing.assert_allclose(s.jtjx(x), dense.T @ (dense @ x) + D * D * x, rtol=0, atol=1e-12)
    s.close()
