"""Reference-held test vectors, re-entered as plain arrays (test infrastructure: only tests/ and tests/golden/make_golden.py import this).

* `linear_least_squares_problem(i)`: the data of the reference's LinearLeastSquaresProblem0..6
  (internal/ceres/linear_least_squares_problems.cc:78-965), with the hand-computed answers its comments carry.
* `block_sparse_test_matrix(i)`: the three hand-written matrices of internal/ceres/block_sparse_matrix_test.cc:50-146 with the
  dense form its comments spell out.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import __graft_entry__ as _entry  # noqa: E402

_pkg = _entry.load_package()
BlockStructure = _pkg.BlockStructure
LinearProblem = _pkg.problems.LinearProblem


def _scalar_problem(num_cols, rows, values, nelim, D, known):
    bs = BlockStructure.from_rows([1] * num_cols, [(1, cells) for cells in rows])
    n_rows = len(rows)
    return LinearProblem(bs, np.asarray(values, dtype=np.float64), np.arange(n_rows, dtype=np.float64),
                         np.asarray(D, dtype=np.float64), nelim, known)


def linear_least_squares_problem(problem_id: int) -> LinearProblem:
    if problem_id == 0:
        # dense 3x2, one row block / one column block; x and x_D from the reference comment :64-77
        bs = BlockStructure.from_rows([2], [(3, [(0, 0)])])
        return LinearProblem(bs, np.array([1., 2., 3., 4., 6., -10.]), np.array([8., 18., -18.]),
                             np.array([1., 2.]), 0,
                             {"x": np.array([2., 3.]), "x_D": np.array([1.78448275, 2.82327586])})
    if problem_id == 2:  # :301-439, answers :135-185 (computed with D = 0)
        rows = [[(0, 0), (2, 1)], [(0, 2), (3, 3)], [(1, 4), (4, 5)], [(1, 6), (2, 7)], [(1, 8), (2, 9)],
                [(2, 10), (3, 11), (4, 12)]]
        vals = [1, 2, 3, 4, 5, 6, 7, 8, 9, 1, 1, 1, 1]
        known = {
            "AtA": np.array([[10, 0, 2, 12, 0], [0, 155, 65, 0, 30], [2, 65, 70, 1, 1], [12, 0, 1, 17, 1],
                             [0, 30, 1, 1, 37]], dtype=np.float64),
            "Atb": np.array([3., 67., 33., 9., 17.]),
            "S": np.array([[42.3419, -1.4000, -11.5806], [-1.4000, 2.6000, 1.0000], [-11.5806, 1.0000, 31.1935]]),
            "r": np.array([4.3032, 5.4000, 4.0323]),
            "S_solve_r": np.array([0.2102, 2.1367, 0.1388]),
            "x": np.array([-2.3061, 0.3172, 0.2102, 2.1367, 0.1388]),
        }
        return _scalar_problem(5, rows, vals, 2, [1] * 5, known)
    if problem_id == 3:  # :441-525, no F blocks
        rows = [[(0, 0)], [(0, 1)], [(1, 2)], [(1, 3)], [(1, 4)]]
        return _scalar_problem(2, rows, [1, 3, 5, 7, 9], 2, [1] * 2, {})
    if problem_id == 4:  # :548-617
        bs = BlockStructure.from_rows([2, 3, 2], [(2, [(0, 0), (2, 4)]), (1, [(1, 8), (2, 11)])])
        vals = [1, 2, 1, 4, 1, 1, 5, 6, 9, 0, 0, 3, 1]
        return LinearProblem(bs, np.array(vals, dtype=np.float64), np.arange(3, dtype=np.float64),
                             np.arange(1, 8, dtype=np.float64) * 100, 1, {})
    if problem_id == 5:  # :678-800, answers :620-675 (D = 0)
        rows = [[(0, 0), (4, 1)], [(0, 2), (4, 3)], [(1, 4), (3, 5)], [(1, 6), (3, 7)], [(1, 8), (2, 9)],
                [(1, 10), (2, 11)]]
        vals = [-1, 2, 3, 4, -1, 1, -3, 1, -1, 3, -2, 1]
        known = {
            "S": np.array([[8.3333, -1.3333, 0], [-1.3333, 0.9333, 0], [0, 0, 10.0000]]),
            "r": np.array([8.6667, -1.6667, 1.0000]),
            "S_solve_r": np.array([0.9778, -0.3889, 0.1000]),
            "x": np.array([0.2, -1.4444, 0.9777, -0.3888, 0.1]),
        }
        return _scalar_problem(5, rows, vals, 2, [1] * 5, known)
    if problem_id == 6:  # :829-925
        bs = BlockStructure.from_rows([2, 3, 2], [(2, [(0, 0), (2, 4)]), (2, [(0, 8), (2, 12)]),
                                                  (1, [(1, 16), (2, 19)])])
        vals = [1, 2, 1, 4, 1, 1, 5, 6, 3, 4, 5, 6, 7, 8, 9, 0, 9, 0, 0, 3, 1]
        return LinearProblem(bs, np.array(vals, dtype=np.float64), np.arange(5, dtype=np.float64),
                             np.arange(1, 8, dtype=np.float64) * 100, 1, {})
    raise ValueError(f"no block-sparse problem with id {problem_id}")


def block_sparse_test_matrix(matrix_id: int):
    """The three hand-written matrices of the reference's BlockSparseMatrix tests (CreateTestMatrixFromId,
    internal/ceres/block_sparse_matrix_test.cc:50-146) WITH the dense form its comments spell out: a 2-cell block-diagonal
    layout, a row with two cells, and the E|F-split value ordering (cells of the left submatrix before the right one).
    Returns (BlockStructure, values, dense)."""
    if matrix_id == 0:
        bs = BlockStructure.from_rows([2, 3, 1], [(2, [(0, 0)]), (2, [(1, 4)])])
        dense = [[1, 2, 0, 0, 0, 0], [3, 4, 0, 0, 0, 0], [0, 0, 5, 6, 7, 0], [0, 0, 8, 9, 10, 0]]
        n = 10
    elif matrix_id == 1:
        bs = BlockStructure.from_rows([2, 1, 2, 1], [(2, [(0, 0), (2, 4)]), (1, [(1, 8)])])
        dense = [[1, 2, 0, 5, 6, 0], [3, 4, 0, 7, 8, 0], [0, 0, 9, 0, 0, 0]]
        n = 9
    elif matrix_id == 2:
        bs = BlockStructure.from_rows([2, 1, 2, 1], [(2, [(0, 0), (2, 5)]), (1, [(1, 4), (3, 9)])])
        dense = [[1, 2, 0, 6, 7, 0], [3, 4, 0, 8, 9, 0], [0, 0, 5, 0, 0, 10]]
        n = 10
    else:
        raise ValueError(matrix_id)
    return bs, np.arange(1, n + 1, dtype=np.float64), np.array(dense, dtype=np.float64)
