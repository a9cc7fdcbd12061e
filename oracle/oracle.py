"""ctypes wrapper around oracle/libceres_oracle.so.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py (see oracle/ceres_oracle.h).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_char, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libceres_oracle.so")


class CStruct(ctypes.Structure):
    _fields_ = [
        ("num_row_blocks", c_int32), ("num_col_blocks", c_int32),
        ("row_block_size", POINTER(c_int32)), ("row_block_pos", POINTER(c_int32)),
        ("col_block_size", POINTER(c_int32)), ("col_block_pos", POINTER(c_int32)),
        ("row_cell_ptr", POINTER(c_int32)), ("cell_col_block", POINTER(c_int32)),
        ("cell_value_pos", POINTER(c_int32)),
    ]


class CSummary(ctypes.Structure):
    _fields_ = [("residual_norm", c_double), ("num_iterations", c_int32), ("termination_type", c_int32),
                ("message", c_char * 256)]


class LmOptions(ctypes.Structure):
    _fields_ = [("max_num_iterations", c_int), ("initial_radius", c_double), ("max_radius", c_double),
                ("min_radius", c_double), ("min_lm_diagonal", c_double), ("max_lm_diagonal", c_double),
                ("min_relative_decrease", c_double), ("eta", c_double), ("function_tolerance", c_double),
                ("gradient_tolerance", c_double), ("parameter_tolerance", c_double), ("jacobi_scaling", c_int),
                ("max_consecutive_invalid_steps", c_int)]


class LmIteration(ctypes.Structure):
    _fields_ = [("cost", c_double), ("cost_change", c_double), ("gradient_max_norm", c_double),
                ("step_norm", c_double), ("relative_decrease", c_double), ("radius", c_double),
                ("step_is_successful", c_int32), ("step_is_valid", c_int32),
                ("linear_solver_iterations", c_int32), ("linear_termination", c_int32)]


class LmSummary(ctypes.Structure):
    _fields_ = [("initial_cost", c_double), ("final_cost", c_double), ("num_successful_steps", c_int32),
                ("num_unsuccessful_steps", c_int32), ("num_linear_solves", c_int32), ("termination", c_int32),
                ("linear_solver_seconds", c_double), ("total_seconds", c_double),
                ("num_iterations_logged", c_int32), ("iterations", LmIteration * 256), ("message", c_char * 256)]


class SolverCtx(ctypes.Structure):
    _fields_ = [("m", c_void_p), ("solver_type", c_int), ("preconditioner", c_int), ("min_it", c_int),
                ("max_it", c_int), ("reset_period", c_int)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(None, c_void_p, POINTER(c_double), c_int64)
LINEAR_SOLVE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                   c_double, c_double, POINTER(c_double), POINTER(CSummary))

_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("ceres_oracle.cc", "bal_harness.cc", "ceres_oracle.h")):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp, ip = POINTER(c_double), POINTER(c_int32)
        L.oracle_matrix_create.restype = c_void_p
        L.oracle_matrix_create.argtypes = [POINTER(CStruct), c_int]
        L.oracle_matrix_destroy.argtypes = [c_void_p]
        for f in ("num_rows", "num_cols", "num_cols_e", "num_cols_f", "num_row_blocks_e"):
            getattr(L, "oracle_matrix_" + f).argtypes = [c_void_p]
            getattr(L, "oracle_matrix_" + f).restype = c_int
        L.oracle_matrix_num_nonzeros.argtypes = [c_void_p]
        L.oracle_matrix_num_nonzeros.restype = c_int64
        L.oracle_detect_structure.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        for f in ("right_multiply", "left_multiply", "right_multiply_e", "right_multiply_f", "left_multiply_e",
                  "left_multiply_f"):
            getattr(L, "oracle_" + f).argtypes = [c_void_p, dp, dp, dp]
        L.oracle_squared_column_norm.argtypes = [c_void_p, dp, dp]
        L.oracle_scale_columns.argtypes = [c_void_p, dp, dp]
        L.oracle_to_dense.argtypes = [c_void_p, dp, dp]
        L.oracle_block_diagonal_ete.argtypes = [c_void_p, dp, dp]
        L.oracle_block_diagonal_ftf.argtypes = [c_void_p, dp, dp]
        L.oracle_invert_psd.argtypes = [c_int, dp]
        L.oracle_isc_create.restype = c_void_p
        L.oracle_isc_create.argtypes = [c_void_p]
        L.oracle_isc_destroy.argtypes = [c_void_p]
        L.oracle_isc_init.argtypes = [c_void_p, dp, dp, dp]
        L.oracle_isc_sx.argtypes = [c_void_p, dp, dp]
        L.oracle_isc_rhs.argtypes = [c_void_p, dp]
        L.oracle_isc_ete_inverse.argtypes = [c_void_p, dp]
        L.oracle_isc_back_substitute.argtypes = [c_void_p, dp, dp]
        L.oracle_schur_eliminate.argtypes = [c_void_p, dp, dp, dp, c_int, dp, dp]
        L.oracle_schur_back_substitute.argtypes = [c_void_p, dp, dp, dp, dp, dp]
        L.oracle_block_jacobi.argtypes = [c_void_p, dp, dp, dp, dp]
        L.oracle_schur_jacobi.argtypes = [c_void_p, dp, dp, dp, dp]
        L.oracle_block_diagonal_apply.argtypes = [c_int, ip, dp, dp, dp]
        L.oracle_cg_dense.argtypes = [c_int, dp, dp, dp, c_int, c_int, c_int, c_double, c_double, dp, POINTER(CSummary)]
        solve_args = [c_void_p, dp, dp, dp, c_int, c_int, c_int, c_int, c_double, c_double, dp, POINTER(CSummary)]
        L.oracle_cgnr_solve.argtypes = solve_args
        L.oracle_iterative_schur_solve.argtypes = solve_args
        L.oracle_cgnr_solve_sharded.argtypes = solve_args + [ALLREDUCE_FN, c_void_p]
        L.oracle_iterative_schur_solve_sharded.argtypes = solve_args + [ALLREDUCE_FN, c_void_p]
        L.oracle_iterative_schur_solve_spse.argtypes = [c_void_p, dp, dp, dp, c_int, c_int, c_int, c_int, c_double, c_double,
                                                        c_int, c_int, c_double, dp, POINTER(CSummary)]
        L.oracle_isc_compute_ftf_inverse.argtypes = [c_void_p]
        L.oracle_isc_power_series_operator.argtypes = [c_void_p, dp, dp]
        L.oracle_isc_spse_apply.argtypes = [c_void_p, dp, dp, c_int, c_double]
        L.oracle_angle_axis_rotate_points.argtypes = [c_int64, dp, dp, dp]
        L.oracle_snavely_batch.argtypes = [c_int64, dp, dp, dp, dp, dp, dp]
        L.oracle_bal_generate.restype = c_void_p
        L.oracle_bal_generate.argtypes = [c_int, c_int, c_int64, c_double, c_double, c_double, c_uint64]
        L.oracle_bal_read.restype = c_void_p
        L.oracle_bal_read.argtypes = [c_char_p]
        L.oracle_bal_write.argtypes = [c_void_p, c_char_p]
        L.oracle_bal_destroy.argtypes = [c_void_p]
        L.oracle_bal_num_cameras.argtypes = [c_void_p]
        L.oracle_bal_num_points.argtypes = [c_void_p]
        L.oracle_bal_num_observations.argtypes = [c_void_p]
        L.oracle_bal_num_observations.restype = c_int64
        L.oracle_bal_parameters.argtypes = [c_void_p]
        L.oracle_bal_parameters.restype = dp
        L.oracle_bal_build_structure.argtypes = [c_void_p, c_int, POINTER(CStruct)]
        L.oracle_bal_get_state.argtypes = [c_void_p, dp]
        L.oracle_bal_set_state.argtypes = [c_void_p, dp]
        L.oracle_bal_evaluate.argtypes = [c_void_p, dp, dp, dp]
        L.oracle_bal_evaluate.restype = c_double
        L.oracle_lm_default_options.argtypes = [POINTER(LmOptions)]
        L.oracle_lm_solve.argtypes = [c_void_p, POINTER(CStruct), POINTER(LmOptions), LINEAR_SOLVE_FN, c_void_p,
                                      POINTER(LmSummary)]
        L.oracle_solver_callback.argtypes = [c_void_p, dp, dp, dp, c_double, c_double, dp, POINTER(CSummary)]
        L.oracle_solver_callback.restype = c_int
        _lib = L
    return _lib


def _dp(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(POINTER(c_double))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def angle_axis_rotate_points(angle_axis, pts):
    aa, pts = _f64(angle_axis).reshape(-1, 3), _f64(pts).reshape(-1, 3)
    out = np.empty_like(pts)
    lib().oracle_angle_axis_rotate_points(aa.shape[0], _dp(aa), _dp(pts), _dp(out))
    return out


def snavely_batch(cams, pts, obs):
    """(residuals (n,2), d/dcamera (n,2,9), d/dpoint (n,2,3)) of the Snavely reprojection error."""
    cams, pts, obs = _f64(cams).reshape(-1, 9), _f64(pts).reshape(-1, 3), _f64(obs).reshape(-1, 2)
    n = cams.shape[0]
    r, jc, jp = np.empty((n, 2)), np.empty((n, 2, 9)), np.empty((n, 2, 3))
    lib().oracle_snavely_batch(n, _dp(cams), _dp(pts), _dp(obs), _dp(r), _dp(jc), _dp(jp))
    return r, jc, jp


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def _cstruct(bs):
    p = lambda a: a.ctypes.data_as(POINTER(c_int32))
    c = CStruct(bs.num_row_blocks, bs.num_col_blocks, p(bs.row_block_size), p(bs.row_block_pos),
                p(bs.col_block_size), p(bs.col_block_pos), p(bs.row_cell_ptr), p(bs.cell_col_block),
                p(bs.cell_value_pos))
    c._keep = bs
    return c


class Summary:
    NAMES = {0: "SUCCESS", 1: "NO_CONVERGENCE", 2: "FAILURE", 3: "FATAL_ERROR"}

    def __init__(self, c: CSummary):
        self.num_iterations = int(c.num_iterations)
        self.termination_type = int(c.termination_type)
        self.residual_norm = float(c.residual_norm)
        self.message = c.message.decode(errors="replace")

    def __repr__(self):
        return f"Summary({self.NAMES.get(self.termination_type)}, it={self.num_iterations}, {self.message!r})"


class Matrix:
    """oracle_matrix: structure + transpose + chunks (BlockSparseMatrix / PartitionedMatrixView /
    SchurEliminator::Init)."""

    def __init__(self, bs, num_eliminate_blocks=0):
        self.bs = bs
        self.nelim = int(num_eliminate_blocks)
        self._c = _cstruct(bs)
        self.h = lib().oracle_matrix_create(byref(self._c), self.nelim)
        L = lib()
        self.num_rows = L.oracle_matrix_num_rows(self.h)
        self.num_cols = L.oracle_matrix_num_cols(self.h)
        self.num_cols_e = L.oracle_matrix_num_cols_e(self.h)
        self.num_cols_f = L.oracle_matrix_num_cols_f(self.h)
        self.num_row_blocks_e = L.oracle_matrix_num_row_blocks_e(self.h)
        self.num_nonzeros = L.oracle_matrix_num_nonzeros(self.h)
        sizes = bs.col_block_size.astype(np.int64)
        self.e_diag_len = int((sizes[: self.nelim] ** 2).sum())
        self.f_diag_len = int((sizes[self.nelim:] ** 2).sum())
        self.all_diag_len = self.e_diag_len + self.f_diag_len

    def __del__(self):
        try:
            lib().oracle_matrix_destroy(self.h)
        except Exception:
            pass

    def detect_structure(self):
        r, e, f = c_int(), c_int(), c_int()
        lib().oracle_detect_structure(self.h, byref(r), byref(e), byref(f))
        return r.value, e.value, f.value

    def _mul(self, fn, values, x, n_out, y=None):
        values, x = _f64(values), _f64(x)
        y = np.zeros(n_out) if y is None else _f64(y).copy()
        getattr(lib(), "oracle_" + fn)(self.h, _dp(values), _dp(x), _dp(y))
        return y

    def right_multiply(self, values, x, y=None): return self._mul("right_multiply", values, x, self.num_rows, y)
    def left_multiply(self, values, x, y=None): return self._mul("left_multiply", values, x, self.num_cols, y)
    def right_multiply_e(self, values, x, y=None): return self._mul("right_multiply_e", values, x, self.num_rows, y)
    def right_multiply_f(self, values, x, y=None): return self._mul("right_multiply_f", values, x, self.num_rows, y)
    def left_multiply_e(self, values, x, y=None): return self._mul("left_multiply_e", values, x, self.num_cols_e, y)
    def left_multiply_f(self, values, x, y=None): return self._mul("left_multiply_f", values, x, self.num_cols_f, y)

    def squared_column_norm(self, values):
        values = _f64(values)
        out = np.zeros(self.num_cols)
        lib().oracle_squared_column_norm(self.h, _dp(values), _dp(out))
        return out

    def scale_columns(self, values, scale):
        v = _f64(values).copy()
        lib().oracle_scale_columns(self.h, _dp(v), _dp(_f64(scale)))
        return v

    def to_dense(self, values):
        out = np.zeros((self.num_rows, self.num_cols))
        lib().oracle_to_dense(self.h, _dp(_f64(values)), _dp(out))
        return out

    def block_diagonal_ete(self, values):
        out = np.zeros(self.e_diag_len)
        lib().oracle_block_diagonal_ete(self.h, _dp(_f64(values)), _dp(out))
        return out

    def block_diagonal_ftf(self, values):
        out = np.zeros(self.f_diag_len)
        lib().oracle_block_diagonal_ftf(self.h, _dp(_f64(values)), _dp(out))
        return out

    def schur_eliminate(self, values, b, D, diagonal_only=False, want_rhs=True):
        n = self.num_cols_f
        lhs = np.zeros(self.f_diag_len if diagonal_only else n * n)
        rhs = np.zeros(n) if (want_rhs and b is not None) else None
        lib().oracle_schur_eliminate(self.h, _dp(_f64(values)), _dp(_f64(b)), _dp(_f64(D)), int(diagonal_only),
                                     _dp(lhs), _dp(rhs))
        return (lhs if diagonal_only else lhs.reshape(n, n)), rhs

    def schur_back_substitute(self, values, b, D, z):
        y = np.zeros(self.num_cols)
        lib().oracle_schur_back_substitute(self.h, _dp(_f64(values)), _dp(_f64(b)), _dp(_f64(D)), _dp(_f64(z)), _dp(y))
        return y

    def block_jacobi(self, values, D):
        inv, raw = np.zeros(self.all_diag_len), np.zeros(self.all_diag_len)
        lib().oracle_block_jacobi(self.h, _dp(_f64(values)), _dp(_f64(D)), _dp(inv), _dp(raw))
        return inv, raw

    def schur_jacobi(self, values, D):
        inv, raw = np.zeros(self.f_diag_len), np.zeros(self.f_diag_len)
        lib().oracle_schur_jacobi(self.h, _dp(_f64(values)), _dp(_f64(D)), _dp(inv), _dp(raw))
        return inv, raw

    def _solve(self, fn, values, b, D, preconditioner, min_it, max_it, reset_period, q_tol, r_tol, allreduce=None):
        x = np.full(self.num_cols, np.nan)
        s = CSummary()
        values, b, D = _f64(values), _f64(b), _f64(D)
        args = [self.h, _dp(values), _dp(b), _dp(D), int(preconditioner), int(min_it), int(max_it),
                int(reset_period), float(q_tol), float(r_tol), _dp(x), byref(s)]
        if allreduce is None:
            getattr(lib(), "oracle_" + fn)(*args)
        else:
            def _cb(_ctx, buf, n):
                view = np.ctypeslib.as_array(buf, shape=(n,))
                allreduce(view)
            cb = ALLREDUCE_FN(_cb)
            getattr(lib(), "oracle_" + fn + "_sharded")(*args, cb, None)
        return x, Summary(s)

    def cgnr_solve(self, values, b, D, preconditioner=1, min_it=0, max_it=500, reset_period=10, q_tol=0.0,
                   r_tol=0.0, allreduce=None):
        return self._solve("cgnr_solve", values, b, D, preconditioner, min_it, max_it, reset_period, q_tol, r_tol,
                           allreduce)

    def iterative_schur_solve(self, values, b, D, preconditioner=2, min_it=0, max_it=500, reset_period=10,
                              q_tol=0.0, r_tol=0.0, allreduce=None):
        return self._solve("iterative_schur_solve", values, b, D, preconditioner, min_it, max_it, reset_period,
                           q_tol, r_tol, allreduce)


def iterative_schur_solve_spse(matrix, values, b, D, preconditioner=3, min_it=0, max_it=500, reset_period=10, q_tol=0.0,
                               r_tol=0.0, use_spse_initialization=False, max_num_spse_iterations=5, spse_tolerance=0.1):
    x = np.full(matrix.num_cols, np.nan)
    s = CSummary()
    values, b, D = _f64(values), _f64(b), _f64(D)
    lib().oracle_iterative_schur_solve_spse(matrix.h, _dp(values), _dp(b), _dp(D), int(preconditioner), int(min_it), int(max_it),
                                            int(reset_period), float(q_tol), float(r_tol), int(use_spse_initialization),
                                            int(max_num_spse_iterations), float(spse_tolerance), _dp(x), byref(s))
    return x, Summary(s)


class ImplicitSchurComplement:
    def __init__(self, matrix: Matrix):
        self.m = matrix
        self.h = lib().oracle_isc_create(matrix.h)
        self._keep = None

    def __del__(self):
        try:
            lib().oracle_isc_destroy(self.h)
        except Exception:
            pass

    def init(self, values, D, b):
        self._keep = (_f64(values), _f64(D), _f64(b))
        lib().oracle_isc_init(self.h, _dp(self._keep[0]), _dp(self._keep[1]), _dp(self._keep[2]))

    def sx(self, x):
        y = np.full(self.m.num_cols_f, np.nan)
        lib().oracle_isc_sx(self.h, _dp(_f64(x)), _dp(y))
        return y

    def rhs(self):
        out = np.zeros(self.m.num_cols_f)
        lib().oracle_isc_rhs(self.h, _dp(out))
        return out

    def ete_inverse(self):
        out = np.zeros(self.m.e_diag_len)
        lib().oracle_isc_ete_inverse(self.h, _dp(out))
        return out

    def compute_ftf_inverse(self):
        lib().oracle_isc_compute_ftf_inverse(self.h)

    def power_series_operator(self, x, y=None):
        y = np.zeros(self.m.num_cols_f) if y is None else _f64(y).copy()
        lib().oracle_isc_power_series_operator(self.h, _dp(_f64(x)), _dp(y))
        return y

    def spse_apply(self, x, max_num_spse_iterations=5, spse_tolerance=0.0):
        y = np.full(self.m.num_cols_f, np.nan)
        lib().oracle_isc_spse_apply(self.h, _dp(_f64(x)), _dp(y), int(max_num_spse_iterations), float(spse_tolerance))
        return y

    def back_substitute(self, z):
        x = np.zeros(self.m.num_cols)
        lib().oracle_isc_back_substitute(self.h, _dp(_f64(z)) if z is not None else None, _dp(x))
        return x


def invert_psd(a):
    a = np.array(a, dtype=np.float64, order="C")
    rc = lib().oracle_invert_psd(a.shape[0], _dp(a))
    assert rc == 0
    return a


def block_diagonal_apply(block_sizes, blocks, x, y=None):
    bsz = np.ascontiguousarray(block_sizes, dtype=np.int32)
    y = np.zeros_like(x) if y is None else y.copy()
    lib().oracle_block_diagonal_apply(len(bsz), bsz.ctypes.data_as(POINTER(c_int32)), _dp(_f64(blocks)), _dp(_f64(x)), _dp(y))
    return y


def cg_dense(A, b, Minv=None, min_it=0, max_it=100, reset_period=10, q_tol=0.0, r_tol=1e-10, x0=None):
    A, b = _f64(A), _f64(b)
    n = b.shape[0]
    x = np.zeros(n) if x0 is None else _f64(x0).copy()
    s = CSummary()
    lib().oracle_cg_dense(n, _dp(A), _dp(b), _dp(_f64(Minv)), min_it, max_it, reset_period, q_tol, r_tol, _dp(x), byref(s))
    return x, Summary(s)


class BalProblem:
    """Synthetic or file-backed BAL problem + Snavely evaluator + LM loop (bal_harness.cc)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("could not create BAL problem")
        self.h = handle
        self.num_cameras = lib().oracle_bal_num_cameras(handle)
        self.num_points = lib().oracle_bal_num_points(handle)
        self.num_observations = lib().oracle_bal_num_observations(handle)
        self.bs = None
        self.nelim = 0

    @classmethod
    def generate(cls, num_cameras, num_points, num_observations, skew=0.0, pixel_noise=0.5, param_noise=0.02, seed=38401):
        return cls(lib().oracle_bal_generate(num_cameras, num_points, num_observations, skew, pixel_noise, param_noise, seed))

    @classmethod
    def read(cls, filename):
        return cls(lib().oracle_bal_read(filename.encode()))

    def write(self, filename):
        return lib().oracle_bal_write(self.h, filename.encode())

    def __del__(self):
        try:
            lib().oracle_bal_destroy(self.h)
        except Exception:
            pass

    def build_structure(self, schur_ordering=True):
        from importlib import import_module
        import sys
        c = CStruct()
        self.nelim = lib().oracle_bal_build_structure(self.h, int(bool(schur_ordering)), byref(c))
        n_rb, n_cb = c.num_row_blocks, c.num_col_blocks
        arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
        n_cells = int(np.ctypeslib.as_array(c.row_cell_ptr, shape=(n_rb + 1,))[-1])
        BlockStructure = sys.modules["ceres_solver_amd"].BlockStructure
        self.bs = BlockStructure(arr(c.row_block_size, n_rb), arr(c.row_block_pos, n_rb), arr(c.col_block_size, n_cb),
                                 arr(c.col_block_pos, n_cb), arr(c.row_cell_ptr, n_rb + 1), arr(c.cell_col_block, n_cells),
                                 arr(c.cell_value_pos, n_cells))
        return self.bs, self.nelim

    def indices(self):
        """(camera_index, point_index, observations[n, 2]) in observation (file) order."""
        n = int(self.num_observations)
        L = lib()
        L.oracle_bal_camera_index.restype = POINTER(c_int32)
        L.oracle_bal_point_index.restype = POINTER(c_int32)
        L.oracle_bal_observations.restype = POINTER(c_double)
        for f in (L.oracle_bal_camera_index, L.oracle_bal_point_index, L.oracle_bal_observations):
            f.argtypes = [c_void_p]
        cam = np.ctypeslib.as_array(L.oracle_bal_camera_index(self.h), shape=(n,)).copy()
        pt = np.ctypeslib.as_array(L.oracle_bal_point_index(self.h), shape=(n,)).copy()
        obs = np.ctypeslib.as_array(L.oracle_bal_observations(self.h), shape=(2 * n,)).copy().reshape(n, 2)
        return cam, pt, obs

    def state(self):
        x = np.zeros(self.bs.num_cols)
        lib().oracle_bal_get_state(self.h, _dp(x))
        return x

    def set_state(self, x):
        lib().oracle_bal_set_state(self.h, _dp(_f64(x)))

    def evaluate(self, state, jacobian=True):
        res = np.zeros(self.bs.num_rows)
        vals = np.zeros(self.bs.num_nonzeros) if jacobian else None
        cost = lib().oracle_bal_evaluate(self.h, _dp(_f64(state)), _dp(res), _dp(vals))
        return cost, res, vals

    def lm_solve(self, solve_fn=None, solver_type=5, preconditioner=2, min_it=0, max_it=500, reset_period=10, **opts):
        """solve_fn(values, b, D, q_tol, r_tol) -> (x, termination_type, num_iterations); if None the
        oracle's own solver of the given type is used."""
        o = LmOptions()
        lib().oracle_lm_default_options(byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        S = LmSummary()
        cs = _cstruct(self.bs)
        if solve_fn is None:
            m = Matrix(self.bs, self.nelim)
            ctx = SolverCtx(m.h, solver_type, preconditioner, min_it, max_it, reset_period)
            cb = ctypes.cast(lib().oracle_solver_callback, LINEAR_SOLVE_FN)
            lib().oracle_lm_solve(self.h, byref(cs), byref(o), cb, ctypes.cast(byref(ctx), c_void_p), byref(S))
        else:
            nnz, nr, nc = self.bs.num_nonzeros, self.bs.num_rows, self.bs.num_cols

            def _cb(_ctx, values, b, D, q_tol, r_tol, x, summ):
                v = np.ctypeslib.as_array(values, shape=(nnz,))
                bb = np.ctypeslib.as_array(b, shape=(nr,))
                dd = np.ctypeslib.as_array(D, shape=(nc,))
                xs, term, its = solve_fn(v, bb, dd, q_tol, r_tol)
                np.ctypeslib.as_array(x, shape=(nc,))[:] = xs
                summ.contents.termination_type = term
                summ.contents.num_iterations = its
                return 0
            cb = LINEAR_SOLVE_FN(_cb)
            lib().oracle_lm_solve(self.h, byref(cs), byref(o), cb, None, byref(S))
        return S
