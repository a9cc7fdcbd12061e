"""Python host side above the C ABI (include/ceres_hip.h).

Mirrors the reference's plugin interface for this path — same names, argument meaning
and error behaviour — so that the parity tests read like the reference's own tests:

    LinearSolverOptions   ~ ceres::internal::LinearSolver::Options         internal/ceres/linear_solver.h:148-230
    PerSolveOptions       ~ LinearSolver::PerSolveOptions                   internal/ceres/linear_solver.h:237-318
    Summary               ~ LinearSolver::Summary                           internal/ceres/linear_solver.h:320-326
    HipLinearSolver.solve ~ LinearSolver::Solve(A, b, per_solve_options, x) internal/ceres/linear_solver.h:339-342
    create_linear_solver  ~ LinearSolver::Create incl. its fall-backs        internal/ceres/linear_solver.cc:51-126

There is no CPU fall-back here: without libceres_hip.so or without a gfx950 device
every constructor raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char, c_char_p, c_double, c_int32, c_int64, c_uint8, c_uint32, c_void_p
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .block_structure import BlockStructure, CBlockStructure

# enum values equal the reference's (include/ceres/types.h:57-141, internal/ceres/linear_solver.h:57-74)
DENSE_SCHUR, ITERATIVE_SCHUR, CGNR = 3, 5, 6
IDENTITY, JACOBI, SCHUR_JACOBI, SCHUR_POWER_SERIES_EXPANSION = 0, 1, 2, 3
SUCCESS, NO_CONVERGENCE, FAILURE, FATAL_ERROR = 0, 1, 2, 3
PATH_GENERIC, PATH_BAL = 0, 1
E_INVALID, E_UNSUPPORTED, E_HIP, E_COMM, E_NODEVICE = -1, -2, -3, -4, -5   # CERES_HIP_E_* (include/ceres_hip.h)
TERMINATION_NAMES = {0: "SUCCESS", 1: "NO_CONVERGENCE", 2: "FAILURE", 3: "FATAL_ERROR"}
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64

(TIMED_JTJX, TIMED_SX, TIMED_SCHUR_INIT, TIMED_SCHUR_JACOBI, TIMED_BACK_SUBSTITUTE, TIMED_PACK, TIMED_BLOCK_JACOBI, TIMED_COPY,
 TIMED_READ_STREAM, TIMED_CGNR_SETUP) = range(1, 11)

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    # CERES_HIP_LIBRARY: an A/B build of the same sources (tools/build_variant.sh); never a different implementation
    return os.environ.get("CERES_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libceres_hip.so")


class COptions(ctypes.Structure):
    _fields_ = [("solver_type", c_int32), ("preconditioner_type", c_int32), ("min_num_iterations", c_int32),
                ("max_num_iterations", c_int32), ("residual_reset_period", c_int32), ("num_eliminate_blocks", c_int32),
                ("device", c_int32), ("force_generic_path", c_int32), ("cg_check_interval", c_int32),
                ("jacobian_storage", c_int32), ("max_num_spse_iterations", c_int32), ("use_spse_initialization", c_int32),
                ("spse_tolerance", c_double), ("use_explicit_schur_complement", c_int32), ("reserved", c_int32)]


class CSummary(ctypes.Structure):
    _fields_ = [("residual_norm", c_double), ("num_iterations", c_int32), ("termination_type", c_int32),
                ("message", c_char * 256)]


class CInfo(ctypes.Structure):
    _fields_ = [("kernel_path", c_int32), ("num_rows", c_int32), ("num_cols", c_int32), ("num_cols_e", c_int32),
                ("num_cols_f", c_int32), ("num_row_blocks_e", c_int32), ("num_e_blocks", c_int32),
                ("num_f_blocks", c_int32), ("row_block_size", c_int32), ("e_block_size", c_int32),
                ("f_block_size", c_int32), ("num_nonzeros", c_int64), ("num_observations", c_int64),
                ("num_tiles", c_int64), ("device_bytes", c_int64), ("camera_accum_in_lds", c_int32),
                ("world_size", c_int32), ("rank", c_int32), ("p2p_enabled", c_int32), ("p2p_fine_grained", c_int32),
                ("camera_accum_hybrid", c_int32), ("hybrid_popular_rows", c_int32), ("num_observations_in_lds", c_int64),
                ("points_renumbered", c_int32), ("cg_iteration_in_operator", c_int32), ("collectives_last_step", c_int32)]


class CTiming(ctypes.Structure):
    _fields_ = [("upload_ms", c_double), ("pack_ms", c_double), ("setup_ms", c_double), ("preconditioner_ms", c_double),
                ("cg_ms", c_double), ("back_substitute_ms", c_double), ("download_ms", c_double), ("total_ms", c_double),
                ("operator_applications", c_int32), ("reserved", c_int32)]


class CLmOptions(ctypes.Structure):
    _fields_ = [("radius", c_double), ("min_diagonal", c_double), ("max_diagonal", c_double), ("eta", c_double),
                ("reuse_diagonal", c_int32), ("values_unchanged", c_int32)]


class CLmResult(ctypes.Structure):
    _fields_ = [("linear_solver", CSummary), ("model_cost_change", c_double), ("step_is_finite", c_int32),
                ("reserved", c_int32)]


# every symbol include/ceres_hip.h declares: (name, restype, argtypes)
_DP = POINTER(c_double)
ABI_VERSION = 2   # CERES_HIP_ABI_VERSION of include/ceres_hip.h this binding mirrors (struct layouts below)

ABI = [
    ("ceres_hip_abi_version", c_int32, []),
    ("ceres_hip_device_count", c_int32, []),
    ("ceres_hip_create", c_void_p, [POINTER(COptions)]),
    ("ceres_hip_destroy", None, [c_void_p]),
    ("ceres_hip_last_error", c_char_p, [c_void_p]),
    ("ceres_hip_set_structure", c_int32, [c_void_p, POINTER(CBlockStructure)]),
    ("ceres_hip_get_info", c_int32, [c_void_p, POINTER(CInfo)]),
    ("ceres_hip_comm_get_unique_id", c_int32, [POINTER(c_uint8)]),
    ("ceres_hip_comm_init", c_int32, [c_void_p, POINTER(c_uint8), c_int32, c_int32]),
    ("ceres_hip_comm_p2p_prepare", c_int32, [c_void_p, c_int32, c_int32, c_int64, POINTER(c_uint8)]),
    ("ceres_hip_comm_p2p_connect", c_int32, [c_void_p, POINTER(c_uint8)]),
    ("ceres_hip_comm_p2p_selftest", c_int32, [c_void_p]),
    ("ceres_hip_comm_p2p_disable", c_int32, [c_void_p]),
    ("ceres_hip_debug_allreduce_timing", c_int32, [c_void_p, c_int64, c_int32, _DP]),
    ("ceres_hip_solve", c_int32, [c_void_p, _DP, _DP, _DP, c_double, c_double, _DP, POINTER(CSummary)]),
    ("ceres_hip_solve_device", c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, POINTER(CSummary)]),
    ("ceres_hip_solve_unchanged_values", c_int32, [c_void_p, _DP, c_double, c_double, _DP, POINTER(CSummary)]),
    ("ceres_hip_load", c_int32, [c_void_p, _DP, _DP, _DP]),
    ("ceres_hip_load_device", c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("ceres_hip_op_right_multiply", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_left_multiply", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_right_multiply_e", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_right_multiply_f", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_left_multiply_e", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_left_multiply_f", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_block_diagonal_ete", c_int32, [c_void_p, _DP, c_int64]),
    ("ceres_hip_op_block_diagonal_ftf", c_int32, [c_void_p, _DP, c_int64]),
    ("ceres_hip_op_squared_column_norm", c_int32, [c_void_p, _DP]),
    ("ceres_hip_op_jtjx", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_jtb", c_int32, [c_void_p, _DP]),
    ("ceres_hip_op_schur_init", c_int32, [c_void_p]),
    ("ceres_hip_get_schur_rhs", c_int32, [c_void_p, _DP]),
    ("ceres_hip_get_ete_inverse", c_int32, [c_void_p, _DP, c_int64]),
    ("ceres_hip_op_schur_sx", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_back_substitute", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_power_series_operator", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_spse_apply", c_int32, [c_void_p, _DP, _DP, c_int32, c_double]),
    ("ceres_hip_op_block_jacobi_update", c_int32, [c_void_p]),
    ("ceres_hip_op_schur_jacobi_update", c_int32, [c_void_p]),
    ("ceres_hip_get_preconditioner_blocks", c_int32, [c_void_p, c_int32, _DP, c_int64]),
    ("ceres_hip_op_precond_apply", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_schur_eliminate_dense", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_schur_storage_info", c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    ("ceres_hip_op_schur_eliminate_sparse", c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int64), _DP, c_int64, c_int64]),
    ("ceres_hip_op_schur_symmetric_multiply", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_eliminator_back_substitute", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_op_dot", c_int32, [c_void_p, _DP, _DP, c_int64, _DP]),
    ("ceres_hip_op_axpby", c_int32, [c_void_p, c_double, _DP, c_double, _DP, c_int64, _DP]),
    ("ceres_hip_lm_compute_step", c_int32, [c_void_p, _DP, _DP, POINTER(CLmOptions), _DP, POINTER(CLmResult)]),
    ("ceres_hip_lm_compute_step_device", c_int32, [c_void_p, c_void_p, c_void_p, POINTER(CLmOptions), c_void_p, POINTER(CLmResult)]),
    ("ceres_hip_get_lm_diagonal", c_int32, [c_void_p, _DP]),
    ("ceres_hip_values_begin", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_values_ready", c_int32, [c_void_p, c_int32, c_int32]),
    ("ceres_hip_values_end", c_int32, [c_void_p, _DP]),
    ("ceres_hip_get_stream_stats", c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
    ("ceres_hip_op_scale_columns", c_int32, [c_void_p, _DP, _DP]),
    ("ceres_hip_time_op", c_int32, [c_void_p, c_int32, c_int32, _DP]),
    ("ceres_hip_get_last_timing", c_int32, [c_void_p, POINTER(CTiming)]),
    ("ceres_hip_set_phase_timing", c_int32, [c_void_p, c_int32]),
    ("ceres_hip_debug_comm_loopback", c_int32, [c_void_p, c_int32]),
    ("ceres_hip_debug_comm_ghost_peers", c_int32, [c_void_p, c_int32, c_int64]),
    ("ceres_hip_debug_plan", c_int32, [POINTER(CBlockStructure), c_int32, POINTER(c_int32), POINTER(c_int64),
                                       POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_uint32),
                                       POINTER(c_int32), POINTER(c_int32), c_int64, c_char_p, c_int32]),
    ("ceres_hip_op_dense_cholesky_solve", c_int32, [c_void_p, c_int32, _DP, _DP, _DP, c_int32, POINTER(c_double), POINTER(c_int32)]),
    ("ceres_hip_debug_hybrid_plan", c_int32, [POINTER(CBlockStructure), c_int32, c_int32, c_int32, POINTER(c_int64)] + [POINTER(c_int32)] * 8 +
     [c_int64, c_int64, c_int64]),
    ("ceres_hip_debug_staged_x_plan", c_int32, [POINTER(CBlockStructure), c_int32, POINTER(c_int64), POINTER(c_int32), POINTER(c_int32), c_int64, c_int64]),
    ("ceres_hip_debug_long_rounds", c_int32, [POINTER(CBlockStructure), c_int32, c_int32, c_int32, c_int32, POINTER(c_int64)] +
     [POINTER(c_int32)] * 7 + [POINTER(c_uint32), c_int64, c_int64, c_int64]),
]

_lib = None


class CMinimizerOptions(ctypes.Structure):
    _fields_ = [("max_num_iterations", c_int32), ("jacobi_scaling", c_int32), ("max_consecutive_invalid_steps", c_int32),
                ("reserved", c_int32), ("initial_trust_region_radius", c_double), ("max_trust_region_radius", c_double),
                ("min_trust_region_radius", c_double), ("min_lm_diagonal", c_double), ("max_lm_diagonal", c_double),
                ("min_relative_decrease", c_double), ("eta", c_double), ("function_tolerance", c_double),
                ("gradient_tolerance", c_double), ("parameter_tolerance", c_double)]


class CIterationSummary(ctypes.Structure):
    _fields_ = [("cost", c_double), ("cost_change", c_double), ("gradient_max_norm", c_double), ("step_norm", c_double),
                ("relative_decrease", c_double), ("trust_region_radius", c_double), ("step_is_successful", c_int32),
                ("step_is_valid", c_int32), ("linear_solver_iterations", c_int32), ("linear_solver_termination", c_int32)]


MAX_LOGGED_ITERATIONS = 256


class CMinimizerSummary(ctypes.Structure):
    _fields_ = [("initial_cost", c_double), ("final_cost", c_double), ("num_successful_steps", c_int32),
                ("num_unsuccessful_steps", c_int32), ("num_linear_solves", c_int32), ("termination_type", c_int32),
                ("linear_solver_seconds", c_double), ("evaluation_seconds", c_double), ("total_seconds", c_double),
                ("num_iterations_logged", c_int32), ("reserved", c_int32),
                ("iterations", CIterationSummary * MAX_LOGGED_ITERATIONS), ("message", ctypes.c_char * 256)]


ABI += [
    ("ceres_hip_bal_create", c_void_p, [POINTER(COptions), c_int32, c_int32, c_int64, POINTER(c_int32), POINTER(c_int32), _DP]),
    ("ceres_hip_bal_destroy", None, [c_void_p]),
    ("ceres_hip_bal_last_error", c_char_p, [c_void_p]),
    ("ceres_hip_bal_linear_solver", c_void_p, [c_void_p]),
    ("ceres_hip_debug_bal_evaluate_tiles_timing", c_int32, [c_void_p, _DP, c_int32, c_int32, _DP]),
    ("ceres_hip_bal_sizes", c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    ("ceres_hip_bal_get_row_order", c_int32, [c_void_p, POINTER(c_int32)]),
    ("ceres_hip_bal_evaluate", c_int32, [c_void_p, _DP, _DP, _DP, _DP, _DP]),
    ("ceres_hip_minimizer_default_options", None, [POINTER(CMinimizerOptions)]),
    ("ceres_hip_bal_minimize", c_int32, [c_void_p, POINTER(CMinimizerOptions), _DP, POINTER(CMinimizerSummary)]),
]


def load_library():
    """dlopen csrc/libceres_hip.so and bind every ABI symbol; raises if anything is missing."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl (same
        # SONAMEs as /opt/rocm's).  Two HIP runtimes in one process cannot both own the GPU, so
        # if torch is going to be used in this process it must be loaded FIRST; our library's
        # NEEDED entries then bind to the copies torch already mapped.
        import sys
        if "torch" not in sys.modules and os.environ.get("CERES_HIP_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py` (build()) first; "
                               "there is no fall-back implementation")
        lib = ctypes.CDLL(path)
        for name, restype, argtypes in ABI:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.ceres_hip_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ABI version mismatch: the library says {lib.ceres_hip_abi_version()}, this binding is written for {ABI_VERSION}")
        _lib = lib
    return _lib


def device_count() -> int:
    return int(load_library().ceres_hip_device_count())


class HipError(RuntimeError):
    pass


@dataclass
class LinearSolverOptions:
    """LinearSolver::Options (internal/ceres/linear_solver.h:148-230), the fields this path reads."""
    type: int = ITERATIVE_SCHUR
    preconditioner_type: int = JACOBI
    min_num_iterations: int = 1          # reference default
    max_num_iterations: int = 1          # reference default (!); callers set it
    residual_reset_period: int = 10
    elimination_groups: List[int] = field(default_factory=list)
    # device-side knobs (not in the reference)
    device: int = 0
    force_generic_path: bool = False
    cg_check_interval: int = 0
    jacobian_storage: int = 0   # 1: fp32 tiles on the <2,3,9> path (accuracy mode, not parity)
    max_num_spse_iterations: int = 5
    use_spse_initialization: bool = False
    spse_tolerance: float = 0.1
    use_explicit_schur_complement: bool = False   # ITERATIVE_SCHUR on an explicitly formed (dense) S; SCHUR_JACOBI only


@dataclass
class PerSolveOptions:
    """LinearSolver::PerSolveOptions (internal/ceres/linear_solver.h:237-318)."""
    D: Optional[np.ndarray] = None
    q_tolerance: float = 0.0
    r_tolerance: float = 0.0


@dataclass
class Summary:
    residual_norm: float = -1.0
    num_iterations: int = -1
    termination_type: int = FAILURE
    message: str = ""

    @property
    def termination_name(self):
        return TERMINATION_NAMES.get(self.termination_type, "?")


def _f64(a, n=None, name="array"):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.shape[0] < n:
        raise ValueError(f"{name} has {a.shape[0]} elements, need {n}")
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(_DP)


class HipLinearSolver:
    """One device-resident solver instance = one ceres::internal::LinearSolver object.

    The structure is set once (the reference guarantees constant sparsity per instance,
    internal/ceres/linear_solver.h:137-142); each `solve` uploads values / b / D and returns x.
    """

    def __init__(self, options: LinearSolverOptions, comm_id: Optional[bytes] = None, rank: int = 0,
                 world_size: int = 1, loopback_world: int = 0, p2p_exchange=None, p2p_max_elements: int = 0, ghost_world: int = 0):
        """comm_id: RCCL unique id (ceres_hip_comm_init).  p2p_exchange: callable(bytes) -> list of every rank's bytes in
        rank order (e.g. a torch.distributed all_gather); connects the one-shot peer-to-peer all-reduce for vectors of
        up to p2p_max_elements doubles (99 * number of F blocks covers a step: blocks, rhs and column norms go through ONE all-reduce)."""
        self._lib = load_library()
        self.options = options
        nelim = options.elimination_groups[0] if options.elimination_groups else 0
        c = COptions(options.type, options.preconditioner_type, options.min_num_iterations,
                     options.max_num_iterations, options.residual_reset_period, nelim, options.device,
                     int(options.force_generic_path), options.cg_check_interval, options.jacobian_storage,
                     options.max_num_spse_iterations, int(options.use_spse_initialization), options.spse_tolerance,
                     int(options.use_explicit_schur_complement))
        self._h = self._lib.ceres_hip_create(byref(c))
        if not self._h:
            raise HipError(self._lib.ceres_hip_last_error(None).decode())
        self.bs = None
        self._values_extent = None
        if world_size > 1 and comm_id is not None:
            buf = (c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(comm_id)
            self._check(self._lib.ceres_hip_comm_init(self._h, buf, rank, world_size))
        self.p2p_ok = False
        self.p2p_error = ""
        if world_size > 1 and p2p_exchange is not None:
            # every step is collective: a rank that fails still takes part in the exchange, and reports through p2p_ok
            mine = (c_uint8 * IPC_HANDLE_BYTES)()
            ok = self._lib.ceres_hip_comm_p2p_prepare(self._h, rank, world_size, int(p2p_max_elements), mine) == 0
            if not ok:
                self.p2p_error = self._lib.ceres_hip_last_error(self._h).decode()
            handles = p2p_exchange(bytes(mine))
            if len(handles) != world_size or any(len(h) != IPC_HANDLE_BYTES for h in handles):
                raise HipError("p2p_exchange must return one 64-byte handle per rank")
            if ok:
                allh = (c_uint8 * (IPC_HANDLE_BYTES * world_size)).from_buffer_copy(b"".join(handles))
                ok = self._lib.ceres_hip_comm_p2p_connect(self._h, allh) == 0
                if not ok:
                    self.p2p_error = self._lib.ceres_hip_last_error(self._h).decode()
            self.p2p_ok = ok
            if not ok and comm_id is None:
                raise HipError("peer-to-peer communicator failed and there is no RCCL communicator: " + self.p2p_error)
        if loopback_world > 1:  # debug: sharded code paths on one GPU (see include/ceres_hip.h)
            self._check(self._lib.ceres_hip_debug_comm_loopback(self._h, loopback_world))
        if ghost_world > 1:  # measurement: rank 0 of ghost_world ranks, the peers are local dummy buffers (include/ceres_hip.h)
            self._check(self._lib.ceres_hip_debug_comm_ghost_peers(self._h, ghost_world, int(p2p_max_elements)))
            self.p2p_ok = True

    def p2p_selftest(self) -> bool:
        """Collective: all-reduces of known values through the peer-to-peer path.  ceres_hip_comm_p2p_connect has already run it once
        (the path is never enabled untested); this repeats it on request."""
        if not self.p2p_ok:
            return False
        rc = self._lib.ceres_hip_comm_p2p_selftest(self._h)
        if rc != 0:
            self.p2p_error = self._lib.ceres_hip_last_error(self._h).decode()
            self.p2p_ok = False
        return rc == 0

    def allreduce_timing(self, n: int, iters: int = 200) -> float:
        """Collective: average microseconds per all-reduce of n doubles (measurement helper)."""
        out = np.zeros(1)
        self._check(self._lib.ceres_hip_debug_allreduce_timing(self._h, int(n), int(iters), _p(out)))
        return float(out[0])

    def p2p_disable(self):
        self._check(self._lib.ceres_hip_comm_p2p_disable(self._h))
        self.p2p_ok = False

    # -- plumbing ----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise HipError(f"ceres_hip error {rc}: {self._lib.ceres_hip_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ceres_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- structure ---------------------------------------------------------
    def set_structure(self, bs: BlockStructure):
        self.bs = bs
        self._cbs = bs.as_ctypes()
        self._check(self._lib.ceres_hip_set_structure(self._h, byref(self._cbs)))
        self._info = self.info()
        self._values_extent = bs.values_extent()  # O(num_cells) to compute: once, not per solve

    def info(self) -> CInfo:
        i = CInfo()
        self._check(self._lib.ceres_hip_get_info(self._h, byref(i)))
        return i

    # -- the boundary call -------------------------------------------------
    def solve(self, values, b, per_solve_options: PerSolveOptions = PerSolveOptions()):
        """LinearSolver::Solve.  Returns (x, Summary).  x is pre-poisoned with NaN like the
        caller does (internal/ceres/levenberg_marquardt_strategy.cc:110)."""
        n = self._info
        values = _f64(values, self._values_extent, "values")
        b = _f64(b, n.num_rows, "b")
        D = _f64(per_solve_options.D, n.num_cols, "D")
        x = np.full(n.num_cols, np.nan)
        s = CSummary()
        rc = self._lib.ceres_hip_solve(self._h, _p(values), _p(b), _p(D), per_solve_options.q_tolerance,
                                       per_solve_options.r_tolerance, _p(x), byref(s))
        summary = Summary(s.residual_norm, s.num_iterations, s.termination_type, s.message.decode(errors="replace"))
        if rc != 0:
            summary.termination_type = FATAL_ERROR
            summary.message = self._lib.ceres_hip_last_error(self._h).decode()
        return x, summary

    def solve_unchanged_values(self, per_solve: "PerSolveOptions"):
        """LinearSolver::Solve again on the values and b of the previous solve() / load() with a new D (the retry after a rejected
        trust-region step): only D crosses PCIe, the tiles are not rebuilt."""
        n = self._info
        D = None if per_solve.D is None else _f64(per_solve.D, n.num_cols, "D")
        x = np.full(n.num_cols, np.nan)
        s = CSummary()
        rc = self._lib.ceres_hip_solve_unchanged_values(self._h, _p(D) if D is not None else None, per_solve.q_tolerance,
                                                        per_solve.r_tolerance, _p(x), byref(s))
        summary = Summary(s.residual_norm, s.num_iterations, s.termination_type, s.message.decode(errors="replace"))
        if rc != 0:
            summary.termination_type = FATAL_ERROR
            summary.message = self._lib.ceres_hip_last_error(self._h).decode()
        return x, summary

    def solve_device(self, d_values: int, d_b: int, d_D: int, d_x: int, q_tolerance=0.0, r_tolerance=0.0):
        """Same with raw device pointers (e.g. torch tensor .data_ptr()); nothing crosses PCIe."""
        s = CSummary()
        rc = self._lib.ceres_hip_solve_device(self._h, d_values, d_b, d_D or None, q_tolerance, r_tolerance, d_x, byref(s))
        summary = Summary(s.residual_norm, s.num_iterations, s.termination_type, s.message.decode(errors="replace"))
        if rc != 0:
            summary.termination_type = FATAL_ERROR
            summary.message = self._lib.ceres_hip_last_error(self._h).decode()
        return summary

    # -- f1: one trust-region step's linear algebra on the device ----------------
    def lm_compute_step(self, values, residuals, radius, eta=0.1, min_diagonal=1e-6, max_diagonal=1e32,
                        reuse_diagonal=False, values_unchanged=False, out=None):
        """LevenbergMarquardtStrategy::ComputeStep + the model-cost bookkeeping of
        TrustRegionMinimizer::ComputeTrustRegionStep.  Returns (step, Summary, model_cost_change).
        values_unchanged: the retry after a rejected step — the solver keeps the Jacobian and residuals it holds (values / residuals
        may be None): no host-to-device copy, no re-layout."""
        n = self._info
        o = CLmOptions(radius, min_diagonal, max_diagonal, eta, int(reuse_diagonal), int(values_unchanged))
        r = CLmResult()
        step = np.full(n.num_cols, np.nan) if out is None else _f64(out, n.num_cols, "out")   # out: a caller-owned (e.g. pinned) step buffer
        if values_unchanged:
            pv = pr = None
        else:
            values = _f64(values, self._values_extent, "values")
            residuals = _f64(residuals, n.num_rows, "residuals")
            pv, pr = _p(values), _p(residuals)
        self._check(self._lib.ceres_hip_lm_compute_step(self._h, pv, pr, byref(o), _p(step), byref(r)))
        s = r.linear_solver
        return step, Summary(s.residual_norm, s.num_iterations, s.termination_type, s.message.decode(errors="replace")), \
            float(r.model_cost_change)

    def lm_compute_step_device(self, d_values: int, d_residuals: int, d_step: int, radius, eta=0.1, min_diagonal=1e-6,
                               max_diagonal=1e32, reuse_diagonal=False, values_unchanged=False):
        o = CLmOptions(radius, min_diagonal, max_diagonal, eta, int(reuse_diagonal), int(values_unchanged))
        r = CLmResult()
        self._check(self._lib.ceres_hip_lm_compute_step_device(self._h, d_values, d_residuals, byref(o), d_step, byref(r)))
        s = r.linear_solver
        return Summary(s.residual_norm, s.num_iterations, s.termination_type, s.message.decode(errors="replace")), \
            float(r.model_cost_change), bool(r.step_is_finite)

    def lm_stepper(self, d_values: int, d_residuals: int, d_step: int, radius, eta=0.1, min_diagonal=1e-6, max_diagonal=1e32):
        """A callable that runs lm_compute_step_device on fixed device arrays with the argument structs built ONCE (what a C++ caller
        does: the structs live across trust-region iterations); returns the raw CLmResult, which the next call overwrites.  A timed
        loop through it pays one ctypes call per step and none of this wrapper's per-call Python work (three struct constructions,
        a message decode), during which the device idles between two steps."""
        o = CLmOptions(radius, min_diagonal, max_diagonal, eta, 0, 0)
        r = CLmResult()
        fn, h, po, pr, check = self._lib.ceres_hip_lm_compute_step_device, self._h, byref(o), byref(r), self._check
        keep = (o, r)

        def step():
            rc = fn(h, d_values, d_residuals, po, d_step, pr)
            if rc != 0:
                check(rc)
            return keep[1]
        return step

    # -- the upload hidden behind the evaluator ---------------------------------
    def values_begin(self, values, residuals):
        """values / residuals: the (pinned) host arrays the evaluator is about to fill; kept alive until values_end."""
        self._stream_keep = (_f64(values, self._values_extent, "values"), _f64(residuals, self._info.num_rows, "residuals"))
        if self._stream_keep[0] is not values or self._stream_keep[1] is not residuals:
            raise ValueError("values_begin needs contiguous float64 arrays (they are filled in place while rows go up)")
        self._check(self._lib.ceres_hip_values_begin(self._h, _p(values), _p(residuals)))

    def values_ready(self, first_row_block: int, num_row_blocks: int):
        """Thread-safe: row blocks [first, first + n) are complete."""
        rc = self._lib.ceres_hip_values_ready(self._h, int(first_row_block), int(num_row_blocks))
        if rc != 0:
            raise HipError(f"ceres_hip_values_ready: error {rc}")

    def values_end(self, column_scale=None):
        cs = _f64(column_scale, self._info.num_cols, "column_scale")
        self._check(self._lib.ceres_hip_values_end(self._h, _p(cs)))
        self._stream_keep = None

    def stream_stats(self):
        a, b, k = c_int64(0), c_int64(0), c_int32(0)
        self._check(self._lib.ceres_hip_get_stream_stats(self._h, byref(a), byref(b), byref(k)))
        return int(a.value), int(b.value), int(k.value)

    def lm_diagonal(self):
        out = np.full(self._info.num_cols, np.nan)
        self._check(self._lib.ceres_hip_get_lm_diagonal(self._h, _p(out)))
        return out

    def scale_columns(self, scale):
        scale = _f64(scale, self._info.num_cols, "scale")
        out = np.full(self._values_extent, np.nan)
        self._check(self._lib.ceres_hip_op_scale_columns(self._h, _p(scale), _p(out)))
        return out

    def set_phase_timing(self, enable: bool = True):
        """Record the per-phase HIP events of the following solves / steps (last_timing); off by default: each event record between two
        kernels idles the device for about 6 us."""
        self._check(self._lib.ceres_hip_set_phase_timing(self._h, int(bool(enable))))

    def last_timing(self) -> CTiming:
        t = CTiming()
        self._check(self._lib.ceres_hip_get_last_timing(self._h, byref(t)))
        return t

    # -- operator level ----------------------------------------------------
    def load(self, values, b=None, D=None):
        n = self._info
        self._keep = (_f64(values, self._values_extent, "values"), _f64(b, n.num_rows, "b"), _f64(D, n.num_cols, "D"))
        self._check(self._lib.ceres_hip_load(self._h, *map(_p, self._keep)))

    def load_device(self, d_values: int, d_b: int = 0, d_D: int = 0):
        self._check(self._lib.ceres_hip_load_device(self._h, d_values, d_b or None, d_D or None))

    def _xy(self, fn, x, n_out, y=None):
        x = _f64(x)
        y = np.zeros(n_out) if y is None else _f64(y).copy()
        self._check(fn(self._h, _p(x), _p(y)))
        return y

    def right_multiply(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_right_multiply, x, self._info.num_rows, y)

    def left_multiply(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_left_multiply, x, self._info.num_cols, y)

    # PartitionedMatrixView products (internal/ceres/partitioned_matrix_view_impl.h:112-375); all accumulate into y
    def right_multiply_e(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_right_multiply_e, x, self._info.num_rows, y)

    def right_multiply_f(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_right_multiply_f, x, self._info.num_rows, y)

    def left_multiply_e(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_left_multiply_e, x, self._info.num_cols_e, y)

    def left_multiply_f(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_left_multiply_f, x, self._info.num_cols_f, y)

    def _block_diagonal(self, fn, sizes):
        out = np.full(int((sizes.astype(np.int64) ** 2).sum()), np.nan)
        self._check(fn(self._h, _p(out), out.shape[0]))
        return out

    def block_diagonal_ete(self):
        return self._block_diagonal(self._lib.ceres_hip_op_block_diagonal_ete, self.bs.col_block_size[: self._info.num_e_blocks])

    def block_diagonal_ftf(self):
        return self._block_diagonal(self._lib.ceres_hip_op_block_diagonal_ftf, self.bs.col_block_size[self._info.num_e_blocks:])

    def squared_column_norm(self):
        out = np.zeros(self._info.num_cols)
        self._check(self._lib.ceres_hip_op_squared_column_norm(self._h, _p(out)))
        return out

    def jtjx(self, x):
        return self._xy(self._lib.ceres_hip_op_jtjx, x, self._info.num_cols, np.full(self._info.num_cols, np.nan))

    def jtb(self):
        out = np.full(self._info.num_cols, np.nan)
        self._check(self._lib.ceres_hip_op_jtb(self._h, _p(out)))
        return out

    def schur_init(self):
        self._check(self._lib.ceres_hip_op_schur_init(self._h))

    def schur_rhs(self):
        out = np.full(self._info.num_cols_f, np.nan)
        self._check(self._lib.ceres_hip_get_schur_rhs(self._h, _p(out)))
        return out

    def ete_inverse(self):
        sizes = self.bs.col_block_size[: self._info.num_e_blocks].astype(np.int64)
        out = np.full(int((sizes ** 2).sum()), np.nan)
        self._check(self._lib.ceres_hip_get_ete_inverse(self._h, _p(out), out.shape[0]))
        return out

    def schur_sx(self, x):
        return self._xy(self._lib.ceres_hip_op_schur_sx, x, self._info.num_cols_f, np.full(self._info.num_cols_f, np.nan))

    def power_series_operator(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_power_series_operator, x, self._info.num_cols_f, y)

    def spse_apply(self, x, max_num_spse_iterations=5, spse_tolerance=0.0):
        x = _f64(x)
        y = np.full(self._info.num_cols_f, np.nan)
        self._check(self._lib.ceres_hip_op_spse_apply(self._h, _p(x), _p(y), max_num_spse_iterations, spse_tolerance))
        return y

    def back_substitute(self, z):
        z = _f64(z) if z is not None else np.zeros(1)
        x = np.full(self._info.num_cols, np.nan)
        self._check(self._lib.ceres_hip_op_back_substitute(self._h, _p(z), _p(x)))
        return x

    def block_jacobi_update(self):
        self._check(self._lib.ceres_hip_op_block_jacobi_update(self._h))

    def schur_jacobi_update(self):
        self._check(self._lib.ceres_hip_op_schur_jacobi_update(self._h))

    def preconditioner_blocks(self, not_inverted=False):
        i = self._info
        sizes = self.bs.col_block_size.astype(np.int64)
        if self.options.type == ITERATIVE_SCHUR:
            sizes = sizes[i.num_e_blocks:]
        out = np.full(int((sizes ** 2).sum()), np.nan)
        self._check(self._lib.ceres_hip_get_preconditioner_blocks(self._h, int(not_inverted), _p(out), out.shape[0]))
        return out

    def precond_apply(self, x, y=None):
        n = self._info.num_cols_f if self.options.type == ITERATIVE_SCHUR else self._info.num_cols
        return self._xy(self._lib.ceres_hip_op_precond_apply, x, n, y)

    def schur_eliminate_dense(self, want_rhs=True):
        n = self._info.num_cols_f
        lhs, rhs = np.full(n * n, np.nan), (np.full(n, np.nan) if want_rhs else None)
        self._check(self._lib.ceres_hip_op_schur_eliminate_dense(self._h, _p(lhs), _p(rhs)))
        return lhs.reshape(n, n), rhs

    def schur_eliminate_sparse(self):
        """(pair_i, pair_j, pair_offset, values) of the explicit Schur complement in BlockRandomAccessSparseMatrix storage."""
        npairs, nvals = c_int64(), c_int64()
        self._check(self._lib.ceres_hip_schur_storage_info(self._h, byref(npairs), byref(nvals)))
        pi, pj = np.zeros(npairs.value, np.int32), np.zeros(npairs.value, np.int32)
        off, vals = np.zeros(npairs.value, np.int64), np.full(nvals.value, np.nan)
        ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
        self._check(self._lib.ceres_hip_op_schur_eliminate_sparse(self._h, ip(pi), ip(pj), off.ctypes.data_as(POINTER(c_int64)), _p(vals),
                                                                  npairs.value, nvals.value))
        return pi, pj, off, vals

    def schur_symmetric_multiply(self, x, y=None):
        return self._xy(self._lib.ceres_hip_op_schur_symmetric_multiply, x, self._info.num_cols_f, y)

    def eliminator_back_substitute(self, z):
        x = np.full(self._info.num_cols, np.nan)
        self._check(self._lib.ceres_hip_op_eliminator_back_substitute(self._h, _p(_f64(z)), _p(x)))
        return x

    def dot(self, x, y):
        x, y = _f64(x), _f64(y)
        out = np.zeros(1)
        self._check(self._lib.ceres_hip_op_dot(self._h, _p(x), _p(y), x.shape[0], _p(out)))
        return float(out[0])

    def axpby(self, a, x, b, y):
        x, y = _f64(x), _f64(y)
        z = np.zeros_like(x)
        self._check(self._lib.ceres_hip_op_axpby(self._h, a, _p(x), b, _p(y), x.shape[0], _p(z)))
        return z

    def dense_cholesky_solve(self, A: np.ndarray, b: np.ndarray, repeats: int = 1):
        """DenseCholesky::FactorAndSolve (internal/ceres/dense_cholesky.cc) on a caller-supplied SPD matrix (upper triangle
        authoritative): returns (x, average factorisation ms, failed).  What DENSE_SCHUR runs on its reduced system."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        n = A.shape[0]
        assert A.shape == (n, n) and b.shape == (n,)
        x = np.zeros(n)
        ms, failed = c_double(0.0), c_int32(0)
        self._check(self._lib.ceres_hip_op_dense_cholesky_solve(self._h, n, _p(A), _p(b), _p(x), repeats, byref(ms), byref(failed)))
        return x, float(ms.value), bool(failed.value)

    def time_op(self, op: int, iters: int = 20) -> float:
        """Average milliseconds per application (HIP events on the solver's stream)."""
        out = np.zeros(1)
        self._check(self._lib.ceres_hip_time_op(self._h, op, iters, _p(out)))
        return float(out[0])


def comm_unique_id() -> bytes:
    buf = (c_uint8 * UNIQUE_ID_BYTES)()
    rc = load_library().ceres_hip_comm_get_unique_id(buf)
    if rc != 0:
        raise HipError(f"ncclGetUniqueId failed ({rc})")
    return bytes(buf)


def create_linear_solver(options: LinearSolverOptions, bs: BlockStructure, **kw) -> HipLinearSolver:
    """LinearSolver::Create with the reference's fall-backs (internal/ceres/linear_solver.cc:51-73,
    internal/ceres/preconditioner.cc:39-47): ITERATIVE_SCHUR without eliminated blocks becomes CGNR,
    and SCHUR_JACOBI becomes JACOBI with it."""
    import copy
    o = copy.copy(options)
    nelim = o.elimination_groups[0] if o.elimination_groups else 0
    if o.type == ITERATIVE_SCHUR and nelim == 0:
        o.type = CGNR
        if o.preconditioner_type == SCHUR_JACOBI:
            o.preconditioner_type = JACOBI
    s = HipLinearSolver(o, **kw)
    s.set_structure(bs)
    return s


def debug_plan(bs: BlockStructure, num_eliminate_blocks: int):
    """The host-side tile packing plan (no device needed).  Returns a dict or {'eligible': False, 'why': ...}."""
    lib = load_library()
    c = bs.as_ctypes()
    eligible, n_tiles = c_int32(), c_int64()
    why = ctypes.create_string_buffer(256)
    null32, nullu = POINTER(c_int32)(), POINTER(c_uint32)()
    rc = lib.ceres_hip_debug_plan(byref(c), num_eliminate_blocks, byref(eligible), byref(n_tiles), null32, null32, null32,
                                  nullu, null32, null32, 0, why, 256)
    if rc != 0 or not eligible.value:
        return {"eligible": False, "why": why.value.decode()}
    nt = n_tiles.value
    ns = nt * 64
    row, cam, pt = (np.zeros(ns, np.int32) for _ in range(3))
    seg = np.zeros(ns, np.uint32)
    kind, aux = np.zeros(nt, np.int32), np.zeros(nt, np.int32)
    ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
    rc = lib.ceres_hip_debug_plan(byref(c), num_eliminate_blocks, byref(eligible), byref(n_tiles), ip(row), ip(cam), ip(pt),
                                  seg.ctypes.data_as(POINTER(c_uint32)), ip(kind), ip(aux), ns, why, 256)
    assert rc == 0
    return {"eligible": True, "n_tiles": nt, "slot_row": row, "slot_cam": cam, "slot_pt": pt, "seg_first": seg & 0xff,
            "seg_last": (seg >> 8) & 0xff, "valid": (seg >> 16) & 1, "tail_a": (seg >> 17) & 63, "has_a": (seg >> 23) & 1,
            "tail_b": (seg >> 24) & 63, "has_b": (seg >> 30) & 1, "tile_kind": kind, "tile_aux": aux}


def debug_staged_x_plan(bs: BlockStructure, num_eliminate_blocks: int):
    """Which cameras' part of x the streaming kernels keep in LDS beside the accumulators (csrc/plan.cc, BalPlan::xhot_cam; no device
    needed).  Returns None for structures off the fused path."""
    lib = load_library()
    c = bs.as_ctypes()
    counts = (c_int64 * 4)()
    n32 = POINTER(c_int32)()
    fn = lib.ceres_hip_debug_staged_x_plan
    if fn(byref(c), c_int32(num_eliminate_blocks), counts, n32, n32, c_int64(0), c_int64(0)) != 0:
        return None
    nt, ncam, nst, acc_bytes = (int(v) for v in counts)
    staged, word = np.zeros(max(nst, 1), np.int32), np.zeros(nt * 64, np.int32)
    ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
    assert fn(byref(c), c_int32(num_eliminate_blocks), counts, ip(staged), ip(word), c_int64(nst), c_int64(nt * 64)) == 0
    return {"n_tiles": nt, "n_cameras": ncam, "staged_cam": staged[:nst], "accumulator_bytes": acc_bytes, "slot_word": word}


def debug_hybrid_plan(bs: BlockStructure, num_eliminate_blocks: int, groups: int, rows: int):
    """The camera-accumulation plan for more cameras than LDS rows (csrc/plan.cc; no device needed): hybrid for groups >= 2,
    spill-everything for groups = 0.  Returns None when the structure is not <2,3,9>-shaped or its cameras fit in LDS."""
    lib = load_library()
    c = bs.as_ctypes()
    counts = (c_int64 * 8)()
    n32 = POINTER(c_int32)()
    fn = lib.ceres_hip_debug_hybrid_plan
    rc = fn(byref(c), c_int32(num_eliminate_blocks), c_int32(groups), c_int32(rows), counts, n32, n32, n32, n32, n32, n32, n32, n32,
            c_int64(0), c_int64(0), c_int64(0))
    if rc != 0:
        return None
    nt, hyb, k, kh, flush0, ring_rows, ne, nu = (int(v) for v in counts)
    word, row = np.zeros(nt * 64, np.int32), np.zeros(nt * 64, np.int32)
    zbase, grp = np.zeros(nt, np.int32), np.zeros(max(groups, 0) + 1, np.int32)
    ent, ucam, ub, ue = np.zeros(ne, np.int32), np.zeros(nu, np.int32), np.zeros(nu, np.int32), np.zeros(nu, np.int32)
    ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
    rc = fn(byref(c), c_int32(num_eliminate_blocks), c_int32(groups), c_int32(rows), counts, ip(word), ip(row), ip(zbase), ip(grp), ip(ent),
            ip(ucam), ip(ub), ip(ue), c_int64(nt * 64), c_int64(ne), c_int64(nu))
    assert rc == 0
    valid = word != -1   # (a spilled slot's word has its top bits set: negative as int32, but never -1)
    uw = word.view(np.uint32)
    return {"n_tiles": nt, "hybrid": bool(hyb), "rows": k, "hot_rows": kh, "flush_row0": flush0, "ring_rows": ring_rows,
            "valid": valid, "slot_cam": np.where(valid, (uw & 0xFFFFF).astype(np.int64), -1),
            "slot_acc": np.where(valid, (uw >> 20).astype(np.int64), 0xFFF),
            "slot_row": row, "tile_zbase": zbase, "grp_tile_ptr": grp if hyb else None, "entry_row": ent, "unit_cam": ucam,
            "unit_begin": ub, "unit_end": ue}


def debug_long_rounds(bs: BlockStructure, num_eliminate_blocks: int, renumber: bool = True, groups: int = 0, rows: int = 0):
    """Where the plan puts the points of more than 64 observations and the rounds the streaming kernels take them in
    (csrc/plan.cc; no device needed).  Returns None when the structure is not <2,3,9>-shaped."""
    lib = load_library()
    c = bs.as_ctypes()
    counts = (c_int64 * 5)()
    n32, nu = POINTER(c_int32)(), POINTER(c_uint32)()
    fn = lib.ceres_hip_debug_long_rounds
    args = (byref(c), c_int32(num_eliminate_blocks), c_int32(1 if renumber else 0), c_int32(groups), c_int32(rows), counts)
    if fn(*args, n32, n32, n32, n32, n32, n32, n32, nu, c_int64(0), c_int64(0), c_int64(0)) != 0:
        return None
    nt, nr, nrounds, behind, nseq = (int(v) for v in counts)
    kind, aux = np.zeros(nt, np.int32), np.zeros(nt, np.int32)
    rtp, lp, rp = np.zeros(nr + 1, np.int32), np.zeros(nr, np.int32), np.zeros(nr + 1, np.int32)
    sp, flag = np.zeros(nrounds + 1, np.int32), np.zeros(max(nrounds, 1), np.int32)
    words = np.zeros(max(nrounds, 1) * 8, np.uint32)
    ip = lambda a: a.ctypes.data_as(POINTER(c_int32))
    rc = fn(*args, ip(kind), ip(aux), ip(rtp), ip(lp), ip(rp), ip(sp), ip(flag), words.ctypes.data_as(POINTER(c_uint32)), c_int64(nt),
            c_int64(nr), c_int64(nrounds))
    assert rc == 0
    return {"n_tiles": nt, "tile_kind": kind, "tile_aux": aux, "range_tile_ptr": rtp, "long_ptr": lp, "round_ptr": rp,
            "seq_ptr": sp[:nseq + 1], "round_flag": flag[:nrounds], "round_word": words[:nrounds * 8].reshape(nrounds, 8),
            "long_behind": bool(behind)}


CONVERGENCE, MINIMIZER_NO_CONVERGENCE, MINIMIZER_FAILURE = 0, 1, 2


class BalProblem:
    """Bundle adjustment in BAL form on the device: the Evaluator of the reduced,
    Schur-ordered program (internal/ceres/evaluator.h:98-158) for the Snavely reprojection error
    (examples/snavely_reprojection_error.h:53-105, examples/bal_problem.cc:75-135) and
    TrustRegionMinimizer::Minimize around the linear solver `options` selects.

    state = [3 doubles per point | 9 doubles per camera]; `state_from_bal` / `state_to_bal`
    convert from the file order (cameras, then points)."""

    def __init__(self, options: LinearSolverOptions, num_cameras, num_points, camera_index, point_index, observations):
        self._lib = load_library()
        self.options = options
        self.num_cameras, self.num_points = int(num_cameras), int(num_points)
        cam = np.ascontiguousarray(camera_index, dtype=np.int32)
        pt = np.ascontiguousarray(point_index, dtype=np.int32)
        obs = _f64(observations).reshape(-1)
        self.num_observations = int(cam.shape[0])
        if pt.shape[0] != self.num_observations or obs.shape[0] != 2 * self.num_observations:
            raise ValueError("camera_index, point_index and observations disagree on the number of observations")
        c = COptions(options.type, options.preconditioner_type, options.min_num_iterations,
                     options.max_num_iterations, options.residual_reset_period, self.num_points, options.device,
                     int(options.force_generic_path), options.cg_check_interval, options.jacobian_storage,
                     options.max_num_spse_iterations, int(options.use_spse_initialization), options.spse_tolerance,
                     int(options.use_explicit_schur_complement))
        self._h = self._lib.ceres_hip_bal_create(byref(c), self.num_cameras, self.num_points, self.num_observations,
                                                 cam.ctypes.data_as(POINTER(c_int32)), pt.ctypes.data_as(POINTER(c_int32)),
                                                 _p(obs))
        if not self._h:
            raise HipError(self._lib.ceres_hip_bal_last_error(None).decode())
        n, m, v = c_int64(), c_int64(), c_int64()
        self._check(self._lib.ceres_hip_bal_sizes(self._h, byref(n), byref(m), byref(v)))
        self.num_parameters, self.num_residuals, self.num_jacobian_values = n.value, m.value, v.value

    @classmethod
    def from_file(cls, options: LinearSolverOptions, filename):
        """BALProblem(filename) (examples/bal_problem.cc:75-135).  Returns (problem, initial state)."""
        from . import problems
        nc, npts, cam, pt, obs, par = problems.read_bal(filename)
        p = cls(options, nc, npts, cam, pt, obs)
        return p, p.state_from_bal(par)

    def _check(self, rc):
        if rc != 0:
            raise HipError(f"ceres_hip error {rc}: {self._lib.ceres_hip_bal_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ceres_hip_bal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solver_info(self) -> CInfo:
        """ceres_hip_get_info of the linear solver inside (which kernels its structure runs on, the tile plan)."""
        i = CInfo()
        self._check(self._lib.ceres_hip_get_info(self._lib.ceres_hip_bal_linear_solver(self._h), byref(i)))
        return i

    def preconditioner_blocks(self, not_inverted=False):
        """The block-diagonal preconditioner of the LAST linear solve inside minimize (camera blocks; CGNR: point blocks first)."""
        n = 81 * self.num_cameras + (9 * self.num_points if self.options.type == CGNR else 0)
        out = np.full(n, np.nan)
        inner = self._lib.ceres_hip_bal_linear_solver(self._h)
        rc = self._lib.ceres_hip_get_preconditioner_blocks(inner, int(not_inverted), _p(out), n)
        if rc != 0:
            raise HipError(f"ceres_hip error {rc}: {self._lib.ceres_hip_last_error(inner).decode()}")
        return out

    def evaluate_tiles_timing(self, state, flags=0, iters=20) -> float:
        """Average microseconds of the tile-order evaluator's launch (experiments: flags switch groups of its stores off)."""
        x = _f64(state, self.num_parameters)
        out = np.zeros(1)
        self._check(self._lib.ceres_hip_debug_bal_evaluate_tiles_timing(self._h, _p(x), int(flags), int(iters), _p(out)))
        return float(out[0])

    def row_order(self):
        out = np.empty(self.num_observations, dtype=np.int32)
        self._check(self._lib.ceres_hip_bal_get_row_order(self._h, out.ctypes.data_as(POINTER(c_int32))))
        return out

    def state_from_bal(self, parameters):
        """BAL file order (9 per camera, then 3 per point) -> state."""
        a = _f64(parameters, 9 * self.num_cameras + 3 * self.num_points)
        return np.concatenate([a[9 * self.num_cameras:9 * self.num_cameras + 3 * self.num_points], a[:9 * self.num_cameras]])

    def state_to_bal(self, state):
        a = _f64(state, self.num_parameters)
        return np.concatenate([a[3 * self.num_points:], a[:3 * self.num_points]])

    def evaluate(self, state, residuals=False, gradient=False, jacobian=False):
        """Evaluator::Evaluate: returns (cost, residuals|None, gradient|None, jacobian values|None)."""
        x = _f64(state, self.num_parameters)
        cost = np.zeros(1)
        r = np.empty(self.num_residuals) if residuals else None
        g = np.empty(self.num_parameters) if gradient else None
        v = np.empty(self.num_jacobian_values) if jacobian else None
        self._check(self._lib.ceres_hip_bal_evaluate(self._h, _p(x), _p(cost), _p(r), _p(g), _p(v)))
        return float(cost[0]), r, g, v

    def minimize(self, state, **opts):
        """TrustRegionMinimizer::Minimize (LEVENBERG_MARQUARDT).  Returns (state, CMinimizerSummary)."""
        o = CMinimizerOptions()
        self._lib.ceres_hip_minimizer_default_options(byref(o))
        for k, val in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown minimizer option {k}")
            setattr(o, k, val)
        x = _f64(state, self.num_parameters).copy()
        S = CMinimizerSummary()
        self._check(self._lib.ceres_hip_bal_minimize(self._h, byref(o), _p(x), byref(S)))
        return x, S
