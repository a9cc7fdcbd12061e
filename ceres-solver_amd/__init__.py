"""MI355X-native Levenberg-Marquardt linear-solve path for Ceres-style bundle
adjustment.  The compute path is the HIP library behind
include/ceres_hip.h (built from csrc/ into csrc/libceres_hip.so); this package is
the Python host side above that C ABI.

The directory name contains a hyphen (it is fixed by the build contract), so it is
not importable with a plain `import`; `__graft_entry__.load_package()` registers it
as the module `ceres_solver_amd`.
"""
from .block_structure import BlockStructure, CBlockStructure  # noqa: F401
from . import problems  # noqa: F401

__all__ = ["BlockStructure", "CBlockStructure", "problems"]


def __getattr__(name):
    # hip_solver pulls in ctypes bindings to libceres_hip.so; keep the import lazy so
    # that structure/problem helpers work where the library has not been built.
    if name in ("hip_solver", "HipLinearSolver", "LinearSolverOptions", "PerSolveOptions", "Summary",
                "partition", "library_path", "load_library"):
        import importlib
        mod = importlib.import_module(__name__ + ".hip_solver")
        if name == "hip_solver":
            return mod
        if name == "partition":
            return importlib.import_module(__name__ + ".partition")
        return getattr(mod, name)
    raise AttributeError(name)
