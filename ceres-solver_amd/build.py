"""Builds csrc/ into csrc/libceres_hip.so with hipcc for gfx950 (in-tree, so that the
built library travels to the GPU box with the repository snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libceres_hip.so")
# the fused kernels are compiled once per SHAPE (camera width, shared strip): kernels_bal.inc through kernels_bal_shape_*.hip (common.h)
# (round 6: every camera width 2 .. 10 — the widths the reference reaches through its (2,3,d) / (2,4,d) / (2,2,d) dynamic specialisations)
BAL_SHAPES = [(2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (9, 0), (10, 0), (6, 4), (6, 8), (9, 4), (9, 8)]
# point blocks other than 3 wide (round 5): (ne, nf), no strip
BAL_SHAPES_E = [(2, 2), (2, 3), (2, 4), (2, 6), (2, 9), (4, 2), (4, 3), (4, 4), (4, 5), (4, 6), (4, 7), (4, 8), (4, 9), (4, 10)]
# row blocks other than 2 high (round 5): (nr, ne, nf), no strip — the reference's (3,3,3), (4,4,2), (4,4,3), (4,4,4)
BAL_SHAPES_R = [(3, 3, 3), (4, 4, 2), (4, 4, 3), (4, 4, 4)]
SOURCES = (["plan.cc", "kernels_generic.hip", "kernels_cg.hip", "kernels_bal_common.hip"] + [f"kernels_bal_shape_f{nf}_s{ns}.hip" for nf, ns in BAL_SHAPES] +
           [f"kernels_bal_shape_e{ne}_f{nf}_s0.hip" for ne, nf in BAL_SHAPES_E] +
           [f"kernels_bal_shape_r{nr}_e{ne}_f{nf}_s0.hip" for nr, ne, nf in BAL_SHAPES_R] +
           ["kernels_schur.hip", "kernels_evaluator.hip", "solver.hip"])
HEADERS = ["common.h", "device.h", "p2p.h", "snavely.h", "bal_frontend.inc", "solver_comm.inc", "solver_stream.inc", "solver_ops.inc", "solver_debug.inc", "kernels_bal.inc", os.path.join("..", "..", "include", "ceres_hip.h")]
HOST_DRIVER_SRC = os.path.join(HERE, "host", "host_driver.cc")
HOST_DRIVER = os.path.join(HERE, "host", "host_driver")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    objs, cmds = [], []
    for src in srcs:
        obj = os.path.splitext(src)[0] + ".o"
        objs.append(obj)
        if force or _stale(obj, [src] + [os.path.join(CSRC, h) for h in HEADERS]):
            cmds.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                         "-x", "hip", "-c", src, "-o", obj])
    if cmds:  # one hipcc per translation unit, a few at a time (the per-shape units take ~25 s each)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=max(1, min(len(cmds), (os.cpu_count() or 2)))) as ex:
            list(ex.map(run, cmds))
    if force or _stale(OUT, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


def build_host_driver(force=False, verbose=False):
    """The C++ host-side mirror of ceres::internal::LinearSolver + a small driver (g++, links the C ABI)."""
    if not os.path.exists(HOST_DRIVER_SRC):
        return None
    deps = [HOST_DRIVER_SRC, os.path.join(HERE, "host", "hip_linear_solver.h"), os.path.join(HERE, "host", "flatten_block_structure.h"),
            os.path.join(HERE, "host", "hip_bal_problem.h"), OUT]
    if force or _stale(HOST_DRIVER, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(HERE, "..", "include"), "-I", os.path.join(HERE, "host"),
               HOST_DRIVER_SRC, "-o", HOST_DRIVER, "-pthread", "-L", CSRC, "-lceres_hip", "-Wl,-rpath," + CSRC,
               "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_DRIVER


def build_all(force=False, verbose=False):
    build_library(force, verbose)
    build_host_driver(force, verbose)
    return OUT


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
