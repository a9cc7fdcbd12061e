#!/usr/bin/env python3
"""DENSE_SCHUR's factorisation (csrc/kernels_schur.hip) at n = 2048 / 4096 / 8192 on random SPD matrices: time, flop rate (n^3 / 3) against
the fp64 MFMA peak (78.6 TFLOP/s: AMD's MI355X datasheet, FP64 matrix = FP64 vector; MI355X_MICROARCH.md lists no f64 row), and the
error of the solve against the known solution.  usage: dense_cholesky_times.py [n ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
hs = pkg.hip_solver
PEAK = 78.6
# reduced camera systems are 9 x cameras wide: 228 / 456 / 910 cameras (a power-of-two n puts every row of a panel column in the same
# HBM channel: n = 4096 runs 2.6x slower than n = 4104)
sizes = [int(a) for a in sys.argv[1:]] or [2052, 4104, 8190]
s = hs.HipLinearSolver(hs.LinearSolverOptions(type=hs.DENSE_SCHUR, elimination_groups=[1], max_num_iterations=1))
for n in sizes:
    rng = np.random.default_rng(n)
    # SPD with a known solution: diagonally dominant random symmetric matrix (cheap to build at n = 8192)
    B = rng.standard_normal((n, n)) * 0.5
    A = (B + B.T) / 2 + n * 0.6 * np.eye(n)
    xt = rng.standard_normal(n)
    b = A @ xt
    x, ms, failed = s.dense_cholesky_solve(np.triu(A), b, repeats=5 if n <= 4096 else 3)
    err = float(np.linalg.norm(x - xt) / np.linalg.norm(xt))
    tf = n ** 3 / 3.0 / ms / 1e9
    print(json.dumps({"n": n, "factor_ms": round(ms, 4), "TFLOPs": round(tf, 2), "frac_of_fp64_mfma_peak": round(tf / PEAK, 4), "peak_TFLOPs": PEAK,
                      "failed": failed, "rel_err_of_solve": err}), flush=True)
s.close()
