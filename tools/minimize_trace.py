#!/usr/bin/env python3
"""ceres_hip_bal_minimize on the Venice-shaped synthetic scene, alone in a process: the thing to wrap in
`rocprofv3 --kernel-trace --stats` when the question is where a trust-region iteration's time goes."""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ceres-solver_amd")
hs = pkg.hip_solver

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="venice1778")
ap.add_argument("--solver", default="iterative_schur")
ap.add_argument("--iterations", type=int, default=3)
ap.add_argument("--repeat", type=int, default=3)
a = ap.parse_args()
nc, npt, cam_i, pt_i, obs, par = pkg.problems.bal_scene(a.workload, seed=38401)
typ, pre = (hs.CGNR, hs.JACOBI) if a.solver == "cgnr" else (hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI)
bp = hs.BalProblem(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500), nc, npt, cam_i, pt_i, obs)
x0 = bp.state_from_bal(par)
bp.minimize(x0, max_num_iterations=1)
for r in range(a.repeat):
    t0 = time.perf_counter()
    _, S = bp.minimize(x0, max_num_iterations=a.iterations)
    wall = time.perf_counter() - t0
    nit = S.num_successful_steps + S.num_unsuccessful_steps
    print(json.dumps({"repeat": r, "lm_iterations": nit, "linear_solves": S.num_linear_solves, "total_ms": round(1e3 * S.total_seconds, 3),
                      "wall_ms": round(1e3 * wall, 3), "ms_per_lm_iteration": round(1e3 * S.total_seconds / max(nit, 1), 3),
                      "linear_solver_ms": round(1e3 * S.linear_solver_seconds, 3), "evaluation_ms": round(1e3 * S.evaluation_seconds, 3),
                      "cg_iterations": [S.iterations[i].linear_solver_iterations for i in range(1, S.num_iterations_logged)],
                      "final_cost": S.final_cost, "termination": S.message.decode(errors="replace")}), flush=True)
