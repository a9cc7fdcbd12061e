#!/usr/bin/env python3
"""Host-side cost of ceres_hip_set_structure (analysis, tile plan, camera-major lists, uploads): once per block structure.
usage: set_structure_times.py [workload ...]     CERES_HIP_PLAN_TIMING=1 prints the plan's phases to stderr"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import __graft_entry__ as entry
pkg = entry.load_package()
hs = pkg.hip_solver
for wl in (sys.argv[1:] or ["ladybug1723", "venice1778"]):
    many = wl in ("synthetic1M", "synthetic10M")
    prob = pkg.problems.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6, with_values=not many)
    for typ, pre, name in ((hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI, "iterative_schur"), (hs.CGNR, hs.JACOBI, "cgnr")):
        ts = []
        for rep in range(2):
            s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, max_num_iterations=500, elimination_groups=[prob.num_eliminate_blocks]))
            t0 = time.perf_counter()
            s.set_structure(prob.bs)
            ts.append(time.perf_counter() - t0)
            s.close()
        print(json.dumps({"workload": wl, "solver": name, "set_structure_s": [round(t, 3) for t in ts], "observations": int(prob.bs.num_row_blocks)}), flush=True)
