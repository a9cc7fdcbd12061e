#!/usr/bin/env python3
"""A Venice-shaped problem with 1 % prior rows (trailing row blocks on cameras only, no point cell: SchurEliminator::NoEBlockRowsUpdate's
rows) against the pure BAL problem: the fused tiles take the conforming rows, small generic kernels add the remainder's sums.  One JSON
line: device time of a solve (set-up + preconditioner + CG + back-substitution, HIP events; uploads excluded) for both, both solvers.
usage: leftover_rows_times.py [workload] [fraction]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
hs, P = pkg.hip_solver, pkg.problems
wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778"
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
base = P.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
n_extra = int(round(frac * base.bs.num_row_blocks))
withrows = P.add_camera_rows(base, n_extra, seed=7, row_size=9, pair_fraction=0.0)
out = {"workload": wl, "observations": int(base.bs.num_row_blocks), "leftover_rows": n_extra}
for name, prob in (("pure", base), ("with_leftover_rows", withrows)):
    for solver, typ, pre in (("iterative_schur", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI), ("cgnr", hs.CGNR, hs.JACOBI)):
        s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                      elimination_groups=[prob.num_eliminate_blocks]))
        s.set_structure(prob.bs)
        s.set_phase_timing(True)   # (last_timing below: the phase events are opt-in)
        assert s.info().kernel_path == hs.PATH_BAL
        best = None
        for _ in range(5):
            x, summ = s.solve(prob.values, prob.b, hs.PerSolveOptions(D=prob.D, q_tolerance=0.1, r_tolerance=-1.0))
            t = s.last_timing()
            ms = t.setup_ms + t.preconditioner_ms + t.cg_ms + t.back_substitute_ms
            best = ms if best is None else min(best, ms)
        out[f"{name}:{solver}"] = {"solve_device_ms": round(best, 4), "cg_iterations": summ.num_iterations}
        s.close()
for solver in ("iterative_schur", "cgnr"):
    out[f"ratio:{solver}"] = round(out[f"with_leftover_rows:{solver}"]["solve_device_ms"] / out[f"pure:{solver}"]["solve_device_ms"], 4)
print(json.dumps(out))
