import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
hs, P = pkg.hip_solver, pkg.problems
for nc, npts, nobs in ((228, 50000, 250000), (456, 100000, 500000)):
    p = P.synthetic_bal(None, layout="schur", seed=3, skew=0.5, num_cameras=nc, num_points=npts, num_observations=nobs)
    o = hs.LinearSolverOptions(type=hs.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1)
    s = hs.HipLinearSolver(o)
    t = time.time(); s.set_structure(p.bs); ts = time.time() - t; s.set_phase_timing(True)
    for _ in range(2):
        x, summ = s.solve(p.values, p.b, hs.PerSolveOptions(D=p.D))
        tm = s.last_timing()
    print(json.dumps({"cameras": nc, "n": 9 * nc, "observations": nobs, "set_structure_s": round(ts, 2), "termination": summ.termination_type,
                      "init_ms": round(tm.setup_ms, 3), "eliminate_ms": round(tm.preconditioner_ms, 3), "factor_and_solve_ms": round(tm.cg_ms, 3), "backsub_ms": round(tm.back_substitute_ms, 3)}))
    s.close()
