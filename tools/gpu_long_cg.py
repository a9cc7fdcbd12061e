#!/usr/bin/env python3
"""Times fixed-length CG solves (min = max = K iterations) on a cached Venice-shaped problem.
usage: gpu_long_cg.py [K]; honours CERES_HIP_CG_FUSE."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
hs = pkg.hip_solver
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prob = pkg.problems.synthetic_bal("venice1778", layout="schur", seed=38401, skew=0.6)
dev = torch.device("cuda", 0)
tv, tb, tD = (torch.from_numpy(a).to(dev) for a in (prob.values, prob.b, prob.D))
tx = torch.empty(prob.num_cols, dtype=torch.float64, device=dev)
out = {"K": K, "cg_fused": os.environ.get("CERES_HIP_CG_FUSED", "1")}
for name, typ, pre in (("cgnr", hs.CGNR, hs.JACOBI), ("schur", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI)):
    s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=K, max_num_iterations=K,
                                                  elimination_groups=[prob.num_eliminate_blocks]))
    s.set_structure(prob.bs)
    s.set_phase_timing(True)   # (last_timing below: the phase events are opt-in)
    for _ in range(2):
        s.solve_device(tv.data_ptr(), tb.data_ptr(), tD.data_ptr(), tx.data_ptr(), -1.0, -1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        summ = s.solve_device(tv.data_ptr(), tb.data_ptr(), tD.data_ptr(), tx.data_ptr(), -1.0, -1.0)
    torch.cuda.synchronize()
    out[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    out[name + "_cg_ms"] = round(s.last_timing().cg_ms, 3)
    out[name + "_its"] = summ.num_iterations
    s.close()
print(json.dumps(out))
