#!/usr/bin/env python3
"""CPU-only: seconds per Venice-shaped ITERATIVE_SCHUR LM step of the oracle by OpenMP thread count, under whatever OMP_* placement
environment the caller set (OMP_PROC_BIND / OMP_PLACES must be set before libgomp starts: one process per setting).
usage: cpu_oracle_probe.py [workload] [threads ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
oracle = entry.load_oracle()
wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778"
threads = [int(a) for a in sys.argv[2:]] or [16, 32, 64]
prob = pkg.problems.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
m = oracle.Matrix(prob.bs, prob.num_eliminate_blocks)
m_all = oracle.Matrix(prob.bs, 0)
out = {"workload": wl, "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"), "cpus": os.cpu_count(), "seconds_per_step": {}}
for th in threads:
    oracle.set_num_threads(th)
    best = 1e9
    for _ in range(2):
        t = time.perf_counter()
        diag = np.clip(m_all.squared_column_norm(prob.values), 1e-6, 1e32)
        D = np.sqrt(diag / 1e4)
        x, summ = m.iterative_schur_solve(prob.values, prob.b, D, preconditioner=2, min_it=0, max_it=500, q_tol=0.1, r_tol=-1.0)
        model = m_all.right_multiply(prob.values, -x)
        _ = -model @ (prob.b + model / 2.0)
        best = min(best, time.perf_counter() - t)
    out["seconds_per_step"][th] = round(best, 3)
print(json.dumps(out))
