#!/usr/bin/env python3
"""CPU-only: where the oracle's LM step spends its time, by OpenMP thread count (seconds, best of 2).
usage: cpu_oracle_phases.py [workload] [threads ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
oracle = entry.load_oracle()
wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778"
threads = [int(a) for a in sys.argv[2:]] or [1, 16, 32]
prob = pkg.problems.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
m = oracle.Matrix(prob.bs, prob.num_eliminate_blocks)
m_all = oracle.Matrix(prob.bs, 0)
v, b = prob.values, prob.b
rng = np.random.default_rng(0)
xf, xe, xr = rng.standard_normal(m.num_cols_f), rng.standard_normal(m.num_cols_e), rng.standard_normal(m.num_rows)
D = np.sqrt(np.clip(m_all.squared_column_norm(v), 1e-6, 1e32) / 1e4)


def best(f, *a):
    t = []
    for _ in range(2):
        t0 = time.perf_counter(); f(*a); t.append(time.perf_counter() - t0)
    return round(min(t), 4)


for th in threads:
    oracle.set_num_threads(th)
    isc = oracle.ImplicitSchurComplement(m)
    out = {"threads": th,
           "colnorm": best(m_all.squared_column_norm, v),
           "isc_init": best(isc.init, v, D, b),
           "isc_sx": best(isc.sx, xf),
           "schur_jacobi": best(m.schur_jacobi, v, D),
           "back_substitute": best(isc.back_substitute, xf),
           "model_cost_Jx": best(m_all.right_multiply, v, np.concatenate([xe, xf])),
           "F x": best(m.right_multiply_f, v, xf), "E^T r": best(m.left_multiply_e, v, xr), "E x": best(m.right_multiply_e, v, xe),
           "F^T r": best(m.left_multiply_f, v, xr), "blockdiag E^T E": best(m.block_diagonal_ete, v),
           "whole solve": best(lambda: m.iterative_schur_solve(v, b, D, preconditioner=2, min_it=0, max_it=500, q_tol=0.1, r_tol=-1.0))}
    print(json.dumps(out), flush=True)
