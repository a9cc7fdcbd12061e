#!/usr/bin/env python3
"""S.x / JtJx on REAL visibility replicated to size: the libmv problems the reference ships (tests/golden/libmv_problems.npz; a few dozen
tracks through hundreds of consecutive frames), `copies` disjoint replicas side by side, N(0,1) values.  One JSON line per case with the
operator times, the fraction of the 8 TB/s HBM peak on the algorithmic bytes and how many observations the tile pass sums in LDS.
usage: real_graph_times.py [problem:copies ...]   default 2:4 (1760 cameras: accumulators in LDS) 2:120 (52 800 cameras: hybrid) 3:100"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
hs, P = pkg.hip_solver, pkg.problems
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(2, 4), (2, 120), (3, 100), (0, 1778)]
for problem, copies in cases:
    if problem == 0:
        # heavy-tailed tracks, as internet photo collections have them: 2 (1 + Pareto(1.6)) observations per point capped at 1000 (mean 5,
        # an eighth of the observations in tracks of more than 64), random cameras: "0:<cameras>", 1 M points
        rng = np.random.default_rng(38401)
        lengths = np.minimum(np.floor(2.0 * (1.0 + rng.pareto(1.6, 1000000))), min(1000, copies)).astype(np.int64)
        prob = P.bal_from_tracks(lengths, copies, seed=38401)
    else:
        prob = P.libmv_bal(problem, copies)
    n_p = prob.num_eliminate_blocks
    n_c = prob.bs.num_col_blocks - n_p
    n_o = prob.bs.num_row_blocks
    B_jtjx = n_o * 200 + (3 * n_p + 9 * n_c) * 32
    B_sx = n_o * 200 + n_p * 72 + n_c * 288
    out = {"graph": f"libmv problem_0{problem} x {copies}" if problem else f"power-law tracks on {copies} cameras", "cameras": n_c, "points": n_p, "observations": n_o}
    for solver, typ, pre, op, nbytes in (("cgnr", hs.CGNR, hs.JACOBI, hs.TIMED_JTJX, B_jtjx), ("schur", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI, hs.TIMED_SX, B_sx)):
        s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                      elimination_groups=[n_p]))
        s.set_structure(prob.bs)
        s.set_phase_timing(True)   # (last_timing below: the phase events are opt-in)
        info = s.info()
        s.load(prob.values, prob.b, prob.D)
        ms = min(s.time_op(op, 30) for _ in range(3))
        name = "jtjx" if solver == "cgnr" else "sx"
        out[name + "_ms"] = round(ms, 4)
        out[name + "_frac"] = round(nbytes / ms / 1e6 / 8000, 4)
        out["tiles"] = int(info.num_tiles)
        out["padding"] = round(1.0 - n_o / (64.0 * info.num_tiles), 4)
        out["accumulators_in_lds"] = int(info.camera_accum_in_lds)
        out["hybrid"] = int(info.camera_accum_hybrid)
        out["observations_in_lds"] = round(info.num_observations_in_lds / float(n_o), 4)
        x, summ = s.solve(prob.values, prob.b, hs.PerSolveOptions(D=prob.D, q_tolerance=0.1, r_tolerance=-1.0))
        t = s.last_timing()
        out[solver + "_solve"] = {"its": summ.num_iterations, "setup_ms": round(t.setup_ms, 3), "precond_ms": round(t.preconditioner_ms, 3), "cg_ms": round(t.cg_ms, 3),
                                  "backsub_ms": round(t.back_substitute_ms, 3)}
        s.close()
    print(json.dumps(out), flush=True)
