#!/usr/bin/env python3
"""S.x / JtJx / the step's set-up passes on the fused path for shapes other than <2,3,9> (GPU box helper): one JSON line per case.
usage: shape_times.py [ladybug|venice|libmv|all]
Algorithmic bytes as SURVEY.md §8(d) counts them, with the slot's own size: per observation (6 + 2 nf + 2 ns) doubles of Jacobian + 8 bytes
of indices; per point 9 doubles ((E'E)^-1) for S.x or 4 x 3 for JtJx; per camera-space scalar 4 doubles."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (device runtime first)
import __graft_entry__ as entry
pkg = entry.load_package()
hs, P = pkg.hip_solver, pkg.problems
what = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
SHAPES = {"ladybug": (1723, 156502, 678718), "venice": (1778, 993923, 5001946)}


def run(tag, prob, nf, ns, extra=None):
    n_o = prob.bs.num_row_blocks
    n_p = prob.num_eliminate_blocks
    n_fs = int(prob.bs.col_block_size[n_p:].sum())
    slot = (6 + 2 * nf + 2 * ns) * 8 + 8
    B = {"sx": n_o * slot + n_p * 72 + n_fs * 32, "jtjx": n_o * slot + (3 * n_p + n_fs) * 32}
    out = {"case": tag, "camera_width": nf, "strip": ns, "observations": n_o, "points": n_p, "camera_side_scalars": n_fs, "bytes_per_slot": slot}
    for solver, typ, pre, ops in (("schur", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI, [("sx", hs.TIMED_SX), ("schur_init", hs.TIMED_SCHUR_INIT), ("schur_jacobi", hs.TIMED_SCHUR_JACOBI),
                                                                                     ("back_substitute", hs.TIMED_BACK_SUBSTITUTE)]),
                                  ("cgnr", hs.CGNR, hs.JACOBI, [("jtjx", hs.TIMED_JTJX), ("cgnr_setup", hs.TIMED_CGNR_SETUP)])):
        s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                      elimination_groups=[n_p]))
        s.set_structure(prob.bs)
        s.set_phase_timing(True)   # (last_timing below: the phase events are opt-in)
        info = s.info()
        out["kernel_path"] = "fused" if info.kernel_path == hs.PATH_BAL else "generic"
        out["accumulators_in_lds"] = int(info.camera_accum_in_lds)
        s.load(prob.values, prob.b, prob.D)
        for name, op in ops:
            if info.kernel_path != hs.PATH_BAL and name in ("cgnr_setup",):
                continue
            ms = min(s.time_op(op, 30) for _ in range(3))
            out[name + "_ms"] = round(ms, 5)
            if name in B:
                out[name + "_GBs"] = round(B[name] / ms / 1e6, 1)
                out[name + "_frac"] = round(B[name] / ms / 1e6 / 8000, 4)
        x, summ = s.solve(prob.values, prob.b, hs.PerSolveOptions(D=prob.D, q_tolerance=0.1, r_tolerance=-1.0))
        out[solver + "_solve_ms"] = round(s.last_timing().total_ms - s.last_timing().upload_ms - s.last_timing().download_ms, 4)
        out[solver + "_its"] = summ.num_iterations
        s.close()
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


for shape in ("ladybug", "venice"):
    if what not in (shape, "all"):
        continue
    n_c, n_pt, n_ob = SHAPES[shape]
    for nf in ((9, 10, 6, 3, 4, 8) if shape == "ladybug" else (9, 10, 6)):
        p = P.synthetic_structured(n_c, n_pt, n_ob, camera_width=nf, seed=38401, skew=0.6)
        run(f"{shape}-shaped <2,3,{nf}>", p, nf, 0)
    p = P.synthetic_structured(n_c, n_pt, n_ob, camera_width=6, shared_widths=(8,), locked_cameras=(0,), seed=38401, skew=0.6)
    run(f"{shape}-shaped libmv structure <2, 8|6, 3>", p, 6, 8)
if what in ("libmv", "all"):
    for copies in (1, 120):
        p = P.libmv_structured(2, copies)
        run(f"libmv problem_02 x {copies}: intrinsics 8 + pose 6 + point 3, first camera constant", p, 6, 8)
        p = P.libmv_bal(2, copies)
        run(f"libmv problem_02 x {copies} visibility as <2,3,9> (round-3 form)", p, 9, 0)
