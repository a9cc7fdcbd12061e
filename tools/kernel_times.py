#!/usr/bin/env python3
"""Times the dominant operators on a cached Venice-shaped problem (GPU box helper).
usage: kernel_times.py [workload] [--force-generic] ; honours CERES_HIP_BAL_BLOCK.  --force-generic: the thread-per-scalar generic kernels
(any block sizes; one reference operator per launch) on the same problem, so that their cost is a number."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
hs = pkg.hip_solver
force_generic = "--force-generic" in sys.argv
operators_only = "--operators-only" in sys.argv   # tools/pmc_live.py: S.x and JtJx only, a dozen applications each, no solve
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args_[0] if args_ else "venice1778"
cache = f"/tmp/{wl}.npz"
P = pkg.problems
many = wl in ("synthetic1M", "synthetic10M", "banded50k") and os.environ.get("KERNEL_TIMES_HOST_VALUES", "0") != "1"
dev_vals = None
if wl == "banded50k":   # bench.py's extra.banded50k: synthetic1M's block counts, every point seen by consecutive cameras
    P.BAL_SHAPES["banded50k"] = P.BAL_SHAPES["synthetic1M"]
if many:   # values generated in HBM, like bench.py's many-camera workloads (5.76 GB of Jacobian for synthetic10M)
    if wl == "banded50k":
        bc, bp, bo = P.BAL_SHAPES[wl]
        prob = P.banded_bal(None, seed=38401, num_cameras=bc, num_points=bp, num_observations=bo, with_values=False)
    else:
        prob = P.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6, with_values=False)
    g = torch.Generator(device="cuda")
    g.manual_seed(38401)
    nrb = prob.bs.num_row_blocks
    dev_vals = (torch.randn(24 * nrb, dtype=torch.float64, device="cuda", generator=g), torch.randn(2 * nrb, dtype=torch.float64, device="cuda", generator=g),
                torch.rand(prob.bs.num_cols, dtype=torch.float64, device="cuda", generator=g) * 0.1 + 0.05)
elif os.path.exists(cache):
    z = np.load(cache)
    bs = pkg.BlockStructure(*(z[k] for k in ("rsz", "rpos", "csz", "cpos", "rptr", "ccol", "cval")))
    prob = P.LinearProblem(bs, z["values"], z["b"], z["D"], int(z["nelim"]))
else:
    prob = P.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
    b = prob.bs
    np.savez(cache, rsz=b.row_block_size, rpos=b.row_block_pos, csz=b.col_block_size, cpos=b.col_block_pos, rptr=b.row_cell_ptr,
             ccol=b.cell_col_block, cval=b.cell_value_pos, values=prob.values, b=prob.b, D=prob.D, nelim=prob.num_eliminate_blocks)
n_c, n_p, n_o = P.BAL_SHAPES[wl]
B_jtjx = n_o * 200 + (3 * n_p + 9 * n_c) * 32
B_sx = n_o * 200 + n_p * 72 + n_c * 288
out = {"block": os.environ.get("CERES_HIP_BAL_BLOCK", "default"), "workload": wl, "kernel_path": "generic" if force_generic else "fused<2,3,9>"}
skew = 0.6
for solver, typ, pre, ops in (("cgnr", hs.CGNR, hs.JACOBI, [("jtjx", hs.TIMED_JTJX, B_jtjx), ("block_jacobi", hs.TIMED_BLOCK_JACOBI, None), ("cgnr_setup", hs.TIMED_CGNR_SETUP, None)]),
                              ("schur", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI, [("sx", hs.TIMED_SX, B_sx), ("schur_init", hs.TIMED_SCHUR_INIT, None),
                                                                                  ("schur_jacobi", hs.TIMED_SCHUR_JACOBI, None), ("back_substitute", hs.TIMED_BACK_SUBSTITUTE, None),
                                                                                  ("pack", hs.TIMED_PACK, None), ("read_stream", hs.TIMED_READ_STREAM, "tiles")])):
    storage = int(os.environ.get("STORAGE", "0"))
    out["storage"] = "fp32" if storage else "fp64"
    if storage:
        ops = [o for o in ops if o[0] not in ("read_stream",)]
    if force_generic:
        ops = [o for o in ops if o[0] in ("jtjx", "block_jacobi", "sx", "schur_init", "schur_jacobi", "back_substitute")]
    if operators_only:
        ops = [o for o in ops if o[0] in ("jtjx", "sx")]
    s = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                  elimination_groups=[prob.num_eliminate_blocks], jacobian_storage=storage, force_generic_path=force_generic))
    s.set_structure(prob.bs)
    s.set_phase_timing(True)   # (last_timing below: the phase events are opt-in)
    if dev_vals is not None:
        s.load_device(*(t.data_ptr() for t in dev_vals))
    else:
        s.load(prob.values, prob.b, prob.D)
    for name, op, nbytes in ops:
        if nbytes == "tiles":
            nbytes = int(s.info().num_tiles) * 12288
        ms = min(s.time_op(op, 5 if force_generic else (12 if operators_only else 30)) for _ in range(1 if operators_only else (2 if force_generic else 3)))
        out[name + "_ms"] = round(ms, 4)
        if nbytes:
            out[name + "_GBs"] = round(nbytes / ms / 1e6, 1)
            out[name + "_frac"] = round(nbytes / ms / 1e6 / 8000, 4)
    if dev_vals is not None or operators_only:
        s.close()
        continue
    x, summ = s.solve(prob.values, prob.b, hs.PerSolveOptions(D=prob.D, q_tolerance=0.1, r_tolerance=-1.0))
    t = s.last_timing()
    out[solver + "_solve"] = {"its": summ.num_iterations, "total_ms": round(t.total_ms, 3), "upload_ms": round(t.upload_ms, 3), "pack_ms": round(t.pack_ms, 3),
                              "setup_ms": round(t.setup_ms, 3), "precond_ms": round(t.preconditioner_ms, 3), "cg_ms": round(t.cg_ms, 3),
                              "backsub_ms": round(t.back_substitute_ms, 3), "download_ms": round(t.download_ms, 3)}
    s.close()
print(json.dumps(out))
