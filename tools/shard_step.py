#!/usr/bin/env python3
"""One rank's shard of a workload alone on the device (bench.shard_ceiling), for rocprofv3 --kernel-trace timelines.
usage: shard_step.py <workload> <ranks (1 = the whole problem, unsharded)> [solver] [steps] [cg iterations]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import torch  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
solver = sys.argv[3] if len(sys.argv) > 3 else "iterative_schur"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
k = int(sys.argv[5]) if len(sys.argv) > 5 else 2
pkg = bench.entry.load_package()
hs = pkg.hip_solver
hs.load_library()
dev = torch.device("cuda", 0)
prob = pkg.problems.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
if n <= 1:
    s = bench.make_solver(hs, prob.bs, prob.num_eliminate_blocks, solver, 0)
    tv, tb = torch.from_numpy(prob.values).to(dev), torch.from_numpy(prob.b).to(dev)
    tx = torch.empty(prob.bs.num_cols, dtype=torch.float64, device=dev)
    el, its, _ = bench.timed_steps(s, (tv, tb, None, tx), steps, 3, torch.cuda.synchronize, "lm_step", 0.1)
    print(json.dumps({"workload": wl, "ranks": 1, "ms_per_step": round(1e3 * el / steps, 4), "cg_iterations": its[-1]}))
else:
    r = bench.shard_ceiling(pkg, hs, prob, solver, 0, 1.0, k, 0.1, worlds=(n,), steps=steps, dev=dev)
    print(json.dumps(r["cases"][0]))
