#!/usr/bin/env python3
"""One rank's shard of a workload alone on the device (bench.shard_ceiling), also for rocprofv3 --kernel-trace timelines.
usage: shard_step.py <workload> <ranks: 1 = the whole problem, or a list like 2,4,8 (T_1 is then measured first)> [solver] [steps] [cg iterations]
(the LAST step of a run is bench.phase_timing's: it carries the phase events — read the step before it in a timeline)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import torch  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778"
ranks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]
solver = sys.argv[3] if len(sys.argv) > 3 else "iterative_schur"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
k = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pkg = bench.entry.load_package()
hs = pkg.hip_solver
hs.load_library()
dev = torch.device("cuda", 0)
many = wl in ("synthetic1M", "synthetic10M")
t1 = None
prob = None if many else pkg.problems.synthetic_bal(wl, layout="schur", seed=38401, skew=0.6)
if 1 in ranks or len(ranks) > 1 or k == 0:
    if many:
        r1 = bench.shard_ceiling(pkg, hs, None, solver, 0, 1.0, k or 2, 0.1, worlds=(1,), steps=steps, dev=dev, many_cameras=wl)
        t1, k1 = r1["cases"][0]["ms_per_step"], r1["cases"][0]["cg_iterations"]
    else:
        s = bench.make_solver(hs, prob.bs, prob.num_eliminate_blocks, solver, 0)
        tv, tb = torch.from_numpy(prob.values).to(dev), torch.from_numpy(prob.b).to(dev)
        tx = torch.empty(prob.bs.num_cols, dtype=torch.float64, device=dev)
        el, its, _ = bench.timed_steps(s, (tv, tb, None, tx), steps, 3, torch.cuda.synchronize, "lm_step", 0.1)
        t1, k1 = 1e3 * el / steps, int(its[-1])
        s.close()
        del tv, tb, tx
    k = k or k1
    print(json.dumps({"workload": wl, "solver": solver, "ranks": 1, "ms_per_step": round(t1, 4), "cg_iterations": k1}), flush=True)
rest = tuple(n for n in ranks if n > 1)
if rest:
    r = bench.shard_ceiling(pkg, hs, prob, solver, 0, t1 or 1.0, k, 0.1, worlds=rest, steps=steps, dev=dev, many_cameras=wl if many else None)
    for c in r["cases"]:
        c.update({"workload": wl, "solver": solver})
        if t1 is None:
            c.pop("efficiency_ceiling", None)
        print(json.dumps(c), flush=True)
