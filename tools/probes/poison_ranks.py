#!/usr/bin/env python3
"""How every rank of a sharded call ends when ONE rank's shard cannot be solved (tests/test_gpu_multirank.py: the poison scenario),
over worlds, poisoned ranks, kernel paths and regimes.  One JSON line per case."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import __graft_entry__ as entry
pkg = entry.load_package()
hip = pkg.hip_solver
from test_gpu_multirank import run_ranks

if __name__ == "__main__":
    for world, pr, what, generic, nc, npts, no in [(2, 0, "nan", False, 20, 1500, 7000), (3, 1, "singular_point", False, 20, 1500, 7000),
                                                   (8, 5, "nan", False, 20, 1500, 7000), (2, 1, "nan", True, 20, 1500, 7000),
                                                   (2, 0, "singular_point", True, 20, 1500, 7000), (4, 2, "nan", False, 2600, 20000, 90000),
                                                   (4, 0, "singular_point", False, 2600, 20000, 90000), (8, 7, "singular_point", False, 20, 1500, 7000)]:
        kw = dict(kind="bal", seed=31, nc=nc, np=npts, no=no, skew=0.5, solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)],
                  poison=(pr, what), p2p_timeout=5, force_generic=generic)
        try:
            res = run_ranks([("poison", kw)], world, timeout=120)
            for solver in kw["solvers"]:
                recs = [res[r][("poison",) + solver] for r in range(world)]
                print(json.dumps(dict(world=world, poisoned=pr, what=what, generic=generic, cameras=nc, solver=solver,
                                      bad_solve=[str(rec["bad_solve"][:2]) + rec["bad_solve"][2][:60] for rec in recs],
                                      bad_step=[str(rec["bad_step"][:2])[:160] + (str(rec["bad_step"][2])[:60] if len(rec["bad_step"]) > 2 else "") for rec in recs],
                                      good=[rec["good_solve"][1:3] for rec in recs])), flush=True)
        except Exception as ex:
            print(json.dumps(dict(world=world, poisoned=pr, what=what, generic=generic, cameras=nc, error=repr(ex)[:1500])), flush=True)
