#!/usr/bin/env python3
"""Latency of the one-shot peer-to-peer all-reduce between two ranks that SHARE one GPU (the only multi-rank set-up a one-GPU
box offers; across xGMI the flag / payload hops are longer): microseconds per all-reduce by message size.  Sizes stay small
enough that both ranks' grids are co-resident on the shared GPU (one block per 2048 doubles; a rank's block c waits for its
peer's block c, which on a SHARED device may not be scheduled while the first rank's blocks fill every CU).
usage: p2p_latency.py  -> one JSON line"""
import json, multiprocessing as mp, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, conn):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("CERES_HIP_P2P_TIMEOUT", "5")
    import torch  # noqa
    import __graft_entry__ as entry
    hs = entry.load_package().hip_solver
    hs.load_library()

    def exchange(mine):
        conn.send(mine)
        return conn.recv()
    o = hs.LinearSolverOptions(type=hs.ITERATIVE_SCHUR, preconditioner_type=hs.SCHUR_JACOBI, max_num_iterations=10, elimination_groups=[1])
    s = hs.HipLinearSolver(o, rank=rank, world_size=world, p2p_exchange=exchange, p2p_max_elements=1 << 22)
    assert s.p2p_selftest(), s.p2p_error
    out = {}
    for n in (2, 16002, 144018, 450000):
        out[n] = round(s.allreduce_timing(n, 200), 2)
    conn.send(out)
    s.close()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    world = 2
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=rank_main, args=(r, world, pipes[r][1])) for r in range(world)]
    [p.start() for p in procs]
    handles = [pipes[r][0].recv() for r in range(world)]
    [pipes[r][0].send(handles) for r in range(world)]
    res = [pipes[r][0].recv() for r in range(world)]
    [p.join(60) for p in procs]
    print(json.dumps({"what": "us per one-shot peer-to-peer all-reduce (fp64 sum), 2 ranks sharing one MI355X, 200 back-to-back on the stream",
                      "us_by_doubles": {str(k): max(r[k] for r in res) for k in res[0]}}))
