"""Sharding by point on CPU: the host-side partition (ceres-solver_amd/partition.py)
and the sharded algorithm — what is summed over ranks and where — exercised with two gloo
processes whose local compute is the oracle.  The GPU path performs the same all-reduces with
RCCL inside the C ABI."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg


def test_partition_covers_everything_once(problems):
    from ceres_solver_amd import partition
    p = problems.synthetic_bal(None, num_cameras=12, num_points=500, num_observations=2300, seed=8, skew=0.7)
    for world in (1, 2, 3, 8):
        shards = [partition.shard_by_point(p.bs, p.num_eliminate_blocks, world, r) for r in range(world)]
        rows = np.concatenate([np.arange(*s.row_block_range) for s in shards])
        assert np.array_equal(rows, np.arange(p.bs.num_row_blocks))
        vals = np.concatenate([s.value_index for s in shards])
        assert np.array_equal(np.sort(vals), np.arange(p.bs.num_nonzeros))
        counts = [s.bs.num_row_blocks for s in shards]
        assert max(counts) - min(counts) <= 0.1 * max(counts) + 64  # balanced by observations
        for s in shards:
            assert s.bs.num_col_blocks == s.num_eliminate_blocks + 12
            # the local sub-matrix is the global one restricted to the shard's rows / columns
            A = p.bs.to_dense(p.values)[np.ix_(s.row_index, s.col_index)] if world <= 3 else None
            if A is not None:
                np.testing.assert_array_equal(s.bs.to_dense(s.local_values(p.values)), A)


def test_partition_gives_every_rank_a_point(problems):
    """A handful of points of very different track lengths: balancing by values alone left a rank without eliminated blocks (an
    ITERATIVE_SCHUR shard of no E blocks is refused, its peers then wait for it; tools/fuzz_multirank.py).  Fewer points than ranks
    cannot be sharded by point."""
    from ceres_solver_amd import partition
    p = problems.bal_from_tracks([36, 3, 3, 3, 3], 64, seed=1)
    for world in (2, 3, 4, 5):
        bounds = partition.split_points(p.bs, p.num_eliminate_blocks, world)
        assert bounds[0] == 0 and bounds[-1] == p.num_eliminate_blocks and (np.diff(bounds) >= 1).all(), bounds
        shards = [partition.shard_by_point(p.bs, p.num_eliminate_blocks, world, r) for r in range(world)]
        assert all(s.num_eliminate_blocks >= 1 for s in shards)
        assert np.array_equal(np.concatenate([np.arange(*s.row_block_range) for s in shards]), np.arange(p.bs.num_row_blocks))
    with pytest.raises(ValueError):
        partition.split_points(p.bs, p.num_eliminate_blocks, 6)


def test_partition_keeps_e_free_rows(problems):
    from ceres_solver_amd import partition
    p = problems.random_schur_problem(num_e_blocks=10, num_f_blocks=4, num_no_e_rows=3, seed=3)
    shards = [partition.shard_by_point(p.bs, p.num_eliminate_blocks, 2, r) for r in range(2)]
    assert len(shards[0].extra_row_blocks) == 0 and len(shards[1].extra_row_blocks) == 3
    total = sum(s.bs.num_row_blocks for s in shards)
    assert total == p.bs.num_row_blocks


MAX_IT = 5000
R_TOL = {"iterative_schur_solve": 1e-12, "cgnr_solve": 1e-10}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pk = entry.load_package()
    oracle = entry.load_oracle()
    from ceres_solver_amd import partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def allreduce(view):
            t = torch.from_numpy(view)
            dist.all_reduce(t)

        out = {}
        for name, prob in (("bal", pk.problems.synthetic_bal(None, num_cameras=9, num_points=300, num_observations=1400, seed=2)),
                           ("general", pk.problems.random_schur_problem(num_e_blocks=14, num_f_blocks=5, num_no_e_rows=2, seed=6))):
            sh = partition.shard_by_point(prob.bs, prob.num_eliminate_blocks, world, rank)
            m = oracle.Matrix(sh.bs, sh.num_eliminate_blocks)
            v, b, D = sh.local_values(prob.values), sh.local_rows(prob.b), sh.local_cols(prob.D)
            for solver, pre in (("iterative_schur_solve", 2), ("iterative_schur_solve", 1), ("cgnr_solve", 1), ("cgnr_solve", 0)):
                x, s = getattr(m, solver)(v, b, D, preconditioner=pre, max_it=MAX_IT, q_tol=-1.0, r_tol=R_TOL[solver], allreduce=allreduce)
                out[(name, solver, pre)] = (sh.col_index, int(sh.bs.col_block_size[: sh.num_eliminate_blocks].sum()), x,
                                            s.termination_type, s.num_iterations)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_solvers_match_single_rank(oracle, problems):
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for name, prob in (("bal", problems.synthetic_bal(None, num_cameras=9, num_points=300, num_observations=1400, seed=2)),
                       ("general", problems.random_schur_problem(num_e_blocks=14, num_f_blocks=5, num_no_e_rows=2, seed=6))):
        m = oracle.Matrix(prob.bs, prob.num_eliminate_blocks)
        for solver, pre in (("iterative_schur_solve", 2), ("iterative_schur_solve", 1), ("cgnr_solve", 1), ("cgnr_solve", 0)):
            ref, sref = getattr(m, solver)(prob.values, prob.b, prob.D, preconditioner=pre, max_it=MAX_IT, q_tol=-1.0, r_tol=R_TOL[solver])
            x = np.full(prob.num_cols, np.nan)
            cams = []
            for r in range(world):
                col_index, n_e, xl, term, its = results[r][(name, solver, pre)]
                x[col_index[:n_e]] = xl[:n_e]
                cams.append(xl[n_e:])
                assert term == sref.termination_type == 0
                assert abs(its - sref.num_iterations) <= max(2, sref.num_iterations // 20)
            x[col_index[n_e:]] = cams[0]
            np.testing.assert_array_equal(cams[0], cams[1])  # replicated camera space stays bit-identical
            assert np.linalg.norm(x - ref) <= (1e-8 if solver.startswith('iter') else 1e-6) * np.linalg.norm(ref), (name, solver, pre)
