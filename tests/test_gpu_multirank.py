"""SURVEY.md §8(e) with PRODUCT code on real ranks: two processes, one rank each, sharded by point
(ceres-solver_amd/partition.py), camera-space sums combined by the library's one-shot peer-to-peer all-reduce over
hipIpc-mapped buffers.  The GPU box has one MI355X, so both ranks run on device 0 — the same code path (hipIpc
mapping, system-scope flags, rank-ordered sums) that runs over xGMI between the GPUs of a node.  Rung (5) of the
parity ladder: multi-GPU vs 1-GPU / oracle, same tolerances (sums are re-associated again)."""
import multiprocessing as mp

import numpy as np
import pytest

from test_gpu_operators import rel

pytestmark = pytest.mark.gpu

WORLD = 2


def run_ranks(scenario, timeout=420):
    from multirank_worker import run_rank
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(WORLD)]
    procs = [ctx.Process(target=run_rank, args=(r, WORLD, pipes[r][1], 0, scenario)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [None] * WORLD
    try:
        while any(r is None for r in results):
            msgs = []
            for r in range(WORLD):
                if results[r] is not None:
                    continue
                if not pipes[r][0].poll(timeout):
                    raise TimeoutError(f"rank {r} sent nothing for {timeout} s")
                msgs.append((r, pipes[r][0].recv()))
            kinds = {m[1][0] for m in msgs}
            for r, (kind, payload) in msgs:
                if kind == "error":
                    raise AssertionError(f"rank {r} failed:\n{payload}")
            assert len(kinds) == 1, f"ranks out of step: {kinds}"
            if kinds == {"exchange"}:
                handles = [m[1][1] for m in sorted(msgs)]
                for r, _ in msgs:
                    pipes[r][0].send(handles)
            else:
                for r, (_, payload) in msgs:
                    results[r] = payload
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()  # the exact process we started
    return results


def assemble(partition_mod, recs, num_cols, key):
    x = np.full(num_cols, np.nan)
    for rec in recs:
        x[rec["col_index"][: rec["n_e"]]] = rec[key][0][: rec["n_e"]]
    x[recs[0]["col_index"][recs[0]["n_e"]:]] = recs[0][key][0][recs[0]["n_e"]:]
    return x


def test_two_ranks_bal_both_solvers_against_the_oracle(hip, oracle, problems):
    kw = dict(kind="bal", seed=31, nc=37, np=6000, no=26000, skew=0.5,
              solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)])
    res = run_ranks([("bal", kw)])
    p = problems.synthetic_bal(None, layout="schur", seed=31, skew=0.5, num_cameras=37, num_points=6000, num_observations=26000)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    m0 = oracle.Matrix(p.bs, 0)
    nce = m.num_cols_e
    for solver_type, pre in kw["solvers"]:
        recs = [res[r][("bal", solver_type, pre)] for r in range(WORLD)]
        assert all(rec["path"] == hip.PATH_BAL and rec["world"] == WORLD for rec in recs) and [rec["rank"] for rec in recs] == [0, 1]
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        # converged solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=400, q_tol=-1.0, r_tol=1e-12)
        assert all(rec["converged"][1] == hip.SUCCESS for rec in recs)
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-8
        # the camera part is REPLICATED: identical bits on both ranks (the all-reduce sums in rank order everywhere)
        for key in ("converged", "lm_style"):
            a, b = recs[0][key][0][recs[0]["n_e"]:], recs[1][key][0][recs[1]["n_e"]:]
            assert np.array_equal(a, b), key
            assert recs[0][key][2] == recs[1][key][2]   # same iteration count on every rank
        # LM-style call: iteration count within 1 of the oracle, step 1e-9 when the counts coincide
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=400, q_tol=0.1, r_tol=-1.0)
        assert abs(recs[0]["lm_style"][2] - so.num_iterations) <= 1
        if recs[0]["lm_style"][2] == so.num_iterations:
            assert rel(assemble(None, recs, p.bs.num_cols, "lm_style"), xo) <= 1e-9
        # the LM step on the device: step = -(solve), model cost change summed over ranks and equal on both
        step = assemble(None, recs, p.bs.num_cols, "lm_step")
        assert recs[0]["lm_step"][3] == recs[1]["lm_step"][3] > 0
        if recs[0]["lm_step"][2] == so.num_iterations:
            assert rel(step, -xo) <= 1e-9
            Jx = m0.right_multiply(p.values, step)
            assert abs(recs[0]["lm_step"][3] - (-(Jx @ (p.b + Jx / 2)))) <= 1e-9 * abs(recs[0]["lm_step"][3])
        if solver_type == hip.ITERATIVE_SCHUR:
            isc = oracle.ImplicitSchurComplement(m)
            isc.init(p.values, p.D, p.b)
            xf = np.random.default_rng(5).standard_normal(m.num_cols_f)
            inv, _ = m.schur_jacobi(p.values, p.D)
            for rec in recs:
                assert rel(rec["rhs"], isc.rhs()) <= 1e-12 and rel(rec["sx"], isc.sx(xf)) <= 1e-12 and rel(rec["precond"], inv) <= 1e-11
        else:
            xx = np.random.default_rng(6).standard_normal(p.bs.num_cols)
            want = m0.left_multiply(p.values, m0.right_multiply(p.values, xx)) + p.D ** 2 * xx
            g = m0.left_multiply(p.values, p.b)
            for rec in recs:
                ci = rec["col_index"]
                assert rel(rec["jtjx"], want[ci]) <= 1e-12 and rel(rec["jtb"], g[ci]) <= 1e-12


def test_two_ranks_generic_structure(hip, oracle, problems):
    kw = dict(kind="general", seed=6, ne=40, nf=7, solvers=[(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)], max_it=3000)
    res = run_ranks([("general", kw)])
    p = problems.random_schur_problem(num_e_blocks=40, num_f_blocks=7, num_no_e_rows=2, seed=6)
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    m0 = oracle.Matrix(p.bs, 0)
    for solver_type, pre in kw["solvers"]:
        recs = [res[r][("general", solver_type, pre)] for r in range(WORLD)]
        assert all(rec["path"] == hip.PATH_GENERIC for rec in recs)
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        xo, so = fn(p.values, p.b, p.D, preconditioner=pre, min_it=0, max_it=3000, q_tol=-1.0, r_tol=1e-12)
        assert all(rec["converged"][1] == hip.SUCCESS for rec in recs), [rec["converged"][1:] for rec in recs]
        assert rel(assemble(None, recs, p.bs.num_cols, "converged"), xo) <= 1e-7


def test_two_ranks_cgnr_fused_iteration_equals_the_six_kernel_iteration(hip, problems):
    """Sharded CGNR on the <2,3,9> path: the two-launch iteration (the shard's p.q share travels with the camera vector through
    the operator's all-reduce, r.z / Q1 / |r|^2 of the shard in one 4-double all-reduce) against the six-kernel sequence with
    its four scalar all-reduces per iteration: same iteration counts, same termination, solutions equal to rounding — for a
    converged solve (incl. residual resets every 10th iteration) and for the LM step."""
    base = dict(kind="bal", seed=77, nc=45, np=5000, no=23000, skew=0.4, solvers=[(hip.CGNR, hip.JACOBI)], max_it=400)
    res = run_ranks([("fused", dict(base, cg_fused="1")), ("unfused", dict(base, cg_fused="0"))])
    for r in range(WORLD):
        a, b = res[r][("fused", hip.CGNR, hip.JACOBI)], res[r][("unfused", hip.CGNR, hip.JACOBI)]
        for key in ("converged", "lm_style", "lm_step"):
            assert a[key][1] == b[key][1] == hip.SUCCESS and a[key][2] == b[key][2], (key, a[key][1:3], b[key][1:3])
            assert rel(a[key][0], b[key][0]) <= 1e-10, key
        assert a["converged"][2] > 12   # long enough to contain a residual reset
        assert abs(a["lm_step"][3] - b["lm_step"][3]) <= 1e-10 * abs(b["lm_step"][3])
    # and the ranks agree with each other on the replicated part, bit for bit, in the fused mode too
    a0, a1 = res[0][("fused", hip.CGNR, hip.JACOBI)], res[1][("fused", hip.CGNR, hip.JACOBI)]
    assert np.array_equal(a0["converged"][0][a0["n_e"]:], a1["converged"][0][a1["n_e"]:])
