#!/usr/bin/env python3
"""The randomised parity campaign on SHARDED instances (GPU box; the ranks share device 0 like tests/test_gpu_multirank.py): a random
structure of tests/fuzz_cases.py sharded by point over 2 .. 4 ranks — ITERATIVE_SCHUR + SCHUR_JACOBI, and CGNR + JACOBI where the columns
are points-then-cameras — converged solve, the LM-style solve and the LM step on the device assembled over ranks against the oracle on the
whole problem; the replicated camera part must be bit-identical on every rank.

usage: fuzz_multirank.py [first_seed] [count]      one JSON line per case; exit code 1 if any case failed"""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
oracle = entry.load_oracle()
hip = pkg.hip_solver
P = pkg.problems
import fuzz_cases  # noqa: E402
from step_check import assert_lm_style_step  # noqa: E402
from test_gpu_multirank import assemble, run_ranks  # noqa: E402
from test_gpu_operators import rel  # noqa: E402


BIG = "--big" in sys.argv   # half a million to three million observations (every rank builds the whole problem: a minute per case)


def run_case(seed):
    case, k, _ = fuzz_cases.draw_case(seed, BIG)
    rng = np.random.default_rng(seed + 123)
    world = int(rng.choice([2, 3, 4, 8]))
    out = dict(case, world=world)
    if case["n_points"] < world or (case["n_obs"] > 60000 and not BIG):
        return dict(out, ok=True, skipped="fewer points than ranks, or a large case (kept short: every case starts its own processes)")
    p = fuzz_cases.build(pkg.problems, case, k)
    nr, ne, nf = case["shape"]
    solvers = [(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)]
    if not case["shared"] and nf != ne:   # sharded CGNR tells points from cameras by their width, in points-then-cameras columns
        solvers.append((hip.CGNR, hip.JACOBI))
    kw = dict(kind="fuzz", seed=seed, big=BIG, solvers=solvers, radius=1.0, max_it=600, p2p_timeout=60 if BIG else 8)
    t0 = time.time()
    res = run_ranks([("fuzz", kw)], world)
    m, m0 = oracle.Matrix(p.bs, p.num_eliminate_blocks), oracle.Matrix(p.bs, 0)
    diag = np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32)
    worst = {}
    for solver_type, pre in solvers:
        recs = [res[r][("fuzz", solver_type, pre)] for r in range(world)]
        tag = "schur" if solver_type == hip.ITERATIVE_SCHUR else "cgnr"
        out[tag + "_path"] = [int(rec["path"]) for rec in recs]
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        solve = lambda lo, hi, q, r, D=p.D: fn(p.values, p.b, D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r)
        xo, so = solve(0, 600, -1.0, 1e-12)
        assert all(rec["converged"][1] == so.termination_type for rec in recs), ([rec["converged"][1:] for rec in recs], so)
        worst[tag + ":converged"] = float(rel(assemble(None, recs, p.bs.num_cols, "converged"), xo))
        for key in ("converged", "lm_style", "lm_step", "retry"):   # the camera part is replicated: identical bits on every rank
            a = recs[0][key][0][recs[0]["n_e"]:]
            for rec in recs[1:]:
                assert np.array_equal(a, rec[key][0][rec["n_e"]:]), (key, "replicated part differs between ranks")
                assert recs[0][key][2] == rec[key][2] and recs[0][key][4] == rec[key][4], (key, recs[0][key][1:], rec[key][1:])

        class S:
            termination_type, num_iterations, message = recs[0]["lm_style"][1], recs[0]["lm_style"][2], recs[0]["lm_style"][4]
        if "zeta" in S.message:
            assert_lm_style_step(assemble(None, recs, p.bs.num_cols, "lm_style"), S, solve, 0.1, hip.SUCCESS)
        step = assemble(None, recs, p.bs.num_cols, "lm_step")
        S.termination_type, S.num_iterations, S.message = recs[0]["lm_step"][1], recs[0]["lm_step"][2], recs[0]["lm_step"][4]
        if "zeta" in S.message:
            Dlm = np.sqrt(diag / 1.0)
            assert_lm_style_step(-step, S, lambda lo, hi, q, r: solve(lo, hi, q, r, Dlm), 0.1, hip.SUCCESS)
            Jx = m0.right_multiply(p.values, step)
            want = -(Jx @ (p.b + Jx / 2))
            worst[tag + ":model_cost"] = float(abs(recs[0]["lm_step"][3] - want) / max(abs(want), 1e-300))
        for rec in recs:   # the streamed upload of a shard: the step it leads to is the plain call's
            assert (rec["streamed"][1], rec["streamed"][2]) == (rec["lm_step"][1], rec["lm_step"][2]), (rec["streamed"][1:], rec["lm_step"][1:])
            if np.isfinite(rec["lm_step"][0]).all():
                worst[tag + ":streamed"] = max(worst.get(tag + ":streamed", 0.0), float(rel(rec["streamed"][0], rec["lm_step"][0])) * 1e3)   # (1e-11 counts as the 1e-8 bar)
        step = assemble(None, recs, p.bs.num_cols, "retry")   # values_unchanged at half the radius (TrustRegionMinimizer after a rejected step)
        S.termination_type, S.num_iterations, S.message = recs[0]["retry"][1], recs[0]["retry"][2], recs[0]["retry"][4]
        if "zeta" in S.message:
            Dlm = np.sqrt(diag / 0.5)
            assert_lm_style_step(-step, S, lambda lo, hi, q, r: solve(lo, hi, q, r, Dlm), 0.1, hip.SUCCESS)
            Jx = m0.right_multiply(p.values, step)
            want = -(Jx @ (p.b + Jx / 2))
            worst[tag + ":retry_model_cost"] = float(abs(recs[0]["retry"][3] - want) / max(abs(want), 1e-300))
    bad = {a: b for a, b in worst.items() if not (b <= 1e-8)}
    return dict(out, ok=not bad, worst=max(worst.values()), worst_key=max(worst, key=worst.get), bad=bad, seconds=round(time.time() - t0, 1))


def run_generic(seed):
    """--generic: a structure that is NOT bundle adjustment (random E|F-partitioned blocks 1 .. 4 wide: the generic kernels), sharded by
    E block: converged solves, the LM-style call, the LM step and its retry, the operators that sum over ranks — against the oracle."""
    rng = np.random.default_rng(7000003 * seed + 11)
    static = [None, None, (2, 3, 6), (1, 1, 1), (3, 2, 4)][int(rng.integers(5))]
    world = int(rng.choice([2, 3, 4, 8]))
    pk = dict(num_e_blocks=int(rng.choice([8, 40, 300])), num_f_blocks=int(rng.choice([1, 2, 9, 40])), max_rows_per_e=int(rng.choice([1, 4, 9])),
              num_no_e_rows=int(rng.choice([0, 3, 20])), static_sizes=static, seed=seed)
    p = P.random_schur_problem(**pk)
    out = dict(generic=True, world=world, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in pk.items()})
    solvers = [(hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI), (hip.CGNR, hip.JACOBI)]
    kw = dict(kind="general_fuzz", problem=pk, solvers=solvers, radius=1.0, max_it=3000, p2p_timeout=8)
    t0 = time.time()
    res = run_ranks([("gfuzz", kw)], world)
    m, m0 = oracle.Matrix(p.bs, p.num_eliminate_blocks), oracle.Matrix(p.bs, 0)
    diag = np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32)
    worst = {}
    for solver_type, pre in solvers:
        recs = [res[r][("gfuzz", solver_type, pre)] for r in range(world)]
        tag = "schur" if solver_type == hip.ITERATIVE_SCHUR else "cgnr"
        out[tag + "_path"] = [int(rec["path"]) for rec in recs]
        fn = m.iterative_schur_solve if solver_type == hip.ITERATIVE_SCHUR else m0.cgnr_solve
        solve = lambda lo, hi, q, r, D=p.D: fn(p.values, p.b, D, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=r)
        xo, so = solve(0, 3000, -1.0, 1e-12)
        assert all(rec["converged"][1] == so.termination_type for rec in recs), ([rec["converged"][1:] for rec in recs], so)
        worst[tag + ":converged"] = float(rel(assemble(None, recs, p.bs.num_cols, "converged"), xo)) * 1e-2   # (1e-12 on |r|: 1e-6 on x at these condition numbers; scaled to the 1e-8 bar)
        for key in ("converged", "lm_style", "lm_step", "retry"):
            for rec in recs[1:]:
                assert (recs[0][key][1], recs[0][key][2]) == (rec[key][1], rec[key][2]), (key, recs[0][key][1:], rec[key][1:])

        class S:
            termination_type, num_iterations, message = recs[0]["lm_style"][1], recs[0]["lm_style"][2], recs[0]["lm_style"][4]
        # (a reduced system of one or two unknowns is solved EXACTLY: rho = r'z = 0 in one implementation, 1e-64 and "zeta" in the other — a tie)
        exact = lambda D: "rho = r'z" in solve(0, 500, 0.1, -1.0, D)[1].message
        if "zeta" in S.message and not exact(p.D):
            assert_lm_style_step(assemble(None, recs, p.bs.num_cols, "lm_style"), S, solve, 0.1, hip.SUCCESS)
        for key, radius in (("lm_step", 1.0), ("retry", 0.5)):
            step = assemble(None, recs, p.bs.num_cols, key)
            S.termination_type, S.num_iterations, S.message = recs[0][key][1], recs[0][key][2], recs[0][key][4]
            Dlm = np.sqrt(diag / radius)
            if "zeta" in S.message and not exact(Dlm):
                assert_lm_style_step(-step, S, lambda lo, hi, q, r: solve(lo, hi, q, r, Dlm), 0.1, hip.SUCCESS)
                Jx = m0.right_multiply(p.values, step)
                want = -(Jx @ (p.b + Jx / 2))
                worst[f"{tag}:{key}_model_cost"] = float(abs(recs[0][key][3] - want) / max(abs(want), 1e-300))
        if solver_type == hip.ITERATIVE_SCHUR:
            isc = oracle.ImplicitSchurComplement(m)
            isc.init(p.values, p.D, p.b)
            xf = np.random.default_rng(5).standard_normal(m.num_cols_f)
            inv = m.schur_jacobi(p.values, p.D)[0]
            for rec in recs:
                worst["schur:rhs"] = max(worst.get("schur:rhs", 0.0), float(rel(rec["rhs"], isc.rhs())))
                worst["schur:sx"] = max(worst.get("schur:sx", 0.0), float(rel(rec["sx"], isc.sx(xf))))
                worst["schur:precond"] = max(worst.get("schur:precond", 0.0), float(rel(rec["precond"], inv)))
        else:
            xx = np.random.default_rng(6).standard_normal(p.bs.num_cols)
            want = m0.left_multiply(p.values, m0.right_multiply(p.values, xx)) + p.D ** 2 * xx
            g = m0.left_multiply(p.values, p.b)
            for rec in recs:
                ci = rec["col_index"]
                worst["cgnr:jtjx"] = max(worst.get("cgnr:jtjx", 0.0), float(rel(rec["jtjx"], want[ci])))
                worst["cgnr:jtb"] = max(worst.get("cgnr:jtb", 0.0), float(rel(rec["jtb"], g[ci])))
    bad = {a: b for a, b in worst.items() if not (b <= 1e-8)}
    return dict(out, ok=not bad, worst=max(worst.values()), worst_key=max(worst, key=worst.get), bad=bad, seconds=round(time.time() - t0, 1))


def run_variants(seed):
    """--variants: the solver OPTIONS sharded on a random structure (DENSE_SCHUR and the explicit Schur complement where the reduced
    system is small, the power series as preconditioner and as initialisation, JACOBI, IDENTITY, reset period 1; fused and generic
    kernels) — three iterations each, against ONE instance on the whole problem with the same options."""
    from test_gpu_multirank import _variants
    case, k, _ = fuzz_cases.draw_case(seed)
    rng = np.random.default_rng(seed + 977)
    world = int(rng.choice([2, 3, 4, 8]))
    out = dict(case, world=world)
    if case["n_points"] < world or case["n_obs"] > 30000:
        return dict(out, ok=True, skipped="fewer points than ranks, or a large case")
    p = fuzz_cases.build(pkg.problems, case, k)
    nf_cols = int(p.bs.col_block_size[p.num_eliminate_blocks:].sum())
    nr, ne, nf = case["shape"]
    variants = [v for v in _variants(hip)
                if (nf_cols <= 1200 or not (v["type"] == hip.DENSE_SCHUR or v.get("use_explicit_schur_complement")))
                and (v["type"] != hip.CGNR or (not case["shared"] and nf != ne))]
    variants = [variants[i] for i in sorted(rng.choice(len(variants), size=min(6, len(variants)), replace=False))]
    kw = dict(kind="fuzz", seed=seed, solvers=[], variants=variants, p2p_timeout=30)   # (eight ranks load their code objects on the two host cores of the test box: seconds apart)
    t0 = time.time()
    res = run_ranks([("v", kw)], world)
    worst = {}
    for vi, var in enumerate(variants):
        var = dict(var)
        q_tol, r_tol = var.pop("q_tolerance", -1.0), var.pop("r_tolerance", -1.0)
        tag = f"{var['type']}:{var.get('preconditioner_type', 1)}:{'explicit:' if var.get('use_explicit_schur_complement') else ''}{'spse_init:' if var.get('use_spse_initialization') else ''}{'generic' if var.get('force_generic_path') else 'fused'}"
        recs = [res[r][("v", "variant", vi)] for r in range(world)]
        one = hip.HipLinearSolver(hip.LinearSolverOptions(elimination_groups=[p.num_eliminate_blocks], **var))
        one.set_structure(p.bs)
        xo, so = one.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
        one.close()
        errors = [rec.get("error") for rec in recs]
        assert not any(errors), (tag, errors)
        if "rho = r'z" in so.message or any("rho = r'z" in rec["x"][4] for rec in recs):
            continue   # (a system solved exactly: a tie)
        assert all((rec["x"][1], rec["x"][2]) == (so.termination_type, so.num_iterations) for rec in recs), (tag, [rec["x"][1:] for rec in recs], so)
        x = np.full(p.bs.num_cols, np.nan)
        for rec in recs:
            x[rec["col_index"][: rec["n_e"]]] = rec["x"][0][: rec["n_e"]]
        x[recs[0]["col_index"][recs[0]["n_e"]:]] = recs[0]["x"][0][recs[0]["n_e"]:]
        for rec in recs[1:]:
            assert np.array_equal(recs[0]["x"][0][recs[0]["n_e"]:], rec["x"][0][rec["n_e"]:], equal_nan=True), (tag, "replicated part differs between ranks")
        nx, no = np.isnan(x), np.isnan(xo)
        assert np.array_equal(nx, no), (tag, int(nx.sum()), int(no.sum()))
        worst[tag] = max(worst.get(tag, 0.0), float(rel(x[~nx], xo[~no])) if (~nx).any() else 0.0)
    bad = {a: b for a, b in worst.items() if not (b <= 1e-8)}
    return dict(out, ok=not bad, variants=len(variants), worst=max(worst.values()) if worst else 0.0, worst_key=max(worst, key=worst.get) if worst else "",
                bad=bad, seconds=round(time.time() - t0, 1))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if args else 0
    count = int(args[1]) if len(args) > 1 else 20
    failed = 0
    for seed in range(first, first + count):
        try:
            r = run_generic(seed) if "--generic" in sys.argv else run_variants(seed) if "--variants" in sys.argv else run_case(seed)
        except Exception as ex:
            r = dict(fuzz_cases.draw_case(seed, BIG)[0] if "--generic" not in sys.argv else dict(seed=seed, generic=True), ok=False, error=repr(ex)[:700], trace=traceback.format_exc()[-1200:])
        failed += 0 if r["ok"] else 1
        print(json.dumps(r), flush=True)
    print(json.dumps({"cases": count, "failed": failed}), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
