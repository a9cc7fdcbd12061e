#!/usr/bin/env python3
"""Randomised parity campaign (GPU box): BAL-like structures drawn at random — camera counts around the wavefront / LDS limits, track
lengths around the tile size (63 / 64 / 65, 127 .. 129, 511 .. 513, 700), single-observation points, every compiled camera / point
width and row height, shared blocks, locked cameras, rows without a point cell — each through every operator of both solvers, two
fixed-count solves and one LM step, against the oracle (the checks of tests/test_gpu_operators.py, test_gpu_lm_step.py).

Fixed-count solves run with r_tolerance = -1 (as LevenbergMarquardtStrategy calls them): with r_tolerance = 0 a six-unknown system whose
residual becomes EXACTLY zero in one implementation and 1e-17 in the other ends "converged" here and "maximum iterations" there.

usage: fuzz_parity.py [first_seed] [count] [--stop] [--generic] [--big]     one JSON line per case; exit code 1 if any case failed
(--generic: random E|F-partitioned structures with blocks 1 .. 4 wide instead of bundle-adjustment ones)
"""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (before the HIP library: hip_solver.load_library)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
oracle = entry.load_oracle()
hip = pkg.hip_solver
hip.load_library()
P = pkg.problems
from test_gpu_operators import check_cgnr_operators, check_schur_operators, make_solver, rel  # noqa: E402
from test_gpu_lm_step import check_step  # noqa: E402

OP_TOL = 1e-10      # (the tests assert 1e-12 on their shapes; a few hundred tiny cameras' raw blocks reach 1e-12 .. 1e-11 in max-norm)
STEP_TOL = 1e-9
from fuzz_cases import SHAPES, draw_case  # noqa: E402,F401
import fuzz_cases  # noqa: E402


def build(case, k, rng, layout):
    return fuzz_cases.build(P, case, k, layout)


def rel_x(x, xo):
    """rel(), for solves that may END IN FAILURE: CG forced on past exact convergence (a reduced system with two distinct eigenvalues) fails
    with rho = r'z = 0 and leaves NaN in x — in the reference's algorithm as here; the two must then fail alike (same NaN positions)."""
    nx, no = np.isnan(x), np.isnan(xo)
    if nx.any() or no.any():
        assert np.array_equal(nx, no), "NaN in different places"
        return float(rel(x[~nx], xo[~no])) if (~nx).any() else 0.0
    return float(rel(x, xo))


def path_of(p, typ, pre):
    s = make_solver(hip, p, typ, pre)
    path = s.info().kernel_path
    s.close()
    return path


BIG = "--big" in sys.argv


def run_case(seed):
    case, k, rng = draw_case(seed, BIG)
    out = dict(case)
    t0 = time.time()
    worst = {}
    p = build(case, k, rng, "schur")
    # ---- ITERATIVE_SCHUR side
    path = path_of(p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI)
    out["schur_path"] = int(path)
    errs = check_schur_operators(hip, oracle, p, False, path)
    worst.update({"schur:" + a: float(b) for a, b in errs.items()})
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for kk in sorted({min(kk, max(1, m.num_cols_f // 2)) for kk in (1, 4)}):   # (CG forced on past exact convergence of a two-unknown system is 0 / 0)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, min_it=kk, max_it=kk)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=hip.SCHUR_JACOBI, min_it=kk, max_it=kk, q_tol=-1.0, r_tol=-1.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        worst[f"schur:solve_k{kk}"] = rel_x(x, xo)
    # ---- CGNR side (no elimination order: a 3-wide camera or shared block cannot be told from a point -> generic kernels, still checked)
    q = type(p)(p.bs, p.values, p.b, p.D, 0)
    cpath = path_of(q, hip.CGNR, hip.JACOBI)
    out["cgnr_path"] = int(cpath)
    errs = check_cgnr_operators(hip, oracle, p, False, cpath)
    worst.update({"cgnr:" + a: float(b) for a, b in errs.items()})
    m0 = oracle.Matrix(p.bs, 0)
    for kk in (1, 4):
        s = make_solver(hip, q, hip.CGNR, hip.JACOBI, min_it=kk, max_it=kk)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        xo, so = m0.cgnr_solve(p.values, p.b, p.D, preconditioner=hip.JACOBI, min_it=kk, max_it=kk, q_tol=-1.0, r_tol=-1.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        worst[f"cgnr:solve_k{kk}"] = rel_x(x, xo)
    # ---- one LM step on the device, both solvers
    radius = 1.0   # (D = sqrt(diag(J'J)): well conditioned, see build())
    diag = np.clip(m0.squared_column_norm(p.values), 1e-6, 1e32)
    for typ, pre, pp in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, p), (hip.CGNR, hip.JACOBI, q)):
        s = make_solver(hip, pp, typ, pre, max_it=500)
        step, summ, mcc = s.lm_compute_step(p.values, p.b, radius, 0.1)
        s.close()
        if "zeta" not in summ.message:
            # a solve that does not end on the zeta test (e.g. "Convergence. |b| = 0." when every camera of the problem is locked): the
            # oracle must end the same way with the same iterate
            mm = oracle.Matrix(pp.bs, pp.num_eliminate_blocks if typ == hip.ITERATIVE_SCHUR else 0)
            fn = mm.iterative_schur_solve if typ == hip.ITERATIVE_SCHUR else mm.cgnr_solve
            xo, so = fn(p.values, p.b, np.sqrt(diag / radius), preconditioner=pre, min_it=0, max_it=500, q_tol=0.1, r_tol=-1.0)
            assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
            worst[f"lm_step_other_termination:{typ}"] = float(np.linalg.norm(-step - xo) / max(np.linalg.norm(xo), 1e-300)) if np.linalg.norm(xo) > 0 else float(np.linalg.norm(step))
            continue
        check_step(oracle, hip, pp, typ, pre, np.sqrt(diag / radius), step, summ, mcc, 0.1, STEP_TOL)
    if not BIG:
        extras(case, p, q, m, m0, diag, worst)
    if case["n_obs"] <= 40000:
        extras_boundary(case, p, q, worst)
        extras_variants(case, p, q, m, m0, worst)
    bad = {a: b for a, b in worst.items() if not (b <= (STEP_TOL if ("solve" in a or "retry" in a or "dense" in a or "streamed" in a or "device_pointers" in a or "variant" in a) else OP_TOL))}
    out.update(ok=not bad, worst=max(worst.values()), worst_key=max(worst, key=worst.get), bad=bad, seconds=round(time.time() - t0, 2))
    return out


def extras(case, p, q, m, m0, diag, worst):
    """Second round of the campaign: the retry after a rejected step, CGNR on the same rows in another order, the power-series operator
    and preconditioner, and — where the reduced system is small — the explicit Schur complement and DENSE_SCHUR."""
    rng = np.random.default_rng(case["seed"] + 77)
    # ---- LM step, then the retry at half the radius without re-sending the values (TrustRegionMinimizer after a rejected step)
    for typ, pre, pp in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, p), (hip.CGNR, hip.JACOBI, q)):
        s = make_solver(hip, pp, typ, pre, max_it=500)
        s.lm_compute_step(p.values, p.b, 1.0, 0.1)
        step, summ, mcc = s.lm_compute_step(None, None, 0.5, 0.1, reuse_diagonal=True, values_unchanged=True)
        s.close()
        if "zeta" in summ.message:
            check_step(oracle, hip, pp, typ, pre, np.sqrt(diag / 0.5), step, summ, mcc, 0.1, STEP_TOL)
            worst[f"retry:{typ}"] = 0.0
    # ---- CGNR with the row blocks in a random order (no elimination order: rows are in residual-block order)
    nrb = p.bs.num_row_blocks
    if nrb <= 60000:
        qp = P.permute_rows(q, rng.permutation(nrb))
        cpath = path_of(qp, hip.CGNR, hip.JACOBI)
        errs = check_cgnr_operators(hip, oracle, qp, False, cpath)
        worst.update({"cgnr_permuted:" + a: float(b) for a, b in errs.items()})
        s = make_solver(hip, qp, hip.CGNR, hip.JACOBI, min_it=3, max_it=3)
        x, summ = s.solve(qp.values, qp.b, hip.PerSolveOptions(D=qp.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        xo, so = oracle.Matrix(qp.bs, 0).cgnr_solve(qp.values, qp.b, qp.D, preconditioner=hip.JACOBI, min_it=3, max_it=3, q_tol=-1.0, r_tol=-1.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        worst["cgnr_permuted:solve_k3"] = float(rel(x, xo))
    # ---- SCHUR_POWER_SERIES_EXPANSION: the operator and the preconditioner's apply
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    isc.compute_ftf_inverse()
    s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_POWER_SERIES_EXPANSION, min_num_iterations=0,
                                                    max_num_iterations=100, elimination_groups=[p.num_eliminate_blocks]))
    s.set_structure(p.bs)
    s.load(p.values, p.b, p.D)
    s.schur_init()
    xf, y0 = rng.standard_normal(m.num_cols_f), rng.standard_normal(m.num_cols_f)
    worst["spse:operator"] = float(rel(s.power_series_operator(xf, y0), isc.power_series_operator(xf, y0)))
    worst["spse:apply_5"] = float(rel(s.spse_apply(xf, 5, 0.0), isc.spse_apply(xf, 5, 0.0)))
    s.close()
    # ---- explicit Schur complement (block-sparse lhs) and DENSE_SCHUR where the reduced system is small
    if m.num_cols_f <= 1200 and nrb <= 20000:
        lhs, rhs = m.schur_eliminate(p.values, p.b, p.D)
        nf = rhs.shape[0]
        S = np.triu(lhs.reshape(nf, nf))
        S = S + np.triu(S, 1).T
        sizes = p.bs.col_block_size[p.num_eliminate_blocks:]
        Minv = np.zeros_like(S)
        o = 0
        for n in sizes:
            Minv[o:o + n, o:o + n] = np.linalg.inv(S[o:o + n, o:o + n])
            o += n
        K = 3
        z, so = oracle.cg_dense(S, rhs, Minv=Minv, min_it=K, max_it=K, q_tol=-1.0, r_tol=-1.0)
        s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, elimination_groups=[p.num_eliminate_blocks],
                                                        use_explicit_schur_complement=True, min_num_iterations=K, max_num_iterations=K))
        s.set_structure(p.bs)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        assert summ.num_iterations == so.num_iterations == K, (summ, so)
        ne = p.bs.col_block_pos[p.num_eliminate_blocks]
        worst["explicit:solve_k3"] = float(rel(x[ne:], z))   # NO_CONVERGENCE at the cap: no back-substitution (schur_complement_solver.cc:150-154)
        s = hip.HipLinearSolver(hip.LinearSolverOptions(type=hip.DENSE_SCHUR, elimination_groups=[p.num_eliminate_blocks], max_num_iterations=1))
        s.set_structure(p.bs)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D))
        s.close()
        assert summ.termination_type == hip.SUCCESS, summ
        zd = np.linalg.solve(S, rhs)
        xd = m.schur_back_substitute(p.values, p.b, p.D, zd)
        worst["dense_schur"] = float(rel(x, xd))


def extras_boundary(case, p, q, worst):
    """Third round: the other ways values reach the solver — streamed behind an 'evaluator' (ceres_hip_values_begin / _ready / _end: runs
    of row blocks announced in a random order, some never announced, with and without a Jacobi scale applied on the device), device
    pointers, and the fp32-tile accuracy mode — each against the plain step of the same instance type (which run_case has checked
    against the oracle)."""
    rng = np.random.default_rng(case["seed"] + 311)
    nrb = p.bs.num_row_blocks
    for typ, pre, pp in ((hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, p), (hip.CGNR, hip.JACOBI, q)):
        ref = make_solver(hip, pp, typ, pre, max_it=500)
        scale = None
        vals = p.values
        if rng.random() < 0.5:   # TrustRegionMinimizer's jacobian_scaling_: the UNSCALED values go up, the device scales its copy
            ref.load(p.values, p.b)
            scale = 1.0 / (1.0 + np.sqrt(np.clip(oracle.Matrix(p.bs, 0).squared_column_norm(p.values), 0, None)))
            vals = ref.scale_columns(scale)
        step0, summ0, mcc0 = ref.lm_compute_step(vals, p.b, 1.0, 0.1)
        # ---- streamed
        s = make_solver(hip, pp, typ, pre, max_it=500)
        hv, hb = np.full(p.values.shape[0], np.nan), np.full(p.b.shape[0], np.nan)
        s.values_begin(hv, hb)
        run = int(rng.choice([1, 7, 97, 1000]))
        runs = [(r0, min(nrb, r0 + run)) for r0 in range(0, nrb, run)]
        rng.shuffle(runs)
        if len(runs) > 4000:   # (row-by-row announcements of a large case: announce the first few thousand, _end sends the rest)
            runs = runs[:4000]
        ptr = p.bs.row_cell_ptr.astype(np.int64)
        hv[:] = p.values   # (filled at once: the evaluator's timing is not what this campaign varies)
        hb[:] = p.b
        for r0, r1 in runs:
            if rng.random() < 0.15:
                continue   # never announced
            s.values_ready(r0, r1 - r0)
        s.values_end(scale)
        step, summ, mcc = s.lm_compute_step(None, None, 1.0, 0.1, values_unchanged=True)
        s.close()
        assert summ.num_iterations == summ0.num_iterations and summ.termination_type == summ0.termination_type, (summ, summ0)
        worst[f"streamed:{typ}"] = rel_x(step, step0)
        # ---- device pointers
        s = make_solver(hip, pp, typ, pre, max_it=500)
        tv, tb = torch.from_numpy(np.ascontiguousarray(vals)).cuda(), torch.from_numpy(np.ascontiguousarray(p.b)).cuda()
        tx = torch.empty(p.bs.num_cols, dtype=torch.float64, device="cuda")
        summ_d, mcc_d, finite = s.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), 1.0, 0.1)
        torch.cuda.synchronize()
        s.close()
        assert summ_d.num_iterations == summ0.num_iterations and finite, (summ_d, summ0)
        worst[f"device_pointers:{typ}"] = rel_x(tx.cpu().numpy(), step0)
        ref.close()
    # ---- fp32 tiles (accuracy mode: arithmetic in fp64 on values rounded to fp32; never parity) — <2,3,9> only
    if case["shape"] == [2, 3, 9] and not case["shared"]:
        o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR, preconditioner_type=hip.SCHUR_JACOBI, max_num_iterations=500,
                                    elimination_groups=[p.num_eliminate_blocks], jacobian_storage=1)
        s = hip.HipLinearSolver(o)
        s.set_structure(p.bs)
        step32, summ32, _ = s.lm_compute_step(p.values, p.b, 1.0, 0.1)
        s.close()
        ref = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, max_it=500)
        step64, summ64, _ = ref.lm_compute_step(p.values, p.b, 1.0, 0.1)
        ref.close()
        if summ32.num_iterations == summ64.num_iterations and np.linalg.norm(step64) > 0:
            worst["fp32_tiles_accuracy"] = min(rel_x(step32, step64) * 1e-4, 1.0)   # (scaled: 1e-6 of accuracy counts as 1e-10 here)


def extras_variants(case, p, q, m, m0, worst):
    """Fourth round: the solver OPTIONS — every preconditioner each solver takes (IDENTITY, JACOBI, SCHUR_JACOBI,
    SCHUR_POWER_SERIES_EXPANSION with and without its initialisation), residual reset periods, min / max iteration counts, and the three
    ways CG ends (zeta, |r|, the cap) — against the oracle's solver with the same options: same termination type, counts within one,
    the iterate of the product's own count to 1e-9 (tests/step_check.py's rule, also for |r| terminations)."""
    rng = np.random.default_rng(case["seed"] + 523)
    for rep in range(3):
        schur = rng.random() < 0.6
        pre = int(rng.choice([hip.IDENTITY, hip.JACOBI, hip.SCHUR_JACOBI, hip.SCHUR_POWER_SERIES_EXPANSION])) if schur else int(rng.choice([hip.IDENTITY, hip.JACOBI]))
        reset = int(rng.choice([1, 3, 10, 50]))
        how = str(rng.choice(["zeta", "residual", "cap"]))
        q_tol, r_tol, lo, hi = {"zeta": (float(rng.choice([0.3, 0.1, 0.01])), -1.0, 0, 500),
                                "residual": (-1.0, float(rng.choice([1e-2, 1e-6])), int(rng.choice([0, 3])), 500),
                                "cap": (-1.0, -1.0, 0, int(rng.choice([2, 6])))}[how]
        spse_init = bool(schur and rng.random() < 0.3)
        pp = p if schur else q
        o = hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR if schur else hip.CGNR, preconditioner_type=pre, min_num_iterations=lo, max_num_iterations=hi,
                                    residual_reset_period=reset, elimination_groups=[pp.num_eliminate_blocks],
                                    use_spse_initialization=spse_init, max_num_spse_iterations=5, spse_tolerance=0.1)
        s = hip.HipLinearSolver(o)
        s.set_structure(pp.bs)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=q_tol, r_tolerance=r_tol))
        s.close()

        def osolve(lo_, hi_, q_, r_):
            if schur and (pre == hip.SCHUR_POWER_SERIES_EXPANSION or spse_init):
                return oracle.iterative_schur_solve_spse(m, p.values, p.b, p.D, preconditioner=pre, min_it=lo_, max_it=hi_, reset_period=reset, q_tol=q_, r_tol=r_,
                                                         use_spse_initialization=spse_init, max_num_spse_iterations=5, spse_tolerance=0.1)
            fn = m.iterative_schur_solve if schur else m0.cgnr_solve
            return fn(p.values, p.b, p.D, preconditioner=pre, min_it=lo_, max_it=hi_, reset_period=reset, q_tol=q_, r_tol=r_)
        xo, so = osolve(lo, hi, q_tol, r_tol)
        tag = f"variant:{'schur' if schur else 'cgnr'}:pre{pre}:reset{reset}:{how}{':spse_init' if spse_init else ''}"
        if "rho = r'z" in summ.message or "rho = r'z" in so.message:
            continue   # (exact convergence: see rel_x)
        assert summ.termination_type == so.termination_type, (tag, summ, so)
        # (a) the SEQUENCE: three iterations with these options, iterate against iterate (the parity statement)
        dim = m.num_cols_f if schur else m0.num_cols
        k3 = min(3, max(1, dim // 2))
        o3 = hip.LinearSolverOptions(type=o.type, preconditioner_type=pre, min_num_iterations=k3, max_num_iterations=k3, residual_reset_period=reset,
                                     elimination_groups=[pp.num_eliminate_blocks], use_spse_initialization=spse_init, max_num_spse_iterations=5, spse_tolerance=0.1)
        s3 = hip.HipLinearSolver(o3)
        s3.set_structure(pp.bs)
        x3, summ3 = s3.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s3.close()
        xo3, so3 = osolve(k3, k3, -1.0, -1.0)
        if "rho = r'z" not in summ3.message and "rho = r'z" not in so3.message:
            assert (summ3.termination_type, summ3.num_iterations) == (so3.termination_type, so3.num_iterations), (tag, summ3, so3)
            worst[tag + ":k3"] = rel_x(x3, xo3)
        # (b) where it ENDS: a short solve must end within one iteration of the oracle's, on the same iterate; a long one (unpreconditioned
        # CG takes 70 .. 130 iterations on these systems, and the two sequences drift apart in the sixth digit on the way) only in the
        # same way — the deviation is recorded, not judged
        # (second campaign with this round: unpreconditioned CG on column norms spanning 1e-4 .. 1e5 is chaotic — the ORACLE's own iterate
        # moves by 2e-2 at iteration 20 when b is perturbed by 1e-15, and by 1e-7 when only the reset period changes — and on a
        # one-observation problem it runs on past convergence, where every iteration multiplies the rounding noise by twenty:
        # tools/probes/variant_seed.py.  So the end of a solve is compared by kind and, for short solves, by count; never by iterate.)
        if max(summ.num_iterations, so.num_iterations) <= 12:
            assert abs(summ.num_iterations - so.num_iterations) <= 1, (tag, summ, so)


def run_generic(seed):
    """A structure that is NOT bundle adjustment: random E|F-partitioned blocks of sizes 1 .. 4 (the generic kernels)."""
    rng = np.random.default_rng(7000003 * seed + 3)
    static = [None, None, (2, 3, 6), (1, 1, 1), (3, 2, 4)][int(rng.integers(5))]
    kw = dict(num_e_blocks=int(rng.choice([1, 3, 40, 300])), num_f_blocks=int(rng.choice([1, 2, 9, 40])), max_rows_per_e=int(rng.choice([1, 4, 9])),
              num_no_e_rows=int(rng.choice([0, 3, 20])), static_sizes=static, seed=seed)
    p = P.random_schur_problem(**kw)
    t0 = time.time()
    worst = {}
    errs = check_schur_operators(hip, oracle, p, False, path_of(p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI))
    worst.update({"schur:" + a: float(b) for a, b in errs.items()})
    errs = check_cgnr_operators(hip, oracle, p, False, path_of(type(p)(p.bs, p.values, p.b, p.D, 0), hip.CGNR, hip.JACOBI))
    worst.update({"cgnr:" + a: float(b) for a, b in errs.items()})
    m = oracle.Matrix(p.bs, p.num_eliminate_blocks)
    for kk in sorted({min(kk, max(1, m.num_cols_f // 2)) for kk in (1, 4)}):   # (CG forced on past exact convergence of a two-unknown system is 0 / 0)
        s = make_solver(hip, p, hip.ITERATIVE_SCHUR, hip.SCHUR_JACOBI, min_it=kk, max_it=kk)
        x, summ = s.solve(p.values, p.b, hip.PerSolveOptions(D=p.D, q_tolerance=-1.0, r_tolerance=-1.0))
        s.close()
        xo, so = m.iterative_schur_solve(p.values, p.b, p.D, preconditioner=hip.SCHUR_JACOBI, min_it=kk, max_it=kk, q_tol=-1.0, r_tol=-1.0)
        assert (summ.termination_type, summ.num_iterations) == (so.termination_type, so.num_iterations), (summ, so)
        worst[f"schur:solve_k{kk}"] = rel_x(x, xo)
    bad = {a: b for a, b in worst.items() if not (b <= (STEP_TOL if "solve" in a else OP_TOL))}
    return dict(generic=True, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}, ok=not bad, worst=max(worst.values()),
                worst_key=max(worst, key=worst.get), bad=bad, seconds=round(time.time() - t0, 2))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    first = int(args[0]) if args else 0
    count = int(args[1]) if len(args) > 1 else 50
    stop = "--stop" in sys.argv
    failed = 0
    generic = "--generic" in sys.argv
    for seed in range(first, first + count):
        try:
            r = run_generic(seed) if generic else run_case(seed)
        except Exception as ex:   # an assertion of the shared checkers, or an error code of the library
            case = dict(seed=seed, generic=True) if generic else draw_case(seed, BIG)[0]
            r = dict(case, ok=False, error=repr(ex)[:600], trace=traceback.format_exc()[-900:])
            if isinstance(ex, AssertionError) and "rho = r'z = 0.000000e+00" in repr(ex) and ("Maximum number of iterations" in repr(ex) or "zeta = -0.0" in repr(ex) or "zeta = 0.0" in repr(ex)):
                # CG forced on past EXACT convergence: r'z is exactly 0 in one implementation (FAILURE, as the reference would report) and
                # 1e-33 in the other (which iterates on): a tie on a rounding, in systems of a handful of distinct eigenvalues
                r.update(ok=True, tie_at_exact_convergence=True)
        failed += 0 if r["ok"] else 1
        print(json.dumps(r), flush=True)
        if failed and stop:
            break
    print(json.dumps({"cases": count, "failed": failed}), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
