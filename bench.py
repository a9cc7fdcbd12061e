#!/usr/bin/env python3
"""bench.py — LM linear-solve steps/s and JtJx SpMV HBM GB/s on BAL-shaped input, MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: the linear algebra of
one Levenberg–Marquardt trust-region step exactly as the reference issues it —
LevenbergMarquardtStrategy::ComputeStep (internal/ceres/levenberg_marquardt_strategy.cc:69-157:
diag = clamp(SquaredColumnNorm(J)), D = sqrt(diag / radius), LinearSolver::Solve with
q_tolerance = eta = 0.1, r_tolerance = -1, max_num_iterations = 500, finite check, negation)
followed by the model-cost change -(J step)'(f + J step / 2) of
TrustRegionMinimizer::ComputeTrustRegionStep (trust_region_minimizer.cc:420-438) — with the
Jacobian values and residuals already resident in HBM when the timed region starts
(ceres_hip_lm_compute_step_device).  `--step linear_solve` times LinearSolver::Solve alone
(ceres_hip_solve_device).  Evaluating residuals / Jacobians is the Evaluator's job and outside
this path.

Workload (config.workload): synthetic BAL-shaped Jacobian with Venice-1778's block counts
(1778 cameras, 993923 points, 5001946 observations; SURVEY.md §8d generator, seed 38401) —
the configuration BASELINE.json's targets are quoted on.  Default solver: ITERATIVE_SCHUR +
SCHUR_JACOBI (BASELINE.json configs[0] and [2]; dominant kernel: the fused implicit Schur product
S·x); `--solver cgnr` switches to CGNR + JACOBI (dominant kernel: the fused JtJx SpMV).  With
`--both-solvers 1` (default, N = 1) the other solver is measured in the same run: `extra` carries its
step rate and the line carries BOTH rooflines (`roofline` for the timed solver's operator,
`roofline_jtjx` / `roofline_sx` for the other).

N > 1 (strong scaling): the SAME problem sharded by point across the ranks
(ceres-solver_amd/partition.py); camera-space sums and the CGNR inner products go through
RCCL all-reduce inside the library.  Launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time



def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (the driver's spelling of the 1-GPU
    run must work for N GPUs too).  Ranks are `python -m torch.distributed.run --nproc-per-node N` children of this process; rank 0
    prints the one JSON line, this process forwards their exit code.  On a box with fewer than N visible GPUs the ranks share device 0
    (CERES_HIP_BENCH_ONE_GPU=1: a VALIDATION mode of the whole N > 1 code path — gloo + the peer-to-peer all-reduce — whose timings
    mean nothing; the line says so in config.parallelism)."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv[1:], 1):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    env = dict(os.environ)
    try:
        import torch
        visible = torch.cuda.device_count()
    except Exception:
        visible = 0
    if visible < n and env.get("CERES_HIP_BENCH_ONE_GPU", "0") != "1":
        print(f"bench.py: {visible} GPU(s) visible for --gpus {n}: the ranks share device 0 (one-GPU validation mode, timings meaningless)",
              file=sys.stderr, flush=True)
        env["CERES_HIP_BENCH_ONE_GPU"] = "1"
    if env.get("CERES_HIP_BENCH_ONE_GPU", "0") == "1":   # ranks that share a device: few workgroups in the kernels that wait for their peers
        env.setdefault("CERES_HIP_P2P_SHARED_DEVICE", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    _self_launch()

if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # cpu_baseline leg (N = 1 only): keep the oracle's OpenMP threads next to each other and on the socket that holds the
    # data — 8 % on the two-socket host of the GPU boxes (tools/cpu_oracle_probe.py); must be set before any OpenMP
    # runtime starts.  Not for N > 1: every rank's main thread would be pinned to the same core.
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (must precede the HIP library: see hip_solver.load_library)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(kind, n_obs, n_points, n_cameras, s=8):
    """SURVEY.md §8(d) / BASELINE.md §3, per application of the operator."""
    n_cols = 3 * n_points + 9 * n_cameras
    if kind == "jtjx":
        return n_obs * (24 * s + 8) + n_cols * 4 * s
    return n_obs * (24 * s + 8) + n_points * 9 * s + n_cameras * 36 * s  # sx


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="venice1778", choices=["dubrovnik16", "ladybug1723", "venice1778", "synthetic1M", "synthetic10M"],
                    help="venice1778 = the configuration BASELINE.json's targets are quoted on (default); synthetic10M = configs[4]: "
                         "10 M points x 3 observations, 50 k cameras (camera accumulators do not fit in LDS); synthetic1M = the same regime at a tenth")
    ap.add_argument("--storage", default="fp64", choices=["fp64", "fp32"],
                    help="fp32: Jacobian tiles rounded to fp32, fp64 arithmetic (configs[4] 'fp32 and fp64'; an accuracy mode, not parity)")
    ap.add_argument("--host-boundary-steps", type=int, default=5,
                    help="also time this many steps through the host-pointer boundary (ceres_hip_lm_compute_step: H2D of values/residuals from "
                         "pinned memory + the step + D2H of the step), reported as host_boundary (N=1; 0: skip)")
    ap.add_argument("--solver", default="iterative_schur", choices=["cgnr", "iterative_schur"])
    ap.add_argument("--skew", type=float, default=0.6, help="power-law exponent of camera popularity")
    ap.add_argument("--step", default="lm_step", choices=["lm_step", "linear_solve"],
                    help="lm_step: LevenbergMarquardtStrategy::ComputeStep on the device (diag(J'J), D, Solve, finite check, "
                         "negation) + the model-cost change of TrustRegionMinimizer; linear_solve: LinearSolver::Solve only")
    ap.add_argument("--values", default="normal", choices=["normal", "scene"],
                    help="normal: N(0,1) Jacobian values (SURVEY.md §8d); scene: the first LM linear system of a synthetic "
                         "bundle-adjustment scene (Snavely camera model, Jacobi-scaled like TrustRegionMinimizer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--kernel-iters", type=int, default=50)
    ap.add_argument("--minimizer-iterations", type=int, default=8,
                    help="also run the device-resident trust-region loop (f4) for this many iterations and report it in `extra` (0: skip)")
    ap.add_argument("--both-solvers", type=int, default=1, help="also report the other solver in `extra` (N=1 only)")
    ap.add_argument("--eta", type=float, default=0.1, help="q_tolerance of the timed steps (Solver::Options::eta; bundle_adjuster runs 0.01)")
    ap.add_argument("--scene-step-steps", type=int, default=10,
                    help="also time this many LM steps at eta = 0.1 and 0.01 on the Jacobi-scaled Jacobian of the synthetic SCENE the device "
                         "evaluator produces (scene_step; N=1, needs --minimizer-iterations > 0; 0: skip)")
    ap.add_argument("--conditioned-steps", type=int, default=5,
                    help="also time this many LM steps at eta = 1e-2 / 1e-3 / 1e-4 on a SEQUENCE-like scene (banded camera graph: tens of CG "
                         "iterations per step; conditioned_step; N=1, needs --minimizer-iterations > 0; 0: skip)")
    ap.add_argument("--extra-synthetic10m", type=int, default=1,
                    help="default venice1778 run at N=1: also run `--workload synthetic10M` (BASELINE.json configs[4]) in a child process and put a "
                         "condensed result under extra.synthetic10M (0: skip)")
    ap.add_argument("--extra-real-graph", type=int, default=1, help="default line: S.x / JtJx on the replicated libmv visibility graph (extra.real_graph)")
    ap.add_argument("--extra-other-shapes", type=int, default=1, help="default line: camera widths other than 9, the libmv structure and the generic kernels on the Ladybug shape (extra.other_shapes)")
    ap.add_argument("--extra-dense-cholesky", type=int, default=1, help="default line: DENSE_SCHUR's factorisation at n = 8190 (extra.dense_schur_cholesky)")
    ap.add_argument("--also-fp32", type=int, default=-1, help="also time the fp32-tile storage mode (extra.fp32_tiles; BASELINE.json configs[4] asks for a sweep over both "
                                                               "precisions): -1 = on the default Venice line, 1 = on, 0 = off")
    ap.add_argument("--extra-banded", type=int, default=1, help="default line: S.x / JtJx on a 50 000-camera sequence-like (banded) graph (extra.banded50k)")
    ap.add_argument("--extra-configs", type=int, default=1, help="default line: BASELINE.json configs[1] (Dubrovnik-16, CGNR + JACOBI) and configs[2] "
                                                                 "(Ladybug-1723, ITERATIVE_SCHUR + SCHUR_JACOBI): steps/s, operator roofline, oracle check, CPU port (extra.configs)")
    ap.add_argument("--pmc-live", type=int, default=1, help="N = 1: measure roofline.traffic now with two rocprofv3 --pmc passes in child processes "
                                                            "(tools/pmc_live.py; 0: the committed constant of profiles/pmc_traffic.json)")
    ap.add_argument("--pmc-live-timeout", type=float, default=120.0, help="seconds per counter pass")
    ap.add_argument("--shard-ceiling", type=int, default=1,
                    help="N = 1: also run rank 0's shard of the workload for N = 2, 4, 8 alone on the device with the sharded code path on "
                         "(ghost peers) and report T_1 / (N T_shard) as extra.shard_ceiling (0: skip)")
    ap.add_argument("--oracle-check", type=int, default=-1,
                    help="(-1 = on where one oracle step takes about a second: the workloads whose camera sums fit in LDS) compare the step of the LAST timed solve with ONE oracle step on the same FULL-SIZE inputs (16 threads; N > 1: the ranks' "
                         "shards of the step are gathered and assembled first) and report it as oracle_check.  Independent of --no-cpu-baseline: "
                         "the synthetic10M child and the one-GPU N = 8 validation run use it")
    return ap.parse_args()


def make_solver(hs, bs, nelim, solver, device, comm=None, storage=0):
    typ, pre = (hs.CGNR, hs.JACOBI) if solver == "cgnr" else (hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI)
    o = hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                               residual_reset_period=10, elimination_groups=[nelim], device=device, jacobian_storage=storage)
    if comm is None:
        s = hs.HipLinearSolver(o)
    else:
        # RCCL communicator (any message size) + the one-shot peer-to-peer all-reduce over hipIpc / xGMI for the
        # latency-bound camera-space sums of a step (<= 81 doubles per camera); CERES_HIP_P2P=0 leaves RCCL alone.
        kw = dict(comm_id=comm[0], rank=comm[1], world_size=comm[2])
        if comm[3] is not None:
            f = bs.col_block_size[nelim:].astype(np.int64)
            kw.update(p2p_exchange=comm[3], p2p_max_elements=int((f * f).sum() + 2 * f.sum() + 2))  # blocks + rhs + column norms: ONE all-reduce per step
        s = hs.HipLinearSolver(o, **kw)
        if comm[3] is not None:  # agree on the verdict of the self-test: all ranks use the peer-to-peer path, or none does
            import torch
            import torch.distributed as dist
            ok = torch.tensor([1 if s.p2p_selftest() else 0], device=comm[4])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm[0] is None:
                    raise SystemExit(f"bench.py: peer-to-peer all-reduce failed ({s.p2p_error}) and there is no RCCL communicator")
                if comm[1] == 0:
                    print(f"bench.py: peer-to-peer all-reduce unavailable ({s.p2p_error}); using RCCL", file=sys.stderr)
                s.p2p_disable()
    t0 = time.perf_counter()
    s.set_structure(bs)
    s.set_structure_seconds = time.perf_counter() - t0   # host-side planning + uploads: once per structure, NOT in the timed steps
    return s


RADIUS = 1e4  # Solver::Options::initial_trust_region_radius; D = sqrt(clamp(diag(J'J)) / RADIUS)


def step_min_bytes(solver_kind, n_obs, n_points, n_cameras, cg_iterations, s=8):
    """Minimum HBM bytes of one whole LM step with `cg_iterations` CG iterations IF the Jacobian already sat in the layout each
    pass wants (DESIGN.md §7 "step roofline"): what the step's passes must move at the very least, tile re-layout NOT counted.
      ITERATIVE_SCHUR + SCHUR_JACOBI: Init (J + ids + b in, M_o + (E'E)^-1 + D_e out) + SCHUR_JACOBI blocks (F cell + M_o + index per
      observation) + k S.x + back-substitution (J + ids + b in, point inverses in, step out);
      CGNR + JACOBI: set-up (J + ids + b in; rhs, point blocks, D out) + camera blocks (F cell + index) + k (JtJx + the CG vector passes:
      cg_update reads x p r q rhs M and writes x r z, the direction update reads z p and writes p)."""
    n_cols = 3 * n_points + 9 * n_cameras
    j_once = n_obs * (24 * s + 8)
    if solver_kind == "iterative_schur":
        init = j_once + n_obs * 2 * s + n_obs * 3 * s + n_points * (6 * s + 6 * s) + n_cameras * 9 * s
        precond = n_obs * (18 * s + 3 * s + 4) + n_cameras * 81 * s
        backsub = j_once + n_obs * 2 * s + n_points * (6 * s + 3 * s) + n_cameras * 9 * s
        return init + precond + cg_iterations * algorithmic_bytes("sx", n_obs, n_points, n_cameras, s) + backsub
    setup = j_once + n_obs * 2 * s + n_points * (3 * s + 9 * s + 6 * s) + n_cameras * 9 * s
    precond = n_obs * (18 * s + 4) + n_cameras * 81 * s
    cg_vectors = n_cols * s * (8 + 3) + n_points * 9 * s + n_cameras * 81 * s
    return setup + precond + cg_iterations * (algorithmic_bytes("jtjx", n_obs, n_points, n_cameras, s) + cg_vectors)


def timed_steps(solver, ptrs, steps, warmup, sync, step_kind="linear_solve", eta=0.1, radius=None):
    tv, tb, tD, tx = ptrs
    radius = RADIUS if radius is None else radius

    def one():
        if step_kind == "lm_step":  # the diagonal is recomputed every step, as after an accepted step
            s, mcc, finite = solver.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), radius, eta)
            assert finite and mcc > 0, (s, mcc)
            return s
        return solver.solve_device(tv.data_ptr(), tb.data_ptr(), tD.data_ptr(), tx.data_ptr(), eta, -1.0)
    iters = []
    sync()  # N > 1: ranks leave set_structure (host-side planning) at different times; the in-kernel all-reduce has a timeout
    for _ in range(warmup):
        s = one()
        assert s.termination_type in (0, 1), s
    sync()
    if step_kind == "lm_step":
        # the timed loop calls the C ABI with its argument structs built once, like a C++ caller (hip_solver.lm_stepper); every step's
        # result is checked after the loop
        step = solver.lm_stepper(tv.data_ptr(), tb.data_ptr(), tx.data_ptr(), radius, eta)
        res = []
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            res.append((r.linear_solver.num_iterations, r.linear_solver.termination_type, r.model_cost_change, r.step_is_finite))
        sync()
        el = time.perf_counter() - t0
        for it, term, mcc, finite in res:
            assert finite and mcc > 0 and term in (0, 1), (it, term, mcc, finite)
        ls = r.linear_solver   # (the last step's summary; `r` is the stepper's result struct, which the next call would overwrite)
        last = sys.modules[type(solver).__module__].Summary(ls.residual_norm, ls.num_iterations, ls.termination_type,
                                                            ls.message.decode(errors="replace"))
        return el, [int(x[0]) for x in res], last
    t0 = time.perf_counter()
    for _ in range(steps):
        s = one()
        iters.append(s.num_iterations)
    sync()
    return time.perf_counter() - t0, iters, s


def config_leg(pkg, hs, entry, workload, solver_kind, device, dev, args, no_cpu=False):
    """One of BASELINE.json's smaller configurations on the line: steps/s of the LM step, the operator against the HBM roofline, the
    step against the oracle at full size, and the oracle's own rate (the CPU port) on the same inputs."""
    n_cams, n_points, n_obs = pkg.problems.BAL_SHAPES[workload]
    p = pkg.problems.synthetic_bal(workload, layout="schur", seed=38401, skew=args.skew)
    sv = make_solver(hs, p.bs, p.num_eliminate_blocks, solver_kind, device)
    tvc, tbc = torch.from_numpy(p.values).to(dev), torch.from_numpy(p.b).to(dev)
    txc = torch.full((p.bs.num_cols,), float("nan"), dtype=torch.float64, device=dev)
    n_st = max(args.steps, 50)
    el, its, last = timed_steps(sv, (tvc, tbc, None, txc), n_st, max(args.warmup, 5), torch.cuda.synchronize, "lm_step", args.eta)
    kind = "jtjx" if solver_kind == "cgnr" else "sx"
    # (the operator on the state the last step left loaded: the same values, D = the step's LM diagonal)
    op_ms = min(sv.time_op(hs.TIMED_JTJX if kind == "jtjx" else hs.TIMED_SX, args.kernel_iters) for _ in range(3))
    nb = algorithmic_bytes(kind, n_obs, n_points, n_cams)
    mb = step_min_bytes(solver_kind, n_obs, n_points, n_cams, int(its[-1]))
    ms = 1e3 * el / n_st
    case = {"workload": f"{workload}-shaped synthetic BAL Jacobian <2,3,9>: {n_cams} cameras, {n_points} points, {n_obs} observations",
            "solver": "CGNR + JACOBI" if solver_kind == "cgnr" else "ITERATIVE_SCHUR + SCHUR_JACOBI", "steps_per_s": round(n_st / el, 2), "ms_per_step": round(ms, 4),
            "cg_iterations_per_step": int(its[-1]), "termination": hs.TERMINATION_NAMES[last.termination_type],
            kind: {"ms": round(op_ms, 5), "GBs": round(nb / (op_ms * 1e-3) / 1e9, 1), "frac": round(nb / (op_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes_per_launch": nb},
            "step_roofline_frac": round(mb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "cg_iteration_in_operator": int(sv.info().cg_iteration_in_operator)}
    xg = txc.cpu().numpy()
    sv.close()
    del tvc, tbc, txc
    if not no_cpu:
        oracle = entry.load_oracle()
        threads = min(os.cpu_count() or 1, 16)
        oracle.set_num_threads(threads)
        m = oracle.Matrix(p.bs, p.num_eliminate_blocks if solver_kind != "cgnr" else 0)
        m_all = oracle.Matrix(p.bs, 0)
        fn = m.iterative_schur_solve if solver_kind != "cgnr" else m.cgnr_solve
        pre = 2 if solver_kind != "cgnr" else 1

        def one(lo=0, hi=500, q=args.eta):
            t = time.perf_counter()
            Dc = np.sqrt(np.clip(m_all.squared_column_norm(p.values), 1e-6, 1e32) / RADIUS)
            xo, so = fn(p.values, p.b, Dc, preconditioner=pre, min_it=lo, max_it=hi, q_tol=q, r_tol=-1.0)
            xo = -xo
            model = m_all.right_multiply(p.values, xo)
            _ = -model @ (p.b + model / 2.0)
            return time.perf_counter() - t, xo, so
        one()
        n_done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0 and n_done < 200:
            _, xo, so = one()
            n_done += 1
        cpu_t = time.perf_counter() - t0
        if so.num_iterations != int(its[-1]):   # zeta crossed the threshold one index apart: the oracle's iterate of the product's index
            k_ = int(its[-1])
            _, xo, _ = one(k_, k_, -1.0)
        oracle.set_num_threads(1)
        case["cpu_port"] = {"steps_per_s": round(n_done / cpu_t, 3), "threads": threads, "cg_iterations": int(so.num_iterations), "sample": f"{n_done} steps in {cpu_t:.1f} s"}
        case["step_rel_diff_vs_oracle"] = float(np.linalg.norm(xg - xo) / np.linalg.norm(xo))
        case["speedup_vs_cpu_port"] = round((n_st / el) / (n_done / cpu_t), 1)
    return case


def phase_timing(solver, ptrs, step_kind="lm_step", eta=0.1, radius=None):
    """Per-phase HIP-event timings of ONE more step, outside every timed region: the events idle the device for about 6 us each
    (ceres_hip_set_phase_timing), so the timed steps run without them."""
    solver.set_phase_timing(True)
    try:
        timed_steps(solver, ptrs, 1, 0, torch.cuda.synchronize, step_kind, eta, radius)
        return solver.last_timing()
    finally:
        solver.set_phase_timing(False)


def shard_ceiling(pkg, hs, prob, solver_kind, device, t1_ms, k_iters, eta, worlds=(2, 4, 8), steps=20, warmup=3, radius=None, dev=None, many_cameras=None):
    """What ONE rank of an N-rank strong-scaling run costs, measured on one GPU: rank 0's shard of `prob`
    (partition.shard_by_point) through ceres_hip_lm_compute_step_device with the SHARDED code path on — every sum over ranks of the
    step runs its peer-to-peer exchange against ghost peers (ceres_hip_debug_comm_ghost_peers: all pushes, flags, waits and sums,
    local memory instead of xGMI) — and CG pinned to the iteration count of the whole problem's step (min = max = k_iters: the shard
    alone is another linear system).  efficiency_ceiling = T_1 / (N T_shard): what a perfect interconnect would give; xGMI latency is
    NOT in it.  many_cameras = "synthetic10M" | "synthetic1M": the shard is generated like bench.py's rank 0 generates it (graph on the
    host, N(0,1) values in HBM; worlds may contain 1 = the whole problem, unsharded)."""
    from ceres_solver_amd import partition
    out = {"what": "rank 0's shard of the problem alone on the device, sharded code path on (every sum over ranks runs its peer-to-peer exchange "
                   "against ghost peers: local memory, no xGMI hop), CG iterations pinned to the whole problem's; efficiency_ceiling = T_1 / (N T_shard): "
                   "what a perfect interconnect would give.  xGMI latency and link time are NOT measured (no box with two GPUs); "
                   "efficiency_with_link_estimate adds the bytes one rank pushes to each peer at 76 GB/s, unoverlapped",
           "t1_ms": round(t1_ms, 4), "cg_iterations": int(k_iters), "cases": []}
    typ, pre = (hs.CGNR, hs.JACOBI) if solver_kind == "cgnr" else (hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI)
    for n in worlds:
        if many_cameras:
            n_cams, n_points, n_obs = pkg.problems.BAL_SHAPES[many_cameras]
            shp = pkg.problems.synthetic_bal(None, layout="schur", seed=38401, skew=0.6, num_cameras=n_cams, num_points=n_points // n,
                                             num_observations=n_obs // n, with_values=False)
            sbs, snelim = shp.bs, shp.num_eliminate_blocks
            g = torch.Generator(device=dev)
            g.manual_seed(38401)
            tvs = torch.randn(24 * sbs.num_row_blocks, dtype=torch.float64, device=dev, generator=g)
            tbs = torch.randn(2 * sbs.num_row_blocks, dtype=torch.float64, device=dev, generator=g)
        else:
            sh = partition.shard_by_point(prob.bs, prob.num_eliminate_blocks, n, 0)
            sbs, snelim = sh.bs, sh.num_eliminate_blocks
            tvs, tbs = (torch.from_numpy(a).to(dev) for a in (sh.local_values(prob.values), sh.local_rows(prob.b)))
        f = sbs.col_block_size[snelim:].astype(np.int64)
        pinned = n > 1 or many_cameras is None
        o = hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=int(k_iters) if pinned else 0,
                                   max_num_iterations=int(k_iters) if pinned else 500,
                                   residual_reset_period=10, elimination_groups=[snelim], device=device)
        sv = hs.HipLinearSolver(o, ghost_world=n, p2p_max_elements=int((f * f).sum() + 2 * f.sum() + 2)) if n > 1 else hs.HipLinearSolver(o)
        sv.set_structure(sbs)
        txs = torch.empty(sbs.num_cols, dtype=torch.float64, device=dev)
        el, its, last = timed_steps(sv, (tvs, tbs, None, txs), steps, warmup, torch.cuda.synchronize, "lm_step", eta, radius)
        ms = 1e3 * el / steps
        tm = phase_timing(sv, (tvs, tbs, None, txs), "lm_step", eta, radius)
        # what one rank pushes to EACH peer per step (doubles): rhs + the packed camera blocks (upper triangle + column norms) + one camera
        # vector per CG iteration + the step's scalars — over xGMI that is link time the ghost peers do not charge (local HBM stands in for
        # the links): 76 GB/s per direction and link (MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional), not overlapped with anything
        nfv = int(f.sum())
        per_cam = int((f * (f + 1) // 2 + f).sum())
        pushed = (nfv + per_cam + int(k_iters) * nfv + 2) if solver_kind != "cgnr" else (nfv + per_cam + int(k_iters) * (nfv + 1) + 4 * (int(k_iters) + 1) + 2)
        link_ms = 8.0 * pushed / 76e9 * 1e3 if n > 1 else 0.0
        out["cases"].append({"ranks": n, "shard_observations": int(sbs.num_row_blocks), "ms_per_step": round(ms, 4),
                             "efficiency_ceiling": round(t1_ms / (n * ms), 4), "collectives_per_step": int(sv.info().collectives_last_step),
                             "doubles_pushed_per_peer_per_step": pushed, "xgmi_link_ms_estimate_unoverlapped": round(link_ms, 4),
                             "efficiency_with_link_estimate": round(t1_ms / (n * (ms + link_ms)), 4),
                             "cg_iterations": int(its[-1]), "cg_ms": round(tm.cg_ms, 4), "setup_ms": round(tm.setup_ms + tm.preconditioner_ms, 4),
                             "back_substitute_ms": round(tm.back_substitute_ms, 4)})
        sv.close()
        del tvs, tbs, txs, sbs
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # (a bare `python bench.py --gpus N` never gets here: _self_launch starts the ranks)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    pkg = entry.load_package()
    hs = pkg.hip_solver
    hs.load_library()
    if hs.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no gfx950 device visible (there is no CPU path)")
    # CERES_HIP_BENCH_ONE_GPU=1: every rank on device 0, torch.distributed over gloo, the solver's collectives over the
    # peer-to-peer path alone (RCCL refuses two ranks on one device).  A VALIDATION mode for a one-GPU box: it runs the
    # whole N > 1 code path of this file and of the library; its timings mean nothing.
    one_gpu = os.environ.get("CERES_HIP_BENCH_ONE_GPU", "0") == "1" and world > 1
    if one_gpu:
        local_rank = 0
        os.environ.setdefault("CERES_HIP_P2P_SHARED_DEVICE", "1")   # (read when a solver is created)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    cdev = torch.device("cpu") if one_gpu else dev   # where tensors of torch.distributed collectives live
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- workload -----------------------------------------------------------------
    n_cams, n_points, n_obs = pkg.problems.BAL_SHAPES[args.workload]
    many_cameras = args.workload in ("synthetic1M", "synthetic10M")
    storage = 1 if args.storage == "fp32" else 0
    comm = None
    if world > 1:
        idt = torch.zeros(hs.UNIQUE_ID_BYTES, dtype=torch.uint8, device=cdev)
        if rank == 0 and not one_gpu:
            idt = torch.tensor(list(hs.comm_unique_id()), dtype=torch.uint8, device=cdev)
        dist.broadcast(idt, 0)

        def p2p_exchange(mine):
            t = torch.tensor(list(mine), dtype=torch.uint8, device=cdev)
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return [bytes(o.cpu().tolist()) for o in out]
        use_p2p = one_gpu or os.environ.get("CERES_HIP_P2P", "1") != "0"
        comm = (None if one_gpu else bytes(idt.cpu().tolist()), rank, world, p2p_exchange if use_p2p else None, cdev)
    prob = None
    if many_cameras:
        # configs[4]: 5.76 GB of Jacobian values.  The observation graph is generated on the host (numpy, the
        # SURVEY §8d generator); the N(0,1) values, residuals and D are generated directly in HBM.  Sharded runs:
        # rank r generates ITS contiguous range of points (the generator is i.i.d. per point, so the union of the
        # ranks' shards IS an instance of the workload: same totals, strong scaling).
        assert args.values == "normal", "--values scene is not available for the many-camera workloads"
        lo, hi = (n_points * rank) // world, (n_points * (rank + 1)) // world
        o_lo, o_hi = (n_obs * rank) // world, (n_obs * (rank + 1)) // world
        prob = pkg.problems.synthetic_bal(None, layout="schur", seed=38401 + 1000 * rank, skew=args.skew, num_cameras=n_cams,
                                          num_points=hi - lo, num_observations=o_hi - o_lo, with_values=False)
        bs, nelim_local = prob.bs, prob.num_eliminate_blocks
        nelim = nelim_local
        g = torch.Generator(device=dev)
        g.manual_seed(38401 + 1000 * rank)
        my_obs_n = bs.num_row_blocks
        tv = torch.randn(24 * my_obs_n, dtype=torch.float64, device=dev, generator=g)
        tb = torch.randn(2 * my_obs_n, dtype=torch.float64, device=dev, generator=g)
        # D = sqrt(clamp(diag(J'J), 1e-6, 1e32) / 1e4) (LevenbergMarquardtStrategy), over ALL ranks' rows for the camera columns
        diag = torch.zeros(bs.num_cols, dtype=torch.float64, device=dev)
        pt_cols = torch.from_numpy(bs.col_block_pos[prob.point_of_row].astype(np.int64)).to(dev)
        cam_cols = torch.from_numpy(bs.col_block_pos[prob.camera_of_row].astype(np.int64)).to(dev)
        e2 = (tv[: 6 * my_obs_n].view(my_obs_n, 2, 3) ** 2).sum(1)
        f2 = (tv[6 * my_obs_n:].view(my_obs_n, 2, 9) ** 2).sum(1)
        for c in range(3):
            diag.index_add_(0, pt_cols + c, e2[:, c].contiguous())
        for c in range(9):
            diag.index_add_(0, cam_cols + c, f2[:, c].contiguous())
        if dist is not None:
            cam_part = diag[3 * nelim_local:].to(cdev)
            dist.all_reduce(cam_part)
            diag[3 * nelim_local:] = cam_part.to(dev)
        tD = torch.sqrt(torch.clamp(diag, 1e-6, 1e32) / RADIUS)
        del diag, e2, f2, pt_cols, cam_cols
    else:
        # Schur ordering (points then cameras) serves both solvers and is what sharding needs.
        prob = pkg.problems.synthetic_bal(args.workload, layout="schur", seed=38401, skew=args.skew)
        if args.values == "scene":
            prob = pkg.problems.scene_values(prob, n_cams, n_points, seed=38401)
        nelim = prob.num_eliminate_blocks
        if world > 1:
            from ceres_solver_amd import partition
            sh = partition.shard_by_point(prob.bs, nelim, world, rank)
            bs, values, b, D, nelim_local = sh.bs, sh.local_values(prob.values), sh.local_rows(prob.b), sh.local_cols(prob.D), sh.num_eliminate_blocks
        else:
            bs, values, b, D, nelim_local = prob.bs, prob.values, prob.b, prob.D, nelim
        tv, tb, tD = (torch.from_numpy(a).to(dev) for a in (values, b, D))
    solver = make_solver(hs, bs, nelim_local, args.solver, local_rank, comm, storage)
    info = solver.info()
    tx = torch.full((bs.num_cols,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    # ---- timed region: K LM linear solves, inputs resident in HBM -------------------
    elapsed, iters, last = timed_steps(solver, (tv, tb, tD, tx), args.steps, args.warmup, sync, args.step, args.eta)
    collectives_last_step = int(solver.info().collectives_last_step)   # (of the last timed step: read before any other leg runs a solve)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ok = bool(torch.isfinite(tx).all().item())
    timing = phase_timing(solver, (tv, tb, tD, tx), args.step, args.eta)   # (one more step of the same kind, untimed, with the phase events on)

    # ---- dominant kernel against the HBM roofline (HIP events on the solver's stream) ----
    kind = "jtjx" if args.solver == "cgnr" else "sx"
    scalar_bytes = 4 if storage else 8
    my_obs = int(info.num_observations)
    my_points = int(info.num_e_blocks)

    def kernel_name(k):
        mode = "kJtJx" if k == "jtjx" else "kSx"
        # (the pipelined kernel runs unless an A/B switch says otherwise — kernels_bal.inc: UsePipeline, F32Pipelined)
        pipelined = os.environ.get("CERES_HIP_PIPELINE", "1") != "0" and not (storage and os.environ.get("CERES_HIP_F32_PIPELINE", "1") == "0")
        kern = "bal_stream_kernel" if pipelined else "bal_fused_kernel"
        body = f"{kern}<{mode}, fp32 tiles>" if storage else f"{kern}<{mode}>"
        if info.camera_accum_in_lds:
            return body + " + bal_reduce_partials_kernel"
        return body + " (per-slot F'z, cameras do not fit in LDS) + bal_camera_chunk_kernel + bal_reduce_partials_kernel"

    def measure_operator(slv, k):
        slv.load_device(tv.data_ptr(), tb.data_ptr(), tD.data_ptr())
        ms = slv.time_op(hs.TIMED_JTJX if k == "jtjx" else hs.TIMED_SX, args.kernel_iters)
        nbytes = algorithmic_bytes(k, my_obs, my_points, n_cams, scalar_bytes)
        return ms, nbytes, nbytes / (ms * 1e-3) / 1e9

    pmc_live = {}   # filled once (both operators come out of the same two counter passes)

    def pmc_traffic(k):
        # HBM bytes per application from rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected and corrected as
        # MI355X_MICROARCH.md prescribes).  LIVE (default at N = 1 on the fp64 tiles): tools/pmc_live.py runs the two passes now, in
        # child processes, around tools/kernel_times.py <workload> --operators-only — the same operators on the same generated
        # problem.  Otherwise, and whenever the live passes fail (no rocprofv3, this process itself under a profiler, a time-out), the
        # constant of an EARLIER run of the same passes: profiles/pmc_traffic.json names its files.
        f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        key = f"{args.workload}:{k}" + (":fp32" if storage else "")
        try:
            rec = json.load(open(f))
            committed, committed_src = rec.get(key), rec.get("source", {}).get(key, rec.get("_source"))
        except Exception:
            committed, committed_src = None, None
        if args.pmc_live and world == 1 and not storage:
            if not pmc_live:
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import pmc_live as pl
                    if pl.under_a_profiler():
                        raise RuntimeError("this process already runs under a profiler")
                    pmc_live.update(pl.measure(args.workload, timeout=args.pmc_live_timeout,
                                               env_extra={"CERES_HIP_PROBLEM_CACHE": os.environ.get("CERES_HIP_PROBLEM_CACHE", "/tmp/ceres_problem_cache")}))
                except Exception as ex:
                    pmc_live["error"] = repr(ex)[:300]
            if k in pmc_live:
                return pmc_live[k], (f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes run by this invocation (tools/pmc_live.py, "
                                     f"{pmc_live.get('seconds')} s; 1024 (2 FETCH_SIZE + WRITE_SIZE) over {sorted(pmc_live['breakdown_KiB'][k])}); "
                                     f"the committed constant of an earlier run (profiles/pmc_traffic.json): {committed}")
            committed_src = f"{committed_src}; live passes failed: {pmc_live.get('error')}"
        return committed, committed_src

    op_ms, alg_bytes, achieved = measure_operator(solver, kind)
    extra = {"pack_ms": solver.time_op(hs.TIMED_PACK, 10),
             # once per block structure (Ceres keeps one for a whole Solver::Solve): analysis, the tile plan, camera-major lists on the
             # host + their upload; never inside the timed steps
             "set_structure_s": round(getattr(solver, "set_structure_seconds", float("nan")), 3)}
    if not storage:
        copy_ms = solver.time_op(hs.TIMED_COPY, 10)
        extra["device_copy_GBs"] = round(2 * 8 * min(int(info.num_nonzeros), int(info.num_tiles) * 64 * 24) / (copy_ms * 1e-3) / 1e9, 1)
        extra["read_stream_probe_GBs"] = round(int(info.num_tiles) * 12288 / (solver.time_op(hs.TIMED_READ_STREAM, 10) * 1e-3) / 1e9, 1)
    if dist is not None:
        t = torch.tensor([achieved], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # aggregate GB/s over ranks, each on its shard
        achieved = float(t.item())
    traffic, traffic_source = pmc_traffic(kind) if world == 1 else (None, None)
    roofline = {"bound": "hbm", "kernel": kernel_name(kind),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "frac": round(achieved / (HBM_PEAK_GBS * world), 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(op_ms, 5)}

    # ---- the other solver, for the record (N = 1) -------------------------------------
    roofline_other = None
    if world == 1 and args.both_solvers:
        other = "iterative_schur" if args.solver == "cgnr" else "cgnr"
        s2 = make_solver(hs, bs, nelim_local, other, local_rank, None, storage)
        tx2 = torch.empty_like(tx)  # keep the primary solver's step in tx for the parity check below
        n2 = max(3, args.steps // 4)
        e2, it2, _ = timed_steps(s2, (tv, tb, tD, tx2), n2, 1, sync, args.step, args.eta)
        k2 = "sx" if other == "iterative_schur" else "jtjx"
        ms2, nb2, gb2 = measure_operator(s2, k2)
        tr2, src2 = pmc_traffic(k2)
        roofline_other = {"bound": "hbm", "kernel": kernel_name(k2),
                          "achieved": round(gb2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb2 / HBM_PEAK_GBS, 4),
                          "traffic": tr2, "traffic_source": src2, "algorithmic_bytes_per_launch": nb2, "avg_launch_ms": round(ms2, 5)}
        extra[other] = {"steps_per_s": round(n2 / e2, 3), "ms_per_step": round(1e3 * e2 / n2, 4), "cg_iterations": it2[-1],
                        f"{k2}_GBs": round(gb2, 1), f"{k2}_frac_hbm": round(gb2 / HBM_PEAK_GBS, 4), f"{k2}_ms": round(ms2, 5)}
        if other == "iterative_schur":
            extra[other]["schur_init_ms"] = round(s2.time_op(hs.TIMED_SCHUR_INIT, 10), 4)
            extra[other]["schur_jacobi_ms"] = round(s2.time_op(hs.TIMED_SCHUR_JACOBI, 10), 4)
        else:
            extra[other]["cgnr_setup_ms"] = round(s2.time_op(hs.TIMED_CGNR_SETUP, 10), 4)
        s2.close()
        del tx2
    if args.solver == "iterative_schur":
        extra["schur_init_ms"] = round(solver.time_op(hs.TIMED_SCHUR_INIT, 10), 4)
        extra["schur_jacobi_ms"] = round(solver.time_op(hs.TIMED_SCHUR_JACOBI, 10), 4)
        extra["back_substitute_ms"] = round(solver.time_op(hs.TIMED_BACK_SUBSTITUTE, 10), 4)

    # ---- through the drop-in boundary itself: host pointers in, host step out (SURVEY §8d "GPU timing") ----
    host_boundary = None
    if world == 1 and args.host_boundary_steps > 0 and not many_cameras:
        # what a Ceres process pays per LM step when the Evaluator wrote the Jacobian into pinned host memory
        # (BlockSparseMatrix(use_page_locked_memory = true), internal/ceres/block_jacobian_writer.cc:261-262):
        # H2D of values + residuals, the step, D2H of the step.  Never `value`.
        hv = torch.from_numpy(prob.values).pin_memory()
        hb = torch.from_numpy(prob.b).pin_memory()
        nh = args.host_boundary_steps
        solver.set_phase_timing(True)   # upload_ms / download_ms below come from the phase events (6 us each: nothing against the PCIe copies)
        solver.lm_compute_step(hv.numpy(), hb.numpy(), RADIUS, 0.1)
        per_step = []
        for _ in range(nh):
            t0 = time.perf_counter()
            _, sh_, _ = solver.lm_compute_step(hv.numpy(), hb.numpy(), RADIUS, 0.1)
            per_step.append(time.perf_counter() - t0)
        th = float(np.median(per_step))  # the host side (page faults of the freshly allocated step vector) jitters by milliseconds
        tm = solver.last_timing()
        # the retry after a REJECTED step (values_unchanged = 1: the minimizer has not re-evaluated, trust_region_minimizer.cc:832-837):
        # nothing goes up, the step's first pass reads the resident tiles; radius halved as StepRejected does
        retry = []
        h_step = torch.empty(bs.num_cols, dtype=torch.float64).pin_memory()   # the caller's step vector (Ceres allocates it once per Solve())
        h_stream_step = torch.empty(bs.num_cols, dtype=torch.float64).pin_memory()
        for k in range(nh):
            t0 = time.perf_counter()
            solver.lm_compute_step(None, None, RADIUS / 2.0, 0.1, reuse_diagonal=True, values_unchanged=True, out=h_step.numpy())
            retry.append(time.perf_counter() - t0)
        tr_ = solver.last_timing()
        solver.set_phase_timing(False)
        dev_retry_ms = None
        if nh > 1:   # the same retry with J and f resident (the device evaluator's case)
            tx_retry = torch.empty_like(tx)   # (its own step vector: tx keeps the step of the last TIMED solve for the parity figures below)
            solver.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx_retry.data_ptr(), RADIUS, 0.1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(nh):
                solver.lm_compute_step_device(tv.data_ptr(), tb.data_ptr(), tx_retry.data_ptr(), RADIUS / 2.0, 0.1, reuse_diagonal=True, values_unchanged=True)
            torch.cuda.synchronize()
            dev_retry_ms = 1e3 * (time.perf_counter() - t0) / nh
            del tx_retry
        # the upload HIDDEN behind the evaluator (ceres_hip_values_begin / _ready / _end): eight threads stand in for
        # ProgramEvaluator::Evaluate's parallel loop — each "evaluates" runs of row blocks by copying them from a source Jacobian into the
        # pinned arrays (the fastest evaluator there can be: a real one spends a microsecond per residual) and announces every run; what the
        # minimizer then waits for is the time from the LAST announcement to the step on the host
        streamed = None
        try:
            import threading
            from concurrent.futures import ThreadPoolExecutor
            src_v, src_b = prob.values, prob.b
            n_rows_b = bs.num_row_blocks
            e0 = bs.cell_value_pos[bs.row_cell_ptr[:-1]].astype(np.int64)           # first cell of every row (E|F split: the E stream)
            f0 = bs.cell_value_pos[bs.row_cell_ptr[:-1] + 1].astype(np.int64)       # its second cell (the F stream)
            n_chunks, n_thr = 256, 8
            bounds = np.linspace(0, n_rows_b, n_chunks + 1).astype(np.int64)
            runs_ms = []
            for rep in range(3):
                hv.zero_(); hb.zero_()
                hvn, hbn = hv.numpy(), hb.numpy()
                t_last = [0.0]
                lock = threading.Lock()

                def evaluate(k):
                    r0, r1 = int(bounds[k]), int(bounds[k + 1])
                    hvn[e0[r0]:e0[r1 - 1] + 6] = src_v[e0[r0]:e0[r1 - 1] + 6]
                    hvn[f0[r0]:f0[r1 - 1] + 18] = src_v[f0[r0]:f0[r1 - 1] + 18]
                    hbn[2 * r0:2 * r1] = src_b[2 * r0:2 * r1]
                    solver.values_ready(r0, r1 - r0)
                    t = time.perf_counter()
                    with lock:
                        t_last[0] = max(t_last[0], t)
                t_begin = time.perf_counter()
                solver.values_begin(hvn, hbn)
                with ThreadPoolExecutor(n_thr) as ex:
                    list(ex.map(evaluate, range(n_chunks)))
                solver.values_end(None)
                _, s_st, _ = solver.lm_compute_step(None, None, RADIUS, 0.1, values_unchanged=True, out=h_stream_step.numpy())
                t_done = time.perf_counter()
                runs_ms.append((1e3 * (t_done - t_last[0]), 1e3 * (t_last[0] - t_begin), 1e3 * (t_done - t_begin), int(s_st.num_iterations)))
            best = min(runs_ms)
            early_b, late_b, n_streams = solver.stream_stats()
            ref_step, _, _ = solver.lm_compute_step(hv.numpy(), hb.numpy(), RADIUS, 0.1)
            streamed = {"what": "ceres_hip_values_begin / _ready / _end: 8 threads copy 256 runs of row blocks from a source Jacobian into the pinned arrays ('evaluation' "
                                "at memcpy speed) and announce each run; then ceres_hip_lm_compute_step with values_unchanged = 1 (D2H of the step included)",
                        "ms_after_last_push": round(best[0], 3), "evaluator_ms": round(best[1], 3), "ms_begin_to_step": round(best[2], 3),
                        "ms_after_last_push_each_run": [round(r[0], 3) for r in runs_ms], "cg_iterations": best[3],
                        "value_streams": n_streams, "bytes_sent_before_end": early_b, "bytes_sent_in_end": late_b,
                        "step_rel_diff_vs_plain_step": float(np.linalg.norm(h_stream_step.numpy() - ref_step) / np.linalg.norm(ref_step))}
        except Exception as ex:  # the default line must not depend on it
            streamed = {"error": repr(ex)[:400]}
        host_boundary = {"steps_per_s": round(1.0 / th, 3), "ms_per_step": round(1e3 * th, 3), "upload_ms": round(tm.upload_ms, 3),
                         "streamed": streamed,
                         "download_ms": round(tm.download_ms, 3), "bytes_h2d": int(8 * (prob.values.shape[0] + prob.b.shape[0])),
                         "h2d_GBs": round(8 * (prob.values.shape[0] + prob.b.shape[0]) / max(tm.upload_ms, 1e-9) / 1e6, 1),
                         "what": "ceres_hip_lm_compute_step with pinned host values/residuals (PCIe H2D + step + D2H), median of " + str(nh) + " steps",
                         "ms_each_step": [round(1e3 * t, 2) for t in per_step],
                         "retry_after_rejection": {
                             "what": "ceres_hip_lm_compute_step with values_unchanged = 1, reuse_diagonal = 1, radius / 2: the Jacobian the solver "
                                     "already holds is neither re-sent nor re-laid-out (D2H of the step included)",
                             "ms_per_step": round(1e3 * float(np.median(retry)), 3), "upload_ms": round(tr_.upload_ms, 3),
                             "device_pointer_retry_ms_per_step": None if dev_retry_ms is None else round(dev_retry_ms, 3)}}
        del hv, hb, h_step, h_stream_step

    # ---- the whole trust-region loop on the device, for the record (N = 1) ----------
    scene_tr = None
    scene_step = None
    if world == 1 and args.minimizer_iterations > 0 and not many_cameras and not storage:
        free_b, _ = torch.cuda.mem_get_info()
        if free_b > 40 * n_obs * 24:  # its own Jacobian, tiles and vectors next to the bench's
            nc_, np_, cam_i, pt_i, obs_, par_ = pkg.problems.bal_scene(args.workload, seed=38401, skew=args.skew)
            typ, pre = (hs.CGNR, hs.JACOBI) if args.solver == "cgnr" else (hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI)
            bp = hs.BalProblem(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                      device=local_rank), nc_, np_, cam_i, pt_i, obs_)
            x0 = bp.state_from_bal(par_)
            bp.minimize(x0, max_num_iterations=1)  # warm-up
            _, Sm = bp.minimize(x0, max_num_iterations=args.minimizer_iterations)
            nit = Sm.num_successful_steps + Sm.num_unsuccessful_steps
            # The N(0,1) Jacobian of `value` is very well conditioned (2 CG iterations per step); a scene-valued problem
            # needs more.  This is the whole LM loop (evaluation included) on such a scene, first-class on the line.
            scene_tr = {
                "what": "ceres_hip_bal_minimize: Snavely evaluator (analytic Jacobian) + LM step + candidate evaluation per iteration, "
                        f"{args.workload}-shaped synthetic scene, state and Jacobian resident in HBM",
                "lm_iterations": nit, "lm_iterations_per_s": round(nit / Sm.total_seconds, 3) if Sm.total_seconds > 0 else None,
                "ms_per_lm_iteration": round(1e3 * Sm.total_seconds / max(nit, 1), 3),
                "linear_solves_per_s": round(Sm.num_linear_solves / Sm.linear_solver_seconds, 3) if Sm.linear_solver_seconds > 0 else None,
                "linear_solver_ms_per_iteration": round(1e3 * Sm.linear_solver_seconds / max(Sm.num_linear_solves, 1), 3),
                "evaluation_ms_total": round(1e3 * Sm.evaluation_seconds, 3),
                "initial_cost": Sm.initial_cost, "final_cost": Sm.final_cost, "successful_steps": Sm.num_successful_steps,
                "cg_iterations": [Sm.iterations[i].linear_solver_iterations for i in range(1, Sm.num_iterations_logged)],
                "termination": Sm.message.decode(errors="replace")}
            # scene_step: the LM step of `value` (same entry point, same solver instance) on the Jacobian the device evaluator
            # produces for this scene at its start point, Jacobi-scaled like TrustRegionMinimizer (trust_region_minimizer.cc:263-279),
            # at eta = 0.1 (Solver::Options default) and 0.01 (bundle_adjuster's): a step whose time is mostly CG, i.e. the hot kernel
            if args.scene_step_steps > 0 and np.array_equal(bp.row_order(), np.arange(n_obs, dtype=np.int32)):
                _, res_s, _, vals_s = bp.evaluate(x0, residuals=True, jacobian=True)
                tvs, tbs = torch.from_numpy(vals_s).to(dev), torch.from_numpy(res_s).to(dev)
                del vals_s, res_s
                pt_cols = torch.from_numpy(bs.col_block_pos[prob.point_of_row].astype(np.int64)).to(dev)
                cam_cols = torch.from_numpy(bs.col_block_pos[prob.camera_of_row].astype(np.int64)).to(dev)
                txs = torch.empty_like(tx)
                E, F = tvs[: 6 * n_obs].view(n_obs, 2, 3), tvs[6 * n_obs:].view(n_obs, 2, 9)
                cn = torch.zeros(bs.num_cols, dtype=torch.float64, device=dev)
                for c in range(3):
                    cn.index_add_(0, pt_cols + c, (E[:, :, c] ** 2).sum(1))
                for c in range(9):
                    cn.index_add_(0, cam_cols + c, (F[:, :, c] ** 2).sum(1))
                scale = 1.0 / (1.0 + torch.sqrt(cn))
                for c in range(3):
                    E[:, :, c] *= scale[pt_cols + c][:, None]
                for c in range(9):
                    F[:, :, c] *= scale[cam_cols + c][:, None]
                del pt_cols, cam_cols, cn, scale
                scene_step = {"what": f"ceres_hip_lm_compute_step_device on the Jacobi-scaled Snavely Jacobian of the {args.workload}-shaped synthetic scene "
                                      "(values and residuals from the device evaluator at the start point), inputs resident in HBM"}
                for eta_s in (0.1, 0.01):
                    es, its, _ = timed_steps(solver, (tvs, tbs, None, txs), args.scene_step_steps, 2, sync, "lm_step", eta_s)
                    tms = phase_timing(solver, (tvs, tbs, None, txs), "lm_step", eta_s)
                    k_it = int(its[-1])
                    mb = step_min_bytes(args.solver, n_obs, n_points, n_cams, k_it)
                    scene_step[f"eta_{eta_s}"] = {
                        "ms_per_step": round(1e3 * es / args.scene_step_steps, 4), "steps_per_s": round(args.scene_step_steps / es, 3),
                        "cg_iterations": k_it, "cg_ms": round(tms.cg_ms, 4), "cg_share_of_step": round(tms.cg_ms / max(tms.total_ms, 1e-9), 3),
                        "step_roofline_frac": round(mb / (es / args.scene_step_steps) / 1e9 / HBM_PEAK_GBS, 4)}
                del tvs, tbs, txs
            bp.close()
            # ---- a step in which the hot kernel DOMINATES: the same camera model on a SEQUENCE — every point seen by
            # a run of consecutive cameras (problems.banded_bal) — whose reduced system is as badly conditioned as video-like scenes are:
            # tens of CG iterations at eta = 1e-3 / 1e-4, where the randomly connected scene above stops after 2-7.  (Jacobi scaling does
            # not change the count: with the LM diagonal the scaled system is the unscaled one under a diagonal congruence, and
            # SCHUR_JACOBI-preconditioned CG is invariant under it.)
            if args.conditioned_steps > 0:
                try:
                    nc2, np2, cam2, pt2, obs2, par2 = pkg.problems.bal_scene(args.workload, seed=38401, visibility="banded")
                    bp2 = hs.BalProblem(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500,
                                                               device=local_rank), nc2, np2, cam2, pt2, obs2)
                    x02 = bp2.state_from_bal(par2)
                    _, res2, _, vals2 = bp2.evaluate(x02, residuals=True, jacobian=True)
                    bp2.close()
                    pb2 = pkg.problems.banded_bal(args.workload, seed=38401, with_values=False)
                    tv2, tb2 = torch.from_numpy(vals2).to(dev), torch.from_numpy(res2).to(dev)
                    del vals2, res2
                    s2c = make_solver(hs, pb2.bs, pb2.num_eliminate_blocks, args.solver, local_rank, None, 0)
                    tx2c = torch.empty(pb2.bs.num_cols, dtype=torch.float64, device=dev)
                    conditioned = {"what": f"ceres_hip_lm_compute_step_device on the Snavely Jacobian of a {args.workload}-shaped synthetic SEQUENCE (every point seen by "
                                           "consecutive cameras: a banded camera graph; problems.banded_bal), values and residuals from the device evaluator, "
                                           "inputs resident in HBM, radius 1e4"}
                    for eta_c in (1e-2, 1e-3, 1e-4):
                        es, its, last_c = timed_steps(s2c, (tv2, tb2, None, tx2c), args.conditioned_steps, 1, sync, "lm_step", eta_c)
                        tms = phase_timing(s2c, (tv2, tb2, None, tx2c), "lm_step", eta_c)
                        k_it = int(its[-1])
                        mb = step_min_bytes(args.solver, n_obs, n_points, n_cams, k_it)
                        conditioned[f"eta_{eta_c:g}"] = {
                            "ms_per_step": round(1e3 * es / args.conditioned_steps, 4), "cg_iterations": k_it, "cg_ms": round(tms.cg_ms, 4),
                            "cg_share_of_step": round(tms.cg_ms / max(tms.total_ms, 1e-9), 3), "ms_per_cg_iteration": round(tms.cg_ms / max(k_it, 1), 4),
                            "termination": hs.TERMINATION_NAMES[last_c.termination_type],
                            "step_roofline_frac": round(mb / (es / args.conditioned_steps) / 1e9 / HBM_PEAK_GBS, 4)}
                    s2c.close()
                    del tv2, tb2, tx2c
                    extra["conditioned_step"] = conditioned
                except Exception as ex:  # the default line must not depend on it
                    extra["conditioned_step"] = {"error": repr(ex)[:400]}
    # ---- BASELINE.json configs[1] and configs[2] on the driver line: the same step on the two smaller shapes ----
    if world == 1 and args.extra_configs and args.workload == "venice1778" and not storage:
        try:
            extra["configs"] = {"what": "BASELINE.json configs[1] (BAL 16-22106, CGNR + JACOBI) and configs[2] (BAL Ladybug 1723-156502, ITERATIVE_SCHUR + "
                                        "SCHUR_JACOBI) as synthetic Jacobians of those block counts: the same LM step through the same entry point, inputs "
                                        "resident in HBM; operator = the step's dominant kernel(s) by HIP events; the oracle on the same inputs beside it",
                                "cases": [config_leg(pkg, hs, entry, wl_c, sv_c, local_rank, dev, args, no_cpu=args.no_cpu_baseline)
                                          for wl_c, sv_c in (("dubrovnik16", "cgnr"), ("ladybug1723", "iterative_schur"))]}
        except Exception as ex:  # the default line must not depend on it
            extra["configs"] = {"error": repr(ex)[:400]}

    # ---- strong-scaling ceiling measured on ONE GPU: rank 0's shard with the sharded code path on ----
    if world == 1 and args.shard_ceiling and not storage and info.kernel_path == hs.PATH_BAL:
        try:
            t1_ms = 1e3 * elapsed / args.steps
            if many_cameras:
                extra["shard_ceiling"] = shard_ceiling(pkg, hs, None, args.solver, local_rank, t1_ms, int(iters[-1]), args.eta, worlds=(8,), steps=max(3, args.steps),
                                                       warmup=1, dev=dev, many_cameras=args.workload)
            else:
                extra["shard_ceiling"] = shard_ceiling(pkg, hs, prob, args.solver, local_rank, t1_ms, int(iters[-1]), args.eta, worlds=(2, 4, 8), steps=args.steps,
                                                       warmup=args.warmup, dev=dev)
        except Exception as ex:  # the default line must not depend on it
            extra["shard_ceiling"] = {"error": repr(ex)[:400]}
    extra["solve_phases_ms"] = {k: round(getattr(timing, k), 4) for k in
                                ("pack_ms", "setup_ms", "preconditioner_ms", "cg_ms", "back_substitute_ms", "total_ms")}
    extra["operator_launches_enqueued_last_step"] = int(timing.operator_applications)
    extra["device_bytes"] = int(info.device_bytes)

    # ---- many-camera workloads: the fp32-tile storage mode next to fp64 (BASELINE.json configs[4]: "fp32 and fp64") ----
    if world == 1 and (args.also_fp32 > 0 or (args.also_fp32 < 0 and args.workload == "venice1778" and args.extra_synthetic10m)) and not storage and info.kernel_path == hs.PATH_BAL:
        s32 = make_solver(hs, bs, nelim_local, args.solver, local_rank, None, 1)
        tx32 = torch.empty_like(tx)
        n32 = max(3, args.steps // 2)
        e32, it32, _ = timed_steps(s32, (tv, tb, tD, tx32), n32, 1, sync, args.step, args.eta)
        s32.load_device(tv.data_ptr(), tb.data_ptr(), tD.data_ptr())
        ms32 = s32.time_op(hs.TIMED_JTJX if kind == "jtjx" else hs.TIMED_SX, args.kernel_iters)
        nb32 = algorithmic_bytes(kind, my_obs, my_points, n_cams, 4)
        extra["fp32_tiles"] = {"what": "Jacobian tiles rounded to fp32, fp64 arithmetic (BASELINE.json configs[4] asks for both precisions): an accuracy mode, never parity; "
                                       "through the software-pipelined kernels like the fp64 tiles (CERES_HIP_F32_PIPELINE=0: the unpipelined ones)",
                               "steps_per_s": round(n32 / e32, 3), "ms_per_step": round(1e3 * e32 / n32, 4), "cg_iterations": it32[-1],
                               f"{kind}_ms": round(ms32, 5), f"{kind}_frac_hbm_of_fp32_bytes": round(nb32 / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "step_rel_diff_vs_fp64": float((torch.linalg.norm(tx32 - tx) / torch.linalg.norm(tx)).item())}
        s32.close()
        del tx32

    # ---- BASELINE.json configs[4] on the DEFAULT line: synthetic10M in a child process (its own 20 GB of HBM), condensed ----
    if world == 1 and args.extra_synthetic10m and args.workload == "venice1778" and not storage:
        import subprocess
        t10 = time.perf_counter()
        free_b, _ = torch.cuda.mem_get_info()
        if free_b < 60e9:
            extra["synthetic10M"] = {"skipped": f"{free_b / 1e9:.0f} GB of HBM free, the child wants 60"}
        else:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "synthetic10M", "--steps", "5", "--warmup", "1", "--kernel-iters", "20",
                   "--no-cpu-baseline", "--oracle-check", "1", "--host-boundary-steps", "0", "--minimizer-iterations", "0", "--extra-synthetic10m", "0", "--pmc-live", "0", "--also-fp32", "1",
                   "--solver", args.solver, "--eta", str(args.eta)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                other = "cgnr" if args.solver == "iterative_schur" else "iterative_schur"
                ro, rj = d["roofline"], d.get("roofline_jtjx") or d.get("roofline_sx") or {}
                extra["synthetic10M"] = {
                    "what": "bench.py --workload synthetic10M in a child process: 10 M points x 3 observations, 50 000 cameras (camera accumulators do "
                            "not fit in LDS), fp64, values generated in HBM; same step, same entry point",
                    "workload": d["config"]["workload"], "steps_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                    "cg_iterations_per_step": d["config"]["cg_iterations_per_step"], "solver": d["config"]["solver"],
                    ("sx" if kind == "sx" else "jtjx"): {"frac": ro["frac"], "GBs": ro["achieved"], "ms": ro["avg_launch_ms"], "kernel": ro["kernel"]},
                    ("jtjx" if kind == "sx" else "sx"): {"frac": rj.get("frac"), "GBs": rj.get("achieved"), "ms": rj.get("avg_launch_ms")},
                    other: d["extra"].get(other), "fp32_tiles": d["extra"].get("fp32_tiles"), "shard_ceiling": d["extra"].get("shard_ceiling"),
                    "step_rel_diff_vs_oracle": (d.get("oracle_check") or {}).get("step_rel_diff_vs_oracle"), "oracle_check": d.get("oracle_check"),
                    "child_wall_s": round(time.perf_counter() - t10, 1)}
            except Exception as ex:  # the default line must not depend on the child
                extra["synthetic10M"] = {"error": repr(ex)[:300]}

    # ---- REAL visibility next to the synthetic shapes: the libmv problems the reference ships, replicated to size (tools/real_graph_times.py) ----
    if world == 1 and args.extra_real_graph and args.workload == "venice1778" and not storage:
        try:
            rg = {"what": "S.x / JtJx (HIP events, as roofline) on the visibility graph of data/libmv-ba-problems/problem_02.bin (tests/golden/libmv_problems.npz: 71 tracks "
                          "through 440 consecutive frames) replicated side by side, N(0,1) values: every point has far more than 64 observations (it owns whole tiles), "
                          "neighbouring cameras see the same points", "cases": []}
            for copies in (4, 120):
                rp = pkg.problems.libmv_bal(2, copies)
                r_np = rp.num_eliminate_blocks
                r_nc, r_no = rp.bs.num_col_blocks - r_np, rp.bs.num_row_blocks
                case = {"copies": copies, "cameras": r_nc, "points": r_np, "observations": r_no}
                for sv, kd in (("iterative_schur", "sx"), ("cgnr", "jtjx")):
                    rs = make_solver(hs, rp.bs, r_np, sv, local_rank, None, 0)
                    ri = rs.info()
                    rs.load(rp.values, rp.b, rp.D)
                    rms = min(rs.time_op(hs.TIMED_SX if kd == "sx" else hs.TIMED_JTJX, 20) for _ in range(3))
                    case[kd] = {"ms": round(rms, 5), "frac": round(algorithmic_bytes(kd, r_no, r_np, r_nc, 8) / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    case.update({"accumulators_in_lds": int(ri.camera_accum_in_lds), "hybrid": int(ri.camera_accum_hybrid),
                                 "observations_summed_in_lds": round(ri.num_observations_in_lds / float(r_no), 4)})
                    rs.close()
                rg["cases"].append(case)
            extra["real_graph"] = rg
        except Exception as ex:  # the default line must not depend on it
            extra["real_graph"] = {"error": repr(ex)[:300]}

    # ---- the many-camera regime on a graph that HAS locality: 50 000 cameras, every point seen by consecutive cameras ----
    if world == 1 and args.extra_banded and args.workload == "venice1778" and not storage:
        try:
            bc, bp, bo = pkg.problems.BAL_SHAPES["synthetic1M"]
            pb = pkg.problems.banded_bal(None, seed=38401, num_cameras=bc, num_points=bp, num_observations=bo, with_values=False)
            gb = torch.Generator(device=dev)
            gb.manual_seed(38401)
            vb = torch.randn(24 * bo, dtype=torch.float64, device=dev, generator=gb)
            bb = torch.randn(2 * bo, dtype=torch.float64, device=dev, generator=gb)
            Db = torch.rand(pb.bs.num_cols, dtype=torch.float64, device=dev, generator=gb) * 0.1 + 0.05
            case = {"what": "S.x / JtJx (HIP events, as roofline) on a SEQUENCE-like graph with synthetic1M's block counts: 50 000 cameras, 1 M points, 3 M observations, "
                            "every point seen by consecutive cameras (problems.banded_bal), N(0,1) values generated in HBM: the camera accumulators do not fit in LDS, "
                            "but a workgroup's tiles touch few cameras once the plan has grouped the points by camera window",
                    "cameras": bc, "points": bp, "observations": bo}
            for sv_, kd_ in (("iterative_schur", "sx"), ("cgnr", "jtjx")):
                sb_ = make_solver(hs, pb.bs, pb.num_eliminate_blocks, sv_, local_rank, None, 0)
                ib_ = sb_.info()
                sb_.load_device(vb.data_ptr(), bb.data_ptr(), Db.data_ptr())
                ms_ = min(sb_.time_op(hs.TIMED_SX if kd_ == "sx" else hs.TIMED_JTJX, 20) for _ in range(3))
                nb_ = algorithmic_bytes(kd_, bo, bp, bc, 8)
                case[kd_] = {"ms": round(ms_, 5), "frac": round(nb_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": nb_}
                try:   # HBM bytes by PMC of an EARLIER run of tools/kernel_times.py banded50k (profiles/pmc_traffic.json), like roofline.traffic
                    rec_ = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                    if rec_.get(f"banded50k:{kd_}"):
                        case[kd_].update({"traffic": rec_[f"banded50k:{kd_}"], "traffic_over_algorithmic": round(rec_[f"banded50k:{kd_}"] / nb_, 3),
                                          "traffic_source": rec_.get("source", {}).get(f"banded50k:{kd_}")})
                except Exception:
                    pass
                case.update({"accumulators_in_lds": int(ib_.camera_accum_in_lds), "hybrid": int(ib_.camera_accum_hybrid),
                             "observations_summed_in_lds": round(ib_.num_observations_in_lds / float(bo), 4)})
                sb_.close()
            extra["banded50k"] = case
            del vb, bb, Db
        except Exception as ex:  # the default line must not depend on it
            extra["banded50k"] = {"error": repr(ex)[:300]}

    # ---- the fused path beyond <2,3,9> and the generic path, on the default line ----
    if world == 1 and args.extra_other_shapes and args.workload == "venice1778" and not storage:
        try:
            lb_c, lb_p, lb_o = pkg.problems.BAL_SHAPES["ladybug1723"]
            shapes = {"what": "S.x / JtJx (HIP events) and one LM step against the oracle for structures other than <2,3,9>, Ladybug-1723 block counts "
                              "unless stated: bytes per application counted like SURVEY.md §8(d) with the slot's own size", "cases": []}
            oracle_s = None if args.no_cpu_baseline else entry.load_oracle()
            if oracle_s is not None:
                oracle_s.set_num_threads(min(os.cpu_count() or 1, 16))
            cases = [("<2,3,10> quaternion cameras (bundle_adjuster --use_quaternions)", dict(camera_width=10), False),
                     ("<2,3,6>", dict(camera_width=6), False),
                     ("<2,3,7> (a width the reference reaches through its dynamic (2,3,d) specialisation; round 6: every camera width 2 .. 10 is compiled)", dict(camera_width=7), False),
                     ("<2,3,5>", dict(camera_width=5), False),
                     ("<2,3,9> on the GENERIC kernels (force_generic_path)", dict(camera_width=9), True),
                     ("libmv structure <2, 8 | 6, 3>: shared intrinsics + 6-wide pose + point, first camera constant", dict(camera_width=6, shared_widths=(8,), locked_cameras=(0,)), False),
                     ("<2,4,9> homogeneous points (the reference's (2,4,9) specialisation; round 5: point blocks 2 and 4 wide on the fused path)", dict(camera_width=9, point_width=4), False),
                     ("<3,3,3> rows of three residuals (the reference's (3,3,3) specialisation; round 5: rows 3 and 4 high on the fused path)", dict(camera_width=3, point_width=3, row_height=3), False)]
            for label, kw, force_generic in cases:
                sp = pkg.problems.synthetic_structured(lb_c, lb_p, lb_o, seed=38401, skew=args.skew, **kw)
                nf_, ns_, pw_, rh_ = kw["camera_width"], sum(kw.get("shared_widths", ())), kw.get("point_width", 3), kw.get("row_height", 2)
                slot_b = rh_ * (pw_ + nf_ + ns_) * 8 + 8
                n_fs = int(sp.bs.col_block_size[sp.num_eliminate_blocks:].sum())
                case = {"structure": label, "bytes_per_observation": slot_b}
                for sv, kd, typ, pre in (("iterative_schur", "sx", hs.ITERATIVE_SCHUR, hs.SCHUR_JACOBI), ("cgnr", "jtjx", hs.CGNR, hs.JACOBI)):
                    if (pw_ != 3 or rh_ != 2) and sv == "cgnr":
                        continue   # (point blocks that are not 3 wide: the Schur solvers run fused, CGNR on the generic kernels)
                    so_ = hs.HipLinearSolver(hs.LinearSolverOptions(type=typ, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500, residual_reset_period=10,
                                                                    elimination_groups=[sp.num_eliminate_blocks], device=local_rank, force_generic_path=force_generic))
                    so_.set_structure(sp.bs)
                    case["kernel_path"] = "fused" if so_.info().kernel_path == hs.PATH_BAL else "generic"
                    so_.load(sp.values, sp.b, sp.D)
                    ms_ = min(so_.time_op(hs.TIMED_SX if kd == "sx" else hs.TIMED_JTJX, 20) for _ in range(3))
                    nb_ = lb_o * slot_b + (lb_p * pw_ * pw_ * 8 + n_fs * 32 if kd == "sx" else (pw_ * lb_p + n_fs) * 32)
                    case[kd] = {"ms": round(ms_, 5), "frac": round(nb_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    if sv == "iterative_schur":
                        so_.set_phase_timing(True)
                        step_, summ_, mcc_ = so_.lm_compute_step(sp.values, sp.b, RADIUS, args.eta)
                        case["lm_step"] = {"cg_iterations": summ_.num_iterations, "ms_device": round(so_.last_timing().total_ms - so_.last_timing().upload_ms - so_.last_timing().download_ms, 4)}
                        if oracle_s is not None:
                            mo_s = oracle_s.Matrix(sp.bs, sp.num_eliminate_blocks)
                            Ds = np.sqrt(np.clip(oracle_s.Matrix(sp.bs, 0).squared_column_norm(sp.values), 1e-6, 1e32) / RADIUS)
                            k_ = int(summ_.num_iterations)
                            xo_s, so_s = mo_s.iterative_schur_solve(sp.values, sp.b, Ds, preconditioner=2, min_it=k_, max_it=k_, q_tol=-1.0, r_tol=-1.0)
                            case["lm_step"]["step_rel_diff_vs_oracle_iterate_of_the_same_index"] = float(np.linalg.norm(step_ + xo_s) / np.linalg.norm(xo_s))
                    so_.close()
                shapes["cases"].append(case)
            if oracle_s is not None:
                oracle_s.set_num_threads(1)
            extra["other_shapes"] = shapes
        except Exception as ex:  # the default line must not depend on it
            extra["other_shapes"] = {"error": repr(ex)[:400]}

    # ---- DENSE_SCHUR's factorisation on the matrix pipe (the one real contraction of the path: SURVEY §8 f2) ----
    if world == 1 and args.extra_dense_cholesky and args.workload == "venice1778" and not storage:
        try:
            n_dc = 8190   # 910 cameras x 9
            rng_dc = np.random.default_rng(8190)
            Bm = rng_dc.standard_normal((n_dc, n_dc)) * 0.5
            Am = (Bm + Bm.T) / 2 + n_dc * 0.6 * np.eye(n_dc)
            xt = rng_dc.standard_normal(n_dc)
            xs, ms_dc, failed_dc = solver.dense_cholesky_solve(np.triu(Am), Am @ xt, repeats=3)
            tf = n_dc ** 3 / 3.0 / ms_dc / 1e9
            extra["dense_schur_cholesky"] = {
                "what": "DenseCholesky::FactorAndSolve of a random SPD matrix the size of a 910-camera reduced system (blocked, 128-wide panels, "
                        "trailing update on v_mfma_f64_4x4x4_4b_f64 beside the potrf / trsm panel chain, which is what the factorisation waits "
                        "for; csrc/kernels_schur.hip, design/11_round4.md 11.6)", "n": n_dc, "factor_ms": round(ms_dc, 3),
                "TFLOPs": round(tf, 2), "peak_TFLOPs_datasheet_fp64_matrix": 78.6, "frac_of_datasheet_peak": round(tf / 78.6, 4),
                "mfma_f64_sustained_TFLOPs_probe": 77.0, "frac_of_probe": round(tf / 77.0, 4),
                "probe": "tools/probes/mfma_f64_probe.hip, profiles/r04m_mfma_f64_probe.txt: v_mfma_f64_4x4x4_4b_f64 75-77.8 TFLOP/s (the 16x16x4 form: 35)",
                "failed": bool(failed_dc), "rel_err_of_solve": float(np.linalg.norm(xs - xt) / np.linalg.norm(xt))}
            del Bm, Am
        except Exception as ex:
            extra["dense_schur_cholesky"] = {"error": repr(ex)[:300]}

    # ---- full-size parity of the timed step itself (rungs 4 / 5 of the ladder at the size the line is quoted on) ----
    oracle_check = None
    if (args.oracle_check > 0 or (args.oracle_check < 0 and not many_cameras)) and not (many_cameras and world > 1) and not storage:
        xl = tx.cpu().numpy()
        if world > 1:   # every rank's shard of the step -> rank 0, assembled through the shards' column maps
            parts = [None] * world
            dist.gather_object((xl, sh.col_index, int(bs.col_block_size[:nelim_local].sum())), parts if rank == 0 else None, dst=0)
            if rank == 0:
                xg_full = np.full(prob.bs.num_cols, np.nan)
                for xl_r, ci_r, ne_r in parts:
                    xg_full[ci_r[:ne_r]] = xl_r[:ne_r]
                xg_full[parts[0][1][parts[0][2]:]] = parts[0][0][parts[0][2]:]
        else:
            xg_full = xl
        if rank == 0:
            oracle = entry.load_oracle()
            oracle.set_num_threads(min(os.cpu_count() or 1, 16))
            if many_cameras:
                o_values, o_b = tv.cpu().numpy(), tb.cpu().numpy()
            else:
                o_values, o_b = prob.values, prob.b
            o_nelim = prob.num_eliminate_blocks
            mo_ = oracle.Matrix(prob.bs, o_nelim if args.solver == "iterative_schur" else 0)
            mo_all = oracle.Matrix(prob.bs, 0)
            fn_o = mo_.iterative_schur_solve if args.solver == "iterative_schur" else mo_.cgnr_solve
            pre_o = 2 if args.solver == "iterative_schur" else 1
            t_o = time.perf_counter()
            if args.step == "lm_step":
                D_o = np.sqrt(np.clip(mo_all.squared_column_norm(o_values), 1e-6, 1e32) / RADIUS)
            else:
                D_o = tD.cpu().numpy() if many_cameras else prob.D
            xo_, so_ = fn_o(o_values, o_b, D_o, preconditioner=pre_o, min_it=0, max_it=500, q_tol=args.eta, r_tol=-1.0)
            t_o = time.perf_counter() - t_o
            k_gpu, k_or = int(iters[-1]), int(so_.num_iterations)
            same_index = True
            if k_gpu != k_or:   # zeta crossed the threshold one index apart: compare with the oracle's iterate of the SAME index (tests/step_check.py)
                xo_, _ = fn_o(o_values, o_b, D_o, preconditioner=pre_o, min_it=k_gpu, max_it=k_gpu, q_tol=-1.0, r_tol=-1.0)
                same_index = False
            sign = -1.0 if args.step == "lm_step" else 1.0
            oracle_check = {"what": "the step of the last timed solve against ONE oracle step on the same full-size inputs "
                                    "(the oracle's CG iterate of the same iteration number; tolerance of the parity ladder: 1e-9)",
                            "step_rel_diff_vs_oracle": float(np.linalg.norm(xg_full - sign * xo_) / np.linalg.norm(xo_)),
                            "cg_iterations_gpu": k_gpu, "cg_iterations_oracle": k_or, "oracle_rerun_at_gpu_iteration_count": not same_index,
                            "oracle_seconds": round(t_o, 2), "oracle_threads": min(os.cpu_count() or 1, 16), "ranks": world,
                            "observations": int(prob.bs.num_row_blocks)}
            oracle.set_num_threads(1)
            del xo_, mo_, mo_all

    # ---- CPU baseline: the oracle (a restatement of Ceres' algorithm, "port") on this box's cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        oracle = entry.load_oracle()
        ncpu = os.cpu_count() or 1
        if many_cameras:
            h_values, h_b, h_D = tv.cpu().numpy(), tb.cpu().numpy(), tD.cpu().numpy()
        else:
            h_values, h_b, h_D = prob.values, prob.b, prob.D
        if storage:  # compare like with like: the oracle sees the fp32-rounded Jacobian the tiles hold
            h_values = h_values.astype(np.float32).astype(np.float64)
        m = oracle.Matrix(prob.bs, nelim if args.solver == "iterative_schur" else 0)
        fn = m.iterative_schur_solve if args.solver == "iterative_schur" else m.cgnr_solve
        pre = 2 if args.solver == "iterative_schur" else 1

        m_all = oracle.Matrix(prob.bs, 0)

        def one():
            t = time.perf_counter()
            if args.step == "lm_step":  # same work as the GPU step: diag(J'J), D, solve, negate, model cost
                diag = np.clip(m_all.squared_column_norm(h_values), 1e-6, 1e32)
                Dc = np.sqrt(diag / RADIUS)
            else:
                Dc = h_D
            xo_, so_ = fn(h_values, h_b, Dc, preconditioner=pre, min_it=0, max_it=500, q_tol=args.eta, r_tol=-1.0)
            if args.step == "lm_step":
                xo_ = -xo_
                model = m_all.right_multiply(h_values, xo_)
                _ = -model @ (h_b + model / 2.0)
            return time.perf_counter() - t, xo_, so_
        # memory-bound sparse kernels do not scale to every core of a big host: probe a few thread
        # counts with one solve each, then spend the rest of the budget on the fastest
        # (cheapest candidates first: on some boxes one step with every core takes 20+ s)
        probe = {}
        for th in dict.fromkeys([min(ncpu, 16), min(ncpu, 32), min(ncpu, 64), 1, ncpu]):
            if probe and sum(probe.values()) > args.cpu_seconds / 2:
                break
            oracle.set_num_threads(th)
            probe[th] = one()[0]
        cores = min(probe, key=probe.get)
        oracle.set_num_threads(cores)
        n_done, t0, cpu_iters = 0, time.perf_counter(), None
        while True:
            _, xo, so = one()
            n_done += 1
            cpu_iters = so.num_iterations
            if time.perf_counter() - t0 > args.cpu_seconds / 2 or n_done >= args.steps:
                break
        cpu_t = time.perf_counter() - t0
        xg = tx.cpu().numpy()
        parity = float(np.linalg.norm(xg - xo) / np.linalg.norm(xo)) if cpu_iters == iters[-1] else None
        probe_txt = ""
        try:  # tools/probe.sh: can real Ceres be built on this box (Eigen3 / abseil present)?  recorded, not required
            import subprocess
            r = subprocess.run(["bash", os.path.join(ROOT, "tools", "probe.sh"), "--brief"], capture_output=True, text=True, timeout=60)
            probe_txt = "; tools/probe.sh: " + r.stdout.strip().replace("\n", ", ")
        except Exception:
            pass
        cpu = {"value": round(n_done / cpu_t, 4), "unit": "steps/s", "cores": cores, "kind": "port",
               "sample": f"{n_done} full {args.workload}-shaped {args.solver} " + ("LM steps (diag, solve, model cost)" if args.step == "lm_step" else "solves") + f" (same inputs, eta={args.eta}), "
                         f"oracle/libceres_oracle.so with OpenMP over {cores} threads, {cpu_t:.1f} s; "
                         f"one-step probe seconds by thread count: { {k: round(v, 2) for k, v in probe.items()} } on {ncpu} host cpus" + probe_txt,
               "cg_iterations": cpu_iters, "step_rel_diff_vs_gpu": parity}
        oracle.set_num_threads(1)

    if rank == 0:
        value = args.steps / elapsed
        line = {
            "metric": "LM trust-region steps/sec (linear-solve hot path) + JtJx SpMV HBM GB/s on BAL",
            "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped synthetic BAL Jacobian <2,3,9>: {n_cams} cameras, {n_points} points, "
                                   f"{n_obs} observations, " + ("N(0,1) values" if args.values == "normal" else "values = Jacobi-scaled Snavely Jacobian "
                                   "of a synthetic scene (first LM step)") + f", seed 38401, camera popularity skew {args.skew}"
                                   + (", values generated in HBM (torch.randn), one generator per rank" if many_cameras else ""),
                       "solver": "CGNR + JACOBI" if args.solver == "cgnr" else "ITERATIVE_SCHUR + SCHUR_JACOBI",
                       "step": ("LevenbergMarquardtStrategy::ComputeStep (diag(J'J), D = sqrt(diag/1e4), Solve, finite check, negate) + "
                                "model cost change, all on the device" if args.step == "lm_step" else "LinearSolver::Solve"),
                       "eta": args.eta, "max_num_iterations": 500, "cg_iterations_per_step": iters[-1],
                       "termination": hs.TERMINATION_NAMES[last.termination_type],
                       "parallelism": (f"points sharded over {world} rank(s), camera-space sums by " +
                                       ("the one-shot peer-to-peer all-reduce (hipIpc / xGMI)" if solver.p2p_ok else "RCCL all-reduce") +
                                       (" — ONE-GPU VALIDATION MODE, timings meaningless" if one_gpu else ""))
                       if world > 1 else "1 GPU",
                       "collectives_per_step": collectives_last_step if world > 1 and args.step == "lm_step" else 0,
                       "inputs_resident_in_hbm": True, "step_finite": step_ok,
                       "jacobian_storage": "fp32 tiles, fp64 arithmetic (accuracy mode, not parity)" if storage else "fp64",
                       "kernel_path": "fused<2,3,9>" if info.kernel_path == hs.PATH_BAL else "generic",
                       "camera_accumulators_in_lds": bool(info.camera_accum_in_lds)},
            "roofline": roofline, "cpu_baseline": cpu, "oracle_check": oracle_check, "host_boundary": host_boundary, "scene_trust_region": scene_tr, "scene_step": scene_step,
            "extra": extra,
        }
        if args.step == "lm_step":  # the whole step against ITS roofline: minimum bytes of all its passes / measured time / peak
            tot_obs, tot_pts = (n_obs, n_points)
            mb = step_min_bytes(args.solver, tot_obs, tot_pts, n_cams, int(iters[-1]), scalar_bytes)
            line["step_roofline"] = {"bound": "hbm", "min_bytes_per_step": int(mb), "achieved": round(mb / (elapsed / args.steps) / 1e9, 1),
                                     "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(mb / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
                                     "what": "minimum HBM bytes of every pass of the step (J in the layout each pass wants; re-layout not counted) / ms_per_step"}
        if roofline_other is not None:
            line["roofline_jtjx" if args.solver == "iterative_schur" else "roofline_sx"] = roofline_other
        # what a Ceres process with a CPU evaluator gets through the drop-in boundary (PCIe H2D of J and f every accepted step): next to
        # `value`, which is the device-resident rate
        line["value_host_boundary"] = None if host_boundary is None else host_boundary["steps_per_s"]
        print(json.dumps(line), flush=True)
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
