"""Worker process of tests/test_gpu_multirank.py: ONE RANK of a sharded solve, product code only
(ceres-solver_amd through the C ABI).  Ranks may share a GPU: the one-shot peer-to-peer all-reduce maps its
peers' buffers with hipIpc, which works between processes on the same device as well as across xGMI.

Protocol with the parent over a multiprocessing connection: ("exchange", bytes) -> list of every rank's bytes;
("done", results) ends the rank; ("error", text) on failure."""
import os
import sys
import traceback

import numpy as np


def run_rank(rank, world, conn, device, scenario):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.setdefault("CERES_HIP_P2P_TIMEOUT", "20")
    # the ranks of these tests SHARE one device: kernels that wait for their peers must fit on it beside each other (solver.hip: p2p_grid_cap)
    os.environ.setdefault("CERES_HIP_P2P_SHARED_DEVICE", "1")
    try:
        import torch  # noqa: F401  (before the HIP library: see hip_solver.load_library)
        import __graft_entry__ as entry
        pkg = entry.load_package()
        hs = pkg.hip_solver
        from ceres_solver_amd import partition
        hs.load_library()

        def exchange(mine):
            conn.send(("exchange", mine))
            return conn.recv()

        out = {}
        for name, kw in scenario:
            os.environ["CERES_HIP_CG_FUSED"] = kw.get("cg_fused", "1")  # read when a solver is created
            os.environ["CERES_HIP_P2P_TIMEOUT"] = str(kw.get("p2p_timeout", 20))
            kind = kw["kind"]
            if kind == "fuzz":   # a random structure of the parity campaign (tests/fuzz_cases.py), built alike on every rank
                import fuzz_cases
                case, k, _ = fuzz_cases.draw_case(kw["seed"], kw.get("big", False))
                prob = fuzz_cases.build(pkg.problems, case, k)
            elif kind == "bal" and kw.get("structured"):   # camera widths other than 9, shared blocks, locked cameras (problems.synthetic_structured)
                prob = pkg.problems.synthetic_structured(kw["nc"], kw["np"], kw["no"], seed=kw["seed"], skew=kw.get("skew", 0.5), **kw["structured"])
            elif kind == "bal":
                prob = pkg.problems.synthetic_bal(None, layout="schur", seed=kw["seed"], skew=kw.get("skew", 0.5),
                                                  num_cameras=kw["nc"], num_points=kw["np"], num_observations=kw["no"])
                if kw.get("camera_rows", 0):  # rows without a point cell: partition.py hands them to the last rank
                    prob = pkg.problems.add_camera_rows(prob, kw["camera_rows"], seed=kw["seed"], pair_fraction=0.3)
            elif kind == "general_fuzz":   # tools/fuzz_multirank.py --generic: random E|F-partitioned structures, blocks 1 .. 4 wide
                prob = pkg.problems.random_schur_problem(**kw["problem"])
            else:
                prob = pkg.problems.random_schur_problem(num_e_blocks=kw["ne"], num_f_blocks=kw["nf"], num_no_e_rows=2, seed=kw["seed"])
            sh = partition.shard_by_point(prob.bs, prob.num_eliminate_blocks, world, rank)
            v, b, D = sh.local_values(prob.values), sh.local_rows(prob.b), sh.local_cols(prob.D)
            n_f_blocks = sh.bs.num_col_blocks - sh.num_eliminate_blocks
            fsz = sh.bs.col_block_size[sh.num_eliminate_blocks:].astype(np.int64)
            max_elems = int((fsz ** 2).sum() + 2 * fsz.sum() + 2)   # a step sums [blocks | rhs | column norms] in one all-reduce
            for vi, var in enumerate(kw.get("variants", ())):
                # the solver OPTIONS sharded (DENSE_SCHUR, the explicit Schur complement, the power-series preconditioner and initialisation,
                # JACOBI / IDENTITY): one Solve each, compared by the parent with a single-rank instance on the whole problem
                var = dict(var)
                q_tol, r_tol = var.pop("q_tolerance", -1.0), var.pop("r_tolerance", -1.0)
                o = hs.LinearSolverOptions(elimination_groups=[sh.num_eliminate_blocks], device=device, **var)
                s = hs.HipLinearSolver(o, rank=rank, world_size=world, p2p_exchange=exchange, p2p_max_elements=max_elems)
                try:
                    s.set_structure(sh.bs)
                    x, summ = s.solve(v, b, hs.PerSolveOptions(D=D, q_tolerance=q_tol, r_tolerance=r_tol))
                    out[(name, "variant", vi)] = {"x": (x, summ.termination_type, summ.num_iterations, None, summ.message), "col_index": sh.col_index,
                                                  "n_e": int(sh.bs.col_block_size[: sh.num_eliminate_blocks].sum()), "path": int(s.info().kernel_path)}
                except hs.HipError as ex:
                    out[(name, "variant", vi)] = {"error": str(ex)}
                    print(f"rank {rank} variant {vi} {var}: {ex}", file=sys.stderr, flush=True)
                s.close()
            for solver_type, pre in kw["solvers"]:
                o = hs.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=kw.get("max_it", 400),
                                           elimination_groups=[sh.num_eliminate_blocks], device=device,
                                           force_generic_path=bool(kw.get("force_generic", False)))
                s = hs.HipLinearSolver(o, rank=rank, world_size=world, p2p_exchange=exchange, p2p_max_elements=max_elems)
                assert s.p2p_selftest(), s.p2p_error
                s.set_structure(sh.bs)
                info = s.info()
                if "drop_rank" in kw:  # no-hang scenario: one rank leaves after the communicator is up, the others must not hang
                    import time
                    if rank == kw["drop_rank"]:
                        out[(name, solver_type, pre)] = {"dropped": True}
                        time.sleep(1.0)
                        s.close()
                        continue
                    t0 = time.perf_counter()
                    try:
                        s.lm_compute_step(v, b, 1e4, 0.1)
                        err = None
                    except hs.HipError as ex:
                        err = str(ex)
                    out[(name, solver_type, pre)] = {"dropped": False, "error": err, "seconds": time.perf_counter() - t0,
                                                     "p2p_enabled": int(info.p2p_enabled), "fine_grained": int(info.p2p_fine_grained)}
                    s.close()
                    continue
                if "poison" in kw:
                    # ONE rank's shard cannot be solved (a NaN in a Jacobian cell, or a point block that is singular: a zeroed E cell and
                    # no D): every rank must end the call the same way — the ranks of a step that disagree about its verdict would
                    # part ways in the caller's loop — and the instances stay usable for a good call afterwards
                    pr, what = kw["poison"]
                    vb, Db = v.copy(), D.copy()
                    if rank == pr:
                        if what == "nan":
                            vb[int(sh.bs.cell_value_pos[0]) + 1] = np.nan
                        else:   # the first row's point: all its E cells zero (its rows are the first of this shard), D = 0 on its columns
                            c0 = int(sh.bs.cell_col_block[0])
                            rows = [r for r in range(min(sh.bs.num_row_blocks, 4096)) if int(sh.bs.cell_col_block[sh.bs.row_cell_ptr[r]]) == c0]
                            for r in rows:
                                q = int(sh.bs.row_cell_ptr[r])
                                n = int(sh.bs.row_block_size[r]) * int(sh.bs.col_block_size[c0])
                                vb[int(sh.bs.cell_value_pos[q]): int(sh.bs.cell_value_pos[q]) + n] = 0.0
                            p0 = int(sh.bs.col_block_pos[c0])
                            Db[p0: p0 + int(sh.bs.col_block_size[c0])] = 0.0
                    rec = {"rank": rank}
                    x, summ = s.solve(vb, b, hs.PerSolveOptions(D=Db, q_tolerance=0.1, r_tolerance=-1.0))
                    rec["bad_solve"] = (summ.termination_type, summ.num_iterations, summ.message, bool(np.isfinite(x).all()))
                    try:
                        step, summ, mcc = s.lm_compute_step(vb, b, 1e4, 0.1, min_diagonal=0.0 if what != "nan" else 1e-6)
                        rec["bad_step"] = (summ.termination_type, summ.num_iterations, summ.message, bool(np.isfinite(step).all()))
                    except hs.HipError as ex:
                        rec["bad_step"] = ("error", str(ex))
                    x, summ = s.solve(v, b, hs.PerSolveOptions(D=D, q_tolerance=0.1, r_tolerance=-1.0))
                    rec["good_solve"] = (x, summ.termination_type, summ.num_iterations, None, summ.message)
                    rec["col_index"], rec["n_e"] = sh.col_index, int(sh.bs.col_block_size[: sh.num_eliminate_blocks].sum())
                    s.close()
                    out[(name, solver_type, pre)] = rec
                    continue
                rec = {"path": int(info.kernel_path), "world": int(info.world_size), "rank": int(info.rank),
                       "col_index": sh.col_index, "n_e": int(sh.bs.col_block_size[: sh.num_eliminate_blocks].sum())}
                # (1) converged solve
                x, summ = s.solve(v, b, hs.PerSolveOptions(D=D, q_tolerance=-1.0, r_tolerance=1e-12))
                rec["converged"] = (x, summ.termination_type, summ.num_iterations, None, summ.message)
                # (2) the call LM makes
                x, summ = s.solve(v, b, hs.PerSolveOptions(D=D, q_tolerance=0.1, r_tolerance=-1.0))
                rec["lm_style"] = (x, summ.termination_type, summ.num_iterations, None, summ.message)
                # (3) the whole LM step on the device (f1), incl. the all-reduced {finite flag, model cost}
                step, summ, mcc = s.lm_compute_step(v, b, kw.get("radius", 1e4), 0.1)
                rec["lm_step"] = (step, summ.termination_type, summ.num_iterations, mcc, summ.message)
                rec["collectives"] = int(s.info().collectives_last_step)
                # (3b) the retry after a rejected step: the shard keeps its tiles, the diagonal is reused at half the radius
                step, summ, mcc = s.lm_compute_step(None, None, kw.get("radius", 1e4) / 2, 0.1, reuse_diagonal=True, values_unchanged=True)
                rec["retry"] = (step, summ.termination_type, summ.num_iterations, mcc, summ.message)
                # (3c) the same step with the values streamed up behind an "evaluator" (ceres_hip_values_begin / _ready / _end): a shard's upload
                if not kw.get("no_streamed"):
                    hv, hb = np.full(v.shape[0], np.nan), np.full(b.shape[0], np.nan)
                    s.values_begin(hv, hb)
                    hv[:] = v
                    hb[:] = b
                    nrb = sh.bs.num_row_blocks
                    for r0 in range(0, nrb, max(1, nrb // 5)):
                        if (r0 // max(1, nrb // 5)) % 3 != 2:   # (some runs of row blocks never announced: _end sends them)
                            s.values_ready(r0, min(max(1, nrb // 5), nrb - r0))
                    s.values_end(None)
                    step, summ, mcc = s.lm_compute_step(None, None, kw.get("radius", 1e4), 0.1, values_unchanged=True)
                    rec["streamed"] = (step, summ.termination_type, summ.num_iterations, mcc, summ.message)
                # (4) operators that sum over ranks
                if solver_type == hs.ITERATIVE_SCHUR:
                    s.load(v, b, D)
                    s.schur_init()
                    xf = np.random.default_rng(5).standard_normal(info.num_cols_f)
                    rec["rhs"] = s.schur_rhs()
                    rec["sx"] = s.schur_sx(xf)
                    s.schur_jacobi_update()
                    rec["precond"] = s.preconditioner_blocks()
                else:
                    s.load(v, b, D)
                    xx = np.random.default_rng(6).standard_normal(prob.bs.num_cols)[sh.col_index]
                    rec["jtjx"] = s.jtjx(xx)
                    rec["jtb"] = s.jtb()
                s.close()
                out[(name, solver_type, pre)] = rec
        conn.send(("done", out))
    except Exception:
        conn.send(("error", traceback.format_exc()))
    finally:
        conn.close()
