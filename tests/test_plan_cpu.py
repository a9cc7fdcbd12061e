"""Host logic of the <2,3,9> path: the tile packing plan (csrc/plan.cc) checked on CPU, and the
packed formulation the fused kernels implement (per-point segment sums, per-camera scatter)
emulated in numpy against the oracle."""
import numpy as np
import pytest

import os

from conftest import ROOT, pkg


def plan_of(p):
    return pkg.hip_solver.debug_plan(p.bs, p.num_eliminate_blocks)


def check_plan_invariants(p, plan):
    assert plan["eligible"]
    nt = plan["n_tiles"]
    row, pt, cam = plan["slot_row"], plan["slot_pt"], plan["slot_cam"]
    valid = plan["valid"].astype(bool)
    # every row block appears exactly once
    assert np.array_equal(np.sort(row[valid]), np.arange(p.bs.num_row_blocks))
    assert (row[~valid] == -1).all() and (pt[~valid] == -1).all()
    first, last = plan["seg_first"].astype(int), plan["seg_last"].astype(int)
    lane = np.tile(np.arange(64), nt)
    assert (first <= lane).all() and (lane <= last).all() and (last < 64).all()
    kind, aux = plan["tile_kind"], plan["tile_aux"]
    # inside a normal tile a segment is exactly one whole point
    track = np.bincount(pt[valid])
    for t in np.flatnonzero(kind == 0):
        sl = slice(t * 64, t * 64 + 64)
        v = valid[sl]
        pts = pt[sl][v]
        assert (np.diff(pts) >= 0).all()
        for q in np.unique(pts):
            lanes = np.flatnonzero(v & (pt[sl] == q))
            assert len(lanes) == track[q] <= 64
            assert (first[sl][lanes] == lanes[0]).all() and (last[sl][lanes] == lanes[-1]).all()
            assert np.array_equal(lanes, np.arange(lanes[0], lanes[-1] + 1))
        longest = max(np.bincount(pts).max(), 1)
        assert aux[t] & 0xff == longest and aux[t] >> 8 == len(np.unique(pts))
        # lane L owns scalars L and 64 + L of the tile's point range: tail lanes of points L/3, (64+L)/3
        upts = np.unique(pts)
        # ... so a tile's point range must fit the 128 scalars two-per-lane ownership covers, and every one of
        # its 3 * npts scalars must have a store bit (a tile of 43+ one-observation points would silently lose
        # the tail of its point range in the cooperative JtJx path)
        assert 3 * len(upts) <= 128 and len(upts) <= 42
        owned = sorted([L for L in range(64) if plan["has_a"][sl][L]] + [64 + L for L in range(64) if plan["has_b"][sl][L]])
        assert owned == list(range(3 * len(upts)))
        tails = [np.flatnonzero(v & (pt[sl] == q))[-1] for q in upts]
        for L in range(64):
            for off, has, tail in ((0, plan["has_a"][sl][L], plan["tail_a"][sl][L]), (64, plan["has_b"][sl][L], plan["tail_b"][sl][L])):
                i = (off + L) // 3
                assert bool(has) == (i < len(upts))
                if has:
                    assert tail == tails[i]
    # a long point owns its tiles
    t = 0
    while t < nt:
        if kind[t] == 3:        # the streaming kernels take it in rounds (test_long_point_rounds)
            n = aux[t]
            assert n >= 2 and (kind[t + 1:t + n] == 2).all()
            sl = slice(t * 64, (t + n) * 64)
            q = np.unique(pt[sl][valid[sl]])
            assert len(q) == 1 and track[q[0]] == valid[sl].sum() > 64
            t += n
        else:
            assert kind[t] == 0
            t += 1
    # cameras of one point are distinct
    key = pt[valid].astype(np.int64) * (cam.max() + 1) + cam[valid]
    assert len(np.unique(key)) == valid.sum()
    return track


@pytest.mark.parametrize("layout", ["schur", "cgnr"])
def test_plan_invariants_and_padding(problems, layout):
    p = problems.synthetic_bal(None, layout=layout, num_cameras=40, num_points=2500, num_observations=11000, seed=5)
    plan = plan_of(p)
    track = check_plan_invariants(p, plan)
    waste = 1.0 - p.bs.num_row_blocks / (plan["n_tiles"] * 64)
    assert waste < 0.08, waste


def test_plan_long_points(problems):
    # a few cameras-heavy points: tracks of 65..200 observations
    p = problems.synthetic_bal(None, num_cameras=220, num_points=300, num_observations=9000, seed=9, skew=0.0)
    plan = plan_of(p)
    track = check_plan_invariants(p, plan)
    assert np.isin(plan["tile_kind"], (1, 3)).sum() == (track > 64).sum()


def check_long_rounds(r):
    """The long points sit behind the normal tiles of their range; every one of up to 8 tiles is in exactly one round, every longer
    one has a sequence of rounds of its own: sum rounds over its tiles, then apply rounds over the same tiles."""
    kind, aux, flag, seq = r["tile_kind"], r["tile_aux"], r["round_flag"], r["seq_ptr"]
    assert r["long_behind"]
    assert seq[0] == 0 and seq[-1] == len(flag) and (np.diff(seq) >= 1).all()
    seen = np.zeros(r["n_tiles"], int)
    for g in range(len(r["long_ptr"])):
        t0, tl, t1 = r["range_tile_ptr"][g], r["long_ptr"][g], r["range_tile_ptr"][g + 1]
        assert t0 <= tl <= t1
        assert (kind[t0:tl] == 0).all() and (kind[tl:t1] != 0).all()
        lengths = []
        for q in range(r["round_ptr"][g], r["round_ptr"][g + 1]):
            rounds = range(seq[q], seq[q + 1])
            lengths.append(len(rounds))
            if len(rounds) > 1 or flag[rounds[0]] != 0:       # ONE point of more than 8 tiles
                head = int(r["round_word"][rounds[0]][0] & 0x3FFFFFF)
                n = aux[head]
                assert kind[head] == 3 and n > 8 and len(rounds) == 2 * ((n + 7) // 8)
                tiles = {1: [], 2: []}
                for i, q_r in enumerate(rounds):
                    phase = 1 if i < len(rounds) // 2 else 2
                    last = (i + 1) % (len(rounds) // 2) == 0
                    assert flag[q_r] == phase + (4 if last else 0)
                    words = r["round_word"][q_r]
                    act = words != 0xFFFFFFFF
                    k = int(act.sum())
                    assert act[:k].all() and k == (8 if not last else n - 8 * ((n - 1) // 8))
                    assert ((words[:k] >> 29) & 7 == k - 1).all() and ((words[:k] >> 26) & 7 == 0).all()
                    tiles[phase] += [int(w & 0x3FFFFFF) for w in words[:k]]
                assert tiles[1] == tiles[2] == list(range(head, head + n))
                seen[head:head + n] += 1
                continue
            words = r["round_word"][rounds[0]]
            w = 0
            assert words[0] != 0xFFFFFFFF                     # (an idle wave issues the loads of the round's first tile)
            while w < 8:
                if words[w] == 0xFFFFFFFF:
                    assert (words[w:] == 0xFFFFFFFF).all()    # idle waves come last
                    break
                tile, w0, n = int(words[w] & 0x3FFFFFF), int((words[w] >> 26) & 7), int((words[w] >> 29) & 7) + 1
                assert w0 == w and w + n <= 8 and kind[tile] == 3 and aux[tile] == n and tl <= tile < t1
                for k in range(n):
                    assert words[w + k] == np.uint32((tile + k) | (w0 << 26) | ((n - 1) << 29))
                    seen[tile + k] += 1
                w += n
        assert lengths == sorted(lengths, reverse=True)        # the longest sequences first
    in_round = np.zeros(r["n_tiles"], bool)
    for t in np.flatnonzero(kind == 3):
        in_round[t:t + aux[t]] = True
    assert np.array_equal(seen, in_round.astype(int))
    assert not (kind == 1).any()


@pytest.mark.parametrize("renumber", [False, True])
def test_long_point_rounds(problems, renumber):
    tracks = [3, 70, 2, 2, 129, 64, 65, 5, 513, 1, 300, 512, 7, 449, 200, 100, 66] + [4] * 50 + [90, 1000, 3]
    p = problems.bal_from_tracks(tracks, 1100, seed=3)
    r = pkg.hip_solver.debug_long_rounds(p.bs, p.num_eliminate_blocks, renumber)
    check_long_rounds(r)
    assert (r["tile_kind"] == 3).sum() == 12
    # 1000 observations: 2 x (8 + 8) tiles, 513: 2 x (8 + 1); then tightest fit, longest first: (8) (8) (5, 3) (4, 2, 2) (2, 2, 2 + two idle waves)
    assert np.diff(r["seq_ptr"]).tolist() == [4, 4, 1, 1, 1, 1, 1]
    assert (r["round_word"][r["round_flag"] == 0] == 0xFFFFFFFF).sum() == 2
    p = problems.synthetic_bal(None, num_cameras=220, num_points=300, num_observations=9000, seed=9, skew=0.0)
    check_long_rounds(pkg.hip_solver.debug_long_rounds(p.bs, p.num_eliminate_blocks, renumber))


def test_rounds_schedule_computes_the_schur_product(oracle, problems):
    """S x computed by walking the plan the way a workgroup of the streaming kernel does (csrc/kernels_bal.hip: the pipelined loop over
    the normal tiles, then sequence by sequence, round by round: a tile per wave, the waves [w0, w0 + cnt) of a point exchange their
    tile sums; a point of more than 8 tiles: sum rounds into a running total, then apply rounds with it) equals the oracle's product —
    the contract between the plan's round words / flags / sequences and the kernel, on CPU."""
    tracks = [3, 70, 2, 129, 64, 65, 5, 513, 1, 300, 512, 7, 449, 200, 66] + [4] * 30 + [90, 1000, 3] + [130, 2] * 6
    p = problems.bal_from_tracks(tracks, 1100, seed=5)
    n_o, n_p, n_c = p.bs.num_row_blocks, p.num_eliminate_blocks, 1100
    plan = plan_of(p)                                               # (caller order: debug_plan's default)
    r = pkg.hip_solver.debug_long_rounds(p.bs, p.num_eliminate_blocks, False)
    assert r["n_tiles"] == plan["n_tiles"] and np.array_equal(r["tile_kind"], plan["tile_kind"])
    nt = plan["n_tiles"]
    valid = plan["valid"].astype(bool).reshape(nt, 64)
    row, pt, cam = (plan[k].reshape(nt, 64) for k in ("slot_row", "slot_pt", "slot_cam"))
    E = p.values[: 6 * n_o].reshape(n_o, 2, 3)
    F = p.values[6 * n_o:].reshape(n_o, 2, 9)
    De, Df = p.D[: 3 * n_p].reshape(n_p, 3), p.D[3 * n_p:]
    ete = np.einsum("pi,ij->pij", De ** 2, np.eye(3))
    np.add.at(ete, p.point_of_row, np.einsum("oki,okj->oij", E, E))
    einv = np.linalg.inv(ete)
    x = np.random.default_rng(3).standard_normal(9 * n_c).reshape(n_c, 9)
    y = np.zeros((n_c, 9))

    def tile_parts(t):            # what a wave has after loading tile t: rows, F x per slot, its share of E^T F x
        v = valid[t]
        rows, cams = row[t][v], cam[t][v]
        tx = np.einsum("oij,oj->oi", F[rows], x[cams])
        return rows, cams, tx, np.einsum("oij,oi->j", E[rows], tx)

    def finish(t, parts, u):      # the tile from "registers", with the point's sum u
        rows, cams, tx, _ = parts
        point = pt[t][0]
        z = tx - E[rows] @ (einv[point] @ u)
        np.add.at(y, cams, np.einsum("oij,oi->oj", F[rows], z))

    done = np.zeros(nt, int)
    for t in np.flatnonzero(r["tile_kind"] == 0):                   # the pipelined loop: a tile holds whole points
        rows, cams, tx, _ = tile_parts(t)
        pts = pt[t][valid[t]]
        u = np.zeros((n_p, 3))
        np.add.at(u, pts, np.einsum("oij,oi->oj", E[rows], tx))
        z = tx - np.einsum("oij,oj->oi", E[rows], np.einsum("oij,oj->oi", einv[pts], u[pts]))
        np.add.at(y, cams, np.einsum("oij,oi->oj", F[rows], z))
        done[t] += 1
    seq, flag, words = r["seq_ptr"], r["round_flag"], r["round_word"]
    for q in range(len(seq) - 1):                                   # one workgroup per sequence, its rounds in order
        carry = np.zeros(3)
        for rd in range(seq[q], seq[q + 1]):
            w = words[rd]
            act = w != 0xFFFFFFFF
            tiles = (w & 0x3FFFFFF).astype(int)
            w0s, cnts = ((w >> 26) & 7).astype(int), ((w >> 29) & 7).astype(int) + 1
            if flag[rd] & 2:                                        # apply round: the running total of the sum rounds
                for k in np.flatnonzero(act):
                    finish(tiles[k], tile_parts(tiles[k]), carry)
                    done[tiles[k]] += 1
                if flag[rd] & 4:
                    carry = np.zeros(3)
                continue
            parts = {k: tile_parts(tiles[k]) for k in np.flatnonzero(act)}
            red = {k: parts[k][3] for k in parts}                   # the exchange area
            if flag[rd] & 1:                                        # sum round: every wave adds the round's total to its carry
                carry = carry + sum(red[k] for k in range(cnts[0]))
                continue
            for k in parts:
                finish(tiles[k], parts[k], sum(red[j] for j in range(w0s[k], w0s[k] + cnts[k])))
                done[tiles[k]] += 1
    assert (done == 1).all()
    y = y.reshape(-1) + Df ** 2 * x.reshape(-1)
    m = oracle.Matrix(p.bs, n_p)
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    ref = isc.sx(x.reshape(-1))
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-11 * np.abs(ref).max())


def test_long_point_rounds_of_hybrid_groups(problems):
    # real tracks (every point a long one) on more cameras than LDS holds: rounds per group
    p = problems.libmv_bal(2, 8, with_values=False)
    r = pkg.hip_solver.debug_long_rounds(p.bs, p.num_eliminate_blocks, True, groups=16, rows=400)
    assert len(r["long_ptr"]) == 16
    check_long_rounds(r)
    assert (r["round_ptr"][1:] > r["round_ptr"][:-1]).sum() >= 8   # the groups have points, hence rounds


def test_plan_caps_points_per_tile(problems):
    # runs of one-observation points: without the cap a tile would take up to 64 of them (192 point-space scalars)
    tracks = [1] * 500 + [2] * 40 + [1] * 90 + [3, 1, 1, 70, 1, 1] + [1] * 64
    p = problems.bal_from_tracks(tracks, 80, seed=4)
    plan = plan_of(p)
    check_plan_invariants(p, plan)
    per_tile = [len(np.unique(plan["slot_pt"][t * 64:t * 64 + 64][plan["valid"][t * 64:t * 64 + 64].astype(bool)]))
                for t in np.flatnonzero(plan["tile_kind"] == 0)]
    assert max(per_tile) == 42
    p = problems.bal_from_tracks(tracks, 80, layout="cgnr", seed=4)
    check_plan_invariants(p, plan_of(p))


def test_plan_rejections(problems):
    hs = pkg.hip_solver
    assert not hs.debug_plan(problems.linear_least_squares_problem(2).bs, 2)["eligible"]
    p = problems.random_schur_problem(static_sizes=(2, 3, 6), seed=1)   # rows with several 6-wide F cells on many blocks: no camera / shared split
    r = hs.debug_plan(p.bs, p.num_eliminate_blocks)
    assert not r["eligible"] and "shared strip" in r["why"]
    p = problems.synthetic_structured(12, 200, 900, camera_width=11, seed=9)   # a camera width no kernels are compiled for (round 6: 2 .. 10 all are)
    r = hs.debug_plan(p.bs, p.num_eliminate_blocks)
    assert not r["eligible"] and "no fused kernels" in r["why"]
    # a point that sees the same camera twice cannot use the fused SCHUR_JACOBI kernel
    q = problems.synthetic_bal(None, num_cameras=5, num_points=6, num_observations=14, seed=2)
    cc = q.bs.cell_col_block.copy()
    cc[3] = cc[1]  # second observation of point 0 -> same camera as the first
    from ceres_solver_amd import BlockStructure
    bad = BlockStructure(q.bs.row_block_size, q.bs.row_block_pos, q.bs.col_block_size, q.bs.col_block_pos, q.bs.row_cell_ptr,
                         cc, q.bs.cell_value_pos)
    r = hs.debug_plan(bad, q.num_eliminate_blocks)
    assert not r["eligible"] and "twice" in r["why"]


@pytest.mark.parametrize("kw", [dict(camera_width=10), dict(camera_width=6), dict(camera_width=3), dict(camera_width=6, shared_widths=(8,), locked_cameras=(0,)),
                                dict(camera_width=9, shared_widths=(5, 3), shared_first=False),
                                dict(point_width=4, camera_width=9), dict(point_width=4, camera_width=3, locked_cameras=(2,)), dict(point_width=2, camera_width=4),
                                dict(row_height=3, point_width=3, camera_width=3), dict(row_height=4, point_width=4, camera_width=2, locked_cameras=(1,))])
def test_plan_of_other_shapes(problems, kw):
    """Camera widths other than 9, shared blocks, rows without a camera cell, point blocks 2 and 4 wide, rows 3 and 4 high (common.h: shapes): the tiles still
    hold every row once, grouped by point; a row without a camera cell is a valid slot whose camera is -2."""
    hs = pkg.hip_solver
    p = problems.synthetic_structured(25, 700, 3300, seed=11, with_values=False, **kw)
    r = hs.debug_plan(p.bs, p.num_eliminate_blocks)
    assert r["eligible"], r
    valid = r["valid"].astype(bool)
    rows = r["slot_row"][valid]
    assert np.array_equal(np.sort(rows), np.arange(p.bs.num_row_blocks))
    cams = r["slot_cam"][valid]
    want_cam = p.camera_of_row[rows]          # column block of the row's camera cell, -1: none (a locked camera)
    locked = want_cam < 0
    assert (cams[locked] == -2).all() and locked.sum() == (p.camera_of_row < 0).sum()
    # camera ids = ranks of the camera blocks among the camera-side blocks that are not shared (structured_bal: cameras are contiguous blocks)
    n_shared = len(kw.get("shared_widths", ()))
    cam0 = p.num_eliminate_blocks + (n_shared if kw.get("shared_first", True) else 0)
    assert np.array_equal(cams[~locked], want_cam[~locked] - cam0)
    # rows of a point are adjacent slots of one tile (or whole tiles): slot_pt is non-decreasing inside a tile
    pt = r["slot_pt"].reshape(-1, 64)
    vm = valid.reshape(-1, 64)
    for t in range(pt.shape[0]):
        q = pt[t][vm[t]]
        assert (np.diff(q) >= 0).all()


def test_packed_formulation_matches_oracle(oracle, problems):
    """S x computed the way the fused kernel does it (segment sum of E^T F x over the slots of a
    point, 3x3 inverse, per-observation correction, scatter by camera) equals the oracle's
    four-pass ImplicitSchurComplement product."""
    p = problems.synthetic_bal(None, num_cameras=30, num_points=700, num_observations=3300, seed=21)
    plan = plan_of(p)
    n_o, n_p, n_c = p.bs.num_row_blocks, p.num_eliminate_blocks, 30
    E = p.values[: 6 * n_o].reshape(n_o, 2, 3)
    F = p.values[6 * n_o:].reshape(n_o, 2, 9)
    valid = plan["valid"].astype(bool)
    rows, pts, cams = plan["slot_row"][valid], plan["slot_pt"][valid], plan["slot_cam"][valid]
    Es, Fs = E[rows], F[rows]
    De, Df = p.D[: 3 * n_p].reshape(n_p, 3), p.D[3 * n_p:]
    rng = np.random.default_rng(3)
    x = rng.standard_normal(9 * n_c)
    t = np.einsum("oij,oj->oi", Fs, x.reshape(n_c, 9)[cams])
    u = np.zeros((n_p, 3))
    np.add.at(u, pts, np.einsum("oij,oi->oj", Es, t))
    ete = np.zeros((n_p, 3, 3))
    np.add.at(ete, pts, np.einsum("oki,okj->oij", Es, Es))
    ete += np.einsum("pi,ij->pij", De ** 2, np.eye(3))
    v = np.linalg.solve(ete, u[:, :, None])[:, :, 0]
    z = t - np.einsum("oij,oj->oi", Es, v[pts])
    y = np.zeros((n_c, 9))
    np.add.at(y, cams, np.einsum("oij,oi->oj", Fs, z))
    y = y.reshape(-1) + Df ** 2 * x
    m = oracle.Matrix(p.bs, n_p)
    isc = oracle.ImplicitSchurComplement(m)
    isc.init(p.values, p.D, p.b)
    ref = isc.sx(x)
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    # and the SCHUR_JACOBI diagonal blocks through M_o = I - E (E^T E)^-1 E^T
    einv = np.linalg.inv(ete)
    M = np.eye(2)[None] - np.einsum("oij,ojk,olk->oil", Es, einv[pts], Es)
    blocks = np.zeros((n_c, 9, 9))
    np.add.at(blocks, cams, np.einsum("oia,oij,ojb->oab", Fs, M, Fs))
    blocks += np.einsum("ci,ij->cij", Df.reshape(n_c, 9) ** 2, np.eye(9))
    _, raw = m.schur_jacobi(p.values, p.D)
    raw = raw.reshape(n_c, 9, 9)
    iu = np.triu_indices(9)
    np.testing.assert_allclose(blocks[:, iu[0], iu[1]], raw[:, iu[0], iu[1]], rtol=0, atol=1e-12 * np.abs(raw).max())


def test_bal_scene_generator_is_a_consistent_bal_problem(problems):
    # the front-end's input for SURVEY §8 f4 (bench.py, tests): indices in range, every point observed by
    # distinct cameras, observations near the projection of the perturbed start (a solvable problem)
    nc, npts, cam, pt, obs, par = problems.bal_scene(None, num_cameras=12, num_points=400, num_observations=1900, seed=7, skew=0.5)
    assert (nc, npts) == (12, 400) and cam.shape == pt.shape == (1900,) and obs.shape == (1900, 2)
    assert par.shape == (9 * nc + 3 * npts,)
    assert cam.min() >= 0 and cam.max() < nc and pt.min() == 0 and pt.max() == npts - 1
    pairs = set(zip(pt.tolist(), cam.tolist()))
    assert len(pairs) == 1900
    cams, pts = par[:9 * nc].reshape(nc, 9), par[9 * nc:].reshape(npts, 3)
    proj = problems._snavely(cams[cam], pts[pt], False)
    r = proj - obs
    assert np.sqrt((r * r).mean()) < 60.0   # pixels: perturbed start, not garbage
    again = problems.bal_scene(None, num_cameras=12, num_points=400, num_observations=1900, seed=7, skew=0.5)
    np.testing.assert_array_equal(again[4], obs)


@pytest.mark.parametrize("seed,nc,npts,nobs,skew", [(1, 12, 900, 4000, 0.0), (2, 300, 250, 9000, 0.0), (3, 60, 5000, 21000, 0.7)])
def test_point_ids_are_recoverable_from_the_segment_words(problems, seed, nc, npts, nobs, skew):
    # The kernels do not load a per-slot point id: finish_slot() (csrc/kernels_bal.hip) rebuilds it as
    # tile_pt0 + (number of segment heads at or below the lane) - 1, tile_pt0 = point of lane 0.  Emulate
    # that on the plan (normal tiles, and the one-segment tiles of long points) and compare with slot_pt.
    p = problems.synthetic_bal(None, num_cameras=nc, num_points=npts, num_observations=nobs, seed=seed, skew=skew)
    plan = plan_of(p)
    nt = plan["n_tiles"]
    valid = plan["valid"].astype(bool).reshape(nt, 64)
    first = plan["seg_first"].astype(int).reshape(nt, 64)
    pt = plan["slot_pt"].reshape(nt, 64)
    lane = np.arange(64)[None, :]
    assert valid[:, 0].all()                          # lane 0 of every tile is a valid slot
    heads = valid & (first == lane)
    rank = np.cumsum(heads, axis=1) - 1               # popcount(ballot(heads) & mask_le(lane)) - 1
    rebuilt = pt[:, :1] + rank
    assert np.array_equal(rebuilt[valid], pt[valid])
    # valid slots are a prefix of the tile (no holes), so the heads seen below a lane are exactly its predecessors
    assert (np.diff(valid.astype(int), axis=1) <= 0).all()
    if nc >= 100:
        assert np.isin(plan["tile_kind"], (1, 3)).any()        # the long-point case is exercised


@pytest.mark.parametrize("layout", ["schur", "cgnr"])
def test_trailing_rows_without_a_point_cell_are_a_remainder_not_a_rejection(problems, layout):
    """Rows behind the BAL rows that touch camera blocks only (priors, regularisers: the rows of
    SchurEliminator::NoEBlockRowsUpdate, internal/ceres/schur_eliminator_impl.h:574-666) leave the tile plan of the BAL rows
    untouched; the same rows in FRONT of BAL rows, or a trailing row with a point cell, still send the problem to the generic path."""
    p = problems.synthetic_bal(None, layout=layout, num_cameras=15, num_points=700, num_observations=3300, seed=4, skew=0.3)
    nelim = p.num_eliminate_blocks
    base = plan_of(p)
    q = problems.add_camera_rows(p, 25, seed=2, row_size=5, pair_fraction=0.5)
    with_rows = pkg.hip_solver.debug_plan(q.bs, nelim)
    assert with_rows["eligible"] and with_rows["n_tiles"] == base["n_tiles"]
    for k in ("slot_row", "slot_cam", "slot_pt", "tile_kind", "tile_aux"):
        assert np.array_equal(with_rows[k], base[k]), k
    # round 5: the same next to every compiled shape without a shared strip (the remainder kernels are templated on the camera width) ...
    if layout == "schur":
        for kw in (dict(camera_width=10), dict(camera_width=6, point_width=4), dict(camera_width=3, point_width=3, row_height=3)):
            o = problems.synthetic_structured(15, 700, 3300, seed=4, skew=0.3, **kw)
            o_base = pkg.hip_solver.debug_plan(o.bs, o.num_eliminate_blocks)
            o_rows = pkg.hip_solver.debug_plan(problems.add_camera_rows(o, 25, seed=2, row_size=5, pair_fraction=0.5, camera_width=kw["camera_width"]).bs,
                                               o.num_eliminate_blocks)
            assert o_base["eligible"] and o_rows["eligible"] and o_rows["n_tiles"] == o_base["n_tiles"], kw
            assert np.array_equal(o_rows["slot_row"], o_base["slot_row"])
        # ... but not next to a shared strip (a prior on the shared block would have to join the strip's sums)
        o = problems.synthetic_structured(15, 700, 3300, seed=4, skew=0.3, camera_width=6, shared_widths=(8,))
        assert pkg.hip_solver.debug_plan(o.bs, o.num_eliminate_blocks)["eligible"]
        assert not pkg.hip_solver.debug_plan(problems.add_camera_rows(o, 25, seed=2, row_size=5, camera_width=6).bs, o.num_eliminate_blocks)["eligible"]
    # a row higher than the generic kernels take
    tall = problems.add_camera_rows(p, 3, seed=2, row_size=17)
    assert not pkg.hip_solver.debug_plan(tall.bs, nelim)["eligible"]
    # camera-only rows that are NOT trailing: put one BAL row behind them
    from ceres_solver_amd.block_structure import BlockStructure
    b = q.bs
    nrb = b.num_row_blocks
    ptr = b.row_cell_ptr.astype(np.int64)
    rs = np.concatenate([b.row_block_size, [2]])
    rp = np.concatenate([b.row_block_pos, [b.num_rows]])
    cells0 = b.cell_col_block[ptr[0]:ptr[1]]
    bad = BlockStructure(rs, rp, b.col_block_size, b.col_block_pos, np.concatenate([ptr, [ptr[-1] + 2]]),
                         np.concatenate([b.cell_col_block, cells0]), np.concatenate([b.cell_value_pos, b.cell_value_pos[ptr[0]:ptr[1]]]))
    assert bad.num_row_blocks == nrb + 1
    if layout == "cgnr":   # (in the Schur ordering an E row behind E-free rows is not a valid structure for a Schur solver at all)
        assert not pkg.hip_solver.debug_plan(bad, nelim)["eligible"]   # (that copied row observes a camera a second time)
        # round 5: without an elimination order the rows without a point cell may sit ANYWHERE among the observation rows (a prior
        # added together with its camera): the tiles still hold exactly the observation rows, in the same point-major arrangement
        rng = np.random.default_rng(3)
        n_obs = p.bs.num_row_blocks
        keys = np.concatenate([np.arange(n_obs, dtype=np.float64), rng.uniform(-1, n_obs, 25)])   # the 25 camera rows go anywhere
        mixed = problems.permute_rows(q, np.argsort(keys, kind="stable"))
        got = pkg.hip_solver.debug_plan(mixed.bs, nelim)
        assert got["eligible"] and got["n_tiles"] == base["n_tiles"]
        for k in ("slot_row", "slot_cam", "slot_pt", "tile_kind", "tile_aux"):   # (slot_row: compact ids of the observation rows)
            assert np.array_equal(got[k], base[k]), k
    else:   # Schur ordering: the same rows in the middle are not a structure the reference's Schur solvers accept
        order = np.arange(q.bs.num_row_blocks)
        order[[5, -1]] = order[[-1, 5]]
        assert not pkg.hip_solver.debug_plan(problems.permute_rows(q, order).bs, nelim)["eligible"]


@pytest.mark.parametrize("shape,kw", [("ladybug1723", {}), (None, dict(num_cameras=300, num_points=250, num_observations=9000)),
                                      (None, dict(num_cameras=40, num_points=30000, num_observations=70000))])
def test_renumbered_points_fill_the_tiles(problems, shape, kw, monkeypatch):
    """The plan of the Schur solvers renumbers the points (windowed best fit, csrc/plan.cc): same rows, each exactly once, whole
    points per tile with consecutive INTERNAL ids, a point's rows still in the caller's order — and far fewer padding slots."""
    p = problems.synthetic_bal(shape, layout="schur", seed=5, skew=0.6, **kw)
    plain = plan_of(p)
    monkeypatch.setenv("CERES_HIP_DEBUG_PLAN_REORDER", "1")
    plan = plan_of(p)
    assert plan["eligible"]
    valid = plan["valid"].astype(bool)
    rows = plan["slot_row"][valid]
    assert np.array_equal(np.sort(rows), np.arange(p.bs.num_row_blocks))           # every row, once
    ipt = plan["slot_pt"][valid]                                                      # internal point ids
    caller_pt = p.point_of_row[rows]
    # internal ids are a renumbering of the caller's points: one-to-one
    pairs = np.unique(np.stack([ipt, caller_pt], 1), axis=0)
    assert len(pairs) == len(np.unique(ipt)) == len(np.unique(caller_pt))
    # slots in tile order: internal ids never decrease, a point's rows are consecutive slots in the caller's row order
    assert (np.diff(ipt) >= 0).all()
    same = np.diff(ipt) == 0
    assert (np.diff(rows)[same] > 0).all()
    nt = plan["n_tiles"]
    # no normal tile splits a point; tiles hold at most 42 points
    tile_of = (np.flatnonzero(valid) // 64)
    normal = plan["tile_kind"][tile_of] == 0
    first_tile = {}
    for t, q in zip(tile_of[normal].tolist(), ipt[normal].tolist()):
        assert first_tile.setdefault(q, t) == t
    assert plan["n_tiles"] <= plain["n_tiles"]
    waste = 1.0 - valid.sum() / (64.0 * nt)
    waste_plain = 1.0 - plain["valid"].sum() / (64.0 * plain["n_tiles"])
    print(f"padding {waste_plain:.4f} -> {waste:.4f} ({plain['n_tiles']} -> {nt} tiles)")
    if shape == "ladybug1723":
        assert waste < 0.012 and waste < 0.4 * waste_plain


def simulate_camera_accumulation(pl, n_cameras, groups, rng):
    """The two passes of the many-camera regime on the host, with one scalar per observation in place of the 72-byte row:
    tile pass (LDS rows per group, spilled rows into the ring, flush) and camera-major pass (units over ring rows)."""
    valid, acc, cam = pl["valid"], pl["slot_acc"], pl["slot_cam"]
    nt = pl["n_tiles"]
    v = rng.standard_normal(valid.shape[0])
    v[~valid] = 0.0
    want = np.bincount(cam[valid], weights=v[valid], minlength=n_cameras)
    ring = np.full(pl["ring_rows"], np.nan)
    spill = valid & (acc == 0xFFF)
    sp2 = spill.reshape(nt, 64)
    rows_of = (pl["tile_zbase"][:, None] + np.cumsum(sp2, axis=1) - sp2).reshape(-1)   # a slot's row = tile base + rank among the tile's spilled slots
    assert len(np.unique(rows_of[spill])) == spill.sum()
    ring[rows_of[spill]] = v[spill]
    if pl["hybrid"]:
        K = pl["rows"]
        sizes = np.diff(pl["grp_tile_ptr"])
        assert pl["grp_tile_ptr"][0] == 0 and pl["grp_tile_ptr"][-1] == nt and (sizes >= 0).all()
        slot_group = np.repeat(np.repeat(np.arange(groups), sizes), 64)
        loc = valid & ~spill
        assert (acc[loc] < K).all()
        accs = np.zeros(groups * K)
        np.add.at(accs, slot_group[loc] * K + acc[loc], v[loc])
        ring[pl["flush_row0"]: pl["flush_row0"] + groups * K] = accs
        # a row of the shared (popular) part means the same camera in every group; inside a group a window row means one camera and a
        # camera has one row (windows of different groups may overlap: the same camera, a row in each)
        hot = loc & (acc < pl["hot_rows"])
        pairs = np.unique(np.stack([acc[hot], cam[hot]], 1), axis=0)
        assert len(np.unique(pairs[:, 0])) == len(pairs)
        win = loc & (acc >= pl["hot_rows"])
        trip = np.unique(np.stack([slot_group[win], cam[win], acc[win]], 1), axis=0)
        assert len(np.unique(trip[:, :2], axis=0)) == len(trip) and len(np.unique(trip[:, [0, 2]], axis=0)) == len(trip)
    else:
        assert not (valid & ~spill).any()
    assert not np.isnan(ring).any()
    ub, ue, uc, ent = pl["unit_begin"], pl["unit_end"], pl["unit_cam"], pl["entry_row"]
    assert ((ue - ub) <= 64).all() and (ue > ub).all() and ub[0] == 0 and ue[-1] == len(ent) and (ub[1:] == ue[:-1]).all()
    assert len(np.unique(ent)) == len(ent)          # every ring row is summed at most once
    got = np.bincount(np.repeat(uc, ue - ub), weights=ring[ent], minlength=n_cameras)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    return float((valid & ~spill).sum()) / float(valid.sum())


@pytest.mark.parametrize("kw,groups,rows,min_local", [
    (dict(num_cameras=5000, num_points=40000, num_observations=120000), 64, 256, 0.45),     # 3 observations per point: one stays, + the popular ones
    (dict(num_cameras=3000, num_points=3000, num_observations=30000), 16, 512, 0.3),       # 10 per point
    (dict(num_cameras=2400, num_points=900, num_observations=90000), 8, 320, 0.1),          # long points (100 per point) own their tiles
    (dict(num_cameras=40000, num_points=30000, num_observations=60000), 4, 64, 0.0),        # more cameras than groups x rows: some are in no window
    # the device's own 256 workgroups x 2176 rows on a few thousand cameras: far fewer useful window positions than workgroups, the
    # workgroups SHARE windows (the vote is per distinct window, then the lightest workgroup of it)
    (dict(num_cameras=4500, num_points=60000, num_observations=240000), 256, 2176, 0.6),
    (dict(num_cameras=2300, num_points=30000, num_observations=120000), 256, 2176, 0.9),     # one window position: every workgroup has it
])
def test_hybrid_camera_accumulation_plan(problems, kw, groups, rows, min_local):
    """More cameras than LDS rows (csrc/plan.cc): popular cameras in every workgroup's LDS, the others in ONE workgroup's window, each
    point with a workgroup that holds one of its cameras, everything else spilled — replayed on the host, pass by pass, against plain
    per-camera sums; and the spill-everything plan (CGNR on caller-ordered vectors, chunked rings) the same way."""
    p = problems.synthetic_bal(None, layout="schur", seed=11, skew=0.6, with_values=False, **kw)
    rng = np.random.default_rng(5)
    pl = pkg.hip_solver.debug_hybrid_plan(p.bs, p.num_eliminate_blocks, groups, rows)
    assert pl is not None and pl["hybrid"] and pl["rows"] == rows
    rows_seen = pl["slot_row"][pl["valid"]]
    assert np.array_equal(np.sort(rows_seen), np.arange(p.bs.num_row_blocks))
    local = simulate_camera_accumulation(pl, kw["num_cameras"], groups, rng)
    sizes = np.diff(pl["grp_tile_ptr"])
    print(f"summed in LDS: {local:.3f} of the observations; tiles per group {sizes.min()}..{sizes.max()}")
    assert local >= min_local
    assert sizes.max() <= 1.06 * sizes.mean() + 2      # balance before locality
    plain = pkg.hip_solver.debug_hybrid_plan(p.bs, p.num_eliminate_blocks, 0, 0)
    assert plain is not None and not plain["hybrid"]
    assert simulate_camera_accumulation(plain, kw["num_cameras"], 0, rng) == 0.0


def test_cameras_that_fit_in_lds_have_no_ring(problems):
    p = problems.synthetic_bal("ladybug1723", layout="schur", seed=5, with_values=False)
    assert pkg.hip_solver.debug_hybrid_plan(p.bs, p.num_eliminate_blocks, 256, 1728) is None


@pytest.mark.parametrize("problem,copies", [(1, 1), (2, 1), (3, 1), (2, 8)])
def test_plan_of_real_visibility(problems, problem, copies, monkeypatch):
    """The three bundle-adjustment problems the reference ships (data/libmv-ba-problems, committed as tests/golden/libmv_problems.npz):
    26-71 tracks followed through 333-500 consecutive frames — every point is a LONG point (more than 64 observations: it owns whole
    tiles), the opposite corner of the plan from the BAL shapes."""
    p = problems.libmv_bal(problem, copies, with_values=False)
    plan = plan_of(p)
    check_plan_invariants(p, plan)
    assert (plan["tile_kind"] != 0).mean() > 0.9
    monkeypatch.setenv("CERES_HIP_DEBUG_PLAN_REORDER", "1")
    plan2 = plan_of(p)
    assert plan2["eligible"] and plan2["n_tiles"] <= plan["n_tiles"]
    assert np.array_equal(np.sort(plan2["slot_row"][plan2["valid"].astype(bool)]), np.arange(p.bs.num_row_blocks))


def test_hybrid_plan_keeps_real_tracks_in_lds(problems):
    """Replicated to 13 200 cameras (30 copies of problem_02 side by side) the cameras no longer fit in LDS; a copy's 440 cameras fit one
    window and windows overlap, so nearly every observation of a track is summed in its workgroup's LDS (random visibility: 56 %)."""
    p = problems.libmv_bal(2, 30, with_values=False)
    n_c = p.bs.num_col_blocks - p.num_eliminate_blocks
    pl = pkg.hip_solver.debug_hybrid_plan(p.bs, p.num_eliminate_blocks, 64, 1728)
    assert pl is not None and pl["hybrid"] and pl["hot_rows"] == 0     # no camera is "popular" here: every row goes to the windows
    local = simulate_camera_accumulation(pl, n_c, 64, np.random.default_rng(2))
    print(f"summed in LDS: {local:.3f}")
    assert local > 0.9


@pytest.mark.parametrize("n_cams,n_pts,n_obs", [(1778, 60000, 300000), (2250, 30000, 150000), (300, 20000, 90000), (16, 3000, 12000)])
def test_staged_x_plan_fits_the_lds_and_takes_the_most_observed_cameras(problems, n_cams, n_pts, n_obs):
    """Round 5 (csrc/plan.cc, BalPlan::xhot_cam): with every camera's accumulator row in LDS the bytes left (160 KiB minus 1 KiB for the
    kernels' static arrays) hold the 9 scalars of x for as many cameras as fit, most observed first; their slots carry row + 1 above the
    camera id of the index word, every other slot the camera id alone."""
    p = problems.synthetic_bal(None, layout="schur", num_cameras=n_cams, num_points=n_pts, num_observations=n_obs, seed=3, skew=0.6, with_values=False)
    plan = pkg.hip_solver.debug_staged_x_plan(p.bs, p.num_eliminate_blocks)
    assert plan is not None and plan["n_cameras"] == n_cams and plan["accumulator_bytes"] == 72 * n_cams
    staged, word = plan["staged_cam"], plan["slot_word"]
    want = min(n_cams, (160 * 1024 - 1024 - 72 * n_cams) // 72)
    assert len(staged) == want and len(set(staged.tolist())) == want
    assert 72 * n_cams + 72 * want + 1024 <= 160 * 1024
    valid = word >= 0
    cam = word[valid] & ((1 << 20) - 1)
    row = (word[valid].astype(np.int64) >> 20) & 0xFFF
    deg = np.bincount(cam, minlength=n_cams)
    # the staged cameras are a top-`want` set by number of observations, listed in non-increasing order
    assert (np.diff(deg[staged]) <= 0).all()
    if want < n_cams:
        rest = np.setdiff1d(np.arange(n_cams), staged)
        assert deg[staged].min() >= deg[rest].max()
    # slot words: row + 1 of the slot's camera, or nothing
    row_of = np.full(n_cams, -1)
    row_of[staged] = np.arange(want)
    assert np.array_equal(row, row_of[cam] + 1)
    assert row.max() < 0xFFF    # all ones is the hybrid plan's "spilled"


def test_no_staged_x_where_the_accumulators_do_not_fit(problems):
    p = problems.synthetic_bal(None, layout="schur", num_cameras=3000, num_points=9000, num_observations=40000, seed=3, skew=0.4, with_values=False)
    plan = pkg.hip_solver.debug_staged_x_plan(p.bs, p.num_eliminate_blocks)
    w = plan["slot_word"].view(np.uint32)
    real = w != 0xFFFFFFFF    # (padding slots)
    assert plan is not None and len(plan["staged_cam"]) == 0 and plan["accumulator_bytes"] > 160 * 1024 - 1024
    # the export builds the plan set_structure builds (round 6: the hybrid request of the device): the row field is a hybrid accumulator
    # row or "spilled" (0xFFF), never a staged-x row — there is no staged x in this regime
    rows = w[real] >> 20
    assert (rows == 0xFFF).any() and (rows < 0xFFF).any()


def test_malformed_structures_are_refused_or_planned_never_a_crash():
    """The boundary hands over raw int arrays (ceres_hip_block_structure): 400 random single-entry corruptions of a valid structure —
    negative, zero, huge, INT32_MAX, off by one, swapped — with elimination counts in and out of range go through the host-side analysis
    and the tile plan in a child process; each is refused with a reason or planned, none crashes (sums that leave `int` used to wrap:
    a last row block of INT32_MAX rows sized a vector with a negative number).  Three overflow cases must be refused by name."""
    import subprocess
    import sys
    code = r'''
import copy, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import os

from conftest import ROOT, pkg
hs, P = pkg.hip_solver, pkg.problems
base = P.synthetic_bal(None, layout="schur", num_cameras=7, num_points=40, num_observations=150, seed=3)
rng = np.random.default_rng(5)
fields = ["row_block_size", "row_block_pos", "col_block_size", "col_block_pos", "row_cell_ptr", "cell_col_block", "cell_value_pos"]
for trial in range(400):
    bs = copy.deepcopy(base.bs)
    field = fields[int(rng.integers(len(fields)))]
    a = getattr(bs, field).copy()
    k = int(rng.integers(a.shape[0]))
    how = int(rng.integers(6))
    if how == 0: a[k] = -int(rng.integers(1, 5))
    elif how == 1: a[k] = 0
    elif how == 2: a[k] = int(rng.integers(10**6, 10**8))
    elif how == 3: a[k] = 2**31 - 1
    elif how == 4: a[k] += 1
    else:
        j = int(rng.integers(a.shape[0])); a[k], a[j] = a[j], a[k]
    setattr(bs, field, a)
    nelim = int(rng.choice([0, base.num_eliminate_blocks, base.num_eliminate_blocks + 1, -1, 10**6]))
    print("TRY", trial, field, k, how, nelim, flush=True)
    hs.debug_plan(bs, nelim)
    hs.debug_staged_x_plan(bs, nelim)
for field, k, want in (("row_block_size", -1, "rows"), ("col_block_size", -1, "columns"), ("cell_value_pos", 3, "value positions")):
    bs = copy.deepcopy(base.bs)
    a = getattr(bs, field).copy(); a[k] = 2**31 - 1; setattr(bs, field, a)
    r = hs.debug_plan(bs, base.num_eliminate_blocks)
    print("OVERFLOW", field, r["eligible"], r.get("why", ""), flush=True)
    assert not r["eligible"] and want in r["why"], r
print("DONE", flush=True)
''' % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    lines = p.stdout.strip().splitlines()
    assert p.returncode == 0 and lines and lines[-1] == "DONE", (p.returncode, lines[-3:], p.stderr[-800:])
