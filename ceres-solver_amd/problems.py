"""Host-side problem builders: the reference's hand-written known-answer problems
and synthetic BAL-shaped block-sparse Jacobians.

* (the reference's hand-written known-answer problems — LinearLeastSquaresProblem0..6, the BlockSparseMatrix test matrices — are
  test vectors and live with the tests: tests/reference_fixtures.py)
* `random_schur_problem`: random block sizes with an E|F partition, rows without an
  E block and rows with several F cells (the structure cases of problems 2, 4, 6).
* `synthetic_bal`: the generator SURVEY.md §8(d) specifies — BAL block counts
  (`BAL_SHAPES`), every point seen by >= 2 distinct cameras, rows grouped by point,
  E|F-split or row-sequential value layout, values/b ~ N(0,1),
  D = sqrt(clamp(diag(J^T J), 1e-6, 1e32) / 1e4)
  (internal/ceres/levenberg_marquardt_strategy.cc:84-96), seed 38401
  (examples/bundle_adjuster.cc:138).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .block_structure import BlockStructure

# cameras, points, observations  (SURVEY.md §8 header / BASELINE.md §2)
BAL_SHAPES = {
    "dubrovnik16": (16, 22106, 83718),
    "ladybug1723": (1723, 156502, 678718),
    "venice1778": (1778, 993923, 5001946),
    "synthetic10M": (50000, 10_000_000, 30_000_000),
    # the 50 k-camera regime of synthetic10M at a tenth of its points (camera accumulators do not fit in LDS)
    "synthetic1M": (50000, 1_000_000, 3_000_000),
    "synthetic10M": (50000, 10_000_000, 30_000_000),   # 10 M points x 3 observations, 50 k cameras, 5.76 GB of J
}


@dataclass
class LinearProblem:
    bs: BlockStructure
    values: np.ndarray
    b: np.ndarray
    D: Optional[np.ndarray]
    num_eliminate_blocks: int
    known: dict = field(default_factory=dict)  # hand-computed answers from the reference
    # BAL-shaped problems only: per-row (observation) block ids
    camera_of_row: Optional[np.ndarray] = None
    point_of_row: Optional[np.ndarray] = None

    @property
    def num_rows(self):
        return self.bs.num_rows

    @property
    def num_cols(self):
        return self.bs.num_cols


def random_schur_problem(num_e_blocks=7, num_f_blocks=5, max_rows_per_e=4, num_no_e_rows=3,
                         block_sizes=(1, 2, 3, 4), static_sizes=None, seed=0, shuffle_values=True,
                         with_D=True) -> LinearProblem:
    """Random E|F-partitioned Jacobian.  static_sizes=(row, e, f) forces one size
    triple on every E row (BAL is (2, 3, 9)); otherwise sizes vary per block."""
    rng = np.random.default_rng(seed)
    pick = lambda: int(rng.choice(block_sizes))
    if static_sizes:
        rs0, es0, fs0 = static_sizes
        col_sizes = [es0] * num_e_blocks + [fs0] * num_f_blocks
    else:
        col_sizes = [pick() for _ in range(num_e_blocks + num_f_blocks)]
    shapes = []  # (row_size, [col blocks])
    for e in range(num_e_blocks):
        for _ in range(int(rng.integers(1, max_rows_per_e + 1))):
            nf = int(rng.integers(0 if not static_sizes else 1, min(3, num_f_blocks) + 1))
            if static_sizes:
                nf = 1 if static_sizes == (2, 3, 9) else max(nf, 1)
            fs = sorted(rng.choice(num_f_blocks, size=nf, replace=False).tolist())
            shapes.append((static_sizes[0] if static_sizes else pick(), [e] + [num_e_blocks + f for f in fs]))
    for _ in range(num_no_e_rows):
        nf = int(rng.integers(1, min(3, num_f_blocks) + 1))
        fs = sorted(rng.choice(num_f_blocks, size=nf, replace=False).tolist())
        shapes.append((pick(), [num_e_blocks + f for f in fs]))
    # value positions: sequential, optionally in a shuffled cell order (arbitrary cell.position)
    cells = [(i, j) for i, (_, cols) in enumerate(shapes) for j in cols]
    order = rng.permutation(len(cells)) if shuffle_values else np.arange(len(cells))
    pos = {}
    cursor = 0
    for idx in order:
        i, j = cells[idx]
        pos[(i, j)] = cursor
        cursor += shapes[i][0] * col_sizes[j]
    rows = [(rs, [(j, pos[(i, j)]) for j in cols]) for i, (rs, cols) in enumerate(shapes)]
    bs = BlockStructure.from_rows(col_sizes, rows)
    values = rng.standard_normal(cursor)
    b = rng.standard_normal(bs.num_rows)
    D = (0.5 + rng.random(bs.num_cols)) if with_D else None
    return LinearProblem(bs, values, b, D, num_e_blocks)


def random_block_sparse(num_row_blocks=40, num_col_blocks=12, density=0.3, block_sizes=(1, 2, 3, 5), seed=0):
    """Unpartitioned random matrix in the spirit of BlockSparseMatrix::CreateRandomMatrix
    (internal/ceres/block_sparse_matrix.cc:712-782): every row block gets >= 1 cell."""
    rng = np.random.default_rng(seed)
    col_sizes = [int(rng.choice(block_sizes)) for _ in range(num_col_blocks)]
    rows, cursor = [], 0
    for _ in range(num_row_blocks):
        rs = int(rng.choice(block_sizes))
        cols = np.flatnonzero(rng.random(num_col_blocks) < density)
        if len(cols) == 0:
            cols = np.array([int(rng.integers(num_col_blocks))])
        cells = []
        for j in cols:
            cells.append((int(j), cursor))
            cursor += rs * col_sizes[j]
        rows.append((rs, cells))
    bs = BlockStructure.from_rows(col_sizes, rows)
    return LinearProblem(bs, rng.standard_normal(cursor), rng.standard_normal(bs.num_rows),
                         0.5 + rng.random(bs.num_cols), 0)


# --------------------------------------------------------------------------
# Synthetic BAL-shaped Jacobians
# --------------------------------------------------------------------------
def _track_lengths(rng, n_cams, n_points, n_obs):
    kmin = min(2, n_cams)
    if n_obs < kmin * n_points:
        raise ValueError("need at least 2 observations per point (reference precondition, "
                         "internal/ceres/reorder_program.cc:313-317)")
    mean_extra = n_obs / n_points - kmin
    k = np.full(n_points, kmin, dtype=np.int64)
    if mean_extra > 0:
        k += np.minimum(rng.geometric(1.0 / (1.0 + mean_extra), size=n_points) - 1, n_cams - kmin)
    diff = int(n_obs - k.sum())
    while diff != 0:  # nudge random points until the total is exact
        idx = rng.integers(0, n_points, size=abs(diff))
        if diff > 0:
            ok = idx[k[idx] < n_cams]
            ok = np.unique(ok)
            k[ok] += 1
            diff -= len(ok)
        else:
            ok = np.unique(idx[k[idx] > kmin])
            k[ok] -= 1
            diff += len(ok)
    return k


def _distinct_cameras(rng, n_cams, point_of_obs, weights):
    n_obs = point_of_obs.shape[0]
    draw = (lambda n: rng.choice(n_cams, size=n, p=weights)) if weights is not None else (
        lambda n: rng.integers(0, n_cams, size=n))
    cam = draw(n_obs).astype(np.int64)
    for _ in range(200):
        key = point_of_obs * n_cams + cam
        order = np.argsort(key, kind="stable")
        dup = np.zeros(n_obs, dtype=bool)
        dup[order[1:]] = key[order[1:]] == key[order[:-1]]
        n_dup = int(dup.sum())
        if n_dup == 0:
            break
        cam[dup] = draw(n_dup)
    else:  # pathological (track length close to n_cams): fix the stragglers point by point
        for p in np.unique(point_of_obs[dup]):
            sel = np.flatnonzero(point_of_obs == p)
            cam[sel] = rng.permutation(n_cams)[: len(sel)]
    return cam


def synthetic_bal(shape="dubrovnik16", layout="schur", seed=38401, skew=0.0, num_cameras=None,
                  num_points=None, num_observations=None, with_values=True) -> LinearProblem:
    """BAL-shaped <2,3,9> Jacobian.  layout="schur": column blocks = points then cameras,
    values E|F-split (ITERATIVE_SCHUR, internal/ceres/block_jacobian_writer.cc:68-167);
    layout="cgnr": column blocks in order of first use (camera, point interleaved), cells
    sorted by column block, values row-sequential (internal/ceres/block_jacobian_writer.cc:59-64)."""
    n_cams, n_points, n_obs = BAL_SHAPES[shape] if shape else (num_cameras, num_points, num_observations)
    if num_cameras is not None:
        n_cams, n_points, n_obs = num_cameras, num_points, num_observations
    # Optional on-disk cache (CERES_HIP_PROBLEM_CACHE=<dir>): the generator is deterministic, and a GPU-box session that
    # runs bench.py / the tools several times should not spend 20 s of numpy per run on the same Venice-shaped problem.
    cache_dir = os.environ.get("CERES_HIP_PROBLEM_CACHE")
    cache = None
    if cache_dir and n_obs >= 500_000:
        cache = os.path.join(cache_dir, f"bal_{layout}_{seed}_{skew}_{n_cams}_{n_points}_{n_obs}_{int(with_values)}.npz")
        if os.path.exists(cache):
            z = np.load(cache)
            bs = BlockStructure(*(z[k] for k in ("rsz", "rpos", "csz", "cpos", "rptr", "ccol", "cval")))
            D = z["D"] if with_values else None
            return LinearProblem(bs, z["values"], z["b"], D, int(z["nelim"]), {}, z["cam"], z["pt"])
    p = _generate_bal(n_cams, n_points, n_obs, layout, seed, skew, with_values)
    if cache:
        try:
            os.makedirs(cache_dir, exist_ok=True)
            b_ = p.bs
            np.savez(cache, rsz=b_.row_block_size, rpos=b_.row_block_pos, csz=b_.col_block_size, cpos=b_.col_block_pos, rptr=b_.row_cell_ptr,
                     ccol=b_.cell_col_block, cval=b_.cell_value_pos, values=p.values, b=p.b, D=p.D if p.D is not None else np.zeros(0),
                     nelim=p.num_eliminate_blocks, cam=p.camera_of_row, pt=p.point_of_row)
        except OSError:
            pass
    return p


def _generate_bal(n_cams, n_points, n_obs, layout, seed, skew, with_values) -> LinearProblem:
    rng = np.random.default_rng(seed)
    k = _track_lengths(rng, n_cams, n_points, n_obs)
    point_of_obs = np.repeat(np.arange(n_points, dtype=np.int64), k)
    weights = None
    if skew > 0:
        weights = np.arange(1, n_cams + 1, dtype=np.float64) ** (-skew)
        weights /= weights.sum()
    cam_of_obs = _distinct_cameras(rng, n_cams, point_of_obs, weights)
    # sort cameras inside each point so that rows look like a real BAL file
    order = np.lexsort((cam_of_obs, point_of_obs))
    cam_of_obs = cam_of_obs[order]
    return _assemble_bal(rng, n_cams, n_points, point_of_obs, cam_of_obs, layout, with_values)


def bal_from_tracks(track_lengths, num_cameras, layout="schur", seed=0, with_values=True) -> LinearProblem:
    """BAL-shaped <2,3,9> Jacobian with a GIVEN number of observations per point (any value >= 1, e.g. runs of
    single-observation points, which the reference allows: an eliminated block needs >= 1 residual,
    internal/ceres/reorder_program.cc:313-317); cameras of a point are distinct and random."""
    rng = np.random.default_rng(seed)
    k = np.asarray(track_lengths, dtype=np.int64)
    if k.min() < 1 or k.max() > num_cameras:
        raise ValueError("track lengths must lie in [1, num_cameras]")
    point_of_obs = np.repeat(np.arange(k.shape[0], dtype=np.int64), k)
    cam_of_obs = _distinct_cameras(rng, num_cameras, point_of_obs, None)
    order = np.lexsort((cam_of_obs, point_of_obs))
    return _assemble_bal(rng, int(num_cameras), int(k.shape[0]), point_of_obs, cam_of_obs[order], layout, with_values)


LIBMV_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "libmv_problems.npz")


def libmv_visibility(problem=2):
    """(num_cameras, num_points, camera_of_marker, point_of_marker) of one of the three bundle-adjustment problems the reference
    ships (data/libmv-ba-problems/problem_0{1,2,3}.bin, read by examples/libmv_bundle_adjuster.cc:376-455; committed as
    tests/golden/libmv_problems.npz by tests/golden/make_libmv_fixture.py): REAL visibility — a few dozen tracks followed through
    hundreds of consecutive frames, i.e. every point is seen by far more than 64 cameras and neighbouring cameras see the same points."""
    z = np.load(LIBMV_FIXTURE)
    img, trk = z[f"p{problem}_marker_image"].astype(np.int64), z[f"p{problem}_marker_track"].astype(np.int64)
    cams, cam_of = np.unique(img, return_inverse=True)
    pts, pt_of = np.unique(trk, return_inverse=True)
    assert np.isin(cams, z[f"p{problem}_camera_image"]).all() and np.isin(pts, z[f"p{problem}_point_track"]).all()
    return int(cams.shape[0]), int(pts.shape[0]), cam_of.astype(np.int64), pt_of.astype(np.int64)


def libmv_bal(problem=2, copies=1, layout="schur", seed=38401, with_values=True) -> LinearProblem:
    """<2,3,9> Jacobian with the visibility graph of a libmv problem, `copies` disjoint replicas of it side by side (cameras and points
    of replica j follow those of replica j - 1: the graph "replicated to size"), values / b ~ N(0,1) like the synthetic workloads."""
    n_c, n_p, cam_of, pt_of = libmv_visibility(problem)
    rng = np.random.default_rng(seed)
    reps = np.arange(copies, dtype=np.int64)
    point_of_obs = (pt_of[None, :] + n_p * reps[:, None]).reshape(-1)
    cam_of_obs = (cam_of[None, :] + n_c * reps[:, None]).reshape(-1)
    order = np.lexsort((cam_of_obs, point_of_obs))
    return _assemble_bal(rng, n_c * copies, n_p * copies, point_of_obs[order], cam_of_obs[order], layout, with_values)


def _assemble_bal(rng, n_cams, n_points, point_of_obs, cam_of_obs, layout, with_values) -> LinearProblem:
    n_obs = int(point_of_obs.shape[0])
    r = np.arange(n_obs, dtype=np.int64)
    if layout == "schur":
        col_sizes = np.concatenate([np.full(n_points, 3, np.int32), np.full(n_cams, 9, np.int32)])
        point_block = point_of_obs
        cam_block = n_points + cam_of_obs
        cell_col = np.stack([point_block, cam_block], axis=1).reshape(-1)
        cell_pos = np.stack([6 * r, 6 * n_obs + 18 * r], axis=1).reshape(-1)
        nelim = n_points
    elif layout == "cgnr":
        # program order: first use in (camera, point) argument order over the observation list
        first_cam = np.full(n_cams, 2 * n_obs, dtype=np.int64)
        np.minimum.at(first_cam, cam_of_obs, 2 * r)
        first_pt = np.full(n_points, 2 * n_obs, dtype=np.int64)
        np.minimum.at(first_pt, point_of_obs, 2 * r + 1)
        keys = np.concatenate([first_cam, first_pt])
        rank = np.empty(n_cams + n_points, dtype=np.int64)
        rank[np.argsort(keys, kind="stable")] = np.arange(n_cams + n_points)
        cam_block = rank[cam_of_obs]
        point_block = rank[n_cams + point_of_obs]
        col_sizes = np.empty(n_cams + n_points, dtype=np.int32)
        col_sizes[rank[:n_cams]] = 9
        col_sizes[rank[n_cams:]] = 3
        cam_first = cam_block < point_block
        cell_col = np.where(cam_first[:, None], np.stack([cam_block, point_block], 1),
                            np.stack([point_block, cam_block], 1)).reshape(-1)
        cell_pos = np.where(cam_first[:, None], np.stack([24 * r, 24 * r + 18], 1),
                            np.stack([24 * r, 24 * r + 6], 1)).reshape(-1)
        nelim = 0
    else:
        raise ValueError(layout)
    if cell_pos.max(initial=0) + 18 >= 2 ** 31:
        raise ValueError("value offsets exceed int32 (reference limit, internal/ceres/block_sparse_matrix.h:171-172)")
    col_pos = np.concatenate([[0], np.cumsum(col_sizes.astype(np.int64))[:-1]])
    bs = BlockStructure(np.full(n_obs, 2, np.int32), 2 * r, col_sizes, col_pos, 2 * np.arange(n_obs + 1, dtype=np.int64),
                        cell_col, cell_pos)
    if not with_values:
        return LinearProblem(bs, np.zeros(0), np.zeros(0), None, nelim, {}, cam_block, point_block)
    values = rng.standard_normal(24 * n_obs)
    b = rng.standard_normal(2 * n_obs)
    # D = sqrt(clamp(diag(J^T J)) / radius), radius = initial_trust_region_radius = 1e4
    diag = np.zeros(bs.num_cols)
    if layout == "schur":
        E = values[: 6 * n_obs].reshape(n_obs, 2, 3)
        F = values[6 * n_obs:].reshape(n_obs, 2, 9)
    else:
        v = values.reshape(n_obs, 24)
        cam_first = (cam_block < point_block)[:, None]
        F = np.where(cam_first, v[:, :18], v[:, 6:]).reshape(n_obs, 2, 9)
        E = np.where(cam_first, v[:, 18:], v[:, :6]).reshape(n_obs, 2, 3)
    e2 = (E * E).sum(axis=1)
    f2 = (F * F).sum(axis=1)
    for c in range(3):
        diag += np.bincount(col_pos[point_block] + c, weights=e2[:, c], minlength=bs.num_cols)
    for c in range(9):
        diag += np.bincount(col_pos[cam_block] + c, weights=f2[:, c], minlength=bs.num_cols)
    D = np.sqrt(np.clip(diag, 1e-6, 1e32) / 1e4)
    return LinearProblem(bs, values, b, D, nelim, {}, cam_block, point_block)


def structured_bal(n_cams, n_points, point_of_obs, cam_of_obs, camera_width=9, shared_widths=(), shared_first=True, locked_cameras=(),
                   shared_of_obs=None, layout="schur", seed=38401, with_values=True, point_width=3, row_height=2) -> LinearProblem:
    """Bundle-adjustment Jacobian of the structures the reference's examples produce beyond <2,3,9>: rows are 2 high and hold one point
    cell (2 x 3), the observing camera's cell (2 x camera_width; none if the camera is in `locked_cameras`: SetParameterBlockConstant,
    examples/libmv_bundle_adjuster.cc:725-728) and a cell on every SHARED block (`shared_widths`: libmv's camera intrinsics, <2, 8, 6, 3>,
    examples/libmv_bundle_adjuster.cc:700-722; shared_of_obs: optional (n_obs, n_shared) mask of which shared blocks a row references).
    Rows grouped by point (observations must come sorted by point).  layout "schur": column blocks = points, then the camera-side blocks in
    program order (shared blocks first, like libmv's intrinsics, or last), values E|F-split, the F cells of a row back to back in column
    order (internal/ceres/block_jacobian_writer.cc:68-167); "cgnr": the same column order, values row-sequential, no elimination.
    camera_width 10 = bundle_adjuster --use_quaternions (examples/snavely_reprojection_error.h:164).  point_width: the width of the point
    (E) blocks — 3, or 2 / 4 for the reference's (2,2,*) / (2,4,*) specialisations (generate_template_specializations.py:55-75; 4 =
    homogeneous points); row_height: residuals per observation — 2, or 3 / 4 for (3,3,3) and (4,4,*)."""
    pw, rh = int(point_width), int(row_height)
    rng = np.random.default_rng(seed)
    n_obs = int(point_of_obs.shape[0])
    nsb = len(shared_widths)
    locked = np.zeros(n_cams, dtype=bool)
    locked[list(locked_cameras)] = True
    # camera-side column blocks in program order
    f_sizes = ([int(w) for w in shared_widths] if shared_first else []) + [int(camera_width)] * n_cams + ([] if shared_first else [int(w) for w in shared_widths])
    sh0 = 0 if shared_first else n_cams          # F index of the first shared block
    cam0 = nsb if shared_first else 0            # F index of camera 0
    col_sizes = np.concatenate([np.full(n_points, pw, np.int64), np.asarray(f_sizes, np.int64)]).astype(np.int32)
    col_pos = np.concatenate([[0], np.cumsum(col_sizes.astype(np.int64))[:-1]])
    has_cam = ~locked[cam_of_obs]
    sh_mask = np.ones((n_obs, nsb), dtype=bool) if shared_of_obs is None else np.asarray(shared_of_obs, dtype=bool)
    # cells per row, in column order: point, [shared...], camera  (or camera before the shared blocks)
    per_row = 1 + has_cam.astype(np.int64) + sh_mask.sum(axis=1)
    ptr = np.concatenate([[0], np.cumsum(per_row)])
    n_cells = int(ptr[-1])
    cell_col = np.empty(n_cells, dtype=np.int64)
    cell_w = np.empty(n_cells, dtype=np.int64)
    cur = ptr[:-1].copy()
    cell_col[cur] = point_of_obs; cell_w[cur] = pw; cur += 1

    def put_shared():
        for q in range(nsb):
            m = sh_mask[:, q]
            cell_col[cur[m]] = n_points + sh0 + q; cell_w[cur[m]] = shared_widths[q]; cur[m] += 1

    def put_cameras():
        m = has_cam
        cell_col[cur[m]] = n_points + cam0 + cam_of_obs[m]; cell_w[cur[m]] = camera_width; cur[m] += 1
    if shared_first:
        put_shared(); put_cameras()
    else:
        put_cameras(); put_shared()
    cell_len = rh * cell_w
    is_e = np.zeros(n_cells, dtype=bool)
    is_e[ptr[:-1]] = True
    cell_pos = np.empty(n_cells, dtype=np.int64)
    if layout == "schur":
        cell_pos[is_e] = rh * pw * np.arange(n_obs)
        f_len = cell_len[~is_e]
        cell_pos[~is_e] = rh * pw * n_obs + np.concatenate([[0], np.cumsum(f_len)[:-1]])
        nelim = n_points
    elif layout == "cgnr":
        cell_pos[:] = np.concatenate([[0], np.cumsum(cell_len)[:-1]])
        nelim = 0
    else:
        raise ValueError(layout)
    nnz = int(cell_len.sum())
    r = np.arange(n_obs, dtype=np.int64)
    bs = BlockStructure(np.full(n_obs, rh, np.int32), rh * r, col_sizes, col_pos, ptr, cell_col, cell_pos)
    cam_block = np.where(has_cam, n_points + cam0 + cam_of_obs, -1)
    if not with_values:
        return LinearProblem(bs, np.zeros(0), np.zeros(0), None, nelim, {}, cam_block, point_of_obs.copy())
    values = rng.standard_normal(nnz)
    b = rng.standard_normal(rh * n_obs)
    diag = np.zeros(bs.num_cols)
    for w in np.unique(cell_w):   # column square sums, cells of one width at a time
        idx = np.flatnonzero(cell_w == w)
        v = values[(cell_pos[idx][:, None] + np.arange(rh * w)[None, :])].reshape(-1, rh, w)
        sq = (v * v).sum(axis=1)
        for c in range(int(w)):
            diag += np.bincount(col_pos[cell_col[idx]] + c, weights=sq[:, c], minlength=bs.num_cols)
    D = np.sqrt(np.clip(diag, 1e-6, 1e32) / 1e4)
    return LinearProblem(bs, values, b, D, nelim, {}, cam_block, point_of_obs.copy())


def synthetic_structured(num_cameras, num_points, num_observations, camera_width=9, shared_widths=(), shared_first=True, locked_cameras=(),
                         layout="schur", seed=38401, skew=0.0, with_values=True, point_width=3, row_height=2) -> LinearProblem:
    """structured_bal on the random visibility of synthetic_bal (every point seen by >= 2 distinct cameras)."""
    rng = np.random.default_rng(seed)
    k = _track_lengths(rng, num_cameras, num_points, num_observations)
    point_of_obs = np.repeat(np.arange(num_points, dtype=np.int64), k)
    weights = None
    if skew > 0:
        weights = np.arange(1, num_cameras + 1, dtype=np.float64) ** (-skew)
        weights /= weights.sum()
    cam_of_obs = _distinct_cameras(rng, num_cameras, point_of_obs, weights)
    order = np.lexsort((cam_of_obs, point_of_obs))
    return structured_bal(num_cameras, num_points, point_of_obs, cam_of_obs[order], camera_width, shared_widths, shared_first, locked_cameras,
                          None, layout, seed + 1, with_values, point_width, row_height)


def libmv_structured(problem=2, copies=1, intrinsics_width=8, lock_first_camera=True, layout="schur", seed=38401, with_values=True) -> LinearProblem:
    """The Jacobian STRUCTURE of examples/libmv_bundle_adjuster.cc on the real visibility of one of the reference's libmv problems:
    AutoDiffCostFunction<OpenCVReprojectionError, 2, 8, 6, 3> (:697-722) — every row references the ONE shared intrinsics block (8 wide,
    or the tangent size a SubsetManifold leaves, :754-771), the 6-wide pose of its camera and its 3-wide point; the first camera is
    held constant (:725-728), so its rows have no pose cell; points are elimination group 0 (ordering, :775-790), the intrinsics block
    is the first camera-side column block (it was added to the problem first).  N(0,1) values like the other synthetic workloads."""
    n_c, n_p, cam_of, pt_of = libmv_visibility(problem)
    reps = np.arange(copies, dtype=np.int64)
    point_of_obs = (pt_of[None, :] + n_p * reps[:, None]).reshape(-1)
    cam_of_obs = (cam_of[None, :] + n_c * reps[:, None]).reshape(-1)
    order = np.lexsort((cam_of_obs, point_of_obs))
    return structured_bal(n_c * copies, n_p * copies, point_of_obs[order], cam_of_obs[order], 6, (intrinsics_width,), True,
                          (0,) if lock_first_camera else (), None, layout, seed, with_values)


def add_camera_rows(prob: LinearProblem, num_rows: int, seed=0, row_size=9, pair_fraction=0.0, camera_width=9) -> LinearProblem:
    """Appends `num_rows` row blocks WITHOUT a point cell to a BAL-shaped problem (either layout): priors / regularisers on
    cameras — one `row_size` x camera_width cell on camera i mod n_cameras, or (a `pair_fraction` of the rows) two cells coupling two
    cameras.  The cameras are the column blocks behind the eliminated ones (Schur ordering), or, without an elimination order, the
    blocks that are camera_width wide.  In the Schur ordering these are the rows behind num_row_blocks_e that SchurEliminator::NoEBlockRowsUpdate
    (internal/ceres/schur_eliminator_impl.h:574-666) and PartitionedMatrixView's second loops
    (internal/ceres/partitioned_matrix_view_impl.h:171-190) handle.  Values are appended behind the existing ones."""
    rng = np.random.default_rng(seed + 991)
    bs = prob.bs
    w = int(camera_width)
    if prob.num_eliminate_blocks > 0:
        cam_blocks = np.arange(prob.num_eliminate_blocks, bs.num_col_blocks)
        cam_blocks = cam_blocks[bs.col_block_size[cam_blocks] == w]
    else:
        cam_blocks = np.flatnonzero(bs.col_block_size == w)
    n_cams = cam_blocks.shape[0]
    two = rng.random(num_rows) < pair_fraction
    c0 = cam_blocks[np.arange(num_rows) % n_cams]
    c1 = cam_blocks[(np.arange(num_rows) * 7 + 3) % n_cams]
    two &= c0 != c1
    ncell = np.where(two, 2, 1)
    row_sizes = np.full(num_rows, row_size, np.int32)
    row_pos = bs.num_rows + row_size * np.arange(num_rows, dtype=np.int64)
    extent0 = bs.values_extent() if bs.num_cells else 0
    cell_cols, cell_pos, pos = [], [], extent0
    for i in range(num_rows):
        cols = sorted({int(c0[i]), int(c1[i])}) if two[i] else [int(c0[i])]
        for c in cols:
            cell_cols.append(c)
            cell_pos.append(pos)
            pos += row_size * w
    new_bs = BlockStructure(np.concatenate([bs.row_block_size, row_sizes]), np.concatenate([bs.row_block_pos, row_pos]),
                            bs.col_block_size, bs.col_block_pos,
                            np.concatenate([bs.row_cell_ptr.astype(np.int64), bs.row_cell_ptr[-1] + np.cumsum(ncell)]),
                            np.concatenate([bs.cell_col_block, np.asarray(cell_cols, np.int32)]),
                            np.concatenate([bs.cell_value_pos, np.asarray(cell_pos, np.int64)]))
    if prob.values is None or prob.values.shape[0] == 0:
        return LinearProblem(new_bs, np.zeros(0), np.zeros(0), None, prob.num_eliminate_blocks, {}, prob.camera_of_row, prob.point_of_row)
    extra = rng.standard_normal(pos - extent0)
    values = np.concatenate([prob.values[:extent0], extra])
    b = np.concatenate([prob.b, rng.standard_normal(row_size * num_rows)])
    D = prob.D
    if D is not None:
        diag = (D * D) * 1e4
        cp = bs.col_block_pos.astype(np.int64)
        for k, c in enumerate(cell_cols):
            blk = extra[cell_pos[k] - extent0: cell_pos[k] - extent0 + row_size * w].reshape(row_size, w)
            diag[cp[c]: cp[c] + w] += (blk * blk).sum(0)
        D = np.sqrt(np.clip(diag, 1e-6, 1e32) / 1e4)
    return LinearProblem(new_bs, values, b, D, prob.num_eliminate_blocks, {}, prob.camera_of_row, prob.point_of_row)


def permute_rows(prob: LinearProblem, order) -> LinearProblem:
    """The same linear problem with its ROW BLOCKS in another order (`order[k]` = the old row block that becomes row block k): the
    value array is untouched (cells keep their positions, which BlockSparseMatrix allows: cell.position is data), the residuals move
    with their rows.  What a Program whose residual blocks were added in another order produces without an elimination ordering
    (CGNR: internal/ceres/block_jacobian_writer.cc:198-263 lays rows out in residual-block order) — e.g. a prior added together with
    its camera, in front of that camera's observations."""
    bs = prob.bs
    order = np.asarray(order, dtype=np.int64)
    assert np.array_equal(np.sort(order), np.arange(bs.num_row_blocks))
    ptr = bs.row_cell_ptr.astype(np.int64)
    sizes = bs.row_block_size[order]
    pos = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))[:-1]])
    ncell = (ptr[1:] - ptr[:-1])[order]
    new_ptr = np.concatenate([[0], np.cumsum(ncell)])
    cell_idx = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in order]) if len(order) else np.zeros(0, np.int64)
    new_bs = BlockStructure(sizes, pos, bs.col_block_size, bs.col_block_pos, new_ptr, bs.cell_col_block[cell_idx], bs.cell_value_pos[cell_idx])
    b = None
    if prob.b is not None and prob.b.shape[0]:
        old_pos = bs.row_block_pos.astype(np.int64)
        b = np.concatenate([prob.b[old_pos[r]: old_pos[r] + bs.row_block_size[r]] for r in order])
    return LinearProblem(new_bs, prob.values, b if b is not None else prob.b, prob.D, prob.num_eliminate_blocks, {}, None, None)


# --------------------------------------------------------------------------
# Scene-based values: the first LM linear system of a synthetic bundle-adjustment problem
# --------------------------------------------------------------------------
class _Dual:
    """Forward-mode dual numbers vectorised over observations: v (N,), d (N, 12) =
    d/d(camera[0..8], point[0..2]) — what AutoDiffCostFunction<SnavelyReprojectionError, 2, 9, 3>
    evaluates with Jets (examples/snavely_reprojection_error.h:53-105)."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v, self.d = v, d

    @staticmethod
    def var(v, k):
        d = np.zeros((v.shape[0], 12))
        d[:, k] = 1.0
        return _Dual(v, d)

    def _lift(self, o):
        return o if isinstance(o, _Dual) else _Dual(np.broadcast_to(np.asarray(o, dtype=np.float64), self.v.shape), 0.0)

    def __add__(self, o):
        o = self._lift(o)
        return _Dual(self.v + o.v, self.d + o.d)
    __radd__ = __add__

    def __sub__(self, o):
        o = self._lift(o)
        return _Dual(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        return self._lift(o) - self

    def __neg__(self):
        return _Dual(-self.v, -self.d)

    def __mul__(self, o):
        o = self._lift(o)
        return _Dual(self.v * o.v, self.d * o.v[:, None] + np.asarray(o.d) * self.v[:, None])
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._lift(o)
        q = self.v / o.v
        return _Dual(q, (self.d - np.asarray(o.d) * q[:, None]) / o.v[:, None])

    def sqrt(self):
        r = np.sqrt(self.v)
        return _Dual(r, self.d * (0.5 / r)[:, None])

    def sin(self):
        return _Dual(np.sin(self.v), self.d * np.cos(self.v)[:, None])

    def cos(self):
        return _Dual(np.cos(self.v), self.d * (-np.sin(self.v))[:, None])


def _snavely(cam, pt, dual):
    """Snavely projection (no observation subtracted).  cam (N,9), pt (N,3).  With dual=True
    returns (value (N,2), jacobian (N,2,12)), else value only.  Rotation: Rodrigues formula
    (include/ceres/rotation.h AngleAxisRotatePoint, theta != 0 branch)."""
    if dual:
        c = [_Dual.var(cam[:, i], i) for i in range(9)]
        p = [_Dual.var(pt[:, i], 9 + i) for i in range(3)]
        sqrt, sin, cos = (lambda a: a.sqrt()), (lambda a: a.sin()), (lambda a: a.cos())
    else:
        c = [cam[:, i] for i in range(9)]
        p = [pt[:, i] for i in range(3)]
        sqrt, sin, cos = np.sqrt, np.sin, np.cos
    theta = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])
    ct, st = cos(theta), sin(theta)
    w = [c[i] / theta for i in range(3)]
    wxp = [w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]]
    tmp = (w[0] * p[0] + w[1] * p[1] + w[2] * p[2]) * (1.0 - ct)
    q = [p[i] * ct + wxp[i] * st + w[i] * tmp + c[3 + i] for i in range(3)]
    xp, yp = -(q[0] / q[2]), -(q[1] / q[2])
    r2 = xp * xp + yp * yp
    dist = 1.0 + r2 * (c[7] + c[8] * r2)
    u, v = c[6] * dist * xp, c[6] * dist * yp
    if dual:
        return np.stack([u.v, v.v], 1), np.stack([u.d, v.d], 1)
    return np.stack([u, v], 1)


def _scene(rng, n_cams, n_points):
    """Cameras on a ring looking at a point cloud around the origin (BAL camera model: looks
    down -z, 9 parameters = angle-axis, translation, focal, k1, k2)."""
    ang = 2 * np.pi * (np.arange(n_cams) + 0.25 * rng.random(n_cams)) / n_cams
    rad = 8.0 + 4.0 * rng.random(n_cams)
    C = np.stack([rad * np.cos(ang), rad * np.sin(ang), 1.5 * rng.standard_normal(n_cams)], 1)
    z = C / np.linalg.norm(C, axis=1, keepdims=True)
    up = np.stack([0.05 * rng.standard_normal(n_cams), 0.05 * rng.standard_normal(n_cams), np.ones(n_cams)], 1)
    x = np.cross(up, z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 1)  # rows = camera axes
    tr = np.clip((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0) / 2.0, -1.0, 1.0)
    theta = np.arccos(tr)
    ax = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1)
    ax /= np.maximum(np.linalg.norm(ax, axis=1, keepdims=True), 1e-12)
    cams = np.zeros((n_cams, 9))
    cams[:, :3] = ax * theta[:, None]
    cams[:, 3:6] = -np.einsum("cij,cj->ci", R, C)
    cams[:, 6] = 800.0 + 400.0 * rng.random(n_cams)
    cams[:, 7] = 2e-2 * rng.standard_normal(n_cams)
    cams[:, 8] = 1e-3 * rng.standard_normal(n_cams)
    pts = np.stack([1.5 * rng.standard_normal(n_points), 1.5 * rng.standard_normal(n_points), 0.8 * rng.standard_normal(n_points)], 1)
    return cams, pts


def scene_values(prob: LinearProblem, n_cams: int, n_points: int, seed=38401, pixel_noise=0.5, param_noise=0.02,
                 jacobi_scaling=True, radius=1e4, chunk=400_000):
    """Replace the N(0,1) values of a BAL-shaped problem (either layout) by the Jacobian, residuals
    and LM diagonal of the FIRST trust-region step of a synthetic bundle-adjustment scene: J at the
    perturbed start, column-scaled as TrustRegionMinimizer does when jacobi_scaling is on
    (internal/ceres/trust_region_minimizer.cc:263-279), D = sqrt(clamp(diag(J^T J), 1e-6, 1e32) / radius)."""
    rng = np.random.default_rng(seed + 17)
    bs = prob.bs
    n_obs = bs.num_row_blocks
    pt_ids = np.unique(prob.point_of_row, return_inverse=True)[1]   # 0..n_points-1 in column order
    cam_ids = np.unique(prob.camera_of_row, return_inverse=True)[1]
    cams_true, pts_true = _scene(rng, n_cams, n_points)
    cams0 = cams_true.copy()
    cams0[:, :3] += 0.02 * param_noise * rng.standard_normal((n_cams, 3))
    cams0[:, 3:6] += 0.2 * param_noise * rng.standard_normal((n_cams, 3))
    cams0[:, 6] *= 1.0 + 0.02 * param_noise * rng.standard_normal(n_cams)
    pts0 = pts_true + param_noise * rng.standard_normal((n_points, 3))
    values = np.empty(24 * n_obs)
    b = np.empty(2 * n_obs)
    # where each row's E and F cells live
    first_is_pt = bs.col_block_size[bs.cell_col_block[0::2]] == 3
    epos = np.where(first_is_pt, bs.cell_value_pos[0::2], bs.cell_value_pos[1::2]).astype(np.int64)
    fpos = np.where(first_is_pt, bs.cell_value_pos[1::2], bs.cell_value_pos[0::2]).astype(np.int64)
    for lo in range(0, n_obs, chunk):
        hi = min(n_obs, lo + chunk)
        ci, pi = cam_ids[lo:hi], pt_ids[lo:hi]
        observed = _snavely(cams_true[ci], pts_true[pi], False) + pixel_noise * rng.standard_normal((hi - lo, 2))
        val, jac = _snavely(cams0[ci], pts0[pi], True)
        b[2 * lo:2 * hi] = (val - observed).reshape(-1)
        values[(epos[lo:hi, None] + np.arange(6)[None, :]).reshape(-1)] = jac[:, :, 9:].reshape(-1)
        values[(fpos[lo:hi, None] + np.arange(18)[None, :]).reshape(-1)] = jac[:, :, :9].reshape(-1)
    col_pos = bs.col_block_pos.astype(np.int64)
    pcol, ccol = col_pos[prob.point_of_row], col_pos[prob.camera_of_row]

    def colnorm2(vals):
        E = vals[(epos[:, None] + np.arange(6)[None, :])].reshape(n_obs, 2, 3)
        F = vals[(fpos[:, None] + np.arange(18)[None, :])].reshape(n_obs, 2, 9)
        d = np.zeros(bs.num_cols)
        e2, f2 = (E * E).sum(1), (F * F).sum(1)
        for k in range(3):
            d += np.bincount(pcol + k, weights=e2[:, k], minlength=bs.num_cols)
        for k in range(9):
            d += np.bincount(ccol + k, weights=f2[:, k], minlength=bs.num_cols)
        return d
    if jacobi_scaling:
        scale = 1.0 / (1.0 + np.sqrt(colnorm2(values)))
        se = scale[(pcol[:, None] + np.arange(3)[None, :])]            # (n_obs, 3)
        sf = scale[(ccol[:, None] + np.arange(9)[None, :])]            # (n_obs, 9)
        idx_e = (epos[:, None] + np.arange(6)[None, :])
        idx_f = (fpos[:, None] + np.arange(18)[None, :])
        values[idx_e] *= np.tile(se, (1, 2))
        values[idx_f] *= np.tile(sf, (1, 2))
    D = np.sqrt(np.clip(colnorm2(values), 1e-6, 1e32) / radius)
    prob.values, prob.b, prob.D = values, b, D
    return prob


def banded_bal(shape="dubrovnik16", seed=38401, num_cameras=None, num_points=None, num_observations=None, with_values=True) -> LinearProblem:
    """BAL-shaped <2,3,9> problem whose visibility is a SEQUENCE: every point is seen by a run of consecutive cameras (a track through
    consecutive frames of a turntable / video sequence; the camera ids wrap around), the same track-length distribution as
    synthetic_bal.  The camera graph is then a band (a ring of width ~ the track length) instead of an expander, and the reduced
    system is as badly conditioned as sequences are: SCHUR_JACOBI-preconditioned CG on the Snavely Jacobian of such a scene needs 11
    iterations at eta = 1e-3 and ~40 at 1e-4, where the randomly connected scene needs 4 and 7."""
    n_cams, n_points, n_obs = BAL_SHAPES[shape] if shape else (num_cameras, num_points, num_observations)
    if num_cameras is not None:
        n_cams, n_points, n_obs = num_cameras, num_points, num_observations
    rng = np.random.default_rng(seed)
    k = _track_lengths(rng, n_cams, n_points, n_obs)
    point_of_obs = np.repeat(np.arange(n_points, dtype=np.int64), k)
    start = rng.integers(0, n_cams, size=n_points)
    first = np.concatenate([[0], np.cumsum(k)[:-1]])
    offset = np.arange(int(k.sum()), dtype=np.int64) - np.repeat(first, k)
    cam_of_obs = (np.repeat(start, k) + offset) % n_cams
    order = np.lexsort((cam_of_obs, point_of_obs))
    return _assemble_bal(rng, n_cams, n_points, point_of_obs, cam_of_obs[order], "schur", with_values)


def bal_scene(shape="dubrovnik16", seed=38401, skew=0.0, num_cameras=None, num_points=None, num_observations=None,
              pixel_noise=0.5, param_noise=0.02, chunk=400_000, visibility="random"):
    """A synthetic bundle-adjustment problem in BAL form (examples/bal_problem.cc:75-135) with the
    observation graph of `synthetic_bal`: returns (num_cameras, num_points, camera_index, point_index,
    observations (n,2), parameters) — parameters in BAL file order (9 per camera, then 3 per point),
    perturbed away from the scene that generated the observations."""
    if visibility == "banded":   # a sequence: see banded_bal
        prob = banded_bal(shape, seed=seed, num_cameras=num_cameras, num_points=num_points, num_observations=num_observations, with_values=False)
    else:
        prob = synthetic_bal(shape, layout="schur", seed=seed, skew=skew, num_cameras=num_cameras, num_points=num_points,
                             num_observations=num_observations, with_values=False)
    n_points = int(prob.num_eliminate_blocks)
    n_cams = int(prob.bs.num_col_blocks - n_points)
    pt = np.asarray(prob.point_of_row, dtype=np.int64)
    cam = np.asarray(prob.camera_of_row, dtype=np.int64) - n_points
    n_obs = pt.shape[0]
    rng = np.random.default_rng(seed + 17)
    cams_true, pts_true = _scene(rng, n_cams, n_points)
    cams0 = cams_true.copy()
    cams0[:, :3] += 0.02 * param_noise * rng.standard_normal((n_cams, 3))
    cams0[:, 3:6] += 0.2 * param_noise * rng.standard_normal((n_cams, 3))
    cams0[:, 6] *= 1.0 + 0.02 * param_noise * rng.standard_normal(n_cams)
    pts0 = pts_true + param_noise * rng.standard_normal((n_points, 3))
    obs = np.empty((n_obs, 2))
    for lo in range(0, n_obs, chunk):
        hi = min(n_obs, lo + chunk)
        obs[lo:hi] = _snavely(cams_true[cam[lo:hi]], pts_true[pt[lo:hi]], False) + pixel_noise * rng.standard_normal((hi - lo, 2))
    return n_cams, n_points, cam.astype(np.int32), pt.astype(np.int32), obs, np.concatenate([cams0.reshape(-1), pts0.reshape(-1)])


def read_bal(filename):
    """BAL text format (examples/bal_problem.cc:75-135): "num_cameras num_points num_observations", then one
    "camera point x y" line per observation, then 9 doubles per camera and 3 per point, whitespace-separated.
    Returns the tuple `bal_scene` returns: (num_cameras, num_points, camera_index, point_index, observations, parameters)."""
    with open(filename, "rb") as f:
        tok = np.array(f.read().split())
    if tok.shape[0] < 3:
        raise ValueError(f"{filename}: not a BAL file")
    n_cams, n_points, n_obs = (int(t) for t in tok[:3])
    need = 3 + 4 * n_obs + 9 * n_cams + 3 * n_points
    if tok.shape[0] < need:
        raise ValueError(f"{filename}: {tok.shape[0]} tokens, the header promises {need}")
    body = tok[3:3 + 4 * n_obs].reshape(n_obs, 4)
    cam = body[:, 0].astype(np.int64).astype(np.int32)
    pt = body[:, 1].astype(np.int64).astype(np.int32)
    obs = body[:, 2:].astype(np.float64)
    params = tok[3 + 4 * n_obs:need].astype(np.float64)
    if n_obs and (cam.min() < 0 or cam.max() >= n_cams or pt.min() < 0 or pt.max() >= n_points):
        raise ValueError(f"{filename}: observation index out of range")
    return n_cams, n_points, cam, pt, np.ascontiguousarray(obs), params


def write_bal(filename, num_cameras, num_points, camera_index, point_index, observations, parameters):
    """Inverse of `read_bal` (BALProblem::WriteToFile, examples/bal_problem.cc:137-167)."""
    obs = np.asarray(observations, dtype=np.float64).reshape(-1, 2)
    with open(filename, "w") as f:
        f.write(f"{int(num_cameras)} {int(num_points)} {obs.shape[0]}\n")
        for c, q, (x, y) in zip(np.asarray(camera_index).tolist(), np.asarray(point_index).tolist(), obs.tolist()):
            f.write(f"{c} {q} {x:.16e} {y:.16e}\n")
        for v in np.asarray(parameters, dtype=np.float64).tolist():
            f.write(f"{v:.16e}\n")

