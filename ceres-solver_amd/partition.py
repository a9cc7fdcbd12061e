"""Sharding of a Schur-ordered Jacobian by point (E block) across the GPUs of one node.

SURVEY.md §8(e): E^T E is block diagonal and every chunk (the rows of one point) touches
only its own point and the cameras observing it, so contiguous ranges of chunks — balanced
by observation count, the same idea as the reference's cumulative-nnz partition
(internal/ceres/partitioned_matrix_view_impl.h:89-103) — give each rank a sub-Jacobian whose
E column blocks are disjoint from every other rank's and whose F (camera) column blocks are
the full replicated set.  Each rank hands ITS sub-structure to ceres_hip_set_structure; the
library sums camera-space quantities over ranks with RCCL.  Pure host logic (numpy), tested
on CPU with gloo in tests/test_distributed_cpu.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .block_structure import BlockStructure


@dataclass
class Shard:
    rank: int
    world_size: int
    bs: BlockStructure            # local structure: [local E blocks | all F blocks]
    num_eliminate_blocks: int     # local E blocks
    e_block_range: tuple          # [lo, hi) in the global E numbering
    row_block_range: tuple        # [lo, hi) global row blocks holding those E blocks
    extra_row_blocks: np.ndarray  # global E-free row blocks assigned to this rank
    value_index: np.ndarray       # values_local = values_global[value_index]
    row_index: np.ndarray         # b_local = b_global[row_index]
    col_index: np.ndarray         # D_local = D_global[col_index]; x_global[col_index] = x_local

    def local_values(self, values):
        return np.ascontiguousarray(values[self.value_index])

    def local_rows(self, b):
        return np.ascontiguousarray(b[self.row_index])

    def local_cols(self, D):
        return None if D is None else np.ascontiguousarray(D[self.col_index])


def split_points(bs: BlockStructure, num_eliminate_blocks: int, world_size: int) -> np.ndarray:
    """Boundaries (world_size + 1 E-block ids) of contiguous E-block ranges with nearly equal
    numbers of Jacobian values."""
    nelim = num_eliminate_blocks
    first_col = bs.cell_col_block[bs.row_cell_ptr[:-1].clip(max=max(bs.num_cells - 1, 0))] if bs.num_cells else np.zeros(0, np.int32)
    has_cells = np.diff(bs.row_cell_ptr) > 0
    e_of_row = np.where(has_cells & (first_col < nelim), first_col, -1)
    rows_of_cell = np.repeat(np.arange(bs.num_row_blocks), np.diff(bs.row_cell_ptr))
    cell_nnz = bs.row_block_size[rows_of_cell].astype(np.int64) * bs.col_block_size[bs.cell_col_block].astype(np.int64)
    row_nnz = np.bincount(rows_of_cell, weights=cell_nnz, minlength=bs.num_row_blocks)
    e_rows = e_of_row >= 0
    work = np.bincount(e_of_row[e_rows], weights=row_nnz[e_rows], minlength=nelim)
    cum = np.concatenate([[0.0], np.cumsum(work)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(cum, targets, side="left")
    bounds = np.maximum.accumulate(np.concatenate([[0], cuts, [nelim]]).astype(np.int64))
    # every rank gets at least one E block (a handful of points of very different track lengths would otherwise leave a rank without
    # eliminated blocks — ITERATIVE_SCHUR refuses such a shard and its peers wait for it until their exchange times out; found by
    # tools/fuzz_multirank.py).  Fewer E blocks than ranks cannot be sharded by point at all.
    if nelim < world_size:
        raise ValueError(f"{nelim} eliminated blocks cannot be sharded over {world_size} ranks (one point block per rank at least)")
    for r in range(1, world_size):          # forward: strictly increasing
        bounds[r] = max(bounds[r], bounds[r - 1] + 1)
    for r in range(world_size - 1, 0, -1):  # backward: room for the ranks behind
        bounds[r] = min(bounds[r], bounds[r + 1] - 1)
    return bounds


def shard_by_point(bs: BlockStructure, num_eliminate_blocks: int, world_size: int, rank: int) -> Shard:
    nelim = num_eliminate_blocks
    if nelim <= 0:
        raise ValueError("sharding needs the Schur ordering: eliminated (point) blocks first")
    bounds = split_points(bs, nelim, world_size)
    e_lo, e_hi = int(bounds[rank]), int(bounds[rank + 1])
    ptr = bs.row_cell_ptr.astype(np.int64)
    has_cells = np.diff(ptr) > 0
    first_col = np.full(bs.num_row_blocks, np.iinfo(np.int32).max, dtype=np.int64)
    first_col[has_cells] = bs.cell_col_block[ptr[:-1][has_cells]]
    is_e = first_col < nelim
    mine = is_e & (first_col >= e_lo) & (first_col < e_hi)
    e_rows = np.flatnonzero(mine)
    if len(e_rows) and not np.array_equal(e_rows, np.arange(e_rows[0], e_rows[-1] + 1)):
        raise ValueError("rows of an E-block range are not contiguous: not a Schur-ordered Jacobian")
    # E-free rows (regularisers): their contribution is additive, give them to the last rank
    extra = np.flatnonzero(~is_e) if rank == world_size - 1 else np.zeros(0, dtype=np.int64)
    rows = np.concatenate([e_rows, extra]).astype(np.int64)

    n_f = bs.num_col_blocks - nelim
    n_loc_e = e_hi - e_lo
    col_blocks = np.concatenate([np.arange(e_lo, e_hi), np.arange(nelim, bs.num_col_blocks)]).astype(np.int64)
    col_size = bs.col_block_size[col_blocks]
    col_pos = np.concatenate([[0], np.cumsum(col_size.astype(np.int64))[:-1]])
    # global column block -> local column block
    remap = np.full(bs.num_col_blocks, -1, dtype=np.int64)
    remap[col_blocks] = np.arange(len(col_blocks))

    row_size = bs.row_block_size[rows]
    row_pos = np.concatenate([[0], np.cumsum(row_size.astype(np.int64))[:-1]])
    counts = (ptr[rows + 1] - ptr[rows])
    new_ptr = np.concatenate([[0], np.cumsum(counts)])
    # cells of the selected rows, in order
    cell_ids = np.repeat(ptr[rows], counts) + (np.arange(new_ptr[-1]) - np.repeat(new_ptr[:-1], counts))
    cell_col = remap[bs.cell_col_block[cell_ids]]
    if (cell_col < 0).any():
        raise ValueError("a selected row touches an E block of another rank")
    cell_rows_local = np.repeat(np.arange(len(rows)), counts)
    cell_len = row_size[cell_rows_local].astype(np.int64) * col_size[cell_col].astype(np.int64)
    new_pos = np.concatenate([[0], np.cumsum(cell_len)[:-1]]) if len(cell_len) else np.zeros(0, np.int64)
    old_pos = bs.cell_value_pos[cell_ids].astype(np.int64)
    total = int(cell_len.sum())
    value_index = np.repeat(old_pos - new_pos, cell_len) + np.arange(total)
    row_index = np.repeat(bs.row_block_pos[rows].astype(np.int64) - row_pos, row_size) + np.arange(int(row_size.sum()))
    gcol_pos = bs.col_block_pos[col_blocks].astype(np.int64)
    col_index = np.repeat(gcol_pos - col_pos, col_size) + np.arange(int(col_size.sum()))

    local = BlockStructure(row_size, row_pos, col_size, col_pos, new_ptr, cell_col, new_pos)
    r_lo = int(e_rows[0]) if len(e_rows) else 0
    r_hi = int(e_rows[-1]) + 1 if len(e_rows) else 0
    return Shard(rank, world_size, local, n_loc_e, (e_lo, e_hi), (r_lo, r_hi), extra, value_index, row_index, col_index)


def assemble_solution(shards, local_solutions, num_cols):
    """x_global from the per-rank x_local (camera part taken from rank 0; all ranks agree)."""
    x = np.full(num_cols, np.nan)
    for sh, xl in zip(shards, local_solutions):
        n_e = int(sh.bs.col_block_size[: sh.num_eliminate_blocks].sum())
        x[sh.col_index[:n_e]] = xl[:n_e]
    sh0 = shards[0]
    n_e0 = int(sh0.bs.col_block_size[: sh0.num_eliminate_blocks].sum())
    x[sh0.col_index[n_e0:]] = local_solutions[0][n_e0:]
    return x
