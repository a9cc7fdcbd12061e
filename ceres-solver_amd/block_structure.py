"""Host-side mirror of the reference's block-sparse Jacobian description.

`BlockStructure` is the flattened form of `CompressedRowBlockStructure`
(reference internal/ceres/block_structure.h:52-182) that crosses the C ABI
(`ceres_hip_block_structure` in include/ceres_hip.h): SoA int32 arrays instead of
a vector of per-row heap vectors.  Names follow the reference: cols[j] =
(size, position), rows[i].block = (size, position), rows[i].cells[k] =
(block_id, position).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np


class CBlockStructure(ctypes.Structure):
    """ctypes image of ceres_hip_block_structure."""

    _fields_ = [
        ("num_row_blocks", ctypes.c_int32),
        ("num_col_blocks", ctypes.c_int32),
        ("row_block_size", ctypes.POINTER(ctypes.c_int32)),
        ("row_block_pos", ctypes.POINTER(ctypes.c_int32)),
        ("col_block_size", ctypes.POINTER(ctypes.c_int32)),
        ("col_block_pos", ctypes.POINTER(ctypes.c_int32)),
        ("row_cell_ptr", ctypes.POINTER(ctypes.c_int32)),
        ("cell_col_block", ctypes.POINTER(ctypes.c_int32)),
        ("cell_value_pos", ctypes.POINTER(ctypes.c_int32)),
    ]


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


@dataclass
class BlockStructure:
    row_block_size: np.ndarray
    row_block_pos: np.ndarray
    col_block_size: np.ndarray
    col_block_pos: np.ndarray
    row_cell_ptr: np.ndarray
    cell_col_block: np.ndarray
    cell_value_pos: np.ndarray

    def __post_init__(self):
        for name in ("row_block_size", "row_block_pos", "col_block_size", "col_block_pos",
                     "row_cell_ptr", "cell_col_block", "cell_value_pos"):
            setattr(self, name, _i32(getattr(self, name)))
        if self.row_cell_ptr.shape[0] != self.num_row_blocks + 1:
            raise ValueError("row_cell_ptr must have num_row_blocks + 1 entries")

    # -- construction ------------------------------------------------------
    @classmethod
    def from_rows(cls, col_sizes, rows):
        """rows: list of (row_block_size, [(col_block_id, value_position), ...]);
        row and column positions are the running sums, as in the reference."""
        col_sizes = _i32(col_sizes)
        col_pos = np.concatenate([[0], np.cumsum(col_sizes)[:-1]]) if len(col_sizes) else np.zeros(0)
        rsz, rpos, ptr, ccol, cpos = [], [], [0], [], []
        pos = 0
        for size, cells in rows:
            rsz.append(size)
            rpos.append(pos)
            pos += size
            for block_id, position in cells:
                ccol.append(block_id)
                cpos.append(position)
            ptr.append(len(ccol))
        return cls(rsz, rpos, col_sizes, col_pos, ptr, ccol, cpos)

    # -- derived sizes -----------------------------------------------------
    @property
    def num_row_blocks(self) -> int:
        return int(self.row_block_size.shape[0])

    @property
    def num_col_blocks(self) -> int:
        return int(self.col_block_size.shape[0])

    @property
    def num_cells(self) -> int:
        return int(self.cell_col_block.shape[0])

    @property
    def num_rows(self) -> int:
        return int(self.row_block_size.sum())

    @property
    def num_cols(self) -> int:
        return int(self.col_block_size.sum())

    @property
    def num_nonzeros(self) -> int:
        rows_of_cell = np.repeat(np.arange(self.num_row_blocks), np.diff(self.row_cell_ptr))
        return int((self.row_block_size[rows_of_cell].astype(np.int64)
                    * self.col_block_size[self.cell_col_block].astype(np.int64)).sum())

    def values_extent(self) -> int:
        """Smallest length of a values array that every cell fits in."""
        if self.num_cells == 0:
            return 0
        rows_of_cell = np.repeat(np.arange(self.num_row_blocks), np.diff(self.row_cell_ptr))
        ends = (self.cell_value_pos.astype(np.int64)
                + self.row_block_size[rows_of_cell].astype(np.int64)
                * self.col_block_size[self.cell_col_block].astype(np.int64))
        return int(ends.max())

    # -- ABI ---------------------------------------------------------------
    def as_ctypes(self) -> CBlockStructure:
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        c = CBlockStructure(self.num_row_blocks, self.num_col_blocks, p(self.row_block_size),
                            p(self.row_block_pos), p(self.col_block_size), p(self.col_block_pos),
                            p(self.row_cell_ptr), p(self.cell_col_block), p(self.cell_value_pos))
        c._keepalive = self  # the arrays must outlive the struct
        return c

    # -- dense image (tests only; small problems) ---------------------------
    def to_dense(self, values: np.ndarray) -> np.ndarray:
        a = np.zeros((self.num_rows, self.num_cols))
        for i in range(self.num_row_blocks):
            rs, rp = int(self.row_block_size[i]), int(self.row_block_pos[i])
            for k in range(int(self.row_cell_ptr[i]), int(self.row_cell_ptr[i + 1])):
                j = int(self.cell_col_block[k])
                cs, cp, vp = int(self.col_block_size[j]), int(self.col_block_pos[j]), int(self.cell_value_pos[k])
                a[rp:rp + rs, cp:cp + cs] = values[vp:vp + rs * cs].reshape(rs, cs)
        return a
