// flatten_block_structure.h — the one piece of the Ceres-side adapter that touches the reference's data structures: the
// vector-of-vectors CompressedRowBlockStructure (internal/ceres/block_structure.h:52-182) flattened into the plain arrays of
// ceres_hip_block_structure (include/ceres_hip.h).  A template on the structure type, so that the SAME code is what
//   * INTEGRATION.md's HipBlockSparseLinearSolver::SolveImpl runs on ceres::internal::CompressedRowBlockStructure,
//   * the C++ mirror (hip_linear_solver.h) runs on its dependency-free copy of that struct, and
//   * tests/test_integration_flatten_cpu.py compiles against the reference's REAL internal/ceres/block_structure.h (field names and
//     types checked by the compiler, a hand-built structure flattened and compared) wherever the reference tree is present.
// Needs: S::cols (size, position), S::rows (block.size, block.position, cells (block_id, position)).
#ifndef CERES_HIP_HOST_FLATTEN_BLOCK_STRUCTURE_H_
#define CERES_HIP_HOST_FLATTEN_BLOCK_STRUCTURE_H_

#include <cstdint>
#include <vector>

#include "ceres_hip.h"

namespace ceres_hip {

struct FlatBlockStructure {
  std::vector<int32_t> row_block_size, row_block_pos, col_block_size, col_block_pos, row_cell_ptr{0}, cell_col_block, cell_value_pos;
  // the view the C ABI takes (valid while this object lives and is not modified)
  ceres_hip_block_structure view() const {
    return ceres_hip_block_structure{static_cast<int32_t>(row_block_size.size()), static_cast<int32_t>(col_block_size.size()),
                                     row_block_size.data(), row_block_pos.data(), col_block_size.data(), col_block_pos.data(),
                                     row_cell_ptr.data(), cell_col_block.data(), cell_value_pos.data()};
  }
};

template <class CompressedRowBlockStructureT>
FlatBlockStructure FlattenBlockStructure(const CompressedRowBlockStructureT& bs) {
  FlatBlockStructure f;
  f.col_block_size.reserve(bs.cols.size());
  f.col_block_pos.reserve(bs.cols.size());
  for (const auto& c : bs.cols) {
    f.col_block_size.push_back(static_cast<int32_t>(c.size));
    f.col_block_pos.push_back(static_cast<int32_t>(c.position));
  }
  f.row_block_size.reserve(bs.rows.size());
  f.row_block_pos.reserve(bs.rows.size());
  f.row_cell_ptr.reserve(bs.rows.size() + 1);
  for (const auto& r : bs.rows) {
    f.row_block_size.push_back(static_cast<int32_t>(r.block.size));
    f.row_block_pos.push_back(static_cast<int32_t>(r.block.position));
    for (const auto& cell : r.cells) {
      f.cell_col_block.push_back(static_cast<int32_t>(cell.block_id));
      f.cell_value_pos.push_back(static_cast<int32_t>(cell.position));
    }
    f.row_cell_ptr.push_back(static_cast<int32_t>(f.cell_col_block.size()));
  }
  return f;
}

}  // namespace ceres_hip
#endif
