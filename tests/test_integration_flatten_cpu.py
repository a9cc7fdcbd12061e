"""INTEGRATION.md's structure flattening against the reference's REAL internal/ceres/block_structure.h: the code
the adapter runs (ceres-solver_amd/host/flatten_block_structure.h, a template on the structure type) is compiled with g++ against the
header where it lies under /root/reference and run on a hand-built structure.  The header needs <vector>, <cstdint> and
"ceres/internal/export.h" — a file Ceres' CMake GENERATES (visibility macros only); the test writes an empty-macro one into a
temporary directory.  Nothing of the reference is copied; where the reference tree is absent (the GPU box) the test is skipped."""
import os
import subprocess
import textwrap

import pytest

from conftest import ROOT

REF = "/root/reference"
HEADER = os.path.join(REF, "internal", "ceres", "block_structure.h")


@pytest.mark.skipif(not os.path.exists(HEADER), reason="the reference tree is not present on this box")
def test_flattening_compiles_and_runs_against_the_real_block_structure_header(tmp_path):
    inc = tmp_path / "gen" / "ceres" / "internal"
    inc.mkdir(parents=True)
    (inc / "export.h").write_text("#ifndef CERES_EXPORT\n#define CERES_EXPORT\n#define CERES_NO_EXPORT\n#endif\n")
    src = tmp_path / "flatten_check.cc"
    src.write_text(textwrap.dedent("""
        #include <cstdio>
        #include "ceres/block_structure.h"          // the reference's own header
        #include "flatten_block_structure.h"        // what the adapter runs
        using ceres::internal::Block;
        using ceres::internal::Cell;
        using ceres::internal::CompressedRow;
        using ceres::internal::CompressedRowBlockStructure;
        int main() {
          // two 3-wide points, two 9-wide cameras, three 2-high observation rows, E|F-split values
          CompressedRowBlockStructure bs;
          bs.cols = {Block(3, 0), Block(3, 3), Block(9, 6), Block(9, 15)};
          const int pt[3] = {0, 0, 1}, cam[3] = {2, 3, 2};
          for (int r = 0; r < 3; ++r) {
            CompressedRow row(2);
            row.block = Block(2, 2 * r);
            row.cells[0] = Cell(pt[r], 6 * r);
            row.cells[1] = Cell(cam[r], 18 + 18 * r);
            bs.rows.push_back(row);
          }
          const ceres_hip::FlatBlockStructure f = ceres_hip::FlattenBlockStructure(bs);
          const ceres_hip_block_structure v = f.view();
          bool ok = v.num_row_blocks == 3 && v.num_col_blocks == 4;
          const int want_ptr[4] = {0, 2, 4, 6}, want_col[6] = {0, 2, 0, 3, 1, 2}, want_val[6] = {0, 18, 6, 36, 12, 54};
          for (int i = 0; i < 4; ++i) ok = ok && v.row_cell_ptr[i] == want_ptr[i];
          for (int i = 0; i < 6; ++i) ok = ok && v.cell_col_block[i] == want_col[i] && v.cell_value_pos[i] == want_val[i];
          for (int r = 0; r < 3; ++r) ok = ok && v.row_block_size[r] == 2 && v.row_block_pos[r] == 2 * r;
          ok = ok && v.col_block_size[2] == 9 && v.col_block_pos[3] == 15;
          std::printf(ok ? "flatten ok\\n" : "flatten MISMATCH\\n");
          return ok ? 0 : 1;
        }
        """))
    exe = tmp_path / "flatten_check"
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(REF, "internal"), "-I", str(tmp_path / "gen"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "ceres-solver_amd", "host"), str(src), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "flatten ok" in r.stdout, r.stdout + r.stderr
