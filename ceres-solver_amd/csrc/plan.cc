// plan.cc — host analysis of the block structure (no HIP in this file).
//
// What the reference spreads over BlockSparseMatrix's ctor (transpose structure,
// I/block_sparse_matrix.cc:178-216,784-808), PartitionedMatrixView's ctor
// (num_row_blocks_e, I/partitioned_matrix_view_impl.h:47-105), DetectStructure
// (I/detect_structure.cc:39-121) and SchurEliminator::Init (chunks,
// I/schur_eliminator_impl.h:87-181) happens here once per solver instance, and the
// result is flattened to SoA int32 arrays for upload.  On top of that the <2,3,9>
// case gets a packing plan: observations are grouped by point into 64-slot tiles
// (one wavefront each) such that no point straddles a tile unless it has more than
// 64 observations, which is what lets the fused kernels do every per-point
// reduction with wavefront shuffles and no inter-workgroup traffic.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "common.h"

namespace chip {

// CERES_HIP_PLAN_TIMING=1: seconds since the previous mark, to stderr (where set_structure's host time goes)
namespace {
struct PlanClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  PlanClock() : on(getenv("CERES_HIP_PLAN_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[plan] %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(n - t).count());
    t = n;
  }
};
}  // namespace

std::string AnalyzeStructure(const ceres_hip_block_structure& bs, int nelim, HostStructure* hs) {
  HostStructure& h = *hs;
  h = HostStructure();
  if (bs.num_row_blocks < 0 || bs.num_col_blocks < 0) return "negative block count";
  h.nrb = bs.num_row_blocks;
  h.ncb = bs.num_col_blocks;
  if (nelim < 0 || nelim > h.ncb) return "num_eliminate_blocks out of range";
  h.nelim = nelim;
  h.rsz.assign(bs.row_block_size, bs.row_block_size + h.nrb);
  h.rpos.assign(bs.row_block_pos, bs.row_block_pos + h.nrb);
  h.csz.assign(bs.col_block_size, bs.col_block_size + h.ncb);
  h.cpos.assign(bs.col_block_pos, bs.col_block_pos + h.ncb);
  h.rptr.assign(bs.row_cell_ptr, bs.row_cell_ptr + h.nrb + 1);
  if (h.rptr[0] != 0) return "row_cell_ptr[0] != 0";
  for (int i = 0; i < h.nrb; ++i)
    if (h.rptr[i + 1] < h.rptr[i]) return "row_cell_ptr not monotone";
  h.ncells = h.rptr[h.nrb];
  h.ccol.assign(bs.cell_col_block, bs.cell_col_block + h.ncells);
  h.cval.assign(bs.cell_value_pos, bs.cell_value_pos + h.ncells);

  // Rows and columns must tile [0, num_rows) / [0, num_cols) in order, as every
  // BlockSparseMatrix does (I/block_sparse_matrix.cc:178-216).
  // (positions are `int` in the reference's CompressedRowBlockStructure: sums that leave that range are refused, not wrapped)
  int64_t pos = 0;
  for (int i = 0; i < h.nrb; ++i) {
    if (h.rsz[i] <= 0) return "row block with non-positive size";
    if (h.rpos[i] != pos) return "row block positions are not the running sum of sizes";
    pos += h.rsz[i];
    if (pos > INT32_MAX) return "more than 2^31 - 1 rows";
    h.max_block = std::max(h.max_block, h.rsz[i]);
  }
  h.num_rows = int(pos);
  pos = 0;
  for (int j = 0; j < h.ncb; ++j) {
    if (h.csz[j] <= 0) return "column block with non-positive size";
    if (h.cpos[j] != pos) return "column block positions are not the running sum of sizes";
    pos += h.csz[j];
    if (pos > INT32_MAX) return "more than 2^31 - 1 columns";
    h.max_block = std::max(h.max_block, h.csz[j]);
    (j < nelim ? h.num_cols_e : h.num_cols_f) += h.csz[j];
  }
  h.num_cols = int(pos);
  h.row_block_of.resize(h.num_rows);
  for (int i = 0; i < h.nrb; ++i) std::fill_n(h.row_block_of.begin() + h.rpos[i], h.rsz[i], i);
  h.col_block_of.resize(h.num_cols);
  for (int j = 0; j < h.ncb; ++j) std::fill_n(h.col_block_of.begin() + h.cpos[j], h.csz[j], j);

  for (int i = 0; i < h.nrb; ++i)
    for (int k = h.rptr[i]; k < h.rptr[i + 1]; ++k) {
      const int j = h.ccol[k];
      if (j < 0 || j >= h.ncb) return "cell with column block id out of range";
      if (h.cval[k] < 0) return "cell with negative value position";
      const int64_t n = int64_t(h.rsz[i]) * h.csz[j];
      if (int64_t(h.cval[k]) + n - 1 > INT32_MAX) return "cell beyond the range of int value positions";
      h.nnz += n;
      h.values_extent = std::max(h.values_extent, int64_t(h.cval[k]) + n);
    }

  // Diagonal block stores.
  h.diag_off_all.assign(h.ncb + 1, 0);
  for (int j = 0; j < h.ncb; ++j) h.diag_off_all[j + 1] = h.diag_off_all[j] + int64_t(h.csz[j]) * h.csz[j];
  h.diag_off_e.assign(nelim + 1, 0);
  for (int j = 0; j < nelim; ++j) h.diag_off_e[j + 1] = h.diag_off_e[j] + int64_t(h.csz[j]) * h.csz[j];
  h.diag_off_f.assign(h.ncb - nelim + 1, 0);
  for (int j = nelim; j < h.ncb; ++j)
    h.diag_off_f[j - nelim + 1] = h.diag_off_f[j - nelim] + int64_t(h.csz[j]) * h.csz[j];

  // Transpose: counting sort by column block keeps row order inside a column.
  h.tptr.assign(h.ncb + 1, 0);
  for (int k = 0; k < h.ncells; ++k) ++h.tptr[h.ccol[k] + 1];
  for (int j = 0; j < h.ncb; ++j) h.tptr[j + 1] += h.tptr[j];
  h.trow.resize(h.ncells);
  h.tcell.resize(h.ncells);
  {
    std::vector<int32_t> cur(h.tptr.begin(), h.tptr.end() - 1);
    for (int i = 0; i < h.nrb; ++i)
      for (int k = h.rptr[i]; k < h.rptr[i + 1]; ++k) {
        const int p = cur[h.ccol[k]]++;
        h.trow[p] = i;
        h.tcell[p] = k;
      }
  }

  // E rows: first cell in an eliminated block.  They must precede all other rows
  // (PartitionedMatrixView indexes "the first num_row_blocks_e_ rows") and the rows
  // of one E block must be contiguous (SchurEliminator's chunks).
  h.row_e_block.assign(h.nrb, -1);
  h.chunk_start.assign(nelim, 0);
  h.chunk_size.assign(nelim, 0);
  bool seen_non_e = false;
  for (int i = 0; i < h.nrb; ++i) {
    const bool is_e = h.rptr[i] < h.rptr[i + 1] && h.ccol[h.rptr[i]] < nelim;
    if (!is_e) { seen_non_e = true; continue; }
    if (seen_non_e) h.chunks_contiguous = false;
    const int e = h.ccol[h.rptr[i]];
    h.row_e_block[i] = e;
    ++h.num_row_blocks_e;
    if (h.chunk_size[e] == 0) h.chunk_start[e] = i;
    else if (h.chunk_start[e] + h.chunk_size[e] != i) h.chunks_contiguous = false;
    ++h.chunk_size[e];
    for (int k = h.rptr[i] + 1; k < h.rptr[i + 1]; ++k)
      if (h.ccol[k] < nelim) return "row with more than one cell in the eliminated (E) blocks";
  }

  // DetectStructure: only E rows vote; two different values make a size dynamic (-1).
  auto vote = [](int* cur, int v) { if (*cur == 0) *cur = v; else if (*cur != -1 && *cur != v) *cur = -1; };
  for (int i = 0; i < h.nrb; ++i) {
    if (h.row_e_block[i] < 0) continue;
    vote(&h.det_row, h.rsz[i]);
    vote(&h.det_e, h.csz[h.row_e_block[i]]);
    for (int k = h.rptr[i] + 1; k < h.rptr[i + 1]; ++k) vote(&h.det_f, h.csz[h.ccol[k]]);
  }
  return "";
}

void BuildBalPlan(const HostStructure& h, int reorder_mode, const HybridRequest& hyb_in, BalPlan* plan) {
  BalPlan& P = *plan;
  P = BalPlan();
  auto no = [&](const char* why) { P.eligible = false; P.why_not = why; };
  if (h.nrb == 0) return no("empty matrix");
  PlanClock clk;
  // Classify column blocks: POINTS (the eliminated blocks; without an elimination order: the 3-wide ones), and among the others the
  // CAMERAS (at most one cell per row, all of one width nf) and a few SHARED blocks (common.h: the strip).
  // The eliminated blocks are all of ONE width: 3, or — with an elimination order — 2 or 4 (common.h: BalShapeCompiled; the reference's
  // (2,2,*) and (2,4,*) specialisations).
  P.ne = h.nelim > 0 ? h.csz[0] : 3;
  if (P.ne < 2 || P.ne > 4) return no("eliminated blocks that are not 2, 3 or 4 wide");
  auto is_point = [&](int j) { return h.nelim > 0 ? j < h.nelim : h.csz[j] == 3; };
  for (int j = 0; j < h.ncb; ++j)
    if (is_point(j) && h.csz[j] != P.ne) return no("eliminated blocks of different widths");

  // Remainder: the rows that touch camera-side blocks only (no point cell; possibly no cell at all).  A conforming row has a point cell,
  // so the split is unambiguous.  With an elimination order they must TRAIL (the reference's E rows come first: an E row behind an
  // E-free row is not a structure its Schur solvers accept, I/partitioned_matrix_view_impl.h:124-136); without one (CGNR: rows in
  // the order the residual blocks were added) they may sit ANYWHERE among the observation rows (round 5) — a prior added together with
  // its camera, in front of that camera's observations.  `orig[q]` = the row block of conforming row q (compact ids from here on).
  std::vector<int32_t> orig;
  orig.reserve(h.nrb);
  P.rem_list.clear();
  for (int i = 0; i < h.nrb; ++i) {
    bool camera_only = true;
    for (int k = h.rptr[i]; k < h.rptr[i + 1] && camera_only; ++k) camera_only = !is_point(h.ccol[k]);
    if (!camera_only) { orig.push_back(i); continue; }
    if (h.rsz[i] > kMaxGenericBlock) return no("a row without a point cell is higher than the generic kernels take");
    P.rem_list.push_back(i);
  }
  const int n_conf = int(orig.size());
  if (n_conf == 0) return no("no row with a point cell");
  P.nr = h.rsz[orig[0]];   // every conforming row is this high (checked below): 2, or 3 / 4 for the reference's (3,3,3) and (4,4,*)
  if (P.nr < 2 || P.nr > 4) return no("row blocks that are not 2, 3 or 4 high");
  const bool rem_trailing = P.rem_list.empty() || P.rem_list.front() == n_conf;
  if (h.nelim > 0 && !rem_trailing) return no("a row without a point cell in front of rows with one (Schur ordering)");
  P.rem_row0 = rem_trailing ? n_conf : -1;
  P.n_rem_rows = int(P.rem_list.size());

  // Shared blocks: as long as some row holds more than one cell outside the points and the shared set, the most referenced block
  // among those rows' cells joins the shared set (libmv: the one intrinsics block every row references).
  std::vector<uint8_t> shared(h.ncb, 0);
  {
    std::vector<int32_t> refs(h.ncb, 0);
    for (int q = 0; q < n_conf; ++q)
      for (int k = h.rptr[orig[q]]; k < h.rptr[orig[q] + 1]; ++k) ++refs[h.ccol[k]];
    int n_shared_scalars = 0;
    for (int round = 0;; ++round) {
      int pick = -1;
      for (int q = 0; q < n_conf; ++q) {
        const int i = orig[q];
        int others = 0;
        for (int k = h.rptr[i]; k < h.rptr[i + 1]; ++k) others += !is_point(h.ccol[k]) && !shared[h.ccol[k]];
        if (others < 2) continue;
        for (int k = h.rptr[i]; k < h.rptr[i + 1]; ++k) {
          const int j = h.ccol[k];
          if (!is_point(j) && !shared[j] && (pick < 0 || refs[j] > refs[pick])) pick = j;
        }
      }
      if (pick < 0) break;
      shared[pick] = 1;
      n_shared_scalars += h.csz[pick];
      if (n_shared_scalars > kMaxSharedScalars) return no("rows with several camera-side cells whose common blocks are wider than the shared strip");
    }
  }
  P.pt_block.clear();
  P.cam_block.clear();
  std::vector<int32_t> id_of(h.ncb, -1);  // point id, camera id or shared-block index of a column block
  for (int j = 0; j < h.ncb; ++j) {
    if (is_point(j)) { id_of[j] = int(P.pt_block.size()); P.pt_block.push_back(j); }
    else if (shared[j]) { id_of[j] = int(P.sh_block.size()); P.sh_off.push_back(P.ns_used); P.sh_block.push_back(j); P.ns_used += h.csz[j]; }
    else { id_of[j] = int(P.cam_block.size()); P.cam_block.push_back(j); }
  }
  P.n_points = int(P.pt_block.size());
  P.n_cameras = int(P.cam_block.size());
  if (P.n_points == 0 || P.n_cameras == 0) return no("no point or no camera blocks");
  P.nf = h.csz[P.cam_block[0]];
  for (int c = 0; c < P.n_cameras; ++c)
    if (h.csz[P.cam_block[c]] != P.nf) return no("camera blocks of different widths");
  P.ns = BalStripWidthFor(P.ns_used);
  if (P.ns < 0 || !BalShapeCompiled(P.nr, P.ne, P.nf, P.ns)) return no("no fused kernels are compiled for this row height / point width / camera width / shared strip");
  // (kernels_generic.hip: rem_* are templated on the camera width — any compiled shape; a remainder row's cell on a SHARED block would
  // have to join the strip's sums, which those kernels do not form)
  if (P.n_rem_rows > 0 && P.ns != 0) return no("rows without a point cell next to a shared strip");
  for (size_t q = 0; q < P.sh_block.size(); ++q)
    for (int k = 0; k < h.csz[P.sh_block[q]]; ++k) P.sh_pos.push_back(h.cpos[P.sh_block[q]] - h.num_cols_e + k);

  clk.mark("classify, shared blocks");
  // Every conforming row: 2 scalar rows, exactly one point cell, at most one camera cell, at most kMaxSharedCellsPerRow shared cells.
  std::vector<int32_t> row_pt(n_conf), row_cam(n_conf, -1), row_epos(n_conf), row_fpos(n_conf, -1);
  std::vector<int32_t> row_hpos[kMaxSharedCellsPerRow], row_hdesc[kMaxSharedCellsPerRow];
  if (P.ns > 0)
    for (int q = 0; q < kMaxSharedCellsPerRow; ++q) { row_hpos[q].assign(n_conf, -1); row_hdesc[q].assign(n_conf, 0); }
  for (int i = 0; i < n_conf; ++i) {   // i: compact id of the conforming row, ro: its row block
    const int ro = orig[i];
    if (h.rsz[ro] != P.nr) return no(P.nr == 2 ? "row block that is not 2 high" : "row blocks of different heights");
    int n_pt = 0, n_cam = 0, n_sh = 0;
    for (int k = h.rptr[ro]; k < h.rptr[ro + 1]; ++k) {
      const int j = h.ccol[k];
      if (is_point(j)) {
        if (h.nelim > 0 && k != h.rptr[ro]) return no("E cell is not the first cell of its row");
        if (n_pt++) return no("row with two point cells");
        row_pt[i] = id_of[j]; row_epos[i] = h.cval[k];
      } else if (shared[j]) {
        if (n_sh >= kMaxSharedCellsPerRow) return no("row with more shared cells than the tiles take");
        for (int q = 0; q < n_sh; ++q)
          if ((row_hdesc[q][i] & 0xff) == P.sh_off[id_of[j]]) return no("row with two cells on one shared block");
        row_hpos[n_sh][i] = h.cval[k];
        row_hdesc[n_sh][i] = P.sh_off[id_of[j]] | (h.csz[j] << 8);
        ++n_sh;
      } else {
        if (n_cam++) return no("row with two camera cells");   // (cannot happen: the shared set absorbed one of them)
        row_cam[i] = id_of[j]; row_fpos[i] = h.cval[k];
      }
    }
    if (n_pt != 1) return no("row is not one point cell plus camera-side cells");
  }
  if (h.nelim > 0 && !h.chunks_contiguous) return no("rows of one E block are not contiguous");

  clk.mark("rows");
  // Observations grouped by point (stable: keeps the caller's order inside a point).
  std::vector<int32_t> order(n_conf);
  std::iota(order.begin(), order.end(), 0);
  bool sorted = true;
  for (int i = 1; i < n_conf && sorted; ++i) sorted = row_pt[i - 1] <= row_pt[i];
  if (!sorted) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return row_pt[a] < row_pt[b]; });
  std::vector<int32_t> track(P.n_points, 0);
  for (int i = 0; i < n_conf; ++i) ++track[row_pt[i]];
  for (int p = 0; p < P.n_points; ++p) {
    if (track[p] == 0) return no("point without observations");
    P.max_track = std::max(P.max_track, track[p]);
  }
  // Distinct cameras inside a point (the fused SCHUR_JACOBI kernel relies on it; BAL
  // always satisfies it).
  {
    std::vector<int32_t> last_point_of_cam(P.n_cameras, -1);
    for (int idx = 0; idx < n_conf; ++idx) {
      const int i = order[idx];
      if (row_cam[i] < 0) continue;   // (a row without a camera cell: a locked camera, libmv_bundle_adjuster.cc:725-728)
      if (last_point_of_cam[row_cam[i]] == row_pt[i]) return no("a point observes one camera twice");
      last_point_of_cam[row_cam[i]] = row_pt[i];
      ++P.n_cam_cells;
    }
  }
  P.n_obs = n_conf;

  // rows of each point (contiguous in `order`)
  std::vector<int32_t> row_start(P.n_points + 1, 0);
  for (int p = 0; p < P.n_points; ++p) row_start[p + 1] = row_start[p] + track[p];

  // the caller's column layout: points-then-cameras, back to back?
  P.caller_contiguous = true;
  for (int p = 0; p < P.n_points && P.caller_contiguous; ++p) P.caller_contiguous = h.cpos[P.pt_block[p]] == P.ne * p;
  for (int c = 0; c < P.n_cameras && P.caller_contiguous; ++c) P.caller_contiguous = h.cpos[P.cam_block[c]] - h.num_cols_e == P.nf * c;
  if (P.ns > 0) P.caller_contiguous = false;   // (a shared block sits somewhere among the camera-side columns)
  const bool reorder_points = reorder_mode == kReorderAlways || (reorder_mode == kReorderIfContiguous && P.caller_contiguous);
  P.renumbered = reorder_points;

  clk.mark("group by point, distinct cameras");
  // Cameras whose 9-double accumulators do not fit in LDS (decided here: the point order below depends on it).
  // (1 KiB of the 160 stays free for the kernels' static LDS: workgroup reductions, the exchange area of the long points' rounds)
  P.cameras_in_lds = (size_t(P.nf) * P.n_cameras + size_t(P.ns)) * sizeof(double) <= kLdsBytesPerCu - 1024;
  if (!P.cameras_in_lds && P.n_cameras >= kSlotNoCamera) return no("more cameras than the slot word holds");
  int64_t chunk_mib = 0;  // CERES_HIP_Z_CHUNK_MIB=<n>: bound the F^T z ring to n MiB (memory-constrained runs)
  if (const char* e = getenv("CERES_HIP_Z_CHUNK_MIB")) chunk_mib = atoll(e);
  if (!P.cameras_in_lds && P.n_cameras >= (1 << kSlotCamBits)) return no("more cameras than the slot word holds");

  // HYBRID camera accumulation (more cameras than LDS holds; needs the freedom to renumber the points).  The tile pass runs
  // hyb.groups workgroups, each with hyb.rows accumulator rows (9 doubles) in LDS:
  //   rows [0, K_h)   the POPULAR cameras (at least twice the mean number of observations, at most as many as leave room for the
  //                   windows to cover everything) — the same in every workgroup;
  //   rows [K_h, K)   the workgroup's WINDOW: K_w = K - K_h consecutive ids of the other ("cold") cameras, starting stride g cameras
  //                   in: windows overlap wherever groups x K_w exceeds the number of cold cameras, i.e. a camera may have a row in
  //                   several workgroups (cameras of neighbouring ids see the same points in real scenes: a scene whose cameras fit
  //                   a window keeps whole tracks in LDS).
  // Every point is handed to the workgroup whose window holds most of its cold cameras (the least loaded of them; above a load cap:
  // the least loaded workgroup overall — balance before locality), so of a point's k observations the popular ones and at least one
  // cold one are summed in LDS; the others are SPILLED (a 72-byte row F_o^T z_o into a ring, summed by the camera-major second pass,
  // which also collects the rows every workgroup flushes at its end).  A spill-everything pass moves 72 + ~200 bytes per observation
  // on top of the 204 it reads (the second pass fetches whole lines around its 72-byte rows); with three observations per point on
  // 50 000 cameras of random visibility 56 % of the observations stay in LDS, on the replicated libmv graphs (long tracks over
  // consecutive cameras, tests/golden/libmv_problems.npz) all of them.
  HybridRequest hyb = hyb_in;
  if (hyb.rows == 0 && hyb.lds_bytes > 0)
    hyb.rows = int((hyb.lds_bytes - int64_t(512 / kTile) * kTile * P.nf * 8) / (int64_t(P.nf) * 8)) / kTile * kTile;
  if (hyb.rows >= kSlotSpill) hyb.rows = (kSlotSpill - 1) / kTile * kTile;   // (narrow cameras: the slot word has 12 bits for the row)
  bool hybrid = reorder_points && !P.cameras_in_lds && hyb.groups >= 2 && hyb.rows >= 64 && chunk_mib <= 0 && hyb.rows < kSlotSpill;
  if (const char* e = getenv("CERES_HIP_HYBRID")) hybrid = hybrid && atoi(e) != 0;
  std::vector<int32_t> hot_row;                 // camera -> accumulator row < K_h, or -1
  std::vector<int32_t> cold_rank;               // camera -> its rank among the cold cameras (id order), or -1
  std::vector<int32_t> win_start;               // group -> cold rank of its window's first camera (non-decreasing)
  std::vector<std::vector<int32_t>> grp_points; // caller-order points of each group
  int K_h = 0, K_w = 0, n_cold = 0;
  // groups whose window holds the cold camera of rank q: [first, last]
  auto groups_of = [&](int q, int* first, int* last) {
    *last = int(std::upper_bound(win_start.begin(), win_start.end(), q) - win_start.begin()) - 1;
    *first = int(std::upper_bound(win_start.begin(), win_start.end(), q - K_w) - win_start.begin());
  };
  if (hybrid) {
    const int G = hyb.groups, K = hyb.rows;
    std::vector<int64_t> deg(P.n_cameras, 0);
    for (int i = 0; i < n_conf; ++i) if (row_cam[i] >= 0) ++deg[row_cam[i]];
    // most rows the popular cameras may take: the windows must still cover the others, (G - 1) stride + K_w >= n_cold with stride <= K_w
    const int max_hot = K - std::min(K, std::max(1, (P.n_cameras - K + (G - 2)) / (G - 1)));
    std::vector<int32_t> by_deg(P.n_cameras);
    std::iota(by_deg.begin(), by_deg.end(), 0);
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] > deg[b]; });
    const double popular = 2.0 * double(n_conf) / double(P.n_cameras);
    const int hot_cap = std::min(max_hot, P.n_cameras);   // (rows may exceed the cameras: ceres_hip_debug_hybrid_plan takes any `rows`)
    while (K_h < hot_cap && double(deg[by_deg[K_h]]) >= popular) ++K_h;
    if (const char* e = getenv("CERES_HIP_HYB_HOT")) K_h = std::max(0, std::min(atoi(e), hot_cap));   // (experiments)
    K_w = K - K_h;
    hot_row.assign(P.n_cameras, -1);
    cold_rank.assign(P.n_cameras, -1);
    for (int r = 0; r < K_h; ++r) hot_row[by_deg[r]] = r;
    for (int c = 0; c < P.n_cameras; ++c) if (hot_row[c] < 0) cold_rank[c] = n_cold++;
    const int last_start = std::max(0, n_cold - K_w);
    // Window starts.  stride = the smallest step with which G windows reach the last cold camera; where that would put a camera
    // into more than 16 windows (a few thousand cameras: stride of a handful, K_w / stride windows around every camera) the step is
    // raised to K_w / 16 and the G workgroups SHARE the fewer distinct windows, in contiguous blocks: the vote below is per distinct
    // window (at most 17 hold a camera), the point then goes to the lightest workgroup of the window it chose.
    int stride = (last_start + (G - 2)) / (G - 1);
    if (stride == 0 || K_w / stride > 16) stride = std::max(1, K_w / 16);
    const int W = std::min<int64_t>(G, last_start == 0 ? 1 : (last_start + stride - 1) / stride + 1);   // distinct windows
    win_start.resize(G);
    std::vector<int32_t> win_of(G), win_grp(W + 1, G);   // group -> window; window -> its first group
    for (int g = G - 1; g >= 0; --g) {
      win_of[g] = int(int64_t(g) * W / G);
      win_grp[win_of[g]] = g;
      win_start[g] = int(std::min<int64_t>(int64_t(win_of[g]) * stride, last_start));
    }
    // points -> groups
    grp_points.assign(G, {});
    // load in tile slots: a point of more than 64 observations owns whole tiles
    auto slots_of = [&](int p) { return track[p] > kTile ? int64_t((track[p] + kTile - 1) / kTile) * kTile : int64_t(track[p]); };
    std::vector<int64_t> load(G, 0);
    int64_t total_slots = 0;
    for (int p = 0; p < P.n_points; ++p) total_slots += slots_of(p);
    const int64_t cap = (total_slots * 103 / 100 + G - 1) / G + kTile;
    int least = 0, since_scan = 0;
    std::vector<int32_t> votes(W, 0), touched;
    auto lightest_of = [&](int w) {   // the least loaded workgroup of window w
      int b = win_grp[w];
      for (int g = b + 1; g < win_grp[w + 1]; ++g) if (load[g] < load[b]) b = g;
      return b;
    };
    for (int p = 0; p < P.n_points; ++p) {
      if ((since_scan++ & 1023) == 0) least = int(std::min_element(load.begin(), load.end()) - load.begin());
      touched.clear();
      for (int q = row_start[p]; q < row_start[p + 1]; ++q) {
        if (row_cam[order[q]] < 0) continue;   // a row without a camera cell votes for nobody
        const int r = cold_rank[row_cam[order[q]]];
        if (r < 0) continue;
        int g0, g1;
        groups_of(r, &g0, &g1);
        if (g0 > g1) continue;
        for (int w = win_of[g0]; w <= win_of[g1]; ++w) if (votes[w]++ == 0) touched.push_back(w);
      }
      int best = -1, best_votes = 0;
      for (int w : touched) {   // most of the point's cameras, then the lighter load
        const int g = lightest_of(w);
        if (load[g] < cap && (best < 0 || votes[w] > best_votes || (votes[w] == best_votes && load[g] < load[best]))) { best = g; best_votes = votes[w]; }
      }
      for (int w : touched) votes[w] = 0;
      if (best < 0) {
        if (load[least] >= cap) least = int(std::min_element(load.begin(), load.end()) - load.begin());
        best = least;
      }
      grp_points[best].push_back(p);
      load[best] += slots_of(p);
    }
  }

  clk.mark("hybrid assignment");
  // INTERNAL point order.  A tile holds whole points, so the greedy packing in the caller's order wastes the slots behind the last
  // point that fits: about half a track per tile, 5 % of the slots on the Venice shape — 5 % of the bytes of EVERY pass over the
  // tiles.  With reorder_points (no CG vector walks the caller's point order in tile order: the Schur solvers, and CGNR on internal
  // vectors; the kernels reach point-indexed data of the caller — D, the step — through pt_pos) the points are renumbered: when the
  // next point does not fit, the largest point among the next kReorderWindow ones that does fit is pulled forward (best fit in a
  // window: the gathers from the caller's layout stay local).  seq[i] = caller-order point of internal point i.  Hybrid: group by
  // group, every group starting a tile of its own (grp_pt_ptr = first internal point of each group).
  // Points of more than one tile ("long": they own whole tiles) come BEHIND the others — of their group, or of everything: the
  // streaming kernels pipeline over the normal tiles and then take the long points in cooperative ROUNDS (below).  Not with a
  // chunked ring (a chunk is a tile range of its own launch).
  const bool segregate = chunk_mib <= 0;
  constexpr int kReorderWindow = 96;
  std::vector<int32_t> seq;
  seq.reserve(P.n_points);
  std::vector<int32_t> grp_pt_ptr;
  auto pack_order = [&](const int32_t* pts, int n) {   // appends pts[0 .. n) in packing order; starts on a fresh tile
    std::vector<uint8_t> taken(n, 0);
    std::vector<int32_t> longs;   // points of more than a tile: behind the others (they own their tiles; see the rounds below)
    int next = 0, placed = 0, used = kTile, npts = 0;
    auto put = [&](int i) { taken[i] = 1; seq.push_back(pts[i]); ++placed; };
    while (placed < n) {
      while (taken[next]) ++next;
      const int k = track[pts[next]];
      if (k > kTile) {
        if (segregate) { taken[next] = 1; ++placed; longs.push_back(pts[next]); }
        else { put(next); used = kTile; npts = 0; }   // a long point owns its tiles
        continue;
      }
      if (used + k <= kTile && npts < kMaxPointsPerTile) { put(next); used += k; ++npts; continue; }
      int best = -1;
      const int room = kTile - used;
      if (room > 0 && npts < kMaxPointsPerTile) {
        int seen = 0;
        for (int q = next + 1; q < n && seen < kReorderWindow; ++q) {
          if (taken[q]) continue;
          ++seen;
          const int kq = track[pts[q]];
          if (kq <= room && (best < 0 || kq > track[pts[best]])) { best = q; if (kq == room) break; }
        }
      }
      if (best >= 0) { put(best); used += track[pts[best]]; ++npts; continue; }
      put(next); used = k; npts = 1;   // new tile
    }
    seq.insert(seq.end(), longs.begin(), longs.end());
  };
  if (!reorder_points) {
    seq.resize(P.n_points);
    std::iota(seq.begin(), seq.end(), 0);
  } else if (hybrid) {
    for (const auto& g : grp_points) { grp_pt_ptr.push_back(int32_t(seq.size())); pack_order(g.data(), int(g.size())); }
    grp_pt_ptr.push_back(int32_t(seq.size()));
  } else {
    std::vector<int32_t> all(P.n_points);
    std::iota(all.begin(), all.end(), 0);
    pack_order(all.data(), P.n_points);
  }
  if (reorder_points) {
    std::vector<int32_t> blk(P.n_points), trk(P.n_points);
    for (int i = 0; i < P.n_points; ++i) { blk[i] = P.pt_block[seq[i]]; trk[i] = track[seq[i]]; }
    P.pt_block.swap(blk);
    track.swap(trk);
  }

  clk.mark("point order");
  // Vector offsets.
  P.pt_pos.resize(P.n_points);
  P.cam_pos.resize(P.n_cameras);
  P.points_contiguous = P.cameras_contiguous = true;
  for (int p = 0; p < P.n_points; ++p) {
    P.pt_pos[p] = h.cpos[P.pt_block[p]];
    if (P.pt_pos[p] != P.ne * p) P.points_contiguous = false;
  }
  for (int c = 0; c < P.n_cameras; ++c) {
    P.cam_pos[c] = h.cpos[P.cam_block[c]] - h.num_cols_e;
    if (c == 0) P.cam_base = P.cam_pos[0];
    if (P.cam_pos[c] != P.cam_base + P.nf * c) P.cameras_contiguous = false;
  }
  P.contiguous_layout = P.points_contiguous && P.cameras_contiguous;

  // Tiles.  Greedy in point order; a point longer than one tile gets tiles of its own.
  auto new_tile = [&](int kind, int aux) {
    P.tile_kind.push_back(kind);
    P.tile_aux.push_back(aux);
    P.slot_epos.resize(P.slot_epos.size() + kTile, -1);
    P.slot_fpos.resize(P.slot_fpos.size() + kTile, -1);
    P.slot_bpos.resize(P.slot_bpos.size() + kTile, -1);
    P.slot_cam.resize(P.slot_cam.size() + kTile, -1);
    P.slot_pt.resize(P.slot_pt.size() + kTile, -1);
    P.slot_row.resize(P.slot_row.size() + kTile, -1);
    P.slot_seg.resize(P.slot_seg.size() + kTile, 0u);
    if (P.ns > 0)
      for (int q = 0; q < kMaxSharedCellsPerRow; ++q) { P.slot_hpos[q].resize(P.slot_hpos[q].size() + kTile, -1); P.slot_hdesc[q].resize(P.slot_hdesc[q].size() + kTile, 0); }
    return int64_t(P.tile_kind.size()) - 1;
  };
  int64_t tile = -1;
  int used = kTile;  // slots used in the current tile (kTile forces a new one)
  int npts_in_tile = 0;
  auto put_shared = [&](int64_t s, int i) {
    if (P.ns == 0) return;
    for (int q = 0; q < kMaxSharedCellsPerRow; ++q) { P.slot_hpos[q][s] = row_hpos[q][i]; P.slot_hdesc[q][s] = row_hdesc[q][i]; }
  };
  auto emit_point = [&](int p) {
    const int k = track[p];
    int idx = row_start[seq[p]];   // the point's rows, in the caller's order
    if (k > kTile) {
      const int nt = (k + kTile - 1) / kTile;
      for (int t = 0; t < nt; ++t) {
        tile = new_tile(t == 0 ? 1 : 2, t == 0 ? nt : 0);
        const int cnt = std::min(kTile, k - t * kTile);
        for (int l = 0; l < cnt; ++l) {
          const int i = order[idx++];
          const int64_t s = tile * kTile + l;
          P.slot_epos[s] = row_epos[i]; P.slot_fpos[s] = row_fpos[i]; P.slot_bpos[s] = h.rpos[orig[i]];
          P.slot_cam[s] = row_cam[i] >= 0 ? row_cam[i] : -2; P.slot_pt[s] = p; P.slot_row[s] = i;   // -2: a valid slot without a camera cell
          put_shared(s, i);
          P.slot_seg[s] = 0u | (uint32_t(cnt - 1) << 8) | (1u << 16);
        }
      }
      used = kTile;  // nothing shares a tile with a long point
      return;
    }
    // A normal tile holds at most kMaxPointsPerTile points: the cooperative point-space path of the
    // fused JtJx (kernels_bal.hip, compute_stream / issue_aux) gives lane L the scalars L and 64 + L of
    // the tile's point range, i.e. 128 scalars = 42 whole points.  Only tiles full of 1- and
    // 2-observation points ever reach the cap.
    if (used + k > kTile || npts_in_tile == kMaxPointsPerTile) { tile = new_tile(0, 0); used = 0; npts_in_tile = 0; }
    ++npts_in_tile;
    for (int l = 0; l < k; ++l) {
      const int i = order[idx++];
      const int64_t s = tile * kTile + used + l;
      P.slot_epos[s] = row_epos[i]; P.slot_fpos[s] = row_fpos[i]; P.slot_bpos[s] = h.rpos[orig[i]];
      P.slot_cam[s] = row_cam[i] >= 0 ? row_cam[i] : -2; P.slot_pt[s] = p; P.slot_row[s] = i;
      put_shared(s, i);
      P.slot_seg[s] = uint32_t(used) | (uint32_t(used + k - 1) << 8) | (1u << 16);
    }
    used += k;
  };
  // one range of points = one hybrid group (whose tiles are its own), or everything.  segregate: the normal tiles first (a tile
  // still ends where a long point sat: the points of a tile are consecutive ids), then the long points' (long_ptr = where they begin)
  std::vector<int32_t> long_ptr;
  auto emit_range = [&](int p0, int p1) {
    used = kTile;
    if (!segregate) {
      for (int p = p0; p < p1; ++p) emit_point(p);
      long_ptr.push_back(int32_t(P.tile_kind.size()));
      return;
    }
    for (int p = p0; p < p1; ++p) {
      if (track[p] > kTile) used = kTile;
      else emit_point(p);
    }
    long_ptr.push_back(int32_t(P.tile_kind.size()));
    for (int p = p0; p < p1; ++p) if (track[p] > kTile) emit_point(p);
    used = kTile;
  };
  if (grp_pt_ptr.empty()) {
    emit_range(0, P.n_points);
  } else {
    for (size_t g = 0; g + 1 < grp_pt_ptr.size(); ++g) {   // hybrid: a group's tiles are its own (empty groups own none)
      P.grp_tile_ptr.push_back(int32_t(P.tile_kind.size()));
      emit_range(grp_pt_ptr[g], grp_pt_ptr[g + 1]);
    }
    P.grp_tile_ptr.push_back(int32_t(P.tile_kind.size()));
  }
  P.n_tiles = int64_t(P.tile_kind.size());
  P.tile_pt0.resize(P.n_tiles);
  for (int64_t t = 0; t < P.n_tiles; ++t) P.tile_pt0[t] = P.slot_pt[t * kTile];
  // Normal tiles: tile_aux = longest track in the tile (bounds the segmented-scan steps).
  for (int64_t t = 0; t < P.n_tiles; ++t) {
    if (P.tile_kind[t] != 0) continue;
    int longest = 1;
    for (int l = 0; l < kTile; ++l) {
      const uint32_t sg = P.slot_seg[t * kTile + l];
      if (sg & (1u << 16)) longest = std::max(longest, int((sg >> 8) & 0xff) - int(sg & 0xff) + 1);
    }
    // Points of a normal tile are consecutive ids, so their 3-vectors are one contiguous range
    // [3 p0, 3 p0 + 3 npts) of every point-space vector.  Lane L of the tile owns scalars L and
    // 64 + L of that range; it needs to know which lane holds the finished sum of "its" point:
    // tailA = last lane of point L / 3, tailB = last lane of point (64 + L) / 3.
    int tails[kTile], npts = 0;
    for (int l = 0; l < kTile; ++l) {
      const uint32_t sg = P.slot_seg[t * kTile + l];
      if ((sg & (1u << 16)) && int((sg >> 8) & 0xff) == l) tails[npts++] = l;
    }
    for (int l = 0; l < kTile; ++l) {
      uint32_t& sg = P.slot_seg[t * kTile + l];
      const int ia = l / 3, ib = (kTile + l) / 3;
      if (ia < npts) sg |= (uint32_t(tails[ia]) << 17) | (1u << 23);
      if (ib < npts) sg |= (uint32_t(tails[ib]) << 24) | (1u << 30);
    }
    P.tile_aux[t] = longest | (npts << 8);
  }
  // Padding slots form singleton segments so that shuffles stay in range.
  for (int64_t s = 0; s < P.n_tiles * kTile; ++s)
    if (!(P.slot_seg[s] & (1u << 16))) { const uint32_t l = uint32_t(s % kTile); P.slot_seg[s] |= l | (l << 8); }

  clk.mark("tiles");
  // ROUNDS of long points.  A streaming kernel's workgroup has kRoundWaves waves; in a round each of them takes ONE tile, the waves
  // of a point exchange their tile sums through LDS, and every wave finishes its tile from registers: a long point is read once,
  // with a tile per wave in flight, instead of one wave walking all its tiles twice.  A round holds whole points (tightest fit,
  // longest first); word = tile | first wave of the point << 26 | (waves of the point - 1) << 29, kRoundIdle for a wave without a
  // tile.  A point of MORE than kRoundWaves tiles has rounds of its own, twice: kRoundSum rounds (every wave keeps the running sum of
  // the rounds' totals) and then kRoundApply rounds over the same tiles (L2-warm), which finish them with that sum; kRoundLast marks
  // the last round of either phase.  The workgroup-level unit of work is the SEQUENCE (one packed round, or all rounds of one such
  // point): a workgroup takes whole sequences, those of range g (the hybrid group, or everything) are [round_ptr[g], round_ptr[g + 1]),
  // the longest first.  Every head tile in a round has kind 3 (kind 1: no rounds — a chunked ring, or more tiles than a word holds).
  P.long_ptr = long_ptr;
  P.long_behind = segregate;
  P.round_ptr.assign(1, 0);
  P.seq_ptr.assign(1, 0);
  if (segregate && P.n_tiles < (int64_t(1) << 26)) {
    const size_t n_ranges = long_ptr.size();
    auto new_round = [&](int flag) {
      P.round_word.resize(P.round_word.size() + kRoundWaves, kRoundIdle);
      P.round_flag.push_back(flag);
      return P.round_word.size() / kRoundWaves - 1;
    };
    for (size_t g = 0; g < n_ranges; ++g) {
      const int64_t t0 = long_ptr[g], t1 = P.grp_tile_ptr.empty() ? P.n_tiles : int64_t(P.grp_tile_ptr[g + 1]);
      std::vector<std::pair<int, int64_t>> heads;   // (tiles, head tile)
      for (int64_t t = t0; t < t1; ++t)
        if (P.tile_kind[t] == 1) { heads.emplace_back(P.tile_aux[t], t); P.tile_kind[t] = 3; }
      std::stable_sort(heads.begin(), heads.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
      size_t h = 0;
      for (; h < heads.size() && heads[h].first > kRoundWaves; ++h) {   // rounds of its own, two phases
        const int nt = heads[h].first;
        for (int phase = 0; phase < 2; ++phase) {
          for (int t = 0; t < nt; t += kRoundWaves) {
            const int n = std::min(kRoundWaves, nt - t);
            const size_t r = new_round((phase == 0 ? kRoundSum : kRoundApply) | (t + n == nt ? kRoundLast : 0));
            for (int w = 0; w < n; ++w) P.round_word[r * kRoundWaves + w] = uint32_t(heads[h].second + t + w) | (uint32_t(n - 1) << 29);
          }
        }
        P.seq_ptr.push_back(int32_t(P.round_flag.size()));
      }
      const size_t round0 = P.round_flag.size();
      std::vector<int> fill;                               // waves taken in each packed round of this range
      std::vector<size_t> open_by_room[kRoundWaves + 1];   // rounds with exactly `room` free waves (stacks)
      for (; h < heads.size(); ++h) {
        const int nt = heads[h].first;
        size_t r = size_t(-1);
        for (int room = nt; room <= kRoundWaves && r == size_t(-1); ++room)   // tightest fit
          if (!open_by_room[room].empty()) { r = open_by_room[room].back(); open_by_room[room].pop_back(); }
        if (r == size_t(-1)) { r = fill.size(); fill.push_back(0); new_round(0); }
        const int w0 = fill[r];
        for (int t = 0; t < nt; ++t)
          P.round_word[(round0 + r) * kRoundWaves + w0 + t] = uint32_t(heads[h].second + t) | (uint32_t(w0) << 26) | (uint32_t(nt - 1) << 29);
        fill[r] += nt;
        if (fill[r] < kRoundWaves) open_by_room[kRoundWaves - fill[r]].push_back(r);
      }
      for (size_t r = 0; r < fill.size(); ++r) P.seq_ptr.push_back(int32_t(round0 + r + 1));
      P.round_ptr.push_back(int32_t(P.seq_ptr.size() - 1));
    }
  } else {
    P.round_ptr.resize(long_ptr.size() + 1, 0);
  }

  clk.mark("rounds");
  // Camera-major lists (counting sort over slots keeps point order inside a camera).
  P.cam_ptr.assign(P.n_cameras + 1, 0);
  for (int64_t s = 0; s < P.n_tiles * kTile; ++s) if (P.slot_cam[s] >= 0) ++P.cam_ptr[P.slot_cam[s] + 1];
  for (int c = 0; c < P.n_cameras; ++c) {
    P.max_camera_degree = std::max(P.max_camera_degree, P.cam_ptr[c + 1]);
    P.cam_ptr[c + 1] += P.cam_ptr[c];
  }
  P.cam_fpos.resize(P.n_obs);
  P.cam_slot.resize(P.n_obs);
  if (!hybrid) {
    // CERES_HIP_MO_CAMERA_MAJOR=1 (experiment, round 4): the M_o records in CAMERA-major order — kInit scatters its 32-byte records,
    // the camera-major pass reads them back to back instead of a 128-byte line per record.  Measured on the Venice shape
    // (profiles/r04zi_mo_camera_major_ab.txt): the pass 0.332 -> 0.278 ms, kInit 0.471 -> 0.526 ms — the scattered stores cost what
    // the coalesced reads save; off.
    bool mo_cm = false;
    if (const char* e = getenv("CERES_HIP_MO_CAMERA_MAJOR")) mo_cm = atoi(e) != 0;
    if (mo_cm) P.mo_index.assign(size_t(P.n_tiles) * kTile, 0);
    int64_t extra = P.n_obs;   // records of valid slots without a camera cell: behind the lists (nobody reads them)
    std::vector<int32_t> cur(P.cam_ptr.begin(), P.cam_ptr.end() - 1);
    for (int64_t s = 0; s < P.n_tiles * kTile; ++s) {
      if (P.slot_cam[s] < 0) {
        if (mo_cm && P.slot_cam[s] == -2) P.mo_index[s] = int32_t(std::min<int64_t>(extra++, P.n_tiles * kTile - 1));
        continue;
      }
      const int q = cur[P.slot_cam[s]]++;
      P.cam_fpos[q] = P.slot_fpos[s];
      P.cam_slot[q] = mo_cm ? q : int32_t(s);
      if (mo_cm) P.mo_index[s] = q;
    }
  } else {
    // Hybrid plans scatter a camera's observations over the workgroups' tile ranges: a list in slot order would make the camera-major
    // pass jump back and forth through the caller's value array and through M_o (synthetic10M: 2.13 -> 2.66 ms).  The lists go in the
    // CALLER'S row order instead, and M_o is indexed by row (mo_index): both gathers sweep forward again.
    std::vector<int32_t> row_slot(n_conf, -1);
    P.mo_index.assign(size_t(P.n_tiles) * kTile, 0);
    for (int64_t s = 0; s < P.n_tiles * kTile; ++s) {
      if (P.slot_cam[s] == -1) { P.mo_index[s] = int32_t(s % n_conf); continue; }   // padding: any record (never written: the slot is not valid)
      P.mo_index[s] = P.slot_row[s];   // (also a valid slot without a camera cell: its own row's record, which no camera list refers to)
      row_slot[P.slot_row[s]] = int32_t(s);
    }
    std::vector<int32_t> cur(P.cam_ptr.begin(), P.cam_ptr.end() - 1);
    for (int i = 0; i < n_conf; ++i) {
      const int64_t s = row_slot[i];
      if (P.slot_cam[s] < 0) continue;   // a row without a camera cell is in no camera's list
      const int q = cur[P.slot_cam[s]]++;
      P.cam_fpos[q] = P.slot_fpos[s];
      P.cam_slot[q] = i;
    }
  }
  // Item size: kCamChunk at scale; smaller problems get shorter items so that they too spread over the chip, down to one tile's
  // worth (short items go to the matrix-pipe kernel, whose sums need no wavefront reduction: LaunchBalCameraItems; with the
  // lane-per-observation kernel alone, items under 256 observations spent more time reducing than accumulating).
  int64_t chunk_min = kTile;
  if (const char* e = getenv("CERES_HIP_CAM_CHUNK_MIN")) chunk_min = std::max<int64_t>(kTile, atoll(e));   // (experiments)
  const int chunk = int(std::max<int64_t>(chunk_min, std::min<int64_t>(kCamChunk, ((P.n_obs / 8192 + kTile - 1) / kTile) * kTile)));
  P.cam_item_ptr.assign(P.n_cameras + 1, 0);
  for (int c = 0; c < P.n_cameras; ++c) {
    int b = P.cam_ptr[c];
    do {  // a camera without observations still gets one (empty) item so that D^2 is added
      const int e = std::min(P.cam_ptr[c + 1], b + chunk);
      P.item_cam.push_back(c); P.item_begin.push_back(b); P.item_end.push_back(e);
      b = e;
    } while (b < P.cam_ptr[c + 1]);
    P.cam_item_ptr[c + 1] = int32_t(P.item_cam.size());  // the items of a camera are consecutive
  }
  if (P.n_tiles * kTile >= (int64_t(1) << 31)) return no("more than 2^31 slots");

  clk.mark("camera-major lists, items");
  // The word the kernels read per slot: camera id | accumulator row << kSlotCamBits (kSlotSpill: no LDS row, the slot's F^T z is
  // spilled).  With the accumulators of ALL cameras in LDS the accumulator row is the camera id itself; the word's row field then says
  // whether the camera's part of x is STAGED in LDS (row + 1 of xhot_cam, 0: not) — set whenever the plan ranks cameras, also for a solver
  // that ends up launching without the staged x (few tiles per workgroup, a kernel whose static LDS leaves no room): every LDS-mode
  // consumer masks the field (finish_slot, kernels_bal.inc); nothing may read the word raw as a camera id.
  P.slot_word.assign(P.slot_cam.begin(), P.slot_cam.end());
  for (auto& w : P.slot_word) if (w == -2) w = 0;   // no camera cell: the tiles hold zeros for F, any camera's row takes the zero sums
  P.xhot_cam.clear();
  if (P.cameras_in_lds) {
    // x_f of the most observed cameras in what LDS the accumulators leave (1 KiB stays for the kernels' static arrays, as above)
    static const bool enabled = [] { const char* e = getenv("CERES_HIP_XHOT"); return !e || atoi(e) != 0; }();
    const size_t used = (size_t(P.nf) * P.n_cameras + size_t(P.ns)) * sizeof(double) + 1024;
    int64_t n_hot = enabled && kLdsBytesPerCu > used ? int64_t((kLdsBytesPerCu - used) / (sizeof(double) * P.nf)) : 0;
    n_hot = std::min<int64_t>({n_hot, int64_t(P.n_cameras), int64_t(kSlotSpill) - 2});   // (the word's row field: 12 bits, all ones = kSlotSpill)
    if (n_hot > 0) {
      std::vector<int64_t> deg(P.n_cameras, 0);
      for (int64_t sl = 0; sl < P.n_tiles * kTile; ++sl) if (P.slot_cam[sl] >= 0) ++deg[P.slot_cam[sl]];
      std::vector<int32_t> by_deg(P.n_cameras);
      std::iota(by_deg.begin(), by_deg.end(), 0);
      std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] > deg[b]; });
      P.xhot_cam.assign(by_deg.begin(), by_deg.begin() + n_hot);
      std::vector<int32_t> row_of(P.n_cameras, -1);
      for (int r = 0; r < int(n_hot); ++r) row_of[P.xhot_cam[r]] = r;
      for (int64_t sl = 0; sl < P.n_tiles * kTile; ++sl) {
        const int c = P.slot_cam[sl];
        if (c >= 0 && row_of[c] >= 0) P.slot_word[sl] = int32_t(uint32_t(c) | (uint32_t(row_of[c] + 1) << kSlotCamBits));
      }
    }
  }
  if (!P.cameras_in_lds) {
    // Camera-major second pass.  The tile pass leaves the spilled rows of a tile back to back in a ring (tile_zbase = the tile's
    // first row; a slot's row = tile_zbase + its rank among the tile's spilled slots), the hybrid workgroups append their accumulator
    // rows behind them (z_flush_row0 + group * hyb_rows + row), and every camera has the list of ring rows that are its own, cut into
    // units of <= kZUnit entries.  Default: ONE chunk (the whole problem).  Chunks sized for the Infinity Cache (64-128 MiB of ring)
    // were measured and do not pay on this kernel pair (profiles/r02g_chunk_sweep_synthetic1M.txt), although a streaming write ->
    // read-back hand-off of that size does stay in the cache (profiles/r02_infinity_cache_handoff_probe.txt); CERES_HIP_Z_CHUNK_MIB
    // bounds the ring (no hybrid then: a chunk is a launch of its own; the chunked path is covered by tests).
    const int G = hybrid ? hyb.groups : 0, K = hybrid ? hyb.rows : 0;
    P.hybrid = hybrid;
    P.hyb_groups = G; P.hyb_rows = K; P.hyb_hot = K_h;
    std::vector<int32_t> tile_group(P.n_tiles, 0);
    if (hybrid)
      for (int g = 0; g < G; ++g)
        for (int t = P.grp_tile_ptr[g]; t < P.grp_tile_ptr[g + 1]; ++t) tile_group[t] = g;
    int64_t n_local = 0;
    for (int64_t s = 0; s < P.n_tiles * kTile; ++s) {
      const int c = P.slot_cam[s];
      if (c == -2) P.slot_word[s] = int32_t(uint32_t(kSlotNoCamera) | (uint32_t(kSlotSpill) << kSlotCamBits));   // valid, no camera cell: neither summed nor spilled
      if (c < 0) continue;
      int row = kSlotSpill;
      if (hybrid) {
        if (hot_row[c] >= 0) row = hot_row[c];
        else {
          const int st = win_start[tile_group[s / kTile]], r = cold_rank[c];
          if (r >= st && r < st + K_w) row = K_h + (r - st);
        }
        if (row != kSlotSpill) ++n_local;
      }
      P.slot_word[s] = int32_t(uint32_t(c) | (uint32_t(row) << kSlotCamBits));
    }
    P.n_local_obs = n_local;
    const int64_t want_tiles = chunk_mib > 0 ? std::max<int64_t>(1, chunk_mib * (int64_t(1) << 20) / (int64_t(kTile) * 8 * P.nf)) : P.n_tiles;
    P.zc_tile_ptr.assign(1, 0);
    for (int64_t t = 0; t < P.n_tiles;) {
      int64_t e = std::min<int64_t>(P.n_tiles, t + want_tiles);
      while (e < P.n_tiles && P.tile_kind[e] == 2) ++e;  // keep a long point's tiles together
      P.zc_tile_ptr.push_back(int32_t(e));
      t = e;
    }
    const int n_chunks = int(P.zc_tile_ptr.size()) - 1;
    P.tile_zbase.assign(P.n_tiles, 0);
    std::vector<int32_t> slot_zrow(size_t(P.n_tiles) * kTile, -1);
    P.zc_unit_ptr.assign(1, 0);
    std::vector<int32_t> count(P.n_cameras + 1), cur(P.n_cameras);
    int64_t base = 0;  // entries emitted so far
    for (int k = 0; k < n_chunks; ++k) {
      const int64_t t0 = P.zc_tile_ptr[k], t1 = P.zc_tile_ptr[k + 1];
      int64_t row = 0;
      std::fill(count.begin(), count.end(), 0);
      for (int64_t t = t0; t < t1; ++t) {
        P.tile_zbase[t] = int32_t(row);
        for (int l = 0; l < kTile; ++l) {
          const int64_t s = t * kTile + l;
          if (P.slot_cam[s] < 0 || (uint32_t(P.slot_word[s]) >> kSlotCamBits) != uint32_t(kSlotSpill)) continue;
          slot_zrow[s] = int32_t(row++);
          ++count[P.slot_cam[s] + 1];
        }
      }
      const int64_t flush_row0 = row;
      if (hybrid) {
        P.z_flush_row0 = flush_row0;
        row += int64_t(G) * K;
        for (int c = 0; c < P.n_cameras; ++c) {   // popular: a row in every workgroup; cold: in every workgroup whose window holds it
          if (hot_row[c] >= 0) { count[c + 1] += G; continue; }
          int g0, g1;
          groups_of(cold_rank[c], &g0, &g1);
          count[c + 1] += std::max(0, g1 - g0 + 1);
        }
      }
      if (row >= (int64_t(1) << 31)) return no("F^T z ring beyond 2^31 rows");
      P.z_ring_rows = std::max<int64_t>(P.z_ring_rows, row);
      if (base + count[P.n_cameras] >= (int64_t(1) << 31)) return no("more than 2^31 ring entries");
      std::vector<int32_t> start(P.n_cameras + 1, 0);
      for (int c = 0; c < P.n_cameras; ++c) {
        const int n = count[c + 1];
        start[c + 1] = start[c] + n;
        cur[c] = start[c];
        for (int b = 0; b < n; b += kZUnit) {
          P.zu_cam.push_back(c);
          P.zu_begin.push_back(int32_t(base + start[c] + b));
          P.zu_end.push_back(int32_t(base + start[c] + std::min(n, b + kZUnit)));
          P.zu_shared.push_back(n > kZUnit ? 1 : 0);
        }
      }
      P.zc_slot.resize(size_t(base + start[P.n_cameras]));
      for (int64_t s = t0 * kTile; s < t1 * kTile; ++s)
        if (slot_zrow[s] >= 0) P.zc_slot[base + cur[P.slot_cam[s]]++] = slot_zrow[s];
      if (hybrid)
        for (int c = 0; c < P.n_cameras; ++c) {
          if (hot_row[c] >= 0) {
            for (int g = 0; g < G; ++g) P.zc_slot[base + cur[c]++] = int32_t(flush_row0 + int64_t(g) * K + hot_row[c]);
            continue;
          }
          int g0, g1;
          groups_of(cold_rank[c], &g0, &g1);
          for (int g = g0; g <= g1; ++g) P.zc_slot[base + cur[c]++] = int32_t(flush_row0 + int64_t(g) * K + K_h + (cold_rank[c] - win_start[g]));
        }
      base += start[P.n_cameras];
      P.zc_unit_ptr.push_back(int32_t(P.zu_cam.size()));
    }
  }
  clk.mark("slot words, ring");
  P.eligible = true;
}

void BuildSchurStorage(const HostStructure& h, SchurStorage* out) {
  SchurStorage& S = *out;
  S = SchurStorage();
  S.nf = h.ncb - h.nelim;
  S.cell_row.resize(h.ncells);
  for (int i = 0; i < h.nrb; ++i)
    for (int k = h.rptr[i]; k < h.rptr[i + 1]; ++k) S.cell_row[k] = i;
  struct Trip { int64_t key; int32_t e, k1, k2; };
  std::vector<Trip> trips;
  const int64_t nf = S.nf;
  auto add = [&](int e, int k1, int k2) {
    const int b1 = h.ccol[k1] - h.nelim, b2 = h.ccol[k2] - h.nelim;
    if (b1 > b2) return;  // upper block triangle only (I/schur_eliminator_impl.h:548-565); equal blocks keep both orders
    trips.push_back({int64_t(b1) * nf + b2, e, k1, k2});
  };
  std::vector<int32_t> cells;
  // chunks: all F cells of the rows of one E block, every ordered pair of them
  for (int e = 0; e < h.nelim; ++e) {
    cells.clear();
    for (int i = h.chunk_start[e]; i < h.chunk_start[e] + h.chunk_size[e]; ++i)
      for (int k = h.rptr[i] + 1; k < h.rptr[i + 1]; ++k) cells.push_back(k);
    for (int k1 : cells) for (int k2 : cells) add(e, k1, k2);
  }
  // E-free rows go into the Schur complement as plain outer products (:274-290)
  for (int i = 0; i < h.nrb; ++i) {
    if (h.row_e_block[i] >= 0) continue;
    for (int k1 = h.rptr[i]; k1 < h.rptr[i + 1]; ++k1)
      for (int k2 = h.rptr[i]; k2 < h.rptr[i + 1]; ++k2) add(-1, k1, k2);
  }
  std::stable_sort(trips.begin(), trips.end(), [](const Trip& a, const Trip& b) { return a.key < b.key; });
  // pairs = diagonal pairs (always present, :231-234) merged with the keys that occur
  size_t t = 0;
  S.row_ptr.assign(S.nf + 1, 0);
  S.pair_off.push_back(0);
  S.trip_ptr.push_back(0);
  for (int i = 0; i < S.nf; ++i) {
    S.row_ptr[i] = int32_t(S.pair_i.size());
    bool diag_done = false;
    auto emit = [&](int j) {
      S.pair_i.push_back(i);
      S.pair_j.push_back(j);
      S.pair_off.push_back(S.pair_off.back() + int64_t(h.csz[h.nelim + i]) * h.csz[h.nelim + j]);
      while (t < trips.size() && trips[t].key == int64_t(i) * nf + j) {
        S.trip_e.push_back(trips[t].e); S.trip_k1.push_back(trips[t].k1); S.trip_k2.push_back(trips[t].k2);
        ++t;
      }
      S.trip_ptr.push_back(int64_t(S.trip_e.size()));
    };
    while (t < trips.size() && trips[t].key / nf == i) {
      const int j = int(trips[t].key % nf);
      if (!diag_done && j > i) { emit(i); diag_done = true; continue; }
      if (j == i) diag_done = true;
      emit(j);
    }
    if (!diag_done) emit(i);
  }
  S.row_ptr[S.nf] = int32_t(S.pair_i.size());
  // work items: <= kSchurItem triples of one pair (a pair without triples still gets one: its block is written, D^2 joins the diagonal)
  S.pair_item_ptr.assign(1, 0);
  S.item_off.assign(1, 0);
  for (size_t p = 0; p < S.pair_i.size(); ++p) {
    const int64_t nent = S.pair_off[p + 1] - S.pair_off[p];
    int64_t t0 = S.trip_ptr[p];
    do {
      const int64_t t1 = std::min<int64_t>(S.trip_ptr[p + 1], t0 + kSchurItem);
      S.item_pair.push_back(int32_t(p)); S.item_t0.push_back(t0); S.item_t1.push_back(t1);
      S.item_off.push_back(S.item_off.back() + nent);
      t0 = t1;
    } while (t0 < S.trip_ptr[p + 1]);
    S.pair_item_ptr.push_back(int32_t(S.item_pair.size()));
  }
  // transpose half: for block i the pairs (j, i), j < i
  S.col_ptr.assign(S.nf + 1, 0);
  const int np = int(S.pair_i.size());
  for (int p = 0; p < np; ++p) if (S.pair_i[p] != S.pair_j[p]) ++S.col_ptr[S.pair_j[p] + 1];
  for (int i = 0; i < S.nf; ++i) S.col_ptr[i + 1] += S.col_ptr[i];
  S.col_pair.resize(S.col_ptr[S.nf]);
  std::vector<int32_t> cur(S.col_ptr.begin(), S.col_ptr.end() - 1);
  for (int p = 0; p < np; ++p) if (S.pair_i[p] != S.pair_j[p]) S.col_pair[cur[S.pair_j[p]]++] = p;
}

}  // namespace chip
