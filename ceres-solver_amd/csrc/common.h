// common.h — shared declarations of the gfx950 LM linear-solve library.
//
// Layering inside csrc/:
//   plan.cc            host analysis of the block structure (no HIP): flattening,
//                      transpose, chunks, BAL detection, tile packing plan
//   kernels_generic.hip any block sizes; multi-pass kernels that mirror the
//                      reference's operator decomposition
//   kernels_bal.hip    static <2,3,9>; fused single-pass kernels over packed tiles
//   kernels_cg.hip     device-resident preconditioned CG (vector kernels)
//   solver.hip         the C ABI of include/ceres_hip.h
//   comm.hip           RCCL communicator (multi-GPU sharding by point)
#ifndef CERES_HIP_COMMON_H_
#define CERES_HIP_COMMON_H_

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ceres_hip.h"

namespace chip {

constexpr int kTile = 64;           // observation slots per tile = one wavefront
constexpr int kMaxGenericBlock = 16;  // largest block dimension the generic kernels take
constexpr int kCamChunk = 512;      // max observations per work item (one wavefront) of the camera-major kernels
constexpr int kPairsPerSlot = 12;   // 24 Jacobian doubles per observation as 12 double2
// double2 elements from one tile to the next.  A/B builds only (tools/build_variant.sh): 13 rows leave a 1 KiB gap behind each
// tile (an experiment on where JtJx's point-space output should live); the product packs the tiles back to back.
#ifndef CERES_HIP_AB_TILE_ROWS
#define CERES_HIP_AB_TILE_ROWS 12
#endif
constexpr int kTilePitch = CERES_HIP_AB_TILE_ROWS * 64;
constexpr size_t kLdsBytesPerCu = 160 * 1024;  // LDS per CU on gfx950
constexpr int kSchurItem = 128;        // max triples of one work item of the explicit Schur elimination
constexpr int kZUnit = 64;             // max entries of one work unit of the chunked camera-major pass (cameras not in LDS)
constexpr int kMaxPointsPerTile = 42;  // 3 * 42 = 126 <= 128 point-space scalars per tile (two per lane)
// The per-slot index word of the tiles: camera id in the low kSlotCamBits bits, LDS accumulator row above them (kSlotSpill = none)
constexpr int kSlotCamBits = 20;
constexpr int kSlotSpill = 0xFFF;
constexpr int kSlotNoCamera = (1 << kSlotCamBits) - 1;   // camera field of a valid slot WITHOUT a camera cell when the accumulators are not all in LDS

// ---------------------------------------------------------------------------
// Host-side analysis (plan.cc).  Pure C++, unit-testable without a GPU through
// the ceres_hip_debug_* exports.
// ---------------------------------------------------------------------------
struct HostStructure {
  int nrb = 0, ncb = 0, nelim = 0, ncells = 0;
  std::vector<int32_t> rsz, rpos, csz, cpos, rptr, ccol, cval;
  int num_rows = 0, num_cols = 0, num_cols_e = 0, num_cols_f = 0, num_row_blocks_e = 0;
  int64_t nnz = 0;
  int64_t values_extent = 0;  // 1 + largest value index any cell touches
  int max_block = 0;
  int det_row = 0, det_e = 0, det_f = 0;  // DetectStructure; -1 dynamic, 0 none
  // transpose (column block -> cells, in row order)
  std::vector<int32_t> tptr, trow, tcell;
  // scalar row/col -> owning block (for thread-per-scalar generic kernels)
  std::vector<int32_t> row_block_of, col_block_of;
  // diagonal block stores
  std::vector<int64_t> diag_off_all;  // ncb+1, offsets of csz^2 blocks over all column blocks
  std::vector<int64_t> diag_off_f;    // nf+1
  std::vector<int64_t> diag_off_e;    // nelim+1
  // E block of each row (-1 if the row has no E cell)
  std::vector<int32_t> row_e_block;
  // chunk = rows of one E block (contiguous): start/size per E block; size 0 if none
  std::vector<int32_t> chunk_start, chunk_size;
  bool chunks_contiguous = true;
};

// SHAPES of the fused path.  A conforming row is 2 high and holds one point cell (2 x 3), at most ONE camera cell (2 x nf: the cameras
// of a problem are all nf wide) and any cells on a few SHARED column blocks (libmv's intrinsics, examples/libmv_bundle_adjuster.cc:718-722:
// AddResidualBlock(cost, NULL, camera_intrinsics, current_camera_R_t, &point->X(0)) = <2, 8, 6, 3>), which the tiles hold as a dense
// 2 x ns STRIP per row (ns = the shared blocks' widths together, rounded up to a strip width that is compiled; zero where a row has no
// cell).  The kernels are compiled once per (nf, ns) — kernels_bal.inc, one translation unit each — for the camera widths of the
// reference's static specialisations with a 3-wide E block (internal/ceres/generate_template_specializations.py:55-75: (2,3,3), (2,3,4),
// (2,3,6), (2,3,9)), 8 (its (2,4,8) camera), 10 (bundle_adjuster --use_quaternions: examples/snavely_reprojection_error.h:164), and strips
// of 4 / 8 scalars next to the 6- and 9-wide cameras.
constexpr int kMaxSharedScalars = 8;   // strip width limit (BalArgs::sh_pos)
constexpr int kMaxSharedCellsPerRow = 2;
// Point blocks: 3 wide, and — round 5 — the other E widths the reference specialises with a 2-high row
// (generate_template_specializations.py:55-75): (2,2,2) (2,2,3) (2,2,4) and (2,4,3) (2,4,4) (2,4,6) (2,4,8) (2,4,9), no strip.
// Row blocks: 2 high, and — round 5 — the reference's remaining static specialisations (3,3,3), (4,4,2), (4,4,3), (4,4,4), no strip.
inline bool BalShapeCompiled(int nr, int ne, int nf, int ns) {
  if (nr == 3) return ne == 3 && nf == 3 && ns == 0;
  if (nr == 4) return ne == 4 && ns == 0 && (nf == 2 || nf == 3 || nf == 4);
  if (nr != 2) return false;
  // (round 6: every camera width 2 .. 10 next to 3- and 4-wide points — what the reference's dynamic-size specialisations (2,3,d),
  // (2,4,d) cover in practice: 5-wide cameras without distortion, 7-wide quaternion poses — and 6 / 9 next to 2-wide points)
  if (ne == 2) return ns == 0 && (nf == 2 || nf == 3 || nf == 4 || nf == 6 || nf == 9);
  if (ne == 4) return ns == 0 && nf >= 2 && nf <= 10;
  if (ne != 3) return false;
  if (ns == 0) return nf >= 2 && nf <= 10;
  return (nf == 6 || nf == 9) && (ns == 4 || ns == 8);
}
inline int BalStripWidthFor(int ns_used) { return ns_used <= 0 ? 0 : (ns_used <= 4 ? 4 : (ns_used <= 8 ? 8 : -1)); }

struct BalPlan {
  bool eligible = false;
  std::string why_not;         // reason the fused path was not selected
  int n_points = 0, n_cameras = 0;
  int nr = 2;                  // height of the row blocks (2, 3 or 4: BalShapeCompiled)
  int ne = 3;                  // width of the point blocks (2, 3 or 4: BalShapeCompiled)
  int nf = 9;                  // width of the camera blocks
  int ns_used = 0, ns = 0;     // shared strip: scalars in use, compiled strip width (BalStripWidthFor)
  std::vector<int32_t> sh_block;   // shared column blocks, in strip order
  std::vector<int32_t> sh_off;     // strip offset of each of them
  std::vector<int32_t> sh_pos;     // ns_used entries: scalar position of each strip scalar in x (minus num_cols_e)
  int cam_base = 0;                // cameras_contiguous: cam_pos[c] == cam_base + nf c
  // per slot, per shared cell of its row (kMaxSharedCellsPerRow lists of n_tiles * 64): value offset (-1: none) and strip offset | width << 8
  std::vector<int32_t> slot_hpos[kMaxSharedCellsPerRow], slot_hdesc[kMaxSharedCellsPerRow];
  int64_t n_cam_cells = 0;         // observations WITH a camera cell (a locked camera leaves rows without one)
  int64_t n_obs = 0, n_tiles = 0;
  bool contiguous_layout = false;  // pt_pos = 3p, cam_pos(F-relative) = cam_base + nf c
  bool points_contiguous = false, cameras_contiguous = false;  // each half of it (points may be renumbered: plan.cc)
  bool caller_contiguous = false;  // the CALLER's column layout is points-then-cameras back to back (before any renumbering)
  bool renumbered = false;         // internal point p is the caller's point of column block pt_block[p], at pt_pos[p]
  std::vector<int32_t> pt_block, cam_block;   // column block of each point / camera
  std::vector<int32_t> pt_pos, cam_pos;       // scalar offset in x (camera: minus num_cols_e)
  // per slot (n_tiles * 64)
  std::vector<int32_t> slot_epos, slot_fpos, slot_bpos;  // value / residual offsets, -1 = padding
  std::vector<int32_t> slot_cam, slot_pt;                // ids, -1 = padding
  std::vector<int32_t> slot_row;                         // row block of the slot, -1 = padding
  std::vector<int32_t> mo_index;                         // where a slot's M_o record lives: its place in the camera-major list (hybrid plans: its row); empty: at the slot
  // first | last<<8 | valid<<16 | tailA<<17 | hasA<<23 | tailB<<24 | hasB<<30   (tails: see plan.cc)
  std::vector<uint32_t> slot_seg;
  // per tile: 0 normal (tile_aux = longest track | #points<<8), 1 / 3 head of a long point (tile_aux = #tiles; 3: the streaming
  // kernels take it in a round), 2 continuation
  std::vector<int32_t> tile_kind, tile_aux;
  // Long points sit behind the normal tiles of their range (a hybrid group, or everything): long_ptr[g] = first long tile of range g.
  // round_word: kRoundWaves words per round, round_flag: kRoundSum / kRoundApply / kRoundLast (plan.cc); sequence q = the rounds
  // [seq_ptr[q], seq_ptr[q + 1]) that ONE workgroup takes in order; sequences [round_ptr[g], round_ptr[g + 1]) belong to range g.
  bool long_behind = false;
  std::vector<int32_t> long_ptr, round_ptr, seq_ptr, round_flag;
  std::vector<uint32_t> round_word;
  std::vector<int32_t> tile_pt0;  // point id of lane 0 of each tile (every tile's lane 0 is a valid slot)
  // camera-major lists
  std::vector<int32_t> cam_ptr;    // n_cameras+1
  std::vector<int32_t> cam_fpos;   // F value offset of each observation, camera-major
  std::vector<int32_t> cam_slot;   // M_o record of each observation, camera-major (see mo_index: the identity, or the row in hybrid plans)
  // work items of the camera-block kernel: (camera, [begin,end) in the camera-major list)
  std::vector<int32_t> item_cam, item_begin, item_end;
  std::vector<int32_t> cam_item_ptr;  // n_cameras+1: items [cam_item_ptr[c], cam_item_ptr[c+1]) belong to camera c
  // Cameras whose 9-double accumulators do not fit in LDS (more than ~2270): the tile pass leaves F^T z per slot and a
  // camera-major pass sums it.  Both can run CHUNK by chunk of tiles through a ring buffer (plan.cc: default one chunk).
  bool cameras_in_lds = true;
  std::vector<int32_t> slot_word;            // per slot, what the kernels read: camera | accumulator row << kSlotCamBits (plan.cc)
  // Accumulators of ALL cameras in LDS (cameras_in_lds): the LDS left over holds a copy of x_f for the xhot_cam.size() most observed
  // cameras (row r = camera xhot_cam[r]); their slots carry r + 1 in the (otherwise unused) row field of the word.  The streaming
  // kernels read those cameras' part of x from LDS: the gathers of the popular cameras' 72 bytes were what S.x / JtJx lost against
  // the bare tile stream (round 5: profiles/r05m_*).
  std::vector<int32_t> xhot_cam;
  int64_t z_ring_rows = 0;                   // 72-byte rows of the largest chunk's ring (spilled slots + the hybrid flush rows)
  std::vector<int32_t> tile_zbase;           // per tile: ring row of its first spilled slot (chunk-relative)
  std::vector<int32_t> zc_tile_ptr;          // n_chunks+1 tile boundaries (never inside a long point)
  std::vector<int32_t> zc_unit_ptr;          // n_chunks+1 into the unit arrays
  std::vector<int32_t> zu_cam, zu_begin, zu_end;  // unit = <= kZUnit entries of ONE camera inside ONE chunk; [begin, end) into zc_slot
  std::vector<int32_t> zu_shared;            // 1: the camera has more units in this chunk (combine with atomics), 0: plain read-modify-write
  std::vector<int32_t> zc_slot;              // per entry: ring row (chunk-relative); chunk-major, camera-major inside
  // Hybrid accumulation (plan.cc): hyb_groups workgroups with hyb_rows LDS accumulator rows each (the first hyb_hot of them the
  // same popular cameras everywhere, the rest the workgroup's own window); group g walks tiles [grp_tile_ptr[g], grp_tile_ptr[g+1])
  // and flushes its rows to ring rows [z_flush_row0 + g hyb_rows, + hyb_rows)
  bool hybrid = false;
  int hyb_groups = 0, hyb_rows = 0, hyb_hot = 0;
  int64_t z_flush_row0 = 0, n_local_obs = 0;  // n_local_obs: observations summed in LDS (the others are spilled)
  std::vector<int32_t> grp_tile_ptr;
  int max_track = 0, max_camera_degree = 0;
  // REMAINDER: the row blocks that are not "one point cell + camera-side cells" but touch camera blocks only
  // (rows without an E block: priors / regularisers on cameras, the rows SchurEliminator::NoEBlockRowsUpdate handles,
  // I/schur_eliminator_impl.h:574-666, and PartitionedMatrixView's second loops, I/partitioned_matrix_view_impl.h:171-190).  The fused
  // tiles cover the other (conforming) rows; the remainder's contributions are sums over its rows and are added by small generic
  // kernels.  rem_list: their row blocks, ascending.  With an elimination order they trail, [rem_row0, nrb); without one they may sit
  // anywhere (rem_row0 = -1 unless they happen to trail).  slot_row / mo_index / cam_slot hold COMPACT ids of the conforming rows
  // (= the row block itself when the remainder trails).
  int rem_row0 = 0, n_rem_rows = 0;
  std::vector<int32_t> rem_list;
};

// Storage of an EXPLICIT Schur complement: what SparseSchurComplementSolver::InitStorage builds
// (I/schur_complement_solver.cc:224-290) — the set of block pairs (i <= j) of F blocks that share a chunk (or an E-free
// row), as a BlockRandomAccessSparseMatrix (I/block_random_access_sparse_matrix.cc:51-110: cells of a block row are
// consecutive, each cell row-major n_i x n_j) — plus, per pair, the list of (chunk, cell of block i, cell of block j)
// contributions, so that SchurEliminator::Eliminate becomes a gather per stored block: deterministic, no atomics, no mutexes.
struct SchurStorage {
  int nf = 0;
  std::vector<int32_t> pair_i, pair_j;      // F-relative block ids, sorted by (i, j), i <= j; every (i, i) is present
  std::vector<int64_t> pair_off;            // npairs + 1 value offsets
  std::vector<int32_t> row_ptr;             // nf + 1: pairs whose first block is i
  std::vector<int32_t> col_ptr, col_pair;   // nf + 1 / list: pairs (j, i) with j < i, per second block i (the transpose half)
  std::vector<int64_t> trip_ptr;            // npairs + 1
  std::vector<int32_t> trip_e, trip_k1, trip_k2;  // E block of the chunk (-1: an E-free row), cell of block i, cell of block j
  std::vector<int32_t> cell_row;            // row block of every cell
  // Work ITEMS of the elimination: at most kSchurItem consecutive triples of ONE pair (a popular camera's diagonal block sums
  // thousands: one wavefront walking that list alone was the whole kernel's duration); items of a pair are consecutive
  std::vector<int32_t> item_pair;           // n_items
  std::vector<int64_t> item_t0, item_t1;    // triples [t0, t1)
  std::vector<int64_t> item_off;            // n_items + 1: where an item's partial block (n_i x n_j values) sits in the scratch
  std::vector<int32_t> pair_item_ptr;       // npairs + 1
  int64_t num_values() const { return pair_off.empty() ? 0 : pair_off.back(); }
  int64_t scratch_values() const { return item_off.empty() ? 0 : item_off.back(); }
};
void BuildSchurStorage(const HostStructure& hs, SchurStorage* out);

// Fills hs from the ABI structure; returns "" or an error message.
std::string AnalyzeStructure(const ceres_hip_block_structure& bs, int nelim, HostStructure* hs);
// Decides whether the fused <2,3,9> path applies and, if so, builds the packing plan.
// hyb: the workgroups and LDS accumulator rows of the tile pass (hybrid camera accumulation when the cameras do not fit in LDS;
// needs reorder_points; groups = 0: never)
constexpr int kRoundWaves = 8;               // waves of a streaming workgroup = tiles of a round of long points (plan.cc)
constexpr uint32_t kRoundIdle = 0xFFFFFFFFu;  // round word of a wave without a tile
constexpr int kRoundSum = 1, kRoundApply = 2, kRoundLast = 4;   // round flags of a point of more than kRoundWaves tiles (plan.cc)
struct HybridRequest { int groups = 0, rows = 0; int64_t lds_bytes = 0; };   // rows == 0: as many as lds_bytes hold next to eight waves' spill strips, for the plan's camera width
// reorder_mode: renumber the points (fuller tiles; hybrid groups) never / always (Schur solvers: no CG vector lives in point space) /
// only if the caller's layout is points-then-cameras back to back (CGNR: its CG vectors then ARE the caller's with the points renumbered)
constexpr int kReorderNever = 0, kReorderAlways = 1, kReorderIfContiguous = 2;
void BuildBalPlan(const HostStructure& hs, int reorder_mode, const HybridRequest& hyb, BalPlan* plan);

}  // namespace chip
#endif
