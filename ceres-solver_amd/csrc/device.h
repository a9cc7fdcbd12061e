// device.h — kernel argument structs and launcher declarations shared by the .hip files.
#ifndef CERES_HIP_DEVICE_H_
#define CERES_HIP_DEVICE_H_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace chip {

int BalBlockFor(int mode);                   // threads per workgroup of the fused kernels (512 | 1024)
constexpr size_t kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950
constexpr int kVecBlock = 256;
constexpr int kMaxVecGrid = 512;             // partial sums per inner product

// ---- one-shot peer-to-peer exchange (p2p.h: the device side; kernels_cg.hip: the stand-alone all-reduce) ----
constexpr int kP2pMaxWorld = 8;    // one node: 8 GPUs on the xGMI mesh
constexpr int kP2pChunk = 512;     // slot capacity is rounded up to a multiple of this
constexpr int kP2pFlagChunk = 8;   // arrival flags per (parity, rank): cap / 8 — a chunk of p2p_exchange_wave is at most one wavefront of slots,
                                   // and may be as few as the 8 packed sums of a 2-wide camera block (bal_invert_exchange_kernel: a chunk per camera)
constexpr int kP2pMaxGrid = 256;   // workgroups of one all-reduce (each takes every kP2pMaxGrid-th chunk): never the whole device spinning
struct P2pPeers {
  double* slots[kP2pMaxWorld];               // rank q's receive slots [2][world][cap], as mapped into THIS process
  unsigned long long* flags[kP2pMaxWorld];   // rank q's arrival flags [2][world][chunks_cap]
};
struct P2pComm {   // what a kernel needs of the communicator (by value)
  P2pPeers peers;
  int rank = 0, world = 1;
  unsigned long long epoch = 0;
  long long cap = 0;          // doubles per slot
  int chunks_cap = 0;         // flags per (parity, source rank)
  int* error_flag = nullptr;  // mapped host memory
  int* error_seen = nullptr;  // device memory
  long long timeout_ticks = 0;
  int fences = 0;             // 1: system-scope release / acquire fences around the flags as well (p2p.h)
};

// ---- fused <2,3,9> kernels (kernels_bal.hip) ------------------------------
enum BalMode { kBalSx = 0, kBalJtJx = 1, kBalJtb = 2, kBalInit = 3, kBalEte = 4, kBalBackSub = 5, kBalCgnrInit = 6, kBalColNorm = 7, kBalJx = 8, kBalSpseZ = 9,
               kBalShBlocks = 10 /* the strip's own diagonal block sums, per workgroup into BalArgs::scalar_out (shapes with a shared strip) */ };

struct CgScalars;
// The rest of a CG iteration at the end of the S.x pass, for camera spaces of at most kCgTailMax scalars (run_cg, solver.hip): the
// workgroup that flushes its partial sums LAST (an atomic ticket; nobody waits for anybody) adds the partials up, and does what
// bal_reduce_partials_kernel, cg_update_kernel and cg_finalize_direction_kernel do for such a vector in three more launches of
// ~5 us each — most of a CG iteration on a problem like problem-16-22106 (BASELINE.json configs[0]: 144 camera scalars).
constexpr int kCgTailMax = 512;     // 56 cameras: at least one thread of the last workgroup per camera scalar
constexpr int kCgTailLoads = 64;    // partial sums one thread of the last workgroup adds up (all in flight at once): workgroups <=
                                    // kCgTailLoads * (512 / camera scalars), or the iteration runs as the usual three kernels
struct CgTail {
  int enabled = 0, it = 0;
  unsigned int* ticket = nullptr;   // zero between launches
  double *x = nullptr, *r = nullptr, *p = nullptr;   // (q = S p arrives in BalArgs::y_f... the caller's z, which becomes M^-1 r)
  double* z = nullptr;
  const double* rhs = nullptr;
  const double* blocks = nullptr;   // M^-1: 81 doubles per camera, back to back
  const double* D_f = nullptr;
  CgScalars* S = nullptr;
};

struct BalArgs {
  // packed problem
  const double2* J = nullptr;   // [n_tiles][3 + nf + ns][64]
  const float4* Jf = nullptr;   // [n_tiles][6][64]  fp32 storage mode (then J is unused)
  const double2* b = nullptr;   // [n_tiles][b_pairs][64]: the slot's nr residuals (2-high rows: one double2 per slot)
  // fused re-layout: when src_values != nullptr the kernel gathers from the caller's layout and
  // writes the tiles (J_out, b_out) as it goes (first pass of a step)
  const double* src_values = nullptr;
  const double* src_b = nullptr;
  const int32_t *slot_epos = nullptr, *slot_fpos = nullptr, *slot_bpos = nullptr;   // slot_fpos < 0: a row without a camera cell
  // shapes with a shared strip: the row's shared cells (value offset, -1 none; strip offset | width << 8), common.h
  const int32_t* slot_hpos[kMaxSharedCellsPerRow] = {};
  const int32_t* slot_hdesc[kMaxSharedCellsPerRow] = {};
  double2* J_out = nullptr;
  float4* Jf_out = nullptr;
  double2* b_out = nullptr;
  const int32_t* slot_cam = nullptr;
  const int32_t* tile_pt0 = nullptr;  // first point id of each tile (points of a tile are consecutive ids)
  const uint32_t* slot_seg = nullptr;
  const int32_t* tile_kind = nullptr;
  const int32_t* tile_aux = nullptr;
  int64_t n_tiles = 0, n_slots = 0;
  // tiles this launch walks: [tile_begin, tile_end); tile_end = 0 means all.  Chunked launches (cameras not in LDS) write their
  // per-slot F^T z into a ring buffer indexed by slot - z_slot0.
  int64_t tile_begin = 0, tile_end = 0;
  // cameras not in LDS (plan.cc): tile_zbase = ring row of each tile's first spilled slot; hybrid accumulation: workgroup g walks the
  // tiles [grp_tile_ptr[g], grp_tile_ptr[g + 1]) with hyb_rows accumulator rows in LDS, flushed to ring rows z_flush_row0 + g hyb_rows ..
  const int32_t* tile_zbase = nullptr;
  const int32_t* grp_tile_ptr = nullptr;
  int hyb_rows = 0;
  // long points (plan.cc): the tiles of range g (the hybrid group, or 0 = everything) from long_ptr[g] on belong to points of more
  // than 64 observations; the streaming kernels take them in rounds of kRoundWaves words, sequence by sequence (common.h, plan.cc)
  int long_behind = 0;   // 0: long points sit among the normal tiles (chunked ring), no rounds
  const int32_t* long_ptr = nullptr;
  const int32_t* round_ptr = nullptr;    // per range: its sequences
  const int32_t* seq_ptr = nullptr;      // per sequence: its rounds
  int n_seq = 0;                         // sequences of all ranges
  const int32_t* round_flag = nullptr;
  const uint32_t* round_word = nullptr;
  int64_t z_flush_row0 = 0;
  int pq_accumulate = 0;       // kJtJx chunked: pq_out[workgroup] += instead of = (later chunks of one application)
  const int32_t* pt_pos = nullptr;   // nullptr => ne*p (ne: the shape's point width)
  const int32_t* cam_pos = nullptr;  // nullptr => cam_base + nf*c   (relative to the F base pointer)
  int cam_base = 0;
  int sh_pos[kMaxSharedScalars] = {};   // the strip's scalars in the F-space vectors (shapes with a shared strip)
  // CGNR on internally numbered points: x_e / y_e / point_blocks are internal (pt_pos == nullptr) while D_e, lm_diag_e, lm_D_e are the
  // caller's: d_pos[p] = the caller's offset of internal point p (nullptr: the same offsets as x_e); D_int_out: D_e in the internal order
  const int32_t* d_pos = nullptr;
  double* D_int_out = nullptr;
  // vectors: *_e indexed by pt_pos, *_f by cam_pos
  const double* x_e = nullptr;
  const double* x_f = nullptr;
  int flags = 0;                    // 1: per-lane point-space accesses in JtJx, 2: blocked tile walk (experiments); 4: plain (not non-temporal) tile stores
  double* y_e = nullptr;
  const double* D_e = nullptr;  // nullptr => no regularisation on the point part
  // fused LM diagonal (lm_radius > 0): D_e is formed in the kernel from the point block's own diagonal
  double lm_radius = 0.0, lm_min = 0.0, lm_max = 0.0;
  double* lm_diag_e = nullptr;  // clamped diag(J^T J), point part (indexed like x_e)
  double* lm_D_e = nullptr;     // D, point part
  // per-point ne x ne inverses, packed upper triangle at a pitch of BalOps::etei_pitch doubles (3-wide points: 6 doubles / point)
  double* etei = nullptr;
  double* point_blocks = nullptr;         // dense ne x ne output (CGNR JACOBI) or nullptr
  const int64_t* pt_diag_off = nullptr;   // offsets into point_blocks; nullptr => ne*ne*p
  double* Mo = nullptr;                   // [n_slots][mo_pitch] symmetric nr x nr per observation, packed upper triangle (2-high rows: m00 m01 m11 m01) (kInit)
  const int32_t* mo_index = nullptr;      // record of each slot in Mo (nullptr: the slot itself; hybrid plans: the slot's row)
  int have_b = 0;
  // camera accumulation
  double* partials = nullptr;    // [grid][n_acc]   (LDS mode)
  double* zbuf = nullptr;        // [rows][nf]     (cameras do not fit in LDS: the ring of spilled / flushed F^T z rows, second pass by camera)
  double* strip_sums = nullptr;  // accumulators outside LDS, shapes with a strip: the strip's ns entries of the global sums (zeroed per application)
  int n_acc = 0;                 // nf * n_cameras + ns: the camera-space accumulator entries (the strip's behind the cameras')
  // LDS mode, streaming kernels: x_f of the n_xhot most observed cameras staged in LDS behind the accumulators (BalPlan::xhot_cam;
  // the slot word's row field = row + 1)
  const int32_t* xhot_cam = nullptr;
  int n_xhot = 0;
  // kBackSub / kJx (no accumulators in LDS): the whole camera part of x staged in LDS, nf * n_cameras scalars (0: gathered from
  // memory; the dispatcher clears it for every other mode)
  int x_lds_scalars = 0;
  double* scalar_out = nullptr;  // kJx: one partial sum per workgroup
  // kBackSub: the reduced solution z is also the camera part of x (ImplicitSchurComplement::BackSubstitute copies it): done by the kernel
  const double* copy_src = nullptr;
  double* copy_dst = nullptr;
  int copy_n = 0;
  // kBackSub as the last kernel of an LM step: store -x (the step) instead of x and raise *nonfinite if an entry is not
  // finite (LevenbergMarquardtStrategy::ComputeStep's IsArrayValid + negation, I/levenberg_marquardt_strategy.cc:123-153)
  int negate_out = 0;
  int* nonfinite = nullptr;
  // Speculative tail of an LM step: the kernel is enqueued BEFORE the host knows whether CG has ended, and does nothing
  // unless *run_after_cg is a terminal status that lets the solution be used (CgStatusAllowsSolution)
  const int* run_after_cg = nullptr;
  double* pq_out = nullptr;      // kJtJx: partial x_e . y_e of the point part, one per workgroup (CG's p.q without a pass of its own)
  const int* status = nullptr;   // CG status word; non-zero => kernel returns immediately
  // first pass of a solve (kInit / kCgnrInit; neither raises them): the step's finite-step and factorization flags, two adjacent ints,
  // are cleared here instead of by a memset command in front of the pass (a fill kernel and its launch per step)
  int* clear_flags = nullptr;
  CgTail tail;                   // kSx, pipelined kernel, every camera's accumulator in LDS: finish the CG iteration (see CgTail)
};

// Fused LM diagonal of the camera columns, applied by bal_invert_kernel (radius > 0 to enable).
struct LmFuse {
  double radius = 0.0, min_d = 0.0, max_d = 0.0;
  const double* camsq = nullptr;     // [nf c + k] column norms; nullptr => the block's own diagonal (F^T F blocks)
  const int32_t* cam_pos = nullptr;  // nullptr => cam_base + nf c
  int cam_base = 0;
  double* diag_f = nullptr;
  double* D_f = nullptr;
};

// Work items of the camera-block kernel: <= kCamChunk consecutive observations of one camera.
struct CamItems {
  const int32_t *cam = nullptr, *begin = nullptr, *end = nullptr;
  int count = 0;
  int64_t observations = 0;   // of all items together (the launcher picks the kernel by the mean item length)
  // <2,3,9> with the device evaluator (bal_frontend.inc): the pass EVALUATES its F cells instead of reading them — the records the
  // tile-order evaluator left ([state | scale] per camera / point), and the point and pixel of every entry of the camera-major list
  const double *ev_cam_pack = nullptr, *ev_pt_pack = nullptr;
  const int32_t* ev_pt = nullptr;
  const double2* ev_obs = nullptr;
};
struct CamGather {
  const double* parts = nullptr;          // nullptr: the blocks are already assembled in memory
  const int32_t* cam_item_ptr = nullptr;
  const double* D_f = nullptr;            // added squared to the diagonal (indexed through cam_pos)
  const int32_t* cam_pos = nullptr;
  int cam_base = 0;                       // cam_pos == nullptr: camera c at cam_base + nf c
  int want_sq = 0;                        // SCHUR items: the fused LM diagonal takes the camera columns' square sums from the items
  int few = 0;                            // every camera has a handful of items at most: seven cameras per wavefront gather their own
  const double* extra = nullptr;          // [nf nf c + nf a + b] raw sums added to the camera's block (rows outside the tiles: LaunchRemCameraBlocks)
  const double* packed = nullptr;         // [cam_part c + e] the cameras' item sums already added up, and summed over ranks (camera_exchange); parts unused
};
// Camera-major pass of one chunk (cameras not in LDS): acc[nf c + k] += sum over the unit's entries of ring[nf slot + k].
struct ZUnits {
  const int32_t *cam = nullptr, *begin = nullptr, *end = nullptr, *shared = nullptr;  // units [first, first + count)
  const int32_t* slot = nullptr;                                                       // entry -> ring row (chunk-relative)
  int first = 0, count = 0;
};

// accumulator entry -> place in the F-space vectors (bal_reduce_partials_kernel): camera c's scalars at cam_pos[c] (nullptr: cam_base +
// nf c), behind them the shared strip's at sh_pos
struct FMap {
  const int32_t* cam_pos = nullptr;
  int cam_base = 0, n_cam_scalars = 0;
  int sh_pos[kMaxSharedScalars] = {};
};

// the shared column blocks of a plan: strip offset, width, where the block goes in the F-block store, the F-space position of its first scalar
struct StripBlocks {
  int count = 0;
  int off[kMaxSharedScalars] = {}, width[kMaxSharedScalars] = {}, pos[kMaxSharedScalars] = {};
  int64_t out[kMaxSharedScalars] = {};
};

// The fused kernels are compiled once per SHAPE (camera width nf, shared strip ns: common.h, kernels_bal.inc — one translation unit
// per shape); this is one shape's launchers.
struct BalOps {
  // tile_pitch: double2 elements per tile; cam_part: doubles per item of the camera-major pass; etei_pitch: doubles per point in the store
  // of packed (E^T E)^-1; b_pairs: double2 per slot of the residual tiles; mo_pitch: doubles per M_o record
  int nr, ne, nf, ns, pairs, tile_pitch, cam_part, etei_pitch, b_pairs, mo_pitch, has_f32, has_cg_tail;
  hipError_t (*fused)(int mode, const BalArgs& A, bool lds, int grid, hipStream_t stream);
  // whether fused(kBalSx, A, lds, ..) runs the pipelined kernel — the one that can finish a CG iteration (A.tail)
  bool (*sx_runs_pipelined)(const BalArgs& A);
  // y_f[pos(i)] = sum over the workgroups' partials (+ D_f^2 x_f); pq_out != nullptr: also partial x_f . y_f, one per workgroup (*n_pq of
  // them; needs x_f); sum_out: *sum_out = sum(sum_in[0 .. n_sum_in)), see the kernel
  hipError_t (*reduce_partials)(const double* partials, int nparts, int n_acc, const FMap& map, const double* D_f, const double* x_f,
                                double* y_f, const int* status, double* pq_out, int* n_pq, hipStream_t stream, const double* sum_in,
                                int n_sum_in, double* sum_out);
  // the same for a SHARDED instance with the sum over ranks inside (p2p.h; one launch instead of reduction, all-reduce kernel and
  // add_f_diagonal): y_f = sum over ranks (sum over the workgroups' partials) + D_f^2 x_f over the n_fv F-space positions; *sum_out
  // (position n_fv) likewise summed over ranks.  Cameras back to back in F space only (map.cam_pos == nullptr).
  hipError_t (*reduce_exchange)(const double* partials, int nparts, int n_acc, int n_fv, const FMap& map, const double* D_f, const double* x_f,
                                double* y_f, const int* status, double* pq_out, int* n_pq, hipStream_t stream, const double* sum_in,
                                int n_sum_in, double* sum_out, const P2pComm& comm, int grid_cap);
  hipError_t (*stream_probe)(const double2* J, int64_t n_tiles, int grid, double* out, hipStream_t stream);
  hipError_t (*add_f_diagonal)(int n_acc, const FMap& map, const double* D_f, const double* x_f, double* y_f, const int* status,
                               double* pq_out, int* n_pq, hipStream_t stream);
  hipError_t (*pack)(const BalArgs& A, hipStream_t stream);   // A.src_values / src_b / slot_* / J_out (Jf_out) / b_out set
  hipError_t (*invert)(double* blocks, const int64_t* cam_diag_off, int n_cameras, int* fail_flag, const LmFuse& lm, const CamGather& gather,
                       hipStream_t stream);
  // sharded: every camera's items added up and summed over ranks (p2p.h: slots cam_part c + e, chunks ceil(cam_part / 64) c + q), left
  // packed — cam_part doubles per camera — for invert (CamGather::packed)
  hipError_t (*camera_exchange)(const double* parts, const int32_t* cam_item_ptr, int n_cameras, const double* extra, double* packed,
                                const P2pComm& comm, int grid_cap, hipStream_t stream, bool few /* a handful of items per camera */);
  // Per-camera blocks in two steps: every item (<= kCamChunk observations of one camera) leaves its upper-triangle sums + nf column
  // square sums in parts[item][cam_part]; the items of a camera (cam_item_ptr) are then added in list order either by camera_finish
  // (raw sums to memory, + D_f^2 if given) or by the load phase of invert (CamGather).
  hipError_t (*camera_items)(bool schur, const double* values, const CamItems& items, const int32_t* cam_fpos, const int32_t* cam_slot,
                             const double* Mo, double* parts, hipStream_t stream);
  hipError_t (*camera_finish)(const double* parts, const int32_t* cam_item_ptr, const double* D_f, const int32_t* cam_pos, int cam_base,
                              const int64_t* cam_diag_off, double* blocks, double* camsq, int n_cameras, hipStream_t stream, const double* extra,
                              bool few /* every camera has a handful of items at most: kPerWave cameras per wavefront */);
  hipError_t (*camera_chunk)(const ZUnits& units, const double* ring, double* acc, const int* status, hipStream_t stream);
  // shapes with a shared strip: the packed upper triangle of the strip's ns x ns matrix from the workgroups' partial sums (kBalShBlocks),
  // cut into the shared blocks' dense diagonal blocks (+ D^2) in the F-block store
  hipError_t (*strip_finish)(const double* parts, int nparts, const StripBlocks& sb, const double* D_f, double* blocks, hipStream_t stream);
};
const BalOps* GetBalOps(int nr, int ne, int nf, int ns);   // nullptr: not compiled (common.h: BalShapeCompiled)
// the dynamic-LDS ceiling of a kernel, raised once per (kernel, device)
hipError_t AllowMaxLds(const void* kernel);

// ---- generic kernels (kernels_generic.hip) --------------------------------
// Work ITEMS of the grouped generic kernels over heavy column blocks (the cameras of a BAL-like problem): at most kGenItem consecutive
// cells of ONE block's transpose list, never cutting the cells a block has inside one chunk apart.  A wave per item leaves the item's
// partial result in `scratch` ([item][kGenItemValues]); a second kernel adds a block's items in list order (deterministic).  Without
// them one wave walks a popular camera's thousands of cells alone and the kernel lasts as long as that chain (1.3 ms on the Ladybug shape).
constexpr int kGenItem = 128;
constexpr int kGenItemValues = 55;   // a 10 x 10 upper triangle
struct GenItems {
  int count = 0, first_block = 0, nblocks = 0;
  const int32_t *block = nullptr, *t0 = nullptr, *t1 = nullptr;   // per item
  const int32_t* block_ptr = nullptr;                             // per block of the range: its items [block_ptr[q], block_ptr[q + 1])
  double* scratch = nullptr;
};
struct GenStructure {
  int nrb = 0, ncb = 0, nelim = 0, nrbe = 0, num_rows = 0, num_cols = 0, nce = 0, ncf = 0;
  const int32_t *rsz = nullptr, *rpos = nullptr, *rptr = nullptr, *ccol = nullptr, *cval = nullptr;
  const int32_t *csz = nullptr, *cpos = nullptr;
  const int32_t *tptr = nullptr, *trow = nullptr, *tcell = nullptr;
  // per transpose entry, what the itemized kernels would otherwise chase through three more arrays: the cell's value offset, its row
  // block's first scalar row, and the row block's height | (1 << 8 if the cell is the E cell of its row)
  const int32_t *tval = nullptr, *trpos = nullptr, *tinfo = nullptr;
  const int32_t *row_block_of = nullptr, *col_block_of = nullptr, *row_e_block = nullptr;
  const int64_t *diag_off_all = nullptr, *diag_off_e = nullptr, *diag_off_f = nullptr;
  // chunk = the (contiguous) rows of one E block: SchurEliminator's unit of work (I/schur_eliminator_impl.h:87-181)
  const int32_t *chunk_start = nullptr, *chunk_size = nullptr;
  // hints for the GROUPED kernels (L lanes per column block / chunk, kernels_generic.hip): largest block sizes per part, lanes per group
  // chosen from the average number of cells per block (power of two, 4 .. 64); 0 = use the thread-per-scalar kernels
  int max_csz_e = 0, max_csz_f = 0, max_csz = 0, max_rsz = 0;
  int lanes_e = 0, lanes_f = 0, lanes_all = 0, lanes_chunk = 0;
  GenItems items;   // over the F blocks (all blocks without an elimination order); count == 0: none
};
enum GenPart { kAll = 0, kE = 1, kF = 2 };

// y += A_part x   (x, y are the base pointers of the part's own index space)
hipError_t LaunchGenRightMultiply(const GenStructure& G, const double* values, int part, const double* x, double* y,
                                  const int* status, hipStream_t stream);
// y += A_part^T x
hipError_t LaunchGenLeftMultiply(const GenStructure& G, const double* values, int part, const double* x, double* y,
                                 const int* status, hipStream_t stream);
// blocks = blockdiag(A_part^T A_part) (+ D^2 if D != nullptr; D is the FULL num_cols vector)
hipError_t LaunchGenBlockDiagonal(const GenStructure& G, const double* values, int part, const double* D,
                                  double* blocks, int64_t total_entries, hipStream_t stream);
// x[j] = |A_j|^2
hipError_t LaunchGenSquaredColumnNorm(const GenStructure& G, const double* values, double* x, hipStream_t stream);
// In-place SPD inverse (upper triangle authoritative) of nblocks dense blocks; blocks
// [first_block, first_block + nblocks) of the column-block list, offsets relative to off[0].
hipError_t LaunchGenInvertBlocks(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off,
                                 double* blocks, int* fail_flag, hipStream_t stream);
// Per chunk (E block e, rows of the chunk): w = (E^T E + D^2)^-1 E^T t over the chunk's rows (ete_inv: the inverted blocks), then
// t_r -= E_r w on those rows (update_t) and / or x_e[cpos(e) ..] = w (x_e != nullptr).  One launch for what LeftMultiplyE, the block
// diagonal apply and RightMultiplyE do in ImplicitSchurComplement::RightMultiplyAndAccumulate / UpdateRhs / BackSubstitute
// (I/implicit_schur_complement.cc:106-144, 208-276).  Returns hipErrorNotSupported when the block sizes exceed the grouped kernels'
// (the caller then runs the separate passes).
hipError_t LaunchGenChunkProject(const GenStructure& G, const double* values, const double* ete_inv, double* t_rows, int update_t,
                                 double* x_e, const int* status, hipStream_t stream);
// z_rows (rows of the chunks) = F x_f - E (E^T E)^-1 E^T F x_f in one launch (the row-space half of S x); hipErrorNotSupported beyond the
// compiled block sizes.  Rows WITHOUT an E block are not touched: LaunchGenRightMultiplyFrom(first_row = their first scalar row).
hipError_t LaunchGenChunkSx(const GenStructure& G, const double* values, const double* ete_inv, const double* x_f, double* z_rows,
                            const int* status, hipStream_t stream);
// LaunchGenRightMultiply over the scalar rows [first_row, num_rows)
hipError_t LaunchGenRightMultiplyFrom(const GenStructure& G, const double* values, int part, int first_row, const double* x, double* y,
                                      const int* status, hipStream_t stream);
// y += blockdiag x over column blocks [first_block, first_block+nblocks); vectors start at that block.
hipError_t LaunchGenBlockDiagonalApply(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off,
                                       const double* blocks, const double* x, double* y, const int* status,
                                       hipStream_t stream);
// blocks(j)(a,a) += D[col]^2 over column blocks [first_block, first_block+nblocks); D is the FULL vector.
hipError_t LaunchAddBlockDiagonalSquares(const GenStructure& G, int first_block, int nblocks, const int64_t* diag_off,
                                         const double* D, double* blocks, hipStream_t stream);
// Diagonal blocks of the Schur complement (SchurEliminator::Eliminate into a
// block-diagonal lhs): needs the inverted E^T E blocks.  D may be nullptr; add_f_diag
// controls whether D_f^2 is added (sharded runs add it after the all-reduce).
hipError_t LaunchGenSchurJacobi(const GenStructure& G, const double* values, const double* ete_inv, const double* D,
                                int add_f_diag, double* blocks, int64_t total_entries, hipStream_t stream);
// Dense S (upper block triangle) and nothing else; rhs comes from the ISC path.
// explicit Schur complement (f2): mirror the stored upper block triangle; y = S x; diagonal blocks of S
hipError_t LaunchGenSymmetrizeDense(const GenStructure& G, double* lhs, hipStream_t stream);
hipError_t LaunchGenDenseSymv(const double* S, int n, const double* x, double* y, const int* status, hipStream_t stream);
hipError_t LaunchGenExtractDiagBlocks(const GenStructure& G, const double* S, const int64_t* diag_off_f, double* blocks,
                                      hipStream_t stream);
hipError_t LaunchGenSchurDense(const GenStructure& G, const double* values, const double* ete_inv, const double* D,
                               double* lhs, hipStream_t stream);

// ---- remainder rows of the fused path: rows without a point cell, R = their own GenStructure (kernels_generic.hip); nf = the camera width ----
hipError_t LaunchRemCameraBlocks(const GenStructure& R, const double* values, const int32_t* cam_block, int n_cameras, int nf, double* out, hipStream_t stream);
// y_f[pos(c) ..] += F_R^T t over the remainder rows' cells on the nf-wide camera blocks (one wavefront per camera)
// (cam_pos: each camera's offset in the F-space vectors, or nullptr: back to back from cam_base)
hipError_t LaunchRemLeftMultiply(const GenStructure& R, const double* values, const int32_t* cam_block, const int32_t* cam_pos, int cam_base, int n_cameras,
                                 int nf, const double* t_rows, double* y_f, const int* status, hipStream_t stream);
hipError_t LaunchRemAddDiag(const double* blocks, const int32_t* cam_pos, int cam_base, int n_cameras, int nf, double* y, hipStream_t stream);
hipError_t LaunchGatherRows(const double* in, const int32_t* map, int n, double* out, hipStream_t stream);   // out[r] = in[map[r]]
hipError_t LaunchRemModelCost(const double* m, const double* f, int n, int mode, double* out, hipStream_t stream);

// ---- explicit Schur complement solvers (kernels_schur.hip) ------------------
struct SchurPairs {  // device image of SchurStorage (common.h)
  int npairs = 0;
  const int32_t *pair_i = nullptr, *pair_j = nullptr, *row_ptr = nullptr, *col_ptr = nullptr, *col_pair = nullptr;
  const int64_t *pair_off = nullptr, *trip_ptr = nullptr;
  const int32_t *trip_e = nullptr, *trip_k1 = nullptr, *trip_k2 = nullptr, *cell_row = nullptr;
  // work items of the elimination (common.h): item -> pair, triples [t0, t1), offset of its partial block in `scratch`
  int n_items = 0;
  const int32_t *item_pair = nullptr, *pair_item_ptr = nullptr;
  const int64_t *item_t0 = nullptr, *item_t1 = nullptr, *item_off = nullptr;
  double* scratch = nullptr;
  int64_t total_values = 0;   // values of the block-sparse S
};
// S (block-sparse, upper block triangle) = SchurEliminator::Eliminate; D may be nullptr (then no D_f^2 on the diagonal)
hipError_t LaunchSchurSparseEliminate(const GenStructure& G, const SchurPairs& P, const double* values, const double* ete_inv,
                                      const double* D, double* S, hipStream_t stream);
// dense lhs (num_cols_f^2, zeroed by the caller) <- the stored blocks of S (`total` values): DENSE_SCHUR eliminates into the block-sparse
// storage (a gather per block) and factors the dense image
hipError_t LaunchSchurBlocksToDense(const GenStructure& G, const SchurPairs& P, const double* S, int64_t total, double* lhs, hipStream_t stream);
// y (+)= S x   BlockRandomAccessSparseMatrix::SymmetricRightMultiplyAndAccumulate
hipError_t LaunchSchurSparseSymv(const GenStructure& G, const SchurPairs& P, const double* S, const double* x, double* y,
                                 const int* status, int accumulate, hipStream_t stream);
hipError_t LaunchSchurSparseDiag(const GenStructure& G, const SchurPairs& P, const double* S, const int64_t* diag_off_f, double* blocks,
                                 int64_t total, hipStream_t stream);
// dense SPD n x n, upper triangle authoritative: in-place factor (L in the lower triangle), then x <- A^-1 x
hipError_t LaunchDenseCholesky(double* A, int n, int* fail_flag, hipStream_t stream);
hipError_t LaunchDenseCholeskySolve(const double* A, int n, double* x, hipStream_t stream);

// ---- vector kernels + device-resident CG (kernels_cg.hip) ------------------
hipError_t LaunchSet(double* x, double v, int64_t n, hipStream_t stream);
hipError_t LaunchAxpby(double a, const double* x, double b, const double* y, double* z, int64_t n, hipStream_t stream);
// y = D^2 .* x  (D may be nullptr => y = 0)
hipError_t LaunchSquareScale(const double* D, const double* x, double* y, int64_t n, const int* status, hipStream_t stream);
// y += D^2 .* x
hipError_t LaunchAddSquareScale(const double* D, const double* x, double* y, int64_t n, const int* status, hipStream_t stream);
// out[0] = x . y  (two-stage, deterministic); partials has kMaxVecGrid doubles
hipError_t LaunchDot(const double* x, const double* y, int64_t n, double* partials, double* out, hipStream_t stream);
// D = sqrt(clamp(diag, lo, hi) / radius); diag is clamped in place (LevenbergMarquardtStrategy::ComputeStep,
// I/levenberg_marquardt_strategy.cc:84-96)
hipError_t LaunchLmDiagonal(double* diag, double lo, double hi, double radius, double* D, int64_t n, hipStream_t stream);
// x = -x and *nonfinite += number of non-finite entries (IsArrayValid + negation, :124-132)
// CGNR: y.g + y.r + |D y|^2 over [begin, end) as per-workgroup partials (nparts <= kMaxVecGrid)
// neg_out (optional): also neg_out[i] = -y[i] over the range and *nonfinite += (entries that are not finite) — the LM step's
// finite check + negation, read from the CG solution in the same pass
// gate (optional): a CG status word; the kernel does nothing unless CgStatusAllowsSolution(*gate) (speculative LM tail)
// CGNR on internally numbered points (solver.hip): the CG vectors hold internal point p at [3 p, 3 p + 3), the caller's vectors at
// pt_pos[p]; D_e = the point part of D in the internal order
struct PointPerm {
  const int32_t* pt_pos = nullptr;
  const double* D_e = nullptr;
  int64_t n_e = 0;
};
hipError_t LaunchCgnrModelCost(const double* y, const double* g, const double* r, const double* D, int64_t begin, int64_t end,
                               double* partials, int* nparts, hipStream_t stream, double* neg_out = nullptr, int* nonfinite = nullptr,
                               const int* gate = nullptr, const PointPerm& perm = PointPerm());
// the 3-wide point blocks [0, n_e) between the caller's order and the internal one (to_internal: out[i] = in[pos(i)]); [n_e, n) copied
hipError_t LaunchPermutePoints(const double* in, double* out, const int32_t* pt_pos, int64_t n_e, int64_t n, bool to_internal, hipStream_t stream);
hipError_t LaunchNegateAndCheck(double* x, int64_t n, int* nonfinite, hipStream_t stream);
// values(cell)[r][c] *= scale[col]: BlockSparseMatrix::ScaleColumns (I/block_sparse_matrix.cc:403-450)
hipError_t LaunchGenScaleColumns(const GenStructure& G, double* values, const double* scale, hipStream_t stream);
// dense[off[p] ..] (ne x ne) = the symmetric matrix whose packed upper triangle sits at packed[p * pitch ..]
hipError_t LaunchExpandSym(const double* packed, int ne, int pitch, double* dense, const int64_t* pt_diag_off, int n_points, hipStream_t stream);
// out[off[p] + k] = in[9 p + k], k < 9 (in and out must not overlap)
hipError_t LaunchScatterBlocks9(const double* in, double* out, const int64_t* off, int n_points, hipStream_t stream);

// Status word values (device side) — 0 means "keep iterating".
enum CgStatus {
  kCgRunning = 0,
  kCgConvergedZeta = 1,
  kCgConvergedResidual = 2,
  kCgMaxIterations = 3,
  kCgFailRho = 4,
  kCgFailBeta = 5,
  kCgIndefinite = 6,
  kCgFailAlpha = 7,
  kCgZeroRhs = 8,          // |b| = 0: x = 0, SUCCESS
  kCgInitialResidual = 9,  // min_num_iterations == 0 and |r0| <= tol
  kCgSetupFailed = 10,     // the preconditioner blocks could not be factorized (flag checked by the init kernels, no host round trip)
};
// Terminal CG states whose solution the caller uses: SUCCESS and NO_CONVERGENCE (solve_loaded back-substitutes for these and
// for no others; fill_summary maps the rest to FAILURE).
__host__ __device__ inline bool CgStatusAllowsSolution(int st) {
  return st != kCgRunning && st != kCgFailRho && st != kCgFailBeta && st != kCgFailAlpha && st != kCgSetupFailed;
}

struct CgScalars {
  double norm_rhs, tol_r, q_tol;
  double rho, rho_new, beta, pq, alpha, Q0, Q1, zeta, norm_r, norm_p, norm_q;
  int iter, status, min_it, max_it;
  int fail_dir, fail_step;  // failures staged by direction / step, committed by finalize
  // Fused iteration (cg_update_kernel + cg_finalize_direction_kernel): rho_i and the Q0 that iteration i tests
  // against live at index i & 1, so that no multi-workgroup kernel reads a scalar another workgroup of the same
  // launch writes (iteration i reads [i & 1] and writes [(i + 1) & 1]).
  double rho_pp[2], Q0_pp[2];
};

struct CgBuffers {
  int64_t n = 0;
  // Sharded CGNR: the first n_local elements (this rank's points) are a shard, the rest
  // (cameras) are replicated.  Workgroups [0, grid_e) cover the shard, [grid_e, grid) the
  // rest; the shard's share of every inner product is collapsed into comm[slot], summed
  // over ranks by the host (ncclAllReduce) and added by the consumers.  Unsharded:
  // n_local = 0, grid_e = 0 and comm is ignored.
  int64_t n_local = 0;
  int grid = 1, grid_e = 0;
  double *x = nullptr, *r = nullptr, *p = nullptr, *z = nullptr;
  const double* rhs = nullptr;
  double* partials = nullptr;  // 4 slots * kMaxVecGrid doubles
  double* comm = nullptr;      // 4 doubles
  CgScalars* S = nullptr;      // device
  const int* setup_fail = nullptr;  // optional device flag: non-zero => CG starts in kCgSetupFailed
  // p.q of the current iteration as n_pq partial sums (written by the operator's own kernels where they have p and q
  // in registers, otherwise by cg_dot_pq_kernel into slot 1); summed in index order by cg_update_kernel
  const double* pq_parts = nullptr;
  int n_pq = 0;
  // sharded CGNR, fused iteration: one more term of p.q — the shard's share, already summed over ranks (it travelled with
  // the camera vector through the operator's all-reduce); nullptr otherwise
  const double* pq_extra = nullptr;
};
constexpr int kMaxPqParts = 2048;

// z = M^-1 r (block-diagonal, or copy when blocks == nullptr) and partial r.z
// slot 0 <- partial |rhs|^2
hipError_t LaunchCgRhsNorm(const CgBuffers& B, hipStream_t stream);
// x = 0, r = rhs, scalars initialised from slot 0 (I/conjugate_gradients_solver.h:130-159 with x0 = 0)
hipError_t LaunchCgInit(const CgBuffers& B, double q_tol, double r_tol, int min_it, int max_it, hipStream_t stream);
// z = M^-1 r (block diagonal over column blocks [first_block, ..), vectors start at scalar
// column col_begin; blocks == nullptr => identity) and slot 0 <- partial r.z
hipError_t LaunchCgPrecondition(const CgBuffers& B, const GenStructure& G, int first_block, int col_begin, int nblocks,
                                int n_local_blocks, const int64_t* diag_off, const double* blocks, hipStream_t stream);
// slot <- partial x.y over the CG vectors' range (used when the preconditioner is an operator)
hipError_t LaunchCgDotSlot(const CgBuffers& B, const double* x, const double* y, int slot, hipStream_t stream);
// Non-zero initial guess (use_spse_initialization): x is kept, r = rhs - tmp, slots 2,3 <- partial -x.(rhs + r), |r|^2
hipError_t LaunchCgInitFromGuess(const CgBuffers& B, const double* tmp, double q_tol, double r_tol, int min_it, int max_it,
                                 hipStream_t stream);
// rho = sum(slot 0); beta; p = z + beta p
hipError_t LaunchCgDirection(const CgBuffers& B, hipStream_t stream);
// slot 1 <- partial p.q  (q lives in z)
hipError_t LaunchCgDotPq(const CgBuffers& B, hipStream_t stream);
// pq = sum(slot 1); alpha; x += alpha p; unless reset: r -= alpha q, slots 2,3 <- partial Q1, |r|^2
hipError_t LaunchCgStep(const CgBuffers& B, int reset, hipStream_t stream);
// r = rhs - tmp; slots 2,3
hipError_t LaunchCgResidualReset(const CgBuffers& B, const double* tmp, hipStream_t stream);
// termination tests in the reference's order (:273-302); iter++
hipError_t LaunchCgFinalize(const CgBuffers& B, hipStream_t stream);
// Fused iteration, second half of iteration `it` (I/conjugate_gradients_solver.h:196-249 + :162-167 of the next one):
// pq = sum(pq_parts), alpha; x += alpha p; unless reset: r -= alpha q, partial Q1 / |r|^2 -> slots 2, 3 and, block by block,
// z = M^-1 r with partial r.z -> slot 0 (the next iteration's rho).  One thread per column block.
// nine_from: blocks [nine_from, nblocks) of the range are all 9 wide (handled nine lanes per block); nblocks if unknown.
// Sharded (B.grid_e > 0): the caller guarantees that blocks [0, nine_from) are exactly the shard; workgroups [0, grid_e) then take
// them and [grid_e, grid) the replicated 9-wide blocks, so that the partial sums split the way total_of() expects.
hipError_t LaunchCgUpdate(const CgBuffers& B, const GenStructure& G, int first_block, int col_begin, int nblocks,
                          const int64_t* diag_off, const double* blocks, int reset, int it, int nine_from, hipStream_t stream);
// Start of a solve with x0 = 0 in two launches: LaunchCgUpdate(..., it = 0) [x = 0, r = rhs, z = M^-1 r, partial |rhs|^2 and r.z] and
// LaunchCgBegin [scalars as LaunchCgInit sets them, p = z] — instead of rhs-norm, init, precondition, direction.
hipError_t LaunchCgBegin(const CgBuffers& B, double q_tol, double r_tol, int min_it, int max_it, hipStream_t stream);
// Fused: the termination tests of iteration `it` in the reference's order (:273-302) and, if CG goes on, the
// direction of iteration it + 1: rho = sum(slot 0), beta = rho / rho_it, p = z + beta p (:167-191).
hipError_t LaunchCgFinalizeDirection(const CgBuffers& B, int it, hipStream_t stream);
// comm[slot] = sum over the shard's workgroups of partials[slot] for slot in [first, first+count)
hipError_t LaunchCgCollapse(const CgBuffers& B, int first_slot, int count, hipStream_t stream);
// ... summed over ranks in the same launch (p2p.h)
hipError_t LaunchCgCollapseExchange(const CgBuffers& B, int first_slot, int count, const P2pComm& comm, hipStream_t stream);

// out[0] = (*flag != 0), out[1] = sum of parts[0 .. n) (one workgroup, fixed order)
hipError_t LaunchCollectScalars(const int* flag, const double* parts, int n, double* out, hipStream_t stream);
// ... summed over ranks in the same launch (p2p.h)
// image != nullptr: also the mailbox read-back of image[0 .. n_image) (out must lie inside it), stamped with `stamp`
hipError_t LaunchCollectScalarsExchange(const int* flag, const double* parts, int n, double* out, const P2pComm& comm, hipStream_t stream,
                                        const double* image = nullptr, int n_image = 0, double* host_dst = nullptr,
                                        unsigned long long* host_stamp = nullptr, unsigned long long stamp = 0);

// ---- one-shot peer-to-peer all-reduce (kernels_cg.hip) ----
// out = sum over ranks of in (n <= cap; in may alias out); comm.epoch counts this communicator's exchanges from 1 (solver.hip: next_exchange).
hipError_t LaunchP2pAllReduce(const double* in, double* out, int64_t n, const P2pComm& comm, int grid_cap, hipStream_t stream);

// host_dst[0 .. n) = src[0 .. n), then *host_stamp = stamp (both in mapped host memory): the solver's read-back "mailbox"
hipError_t LaunchMailbox(const double* src, int n, double* host_dst, unsigned long long* host_stamp, unsigned long long stamp, hipStream_t stream);

// ---- f4: BAL evaluator (kernels_evaluator.hip) ----
struct BalEvalArgs {
  int64_t n_rows = 0;                 // observations (= residual row blocks), grouped by point
  const int32_t* row_cam = nullptr;
  const int32_t* row_pt = nullptr;
  const double2* row_obs = nullptr;   // observed pixel of the row
  const double* state = nullptr;      // [3 n_p | 9 n_c]
  int64_t cam_base = 0;               // 3 n_p
  const double* scale = nullptr;      // Jacobi column scaling or nullptr
  double* residuals = nullptr;        // 2 per row, or nullptr
  double* values = nullptr;           // E cell at 6 r, F cell at 6 n_rows + 18 r
  double* partials = nullptr;         // cost partial per workgroup (<= 2048)
};
hipError_t LaunchBalEvaluate(const BalEvalArgs& A, bool jacobian, int* nparts, hipStream_t stream);
// The same in TILE order for the <2,3,9> fused path: the Jacobian lands in the solver's tiles (J_out, b_out: BalArgs' layout), the F
// cells also at slot_fpos of e.values (nullptr: not), the residuals in e.residuals; e.values' E cells are NOT written.
struct BalEvalTilesArgs {
  BalEvalArgs e;
  int64_t n_tiles = 0;
  const int32_t* slot_bpos = nullptr;   // 2 x row of the slot, -1: padding
  const int32_t* slot_fpos = nullptr;   // where the row's F cell starts in e.values
  const int32_t* slot_cam = nullptr;    // the slot's camera / point (state order) / observed pixel: e.row_* in slot order
  const int32_t* slot_pt = nullptr;
  const double2* slot_obs = nullptr;
  double* pt_pack = nullptr;            // scratch: [state 3 | scale 3] per point, [state 9 | scale 9] per camera (filled by the launch)
  double* cam_pack = nullptr;
  int debug_flags = 0;                  // timing experiments only (ceres_hip_debug_bal_evaluate_tiles_timing): 1 no F copy, 2 no tile J, 4 no b / residuals, 16 plain (not non-temporal) F copy
  double2* J_out = nullptr;
  int64_t tile_pitch = 0;               // double2 elements from one tile to the next
  double2* b_out = nullptr;
};
hipError_t LaunchBalEvaluateTiles(const BalEvalTilesArgs& T, int64_t n_points, int64_t n_cameras, int* nparts, hipStream_t stream);
// delta = step .* scale, cand = x + delta; partials[0..g) = |x|^2, [g..2g) = |delta|^2 partial sums
hipError_t LaunchBalCandidate(const double* x, const double* step, const double* scale, double* delta, double* cand, int64_t n,
                              double* partials, int* nparts, hipStream_t stream);
// partial maxima of |g_i / scale_i|
hipError_t LaunchBalGradientMax(const double* g, const double* scale, int64_t n, double* partials, int* nparts, hipStream_t stream);
hipError_t LaunchBalJacobiScale(const double* colnorm2, double* scale, int64_t n, hipStream_t stream);

}  // namespace chip
#endif
