// solver.hip — the C ABI of include/ceres_hip.h: solver objects, the LinearSolver implementations (CGNR, ITERATIVE_SCHUR implicit
// and explicit, DENSE_SCHUR), the Levenberg-Marquardt step around them, operator-level entry points, communicators and timing.
//
// A solver instance is device resident: the structure (and, for <2,3,9> problems, the
// tile packing plan) is uploaded once; per solve only values/b/D go up and x comes down
// (SURVEY.md §7 "design stance").  Everything runs on one HIP stream owned by the
// instance; the host blocks only to poll the CG status word — with the rest of an LM step already enqueued behind the
// iterations, gated on that word (speculative tail) — and to hand back x.  Sharded instances (one process per GPU) sum their
// camera-space vectors with the one-shot peer-to-peer all-reduce of kernels_cg.hip, RCCL as fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <exception>
#include <new>
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "device.h"

namespace chip {
namespace {

thread_local std::string g_create_error;

// read-back image: 2 kMaxVecGrid partial sums, one double holding the two int flags, one spare, two sums over ranks, then the CG scalars
constexpr int kSumsOffset = 2 * kMaxVecGrid + 2;      // sharded speculative tail: {finite-step flag, model cost} summed over ranks
constexpr int kScalarsOffset = 2 * kMaxVecGrid + 4;
constexpr int kReadbackDoubles = kScalarsOffset + int((sizeof(CgScalars) + sizeof(double) - 1) / sizeof(double));

struct Buf {  // device allocation owned by a solver
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace
}  // namespace chip

using namespace chip;

struct ceres_hip_solver {
  ceres_hip_options opt{};
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev[10] = {};
  int num_cus = 256;
  bool have_structure = false, loaded = false, have_b = false, have_D = false;
  bool packed = false;  // <2,3,9> path: tiles hold the currently loaded values
  bool tiles_only = false;  // the loaded Jacobian exists ONLY as tiles (the device evaluator wrote them: bal_frontend.inc); `values` is not readable
  int path = CERES_HIP_PATH_GENERIC;
  HostStructure hs;
  BalPlan plan;
  const BalOps* ops = nullptr;   // the fused kernels of the plan's shape (camera width, shared strip): kernels_bal.inc
  int32_t* d_slot_hpos[kMaxSharedCellsPerRow] = {};   // shapes with a shared strip: the rows' shared cells (plan.cc)
  int32_t* d_slot_hdesc[kMaxSharedCellsPerRow] = {};
  double* d_strip_parts = nullptr; // [fused_grid][ns (ns + 1) / 2]: the workgroups' shares of the strip's own matrix (kBalShBlocks)
  int64_t* d_cam_foff = nullptr;   // where camera c's block sits in the F-block store (diag_off_f), when not all F blocks are cameras
  std::vector<void*> allocs;
  int64_t device_bytes = 0;

  // structure on device
  GenStructure G;
  // BAL plan on device
  int32_t *d_slot_epos = nullptr, *d_slot_fpos = nullptr, *d_slot_bpos = nullptr, *d_slot_cam = nullptr, *d_tile_pt0 = nullptr;
  uint32_t* d_slot_seg = nullptr;
  int32_t *d_tile_kind = nullptr, *d_tile_aux = nullptr, *d_pt_pos = nullptr, *d_cam_pos = nullptr;
  int32_t *d_cam_ptr = nullptr, *d_cam_fpos = nullptr, *d_cam_slot = nullptr;
  int32_t *d_tile_zbase = nullptr, *d_grp_tile_ptr = nullptr;  // cameras not in LDS: ring rows of the tiles; hybrid groups (plan.cc)
  unsigned int* d_cg_ticket = nullptr;   // CgTail: the S.x pass of a small camera space finishes the CG iteration (device.h)
  bool cg_tail_enabled = [] { const char* e = getenv("CERES_HIP_CG_TAIL"); return !e || atoi(e) != 0; }();   // (A/B switch)
  bool cam_exchange_few = false;
  bool cam_items_few = false;   // no camera has more than a handful of items: bal_invert9_kernel gathers seven cameras per wavefront
  int32_t *d_long_ptr = nullptr, *d_round_ptr = nullptr, *d_seq_ptr = nullptr, *d_round_flag = nullptr;  // long points: where they begin, their rounds (plan.cc)
  uint32_t* d_round_word = nullptr;
  int32_t* d_mo_index = nullptr;                               // M_o record of each slot (camera-major; hybrid plans: the slot's row)
  CamItems cam_items;
  int32_t* d_cam_item_ptr = nullptr;
  double* d_cam_parts = nullptr;   // [items][kCamPart] partial sums of the camera-block pass
  double* d_cam_packed = nullptr;  // [cameras][kCamPart] sharded: the cameras' sums over their items and over ranks (camera_exchange)
  ZUnits zunits;                   // chunked camera-major pass (cameras not in LDS): all chunks' units
  int chunk_grid = 0;              // workgroups of one chunk's tile pass
  // f1: LM step state
  double *lm_diag = nullptr, *lm_D = nullptr, *scalar_partials = nullptr;
  int* d_nonfinite = nullptr;
  bool clear_flags_pending = false;   // the solve's first fused pass still has to clear the two flags (solve_loaded_impl)
  bool fail_flag_clean = false, nonfinite_clean = false;  // cleared together at the start of a solve: the per-operator memsets are skipped once
  bool have_lm_diag = false;
  bool lm_want_model_cost = false;  // op_back_substitute also accumulates the model cost change (fused <2,3,9> path)
  // LM step on the fused path: the last kernel that touches the solution writes the negated step and checks it for finiteness
  // (ITERATIVE_SCHUR: the back-substitution kernel; CGNR: the model-cost kernel, which then also replaces the copy-out of x)
  bool lm_negate_in_solve = false;
  bool lm_negated = false;          // set by whoever did it (or, for CGNR, deferred it: lm_cgnr_copy_pending)
  bool lm_cgnr_copy_pending = false;
  // Speculative tail (one host synchronisation per LM step instead of two): before each poll of the CG status word the solvers
  // enqueue the rest of the step — back-substitution (ITERATIVE_SCHUR) or the model-cost kernel (CGNR), negation and finite
  // check folded in, and the read-back copy of {cost partials, flags} — gated ON THE DEVICE by the status word, so that the
  // poll that sees CG end has the step's results behind it already.  CERES_HIP_SPECULATE=0 switches it off.
  bool speculate = true;            // CERES_HIP_SPECULATE
  bool lm_speculate = false;        // asked for by lm_step_loaded (fused path, one rank)
  bool gate_on_cg_status = false;   // op_back_substitute: launch gated
  bool spec_tail_done = false;      // the gated tail and the copy sit in front of the last poll
  int spec_cgnr_parts = 0;
  int backsub_cost_parts = 0;       // partials it left in scalar_partials
  bool lm_fuse_active = false;      // this step forms D inside the set-up kernels (no separate column-norm pass)
  ceres_hip_lm_options lm_opts{};
  double* d_camsq = nullptr;
  int64_t *d_pt_diag_off = nullptr, *d_cam_diag_off = nullptr;  // into the all-blocks store (CGNR JACOBI)
  int64_t* d_pt_eoff = nullptr;                                 // into the E-block store (ceres_hip_get_ete_inverse)
  double2 *d_J = nullptr, *d_bt = nullptr;
  float4* d_Jf = nullptr;  // fp32 tile storage (options.jacobian_storage == 1)
  double *d_Mo = nullptr, *d_partials = nullptr, *d_global_acc = nullptr;
  int32_t* d_xhot_cam = nullptr;   // BalPlan::xhot_cam (LDS mode: the cameras whose part of x the streaming kernels keep in LDS)
  double* d_zbuf = nullptr;
  // Remainder rows of the fused path (BalPlan::rem_row0: trailing rows without a point cell): their own structure for the generic
  // kernels (compact row space), the residual offset of the first of them, a row-space temporary and their F^T F per camera
  // Sharded <2,3,9> steps sum their per-step camera-space quantities in ONE all-reduce: the buffers sit back to back
  // (ITERATIVE_SCHUR: [preconditioner blocks 81 n_c | rhs 9 n_c | camera column norms 9 n_c]; CGNR: [camera blocks 81 n_c | J^T f,
  // camera part 9 n_c]) and the first producer defers its collective to the last one (solve_loaded sets merge_step_reduce).
  bool merged_layout = false;       // the buffers were allocated back to back (set_structure)
  bool merge_step_reduce = false;   // this solve may defer
  bool rhs_reduce_pending = false;  // rhs holds this rank's raw sums; the all-reduce is still owed
  double* cgnr_rhs_tail = nullptr;  // CGNR: 9 n_c doubles behind the camera blocks of `precond`
  GenStructure GR;
  int rem_rows = 0, rem_b0 = 0;   // rem_b0: where the remainder's residuals begin in b when its rows trail, -1: gathered through d_rem_row_map
  double *rem_tmp = nullptr, *rem_blocks = nullptr, *rem_b = nullptr;
  int32_t* d_rem_row_map = nullptr;  // scalar row of the remainder's compact row space -> scalar row of the whole problem (rows anywhere: CGNR)
  int32_t* d_cam_block = nullptr;
  bool rem_blocks_valid = false;
  int bal_flags = 0;
  bool lds_mode = false;
  int fused_grid = 0;
  // CGNR on INTERNALLY numbered points (the plan renumbered them: fuller tiles, hybrid groups; only where the caller's columns are
  // points-then-cameras back to back, so that block sizes, offsets and the block-diagonal store read the same under the renumbering):
  // the CG vectors, J^T f and the JACOBI point blocks of a solve hold internal point p at [3 p, 3 p + 3); the caller's D is read, and the
  // solution written, through pt_pos.  Operator-level entry points keep the caller's order (BalArgs::pt_pos).
  bool cgnr_internal = false;
  double* D_int = nullptr;        // D's point part in the internal order (written by the set-up kernel, or gathered)
  bool D_int_valid = false;
  bool precond_internal = false;  // s->precond holds the point blocks of a solve (internal order): operator-level readers rebuild it

  // per-solve inputs: either owned copies (host entry points) or caller's device memory
  double *own_values = nullptr, *own_b = nullptr, *own_D = nullptr, *own_x = nullptr;
  const double *values = nullptr, *b = nullptr, *D = nullptr;

  // Schur state
  double* etei = nullptr;         // BAL: 6/point packed; generic: dense e x e blocks (diag_off_e)
  double* rhs_f = nullptr;        // num_cols_f
  double *tmp_rows = nullptr, *tmp_e = nullptr, *tmp_e2 = nullptr;
  // preconditioner blocks: F blocks (diag_off_f) for ITERATIVE_SCHUR, all blocks (diag_off_all) for CGNR
  double* precond = nullptr;
  double* d_S = nullptr;          // explicit Schur complement: dense num_cols_f^2 (DENSE_SCHUR, sharded explicit), or block-sparse values
  bool sparse_S = false;          // use_explicit_schur_complement on one rank: BlockRandomAccessSparseMatrix storage
  // DENSE_SCHUR on one rank: SchurEliminator::Eliminate as the same gather per block into block-sparse storage (d_Sblk), expanded into
  // the dense matrix the Cholesky factors — the thread-per-entry dense elimination took 447 ms for 456 cameras / 500 k observations,
  // against 9 ms of factorisation.  On <2,3,9> problems the fused tile passes do Init / rhs / back-substitution (etei_dense: the
  // point inverses expanded to the dense E-block store the eliminator reads).
  bool dense_from_blocks = false;
  double *d_Sblk = nullptr, *etei_dense = nullptr;
  SchurStorage schur_storage;
  SchurPairs schur_pairs;
  bool precond_valid = false;
  // SCHUR_POWER_SERIES_EXPANSION: blockdiag(F^T F + D_f^2)^-1 and two F-space temporaries
  double *ftf_inv = nullptr, *spse_a = nullptr, *spse_b = nullptr;
  bool ftf_inv_valid = false;
  int* d_fail_flag = nullptr;
  double* d_verdict = nullptr;   // sharded: the factorization verdict summed over ranks (check_factorization)
  // CG
  CgBuffers cg;
  double* cg_rhs = nullptr;
  double* cg_pq_parts = nullptr;   // kMaxPqParts partial sums of p.q left by the operator's kernels
  bool cg_fused = true;            // CERES_HIP_CG_FUSED=0: the five-kernel iteration (A/B measurements)
  int last_cg_iterations = 2;      // of the previous solve: the length of the next solve's first batch of iterations (run_cg)
  int nine_wide_from = 0;          // column blocks [nine_wide_from, ncb) are all 9 wide (BAL: the cameras)
  // ONE pinned buffer, the image of the device's [scalar_partials | flags | CgScalars] (kReadbackDoubles doubles): a poll of the CG
  // status word brings the LM step's partial sums and flags along in the same copy (two copies per poll before: 4.7 us each)
  double* h_pinned = nullptr;
  CgScalars* h_scalars = nullptr;  // = h_pinned + kScalarsOffset
  // The read-back "mailbox": h_pinned is MAPPED host memory; a one-workgroup kernel stores the image there and then a stamp, and the
  // host spins on the stamp (wait_mailbox) — no copy command, no hipStreamSynchronize and its wake-up latency between the device
  // finishing a step and the host knowing it.  CERES_HIP_MAILBOX=0: hipMemcpyAsync + hipStreamSynchronize as before (A/B).
  double* d_h_pinned = nullptr;                 // device view of h_pinned
  unsigned long long* h_stamp = nullptr;        // = h_pinned + kReadbackDoubles (host view), d_stamp its device view
  unsigned long long* d_stamp = nullptr;
  unsigned long long mailbox_seq = 0;
  unsigned long long tail_mailbox_seq = 0;      // != 0: the image of the next poll is already on its way (collect_scalars_exchange_kernel)
  bool final_sync_skippable = [] { const char* e = getenv("CERES_HIP_FINAL_SYNC"); return e && atoi(e) == 0; }();   // (A/B: default keeps the synchronisation)
  bool mailbox = [] { const char* e = getenv("CERES_HIP_MAILBOX"); return !e || atoi(e) != 0; }();
  double* scratch_vec = nullptr;   // num_cols + num_rows doubles for op-level entry points
  // comm: RCCL communicator and / or the one-shot peer-to-peer all-reduce over hipIpc-mapped buffers
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool p2p = false;                 // peers connected
  void* p2p_base = nullptr;         // this rank's buffer: [flags | slots]
  size_t p2p_bytes = 0;
  int64_t p2p_cap = 0;              // doubles per slot
  int p2p_chunks_cap = 0;
  P2pPeers p2p_peers{};
  void* p2p_opened[kP2pMaxWorld] = {};
  unsigned long long p2p_epoch = 0;
  int collectives = 0;   // all-reduces issued since the last LM step began (info.collectives_last_step)
  int* d_comm_error = nullptr;      // raised by a p2p all-reduce whose peer never arrived: device view of h_comm_error
  int* h_comm_error = nullptr;      // mapped pinned host memory
  int* d_comm_error_seen = nullptr; // device memory: set with it, so that later all-reduces do not wait the timeout again
  double p2p_timeout_s = 10.0;
  // Round 6: the sums over ranks happen INSIDE the kernels that produce them (p2p.h: the camera-space reduction of every tile pass, the
  // per-camera blocks, the step's final scalars) instead of in all-reduce launches behind them.  CERES_HIP_P2P_FUSE=0: the stand-alone
  // all-reduce everywhere (A/B).  CERES_HIP_P2P_SHARED_DEVICE=1: the ranks share ONE device (validation runs): kernels that wait for
  // their peers are launched with few workgroups, so that all ranks' waiting workgroups fit on the device beside each other.
  bool spec_agreed = false;           // sharded ITERATIVE_SCHUR: every rank can run the LM step's tail speculatively (agreed likewise)
  bool cam_exchange_agreed = false;   // every rank's camera-major pass exchanges its blocks itself (agreed at the end of set_structure)
  bool rx_agreed = false;             // every rank's camera-space reduction exchanges itself (agreed likewise: reduction_exchanges)
  bool p2p_fences = [] { const char* e = getenv("CERES_HIP_P2P_FENCES"); return e && atoi(e) != 0; }();   // p2p.h: system-scope fences on top (A/B, fall-back)
  bool p2p_fuse = [] { const char* e = getenv("CERES_HIP_P2P_FUSE"); return !e || atoi(e) != 0; }();
  int p2p_grid_cap = [] { const char* e = getenv("CERES_HIP_P2P_SHARED_DEVICE"); return (e && atoi(e) != 0) ? 32 : (1 << 20); }();
  bool p2p_fine_grained = false;   // the receive buffer is a fine-grained allocation (false: the runtime could only export a coarse-grained one)
  // Streamed upload (ceres_hip_values_begin / _ready / _end): while an evaluator is still writing later rows, the rows it has finished go
  // up on a copy stream.  stream_lo / stream_hi[k][r]: the value range row block r owns in value stream k (k = 0: the rows' first cells
  // — or whole rows where rows are contiguous —, k = 1: their other cells); n_value_streams = 0: the layout is not two monotone streams
  // (everything goes up in _end).
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_done = nullptr;
  std::mutex stream_mu;
  bool streaming = false, stream_plan_ready = false;
  int n_value_streams = 0;
  std::vector<int64_t> stream_lo[2], stream_hi[2];
  std::vector<uint8_t> stream_row_sent;
  const double* stream_host_values = nullptr;
  const double* stream_host_b = nullptr;
  int64_t stream_rows_sent = 0, stream_bytes_early = 0, stream_bytes_late = 0;
  ceres_hip_solve_timing timing{};
  // per-phase HIP events of a solve (ceres_hip_get_last_timing): off unless asked for, see rec()
  bool timing_enabled = [] { const char* e = getenv("CERES_HIP_TIMING"); return e && atoi(e) != 0; }();
  std::chrono::steady_clock::time_point call_t0{};
};

namespace {

int fail(ceres_hip_solver* s, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (s) s->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(s, expr)                                                                      \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fail(s, CERES_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr,    \
                                      hipGetErrorString(_e), __FILE__, __LINE__);            \
  } while (0)
#define NCCL_TRY(s, expr)                                                                     \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess) return fail(s, CERES_HIP_E_COMM, "%s failed: %s (%s:%d)", #expr,  \
                                       ncclGetErrorString(_r), __FILE__, __LINE__);          \
  } while (0)
#define TRY(expr)              \
  do {                         \
    int _rc = (expr);          \
    if (_rc != 0) return _rc;  \
  } while (0)

template <typename T>
int dev_alloc(ceres_hip_solver* s, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  void* q = nullptr;
  HIP_TRY(s, hipMalloc(&q, n * sizeof(T)));
  s->allocs.push_back(q);
  s->device_bytes += int64_t(n * sizeof(T));
  *p = static_cast<T*>(q);
  return 0;
}
template <typename T>
int dev_upload(ceres_hip_solver* s, T** p, const std::vector<T>& v) {
  TRY(dev_alloc(s, p, v.size()));
  if (!v.empty()) HIP_TRY(s, hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
  return 0;
}
void free_all(ceres_hip_solver* s) {
  for (void* p : s->allocs) (void)hipFree(p);
  s->allocs.clear();
  s->device_bytes = 0;
}

// The communicator as a kernel argument, for the NEXT epoch: one call per exchange, on every rank in the same order.
P2pComm next_exchange(ceres_hip_solver* s) {
  ++s->collectives;
  P2pComm C;
  C.peers = s->p2p_peers; C.rank = s->rank; C.world = s->world; C.epoch = ++s->p2p_epoch; C.cap = s->p2p_cap; C.chunks_cap = s->p2p_chunks_cap;
  C.fences = s->p2p_fences ? 1 : 0;
  C.error_flag = s->d_comm_error; C.error_seen = s->d_comm_error_seen; C.timeout_ticks = (long long)(s->p2p_timeout_s * 1e8);  // wall_clock64: 100 MHz
  return C;
}
// Sum over ranks, in place, on the solver's stream.  Messages that fit the peer-to-peer slots (everything a solve
// sums: 9 or 81 doubles per camera) go through the one-shot kernel; longer ones are cut into slot-sized pieces, or
// go to RCCL when a communicator exists.
int allreduce(ceres_hip_solver* s, double* dev, size_t n) {
  if (s->world <= 1 || n == 0) return 0;
  if (s->p2p && (int64_t(n) <= s->p2p_cap || !s->comm)) {
    for (size_t off = 0; off < n; off += size_t(s->p2p_cap)) {
      const int64_t len = int64_t(std::min<size_t>(size_t(s->p2p_cap), n - off));
      HIP_TRY(s, LaunchP2pAllReduce(dev + off, dev + off, len, next_exchange(s), s->p2p_grid_cap, s->stream));
    }
    return 0;
  }
  ++s->collectives;
  if (!s->comm) return fail(s, CERES_HIP_E_COMM, "world_size > 1 but no communicator is connected");
  NCCL_TRY(s, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, s->comm, s->stream));
  return 0;
}

// Sharded instance on the fused path whose producers can exchange `slots` doubles in `chunks` wave-sized chunks themselves?
bool exchange_in_producer(const ceres_hip_solver* s, int64_t slots, int64_t chunks) {
  return s->world > 1 && s->p2p && s->p2p_fuse && s->path == CERES_HIP_PATH_BAL && slots <= s->p2p_cap && chunks <= s->p2p_chunks_cap;
}
// ... the camera-space reduction of a tile pass.  The kernel speaks the protocol of the stand-alone all-reduce of the F-space vector, so
// a rank that cannot take it (rows outside its tiles join the raw sums first; cameras scattered over F space) WOULD meet the others in
// the same slots — but the two forms sum the replicated p . q in different groupings, and a rank whose CG scalars differ from its
// peers' in the last bit can leave CG an iteration before them (the peers then wait for an exchange that never comes).  Found by
// tools/fuzz_multirank.py: with camera priors on the last rank the "replicated" camera part was not bit-identical across ranks.
// So all ranks take it or none: reduction_exchanges_local is this rank's view, the ranks agree at the end of set_structure.
bool reduction_exchanges_local(const ceres_hip_solver* s) {
  if (s->path != CERES_HIP_PATH_BAL) return false;
  const int64_t nfv = s->hs.num_cols_f;
  return exchange_in_producer(s, nfv + 1, (nfv + 64) / 64) && s->plan.n_rem_rows == 0 && s->plan.cameras_contiguous;
}
bool reduction_exchanges(const ceres_hip_solver* s) { return s->rx_agreed && s->p2p && reduction_exchanges_local(s); }
// ... the camera-major pass (cam_part packed sums per camera): its exchange has a numbering of its own (a chunk per camera), so ALL
// ranks must take it or none: camera_blocks_exchange_local is this rank's view, the ranks agree at the end of set_structure
bool camera_blocks_exchange_local(const ceres_hip_solver* s) {
  if (s->path != CERES_HIP_PATH_BAL || s->plan.ns != 0) return false;
  const int64_t per = s->ops->cam_part;
  return s->d_cam_packed != nullptr && exchange_in_producer(s, per * s->plan.n_cameras, ((per + 63) / 64) * s->plan.n_cameras);
}
bool camera_blocks_exchange(const ceres_hip_solver* s) { return s->cam_exchange_agreed && s->p2p; }

// Kernels that read the caller-layout value array (the generic operators, also where the fused path borrows them) cannot serve a
// Jacobian that exists only as tiles.
int require_caller_values(ceres_hip_solver* s, const char* what) {
  if (s->tiles_only) return fail(s, CERES_HIP_E_UNSUPPORTED, "%s reads the caller-layout values, which the device evaluator did not write (tiles only)", what);
  return 0;
}

bool is_dense_schur(const ceres_hip_solver* s) { return s->opt.solver_type == CERES_HIP_DENSE_SCHUR; }
// solvers built on the E | F partition (ImplicitSchurComplement / SchurEliminator state, camera-space CG vectors)
bool is_schur(const ceres_hip_solver* s) { return s->opt.solver_type == CERES_HIP_ITERATIVE_SCHUR || is_dense_schur(s); }

// ---------------------------------------------------------------------------
// Operators.  All take device pointers and enqueue on s->stream.
// ---------------------------------------------------------------------------
// Camera space small enough for the S.x pass to finish the CG iteration itself (CgTail, device.h)?  ITERATIVE_SCHUR on the fused path,
// one rank, every camera's accumulator in LDS, no rows outside the tiles, camera blocks back to back, and few enough workgroups for
// one thread of the last one to have all its partial sums in flight at once.
bool cg_tail_possible(const ceres_hip_solver* s) {
  const HostStructure& h = s->hs;
  return s->cg_tail_enabled && s->opt.solver_type == CERES_HIP_ITERATIVE_SCHUR && s->path == CERES_HIP_PATH_BAL && s->world <= 1 &&
         s->ops->has_cg_tail && s->lds_mode && s->plan.n_rem_rows == 0 && s->plan.cameras_contiguous && s->plan.cam_base == 0 &&
         h.num_cols_f == 9 * s->plan.n_cameras &&
         h.num_cols_f > 0 && h.num_cols_f <= kCgTailMax && s->fused_grid <= kCgTailLoads * (512 / std::max(1, h.num_cols_f));
}

BalArgs bal_args(ceres_hip_solver* s) {
  BalArgs A;
  A.J = s->d_J; A.Jf = s->d_Jf; A.b = s->d_bt;
  A.slot_cam = s->d_slot_cam; A.tile_pt0 = s->d_tile_pt0; A.slot_seg = s->d_slot_seg;
  A.tile_kind = s->d_tile_kind; A.tile_aux = s->d_tile_aux;
  A.n_tiles = s->plan.n_tiles; A.n_slots = s->plan.n_tiles * kTile;
  A.pt_pos = s->plan.points_contiguous ? nullptr : s->d_pt_pos;
  A.cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
  A.etei = s->etei;
  A.partials = s->d_partials; A.zbuf = s->d_zbuf;
  A.tile_zbase = s->d_tile_zbase; A.grp_tile_ptr = s->d_grp_tile_ptr;
  A.long_ptr = s->d_long_ptr; A.round_ptr = s->d_round_ptr; A.round_word = s->d_round_word; A.seq_ptr = s->d_seq_ptr; A.round_flag = s->d_round_flag; A.n_seq = int(s->plan.seq_ptr.size()) - 1; A.long_behind = s->plan.long_behind ? 1 : 0;
  A.hyb_rows = s->plan.hybrid ? s->plan.hyb_rows : 0; A.z_flush_row0 = s->plan.z_flush_row0;
  A.mo_index = s->d_mo_index;
  A.n_acc = s->plan.nf * s->plan.n_cameras + s->plan.ns;
  // (the dispatcher drops both for launches of fewer than about a hundred tiles per workgroup: kernels_bal.inc, Fused)
  A.xhot_cam = s->d_xhot_cam; A.n_xhot = s->d_xhot_cam ? int(s->plan.xhot_cam.size()) : 0;
  A.x_lds_scalars = s->lds_mode ? s->plan.nf * s->plan.n_cameras : 0;
  A.cam_base = s->plan.cam_base;
  for (int j = 0; j < kMaxSharedScalars; ++j) A.sh_pos[j] = j < s->plan.ns_used ? s->plan.sh_pos[j] : -1;
  A.have_b = s->have_b ? 1 : 0;
  A.flags = s->bal_flags;
  if (s->lm_fuse_active) {
    A.lm_radius = s->lm_opts.radius; A.lm_min = s->lm_opts.min_diagonal; A.lm_max = s->lm_opts.max_diagonal;
    A.lm_diag_e = s->lm_diag; A.lm_D_e = s->lm_D;
  }
  return A;
}

FMap f_map(const ceres_hip_solver* s, const BalArgs& A) {
  FMap m;
  m.cam_pos = A.cam_pos; m.cam_base = A.cam_base; m.n_cam_scalars = s->plan.nf * s->plan.n_cameras;
  for (int j = 0; j < kMaxSharedScalars; ++j) m.sh_pos[j] = A.sh_pos[j];
  return m;
}
// where the camera blocks sit in a store of all F blocks (nullptr: back to back, nf * nf each — every F block is a camera)
const int64_t* cam_f_offsets(const ceres_hip_solver* s) { return s->plan.ns > 0 ? s->d_cam_foff : nullptr; }

// First pass over a step's Jacobian: let the kernel gather from the caller's layout and write
// the tiles on the way (fused re-layout).  Call right before launching; marks the tiles valid.
// The tiles count as valid only once the gathering kernel has been enqueued successfully: the guard takes `packed` back on every
// error path (a failed launch would otherwise leave later operators reading stale tiles, with no error on a retry).
struct PackGuard {
  ceres_hip_solver* s = nullptr;
  bool armed = false;
  PackGuard() = default;
  PackGuard(ceres_hip_solver* s_, bool a) : s(s_), armed(a) {}
  PackGuard(PackGuard&& o) noexcept : s(o.s), armed(o.armed) { o.armed = false; }
  PackGuard(const PackGuard&) = delete;
  PackGuard& operator=(const PackGuard&) = delete;
  ~PackGuard() { if (armed) s->packed = false; }
  int commit(int rc) { if (rc == 0) armed = false; return rc; }
};
[[nodiscard]] PackGuard use_gather_if_unpacked(ceres_hip_solver* s, BalArgs& A) {
  if (s->packed) return PackGuard();
  A.src_values = s->values;
  A.src_b = s->b;
  A.slot_epos = s->d_slot_epos; A.slot_fpos = s->d_slot_fpos; A.slot_bpos = s->d_slot_bpos;
  for (int q = 0; q < kMaxSharedCellsPerRow; ++q) { A.slot_hpos[q] = s->d_slot_hpos[q]; A.slot_hdesc[q] = s->d_slot_hdesc[q]; }
  A.J_out = s->d_J; A.Jf_out = s->d_Jf; A.b_out = s->d_bt;
  s->packed = true;
  return PackGuard(s, true);
}

int ensure_packed(ceres_hip_solver* s) {
  if (s->path != CERES_HIP_PATH_BAL || s->packed) return 0;
  BalArgs A = bal_args(s);
  PackGuard g = use_gather_if_unpacked(s, A);
  HIP_TRY(s, s->ops->pack(A, s->stream));
  return g.commit(0);
}

LmFuse lm_fuse_for_cameras(ceres_hip_solver* s, bool schur_blocks) {
  LmFuse f;
  if (!s->lm_fuse_active) return f;
  f.radius = s->lm_opts.radius; f.min_d = s->lm_opts.min_diagonal; f.max_d = s->lm_opts.max_diagonal;
  f.camsq = schur_blocks ? s->d_camsq : nullptr;
  f.cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
  f.cam_base = s->plan.cam_base;
  f.diag_f = s->lm_diag + s->hs.num_cols_e;
  f.D_f = s->lm_D + s->hs.num_cols_e;
  return f;
}

// CGNR on internally numbered points: turn the caller's view of the point space (bal_args) into the solve's
void use_internal_points(const ceres_hip_solver* s, BalArgs& A) {
  if (!s->cgnr_internal) return;
  A.d_pos = A.pt_pos;   // D and the LM diagonal stay the caller's
  A.pt_pos = nullptr;
  A.pt_diag_off = nullptr;
}
PointPerm point_perm(const ceres_hip_solver* s) {
  PointPerm p;
  if (s->cgnr_internal) { p.pt_pos = s->d_pt_pos; p.D_e = s->D_int; p.n_e = s->hs.num_cols_e; }
  return p;
}
int ensure_D_int(ceres_hip_solver* s) {
  if (!s->cgnr_internal || !s->D || s->D_int_valid) return 0;
  HIP_TRY(s, LaunchPermutePoints(s->D, s->D_int, s->d_pt_pos, s->hs.num_cols_e, s->hs.num_cols_e, true, s->stream));
  s->D_int_valid = true;
  return 0;
}
// The JACOBI point blocks a CGNR solve left in CG's (internal) point order -> the caller's block order.  LOCAL work (a copy and a
// scatter): the readers that call it are getters — recomputing the preconditioner there would issue an all-reduce on a sharded instance
// (a hang unless every rank called the getter in lockstep) and would use the CURRENT D instead of the D of the solve.
int precond_to_caller_order(ceres_hip_solver* s) {
  if (!s->precond_internal) return 0;
  const size_t n = size_t(9) * s->plan.n_points;
  double* tmp = nullptr;
  HIP_TRY(s, hipMalloc(reinterpret_cast<void**>(&tmp), sizeof(double) * std::max<size_t>(1, n)));
  hipError_t e = hipMemcpyAsync(tmp, s->precond, sizeof(double) * n, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = LaunchScatterBlocks9(tmp, s->precond, s->d_pt_diag_off, s->plan.n_points, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(s, CERES_HIP_E_HIP, "permuting the point blocks failed: %s", hipGetErrorString(e));
  s->precond_internal = false;
  return 0;
}
// the solution of a CGNR solve (CG's order) -> the caller's
int copy_out_cgnr_solution(ceres_hip_solver* s, double* x) {
  const HostStructure& h = s->hs;
  if (s->cgnr_internal) HIP_TRY(s, LaunchPermutePoints(s->cg.x, x, s->d_pt_pos, h.num_cols_e, h.num_cols, false, s->stream));
  else HIP_TRY(s, hipMemcpyAsync(x, s->cg.x, sizeof(double) * h.num_cols, hipMemcpyDeviceToDevice, s->stream));
  return 0;
}

// ---- remainder rows (no point cell) next to the tiles: every contribution is a sum over those rows, added by generic kernels ----
bool has_remainder(const ceres_hip_solver* s) { return s->path == CERES_HIP_PATH_BAL && s->rem_rows > 0; }
// raw F^T F of the remainder rows, per camera (cached until the next load)
int ensure_rem_blocks(ceres_hip_solver* s) {
  if (!has_remainder(s) || s->rem_blocks_valid) return 0;
  HIP_TRY(s, LaunchRemCameraBlocks(s->GR, s->values, s->d_cam_block, s->plan.n_cameras, s->plan.nf, s->rem_blocks, s->stream));
  s->rem_blocks_valid = true;
  return 0;
}
const double* rem_extra_blocks(ceres_hip_solver* s) { return has_remainder(s) ? s->rem_blocks : nullptr; }
// the remainder rows' residuals in their compact row space: a window of b where those rows trail, gathered otherwise (a few thousand
// scalars; gathered at each use so that every way b can arrive — host load, device pointers, the LM step — is covered)
int rem_residuals(ceres_hip_solver* s, const double** out) {
  if (s->rem_b0 >= 0) { *out = s->b + s->rem_b0; return 0; }
  HIP_TRY(s, LaunchGatherRows(s->b, s->d_rem_row_map, s->rem_rows, s->rem_b, s->stream));
  *out = s->rem_b;
  return 0;
}
// y_f += what the remainder rows add to the camera part of a fused operator's output (before any all-reduce: the rows live on one rank)
int add_remainder(ceres_hip_solver* s, int mode, const double* x_f, double* y_f, const int* status) {
  if (!has_remainder(s)) return 0;
  hipStream_t st = s->stream;
  const int32_t* cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
  switch (mode) {
    case kBalSx: case kBalJtJx:   // F_R^T (F_R x)
      HIP_TRY(s, hipMemsetAsync(s->rem_tmp, 0, sizeof(double) * s->rem_rows, st));
      HIP_TRY(s, LaunchGenRightMultiply(s->GR, s->values, kF, x_f, s->rem_tmp, status, st));
      HIP_TRY(s, LaunchRemLeftMultiply(s->GR, s->values, s->d_cam_block, cam_pos, s->plan.cam_base, s->plan.n_cameras, s->plan.nf, s->rem_tmp, y_f, status, st));
      return 0;
    case kBalInit: case kBalJtb: case kBalCgnrInit:   // F_R^T b_R
      if (s->have_b) {
        const double* b_rem = nullptr;
        TRY(rem_residuals(s, &b_rem));
        HIP_TRY(s, LaunchRemLeftMultiply(s->GR, s->values, s->d_cam_block, cam_pos, s->plan.cam_base, s->plan.n_cameras, s->plan.nf, b_rem, y_f, status, st));
      }
      return 0;
    case kBalColNorm:   // diag(F_R^T F_R)
      TRY(ensure_rem_blocks(s));
      HIP_TRY(s, LaunchRemAddDiag(s->rem_blocks, cam_pos, s->plan.cam_base, s->plan.n_cameras, s->plan.nf, y_f, st));
      return 0;
    default: return 0;  // kSpseZ: F^T E (E^T E)^-1 E^T F has no share from rows without an E block
  }
}
// the remainder rows' share of the model cost change into *out (one double); x_f: camera part of the solution (mode 0) / the step (mode 1)
int rem_model_cost(ceres_hip_solver* s, const double* x_f, int mode, double* out) {
  hipStream_t st = s->stream;
  HIP_TRY(s, hipMemsetAsync(s->rem_tmp, 0, sizeof(double) * s->rem_rows, st));
  HIP_TRY(s, LaunchGenRightMultiply(s->GR, s->values, kF, x_f, s->rem_tmp, nullptr, st));
  const double* b_rem = nullptr;
  TRY(rem_residuals(s, &b_rem));
  HIP_TRY(s, LaunchRemModelCost(s->rem_tmp, b_rem, s->rem_rows, mode, out, st));
  return 0;
}

// Run one fused kernel that scatters into camera space and produce y_f.
// add_diag: y_f += D_f^2 x_f (after the all-reduce when sharded).
// pq / n_pq (optional, needs x_f): the kernels that hold x and the finished y in registers also leave the partial sums of
// x . y (CG's p.q) in pq[0 .. *n_pq) — point part from the JtJx tile pass, camera part from the reduction.
// Sharded (pq_extra != nullptr on return): the point-space share of x . y has been summed over ranks into *pq_extra — the element
// right after y_f, which therefore must exist (CG's z vector has the slack) — and pq holds the replicated camera part only.
int bal_scatter(ceres_hip_solver* s, int mode, BalArgs& A, const double* x_f, double* y_f, bool add_diag,
                const int* status, double* pq = nullptr, int* n_pq = nullptr, const double** pq_extra = nullptr,
                bool defer_allreduce = false) {  // sharded: leave this rank's raw sums in y_f, the caller owes the all-reduce
  const int n9 = A.n_acc;                 // accumulator entries (cameras, then the strip)
  const int nfv = s->hs.num_cols_f;       // the F-space vectors themselves
  const FMap map = f_map(s, A);
  const double* D_f = (add_diag && s->D) ? s->D + s->hs.num_cols_e : nullptr;
  A.status = status;
  int n_first = 0;
  // cg_pq_parts holds kMaxPqParts partial sums: the tile pass's (one per workgroup) + the reduction's (<= kMaxVecGrid).  A grid that
  // would not fit leaves p.q to the caller (run_cg then takes one pass over p and q) instead of writing past the buffer.
  if (pq && (s->lds_mode ? s->fused_grid : s->chunk_grid) + kMaxVecGrid > kMaxPqParts) { pq = nullptr; if (n_pq) *n_pq = 0; }
  // sharded with the exchange inside the reduction: that kernel reads the tile pass's partials while its workgroups write the camera
  // part's — the tile pass's go behind them (cg_pq_parts holds kMaxPqParts, checked above)
  double* const pq_tiles = (pq && s->world > 1 && !defer_allreduce && reduction_exchanges(s)) ? pq + kMaxVecGrid : pq;
  if (s->lds_mode) {
    if (pq && mode == kBalJtJx) { A.pq_out = pq_tiles; n_first = s->fused_grid; }
    HIP_TRY(s, s->ops->fused(mode, A, true, s->fused_grid, s->stream));
  } else {
    // cameras do not fit in LDS: the tile pass sums what it can in LDS (hybrid plan) and leaves the other slots' F^T z, and at its
    // end its accumulator rows, in a ring; the camera-major pass adds the ring rows into the camera sums.  (Chunk by chunk when the
    // ring is bounded, CERES_HIP_Z_CHUNK_MIB.)
    const BalPlan& P = s->plan;
    if (pq && mode == kBalJtJx) { A.pq_out = pq_tiles; n_first = s->chunk_grid; }
    HIP_TRY(s, hipMemsetAsync(s->d_global_acc, 0, size_t(n9) * sizeof(double), s->stream));
    if (P.ns > 0) A.strip_sums = s->d_global_acc + size_t(P.nf) * P.n_cameras;
    const int n_chunks = int(P.zc_tile_ptr.size()) - 1;
    for (int k = 0; k < n_chunks; ++k) {
      A.tile_begin = P.zc_tile_ptr[k];
      A.tile_end = P.zc_tile_ptr[k + 1];
      A.pq_accumulate = k > 0 ? 1 : 0;
      HIP_TRY(s, s->ops->fused(mode, A, false, s->chunk_grid, s->stream));
      ZUnits U = s->zunits;
      U.first = P.zc_unit_ptr[k];
      U.count = P.zc_unit_ptr[k + 1] - P.zc_unit_ptr[k];
      HIP_TRY(s, s->ops->camera_chunk(U, s->d_zbuf, s->d_global_acc, status, s->stream));
    }
  }
  const double* parts = s->lds_mode ? s->d_partials : s->d_global_acc;
  const int nparts = s->lds_mode ? s->fused_grid : 1;
  int n_second = 0;
  if (s->world <= 1 && has_remainder(s)) {
    // rows outside the tiles join the raw sums first; D_f^2 x_f and the camera part of x . y follow on the COMPLETE output
    HIP_TRY(s, s->ops->reduce_partials(parts, nparts, n9, map, nullptr, nullptr, y_f, status, nullptr, nullptr, s->stream, nullptr, 0, nullptr));
    TRY(add_remainder(s, mode, x_f, y_f, status));
    HIP_TRY(s, s->ops->add_f_diagonal(n9, map, D_f, x_f, y_f, status, pq ? pq + n_first : nullptr, &n_second, s->stream));
  } else if (s->world <= 1) {
    HIP_TRY(s, s->ops->reduce_partials(parts, nparts, n9, map, D_f, x_f, y_f, status, pq ? pq + n_first : nullptr, &n_second, s->stream, nullptr, 0, nullptr));
  } else if (!defer_allreduce && reduction_exchanges(s)) {
    // the reduction sums over ranks itself (p2p.h), adds D_f^2 x_f and leaves the camera part of x . y: one launch
    const bool pack = pq && pq_extra && n_first > 0;  // CGNR's p.q: the shard's share rides along as slot n9 of the exchange
    HIP_TRY(s, s->ops->reduce_exchange(parts, nparts, n9, nfv, map, D_f, x_f, y_f, status, pq, &n_second, s->stream,
                                       pack ? pq + kMaxVecGrid : nullptr, n_first, pack ? y_f + nfv : nullptr, next_exchange(s), s->p2p_grid_cap));
    if (pack) *pq_extra = y_f + nfv;
    else if (n_first > 0) { pq = nullptr; n_second = 0; }   // the shard's point part without the packing: no complete p.q from here
    n_first = 0;   // (the tile pass's partials — behind the reduction's in pq, see below — are consumed: the camera part sits at pq[0 ..))
  } else {
    const bool pack = pq && pq_extra && n_first > 0;  // CGNR's p.q: the shard's share rides along as element n9 of the all-reduce
    HIP_TRY(s, s->ops->reduce_partials(parts, nparts, n9, map, nullptr, nullptr, y_f, status, nullptr, nullptr, s->stream,
                                       pack ? pq : nullptr, n_first, pack ? y_f + nfv : nullptr));
    TRY(add_remainder(s, mode, x_f, y_f, status));  // this rank's rows without a point cell (partition.py hands them to one rank)
    // camera scalars are contiguous in the Schur-ordered (sharded) layout
    if (!defer_allreduce) TRY(allreduce(s, y_f, size_t(nfv) + (pack ? 1 : 0)));
    if (pack) { *pq_extra = y_f + nfv; n_first = 0; }  // the tile pass's partials are consumed; the camera part goes to pq[0 ..)
    else if (n_first > 0) { pq = nullptr; n_first = 0; }  // the shard's point part without the packing: no complete p.q from here
    // (S.x has no point part: its p.q is the replicated camera part, formed below as on one rank)
    HIP_TRY(s, s->ops->add_f_diagonal(n9, map, D_f, x_f, y_f, status, pq, &n_second, s->stream));
  }
  if (n_pq) *n_pq = n_first + n_second;
  return 0;
}

// y = S x on F-space vectors.  ImplicitSchurComplement::RightMultiplyAndAccumulate.
int op_sx(ceres_hip_solver* s, const double* x, double* y, const int* status, double* pq = nullptr, int* n_pq = nullptr) {
  const HostStructure& h = s->hs;
  if (n_pq) *n_pq = 0;
  if (s->path == CERES_HIP_PATH_BAL) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.x_f = x;
    return bal_scatter(s, kBalSx, A, x, y, true, status, pq, n_pq);
  }
  const double* v = s->values;
  hipStream_t st = s->stream;
  // the row-space half in one launch where the block sizes allow: z = F x - E (E^T E)^-1 E^T F x, a group of lanes per chunk
  const int first_free_row = h.num_row_blocks_e < h.nrb ? h.rpos[h.num_row_blocks_e] : h.num_rows;   // rows without an E block: z = F x
  const hipError_t fe = LaunchGenChunkSx(s->G, v, s->etei, x, s->tmp_rows, status, st);
  hipError_t pe = hipSuccess;
  if (fe == hipSuccess) {
    if (first_free_row < h.num_rows) {
      HIP_TRY(s, hipMemsetAsync(s->tmp_rows + first_free_row, 0, sizeof(double) * (h.num_rows - first_free_row), st));
      HIP_TRY(s, LaunchGenRightMultiplyFrom(s->G, v, kF, first_free_row, x, s->tmp_rows, status, st));
    }
  } else if (fe != hipErrorNotSupported) {
    HIP_TRY(s, fe);
  } else {
    HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
    HIP_TRY(s, LaunchGenRightMultiply(s->G, v, kF, x, s->tmp_rows, status, st));
    // t -= E (E^T E)^-1 E^T t, chunk by chunk in one launch; block sizes beyond the grouped kernel's: the reference's three passes
    pe = LaunchGenChunkProject(s->G, v, s->etei, s->tmp_rows, 1, nullptr, status, st);
  }
  if (pe == hipErrorNotSupported) {
    HIP_TRY(s, hipMemsetAsync(s->tmp_e, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, v, kE, s->tmp_rows, s->tmp_e, status, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_e2, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, 0, h.nelim, s->G.diag_off_e, s->etei, s->tmp_e, s->tmp_e2, status, st));
    HIP_TRY(s, LaunchAxpby(-1.0, s->tmp_e2, 0.0, s->tmp_e2, s->tmp_e2, h.num_cols_e, st));
    HIP_TRY(s, LaunchGenRightMultiply(s->G, v, kE, s->tmp_e2, s->tmp_rows, status, st));
  } else {
    HIP_TRY(s, pe);
  }
  const double* D_f = s->D ? s->D + h.num_cols_e : nullptr;
  if (s->world <= 1) {
    HIP_TRY(s, LaunchSquareScale(D_f, x, y, h.num_cols_f, status, st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, v, kF, s->tmp_rows, y, status, st));
  } else {
    HIP_TRY(s, LaunchSquareScale(nullptr, x, y, h.num_cols_f, status, st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, v, kF, s->tmp_rows, y, status, st));
    TRY(allreduce(s, y, size_t(h.num_cols_f)));
    HIP_TRY(s, LaunchAddSquareScale(D_f, x, y, h.num_cols_f, status, st));
  }
  return 0;
}

// y = (A^T A + D^2) x on full-space vectors.  CgnrLinearOperator (y zeroed by CG first).
// internal: x and y are a solve's CG vectors (internally numbered points where the plan renumbered them, D_int must be valid)
int op_jtjx(ceres_hip_solver* s, const double* x, double* y, const int* status, double* pq = nullptr, int* n_pq = nullptr,
            const double** pq_extra = nullptr, bool internal = false) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  if (n_pq) *n_pq = 0;
  if (pq_extra) *pq_extra = nullptr;
  if (s->path == CERES_HIP_PATH_BAL) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.x_e = x; A.x_f = x + h.num_cols_e; A.y_e = y; A.D_e = s->D;
    if (internal && s->cgnr_internal) { use_internal_points(s, A); A.d_pos = nullptr; if (s->D) A.D_e = s->D_int; }
    return bal_scatter(s, kBalJtJx, A, x + h.num_cols_e, y + h.num_cols_e, true, status, pq, n_pq, pq_extra);
  }
  HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
  HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kAll, x, s->tmp_rows, status, st));
  HIP_TRY(s, LaunchSquareScale(nullptr, x, y, h.num_cols, status, st));
  HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kAll, s->tmp_rows, y, status, st));
  if (s->world > 1) TRY(allreduce(s, y + h.num_cols_e, size_t(h.num_cols_f)));
  HIP_TRY(s, LaunchAddSquareScale(s->D, x, y, h.num_cols, status, st));
  return 0;
}

// y = A^T b
int op_jtb(ceres_hip_solver* s, double* y) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  if (!s->have_b) return fail(s, CERES_HIP_E_INVALID, "no residual vector loaded");
  if (s->path == CERES_HIP_PATH_BAL) {
    BalArgs A = bal_args(s);
    A.y_e = y;
    PackGuard g = use_gather_if_unpacked(s, A);  // first pass over freshly loaded values (the evaluator's gradient): fused with the re-layout
    return g.commit(bal_scatter(s, kBalJtb, A, nullptr, y + h.num_cols_e, false, nullptr));
  }
  HIP_TRY(s, hipMemsetAsync(y, 0, sizeof(double) * h.num_cols, st));
  HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kAll, s->b, y, nullptr, st));
  if (s->world > 1) TRY(allreduce(s, y + h.num_cols_e, size_t(h.num_cols_f)));
  return 0;
}

// ImplicitSchurComplement::Init: (E^T E + D_e^2)^-1 blocks and rhs.
int op_schur_init(ceres_hip_solver* s, bool want_Mo) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  if (s->path == CERES_HIP_PATH_BAL) {
    BalArgs A = bal_args(s);
    A.D_e = s->D;
    A.Mo = want_Mo ? s->d_Mo : nullptr;
    if (s->clear_flags_pending) { A.clear_flags = s->d_nonfinite; s->clear_flags_pending = false; }
    PackGuard g = use_gather_if_unpacked(s, A);  // the step's first pass over J also writes the tiles
    if (s->have_b) {
      const bool defer = s->world > 1 && s->merge_step_reduce && s->merged_layout;
      TRY(g.commit(bal_scatter(s, kBalInit, A, nullptr, s->rhs_f, false, nullptr, nullptr, nullptr, nullptr, defer)));
      s->rhs_reduce_pending = defer;
      return 0;
    }
    // no residuals: only the inverses (and M_o) are needed; nothing is scattered
    // (cameras not in LDS: the scattering modes walk the hybrid groups, one workgroup each — chunk_grid, never fused_grid)
    HIP_TRY(s, s->ops->fused(kBalInit, A, s->lds_mode, s->lds_mode ? s->fused_grid : s->chunk_grid, st));
    return g.commit(0);
  }
  // block diagonal of E^T E + D_e^2, inverted in place
  HIP_TRY(s, LaunchGenBlockDiagonal(s->G, s->values, kE, s->D, s->etei, h.diag_off_e.back(), st));
  HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
  HIP_TRY(s, LaunchGenInvertBlocks(s->G, 0, h.nelim, s->G.diag_off_e, s->etei, s->d_fail_flag, st));
  if (!s->have_b) return 0;
  // rhs = F^T (b - E (E^T E)^-1 E^T b)           UpdateRhs
  HIP_TRY(s, hipMemcpyAsync(s->tmp_rows, s->b, sizeof(double) * h.num_rows, hipMemcpyDeviceToDevice, st));
  const hipError_t pe = LaunchGenChunkProject(s->G, s->values, s->etei, s->tmp_rows, 1, nullptr, nullptr, st);
  if (pe == hipErrorNotSupported) {
    HIP_TRY(s, hipMemsetAsync(s->tmp_e, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kE, s->b, s->tmp_e, nullptr, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_e2, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, 0, h.nelim, s->G.diag_off_e, s->etei, s->tmp_e, s->tmp_e2, nullptr, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
    HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kE, s->tmp_e2, s->tmp_rows, nullptr, st));
    HIP_TRY(s, LaunchAxpby(1.0, s->b, -1.0, s->tmp_rows, s->tmp_rows, h.num_rows, st));
  } else {
    HIP_TRY(s, pe);
  }
  HIP_TRY(s, hipMemsetAsync(s->rhs_f, 0, sizeof(double) * std::max(1, h.num_cols_f), st));
  HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kF, s->tmp_rows, s->rhs_f, nullptr, st));
  TRY(allreduce(s, s->rhs_f, size_t(h.num_cols_f)));
  return 0;
}

// ImplicitSchurComplement::BackSubstitute: z (F space, may be nullptr if there are no F
// blocks) -> x (full space).
int op_back_substitute(ceres_hip_solver* s, const double* z, double* x) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  if (!s->have_b) return fail(s, CERES_HIP_E_INVALID, "no residual vector loaded");
  if (s->path == CERES_HIP_PATH_BAL) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.x_f = z; A.y_e = x;
    // f1: the LM step wants the model cost change of -x; the kernel has J x in hand (one partial per workgroup)
    s->backsub_cost_parts = 0;
    if (s->lm_want_model_cost && z != nullptr) { A.scalar_out = s->scalar_partials; s->backsub_cost_parts = s->fused_grid; }
    if (h.num_cols_f > 0 && z != nullptr) { A.copy_src = z; A.copy_dst = x + h.num_cols_e; A.copy_n = h.num_cols_f; }
    if (s->gate_on_cg_status) A.run_after_cg = &s->cg.S->status;
    if (s->lm_negate_in_solve && z != nullptr) {
      if (!s->nonfinite_clean) HIP_TRY(s, hipMemsetAsync(s->d_nonfinite, 0, sizeof(int), st));
      s->nonfinite_clean = false;
      A.negate_out = 1; A.nonfinite = s->d_nonfinite;
      s->lm_negated = true;
    }
    HIP_TRY(s, s->ops->fused(kBalBackSub, A, false, s->fused_grid, st));
    if (s->backsub_cost_parts > 0 && has_remainder(s)) {  // the rows outside the tiles: one more partial sum behind the workgroups'
      TRY(rem_model_cost(s, z, 0, s->scalar_partials + s->backsub_cost_parts));
      ++s->backsub_cost_parts;
    }
    return 0;
  } else {
    HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
    if (h.num_cols_f > 0) HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kF, z, s->tmp_rows, nullptr, st));
    HIP_TRY(s, LaunchAxpby(1.0, s->b, -1.0, s->tmp_rows, s->tmp_rows, h.num_rows, st));
    HIP_TRY(s, hipMemsetAsync(x, 0, sizeof(double) * h.num_cols, st));
    const hipError_t pe = LaunchGenChunkProject(s->G, s->values, s->etei, s->tmp_rows, 0, x, nullptr, st);   // x_e = (E^T E)^-1 E^T t, chunk by chunk
    if (pe == hipErrorNotSupported) {
      HIP_TRY(s, hipMemsetAsync(s->tmp_e, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
      HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kE, s->tmp_rows, s->tmp_e, nullptr, st));
      HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, 0, h.nelim, s->G.diag_off_e, s->etei, s->tmp_e, x, nullptr, st));
    } else {
      HIP_TRY(s, pe);
    }
  }
  if (h.num_cols_f > 0)
    HIP_TRY(s, hipMemcpyAsync(x + h.num_cols_e, z, sizeof(double) * h.num_cols_f, hipMemcpyDeviceToDevice, st));
  return 0;
}

// The diagonal blocks of the SHARED column blocks (shapes with a strip) into the F-block store `out`: the strip's NS x NS matrix
// H^T H (JACOBI) or H^T H - sum over points G^T (E^T E)^-1 G, G = sum over the point's rows of E^T H (SCHUR_JACOBI: the rows of a point
// all hold the shared cell, so the block is not a sum of per-observation terms like a camera's), from one tile pass (kBalShBlocks);
// + D^2, inverted in place like every other block.
// with_D / invert: a sharded run calls once for this rank's raw sums (the caller all-reduces the block store, adds D^2 to every F block)
// and then invert_shared_blocks; one rank does it all in one call
int invert_shared_blocks(ceres_hip_solver* s, double* out) {
  const BalPlan& P = s->plan;
  const HostStructure& h = s->hs;
  if (s->path != CERES_HIP_PATH_BAL || P.ns == 0) return 0;
  for (size_t q = 0; q < P.sh_block.size(); ++q) {   // (the generic kernel addresses blocks relative to the first of its range)
    const int j = P.sh_block[q];
    const int64_t off = is_schur(s) ? h.diag_off_f[j - h.nelim] : h.diag_off_all[j];
    HIP_TRY(s, LaunchGenInvertBlocks(s->G, j, 1, is_schur(s) ? s->G.diag_off_f + (j - h.nelim) : s->G.diag_off_all + j, out + off, s->d_fail_flag, s->stream));
  }
  return 0;
}
int shared_preconditioner_blocks(ceres_hip_solver* s, bool schur, double* out, bool invert, bool with_D = true) {
  const BalPlan& P = s->plan;
  if (s->path != CERES_HIP_PATH_BAL || P.ns == 0) return 0;
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  TRY(ensure_packed(s));
  BalArgs A = bal_args(s);
  if (!schur) A.etei = nullptr;   // JACOBI: blockdiag(F^T F) — the H^T H part alone
  A.scalar_out = s->d_strip_parts;
  HIP_TRY(s, s->ops->fused(kBalShBlocks, A, false, s->fused_grid, st));
  StripBlocks sb;
  sb.count = int(P.sh_block.size());
  for (int q = 0; q < sb.count; ++q) {
    const int j = P.sh_block[q];
    sb.off[q] = P.sh_off[q]; sb.width[q] = h.csz[j]; sb.pos[q] = h.cpos[j] - h.num_cols_e;
    sb.out[q] = is_schur(s) ? h.diag_off_f[j - h.nelim] : h.diag_off_all[j];
  }
  const double* D_f = (with_D && s->D) ? s->D + h.num_cols_e : nullptr;
  HIP_TRY(s, s->ops->strip_finish(s->d_strip_parts, s->fused_grid, sb, D_f, out, st));
  if (invert) TRY(invert_shared_blocks(s, out));
  return 0;
}

// Preconditioner blocks into `out`; invert = false leaves them as assembled.
//   ITERATIVE_SCHUR: SCHUR_JACOBI (diag blocks of S) or JACOBI (blockdiag(F^T F + D_f^2)), F blocks
//   CGNR:            JACOBI, all column blocks
int op_preconditioner(ceres_hip_solver* s, int type, double* out, bool invert) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  const double* D_f = s->D ? s->D + h.num_cols_e : nullptr;
  if (!s->fail_flag_clean) HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
  s->fail_flag_clean = false;
  if (is_schur(s)) {
    const int nf = h.ncb - h.nelim;
    const int64_t len = h.diag_off_f.back();
    if (s->path == CERES_HIP_PATH_BAL) {
      const bool schur = type == CERES_HIP_SCHUR_JACOBI;
      const bool fuse = s->lm_fuse_active && invert;
      HIP_TRY(s, s->ops->camera_items(schur, s->values, s->cam_items, s->d_cam_fpos, s->d_cam_slot, s->d_Mo, s->d_cam_parts, st));
      TRY(ensure_rem_blocks(s));  // rows without a point cell add their F^T F to the diagonal cells (NoEBlockRowsUpdate) and to the column norms
      if (s->world > 1 && invert && camera_blocks_exchange(s) && !s->rhs_reduce_pending) {
        // sharded: the inversion kernel adds up this rank's items, sums the packed blocks (and column norms) over ranks (p2p.h), adds
        // D_f^2 — or forms the fused LM diagonal from the summed norms — and inverts: one launch, one exchange
        HIP_TRY(s, s->ops->camera_exchange(s->d_cam_parts, s->d_cam_item_ptr, s->plan.n_cameras, rem_extra_blocks(s), s->d_cam_packed, next_exchange(s),
                                           s->p2p_grid_cap, st, s->cam_exchange_few));
        CamGather g;
        g.packed = s->d_cam_packed; g.want_sq = (fuse && schur) ? 1 : 0;
        g.D_f = fuse ? nullptr : D_f;
        g.cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
        g.cam_base = s->plan.cam_base;
        HIP_TRY(s, s->ops->invert(out, cam_f_offsets(s), s->plan.n_cameras, s->d_fail_flag, fuse ? lm_fuse_for_cameras(s, false) : LmFuse(), g, st));
      } else if (s->world <= 1 && invert) {
        // the inversion kernel adds up a camera's items itself (+ D_f^2, or the fused LM diagonal it forms from them)
        CamGather g;
        g.parts = s->d_cam_parts; g.cam_item_ptr = s->d_cam_item_ptr; g.few = s->cam_items_few ? 1 : 0; g.want_sq = (fuse && schur) ? 1 : 0;
        g.extra = rem_extra_blocks(s);
        g.D_f = fuse ? nullptr : D_f;
        g.cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
        g.cam_base = s->plan.cam_base;
        HIP_TRY(s, s->ops->invert(out, cam_f_offsets(s), s->plan.n_cameras, s->d_fail_flag, fuse ? lm_fuse_for_cameras(s, false) : LmFuse(), g, st));
        TRY(shared_preconditioner_blocks(s, schur, out, true));
      } else {
        // raw sums in memory first (no diagonal when sharded: the ranks' sums are added up, then D_f^2 joins once)
        HIP_TRY(s, s->ops->camera_finish(s->d_cam_parts, s->d_cam_item_ptr, (s->world > 1 || fuse) ? nullptr : D_f,
                                         s->plan.cameras_contiguous ? nullptr : s->d_cam_pos, s->plan.cam_base, cam_f_offsets(s), out,
                                         (fuse && schur) ? s->d_camsq : nullptr, s->plan.n_cameras, st, rem_extra_blocks(s), s->cam_items_few));
        if (s->world > 1) TRY(shared_preconditioner_blocks(s, schur, out, false, false));   // this rank's raw sums join the all-reduce of the block store
        if (s->world > 1) {
          const size_t n9c = size_t(h.num_cols_f);   // rhs / column norms: every camera-side scalar (the cameras' and a shared strip's)
          if (s->rhs_reduce_pending && out == s->precond && s->merged_layout) {
            // ONE collective for the step's camera-space sums: [blocks | rhs | column norms] are contiguous (set_structure)
            TRY(allreduce(s, out, size_t(len) + n9c + ((fuse && schur) ? n9c : 0)));
            s->rhs_reduce_pending = false;
          } else {
            TRY(allreduce(s, out, size_t(len)));
            if (fuse && schur) TRY(allreduce(s, s->d_camsq, n9c));  // column norms of the camera columns
          }
          // fused LM diagonal: D_f does not exist yet, bal_invert9_kernel forms it from the reduced sums and adds it
          if (D_f && !fuse) HIP_TRY(s, LaunchAddBlockDiagonalSquares(s->G, h.nelim, nf, s->G.diag_off_f, s->D, out, st));
        }
        if (invert) HIP_TRY(s, s->ops->invert(out, cam_f_offsets(s), s->plan.n_cameras, s->d_fail_flag, fuse ? lm_fuse_for_cameras(s, schur) : LmFuse(), CamGather(), st));
        if (s->world > 1) { if (invert) TRY(invert_shared_blocks(s, out)); }
        else TRY(shared_preconditioner_blocks(s, schur, out, invert));
      }
    } else {
      if (type == CERES_HIP_SCHUR_JACOBI) {
        HIP_TRY(s, LaunchGenSchurJacobi(s->G, s->values, s->etei, s->D, s->world > 1 ? 0 : 1, out, len, st));
      } else {
        HIP_TRY(s, LaunchGenBlockDiagonal(s->G, s->values, kF, s->world > 1 ? nullptr : s->D, out, len, st));
      }
      if (s->world > 1) {
        TRY(allreduce(s, out, size_t(len)));
        if (D_f) HIP_TRY(s, LaunchAddBlockDiagonalSquares(s->G, h.nelim, nf, s->G.diag_off_f, s->D, out, st));
      }
      if (invert) HIP_TRY(s, LaunchGenInvertBlocks(s->G, h.nelim, nf, s->G.diag_off_f, out, s->d_fail_flag, st));
    }
    return 0;
  }
  // CGNR JACOBI (the caller's block order)
  const int64_t len = h.diag_off_all.back();
  if (out == s->precond) s->precond_internal = false;
  if (s->path == CERES_HIP_PATH_BAL && s->world <= 1 && invert) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.etei = nullptr;
    A.D_e = s->D;
    A.point_blocks = out;
    A.pt_diag_off = s->d_pt_diag_off;
    HIP_TRY(s, s->ops->fused(kBalEte, A, false, s->fused_grid, st));
    HIP_TRY(s, s->ops->camera_items(false, s->values, s->cam_items, s->d_cam_fpos, s->d_cam_slot, nullptr, s->d_cam_parts, st));
    TRY(ensure_rem_blocks(s));
    CamGather g;
    g.parts = s->d_cam_parts; g.cam_item_ptr = s->d_cam_item_ptr; g.few = s->cam_items_few ? 1 : 0; g.D_f = D_f;
    g.extra = rem_extra_blocks(s);
    g.cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
    g.cam_base = s->plan.cam_base;
    HIP_TRY(s, s->ops->invert(out, s->d_cam_diag_off, s->plan.n_cameras, s->d_fail_flag, LmFuse(), g, st));
    TRY(shared_preconditioner_blocks(s, false, out, true));
    return 0;
  }
  TRY(require_caller_values(s, "the uninverted JACOBI blocks"));
  HIP_TRY(s, LaunchGenBlockDiagonal(s->G, s->values, kAll, s->world > 1 ? nullptr : s->D, out, len, st));
  if (s->world > 1) {
    // shared (camera) blocks follow the local (point) blocks in the sharded layout
    const int64_t first = h.diag_off_all[h.nelim];
    TRY(allreduce(s, out + first, size_t(len - first)));
    if (s->D) HIP_TRY(s, LaunchAddBlockDiagonalSquares(s->G, 0, h.ncb, s->G.diag_off_all, s->D, out, st));
  }
  if (invert) HIP_TRY(s, LaunchGenInvertBlocks(s->G, 0, h.ncb, s->G.diag_off_all, out, s->d_fail_flag, st));
  return 0;
}

// CGNR set-up on the <2,3,9> path in ONE pass over J: rhs = J^T b and (JACOBI) the inverted
// point blocks; the camera blocks follow from the camera-major pass.  Replaces
// BlockSparseJacobiPreconditioner::UpdateImpl + A->LeftMultiplyAndAccumulate(b)
// (I/cgnr_solver.cc:152-191) = two passes over J, and absorbs the re-layout pass.
int op_cgnr_setup_bal(ceres_hip_solver* s, bool jacobi, double* rhs, double* blocks) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  const double* D_f = s->D ? s->D + h.num_cols_e : nullptr;
  const int64_t len = h.diag_off_all.back();
  if (!s->fail_flag_clean) HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
  s->fail_flag_clean = false;
  BalArgs A = bal_args(s);
  if (s->clear_flags_pending) { A.clear_flags = s->d_nonfinite; s->clear_flags_pending = false; }
  A.etei = nullptr;
  A.D_e = s->D;
  A.y_e = rhs;
  A.point_blocks = jacobi ? blocks : nullptr;
  A.pt_diag_off = s->d_pt_diag_off;
  if (s->cgnr_internal) {  // rhs and the point blocks in CG's order; D read at the caller's offsets and left in CG's order on the way
    use_internal_points(s, A);
    A.D_int_out = s->D_int;
    s->D_int_valid = jacobi && (s->D != nullptr || s->lm_fuse_active);
    if (jacobi && blocks == s->precond) s->precond_internal = true;
  }
  // sharded with JACOBI: the camera part of J^T f is left raw behind the camera blocks and rides in their all-reduce
  const bool merge = s->world > 1 && jacobi && s->merge_step_reduce && s->merged_layout && blocks == s->precond && s->cgnr_rhs_tail;
  {
    PackGuard g = use_gather_if_unpacked(s, A);
    TRY(g.commit(bal_scatter(s, kBalCgnrInit, A, nullptr, merge ? s->cgnr_rhs_tail : rhs + h.num_cols_e, false, nullptr, nullptr, nullptr, nullptr, merge)));
  }
  if (!jacobi) return 0;
  const int32_t* cam_pos = s->plan.cameras_contiguous ? nullptr : s->d_cam_pos;
  HIP_TRY(s, s->ops->camera_items(false, s->values, s->cam_items, s->d_cam_fpos, s->d_cam_slot, nullptr, s->d_cam_parts, st));
  TRY(ensure_rem_blocks(s));
  if (s->world <= 1) {
    CamGather g;
    g.parts = s->d_cam_parts; g.cam_item_ptr = s->d_cam_item_ptr; g.few = s->cam_items_few ? 1 : 0; g.cam_pos = cam_pos;
    g.cam_base = s->plan.cam_base;
    g.extra = rem_extra_blocks(s);
    g.D_f = s->lm_fuse_active ? nullptr : D_f;  // fused LM diagonal: bal_invert9_kernel forms D_f from the block's own diagonal and adds it
    HIP_TRY(s, s->ops->invert(blocks, s->d_cam_diag_off, s->plan.n_cameras, s->d_fail_flag, lm_fuse_for_cameras(s, false), g, st));
    TRY(shared_preconditioner_blocks(s, false, blocks, true));
    return 0;
  }
  if (!merge && camera_blocks_exchange(s)) {   // sharded, the sums over ranks inside the inversion kernel (see op_preconditioner)
    HIP_TRY(s, s->ops->camera_exchange(s->d_cam_parts, s->d_cam_item_ptr, s->plan.n_cameras, rem_extra_blocks(s), s->d_cam_packed, next_exchange(s),
                                       s->p2p_grid_cap, st, s->cam_exchange_few));
    CamGather g;
    g.packed = s->d_cam_packed; g.cam_pos = cam_pos;
    g.cam_base = s->plan.cam_base;
    g.D_f = s->lm_fuse_active ? nullptr : D_f;
    HIP_TRY(s, s->ops->invert(blocks, s->d_cam_diag_off, s->plan.n_cameras, s->d_fail_flag, lm_fuse_for_cameras(s, false), g, st));
    return 0;
  }
  // sharded: raw F^T F sums, all-reduce, then the diagonal (camera blocks are contiguous
  // behind the point blocks in the Schur-ordered layout a sharded run requires)
  HIP_TRY(s, s->ops->camera_finish(s->d_cam_parts, s->d_cam_item_ptr, nullptr, cam_pos, s->plan.cam_base, s->d_cam_diag_off, blocks, nullptr,
                                   s->plan.n_cameras, st, rem_extra_blocks(s), s->cam_items_few));
  TRY(shared_preconditioner_blocks(s, false, blocks, false, false));   // this rank's raw H^T H joins the all-reduce
  const int64_t first = h.diag_off_all[h.nelim];
  if (merge) {
    TRY(allreduce(s, blocks + first, size_t(len - first) + size_t(h.num_cols_f)));
    HIP_TRY(s, hipMemcpyAsync(rhs + h.num_cols_e, s->cgnr_rhs_tail, sizeof(double) * h.num_cols_f, hipMemcpyDeviceToDevice, st));
  } else {
    TRY(allreduce(s, blocks + first, size_t(len - first)));
  }
  if (s->D && !s->lm_fuse_active)  // fused LM diagonal: bal_invert9_kernel forms D_f from the reduced diagonal and adds it
    HIP_TRY(s, LaunchAddBlockDiagonalSquares(s->G, h.nelim, h.ncb - h.nelim, s->G.diag_off_all + h.nelim, s->D, blocks + first, st));
  HIP_TRY(s, s->ops->invert(blocks, s->d_cam_diag_off, s->plan.n_cameras, s->d_fail_flag, lm_fuse_for_cameras(s, false), CamGather(), st));
  TRY(invert_shared_blocks(s, blocks));
  return 0;
}

// ---- SCHUR_POWER_SERIES_EXPANSION ---------------------------------------
int ensure_ftf_inverse(ceres_hip_solver* s) {
  if (s->ftf_inv_valid) return 0;
  TRY(op_preconditioner(s, CERES_HIP_JACOBI, s->ftf_inv, true));
  s->ftf_inv_valid = true;
  return 0;
}

// y += (F^T F)^-1 F^T E (E^T E)^-1 E^T F x.  InversePowerSeriesOperatorRightMultiplyAccumulate.
// Uses spse_b as the F-space temporary.
int op_power_series(ceres_hip_solver* s, const double* x, double* y, const int* status) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  TRY(ensure_ftf_inverse(s));
  double* t = s->spse_b;
  if (s->path == CERES_HIP_PATH_BAL) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.x_f = x;
    TRY(bal_scatter(s, kBalSpseZ, A, x, t, false, status));
  } else {
    HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
    HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kF, x, s->tmp_rows, status, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_e, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kE, s->tmp_rows, s->tmp_e, status, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_e2, 0, sizeof(double) * std::max(1, h.num_cols_e), st));
    HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, 0, h.nelim, s->G.diag_off_e, s->etei, s->tmp_e, s->tmp_e2, status, st));
    HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
    HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kE, s->tmp_e2, s->tmp_rows, status, st));
    HIP_TRY(s, hipMemsetAsync(t, 0, sizeof(double) * h.num_cols_f, st));
    HIP_TRY(s, LaunchGenLeftMultiply(s->G, s->values, kF, s->tmp_rows, t, status, st));
    TRY(allreduce(s, t, size_t(h.num_cols_f)));
  }
  HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, h.nelim, h.ncb - h.nelim, s->G.diag_off_f, s->ftf_inv, t, y, status, st));
  return 0;
}

// y = sum_k Z^k (F^T F)^-1 x.  PowerSeriesExpansionPreconditioner::RightMultiplyAndAccumulate.
// tolerance == 0 (the preconditioner's setting) runs a fixed number of terms with no host sync.
int op_spse_apply(ceres_hip_solver* s, const double* x, double* y, int max_iters, double tolerance, const int* status) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  const int64_t n = h.num_cols_f;
  TRY(ensure_ftf_inverse(s));
  HIP_TRY(s, hipMemsetAsync(y, 0, sizeof(double) * n, st));
  HIP_TRY(s, LaunchGenBlockDiagonalApply(s->G, h.nelim, h.ncb - h.nelim, s->G.diag_off_f, s->ftf_inv, x, y, status, st));
  double* prev = s->spse_a;
  double* term = s->own_x;  // num_cols >= num_cols_f doubles, free during CG
  HIP_TRY(s, hipMemcpyAsync(prev, y, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  double threshold = 0.0;
  auto host_norm = [&](const double* v, double* out) -> int {
    HIP_TRY(s, LaunchDot(v, v, n, s->scalar_partials, s->cg.comm + 3, st));
    double d = 0;
    HIP_TRY(s, hipMemcpyAsync(&d, s->cg.comm + 3, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
    *out = std::sqrt(d);
    return 0;
  };
  if (tolerance > 0.0) { double ny = 0; TRY(host_norm(y, &ny)); threshold = tolerance * ny; }
  for (int i = 1;; ++i) {
    HIP_TRY(s, hipMemsetAsync(term, 0, sizeof(double) * n, st));
    TRY(op_power_series(s, prev, term, status));
    HIP_TRY(s, LaunchAxpby(1.0, y, 1.0, term, y, n, st));
    if (i >= max_iters) break;
    if (tolerance > 0.0) { double nt = 0; TRY(host_norm(term, &nt)); if (nt < threshold) break; }
    std::swap(prev, term);
  }
  return 0;
}

// diag(J^T J) into out (num_cols).  BlockSparseMatrix::SquaredColumnNorm.
int op_squared_column_norm(ceres_hip_solver* s, double* out) {
  const HostStructure& h = s->hs;
  if (s->path == CERES_HIP_PATH_BAL && s->lds_mode) {  // sharded: bal_scatter all-reduces the camera part
    BalArgs A = bal_args(s);
    A.y_e = out;
    PackGuard g = use_gather_if_unpacked(s, A);
    return g.commit(bal_scatter(s, kBalColNorm, A, nullptr, out + h.num_cols_e, false, nullptr));
  }
  TRY(require_caller_values(s, "SquaredColumnNorm with the cameras' sums outside LDS"));
  HIP_TRY(s, LaunchGenSquaredColumnNorm(s->G, s->values, out, s->stream));
  if (s->world > 1) TRY(allreduce(s, out + h.num_cols_e, size_t(h.num_cols_f)));
  return 0;
}

// -(J x)'(b + J x / 2) into *host_out.
// -(J x)'(b + J x / 2): enqueues the kernels; the result is the sum of `*nparts` doubles at
// `*dev_parts` once the stream has drained (fixed summation order on the host: deterministic).
int enqueue_model_cost_change(ceres_hip_solver* s, const double* x, const double** dev_parts, int* nparts) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  if (s->path == CERES_HIP_PATH_BAL) {
    TRY(ensure_packed(s));
    BalArgs A = bal_args(s);
    A.x_e = x; A.x_f = x + h.num_cols_e;
    A.scalar_out = s->scalar_partials;
    *nparts = std::min(s->fused_grid, kMaxVecGrid - 1);
    *dev_parts = s->scalar_partials;
    HIP_TRY(s, s->ops->fused(kBalJx, A, false, *nparts, st));
    if (has_remainder(s)) { TRY(rem_model_cost(s, x + h.num_cols_e, 1, s->scalar_partials + *nparts)); ++*nparts; }
    return 0;
  }
  // model = J x ; partial sums of -model .* (b + model / 2)
  HIP_TRY(s, hipMemsetAsync(s->tmp_rows, 0, sizeof(double) * h.num_rows, st));
  HIP_TRY(s, LaunchGenRightMultiply(s->G, s->values, kAll, x, s->tmp_rows, nullptr, st));
  double* t2 = s->scratch_vec + h.num_cols;  // num_rows doubles
  HIP_TRY(s, LaunchAxpby(-1.0, s->b, -0.5, s->tmp_rows, t2, h.num_rows, st));
  HIP_TRY(s, LaunchDot(s->tmp_rows, t2, h.num_rows, s->scalar_partials, s->cg.comm, st));
  *nparts = 1;
  *dev_parts = s->cg.comm;
  return 0;
}
// ---------------------------------------------------------------------------
// Conjugate gradients driver (I/conjugate_gradients_solver.h:108-306).
// ---------------------------------------------------------------------------
struct CgSpec {
  int64_t n = 0;
  int64_t n_local = 0;                                  // sharded CGNR: E-space prefix
  std::function<int(const double*, double*)> apply;     // y = A x (assigns)
  // optional: y = A x AND the partial sums of x . y into pq[0 .. *n_pq) (n_pq <= kMaxPqParts); *n_pq = 0 if not produced
  // sharded: *extra = device pointer to the shard's share of x . y, already summed over ranks (nullptr if not produced)
  std::function<int(const double*, double*, double*, int*, const double**)> apply_dot;
  std::function<int()> before_poll;                     // enqueued before every poll of the status word (speculative LM tail)
  // optional (small camera spaces, CgTail in device.h): q = A p AND the rest of fused iteration `it` in one launch; returns 1 if it
  // did, 0 if this launch cannot (the caller then runs the iteration the usual way), < 0 on error
  std::function<int(int)> iteration;
  bool shard_fused = false;                             // sharded CG vectors, and apply_dot + the block layout support the fused iteration
  std::function<int(const double*, double*)> precondition;  // z = M^-1 r as an operator (SPSE); empty = block diagonal
  bool x0_nonzero = false;                              // B.x holds an initial guess
  const double* rhs = nullptr;                          // right-hand side (nullptr: s->cg_rhs)
  int first_block = 0, col_begin = 0, nblocks = 0, n_local_blocks = 0;
  const int64_t* diag_off = nullptr;
  const double* blocks = nullptr;                       // nullptr = IDENTITY
  const int* setup_fail = nullptr;  // device flag raised by the preconditioner set-up; checked on the device by the CG init kernels
};

void fill_summary(const CgScalars& S, int device_status, ceres_hip_summary* out) {
  out->residual_norm = -1.0;
  out->num_iterations = S.iter;
  char* m = out->message;
  const size_t n = sizeof(out->message);
  switch (device_status) {
    case kCgZeroRhs:
      out->termination_type = CERES_HIP_SUCCESS; out->num_iterations = 0;
      snprintf(m, n, "Convergence. |b| = 0."); break;
    case kCgInitialResidual:
      out->termination_type = CERES_HIP_SUCCESS; out->num_iterations = 0;
      snprintf(m, n, "Convergence. |r| = %e <= %e.", S.norm_r, S.tol_r); break;
    case kCgConvergedZeta:
      out->termination_type = CERES_HIP_SUCCESS;
      snprintf(m, n, "Iteration: %d Convergence: zeta = %e < %e. |r| = %e", S.iter, S.zeta, S.q_tol, S.norm_r); break;
    case kCgConvergedResidual:
      out->termination_type = CERES_HIP_SUCCESS;
      snprintf(m, n, "Iteration: %d Convergence. |r| = %e <= %e.", S.iter, S.norm_r, S.tol_r); break;
    case kCgMaxIterations:
      out->termination_type = CERES_HIP_NO_CONVERGENCE;
      snprintf(m, n, "Maximum number of iterations reached."); break;
    case kCgFailRho:
      out->termination_type = CERES_HIP_FAILURE;
      snprintf(m, n, "Numerical failure. rho = r'z = %e.", S.rho_new); break;
    case kCgFailBeta:
      out->termination_type = CERES_HIP_FAILURE;
      snprintf(m, n, "Numerical failure. beta = rho_n / rho_{n-1} = %e, rho_n = %e, rho_{n-1} = %e", S.beta, S.rho_new, S.rho); break;
    case kCgIndefinite:
      out->termination_type = CERES_HIP_NO_CONVERGENCE;
      snprintf(m, n, "Matrix is indefinite, no more progress can be made. p'q = %e.", S.pq); break;
    case kCgSetupFailed:  // Preconditioner::Update returned false, I/iterative_schur_complement_solver.cc:113-121, I/cgnr_solver.cc:176-183
      out->termination_type = CERES_HIP_FAILURE; out->num_iterations = 0;
      snprintf(m, n, "Preconditioner update failed."); break;
    case kCgFailAlpha:
      out->termination_type = CERES_HIP_FAILURE;
      snprintf(m, n, "Numerical failure. alpha = rho / pq = %e, rho = %e, pq = %e.", S.alpha, S.rho_new, S.pq); break;
    default:
      out->termination_type = CERES_HIP_FATAL_ERROR;
      snprintf(m, n, "CG ended in unknown device state %d", device_status);
  }
}

int check_comm_error(ceres_hip_solver* s) {  // after a stream synchronisation
  // the flag lives in mapped pinned host memory (a timed-out kernel stores into it with system scope): no copy, no extra synchronisation
  if (!s->p2p || !s->h_comm_error) return 0;
  if (*static_cast<volatile int*>(s->h_comm_error))
    return fail(s, CERES_HIP_E_COMM, "peer-to-peer all-reduce timed out after %.1f s: a rank did not arrive (its output was poisoned with NaN)", s->p2p_timeout_s);
  return 0;
}

// Host side of the mailbox: spin on the stamp; every few thousand looks ask the runtime whether the stream has drained (a faulted
// kernel never stamps).
int wait_mailbox(ceres_hip_solver* s, unsigned long long seq) {
  for (unsigned spin = 1;; ++spin) {
    if (__atomic_load_n(s->h_stamp, __ATOMIC_ACQUIRE) == seq) break;
    if ((spin & 4095u) == 0) {
      const hipError_t q = hipStreamQuery(s->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(s->h_stamp, __ATOMIC_ACQUIRE) == seq) break;
        return fail(s, CERES_HIP_E_HIP, "read-back kernel finished without stamping the mailbox");
      }
      if (q != hipErrorNotReady) return fail(s, CERES_HIP_E_HIP, "stream failed while waiting for a read-back: %s", hipGetErrorString(q));
    }
    __builtin_ia32_pause();
  }
  return check_comm_error(s);
}
// h_pinned[0 .. n) <- src[0 .. n) (device), and wait for it
int read_back(ceres_hip_solver* s, const double* src, int n) {
  if (s->mailbox && s->d_h_pinned) {
    const unsigned long long seq = ++s->mailbox_seq;
    HIP_TRY(s, LaunchMailbox(src, n, s->d_h_pinned, s->d_stamp, seq, s->stream));
    return wait_mailbox(s, seq);
  }
  HIP_TRY(s, hipMemcpyAsync(s->h_pinned, src, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return check_comm_error(s);
}
int poll_scalars(ceres_hip_solver* s) {
  if (s->tail_mailbox_seq != 0) {   // the speculative tail's last kernel was this poll's read-back already
    const unsigned long long seq = s->tail_mailbox_seq;
    s->tail_mailbox_seq = 0;
    return wait_mailbox(s, seq);
  }
  return read_back(s, s->scalar_partials, kReadbackDoubles);
}

int collapse_and_reduce(ceres_hip_solver* s, int first_slot, int count) {
  if (s->cg.grid_e == 0) return 0;
  if (s->world > 1 && s->p2p && s->p2p_fuse && count <= 64) {   // collapse and sum over ranks in one launch (the slots of the all-reduce of `count` doubles)
    HIP_TRY(s, LaunchCgCollapseExchange(s->cg, first_slot, count, next_exchange(s), s->stream));
    return 0;
  }
  HIP_TRY(s, LaunchCgCollapse(s->cg, first_slot, count, s->stream));
  return allreduce(s, s->cg.comm + first_slot, size_t(count));
}

// rhs in s->cg_rhs; solution left in s->cg.x.
int run_cg(ceres_hip_solver* s, const CgSpec& spec, double q_tol, double r_tol, ceres_hip_summary* summary) {
  hipStream_t st = s->stream;
  CgBuffers& B = s->cg;
  B.n = spec.n;
  B.n_local = (s->world > 1) ? spec.n_local : 0;
  B.rhs = spec.rhs ? spec.rhs : s->cg_rhs;
  B.setup_fail = spec.setup_fail;
  // one element per thread until the cap: the short camera-space vectors of ITERATIVE_SCHUR (16 k doubles) are
  // latency-bound, and more, shorter workgroups finish sooner than 16 long ones
  auto grid_for = [](int64_t n, int cap) {
    int64_t g = (n + int64_t(kVecBlock) - 1) / int64_t(kVecBlock);
    return int(std::max<int64_t>(1, std::min<int64_t>(g, cap)));
  };
  if (B.n_local > 0) {
    B.grid_e = grid_for(B.n_local, kMaxVecGrid / 2);
    B.grid = B.grid_e + grid_for(B.n - B.n_local, kMaxVecGrid / 2);
  } else {
    B.grid_e = 0;
    B.grid = grid_for(B.n, kMaxVecGrid);
  }
  const int min_it = s->opt.min_num_iterations, max_it = s->opt.max_num_iterations;
  const int reset_period = std::max(1, s->opt.residual_reset_period);
  const int* status = &B.S->status;
  // Iterations enqueued between polls of the device status word: fixed if the caller asked
  // for it, otherwise 2, 4, 8, 16, 16, ... (short solves do not pay for no-op launches,
  // long ones poll rarely).
  const bool adaptive = s->opt.cg_check_interval <= 0;
  // The FIRST batch is as long as the previous solve of this instance was (consecutive trust-region steps need about the same number
  // of iterations: no surplus launches, one poll); then 2, 4, 8, 16, 16, ...
  int interval = adaptive ? std::max(2, std::min(s->last_cg_iterations, 64)) : s->opt.cg_check_interval;
  bool first_batch = true;

  // Fused iteration: operator (+ p.q where its kernels have p and q in registers) -> cg_update (alpha, x, r, M^-1 r)
  // -> cg_finalize_direction (tests, beta, p): 2 launches after the operator instead of 5.  Needs a block-diagonal
  // (or no) preconditioner and unsharded CG vectors (ITERATIVE_SCHUR's camera-space vectors are replicated: fine).
  const bool fused = s->cg_fused && !spec.precondition && (B.grid_e == 0 || spec.shard_fused);
  const bool fused_start = fused && !spec.x0_nonzero;  // x0 = 0: the whole start of the solve is two launches
  if (fused_start) {
    HIP_TRY(s, LaunchCgUpdate(B, s->G, spec.first_block, spec.col_begin, spec.nblocks, spec.diag_off, spec.blocks, 0, 0,
                              s->nine_wide_from - spec.first_block, st));
    TRY(collapse_and_reduce(s, 0, 4));  // sharded: |rhs|^2 and r.z of the shard, summed over ranks (no-op otherwise)
    HIP_TRY(s, LaunchCgBegin(B, q_tol, r_tol, min_it, max_it, st));
  } else {
  HIP_TRY(s, LaunchCgRhsNorm(B, st));
  TRY(collapse_and_reduce(s, 0, 1));
  if (spec.x0_nonzero) {  // r = rhs - A x0, Q0 = -x0.(rhs + r)   (:138-159)
    // the operator kernels return early on a non-zero status word: clear what the previous solve left
    HIP_TRY(s, hipMemsetAsync(&B.S->status, 0, sizeof(int), st));
    TRY(spec.apply(B.x, B.z));
    HIP_TRY(s, LaunchCgInitFromGuess(B, B.z, q_tol, r_tol, min_it, max_it, st));
  } else {
    HIP_TRY(s, LaunchCgInit(B, q_tol, r_tol, min_it, max_it, st));
  }
  }
  // No poll here: if |b| = 0 or r0 already meets the tolerance the status word is set and the
  // first batch below is a string of no-ops; the first poll comes after it.
  s->h_scalars->status = kCgRunning;
  s->timing.operator_applications = 0;
  int it = 1;
  auto precondition = [&]() -> int {
    if (spec.precondition) {
      TRY(spec.precondition(B.r, B.z));
      HIP_TRY(s, LaunchCgDotSlot(B, B.r, B.z, 0, st));
    } else {
      HIP_TRY(s, LaunchCgPrecondition(B, s->G, spec.first_block, spec.col_begin, spec.nblocks,
                                      B.grid_e > 0 ? spec.n_local_blocks : 0, spec.diag_off, spec.blocks, st));
    }
    return 0;
  };
  if (fused && !fused_start) {  // iteration 1's z = M^-1 r0, rho_1 and p = z; later directions come out of cg_finalize_direction
    TRY(precondition());
    TRY(collapse_and_reduce(s, 0, 1));
    HIP_TRY(s, LaunchCgDirection(B, st));
  }
  while (s->h_scalars->status == kCgRunning) {
    const int batch_end = std::min(max_it, it + interval - 1);
    for (; it <= batch_end; ++it) {
      const int reset = (it % reset_period == 0) ? 1 : 0;
      if (fused && spec.iteration && !reset) {
        const int done = spec.iteration(it);
        if (done < 0) return CERES_HIP_E_HIP;
        if (done > 0) { ++s->timing.operator_applications; continue; }
      }
      if (fused) {
        int n_pq = 0;
        const double* extra = nullptr;
        if (spec.apply_dot) TRY(spec.apply_dot(B.p, B.z, s->cg_pq_parts, &n_pq, &extra));
        else TRY(spec.apply(B.p, B.z));
        B.pq_extra = extra;
        if (n_pq > 0) { B.pq_parts = s->cg_pq_parts; B.n_pq = n_pq; }
        else if (B.grid_e > 0) return fail(s, CERES_HIP_E_INVALID, "internal: sharded fused CG needs p.q from the operator");
        else {  // the operator did not leave p.q behind: one pass over p and q
          HIP_TRY(s, LaunchCgDotPq(B, st));
          B.pq_parts = B.partials + kMaxVecGrid; B.n_pq = B.grid;
        }
        HIP_TRY(s, LaunchCgUpdate(B, s->G, spec.first_block, spec.col_begin, spec.nblocks, spec.diag_off, spec.blocks, reset, it,
                                  s->nine_wide_from - spec.first_block, st));
        ++s->timing.operator_applications;
        if (reset) {  // r = rhs - A x (:235-239), then the next iteration's z = M^-1 r
          TRY(spec.apply(B.x, B.z));
          HIP_TRY(s, LaunchCgResidualReset(B, B.z, st));
          TRY(precondition());
          ++s->timing.operator_applications;
        }
        TRY(collapse_and_reduce(s, 0, 4));  // sharded: the shard's r.z, Q1, |r|^2 in one 4-double all-reduce (no-op otherwise)
        HIP_TRY(s, LaunchCgFinalizeDirection(B, it, st));
        continue;
      }
      TRY(precondition());
      TRY(collapse_and_reduce(s, 0, 1));
      HIP_TRY(s, LaunchCgDirection(B, st));
      TRY(spec.apply(B.p, B.z));
      HIP_TRY(s, LaunchCgDotPq(B, st));
      TRY(collapse_and_reduce(s, 1, 1));
      HIP_TRY(s, LaunchCgStep(B, reset, st));
      ++s->timing.operator_applications;
      if (reset) {
        TRY(spec.apply(B.x, B.z));
        HIP_TRY(s, LaunchCgResidualReset(B, B.z, st));
        ++s->timing.operator_applications;
      }
      TRY(collapse_and_reduce(s, 2, 2));
      HIP_TRY(s, LaunchCgFinalize(B, st));
    }
    if (spec.before_poll) TRY(spec.before_poll());
    TRY(poll_scalars(s));
    if (adaptive) { interval = first_batch ? 2 : std::min(16, interval * 2); first_batch = false; }
    if (it > max_it && s->h_scalars->status == kCgRunning)
      return fail(s, CERES_HIP_E_INVALID, "CG did not terminate after max_num_iterations (device status 0)");
  }
  (void)status;
  if (s->h_scalars->status == kCgZeroRhs && spec.x0_nonzero) HIP_TRY(s, hipMemsetAsync(B.x, 0, sizeof(double) * B.n, st));
  fill_summary(*s->h_scalars, s->h_scalars->status, summary);
  s->last_cg_iterations = summary->num_iterations;
  return 0;
}

int check_factorization(ceres_hip_solver* s, bool* failed) {
  int flag = 0;
  HIP_TRY(s, hipMemcpyAsync(&flag, s->d_fail_flag, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (s->world > 1) {
    // A verdict the ranks SHARE: the flag is raised by this rank's blocks (a point block of its shard), and a rank that left the solve
    // here on its own would leave its peers waiting in the next all-reduce — and with another answer (tools/probes/poison_ranks.py:
    // one rank "E^T E + D^2 is not positive definite", the other a time-out, and the exchange epochs of the two apart for good).
    // Every rank reaches these checks in the same order (they depend on the options and the kernel path only).
    double v = flag != 0 ? 1.0 : 0.0;
    if (!s->d_verdict) TRY(dev_alloc(s, &s->d_verdict, 1));
    HIP_TRY(s, hipMemcpyAsync(s->d_verdict, &v, sizeof(double), hipMemcpyHostToDevice, s->stream));
    TRY(allreduce(s, s->d_verdict, 1));
    HIP_TRY(s, hipMemcpyAsync(&v, s->d_verdict, sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    TRY(check_comm_error(s));
    flag = !(v == 0.0);   // (NaN: an exchange that timed out)
  }
  *failed = flag != 0;
  return 0;
}

int require_loaded(ceres_hip_solver* s) {
  if (!s) return CERES_HIP_E_INVALID;
  if (!s->have_structure) return fail(s, CERES_HIP_E_INVALID, "ceres_hip_set_structure has not been called");
  if (!s->loaded) return fail(s, CERES_HIP_E_INVALID, "no values loaded (ceres_hip_load / ceres_hip_solve)");
  return 0;
}

// values/b/D are device pointers valid until the next load.
int load_device(ceres_hip_solver* s, const double* dv, const double* db, const double* dD) {
  if (!s->have_structure) return fail(s, CERES_HIP_E_INVALID, "ceres_hip_set_structure has not been called");
  if (!dv) return fail(s, CERES_HIP_E_INVALID, "values == NULL");
  s->values = dv; s->b = db; s->D = dD;
  s->have_b = db != nullptr;
  s->have_D = dD != nullptr;
  s->precond_valid = false;
  s->ftf_inv_valid = false;
  s->packed = false;  // the tiles are (re)built by the first kernel that walks J, or by ensure_packed()
  s->tiles_only = false;
  // (values from anybody but the device evaluator: the camera-major pass reads its cells again — bal_frontend.inc sets these after its load)
  s->cam_items.ev_cam_pack = nullptr; s->cam_items.ev_pt_pack = nullptr; s->cam_items.ev_pt = nullptr; s->cam_items.ev_obs = nullptr;
  s->rem_blocks_valid = false;
  s->D_int_valid = false;
  s->loaded = true;
  return 0;
}

// The retry after a rejected step: same values, same residuals (ceres_hip_lm_options::values_unchanged); only what contains D is stale.
// The packed tiles, the remainder rows' blocks and the loaded pointers stay.
void keep_loaded_values(ceres_hip_solver* s) {
  s->precond_valid = false;
  s->ftf_inv_valid = false;
  s->D_int_valid = false;
}

int load_host(ceres_hip_solver* s, const double* hv, const double* hb, const double* hD) {
  if (!s->have_structure) return fail(s, CERES_HIP_E_INVALID, "ceres_hip_set_structure has not been called");
  if (!hv) return fail(s, CERES_HIP_E_INVALID, "values == NULL");
  const HostStructure& h = s->hs;
  HIP_TRY(s, hipMemcpyAsync(s->own_values, hv, sizeof(double) * h.values_extent, hipMemcpyHostToDevice, s->stream));
  if (hb) HIP_TRY(s, hipMemcpyAsync(s->own_b, hb, sizeof(double) * h.num_rows, hipMemcpyHostToDevice, s->stream));
  if (hD) HIP_TRY(s, hipMemcpyAsync(s->own_D, hD, sizeof(double) * h.num_cols, hipMemcpyHostToDevice, s->stream));
  return load_device(s, s->own_values, hb ? s->own_b : nullptr, hD ? s->own_D : nullptr);
}

float elapsed(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  // An event of a phase that an early return skipped was never recorded: the query fails, the
  // phase reads as 0 ms, and the runtime's sticky last-error must not leak into the next launch check.
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0.0f; }
  return ms;
}

// Phase events (ceres_hip_get_last_timing): recorded only when the caller asked for them (ceres_hip_set_phase_timing / CERES_HIP_TIMING=1).
// An event record between two kernels is a barrier packet with a completion signal: about 6 us of idle device each (rocprofv3
// --kernel-trace, profiles/r06a_*), seven of them in an LM step — a tenth of what one rank of eight spends on a Venice-sized step.
int rec(ceres_hip_solver* s, int i) {
  if (i == 0) s->call_t0 = std::chrono::steady_clock::now();
  if (!s->timing_enabled) return 0;
  HIP_TRY(s, hipEventRecord(s->ev[i], s->stream));
  return 0;
}

// The two LinearSolver::SolveImpl bodies.  x is a device pointer (num_cols).
int solve_loaded_impl(ceres_hip_solver* s, double q_tol, double r_tol, double* x, ceres_hip_summary* summary);
int solve_loaded(ceres_hip_solver* s, double q_tol, double r_tol, double* x, ceres_hip_summary* summary) {
  int rc = solve_loaded_impl(s, q_tol, r_tol, x, summary);
  // a deferred rhs all-reduce nobody took along (an early return between op_schur_init and the preconditioner): pay it now, so that
  // every rank issues the same collectives and a later op-level call does not find a stale promise
  if (s->rhs_reduce_pending) {
    s->rhs_reduce_pending = false;
    if (rc == 0) rc = allreduce(s, s->rhs_f, size_t(s->hs.num_cols_f));
  }
  // "the flags were cleared together at the start" holds INSIDE one solve only: paths that never consume the promise (DENSE_SCHUR,
  // explicit S, IDENTITY, early returns) clear and raise d_fail_flag themselves, and a later op-level call must not skip its memset
  s->fail_flag_clean = false;
  s->nonfinite_clean = false;
  if (s->clear_flags_pending) {   // no fused first pass took the clearing along: cannot happen on the paths that set it — an error, not a silent skip
    s->clear_flags_pending = false;
    if (rc == 0) rc = fail(s, CERES_HIP_E_INVALID, "internal: the solve's flags were never cleared (no first pass ran)");
  }
  return rc;
}
int solve_loaded_impl(ceres_hip_solver* s, double q_tol, double r_tol, double* x, ceres_hip_summary* summary) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  memset(summary, 0, sizeof(*summary));
  summary->residual_norm = -1.0;
  if (!s->have_b) return fail(s, CERES_HIP_E_INVALID, "Solve needs the residual vector b");
  // D may have changed since the last solve on the same values (a rejected trust-region step retries with a
  // smaller radius, bal_frontend.inc): everything cached that contains D is rebuilt, as
  // ImplicitSchurComplement::Init / Preconditioner::Update do on every Solve.
  s->ftf_inv_valid = false;
  s->precond_valid = false;
  // finite-step flag + factorization flag, adjacent.  Fused path: the solve's first pass clears them itself (BalArgs::clear_flags:
  // op_schur_init / op_cgnr_setup_bal, which every fused solve starts with) — no fill command in front of it
  s->clear_flags_pending = s->path == CERES_HIP_PATH_BAL && s->ops != nullptr;
  if (!s->clear_flags_pending) HIP_TRY(s, hipMemsetAsync(s->d_nonfinite, 0, 2 * sizeof(int), st));
  s->fail_flag_clean = true;
  s->nonfinite_clean = true;
  TRY(rec(s, 2));
  if (is_schur(s)) {
    // IterativeSchurComplementSolver::SolveImpl, I/iterative_schur_complement_solver.cc:64-157
    const int pre = s->opt.preconditioner_type;
    // sharded fused path, block-diagonal preconditioner from the camera-major pass: rhs, the blocks and the camera column norms go
    // through ONE all-reduce, issued by op_preconditioner
    // (armed only where the code below REACHES that op_preconditioner call: the implicit iterative branch with F blocks — a direct or
    // explicit-S solve, or the no-F-blocks shortcut, would otherwise factor / iterate on this rank's raw rhs)
    const bool implicit_iterative = !is_dense_schur(s) && !s->sparse_S && !s->opt.use_explicit_schur_complement && h.ncb - h.nelim > 0;
    // (round 6: not when the camera-major pass exchanges its blocks itself — a verdict all ranks share —: rhs is then summed over ranks by
    // the first pass's own reduction, or by the stand-alone all-reduce on a rank whose reduction cannot: the same slots either way)
    s->merge_step_reduce = s->world > 1 && s->path == CERES_HIP_PATH_BAL && (pre == CERES_HIP_SCHUR_JACOBI || pre == CERES_HIP_JACOBI) &&
                           !s->opt.use_spse_initialization && implicit_iterative && !camera_blocks_exchange(s);
    const int rc_init = op_schur_init(s, pre == CERES_HIP_SCHUR_JACOBI);
    s->merge_step_reduce = false;
    TRY(rc_init);
    bool bad = false;
    if (s->path != CERES_HIP_PATH_BAL) TRY(check_factorization(s, &bad));
    if (bad) {
      summary->termination_type = CERES_HIP_FAILURE;
      snprintf(summary->message, sizeof(summary->message), "E^T E + D^2 is not positive definite.");
      return 0;
    }
    TRY(rec(s, 3));
    const int nf_blocks = h.ncb - h.nelim;
    if (nf_blocks == 0) {  // :88-95
      summary->termination_type = CERES_HIP_SUCCESS;
      summary->num_iterations = 0;
      TRY(op_back_substitute(s, nullptr, x));
      TRY(rec(s, 4));
      TRY(rec(s, 5));
      TRY(rec(s, 6));
      return 0;
    }
    if (is_dense_schur(s)) {
      // DenseSchurComplementSolver (I/schur_complement_solver.cc:163-222): S dense by elimination, Cholesky of its upper
      // triangle (DenseCholesky::FactorAndSolve), back-substitution on SUCCESS.
      const int64_t nf = h.num_cols_f;
      if (s->dense_from_blocks) {
        const double* ei = s->etei;
        if (s->path == CERES_HIP_PATH_BAL) {  // packed 6-per-point inverses (internal point order) -> the dense E-block store
          HIP_TRY(s, LaunchExpandSym(s->etei, s->ops->ne, s->ops->etei_pitch, s->etei_dense, s->d_pt_eoff, s->plan.n_points, st));
          ei = s->etei_dense;
        }
        HIP_TRY(s, LaunchSchurSparseEliminate(s->G, s->schur_pairs, s->values, ei, s->D, s->d_Sblk, st));
        HIP_TRY(s, hipMemsetAsync(s->d_S, 0, sizeof(double) * size_t(nf) * size_t(nf), st));
        HIP_TRY(s, LaunchSchurBlocksToDense(s->G, s->schur_pairs, s->d_Sblk, s->schur_storage.num_values(), s->d_S, st));
      } else {
        HIP_TRY(s, LaunchGenSchurDense(s->G, s->values, s->etei, (s->world > 1 && s->rank != 0) ? nullptr : s->D, s->d_S, st));
        if (s->world > 1) TRY(allreduce(s, s->d_S, size_t(nf * nf)));
      }
      TRY(rec(s, 4));
      HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
      HIP_TRY(s, LaunchDenseCholesky(s->d_S, int(nf), s->d_fail_flag, st));
      TRY(check_factorization(s, &bad));
      summary->num_iterations = 1;
      if (bad) {  // EigenDenseCholesky::Factorize, I/dense_cholesky.cc
        summary->termination_type = CERES_HIP_FAILURE;
        snprintf(summary->message, sizeof(summary->message), "Eigen failure. Unable to perform dense Cholesky factorization.");
        return 0;
      }
      HIP_TRY(s, hipMemcpyAsync(s->cg.x, s->rhs_f, sizeof(double) * nf, hipMemcpyDeviceToDevice, st));
      HIP_TRY(s, LaunchDenseCholeskySolve(s->d_S, int(nf), s->cg.x, st));
      TRY(rec(s, 5));
      summary->termination_type = CERES_HIP_SUCCESS;
      snprintf(summary->message, sizeof(summary->message), "Success.");
      TRY(op_back_substitute(s, s->cg.x, x));
      TRY(rec(s, 6));
      return 0;
    }
    if (s->sparse_S) {
      // SparseSchurComplementSolver with use_explicit_schur_complement: S in BlockRandomAccessSparseMatrix storage
      // (InitStorage's block pairs), SchurEliminator::Eliminate as a gather per stored block, SCHUR_JACOBI = inverted
      // diagonal blocks OF S, CG with SymmetricRightMultiplyAndAccumulate, back-substitution only on SUCCESS
      // (I/schur_complement_solver.cc:100-158, 224-290, 337-408).
      const int64_t nf = h.num_cols_f;
      HIP_TRY(s, LaunchSchurSparseEliminate(s->G, s->schur_pairs, s->values, s->etei, s->D, s->d_S, st));
      HIP_TRY(s, LaunchSchurSparseDiag(s->G, s->schur_pairs, s->d_S, s->G.diag_off_f, s->precond, h.diag_off_f.back(), st));
      HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
      HIP_TRY(s, LaunchGenInvertBlocks(s->G, h.nelim, h.ncb - h.nelim, s->G.diag_off_f, s->precond, s->d_fail_flag, st));
      TRY(check_factorization(s, &bad));
      if (bad) {
        summary->termination_type = CERES_HIP_FAILURE;
        snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
        return 0;
      }
      s->precond_valid = true;
      TRY(rec(s, 4));
      CgSpec spec;
      spec.rhs = s->rhs_f;
      spec.n = nf;
      spec.n_local = 0;
      const int* status = &s->cg.S->status;
      spec.apply = [s, status](const double* in, double* out) -> int {
        HIP_TRY(s, LaunchSchurSparseSymv(s->G, s->schur_pairs, s->d_S, in, out, status, 0, s->stream));
        return 0;
      };
      spec.first_block = h.nelim;
      spec.nblocks = h.ncb - h.nelim;
      spec.col_begin = h.num_cols_e;
      spec.diag_off = s->G.diag_off_f;
      spec.blocks = s->precond;
      TRY(run_cg(s, spec, q_tol, r_tol, summary));
      TRY(rec(s, 5));
      if (summary->termination_type == CERES_HIP_SUCCESS) {
        TRY(op_back_substitute(s, s->cg.x, x));
      } else {  // x was zero-filled (:135), the reduced solution sits in its tail
        HIP_TRY(s, hipMemsetAsync(x, 0, sizeof(double) * h.num_cols_e, st));
        HIP_TRY(s, hipMemcpyAsync(x + h.num_cols_e, s->cg.x, sizeof(double) * nf, hipMemcpyDeviceToDevice, st));
      }
      TRY(rec(s, 6));
      return 0;
    }
    if (s->opt.use_explicit_schur_complement) {
      // sharded runs keep S DENSE (one all-reduce of the ranks' contributions):
      // SchurComplementSolver::SolveImpl (I/schur_complement_solver.cc:100-158) with
      // SolveReducedLinearSystemUsingConjugateGradients (:337-408): S and rhs by elimination,
      // SCHUR_JACOBI = inverted diagonal blocks OF S, CG on S, back-substitution only on SUCCESS.
      const int64_t nf = h.num_cols_f;
      // D_f^2 joins the diagonal once: on rank 0 when the elimination is sharded by point
      HIP_TRY(s, LaunchGenSchurDense(s->G, s->values, s->etei, (s->world > 1 && s->rank != 0) ? nullptr : s->D, s->d_S, st));
      if (s->world > 1) TRY(allreduce(s, s->d_S, size_t(nf * nf)));  // "reduced-system contributions combined via all-reduce"
      HIP_TRY(s, LaunchGenSymmetrizeDense(s->G, s->d_S, st));
      HIP_TRY(s, LaunchGenExtractDiagBlocks(s->G, s->d_S, s->G.diag_off_f, s->precond, st));
      HIP_TRY(s, hipMemsetAsync(s->d_fail_flag, 0, sizeof(int), st));
      HIP_TRY(s, LaunchGenInvertBlocks(s->G, h.nelim, h.ncb - h.nelim, s->G.diag_off_f, s->precond, s->d_fail_flag, st));
      TRY(check_factorization(s, &bad));
      if (bad) {
        summary->termination_type = CERES_HIP_FAILURE;
        snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
        return 0;
      }
      s->precond_valid = true;
      TRY(rec(s, 4));
      HIP_TRY(s, hipMemcpyAsync(s->cg_rhs, s->rhs_f, sizeof(double) * nf, hipMemcpyDeviceToDevice, st));
      CgSpec spec;
      spec.n = nf;
      spec.n_local = 0;
      const int* status = &s->cg.S->status;
      spec.apply = [s, status, nf](const double* in, double* out) -> int {
        HIP_TRY(s, LaunchGenDenseSymv(s->d_S, int(nf), in, out, status, s->stream));
        return 0;
      };
      spec.first_block = h.nelim;
      spec.nblocks = h.ncb - h.nelim;
      spec.col_begin = h.num_cols_e;
      spec.diag_off = s->G.diag_off_f;
      spec.blocks = s->precond;
      TRY(run_cg(s, spec, q_tol, r_tol, summary));
      TRY(rec(s, 5));
      if (summary->termination_type == CERES_HIP_SUCCESS) {
        TRY(op_back_substitute(s, s->cg.x, x));
      } else {  // x was zero-filled (:135), the reduced solution sits in its tail
        HIP_TRY(s, hipMemsetAsync(x, 0, sizeof(double) * h.num_cols_e, st));
        HIP_TRY(s, hipMemcpyAsync(x + h.num_cols_e, s->cg.x, sizeof(double) * nf, hipMemcpyDeviceToDevice, st));
      }
      TRY(rec(s, 6));
      return 0;
    }
    const bool spse_pre = pre == CERES_HIP_SCHUR_POWER_SERIES_EXPANSION;
    const int spse_iters = s->opt.max_num_spse_iterations > 0 ? s->opt.max_num_spse_iterations : 5;
    if (spse_pre || s->opt.use_spse_initialization) {
      TRY(ensure_ftf_inverse(s));
      TRY(check_factorization(s, &bad));
      if (bad) {
        summary->termination_type = CERES_HIP_FAILURE;
        snprintf(summary->message, sizeof(summary->message), "F^T F + D^2 is not positive definite.");
        return 0;
      }
    }
    // On the fused path the factorization flag of the preconditioner blocks is not read back here
    // (a host round trip in front of CG): the CG init kernel looks at it and starts in kCgSetupFailed.
    const bool defer_check = s->path == CERES_HIP_PATH_BAL;
    if (pre != CERES_HIP_IDENTITY && !spse_pre) {
      TRY(op_preconditioner(s, pre, s->precond, true));
      if (!defer_check) {
        TRY(check_factorization(s, &bad));
        if (bad) {  // Preconditioner::Update returned false, :113-121
          summary->termination_type = CERES_HIP_FAILURE;
          snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
          return 0;
        }
      }
      s->precond_valid = true;
    }
    if (s->rhs_reduce_pending) {  // nobody took the deferred all-reduce along (cannot happen on the paths that set the flag; kept as a guard)
      TRY(allreduce(s, s->rhs_f, size_t(h.num_cols_f)));
      s->rhs_reduce_pending = false;
    }
    TRY(rec(s, 4));
    CgSpec spec;
    spec.rhs = s->rhs_f;  // CG only reads it: no copy into cg_rhs
    spec.n = h.num_cols_f;
    spec.n_local = 0;  // camera space is replicated: no inner product crosses ranks
    const int* status = &s->cg.S->status;
    spec.apply = [s, status](const double* in, double* out) { return op_sx(s, in, out, status); };
    if (s->path == CERES_HIP_PATH_BAL)
      spec.apply_dot = [s, status](const double* in, double* out, double* pq, int* n_pq, const double** extra) {
        *extra = nullptr;
        return op_sx(s, in, out, status, pq, n_pq);
      };
    spec.first_block = h.nelim;
    spec.nblocks = h.ncb - h.nelim;
    spec.col_begin = h.num_cols_e;
    spec.diag_off = s->G.diag_off_f;
    spec.blocks = (pre == CERES_HIP_IDENTITY || spse_pre) ? nullptr : s->precond;
    if (cg_tail_possible(s) && spec.blocks) {
      // a camera space this small: the S.x pass finishes the CG iteration itself (one launch instead of four)
      spec.iteration = [s, status](int it) -> int {
        if (ensure_packed(s)) return -1;
        BalArgs A = bal_args(s);
        if (!s->ops->sx_runs_pipelined(A)) return 0;
        CgBuffers& B = s->cg;
        A.x_f = B.p;
        A.status = status;
        A.tail.enabled = 1; A.tail.it = it; A.tail.ticket = s->d_cg_ticket;
        A.tail.x = B.x; A.tail.r = B.r; A.tail.p = B.p; A.tail.z = B.z; A.tail.rhs = B.rhs;
        A.tail.blocks = s->precond; A.tail.D_f = s->D ? s->D + s->hs.num_cols_e : nullptr; A.tail.S = B.S;
        if (hipError_t e = s->ops->fused(kBalSx, A, true, s->fused_grid, s->stream); e != hipSuccess) {
          fail(s, CERES_HIP_E_HIP, "S.x + CG iteration launch: %s", hipGetErrorString(e));
          return -1;
        }
        return 1;
      };
    }
    if (defer_check && spec.blocks) spec.setup_fail = s->d_fail_flag;
    if (spse_pre)  // tolerance 0: the preconditioner must stay fixed during CG (:178-186)
      spec.precondition = [s, spse_iters, status](const double* in, double* out) { return op_spse_apply(s, in, out, spse_iters, 0.0, status); };
    if (s->opt.use_spse_initialization) {  // :97-111
      TRY(op_spse_apply(s, s->rhs_f, s->cg.x, spse_iters, s->opt.spse_tolerance, nullptr));
      spec.x0_nonzero = true;
    }
    s->spec_tail_done = false;
    if (s->lm_speculate && s->path == CERES_HIP_PATH_BAL && h.num_cols_f > 0) {
      spec.before_poll = [s, x]() {
        s->gate_on_cg_status = true;
        const int rc = op_back_substitute(s, s->cg.x, x);
        s->gate_on_cg_status = false;
        if (rc) return rc;
        // sharded: {finite-step flag, this rank's share of the model cost} summed over ranks into the read-back image, in the same tail
        // (the exchange runs whether CG has ended or not — a handshake per epoch on every rank, p2p.h; its sums only mean something once it has)
        if (s->world > 1) {
          if (s->mailbox && s->d_h_pinned) {   // ... and the poll's read-back (the whole image, stamped): one launch for the two
            s->tail_mailbox_seq = ++s->mailbox_seq;
            HIP_TRY(s, LaunchCollectScalarsExchange(s->d_nonfinite, s->scalar_partials, s->backsub_cost_parts, s->scalar_partials + kSumsOffset,
                                                    next_exchange(s), s->stream, s->scalar_partials, kReadbackDoubles, s->d_h_pinned, s->d_stamp,
                                                    s->tail_mailbox_seq));
          } else {
            HIP_TRY(s, LaunchCollectScalarsExchange(s->d_nonfinite, s->scalar_partials, s->backsub_cost_parts, s->scalar_partials + kSumsOffset,
                                                    next_exchange(s), s->stream));
          }
        }
        s->spec_tail_done = true;   // (the poll that follows copies the partial sums and flags together with the CG scalars)
        return 0;
      };
    }
    TRY(run_cg(s, spec, q_tol, r_tol, summary));
    TRY(rec(s, 5));
    if (summary->termination_type != CERES_HIP_FAILURE && summary->termination_type != CERES_HIP_FATAL_ERROR && !s->spec_tail_done)
      TRY(op_back_substitute(s, s->cg.x, x));
    TRY(rec(s, 6));
    return 0;
  }
  // CgnrSolver::SolveImpl, I/cgnr_solver.cc:146-207
  const int pre = s->opt.preconditioner_type;
  TRY(rec(s, 3));
  if (s->path == CERES_HIP_PATH_BAL) {
    s->merge_step_reduce = s->world > 1 && !camera_blocks_exchange(s);   // (a choice every rank makes alike: the verdict was agreed on)
    s->D_int_valid = false;
    const int rc_setup = op_cgnr_setup_bal(s, pre == CERES_HIP_JACOBI, s->cg_rhs, s->precond);
    s->merge_step_reduce = false;
    TRY(rc_setup);
    TRY(ensure_D_int(s));  // (IDENTITY: the set-up kernel forms no point blocks and leaves D where it is)
  } else {
    if (pre == CERES_HIP_JACOBI) TRY(op_preconditioner(s, pre, s->precond, true));
    TRY(op_jtb(s, s->cg_rhs));
  }
  const bool defer_check = s->path == CERES_HIP_PATH_BAL;  // see the ITERATIVE_SCHUR branch
  if (pre == CERES_HIP_JACOBI) {
    if (!defer_check) {
      bool bad = false;
      TRY(check_factorization(s, &bad));
      if (bad) {
        summary->termination_type = CERES_HIP_FAILURE;
        snprintf(summary->message, sizeof(summary->message), "Preconditioner update failed.");
        return 0;
      }
    }
    s->precond_valid = true;
  }
  TRY(rec(s, 4));
  CgSpec spec;
  spec.n = h.num_cols;
  spec.n_local = h.num_cols_e;  // sharded: first nelim column blocks are this rank's points
  const int* status = &s->cg.S->status;
  spec.apply = [s, status](const double* in, double* out) { return op_jtjx(s, in, out, status, nullptr, nullptr, nullptr, true); };
  if (s->path == CERES_HIP_PATH_BAL) {
    spec.apply_dot = [s, status](const double* in, double* out, double* pq, int* n_pq, const double** extra) {
      return op_jtjx(s, in, out, status, pq, n_pq, extra, true);
    };
    // sharded: the fused iteration needs the shard to be exactly the blocks in front of the 9-wide camera blocks (it is, on this
    // path: points then cameras) so that cg_update's two workgroup ranges are the shard and the replicated part
    spec.shard_fused = s->world > 1 && s->nine_wide_from == h.nelim && h.ncb > h.nelim;
  }
  spec.first_block = 0;
  spec.nblocks = h.ncb;
  spec.n_local_blocks = h.nelim;
  spec.col_begin = 0;
  spec.diag_off = s->G.diag_off_all;
  spec.blocks = pre == CERES_HIP_JACOBI ? s->precond : nullptr;
  if (defer_check && spec.blocks) spec.setup_fail = s->d_fail_flag;
  s->spec_tail_done = false;
  if (s->lm_speculate && s->lm_negate_in_solve && s->path == CERES_HIP_PATH_BAL) {
    spec.before_poll = [s, x]() {
      const HostStructure& hh = s->hs;
      if (!s->nonfinite_clean) HIP_TRY(s, hipMemsetAsync(s->d_nonfinite, 0, sizeof(int), s->stream));
      s->nonfinite_clean = false;
      HIP_TRY(s, LaunchCgnrModelCost(s->cg.x, s->cg_rhs, s->cg.r, s->D, 0, hh.num_cols, s->scalar_partials, &s->spec_cgnr_parts, s->stream,
                                     x, s->d_nonfinite, &s->cg.S->status, point_perm(s)));
      s->spec_tail_done = true;   // (the poll that follows copies the partial sums and flags together with the CG scalars)
      return 0;
    };
  }
  TRY(run_cg(s, spec, q_tol, r_tol, summary));
  TRY(rec(s, 5));
  // LM step: the model-cost kernel reads the solution anyway and writes the negated step (no copy-out, no separate negation pass)
  if (s->lm_negate_in_solve) s->lm_cgnr_copy_pending = true;
  else TRY(copy_out_cgnr_solution(s, x));
  TRY(rec(s, 6));
  return 0;
}

void collect_timing(ceres_hip_solver* s) {
  ceres_hip_solve_timing& t = s->timing;
  if (!s->timing_enabled) {   // no events were recorded: the call's wall-clock time is all there is
    const int apps = t.operator_applications;
    t = ceres_hip_solve_timing{};
    t.operator_applications = apps;
    t.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s->call_t0).count();
    return;
  }
  t.upload_ms = elapsed(s->ev[0], s->ev[1]);
  t.pack_ms = elapsed(s->ev[1], s->ev[2]);
  t.setup_ms = elapsed(s->ev[2], s->ev[3]);
  t.preconditioner_ms = elapsed(s->ev[3], s->ev[4]);
  t.cg_ms = elapsed(s->ev[4], s->ev[5]);
  t.back_substitute_ms = elapsed(s->ev[5], s->ev[6]);
  t.download_ms = elapsed(s->ev[6], s->ev[7]);
  t.total_ms = elapsed(s->ev[0], s->ev[7]);
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int ceres_hip_abi_version(void) { return CERES_HIP_ABI_VERSION; }

int ceres_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

const char* ceres_hip_last_error(const ceres_hip_solver* s) { return s ? s->err.c_str() : g_create_error.c_str(); }

ceres_hip_solver* ceres_hip_create(const ceres_hip_options* o) {
  if (!o) { fail(nullptr, CERES_HIP_E_INVALID, "options == NULL"); return nullptr; }
  if (o->solver_type != CERES_HIP_CGNR && o->solver_type != CERES_HIP_ITERATIVE_SCHUR && o->solver_type != CERES_HIP_DENSE_SCHUR) {
    fail(nullptr, CERES_HIP_E_UNSUPPORTED, "solver_type %d is not CGNR, ITERATIVE_SCHUR or DENSE_SCHUR", o->solver_type);
    return nullptr;
  }
  const int pre = o->preconditioner_type;
  const bool pre_ok = o->solver_type == CERES_HIP_DENSE_SCHUR ? true  // a direct solver: the field is ignored
                      : o->solver_type == CERES_HIP_CGNR ? (pre == CERES_HIP_IDENTITY || pre == CERES_HIP_JACOBI)
                                                       : (pre == CERES_HIP_IDENTITY || pre == CERES_HIP_JACOBI || pre == CERES_HIP_SCHUR_JACOBI ||
                                                          pre == CERES_HIP_SCHUR_POWER_SERIES_EXPANSION);
  if (!pre_ok) {  // CgnrSolver's ctor LOG(FATAL)s on the same condition, I/cgnr_solver.cc:119-128
    fail(nullptr, CERES_HIP_E_UNSUPPORTED, "preconditioner_type %d is not available for solver_type %d", pre, o->solver_type);
    return nullptr;
  }
  if (o->use_explicit_schur_complement) {
    if (o->solver_type != CERES_HIP_ITERATIVE_SCHUR) {
      fail(nullptr, CERES_HIP_E_INVALID, "use_explicit_schur_complement applies to ITERATIVE_SCHUR");
      return nullptr;
    }
    if (pre != CERES_HIP_SCHUR_JACOBI || o->use_spse_initialization) {  // CHECK_EQ(preconditioner_type, SCHUR_JACOBI), I/schur_complement_solver.cc:353
      fail(nullptr, CERES_HIP_E_UNSUPPORTED, "Only SCHUR_JACOBI is supported with use_explicit_schur_complement");
      return nullptr;
    }
  }
  if (o->jacobian_storage != 0 && o->jacobian_storage != 1) {
    fail(nullptr, CERES_HIP_E_INVALID, "jacobian_storage must be 0 (fp64) or 1 (fp32 tiles)");
    return nullptr;
  }
  if (o->max_num_iterations < 1 || o->min_num_iterations < 0) {
    fail(nullptr, CERES_HIP_E_INVALID, "bad iteration limits");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fail(nullptr, CERES_HIP_E_NODEVICE, "no HIP device is visible; this library has no CPU fallback");
    return nullptr;
  }
  if (o->device < 0 || o->device >= ndev) { fail(nullptr, CERES_HIP_E_INVALID, "device %d out of range", o->device); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, o->device) != hipSuccess) { fail(nullptr, CERES_HIP_E_HIP, "hipGetDeviceProperties failed"); return nullptr; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fail(nullptr, CERES_HIP_E_NODEVICE, "device %d is %s; the kernels are built for gfx950 only", o->device, prop.gcnArchName);
    return nullptr;
  }
  auto s = std::make_unique<ceres_hip_solver>();
  s->opt = *o;
  s->num_cus = prop.multiProcessorCount;
  if (hipSetDevice(o->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
    fail(nullptr, CERES_HIP_E_HIP, "could not create a stream on device %d", o->device);
    return nullptr;
  }
  for (auto& e : s->ev) (void)hipEventCreate(&e);
  if (hipHostMalloc(reinterpret_cast<void**>(&s->h_pinned), sizeof(double) * (kReadbackDoubles + 8), hipHostMallocMapped) != hipSuccess) {
    fail(nullptr, CERES_HIP_E_HIP, "hipHostMalloc failed");
    return nullptr;
  }
  memset(s->h_pinned, 0, sizeof(double) * (kReadbackDoubles + 8));
  s->h_scalars = reinterpret_cast<CgScalars*>(s->h_pinned + kScalarsOffset);
  s->h_stamp = reinterpret_cast<unsigned long long*>(s->h_pinned + kReadbackDoubles);
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&s->d_h_pinned), s->h_pinned, 0) == hipSuccess)
    s->d_stamp = reinterpret_cast<unsigned long long*>(s->d_h_pinned + kReadbackDoubles);
  else { s->d_h_pinned = nullptr; (void)hipGetLastError(); }   // (no mapping: read_back falls back to the copy + synchronise)
  return s.release();
}

void ceres_hip_destroy(ceres_hip_solver* s) {
  if (!s) return;
  (void)hipSetDevice(s->opt.device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->comm) (void)ncclCommDestroy(s->comm);
  for (int q = 0; q < kP2pMaxWorld; ++q) if (s->p2p_opened[q]) (void)hipIpcCloseMemHandle(s->p2p_opened[q]);
  if (s->p2p_base) (void)hipFree(s->p2p_base);
  if (s->h_comm_error) (void)hipHostFree(s->h_comm_error);
  if (s->d_comm_error_seen) (void)hipFree(s->d_comm_error_seen);
  free_all(s);
  if (s->copy_done) (void)hipEventDestroy(s->copy_done);
  if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
  if (s->h_pinned) (void)hipHostFree(s->h_pinned);
  for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// A few doubles summed over ranks from the host (agreements at set-up time: ranks reach them at different moments — host-side planning
// takes tenths of a second —, so the exchange's time-out is raised for the call).
static int allreduce_host_doubles(ceres_hip_solver* s, double* v, int n) {
  double* d = nullptr;
  TRY(dev_alloc(s, &d, size_t(n)));
  HIP_TRY(s, hipMemcpyAsync(d, v, sizeof(double) * n, hipMemcpyHostToDevice, s->stream));
  const double keep = s->p2p_timeout_s;
  s->p2p_timeout_s = std::max(keep, 120.0);
  const int rc = allreduce(s, d, size_t(n));
  s->p2p_timeout_s = keep;
  TRY(rc);
  HIP_TRY(s, hipMemcpyAsync(v, d, sizeof(double) * n, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return check_comm_error(s);
}

static int set_structure_impl(ceres_hip_solver* s, const ceres_hip_block_structure* bs);
// (the host-side analysis and the tile plan allocate with the structure's sizes: no C++ exception may cross the C boundary)
int ceres_hip_set_structure(ceres_hip_solver* s, const ceres_hip_block_structure* bs) {
  if (!s || !bs) return CERES_HIP_E_INVALID;
  try {
    return set_structure_impl(s, bs);
  } catch (const std::bad_alloc&) {
    return fail(s, CERES_HIP_E_INVALID, "out of host memory while analysing the block structure");
  } catch (const std::exception& ex) {
    return fail(s, CERES_HIP_E_INVALID, "block structure analysis failed: %s", ex.what());
  }
}
static int set_structure_impl(ceres_hip_solver* s, const ceres_hip_block_structure* bs) {
  HIP_TRY(s, hipSetDevice(s->opt.device));
  if (s->have_structure) return fail(s, CERES_HIP_E_INVALID, "structure already set: one instance sees one sparsity (I/linear_solver.h:137-142)");
  const int nelim = s->opt.num_eliminate_blocks;
  std::string e = AnalyzeStructure(*bs, nelim, &s->hs);
  if (!e.empty()) return fail(s, CERES_HIP_E_INVALID, "invalid block structure: %s", e.c_str());
  HostStructure& h = s->hs;
  if (is_schur(s)) {
    if (nelim <= 0) return fail(s, CERES_HIP_E_INVALID, "ITERATIVE_SCHUR needs num_eliminate_blocks > 0 (LinearSolver::Create falls back to CGNR otherwise, I/linear_solver.cc:51-73: do that on the host)");
    if (!h.chunks_contiguous) return fail(s, CERES_HIP_E_INVALID, "rows are not ordered for a Schur solver: E rows must come first, grouped by E block (I/reorder_program.cc:278-360)");
  }
  if (s->world > 1 && !h.chunks_contiguous) return fail(s, CERES_HIP_E_INVALID, "sharded runs need the Schur ordering");
  s->nine_wide_from = h.ncb;
  while (s->nine_wide_from > 0 && h.csz[s->nine_wide_from - 1] == 9) --s->nine_wide_from;
  // Schur solvers: no CG vector lives in point space, so the points may be renumbered to fill the tiles (plan.cc); CGNR walks x, r, p,
  // q in tile order and keeps the caller's numbering.  CERES_HIP_REORDER_POINTS=0 switches the renumbering off (A/B measurements).
  {
    const char* e = getenv("CERES_HIP_REORDER_POINTS");
    // more cameras than LDS rows: one workgroup per CU, each with the accumulator rows that fit next to its eight waves' spill strips
    HybridRequest hyb;
    hyb.groups = s->num_cus;
    // (sized for 9-wide cameras; the plan of another width that needs the rows — more cameras than LDS holds — resizes them below)
    hyb.rows = 0;
    hyb.lds_bytes = int64_t(kMaxLdsBytes) - 512;   // the plan sizes the rows for its camera width
    // Schur solvers: no CG vector lives in point space; CGNR: only where its CG vectors can be the caller's with the points renumbered
    BuildBalPlan(h, (e && atoi(e) == 0) ? kReorderNever : (is_schur(s) ? kReorderAlways : kReorderIfContiguous), hyb, &s->plan);
  }
  s->ops = s->plan.eligible ? GetBalOps(s->plan.nr, s->plan.ne, s->plan.nf, s->plan.ns) : nullptr;
  if (s->plan.eligible && (!s->ops || (s->opt.jacobian_storage == 1 && !s->ops->has_f32))) {
    // (fp32 tiles exist for the 9-wide shape only)
    s->plan.eligible = false;
    s->plan.why_not = !s->ops ? "no kernels for this shape" : "option not available for this shape";
  }
  if (s->plan.eligible && (s->plan.ne != 3 || s->plan.nr != 2) && !is_schur(s)) {
    // point blocks of 2 or 4 scalars, rows of 3 or 4 residuals: the Schur solvers' tile passes, one device or sharded (CGNR sees no
    // elimination order — its "points" are whatever is 3 wide — and its vector kernels hold points as 3-vectors)
    s->plan.eligible = false;
    s->plan.why_not = "point blocks that are not 3 wide or rows that are not 2 high: the fused path takes the Schur solvers";
  }
  s->path = (s->plan.eligible && !s->opt.force_generic_path && !s->opt.use_explicit_schur_complement && !(is_dense_schur(s) && s->world > 1)) ? CERES_HIP_PATH_BAL : CERES_HIP_PATH_GENERIC;
  if (s->world > 1) {
    // The kernel path is a property of THIS rank's shard — a shard of a general structure can look like bundle adjustment (every row of
    // it one point cell and one camera cell) while its neighbour's does not — and the two paths issue different sequences of exchanges:
    // ranks that disagreed waited for each other until the time-out (tools/fuzz_multirank.py --generic).  All ranks take the fused
    // path, with the same shape, or none does; a shard whose columns are not points-then-cameras votes against it too.
    const bool fused_ok = s->path == CERES_HIP_PATH_BAL && (is_schur(s) ? s->plan.cameras_contiguous : s->plan.caller_contiguous);
    const double dims[4] = {double(s->plan.nr), double(s->plan.ne), double(s->plan.nf), double(s->plan.ns)};
    double v[10] = {fused_ok ? 0.0 : 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0};   // [9]: the ranks that take part (a loop-back communicator: one)
    for (int i = 0; i < 4; ++i) { v[1 + i] = fused_ok ? dims[i] : 0.0; v[5 + i] = fused_ok ? dims[i] * dims[i] : 0.0; }
    TRY(allreduce_host_doubles(s, v, 10));
    bool same = v[0] == 0.0;
    for (int i = 0; i < 4 && same; ++i) same = v[1 + i] * v[1 + i] == v[9] * v[5 + i];   // (equal iff every rank holds the same number)
    if (!same) s->path = CERES_HIP_PATH_GENERIC;   // (as with force_generic_path: the plan stays, nothing uses it)
  }
  s->dense_from_blocks = is_dense_schur(s) && s->world <= 1;
  if (s->path == CERES_HIP_PATH_GENERIC && h.max_block > kMaxGenericBlock)
    return fail(s, CERES_HIP_E_UNSUPPORTED, "block size %d exceeds the generic kernels' limit of %d", h.max_block, kMaxGenericBlock);
  s->cgnr_internal = !is_schur(s) && s->path == CERES_HIP_PATH_BAL && s->plan.renumbered;
  if (s->world > 1 && s->path == CERES_HIP_PATH_BAL && !(is_schur(s) ? s->plan.cameras_contiguous : s->plan.caller_contiguous))
    return fail(s, CERES_HIP_E_UNSUPPORTED, "sharded <2,3,9> runs need points-then-cameras column order");

  // explicit S: block-sparse (BlockRandomAccessSparseMatrix) on one rank; dense for DENSE_SCHUR and for sharded runs (one all-reduce of S)
  s->sparse_S = s->opt.use_explicit_schur_complement && !is_dense_schur(s) && s->world <= 1;
  const bool dense_S = is_dense_schur(s) || (s->opt.use_explicit_schur_complement && !s->sparse_S);
  if (dense_S && h.num_cols_f > CERES_HIP_MAX_EXPLICIT_SCHUR_COLS)
    return fail(s, CERES_HIP_E_UNSUPPORTED, "a dense Schur complement of %d reduced columns exceeds the limit of %d",
                h.num_cols_f, CERES_HIP_MAX_EXPLICIT_SCHUR_COLS);

  // ---- structure arrays ----
  GenStructure& G = s->G;
  G.nrb = h.nrb; G.ncb = h.ncb; G.nelim = h.nelim; G.nrbe = h.num_row_blocks_e;
  G.num_rows = h.num_rows; G.num_cols = h.num_cols; G.nce = h.num_cols_e; G.ncf = h.num_cols_f;
  int32_t* p32 = nullptr;
  int64_t* p64 = nullptr;
#define UP32(field, vec) TRY(dev_upload(s, &p32, vec)); G.field = p32
#define UP64(field, vec) TRY(dev_upload(s, &p64, vec)); G.field = p64
  UP32(rsz, h.rsz); UP32(rpos, h.rpos); UP32(rptr, h.rptr); UP32(ccol, h.ccol); UP32(cval, h.cval);
  UP32(csz, h.csz); UP32(cpos, h.cpos); UP32(tptr, h.tptr); UP32(trow, h.trow); UP32(tcell, h.tcell);
  UP32(row_block_of, h.row_block_of); UP32(col_block_of, h.col_block_of); UP32(row_e_block, h.row_e_block);
  UP64(diag_off_all, h.diag_off_all); UP64(diag_off_e, h.diag_off_e); UP64(diag_off_f, h.diag_off_f);
  if (h.chunks_contiguous && h.nelim > 0) { UP32(chunk_start, h.chunk_start); UP32(chunk_size, h.chunk_size); }
  {
    std::vector<int32_t> tval(h.ncells), trpos(h.ncells), tinfo(h.ncells);
    for (int t = 0; t < h.ncells; ++t) {
      const int i = h.trow[t], k = h.tcell[t];
      tval[t] = h.cval[k]; trpos[t] = h.rpos[i];
      tinfo[t] = h.rsz[i] | ((i < h.num_row_blocks_e && k == h.rptr[i]) ? 256 : 0);
    }
    UP32(tval, tval); UP32(trpos, trpos); UP32(tinfo, tinfo);
  }
#undef UP32
#undef UP64
  {
    // hints for the grouped generic kernels (kernels_generic.hip): block sizes per part, lanes per group from the average cell count.
    // CERES_HIP_GENERIC_GROUPED=0 keeps the thread-per-scalar kernels (A/B measurements, tests of the old kernels).
    const char* e = getenv("CERES_HIP_GENERIC_GROUPED");
    if (!(e && atoi(e) == 0)) {
      auto lanes_for = [](double avg) { int L = 4; while (L < 64 && L < avg) L *= 2; return L; };
      int64_t cells_e = 0, cells_f = 0;
      for (int j = 0; j < h.ncb; ++j) {
        (j < h.nelim ? G.max_csz_e : G.max_csz_f) = std::max(j < h.nelim ? G.max_csz_e : G.max_csz_f, h.csz[j]);
        (j < h.nelim ? cells_e : cells_f) += h.tptr[j + 1] - h.tptr[j];
      }
      G.max_csz = std::max(G.max_csz_e, G.max_csz_f);
      for (int i = 0; i < h.nrb; ++i) G.max_rsz = std::max(G.max_rsz, h.rsz[i]);
      G.lanes_e = lanes_for(h.nelim > 0 ? double(cells_e) / h.nelim : 0.0);
      G.lanes_f = lanes_for(h.ncb > h.nelim ? double(cells_f) / (h.ncb - h.nelim) : 0.0);
      G.lanes_all = lanes_for(h.ncb > 0 ? double(cells_e + cells_f) / h.ncb : 0.0);
      G.lanes_chunk = (h.chunks_contiguous && h.nelim > 0) ? lanes_for(double(h.num_row_blocks_e) / h.nelim) : 0;
      // heavy column blocks (the F blocks; every block without an elimination order): ITEMS of <= kGenItem cells (device.h)
      const int j0 = h.nelim, nb = h.ncb - h.nelim;
      int max_cells = 0;
      for (int j = j0; j < h.ncb; ++j) max_cells = std::max(max_cells, h.tptr[j + 1] - h.tptr[j]);
      if (nb > 0 && ((h.nelim > 0 ? G.lanes_f : G.lanes_all) == 64 || max_cells > 8 * kGenItem)) {
        std::vector<int32_t> ib, i0, i1, bp(nb + 1, 0);
        for (int q = 0; q < nb; ++q) {
          const int j = j0 + q;
          bp[q] = int32_t(ib.size());
          int t = h.tptr[j];
          const int te = h.tptr[j + 1];
          do {   // (a block without cells still gets one empty item: its block is written, D^2 joins the diagonal)
            int cut = std::min(te, t + kGenItem);
            // the cells a block has inside ONE chunk stay together (the Schur complement's diagonal block couples them)
            while (cut < te && cut > t && h.row_e_block[h.trow[cut]] >= 0 && h.row_e_block[h.trow[cut]] == h.row_e_block[h.trow[cut - 1]]) ++cut;
            ib.push_back(j); i0.push_back(t); i1.push_back(cut);
            t = cut;
          } while (t < te);
        }
        bp[nb] = int32_t(ib.size());
        int32_t* q32 = nullptr;
        TRY(dev_upload(s, &q32, ib)); G.items.block = q32;
        TRY(dev_upload(s, &q32, i0)); G.items.t0 = q32;
        TRY(dev_upload(s, &q32, i1)); G.items.t1 = q32;
        TRY(dev_upload(s, &q32, bp)); G.items.block_ptr = q32;
        G.items.count = int(ib.size()); G.items.first_block = j0; G.items.nblocks = nb;
        double* sc = nullptr;
        TRY(dev_alloc(s, &sc, size_t(ib.size()) * kGenItemValues));
        G.items.scratch = sc;
      }
    }
  }

  // ---- inputs, temporaries, CG state ----
  TRY(dev_alloc(s, &s->own_values, size_t(h.values_extent)));
  TRY(dev_alloc(s, &s->own_b, size_t(h.num_rows)));
  TRY(dev_alloc(s, &s->own_D, size_t(h.num_cols)));
  TRY(dev_alloc(s, &s->own_x, size_t(h.num_cols)));
  TRY(dev_alloc(s, &s->scratch_vec, size_t(h.num_cols) + size_t(h.num_rows)));

  TRY(dev_alloc(s, &s->lm_diag, size_t(h.num_cols)));
  TRY(dev_alloc(s, &s->lm_D, size_t(h.num_cols)));
  // [0, 2 kMaxVecGrid) partial sums, then two int flags in one double: one memset clears both flags, one D2H copy at the
  // end of an LM step brings {model-cost partials, finite-step flag} back
  TRY(dev_alloc(s, &s->scalar_partials, size_t(kReadbackDoubles)));   // [partial sums | flags | CgScalars]: one read-back (poll_scalars)
  s->d_nonfinite = reinterpret_cast<int*>(s->scalar_partials + 2 * kMaxVecGrid);
  s->d_fail_flag = s->d_nonfinite + 1;
  HIP_TRY(s, hipMemsetAsync(s->d_nonfinite, 0, 2 * sizeof(int), s->stream));
  TRY(dev_alloc(s, &s->rhs_f, size_t(h.num_cols_f)));
  if (dense_S) TRY(dev_alloc(s, &s->d_S, std::max<size_t>(1, size_t(h.num_cols_f) * size_t(h.num_cols_f))));
  if (s->sparse_S || s->dense_from_blocks) {
    SchurStorage& Q = s->schur_storage;
    BuildSchurStorage(h, &Q);
    SchurPairs& P = s->schur_pairs;
    P.npairs = int(Q.pair_i.size());
    int32_t* q32 = nullptr; int64_t* q64 = nullptr;
    TRY(dev_upload(s, &q32, Q.pair_i)); P.pair_i = q32;
    TRY(dev_upload(s, &q32, Q.pair_j)); P.pair_j = q32;
    TRY(dev_upload(s, &q32, Q.row_ptr)); P.row_ptr = q32;
    TRY(dev_upload(s, &q32, Q.col_ptr)); P.col_ptr = q32;
    TRY(dev_upload(s, &q32, Q.col_pair)); P.col_pair = q32;
    TRY(dev_upload(s, &q64, Q.pair_off)); P.pair_off = q64;
    TRY(dev_upload(s, &q64, Q.trip_ptr)); P.trip_ptr = q64;
    TRY(dev_upload(s, &q32, Q.trip_e)); P.trip_e = q32;
    TRY(dev_upload(s, &q32, Q.trip_k1)); P.trip_k1 = q32;
    TRY(dev_upload(s, &q32, Q.trip_k2)); P.trip_k2 = q32;
    TRY(dev_upload(s, &q32, Q.cell_row)); P.cell_row = q32;
    P.n_items = int(Q.item_pair.size());
    P.total_values = Q.num_values();
    TRY(dev_upload(s, &q32, Q.item_pair)); P.item_pair = q32;
    TRY(dev_upload(s, &q32, Q.pair_item_ptr)); P.pair_item_ptr = q32;
    TRY(dev_upload(s, &q64, Q.item_t0)); P.item_t0 = q64;
    TRY(dev_upload(s, &q64, Q.item_t1)); P.item_t1 = q64;
    TRY(dev_upload(s, &q64, Q.item_off)); P.item_off = q64;
    { double* sc = nullptr; TRY(dev_alloc(s, &sc, std::max<size_t>(1, size_t(Q.scratch_values())))); P.scratch = sc; }
    TRY(dev_alloc(s, s->sparse_S ? &s->d_S : &s->d_Sblk, std::max<size_t>(1, size_t(Q.num_values()))));
    if (s->dense_from_blocks && s->path == CERES_HIP_PATH_BAL) TRY(dev_alloc(s, &s->etei_dense, size_t(h.diag_off_e.back())));
  }
  const int64_t cg_n = is_schur(s) ? h.num_cols_f : h.num_cols;
  TRY(dev_alloc(s, &s->cg.x, size_t(cg_n)));
  TRY(dev_alloc(s, &s->cg.r, size_t(cg_n)));
  TRY(dev_alloc(s, &s->cg.p, size_t(cg_n)));
  TRY(dev_alloc(s, &s->cg.z, size_t(cg_n) + 8));  // slack: a sharded operator all-reduces one scalar behind the camera part (bal_scatter)
  TRY(dev_alloc(s, &s->cg_rhs, size_t(cg_n)));
  TRY(dev_alloc(s, &s->cg.partials, size_t(4 * kMaxVecGrid)));
  TRY(dev_alloc(s, &s->cg.comm, 4));
  TRY(dev_alloc(s, &s->cg_pq_parts, size_t(kMaxPqParts)));
  { const char* e = getenv("CERES_HIP_CG_FUSED"); s->cg_fused = !(e && atoi(e) == 0); }
  { const char* e = getenv("CERES_HIP_SPECULATE"); s->speculate = !(e && atoi(e) == 0); }
  s->cg.S = reinterpret_cast<CgScalars*>(s->scalar_partials + kScalarsOffset);
  HIP_TRY(s, hipMemsetAsync(s->cg.S, 0, sizeof(CgScalars), s->stream));
  // (+ 18 doubles per F block: room for the step's other camera-space sums right behind the blocks, see merged_layout)
  TRY(dev_alloc(s, &s->precond, size_t(is_schur(s) ? h.diag_off_f.back() : h.diag_off_all.back()) + 2 * size_t(h.num_cols_f)));
  if (is_schur(s)) {
    TRY(dev_alloc(s, &s->ftf_inv, size_t(h.diag_off_f.back())));
    TRY(dev_alloc(s, &s->spse_a, size_t(h.num_cols_f)));
    TRY(dev_alloc(s, &s->spse_b, size_t(h.num_cols_f)));
  }

  if (s->path == CERES_HIP_PATH_BAL) {
    BalPlan& P = s->plan;
    TRY(dev_upload(s, &s->d_slot_epos, P.slot_epos));
    TRY(dev_upload(s, &s->d_slot_fpos, P.slot_fpos));
    TRY(dev_upload(s, &s->d_slot_bpos, P.slot_bpos));
    if (P.ns > 0) {
      for (int q = 0; q < kMaxSharedCellsPerRow; ++q) { TRY(dev_upload(s, &s->d_slot_hpos[q], P.slot_hpos[q])); TRY(dev_upload(s, &s->d_slot_hdesc[q], P.slot_hdesc[q])); }
      std::vector<int64_t> cfo(P.n_cameras);
      for (int c = 0; c < P.n_cameras; ++c) cfo[c] = h.diag_off_f[P.cam_block[c] - h.nelim];
      TRY(dev_upload(s, &s->d_cam_foff, cfo));
    }
    TRY(dev_upload(s, &s->d_slot_cam, P.slot_word));  // camera | accumulator row << kSlotCamBits
    if (!P.xhot_cam.empty()) TRY(dev_upload(s, &s->d_xhot_cam, P.xhot_cam));
    TRY(dev_upload(s, &s->d_tile_pt0, P.tile_pt0));
    TRY(dev_upload(s, &s->d_slot_seg, P.slot_seg));
    TRY(dev_upload(s, &s->d_tile_kind, P.tile_kind));
    TRY(dev_alloc(s, &s->d_cg_ticket, 1));
    HIP_TRY(s, hipMemsetAsync(s->d_cg_ticket, 0, sizeof(unsigned int), s->stream));
    TRY(dev_upload(s, &s->d_long_ptr, P.long_ptr));
    TRY(dev_upload(s, &s->d_round_ptr, P.round_ptr));
    s->d_round_word = nullptr;
    if (!P.round_word.empty()) {
      TRY(dev_upload(s, &s->d_round_word, P.round_word));
      TRY(dev_upload(s, &s->d_seq_ptr, P.seq_ptr));
      TRY(dev_upload(s, &s->d_round_flag, P.round_flag));
    }
    TRY(dev_upload(s, &s->d_tile_aux, P.tile_aux));
    TRY(dev_upload(s, &s->d_pt_pos, P.pt_pos));
    TRY(dev_upload(s, &s->d_cam_pos, P.cam_pos));
    TRY(dev_upload(s, &s->d_cam_ptr, P.cam_ptr));
    TRY(dev_upload(s, &s->d_cam_fpos, P.cam_fpos));
    TRY(dev_upload(s, &s->d_cam_slot, P.cam_slot));
    {
      int32_t *ic = nullptr, *ib = nullptr, *ie = nullptr;
      TRY(dev_upload(s, &ic, P.item_cam));
      TRY(dev_upload(s, &ib, P.item_begin));
      TRY(dev_upload(s, &ie, P.item_end));
      s->cam_items.cam = ic; s->cam_items.begin = ib; s->cam_items.end = ie;
      s->cam_items.count = int(P.item_cam.size());
      s->cam_items.observations = P.n_cam_cells;
      TRY(dev_upload(s, &s->d_cam_item_ptr, P.cam_item_ptr));
      {
        int most = 0;
        for (int c = 0; c < P.n_cameras; ++c) most = std::max(most, P.cam_item_ptr[c + 1] - P.cam_item_ptr[c]);
        s->cam_items_few = most <= 4 && P.n_cameras >= 1024;
        // the sharded exchange of the cameras' sums: four cameras per wavefront where the AVERAGE camera has a handful of items (a popular
        // camera's hundred items are a hundred independent loads for its lane; one workgroup per camera is one exchange round trip per camera)
        // (measured cross-over: 1778 cameras 37 us against 14 us for a workgroup per camera, 50 000 cameras 111 against 267: profiles/r06t_*)
        s->cam_exchange_few = int64_t(P.item_cam.size()) <= int64_t(4) * P.n_cameras && P.n_cameras >= 8192;
        { const char* e = getenv("CERES_HIP_CAM_EXCHANGE_FEW"); if (e) s->cam_exchange_few = atoi(e) != 0; }   // (A/B switch)
      }
      TRY(dev_alloc(s, &s->d_cam_parts, size_t(P.item_cam.size()) * s->ops->cam_part));
      if (s->world > 1) TRY(dev_alloc(s, &s->d_cam_packed, size_t(std::max(1, P.n_cameras)) * s->ops->cam_part));
    }
    std::vector<int64_t> pdo(P.n_points), cdo(P.n_cameras);
    for (int p = 0; p < P.n_points; ++p) pdo[p] = h.diag_off_all[P.pt_block[p]];
    for (int c = 0; c < P.n_cameras; ++c) cdo[c] = h.diag_off_all[P.cam_block[c]];
    TRY(dev_upload(s, &s->d_pt_diag_off, pdo));
    TRY(dev_upload(s, &s->d_cam_diag_off, cdo));
    if (h.nelim > 0) {  // where each (internally numbered) point's dense 3x3 block sits in the E-block store
      std::vector<int64_t> peo(P.n_points);
      for (int p = 0; p < P.n_points; ++p) peo[p] = h.diag_off_e[P.pt_block[p]];
      TRY(dev_upload(s, &s->d_pt_eoff, peo));
    }
    const size_t n_slots = size_t(P.n_tiles) * kTile;
    if (s->opt.jacobian_storage == 1) TRY(dev_alloc(s, &s->d_Jf, n_slots * 6));   // (has_f32 shapes only: checked above)
    else TRY(dev_alloc(s, &s->d_J, size_t(P.n_tiles) * s->ops->tile_pitch));
    TRY(dev_alloc(s, &s->d_bt, n_slots * s->ops->b_pairs));
    TRY(dev_alloc(s, &s->d_Mo, size_t(s->ops->mo_pitch) * n_slots));
    TRY(dev_alloc(s, &s->etei, size_t(P.n_points) * s->ops->etei_pitch));
    const size_t n9 = size_t(P.nf) * P.n_cameras + size_t(P.ns);   // accumulator entries: the cameras' scalars, then the strip
    s->lds_mode = P.cameras_in_lds;
    // (cameras beyond LDS: four workgroups per CU for the passes that scatter nothing, but fewer than the 2 kMaxVecGrid partial sums the
    // LM step's read-back image holds — back-substitution leaves the model cost's partials there, one per workgroup (+ one for rows outside
    // the tiles); at 1024 a separate pass over J formed it: 165 us of a 2.2 ms step on one rank's eighth of synthetic10M, profiles/r06e_*)
    s->fused_grid = s->lds_mode ? s->num_cus : std::min(s->num_cus * 4, 2 * kMaxVecGrid - 2);
    { const char* e = getenv("CERES_HIP_FUSED_GRID"); if (e && atoi(e) > 0 && s->lds_mode) s->fused_grid = std::min(s->fused_grid, atoi(e)); }   // (experiments)
    const int64_t tiles_per_wg = 512 / kTile;
    s->fused_grid = int(std::max<int64_t>(1, std::min<int64_t>(s->fused_grid, (P.n_tiles + tiles_per_wg - 1) / tiles_per_wg)));
    s->fused_grid = std::min(s->fused_grid, kMaxPqParts - kMaxVecGrid);  // one p.q partial per workgroup must fit cg_pq_parts (bal_scatter)
    TRY(dev_alloc(s, &s->d_partials, s->lds_mode ? size_t(s->fused_grid) * n9 : 1));
    TRY(dev_alloc(s, &s->d_global_acc, n9));
    TRY(dev_alloc(s, &s->d_zbuf, s->lds_mode ? size_t(1) : size_t(P.z_ring_rows) * P.nf));  // ring of ONE chunk's F^T z rows
    if (!P.mo_index.empty()) TRY(dev_upload(s, &s->d_mo_index, P.mo_index));   // where kInit puts a slot's M_o record (plan.cc)
    if (!s->lds_mode) {
      TRY(dev_upload(s, &s->d_tile_zbase, P.tile_zbase));
      if (P.hybrid) TRY(dev_upload(s, &s->d_grp_tile_ptr, P.grp_tile_ptr));
      int32_t *uc = nullptr, *ub = nullptr, *ue = nullptr, *us = nullptr, *zs = nullptr;
      TRY(dev_upload(s, &uc, P.zu_cam)); TRY(dev_upload(s, &ub, P.zu_begin)); TRY(dev_upload(s, &ue, P.zu_end));
      TRY(dev_upload(s, &us, P.zu_shared)); TRY(dev_upload(s, &zs, P.zc_slot));
      s->zunits.cam = uc; s->zunits.begin = ub; s->zunits.end = ue; s->zunits.shared = us; s->zunits.slot = zs;
      // a chunk's tile pass: every wave should see a few tiles (the pipelined kernel has a prologue), at most 4 workgroups per CU
      int64_t chunk_tiles = 0;
      for (size_t k = 0; k + 1 < P.zc_tile_ptr.size(); ++k) chunk_tiles = std::max<int64_t>(chunk_tiles, P.zc_tile_ptr[k + 1] - P.zc_tile_ptr[k]);
      // (the pipelined kernel keeps 8 waves per CU resident: one workgroup per CU, no second round with a ragged tail)
      s->chunk_grid = P.hybrid ? P.hyb_groups : int(std::max<int64_t>(1, std::min<int64_t>(s->num_cus, (chunk_tiles + 15) / 16)));
    }
    TRY(dev_alloc(s, &s->d_camsq, n9));
    if (P.ns > 0) TRY(dev_alloc(s, &s->d_strip_parts, size_t(s->fused_grid) * size_t(P.ns * (P.ns + 1) / 2)));
    if (s->cgnr_internal) TRY(dev_alloc(s, &s->D_int, size_t(h.num_cols_e)));
    if (s->world > 1 && (is_schur(s) ? (P.cameras_contiguous && P.cam_base == 0) : P.caller_contiguous) && int64_t(n9) == int64_t(h.num_cols_f)) {
      // one all-reduce per step for the camera-space sums: rhs and the column norms live right behind the preconditioner blocks
      s->merged_layout = true;
      if (is_schur(s)) {
        s->rhs_f = s->precond + h.diag_off_f.back();
        s->d_camsq = s->rhs_f + n9;
      } else {
        s->cgnr_rhs_tail = s->precond + h.diag_off_all.back();
      }
    }
    if (P.n_rem_rows > 0) {
      // the remainder rows as a structure of their own: compact row space, cells and transpose lists rebased, columns shared with G
      // (P.rem_list: their row blocks, ascending — trailing, or anywhere among the observation rows when there is no elimination order)
      const int nr = P.n_rem_rows;
      GenStructure& R = s->GR;
      R = s->G;
      R.items = GenItems();   // (G's items and group widths describe G's transpose lists, not the remainder's)
      R.lanes_e = R.lanes_f = R.lanes_all = R.lanes_chunk = 0;
      R.nrb = nr; R.nrbe = 0;
      std::vector<int32_t> rsz(nr), rpos(nr), rptr(nr + 1, 0), ccol, cval, rbo, row_map, tptr(h.ncb + 1, 0), trow, tcell;
      std::vector<int32_t> rem_of(h.nrb, -1), cell_of(h.rptr[h.nrb], -1);   // row block -> remainder row, cell -> remainder cell
      int nscal = 0;
      for (int i = 0; i < nr; ++i) {
        const int ro = P.rem_list[i];
        rem_of[ro] = i;
        rsz[i] = h.rsz[ro];
        rpos[i] = nscal;
        for (int r = 0; r < h.rsz[ro]; ++r) { rbo.push_back(i); row_map.push_back(h.rpos[ro] + r); }
        nscal += h.rsz[ro];
        for (int k = h.rptr[ro]; k < h.rptr[ro + 1]; ++k) { cell_of[k] = int32_t(ccol.size()); ccol.push_back(h.ccol[k]); cval.push_back(h.cval[k]); }
        rptr[i + 1] = int32_t(ccol.size());
      }
      s->rem_rows = nscal;
      R.num_rows = nscal;
      s->rem_b0 = P.rem_row0 >= 0 ? h.rpos[P.rem_row0] : -1;   // a window of b (the rows trail, back to back), or gathered
      for (int j = 0; j < h.ncb; ++j) {  // the transpose lists are in row order: the remainder's entries keep theirs
        tptr[j] = int32_t(trow.size());
        for (int t = h.tptr[j]; t < h.tptr[j + 1]; ++t)
          if (rem_of[h.trow[t]] >= 0) { trow.push_back(rem_of[h.trow[t]]); tcell.push_back(cell_of[h.tcell[t]]); }
      }
      tptr[h.ncb] = int32_t(trow.size());
      int32_t* q = nullptr;
      TRY(dev_upload(s, &q, rsz)); R.rsz = q;
      TRY(dev_upload(s, &q, rpos)); R.rpos = q;
      TRY(dev_upload(s, &q, rptr)); R.rptr = q;
      TRY(dev_upload(s, &q, ccol)); R.ccol = q;
      TRY(dev_upload(s, &q, cval)); R.cval = q;
      TRY(dev_upload(s, &q, rbo)); R.row_block_of = q;
      TRY(dev_upload(s, &q, tptr)); R.tptr = q;
      TRY(dev_upload(s, &q, trow)); R.trow = q;
      TRY(dev_upload(s, &q, tcell)); R.tcell = q;
      R.row_e_block = nullptr;
      TRY(dev_upload(s, &s->d_cam_block, P.cam_block));
      TRY(dev_alloc(s, &s->rem_tmp, size_t(s->rem_rows)));
      if (s->rem_b0 < 0) {
        TRY(dev_upload(s, &s->d_rem_row_map, row_map));
        TRY(dev_alloc(s, &s->rem_b, size_t(s->rem_rows)));
      }
      TRY(dev_alloc(s, &s->rem_blocks, size_t(P.nf) * P.nf * P.n_cameras));   // raw F^T F of the remainder rows, one nf x nf block per camera
    }
    {
      const char* e = getenv("CERES_HIP_COOP");  // 0: per-lane strided point-space accesses in JtJx instead of the cooperative ones
      s->bal_flags = (e && atoi(e) == 0) ? 1 : 0;
      const char* w = getenv("CERES_HIP_TILE_WALK");  // "blocked": a workgroup of the pipelined kernels takes one contiguous run of tiles (experiment)
      if (w && strcmp(w, "blocked") == 0) s->bal_flags |= 2;
      // tiles that fit the Infinity Cache (the bound launch_stream uses for plain LOADS): the first pass writes them with plain stores
      // — opt-in (CERES_HIP_PLAIN_TILE_STORES=1): measured level or behind non-temporal stores on every shape tried (one rank's eighth of
      // Venice 0.352 against 0.346 ms per step, Ladybug and Dubrovnik level; profiles/r06u_prefetch_ab.jsonl)
      const char* ps = getenv("CERES_HIP_PLAIN_TILE_STORES");
      const int64_t tile_bytes = s->opt.jacobian_storage == 1 ? int64_t(P.n_tiles) * kTile * 96 : int64_t(P.n_tiles) * s->ops->tile_pitch * 16;   // (what launch_stream counts)
      if (ps && atoi(ps) != 0 && tile_bytes <= (int64_t(200) << 20)) s->bal_flags |= 4;
    }
  } else {
    TRY(dev_alloc(s, &s->etei, size_t(h.diag_off_e.back())));
    TRY(dev_alloc(s, &s->tmp_rows, size_t(h.num_rows)));
    TRY(dev_alloc(s, &s->tmp_e, size_t(h.num_cols_e)));
    TRY(dev_alloc(s, &s->tmp_e2, size_t(h.num_cols_e)));
  }
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  // Sharded over the peer-to-peer communicator: do ALL ranks' camera-major passes exchange their blocks themselves (p2p.h)?  One
  // all-reduce of the ranks' verdicts — set_structure is collective for such instances (the ranks plan for about as long as each other;
  // this one exchange waits up to two minutes).
  s->cam_exchange_agreed = false;
  s->spec_agreed = false;
  s->rx_agreed = false;
  if (s->world > 1 && s->p2p && s->p2p_fuse) {
    double* d_agree = nullptr;
    TRY(dev_alloc(s, &d_agree, 3));
    // [0]: the camera-major pass cannot exchange here; [1]: an LM step cannot run its tail speculatively here (lm_step_loaded)
    const bool spec_local = s->speculate && is_schur(s) && !is_dense_schur(s) && s->path == CERES_HIP_PATH_BAL && !s->opt.use_explicit_schur_complement &&
                            s->fused_grid < 2 * kMaxVecGrid && !has_remainder(s) && exchange_in_producer(s, 2, 1);
    // [2]: the camera-space reduction of a tile pass cannot exchange itself here
    const double mine[3] = {camera_blocks_exchange_local(s) ? 0.0 : 1.0, spec_local ? 0.0 : 1.0, reduction_exchanges_local(s) ? 0.0 : 1.0};
    double sum[3] = {1.0, 1.0, 1.0};
    HIP_TRY(s, hipMemcpyAsync(d_agree, mine, sizeof(mine), hipMemcpyHostToDevice, s->stream));
    const double keep = s->p2p_timeout_s;
    s->p2p_timeout_s = std::max(keep, 120.0);
    const int rc = allreduce(s, d_agree, 3);
    s->p2p_timeout_s = keep;
    TRY(rc);
    HIP_TRY(s, hipMemcpyAsync(sum, d_agree, sizeof(sum), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    TRY(check_comm_error(s));
    s->cam_exchange_agreed = sum[0] == 0.0;
    s->spec_agreed = sum[1] == 0.0;
    s->rx_agreed = sum[2] == 0.0;
  }
  s->have_structure = true;
  return 0;
}

int ceres_hip_get_info(const ceres_hip_solver* s, ceres_hip_info* info) {
  if (!s || !info) return CERES_HIP_E_INVALID;
  memset(info, 0, sizeof(*info));
  const HostStructure& h = s->hs;
  info->kernel_path = s->path;
  info->num_rows = h.num_rows; info->num_cols = h.num_cols;
  info->num_cols_e = h.num_cols_e; info->num_cols_f = h.num_cols_f;
  info->num_row_blocks_e = h.num_row_blocks_e;
  info->num_e_blocks = h.nelim; info->num_f_blocks = h.ncb - h.nelim;
  info->row_block_size = h.det_row; info->e_block_size = h.det_e; info->f_block_size = h.det_f;
  info->num_nonzeros = h.nnz;
  info->num_observations = s->path == CERES_HIP_PATH_BAL ? s->plan.n_obs : 0;
  info->num_tiles = s->path == CERES_HIP_PATH_BAL ? s->plan.n_tiles : 0;
  info->device_bytes = s->device_bytes;
  info->camera_accum_in_lds = s->lds_mode ? 1 : 0;
  info->world_size = s->world; info->rank = s->rank;
  info->p2p_enabled = s->p2p ? 1 : 0;
  info->p2p_fine_grained = s->p2p_fine_grained ? 1 : 0;
  info->collectives_last_step = s->collectives;
  if (s->path == CERES_HIP_PATH_BAL) {
    info->camera_accum_hybrid = s->plan.hybrid ? 1 : 0;
    info->hybrid_popular_rows = s->plan.hybrid ? s->plan.hyb_hot : 0;
    info->num_observations_in_lds = s->lds_mode ? s->plan.n_obs : s->plan.n_local_obs;
    info->points_renumbered = s->plan.renumbered ? 1 : 0;
    // (the same condition solve_loaded applies: a block-diagonal preconditioner — not IDENTITY, not the power-series operator)
    info->cg_iteration_in_operator = (cg_tail_possible(s) && s->opt.preconditioner_type != CERES_HIP_IDENTITY &&
                                      s->opt.preconditioner_type != CERES_HIP_SCHUR_POWER_SERIES_EXPANSION) ? 1 : 0;
  }
  return 0;
}

#include "solver_comm.inc"

int ceres_hip_load(ceres_hip_solver* s, const double* hv, const double* hb, const double* hD) {
  if (!s) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  TRY(load_host(s, hv, hb, hD));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return 0;
}
int ceres_hip_load_device(ceres_hip_solver* s, const double* dv, const double* db, const double* dD) {
  if (!s) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  TRY(load_device(s, dv, db, dD));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return 0;
}

int ceres_hip_solve(ceres_hip_solver* s, const double* hv, const double* hb, const double* hD, double q_tol,
                    double r_tol, double* hx, ceres_hip_summary* summary) {
  if (!s || !summary || !hx) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  auto fatal = [&](int rc) {
    summary->termination_type = CERES_HIP_FATAL_ERROR;
    summary->residual_norm = -1;
    snprintf(summary->message, sizeof(summary->message), "%s", s->err.c_str());
    return rc;
  };
  if (!s->have_structure) return fatal(fail(s, CERES_HIP_E_INVALID, "ceres_hip_set_structure has not been called"));
  (void)rec(s, 0);
  // upload (ev0..ev1), pack (ev1..ev2)
  const HostStructure& h = s->hs;
  if (!hv || !hb) return fatal(fail(s, CERES_HIP_E_INVALID, "values and b must not be NULL"));
  if (hipMemcpyAsync(s->own_values, hv, sizeof(double) * h.values_extent, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
      hipMemcpyAsync(s->own_b, hb, sizeof(double) * h.num_rows, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
      (hD && hipMemcpyAsync(s->own_D, hD, sizeof(double) * h.num_cols, hipMemcpyHostToDevice, s->stream) != hipSuccess))
    return fatal(fail(s, CERES_HIP_E_HIP, "host-to-device copy failed"));
  (void)rec(s, 1);
  int rc = load_device(s, s->own_values, s->own_b, hD ? s->own_D : nullptr);
  if (rc) return fatal(rc);
  rc = solve_loaded(s, q_tol, r_tol, s->own_x, summary);
  if (rc) return fatal(rc);
  if (summary->termination_type != CERES_HIP_FAILURE && summary->termination_type != CERES_HIP_FATAL_ERROR) {
    if (hipMemcpyAsync(hx, s->own_x, sizeof(double) * h.num_cols, hipMemcpyDeviceToHost, s->stream) != hipSuccess)
      return fatal(fail(s, CERES_HIP_E_HIP, "device-to-host copy failed"));
  }
  (void)rec(s, 7);
  if (hipStreamSynchronize(s->stream) != hipSuccess) return fatal(fail(s, CERES_HIP_E_HIP, "stream synchronisation failed: %s", hipGetErrorString(hipGetLastError())));
  collect_timing(s);
  return 0;
}

int ceres_hip_solve_unchanged_values(ceres_hip_solver* s, const double* hD, double q_tol, double r_tol, double* hx, ceres_hip_summary* summary) {
  if (!s || !summary || !hx) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  auto fatal = [&](int rc) {
    summary->termination_type = CERES_HIP_FATAL_ERROR;
    summary->residual_norm = -1;
    snprintf(summary->message, sizeof(summary->message), "%s", s->err.c_str());
    return rc;
  };
  if (!s->have_structure || !s->loaded || !s->have_b)
    return fatal(fail(s, CERES_HIP_E_INVALID, "ceres_hip_solve_unchanged_values needs the values and b of a previous ceres_hip_solve / ceres_hip_load"));
  (void)rec(s, 0);
  const HostStructure& h = s->hs;
  if (hD && hipMemcpyAsync(s->own_D, hD, sizeof(double) * h.num_cols, hipMemcpyHostToDevice, s->stream) != hipSuccess)
    return fatal(fail(s, CERES_HIP_E_HIP, "host-to-device copy failed"));
  (void)rec(s, 1);
  s->D = hD ? s->own_D : nullptr;
  s->have_D = hD != nullptr;
  keep_loaded_values(s);
  int rc = solve_loaded(s, q_tol, r_tol, s->own_x, summary);
  if (rc) return fatal(rc);
  if (summary->termination_type != CERES_HIP_FAILURE && summary->termination_type != CERES_HIP_FATAL_ERROR) {
    if (hipMemcpyAsync(hx, s->own_x, sizeof(double) * h.num_cols, hipMemcpyDeviceToHost, s->stream) != hipSuccess)
      return fatal(fail(s, CERES_HIP_E_HIP, "device-to-host copy failed"));
  }
  (void)rec(s, 7);
  if (hipStreamSynchronize(s->stream) != hipSuccess) return fatal(fail(s, CERES_HIP_E_HIP, "stream synchronisation failed: %s", hipGetErrorString(hipGetLastError())));
  collect_timing(s);
  return 0;
}

int ceres_hip_solve_device(ceres_hip_solver* s, const double* dv, const double* db, const double* dD, double q_tol,
                           double r_tol, double* dx, ceres_hip_summary* summary) {
  if (!s || !summary || !dx) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  auto fatal = [&](int rc) {
    summary->termination_type = CERES_HIP_FATAL_ERROR;
    summary->residual_norm = -1;
    snprintf(summary->message, sizeof(summary->message), "%s", s->err.c_str());
    return rc;
  };
  (void)rec(s, 0);
  (void)rec(s, 1);
  int rc = load_device(s, dv, db, dD);
  if (rc) return fatal(rc);
  rc = solve_loaded(s, q_tol, r_tol, dx, summary);
  if (rc) return fatal(rc);
  (void)rec(s, 7);
  if (hipStreamSynchronize(s->stream) != hipSuccess) return fatal(fail(s, CERES_HIP_E_HIP, "stream synchronisation failed: %s", hipGetErrorString(hipGetLastError())));
  collect_timing(s);
  return 0;
}

namespace {
int lm_step_loaded(ceres_hip_solver* s, const ceres_hip_lm_options* o, double* dx, ceres_hip_lm_result* res) {
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  memset(res, 0, sizeof(*res));
  s->collectives = 0;
  // (Solver::Options::IsValid, I/solver.cc:414-416: min_lm_diagonal >= 0, max_lm_diagonal >= 0, min <= max; a NaN fails every test)
  if (!(o->radius > 0) || !(o->min_diagonal >= 0) || !(o->max_diagonal >= 0) || !(o->min_diagonal <= o->max_diagonal))
    return fail(s, CERES_HIP_E_INVALID, "bad LM options");
  // A fresh diagonal is diag(J^T J), which the set-up kernels of the <2,3,9> path have in hand anyway
  // (the point block's own diagonal; the camera columns' norms in the camera-major pass): then D
  // is formed inside them and no separate column-norm pass runs.
  const bool fresh = !o->reuse_diagonal || !s->have_lm_diag;
  // (DENSE_SCHUR forms no preconditioner blocks: nobody would form the camera part of a fused diagonal)
  // (a shared strip's column norms are not formed by the set-up kernels: such shapes take the separate column-norm pass)
  s->lm_fuse_active = fresh && s->path == CERES_HIP_PATH_BAL && s->opt.preconditioner_type != CERES_HIP_IDENTITY && !is_dense_schur(s) && s->plan.ns == 0;
  s->lm_opts = *o;
  if (fresh && !s->lm_fuse_active) TRY(op_squared_column_norm(s, s->lm_diag));
  if (!s->lm_fuse_active)
    HIP_TRY(s, LaunchLmDiagonal(s->lm_diag, o->min_diagonal, o->max_diagonal, o->radius, s->lm_D, h.num_cols, st));
  s->have_lm_diag = true;
  s->D = s->lm_D;
  s->have_D = true;
  s->lm_want_model_cost = is_schur(s) && s->path == CERES_HIP_PATH_BAL && s->fused_grid < 2 * kMaxVecGrid;  // (< : room for the remainder rows' partial)
  s->backsub_cost_parts = 0;
  // fused <2,3,9> path, implicit solvers: negation + finite check ride in the last kernel that touches the solution
  s->lm_negate_in_solve = s->path == CERES_HIP_PATH_BAL && !s->opt.use_explicit_schur_complement && !is_dense_schur(s);
  s->lm_negated = false;
  s->lm_cgnr_copy_pending = false;
  // (rows outside the tiles add ungated generic launches to the ITERATIVE_SCHUR tail: no speculation then)
  // (sharded: ITERATIVE_SCHUR where every rank can — the tail then holds one exchange per poll, which all ranks must issue alike)
  s->lm_speculate = s->speculate && s->lm_negate_in_solve &&
                    (s->world <= 1 ? (!is_schur(s) || (s->lm_want_model_cost && !has_remainder(s))) : (is_schur(s) && s->spec_agreed && s->p2p && s->lm_want_model_cost));
  s->spec_tail_done = false;
  const int rc = solve_loaded(s, o->eta, -1.0, dx, &res->linear_solver);
  const bool spec_done = s->spec_tail_done;
  s->spec_tail_done = false;
  s->lm_speculate = false;
  s->lm_fuse_active = false;
  s->lm_want_model_cost = false;
  s->lm_negate_in_solve = false;
  const bool cgnr_deferred = s->lm_cgnr_copy_pending;
  s->lm_cgnr_copy_pending = false;
  if (rc) return rc;
  res->step_is_finite = 0;
  const int term = res->linear_solver.termination_type;
  if (term == CERES_HIP_FAILURE || term == CERES_HIP_FATAL_ERROR) {
    if (cgnr_deferred) TRY(copy_out_cgnr_solution(s, dx));  // what Solve leaves in x
    return 0;
  }
  // Finite check + negation and the model cost change are enqueued together and read back with ONE
  // synchronisation (and, sharded, one all-reduce of {flag, cost}); a non-finite step makes the
  // cost meaningless, it is then ignored.
  if (!s->lm_negated && !spec_done) {
    if (!s->nonfinite_clean) HIP_TRY(s, hipMemsetAsync(s->d_nonfinite, 0, sizeof(int), st));
    s->nonfinite_clean = false;
    if (!cgnr_deferred) HIP_TRY(s, LaunchNegateAndCheck(dx, h.num_cols, s->d_nonfinite, st));
  }
  double* const neg = cgnr_deferred ? dx : nullptr;  // CGNR: the model-cost kernel writes dx = -y and checks it
  // parts_local: this rank's share (summed over ranks); parts_shared: replicated quantities (counted once)
  const double *parts_local = nullptr, *parts_shared = nullptr;
  int n_local = 0, n_shared = 0;
  if (!is_schur(s)) {
    // CGNR: no pass over J is needed.  With y the CG solution (step = -y), g = J^T f and r = g - (J^T J + D^2) y
    // the residual CG carries:  -(J step)'(f + J step / 2) = y.g - |J y|^2 / 2 = (y.g + y.r + |D y|^2) / 2.
    if (s->world > 1) {  // the point part is sharded, the camera part replicated
      HIP_TRY(s, LaunchCgnrModelCost(s->cg.x, s->cg_rhs, s->cg.r, s->D, 0, h.num_cols_e, s->scalar_partials, &n_local, st, neg, s->d_nonfinite, nullptr, point_perm(s)));
      HIP_TRY(s, LaunchCgnrModelCost(s->cg.x, s->cg_rhs, s->cg.r, s->D, h.num_cols_e, h.num_cols, s->scalar_partials + kMaxVecGrid, &n_shared, st, neg, s->d_nonfinite, nullptr, point_perm(s)));
      if (h.num_cols_e <= 0) n_local = 0;
      if (h.num_cols <= h.num_cols_e) n_shared = 0;
      parts_local = s->scalar_partials;
      parts_shared = s->scalar_partials + kMaxVecGrid;
    } else {
      if (spec_done) n_shared = s->spec_cgnr_parts;  // the gated kernel in front of the last poll already did all of it
      else HIP_TRY(s, LaunchCgnrModelCost(s->cg.x, s->cg_rhs, s->cg.r, s->D, 0, h.num_cols, s->scalar_partials, &n_shared, st, neg, s->d_nonfinite, nullptr, point_perm(s)));
      if (h.num_cols <= 0) n_shared = 0;
      parts_shared = s->scalar_partials;
    }
  } else if (s->backsub_cost_parts > 0) {  // the back-substitution kernel already formed it
    parts_local = s->scalar_partials;
    n_local = s->backsub_cost_parts;
  } else {
    TRY(enqueue_model_cost_change(s, dx, &parts_local, &n_local));  // rows are sharded: every rank holds a share
  }
  double* hp = s->h_pinned;
  int* h_flag = reinterpret_cast<int*>(hp + 2 * kMaxVecGrid);
  double v[2] = {0.0, 0.0};
  bool synced = false;   // the results are in h_pinned already (read_back waited for them)
  if (s->world > 1 && spec_done) {
    // the speculative tail summed them over ranks, and the poll that saw CG end brought them along
    hp[0] = hp[kSumsOffset]; hp[1] = hp[kSumsOffset + 1];
    synced = true;
  } else if (s->world > 1) {
    // sharded: {flag, this rank's share of the cost} are summed over ranks ON THE DEVICE, then read back once
    if (exchange_in_producer(s, 2, 1)) {
      HIP_TRY(s, LaunchCollectScalarsExchange(s->d_nonfinite, parts_local, n_local, s->cg.comm, next_exchange(s), st));
    } else {
      HIP_TRY(s, LaunchCollectScalars(s->d_nonfinite, parts_local, n_local, s->cg.comm, st));
      TRY(allreduce(s, s->cg.comm, 2));
    }
    if (n_shared == 0) { TRY(read_back(s, s->cg.comm, 2)); synced = true; }   // (nothing else to bring back: the mailbox, no synchronisation)
    else HIP_TRY(s, hipMemcpyAsync(hp, s->cg.comm, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  }
  // unsharded: the partial sums and the flag word live in one buffer (scalar_partials | flags): ONE copy brings them back
  const double* sp = s->scalar_partials;
  auto inside = [&](const double* q, int n) { return n == 0 || (q >= sp && q + n <= sp + 2 * kMaxVecGrid); };
  const bool one_copy = s->world <= 1 && inside(parts_local, n_local) && inside(parts_shared, n_shared);
  double* hl = hp;                  // host images of the two partial lists
  double* hsh = hp + kMaxVecGrid;
  if (one_copy) {
    // speculative tail: this copy was enqueued in front of the poll that saw CG end, and that poll synchronised
    if (!spec_done) { TRY(read_back(s, sp, 2 * kMaxVecGrid + 1)); synced = true; }
    h_flag = reinterpret_cast<int*>(hp + 2 * kMaxVecGrid);
    if (n_local > 0) hl = hp + (parts_local - sp);
    if (n_shared > 0) hsh = hp + (parts_shared - sp);
  } else {
    if (s->world <= 1) {
      HIP_TRY(s, hipMemcpyAsync(h_flag, s->d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
      if (n_local > 0) HIP_TRY(s, hipMemcpyAsync(hp, parts_local, sizeof(double) * n_local, hipMemcpyDeviceToHost, st));
    }
    if (n_shared > 0) HIP_TRY(s, hipMemcpyAsync(hp + kMaxVecGrid, parts_shared, sizeof(double) * n_shared, hipMemcpyDeviceToHost, st));
  }
  if (!(spec_done && one_copy) && !synced) {
    HIP_TRY(s, hipStreamSynchronize(st));
    TRY(check_comm_error(s));
  }
  if (s->world > 1) { v[0] = hp[0]; v[1] = hp[1]; }
  else {
    v[0] = double(*h_flag != 0);
    for (int i = 0; i < n_local; ++i) v[1] += hl[i];
  }
  double shared = 0;
  for (int i = 0; i < n_shared; ++i) shared += hsh[i];
  if (v[0] != 0.0) {  // "Linear solver failure. Failed to compute a finite step."  :124-128
    res->linear_solver.termination_type = CERES_HIP_FAILURE;
    snprintf(res->linear_solver.message, sizeof(res->linear_solver.message), "Failed to compute a finite step.");
    return 0;
  }
  res->step_is_finite = 1;
  res->model_cost_change = is_schur(s) ? v[1] : 0.5 * (v[1] + shared);
  return 0;
}
}  // namespace

int ceres_hip_lm_compute_step_device(ceres_hip_solver* s, const double* dv, const double* db,
                                     const ceres_hip_lm_options* o, double* dx, ceres_hip_lm_result* res) {
  if (!s || !o || !dx || !res) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  (void)rec(s, 0);
  (void)rec(s, 1);
  if (o->values_unchanged && s->loaded && s->have_b && s->values == dv && s->b == db) keep_loaded_values(s);
  else if (o->values_unchanged) return fail(s, CERES_HIP_E_INVALID, "values_unchanged = 1 but these are not the pointers of the previous load");
  else TRY(load_device(s, dv, db, nullptr));
  TRY(lm_step_loaded(s, o, dx, res));
  (void)rec(s, 7);
  // (the step's scalars came back through the mailbox, which the device stamps after everything else of the step: with nothing
  // enqueued behind it — no phase event — the stream has drained, and a synchronisation would only cost the host its round trip)
  if (s->timing_enabled || !s->mailbox || !s->final_sync_skippable) HIP_TRY(s, hipStreamSynchronize(s->stream));
  collect_timing(s);
  return 0;
}

int ceres_hip_lm_compute_step(ceres_hip_solver* s, const double* hv, const double* hb, const ceres_hip_lm_options* o,
                              double* hx, ceres_hip_lm_result* res) {
  if (!s || !o || !hx || !res) return CERES_HIP_E_INVALID;
  HIP_TRY(s, hipSetDevice(s->opt.device));
  (void)rec(s, 0);
  if (o->values_unchanged) {   // the copies of the previous call are still in HBM: nothing crosses PCIe but the step
    if (!(s->loaded && s->have_b && s->values == s->own_values && s->b == s->own_b))
      return fail(s, CERES_HIP_E_INVALID, "values_unchanged = 1 needs a previous ceres_hip_lm_compute_step / ceres_hip_load with host values and residuals");
    keep_loaded_values(s);
  } else {
    TRY(load_host(s, hv, hb, nullptr));
  }
  (void)rec(s, 1);
  TRY(lm_step_loaded(s, o, s->own_x, res));
  const int term = res->linear_solver.termination_type;
  if (term != CERES_HIP_FAILURE && term != CERES_HIP_FATAL_ERROR)
    HIP_TRY(s, hipMemcpyAsync(hx, s->own_x, sizeof(double) * s->hs.num_cols, hipMemcpyDeviceToHost, s->stream));
  (void)rec(s, 7);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  collect_timing(s);
  return 0;
}

#include "solver_stream.inc"

int ceres_hip_get_lm_diagonal(ceres_hip_solver* s, double* host_D) {
  TRY(require_loaded(s));
  if (!s->have_lm_diag) return fail(s, CERES_HIP_E_INVALID, "no LM step has been computed");
  HIP_TRY(s, hipMemcpyAsync(host_D, s->lm_D, sizeof(double) * s->hs.num_cols, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return 0;
}

int ceres_hip_op_scale_columns(ceres_hip_solver* s, const double* host_scale, double* host_values_out) {
  TRY(require_loaded(s));
  HIP_TRY(s, hipSetDevice(s->opt.device));
  if (s->values != s->own_values) return fail(s, CERES_HIP_E_INVALID, "scale_columns works on the copy made by ceres_hip_load");
  HIP_TRY(s, hipMemcpyAsync(s->scratch_vec, host_scale, sizeof(double) * s->hs.num_cols, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(s, LaunchGenScaleColumns(s->G, s->own_values, s->scratch_vec, s->stream));
  s->packed = false;
  s->rem_blocks_valid = false;
  s->precond_valid = false;
  s->ftf_inv_valid = false;
  if (host_values_out)
    HIP_TRY(s, hipMemcpyAsync(host_values_out, s->own_values, sizeof(double) * s->hs.values_extent, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return 0;
}

int ceres_hip_set_phase_timing(ceres_hip_solver* s, int32_t enable) {
  if (!s) return CERES_HIP_E_INVALID;
  s->timing_enabled = enable != 0;
  return 0;
}

int ceres_hip_get_last_timing(const ceres_hip_solver* s, ceres_hip_solve_timing* t) {
  if (!s || !t) return CERES_HIP_E_INVALID;
  *t = s->timing;
  return 0;
}

#include "solver_ops.inc"

int ceres_hip_time_op(ceres_hip_solver* s, int32_t op, int32_t iters, double* avg_ms) {
  TRY(require_loaded(s));
  HIP_TRY(s, hipSetDevice(s->opt.device));
  if (iters < 1 || !avg_ms) return CERES_HIP_E_INVALID;
  const HostStructure& h = s->hs;
  hipStream_t st = s->stream;
  const int64_t n = is_schur(s) ? h.num_cols_f : h.num_cols;
  std::function<int()> body;
  switch (op) {
    case CERES_HIP_TIMED_JTJX:
      if (is_schur(s)) return fail(s, CERES_HIP_E_INVALID, "jtjx needs a CGNR instance");
      TRY(ensure_D_int(s));
      body = [&] { return op_jtjx(s, s->cg.p, s->cg.z, nullptr, nullptr, nullptr, nullptr, true); };  // as CG applies it
      break;
    case CERES_HIP_TIMED_SX:
      if (!is_schur(s)) return fail(s, CERES_HIP_E_INVALID, "Sx needs an ITERATIVE_SCHUR instance");
      TRY(op_schur_init(s, true));
      body = [&] { return op_sx(s, s->cg.p, s->cg.z, nullptr); };
      break;
    case CERES_HIP_TIMED_SCHUR_INIT:  // as a solve runs it: fused with the re-layout of the step's values
      if (!is_schur(s)) return fail(s, CERES_HIP_E_INVALID, "needs an ITERATIVE_SCHUR instance");
      body = [&] { s->packed = false; return op_schur_init(s, true); };
      break;
    case CERES_HIP_TIMED_CGNR_SETUP:
      if (is_schur(s) || s->path != CERES_HIP_PATH_BAL) return fail(s, CERES_HIP_E_INVALID, "needs a CGNR instance on the <2,3,9> path");
      body = [&] { s->packed = false; return op_cgnr_setup_bal(s, s->opt.preconditioner_type == CERES_HIP_JACOBI, s->cg_rhs, s->precond); };
      break;
    case CERES_HIP_TIMED_SCHUR_JACOBI:
      if (!is_schur(s)) return fail(s, CERES_HIP_E_INVALID, "needs an ITERATIVE_SCHUR instance");
      TRY(op_schur_init(s, true));
      body = [&] { return op_preconditioner(s, CERES_HIP_SCHUR_JACOBI, s->precond, true); };
      break;
    case CERES_HIP_TIMED_BLOCK_JACOBI:
      if (is_schur(s)) TRY(op_schur_init(s, false));
      body = [&] { return op_preconditioner(s, CERES_HIP_JACOBI, s->precond, true); };
      break;
    case CERES_HIP_TIMED_BACK_SUBSTITUTE:
      if (!is_schur(s)) return fail(s, CERES_HIP_E_INVALID, "needs an ITERATIVE_SCHUR instance");
      TRY(op_schur_init(s, false));
      body = [&] { return op_back_substitute(s, s->cg.p, s->own_x); };
      break;
    case CERES_HIP_TIMED_PACK:
      if (s->path != CERES_HIP_PATH_BAL) return fail(s, CERES_HIP_E_INVALID, "pack exists on the <2,3,9> path only");
      body = [&] { s->packed = false; return ensure_packed(s); };
      break;
    case CERES_HIP_TIMED_COPY:
      if (s->d_Jf) return fail(s, CERES_HIP_E_UNSUPPORTED, "copy probe needs the fp64 tile buffer");
      if (s->path != CERES_HIP_PATH_BAL) return fail(s, CERES_HIP_E_INVALID, "copy probe uses the packed buffer of the <2,3,9> path");
      body = [&] {
        HIP_TRY(s, hipMemcpyAsync(s->d_J, s->values, sizeof(double) * std::min<int64_t>(h.values_extent, s->plan.n_tiles * int64_t(s->ops->tile_pitch) * 2), hipMemcpyDeviceToDevice, st));
        return 0;
      };
      break;
    case CERES_HIP_TIMED_READ_STREAM:
      if (s->d_Jf) return fail(s, CERES_HIP_E_UNSUPPORTED, "read-stream probe needs the fp64 tile buffer");
      if (s->path != CERES_HIP_PATH_BAL) return fail(s, CERES_HIP_E_INVALID, "read-stream probe uses the packed tiles of the <2,3,9> path");
      body = [&] { HIP_TRY(s, s->ops->stream_probe(s->d_J, s->plan.n_tiles, std::min(s->num_cus, s->plan.nf * s->plan.n_cameras), s->d_global_acc, st)); return 0; };
      break;
    default:
      return fail(s, CERES_HIP_E_INVALID, "unknown timed op %d", op);
  }
  TRY(ensure_packed(s));
  HIP_TRY(s, LaunchSet(s->cg.p, 1.0, n, st));
  for (int w = 0; w < 3; ++w) TRY(body());
  HIP_TRY(s, hipEventRecord(s->ev[8], st));
  for (int i = 0; i < iters; ++i) TRY(body());
  HIP_TRY(s, hipEventRecord(s->ev[9], st));
  HIP_TRY(s, hipStreamSynchronize(st));
  *avg_ms = double(elapsed(s->ev[8], s->ev[9])) / iters;
  if (op == CERES_HIP_TIMED_COPY) { s->packed = false; TRY(ensure_packed(s)); }  // restore the tiles
  HIP_TRY(s, hipStreamSynchronize(st));
  return 0;
}

#include "solver_debug.inc"

#include "bal_frontend.inc"
