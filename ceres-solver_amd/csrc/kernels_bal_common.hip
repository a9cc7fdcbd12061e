// kernels_bal_common.hip — what the per-shape translation units of the fused kernels (kernels_bal.inc, kernels_bal_shape_*.hip) share:
// the table of compiled shapes, the dynamic-LDS ceiling, the workgroup size of a mode.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "device.h"

namespace chip {

// The dynamic-LDS ceiling is a per-device attribute of a kernel: set it once per (kernel, device), whichever thread
// gets there first.  Keyed by the kernel's ADDRESS (all kernels here share one function type, so a static per
// template instantiation would be shared between them).
hipError_t AllowMaxLds(const void* kernel) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned long long> done;  // kernel -> mask of devices 0..63
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> lock(mu);
  unsigned long long& mask = done[kernel];
  if (mask & bit) return hipSuccess;
  // static + dynamic LDS must fit the CU's 160 KB: kernels with a few bytes of static __shared__ (cross-wave reductions) get that much less
  hipFuncAttributes fa;
  if (hipError_t e = hipFuncGetAttributes(&fa, kernel); e != hipSuccess) return e;
  const int dyn = int(kMaxLdsBytes) - int(fa.sharedSizeBytes);
  if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); e != hipSuccess) return e;
  mask |= bit;
  return hipSuccess;
}

// Threads per workgroup of the streaming kernels: 1024 (16 waves per CU) hides HBM latency
// best for the light modes; the set-up modes need more registers and run at 512.
int BalBlockFor(int mode) {
  static int forced = [] { const char* e = getenv("CERES_HIP_BAL_BLOCK"); return e ? atoi(e) : 0; }();
  if (mode == kBalInit || mode == kBalCgnrInit || mode == kBalColNorm || mode == kBalShBlocks) return 512;
  if (forced == 512 || forced == 1024) return forced;
  return 1024;
}

// one per kernels_bal_shape_*.hip; the list is common.h's BalShapeCompiled
const BalOps* BalOps_bal_f2_s0();
const BalOps* BalOps_bal_f3_s0();
const BalOps* BalOps_bal_f4_s0();
const BalOps* BalOps_bal_f5_s0();
const BalOps* BalOps_bal_f6_s0();
const BalOps* BalOps_bal_f7_s0();
const BalOps* BalOps_bal_f8_s0();
const BalOps* BalOps_bal_f9_s0();
const BalOps* BalOps_bal_f10_s0();
const BalOps* BalOps_bal_f6_s4();
const BalOps* BalOps_bal_f6_s8();
const BalOps* BalOps_bal_f9_s4();
const BalOps* BalOps_bal_f9_s8();
const BalOps* BalOps_bal_e2_f2_s0();
const BalOps* BalOps_bal_e2_f3_s0();
const BalOps* BalOps_bal_e2_f4_s0();
const BalOps* BalOps_bal_e2_f6_s0();
const BalOps* BalOps_bal_e2_f9_s0();
const BalOps* BalOps_bal_e4_f2_s0();
const BalOps* BalOps_bal_e4_f3_s0();
const BalOps* BalOps_bal_e4_f5_s0();
const BalOps* BalOps_bal_e4_f7_s0();
const BalOps* BalOps_bal_e4_f10_s0();
const BalOps* BalOps_bal_e4_f4_s0();
const BalOps* BalOps_bal_e4_f6_s0();
const BalOps* BalOps_bal_e4_f8_s0();
const BalOps* BalOps_bal_e4_f9_s0();
const BalOps* BalOps_bal_r3_e3_f3_s0();
const BalOps* BalOps_bal_r4_e4_f2_s0();
const BalOps* BalOps_bal_r4_e4_f3_s0();
const BalOps* BalOps_bal_r4_e4_f4_s0();

const BalOps* GetBalOps(int nr, int ne, int nf, int ns) {
  if (!BalShapeCompiled(nr, ne, nf, ns)) return nullptr;
  if (nr == 3) return BalOps_bal_r3_e3_f3_s0();
  if (nr == 4) {
    switch (nf) {
      case 2: return BalOps_bal_r4_e4_f2_s0();
      case 3: return BalOps_bal_r4_e4_f3_s0();
      case 4: return BalOps_bal_r4_e4_f4_s0();
    }
    return nullptr;
  }
  if (ne == 2) {
    switch (nf) {
      case 2: return BalOps_bal_e2_f2_s0();
      case 3: return BalOps_bal_e2_f3_s0();
      case 4: return BalOps_bal_e2_f4_s0();
      case 6: return BalOps_bal_e2_f6_s0();
      case 9: return BalOps_bal_e2_f9_s0();
    }
    return nullptr;
  }
  if (ne == 4) {
    switch (nf) {
      case 2: return BalOps_bal_e4_f2_s0();
      case 3: return BalOps_bal_e4_f3_s0();
      case 4: return BalOps_bal_e4_f4_s0();
      case 5: return BalOps_bal_e4_f5_s0();
      case 6: return BalOps_bal_e4_f6_s0();
      case 7: return BalOps_bal_e4_f7_s0();
      case 8: return BalOps_bal_e4_f8_s0();
      case 9: return BalOps_bal_e4_f9_s0();
      case 10: return BalOps_bal_e4_f10_s0();
    }
    return nullptr;
  }
  switch (nf * 100 + ns) {
    case 200: return BalOps_bal_f2_s0();
    case 300: return BalOps_bal_f3_s0();
    case 400: return BalOps_bal_f4_s0();
    case 500: return BalOps_bal_f5_s0();
    case 600: return BalOps_bal_f6_s0();
    case 700: return BalOps_bal_f7_s0();
    case 800: return BalOps_bal_f8_s0();
    case 900: return BalOps_bal_f9_s0();
    case 1000: return BalOps_bal_f10_s0();
    case 604: return BalOps_bal_f6_s4();
    case 608: return BalOps_bal_f6_s8();
    case 904: return BalOps_bal_f9_s4();
    case 908: return BalOps_bal_f9_s8();
  }
  return nullptr;
}

}  // namespace chip
