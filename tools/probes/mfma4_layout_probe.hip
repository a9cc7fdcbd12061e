// mfma4_layout_probe.hip — which lane holds which element of v_mfma_f64_4x4x4_4b_f64's operands (four independent 4 x 4 x 4 blocks per
// instruction, one double per lane for A, B and D)?  Runs the instruction on known values and searches the 6 x 6 x 6 ways of assigning
// the three 2-bit lane fields to (block, row, k) / (block, k, column) / (block, row, column).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
static int field(int lane, int which) { return (lane >> (2 * which)) & 3; }   // which = 0: bits 0-1, 1: bits 2-3, 2: bits 4-5
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + l; hb[l] = 1.0 / (3.0 + l); }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  // perm p: fields (first, second, third index) live in lane fields p[0], p[1], p[2]
  int found = 0;
  for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
    // A[blk][i][kk], B[blk][kk][j], D[blk][i][j]
    double A[4][4][4], B[4][4][4];
    for (int l = 0; l < 64; ++l) {
      A[field(l, perms[pa][0])][field(l, perms[pa][1])][field(l, perms[pa][2])] = ha[l];
      B[field(l, perms[pb][0])][field(l, perms[pb][1])][field(l, perms[pb][2])] = hb[l];
    }
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l) {
      const int blk = field(l, perms[pd][0]), i = field(l, perms[pd][1]), j = field(l, perms[pd][2]);
      double s = 0;
      for (int kk = 0; kk < 4; ++kk) s += A[blk][i][kk] * B[blk][kk][j];
      ok = std::fabs(s - hd[l]) <= 1e-12 * std::fabs(s);
    }
    if (ok) {
      ++found;
      printf("MATCH: A(block,row,k) in lane fields (%d,%d,%d); B(block,k,col) in (%d,%d,%d); D(block,row,col) in (%d,%d,%d)   [field f = lane bits 2f..2f+1]\n",
             perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2], perms[pd][0], perms[pd][1], perms[pd][2]);
    }
  }
  printf("%d matching layouts; d[0..7] = %g %g %g %g %g %g %g %g\n", found, hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7]);
  return 0;
}
