// The fused kernels for camera blocks 7 wide and a shared strip of 0 scalars (common.h: shapes; kernels_bal.inc: the kernels).
#define CERES_HIP_NF 7
#define CERES_HIP_NS 0
#define CERES_HIP_SHAPE bal_f7_s0
#include "kernels_bal.inc"
