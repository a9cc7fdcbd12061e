// The fused kernels for camera blocks 9 wide and a shared strip of 4 scalars (common.h: shapes; kernels_bal.inc: the kernels).
#define CERES_HIP_NF 9
#define CERES_HIP_NS 4
#define CERES_HIP_SHAPE bal_f9_s4
#include "kernels_bal.inc"
