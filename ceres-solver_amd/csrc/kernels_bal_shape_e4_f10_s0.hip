// The fused kernels for point blocks 4 wide and camera blocks 10 wide, no shared strip: a camera width the reference reaches through its
// dynamic-size specialisations (2,4,d) (internal/ceres/generate_template_specializations.py:55-75; common.h: shapes; kernels_bal.inc: the kernels).
#define CERES_HIP_NE 4
#define CERES_HIP_NF 10
#define CERES_HIP_NS 0
#define CERES_HIP_SHAPE bal_e4_f10_s0
#include "kernels_bal.inc"
