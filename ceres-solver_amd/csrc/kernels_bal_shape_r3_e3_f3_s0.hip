// The fused kernels for row blocks 3 high, point blocks 3 wide and camera blocks 3 wide, no shared strip: the reference's (3,3,3)
// specialisation (internal/ceres/generate_template_specializations.py:55-75; common.h: shapes; kernels_bal.inc: the kernels).
#define CERES_HIP_NR 3
#define CERES_HIP_NE 3
#define CERES_HIP_NF 3
#define CERES_HIP_NS 0
#define CERES_HIP_SHAPE bal_r3_e3_f3_s0
#include "kernels_bal.inc"
