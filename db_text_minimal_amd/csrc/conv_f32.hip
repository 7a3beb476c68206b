// exact-fp32 instantiations of the implicit-GEMM kernel (v_mfma_f32_32x32x2_f32; BASELINE configs[1], the headline path)
#include "igemm_kernel.h"

int dbn_launch_igemm_f32(IgemmParams& p, int cfg, int mode, hipStream_t st) { return launch_igemm_cfg<0, 0>(p, cfg, mode, st); }
