// 16-bit storage instantiations: bf16 (BASELINE configs[2]/[3]) and fp16 (configs[4], inference) activations, gradients and
// weight panels in HBM; LDS-DMA ring and pixel-patch main loops
#include "igemm_kernel.h"

int dbn_launch_igemm_b16(IgemmParams& p, int cfg, int mode, int at, hipStream_t st) {
    if (at == 1) return launch_igemm_cfg<1, 1>(p, cfg, mode, st);
    if (at == 2) return launch_igemm_cfg<1, 2>(p, cfg, mode, st);
    return DBN_ERR_ARG;
}
