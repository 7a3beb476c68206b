// fp32 tensors, products on the bf16 matrix pipe: ns = 3 exact three-way operand split ('bf16x3'), ns = 1 operands rounded
// when staged ('bf16c'); with -DDBN_EXPERIMENTS also the pre-split plane source (at = 3)
#include "igemm_kernel.h"

int dbn_launch_igemm_x(IgemmParams& p, int cfg, int mode, int ns, int at, hipStream_t st) {
#if DBN_HAS_EXPERIMENTS
    if (at == 3) return ns == 3 ? launch_igemm_cfg<3, 3>(p, cfg, mode, st) : DBN_ERR_ARG;
#endif
    if (at != 0) return DBN_ERR_ARG;
    if (ns == 1) return launch_igemm_cfg<1, 0>(p, cfg, mode, st);
    if (ns == 3) return launch_igemm_cfg<3, 0>(p, cfg, mode, st);
    return DBN_ERR_ARG;
}
