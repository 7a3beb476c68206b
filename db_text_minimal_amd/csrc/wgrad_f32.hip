// exact-fp32 weight-gradient kernels (BASELINE configs[1], the headline path)
#include "wgrad_kernels.h"

int dbn_launch_wgrad_f32(const WgradParams& p, int kind, int bm, int bn, dim3 grid, hipStream_t st) {
    if (kind == 0) return launch_wgrad_tiles<0, 0>(p, bm, bn, grid, st);
#if DBN_HAS_EXPERIMENTS
    if (kind == 3) {
        if (bn == 192)
            hipLaunchKernelGGL((wgrad_dma_kernel<64, 192, 2, 2>), grid, dim3(256), 0, st, p);
        else if (bm == 128 && bn == 128)
            hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
        else if (bn == 128)
            hipLaunchKernelGGL((wgrad_dma_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((wgrad_dma_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
        return dbn_status();
    }
#endif
    return DBN_ERR_ARG;
}
