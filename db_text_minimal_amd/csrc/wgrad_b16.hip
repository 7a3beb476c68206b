// weight-gradient kernels on the bf16 matrix pipe: fp32 tensors (ns 1 / 3) and stored-bf16 tensors (at 1)
#include "wgrad_kernels.h"

int dbn_launch_wgrad_b16(const WgradParams& p, int kind, int ns, int at, int bm, int bn, dim3 grid, hipStream_t st) {
    if (kind == 2) {  // pixel-patch kernel (3x3, stride 1)
        if (at == 1) hipLaunchKernelGGL((wgrad_patch_kernel<1, 1>), grid, dim3(256), 0, st, p);
        else if (ns == 1) hipLaunchKernelGGL((wgrad_patch_kernel<1, 0>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_patch_kernel<3, 0>), grid, dim3(256), 0, st, p);
        return dbn_status();
    }
    if (kind == 1) {  // stored bf16: LDS-DMA + transposing reads
        if (bn == 192)
            hipLaunchKernelGGL((wgrad_tr_kernel<64, 192, 2, 2>), grid, dim3(256), 0, st, p);
        else if (bm == 128 && bn == 128)
            hipLaunchKernelGGL((wgrad_tr_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
        else if (bn == 128)
            hipLaunchKernelGGL((wgrad_tr_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((wgrad_tr_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
        return dbn_status();
    }
    if (kind != 0) return DBN_ERR_ARG;
    if (at == 1) return launch_wgrad_tiles<1, 1>(p, bm, bn, grid, st);
#if DBN_HAS_EXPERIMENTS
    if (at == 3) return launch_wgrad_tiles<3, 3>(p, bm, bn, grid, st);
#endif
    if (at != 0) return DBN_ERR_ARG;
    if (ns == 1) return launch_wgrad_tiles<1, 0>(p, bm, bn, grid, st);
    if (ns == 3) return launch_wgrad_tiles<3, 0>(p, bm, bn, grid, st);
    return DBN_ERR_ARG;
}
