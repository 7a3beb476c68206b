// Realistic ceiling of v_mfma_f32_32x32x16_{f16,bf16} on this box (power-capped clock, non-zero operands): W waves per workgroup, C
// accumulator chains per wave, N MFMAs per chain; operands from registers (random bits), no memory traffic in the loop.
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/mfma_peak.hip -o gpurun_out/libmfma_peak.so; driver: tools/mfma_peak.py
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(512) void mfma_peak_kernel(const f32x4* __restrict__ in, float* __restrict__ out, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(f16x8, in[(threadIdx.x * 8 + i) & 4095]);
        b[i] = __builtin_bit_cast(f16x8, in[(threadIdx.x * 8 + 4 + i) & 4095]);
    }
    f32x16 acc[C];
    for (int c = 0; c < C; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + c) & 3], b[k & 3], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int mfma_peak(int chains, int waves_per_wg, int wgs, const void* in, float* out, int iters, void* stream) {
    dim3 g(wgs), b(waves_per_wg * 64);
    if (chains == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, g, b, 0, (hipStream_t)stream, (const f32x4*)in, out, iters);
    else if (chains == 2) hipLaunchKernelGGL(mfma_peak_kernel<2>, g, b, 0, (hipStream_t)stream, (const f32x4*)in, out, iters);
    else hipLaunchKernelGGL(mfma_peak_kernel<4>, g, b, 0, (hipStream_t)stream, (const f32x4*)in, out, iters);
    return (int)hipGetLastError();
}
