// Probe: does a vector-ALU instruction cost fp32-MFMA time?  Each wave issues 8 independent v_mfma_f32_32x32x2_f32 per step (the
// Winograd weight-gradient kernel's k-step) and, between them, NV vector instructions on live registers — scalar v_fma_f32 (PK = 0) or
// v_pk_fma_f32 (PK = 1).  WPS waves per SIMD.  Prints the sustained fraction of the fp32-MFMA peak and the shader clocks per step and
// wave: if the vector instructions ran in the shadow of the MFMAs the clocks would not move with NV.
// GROUPED = 1: the step's 8 MFMAs back to back, then its 8 NV vector instructions in one batch (same totals).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int PK, int GROUPED>
__global__ __launch_bounds__(256) void k(float* out, int steps, unsigned long long* cyc) {
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float v[16];
    f32x2 p[8];
    for (int i = 0; i < 16; ++i) v[i] = 1.f + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
    const float c = 1.0f + 1e-7f * threadIdx.x;
    const f32x2 c2 = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < steps; ++it) {
        if (GROUPED) {  // the 8 MFMAs back to back, then all 8 NV vector instructions
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[m], v[m + 8], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    if (PK) p[(m + q) & 7] = __builtin_elementwise_fma(p[(m + q) & 7], c2, c2);
                    else v[(2 * m + q) & 15] = __builtin_fmaf(v[(2 * m + q) & 15], c, c);
                }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[m], v[m + 8], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (PK) p[(m + q) & 7] = __builtin_elementwise_fma(p[(m + q) & 7], c2, c2);
                else v[(2 * m + q) & 15] = __builtin_fmaf(v[(2 * m + q) & 15], c, c);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int PK, int GROUPED = 0>
void run(int wps, float* out, unsigned long long* cyc) {
    const int steps = 4000, grid = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, PK, GROUPED>), dim3(grid), dim3(256), 0, 0, out, 200, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, PK, GROUPED>), dim3(grid), dim3(256), 0, 0, out, steps, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0;
    hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * steps * 8 * 32 * 32 * 2 * 2;
    printf("%d vector instruction(s) per MFMA (%s, %s), %d wave(s) per SIMD: %6.1f TFLOP/s = %.3f of 157.3   (%5.0f shader clocks per step and wave; 512 x waves ideal)\n",
           NV, PK ? "v_pk_fma_f32" : "v_fma_f32   ", GROUPED ? "8 MFMAs then the batch" : "interleaved           ", wps, flops / ms / 1e9, flops / ms / 1e9 / 157.3, (double)c0 / steps);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMalloc(&cyc, 8 * 1024);
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 0>(wps, out, cyc);
        run<1, 0>(wps, out, cyc);
        run<2, 0>(wps, out, cyc);
        run<4, 0>(wps, out, cyc);
        run<8, 0>(wps, out, cyc);
        run<1, 1>(wps, out, cyc);
        run<2, 1>(wps, out, cyc);
        run<4, 1>(wps, out, cyc);
        run<1, 0, 1>(wps, out, cyc);
        run<2, 0, 1>(wps, out, cyc);
        run<4, 0, 1>(wps, out, cyc);
        run<2, 1, 1>(wps, out, cyc);
    }
    return 0;
}
