// Probe: what fraction of the fp32 MFMA peak (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, 157.3 TFLOP/s at 2.4 GHz x 256 CUs) a
// workgroup of four waves sustains (a) with nothing but MFMAs, (b) with the LDS fragment reads of the igemm k-loop, (c) with
// the reads AND one workgroup barrier per 16-k step — the structure of igemm_f32_kernel's main loop without any global memory.
// WPS = workgroups per CU (waves per SIMD); ACC = accumulators per wave (MI*NI of the tile).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMAs only; 1: + LDS fragment reads; 2: + barrier per k-step; 3: + the staging ds_write_b128 (3 per thread and k-step, from
// registers); 4: + the global loads that feed them (3 x 16 B per thread and k-step, prefetch distance two, L2-resident source with
// the igemm gather's footprint: 4 lanes per 64-byte run, one run per row)
template <int ACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int steps, unsigned long long* cyc, const f32x4* __restrict__ src, unsigned mask) {
    __shared__ f32x4 smem[2 * (4 * 130 + 4 * 130)];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * 1040; i += 256) smem[i] = f32x4{1.f + i, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[ACC];
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 af[2][2], bf[2][2];
    for (int s = 0; s < 2; ++s)
        for (int a = 0; a < 2; ++a) {
            af[s][a] = smem[(2 * s + lh) * 130 + a * 32 + li];
            bf[s][a] = smem[520 + (2 * s + lh) * 130 + a * 32 + li];
        }
    f32x4 st[2][3];
    for (int u = 0; u < 2; ++u)
        for (int j = 0; j < 3; ++j) st[u][j] = f32x4{1.f, 2.f, 3.f, 4.f};
    // gather-like addresses: row = tid >> 2 (+ 64), 64-byte run per row at a 9 KB row pitch region walked cyclically
    unsigned goff0 = ((blockIdx.x * 131u + (tid >> 2)) * 576u + (tid & 3) * 4u) & mask, goff1 = (goff0 + 64u * 576u) & mask;
    unsigned goffb = (blockIdx.x * 977u + tid) & mask;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < steps; ++it) {
        const f32x4* As = smem + (it & 1) * 1040;
        if (MODE >= 4) {
            st[it & 1][0] = src[goff0];
            st[it & 1][1] = src[goff1];
            st[it & 1][2] = src[goffb];
            goff0 = (goff0 + 16u) & mask;
            goff1 = (goff1 + 16u) & mask;
            goffb = (goffb + 256u) & mask;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    af[s][a] = As[(2 * s + lh) * 130 + a * 32 + li];
                    bf[s][a] = As[520 + (2 * s + lh) * 130 + a * 32 + li];
                }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < ACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][a & 1][e], bf[s][(a >> 1) & 1][e], acc[a], 0, 0, 0);
        if (MODE >= 3) {
            f32x4* Ws = smem + ((it + 1) & 1) * 1040;
            Ws[(tid & 3) * 130 + (tid >> 2)] = st[(it + 1) & 1][0];
            Ws[(tid & 3) * 130 + (tid >> 2) + 64] = st[(it + 1) & 1][1];
            Ws[520 + (tid >> 6) * 66 + (tid & 63)] = st[(it + 1) & 1][2];
        }
        if (MODE >= 2) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = st[0][0][0] + st[1][1][1] + st[0][2][2];
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int ACC, int MODE>
void run(const char* what, int wg_per_cu) {
    float* out;
    unsigned long long* cyc;
    static f32x4* src = nullptr;
    const unsigned mask = (64u << 20) / 16 - 1;  // 64 MB of source, in 16-byte units
    if (!src) {
        hipMalloc(&src, 64u << 20);
        hipMemset(src, 0, 64u << 20);
    }
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 8);
    const int steps = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ACC, MODE>), dim3(grid), dim3(256), 0, 0, out, 100, cyc, src, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC, MODE>), dim3(grid), dim3(256), 0, 0, out, steps, cyc, src, mask);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * steps * 8.0 * ACC * 4096.0;
    printf("%-52s acc %d  wg/CU %d: %7.1f TFLOP/s = %.3f of 157.3   (%.0f shader clocks per k-step and wave; ideal %d)\n", what, ACC,
           wg_per_cu, flops / ms / 1e9, flops / ms / 1e9 / 157.3, (double)c / steps, 8 * ACC * 64 * wg_per_cu);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; ++w) run<2, 0>("MFMA only (128x64 tile: 2 accumulators)", w);
    for (int w = 1; w <= 3; ++w) run<4, 0>("MFMA only (128x128 tile: 4 accumulators)", w);
    for (int w = 1; w <= 4; ++w) run<2, 1>("MFMA + LDS fragment reads", w);
    for (int w = 1; w <= 3; ++w) run<4, 1>("MFMA + LDS fragment reads", w);
    for (int w = 1; w <= 4; ++w) run<2, 2>("MFMA + LDS reads + barrier per k-step", w);
    for (int w = 1; w <= 3; ++w) run<4, 2>("MFMA + LDS reads + barrier per k-step", w);
    for (int w = 1; w <= 4; ++w) run<2, 3>("... + staging ds_write_b128 x3", w);
    for (int w = 1; w <= 3; ++w) run<4, 3>("... + staging ds_write_b128 x3", w);
    for (int w = 1; w <= 4; ++w) run<2, 4>("... + global loads x3 (distance 2)", w);
    for (int w = 1; w <= 3; ++w) run<4, 4>("... + global loads x3 (distance 2)", w);
    return 0;
}
