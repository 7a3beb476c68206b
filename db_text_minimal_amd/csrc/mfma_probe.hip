// What the matrix pipe SUSTAINS on this box with non-zero operands (bench.py: roofline.peak_sustained).  The chip clocks to its power
// budget (MI355X_MICROARCH.md "DVFS give-back"): the nominal peaks — 2.5 PFLOP/s bf16 / fp16, 157.3 TFLOP/s fp32 — are products of the
// 2.4 GHz maximum clock; under back-to-back MFMAs on random operands the clock settles far below it (round 6: fp16 1650 TFLOP/s =
// 0.66 of nominal on the benchmark's boxes, 2496 with all-zero operands).  This kernel is the yardstick the conv kernels' fractions are
// read against besides the nominal one: 8 waves per CU (two per SIMD, as the conv kernels run), one accumulator chain per wave,
// operands from registers, no memory traffic in the loop.  (The reference has no counterpart: measurement infrastructure only.)
#include "common.h"

namespace {
typedef __bf16 p_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 p_f16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_f32_32x32x16_f16
template <int KIND>
__global__ __launch_bounds__(512) void mfma_sustained_kernel(const f32x4* __restrict__ in, float* __restrict__ out, int iters) {
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(threadIdx.x * 8 + i) & 4095];
        b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 3][k >> 2], b[(k + 1) & 3][k >> 2], acc, 0, 0, 0);
            else if constexpr (KIND == 1)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(p_bf16x8, a[k & 3]), __builtin_bit_cast(p_bf16x8, b[(k + 1) & 3]), acc, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p_f16x8, a[k & 3]), __builtin_bit_cast(p_f16x8, b[(k + 1) & 3]), acc, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

extern "C" {
// operands: 4096 x 16 bytes of the operand type (fp32 / bf16 / fp16 values as the caller chose them); out: 1024 * 512 floats (ignored
// values); one launch of 1024 workgroups x 8 waves, `iters` x 16 MFMAs per wave.  FLOPs of the launch: dbn_mfma_sustained_flops.
int dbn_mfma_sustained(int kind, const void* operands, float* out, int iters, void* stream) {
    DBN_REQUIRE(operands && out && iters > 0 && kind >= 0 && kind <= 2);
    const dim3 g(1024), b(512);
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(mfma_sustained_kernel<0>, g, b, 0, st, (const f32x4*)operands, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(mfma_sustained_kernel<1>, g, b, 0, st, (const f32x4*)operands, out, iters);
    else hipLaunchKernelGGL(mfma_sustained_kernel<2>, g, b, 0, st, (const f32x4*)operands, out, iters);
    return dbn_status();
}
long dbn_mfma_sustained_flops(int kind, int iters) {
    return 2L * 32 * 32 * (kind == 0 ? 2 : 16) * 16L * iters * 8L * 1024L;
}
}
