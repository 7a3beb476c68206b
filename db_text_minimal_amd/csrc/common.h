// Shared device/host helpers for the DBNet gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DBN_OK 0
#define DBN_ERR_ARG 1

#define DBN_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return DBN_ERR_ARG; \
    } while (0)

static inline int dbn_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DBN_OK : 1000 + (int)e;
}

static inline int dbn_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Grid for a grid-stride pointwise kernel: enough blocks to fill 256 CUs x 8, capped.
static inline int dbn_grid(long work_items, int block = 256, int cap = 4096) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// MI355X: 8 XCDs, block b runs on XCD b % 8 (observed; used for L2 locality only).
// Bijective remap that gives each XCD a contiguous run of tiles.
__device__ __forceinline__ int dbn_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float dbn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double dbn_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Partial sums are stored transposed, part[idx*nb + b] (b = producing block), so that the 32 lanes
// folding one output read one contiguous 128-byte line per step.  Call with all 32 lanes of an
// aligned half-wave (lane32 = threadIdx.x & 31); every lane returns the total (fp64 fold).
__device__ __forceinline__ double dbn_team32_fold(const float* __restrict__ part, int nb, long idx, int lane32) {
    const float* row = part + idx * nb;
    double s = 0.0;
    int b = lane32;
    for (; b + 96 < nb; b += 128) {  // four independent loads in flight per lane
        const float v0 = row[b], v1 = row[b + 32], v2 = row[b + 64], v3 = row[b + 96];
        s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; b < nb; b += 32) s += (double)row[b];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

// BatchNorm affine + optional ReLU written once so that every kernel that
// recomputes an activation gets bit-identical values (maxpool backward relies
// on float equality with the forward).
__device__ __forceinline__ float dbn_affine(float y, float sc, float sh) { return fmaf(y, sc, sh); }
__device__ __forceinline__ float dbn_affine_relu(float y, float sc, float sh) { return fmaxf(fmaf(y, sc, sh), 0.f); }
