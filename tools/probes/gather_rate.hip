// Probe: throughput of 16-byte-per-lane buffer loads as a function of how many lanes share a 128-byte line, for register loads and for
// LDS-DMA loads, on an L2/L1-resident footprint.  What the im2col gather of the 16-bit convolution kernels costs per wave instruction.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// lanes per line L (1, 2, 4, 8): lane -> line = lane / L, piece = lane % L; successive iterations walk the next group of 64/L lines
template <int L, int DMA>
__global__ __launch_bounds__(256) void k(const float* src, float* out, unsigned bytes, int iters, unsigned region) {
    __shared__ f32x4 smem[4 * 64 * 4];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (blockIdx.x * 4 + wave) * region;  // each wave walks its own region (bytes), cyclically
    unsigned off = (lane / L) * 128u + (lane % L) * 16u;
    const unsigned step = (64 / L) * 128u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned o = base + (off & (region - 1));
            if (DMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 4 + u) * 64), 16, o, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0);
                acc += __builtin_bit_cast(f32x4, v);
            }
            off += step;
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = smem[threadIdx.x];
    }
    if (acc[0] == 123.456f) out[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

template <int L, int DMA>
void run(const float* src, float* out, unsigned bytes, unsigned region, const char* what) {
    const int blocks = 256 * 4, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<L, DMA><<<blocks, 256>>>(src, out, bytes, iters, region);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<L, DMA><<<blocks, 256>>>(src, out, bytes, iters, region);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks * 4 * iters / 256.0;
    const double cyc = ms * 1e-3 * 2.1e9 / instr_per_cu;  // at ~2.1 GHz
    printf("%-28s lanes/line %d  region %6u B/wave: %.1f cycles per wave instruction per CU, %.1f B/clk/CU\n", what, L, region, cyc, 1024.0 / cyc);
}

int main() {
    const unsigned bytes = 256u << 20;
    float *src, *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096);
    hipMemset(src, 0, bytes);
    for (unsigned region : {8192u, 65536u}) {  // 8 KB/wave: L1-resident (32 KB per block); 64 KB/wave: L2-resident (64 MB total > L2? 1024 blocks x 256 KB = 256 MB: HBM/MALL)
        run<1, 0>(src, out, bytes, region, "register loads");
        run<2, 0>(src, out, bytes, region, "register loads");
        run<4, 0>(src, out, bytes, region, "register loads");
        run<8, 0>(src, out, bytes, region, "register loads");
        run<1, 1>(src, out, bytes, region, "LDS-DMA loads");
        run<2, 1>(src, out, bytes, region, "LDS-DMA loads");
        run<4, 1>(src, out, bytes, region, "LDS-DMA loads");
        run<8, 1>(src, out, bytes, region, "LDS-DMA loads");
    }
    return 0;
}
