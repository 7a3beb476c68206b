// Probe: does buffer_load ... lds write ZEROS to LDS for lanes whose buffer offset is out of range?  (The weight-gradient
// kernel relies on it for padding taps and pixel tails.)  Build: hipcc --offload-arch=gfx950 -O3 lds_dma_oob.hip -o lds_dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, unsigned bytes) {
    __shared__ f32x4 smem[256];
    smem[threadIdx.x] = f32x4{-7.f, -7.f, -7.f, -7.f};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    unsigned off = threadIdx.x * 16u;
    if (threadIdx.x & 1) off = 0xF8000000u;
    const unsigned wave_base = (threadIdx.x & ~63u) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)smem + wave_base), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 v = smem[threadIdx.x];
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    float *src, *out, h[1024], ho[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
    hipMalloc(&src, 4096); hipMalloc(&out, 4096);
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256>>>(src, out, 4096);
    hipMemcpy(ho, out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t)
        for (int e = 0; e < 4; ++e) {
            const float want = (t & 1) ? 0.f : h[t * 4 + e];
            if (ho[t * 4 + e] != want) { if (bad < 8) printf("lane %d e %d: got %g want %g\n", t, e, ho[t * 4 + e], want); ++bad; }
        }
    printf("LDS-DMA OOB probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK: out-of-range lanes wrote zeros", bad);
    return bad != 0;
}
