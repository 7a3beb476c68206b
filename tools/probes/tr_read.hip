// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds element index e at 16-bit slot e; every lane passes its own byte
// address; print what each lane receives.  Build: hipcc --offload-arch=gfx950 -O3 -w tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned l = threadIdx.x;
    const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                   // lane l -> 8 consecutive bytes (4 elements) at element 4l
    else if (mode == 1) addr = ((l & 3) * 8) + (l >> 2) * 256;     // row-major [row = l/4][128 elements]: lane (r, c) -> row r, cols 4c..4c+3
    else addr = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;  // per 16-lane group g: rows (l&15)/4 of a [4][.] block at cols 16g + 4(l&3)
    u32x2 v;
    addr += base;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xFFFF; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xFFFF; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, 1024);
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, mode);
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        if (e1 != hipSuccess || e2 != hipSuccess) printf("launch %s sync %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
