// Deformable convolution (DCNv1, deformable_groups = 1) for the reference's DCN backbones
// (src/modules/resnet.py:54-65,81-82,111-124,145-146: conv2_offset -> torchvision.ops.DeformConv2d), NHWC fp32.
//
// Lowering: the learned offsets only change WHERE the 3x3 taps sample, so the op is
//   cols[m][tap][c] = bilinear(x[n, :, :, c], ho*stride - pad + r + dy, wo*stride - pad + s + dx)   (this file, HBM-bound)
//   y[m][co]        = sum_{tap,c} cols[m][tap][c] * W[co][c][tap]                                      (igemm 1x1, MFMA)
// and the backward is the 1x1 data/weight gradients of the GEMM (igemm / wgrad kernels) plus the adjoint of the
// sampling: dx (scattered with float atomics: the only non-bit-reproducible kernel of the library) and
// d(offset) (a reduction over channels).  Sampling rule = torchvision's bilinear_interpolate: a sample outside
// (-1, H) x (-1, W) is zero; each corner contributes only if its index is inside the image.
// Offsets: channel 2k = dy, 2k+1 = dx of tap k = r*S + s, stored [M][off_stride] (off_stride >= 2*R*S).
#include "common.h"
#include <stdlib.h>

namespace {

struct DeformDims {
    int N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride;
};

// one 32-lane team per (output pixel m, tap k); lanes stride over the C/4 channel quads
struct Sample {
    bool inside;
    int y0, x0;
    float ly, lx;
    bool ok[4];
};

template <int AT = 0>
__device__ __forceinline__ Sample sample_of(const DeformDims& d, const void* __restrict__ offset, int m, int k, int& n, int& ho,
                                            int& wo) {
    const int HWo = d.Ho * d.Wo;
    n = m / HWo;
    const int rem = m - n * HWo;
    ho = rem / d.Wo;
    wo = rem - ho * d.Wo;
    const int r = k / d.S, s = k - r * d.S;
    const float y = (float)(ho * d.stride - d.pad + r) + dbn_ld1<AT>(offset, (long)m * d.off_stride + 2 * k);
    const float x = (float)(wo * d.stride - d.pad + s) + dbn_ld1<AT>(offset, (long)m * d.off_stride + 2 * k + 1);
    Sample sp;
    sp.inside = y > -1.f && y < (float)d.H && x > -1.f && x < (float)d.W;
    const float fy = floorf(y), fx = floorf(x);
    sp.y0 = (int)fy;
    sp.x0 = (int)fx;
    sp.ly = y - fy;
    sp.lx = x - fx;
    const bool ya = sp.y0 >= 0 && sp.y0 <= d.H - 1, yb = sp.y0 + 1 >= 0 && sp.y0 + 1 <= d.H - 1;
    const bool xa = sp.x0 >= 0 && sp.x0 <= d.W - 1, xb = sp.x0 + 1 >= 0 && sp.x0 + 1 <= d.W - 1;
    sp.ok[0] = sp.inside && ya && xa;
    sp.ok[1] = sp.inside && ya && xb;
    sp.ok[2] = sp.inside && yb && xa;
    sp.ok[3] = sp.inside && yb && xb;
    return sp;
}

// AT: storage type of x, offset and cols (16-bit: BASELINE configs[3] in native bf16)
template <int AT>
__global__ __launch_bounds__(256) void deform_im2col_kernel(const void* __restrict__ x, const void* __restrict__ offset,
                                                            void* __restrict__ cols, DeformDims d, long teams) {
    const int lane = threadIdx.x & 31;
    const int RS = d.R * d.S, c4n = d.C >> 2;
    for (long t = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 5; t < teams; t += ((long)gridDim.x * blockDim.x) >> 5) {
        const int m = (int)(t / RS), k = (int)(t - (long)m * RS);
        int n, ho, wo;
        const Sample sp = sample_of<AT>(d, offset, m, k, n, ho, wo);
        const float w00 = (1.f - sp.ly) * (1.f - sp.lx), w01 = (1.f - sp.ly) * sp.lx, w10 = sp.ly * (1.f - sp.lx), w11 = sp.ly * sp.lx;
        const long base = ((long)n * d.H + sp.y0) * d.W + sp.x0;  // pixel index of corner (y0, x0); only dereferenced if ok
        const long out4 = ((long)m * RS + k) * c4n;
        for (int c4 = lane; c4 < c4n; c4 += 32) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v00 = sp.ok[0] ? dbn_ld4<AT>(x, base * c4n + c4) : z;
            const f32x4 v01 = sp.ok[1] ? dbn_ld4<AT>(x, (base + 1) * c4n + c4) : z;
            const f32x4 v10 = sp.ok[2] ? dbn_ld4<AT>(x, (base + d.W) * c4n + c4) : z;
            const f32x4 v11 = sp.ok[3] ? dbn_ld4<AT>(x, (base + d.W + 1) * c4n + c4) : z;
            dbn_st4<AT>(cols, out4 + c4, w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11);
        }
    }
}

__global__ __launch_bounds__(256) void deform_col2im_kernel(const float* __restrict__ dcols, const float* __restrict__ x,
                                                            const float* __restrict__ offset, float* __restrict__ dx,
                                                            float* __restrict__ doffset, DeformDims d, long teams) {
    const int lane = threadIdx.x & 31;
    const int RS = d.R * d.S, c4n = d.C >> 2;
    for (long t = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 5; t < teams; t += ((long)gridDim.x * blockDim.x) >> 5) {
        const int m = (int)(t / RS), k = (int)(t - (long)m * RS);
        int n, ho, wo;
        const Sample sp = sample_of(d, offset, m, k, n, ho, wo);
        const float hy = 1.f - sp.ly, hx = 1.f - sp.lx;
        const float w00 = hy * hx, w01 = hy * sp.lx, w10 = sp.ly * hx, w11 = sp.ly * sp.lx;
        const long base = ((long)n * d.H + sp.y0) * d.W + sp.x0;
        const f32x4* g4 = reinterpret_cast<const f32x4*>(dcols + ((long)m * RS + k) * d.C);
        float gy = 0.f, gx = 0.f;  // d(sample)/d(y), d(sample)/d(x) contracted with the column gradient
        for (int c4 = lane; c4 < c4n; c4 += 32) {
            const f32x4 g = g4[c4];
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v00 = sp.ok[0] ? reinterpret_cast<const f32x4*>(x + base * d.C)[c4] : z;
            const f32x4 v01 = sp.ok[1] ? reinterpret_cast<const f32x4*>(x + (base + 1) * d.C)[c4] : z;
            const f32x4 v10 = sp.ok[2] ? reinterpret_cast<const f32x4*>(x + (base + d.W) * d.C)[c4] : z;
            const f32x4 v11 = sp.ok[3] ? reinterpret_cast<const f32x4*>(x + (base + d.W + 1) * d.C)[c4] : z;
            const f32x4 dvy = hx * (v10 - v00) + sp.lx * (v11 - v01);
            const f32x4 dvx = hy * (v01 - v00) + sp.ly * (v11 - v10);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gy += g[e] * dvy[e];
                gx += g[e] * dvx[e];
                const int c = 4 * c4 + e;
                if (sp.ok[0]) atomicAdd(dx + base * d.C + c, w00 * g[e]);
                if (sp.ok[1]) atomicAdd(dx + (base + 1) * d.C + c, w01 * g[e]);
                if (sp.ok[2]) atomicAdd(dx + (base + d.W) * d.C + c, w10 * g[e]);
                if (sp.ok[3]) atomicAdd(dx + (base + d.W + 1) * d.C + c, w11 * g[e]);
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            gy += __shfl_xor(gy, o, 64);
            gx += __shfl_xor(gx, o, 64);
        }
        if (lane == 0) {
            doffset[(long)m * d.off_stride + 2 * k] = gy;
            doffset[(long)m * d.off_stride + 2 * k + 1] = gx;
        }
    }
}

// Tiled adjoint of the sampling.  The 9 taps of neighbouring output pixels land on the same few input pixels, so a
// workgroup owns a T x T tile of output pixels of one image and one chunk of 32 channels, accumulates their corner
// contributions in an LDS patch of the input region (tile footprint + HALO pixels for the learned offsets; native
// ds_add_f32) and flushes each patch element with ONE global atomic — 12x fewer global atomics than the scatter per
// sample.  Samples that leave the patch (|offset| > HALO) fall back to global atomics.  d(offset) sums over channels:
// a team reduction per (pixel, tap) and one atomic per channel chunk (doffset is zeroed by the caller).
constexpr int COL2IM_T = 8, COL2IM_HALO = 2, COL2IM_CC = 32;

// AT: storage type of dcols, x and offset; dx and doffset are ALWAYS fp32 (they are accumulated with float atomics: in 16-bit
// storage the caller gives fp32 scratch and rounds once afterwards, dbn_cast_f32)
template <int AT>
__global__ __launch_bounds__(256) void deform_col2im_tiled_kernel(const void* __restrict__ dcols, const void* __restrict__ x,
                                                                  const void* __restrict__ offset, float* __restrict__ dx,
                                                                  float* __restrict__ doffset, DeformDims d, int tiles_x, int tiles_y,
                                                                  int PD) {
    extern __shared__ float patch[];  // [PD][PD][CC]
    constexpr int T = COL2IM_T, CC = COL2IM_CC;
    const int lane = threadIdx.x & 31, team = threadIdx.x >> 5;
    const int chunk = blockIdx.y, c = chunk * CC + lane;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int oy0 = ty * T, ox0 = tx * T;                                                       // first output pixel of the tile
    const int py0 = oy0 * d.stride - d.pad - COL2IM_HALO, px0 = ox0 * d.stride - d.pad - COL2IM_HALO;  // patch origin (input coords)
    for (int i = threadIdx.x; i < PD * PD * CC; i += blockDim.x) patch[i] = 0.f;
    __syncthreads();
    const int RS = d.R * d.S;
    const bool cok = c < d.C;
    for (int pair = team; pair < T * T * RS; pair += 8) {
        const int pi = pair / RS, k = pair - pi * RS;
        const int ho = oy0 + pi / T, wo = ox0 + pi % T;
        if (ho >= d.Ho || wo >= d.Wo) continue;
        const int m = (n * d.Ho + ho) * d.Wo + wo;
        int n_, ho_, wo_;
        const Sample sp = sample_of<AT>(d, offset, m, k, n_, ho_, wo_);
        if (!sp.inside) continue;  // zero sample, zero gradients (doffset stays 0)
        const float hy = 1.f - sp.ly, hx = 1.f - sp.lx;
        const float wgt[4] = {hy * hx, hy * sp.lx, sp.ly * hx, sp.ly * sp.lx};
        const float g = cok ? dbn_ld1<AT>(dcols, ((long)m * RS + k) * d.C + c) : 0.f;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = sp.y0 + (q >> 1), xx = sp.x0 + (q & 1);
            v[q] = (sp.ok[q] && cok) ? dbn_ld1<AT>(x, (((long)n * d.H + yy) * d.W + xx) * d.C + c) : 0.f;
            if (sp.ok[q] && cok) {
                const int ry = yy - py0, rx = xx - px0;
                const float add = wgt[q] * g;
                if ((unsigned)ry < (unsigned)PD && (unsigned)rx < (unsigned)PD)
                    atomicAdd(&patch[(ry * PD + rx) * CC + lane], add);
                else
                    atomicAdd(dx + (((long)n * d.H + yy) * d.W + xx) * d.C + c, add);
            }
        }
        float gy = g * (hx * (v[2] - v[0]) + sp.lx * (v[3] - v[1]));
        float gx = g * (hy * (v[1] - v[0]) + sp.ly * (v[3] - v[2]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            gy += __shfl_xor(gy, o, 64);
            gx += __shfl_xor(gx, o, 64);
        }
        if (lane == 0) {
            atomicAdd(doffset + (long)m * d.off_stride + 2 * k, gy);
            atomicAdd(doffset + (long)m * d.off_stride + 2 * k + 1, gx);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PD * PD * CC; i += blockDim.x) {
        const float vsum = patch[i];
        if (vsum == 0.f) continue;
        const int cc = i % CC, pix = i / CC;
        const int yy = py0 + pix / PD, xx = px0 + pix % PD;
        if ((unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W && chunk * CC + cc < d.C)
            atomicAdd(dx + (((long)n * d.H + yy) * d.W + xx) * d.C + chunk * CC + cc, vsum);
    }
}

// fp32 -> activation storage type (one rounding)
template <int AT>
__global__ void cast_f32_kernel(const float* __restrict__ src, void* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        dbn_st4<AT>(dst, i, reinterpret_cast<const f32x4*>(src)[i]);
}

// dst[o][t][c] = src[o][c][t] (to_ohwi) or dst[o][c][t] = scale * src[o][t][c]: weight layout between OIHW and the
// GEMM's (tap, channel) column order
__global__ void permute_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int C, int T, int to_ohwi,
                                      float scale) {
    const long total = (long)O * C * T;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i / ((long)C * T));
        const int rem = (int)(i - (long)o * C * T);
        if (to_ohwi) {
            const int t = rem / C, c = rem - t * C;
            dst[i] = scale * src[((long)o * C + c) * T + t];
        } else {
            const int c = rem / T, t = rem - c * T;
            dst[i] = scale * src[((long)o * T + t) * C + c];
        }
    }
}

bool dims_ok(const DeformDims& d) {
    return d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.C % 4 == 0 && d.R > 0 && d.S > 0 && d.stride > 0 && d.pad >= 0 &&
           d.Ho == (d.H + 2 * d.pad - d.R) / d.stride + 1 && d.Wo == (d.W + 2 * d.pad - d.S) / d.stride + 1 &&
           d.off_stride >= 2 * d.R * d.S && (long)d.N * d.Ho * d.Wo * d.R * d.S < (1L << 31);
}

}  // namespace

extern "C" {

// cols[N*Ho*Wo][R*S][C] = bilinear samples of x[N,H,W,C] at the offset tap positions
int dbn_deform_im2col_t(int at, const void* x, const void* offset, void* cols, int N, int H, int W, int C, int Ho, int Wo, int R, int S,
                        int stride, int pad, int off_stride, void* stream) {
    const DeformDims d{N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride};
    DBN_REQUIRE(x && offset && cols && dims_ok(d));
    const long teams = (long)N * Ho * Wo * R * S;
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(deform_im2col_kernel<AT>, dim3(dbn_grid(teams * 32, 256, 1 << 16)), dim3(256), 0,
                                           (hipStream_t)stream, x, offset, cols, d, teams));
    return dbn_status();
}
int dbn_deform_im2col(const float* x, const float* offset, float* cols, int N, int H, int W, int C, int Ho, int Wo, int R, int S,
                      int stride, int pad, int off_stride, void* stream) {
    return dbn_deform_im2col_t(0, x, offset, cols, N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride, stream);
}

// dst (activation type `at`) = src (fp32), n % 4 == 0: the one rounding of results that had to be accumulated in fp32
int dbn_cast_f32(int at, const float* src, void* dst, long n, void* stream) {
    DBN_REQUIRE(src && dst && n > 0 && n % 4 == 0);
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(cast_f32_kernel<AT>, dim3(dbn_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4));
    return dbn_status();
}

// adjoint of the sampling: dx[N,H,W,C] += scatter(dcols) (float atomics; the caller initialises dx),
// doffset[N*Ho*Wo][off_stride]: channels 0..2RS-1 written, the rest set to zero
// dcols / x / offset in the activation type `at`; dx and doffset are fp32 in every mode (float atomics) — 16-bit callers pass
// fp32 scratch and round with dbn_cast_f32
int dbn_deform_col2im_t(int at, const void* dcols, const void* x, const void* offset, float* dx, float* doffset, int N, int H, int W,
                        int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream) {
    const DeformDims d{N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride};
    DBN_REQUIRE(dcols && x && offset && dx && doffset && dims_ok(d));
    hipStream_t st = (hipStream_t)stream;
    static const int tiled = dbn_env_int("DBN_COL2IM_TILED", 1);  // (0: per-sample scatter; -DDBN_EXPERIMENTS builds only)
    const int PD = (COL2IM_T - 1) * stride + R + 2 * COL2IM_HALO;  // patch edge; R == S for every DCN layer of the reference
    const size_t lds = (size_t)PD * PD * COL2IM_CC * sizeof(float);
    if (tiled && R == S && lds <= 64 * 1024) {
        if (hipMemsetAsync(doffset, 0, (size_t)N * Ho * Wo * off_stride * sizeof(float), st) != hipSuccess) return dbn_status();
        const int tiles_x = dbn_ceil_div(Wo, COL2IM_T), tiles_y = dbn_ceil_div(Ho, COL2IM_T);
        DBN_DISPATCH_AT(at, hipLaunchKernelGGL(deform_col2im_tiled_kernel<AT>, dim3(N * tiles_y * tiles_x, dbn_ceil_div(C, COL2IM_CC)),
                                               dim3(256), lds, st, dcols, x, offset, dx, doffset, d, tiles_x, tiles_y, PD));
        return dbn_status();
    }
    DBN_REQUIRE(at == 0);  // the per-sample scatter below exists for fp32 tensors only
    if (off_stride > 2 * R * S &&
        hipMemsetAsync(doffset, 0, (size_t)N * Ho * Wo * off_stride * sizeof(float), st) != hipSuccess)
        return dbn_status();
    const long teams = (long)N * Ho * Wo * R * S;
    hipLaunchKernelGGL(deform_col2im_kernel, dim3(dbn_grid(teams * 32, 256, 1 << 16)), dim3(256), 0, st, (const float*)dcols,
                       (const float*)x, (const float*)offset, dx, doffset, d, teams);
    return dbn_status();
}
int dbn_deform_col2im(const float* dcols, const float* x, const float* offset, float* dx, float* doffset, int N, int H, int W, int C,
                      int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream) {
    return dbn_deform_col2im_t(0, dcols, x, offset, dx, doffset, N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride, stream);
}

// to_ohwi = 1: dst[O][T][C] = scale * src[O][C][T] (OIHW -> GEMM column order); 0: the inverse
int dbn_permute_weight(const float* src, float* dst, int O, int C, int T, int to_ohwi, float scale, void* stream) {
    DBN_REQUIRE(src && dst && O > 0 && C > 0 && T > 0);
    hipLaunchKernelGGL(permute_weight_kernel, dim3(dbn_grid((long)O * C * T)), dim3(256), 0, (hipStream_t)stream, src, dst, O, C, T,
                       to_ohwi, scale);
    return dbn_status();
}

}  // extern "C"
