// Deformable convolution (DCNv1, deformable_groups = 1) for the reference's DCN backbones
// (src/modules/resnet.py:54-65,81-82,111-124,145-146: conv2_offset -> torchvision.ops.DeformConv2d), NHWC fp32.
//
// Lowering: the learned offsets only change WHERE the 3x3 taps sample, so the op is
//   cols[m][tap][c] = bilinear(x[n, :, :, c], ho*stride - pad + r + dy, wo*stride - pad + s + dx)   (this file, HBM-bound)
//   y[m][co]        = sum_{tap,c} cols[m][tap][c] * W[co][c][tap]                                      (igemm 1x1, MFMA)
// and the backward is the 1x1 data/weight gradients of the GEMM (igemm / wgrad kernels) plus the adjoint of the
// sampling: dx (a scatter, accumulated in 64-bit fixed point so that it is bit-reproducible) and d(offset) (a reduction
// over channels, likewise).  Sampling rule = torchvision's bilinear_interpolate: a sample outside
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

// ---- adjoint of the sampling: DETERMINISTIC (round 3) ---------------------------------------------------------------------------
// dx[n, y, x, c] receives w_corner * dcols[m, k, c] from every sample (m, k) one of whose four bilinear corners is (y, x); which
// samples those are depends on the learned offsets.  Rounds 1-2 scattered with float atomics (the library's only kernel whose
// summation order was not fixed).  Now every contribution is accumulated in FIXED POINT: c -> round(c * 2^k) as a 64-bit integer,
// with 2^k chosen per call from max |dcols| (a reduction pass: absmax_kernel) so that the largest possible contribution is 2^43 —
// integer addition is associative, so LDS and global integer atomics give the same bits in any order; the resolution is
// 2^-43 of the largest column gradient (5e-14 relative: far below the fp32 rounding of the result).  d(offset) — a sum over
// channels of g * d(bilinear)/d(y, x), bounded by 2 max|dcols| max|x| per term — takes the same route with its own scale.
// The 9 taps of neighbouring output pixels land on the same few input pixels, so a workgroup owns a T x T tile of output pixels
// of one image and one chunk of CC channels and accumulates its corner contributions in an LDS patch of the input region first
// (tile footprint + HALO pixels for the learned offsets; ds_add_u64), flushing each touched patch element with ONE global
// integer atomic; samples that leave the patch (|offset| > HALO) go to the global accumulator directly.
constexpr int COL2IM_T = 8, COL2IM_HALO = 2;

// Non-finite inputs: the old float atomics carried a NaN / Inf in dcols into dx; a fixed-point sum cannot (to_fixed(NaN) is 0).
// absmax therefore reports a NaN bit pattern (which orders above every finite value in atomicMax) as soon as ONE element is not
// finite, and the finish kernel then writes NaN to the whole of dx / doffset — a diverged step stays visible downstream.
// Resolution: every addend is rounded to 2^-43 of the GLOBAL max |dcols| (doffset: of 64 max|dcols| max|x|), so one outlier sets
// the absolute error floor of every element: |error| <= (#addends) * 2^-44 * max|dcols| — e.g. 1e6 * 6e-14 = 6e-8 per addend with
// a 1e6 outlier, against fp32 gradients of O(1e-3 .. 1) (tests/test_ops_gpu.py::test_deformable_col2im_non_finite_and_outliers).
constexpr unsigned DEFORM_NONFINITE = 0x7FC00000u;
template <int AT>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ x, long n4, unsigned* __restrict__ out) {
    float m = 0.f;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = dbn_ld4<AT>(x, i);
        const float a = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        bad |= !(fabsf(v[0]) <= 3.0e38f) | !(fabsf(v[1]) <= 3.0e38f) | !(fabsf(v[2]) <= 3.0e38f) | !(fabsf(v[3]) <= 3.0e38f);  // (fmaxf drops a NaN)
        m = fmaxf(a, m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const bool any_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) atomicMax(out, any_bad ? DEFORM_NONFINITE : __builtin_bit_cast(unsigned, m));  // non-negative floats order like their bit patterns
}

// 2^k with (largest magnitude, rounded up to a power of two) * 2^k = 2^43
__device__ __forceinline__ double fixed_scale(float vmax) {
    if (!(vmax > 0.f)) return 1.0;
    int e;
    frexpf(vmax, &e);  // vmax < 2^e
    return ldexp(1.0, 43 - e);
}
__device__ __forceinline__ unsigned long long to_fixed(float v, double scale) { return (unsigned long long)__double2ll_rn((double)v * scale); }

// AT: storage type of dcols, x and offset.  dx64 / doff64: 64-bit fixed-point accumulators (zeroed by the caller).
template <int AT, int CC>
__global__ __launch_bounds__(256) void deform_col2im_tiled_kernel(const void* __restrict__ dcols, const void* __restrict__ x,
                                                                  const void* __restrict__ offset, unsigned long long* __restrict__ dx64,
                                                                  unsigned long long* __restrict__ doff64,
                                                                  const unsigned* __restrict__ maxbits, DeformDims d, int tiles_x,
                                                                  int tiles_y, int PD) {
    extern __shared__ unsigned long long patch[];  // [PD][PD][CC]
    constexpr int T = COL2IM_T, TEAM = CC, TEAMS = 256 / CC;
    const int lane = threadIdx.x % TEAM, team = threadIdx.x / TEAM;
    const int chunk = blockIdx.y, c = chunk * CC + lane;
    const float gmax = __builtin_bit_cast(float, maxbits[0]), xmax = __builtin_bit_cast(float, maxbits[1]);
    const double sdx = fixed_scale(gmax), soff = fixed_scale(64.f * fmaxf(gmax, 1e-30f) * fmaxf(xmax, 1e-30f));
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int oy0 = ty * T, ox0 = tx * T;                                                       // first output pixel of the tile
    const int py0 = oy0 * d.stride - d.pad - COL2IM_HALO, px0 = ox0 * d.stride - d.pad - COL2IM_HALO;  // patch origin (input coords)
    for (int i = threadIdx.x; i < PD * PD * CC; i += blockDim.x) patch[i] = 0ull;
    __syncthreads();
    const int RS = d.R * d.S;
    const bool cok = c < d.C;
    for (int pair = team; pair < T * T * RS; pair += TEAMS) {
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
                const unsigned long long add = to_fixed(wgt[q] * g, sdx);
                if ((unsigned)ry < (unsigned)PD && (unsigned)rx < (unsigned)PD)
                    atomicAdd(&patch[(ry * PD + rx) * CC + lane], add);
                else
                    atomicAdd(dx64 + (((long)n * d.H + yy) * d.W + xx) * d.C + c, add);
            }
        }
        float gy = g * (hx * (v[2] - v[0]) + sp.lx * (v[3] - v[1]));
        float gx = g * (hy * (v[1] - v[0]) + sp.ly * (v[3] - v[2]));
#pragma unroll
        for (int o = TEAM / 2; o > 0; o >>= 1) {  // fixed shuffle tree inside the team
            gy += __shfl_xor(gy, o, 64);
            gx += __shfl_xor(gx, o, 64);
        }
        if (lane == 0) {
            atomicAdd(doff64 + ((long)m * RS + k) * 2, to_fixed(gy, soff));
            atomicAdd(doff64 + ((long)m * RS + k) * 2 + 1, to_fixed(gx, soff));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PD * PD * CC; i += blockDim.x) {
        const unsigned long long vsum = patch[i];
        if (vsum == 0ull) continue;
        const int cc = i % CC, pix = i / CC;
        const int yy = py0 + pix / PD, xx = px0 + pix % PD;
        if ((unsigned)yy < (unsigned)d.H && (unsigned)xx < (unsigned)d.W && chunk * CC + cc < d.C)
            atomicAdd(dx64 + (((long)n * d.H + yy) * d.W + xx) * d.C + chunk * CC + cc, vsum);
    }
}

// fixed point -> storage type: dx = [dx +] dx64 / 2^k,  doffset[m][0 .. 2RS) = doff64 / 2^k', the remaining channels zero
template <int AT>
__global__ void deform_col2im_finish_kernel(const unsigned long long* __restrict__ dx64, const unsigned long long* __restrict__ doff64,
                                            const unsigned* __restrict__ maxbits, void* __restrict__ dx, void* __restrict__ doffset,
                                            long ndx4, long M, int RS2, int off_stride, int accumulate) {
    const float gmax = __builtin_bit_cast(float, maxbits[0]), xmax = __builtin_bit_cast(float, maxbits[1]);
    const bool bad_g = maxbits[0] == DEFORM_NONFINITE, bad_x = maxbits[1] == DEFORM_NONFINITE;  // a non-finite dcols / x element (absmax_kernel)
    const float nan_ = __builtin_bit_cast(float, DEFORM_NONFINITE);
    const double idx = 1.0 / fixed_scale(gmax), ioff = 1.0 / fixed_scale(64.f * fmaxf(gmax, 1e-30f) * fmaxf(xmax, 1e-30f));
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (long i = t0; i < ndx4; i += stride) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bad_g ? nan_ : (float)((double)(long long)dx64[4 * i + e] * idx);
        if (accumulate) v += dbn_ld4<AT>(dx, i);
        dbn_st4<AT>(dx, i, v);
    }
    const long noff4 = M * (off_stride / 4);
    for (long i = t0; i < noff4; i += stride) {
        const long m = i / (off_stride / 4);
        const int ch = (int)(i - m * (off_stride / 4)) * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ch + e < RS2 ? ((bad_g || bad_x) ? nan_ : (float)((double)(long long)doff64[m * RS2 + ch + e] * ioff)) : 0.f;
        dbn_st4<AT>(doffset, i, v);
    }
}

// ---- adjoint of the sampling as a GATHER (round 5): deterministic in plain fp32, no atomics, no fixed point ----------------------------
// Round 3's tiled scatter above is exact but slow (configs[3] trace, round 5: 13.7 ms of the 72 ms fp32 step in deform_col2im_tiled_kernel
// + 3.0 ms in its two absmax passes: 64-bit LDS atomics, one 4-byte load per lane).  The adjoint does not need a scatter:
//   doffset[m][k]  = sum_c dcols[m][k][c] * d(bilinear)/d(y, x)      — a reduction over the channels of ONE sample: a team per (m, k)
//   dx[n][y][x][c] = sum over the samples (m, k) one of whose four corners is (y, x) of  w_corner * dcols[m][k][c]
// and WHICH samples can touch input pixel (y, x) is bounded by the largest learned offset: with E = ceil(max |offset|) (a device-side
// maximum, taken by deform_bbox_kernel on its way) only taps whose undeformed position (ho*stride - pad + r, wo*stride - pad + s) lies within
// E of (y, x) can — (R + 2E)^2 output pixels at stride 1.  A 32-lane team owns an input pixel, in three nested stages: its lanes test 32
// output pixels of that window at a time against the pixel's BOX (deform_bbox_kernel: the input rows / columns its nine samples touch —
// one 16-byte load and four comparisons, so a large window, i.e. ONE large offset somewhere in the map, costs (R + 2E)^2 / 32 cheap rounds,
// not 9x that many position tests); the taps of the pixels that pass are tested three pixels at a time, one lane per (pixel, tap) (offset
// pair -> position -> is (y, x) one of the corners, with which weight — the same float expressions as sample_of / deform_im2col_kernel);
// the hits are walked in (pixel, tap) order (ballot + shuffles: no memory), every lane adding w * dcols[m][k][its channel quads].  One fixed summation order per output element: bit-reproducible, and independent of the
// grid.  Larger offsets only widen the window (E is read on the device: no host synchronisation, no fallback path).  A workgroup owns 8 rows x 2
// columns of input pixels (team = one row), neighbouring workgroups follow in x, so the four corners' re-reads of a dcols row meet in L1 / L2.
// Non-finite values: a NaN / Inf in dcols reaches exactly the dx / doffset elements its sample touches (as float atomics would);
// a NaN offset makes its sample "outside" (contributes nothing) and does not widen the window.
constexpr int GATHER_TILE = 8, GATHER_TILE_X = 2;  // input rows (= teams) x columns (walked by each team) per workgroup

template <int AT>
__global__ __launch_bounds__(256) void deform_doffset_kernel(const void* __restrict__ dcols, const void* __restrict__ x,
                                                             const void* __restrict__ offset, void* __restrict__ doffset, DeformDims d,
                                                             long teams) {
    const int lane = threadIdx.x & 31;
    const int RS = d.R * d.S, c4n = d.C >> 2;
    for (long t = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 5; t < teams; t += ((long)gridDim.x * blockDim.x) >> 5) {
        const int m = (int)(t / RS), k = (int)(t - (long)m * RS);
        int n, ho, wo;
        const Sample sp = sample_of<AT>(d, offset, m, k, n, ho, wo);
        float gy = 0.f, gx = 0.f;
        if (sp.inside) {  // (team-uniform) an outside sample is the constant zero: zero gradients
            const float hy = 1.f - sp.ly, hx = 1.f - sp.lx;
            const long base = ((long)n * d.H + sp.y0) * d.W + sp.x0;
            const long g4 = ((long)m * RS + k) * c4n;
            for (int c4 = lane; c4 < c4n; c4 += 32) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 g = dbn_ld4<AT>(dcols, g4 + c4);
                const f32x4 v0 = sp.ok[0] ? dbn_ld4<AT>(x, base * c4n + c4) : z;
                const f32x4 v1 = sp.ok[1] ? dbn_ld4<AT>(x, (base + 1) * c4n + c4) : z;
                const f32x4 v2 = sp.ok[2] ? dbn_ld4<AT>(x, (base + d.W) * c4n + c4) : z;
                const f32x4 v3 = sp.ok[3] ? dbn_ld4<AT>(x, (base + d.W + 1) * c4n + c4) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gy += g[e] * (hx * (v2[e] - v0[e]) + sp.lx * (v3[e] - v1[e]));
                    gx += g[e] * (hy * (v1[e] - v0[e]) + sp.ly * (v3[e] - v2[e]));
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {  // fixed shuffle tree inside the team
                gy += __shfl_xor(gy, o, 64);
                gx += __shfl_xor(gx, o, 64);
            }
        }
        if (lane == 0) {
            dbn_st1<AT>(doffset, (long)m * d.off_stride + 2 * k, gy);
            dbn_st1<AT>(doffset, (long)m * d.off_stride + 2 * k + 1, gx);
        }
        if (k == 0)  // the padding channels of the offset map (its conv runs on 64): zero gradient
            for (int ch = 2 * RS + lane; ch < d.off_stride; ch += 32) dbn_st1<AT>(doffset, (long)m * d.off_stride + ch, 0.f);
    }
}

// Per output pixel m: the box of input pixels its R*S samples touch (rows ymin..ymax, columns xmin..xmax; empty: ymin > ymax), from the
// same float expressions as sample_of — the gather below tests ONE box per output pixel before it looks at that pixel's taps.
template <int AT>
__global__ __launch_bounds__(256) void deform_bbox_kernel(const void* __restrict__ offset, int4* __restrict__ bbox, DeformDims d, int M,
                                                          unsigned* __restrict__ maxbits) {
    // ... and, while every offset passes through a register anyway, max |offset| over the 2*R*S real channels (absmax_kernel's contract:
    // the bit pattern, 0x7FC00000 as soon as one is not finite) — a separate pass over the 64-channel map cost 15-60 us per layer in the step
    const int RS = d.R * d.S;
    float om = 0.f;
    bool bad = false;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        int ymin = 1 << 30, ymax = -(1 << 30), xmin = 1 << 30, xmax = -(1 << 30);
        for (int k = 0; k < RS; ++k) {
            int n, ho, wo;
            const Sample sp = sample_of<AT>(d, offset, m, k, n, ho, wo);
            const float oy = fabsf(dbn_ld1<AT>(offset, (long)m * d.off_stride + 2 * k)), ox = fabsf(dbn_ld1<AT>(offset, (long)m * d.off_stride + 2 * k + 1));
            bad |= !(oy <= 3.0e38f) | !(ox <= 3.0e38f);  // (fmaxf drops a NaN)
            om = fmaxf(om, fmaxf(oy, ox));
            if (sp.inside) {
                ymin = min(ymin, sp.y0);
                ymax = max(ymax, sp.y0 + 1);
                xmin = min(xmin, sp.x0);
                xmax = max(xmax, sp.x0 + 1);
            }
        }
        bbox[m] = int4{ymin, ymax, xmin, xmax};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) om = fmaxf(om, __shfl_xor(om, o, 64));
    const bool any_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, any_bad ? DEFORM_NONFINITE : __builtin_bit_cast(unsigned, om));  // non-negative floats order like their bit patterns
}

// NQ: channel quads per lane (C <= 128 * NQ).  The kernel is a chain of dependent loads (box -> offsets -> dcols rows) per pixel, so what
// it has to offer the memory system is parallelism: a team walks only GATHER_TILE_X pixels (10 000+ workgroups at 100^2), the boxes of three
// window rounds are loaded together, the next tap group's offsets are loaded before the current group's hits are walked, and 8 (NQ <= 2) or 4
// dcols rows are in flight per batch.
template <int AT, int NQ>
__global__ __launch_bounds__(256) void deform_dx_gather_kernel(const void* __restrict__ dcols, const void* __restrict__ offset,
                                                               const int4* __restrict__ bbox, void* __restrict__ dx,
                                                               const unsigned* __restrict__ maxbits, DeformDims d, int tiles_x, int tiles_y,
                                                               int accumulate) {
    constexpr int HB = NQ <= 2 ? 8 : 4;  // dcols rows per batch
    const int lane = threadIdx.x & 31, team = threadIdx.x >> 5, half = (threadIdx.x >> 5) & 1, lbase = 32 * half;
    const int RS = d.R * d.S, c4n = d.C >> 2;
    int bid = dbn_xcd_remap(blockIdx.x, gridDim.x);  // (an XCD owns a contiguous run of tiles: the corner re-reads of a dcols row meet in ITS L2)
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int y = ty * GATHER_TILE + team;
    // E = ceil(max |offset|): |base - pixel| <= E for every sample that can touch the pixel.  A non-finite maximum (a NaN / Inf offset: its
    // sample is outside by the comparison rules, or infinitely far) must not widen the window: the finite offsets' bound is not known then, so
    // the whole map is searched — correct, slow, and only in a step that has diverged anyway.
    const unsigned mb = maxbits[0];
    const float omax = __builtin_bit_cast(float, mb);
    const int emax = max(d.H, d.W) + d.R;
    const int E = (mb == DEFORM_NONFINITE || !(omax < (float)emax)) ? emax : (int)ceilf(omax);
    if (y >= d.H) return;  // (team-uniform; no barrier below)
    const int ay = y - E + d.pad - (d.R - 1), by = y + E + d.pad;
    const int ho_lo = ay <= 0 ? 0 : (ay + d.stride - 1) / d.stride, ho_hi = min(d.Ho - 1, by / d.stride);
    const int nho = ho_hi - ho_lo + 1;
    const int PPR = 32 / RS;  // output pixels whose taps are tested side by side (three for 3 x 3)
    const int slot = lane / RS, k = lane - slot * RS;
    const int r = k / d.S, s_ = k - r * d.S;
    const int HWo = d.Ho * d.Wo;
    for (int ix = 0; ix < GATHER_TILE_X; ++ix) {
        const int x = tx * GATHER_TILE_X + ix;
        if (x >= d.W) break;
        const int ax = x - E + d.pad - (d.S - 1), bx = x + E + d.pad;
        const int wo_lo = ax <= 0 ? 0 : (ax + d.stride - 1) / d.stride, wo_hi = min(d.Wo - 1, bx / d.stride);
        const int nwo = wo_hi - wo_lo + 1;
        const int npix = (nho > 0 && nwo > 0) ? nho * nwo : 0;
        f32x4 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // one tap group: up to PPR passing pixels taken off the mask `pm` (their m from lane registers `m_l`), this lane's (pixel, tap)
        // and its offset pair, loaded but not yet used
        struct Group {
            int m_s;
            float oy, ox;
        };
        auto take_group = [&](unsigned& pm, int m_l) {
            Group g;
            g.m_s = -1;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u < PPR) {
                    const bool have = pm != 0u;
                    const int src = have ? __builtin_ctz(pm) : 0;
                    pm &= pm - 1u;
                    const int mm = __shfl(m_l, lbase + src, 64);
                    if (have && slot == u) g.m_s = mm;
                }
            }
            g.oy = g.ox = 0.f;
            if (g.m_s >= 0) {
                g.oy = dbn_ld1<AT>(offset, (long)g.m_s * d.off_stride + 2 * k);
                g.ox = dbn_ld1<AT>(offset, (long)g.m_s * d.off_stride + 2 * k + 1);
            }
            return g;
        };
        auto walk_group = [&](const Group& g) {
            float w = 0.f;
            int mk = 0;
            if (g.m_s >= 0) {
                const int rem = g.m_s - n * HWo;
                const int ho = rem / d.Wo, wo = rem - ho * d.Wo;
                const float py = (float)(ho * d.stride - d.pad + r) + g.oy;
                const float px = (float)(wo * d.stride - d.pad + s_) + g.ox;
                if (py > -1.f && py < (float)d.H && px > -1.f && px < (float)d.W) {  // (sample_of's `inside`)
                    const float fy = floorf(py), fx = floorf(px);
                    const int y0 = (int)fy, x0 = (int)fx;
                    const float ly = py - fy, lx = px - fx;
                    const float wy = y == y0 ? 1.f - ly : (y == y0 + 1 ? ly : 0.f);
                    const float wx = x == x0 ? 1.f - lx : (x == x0 + 1 ? lx : 0.f);
                    w = wy * wx;  // (the corner weights of deform_im2col_kernel: (1 - ly | ly) * (1 - lx | lx))
                    mk = g.m_s * RS + k;
                }
            }
            unsigned hits = (unsigned)(__ballot(w != 0.f) >> lbase);
            while (hits) {  // (team-uniform) the hits in lane = (pixel, tap) order, HB rows in flight
                int hm[HB];
                float hw[HB];
#pragma unroll
                for (int u = 0; u < HB; ++u) {
                    const bool have = hits != 0u;
                    const int src = have ? __builtin_ctz(hits) : 0;
                    hits &= hits - 1u;
                    const int mm = __shfl(mk, lbase + src, 64);
                    const float ww = __shfl(w, lbase + src, 64);
                    hm[u] = have ? mm : -1;
                    hw[u] = have ? ww : 0.f;
                }
                f32x4 gq[HB][NQ];
#pragma unroll
                for (int u = 0; u < HB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c4 = lane + 32 * q;
                        gq[u][q] = (hm[u] >= 0 && c4 < c4n) ? dbn_ld4<AT>(dcols, (long)hm[u] * c4n + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int u = 0; u < HB; ++u)
                    if (hm[u] >= 0) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[q] += hw[u] * gq[u][q];
                    }
            }
        };
        for (int i0 = 0; i0 < npix; i0 += 96) {
            // stage 1: up to 96 output pixels of the window against their boxes, the three loads issued together
            int m_l[3];
            int4 bb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int i = i0 + 32 * t + lane;
                const int iho = i / nwo, iwo = i - iho * nwo;
                m_l[t] = i < npix ? (n * d.Ho + ho_lo + iho) * d.Wo + wo_lo + iwo : -1;
                bb[t] = m_l[t] >= 0 ? bbox[m_l[t]] : int4{1, 0, 1, 0};
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (i0 + 32 * t >= npix) break;  // (team-uniform)
                const bool pass = m_l[t] >= 0 && y >= bb[t].x && y <= bb[t].y && x >= bb[t].z && x <= bb[t].w;
                unsigned pm = (unsigned)(__ballot(pass) >> lbase);
                if (!pm) continue;
                // stage 2 + 3: the passing pixels' taps, PPR pixels per group; the next group's offsets are in flight while this one's hits are walked
                Group cur = take_group(pm, m_l[t]);
                while (true) {
                    const bool more = pm != 0u;  // (team-uniform)
                    Group nxt;
                    nxt.m_s = -1;
                    nxt.oy = nxt.ox = 0.f;
                    if (more) nxt = take_group(pm, m_l[t]);
                    walk_group(cur);
                    if (!more) break;
                    cur = nxt;
                }
            }
        }
        const long o4 = (((long)n * d.H + y) * d.W + x) * c4n;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c4 = lane + 32 * q;
            if (c4 < c4n) dbn_st4<AT>(dx, o4 + c4, accumulate ? acc[q] + dbn_ld4<AT>(dx, o4 + c4) : acc[q]);
        }
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

// Scratch of dbn_deform_col2im_t: two maxima + the 64-bit fixed-point accumulators of dx and of the 2*R*S offset channels
long dbn_deform_col2im_ws_bytes(int N, int H, int W, int C, int Ho, int Wo, int R, int S) {
    return 16L + 8L * ((long)N * H * W * C + (long)N * Ho * Wo * 2 * R * S);
}

// adjoint of the sampling, deterministic (fixed-point accumulation, see above): dx[N,H,W,C] = [dx +] scatter(dcols),
// doffset[N*Ho*Wo][off_stride]: channels 0 .. 2RS-1 written, the rest set to zero.  dcols / x / offset / dx / doffset in the
// activation type `at`; ws: dbn_deform_col2im_ws_bytes(...) bytes.  accumulate = 1: dx holds a gradient already (it is added).
int dbn_deform_col2im_t(int at, const void* dcols, const void* x, const void* offset, void* dx, void* doffset, int accumulate, void* ws,
                        int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream) {
    const DeformDims d{N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride};
    DBN_REQUIRE(dcols && x && offset && dx && doffset && ws && dims_ok(d) && R == S && off_stride % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)N * Ho * Wo, ndx = (long)N * H * W * C, ncols = M * R * S * C;
    if (hipMemsetAsync(ws, 0, (size_t)dbn_deform_col2im_ws_bytes(N, H, W, C, Ho, Wo, R, S), st) != hipSuccess) return dbn_status();
    unsigned* maxbits = reinterpret_cast<unsigned*>(ws);
    unsigned long long* dx64 = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + 16);
    unsigned long long* doff64 = dx64 + ndx;
    const int PD = (COL2IM_T - 1) * stride + R + 2 * COL2IM_HALO;  // patch edge
    const int tiles_x = dbn_ceil_div(Wo, COL2IM_T), tiles_y = dbn_ceil_div(Ho, COL2IM_T);
    // channels per workgroup: 32, or 16 where the 64-bit patch of a stride-2 layer would not fit 64 KB
    const bool cc32 = (size_t)PD * PD * 32 * 8 <= 64 * 1024;
    DBN_REQUIRE(cc32 || (size_t)PD * PD * 16 * 8 <= 64 * 1024);
    DBN_DISPATCH_AT(at, {
        hipLaunchKernelGGL(absmax_kernel<AT>, dim3(dbn_grid(ncols / 4, 256, 2048)), dim3(256), 0, st, dcols, ncols / 4, maxbits);
        hipLaunchKernelGGL(absmax_kernel<AT>, dim3(dbn_grid(ndx / 4, 256, 2048)), dim3(256), 0, st, x, ndx / 4, maxbits + 1);
        if (cc32)
            hipLaunchKernelGGL((deform_col2im_tiled_kernel<AT, 32>), dim3(N * tiles_y * tiles_x, dbn_ceil_div(C, 32)), dim3(256),
                               (size_t)PD * PD * 32 * 8, st, dcols, x, offset, dx64, doff64, maxbits, d, tiles_x, tiles_y, PD);
        else
            hipLaunchKernelGGL((deform_col2im_tiled_kernel<AT, 16>), dim3(N * tiles_y * tiles_x, dbn_ceil_div(C, 16)), dim3(256),
                               (size_t)PD * PD * 16 * 8, st, dcols, x, offset, dx64, doff64, maxbits, d, tiles_x, tiles_y, PD);
        hipLaunchKernelGGL(deform_col2im_finish_kernel<AT>, dim3(dbn_grid(ndx / 4)), dim3(256), 0, st, dx64, doff64, maxbits, dx, doffset,
                           ndx / 4, M, 2 * R * S, off_stride, accumulate);
    });
    return dbn_status();
}
int dbn_deform_col2im(const float* dcols, const float* x, const float* offset, float* dx, float* doffset, int accumulate, void* ws, int N,
                      int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream) {
    return dbn_deform_col2im_t(0, dcols, x, offset, dx, doffset, accumulate, ws, N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride, stream);
}

// The same adjoint as a gather in plain fp32 (round 5; see deform_dx_gather_kernel): dx[N,H,W,C] = [dx +] sum over the samples that touch
// each pixel, doffset[N*Ho*Wo][off_stride] (channels >= 2RS zero).  Deterministic (one fixed summation order, independent of the grid), no
// atomics, equal to dbn_deform_col2im_t up to fp32 rounding of the sums.  ws: dbn_deform_col2im_gather_ws_bytes(N, Ho, Wo) bytes (the device-side
// maximum of |offset| that bounds the search window + one box per output pixel).  C <= 512.
long dbn_deform_col2im_gather_ws_bytes(int N, int Ho, int Wo) { return 16 + 16L * N * Ho * Wo; }
int dbn_deform_col2im_gather_t(int at, const void* dcols, const void* x, const void* offset, void* dx, void* doffset, int accumulate,
                               void* ws, int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride,
                               void* stream) {
    const DeformDims d{N, H, W, C, Ho, Wo, R, S, stride, pad, off_stride};
    DBN_REQUIRE(dcols && x && offset && dx && doffset && ws && dims_ok(d) && off_stride % 4 == 0 && C <= 512 && R * S <= 32 && R * S >= 9);
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)N * Ho * Wo;
    if (hipMemsetAsync(ws, 0, 16, st) != hipSuccess) return dbn_status();
    unsigned* maxbits = reinterpret_cast<unsigned*>(ws);
    int4* bbox = reinterpret_cast<int4*>(reinterpret_cast<char*>(ws) + 16);
    const int tiles_x = dbn_ceil_div(W, GATHER_TILE_X), tiles_y = dbn_ceil_div(H, GATHER_TILE);
    DBN_DISPATCH_AT(at, {
        hipLaunchKernelGGL(deform_bbox_kernel<AT>, dim3(dbn_grid(M, 256, 2048)), dim3(256), 0, st, offset, bbox, d, (int)M, maxbits);
        hipLaunchKernelGGL(deform_doffset_kernel<AT>, dim3(dbn_grid(M * R * S * 32, 256, 1 << 16)), dim3(256), 0, st, dcols, x, offset, doffset,
                           d, M * R * S);
        if (C <= 128)
            hipLaunchKernelGGL((deform_dx_gather_kernel<AT, 1>), dim3(N * tiles_y * tiles_x), dim3(256), 0, st, dcols, offset, bbox, dx, maxbits, d,
                               tiles_x, tiles_y, accumulate);
        else if (C <= 256)
            hipLaunchKernelGGL((deform_dx_gather_kernel<AT, 2>), dim3(N * tiles_y * tiles_x), dim3(256), 0, st, dcols, offset, bbox, dx, maxbits, d,
                               tiles_x, tiles_y, accumulate);
        else
            hipLaunchKernelGGL((deform_dx_gather_kernel<AT, 4>), dim3(N * tiles_y * tiles_x), dim3(256), 0, st, dcols, offset, bbox, dx, maxbits, d,
                               tiles_x, tiles_y, accumulate);
    });
    return dbn_status();
}

// out_bits[0] = bit pattern of max |offset[i]| over n elements (n % 4 == 0) of the activation type `at`, 0x7FC00000 as soon as one element is
// not finite: the number the caller reads back (one step late, no synchronisation in the step) to choose between the two adjoints — the
// gather's search window grows with it.
int dbn_deform_offset_absmax_t(int at, const void* offset, long n, unsigned* out_bits, void* stream) {
    DBN_REQUIRE(offset && out_bits && n > 0 && n % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out_bits, 0, 4, st) != hipSuccess) return dbn_status();
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(absmax_kernel<AT>, dim3(dbn_grid(n / 4, 256, 2048)), dim3(256), 0, st, offset, n / 4, out_bits));
    return dbn_status();
}

// to_ohwi = 1: dst[O][T][C] = scale * src[O][C][T] (OIHW -> GEMM column order); 0: the inverse
int dbn_permute_weight(const float* src, float* dst, int O, int C, int T, int to_ohwi, float scale, void* stream) {
    DBN_REQUIRE(src && dst && O > 0 && C > 0 && T > 0);
    hipLaunchKernelGGL(permute_weight_kernel, dim3(dbn_grid((long)O * C * T)), dim3(256), 0, (hipStream_t)stream, src, dst, O, C, T,
                       to_ohwi, scale);
    return dbn_status();
}

}  // extern "C"
