// HBM-bound NHWC kernels around the convolutions: train-mode BatchNorm (statistics,
// affine+ReLU(+residual) apply, backward), stem max-pool fused with BN+ReLU, nearest
// upsample/concat of the FPN, layout packing, column sums (bias gradients) and the
// flat fused Adam step.
//
// Replaces, on the reference's hot path: nn.BatchNorm2d / nn.ReLU (resnet.py:73-91,
// basic.py:32-36), nn.MaxPool2d (resnet.py:185,235), F.interpolate(nearest) + add/cat
// (segmentation_body.py:79-87), torch.optim.Adam.step (train.py:114-117,172).
#include "common.h"

namespace {

constexpr int MAX_PART = 1024;  // max partial blocks for per-channel reductions
// The streaming kernels of this file run beside the matrix kernels of the step's other stream.  A small fixed footprint —
// 3 workgroups of 256 threads per CU, 12 of its 32 wave slots — leaves the MFMA kernels their residency, and UNROLL
// independent 16-byte loads per lane keep HBM busy without relying on occupancy (8 TB/s x ~1 us needs ~32 KB in flight per CU).
constexpr int STREAM_BLOCKS = 768;
constexpr int UNROLL = 4;

// ----------------------------------------------------------------------------------
// per-channel reductions over [M][C] (C % 4 == 0): each block reduces a contiguous row range of one chunk of
// <= 1024 channels (blockIdx.y) into part[NV*C][block]; a second kernel folds the partials in
// double precision.  Thread t owns channel quad t % (Cc/4) of the chunk and walks rows t / (Cc/4).
// ----------------------------------------------------------------------------------
constexpr int CHUNK_C = 1024;

template <int NV, class F>
__device__ __forceinline__ void channel_reduce(int M, int Cfull, float* __restrict__ part, F&& body) {
    const int cb = blockIdx.y * CHUNK_C, C = min(CHUNK_C, Cfull - cb);
    const int c4n = C >> 2, cq = cb >> 2;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const int nrl = blockDim.x / c4n;
    const int rows_per = (M + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rl < nrl) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += nrl) body(r, cq + c4, acc);  // no stores in the loop: the unrolled loads issue together
    }
    extern __shared__ float red[];  // [nrl][NV][C]
    if (rl < nrl) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(red + ((long)rl * NV + v) * C + 4 * c4) = acc[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV * C; i += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nrl; ++k) s += red[(long)k * NV * C + i];
        const int v = i / C, c = i - v * C;
        part[((long)v * Cfull + cb + c) * gridDim.x + blockIdx.x] = s;  // transposed: [NV*Cfull][blocks]
    }
}

template <int AT>
__global__ void bn_stats_kernel(const void* __restrict__ y, int M, int C, float* __restrict__ part) {
    // shifted sums around the first row (pivot) to avoid E[x^2]-E[x]^2 cancellation
    const int c4n = C >> 2;
    channel_reduce<2>(M, C, part, [&](int r, int c4, f32x4* acc) {
        const f32x4 pv = dbn_ld4<AT>(y, c4);
        const f32x4 v = dbn_ld4<AT>(y, (long)r * c4n + c4) - pv;
        acc[0] += v;
        acc[1] += v * v;
    });
}

// out: scale = gamma*rstd, shift = beta - mean*scale, saved mean / rstd; running stats
// updated in place like F.batch_norm(training=True, momentum) does (unbiased var).
template <int AT>
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nb, const void* __restrict__ y, int M, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // 8 channels per 256-thread block; 32 lanes fold the partials of one channel
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (c >= C) return;
    const double s1 = dbn_team32_fold(part, nb, c, l32);
    const double s2 = dbn_team32_fold(part, nb, (long)C + c, l32);
    if (l32 != 0) return;
    const double pv = (double)dbn_ld1<AT>(y, c);
    const double dm = s1 / M;
    const double mean = pv + dm;
    double var = s2 / M - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = fmaf(-meanf, sc, beta[c]);
    mean_out[c] = meanf;
    rstd_out[c] = rstd;
    if (run_mean) {
        const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * meanf;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_coef_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ run_mean, const float* __restrict__ run_var, float eps,
                                    float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(run_var[c] + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = fmaf(-run_mean[c], sc, beta[c]);
}

// out = act(y*sc+sh [+ res*rsc+rsh | + res]).  An item is QW channel quads moved by one 16-byte access (QW = 1: four fp32 channels; QW = 2,
// 16-bit storage with C % 8 == 0: eight channels — round 5: the 8-byte accesses of the four-channel form ran the bf16 / fp16 passes at
// 0.55-0.7 of the 16-byte rate); `total` counts items, C / (4 QW) items per pixel.
template <int AT, int QW>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ y, const float* __restrict__ sc,
                                                       const float* __restrict__ sh, const void* __restrict__ res,
                                                       const float* __restrict__ rsc, const float* __restrict__ rsh,
                                                       void* __restrict__ out, long total, int C, int relu) {
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    // the grid stride is a multiple of the items per pixel (host side), so a thread keeps its channels: per-channel coefficients are
    // loaded once, and no 64-bit modulo sits in the streaming loop
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int c = (int)(i0 % cin) * 4 * QW;
    f32x4 s[QW], h[QW], s2[QW], h2[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        s[q] = *reinterpret_cast<const f32x4*>(sc + c + 4 * q);
        h[q] = *reinterpret_cast<const f32x4*>(sh + c + 4 * q);
        s2[q] = s[q];
        h2[q] = h[q];
        if (res && rsc) {
            s2[q] = *reinterpret_cast<const f32x4*>(rsc + c + 4 * q);
            h2[q] = *reinterpret_cast<const f32x4*>(rsh + c + 4 * q);
        }
    }
    auto ld = [&](const void* ptr, long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(ptr, i);
        else dbn_ldq<AT>(ptr, i, v);
    };
    auto st = [&](long i, const f32x4 (&v)[QW]) {
        if constexpr (QW == 1) dbn_st4<AT>(out, i, v[0]);
        else dbn_stq<AT>(out, i, v);
    };
    auto one = [&](long i, const f32x4 (&v)[QW], const f32x4 (&r)[QW]) {
        f32x4 o[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[q][e] = dbn_affine(v[q][e], s[q][e], h[q][e]);
            if (res) {
                if (rsc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[q][e] += dbn_affine(r[q][e], s2[q][e], h2[q][e]);
                } else {
                    o[q] += r[q];
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[q][e] = fmaxf(o[q][e], 0.f);
            }
        }
        st(i, o);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    long i = i0;
    for (; i + (UNROLL - 1) * stride < total; i += UNROLL * stride) {  // UNROLL (x2 with a residual) loads in flight per lane
        f32x4 v[UNROLL][QW], r[UNROLL][QW];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) ld(y, i + u * stride, v[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (res) ld(res, i + u * stride, r[u]);
            else {
#pragma unroll
                for (int q = 0; q < QW; ++q) r[u][q] = zero;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) one(i + u * stride, v[u], r[u]);
    }
    for (; i < total; i += stride) {
        f32x4 v[QW], r[QW];
        ld(y, i, v);
        if (res) ld(res, i, r);
        else {
#pragma unroll
            for (int q = 0; q < QW; ++q) r[q] = zero;
        }
        one(i, v, r);
    }
}

// backward reductions: g = dout * (zmask > 0);  sums of g and g*xhat
// ReLU mask of g: from a saved activation (zmask), or recomputed from y with the forward's own
// scale/shift (msc/msh; bit-identical to the forward because both use dbn_affine), or none.
template <int AT>
__global__ void bn_bwd_reduce_kernel(const void* __restrict__ y, const void* __restrict__ zmask, const float* __restrict__ msc,
                                     const float* __restrict__ msh, const void* __restrict__ dout,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, int M, int C,
                                     float* __restrict__ part) {
    const int c4n = C >> 2;
    channel_reduce<2>(M, C, part, [&](int r, int c4, f32x4* acc) {
        const long off = (long)r * c4n + c4;
        f32x4 g = dbn_ld4<AT>(dout, off);
        const f32x4 v = dbn_ld4<AT>(y, off);
        if (zmask) {
            const f32x4 z = dbn_ld4<AT>(zmask, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
        } else if (msc) {
            const f32x4 s_ = *reinterpret_cast<const f32x4*>(msc + 4 * c4);
            const f32x4 h_ = *reinterpret_cast<const f32x4*>(msh + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = dbn_affine(v[e], s_[e], h_[e]) > 0.f ? g[e] : 0.f;
        }
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + 4 * c4);
        acc[0] += g;
        acc[1] += g * ((v - mu) * rs);
    });
}

// dgamma, dbeta, and the two per-channel means used by the apply pass
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nb, int M, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2, float gscale) {
    // 8 channels per 256-thread block; 32 lanes fold the partials of one channel
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (c >= C) return;
    const double s1 = dbn_team32_fold(part, nb, c, l32);
    const double s2 = dbn_team32_fold(part, nb, (long)C + c, l32);
    if (l32 != 0) return;
    dbeta[c] = (float)(s1 * gscale);
    dgamma[c] = (float)(s2 * gscale);
    c1[c] = (float)(s1 / M);
    c2[c] = (float)(s2 / M);
}

// The same for MANY partials per channel (the sums produced per output tile by a data gradient's epilogue: 6400 rows at bs16
// 160x160 with 64-row tiles): one 256-thread block per channel instead of a 32-lane team — the team's chain of 50 dependent
// load rounds took 20-30 us per BatchNorm layer, the block takes a few.  Fixed order: thread t adds rows t, t+256, ... in fp64,
// then a fixed tree over the threads.
__global__ __launch_bounds__(256) void bn_bwd_finalize_wide_kernel(const float* __restrict__ part, int nb, int M, int C,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ c1, float* __restrict__ c2, float gscale) {
    __shared__ double red[2][256];
    const int c = blockIdx.x, t = threadIdx.x;
    const float* p1 = part + (long)c * nb;
    const float* p2 = part + ((long)C + c) * nb;
    double s1 = 0.0, s2 = 0.0;
    int b = t;
    for (; b + 768 < nb; b += 1024) {  // eight independent loads in flight per thread
        const float a0 = p1[b], a1 = p1[b + 256], a2 = p1[b + 512], a3 = p1[b + 768];
        const float q0 = p2[b], q1 = p2[b + 256], q2 = p2[b + 512], q3 = p2[b + 768];
        s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        s2 += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
    }
    for (; b < nb; b += 256) {
        s1 += (double)p1[b];
        s2 += (double)p2[b];
    }
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            red[0][t] += red[0][t + o];
            red[1][t] += red[1][t + o];
        }
        __syncthreads();
    }
    if (t != 0) return;
    s1 = red[0][0];
    s2 = red[1][0];
    dbeta[c] = (float)(s1 * gscale);
    dgamma[c] = (float)(s2 * gscale);
    c1[c] = (float)(s1 / M);
    c2[c] = (float)(s2 / M);
}

// dy = gamma*rstd*(g - c1 - xhat*c2); optionally also emits g (the ReLU-masked dout).
// bias_part (optional, needs 256 % (C/4) == 0): per-block column sums of dy, [C][gridDim.x] — the gradient of the bias of the
// conv that feeds this BatchNorm (analytically zero; the reference's value is the round-off of exactly this sum), so that no
// separate pass re-reads dy for it.
template <int AT, int QW>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ y, const void* __restrict__ zmask,
                                                           const float* __restrict__ msc, const float* __restrict__ msh,
                                                           const void* __restrict__ dout, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           void* __restrict__ dy, void* __restrict__ gout, int gout_acc,
                                                           long total, int C, float* __restrict__ bias_part) {
    // an item = QW channel quads moved by one 16-byte access (QW = 2: 16-bit storage, C % 8 == 0; see bn_apply_kernel)
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int c = (int)(i0 % cin) * 4 * QW;  // constant per thread: the grid stride is a multiple of the items per pixel (host side)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s_[QW], h_[QW], mu[QW], rs[QW], k1[QW], k2[QW], gr[QW], bsum[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        s_[q] = h_[q] = zero;
        if (msc) {
            s_[q] = *reinterpret_cast<const f32x4*>(msc + c + 4 * q);
            h_[q] = *reinterpret_cast<const f32x4*>(msh + c + 4 * q);
        }
        mu[q] = *reinterpret_cast<const f32x4*>(mean + c + 4 * q);
        rs[q] = *reinterpret_cast<const f32x4*>(rstd + c + 4 * q);
        k1[q] = *reinterpret_cast<const f32x4*>(c1 + c + 4 * q);
        k2[q] = *reinterpret_cast<const f32x4*>(c2 + c + 4 * q);
        gr[q] = *reinterpret_cast<const f32x4*>(gamma + c + 4 * q) * rs[q];
        bsum[q] = zero;
    }
    const bool acc = gout && gout_acc;
    auto ld = [&](const void* ptr, long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(ptr, i);
        else dbn_ldq<AT>(ptr, i, v);
    };
    auto st = [&](void* ptr, long i, const f32x4 (&v)[QW]) {
        if constexpr (QW == 1) dbn_st4<AT>(ptr, i, v[0]);
        else dbn_stq<AT>(ptr, i, v);
    };
    auto one = [&](long i, f32x4 (&g)[QW], const f32x4 (&v)[QW], const f32x4 (&z)[QW], const f32x4 (&old)[QW]) {
        f32x4 d[QW], go[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            if (zmask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[q][e] = z[q][e] > 0.f ? g[q][e] : 0.f;
            } else if (msc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[q][e] = dbn_affine(v[q][e], s_[q][e], h_[q][e]) > 0.f ? g[q][e] : 0.f;
            }
            const f32x4 xh = (v[q] - mu[q]) * rs[q];
            d[q] = gr[q] * (g[q] - k1[q] - xh * k2[q]);
            bsum[q] += d[q];
            go[q] = acc ? g[q] + old[q] : g[q];
        }
        st(dy, i, d);
        if (gout) st(gout, i, go);
    };
    auto zeros = [&](f32x4 (&v)[QW]) {
#pragma unroll
        for (int q = 0; q < QW; ++q) v[q] = zero;
    };
    long i = i0;
    for (; i + (UNROLL - 1) * stride < total; i += UNROLL * stride) {  // 2..4 x UNROLL independent loads in flight per lane
        f32x4 g[UNROLL][QW], v[UNROLL][QW], z[UNROLL][QW], o[UNROLL][QW];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            ld(dout, i + u * stride, g[u]);
            ld(y, i + u * stride, v[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (zmask) ld(zmask, i + u * stride, z[u]);
            else zeros(z[u]);
            if (acc) ld(gout, i + u * stride, o[u]);
            else zeros(o[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) one(i + u * stride, g[u], v[u], z[u], o[u]);
    }
    for (; i < total; i += stride) {
        f32x4 g[QW], v[QW], z[QW], o[QW];
        ld(dout, i, g);
        ld(y, i, v);
        if (zmask) ld(zmask, i, z);
        else zeros(z);
        if (acc) ld(gout, i, o);
        else zeros(o);
        one(i, g, v, z, o);
    }
    if (bias_part) {  // threads t, t + cin, t + 2 cin, ... of the block hold the same channels
        __shared__ f32x4 red[QW][256];
#pragma unroll
        for (int q = 0; q < QW; ++q) red[q][threadIdx.x] = bsum[q];
        __syncthreads();
        if ((int)threadIdx.x < cin) {
            const int cq = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) % cin) * 4 * QW;
#pragma unroll
            for (int q = 0; q < QW; ++q) {
                f32x4 t = zero;
                for (int k = threadIdx.x; k < 256; k += cin) t += red[q][k];
#pragma unroll
                for (int e = 0; e < 4; ++e) bias_part[(long)(cq + 4 * q + e) * gridDim.x + blockIdx.x] = t[e];
            }
        }
    }
}

// generic column sum [M][C] -> part (NV=1)
template <int AT>
__global__ void col_sum_kernel(const void* __restrict__ x, int M, int C, float* __restrict__ part) {
    const int c4n = C >> 2;
    channel_reduce<1>(M, C, part, [&](int r, int c4, f32x4* acc) { acc[0] += dbn_ld4<AT>(x, (long)r * c4n + c4); });
}

__global__ void fold_partials_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ out, float scale) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (i >= n) return;
    const double s = dbn_team32_fold(part, nb, i, l32);
    if (l32 == 0) out[i] = (float)(s * scale);
}

// ----------------------------------------------------------------------------------
// stem: relu(bn(y)) -> maxpool 3x3 s2 p1, forward and backward
// ----------------------------------------------------------------------------------
// One thread: the channel quad of TWO vertically adjacent outputs (rows 2k, 2k+1): their windows share input row 4k+1, so the 5
// input rows are read once (15 loads for 2 outputs instead of 18).  Horizontal neighbours re-read through L1/L2 as before; it was
// the vertical overlap that went back to HBM (FETCH_SIZE 1.54x the tensor with one output per thread, 1.25x the ideal now).
// QW = 2 (16-bit storage, C % 8 == 0; round 5): eight channels per lane and access — the 8-byte accesses of the four-channel form reach
// 0.55-0.7 of the 16-byte rate (cfg5: 621 us for 2.1 GB)
template <int AT, int QW>
__global__ void bnrelu_maxpool_fwd_kernel(const void* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                          void* __restrict__ out, int N, int H, int W, int C, int Ho, int Wo) {
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const int Hp = (Ho + 1) >> 1;
    const long total = (long)N * Hp * Wo * cin;
    auto ld = [&](long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(y, i);
        else dbn_ldq<AT>(y, i, v);
    };
    auto st = [&](long i, const f32x4 (&v)[QW]) {
        if constexpr (QW == 1) dbn_st4<AT>(out, i, v[0]);
        else dbn_stq<AT>(out, i, v);
    };
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        long t = i / cin;
        const int ow = (int)(t % Wo);
        t /= Wo;
        const int k = (int)(t % Hp);
        const int n = (int)(t / Hp);
        f32x4 s[QW], h[QW], m0[QW], m1[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            s[q] = *reinterpret_cast<const f32x4*>(sc + 4 * (QW * ci + q));
            h[q] = *reinterpret_cast<const f32x4*>(sh + 4 * (QW * ci + q));
            m0[q] = m1[q] = f32x4{0.f, 0.f, 0.f, 0.f};  // relu output >= 0, so 0 is a neutral start (padding never wins)
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int ih = 4 * k - 1 + r;
            if ((unsigned)ih >= (unsigned)H) continue;
            f32x4 rm[QW];
#pragma unroll
            for (int q = 0; q < QW; ++q) rm[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) {
                const int iw = ow * 2 - 1 + qq;
                if ((unsigned)iw >= (unsigned)W) continue;
                f32x4 v[QW];
                ld((((long)n * H + ih) * W + iw) * cin + ci, v);
#pragma unroll
                for (int q = 0; q < QW; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rm[q][e] = fmaxf(rm[q][e], dbn_affine_relu(v[q][e], s[q][e], h[q][e]));
            }
#pragma unroll
            for (int q = 0; q < QW; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (r <= 2) m0[q][e] = fmaxf(m0[q][e], rm[q][e]);
                    if (r >= 2) m1[q][e] = fmaxf(m1[q][e], rm[q][e]);
                }
        }
        const long o = (((long)n * Ho + 2 * k) * Wo + ow) * cin + ci;
        st(o, m0);
        if (2 * k + 1 < Ho) st(o + (long)Wo * cin, m1);
    }
}

// dz[n,ih,iw,c] = [z>0] * sum over windows containing (ih,iw) whose max equals z of dpool
// 16-bit storage: the pooled value is the ROUNDED maximum, so equality is tested on the rounded activation
template <int AT>
__device__ __forceinline__ float round_to_storage(float v) {
    if constexpr (AT == 1) return (float)(__bf16)v;
    else if constexpr (AT == 2) return (float)(_Float16)v;
    else return v;
}
// bn_part (optional, needs 256 % (C/4) == 0 so that a thread keeps its channel quad): per-block partial sums of the two
// reductions of the BatchNorm backward that consumes dz — sum(dz) and sum(dz * xhat) per channel, [2*C][gridDim.x] — so that
// the stem's BatchNorm backward does not re-read y and dz (2 x 420 MB at bs16) to form them.
template <int AT, int QW>
__global__ void bnrelu_maxpool_bwd_kernel(const void* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                          const void* __restrict__ pooled, const void* __restrict__ dpool,
                                          void* __restrict__ dz, int N, int H, int W, int C, int Ho, int Wo,
                                          const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ bn_part) {
    // One thread: QW channel quads (QW = 2: 16-bit storage, eight channels = one 16-byte access; round 5) of a 2x2 block of input pixels
    // (rows 2a, 2a+1, columns 2b, 2b+1).  The four pixels lie in the windows (a..a+1) x (b..b+1) only, so 4 loads of the pooled maxima
    // and 4 of their gradients serve all four (a thread per pixel read 9 + 9: FETCH_SIZE 1.43x the tensors).
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long total = (long)N * Hb * Wb * cin;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s1[QW], s2[QW], mu[QW], rs[QW];
    const int cq0 = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) % cin) * 4 * QW;  // constant per thread (256 % cin == 0, host side)
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        s1[q] = s2[q] = mu[q] = rs[q] = zero;
        if (bn_part) {
            mu[q] = *reinterpret_cast<const f32x4*>(mean + cq0 + 4 * q);
            rs[q] = *reinterpret_cast<const f32x4*>(rstd + cq0 + 4 * q);
        }
    }
    auto ld = [&](const void* ptr, long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(ptr, i);
        else dbn_ldq<AT>(ptr, i, v);
    };
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        long t = i / cin;
        const int b = (int)(t % Wb);
        t /= Wb;
        const int a = (int)(t % Hb);
        const int n = (int)(t / Hb);
        f32x4 s[QW], h[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            s[q] = *reinterpret_cast<const f32x4*>(sc + 4 * (QW * ci + q));
            h[q] = *reinterpret_cast<const f32x4*>(sh + 4 * (QW * ci + q));
        }
        // windows (a + u, b + w), u, w in {0, 1}
        f32x4 pm[2][2][QW], dp[2][2][QW];
        bool wok[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                wok[u][w] = a + u < Ho && b + w < Wo;
                const long o = (((long)n * Ho + a + u) * Wo + b + w) * cin + ci;
#pragma unroll
                for (int q = 0; q < QW; ++q) pm[u][w][q] = dp[u][w][q] = zero;
                if (wok[u][w]) {
                    ld(pooled, o, pm[u][w]);
                    ld(dpool, o, dp[u][w]);
                }
            }
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                const int ih = 2 * a + dy_, iw = 2 * b + dx_;
                if (ih >= H || iw >= W) continue;
                const long pi = (((long)n * H + ih) * W + iw) * cin + ci;
                f32x4 v[QW], g[QW];
                ld(y, pi, v);
#pragma unroll
                for (int q = 0; q < QW; ++q) {
                    f32x4 z;
                    g[q] = zero;
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = round_to_storage<AT>(dbn_affine_relu(v[q][e], s[q][e], h[q][e]));
                    // pixel row 2a lies in window row a only, row 2a+1 in rows a and a+1 (same for columns)
#pragma unroll
                    for (int u = 0; u <= dy_; ++u)
#pragma unroll
                        for (int w = 0; w <= dx_; ++w) {
                            if (!wok[u][w]) continue;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (z[e] > 0.f && z[e] == pm[u][w][q][e]) g[q][e] += dp[u][w][q][e];
                        }
                }
                if constexpr (QW == 1) dbn_st4<AT>(dz, pi, g[0]);
                else dbn_stq<AT>(dz, pi, g);
                if (bn_part) {
#pragma unroll
                    for (int q = 0; q < QW; ++q) {
                        f32x4 gr = g[q];  // the BatchNorm backward reads the STORED gradient
#pragma unroll
                        for (int e = 0; e < 4; ++e) gr[e] = round_to_storage<AT>(g[q][e]);
                        s1[q] += gr;
                        s2[q] += gr * ((v[q] - mu[q]) * rs[q]);
                    }
                }
            }
    }
    if (bn_part) {  // threads t, t + cin, ... of the block hold the same channels
        __shared__ f32x4 red[2][QW][256];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            red[0][q][threadIdx.x] = s1[q];
            red[1][q][threadIdx.x] = s2[q];
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * cin * QW) {
            const int which = threadIdx.x / (cin * QW), r_ = threadIdx.x - which * cin * QW, t0 = r_ / QW, q = r_ - t0 * QW;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int k = t0; k < 256; k += cin) t += red[which][q][k];
            const int cq = (int)((blockIdx.x * (long)blockDim.x + t0) % cin) * 4 * QW + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) bn_part[((long)which * C + cq + e) * gridDim.x + blockIdx.x] = t[e];
        }
    }
}

// ---- the same pair with a RECORDED argmax (late in round 5) ------------------------------------------------------------------------
// Forward: besides the pooled maximum, the position of the window's FIRST maximum in scan order (rows, then columns: the rule of
// /root/reference's nn.MaxPool2d, whose backward routes a window's gradient to that one position) as a code 3 r + q per element — 15 where the
// pooled value is 0, i.e. where ReLU passes no gradient — and the PRE-BatchNorm value y at that position.  With them the stem's backward needs
// no gradient tensor at the conv's resolution: the two channel sums of the BatchNorm backward come from the POOLED tensors (a window's gradient
// lands on exactly one position, so sum(g) = sum(dpool [code != 15]) and sum(g xhat) = sum(dpool xhat(ypool))), and one pass over y then
// writes dy = gamma rstd (g - c1 - xhat c2) directly.  (The kernels above give the gradient to EVERY position that ties with the maximum and
// therefore cannot take their sums from the pooled side.)
template <int AT, int QW>
__global__ __launch_bounds__(256) void bnrelu_maxpool_fwd_arg_kernel(const void* __restrict__ y, const float* __restrict__ sc,
                                                                     const float* __restrict__ sh, void* __restrict__ out,
                                                                     unsigned* __restrict__ idx, void* __restrict__ ypool, int N, int H, int W,
                                                                     int C, int Ho, int Wo) {
    // One thread: the channel quad(s) of two vertically adjacent outputs (rows 2k, 2k+1), as bnrelu_maxpool_fwd_kernel: the 5 input rows are
    // read once; row 2 belongs to both windows.  Each loaded value updates the window maxima directly, in scan order (strictly greater: the
    // first maximum keeps the window).
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const int Hp = (Ho + 1) >> 1;
    const long total = (long)N * Hp * Wo * cin;
    auto ld = [&](long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(y, i);
        else dbn_ldq<AT>(y, i, v);
    };
    auto st = [&](void* ptr, long i, const f32x4 (&v)[QW]) {
        if constexpr (QW == 1) dbn_st4<AT>(ptr, i, v[0]);
        else dbn_stq<AT>(ptr, i, v);
    };
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        long t = i / cin;
        const int ow = (int)(t % Wo);
        t /= Wo;
        const int k = (int)(t % Hp);
        const int n = (int)(t / Hp);
        f32x4 s[QW], h[QW], m0[QW], m1[QW], y0[QW], y1[QW];
        unsigned c0[QW], c1[QW];  // four one-byte codes per quad
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            s[q] = *reinterpret_cast<const f32x4*>(sc + 4 * (QW * ci + q));
            h[q] = *reinterpret_cast<const f32x4*>(sh + 4 * (QW * ci + q));
            m0[q] = m1[q] = f32x4{-1.f, -1.f, -1.f, -1.f};  // below every ReLU output: the first valid position always takes the window
            y0[q] = y1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            c0[q] = c1[q] = 0u;
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int ih = 4 * k - 1 + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) {
                const int iw = ow * 2 - 1 + qq;
                if ((unsigned)iw >= (unsigned)W) continue;
                f32x4 v[QW];
                ld((((long)n * H + ih) * W + iw) * cin + ci, v);
#pragma unroll
                for (int q = 0; q < QW; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = dbn_affine_relu(v[q][e], s[q][e], h[q][e]);
                        if (r <= 2 && z > m0[q][e]) {
                            m0[q][e] = z;
                            y0[q][e] = v[q][e];
                            c0[q] = (c0[q] & ~(0xFFu << (8 * e))) | ((unsigned)(3 * r + qq) << (8 * e));
                        }
                        if (r >= 2 && z > m1[q][e]) {
                            m1[q][e] = z;
                            y1[q][e] = v[q][e];
                            c1[q] = (c1[q] & ~(0xFFu << (8 * e))) | ((unsigned)(3 * (r - 2) + qq) << (8 * e));
                        }
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < QW; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!(m0[q][e] > 0.f)) c0[q] = (c0[q] & ~(0xFFu << (8 * e))) | (15u << (8 * e));
                if (!(m1[q][e] > 0.f)) c1[q] = (c1[q] & ~(0xFFu << (8 * e))) | (15u << (8 * e));
                m0[q][e] = fmaxf(m0[q][e], 0.f);
                m1[q][e] = fmaxf(m1[q][e], 0.f);
            }
        const long o = (((long)n * Ho + 2 * k) * Wo + ow) * cin + ci;
        st(out, o, m0);
        st(ypool, o, y0);
#pragma unroll
        for (int q = 0; q < QW; ++q) idx[o * QW + q] = c0[q];
        if (2 * k + 1 < Ho) {
            const long o1 = o + (long)Wo * cin;
            st(out, o1, m1);
            st(ypool, o1, y1);
#pragma unroll
            for (int q = 0; q < QW; ++q) idx[o1 * QW + q] = c1[q];
        }
    }
}

#ifndef DBN_POOL_STATS_UNROLL
#define DBN_POOL_STATS_UNROLL 1  // (fp32 729.1 -> 732.2 images/s over three interleaved pairs, bf16 unchanged; same summation order)
#endif
// part[2 * C][gridDim.x]: per-block sums of g and g * xhat over the POOLED elements (g = dpool where the code is not 15); threads t, t + C/4, ...
// of a block hold the same channel quad (256 % (C/4) == 0, host side)
template <int AT>
__global__ __launch_bounds__(256) void maxpool_bn_stats_kernel(const void* __restrict__ dpool, const unsigned* __restrict__ idx,
                                                               const void* __restrict__ ypool, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ part, long total, int C) {
    const int cin = C / 4;
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c = (int)(i0 % cin) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    const long stride = (long)gridDim.x * blockDim.x;
    auto add = [&](const f32x4& dp, const f32x4& yp, unsigned code) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = ((code >> (8 * e)) & 0xFFu) != 15u ? dp[e] : 0.f;
        s1 += g;
        s2 += g * ((yp - mu) * rs);
    };
    long i = i0;
#if DBN_POOL_STATS_UNROLL
    // four items (twelve loads) in flight per thread: beside the last weight-gradient kernel of the step this pass gets few wave slots
    for (; i + 3 * stride < total; i += 4 * stride) {
        f32x4 dp[4], yp[4];
        unsigned code[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dp[k] = dbn_ld4<AT>(dpool, i + k * stride);
            yp[k] = dbn_ld4<AT>(ypool, i + k * stride);
            code[k] = idx[i + k * stride];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) add(dp[k], yp[k], code[k]);
    }
#endif
    for (; i < total; i += stride) add(dbn_ld4<AT>(dpool, i), dbn_ld4<AT>(ypool, i), idx[i]);
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < 2 * cin) {
        const int which = threadIdx.x / cin, t0 = threadIdx.x - which * cin;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = t0; k < 256; k += cin) t += red[which][k];
        const int cq = (int)((blockIdx.x * (long)blockDim.x + t0) % cin) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) part[((long)which * C + cq + e) * gridDim.x + blockIdx.x] = t[e];
    }
}

// dy[n,ih,iw,c] = gamma rstd (g - c1 - xhat c2), g = the sum of dpool over the windows whose recorded first maximum is (ih, iw).
// One thread: a channel quad of a 2 x 2 block of input pixels (rows 2a, 2a+1, columns 2b, 2b+1), which lies in the windows
// (a..a+1) x (b..b+1) only; pixel (2a + dy, 2b + dx) is position (dy - 2u + 1, dx - 2w + 1) of window (a + u, b + w).
template <int AT, int QW>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_kernel(const void* __restrict__ y, const void* __restrict__ dpool,
                                                                   const unsigned* __restrict__ idx, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                                                   void* __restrict__ dy, int N, int H, int W, int C, int Ho, int Wo) {
    // QW = 2 (16-bit storage, C % 8 == 0): eight channels per 16-byte access
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long total = (long)N * Hb * Wb * cin;
    auto ld = [&](const void* ptr, long i, f32x4 (&v)[QW]) {
        if constexpr (QW == 1) v[0] = dbn_ld4<AT>(ptr, i);
        else dbn_ldq<AT>(ptr, i, v);
    };
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        long t = i / cin;
        const int b = (int)(t % Wb);
        t /= Wb;
        const int a = (int)(t % Hb);
        const int n = (int)(t / Hb);
        f32x4 mu[QW], rs[QW], k1[QW], k2[QW], gr[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
            const int c = 4 * (QW * ci + q);
            mu[q] = *reinterpret_cast<const f32x4*>(mean + c);
            rs[q] = *reinterpret_cast<const f32x4*>(rstd + c);
            k1[q] = *reinterpret_cast<const f32x4*>(c1 + c);
            k2[q] = *reinterpret_cast<const f32x4*>(c2 + c);
            gr[q] = *reinterpret_cast<const f32x4*>(gamma + c) * rs[q];
        }
        f32x4 dp[2][2][QW];
        unsigned cd[2][2][QW];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
#pragma unroll
                for (int q = 0; q < QW; ++q) {
                    dp[u][w][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    cd[u][w][q] = 0x0F0F0F0Fu;  // (no window there: no position matches)
                }
                if (a + u < Ho && b + w < Wo) {
                    const long o = (((long)n * Ho + a + u) * Wo + b + w) * cin + ci;
                    ld(dpool, o, dp[u][w]);
#pragma unroll
                    for (int q = 0; q < QW; ++q) cd[u][w][q] = idx[o * QW + q];
                }
            }
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                const int ih = 2 * a + dy_, iw = 2 * b + dx_;
                if (ih >= H || iw >= W) continue;
                const long pi = (((long)n * H + ih) * W + iw) * cin + ci;
                f32x4 v[QW], d[QW];
                ld(y, pi, v);
#pragma unroll
                for (int q = 0; q < QW; ++q) {
                    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u <= dy_; ++u)
#pragma unroll
                        for (int w = 0; w <= dx_; ++w) {
                            const unsigned code = 3u * (unsigned)(dy_ - 2 * u + 1) + (unsigned)(dx_ - 2 * w + 1);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (((cd[u][w][q] >> (8 * e)) & 0xFFu) == code) g[e] += dp[u][w][q][e];
                        }
                    const f32x4 xh = (v[q] - mu[q]) * rs[q];
                    d[q] = gr[q] * (g - k1[q] - xh * k2[q]);
                }
                if constexpr (QW == 1) dbn_st4<AT>(dy, pi, d[0]);
                else dbn_stq<AT>(dy, pi, d);
            }
    }
}

// ----------------------------------------------------------------------------------
// nearest upsample (F.interpolate(size=...) semantics: src = min(floor(dst*in/out), in-1))
// ----------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int d, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)d * scale);
    return s < in - 1 ? s : in - 1;
}

// dst[n,h,w,coff+c] = src[n,nh,nw,c] (+ addend[n,h,w,c])
template <int AT, int QW>
__global__ void nearest_up_fwd_kernel(const void* __restrict__ src, const void* __restrict__ addend, void* __restrict__ dst,
                                      int N, int Hs, int Ws, int C, int H, int W, int Cdst, int coff) {
    // QW = 2 (16-bit storage, C, Cdst, coff multiples of 8; round 5): eight channels per 16-byte access
    static_assert(QW == 1 || (QW == 2 && AT != 0), "two quads per access: 16-bit storage");
    const int cin = C / (4 * QW);
    const long total = (long)N * H * W * cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cin) * 4 * QW;
        long t = i / cin;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int sh_ = nearest_src(h, Hs, H), sw_ = nearest_src(w, Ws, W);
        const long is = ((((long)n * Hs + sh_) * Ws + sw_) * C + c) / (4 * QW), ia = ((((long)n * H + h) * W + w) * C + c) / (4 * QW);
        const long id = ((((long)n * H + h) * W + w) * Cdst + coff + c) / (4 * QW);
        if constexpr (QW == 1) {
            f32x4 v = dbn_ld4<AT>(src, is);
            if (addend) v += dbn_ld4<AT>(addend, ia);
            dbn_st4<AT>(dst, id, v);
        } else {
            f32x4 v[QW], a[QW];
            dbn_ldq<AT>(src, is, v);
            if (addend) {
                dbn_ldq<AT>(addend, ia, a);
#pragma unroll
                for (int q = 0; q < QW; ++q) v[q] += a[q];
            }
            dbn_stq<AT>(dst, id, v);
        }
    }
}

// dsrc[n,hs,ws,c] (+)= sum over (h,w) mapping to (hs,ws) of dbig[n,h,w,coff+c]
template <int AT>
__global__ void nearest_up_bwd_kernel(const void* __restrict__ dbig, void* __restrict__ dsrc, int N, int Hs, int Ws, int C,
                                      int H, int W, int Cbig, int coff, int accumulate) {
    const int c4n = C >> 2;
    const long total = (long)N * Hs * Ws * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        long t = i / c4n;
        const int ws = (int)(t % Ws);
        t /= Ws;
        const int hs = (int)(t % Hs);
        const int n = (int)(t / Hs);
        // candidate destination range (generous by one on both sides, then filtered exactly)
        int h0 = (int)((long)hs * H / Hs) - 1, h1 = (int)(((long)hs + 1) * H / Hs) + 1;
        int w0 = (int)((long)ws * W / Ws) - 1, w1 = (int)(((long)ws + 1) * W / Ws) + 1;
        h0 = max(h0, 0); h1 = min(h1, H - 1);
        w0 = max(w0, 0); w1 = min(w1, W - 1);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int h = h0; h <= h1; ++h) {
            if (nearest_src(h, Hs, H) != hs) continue;
            for (int w = w0; w <= w1; ++w) {
                if (nearest_src(w, Ws, W) != ws) continue;
                g += dbn_ld4<AT>(dbig, ((((long)n * H + h) * W + w) * Cbig + coff + c) >> 2);
            }
        }
        if (accumulate) g += dbn_ld4<AT>(dsrc, i);
        dbn_st4<AT>(dsrc, i, g);
    }
}

// ----------------------------------------------------------------------------------
// F.interpolate(mode='bilinear', align_corners=True) of NCHW planes (models.py:43-46): only a real
// resample when H or W is not a multiple of 32 (e.g. 428 -> 427 rows in the inference CLIs).
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_src(int d, float scale, int in, int& i0, int& i1, float& w1) {
    const float x = scale * (float)d;  // align_corners=True: scale = (in-1)/(out-1)
    i0 = (int)x;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = x - (float)i0;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, long planes, int Hs, int Ws, int H, int W,
                                    float sh, float sw) {
    const long total = planes * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long t = i / W;
        const int h = (int)(t % H);
        const long pl = t / H;
        int h0, h1, w0, w1;
        float lh, lw;
        bilinear_src(h, sh, Hs, h0, h1, lh);
        bilinear_src(w, sw, Ws, w0, w1, lw);
        const float* s = src + pl * (long)Hs * Ws;
        dst[i] = (1.f - lh) * ((1.f - lw) * s[(long)h0 * Ws + w0] + lw * s[(long)h0 * Ws + w1]) +
                 lh * ((1.f - lw) * s[(long)h1 * Ws + w0] + lw * s[(long)h1 * Ws + w1]);
    }
}

// adjoint, gather form (deterministic): dsrc[hs,ws] = sum over dst pixels whose footprint contains (hs,ws)
__global__ void bilinear_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, long planes, int Hs, int Ws, int H, int W,
                                    float sh, float sw) {
    const long total = planes * Hs * Ws;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ws = (int)(i % Ws);
        const long t = i / Ws;
        const int hs = (int)(t % Hs);
        const long pl = t / Hs;
        // candidate destination rows/cols: x = scale*d in (hs-1, hs+1)
        const float ih = sh > 0.f ? 1.f / sh : 0.f, iw = sw > 0.f ? 1.f / sw : 0.f;
        int hlo = sh > 0.f ? (int)floorf((float)(hs - 1) * ih) - 1 : 0, hhi = sh > 0.f ? (int)ceilf((float)(hs + 1) * ih) + 1 : H - 1;
        int wlo = sw > 0.f ? (int)floorf((float)(ws - 1) * iw) - 1 : 0, whi = sw > 0.f ? (int)ceilf((float)(ws + 1) * iw) + 1 : W - 1;
        hlo = max(hlo, 0); hhi = min(hhi, H - 1);
        wlo = max(wlo, 0); whi = min(whi, W - 1);
        const float* d = ddst + pl * (long)H * W;
        float acc = 0.f;
        for (int h = hlo; h <= hhi; ++h) {
            int h0, h1;
            float lh;
            bilinear_src(h, sh, Hs, h0, h1, lh);
            const float wh = (h0 == hs ? 1.f - lh : 0.f) + (h1 == hs ? lh : 0.f);
            if (wh == 0.f) continue;
            for (int w = wlo; w <= whi; ++w) {
                int w0, w1;
                float lw;
                bilinear_src(w, sw, Ws, w0, w1, lw);
                const float ww = (w0 == ws ? 1.f - lw : 0.f) + (w1 == ws ? lw : 0.f);
                if (ww != 0.f) acc += wh * ww * d[(long)h * W + w];
            }
        }
        dsrc[i] = acc;
    }
}

// ----------------------------------------------------------------------------------
// FPN output conv over [p2 | up2(p3) | up4(p4) | up8(p5)] (segmentation_body.py:55-61,82-87): a 3x3 conv of a
// nearest-upsampled tensor re-reads every low-resolution pixel f*f times with only (f+2)^2 distinct tap sums.
// For group g (upsample factor f = 1,2,4,8) the conv is the adjoint of a (f+2)x(f+2), stride-f, pad-1 conv
// with the COMBINED weights  Wd_g[ci][co][u][v] = sum_{r in R(u)} sum_{s in R(v)} W[co][64g+ci][r][s],
// R(u) = { r in 0..2 : 2-u <= r <= f+1-u }.  Data and weight gradients of the big 256->256 conv then cost
// 9, 4, 2.25 and 1.56 taps per output pixel instead of 9 each (47 % of the MACs) and need no concat tensor.
// ----------------------------------------------------------------------------------
__global__ void fpn_combine_weights_kernel(const float* __restrict__ w, int Co, int Cin, int g, int Cg, int f,
                                           float* __restrict__ wd) {
    const int k = f + 2;
    const long total = (long)Cg * Co * k * k;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % k);
        long t = i / k;
        const int u = (int)(t % k);
        t /= k;
        const int co = (int)(t % Co);
        const int ci = (int)(t / Co);
        const float* src = w + (((long)co * Cin + g * Cg + ci) * 3) * 3;
        float acc = 0.f;
        for (int r = max(0, 2 - u); r <= min(2, f + 1 - u); ++r)
            for (int s_ = max(0, 2 - v); s_ <= min(2, f + 1 - v); ++s_) acc += src[r * 3 + s_];
        wd[i] = acc;  // [ci][co][u][v]: OIHW of the strided conv dY -> dP_g
    }
}

// adjoint of the combination: dW[co][64g+ci][r][s] = sum_{u: r in R(u)} sum_{v: s in R(v)} T_g[ci][co][u][v]
__global__ void fpn_scatter_wgrad_kernel(const float* __restrict__ t0, const float* __restrict__ t1, const float* __restrict__ t2,
                                         const float* __restrict__ t3, int Co, int Cin, int Cg, float* __restrict__ dw) {
    const long total = (long)Co * Cin * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s_ = (int)(i % 3);
        long t = i / 3;
        const int r = (int)(t % 3);
        t /= 3;
        const int cin = (int)(t % Cin);
        const int co = (int)(t / Cin);
        const int g = cin / Cg, ci = cin - g * Cg;
        const int f = 1 << g, k = f + 2;
        const float* tg = g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3;
        const float* base = tg + ((long)ci * Co + co) * k * k;
        float acc = 0.f;
        for (int u = 2 - r; u <= f + 1 - r; ++u)
            for (int v = 2 - s_; v <= f + 1 - s_; ++v) acc += base[u * k + v];
        dw[i] = acc;
    }
}

// [N,3,H,W] fp32 -> [N,H,W,4] (4th channel zero); 16-bit storage: [N,H,W,16] (channels 3..15 zero — the MFMA gather of the
// 16-bit path fetches 8-channel pieces of 16-channel blocks)
template <int AT>
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, void* __restrict__ out, int N, long HW, int packed) {
    // grid (pixel blocks, N): the image index comes from blockIdx.y (a flat index cost a 64-bit division per pixel)
    const long n = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const long i = n * HW + p;
        const float* b = x + n * 3 * HW + p;
        const f32x4 v = {b[0], b[HW], b[2 * HW], 0.f};
        if (AT == 0 || packed) {
            dbn_st4<AT>(out, i, v);
        } else {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            dbn_st4<AT>(out, 4 * i, v);
            dbn_st4<AT>(out, 4 * i + 1, z);
            dbn_st4<AT>(out, 4 * i + 2, z);
            dbn_st4<AT>(out, 4 * i + 3, z);
        }
    }
}

// 16-bit storage, training: the 16-channel-block form (the stem conv's source) AND the packed 4-channel form (the X operand of the stem's
// weight gradient) in ONE pass over the image, with 16-byte stores (round 5: the two launches of the kernel above — three strided
// 4-byte loads and four 8-byte stores per thread each — were 0.22 ms at the head of every bf16 step with nothing beside them)
template <int AT>
__global__ void nchw3_to_nhwc16_and_4_kernel(const float* __restrict__ x, void* __restrict__ out16, void* __restrict__ out4, int N, long HW) {
    static_assert(AT != 0, "16-bit storage");
    const long total = (long)N * HW;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i - n * HW;
        const float* b = x + n * 3 * HW + p;
        const f32x4 v = {b[0], b[HW], b[2 * HW], 0.f};
        const f32x4 lo[2] = {v, z}, hi[2] = {z, z};
        dbn_stq<AT>(out16, 2 * i, lo);
        dbn_stq<AT>(out16, 2 * i + 1, hi);
        if (out4) dbn_st4<AT>(out4, i, v);
    }
}

template <int AT>
__global__ void axpy_kernel(const void* __restrict__ x, void* __restrict__ y, long total4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x)
        dbn_st4<AT>(y, i, dbn_ld4<AT>(y, i) + dbn_ld4<AT>(x, i));
}

// ----------------------------------------------------------------------------------
// Adam over one flat fp32 buffer (torch.optim.Adam semantics, amsgrad=False, wd=0)
// ----------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr_over_bc1, float b1, float b2, float eps, float inv_sqrt_bc2, float gscale) {
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
            const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
            pp[e] -= lr_over_bc1 * (mm[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    // tail
    const long t = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t < n) {
        const float gg = g[t] * gscale;
        const float mm = b1 * m[t] + (1.f - b1) * gg;
        const float vv = b2 * v[t] + (1.f - b2) * gg * gg;
        m[t] = mm;
        v[t] = vv;
        p[t] -= lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    }
}

inline int part_blocks(int M, int C) {
    // rows per block >= 64 so partials stay small; <= MAX_PART blocks
    int nb = (M + 63) / 64;
    if (nb > STREAM_BLOCKS) nb = STREAM_BLOCKS;
    if (nb < 1) nb = 1;
    (void)C;
    return nb;
}
// grid of a streaming BN kernel whose threads keep their channel quad: (grid * 256) % (C/4) == 0
inline int bn_stream_grid(long total4, int C) {
    int g = dbn_grid(total4, 256, STREAM_BLOCKS);
    int a = C / 4, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }  // a = gcd(C/4, 256)
    const int m = (C / 4) / a;                       // grid must be a multiple of m
    return (g + m - 1) / m * m;
}
inline dim3 red_grid(int nb, int C) { return dim3(nb, (C + CHUNK_C - 1) / CHUNK_C); }
inline size_t red_smem(int Cfull, int nv) {
    const int C = Cfull < CHUNK_C ? Cfull : CHUNK_C;
    const int nrl = 256 / (C / 4);
    return (size_t)(nrl > 0 ? nrl : 1) * nv * C * sizeof(float);
}

}  // namespace

// Measurement aid (bench.py): the shader clock the chip SUSTAINS while the step runs.  One wave samples s_memtime (ticks at
// the shader clock) and s_memrealtime (constant reference clock) around `ref_ticks` of s_sleep; launched on its own stream
// beside the step's kernels it costs one wave slot and no matrix-pipe time.  out = {shader cycles, reference ticks}.
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned long long ref_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ref_ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

extern "C" {

// Entry points come in two forms: `dbn_x(...)` with fp32 activation tensors (the contract of BASELINE configs[1]) and
// `dbn_x_t(at, ...)` with the activation storage type first (DBN_AT_F32 / _BF16 / _F16): tensors marked `void*` are stored
// in that type, everything `float*` (coefficients, statistics, partial sums, gradients of parameters) stays fp32.

// floats of scratch needed by the per-channel reduction entry points below
int dbn_reduce_ws_floats(int C) { return MAX_PART * 2 * C; }

int dbn_bn_train_stats_t(int at, const void* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                         float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                         float* ws, void* stream) {
    DBN_REQUIRE(y && gamma && beta && scale && shift && save_mean && save_rstd && ws);
    DBN_REQUIRE(M > 0 && C % 4 == 0 && C >= 4 && C <= 4096 && (C <= CHUNK_C || C % CHUNK_C == 0));
    hipStream_t st = (hipStream_t)stream;
    const int nb = part_blocks(M, C);
    DBN_DISPATCH_AT(at, {
        hipLaunchKernelGGL(bn_stats_kernel<AT>, red_grid(nb, C), dim3(256), red_smem(C, 2), st, y, M, C, ws);
        hipLaunchKernelGGL(bn_finalize_kernel<AT>, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, ws, nb, y, M, C, gamma, beta, eps,
                           momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    });
    return dbn_status();
}
int dbn_bn_train_stats(const float* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                       float* ws, void* stream) {
    return dbn_bn_train_stats_t(0, y, M, C, gamma, beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream);
}

int dbn_bn_eval_coef(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var, float eps,
                     float* scale, float* shift, void* stream) {
    DBN_REQUIRE(gamma && beta && run_mean && run_var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(dbn_ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                       run_mean, run_var, eps, scale, shift);
    return dbn_status();
}

// Eval-mode BatchNorm folded into the conv in front of it (inference; /root/reference/src/modules/basic.py:32-36 under model.eval(),
// test.py:53-59): w' [O][inner] = w * s[o], bias' [o] = beta + (bias - mean) * s[o], s = gamma / sqrt(var + eps) — the arithmetic of
// bn_eval_coef_kernel.  One launch per layer whenever the parameters or the running statistics change, i.e. once per checkpoint.
__global__ void fold_bn_eval_kernel(const float* __restrict__ w, int O, long inner, const float* __restrict__ bias, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                    float* __restrict__ w_out, float* __restrict__ b_out) {
    const long total = (long)O * inner;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i / inner);
        const float sc = gamma[o] * (1.f / sqrtf(var[o] + eps));  // (bn_eval_coef_kernel's scale, bit for bit)
        w_out[i] = w[i] * sc;
        if (i - (long)o * inner == 0) b_out[o] = fmaf((bias ? bias[o] : 0.f) - mean[o], sc, beta[o]);
    }
}
int dbn_fold_bn_eval(const float* w, int O, long inner, const float* bias, const float* gamma, const float* beta, const float* run_mean,
                     const float* run_var, float eps, float* w_out, float* b_out, void* stream) {
    DBN_REQUIRE(w && gamma && beta && run_mean && run_var && w_out && b_out && O > 0 && inner > 0);
    hipLaunchKernelGGL(fold_bn_eval_kernel, dim3(dbn_grid((long)O * inner)), dim3(256), 0, (hipStream_t)stream, w, O, inner, bias, gamma, beta,
                       run_mean, run_var, eps, w_out, b_out);
    return dbn_status();
}

int dbn_bn_apply_t(int at, const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                   const float* res_shift, void* out, long M, int C, int relu, void* stream) {
    DBN_REQUIRE(y && scale && shift && out && M > 0 && C % 4 == 0);
    DBN_REQUIRE((res_scale == nullptr) == (res_shift == nullptr));
    const long total4 = M * (C / 4);
    if (at != 0 && C % 8 == 0) {  // 16-bit storage: eight channels per 16-byte access
        const long total8 = M * (C / 8);
        if (at == 1)
            hipLaunchKernelGGL((bn_apply_kernel<1, 2>), dim3(bn_stream_grid(total8, C / 2)), dim3(256), 0, (hipStream_t)stream, y, scale, shift, res,
                               res_scale, res_shift, out, total8, C, relu);
        else
            hipLaunchKernelGGL((bn_apply_kernel<2, 2>), dim3(bn_stream_grid(total8, C / 2)), dim3(256), 0, (hipStream_t)stream, y, scale, shift, res,
                               res_scale, res_shift, out, total8, C, relu);
        return dbn_status();
    }
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL((bn_apply_kernel<AT, 1>), dim3(bn_stream_grid(total4, C)), dim3(256), 0, (hipStream_t)stream, y,
                                           scale, shift, res, res_scale, res_shift, out, total4, C, relu));
    return dbn_status();
}
int dbn_bn_apply(const float* y, const float* scale, const float* shift, const float* res, const float* res_scale,
                 const float* res_shift, float* out, long M, int C, int relu, void* stream) {
    return dbn_bn_apply_t(0, y, scale, shift, res, res_scale, res_shift, out, M, C, relu, stream);
}

// The general BatchNorm backward.  `sums` optional: [2][C] reductions already produced by the kernel that wrote dout (then
// only finalize + apply run).  `dbias_conv` optional [C]: column sums of dy (times grad_scale) = gradient of the bias of the
// convolution that produced y, formed inside the apply pass instead of by a dbn_col_sum pass over dy (needs 256 % (C/4) == 0).
// sums_parts: number of partial columns of `sums` ([2*C][sums_parts], 1 = already folded; ignored without `sums`).
int dbn_bn_backward_t(int at, const float* sums, int sums_parts, const void* y, const void* zmask, const float* mask_scale, const float* mask_shift,
                      const void* dout, const float* save_mean, const float* save_rstd, const float* gamma, void* dy, void* gout,
                      int gout_accumulate, float* dgamma, float* dbeta, float* dbias_conv, int M, int C, float grad_scale, float* ws,
                      void* stream) {
    DBN_REQUIRE(y && dout && save_mean && save_rstd && gamma && dy && dgamma && dbeta && ws);
    DBN_REQUIRE(!dbias_conv || 256 % (C / 4) == 0);
    DBN_REQUIRE(M > 0 && C % 4 == 0 && C >= 4 && C <= 4096 && (C <= CHUNK_C || C % CHUNK_C == 0));
    DBN_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr) && !(zmask && mask_scale));
    hipStream_t st = (hipStream_t)stream;
    const int nb = part_blocks(M, C);
    float* c1 = ws + (long)MAX_PART * 2 * C - 2 * C;  // tail of the scratch (nb <= MAX_PART-1 partial rows used)
    float* c2 = c1 + C;
    const int nbu = nb < MAX_PART ? nb : MAX_PART - 1;
    const bool wide = at != 0 && C % 8 == 0 && (!dbias_conv || 256 % (C / 8) == 0);  // 16-bit storage: eight channels per 16-byte access
    const long total4 = (long)M * (C / (wide ? 8 : 4));
    const int grid = bn_stream_grid(total4, wide ? C / 2 : C);
    DBN_DISPATCH_AT(at, {
        if (sums && sums_parts < 0) {
            // already finalized by the producing kernel (dbn_bnb_final): sums = [2][C] = c1, c2; dgamma / dbeta are written
            c1 = const_cast<float*>(sums);
            c2 = c1 + C;
        } else if (sums && sums_parts >= 512) {
            hipLaunchKernelGGL(bn_bwd_finalize_wide_kernel, dim3(C), dim3(256), 0, st, sums, sums_parts, M, C, dgamma, dbeta, c1, c2,
                               grad_scale);
        } else if (sums) {
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, sums, sums_parts > 0 ? sums_parts : 1, M,
                               C, dgamma, dbeta, c1, c2, grad_scale);
        } else {
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<AT>, red_grid(nbu, C), dim3(256), red_smem(C, 2), st, y, zmask, mask_scale,
                               mask_shift, dout, save_mean, save_rstd, M, C, ws);
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, ws, nbu, M, C, dgamma, dbeta, c1, c2,
                               grad_scale);
        }
        // the reduce partials at the front of ws have been consumed by the finalize kernel: the bias partials [C][grid] reuse them
#ifndef DBN_DBG_SKIP_BN_BWD_APPLY
#define DBN_DBG_SKIP_BN_BWD_APPLY 0  // 1 (timing-only A/B build, wrong results): the apply pass is not launched — the upper bound of what fusing it into its consumers could save
#endif
        if constexpr (AT != 0) {
            if (wide && !DBN_DBG_SKIP_BN_BWD_APPLY)
                hipLaunchKernelGGL((bn_bwd_apply_kernel<AT, 2>), dim3(grid), dim3(256), 0, st, y, zmask, mask_scale, mask_shift, dout, save_mean,
                                   save_rstd, gamma, c1, c2, dy, gout, gout_accumulate, total4, C, dbias_conv ? ws : nullptr);
        }
        if (!wide && !DBN_DBG_SKIP_BN_BWD_APPLY)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<AT, 1>), dim3(grid), dim3(256), 0, st, y, zmask, mask_scale, mask_shift, dout, save_mean,
                               save_rstd, gamma, c1, c2, dy, gout, gout_accumulate, total4, C, dbias_conv ? ws : nullptr);
    });
    if (dbias_conv)
        hipLaunchKernelGGL(fold_partials_kernel, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, ws, grid, C, dbias_conv, grad_scale);
    return dbn_status();
}

int dbn_bn_backward(const float* y, const float* zmask, const float* mask_scale, const float* mask_shift, const float* dout,
                    const float* save_mean, const float* save_rstd, const float* gamma, float* dy, float* gout, int gout_accumulate,
                    float* dgamma, float* dbeta, int M, int C, float grad_scale, float* ws, void* stream) {
    return dbn_bn_backward_t(0, nullptr, 0, y, zmask, mask_scale, mask_shift, dout, save_mean, save_rstd, gamma, dy, gout, gout_accumulate,
                             dgamma, dbeta, nullptr, M, C, grad_scale, ws, stream);
}

// BatchNorm backward whose two per-channel reductions were already produced by the kernel that wrote dout
// (sums[0][c] = sum of the ReLU-masked dout, sums[1][c] = sum of masked dout * xhat): only finalize + apply run.
int dbn_bn_backward_from_sums(const float* sums, const float* y, const float* zmask, const float* mask_scale, const float* mask_shift,
                              const float* dout, const float* save_mean, const float* save_rstd, const float* gamma, float* dy,
                              float* gout, int gout_accumulate, float* dgamma, float* dbeta, int M, int C, float grad_scale,
                              float* ws, void* stream) {
    DBN_REQUIRE(sums);
    return dbn_bn_backward_t(0, sums, 1, y, zmask, mask_scale, mask_shift, dout, save_mean, save_rstd, gamma, dy, gout, gout_accumulate,
                             dgamma, dbeta, nullptr, M, C, grad_scale, ws, stream);
}

int dbn_bn_backward_ex(const float* sums, const float* y, const float* zmask, const float* mask_scale, const float* mask_shift,
                       const float* dout, const float* save_mean, const float* save_rstd, const float* gamma, float* dy, float* gout,
                       int gout_accumulate, float* dgamma, float* dbeta, float* dbias_conv, int M, int C, float grad_scale, float* ws,
                       void* stream) {
    return dbn_bn_backward_t(0, sums, 1, y, zmask, mask_scale, mask_shift, dout, save_mean, save_rstd, gamma, dy, gout, gout_accumulate,
                             dgamma, dbeta, dbias_conv, M, C, grad_scale, ws, stream);
}

int dbn_col_sum_t(int at, const void* x, int M, int C, float* out, float scale, float* ws, void* stream) {
    DBN_REQUIRE(x && out && ws && M > 0 && C % 4 == 0 && C >= 4 && C <= 4096 && (C <= CHUNK_C || C % CHUNK_C == 0));
    hipStream_t st = (hipStream_t)stream;
    const int nb = part_blocks(M, C);
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(col_sum_kernel<AT>, red_grid(nb, C), dim3(256), red_smem(C, 1), st, x, M, C, ws));
    hipLaunchKernelGGL(fold_partials_kernel, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, ws, nb, C, out, scale);
    return dbn_status();
}
int dbn_col_sum(const float* x, int M, int C, float* out, float scale, float* ws, void* stream) {
    return dbn_col_sum_t(0, x, M, C, out, scale, ws, stream);
}

int dbn_bnrelu_maxpool_fwd_t(int at, const void* y, const float* scale, const float* shift, void* out, int N, int H, int W, int C,
                             void* stream) {
    DBN_REQUIRE(y && scale && shift && out && C % 4 == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if ((at == 1 || at == 2) && C % 8 == 0) {
        const dim3 grid(dbn_grid((long)N * ((Ho + 1) / 2) * Wo * (C / 8)));
        if (at == 1)
            hipLaunchKernelGGL((bnrelu_maxpool_fwd_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, N, H, W, C, Ho, Wo);
        else
            hipLaunchKernelGGL((bnrelu_maxpool_fwd_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, N, H, W, C, Ho, Wo);
        return dbn_status();
    }
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL((bnrelu_maxpool_fwd_kernel<AT, 1>), dim3(dbn_grid((long)N * ((Ho + 1) / 2) * Wo * (C / 4))), dim3(256), 0,
                                           (hipStream_t)stream, y, scale, shift, out, N, H, W, C, Ho, Wo));
    return dbn_status();
}
int dbn_bnrelu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, int N, int H, int W, int C,
                           void* stream) {
    return dbn_bnrelu_maxpool_fwd_t(0, y, scale, shift, out, N, H, W, C, stream);
}

// bn_mean / bn_rstd / bn_part optional (all or none): also emit the partial sums of the following BatchNorm backward,
// bn_part = [2*C][dbn_maxpool_bwd_parts(...)] floats — feed them to dbn_bn_backward_t as `sums` with that `sums_parts`.
int dbn_maxpool_bwd_parts(int N, int H, int W, int C) { return dbn_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4), 256, 2048); }
int dbn_bnrelu_maxpool_bwd_t(int at, const void* y, const float* scale, const float* shift, const void* pooled, const void* dpool,
                             void* dz, int N, int H, int W, int C, const float* bn_mean, const float* bn_rstd, float* bn_part,
                             void* stream) {
    DBN_REQUIRE(y && scale && shift && pooled && dpool && dz && C % 4 == 0);
    DBN_REQUIRE((bn_part == nullptr) == (bn_mean == nullptr) && (bn_part == nullptr) == (bn_rstd == nullptr));
    DBN_REQUIRE(!bn_part || 256 % (C / 4) == 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    // (the grid — and with it the number of partial rows the caller sized bn_part for, dbn_maxpool_bwd_parts — is the same for both forms;
    // the 16-byte form just walks half as many items)
#ifndef DBN_POOL_BWD_QW2
#define DBN_POOL_BWD_QW2 0  // the 16-byte form of the BACKWARD measured slower (its 2 x 2 x 2 window arrays double: bf16 step 1663 / 1671 / 1662 / 1665 images/s with the four-channel form, 1636 / 1632 / 1639 / 1631 with this one, interleaved on one box): off
#endif
    if (DBN_POOL_BWD_QW2 && (at == 1 || at == 2) && C % 8 == 0) {
        const dim3 grid(dbn_maxpool_bwd_parts(N, H, W, C));
        if (at == 1)
            hipLaunchKernelGGL((bnrelu_maxpool_bwd_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, pooled, dpool, dz, N, H, W,
                               C, Ho, Wo, bn_mean, bn_rstd, bn_part);
        else
            hipLaunchKernelGGL((bnrelu_maxpool_bwd_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, pooled, dpool, dz, N, H, W,
                               C, Ho, Wo, bn_mean, bn_rstd, bn_part);
        return dbn_status();
    }
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL((bnrelu_maxpool_bwd_kernel<AT, 1>), dim3(dbn_maxpool_bwd_parts(N, H, W, C)), dim3(256), 0,
                                           (hipStream_t)stream, y, scale, shift, pooled, dpool, dz, N, H, W, C, Ho, Wo, bn_mean, bn_rstd,
                                           bn_part));
    return dbn_status();
}
int dbn_bnrelu_maxpool_bwd(const float* y, const float* scale, const float* shift, const float* pooled, const float* dpool,
                           float* dz, int N, int H, int W, int C, void* stream) {
    return dbn_bnrelu_maxpool_bwd_t(0, y, scale, shift, pooled, dpool, dz, N, H, W, C, nullptr, nullptr, nullptr, stream);
}

// ---- MaxPool2d(3, 2, 1) over relu(bn(y)) with the recorded argmax, and the backward through pool, ReLU and the BatchNorm in one pass
// (see bnrelu_maxpool_fwd_arg_kernel).  idx: N*Ho*Wo*C bytes; ypool: [N,Ho,Wo,C] in the storage type.
int dbn_bnrelu_maxpool_fwd_arg_t(int at, const void* y, const float* scale, const float* shift, void* out, void* idx, void* ypool, int N, int H,
                                 int W, int C, void* stream) {
    DBN_REQUIRE(y && scale && shift && out && idx && ypool && C % 4 == 0 && N > 0 && H > 0 && W > 0);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    unsigned* ix = reinterpret_cast<unsigned*>(idx);
    if ((at == 1 || at == 2) && C % 8 == 0) {
        const dim3 grid(dbn_grid((long)N * ((Ho + 1) / 2) * Wo * (C / 8)));
        if (at == 1)
            hipLaunchKernelGGL((bnrelu_maxpool_fwd_arg_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, ix, ypool, N, H, W, C, Ho, Wo);
        else
            hipLaunchKernelGGL((bnrelu_maxpool_fwd_arg_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, ix, ypool, N, H, W, C, Ho, Wo);
        return dbn_status();
    }
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL((bnrelu_maxpool_fwd_arg_kernel<AT, 1>), dim3(dbn_grid((long)N * ((Ho + 1) / 2) * Wo * (C / 4))), dim3(256), 0,
                                           (hipStream_t)stream, y, scale, shift, out, ix, ypool, N, H, W, C, Ho, Wo));
    return dbn_status();
}
#ifndef DBN_POOL_APPLY_QW2
#define DBN_POOL_APPLY_QW2 1  // 16-bit storage: the apply pass with 16-byte accesses (A/B switch)
#endif
static int maxpool_bn_parts(int N, int Ho, int Wo, int C) { return dbn_grid((long)N * Ho * Wo * (C / 4), 256, 2048); }
// floats of scratch dbn_maxpool_bn_backward_t needs
long dbn_maxpool_bn_backward_ws_floats(int N, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    return 2L * C * maxpool_bn_parts(N, Ho, Wo, C) + 2L * C;
}
// dy [N,H,W,C] = the gradient at the conv output y of  pool(relu(bn(y)))  given dpool [N,Ho,Wo,C] (train-mode BatchNorm: saved mean / rstd,
// gamma; dgamma / dbeta are written, times grad_scale).  idx / ypool: as written by dbn_bnrelu_maxpool_fwd_arg_t for this y.
int dbn_maxpool_bn_backward_t(int at, const void* y, const void* dpool, const void* idx, const void* ypool, const float* save_mean,
                              const float* save_rstd, const float* gamma, void* dy, float* dgamma, float* dbeta, int N, int H, int W, int C,
                              float grad_scale, float* ws, void* stream) {
    DBN_REQUIRE(y && dpool && idx && ypool && save_mean && save_rstd && gamma && dy && dgamma && dbeta && ws);
    DBN_REQUIRE(N > 0 && H > 0 && W > 0 && C % 4 == 0 && 256 % (C / 4) == 0 && (long)N * H * W < (1L << 31));
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int parts = maxpool_bn_parts(N, Ho, Wo, C);
    float* part = ws;
    float* c1 = ws + 2L * C * parts;
    float* c2 = c1 + C;
    hipStream_t st = (hipStream_t)stream;
    const unsigned* ix = reinterpret_cast<const unsigned*>(idx);
    const long pooled4 = (long)N * Ho * Wo * (C / 4);
    const int M = N * H * W;
    const dim3 agrid(dbn_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4)));
    DBN_DISPATCH_AT(at, {
        hipLaunchKernelGGL(maxpool_bn_stats_kernel<AT>, dim3(parts), dim3(256), 0, st, dpool, ix, ypool, save_mean, save_rstd, part, pooled4, C);
        if (parts >= 512)
            hipLaunchKernelGGL(bn_bwd_finalize_wide_kernel, dim3(C), dim3(256), 0, st, part, parts, M, C, dgamma, dbeta, c1, c2, grad_scale);
        else
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dbn_ceil_div(C, 8)), dim3(256), 0, st, part, parts, M, C, dgamma, dbeta, c1, c2, grad_scale);
        bool wide = false;
        if constexpr (AT != 0) {
            if (DBN_POOL_APPLY_QW2 && C % 8 == 0) {
                wide = true;
                const dim3 wgrid(dbn_grid((long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8)));
                hipLaunchKernelGGL((maxpool_bn_bwd_apply_kernel<AT, 2>), wgrid, dim3(256), 0, st, y, dpool, ix, save_mean, save_rstd, gamma, c1, c2, dy, N,
                                   H, W, C, Ho, Wo);
            }
        }
        if (!wide)
            hipLaunchKernelGGL((maxpool_bn_bwd_apply_kernel<AT, 1>), agrid, dim3(256), 0, st, y, dpool, ix, save_mean, save_rstd, gamma, c1, c2, dy, N, H,
                               W, C, Ho, Wo);
    });
    return dbn_status();
}

int dbn_nearest_up_fwd_t(int at, const void* src, const void* addend, void* dst, int N, int Hs, int Ws, int C, int H, int W, int Cdst,
                         int coff, void* stream) {
    DBN_REQUIRE(src && dst && C % 4 == 0 && Cdst % 4 == 0 && coff % 4 == 0 && coff + C <= Cdst);
    DBN_REQUIRE(addend == nullptr || (Cdst == C && coff == 0));
    if ((at == 1 || at == 2) && C % 8 == 0 && Cdst % 8 == 0 && coff % 8 == 0) {
        const dim3 grid(dbn_grid((long)N * H * W * (C / 8)));
        if (at == 1)
            hipLaunchKernelGGL((nearest_up_fwd_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, src, addend, dst, N, Hs, Ws, C, H, W, Cdst, coff);
        else
            hipLaunchKernelGGL((nearest_up_fwd_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, src, addend, dst, N, Hs, Ws, C, H, W, Cdst, coff);
        return dbn_status();
    }
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL((nearest_up_fwd_kernel<AT, 1>), dim3(dbn_grid((long)N * H * W * (C / 4))), dim3(256), 0,
                                           (hipStream_t)stream, src, addend, dst, N, Hs, Ws, C, H, W, Cdst, coff));
    return dbn_status();
}
int dbn_nearest_up_fwd(const float* src, const float* addend, float* dst, int N, int Hs, int Ws, int C, int H, int W, int Cdst,
                       int coff, void* stream) {
    return dbn_nearest_up_fwd_t(0, src, addend, dst, N, Hs, Ws, C, H, W, Cdst, coff, stream);
}

int dbn_nearest_up_bwd_t(int at, const void* dbig, void* dsrc, int N, int Hs, int Ws, int C, int H, int W, int Cbig, int coff,
                         int accumulate, void* stream) {
    DBN_REQUIRE(dbig && dsrc && C % 4 == 0 && Cbig % 4 == 0 && coff % 4 == 0 && coff + C <= Cbig);
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(nearest_up_bwd_kernel<AT>, dim3(dbn_grid((long)N * Hs * Ws * (C / 4))), dim3(256), 0,
                                           (hipStream_t)stream, dbig, dsrc, N, Hs, Ws, C, H, W, Cbig, coff, accumulate));
    return dbn_status();
}
int dbn_nearest_up_bwd(const float* dbig, float* dsrc, int N, int Hs, int Ws, int C, int H, int W, int Cbig, int coff,
                       int accumulate, void* stream) {
    return dbn_nearest_up_bwd_t(0, dbig, dsrc, N, Hs, Ws, C, H, W, Cbig, coff, accumulate, stream);
}

// x: [N,3,H,W] fp32.  out: [N,H,W,4] fp32 (at = 0) or [N,H,W,16] in the 16-bit storage type (channels 3.. zero).
int dbn_nchw3_to_nhwc4_t(int at, const float* x, void* out, int N, int H, int W, void* stream) {
    DBN_REQUIRE(x && out && N > 0);
    DBN_REQUIRE(N <= 65535);
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(nchw3_to_nhwc4_kernel<AT>, dim3(dbn_grid((long)H * W), N), dim3(256), 0, (hipStream_t)stream,
                                           x, out, N, (long)H * W, 0));
    return dbn_status();
}
// [N,H,W,4] in the storage type whatever it is (16-bit: 8 bytes per pixel): the X operand of the stem's WEIGHT GRADIENT, whose
// columns are (tap, channel) — over the 16-channel form 13 of every 16 columns would be zeros (measured in bf16: 0.48 ms, the
// largest weight gradient of the step, against 0.13 ms on this form)
int dbn_nchw3_to_nhwc4_packed_t(int at, const float* x, void* out, int N, int H, int W, void* stream) {
    DBN_REQUIRE(x && out && N > 0);
    DBN_REQUIRE(N <= 65535);
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(nchw3_to_nhwc4_kernel<AT>, dim3(dbn_grid((long)H * W), N), dim3(256), 0, (hipStream_t)stream,
                                           x, out, N, (long)H * W, 1));
    return dbn_status();
}
// 16-bit storage: out16 [N,H,W,16] (channels 3.. zero) and — out4 non-NULL — the packed [N,H,W,4] form, in one launch
int dbn_nchw3_to_nhwc16_and_4_t(int at, const float* x, void* out16, void* out4, int N, int H, int W, void* stream) {
    DBN_REQUIRE(x && out16 && N > 0 && (at == 1 || at == 2));
    if (at == 1)
        hipLaunchKernelGGL(nchw3_to_nhwc16_and_4_kernel<1>, dim3(dbn_grid((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x, out16, out4, N, (long)H * W);
    else
        hipLaunchKernelGGL(nchw3_to_nhwc16_and_4_kernel<2>, dim3(dbn_grid((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x, out16, out4, N, (long)H * W);
    return dbn_status();
}
int dbn_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, void* stream) { return dbn_nchw3_to_nhwc4_t(0, x, out, N, H, W, stream); }

static float bilinear_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// dst[planes,H,W] = bilinear(align_corners=True) of src[planes,Hs,Ws] (NCHW planes, planes = N*C)
int dbn_bilinear_fwd(const float* src, float* dst, long planes, int Hs, int Ws, int H, int W, void* stream) {
    DBN_REQUIRE(src && dst && planes > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(dbn_grid(planes * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, Hs, Ws, H,
                       W, bilinear_scale(Hs, H), bilinear_scale(Ws, W));
    return dbn_status();
}

// dsrc[planes,Hs,Ws] = adjoint of the above applied to ddst[planes,H,W]
int dbn_bilinear_bwd(const float* ddst, float* dsrc, long planes, int Hs, int Ws, int H, int W, void* stream) {
    DBN_REQUIRE(ddst && dsrc && planes > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(dbn_grid(planes * Hs * Ws)), dim3(256), 0, (hipStream_t)stream, ddst, dsrc, planes, Hs,
                       Ws, H, W, bilinear_scale(Hs, H), bilinear_scale(Ws, W));
    return dbn_status();
}

// wd: [Cg][Co][f+2][f+2] combined weights of input-channel group g (upsample factor f = 2^g) of a 3x3 conv [Co][Cin][3][3]
int dbn_fpn_combine_weights(const float* w, int Co, int Cin, int group, int Cg, float* wd, void* stream) {
    DBN_REQUIRE(w && wd && group >= 0 && group < 4 && Cg > 0 && (group + 1) * Cg <= Cin);
    const int f = 1 << group;
    hipLaunchKernelGGL(fpn_combine_weights_kernel, dim3(dbn_grid((long)Cg * Co * (f + 2) * (f + 2))), dim3(256), 0, (hipStream_t)stream,
                       w, Co, Cin, group, Cg, f, wd);
    return dbn_status();
}

// dw[Co][4*Cg][3][3] from the four strided-conv weight gradients t_g[Cg][Co][2^g+2][2^g+2]
int dbn_fpn_scatter_wgrad(const float* t0, const float* t1, const float* t2, const float* t3, int Co, int Cg, float* dw,
                          void* stream) {
    DBN_REQUIRE(t0 && t1 && t2 && t3 && dw && Co > 0 && Cg > 0);
    hipLaunchKernelGGL(fpn_scatter_wgrad_kernel, dim3(dbn_grid((long)Co * 4 * Cg * 9)), dim3(256), 0, (hipStream_t)stream, t0, t1, t2,
                       t3, Co, 4 * Cg, Cg, dw);
    return dbn_status();
}

int dbn_add_inplace(const float* x, float* y, long n, void* stream) {
    DBN_REQUIRE(x && y && n % 4 == 0);
    hipLaunchKernelGGL(axpy_kernel<0>, dim3(dbn_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4);
    return dbn_status();
}

int dbn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, int step,
                  float grad_scale, void* stream) {
    DBN_REQUIRE(p && g && m && v && n > 0 && step >= 1);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(dbn_grid(n / 4 + 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    return dbn_status();
}

int dbn_clock_probe(void* out2, int microseconds, void* stream) {
    DBN_REQUIRE(out2 && microseconds > 0 && microseconds <= 100000);
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        khz = 100000;  // gfx9 reference clock: 100 MHz
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out2,
                       (unsigned long long)microseconds * (unsigned long long)khz / 1000ull);
    return dbn_status();
}

int dbn_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        return 100000;
    return khz;
}

}  // extern "C"
