// Differentiable-binarization head tail and the DBLoss stack (HBM-bound kernels).
//
//   dbn_head_tail_fwd   last ConvTranspose2d(64->1,k2,s2)+Sigmoid of both branches and
//                       B = 1/(1+exp(-k(P-T)))  -> NCHW planes [N,3|2,H,W]
//                       (/root/reference/src/modules/segmentation_head.py:28-29,35-45,77-79,106-108)
//   dbn_head_tail_bwd   d(preds) -> gradients of the two 64-channel inputs, the two
//                       ConvT weights and biases
//   dbn_db_loss_fwd     OHEM-BCE / masked L1 / Dice sums -> 5 losses
//                       (/root/reference/src/losses.py:18-40,48-66,75-82,105-139)
//   dbn_db_loss_bwd     d(losses)/d(preds)
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// 16 lanes cooperate on one quarter-resolution pixel: lane q owns channels 4q..4q+3.
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// xb/xt: [N,Hq,Wq,64] inputs of the last ConvT (already BN+ReLU'd);  wb/wt: [64][4]
// (ConvTranspose2d weight [64,1,2,2]); out: [N,CH,2Hq,2Wq], CH=3 (train) or 2 (eval).
__global__ void head_tail_fwd_kernel(const float* __restrict__ xb, const float* __restrict__ xt, const float* __restrict__ wb,
                                     const float* __restrict__ wt, const float* __restrict__ bias_b,
                                     const float* __restrict__ bias_t, float* __restrict__ out, int N, int Hq, int Wq, int CH,
                                     float kstep) {
    const int q = threadIdx.x & 15;
    const long npx = (long)N * Hq * Wq;
    const long gstride = (long)gridDim.x * (blockDim.x >> 4);
    // weights of this lane's 4 channels: w[ci][ab]
    f32x4 wbq[4], wtq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wbq[e] = *reinterpret_cast<const f32x4*>(wb + (4 * q + e) * 4);
        wtq[e] = *reinterpret_cast<const f32x4*>(wt + (4 * q + e) * 4);
    }
    const float bb = bias_b[0], bt = bias_t[0];
    const int H = 2 * Hq, W = 2 * Wq;
    const long HW = (long)H * W;
    for (long px = blockIdx.x * (long)(blockDim.x >> 4) + (threadIdx.x >> 4); px < npx; px += gstride) {
        const f32x4 vb = *reinterpret_cast<const f32x4*>(xb + px * 64 + 4 * q);
        const f32x4 vt = *reinterpret_cast<const f32x4*>(xt + px * 64 + 4 * q);
        f32x4 sb = {0.f, 0.f, 0.f, 0.f}, stt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sb += vb[e] * wbq[e];
            stt += vt[e] * wtq[e];
        }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            sb[ab] = group16_sum(sb[ab]);
            stt[ab] = group16_sum(stt[ab]);
        }
        if (q < 4) {
            const int ab = q;
            const float lp = (ab == 0 ? sb[0] : ab == 1 ? sb[1] : ab == 2 ? sb[2] : sb[3]) + bb;
            const float lt = (ab == 0 ? stt[0] : ab == 1 ? stt[1] : ab == 2 ? stt[2] : stt[3]) + bt;
            const float P = sigmoid_acc(lp), T = sigmoid_acc(lt);
            const long n = px / ((long)Hq * Wq);
            const long rem = px - n * (long)Hq * Wq;
            const int hq = (int)(rem / Wq), wq = (int)(rem - (long)hq * Wq);
            const long o = (long)(2 * hq + (ab >> 1)) * W + 2 * wq + (ab & 1);
            float* base = out + n * CH * HW + o;
            base[0] = P;
            base[HW] = T;
            if (CH == 3) base[2 * HW] = 1.f / (1.f + expf(-kstep * (P - T)));
        }
    }
}

// Backward.  For each quarter pixel: dl_b[ab], dl_t[ab] (grad wrt the two logits) from
// dpreds and the saved maps; dxb = sum_ab dl_b[ab]*wb[ci][ab]; dwb[ci][ab] += xb[ci]*dl_b[ab].
// part: [grid][2*(256+1)] block partials of (dwb[64*4], dbias_b, dwt[64*4], dbias_t).
__global__ void head_tail_bwd_kernel(const float* __restrict__ xb, const float* __restrict__ xt, const float* __restrict__ wb,
                                     const float* __restrict__ wt, const float* __restrict__ preds,
                                     const float* __restrict__ dpreds, float* __restrict__ dxb, float* __restrict__ dxt,
                                     float* __restrict__ part, int N, int Hq, int Wq, int CH, float kstep) {
    const int q = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const long npx = (long)N * Hq * Wq;
    const long gstride = (long)gridDim.x * (blockDim.x >> 4);
    f32x4 wbq[4], wtq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wbq[e] = *reinterpret_cast<const f32x4*>(wb + (4 * q + e) * 4);
        wtq[e] = *reinterpret_cast<const f32x4*>(wt + (4 * q + e) * 4);
    }
    const int H = 2 * Hq, W = 2 * Wq;
    const long HW = (long)H * W;
    f32x4 awb[4], awt[4];  // [e][ab]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        awb[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        awt[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float abb = 0.f, abt = 0.f;
    for (long px = blockIdx.x * (long)(blockDim.x >> 4) + grp; px < npx; px += gstride) {
        const long n = px / ((long)Hq * Wq);
        const long rem = px - n * (long)Hq * Wq;
        const int hq = (int)(rem / Wq), wq = (int)(rem - (long)hq * Wq);
        // lanes 0..3 of the group each evaluate one (a,b) position, then broadcast
        float dlb = 0.f, dlt = 0.f;
        if (q < 4) {
            const long o = (long)(2 * hq + (q >> 1)) * W + 2 * wq + (q & 1);
            const float* pb = preds + n * CH * HW + o;
            const float* db = dpreds + n * CH * HW + o;
            const float P = pb[0], T = pb[HW];
            float dP = db[0], dT = db[HW];
            if (CH == 3) {
                const float B = pb[2 * HW];
                const float gB = db[2 * HW] * kstep * B * (1.f - B);
                dP += gB;
                dT -= gB;
            }
            dlb = dP * P * (1.f - P);
            dlt = dT * T * (1.f - T);
        }
        f32x4 lb, lt;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            lb[ab] = __shfl(dlb, (threadIdx.x & 48) + ab, 64);
            lt[ab] = __shfl(dlt, (threadIdx.x & 48) + ab, 64);
        }
        const f32x4 vb = *reinterpret_cast<const f32x4*>(xb + px * 64 + 4 * q);
        const f32x4 vt = *reinterpret_cast<const f32x4*>(xt + px * 64 + 4 * q);
        f32x4 gb, gt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gb[e] = wbq[e][0] * lb[0] + wbq[e][1] * lb[1] + wbq[e][2] * lb[2] + wbq[e][3] * lb[3];
            gt[e] = wtq[e][0] * lt[0] + wtq[e][1] * lt[1] + wtq[e][2] * lt[2] + wtq[e][3] * lt[3];
            awb[e] += vb[e] * lb;
            awt[e] += vt[e] * lt;
        }
        *reinterpret_cast<f32x4*>(dxb + px * 64 + 4 * q) = gb;
        *reinterpret_cast<f32x4*>(dxt + px * 64 + 4 * q) = gt;
        if (q == 0) {
            abb += lb[0] + lb[1] + lb[2] + lb[3];
            abt += lt[0] + lt[1] + lt[2] + lt[3];
        }
    }
    // block reduction: groups with the same q hold the same channel slots
    __shared__ float red[16][2 * 257];  // [grp][...]
    {
        float* r = red[grp];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                r[(4 * q + e) * 4 + ab] = awb[e][ab];
                r[257 + (4 * q + e) * 4 + ab] = awt[e][ab];
            }
        if (q == 0) {
            r[256] = abb;
            r[257 + 256] = abt;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 257; i += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < 16; ++g) s += red[g][i];
        part[(long)i * gridDim.x + blockIdx.x] = s;  // transposed: [514][blocks]
    }
}

__global__ void fold_partials_d_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ out, float scale) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (i >= n) return;
    const double s = dbn_team32_fold(part, nb, i, l32);
    if (l32 == 0) out[i] = (float)(s * scale);
}

// ----------------------------------------------------------------------------------
// loss
// ----------------------------------------------------------------------------------
enum { S_POS = 0, S_NEG, S_BCE, S_L1, S_A, S_BGM, S_BM, NSUM };

__global__ void db_loss_fwd_kernel(const float* __restrict__ preds, const float* __restrict__ gts, int N, long HW, int CH,
                                   double* __restrict__ part) {
    const long total4 = (long)N * HW / 4;
    const long NHW = (long)N * HW;
    float s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = 0.f;
    double ds[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) ds[k] = 0.0;
    int cnt = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long p = i * 4;
        const long n = p / HW, o = p - n * HW;
        const float* pb = preds + n * CH * HW + o;
        const f32x4 P = *reinterpret_cast<const f32x4*>(pb);
        const f32x4 T = *reinterpret_cast<const f32x4*>(pb + HW);
        f32x4 B = {0.f, 0.f, 0.f, 0.f};
        if (CH == 3) B = *reinterpret_cast<const f32x4*>(pb + 2 * HW);
        const f32x4 G = *reinterpret_cast<const f32x4*>(gts + p);
        const f32x4 M = *reinterpret_cast<const f32x4*>(gts + NHW + p);
        const f32x4 Tg = *reinterpret_cast<const f32x4*>(gts + 2 * NHW + p);
        const f32x4 A = *reinterpret_cast<const f32x4*>(gts + 3 * NHW + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[S_POS] += G[e] * M[e];
            s[S_NEG] += (1.f - G[e]) * M[e];
            const float lp = fmaxf(logf(P[e]), -100.f);
            const float lq = fmaxf(log1pf(-P[e]), -100.f);
            s[S_BCE] += (G[e] - 1.f) * lq - G[e] * lp;
            s[S_L1] += fabsf(T[e] - Tg[e]) * A[e];
            s[S_A] += A[e];
            s[S_BGM] += B[e] * G[e] * M[e];
            s[S_BM] += B[e] * M[e];
        }
        if (++cnt == 64) {  // flush the fp32 running sums into doubles regularly
#pragma unroll
            for (int k = 0; k < NSUM; ++k) {
                ds[k] += (double)s[k];
                s[k] = 0.f;
            }
            cnt = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < NSUM; ++k) ds[k] += (double)s[k];
    __shared__ double red[4][NSUM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        const double w = dbn_wave_sum_d(ds[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NSUM) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x];
        part[(long)blockIdx.x * NSUM + threadIdx.x] = t;
    }
}

// losses[5] = prob, thresh, binary, prob+beta*thresh, total;  coef[8] for the backward:
//   0: c_bce = (sum_pos + n_neg)/(n_pos+n_neg+eps)/px   1: 1/(sum_A+eps)
//   2: dice U   3: dice I   4: has_pos flag (n_pos + n_neg > 0 ... always 1; kept for clarity)
__global__ void db_loss_finalize_kernel(const double* __restrict__ part, int nb, long px, int CH, float alpha, float beta,
                                        float negative_ratio, float eps, float* __restrict__ losses, float* __restrict__ coef) {
    // one 256-thread block: 32 lanes fold each of the NSUM (<= 8) partial columns, thread 0 finishes
    __shared__ double sums[8];
    {
        const int k = threadIdx.x >> 5, l32 = threadIdx.x & 31;
        double t = 0.0;
        if (k < NSUM)
            for (int b = l32; b < nb; b += 32) t += part[(long)b * NSUM + k];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (l32 == 0) sums[k] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double s[NSUM];
    for (int k = 0; k < NSUM; ++k) s[k] = sums[k];
    // losses.py:25-28 — int() truncations
    const long n_pos = (long)(float)s[S_POS];
    const long n_neg_expect = (long)((double)n_pos * (double)negative_ratio);
    const long n_neg_cur = (long)(float)s[S_NEG];
    const long n_neg = n_neg_expect < n_neg_cur ? n_neg_expect : n_neg_cur;
    const float bce = (float)(s[S_BCE] / (double)px);  // reduction='mean': scalar over all pixels
    // positive_loss.sum() = bce*sum_pos; topk(bce*negative, n_neg).sum() = bce*n_neg for binary maps
    const float denom = (float)((double)(n_pos + n_neg) + (double)eps);
    const float num_w = (float)s[S_POS] + (float)n_neg;
    const float prob = bce * num_w / denom;
    const float thr = (float)s[S_L1] / ((float)s[S_A] + eps);
    const float pt = prob + beta * thr;
    losses[0] = prob;
    losses[1] = thr;
    coef[0] = num_w / denom / (float)px;
    coef[1] = 1.f / ((float)s[S_A] + eps);
    if (CH == 3) {
        const float U = (float)s[S_BM] + (float)s[S_POS] + eps;
        const float I = (float)s[S_BGM];
        const float binl = 1.f - 2.f * I / U;
        losses[2] = binl;
        losses[3] = pt;
        losses[4] = alpha * binl + pt;
        coef[2] = U;
        coef[3] = I;
    } else {
        losses[2] = 0.f;
        losses[3] = pt;
        losses[4] = pt;
        coef[2] = 1.f;
        coef[3] = 0.f;
    }
}

// gout[5]: upstream grads of the 5 returned losses (device); dpreds planes like preds.
__global__ void db_loss_bwd_kernel(const float* __restrict__ preds, const float* __restrict__ gts, const float* __restrict__ coef,
                                   const float* __restrict__ gout, float alpha, float beta, int N, long HW, int CH,
                                   float* __restrict__ dpreds) {
    const long total4 = (long)N * HW / 4;
    const long NHW = (long)N * HW;
    // effective weights of the three base losses
    float w_prob, w_thr, w_bin;
    if (CH == 3) {
        w_prob = gout[0] + gout[3] + gout[4];
        w_thr = gout[1] + beta * (gout[3] + gout[4]);
        w_bin = gout[2] + alpha * gout[4];
    } else {
        w_prob = gout[4];
        w_thr = beta * gout[4];
        w_bin = 0.f;
    }
    const float cb = coef[0] * w_prob;
    const float ca = coef[1] * w_thr;
    const float U = coef[2], I = coef[3];
    const float cd = -2.f * w_bin / (U * U);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long p = i * 4;
        const long n = p / HW, o = p - n * HW;
        const float* pb = preds + n * CH * HW + o;
        float* db = dpreds + n * CH * HW + o;
        const f32x4 P = *reinterpret_cast<const f32x4*>(pb);
        const f32x4 T = *reinterpret_cast<const f32x4*>(pb + HW);
        const f32x4 G = *reinterpret_cast<const f32x4*>(gts + p);
        const f32x4 M = *reinterpret_cast<const f32x4*>(gts + NHW + p);
        const f32x4 Tg = *reinterpret_cast<const f32x4*>(gts + 2 * NHW + p);
        const f32x4 A = *reinterpret_cast<const f32x4*>(gts + 3 * NHW + p);
        f32x4 dP, dT, dB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // ATen binary_cross_entropy backward: (x - t) / max((1-x)*x, 1e-12)
            dP[e] = cb * (P[e] - G[e]) / fmaxf((1.f - P[e]) * P[e], 1e-12f);
            const float d = T[e] - Tg[e];
            dT[e] = ca * A[e] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            dB[e] = cd * M[e] * (G[e] * U - I);
        }
        *reinterpret_cast<f32x4*>(db) = dP;
        *reinterpret_cast<f32x4*>(db + HW) = dT;
        if (CH == 3) *reinterpret_cast<f32x4*>(db + 2 * HW) = dB;
    }
}

// 2x2 confusion matrix of the per-step pixel metric (text_metrics.py:14-24,63-82):
// pred = (P*M > thresh), gt = int(G*M); hist[2*gt + pred] += 1 over ALL pixels.
__global__ void pixel_confusion_kernel(const float* __restrict__ preds, long batch_stride, const float* __restrict__ gt,
                                       const float* __restrict__ mask, int N, long HW, float thresh, double* __restrict__ hist) {
    const long total = (long)N * HW;
    unsigned c01 = 0, c10 = 0, c11 = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, o = i - n * HW;
        const float m = mask[i];
        const int p = preds[n * batch_stride + o] * m > thresh;
        const int g = (int)(gt[i] * m) != 0;  // binary maps: int() truncation like .astype(np.int32)
        c01 += (!g) & p;
        c10 += g & (!p);
        c11 += g & p;
    }
    __shared__ unsigned red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned v[3] = {c01, c10, c11};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if (lane == 0) red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x];
        atomicAdd(&hist[1 + threadIdx.x], (double)t);  // integer-valued doubles: the sum is exact and order-independent
    }
    if (blockIdx.x == 0 && threadIdx.x == 3) atomicAdd(&hist[4], (double)total);  // hist[0] = total - others, formed by the reader
}

}  // namespace

extern "C" {

int dbn_head_tail_fwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* bias_b,
                      const float* bias_t, float* out, int N, int Hq, int Wq, int channels, float kstep, void* stream) {
    DBN_REQUIRE(xb && xt && wb && wt && bias_b && bias_t && out && (channels == 2 || channels == 3));
    const long npx = (long)N * Hq * Wq;
    hipLaunchKernelGGL(head_tail_fwd_kernel, dim3(dbn_grid(npx * 16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, xb, xt, wb,
                       wt, bias_b, bias_t, out, N, Hq, Wq, channels, kstep);
    return dbn_status();
}

int dbn_head_tail_bwd_ws_floats() { return 2048 * 2 * 257; }

// dw_b/dw_t: [64*4] (ConvTranspose2d weight grads), dbias_b/dbias_t: [1]
int dbn_head_tail_bwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* preds,
                      const float* dpreds, float* dxb, float* dxt, float* dw_b, float* dbias_b, float* dw_t, float* dbias_t,
                      int N, int Hq, int Wq, int channels, float kstep, float grad_scale, float* ws, void* stream) {
    DBN_REQUIRE(xb && xt && wb && wt && preds && dpreds && dxb && dxt && dw_b && dbias_b && dw_t && dbias_t && ws);
    DBN_REQUIRE(channels == 2 || channels == 3);
    hipStream_t st = (hipStream_t)stream;
    const long npx = (long)N * Hq * Wq;
    const int nb = dbn_grid(npx * 16, 256, 2047);
    hipLaunchKernelGGL(head_tail_bwd_kernel, dim3(nb), dim3(256), 0, st, xb, xt, wb, wt, preds, dpreds, dxb, dxt, ws, N, Hq, Wq,
                       channels, kstep);
    // fold partials: layout [dwb 256][dbias_b][dwt 256][dbias_t] -> staged in the tail of ws, then scattered by 4 tiny copies
    float* folded = ws + (long)2048 * 2 * 257 - 2 * 257;
    // nb <= 2047 partial rows may be used without touching the tail
    if (nb >= 2048) return DBN_ERR_ARG;
    hipLaunchKernelGGL(fold_partials_d_kernel, dim3(dbn_ceil_div(2 * 257, 8)), dim3(256), 0, st, ws, nb, 2 * 257, folded,
                       grad_scale);
    (void)hipMemcpyAsync(dw_b, folded, 256 * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(dbias_b, folded + 256, sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(dw_t, folded + 257, 256 * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(dbias_t, folded + 257 + 256, sizeof(float), hipMemcpyDeviceToDevice, st);
    return dbn_status();
}

int dbn_db_loss_ws_bytes() { return 1024 * NSUM * (int)sizeof(double); }

// losses: [5] floats, coef: [8] floats (kept for the backward), ws: dbn_db_loss_ws_bytes()
int dbn_db_loss_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                    float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream) {
    DBN_REQUIRE(preds && gts && losses && coef && ws && (channels == 2 || channels == 3));
    const long HW = (long)H * W;
    DBN_REQUIRE(HW % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    const int nb = dbn_grid((long)N * HW / 4, 256, 1024);
    hipLaunchKernelGGL(db_loss_fwd_kernel, dim3(nb), dim3(256), 0, st, preds, gts, N, HW, channels, (double*)ws);
    hipLaunchKernelGGL(db_loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, (long)N * HW, channels, alpha,
                       beta, negative_ratio, eps, losses, coef);
    return dbn_status();
}

int dbn_db_loss_bwd(const float* preds, const float* gts, const float* coef, const float* grad_losses, float alpha, float beta,
                    int N, int H, int W, int channels, float* dpreds, void* stream) {
    DBN_REQUIRE(preds && gts && coef && grad_losses && dpreds && (channels == 2 || channels == 3));
    const long HW = (long)H * W;
    DBN_REQUIRE(HW % 4 == 0);
    hipLaunchKernelGGL(db_loss_bwd_kernel, dim3(dbn_grid((long)N * HW / 4)), dim3(256), 0, (hipStream_t)stream, preds, gts, coef,
                       grad_losses, alpha, beta, N, HW, channels, dpreds);
    return dbn_status();
}

// hist: 5 doubles, ACCUMULATED into (running confusion matrix): [unused, n01, n10, n11, total]; n00 = total-n01-n10-n11.
// preds points at the probability plane of image 0; batch_stride = floats between consecutive images (C*H*W).
int dbn_pixel_confusion(const float* preds, long batch_stride, const float* gt, const float* mask, int N, int H, int W, float thresh,
                        double* hist, void* stream) {
    DBN_REQUIRE(preds && gt && mask && hist && N > 0 && H > 0 && W > 0);
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(pixel_confusion_kernel, dim3(dbn_grid(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, preds, batch_stride,
                       gt, mask, N, (long)H * W, thresh, hist);
    return dbn_status();
}

}  // extern "C"
