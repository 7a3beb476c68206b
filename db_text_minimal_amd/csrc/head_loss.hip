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
// v_exp_f32 + v_rcp_f32 (1 ulp each): |error| < 3e-7 on a value in (0, 1).  expf() and an IEEE division are ~60 VALU instructions
// per sigmoid; with three of them per output pixel (executed by the whole wave for its four active lanes per group) the
// head-tail forward kernel was VALU-bound, not HBM-bound (215 us for 917 MB; 175 us in bf16 storage with half the bytes).
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// 16 lanes cooperate on one quarter-resolution pixel: lane q owns channels 4q..4q+3.
// The 16-lane sums and broadcasts use DPP row operations (a DPP "row" IS 16 lanes): full-rate VALU instructions.  __shfl_xor
// compiles to ds_bpermute_b32, and at 32 of them per four pixels the forward kernel was bound by the LDS pipe, not by HBM
// (215 us for 917 MB = 4.3 TB/s).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);         // every lane of a quad: the quad's sum
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);   // lane i <-> 7-i: the other quad of the half
    v += dpp_f32<DPP_ROW_MIRROR>(v);        // lane i <-> 15-i: the other half
    return v;
}
// value of lane `k` (0..3) of this lane's quad
#ifndef DBN_HT_NODPP
#define DBN_HT_NODPP 0  // 1 (A/B build, tools/cotenancy_diff.py): the quad broadcasts of head_tail_bwd through ds_bpermute instead of DPP
#endif
#ifndef DBN_HT_CHECK
#define DBN_HT_CHECK 0  // 1 (A/B build): head_tail_bwd_kernel re-loads its loop-invariant operands at the end and counts the lanes whose registers differ
#endif
#ifndef DBN_HT_WAVES
#define DBN_HT_WAVES 0  // n > 0 (A/B build): head_tail_bwd_kernel compiled for n waves per SIMD (register budget 512 / n)
#endif
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
#if DBN_HT_NODPP
    return __shfl(v, (int)((threadIdx.x & 63u & ~3u) | K), 64);
#else
    return dpp_f32<K | (K << 2) | (K << 4) | (K << 6)>(v);
#endif
}

__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);   // lane i <-> 7-i of its 8-lane half row
    return v;
}

// xb/xt: [N,Hq,Wq,64] inputs of the last ConvT (already BN+ReLU'd);  wb/wt: [64][4]
// (ConvTranspose2d weight [64,1,2,2]); out: [N,CH,2Hq,2Wq], CH=3 (train) or 2 (eval).
// LPP lanes per pixel: 16 (a lane owns 4 channels: fp32 storage, one 16-byte load) or 8 (round 5, 16-bit storage: a lane owns 8 channels —
// again one 16-byte load instead of an 8-byte one, which reaches 0.55-0.7 of its rate: 1.01 ms for 3.4 GB at cfg5)
template <int AT, int LPP>
__global__ __launch_bounds__(256) void head_tail_fwd_kernel(const void* __restrict__ xb, const void* __restrict__ xt, const float* __restrict__ wb,
                                     const float* __restrict__ wt, const float* __restrict__ bias_b,
                                     const float* __restrict__ bias_t, const float* __restrict__ sc_b,
                                     const float* __restrict__ sh_b, const float* __restrict__ sc_t,
                                     const float* __restrict__ sh_t, float* __restrict__ out, int N, int Hq, int Wq, int CH,
                                     float kstep, long per) {
    static_assert((LPP == 16) || (LPP == 8 && AT != 0), "eight lanes per pixel: 16-bit storage");  // (LPP = 16 serves every storage type)
    constexpr int NQ = 16 / LPP;  // channel quads per lane
    const int q = threadIdx.x & (LPP - 1);
    // optional fused BatchNorm + ReLU of the inputs (xb/xt are then the pre-BN conv outputs): this lane's channels
    const bool bn = sc_b != nullptr;
    f32x4 scb[NQ], shb[NQ], sct[NQ], sht[NQ];
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
        scb[h] = sct[h] = f32x4{1.f, 1.f, 1.f, 1.f};
        shb[h] = sht[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bn) {
            scb[h] = *reinterpret_cast<const f32x4*>(sc_b + 4 * (NQ * q + h));
            shb[h] = *reinterpret_cast<const f32x4*>(sh_b + 4 * (NQ * q + h));
            sct[h] = *reinterpret_cast<const f32x4*>(sc_t + 4 * (NQ * q + h));
            sht[h] = *reinterpret_cast<const f32x4*>(sh_t + 4 * (NQ * q + h));
        }
    }
    const long npx = (long)N * Hq * Wq;
    // a block streams ONE contiguous run of `per` pixels (a multiple of 64), 256 / LPP pixels per trip, one per lane group.  Measured at
    // 16x320x320: 178 us with 128-pixel runs (12 800 blocks), 192-198 us with 4096-8192 blocks, 300 us with one trip per block;
    // the grid-stride sweep this replaces took 192-200 us
    const long gstride = blockDim.x / LPP;
    const long r1 = min(npx, (blockIdx.x + 1) * per);
    // weights of this lane's channels: w[ci][ab]
    f32x4 wbq[4 * NQ], wtq[4 * NQ];
#pragma unroll
    for (int e = 0; e < 4 * NQ; ++e) {
        wbq[e] = *reinterpret_cast<const f32x4*>(wb + (4 * NQ * q + e) * 4);
        wtq[e] = *reinterpret_cast<const f32x4*>(wt + (4 * NQ * q + e) * 4);
    }
    const float bb = bias_b[0], bt = bias_t[0];
    const int H = 2 * Hq, W = 2 * Wq;
    const long HW = (long)H * W;
    // (n, hq, wq) of the pixel: divided out once, then advanced by the stride with carries — a 64-bit division per pixel
    // (~100 VALU instructions, executed by the whole wave) had made this kernel instruction-bound at 4.3 TB/s
    const long px0 = blockIdx.x * per + (threadIdx.x / LPP);
    const long HWq = (long)Hq * Wq;
    long n = px0 / HWq;
    int hq = (int)((px0 - n * HWq) / Wq), wq = (int)((px0 - n * HWq) - (long)hq * Wq);
    const long g_n = gstride / HWq;
    const int g_h = (int)((gstride - g_n * HWq) / Wq), g_w = (int)((gstride - g_n * HWq) - (long)g_h * Wq);
    auto advance = [&]() {
        wq += g_w;
        if (wq >= Wq) { wq -= Wq; ++hq; }
        hq += g_h;
        if (hq >= Hq) { hq -= Hq; ++n; }
        n += g_n;
    };
    auto load = [&](const void* x, long px, f32x4 (&v)[NQ]) {
        if constexpr (NQ == 1) v[0] = dbn_ld4<AT>(x, px * 16 + q);
        else dbn_ldq<AT>(x, px * 8 + q, v);
    };
    auto finish = [&](f32x4 (&vb)[NQ], f32x4 (&vt)[NQ], long pn, int ph, int pw) {
        if (bn) {
#pragma unroll
            for (int h = 0; h < NQ; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vb[h][e] = dbn_affine_relu(vb[h][e], scb[h][e], shb[h][e]);
                    vt[h][e] = dbn_affine_relu(vt[h][e], sct[h][e], sht[h][e]);
                }
        }
        f32x4 sb = {0.f, 0.f, 0.f, 0.f}, stt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NQ; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sb += vb[h][e] * wbq[4 * h + e];
                stt += vt[h][e] * wtq[4 * h + e];
            }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            sb[ab] = LPP == 16 ? group16_sum(sb[ab]) : group8_sum(sb[ab]);
            stt[ab] = LPP == 16 ? group16_sum(stt[ab]) : group8_sum(stt[ab]);
        }
        if (q < 4) {
            const int ab = q;
            const float lp = (ab == 0 ? sb[0] : ab == 1 ? sb[1] : ab == 2 ? sb[2] : sb[3]) + bb;
            const float lt = (ab == 0 ? stt[0] : ab == 1 ? stt[1] : ab == 2 ? stt[2] : stt[3]) + bt;
            const float P = sigmoid_fast(lp), T = sigmoid_fast(lt);
            const long o = (long)(2 * ph + (ab >> 1)) * W + 2 * pw + (ab & 1);
            float* base = out + pn * CH * HW + o;
            base[0] = P;
            base[HW] = T;
            if (CH == 3) base[2 * HW] = sigmoid_fast(kstep * (P - T));
        }
    };
    // four pixels per trip, all eight loads issued before the first is used
    long px = px0;
    for (; px + 3 * gstride < r1; px += 4 * gstride) {
        f32x4 vb[4][NQ], vt[4][NQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load(xb, px + u * gstride, vb[u]);
            load(xt, px + u * gstride, vt[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            finish(vb[u], vt[u], n, hq, wq);
            advance();
        }
    }
    for (; px < r1; px += gstride, advance()) {
        f32x4 vb[NQ], vt[NQ];
        load(xb, px, vb);
        load(xt, px, vt);
        finish(vb, vt, n, hq, wq);
    }
}

// Backward.  For each quarter pixel: dl_b[ab], dl_t[ab] (grad wrt the two logits) from
// dpreds and the saved maps; dxb = sum_ab dl_b[ab]*wb[ci][ab]; dwb[ci][ab] += xb[ci]*dl_b[ab].
// part: [grid][2*(256+1)] block partials of (dwb[64*4], dbias_b, dwt[64*4], dbias_t).
#if DBN_HT_WAVES
#define DBN_HT_OCC __attribute__((amdgpu_waves_per_eu(DBN_HT_WAVES, DBN_HT_WAVES)))
#else
#define DBN_HT_OCC
#endif
template <int AT>
__global__ __launch_bounds__(256) DBN_HT_OCC void head_tail_bwd_kernel(const void* __restrict__ xb, const void* __restrict__ xt, const float* __restrict__ wb,
                                     const float* __restrict__ wt, const float* __restrict__ preds,
                                     const float* __restrict__ dpreds, const float* __restrict__ sc_b,
                                     const float* __restrict__ sh_b, const float* __restrict__ sc_t,
                                     const float* __restrict__ sh_t, const float* __restrict__ mean_b,
                                     const float* __restrict__ rstd_b, const float* __restrict__ mean_t,
                                     const float* __restrict__ rstd_t, void* __restrict__ dxb, void* __restrict__ dxt,
                                     float* __restrict__ part, int N, int Hq, int Wq, int CH, float kstep) {
    const int q = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const bool bn = sc_b != nullptr;  // xb/xt are pre-BN conv outputs: the activations are recomputed (same fmaf as the forward)
    f32x4 scb = {1.f, 1.f, 1.f, 1.f}, shb = {0.f, 0.f, 0.f, 0.f}, sct = scb, sht = shb;
    if (bn) {
        scb = *reinterpret_cast<const f32x4*>(sc_b + 4 * q);
        shb = *reinterpret_cast<const f32x4*>(sh_b + 4 * q);
        sct = *reinterpret_cast<const f32x4*>(sc_t + 4 * q);
        sht = *reinterpret_cast<const f32x4*>(sh_t + 4 * q);
    }
    const long npx = (long)N * Hq * Wq;
    const long gstride = (long)gridDim.x * (blockDim.x >> 4);
    f32x4 wbq[4], wtq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wbq[e] = *reinterpret_cast<const f32x4*>(wb + (4 * q + e) * 4);
        wtq[e] = *reinterpret_cast<const f32x4*>(wt + (4 * q + e) * 4);
    }
    // optional: the two per-channel reductions of the BatchNorm backward that consumes dxb/dxt (sum of the masked
    // gradient, sum of masked gradient * xhat), so that it does not have to re-read 2 x 2 x 420 MB to form them
    const bool bnsum = bn && mean_b != nullptr;
    f32x4 mub = {0.f, 0.f, 0.f, 0.f}, rsb = mub, mut = mub, rst = mub;
    if (bnsum) {
        mub = *reinterpret_cast<const f32x4*>(mean_b + 4 * q);
        rsb = *reinterpret_cast<const f32x4*>(rstd_b + 4 * q);
        mut = *reinterpret_cast<const f32x4*>(mean_t + 4 * q);
        rst = *reinterpret_cast<const f32x4*>(rstd_t + 4 * q);
    }
    f32x4 s1b = {0.f, 0.f, 0.f, 0.f}, s2b = s1b, s1t = s1b, s2t = s1b;
    const int H = 2 * Hq, W = 2 * Wq;
    const long HW = (long)H * W;
    f32x4 awb[4], awt[4];  // [e][ab]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        awb[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        awt[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float abb = 0.f, abt = 0.f;
    const long px0 = blockIdx.x * (long)(blockDim.x >> 4) + grp;
    const long HWq = (long)Hq * Wq;
    long n = px0 / HWq;  // pixel walk by the grid stride with carries (no division in the loop, see the forward kernel)
    int hq = (int)((px0 - n * HWq) / Wq), wq = (int)((px0 - n * HWq) - (long)hq * Wq);
    const long g_n = gstride / HWq;
    const int g_h = (int)((gstride - g_n * HWq) / Wq), g_w = (int)((gstride - g_n * HWq) - (long)g_h * Wq);
    auto advance = [&]() {
        wq += g_w;
        if (wq >= Wq) { wq -= Wq; ++hq; }
        hq += g_h;
        if (hq >= Hq) { hq -= Hq; ++n; }
        n += g_n;
    };
    for (long px = px0; px < npx; px += gstride, advance()) {
        // lane k of every quad evaluates the (a,b) = (k>>1, k&1) position (the four quads redundantly: same cache lines, same
        // instruction count), then a quad-local DPP broadcast hands all four to every lane — no cross-lane LDS traffic
        float dlb = 0.f, dlt = 0.f;
        {
            const int ab = q & 3;
            const long o = (long)(2 * hq + (ab >> 1)) * W + 2 * wq + (ab & 1);
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
        const f32x4 lb = {quad_bcast<0>(dlb), quad_bcast<1>(dlb), quad_bcast<2>(dlb), quad_bcast<3>(dlb)};
        const f32x4 lt = {quad_bcast<0>(dlt), quad_bcast<1>(dlt), quad_bcast<2>(dlt), quad_bcast<3>(dlt)};
        f32x4 vb = dbn_ld4<AT>(xb, px * 16 + q);
        f32x4 vt = dbn_ld4<AT>(xt, px * 16 + q);
        const f32x4 yb = vb, yt = vt;
        if (bn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vb[e] = dbn_affine_relu(vb[e], scb[e], shb[e]);
                vt[e] = dbn_affine_relu(vt[e], sct[e], sht[e]);
            }
        }
        f32x4 gb, gt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gb[e] = wbq[e][0] * lb[0] + wbq[e][1] * lb[1] + wbq[e][2] * lb[2] + wbq[e][3] * lb[3];
            gt[e] = wtq[e][0] * lt[0] + wtq[e][1] * lt[1] + wtq[e][2] * lt[2] + wtq[e][3] * lt[3];
            awb[e] += vb[e] * lb;
            awt[e] += vt[e] * lt;
        }
        dbn_st4<AT>(dxb, px * 16 + q, gb);
        dbn_st4<AT>(dxt, px * 16 + q, gt);
        if (bnsum) {
            if constexpr (AT != 0) {  // the BatchNorm backward's apply pass will read the ROUNDED gradient: sum exactly that
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gb[e] = AT == 1 ? (float)(__bf16)gb[e] : (float)(_Float16)gb[e];
                    gt[e] = AT == 1 ? (float)(__bf16)gt[e] : (float)(_Float16)gt[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mb = vb[e] > 0.f ? gb[e] : 0.f, mt = vt[e] > 0.f ? gt[e] : 0.f;  // ReLU mask as in bn_bwd ('self')
                s1b[e] += mb;
                s2b[e] += mb * ((yb[e] - mub[e]) * rsb[e]);
                s1t[e] += mt;
                s2t[e] += mt * ((yt[e] - mut[e]) * rst[e]);
            }
        }
        if (q == 0) {
            abb += lb[0] + lb[1] + lb[2] + lb[3];
            abt += lt[0] + lt[1] + lt[2] + lt[3];
        }
    }
#if DBN_HT_CHECK
    // A/B build (tools/cotenancy_diff.py): do the loop-invariant registers still hold what was loaded into them?  Counts, behind the partial
    // rows of `part` ([770 x 2047] used of 2048 x 770): [0] lanes whose wbq differs from a fresh load, [1] wtq, [2] scale / shift, [3] mean / rstd,
    // [4] lanes whose register copy differs in the LOW 16 bits only
    {
        float* dbg = part + (long)(2 * 257 + 4 * 64) * gridDim.x;
        int bad_w = 0, bad_t = 0, low_only = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 a = *reinterpret_cast<const volatile f32x4*>(wb + (4 * q + e) * 4), b = *reinterpret_cast<const volatile f32x4*>(wt + (4 * q + e) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned ua = __builtin_bit_cast(unsigned, a[k]), ur = __builtin_bit_cast(unsigned, wbq[e][k]);
                bad_w += ua != ur;
                low_only += (ua != ur) && ((ua >> 16) == (ur >> 16));
                bad_t += __builtin_bit_cast(unsigned, b[k]) != __builtin_bit_cast(unsigned, wtq[e][k]);
            }
        }
        int bad_s = 0;
        if (bn) {
            const f32x4 a = *reinterpret_cast<const volatile f32x4*>(sc_b + 4 * q), b = *reinterpret_cast<const volatile f32x4*>(sh_b + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) bad_s += (__builtin_bit_cast(unsigned, a[k]) != __builtin_bit_cast(unsigned, scb[k])) + (__builtin_bit_cast(unsigned, b[k]) != __builtin_bit_cast(unsigned, shb[k]));
        }
        if (bad_w) atomicAdd(dbg + 0, 1.f);
        if (bad_t) atomicAdd(dbg + 1, 1.f);
        if (bad_s) atomicAdd(dbg + 2, 1.f);
        if (low_only) atomicAdd(dbg + 4, 1.f);
    }
#endif
    // block reduction: groups with the same q hold the same channel slots
    constexpr int ROWS = 2 * 257 + 4 * 64;  // dwb 256, dbias_b, dwt 256, dbias_t, then the BN sums s1_b, s2_b, s1_t, s2_t [64] each
    __shared__ float red[16][ROWS];  // [grp][...]
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[514 + 4 * q + e] = s1b[e];
            r[514 + 64 + 4 * q + e] = s2b[e];
            r[514 + 128 + 4 * q + e] = s1t[e];
            r[514 + 192 + 4 * q + e] = s2t[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS; i += blockDim.x) {
        float s = 0.f;
        for (int g = 0; g < 16; ++g) s += red[g][i];
        part[(long)i * gridDim.x + blockIdx.x] = s;  // transposed: [ROWS][blocks]
    }
}

__global__ void fold_partials_d_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ out, float scale) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (i >= n) return;
    const double s = dbn_team32_fold(part, nb, i, l32);
    if (l32 == 0) out[i] = (float)(s * scale);
}

// partial rows [dwb 256][dbias_b][dwt 256][dbias_t] folded straight into their four destinations
__global__ void fold_head_grads_kernel(const float* __restrict__ part, int nb, float* __restrict__ dw_b, float* __restrict__ dbias_b,
                                       float* __restrict__ dw_t, float* __restrict__ dbias_t, float scale) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l32 = threadIdx.x & 31;
    if (i >= 2 * 257) return;
    const double s = dbn_team32_fold(part, nb, i, l32);
    if (l32 != 0) return;
    const int j = i < 257 ? i : i - 257;
    float* dst = i < 257 ? (j < 256 ? dw_b + j : dbias_b) : (j < 256 ? dw_t + j : dbias_t);
    *dst = (float)(s * scale);
}

// ----------------------------------------------------------------------------------
// loss
// ----------------------------------------------------------------------------------
enum { S_POS = 0, S_NEG, S_BCE, S_L1, S_A, S_BGM, S_BM, S_NONBIN /* pixels whose gt or mask is not 0 / 1 */, NSUM };

// loads/stores of V consecutive floats (V = 4: one 16-byte access when H*W % 4 == 0; V = 1: any size)
template <int V>
__device__ __forceinline__ f32x4 ldv(const float* p) {
    if constexpr (V == 4) return *reinterpret_cast<const f32x4*>(p);
    return f32x4{p[0], 0.f, 0.f, 0.f};
}
template <int V>
__device__ __forceinline__ void stv(float* p, f32x4 v) {
    if constexpr (V == 4) *reinterpret_cast<f32x4*>(p) = v;
    else p[0] = v[0];
}



struct DbLossFinal {
    // arguments of the finalize step, run by the LAST workgroup of db_loss_fwd_kernel
    long px;
    int CH;
    float alpha, beta, negative_ratio, eps;
    int bce_sum;
    float* losses;
    float* coef;
    unsigned* counter;  // arrival counter (zero on entry, left zero)
};

// losses[5] = prob, thresh, binary, prob+beta*thresh, total;  coef[8] for the backward:
//   0: c_bce = (sum_pos + n_neg)/(n_pos+n_neg+eps)/px   1: 1/(sum_A+eps)
//   2: dice U   3: dice I   4..6: per-pixel OHEM (ohem_finalize_kernel)
//   7: number of pixels whose prob_gt or supervision_mask is neither 0 nor 1 — for those the closed form below is NOT the
//      reference's value (losses.py:33-39 takes topk over loss * negative): the host side reads it and refuses such maps
//      unless the caller asked for the literal form (dbn_db_loss_frac_fwd)
__device__ __forceinline__ void db_loss_finish(const double (&s)[NSUM], const DbLossFinal& f) {
    // losses.py:25-28 — int() truncations
    const long n_pos = (long)(float)s[S_POS];
    const long n_neg_expect = (long)((double)n_pos * (double)f.negative_ratio);
    const long n_neg_cur = (long)(float)s[S_NEG];
    const long n_neg = n_neg_expect < n_neg_cur ? n_neg_expect : n_neg_cur;
    // reduction='mean' (default) / 'sum': F.binary_cross_entropy returns ONE scalar over all pixels (losses.py:30)
    const double bce_div = f.bce_sum ? 1.0 : (double)f.px;
    const float bce = (float)(s[S_BCE] / bce_div);
    // positive_loss.sum() = bce*sum_pos; topk(bce*negative, n_neg).sum() = bce*n_neg for binary maps
    const float denom = (float)((double)(n_pos + n_neg) + (double)f.eps);
    const float num_w = (float)s[S_POS] + (float)n_neg;
    const float prob = bce * num_w / denom;
    const float thr = (float)s[S_L1] / ((float)s[S_A] + f.eps);
    const float pt = prob + f.beta * thr;
    float* losses = f.losses;
    float* coef = f.coef;
    losses[0] = prob;
    losses[1] = thr;
    coef[0] = num_w / denom / (float)bce_div;
    coef[1] = 1.f / ((float)s[S_A] + f.eps);
    coef[7] = (float)s[S_NONBIN];
    if (f.CH == 3) {
        const float U = (float)s[S_BM] + (float)s[S_POS] + f.eps;
        const float I = (float)s[S_BGM];
        const float binl = 1.f - 2.f * I / U;
        losses[2] = binl;
        losses[3] = pt;
        losses[4] = f.alpha * binl + pt;
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

// The last workgroup (256 threads) folds every workgroup's row in a FIXED order and thread 0 finishes.  The partials were written
// by other workgroups of THIS launch: agent-scope loads (never a stale line of this CU's L1).  All of a thread's loads are issued
// before the first is used — a 32-lane team walking its column with one dependent load per trip took 30 us of a 50 us kernel.
__device__ __forceinline__ void db_loss_fold_and_finish(const double* __restrict__ part, int nb, const DbLossFinal& f, double* red /* [4][8] */) {
    constexpr int PER = 4;  // rows per thread of the first four waves: nb <= 1024 = 256 * PER
    double v[PER][NSUM];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = threadIdx.x + i * 256;
#pragma unroll
        for (int k = 0; k < NSUM; ++k)
            v[i][k] = (b < nb && threadIdx.x < 256) ? __hip_atomic_load(part + (long)b * NSUM + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // (red was read by the partial-row store above)
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        const double w = dbn_wave_sum_d((v[0][k] + v[1][k]) + (v[2][k] + v[3][k]));
        if (lane == 0 && wave < 4) red[wave * 8 + k] = w;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double s[NSUM];
    for (int k = 0; k < NSUM; ++k) s[k] = (red[k] + red[8 + k]) + (red[16 + k] + red[24 + k]);
    db_loss_finish(s, f);
}

// BCE term (t - 1) * max(log1p(-x), -100) - t * max(log(x), -100)  (ATen binary_cross_entropy; losses.py:30).
// The targets of the reference's loader are binary (data_loaders.py:158-165), so ONE logarithm per pixel suffices: of x where t = 1,
// of 1 - x where t = 0 — v_log_f32 (1 ulp) instead of two libm calls (~100 VALU instructions per pixel: the kernel had been as
// much VALU- as HBM-bound, 40 us for 183 MB).  log1p(u) is evaluated as log(w) * u / (w - 1) with w = fl(1 + u) (the rounding
// error of w cancels in the quotient; w == 1 -> u).  A pixel with a fractional target takes the two-logarithm form.
__device__ __forceinline__ float bce_term(float x, float t) {
    if (t == 1.f || t == 0.f) {
        const bool pos = t == 1.f;
        const float u = -x, w = pos ? x : 1.f + u;
        float l = __logf(w);
        if (!pos) l = (w == 1.f) ? u : l * (u * __builtin_amdgcn_rcpf(w - 1.f));
        return -fmaxf(l, -100.f);
    }
    const float lp = fmaxf(logf(x), -100.f);
    const float lq = fmaxf(log1pf(-x), -100.f);
    return (t - 1.f) * lq - t * lp;
}

template <int V>
__global__ void db_loss_fwd_kernel(const float* __restrict__ preds, const float* __restrict__ gts, int N, long HW, int CH,
                                   double* __restrict__ part, DbLossFinal fin) {
    const long total4 = (long)N * HW / V;
    const long NHW = (long)N * HW;
    float s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = 0.f;
    double ds[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) ds[k] = 0.0;
    int cnt = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long p = i * V;
        const long n = p / HW, o = p - n * HW;
        const float* pb = preds + n * CH * HW + o;
        const f32x4 P = ldv<V>(pb);
        const f32x4 T = ldv<V>(pb + HW);
        f32x4 B = {0.f, 0.f, 0.f, 0.f};
        if (CH == 3) B = ldv<V>(pb + 2 * HW);
        const f32x4 G = ldv<V>(gts + p);
        const f32x4 M = ldv<V>(gts + NHW + p);
        const f32x4 Tg = ldv<V>(gts + 2 * NHW + p);
        const f32x4 A = ldv<V>(gts + 3 * NHW + p);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            s[S_POS] += G[e] * M[e];
            s[S_NEG] += (1.f - G[e]) * M[e];
            s[S_BCE] += bce_term(P[e], G[e]);
            s[S_L1] += fabsf(T[e] - Tg[e]) * A[e];
            s[S_A] += A[e];
            s[S_BGM] += B[e] * G[e] * M[e];
            s[S_BM] += B[e] * M[e];
            s[S_NONBIN] += (G[e] * (1.f - G[e]) != 0.f || M[e] * (1.f - M[e]) != 0.f) ? 1.f : 0.f;
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
    __shared__ double red[8][8];  // (up to eight waves)
    __shared__ int s_last;
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
        double* dst = part + (long)blockIdx.x * NSUM + threadIdx.x;
        __hip_atomic_store(dst, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read by the last workgroup: memory-side store
    }
    // In-kernel finalize: the workgroup that arrives last folds every workgroup's row (fixed order) and writes the five losses and
    // the backward's coefficients — the separate one-block finalize launch (10 us of a 50 us bracket) is gone.  Hand-over as in
    // igemm_kernel.h bnb_finish: 8-byte agent-scope (sc1) stores drained by s_waitcnt before the agent-scope counter goes up,
    // agent-scope loads on the reading side; the counter is left at zero for the next call.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DBN_RACE_JITTER();
    if (threadIdx.x == 0) {
        const int last = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (last) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    db_loss_fold_and_finish(part, (int)gridDim.x, fin, &red[0][0]);
}

// gout[5]: upstream grads of the 5 returned losses (device); dpreds planes like preds.
template <int V>
__global__ void db_loss_bwd_kernel(const float* __restrict__ preds, const float* __restrict__ gts, const float* __restrict__ coef,
                                   const float* __restrict__ gout, const float* __restrict__ ohem_v, float alpha, float beta, int N,
                                   long HW, int CH, float* __restrict__ dpreds) {
    const long total4 = (long)N * HW / V;
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
    const bool per_pixel = ohem_v != nullptr;
    const float tau = per_pixel ? coef[4] : 0.f, tie_w = per_pixel ? coef[5] : 0.f;
    const float ca = coef[1] * w_thr;
    const float U = coef[2], I = coef[3];
    const float cd = -2.f * w_bin / (U * U);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long p = i * V;
        const long n = p / HW, o = p - n * HW;
        const float* pb = preds + n * CH * HW + o;
        float* db = dpreds + n * CH * HW + o;
        const f32x4 P = ldv<V>(pb);
        const f32x4 T = ldv<V>(pb + HW);
        const f32x4 G = ldv<V>(gts + p);
        const f32x4 M = ldv<V>(gts + NHW + p);
        const f32x4 Tg = ldv<V>(gts + 2 * NHW + p);
        const f32x4 A = ldv<V>(gts + 3 * NHW + p);
        f32x4 dP = {0.f, 0.f, 0.f, 0.f}, dT = dP, dB = dP;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            // ATen binary_cross_entropy backward: (x - t) / max((1-x)*x, 1e-12)
            float wsel = 1.f;  // 'mean': the scalar BCE weights every pixel alike
            if (per_pixel) {     // 'none': positives + the selected (top n_neg) negatives
                const float vv = ohem_v[p + e];
                const float sel = vv > tau ? 1.f : (vv == tau ? tie_w : 0.f);
                wsel = G[e] * M[e] + (1.f - G[e]) * M[e] * sel;
            }
            dP[e] = cb * wsel * (P[e] - G[e]) / fmaxf((1.f - P[e]) * P[e], 1e-12f);
            const float d = T[e] - Tg[e];
            dT[e] = ca * A[e] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            dB[e] = cd * M[e] * (G[e] * U - I);
        }
        stv<V>(db, dP);
        stv<V>(db + HW, dT);
        if (CH == 3) stv<V>(db + 2 * HW, dB);
    }
}

// ----------------------------------------------------------------------------------
// true per-pixel OHEM (DBLoss(reduction='none'), losses.py:30-39): device radix select of the
// n_neg largest negative losses.  State (uint64 words): [0] k, [1] prefix bits, [2] count above the
// prefix so far, [3] tau bits, [4] cnt_eq;  hist: 3 x 2048 uint32.
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float bce_px(float P, float G) {
    const float lp = fmaxf(logf(P), -100.f);
    const float lq = fmaxf(log1pf(-P), -100.f);
    return (G - 1.f) * lq - G * lp;
}

constexpr int OHEM_BINS = 2048;
__device__ __forceinline__ unsigned ohem_digit(unsigned bits, int pass) {
    return pass == 0 ? bits >> 21 : pass == 1 ? (bits >> 10) & 2047u : bits & 1023u;
}
__device__ __forceinline__ bool ohem_prefix_match(unsigned bits, unsigned prefix, int pass) {
    return pass == 0 ? true : pass == 1 ? (bits >> 21) == (prefix >> 21) : (bits >> 10) == (prefix >> 10);
}

// k = n_neg from the folded sums; clears the select state.  One block.
__global__ void ohem_prepare_kernel(const double* __restrict__ part, int nb, float negative_ratio, unsigned long long* __restrict__ st,
                                    unsigned* __restrict__ hist) {
    __shared__ double sums[2];
    if (threadIdx.x < 64) {
        const int k = threadIdx.x >> 5, l32 = threadIdx.x & 31;
        double t = 0.0;
        for (int b = l32; b < nb; b += 32) t += part[(long)b * NSUM + (k == 0 ? S_POS : S_NEG)];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (l32 == 0) sums[k] = t;
    }
    for (int i = threadIdx.x; i < 3 * OHEM_BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long n_pos = (long)(float)sums[0];
        const long n_exp = (long)((double)n_pos * (double)negative_ratio);
        const long n_cur = (long)(float)sums[1];
        st[0] = (unsigned long long)(n_exp < n_cur ? n_exp : n_cur);
        st[1] = 0;
        st[2] = 0;
        st[3] = 0xFFFFFFFFull;  // tau bits: "select nothing" until the scans say otherwise
        st[4] = 0;
    }
}

// v[i] = bce_i * negative_i; partial sums of bce_i * positive_i (double, [block]).
// frac (the scalar-BCE reductions on non-binary maps, dbn_db_loss_frac_fwd): v[i] = negative_i alone — the scalar bce >= 0 scales
// every element alike, so topk(bce * negative, k).sum() = bce * (sum of the k largest negative_i).
__global__ void ohem_values_kernel(const float* __restrict__ preds, const float* __restrict__ gts, int N, long HW, int CH,
                                   float* __restrict__ v, double* __restrict__ part, int frac) {
    const long total = (long)N * HW, NHW = total;
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, o = i - n * HW;
        const float G = gts[i], M = gts[NHW + i];
        const float l = frac ? 1.f : bce_px(preds[n * CH * HW + o], G);
        v[i] = fabsf(l * ((1.f - G) * M));  // +0 (never -0): the select orders values by their bit pattern
        acc += (double)(l * (G * M));
    }
    __shared__ double red[4];
    const double w = dbn_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
        part[blockIdx.x] = t;
    }
}

__global__ void ohem_hist_kernel(const float* __restrict__ v, long total, const unsigned long long* __restrict__ st, int pass,
                                 unsigned* __restrict__ hist) {
    __shared__ unsigned lh[OHEM_BINS];
    for (int i = threadIdx.x; i < OHEM_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const unsigned prefix = (unsigned)st[1];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const unsigned bits = __builtin_bit_cast(unsigned, v[i]);  // v >= 0: the bit pattern orders like the value
        if (ohem_prefix_match(bits, prefix, pass)) atomicAdd(&lh[ohem_digit(bits, pass)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < OHEM_BINS; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[pass * OHEM_BINS + i], lh[i]);
}

// walk the bins from the top until the k-th largest value's bin; extend the prefix.  One thread.
__global__ void ohem_scan_kernel(unsigned long long* __restrict__ st, const unsigned* __restrict__ hist, int pass) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long k = st[0];
    if (k == 0) return;
    unsigned long long above = st[2];
    const int nb = pass == 2 ? 1024 : OHEM_BINS;
    int b = nb - 1;
    for (; b > 0; --b) {
        const unsigned c = hist[pass * OHEM_BINS + b];
        if (above + c >= k) break;
        above += c;
    }
    st[2] = above;
    const unsigned long long shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
    st[1] |= (unsigned long long)b << shift;
    if (pass == 2) {
        st[3] = st[1];                        // tau bits
        st[4] = hist[2 * OHEM_BINS + b];      // elements equal to tau
    }
}

// partial[block] = {sum of v > tau, count of v > tau}
__global__ void ohem_sum_kernel(const float* __restrict__ v, long total, const unsigned long long* __restrict__ st,
                                double* __restrict__ part) {
    const unsigned tau = (unsigned)st[3];
    double acc = 0.0, cnt = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float x = v[i];
        if (__builtin_bit_cast(unsigned, x) > tau) {
            acc += (double)x;
            cnt += 1.0;
        }
    }
    __shared__ double red[4][2];
    const double w0 = dbn_wave_sum_d(acc), w1 = dbn_wave_sum_d(cnt);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = w0;
        red[threadIdx.x >> 6][1] = w1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k][threadIdx.x];
        part[2 * blockIdx.x + threadIdx.x] = t;
    }
}

// prob_loss of the per-pixel OHEM; overwrites losses[0], [3], [4] and coef[0], coef[4..6] left by the 'mean' finalize.
__global__ void ohem_finalize_kernel(const double* __restrict__ part_sums, int nb_sums, const double* __restrict__ part_pos, int nb_pos,
                                     const double* __restrict__ part_sel, int nb_sel, const unsigned long long* __restrict__ st,
                                     int CH, float alpha, float beta, float eps, float* __restrict__ losses, float* __restrict__ coef,
                                     int frac, double bce_div) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s_pos = 0.0, pos_loss = 0.0, sel = 0.0, cnt_gt = 0.0, s_bce = 0.0;
    for (int b = 0; b < nb_sums; ++b) s_pos += part_sums[(long)b * NSUM + S_POS];
    if (frac)
        for (int b = 0; b < nb_sums; ++b) s_bce += part_sums[(long)b * NSUM + S_BCE];
    for (int b = 0; b < nb_pos; ++b) pos_loss += part_pos[b];
    for (int b = 0; b < nb_sel; ++b) {
        sel += part_sel[2 * b];
        cnt_gt += part_sel[2 * b + 1];
    }
    const double k = (double)st[0];
    const float tau = k > 0 ? __builtin_bit_cast(float, (unsigned)st[3]) : 0.f;
    const double n_tie = k - cnt_gt;  // taken from the elements equal to tau
    const double cnt_eq = (double)st[4];
    const long n_pos = (long)(float)s_pos;
    const float denom = (float)((double)n_pos + k + (double)eps);
    float prob = (float)(pos_loss + sel + n_tie * (double)tau) / denom;
    if (frac) {
        // losses.py:30-39 with a scalar bce: (bce * positive).sum() + topk(bce * negative, k).sum() = bce * (sum_pos + top-k sum);
        // pos_loss holds sum_pos here (l = 1 in ohem_values_kernel).  The backward stays the scalar-BCE one: coef[0] only.
        const float bce = (float)(s_bce / bce_div);
        const float num_w = (float)(pos_loss + sel + n_tie * (double)tau);
        prob = bce * num_w / denom;
        const float thr_f = losses[1], pt_f = prob + beta * thr_f;
        losses[0] = prob;
        losses[3] = pt_f;
        losses[4] = (CH == 3 ? alpha * losses[2] : 0.f) + pt_f;
        coef[0] = num_w / denom / (float)bce_div;
        coef[7] = 0.f;  // the maps were evaluated literally: nothing to refuse
        return;
    }
    const float thr = losses[1];
    const float pt = prob + beta * thr;
    losses[0] = prob;
    losses[3] = pt;
    losses[4] = (CH == 3 ? alpha * losses[2] : 0.f) + pt;
    coef[0] = 1.f / denom;
    coef[4] = k > 0 ? tau : 3.0e38f;  // nothing is above FLT_MAX-ish -> no negatives selected
    coef[5] = (k > 0 && cnt_eq > 0) ? (float)(n_tie / cnt_eq) : 0.f;
    coef[6] = 1.f;  // per-pixel mode
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

// xb/xt are stored in the activation type `at`; the maps `out` are always fp32 (what DBLoss and postprocess.py consume)
static int g_head_tail_wide = 0;  // 0: by size (see dbn_head_tail_fwd_t); 1 / -1: always / never the eight-lane form on 16-bit storage (test hook)
int dbn_set_head_tail_wide(int mode) {
    const int old = g_head_tail_wide;
    g_head_tail_wide = mode > 0 ? 1 : mode < 0 ? -1 : 0;
    return old;
}
int dbn_head_tail_fwd_t(int at, const void* xb, const void* xt, const float* wb, const float* wt, const float* bias_b,
                        const float* bias_t, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                        const float* bn_shift_t, float* out, int N, int Hq, int Wq, int channels, float kstep, void* stream) {
    DBN_REQUIRE(xb && xt && wb && wt && bias_b && bias_t && out && (channels == 2 || channels == 3));
    DBN_REQUIRE((bn_scale_b && bn_shift_b && bn_scale_t && bn_shift_t) || (!bn_scale_b && !bn_shift_b && !bn_scale_t && !bn_shift_t));
    const long npx = (long)N * Hq * Wq;
    const long per = ((npx + 16383) / 16384 + 63) / 64 * 64;  // pixels per block
    const dim3 grid((unsigned)((npx + per - 1) / per));
    // 16-bit storage: eight lanes x eight channels per pixel (16-byte loads) on LARGE maps only — measured: 32 x 640^2 quarter pixels (cfg5)
    // 1015 -> 952 us, but 16 x 320^2 (the bf16 train step) 132 -> 155 us (twice the registers, short runs per block)
    const bool wide = (at == 1 || at == 2) && (g_head_tail_wide > 0 || (g_head_tail_wide == 0 && npx >= (1L << 22)));
    if (wide && at == 1)
        hipLaunchKernelGGL((head_tail_fwd_kernel<1, 8>), grid, dim3(256), 0, (hipStream_t)stream, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b,
                           bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq, channels, kstep, per);
    else if (wide && at == 2)
        hipLaunchKernelGGL((head_tail_fwd_kernel<2, 8>), grid, dim3(256), 0, (hipStream_t)stream, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b,
                           bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq, channels, kstep, per);
    else if (at == 1)
        hipLaunchKernelGGL((head_tail_fwd_kernel<1, 16>), grid, dim3(256), 0, (hipStream_t)stream, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b,
                           bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq, channels, kstep, per);
    else if (at == 2)
        hipLaunchKernelGGL((head_tail_fwd_kernel<2, 16>), grid, dim3(256), 0, (hipStream_t)stream, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b,
                           bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq, channels, kstep, per);
    else if (at == 0)
        hipLaunchKernelGGL((head_tail_fwd_kernel<0, 16>), grid, dim3(256), 0, (hipStream_t)stream, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b,
                           bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq, channels, kstep, per);
    else
        return DBN_ERR_ARG;
    return dbn_status();
}
int dbn_head_tail_fwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* bias_b,
                      const float* bias_t, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                      const float* bn_shift_t, float* out, int N, int Hq, int Wq, int channels, float kstep, void* stream) {
    return dbn_head_tail_fwd_t(0, xb, xt, wb, wt, bias_b, bias_t, bn_scale_b, bn_shift_b, bn_scale_t, bn_shift_t, out, N, Hq, Wq,
                               channels, kstep, stream);
}

int dbn_head_tail_bwd_ws_floats() { return 2048 * (2 * 257 + 4 * 64); }

// dw_b/dw_t: [64*4] (ConvTranspose2d weight grads), dbias_b/dbias_t: [1]
int dbn_head_tail_bwd_t(int at, const void* xb, const void* xt, const float* wb, const float* wt, const float* preds,
                        const float* dpreds, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                        const float* bn_shift_t, const float* bn_mean_b, const float* bn_rstd_b, const float* bn_mean_t,
                        const float* bn_rstd_t, float* bn_sums, void* dxb, void* dxt, float* dw_b, float* dbias_b, float* dw_t,
                        float* dbias_t, int N, int Hq, int Wq, int channels, float kstep, float grad_scale, float* ws, void* stream) {
    DBN_REQUIRE(xb && xt && wb && wt && preds && dpreds && dxb && dxt && dw_b && dbias_b && dw_t && dbias_t && ws);
    DBN_REQUIRE((bn_scale_b && bn_shift_b && bn_scale_t && bn_shift_t) || (!bn_scale_b && !bn_shift_b && !bn_scale_t && !bn_shift_t));
    DBN_REQUIRE(channels == 2 || channels == 3);
    const bool sums = bn_sums != nullptr;
    DBN_REQUIRE(!sums || (bn_scale_b && bn_mean_b && bn_rstd_b && bn_mean_t && bn_rstd_t));
    hipStream_t st = (hipStream_t)stream;
    const long npx = (long)N * Hq * Wq;
    const int nb = dbn_grid(npx * 16, 256, 2047);  // < 2048 partial rows: fits dbn_head_tail_bwd_ws_floats()
    DBN_DISPATCH_AT(at, hipLaunchKernelGGL(head_tail_bwd_kernel<AT>, dim3(nb), dim3(256), 0, st, xb, xt, wb, wt, preds, dpreds, bn_scale_b,
                                           bn_shift_b, bn_scale_t, bn_shift_t, sums ? bn_mean_b : nullptr, bn_rstd_b, bn_mean_t,
                                           bn_rstd_t, dxb, dxt, ws, N, Hq, Wq, channels, kstep));
    hipLaunchKernelGGL(fold_head_grads_kernel, dim3(dbn_ceil_div(2 * 257, 8)), dim3(256), 0, st, ws, nb, dw_b, dbias_b, dw_t, dbias_t,
                       grad_scale);
    if (sums)  // [s1_b | s2_b | s1_t | s2_t]: two [2][64] blocks for dbn_bn_backward_from_sums
        hipLaunchKernelGGL(fold_partials_d_kernel, dim3(dbn_ceil_div(256, 8)), dim3(256), 0, st, ws + (long)514 * nb, nb, 256, bn_sums,
                           1.0f);
    return dbn_status();
}

int dbn_head_tail_bwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* preds,
                      const float* dpreds, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                      const float* bn_shift_t, const float* bn_mean_b, const float* bn_rstd_b, const float* bn_mean_t,
                      const float* bn_rstd_t, float* bn_sums, float* dxb, float* dxt, float* dw_b, float* dbias_b, float* dw_t,
                      float* dbias_t, int N, int Hq, int Wq, int channels, float kstep, float grad_scale, float* ws, void* stream) {
    return dbn_head_tail_bwd_t(0, xb, xt, wb, wt, preds, dpreds, bn_scale_b, bn_shift_b, bn_scale_t, bn_shift_t, bn_mean_b, bn_rstd_b,
                               bn_mean_t, bn_rstd_t, bn_sums, dxb, dxt, dw_b, dbias_b, dw_t, dbias_t, N, Hq, Wq, channels, kstep,
                               grad_scale, ws, stream);
}

// [1024][NSUM] partial sums (doubles) + the arrival counter of the in-kernel finalize.  The caller's workspace needs NO
// initialisation: the counter is cleared on the stream in front of every launch (a 4-byte hipMemsetAsync, graph-capturable), so an
// uninitialised workspace, one shared with another call, or a counter left non-zero by an aborted launch cannot keep a later call
// from finding its last workgroup (round-4 advisor finding: the old contract was "zero before the first call" and failed silently).
static const long DB_LOSS_PART_BYTES = 1024L * NSUM * 8;
int dbn_db_loss_ws_bytes() { return (int)DB_LOSS_PART_BYTES + 64; }

// workspace of the per-pixel OHEM path: sums partials | pos partials | select partials | state | hist | v[N*H*W]
static const long OHEM_OFF_POS = DB_LOSS_PART_BYTES + 64, OHEM_OFF_SEL = OHEM_OFF_POS + 1024L * 8, OHEM_OFF_ST = OHEM_OFF_SEL + 2048L * 8,
                  OHEM_OFF_HIST = OHEM_OFF_ST + 64, OHEM_OFF_V = OHEM_OFF_HIST + 3L * OHEM_BINS * 4;
long dbn_db_loss_ohem_ws_bytes(int N, int H, int W) { return OHEM_OFF_V + (long)N * H * W * 4; }

static int db_loss_fwd_run(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                           float negative_ratio, float eps, int per_pixel, float* losses, float* coef, void* ws, void* stream) {
    DBN_REQUIRE(preds && gts && losses && coef && ws && (channels == 2 || channels == 3));
    const long HW = (long)H * W;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = HW % 4 == 0;
    // 512-thread workgroups (eight waves per SIMD at 1024 workgroups: twice the loads in flight of the 256-thread form)
    const int nb = dbn_grid((long)N * HW / (vec ? 4 : 1), 512, 1024);
    char* base = (char*)ws;
    // per_pixel: 0 'mean', 1 'none' (per-pixel OHEM), 2 'sum', 3 / 4 'mean' / 'sum' on non-binary maps (literal top-k of negative)
    const int frac = per_pixel >= 3, bce_sum = (per_pixel == 2 || per_pixel == 4) ? 1 : 0;
    DbLossFinal fin = {(long)N * HW, channels, alpha, beta, negative_ratio, eps, bce_sum, losses, coef,
                       (unsigned*)(base + DB_LOSS_PART_BYTES)};
    if (hipMemsetAsync(fin.counter, 0, sizeof(unsigned), st) != hipSuccess) return dbn_status();
    if (vec)
        hipLaunchKernelGGL(db_loss_fwd_kernel<4>, dim3(nb), dim3(512), 0, st, preds, gts, N, HW, channels, (double*)ws, fin);
    else
        hipLaunchKernelGGL(db_loss_fwd_kernel<1>, dim3(nb), dim3(512), 0, st, preds, gts, N, HW, channels, (double*)ws, fin);
    if (per_pixel != 1 && !frac) return dbn_status();
    double* part_pos = (double*)(base + OHEM_OFF_POS);
    double* part_sel = (double*)(base + OHEM_OFF_SEL);
    unsigned long long* state = (unsigned long long*)(base + OHEM_OFF_ST);
    unsigned* hist = (unsigned*)(base + OHEM_OFF_HIST);
    float* v = (float*)(base + OHEM_OFF_V);
    const long total = (long)N * HW;
    const int nbv = dbn_grid(total, 256, 1024);
    hipLaunchKernelGGL(ohem_prepare_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, negative_ratio, state, hist);
    hipLaunchKernelGGL(ohem_values_kernel, dim3(nbv), dim3(256), 0, st, preds, gts, N, HW, channels, v, part_pos, frac);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(ohem_hist_kernel, dim3(nbv), dim3(256), 0, st, v, total, state, pass, hist);
        hipLaunchKernelGGL(ohem_scan_kernel, dim3(1), dim3(64), 0, st, state, hist, pass);
    }
    hipLaunchKernelGGL(ohem_sum_kernel, dim3(nbv), dim3(256), 0, st, v, total, state, part_sel);
    hipLaunchKernelGGL(ohem_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nb, part_pos, nbv, part_sel, nbv, state,
                       channels, alpha, beta, eps, losses, coef, frac, bce_sum ? 1.0 : (double)((long)N * HW));
    return dbn_status();
}

// losses: [5] floats, coef: [8] floats (kept for the backward), ws: dbn_db_loss_ws_bytes()
int dbn_db_loss_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                    float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream) {
    return db_loss_fwd_run(preds, gts, N, H, W, channels, alpha, beta, negative_ratio, eps, 0, losses, coef, ws, stream);
}

// DBLoss(reduction='sum'): like dbn_db_loss_fwd with the scalar BCE summed instead of averaged (losses.py:30 forwards the
// reduction string to F.binary_cross_entropy); backward through dbn_db_loss_bwd with the coef written here.
int dbn_db_loss_sum_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                        float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream) {
    return db_loss_fwd_run(preds, gts, N, H, W, channels, alpha, beta, negative_ratio, eps, 2, losses, coef, ws, stream);
}

// DBLoss(reduction='none'): true per-pixel OHEM (top n_neg negative losses, device radix select).
// ws: dbn_db_loss_ohem_ws_bytes(N,H,W) bytes, must stay untouched until dbn_db_loss_ohem_bwd has run.
int dbn_db_loss_ohem_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                         float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream) {
    return db_loss_fwd_run(preds, gts, N, H, W, channels, alpha, beta, negative_ratio, eps, 1, losses, coef, ws, stream);
}

// DBLoss(reduction='mean' | 'sum') on NON-BINARY prob_gt / supervision_mask maps, literally as losses.py:33-39 evaluates it:
// bce * (sum(positive) + sum of the n_neg largest negative_i) / (n_pos + n_neg + eps), the top-k sum by the device radix select of
// the per-pixel OHEM path.  (For binary maps it equals dbn_db_loss_fwd, which needs no select.)  `sum` != 0: reduction='sum'.
// ws: dbn_db_loss_ohem_ws_bytes(N,H,W) bytes.  Backward: dbn_db_loss_bwd with the coef written here.
int dbn_db_loss_frac_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                         float negative_ratio, float eps, int sum, float* losses, float* coef, void* ws, void* stream) {
    return db_loss_fwd_run(preds, gts, N, H, W, channels, alpha, beta, negative_ratio, eps, sum ? 4 : 3, losses, coef, ws, stream);
}

static int db_loss_bwd_run(const float* preds, const float* gts, const float* coef, const float* grad_losses, const float* ohem_v,
                           float alpha, float beta, int N, int H, int W, int channels, float* dpreds, void* stream) {
    DBN_REQUIRE(preds && gts && coef && grad_losses && dpreds && (channels == 2 || channels == 3));
    const long HW = (long)H * W;
    if (HW % 4 == 0)
        hipLaunchKernelGGL(db_loss_bwd_kernel<4>, dim3(dbn_grid((long)N * HW / 4)), dim3(256), 0, (hipStream_t)stream, preds, gts,
                           coef, grad_losses, ohem_v, alpha, beta, N, HW, channels, dpreds);
    else
        hipLaunchKernelGGL(db_loss_bwd_kernel<1>, dim3(dbn_grid((long)N * HW)), dim3(256), 0, (hipStream_t)stream, preds, gts, coef,
                           grad_losses, ohem_v, alpha, beta, N, HW, channels, dpreds);
    return dbn_status();
}

int dbn_db_loss_bwd(const float* preds, const float* gts, const float* coef, const float* grad_losses, float alpha, float beta,
                    int N, int H, int W, int channels, float* dpreds, void* stream) {
    return db_loss_bwd_run(preds, gts, coef, grad_losses, nullptr, alpha, beta, N, H, W, channels, dpreds, stream);
}

int dbn_db_loss_ohem_bwd(const float* preds, const float* gts, const float* coef, const float* grad_losses, const void* ws,
                         float alpha, float beta, int N, int H, int W, int channels, float* dpreds, void* stream) {
    DBN_REQUIRE(ws);
    return db_loss_bwd_run(preds, gts, coef, grad_losses, (const float*)((const char*)ws + OHEM_OFF_V), alpha, beta, N, H, W, channels,
                           dpreds, stream);
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
