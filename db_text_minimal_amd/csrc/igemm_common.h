// Shared definitions of the implicit-GEMM translation units (conv_*.hip, conv.hip, wgrad*.hip, pack.hip).
//
//
//   dbn_igemm_f32 / dbn_igemm_bf16s   forward conv (gather mode 0) and data-gradient / transposed conv
//                                     (gather mode 1; stride 2 as four output-parity problems) over NHWC
//   dbn_wgrad_f32 / dbn_wgrad_bf16s   weight gradient: split-K over output pixels into fp32 slabs
//                                     + deterministic slab reduction scattered into OIHW gradients
//   dbn_pack_weights[_bf16s]          OIHW -> GEMM panels
//
// Replaces the ATen convolution calls under /root/reference/src/modules/resnet.py:70-91,231-242,
// modules/basic.py:32-36, modules/segmentation_body.py:64-77 and modules/segmentation_head.py:24-29,64-79
// (Conv2d / ConvTranspose2d forward and their autograd backward).
//
// Matrix instruction (template parameter NS):
//   NS = 0  v_mfma_f32_32x32x2_f32 — exact fp32 products, 64 FLOP/clk/SIMD, 157 TFLOP/s chip peak (default)
//   NS = 3  v_mfma_f32_32x32x16_bf16 on an exact three-way bf16 split of every fp32 operand, six partial
//           products, fp32 accumulate: fp32-accurate at 6/16 of the fp32-MFMA cost
//   NS = 1  the same with operands rounded to bf16 (BASELINE configs[2] compute mode)
//
// Tiling: a workgroup of 4 waves owns a BM x BN output tile, each wave a (BM/WM) x (BN/WN)
// sub-tile of 32x32 f32 accumulators.  K is walked in steps of 16; the A panel (im2col gather, 16 B per lane
// = 4 consecutive input channels of one tap, branch-free buffer loads that return 0 out of range) and the B
// panel (pre-packed weights) are staged through a double-buffered LDS image laid out [k/4][row][4 f32]
// (NS = 0) or [split][k/8][row][8 bf16] (NS > 0) so that every lane fetches the k-values of its MFMAs with
// conflict-free ds_read_b128.
#pragma once
#include "common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include <algorithm>

// Geometry of one GEMM problem of a launch.  A stride-f data gradient / transposed conv (f = 2, 4, 8) is split into
// f*f output-parity classes — each a dense stride-1 transposed conv over 1/f^2 of the output pixels with only the
// taps that reach it — so no zero taps are multiplied.  Classes are derived from the class index by class_geom()
// on both host and device; the launch carries only per-class tile ranges and weight-panel offsets.
struct IgemmClass {
    int Hd, Wd;        // output sub-grid of this problem
    int M, K, KT;      // rows, reduction length, k-tiles
    int R, S;          // taps of this problem
    int pad_h, pad_w;  // MODE 0: conv padding; MODE 1/2: hs = hd + pad_h - r
    int oh0, ow0;      // MODE 2: dst pixel = (f*hd + oh0, f*wd + ow0)
};

constexpr int MAX_CLASSES = 64;

__host__ __device__ inline int taps_of_class(int R, int ph, int f) { return ph < R ? (R - ph + f - 1) / f : 0; }

// class c = ph*f + pw of a transposed conv (R x S taps, stride f, padding pad) onto an Hdf x Wdf output
__host__ __device__ inline IgemmClass class_geom(int c, int f, int R, int S, int pad, int N, int Hdf, int Wdf, int Cs) {
    IgemmClass q;
    const int ph = c / f, pw = c - ph * f;
    q.R = taps_of_class(R, ph, f);
    q.S = taps_of_class(S, pw, f);
    q.oh0 = (((ph - pad) % f) + f) % f;
    q.ow0 = (((pw - pad) % f) + f) % f;
    q.Hd = q.oh0 < Hdf ? (Hdf - q.oh0 + f - 1) / f : 0;
    q.Wd = q.ow0 < Wdf ? (Wdf - q.ow0 + f - 1) / f : 0;
    q.pad_h = (q.oh0 + pad - ph) / f;  // exact: oh0 + pad - ph is a multiple of f
    q.pad_w = (q.ow0 + pad - pw) / f;
    q.M = N * q.Hd * q.Wd;
    q.K = q.R * q.S * Cs;
    q.KT = (q.K + 15) / 16;
    return q;
}

struct IgemmParams {
    const void* src;    // [N,Hs,Ws,Cs], activation type AT
    const float* wpk;   // per problem: [KT*4][Cd][4]
    const float* bias;  // [Cd] or null
    void* dst;          // [N,Hdf,Wdf,Cd], activation type AT (split-K: fp32 slabs)
    int N, Hs, Ws, Cs, Cd, Hdf, Wdf, R, S, stride, pad, accumulate, ncls;
    // ---- (kept at the FRONT of the block, in the kernel-argument lines a workgroup reads first: a workgroup of winograd_f32_kernel —
    // two resident per CU — spent ~4 us between launch and its first load, much of it in dependent scalar-load round trips to the far
    // end of this 1.2 KB argument block and in three integer divisions; round-5 trace)
    // winograd_f32_kernel only: src is the INPUT of a BatchNorm + ReLU whose output the conv consumes; relu(fma(src, in_scale[c],
    // in_shift[c])) is applied while the patch is staged (the activation tensor is never written; same arithmetic as bn_apply_kernel)
    const float *in_scale, *in_shift;
    // Inference epilogue (round 5: eval-mode BatchNorm folded into the weights, `bias` = its shift): `res` non-NULL (with accumulate = 1) —
    // the tensor that is added comes from `res` (same shape / storage type as dst) instead of from dst itself: a residual connection;
    // relu != 0: max(., 0) on the final value (after bias and the addition)
    const void* res;
    int relu;
    // winograd_f32_kernel, round 5: persistent workgroups pull (patch, channel tile) items from `work` — [8 per-XCD counters][1 exit counter]
    // ints, zero on entry and left zero; NULL: one workgroup per item (gridDim.x == work_items)
    int work_items;
    int* work;
    // winograd_f32_kernel: the item -> (image, patch row, patch column, channel tile) divisions as host-made multipliers,
    // x / d = (x * magic(d)) >> 40, magic(d) = ceil(2^40 / d) (41 bits; exact for x < 2^23, d <= 2^16): d = channel tiles, patches (LIN:
    // tile groups) per image, patches per row
    unsigned long long wino_m_ntn, wino_m_tpi, wino_m_tw;
    int wino_tpi, wino_tw;
    int stat_rows;      // rows of the partials array (all M-tiles of the call; a call over many images runs as several launches)
    int stat_row0;      // first row this launch writes
    float* stats;       // optional BatchNorm partials [3][Cd][stat_rows] (pivot, sum, sum sq) + [stat_rows] counts
    // optional (EPI = 1 instantiations): dst is the gradient dz of a BatchNorm's (ReLU'd) output; the epilogue also reduces, per
    // channel and tile, the two sums of that BatchNorm's backward over the FINAL dst values (after accumulate):
    //   g = dz * [mask > 0],  bnb_part[0][c][row] = sum g,  bnb_part[1][c][row] = sum g * (y - mean[c]) * rstd[c]
    // y (the BatchNorm's input, same shape / layout / storage type as dst); mask = bnb_zmask (a saved activation of that shape)
    // or, without it, the BatchNorm's own output recomputed as fma(y, bnb_msc[c], bnb_msh[c]); rows as for `stats`.
    const void* bnb_y;
    const void* bnb_zmask;
    const float *bnb_msc, *bnb_msh, *bnb_mean, *bnb_rstd;
    float* bnb_part;
    // a SECOND BatchNorm consuming the same dz under the same mask tensor (a residual block with a projection shortcut: bn2 and
    // the downsample BatchNorm, resnet.py:84-91): its input, statistics and partials (null: none; needs bnb_zmask)
    const void* bnb_y2;
    const float *bnb_mean2, *bnb_rstd2;
    float* bnb_part2;
    // optional in-kernel finalize of those sums (null: the caller folds the partial rows itself): the last workgroup to finish
    // in each group of 64 partial rows folds the group, the last group-folder of an output-channel tile folds the groups and
    // writes that tile's channels of the results — fixed summation order, no float atomics (the counters are integers and are
    // left at zero).  bnb_cnt: [Cd/BN][1 + groups] ints; bnb_grp: [2 BatchNorms][2][Cd][groups] floats; results per BatchNorm b:
    // bnb_c1c2[b] = [2][Cd] (sum / M, sum_xhat / M: what the apply pass needs), bnb_dgamma[b], bnb_dbeta[b] (times bnb_gscale).
    int* bnb_cnt;
    float* bnb_grp;
    float *bnb_c1c2[2], *bnb_dgamma[2], *bnb_dbeta[2];
    float bnb_gscale, bnb_invM;
    // optional in-kernel finalize of the train-mode BatchNorm STATISTICS (`stats` rows; round 6, dbn_conv_bn_set_final): the workgroup that
    // completes a group of 64 partial rows folds the group, the one that completes the last group folds the groups and writes what
    // bn_finalize_tiles_kernel would (scale / shift / saved mean / rstd / running statistics) — no finalize launch behind the conv.
    // bnf_cnt: [Cd / 64][1 + groups] ints, zero on entry and left zero; bnf_grp: [Cd][groups][4] doubles of scratch.
    int* bnf_cnt;
    double* bnf_grp;
    const float *bnf_gamma, *bnf_beta;
    float bnf_eps, bnf_momentum;
    float *bnf_run_mean, *bnf_run_var, *bnf_scale, *bnf_shift, *bnf_mean, *bnf_rstd;
    unsigned src_bytes;
    unsigned plane_bytes;  // AT = 3: distance between the three bf16 planes of src (0 otherwise)
    // MODE 2 only: per class its number of tiles, first M-tile index, weight-panel offset (floats)
    int tile_end[MAX_CLASSES], row_base[MAX_CLASSES], wpk_off[MAX_CLASSES];
    // split-K (MODE 0/1, Cs % 16 == 0): workgroup row blockIdx.y reduces k-tiles [y*kt_per, (y+1)*kt_per) into slab y of dst
    int ksplit, kt_per;
    // First-round stagger (see igemm_f32_kernel): workgroups with blockIdx.x < stagger_blocks wait (their wave slot) * stagger_units
    // * 1024 clocks before they start, so that the workgroups sharing a CU run out of phase.  0: off.
    int stagger_units, stagger_blocks;
    int phase_prio;  // 1: the prologue and the epilogue run at raised wave priority (s_setprio), the main loop at the default
    // -DDBN_TRACE=1 builds (make TRACE=1, tools/trace_probe.py): [gridDim.x][8] timestamps (s_memrealtime, 100 MHz) written by thread 0 —
    // 0 entry, 1 main loop entered, 2 main loop done, 3 epilogue done, 7 HW_ID; null otherwise
    unsigned long long* trace;
    int launch_rows;    // host only: M-tiles of this launch (set by launch_igemm_ns)
    int patch;          // host only: use the pixel-patch form (3x3, stride 1; see igemm_dispatch)
    // MODE 3 only (pyramid conv): level g source [N, Hdf>>g, Wdf>>g, Cs] and its stride-2^g transposed-conv panels
    const void* seg_src[4];
    const float* seg_wpk[4];
    unsigned seg_bytes[4];
    unsigned seg_plane_bytes[4];  // AT = 3
    int first_level;  // MODE 3, exact-fp32 loop only: levels [first_level, 4) (the finer ones are in dst already: accumulate = 1)
    int pyr_group;    // MODE 3 tile order: 0 = row tile major, the 64 pixel classes minor; G > 0 = blocks of G row tiles x 8 classes of one class row (launch_wide)
};

// permille of the nominal first-round stagger (0 = off): dbn_set_stagger
extern "C" int dbn_g_stagger;
extern "C" int dbn_g_phase_prio;
extern "C" unsigned long long* dbn_g_trace;
extern "C" long dbn_g_trace_blocks;
#ifndef DBN_TRACE
#define DBN_TRACE 0
#endif
#if DBN_TRACE
#define DBN_TRACE_MARK(i)                                                                                      \
    do {                                                                                                       \
        if (p.trace && threadIdx.x == 0) p.trace[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DBN_TRACE_MARK(i) ((void)0)
#endif

namespace {

constexpr unsigned OOB_OFFSET = 0xF8000000u;  // beyond any tensor (< 0xF0000000 bytes): buffer loads return 0

__device__ __forceinline__ f32x4 buffer_load_f32x4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

// q = p / d, r = p % d for 0 <= p < 2^24 using a float reciprocal (exact after one correction step)
__device__ __forceinline__ void divmod24(int p, int d, float rd, int& q, int& r) {
    q = (int)((float)p * rd);
    r = p - q * d;
    if (r < 0) {
        r += d;
        --q;
    } else if (r >= d) {
        r -= d;
        ++q;
    }
}


// ---- split-bf16 matrix math (NS > 0) ------------------------------------------------------------
// NS = 1: operands rounded to bf16 (bf16 MFMA, fp32 accumulate).
// NS = 3: every fp32 operand is split exactly into three bf16 terms (a = a0 + a1 + a2, 24 mantissa
//         bits) and the product is evaluated as a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 on the bf16
//         matrix pipe with fp32 accumulation: fp32-accurate (mean rel. error 1.3e-7 at K=2304, lower than
//         a plain fp32 fmaf chain) at 6/16 of the fp32-MFMA cost.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_bits_rne(float x) {  // round-to-nearest-even fp32 -> bf16 bit pattern
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// four fp32 values -> NS x (four bf16 packed in 8 bytes).  The casts compile to v_cvt_pk_bf16_f32
// (round-to-nearest-even); the residual a - bf16(a) is exact in fp32.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NS>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[NS > 0 ? NS : 1]) {
    float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bf16x2 lo = {(__bf16)r[0], (__bf16)r[1]};
        const bf16x2 hi = {(__bf16)r[2], (__bf16)r[3]};
        const unsigned ulo = __builtin_bit_cast(unsigned, lo), uhi = __builtin_bit_cast(unsigned, hi);
        out[s] = u32x2{ulo, uhi};
        if (s + 1 < NS) {
            r[0] -= __builtin_bit_cast(float, ulo << 16);
            r[1] -= __builtin_bit_cast(float, ulo & 0xFFFF0000u);
            r[2] -= __builtin_bit_cast(float, uhi << 16);
            r[3] -= __builtin_bit_cast(float, uhi & 0xFFFF0000u);
        }
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NS, int MI, int NI, int F16 = 0>
__device__ __forceinline__ void mfma_split(const bf16x8 (&af)[NS > 0 ? NS : 1][MI], const bf16x8 (&bf)[NS > 0 ? NS : 1][NI],
                                           f32x16 (&acc)[MI][NI]) {
    if constexpr (F16) {  // fp16 operands (inference, BASELINE configs[4]): the same 16-byte fragments, v_mfma_f32_32x32x16_f16
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][a]), __builtin_bit_cast(f16x8, bf[0][b]),
                                                                   acc[a][b], 0, 0, 0);
        return;
    }
    // smallest terms first
    constexpr int NPROD = NS == 3 ? 6 : 1;
    constexpr int pi[6] = {2, 0, 1, 1, 0, 0};
    constexpr int pj[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < NPROD; ++t) {
        const int i = NS == 3 ? pi[t] : 0, j = NS == 3 ? pj[t] : 0;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][a], bf[j][b], acc[a][b], 0, 0, 0);
    }
}

// ---- in-kernel finalize of the BatchNorm statistics rows (IgemmParams::bnf_*).  Called by EVERY thread of a workgroup after it wrote its
// partial row `trow` (the row's stores through dbn_stat_put: memory-side when the finalize is on).  BN: channels [n0, n0 + BN) of tile column
// nt belong to this workgroup's rows; blockDim.x % BN == 0.  s_flag: one int of LDS; sd: 3 * blockDim.x doubles of LDS (dead panels).
// Hand-over as igemm_kernel.h's bnb_finish: sc1 stores of the payload, s_waitcnt vmcnt(0), agent-scope integer counters, sc1 loads — no
// device-scope fence.  The merge is bn_finalize_tiles_kernel's (every partial shifted to a common pivot, fp64), in two levels; the
// order of every sum is fixed by (blockDim.x, BN, rows): bit-reproducible.
// (the kernels hold their IgemmParams in different address spaces — a by-value copy of the fields the finalize needs travels instead)
struct BnStatFinal {
    float* stats;
    int stat_rows, Cd;
    int* bnf_cnt;
    double* bnf_grp;
    const float *bnf_gamma, *bnf_beta;
    float bnf_eps, bnf_momentum;
    float *bnf_run_mean, *bnf_run_var, *bnf_scale, *bnf_shift, *bnf_mean, *bnf_rstd;
};
#define DBN_BNF_ARGS(p)                                                                                                              \
    BnStatFinal {                                                                                                                    \
        (p).stats, (p).stat_rows, (p).Cd, (p).bnf_cnt, (p).bnf_grp, (p).bnf_gamma, (p).bnf_beta, (p).bnf_eps, (p).bnf_momentum,        \
            (p).bnf_run_mean, (p).bnf_run_var, (p).bnf_scale, (p).bnf_shift, (p).bnf_mean, (p).bnf_rstd                                \
    }
__device__ __forceinline__ void dbn_stat_put(bool fin, float* ptr, float v) {
    if (fin) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *ptr = v;
}
__device__ __forceinline__ void dbn_bn_stats_finish(const BnStatFinal p, int trow, int nt, int n0, int BN, int* s_flag, double* sd) {
    if (!p.bnf_cnt) return;
    constexpr int G = 64;
    const int rows = p.stat_rows, NG = (rows + G - 1) / G, g = trow / G;
    const int tid = threadIdx.x, NT = blockDim.x, parts = NT / BN;
    const int cl = tid % BN, part = tid / BN;
    const long c = n0 + cl;
    int* const cnt = p.bnf_cnt + nt * (NG + 1);
    auto xld = [](const float* q) { return (double)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto xldd = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto xstd = [](double* q, double v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto combine = [&](double& n, double& a1, double& a2) {  // the `parts` partial tuples of a channel (same pivot), in the order of the parts
        __syncthreads();
        sd[tid] = n;
        sd[NT + tid] = a1;
        sd[2 * NT + tid] = a2;
        __syncthreads();
        if (part == 0)
            for (int q = 1; q < parts; ++q) {
                n += sd[q * BN + cl];
                a1 += sd[NT + q * BN + cl];
                a2 += sd[2 * NT + q * BN + cl];
            }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial-row stores have completed
    __syncthreads();
    DBN_RACE_JITTER();
    if (tid == 0) {
        const int gsize = min(G, rows - g * G);
        const int last = __hip_atomic_fetch_add(cnt + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
        if (last) __hip_atomic_store(cnt + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (nobody touches it again in this call)
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    asm volatile("" ::: "memory");
    {   // ---- this group's rows -> one tuple (pivot of its first row, count, sum, sum of squares about that pivot) per channel
        const int r0 = g * G, r1 = min(rows, r0 + G);
        const float* const pv = p.stats + (0L * p.Cd + c) * rows;
        const float* const s1 = p.stats + (1L * p.Cd + c) * rows;
        const float* const s2 = p.stats + (2L * p.Cd + c) * rows;
        const float* const cn = p.stats + 3L * p.Cd * rows;
        const double P0 = xld(pv + r0);
        double n = 0.0, a1 = 0.0, a2 = 0.0;
        int r = r0 + part;
        for (; r + 3 * parts < r1; r += 4 * parts) {  // four rows in flight
            double nq[4], dq[4], tq[4], uq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                nq[u] = xld(cn + r + u * parts);
                dq[u] = xld(pv + r + u * parts);
                tq[u] = xld(s1 + r + u * parts);
                uq[u] = xld(s2 + r + u * parts);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double d = dq[u] - P0;
                n += nq[u];
                a1 += tq[u] + nq[u] * d;
                a2 += uq[u] + d * (2.0 * tq[u] + nq[u] * d);
            }
        }
        for (; r < r1; r += parts) {
            const double nq = xld(cn + r), d = xld(pv + r) - P0, tq = xld(s1 + r);
            n += nq;
            a1 += tq + nq * d;
            a2 += xld(s2 + r) + d * (2.0 * tq + nq * d);
        }
        combine(n, a1, a2);
        if (part == 0) {
            double* const Gq = p.bnf_grp + ((long)c * NG + g) * 4;
            xstd(Gq + 0, P0);
            xstd(Gq + 1, n);
            xstd(Gq + 2, a1);
            xstd(Gq + 3, a2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DBN_RACE_JITTER();
    if (tid == 0) {
        const int last = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NG - 1;
        if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    asm volatile("" ::: "memory");
    {   // ---- the groups -> the BatchNorm's coefficients (bn_finalize_tiles_kernel's arithmetic)
        const double* const Gc = p.bnf_grp + (long)c * NG * 4;
        const double P0 = xldd(Gc);
        double n = 0.0, a1 = 0.0, a2 = 0.0;
        for (int q = part; q < NG; q += parts) {
            const double pq = xldd(Gc + 4 * q), nq = xldd(Gc + 4 * q + 1), tq = xldd(Gc + 4 * q + 2), uq = xldd(Gc + 4 * q + 3);
            const double d = pq - P0;
            n += nq;
            a1 += tq + nq * d;
            a2 += uq + d * (2.0 * tq + nq * d);
        }
        combine(n, a1, a2);
        if (part == 0) {
            const double m1 = a1 / n, mean = P0 + m1, m2 = a2 - a1 * m1;
            double var = m2 / n;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.bnf_eps));
            const float meanf = (float)mean;
            const float sc = p.bnf_gamma[c] * rstd;
            p.bnf_scale[c] = sc;
            p.bnf_shift[c] = fmaf(-meanf, sc, p.bnf_beta[c]);
            p.bnf_mean[c] = meanf;
            p.bnf_rstd[c] = rstd;
            if (p.bnf_run_mean) {
                const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
                p.bnf_run_mean[c] = (1.f - p.bnf_momentum) * p.bnf_run_mean[c] + p.bnf_momentum * meanf;
                p.bnf_run_var[c] = (1.f - p.bnf_momentum) * p.bnf_run_var[c] + p.bnf_momentum * (float)unb;
            }
        }
    }
}

// panel size in floats of one problem: fp32 panels Kpad*Cd; split-bf16 panels Kpad*Cd*ns/2
inline long panel_floats(int K, int Cd, int ns) {
    const long kp = ((K + 15) / 16) * 16;
    return ns == 0 ? kp * Cd : kp * Cd * ns / 2;
}

}  // namespace

// ---- image chunking ---------------------------------------------------------------------------------------------
// The kernels index pixels with 24-bit reciprocal divisions and address tensors through raw buffer descriptors with 32-bit
// byte offsets.  Images are independent in every convolution, so a call whose tensors exceed those ranges runs as several
// launches over consecutive image ranges (same tile configuration; BatchNorm partial rows simply continue).
// (defined in conv.hip; dbn_set_index_limits lowers them for the tests)
extern long dbn_g_pixel_limit;  // rows per launch (divmod24)
extern long dbn_g_byte_limit;   // bytes addressable through one buffer descriptor
extern long dbn_g_elem_limit;   // 32-bit element offsets of the epilogue

static inline int chunk_images(int N, long px_rows, long src_bytes_per_image, long dst_elems_per_image, long src2_bytes_per_image = 0) {
    long n = N;
    auto fit = [&](long per_image, long limit) {
        if (per_image > 0 && per_image * n >= limit) n = (limit - 1) / per_image;
    };
    fit(px_rows, dbn_g_pixel_limit);
    fit(src_bytes_per_image, dbn_g_byte_limit);
    fit(src2_bytes_per_image, dbn_g_byte_limit);
    fit(dst_elems_per_image, dbn_g_elem_limit);
    return (int)n;  // 0: a single image does not fit
}

// launchers of the kernel translation units.  cfg: 1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64; mode: kernel MODE 0..3
int dbn_launch_convt_f32(IgemmParams& p, hipStream_t st);  // convt_f32.hip
// conv.hip: folds [3][C][rows] (+ [rows] counts) tile statistics into the train-mode BatchNorm's coefficients and running statistics
int dbn_launch_bn_finalize_tiles(const float* ws, int rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                                 float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st);
int dbn_launch_winograd_f32(IgemmParams& p, hipStream_t st);                                       // winograd_f32.hip
int dbn_launch_winograd_pack(const float* w, int O, int I, int Cs, int dgrad, float* out, hipStream_t st);
int dbn_launch_winograd_pack_many(const void* jobs, int n, hipStream_t st);
int dbn_winograd_linear(int H, int W);  // winograd_f32.hip: the map runs in the consecutive-tile (LIN) form
extern "C" int dbn_winograd_rows(int N, int H, int W);  // winograd_f32.hip
int dbn_convt_f32_rows(int M);
int dbn_launch_igemm_f32(IgemmParams& p, int cfg, int mode, hipStream_t st);                    // conv_f32.hip: exact fp32 (ns 0, at 0)
int dbn_launch_igemm_x(IgemmParams& p, int cfg, int mode, int ns, int at, hipStream_t st);      // conv_x3.hip: fp32 tensors, bf16 math
int dbn_launch_igemm_b16(IgemmParams& p, int cfg, int mode, int at, hipStream_t st);            // conv_b16.hip: bf16 / fp16 storage
// wres16.hip: the weight-resident 3x3 / stride-1 kernel of the 16-bit storage types (takes the pixel-patch launches it is eligible for)

// The 128 x 256 tile of the 16-bit storage types (igemm_kernel.h launch_wide; dbn_set_pyramid_wide): which launches take it, by geometry.
// mode 3 = the pyramid conv (tiles of 128 8 x 8 blocks x 64 pixel classes), 0 / 1 = plain forward / stride-1 data gradient of the generic
// loop.  Enough tiles for eight rounds of the 512 resident workgroups — below that the narrower tile's finer tail wins (bf16 training at
// 16 x 640^2: 3200 wide tiles, 1774 against 1784 images/s; the 512 -> 512 convs of configs[4] at 40 x 40: 800 wide tiles, 11.15 against
// 10.94 ms per forward); dbn_g_wide_tile: 0 off, 1 the pyramid form, 2 also the generic launches, 3 as 2 whatever the size (tests).
extern "C" int dbn_g_wide_tile;
extern "C" int dbn_g_pyr_group;  // conv.hip: row tiles per block of the wide pyramid tile's order (0 = class-minor order of rounds 3-5; DBN_PYR_GROUP)
static inline bool dbn_wide_tile_geom_ok(int mode, long N, int Hdf, int Wdf, int Cs, int Cd) {
    if (Cd % 256 != 0 || (Cs & 15) != 0) return false;
    const long min_tiles = dbn_g_wide_tile >= 3 ? 0 : 4096;
    if (mode == 3) return dbn_g_wide_tile >= 1 && N * (Hdf >> 3) * (Wdf >> 3) / 2 * (Cd / 256) >= min_tiles;
    if (mode == 0 || mode == 1) return dbn_g_wide_tile >= 2 && N * Hdf * Wdf / 128 * (Cd / 256) >= min_tiles;
    return false;
}
bool dbn_wres16_eligible(int at, int mode, int N, int H, int W, int Cs, int Cd, bool bnb, bool y2, bool stats);
int dbn_launch_wres16(IgemmParams& p, int mode, int at, hipStream_t st);
