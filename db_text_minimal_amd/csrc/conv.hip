// Convolution entry points of the C ABI (include/dbnet_hip.h): tile choice, image chunking, split-K, parity classes, the fused
// conv + BatchNorm-statistics call and the pyramid conv.  The kernels live in igemm_kernel.h (instantiated by conv_f32.hip,
// conv_x3.hip, conv_b16.hip).
//
// Replaces the ATen convolution calls under /root/reference/src/modules/resnet.py:70-91,231-242,
// modules/basic.py:32-36, modules/segmentation_body.py:64-77 and modules/segmentation_head.py:24-29,64-79
// (Conv2d / ConvTranspose2d forward and their data gradients).
#include "igemm_common.h"
#include "../../include/dbnet_hip.h"

long dbn_g_pixel_limit = 1L << 24;
long dbn_g_byte_limit = 0xF0000000L;
long dbn_g_elem_limit = 1L << 32;

namespace {

// Merge the per-tile BatchNorm partials written by the igemm epilogue (Chan et al. parallel variance, fp64)
// into scale/shift, saved mean/rstd and the running statistics.  One 256-thread block per channel.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = dbn_wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

__global__ void bn_finalize_tiles_kernel(const float* __restrict__ stats, int rows, int C, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, float momentum, float* __restrict__ run_mean,
                                         float* __restrict__ run_var, float* __restrict__ scale, float* __restrict__ shift,
                                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    const float* pv = stats + (0L * C + c) * rows;
    const float* s1 = stats + (1L * C + c) * rows;
    const float* s2 = stats + (2L * C + c) * rows;
    const float* cn = stats + 3L * C * rows;
    // one pass: shift every tile's sums from its own pivot to the first tile's pivot P0 (exact algebra, fp64):
    //   sum (x-P0) = s1 + n d,   sum (x-P0)^2 = s2 + 2 d s1 + n d^2,   d = pivot - P0
    const double p0 = (double)pv[0];
    double n = 0.0, a1 = 0.0, a2 = 0.0;
    for (int t = threadIdx.x; t < rows; t += blockDim.x) {
        const double nt = (double)cn[t], d = (double)pv[t] - p0, t1 = (double)s1[t];
        n += nt;
        a1 += t1 + nt * d;
        a2 += (double)s2[t] + d * (2.0 * t1 + nt * d);
    }
    n = block_sum_d(n, red);
    a1 = block_sum_d(a1, red);
    a2 = block_sum_d(a2, red);
    const double m1 = a1 / n;
    const double mean = p0 + m1;
    const double m2 = a2 - a1 * m1;
    if (threadIdx.x != 0) return;
    double var = m2 / n;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = fmaf(-meanf, sc, beta[c]);
    mean_out[c] = meanf;
    rstd_out[c] = rstd;
    if (run_mean) {
        const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * meanf;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// dst = [dst +] bias + sum over the split-K slabs, fixed order
template <int AT>
__global__ void splitk_sum_kernel(const float* __restrict__ slab, int splits, long total4, int Cd, const float* __restrict__ bias,
                                  int accumulate, void* __restrict__ dst) {
    const int c4n = Cd >> 2;
    const long stride4 = total4 + 272;  // slabs are 1088 floats apart beyond their size (HBM channel rotation, see wgrad_slab_stride)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slab) + i;
        f32x4 v = src[0];
        for (int z = 1; z < splits; ++z) v += src[(long)z * stride4];
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + 4 * (int)(i % c4n));
        if (accumulate) v += dbn_ld4<AT>(dst, i);
        dbn_st4<AT>(dst, i, v);
    }
}

// ns: matrix math (0 exact fp32, 1 one 16-bit plane, 3 bf16x3); at: activation storage (0 fp32, 1 bf16, 2 fp16; 16-bit storage needs ns 1)
int launch_igemm(IgemmParams& p, int cfg, int mode, int ns, hipStream_t st, int at) {
    if (at == 1 || at == 2) return ns == 1 ? dbn_launch_igemm_b16(p, cfg, mode, at, st) : DBN_ERR_ARG;
    if (at == 3) return dbn_launch_igemm_x(p, cfg, mode, ns, at, st);
    if (ns == 0) return dbn_launch_igemm_f32(p, cfg, mode, st);
    return dbn_launch_igemm_x(p, cfg, mode, ns, 0, st);
}

}  // namespace

int dbn_launch_bn_finalize_tiles(const float* ws, int rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                                 float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(C), dim3(rows >= 2048 ? 1024 : 256), 0, st, ws, rows, C, gamma, beta, eps, momentum,
                       run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
}

extern "C" {

int dbn_has_experiments(void) { return DBN_HAS_EXPERIMENTS; }


// Tile configuration dbn_igemm_f32 picks for an M x Cd output (tile_hint 0):
// 1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64 — the largest tile that still yields
// >= ~2 workgroups per CU on 256 CUs.
static int tile_config_for(int M, int Cd, int ns) {
    // Workgroups are handed to the 256 CUs as they free up, so a launch lasts about
    // ceil(blocks/256) tiles per CU; pick the tile that minimises tiles-per-CU x tile area / efficiency
    // (efficiency = measured steady-state MFMA utilisation of each variant).
    // Round 2 re-measured the variants alone on the backbone's four stage shapes (tools/tile_probe.py): the smaller tiles are the
    // faster ones in isolation (64x64: 115 / 125 / 112 TFLOP/s at 160x160x64 / 80x80x128 / 40x40x256 against 108 / 114 / 97 for
    // 128x64 and 103 / 95 for 128x128, round-3 kernels).  Inside the two-stream step that did not carry over in round 2 (+0.5 %,
    // within noise); with round 3's k-loop (scalar-offset addressing, no drained prefetch) it does: efficiencies
    // {0.89, 0.85, 0.845, 0.83} give 524 images/s against 511 for round 1's {0.89, 0.83, 0.80, 0.72} (interleaved A/B on one box,
    // 510.9 / 524.4 / 523.9 / 510.3 / 523.9; a table favouring 64x64 even more, {0.80, 0.76, 0.86, 0.93}, measures the same).
    // (-DDBN_EXPERIMENTS builds: DBN_TILE_EFF_R2=0 selects round 1's table, 2 the third one)
    const int bm[4] = {128, 256, 128, 64}, bn[4] = {128, 64, 64, 64};
    // (the 16-bit matrix modes keep round 1's table: their DMA-ring kernels lose with the small tiles — bf16 1619 -> 1565 images/s)
    static const int r2_env = dbn_env_int("DBN_TILE_EFF_R2", 1);
    const int r2_eff = ns == 0 ? r2_env : 0;
    const double eff_r1[4] = {0.89, 0.83, 0.80, 0.72}, eff_r2[4] = {0.89, 0.85, 0.845, 0.83}, eff_r3[4] = {0.80, 0.76, 0.86, 0.93};
    const double* eff = r2_eff == 2 ? eff_r3 : r2_eff ? eff_r2 : eff_r1;
    int best = 4;
    double best_t = 1e300;
    for (int c = 0; c < 4; ++c) {
        if (Cd % bn[c]) continue;
        const long blocks = (long)dbn_ceil_div(M, bm[c]) * (Cd / bn[c]);
        const long per_cu = (blocks + 255) / 256;
        double t = (double)per_cu * bm[c] * bn[c] / eff[c];
        if (blocks < 512) t *= 1.0 + 0.25 * (512 - blocks) / 512.0;  // too few workgroups to hide latency
        if (t < best_t) {
            best_t = t;
            best = c + 1;
        }
    }
    return best;
}
int dbn_igemm_tile_config(int M, int Cd) { return tile_config_for(M, Cd, 0); }  // exact-fp32 choice
int dbn_igemm_tile_config_ns(int M, int Cd, int ns) { return tile_config_for(M, Cd, ns); }

// ConvTranspose2d(2x2, stride 2) forward in exact fp32: the dedicated kernel of convt_f32.hip (input tile resident in LDS, the four
// parity classes walked inside the workgroup) instead of the general parity-class launch
static int g_convt_enabled = 1;
int dbn_set_convt_kernel(int on) {  // test / A-B hook: 0 routes the layer through igemm_f32_kernel MODE 2 again
    const int old = g_convt_enabled;
    g_convt_enabled = on != 0;
    return old;
}
static bool convt_eligible(int mode, int ns, int at, int tile_hint, int R, int S, int stride, int pad, int Hs, int Ws, int Hd, int Wd, int Cs,
                           int Cd, int accumulate, int ksplit) {
    return g_convt_enabled && mode == 1 && ns == 0 && at == 0 && tile_hint == 0 && R == 2 && S == 2 && stride == 2 && pad == 0 && Hd == 2 * Hs &&
           Wd == 2 * Ws && (Cs == 16 || Cs == 32 || Cs == 48 || Cs == 64) && Cd % 64 == 0 && !accumulate && ksplit <= 1;
    // (+ per launch: convt_launch_ok)
}

// ... and the output of one launch (n images) below 4 GB: the kernel addresses rows with 32-bit byte offsets
static bool convt_launch_ok(int n, int Hd, int Wd, int Cd) { return (long)n * Hd * Wd * Cd * 4 < (1L << 32); }

unsigned long long* dbn_g_trace = nullptr;
long dbn_g_trace_blocks = 0;
// -DDBN_TRACE=1 builds: the next exact-fp32 implicit-GEMM launches of at most `max_blocks` workgroups write [grid][8] timestamps to buf
// (see IgemmParams::trace); null switches it off.  Product builds accept the call and ignore it.
int dbn_set_trace(void* buf, long max_blocks) {
    dbn_g_trace = reinterpret_cast<unsigned long long*>(buf);
    dbn_g_trace_blocks = max_blocks;
    return DBN_TRACE;
}
int dbn_g_phase_prio = 0;
// 1: implicit-GEMM workgroups run their prologue and epilogue at raised wave priority (igemm_kernel.h); returns the old setting
int dbn_set_phase_priority(int on) {
    const int old = dbn_g_phase_prio;
    dbn_g_phase_prio = on != 0;
    return old;
}
int dbn_g_stagger = 0;
// permille of the nominal first-round stagger of the exact-fp32 implicit-GEMM launches (igemm_kernel.h); 0 = off.  Returns the old value.
int dbn_set_stagger(int permille) {
    const int old = dbn_g_stagger;
    dbn_g_stagger = permille < 0 ? 0 : permille;
    return old;
}

static bool patch_eligible_fwd(int mode, int at, int H, int W, int Cs);
// the weight-resident 3x3 kernel of the 16-bit storage types (wres16.hip): 1 = takes every pixel-patch launch it is eligible for
// (default), 0 = off (test / A-B hook: the pixel-patch kernel again), 2 = also maps whose width is not a multiple of the 32-column strip
// (correct but measured slower than the pixel-patch kernel there — dbn_wres16_eligible; the tests of the ragged last strip use it)
int dbn_g_wres16 = getenv("DBN_WRES16") ? atoi(getenv("DBN_WRES16")) : 1;  // (DBN_WRES16=0: A/B runs of whole programs)
int dbn_set_wres16(int on) {
    const int old = dbn_g_wres16;
    dbn_g_wres16 = on < 0 ? 0 : on > 2 ? 2 : on;
    return old;
}
// Would a dbn_igemm_t / dbn_conv_bn_t / dbn_igemm_bnsums_t call with this geometry (3x3, stride 1, pad 1 implied) launch
// conv3x3_wres16_kernel<at, Cs, Cd, mode, epi>?  bnb / y2: the call carries the BatchNorm-backward sums / their second BatchNorm.
int dbn_wres16_would_run(int at, int mode, int N, int H, int W, int Cs, int Cd, int bnb, int y2) {
    return patch_eligible_fwd(mode, at, H, W, Cs) && Cd % 64 == 0 && dbn_wres16_eligible(at, mode, N, H, W, Cs, Cd, bnb != 0, y2 != 0, false);
}
static int g_patch_enabled = 1;  // 0: never; 1: default; 2: the 16-bit matrix modes only (exact fp32 takes the gather loop); 3: exact fp32 on every eligible launch
static const int g_patch_bn64 = dbn_env_int("DBN_PATCH_BN64", 1);
int dbn_set_patch_conv(int on) {  // test / A-B hook: 0 routes the 3x3 stride-1 convolutions through the generic gather loop again
    const int old = g_patch_enabled;
    g_patch_enabled = (on == 2 || on == 3) ? on : (on != 0);
    return old;
}
// pixel-patch form: 3x3 / stride 1 / pad 1 on the bf16 matrix pipe, whole 8 x 16 patches, 128-row tiles (the BatchNorm
// partial rows of a launch are the same N*H*W/128 either way); kmode: kernel MODE
// Exact fp32 (ns = 0, round 4): the same form with 16-channel blocks on v_mfma_f32_32x32x2_f32, 128 x 64 tiles whatever the generic
// heuristic picked (its choice for these layers is the 64 x 64 gather tile) — unless the caller names another tile.
static bool patch_eligible(int kmode, int ns, int at, int cfg, int R, int S, int stride, int pad, int Hs, int Ws, int Hd, int Wd, int Cs,
                           int ksplit, int tile_hint = 0) {
    const bool geom = (kmode == 0 || kmode == 1) && R == 3 && S == 3 && stride == 1 && pad == 1 && Hs == Hd && Ws == Wd && Hd % 8 == 0 &&
                      Wd % 16 == 0 && Cs % 32 == 0 && ksplit <= 1;
    // exact fp32: an explicit 128 x 64 request (tile_hint 3) takes it, and dbn_set_patch_conv(3) makes it the library's own choice
    // (tile_hint 0).  NOT the default: alone it wins only on the long-K forward convs (the head's 256 -> 64 at 160^2: 871-892 us against
    // 907-933 for the 64 x 64 gather tile; at K = 576 its heavier prologue / epilogue make it a tie or a loss: 64 -> 64 at 160^2
    // 286-300 vs 277-293 us), and inside the two-stream step neither choice moved the result (every eligible layer: 538-540 vs
    // 543-545 images/s; the head's forward convs only: 543.2 vs 542.6, within the run-to-run spread) — profiles/r04_fp32_patch.txt
    if (ns == 0) return (g_patch_enabled == 1 || g_patch_enabled == 3) && at == 0 && geom && (tile_hint == 3 || (tile_hint == 0 && g_patch_enabled == 3));
    return g_patch_enabled && ns > 0 && at != 3 && (cfg == 1 || cfg == 3) && geom;
}
static bool patch_eligible_fwd(int mode, int at, int H, int W, int Cs) {  // a 16-bit 3x3 / stride-1 / pad-1 call at the library's own tile choice
    return patch_eligible(mode, 1, at, 3, 3, 3, 1, 1, H, W, H, W, Cs, 1, 0);
}
static int patch_cfg(int cfg, int ns = 1) { return (ns == 0 || (cfg == 1 && g_patch_bn64)) ? 3 : cfg; }

// Tile configuration of a dbn_igemm / dbn_conv_bn call (`mode`, `stride` as the caller passes them).  The convolutions that can take
// the pixel-patch kernel get a 128-row tile whatever the generic heuristic says.
static int resolve_cfg(int M_total, int Cd, int tile_hint, int at = 0, int ns = 0, int mode = 0, int R = 0, int S = 0, int stride = 1,
                       int pad = 0, int Hs = 0, int Ws = 0, int Hd = 0, int Wd = 0, int Cs = 0, int ksplit = 1) {
    int cfg = tile_hint > 0 ? tile_hint : tile_config_for(M_total, Cd, ns);
    if (cfg == 1 && Cd % 128 != 0) cfg = 3;
    if (tile_hint == 0 && !(mode == 1 && stride > 1) && Cd % 64 == 0 && patch_eligible(mode, ns, at, 3, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, ksplit))
        cfg = 3;
    return cfg;
}

// What one (unchunked) dbn_igemm_t call launches: tile configuration (1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64) + 16 if the
// pixel-patch kernel is used — i.e. the template arguments <BM,BN,WM,WN,MODE,NS,AT,PATCH> of its rocprofv3 symbol.
// kmode: 0 forward, 1 stride-1 data gradient, 2 parity classes, 3 pyramid.
int dbn_igemm_kernel_config(int at, int ns, int kmode, int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride,
                            int pad, int tile_hint, int ksplit) {
    const int cfg = resolve_cfg(N * Hd * Wd, Cd, tile_hint, at, ns, kmode >= 2 ? 1 : kmode, R, S, kmode == 2 && stride == 1 ? 2 : stride, pad, Hs,
                                Ws, Hd, Wd, Cs, ksplit);
    if (Cd % 64 == 0 && patch_eligible(kmode, ns, at, cfg, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, ksplit, tile_hint)) {
        // + 64: a PLAIN call (no sums / statistics epilogue) launches conv3x3_wres16_kernel<at, Cs, Cd, mode, 0>
        const int wres = (ns == 1 && patch_cfg(cfg, ns) == 3 && dbn_wres16_eligible(at, kmode, N, Hd, Wd, Cs, Cd, false, false, false)) ? 64 : 0;
        return patch_cfg(cfg, ns) + 16 + wres;
    }
    // + 32: the launch is convt2x2_f32_kernel<Cs> (a non-accumulating transposed 2x2 / stride-2 conv in exact fp32)
    if (kmode == 2 && convt_eligible(1, ns, at, tile_hint, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, Cd, 0, ksplit)) return cfg + 32;
    return cfg;
}

static int igemm_dispatch(IgemmParams& p, int kmode, int ns, int tile_hint, hipStream_t st, int at = 0) {
    // tile choice from the total row count (for parity classes: all classes together)
    int cfg = tile_hint > 0 ? tile_hint : tile_config_for(p.N * p.Hdf * p.Wdf, p.Cd, ns);
    if (cfg == 1 && p.Cd % 128 != 0) cfg = 3;
    p.patch = p.Cd % 64 == 0 && patch_eligible(kmode, ns, at, cfg, p.R, p.S, p.stride, p.pad, p.Hs, p.Ws, p.Hdf, p.Wdf, p.Cs, p.ksplit, tile_hint);
    // 128 x 64 tiles also where the generic loop takes 128 x 128: K is short (two to eight channel blocks), so twice the workgroups
    // hide the prologue / epilogue better than the wider tile saves weight traffic (measured: 120.8 GFLOP launch 509 -> ~270 us;
    // step +1-3 %); the BatchNorm partial rows depend on BM only.  DBN_PATCH_BN64=0 keeps the generic choice.
    if (p.patch) cfg = patch_cfg(cfg, ns);
    return launch_igemm(p, cfg, kmode, ns, st, at);
}

// bytes per element of the destination of a conv (at = 3: pre-split bf16 planes in, fp32 out) and planes of its source
static inline int dst_esize(int at) { return (at == 0 || at == 3) ? 4 : 2; }
static inline int src_planes(int at) { return at == 3 ? 3 : 1; }

// optional in-kernel finalize of the train-mode BatchNorm statistics (IgemmParams::bnf_*; dbn_conv_bn_set_final)
struct BnFin {
    int* cnt;
    double* grp;
    const float *gamma, *beta;
    float eps, momentum;
    float *run_mean, *run_var, *scale, *shift, *mean, *rstd;
};
static thread_local int* g_bnf_cnt = nullptr;
static thread_local double* g_bnf_grp = nullptr;
// The NEXT dbn_conv_bn_t / dbn_winograd_conv_bn[_act]_f32 call of this thread folds its statistics rows itself (the workgroup that
// completes the last group of 64 rows writes scale / shift / mean / rstd / running statistics): no bn_finalize_tiles_kernel launch behind
// the conv.  counters: dbn_igemm_bn_final_counters(rows, Cd) ints, ZERO before the first use (the kernels leave them zero), one set per
// call that may be in flight at a time; group: dbn_conv_bn_final_group_doubles(rows, Cd) doubles of scratch.  Consumed (or dropped, where
// the launch has no such epilogue: the 2x2 ConvTranspose kernel) by that next call; NULL clears it.
int dbn_conv_bn_set_final(int* counters, double* group) {
    g_bnf_cnt = counters;
    g_bnf_grp = counters ? group : nullptr;
    return (counters && !group) ? DBN_ERR_ARG : DBN_OK;
}
long dbn_conv_bn_final_group_doubles(int rows, int Cd) { return 4L * Cd * ((rows + 63) / 64); }

// optional BatchNorm-backward sums of a call (IgemmParams::bnb_*); y / zmask advance with dst over image chunks
struct IgemmBnb {
    const void *y, *zmask;
    const float *msc, *msh, *mean, *rstd;
    float* part;
    const void* y2;  // optional second BatchNorm over the same dz and mask tensor
    const float *mean2, *rstd2;
    float* part2;
    const dbn_bnb_final* fin;  // optional in-kernel finalize
    int M_total;               // pixels of the whole call (c1 = sum / M)
};

static int igemm_run_one(const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                         int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int cfg, int ns,
                         hipStream_t st, float* stats, int stat_rows, int stat_row0, int ksplit, float* slab, int* rows_out, int at,
                         long plane_bytes, const IgemmBnb* bnb = nullptr, int tile_hint = -1, const void* res = nullptr, int relu = 0,
                         const BnFin* bnf = nullptr) {
    IgemmParams p{};
    if (bnf) {
        p.bnf_cnt = bnf->cnt; p.bnf_grp = bnf->grp; p.bnf_gamma = bnf->gamma; p.bnf_beta = bnf->beta; p.bnf_eps = bnf->eps;
        p.bnf_momentum = bnf->momentum; p.bnf_run_mean = bnf->run_mean; p.bnf_run_var = bnf->run_var; p.bnf_scale = bnf->scale;
        p.bnf_shift = bnf->shift; p.bnf_mean = bnf->mean; p.bnf_rstd = bnf->rstd;
    }
    p.res = res; p.relu = relu;  // inference epilogue (dbn_igemm_act_t)
    p.bnb_y = bnb ? bnb->y : nullptr; p.bnb_zmask = bnb ? bnb->zmask : nullptr;
    p.bnb_msc = bnb ? bnb->msc : nullptr; p.bnb_msh = bnb ? bnb->msh : nullptr;
    p.bnb_mean = bnb ? bnb->mean : nullptr; p.bnb_rstd = bnb ? bnb->rstd : nullptr;
    p.bnb_part = bnb ? bnb->part : nullptr;
    p.bnb_y2 = bnb ? bnb->y2 : nullptr; p.bnb_mean2 = bnb ? bnb->mean2 : nullptr; p.bnb_rstd2 = bnb ? bnb->rstd2 : nullptr;
    p.bnb_part2 = bnb ? bnb->part2 : nullptr;
    p.bnb_cnt = nullptr; p.bnb_grp = nullptr; p.bnb_gscale = 1.f; p.bnb_invM = 0.f;
    for (int b_ = 0; b_ < 2; ++b_) p.bnb_c1c2[b_] = p.bnb_dgamma[b_] = p.bnb_dbeta[b_] = nullptr;
    if (bnb && bnb->fin) {
        const dbn_bnb_final* f = bnb->fin;
        p.bnb_cnt = f->counters; p.bnb_grp = f->group; p.bnb_gscale = f->grad_scale; p.bnb_invM = 1.0f / (float)bnb->M_total;
        p.bnb_c1c2[0] = f->c1c2; p.bnb_dgamma[0] = f->dgamma; p.bnb_dbeta[0] = f->dbeta;
        p.bnb_c1c2[1] = f->c1c2_2; p.bnb_dgamma[1] = f->dgamma_2; p.bnb_dbeta[1] = f->dbeta_2;
    }
    p.src = src; p.wpk = wpk; p.bias = bias; p.dst = dst;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.Cd = Cd; p.Hdf = Hd; p.Wdf = Wd;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.accumulate = res ? 1 : accumulate;
    p.stats = stats; p.stat_rows = stat_rows; p.stat_row0 = stat_row0; p.launch_rows = 0;
    p.ksplit = 1; p.kt_per = 0;
    // (at = 3: planes 1 and 2 lie plane_bytes and 2*plane_bytes behind the image range of plane 0 this launch covers)
    p.plane_bytes = (unsigned)plane_bytes;
    p.src_bytes = (unsigned)((long)N * Hs * Ws * Cs * dbn_esize(at) + (src_planes(at) - 1) * plane_bytes);
    if (!(mode == 1 && stride > 1)) {
        p.ncls = 1;
        if (ksplit <= 1) {
            const int rc = igemm_dispatch(p, mode, ns, cfg, st, at);
            *rows_out = p.launch_rows;
            return rc;
        }
        // split-K: slabs of partial sums, then a fixed-order reduction that also applies bias / accumulate
        const int KT = (R * S * Cs + 15) / 16;
        DBN_REQUIRE(slab && !stats && !bnb && Cs % 16 == 0 && ksplit <= KT && ksplit <= 64);
        p.kt_per = dbn_ceil_div(KT, ksplit);
        p.ksplit = dbn_ceil_div(KT, p.kt_per);
        p.dst = slab; p.bias = nullptr; p.accumulate = 0;
        const int rc = igemm_dispatch(p, mode, ns, cfg, st, at);
        *rows_out = p.launch_rows;
        if (rc) return rc;
        const long total4 = (long)N * Hd * Wd * Cd / 4;
        DBN_DISPATCH_AT(at == 3 ? 0 : at, hipLaunchKernelGGL(splitk_sum_kernel<AT>, dim3(dbn_grid(total4)), dim3(256), 0, st, slab, p.ksplit,
                                                             total4, Cd, bias, accumulate, dst));
        return dbn_status();
    }
    DBN_REQUIRE(ksplit <= 1);
    // stride-f data gradient / transposed conv: one problem per output parity class
    p.ncls = stride * stride;
    long off = 0;
    int covered = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const IgemmClass q = class_geom(c, stride, R, S, pad, N, Hd, Wd, Cs);
        p.wpk_off[c] = (int)off;
        if (q.K > 0 && q.M > 0) ++covered;
        off += panel_floats(q.K, Cd, ns);
    }
    DBN_REQUIRE(off < (1L << 31));
    DBN_REQUIRE(covered == p.ncls || !bias);  // a bias would have to reach the tap-less pixels too
    DBN_REQUIRE(covered == p.ncls || !bnb);   // ... and so would the BatchNorm-backward sums
    if (covered < p.ncls && !accumulate) {  // some output pixels receive no tap: they are zero
        if (hipMemsetAsync(dst, 0, (size_t)N * Hd * Wd * Cd * dst_esize(at), st) != hipSuccess) return dbn_status();
        p.accumulate = 1;
    }
    *rows_out = 0;
    if (covered == 0) return DBN_OK;
    if (!bnb && convt_eligible(mode, ns, at, tile_hint, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, Cd, accumulate, ksplit) &&
        convt_launch_ok(N, Hd, Wd, Cd)) {
        const int rc = dbn_launch_convt_f32(p, st);
        *rows_out = p.launch_rows;
        return rc;
    }
    const int rc = igemm_dispatch(p, 2, ns, cfg, st, at);
    *rows_out = p.launch_rows;
    return rc;
}

// Rows of BatchNorm partials ONE launch over n images produces
static int bn_tile_rows_one(int n, int Hd, int Wd, int mode, int stride, int cfg) {
    const int bm_of[5] = {0, 128, 256, 128, 64};
    if (!(mode == 1 && stride > 1)) return dbn_ceil_div((long)n * Hd * Wd, bm_of[cfg]);
    int rows = 0;
    for (int c = 0; c < stride * stride; ++c) {  // BN follows only tap-complete transposed convs: every class has pixels
        const int ph = c / stride, pw = c % stride;
        const int Hc = ph < Hd ? (Hd - ph + stride - 1) / stride : 0, Wc = pw < Wd ? (Wd - pw + stride - 1) / stride : 0;
        if (Hc > 0 && Wc > 0) rows += dbn_ceil_div((long)n * Hc * Wc, bm_of[cfg]);
    }
    return rows;
}

static int igemm_run(const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                     int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                     void* stream, float* stats = nullptr, int ksplit = 1, float* slab = nullptr, int stat_rows_total = 0, int at = 0,
                     const IgemmBnb* bnb = nullptr, const void* res = nullptr, int relu = 0, const BnFin* bnf = nullptr) {
    DBN_REQUIRE((!res && !relu) || (ksplit <= 1 && !stats && !bnb && !(mode == 1 && stride > 1)));  // inference epilogue: plain launches only
    DBN_REQUIRE(src && wpk && dst && (ns == 0 || ns == 1 || ns == 3));
    // 16-bit storage / pre-split planes: 8-channel pieces of 16-channel blocks
    DBN_REQUIRE(at == 0 || ((at == 1 || at == 2) && ns == 1 && Cs % 16 == 0) || (at == 3 && ns == 3 && Cs % 16 == 0));
    DBN_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && R > 0 && S > 0 && pad >= 0);
    DBN_REQUIRE(Cs % 4 == 0 && Cd % 64 == 0 && (mode == 0 || mode == 1));
    DBN_REQUIRE(stride == 1 || stride == 2 || stride == 4 || stride == 8 || (mode == 0 && stride >= 1));
    DBN_REQUIRE(tile_hint >= 0 && tile_hint <= 4);
    hipStream_t st = (hipStream_t)stream;
    const int es = dbn_esize(at), des = dst_esize(at);
    const long plane_bytes = at == 3 ? (long)N * Hs * Ws * Cs * 2 : 0;  // the planes of the WHOLE tensor are this far apart
    DBN_REQUIRE(3 * plane_bytes < dbn_g_byte_limit);
    const int nmax = chunk_images(N, (long)Hd * Wd, (long)Hs * Ws * Cs * es, (long)Hd * Wd * Cd);
    DBN_REQUIRE(nmax >= 1);               // one image must fit the kernel's index ranges
    DBN_REQUIRE(nmax >= N || ksplit <= 1);  // split-K is for small outputs only
    const int cfg = resolve_cfg((int)std::min<long>((long)N * Hd * Wd, 0x7FFFFFFF), Cd, tile_hint, at, ns, mode, R, S, stride, pad, Hs, Ws,
                                Hd, Wd, Cs, ksplit);
    int row0 = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int n = std::min(nmax, N - n0);
        int rows = 0;
        IgemmBnb b;
        if (bnb) {
            b = *bnb;
            b.y = reinterpret_cast<const char*>(bnb->y) + (long)n0 * Hd * Wd * Cd * des;
            if (bnb->zmask) b.zmask = reinterpret_cast<const char*>(bnb->zmask) + (long)n0 * Hd * Wd * Cd * des;
            if (bnb->y2) b.y2 = reinterpret_cast<const char*>(bnb->y2) + (long)n0 * Hd * Wd * Cd * des;
        }
        const int rc = igemm_run_one(reinterpret_cast<const char*>(src) + (long)n0 * Hs * Ws * Cs * es, wpk, bias,
                                     reinterpret_cast<char*>(dst) + (long)n0 * Hd * Wd * Cd * des, n, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride,
                                     pad, mode, accumulate, cfg, ns, st, stats, stat_rows_total, row0, ksplit, slab, &rows, at, plane_bytes,
                                     bnb ? &b : nullptr, tile_hint,
                                     res ? reinterpret_cast<const char*>(res) + (long)n0 * Hd * Wd * Cd * des : nullptr, relu, bnf);
        if (rc) return rc;
        row0 += rows;
    }
    return DBN_OK;
}

// General form.  at: activation storage type of src / dst (DBN_AT_*; 16-bit storage needs ns = 1 and Cs % 16 == 0, panels from
// dbn_pack_weights_t with kind 1 (bf16) / 2 (fp16)).  ns: matrix math (0, 1, 3).  ksplit > 1: split-K with `slab` scratch.
int dbn_igemm_t(int at, int ns, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ksplit, float* slab,
                void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream, nullptr,
                     ksplit < 1 ? 1 : ksplit, slab, 0, at);
}

// Inference form (round 5): dst = [relu]( conv(src) + bias [+ res] ) in ONE launch — what an eval-mode conv -> BatchNorm -> (+ residual) ->
// ReLU chain (/root/reference/src/modules/basic.py:32-36, resnet.py:70-91 in model.eval()) becomes once the BatchNorm's running
// statistics are folded into the weights (dbn_fold_bn_eval: w' = w * gamma / sqrt(var + eps), bias' = beta + (bias - mean) * that): no
// separate BatchNorm pass over the activation.  res: NULL or a tensor of dst's shape and storage type; mode 0 (any stride) or mode 1
// with stride 1; no split-K.
int dbn_igemm_act_t(int at, int ns, const void* src, const float* wpk, const float* bias, const void* res, int relu, void* dst, int N,
                    int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride, int pad, int mode, int tile_hint,
                    void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, 0, tile_hint, ns, stream, nullptr, 1, nullptr, 0,
                     at, nullptr, res, relu ? 1 : 0);
}

int dbn_igemm_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                  int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, 0, stream);
}

// Rows of BatchNorm partials a conv with this output shape produces (see dbn_conv_bn_f32); follows igemm_run's chunking
// (conv_bn: the call is a dbn_conv_bn_t — no BatchNorm-backward sums — with this `accumulate`: it may take the ConvT kernel)
static int bn_tile_rows(int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int mode, int stride, int tile_hint, int at, int ns, int R,
                        int S, int pad, bool conv_bn = false, int accumulate = 0) {
    const int nmax = chunk_images(N, (long)Hd * Wd, (long)Hs * Ws * Cs * dbn_esize(at), (long)Hd * Wd * Cd);
    if (nmax < 1) return 0;
    const bool convt = conv_bn && convt_eligible(mode, ns, at, tile_hint, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, Cd, accumulate, 1);
    const int cfg = resolve_cfg((int)std::min<long>((long)N * Hd * Wd, 0x7FFFFFFF), Cd, tile_hint, at, ns, mode, R, S, stride, pad, Hs, Ws,
                                Hd, Wd, Cs, 1);
    int rows = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int n = std::min(nmax, N - n0);
        rows += (convt && convt_launch_ok(n, Hd, Wd, Cd)) ? dbn_convt_f32_rows(n * Hs * Ws) : bn_tile_rows_one(n, Hd, Wd, mode, stride, cfg);
    }
    return rows;
}

// Rows of per-tile partials a dbn_conv_bn_t / dbn_igemm_bnsums_t call with these arguments writes (its `part` array is
// [2][Cd][rows] floats; dbn_bn_backward_t takes it as `sums` with sums_parts = rows).
int dbn_igemm_bn_rows(int at, int ns, int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride, int pad, int mode,
                      int tile_hint) {
    return bn_tile_rows(N, Hs, Ws, Cs, Hd, Wd, Cd, mode, stride, tile_hint, at, ns, R, S, pad);
}

// dbn_igemm_t (no split-K) whose epilogue ALSO reduces, per channel and output tile, the two sums of the BatchNorm backward that
// consumes dst: dst is the gradient of the (ReLU'd) output of a BatchNorm with input y (same shape / storage as dst), saved
// mean / rstd, and ReLU mask either `zmask` > 0 (a saved activation of that shape) or fma(y, mask_scale, mask_shift) > 0 (the
// BatchNorm's own output recomputed).  The sums are taken over the FINAL dst values (after `accumulate`), so the call must be
// the last writer of dst.  part: [2][Cd][dbn_igemm_bn_rows(...)] floats.  Replaces the reduce pass of dbn_bn_backward_t (which
// re-reads dst and y): pass `part` as its `sums`.  fp32 tensors (at = 0, any matrix math) and bf16 tensors (at = 1): in bf16
// storage the sums are taken over the values as stored (rounded), y / zmask are bf16 like dst.
// y2 / save_mean2 / save_rstd2 / part2 (optional, with zmask): a second BatchNorm that consumes the same dst under the same mask.
int dbn_igemm_bnsums_t(int at, int ns, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                       int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, const void* y,
                       const void* zmask, const float* mask_scale, const float* mask_shift, const float* save_mean,
                       const float* save_rstd, float* part, const void* y2, const float* save_mean2, const float* save_rstd2,
                       float* part2, const dbn_bnb_final* fin, void* stream) {
    DBN_REQUIRE(((at == 0 && (ns == 0 || ns == 1 || ns == 3)) || (at == 1 && ns == 1)) && y && save_mean && save_rstd && part &&
                (zmask || (mask_scale && mask_shift)));
    DBN_REQUIRE(!y2 || (zmask && save_mean2 && save_rstd2 && part2));
    DBN_REQUIRE(!fin || (fin->counters && fin->group && fin->c1c2 && fin->dgamma && fin->dbeta &&
                         (!y2 || (fin->c1c2_2 && fin->dgamma_2 && fin->dbeta_2))));
    const int rows = bn_tile_rows(N, Hs, Ws, Cs, Hd, Wd, Cd, mode, stride, tile_hint, at, ns, R, S, pad);
    DBN_REQUIRE(rows > 0);
    const IgemmBnb b{y, zmask, mask_scale, mask_shift, save_mean, save_rstd, part, y2, save_mean2, save_rstd2, part2, fin, N * Hd * Wd};
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream, nullptr, 1,
                     nullptr, rows, at, &b);
}

// sizes of the in-kernel finalize's scratch (dbn_bnb_final) for `rows` partial rows and Cd channels
long dbn_igemm_bn_final_counters(int rows, int Cd) { return (long)(Cd / 64) * ((rows + 63) / 64 + 1); }
long dbn_igemm_bn_final_group_floats(int rows, int Cd) { return 4L * Cd * ((rows + 63) / 64); }

// floats of scratch for the fused conv + BatchNorm-statistics call
long dbn_conv_bn_ws_floats(int N, int Hd, int Wd, int Cd, int mode, int stride) {
    // worst case over the tile configurations and the image chunking (one launch per image: every launch rounds up)
    long worst = 0;
    for (int t = 1; t <= 4; ++t) {
        const long r = (long)N * bn_tile_rows_one(1, Hd, Wd, mode, stride, t);
        worst = r > worst ? r : worst;
    }
    return (3L * Cd + 1) * worst;
}

// Convolution (dbn_igemm_f32 / _bf16s contract; ns = 0, 1, 3) whose epilogue also accumulates the train-mode
// BatchNorm statistics of its output, followed by the finalize kernel: replaces conv -> separate statistics pass.
// Outputs like dbn_bn_train_stats.  ws: dbn_conv_bn_ws_floats(...) floats.
int dbn_conv_bn_t(int at, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                  int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                  const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                  float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    DBN_REQUIRE(gamma && beta && scale && shift && save_mean && save_rstd && ws);
    const int rows = bn_tile_rows(N, Hs, Ws, Cs, Hd, Wd, Cd, mode, stride, tile_hint, at, ns, R, S, pad, true, accumulate);
    DBN_REQUIRE(rows > 0);
    // the in-kernel finalize, when the caller announced it (dbn_conv_bn_set_final) and the launch has that epilogue (every igemm_f32_kernel /
    // conv3x3_wres16_kernel launch; not the 2x2 ConvTranspose kernel)
    int* const fcnt = g_bnf_cnt;
    double* const fgrp = g_bnf_grp;
    g_bnf_cnt = nullptr;
    g_bnf_grp = nullptr;
    const bool fin = fcnt && !convt_eligible(mode, ns, at, tile_hint, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, Cd, accumulate, 1);
    const BnFin f{fcnt, fgrp, gamma, beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd};
    // (16-bit storage: the statistics are those of the fp32 accumulators, i.e. of the values BEFORE they are rounded for storage)
    const int rc = igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream, ws,
                             1, nullptr, rows, at, nullptr, nullptr, 0, fin ? &f : nullptr);
    if (rc || fin) return rc;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(Cd), dim3(rows >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, ws, rows, Cd, gamma,
                       beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
}

// ---- 3x3 / stride 1 / pad 1 forward convolution through Winograd F(2x2, 3x3) in fp32 (winograd_f32.hip): 2.25x fewer MFMA FLOPs
// than the direct form.  fp32 NHWC tensors, H % 8 == 0, W % 16 == 0, Cs % 16 == 0 (the source's channel count; I <= Cs real input
// channels), Cd % 64 == 0, tensors below 3.75 GB; upanel from dbn_winograd_pack (dbn_winograd_panel_floats floats: the filters
// transformed once per parameter update).  gamma != NULL: the train-mode BatchNorm that follows is folded in as in dbn_conv_bn_t
// (ws: dbn_winograd_ws_floats floats).  Not bit-identical to the direct convolution (another fp32 summation: ~1e-6 relative).
// (any map size runs: right / bottom patches may be ragged — their pixels past the map are masked — but the kernel only pays when most
// of the 8 x 16 patches are real: >= 3/4 of the patch area, e.g. 40 x 40 maps (40 x 48 computed) yes, 20 x 20 (24 x 32) no)
int dbn_winograd_eligible(int N, int H, int W, int Cs, int Cd) {
    if (!(N > 0 && H > 0 && W > 0 && Cs > 0 && Cs % 16 == 0 && Cd > 0 && Cd % 64 == 0)) return 0;
    const long Hp = (H + 7) / 8 * 8, Wp = (W + 15) / 16 * 16;
    // small maps (the band of 32 consecutive tiles fits the LDS patch) run in the consecutive-tile form: no ragged patches at all
    const long tiles = (long)((H + 1) / 2) * ((W + 1) / 2);
    const bool pays = dbn_winograd_linear(H, W) ? 5 * tiles >= 3 * ((tiles + 31) / 32 * 32) : 4L * H * W >= 3L * Hp * Wp;  // (>= 60 % / 75 % real work)
    return pays && (long)N * H * W * std::max(Cs, Cd) * 4 < dbn_g_byte_limit && (long)N * Hp * Wp < dbn_g_pixel_limit;
}
long dbn_winograd_panel_floats(int O, int Cs) { return (long)Cs * 16 * O; }
// dgrad = 0: panel of the forward conv of w [O][I][3][3] over a source with Cs >= I channels.  dgrad = 1: panel of the DATA GRADIENT of
// the conv with weights w [I][O][3][3] (w's own output channels are this conv's I input channels, its input channels the O outputs):
// the same kernel then maps dy (Cs >= I channels) to dx (O channels) — filters rotated by 180 degrees, channel roles swapped.
int dbn_winograd_pack(const float* w_oihw, int O, int I, int Cs, int dgrad, float* out, void* stream) {
    DBN_REQUIRE(w_oihw && out && O > 0 && I > 0 && I <= Cs && Cs % 16 == 0 && O % 64 == 0 && (dgrad == 0 || dgrad == 1));
    return dbn_launch_winograd_pack(w_oihw, O, I, Cs, dgrad, out, (hipStream_t)stream);
}
// n dbn_winograd_pack calls in one launch.  jobs: DEVICE array of n records { const float* w; float* out; int O, I, Cs, dgrad; } (32 bytes)
int dbn_winograd_pack_batched(const void* jobs, int n, void* stream) {
    DBN_REQUIRE(jobs && n > 0);
    return dbn_launch_winograd_pack_many(jobs, n, (hipStream_t)stream);
}
long dbn_winograd_ws_floats(int N, int H, int W, int Cd) { return (3L * Cd + 1) * dbn_winograd_rows(N, H, W); }
// ... _act_: the conv's input is relu(src * in_scale[c] + in_shift[c]) ([Cs] floats each; NULL, NULL: src itself) — src is the INPUT of
// the BatchNorm + ReLU in front of the conv (basic.py:32-36 / resnet.py:77-80), applied while the patch is staged with bn_apply's own
// arithmetic, so that activation tensor is never written or re-read (bit-identical to bn_apply followed by dbn_winograd_conv_bn_f32).
int dbn_winograd_conv_bn_act_f32(const float* src, const float* in_scale, const float* in_shift, const float* upanel, const float* bias,
                                 float* dst, int N, int H, int W, int Cs, int Cd, const float* gamma, const float* beta, float eps,
                                 float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean,
                                 float* save_rstd, float* ws, void* stream) {
    DBN_REQUIRE(src && upanel && dst && dbn_winograd_eligible(N, H, W, Cs, Cd) && !in_scale == !in_shift);
    DBN_REQUIRE(!gamma || (beta && scale && shift && save_mean && save_rstd && ws));
    IgemmParams p{};
    p.in_scale = in_scale; p.in_shift = in_shift;
    p.src = src; p.wpk = upanel; p.bias = bias; p.dst = dst;
    p.N = N; p.Hs = H; p.Ws = W; p.Cs = Cs; p.Cd = Cd; p.Hdf = H; p.Wdf = W; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1;
    p.src_bytes = (unsigned)((long)N * H * W * Cs * 4);
    const int rows = dbn_winograd_rows(N, H, W);
    p.stats = gamma ? ws : nullptr; p.stat_rows = rows; p.stat_row0 = 0;
    int* const fcnt = g_bnf_cnt;
    double* const fgrp = g_bnf_grp;
    g_bnf_cnt = nullptr;
    g_bnf_grp = nullptr;
    const bool fin = gamma && fcnt;
    if (fin) {  // the statistics rows are folded by the kernel's own last workgroups (dbn_conv_bn_set_final)
        p.bnf_cnt = fcnt; p.bnf_grp = fgrp; p.bnf_gamma = gamma; p.bnf_beta = beta; p.bnf_eps = eps; p.bnf_momentum = momentum;
        p.bnf_run_mean = run_mean; p.bnf_run_var = run_var; p.bnf_scale = scale; p.bnf_shift = shift; p.bnf_mean = save_mean; p.bnf_rstd = save_rstd;
    }
    const int rc = dbn_launch_winograd_f32(p, (hipStream_t)stream);
    if (rc || !gamma || fin) return rc;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(Cd), dim3(rows >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, ws, rows, Cd, gamma,
                       beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
}

// Inference form of the Winograd conv: dst = [relu]( conv3x3(src) + bias [+ res] ) (see dbn_igemm_act_t)
int dbn_winograd_conv_act_f32(const float* src, const float* upanel, const float* bias, const float* res, int relu, float* dst, int N,
                              int H, int W, int Cs, int Cd, void* stream) {
    DBN_REQUIRE(src && upanel && dst && dbn_winograd_eligible(N, H, W, Cs, Cd));
    IgemmParams p{};
    p.src = src; p.wpk = upanel; p.bias = bias; p.dst = dst; p.res = res; p.relu = relu ? 1 : 0; p.accumulate = res ? 1 : 0;
    p.N = N; p.Hs = H; p.Ws = W; p.Cs = Cs; p.Cd = Cd; p.Hdf = H; p.Wdf = W; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1;
    p.src_bytes = (unsigned)((long)N * H * W * Cs * 4);
    p.stat_rows = dbn_winograd_rows(N, H, W);
    return dbn_launch_winograd_f32(p, (hipStream_t)stream);
}

int dbn_winograd_conv_bn_f32(const float* src, const float* upanel, const float* bias, float* dst, int N, int H, int W, int Cs, int Cd,
                             const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                             float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    return dbn_winograd_conv_bn_act_f32(src, nullptr, nullptr, upanel, bias, dst, N, H, W, Cs, Cd, gamma, beta, eps, momentum, run_mean, run_var,
                                        scale, shift, save_mean, save_rstd, ws, stream);
}

// The data gradient of a 3x3 / stride-1 / pad-1 conv through the same kernel (upanel from dbn_winograd_pack(..., dgrad = 1)): dx [N,H,W,Cd]
// = [dx +] conv(dy [N,H,W,Cs]).  y non-NULL: the epilogue also produces the two per-channel sums of the BatchNorm backward that consumes
// dx — arguments and semantics of dbn_igemm_bnsums_t (part: [2][Cd][dbn_winograd_rows(N,H,W)] floats; fin: optional in-kernel finalize,
// counters dbn_igemm_bn_final_counters(rows, Cd) ints as there).
int dbn_winograd_dgrad_bnsums_f32(const float* dy, const float* upanel, float* dx, int N, int H, int W, int Cs, int Cd, int accumulate,
                                  const void* y, const void* zmask, const float* mask_scale, const float* mask_shift,
                                  const float* save_mean, const float* save_rstd, float* part, const void* y2, const float* save_mean2,
                                  const float* save_rstd2, float* part2, const dbn_bnb_final* fin, void* stream) {
    DBN_REQUIRE(dy && upanel && dx && dbn_winograd_eligible(N, H, W, Cs, Cd));
    DBN_REQUIRE(!y || (save_mean && save_rstd && part && (zmask || (mask_scale && mask_shift))));
    DBN_REQUIRE(!y2 || (y && zmask && save_mean2 && save_rstd2 && part2));
    DBN_REQUIRE(!fin || (y && fin->counters && fin->group && fin->c1c2 && fin->dgamma && fin->dbeta &&
                         (!y2 || (fin->c1c2_2 && fin->dgamma_2 && fin->dbeta_2))));
    IgemmParams p{};
    p.src = dy; p.wpk = upanel; p.bias = nullptr; p.dst = dx;
    p.N = N; p.Hs = H; p.Ws = W; p.Cs = Cs; p.Cd = Cd; p.Hdf = H; p.Wdf = W; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1;
    p.accumulate = accumulate;
    p.src_bytes = (unsigned)((long)N * H * W * Cs * 4);
    p.stat_rows = dbn_winograd_rows(N, H, W); p.stat_row0 = 0;
    if (y) {
        p.bnb_y = y; p.bnb_zmask = zmask; p.bnb_msc = mask_scale; p.bnb_msh = mask_shift; p.bnb_mean = save_mean; p.bnb_rstd = save_rstd;
        p.bnb_part = part; p.bnb_y2 = y2; p.bnb_mean2 = save_mean2; p.bnb_rstd2 = save_rstd2; p.bnb_part2 = part2;
        if (fin) {
            p.bnb_cnt = fin->counters; p.bnb_grp = fin->group; p.bnb_gscale = fin->grad_scale; p.bnb_invM = 1.0f / (float)((long)N * H * W);
            p.bnb_c1c2[0] = fin->c1c2; p.bnb_dgamma[0] = fin->dgamma; p.bnb_dbeta[0] = fin->dbeta;
            p.bnb_c1c2[1] = fin->c1c2_2; p.bnb_dgamma[1] = fin->dgamma_2; p.bnb_dbeta[1] = fin->dbeta_2;
        }
    }
    return dbn_launch_winograd_f32(p, (hipStream_t)stream);
}

int dbn_conv_bn_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                    int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                    const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                    float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    return dbn_conv_bn_t(0, src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, gamma, beta,
                         eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream);
}

// Same contract as dbn_igemm_f32 with the products evaluated on the bf16 matrix pipe: ns = 3 fp32-accurate
// three-way operand split (6 bf16 MFMAs per product group), ns = 1 plain bf16 operands.  Panels from
// dbn_pack_weights_bf16s with the same (mode, stride, ns).
int dbn_igemm_bf16s(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                    int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                    void* stream) {
    DBN_REQUIRE(ns == 1 || ns == 3);
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream);
}

// Split-K plan for a conv whose output grid alone cannot fill the chip (few pixels x few channels, long reduction —
// the coarse FPN levels' data gradients): number of K splits (1 = none) for M rows, Cd channels, K = R*S*Cs.
int dbn_igemm_splitk_plan_ns(int M, int Cd, int K, int Cs, int ns) {
    if (Cs % 16 != 0) return 1;
    const int cfg0 = tile_config_for(M, Cd, ns);
    const int cfg = (cfg0 == 1 && Cd % 128 != 0) ? 3 : cfg0;
    static const int bm_of[5] = {0, 128, 256, 128, 64}, bn_of[5] = {0, 128, 64, 64, 64};
    const long tiles = (long)dbn_ceil_div(M, bm_of[cfg]) * (Cd / bn_of[cfg]);
    const int KT = (K + 15) / 16;
    static const int max_tiles = dbn_env_int("DBN_SPLITK_MAX_TILES", 256);
    if (tiles > max_tiles) return 1;
#ifndef DBN_SPLITK_TARGET16
#define DBN_SPLITK_TARGET16 256  // round 6: the 16-bit loop's stages are latency-bound (~1 us each whatever their size), so more splits per CU only add stages and slab
#endif                           // traffic: pyramid level 2 / 3 data gradients 172 / 163 us at 5 / 10 splits, 143 / 142 us at 1 / 2 (tools/trace_probe_levels.py); bf16 step 1798 -> 1811
#ifndef DBN_SPLITK_TARGET32
#define DBN_SPLITK_TARGET32 1024  // (512 / 2048 measured equal on the fp32 step: 731.6 / 734.0 and 732.6 / 730.4 against 731.8 / 732.7 images/s)
#endif
    long sk = (ns == 1 ? DBN_SPLITK_TARGET16 : DBN_SPLITK_TARGET32) / tiles;      // exact fp32: about four workgroups per CU; one 16-bit plane: about one
    if (sk > KT / 32) sk = KT / 32;  // at least 32 k-tiles per split
    if (sk > 64) sk = 64;
    return sk < 2 ? 1 : (int)sk;
}
int dbn_igemm_splitk_plan(int M, int Cd, int K, int Cs) { return dbn_igemm_splitk_plan_ns(M, Cd, K, Cs, 0); }

// dbn_igemm_f32 with the reduction split `ksplit` ways (mode 0, or mode 1 with stride 1; Cs % 16 == 0).
// slab: dbn_igemm_splitk_slab_floats(ksplit, N, Hd, Wd, Cd) = ksplit * (N*Hd*Wd*Cd + 1088) floats of scratch (the slabs lie 1088 floats
// further apart than their size: HBM channel rotation).  Bit-reproducible (fixed summation order).
int dbn_igemm_splitk_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                         int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                         int ksplit, float* slab, void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream,
                     nullptr, ksplit, slab);
}

long dbn_igemm_splitk_slab_floats(int ksplit, int N, int Hd, int Wd, int Cd) { return (long)ksplit * ((long)N * Hd * Wd * Cd + 1088); }

int dbn_igemm_packed_floats(int K, int Cd) { return ((K + 15) / 16) * 16 * Cd; }

// ---- pyramid conv (MODE 3): conv3x3 over [s0 | up2(s1) | up4(s2) | up8(s3)] without the concatenation ----
// the 128 x 256 tile of the 16-bit storage types (igemm_kernel.h launch_wide): 0 off, 1 the pyramid conv (Cd % 256 == 0, large maps), 2 also
// plain forward / stride-1 data-gradient launches of the generic loop, 3 as 2 whatever the launch's size (tests)
int dbn_g_wide_tile = getenv("DBN_PYR_WIDE") ? atoi(getenv("DBN_PYR_WIDE")) : 1;  // (A/B runs of whole programs)
int dbn_g_pyr_group = getenv("DBN_PYR_GROUP") ? atoi(getenv("DBN_PYR_GROUP")) : 0;  // tile order of the wide pyramid tile (igemm_kernel.h launch_wide)
// Does a dbn_pyramid_conv_* call (16-bit storage, at = 1 | 2) with this geometry launch igemm_f32_kernel<128,256,2,2,3,1,at>?  (profiler labels)
int dbn_pyramid_wide_would_run(int at, int N, int H, int W, int Cs, int Cd) {
    return (at == 1 || at == 2) && dbn_wide_tile_geom_ok(3, N, H, W, Cs, Cd);
}
int dbn_set_pyramid_wide(int on) {  // test / A-B hook; returns the previous setting
    const int old = dbn_g_wide_tile;
    dbn_g_wide_tile = on < 0 ? 0 : on > 3 ? 3 : on;
    return old;
}
static int pyramid_chunk(int N, int H, int W, int Cs, int Cd, int at = 0) {
    return chunk_images(N, (long)H * W, (long)H * W * Cs * dbn_esize(at), (long)H * W * Cd);
}
static int pyramid_rows_one(int n, int H, int W) { return 64 * dbn_ceil_div((long)n * (H >> 3) * (W >> 3), 128); }

long dbn_pyramid_conv_ws_floats(int N, int H, int W, int Cd) { return (3L * Cd + 1) * N * pyramid_rows_one(1, H, W); }

// first_level = 1 (exact fp32, and — round 5 — 16-bit storage): dst already holds level 0's part of the sum (the plain 3x3 conv of s0, bias included — e.g. from
// dbn_winograd_conv_bn_f32); the launch adds levels 1-3 to it (s0, w0, bias are not read).
static int pyramid_run(int first_level, int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0,
                       const float* w1, const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs,
                       int Cd, int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum,
                       float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws,
                       void* stream, int relu) {
    DBN_REQUIRE(first_level == 0 || (first_level == 1 && ((at == 0 && ns == 0) || ((at == 1 || at == 2) && ns == 1))));
    DBN_REQUIRE(!relu || !gamma);  // (the inference epilogue and the train-mode statistics exclude each other)
    if (first_level) { s0 = s1; w0 = w1; bias = nullptr; }
    DBN_REQUIRE(s0 && s1 && s2 && s3 && w0 && w1 && w2 && w3 && dst && (ns == 0 || ns == 1 || ns == 3));
    DBN_REQUIRE(at == 0 || ((at == 1 || at == 2) && ns == 1) || (at == 3 && ns == 3));
    const int es = dbn_esize(at), des = dst_esize(at);
    DBN_REQUIRE(N > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && Cs % 16 == 0 && Cd % 128 == 0);
    DBN_REQUIRE(tile_hint == 0 || tile_hint == 1);
    const bool bn = gamma != nullptr;
    DBN_REQUIRE(!bn || (beta && scale && shift && save_mean && save_rstd && ws));
    const int nmax = pyramid_chunk(N, H, W, Cs, Cd, at);  // images per launch (24-bit pixel indices, 32-bit offsets)
    DBN_REQUIRE(nmax >= 1);
    int rows_total = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) rows_total += pyramid_rows_one(std::min(nmax, N - n0), H, W);
    hipStream_t st = (hipStream_t)stream;
    const char* srcs[4] = {(const char*)s0, (const char*)s1, (const char*)s2, (const char*)s3};
    const float* wpks[4] = {w0, w1, w2, w3};
    int row0 = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int n = std::min(nmax, N - n0);
        IgemmParams p{};
        for (int g = 0; g < 4; ++g) {
            p.seg_src[g] = srcs[g] + (long)n0 * (H >> g) * (W >> g) * Cs * es;
            p.seg_wpk[g] = wpks[g];
            const long pl = at == 3 ? (long)N * (H >> g) * (W >> g) * Cs * 2 : 0;  // plane distance of level g (whole tensor)
            p.seg_plane_bytes[g] = (unsigned)pl;
            p.seg_bytes[g] = (unsigned)((long)n * (H >> g) * (W >> g) * Cs * es + 2 * pl);
        }
        p.plane_bytes = p.seg_plane_bytes[0];
        p.src = p.seg_src[0]; p.wpk = w0; p.bias = bias; p.dst = (char*)dst + (long)n0 * H * W * Cd * des;
        p.N = n; p.Hs = H; p.Ws = W; p.Cs = Cs; p.Cd = Cd; p.Hdf = H; p.Wdf = W;
        p.R = 3; p.S = 3; p.stride = 8; p.pad = 1; p.accumulate = first_level ? 1 : 0; p.ncls = 64;
        p.first_level = first_level;
        p.bnb_y = p.bnb_zmask = p.bnb_y2 = nullptr; p.bnb_msc = p.bnb_msh = p.bnb_mean = p.bnb_rstd = p.bnb_mean2 = p.bnb_rstd2 = nullptr;
        p.bnb_part = p.bnb_part2 = nullptr;
        p.bnb_cnt = nullptr; p.bnb_grp = nullptr; p.bnb_gscale = 1.f; p.bnb_invM = 0.f;
        for (int b_ = 0; b_ < 2; ++b_) p.bnb_c1c2[b_] = p.bnb_dgamma[b_] = p.bnb_dbeta[b_] = nullptr;
        p.stats = bn ? ws : nullptr;
        p.stat_rows = rows_total; p.stat_row0 = row0; p.launch_rows = 0;
        p.ksplit = 1; p.kt_per = 0; p.patch = 0;
        p.relu = relu ? 1 : 0;
        p.src_bytes = p.seg_bytes[0];
        const int rc = launch_igemm(p, 1, 3, ns, st, at);  // (16-bit storage, Cd % 256 == 0, large maps: the 128 x 256 tile — launch_wide)
        if (rc) return rc;
        row0 += p.launch_rows;
    }
    if (!bn) return DBN_OK;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(Cd), dim3(rows_total >= 2048 ? 1024 : 256), 0, st, ws, rows_total, Cd, gamma,
                       beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
}

int dbn_pyramid_conv_from_t(int first_level, int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0,
                            const float* w1, const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs,
                            int Cd, int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum,
                            float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws,
                            void* stream) {
    return pyramid_run(first_level, at, s0, s1, s2, s3, w0, w1, w2, w3, bias, dst, N, H, W, Cs, Cd, tile_hint, ns, gamma, beta, eps, momentum,
                       run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream, 0);
}

// Inference form of the pyramid conv (see dbn_igemm_act_t): dst = [relu]( [dst +] pyramid(levels first_level..3) + bias ), no statistics.
int dbn_pyramid_conv_act_t(int first_level, int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0,
                           const float* w1, const float* w2, const float* w3, const float* bias, int relu, void* dst, int N, int H, int W,
                           int Cs, int Cd, int ns, void* stream) {
    return pyramid_run(first_level, at, s0, s1, s2, s3, w0, w1, w2, w3, bias, dst, N, H, W, Cs, Cd, 0, ns, nullptr, nullptr, 0.f, 0.f, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream, relu);
}

int dbn_pyramid_conv_t(int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0, const float* w1,
                       const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs, int Cd,
                       int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                       float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    return dbn_pyramid_conv_from_t(0, at, s0, s1, s2, s3, w0, w1, w2, w3, bias, dst, N, H, W, Cs, Cd, tile_hint, ns, gamma, beta, eps, momentum,
                                   run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream);
}

int dbn_pyramid_conv_f32(const float* s0, const float* s1, const float* s2, const float* s3, const float* w0, const float* w1,
                         const float* w2, const float* w3, const float* bias, float* dst, int N, int H, int W, int Cs, int Cd,
                         int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                         float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    return dbn_pyramid_conv_t(0, s0, s1, s2, s3, w0, w1, w2, w3, bias, dst, N, H, W, Cs, Cd, tile_hint, ns, gamma, beta, eps, momentum,
                              run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream);
}

// Test hook: lower the per-launch index ranges so that the image chunking runs at small sizes (0 restores a default).
int dbn_set_index_limits(long pixel_rows, long bytes, long elems) {
    dbn_g_pixel_limit = pixel_rows > 0 ? pixel_rows : (1L << 24);
    dbn_g_byte_limit = bytes > 0 ? bytes : 0xF0000000L;
    dbn_g_elem_limit = elems > 0 ? elems : (1L << 32);
    return DBN_OK;
}

}  // extern "C"
