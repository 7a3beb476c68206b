// Weight-gradient entry points of the C ABI: tile / pixel-split choice, image chunking, the slab reduction kernels.  The matrix
// kernels live in wgrad_kernels.h (instantiated by wgrad_f32.hip and wgrad_b16.hip).
//
// Replaces the autograd weight gradients of the Conv2d / ConvTranspose2d calls under /root/reference/src/modules/resnet.py:70-91,
// modules/basic.py:32-36, modules/segmentation_body.py:64-77, modules/segmentation_head.py:24-29,64-79.
#include "wgrad_kernels.h"

namespace {

// (the stem: 64 x 196 gradient elements and hundreds of pixel splits — a thread that walked all the splits of its positions was a chain of
// dependent loads, 51 us alone at the end of the step; now WRG thread groups share the splits of 16 position quads and are combined in group
// order through LDS: fixed summation order, bit-reproducible)
constexpr int WRG = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splitk, int O, int J, int Jp, int BM, int BN,
                                                           int Cb, int I, int R, int S, float* __restrict__ grad, float scale, int natural) {
    // one thread: 4 consecutive slab positions (one b128 load per split) of the splits g, g + WRG, ...; 4 loads in flight
    __shared__ double part[WRG][16][4];
    const long total = (long)O * Jp, count4 = total >> 2, total4 = wgrad_slab_stride(O, Jp) >> 2;
    const int pos = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long q = blockIdx.x * 16L + pos;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (q < count4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slab) + q;
        int z = g;
        for (; z + 3 * WRG < splitk; z += 4 * WRG) {
            const f32x4 v0 = src[(long)z * total4], v1 = src[(long)(z + WRG) * total4];
            const f32x4 v2 = src[(long)(z + 2 * WRG) * total4], v3 = src[(long)(z + 3 * WRG) * total4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += ((double)v0[e] + (double)v1[e]) + ((double)v2[e] + (double)v3[e]);
        }
        for (; z < splitk; z += WRG) {
            const f32x4 v = src[(long)z * total4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[g][pos][e] = s[e];
    __syncthreads();
    if (g != 0 || q >= count4) return;
#pragma unroll
    for (int k = 1; k < WRG; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += part[k][pos][e];
    const long idx = q << 2;
    const int prow = (int)(idx / Jp);
    const int pcol0 = (int)(idx - (long)prow * Jp);
    const int o = natural ? prow : (prow / BM) * BM + tile_pos_to_index(prow % BM, BM);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int pcol = pcol0 + e;
        const int j = natural ? pcol : (pcol / BN) * BN + tile_pos_to_index(pcol % BN, BN);
        if (j >= J) continue;
        const int tap = j / Cb, i = j - tap * Cb;
        if (i >= I) continue;
        grad[((long)o * I + i) * (R * S) + tap] = (float)(s[e] * scale);
    }
}

// The same reduction for layers whose input-channel count is a multiple of 64 (every layer but the stem): one workgroup
// per (output channel o, block of 64 input channels), so that BOTH sides are coalesced — the slab is read in 64-byte runs
// (a b128 per lane: four positions = channels 4u+e of one tap) and the OIHW gradient leaves as one contiguous run of
// 64*R*S floats staged through LDS (the per-element kernel above scatters 4-byte stores at a stride of R*S floats).
// Layers with few output elements and hundreds of pixel splits (64-channel layers at 160^2: 3 tiles x 340 splits) are
// latency-bound on the chain of split loads: G thread groups share the splits (group g takes splits g, g+G, ...), their
// fp64 partial sums are combined through LDS in group order — fixed summation order, bit-reproducible.
__device__ __forceinline__ void wgrad_reduce64_body(double* dsm, int o, int iy, const float* __restrict__ slab, int splitk, int O, int J,
                                                    int Jp, int BM, int BN, int Cb, int I, int RS, int G, float* __restrict__ grad,
                                                    float scale, int natural) {
    const int items = RS * 16;
    float* stage = reinterpret_cast<float*>(dsm + (G > 1 ? (size_t)G * items * 4 : 0));
    const int i0 = iy * 64;
    const int om = o % BM;
    const int prow = natural ? o : (o / BM) * BM + (om & 3) * (BM / 4) + (om >> 2);  // inverse of tile_pos_to_index
    const long total4 = wgrad_slab_stride(O, Jp) >> 2;
    const int nthr = blockDim.x;
    for (int w = threadIdx.x; w < items * G; w += nthr) {
        const int g = w / items, t = w - g * items;
        const int tap = t >> 4, e = (t >> 2) & 3, cq = t & 3;
        const int j0 = tap * Cb + i0;                 // multiple of 64: the 64 channels lie inside one BN-wide tile
        const int jt = j0 / BN, jl0 = j0 - jt * BN;
        // position space: positions pos..pos+3 hold channels i0 + 16cq + 4u + e; natural order: lane (e, cq) takes the four
        // consecutive channels i0 + 4*(4e + cq) + u
        const int pos = natural ? jl0 + 4 * (4 * e + cq) : e * (BN / 4) + (jl0 >> 2) + 4 * cq;
        const f32x4* src = reinterpret_cast<const f32x4*>(slab + (long)prow * Jp + jt * BN + pos);
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        int z = g;
        for (; z + 3 * G < splitk; z += 4 * G) {
            const f32x4 v0 = src[(long)z * total4], v1 = src[(long)(z + G) * total4];
            const f32x4 v2 = src[(long)(z + 2 * G) * total4], v3 = src[(long)(z + 3 * G) * total4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += ((double)v0[u] + (double)v1[u]) + ((double)v2[u] + (double)v3[u]);
        }
        for (; z < splitk; z += G) {
            const f32x4 v = src[(long)z * total4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += (double)v[u];
        }
        if (G > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) dsm[((size_t)g * items + t) * 4 + u] = s[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) stage[(natural ? 4 * (4 * e + cq) + u : 16 * cq + 4 * u + e) * RS + tap] = (float)(s[u] * scale);
        }
    }
    if (G > 1) {
        __syncthreads();
        for (int t = threadIdx.x; t < items; t += nthr) {
            const int tap = t >> 4, e = (t >> 2) & 3, cq = t & 3;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += dsm[((size_t)g * items + t) * 4 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) stage[(natural ? 4 * (4 * e + cq) + u : 16 * cq + 4 * u + e) * RS + tap] = (float)(s[u] * scale);
        }
    }
    __syncthreads();
    const int n = min(64, I - i0) * RS;  // channels >= I are padding of the activation tensor
    float* dst = grad + ((long)o * I + i0) * RS;
    for (int k = threadIdx.x; k < n; k += nthr) dst[k] = stage[k];
}

__global__ __launch_bounds__(1024) void wgrad_reduce64_kernel(const float* __restrict__ slab, int splitk, int O, int J, int Jp, int BM,
                                                              int BN, int Cb, int I, int RS, int G, float* __restrict__ grad,
                                                              float scale, int natural) {
    extern __shared__ double dsm[];  // [G][items][4] partial sums (G > 1), then the [64][RS] float staging image
    wgrad_reduce64_body(dsm, blockIdx.x, blockIdx.y, slab, splitk, O, J, Jp, BM, BN, Cb, I, RS, G, grad, scale, natural);
}

// The slab reductions of MANY layers in one launch (dbn_wgrad_reduce_many): the side stream of a training step otherwise carries one
// small reduction behind every weight-gradient kernel — 34 launches of 5-40 us that each hold the stream's next matrix kernel back
// until their last workgroup has drained (0.46 ms of work, 4.6 ms in flight at bs16 640^2).  Workgroup b looks its job up in the
// prefix table `first` (jobs' first workgroup; n_jobs + 1 entries) and runs wgrad_reduce64_kernel's body on it: same sums, same order.
struct WgradReduceJob {  // == dbn_wgrad_reduce_job (include/dbnet_hip.h)
    const float* slab;
    float* grad;
    int splitk, O, J, Jp, BM, BN, Cb, I, RS, G, natural, blocks;
    float scale;
    int smem_bytes;
};
__global__ __launch_bounds__(1024) void wgrad_reduce64_many_kernel(const WgradReduceJob* __restrict__ jobs, const int* __restrict__ first,
                                                                   int n_jobs) {
    extern __shared__ double dsm[];
    int lo = 0, hi = n_jobs - 1;  // the last job whose first workgroup is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const WgradReduceJob j = jobs[lo];
    const int local = blockIdx.x - first[lo], ny = j.Cb / 64;
    wgrad_reduce64_body(dsm, local / ny, local % ny, j.slab, j.splitk, j.O, j.J, j.Jp, j.BM, j.BN, j.Cb, j.I, j.RS, j.G, j.grad, j.scale,
                        j.natural);
}

}  // namespace

extern "C" {

static void wgrad_tiles(int O, int J, int& bm, int& bn) {
    static const int force192 = dbn_env_int("DBN_WGRAD_192", 1);  // 0: never, 1: 64-output-channel layers, 2: every layer with J % 192 == 0
    if (J % 192 == 0 && ((force192 == 1 && O == 64) || force192 == 2)) {
        // 64 x 192: J = taps*Cin of every 3x3 layer is a multiple of 192 (no padded columns), and the 64 + 192
        // staging threads are exactly the 4 waves of the workgroup
        bm = 64;
        bn = 192;
        return;
    }
    bm = (O % 128 == 0 && J >= 128) ? 128 : 64;
    bn = (J >= 128) ? 128 : 64;
}

// Pixel splits of ONE launch over n images
static int wgrad_splitk_one(int n, int Ho, int Wo, int O, int Cb, int R, int S) {
    const long P = (long)n * Ho * Wo;
    const int J = R * S * Cb;
    int bm, bn;
    wgrad_tiles(O, J, bm, bn);
    const long tiles = (long)(O / bm) * ((J + bn - 1) / bn);
    // splits such that tiles*splits fills whole rounds of the 256 CUs (k workgroups per CU, k = 4..2)
    long maxsk = (P + 511) / 512;  // at least 512 pixels per split
    if (maxsk < 1) maxsk = 1;
    long sk = 1;
    double best = -1.0;
    // with the two-stream step fewer, longer splits win (32.4 -> 32.2 ms); single stream: 0.02
    static const double prefer = dbn_env_double("DBN_WGRAD_PREFER", 0.0);
    static const int kmax = dbn_env_int("DBN_WGRAD_KMAX", 4);
    for (int k = kmax; k >= 2; --k) {
        long cand = (256L * k) / tiles;
        if (cand < 1) cand = 1;
        if (cand > maxsk) cand = maxsk;
        const long blocks = tiles * cand;
        const double util = (double)blocks / (double)(((blocks + 255) / 256) * 256);
        const double score = util + prefer * k;  // tie-break between 2..4 workgroups per CU (DBN_WGRAD_PREFER)
        if (score > best) {
            best = score;
            sk = cand;
        }
    }
    long pchunk = ((P + sk - 1) / sk + 15) / 16 * 16;
    return (int)((P + pchunk - 1) / pchunk);
}

// images per launch: 24-bit pixel indices, 32-bit byte offsets into dY and X
static int wgrad_chunk(int N, int Ho, int Wo, int O, int H, int W, int Cb, int at = 0) {
    long n = N;
    const long px = (long)Ho * Wo;
    if (px * n >= dbn_g_pixel_limit - 64) n = (dbn_g_pixel_limit - 65) / px;
    return (int)std::min<long>(n, chunk_images(N, 0, px * O * dbn_esize(at), 0, (long)H * W * Cb * dbn_esize(at)));
}

// Number of pixel splits (slabs) dbn_wgrad_f32 will use, over all its launches.
int dbn_wgrad_splitk_hw(int N, int Ho, int Wo, int O, int H, int W, int Cb, int R, int S) {
    const int nmax = wgrad_chunk(N, Ho, Wo, O, H, W, Cb);
    if (nmax < 1) return 0;
    int tot = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) tot += wgrad_splitk_one(std::min(nmax, N - n0), Ho, Wo, O, Cb, R, S);
    return tot;
}
// (input size unknown: assumes a stride <= 8 conv, X no larger than 64x dY's pixel count — only the chunking depends on it)
int dbn_wgrad_splitk(int N, int Ho, int Wo, int O, int Cb, int R, int S) { return dbn_wgrad_splitk_hw(N, Ho, Wo, O, Ho, Wo, Cb, R, S); }

// Floats of slab scratch dbn_wgrad_f32 / dbn_wgrad_t need: total splits * O * (R*S*Cb rounded up to the tile width).
// es: bytes per activation element (4, or 2 for bf16 storage) — the image chunking depends on the tensors' byte sizes.
long dbn_wgrad_slab_floats_hw(int N, int Ho, int Wo, int O, int H, int W, int Cb, int R, int S, int es) {
    const int J = R * S * Cb;
    int bm, bn;
    wgrad_tiles(O, J, bm, bn);
    const long Jp = (long)((J + bn - 1) / bn) * bn;
    const int nmax = wgrad_chunk(N, Ho, Wo, O, H, W, Cb, es == 2 ? 1 : 0);
    if (nmax < 1) return 0;
    long splits = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) splits += wgrad_splitk_one(std::min(nmax, N - n0), Ho, Wo, O, Cb, R, S);
    return splits * wgrad_slab_stride(O, Jp);
}
// (without the size of X: exact for calls that run as one launch — N*Ho*Wo < 2^24 pixels and X below 3.75 GB)
long dbn_wgrad_slab_floats(int N, int Ho, int Wo, int O, int Cb, int R, int S) {
    return dbn_wgrad_slab_floats_hw(N, Ho, Wo, O, Ho, Wo, Cb, R, S, 4);
}

// -1: read DBN_WGRAD_DMA on first use; 0: defaults (fp32 tensors: register-transposing kernel; bf16 tensors: LDS-DMA + transposing
// reads); 1: LDS-DMA kernel for exact-fp32 math on fp32 tensors; 2: register-transposing kernel for bf16 tensors too
// 3: as 0 with the general (per-pixel) gather addressing also where the k-tiles are whole row segments (A/B measurements, tests)
static int g_wgrad_variant = -1;
static int g_wgrad_row16 = 1;
int dbn_set_wgrad_variant(int v) {
    DBN_REQUIRE(v == 0 || v == 2 || v == 3 || (v == 1 && DBN_HAS_EXPERIMENTS));  // 1: the LDS-DMA kernel exists in -DDBN_EXPERIMENTS builds only
    g_wgrad_variant = v == 3 ? 0 : v;
    g_wgrad_row16 = v == 3 ? 0 : 1;
    return DBN_OK;
}

// Tile variant dbn_wgrad_* uses for O output channels and J = R*S*Cb columns: 1 = 64x192, 2 = 128x128, 3 = 64x128, 4 = 64x64
// (wgrad_f32_kernel<BM, BN, 2, 2, ns, at> in a rocprofv3 trace)
int dbn_wgrad_tile_config(int O, int J) {
    int bm, bn;
    wgrad_tiles(O, J, bm, bn);
    return bn == 192 ? 1 : (bm == 128 ? 2 : (bn == 128 ? 3 : 4));
}

// phases: 1 = the MFMA kernels (activations -> slabs), 2 = the slab reduction (slabs -> gradient), 3 = both
static bool wgrad_uses_patch(int at, int ns, int O, int Cb, int R, int S, int stride, int pad, int Ho, int Wo, int H, int W) {
    static const int env = dbn_env_int("DBN_WGRAD_PATCH", 1);
    return env && g_wgrad_variant != 2 && ((ns == 1 && (at == 0 || at == 1)) || (ns == 3 && at == 0)) && R == 3 && S == 3 && stride == 1 &&
           pad == 1 && Ho == H && Wo == W && H % 4 == 0 && W % 16 == 0 && O % 64 == 0 && Cb % 64 == 0;
}
static bool wgrad_uses_tr(int at, int ns, int Cb) {
    static const int tr_env = dbn_env_int("DBN_WGRAD_TR", 1);
    return at == 1 && ns == 1 && Cb % 32 == 0 && tr_env && g_wgrad_variant != 2;
}
// (wgrad_uses_patch is defined above)
// tile variant as dbn_wgrad_tile_config, + 16 when the launch is wgrad_tr_kernel<BM,BN,2,2> (bf16 tensors) instead of
// wgrad_f32_kernel<BM,BN,2,2,ns,at> — the rocprofv3 symbol of the matrix kernel of a dbn_wgrad_t call
int dbn_wgrad_kernel_config(int at, int ns, int O, int J, int Cb) {
    return dbn_wgrad_tile_config(O, J) + (wgrad_uses_tr(at, ns, Cb) ? 16 : 0);
}
// ... with the layer geometry: + 32 when the launch is wgrad_patch_kernel<ns, at> (3x3 / stride 1 on the bf16 matrix pipe)
// ... + 64 when wgrad_f32_kernel runs with block-wise k-tiles and scalar-offset addressing (its last template argument, ROW = 1)
int dbn_wgrad_kernel_config_hw(int at, int ns, int O, int Cb, int R, int S, int stride, int pad, int Ho, int Wo, int H, int W) {
    if (wgrad_uses_patch(at, ns, O, Cb, R, S, stride, pad, Ho, Wo, H, W)) return dbn_wgrad_tile_config(O, R * S * Cb) + 32;
    const int cfg = dbn_wgrad_kernel_config(at, ns, O, R * S * Cb, Cb);
    const bool dma = g_wgrad_variant == 1 && ns == 0 && at == 0;
    return cfg + ((cfg & 16) == 0 && at == 0 && !dma && g_wgrad_row16 && wgrad_row_tw(Ho, Wo) ? 64 : 0);
}

// thread groups sharing the splits of wgrad_reduce64_kernel (>= 4 splits each), its block size and dynamic LDS
static void wgrad_reduce64_plan(int RS, int splits_total, int& G, int& threads, size_t& smem) {
    const int items = RS * 16;
    G = std::min(1024 / items, splits_total / 4);
    if (G < 1) G = 1;
    if (G > 32) G = 32;
    if (const int e = dbn_env_int("DBN_REDUCE_G", 0)) G = std::max(1, std::min(e, 1024 / items > 0 ? 1024 / items : 1));  // experiments
    threads = std::min(1024, (items * G + 63) / 64 * 64);
    smem = (G > 1 ? (size_t)G * items * 4 * sizeof(double) : 0) + (size_t)RS * 64 * sizeof(float);
}

static int wgrad_run(const void* sm_, const void* big_, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                     int Cb, int I, int R, int S, int stride, int pad, float scale, int ns, void* stream, int at = 0, int phases = 3,
                     WgradReduceJob* reduce_job = nullptr) {
    DBN_REQUIRE(sm_ && big_ && slab && grad_oihw && (ns == 0 || ns == 1 || ns == 3) && phases >= 1 && phases <= 3);
    DBN_REQUIRE(at == 0 || (at == 1 && ns == 1) || (at == 3 && ns == 3));
    DBN_REQUIRE(O % 64 == 0 && Cb % 4 == 0 && I <= Cb && I > 0);
    const long sm_plane = at == 3 ? (long)N * Ho * Wo * O * 2 : 0, big_plane = at == 3 ? (long)N * H * W * Cb * 2 : 0;
    DBN_REQUIRE(3 * sm_plane < dbn_g_byte_limit && 3 * big_plane < dbn_g_byte_limit);
    const char* sm = reinterpret_cast<const char*>(sm_);
    const char* big = reinterpret_cast<const char*>(big_);
    const int es = dbn_esize(at);
    const int nmax = wgrad_chunk(N, Ho, Wo, O, H, W, Cb, at);
    DBN_REQUIRE(nmax >= 1);
    hipStream_t st = (hipStream_t)stream;
    if (g_wgrad_variant < 0) g_wgrad_variant = dbn_env_int("DBN_WGRAD_DMA", 0);
    // variant 1: the LDS-DMA kernel (natural slab order) for exact-fp32 math on fp32 tensors.  Measured against the register-
    // transposing kernel on the layer shapes of the model (tools/reduce_probe.py): 303-319 vs 268-301 us — not faster (its 48 KB
    // ring admits 3 workgroups per CU instead of 4, and the transposing kernel was not load-latency-bound after all), so it is
    // selectable (DBN_WGRAD_DMA=1 / dbn_set_wgrad_variant) but not the default.
    const bool dma = g_wgrad_variant == 1 && ns == 0 && at == 0;
    // stored bf16 operands: the LDS-DMA + transposing-read kernel (32-column blocks must not straddle taps: Cb % 32 == 0);
    // DBN_WGRAD_TR=0 / dbn_set_wgrad_variant(2) route them through the register-transposing kernel instead
    const bool trk = wgrad_uses_tr(at, ns, Cb);
    // 3x3 / stride 1 / pad 1 on the bf16 matrix pipe with whole 4 x 16 patches: the pixel-patch kernel (natural slabs, row length J)
    const bool pk = wgrad_uses_patch(at, ns, O, Cb, R, S, stride, pad, Ho, Wo, H, W);
    const bool natural = dma || trk || pk;
    int bm, bn;
    const int J = R * S * Cb;
    wgrad_tiles(O, J, bm, bn);
    const int njt = (J + bn - 1) / bn;
    const int Jp = pk ? J : njt * bn;
    int splits_total = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int n = std::min(nmax, N - n0);
        WgradParams p;
        p.sm = sm + (long)n0 * Ho * Wo * O * es; p.big = big + (long)n0 * H * W * Cb * es;
        p.slab = slab + (long)splits_total * wgrad_slab_stride(O, Jp);
        p.N = n; p.Ho = Ho; p.Wo = Wo; p.O = O; p.H = H; p.W = W; p.Cb = Cb; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
        p.P = n * Ho * Wo;
        p.J = J;
        p.rcp_HWo = 1.0f / (float)(Ho * Wo);
        p.rcp_Wo = 1.0f / (float)Wo;
        p.sm_bytes = (unsigned)((long)n * Ho * Wo * O * es + 2 * sm_plane);
        p.big_bytes = (unsigned)((long)n * H * W * Cb * es + 2 * big_plane);
        p.row_tw = g_wgrad_row16 ? wgrad_row_tw(Ho, Wo) : 0;
        p.sm_plane_bytes = (unsigned)sm_plane;
        p.big_plane_bytes = (unsigned)big_plane;
        const int splitk = wgrad_splitk_one(n, Ho, Wo, O, Cb, R, S);
        p.pchunk = (int)((((long)p.P + splitk - 1) / splitk + 15) / 16 * 16);
        dim3 grid((O / bm) * njt * splitk);
        if (!(phases & 1)) {
            splits_total += splitk;
            continue;
        }
        int rc_l;
        if (pk) {
            const int patches = n * (H / 4) * (W / 16);
            p.pchunk = (patches + splitk - 1) / splitk;
            const dim3 pgrid((O / 64) * (Cb / 64) * 3 * splitk);  // (a split past the last patch writes a zero slab)
            rc_l = dbn_launch_wgrad_b16(p, 2, ns, at, bm, bn, pgrid, st);
        } else if (trk) {
            rc_l = dbn_launch_wgrad_b16(p, 1, ns, at, bm, bn, grid, st);
        } else if (ns == 0 && at == 0) {
            rc_l = dbn_launch_wgrad_f32(p, dma ? 3 : 0, bm, bn, grid, st);
        } else {
            rc_l = dbn_launch_wgrad_b16(p, 0, ns, at, bm, bn, grid, st);
        }
        if (rc_l) return rc_l;
        splits_total += splitk;
    }
    if (!(phases & 2)) return DBN_OK;
    if (reduce_job) {  // (dbn_wgrad_reduce_job: describe the reduction instead of launching it)
        if (!(Cb % 64 == 0 && R * S * 64 * 4 <= 32 * 1024)) return DBN_ERR_ARG;
        int G, threads;
        size_t smem;
        wgrad_reduce64_plan(R * S, splits_total, G, threads, smem);
        *reduce_job = WgradReduceJob{slab, grad_oihw, splits_total, O, J, Jp, bm, bn, Cb, I, R * S, G, natural ? 1 : 0, O * (Cb / 64), scale,
                                     (int)smem};
        return DBN_OK;
    }
    if (Cb % 64 == 0 && R * S * 64 * 4 <= 32 * 1024) {
        int G, threads;
        size_t smem;
        wgrad_reduce64_plan(R * S, splits_total, G, threads, smem);
        hipLaunchKernelGGL(wgrad_reduce64_kernel, dim3(O, Cb / 64), dim3(threads), smem, st, slab, splits_total, O, J, Jp, bm, bn, Cb, I,
                           R * S, G, grad_oihw, scale, natural ? 1 : 0);
    } else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((long)O * Jp / 4 + 15) / 16)), dim3(256), 0, st, slab, splits_total, O, J, Jp, bm, bn,
                           Cb, I, R, S, grad_oihw, scale, natural ? 1 : 0);
    return dbn_status();
}

// General form: at = activation type of sm / big (0 fp32; 1 bf16, needs ns = 1), ns = matrix math.
int dbn_wgrad_t(int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream) {
    return wgrad_run(sm, big, slab, grad_oihw, N, Ho, Wo, O, H, W, Cb, I, R, S, stride, pad, scale, ns, stream, at);
}
// The two phases of dbn_wgrad_t as separate calls (same arguments): phase 1 = the matrix kernels (-> slabs), phase 2 = the
// slab reduction (-> grad_oihw).  For instrumentation (an event bracket around one kernel symbol) and for callers that want to
// put other work between them.
int dbn_wgrad_phase_t(int phase, int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo,
                      int O, int H, int W, int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream) {
    DBN_REQUIRE(phase == 1 || phase == 2);
    return wgrad_run(sm, big, slab, grad_oihw, N, Ho, Wo, O, H, W, Cb, I, R, S, stride, pad, scale, ns, stream, at, phase);
}

// Phase 2 of a dbn_wgrad_phase_t call DESCRIBED instead of launched: fills `job` (host memory, dbn_wgrad_reduce_job) for
// dbn_wgrad_reduce_many.  Only layers whose activation tensor has a multiple of 64 channels (every layer but the stem) have this
// form: DBN_ERR_ARG otherwise (the caller then runs phase 2 as its own launch).
int dbn_wgrad_reduce_describe(int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O,
                              int H, int W, int Cb, int I, int R, int S, int stride, int pad, float scale, void* job) {
    DBN_REQUIRE(job);
    return wgrad_run(sm, big, slab, grad_oihw, N, Ho, Wo, O, H, W, Cb, I, R, S, stride, pad, scale, ns, nullptr, at, 2,
                     reinterpret_cast<WgradReduceJob*>(job));
}
// One launch for the slab reductions of n_jobs layers.  jobs: DEVICE array of n_jobs dbn_wgrad_reduce_job records (as filled by
// dbn_wgrad_reduce_describe); first: DEVICE array of n_jobs + 1 ints, first[i] = sum of jobs[0..i-1].blocks; max_smem: the largest
// smem_bytes of the jobs.  Same sums in the same order as the per-layer launches: bit-identical gradients.
int dbn_wgrad_reduce_many(const void* jobs, const int* first, int n_jobs, int total_blocks, int max_smem, void* stream) {
    DBN_REQUIRE(jobs && first && n_jobs > 0 && total_blocks > 0 && max_smem > 0 && max_smem <= 64 * 1024);
    hipLaunchKernelGGL(wgrad_reduce64_many_kernel, dim3(total_blocks), dim3(1024), (size_t)max_smem, (hipStream_t)stream,
                       reinterpret_cast<const WgradReduceJob*>(jobs), first, n_jobs);
    return dbn_status();
}

int dbn_wgrad_f32(const float* sm, const float* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                  int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream) {
    return wgrad_run(sm, big, slab, grad_oihw, N, Ho, Wo, O, H, W, Cb, I, R, S, stride, pad, scale, 0, stream);
}

// dbn_wgrad_f32 on the bf16 matrix pipe (ns = 3: fp32-accurate operand split; ns = 1: bf16 operands)
int dbn_wgrad_bf16s(const float* sm, const float* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                    int Cb, int I, int R, int S, int stride, int pad, float scale, int ns, void* stream) {
    DBN_REQUIRE(ns == 1 || ns == 3);
    return wgrad_run(sm, big, slab, grad_oihw, N, Ho, Wo, O, H, W, Cb, I, R, S, stride, pad, scale, ns, stream);
}

}  // extern "C"
