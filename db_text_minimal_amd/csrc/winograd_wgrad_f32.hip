// Weight gradient of a 3x3 / stride-1 / pad-1 convolution in exact-fp32 arithmetic through the Winograd transform F(2x2, 3x3)
// (the transposed problem of winograd_f32.hip):  with  Y = A^T [ U (.) V ] A  per 2 x 2 output tile,  U = G g G^T,  V = B^T d B,
//   dU = sum over tiles of (A dY A^T) (.) (B^T d B),     dg = G^T dU G
// 16 element-wise points per tile instead of 36 multiply-adds per output pair: 2.25x fewer MFMA FLOPs than the direct form
// (wgrad_kernels.h).  The layers: /root/reference/src/modules/resnet.py:70-91 (BasicBlock convs), segmentation_body.py:55-61 (FPN
// smooth convs), segmentation_head.py:24-25,64-68 (the head's 256 -> 64 convs) — torch.autograd's conv weight gradient there.
//
// Mapping onto v_mfma_f32_32x32x2_f32: for point (i, j)  dU_ij[O][I] = DYt_ij[tiles][O]^T x V_ij[tiles][I]  is a GEMM whose K runs
// over the TILES (N * H/2 * W/2 of them).  A workgroup (8 waves) owns one 64 x 64 (O x I) block of all 16 points — wave w: point row
// i = w & 3, output-channel half w >> 2, both 32-wide halves of I: 4 points x 2 blocks x 16 = 128 accumulator registers — and a
// contiguous range of 8 x 16-pixel patches (32 tiles = 16 k-steps each).  Per patch the 10 x 18 window of x (64 channels, zero outside
// the map = the conv's padding) and the 8 x 16 pixels of dY (zero outside the map: ragged edges cost MFMAs, never a mask) are
// brought to LDS; every lane forms the transformed operands of ITS tile and channel on the fly (B^T and A have two non-zeros per
// row: 8 + 4 LDS reads and ~23 vector instructions per 8 MFMAs).  The workgroup's 16 x 64 x 64 partial sums go to a slab; the
// reduction kernel adds the slabs in fp64 in a fixed order and applies G^T . G (deterministic, no atomics).
#include "igemm_common.h"
#include <algorithm>
#ifndef DBN_WWG_EXP
#define DBN_WWG_EXP 0  // timing experiments (wrong results): 1 = the first patch only (no staging / barriers in the loop), 2 = no LDS reads / transforms
#endif

namespace {

constexpr int WG_XPX = 180, WG_XROW = 18, WG_YPX = 128, WG_YROW = 16;
constexpr int WG_XCH = WG_XPX * 16, WG_YCH = WG_YPX * 16;  // 16-byte chunks of the two patches (64 channels = 16 chunks per pixel)

struct WinoWgradParams {
    const float* x;   // [N][H][W][Cx]
    const float* dy;  // [N][H][W][Cy]
    float* slab;      // [nsplit][nob * nib][16 points][64 o][64 i]
    int N, H, W, Cx, Cy;
    int nob, nib;           // 64-channel blocks of dY / x
    int tw, gpi, groups;    // patches per patch row, per image, in all
    int nsplit;
    unsigned x_bytes, dy_bytes;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void winograd_wgrad_f32_kernel(const WinoWgradParams p) {
    __shared__ f32x4 smem[WG_XCH + WG_YCH];  // x patch [180 px][64 ch] (45 KB), dY patch [128 px][64 ch] (32 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int irow = wave & 3, oh = wave >> 2;
    const int nsub = p.nob * p.nib;
    const int b = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = b / nsub, sub = b - split * nsub;
    const int ob = sub / p.nib, ib = sub - ob * p.nib;
    const int g0 = (int)((long)split * p.groups / p.nsplit), g1 = (int)((long)(split + 1) * p.groups / p.nsplit);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);

    // ---- staging pieces: chunk idx = tid + 512 j of the x patch (j < 6; pixel idx >> 4, 16-byte chunk idx & 15), of the dY patch (j < 4)
    unsigned xrel[6], yrel[4];
    int xpos[6];  // py | px << 8, or -1 past the patch
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int idx = tid + j * 512, pix = idx >> 4, ch = idx & 15;
        const int py = pix / WG_XROW, px = pix - py * WG_XROW;
        xrel[j] = (unsigned)((py * p.W + px) * p.Cx + ch * 4) * 4u;
        xpos[j] = idx < WG_XCH ? (py | (px << 8)) : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + j * 512, pix = idx >> 4, ch = idx & 15;
        yrel[j] = (unsigned)(((pix >> 4) * p.W + (pix & 15)) * p.Cy + ch * 4) * 4u;
    }
    f32x4 rx[6], ry[4];
    auto load_group = [&](int g) {  // (uniform g)
        const int n = g / p.gpi, t = g - n * p.gpi, ty = t / p.tw, tx = t - ty * p.tw;
        const int ph0 = ty * 8, pw0 = tx * 16;
        const unsigned bx = (unsigned)(((n * p.H + ph0 - 1) * p.W + pw0 - 1) * p.Cx + ib * 64) * 4u;  // (may wrap: only in-map pieces use it)
        const unsigned by = (unsigned)(((n * p.H + ph0) * p.W + pw0) * p.Cy + ob * 64) * 4u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int py = xpos[j] & 255, px = xpos[j] >> 8;
            const bool v = xpos[j] >= 0 && (unsigned)(ph0 - 1 + py) < (unsigned)p.H && (unsigned)(pw0 - 1 + px) < (unsigned)p.W;
            rx[j] = buffer_load_f32x4(rsX, v ? bx + xrel[j] : OOB_OFFSET);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = (tid + j * 512) >> 4;
            const bool v = ph0 + (pix >> 4) < p.H && pw0 + (pix & 15) < p.W;
            ry[j] = buffer_load_f32x4(rsY, v ? by + yrel[j] : OOB_OFFSET);
        }
    };
    auto store_group = [&]() {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (xpos[j] >= 0) smem[tid + j * 512] = rx[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) smem[WG_XCH + tid + j * 512] = ry[j];
    };

    // ---- this wave's rows of the two transforms
    //   B^T row i (input window): 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3          -> rows a1, a2, sign sa
    //   A row i (dY tile):        0: y0;       1: y0 + y1;  2: y0 - y1;  3: -y1              -> coefficients c0, c1
    const int a1 = irow == 0 ? 0 : (irow == 2 ? 2 : 1), a2 = irow == 0 ? 2 : (irow == 1 ? 2 : (irow == 2 ? 1 : 3));
    const float sa = irow == 1 ? 1.f : -1.f;
    const float c0 = irow == 3 ? 0.f : 1.f, c1 = irow == 0 ? 0.f : (irow == 1 ? 1.f : -1.f);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // lane (li, lh): tile 2 s + lh of k-step s = tile (s >> 2, 2 (s & 3) + lh) of the patch's 4 x 8; x channels 2 li, 2 li + 1 (-> the
    // two 32-wide halves of I: even / odd channels), dY channel 32 oh + li
    const f32x2* const X2 = reinterpret_cast<const f32x2*>(smem) + (2 * lh) * 32 + li;
    const f32x2* const Xr1 = X2 + a1 * WG_XROW * 32;
    const f32x2* const Xr2 = X2 + a2 * WG_XROW * 32;
    const float* const Yb = reinterpret_cast<const float*>(smem + WG_XCH) + (2 * lh) * 64 + oh * 32 + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.f;

    if (g0 < g1) {
        load_group(g0);
        store_group();
    }
    __syncthreads();
    for (int g = g0; g < g1; ++g) {
        if (g + 1 < g1 && DBN_WWG_EXP != 1) load_group(g + 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int xo = (2 * (s >> 2) * WG_XROW + 4 * (s & 3)) * 32, yo = (2 * (s >> 2) * WG_YROW + 4 * (s & 3)) * 64;
            f32x2 R[4];
#if DBN_WWG_EXP == 2
            for (int q = 0; q < 4; ++q) { R[q][0] = rx[q][0]; R[q][1] = rx[q][1]; }
            const float d00 = ry[0][0], d01 = ry[0][1], d10 = ry[1][0], d11 = ry[1][1];
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 u1 = Xr1[xo + q * 32], u2 = Xr2[xo + q * 32];
                R[q][0] = fmaf(sa, u2[0], u1[0]);  // (sa = +-1: exact)
                R[q][1] = fmaf(sa, u2[1], u1[1]);
            }
            const float d00 = Yb[yo], d01 = Yb[yo + 64], d10 = Yb[yo + WG_YROW * 64], d11 = Yb[yo + WG_YROW * 64 + 64];
#endif
            const f32x2 V[4] = {R[0] - R[2], R[1] + R[2], R[2] - R[1], R[1] - R[3]};
            const float r0 = fmaf(c1, d10, c0 * d00), r1 = fmaf(c1, d11, c0 * d01);
            const float D[4] = {r0, r0 + r1, r0 - r1, -r1};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[j], V[j][h], acc[j][h], 0, 0, 0);
        }
        if (g + 1 < g1 && DBN_WWG_EXP != 1) {
            __syncthreads();  // every wave has read this patch
            store_group();
            __syncthreads();
        }
    }
    // ---- partial sums -> slab [split][sub][point][o][i]: accumulator row (r & 3) + 8 (r >> 2) + 4 lh = output channel within this wave's
    //      half, column li of half h = input channel 2 li + h
    float* const S = p.slab + ((long)b * 16 + 4 * irow) * 4096 + (oh * 32 + 4 * lh) * 64 + 2 * li;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x2 v;
            v[0] = acc[j][0][r];
            v[1] = acc[j][1][r];
            *reinterpret_cast<f32x2*>(S + j * 4096 + ((r & 3) + 8 * (r >> 2)) * 64) = v;
        }
}

// slabs -> gradient: fp64 sum over the splits in a fixed order, dg = G^T dU G, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// grid (O, nib); 1024 threads = 16 points x 64 input channels of one output channel
__global__ __launch_bounds__(1024) void winograd_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int nsub, int nib, int I,
                                                                     float* __restrict__ grad, float scale) {
    __shared__ double sh[16][64];
    __shared__ float st[576];
    const int o = blockIdx.x, ib = blockIdx.y, sub = (o >> 6) * nib + ib;
    const int pnt = threadIdx.x >> 6, il = threadIdx.x & 63;
    const float* src = slab + ((long)sub * 16 + pnt) * 4096 + (o & 63) * 64 + il;
    const long stride = (long)nsub * 16 * 4096;
    double s = 0.0;
    int z = 0;
    for (; z + 3 < nsplit; z += 4)
        s += ((double)src[z * stride] + (double)src[(z + 1) * stride]) + ((double)src[(z + 2) * stride] + (double)src[(z + 3) * stride]);
    for (; z < nsplit; ++z) s += (double)src[z * stride];
    sh[pnt][il] = s;
    __syncthreads();
    if (threadIdx.x < 576) {
        const int c = threadIdx.x / 9, tap = threadIdx.x - c * 9, r = tap / 3, q = tap - r * 3;
        // column r of G applied along i, column q along j
        auto gcol = [](int k, double v0, double v1, double v2, double v3) {
            return k == 0 ? v0 + 0.5 * (v1 + v2) : (k == 1 ? 0.5 * (v1 - v2) : 0.5 * (v1 + v2) + v3);
        };
        double t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = gcol(q, sh[4 * i + 0][c], sh[4 * i + 1][c], sh[4 * i + 2][c], sh[4 * i + 3][c]);
        st[threadIdx.x] = (float)(gcol(r, t[0], t[1], t[2], t[3]) * (double)scale);
    }
    __syncthreads();
    const int n = min(64, I - ib * 64) * 9;  // channels >= I are padding of the activation tensor
    float* dst = grad + ((long)o * I + ib * 64) * 9;
    for (int k = threadIdx.x; k < n; k += 1024) dst[k] = st[k];
}

int wwg_splits(int groups, int nsub) {
    static const int per_cu = dbn_env_int("DBN_WWG_PER_CU", 1);  // workgroups per CU the launch aims at (one is resident: 8 waves x 256 registers)
    int ns = std::max(1, 256 * per_cu / nsub);
    return std::min(ns, groups);
}

}  // namespace

extern "C" {

// 1 when dbn_winograd_wgrad_f32 takes the layer: fp32 tensors, 3x3 / stride 1 / pad 1, channels in blocks of 64
int dbn_winograd_wgrad_eligible(int N, int H, int W, int O, int Cb, int I) {
    if (N < 1 || H < 1 || W < 1 || O % 64 || Cb % 64 || I < 1 || I > Cb) return 0;
    const long px = (long)N * H * W;
    if (px * std::max(O, Cb) * 4 >= dbn_g_byte_limit) return 0;
    const long Hp = (H + 7) / 8 * 8, Wp = (W + 15) / 16 * 16;
    return 2L * H * W >= Hp * Wp;  // at least half of the patches' tiles are real (the transform's 2.25x pays from 45 %)
}
long dbn_winograd_wgrad_slab_floats(int N, int H, int W, int O, int Cb) {
    const int groups = N * ((H + 7) / 8) * ((W + 15) / 16), nsub = (O / 64) * (Cb / 64);
    return (long)wwg_splits(groups, nsub) * nsub * 16 * 4096;
}
// dg [O][I][3][3] = scale * the weight gradient of the conv with input x [N][H][W][Cb] (channels >= I: padding) and output gradient
// dy [N][H][W][O].  phases: 1 = the matrix kernel (-> slab), 2 = the reduction (slab -> grad), 3 = both.
int dbn_winograd_wgrad_f32(int phases, const float* dy, const float* x, float* slab, float* grad, int N, int H, int W, int O, int Cb, int I,
                           float scale, void* stream) {
    DBN_REQUIRE(dy && x && slab && grad && phases >= 1 && phases <= 3);
    DBN_REQUIRE(dbn_winograd_wgrad_eligible(N, H, W, O, Cb, I));
    WinoWgradParams p;
    p.x = x;
    p.dy = dy;
    p.slab = slab;
    p.N = N;
    p.H = H;
    p.W = W;
    p.Cx = Cb;
    p.Cy = O;
    p.nob = O / 64;
    p.nib = Cb / 64;
    p.tw = (W + 15) / 16;
    p.gpi = ((H + 7) / 8) * p.tw;
    p.groups = N * p.gpi;
    p.nsplit = wwg_splits(p.groups, p.nob * p.nib);
    p.x_bytes = (unsigned)((long)N * H * W * Cb * 4);
    p.dy_bytes = (unsigned)((long)N * H * W * O * 4);
    hipStream_t st = (hipStream_t)stream;
    if (phases & 1) hipLaunchKernelGGL(winograd_wgrad_f32_kernel, dim3(p.nsplit * p.nob * p.nib), dim3(512), 0, st, p);
    if (phases & 2)
        hipLaunchKernelGGL(winograd_wgrad_reduce_kernel, dim3(O, p.nib), dim3(1024), 0, st, slab, p.nsplit, p.nob * p.nib, p.nib, I, grad, scale);
    return dbn_status();
}

}  // extern "C"
