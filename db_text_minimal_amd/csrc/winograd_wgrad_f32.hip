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
#ifndef DBN_WWG_PIPE
#define DBN_WWG_PIPE 1  // 0: the k-steps in the compiler's order (A/B builds)
#endif
#ifndef DBN_WWG_EXP
#define DBN_WWG_EXP 0  // timing experiments (wrong results): 1 = the first patch only (no staging / barriers in the loop), 2 = no LDS reads / transforms
#endif

namespace {

constexpr int WG_XPX = 180, WG_XROW = 18, WG_YPX = 128, WG_YROW = 16;

struct WinoWgradParams {
    const float* x;   // [N][H][W][Cx]
    const float* dy;  // [N][H][W][Cy]
    float* slab;      // [nsplit][nob * nib][16 points][64 o][64 i]
    int N, H, W, Cx, Cy;
    int nob, nib;           // 64-channel blocks of dY / x
    int tw, gpi, groups;    // patches per patch row, per image, in all
    int nsplit;
    unsigned x_bytes, dy_bytes;
    const float *x_scale, *x_shift;  // optional: x is the input of a BatchNorm + ReLU; relu(fma(x, x_scale[c], x_shift[c])) is applied on load
};

// LIN = false: a workgroup's 32 tiles per k-group are the 4 x 8 tiles of an 8 x 16-pixel patch (large maps; ragged edges zero-filled).
// LIN = true (maps up to 40 pixels wide: layer3 / layer4 of the 640^2 benchmark): the 32 tiles are CONSECUTIVE tiles of one image in
// row-major order over its ceil(H/2) x ceil(W/2) tile grid — a 20 x 20 map fills 52 % of its 8 x 16 patches but 100 / 128 of its tile
// groups — and the LDS holds the band of pixel rows those tiles touch, full width (x: + the one-pixel halo).  The band is too large
// for a register prefetch beside the 128 accumulators: it is loaded and stored between the two barriers of a group.
constexpr int WL_XPX = 336, WL_YPX = 240;  // most band pixels of x / dY in the LIN form (dbn_winograd_wgrad_linear checks the map)
template <bool LIN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void winograd_wgrad_f32_kernel(const WinoWgradParams p) {
    constexpr int XPX = LIN ? WL_XPX : WG_XPX, YPX = LIN ? WL_YPX : WG_YPX;
    constexpr int XCH = XPX * 16, YCH = YPX * 16;
    constexpr int NXP = (XCH + 511) / 512, NYP = (YCH + 511) / 512;  // staging pieces per thread
    // x [pixels][64 ch], dY [pixels][64 ch]: 45 + 32 KB, TWICE in the patch form (patch g + 1 is stored while patch g is multiplied: one
    // barrier per patch; 154 of the CU's 160 KB — one workgroup is resident per CU anyway); LIN: 84 + 60 KB, once
    constexpr int NBUF = LIN ? 1 : 2;
    __shared__ f32x4 smem[NBUF * (XCH + YCH)];
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
    // geometry of the LDS images: row pitch of the x image (with halo) and of the dY image, in pixels
    const int TW = (p.W + 1) >> 1, TI = ((p.H + 1) >> 1) * TW;  // (LIN) tile grid of one image
    const int prow = LIN ? 2 * TW + 2 : WG_XROW, yrow = LIN ? 2 * TW : WG_YROW;
    const float rTW = 1.0f / (float)TW;

    // ---- staging pieces: 16-byte chunk idx = tid + 512 j of the x image (pixel idx >> 4, chunk idx & 15), of the dY image
    unsigned xrel[NXP], yrel[NYP];
    int xpos[NXP], ypos[NYP];  // py | px << 8, or -1 past the image
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int idx = tid + j * 512, pix = idx >> 4, ch = idx & 15;
        int py, px;
        if constexpr (LIN) divmod24(pix, prow, 1.0f / (float)prow, py, px);
        else { py = pix / WG_XROW; px = pix - py * WG_XROW; }
        xrel[j] = (unsigned)((py * p.W + px) * p.Cx + ch * 4) * 4u;
        xpos[j] = idx < XCH ? (py | (px << 8)) : -1;
    }
#pragma unroll
    for (int j = 0; j < NYP; ++j) {
        const int idx = tid + j * 512, pix = idx >> 4, ch = idx & 15;
        int py, px;
        if constexpr (LIN) divmod24(pix, yrow, 1.0f / (float)yrow, py, px);
        else { py = pix >> 4; px = pix & 15; }
        yrel[j] = (unsigned)((py * p.W + px) * p.Cy + ch * 4) * 4u;
        ypos[j] = idx < YCH ? (py | (px << 8)) : -1;
    }
    f32x4 rx[NXP], ry[LIN ? 1 : NYP];
    // apply-on-load: this thread's x pieces are chunk tid & 15 of every pixel = channels 64 ib + 4 (tid & 15) .. + 3 throughout
    // (patch form: held in registers; LIN — registers are short, and its staging does not overlap the matrix phase anyway — re-read per group)
    const bool act = p.x_scale != nullptr;
    f32x4 asc_ = {0.f, 0.f, 0.f, 0.f}, ash_ = {0.f, 0.f, 0.f, 0.f};
    if (act && !LIN) {
        asc_ = *reinterpret_cast<const f32x4*>(p.x_scale + ib * 64 + (tid & 15) * 4);
        ash_ = *reinterpret_cast<const f32x4*>(p.x_shift + ib * 64 + (tid & 15) * 4);
    }
    unsigned xstatic = 0;  // bit j: piece j exists (idx < the image's chunks)
#pragma unroll
    for (int j = 0; j < NXP; ++j) xstatic |= (unsigned)(xpos[j] >= 0) << j;
    unsigned xvalid = 0;  // bit j: piece j of the staged x image lies inside the map (outside: the conv's zero padding of the activation)
    int t0 = 0, tr0 = 0;  // (LIN) first tile of the current group, its tile row
    // origin of group g's LDS images in the tensors: image, first pixel row / column of the x image (hs0, ws0: one before the first
    // output pixel), rows of the x image that belong to the group (LIN: the band; patch form: all 10)
    auto geometry = [&](int g, int& n, int& hs0, int& ws0, int& xrows) {
        n = g / p.gpi;
        const int t = g - n * p.gpi;
        if constexpr (LIN) {
            const int tfirst = t * 32, tlast = min(tfirst + 31, TI - 1);
            const int r0_ = tfirst / TW, r1_ = tlast / TW;
            hs0 = 2 * r0_ - 1;
            ws0 = -1;
            xrows = 2 * (r1_ - r0_ + 1) + 2;
        } else {
            const int ty = t / p.tw, tx = t - ty * p.tw;
            hs0 = ty * 8 - 1;
            ws0 = tx * 16 - 1;
            xrows = 10;
        }
    };
    auto load_x = [&](int g) {  // (uniform g)
        int n, hs0, ws0, xrows;
        geometry(g, n, hs0, ws0, xrows);
        const unsigned bx = (unsigned)(((n * p.H + hs0) * p.W + ws0) * p.Cx + ib * 64) * 4u;  // (may wrap: only in-map pieces use it)
        // (uniform) the whole 10 x 18 window lies inside the map — 72 % of the patches of a 160 x 160 map: one add per piece instead of
        // six compares / selects (a vector instruction costs fp32-MFMA time; the pieces past the image keep OOB_OFFSET in xrel0)
        if (!LIN && hs0 >= 0 && hs0 + 10 <= p.H && ws0 >= 0 && ws0 + WG_XROW <= p.W) {
#pragma unroll
            for (int j = 0; j < NXP; ++j) rx[j] = buffer_load_f32x4(rsX, xpos[j] >= 0 ? bx + xrel[j] : OOB_OFFSET);
            xvalid = xstatic;
            return;
        }
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            const int py = xpos[j] & 255, px = xpos[j] >> 8;
            const bool v = xpos[j] >= 0 && py < xrows && (unsigned)(hs0 + py) < (unsigned)p.H && (unsigned)(ws0 + px) < (unsigned)p.W;
            rx[j] = buffer_load_f32x4(rsX, v ? bx + xrel[j] : OOB_OFFSET);
            xvalid = (xvalid & ~(1u << j)) | ((unsigned)v << j);
        }
    };
    int sbuf = 0;  // (patch form) LDS image the next store_group() fills
    auto store_x = [&](int xrows) {
        f32x4* const dstX = smem + sbuf * (XCH + YCH);
        if (act) {
            const f32x4 asc = LIN ? *reinterpret_cast<const f32x4*>(p.x_scale + ib * 64 + (tid & 15) * 4) : asc_;
            const f32x4 ash = LIN ? *reinterpret_cast<const f32x4*>(p.x_shift + ib * 64 + (tid & 15) * 4) : ash_;
#pragma unroll
            for (int j = 0; j < NXP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) rx[j][e] = ((xvalid >> j) & 1u) ? dbn_affine_relu(rx[j][e], asc[e], ash[e]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NXP; ++j)
            if (xpos[j] >= 0 && (xpos[j] & 255) < xrows) dstX[tid + j * 512] = rx[j];
    };
    auto y_piece = [&](int j, int n, int hs0, int ws0, int xrows) {  // dY image: the x image without its halo
        const int py = ypos[j] & 255, px = ypos[j] >> 8;
        const unsigned by = (unsigned)(((n * p.H + hs0 + 1) * p.W + ws0 + 1) * p.Cy + ob * 64) * 4u;
        if (!LIN && hs0 + 9 <= p.H && ws0 + 17 <= p.W) return buffer_load_f32x4(rsY, by + yrel[j]);  // (uniform) a whole 8 x 16 patch
        const bool v = ypos[j] >= 0 && py < xrows - 2 && hs0 + 1 + py < p.H && ws0 + 1 + px < p.W;
        return buffer_load_f32x4(rsY, v ? by + yrel[j] : OOB_OFFSET);
    };
    auto load_group = [&](int g) {  // patch form: both images into registers (stored after the compute phase of the previous group)
        int n, hs0, ws0, xrows;
        geometry(g, n, hs0, ws0, xrows);
        load_x(g);
        if constexpr (!LIN) {
#pragma unroll
            for (int j = 0; j < NYP; ++j) ry[j] = y_piece(j, n, hs0, ws0, xrows);
        }
    };
    auto store_group = [&]() {
        store_x(10);
        if constexpr (!LIN) {
#pragma unroll
            for (int j = 0; j < NYP; ++j) smem[sbuf * (XCH + YCH) + XCH + tid + j * 512] = ry[j];
        }
        sbuf ^= 1;
    };
    auto stage_lin = [&](int g) {  // LIN: load and store between the group's two barriers; dY in batches of four pieces (registers)
        int n, hs0, ws0, xrows;
        geometry(g, n, hs0, ws0, xrows);
        load_x(g);
        f32x4 q[4];
#pragma unroll
        for (int h = 0; h < NYP; h += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h + j < NYP) q[j] = y_piece(h + j, n, hs0, ws0, xrows);
            if (h == 0) store_x(xrows);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h + j < NYP && ypos[h + j] >= 0 && (ypos[h + j] & 255) < xrows - 2) smem[XCH + tid + (h + j) * 512] = q[j];
        }
        t0 = (g - n * p.gpi) * 32;
        tr0 = t0 / TW;
    };

    // ---- this wave's rows of the two transforms
    //   B^T row i (input window): 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3          -> rows a1, a2, sign sa
    //   A row i (dY tile):        0: y0;       1: y0 + y1;  2: y0 - y1;  3: -y1              -> coefficients c0, c1
    const int a1 = irow == 0 ? 0 : (irow == 2 ? 2 : 1), a2 = irow == 0 ? 2 : (irow == 1 ? 2 : (irow == 2 ? 1 : 3));
    const float sa = irow == 1 ? 1.f : -1.f;
    const float c0 = irow == 3 ? 0.f : 1.f, c1 = irow == 0 ? 0.f : (irow == 1 ? 1.f : -1.f);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // lane (li, lh): tile 2 s + lh of k-step s — patch form: tile (s >> 2, 2 (s & 3) + lh) of the patch's 4 x 8; LIN: tile t0 + 2 s + lh of
    // the image's grid (past the last tile: dY operand zero) —; x channels 2 li, 2 li + 1 (-> the two 32-wide halves of I: even / odd
    // channels), dY channel 32 oh + li
    const f32x2* const X2 = reinterpret_cast<const f32x2*>(smem) + (LIN ? 0 : (2 * lh) * 32) + li;
    const f32x2* const Xr1 = X2 + a1 * prow * 32;
    const f32x2* const Xr2 = X2 + a2 * prow * 32;
    const float* const Yb = reinterpret_cast<const float*>(smem + XCH) + (LIN ? 0 : (2 * lh) * 64) + oh * 32 + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.f;

    if constexpr (!LIN) {
        if (g0 < g1) {
            load_group(g0);
            store_group();
            if (g0 + 1 < g1) load_group(g0 + 1);
        }
        __syncthreads();
    }
    for (int g = g0; g < g1; ++g) {
        int cbuf = 0;  // LDS image this patch is multiplied from
        if constexpr (LIN) {
            if (g > g0) __syncthreads();  // every wave has read the previous band
            stage_lin(g);
            __syncthreads();
        } else {
            // patch g + 1 (in registers since the previous iteration) goes to the OTHER image — every wave left it at the barrier that
            // ended iteration g - 1 — and patch g + 2 starts its way from memory; both overlap this patch's matrix phase
            cbuf = (g - g0) & 1;
            if (DBN_WWG_EXP != 1) {
                if (g + 1 < g1) store_group();
                if (g + 2 < g1) load_group(g + 2);
            }
        }
        const int xoff2 = cbuf * (XCH + YCH) * 2, yoff4 = cbuf * (XCH + YCH) * 4;  // offsets of the image in float2 / float units
#if DBN_WWG_PIPE
        if constexpr (!LIN) {
            // k-steps in a two-stage software pipeline: the LDS reads of step s + 1 go out BEFORE the eight MFMAs of step s, its operands
            // are formed AFTER they are issued (while they run) — the compiler's order is reads -> wait -> ~40 vector instructions ->
            // 8 MFMAs per step, a latency chain the second resident wave of the SIMD only partly covers (-DDBN_WWG_EXP=5/6).
            f32x2 U1[4], U2[4], V[4];
            float dd[4], D[4];
            auto fetch = [&](int s_) {
                const int xo = xoff2 + (2 * (s_ >> 2) * WG_XROW + 4 * (s_ & 3)) * 32, yo = yoff4 + (2 * (s_ >> 2) * WG_YROW + 4 * (s_ & 3)) * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    U1[q] = Xr1[xo + q * 32];
                    U2[q] = Xr2[xo + q * 32];
                }
                dd[0] = Yb[yo];
                dd[1] = Yb[yo + 64];
                dd[2] = Yb[yo + WG_YROW * 64];
                dd[3] = Yb[yo + WG_YROW * 64 + 64];
            };
            auto form = [&]() {
                f32x2 R[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    R[q][0] = fmaf(sa, U2[q][0], U1[q][0]);  // (sa = +-1: exact)
                    R[q][1] = fmaf(sa, U2[q][1], U1[q][1]);
                }
                V[0] = R[0] - R[2];
                V[1] = R[1] + R[2];
                V[2] = R[2] - R[1];
                V[3] = R[1] - R[3];
                const float r0 = fmaf(c1, dd[2], c0 * dd[0]), r1 = fmaf(c1, dd[3], c0 * dd[1]);
                D[0] = r0;
                D[1] = r0 + r1;
                D[2] = r0 - r1;
                D[3] = -r1;
            };
            fetch(0);
            form();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                f32x2 Vc[4];
                float Dc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Vc[j] = V[j];
                    Dc[j] = D[j];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 16) fetch(s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(Dc[j], Vc[j][h], acc[j][h], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 16) form();
            }
        } else
#endif
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int xo, yo;
            float c0s = c0, c1s = c1;
            if constexpr (LIN) {
                const int t = t0 + 2 * s + lh;
                const bool ok = t < TI;
                int ty, tx;
                divmod24(ok ? t : t0, TW, rTW, ty, tx);
                ty -= tr0;
                xo = (2 * ty * prow + 2 * tx) * 32;
                yo = (2 * ty * yrow + 2 * tx) * 64;
                c0s = ok ? c0 : 0.f;
                c1s = ok ? c1 : 0.f;
            } else {
                xo = (2 * (s >> 2) * WG_XROW + 4 * (s & 3)) * 32;
                yo = (2 * (s >> 2) * WG_YROW + 4 * (s & 3)) * 64;
            }
            f32x2 R[4];
#if DBN_WWG_EXP == 2 || DBN_WWG_EXP == 6  // (6: the vector instructions on register operands, no LDS reads)
            for (int q = 0; q < 4; ++q) { R[q][0] = rx[q][0]; R[q][1] = rx[q][1]; }
            const float d00 = rx[0][0], d01 = rx[0][1], d10 = rx[1][0], d11 = rx[1][1];
#if DBN_WWG_EXP == 6
            for (int q = 0; q < 4; ++q) { R[q][0] = fmaf(sa, rx[q][1], R[q][0]); R[q][1] = fmaf(sa, rx[q][2], R[q][1]); }
#endif
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 u1 = Xr1[xoff2 + xo + q * 32], u2 = Xr2[xoff2 + xo + q * 32];
#if DBN_WWG_EXP == 5
                R[q][0] = u1[0];
                R[q][1] = u2[1];
#else
                R[q][0] = fmaf(sa, u2[0], u1[0]);  // (sa = +-1: exact)
                R[q][1] = fmaf(sa, u2[1], u1[1]);
#endif
            }
            const float* const Yc = Yb + yoff4;
            const float d00 = Yc[yo], d01 = Yc[yo + 64], d10 = Yc[yo + yrow * 64], d11 = Yc[yo + yrow * 64 + 64];
#endif
#if DBN_WWG_EXP == 2 || DBN_WWG_EXP == 5  // (5: the LDS reads, no transform arithmetic)
            const f32x2 V[4] = {R[0], R[1], R[2], R[3]};
            const float D[4] = {d00, d01, d10, d11};
#else
            const f32x2 V[4] = {R[0] - R[2], R[1] + R[2], R[2] - R[1], R[1] - R[3]};
            const float r0 = fmaf(c1s, d10, c0s * d00), r1 = fmaf(c1s, d11, c0s * d01);
            const float D[4] = {r0, r0 + r1, r0 - r1, -r1};
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(D[j], V[j][h], acc[j][h], 0, 0, 0);
        }
        if constexpr (!LIN) {
            if (g + 1 < g1 && DBN_WWG_EXP != 1) __syncthreads();  // patch g + 1 is in LDS, and every wave has read patch g
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

// slabs -> gradient, round 4's form: grid (O, nib); 1024 threads = 16 points x 64 input channels of one output channel, each walking the
// splits serially.  Still the launch for up to 64 splits (every layer but the 64 -> 64 ones: 14-22 us alone — the wide form below needs
// 19-55 us there, its 72 KB of LDS and 200 registers per lane leave it one workgroup per CU)
__global__ __launch_bounds__(1024) void winograd_wgrad_reduce_serial_kernel(const float* __restrict__ slab, int nsplit, int nsub, int nib, int I,
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

// slabs -> gradient: fp64 sum over the splits in a fixed order, dg = G^T dU G, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// Round 5 (the round-4 form — grid (O, nib), 1024 threads walking the splits serially with four 4-byte loads in flight — moved 67 MB in
// 68 us = 1 TB/s and sat 0.67 ms per step alone on the chip): workgroups of 256 threads = OG output channels x 8 input-channel quads x
// ZL split lanes (OG * ZL = 32; ZL = 32 / 16 / 4 for >= 32 / >= 16 / fewer splits), grid (O / OG, 2 nib): one half (32 input channels) of
// a 64 x 64 block's row(s).  A thread adds ITS share of the splits (ascending) for all 16 points of its four input channels — sixteen
// 16-byte loads in flight, 128 contiguous bytes per (split, point, o) and quad row — into 64 fp64 accumulators; the upper half of the
// split lanes hands its sums to the lower half through LDS (partner + own), the ZL / 2 results meet in LDS and are added in lane
// order, then G^T . G.  Every sum has ONE fixed order: run-to-run and phase-split bit identity hold.  67 MB in 20 us at 64 -> 64.
template <int ZL>
__global__ __launch_bounds__(256) void winograd_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int nsub, int nib, int I,
                                                                    float* __restrict__ grad, float scale) {
    constexpr int OG = 32 / ZL, ZH = ZL / 2;  // output channels per workgroup; split lanes after the pairwise hand-over
    __shared__ double part[ZH * OG][16][32 + 1];  // [split lane][o][point][input channel of this half] (+1: the fold reads columns)
    __shared__ double sh[OG][16][32];
    __shared__ float st[OG * 288];
    const int ib = blockIdx.y >> 1, ih = blockIdx.y & 1;
    const int iq = threadIdx.x & 7, og = (threadIdx.x >> 3) % OG, zl2 = threadIdx.x / (8 * OG), zl = zl2 % ZH;
    const int o = blockIdx.x * OG + og, sub = (o >> 6) * nib + ib;
    const float* src = slab + (long)sub * 16 * 4096 + (o & 63) * 64 + ih * 32 + iq * 4;
    const long stride = (long)nsub * 16 * 4096;
    const int z0 = (int)((long)zl2 * nsplit / ZL), z1 = (int)((long)(zl2 + 1) * nsplit / ZL);
    double acc[16][4];
#pragma unroll
    for (int pt = 0; pt < 16; ++pt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[pt][e] = 0.0;
    for (int z = z0; z < z1; ++z) {
        f32x4 v[16];
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) v[pt] = *reinterpret_cast<const f32x4*>(src + z * stride + pt * 4096);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pt][e] += (double)v[pt][e];
    }
    if (zl2 >= ZH) {
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[zl * OG + og][pt][iq * 4 + e] = acc[pt][e];
    }
    __syncthreads();
    if (zl2 < ZH) {
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[zl * OG + og][pt][iq * 4 + e] += acc[pt][e];  // (partner + own)
    }
    __syncthreads();
    for (int k = threadIdx.x; k < OG * 16 * 32; k += 256) {  // (o, point, channel): the split lanes' sums in lane order
        const int g = k >> 9, pt = (k >> 5) & 15, c = k & 31;
        double s_ = 0.0;
#pragma unroll
        for (int q = 0; q < ZH; ++q) s_ += part[q * OG + g][pt][c];
        sh[g][pt][c] = s_;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < OG * 288; k += 256) {
        const int g = k / 288, kk = k - g * 288, c = kk / 9, tap = kk - c * 9, r = tap / 3, q = tap - r * 3;
        // column r of G applied along i, column q along j
        auto gcol = [](int w, double v0, double v1, double v2, double v3) {
            return w == 0 ? v0 + 0.5 * (v1 + v2) : (w == 1 ? 0.5 * (v1 - v2) : 0.5 * (v1 + v2) + v3);
        };
        double t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = gcol(q, sh[g][4 * i + 0][c], sh[g][4 * i + 1][c], sh[g][4 * i + 2][c], sh[g][4 * i + 3][c]);
        st[k] = (float)(gcol(r, t[0], t[1], t[2], t[3]) * (double)scale);
    }
    __syncthreads();
    const int c0 = ib * 64 + ih * 32;
    const int n = max(0, min(32, I - c0)) * 9;  // channels >= I are padding of the activation tensor
    for (int k = threadIdx.x; k < OG * n; k += 256) {
        const int g = k / n, kk = k - g * n;
        grad[((long)(blockIdx.x * OG + g) * I + c0) * 9 + kk] = st[g * 288 + kk];
    }
}

int wwg_splits(int groups, int nsub) {
    static const int wgs = dbn_env_int("DBN_WWG_WGS", 256);  // workgroups the launch aims at (one is resident per CU: 8 waves x 256 registers)
    int ns = std::max(1, wgs / nsub);
    return std::min(ns, groups);
}

// LIN form (consecutive tiles, full-width band): 32 tiles span at most (30 + TW) / TW + 1 tile rows; the band must fit the LDS images
bool wwg_linear_fits(int H, int W) {
    const int TW = (W + 1) / 2, TH = (H + 1) / 2;
    const int R = std::min(TH, (30 + TW) / TW + 1);
    return (2 * R + 2) * (2 * TW + 2) <= WL_XPX && 2 * R * 2 * TW <= WL_YPX;
}
// share of real tiles among the tile slots of the two forms
double wwg_fill(int H, int W, bool lin) {
    if (lin) {
        const long TI = (long)((H + 1) / 2) * ((W + 1) / 2);
        return (double)TI / (double)((TI + 31) / 32 * 32);
    }
    return (double)H * W / ((double)((H + 7) / 8 * 8) * ((W + 15) / 16 * 16));
}
bool wwg_linear(int H, int W) {
    static const int env = dbn_env_int("DBN_WWG_LIN", 1);  // 0: patch form everywhere (A/B runs)
    // the band is staged without prefetch (~4 us per group unhidden): measured 16 x 20 x 20 x 512 -> 512: 246 -> 222 us, but 16 x 40 x 40 x
    // 256 -> 256 (83 % patch fill): 160 -> 188 us — only where the patch form wastes much more
    return env && wwg_linear_fits(H, W) && wwg_fill(H, W, true) > wwg_fill(H, W, false) + 0.2;
}
int wwg_groups(int N, int H, int W) {
    if (wwg_linear(H, W)) return N * (int)(((long)((H + 1) / 2) * ((W + 1) / 2) + 31) / 32);
    return N * ((H + 7) / 8) * ((W + 15) / 16);
}

}  // namespace

extern "C" {

// 1 when dbn_winograd_wgrad_f32 takes the layer: fp32 tensors, 3x3 / stride 1 / pad 1, channels in blocks of 64
int dbn_winograd_wgrad_eligible(int N, int H, int W, int O, int Cb, int I) {
    if (N < 1 || H < 1 || W < 1 || O % 64 || Cb % 64 || I < 1 || I > Cb) return 0;
    const long px = (long)N * H * W;
    if (px * std::max(O, Cb) * 4 >= dbn_g_byte_limit) return 0;
    return wwg_fill(H, W, wwg_linear(H, W)) >= 0.5;  // at least half of the tile slots are real (the transform's 2.25x pays from 45 %)
}
// 1: the layer runs in the consecutive-tile form (winograd_wgrad_f32_kernel<true> in a trace)
int dbn_winograd_wgrad_linear(int H, int W) { return wwg_linear(H, W); }
long dbn_winograd_wgrad_slab_floats(int N, int H, int W, int O, int Cb) {
    const int nsub = (O / 64) * (Cb / 64);
    return (long)wwg_splits(wwg_groups(N, H, W), nsub) * nsub * 16 * 4096;
}
// dg [O][I][3][3] = scale * the weight gradient of the conv with input x [N][H][W][Cb] (channels >= I: padding) and output gradient
// dy [N][H][W][O].  x_scale / x_shift non-NULL ([Cb] each): the conv's input is relu(x * x_scale[c] + x_shift[c]) — x is the input of the
// BatchNorm + ReLU in front of the conv, whose output was never written (see dbn_winograd_conv_bn_act_f32).  phases: 1 = the matrix kernel (-> slab), 2 = the reduction (slab -> grad), 3 = both.
int dbn_winograd_wgrad_f32(int phases, const float* dy, const float* x, const float* x_scale, const float* x_shift, float* slab, float* grad,
                           int N, int H, int W, int O, int Cb, int I, float scale, void* stream) {
    DBN_REQUIRE(dy && x && slab && grad && phases >= 1 && phases <= 3 && !x_scale == !x_shift);
    DBN_REQUIRE(dbn_winograd_wgrad_eligible(N, H, W, O, Cb, I));
    WinoWgradParams p;
    p.x = x;
    p.dy = dy;
    p.slab = slab;
    p.x_scale = x_scale;
    p.x_shift = x_shift;
    p.N = N;
    p.H = H;
    p.W = W;
    p.Cx = Cb;
    p.Cy = O;
    p.nob = O / 64;
    p.nib = Cb / 64;
    const bool lin = wwg_linear(H, W);
    p.tw = (W + 15) / 16;
    p.groups = wwg_groups(N, H, W);
    p.gpi = p.groups / N;
    p.nsplit = wwg_splits(p.groups, p.nob * p.nib);
    p.x_bytes = (unsigned)((long)N * H * W * Cb * 4);
    p.dy_bytes = (unsigned)((long)N * H * W * O * 4);
    hipStream_t st = (hipStream_t)stream;
    if (phases & 1) {
        const dim3 grid(p.nsplit * p.nob * p.nib);
        if (lin) hipLaunchKernelGGL(winograd_wgrad_f32_kernel<true>, grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL(winograd_wgrad_f32_kernel<false>, grid, dim3(512), 0, st, p);
    }
    if (phases & 2) {
        const int nsub = p.nob * p.nib;
        // 256 splits (the 64 -> 64 layers: one 64 x 64 block, every CU a split): the wide form (28 -> 20 us alone); fewer: round 4's
        if (p.nsplit >= 128)
            hipLaunchKernelGGL(winograd_wgrad_reduce_kernel<32>, dim3(O, 2 * p.nib), dim3(256), 0, st, slab, p.nsplit, nsub, p.nib, I, grad, scale);
        else
            hipLaunchKernelGGL(winograd_wgrad_reduce_serial_kernel, dim3(O, p.nib), dim3(1024), 0, st, slab, p.nsplit, nsub, p.nib, I, grad, scale);
    }
    return dbn_status();
}

}  // extern "C"
