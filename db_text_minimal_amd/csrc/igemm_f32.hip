// fp32 implicit-GEMM convolution kernels for gfx950 (MI355X), built on the exact-f32
// matrix instruction v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157 TFLOP/s chip peak).
//
//   dbn_igemm_f32      forward conv (gather mode 0) and data-gradient / transposed
//                      conv (gather mode 1) over NHWC activations
//   dbn_wgrad_f32      weight gradient: split-K over output pixels into fp32 slabs
//   dbn_wgrad_reduce   deterministic slab reduction + scatter into OIHW gradients
//   dbn_pack_weights   OIHW -> [K/4][Cd][4] panels read by dbn_igemm_f32
//
// Replaces the ATen convolution calls under /root/reference/src/modules/resnet.py:70-91,
// 231-242, modules/basic.py:32-36, modules/segmentation_body.py:64-77 and
// modules/segmentation_head.py:24-29,64-79 (Conv2d / ConvTranspose2d forward and
// their autograd backward).
//
// Tiling: a workgroup of 4 waves owns a BM x BN output tile,
// each wave a (BM/WM) x (BN/WN) sub-tile held as 32x32 f32 accumulators.  K is walked
// in steps of 16; the A panel (im2col gather, 16 B per lane = 4 consecutive input
// channels of one tap) and the B panel (pre-packed weights) are staged through a
// double-buffered LDS image laid out [k/4][row][4] so that every lane fetches its four
// k-values for four consecutive MFMAs with one conflict-free ds_read_b128.
#include "common.h"

namespace {

struct IgemmParams {
    const float* src;   // [N,Hs,Ws,Cs]
    const float* wpk;   // [KT*4][Cd][4]
    const float* bias;  // [Cd] or null
    float* dst;         // [N,Hd,Wd,Cd]
    int N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate;
    int M, K, KT;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_f32_kernel(const IgemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_LD = BM * 4 / NT;  // float4 gathers per thread per k-tile
    constexpr int B_LD = BN * 4 / NT;
    constexpr int AS = BM + 2, BS = BN + 2;  // chunk strides (float4 units); +2 keeps ds_write_b128 conflict-free
    constexpr int STAGE = 4 * AS + 4 * BS;
    static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for the workgroup");
    __shared__ f32x4 smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = p.Cd / BN;
    const int tile = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread gather state -------------------------------------------------
    const int a_chunk = tid & 3;
    int a_hb[A_LD], a_wb[A_LD];
    long a_base[A_LD];
    bool a_ok[A_LD];
    const int HWd = p.Hd * p.Wd;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int row = (tid >> 2) + j * (NT / 4);
        const int m = m0 + row;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int n = mm / HWd;
        const int rem = mm - n * HWd;
        const int hd = rem / p.Wd;
        const int wd = rem - hd * p.Wd;
        a_base[j] = (long)n * p.Hs * p.Ws * p.Cs;
        if (p.mode == 0) {
            a_hb[j] = hd * p.stride - p.pad;
            a_wb[j] = wd * p.stride - p.pad;
        } else {
            a_hb[j] = hd + p.pad;
            a_wb[j] = wd + p.pad;
        }
    }
    // k-walk of this thread's chunk: k = kt*16 + 4*a_chunk = (r*S + s)*Cs + ci
    int kidx = 4 * a_chunk;
    int k_tap = kidx / p.Cs;
    int k_ci = kidx - k_tap * p.Cs;
    int k_r = k_tap / p.S;
    int k_s = k_tap - k_r * p.S;

    f32x4 ra[A_LD], rb[B_LD];

    auto gather = [&](int kt) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            bool v = a_ok[j] && (kidx < p.K);
            int hs, ws;
            if (p.mode == 0) {
                hs = a_hb[j] + k_r;
                ws = a_wb[j] + k_s;
            } else {
                const int th = a_hb[j] - k_r, tw = a_wb[j] - k_s;
                v = v && th >= 0 && tw >= 0;
                if (p.stride == 1) {
                    hs = th;
                    ws = tw;
                } else if (p.stride == 2) {
                    v = v && (((th | tw) & 1) == 0);
                    hs = th >> 1;
                    ws = tw >> 1;
                } else {
                    hs = th / p.stride;
                    ws = tw / p.stride;
                    v = v && (hs * p.stride == th) && (ws * p.stride == tw);
                }
            }
            v = v && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (v) val = *reinterpret_cast<const f32x4*>(p.src + a_base[j] + ((long)hs * p.Ws + ws) * p.Cs + k_ci);
            ra[j] = val;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int idx = tid + j * NT;
            const int c = idx / BN, n = idx - c * BN;
            rb[j] = *reinterpret_cast<const f32x4*>(p.wpk + ((long)(kt * 4 + c) * p.Cd + n0 + n) * 4);
        }
        // advance the k-walk by one tile (16 k)
        kidx += 16;
        k_ci += 16;
        while (k_ci >= p.Cs) {
            k_ci -= p.Cs;
            if (++k_s == p.S) {
                k_s = 0;
                ++k_r;
            }
        }
    };
    auto stage = [&](int buf) {
        f32x4* As = smem + buf * STAGE;
        f32x4* Bs = As + 4 * AS;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) As[a_chunk * AS + (tid >> 2) + j * (NT / 4)] = ra[j];
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int idx = tid + j * NT;
            const int c = idx / BN, n = idx - c * BN;
            Bs[c * BS + n] = rb[j];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    gather(0);
    stage(0);
    __syncthreads();

    for (int kt = 0; kt < p.KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < p.KT;
        if (more) gather(kt + 1);
        const f32x4* As = smem + buf * STAGE;
        const f32x4* Bs = As + 4 * AS;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) af[a] = As[(2 * s2 + lh) * AS + wm * TM + a * 32 + li];
#pragma unroll
            for (int b = 0; b < NI; ++b) bf[b] = Bs[(2 * s2 + lh) * BS + wn * TN + b * 32 + li];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int b = 0; b < NI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < MI; ++a) {
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = n0 + wn * TN + b * 32 + li;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) {
                    float* d = p.dst + (long)row * p.Cd + col;
                    float v = acc[a][b][r] + bv;
                    if (p.accumulate) v += *d;
                    *d = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(const IgemmParams& p, hipStream_t st) {
    const int grid = dbn_ceil_div(p.M, BM) * (p.Cd / BN);
    hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
    return dbn_status();
}

// --------------------------------------------------------------------------------
// weight gradient
// --------------------------------------------------------------------------------
struct WgradParams {
    const float* sm;   // [N,Ho,Wo,O]  (indexes the reduction)
    const float* big;  // [N,H,W,Cb]
    float* slab;       // [splitk][O][J]
    int N, Ho, Wo, O, H, W, Cb, R, S, stride, pad;
    int P, J, pchunk;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_f32_kernel(const WgradParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_LD = 4 * BM / NT, B_LD = 4 * BN / NT;
    constexpr int STAGE = 16 * (BM + BN);
    static_assert(A_LD >= 1 && B_LD >= 1, "tile too small");
    __shared__ float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int njt = (p.J + BN - 1) / BN;
    const int ot = blockIdx.x / njt, jt = blockIdx.x - ot * njt;
    const int o0 = ot * BM, j0 = jt * BN;
    const int pbeg = blockIdx.y * p.pchunk;
    const int pend = min(p.P, pbeg + p.pchunk);
    const int KT = (pend - pbeg + 15) / 16;

    // A slots: row = pixel within the k-tile, c4 = channel quad
    constexpr int A_C4 = BM / 4, B_C4 = BN / 4;
    const int a_c4 = tid % A_C4, a_row0 = tid / A_C4;
    const int b_c4 = tid % B_C4, b_row0 = tid / B_C4;
    constexpr int A_RSTEP = NT / A_C4, B_RSTEP = NT / B_C4;
    // this thread's B column quad -> (tap, ci)
    const int jj = j0 + 4 * b_c4;
    const bool j_ok = jj < p.J;
    const int tap = j_ok ? jj / p.Cb : 0;
    const int ci = j_ok ? jj - tap * p.Cb : 0;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int HWo = p.Ho * p.Wo;

    f32x4 ra[A_LD], rb[B_LD];
    auto gather = [&](int kt) {
        const int pk = pbeg + kt * 16;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int pp = pk + a_row0 + j * A_RSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < pend) v = *reinterpret_cast<const f32x4*>(p.sm + (long)pp * p.O + o0 + 4 * a_c4);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int pp = pk + b_row0 + j * B_RSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < pend && j_ok) {
                const int n = pp / HWo;
                const int rem = pp - n * HWo;
                const int oh = rem / p.Wo;
                const int ow = rem - oh * p.Wo;
                const int ih = oh * p.stride - p.pad + tr, iw = ow * p.stride - p.pad + ts;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    v = *reinterpret_cast<const f32x4*>(p.big + (((long)n * p.H + ih) * p.W + iw) * p.Cb + ci);
            }
            rb[j] = v;
        }
    };
    auto stage = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + 16 * BM;
#pragma unroll
        for (int j = 0; j < A_LD; ++j)
            *reinterpret_cast<f32x4*>(As + (a_row0 + j * A_RSTEP) * BM + 4 * a_c4) = ra[j];
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            *reinterpret_cast<f32x4*>(Bs + (b_row0 + j * B_RSTEP) * BN + 4 * b_c4) = rb[j];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (KT > 0) {
        gather(0);
        stage(0);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        if (more) gather(kt + 1);
        const float* As = smem + buf * STAGE;
        const float* Bs = As + 16 * BM;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float af[MI], bf[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) af[a] = As[(2 * kk + lh) * BM + wm * TM + a * 32 + li];
#pragma unroll
            for (int b = 0; b < NI; ++b) bf[b] = Bs[(2 * kk + lh) * BN + wn * TN + b * 32 + li];
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }

    float* out = p.slab + (long)blockIdx.y * p.O * p.J;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = j0 + wn * TN + b * 32 + li;
            if (col < p.J) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = o0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    out[(long)row * p.J + col] = acc[a][b][r];
                }
            }
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splitk, int O, int J, int Cb, int I, int R, int S,
                                    float* __restrict__ grad, float scale) {
    const long total = (long)O * J;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx / J);
        const int j = (int)(idx - (long)o * J);
        const int tap = j / Cb, i = j - tap * Cb;
        if (i >= I) continue;
        double s = 0.0;
        for (int z = 0; z < splitk; ++z) s += (double)slab[(long)z * total + idx];
        grad[((long)o * I + i) * (R * S) + tap] = (float)(s * scale);
    }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, int O, int I, int R, int S, int mode, int Cs, int Cd, int K,
                                    int Kpad, float* __restrict__ out) {
    const long total = (long)Kpad * Cd;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        const long q = idx >> 2;
        const int cd = (int)(q % Cd);
        const int kc = (int)(q / Cd);
        const int k = 4 * kc + e;
        float v = 0.f;
        if (k < K) {
            const int tap = k / Cs, cs = k - tap * Cs;
            const int r = tap / S, s = tap - r * S;
            if (mode == 0) {
                if (cs < I) v = w[(((long)cd * I + cs) * R + r) * S + s];
            } else {
                v = w[(((long)cs * I + cd) * R + r) * S + s];
            }
        }
        out[idx] = v;
    }
}

}  // namespace

extern "C" {

// Tile configuration dbn_igemm_f32 picks for an M x Cd output (tile_hint 0):
// 1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64 — the largest tile that still yields
// >= ~2 workgroups per CU on 256 CUs.
int dbn_igemm_tile_config(int M, int Cd) {
    const long b128 = (long)dbn_ceil_div(M, 128) * (Cd / 128);
    const long b256 = (long)dbn_ceil_div(M, 256) * (Cd / 64);
    const long b12864 = (long)dbn_ceil_div(M, 128) * (Cd / 64);
    if (Cd % 128 == 0 && b128 >= 512) return 1;
    if (b256 >= 512) return 2;
    if (b12864 >= 512) return 3;
    return 4;
}

int dbn_igemm_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                  int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, void* stream) {
    DBN_REQUIRE(src && wpk && dst);
    DBN_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    DBN_REQUIRE(Cs % 4 == 0 && Cd % 64 == 0 && (mode == 0 || mode == 1));
    DBN_REQUIRE((long)N * Hd * Wd < (1L << 31) && (long)N * Hs * Ws * Cs < (1L << 40));
    IgemmParams p;
    p.src = src; p.wpk = wpk; p.bias = bias; p.dst = dst;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.Hd = Hd; p.Wd = Wd; p.Cd = Cd;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.mode = mode; p.accumulate = accumulate;
    p.M = N * Hd * Wd;
    p.K = R * S * Cs;
    p.KT = (p.K + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    int cfg = tile_hint > 0 ? tile_hint : dbn_igemm_tile_config(p.M, Cd);
    if (cfg == 1 && Cd % 128 != 0) cfg = 3;
    switch (cfg) {
        case 1: return launch_igemm<128, 128, 2, 2>(p, st);
        case 2: return launch_igemm<256, 64, 4, 1>(p, st);
        case 3: return launch_igemm<128, 64, 2, 2>(p, st);
        default: return launch_igemm<64, 64, 2, 2>(p, st);
    }
}

int dbn_igemm_packed_floats(int K, int Cd) { return ((K + 15) / 16) * 16 * Cd; }

int dbn_pack_weights(const float* w_oihw, int O, int I, int R, int S, int mode, float* out, void* stream) {
    DBN_REQUIRE(w_oihw && out && O > 0 && I > 0 && R > 0 && S > 0 && (mode == 0 || mode == 1));
    const int Cs = (mode == 0) ? ((I + 3) / 4) * 4 : O;
    const int Cd = (mode == 0) ? O : I;
    DBN_REQUIRE(Cs % 4 == 0 && Cd % 64 == 0);
    const int K = R * S * Cs, Kpad = ((K + 15) / 16) * 16;
    const long total = (long)Kpad * Cd;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(dbn_grid(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I, R, S, mode,
                       Cs, Cd, K, Kpad, out);
    return dbn_status();
}

// Number of pixel splits dbn_wgrad_f32 will use (the caller sizes the slab workspace
// as splitk * O * R*S*Cb floats).
int dbn_wgrad_splitk(int N, int Ho, int Wo, int O, int Cb, int R, int S) {
    const long P = (long)N * Ho * Wo;
    const int J = R * S * Cb;
    const int bm = (O % 128 == 0 && J >= 128) ? 128 : 64;
    const int bn = (J >= 128) ? 128 : 64;
    const long tiles = (long)(O / bm) * ((J + bn - 1) / bn);
    long sk = (1024 + tiles - 1) / tiles;
    long maxsk = (P + 255) / 256;  // at least 256 pixels per split
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
    long pchunk = ((P + sk - 1) / sk + 15) / 16 * 16;
    return (int)((P + pchunk - 1) / pchunk);
}

int dbn_wgrad_f32(const float* sm, const float* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                  int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream) {
    DBN_REQUIRE(sm && big && slab && grad_oihw);
    DBN_REQUIRE(O % 64 == 0 && Cb % 4 == 0 && I <= Cb && I > 0);
    WgradParams p;
    p.sm = sm; p.big = big; p.slab = slab;
    p.N = N; p.Ho = Ho; p.Wo = Wo; p.O = O; p.H = H; p.W = W; p.Cb = Cb; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.P = N * Ho * Wo;
    p.J = R * S * Cb;
    const int splitk = dbn_wgrad_splitk(N, Ho, Wo, O, Cb, R, S);
    p.pchunk = (int)((((long)p.P + splitk - 1) / splitk + 15) / 16 * 16);
    hipStream_t st = (hipStream_t)stream;
    const int bm = (O % 128 == 0 && p.J >= 128) ? 128 : 64;
    const int bn = (p.J >= 128) ? 128 : 64;
    dim3 grid((O / bm) * ((p.J + bn - 1) / bn), splitk);
    if (bm == 128 && bn == 128)
        hipLaunchKernelGGL((wgrad_f32_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
    else if (bn == 128)
        hipLaunchKernelGGL((wgrad_f32_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((wgrad_f32_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
    int rc = dbn_status();
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dbn_grid((long)O * p.J)), dim3(256), 0, st, slab, splitk, O, p.J, Cb, I, R, S,
                       grad_oihw, scale);
    return dbn_status();
}

}  // extern "C"
