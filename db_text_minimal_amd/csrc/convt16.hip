// ConvTranspose2d(64 -> 64, 2x2, stride 2) forward in 16-bit storage (bf16 training, fp16 / bf16 inference): the head's up-sampling layers,
// /root/reference/src/modules/segmentation_head.py:27-29,74-76 (+ the BatchNorm statistics of :28 / :75 in training).
//   out[n][2h + a][2w + b][co] = bias[co] + sum_ci x[n][h][w][ci] * W[ci][co][a][b]
// One GEMM per output-parity class (a, b): M = N * H * W input pixels, K = 64 = four k-steps of v_mfma_f32_32x32x16, N = 64.  The layer
// moves 5 output bytes per input byte and 13 GFLOP against 1.3 GB at batch 16 (2.1 GB at 32 x 320 x 320): it is HBM-bound by a wide
// margin, and the generic parity-class launch of the 16-bit loop — an LDS-DMA ring and 256 x 64 tiles around a K of four k-steps — ran it
// at 0.04 of the matrix peak, 2.5x its memory time (round-4 review; 0.82 ms per layer of the 1280^2 fp16 forward).  Here a wave keeps
// the WHOLE weight panel (4 classes x 4 k-steps x 2 column blocks: 128 registers) for its life, walks 32-pixel blocks with a grid
// stride, takes a block's A fragments straight from global memory (row li = 128 contiguous bytes, k-step t / k-half lh = 16 of them)
// ONCE for all four classes, and writes each class's 32 x 64 result through a wave-private LDS transpose as 16-byte row-major stores:
// no LDS panels, no barrier.  Train mode: pivot / sum / sum of squares per channel over the fp32 accumulators (+ bias), kept per WAVE over
// all its blocks and classes — one partial row per wave in bn_finalize_tiles_kernel's format (csrc/stem16.hip does the same).
//
// PW = 1 | 4 (round 5): the SAME kernel as a pointwise conv 64 -> 64 | 256 (nn.Conv2d(64, 64, 1): the FPN lateral reduce_conv_c2 of resnet18,
// /root/reference/src/modules/segmentation_body.py:46,68; inference form: folded BatchNorm in the bias, ReLU in the epilogue): the
// "classes" are the 64-channel blocks of the output pixel instead of its four sub-pixels.  At 32 x 320^2 fp16 the lateral moves 0.84 GB for
// 27 GFLOP; the generic 256 x 64-tile launch (a K of four k-steps: two ring stages per workgroup) took 1.34 ms for it.
#include "igemm_common.h"

namespace {

struct ConvT16Params {
    const void* x;     // [N][H][W][64] 16-bit
    const void* wpk;   // [4 classes][4 k-steps][2 k-halves][64 columns][8] 16-bit (dbn_convt16_pack)
    const float* bias; // [64] (PW: [256]) or NULL
    void* y;           // [N][2H][2W][64] 16-bit (PW: [N][H][W][256])
    int relu;          // PW: max(., 0) in the epilogue
    float* stats;      // optional: [3][64][rows] + [rows], rows = 4 * gridDim.x
    int N, H, W;
    int M;             // N * H * W (< 2^24)
    unsigned x_bytes;
};

constexpr int CT_PITCH = 64 + 8;

template <int AT, int PW = 0>  // PW: 0 = the ConvT; 1 / 4 = pointwise conv 64 -> 64 * PW
#ifndef DBN_CT16_WPE
#define DBN_CT16_WPE 2  // waves per SIMD the register allocation aims at (2: 256 registers, five weight fragments live in scratch; 1: none spilled)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PW == 1 ? 4 : DBN_CT16_WPE, PW == 1 ? 4 : DBN_CT16_WPE))) void convt2x2_b16_kernel(const ConvT16Params p) {
    static_assert(AT == 1 || AT == 2, "16-bit storage");
    __shared__ unsigned short smem[4 * 32 * CT_PITCH];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    constexpr int NCLS = PW ? PW : 4;
    u32x4_ bw[NCLS][4][2];  // [class][k-step][column block]
    {
        const u32x4_* Wp = reinterpret_cast<const u32x4_*>(p.wpk);
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < 2; ++b) bw[c][t][b] = Wp[((c * 4 + t) * 2 + lh) * 64 + b * 32 + li];
    }
    float bv[2];  // (PW: read per class inside the loop — eight more live registers spilled 40 more dwords)
#pragma unroll
    for (int b = 0; b < 2; ++b) bv[b] = (p.bias && !PW) ? p.bias[b * 32 + li] : 0.f;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned short* const T = smem + wave * 32 * CT_PITCH;
    const int HW = p.H * p.W;
    const float r_hw = 1.0f / (float)HW, r_w = 1.0f / (float)p.W;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
    bool have = false;
    int cnt = 0;
    for (int mb = gw; mb * 32 < p.M; mb += nw) {
        const int m0 = mb * 32, nrows = min(32, p.M - m0);
        u32x4_ a[4];
        {
            const unsigned base = (unsigned)min(m0 + li, p.M - 1) * 128u + (unsigned)lh * 16u;  // (rows past M repeat the last pixel; never stored)
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(base + (unsigned)t * 32u), 0, 0);
        }
        // this lane's store pieces: piece = lane + 64 j -> tile row piece >> 3 (input pixel m0 + row), eight channels c8 = piece & 7
        unsigned obase[4];  // (element offsets: the output stays below 2^31 elements, dbn_convt16_eligible)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (lane + 64 * j) >> 3;
            int n, rem, h, w_;
            divmod24(min(m0 + row, p.M - 1), HW, r_hw, n, rem);
            divmod24(rem, p.W, r_w, h, w_);
            obase[j] = PW ? (unsigned)min(m0 + row, p.M - 1) * (unsigned)(64 * PW) + (unsigned)(((lane + 64 * j) & 7) * 8)
                          : (unsigned)(((n * 2 * p.H + 2 * h) * 2 * p.W + 2 * w_) * 64 + ((lane + 64 * j) & 7) * 8);
        }
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float bvc = (PW && p.bias) ? p.bias[c * 64 + b * 32 + li] : bv[b];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = bvc;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (AT == 2)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, bw[c][t][b]), acc[b], 0, 0, 0);
                    else
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, bw[c][t][b]), acc[b], 0, 0, 0);
                }
            if (!PW && p.stats) {
                if (!have) {  // the wave's pivot: row 0 of its first block, class 0
#pragma unroll
                    for (int b = 0; b < 2; ++b) pv[b] = __shfl(acc[b][0], li, 64);
                    have = true;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float d = row < nrows ? acc[b][r] - pv[b] : 0.f;
                        s1[b] += d;
                        s2[b] += d * d;
                    }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    dbn_st1<AT>(T, row * CT_PITCH + b * 32 + li, (PW && p.relu) ? fmaxf(acc[b][r], 0.f) : acc[b][r]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // class (a, b) = (c >> 1, c & 1): output pixel (2h + a, 2w + b); PW: channels 64 c .. 64 c + 63 of the same pixel
            const unsigned coff = PW ? (unsigned)(64 * c) : (unsigned)(((c >> 1) * 2 * p.W + (c & 1)) * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = lane + 64 * j, row = piece >> 3, c8 = piece & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * CT_PITCH + c8 * 8);
                if (row < nrows) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(p.y) + obase[j] + coff) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        cnt += 4 * nrows;
    }
    if (!PW && p.stats) {
        const int rows = nw;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float t1 = s1[b] + __shfl_xor(s1[b], 32, 64), t2 = s2[b] + __shfl_xor(s2[b], 32, 64);
            if (lh == 0) {
                const long c = b * 32 + li;
                p.stats[(0L * 64 + c) * rows + gw] = pv[b];
                p.stats[(1L * 64 + c) * rows + gw] = t1;
                p.stats[(2L * 64 + c) * rows + gw] = t2;
            }
        }
        if (lane == 0) p.stats[3L * 64 * rows + gw] = (float)cnt;
    }
}

// W [ci 64][co 64][2][2] fp32 -> [class ab][k-step t][k-half lh][column co][8] in the 16-bit type: element j is input channel 16 t + 8 lh + j
__global__ void convt16_pack_kernel(const float* __restrict__ w, int f16, unsigned short* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * 4 * 2 * 64 * 8) return;
    const int j = idx & 7, col = (idx >> 3) & 63, lh = (idx >> 9) & 1, t = (idx >> 10) & 3, c = idx >> 12;
    const int ci = 16 * t + 8 * lh + j;
    const float v = w[((ci * 64 + col) * 2 + (c >> 1)) * 2 + (c & 1)];
    if (f16) {
        const _Float16 h = (_Float16)v;
        out[idx] = __builtin_bit_cast(unsigned short, h);
    } else {
        out[idx] = (unsigned short)bf16_bits_rne(v);
    }
}

// W [co 64 * ncls][ci 64] fp32 (a 1x1 conv's OIHW weight) -> the same panel: class = co / 64, column = co % 64
__global__ void pw16_pack_kernel(const float* __restrict__ w, int f16, unsigned short* __restrict__ out, int ncls) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ncls * 4 * 2 * 64 * 8) return;
    const int j = idx & 7, col = (idx >> 3) & 63, lh = (idx >> 9) & 1, t = (idx >> 10) & 3, c = idx >> 12;
    const int ci = 16 * t + 8 * lh + j;
    const float v = w[(c * 64 + col) * 64 + ci];
    if (f16) {
        const _Float16 h = (_Float16)v;
        out[idx] = __builtin_bit_cast(unsigned short, h);
    } else {
        out[idx] = (unsigned short)bf16_bits_rne(v);
    }
}

}  // namespace

extern "C" {

int dbn_convt16_rows(void) { return 4 * 512; }
long dbn_convt16_panel_bytes(void) { return 4L * 4 * 2 * 64 * 8 * 2; }
int dbn_convt16_eligible(int at, int N, int H, int W, int Cin, int Cout) {
    return (at == 1 || at == 2) && Cin == 64 && Cout == 64 && N > 0 && H > 0 && W > 0 && (long)N * H * W < (1L << 23) &&
           (long)N * H * W * 128 < dbn_g_byte_limit;  // (2^23 input pixels: 2^31 output elements)
}
// kind: 1 bf16, 2 fp16; w_iohw: the ConvTranspose2d weight [64][64][2][2]
int dbn_convt16_pack(int kind, const float* w_iohw, void* out, void* stream) {
    DBN_REQUIRE(w_iohw && out && (kind == 1 || kind == 2));
    hipLaunchKernelGGL(convt16_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w_iohw, kind == 2 ? 1 : 0, reinterpret_cast<unsigned short*>(out));
    return dbn_status();
}
// y [N][2H][2W][64] = ConvTranspose2d(x [N][H][W][64]) (+ bias).  gamma non-NULL: + the train-mode BatchNorm that follows, as dbn_conv_bn_t
// (ws: (3 * 64 + 1) * dbn_convt16_rows() floats).
int dbn_convt16_bn_t(int at, const void* x, const void* wpk, const float* bias, void* y, int N, int H, int W, const float* gamma, const float* beta,
                     float eps, float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                     float* ws, void* stream) {
    DBN_REQUIRE(x && wpk && y && dbn_convt16_eligible(at, N, H, W, 64, 64));
    DBN_REQUIRE(!gamma || (beta && scale && shift && save_mean && save_rstd && ws));
    ConvT16Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.y = y; p.stats = gamma ? ws : nullptr; p.relu = 0;
    p.N = N; p.H = H; p.W = W; p.M = N * H * W;
    p.x_bytes = (unsigned)((long)p.M * 128);
    hipStream_t st = (hipStream_t)stream;
    const int grid = 512;  // two workgroups per CU (~200 registers per lane: two waves per SIMD); rows = 4 * grid
    if (at == 1) hipLaunchKernelGGL(convt2x2_b16_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(convt2x2_b16_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    if (!gamma) return dbn_status();
    dbn_launch_bn_finalize_tiles(ws, 4 * grid, 64, gamma, beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd, st);
    return dbn_status();
}

// ---- pointwise conv 64 -> 64 | 256 on 16-bit storage (inference form), same kernel: y [N][H][W][Cout] = [relu](x [N][H][W][64] . W^T + bias)
int dbn_pw16_eligible(int at, int N, int H, int W, int Cin, int Cout) {
    return (at == 1 || at == 2) && Cin == 64 && (Cout == 64 || Cout == 256) && N > 0 && H > 0 && W > 0 && (long)N * H * W < (1L << 23) &&
           (long)N * H * W * 512 < dbn_g_byte_limit;
}
long dbn_pw16_panel_bytes(void) { return dbn_convt16_panel_bytes(); }
// kind: 1 bf16, 2 fp16; w_oihw: the conv weight [Cout][64][1][1], Cout = 64 or 256
int dbn_pw16_pack(int kind, const float* w_oihw, int Cout, void* out, void* stream) {
    DBN_REQUIRE(w_oihw && out && (kind == 1 || kind == 2) && (Cout == 64 || Cout == 256));
    hipLaunchKernelGGL(pw16_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w_oihw, kind == 2 ? 1 : 0, reinterpret_cast<unsigned short*>(out),
                       Cout / 64);
    return dbn_status();
}
int dbn_pw16_act_t(int at, const void* x, const void* wpk, const float* bias, int relu, void* y, int N, int H, int W, int Cout, void* stream) {
    DBN_REQUIRE(x && wpk && y && dbn_pw16_eligible(at, N, H, W, 64, Cout));
    ConvT16Params p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.y = y; p.stats = nullptr; p.relu = relu ? 1 : 0;
    p.N = N; p.H = H; p.W = W; p.M = N * H * W;
    p.x_bytes = (unsigned)((long)p.M * 128);
    hipStream_t st = (hipStream_t)stream;
    const int grid = Cout == 64 ? 1024 : 512;  // (64 -> 64: ~110 registers, four workgroups per CU)
    if (Cout == 256) {
        if (at == 1) hipLaunchKernelGGL((convt2x2_b16_kernel<1, 4>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convt2x2_b16_kernel<2, 4>), dim3(grid), dim3(256), 0, st, p);
    } else {
        if (at == 1) hipLaunchKernelGGL((convt2x2_b16_kernel<1, 1>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convt2x2_b16_kernel<2, 1>), dim3(grid), dim3(256), 0, st, p);
    }
    return dbn_status();
}

}  // extern "C"
