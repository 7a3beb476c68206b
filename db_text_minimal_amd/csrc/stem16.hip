// The stem convolution in 16-bit storage (bf16 training, fp16 / bf16 inference): Conv2d(3 -> 64, 7x7, stride 2, pad 3, no bias) of
// /root/reference/src/modules/resnet.py:167-172,231-235 on a PACKED input — four channels per pixel (r, g, b, 0), 8 bytes — instead of
// the 16-channel blocks the generic 16-bit loop needs (13 of 16 input channels zeros: K = 784 where 147 are real; round-4 review:
// 0.32 ms of the bf16 step at 0.04 of the matrix peak, 2.3 ms of the 1280^2 fp16 forward, plus a 16-channel copy of the image).
//
// GEMM view: M = N * Ho * Wo output pixels, N = 64, K = 7 rows x 8 taps x 4 channels = 224 = 14 k-steps of v_mfma_f32_32x32x16
// (the eighth tap and the fourth channel carry zero weights).  The input is a zero-BORDERED tensor xp [N][H + 6][Wp][4] (the image at
// offset (3, 3); Wp >= W + 6 even), so the conv is a "valid" one and no load needs a bounds test: k-step (r, sq) of output pixel
// (ho, wo) is the 32 contiguous bytes at xp[n][2 ho + r][2 wo + 4 sq ..+3], and lane (li, lh) of the MFMA — row li, k-half lh — takes
// 16 of them straight from global memory into its A fragment (32 lanes x 16 bytes = 512 contiguous bytes per half-wave; neighbouring
// rows overlap: L1 / L2 hits).  The whole weight panel (28 KB) lives in registers for the life of a wave, a wave walks 32-row blocks
// with a grid stride, and the 32 x 64 result leaves through a wave-private LDS transpose as 16-byte row-major stores: no LDS panels,
// no barrier.  Train mode: the BatchNorm statistics (pivot, sum, sum of squares per channel over the fp32 accumulators) are kept per
// WAVE over all its blocks — one partial row per wave in bn_finalize_tiles_kernel's format.
#include "igemm_common.h"

namespace {

struct Stem16Params {
    const void* xp;    // [N][Hp][Wp][4] 16-bit, zero border
    const void* wpk;   // [14 k-steps][2 k-halves][64 columns][8] 16-bit (dbn_stem16_pack)
    void* y;           // [N][Ho][Wo][64] 16-bit
    float* stats;      // optional: [3][64][rows] pivot / sum / sum sq + [rows] counts, rows = 4 * gridDim.x
    int N, Ho, Wo, Hp, Wp;
    int M;             // N * Ho * Wo (< 2^24)
    unsigned x_bytes;
};

constexpr int ST_PITCH = 64 + 8;  // 16-bit elements per row of the wave's transpose buffer (+16 bytes: rotates the banks)

template <int AT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem7x7_b16_kernel(const Stem16Params p) {
    static_assert(AT == 1 || AT == 2, "16-bit storage");
    __shared__ unsigned short smem[4 * 32 * ST_PITCH];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    // the weight panel: fragment of k-step t, column block b = 16 bytes at ((t * 2 + lh) * 64 + b * 32 + li) * 16
    u32x4_ bw[14][2];
    {
        const u32x4_* W = reinterpret_cast<const u32x4_*>(p.wpk);
#pragma unroll
        for (int t = 0; t < 14; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b) bw[t][b] = W[(t * 2 + lh) * 64 + b * 32 + li];
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xp), 0, p.x_bytes, 0x00020000);
    unsigned short* const T = smem + wave * 32 * ST_PITCH;
    const int HWo = p.Ho * p.Wo;
    const float r_hw = 1.0f / (float)HWo, r_w = 1.0f / (float)p.Wo;
    const unsigned rowpitch = (unsigned)p.Wp * 8u;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
    bool have = false;
    int cnt = 0;
    // The A fragments of a block travel in two halves of seven k-steps: the second half is in flight under the first half's MFMAs, the NEXT
    // block's first half under the second half's MFMAs and the whole epilogue (first build: all fourteen loads at the top of a block and
    // nothing in flight while it multiplied, transposed and stored — 4.3 us per block, 860 us for the 1280^2 fp16 forward's 2.1 GB).
    auto block_base = [&](int mb_) {
        const int m = min(mb_ * 32 + li, p.M - 1);  // (rows past M repeat the last pixel: loaded, multiplied, never stored or counted)
        int n, rem, ho, wo;
        divmod24(m, HWo, r_hw, n, rem);
        divmod24(rem, p.Wo, r_w, ho, wo);
        return (unsigned)((n * p.Hp + 2 * ho) * p.Wp + 2 * wo) * 8u + (unsigned)lh * 16u;
    };
    auto load_half = [&](u32x4_ (&a)[7], unsigned base, int half) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = half * 7 + i;
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(base + (unsigned)(t >> 1) * rowpitch + (unsigned)(t & 1) * 32u), 0, 0);
        }
    };
    u32x4_ a0[7], a1[7];
    if (gw * 32 < p.M) load_half(a0, block_base(gw), 0);
    for (int mb = gw; mb * 32 < p.M; mb += nw) {
        const int m0 = mb * 32;
        load_half(a1, block_base(mb), 1);
        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        auto mma = [&](const u32x4_ (&a)[7], int half) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int t = half * 7 + i;
                    if constexpr (AT == 2)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, bw[t][b]), acc[b], 0, 0, 0);
                    else
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bw[t][b]), acc[b], 0, 0, 0);
                }
        };
        mma(a0, 0);
        if ((mb + nw) * 32 < p.M) load_half(a0, block_base(mb + nw), 0);  // (the next block's first half)
        mma(a1, 1);
        const int nrows = min(32, p.M - m0);
        if (p.stats) {
            if (!have) {  // the wave's pivot: row 0 of its first block (accumulator register 0 of the lh = 0 half)
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
            cnt += nrows;
        }
        // 32 x 64 tile -> the wave's transpose buffer in the storage type -> 16-byte row-major stores (a row = 128 contiguous bytes)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                dbn_st1<AT>(T, row * ST_PITCH + b * 32 + li, acc[b][r]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS operations complete in order; the fence is for the compiler)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = lane + 64 * j, row = piece >> 3, c8 = piece & 7;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * ST_PITCH + c8 * 8);
            if (row < nrows) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(p.y) + ((long)(m0 + row) * 64 + c8 * 8)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the buffer is rewritten by the next block)
    }
    if (p.stats) {
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

// ---- inference: stem conv + BatchNorm (running statistics) + ReLU + MaxPool2d(3, 2, 1) in ONE kernel (round 5) -------------------------------
// /root/reference/src/modules/resnet.py:231-235 in eval mode.  The two-kernel form writes the 64-channel conv output (1.7 GB at 32 x 1280^2
// fp16) and reads it back for the pool: 0.67 + 0.49 ms of the 14 ms forward, most of it that round trip.  Here a wave owns a strip of 15
// pooled columns of one image and walks DOWN a chunk of pooled rows: per pooled row it computes the two new conv rows 2py, 2py + 1 (32
// conv columns 30 seg - 1 .. 30 seg + 30 each: one MFMA row block, 31 of them used), applies relu(acc * scale + shift) on the fp32
// accumulators, takes the horizontal 3-maxima of the row through its transpose buffer, and combines them with the maxima of conv row 2py - 1
// kept from the previous pooled row (a chunk's first row computes that one extra).  The post-ReLU values are >= +0, so 16-bit patterns order
// like unsigned integers and the maxima are packed integer maxima; conv columns / rows outside the map enter as 0 (never the maximum, as
// in bnrelu_maxpool_fwd_kernel).  Conv output, BatchNorm pass and pool input never touch memory.  Ho must be even.
struct StemPoolParams {
    const void* xp;   // [N][Hp][Wp][4] 16-bit, zero border
    const void* wpk;  // dbn_stem16_pack
    const float* sc;  // [64] eval-mode BatchNorm scale / shift
    const float* sh;
    void* out;        // [N][Hq][Wq][64] 16-bit
    int N, Ho, Wo, Hp, Wp, Hq, Wq, nseg, nchunk, rpc, ntask;
    unsigned x_bytes;
};

constexpr int SP_COLS = 15;  // pooled columns per strip (2 * 15 + 1 = 31 conv columns of the 32-row MFMA block)

template <int AT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem7x7_pool_b16_kernel(const StemPoolParams p) {
    static_assert(AT == 1 || AT == 2, "16-bit storage");
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
    constexpr int WAVE_LDS = 32 * ST_PITCH + 2 * 128 * 8;  // transpose buffer + this wave's PREV and HA item vectors (16-bit elements)
    __shared__ __attribute__((aligned(16))) unsigned short smem[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    u32x4_ bw[14][2];
    {
        const u32x4_* W = reinterpret_cast<const u32x4_*>(p.wpk);
#pragma unroll
        for (int t = 0; t < 14; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b) bw[t][b] = W[(t * 2 + lh) * 64 + b * 32 + li];
    }
    float scv[2], shv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        scv[b] = p.sc[b * 32 + li];
        shv[b] = p.sh[b * 32 + li];
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xp), 0, p.x_bytes, 0x00020000);
    unsigned short* const T = smem + wave * WAVE_LDS;
    u16x8_* const PREV = reinterpret_cast<u16x8_*>(T + 32 * ST_PITCH);
    u16x8_* const HA = PREV + 128;
    const unsigned rowpitch = (unsigned)p.Wp * 8u;
    auto load_half = [&](u32x4_ (&a)[7], unsigned base, int half) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int t = half * 7 + i;
            a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(base + (unsigned)(t >> 1) * rowpitch + (unsigned)(t & 1) * 32u), 0, 0);
        }
    };
    const u16x8_ zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int task = gw; task < p.ntask; task += nw) {
        const int seg = task % p.nseg, t2 = task / p.nseg, chunk = t2 % p.nchunk, n = t2 / p.nchunk;
        const int c0 = 2 * SP_COLS * seg - 1, py0 = chunk * p.rpc, py1 = min(p.Hq, py0 + p.rpc);
        // this lane's A base of conv row r: pixel (r, c0 + li) — the column may be -1 or >= Wo: valid memory of a neighbouring row (or out of
        // the buffer: zeros), finite values, masked below
        auto base_of = [&](int r) { return (unsigned)(((n * p.Hp + 2 * r) * p.Wp + 2 * (c0 + li)) * 8) + (unsigned)lh * 16u; };
        const int r_first = py0 == 0 ? 0 : 2 * py0 - 1, r_last = 2 * py1 - 1;
        if (py0 == 0) {  // no conv row above the map: its maxima are 0
            PREV[lane] = zero8;
            PREV[lane + 64] = zero8;
        }
        u32x4_ a0[7], a1[7];
        load_half(a0, base_of(r_first), 0);
        for (int r = r_first; r <= r_last; ++r) {
            load_half(a1, base_of(r), 1);
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
            auto mma = [&](const u32x4_ (&a)[7], int half) {
#pragma unroll
                for (int i = 0; i < 7; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int t = half * 7 + i;
                        if constexpr (AT == 2)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, bw[t][b]), acc[b], 0, 0, 0);
                        else
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bw[t][b]), acc[b], 0, 0, 0);
                    }
            };
            mma(a0, 0);
            if (r < r_last) load_half(a0, base_of(r + 1), 0);  // (the next row's first half)
            mma(a1, 1);
            // relu(bn(conv)) of the 32 x 64 tile -> the transpose buffer; columns outside the map are 0
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * lh;
                    const float v = fmaf(acc[b][q], scv[b], shv[b]);
                    const bool ok = (unsigned)(c0 + row) < (unsigned)p.Wo;
                    dbn_st1<AT>(T, row * ST_PITCH + b * 32 + li, (ok && v > 0.f) ? v : 0.f);  // (never -0: the integer maxima below rely on it)
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = lane + 64 * it;  // (pooled column item >> 3 of the strip, channels 8 (item & 7) ..)
                if (item >= SP_COLS * 8) continue;
                const int pxl = item >> 3, c8 = item & 7;
                const unsigned short* t0 = T + (2 * pxl) * ST_PITCH + c8 * 8;
                u16x8_ hm = *reinterpret_cast<const u16x8_*>(t0);
                hm = __builtin_elementwise_max(hm, *reinterpret_cast<const u16x8_*>(t0 + ST_PITCH));
                hm = __builtin_elementwise_max(hm, *reinterpret_cast<const u16x8_*>(t0 + 2 * ST_PITCH));
                if ((r & 1) == 0) {
                    HA[item] = hm;  // conv row 2py: waits for row 2py + 1
                } else {
                    const int py = (r - 1) >> 1;  // conv row 2py + 1 closes pooled row py (and is row 2(py + 1) - 1 of the next)
                    if (py >= py0) {
                        const u16x8_ m = __builtin_elementwise_max(__builtin_elementwise_max(PREV[item], HA[item]), hm);
                        const int px = SP_COLS * seg + pxl;
                        if (px < p.Wq)
                            *reinterpret_cast<u16x8_*>(reinterpret_cast<unsigned short*>(p.out) + (((long)n * p.Hq + py) * p.Wq + px) * 64 + c8 * 8) = m;
                    }
                    PREV[item] = hm;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the transpose buffer is rewritten by the next row)
        }
    }
}

// w [64][3][7][7] fp32 -> [14][2][64][8] in the 16-bit type: k-step t = (r = t >> 1, sq = t & 1), k-half lh, element j: tap 4 sq + 2 lh + (j >> 2),
// channel j & 3; the eighth tap and the fourth channel are zero
__global__ void stem16_pack_kernel(const float* __restrict__ w, int f16, unsigned short* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 14 * 2 * 64 * 8) return;
    const int j = idx & 7, col = (idx >> 3) & 63, lh = (idx >> 9) & 1, t = idx >> 10;
    const int r = t >> 1, sq = t & 1, tap = 4 * sq + 2 * lh + (j >> 2), ch = j & 3;
    const float v = (tap < 7 && ch < 3) ? w[((col * 3 + ch) * 7 + r) * 7 + tap] : 0.f;
    if (f16) {
        const _Float16 h = (_Float16)v;
        out[idx] = __builtin_bit_cast(unsigned short, h);
    } else {
        out[idx] = (unsigned short)bf16_bits_rne(v);
    }
}

// x [N,3,H,W] fp32 -> the interior of xp [N][Hp][Wp][4] (image at (3, 3); the border stays as the caller zeroed it) and, x4 non-NULL, the
// packed [N][H][W][4] form (the X operand of the stem's weight gradient), one pass
// A workgroup walks whole image rows (row = n * H + h: one division per row, uniform per workgroup), its threads the pixels of the row.
// Alone 28.7 us for 184 MB at bs16 640^2 (6.4 TB/s; 206 us for 1.05 GB at cfg5) — as fast as the flat-index form with its two 64-bit
// divisions per pixel (31.1 / 217 us) and as one workgroup per 256-pixel row segment (27.8 / 199 us).  In the training step it shows as
// 107-140 us in every form: it runs beside pack_many_kernel on the second stream there.
template <int AT>
__global__ __launch_bounds__(256) void nchw3_to_padded4_kernel(const float* __restrict__ x, void* __restrict__ xp, void* __restrict__ x4, int N,
                                                               int H, int W, int Hp, int Wp) {
    const long HW = (long)H * W, rows = (long)N * H;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = (int)(row / H), h = (int)(row - (long)n * H);
        const float* b = x + (long)n * 3 * HW + (long)h * W;
        const long po = ((long)n * Hp + h + 3) * Wp + 3;
        for (int w_ = threadIdx.x; w_ < W; w_ += 256) {
            const f32x4 v = {b[w_], b[HW + w_], b[2 * HW + w_], 0.f};
            dbn_st4<AT>(xp, po + w_, v);
            if (x4) dbn_st4<AT>(x4, row * W + w_, v);
        }
    }
}

}  // namespace

extern "C" {

// geometry of the zero-bordered packed input of a [N,3,H,W] image: rows H + 6, row pitch in pixels (even, >= W + 7)
int dbn_stem16_padded_w(int W) { return (W + 8) & ~1; }
int dbn_stem16_padded_h(int H) { return H + 6; }
// partial rows of BatchNorm statistics a call writes (= waves of the launch); ws: (3 * 64 + 1) * rows floats
int dbn_stem16_rows(void) { return 4 * 512; }
long dbn_stem16_panel_bytes(void) { return 14L * 2 * 64 * 8 * 2; }

// kind: 1 bf16, 2 fp16
int dbn_stem16_pack(int kind, const float* w_oihw, void* out, void* stream) {
    DBN_REQUIRE(w_oihw && out && (kind == 1 || kind == 2));
    hipLaunchKernelGGL(stem16_pack_kernel, dim3(56), dim3(256), 0, (hipStream_t)stream, w_oihw, kind == 2 ? 1 : 0, reinterpret_cast<unsigned short*>(out));
    return dbn_status();
}

int dbn_nchw3_to_padded4_t(int at, const float* x, void* xp, void* x4, int N, int H, int W, void* stream) {
    DBN_REQUIRE(x && xp && N > 0 && H > 0 && W > 0 && (at == 1 || at == 2));
    const int Hp = dbn_stem16_padded_h(H), Wp = dbn_stem16_padded_w(W);
    const dim3 grid(dbn_grid((long)N * H, 1));
    if (at == 1)
        hipLaunchKernelGGL(nchw3_to_padded4_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, xp, x4, N, H, W, Hp, Wp);
    else
        hipLaunchKernelGGL(nchw3_to_padded4_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, xp, x4, N, H, W, Hp, Wp);
    return dbn_status();
}

// 1: dbn_stem16_conv_bn_t takes the call (16-bit storage, index ranges)
int dbn_stem16_eligible(int at, int N, int H, int W) {
    if (!((at == 1 || at == 2) && N > 0 && H >= 7 && W >= 7)) return 0;
    const long Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return (long)N * Ho * Wo < (1L << 24) && (long)N * dbn_stem16_padded_h(H) * dbn_stem16_padded_w(W) * 8 < dbn_g_byte_limit &&
           (long)N * Ho * Wo * 64 * 2 < (1L << 40);
}

// y [N][Ho][Wo][64] = conv7x7/2 (xp), Ho = (H - 1) / 2 + 1.  gamma non-NULL: + the train-mode BatchNorm that follows, as dbn_conv_bn_t
// (scale / shift / saved mean / rstd, running statistics; ws: (3 * 64 + 1) * dbn_stem16_rows() floats).
int dbn_stem16_conv_bn_t(int at, const void* xp, const void* wpk, void* y, int N, int H, int W, const float* gamma, const float* beta, float eps,
                         float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                         float* ws, void* stream) {
    DBN_REQUIRE(xp && wpk && y && dbn_stem16_eligible(at, N, H, W));
    DBN_REQUIRE(!gamma || (beta && scale && shift && save_mean && save_rstd && ws));
    Stem16Params p;
    p.xp = xp; p.wpk = wpk; p.y = y; p.stats = gamma ? ws : nullptr;
    p.N = N; p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; p.Hp = dbn_stem16_padded_h(H); p.Wp = dbn_stem16_padded_w(W);
    p.M = N * p.Ho * p.Wo;
    p.x_bytes = (unsigned)((long)N * p.Hp * p.Wp * 8);
    hipStream_t st = (hipStream_t)stream;
    const int grid = 512;  // two workgroups per CU; rows = 4 * grid
    if (at == 1) hipLaunchKernelGGL(stem7x7_b16_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(stem7x7_b16_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    if (!gamma) return dbn_status();
    dbn_launch_bn_finalize_tiles(ws, 4 * grid, 64, gamma, beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd, st);
    return dbn_status();
}

// ---- inference: conv + eval-mode BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one launch (see stem7x7_pool_b16_kernel)
int dbn_stem16_pool_eligible(int at, int N, int H, int W) {
    if (!dbn_stem16_eligible(at, N, H, W)) return 0;
    const int Ho = (H - 1) / 2 + 1;
    return Ho % 2 == 0 && Ho >= 2;
}
// out [N][Hq][Wq][64] = maxpool3x3/2 (relu(conv7x7/2 (xp) * scale + shift)), Hq = (Ho - 1) / 2 + 1; scale / shift: [64] (dbn_bn_eval_coef)
int dbn_stem16_conv_bn_relu_pool_t(int at, const void* xp, const void* wpk, const float* scale, const float* shift, void* out, int N, int H, int W,
                                   void* stream) {
    DBN_REQUIRE(xp && wpk && scale && shift && out && dbn_stem16_pool_eligible(at, N, H, W));
    StemPoolParams p;
    p.xp = xp; p.wpk = wpk; p.sc = scale; p.sh = shift; p.out = out;
    p.N = N; p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; p.Hp = dbn_stem16_padded_h(H); p.Wp = dbn_stem16_padded_w(W);
    p.Hq = (p.Ho - 1) / 2 + 1; p.Wq = (p.Wo - 1) / 2 + 1;
    p.nseg = (p.Wq + SP_COLS - 1) / SP_COLS;
    // pooled rows per task: long enough that the chunk's one extra conv row is small (<= 1 / 40), short enough for ~5+ tasks per wave
    const int grid = 512, waves = 4 * grid;
    int rpc = 20;
    while (rpc > 4 && (long)N * p.nseg * ((p.Hq + rpc - 1) / rpc) < 5L * waves) rpc >>= 1;
    p.rpc = rpc; p.nchunk = (p.Hq + rpc - 1) / rpc;
    p.ntask = N * p.nseg * p.nchunk;
    p.x_bytes = (unsigned)((long)N * p.Hp * p.Wp * 8);
    hipStream_t st = (hipStream_t)stream;
    if (at == 1) hipLaunchKernelGGL(stem7x7_pool_b16_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(stem7x7_pool_b16_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    return dbn_status();
}

}  // extern "C"
