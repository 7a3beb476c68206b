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

// ---- inference: the DB head's tail in ONE launch (round 5) --------------------------------------------------------------------------------
// /root/reference/src/modules/segmentation_head.py:27-29,35-45,74-79 in eval mode, per branch (binarize / thresh):
//   ConvTranspose2d(64, 64, 2, 2) -> BatchNorm (running statistics) -> ReLU -> ConvTranspose2d(64, 1, 2, 2) -> Sigmoid
// Kernel = stride = 2 twice: an input pixel of the quarter-resolution map owns a 4 x 4 block of the full-resolution output and nothing else
// touches it, so the whole chain is local.  The three-kernel form wrote the two 64-channel half-resolution tensors (2 x 1.7 GB at 32 x 1280^2
// fp16) and read them back in the head-tail kernel: 2 x 0.43 + 0.95 ms of the 14 ms forward.  Here a wave takes 32 input pixels; per
// parity class of the first ConvT: 8 MFMAs (weights from a 32 KB LDS copy of the panel), relu((acc + bias) * scale + shift) on the fp32
// accumulators, the 32 x 64 result through the wave's transpose buffer (16-bit) as the B operand of the SECOND ConvT — D^T[a'b'][pixel] =
// W2^T[a'b'][co] x Z^T[co][pixel], 4 MFMAs whose A operand is the 64 x 4 weight padded to 32 rows — sigmoid, and after the four classes
// every pixel's lane stores its 4 x 4 block as four 16-byte rows.  blockIdx.y = branch (output channel).
struct Head16Params {
    const void* x[2];      // [N][Hq][Wq][64] 16-bit: relu(bn(conv3x3)) of the binarize / thresh branch
    const void* wpk[2];    // dbn_convt16_pack of the branch's first ConvT
    const float* bias1[2]; // [64] or NULL
    const float* sc[2];    // [64] eval-mode BatchNorm after the first ConvT
    const float* sh[2];
    const float* w2[2];    // [64][1][2][2] fp32: the second ConvT
    const float* bias2[2]; // [1]
    float* out;            // [N][2][4 Hq][4 Wq]
    int N, Hq, Wq, M;
    unsigned x_bytes;
};

constexpr int H16_WAVES = 8;  // waves per workgroup: they share one 32 KB LDS copy of the weight panel (two workgroups = 16 waves per CU: 69 KB each)

template <int AT>
__global__ __launch_bounds__(64 * H16_WAVES) void head16_tail_eval_kernel(const Head16Params p) {
    static_assert(AT == 1 || AT == 2, "16-bit storage");
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned short smem[4 * 4 * 2 * 64 * 8 + H16_WAVES * 32 * CT_PITCH];
    const int br = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int gw = blockIdx.x * H16_WAVES + wave, nw = gridDim.x * H16_WAVES;
    u32x4_* const PANEL = reinterpret_cast<u32x4_*>(smem);
    {
        const u32x4_* Wp = reinterpret_cast<const u32x4_*>(p.wpk[br]);
        for (int i = threadIdx.x; i < 4 * 4 * 2 * 64; i += 64 * H16_WAVES) PANEL[i] = Wp[i];
    }
    __syncthreads();
    unsigned short* const T = smem + 4 * 4 * 2 * 64 * 8 + wave * 32 * CT_PITCH;
    // the second ConvT's weight as the A operand of D^T = W2^T x Z^T: row li = output sub-pixel a'b' (4 real rows), k = channel
    u32x4_ w2f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned short h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = li < 4 ? p.w2[br][(16 * t + 8 * lh + j) * 4 + li] : 0.f;
            if constexpr (AT == 2) h[j] = __builtin_bit_cast(unsigned short, (_Float16)v);
            else h[j] = (unsigned short)bf16_bits_rne(v);
        }
        w2f[t] = u32x4_{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16),
                        (unsigned)h[6] | ((unsigned)h[7] << 16)};
    }
    float b1[2], scv[2], shv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        b1[b] = p.bias1[br] ? p.bias1[br][b * 32 + li] : 0.f;
        scv[b] = p.sc[br][b * 32 + li];
        shv[b] = p.sh[br][b * 32 + li];
    }
    const float b2 = p.bias2[br][0];
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x[br]), 0, p.x_bytes, 0x00020000);
    const int HW = p.Hq * p.Wq;
    const float r_hw = 1.0f / (float)HW, r_w = 1.0f / (float)p.Wq;
    const int H4 = 4 * p.Hq, W4 = 4 * p.Wq;
    // the A fragments of block mb + nw are fetched while block mb is computed (a wave's blocks are a chain of load -> 48 MFMAs -> stores
    // otherwise, and three waves per SIMD do not cover the load: first build 0.69 ms at cfg5)
    auto load_a = [&](u32x4_ (&a_)[4], int mb_) {
        const unsigned base = (unsigned)min(mb_ * 32 + li, p.M - 1) * 128u + (unsigned)lh * 16u;  // (rows past M repeat the last pixel; never stored)
#pragma unroll
        for (int t = 0; t < 4; ++t) a_[t] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(base + (unsigned)t * 32u), 0, 0);
    };
    u32x4_ a[4], an[4];
    if (gw * 32 < p.M) load_a(a, gw);
    for (int mb = gw; mb * 32 < p.M; mb += nw) {
        const int m0 = mb * 32, nrows = min(32, p.M - m0);
        const bool more = (mb + nw) * 32 < p.M;
        if (more) load_a(an, mb + nw);
        float o[16];  // (lanes of the lower half: this pixel's 4 x 4 block, [class c = 2a + b][a'b'])
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = b1[b];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const u32x4_ bw = PANEL[((c * 4 + t) * 2 + lh) * 64 + b * 32 + li];
                    if constexpr (AT == 2)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, bw), acc[b], 0, 0, 0);
                    else
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, bw), acc[b], 0, 0, 0);
                }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    dbn_st1<AT>(T, row * CT_PITCH + b * 32 + li, dbn_affine_relu(acc[b][r], scv[b], shv[b]));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x4_ zf = *reinterpret_cast<const u32x4_*>(T + li * CT_PITCH + 16 * t + 8 * lh);  // Z[pixel li][channels 16 t + 8 lh ..+7]
                if constexpr (AT == 2)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w2f[t]), __builtin_bit_cast(f16x8, zf), acc2, 0, 0, 0);
                else
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2f[t]), __builtin_bit_cast(bf16x8, zf), acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[c * 4 + r] = __builtin_amdgcn_rcpf(1.f + __expf(-(acc2[r] + b2)));  // (rows 0..3 of D^T live in the lower half)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the transpose buffer is rewritten by the next class)
        }
        if (lh == 0 && li < nrows) {
            int n, rem, h, w_;
            divmod24(m0 + li, HW, r_hw, n, rem);
            divmod24(rem, p.Wq, r_w, h, w_);
            float* const base = p.out + (((long)n * 2 + br) * H4 + 4 * h) * W4 + 4 * w_;
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {  // output row 4h + 2a + a': columns 4w + 2b + b' = class (a, b), sub-pixel (a', b')
                const int a_ = yy >> 1, ap = yy & 1;
                const f32x4 v = {o[(2 * a_ + 0) * 4 + 2 * ap + 0], o[(2 * a_ + 0) * 4 + 2 * ap + 1], o[(2 * a_ + 1) * 4 + 2 * ap + 0],
                                 o[(2 * a_ + 1) * 4 + 2 * ap + 1]};
                *reinterpret_cast<f32x4*>(base + (long)yy * W4) = v;
            }
        }
        if (more) {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = an[t];
        }
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

// ---- inference: both branches of the DB head's tail in one launch (see head16_tail_eval_kernel): out [N][2][4 Hq][4 Wq] fp32 =
// sigmoid(ConvT2(relu(bn(ConvT1(x) + bias1))) + bias2) per branch; x_*: [N][Hq][Wq][64] in the activation type `at`; panel_*: dbn_convt16_pack of
// the first ConvT; scale / shift: the eval-mode BatchNorm coefficients behind it; w2_*: the second ConvT's weight [64][1][2][2], bias2_* [1].
int dbn_head16_eligible(int at, int N, int Hq, int Wq) {
    return dbn_convt16_eligible(at, N, Hq, Wq, 64, 64) && (long)N * Hq * Wq * 32 < (1L << 31);
}
int dbn_head16_tail_eval_t(int at, const void* x_b, const void* x_t, const void* panel_b, const void* panel_t, const float* bias1_b,
                           const float* bias1_t, const float* scale_b, const float* shift_b, const float* scale_t, const float* shift_t,
                           const float* w2_b, const float* w2_t, const float* bias2_b, const float* bias2_t, float* out, int N, int Hq, int Wq,
                           void* stream) {
    DBN_REQUIRE(x_b && x_t && panel_b && panel_t && scale_b && shift_b && scale_t && shift_t && w2_b && w2_t && bias2_b && bias2_t && out);
    DBN_REQUIRE(dbn_head16_eligible(at, N, Hq, Wq));
    Head16Params p;
    p.x[0] = x_b; p.x[1] = x_t; p.wpk[0] = panel_b; p.wpk[1] = panel_t; p.bias1[0] = bias1_b; p.bias1[1] = bias1_t;
    p.sc[0] = scale_b; p.sc[1] = scale_t; p.sh[0] = shift_b; p.sh[1] = shift_t; p.w2[0] = w2_b; p.w2[1] = w2_t;
    p.bias2[0] = bias2_b; p.bias2[1] = bias2_t; p.out = out;
    p.N = N; p.Hq = Hq; p.Wq = Wq; p.M = N * Hq * Wq;
    p.x_bytes = (unsigned)((long)p.M * 128);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(256, 2);  // two workgroups of eight waves per CU (69 KB of LDS each), both branches side by side
    if (at == 1) hipLaunchKernelGGL(head16_tail_eval_kernel<1>, grid, dim3(64 * H16_WAVES), 0, st, p);
    else hipLaunchKernelGGL(head16_tail_eval_kernel<2>, grid, dim3(64 * H16_WAVES), 0, st, p);
    return dbn_status();
}

}  // extern "C"
