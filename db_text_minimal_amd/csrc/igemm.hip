// Implicit-GEMM convolution kernels for gfx950 (MI355X).
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
#include "common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include <algorithm>

namespace {

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
    int stat_rows;      // rows of the partials array (all M-tiles of the call; a call over many images runs as several launches)
    int stat_row0;      // first row this launch writes
    float* stats;       // optional BatchNorm partials [3][Cd][stat_rows] (pivot, sum, sum sq) + [stat_rows] counts
    unsigned src_bytes;
    unsigned plane_bytes;  // AT = 3: distance between the three bf16 planes of src (0 otherwise)
    // MODE 2 only: per class its number of tiles, first M-tile index, weight-panel offset (floats)
    int tile_end[MAX_CLASSES], row_base[MAX_CLASSES], wpk_off[MAX_CLASSES];
    // split-K (MODE 0/1, Cs % 16 == 0): workgroup row blockIdx.y reduces k-tiles [y*kt_per, (y+1)*kt_per) into slab y of dst
    int ksplit, kt_per;
    int launch_rows;    // host only: M-tiles of this launch (set by launch_igemm_ns)
    int patch;          // host only: use the pixel-patch form (3x3, stride 1; see igemm_dispatch)
    // MODE 3 only (pyramid conv): level g source [N, Hdf>>g, Wdf>>g, Cs] and its stride-2^g transposed-conv panels
    const void* seg_src[4];
    const float* seg_wpk[4];
    unsigned seg_bytes[4];
    unsigned seg_plane_bytes[4];  // AT = 3
};

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

// MODE 0: hs = hd*stride - pad + r (forward conv, stride 1 or 2)
// MODE 1: hs = hd + pad - r        (stride-1 data gradient)
// MODE 2: like MODE 1 per output-parity class, dst pixel = (f*hd+oh0, f*wd+ow0) (stride-f data
//         gradient / ConvTranspose2d forward, f = 2, 4, 8)
// MODE 3: pyramid conv: dst = sum over levels g = 0..3 of the transposed conv (k = 2^g + 2, stride 2^g, pad 1) of
//         seg_src[g] — a 3x3 conv over the concatenation of four nearest-upsampled maps without the
//         concatenation.  One workgroup owns a tile of one pixel class mod 8 and walks the four levels' taps
//         in one accumulator: K = Cs * (9 + 4 + {1,2,4} + {1,2,4}) instead of 36 * Cs.
// The gather is branch-free: an invalid tap (padding, M or K tail) gets an out-of-range buffer
// offset, for which the hardware returns zeros.
// Register budget: the 128x128 fp32 tile needs 88 VGPR + 64 AGPR = 3 waves per SIMD.  Asking for 4 waves
// (amdgpu_waves_per_eu) makes the compiler fit it into the unified 128 registers; with a prefetch distance of one k-tile
// that cost 4 spills and gave +1.6 % on large launches, with the second register set of the two-tile prefetch it spills 56
// and loses 35 % — so the default is off (DBN_IGEMM_W4=1 re-enables the experiment).
#ifndef DBN_IGEMM_W4
#define DBN_IGEMM_W4 0
#endif
#if DBN_IGEMM_W4
#define DBN_IGEMM_OCC(BM, BN, NS, MODE, PATCH) __attribute__((amdgpu_waves_per_eu(((BM) == 128 && (BN) == 128 && (NS) == 0 && (MODE) < 3) ? 4 : 1, 8)))
#else
// pixel-patch kernels with three planes: two waves per SIMD (<= 256 registers; the fully unrolled nine stages had taken 257)
#define DBN_IGEMM_OCC(BM, BN, NS, MODE, PATCH) __attribute__((amdgpu_waves_per_eu(((PATCH) && (NS) == 3 && (BN) == 64) ? 2 : 1, 8)))
#endif

// AT (activation storage type of src and dst): 0 fp32; 1 bf16 / 2 fp16 need NS = 1 — the gather then fetches 16-byte pieces
// of EIGHT stored 16-bit channels that go to LDS unchanged (no conversion, the LDS image of the NS = 1 path is exactly the
// stored format), the accumulators are rounded to the storage type on the way out.
// AT = 3 (NS = 3): src is the PRE-SPLIT form of an fp32 tensor — three bf16 planes [3][N,H,W,C] with a0 + a1 + a2 == a exactly
// (dbn_split3) — gathered the same way (3 x 2 pieces per row and k-tile, no conversion: splitting at staging time redid the
// split for every one of the 9 taps that re-reads an element and made the bf16x3 kernels VALU-bound); dst is fp32.
// PATCH (3x3, stride 1, pad 1, 16-bit matrix math; Hd % 8 == 0, Wd % 16 == 0, Cs % 32 == 0): the M tile is an 8 x 16 PIXEL PATCH
// and the A operand is not gathered per tap at all — see the main loop.
template <int BM, int BN, int WM, int WN, int MODE, int NS, int AT = 0, bool PATCH = false>
__global__ __launch_bounds__(WM* WN * 64) DBN_IGEMM_OCC(BM, BN, NS, MODE, PATCH) void igemm_f32_kernel(const IgemmParams p) {
    static_assert(AT == 0 || ((AT == 1 || AT == 2) && NS == 1) || (AT == 3 && NS == 3), "storage type / matrix math combination");
    static_assert(!PATCH || (BM == 128 && WM == 2 && WN == 2 && MODE < 2 && NS > 0 && AT != 3), "patch form");
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int ES = AT == 0 ? 4 : 2;            // bytes per stored source element
    constexpr bool DST_F32 = AT == 0 || AT == 3;
    constexpr int NP = AT == 3 ? 3 : 1;            // 16-bit planes of the source
    constexpr int A_SH = AT == 0 ? 2 : 1;          // pieces per row and plane = 1 << A_SH (4 x 4 fp32 channels, or 2 x 8 16-bit channels)
    constexpr int A_CH = AT == 0 ? 4 : 8;          // channels per 16-byte piece
    constexpr int A_PIECES = (BM << A_SH) * NP;    // 16-byte pieces of the A panel per k-tile
    constexpr int A_LD = (A_PIECES + NT - 1) / NT;  // gathers per thread per k-tile
    constexpr bool A_FULL = A_PIECES % NT == 0;
    // LDS image in 16-byte units.  NS == 0: [k/4][row][4 f32], chunk stride +2 keeps ds_write_b128 conflict-free.
    // NS > 0: per split [k/8][row][8 bf16], row stride +4 (== 64 B mod 128) keeps the ds_write_b64 conflict-free.
    constexpr int AS = NS == 0 ? BM + 2 : BM + 4, BS = NS == 0 ? BN + 2 : BN + 4;
    constexpr int A_IMG = NS == 0 ? 4 * AS : NS * 2 * AS, B_IMG = NS == 0 ? 4 * BS : NS * 2 * BS;
    constexpr int B_PIECES = NS == 0 ? 4 * BN : NS * 2 * BN;  // 16-byte pieces of the weight panel per k-tile
    constexpr int B_LD = (B_PIECES + NT - 1) / NT;
    constexpr bool B_FULL = B_PIECES % NT == 0;  // every thread owns B_LD pieces
    // K advances 16 per UNIT; KU units share one barrier (1 or 4).  Measured on the 16-bit storage paths, whose single MFMA per
    // accumulator and unit (32 cycles) is far shorter than a unit's staging / address walk: KU = 4 is SLOWER (1116 vs 1199 images/s
    // in native bf16; 51 KB of LDS and a second register set of 4 units cost occupancy, and the barrier was not what bounds them —
    // the instruction stream per unit is).  Every path runs with one unit per barrier.
    constexpr int KU = 1;
    constexpr int UNIT = A_IMG + B_IMG;
    constexpr int STAGE = KU * UNIT;
    constexpr int NSX = NS > 0 ? NS : 1;
    static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for the workgroup");
    // 16-bit storage (AT != 0): LDS-DMA ring (see the main loop): stages of DMA_SU units, unpadded images [plane][k/8][row]
    constexpr int DMA_SU = 2;
    constexpr int DMA_UNIT = NP * 2 * BM + NSX * 2 * BN;  // 16-byte slots of one unit
    constexpr int DMA_STAGE = DMA_SU * DMA_UNIT;
    constexpr int DMA_NSTG = DMA_STAGE * 16 * 4 <= 64 * 1024 ? 4 : DMA_STAGE * 16 * 3 <= 160 * 1024 ? 3 : 2;
    // PATCH: two patch buffers [plane][4 k/8 slices][10 x 18 pixels] + a ring of P_NSTG weight stages of two units
    constexpr int P_PATCH = NSX * 4 * 180;
    constexpr int P_BUNIT = NSX * 2 * BN, P_BSTAGE = 2 * P_BUNIT;
    constexpr int P_NSTG = NSX == 1 ? 4 : 3;
    // three planes: ONE patch buffer (refilled between two barriers at a channel-block boundary) keeps the workgroup at 70 KB
    // so that two fit a CU; with a second buffer it was alone on its CU (103 KB, one wave per SIMD: every LDS latency exposed)
    constexpr int P_NBUF = NSX == 1 ? 2 : 1;
    __shared__ f32x4 smem[PATCH ? P_NBUF * P_PATCH + P_NSTG * P_BSTAGE : AT != 0 ? DMA_NSTG * DMA_STAGE : 2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    int tile = dbn_xcd_remap(blockIdx.x, gridDim.x);
    IgemmClass q;
    int q_row_base = 0, q_wpk_off = 0;
    if (MODE == 3) {
        q.Hd = p.Hdf >> 3; q.Wd = p.Wdf >> 3; q.M = p.N * q.Hd * q.Wd;
        // tile order: m-tile major, pixel class minor — the 64 classes of one image region run together, so their
        // overlapping 3x3 neighbourhoods of the finest level are served by the L2 instead of 9 trips to HBM
        const int mtiles = (q.M + BM - 1) / BM, ntn_ = p.Cd / BN;
        const int mt_ = tile / (64 * ntn_), rem_ = tile - mt_ * (64 * ntn_);
        const int c = rem_ / ntn_;
        tile = mt_ * ntn_ + (rem_ - c * ntn_);
        q.oh0 = c >> 3; q.ow0 = c & 7;
        q_row_base = c * mtiles;
        q.R = q.S = q.K = q.KT = q.pad_h = q.pad_w = 0;  // per level, see level_setup
    } else if (MODE == 2) {
        // tile order: position major, class minor.  The classes differ in taps (4/2/2/1 of a 3x3 at stride 2), so a
        // class-major order would hand the heavy class to two of the eight XCDs (dbn_xcd_remap gives each XCD a contiguous
        // run) — measured 1.8x slower; interleaved, every XCD gets the same mix and the classes of one image region share
        // their source pixels through L2.  Classes with fewer tiles than the largest leave their slot empty.
        // The class of slot t rotates with t / (8 * ncls): workgroups reach a CU round-robin (every 32nd of an XCD's run),
        // and without the rotation each CU would again see a single class.
        const int slot = tile / p.ncls;
        const int c = (tile + (slot >> 3)) % p.ncls;
        tile = slot;
        if (tile >= p.tile_end[c]) return;
        q = class_geom(c, p.stride, p.R, p.S, p.pad, p.N, p.Hdf, p.Wdf, p.Cs);
        q_row_base = p.row_base[c];
        q_wpk_off = p.wpk_off[c];
    } else {
        q.Hd = p.Hdf; q.Wd = p.Wdf; q.M = p.N * p.Hdf * p.Wdf; q.R = p.R; q.S = p.S;
        q.K = p.R * p.S * p.Cs; q.KT = (q.K + 15) / 16;
        q.pad_h = q.pad_w = p.pad; q.oh0 = q.ow0 = 0;
    }
    const int ntn = p.Cd / BN;
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int qM = q.M, qHd = q.Hd, qWd = q.Wd;
    int pn = 0, ph0 = 0, pw0 = 0;  // PATCH: image and top-left output pixel of this tile
    if constexpr (PATCH) {
        const int tw = qWd >> 4, tpi = (qHd >> 3) * tw;
        pn = mt / tpi;
        const int t = mt - pn * tpi, ty = t / tw;
        ph0 = ty * 8;
        pw0 = (t - ty * tw) * 16;
    }
    int qK = q.K, qKT = q.KT, qS = q.S, qR = q.R;  // MODE 3 changes these (and the source) per level
    int gHs = p.Hs, gWs = p.Ws;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);

    // ---- per-thread gather state: A_LD rows, one 4-channel chunk -------------------
    // piece q = tid + j*NT of the k-tile: fp32 source: chunk q & 3 of row q >> 2; 16-bit source: half q & 1 of row (q >> 1) % BM
    // of plane (q >> 1) / BM
    const int a_chunk = tid & ((1 << A_SH) - 1);
    auto a_row = [&](int j) { return AT == 0 ? (tid >> 2) + j * (NT / 4) : ((tid + j * NT) >> 1) % BM; };
    auto a_plane = [&](int j) { return AT == 0 ? 0 : ((tid + j * NT) >> 1) / BM; };
    auto a_on = [&](int j) { return A_FULL || tid + j * NT < A_PIECES; };  // (the last j of tiles whose piece count is not a multiple of NT)
    int a_hb[A_LD], a_wb[A_LD], a_nb[A_LD];
    unsigned a_pl[A_LD];  // byte offset of the piece's plane
    unsigned plane_bytes = p.plane_bytes;
    int a_n[MODE == 3 ? A_LD : 1], a_hd[MODE == 3 ? A_LD : 1], a_wd[MODE == 3 ? A_LD : 1];
    const int HWd = qHd * qWd;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int row = a_row(j);
        const int m = m0 + row;
        const bool ok = m < qM && a_on(j);
        a_pl[j] = (unsigned)a_plane(j) * plane_bytes;
        const int mm = ok ? m : 0;
        int n, rem, hd, wd;  // reciprocal divisions (exact below 2^24): an integer division costs ~35 VALU instructions
        divmod24(mm, HWd, 1.0f / (float)HWd, n, rem);
        divmod24(rem, qWd, 1.0f / (float)qWd, hd, wd);
        a_nb[j] = n * p.Hs * p.Ws * p.Cs;
        if (MODE == 3) {
            a_n[j] = n;
            a_hd[j] = ok ? hd : -(1 << 20);
            a_wd[j] = wd;
        } else if (MODE == 0) {
            a_hb[j] = ok ? hd * p.stride - q.pad_h : -(1 << 20);  // a far-away row can never be in range
            a_wb[j] = wd * p.stride - q.pad_w;
        } else {
            a_hb[j] = ok ? hd + q.pad_h : -(1 << 20);
            a_wb[j] = wd + q.pad_w;
        }
    }
    // K order (must match the weight panels, pack_weights_kernel):
    //   Cs % 16 == 0: k = ((cb*R + r)*S + s)*16 + cl with ci = 16*cb + cl — every k-tile is one tap of one
    //                 16-channel block and the R*S taps of a block are consecutive k-tiles, so the 9 re-reads of
    //                 an input pixel's 64-byte slice happen back to back (L1/L2 hits instead of a trip to the fabric);
    //   otherwise (stem, Cs = 4): k = (r*S + s)*Cs + ci.
    // (always, for 16-bit storage and pre-split planes — checked on the host: a compile-time fact there, which removes the
    // non-blocked walk and its branches from the k-loop)
    const bool blocked = AT != 0 || (p.Cs & 15) == 0;
    int kidx = A_CH * a_chunk;
    int k_ci, k_r, k_s;
    int kt_begin = 0, kt_end = qKT;
    if (MODE < 2 && p.ksplit > 1) {
        kt_begin = blockIdx.y * p.kt_per;
        kt_end = min(qKT, kt_begin + p.kt_per);
    }
    if (blocked) {  // k-tile kt is tap (kt % RS) of channel block (kt / RS)
        const int rs = max(1, q.R * qS), cb = kt_begin / rs, tap = kt_begin - cb * rs;
        kidx += 16 * kt_begin;
        k_ci = 16 * cb + A_CH * a_chunk;
        k_r = tap / qS;
        k_s = tap - k_r * qS;
    } else {
        const int k_tap = kidx / p.Cs;
        k_ci = kidx - k_tap * p.Cs;
        k_r = k_tap / qS;
        k_s = k_tap - k_r * qS;
    }

    unsigned aoff[KU][A_LD];
    int kend = (MODE < 2 && p.ksplit > 1) ? min(qK, 16 * kt_end) : qK;  // units past the end (of K, or of this split's range) gather zeros
    auto next_offsets = [&](int u = 0) {  // offsets of the current k position (unit u of the barrier interval), then advance by 16 k
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int hs = MODE == 0 ? a_hb[j] + k_r : a_hb[j] - k_r;
            const int ws = MODE == 0 ? a_wb[j] + k_s : a_wb[j] - k_s;
            const bool v = kidx < kend && (unsigned)hs < (unsigned)gHs && (unsigned)ws < (unsigned)gWs;
            const unsigned off = (unsigned)(a_nb[j] + (hs * gWs + ws) * p.Cs + k_ci) * (unsigned)ES + (AT == 3 ? a_pl[j] : 0u);
            aoff[u][j] = v ? off : OOB_OFFSET;
        }
        kidx += 16;
        if (blocked) {  // next tap of the same channel block; after the last tap, the next block (branch-free)
            ++k_s;
            const bool ws_ = k_s == qS;
            k_s = ws_ ? 0 : k_s;
            k_r += ws_ ? 1 : 0;
            const bool wr_ = k_r == qR;
            k_r = wr_ ? 0 : k_r;
            k_ci += wr_ ? 16 : 0;
        } else {
            k_ci += 16;
            while (k_ci >= p.Cs) {  // stem (Cs = 4): several taps per step
                k_ci -= p.Cs;
                if (++k_s == qS) {
                    k_s = 0;
                    ++k_r;
                }
            }
        }
    };
    const f32x4* bptr[B_LD];
    int b_lds[B_LD];
    bool b_on[B_LD];
    auto panel_setup = [&](const float* panel) {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int idx = tid + j * NT;
            b_on[j] = B_FULL || idx < B_PIECES;
            const int c = b_on[j] ? idx / BN : 0, n = idx - (idx / BN) * BN;  // c: k-chunk (NS==0) or split*2+k8 (NS>0)
            bptr[j] = reinterpret_cast<const f32x4*>(panel) + (long)c * p.Cd + n0 + n;
            b_lds[j] = c * BS + n;
        }
    };
    const long b_step = (NS == 0 ? 4L : 2L * NS) * p.Cd;  // 16-byte pieces per k-tile
    if (MODE != 3) {
        panel_setup(p.wpk + q_wpk_off);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) bptr[j] += kt_begin * b_step;
    }
    // MODE 3: source, tap geometry and weight panel of pyramid level g for this tile's pixel class
    auto level_setup = [&](int g) {
        const int f = 1 << g, kk = f + 2;
        const int oh0g = q.oh0 & (f - 1), ow0g = q.ow0 & (f - 1);
        const int ph = (oh0g + 1) & (f - 1), pw = (ow0g + 1) & (f - 1);
        qR = taps_of_class(kk, ph, f);
        qS = taps_of_class(kk, pw, f);
        const int padh = (oh0g + 1 - ph) >> g, padw = (ow0g + 1 - pw) >> g;
        qK = qR * qS * p.Cs;
        qKT = qK >> 4;
        gHs = p.Hdf >> g;
        gWs = p.Wdf >> g;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.seg_src[g]), 0, p.seg_bytes[g], 0x00020000);
        if (AT == 3) {
            plane_bytes = p.seg_plane_bytes[g];
#pragma unroll
            for (int j = 0; j < A_LD; ++j) a_pl[j] = (unsigned)a_plane(j) * plane_bytes;
        }
        long krows = 0;  // padded-K rows of the classes packed before (ph, pw)
        for (int d = 0; d < ph * f + pw; ++d) krows += taps_of_class(kk, d >> g, f) * taps_of_class(kk, d & (f - 1), f) * p.Cs;
        panel_setup(p.seg_wpk[g] + (NS == 0 ? krows * p.Cd : krows * p.Cd * NS / 2));
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            a_nb[j] = a_n[MODE == 3 ? j : 0] * gHs * gWs * p.Cs;
            a_hb[j] = a_hd[MODE == 3 ? j : 0] * (8 >> g) + (q.oh0 >> g) + padh;
            a_wb[j] = a_wd[MODE == 3 ? j : 0] * (8 >> g) + (q.ow0 >> g) + padw;
        }
        kidx = A_CH * a_chunk;
        k_ci = A_CH * a_chunk;
        k_r = k_s = 0;
    };

    // two register sets: the global loads of k-tile t+2 are issued while tile t is multiplied and tile t+1 (loaded one
    // iteration earlier) is staged to LDS — a full k-step (~1 us) more latency tolerance than a prefetch distance of one
    // Loads and staging are issued UNCONDITIONALLY every k-step (tiles past the end gather zeros through out-of-range buffer
    // offsets and re-read the last weight tile): with a conditional issue the compiler merges the "issued" and "not issued"
    // paths and waits vmcnt(0) before staging — i.e. also for the set that was just issued — which defeats the distance of two.
    f32x4 ra_[2][KU][A_LD], rb_[2][KU][B_LD];
    int b_left = 0;  // weight k-tiles that remain beyond the one bptr points at
    auto issue_loads = [&](auto SET) {
        constexpr int st_ = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) ra_[st_][u][j] = buffer_load_f32x4(rsrc, aoff[u][j]);
            const long adv = b_left > 0 ? b_step : 0;
            --b_left;
#pragma unroll
            for (int j = 0; j < B_LD; ++j) {
                if (B_FULL || b_on[j]) rb_[st_][u][j] = *bptr[j];
                bptr[j] += adv;
            }
        }
    };
    auto stage_unit = [&](int buf, auto SET, auto UU) {
        constexpr int st_ = decltype(SET)::value;
        constexpr int u_ = decltype(UU)::value;
        f32x4 (&ra)[A_LD] = ra_[st_][u_];
        f32x4 (&rb)[B_LD] = rb_[st_][u_];
        f32x4* As = smem + buf * STAGE + u_ * UNIT;
        f32x4* Bs = As + A_IMG;
        if constexpr (NS == 0) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) As[a_chunk * AS + (tid >> 2) + j * (NT / 4)] = ra[j];
        } else if constexpr (AT != 0) {
            // stored 16-bit channels: the piece IS the LDS slot [plane][k/8 = a_chunk][row] of the image
#pragma unroll
            for (int j = 0; j < A_LD; ++j)
                if (a_on(j)) As[(a_plane(j) * 2 + a_chunk) * AS + a_row(j)] = ra[j];
        } else {
            // chunk c holds k = 4c..4c+3 of the k-tile: bf16 image slot [c>>1][row], 8-byte half (c&1)
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                u32x2 sp[NSX];
                split4<NS>(ra[j], sp);
                const int row = (tid >> 2) + j * (NT / 4);
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    reinterpret_cast<u32x2*>(As + (t * 2 + (a_chunk >> 1)) * AS + row)[a_chunk & 1] = sp[t];
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            if (B_FULL || b_on[j]) Bs[b_lds[j]] = rb[j];
    };
    auto stage = [&](int buf, auto SET) {
        stage_unit(buf, SET, std::integral_constant<int, 0>{});
        if constexpr (KU > 1) {
            stage_unit(buf, SET, std::integral_constant<int, 1>{});
            stage_unit(buf, SET, std::integral_constant<int, 2>{});
            stage_unit(buf, SET, std::integral_constant<int, 3>{});
        }
    };
    static_assert(KU == 1 || KU == 4, "stage() spells the units out");

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if constexpr (PATCH) {
    // ---- 3x3 / stride 1 on the bf16 matrix pipe: pixel-patch tiles -------------------------------------------------------------
    // The texture unit handles about one 128-byte line per clock per CU whatever the lanes take from it (tools/probes/
    // gather_rate.hip), and an im2col gather touches one line per row and tap: at one 32-cycle MFMA per accumulator and unit that
    // gather, not the matrix pipe, bounds the kernel (4 x 35-71 clocks per unit against 64).  Here a tile is an 8 x 16 patch of
    // output pixels, and the 10 x 18 input patch around it is brought to LDS ONCE per 32-channel block (contiguous 64/128-byte
    // runs per pixel; fp32 sources are split into their bf16 planes on the way, once instead of once per tap).  The nine taps are
    // then plain LDS address offsets of the fragment reads: image [plane][k/8 slice][patch pixel][16 B], and MFMA row i is the
    // pixel (y, x) = (2*blk + parity(i >> 2), 4*(i >> 3) + (i & 3)) so that each 16-lane group of a ds_read_b128 (lanes
    // {0-3,12-15,20-27}, {4-11,16-19,28-31}) reads 16 consecutive pixels of one patch row — conflict-free for every tap.
    // The weight panels stream through a ring of DMA stages as in the generic 16-bit loop below.
    constexpr int PPX = 180, PROW = 18;
    constexpr int CHUNKS = AT == 0 ? 8 : 4;           // 16-byte pieces per pixel of a 32-channel block
    constexpr int PL = (PPX * CHUNKS + NT - 1) / NT;  // pieces per thread
    constexpr int B_I = NSX * 2 * (BN / 64);          // DMA instructions per unit
    static_assert((2 * B_I) % 4 == 0 && NT == 256, "weight DMA is dealt evenly to four waves");
    constexpr int PWB = 2 * B_I / 4;
    f32x4* const patch = smem;
    f32x4* const ring = smem + P_NBUF * P_PATCH;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    unsigned poff[PL];
    int pslot[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int idx = tid + j * NT;
        const bool on = idx < PPX * CHUNKS;
        const int chunk = idx % CHUNKS, pix = on ? idx / CHUNKS : 0;
        const int py = pix / PROW, px = pix - py * PROW;
        const int hs = ph0 - 1 + py, ws = pw0 - 1 + px;
        const bool v = on && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
        poff[j] = v ? (unsigned)(((pn * p.Hs + hs) * p.Ws + ws) * p.Cs) * (unsigned)ES + (unsigned)chunk * 16u : OOB_OFFSET;
        // fp32 source: chunk = 4 channels = one 8-byte half of slice chunk >> 1 (slot in 8-byte units); 16-bit: chunk = slice
        pslot[j] = !on ? -1 : AT == 0 ? ((chunk >> 1) * PPX + pix) * 2 + (chunk & 1) : chunk * PPX + pix;
    }
    f32x4 pr[PL];
    const int ncb = p.Cs >> 5;
    auto load_patch = [&](int cb) {  // (cb == ncb: the loads are issued all the same, so that the counted waits stay constant)
        const unsigned add = (unsigned)(cb * 32 * ES);
#pragma unroll
        for (int j = 0; j < PL; ++j) pr[j] = buffer_load_f32x4(rsrc, poff[j] == OOB_OFFSET ? OOB_OFFSET : poff[j] + add);
    };
    auto store_patch = [&](int buf) {
        f32x4* const P = patch + buf * P_PATCH;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            if (pslot[j] < 0) continue;
            if constexpr (AT == 0) {
                u32x2 sp[NSX];
                split4<NS>(pr[j], sp);
#pragma unroll
                for (int t = 0; t < NS; ++t) reinterpret_cast<u32x2*>(P + t * 4 * PPX)[pslot[j]] = sp[t];
            } else {
                P[pslot[j]] = pr[j];
            }
        }
    };
    // weight DMA: instruction t = wave + 4 i of a stage: unit t / B_I, (plane, slice) and 64-column group from t % B_I
    int b_lds[PWB], b_u[PWB];
    unsigned b_add[PWB];
#pragma unroll
    for (int i = 0; i < PWB; ++i) {
        const int t = wave_u + 4 * i, u = t / B_I, rr = t - u * B_I, c = rr & 1, gp = rr >> 1;
        const int g = gp % (BN / 64), plane = gp / (BN / 64);
        b_u[i] = u;
        b_lds[i] = u * P_BUNIT + (plane * 2 + c) * BN + 64 * g;
        b_add[i] = (unsigned)((plane * 2 + c) * p.Cd + n0 + 64 * g + lane) * 16u;
    }
    const unsigned bstep_bytes = (unsigned)(2 * NSX * p.Cd) * 16u;
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk + q_wpk_off), 0,
                                                                           (unsigned)qKT * bstep_bytes, 0x00020000);
    int g_kt = 0;  // next k-tile to fetch (two per stage; k-tiles past the end are out of range: zeros)
    auto issue_b = [&](int slot_) {
#pragma unroll
        for (int i = 0; i < PWB; ++i) {
            auto* dst = (__attribute__((address_space(3))) void*)(ring + slot_ * P_BSTAGE + b_lds[i]);
            const int kt = g_kt + b_u[i];
            const unsigned off = kt < qKT ? (unsigned)kt * bstep_bytes + b_add[i] : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, dst, 16, (int)off, 0, 0, 0);
        }
        g_kt += 2;
    };
    // this lane's patch pixel for accumulator block a (rows 2*(wm*MI + a) .. +1 of the tile), before the tap offset
    const int q4 = li >> 2;
    const int a_pix = (2 * wm * MI + (__builtin_popcount(q4) & 1)) * PROW + (q4 >> 1) * 4 + (li & 3) + lh * PPX;

    load_patch(0);
#pragma unroll
    for (int s_ = 0; s_ < P_NSTG - 1; ++s_) issue_b(s_);
    store_patch(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int slot = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const f32x4* const P = patch + (P_NBUF == 2 ? (cb & 1) : 0) * P_PATCH;
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            // stage `st` of this channel block has landed once only the younger stages — and, for the first P_NSTG - 1 stages
            // after their issue, the next block's patch loads — are outstanding
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((P_NSTG - 2) * PWB + ((st >= 1 && st <= P_NSTG - 1) ? PL : 0)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");  // (the barrier builtin is no compiler fence: keep the LDS reads of this stage behind it)
            issue_b(slot == 0 ? P_NSTG - 1 : slot - 1);
            if (st == 0) {
                if constexpr (P_NBUF == 1) {
                    if (cb > 0) {  // everyone is past the barrier above, i.e. done with the previous block's patch: refill it
                        store_patch(0);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                }
                // the counted waits below assume the patch loads are YOUNGER than this interval's weight stage (vmcnt retires in
                // order): keep the compiler from hoisting them above the DMA instructions
                asm volatile("" ::: "memory");
                load_patch(cb + 1);
                asm volatile("" ::: "memory");
            }
            const f32x4* const Bst = ring + slot * P_BSTAGE;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ui = 2 * st + u, h = ui / 9, tap = ui - h * 9;  // compile-time after unrolling
                const int tr = MODE == 0 ? tap / 3 : 2 - tap / 3, ts = MODE == 0 ? tap % 3 : 2 - tap % 3;
                const f32x4* const Bs = Bst + u * P_BUNIT;
                bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
                for (int t = 0; t < NSX; ++t) {
#pragma unroll
                    for (int a = 0; a < MI; ++a)
                        af[t][a] = __builtin_bit_cast(bf16x8, P[(t * 4 + 2 * h) * PPX + a_pix + (2 * a + tr) * PROW + ts]);
#pragma unroll
                    for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BN + wn * TN + b * 32 + li]);
                }
                mfma_split<NSX, MI, NI, AT == 2>(af, bf, acc);
            }
            slot = slot + 1 == P_NSTG ? 0 : slot + 1;
        }
        if (P_NBUF == 2 && cb + 1 < ncb) {
            store_patch((cb + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    } else if constexpr (AT != 0) {
    // ---- stored 16-bit operands: LDS-DMA ring ------------------------------------------------------------------------------
    // One 32x32x16 MFMA per accumulator and unit is 32 cycles; a register-staged loop with a prefetch distance of one unit kept the
    // waves parked on vmcnt / the barrier for 70-80 % of their cycles (SQ_WAIT_ANY; MFMA busy 15 %).  The stored format IS the LDS
    // image ([plane][k/8][row][8 x 16 bit]), so the panels are written by buffer_load ... lds (64 lanes x 16 B = 64 rows of one
    // k/8 slice per instruction, padding taps / row tails / units past the end deliver zeros through out-of-range offsets): no
    // staging registers, hence a ring of DMA_NSTG stages of DMA_SU units with DMA_NSTG - 1 stages in flight across raw barriers
    // (counted vmcnt).  The instructions of a stage are dealt round-robin to the waves; tap / k-tile state is wave-uniform.
    constexpr int NW = NT / 64;
    // A: lanes 2j, 2j+1 fetch the two k/8 slices (32 contiguous bytes) of row j of the instruction's 32 rows — the texture unit
    // handles ~one 128-byte line per clock whatever the lanes take from it (tools/probes/gather_rate.hip: 71 clocks per instruction
    // with 64 lines, 35 with 32), and the gather, not the MFMA, bounds these kernels.  The A image is therefore row-major
    // [plane][row][2 slices]; the fragment reads (stride 32 B) pay a 2-way bank conflict for it.
    constexpr int A_I = NP * (BM / 32), B_I = NSX * 2 * (BN / 64), U_I = A_I + B_I;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && (DMA_SU * U_I) % NW == 0, "DMA instructions are dealt evenly to the waves");
    constexpr int PW = DMA_SU * U_I / NW;
    // Dealing: a wave's slot i of a stage has a COMPILE-TIME kind (A panel / weight panel) and, where a unit has at least one
    // instruction per wave, a compile-time unit — the first version dealt t = wave + 4 i round-robin, which made kind and unit
    // wave-dependent: every slot carried both code paths behind scalar branches and five scalar selects, ~100 SALU instructions
    // per stage against its 2-8 MFMAs per wave (PMC: 24-35 SALU per MFMA in the generic 16-bit kernels).  Slots [0, PA): A panel
    // (A_I % 4 == 0: unit i / RA, instruction wave*RA + i % RA of that unit; otherwise unit wave >> 1, instruction (wave & 1)*RA + i);
    // slots [PA, PW): weight panel, the same way.
    // (a unit with fewer than four instructions per panel is shared by two waves: unit wave >> 1, instructions (wave & 1)*R + j)
    static_assert(NW == 2 * DMA_SU && A_I % 2 == 0 && B_I % 2 == 0, "per-unit dealing, or two waves per unit");
    constexpr bool A_PER_UNIT = A_I % NW == 0, B_PER_UNIT = B_I % NW == 0;
    constexpr int RA = A_PER_UNIT ? A_I / NW : A_I / 2, RB = B_PER_UNIT ? B_I / NW : B_I / 2;
    constexpr int PA = A_PER_UNIT ? DMA_SU * RA : RA;
    static_assert(PA + (B_PER_UNIT ? DMA_SU * RB : RB) == PW, "slot count");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int uw = wave_u >> 1;  // the unit of a slot dealt per stage
    int i_lds[PW];       // slot of the instruction's destination inside its UNIT (16-byte units)
    unsigned i_add[PW];  // A: byte offset of the k/8 slice (+ plane) within a pixel's block; B: byte offset of the lane's piece in a k-tile
    int i_plane[PW];
    int r_hb[PW], r_wb[PW], r_nb[PW];
    int r_n[MODE == 3 ? PW : 1], r_hd[MODE == 3 ? PW : 1], r_wd[MODE == 3 ? PW : 1];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const bool isA = i < PA;
        // index of the instruction inside its unit
        const int r = isA ? (A_PER_UNIT ? wave_u * RA + i % RA : (wave_u & 1) * RA + i)
                          : (B_PER_UNIT ? wave_u * RB + (i - PA) % RB : (wave_u & 1) * RB + (i - PA));
        const int c = r & 1, gp = r >> 1;
        const int g = isA ? r % (BM / 32) : gp % (BN / 64), plane = isA ? r / (BM / 32) : gp / (BN / 64);
        i_plane[i] = plane;
        i_lds[i] = isA ? plane * 2 * BM + 64 * g : NP * 2 * BM + (plane * 2 + c) * BN + 64 * g;
        i_add[i] = isA ? (unsigned)(lane & 1) * 16u : (unsigned)((plane * 2 + c) * p.Cd + n0 + 64 * g + lane) * 16u;
        r_hb[i] = r_wb[i] = r_nb[i] = 0;
        if (MODE == 3) r_n[MODE == 3 ? i : 0] = r_hd[MODE == 3 ? i : 0] = r_wd[MODE == 3 ? i : 0] = 0;
        if (isA) {
            const int m = m0 + 32 * g + (lane >> 1);
            const bool ok = m < qM;
            int n, rem, hd, wd;
            divmod24(ok ? m : 0, HWd, 1.0f / (float)HWd, n, rem);
            divmod24(rem, qWd, 1.0f / (float)qWd, hd, wd);
            r_nb[i] = n * p.Hs * p.Ws * p.Cs;
            if (MODE == 3) {
                r_n[MODE == 3 ? i : 0] = n;
                r_hd[MODE == 3 ? i : 0] = ok ? hd : -(1 << 20);
                r_wd[MODE == 3 ? i : 0] = wd;
            } else if (MODE == 0) {
                r_hb[i] = ok ? hd * p.stride - q.pad_h : -(1 << 20);
                r_wb[i] = wd * p.stride - q.pad_w;
            } else {
                r_hb[i] = ok ? hd + q.pad_h : -(1 << 20);
                r_wb[i] = wd + q.pad_w;
            }
        }
    }
    // wave-uniform walk over the units: tap (g_r, g_s) of channel block g_cb, k-tile g_kt of the weight panel
    int g_left, g_kt, g_r, g_s, g_cb, g_level = 0;
    const unsigned bstep_bytes = (unsigned)(2 * NSX * p.Cd) * 16u;
    __amdgpu_buffer_rsrc_t rsrcB;
    auto level_dma = [&](int g) {  // MODE 3: source, tap geometry, weight panel of pyramid level g (see level_setup)
        const int f = 1 << g, kk = f + 2;
        const int oh0g = q.oh0 & (f - 1), ow0g = q.ow0 & (f - 1);
        const int ph = (oh0g + 1) & (f - 1), pw = (ow0g + 1) & (f - 1);
        qR = taps_of_class(kk, ph, f);
        qS = taps_of_class(kk, pw, f);
        const int padh = (oh0g + 1 - ph) >> g, padw = (ow0g + 1 - pw) >> g;
        qK = qR * qS * p.Cs;
        qKT = qK >> 4;
        gHs = p.Hdf >> g;
        gWs = p.Wdf >> g;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.seg_src[g]), 0, p.seg_bytes[g], 0x00020000);
        if (AT == 3) plane_bytes = p.seg_plane_bytes[g];
        long krows = 0;
        for (int d = 0; d < ph * f + pw; ++d) krows += taps_of_class(kk, d >> g, f) * taps_of_class(kk, d & (f - 1), f) * p.Cs;
        rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg_wpk[g] + krows * p.Cd * NSX / 2), 0, (unsigned)qKT * bstep_bytes,
                                                  0x00020000);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            r_nb[i] = r_n[MODE == 3 ? i : 0] * gHs * gWs * p.Cs;
            r_hb[i] = r_hd[MODE == 3 ? i : 0] * (8 >> g) + (q.oh0 >> g) + padh;
            r_wb[i] = r_wd[MODE == 3 ? i : 0] * (8 >> g) + (q.ow0 >> g) + padw;
        }
        g_left = qKT;
        g_kt = g_r = g_s = g_cb = 0;
    };
    int nstages;
    if (MODE == 3) {
        nstages = 0;
        for (int g = 0; g < 4; ++g) {
            const int f = 1 << g, kk = f + 2;
            const int ph = ((q.oh0 & (f - 1)) + 1) & (f - 1), pw = ((q.ow0 & (f - 1)) + 1) & (f - 1);
            nstages += (taps_of_class(kk, ph, f) * taps_of_class(kk, pw, f) * (p.Cs >> 4) + DMA_SU - 1) / DMA_SU;
        }
        level_dma(0);
    } else {
        const int rs = max(1, q.R * qS), cb = kt_begin / rs, tap = kt_begin - cb * rs;
        g_cb = 16 * cb;
        g_r = tap / qS;
        g_s = tap - g_r * qS;
        g_kt = kt_begin;
        g_left = kt_end - kt_begin;
        nstages = (g_left + DMA_SU - 1) / DMA_SU;
        rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk + q_wpk_off), 0, (unsigned)qKT * bstep_bytes, 0x00020000);
    }
    auto issue_stage = [&](int slot) {
        if (MODE == 3 && g_left <= 0 && g_level < 3) level_dma(++g_level);
        bool uv[DMA_SU];
        int ur[DMA_SU], us[DMA_SU], ucb[DMA_SU], ukt[DMA_SU];
#pragma unroll
        for (int u = 0; u < DMA_SU; ++u) {
            uv[u] = g_left > 0;
            ur[u] = g_r; us[u] = g_s; ucb[u] = g_cb; ukt[u] = g_kt;
            --g_left;
            ++g_kt;
            ++g_s;
            const bool ws_ = g_s == qS;
            g_s = ws_ ? 0 : g_s;
            g_r += ws_ ? 1 : 0;
            const bool wr_ = g_r == qR;
            g_r = wr_ ? 0 : g_r;
            g_cb += wr_ ? 16 : 0;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const bool isA = i < PA;  // compile-time after unrolling, like `fixed` and `uc`
            const bool fixed = isA ? A_PER_UNIT : B_PER_UNIT;
            const int uc = isA ? i / RA : (i - PA) / RB;  // the unit of a per-unit slot
            const int u = fixed ? uc : uw;
            const bool v_u = fixed ? uv[uc] : (uw ? uv[DMA_SU - 1] : uv[0]);
            const int tr = fixed ? ur[uc] : (uw ? ur[DMA_SU - 1] : ur[0]), ts = fixed ? us[uc] : (uw ? us[DMA_SU - 1] : us[0]);
            const int tcb = fixed ? ucb[uc] : (uw ? ucb[DMA_SU - 1] : ucb[0]), tkt = fixed ? ukt[uc] : (uw ? ukt[DMA_SU - 1] : ukt[0]);
            auto* dst = (__attribute__((address_space(3))) void*)(smem + slot * DMA_STAGE + u * DMA_UNIT + i_lds[i]);
            if (isA) {
                const int hs = MODE == 0 ? r_hb[i] + tr : r_hb[i] - tr;
                const int ws = MODE == 0 ? r_wb[i] + ts : r_wb[i] - ts;
                const bool v = v_u && (unsigned)hs < (unsigned)gHs && (unsigned)ws < (unsigned)gWs;
                const unsigned off = (unsigned)(r_nb[i] + (hs * gWs + ws) * p.Cs + tcb) * 2u + i_add[i] + (AT == 3 ? (unsigned)i_plane[i] * plane_bytes : 0u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, (int)(v ? off : OOB_OFFSET), 0, 0, 0);
            } else {
                const unsigned off = v_u ? (unsigned)tkt * bstep_bytes + i_add[i] : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, dst, 16, (int)off, 0, 0, 0);
            }
        }
    };
    static_assert(DMA_SU == 2, "issue_stage selects between two units");
#pragma unroll
    for (int s_ = 0; s_ < DMA_NSTG - 1; ++s_) issue_stage(s_);
    int slot = 0;
    for (int st_ = 0; st_ < nstages; ++st_) {
        // this wave's part of stage st_ has landed once at most the DMA_NSTG - 2 younger stages are outstanding
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PW * (DMA_NSTG - 2)) : "memory");
        __builtin_amdgcn_s_barrier();  // every wave's part is in LDS, and everyone is done reading the slot of stage st_ - 1
        asm volatile("" ::: "memory");  // (the barrier builtin is no compiler fence)
        const int fill = slot == 0 ? DMA_NSTG - 1 : slot - 1;
        issue_stage(fill);
        const f32x4* Sg = smem + slot * DMA_STAGE;
#pragma unroll
        for (int u = 0; u < DMA_SU; ++u) {
            const f32x4* As = Sg + u * DMA_UNIT;
            const f32x4* Bs = As + NP * 2 * BM;
            bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
            for (int t = 0; t < NSX; ++t) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[t][a] = __builtin_bit_cast(bf16x8, As[(t * BM + wm * TM + a * 32 + li) * 2 + lh]);
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BN + wn * TN + b * 32 + li]);
            }
            mfma_split<NSX, MI, NI, AT == 2>(af, bf, acc);
        }
        slot = slot + 1 == DMA_NSTG ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero-filled stages issued past the end
    __syncthreads();
    } else {
    for (int level = 0; level < (MODE == 3 ? 4 : 1); ++level) {
    if (MODE == 3) {
        level_setup(level);
        kt_end = qKT;
        kend = qK;
    }
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    b_left = kt_end - kt_begin - 1;
    auto offsets_of_interval = [&]() {
#pragma unroll
        for (int u = 0; u < KU; ++u) next_offsets(u);
    };
    offsets_of_interval();
    issue_loads(C0{});
    offsets_of_interval();  // offsets of interval 1
    issue_loads(C1{});
    offsets_of_interval();  // offsets of interval 2
    stage(0, C0{});
    __syncthreads();

    auto k_step = [&](int kt, auto PAR) {
        constexpr int buf = decltype(PAR)::value;  // parity of the interval: LDS buffer and register set of its tiles
        issue_loads(PAR);  // interval +2 into the register set this interval was staged from
        const f32x4* As = smem + buf * STAGE;
        const f32x4* Bs = As + A_IMG;
        if constexpr (KU > 1) {
            // four 16-k units per barrier: fragments of unit u+1 are read while unit u multiplies
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                bf16x8 af[1][MI], bf[1][NI];
#pragma unroll
                for (int a = 0; a < MI; ++a) af[0][a] = __builtin_bit_cast(bf16x8, As[u * UNIT + lh * AS + wm * TM + a * 32 + li]);
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[0][b] = __builtin_bit_cast(bf16x8, Bs[u * UNIT + lh * BS + wn * TN + b * 32 + li]);
                mfma_split<1, MI, NI, AT == 2>(af, bf, acc);
            }
            offsets_of_interval();
        } else if constexpr (NS == 0) {
            // all fragment reads of the k-tile up front: the second half's LDS latency hides under the first half's MFMAs
            f32x4 af[2][MI], bf[2][NI];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[s2][a] = As[(2 * s2 + lh) * AS + wm * TM + a * 32 + li];
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[s2][b] = Bs[(2 * s2 + lh) * BS + wn * TN + b * 32 + li];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < MI; ++a)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2][a][e], bf[s2][b][e], acc[a][b], 0, 0, 0);
                if (s2 == 0) next_offsets();  // address math of tile kt+2 in the shadow of the MFMAs
            }
        } else {
            // one 32x32x16 bf16 MFMA k-step per k-tile: lane half lh owns k = 8*lh .. 8*lh+7
            bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[t][a] = __builtin_bit_cast(bf16x8, As[(t * 2 + lh) * AS + wm * TM + a * 32 + li]);
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BS + wn * TN + b * 32 + li]);
            }
            mfma_split<NS, MI, NI, AT == 2>(af, bf, acc);
            next_offsets();
        }
        stage(buf ^ 1, std::integral_constant<int, buf ^ 1>{});
        __syncthreads();
    };
    for (int kt = kt_begin; kt < kt_end; kt += 2 * KU) {
        k_step(kt, C0{});
        if (kt + KU < kt_end) k_step(kt + KU, C1{});
    }
    }

    }
    // ---- accumulate mode: fold the previous contents of dst into the accumulators first, so that the BatchNorm
    // statistics below and the store loop both see the final values
    // split-K launches write fp32 slabs whatever the activation type (the slab sum rounds once)
    const bool to_slab = MODE < 2 && p.ksplit > 1;
    float* const slabp = reinterpret_cast<float*>(p.dst) + (to_slab ? (long)blockIdx.y * ((long)qM * p.Cd + 1088) : 0L);
    void* const dstv = p.dst;
    auto ld_dst = [&](long off) -> float { return (DST_F32 || to_slab) ? slabp[off] : dbn_ld1 < DST_F32 ? 0 : AT > (dstv, off); };
    auto st_dst = [&](long off, float v) {
        if (DST_F32 || to_slab) slabp[off] = v;
        else dbn_st1 < DST_F32 ? 0 : AT > (dstv, off, v);
    };
    const float rcp_hw = 1.0f / (float)HWd, rcp_w = 1.0f / (float)qWd;
    // fn(r, doff) for the 16 rows this lane holds of accumulator block a (rows base + (r&3) + 8*(r>>2)) that are < M.
    // MODE >= 2 scatters to the parity class's pixels of the full-resolution output: the pixel (n, hd, wd) of the first row
    // comes from two reciprocal divisions, the other 15 by stepping +1,+1,+1,+5 with carries — the per-row divisions
    // were 1300 of the 2450 VALU instructions a wave spends on a K = 64 tile (ConvTranspose 2x2), as many cycles as its MFMAs.
    // offsets are formed in 32 bits (element index < 2^31 is checked on the host) and widened once per row
    auto for_rows = [&](int a, auto&& fn) {
        const int rbase = m0 + wm * TM + a * 32 + 4 * lh;
        if constexpr (PATCH) {  // row i = (r & 3) + 8 (r >> 2) + 4 lh is the pixel (2 blk + parity(i >> 2), 4 (i >> 3) + (i & 3))
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = ph0 + 2 * (wm * MI + a) + ((__builtin_popcount(r >> 2) + lh) & 1), x = pw0 + (r >> 2) * 4 + (r & 3);
                fn(r, true, (long)((unsigned)((pn * p.Hdf + y) * p.Wdf + x) * (unsigned)p.Cd));
            }
        } else if (MODE >= 2) {
            int n, rem, hd, wd;
            divmod24(min(rbase, qM - 1), HWd, rcp_hw, n, rem);
            divmod24(rem, qWd, rcp_w, hd, wd);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r > 0) {
                    wd += (r & 3) ? 1 : 5;
                    while (wd >= qWd) {
                        wd -= qWd;
                        if (++hd == qHd) {
                            hd = 0;
                            ++n;
                        }
                    }
                }
                const bool ok = rbase + (r & 3) + 8 * (r >> 2) < qM;
                const int nn = ok ? n : 0, hh = ok ? hd : 0, ww = ok ? wd : 0;  // invalid rows point at a valid pixel
                fn(r, ok, (long)((unsigned)((nn * p.Hdf + p.stride * hh + q.oh0) * p.Wdf + p.stride * ww + q.ow0) * (unsigned)p.Cd));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const bool ok = row < qM;
                fn(r, ok, (long)((unsigned)(ok ? row : m0) * (unsigned)p.Cd));
            }
        }
    };
    // destination offset (elements) of tile row `row` (0 .. BM-1) — the row-major passes of 16-bit destinations below
    auto tile_row = [&](int row, bool& ok, long& doff) {
        if constexpr (PATCH) {
            const int blk = row >> 5, q4r = (row & 31) >> 2;
            const int y = ph0 + 2 * blk + (__builtin_popcount(q4r) & 1), x = pw0 + (q4r >> 1) * 4 + (row & 3);
            ok = true;
            doff = (long)((unsigned)((pn * p.Hdf + y) * p.Wdf + x) * (unsigned)p.Cd);
        } else if (MODE >= 2) {
            int n, rem, hd, wd;
            const int m = m0 + row;
            ok = m < qM;
            divmod24(ok ? m : 0, HWd, rcp_hw, n, rem);
            divmod24(rem, qWd, rcp_w, hd, wd);
            doff = (long)((unsigned)((n * p.Hdf + p.stride * hd + q.oh0) * p.Wdf + p.stride * wd + q.ow0) * (unsigned)p.Cd);
        } else {
            ok = m0 + row < qM;
            doff = (long)((unsigned)(m0 + row) * (unsigned)p.Cd);
        }
    };
    constexpr int T_PITCH = BN + 8;  // 16-bit elements; +16 bytes keeps the 16-byte accesses aligned and rotates the banks
    constexpr int T_LPR = BN / 8, T_RPP = NT / T_LPR;  // lanes per row (16 B each), rows per pass
    bool acc_done = false;
    if constexpr (!DST_F32) {
        if (p.accumulate && !to_slab) {
            // 16-bit destination: the old tile comes in row-major, 16 bytes per lane, through LDS — a lane holds one column of 16
            // rows, so reading its own elements directly is MI*NI*16 two-byte loads per lane (measured: a bf16 data gradient with
            // accumulate took 97 us against 58 us for the same convolution without)
            static_assert((long)BM * T_PITCH * 2 <= (long)sizeof(smem) && BM % T_RPP == 0, "tile must fit the LDS panels");
            unsigned short* const T = reinterpret_cast<unsigned short*>(smem);
            const int piece = tid % T_LPR;
#pragma unroll
            for (int ps = 0; ps < BM / T_RPP; ++ps) {
                const int row = ps * T_RPP + tid / T_LPR;
                bool ok;
                long doff;
                tile_row(row, ok, doff);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(dstv) + doff + n0 + piece * 8);
                *reinterpret_cast<f32x4*>(T + row * T_PITCH + piece * 8) = v;
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        acc[a][b][r] += dbn_ld1<DST_F32 ? 1 : AT>(T, row * T_PITCH + wn * TN + b * 32 + li);
                    }
            __syncthreads();  // (the statistics scratch and the output staging reuse the region)
            acc_done = true;
        }
    }
    if (p.accumulate && !acc_done) {
#pragma unroll
        for (int a = 0; a < MI; ++a)
        {
            if constexpr (MODE < 2) {
                // unpredicated loads (rows past M read a valid pixel and add 0), issued four rows (4 x NI loads) at a time
                // before their adds: left alone, the scheduler put each load right before its use with a full wait — 64
                // dependent round trips per tile; whole blocks in flight would cost an occupancy step in registers
                float old[4][NI];
                bool okr[4];
                for_rows(a, [&](int r, bool ok, long doff) {
                    const long d = n0 + wn * TN + li + doff;
                    okr[r & 3] = ok;
#pragma unroll
                    for (int b = 0; b < NI; ++b) old[r & 3][b] = ld_dst(d + b * 32);
                    if ((r & 3) == 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int b = 0; b < NI; ++b) acc[a][b][r - 3 + i] += okr[i] ? old[i][b] : 0.f;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            } else {  // parity-class scatter: row by row (the pixel walk plus batched loads costs 40 VGPRs = an occupancy step)
                for_rows(a, [&](int r, bool ok, long doff) {
                    if (ok) {
                        const long d = n0 + wn * TN + li + doff;
#pragma unroll
                        for (int b = 0; b < NI; ++b) acc[a][b][r] += ld_dst(d + b * 32);
                    }
                });
            }
        }
    }

    // ---- optional BatchNorm statistics of this tile (train-mode BN follows the conv): per output channel the
    // pivot (first row of the tile), sum and sum of squares of (value - pivot) over the tile's valid rows.  A
    // per-tile pivot keeps the fp32 sums free of cancellation; the finalize kernel merges tiles in fp64.
    if (p.stats) {
        float* red = reinterpret_cast<float*>(smem);  // the LDS panels are dead after the last barrier of the k-loop
        float* piv = red;                             // [BN]
        float* r1 = red + BN;                         // [WM][BN]
        float* r2 = r1 + WM * BN;                     // [WM][BN]
        if (wm == 0 && lh == 0) {
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                const int cl = wn * TN + b * 32 + li;
                piv[cl] = acc[0][b][0] + (p.bias ? p.bias[n0 + cl] : 0.f);  // row m0 (< M always)
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int cl = wn * TN + b * 32 + li;
            const float pv = piv[cl], bv = p.bias ? p.bias[n0 + cl] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float d = (acc[a][b][r] + bv) - pv;
                    s1 += row < qM ? d : 0.f;
                    s2 += row < qM ? d * d : 0.f;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                r1[wm * BN + cl] = s1;
                r2[wm * BN + cl] = s2;
            }
        }
        __syncthreads();
        const int trow = p.stat_row0 + q_row_base + mt;
        for (int cl = tid; cl < BN; cl += NT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s1 += r1[w * BN + cl];
                s2 += r2[w * BN + cl];
            }
            const long c = n0 + cl;
            p.stats[(0L * p.Cd + c) * p.stat_rows + trow] = piv[cl];
            p.stats[(1L * p.Cd + c) * p.stat_rows + trow] = s1;
            p.stats[(2L * p.Cd + c) * p.stat_rows + trow] = s2;
        }
        if (nt == 0 && tid == 0) p.stats[3L * p.Cd * p.stat_rows + trow] = (float)min(BM, qM - m0);
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // the bias values of this lane's NI columns are loaded once (inside the row loop the compiler re-loaded them for every
    // row, behind a vmcnt(0) wait, because the stores may alias them)
    float bv[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) bv[b] = p.bias ? p.bias[n0 + wn * TN + b * 32 + li] : 0.f;
    // ... and pinned in registers BEFORE the (row-predicated) store blocks: a load still pending when a predicated block
    // is entered makes the compiler wait vmcnt(0) in each of them — and on gfx9 vmcnt also counts the stores, so every
    // row's store waited for the previous row's store to complete.
#pragma unroll
    for (int b = 0; b < NI; ++b) asm volatile("" : "+v"(bv[b]));  // the loads have landed here, once
    const long dcol = n0 + wn * TN + li;
    if constexpr (!DST_F32) {
        if (!to_slab) {
            // 16-bit output: a lane holds ONE column of 16 rows, so direct stores are 2-byte scatters (MI*NI*16 store instructions
            // per lane, 64 contiguous bytes per row each) — as many texture-unit cycles as the whole k-loop of a K = 576 tile.  The
            // tile goes through LDS instead: written in the storage type, read back row-major, stored 16 bytes per lane
            // (BN/8 lanes cover a row's 2*BN contiguous bytes): BM*BN/(8*NT) store instructions per lane.
            constexpr int PITCH = T_PITCH;
            static_assert((long)BM * PITCH * 2 <= (long)sizeof(smem), "output tile must fit the LDS panels");
            unsigned short* const T = reinterpret_cast<unsigned short*>(smem);
            __syncthreads();  // the panels / the statistics scratch are dead
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        dbn_st1<DST_F32 ? 1 : AT>(T, row * PITCH + wn * TN + b * 32 + li, acc[a][b][r] + bv[b]);
                    }
            __syncthreads();
            constexpr int LPR = T_LPR, RPP = T_RPP;
            static_assert(BM % RPP == 0, "whole passes");
            const int piece = tid % LPR;
#pragma unroll
            for (int ps = 0; ps < BM / RPP; ++ps) {
                const int row = ps * RPP + tid / LPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * PITCH + piece * 8);
                bool ok;
                long doff;
                tile_row(row, ok, doff);
                if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(dstv) + doff + n0 + piece * 8) = v;
            }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
        for_rows(a, [&](int r, bool ok, long doff) {
            if (ok) {
#pragma unroll
                for (int b = 0; b < NI; ++b) st_dst(dcol + doff + b * 32, acc[a][b][r] + bv[b]);
            }
        });
}

template <int BM, int BN, int WM, int WN, int NS, int AT = 0>
int launch_igemm_ns(IgemmParams& p, int mode, hipStream_t st) {
    int grid = 0, rows = 0;
    if (mode == 3) {
        rows = 64 * dbn_ceil_div(p.N * (p.Hdf >> 3) * (p.Wdf >> 3), BM);
        grid = rows * (p.Cd / BN);
    } else if (mode == 2) {
        for (int c = 0; c < p.ncls; ++c) {
            const IgemmClass q = class_geom(c, p.stride, p.R, p.S, p.pad, p.N, p.Hdf, p.Wdf, p.Cs);
            const int mtiles = (q.K > 0 && q.M > 0) ? dbn_ceil_div(q.M, BM) : 0;
            p.row_base[c] = rows;
            rows += mtiles;
            p.tile_end[c] = mtiles * (p.Cd / BN);
            grid = p.tile_end[c] > grid ? p.tile_end[c] : grid;
        }
        grid *= p.ncls;  // class-interleaved tile order: ncls slots per position (see the kernel)
    } else {
        rows = dbn_ceil_div(p.N * p.Hdf * p.Wdf, BM);
        grid = rows * (p.Cd / BN);
    }
    if (p.stat_rows <= 0) p.stat_rows = rows;  // a chunked call sets the total itself
    p.launch_rows = rows;
    if (grid == 0) return DBN_OK;
    const int gy = (mode < 2 && p.ksplit > 1) ? p.ksplit : 1;
    if constexpr (BM == 128 && WM == 2 && WN == 2 && NS > 0 && AT != 3) {
        if (p.patch && mode < 2) {
            if (mode == 0)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT, true>), dim3(grid), dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 1, NS, AT, true>), dim3(grid), dim3(256), 0, st, p);
            return dbn_status();
        }
    }
    if (p.patch) return DBN_ERR_ARG;
    if (mode == 0)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT>), dim3(grid, gy), dim3(WM * WN * 64), 0, st, p);
    else if (mode == 1)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 1, NS, AT>), dim3(grid, gy), dim3(WM * WN * 64), 0, st, p);
    else if (mode == 2)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 2, NS, AT>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
    else if constexpr (BM == 128 && BN == 128)  // the pyramid conv is built for the 128x128 tile only
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 3, NS, AT>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
    else
        return DBN_ERR_ARG;
    return dbn_status();
}

// ns: matrix math (0 exact fp32, 1 one 16-bit plane, 3 bf16x3); at: activation storage (0 fp32, 1 bf16, 2 fp16; 16-bit storage needs ns 1)
template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmParams& p, int mode, int ns, hipStream_t st, int at = 0) {
    if (at == 1) return ns == 1 ? launch_igemm_ns<BM, BN, WM, WN, 1, 1>(p, mode, st) : DBN_ERR_ARG;
    if (at == 2) return ns == 1 ? launch_igemm_ns<BM, BN, WM, WN, 1, 2>(p, mode, st) : DBN_ERR_ARG;
    if (at == 3) return ns == 3 ? launch_igemm_ns<BM, BN, WM, WN, 3, 3>(p, mode, st) : DBN_ERR_ARG;
    if (ns == 0) return launch_igemm_ns<BM, BN, WM, WN, 0>(p, mode, st);
    if (ns == 1) return launch_igemm_ns<BM, BN, WM, WN, 1>(p, mode, st);
    return launch_igemm_ns<BM, BN, WM, WN, 3>(p, mode, st);
}

// --------------------------------------------------------------------------------
// weight gradient
// --------------------------------------------------------------------------------
// Distance (floats) between the slabs of consecutive pixel splits.  O*Jp alone is a multiple of 64 KB for most layers
// (e.g. 64 x 2304 floats = 9 x 64 KB): the reduction then reads its `splits` addends from addresses that all map to the same
// HBM channel / L2 slice and crawls (50 MB in 130 us).  4352 bytes of padding rotate consecutive slabs across the channels.
__host__ __device__ inline long wgrad_slab_stride(long O, long Jp) { return O * Jp + 1088; }

struct WgradParams {
    const void* sm;    // [N,Ho,Wo,O]  (indexes the reduction), activation type AT
    const void* big;   // [N,H,W,Cb], activation type AT
    float* slab;       // [splitk][O][J]
    int N, Ho, Wo, O, H, W, Cb, R, S, stride, pad;
    int P, J, pchunk;
    float rcp_HWo, rcp_Wo;
    unsigned sm_bytes, big_bytes;
    unsigned sm_plane_bytes, big_plane_bytes;  // AT = 3: distance of the three bf16 planes of each operand
};

// Position <-> index permutation of a tile edge of length B (B % 4 == 0): the staging threads
// transpose 4x4 blocks (4 pixels x 4 channels) in registers and write channel 4c+e to LDS position
// e*(B/4)+c, which keeps both the ds_write_b128 of the staging pass and the ds_read_b128 of the MFMA
// fragments conflict-free.  The accumulators (and the slabs) therefore live in "position space".
__host__ __device__ __forceinline__ int tile_pos_to_index(int pos, int B) { return 4 * (pos % (B / 4)) + pos / (B / 4); }

// AT = 1 (bf16 activations and gradients in HBM, NS = 1): the staging threads fetch their 4 pixels x 4 channels as four
// 8-byte loads and transpose the 16-bit values with two bit operations per output word — no conversion.
template <int BM, int BN, int WM, int WN, int NS, int AT = 0>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_f32_kernel(const WgradParams p) {
    // AT = 3 (NS = 3): both operands are pre-split fp32 tensors (three bf16 planes each, dbn_split3): as AT = 1, three times.
    static_assert(AT == 0 || (AT == 1 && NS == 1) || (AT == 3 && NS == 3), "storage type / matrix math combination");
    constexpr unsigned ES = AT == 0 ? 4u : 2u;
    constexpr int NP = AT == 3 ? 3 : 1;
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int AS = NS == 0 ? BM + 2 : BM + 4, BS = NS == 0 ? BN + 2 : BN + 4;  // strides in 16-byte units
    constexpr int A_IMG = NS == 0 ? 4 * AS : NS * 2 * AS, B_IMG = NS == 0 ? 4 * BS : NS * 2 * BS;
    constexpr int STAGE = A_IMG + B_IMG;
    constexpr int NSX = NS > 0 ? NS : 1;
    static_assert(BM + BN <= NT && BM % 64 == 0 && BN % 64 == 0, "staging roles must fit the workgroup in whole waves");
    __shared__ f32x4 smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // 1-D grid of tiles x splits.  The XCD remap gives each XCD (private L2) a contiguous run of work items,
    // ordered split-major, so all (o, j) tiles of one pixel range run on the same XCD and share dY / X in its L2.
    const int njt = (p.J + BN - 1) / BN;
    const int ntiles = (p.O / BM) * njt;
    const int work = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = work / ntiles, tile_ = work - split * ntiles;
    const int ot = tile_ / njt, jt = tile_ - ot * njt;
    const int o0 = ot * BM, j0 = jt * BN;
    const int pbeg = split * p.pchunk;
    const int pend = min(p.P, pbeg + p.pchunk);
    const int KT = (pend - pbeg + 15) / 16;

    const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sm), 0, p.sm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.big), 0, p.big_bytes, 0x00020000);

    // staging roles (wave-uniform): threads [0,BM) transpose the A panel (sm: 16 pixels x BM channels),
    // threads [NT-BN,NT) the B panel (gathered big: 16 pixels x BN (tap,channel) columns).
    // roles are whole waves (BM, BN multiples of 64): derived from a scalar so that the role branches are scalar branches
    // and the buffer descriptor of each load is provably uniform — with per-lane predicates the compiler had merged the two
    // branches and wrapped every buffer load in a readfirstlane "waterfall" loop over the descriptor.
    const int wave_first = __builtin_amdgcn_readfirstlane(tid) & ~63;
    const bool is_a = wave_first < BM;
    const bool is_b = wave_first >= NT - BN;
    const int slot = is_a ? tid : tid - (NT - BN);
    const int qn = is_a ? BM / 4 : BN / 4;
    const int s_c = slot % qn, s_g = slot / qn;  // column quad, pixel group (rows 4g..4g+3)
    // B column quad -> (tap, ci)
    const int jj = j0 + 4 * s_c;
    const bool j_ok = is_b && jj < p.J;
    const int tap = j_ok ? jj / p.Cb : 0;
    const int ci = j_ok ? jj - tap * p.Cb : 0;
    const int tr = tap / p.S - p.pad, ts = tap % p.S - p.pad;
    const int HWo = p.Ho * p.Wo;
    const int ld = is_a ? AS : BS;
    const int lds_base = (is_a ? 0 : A_IMG) + s_c;

    // Prefetch distance D (register sets).  The bf16 matrix math makes a 16-pixel k-step 96 (one plane) or 576 (three planes)
    // matrix-pipe clocks per wave; with a distance of one every k-step waited out a full memory round trip (measured: 2.05 us per
    // k-step round whatever the number of resident workgroups — more pixel splits per CU changed nothing), i.e. the kernel ran at
    // (workgroups per CU) k-steps per latency.  A stored-bf16 set is 8 registers, an fp32 one 16: D = 4 / 3 keep the occupancy.
    // Exact fp32 (NS = 0: 1536 clocks per k-step and wave): two for the 64-row tiles (120 registers, still four waves per SIMD:
    // the head convs' weight gradients 1.084 -> 1.045 ms, all weight gradients 0.648 -> 0.663 of peak, step +0.4 %), one for
    // 128 x 128 (152 registers = an occupancy step: 0.256 -> 0.279 ms).  Round 1's attempt at two had lost 28 % — with the loads
    // inside role branches the compiler drained every set each k-step (see issue_loads).
    constexpr int D = AT == 3 ? 1 : NS == 0 ? (BM == 64 ? 2 : 1) : (AT == 0 ? 3 : 4);
    f32x4 rr_[D][4];        // AT = 0: 4 pixels x 4 fp32 channels
    u32x2 rh_[D][NP][4];    // AT = 1 / 3: per plane 4 pixels x 4 bf16 channels
    unsigned woff[4] = {OOB_OFFSET, OOB_OFFSET, OOB_OFFSET, OOB_OFFSET};
    // byte offsets of this thread's 4 loads for k-tile kt (address math kept apart from the loads so that it
    // can be issued in the shadow of the previous tile's MFMAs)
    // B role: pixel (n, oh, ow) of this thread's first row in the current k-tile, advanced by 16 pixels per call (offsets()
    // is called for k-tiles 0, 1, 2, ... in order) instead of two divisions per k-tile
    int w_n, w_oh, w_ow;
    {
        int rem;
        divmod24(pbeg + 4 * s_g, HWo, p.rcp_HWo, w_n, rem);
        divmod24(rem, p.Wo, p.rcp_Wo, w_oh, w_ow);
    }
    auto offsets = [&](int kt) {
        const int pp0 = pbeg + kt * 16 + 4 * s_g;
        if (is_a) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pp = pp0 + i;
                woff[i] = pp < pend ? (unsigned)(pp * p.O + o0 + 4 * s_c) * ES : OOB_OFFSET;
            }
        } else if (is_b) {
            int n = w_n, oh = w_oh, ow = w_ow;
            w_ow += 16;
            while (w_ow >= p.Wo) {
                w_ow -= p.Wo;
                if (++w_oh == p.Ho) {
                    w_oh = 0;
                    ++w_n;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ih = oh * p.stride + tr, iw = ow * p.stride + ts;
                const bool v = j_ok && (pp0 + i) < pend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                woff[i] = v ? (unsigned)(((n * p.H + ih) * p.W + iw) * p.Cb + ci) * ES : OOB_OFFSET;
                // next pixel (row-major over n, oh, ow), branch-free carry
                ++ow;
                const bool cw = ow == p.Wo;
                ow = cw ? 0 : ow;
                oh += cw ? 1 : 0;
                const bool ch = oh == p.Ho;
                oh = ch ? 0 : oh;
                n += ch ? 1 : 0;
            }
        }
    };
    auto issue_loads = [&](auto SET) {
        f32x4 (&rr)[4] = rr_[decltype(SET)::value];
        u32x2 (&rh)[NP][4] = rh_[decltype(SET)::value];
        if constexpr (D > 1) {
            // ONE code path for both roles (descriptor and plane distance picked by the wave-uniform role; threads without a role
            // load from out-of-range offsets): with the loads inside role branches the compiler's wait counts at the merge point
            // fell back to vmcnt(0) and drained every set each k-step
            const __amdgpu_buffer_rsrc_t rs = is_a ? rs_sm : rs_big;
            if constexpr (AT == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = buffer_load_f32x4(rs, woff[i]);
            } else {
                const unsigned pl = is_a ? p.sm_plane_bytes : p.big_plane_bytes;
#pragma unroll
                for (int t = 0; t < NP; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        rh[t][i] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(woff[i] == OOB_OFFSET ? OOB_OFFSET : woff[i] + t * pl), 0, 0);
            }
            return;
        }
        if constexpr (AT == 0) {
            if (is_a) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = buffer_load_f32x4(rs_sm, woff[i]);
            } else if (is_b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = buffer_load_f32x4(rs_big, woff[i]);
            }
        } else {
            // (an out-of-range offset must stay out of range: the plane distance is added to valid offsets only)
            const unsigned pl = is_a ? p.sm_plane_bytes : p.big_plane_bytes;
            if (is_a) {
#pragma unroll
                for (int t = 0; t < NP; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        rh[t][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_sm, (int)(woff[i] == OOB_OFFSET ? OOB_OFFSET : woff[i] + t * pl), 0, 0);
            } else if (is_b) {
#pragma unroll
                for (int t = 0; t < NP; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        rh[t][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_big, (int)(woff[i] == OOB_OFFSET ? OOB_OFFSET : woff[i] + t * pl), 0, 0);
            }
        }
    };
    auto stage = [&](int buf, auto SET) {
        f32x4 (&rr)[4] = rr_[decltype(SET)::value];
        u32x2 (&rh)[NP][4] = rh_[decltype(SET)::value];
        if (is_a || is_b) {
            f32x4* dst = smem + buf * STAGE + lds_base;
            if constexpr (NS == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[s_g * ld + e * qn] = f32x4{rr[0][e], rr[1][e], rr[2][e], rr[3][e]};
            } else if constexpr (AT != 0) {
                // channel e of pixels 0..3 -> one 8-byte half slot: word = (pixel a | pixel b << 16) of the channel's 16 bits
#pragma unroll
                for (int t = 0; t < NP; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned a0 = rh[t][0][e >> 1], a1 = rh[t][1][e >> 1], a2 = rh[t][2][e >> 1], a3 = rh[t][3][e >> 1];
                        const u32x2 o = (e & 1) ? u32x2{(a0 >> 16) | (a1 & 0xFFFF0000u), (a2 >> 16) | (a3 & 0xFFFF0000u)}
                                                : u32x2{(a0 & 0xFFFFu) | (a1 << 16), (a2 & 0xFFFFu) | (a3 << 16)};
                        reinterpret_cast<u32x2*>(dst + (t * 2 + (s_g >> 1)) * ld + e * qn)[s_g & 1] = o;
                    }
            } else {
                // pixel group g = k 4g..4g+3 of the k-tile: bf16 image slot [g>>1][pos], 8-byte half (g&1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32x2 sp[NSX];
                    split4<NS>(f32x4{rr[0][e], rr[1][e], rr[2][e], rr[3][e]}, sp);
#pragma unroll
                    for (int t = 0; t < NS; ++t)
                        reinterpret_cast<u32x2*>(dst + (t * 2 + (s_g >> 1)) * ld + e * qn)[s_g & 1] = sp[t];
                }
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Address math placement (measured): with the long fp32 MFMAs (NS = 0) computing the offsets right before
    // the loads is faster (fewer live registers across the MFMA block); with the short bf16 MFMAs (NS > 0) they
    // are computed one tile ahead, in the shadow of the previous tile's MFMAs.
    // (a prefetch distance of two k-tiles, which helps the igemm kernel, costs this kernel its occupancy — every thread
    // holds a 4x4 block per set for the register transpose: 64 -> 130 VGPRs, 100 -> 72 TFLOP/s measured — so it stays at one)
    using C0 = std::integral_constant<int, 0>;
    if constexpr (D == 1) {
    if (KT > 0) {
        offsets(0);
        issue_loads(C0{});
        if (NS > 0) offsets(1);
        stage(0, C0{});
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        if (more) {
            if (NS == 0) offsets(kt + 1);
            issue_loads(C0{});
        }
        const f32x4* As = smem + buf * STAGE;
        const f32x4* Bs = As + A_IMG;
        if constexpr (NS == 0) {
            f32x4 af[2][MI], bf[2][NI];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[s2][a] = As[(2 * s2 + lh) * AS + wm * TM + a * 32 + li];
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[s2][b] = Bs[(2 * s2 + lh) * BS + wn * TN + b * 32 + li];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < MI; ++a)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2][a][e], bf[s2][b][e], acc[a][b], 0, 0, 0);
        } else {
            bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[t][a] = __builtin_bit_cast(bf16x8, As[(t * 2 + lh) * AS + wm * TM + a * 32 + li]);
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BS + wn * TN + b * 32 + li]);
            }
            mfma_split<NS, MI, NI>(af, bf, acc);
        }
        if (NS > 0) offsets(kt + 2);  // independent of the MFMAs above: overlaps their execution
        if (more) stage(buf ^ 1, C0{});
        __syncthreads();
    }
    } else {
    // distance D: sets hold k-tiles kt+1 .. kt+D-1 (+ the one being issued); loads and staging are unconditional (k-tiles past the
    // end gather zeros through out-of-range offsets) so that the compiler's counted waits stay partial
#pragma unroll
    for (int d = 0; d < D; ++d) {
        offsets(d);
        if (d == 0) issue_loads(std::integral_constant<int, 0>{});
        if (d == 1) issue_loads(std::integral_constant<int, 1 % D>{});
        if (d == 2) issue_loads(std::integral_constant<int, 2 % D>{});
        if (d == 3) issue_loads(std::integral_constant<int, 3 % D>{});
    }
    stage(0, C0{});
    __syncthreads();
    auto step = [&](int kt, auto UU) {
        constexpr int U = decltype(UU)::value;
        const int buf = kt & 1;
        offsets(kt + D);
        issue_loads(UU);  // the set k-tile kt was staged from
        const f32x4* As = smem + buf * STAGE;
        const f32x4* Bs = As + A_IMG;
        if constexpr (NS == 0) {
            f32x4 af[2][MI], bf[2][NI];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[s2][a] = As[(2 * s2 + lh) * AS + wm * TM + a * 32 + li];
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[s2][b] = Bs[(2 * s2 + lh) * BS + wn * TN + b * 32 + li];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < MI; ++a)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2][a][e], bf[s2][b][e], acc[a][b], 0, 0, 0);
        } else {
            bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[t][a] = __builtin_bit_cast(bf16x8, As[(t * 2 + lh) * AS + wm * TM + a * 32 + li]);
#pragma unroll
                for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BS + wn * TN + b * 32 + li]);
            }
            mfma_split<NS, MI, NI>(af, bf, acc);
        }
        stage(buf ^ 1, std::integral_constant<int, (U + 1) % D>{});
        __syncthreads();
    };
    static_assert(D <= 4, "the k-loop spells the sets out");
    // whole rounds of D steps (no conditional step inside the loop: a skipped step would reach the loop header with a different
    // number of loads pending, and the compiler then drains with vmcnt(0) there), then the remainder
    int kt = 0;
    for (; kt + D <= KT; kt += D) {
        step(kt, std::integral_constant<int, 0>{});
        if constexpr (D > 1) step(kt + 1, std::integral_constant<int, 1 % D>{});
        if constexpr (D > 2) step(kt + 2, std::integral_constant<int, 2 % D>{});
        if constexpr (D > 3) step(kt + 3, std::integral_constant<int, 3 % D>{});
    }
    if (kt < KT) step(kt, std::integral_constant<int, 0>{});
    if (D > 2 && kt + 1 < KT) step(kt + 1, std::integral_constant<int, 1 % D>{});
    if (D > 3 && kt + 2 < KT) step(kt + 2, std::integral_constant<int, 2 % D>{});
    }

    // slab in position space: [split][O (tile-major positions)][Jp = njt*BN]
    const int Jp = njt * BN;
    float* out = p.slab + (long)split * wgrad_slab_stride(p.O, Jp);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = j0 + wn * TN + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(long)row * Jp + col] = acc[a][b][r];
            }
        }
}

// ---- weight gradient, LDS-DMA form (exact fp32 MFMA; the default for fp32 tensors) -------------------------------------------
// Same GEMM (M = Cout tile, N = (tap, ci) tile, K = a range of output pixels) with the operands staged the way they lie in memory:
// the LDS image of a stage is [16 pixels][BM] of dY and [16 pixels][BN] of the gathered X, PIXEL-major.  The f32 MFMA wants, per lane,
// A[i = lane & 31][k = lane >> 5] — one word of pixel row k, channel i: lanes 0-31 read 32 consecutive words of a pixel row
// (ds_read_b32, conflict-free), so no transpose is needed anywhere and the panels can be written by LDS-DMA
// (buffer_load_dwordx4 ... lds: 64 lanes x 16 B = 1 KiB per instruction, out-of-range lanes — padding taps, pixel tails,
// columns past J — deliver zeros).  No staging registers, hence a three-stage ring with two stages in flight across raw barriers
// (counted vmcnt), where the register-transposing kernel above could only afford a prefetch distance of one: its loads were
// exposed every k-step once the workgroups of a launch (all started together, equally long) ran in lockstep.
// The accumulators, and therefore the slabs, are in natural (o, j) order here (`natural` flag of the reduction kernels).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgradParams p) {
    constexpr int KP = 16;  // pixels per stage
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int A_F4 = KP * BM / 4, B_F4 = KP * BN / 4;    // float4 items of the two panels of a stage
    constexpr int A_INSTR = A_F4 / 64, B_INSTR = B_F4 / 64;  // wave-level DMA instructions (1 KiB each)
    constexpr int PER_WAVE = (A_INSTR + B_INSTR) / 4;
    static_assert((A_INSTR + B_INSTR) % 4 == 0 && WM * WN == 4, "DMA instructions are dealt to four waves");
    constexpr int NSTG = 3;
    constexpr int STAGE_F4 = A_F4 + B_F4;
    __shared__ f32x4 smem[NSTG * STAGE_F4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int njt = (p.J + BN - 1) / BN;
    const int ntiles = (p.O / BM) * njt;
    const int work = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = work / ntiles, tile_ = work - split * ntiles;
    const int ot = tile_ / njt, jt = tile_ - ot * njt;
    const int o0 = ot * BM, j0 = jt * BN;
    const int pbeg = split * p.pchunk;
    const int pend = min(p.P, pbeg + p.pchunk);
    const int KT = (pend - pbeg + KP - 1) / KP;

    const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sm), 0, p.sm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.big), 0, p.big_bytes, 0x00020000);

    // this wave's DMA instructions t = wave*PER_WAVE + i (A panel first, then B); per instruction the lane's item:
    //   A: pixel row ra, channel quad qa   -> byte offset ((pixel*O + o0 + 4 qa) * 4), advancing 16 pixels per stage
    //   B: pixel row rb, column quad qb    -> (tap, ci) fixed, the pixel (n, oh, ow) walks 16 pixels per stage
    const int HWo = p.Ho * p.Wo;
    unsigned a_off[PER_WAVE];          // A: offset of the lane's item in stage 0 (OOB handled per stage)
    int a_px[PER_WAVE];                // A: pixel index of the item in the current stage
    int b_n[PER_WAVE], b_oh[PER_WAVE], b_ow[PER_WAVE], b_px[PER_WAVE], b_ci[PER_WAVE], b_tr[PER_WAVE], b_ts[PER_WAVE];
    bool b_ok[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int t = wave * PER_WAVE + i;
        a_off[i] = 0; a_px[i] = 0; b_n[i] = b_oh[i] = b_ow[i] = b_px[i] = b_ci[i] = b_tr[i] = b_ts[i] = 0; b_ok[i] = false;
        if (t < A_INSTR) {
            const int f = t * 64 + lane, row = f / (BM / 4), quad = f - row * (BM / 4);
            a_px[i] = pbeg + row;
            a_off[i] = (unsigned)(o0 + 4 * quad) * 4u;
        } else {
            const int f = (t - A_INSTR) * 64 + lane, row = f / (BN / 4), quad = f - row * (BN / 4);
            const int jj = j0 + 4 * quad;
            b_ok[i] = jj < p.J;
            const int tap = b_ok[i] ? jj / p.Cb : 0;
            b_ci[i] = b_ok[i] ? jj - tap * p.Cb : 0;
            b_tr[i] = tap / p.S - p.pad;
            b_ts[i] = tap % p.S - p.pad;
            b_px[i] = pbeg + row;
            int rem;
            divmod24(min(b_px[i], p.P - 1), HWo, p.rcp_HWo, b_n[i], rem);
            divmod24(rem, p.Wo, p.rcp_Wo, b_oh[i], b_ow[i]);
        }
    }
    auto issue_stage = [&](int slot) {  // DMA of the NEXT not yet issued stage into ring slot `slot`; advances the per-item pixel state
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int t = wave * PER_WAVE + i;
            auto* dst = (__attribute__((address_space(3))) void*)(smem + slot * STAGE_F4 + t * 64);
            if (t < A_INSTR) {
                const unsigned off = a_px[i] < pend ? (unsigned)a_px[i] * (unsigned)p.O * 4u + a_off[i] : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sm, dst, 16, (int)off, 0, 0, 0);
                a_px[i] += KP;
            } else {
                const int ih = b_oh[i] * p.stride + b_tr[i], iw = b_ow[i] * p.stride + b_ts[i];
                const bool v = b_ok[i] && b_px[i] < pend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const unsigned off = v ? (unsigned)(((b_n[i] * p.H + ih) * p.W + iw) * p.Cb + b_ci[i]) * 4u : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, dst, 16, (int)off, 0, 0, 0);
                b_px[i] += KP;
                b_ow[i] += KP;
                while (b_ow[i] >= p.Wo) {  // next pixel row(s) / image
                    b_ow[i] -= p.Wo;
                    if (++b_oh[i] == p.Ho) {
                        b_oh[i] = 0;
                        ++b_n[i];
                    }
                }
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (KT > 0) issue_stage(0);
    if (KT > 1) issue_stage(1);
    const float* lds = reinterpret_cast<const float*>(smem);
    for (int kt = 0; kt < KT; ++kt) {
        // this wave's DMA of stage kt has landed once at most the PER_WAVE instructions of stage kt+1 are outstanding
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's part of stage kt is in LDS; everyone is done reading slot (kt-1) % 3
        asm volatile("" ::: "memory");  // (the barrier builtin is no compiler fence)
        if (kt + 2 < KT) issue_stage((kt + 2) % NSTG);
        const float* As = lds + (kt % NSTG) * STAGE_F4 * 4;
        const float* Bs = As + A_F4 * 4;
#pragma unroll
        for (int kk = 0; kk < KP / 2; ++kk) {
            const int r = 2 * kk + lh;
            float af[MI], bf[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) af[a] = As[r * BM + wm * TM + a * 32 + li];
#pragma unroll
            for (int b = 0; b < NI; ++b) bf[b] = Bs[r * BN + wn * TN + b * 32 + li];
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }

    // slab in natural order: [split][O][Jp = njt*BN]
    const int Jp = njt * BN;
    float* out = p.slab + (long)split * wgrad_slab_stride(p.O, Jp);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = j0 + wn * TN + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(long)row * Jp + col] = acc[a][b][r];
            }
        }
}

// compile-time loop (the transposing LDS reads below take their offsets as instruction immediates)
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
template <int OFF>
__device__ __forceinline__ u32x2 tr_read_b64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// ---- weight gradient on stored bf16 operands: LDS-DMA + transposing LDS reads ----------------------------------------------------
// The register-transposing kernel above spends ~27 VALU instructions per MFMA on bf16 tensors (four pixel addresses per thread and
// k-step for 8-byte loads, 4x4 transposes as bit operations): with one 32-cycle MFMA per accumulator and k-step it is bound by
// instruction issue, not by the matrix pipe or memory (measured: 0.14 of the bf16 peak alone, unchanged by more workgroups per CU
// or a deeper prefetch).  Here both operands stay the way they lie in memory — pixel-major, per 32-channel block an LDS image
// [32 pixels][64 B] written by LDS-DMA (16 pixels x 4 pieces of 16 B per instruction; out-of-range lanes deliver zeros) — and the
// K(pixel)-contiguous MFMA fragments come out of ds_read_b64_tr_b16: the 16 lanes of a group pass the addresses of a [4 pixels][16
// channels] block (lane 4r+c: pixel r, channels 4c..4c+3) and lane t receives channel t of the four pixels
// (tools/probes/tr_read.hip).  Two such reads are the 8 k-values of a 32x32x16 fragment; the 64-byte rows make the four rows of
// the two groups of a 32-lane phase cover 256 distinct bytes: conflict-free.  Slabs in natural (o, j) order.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const WgradParams p) {
    constexpr int KP = 32;  // pixels per stage (two 16-wide MFMA k-steps)
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int A_BLK = BM / 32, B_BLK = BN / 32;      // 32-wide channel / column blocks
    constexpr int BLK_SL = KP * 4;                       // 16-byte slots of one block image [KP pixels][64 B]
    constexpr int INSTR = (A_BLK + B_BLK) * (KP / 16);   // DMA instructions per stage (1 KiB each)
    static_assert(INSTR % 4 == 0 && WM * WN == 4 && MI >= 1 && NI >= 1, "DMA instructions are dealt to four waves");
    constexpr int PER_WAVE = INSTR / 4;
    constexpr int NSTG = 3;
    constexpr int STAGE_SL = (A_BLK + B_BLK) * BLK_SL;
    __shared__ f32x4 smem[NSTG * STAGE_SL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int njt = (p.J + BN - 1) / BN;
    const int ntiles = (p.O / BM) * njt;
    const int work = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = work / ntiles, tile_ = work - split * ntiles;
    const int ot = tile_ / njt, jt = tile_ - ot * njt;
    const int o0 = ot * BM, j0 = jt * BN;
    const int pbeg = split * p.pchunk;
    const int pend = min(p.P, pbeg + p.pchunk);
    const int KT = (pend - pbeg + KP - 1) / KP;

    const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sm), 0, p.sm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.big), 0, p.big_bytes, 0x00020000);

    // Dealing with COMPILE-TIME kinds per slot (see the generic 16-bit convolution loop): slots [0, PA) of a wave are dY
    // instructions a = wave*PA + j, slots [PA, PER_WAVE) X instructions b = wave*PB + j; instruction x covers block x / 2,
    // pixel half x % 2; the lane's item is pixel 16*half + lane/4 of the stage, piece lane % 4 (8 channels)
    constexpr int PA = A_BLK * (KP / 16) / 4, PB = B_BLK * (KP / 16) / 4;
    static_assert(PA * 4 == A_BLK * (KP / 16) && PB * 4 == B_BLK * (KP / 16) && PA + PB == PER_WAVE, "whole slots per wave");
    const int HWo = p.Ho * p.Wo;
    unsigned a_off[PA];
    int a_px[PA], a_lds[PA];
    int b_n[PB], b_oh[PB], b_ow[PB], b_px[PB], b_ci[PB], b_tr[PB], b_ts[PB], b_lds[PB];
    bool b_ok[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int x = wave * PA + j, blk = x >> 1, half = x & 1;
        a_px[j] = pbeg + 16 * half + (lane >> 2);
        a_off[j] = (unsigned)(o0 + 32 * blk + 8 * (lane & 3)) * 2u;
        a_lds[j] = blk * BLK_SL + half * 64;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int x = wave * PB + j, blk = x >> 1, half = x & 1;
        const int px = pbeg + 16 * half + (lane >> 2);
        const int jj = j0 + 32 * blk + 8 * (lane & 3);
        b_px[j] = px;
        b_lds[j] = (A_BLK + blk) * BLK_SL + half * 64;
        b_ok[j] = jj < p.J;
        const int tap = b_ok[j] ? jj / p.Cb : 0;
        b_ci[j] = b_ok[j] ? jj - tap * p.Cb : 0;
        b_tr[j] = tap / p.S - p.pad;
        b_ts[j] = tap % p.S - p.pad;
        int rem;
        divmod24(min(px, p.P - 1), HWo, p.rcp_HWo, b_n[j], rem);
        divmod24(rem, p.Wo, p.rcp_Wo, b_oh[j], b_ow[j]);
    }
    auto issue_stage = [&](int slot) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            auto* dst = (__attribute__((address_space(3))) void*)(smem + slot * STAGE_SL + a_lds[j]);
            const unsigned off = a_px[j] < pend ? (unsigned)a_px[j] * (unsigned)p.O * 2u + a_off[j] : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sm, dst, 16, (int)off, 0, 0, 0);
            a_px[j] += KP;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            auto* dst = (__attribute__((address_space(3))) void*)(smem + slot * STAGE_SL + b_lds[j]);
            const int ih = b_oh[j] * p.stride + b_tr[j], iw = b_ow[j] * p.stride + b_ts[j];
            const bool v = b_ok[j] && b_px[j] < pend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned off = v ? (unsigned)(((b_n[j] * p.H + ih) * p.W + iw) * p.Cb + b_ci[j]) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, dst, 16, (int)off, 0, 0, 0);
            b_px[j] += KP;
            b_ow[j] += KP;
            while (b_ow[j] >= p.Wo) {
                b_ow[j] -= p.Wo;
                if (++b_oh[j] == p.Ho) {
                    b_oh[j] = 0;
                    ++b_n[j];
                }
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposing read: group g = lane >> 4 covers channels 16 (g & 1) .. +15 of the block and k = 8 (g >> 1) .. +7 (two reads of 4)
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) f32x4*)smem;
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned a_base = lds0 + lane_off + (unsigned)(wm * MI) * (BLK_SL * 16);
    const unsigned b_base = lds0 + lane_off + (unsigned)(A_BLK + wn * NI) * (BLK_SL * 16);

    if (KT > 0) issue_stage(0);
    if (KT > 1) issue_stage(1);
    int slot = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < KT) issue_stage(slot >= 1 ? slot - 1 : NSTG - 1);
        const unsigned sbase = (unsigned)slot * (STAGE_SL * 16);
        // all fragment reads of the stage are issued first; the second k-step's land under the first one's MFMAs
        constexpr int KS = KP / 16, RPK = 2 * (MI + NI);  // k-steps per stage, reads per k-step
        u32x2 fa[KS][MI][2], fb[KS][NI][2];
        static_for<KS>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            static_for<MI>([&](auto A_) {
                static_for<2>([&](auto H_) {
                    constexpr int a = decltype(A_)::value, h2 = decltype(H_)::value;
                    fa[kk][a][h2] = tr_read_b64<a * BLK_SL * 16 + (16 * kk + 4 * h2) * 64>(a_base + sbase);
                });
            });
            static_for<NI>([&](auto B_) {
                static_for<2>([&](auto H_) {
                    constexpr int b = decltype(B_)::value, h2 = decltype(H_)::value;
                    fb[kk][b][h2] = tr_read_b64<b * BLK_SL * 16 + (16 * kk + 4 * h2) * 64>(b_base + sbase);
                });
            });
        });
        static_for<KS>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            // the compiler does not track inline-asm LDS reads: counted wait (LDS returns in order), and every result register
            // of this k-step is tied to it
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[kk][0][0]) : "n"((KS - 1 - kk) * RPK) : "memory");
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) asm volatile("" : "+v"(fa[kk][a][h2]));
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) asm volatile("" : "+v"(fb[kk][b][h2]));
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            bf16x8 af[MI], bf[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a)
                af[a] = __builtin_bit_cast(bf16x8, u32x4_{fa[kk][a][0][0], fa[kk][a][0][1], fa[kk][a][1][0], fa[kk][a][1][1]});
#pragma unroll
            for (int b = 0; b < NI; ++b)
                bf[b] = __builtin_bit_cast(bf16x8, u32x4_{fb[kk][b][0][0], fb[kk][b][0][1], fb[kk][b][1][0], fb[kk][b][1][1]});
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        });
        slot = slot + 1 == NSTG ? 0 : slot + 1;
    }

    // slab in natural order: [split][O][Jp = njt*BN]
    const int Jp = njt * BN;
    float* out = p.slab + (long)split * wgrad_slab_stride(p.O, Jp);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) {
            const int col = j0 + wn * TN + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o0 + wm * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(long)row * Jp + col] = acc[a][b][r];
            }
        }
}

// ---- weight gradient of 3x3 / stride-1 convolutions on the bf16 matrix pipe: pixel patches --------------------------------------
// dW[o][(tap, ci)] = sum_p dY[p][o] * X[p + tap][ci].  A workgroup owns 64 output channels x 64 input channels x ONE TAP ROW r (three
// taps: a 64 x 192 accumulator tile like <64,192> above) and walks a range of 4 x 16 pixel patches.  Per patch the dY patch (64
// pixels x 64 channels) and the four X rows it needs (rows y + r - 1, 18 pixels wide: 72 pixels x 64 channels) go to LDS once, in
// memory order — per 32-channel block an image [pixel][64 B], fp32 sources split into their bf16 planes on the way — and the three
// taps of the row are LDS address offsets of the B-fragment reads.  K(pixel)-contiguous fragments come out of ds_read_b64_tr_b16
// (see wgrad_tr_kernel): a k-step of 16 is one patch row.  Against wgrad_tr_kernel / the register-transposing kernel this loads
// 17 KB instead of 32 KB per 64 pixels (bf16), computes no per-tap pixel addresses and transposes nothing in registers.
// Needs H % 4 == 0, W % 16 == 0, O % 64 == 0, Cb % 64 == 0.  Slabs in natural (o, j) order, one per patch range.
// (three planes: 172 registers as written = two waves per SIMD; the attribute asks for three — 52 KB of LDS admit three workgroups)
template <int NS, int AT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS == 3 ? 3 : 1, 8))) void wgrad_patch_kernel(const WgradParams p) {
    static_assert((NS == 1 || NS == 3) && (AT == 0 || AT == 1), "bf16 matrix math on fp32 or stored-bf16 tensors");
    constexpr int ES = AT == 0 ? 4 : 2;
    constexpr int CH = 16 / ES;                  // channels per 16-byte piece
    constexpr int CPP = 64 / CH;                 // pieces per pixel (64 channels)
    constexpr int APX = 64, BPX = 72;            // pixels of the dY patch / of the four X rows
    constexpr int A_SL = 2 * APX * 4, B_SL = 2 * BPX * 4;   // 16-byte slots per plane: [2 blocks][pixels][64 B]
    constexpr int PLANE_SL = A_SL + B_SL;
    constexpr int ITEMS = (APX + BPX) * CPP;
    constexpr int PL = (ITEMS + 255) / 256;
    constexpr int AJ = APX * CPP / 256;          // items j < AJ of every thread are dY pieces, the others X pieces
    static_assert(APX * CPP % 256 == 0, "the operand of an item must not depend on the thread (uniform buffer descriptor)");
    __shared__ f32x4 smem[NS * PLANE_SL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ncb = p.Cb >> 6;
    const int ntiles = (p.O >> 6) * ncb * 3;
    const int work = dbn_xcd_remap(blockIdx.x, gridDim.x);
    const int split = work / ntiles, tile_ = work - split * ntiles;
    const int ot = tile_ / (ncb * 3), rem_ = tile_ - ot * (ncb * 3);
    const int cb = rem_ / 3, r = rem_ - cb * 3;       // input-channel block, tap row
    const int o0 = ot * 64, ci0 = cb * 64;
    // p.pchunk: patches per split here
    const int PWn = p.W >> 4, PPI = (p.H >> 2) * PWn;  // patches per row / per image
    const int qbeg = split * p.pchunk, qend = min(p.N * PPI, qbeg + p.pchunk);

    const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sm), 0, p.sm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.big), 0, p.big_bytes, 0x00020000);

    // this thread's pieces: item = tid + j*256; items [0, APX*CPP): dY pixel item / CPP, channel piece item % CPP; then the X rows
    bool it_on[PL];
    int it_y[PL], it_x[PL], it_lds[PL];   // pixel inside the patch (X: row 0..3, column 0..17), LDS slot (16-byte units; fp32: 8-byte units)
    unsigned it_c[PL];                    // byte offset of the piece inside its pixel
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int item = tid + j * 256;
        it_on[j] = item < ITEMS;
        const bool isa = j < AJ;
        const int q = isa ? item : (it_on[j] ? item - APX * CPP : 0);
        const int px = q / CPP, piece = q - px * CPP;
        const int roww = isa ? 16 : 18;
        it_y[j] = px / roww;
        it_x[j] = px - it_y[j] * roww;
        const int ch = piece * CH;            // channel inside the 64-channel block
        it_c[j] = (unsigned)((isa ? o0 : ci0) + ch) * (unsigned)ES;
        // slot of the piece: [block ch/32][pixel][64 B]; 16-bit source: 16-byte slot; fp32 source: 8-byte half slots
        const int base = (isa ? 0 : A_SL) + (ch >> 5) * (isa ? APX : BPX) * 4 + px * 4;
        it_lds[j] = AT == 0 ? base * 2 + ((ch & 31) >> 2) : base + ((ch & 31) >> 3);
    }
    f32x4 pr[PL];
    auto load_patch = [&](int q) {  // patch q: image n, top-left output pixel (h0, w0)
        int n, rem, ty, tx;
        divmod24(q < qend ? q : qbeg, PPI, 1.0f / (float)PPI, n, rem);
        divmod24(rem, PWn, 1.0f / (float)PWn, ty, tx);
        const int h0 = ty * 4, w0 = tx * 16;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const bool isa = j < AJ;  // (compile-time after unrolling)
            const int h = isa ? h0 + it_y[j] : h0 - 1 + r + it_y[j];
            const int w = isa ? w0 + it_x[j] : w0 - 1 + it_x[j];
            const bool v = it_on[j] && q < qend && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const unsigned off = (unsigned)((n * p.H + h) * p.W + w) * (unsigned)((isa ? p.O : p.Cb) * ES) + it_c[j];
            pr[j] = buffer_load_f32x4(isa ? rs_sm : rs_big, v ? off : OOB_OFFSET);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            if (!it_on[j]) continue;
            if constexpr (AT == 0) {
                u32x2 sp[NS];
                split4<NS>(pr[j], sp);
#pragma unroll
                for (int t = 0; t < NS; ++t) reinterpret_cast<u32x2*>(smem + t * PLANE_SL)[it_lds[j]] = sp[t];
            } else {
                smem[it_lds[j]] = pr[j];
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    // transposing reads (wgrad_tr_kernel): group g = lane >> 4 covers channels 16 (g & 1) .. +15 of a 32-channel block and
    // k = 8 (g >> 1) .. +7 of the 16-pixel row (two reads of 4 pixels)
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) f32x4*)smem;
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned a_base = lds0 + lane_off + (unsigned)wm * (APX * 64);
    unsigned b_base[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int cblk = 3 * wn + b, s_ = cblk >> 1, half = cblk & 1;  // tap of the row, 32-channel half of the input block
        b_base[b] = lds0 + lane_off + (unsigned)(A_SL * 16 + half * (BPX * 64) + s_ * 64);
    }

    load_patch(qbeg);
    for (int q = qbeg; q < qend; ++q) {
        __syncthreads();        // everyone is done with the previous patch's fragments
        store_patch();
        load_patch(q + 1);      // (past the end: out-of-range loads, so the waits stay the same)
        __syncthreads();
        static_for<4>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;   // patch row = k-step of 16 pixels
            u32x2 fa[NS][2], fb[NS][3][2];
            static_for<NS>([&](auto T_) {
                constexpr int t = decltype(T_)::value;
                static_for<2>([&](auto H_) {
                    constexpr int h2 = decltype(H_)::value;
                    fa[t][h2] = tr_read_b64<t * PLANE_SL * 16 + (16 * kk + 4 * h2) * 64>(a_base);
                    fb[t][0][h2] = tr_read_b64<t * PLANE_SL * 16 + (18 * kk + 4 * h2) * 64>(b_base[0]);
                    fb[t][1][h2] = tr_read_b64<t * PLANE_SL * 16 + (18 * kk + 4 * h2) * 64>(b_base[1]);
                    fb[t][2][h2] = tr_read_b64<t * PLANE_SL * 16 + (18 * kk + 4 * h2) * 64>(b_base[2]);
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]) : : "memory");
#pragma unroll
            for (int t = 0; t < NS; ++t)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    asm volatile("" : "+v"(fa[t][h2]));
#pragma unroll
                    for (int b = 0; b < 3; ++b) asm volatile("" : "+v"(fb[t][b][h2]));
                }
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            bf16x8 af[NS][1], bf[NS][3];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                af[t][0] = __builtin_bit_cast(bf16x8, u32x4_{fa[t][0][0], fa[t][0][1], fa[t][1][0], fa[t][1][1]});
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    bf[t][b] = __builtin_bit_cast(bf16x8, u32x4_{fb[t][b][0][0], fb[t][b][0][1], fb[t][b][1][0], fb[t][b][1][1]});
            }
            f32x16 (&acc2)[1][3] = reinterpret_cast<f32x16 (&)[1][3]>(acc);
            mfma_split<NS, 1, 3>(af, bf, acc2);
        });
    }

    // slab in natural order [split][O][J]: this tile's columns are the three taps (3r + s) of input channels ci0 ..
    float* out = p.slab + (long)split * wgrad_slab_stride(p.O, p.J);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int cblk = 3 * wn + b, s_ = cblk >> 1, half = cblk & 1;
        const int col = (3 * r + s_) * p.Cb + ci0 + 32 * half + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = o0 + 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * lh;
            out[(long)row * p.J + col] = acc[b][e];
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splitk, int O, int J, int Jp, int BM, int BN, int Cb, int I,
                                    int R, int S, float* __restrict__ grad, float scale, int natural) {
    // one thread: 4 consecutive slab positions (one b128 load per split), 4 splits in flight; fixed summation order
    const long total = (long)O * Jp, count4 = total >> 2, total4 = wgrad_slab_stride(O, Jp) >> 2;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < count4; q += (long)gridDim.x * blockDim.x) {
        const long idx = q << 2;
        const f32x4* src = reinterpret_cast<const f32x4*>(slab) + q;
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        int z = 0;
        for (; z + 4 <= splitk; z += 4) {
            const f32x4 v0 = src[(long)z * total4], v1 = src[(long)(z + 1) * total4];
            const f32x4 v2 = src[(long)(z + 2) * total4], v3 = src[(long)(z + 3) * total4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += ((double)v0[e] + (double)v1[e]) + ((double)v2[e] + (double)v3[e]);
        }
        for (; z < splitk; ++z) {
            const f32x4 v = src[(long)z * total4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
        }
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
}

// The same reduction for layers whose input-channel count is a multiple of 64 (every layer but the stem): one workgroup
// per (output channel o, block of 64 input channels), so that BOTH sides are coalesced — the slab is read in 64-byte runs
// (a b128 per lane: four positions = channels 4u+e of one tap) and the OIHW gradient leaves as one contiguous run of
// 64*R*S floats staged through LDS (the per-element kernel above scatters 4-byte stores at a stride of R*S floats).
// Layers with few output elements and hundreds of pixel splits (64-channel layers at 160^2: 3 tiles x 340 splits) are
// latency-bound on the chain of split loads: G thread groups share the splits (group g takes splits g, g+G, ...), their
// fp64 partial sums are combined through LDS in group order — fixed summation order, bit-reproducible.
__global__ __launch_bounds__(1024) void wgrad_reduce64_kernel(const float* __restrict__ slab, int splitk, int O, int J, int Jp, int BM,
                                                              int BN, int Cb, int I, int RS, int G, float* __restrict__ grad,
                                                              float scale, int natural) {
    extern __shared__ double dsm[];  // [G][items][4] partial sums (G > 1), then the [64][RS] float staging image
    const int items = RS * 16;
    float* stage = reinterpret_cast<float*>(dsm + (G > 1 ? (size_t)G * items * 4 : 0));
    const int o = blockIdx.x, i0 = blockIdx.y * 64;
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

// OIHW -> [Kpad/4][Cd][4] panels.  mode 0: k = (r*S+s)*Cs + cs -> w[cd][cs][r][s] (cs < I);
// mode 1: data-gradient panels, taps r = r0 + rstep*r', s = s0 + rstep*s' (R', S' of them):
//         k = (r'*S'+s')*Cs + cs -> w[cs][cd][r][s].
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

// Problem of one block row of a pack launch: the whole kernel (f == 1), or parity class blockIdx.y of a stride-f
// transposed conv — its taps r = ph + f*rp and the offset (in padded-K rows) of its panel behind the earlier classes.
struct PackClass {
    int Rp, Sp, r0, s0, rstep, K, Kpad;
    long krow0;
};
__device__ inline PackClass pack_class(int f, int R, int S, int Cs) {
    PackClass q;
    q.krow0 = 0;
    if (f <= 1) {
        q.Rp = R; q.Sp = S; q.r0 = q.s0 = 0; q.rstep = 1;
    } else {
        const int c = blockIdx.y;
        for (int d = 0; d < c; ++d)
            q.krow0 += (taps_of_class(R, d / f, f) * taps_of_class(S, d % f, f) * Cs + 15) / 16 * 16;
        q.r0 = c / f; q.s0 = c % f; q.rstep = f;
        q.Rp = taps_of_class(R, q.r0, f);
        q.Sp = taps_of_class(S, q.s0, f);
    }
    q.K = q.Rp * q.Sp * Cs;
    q.Kpad = (q.K + 15) / 16 * 16;
    return q;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, int O, int I, int R, int S, int mode, int Cs, int Cd, int f,
                                    float* __restrict__ out) {
    const PackClass q = pack_class(f, R, S, Cs);
    const int K = q.K, Rp = q.Rp, Sp = q.Sp, r0 = q.r0, s0 = q.s0, rstep = q.rstep;
    out += q.krow0 * Cd;
    const long total = (long)q.Kpad * Cd;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        const long qd = idx >> 2;
        const int cd = (int)(qd % Cd);
        const int kc = (int)(qd / Cd);
        const int k = 4 * kc + e;
        float v = 0.f;
        if (k < K) {
            int tap, cs;
            if ((Cs & 15) == 0) {  // channel-block-major K order (see igemm_f32_kernel)
                const int blk = k >> 4;
                tap = blk % (Rp * Sp);
                cs = (blk / (Rp * Sp)) * 16 + (k & 15);
            } else {
                tap = k / Cs;
                cs = k - tap * Cs;
            }
            const int rp = tap / Sp, sp = tap - rp * Sp;
            const int r = r0 + rstep * rp, sx = s0 + rstep * sp;
            if (mode == 0) {
                if (cs < I) v = w[(((long)cd * I + cs) * R + r) * S + sx];
            } else {
                v = w[(((long)cs * I + cd) * R + r) * S + sx];
            }
        }
        out[idx] = v;
    }
}

// Split-bf16 weight panels for the NS > 0 kernels: [KT][NS][2][Cd][8 bf16]; element (k, cd, split t)
// at ((kt*NS + t)*2 + k8)*Cd*8 + cd*8 + e with k = 16*kt + 8*k8 + e.  Same (mode, tap subset) semantics as above.
__device__ __forceinline__ unsigned f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }

// f16 != 0 (NS = 1): fp16 panels for the fp16 inference path instead of bf16
__global__ void pack_weights_bf16s_kernel(const float* __restrict__ w, int O, int I, int R, int S, int mode, int Cs, int Cd, int f,
                                          int NS, int f16, unsigned short* __restrict__ out) {
    const PackClass q = pack_class(f, R, S, Cs);
    const int K = q.K, Rp = q.Rp, Sp = q.Sp, r0 = q.r0, s0 = q.s0, rstep = q.rstep;
    out += q.krow0 * Cd * NS;
    const long total = (long)q.Kpad * Cd;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        const long qd = idx >> 3;
        const int cd = (int)(qd % Cd);
        const int k8g = (int)(qd / Cd);  // global k/8
        const int k = 8 * k8g + e;
        float v = 0.f;
        if (k < K) {
            int tap, cs;
            if ((Cs & 15) == 0) {  // channel-block-major K order (see igemm_f32_kernel)
                const int blk = k >> 4;
                tap = blk % (Rp * Sp);
                cs = (blk / (Rp * Sp)) * 16 + (k & 15);
            } else {
                tap = k / Cs;
                cs = k - tap * Cs;
            }
            const int rp = tap / Sp, sp = tap - rp * Sp;
            const int r = r0 + rstep * rp, sx = s0 + rstep * sp;
            if (mode == 0) {
                if (cs < I) v = w[(((long)cd * I + cs) * R + r) * S + sx];
            } else {
                v = w[(((long)cs * I + cd) * R + r) * S + sx];
            }
        }
        const int kt = k8g >> 1, k8 = k8g & 1;
        if (f16) {
            out[(((long)kt * 2 + k8) * Cd + cd) * 8 + e] = (unsigned short)f16_bits(v);
            continue;
        }
        for (int t = 0; t < NS; ++t) {
            const unsigned bits = bf16_bits_rne(v);
            out[((((long)kt * NS + t) * 2 + k8) * Cd + cd) * 8 + e] = (unsigned short)bits;
            v -= bf16_bits_to_f32(bits);
        }
    }
}

// ---- all weight panels of a model in ONE launch (after every optimizer step every panel is stale) ----
// job = one dbn_pack_weights call; blockIdx.y = job, the job's parity classes are walked inside.
struct PackJob {
    const float* w;
    void* out;
    int O, I, R, S, mode, Cs, Cd, f;
};

template <int BF16>
__global__ void pack_many_kernel(const PackJob* __restrict__ jobs, int NS) {
    const PackJob j = jobs[blockIdx.y];
    const float* __restrict__ w = j.w;
    const int ncls = j.f > 1 ? j.f * j.f : 1;
    long krow0 = 0;
    for (int c = 0; c < ncls; ++c) {
        int Rp = j.R, Sp = j.S, r0 = 0, s0 = 0, rstep = 1;
        if (j.f > 1) {
            r0 = c / j.f; s0 = c % j.f; rstep = j.f;
            Rp = taps_of_class(j.R, r0, j.f);
            Sp = taps_of_class(j.S, s0, j.f);
        }
        const int K = Rp * Sp * j.Cs, Kpad = (K + 15) / 16 * 16;
        const long total = (long)Kpad * j.Cd;
        constexpr int G = BF16 ? 8 : 4;  // k-values per 16-byte group of the panel
        for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int e = (int)(idx % G);
            const long qd = idx / G;
            const int cd = (int)(qd % j.Cd);
            const int kg = (int)(qd / j.Cd);
            const int k = G * kg + e;
            float v = 0.f;
            if (k < K) {
                int tap, cs;
                if ((j.Cs & 15) == 0) {
                    const int blk = k >> 4;
                    tap = blk % (Rp * Sp);
                    cs = (blk / (Rp * Sp)) * 16 + (k & 15);
                } else {
                    tap = k / j.Cs;
                    cs = k - tap * j.Cs;
                }
                const int rp = tap / Sp, sp = tap - rp * Sp;
                const int r = r0 + rstep * rp, sx = s0 + rstep * sp;
                if (j.mode == 0) {
                    if (cs < j.I) v = w[(((long)cd * j.I + cs) * j.R + r) * j.S + sx];
                } else {
                    v = w[(((long)cs * j.I + cd) * j.R + r) * j.S + sx];
                }
            }
            if constexpr (BF16 == 2) {  // fp16 panels (one plane)
                unsigned short* out = reinterpret_cast<unsigned short*>(j.out) + krow0 * j.Cd;
                const int kt = kg >> 1, k8 = kg & 1;
                out[(((long)kt * 2 + k8) * j.Cd + cd) * 8 + e] = (unsigned short)f16_bits(v);
            } else if constexpr (BF16 == 1) {
                unsigned short* out = reinterpret_cast<unsigned short*>(j.out) + krow0 * j.Cd * NS;
                const int kt = kg >> 1, k8 = kg & 1;
                for (int t = 0; t < NS; ++t) {
                    const unsigned bits = bf16_bits_rne(v);
                    out[((((long)kt * NS + t) * 2 + k8) * j.Cd + cd) * 8 + e] = (unsigned short)bits;
                    v -= bf16_bits_to_f32(bits);
                }
            } else {
                reinterpret_cast<float*>(j.out)[krow0 * j.Cd + idx] = v;
            }
        }
        krow0 += Kpad;
    }
}

// fp32 -> three bf16 planes with a0 + a1 + a2 == a exactly (round-to-nearest-even at each step; 24 mantissa bits = 3 x 8):
// planes[t][i], t = 0..2, plane distance `plane_elems` elements.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ planes, long n4,
                                                     long plane_elems) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        u32x2 sp[3];
        split4<3>(v, sp);
#pragma unroll
        for (int t = 0; t < 3; ++t) reinterpret_cast<u32x2*>(planes + t * plane_elems)[i] = sp[t];
    }
}

}  // namespace

extern "C" {

// Tile configuration dbn_igemm_f32 picks for an M x Cd output (tile_hint 0):
// 1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64 — the largest tile that still yields
// >= ~2 workgroups per CU on 256 CUs.
int dbn_igemm_tile_config(int M, int Cd) {
    // Workgroups are handed to the 256 CUs as they free up, so a launch lasts about
    // ceil(blocks/256) tiles per CU; pick the tile that minimises tiles-per-CU x tile area / efficiency
    // (efficiency = measured steady-state MFMA utilisation of each variant).
    // Round 2 re-measured the variants alone on the backbone's four stage shapes (tools/tile_probe.py): with the two-tile prefetch
    // the 64x64 tile is the fastest at 80x80x128 and 40x40x256 (115 / 103 TFLOP/s against 109 / 97 for the choices below).  In
    // the two-stream step that does not carry over: efficiencies {0.89, 0.85, 0.845, 0.83} everywhere gave +0.5 % (within noise)
    // with the dominant kernel's in-step rate down from 0.60 to 0.56, on the layer3/4-sized grids only -0.6 % — small tiles lose
    // more to the co-resident weight-gradient workgroups.  The table stays (DBN_TILE_EFF_R2=1 selects the re-measured one).
    const int bm[4] = {128, 256, 128, 64}, bn[4] = {128, 64, 64, 64};
    static const bool r2_eff = getenv("DBN_TILE_EFF_R2") != nullptr;
    const double eff_r1[4] = {0.89, 0.83, 0.80, 0.72}, eff_r2[4] = {0.89, 0.85, 0.845, 0.83};
    const double* eff = r2_eff ? eff_r2 : eff_r1;
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

static int g_patch_enabled = 1;
static int g_patch_bn64 = getenv("DBN_PATCH_BN64") ? atoi(getenv("DBN_PATCH_BN64")) : 1;
int dbn_set_patch_conv(int on) {  // test / A-B hook: 0 routes the 3x3 stride-1 convolutions through the generic gather loop again
    const int old = g_patch_enabled;
    g_patch_enabled = on != 0;
    return old;
}
// pixel-patch form: 3x3 / stride 1 / pad 1 on the bf16 matrix pipe, whole 8 x 16 patches, 128-row tiles (the BatchNorm
// partial rows of a launch are the same N*H*W/128 either way); kmode: kernel MODE
static bool patch_eligible(int kmode, int ns, int at, int cfg, int R, int S, int stride, int pad, int Hs, int Ws, int Hd, int Wd, int Cs,
                           int ksplit) {
    return g_patch_enabled && (kmode == 0 || kmode == 1) && ns > 0 && at != 3 && (cfg == 1 || cfg == 3) && R == 3 && S == 3 && stride == 1 &&
           pad == 1 && Hs == Hd && Ws == Wd && Hd % 8 == 0 && Wd % 16 == 0 && Cs % 32 == 0 && ksplit <= 1;
}
static int patch_cfg(int cfg) { return (cfg == 1 && g_patch_bn64) ? 3 : cfg; }

// Tile configuration of a dbn_igemm / dbn_conv_bn call (`mode`, `stride` as the caller passes them).  The convolutions that can take
// the pixel-patch kernel get a 128-row tile whatever the generic heuristic says.
static int resolve_cfg(int M_total, int Cd, int tile_hint, int at = 0, int ns = 0, int mode = 0, int R = 0, int S = 0, int stride = 1,
                       int pad = 0, int Hs = 0, int Ws = 0, int Hd = 0, int Wd = 0, int Cs = 0, int ksplit = 1) {
    int cfg = tile_hint > 0 ? tile_hint : dbn_igemm_tile_config(M_total, Cd);
    if (cfg == 1 && Cd % 128 != 0) cfg = 3;
    if (tile_hint == 0 && !(mode == 1 && stride > 1) && patch_eligible(mode, ns, at, 3, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, ksplit)) cfg = 3;
    return cfg;
}

// What one (unchunked) dbn_igemm_t call launches: tile configuration (1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64) + 16 if the
// pixel-patch kernel is used — i.e. the template arguments <BM,BN,WM,WN,MODE,NS,AT,PATCH> of its rocprofv3 symbol.
// kmode: 0 forward, 1 stride-1 data gradient, 2 parity classes, 3 pyramid.
int dbn_igemm_kernel_config(int at, int ns, int kmode, int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride,
                            int pad, int tile_hint, int ksplit) {
    const int cfg = resolve_cfg(N * Hd * Wd, Cd, tile_hint, at, ns, kmode >= 2 ? 1 : kmode, R, S, kmode == 2 && stride == 1 ? 2 : stride, pad, Hs,
                                Ws, Hd, Wd, Cs, ksplit);
    if (patch_eligible(kmode, ns, at, cfg, R, S, stride, pad, Hs, Ws, Hd, Wd, Cs, ksplit)) return patch_cfg(cfg) + 16;
    return cfg;
}

static int igemm_dispatch(IgemmParams& p, int kmode, int ns, int tile_hint, hipStream_t st, int at = 0) {
    // tile choice from the total row count (for parity classes: all classes together)
    int cfg = tile_hint > 0 ? tile_hint : dbn_igemm_tile_config(p.N * p.Hdf * p.Wdf, p.Cd);
    if (cfg == 1 && p.Cd % 128 != 0) cfg = 3;
    p.patch = patch_eligible(kmode, ns, at, cfg, p.R, p.S, p.stride, p.pad, p.Hs, p.Ws, p.Hdf, p.Wdf, p.Cs, p.ksplit);
    // 128 x 64 tiles also where the generic loop takes 128 x 128: K is short (two to eight channel blocks), so twice the workgroups
    // hide the prologue / epilogue better than the wider tile saves weight traffic (measured: 120.8 GFLOP launch 509 -> ~270 us;
    // step +1-3 %); the BatchNorm partial rows depend on BM only.  DBN_PATCH_BN64=0 keeps the generic choice.
    if (p.patch) cfg = patch_cfg(cfg);
    switch (cfg) {
        case 1: return launch_igemm<128, 128, 2, 2>(p, kmode, ns, st, at);
        case 2: return launch_igemm<256, 64, 4, 1>(p, kmode, ns, st, at);
        case 3: return launch_igemm<128, 64, 2, 2>(p, kmode, ns, st, at);
        default: return launch_igemm<64, 64, 2, 2>(p, kmode, ns, st, at);
    }
}

// panel size in floats of one problem: fp32 panels Kpad*Cd; split-bf16 panels Kpad*Cd*ns/2
static long panel_floats(int K, int Cd, int ns) {
    const long kp = ((K + 15) / 16) * 16;
    return ns == 0 ? kp * Cd : kp * Cd * ns / 2;
}

// ---- image chunking ---------------------------------------------------------------------------------------------
// The kernels index pixels with 24-bit reciprocal divisions and address tensors through raw buffer descriptors with 32-bit
// byte offsets.  Images are independent in every convolution, so a call whose tensors exceed those ranges runs as several
// launches over consecutive image ranges (same tile configuration; BatchNorm partial rows simply continue).
static long g_pixel_limit = 1L << 24;        // rows per launch (divmod24)
static long g_byte_limit = 0xF0000000L;      // bytes addressable through one buffer descriptor
static long g_elem_limit = 1L << 32;         // 32-bit element offsets of the epilogue

static int chunk_images(int N, long px_rows, long src_bytes_per_image, long dst_elems_per_image, long src2_bytes_per_image = 0) {
    long n = N;
    auto fit = [&](long per_image, long limit) {
        if (per_image > 0 && per_image * n >= limit) n = (limit - 1) / per_image;
    };
    fit(px_rows, g_pixel_limit);
    fit(src_bytes_per_image, g_byte_limit);
    fit(src2_bytes_per_image, g_byte_limit);
    fit(dst_elems_per_image, g_elem_limit);
    return (int)n;  // 0: a single image does not fit
}

// bytes per element of the destination of a conv (at = 3: pre-split bf16 planes in, fp32 out) and planes of its source
static inline int dst_esize(int at) { return (at == 0 || at == 3) ? 4 : 2; }
static inline int src_planes(int at) { return at == 3 ? 3 : 1; }

static int igemm_run_one(const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                         int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int cfg, int ns,
                         hipStream_t st, float* stats, int stat_rows, int stat_row0, int ksplit, float* slab, int* rows_out, int at,
                         long plane_bytes) {
    IgemmParams p;
    p.src = src; p.wpk = wpk; p.bias = bias; p.dst = dst;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.Cd = Cd; p.Hdf = Hd; p.Wdf = Wd;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.accumulate = accumulate;
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
        DBN_REQUIRE(slab && !stats && Cs % 16 == 0 && ksplit <= KT && ksplit <= 64);
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
    if (covered < p.ncls && !accumulate) {  // some output pixels receive no tap: they are zero
        if (hipMemsetAsync(dst, 0, (size_t)N * Hd * Wd * Cd * dst_esize(at), st) != hipSuccess) return dbn_status();
        p.accumulate = 1;
    }
    *rows_out = 0;
    if (covered == 0) return DBN_OK;
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
                     void* stream, float* stats = nullptr, int ksplit = 1, float* slab = nullptr, int stat_rows_total = 0, int at = 0) {
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
    DBN_REQUIRE(3 * plane_bytes < g_byte_limit);
    const int nmax = chunk_images(N, (long)Hd * Wd, (long)Hs * Ws * Cs * es, (long)Hd * Wd * Cd);
    DBN_REQUIRE(nmax >= 1);               // one image must fit the kernel's index ranges
    DBN_REQUIRE(nmax >= N || ksplit <= 1);  // split-K is for small outputs only
    const int cfg = resolve_cfg((int)std::min<long>((long)N * Hd * Wd, 0x7FFFFFFF), Cd, tile_hint, at, ns, mode, R, S, stride, pad, Hs, Ws,
                                Hd, Wd, Cs, ksplit);
    int row0 = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int n = std::min(nmax, N - n0);
        int rows = 0;
        const int rc = igemm_run_one(reinterpret_cast<const char*>(src) + (long)n0 * Hs * Ws * Cs * es, wpk, bias,
                                     reinterpret_cast<char*>(dst) + (long)n0 * Hd * Wd * Cd * des, n, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride,
                                     pad, mode, accumulate, cfg, ns, st, stats, stat_rows_total, row0, ksplit, slab, &rows, at, plane_bytes);
        if (rc) return rc;
        row0 += rows;
    }
    return DBN_OK;
}

// planes: [3][n] bf16 (n % 4 == 0) — the pre-split form of an fp32 tensor that the at = 3 entry points consume
int dbn_split3(const float* src, void* planes, long n, void* stream) {
    DBN_REQUIRE(src && planes && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(split3_kernel, dim3(dbn_grid(n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned short*>(planes), n / 4, n);
    return dbn_status();
}

// General form.  at: activation storage type of src / dst (DBN_AT_*; 16-bit storage needs ns = 1 and Cs % 16 == 0, panels from
// dbn_pack_weights_t with kind 1 (bf16) / 2 (fp16)).  ns: matrix math (0, 1, 3).  ksplit > 1: split-K with `slab` scratch.
int dbn_igemm_t(int at, int ns, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ksplit, float* slab,
                void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream, nullptr,
                     ksplit < 1 ? 1 : ksplit, slab, 0, at);
}

int dbn_igemm_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                  int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, 0, stream);
}

// Rows of BatchNorm partials a conv with this output shape produces (see dbn_conv_bn_f32); follows igemm_run's chunking
static int bn_tile_rows(int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int mode, int stride, int tile_hint, int at, int ns, int R,
                        int S, int pad) {
    const int nmax = chunk_images(N, (long)Hd * Wd, (long)Hs * Ws * Cs * dbn_esize(at), (long)Hd * Wd * Cd);
    if (nmax < 1) return 0;
    const int cfg = resolve_cfg((int)std::min<long>((long)N * Hd * Wd, 0x7FFFFFFF), Cd, tile_hint, at, ns, mode, R, S, stride, pad, Hs, Ws,
                                Hd, Wd, Cs, 1);
    int rows = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) rows += bn_tile_rows_one(std::min(nmax, N - n0), Hd, Wd, mode, stride, cfg);
    return rows;
}

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
    const int rows = bn_tile_rows(N, Hs, Ws, Cs, Hd, Wd, Cd, mode, stride, tile_hint, at, ns, R, S, pad);
    DBN_REQUIRE(rows > 0);
    // (16-bit storage: the statistics are those of the fp32 accumulators, i.e. of the values BEFORE they are rounded for storage)
    const int rc = igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream, ws,
                             1, nullptr, rows, at);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(Cd), dim3(rows >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, ws, rows, Cd, gamma,
                       beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
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
int dbn_igemm_splitk_plan(int M, int Cd, int K, int Cs) {
    if (Cs % 16 != 0) return 1;
    const int cfg0 = dbn_igemm_tile_config(M, Cd);
    const int cfg = (cfg0 == 1 && Cd % 128 != 0) ? 3 : cfg0;
    static const int bm_of[5] = {0, 128, 256, 128, 64}, bn_of[5] = {0, 128, 64, 64, 64};
    const long tiles = (long)dbn_ceil_div(M, bm_of[cfg]) * (Cd / bn_of[cfg]);
    const int KT = (K + 15) / 16;
    static int max_tiles = -1;
    if (max_tiles < 0) {
        const char* e = getenv("DBN_SPLITK_MAX_TILES");
        max_tiles = e ? atoi(e) : 256;
    }
    if (tiles > max_tiles) return 1;
    long sk = 1024 / tiles;      // about four workgroups per CU
    if (sk > KT / 32) sk = KT / 32;  // at least 32 k-tiles per split
    if (sk > 64) sk = 64;
    return sk < 2 ? 1 : (int)sk;
}

// dbn_igemm_f32 with the reduction split `ksplit` ways (mode 0, or mode 1 with stride 1; Cs % 16 == 0).
// slab: ksplit * N*Hd*Wd*Cd floats of scratch.  Bit-reproducible (fixed summation order).
int dbn_igemm_splitk_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                         int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                         int ksplit, float* slab, void* stream) {
    return igemm_run(src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, accumulate, tile_hint, ns, stream,
                     nullptr, ksplit, slab);
}

int dbn_igemm_packed_floats(int K, int Cd) { return ((K + 15) / 16) * 16 * Cd; }

// ---- pyramid conv (MODE 3): conv3x3 over [s0 | up2(s1) | up4(s2) | up8(s3)] without the concatenation ----
static int pyramid_chunk(int N, int H, int W, int Cs, int Cd, int at = 0) {
    return chunk_images(N, (long)H * W, (long)H * W * Cs * dbn_esize(at), (long)H * W * Cd);
}
static int pyramid_rows_one(int n, int H, int W) { return 64 * dbn_ceil_div((long)n * (H >> 3) * (W >> 3), 128); }

long dbn_pyramid_conv_ws_floats(int N, int H, int W, int Cd) { return (3L * Cd + 1) * N * pyramid_rows_one(1, H, W); }

int dbn_pyramid_conv_t(int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0, const float* w1,
                       const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs, int Cd,
                       int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                       float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
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
        IgemmParams p;
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
        p.R = 3; p.S = 3; p.stride = 8; p.pad = 1; p.accumulate = 0; p.ncls = 64;
        p.stats = bn ? ws : nullptr;
        p.stat_rows = rows_total; p.stat_row0 = row0; p.launch_rows = 0;
        p.ksplit = 1; p.kt_per = 0; p.patch = 0;
        p.src_bytes = p.seg_bytes[0];
        const int rc = launch_igemm<128, 128, 2, 2>(p, 3, ns, st, at);
        if (rc) return rc;
        row0 += p.launch_rows;
    }
    if (!bn) return DBN_OK;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(Cd), dim3(rows_total >= 2048 ? 1024 : 256), 0, st, ws, rows_total, Cd, gamma,
                       beta, eps, momentum, run_mean, run_var, scale, shift, save_mean, save_rstd);
    return dbn_status();
}

int dbn_pyramid_conv_f32(const float* s0, const float* s1, const float* s2, const float* s3, const float* w0, const float* w1,
                         const float* w2, const float* w3, const float* bias, float* dst, int N, int H, int W, int Cs, int Cd,
                         int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                         float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream) {
    return dbn_pyramid_conv_t(0, s0, s1, s2, s3, w0, w1, w2, w3, bias, dst, N, H, W, Cs, Cd, tile_hint, ns, gamma, beta, eps, momentum,
                              run_mean, run_var, scale, shift, save_mean, save_rstd, ws, stream);
}

// Floats of the weight panels for (mode, stride): mode 0 and mode 1/stride 1 -> one panel;
// mode 1/stride f -> f*f parity-class panels back to back.
static long panel_floats_all(int O, int I, int R, int S, int mode, int stride, int ns, int cs = 0) {
    if (mode == 0) return panel_floats(R * S * (cs > 0 ? cs : (I + 3) / 4 * 4), O, ns);
    if (stride == 1) return panel_floats(R * S * O, I, ns);
    long tot = 0;
    for (int c = 0; c < stride * stride; ++c)
        tot += panel_floats(taps_of_class(R, c / stride, stride) * taps_of_class(S, c % stride, stride) * O, I, ns);
    return tot;
}

long dbn_igemm_panel_floats(int O, int I, int R, int S, int mode, int stride) { return panel_floats_all(O, I, R, S, mode, stride, 0); }
long dbn_igemm_bf16s_panel_floats(int O, int I, int R, int S, int mode, int stride, int ns) {
    return panel_floats_all(O, I, R, S, mode, stride, ns);
}
// kind: 0 fp32, 1 bf16, 3 bf16x3, 2 fp16 (sized like kind 1).  cs: channels of the source tensor for mode 0 (0: I rounded up
// to 4; the 16-bit stem input is stored with 16 channels).
long dbn_igemm_panel_floats_t(int kind, int O, int I, int R, int S, int mode, int stride, int cs) {
    return panel_floats_all(O, I, R, S, mode, stride, kind == 2 ? 1 : kind, cs);
}

static int pack_run(const float* w_oihw, int O, int I, int R, int S, int mode, int stride, int ns, float* out, void* stream, int cs = 0,
                    int f16 = 0) {
    DBN_REQUIRE(w_oihw && out && O > 0 && I > 0 && R > 0 && S > 0 && (mode == 0 || mode == 1));
    DBN_REQUIRE(stride == 1 || stride == 2 || stride == 4 || stride == 8 || (mode == 0 && stride >= 1));
    DBN_REQUIRE(cs == 0 || (mode == 0 && cs >= I && cs % 4 == 0));
    const int Cs = (mode == 0) ? (cs > 0 ? cs : ((I + 3) / 4) * 4) : O;
    const int Cd = (mode == 0) ? O : I;
    DBN_REQUIRE(Cs % 4 == 0 && Cd % 64 == 0);
    hipStream_t st = (hipStream_t)stream;
    // one launch; a strided transposed conv packs its stride^2 parity classes as block rows (class order and
    // panel offsets as in igemm_run)
    const int f = (mode == 1 && stride > 1) ? stride : 1;
    const long total = (long)(((dbn_ceil_div(R, f) * dbn_ceil_div(S, f) * Cs + 15) / 16) * 16) * Cd;  // largest class
    const dim3 grid(dbn_grid(total, 256, f > 2 ? 64 : 1024), f * f);
    if (ns == 0)
        hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, st, w_oihw, O, I, R, S, mode, Cs, Cd, f, out);
    else
        hipLaunchKernelGGL(pack_weights_bf16s_kernel, grid, dim3(256), 0, st, w_oihw, O, I, R, S, mode, Cs, Cd, f, ns, f16,
                           reinterpret_cast<unsigned short*>(out));
    return dbn_status();
}

int dbn_pack_weights_t(int kind, const float* w_oihw, int O, int I, int R, int S, int mode, int stride, int cs, float* out, void* stream) {
    DBN_REQUIRE(kind == 0 || kind == 1 || kind == 2 || kind == 3);
    return pack_run(w_oihw, O, I, R, S, mode, stride, kind == 2 ? 1 : kind, out, stream, cs, kind == 2);
}

// Every weight panel of a model in one launch.  jobs: DEVICE array of n records {const float* w; void* out; int O, I, R, S,
// mode, Cs, Cd, f;} (two pointers + eight ints, 48 bytes; Cs/Cd/f as dbn_pack_weights derives them: Cs = mode 0 ? I rounded
// up to 4 : O, Cd = mode 0 ? O : I, f = (mode 1 and stride > 1) ? stride : 1).  ns = 0: fp32 panels, 1 / 3: split-bf16.
int dbn_pack_weights_batched(const void* jobs, int n, int ns, void* stream) {  // ns: panel kind (0 fp32, 1 bf16, 3 bf16x3, 2 fp16)
    DBN_REQUIRE(jobs && n > 0 && (ns == 0 || ns == 1 || ns == 2 || ns == 3));
    const dim3 grid(48, n);
    if (ns == 0)
        hipLaunchKernelGGL(pack_many_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackJob*>(jobs), 0);
    else if (ns == 2)
        hipLaunchKernelGGL(pack_many_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackJob*>(jobs), 1);
    else
        hipLaunchKernelGGL(pack_many_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackJob*>(jobs), ns);
    return dbn_status();
}

int dbn_pack_weights(const float* w_oihw, int O, int I, int R, int S, int mode, int stride, float* out, void* stream) {
    return pack_run(w_oihw, O, I, R, S, mode, stride, 0, out, stream);
}
int dbn_pack_weights_bf16s(const float* w_oihw, int O, int I, int R, int S, int mode, int stride, int ns, float* out, void* stream) {
    DBN_REQUIRE(ns == 1 || ns == 3);
    return pack_run(w_oihw, O, I, R, S, mode, stride, ns, out, stream);
}

static void wgrad_tiles(int O, int J, int& bm, int& bn) {
    static int force192 = -1;
    if (force192 < 0) {
        const char* e = getenv("DBN_WGRAD_192");
        force192 = e ? atoi(e) : 1;  // 0: never, 1: 64-output-channel layers, 2: every layer with J % 192 == 0
    }
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
    static double prefer = -100.0;
    if (prefer < -99.0) {
        const char* e = getenv("DBN_WGRAD_PREFER");
        prefer = e ? atof(e) : 0.0;  // with the two-stream step fewer, longer splits win (32.4 -> 32.2 ms); single stream: 0.02
    }
    static int kmax = -1;
    if (kmax < 0) kmax = getenv("DBN_WGRAD_KMAX") ? atoi(getenv("DBN_WGRAD_KMAX")) : 4;
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
    if (px * n >= g_pixel_limit - 64) n = (g_pixel_limit - 65) / px;
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
static int g_wgrad_variant = -1;
int dbn_set_wgrad_variant(int v) {
    DBN_REQUIRE(v == 0 || v == 1 || v == 2);
    g_wgrad_variant = v;
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
    static const int env = getenv("DBN_WGRAD_PATCH") ? atoi(getenv("DBN_WGRAD_PATCH")) : 1;
    return env && g_wgrad_variant != 2 && ((ns == 1 && (at == 0 || at == 1)) || (ns == 3 && at == 0)) && R == 3 && S == 3 && stride == 1 &&
           pad == 1 && Ho == H && Wo == W && H % 4 == 0 && W % 16 == 0 && O % 64 == 0 && Cb % 64 == 0;
}
static bool wgrad_uses_tr(int at, int ns, int Cb) {
    static const int tr_env = getenv("DBN_WGRAD_TR") ? atoi(getenv("DBN_WGRAD_TR")) : 1;
    return at == 1 && ns == 1 && Cb % 32 == 0 && tr_env && g_wgrad_variant != 2;
}
// (wgrad_uses_patch is defined above)
// tile variant as dbn_wgrad_tile_config, + 16 when the launch is wgrad_tr_kernel<BM,BN,2,2> (bf16 tensors) instead of
// wgrad_f32_kernel<BM,BN,2,2,ns,at> — the rocprofv3 symbol of the matrix kernel of a dbn_wgrad_t call
int dbn_wgrad_kernel_config(int at, int ns, int O, int J, int Cb) {
    return dbn_wgrad_tile_config(O, J) + (wgrad_uses_tr(at, ns, Cb) ? 16 : 0);
}
// ... with the layer geometry: + 32 when the launch is wgrad_patch_kernel<ns, at> (3x3 / stride 1 on the bf16 matrix pipe)
int dbn_wgrad_kernel_config_hw(int at, int ns, int O, int Cb, int R, int S, int stride, int pad, int Ho, int Wo, int H, int W) {
    if (wgrad_uses_patch(at, ns, O, Cb, R, S, stride, pad, Ho, Wo, H, W)) return dbn_wgrad_tile_config(O, R * S * Cb) + 32;
    return dbn_wgrad_kernel_config(at, ns, O, R * S * Cb, Cb);
}

static int wgrad_run(const void* sm_, const void* big_, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                     int Cb, int I, int R, int S, int stride, int pad, float scale, int ns, void* stream, int at = 0, int phases = 3) {
    DBN_REQUIRE(sm_ && big_ && slab && grad_oihw && (ns == 0 || ns == 1 || ns == 3) && phases >= 1 && phases <= 3);
    DBN_REQUIRE(at == 0 || (at == 1 && ns == 1) || (at == 3 && ns == 3));
    DBN_REQUIRE(O % 64 == 0 && Cb % 4 == 0 && I <= Cb && I > 0);
    const long sm_plane = at == 3 ? (long)N * Ho * Wo * O * 2 : 0, big_plane = at == 3 ? (long)N * H * W * Cb * 2 : 0;
    DBN_REQUIRE(3 * sm_plane < g_byte_limit && 3 * big_plane < g_byte_limit);
    const char* sm = reinterpret_cast<const char*>(sm_);
    const char* big = reinterpret_cast<const char*>(big_);
    const int es = dbn_esize(at);
    const int nmax = wgrad_chunk(N, Ho, Wo, O, H, W, Cb, at);
    DBN_REQUIRE(nmax >= 1);
    hipStream_t st = (hipStream_t)stream;
    if (g_wgrad_variant < 0) {
        const char* e = getenv("DBN_WGRAD_DMA");
        g_wgrad_variant = e ? atoi(e) : 0;
    }
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
        p.sm_plane_bytes = (unsigned)sm_plane;
        p.big_plane_bytes = (unsigned)big_plane;
        const int splitk = wgrad_splitk_one(n, Ho, Wo, O, Cb, R, S);
        p.pchunk = (int)((((long)p.P + splitk - 1) / splitk + 15) / 16 * 16);
        dim3 grid((O / bm) * njt * splitk);
        if (!(phases & 1)) {
            splits_total += splitk;
            continue;
        }
#define DBN_WGRAD_LAUNCH(NS_, AT_)                                                                            \
    do {                                                                                                    \
        if (bn == 192)                                                                                      \
            hipLaunchKernelGGL((wgrad_f32_kernel<64, 192, 2, 2, NS_, AT_>), grid, dim3(256), 0, st, p);     \
        else if (bm == 128 && bn == 128)                                                                    \
            hipLaunchKernelGGL((wgrad_f32_kernel<128, 128, 2, 2, NS_, AT_>), grid, dim3(256), 0, st, p);    \
        else if (bn == 128)                                                                                 \
            hipLaunchKernelGGL((wgrad_f32_kernel<64, 128, 2, 2, NS_, AT_>), grid, dim3(256), 0, st, p);     \
        else                                                                                                \
            hipLaunchKernelGGL((wgrad_f32_kernel<64, 64, 2, 2, NS_, AT_>), grid, dim3(256), 0, st, p);      \
    } while (0)
        if (pk) {
            const int patches = n * (H / 4) * (W / 16);
            p.pchunk = (patches + splitk - 1) / splitk;
            const dim3 pgrid((O / 64) * (Cb / 64) * 3 * splitk);  // (a split past the last patch writes a zero slab)
            if (at == 1) hipLaunchKernelGGL((wgrad_patch_kernel<1, 1>), pgrid, dim3(256), 0, st, p);
            else if (ns == 1) hipLaunchKernelGGL((wgrad_patch_kernel<1, 0>), pgrid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((wgrad_patch_kernel<3, 0>), pgrid, dim3(256), 0, st, p);
        } else if (trk) {
            if (bn == 192)
                hipLaunchKernelGGL((wgrad_tr_kernel<64, 192, 2, 2>), grid, dim3(256), 0, st, p);
            else if (bm == 128 && bn == 128)
                hipLaunchKernelGGL((wgrad_tr_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
            else if (bn == 128)
                hipLaunchKernelGGL((wgrad_tr_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((wgrad_tr_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
        } else if (at == 1) DBN_WGRAD_LAUNCH(1, 1);
        else if (at == 3) DBN_WGRAD_LAUNCH(3, 3);
        else if (ns == 0 && dma) {
            if (bn == 192)
                hipLaunchKernelGGL((wgrad_dma_kernel<64, 192, 2, 2>), grid, dim3(256), 0, st, p);
            else if (bm == 128 && bn == 128)
                hipLaunchKernelGGL((wgrad_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
            else if (bn == 128)
                hipLaunchKernelGGL((wgrad_dma_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((wgrad_dma_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
        } else if (ns == 0) DBN_WGRAD_LAUNCH(0, 0);
        else if (ns == 1) DBN_WGRAD_LAUNCH(1, 0);
        else DBN_WGRAD_LAUNCH(3, 0);
#undef DBN_WGRAD_LAUNCH
        const int rc = dbn_status();
        if (rc) return rc;
        splits_total += splitk;
    }
    if (!(phases & 2)) return DBN_OK;
    if (Cb % 64 == 0 && R * S * 64 * 4 <= 32 * 1024) {
        const int items = R * S * 16;
        int G = std::min(1024 / items, splits_total / 4);  // thread groups sharing the splits (>= 4 splits each)
        if (G < 1) G = 1;
        if (G > 32) G = 32;
        if (const char* e = getenv("DBN_REDUCE_G")) G = std::max(1, std::min(atoi(e), 1024 / items > 0 ? 1024 / items : 1));  // experiments
        const int threads = std::min(1024, (items * G + 63) / 64 * 64);
        const size_t smem = (G > 1 ? (size_t)G * items * 4 * sizeof(double) : 0) + (size_t)R * S * 64 * sizeof(float);
        hipLaunchKernelGGL(wgrad_reduce64_kernel, dim3(O, Cb / 64), dim3(threads), smem, st, slab, splits_total, O, J, Jp, bm, bn, Cb, I,
                           R * S, G, grad_oihw, scale, natural ? 1 : 0);
    } else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dbn_grid((long)O * Jp / 4)), dim3(256), 0, st, slab, splits_total, O, J, Jp, bm, bn,
                           Cb, I, R, S, grad_oihw, scale, natural ? 1 : 0);
    return dbn_status();
}

// Test hook: lower the per-launch index ranges so that the image chunking runs at small sizes (0 restores a default).
int dbn_set_index_limits(long pixel_rows, long bytes, long elems) {
    g_pixel_limit = pixel_rows > 0 ? pixel_rows : (1L << 24);
    g_byte_limit = bytes > 0 ? bytes : 0xF0000000L;
    g_elem_limit = elems > 0 ? elems : (1L << 32);
    return DBN_OK;
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
