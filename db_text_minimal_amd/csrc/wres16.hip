// 3x3 / stride-1 / pad-1 convolutions and their data gradients on 16-bit storage with the WEIGHT PANEL RESIDENT IN REGISTERS
// (round 6; replaces the pixel-patch launches of igemm_f32_kernel<128,64,...,PATCH> for 64 -> 64, 128 -> 128 and 256 -> 64 channels:
// /root/reference/src/modules/resnet.py:70-91 layer1 / layer2, segmentation_body.py:55-61 smooth convs, segmentation_head.py:24-29 /
// 64-68 first conv of each branch, forward and autograd backward).
//
// What bounded the pixel-patch kernel (profiles/r05_patch16_trace.txt): a 128-pixel tile streams the whole weight panel — 74 KB for
// 64 -> 64, 295 KB for 256 -> 64 — through an LDS ring for 16-64 KB of activations; every two-tap ring stage (128 clocks of MFMA
// work per wave) waited ~1000 clocks for weight fragments.  Here the panel never moves after the prologue:
//
//   * wave (ks, oc) of a workgroup owns input channels [64 ks, 64 ks + 64) x output channels [32 oc, 32 oc + 32) x 9 taps = 36 MFMA
//     A-operand fragments = 144 registers, loaded once.  Workgroup = (Cs / 64) x (Cd / 32) waves: 2 (64 -> 64) or 8 waves, two waves
//     per SIMD; all the waves of a CU together hold the panel.
//   * the activations stream: a workgroup walks DOWN a strip of 32 output columns, one output row (32 pixels) per step.  The 34-pixel
//     input rows arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, whole 128-byte lines per pixel) into a ring of D rows, PF = D - 3
//     rows ahead of their first use; every input row is fetched once per strip (34 / 32 of the tensor), never per tap.  LDS image of a
//     row: [pixel][Cs / 8 slices + 1 pad slot] x 16 B — the pad rotates the banks so that the ds_read_b128 fragment reads
//     (lane = pixel, 16 B = 8 channels) are conflict-free for every tap, and a tap is an IMMEDIATE offset of the read.
//   * operands swapped: A = weights (rows = output channels), B = pixels (columns), D[channel][pixel]: a lane holds ONE pixel and, per
//     accumulator register group, four consecutive channels — after v_permlane32_swap of group pairs 16 contiguous bytes, stored
//     directly (no LDS transpose of the output tile).
//   * Cs > 64: the (ks) waves' partial accumulators are summed through LDS in fixed order (ks = 0, 1, ...): wave (ks, oc) finishes
//     the register groups [ks (4 / KS), (ks + 1)(4 / KS)) of block oc.
//   * work = row blocks (image, strip, row) in that linear order, dealt to the workgroups of the launch in equal contiguous ranges
//     (a range may cross into the next strip: the ring is primed again there); one partial row of BatchNorm statistics /
//     BatchNorm-backward sums per workgroup — the partial rows a pixel-patch launch would have written beyond that are written as
//     empty rows (count 0), so the host-side row plumbing (conv.hip) is unchanged.
//
// One s_barrier per step (two with the K-split reduction) against one per 128 MFMA clocks before; the weights' LDS traffic (1 read per
// MFMA) and their L2 -> LDS DMA are gone: LDS carries one B fragment per MFMA (half its read bandwidth).
#include "igemm_common.h"

// profile by deletion (tools/flavour.sh <name> wres16.hip "-DDBN_WRES_DBG=<bits>"; timing only, results are wrong): 1 no fragment reads,
// 2 no DMA, 4 no stores, 8 no barriers, 16 no MFMAs
#ifndef DBN_WRES_DBG
#define DBN_WRES_DBG 0
#endif
extern "C" int dbn_g_wres16;  // conv.hip: dbn_set_wres16 (test / A-B hook; 1 = on)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CS, int CD>
struct WresGeom {
    static constexpr int KS = CS / 64, OC = CD / 32, NW = KS * OC, NT = NW * 64;
    static constexpr int SPP = CS / 8, PIX = SPP + 1, ROWPX = 34;  // 16-byte slots per pixel (+ 1 pad), pixels per input row
    static constexpr int ROW_SLOTS = ROWPX * PIX, ROW_DMA = (ROW_SLOTS + 63) / 64, ROW_PITCH = ROW_DMA * 64;
    static constexpr int DMA_PW = (ROW_DMA + NW - 1) / NW;  // DMA instructions per wave and row (the last round of a row may be short)
#ifndef DBN_WRES_D64
#define DBN_WRES_D64 6
#endif
#ifndef DBN_WRES_D128
#define DBN_WRES_D128 8
#endif
    static constexpr int D = CS == 128 ? DBN_WRES_D128 : CS == 256 ? 5 : DBN_WRES_D64, PF = D - 3;  // ring depth; rows in flight ahead of the three a step reads
    static constexpr int GPW = 4 / KS;                       // accumulator register groups (4 channels x 32 pixels) a wave finishes
    static constexpr int RED = KS > 1 ? 2 * OC * 4 * (KS - 1) * 64 : 0;  // K-split exchange: two buffers of every wave's NON-own register groups
    static constexpr int NCONST = 6, CONST_SLOTS = NCONST * 8;  // per wave: bias, pivot, mean, mask scale, mask shift, mean2 (32 floats each)
    static constexpr int SMEM = D * ROW_PITCH + 64 + RED + NW * CONST_SLOTS + 1;
    static constexpr int WG_PER_CU = NW == 2 ? 4 : 1;
    static_assert(KS == 1 || KS == 2 || KS == 4, "64, 128 or 256 input channels");
    static_assert(SMEM * 16 * WG_PER_CU <= 160 * 1024, "LDS");
};

enum { C_BIAS = 0, C_PIV, C_MEAN, C_MSC, C_MSH, C_MEAN2 };

// s_waitcnt vmcnt(n) for a compile-time n (the instruction takes an immediate)
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int AT>
__device__ __forceinline__ f32x4 cvt4(const u32x2 w) {  // four stored 16-bit values -> fp32
    if constexpr (AT == 1) {
        return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xFFFF0000u), __builtin_bit_cast(float, w[1] << 16),
                     __builtin_bit_cast(float, w[1] & 0xFFFF0000u)};
    } else {
        const dbn_f16x4 h = __builtin_bit_cast(dbn_f16x4, w);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
}
template <int AT>
__device__ __forceinline__ u32x2 pack4(const f32x4 v) {  // round to the storage type (nearest even), as dbn_st4
    if constexpr (AT == 1) {
        const dbn_bf16x2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
        return u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    } else {
        const dbn_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return __builtin_bit_cast(u32x2, h);
    }
}

// f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>): a fully unrolled loop with a compile-time index
template <int B, int E, class F>
__device__ __forceinline__ void dbn_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        dbn_static_for<B + 1, E>(f);
    }
}

// sum over the 32 lanes of each half-wave (every lane gets its half's total)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// MODE 0: forward conv; MODE 1: stride-1 data gradient (the panel's tap t pairs with the input offset (2 - t / 3, 2 - t % 3), as in
// igemm_f32_kernel's pixel-patch loop).  EPI 0: plain; 1: + the sums of the BatchNorm backward that consumes dst (IgemmParams::bnb_*);
// 2: + the statistics of the train-mode BatchNorm that follows (IgemmParams::stats).  Compile-time, because the per-lane sums are
// registers the plain kernel cannot spare: 144 of a wave's 256 hold the panel.
template <int AT, int CS, int CD, int MODE, int EPI>
__global__ __launch_bounds__((CS / 64) * (CD / 32) * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_wres16_kernel(const IgemmParams p, const int T, const int nstrip) {
    using G = WresGeom<CS, CD>;
    constexpr int KS = G::KS, OC = G::OC, NW = G::NW, NT = G::NT, PIX = G::PIX, SPP = G::SPP, ROW_DMA = G::ROW_DMA, ROW_PITCH = G::ROW_PITCH;
    constexpr int DMA_PW = G::DMA_PW, D = G::D, PF = G::PF, GPW = G::GPW;
    static_assert(AT == 1 || AT == 2, "bf16 / fp16 storage");
    static_assert(EPI == 0 || AT == 1, "sums / statistics epilogues: bf16 storage (training)");
    // the second BatchNorm of the sums epilogue (a projection shortcut's, IgemmParams::bnb_y2) exists from 128 channels on (resnet.py:84-91);
    // with 64 channels a lane finishes 16 channels and a third sum per channel does not fit the registers
    constexpr bool Y2 = EPI == 1 && KS > 1;

    __shared__ f32x4 smem[G::SMEM];
    f32x4* const ring = smem;
    f32x4* const dummy = smem + D * ROW_PITCH;
    f32x4* const red = dummy + 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / OC, oc = wave - ks * OC;
    const int li = lane & 31, lh = lane >> 5;
    float* const cst = reinterpret_cast<float*>(red + G::RED + wave * G::CONST_SLOTS);  // this wave's constants: [NCONST][32]
    const int H = p.Hdf, W = p.Wdf;

    // ---- the weight panel: 36 fragments of 16 bytes per lane, fetched once.  Panel layout (pack_weights_bf16s_kernel): k-tile
    // kt = (16-channel block) * 9 + tap, [kt][k half][Cd][8]
    u32x4 wf[9][4];
    {
        const __amdgpu_buffer_rsrc_t rsrcW =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (unsigned)(9 * (CS / 16) * 2 * CD * 16), 0x00020000);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kt = (ks * 4 + kk) * 9 + tap;
                wf[tap][kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)(((kt * 2 + lh) * CD + oc * 32 + li) * 16), 0, 0);
            }
    }
    // ---- per-channel constants of this wave's 32 output channels, to LDS (read back per step as 16-byte pieces: registers are for
    // the panel)
    if (lane < 32) {
        const int c = oc * 32 + lane;
        cst[C_BIAS * 32 + lane] = p.bias ? p.bias[c] : 0.f;
        cst[C_PIV * 32 + lane] = 0.f;
        if constexpr (EPI == 1) {
            cst[C_MEAN * 32 + lane] = p.bnb_mean[c];
            cst[C_MSC * 32 + lane] = p.bnb_zmask ? 0.f : p.bnb_msc[c];
            cst[C_MSH * 32 + lane] = p.bnb_zmask ? 0.f : p.bnb_msh[c];
            cst[C_MEAN2 * 32 + lane] = p.bnb_y2 ? p.bnb_mean2[c] : 0.f;
        }
    }
    // ---- this workgroup's range of row blocks
    const int nwg = (int)gridDim.x;
    int t = (int)((long)blockIdx.x * T / nwg);
    const int t_end = (int)((long)(blockIdx.x + 1) * T / nwg);

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);
    const unsigned dst_bytes = (unsigned)((long)p.N * H * W * CD * 2);
    const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.dst), 0, dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(EPI == 1 ? p.bnb_y : p.dst), 0, dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcZ =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((EPI == 1 && p.bnb_zmask) ? p.bnb_zmask : p.dst), 0, dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((EPI == 1 && p.bnb_y2) ? p.bnb_y2 : p.dst), 0, dst_bytes, 0x00020000);

    // DMA instruction i of this wave covers slots [64 j, 64 j + 64) of a row, j = wave + i NW: lane -> (pixel, slice)
    int d_pp[DMA_PW], d_sl[DMA_PW];
#pragma unroll
    for (int i = 0; i < DMA_PW; ++i) {
        const int j = wave + i * NW, q = 64 * j + lane;
        const int pp = q / PIX, sl = q - pp * PIX;
        const bool on = j < ROW_DMA && pp < G::ROWPX && sl < SPP;
        d_pp[i] = on ? pp : -4096;  // (never inside the map)
        d_sl[i] = sl;
    }
    // fragment reads: lane (li, lh) reads slice (8 ks + 2 kk + lh) of pixel li + dx of a row slot
    const int frag_base = li * PIX + lh + ks * 8;

    // statistics of this workgroup's pixels (train-mode BatchNorm of the output: IgemmParams::stats), per register
    constexpr bool want_stats = EPI == 2;
    constexpr int NS1 = EPI != 0 ? GPW : 1, NS4 = Y2 ? GPW : 1;
    f32x4 s1[NS1], s2[NS1], s4[NS4];
#pragma unroll
    for (int gq = 0; gq < NS1; ++gq) s1[gq] = s2[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < NS4; ++gq) s4[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
    int npx = 0;         // valid pixels of this workgroup
    bool first = true;   // first finished row of the workgroup: its first pixel gives the pivot of the statistics
    bool primed = false;

    // ---- the epilogue of one output row, in pieces (KS > 1: called from inside the NEXT row's MFMA stream).  v: this wave's register
    // groups — channel 32 oc + 8 (ks GPW + gq) + 4 lh + e of pixel li — off: byte offset of that pixel's channel 0 in dst / y / mask / res
    // (out of range past the map's edge: loads return 0, stores are dropped), ok: the pixel is inside the map
    u32x2 oldv[GPW], yv[GPW], zv[GPW], y2v[GPW];  // operands (pre-swap layout: 8 bytes = 4 channels per register group)
    auto fetch_operands = [&](auto G0, auto G1, unsigned off) {
#pragma unroll
        for (int gq = decltype(G0)::value; gq < decltype(G1)::value; ++gq) {
            const int cb = (oc * 32 + 8 * (ks * GPW + gq) + 4 * lh) * 2;
            if (p.accumulate) oldv[gq] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrcA, (int)(off + cb), 0, 0));
            if constexpr (EPI == 1) {
                yv[gq] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrcY, (int)(off + cb), 0, 0));
                if (p.bnb_zmask) zv[gq] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrcZ, (int)(off + cb), 0, 0));
                if constexpr (Y2) {
                    if (p.bnb_y2) y2v[gq] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrcY2, (int)(off + cb), 0, 0));
                }
            }
        }
    };
    // bias / accumulate / ReLU, statistics or sums, rounding: v -> pk
    auto finish_math = [&](auto G0, auto G1, f32x4 (&v)[GPW], u32x2 (&pk)[GPW], bool ok) {
        constexpr int g0 = decltype(G0)::value, g1 = decltype(G1)::value;
#pragma unroll
        for (int gq = g0; gq < g1; ++gq) {
            const int cl = 8 * (ks * GPW + gq) + 4 * lh;  // first of the four channels, within this wave's 32
            v[gq] += *reinterpret_cast<const f32x4*>(cst + C_BIAS * 32 + cl);
            if (p.accumulate) v[gq] += cvt4<AT>(oldv[gq]);
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[gq][e] = fmaxf(v[gq][e], 0.f);
            }
        }
        if constexpr (want_stats) {
            if (first) {  // pivot = the workgroup's first pixel (lanes 0 and 32 hold it)
                if (li == 0) {
#pragma unroll
                    for (int gq = g0; gq < g1; ++gq) *reinterpret_cast<f32x4*>(cst + C_PIV * 32 + 8 * (ks * GPW + gq) + 4 * lh) = v[gq];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int gq = g0; gq < g1; ++gq) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(cst + C_PIV * 32 + 8 * (ks * GPW + gq) + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = ok ? v[gq][e] - pv[e] : 0.f;
                    s1[gq][e] += d;
                    s2[gq][e] += d * d;
                }
            }
        }
#pragma unroll
        for (int gq = g0; gq < g1; ++gq) pk[gq] = pack4<AT>(v[gq]);
        if constexpr (EPI == 1) {
            // sums of the BatchNorm backward over the values AS STORED: g = dz [mask > 0]; s1 += g; s2 += g (y - mean) (rstd at the end)
#pragma unroll
            for (int gq = g0; gq < g1; ++gq) {
                const int cl = 8 * (ks * GPW + gq) + 4 * lh;
                const f32x4 dz = cvt4<AT>(pk[gq]), yy = cvt4<AT>(yv[gq]);
                const f32x4 mu = *reinterpret_cast<const f32x4*>(cst + C_MEAN * 32 + cl);
                f32x4 m;
                if (p.bnb_zmask) {
                    m = cvt4<AT>(zv[gq]);
                } else {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(cst + C_MSC * 32 + cl), sh = *reinterpret_cast<const f32x4*>(cst + C_MSH * 32 + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = dbn_affine(yy[e], sc[e], sh[e]);
                }
                f32x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = (ok && m[e] > 0.f) ? dz[e] : 0.f;
                    s1[gq][e] += g[e];
                    s2[gq][e] += g[e] * (yy[e] - mu[e]);
                }
                if constexpr (Y2) if (p.bnb_y2) {
                    const f32x4 y2 = cvt4<AT>(y2v[gq]), mu2 = *reinterpret_cast<const f32x4*>(cst + C_MEAN2 * 32 + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s4[gq][e] += g[e] * (y2[e] - mu2[e]);
                }
            }
        }
    };
    // store: a pair of register groups exchanged between the half-waves -> 16 contiguous bytes per lane (one group: 8 bytes)
    auto store_groups = [&](auto G0, const u32x2 (&pk)[GPW], unsigned off) {
        constexpr int g0 = decltype(G0)::value;
        if constexpr (DBN_WRES_DBG & 4) off = pk[g0][0] == 0x12345678u ? off : OOB_OFFSET;
        if constexpr (GPW >= 2) {
            const auto a = __builtin_amdgcn_permlane32_swap(pk[g0][0], pk[g0 + 1][0], false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(pk[g0][1], pk[g0 + 1][1], false, false);
            const u32x4 w = {a[0], b[0], a[1], b[1]};
            const int cb = (oc * 32 + 8 * (ks * GPW + g0 + lh)) * 2;
            __builtin_amdgcn_raw_buffer_store_b128(w, rsrcD, (int)(off + cb), 0, 0);
        } else {
            const int cb = (oc * 32 + 8 * (ks * GPW) + 4 * lh) * 2;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk[0]), rsrcD, (int)(off + cb), 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IG = std::integral_constant<int, GPW>;
    constexpr int NST = GPW >= 2 ? GPW / 2 : 1;  // stores per finished row and wave
    // K-split exchange (KS > 1): a wave keeps its OWN register groups and writes the others' to red[buffer][oc][group][source != owner];
    // the owner sums in the order ks = 0, 1, ... (its own share at position ks).  Two buffers: row y's partials are read inside row y + 1's
    // MFMA stream while row y + 1's are being produced — one barrier per row.
    constexpr int RED1 = KS > 1 ? OC * 4 * (KS - 1) * 64 : 0;
    f32x4 own[GPW];            // KS > 1: this wave's share of its own groups of the previous row
    f32x4 oth[GPW][KS > 1 ? KS - 1 : 1];
    bool have_prev = false;    // KS > 1: a row waits for its epilogue
    unsigned prev_off = OOB_OFFSET;
    bool prev_ok = false;
    int prev_buf = 0;
    auto read_partials = [&]() {
        if constexpr (KS > 1) {
#pragma unroll
            for (int gq = 0; gq < GPW; ++gq)
#pragma unroll
                for (int q = 0; q < KS - 1; ++q) oth[gq][q] = red[prev_buf * RED1 + ((oc * 4 + ks * GPW + gq) * (KS - 1) + q) * 64 + lane];
        }
    };
    auto sum_partials = [&](f32x4 (&v)[GPW]) {
        if constexpr (KS > 1) {
#pragma unroll
            for (int gq = 0; gq < GPW; ++gq) {
                // sources in the order ks' = 0 .. KS - 1; position q of `oth` holds source (q < ks ? q : q + 1)
                f32x4 s = ks == 0 ? own[gq] : oth[gq][0];
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) s += (k2 == ks) ? own[gq] : oth[gq][k2 < ks ? k2 : k2 - 1];
                v[gq] = s;
            }
        }
    };

    while (t < t_end) {
        // ---- segment: rows [y0, y1) of strip sx of image n
        const int sid = t / H, y0 = t - sid * H;
        const int n = sid / nstrip, sx = sid - n * nstrip;
        const int x0 = sx * 32;
        const int y1 = min(H, y0 + (t_end - t));
        t += y1 - y0;
        npx += (y1 - y0) * min(32, W - x0);
        bool xv[DMA_PW];
        unsigned coloff[DMA_PW];
#pragma unroll
        for (int i = 0; i < DMA_PW; ++i) {
            const int x = x0 - 1 + d_pp[i];
            xv[i] = (unsigned)x < (unsigned)W;
            coloff[i] = (unsigned)(x * (CS * 2) + d_sl[i] * 16);
        }
        const unsigned img_base = (unsigned)n * (unsigned)(H * W) * (unsigned)(CS * 2);
        // DMA instruction i of this wave for row r (none where wave + i NW >= ROW_DMA: the last round of a row is short)
        auto issue_piece = [&](int r, int slot, auto I) {
            constexpr int i = decltype(I)::value;
            const int j = wave + i * NW;
            if (j >= ROW_DMA) return;  // (wave-uniform)
            const bool rv = (unsigned)r < (unsigned)H && r <= y1;  // (rows past this segment's halo: never read — zeros, no traffic)
            const unsigned rb = img_base + (unsigned)r * (unsigned)(W * CS * 2);
            const unsigned off = (rv && xv[i] && !(DBN_WRES_DBG & 2)) ? rb + coloff[i] : OOB_OFFSET;
            if constexpr (DBN_WRES_DBG & 2) {
                if (r > -100) return;  // (always: keeps the code, skips the instruction)
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ring + slot * ROW_PITCH + 64 * j), 16, (int)off, 0, 0, 0);
        };
        auto issue_row = [&](int r, int slot) { dbn_static_for<0, DMA_PW>([&](auto I) { issue_piece(r, slot, I); }); };
        if (primed) {
            // the previous segment's last reads are done in every wave, and none of its rows is still landing
            wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        primed = true;
#pragma unroll
        for (int k = 0; k < PF + 2; ++k) issue_row(y0 - 1 + k, k);
        int sl0 = 0;  // ring slot of row y - 1
        // this lane's output pixel: byte offset of its channel 0 in dst (and y / mask / res), or out of range past the map's edge
        const bool px_ok = x0 + li < W;
        unsigned pix_off = (unsigned)((n * H + y0) * W + x0 + li) * (unsigned)(CD * 2);
        for (int y = y0; y < y1; ++y) {
            // ---- row y + 1 has landed: completion is in order, so at most the operations issued after its DMA may be outstanding — the
            // (PF - 1) younger rows and the stores issued since (NST per finished row; the epilogue's loads only add to what is younger:
            // leaving them out of the count is the strict side).  KS == 1: the stores of steps k - PF .. k - 1 (those that exist);
            // KS > 1: a step stores the PREVIOUS row, so the segment's first step issues none.
            const int k = y - y0;
            const int nst = KS == 1 ? (k < PF ? k : PF) : (k > PF ? PF : (k > 0 ? k - 1 : 0));
            auto wait_row = [&](auto ND) {  // ND: DMA instructions of this wave per row
                constexpr int nd = decltype(ND)::value;
                if (nst >= PF) wait_vm<(PF - 1) * nd + PF * NST>();
                else if (nst == 0) wait_vm<(PF - 1) * nd>();
                else if (nst == 1) wait_vm<(PF - 1) * nd + NST>();
                else if (nst == 2) wait_vm<(PF - 1) * nd + 2 * NST>();
                else if (nst == 3) wait_vm<(PF - 1) * nd + 3 * NST>();
                else wait_vm<(PF - 1) * nd + 4 * NST>();
            };
            constexpr int FULLW = ROW_DMA - NW * (DMA_PW - 1);  // waves 0 .. FULLW - 1 issue DMA_PW instructions per row, the others one fewer
            if (wave < FULLW) wait_row(std::integral_constant<int, DMA_PW>{});
            else wait_row(std::integral_constant<int, DMA_PW - 1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(DBN_WRES_DBG & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the epilogue's operands, fetched under the MFMAs and issued BEFORE this step's DMA (completion is in order: waiting for
            // them must not mean waiting for the youngest row).  KS == 1: of this row — the first PAIR of register groups only (a lane
            // finishes 16 channels there: the second pair's follow the MFMAs, the registers do not hold both beside the panel and the
            // per-lane sums); KS > 1: of the previous row, whose epilogue runs inside this row's MFMA stream.
            const unsigned my_off = px_ok ? pix_off : OOB_OFFSET;
            constexpr int GFIRST = GPW >= 2 ? 2 : 1;
            if constexpr (KS == 1) fetch_operands(I0{}, std::integral_constant<int, GFIRST>{}, my_off);
            else if (have_prev) fetch_operands(I0{}, IG{}, prev_off);
            // every wave is past step y - 1: the slot of row y - 2 is free for row y + 1 + PF — its DMA instructions are dealt into the MFMA
            // stream below (an LDS-DMA instruction takes 60-180 clocks to issue: at the head of the step, with every wave of the workgroup
            // just released by the barrier, the matrix pipe stood still for all of them; even / odd waves take different slots)
            const int dma_row = y + 1 + PF, dma_slot = sl0 == 0 ? D - 1 : sl0 - 1;
            asm volatile("" ::: "memory");

            // ---- 36 MFMAs: D[channel][pixel] += W[channel][k] . X[k][pixel].  The B fragments come from LDS in a ROLLING prefetch, BDEPTH
            // reads ahead of their MFMA, the order pinned by scheduling barriers: left to the scheduler the loop read two fragments,
            // waited, issued two MFMAs — every pair exposed a whole LDS latency (first build: 0.40 of the roofline, 444 clocks per pair)
            // ONE accumulator chain per wave: with two waves per SIMD that is what the matrix pipe wants (tools/mfma_peak.py, random fp16
            // operands: 1 chain x 2 waves 1640 TFLOP/s = the box's power-capped ceiling, 2 chains x 2 waves 1245, 4 x 2 1670; a two-chain
            // build of this kernel measured the same as this one and cost 16 registers)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const f32x4* rp3[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                int slot = sl0 + dy;
                slot = slot >= D ? slot - D : slot;
                rp3[dy] = ring + slot * ROW_PITCH + frag_base;
            }
            // (tried and dropped: the first fragments of the NEXT step fetched before the barrier, the bias as the accumulators' start value —
            // 986 vs 950 us on 256 -> 64 at 32 x 320^2, and the carried registers made the statistics / sums variants spill)
            constexpr int BDEPTH = EPI == 0 ? 4 : (KS == 1 ? 3 : 2);  // (the epilogues with per-lane sums have fewer registers to spare)
            f32x4 bq[36];
            auto rd = [&](auto I) {
                constexpr int i = decltype(I)::value, dy = i / 12, dx = (i % 12) / 4, kk = i % 4;
                if constexpr (DBN_WRES_DBG & 1) bq[i] = __builtin_bit_cast(f32x4, wf[(i + 1) % 9][kk]);
                else bq[i] = rp3[dy][dx * PIX + kk * 2];
            };
            auto mm = [&](auto I) {
                constexpr int i = decltype(I)::value, dy = i / 12, dx = (i % 12) / 4, kk = i % 4;
                constexpr int tap = MODE == 0 ? dy * 3 + dx : (2 - dy) * 3 + (2 - dx);
                if constexpr (DBN_WRES_DBG & 16) {
                    acc[i % 16] += bq[i][0] + bq[i][3];
                    return;
                }
                if constexpr (AT == 2)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[tap][kk]), __builtin_bit_cast(f16x8, bq[i]), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[tap][kk]), __builtin_bit_cast(bf16x8, bq[i]), acc, 0, 0, 0);
            };
            f32x4 pv_[GPW];   // KS > 1: the previous row's values on their way through the epilogue
            u32x2 ppk[GPW];
            dbn_static_for<0, BDEPTH>([&](auto I) { rd(I); });
            dbn_static_for<0, 36>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i + BDEPTH < 36) rd(std::integral_constant<int, i + BDEPTH>{});
                if constexpr (i >= 4 && i <= 24 && i % 4 == 0) {
                    constexpr int slot_ = (i - 4) / 4;  // 0 .. 5: even waves take slots 0, 2, 4, odd waves 1, 3, 5
                    if ((slot_ & 1) == (wave & 1)) {
                        if constexpr (slot_ / 2 < DMA_PW) issue_piece(dma_row, dma_slot, std::integral_constant<int, (slot_ / 2 < DMA_PW ? slot_ / 2 : 0)>{});
                    }
                }
                if constexpr (KS > 1) {
                    // the previous row's epilogue, dealt into this row's MFMA stream (wave-uniform branches; the matrix pipe keeps running)
                    if constexpr (i == 2) { if (have_prev) read_partials(); }
                    if constexpr (i == 10) { if (have_prev) sum_partials(pv_); }
                    if constexpr (i == 18) { if (have_prev) finish_math(I0{}, IG{}, pv_, ppk, prev_ok); }
                    if constexpr (i == 27) { if (have_prev) { store_groups(I0{}, ppk, prev_off); first = false; } }
                }
                __builtin_amdgcn_sched_barrier(0);
                mm(I);
                __builtin_amdgcn_sched_barrier(0);
            });

            // ---- acc register r holds channel 32 oc + 8 (r >> 2) + 4 lh + (r & 3) of pixel li
            if constexpr (KS == 1) {
                f32x4 v[GPW];
                u32x2 pk[GPW];
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                asm volatile("" ::: "memory");
                fetch_operands(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, my_off);
                finish_math(I0{}, std::integral_constant<int, 2>{}, v, pk, px_ok);
                store_groups(I0{}, pk, my_off);
                asm volatile("" ::: "memory");
                finish_math(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, v, pk, px_ok);
                store_groups(std::integral_constant<int, 2>{}, pk, my_off);
                first = false;
            } else {
                const int buf = k & 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 part = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    const int ko = g / GPW;  // the wave row that finishes group g
                    if (ko == ks) own[g % GPW] = part;  // (wave-uniform)
                    else red[buf * RED1 + ((oc * 4 + g) * (KS - 1) + (ks < ko ? ks : ks - 1)) * 64 + lane] = part;
                }
                have_prev = true;
                prev_off = my_off;
                prev_ok = px_ok;
                prev_buf = buf;
            }
            pix_off += (unsigned)(W * CD * 2);
            sl0 = sl0 + 1 == D ? 0 : sl0 + 1;
        }
        if constexpr (KS > 1) {
            // the segment's last row: its partials published, then its epilogue alone
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            f32x4 v[GPW];
            u32x2 pk[GPW];
            fetch_operands(I0{}, IG{}, prev_off);
            read_partials();
            sum_partials(v);
            finish_math(I0{}, IG{}, v, pk, prev_ok);
            store_groups(I0{}, pk, prev_off);
            first = false;
            have_prev = false;
        }
    }
    wait_vm<0>();

    // ---- this workgroup's partial row(s)
    const int trow = p.stat_row0 + (int)blockIdx.x;
    if constexpr (want_stats) {
#pragma unroll
        for (int gq = 0; gq < GPW; ++gq) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(cst + C_PIV * 32 + 8 * (ks * GPW + gq) + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = half_sum(s1[gq][e]), b = half_sum(s2[gq][e]);
                if (li == 0) {
                    const long c = oc * 32 + 8 * (ks * GPW + gq) + 4 * lh + e;
                    dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (0L * CD + c) * p.stat_rows + trow, pv[e]);
                    dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (1L * CD + c) * p.stat_rows + trow, a);
                    dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (2L * CD + c) * p.stat_rows + trow, b);
                }
            }
        }
        if (tid == 0) dbn_stat_put(p.bnf_cnt != nullptr, p.stats + 3L * CD * p.stat_rows + trow, (float)npx);
        // the rows a pixel-patch launch would have written beyond this launch's workgroups: empty
        for (int r = (int)blockIdx.x + nwg; r < p.launch_rows; r += nwg) {
            for (int i = tid; i < 3 * CD; i += NT) dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (long)i * p.stat_rows + p.stat_row0 + r, 0.f);
            if (tid == 0) dbn_stat_put(p.bnf_cnt != nullptr, p.stats + 3L * CD * p.stat_rows + p.stat_row0 + r, 0.f);
        }
        // optional in-kernel finalize (IgemmParams::bnf_cnt, igemm_common.h): one arrival per partial row, the empty ones included
        for (int r = (int)blockIdx.x; r < p.launch_rows; r += nwg) {
            __syncthreads();
            dbn_bn_stats_finish(DBN_BNF_ARGS(p), p.stat_row0 + r, 0, 0, CD, reinterpret_cast<int*>(smem + G::SMEM - 1), reinterpret_cast<double*>(smem));
        }
    }
    if constexpr (EPI == 1) {
        auto xst = [](float* ptr, float v_) { __hip_atomic_store(ptr, v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto xld = [](const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto put = [&](float* ptr, float v_) {
            if (p.bnb_cnt) xst(ptr, v_);  // read by another workgroup of this launch: memory-side store (igemm_kernel.h bnb_finish)
            else *ptr = v_;
        };
#pragma unroll
        for (int gq = 0; gq < GPW; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = half_sum(s1[gq][e]), b = half_sum(s2[gq][e]);
                float d = 0.f;
                if constexpr (Y2) d = p.bnb_y2 ? half_sum(s4[gq][e]) : 0.f;
                if (li == 0) {
                    const long c = oc * 32 + 8 * (ks * GPW + gq) + 4 * lh + e;
                    put(p.bnb_part + (0L * CD + c) * p.stat_rows + trow, a);
                    put(p.bnb_part + (1L * CD + c) * p.stat_rows + trow, b * p.bnb_rstd[c]);
                    if (p.bnb_y2) {
                        put(p.bnb_part2 + (0L * CD + c) * p.stat_rows + trow, a);
                        put(p.bnb_part2 + (1L * CD + c) * p.stat_rows + trow, d * p.bnb_rstd2[c]);
                    }
                }
            }
        const int nbn = p.bnb_y2 ? 2 : 1;
        for (int r = (int)blockIdx.x + nwg; r < p.launch_rows; r += nwg)
            for (int i = tid; i < nbn * 2 * CD; i += NT) {
                const int b = i / (2 * CD), j = i - b * 2 * CD;
                put((b ? p.bnb_part2 : p.bnb_part) + (long)j * p.stat_rows + p.stat_row0 + r, 0.f);
            }
        // optional in-kernel finalize (IgemmParams::bnb_cnt): the protocol of igemm_kernel.h's bnb_finish, one arrival per partial row
        if (p.bnb_cnt) {
            constexpr int GR = 64;
            const int NG = (p.stat_rows + GR - 1) / GR;
            int* const cnt = p.bnb_cnt;  // (one tile column: Cd == CD)
            int* const s_flag = reinterpret_cast<int*>(smem + G::SMEM - 1);
            for (int r = (int)blockIdx.x; r < p.launch_rows; r += nwg) {
                const int trow_ = p.stat_row0 + r, g = trow_ / GR;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                DBN_RACE_JITTER();
                if (tid == 0) {
                    const int gsize = min(GR, p.stat_rows - g * GR);
                    const int last = __hip_atomic_fetch_add(cnt + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
                    if (last) __hip_atomic_store(cnt + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_flag = last;
                }
                __syncthreads();
                if (!*s_flag) continue;
                asm volatile("" ::: "memory");
                const int r0 = g * GR, r1_ = min(p.stat_rows, r0 + GR);
                for (int it = tid; it < nbn * 2 * CD; it += NT) {
                    const int b = it / (2 * CD), k2 = (it / CD) & 1, cl = it % CD;
                    const float* src_ = (b ? p.bnb_part2 : p.bnb_part) + ((long)k2 * CD + cl) * p.stat_rows;
                    double s = 0.0;
                    int rr = r0;
                    for (; rr + 3 < r1_; rr += 4) s += ((double)xld(src_ + rr) + (double)xld(src_ + rr + 1)) + ((double)xld(src_ + rr + 2) + (double)xld(src_ + rr + 3));
                    for (; rr < r1_; ++rr) s += (double)xld(src_ + rr);
                    xst(p.bnb_grp + (((long)b * 2 + k2) * CD + cl) * NG + g, (float)s);
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
                if (!*s_flag) continue;
                asm volatile("" ::: "memory");
                for (int it = tid; it < nbn * CD; it += NT) {
                    const int b = it / CD, c = it % CD;
                    const float* g1 = p.bnb_grp + (((long)b * 2 + 0) * CD + c) * NG;
                    const float* g2 = p.bnb_grp + (((long)b * 2 + 1) * CD + c) * NG;
                    double a_ = 0.0, b_ = 0.0;
                    for (int q_ = 0; q_ < NG; ++q_) {
                        a_ += (double)xld(g1 + q_);
                        b_ += (double)xld(g2 + q_);
                    }
                    p.bnb_dbeta[b][c] = (float)(a_ * p.bnb_gscale);
                    p.bnb_dgamma[b][c] = (float)(b_ * p.bnb_gscale);
                    p.bnb_c1c2[b][c] = (float)(a_ * p.bnb_invM);
                    p.bnb_c1c2[b][CD + c] = (float)(b_ * p.bnb_invM);
                }
            }
        }
    }
}

template <int AT, int CS, int CD>
int launch_wres(IgemmParams& p, int mode, hipStream_t st) {
    using G = WresGeom<CS, CD>;
    const int nstrip = (p.Wdf + 31) / 32;
    const long T = (long)p.N * nstrip * p.Hdf;
    const int rows = dbn_ceil_div((long)p.N * p.Hdf * p.Wdf, 128);  // the partial rows of the pixel-patch launch this replaces
    if (p.stat_rows <= 0) p.stat_rows = rows;
    p.launch_rows = rows;
    long grid = 256L * G::WG_PER_CU;
    grid = grid > rows ? rows : grid;
    grid = grid > T ? T : grid;
    if (grid <= 0) return DBN_OK;
    const dim3 g((unsigned)grid), b(G::NT);
    if (p.bnb_part || p.stats) {
        if constexpr (AT == 1) {
            if (p.bnb_part && p.stats) return DBN_ERR_ARG;
            if constexpr (CS != 256) {  // (the sums epilogue is built for 256 -> 64 only: dbn_wres16_eligible)
                if (p.bnb_part) return DBN_ERR_ARG;
            }
            if (p.bnb_part) {
                if constexpr (CS == 256) {
                    if (mode == 0) hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 0, 1>), g, b, 0, st, p, (int)T, nstrip);
                    else hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 1, 1>), g, b, 0, st, p, (int)T, nstrip);
                }
            } else {
                if (mode == 0) hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 0, 2>), g, b, 0, st, p, (int)T, nstrip);
                else hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 1, 2>), g, b, 0, st, p, (int)T, nstrip);
            }
            return dbn_status();
        } else {
            return DBN_ERR_ARG;
        }
    }
    if (mode == 0) hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 0, 0>), g, b, 0, st, p, (int)T, nstrip);
    else hipLaunchKernelGGL((conv3x3_wres16_kernel<AT, CS, CD, 1, 0>), g, b, 0, st, p, (int)T, nstrip);
    return dbn_status();
}

}  // namespace

// Does this pixel-patch-eligible launch (3x3, stride 1, pad 1, 16-bit storage, mode 0 / 1, no split-K: checked by the caller) take the
// weight-resident kernel?  Channel pairs whose panel fits the registers of one workgroup; maps whose ragged last strip wastes at most a
// quarter of the MFMA work (W % 32 == 0, or at least 16 of the last 32 columns real and three strips).  bnb / y2: the call carries the
// BatchNorm-backward sums epilogue / its second BatchNorm.
bool dbn_wres16_eligible(int at, int mode, int N, int H, int W, int Cs, int Cd, bool bnb, bool y2, bool stats) {
    if (!dbn_g_wres16 || !(at == 1 || at == 2) || !(mode == 0 || mode == 1)) return false;
    if (!((Cs == 64 && Cd == 64) || (Cs == 128 && Cd == 128) || (Cs == 256 && Cd == 64))) return false;
    if ((bnb || stats) && at != 1) return false;
    // 64 -> 64 with the sums epilogue: a lane finishes 16 channels x 2 sums beside the 144 panel registers — the allocator spills panel
    // fragments into the MFMA loop (36 VGPRs at two waves per SIMD): those launches stay on the pixel-patch kernel
    if (bnb && Cs == 64) return false;
    // 128 -> 128 with the sums epilogue: the software-pipelined epilogue (previous row's sums inside this row's MFMA stream) beside three
    // per-channel sums and four operand tensors spills 36 registers: pixel-patch kernel
    if (bnb && Cs == 128) return false;
    // maps whose width is not a multiple of the 32-column strip: measured slower than the pixel-patch kernel (128 -> 128 at 16 x 80^2:
    // 38-40 vs 35-36 us — a third of the last strip's MFMAs are padding); dbn_set_wres16(2) lifts the rule (tests of the ragged strip)
    if (W % 32 != 0 && dbn_g_wres16 != 2) return false;
    return (long)N * H * W * Cd * 2 < 0xF0000000L && (long)N * ((W + 31) / 32) * H < 0x7FFFFFFFL;
}

int dbn_launch_wres16(IgemmParams& p, int mode, int at, hipStream_t st) {
    if (at == 1) {
        if (p.Cs == 64) return launch_wres<1, 64, 64>(p, mode, st);
        if (p.Cs == 128) return launch_wres<1, 128, 128>(p, mode, st);
        return launch_wres<1, 256, 64>(p, mode, st);
    }
    if (p.Cs == 64) return launch_wres<2, 64, 64>(p, mode, st);
    if (p.Cs == 128) return launch_wres<2, 128, 128>(p, mode, st);
    return launch_wres<2, 256, 64>(p, mode, st);
}
