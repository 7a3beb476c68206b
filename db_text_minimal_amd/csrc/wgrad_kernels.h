// Weight-gradient kernels; instantiated by wgrad_f32.hip (exact fp32) and wgrad_b16.hip (bf16 math).
#pragma once
#include "igemm_common.h"

// (global scope: the launchers of the kernel translation units take it across translation units)
struct WgradParams {
    const void* sm;    // [N,Ho,Wo,O]  (indexes the reduction), activation type AT
    const void* big;   // [N,H,W,Cb], activation type AT
    float* slab;       // [splitk][O][J]
    int N, Ho, Wo, O, H, W, Cb, R, S, stride, pad;
    int P, J, pchunk;
    float rcp_HWo, rcp_Wo;
    unsigned sm_bytes, big_bytes;
    unsigned sm_plane_bytes, big_plane_bytes;  // AT = 3: distance of the three bf16 planes of each operand
    int row_tw;                                // ROW addressing: width of the 16-pixel blocks (16, 8, 4); 0 = general gather
};

// Width TW of the TW x 16/TW pixel blocks the reduction can be cut into (0: none — odd sizes keep the general gather)
static inline int wgrad_row_tw(int Ho, int Wo) { return Wo % 16 == 0 ? 16 : (Wo % 8 == 0 && Ho % 2 == 0) ? 8 : (Wo % 4 == 0 && Ho % 4 == 0) ? 4 : 0; }

#ifndef DBN_DBG
#define DBN_DBG 0
#endif

namespace {

// --------------------------------------------------------------------------------
// weight gradient
// --------------------------------------------------------------------------------
// Distance (floats) between the slabs of consecutive pixel splits.  O*Jp alone is a multiple of 64 KB for most layers
// (e.g. 64 x 2304 floats = 9 x 64 KB): the reduction then reads its `splits` addends from addresses that all map to the same
// HBM channel / L2 slice and crawls (50 MB in 130 us).  4352 bytes of padding rotate consecutive slabs across the channels.
__host__ __device__ inline long wgrad_slab_stride(long O, long Jp) { return O * Jp + 1088; }


// Position <-> index permutation of a tile edge of length B (B % 4 == 0): the staging threads
// transpose 4x4 blocks (4 pixels x 4 channels) in registers and write channel 4c+e to LDS position
// e*(B/4)+c, which keeps both the ds_write_b128 of the staging pass and the ds_read_b128 of the MFMA
// fragments conflict-free.  The accumulators (and the slabs) therefore live in "position space".
__host__ __device__ __forceinline__ int tile_pos_to_index(int pos, int B) { return 4 * (pos % (B / 4)) + pos / (B / 4); }

// AT = 1 (bf16 activations and gradients in HBM, NS = 1): the staging threads fetch their 4 pixels x 4 channels as four
// 8-byte loads and transpose the 16-bit values with two bit operations per output word — no conversion.
// ROW = 1 (host: p.row_tw = TW in {16, 8, 4} with TW | Wo and TH = 16 / TW | Ho): the reduction runs over the pixels in blocks of
// TW x TH (any order of a sum is a weight gradient; the order is fixed, so the result stays deterministic) — a k-tile is one such
// block, its position (n, oh0, ow0) is workgroup-uniform and moves with scalar instructions; each load is a loop-invariant
// per-thread byte offset + a scalar offset (the s-offset of the buffer instruction), and what is left for the vector unit per
// k-tile is the padding test of the gather (14 instructions instead of ~60: profile by deletion had put the address math at a
// fifth of the kernel's time, tools/wgrad_deletion_probe.py).  The hardware does not range-check the scalar offset, hence the
// scalar position stops at the last k-tile of the split (the prefetch runs D tiles past the end and re-reads it).
// TW = 16 visits the pixels in the order of the general gather (ROW = 0): bit-identical results.
template <int BM, int BN, int WM, int WN, int NS, int AT = 0, int ROW = 0>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_f32_kernel(const WgradParams p) {
    // AT = 3 (NS = 3): both operands are pre-split fp32 tensors (three bf16 planes each, dbn_split3): as AT = 1, three times.
    static_assert(AT == 0 || (AT == 1 && NS == 1) || (AT == 3 && NS == 3), "storage type / matrix math combination");
    static_assert(!ROW || AT == 0, "scalar-offset addressing is built for fp32 storage");
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
    // (DBN_DBG bit 8: every split reads the FIRST pixel range — the same work with all gathers served from L2)
    const int pbeg = (DBN_DBG & 8) ? 0 : split * p.pchunk;
    const int pend = min(p.P, pbeg + p.pchunk);
    const int KT = (pend - pbeg + 15) / 16;

    const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sm), 0, p.sm_bytes, 0x00020000);
    // ROW: taps reach `pad` rows / columns before a pixel: the per-thread offsets are made non-negative by moving the base back
    const unsigned row_shift = ROW ? (unsigned)((p.pad * p.W + p.pad) * p.Cb) * ES : 0u;
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.big) - row_shift), 0, p.big_bytes + row_shift, 0x00020000);

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
    // (DBN_DBG bit 16: every tap of a row reads the centre tap's pixel — same instructions, a third of the distinct lines)
    const int tr = tap / p.S - p.pad, ts = (DBN_DBG & 16) ? 0 : tap % p.S - p.pad;
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
    // ROW (round 3): the scalar-offset addressing frees the registers of the address walk — two sets fit 128 x 128 at 149 registers
    // (three waves per SIMD): 0.80 -> 0.84 at 80 x 80, no change at 40 x 40; three / four sets on the 64-row tiles +0..2 %: not taken
    constexpr int D = AT == 3 ? 1 : NS == 0 ? ((BM == 64 || ROW) ? 2 : 1) : (AT == 0 ? 3 : 4);
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
    // ROW: loop-invariant offsets; input row / columns of this thread's 4 pixels relative to the block's first input pixel;
    // scalar position of the block
    unsigned vcon[4] = {OOB_OFFSET, OOB_OFFSET, OOB_OFFSET, OOB_OFFSET};
    int ccol[4] = {0, 0, 0, 0};
    int crow = 0;
    int s_n = 0, s_oh = 0, s_ow = 0;
    unsigned soff = 0;
    const int TW = ROW ? p.row_tw : 16, TH = 16 / TW;
    if constexpr (ROW) {
        const bool live = KT > 0;
        const int qy = (4 * s_g) / TW;  // (TW >= 4: the 4 pixels of a thread share a row of the block)
        crow = qy * p.stride + tr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qx = (4 * s_g + i) & (TW - 1);
            if (is_a) {
                vcon[i] = live ? (unsigned)((qy * p.Wo + qx) * p.O + o0 + 4 * s_c) * ES : OOB_OFFSET;
                woff[i] = vcon[i];
            } else if (is_b) {
                ccol[i] = qx * p.stride + ts;
                vcon[i] = (live && j_ok) ? row_shift + (unsigned)((crow * p.W + ccol[i]) * p.Cb + ci) * ES : OOB_OFFSET;
            }
        }
        const int bw = p.Wo / TW, bh = p.Ho / TH;
        const int t0 = __builtin_amdgcn_readfirstlane(pbeg >> 4);
        const int r0 = t0 / bw;
        s_ow = (t0 - r0 * bw) * TW;
        s_n = r0 / bh;
        s_oh = (r0 - s_n * bh) * TH;
    }
    auto offsets = [&](int kt) {
        if (DBN_DBG & 2) return;
        if constexpr (ROW) {
            if (is_a) {
                soff = (unsigned)(((s_n * p.Ho + s_oh) * p.Wo + s_ow) * p.O) * ES;
            } else if (is_b) {
                const int ihb = s_oh * p.stride, iwb = s_ow * p.stride;
                soff = (unsigned)(((s_n * p.H + ihb) * p.W + iwb) * p.Cb) * ES;
                const bool rok = (unsigned)(ihb + crow) < (unsigned)p.H;
#pragma unroll
                for (int i = 0; i < 4; ++i) woff[i] = (rok && (unsigned)(iwb + ccol[i]) < (unsigned)p.W) ? vcon[i] : OOB_OFFSET;
            }
            if (kt + 1 < KT) {  // (scalar) next block of the split; the last one is re-read by the prefetch past the end
                s_ow += TW;
                if (s_ow == p.Wo) {
                    s_ow = 0;
                    s_oh += TH;
                    if (s_oh == p.Ho) {
                        s_oh = 0;
                        ++s_n;
                    }
                }
            }
            return;
        }
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
        if constexpr (ROW) {
            const __amdgpu_buffer_rsrc_t rs = is_a ? rs_sm : rs_big;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!(DBN_DBG & 1)) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    rr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)woff[i], (int)soff, 0));
                }
            return;
        }
        if constexpr (D > 1) {
            // ONE code path for both roles (descriptor and plane distance picked by the wave-uniform role; threads without a role
            // load from out-of-range offsets): with the loads inside role branches the compiler's wait counts at the merge point
            // fell back to vmcnt(0) and drained every set each k-step
            const __amdgpu_buffer_rsrc_t rs = is_a ? rs_sm : rs_big;
            if constexpr (AT == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(DBN_DBG & 1)) rr[i] = buffer_load_f32x4(rs, woff[i]);
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
        if ((is_a || is_b) && !(DBN_DBG & 4)) {
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
    constexpr bool AHEAD = NS > 0;  // (measured for ROW && NS == 0 too: 3-12 % slower than offsets right before the loads)
    if constexpr (D == 1) {
    if (KT > 0) {
        offsets(0);
        issue_loads(C0{});
        if (AHEAD) offsets(1);
        stage(0, C0{});
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        if (more) {
            if (!AHEAD) offsets(kt + 1);
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
        if (AHEAD) offsets(kt + 2);  // independent of the MFMAs above: overlaps their execution
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
                            // (scheduling barriers loads | MFMAs | staging measured: 0.71 -> 0.66 of peak on 64 x 192 — the
                            // compiler's interleaving of the staging moves with the MFMAs is the better schedule)
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

#if DBN_HAS_EXPERIMENTS
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
#endif  // DBN_HAS_EXPERIMENTS

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

// launch the register-transposing kernel for tile (bm, bn)
template <int NS, int AT, int ROW>
int launch_wgrad_tiles_row(const WgradParams& p, int bm, int bn, dim3 grid, hipStream_t st) {
    if (bn == 192)
        hipLaunchKernelGGL((wgrad_f32_kernel<64, 192, 2, 2, NS, AT, ROW>), grid, dim3(256), 0, st, p);
    else if (bm == 128 && bn == 128)
        hipLaunchKernelGGL((wgrad_f32_kernel<128, 128, 2, 2, NS, AT, ROW>), grid, dim3(256), 0, st, p);
    else if (bn == 128)
        hipLaunchKernelGGL((wgrad_f32_kernel<64, 128, 2, 2, NS, AT, ROW>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((wgrad_f32_kernel<64, 64, 2, 2, NS, AT, ROW>), grid, dim3(256), 0, st, p);
    return dbn_status();
}
// (scalar-offset addressing over TW x 16/TW pixel blocks: fp32 storage, p.row_tw chosen by wgrad_row_tw(), whole-k-tile splits)
template <int NS, int AT>
int launch_wgrad_tiles(const WgradParams& p, int bm, int bn, dim3 grid, hipStream_t st) {
    if constexpr (AT == 0)
        if (p.row_tw && p.pchunk % 16 == 0) return launch_wgrad_tiles_row<NS, AT, 1>(p, bm, bn, grid, st);
    return launch_wgrad_tiles_row<NS, AT, 0>(p, bm, bn, grid, st);
}

}  // namespace

// launchers of the kernel translation units (kind: 0 register-transposing, 1 wgrad_tr_kernel, 2 wgrad_patch_kernel, 3 wgrad_dma_kernel)
int dbn_launch_wgrad_f32(const WgradParams& p, int kind, int bm, int bn, dim3 grid, hipStream_t st);                 // wgrad_f32.hip
int dbn_launch_wgrad_b16(const WgradParams& p, int kind, int ns, int at, int bm, int bn, dim3 grid, hipStream_t st);  // wgrad_b16.hip
