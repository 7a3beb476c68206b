// 3x3 / stride-1 / pad-1 forward convolution in exact-fp32 arithmetic through the Winograd transform F(2x2, 3x3):
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2 x 2 output tile, 4 x 4 input window d, 3 x 3 filter g
// 16 element-wise "points" per tile instead of 36 multiply-adds per output pair: 2.25x fewer MFMA FLOPs than the direct
// convolution (the convolutions of /root/reference/src/modules/resnet.py:70-91 (BasicBlock), segmentation_body.py:55-61 (FPN smooth
// convs) and segmentation_head.py:24-25,64-68 (the head's 256 -> 64 convs) are all 3x3 / stride 1 / pad 1).
//
// Mapping onto v_mfma_f32_32x32x2_f32: a workgroup owns an 8 x 16 patch of output pixels of one image = 4 x 8 = 32 tiles, and 64
// output channels.  For point (i, j) the products are a GEMM  M_ij[32 tiles][64 cout] = V_ij[32 tiles][Cin] x U_ij[Cin][64 cout]:
// the 32 tiles are exactly one MFMA row block.  Wave i of the workgroup owns the four points (i, 0..3): 4 x 2 accumulator blocks
// (128 AGPRs).  Per 16-channel block the 10 x 18 input patch is brought to LDS once (the pixel-patch staging of igemm_kernel.h),
// every lane forms its V fragments on the fly — V_ij = sum of FOUR patch pixels with signs (B^T has two non-zeros per row: four
// ds_read_b128 + 12 vector instructions per fragment) — and the pre-transformed weights U = G g G^T come straight from a packed
// panel into registers (dbn_winograd_pack).  The epilogue applies A^T . A: along j inside each wave (registers), along i across the
// four waves (through LDS), adds the bias, optionally produces the train-mode BatchNorm tile statistics (same partial-row format as
// the implicit-GEMM kernels: bn_finalize_tiles_kernel folds them), and stores with raw buffer stores.
// Arithmetic is fp32 throughout (no reduced precision); the result differs from the direct convolution by fp32 rounding of a
// different summation (max relative error ~1e-6 of the output scale), not bit for bit.
#include "igemm_common.h"
#include <algorithm>
#include <mutex>
#include <unordered_map>
#ifndef DBN_WINO_BATCH
#define DBN_WINO_BATCH 0  // 1: each point's vector instructions fenced into one batch in front of its MFMAs (measured: 64->64 189 -> 191 us, 256->64 575 -> 593, the LIN form 173 -> 193 with two spills: the LDS latency in front of the batch is then exposed)
#endif
#ifndef DBN_WINO_EXP
#define DBN_WINO_EXP 0
#endif
#ifndef DBN_WINO_PAIR12
#define DBN_WINO_PAIR12 0  // (measured neutral: 64->64 176.7 vs 176.0 us, 256->64 549 vs 550, step 728.7 / 729.0 vs 728.3 / 727.2 images/s interleaved: off) round 5: points 1 and 2 (both V_1 = r_1 + r_2 and V_2 = r_2 - r_1 come from the same two row combinations) as ONE phase: 32 MFMAs over FOUR accumulator chains instead of twice 16 over two (profiles/r03_mfma_peak_probe.txt: 0.93 -> 0.99 of the pipe)
#endif
#ifndef DBN_WINO_ROWCOMB
#define DBN_WINO_ROWCOMB 1  // round 5: V through the shared row combinations r_c = d[a1][c] + sa * d[a2][c], formed lazily (see below)
#endif
// timing experiments (wrong results by construction; tools/winograd_probe.py with DBN_LIB_PATH): 1 weight fragments of the first channel
// block only, 2 no LDS reads / transform arithmetic in the loop, 3 no exchange / statistics in the epilogue, 5 no patch store / barrier
// in the loop, 6 = 3 + 5, 7 = 2 + 3 + 5 (what is left: prologue, MFMAs, weight loads, plain stores), 8 = 7 + 1 (... without the weight loads)
#define DBN_WX_NOXFORM (DBN_WINO_EXP == 2 || DBN_WINO_EXP == 7 || DBN_WINO_EXP == 8)
#define DBN_WX_NOEXCH (DBN_WINO_EXP == 3 || DBN_WINO_EXP == 6 || DBN_WINO_EXP == 7 || DBN_WINO_EXP == 8)
#define DBN_WX_NOBAR (DBN_WINO_EXP == 5 || DBN_WINO_EXP == 6 || DBN_WINO_EXP == 7 || DBN_WINO_EXP == 8)

namespace {

constexpr int W_PPX = 180, W_PROW = 18;  // 10 x 18 patch pixels

// LIN = false: a workgroup's 32 tiles are the 4 x 8 tiles of an 8 x 16-pixel patch (large maps).  LIN = true (maps up to ~50 pixels wide:
// layer3 / layer4 of the 640^2 benchmark, 40 x 40 and 20 x 20): the 32 tiles are CONSECUTIVE tiles of one image in row-major order over
// its ceil(H/2) x ceil(W/2) tile grid — no ragged patches (a 20 x 20 map is 52 % of its 8 x 16 patches, but 100 / 128 of its tile
// groups) — and the LDS patch is the band of pixel rows those tiles touch, full width (+ the one-pixel halo).
constexpr int W_LIN_PPX = 448;  // most patch pixels of the LIN form (dbn_winograd_eligible checks the map against it)
// PERSIST: the item loop of the persistent forms (round 5; measured neutral, see DESIGN: off by default).  CBS: channel blocks staged per
// barrier (2: the patch form on Cs % 32 == 0 — half the barriers, twice the patch registers and LDS; the LIN band has no room for it).
template <bool LIN, bool PERSIST, int CBS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void winograd_f32_kernel(const IgemmParams p) {
    static_assert(CBS == 1 || (CBS == 2 && !LIN), "two blocks per barrier: patch form only");
    // LDS: loop = two patch buffers [4 chunks][patch pixels] f32x4 (23 KB; LIN: up to 57 KB); epilogue = the cross-wave exchange
    // [4 waves][2 dx][2 b][16][64] floats (64 KB) + the statistics scratch
    constexpr int PPX_MAX = LIN ? W_LIN_PPX : W_PPX;
    constexpr int P_PATCH = 4 * PPX_MAX;
    constexpr int X_FLOATS = 4 * 2 * 2 * 16 * 64;
    constexpr int P_STAGE = CBS * P_PATCH;  // one stage = the patches of CBS consecutive channel blocks; two stages
    static_assert(2 * P_STAGE <= X_FLOATS / 4, "the patch buffers live inside the exchange region");
    // + the statistics / sums scratch [<= 3][4][64] floats, a flag, 8 counts | the apply-on-load coefficients of <= 512 channels | the next item
#ifndef DBN_WINO_LDSPAD
#define DBN_WINO_LDSPAD 0  // (timing experiment: extra LDS in 16-byte units — 1024 pushes a workgroup over half a CU's LDS: one resident workgroup per CU)
#endif
    __shared__ f32x4 smem[X_FLOATS / 4 + (3 * 4 * 64) / 4 + 4 + 2 * 128 + 1 + DBN_WINO_LDSPAD];
    const int tid = threadIdx.x;
    // ---- PERSISTENT workgroups (round 5): the grid is at most two workgroups per CU; each pulls (patch, 64-channel tile) items from
    // per-XCD counters until none is left (dbn_xcd_remap's layout: XCD x owns a contiguous run of items, so the halo rows of neighbouring
    // patches and the weight panels of one channel tile stay in that XCD's L2).  What it
    // buys over one workgroup per item (3200 launches of ~23 us at 64 -> 64): no workgroup launch / register and LDS allocation per item,
    // the per-thread invariants (descriptors, the wave's transform rows, the apply-on-load coefficients) set up once, and the pull is
    // greedy, so a launch that shares the chip with the weight-gradient stream — whose workgroups take whole CUs — still balances.
    // p.work: [8 XCD counters][1 exit counter] ints, zero on entry, left zero (the last workgroup out clears them).
    int* const s_next = reinterpret_cast<int*>(smem + X_FLOATS / 4 + (3 * 4 * 64) / 4 + 4 + 2 * 128);
    const int total = p.work_items;
    const int xcd = blockIdx.x & 7, xq = total >> 3, xr = total & 7;
    auto xsize = [&](int x) { return xq + (x < xr ? 1 : 0); };
    auto xbase = [&](int x) { return x < xr ? x * (xq + 1) : xr * (xq + 1) + (x - xr) * xq; };
    // (thread 0) the next item of this XCD's run, -1 when it is used up.  No stealing from other XCDs: alone on the chip the eight runs
    // finish together, and a failed pull costs a full atomic round trip (~2 us under load) — eight of them per workgroup at the end of a
    // launch were +30 us on a 180 us launch in the first build
    auto resolve = [&](int idx) { return idx < xsize(xcd) ? xbase(xcd) + idx : -1; };
    auto pull = [&]() { return __hip_atomic_fetch_add(p.work + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    int item;
    if (PERSIST && p.work) {
        if (tid == 0) *s_next = resolve(pull());
        __syncthreads();
        item = *s_next;
    } else {
        // p.work == NULL: one workgroup per item (gridDim.x == total), or — gridDim.x < total, dbn_set_winograd_persistent(2) — a STATIC
        // schedule: workgroup b takes positions b, b + gridDim.x, ... of the XCD-contiguous order (no atomics; for launches that own the chip)
        item = dbn_xcd_remap(blockIdx.x, total);
    }
    int item_pos = blockIdx.x;
    // ---- phase stagger of the two workgroups that share a CU (persistent forms).  All workgroups of a launch start together and every item
    // takes the same time, so the two residents of a CU stay in LOCKSTEP: both in their prologue (the matrix pipe idles ~2 us behind the
    // first loads), both in the loop (sharing the pipe), both in the epilogue (idle again) — a round is prologue + 2 x MFMA + epilogue, not
    // max(...).  The timing builds of round 5 say so: with exchange, statistics, staging, barriers and the transform all compiled out the
    // 64 -> 64 launch still took 151 us against 85 us of matrix time, and neither pulled nor static persistence moved it (176 / 182 / 183 us).
    // The workgroup in the SECOND wave slot of its SIMDs therefore sleeps p.stagger_units x 1024 clocks once, at launch (about the
    // matrix time of one item): from then on one resident's epilogue / prologue runs beside the other's loop.
    if (PERSIST && p.stagger_units > 0 && total > (int)gridDim.x) {
        int* const s_slot = s_next + 1;
        if (tid == 0) *s_slot = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u);  // HW_REG_HW_ID.WAVE_ID of wave 0
        __syncthreads();
        if (*s_slot) {
            for (int u = 0; u < p.stagger_units; ++u) {
#pragma unroll
                for (int q = 0; q < 2; ++q) __builtin_amdgcn_s_sleep(8);  // 2 x 8 x 64 clocks
            }
        }
    }
    // apply-on-load (IgemmParams::in_scale): this thread's pieces are chunk tid & 3 of every pixel = channels 16 cb + 4 (tid & 3) .. + 3; the
    // coefficients (Cs <= 512) sit in LDS behind the epilogue's regions for the whole life of the workgroup
    const bool act = p.in_scale != nullptr;
    f32x4* const ACT = smem + X_FLOATS / 4 + (3 * 4 * 64) / 4 + 4;  // [Cs / 4] scale, [Cs / 4] shift
    if (act) {
        for (int i = tid; i < (p.Cs >> 2); i += 256) {
            ACT[i] = reinterpret_cast<const f32x4*>(p.in_scale)[i];
            ACT[(p.Cs >> 2) + i] = reinterpret_cast<const f32x4*>(p.in_shift)[i];
        }
        __syncthreads();
    }
    // (this wave's row of the input transform — B^T row i = wave has two non-zeros: V = d[a1] + sa * d[a2];  i = 0: d0 - d2;  1: d1 + d2;
    // 2: d2 - d1;  3: d1 - d3 — is re-derived per item below)
    typedef const __attribute__((address_space(4))) IgemmParams* KernArgs;
    const unsigned long long kargs = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();  // (p is the kernel's only argument)
    const int tid_wg = tid;
    while (item >= 0) {
        // Per-item state is re-derived from laundered copies of the thread index and of the argument block: left visible as loop
        // invariants, the address arithmetic of the whole body (~100 values per lane) and every kernel argument were hoisted out of the
        // item loop, stayed live across it and spilled (342 vector + 111 scalar registers in the first build).
        int tid_l = tid_wg;
        unsigned long long ka_l = kargs;
        if constexpr (PERSIST) asm volatile("" : "+v"(tid_l), "+s"(ka_l));
        const int tid = tid_l, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
        const auto& p = *reinterpret_cast<KernArgs>(ka_l);
        const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1), a2 = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
        const float sa = wave == 1 ? 1.f : -1.f;
        const int ntn = p.Cd >> 6, ncb = p.Cs >> 4;
        const bool act = p.in_scale != nullptr;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcW =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (unsigned)((long)p.Cs * 16 * p.Cd * 4), 0x00020000);
        const int tile = item;
        auto mdiv = [](int x, unsigned long long m) { return (int)(((unsigned long long)(unsigned)x * m) >> 40); };  // x / d with the host's magic(d)
        const int mt = mdiv(tile, p.wino_m_ntn), nt = tile - mt * ntn, n0 = nt * 64;
        // patch geometry: image pn; the patch's first pixel row / column in the image (hs0, ws0: the halo starts one pixel before the
        // first output pixel), its row pitch `prow` and pixel count `ppx` (= the LDS stride between the four 16-byte chunks of a block)
        int pn, ph0 = 0, pw0 = 0, hs0, ws0, prow, ppx, t0 = 0, TWl = 1, ntiles = 0;
        if constexpr (LIN) {
            TWl = (p.Wdf + 1) >> 1;
            ntiles = ((p.Hdf + 1) >> 1) * TWl;
            const int groups = p.wino_tpi;  // (= (ntiles + 31) >> 5)
            pn = mdiv(mt, p.wino_m_tpi);
            t0 = (mt - pn * groups) << 5;
            const int tr0 = t0 / TWl, tr1 = min(t0 + 31, ntiles - 1) / TWl;  // first / last tile row of this group
            hs0 = 2 * tr0 - 1;
            ws0 = -1;
            prow = 2 * TWl + 2;
            ppx = (2 * (tr1 - tr0 + 1) + 2) * prow;
        } else {
            const int tw = p.wino_tw, tpi = p.wino_tpi;  // ((W + 15) >> 4 patches per row, ((H + 7) >> 3) * tw per image; ragged right / bottom patches: pixels past the map are masked)
            pn = mdiv(mt, p.wino_m_tpi);
            const int t_ = mt - pn * tpi, ty_ = mdiv(t_, p.wino_m_tw);
            ph0 = ty_ * 8;
            pw0 = (t_ - ty_ * tw) * 16;
            hs0 = ph0 - 1;
            ws0 = pw0 - 1;
            prow = W_PROW;
            ppx = W_PPX;
        }

        // ---- patch staging: patch pixels x 4 chunks of 16 bytes per channel block, PL pieces per thread
        constexpr int PL = (PPX_MAX * 4 + 255) / 256;
        unsigned poff[PL];
        int pslot[PL];
        const float rprow = 1.0f / (float)prow;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int idx = tid + j * 256;
            const bool on = idx < ppx * 4;
            const int chunk = idx & 3, pix = on ? idx >> 2 : 0;
            int py, px;
            if constexpr (LIN) divmod24(pix, prow, rprow, py, px);
            else { py = pix / W_PROW; px = pix - py * W_PROW; }
            const int hs = hs0 + py, ws = ws0 + px;
            const bool v = on && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
            poff[j] = v ? (unsigned)(((pn * p.Hs + hs) * p.Ws + ws) * p.Cs) * 4u + (unsigned)chunk * 16u : OOB_OFFSET;
            pslot[j] = on ? chunk * ppx + pix : -1;
        }
        // ONE register set whatever CBS: with two channel blocks per barrier the second block's patch is fetched once the first one's has
        // gone to LDS, half a stage later (two sets had the compiler spill the patch registers inside the loop: 43 dwords)
        f32x4 pr[PL];
        auto load_patch = [&](int st, int u) {  // stage st = channel blocks CBS st .. CBS st + CBS - 1; u: block within the stage
            const unsigned add = (unsigned)((st * CBS + u) * 64);
#pragma unroll
            for (int j = 0; j < PL; ++j) pr[j] = buffer_load_f32x4(rsrc, poff[j] == OOB_OFFSET ? OOB_OFFSET : poff[j] + add);
        };
        auto store_patch = [&](int buf, int st, int u) {
            f32x4* const P = smem + buf * P_STAGE + u * P_PATCH;
            if (act) {  // (pixels outside the map are the conv's zero padding of the ACTIVATION: they stay zero)
                const int cb = st * CBS + u;
                const f32x4 asc = ACT[cb * 4 + (tid & 3)], ash = ACT[(p.Cs >> 2) + cb * 4 + (tid & 3)];
#pragma unroll
                for (int j = 0; j < PL; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr[j][e] = poff[j] == OOB_OFFSET ? 0.f : dbn_affine_relu(pr[j][e], asc[e], ash[e]);
            }
#pragma unroll
            for (int j = 0; j < PL; ++j)
                if (pslot[j] >= 0) P[pslot[j]] = pr[j];
        };

        // MFMA row li = tile (ty, tx) — patch form: (li >> 3, li & 7); LIN: tile t0 + li of the image's grid (past the last tile: tile 0 of
        // the band, masked in the epilogue) — whose 4 x 4 input window starts at patch pixel (2 ty, 2 tx); chunk 2*s2 + lh
        int lty = li >> 3, ltx = li & 7;
        if constexpr (LIN) {
            const int t = t0 + li < ntiles ? t0 + li : t0;
            divmod24(t, TWl, 1.0f / (float)TWl, lty, ltx);
            lty -= t0 / TWl;
        }
        const int vbase = 2 * lty * prow + 2 * ltx + lh * ppx;
        const int row1 = vbase + a1 * prow, row2 = vbase + a2 * prow;
        const int ppx2 = 2 * ppx;  // chunk 2*s2 + lh: s2 = 1 lies two chunk planes further

        // ---- weight fragments: panel [cb][16 points][4 chunks][Cd][4]; lane (li, lh): chunk 2*s2 + lh, column n0 + 32 b + li
        unsigned wvo[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < 2; ++b) wvo[s2][b] = (unsigned)((2 * s2 + lh) * p.Cd + n0 + b * 32 + li) * 16u;
        const unsigned point_bytes = (unsigned)(4 * p.Cd) * 16u;  // one point of one channel block
        int w_next = 0;  // next (channel block, point) to fetch: g = cb * 4 + j; clamped at the end (the surplus fetch is never used)
        const int w_last = ncb * 4 - 1;  // (the two surplus fetches at the end re-read the last one and are never used)
        f32x4 rw[4][2][2];  // one fragment set per point of a block, fetched TWO points ahead (one point = 16 MFMAs ~ 0.4-1.2 us: an L2 hit under load takes about as long)
        auto issue_w = [&](auto SET) {
            constexpr int st_ = decltype(SET)::value;
            const int g = min(w_next, w_last);
            ++w_next;
#if DBN_WINO_EXP == 1 || DBN_WINO_EXP == 8  // (timing experiment, wrong results: the weight fragments of the first channel block only — no L2 weight traffic; 8 = 7 + this)
            if (w_next > 4) return;
#endif
            const unsigned so = (unsigned)((g >> 2) * 16 + 4 * wave + (g & 3)) * point_bytes;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)wvo[s2][b], (int)so, 0);
                    rw[st_][s2][b] = __builtin_bit_cast(f32x4, v_);
                }
        };

        // (not zeroed: the first MFMA of every accumulator — channel block 0 — takes the constant 0 as its addend.  128 v_mov per wave
        // otherwise, and a vector instruction costs fp32-MFMA time on this chip whichever wave issues it: DESIGN 7.12)
        f32x16 acc[4][2];
        if constexpr (!LIN && PERSIST) {
            // (persistent loop: "not yet written" must not read as "whatever the previous item left" — the compiler then carries all 128
            // accumulator registers around the item loop, through the epilogue, and spills the epilogue's loads instead: an empty asm
            // DEFINES them here, at no instruction)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) asm volatile("" : "=v"(acc[j][b]));
        }
        if constexpr (LIN) {  // (the LIN form has no registers to spare for the peeled first block: 68 spills — it zeroes its accumulators)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;
        }

        DBN_TRACE_MARK(0);
#if DBN_TRACE
        if (p.trace && threadIdx.x == 0)  // HW_REG_HW_ID (wave / SIMD / CU / SH / SE) | HW_REG_XCC_ID << 32: which CU this workgroup ran on
            p.trace[(long)blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 32);
#endif
#if DBN_TRACE
        unsigned long long tr_bar = 0, tr_t = 0;
#endif
        load_patch(0, 0);
        issue_w(std::integral_constant<int, 0>{});
        issue_w(std::integral_constant<int, 1>{});
        if constexpr (DBN_WINO_ROWCOMB && DBN_WINO_PAIR12 && !(DBN_WINO_EXP == 2 || DBN_WINO_EXP == 7 || DBN_WINO_EXP == 8) && !LIN)
            issue_w(std::integral_constant<int, 2>{});  // (paired order: points 1 and 2 start together)
        store_patch(0, 0, 0);
        if constexpr (CBS == 2) {
            load_patch(0, 1);
            store_patch(0, 0, 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DBN_TRACE_MARK(1);
#ifndef DBN_WINO_PRIO
#define DBN_WINO_PRIO 0  // wave priority experiment: 1 = the main loop above the set-up / epilogue phases of the co-resident workgroup, 2 = below
#endif
        if (DBN_WINO_PRIO == 1) __builtin_amdgcn_s_setprio(3);
        if (DBN_WINO_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        const int nst = ncb / CBS;  // stages (CBS == 2: the launcher checked that ncb is even)
        for (int st = 0; st < nst; ++st) {
            load_patch(st + 1, 0);  // (past the last block: out-of-range offsets, zeros, never stored)
#pragma unroll
            for (int u = 0; u < CBS; ++u) {
            const int cb = st * CBS + u;
            if (CBS == 2 && u == 1 && !DBN_WX_NOBAR) {
                // the next stage's FIRST block goes to LDS already (its buffer was last read in stage st - 1, and every wave has passed the
                // barrier that ended it), and the second block's fetch takes over the registers
                if (st + 1 < nst) store_patch((st + 1) & 1, st + 1, 0);
                load_patch(st + 1, 1);
            }
            const f32x4* const P = smem + (st & 1) * P_STAGE + u * P_PATCH;
            // Per point (i, j): V = (d[a1][b1] +- d[a1][b2]) + sa * (d[a2][b1] +- d[a2][b2]) — four LDS reads, twelve vector instructions per
            // chunk.  (Measured and not kept: forming all four points' V at the start of a block from the shared row combination — 16 reads
            // + 44 vector instructions per block instead of 32 + 96 — 3-8 % SLOWER: one long vector phase per block overlaps the other
            // resident wave's MFMAs worse than four short ones; issuing the next point's LDS reads ahead of the current point's MFMAs:
            // neutral, and its 32 registers are better spent on the weight fragments' prefetch distance.)
            constexpr bool RC = DBN_WINO_ROWCOMB && !DBN_WX_NOXFORM && !LIN;  // (the LIN form has no registers for it: 51 spilled dwords)
            constexpr bool PAIRED = RC && DBN_WINO_PAIR12;
            // The wave's four points (i, 0..3) take the SAME two patch rows (a1, a2) and differ in the column pair only: with
            //   r_c = d[a1][c] + sa * d[a2][c]   (c = 0..3),   V_0 = r_0 - r_2,  V_1 = r_1 + r_2,  V_2 = r_2 - r_1,  V_3 = r_1 - r_3
            // a channel block needs 16 LDS reads and 64 vector instructions instead of 32 and 96.  Round 4 had tried this with all four V formed
            // at the start of the block (one long vector phase: 3-8 % slower); here each r_c is formed in front of the first point that needs it
            // (point 0: r_0, r_2 — 8 reads, 24 vector instructions as before; point 1: r_1 — 4 + 16; point 2: nothing to read, 8; point 3: r_3 —
            // 4 + 16), so the phases only get shorter.  Rounding: fma(sa, d2, d1) then +-, where the old form took +- first: equally exact
            // (sa = +-1), a different last bit.
            f32x4 rc[RC ? 4 : 1][2];
            auto form_r = [&](auto C) {
                constexpr int c = RC ? decltype(C)::value : 0;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f32x4 x1 = P[s2 * ppx2 + row1 + c], x2 = P[s2 * ppx2 + row2 + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) rc[c][s2][e] = fmaf(sa, x2[e], x1[e]);  // (sa = +-1: exact)
                }
            };
            auto point = [&](auto J, auto FIRST) {
                constexpr int j = decltype(J)::value;
                constexpr bool first = decltype(FIRST)::value;  // channel block 0: the accumulators start here
                if constexpr (!PAIRED) {
                    issue_w(std::integral_constant<int, (j + 2) & 3>{});  // the weight fragments of the point after the next (set = point index)
                } else if constexpr (j == 3) {  // (paired order, see pair12: sets 1 and 2 of the NEXT block, two phases ahead of its pair)
                    issue_w(std::integral_constant<int, 1>{});
                    issue_w(std::integral_constant<int, 2>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                // B^T column j: 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
                constexpr int b1 = j == 0 ? 0 : (j == 2 ? 2 : 1), b2 = j == 0 ? 2 : (j == 1 ? 2 : (j == 2 ? 1 : 3));
                constexpr bool plus = j == 1;
                f32x4 v[2];
#if DBN_WX_NOXFORM  // (timing experiment, wrong results: no LDS reads / transform arithmetic in the loop)
                // (round 5: NOT the patch registers — `pr` is rewritten by load_patch at the top of every channel block, so MFMAs fed from it
                // waited a global-load latency per block and builds 2 / 7 / 8 measured that wait, not the skeleton.  An opaque per-lane value.)
                v[0] = f32x4{(float)lane, 1.f, 2.f, 3.f};
                v[1] = f32x4{4.f, (float)li, 6.f, 7.f};
                asm volatile("" : "+v"(v[0]), "+v"(v[1]));
#else
                if constexpr (RC) {
                    if constexpr (j == 0) {
                        form_r(std::integral_constant<int, 0>{});
                        form_r(std::integral_constant<int, 2>{});
                    } else if constexpr (j == 1) {
                        form_r(std::integral_constant<int, 1>{});
                    } else if constexpr (j == 3) {
                        form_r(std::integral_constant<int, 3>{});
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) v[s2] = plus ? rc[RC ? b1 : 0][s2] + rc[RC ? b2 : 0][s2] : rc[RC ? b1 : 0][s2] - rc[RC ? b2 : 0][s2];
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const f32x4 x11 = P[s2 * ppx2 + row1 + b1], x12 = P[s2 * ppx2 + row1 + b2];
                        const f32x4 x21 = P[s2 * ppx2 + row2 + b1], x22 = P[s2 * ppx2 + row2 + b2];
                        const f32x4 t1 = plus ? x11 + x12 : x11 - x12, t2 = plus ? x21 + x22 : x21 - x22;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[s2][e] = fmaf(sa, t2[e], t1[e]);  // (sa = +-1: exact)
                    }
                }
#endif
                // (DBN_WINO_BATCH: the point's 24 vector instructions fenced into ONE batch in front of its 16 MFMAs — a vector instruction
                // between two MFMAs costs ~13 clocks of the fp32 matrix pipe, in a batch ~5 — measured slower here, see the define)
#if DBN_WINO_BATCH
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                        {
                            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s2][e], rw[j][s2][b][e], (first && s2 == 0 && e == 0) ? zero16 : acc[j][b], 0, 0, 0);
                        }
#if DBN_WINO_BATCH
                __builtin_amdgcn_sched_barrier(0);
#endif
            };
            auto pair12 = [&](auto FIRST) {
                constexpr bool first = decltype(FIRST)::value;
                // weight prefetch in the paired order (every set at least two point-times ahead of its use, none rewritten while in use):
                // here point 3 of this block and point 0 of the next; point 3 then fetches sets 1 and 2 of the next block; the prologue 0, 1, 2
                issue_w(std::integral_constant<int, 3>{});
                issue_w(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                form_r(std::integral_constant<int, 1>{});
                f32x4 v1[2], v2[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    v1[s2] = rc[RC ? 1 : 0][s2] + rc[RC ? 2 : 0][s2];
                    v2[s2] = rc[RC ? 2 : 0][s2] - rc[RC ? 1 : 0][s2];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s2][e], rw[1][s2][b][e], (first && s2 == 0 && e == 0) ? zero16 : acc[1][b], 0, 0, 0);
                            acc[2][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s2][e], rw[2][s2][b][e], (first && s2 == 0 && e == 0) ? zero16 : acc[2][b], 0, 0, 0);
                        }
            };
            if constexpr (PAIRED) {
                if (cb == 0) {
                    point(std::integral_constant<int, 0>{}, std::true_type{});
                    pair12(std::true_type{});
                    point(std::integral_constant<int, 3>{}, std::true_type{});
                } else {
                    point(std::integral_constant<int, 0>{}, std::false_type{});
                    pair12(std::false_type{});
                    point(std::integral_constant<int, 3>{}, std::false_type{});
                }
            } else if (!LIN && cb == 0) {
                point(std::integral_constant<int, 0>{}, std::true_type{});
                point(std::integral_constant<int, 1>{}, std::true_type{});
                point(std::integral_constant<int, 2>{}, std::true_type{});
                point(std::integral_constant<int, 3>{}, std::true_type{});
            } else {
                point(std::integral_constant<int, 0>{}, std::false_type{});
                point(std::integral_constant<int, 1>{}, std::false_type{});
                point(std::integral_constant<int, 2>{}, std::false_type{});
                point(std::integral_constant<int, 3>{}, std::false_type{});
            }
            }  // (u)
            if (st + 1 < nst && !DBN_WX_NOBAR) {
                // the other buffer was last read in block cb - 1, and every wave has passed the barrier that ended it
#if DBN_TRACE
                tr_t = __builtin_amdgcn_s_memrealtime();
#endif
                store_patch((st + 1) & 1, st + 1, CBS - 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#if DBN_TRACE
                tr_bar += __builtin_amdgcn_s_memrealtime() - tr_t;
#endif
            }
        }
        DBN_TRACE_MARK(2);
        if (DBN_WINO_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (DBN_WINO_PRIO == 2) __builtin_amdgcn_s_setprio(3);
#if DBN_TRACE
        if (p.trace && threadIdx.x == 0) p.trace[(long)blockIdx.x * 8 + 5] = tr_bar;  // ticks spent from "MFMAs issued" to "past the barrier"
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave is done with the patches: the region becomes the exchange buffer
        // the pull for the NEXT item goes out here, so that its round trip runs under the epilogue (it is consumed at the item's end)
        int next_idx = 0;
        if (PERSIST && p.work && tid == 0) next_idx = pull();

#if DBN_WX_NOEXCH  // (timing experiment, wrong results: no exchange, no statistics — what would an epilogue that stays in registers cost?)
        {
            const unsigned pitch_ = (unsigned)p.Cd * 4u;
            const __amdgpu_buffer_rsrc_t rsrcD_ = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, (unsigned)((long)p.N * p.Hdf * p.Wdf * p.Cd * 4), 0x00020000);
            const unsigned base_ = (unsigned)((pn * p.Hdf + ph0 + (wave >> 1)) * p.Wdf + pw0 + (wave & 1) + 8 * lh) * pitch_ + (unsigned)(n0 + li) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float v_ = ((acc[0][b][r] + acc[1][b][r]) + acc[2][b][r]) - acc[3][b][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v_), rsrcD_, (int)(base_ + (unsigned)(2 * (r >> 2) * p.Wdf + 2 * (r & 3)) * pitch_) + b * 128, 0, 0);
                }
        }
#else
        // ---- output transform.  Along j (this wave holds M_i0 .. M_i3): T_i[dx] = A^T row dx: dx 0: M0 + M1 + M2;  dx 1: M1 - M2 - M3
        // Exchange layout: [(i*2 + dx)*2 + b][lane][4 groups of four rows], 16-byte accesses; the group index is XOR-swizzled with bits 1-2
        // of the lane so that the eight lanes of one LDS cycle (64 bytes apart) hit eight different 16-byte bank groups.  (First form:
        // [..][r][lane] with 64 four-byte writes and 96 four-byte reads per lane.)
        f32x4* const X = smem;
        const int xsw = (lane >> 1) & 3;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                f32x4 t0_, t1_;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rb + e;
                    const float m0_ = acc[0][b][r], m1_ = acc[1][b][r], m2_ = acc[2][b][r], m3_ = acc[3][b][r];
                    t0_[e] = (m0_ + m1_) + m2_;
                    t1_[e] = (m1_ - m2_) - m3_;
                }
                X[(((wave * 2 + 0) * 2 + b) * 64 + lane) * 4 + (rb ^ xsw)] = t0_;
                X[(((wave * 2 + 1) * 2 + b) * 64 + lane) * 4 + (rb ^ xsw)] = t1_;
            }
        __builtin_amdgcn_sched_barrier(0);  // (the loads below must not be scheduled above the writes that free the accumulators' registers)
        // (the accumulators are dead from here on: their registers take the epilogue's global loads, ALL issued before the exchange
        // barrier so that their latency hides behind it and the LDS reads below — in groups of four rows behind scheduling fences they
        // were four exposed round trips per tile: the head's accumulating data gradient with sums took 728 us against 558 us without)
        const int dy = wave >> 1, dx = wave & 1;
        // destination offsets: tile t = (r & 3) + 8 (r >> 2) + 4 lh -> pixel (ph0 + 2 (t >> 3) + dy, pw0 + 2 (t & 7) + dx), channel n0 + 32 b + li
        const unsigned dst_bytes = (unsigned)((long)p.N * p.Hdf * p.Wdf * p.Cd * 4);
        const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, dst_bytes, 0x00020000);
        const unsigned pitch = (unsigned)p.Cd * 4u;
        // t >> 3 = r >> 2 (rows of tiles), t & 7 = (r & 3) + 4 lh
        // byte offset of row r's pixel (tile (r & 3) + 8 (r >> 2) + 4 lh of the workgroup's 32), or OOB_OFFSET past the map: loads read 0,
        // stores are dropped, and bit r of `vmask` keeps the pixel out of the statistics
        unsigned vmask = 0;
        unsigned roff[16];
        bool allv = false;  // (uniform) every pixel of this workgroup's tiles lies inside the map: no per-pixel validity arithmetic below
        if constexpr (LIN) {
            const float rTW = 1.0f / (float)TWl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int ty, tx;
                divmod24(min(t, ntiles - 1), TWl, rTW, ty, tx);
                const int yy = 2 * ty + dy, xx = 2 * tx + dx;
                const bool ok = t < ntiles && yy < p.Hdf && xx < p.Wdf;
                vmask |= (unsigned)ok << r;
                roff[r] = ok ? (unsigned)((pn * p.Hdf + yy) * p.Wdf + xx) * pitch + (unsigned)(n0 + li) * 4u : OOB_OFFSET;
            }
        } else {
            // t >> 3 = r >> 2 (rows of tiles), t & 7 = (r & 3) + 4 lh
            const unsigned base = (unsigned)((pn * p.Hdf + ph0 + dy) * p.Wdf + pw0 + dx + 8 * lh) * pitch + (unsigned)(n0 + li) * 4u;
            allv = ph0 + 8 <= p.Hdf && pw0 + 16 <= p.Wdf;
            if (allv) {  // a whole patch (every patch of a map whose sides are multiples of 8 / 16): one add per row, scalar row offsets
                vmask = 0xFFFFu;
#pragma unroll
                for (int r = 0; r < 16; ++r) roff[r] = base + (unsigned)(2 * (r >> 2) * p.Wdf + 2 * (r & 3)) * pitch;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = (ph0 + 2 * (r >> 2) + dy < p.Hdf) && (pw0 + 2 * ((r & 3) + 4 * lh) + dx < p.Wdf);
                    vmask |= (unsigned)ok << r;
                    roff[r] = ok ? base + (unsigned)(2 * (r >> 2) * p.Wdf + 2 * (r & 3)) * pitch : OOB_OFFSET;
                }
            }
        }
        auto row_off = [&](int r) { return roff[r]; };
        auto ldf = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0)); };
        const bool sums = p.bnb_part != nullptr, zm = p.bnb_zmask != nullptr, two = p.bnb_y2 != nullptr;
        float oldv[2][16], yv[2][16], zv[2][16], y2v[2][16];
        // (conditionally loaded, conditionally used: inside the item loop "not loaded" must not read as "the previous item's value", or the
        // four arrays stay live around the whole loop — empty asm statements define them here, at no instruction; see acc above)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if constexpr (PERSIST) asm volatile("" : "=v"(oldv[b][r]), "=v"(yv[b][r]), "=v"(zv[b][r]), "=v"(y2v[b][r]));
        if (p.accumulate) {  // (inference epilogue: the addend is a residual input, IgemmParams::res, instead of dst itself)
            const __amdgpu_buffer_rsrc_t rsrcA = p.res ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, dst_bytes, 0x00020000) : rsrcD;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int b = 0; b < 2; ++b) oldv[b][r] = ldf(rsrcA, row_off(r) + b * 128);
        }
        if (sums) {
            const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bnb_y), 0, dst_bytes, 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int b = 0; b < 2; ++b) yv[b][r] = ldf(rsY, row_off(r) + b * 128);
            if (zm) {
                const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bnb_zmask), 0, dst_bytes, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b) zv[b][r] = ldf(rsZ, row_off(r) + b * 128);
            }
            if (two) {
                const __amdgpu_buffer_rsrc_t rsY2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bnb_y2), 0, dst_bytes, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b) y2v[b][r] = ldf(rsY2, row_off(r) + b * 128);
            }
        }
        __syncthreads();
        // along i, across the waves: wave w produces the output pixel (dy, dx) = (w >> 1, w & 1) of every tile:
        //   dy 0: T_0 + T_1 + T_2;  dy 1: T_1 - T_2 - T_3
        float y[2][16];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                auto T = [&](int i) { return X[(((i * 2 + dx) * 2 + b) * 64 + lane) * 4 + (rb ^ xsw)]; };
                const f32x4 t1 = T(1), t2 = T(2), t03 = T(dy == 0 ? 0 : 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[b][4 * rb + e] = dy == 0 ? (t03[e] + t1[e]) + t2[e] : (t1[e] - t2[e]) - t03[e];
            }
        }
        if (p.bias) {  // (the convs in front of a BatchNorm have none: 32 vector instructions saved)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float bias = p.bias[n0 + b * 32 + li];
#pragma unroll
                for (int r = 0; r < 16; ++r) y[b][r] += bias;
            }
        }
        if (p.accumulate) {  // (data gradients that add into an existing gradient: the sums below see the final values)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[b][r] += oldv[b][r];
        }
        if (p.relu) {  // inference epilogue: eval-mode BatchNorm folded into the panel (bias = its shift), ReLU here
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[b][r] = fmaxf(y[b][r], 0.f);
        }
        // ---- optional: the two per-channel sums of the BatchNorm backward that consumes dst (IgemmParams::bnb_*; igemm_kernel.h EPI = 1):
        //      g = dz * [mask > 0],  part[0][c][row] = sum g,  part[1][c][row] = sum g * (ybn - mean[c]) * rstd[c]  over this 128-pixel tile
        if (sums) {
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, s4[2] = {0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int c = n0 + b * 32 + li;
                const float mu = p.bnb_mean[c], rsd = p.bnb_rstd[c];
                const float nmr = -mu * rsd, nmr2 = two ? -p.bnb_mean2[c] * p.bnb_rstd2[c] : 0.f;  // xhat = fma(y, rstd, -mean * rstd): one instruction per pixel less
                const float msc = zm ? 0.f : p.bnb_msc[c], msh = zm ? 0.f : p.bnb_msh[c];
                const float rs2 = two ? p.bnb_rstd2[c] : 0.f;
                if (allv) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float m_ = zm ? zv[b][r] : dbn_affine(yv[b][r], msc, msh);
                        const float gq = m_ > 0.f ? y[b][r] : 0.f;
                        s1[b] += gq;
                        s2[b] = fmaf(gq, fmaf(yv[b][r], rsd, nmr), s2[b]);
                        if (two) s4[b] = fmaf(gq, fmaf(y2v[b][r], rs2, nmr2), s4[b]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float m_ = zm ? zv[b][r] : dbn_affine(yv[b][r], msc, msh);
                        const float gq = (m_ > 0.f && ((vmask >> r) & 1u)) ? y[b][r] : 0.f;
                        s1[b] += gq;
                        s2[b] = fmaf(gq, fmaf(yv[b][r], rsd, nmr), s2[b]);
                        if (two) s4[b] = fmaf(gq, fmaf(y2v[b][r], rs2, nmr2), s4[b]);
                    }
                }
            }
            float* const red = reinterpret_cast<float*>(smem) + X_FLOATS;  // [3][4 waves][64]
            __syncthreads();  // (the exchange buffer X has been read by everyone; red lies behind it, but keep the phases apart)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float t1 = s1[b] + __shfl_xor(s1[b], 32, 64), t2 = s2[b] + __shfl_xor(s2[b], 32, 64), t4 = s4[b] + __shfl_xor(s4[b], 32, 64);
                if (lh == 0) {
                    red[(0 * 4 + wave) * 64 + b * 32 + li] = t1;
                    red[(1 * 4 + wave) * 64 + b * 32 + li] = t2;
                    red[(2 * 4 + wave) * 64 + b * 32 + li] = t4;
                }
            }
            __syncthreads();
            const int trow = p.stat_row0 + mt;
            if (tid < 64) {
                auto fold4 = [&](int k) { return (red[(k * 4 + 0) * 64 + tid] + red[(k * 4 + 1) * 64 + tid]) + (red[(k * 4 + 2) * 64 + tid] + red[(k * 4 + 3) * 64 + tid]); };
                const float t1 = fold4(0), t2 = fold4(1), t4 = fold4(2);
                const long c = n0 + tid;
                // (read by another workgroup of this launch when the in-kernel finalize is on: memory-side stores)
                auto put = [&](float* part, int k, float v) {
                    float* dstp = part + ((long)k * p.Cd + c) * p.stat_rows + trow;
                    if (p.bnb_cnt) __hip_atomic_store(dstp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *dstp = v;
                };
                put(p.bnb_part, 0, t1);
                put(p.bnb_part, 1, t2);
                if (two) {
                    put(p.bnb_part2, 0, t1);
                    put(p.bnb_part2, 1, t4);
                }
            }
            if (p.bnb_cnt) {
                // in-kernel finalize, as igemm_kernel.h bnb_finish (BN = 64 columns per workgroup): the last workgroup of each group of
                // 64 partial rows folds the group, the last group-folder of this column tile folds the groups — fixed order, integer counters
                constexpr int G = 64;
                const int NG = (p.stat_rows + G - 1) / G, gq = trow / G;
                const int nbn = two ? 2 : 1;
                int* const cnt = p.bnb_cnt + nt * (NG + 1);
                int* const s_flag = reinterpret_cast<int*>(red + 3 * 4 * 64);
                auto xld = [](const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                DBN_RACE_JITTER();
                if (tid == 0) {
                    const int gsize = min(G, p.stat_rows - gq * G);
                    const int last = __hip_atomic_fetch_add(cnt + 1 + gq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
                    if (last) __hip_atomic_store(cnt + 1 + gq, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_flag = last;
                }
                __syncthreads();
                if (*s_flag) {
                    const int r0 = gq * G, r1_ = min(p.stat_rows, r0 + G);
                    for (int it = tid; it < nbn * 2 * 64; it += 256) {
                        const int bq = it / 128, ks = (it / 64) & 1, cl = it % 64;
                        const float* src = (bq ? p.bnb_part2 : p.bnb_part) + ((long)ks * p.Cd + n0 + cl) * p.stat_rows;
                        double sd = 0.0;
                        int r = r0;
                        for (; r + 3 < r1_; r += 4) sd += ((double)xld(src + r) + (double)xld(src + r + 1)) + ((double)xld(src + r + 2) + (double)xld(src + r + 3));
                        for (; r < r1_; ++r) sd += (double)xld(src + r);
                        __hip_atomic_store(p.bnb_grp + (((long)bq * 2 + ks) * p.Cd + n0 + cl) * NG + gq, (float)sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    if (*s_flag) {
                        for (int it = tid; it < nbn * 64; it += 256) {
                            const int bq = it / 64, cl = it % 64;
                            const long c = n0 + cl;
                            const float* g1 = p.bnb_grp + (((long)bq * 2 + 0) * p.Cd + c) * NG;
                            const float* g2 = p.bnb_grp + (((long)bq * 2 + 1) * p.Cd + c) * NG;
                            double a1_ = 0.0, a2_ = 0.0;
                            for (int q_ = 0; q_ < NG; ++q_) {
                                a1_ += (double)xld(g1 + q_);
                                a2_ += (double)xld(g2 + q_);
                            }
                            p.bnb_dbeta[bq][c] = (float)(a1_ * p.bnb_gscale);
                            p.bnb_dgamma[bq][c] = (float)(a2_ * p.bnb_gscale);
                            p.bnb_c1c2[bq][c] = (float)(a1_ * p.bnb_invM);
                            p.bnb_c1c2[bq][p.Cd + c] = (float)(a2_ * p.bnb_invM);
                        }
                    }
                }
            }
        }
        // ---- optional BatchNorm statistics of this 128-pixel tile (pivot, sum, sum of squares per channel; igemm_kernel.h's format)
        if (p.stats) {
            float* const piv = reinterpret_cast<float*>(smem) + X_FLOATS;  // [64]
            float* const r1 = piv + 64;                                    // [4][64]
            float* const r2 = r1 + 4 * 64;                                 // [4][64]
            if (wave == 0 && lh == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b) piv[b * 32 + li] = y[b][0];
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float pv = piv[b * 32 + li];
                float s1 = 0.f, s2 = 0.f;
                if (allv) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = y[b][r] - pv;
                        s1 += d;
                        s2 += d * d;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = ((vmask >> r) & 1u) ? y[b][r] - pv : 0.f;
                        s1 += d;
                        s2 += d * d;
                    }
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lh == 0) {
                    r1[wave * 64 + b * 32 + li] = s1;
                    r2[wave * 64 + b * 32 + li] = s2;
                }
            }
            __syncthreads();
            const int trow = p.stat_row0 + mt;
            if (tid < 64) {
                const float s1 = (r1[tid] + r1[64 + tid]) + (r1[128 + tid] + r1[192 + tid]);
                const float s2 = (r2[tid] + r2[64 + tid]) + (r2[128 + tid] + r2[192 + tid]);
                const long c = n0 + tid;
                dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (0L * p.Cd + c) * p.stat_rows + trow, piv[tid]);
                dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (1L * p.Cd + c) * p.stat_rows + trow, s1);
                dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (2L * p.Cd + c) * p.stat_rows + trow, s2);
            }
            if (nt == 0 || p.bnf_cnt) {  // (with the in-kernel finalize every tile column writes the count — the same value: igemm_kernel.h)  // the number of real pixels of this tile group: every wave holds one (dy, dx) of each tile — lanes 0 and 32 its two halves
                int* const cntp = reinterpret_cast<int*>(r2 + 4 * 64);
                __syncthreads();
                if (li == 0) cntp[wave * 2 + lh] = __builtin_popcount(vmask);
                __syncthreads();
                if (tid == 0) {
                    int c_ = 0;
                    for (int k = 0; k < 8; ++k) c_ += cntp[k];
                    dbn_stat_put(p.bnf_cnt != nullptr, p.stats + 3L * p.Cd * p.stat_rows + trow, (float)c_);
                }
            }
        }
        // ---- stores
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = row_off(r);
#pragma unroll
            for (int b = 0; b < 2; ++b) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[b][r]), rsrcD, (int)off + b * 128, 0, 0);
        }
#endif
        DBN_TRACE_MARK(3);
        // optional in-kernel finalize of the statistics rows (IgemmParams::bnf_cnt, igemm_common.h), behind this item's output stores
        if (p.stats && p.bnf_cnt) {
            __syncthreads();
            dbn_bn_stats_finish(DBN_BNF_ARGS(p), p.stat_row0 + mt, nt, n0, 64, reinterpret_cast<int*>(smem) + (sizeof(smem) / 4 - 4), reinterpret_cast<double*>(smem));
            __syncthreads();
        }

        if constexpr (!PERSIST) break;
        if (!p.work) {
            item_pos += gridDim.x;
            if (item_pos >= total) break;
            __syncthreads();  // (this item's epilogue regions become the next item's patches)
            item = dbn_xcd_remap(item_pos, total);
            continue;
        }
        // the next item (one returning atomic per item and workgroup; the other resident workgroup covers its latency), and the barrier that
        // hands the LDS regions of this item's epilogue over to the next item's patches
        if (tid == 0) *s_next = resolve(next_idx);
        __syncthreads();
        item = *s_next;  // (rewritten at the end of the next item only: behind that item's barriers)
    }
    if (PERSIST && p.work && tid == 0) {  // the last workgroup out leaves the counters at zero for the next launch on this stream
        if (__hip_atomic_fetch_add(p.work + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            for (int x = 0; x < 9; ++x) __hip_atomic_store(p.work + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// U = G g G^T per (output, input) channel pair, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; panel [I/16][16 points][4 chunks][O][4]
// dgrad = 1: the panel of the DATA GRADIENT of the conv with weights w [Ow][Iw][3][3] — itself a 3x3 / stride-1 / pad-1 convolution of
// dy (Ow channels) with the filters rotated by 180 degrees and the channel roles swapped: g'[o' = iw][c' = ow][r][s] = w[ow][iw][2-r][2-s];
// then O = Iw output and I = Ow input channels
__global__ void winograd_pack_kernel(const float* __restrict__ w, int O, int I, int Cs, int dgrad, float* __restrict__ out) {
    const long total = (long)Cs * O;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx % O), ci = (int)(idx / O);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = ci >= I ? 0.f : dgrad ? w[(((long)ci * O + o) * 3 + (2 - r)) * 3 + (2 - s)] : w[(((long)o * I + ci) * 3 + r) * 3 + s];
        float t[4][3];  // G g
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * ((g[0][s] + g[1][s]) + g[2][s]);
            t[2][s] = 0.5f * ((g[0][s] - g[1][s]) + g[2][s]);
            t[3][s] = g[2][s];
        }
        const int cb = ci >> 4, c4 = (ci >> 2) & 3, e = ci & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u[4] = {t[i][0], 0.5f * ((t[i][0] + t[i][1]) + t[i][2]), 0.5f * ((t[i][0] - t[i][1]) + t[i][2]), t[i][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) out[((((long)cb * 16 + 4 * i + j) * 4 + c4) * O + o) * 4 + e] = u[j];
        }
    }
}

// every panel of a model in one launch: blockIdx.y = job
struct WinoPackJob {
    const float* w;
    float* out;
    int O, I, Cs, dgrad;
};
#ifndef DBN_WPACK_GRID
// workgroups per job: FEW on purpose — the launch runs on the second stream beside the fp32 stem conv and ends long before the first
// Winograd conv needs a panel; wider grids finish sooner and slow the stem more (fp32 step, one box, interleaved three times: 16: 727 / 724 /
// 724 images/s, 64: 723 / 720 / 721, 256: 720 / 720 / 720) — the opposite of pack_many_kernel in the 16-bit modes (pack.hip)
#define DBN_WPACK_GRID 16
#endif
__global__ void winograd_pack_many_kernel(const WinoPackJob* __restrict__ jobs) {
    const WinoPackJob j = jobs[blockIdx.y];
    const long total = (long)j.Cs * j.O;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx % j.O), ci = (int)(idx / j.O);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = ci >= j.I ? 0.f : j.dgrad ? j.w[(((long)ci * j.O + o) * 3 + (2 - r)) * 3 + (2 - s)] : j.w[(((long)o * j.I + ci) * 3 + r) * 3 + s];
        float t[4][3];  // G g
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * ((g[0][s] + g[1][s]) + g[2][s]);
            t[2][s] = 0.5f * ((g[0][s] - g[1][s]) + g[2][s]);
            t[3][s] = g[2][s];
        }
        const int cb = ci >> 4, c4 = (ci >> 2) & 3, e = ci & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u[4] = {t[i][0], 0.5f * ((t[i][0] + t[i][1]) + t[i][2]), 0.5f * ((t[i][0] - t[i][1]) + t[i][2]), t[i][2]};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) j.out[((((long)cb * 16 + 4 * i + jj) * 4 + c4) * j.O + o) * 4 + e] = u[jj];
        }
    }
}

}  // namespace

int dbn_launch_winograd_pack_many(const void* jobs, int n, hipStream_t st) {
    hipLaunchKernelGGL(winograd_pack_many_kernel, dim3(DBN_WPACK_GRID, n), dim3(256), 0, st, reinterpret_cast<const WinoPackJob*>(jobs));
    return dbn_status();
}

// LIN form (consecutive tiles, full-width band) for maps whose band fits the LDS patch: 32 tiles span at most (31 + TW) / TW + 1 tile rows
int dbn_winograd_linear(int H, int W) {
    const int TW = (W + 1) / 2, TH = (H + 1) / 2;
    const int tile_rows = std::min(TH, (31 + TW - 1) / TW + 1);
    return (2 * tile_rows + 2) * (2 * TW + 2) <= W_LIN_PPX;
}
extern "C" int dbn_winograd_rows(int N, int H, int W) {
    if (dbn_winograd_linear(H, W)) return N * ((((H + 1) / 2) * ((W + 1) / 2) + 31) / 32);
    return N * ((H + 7) / 8) * ((W + 15) / 16);
}

// The work counters of the persistent form: nine ints per stream (launches on one stream are ordered, so they share them; launches on
// different streams may overlap and must not), allocated and zeroed on the stream's first launch — before any graph capture, which the
// engine only starts after eager warm-up steps — and left zero by every launch.
static int g_wino_persistent = 0;  // (measured neutral on MI355X, round 5: off by default)
extern "C" int dbn_set_winograd_persistent(int on) {  // test / A-B hook: 0 = one workgroup per item (round 4's form), 1 = pulled items, 2 = static schedule
    g_wino_persistent = on;
    return DBN_OK;
}
static int g_wino_cbs = 1;
extern "C" int dbn_set_winograd_blocks_per_barrier(int n) {  // test / A-B hook: 1 or 2 channel blocks per barrier (patch form, Cs % 32 == 0)
    g_wino_cbs = n == 2 ? 2 : 1;
    return DBN_OK;
}
static int g_wino_stagger = 1000;
extern "C" int dbn_set_winograd_stagger(int permille) {  // test / A-B hook: 0 = no phase stagger of the two residents of a CU
    g_wino_stagger = permille;
    return DBN_OK;
}
static int* wino_work_counters(hipStream_t st) {
    static std::mutex mu;
    static std::unordered_map<hipStream_t, int*> table;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(st);
    if (it != table.end()) return it->second;
    int* buf = nullptr;
    if (hipMalloc(&buf, 16 * sizeof(int)) != hipSuccess || hipMemset(buf, 0, 16 * sizeof(int)) != hipSuccess) return nullptr;
    table[st] = buf;
    return buf;
}

// Workgroups of a launch: persistent (p.work non-NULL) at most two per CU (226-255 registers, ~71 KB of LDS: what a CU holds), in
// multiples of eight so that blockIdx.x & 7 keeps naming the XCD; otherwise one per item.
int dbn_launch_winograd_f32(IgemmParams& p, hipStream_t st) {
    const bool lin = dbn_winograd_linear(p.Hdf, p.Wdf);
    const int items = dbn_winograd_rows(p.N, p.Hdf, p.Wdf) * (p.Cd >> 6);
    if (items <= 0) return DBN_OK;
    constexpr int slots = 2 * 256;  // MI355X: 256 CUs
    int grid = items;
    p.work = (g_wino_persistent == 1 && items > slots) ? wino_work_counters(st) : nullptr;  // (a single round: nothing to pull)
    if (p.work || (g_wino_persistent == 2 && items > slots)) grid = slots;
    p.work_items = items;
    auto magic = [](int d) { return ((1ULL << 40) + (unsigned long long)d - 1) / (unsigned long long)d; };
    {
        const int tw = lin ? 1 : (p.Wdf + 15) / 16;
        const int tpi = lin ? (((p.Hdf + 1) / 2) * ((p.Wdf + 1) / 2) + 31) / 32 : ((p.Hdf + 7) / 8) * tw;
        if ((long)items >= (1L << 23) || tpi > 65536 || (p.Cd >> 6) > 65536) return DBN_ERR_ARG;  // (the multipliers' exact range)
        p.wino_tw = tw;
        p.wino_tpi = tpi;
        p.wino_m_ntn = magic(p.Cd >> 6);
        p.wino_m_tpi = magic(tpi);
        p.wino_m_tw = magic(tw);
    }
    // the stagger of the persistent forms: g_wino_stagger permille of one item's matrix time (Cs / 16 blocks x 64 MFMAs x 64 clocks per wave)
    p.stagger_units = (grid < items && g_wino_stagger > 0) ? (int)((long)(p.Cs >> 4) * 4096 / 1024 * g_wino_stagger / 1000) : 0;
    p.trace = (DBN_TRACE && dbn_g_trace && grid <= dbn_g_trace_blocks) ? dbn_g_trace : nullptr;
    const bool persist = grid < items;
    const bool cb2 = !lin && g_wino_cbs == 2 && (p.Cs & 31) == 0;
    if (lin && persist) hipLaunchKernelGGL((winograd_f32_kernel<true, true, 1>), dim3(grid), dim3(256), 0, st, p);
    else if (lin) hipLaunchKernelGGL((winograd_f32_kernel<true, false, 1>), dim3(grid), dim3(256), 0, st, p);
    else if (persist && cb2) hipLaunchKernelGGL((winograd_f32_kernel<false, true, 2>), dim3(grid), dim3(256), 0, st, p);
    else if (persist) hipLaunchKernelGGL((winograd_f32_kernel<false, true, 1>), dim3(grid), dim3(256), 0, st, p);
    else if (cb2) hipLaunchKernelGGL((winograd_f32_kernel<false, false, 2>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((winograd_f32_kernel<false, false, 1>), dim3(grid), dim3(256), 0, st, p);
    return dbn_status();
}

int dbn_launch_winograd_pack(const float* w, int O, int I, int Cs, int dgrad, float* out, hipStream_t st) {
    hipLaunchKernelGGL(winograd_pack_kernel, dim3(dbn_grid((long)Cs * O)), dim3(256), 0, st, w, O, I, Cs, dgrad, out);
    return dbn_status();
}
