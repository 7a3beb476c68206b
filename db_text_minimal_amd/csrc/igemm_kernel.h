// The implicit-GEMM convolution kernel and its launcher; instantiated per conv-math family by
// conv_f32.hip / conv_x3.hip / conv_b16.hip.
#pragma once
#include "igemm_common.h"

#ifndef DBN_DBG
#define DBN_DBG 0
#endif

namespace {

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
// and loses 35 % — the experiment (round 1-2: DBN_IGEMM_W4) was removed in round 3.
// pixel-patch kernels with three planes: two waves per SIMD (<= 256 registers; the fully unrolled nine stages had taken 257)
// exact-fp32 16K-element tiles with the BatchNorm-backward epilogue (EPI = 1): three waves per SIMD like the plain kernel (the row
// sweep of that epilogue peaks 1-3 registers above the 104 that three waves allow; the attribute makes the allocator fit it)
// bf16-storage pixel-patch kernels with that epilogue: four waves like the plain kernel (117 registers as written = three)
// exact-fp32 64 x 64 tiles without that epilogue: seven waves per SIMD as before round 4 (the buffer-store epilogue's row offsets peak
// two registers above the 72 that seven waves allow)
#define DBN_IGEMM_OCC(BM, BN, NS, MODE, PATCH, EPI, AT) \
    __attribute__((amdgpu_waves_per_eu(((MODE) == 3 && (BN) == 256) ? 2 : ((PATCH) && (NS) == 3 && (BN) == 64) ? 2 : ((EPI) == 1 && (NS) == 0 && (BM) * (BN) == 16384) ? 3 : \
                                       ((EPI) == 1 && (AT) == 1 && (PATCH) && (BN) == 64) ? 4 : \
                                       ((EPI) == 0 && (NS) == 0 && (AT) == 0 && (BM) * (BN) == 4096 && (MODE) < 2) ? 7 : 1, 8)))

// AT (activation storage type of src and dst): 0 fp32; 1 bf16 / 2 fp16 need NS = 1 — the gather then fetches 16-byte pieces
// of EIGHT stored 16-bit channels that go to LDS unchanged (no conversion, the LDS image of the NS = 1 path is exactly the
// stored format), the accumulators are rounded to the storage type on the way out.
// AT = 3 (NS = 3): src is the PRE-SPLIT form of an fp32 tensor — three bf16 planes [3][N,H,W,C] with a0 + a1 + a2 == a exactly
// (dbn_split3) — gathered the same way (3 x 2 pieces per row and k-tile, no conversion: splitting at staging time redid the
// split for every one of the 9 taps that re-reads an element and made the bf16x3 kernels VALU-bound); dst is fp32.
// PATCH (3x3, stride 1, pad 1, 16-bit matrix math; Hd % 8 == 0, Wd % 16 == 0, Cs % 32 == 0): the M tile is an 8 x 16 PIXEL PATCH
// and the A operand is not gathered per tap at all — see the main loop.
// EPI = 1: the epilogue also produces the partial sums of the BatchNorm backward that consumes dst (IgemmParams::bnb_*).
// BLK = false: the source's channel count is not a multiple of 16 (the stem: Cs = 4) — the K walk then crosses taps inside a
// k-step; forward convs on fp32 tensors only.  A compile-time fact: as a runtime flag its inner `while` put loops and branches
// into every kernel's k-loop (and an s_waitcnt vmcnt(0) at the loop header that drained the two-tile prefetch every iteration).
template <int BM, int BN, int WM, int WN, int MODE, int NS, int AT = 0, bool PATCH = false, int EPI = 0, bool BLK = true>
__global__ __launch_bounds__(WM* WN * 64) DBN_IGEMM_OCC(BM, BN, NS, MODE, PATCH, EPI, AT) void igemm_f32_kernel(const IgemmParams p) {
    static_assert(EPI == 0 || ((AT == 0 || AT == 1) && MODE < 3), "BatchNorm-backward sums: fp32 or bf16 storage, no pyramid form");
    static_assert(BLK || (AT == 0 && MODE == 0 && !PATCH && EPI == 0), "non-blocked K walk: forward conv on fp32 tensors");
    static_assert(AT == 0 || ((AT == 1 || AT == 2) && NS == 1) || (AT == 3 && NS == 3), "storage type / matrix math combination");
    static_assert(!PATCH || (BM == 128 && WM == 2 && WN == 2 && MODE < 2 && (NS > 0 || AT == 0) && AT != 3), "patch form");
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int ES = AT == 0 ? 4 : 2;            // bytes per stored source element
    constexpr bool DST_F32 = AT == 0 || AT == 3;
#ifndef DBN_BUFST
#define DBN_BUFST 1
#endif
    constexpr bool BUFST = DBN_BUFST && DST_F32 && MODE < 2;  // plain epilogue through raw buffer stores (see the end of the kernel)
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
#ifndef DBN_DIRECTB
#define DBN_DIRECTB 1
#endif
    // DIRECTB (exact fp32 on fp32 tensors; round 3): the weight panel's pieces are fetched in MFMA-fragment order straight into
    // registers — lane (li, lh) of wave column wn needs chunk 2*s2 + lh of column wn*TN + b*32 + li: 16 contiguous bytes of the packed
    // panel, a half-wave 512 — one k-tile ahead: no LDS image, no ds_write / ds_read for B, half the LDS footprint (64 x 64 tile:
    // 8.4 KB).  The two waves of a wave column fetch the same pieces (the second from L1).  Same products in the same order.
    // Measured (interleaved A/B on one box, -DDBN_DIRECTB=0 is the staged form): the kernels alone within 1 % either way, the
    // two-stream step +0.6 ... +0.9 % (538.2 / 541.8 against 534.8 / 536.9 images/s) — the smaller footprint co-resides better with
    // the weight-gradient stream's workgroups.
// (also for the bf16 matrix modes on fp32 tensors — fragment = slice lh of plane t of the pre-split panel: bf16x3 739 -> 749 images/s)
#ifndef DBN_DIRECTB_NS
#define DBN_DIRECTB_NS 1
#endif
    constexpr bool DIRECTB = DBN_DIRECTB && (NS == 0 || DBN_DIRECTB_NS) && AT == 0 && !PATCH;
    constexpr int NF = NS == 0 ? 2 : NS;  // B fragments (16 bytes each) per accumulator column block and k-tile
    constexpr int UNIT = A_IMG + (DIRECTB ? 0 : B_IMG);
    constexpr int STAGE = KU * UNIT;
    constexpr int NSX = NS > 0 ? NS : 1;
    static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for the workgroup");
    // 16-bit storage (AT != 0): LDS-DMA ring (see the main loop): stages of DMA_SU units, unpadded images [plane][k/8][row]
    constexpr int DMA_SU = 2;
#ifndef DBN_DBG16
#define DBN_DBG16 0  // profile by deletion of the 16-bit ring loop (tools/flavour.sh <name> conv_b16.hip "-DDBN_DBG16=<bits>"; timing only, wrong results): 1 no per-instruction address arithmetic, 2 no weight-fragment loads, 4 no LDS fragment reads, 8 no A DMA
#endif
#ifndef DBN_DIRECTB16
#define DBN_DIRECTB16 1
#endif
    // DB16 (stored 16-bit operands, generic ring, not the pyramid form; round 3): the weight fragments come straight from the packed
    // panel into registers, one stage ahead (as DIRECTB for fp32) — the ring carries the A panel only: half the DMA instructions and
    // their scalar bookkeeping, no fragment reads for B, a ring half the size.  bf16 step 1624 -> 1683 images/s (interleaved A/B on
    // one box; -DDBN_DIRECTB16=0 builds the ring with both panels)
#ifndef DBN_DIRECTB16_PYR
#define DBN_DIRECTB16_PYR 1  // round 5: ... and the pyramid form (MODE 3) too: its ring is bound by the bytes it can keep in flight
#endif
    constexpr bool DB16 = DBN_DIRECTB16 && AT != 0 && AT != 3 && (MODE != 3 || DBN_DIRECTB16_PYR) && !PATCH;
    constexpr int DMA_UNIT = NP * 2 * BM + (DB16 ? 0 : NSX * 2 * BN);  // 16-byte slots of one unit
    constexpr int DMA_STAGE = DMA_SU * DMA_UNIT;
#ifndef DBN_PYR_NSTG
#define DBN_PYR_NSTG 4  // ring depth of the register-fed pyramid form (A panel only: 8 KB per stage at 128 rows).  6 (48 KB, still three workgroups per CU) measured equal: cfg5 14.55 / 14.50 vs 14.56 / 14.59 ms, bf16 step 1660 / 1671 vs 1665 / 1664 images/s
#endif
#ifndef DBN_PYR_NSTG_WIDE
#define DBN_PYR_NSTG_WIDE 4  // ... and of its 128 x 256 tile (round 6; two workgroups per CU, the ring inside the 64 KB epilogue region)
#endif
    constexpr int DMA_NSTG = (DB16 && MODE == 3 && BN == 256 && DMA_STAGE * 16 * DBN_PYR_NSTG_WIDE <= 64 * 1024) ? DBN_PYR_NSTG_WIDE :
                             (DB16 && MODE == 3 && DMA_STAGE * 16 * DBN_PYR_NSTG <= 48 * 1024) ? DBN_PYR_NSTG :
                             DMA_STAGE * 16 * 4 <= 64 * 1024 ? 4 : DMA_STAGE * 16 * 3 <= 160 * 1024 ? 3 : 2;
    // PATCH: two patch buffers [plane][4 k/8 slices][10 x 18 pixels] + a ring of P_NSTG weight stages of two units
    constexpr int P_PATCH = NSX * 4 * 180;
    constexpr int P_BUNIT = NSX * 2 * BN, P_BSTAGE = 2 * P_BUNIT;
#ifndef DBN_P_NSTG
#define DBN_P_NSTG 4  // weight-ring depth of the single-plane pixel-patch kernels (A/B build; round 5: 6 / 8 stages — fewer workgroups per CU — are slower: cfg5 12.29 -> 12.52 / 12.56 ms, bf16 step 1704 -> 1687 / 1642 images/s)
#endif
    constexpr int P_NSTG = NSX == 1 ? DBN_P_NSTG : 3;
    // three planes: ONE patch buffer (refilled between two barriers at a channel-block boundary) keeps the workgroup at 70 KB
    // so that two fit a CU; with a second buffer it was alone on its CU (103 KB, one wave per SIMD: every LDS latency exposed)
    constexpr int P_NBUF = NSX == 1 ? 2 : 1;
    // (EPI = 1 on fp32 storage: the row-major epilogue needs half a tile — a whole one for one accumulator block per wave — of fp32)
#ifndef DBN_DIRECTBP
#define DBN_DIRECTBP 1
#endif
    // DBP (pixel-patch kernels of the three-plane bf16x3 mode; round 3): the weight fragments straight into registers, one stage (two
    // taps) ahead — no weight ring, and the only barriers left are the two around a patch refill (one channel block = 18 units
    // between them).  Measured (interleaved A/B on one box): bf16x3 714 -> 732 images/s; the single-plane kernels (bf16 storage,
    // two patch buffers, ten barriers per block that also pace the ring) lose with it: 1644 -> 1620 — they keep the ring.
#ifndef DBN_DBP_NS1
#define DBN_DBP_NS1 0  // 1: the single-plane (bf16 / fp16) pixel-patch kernels too (A/B build; round 5 again: cfg5 14.22 / 14.20 vs 14.21 / 14.23 ms, bf16 step 1677 / 1677 vs 1662 / 1652 images/s: the ring stays)
#endif
    constexpr bool DBP = DBN_DIRECTBP && PATCH && (NS == 3 || (DBN_DBP_NS1 && NS == 1));
    constexpr int LOOP_SMEM = PATCH ? P_NBUF * P_PATCH + ((DBP || NS == 0) ? 0 : P_NSTG * P_BSTAGE) : AT != 0 ? DMA_NSTG * DMA_STAGE : 2 * STAGE;
    constexpr int EPI_SMEM = (EPI == 1 && DST_F32) ? BM * BN / (4 * (MI >= 2 ? 2 : 1)) : 0;
    // (16-bit destinations: the output tile is staged through LDS, BM rows of BN + 8 elements)
    constexpr int OUT_SMEM = DST_F32 ? 0 : (BM * (BN + 8) * 2 + 15) / 16;
    constexpr int SMEM_A = LOOP_SMEM > EPI_SMEM ? LOOP_SMEM : EPI_SMEM;
    // (+ one slot: the in-kernel finalize of the BatchNorm sums keeps its flag in the last 16 bytes, behind every scratch region)
    // (... and at least the 3 * NT doubles of the in-kernel statistics finalize, dbn_bn_stats_finish)
    constexpr int FIN_SMEM = (3 * NT * 8 + 15) / 16;
    constexpr int SMEM_B = SMEM_A > OUT_SMEM ? SMEM_A : OUT_SMEM;
    __shared__ f32x4 smem[(SMEM_B > FIN_SMEM ? SMEM_B : FIN_SMEM) + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // First-round stagger.  Every workgroup of a launch takes the same time, and the first 256 x (workgroups per CU) of them start
    // together: the workgroups sharing a CU then run in LOCKSTEP for the whole launch — all in their prologue (index arithmetic, first
    // loads: no MFMA) at the same time, all in their epilogue at the same time — and the matrix pipe idles through every such phase
    // (profile by deletion: ~4 us per 128 x 64 tile exposed whatever K is: 0.79 of peak at K = 576 with every load deleted, 0.94 at
    // K = 2304).  Delaying the first-round workgroup in wave slot j by j / slots of a tile time puts the residents of a CU out of phase;
    // later workgroups inherit the offsets because each starts when its predecessor in the slot ends.
    // Phase priority.  Per-workgroup timestamps (make TRACE=1, tools/trace_probe.py) show what a tile's fixed cost is: with six other
    // workgroups of the CU inside their MFMA loops, a workgroup's prologue (index arithmetic, first loads, first barrier: ~2 us on an
    // idle CU) takes ~14 us and its epilogue ~5 us of a 68 us lifetime at K = 576 — its scalar / vector instructions wait behind the
    // residents' MFMAs for issue — and for that long its four waves feed no MFMA.  Raised wave priority for exactly these phases gets
    // a workgroup into (and out of) its loop sooner; the loop itself runs at the default priority.
    if (p.phase_prio) __builtin_amdgcn_s_setprio(3);
    DBN_TRACE_MARK(0);
#if DBN_TRACE
    if (p.trace && threadIdx.x == 0) p.trace[(long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
#endif
    if (p.stagger_units > 0 && (int)blockIdx.x < p.stagger_blocks) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;  // HW_REG_HW_ID.WAVE_ID: this wave's slot on its SIMD
        const unsigned n = __builtin_amdgcn_readfirstlane(slot * (unsigned)p.stagger_units);
        for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 clocks
    }

    int tile = dbn_xcd_remap(blockIdx.x, gridDim.x);
    IgemmClass q;
    int q_row_base = 0, q_wpk_off = 0;
    if (MODE == 3) {
        q.Hd = p.Hdf >> 3; q.Wd = p.Wdf >> 3; q.M = p.N * q.Hd * q.Wd;
        // tile order: m-tile major, pixel class minor — the 64 classes of one image region run together, so their
        // overlapping 3x3 neighbourhoods of the finest level are served by the L2 instead of 9 trips to HBM
        const int mtiles = (q.M + BM - 1) / BM, ntn_ = p.Cd / BN;
        int mt_ = tile / (64 * ntn_), rem_ = tile - mt_ * (64 * ntn_);
        int c = rem_ / ntn_;
        if (p.pyr_group > 0) {
            // round 6 (the wide tile): G row tiles x the 8 classes of ONE class row run together — the workgroups resident on an XCD then
            // share a few classes' weight panels (the level 1-3 panels are class-specific: 250 KB each, 16 MB for the 64 classes against
            // 4 MB of L2; profile by deletion: the weight-fragment loads are a third of the launch) and still share the activation rows of
            // the three horizontal taps.  Order inside a block of G x 64 x ntn tiles: class row, row tile, class column, column tile
            const int G = p.pyr_group, per = G * 64 * ntn_;
            const int blk = tile / per, t = tile - blk * per;
            const int cr = t / (G * 8 * ntn_), t2 = t - cr * (G * 8 * ntn_);
            const int g = t2 / (8 * ntn_), t3 = t2 - g * (8 * ntn_);
            const int cc = t3 / ntn_;
            mt_ = blk * G + g;
            c = cr * 8 + cc;
            rem_ = c * ntn_ + (t3 - cc * ntn_);
            if (mt_ >= mtiles) return;  // (the last block is padded to G row tiles)
        }
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
    // SOFF (blocked K walk on fp32 tensors): the tap / channel-block part of a gather address is the same for every lane, so it
    // travels in the buffer instruction's SCALAR offset (not bounds-checked) and a lane only supplies its pixel's origin — or an
    // out-of-range offset when the tap falls into the padding.  Per load and k-step that leaves two compares and a select for the
    // vector ALU instead of the multiply-add chain (profile by deletion: the address math cost 0.06 of the MFMA peak).  The scalar
    // offset must not be negative: the resource's base lies `a_shift` bytes in front of the tensor.
    constexpr bool SOFF = BLK && AT == 0 && !PATCH;
    unsigned a_org[A_LD];  // SOFF: byte offset (from the shifted base) of this lane's piece at the origin tap
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
        a_org[j] = 0;
        if (MODE == 3) {
            a_n[j] = n;
            a_hd[j] = ok ? hd : -(1 << 20);
            a_wd[j] = wd;
        } else if (MODE == 0) {
            a_hb[j] = ok ? hd * p.stride - q.pad_h : -(1 << 20);  // a far-away row can never be in range
            a_wb[j] = wd * p.stride - q.pad_w;
            // origin: tap (0, 0) seen from the shifted base = the pixel (hd*stride, wd*stride) itself
            if (SOFF) a_org[j] = (unsigned)(a_nb[j] + (hd * p.stride * p.Ws + wd * p.stride) * p.Cs + A_CH * a_chunk) * (unsigned)ES;
        } else {
            a_hb[j] = ok ? hd + q.pad_h : -(1 << 20);
            a_wb[j] = wd + q.pad_w;
            // origin: tap (R-1, S-1) seen from the shifted base = the pixel (hd + pad_h, wd + pad_w) (may lie below the tensor)
            if (SOFF) a_org[j] = (unsigned)(a_nb[j] + ((hd + q.pad_h) * p.Ws + wd + q.pad_w) * p.Cs + A_CH * a_chunk) * (unsigned)ES;
        }
    }
    // base shift in bytes (see SOFF) and the resource over [src - shift, src + bytes + margin): origins of the transposed forms
    // reach up to pad rows below the tensor, which must still pass the bounds check (everything stays below OOB_OFFSET)
    auto soff_rsrc = [&](const void* base, unsigned bytes, int Ws_, int R_, int S_, int padh, int padw) {
        const unsigned shift = (unsigned)((MODE == 0 ? padh * Ws_ + padw : (R_ - 1) * Ws_ + (S_ - 1)) * p.Cs) * (unsigned)ES;
        const unsigned margin = (unsigned)((padh * Ws_ + padw + 1) * p.Cs) * (unsigned)ES;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base) - shift), 0, bytes + shift + margin,
                                                 0x00020000);
    };
    if (SOFF && MODE != 3) rsrc = soff_rsrc(p.src, p.src_bytes, p.Ws, q.R, q.S, q.pad_h, q.pad_w);
    // K order (must match the weight panels, pack_weights_kernel):
    //   Cs % 16 == 0: k = ((cb*R + r)*S + s)*16 + cl with ci = 16*cb + cl — every k-tile is one tap of one
    //                 16-channel block and the R*S taps of a block are consecutive k-tiles, so the 9 re-reads of
    //                 an input pixel's 64-byte slice happen back to back (L1/L2 hits instead of a trip to the fabric);
    //   otherwise (stem, Cs = 4): k = (r*S + s)*Cs + ci.
    // (always, for 16-bit storage and pre-split planes — checked on the host: a compile-time fact there, which removes the
    // non-blocked walk and its branches from the k-loop)
    constexpr bool blocked = BLK;
    int kidx = A_CH * a_chunk;
    int k_ci, k_r, k_s;
    int kt_begin = 0, kt_end = qKT;
    if (MODE < 2 && p.ksplit > 1) {
        kt_begin = blockIdx.y * p.kt_per;
        kt_end = min(qKT, kt_begin + p.kt_per);
    }
    int k_cb = 0;  // SOFF: first channel of the current 16-channel block (wave-uniform, like k_r / k_s there)
    if (blocked) {  // k-tile kt is tap (kt % RS) of channel block (kt / RS)
        const int rs = max(1, q.R * qS), cb = kt_begin / rs, tap = kt_begin - cb * rs;
        kidx += 16 * kt_begin;
        k_ci = 16 * cb + A_CH * a_chunk;
        k_cb = 16 * cb;
        k_r = tap / qS;
        k_s = tap - k_r * qS;
    } else {
        const int k_tap = kidx / p.Cs;
        k_ci = kidx - k_tap * p.Cs;
        k_r = k_tap / qS;
        k_s = k_tap - k_r * qS;
    }

    unsigned aoff[KU][A_LD];
    unsigned asoff[KU];  // SOFF: the scalar offset of unit u (bytes)
    int kend = (MODE < 2 && p.ksplit > 1) ? min(qK, 16 * kt_end) : qK;  // units past the end (of K, or of this split's range) gather zeros
    auto next_offsets = [&](int u = 0) {  // offsets of the current k position (unit u of the barrier interval), then advance by 16 k
        if constexpr (SOFF) {
            // (k_r, k_s, k_cb derive from the workgroup's k-tile range only: scalar registers)
            const int tr = MODE == 0 ? k_r : qR - 1 - k_r, ts = MODE == 0 ? k_s : qS - 1 - k_s;
            asoff[u] = (unsigned)((tr * gWs + ts) * p.Cs + k_cb) * (unsigned)ES;
            const bool kv = kidx - A_CH * a_chunk < kend;  // (uniform: kidx differs between lanes by the chunk only, kend % 16 == 0 here)
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int hs = MODE == 0 ? a_hb[j] + k_r : a_hb[j] - k_r;
                const int ws = MODE == 0 ? a_wb[j] + k_s : a_wb[j] - k_s;
                const bool v = kv && (unsigned)hs < (unsigned)gHs && (unsigned)ws < (unsigned)gWs;
                aoff[u][j] = v ? a_org[j] : OOB_OFFSET;
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int hs = MODE == 0 ? a_hb[j] + k_r : a_hb[j] - k_r;
                const int ws = MODE == 0 ? a_wb[j] + k_s : a_wb[j] - k_s;
                const bool v = kidx < kend && (unsigned)hs < (unsigned)gHs && (unsigned)ws < (unsigned)gWs;
                const unsigned off = (unsigned)(a_nb[j] + (hs * gWs + ws) * p.Cs + k_ci) * (unsigned)ES + (AT == 3 ? a_pl[j] : 0u);
                aoff[u][j] = v ? off : OOB_OFFSET;
            }
            asoff[u] = 0;
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
            k_cb += wr_ ? 16 : 0;
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
    // weight panel: a lane's piece of a k-tile lies at a fixed offset inside the tile, the tile's offset is wave-uniform and
    // travels as the scalar offset of the buffer load (no per-lane pointer arithmetic in the k-loop)
    unsigned b_voff[B_LD];
    unsigned bf_voff[NF][NI];  // DIRECTB
    int b_lds[B_LD];
    bool b_on[B_LD];
    __amdgpu_buffer_rsrc_t rsrcB;
    int b_kt = 0, b_ktmax = 0;  // next k-tile to load (clamped to the last one: loads past the end re-read it), scalar
    const unsigned b_step_bytes = (unsigned)((NS == 0 ? 4 : 2 * NS) * p.Cd) * 16u;  // bytes per k-tile
    auto panel_setup = [&](const float* panel, int ktiles) {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int idx = tid + j * NT;
            b_on[j] = B_FULL || idx < B_PIECES;
            const int c = b_on[j] ? idx / BN : 0, n = idx - (idx / BN) * BN;  // c: k-chunk (NS==0) or split*2+k8 (NS>0)
            b_voff[j] = (unsigned)(c * p.Cd + n0 + n) * 16u;
            b_lds[j] = c * BS + n;
        }
#pragma unroll
        for (int s2 = 0; s2 < NF; ++s2)  // (NS == 0: chunk 2*s2 + lh of the k-tile; NS > 0: slice lh of plane s2)
#pragma unroll
            for (int b = 0; b < NI; ++b) bf_voff[s2][b] = (unsigned)((2 * s2 + lh) * p.Cd + n0 + wn * TN + b * 32 + li) * 16u;
        rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(panel), 0, (unsigned)ktiles * b_step_bytes, 0x00020000);
    };
    if (MODE != 3) {
        panel_setup(p.wpk + q_wpk_off, qKT);
        b_kt = kt_begin;
        b_ktmax = max(kt_end - 1, kt_begin);
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
        panel_setup(p.seg_wpk[g] + (NS == 0 ? krows * p.Cd : krows * p.Cd * NS / 2), qKT);
        b_kt = 0;
        b_ktmax = max(qKT - 1, 0);
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            a_nb[j] = a_n[MODE == 3 ? j : 0] * gHs * gWs * p.Cs;
            a_hb[j] = a_hd[MODE == 3 ? j : 0] * (8 >> g) + (q.oh0 >> g) + padh;
            a_wb[j] = a_wd[MODE == 3 ? j : 0] * (8 >> g) + (q.ow0 >> g) + padw;
            // (rows past M carry a_hd = -2^20: never valid; their origin is never used, keep it in range of the arithmetic)
            const int hb_ = a_hd[MODE == 3 ? j : 0] < 0 ? 0 : a_hb[j];
            if (SOFF) a_org[j] = (unsigned)(a_nb[j] + (hb_ * gWs + a_wb[j]) * p.Cs + A_CH * a_chunk) * (unsigned)ES;
        }
        if (SOFF) rsrc = soff_rsrc(p.seg_src[g], p.seg_bytes[g], gWs, qR, qS, padh, padw);
        kidx = A_CH * a_chunk;
        k_ci = A_CH * a_chunk;
        k_cb = 0;
        k_r = k_s = 0;
    };

    // two register sets: the global loads of k-tile t+2 are issued while tile t is multiplied and tile t+1 (loaded one
    // iteration earlier) is staged to LDS — a full k-step (~1 us) more latency tolerance than a prefetch distance of one
    // Loads and staging are issued UNCONDITIONALLY every k-step (tiles past the end gather zeros through out-of-range buffer
    // offsets and re-read the last weight tile): with a conditional issue the compiler merges the "issued" and "not issued"
    // paths and waits vmcnt(0) before staging — i.e. also for the set that was just issued — which defeats the distance of two.
    f32x4 ra_[2][KU][A_LD], rb_[2][KU][B_LD];
    f32x4 rbf_[2][NF][NI];  // DIRECTB: B fragments of the current and the next k-tile
    auto issue_b_frag = [&](auto SET) {  // the NEXT weight k-tile (clamped) into fragment set SET
        constexpr int st_ = decltype(SET)::value;
        const unsigned bso = (unsigned)min(b_kt, b_ktmax) * b_step_bytes;
        ++b_kt;
#pragma unroll
        for (int s2 = 0; s2 < NF; ++s2)
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bf_voff[s2][b], (int)bso, 0);
                rbf_[st_][s2][b] = __builtin_bit_cast(f32x4, v_);
            }
    };
    auto issue_loads = [&](auto SET) {
        constexpr int st_ = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j)
                if (!(DBN_DBG & 1)) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)aoff[u][j], (int)asoff[u], 0);
                    ra_[st_][u][j] = __builtin_bit_cast(f32x4, v_);
                }
            if constexpr (DIRECTB) continue;
            const unsigned bso = (unsigned)min(b_kt, b_ktmax) * b_step_bytes;
            ++b_kt;
#pragma unroll
            for (int j = 0; j < B_LD; ++j)
                if (!(DBN_DBG & 2) && (B_FULL || b_on[j])) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)b_voff[j], (int)bso, 0);
                    rb_[st_][u][j] = __builtin_bit_cast(f32x4, v_);
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
        if (DBN_DBG & 4) return;
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
        if constexpr (!DIRECTB) {
#pragma unroll
            for (int j = 0; j < B_LD; ++j)
                if (B_FULL || b_on[j]) Bs[b_lds[j]] = rb[j];
        }
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
    // exact fp32 (NS == 0, round 4): the channel block is ONE k-step wide — 16 channels = four 16-byte chunks per pixel, image
    // [chunk][patch pixel][4 f32]; the fragment of lane (li, lh) for MFMA group s2 is chunk 2*s2 + lh of its pixel, as in the gather loop
    constexpr int CB = NS == 0 ? 16 : 32;             // channels per block
    constexpr int CHUNKS = NS == 0 ? 4 : AT == 0 ? 8 : 4;  // 16-byte pieces per pixel of a channel block
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
        pslot[j] = !on ? -1 : NS == 0 ? chunk * PPX + pix : AT == 0 ? ((chunk >> 1) * PPX + pix) * 2 + (chunk & 1) : chunk * PPX + pix;
    }
    f32x4 pr[PL];
    const int ncb = p.Cs / CB;
    auto load_patch = [&](int cb) {  // (cb == ncb: the loads are issued all the same, so that the counted waits stay constant)
        const unsigned add = (unsigned)(cb * CB * ES);
#pragma unroll
        for (int j = 0; j < PL; ++j) pr[j] = buffer_load_f32x4(rsrc, poff[j] == OOB_OFFSET ? OOB_OFFSET : poff[j] + add);
    };
    auto store_patch = [&](int buf) {
        f32x4* const P = patch + buf * P_PATCH;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            if (pslot[j] < 0) continue;
            if constexpr (NS == 0) {
                P[pslot[j]] = pr[j];
            } else if constexpr (AT == 0) {
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

    if constexpr (NS == 0) {
        // ---- exact fp32 (v_mfma_f32_32x32x2_f32) over the pixel patch.  A 3x3 / stride-1 conv re-reads every input pixel nine times;
        // the gather loop fetches each of them from L1 / L2 again (nine 16-byte gathers per row and channel block, each with its
        // padding test and staging write, one barrier per k-step).  Here a channel block's 10 x 18 patch is fetched ONCE (three
        // 16-byte loads per thread), and the nine k-steps of the block read their A fragments at LDS offsets of the same image:
        // per k-step a wave issues MI * 2 fragment reads, 2 * NI weight-fragment loads (contiguous, straight into registers as in the
        // gather loop) and MI * NI * 8 MFMAs — no address arithmetic, no staging write, no barrier; one barrier per NINE k-steps.
        // Same products in the same k order per accumulator as the gather form: bit-identical results.
        unsigned bvo[2][NI];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < NI; ++b) bvo[s2][b] = (unsigned)((2 * s2 + lh) * p.Cd + n0 + wn * TN + b * 32 + li) * 16u;
        const unsigned bstep32 = (unsigned)(4 * p.Cd) * 16u;  // bytes per k-tile of the fp32 panel
        const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk + q_wpk_off), 0,
                                                                               (unsigned)qKT * bstep32, 0x00020000);
        int w_kt = 0;  // next weight k-tile (clamped to the last one: the fetches past the end re-read it and are never used)
        f32x4 rw[3][2][NI];  // three fragment sets: k-step g uses set g % 3 (nine k-steps per block: tap % 3), fetched two k-steps ahead
        auto issue_w = [&](auto SET) {
            constexpr int st_ = decltype(SET)::value;
            const unsigned so = (unsigned)min(w_kt, qKT - 1) * bstep32;
            ++w_kt;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)bvo[s2][b], (int)so, 0);
                    rw[st_][s2][b] = __builtin_bit_cast(f32x4, v_);
                }
        };
        load_patch(0);
        issue_w(std::integral_constant<int, 0>{});
        issue_w(std::integral_constant<int, 1>{});
        store_patch(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DBN_TRACE_MARK(1);
        if (p.phase_prio) __builtin_amdgcn_s_setprio(0);
        // two channel blocks per trip (Cs % 32 == 0, checked on the host), so that the patch buffer of a block is a compile-time
        // LDS offset of its fragment reads
        auto block = [&](int cb, auto ODD) {
            constexpr int odd = decltype(ODD)::value;
            const f32x4* const P = patch + odd * P_PATCH;
            load_patch(cb + 1);  // (past the last block: out-of-range offsets, zeros, never stored)
            auto tap_step = [&](auto TAP) {
                constexpr int tap = decltype(TAP)::value;
                constexpr int cur = tap % 3;
                constexpr int tr = MODE == 0 ? tap / 3 : 2 - tap / 3, ts = MODE == 0 ? tap % 3 : 2 - tap % 3;
                issue_w(std::integral_constant<int, (tap + 2) % 3>{});  // the weight fragments of the k-step after the next
                __builtin_amdgcn_sched_barrier(0);  // (left alone the loads sink down to one MFMA group before their use)
                f32x4 af[2][MI];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int a = 0; a < MI; ++a) af[s2][a] = P[2 * s2 * PPX + a_pix + (2 * a + tr) * PROW + ts];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int a = 0; a < MI; ++a)
#pragma unroll
                            for (int b = 0; b < NI; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s2][a][e], rw[cur][s2][b][e], acc[a][b], 0, 0, 0);
            };
            tap_step(std::integral_constant<int, 0>{});
            tap_step(std::integral_constant<int, 1>{});
            tap_step(std::integral_constant<int, 2>{});
            tap_step(std::integral_constant<int, 3>{});
            tap_step(std::integral_constant<int, 4>{});
            tap_step(std::integral_constant<int, 5>{});
            tap_step(std::integral_constant<int, 6>{});
            tap_step(std::integral_constant<int, 7>{});
            tap_step(std::integral_constant<int, 8>{});
            if (cb + 1 < ncb) {
                // the other buffer was last read in block cb - 1, and every wave has passed the barrier that ended it
                store_patch(odd ^ 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        for (int cb = 0; cb < ncb; cb += 2) {
            block(cb, std::integral_constant<int, 0>{});
            block(cb + 1, std::integral_constant<int, 1>{});
        }
        DBN_TRACE_MARK(2);
        if (p.phase_prio) __builtin_amdgcn_s_setprio(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else if constexpr (DBP) {
        unsigned bvo[NSX][NI];
#pragma unroll
        for (int t = 0; t < NSX; ++t)
#pragma unroll
            for (int b = 0; b < NI; ++b) bvo[t][b] = (unsigned)((t * 2 + lh) * p.Cd + n0 + wn * TN + b * 32 + li) * 16u;
        // fragment sets: stage st of a channel block uses set (st == 0 ? 2 : st & 1) — nine stages per block, so plain parity would
        // hand stage 8 and the next block's stage 0 the same set
        f32x4 rbP[3][2][NSX][NI];
        auto issue_bp = [&](auto SET) {  // the next two weight k-tiles (one stage) into set SET
            constexpr int st__ = decltype(SET)::value;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kt = g_kt + u;
                const unsigned so = (unsigned)min(kt, qKT - 1) * bstep_bytes;  // (past the end: the last tile again — its A units are zeros... see below)
#pragma unroll
                for (int t = 0; t < NSX; ++t)
#pragma unroll
                    for (int b = 0; b < NI; ++b) {
                        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                        const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)(kt < qKT ? bvo[t][b] : OOB_OFFSET), (int)so, 0);
                        rbP[st__][u][t][b] = __builtin_bit_cast(f32x4, v_);
                    }
            }
            g_kt += 2;
        };
        load_patch(0);
        issue_bp(std::integral_constant<int, 2>{});
        store_patch(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int cb = 0; cb < ncb; ++cb) {
            const f32x4* const P = patch + (P_NBUF == 2 ? (cb & 1) : 0) * P_PATCH;
            load_patch(cb + 1);
            auto stage_p = [&](auto ST) {
                constexpr int st = decltype(ST)::value;
                constexpr int cur = st == 0 ? 2 : (st & 1), nxt = st == 8 ? 2 : ((st + 1) & 1);
                issue_bp(std::integral_constant<int, nxt>{});
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int ui = 2 * st + u, h = ui / 9, tap = ui - h * 9;
                    const int tr = MODE == 0 ? tap / 3 : 2 - tap / 3, ts = MODE == 0 ? tap % 3 : 2 - tap % 3;
                    bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
                    for (int t = 0; t < NSX; ++t) {
#pragma unroll
                        for (int a = 0; a < MI; ++a)
                            af[t][a] = __builtin_bit_cast(bf16x8, P[(t * 4 + 2 * h) * PPX + a_pix + (2 * a + tr) * PROW + ts]);
#pragma unroll
                        for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, rbP[cur][u][t][b]);
                    }
                    mfma_split<NSX, MI, NI, AT == 2>(af, bf, acc);
                }
            };
            stage_p(std::integral_constant<int, 0>{});
            stage_p(std::integral_constant<int, 1>{});
            stage_p(std::integral_constant<int, 2>{});
            stage_p(std::integral_constant<int, 3>{});
            stage_p(std::integral_constant<int, 4>{});
            stage_p(std::integral_constant<int, 5>{});
            stage_p(std::integral_constant<int, 6>{});
            stage_p(std::integral_constant<int, 7>{});
            stage_p(std::integral_constant<int, 8>{});
            if (cb + 1 < ncb) {
                if constexpr (P_NBUF == 1) {  // one buffer: everyone must be done reading it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // (two buffers: the other one was last read in block cb - 1, and every wave has passed the barrier that ended it)
                store_patch(P_NBUF == 2 ? ((cb + 1) & 1) : 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
    load_patch(0);
#pragma unroll
    for (int s_ = 0; s_ < P_NSTG - 1; ++s_) issue_b(s_);
    store_patch(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DBN_TRACE_MARK(1);  // (TRACE builds: tools/trace_probe16.py)
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
    DBN_TRACE_MARK(2);
    }
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
    constexpr int A_I = NP * (BM / 32), B_I = DB16 ? 0 : NSX * 2 * (BN / 64), U_I = A_I + B_I;
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
    static_assert(PA + (DB16 ? 0 : (B_PER_UNIT ? DMA_SU * RB : RB)) == PW, "slot count");
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
                          : (B_PER_UNIT ? wave_u * RB + (i - PA) % (RB > 0 ? RB : 1) : (wave_u & 1) * RB + (i - PA));
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
    int g_left, g_kt, g_r, g_s, g_cb, g_level = (MODE == 3 ? p.first_level : 0);  // (pyramid conv: levels [first_level, 4); round 5 also in the 16-bit loop)
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
        if (!DB16)  // (DB16: the weight side has its own walker, level_b — it runs behind this one)
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
        for (int g = g_level; g < 4; ++g) {
            const int f = 1 << g, kk = f + 2;
            const int ph = ((q.oh0 & (f - 1)) + 1) & (f - 1), pw = ((q.ow0 & (f - 1)) + 1) & (f - 1);
            nstages += (taps_of_class(kk, ph, f) * taps_of_class(kk, pw, f) * (p.Cs >> 4) + DMA_SU - 1) / DMA_SU;
        }
        level_dma(g_level);
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
            const int uc = isA ? i / RA : (i - PA) / (RB > 0 ? RB : 1);  // the unit of a per-unit slot
            const int u = fixed ? uc : uw;
            const bool v_u = fixed ? uv[uc] : (uw ? uv[DMA_SU - 1] : uv[0]);
            const int tr = fixed ? ur[uc] : (uw ? ur[DMA_SU - 1] : ur[0]), ts = fixed ? us[uc] : (uw ? us[DMA_SU - 1] : us[0]);
            const int tcb = fixed ? ucb[uc] : (uw ? ucb[DMA_SU - 1] : ucb[0]), tkt = fixed ? ukt[uc] : (uw ? ukt[DMA_SU - 1] : ukt[0]);
            auto* dst = (__attribute__((address_space(3))) void*)(smem + slot * DMA_STAGE + u * DMA_UNIT + i_lds[i]);
            if (isA) {
                if constexpr ((DBN_DBG16 & 1) != 0) {  // profile by deletion (timing only): no tap / bounds / offset arithmetic per DMA instruction
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, (int)(i_add[i] + (unsigned)r_nb[i] * 2u + (unsigned)u * 4096u), 0, 0, 0);
                    continue;
                }
                if constexpr ((DBN_DBG16 & 8) != 0) continue;  // no A DMA at all
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
    if constexpr (DB16) {
        // B fragments of a stage's two units: lane (li, lh) of wave column wn takes slice t*2 + lh of column wn*TN + b*32 + li
        constexpr int NB = DMA_SU * NSX * NI;  // register loads per stage and wave
        unsigned bvo[NSX][NI];
#pragma unroll
        for (int t = 0; t < NSX; ++t)
#pragma unroll
            for (int b = 0; b < NI; ++b) bvo[t][b] = (unsigned)((t * 2 + lh) * p.Cd + n0 + wn * TN + b * 32 + li) * 16u;
        f32x4 rbB[3][DMA_SU][NSX][NI];  // round 6: THREE sets — a stage's weight fragments are fetched two stages ahead (see the loop below)
        int bkt = kt_begin;  // weight k-tile of the next unit to fetch (clamped: units past the end multiply zeros of A)
        int bkt_max = max(kt_end - 1, kt_begin);
        // MODE 3: the weight side walks the pyramid levels on its own (it runs DMA_NSTG - 2 stages behind the A side's level_dma): level g's
        // panel starts at this class's row offset, has b_left k-tiles, and — like the A side — ends on a stage boundary (the odd unit past a
        // level's end re-reads its last k-tile against zeros of A)
        int b_level = MODE == 3 ? p.first_level : 0, b_left = 0;
        auto level_b = [&](int g) {
            const int f = 1 << g, kk = f + 2;
            const int oh0g = q.oh0 & (f - 1), ow0g = q.ow0 & (f - 1);
            const int ph = (oh0g + 1) & (f - 1), pw = (ow0g + 1) & (f - 1);
            const int kt_g = (taps_of_class(kk, ph, f) * taps_of_class(kk, pw, f) * p.Cs) >> 4;
            long krows = 0;
            for (int d = 0; d < ph * f + pw; ++d) krows += taps_of_class(kk, d >> g, f) * taps_of_class(kk, d & (f - 1), f) * p.Cs;
            rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg_wpk[g] + krows * p.Cd * NSX / 2), 0, (unsigned)kt_g * bstep_bytes,
                                                      0x00020000);
            b_left = kt_g;
            bkt = 0;
            bkt_max = max(kt_g - 1, 0);
        };
        if (MODE == 3) level_b(b_level);
        auto issue_bs = [&](auto SET) {
            constexpr int st__ = decltype(SET)::value;
            if (MODE == 3 && b_left <= 0 && b_level < 3) level_b(++b_level);
#pragma unroll
            for (int u = 0; u < DMA_SU; ++u) {
                const unsigned so = (unsigned)min(bkt, bkt_max) * bstep_bytes;
                ++bkt;
                --b_left;
#pragma unroll
                for (int t = 0; t < NSX; ++t)
#pragma unroll
                    for (int b = 0; b < NI; ++b) {
                        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                        if constexpr ((DBN_DBG16 & 2) != 0) {  // (deletion: no weight-fragment loads)
                            rbB[st__][u][t][b] = f32x4{1.f, 2.f, 3.f, (float)so};
                            continue;
                        }
                        const u32x4_ v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bvo[t][b], (int)so, 0);
                        rbB[st__][u][t][b] = __builtin_bit_cast(f32x4, v_);
                    }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        // prologue: the first DMA_NSTG - 1 stages of A, then the first TWO B sets (no dummy register loads: the compiler deletes loads
        // whose results are overwritten, and the counted waits below must match what is really in flight).
        // Round 6: the weight fragments of stage s are fetched at stage s - 2 (three register sets, the loop unrolled by three).  Profile
        // by deletion (-DDBN_DBG16=2, configs[4]): without the fragment loads the pyramid launch is a third shorter — fetched ONE stage
        // ahead (rounds 3-5) a set had one stage time, about a microsecond, to come back from L2, and every stage ended up waiting for it.
#pragma unroll
        for (int s_ = 0; s_ < DMA_NSTG - 1; ++s_) issue_stage(s_);
        asm volatile("" ::: "memory");
        issue_bs(S0{});
        issue_bs(S1{});
        asm volatile("" ::: "memory");
        int slot = 0;
        // Issue order: A0 A1 A2 B0 B1 | stage 0: B2 A3 | stage 1: B3 A4 | ...  — the register loads of an iteration go out BEFORE its DMA
        // instructions: vmcnt retires in order, so a B set fetched behind a stage of A made the stage that consumes the set wait for that
        // (much younger) A stage as well — the ring's depth was one stage in effect.  Younger than stage s's A DMA (and so allowed to be
        // outstanding when the stage starts) in the steady state: B_s, A_{s+1}, B_{s+1}, A_{s+2} = DMA_NSTG - 2 A stages and two B sets;
        // the first stages have more behind them (the prologue's order), for which the same count is merely stricter than needed.
        auto body = [&](auto SET, auto EARLY) {
            constexpr int set = decltype(SET)::value;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(((DBN_DBG16 & 8) ? 0 : (DMA_NSTG - 2) * PW) + ((DBN_DBG16 & 2) ? 0 : NB * 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_bs(std::integral_constant<int, (set + 2) % 3>{});
            asm volatile("" ::: "memory");
            const int fill = slot == 0 ? DMA_NSTG - 1 : slot - 1;
            issue_stage(fill);
            asm volatile("" ::: "memory");
            const f32x4* Sg = smem + slot * DMA_STAGE;
#pragma unroll
            for (int u = 0; u < DMA_SU; ++u) {
                const f32x4* As = Sg + u * DMA_UNIT;
                bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
                for (int t = 0; t < NSX; ++t) {
#pragma unroll
                    for (int a = 0; a < MI; ++a)
                        af[t][a] = (DBN_DBG16 & 4) ? __builtin_bit_cast(bf16x8, rbB[set][u][t][a % NI]) : __builtin_bit_cast(bf16x8, As[(t * BM + wm * TM + a * 32 + li) * 2 + lh]);
#pragma unroll
                    for (int b = 0; b < NI; ++b) bf[t][b] = __builtin_bit_cast(bf16x8, rbB[set][u][t][b]);
                }
                mfma_split<NSX, MI, NI, AT == 2>(af, bf, acc);
            }
            slot = slot + 1 == DMA_NSTG ? 0 : slot + 1;
        };
        static_assert(DMA_NSTG == 4, "the A-only ring of the register-fed form has four stages (six measured equal in round 5 and on the wide tile)");
        using E0 = std::integral_constant<int, 0>;
        using E1 = std::integral_constant<int, 1>;
        int st_ = 0;
        if (nstages > 0) body(S0{}, E0{});
        if (nstages > 1) body(S1{}, E1{});
        for (st_ = 2; st_ + 3 <= nstages; st_ += 3) {
            body(S2{}, E1{});
            body(S0{}, E1{});
            body(S1{}, E1{});
        }
        if (st_ < nstages) {
            body(S2{}, E1{});
            if (st_ + 1 < nstages) body(S0{}, E1{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
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
    }
    } else {
    for (int level = (MODE == 3 ? p.first_level : 0); level < (MODE == 3 ? 4 : 1); ++level) {
    if (MODE == 3) {
        level_setup(level);
        kt_end = qKT;
        kend = qK;
    }
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    auto offsets_of_interval = [&]() {
#pragma unroll
        for (int u = 0; u < KU; ++u) next_offsets(u);
    };
    offsets_of_interval();
    if constexpr (DIRECTB) issue_b_frag(C0{});  // weight k-tile 0
    issue_loads(C0{});
    offsets_of_interval();  // offsets of interval 1
    issue_loads(C1{});
    offsets_of_interval();  // offsets of interval 2
    stage(0, C0{});
    // every prologue load has landed before the loop is entered (a real S_WAITCNT, which the compiler's wait-count pass tracks):
    // entered with interval 1's loads still pending, the loop header carried `s_waitcnt vmcnt(2) / (1) / (0)` in front of its
    // fragment reads — the entry state merged into the back edge — and drained the two-tile prefetch in EVERY iteration.
    // (vmcnt = 0, expcnt / lgkmcnt untouched)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (NS == 0 && MODE != 3) DBN_TRACE_MARK(1);
    if (p.phase_prio) __builtin_amdgcn_s_setprio(0);

    auto k_step = [&](int kt, auto PAR) {
        constexpr int buf = decltype(PAR)::value;  // parity of the interval: LDS buffer and register set of its tiles
        if constexpr (DIRECTB) issue_b_frag(std::integral_constant<int, buf ^ 1>{});  // weight k-tile +1 (older than the A loads below: its wait leaves them in flight)
        issue_loads(PAR);  // interval +2 into the register set this interval was staged from
        // ... and they stay HERE: left alone the scheduler sinks them below the first MFMAs (it reuses the set's registers for the
        // fragment reads first), which shortens the prefetch distance from two k-steps to about one
        if constexpr (NS == 0) __builtin_amdgcn_sched_barrier(0);
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
                for (int a = 0; a < MI; ++a) af[s2][a] = (DBN_DBG & 8) ? f32x4{(float)kt, 1.f, 2.f, 3.f} : As[(2 * s2 + lh) * AS + wm * TM + a * 32 + li];
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    if constexpr (DIRECTB) bf[s2][b] = rbf_[buf][s2][b];
                    else bf[s2][b] = (DBN_DBG & 8) ? f32x4{1.f, (float)kt, 2.f, 3.f} : Bs[(2 * s2 + lh) * BS + wn * TN + b * 32 + li];
                }
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
                if (s2 == 0 && !(DBN_DBG & 32)) next_offsets();  // address math of tile kt+2 in the shadow of the MFMAs
            }
        } else {
            // one 32x32x16 bf16 MFMA k-step per k-tile: lane half lh owns k = 8*lh .. 8*lh+7
            bf16x8 af[NSX][MI], bf[NSX][NI];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
                for (int a = 0; a < MI; ++a) af[t][a] = __builtin_bit_cast(bf16x8, As[(t * 2 + lh) * AS + wm * TM + a * 32 + li]);
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    if constexpr (DIRECTB) bf[t][b] = __builtin_bit_cast(bf16x8, rbf_[buf][t][b]);
                    else bf[t][b] = __builtin_bit_cast(bf16x8, Bs[(t * 2 + lh) * BS + wn * TN + b * 32 + li]);
                }
            }
            mfma_split<NS, MI, NI, AT == 2>(af, bf, acc);
            next_offsets();
        }
        stage(buf ^ 1, std::integral_constant<int, buf ^ 1>{});
        if (!(DBN_DBG & 16)) __syncthreads();
    };
    // whole pairs of k-steps, then the odd one: with `if (more) k_step(C1)` INSIDE the loop there is a static path around the
    // second step on which the first step's loads reach the loop header unstaged, and the compiler guarded the header's fragment
    // reads (which reuse those registers) with s_waitcnt vmcnt(2) / (1) / (0) — draining the two-tile prefetch every iteration
    int kt = kt_begin;
    for (; kt + 2 * KU <= kt_end; kt += 2 * KU) {
        k_step(kt, C0{});
        k_step(kt + KU, C1{});
    }
    if (kt < kt_end) k_step(kt, C0{});
    if (NS == 0 && MODE != 3) DBN_TRACE_MARK(2);
    if (p.phase_prio && level + 1 == (MODE == 3 ? 4 : 1)) __builtin_amdgcn_s_setprio(3);
    }

    }
    // ---- accumulate mode: fold the previous contents of dst into the accumulators first, so that the BatchNorm
    // statistics below and the store loop both see the final values
    // split-K launches write fp32 slabs whatever the activation type (the slab sum rounds once)
    const bool to_slab = MODE < 2 && p.ksplit > 1;
    float* const slabp = reinterpret_cast<float*>(p.dst) + (to_slab ? (long)blockIdx.y * ((long)qM * p.Cd + 1088) : 0L);
    void* const dstv = p.dst;
    // the tensor `accumulate` adds: dst itself, or — inference epilogue — a residual input of the same shape and storage type (IgemmParams::res)
    const void* const accv = (p.res && !to_slab) ? p.res : p.dst;
    const float* const accf = (p.res && !to_slab) ? reinterpret_cast<const float*>(p.res) : slabp;
    auto ld_dst = [&](long off) -> float { return (DST_F32 || to_slab) ? accf[off] : dbn_ld1 < DST_F32 ? 0 : AT > (accv, off); };
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
                if (ok) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(accv) + doff + n0 + piece * 8);
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
            if constexpr (BUFST) {
                // fp32 destination, linear rows or the pixel patch: raw buffer loads at the store loop's offsets (rows past M are out of
                // range and read as 0), four rows in flight
                const __amdgpu_buffer_rsrc_t rsrcA =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(accf), 0, (unsigned)((long)(PATCH ? p.N * p.Hdf * p.Wdf : qM) * p.Cd * 4), 0x00020000);
                const unsigned pitch = (unsigned)p.Cd * 4u, colb = (unsigned)(n0 + wn * TN + li) * 4u;
                const unsigned row0 = PATCH ? (unsigned)((pn * p.Hdf + ph0 + 2 * (wm * MI + a)) * p.Wdf + pw0) : (unsigned)(m0 + wm * TM + a * 32 + 4 * lh);
                const unsigned base0 = (row0 + (unsigned)((PATCH && lh) ? p.Wdf : 0)) * pitch + colb;
                const unsigned base1 = PATCH ? (row0 + (unsigned)(lh ? 0 : p.Wdf)) * pitch + colb : base0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float old[4][NI];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const unsigned off = PATCH ? ((__builtin_popcount(r >> 2) & 1) ? base1 : base0) + (unsigned)((r >> 2) * 4 + (r & 3)) * pitch
                                                   : base0 + (unsigned)((r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            old[i][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcA, (int)off + b * 128, 0, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int b = 0; b < NI; ++b) acc[a][b][4 * g + i] += old[i][b];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (MODE < 2) {
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
            dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (0L * p.Cd + c) * p.stat_rows + trow, piv[cl]);
            dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (1L * p.Cd + c) * p.stat_rows + trow, s1);
            dbn_stat_put(p.bnf_cnt != nullptr, p.stats + (2L * p.Cd + c) * p.stat_rows + trow, s2);
        }
        // (the count row: tile column 0 writes it; with the in-kernel finalize EVERY column does — the same value — because each column
        // folds its own channels on its own counters and must not read a count that column 0's workgroup has yet to write)
        if ((nt == 0 || p.bnf_cnt) && tid == 0) dbn_stat_put(p.bnf_cnt != nullptr, p.stats + 3L * p.Cd * p.stat_rows + trow, (float)min(BM, qM - m0));
    }
    // optional in-kernel finalize of those rows (IgemmParams::bnf_cnt): the last workgroup of every 64 rows folds them, the last of those the
    // groups.  Called at the END of the workgroup, behind its output stores (first build: right here — every workgroup then sat through
    // an s_waitcnt vmcnt(0) and an atomic round trip before it stored its tile: 716 -> 700 images/s)
    static_assert((long)sizeof(smem) >= 3L * NT * 8 + 16, "LDS scratch of the statistics finalize");
    auto fin_stats = [&]() {
        if (p.stats && p.bnf_cnt) {
            __syncthreads();  // (the LDS staging of the output tile is dead)
            dbn_bn_stats_finish(DBN_BNF_ARGS(p), p.stat_row0 + q_row_base + mt, nt, n0, BN, reinterpret_cast<int*>(smem) + (sizeof(smem) / 4 - 4),
                                reinterpret_cast<double*>(smem));
        }
    };

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
    if (p.relu) {  // inference epilogue (eval-mode BatchNorm folded into the panel, bias = its shift): max(acc + bias, 0), then nothing left to add
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = fmaxf(acc[a][b][r] + bv[b], 0.f);
#pragma unroll
        for (int b = 0; b < NI; ++b) bv[b] = 0.f;
    }
    const long dcol = n0 + wn * TN + li;
    // EPI = 1, optional: in-kernel finalize of the BatchNorm-backward sums (IgemmParams::bnb_cnt).  Called by every thread of the
    // workgroup after it wrote its partial row `trow_`.
    // Cross-workgroup hand-over WITHOUT device-scope fences: a release fence makes every wave write its L2 back (gfx950's L2s
    // are per XCD and not coherent with each other) — measured: the data gradients 3.3x slower with a __threadfence() per tile.
    // Instead the few floats that cross workgroups are moved with agent-scope relaxed atomics (sc1 stores / loads: they go
    // to / come from the memory side, never a stale line of this XCD's L2), a plain s_waitcnt vmcnt(0) lets the stores complete
    // before the (agent-scope, integer) counter goes up, and a consumer only reads after it has seen the count.
    auto xst = [](float* ptr, float v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto xld = [](const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto bnb_finish = [&](int trow_) {
        if constexpr (EPI == 1) {
            if (!p.bnb_cnt) return;
            constexpr int G = 64;
            const int NG = (p.stat_rows + G - 1) / G, g = trow_ / G;
            const int nbn = p.bnb_y2 ? 2 : 1;
            int* const cnt = p.bnb_cnt + nt * (NG + 1);
            int* const s_flag = reinterpret_cast<int*>(smem) + (sizeof(smem) / 4 - 4);  // (behind every scratch region of the epilogue)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial-row stores have completed
            __syncthreads();
            DBN_RACE_JITTER();
            if (tid == 0) {
                const int gsize = min(G, p.stat_rows - g * G);
                const int last = __hip_atomic_fetch_add(cnt + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
                if (last) __hip_atomic_store(cnt + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (nobody touches it again in this call)
                *s_flag = last;
            }
            __syncthreads();
            if (!*s_flag) return;
            asm volatile("" ::: "memory");
            // fold this group's rows for the BN channels of this tile column: item = (BatchNorm, sum, channel)
            const int r0 = g * G, r1_ = min(p.stat_rows, r0 + G);
            for (int it = tid; it < nbn * 2 * BN; it += NT) {
                const int b = it / (2 * BN), ks = (it / BN) & 1, cl = it % BN;
                const float* src = (b ? p.bnb_part2 : p.bnb_part) + ((long)ks * p.Cd + n0 + cl) * p.stat_rows;
                double s = 0.0;
                int r = r0;
                for (; r + 3 < r1_; r += 4) s += ((double)xld(src + r) + (double)xld(src + r + 1)) + ((double)xld(src + r + 2) + (double)xld(src + r + 3));
                for (; r < r1_; ++r) s += (double)xld(src + r);
                xst(p.bnb_grp + (((long)b * 2 + ks) * p.Cd + n0 + cl) * NG + g, (float)s);
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
            if (!*s_flag) return;
            asm volatile("" ::: "memory");
            for (int it = tid; it < nbn * BN; it += NT) {
                const int b = it / BN, cl = it % BN;
                const long c = n0 + cl;
                const float* g1 = p.bnb_grp + (((long)b * 2 + 0) * p.Cd + c) * NG;
                const float* g2 = p.bnb_grp + (((long)b * 2 + 1) * p.Cd + c) * NG;
                double s1_ = 0.0, s2_ = 0.0;
                for (int q_ = 0; q_ < NG; ++q_) {
                    s1_ += (double)xld(g1 + q_);
                    s2_ += (double)xld(g2 + q_);
                }
                p.bnb_dbeta[b][c] = (float)(s1_ * p.bnb_gscale);
                p.bnb_dgamma[b][c] = (float)(s2_ * p.bnb_gscale);
                p.bnb_c1c2[b][c] = (float)(s1_ * p.bnb_invM);
                p.bnb_c1c2[b][p.Cd + c] = (float)(s2_ * p.bnb_invM);
            }
        }
    };
    // ---- EPI = 1: store + the sums of the BatchNorm backward that consumes dst (IgemmParams::bnb_part), ROW-MAJOR through LDS.
    // The accumulators hold the final dz values.  Reading y (and the mask tensor) at the accumulator layout — a lane owns one
    // column of 16 rows — is MI*NI*16 four-byte loads per lane and tensor, with every accumulator copied out of the AGPRs and
    // live beside them (measured: 138-246 registers, 128x128 tiles at two waves per SIMD, the data gradients 13 % slower).
    // Instead the tile goes to LDS once (conflict-free ds_write_b32: a half-wave writes 32 consecutive floats of a row), and a
    // second pass walks it row-major: a thread owns one channel quad, reads dz from LDS and y / mask from HBM as 16-byte pieces
    // (a row's BN*4 contiguous bytes per BN/4 lanes), accumulates its quad's two sums over the rows it visits, and stores dz with
    // 16-byte stores.  Tiles larger than the LDS panels (128x128, 256x64) take two passes (accumulator blocks a < MI/2, then the rest).
    if constexpr (EPI == 1 && DST_F32) {
        constexpr int PITCH = BN;                            // floats per LDS row (the b32 writes and b128 reads below are conflict-free unpadded)
        constexpr int HALVES = ((long)BM * PITCH * 4 > (long)sizeof(smem)) ? 2 : 1;
        constexpr int MI_H = MI / HALVES, RH = BM / HALVES;  // accumulator blocks / tile rows per pass
        constexpr int LPR = BN / 4, RSTEP = NT / LPR;        // lanes per row, rows per sweep of the workgroup
        static_assert(MI % HALVES == 0 && (long)RH * PITCH * 4 <= (long)sizeof(smem) && RH % RSTEP == 0 && NT % LPR == 0, "tile / LDS");
        static_assert(2L * RSTEP * BN * 4 <= (long)sizeof(smem), "reduction scratch");
        float* const T = reinterpret_cast<float*>(smem);
        const float* const yb = reinterpret_cast<const float*>(p.bnb_y);
        const float* const zb = reinterpret_cast<const float*>(p.bnb_zmask);
        float* const dstf = reinterpret_cast<float*>(p.dst);
        const int c4 = tid % LPR, rq = tid / LPR;
        const int cg = n0 + 4 * c4;  // first of this thread's four channels
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bnb_mean + cg), rs = *reinterpret_cast<const f32x4*>(p.bnb_rstd + cg);
        f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = msc;
        if (!zb) {
            msc = *reinterpret_cast<const f32x4*>(p.bnb_msc + cg);
            msh = *reinterpret_cast<const f32x4*>(p.bnb_msh + cg);
        }
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1, s4 = s1;
        const float* const y2b = reinterpret_cast<const float*>(p.bnb_y2);
        f32x4 mu2 = s1, rs2 = s1;
        if (y2b) {
            mu2 = *reinterpret_cast<const f32x4*>(p.bnb_mean2 + cg);
            rs2 = *reinterpret_cast<const f32x4*>(p.bnb_rstd2 + cg);
        }
        auto sweep = [&](int h, auto ZM, auto TWO) {
            constexpr bool kZ = decltype(ZM)::value, k2 = decltype(TWO)::value;
            constexpr int UN = 2;  // rows in flight per thread
            static_assert((RH / RSTEP) % UN == 0, "whole groups");
#pragma unroll 1
            for (int j0 = rq; j0 < RH; j0 += UN * RSTEP) {
                f32x4 yv[UN], zv[UN], y2v[UN];
                long doff[UN];
                bool ok[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int j = j0 + u * RSTEP;                          // LDS row
                    const int w_ = j / (32 * MI_H), rem = j - w_ * (32 * MI_H);
                    const int trow = w_ * TM + h * (32 * MI_H) + rem;      // tile row (rows of block a are 32 apart per wave row)
                    tile_row(trow, ok[u], doff[u]);
                    const long o = ok[u] ? doff[u] + cg : (long)cg;        // rows past M read a valid pixel, contribute nothing
                    yv[u] = *reinterpret_cast<const f32x4*>(yb + o);
                    if constexpr (kZ) zv[u] = *reinterpret_cast<const f32x4*>(zb + o);
                    if constexpr (k2) y2v[u] = *reinterpret_cast<const f32x4*>(y2b + o);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int j = j0 + u * RSTEP;
                    const f32x4 dz = *reinterpret_cast<const f32x4*>(T + j * PITCH + 4 * c4);
                    if (ok[u]) *reinterpret_cast<f32x4*>(dstf + doff[u] + cg) = dz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m_;
                        if constexpr (kZ) m_ = zv[u][e];
                        else m_ = dbn_affine(yv[u][e], msc[e], msh[e]);
                        const float g = (ok[u] & (m_ > 0.f)) ? dz[e] : 0.f;
                        s1[e] += g;
                        s2[e] += g * ((yv[u][e] - mu[e]) * rs[e]);
                        if constexpr (k2) s4[e] += g * ((y2v[u][e] - mu2[e]) * rs2[e]);
                    }
                }
            }
        };
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            __syncthreads();  // the panels / the previous pass are dead
#pragma unroll
            for (int al = 0; al < MI_H; ++al)
#pragma unroll
                for (int b = 0; b < NI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = wm * (32 * MI_H) + al * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        T[j * PITCH + wn * TN + b * 32 + li] = acc[h * MI_H + al][b][r] + bv[b];
                    }
            __syncthreads();
            if (zb && y2b) sweep(h, std::true_type{}, std::true_type{});
            else if (zb) sweep(h, std::true_type{}, std::false_type{});
            else sweep(h, std::false_type{}, std::false_type{});
        }
        s3 = s1;  // (the second BatchNorm's first sum is the same masked gradient sum)
        // fold the RSTEP row groups of each channel (fixed order) and write this tile's partial row
        const int trow_ = p.stat_row0 + q_row_base + mt;
        float* const r1 = T;               // [RSTEP][BN]
        float* const r2 = T + RSTEP * BN;  // [RSTEP][BN]
        auto fold = [&](const f32x4& a_, const f32x4& b_, float* part) {
            __syncthreads();
            *reinterpret_cast<f32x4*>(r1 + rq * BN + 4 * c4) = a_;
            *reinterpret_cast<f32x4*>(r2 + rq * BN + 4 * c4) = b_;
            __syncthreads();
            for (int cl = tid; cl < BN; cl += NT) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < RSTEP; ++w) {
                    t1 += r1[w * BN + cl];
                    t2 += r2[w * BN + cl];
                }
                const long c = n0 + cl;
                if (p.bnb_cnt) {  // read by another workgroup of this launch: memory-side store (see bnb_finish)
                    xst(part + (0L * p.Cd + c) * p.stat_rows + trow_, t1);
                    xst(part + (1L * p.Cd + c) * p.stat_rows + trow_, t2);
                } else {
                    part[(0L * p.Cd + c) * p.stat_rows + trow_] = t1;
                    part[(1L * p.Cd + c) * p.stat_rows + trow_] = t2;
                }
            }
        };
        fold(s1, s2, p.bnb_part);
        if (y2b) fold(s3, s4, p.bnb_part2);
        DBN_TRACE_MARK(3);
        bnb_finish(trow_);
        return;
    }
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
            if constexpr (EPI == 1) {
                // + the sums of the BatchNorm backward that consumes dst (IgemmParams::bnb_part), taken in the same row-major sweep
                // over the values AS STORED (rounded to the storage type: what the apply pass will read back).  A thread owns FOUR
                // channels here (8-byte pieces: with eight, the per-channel constants and sums alone were 70 registers and the
                // kernel dropped from four to two waves per SIMD); y / the mask tensor come in the same way.
                static_assert(AT == 1, "sums epilogue of the 16-bit path: bf16 storage (training)");
                constexpr int LPR4 = BN / 4, RPP4 = NT / LPR4, PASSES = BM / RPP4;
                static_assert(NT % LPR4 == 0 && BM % RPP4 == 0 && PASSES % 2 == 0, "whole passes");
                const int piece4 = tid % LPR4, rq = tid / LPR4;
                const unsigned short* const yb = reinterpret_cast<const unsigned short*>(p.bnb_y);
                const unsigned short* const zb = reinterpret_cast<const unsigned short*>(p.bnb_zmask);
                const unsigned short* const y2b = reinterpret_cast<const unsigned short*>(p.bnb_y2);
                const int cg = n0 + piece4 * 4;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, s4[4] = {0.f, 0.f, 0.f, 0.f};
                auto bf = [](const u32x2& w, int e) { return __builtin_bit_cast(float, (e & 1) ? (w[e >> 1] & 0xFFFF0000u) : (w[e >> 1] << 16)); };
                auto sweep16 = [&](auto ZM, auto TWO) {
                    constexpr bool kZ = decltype(ZM)::value, k2 = decltype(TWO)::value;
                    f32x4 mu = *reinterpret_cast<const f32x4*>(p.bnb_mean + cg), rs = *reinterpret_cast<const f32x4*>(p.bnb_rstd + cg);
                    f32x4 msc = mu, msh = mu, mu2 = mu, rs2 = mu;
                    if constexpr (!kZ) {
                        msc = *reinterpret_cast<const f32x4*>(p.bnb_msc + cg);
                        msh = *reinterpret_cast<const f32x4*>(p.bnb_msh + cg);
                    }
                    if constexpr (k2) {
                        mu2 = *reinterpret_cast<const f32x4*>(p.bnb_mean2 + cg);
                        rs2 = *reinterpret_cast<const f32x4*>(p.bnb_rstd2 + cg);
                    }
                    constexpr int UN = 2;  // rows in flight per thread
#pragma unroll 1
                    for (int ps0 = 0; ps0 < PASSES; ps0 += UN) {
                        u32x2 v[UN], yv[UN], zv[UN], y2v[UN];
                        bool ok[UN];
                        long doff[UN];
#pragma unroll
                        for (int u = 0; u < UN; ++u) {
                            const int row = (ps0 + u) * RPP4 + rq;
                            v[u] = *reinterpret_cast<const u32x2*>(T + row * PITCH + piece4 * 4);
                            tile_row(row, ok[u], doff[u]);
                            const long o = ok[u] ? doff[u] + cg : (long)cg;  // rows past M read a valid pixel, contribute nothing
                            yv[u] = *reinterpret_cast<const u32x2*>(yb + o);
                            if constexpr (kZ) zv[u] = *reinterpret_cast<const u32x2*>(zb + o);
                            if constexpr (k2) y2v[u] = *reinterpret_cast<const u32x2*>(y2b + o);
                        }
#pragma unroll
                        for (int u = 0; u < UN; ++u) {
                            if (ok[u]) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(dstv) + doff[u] + cg) = v[u];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float y_ = bf(yv[u], e);
                                float m_;
                                if constexpr (kZ) m_ = bf(zv[u], e);
                                else m_ = dbn_affine(y_, msc[e], msh[e]);
                                const float g = (ok[u] & (m_ > 0.f)) ? bf(v[u], e) : 0.f;
                                s1[e] += g;
                                s2[e] += g * ((y_ - mu[e]) * rs[e]);
                                if constexpr (k2) s4[e] += g * ((bf(y2v[u], e) - mu2[e]) * rs2[e]);
                            }
                        }
                    }
                };
                if (zb && y2b) sweep16(std::true_type{}, std::true_type{});
                else if (zb) sweep16(std::true_type{}, std::false_type{});
                else sweep16(std::false_type{}, std::false_type{});
                // fold the RPP4 row groups of each channel in fixed order, write this tile's partial row
                static_assert(2L * RPP4 * BN * 4 <= (long)sizeof(smem), "reduction scratch");
                float* const r1 = reinterpret_cast<float*>(smem);  // [RPP4][BN]
                float* const r2 = r1 + RPP4 * BN;                   // [RPP4][BN]
                const int trow_ = p.stat_row0 + q_row_base + mt;
                auto fold16 = [&](const float (&a_)[4], const float (&b_)[4], float* part) {
                    __syncthreads();
                    *reinterpret_cast<f32x4*>(r1 + rq * BN + piece4 * 4) = f32x4{a_[0], a_[1], a_[2], a_[3]};
                    *reinterpret_cast<f32x4*>(r2 + rq * BN + piece4 * 4) = f32x4{b_[0], b_[1], b_[2], b_[3]};
                    __syncthreads();
                    for (int cl = tid; cl < BN; cl += NT) {
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int w = 0; w < RPP4; ++w) {
                            t1 += r1[w * BN + cl];
                            t2 += r2[w * BN + cl];
                        }
                        const long c = n0 + cl;
                        if (p.bnb_cnt) {
                            xst(part + (0L * p.Cd + c) * p.stat_rows + trow_, t1);
                            xst(part + (1L * p.Cd + c) * p.stat_rows + trow_, t2);
                        } else {
                            part[(0L * p.Cd + c) * p.stat_rows + trow_] = t1;
                            part[(1L * p.Cd + c) * p.stat_rows + trow_] = t2;
                        }
                    }
                };
                fold16(s1, s2, p.bnb_part);
                if (y2b) fold16(s1, s4, p.bnb_part2);
                bnb_finish(trow_);
                return;
            }
#pragma unroll
            for (int ps = 0; ps < BM / RPP; ++ps) {
                const int row = ps * RPP + tid / LPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * PITCH + piece * 8);
                bool ok;
                long doff;
                tile_row(row, ok, doff);
                if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned short*>(dstv) + doff + n0 + piece * 8) = v;
            }
            fin_stats();
            return;
        }
    }
    if (DBN_DBG & 64) {  // (profile by deletion: no output stores — one element per lane keeps the accumulators alive)
        float s_ = 0.f;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_ += acc[a][b][r];
        if (s_ == 12345.678f) st_dst(dcol, s_);
        return;
    }
    if constexpr (BUFST) {
        // fp32 destination whose tile rows are a linear walk (MODE 0 / 1) or the pixel patch: raw BUFFER stores.  A row's byte offset is
        // a per-lane base + a wave-uniform multiple of the row pitch (one v_add per row); rows past M fall outside the resource's range
        // and are dropped by the hardware — no compare / saveexec / 64-bit address arithmetic / branch per row.  The predicated form
        // below is ~40 instructions per row; per-workgroup timestamps (tools/trace_probe.py) showed that a workgroup's instructions
        // outside its MFMAs are NOT hidden by the other residents of the CU — every one of them costs matrix time (the epilogue was
        // 750 of a 64 x 64 tile-wave's ~3100 instructions at K = 576).
        const __amdgpu_buffer_rsrc_t rsrcD =
            __builtin_amdgcn_make_buffer_rsrc(slabp, 0, (unsigned)((long)(PATCH ? p.N * p.Hdf * p.Wdf : qM) * p.Cd * 4), 0x00020000);
        const unsigned pitch = (unsigned)p.Cd * 4u;
        const unsigned colb = (unsigned)(n0 + wn * TN + li) * 4u;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            if constexpr (PATCH) {
                // row (r, lh) of block a is pixel y = ph0 + 2 (wm MI + a) + parity(popcount(r >> 2) + lh), x = pw0 + 4 (r >> 2) + (r & 3)
                const unsigned row0 = (unsigned)((pn * p.Hdf + ph0 + 2 * (wm * MI + a)) * p.Wdf + pw0);
                const unsigned base0 = (row0 + (unsigned)(lh ? p.Wdf : 0)) * pitch + colb;  // parity(popcount) == 0 -> y offset lh
                const unsigned base1 = (row0 + (unsigned)(lh ? 0 : p.Wdf)) * pitch + colb;  // parity == 1 -> y offset 1 - lh
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = ((__builtin_popcount(r >> 2) & 1) ? base1 : base0) + (unsigned)((r >> 2) * 4 + (r & 3)) * pitch;
#pragma unroll
                    for (int b = 0; b < NI; ++b)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[a][b][r] + bv[b]), rsrcD, (int)off + b * 128, 0, 0);
                }
            } else {
                const unsigned base = (unsigned)(m0 + wm * TM + a * 32 + 4 * lh) * pitch + colb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = base + (unsigned)((r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
                    for (int b = 0; b < NI; ++b)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[a][b][r] + bv[b]), rsrcD, (int)off + b * 128, 0, 0);
                }
            }
        }
    } else {
#pragma unroll
    for (int a = 0; a < MI; ++a)
        for_rows(a, [&](int r, bool ok, long doff) {
            if (ok) {
#pragma unroll
                for (int b = 0; b < NI; ++b) st_dst(dcol + doff + b * 32, acc[a][b][r] + bv[b]);
            }
        });
    }
    DBN_TRACE_MARK(3);
#if DBN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and once more when this thread's output stores have completed
    DBN_TRACE_MARK(4);
#endif
    fin_stats();
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
    p.stagger_units = p.stagger_blocks = 0;
    p.phase_prio = dbn_g_phase_prio;
    p.trace = (DBN_TRACE && dbn_g_trace && grid <= dbn_g_trace_blocks) ? dbn_g_trace : nullptr;
    if (dbn_g_stagger > 0 && NS == 0 && AT == 0 && mode < 2 && p.ksplit <= 1) {
        // exact fp32: a k-step is MI * NI * 8 MFMAs of 64 clocks per wave; the residents of a SIMD share its matrix pipe, so one slot
        // step = one workgroup's own loop time.  Only when the grid is more than one round of residents (otherwise nothing repeats).
        constexpr int MI_ = BM / WM / 32, NI_ = BN / WN / 32;
        const int slots = (BM * BN <= 4096) ? 7 : (BM * BN <= 8192) ? 4 : 3;  // resident workgroups per CU of these instantiations
        const long kt = ((long)p.R * p.S * p.Cs + 15) / 16;
        const long units = kt * MI_ * NI_ * 8 * 64 / 1024 * dbn_g_stagger / 1000;
        if (grid > 256 * slots && units > 0) {
            p.stagger_units = (int)std::min<long>(units, 4096);
            p.stagger_blocks = 256 * slots;
        }
    }
    const int gy = (mode < 2 && p.ksplit > 1) ? p.ksplit : 1;
    const bool epi_ok = gy == 1 && p.bnb_y && p.bnb_mean && p.bnb_rstd && (p.bnb_zmask || (p.bnb_msc && p.bnb_msh)) &&
                        (!p.bnb_y2 || (p.bnb_zmask && p.bnb_mean2 && p.bnb_rstd2 && p.bnb_part2));
    if constexpr (BM == 128 && WM == 2 && WN == 2 && (NS > 0 || BN == 64) && AT != 3) {  // (exact fp32: the 128 x 64 tile only)
        if (p.patch && mode < 2) {
            if constexpr (NS == 1 && (AT == 1 || AT == 2)) {  // round 6: the weight-resident kernel (wres16.hip) where the panel fits the registers
                if (gy == 1 && (!p.bnb_part || epi_ok) &&
                    dbn_wres16_eligible(AT, mode, p.N, p.Hdf, p.Wdf, p.Cs, p.Cd, p.bnb_part != nullptr, p.bnb_y2 != nullptr, p.stats != nullptr))
                    return dbn_launch_wres16(p, mode, AT, st);
            }
            if constexpr (AT == 0 || AT == 1) {
                if (p.bnb_part) {
                    if (!epi_ok) return DBN_ERR_ARG;
                    if (mode == 0)
                        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT, true, 1>), dim3(grid), dim3(256), 0, st, p);
                    else
                        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 1, NS, AT, true, 1>), dim3(grid), dim3(256), 0, st, p);
                    return dbn_status();
                }
            }
            if (p.bnb_part) return DBN_ERR_ARG;
            if (mode == 0)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT, true>), dim3(grid), dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 1, NS, AT, true>), dim3(grid), dim3(256), 0, st, p);
            return dbn_status();
        }
    }
    if (p.patch) return DBN_ERR_ARG;
    if ((p.Cs & 15) != 0) {  // the stem (Cs = 4): K walk across taps
        if constexpr (AT == 0) {
            if (mode != 0 || p.bnb_part || gy != 1) return DBN_ERR_ARG;
            hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT, false, 0, false>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
            return dbn_status();
        } else {
            return DBN_ERR_ARG;
        }
    }
    if (p.bnb_part) {  // with the sums of the BatchNorm backward that consumes dst (fp32 or bf16 storage)
        if constexpr (AT == 0 || AT == 1) {
            if (!epi_ok) return DBN_ERR_ARG;
            if (mode == 0)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 0, NS, AT, false, 1>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
            else if (mode == 1)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 1, NS, AT, false, 1>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
            else if (mode == 2)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, 2, NS, AT, false, 1>), dim3(grid), dim3(WM * WN * 64), 0, st, p);
            else
                return DBN_ERR_ARG;
            return dbn_status();
        } else {
            return DBN_ERR_ARG;
        }
    }
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

// Round 6: the 16-bit storage types on a 128 x 256 tile — for the pyramid form ONE column tile for the FPN output conv's 256 channels, so
// a workgroup's gathered A rows (the LDS-DMA ring's bytes) feed twice the MFMAs; 230 registers at two waves per SIMD
// (DBN_IGEMM_OCC), two workgroups per CU instead of three.  Same row tiles (BM = 128) as configuration 1, so the BatchNorm partial rows and
// every row count of the host side are unchanged; per output element the same products in the same order: bit-identical results.
// dbn_g_wide_tile (conv.hip, dbn_set_pyramid_wide): 0 off, 1 the pyramid form, 2 also plain forward / stride-1 data-gradient launches,
// 3 as 2 whatever the launch's size.
#ifndef DBN_WIDE_WM
#define DBN_WIDE_WM 1  // wave layout of the wide tile: 1 x 4 waves of 128 x 64 — no two waves load the same weight fragments (the register-fed B side is
                       // 28 GB per configs[4] launch with 2 x 2 waves of 64 x 128, half of it with 1 x 4): configs[4] 10.95 -> 10.74 ms per forward
#endif
template <int AT>
int launch_wide(IgemmParams& p, int mode, hipStream_t st) {
    constexpr int WM_ = DBN_WIDE_WM, WN_ = 4 / DBN_WIDE_WM;
    static_assert(AT == 1 || AT == 2, "16-bit storage");
    const int rows = mode == 3 ? 64 * dbn_ceil_div(p.N * (p.Hdf >> 3) * (p.Wdf >> 3), 128) : dbn_ceil_div(p.N * p.Hdf * p.Wdf, 128);
    int grid = rows * (p.Cd / 256);
    p.pyr_group = 0;
    if (mode == 3 && dbn_g_pyr_group > 0) {  // (tile order of the pyramid form: see the kernel; the last block of row tiles is padded)
        p.pyr_group = dbn_g_pyr_group;
        grid = dbn_ceil_div(rows / 64, p.pyr_group) * p.pyr_group * 64 * (p.Cd / 256);
    }
    if (p.stat_rows <= 0) p.stat_rows = rows;
    p.launch_rows = rows;
    if (grid == 0) return DBN_OK;
    p.stagger_units = p.stagger_blocks = 0;
    p.phase_prio = dbn_g_phase_prio;
    p.trace = nullptr;
    if (mode == 3) hipLaunchKernelGGL((igemm_f32_kernel<128, 256, WM_, WN_, 3, 1, AT>), dim3(grid), dim3(256), 0, st, p);
    else if (mode == 0) hipLaunchKernelGGL((igemm_f32_kernel<128, 256, WM_, WN_, 0, 1, AT>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<128, 256, WM_, WN_, 1, 1, AT>), dim3(grid), dim3(256), 0, st, p);
    return dbn_status();
}
static inline bool wide_tile_ok(const IgemmParams& p, int mode) {  // (the size rule: dbn_wide_tile_geom_ok, igemm_common.h)
    if (p.bnb_part || p.patch || p.ksplit > 1) return false;
    return dbn_wide_tile_geom_ok(mode, p.N, p.Hdf, p.Wdf, p.Cs, p.Cd);
}

#ifndef DBN_CFG1_WM
#define DBN_CFG1_WM 1  // the 128 x 128 tile of the 16-bit storage types as 1 x 4 waves of 128 x 32: no two waves load the same weight fragments from L2 (the
                       // register-fed B side of the generic loop; 2 = the 2 x 2 layout of rounds 3-5).  configs[4] 10.77 -> 10.64 ms, bf16 step 1753 / 1765 -> 1768
#endif
// the four tile configurations of one (NS, AT) family
template <int NS, int AT>
int launch_igemm_cfg(IgemmParams& p, int cfg, int mode, hipStream_t st) {
    if constexpr (NS == 1 && (AT == 1 || AT == 2)) {
        if (cfg == 1 && wide_tile_ok(p, mode)) return launch_wide<AT>(p, mode, st);
    }
    switch (cfg) {
        case 1:
            if constexpr (NS == 1 && (AT == 1 || AT == 2) && DBN_CFG1_WM == 1) return launch_igemm_ns<128, 128, 1, 4, NS, AT>(p, mode, st);
            else return launch_igemm_ns<128, 128, 2, 2, NS, AT>(p, mode, st);
        case 2: return launch_igemm_ns<256, 64, 4, 1, NS, AT>(p, mode, st);
        case 3: return launch_igemm_ns<128, 64, 2, 2, NS, AT>(p, mode, st);
        default: return launch_igemm_ns<64, 64, 2, 2, NS, AT>(p, mode, st);
    }
}

}  // namespace
