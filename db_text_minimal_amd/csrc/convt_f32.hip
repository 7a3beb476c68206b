// ConvTranspose2d(k = 2, stride 2, padding 0) forward in exact fp32 — the DB head's up-sampling layers
// (/root/reference/src/modules/segmentation_head.py:24-29,64-79; BASELINE configs[1]).
//
// As a GEMM the layer is four independent 1x1 problems (one per output parity class (a, b): y[2h+a, 2w+b] = W[:, :, a, b]^T x[h, w])
// with K = Cin — four k-steps of 16 for Cin = 64.  In the general parity-class kernel (igemm_f32_kernel MODE 2) a workgroup owns one
// class of one pixel tile, so loads -> 4 k-steps -> stores are serial phases of a short life: profile by deletion of that launch
// (tools/convt_deletion_probe.py, 64->64 at 160 -> 320, batch 16) gave 211 us as built, 147 us without the output stores, 149 us
// without the loads / staging, 107 us with both gone — they add up instead of overlapping, at algorithmic HBM traffic (534 MB).
// Here a workgroup keeps its 128 input pixels x Cin in LDS for its whole life and walks the four classes over them: the input is
// read once, the weight panel of the next class is fetched while the current one is multiplied, and the stores of class c drain
// while class c + 1 computes.  Optional BatchNorm statistics of the output (the layer is followed by a train-mode BatchNorm):
// one partial row per workgroup over its 4 x 128 output pixels, same record as the general kernel's (pivot, sum, sum of squares,
// count; merged by bn_finalize_tiles_kernel).
#include "igemm_common.h"

namespace {

// 64 input pixels per workgroup (each of the 2 x 2 waves one 32 x 32 accumulator block), three persistent workgroups per CU.
// Measured for 64->64 at 160 -> 320, batch 16 (general parity-class launch: 206 us): one tile per workgroup 172-181 us with 64 or
// 128 pixels and two or three workgroups per CU; persistent with the next tile's input prefetched 169-173 us — what is exposed is
// not the input latency.  By deletion: 143 us without the input loads, 142 us without the output stores, 131 us without both
// (the MFMA floor is 85 us, HBM at 5.5 TB/s 95 us).
constexpr int CT_BN = 64, CT_MI = 1, CT_BM = 64 * CT_MI, CT_WAVES = 3, CT_GRID = 768;

template <int CS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CT_WAVES))) void convt2x2_f32_kernel(const IgemmParams p, int ntiles) {
    constexpr int KC = CS / 4;  // 16-byte chunks (4 channels) per input pixel
    constexpr int AS = CT_BM + 2, BS = CT_BN + 2;  // chunk strides of the LDS images [k/4][row][4 f32] (as igemm_f32_kernel)
    constexpr int A_LD = CT_BM * KC / 256, B_LD = CT_BN * KC / 256;
    static_assert(CS % 16 == 0 && A_LD >= 1 && B_LD >= 1, "whole k-tiles; every thread stages");
    __shared__ f32x4 As[KC * AS];
    __shared__ f32x4 Bs[KC * BS];  // (one buffer and a second barrier per class: a double buffer measured the same)
    __shared__ __attribute__((aligned(16))) unsigned row_off[CT_BM];
    __shared__ float piv[CT_BN], r1[2][CT_BN], r2[2][CT_BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int M = p.N * p.Hs * p.Ws;
    const int n0 = blockIdx.y * CT_BN;
    constexpr unsigned NO_ROW = 0xFFFFFFFFu;
    // raw barrier (+ this wave's LDS operations complete): what the barriers of this kernel order is LDS — a __syncthreads() would
    // also drain the output stores still in flight, the very thing the class walk is there to overlap
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- staging: the input tile (CT_BM pixels x CS channels, 16 lanes per pixel row = 256 contiguous bytes for CS = 64) and the
    // weight panel of one class ([KC][Cd][4 f32], columns n0 .. n0 + 63), through registers
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);
    f32x4 ra[A_LD], rb[B_LD];
    auto load_a = [&](int tile) {  // (a tile index past the end loads nothing: out-of-range offsets)
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int q = tid + j * 256, row = q / KC, ch = q - row * KC;
            const int m = tile * CT_BM + row;
            ra[j] = buffer_load_f32x4(rsrc, (tile < ntiles && m < M) ? (unsigned)((m * CS + 4 * ch) * 4) : OOB_OFFSET);
        }
    };
    auto load_b = [&](int c) {
        const f32x4* w = reinterpret_cast<const f32x4*>(p.wpk + p.wpk_off[c]);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int q = tid + j * 256, col = q & (CT_BN - 1), ch = q / CT_BN;
            rb[j] = w[(long)ch * p.Cd + n0 + col];
        }
    };
    auto stage_b = [&]() {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int q = tid + j * 256, col = q & (CT_BN - 1), ch = q / CT_BN;
            Bs[ch * BS + col] = rb[j];
        }
    };

    const int col = wn * 32 + li;  // this lane's output channel within the tile
    float bv = p.bias ? p.bias[n0 + col] : 0.f;
    // pinned in a register BEFORE the row-predicated store blocks: a load still pending when such a block is entered makes the
    // compiler wait vmcnt(0) inside each of them — and vmcnt also counts the stores, so every row's store waited for the previous one
    asm volatile("" : "+v"(bv));
    char* const dst = reinterpret_cast<char*>(p.dst);
    const unsigned lane_off = (unsigned)(n0 + col) * 4u;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

    // Persistent over the pixel tiles (tile = blockIdx.x, + gridDim.x, ...; equal work per tile, so a static deal): the NEXT tile's
    // input is fetched into registers while the current one is multiplied, and class 3 fetches class 0's panel again — the only
    // global latency a workgroup waits out is that of its first tile.  (One tile per workgroup measured 175 us for 64->64 at 160^2,
    // batch 16, with every tile's input latency exposed: the workgroups of a CU start and finish together and stall together.)
    int tile = blockIdx.x;
    load_a(tile);
    load_b(0);
    stage_b();
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * CT_BM;
        // byte offset of output pixel (n, 2h, 2w) of each input pixel of the tile (class (a, b) adds (a*Wd + b)*Cd elements); the host
        // keeps the output of a launch below 4 GB so that a row is a 32-bit offset from a per-class scalar base (64-bit addresses per
        // row cost 64 registers and the third wave per SIMD); ~0 past the end
        if (tid < CT_BM) {
            const int m = m0 + tid;
            unsigned off = NO_ROW;
            if (m < M) {
                const int hw = p.Hs * p.Ws, n = m / hw, rem = m - n * hw, h = rem / p.Ws, w = rem - h * p.Ws;
                off = (unsigned)(((n * p.Hdf + 2 * h) * p.Wdf + 2 * w) * p.Cd) * 4u;
            }
            row_off[tid] = off;
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {  // (every wave is past the previous tile's last MFMA: the barrier before its last stage_b)
            const int q = tid + j * 256, row = q / KC, ch = q - row * KC;
            As[ch * AS + row] = ra[j];
        }
        lds_barrier();
        load_a(tile + gridDim.x);
        // the output offsets of this lane's rows (accumulator register r of block a: row a*32 + (r & 3) + 8*(r >> 2) + 4*lh)
        u32x4_ ro[CT_MI][4];
#pragma unroll
        for (int a = 0; a < CT_MI; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) ro[a][g] = *reinterpret_cast<const u32x4_*>(&row_off[wm * (32 * CT_MI) + a * 32 + 8 * g + 4 * lh]);
        float pv = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            load_b((c + 1) & 3);
            f32x16 acc[CT_MI];
#pragma unroll
            for (int a = 0; a < CT_MI; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < KC / 2; ++k2) {  // two chunks (8 channels) per step: lanes 0-31 the even chunk, 32-63 the odd one
                const f32x4 bf = Bs[(2 * k2 + lh) * BS + col];
                f32x4 af[CT_MI];
#pragma unroll
                for (int a = 0; a < CT_MI; ++a) af[a] = As[(2 * k2 + lh) * AS + wm * (32 * CT_MI) + a * 32 + li];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < CT_MI; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[e], acc[a], 0, 0, 0);
            }
            lds_barrier();  // the next class's panel replaces this one's once every wave has multiplied it
            stage_b();
            const int ph = c >> 1, pw = c & 1;
            char* const dst_c = dst + (long)((ph * p.Wdf + pw) * p.Cd) * 4;  // (workgroup-uniform)
            if (p.stats) {
                if (c == 0) {  // pivot of every channel: its value at the tile's first pixel, class 0 (m0 < M always)
                    if (wm == 0 && lh == 0) piv[col] = acc[0][0] + bv;
                    lds_barrier();
                    pv = piv[col];
                }
#pragma unroll
                for (int a = 0; a < CT_MI; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * (32 * CT_MI) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float d = (acc[a][r] + bv) - pv;
                        const bool ok = m0 + row < M;
                        s1 += ok ? d : 0.f;
                        s2 += ok ? d * d : 0.f;
                    }
            }
#pragma unroll
            for (int a = 0; a < CT_MI; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned o = ro[a][r >> 2][r & 3];
                    if (o != NO_ROW) *reinterpret_cast<float*>(dst_c + (o + lane_off)) = acc[a][r] + bv;
                }
            lds_barrier();
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
                r1[wm][col] = s1;
                r2[wm][col] = s2;
            }
            lds_barrier();
            const int trow = p.stat_row0 + tile;
            if (tid < CT_BN) {
                const long ch = n0 + tid;
                p.stats[(0L * p.Cd + ch) * p.stat_rows + trow] = piv[tid];
                p.stats[(1L * p.Cd + ch) * p.stat_rows + trow] = r1[0][tid] + r1[1][tid];
                p.stats[(2L * p.Cd + ch) * p.stat_rows + trow] = r2[0][tid] + r2[1][tid];
            }
            if (blockIdx.y == 0 && tid == 0) p.stats[3L * p.Cd * p.stat_rows + trow] = 4.f * (float)min(CT_BM, M - m0);
            lds_barrier();  // (piv / r1 / r2 are rewritten by the next tile)
        }
    }
}

}  // namespace

// rows of BatchNorm partials one launch over M input pixels writes
int dbn_convt_f32_rows(int M) { return dbn_ceil_div(M, CT_BM); }

// p: as for the parity-class launch of igemm_run_one (src, wpk + wpk_off[4], bias, dst, N, Hs, Ws, Cs, Cd, Hdf, Wdf, stats ...)
int dbn_launch_convt_f32(IgemmParams& p, hipStream_t st) {
    const int M = p.N * p.Hs * p.Ws, rows = dbn_convt_f32_rows(M);
    if (p.stat_rows <= 0) p.stat_rows = rows;
    p.launch_rows = rows;
    // persistent workgroups: three per CU (LDS: 51 KB each at Cs = 64) share the tiles evenly
    const int ny = p.Cd / CT_BN;
    const int gx = std::max(1, std::min(rows, CT_GRID / ny));
    const dim3 grid(gx, ny);
    switch (p.Cs) {
        case 16: hipLaunchKernelGGL((convt2x2_f32_kernel<16>), grid, dim3(256), 0, st, p, rows); break;
        case 32: hipLaunchKernelGGL((convt2x2_f32_kernel<32>), grid, dim3(256), 0, st, p, rows); break;
        case 48: hipLaunchKernelGGL((convt2x2_f32_kernel<48>), grid, dim3(256), 0, st, p, rows); break;
        case 64: hipLaunchKernelGGL((convt2x2_f32_kernel<64>), grid, dim3(256), 0, st, p, rows); break;
        default: return DBN_ERR_ARG;
    }
    return dbn_status();
}
