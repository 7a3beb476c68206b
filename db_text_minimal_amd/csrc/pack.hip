// Weight panels of the implicit-GEMM kernels: OIHW parameters -> [K/4][Cd][4] fp32 panels or split-bf16 / fp16 panels, per
// conv, per parity class of a strided transposed conv, or every panel of a model in one launch (dbn_pack_weights_batched).
#include "igemm_common.h"

namespace {

// OIHW -> [Kpad/4][Cd][4] panels.  mode 0: k = (r*S+s)*Cs + cs -> w[cd][cs][r][s] (cs < I);
// mode 1: data-gradient panels, taps r = r0 + rstep*r', s = s0 + rstep*s' (R', S' of them):
//         k = (r'*S'+s')*Cs + cs -> w[cs][cd][r][s].
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

#ifndef DBN_PACK_GRID
// workgroups per job (grid-stride inside; those past a job's size leave at once).  48 until late in round 5: the model's largest jobs
// (512 x 512 x 9) then walked 24 groups of scattered loads per thread, the launch took 256 us in bf16 — on the second stream, but beside the
// input conversion and the stem, which it slowed (the conversion alone: 29 us, in the step: 107).  bf16 step, one box, interleaved three times:
// 48: 1731 / 1736 / 1735 images/s, 192: 1742 / 1745 / 1749, 512: 1754 / 1756 / 1755, 1024: 1755 / 1755 / 1757 (12: 1695); fp32 unchanged
#define DBN_PACK_GRID 512
#endif
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
        constexpr int G = BF16 ? 8 : 4;  // k-values per 16-byte group of the panel
        // one thread per (k-group, output column): its G values are one 16-byte store (per plane), and the tap / channel arithmetic —
        // integer divisions by run-time values — is done once per group instead of once per element (the first form, one element
        // per thread-iteration with 64-bit index arithmetic, took 0.29 ms for the model's 80 panels; on the side stream beside the stem conv, so
        // the step did not move: fp32 718.6 vs 718.9, bf16 1627 vs 1620 images/s interleaved on one box)
        const int total = (Kpad / G) * j.Cd;
        const int RSp = Rp * Sp;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
            const int kg = q / j.Cd, cd = q - kg * j.Cd;
            const int k0 = G * kg;
            float v[G];
            if ((j.Cs & 15) == 0) {  // 16-channel blocks: the group's G <= 8 values share one (tap, 16-channel block)
                const int blk = k0 >> 4;
                const int cb = blk / RSp, tap = blk - cb * RSp;
                const int rp = tap / Sp, sp = tap - rp * Sp;
                const int r = r0 + rstep * rp, sx = s0 + rstep * sp;
                const int cs0 = cb * 16 + (k0 & 15);
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    const int cs = cs0 + e;
                    float t = 0.f;
                    if (k0 + e < K) {
                        if (j.mode == 0) {
                            if (cs < j.I) t = w[(((long)cd * j.I + cs) * j.R + r) * j.S + sx];
                        } else {
                            t = w[(((long)cs * j.I + cd) * j.R + r) * j.S + sx];
                        }
                    }
                    v[e] = t;
                }
            } else {
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    const int k = k0 + e;
                    float t = 0.f;
                    if (k < K) {
                        const int tap = k / j.Cs, cs = k - tap * j.Cs;
                        const int rp = tap / Sp, sp = tap - rp * Sp;
                        const int r = r0 + rstep * rp, sx = s0 + rstep * sp;
                        if (j.mode == 0) {
                            if (cs < j.I) t = w[(((long)cd * j.I + cs) * j.R + r) * j.S + sx];
                        } else {
                            t = w[(((long)cs * j.I + cd) * j.R + r) * j.S + sx];
                        }
                    }
                    v[e] = t;
                }
            }
            if constexpr (BF16 == 2) {  // fp16 panels (one plane)
                unsigned short* out = reinterpret_cast<unsigned short*>(j.out) + krow0 * j.Cd;
                const int kt = kg >> 1, k8 = kg & 1;
                unsigned short h[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (unsigned short)f16_bits(v[e]);
                f32x4 pk;
                __builtin_memcpy(&pk, h, 16);
                *reinterpret_cast<f32x4*>(out + (((long)kt * 2 + k8) * j.Cd + cd) * 8) = pk;
            } else if constexpr (BF16 == 1) {
                unsigned short* out = reinterpret_cast<unsigned short*>(j.out) + krow0 * j.Cd * NS;
                const int kt = kg >> 1, k8 = kg & 1;
                for (int t = 0; t < NS; ++t) {
                    unsigned short h[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned bits = bf16_bits_rne(v[e]);
                        h[e] = (unsigned short)bits;
                        v[e] -= bf16_bits_to_f32(bits);
                    }
                    f32x4 pk;
                    __builtin_memcpy(&pk, h, 16);
                    *reinterpret_cast<f32x4*>(out + ((((long)kt * NS + t) * 2 + k8) * j.Cd + cd) * 8) = pk;
                }
            } else {
                f32x4 pk = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(j.out) + krow0 * j.Cd + (long)q * 4) = pk;
            }
        }
        krow0 += Kpad;
    }
}

// fp32 -> three bf16 planes with a0 + a1 + a2 == a exactly (round-to-nearest-even at each step; 24 mantissa bits = 3 x 8):
// planes[t][i], t = 0..2, plane distance `plane_elems` elements.
#if DBN_HAS_EXPERIMENTS
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

#endif

}  // namespace

extern "C" {

// planes: [3][n] bf16 (n % 4 == 0) — the pre-split form of an fp32 tensor that the at = 3 entry points consume
// (-DDBN_EXPERIMENTS builds only: measured slower than splitting at staging time; otherwise DBN_ERR_ARG)
int dbn_split3(const float* src, void* planes, long n, void* stream) {
#if DBN_HAS_EXPERIMENTS
    DBN_REQUIRE(src && planes && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(split3_kernel, dim3(dbn_grid(n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned short*>(planes), n / 4, n);
    return dbn_status();
#else
    return DBN_ERR_ARG;
#endif
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

// Workgroups per parity class of a stride-4 / -8 transposed conv's panels (the pyramid conv's derived panels: 16 / 64 classes of <= 65 536
// elements, packed every step beside layer1 or — fp32 — in front of the pyramid conv).  FEWER is better here, measured late in round 5 on one
// box, interleaved three times: 1024: bf16 1645-1651 images/s / fp32 715-716, 256: 1645-1648 / 717-718, 64 (until then): 1678-1684 / 724,
// 32: 1687-1691 / 726, 16: 1689-1690 / 725.  (The cap of the other panels — the stem's, levels 0 and 1 — goes the other way: 64 or 16
// instead of 1024 costs fp32 727 -> 717 / 716; those launches sit on the main stream's chain.)
#ifndef DBN_PACKF_CAP
#define DBN_PACKF_CAP 16
#endif
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
    const dim3 grid(dbn_grid(total, 256, f > 2 ? DBN_PACKF_CAP : 1024), f * f);
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
    const dim3 grid(DBN_PACK_GRID, n);
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

}  // extern "C"
