/* C ABI of libdbnet_hip.so — the MI355X (gfx950) DBNet hot path.
 *
 * The reference (huyhoang17/DB_text_minimal) is pure Python/PyTorch and has no FFI
 * layer; its accelerator boundary is the nn.Module call surface of
 *   DBTextModel.forward   /root/reference/src/models.py:34-48
 *   DBLoss.forward        /root/reference/src/losses.py:105-139
 *   the optimizer step    /root/reference/src/train.py:169-172
 * The entry points below are what a binding for that surface calls (see
 * INTEGRATION.md for the ctypes stub).  Conventions:
 *   - all pointers are DEVICE pointers to fp32 unless stated otherwise; the library
 *     never allocates, frees or retains them;
 *   - activations are NHWC ([N,H,W,C], C % 4 == 0); the model input and the three
 *     output maps are NCHW exactly like the reference's tensors;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it;
 *   - return value 0 = ok, 1 = invalid argument, 1000+e = hipError_t e at launch.
 */
#ifndef DBNET_HIP_H
#define DBNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- convolutions (replaces nn.Conv2d / nn.ConvTranspose2d fwd+bwd:
 *      modules/resnet.py:27-34,70-91,167-172; modules/basic.py:17-25;
 *      modules/segmentation_body.py:22-61; modules/segmentation_head.py:24-29,64-79) */

/* OIHW weights -> GEMM panels [ceil(K/16)*4][Cd][4].  mode 0: forward conv panels
 * (Cs = I rounded up to 4, Cd = O); mode 1: data-gradient / ConvTranspose panels
 * (Cs = O, Cd = I) — for stride f = 2, 4, 8: f*f panels, one per output-parity class (only the
 * taps that reach that class).  `out` holds dbn_igemm_panel_floats(...) floats. */
int dbn_pack_weights(const float* w_oihw, int O, int I, int R, int S, int mode, int stride, float* out, void* stream);
long dbn_igemm_panel_floats(int O, int I, int R, int S, int mode, int stride);
int dbn_igemm_packed_floats(int K, int Cd);

/* dst[N,Hd,Wd,Cd] (+)= gather(src[N,Hs,Ws,Cs]) x panels + bias.
 * mode 0: hs = hd*stride - pad + r (Conv2d forward; ConvTranspose2d data gradient)
 * mode 1: hs = (hd + pad - r)/stride when divisible (Conv2d data gradient;
 *         ConvTranspose2d forward); stride f = 2, 4, 8 runs as f*f parity-class problems in one
 *         launch, so no zero taps are multiplied.  bias may be NULL.  tile_hint 0 = auto.
 * `wpk` must come from dbn_pack_weights with the same (mode, stride).
 * Limits (DBN_ERR_ARG otherwise): Cs % 4 == 0, Cd % 64 == 0, N*Hd*Wd < 2^24 output pixels, the source tensor below
 * 0xF0000000 bytes (raw buffer addressing), N*Hd*Wd*Cd < 2^32 output elements (32-bit epilogue offsets). */
int dbn_igemm_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                  int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, void* stream);

/* The same three operations with the products evaluated on the bf16 matrix pipe (fp32 accumulate,
 * fp32 tensors in and out).  ns = 3: every fp32 operand is split exactly into three bf16 terms and
 * six partial products are accumulated -> fp32-accurate (measured error below a plain fp32 fmaf
 * chain) at 6/16 of the fp32-MFMA cost.  ns = 1: operands rounded to bf16 (BASELINE configs[2]
 * "bf16 compute").  Panels must come from dbn_pack_weights_bf16s with the same (mode, stride, ns). */
int dbn_pack_weights_bf16s(const float* w_oihw, int O, int I, int R, int S, int mode, int stride, int ns, float* out, void* stream);
/* All weight panels of a model in one launch (after an optimizer step every panel is stale: ~80 dbn_pack_weights calls of a
 * few microseconds each otherwise).  jobs: DEVICE array of n records
 *   struct { const float* w; void* out; int O, I, R, S, mode, Cs, Cd, f; }   (48 bytes)
 * with Cs = (mode 0 ? I rounded up to 4 : O), Cd = (mode 0 ? O : I), f = (mode 1 && stride > 1 ? stride : 1);
 * out sized by dbn_igemm_panel_floats / dbn_igemm_bf16s_panel_floats.  ns = 0: fp32 panels; 1, 3: split-bf16 panels. */
int dbn_pack_weights_batched(const void* jobs, int n, int ns, void* stream);
long dbn_igemm_bf16s_panel_floats(int O, int I, int R, int S, int mode, int stride, int ns);
int dbn_igemm_bf16s(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                    int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                    void* stream);
int dbn_wgrad_bf16s(const float* sm, const float* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                    int Cb, int I, int R, int S, int stride, int pad, float scale, int ns, void* stream);

/* Data gradient (any dbn_igemm_t mode, no split-K) whose epilogue also reduces the two per-channel sums of the BatchNorm
 * backward that consumes its output — the backward of the conv -> BN -> ReLU chain of resnet.py:70-91 / basic.py:32-36 with one
 * pass over dz and y fewer.  dst is the gradient dz of the (ReLU'd) output of a BatchNorm with input y (same shape and storage as
 * dst), saved mean / rstd; ReLU mask: `zmask` > 0 (a saved activation of that shape) or, with zmask NULL, the BatchNorm's own
 * output recomputed as fma(y, mask_scale[c], mask_shift[c]) > 0.  Per output tile (row) and channel:
 *   part[0][c][row] = sum g,   part[1][c][row] = sum g * (y - mean[c]) * rstd[c],   g = dz_final * [mask > 0]
 * over the FINAL dst values (after `accumulate`): the call must be the last writer of dst.  part: [2][Cd][rows] floats with
 * rows = dbn_igemm_bn_rows(same geometry); hand it to dbn_bn_backward_t as `sums` with sums_parts = rows.  fp32 tensors
 * (at = 0, ns = 0 / 1 / 3) and bf16 tensors (at = 1, ns = 1: sums over the stored, rounded values; y / zmask bf16).  y2 / save_mean2 /
 * save_rstd2 / part2 (all NULL, or all given together with zmask): a SECOND BatchNorm consuming the same dst under the same mask
 * (bn2 and the projection shortcut's BatchNorm of a residual block, resnet.py:84-91); part2 like part.  A strided transposed conv (mode 1, stride > 1) must be tap-complete (R, S >= stride: every
 * output pixel is visited). */
int dbn_igemm_bn_rows(int at, int ns, int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride, int pad, int mode,
                      int tile_hint);
/* Optional in-kernel finalize of those sums (`fin` of dbn_igemm_bnsums_t; NULL: the caller hands `part` to dbn_bn_backward_t,
 * which folds it with its own launch).  The last workgroup to finish in each group of 64 partial rows folds the group, the last
 * group-folder of an output-channel tile folds the groups — fixed summation order, integer counters only — and writes, per
 * BatchNorm, c1c2 = [2][Cd] (sum g / M, sum g*xhat / M with M = N*Hd*Wd: hand it to dbn_bn_backward_t as `sums` with sums_parts
 * = -1, which then launches the apply pass only) and the parameter gradients dgamma = grad_scale * sum g*xhat, dbeta =
 * grad_scale * sum g.  counters: dbn_igemm_bn_final_counters(rows, Cd) ints, ZERO before the first call (the kernels leave
 * them zero); group: dbn_igemm_bn_final_group_floats(rows, Cd) floats of scratch.  The *_2 members belong to the second
 * BatchNorm (y2). */
typedef struct dbn_bnb_final {
    int* counters;
    float* group;
    float *c1c2, *dgamma, *dbeta;
    float *c1c2_2, *dgamma_2, *dbeta_2;
    float grad_scale;
} dbn_bnb_final;
long dbn_igemm_bn_final_counters(int rows, int Cd);
long dbn_igemm_bn_final_group_floats(int rows, int Cd);
int dbn_igemm_bnsums_t(int at, int ns, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                       int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, const void* y,
                       const void* zmask, const float* mask_scale, const float* mask_shift, const float* save_mean,
                       const float* save_rstd, float* part, const void* y2, const float* save_mean2, const float* save_rstd2,
                       float* part2, const dbn_bnb_final* fin, void* stream);

/* Convolution whose epilogue also accumulates the train-mode BatchNorm statistics of its output (per-tile pivot,
 * sum, sum of squares; merged in fp64 by a finalize kernel): one call replaces conv + statistics pass.
 * Arguments: dbn_igemm_f32's, ns = 0 (fp32 MFMA) / 3 / 1 (split-bf16), then dbn_bn_train_stats' BN arguments.
 * With accumulate = 1 the statistics are those of the final (previous dst + this conv) values.  ws: dbn_conv_bn_ws_floats(N,Hd,Wd,Cd,mode,stride) floats. */
long dbn_conv_bn_ws_floats(int N, int Hd, int Wd, int Cd, int mode, int stride);
/* Round 6: the NEXT dbn_conv_bn_t / dbn_winograd_conv_bn[_act]_f32 call of the calling thread folds its statistics rows ITSELF — the
 * workgroup that completes a group of 64 partial rows folds the group, the one that completes the last group folds the groups and writes
 * scale / shift / saved mean / rstd / running statistics: no finalize launch behind the conv (32 launches of a ResNet18-FPN train step).
 * counters: dbn_igemm_bn_final_counters(rows, Cd) ints with rows = dbn_igemm_bn_rows(...) / dbn_winograd_rows(...), ZERO before the first
 * use (the kernels leave them zero) and not shared by calls that may be in flight together; group: dbn_conv_bn_final_group_doubles(rows, Cd)
 * doubles of scratch.  Where the launch has no such epilogue (the 2x2 ConvTranspose kernel) the announcement is dropped and the finalize
 * kernel runs as before.  Same results as the finalize kernel up to fp64 summation order.  NULL clears a pending announcement. */
int dbn_conv_bn_set_final(int* counters, double* group);
long dbn_conv_bn_final_group_doubles(int rows, int Cd);
int dbn_conv_bn_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                    int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                    const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                    float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream);

/* Split-K variant of dbn_igemm_f32 / dbn_igemm_bf16s (ns selects the math) for convs whose output grid cannot fill
 * 256 CUs but whose reduction is long (the coarse FPN levels' data gradients: 6400 pixels x 64 channels, K = 25600):
 * `ksplit` workgroup rows each reduce a contiguous range of k-tiles into their own slab, a second kernel sums the
 * slabs in fixed order and applies bias / accumulate.  mode 0, or mode 1 with stride 1; Cs % 16 == 0.
 * dbn_igemm_splitk_plan returns the split count the library would pick (1 = do not split);
 * slab: dbn_igemm_splitk_slab_floats(...) = ksplit * (N*Hd*Wd*Cd + 1088) floats (the slabs are padded apart so that consecutive splits
 * land on different HBM channels). */
long dbn_igemm_splitk_slab_floats(int ksplit, int N, int Hd, int Wd, int Cd);
int dbn_igemm_splitk_plan(int M, int Cd, int K, int Cs);
int dbn_igemm_splitk_plan_ns(int M, int Cd, int K, int Cs, int ns); /* ... for matrix math ns (the plan follows the tile choice) */
int dbn_igemm_splitk_f32(const float* src, const float* wpk, const float* bias, float* dst, int N, int Hs, int Ws, int Cs, int Hd,
                         int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                         int ksplit, float* slab, void* stream);

/* Pyramid conv — the FPN output conv of segmentation_body.py:75-76 over torch.cat([p2, up2(p3), up4(p4), up8(p5)])
 * (segmentation_body.py:82-87) without building the concatenation:
 *   dst[N,H,W,Cd] = bias + sum_{g=0..3} ConvTranspose2d(k = 2^g + 2, stride 2^g, padding 1)(s_g),  s_g: [N, H>>g, W>>g, Cs]
 * with combined weights from dbn_fpn_combine_weights (level g: [Cs][Cd][k][k]) packed by dbn_pack_weights(mode 1,
 * stride 2^g) into w_g.  One launch; H, W multiples of 8, Cs % 16 == 0, Cd % 128 == 0.  gamma != NULL additionally produces the
 * train-mode BatchNorm coefficients of dst exactly like dbn_conv_bn_f32 (ws: dbn_pyramid_conv_ws_floats floats).
 * tile_hint: 0 | 1 (128x128 is the only tile). */
long dbn_pyramid_conv_ws_floats(int N, int H, int W, int Cd);
int dbn_pyramid_conv_f32(const float* s0, const float* s1, const float* s2, const float* s3, const float* w0, const float* w1,
                         const float* w2, const float* w3, const float* bias, float* dst, int N, int H, int W, int Cs, int Cd,
                         int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                         float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream);

/* tile configuration chosen for tile_hint 0: 1=128x128, 2=256x64, 3=128x64, 4=64x64 */
int dbn_igemm_tile_config(int M, int Cd);
/* ... for matrix math ns (0 exact fp32, 1 / 3 the bf16 pipe: those modes keep the larger tiles) */
int dbn_igemm_tile_config_ns(int M, int Cd, int ns);

/* grad_oihw[O][I][R][S] = scale * sum_p sm[p][o] * big[pixel(p)+tap][i];
 * sm = [N,Ho,Wo,O] (output-side tensor), big = [N,H,W,Cb] (input-side, Cb >= I).
 * slab: dbn_wgrad_slab_floats(...) floats of scratch (split-K partial sums, reduced deterministically). */
int dbn_wgrad_splitk(int N, int Ho, int Wo, int O, int Cb, int R, int S);
/* the same when the size of X (H x W) is known; a call whose tensors exceed the kernels' index ranges (2^24 pixel rows,
 * 32-bit byte offsets) runs as several launches over image ranges, each with its own slabs */
int dbn_wgrad_splitk_hw(int N, int Ho, int Wo, int O, int H, int W, int Cb, int R, int S);
/* slab floats for a call with X of size H x W stored with `es` bytes per element (4 fp32, 2 bf16): exact under the chunking */
long dbn_wgrad_slab_floats_hw(int N, int Ho, int Wo, int O, int H, int W, int Cb, int R, int S, int es);
/* Test hook: lower the per-launch index ranges (pixel rows, bytes per tensor, output elements; 0 = default) so that the image
 * chunking of the conv / weight-gradient entry points can be exercised at small sizes.  Not thread-safe. */
int dbn_set_index_limits(long pixel_rows, long bytes, long elems);
long dbn_wgrad_slab_floats(int N, int Ho, int Wo, int O, int Cb, int R, int S);
int dbn_wgrad_f32(const float* sm, const float* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                  int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream);

/* ---- BatchNorm / ReLU / residual (nn.BatchNorm2d + nn.ReLU: resnet.py:73-91, basic.py:32-36) */
int dbn_reduce_ws_floats(int C);
int dbn_bn_train_stats(const float* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                       float* ws, void* stream);
/* Inference (round 5): eval-mode BatchNorm folded into the conv in front of it (basic.py:32-36, resnet.py:70-91 under model.eval();
 * test.py:53-59).  w_out [O][inner] = w * s[o], b_out [o] = beta + (bias - run_mean) * s[o] with s = gamma / sqrt(run_var + eps) (bias may
 * be NULL); feed w_out / b_out to dbn_pack_weights* / dbn_winograd_pack and the *_act_* entry points below: the conv's epilogue then
 * produces relu(bn(conv(x)) [+ residual]) directly and no BatchNorm pass over the activation runs. */
int dbn_fold_bn_eval(const float* w, int O, long inner, const float* bias, const float* gamma, const float* beta, const float* run_mean,
                     const float* run_var, float eps, float* w_out, float* b_out, void* stream);
/* dst = [relu]( conv(src) + bias [+ res] ) in one launch: dbn_igemm_t's contract (mode 0 at any stride, mode 1 at stride 1; no split-K,
 * no accumulate) plus `res` (NULL or a tensor of dst's shape and storage type: a residual connection) and `relu`. */
int dbn_igemm_act_t(int at, int ns, const void* src, const float* wpk, const float* bias, const void* res, int relu, void* dst, int N,
                    int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride, int pad, int mode, int tile_hint,
                    void* stream);
/* ... the Winograd F(2x2,3x3) form (dbn_winograd_conv_bn_f32's tensors, no statistics) */
int dbn_winograd_conv_act_f32(const float* src, const float* upanel, const float* bias, const float* res, int relu, float* dst, int N,
                              int H, int W, int Cs, int Cd, void* stream);
/* ... and the FPN pyramid conv (dbn_pyramid_conv_from_t's tensors, no statistics): dst = [relu]( [dst +] levels first_level..3 + bias ) */
int dbn_pyramid_conv_act_t(int first_level, int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0,
                           const float* w1, const float* w2, const float* w3, const float* bias, int relu, void* dst, int N, int H, int W,
                           int Cs, int Cd, int ns, void* stream);
int dbn_bn_eval_coef(int C, const float* gamma, const float* beta, const float* run_mean, const float* run_var, float eps,
                     float* scale, float* shift, void* stream);
/* out = act(y*scale+shift [+ res*res_scale+res_shift | + res]) */
int dbn_bn_apply(const float* y, const float* scale, const float* shift, const float* res, const float* res_scale,
                 const float* res_shift, float* out, long M, int C, int relu, void* stream);
/* g = dout * relu_mask; the mask is (zmask > 0) from a saved activation, or recomputed as
 * (y*mask_scale+mask_shift > 0) from the forward's own BN coefficients, or absent (all NULL).
 * dy = BN backward of g; optional gout (+)= g */
int dbn_bn_backward(const float* y, const float* zmask, const float* mask_scale, const float* mask_shift, const float* dout,
                    const float* save_mean, const float* save_rstd,
                    const float* gamma, float* dy, float* gout, int gout_accumulate, float* dgamma, float* dbeta, int M, int C,
                    float grad_scale, float* ws, void* stream);
/* the same with the two per-channel reductions supplied by the kernel that produced dout (sums[2][C]) */
int dbn_bn_backward_from_sums(const float* sums, const float* y, const float* zmask, const float* mask_scale, const float* mask_shift,
                              const float* dout, const float* save_mean, const float* save_rstd, const float* gamma, float* dy,
                              float* gout, int gout_accumulate, float* dgamma, float* dbeta, int M, int C, float grad_scale,
                              float* ws, void* stream);
/* General form: sums optional ([2][C] as above), dbias_conv optional ([C]: column sums of dy x grad_scale = gradient of the bias
 * of the conv that feeds this BatchNorm, formed in the apply pass; needs 256 % (C/4) == 0). */
int dbn_bn_backward_ex(const float* sums, const float* y, const float* zmask, const float* mask_scale, const float* mask_shift,
                       const float* dout, const float* save_mean, const float* save_rstd, const float* gamma, float* dy, float* gout,
                       int gout_accumulate, float* dgamma, float* dbeta, float* dbias_conv, int M, int C, float grad_scale, float* ws,
                       void* stream);
int dbn_col_sum(const float* x, int M, int C, float* out, float scale, float* ws, void* stream);

/* ---- stem pooling (nn.MaxPool2d(3,2,1) over relu(bn1(.)): resnet.py:233-235) */
int dbn_bnrelu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, int N, int H, int W, int C,
                           void* stream);
int dbn_bnrelu_maxpool_bwd(const float* y, const float* scale, const float* shift, const float* pooled, const float* dpool,
                           float* dz, int N, int H, int W, int C, void* stream);

/* ---- FPN nearest upsample + add / concat (segmentation_body.py:79-87) */
int dbn_nearest_up_fwd(const float* src, const float* addend, float* dst, int N, int Hs, int Ws, int C, int H, int W, int Cdst,
                       int coff, void* stream);
int dbn_nearest_up_bwd(const float* dbig, float* dsrc, int N, int Hs, int Ws, int C, int H, int W, int Cbig, int coff,
                       int accumulate, void* stream);

/* ---- final resample of the model (F.interpolate bilinear, align_corners=True: models.py:43-46); the identity
 *      (and elided) when H, W are multiples of 32.  NCHW planes. */
int dbn_bilinear_fwd(const float* src, float* dst, long planes, int Hs, int Ws, int H, int W, void* stream);
int dbn_bilinear_bwd(const float* ddst, float* dsrc, long planes, int Hs, int Ws, int H, int W, void* stream);

/* ---- FPN output conv over [p2 | up2(p3) | up4(p4) | up8(p5)] (segmentation_body.py:55-61,82-87): its data and
 *      weight gradients per upsample group g (factor f = 2^g) are those of a (f+2)x(f+2), stride-f, pad-1 conv with
 *      COMBINED weights (sums of the 3x3 taps that read the same low-resolution pixel): 47 % of the MACs, no concat. */
int dbn_fpn_combine_weights(const float* w, int Co, int Cin, int group, int Cg, float* wd, void* stream);
int dbn_fpn_scatter_wgrad(const float* t0, const float* t1, const float* t2, const float* t3, int Co, int Cg, float* dw,
                          void* stream);

/* ---- layout / misc */
int dbn_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, void* stream);
int dbn_add_inplace(const float* x, float* y, long n, void* stream);

/* ---- DB head tail (segmentation_head.py:28-29,35-45,77-79,106-108): last ConvTranspose2d(64,1,2,2) + Sigmoid of both
 * branches, step_function, concat.  With bn_scale/shift (all four [64], or all NULL) xb/xt are the PRE-BatchNorm outputs
 * of the preceding ConvTranspose2d and BN+ReLU (segmentation_head.py:27,74-75) is applied on load, so the post-BN
 * activations of the two largest tensors of the network are never written. */
int dbn_head_tail_fwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* bias_b,
                      const float* bias_t, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                      const float* bn_shift_t, float* out, int N, int Hq, int Wq, int channels, float kstep, void* stream);
int dbn_head_tail_bwd_ws_floats(void);
/* dxb/dxt: gradients w.r.t. the (post-ReLU) inputs of the last ConvTranspose2d; the BatchNorm backward applies the mask.
 * bn_sums (optional, needs the bn_* pointers incl. the saved mean / rstd): [4][64] = per channel sum of the masked
 * gradient and of masked gradient * xhat for branch b, then t — feed each [2][64] half to dbn_bn_backward_from_sums. */
int dbn_head_tail_bwd(const float* xb, const float* xt, const float* wb, const float* wt, const float* preds,
                      const float* dpreds, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                      const float* bn_shift_t, const float* bn_mean_b, const float* bn_rstd_b, const float* bn_mean_t,
                      const float* bn_rstd_t, float* bn_sums, float* dxb, float* dxt, float* dw_b, float* dbias_b, float* dw_t,
                      float* dbias_t, int N, int Hq, int Wq, int channels, float kstep, float grad_scale, float* ws, void* stream);

/* ---- 3x3 / stride 1 / pad 1 forward convolution through the Winograd transform F(2x2, 3x3), fp32 arithmetic (2.25x fewer matrix
 * FLOPs than the direct form; BasicBlock convs resnet.py:70-91, FPN smooth convs segmentation_body.py:55-61, the head's 256 -> 64
 * convs segmentation_head.py:24-25,64-68).  fp32 NHWC tensors of any H x W (worked in 8 x 16 pixel patches; ragged edges are masked;
 * dbn_winograd_eligible says yes when >= 3/4 of the patch area is real), Cs % 16 == 0 channels in the source
 * tensor (I <= Cs of them real), Cd % 64 == 0 (dbn_winograd_eligible).  upanel: the filters transformed once per parameter update
 * (dbn_winograd_pack; dbn_winograd_panel_floats(O, Cs) floats).  gamma non-NULL: the train-mode BatchNorm that follows is folded
 * in exactly as in dbn_conv_bn_f32 (ws: dbn_winograd_ws_floats floats).  Same result as the direct convolution up to fp32 rounding
 * of a different summation order (not bit for bit). */
/* Round 5: winograd_f32_kernel runs with PERSISTENT workgroups — at most two per CU, pulling (patch, channel tile) items from per-XCD
 * counters the library keeps per stream — whenever a launch has more items than the chip has workgroup slots (512).  0 restores one
 * workgroup per item (round 4's form; A/B and test hook).  Results are bit-identical either way. */
int dbn_set_winograd_persistent(int on);
/* ... and the one-time phase stagger between the two persistent workgroups of a CU, in permille of one item's matrix time (0: off) */
int dbn_set_winograd_stagger(int permille);
/* ... and the channel blocks the patch form stages per barrier: 1 (default) or 2 (Cs % 32 == 0 layers; A/B hook) */
int dbn_set_winograd_blocks_per_barrier(int n);
int dbn_winograd_eligible(int N, int H, int W, int Cs, int Cd);
long dbn_winograd_panel_floats(int O, int Cs);
/* dgrad = 0: panel of the forward conv of w [O][I][3][3] over a source with Cs >= I channels.  dgrad = 1: panel of the DATA GRADIENT
 * of the conv with weights w [I][O][3][3] (filters rotated by 180 degrees, channel roles swapped): maps dy (Cs >= I channels) to dx (O). */
int dbn_winograd_pack(const float* w_oihw, int O, int I, int Cs, int dgrad, float* out, void* stream);
/* n dbn_winograd_pack calls in one launch (after an optimizer step every panel is stale).  jobs: DEVICE array of n records
 *   struct { const float* w; float* out; int O, I, Cs, dgrad; }   (32 bytes) */
int dbn_winograd_pack_batched(const void* jobs, int n, void* stream);
int dbn_winograd_rows(int N, int H, int W);
/* The data gradient through the same kernel: dx [N,H,W,Cd] = [dx +] conv(dy [N,H,W,Cs]) with the dgrad = 1 panel.  y non-NULL: the
 * epilogue also produces the two per-channel sums of the BatchNorm backward that consumes dx — arguments and semantics of
 * dbn_igemm_bnsums_t, part = [2][Cd][dbn_winograd_rows(N,H,W)] floats, fin = optional in-kernel finalize (counters sized by
 * dbn_igemm_bn_final_counters(rows, Cd)). */
int dbn_winograd_dgrad_bnsums_f32(const float* dy, const float* upanel, float* dx, int N, int H, int W, int Cs, int Cd, int accumulate,
                                  const void* y, const void* zmask, const float* mask_scale, const float* mask_shift,
                                  const float* save_mean, const float* save_rstd, float* part, const void* y2, const float* save_mean2,
                                  const float* save_rstd2, float* part2, const dbn_bnb_final* fin, void* stream);
long dbn_winograd_ws_floats(int N, int H, int W, int Cd);
/* ... with the BatchNorm + ReLU in FRONT of the conv applied on load (basic.py:32-36, resnet.py:77-80): the conv's input is
 * relu(src * in_scale[c] + in_shift[c]) ([Cs] floats each, the coefficients of dbn_bn_apply; both NULL: src itself).  src is that
 * BatchNorm's input; its output tensor is never written.  Bit-identical to dbn_bn_apply followed by dbn_winograd_conv_bn_f32. */
int dbn_winograd_conv_bn_act_f32(const float* src, const float* in_scale, const float* in_shift, const float* upanel, const float* bias,
                                 float* dst, int N, int H, int W, int Cs, int Cd, const float* gamma, const float* beta, float eps,
                                 float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean,
                                 float* save_rstd, float* ws, void* stream);
int dbn_winograd_conv_bn_f32(const float* src, const float* upanel, const float* bias, float* dst, int N, int H, int W, int Cs, int Cd,
                             const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                             float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream);

/* Weight gradient of a 3x3 / stride-1 / pad-1 conv through the Winograd transform F(2x2,3x3) in exact-fp32 arithmetic
 * (csrc/winograd_wgrad_f32.hip; replaces torch.autograd's conv weight gradient for resnet.py:70-91, segmentation_body.py:55-61,
 * segmentation_head.py:24-25,64-68).  x [N][H][W][Cb] (channels >= I are zero padding), dy [N][H][W][O], grad [O][I][3][3] =
 * scale * dW (overwritten).  slab: dbn_winograd_wgrad_slab_floats floats of scratch.  phases: 1 = matrix kernel (-> slab),
 * 2 = slab reduction + G^T . G (-> grad), 3 = both.  Deterministic (fixed-order fp64 reduction, no atomics).  x_scale / x_shift
 * non-NULL ([Cb] each): the conv's input is relu(x * x_scale[c] + x_shift[c]), applied on load (see dbn_winograd_conv_bn_act_f32). */
int dbn_winograd_wgrad_eligible(int N, int H, int W, int O, int Cb, int I);
int dbn_winograd_wgrad_linear(int H, int W); /* 1: small map, consecutive-tile form (kernel <true> in a trace) */
long dbn_winograd_wgrad_slab_floats(int N, int H, int W, int O, int Cb);
int dbn_winograd_wgrad_f32(int phases, const float* dy, const float* x, const float* x_scale, const float* x_shift, float* slab, float* grad,
                           int N, int H, int W, int O, int Cb, int I, float scale, void* stream);

/* ---- DBLoss (losses.py:18-40,48-66,75-82,105-139); preds [N,3|2,H,W], gts [4,N,H,W].
 * One launch: the workgroup that finishes last folds every workgroup's partial sums (fixed order) and writes losses[5] and
 * coef[8].  ws: dbn_db_loss_ws_bytes() bytes of scratch, no initialisation required (it ends with the arrival counter of that
 * hand-over, which the library clears on `stream` in front of every launch); not shared between calls that may run concurrently. */
int dbn_db_loss_ws_bytes(void);
int dbn_db_loss_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                    float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream);
int dbn_db_loss_bwd(const float* preds, const float* gts, const float* coef, const float* grad_losses, float alpha, float beta,
                    int N, int H, int W, int channels, float* dpreds, void* stream);
/* DBLoss(reduction='sum'): losses.py:30 forwards the reduction string to F.binary_cross_entropy, so the scalar BCE of the
 * OHEM term is summed instead of averaged; everything else as dbn_db_loss_fwd (same ws; backward: dbn_db_loss_bwd). */
int dbn_db_loss_sum_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                        float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream);
/* The two entry points above evaluate losses.py:33-39's `topk(loss * negative, n_neg).sum()` (loss a SCALAR in these reductions)
 * in closed form, bce * n_neg — exact for binary prob_gt / supervision_mask maps (what data_loaders.py:112-134 produces) and
 * only for those.  coef[7] receives the number of pixels whose gt or mask is neither 0 nor 1 (the host side refuses such maps,
 * db_text_minimal_amd/losses.py).  dbn_db_loss_frac_fwd evaluates the expression literally for ANY maps:
 * bce * (sum(positive) + sum of the n_neg largest negative_i) / (n_pos + n_neg + eps), top-k sum by the device radix select of
 * the per-pixel path below.  sum != 0: reduction='sum'.  ws: dbn_db_loss_ohem_ws_bytes(N,H,W) bytes; backward: dbn_db_loss_bwd. */
int dbn_db_loss_frac_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                         float negative_ratio, float eps, int sum, float* losses, float* coef, void* ws, void* stream);

/* DBLoss(reduction='none') — true per-pixel OHEM (losses.py:30-39 with a per-pixel BCE): the n_neg largest
 * negative losses are found by a 3-pass radix select on device (no sort, no host sync).  `ws` holds
 * dbn_db_loss_ohem_ws_bytes(N,H,W) bytes (no initialisation required, as above) and must stay untouched between _fwd and _bwd. */
long dbn_db_loss_ohem_ws_bytes(int N, int H, int W);
int dbn_db_loss_ohem_fwd(const float* preds, const float* gts, int N, int H, int W, int channels, float alpha, float beta,
                         float negative_ratio, float eps, float* losses, float* coef, void* ws, void* stream);
int dbn_db_loss_ohem_bwd(const float* preds, const float* gts, const float* coef, const float* grad_losses, const void* ws,
                         float alpha, float beta, int N, int H, int W, int channels, float* dpreds, void* stream);

/* ---- per-step pixel metric (cal_text_score / RunningScore._fast_hist, text_metrics.py:14-24,63-82):
 *      2x2 confusion matrix of (P*M > thresh) vs int(G*M), accumulated on device into 5 doubles
 *      [unused, n01, n10, n11, total] (index = 2*gt + pred) — replaces 3 D2H copies + np.bincount per step */
int dbn_pixel_confusion(const float* preds, long batch_stride, const float* gt, const float* mask, int N, int H, int W, float thresh,
                        double* hist, void* stream);

/* ---- optimizer (torch.optim.Adam, train.py:114-117,172) over one flat buffer */
int dbn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, int step,
                  float grad_scale, void* stream);

/* 1 when the library was built with -DDBN_EXPERIMENTS (make -C csrc EXP=1): adds the variants that were measured and rejected —
 * pre-split bf16 planes (at = 3, dbn_split3), the LDS-DMA fp32 weight gradient (dbn_set_wgrad_variant(1)), DBN_* environment
 * overrides of the tile / split heuristics.  The product library (default) has none of them: no getenv anywhere. */
int dbn_has_experiments(void);

/* ---- measurement aid (bench.py, no reference counterpart): one wave samples the shader clock (s_memtime) against the constant
 * reference clock (s_memrealtime) over `microseconds`; out2 = {shader cycles, reference ticks} (two uint64 in device memory).
 * dbn_wall_clock_khz: the reference clock's rate.  sustained MHz = cycles / ticks * khz / 1000 */
int dbn_clock_probe(void* out2, int microseconds, void* stream);
int dbn_wall_clock_khz(void);

/* ---- deformable convolution of the DCN backbones (resnet.py:54-65,81-82,111-124,145-146: conv2_offset ->
 * torchvision.ops.DeformConv2d, deformable_groups = 1), lowered to sampling + GEMM:
 *   cols = dbn_deform_im2col(x, offset);  y = dbn_igemm_f32(cols as [N,Ho,Wo,R*S*C], 1x1 panels of the permuted weight)
 * backward: dcols = 1x1 data gradient, dW = 1x1 weight gradient of (dy, cols) permuted back, then dbn_deform_col2im.
 * offset: [N*Ho*Wo][off_stride] floats, channel 2k = dy and 2k+1 = dx of tap k = r*S + s (torchvision layout);
 * off_stride >= 2*R*S lets the offsets live in the 64-channel output of the (zero-padded) offset conv. */
int dbn_deform_im2col(const float* x, const float* offset, float* cols, int N, int H, int W, int C, int Ho, int Wo, int R, int S,
                      int stride, int pad, int off_stride, void* stream);
/* dx = [dx +] adjoint of the sampling applied to dcols; doffset is written (channels >= 2*R*S zeroed).  DETERMINISTIC since
 * round 3: every contribution is accumulated in 64-bit fixed point (scale from max |dcols| of the call: resolution 2^-43 of the
 * largest column gradient), so LDS / global integer atomics give the same bits in any order.  ws: dbn_deform_col2im_ws_bytes(...)
 * bytes of scratch; accumulate = 1 adds to the gradient dx already holds; off_stride % 4 == 0. */
long dbn_deform_col2im_ws_bytes(int N, int H, int W, int C, int Ho, int Wo, int R, int S);
int dbn_deform_col2im(const float* dcols, const float* x, const float* offset, float* dx, float* doffset, int accumulate, void* ws, int N,
                      int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream);
/* to_ohwi = 1: dst[O][T][C] = scale * src[O][C][T]; 0: dst[O][C][T] = scale * src[O][T][C] */
int dbn_permute_weight(const float* src, float* dst, int O, int C, int T, int to_ohwi, float scale, void* stream);

/* ---- array work of SegDetectorRepresenter ahead of the (host, unchanged) OpenCV contour code ---- */
/* postprocess.py:51-52 `pred > thresh` on channel 0 of pred[N][channels][H][W], as a uint8 {0,1} bitmap out[N][H][W]
 * (H*W % 4 == 0): the D2H copy the contour tracer needs shrinks 4x. */
int dbn_binarize_u8(const float* pred, int N, int channels, int H, int W, float thresh, unsigned char* out, void* stream);
/* postprocess.py:186-198 box_score_fast for K boxes at once: scores[k] = mean of bitmap[H][W] over the cv2.fillPoly
 * mask of boxes[k][P][2] (x, y; float, as produced by get_mini_boxes / approxPolyDP), P <= 64.  Polygons with fewer
 * vertices are padded by repeating the last vertex. */
int dbn_box_scores(const float* bitmap, int H, int W, const float* boxes, int K, int P, float* scores, void* stream);


/* =====================================================================================================================
 * Activation storage types (BASELINE configs[2]-[4]).  Every entry point above that moves activation tensors has a `_t`
 * form with the storage type first: DBN_AT_F32 (0, the fp32 contract above), DBN_AT_BF16 (1: activations, their gradients
 * and the weight panels are stored in bf16 in HBM; bf16 MFMA, fp32 accumulators / BatchNorm statistics / loss sums / weight
 * gradients / master weights), DBN_AT_F16 (2: fp16 inference).  Tensors typed `void*` are stored in that type; everything
 * typed `float*` stays fp32.  The reference has one dtype (fp32, src/train.py:96-98); these are the native data paths of the
 * reduced-precision configurations.
 * ===================================================================================================================== */
#define DBN_AT_F32 0
#define DBN_AT_BF16 1
#define DBN_AT_F16 2
/* conv / weight-gradient entry points only, with ns = 3 (bf16x3): the source operand(s) are PRE-SPLIT fp32 tensors — three
 * bf16 planes [3][N,H,W,C] with a0 + a1 + a2 == a exactly, made by dbn_split3 — gathered without conversion; dst stays fp32 */
#define DBN_AT_SPLIT3 3
int dbn_split3(const float* src, void* planes, long n, void* stream);

/* weight panels: kind 0 fp32, 1 bf16, 3 bf16x3 (three bf16 planes), 2 fp16.  cs: channels of the source tensor for mode 0
 * (0: I rounded up to 4; the 16-bit model input is stored with 16 channels) */
long dbn_igemm_panel_floats_t(int kind, int O, int I, int R, int S, int mode, int stride, int cs);
int dbn_pack_weights_t(int kind, const float* w_oihw, int O, int I, int R, int S, int mode, int stride, int cs, float* out, void* stream);

/* dbn_igemm_f32 / _bf16s / _splitk_f32 in one: at = storage of src / dst (16-bit storage: ns = 1, Cs % 16 == 0), ns = matrix math
 * (0 exact fp32, 1 one 16-bit plane, 3 bf16x3), ksplit > 1 with `slab` (ksplit * (N*Hd*Wd*Cd + 1088) floats) = split-K */
int dbn_igemm_t(int at, int ns, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd,
                int Wd, int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ksplit, float* slab,
                void* stream);
/* dbn_conv_bn_f32 with typed tensors; the BatchNorm statistics are those of the fp32 accumulators (before the storage rounding) */
int dbn_conv_bn_t(int at, const void* src, const float* wpk, const float* bias, void* dst, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                  int Cd, int R, int S, int stride, int pad, int mode, int accumulate, int tile_hint, int ns,
                  const float* gamma, const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                  float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream);
int dbn_pyramid_conv_t(int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0, const float* w1,
                       const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs, int Cd,
                       int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                       float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws, void* stream);
/* first_level = 1 (at = 0 with ns = 0, or 16-bit storage with ns = 1): dst already holds level 0's part (the 3x3 conv of s0 with the bias —
 * dbn_winograd_conv_bn_f32, or dbn_igemm_t in mode 1 with the level-0 panel, which takes the 16-bit pixel-patch kernel);
 * the launch adds levels 1-3 (s0, w0, bias unused).  first_level = 0: dbn_pyramid_conv_t. */
int dbn_pyramid_conv_from_t(int first_level, int at, const void* s0, const void* s1, const void* s2, const void* s3, const float* w0,
                            const float* w1, const float* w2, const float* w3, const float* bias, void* dst, int N, int H, int W, int Cs,
                            int Cd, int tile_hint, int ns, const float* gamma, const float* beta, float eps, float momentum,
                            float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd, float* ws,
                            void* stream);
/* weight gradient with typed dY (sm) and X (big): at = 0 (any ns) or 1 (bf16, ns = 1); slabs and the gradient stay fp32 */
int dbn_wgrad_t(int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O, int H, int W,
                int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream);

/* weight-gradient kernel selection (same results up to the summation order inside a split): 0 = defaults — fp32 tensors: the
 * register-transposing kernel; bf16 tensors: LDS-DMA panels + transposing LDS reads (wgrad_tr_kernel); 1 = LDS-DMA kernel for
 * exact-fp32 math on fp32 tensors (pixel-major LDS image, three-stage ring); 2 = the register-transposing kernel also where the
 * defaults take wgrad_tr_kernel or wgrad_patch_kernel (3x3 / stride-1 layers in the 16-bit matrix modes); 3 = the defaults with
 * the general per-pixel gather also where the block-wise k-tiles apply (measurements, tests) */
int dbn_set_wgrad_variant(int variant);
/* ConvTranspose2d(2x2, stride 2, padding 0) forward in exact fp32 (dbn_igemm_f32 / dbn_conv_bn_f32 with mode 1, stride 2, R = S = 2,
 * Cs in {16, 32, 48, 64}, no accumulate, tile_hint 0) runs as convt2x2_f32_kernel<Cs>: the input tile stays in LDS and the four output
 * parity classes are walked inside the workgroup.  0 routes these calls through the general parity-class launch again (tests, A/B);
 * returns the previous setting. */
int dbn_set_convt_kernel(int on);
/* tile variant as dbn_wgrad_tile_config, + 16 when the matrix kernel is wgrad_tr_kernel<BM,BN,2,2> (its rocprofv3 symbol) */
int dbn_wgrad_kernel_config(int at, int ns, int O, int J, int Cb);
/* ... with the layer geometry: + 32 when the matrix kernel is wgrad_patch_kernel<ns, at> (3x3 / stride 1, 16-bit matrix modes);
 * + 64 when wgrad_f32_kernel<BM,BN,2,2,ns,at,ROW> runs with ROW = 1: fp32 tensors whose output map tiles into 16x1, 8x2 or 4x4 pixel
 * blocks — the reduction then walks such blocks with scalar-offset addressing (same sum; 16x1 blocks: bit-identical to ROW = 0) */
int dbn_wgrad_kernel_config_hw(int at, int ns, int O, int Cb, int R, int S, int stride, int pad, int Ho, int Wo, int H, int W);
/* 0: route 3x3 / stride-1 convolutions of the 16-bit matrix modes through the generic gather loop instead of the pixel-patch
   form (A/B and test hook; returns the previous setting) */
int dbn_set_patch_conv(int on);
/* 1 (default): the 3x3 / stride-1 convolutions of the 16-bit storage types with 64 -> 64, 128 -> 128 or 256 -> 64 channels run in the
 * weight-resident kernel (csrc/wres16.hip: the panel in registers, the activations streamed row by row); 0: the pixel-patch kernel
 * again (test / A-B hook); 2: as 1, and also maps whose width is not a multiple of the kernel's 32-column strip (by default those stay
 * on the pixel-patch kernel, which is faster there; the tests of the ragged last strip use 2).  Returns the previous setting.
 * dbn_igemm_kernel_config reports such a launch with bit 64. */
int dbn_set_wres16(int on);
/* The 128 x 256 tile of the 16-bit storage types (Cd % 256 == 0; one column tile for the FPN output conv's 256 channels: half the gathered
 * activation bytes per output).  0: off (128 x 128); 1 (default): the pyramid conv on launches of >= 4096 such tiles (BASELINE configs[4]:
 * 25 600); 2: also plain forward / stride-1 data-gradient launches of the generic loop; 3: as 2 whatever the size (tests).  Per output
 * element the same products in the same order: bit-identical results.  Returns the previous setting (DBN_PYR_WIDE in the environment sets
 * the initial one). */
int dbn_set_pyramid_wide(int on);
int dbn_pyramid_wide_would_run(int at, int N, int H, int W, int Cs, int Cd); /* 1: such a pyramid conv call launches the 128 x 256 tile */
int dbn_wres16_would_run(int at, int mode, int N, int H, int W, int Cs, int Cd, int bnb, int y2); /* 1: such a call launches that kernel */

/* ---- measurement infrastructure (bench.py: roofline.peak_sustained): what the matrix pipe sustains on this box with non-zero
 * operands — the chip clocks to its power budget, far below the 2.4 GHz of the nominal peaks under back-to-back MFMAs (csrc/mfma_probe.hip).
 * kind 0: v_mfma_f32_32x32x2_f32, 1: ..._32x32x16_bf16, 2: ..._32x32x16_f16.  operands: 65536 bytes of values of that type; out: 524288
 * floats of scratch.  One launch (1024 workgroups x 8 waves, iters x 16 MFMAs per wave) of dbn_mfma_sustained_flops(kind, iters) FLOPs. */
int dbn_mfma_sustained(int kind, const void* operands, float* out, int iters, void* stream);
long dbn_mfma_sustained_flops(int kind, int iters);
/* First-round stagger of the exact-fp32 implicit-GEMM launches (workgroups sharing a CU start out of phase so that their prologues /
 * epilogues overlap other workgroups' MFMA loops), in permille of the nominal delay; 0 = off.  Returns the previous setting. */
int dbn_set_stagger(int permille);
/* 1: the implicit-GEMM workgroups run their prologue and epilogue at raised wave priority (s_setprio), so that they do not wait behind
 * the MFMA loops of the other workgroups on their CU; 0: everything at the default priority.  Returns the previous setting. */
int dbn_set_phase_priority(int on);
/* Diagnostic builds only (make TRACE=1: per-workgroup phase timestamps of the exact-fp32 implicit-GEMM kernels, tools/trace_probe.py);
 * the product library ignores the buffer and returns 0. */
int dbn_set_trace(void* buf, long max_blocks);
/* what one (unchunked) dbn_igemm_t call launches: tile configuration as dbn_igemm_tile_config, + 16 for the pixel-patch kernel
   (kmode: 0 forward, 1 stride-1 data gradient, 2 parity classes, 3 pyramid) — the template arguments of its rocprofv3 symbol */
int dbn_igemm_kernel_config(int at, int ns, int kmode, int N, int Hs, int Ws, int Cs, int Hd, int Wd, int Cd, int R, int S, int stride,
                            int pad, int tile_hint, int ksplit);
/* The slab reductions (phase 2) of many dbn_wgrad_phase_t calls in ONE launch.  dbn_wgrad_reduce_describe takes phase 2's
 * arguments and fills a host-side job record instead of launching (DBN_ERR_ARG for layers without the 64-channel reduction form:
 * the stem); the caller uploads the records and the prefix table of their `blocks` and calls dbn_wgrad_reduce_many.  Same sums
 * in the same order as the per-layer launches.  (A training step's side stream otherwise carries one small reduction behind
 * every weight-gradient kernel; conv / ConvTranspose weight gradients of resnet.py:70-91, basic.py:32-36, segmentation_head.py:24-29.) */
typedef struct dbn_wgrad_reduce_job {
    const float* slab;
    float* grad;
    int splitk, O, J, Jp, BM, BN, Cb, I, RS, G, natural, blocks;
    float scale;
    int smem_bytes;
} dbn_wgrad_reduce_job;
int dbn_wgrad_reduce_describe(int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo, int O,
                              int H, int W, int Cb, int I, int R, int S, int stride, int pad, float scale, void* job);
int dbn_wgrad_reduce_many(const void* jobs, const int* first, int n_jobs, int total_blocks, int max_smem, void* stream);
/* tile variant of the weight-gradient kernel for O output channels, J = R*S*Cb columns: 1 = 64x192, 2 = 128x128, 3 = 64x128, 4 = 64x64 */
int dbn_wgrad_tile_config(int O, int J);
/* dbn_wgrad_t in two calls: phase 1 = matrix kernels (-> slabs), phase 2 = slab reduction (-> grad_oihw) */
int dbn_wgrad_phase_t(int phase, int at, int ns, const void* sm, const void* big, float* slab, float* grad_oihw, int N, int Ho, int Wo,
                      int O, int H, int W, int Cb, int I, int R, int S, int stride, int pad, float scale, void* stream);

/* deformable conv (resnet.py:54-65,111-124) on typed tensors: dcols, x, offset, dx and doffset all in the activation type `at` */
int dbn_deform_im2col_t(int at, const void* x, const void* offset, void* cols, int N, int H, int W, int C, int Ho, int Wo, int R, int S,
                        int stride, int pad, int off_stride, void* stream);
int dbn_deform_col2im_t(int at, const void* dcols, const void* x, const void* offset, void* dx, void* doffset, int accumulate, void* ws,
                        int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride, void* stream);
/* The same adjoint (the backward of torchvision.ops.DeformConv2d's sampling, resnet.py:61-65,119-124) as a GATHER in plain fp32 (round 5):
 * a team per input pixel searches the taps whose undeformed position lies within ceil(max |offset|) of it (the maximum is taken on the
 * device: no host synchronisation, larger offsets only widen the window) and adds their corner contributions in candidate order;
 * doffset is a per-sample reduction over the channels.  One fixed summation order: bit-reproducible; equal to dbn_deform_col2im_t up to
 * the fp32 rounding of the sums; non-finite dcols values reach exactly the elements their samples touch.  ws:
 * dbn_deform_col2im_gather_ws_bytes(N, Ho, Wo) bytes, no initialisation needed.  C <= 512, off_stride % 4 == 0, 9 <= R * S <= 32. */
long dbn_deform_col2im_gather_ws_bytes(int N, int Ho, int Wo);
int dbn_deform_col2im_gather_t(int at, const void* dcols, const void* x, const void* offset, void* dx, void* doffset, int accumulate,
                               void* ws, int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int off_stride,
                               void* stream);
/* out_bits[0] = bit pattern of max |offset| (0x7FC00000 if an element is not finite): what bounds the gather's window (n % 4 == 0) */
int dbn_deform_offset_absmax_t(int at, const void* offset, long n, unsigned* out_bits, void* stream);
int dbn_cast_f32(int at, const float* src, void* dst, long n, void* stream);

int dbn_bn_train_stats_t(int at, const void* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                         float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                         float* ws, void* stream);
int dbn_bn_apply_t(int at, const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                   const float* res_shift, void* out, long M, int C, int relu, void* stream);
/* The general BatchNorm backward: `sums` optional — [2*C][sums_parts] partial sums (or [2][C] with sums_parts = 1) of the masked
 * gradient and of masked gradient * xhat produced by the kernel that wrote dout (dbn_head_tail_bwd, dbn_bnrelu_maxpool_bwd_t,
 * dbn_igemm_bnsums_t); sums_parts = -1: `sums` is the FINALIZED pair [2][C] (sum / M, sum_xhat / M) of a dbn_bnb_final, dgamma /
 * dbeta are already written: only the apply pass runs.  dbias_conv optional as in dbn_bn_backward_ex */
int dbn_bn_backward_t(int at, const float* sums, int sums_parts, const void* y, const void* zmask, const float* mask_scale,
                      const float* mask_shift, const void* dout, const float* save_mean, const float* save_rstd, const float* gamma,
                      void* dy, void* gout, int gout_accumulate, float* dgamma, float* dbeta, float* dbias_conv, int M, int C,
                      float grad_scale, float* ws, void* stream);
int dbn_col_sum_t(int at, const void* x, int M, int C, float* out, float scale, float* ws, void* stream);
int dbn_bnrelu_maxpool_fwd_t(int at, const void* y, const float* scale, const float* shift, void* out, int N, int H, int W, int C,
                             void* stream);
/* bn_mean / bn_rstd / bn_part optional (all or none): also emit the partial sums of the BatchNorm backward that consumes dz,
 * bn_part = [2*C][dbn_maxpool_bwd_parts(N,H,W,C)] floats (needs 256 % (C/4) == 0) */
int dbn_maxpool_bwd_parts(int N, int H, int W, int C);
int dbn_bnrelu_maxpool_bwd_t(int at, const void* y, const float* scale, const float* shift, const void* pooled, const void* dpool,
                             void* dz, int N, int H, int W, int C, const float* bn_mean, const float* bn_rstd, float* bn_part,
                             void* stream);
/* The stem's pool with a RECORDED argmax (replaces nn.MaxPool2d(3, 2, 1) over relu(bn1(conv1(x))) and its autograd,
 * /root/reference/src/modules/resnet.py:167-172,231-235).  Forward: out as dbn_bnrelu_maxpool_fwd_t, plus idx [N,Ho,Wo,C] bytes — the position
 * 3 r + q of the window's FIRST maximum in scan order (PyTorch's rule), 15 where the pooled value is 0 — and ypool [N,Ho,Wo,C] (storage type),
 * the pre-BatchNorm value y at that position.  Backward: dy [N,H,W,C] = gradient at y given dpool, through pool, ReLU and the train-mode
 * BatchNorm, in one pass over y (the BatchNorm's two channel sums come from the pooled tensors); dgamma / dbeta are written (x grad_scale).
 * Needs 256 % (C/4) == 0; ws: dbn_maxpool_bn_backward_ws_floats(N,H,W,C) floats. */
int dbn_bnrelu_maxpool_fwd_arg_t(int at, const void* y, const float* scale, const float* shift, void* out, void* idx, void* ypool, int N, int H,
                                 int W, int C, void* stream);
long dbn_maxpool_bn_backward_ws_floats(int N, int H, int W, int C);
int dbn_maxpool_bn_backward_t(int at, const void* y, const void* dpool, const void* idx, const void* ypool, const float* save_mean,
                              const float* save_rstd, const float* gamma, void* dy, float* dgamma, float* dbeta, int N, int H, int W, int C,
                              float grad_scale, float* ws, void* stream);
int dbn_nearest_up_fwd_t(int at, const void* src, const void* addend, void* dst, int N, int Hs, int Ws, int C, int H, int W, int Cdst,
                         int coff, void* stream);
int dbn_nearest_up_bwd_t(int at, const void* dbig, void* dsrc, int N, int Hs, int Ws, int C, int H, int W, int Cbig, int coff,
                         int accumulate, void* stream);
/* x [N,3,H,W] fp32 -> [N,H,W,4] fp32 (at = 0) or [N,H,W,16] in the 16-bit type (channels 3.. zero) */
int dbn_nchw3_to_nhwc4_t(int at, const float* x, void* out, int N, int H, int W, void* stream);
/* ---- The stem convolution in 16-bit storage on a PACKED input (round 5; csrc/stem16.hip): Conv2d(3 -> 64, 7x7, stride 2, pad 3, no bias) of
 * modules/resnet.py:167-172,231-235.  The image is stored as xp [N][dbn_stem16_padded_h(H)][dbn_stem16_padded_w(W)][4] in the 16-bit type —
 * (r, g, b, 0) per pixel at offset (3, 3) inside a ZERO border the caller provides once (dbn_nchw3_to_padded4_t rewrites the interior only;
 * x4 non-NULL: also the packed [N][H][W][4] form, the X operand of the stem's weight gradient) — so the conv needs no bounds tests and
 * K = 7 x 8 x 4 = 224 instead of the 7 x 7 x 16 = 784 of the 16-channel-block form.  wpk: dbn_stem16_panel_bytes() bytes from
 * dbn_stem16_pack(kind 1 bf16 | 2 fp16, w [64][3][7][7] fp32).  y [N][(H-1)/2+1][(W-1)/2+1][64] in the 16-bit type.  gamma non-NULL: + the
 * train-mode BatchNorm that follows, as dbn_conv_bn_t (ws: (3 * 64 + 1) * dbn_stem16_rows() floats). */
int dbn_stem16_padded_h(int H);
int dbn_stem16_padded_w(int W);
int dbn_stem16_rows(void);
long dbn_stem16_panel_bytes(void);
int dbn_stem16_eligible(int at, int N, int H, int W);
int dbn_stem16_pack(int kind, const float* w_oihw, void* out, void* stream);
int dbn_nchw3_to_padded4_t(int at, const float* x, void* xp, void* x4, int N, int H, int W, void* stream);
int dbn_stem16_conv_bn_t(int at, const void* xp, const void* wpk, void* y, int N, int H, int W, const float* gamma, const float* beta, float eps,
                         float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                         float* ws, void* stream);
/* inference (round 5): out [N][Hq][Wq][64] = MaxPool2d(3, 2, 1)(relu(conv7x7/2 (xp) * scale + shift)) in ONE launch — the stem of
 * modules/resnet.py:231-235 in eval mode; scale / shift: the eval-mode BatchNorm coefficients (dbn_bn_eval_coef).  The 64-channel conv
 * output is never written.  Eligible: dbn_stem16_eligible and an even conv height Ho = (H - 1) / 2 + 1; Hq = (Ho - 1) / 2 + 1. */
int dbn_stem16_pool_eligible(int at, int N, int H, int W);
int dbn_stem16_conv_bn_relu_pool_t(int at, const void* xp, const void* wpk, const float* scale, const float* shift, void* out, int N, int H, int W,
                                   void* stream);
/* ---- ConvTranspose2d(64 -> 64, 2x2, stride 2) forward in 16-bit storage (round 5; csrc/convt16.hip): the head's up-sampling layers,
 * modules/segmentation_head.py:27-29,74-76.  x [N][H][W][64], y [N][2H][2W][64] in the 16-bit type; wpk: dbn_convt16_panel_bytes() bytes from
 * dbn_convt16_pack(kind 1 bf16 | 2 fp16, w [64][64][2][2] fp32 — the module's own [Cin][Cout][kh][kw] layout); bias [64] fp32 or NULL.
 * gamma non-NULL: + the train-mode BatchNorm that follows, as dbn_conv_bn_t (ws: (3 * 64 + 1) * dbn_convt16_rows() floats). */
int dbn_convt16_rows(void);
long dbn_convt16_panel_bytes(void);
int dbn_convt16_eligible(int at, int N, int H, int W, int Cin, int Cout);
int dbn_convt16_pack(int kind, const float* w_iohw, void* out, void* stream);
int dbn_convt16_bn_t(int at, const void* x, const void* wpk, const float* bias, void* y, int N, int H, int W, const float* gamma, const float* beta,
                     float eps, float momentum, float* run_mean, float* run_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                     float* ws, void* stream);
/* ... and the same kernel as a pointwise conv 64 -> 64 | 256 on 16-bit storage, inference form (round 5): y [N][H][W][Cout] =
 * [relu](conv1x1(x [N][H][W][64]) + bias) — nn.Conv2d(64, Cout, 1) with the eval-mode BatchNorm folded into weight / bias (dbn_fold_bn_eval)
 * and the ReLU of modules/basic.py:32-36; the FPN lateral reduce_conv_c2 of resnet18, modules/segmentation_body.py:46,68.  Panel:
 * dbn_pw16_panel_bytes() bytes from the OIHW weight [Cout][64][1][1] (dbn_pw16_pack; kind 1 bf16, 2 fp16). */
int dbn_pw16_eligible(int at, int N, int H, int W, int Cin, int Cout);
long dbn_pw16_panel_bytes(void);
int dbn_pw16_pack(int kind, const float* w_oihw, int Cout, void* out, void* stream);
int dbn_pw16_act_t(int at, const void* x, const void* wpk, const float* bias, int relu, void* y, int N, int H, int W, int Cout, void* stream);
/* inference (round 5): the DB head's tail of BOTH branches in one launch — per branch ConvTranspose2d(64, 64, 2, 2) -> eval-mode BatchNorm ->
 * ReLU -> ConvTranspose2d(64, 1, 2, 2) -> Sigmoid (modules/segmentation_head.py:27-29,35-45,74-79): out [N][2][4 Hq][4 Wq] fp32 (channel 0 the
 * binarize branch, 1 the threshold branch); x_*: [N][Hq][Wq][64] in the activation type; panel_*: dbn_convt16_pack of the first ConvT;
 * scale / shift: dbn_bn_eval_coef of the BatchNorm behind it; w2_*: the second ConvT's weight [64][1][2][2], bias2_*: [1]. */
int dbn_head16_eligible(int at, int N, int Hq, int Wq);
int dbn_head16_tail_eval_t(int at, const void* x_b, const void* x_t, const void* panel_b, const void* panel_t, const float* bias1_b,
                           const float* bias1_t, const float* scale_b, const float* shift_b, const float* scale_t, const float* shift_t,
                           const float* w2_b, const float* w2_t, const float* bias2_b, const float* bias2_t, float* out, int N, int Hq, int Wq,
                           void* stream);

/* ... 16-bit storage: the 16-channel form and (out4 non-NULL) the packed 4-channel form of dbn_nchw3_to_nhwc4_packed_t in ONE launch */
int dbn_nchw3_to_nhwc16_and_4_t(int at, const float* x, void* out16, void* out4, int N, int H, int W, void* stream);
/* the same into [N,H,W,4] of the storage type (16-bit: 4 channels, not a 16-channel block): X operand of the stem weight gradient */
int dbn_nchw3_to_nhwc4_packed_t(int at, const float* x, void* out, int N, int H, int W, void* stream);
/* head tail with typed 64-channel inputs / input gradients; the maps, dpreds and the ConvT parameter gradients are fp32 */
int dbn_head_tail_fwd_t(int at, const void* xb, const void* xt, const float* wb, const float* wt, const float* bias_b,
                        const float* bias_t, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                        const float* bn_shift_t, float* out, int N, int Hq, int Wq, int channels, float kstep, void* stream);
/* test / A-B hook: the 16-bit forward's lane layout — 0 by size (eight lanes x eight channels per pixel from 2^22 quarter pixels), 1 always,
 * -1 never; returns the previous setting */
int dbn_set_head_tail_wide(int mode);
int dbn_head_tail_bwd_t(int at, const void* xb, const void* xt, const float* wb, const float* wt, const float* preds,
                        const float* dpreds, const float* bn_scale_b, const float* bn_shift_b, const float* bn_scale_t,
                        const float* bn_shift_t, const float* bn_mean_b, const float* bn_rstd_b, const float* bn_mean_t,
                        const float* bn_rstd_t, float* bn_sums, void* dxb, void* dxt, float* dw_b, float* dbias_b, float* dw_t,
                        float* dbias_t, int N, int Hq, int Wq, int channels, float kstep, float grad_scale, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif
