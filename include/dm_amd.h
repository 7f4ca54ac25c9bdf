/*
 * dm_amd.h — C ABI of libdm_amd.so, the MI355X (gfx950) kernels behind the DDPM / ContextUnet hot
 * path of Shen-Yuuu/DiffusionModel.
 *
 * The reference has no FFI of its own: its hot path is a chain of torch.nn calls inside
 * new_scripy.py:70-477 (and MNIST_script.py:31-300).  Each entry point below replaces one family of
 * those calls; the reference call sites are cited per function.  All pointers are DEVICE pointers
 * unless stated otherwise, all activations are NHWC ("channels last", C contiguous), `dtype` is
 * DM_F32, DM_BF16 or DM_F16 for activations/packed weights, master weights / statistics / gradients of
 * parameters are always fp32.  Every function enqueues on `stream` and returns immediately
 * (no host synchronisation, no allocation: hipGraph-capturable).  Return value: 0 on success,
 * a negative DM_E* code on invalid arguments, or a positive hipError_t.
 *
 * Python binding: diffusionmodel_amd/_lib.py (ctypes).  See INTEGRATION.md.
 */
#ifndef DM_AMD_H
#define DM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dm_stream_t; /* hipStream_t */

enum { DM_F32 = 0, DM_BF16 = 1, DM_F16 = 2 };   /* DM_F16: IEEE half — the reference's torch.cuda.amp.autocast() dtype (new_scripy.py:784) */
enum { DM_ACT_NONE = 0, DM_ACT_GELU = 1, DM_ACT_RELU = 2, DM_ACT_SIGMOID = 3 };
enum { DM_OK = 0, DM_EINVAL = -1, DM_EUNSUPPORTED = -2 };

int dm_version(void);
const char* dm_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM gather convolution on MFMA (bf16: v_mfma_f32_16x16x32_bf16, f32: v_mfma_f32_16x16x4_f32).
 *   out[orow(m)][coff+n] = act( scale[n] * sum_{t<T} sum_{c<C1+C2} in(pix(m,t))[c] * w[n*ldw + t*(C1+C2) + c] + shift[n] )
 *   m = (b, qy, qx) over B x Hq x Wq;  tap t -> (iy, ix) = (qy*sy + (t/KW)*ty + oy0, qx*sx + (t%KW)*tx + ox0),
 *   out-of-image taps read zero;  `in` is the channel concatenation of in1 (C1 ch) and in2 (C2 ch, may be 0);
 *   orow(m) = (b*Ho + qy*osy + ooy)*Wo + qx*osx + oox.
 * Serves: Conv2d 3x3/s1 (new_scripy.py:184,189,225,243,311,314), 4x4/s2 (:229), 1x1 (:217,222),
 * ConvTranspose2d k/k (:298; MNIST_script.py:88,141), their input-gradients (with transposed packed
 * weights) and eval-mode BatchNorm+GELU folded into scale/shift/act (:185-186 etc. under .eval()).
 * Optional per-channel partial statistics of the pre-activation value (for train-mode BatchNorm):
 *   psum[mb*N + n], psq[mb*N + n] for each 128-row block mb.
 * Requirements: C1 % VE == 0, C2 % VE == 0 (VE = 8 for bf16, 4 for f32), ldw % VE == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct DmConv {
    const void* in1; const void* in2; const void* w;
    const float* scale; const float* shift;      /* per output channel, may be NULL (1 / 0) */
    void* out;                                    /* NHWC `dtype`, or NCHW fp32 when out_nchw_f32 */
    float* psum; float* psq;                      /* NULL, [ceil(M/128)][N], or [stat_slots][N] (see stat_slots) */
    int32_t dtype, act, out_nchw_f32;
    int32_t B, Hi, Wi, C1, C2;
    int32_t Hq, Wq, sy, sx;
    int32_t T, KW, ty, tx, oy0, ox0;
    int32_t Ho, Wo, osy, osx, ooy, oox;
    int32_t N, ldw, ldc, coff;
    int32_t in2_batch;                            /* 0 or B: in2 has B samples.  n < B (n | B): in2 has n samples and sample b reads
                                                     b % n — the CFG sampler's doubled batch over a skip tensor computed once
                                                     (halo-resident 3x3 kernel only) */
    int32_t stat_slots;                           /* 0: psum / psq are [ceil(M/128)][N] partial rows (dm_bn_finalize folds them).  S in 1..64: they are
                                                     ZEROED [S][N] **double** accumulators, tile i adds its column sums into slot i % S with fp64
                                                     atomics and dm_bn_act_fwd_slots folds the S slots itself — no finalize launch */
    const void* addend;                           /* NULL, or a tensor of the output's layout and dtype: out = act(...) + addend.  The
                                                     input-gradient launch of a layer whose input feeds a second consumer adds that
                                                     consumer's gradient here instead of a separate elementwise pass (autograd's
                                                     accumulation of new_scripy.py's skip / residual tensors) */
} DmConv;
int dm_conv(const DmConv* d, dm_stream_t stream);
/* d4[0..3]: four descriptors that differ only in w, oy0, ox0, ooy, oox — the input-gradient launches of the four output-parity classes
 * of a stride-2 layer (new_scripy.py:229).  One launch of the four-tap halo kernel when it is eligible and the four classes together
 * fill the chip; otherwise equivalent to four dm_conv calls. */
int dm_conv_parity4(const DmConv* d4, dm_stream_t stream);
/* tuning knob: staging pipeline of dm_conv — 1 = register staging, 2..4 = LDS-DMA gather ring with that many stages,
   5 (default) = halo-resident kernel for 3x3 stride-1 layers on 16/32/64-pixel rows with 64-channel multiples, gather ring otherwise */
/* 1 (default): 3x3 stride-1 launches with more output tiles than CUs run as PERSISTENT workgroups that fetch the next tile's first
 * halo chunk and weight stages behind the last chunk of the current tile (igemm_halo_p.hip); 0: one workgroup per tile. */
int dm_set_conv_persist(int on);
int dm_last_conv_persistent(void);   /* 1: the last dm_conv launch ran the persistent form */
int dm_set_conv_variant(int variant);
/* which kernel family the last dm_conv call launched: 0 = gather kernel (conv_igemm*), 1 = conv3x3_halo_kernel, 3 = pointwise,
 * 4 = conv3x3_packtap_kernel, 5 = conv3x3_narrow_kernel (profiling aid) */
int dm_last_conv_path(void);
/* 1 (default): 3x3 stride-1 layers over an 8-channel input (the stem, new_scripy.py:381 -> :184; the head's input gradient, :314) run
 * on the packed-tap kernel (igemm_skinny.hip: K = taps x 8 channels, 3 MFMA k-steps instead of 18), and 3x3 layers with <= 16 output
 * channels (the head, :314) on the narrow halo kernel of the same file; 0: halo kernel's partial chunk / gather kernel */
int dm_set_conv_packtap(int on);
/* the value dm_set_conv_variant last set (default 5) */
int dm_get_conv_variant(void);

/* Weight gradient of the same gather convolution (fp32 atomics into dw, which the caller zeroes or
 * accumulates into):  dw[n*ldw + t*C + c] += sum_m dy[orow(m)][n] * in(pix(m,t))[c],
 * optionally dbias[n] += sum_m dy[orow(m)][n].   dy has row stride ldy (>= N, % VE == 0). */
typedef struct DmWgrad {
    const void* dy; const void* in1; const void* in2;
    float* dw; float* dbias;
    int32_t dtype;
    int32_t B, Hi, Wi, C1, C2;
    int32_t Hq, Wq, sy, sx;
    int32_t T, KW, ty, tx, oy0, ox0;
    int32_t Ho, Wo, osy, osx, ooy, oox;
    int32_t N, ldy, ldw, splitk;
    int32_t overwrite;            /* r04: != 0 -> dw holds garbage: the launch leaves dw = this gradient (dbias still accumulates).  The 3x3 halo
                                     kernel's reduce launch then writes instead of read-modify-writing, so the caller need not have zeroed dw
                                     (FusedAdamW.zero_grad leaves those ranges alone); every other kernel path zeroes dw itself first. */
} DmWgrad;
int dm_conv_wgrad(const DmWgrad* d, dm_stream_t stream);
/* tuning knob: 1 = register staging, 2 = LDS-DMA, 3 (default) = 2 + the halo-resident kernel for 3x3 stride-1 layers
   (needs dm_set_workspace when the pixel range is split over workgroups; falls back to 2 without it) */
int dm_set_wgrad_variant(int variant);
/* 1 when the last dm_conv_wgrad launch used the halo-resident 3x3 kernel, 2 for the 1x1 kernel (wgrad_pw_kernel), 3 for the four-tap form of the halo kernel (4x4 / stride 2), 0 for the per-tap
   kernels (measurement aid) */
int dm_last_wgrad_path(void);   /* 4: wgrad3x3_skinny_kernel (below) */
/* on != 0 (default): the weight gradient of the full-resolution 3x3 layers with 8 (padded) channels on one side — the stem (new_scripy.py:381
   -> :184) and the head (:314) — on wgrad3x3_skinny_kernel (taps x 8 channels packed into the MFMA's column operand); 0: halo kernel */
int dm_set_wgrad_skinny(int on);
/* r04 experiment (igemm_halo4.hip): the 3x3 halo kernel with one wave per SIMD (four waves, 128 x 64 wave tiles) on rows of >= 64
 * pixels, one source.  0 = off (default), 1 = compiler schedule, 2 = fragments prefetched one sub-step ahead on a pinned schedule.
 * Bit-identical results either way; measured 2 - 4 % slower than the eight-wave kernel.  DM_CONV_WAVE4. */
int dm_set_conv_wave4(int mode);
/* on != 0 (default): the weight gradient of the 16-bit 4x4 / stride-2 / pad-1 layers (output rows of 32 / 16 pixels or 8x8 output
   images) on the four-tap form of the halo-resident kernel; 0: per-tap kernel; > 1: also the workgroup count its pixel split aims at
   (default 256).  dm_last_wgrad_path() reports 3.  Measurement knob. */
int dm_set_wgrad_tap4(int on);
/* enable != 0 (default): 16-bit 1x1 layers with min(N, C) in {32, 64, 128} and at least min_pixels output pixels take
   wgrad_pw_kernel; target_blocks > 0 sets the workgroup count its pixel split aims at for 4096-float output blocks (default 384;
   larger blocks get proportionally fewer workgroups: the same volume of fp32 atomics), 0 leaves it; min_pixels >= 0 sets the
   pixel threshold (default 0), < 0 leaves it.  Measurement / test knob. */
int dm_set_wgrad_pw(int enable, int target_blocks, int min_pixels);
/* bit 0: the 4x4 / stride-2 convolution and its input gradient on conv_tap4_halo_kernel (default on; 0 = gather kernel);
   bit 1: set = the short-K 1x1 layers stay on the gather kernel too (default: conv_pw_kernel).  Measurement knob. */
int dm_set_conv_tap4(int on);
/* halo kernels with the channel chunks split over workgroups: 1 (default since r03) the last split to arrive folds the partials and
   runs the epilogue in the same launch (arrival counters at the tail of the workspace; partials handed over by sc1 stores, an
   agent-scope counter add and sc1 loads — no fence); 0 a separate epilogue launch folds them.  (r02's form of 1 published through
   agent-scope fences and was 2 ms per train step slower.) */
int dm_set_splitk_inkernel(int on);
/* Caller-owned device scratch (16-byte aligned) the MFMA kernels may use for split partial sums; it must outlive every
   launch that follows.  One stream at a time: launches that use it are ordered by the stream they are issued on. */
int dm_set_workspace(void* ws, int64_t bytes);

/* Weight repacks. src is the fp32 master in physical layout [N][T][C] (= torch channels_last of OIHW).
 *  dm_pack_w:   dst[n][t][cp]      = c < C ? src[n][t][c] : 0          (cast + channel pad), dst dtype
 *  dm_pack_wT:  dst[c][tt][np]     = n < N ? src[n][taps[tt]][c] : 0   (transpose for input-gradient),
 *               taps = NULL means identity over T.  `taps` is a HOST pointer (<= 64 entries).
 *  dm_unpad_dw: dst[n][t][c] (+)= src[n][t][cp] for c < C   (fp32, accumulate != 0 adds) */
int dm_pack_w(const float* src, void* dst, int dtype, int N, int T, int C, int Cp, dm_stream_t s);
int dm_pack_wT(const float* src, void* dst, int dtype, int N, int T, int C, int Tt, const int32_t* taps, int Np, dm_stream_t s);
int dm_unpad_dw(const float* src, float* dst, int N, int T, int C, int Cp, int accumulate, dm_stream_t s);
/* Many dm_pack_wT in one launch (all transposed packs of a model after an optimiser step).  Device tables:
 *   entries[e] = {src, dst, N, T, C, Tt, Np, dtype} as 8 x int64;  taps[e][16] int32 (Tt <= 16);
 *   dtype | 0x100: `src` points to a bf16 copy of the weights in the same element layout (dm_adamw's shadow) instead of fp32;
 *   blocks[b]  = {e, n tile, c tile, tt} int32: one 64 x 64 tile of entry e per workgroup. */
int dm_pack_multi(const int64_t* entries, const int32_t* taps, const int32_t* blocks, int n_blocks, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d in training mode (new_scripy.py:79-80,185-186,190-191,218-219,226-227) on an [M][C] matrix
 * (NHWC flattened), fused with the following activation.
 * ---------------------------------------------------------------------------------------------- */
/* partial column sums of z ([M][C], dtype) -> psum/psq [nblk][C], nblk = dm_colstat_blocks(M) */
int dm_colstat_blocks(int M);
int dm_col_stats(const void* z, int dtype, int M, int C, float* psum, float* psq, dm_stream_t s);
/* partials -> mean, rstd (biased var, eps); running stats update with momentum (unbiased var);
 * running_* may be NULL */
int dm_bn_finalize(const float* psum, const float* psq, int nblk, int M, int C, float eps, float momentum,
                   float* mean, float* rstd, float* running_mean, float* running_var, dm_stream_t s);
/* y = act(gamma*(z-mean)*rstd + beta) */
int dm_bn_act_fwd(const void* z, void* y, int dtype, int M, int C, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, int act, dm_stream_t s);
/* backward pass 1: partial sums of g = dy*act'(.) and g*xhat  -> p1/p2 [nblk][C] */
int dm_bn_act_bwd_reduce(const void* z, const void* dy, int dtype, int M, int C, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int act, float* p1, float* p2, dm_stream_t s);
/* reduce partials [nblk][C] -> out[C] (out = or += sum) */
int dm_col_reduce(const float* part, int nblk, int C, float* out, int accumulate, dm_stream_t s);
/* the same for two partial arrays of one shape in a single launch: out1 = sum(part1), out2 = sum(part2) */
int dm_col_reduce2(const float* part1, const float* part2, int nblk, int C, float* out1, float* out2, dm_stream_t s);
/* backward pass 2: dz = gamma*rstd*(g - s1/M - xhat*s2/M); s1 = dbeta, s2 = dgamma (already reduced) */
int dm_bn_act_bwd_apply(const void* z, const void* dy, void* dz, int dtype, int M, int C, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int act,
                        const float* s1, const float* s2, dm_stream_t s);
/* Slot-folded forms (train-mode BatchNorm, new_scripy.py:185-186 etc.): the statistics arrive as [slots][C] DOUBLE accumulators
 * (DmConv.stat_slots / dm_bn_act_bwd_reduce_slots add into ZEROED buffers with fp64 atomics) and the consuming kernel folds them in its
 * prologue; workgroup 0 publishes mean / rstd + running statistics (forward) or dbeta / dgamma (backward).  Per layer this removes the
 * dm_bn_finalize and dm_col_reduce2 launches.  C % 8 == 0 (bf16 / fp16) or % 4 (fp32), C <= 8192. */
int dm_bn_act_fwd_slots(const void* z, void* y, int dtype, int M, int C, const void* psum, const void* psq, int slots, float eps,
                        float momentum, const float* gamma, const float* beta, int act, float* mean, float* rstd,
                        float* running_mean, float* running_var, dm_stream_t s);
int dm_bn_act_bwd_reduce_slots(const void* z, const void* dy, int dtype, int M, int C, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int act, void* p1, void* p2, int slots, dm_stream_t s);
int dm_bn_act_bwd_apply_slots(const void* z, const void* dy, void* dz, int dtype, int M, int C, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, const void* p1, const void* p2, int slots,
                              float* dbeta, float* dgamma, dm_stream_t s);
/* eval-mode BN folded to per-channel scale/shift for dm_conv: scale = gamma*rsqrt(var+eps),
 * shift = (conv_bias - mean)*scale + beta  (conv_bias may be NULL) */
int dm_bn_fold(const float* gamma, const float* beta, const float* rmean, const float* rvar, const float* conv_bias,
               float eps, int C, float* scale, float* shift, dm_stream_t s);

/* GroupNorm(G) + activation (new_scripy.py:299-300,312-313,167-168) on NHWC [B][HW][C]. */
int dm_gn_act_fwd(const void* x, void* y, int dtype, int B, int HW, int C, int G, float eps, const float* gamma,
                  const float* beta, int act, float* mean, float* rstd, dm_stream_t s);
/* dgamma/dbeta are accumulated (+=) with atomics: zero them (or pass the running gradient) */
int dm_gn_act_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int HW, int C, int G, const float* gamma,
                  const float* beta, int act, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                  dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * SE block + residual combine (new_scripy.py:143-158, 196-205): out = (res + x2 * s[b,c]) * inv
 * ---------------------------------------------------------------------------------------------- */
int dm_pool_hw(const void* x, int dtype, int B, int HW, int C, float* mean_bc, dm_stream_t s);   /* mean over HW */
/* the SE excitation MLP itself (fc -> GELU -> fc -> sigmoid, no biases) is dm_linear_* + dm_act_* below */
int dm_scale_residual_fwd(const void* x2, const void* res, const float* sgate, void* out, int dtype, int B, int HW,
                          int C, float inv, dm_stream_t s);    /* sgate NULL -> 1 (MNIST block, no SE) */
int dm_scale_residual_bwd_reduce(const void* dout, const void* x2, int dtype, int B, int HW, int C, float inv,
                                 float* dsgate, dm_stream_t s);               /* dsgate[b,c] = sum_hw dout*inv*x2 */
int dm_scale_residual_bwd_apply(const void* dout, const float* sgate, const float* dy_mean, void* dx2, void* dres,
                                int dtype, int B, int HW, int C, float inv, dm_stream_t s);
                                                               /* dx2 = dout*inv*s + dy_mean[b,c]/HW ; dres = dout*inv */

/* The whole squeeze-excite chain in 2 (forward) / 3 (backward) launches instead of 6 / 8 (chain.hip): the pooling pass leaves
 * per-workgroup partial sums in the workspace (dm_set_workspace), one kernel per 16 samples folds them and runs
 * fc -> GELU -> fc -> sigmoid (new_scripy.py:148-157) with the hidden vector in LDS.  C % 4 == 0, R % 4 == 0, R <= 128.
 *   forward:  y[B][C] = mean_hw(x2), hid[B][R] = y W1^T, gh = gelu(hid), sg[B][C] = sigmoid(gh W2^T)
 *             (y / hid / gh all NULL: inference, only sg is written)
 *   backward: dsg = sum_hw dout*inv*x2 (as dm_scale_residual_bwd_reduce), dlogit = dsg sg (1 - sg)  [scratch, B x C],
 *             dhid = (dlogit W2) gelu'(hid)  [scratch, B x R],  dy[B][C] = dhid W1  (overwritten),
 *             dw1[R][C] += dhid^T y,  dw2[C][R] += dlogit^T gh   (fp32 atomics: accumulators) */
int dm_se_fwd(const void* x2, int dtype, int B, int HW, int C, const float* w1, const float* w2, int R, float* y,
              float* hid, float* gh, float* sg, dm_stream_t s);
int dm_se_bwd(const void* dout, const void* x2, int dtype, int B, int HW, int C, float inv, const float* sg,
              const float* hid, const float* gh, const float* y, const float* w1, const float* w2, int R, float* dlogit,
              float* dhid, float* dy, float* dw1, float* dw2, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Coordinate attention (new_scripy.py:97-140): strip pooling and the final gated multiply.
 * ---------------------------------------------------------------------------------------------- */
int dm_ca_pool_fwd(const void* x, int dtype, int B, int H, int W, int C, float* xh, float* xw, dm_stream_t s);
int dm_ca_pool_bwd(const float* dxh, const float* dxw, const void* dout_gate, void* dx, int dtype, int B, int H, int W,
                   int C, dm_stream_t s);                      /* dx = dout_gate + dxh/W + dxw/H ; dout_gate may be NULL (0) */
/* out = x * (al*sig(lh[b,y,c]) + be*sig(lw[b,x,c])),  al = sig(alpha)/(sig(alpha)+sig(beta)+1e-8) */
int dm_ca_gate_fwd(const void* x, const float* lh, const float* lw, const float* alpha, const float* beta, void* out,
                   int dtype, int B, int H, int W, int C, dm_stream_t s);
int dm_ca_gate_bwd(const void* x, const void* dout, const float* lh, const float* lw, const float* alpha,
                   const float* beta, void* dx_gate, float* dlh, float* dlw, float* dalpha_dbeta, int dtype, int B,
                   int H, int W, int C, dm_stream_t s);
/* xo = x + sig(gamma)*y on fp32 strips; backward gives dy = sig*dxo and dgamma */
int dm_sigmix_fwd(const float* x, const float* y, const float* gamma, float* xo, int n, dm_stream_t s);
int dm_sigmix_bwd(const float* dxo, const float* y, const float* gamma, float* dy, float* dgamma, int n, dm_stream_t s);

/* The strip chain of CoordAttn between its pooling pass and its gate pass (new_scripy.py:105-129) in 2 launches per direction
 * instead of 14 / 22 (chain.hip).  All tensors fp32; R = C / reduction; C % 4 == 0, R % 4 == 0, R <= 128.
 *   forward : z = conv1(x) [B*L][R] ; a = gelu(bn1(z)) ; h2w = h2w_proj(a_h) ; w2h = w2h_proj(a_w) ;
 *             x_h' = a_h + sigmoid(gamma_h) adapt_H(w2h) ; x_w' = a_w + sigmoid(gamma_w) adapt_W(h2w)   (adapt = adaptive_avg_pool
 *             along the strip, :119-120: the identity when H == W) ; l_h = conv_h(x_h') [B][H][C] ; l_w = conv_w(x_w') [B][W][C]
 *             train != 0: batch statistics (stat = scratch of (ceil(B H / 16) + ceil(B W / 16)) * 2 R floats), running statistics
 *             updated with `momentum`; train == 0: running statistics.  save != 0: mean / rstd / a / x' are written (backward).
 *   backward: from dlh / dlw to dxh / dxw (overwritten) and every parameter gradient: d_w* / d_b* / d_gam accumulate (+=, fp32
 *             atomics), d_bn_* are overwritten.  Scratch: gh, gw [B*L][R], bnpart [2][B][2 R], dh2w [B*H][R], dw2h [B*W][R]. */
typedef struct DmCaChain {
    int32_t B, H, W, C, R, train, save;
    float eps, momentum;
    const float *xh, *xw;                                   /* strip means [B][H][C], [B][W][C] */
    const float *w1h, *b1h, *w1w, *b1w;                     /* conv1_h / conv1_w: [R][C], [R] */
    const float *bn_h_g, *bn_h_b, *bn_w_g, *bn_w_b;         /* bn1_h / bn1_w weight, bias [R] */
    float *rm_h, *rv_h, *rm_w, *rv_w;                       /* running mean / var [R] */
    const float *whw, *bhw, *wwh, *bwh;                     /* h2w_proj, w2h_proj: [R][R], [R] */
    const float *gam_h, *gam_w;                             /* gamma_h, gamma_w (1 float each) */
    const float *wch, *bch, *wcw, *bcw;                     /* conv_h / conv_w: [C][R], [C] */
    float *zh, *zw, *mean_h, *rstd_h, *mean_w, *rstd_w, *ah, *aw, *xhp, *xwp, *lh, *lw;
    float *stat;
    const float *dlh, *dlw;
    float *gh, *gw, *bnpart, *dh2w, *dw2h;
    float *dxh, *dxw;
    float *d_w1h, *d_b1h, *d_w1w, *d_b1w, *d_bn_h_g, *d_bn_h_b, *d_bn_w_g, *d_bn_w_b, *d_whw, *d_bhw, *d_wwh, *d_bwh, *d_gam, *d_wch,
        *d_bch, *d_wcw, *d_bcw;
} DmCaChain;
int dm_ca_chain_fwd(const DmCaChain* d, dm_stream_t s);
int dm_ca_chain_bwd(const DmCaChain* d, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Small dense layers in fp32 (EmbedFC new_scripy.py:255-268; CoordAttn 1x1 convs on strips :76-91; SE MLP :148-157).
 *   y[M][N] = act(x[M][K] w[N][K]^T + b);   backward: dx = dy w (overwritten), dw += dy^T x, db += column sums of dy
 *   (any of dx / dw / db may be NULL).  K % 4 == 0, N % 4 == 0 and 16-byte aligned tensors take the MFMA kernels of dense.hip
 *   (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation; dw / db by fp32 atomics), anything else a plain SGEMM.
 * ---------------------------------------------------------------------------------------------- */
int dm_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, dm_stream_t s);
int dm_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int K, int N,
                  dm_stream_t s);                              /* dw/db accumulate (+=); dx overwritten, may be NULL */
int dm_act_fwd(const float* x, float* y, int n, int act, dm_stream_t s);
int dm_act_bwd(const float* x, const float* dy, float* dx, int n, int act, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Layout / resampling glue
 * ---------------------------------------------------------------------------------------------- */
/* one-hot(c)*mask -> [B][ncls] fp32 (new_scripy.py:334-340); flip != 0 uses -(1-mask) (MNIST_script.py:170) */
int dm_onehot_mask(const int64_t* c, const float* mask, float* out, int B, int ncls, int flip, dm_stream_t s);
/* NCHW fp32 -> NHWC dtype with channel pad to Cp (zeros), optional batch repeat (CFG doubling) */
int dm_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cp, int repeat, dm_stream_t s);
int dm_nhwc_to_nchw(const void* x, float* y, int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s);
int dm_cast(const void* x, void* y, int from_dtype, int to_dtype, int64_t n, dm_stream_t s);
/* FiLM: y = cemb[b,c]*x + temb[b,c] (new_scripy.py:348-349) */
int dm_film_fwd(const void* x, const float* cemb, const float* temb, void* y, int dtype, int B, int HW, int C, dm_stream_t s);
int dm_film_bwd(const void* x, const void* dy, const float* cemb, void* dx, float* dcemb, float* dtemb, int dtype, int B,
                int HW, int C, dm_stream_t s);
/* concat(x1,x2) along C then bilinear x2 upsample, align_corners=True (new_scripy.py:242,251) */
int dm_upcat_fwd(const void* x1, const void* x2, void* y, int dtype, int B, int H, int W, int C1, int C2, dm_stream_t s);
/* the same with x2 holding B2 samples (B2 | B): sample b reads x2[b % B2] — the CFG sampler's doubled batch over skip tensors
 * that the encoder computed once; forward only */
int dm_upcat_fwd_bcast(const void* x1, const void* x2, void* y, int dtype, int B, int B2, int H, int W, int C1, int C2, dm_stream_t s);
int dm_upcat_bwd(const void* dy, void* dx1, void* dx2, int dtype, int B, int H, int W, int C1, int C2, dm_stream_t s);
/* plain channel concat / split (MNIST UnetUp, MNIST_script.py:95) */
int dm_cat_fwd(const void* x1, const void* x2, void* y, int dtype, int M, int C1, int C2, dm_stream_t s);
int dm_cat_bwd(const void* dy, void* dx1, void* dx2, int dtype, int M, int C1, int C2, dm_stream_t s);
/* AvgPool k (whole-window, stride k) + GELU (new_scripy.py:290) -> fp32 [B][Ho*Wo][C] */
int dm_avgpool_gelu_fwd(const void* x, float* y, int dtype, int B, int H, int W, int C, int k, dm_stream_t s);
int dm_avgpool_gelu_bwd(const void* x, const float* dy, void* dx, int dtype, int B, int H, int W, int C, int k, dm_stream_t s);
/* MaxPool2d(2) (MNIST_script.py:74) */
int dm_maxpool2_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C, dm_stream_t s);
int dm_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, int B, int H, int W, int C, dm_stream_t s);
/* out = (x ? x : 0) + y * [mask[b,pix] > thresh]  (LocalEnhancer, new_scripy.py:172-174; x may be NULL) */
int dm_mask_axpy(const void* x, const void* y, const float* mask, float thresh, void* out, int dtype, int64_t npix, int C, dm_stream_t s);
/* y = a + b (skip-connection gradient joins) */
int dm_add(const void* a, const void* b, void* y, int dtype, int64_t n, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * DDPM wrapper pieces
 * ---------------------------------------------------------------------------------------------- */
/* x_t = sqrtab[ts]*x + sqrtmab[ts]*noise, NCHW fp32 in -> NHWC dtype (Cp channels) out (new_scripy.py:408-411) */
/* The per-sample draws of DDPM.forward (new_scripy.py:405 `_ts = randint(1, n_T+1)`, :413 `bernoulli(1 - drop_prob)`, :415 `_ts / n_T`)
 * on the device: ts[b] in 1..n_T, t_frac[b] = ts[b] / n_T (fp32), keep[b] in {0, 1} with P(1) = keep_prob — Philox4x32-10 under `seed`
 * at stream offset *offset_dev + 1, which the launch stores back (the noise drawn next with dm_randn_dev uses the same offset,
 * a disjoint counter range).  A captured train step therefore draws fresh timesteps / masks on every replay. */
int dm_draw_ts_keep(int64_t* ts, float* t_frac, float* keep, int B, int n_T, float keep_prob, uint64_t seed, uint64_t* offset_dev,
                    dm_stream_t s);
int dm_qsample(const float* x, const float* noise, const int64_t* ts, const float* sqrtab, const float* sqrtmab,
               void* xt, int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s);
/* weighted MSE + masked L1 (new_scripy.py:417-437); pred/noise NCHW fp32, mask [B][H][W];
 * loss (one float, must be zeroed by the caller) ; thresholds/weights: {hi_t, mid_t, hi_w, mid_w, lo_w, feat_w} */
int dm_loss_fwd(const float* pred, const float* noise, const float* mask, const float* cfg6, float* loss, int B, int C,
                int H, int W, dm_stream_t s);
/* d_pred as NHWC dtype with Cp channels (pads zero), or NCHW fp32 when Cp == 0; scaled by *gscale (device scalar) */
int dm_loss_bwd(const float* pred, const float* noise, const float* mask, const float* cfg6, const float* gscale,
                void* dpred, int dtype, int B, int C, int H, int W, int Cp, dm_stream_t s);
/* plain MSE for the MNIST ancestor (MNIST_script.py:252); mask == NULL in dm_loss_* selects it */

/* CFG combine + ancestral update (new_scripy.py:468-475):
 *   eps = (1+w)*eps[0:n] - w*eps[n:2n];  x = a_i*(x - eps*b_i) + s_i*z ; i = *step (device int32), z = 0 when i == 1.
 *   tables: oneover_sqrta, mab_over_sqrtmab, sqrt_beta_t.  z == NULL -> Philox(seed, i) in-kernel N(0,1).
 *   After the update *step is decremented when dec_step != 0 (hipGraph replay needs no host argument). */
int dm_cfg_update(float* x, const float* eps2n, const float* z, float guide_w, const float* oneover_sqrta,
                  const float* mab_over_sqrtmab, const float* sqrt_beta_t, int32_t* step, uint64_t seed, int64_t n_elems,
                  int dec_step, dm_stream_t s);
/* the same update on elements [first_elem, first_elem + n_elems) of a larger batch: the in-kernel noise is the slice of the
 * whole batch's Philox stream, so sampling sharded over ranks reproduces the single-process images (sampling) */
int dm_cfg_update_slice(float* x, const float* eps2n, const float* z, float guide_w, const float* oneover_sqrta,
                        const float* mab_over_sqrtmab, const float* sqrt_beta_t, int32_t* step, uint64_t seed, int64_t n_elems,
                        int64_t first_elem, int dec_step, dm_stream_t s);
/* t[b] = *step / n_T for b < B (feeds the time embedding inside a captured loop) */
int dm_fill_t(float* t, const int32_t* step, int n_T, int B, dm_stream_t s);
/* N(0,1) fill by Philox4x32-10 (used for noise when the caller does not inject it) */
int dm_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, dm_stream_t s);
/* elements [first_elem, first_elem + n) of that stream (first_elem a multiple of 4) */
int dm_randn_slice(float* out, int64_t n, uint64_t seed, uint64_t offset, int64_t first_elem, dm_stream_t s);
/* same stream, the offset read from device memory at execution time (a captured hipGraph draws fresh noise every replay) */
int dm_randn_dev(float* out, int64_t n, uint64_t seed, const uint64_t* offset_dev, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Optimiser: clip_grad_norm_(1.0) + AdamW (new_scripy.py:715-719, 798, 801) on flat fp32 buffers.
 * ---------------------------------------------------------------------------------------------- */
int dm_sumsq(const float* g, int64_t n, float* out /* one float, += */, dm_stream_t s);
/* clip coefficient = min(1, max_norm/(sqrt(*sumsq)*gscale + 1e-6)); gscale folds 1/world etc.
 * hyper = {lr, beta1, beta2, eps, weight_decay, max_norm, gscale, bias_corr1, bias_corr2} (device) */
/* p_bf16 (may be NULL): bf16 shadow of the parameters, refreshed in the same pass (p_bf16[i] = bf16(p[i])) — the conv
 * kernels read it as their packed forward weights, so no per-layer cast launch follows an optimiser step;
 * step_dev (may be NULL): device-resident step count t; when given, the bias corrections 1 - beta^t are computed in the
 * kernel and hyper9[7..8] are ignored (lets a captured hipGraph of the train step be replayed) */
int dm_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, const float* hyper9, void* p_bf16,
             const int64_t* step_dev, dm_stream_t s);
/* dm_adamw with a 16-bit parameter shadow of either kind (p16_dtype DM_BF16 / DM_F16) and dynamic loss scaling
 * (the reference trains under torch.cuda.amp.autocast() + GradScaler: new_scripy.py:390, 784-802):
 * scaler4 (device, may be NULL) = {scale, growth tracker, found_inf of this step, 1 / scale of this step's gradients}.
 * dm_scaler_update — GradScaler.step()'s found-inf test + GradScaler.update() on the device, between dm_sumsq and
 * dm_adamw_scaled: found_inf = sumsq is inf / nan -> the step is skipped (step_dev taken back by one), scale *= backoff,
 * tracker = 0; else tracker += 1 and every `interval` clean steps scale *= growth (torch defaults: 2^16, 2.0, 0.5, 2000). */
int dm_adamw_scaled(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, const float* hyper9, void* p16,
                    int p16_dtype, const int64_t* step_dev, const float* scaler4, dm_stream_t s);
int dm_scaler_update(float* state4, const float* sumsq, int64_t* step_dev, float growth, float backoff, int interval, dm_stream_t s);


/* Evaluation helpers of the drivers (new_scripy.py:1188-1250): per image pair (a_i, b_i), n_per_image floats each,
 * out[i] = {sum a, sum b, sum a^2, sum b^2, sum ab, min a, min b, n} as 8 doubles — global-statistics SSIM and PSNR
 * follow on the host (diffusionmodel_amd/metrics.py). */
int dm_image_moments(const float* a, const float* b, double* out, int n_images, int64_t n_per_image, dm_stream_t s);
/* Attention mask of CrackDataset (new_scripy.py:533-546): `lo` everywhere, `mid` on rows >= S/2, `hi` inside the
 * half-open box boxes[b] = {x0, y0, x1, y1} (already scaled/clamped on the host). out [B][S][S]. */
int dm_attn_mask(const int32_t* boxes, float* out, int B, int S, float lo, float mid, float hi, dm_stream_t s);

/* multi-tensor fp32 copy/add: table_dev[e] = {src_ptr, dst_ptr, count}; gathers the small parameters'
 * gradients into the flat gradient buffer in one launch */
int dm_scatter_copy(const int64_t* table_dev, int n_entries, int add, dm_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Launch plans: one step of the reference's hot loops (new_scripy.py:777-803 train step; :457-475 sampling step) recorded once
 * and re-issued from C as PLAIN stream launches.  The caller captures the step on a stream (hipStreamBeginCapture, or torch's
 * CUDAGraph with keep_graph=True) — the captured hipGraph is never instantiated — and hands the hipGraph_t over:
 * dm_plan_from_graph reads every kernel / memset node back (function, grid, block, argument block) in dependency order.
 * Nodes stay owned by the graph: keep it (and the memory pool of the capture) alive as long as the plan.
 * memcpy / host / child-graph nodes are refused (DM_EUNSUPPORTED): copies inside a planned step go through library kernels;
 * so are kernel nodes without a kernelParams array (arguments in `extra`).  A failing launch makes dm_plan_run return with the
 * failing op's index and name in dm_last_error(); the ops before it HAVE been issued.
 *   dm_plan_marker          launched INSIDE the capture: ends a segment; the host may act between segments on replay
 *                           (the data-parallel all-reduce of a finished gradient bucket); not replayed itself
 *   dm_plan_info            info[6] = {ops, kernel launches, memsets, markers, skipped ordering-only nodes, segments}
 *   dm_plan_segment_marker  id of the marker that ends segment `seg` (-1 for the last segment)
 *   dm_plan_op_name         kernel name of op `idx` into buf (diagnostics); returns 0 kernel / 1 memset / 2 marker / -1
 *   dm_plan_run             issue segments [seg_first, seg_last] on `stream`
 *   dm_plan_run_timed       dm_plan_run with a HIP event pair on `stream` around every kernel whose name contains one of the
 *                           '|'-separated substrings of `substr` ("" = every kernel)
 *   dm_plan_timed_results   synchronises `stream`; ms_out[i] / op_out[i] = duration and op index of the i-th timed launch since
 *                           the last call                                                                                    */
int dm_plan_marker(int id, dm_stream_t stream);
int dm_plan_from_graph(void* hip_graph, void** plan_out);
int dm_plan_info(void* plan, int32_t* info6);
int dm_plan_segment_marker(void* plan, int seg);
int dm_plan_op_name(void* plan, int idx, char* buf, int cap);
int dm_plan_run(void* plan, int seg_first, int seg_last, dm_stream_t stream);
int dm_plan_run_timed(void* plan, int seg_first, int seg_last, const char* substr, dm_stream_t stream);
int dm_plan_timed_results(void* plan, float* ms_out, int32_t* op_out, int cap, int32_t* n_out, dm_stream_t stream);
int dm_plan_destroy(void* plan);

/* Test aid (LDS residue audit): fills all 160 KiB of EVERY CU's LDS with `pattern` (one 160-KiB workgroup per CU).
 * Launched before a kernel under test with a NaN bit pattern, it turns any read of LDS bytes that kernel did not write itself into
 * NaNs in an exact-integer result.  Not used by the product path.                                                               */
int dm_debug_poison_lds(uint32_t pattern, dm_stream_t stream);

/* Zero many float ranges in one launch: table[e] = {pointer, count} (int64 pairs on the device, count in floats, pointers 16-byte aligned,
 * counts multiples of 4) — FusedAdamW.zero_grad for the gradient ranges no overwriting weight-gradient launch will write (r04). */
int dm_zero_ranges(const int64_t* table_dev, int n_entries, dm_stream_t stream);

/* ---- data-parallel collective (allreduce_bucket(ptr, n, dtype, comm, stream)) -----------------------------
 * The reference has no distributed code; what the collective must preserve is its gradient accumulation (new_scripy.py:786,
 * 795-803): rank == micro-batch, gradients SUMMED over the ranks, 1/world applied by the optimiser (hyper9[6] of dm_adamw),
 * clipping on the reduced gradient.  One process per GPU; RCCL over xGMI; librccl.so is bound at run time (dlopen): the path given
 * to dm_comm_load, else $DM_RCCL_LIB, else a librccl the process has already mapped (a torch process: torch's own), else the
 * loader's default search.  A binder that owns no torch.distributed drives data parallelism with these five calls alone:
 *   rank 0: dm_comm_unique_id(id) -> ships the 128 bytes to the other ranks by its own means (file, socket, MPI ...)
 *   every rank: dm_comm_init(&comm, world, rank, id); per step and gradient bucket: dm_allreduce_bucket(grad + lo, n, DM_F32, comm, s)
 *   — in place, SUM, asynchronous on `s` (ordered like a kernel launch on that stream) — ; dm_comm_destroy(comm).
 * Errors: DM_EUNSUPPORTED when librccl.so cannot be found / lacks the nccl* entry points, -1000 - ncclResult_t for an RCCL failure
 * (its text in dm_last_error()).                                                                                              */
int dm_comm_load(const char* librccl_path);
int dm_comm_unique_id(void* id128);                                                     /* out: 128 bytes (ncclUniqueId) */
int dm_comm_init(void** comm_out, int world, int rank, const void* id128);
int dm_allreduce_bucket(void* ptr, int64_t n, int dtype, void* comm, dm_stream_t stream);
int dm_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* DM_AMD_H */
