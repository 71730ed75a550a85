/* vqhip.h — C ABI of libvqhip.so: the MI355X (gfx950) kernels behind the VAE/VQGAN train step.
 *
 * The reference (cloneofsimo/vqgan-training) has NO native code and no FFI: every entry point
 * below replaces a PyTorch library call that the reference makes on its hot path.  Each
 * declaration cites the reference call site (file:line in /root/reference) it stands in for.
 * The Python host (vqgan-training_amd/{ae,utils,vae_trainer}.py) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - Plain C: pointers + sizes + POD descriptors; no torch / C++ types.
 *  - All pointers are DEVICE pointers owned by the caller (PyTorch's caching allocator in our
 *    host code).  The library never allocates, frees or synchronises; it only enqueues on the
 *    `stream` argument (a hipStream_t passed as void*).
 *  - Activations are NHWC ("pixel-major"): [N][H][W][C] with C a multiple of 8, padded channels
 *    hold zeros.  dtype VQ_BF16 (2 bytes, raw bfloat16), VQ_F16 (2 bytes, IEEE binary16) or VQ_F32.
 *  - VQ_F16 is the storage / MFMA operand type of the "reference precision" mode: the reference runs its encoder, LPIPS and
 *    discriminator in fp32 with TF32 matmuls (vae_trainer.py:18-19,538; utils.py:70-71) and gfx950 has no TF32 MFMA, so those
 *    modules run on binary16 operands — TF32's 10-bit mantissa — with fp32 accumulation.  binary16 has 5 exponent bits, so
 *    the tensors that leave [2^-14, 65504] carry a power-of-two scale: packed WEIGHTS are stored times s_w (a power of two
 *    from the tensor's measured |w|max, see vq_pack_weight_*), GRADIENT tensors times the caller's loss scale; the scales are
 *    undone in the kernels' fp32 epilogues (`alpha`, `alpha_dev` below: exact multiplications).  Stores saturate at +-65504.
 *  - VQ_F16X2 (4 bytes per element) is the storage / operand type of the "fp32-tolerance" mode ("f16x3"): every value v is kept as
 *    TWO binary16 numbers, hi = rn16(v) and lo = rn16(v - hi) — 22 significand bits, against fp32's 24 — laid out per group of 8
 *    channels as 16 bytes of hi followed by 16 bytes of lo: a [N][H][W][C] tensor is byte for byte a binary16 tensor of 2C "virtual"
 *    channels [h0..h7 l0..l7 h8..h15 l8..l15 ...], which is what lets the LDS-DMA implicit-GEMM kernels stream it unchanged.  A
 *    product of two such operands is formed on the binary16 MFMA as hi*hi + hi*lo + lo*hi (three MFMAs, fp32 accumulate; the dropped
 *    lo*lo term is <= 2^-22 relative): ~2^-21 relative per product, where the reference's CPU path — the parity target of
 *    BASELINE.json's north_star, vae_trainer.py:525-708 under a no-op autocast — has fp32's 2^-24.  Scales and range events are
 *    VQ_F16's (weights times s_w, gradients times the loss scale; the hi half saturates at +-65504).  The 22 bits hold for |v| >= 2^-3:
 *    the lo piece bottoms out at binary16's smallest subnormal, so the ABSOLUTE resolution of a stored value never goes below 2^-25
 *    (a value of 2^-10 keeps ~15 bits, values below 2^-25 vanish) — activations are stored unscaled, so tensors that live far below
 *    1 want the fp32-storage split modes (VqConvDesc.split 3 / 6) instead.
 *  - Return value: 0 on success, negative VqStatus on failure; vq_last_error() returns a
 *    thread-local message.  Unsupported shapes fail loudly — there is no fallback path.
 *  - Re-entrant; callable from any host thread with the device already current (the autograd
 *    engine thread calls the backward entry points).  The library holds NO mutable process-global state: what selects a
 *    kernel travels in the call (VqConvDesc.kernel_hint), what a kernel reports goes to caller-owned device memory.
 *  - Range events (VQ_F16 only).  A binary16 store saturates silently; the reference's fp32/TF32 path cannot overflow
 *    (vae_trainer.py:18-19,538), so the entry points that WRITE a binary16 tensor of a loss-scaled stack take `range_events`:
 *    a DEVICE pointer to (at least) 3 int32 counters owned by the caller, or NULL.  A kernel adds
 *        [0] += 1 per wave that stored a value beyond +-65504 (or an inf / NaN): the tensor was clipped,
 *        [1] += 1 per wave whose stored values ALL flushed to zero although some were non-zero in fp32: a region vanished,
 *        [2] += 1 per wave that stored a magnitude >= 2^13 (ABI v9): three bits of headroom left — the caller's cue to lower a loss
 *                 scale before anything clips.
 *    Nothing is written in a healthy step (no atomics).  vq_adamw_multi can be told to skip its update when such counters are
 *    non-zero (`skip_flags`), so a step that saw a clipped gradient never reaches the parameters — without a host sync.
 */
#ifndef VQHIP_H_
#define VQHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared between this push and the pop at the end of the file
 * are its ONLY dynamic symbols (tests/test_abi.py compares `nm -D` with this header, name by name). */
#pragma GCC visibility push(default)

enum VqDtype { VQ_BF16 = 0, VQ_F32 = 1, VQ_F16 = 2, VQ_F16X2 = 3 };
enum VqStatus { VQ_OK = 0, VQ_ERR_INVALID = -1, VQ_ERR_UNSUPPORTED = -2, VQ_ERR_HIP = -3, VQ_ERR_WORKSPACE = -4 };

const char* vq_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int vq_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA  (replaces nn.Conv2d forward/backward:
 * ae.py:105-117,133-139 ResnetBlock convs + nin_shortcut; ae.py:146-154 Downsample;
 * ae.py:160-166 Upsample; ae.py:197-199,230-232,282-284,307-309 conv_in/conv_out;
 * utils.py:95-111,148-154 VGG16 features; utils.py:156-185 PatchDiscriminator heads).
 *
 * Virtual input coordinate of output pixel (oy,ox), tap (r,s):
 *     vy = oy*stride + r - pad_t ,  vx = ox*stride + s - pad_l
 * the tap contributes iff vy % dil_in == 0 (same for x) and iy = (vy / dil_in) >> (up==2) lies in
 * [0,H).  `up`=2 folds nn.functional.interpolate(scale 2, nearest) (ae.py:165) into the
 * gather; `dil_in`>1 expresses the data-gradient of a strided conv as a conv over the
 * zero-dilated output gradient.  Bottom/right padding is implicit in the bounds check, which is
 * how Downsample's asymmetric F.pad(0,1,0,1) (ae.py:151-152) is handled without a copy.
 */
typedef struct VqConvDesc {
  int32_t N, H, W, Cin;   /* input  [N][H][W][Cin]  (Cin  = padded channel count, %8 == 0)      */
  int32_t Ho, Wo, Cout;   /* output [N][Ho][Wo][Cout] (Cout = padded channel count, %8 == 0)     */
  int32_t Cin_w, Cout_w;  /* true channel counts of the OIHW weight                             */
  int32_t R, S;           /* kernel height, width                                               */
  int32_t stride, dil_in, up;
  int32_t pad_t, pad_l;
  int32_t dtype;          /* VqDtype of x / y / residual / mask                                 */
  int32_t split;          /* 1: bf16 operands, fp32 accumulate; fp32 storage only: 3 = two bf16 pieces per operand, three products (~2^-16);
                           * 6 = three pieces (all 24 mantissa bits), six products: fp32-exact products */
  int32_t relu;           /* epilogue max(.,0)  (VGG: utils.py:95-111 Conv+ReLU pairs)          */
  int32_t subpix;         /* 0, or 2: sub-pixel (phase-decomposed) convolution, see below        */
  float alpha;            /* the fp32 accumulator is multiplied by alpha * (alpha_dev ? *alpha_dev : 1) before bias /   */
  int32_t kernel_hint;    /* residual (vq_conv2d_fwd) or before it is written / accumulated (vq_conv2d_wgrad: dW and    */
  const float* alpha_dev; /* dbias).  alpha == 0 means 1 (a zero-initialised descriptor scales nothing).  alpha_dev is  */
                          /* a DEVICE scalar: the 1/s_w slot of a VQ_F16 packed weight (vq_pack_weight_*).              */
  int32_t* range_events;  /* vq_conv2d_fwd, dtype VQ_F16: the range-event counters of y's stack (see Conventions), or NULL */
  const struct VqGnBwdFuse* gn_bwd;   /* vq_conv2d_fwd as a data gradient whose output is the dy of a GroupNorm(+SiLU): see below, or NULL */
} VqConvDesc;
/* Fused GroupNorm-backward sums (ABI v7).  The data gradient of the conv that FOLLOWS a GroupNorm(+SiLU) (ae.py:41-53,13-14 inside
 * ResnetBlock, ae.py:96-146) produces exactly the dy that GroupNorm's backward starts from, and that backward's first pass reads x and
 * dy again only to form, per channel, sum(dg) and sum(dg * xhat) with dg = dy * silu'(gamma * xhat + beta), xhat = (x - mean) * rstd.
 * With `gn_bwd` set, vq_conv2d_fwd forms these sums in its epilogue — x is read like a residual operand, dy never leaves the
 * registers for this purpose — and writes them as partial rows; vq_gn_silu_bwd then takes the rows (`part_in`) and skips its
 * reduction pass: four tensor passes instead of five.  Requires vq_conv2d_gnb_rows(desc) > 0, no residual / relu_mask /
 * gn_partials / relu on the same call.  MEASURED: cheaper than the pass it replaces per layer, 0.9 % slower in the full step, and
 * its presence in the shared epilogue cost the default path 1.5 % — so only `make ABLATE=1` libraries carry the path; a release
 * library answers vq_conv2d_gnb_rows() = 0 (and VQ_ERR_UNSUPPORTED to a non-NULL gn_bwd): callers keep the reduction pass. */
typedef struct VqGnBwdFuse {
  const void* x;                      /* the GroupNorm's INPUT, [N][Ho][Wo][Cout] like y, dtype like y             */
  const float* mean; const float* rstd;   /* [N][groups]: the statistics of the forward pass                       */
  const float* gamma; const float* beta;  /* [Cout]                                                                */
  float* part;                        /* out: [N][rows][Cout][2] = (sum dg, sum dg * xhat) per row, rows = vq_conv2d_gnb_rows */
  int32_t groups;
  int32_t silu;                       /* 1: GroupNorm + SiLU, 0: GroupNorm alone                                   */
} VqGnBwdFuse;
/* kernel_hint: 0 = the library chooses the kernel — what a product caller passes, always.  Non-zero values force one of the SHIPPED
 * kernels where the shape admits it (tests reach every instantiation at small shapes that way; tools A/B two kernels on one shape);
 * being part of the descriptor they also steer vq_conv_weight_layout / vq_conv2d_gn_tile / vq_conv2d_wgrad_workspace consistently.
 *   vq_conv2d_fwd:   bits 0-2: 1 = 128x128 tiles, 2 = 32x128 tiles, 3 = the 256x256 tile, 5 = nine-tap kernel wherever eligible,
 *                    6 = no three-tap / nine-tap kernel, 7 = three-tap kernel wherever eligible; +8 = weights staged through LDS;
 *                    +(512 << 4) = the one-tap 256x256 tile where the patch-staged one would run; +(16 << 4) = 128-pixel tiles where
 *                    the short-M rule (<= one 64x128 block per CU) picks 64-pixel ones; +(40 << 4) = the one-tap 32-row tile where the
 *                    nine-tap kernel would serve a layer of <= 32 output channels (hint 5 forces that one at any size);
 *                    +(48 << 4) = the generic tile kernels where the persistent patch-conv data-gradient kernel would run,
 *                    +(56 << 4) = that kernel at any size.
 *   vq_conv2d_wgrad: 64 / 128 / 256 = that one-tap LDS-DMA tile, +4 = never the three-tap kernel, +1 = the 4 B/lane split
 *                    reduction, +16 = the three-tap kernel with unstaggered staging; bits 16-31 = forced split-K count (0 = planned).
 * Any other value selects a compile-time ablation / pricing knob that exists only in `make ABLATE=1` builds: a release library
 * answers VQ_ERR_UNSUPPORTED. */

/* Sub-pixel mode (`subpix` = 2, vq_conv2d_fwd only).  The Cout rows are 4 phase blocks (a,b), a,b in {0,1}, of
 * Cout/4 channels each, block index a*2+b.  Block (a,b) of output pixel (oy,ox) is computed with its window moved
 * by (a,b) — vy = oy*stride + r - pad_t + a, vx likewise with b — and is stored at channel block 0 of pixel
 * (2*oy+a, 2*ox+b) of the [N][2*Ho][2*Wo][Cout/4] tensor `y` (bias has Cout/4 entries, shared by the phase blocks; residual /
 * relu_mask are read at the same place).  With the tap sums of vq_subpixel_weights this runs
 *   - nearest-2x upsample + 3x3 conv (Upsample.forward, ae.py:164-166) as four 2x2 convs of the LOW-resolution
 *     input: 16 instead of 36 multiply-accumulates per input pixel, channel pair and phase group;
 *   - the data gradient of a 3x3 / stride-2 conv (Downsample, ae.py:150-154) as four 2x2 convs over dy instead of
 *     nine taps over the zero-dilated dy (three quarters of which multiply zeros).
 * Requires up == 1, dil_in == 1 and Cout/4 a multiple of 32. */

/* Packed-weight layout the kernel chosen for descriptor `d` expects: 0 = row-major [rows][Kp],
 * 1 = MFMA-fragment order (rows padded to 32; 1-KiB blocks of 32 rows x 16 k in a-operand lane order,
 * fetched straight into registers by the bf16 implicit-GEMM kernels).  Pass it to the pack calls. */
int vq_conv_weight_layout(const VqConvDesc* d);

/* Elements (bf16 units, i.e. 2 bytes each) of a packed weight buffer for `rows` output rows and
 * reduction length R*S*cin_pad; all planes of the split=3 (two) / split=6 (three) formats are included. */
size_t vq_packed_weight_elems(int rows_pad, int R, int S, int cin_pad, int split, int layout);

/* OIHW fp32 master weight -> packed 16-bit operand [Cout_pad][Kp] (K = (r*S+s)*Cin_pad + c, zero padded),
 * + a "lo" plane when split==3, "mid" and "lo" planes when split==6.  Forward operand of vq_conv2d_fwd.
 * op_dtype VQ_BF16: bf16 values, `scale` unused (may be NULL).
 * op_dtype VQ_F16 (split 1 only): binary16 values of w * s_w;  `scale` -> 4 DEVICE floats owned by the caller that the
 * call fills: {|w|max, s_w, 1/s_w, 0} with s_w the power of two that puts |w|max * s_w in [2^14, 2^15) (1 for an all-zero
 * tensor).  Pass scale + 2 as VqConvDesc.alpha_dev of the launches that consume the operand. */
int vq_pack_weight_fwd(const float* w_oihw, int Cout_w, int Cin_w, int R, int S,
                       int Cout_pad, int Cin_pad, int split, int layout, int op_dtype, float* scale, void* packed,
                       void* stream);
/* Same master weight -> operand of the data-gradient conv: rows = Cin, taps rotated 180°,
 * K = (r*S+s)*Cout_pad + co.  */
int vq_pack_weight_dgrad(const float* w_oihw, int Cout_w, int Cin_w, int R, int S,
                         int Cout_pad, int Cin_pad, int split, int layout, int op_dtype, float* scale, void* packed,
                         void* stream);

/* Derived 3x3 weights of the phase-decomposed convolutions (fp32 OIHW in, fp32 OIHW-shaped out; `w` is [O][I][3][3]):
 *   mode 0  Upsample forward (subpix conv over x):   out [4*O][I][2][2], row (a*2+b)*O + o:
 *           out[.][i][u][v] = sum of w[o][i][r][s] over r in Ra(u), s in Rb(v);  R0 = {0},{1,2};  R1 = {0,1},{2}
 *   mode 1  Upsample data gradient = a plain 4x4 / stride-2 / pad-1 conv over dy:   out [I][O][4][4],
 *           out[i][o][ky][kx] = sum of w[o][i][r][s] over r in T(ky), s in T(kx);  T = {2},{1,2},{0,1},{0}
 *   mode 2  Downsample data gradient (subpix conv over dy):   out [4*I][O][2][2], row (a*2+b)*I + i:
 *           out[.][o][u][v] = w[o][i][r][s] with r = D_a(u), s = D_b(v);  D0 = 2,0;  D1 = 1,none (zero tap)
 * The results are ordinary conv weights: pack them with vq_pack_weight_fwd. */
int vq_subpixel_weights(const float* w_oihw, float* out, int O, int I, int mode, void* stream);
/* Weight gradient of the Upsample conv through its transposed form: run vq_conv2d_wgrad on the 4x4 / stride-2 / pad-1
 * descriptor of mode 1 with the roles swapped (x := dy [N][2H][2W][O], dy := the conv's input [N][H][W][I]) to get
 * dw4 [I][O][4][4] (16 instead of 36 multiply-accumulates per input pixel and channel pair), then fold it onto the 3x3 taps:
 * dw[o][i][r][s] (+)= sum of dw4[i][o][ky][kx] over ky in K(r), kx in K(s);  K = {2,3},{1,2},{0,1}.  accumulate != 0 adds. */
int vq_subpixel_wgrad_fold(const float* dw4, float* dw_oihw, int O, int I, int accumulate, void* stream);

/* ---- AttnBlock self-attention --------------------------------------------------- */
/* F.scaled_dot_product_attention as AttnBlock.attention uses it: ae.py:74-90 (tokens = the H*W pixels, heads of
 * head_dim = 64 channels, "b (h d) x y -> b h (x y) d") and tae.py:24-53 (tokens = the T*H*W voxels, 8 heads of
 * head_dim = C/8 channels); scale 1/sqrt(head_dim), no mask.  qkv: channels-last output of the 1x1 qkv conv,
 * [N][T][3C] (q | k | v channel blocks of qkv.chunk(3, dim=1)); out: [N][T][C]; lse: [N * C/head_dim][T] fp32, kept
 * for the backward.  head_dim in {8, 16, 32, 64}, C % head_dim == 0.  fp32 arithmetic whatever the storage dtype. */
int vq_attention_fwd(const void* qkv, void* out, float* lse, int N, int T, int C, int head_dim, int dtype, void* stream);
size_t vq_attention_workspace(int N, int T, int C, int head_dim);
/* dqkv [N][T][3C] = gradient of the same op given dout [N][T][C] (and the forward's qkv, out, lse). */
int vq_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int N, int T, int C,
                     int head_dim, int dtype, void* workspace, size_t ws_bytes, void* stream);

/* ---- input preparation ------------------------------------------------------ */
/* Wavelet front-end of the encoder: utils.py:229-247 wavelet_transform_multi_channel (zero-pad 2, the four
 * fixed 6x6 filters of utils.py:206-219, stride 2, per channel; output channel c*4+f).  x: NCHW fp32
 * [N,C,H,W], H and W even.  out_nhwc=1: y = NHWC [N,H/2,W/2,Cpad] of `dtype`, channels >= 4C zeroed (what
 * encoder.conv_in consumes, ae.py:189-194,240);  out_nhwc=0: y = NCHW fp32 [N,4C,H/2,W/2] (Cpad, dtype unused). */
int vq_wavelet_fwd(const float* x_nchw, void* y, int N, int C, int H, int W, int Cpad, int dtype, int out_nhwc,
                   void* stream);
/* torch.flip along H and/or W of an NCHW fp32 tensor with the sign change the trainer applies to latent
 * channels [neg_c0, neg_c1) (vae_trainer.py:534-536, 567-575, 664-671).  An involution: its own backward. */
int vq_flip_nchw(const float* x, float* y, int N, int C, int H, int W, int flip_h, int flip_w, int neg_c0, int neg_c1,
                 void* stream);
/* F.interpolate(x, size=(H/k, W/k), mode="area") for an integer ratio k (vae_trainer.py:531-533): k x k mean. */
int vq_area_downsample_nchw(const float* x, float* y, int N, int C, int H, int W, int k, void* stream);

/* One weight re-pack as data: vq_pack_job fills a job for (w, layout, fwd|dgrad) exactly as vq_pack_weight_* would
 * run it; the caller copies an array of jobs — with block_start = running sum of vq_pack_job_blocks() — to the
 * device once, and vq_pack_weights_multi re-packs ALL of them in a single launch after every optimizer step
 * (the reference re-reads its fp32 weights in every cuDNN call instead; vae_trainer.py:659,702). */
#define VQ_PACK_ELEMS_PER_BLOCK 4096
typedef struct VqPackJob {
  const float* w;          /* OIHW fp32 master weight */
  void* out;               /* packed bf16 operand */
  int64_t total;           /* elements of one plane of `out` */
  int64_t block_start;     /* first block of this job in the multi launch */
  int32_t Cout_w, Cin_w, R, S;
  int32_t rows_pad, kch_pad, Kp;
  int32_t split, dgrad, layout;
  int32_t tiled;           /* 1: 32x32-channel tiles through LDS (coalesced both sides), 0: element-wise */
  int64_t n_units;         /* blocks this job occupies in the multi launch (tiles, or 4096-element chunks) */
  int32_t op_dtype;        /* VQ_BF16 or VQ_F16 (see vq_pack_weight_fwd) */
  int32_t reserved0;
  float* scale;            /* VQ_F16: the operand's 4-float scale slot (device) */
} VqPackJob;
int vq_pack_job(VqPackJob* job, const float* w_oihw, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                int split, int layout, int dgrad, int op_dtype, float* scale, void* packed);
int64_t vq_pack_job_blocks(const VqPackJob* job);
/* with_scales != 0: some jobs are VQ_F16 — their |w|max is re-measured first (two extra small launches). */
int vq_pack_weights_multi(const VqPackJob* jobs_dev, int n_jobs, int64_t total_blocks, int with_scales, void* stream);

/* y = conv(x, W) [+ bias] [+ residual] [relu] [; y = 0 where relu_mask <= 0]
 * (conv forward, and — with a dgrad-packed weight and the mirrored descriptor — the data
 * gradient that autograd computes for the same nn.Conv2d).  `residual` implements
 * `x + h` (ae.py:140) in the epilogue; `relu_mask` applies ReLU'(.) of the producing layer.
 * `residual` may be `y` itself (in-place accumulation: every element is read and then written by the
 * same lane) — how the temporal taps of tae.py's nn.Conv3d layers are summed (frames = the batch axis N). */
int vq_conv2d_fwd(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias,
                  const void* residual, const void* relu_mask, void* y, float* gn_partials, int gn_groups, void* stream);
/* GroupNorm statistics of the OUTPUT from the conv epilogue (ae.py:131-135: every FP32GroupNorm of the model reads a tensor a
 * convolution has just written): with gn_partials != NULL the kernel also writes, per (image, partial row, group), the pair
 * (mean of y over the row, M2 = sum (y - that mean)^2) — [N][Ho*Wo / row][gn_groups][2] floats, one row per wave of an output
 * tile, no barrier and no atomics; formed from moments about one of the row's own values, never as E[y^2] - E[y]^2 of raw fp32
 * sums, so that the statistics keep F.group_norm's accuracy when |mean| >> std — and
 * vq_gn_stats_finalize(partials, N, rows = Ho*Wo / row, ...) merges the rows (Chan et al., fp64) into mean / rstd, so the
 * consumer skips its statistics pass over the tensor.  vq_conv2d_gn_tile(d, groups) = pixels per partial row for this descriptor, or 0 when its kernel cannot
 * produce the partials (pass NULL then and run vq_gn_stats). */
int vq_conv2d_gn_tile(const VqConvDesc* d, int groups);
/* Partial rows PER IMAGE that vq_conv2d_fwd writes to VqGnBwdFuse.part for this descriptor (one row per wave of an output tile), or 0
 * when its kernel cannot form the fused GroupNorm-backward sums (fp32 storage, depth-to-space outputs, tiles that straddle images):
 * pass gn_bwd = NULL then and let vq_gn_silu_bwd run its own reduction pass. */
int vq_conv2d_gnb_rows(const VqConvDesc* d);
int vq_gn_stats_finalize(const float* partials, int N, int tiles, int64_t HW, int C, int G, float eps, float* mean, float* rstd,
                         void* stream);

/* dW (OIHW fp32) = sum over pixels dY (x) gather(X); split-K partials live in `workspace`.
 * dbias (may be NULL) receives the bias gradient sum_p dY[p][co] — fused into the MFMA kernel as
 * one extra "times ones" product where the LDS-DMA kernel applies, a column-sum pass otherwise.
 * accumulate != 0 adds into dw / dbias. (autograd wgrad of nn.Conv2d, same call sites as above) */
size_t vq_conv2d_wgrad_workspace(const VqConvDesc* d);
int vq_conv2d_wgrad(const VqConvDesc* d, const void* x, const void* dy, float* dw_oihw, float* dbias,
                    int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* per-channel column sum over pixels: out[c] (+)= sum_p t[p][c]   (bias gradient of the convs;
 * workspace >= vq_colsum_workspace bytes) */
size_t vq_colsum_workspace(int64_t pixels, int C);
/* the sums are multiplied by alpha * (alpha_dev ? *alpha_dev : 1) (the inverse loss scale of a VQ_F16 gradient tensor) */
int vq_colsum(const void* t, int64_t pixels, int C, int dtype, float* out, int n_out, int accumulate, float alpha,
              const float* alpha_dev, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the [B,3,H,W] image / z boundary (the reference is NCHW throughout).
 * nchw_to_nhwc optionally applies ScalingLayer (utils.py:60-71): y = (x - shift[c]) / scale[c].
 * `alpha` multiplies every element (1 in the forward direction; the loss scale where a gradient enters / leaves a VQ_F16 region).
 */
int vq_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype,
                    const float* shift, const float* scale, float alpha, int32_t* range_events, void* stream);
/* inverse; `div_scale` (may be NULL) divides channel c by div_scale[c] (ScalingLayer backward). */
int vq_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cpad, int dtype,
                    const float* div_scale, float alpha, void* stream);
/* out[0] = max(out[0], max |t_i|) over n elements (atomic on the float's bit pattern: zero `out` first).  Calibration of the
 * VQ_F16 loss scales (ops.GradScaleMonitor) — never on the step's critical path. */
int vq_absmax(const void* t, int64_t n, int dtype, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * FP32GroupNorm + swish  (ae.py:41-53 + ae.py:13-14; call sites ae.py:131-135,254-255,330-331)
 * Statistics are fp32 regardless of the storage dtype, eps inside the sqrt, biased variance.
 */
size_t vq_gn_workspace(int N, int64_t HW, int C);
/* mean[N*G], rstd[N*G] */
int vq_gn_stats(const void* x, int N, int64_t HW, int C, int G, float eps, int dtype,
                float* mean, float* rstd, void* workspace, size_t ws_bytes, void* stream);
/* y = silu(gn(x))  (silu != 0) or y = gn(x) */
int vq_gn_silu_fwd(const void* x, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, int N, int64_t HW, int C, int G, int C_w, int dtype, int silu,
                   void* y, void* stream);
/* dx = dx_scale * d(silu∘gn)/dx · dy (+ add);  dgamma/dbeta (+)= pg_scale * reductions.  Each scale is a host factor times an
 * optional DEVICE scalar (`*_dev`, may be NULL): 1 / NULL outside the VQ_F16 mode; there dy carries a loss scale, pg_scale
 * removes it from the parameter gradients and dx_scale re-bases a ResnetBlock's branch gradient onto the skip gradient's
 * scale before the two are added (ops._ResnetBlock). */
int vq_gn_silu_bwd(const void* x, const void* dy, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, const void* add, int N, int64_t HW, int C,
                   int G, int C_w, int dtype, int silu, void* dx, float* dgamma, float* dbeta,
                   int accumulate, float dx_scale, const float* dx_scale_dev, float pg_scale, const float* pg_scale_dev,
                   int32_t* range_events /* of dx (VQ_F16), may be NULL */,
                   const float* part_in /* [N][part_rows][C][2] from vq_conv2d_fwd(gn_bwd), or NULL: reduce here */, int part_rows,
                   void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * VGG16 / LPIPS pieces
 */
/* nn.MaxPool2d(2,2) (torchvision features idx 4,9,16,23 inside utils.py:104-111) */
int vq_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* dx = route(dy) [+ add]: dy goes to the first maximum in row-major scan order (PyTorch CPU tie rule).  `add` (same shape as
 * x, may be NULL, may NOT alias dx) is the other gradient of x where x has a second consumer — every VGG feature map feeds the
 * next slice AND an LPIPS tap / a discriminator head (utils.py:116-131,187-203): autograd would sum the two with a separate
 * elementwise kernel (one more write + read of the tensor, and an unsaturated binary16 add).  range_events: of dx (VQ_F16). */
int vq_maxpool2_bwd(const void* x, const void* dy, const void* add, void* dx, int N, int H, int W, int C, int dtype,
                    int32_t* range_events, void* stream);
/* 2x2 sum pool: backward of nearest-2x upsample (ae.py:165) */
int vq_sumpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);

/* LPIPS tap (utils.py:44-57,134-140): per sample n
 *   val[n] += (1/HW) sum_p sum_c w[c] * m[p][c] * (f0/(|f0|+1e-10) - f1/(|f1|+1e-10))^2
 * f0 = feats[n], f1 = feats[n + N] when f1 == NULL is not used; mask m: NULL (all ones), an
 * explicit float mask [N][HW][C] already scaled by 1/(1-p) (tests inject it), or generated from
 * `seed` (!=0) as Bernoulli(0.5)*2 (nn.Dropout in train mode, utils.py:76-89). */
size_t vq_lpips_workspace(int N, int64_t HW);
int vq_lpips_tap_fwd(const void* f0, const void* f1, const float* w, const float* mask,
                     uint64_t seed, int N, int64_t HW, int C, int dtype, float* val, void* workspace,
                     size_t ws_bytes, void* stream);
/* df0 = alpha * d val / d f0 * gval[n]   (gradient flows only to the reconstruction branch; alpha = the loss scale of a
 * VQ_F16 feature stack, else 1).  relu_inputs != 0: f0 is a ReLU output (VGG taps), the gradient is zeroed where f0 <= 0. */
int vq_lpips_tap_bwd(const void* f0, const void* f1, const float* w, const float* mask,
                     uint64_t seed, const float* gval, int N, int64_t HW, int C, int dtype,
                     int relu_inputs, float alpha, void* df0, int32_t* range_events, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scalar reductions of the loss layer — all results stay on the device.
 */
/* out[0] = sum (|x| - mean|x|)^2, out[1] = sum x^2, out[2] = sum |x|, out[3] = count  over n > 0 fp32 elements
 * (vae_trainer.py:202-216: mean(z^2), mean|z|, std|z| = sqrt(out[0] / (n - 1))).  Accumulated in fp64, |x| about the pivot |x[0]|
 * (ABI 10; before: out[0] = sum x, and std|z| was left to  E[x^2] - E[|x|]^2  of the rounded sums).  scratch: 8-byte aligned. */
int vq_moments(const float* x, int64_t n, float* out4, float* scratch /* >= 1024 floats */, void* stream);
/* GradNorm backward (vae_trainer.py:34-48): norm_out[0] = ||g||_2 over n elements */
int vq_l2norm(const float* g, int64_t n, float* norm_out, float* scratch, void* stream);
/* dx = weight * g / (norm[0] + 1e-8) */
int vq_scale_by_norm(const float* g, const float* norm, float weight, int64_t n, float* dx, void* stream);
/* hinge / bce discriminator statistics (vae_trainer.py:63-90): out = {loss_real, loss_fake,
 * mean_real, mean_fake, n_correct, count}; d_real/d_fake (may be NULL) receive dLoss/dlogit for
 * loss = 0.5*(loss_real+loss_fake). disc_type: 0 = bce, 1 = hinge. */
int vq_gan_disc_loss(const float* real, const float* fake, int64_t n, int disc_type, float* out6,
                     float* d_real, float* d_fake, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW (vae_trainer.py:455-475,702-704; torch.optim.AdamW semantics:
 * p *= 1 - lr*wd;  m,v update;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)).
 */
typedef struct VqAdamTensor {
  float* p; const float* g; float* m; float* v;
  int64_t n;
} VqAdamTensor;
/* `table` is a DEVICE array of n_tensors descriptors; `chunk_offsets` a DEVICE int64 array with
 * n_tensors+1 prefix sums of ceil(n/chunk). */
/* skip_flags (may be NULL): n_flags DEVICE int32 values `skip_stride` elements apart — when any of them is non-zero the launch
 * changes NOTHING (parameters, moments): the step that produced a clipped VQ_F16 gradient (range events, see Conventions) is
 * dropped on the device, no host sync.  The caller keeps its own step counter in line (bias corrections) when it reads the flags. */
int vq_adamw_multi(const VqAdamTensor* table, const int64_t* chunk_offsets, int n_tensors,
                   int64_t total_chunks, int chunk, float lr, float wd, float beta1, float beta2,
                   float eps, float bc1, float bc2, float grad_scale, const int32_t* skip_flags, int n_flags, int skip_stride,
                   void* stream);
/* out = x * alpha * (alpha_dev ? alpha_dev[0] : 1)   (fp32; gradient of the mean(z^2) regulariser,
 * vae_trainer.py:202-209, and other scalar-scaled copies) */
int vq_scale(const float* x, float alpha, const float* alpha_dev, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * VQ codebook nearest lookup (NOT in the reference — SURVEY F1; oracle/vq_oracle.py defines it)
 *   d_ij = |z_i|^2 - 2 z_i.e_j + |e_j|^2  evaluated in fp32 with the fixed sequential order
 *   documented in oracle/vq_oracle.py; idx_i = argmin_j d_ij, lowest index wins ties.
 */
size_t vq_vq_workspace(int64_t n_tokens, int n_codes);
int vq_vq_nearest_fwd(const float* z, const float* codebook, int64_t n_tokens, int n_codes, int dim,
                      int64_t* idx, float* zq, float* min_dist, void* workspace, size_t ws_bytes,
                      void* stream);
/* dcodebook[idx_i] += gq_i  (codebook gradient of the straight-through / codebook loss).  Deterministic: the contributions
 * are accumulated as 64-bit fixed-point integers (order-independent) at a power-of-two scale derived from the measured
 * max |gq| and converted back once; workspace >= vq_vq_scatter_workspace bytes; n_tokens * dim must be a multiple of 8. */
size_t vq_vq_scatter_workspace(int n_codes, int dim);
int vq_vq_scatter_add(const float* gq, const int64_t* idx, int64_t n_tokens, int n_codes, int dim,
                      float* dcodebook, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hardware-layout probe (one wave, one MFMA / LDS transpose read, raw per-lane dump); used by
 * tests/test_hw_layout.py to pin the gfx950 register layouts the kernels assume.
 * which: 0 = mfma_f32_32x32x16_bf16, 1 = mfma_f32_16x16x32_bf16, 2 = ds_read_b64_tr_b16, 3 = mfma_f32_32x32x16_f16, 4 = v_permlane32_swap_b32,
 * 5 = mfma_f32_32x32x2_f32 (operands a[64], b[64], c[64][16] as fp32). */
int vq_debug_probe(int which, const void* in, void* out, void* stream);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* VQHIP_H_ */
