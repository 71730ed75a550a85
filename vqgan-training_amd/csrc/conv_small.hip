// Convolutions with 8 (padded) input channels: the 3-channel image layers (conv_in ae.py:197-199,
// VGG16 conv1_1 utils.py:95-111/148-154) and the data gradient of the 3-channel outputs (conv_out
// ae.py:307-309).  K = 9 taps x 8 channels = 72: an implicit-GEMM tile would spend its time in block
// prologues, so these layers get their own HBM-bound kernels.
//
// Forward / dgrad-as-conv  (conv3x3_c8_kernel):  "direct to register" implicit GEMM.
//   For v_mfma_f32_32x32x16_bf16 the b-operand of lane l is 8 consecutive k of pixel (l & 31), k-octet
//   (l >> 5) — with Cin = 8 that is exactly ONE tap's 16-byte channel vector, so the pixel operand is
//   loaded straight from HBM/L2 into the MFMA source registers (no LDS, no staging); taps 2*kk + (l>>5),
//   kk = 0..4 (tap 9 = zero pad).  The weight fragments [Cout/32][5] are loaded once per wave and live in
//   VGPRs for the whole (grid-stride) kernel.  Epilogue = bias / ReLU / ReLU-mask / NHWC store.
//
// Weight gradient  (wgrad_c8_kernel):  dW[co][tap][ci] = sum_p dY[p][co] * X[p+tap][ci].
//   One block = one run of 64 consecutive output pixels (one image-row segment) per step, ALL 9 taps and all
//   Cout: dY is read from HBM exactly once.  The three input rows (66 pixels x 16 B each) are staged in LDS;
//   the (tap,ci) operand is fetched with ds_read_b64_tr_b16 where each 16-lane group covers TWO taps
//   (8 channels each) — four taps per 32-row fragment, three fragments for the 9 taps.
#include "vq_common.h"

struct SmallConvParams {
  VqConvDesc d;
  const vq_bf16* x;
  const vq_bf16* w;      // packed [Cout][Kp], K = tap*8 + c
  const float* bias;
  const vq_bf16* relu_mask;
  vq_bf16* y;
  int M, HoWo, Kp;
  float alpha;             // accumulator scale (VqConvDesc.alpha x *alpha_dev: the 1/s_w of a VQ_F16 packed weight)
  const float* alpha_dev;
};

template <int DT, int FC>   // 16-bit storage / operand type; Cout_pad / 32
__global__ __launch_bounds__(256) void conv3x3_c8_kernel(const SmallConvParams p) {
  const int lane = threadIdx.x & 63;
  const int fr = lane & 31, fh = lane >> 5;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;

  s16x8 wf[FC][5];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      int row = a * 32 + fr;
      if (row >= p.d.Cout) row = p.d.Cout - 1;
      wf[a][kk] = *(const s16x8*)(p.w + (int64_t)row * p.Kp + kk * 16 + fh * 8);
    }
  typedef Store<DT> St;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;
  constexpr int CP = FC * 32, SPR = CP / 8;            // slab row length (channels), 16-byte slots per row
  __shared__ __attribute__((aligned(16))) vq_bf16 slabs[4 * 32 * CP];
  vq_bf16* slab = slabs + (threadIdx.x >> 6) * 32 * CP;
  float bv[FC][4][4];                                  // this lane's bias values (4 couts per accumulator quad), loaded once
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = a * 32 + q * 8 + fh * 4 + e;
        bv[a][q][e] = (p.bias && co < p.d.Cout_w) ? p.bias[co] : 0.f;
      }
  const int ngroups = (p.M + 31) >> 5;
  for (int g = wave_global; g < ngroups; g += nwaves) {
    const int m = g * 32 + fr;
    const bool live = m < p.M;
    const int mm = live ? m : p.M - 1;
    const int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    const int oy = rem / p.d.Wo, ox = rem - oy * p.d.Wo;
    s16x8 bf[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      const int tap = 2 * kk + fh;
      const int r = tap / 3, s = tap - r * 3;
      const int iy = oy + r - 1, ix = ox + s - 1;
      const bool ok = live && tap < 9 && (unsigned)iy < (unsigned)p.d.H && (unsigned)ix < (unsigned)p.d.W;
      s16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = 0;
      bf[kk] = ok ? *(const s16x8*)(p.x + ((int64_t)(n * p.d.H + iy) * p.d.W + ix) * 8) : z;
    }
    f32x16 acc[FC];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk)
#pragma unroll
      for (int a = 0; a < FC; ++a) acc[a] = mfma16<DT>(wf[a][kk], bf[kk], acc[a]);
    // Epilogue: the wave's 32 pixels x Cout tile goes through a wave-private LDS slab as bf16 [pixel][cout] (16-byte slots
    // rotated by the pixel row) and leaves as 16 B per lane, consecutive lanes on consecutive pieces of a pixel row —
    // the accumulator layout (4 couts of one pixel per lane) would touch 32 rows with 16 B each per store instruction.
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = a * 32 + q * 8 + fh * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[a][q * 4 + e] * alpha + bv[a][q][e];
          if (p.d.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        St::store4(slab, fr * CP + ((((co >> 3) + fr) % SPR) << 3) + (co & 4), v);
      }
    vq_wave_sync();
    const int64_t m0 = (int64_t)g * 32;
#pragma unroll
    for (int it = 0; it < (32 * SPR) / 64; ++it) {
      const int i = it * 64 + lane, p_l = i / SPR, sl = i % SPR;
      if (m0 + p_l >= p.M || sl * 8 >= p.d.Cout) continue;
      float v[8];
      St::load8(slab, p_l * CP + (((sl + p_l) % SPR) << 3), v);
      const int64_t off = (m0 + p_l) * p.d.Cout + sl * 8;
      if (p.relu_mask) {
        float mv[8];
        St::load8(p.relu_mask, off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      St::store8(p.y, off, v);
    }
    vq_wave_sync();                                     // the slab is rewritten by the next group
  }
}

// Returns VQ_OK, or 1 when the shape is not handled here (caller falls through to the generic kernel).
int vq_launch_conv_c8(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual,
                      const void* relu_mask, void* y, float alpha, const float* alpha_dev, hipStream_t stream) {
  if (!(d->subpix == 0 && (d->dtype == VQ_BF16 || d->dtype == VQ_F16) && d->split == 1 && d->Cin == 8 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 &&
        d->up == 1 && d->pad_t == 1 && d->pad_l == 1 && residual == nullptr && d->Cout <= 128 && d->Ho == d->H && d->Wo == d->W))
    return 1;
  SmallConvParams p;
  p.d = *d; p.x = (const vq_bf16*)x; p.w = (const vq_bf16*)w_packed; p.bias = bias; p.relu_mask = (const vq_bf16*)relu_mask;
  p.y = (vq_bf16*)y;
  p.M = d->N * d->Ho * d->Wo; p.HoWo = d->Ho * d->Wo;
  p.Kp = vq_round_up(9 * 8, 64);
  p.alpha = alpha; p.alpha_dev = alpha_dev;
  const int ngroups = (p.M + 31) / 32;
  int blocks = (ngroups + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  const int fc = (d->Cout + 31) / 32;
#define VQ_C8(DTv) do { \
    if (fc == 1) hipLaunchKernelGGL((conv3x3_c8_kernel<DTv, 1>), dim3(blocks), dim3(256), 0, stream, p); \
    else if (fc == 2) hipLaunchKernelGGL((conv3x3_c8_kernel<DTv, 2>), dim3(blocks), dim3(256), 0, stream, p); \
    else if (fc == 3) hipLaunchKernelGGL((conv3x3_c8_kernel<DTv, 3>), dim3(blocks), dim3(256), 0, stream, p); \
    else hipLaunchKernelGGL((conv3x3_c8_kernel<DTv, 4>), dim3(blocks), dim3(256), 0, stream, p); } while (0)
  if (d->dtype == VQ_F16) VQ_C8(VQ_F16); else VQ_C8(VQ_BF16);
#undef VQ_C8
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(c8)");
  return VQ_OK;
}

// ---- the same layers in VQ_F16X2 storage (include/vqhip.h): [pixel][hi 8 | lo 8] input, hi*lo + lo*hi + hi*hi per tap pair ------------
// One wave = 32 pixels x ONE 32-channel fragment of the output (work item = (pixel group, fragment): the weight fragments of all
// planes for 128 output channels would not fit the register file beside the accumulators), its hi / lo weight fragments resident in
// VGPRs; the tile leaves through a wave-private fp32 slab as 8 channels = 32 contiguous bytes (hi piece, lo piece) per lane.
// Packed weights: row-major [Cout][Kp], virtual k = tap * 16 + plane * 8 + c (vq_pack_weight_* with op_dtype VQ_F16X2).
template <int FC>
__global__ __launch_bounds__(256) void conv3x3_c8_x2_kernel(const SmallConvParams p) {
  typedef Store<VQ_F16X2> St;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int wave_global = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  __shared__ __attribute__((aligned(16))) float slabs[4 * 32 * 32];
  float* slab = slabs + wave * 32 * 32;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;
  const int ngroups = (p.M + 31) >> 5, nitems = ngroups * FC;
  int a_cur = -1;
  s16x8 wh[5], wl[5];
  float bv[4][4];
  for (int t = wave_global; t < nitems; t += nwaves) {
    const int g = t / FC, a = t - g * FC;
    if (a != a_cur) {                                   // (wave-uniform; once per wave whenever the wave count is a multiple of FC)
      a_cur = a;
      int row = a * 32 + fr;
      if (row >= p.d.Cout) row = p.d.Cout - 1;
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const vq_bf16* src = p.w + (int64_t)row * p.Kp + (2 * kk + fh) * 16;
        wh[kk] = *(const s16x8*)src;
        wl[kk] = *(const s16x8*)(src + 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = a * 32 + q * 8 + fh * 4 + e;
          bv[q][e] = (p.bias && co < p.d.Cout_w) ? p.bias[co] : 0.f;
        }
    }
    const int m = g * 32 + fr;
    const bool live = m < p.M;
    const int mm = live ? m : p.M - 1;
    const int n = mm / p.HoWo, rem = mm - n * p.HoWo;
    const int oy = rem / p.d.Wo, ox = rem - oy * p.d.Wo;
    s16x8 bh[5], bl[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      const int tap = 2 * kk + fh;
      const int r = tap / 3, sx = tap - r * 3;
      const int iy = oy + r - 1, ix = ox + sx - 1;
      const bool ok = live && tap < 9 && (unsigned)iy < (unsigned)p.d.H && (unsigned)ix < (unsigned)p.d.W;
      s16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = 0;
      const vq_bf16* src = p.x + ((int64_t)(n * p.d.H + iy) * p.d.W + ix) * 16;
      bh[kk] = ok ? *(const s16x8*)src : z;
      bl[kk] = ok ? *(const s16x8*)(src + 8) : z;
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      acc = mfma_32x32x16_f16(wh[kk], bl[kk], acc);
      acc = mfma_32x32x16_f16(wl[kk], bh[kk], acc);
      acc = mfma_32x32x16_f16(wh[kk], bh[kk], acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vq_f4 v;
      float t4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t4[e] = acc[q * 4 + e] * alpha + bv[q][e];
        if (p.d.relu) t4[e] = t4[e] > 0.f ? t4[e] : 0.f;
      }
      v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
      *(vq_f4*)(slab + fr * 32 + (((q * 2 + fh) ^ (fr & 7)) << 2)) = v;
    }
    vq_wave_sync();
    const int64_t m0 = (int64_t)g * 32;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = it * 64 + lane, p_l = i >> 2, sl = i & 3;
      const int co = a * 32 + sl * 8;
      const vq_f4 lo4 = *(const vq_f4*)(slab + p_l * 32 + (((2 * sl) ^ (p_l & 7)) << 2));
      const vq_f4 hi4 = *(const vq_f4*)(slab + p_l * 32 + (((2 * sl + 1) ^ (p_l & 7)) << 2));
      if (m0 + p_l >= p.M || co >= p.d.Cout) continue;
      float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
      const int64_t off = (m0 + p_l) * p.d.Cout + co;
      if (p.relu_mask) {
        float mv[8];
        St::load8(p.relu_mask, off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      St::store8(p.y, off, v);
    }
    vq_wave_sync();                                     // the slab is rewritten by the next item
  }
}
// d: the VIRTUALISED descriptor (Cin counts virtual channels: 16 = 8 real) of vq_conv2d_fwd; VQ_OK, or 1 = not this kernel's shape
bool vq_conv_c8_x2_shape(const VqConvDesc* d) {
  return d->dtype == VQ_F16X2 && d->subpix == 0 && d->split == 1 && d->Cin == 16 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 &&
         d->up == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Cout <= 128 && d->Ho == d->H && d->Wo == d->W;
}
int vq_launch_conv_c8_x2(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual,
                         const void* relu_mask, void* y, float alpha, const float* alpha_dev, hipStream_t stream) {
  if (!vq_conv_c8_x2_shape(d) || residual != nullptr) return 1;
  SmallConvParams p;
  p.d = *d; p.x = (const vq_bf16*)x; p.w = (const vq_bf16*)w_packed; p.bias = bias; p.relu_mask = (const vq_bf16*)relu_mask;
  p.y = (vq_bf16*)y;
  p.M = d->N * d->Ho * d->Wo; p.HoWo = d->Ho * d->Wo;
  p.Kp = vq_round_up(9 * 16, 64);
  p.alpha = alpha; p.alpha_dev = alpha_dev;
  const int fc = (d->Cout + 31) / 32;
  const int nitems = ((p.M + 31) / 32) * fc;
  int blocks = (nitems + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  blocks = (blocks + 2) / 3 * 3;                        // 4 * blocks waves: a multiple of every fragment count 1..4
  if (fc == 1) hipLaunchKernelGGL((conv3x3_c8_x2_kernel<1>), dim3(blocks), dim3(256), 0, stream, p);
  else if (fc == 2) hipLaunchKernelGGL((conv3x3_c8_x2_kernel<2>), dim3(blocks), dim3(256), 0, stream, p);
  else if (fc == 3) hipLaunchKernelGGL((conv3x3_c8_x2_kernel<3>), dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv3x3_c8_x2_kernel<4>), dim3(blocks), dim3(256), 0, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(c8, VQ_F16X2)");
  return VQ_OK;
}

// ------------------------------------------------------------------------------------------ wgrad, 8 channels on one side
// One pass over the wide tensor for all 9 taps:  part[tap*8 + c8][c] = sum_p WIDE[p][c] * NARROW[p + tap][c8].
//   Cin == 8  (image layers, VGG conv1_1 / encoder.conv_in):  WIDE = dY, NARROW = X  -> part[tap*8+ci][co] = dW[co][tap][ci]
//   Cout == 8 (decoder.conv_out, 128 -> 3):                   WIDE = X,  NARROW = dY -> with q = p + tap:
//       dW[co][tap][ci] = sum_q X[q][ci] * dY[q - tap][co] = part[(8-tap)*8 + co][ci]     ("swapped", taps mirrored)
struct SmallWgradParams {
  const vq_bf16* narrow;   // [M][8]
  const vq_bf16* wide;     // [M][C]
  float* part;             // [nblk][96 = 12 tap slots x 8][C]
  int H, W, C;
  int M, runs_per_block, nruns;
  int nstride, noff;       // elements between the pixels of `narrow` / offset of the plane to read: 8 / 0, or — VQ_F16X2: [pixel][hi 8 | lo 8] —
};                         // 16 / 0 (hi plane), 16 / 8 (lo plane)

// LDS (two buffers, the next 64-pixel run is staged while the current one is multiplied): narrow rows [3][72 px slots][8 ch]
// (pixel slot j <-> ix = ox0 - 1 + j, slots 66..71 unused), a 64-element zero page for the 3 dead tap slots, and the wide
// tile [64][BT] in the layout of conv_wgrad_glds_kernel (segment-XOR swizzle, LDS-DMA).
template <int DT, int BT>   // 16-bit operand type; channel tile of the wide tensor: 64 or 128
__global__ __launch_bounds__(256) void wgrad_c8_kernel(const SmallWgradParams p) {
  constexpr int RB = BT * 2, SPR = RB / 16, RPP = 1024 / RB, NPC = (64 / RPP) / 4;
  constexpr int XROW = 72 * 8;                      // elements per staged narrow row
  __shared__ __attribute__((aligned(16))) vq_bf16 lds[2 * 3 * XROW + 64 + 2 * 64 * BT];
  vq_bf16* zs = lds + 2 * 3 * XROW;                 // 60 zero elements, then {1, 0, 0, 0}: the "ones" channel of tap slot 9
  auto xs_of = [&](int b) -> vq_bf16* { return lds + b * 3 * XROW; };
  auto ys_of = [&](int b) -> vq_bf16* { return lds + 2 * 3 * XROW + 64 + b * 64 * BT; };
  auto seg_key = [](int row) -> int { return RB >= 256 ? (row & 3) : ((row >> 1) & 1); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p.W, H = p.H;
  // every wave takes all 3 tap fragments (12 tap slots x 8 channels = 96 rows) and one 32-channel fragment of the
  // wide tile: BT = 128 -> 4 fragments / 4 waves; BT = 64 -> waves 2, 3 only help staging
  constexpr int NCF = BT / 32;
  const int my_cf = wave;
  f32x16 acc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

  if (tid < 32) ((unsigned*)zs)[tid] = (tid == 30) ? (unsigned)One16<DT>::BITS : 0u;   // elements 60..63 = {1.0, 0, 0, 0}
  const int gg = lane >> 4, tl = lane & 15;
  // a-operand addressing: 32-row fragment f covers tap slots 4f..4f+3; 16-lane group half (gg&1) covers 2 of them;
  // lane chunk cc = tl & 3: chunks 0,1 -> first tap of the pair (channels 0-3, 4-7), chunks 2,3 -> second tap.
  int a_off[3];     // element offset (relative to xs / zs) of this lane's 8-byte piece for k-row 0 of the fragment
  bool a_zero[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    const int tap = 4 * f + 2 * (gg & 1) + ((tl & 3) >> 1);
    a_zero[f] = tap >= 9;
    const int r = tap / 3, s = tap - r * 3;
    // pixel k of the chunk reads slot k + s (slot j <-> ix = ox0 - 1 + j); k-row of this lane = 8*(gg>>1) + (tl>>2).
    // Tap slot 9, channels 0-3 read {1,0,0,0} for every pixel: row 72 of the partial tile = sum_p WIDE[p][c] (bias gradient)
    a_off[f] = a_zero[f] ? ((tap == 9 && (tl & 1) == 0) ? 60 : 0) : (r * XROW + (8 * (gg >> 1) + (tl >> 2) + s) * 8 + (tl & 1) * 4);
  }
  const int b_row = 8 * (gg >> 1) + (tl >> 2);
  const int b_col = (gg & 1) * 16 + (tl & 3) * 4;

  const int lrow = lane / SPR, lp = lane % SPR;
  const int ct0 = (int)blockIdx.y * BT;            // channel tile of the wide tensor (C > BT: 256 / 512 channels, one tile per blockIdx.y)
  const int run0 = blockIdx.x * p.runs_per_block;
  int nhere = p.nruns - run0;
  if (nhere > p.runs_per_block) nhere = p.runs_per_block;

  auto stage_wide = [&](int run, int b) {           // LDS-DMA, swizzled source: 64 rows x BT channels
    const int m0 = run * 64;
    vq_bf16* ys = ys_of(b);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int row = (wave * NPC + i) * RPP + lrow;
      const int seg = (lp >> 2) ^ seg_key(row);
      const int lsl = ((seg << 2) | (lp & 3)) << 3;
      glds16(p.wide + (int64_t)(m0 + row) * p.C + ct0 + lsl, ys + (wave * NPC + i) * RPP * BT);
    }
  };
  auto load_narrow = [&](int run) -> vq_u4 {        // three rows of 66 pixels x 16 B (threads 0..197)
    vq_u4 v; v.x = v.y = v.z = v.w = 0u;
    if (tid < 3 * 66) {
      const int m0 = run * 64;
      const int n = m0 / (H * W), rem = m0 - n * H * W;
      const int oy = rem / W, ox0 = rem - oy * W;
      const int r = tid / 66, j = tid - r * 66;
      const int iy = oy + r - 1, ix = ox0 - 1 + j;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = *(const vq_u4*)(p.narrow + ((int64_t)(n * H + iy) * W + ix) * p.nstride + p.noff);
    }
    return v;
  };
  auto store_narrow = [&](int b, vq_u4 v) {
    if (tid < 3 * 66) {
      const int r = tid / 66, j = tid - r * 66;
      *(vq_u4*)(xs_of(b) + r * XROW + j * 8) = v;
    }
  };

  if (nhere > 0) {
    stage_wide(run0, 0);
    store_narrow(0, load_narrow(run0));
  }
  for (int rr = 0; rr < nhere; ++rr) {
    const int b = rr & 1;
    __syncthreads();                                // buffer b is staged; every wave is done with buffer b^1
    const bool more = rr + 1 < nhere;               // block-uniform
    vq_u4 nv; nv.x = nv.y = nv.z = nv.w = 0u;
    if (more) {
      stage_wide(run0 + rr + 1, b ^ 1);
      nv = load_narrow(run0 + rr + 1);
    }
    if (my_cf < NCF) {
      const vq_bf16* xs = xs_of(b);
      const char* yb = (const char*)ys_of(b);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // b-operand: WIDE^T fragment of this wave's 32 channels (two transposed reads)
        const int c = my_cf * 32 + b_col;
        const int seg = (c * 2) >> 6, within = (c * 2) & 63;
        const int r0 = kk * 16 + b_row, r1 = r0 + 4;
        s16x4 lo4 = lds_read_tr16_b64((const short*)(yb + r0 * RB + ((seg ^ seg_key(r0)) << 6) + within));
        s16x4 hi4 = lds_read_tr16_b64((const short*)(yb + r1 * RB + ((seg ^ seg_key(r1)) << 6) + within));
        s16x8 bfr;
        bfr[0] = lo4[0]; bfr[1] = lo4[1]; bfr[2] = lo4[2]; bfr[3] = lo4[3];
        bfr[4] = hi4[0]; bfr[5] = hi4[1]; bfr[6] = hi4[2]; bfr[7] = hi4[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          const vq_bf16* base = a_zero[f] ? zs + a_off[f] : xs + a_off[f] + kk * 16 * 8;
          const int step = a_zero[f] ? 0 : 4 * 8;   // +4 pixel slots for the second transposed read
          s16x4 al = lds_read_tr16_b64((const short*)base);
          s16x4 ah = lds_read_tr16_b64((const short*)(base + step));
          s16x8 af;
          af[0] = al[0]; af[1] = al[1]; af[2] = al[2]; af[3] = al[3];
          af[4] = ah[0]; af[5] = ah[1]; af[6] = ah[2]; af[7] = ah[3];
          acc[f] = mfma16<DT>(af, bfr, acc[f]);
        }
      }
    }
    if (more) store_narrow(b ^ 1, nv);
  }
  // partial tile: rows i = f*32 + (e&3)+8*(e>>2)+4*fh  (= tap slot * 8 + narrow channel), cols = wide channels of this wave
  if (my_cf < NCF) {
    float* out = p.part + (int64_t)blockIdx.x * 96 * p.C;
    const int fr = lane & 31, fh = lane >> 5;
    const int c = ct0 + my_cf * 32 + fr;
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = f * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        out[(int64_t)i * p.C + c] = acc[f][e];
      }
  }
}

// dw[co][ci][tap] (+)= sum_blk part[blk][row][col]   — 8 lanes per output element, fixed-order tree.
//   swapped == 0: row = tap*8 + ci, col = co;   swapped == 1: row = (8-tap)*8 + co, col = ci
// Elements total .. total + Cout_w - 1 (dbias != nullptr, not swapped): dbias[co] (+)= sum_blk part[blk][72][co].
__global__ void wgrad_c8_reduce_kernel(const float* __restrict__ part, int nblk, int C, int Cout_w, int Cin_w,
                                       int swapped, int accumulate, float alpha, float* __restrict__ dw, float* __restrict__ dbias) {
  const int total = Cout_w * Cin_w * 9, total_b = total + (dbias ? Cout_w : 0);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, sub = t & 7;
  const bool live = i < total_b, is_bias = i >= total;
  const int ii = (live && !is_bias) ? i : 0;
  const int tap = ii % 9, ci = (ii / 9) % Cin_w, co = is_bias ? (live ? i - total : 0) : ii / (9 * Cin_w);
  const int row = is_bias ? 72 : (swapped ? (8 - tap) * 8 + co : tap * 8 + ci), col = (swapped && !is_bias) ? ci : co;
  float s = 0.f;
  for (int b = sub; b < nblk; b += 8) s += part[((int64_t)b * 96 + row) * C + col];
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  s *= alpha;
  if (live && sub == 0) {
    float* dst = is_bias ? dbias + co : dw + ((int64_t)co * Cin_w + ci) * 9 + tap;
    *dst = accumulate ? (*dst + s) : s;
  }
}

static int c8_wgrad_blocks(const VqConvDesc* d) {
  const int nruns = d->N * d->Ho * d->Wo / 64;
  return nruns < 512 ? nruns : 512;
}
static bool c8_common(const VqConvDesc* d) {
  return (d->dtype == VQ_BF16 || d->dtype == VQ_F16) && d->split == 1 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && d->up == 1 &&
         d->pad_t == 1 && d->pad_l == 1 && d->Ho == d->H && d->Wo == d->W && d->Wo % 64 == 0;
}
// wide side: 64 channels, or any multiple of 128 up to 1024 (one 128-channel tile per blockIdx.y — the wavelet / HR-decoder models of
// the reference's launch scripts end in 256 -> 3 at 512 x 512: on the generic path that launch ran at 7 TFLOP/s, profiles/r4l_*)
static bool c8_wide_ok(int c) { return c == 64 || (c % 128 == 0 && c >= 128 && c <= 1024); }
static bool c8_swapped(const VqConvDesc* d) { return d->Cout == 8 && c8_wide_ok(d->Cin); }
bool vq_wgrad_c8_eligible(const VqConvDesc* d) {
  return c8_common(d) && ((d->Cin == 8 && c8_wide_ok(d->Cout)) || c8_swapped(d));
}
size_t vq_wgrad_c8_workspace(const VqConvDesc* d) {
  return (size_t)c8_wgrad_blocks(d) * 96 * (c8_swapped(d) ? d->Cin : d->Cout) * sizeof(float);
}

// *dbias_done = 1 when the bias gradient came out of the same pass (wide tensor == dY)
int vq_launch_wgrad_c8(const VqConvDesc* d, const void* x, const void* dy, float* dw, float* dbias, int* dbias_done,
                       int accumulate, float alpha, void* workspace, hipStream_t stream) {
  const bool sw = c8_swapped(d);
  SmallWgradParams p;
  p.narrow = (const vq_bf16*)(sw ? dy : x); p.wide = (const vq_bf16*)(sw ? x : dy); p.part = (float*)workspace;
  p.H = d->H; p.W = d->W; p.C = sw ? d->Cin : d->Cout;
  p.M = d->N * d->Ho * d->Wo;
  p.nstride = 8; p.noff = 0;
  p.nruns = p.M / 64;
  const int nblk = c8_wgrad_blocks(d);
  p.runs_per_block = (p.nruns + nblk - 1) / nblk;
  const int used = (p.nruns + p.runs_per_block - 1) / p.runs_per_block;
  const dim3 grid(used, p.C >= 128 ? p.C / 128 : 1);
  if (d->dtype == VQ_F16) {
    if (p.C >= 128) hipLaunchKernelGGL((wgrad_c8_kernel<VQ_F16, 128>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((wgrad_c8_kernel<VQ_F16, 64>), grid, dim3(256), 0, stream, p);
  } else {
    if (p.C >= 128) hipLaunchKernelGGL((wgrad_c8_kernel<VQ_BF16, 128>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((wgrad_c8_kernel<VQ_BF16, 64>), grid, dim3(256), 0, stream, p);
  }
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad(c8)");
  float* db = sw ? nullptr : dbias;
  *dbias_done = db != nullptr;
  const int total = d->Cout_w * d->Cin_w * 9 + (db ? d->Cout_w : 0);
  hipLaunchKernelGGL(wgrad_c8_reduce_kernel, dim3((total * 8 + 255) / 256), dim3(256), 0, stream, (const float*)workspace, used,
                     p.C, d->Cout_w, d->Cin_w, sw ? 1 : 0, accumulate, alpha, dw, db);
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad(c8 reduce)");
  return VQ_OK;
}


// ---- VQ_F16X2 (include/vqhip.h): the same one-pass kernel, twice ---------------------------------------------------------------
// The wide tensor is read as the binary16 tensor of 2C virtual channels it is byte for byte; the narrow one ([pixel][hi 8 | lo 8]) once
// through its hi plane and once through its lo plane (`nstride` 16, `noff` 0 / 8).  Each launch leaves part[blk][96][2C]; the reduction
// adds, per output element, the four products (narrow hi / lo) x (wide hi / lo column of the channel) — hi x hi apart from the three
// small ones — over the blocks in the fixed lane order of wgrad_c8_reduce_kernel.
__global__ void wgrad_c8_reduce_x2_kernel(const float* __restrict__ part_h, const float* __restrict__ part_l, int nblk, int CV, int Cout_w,
                                          int Cin_w, int swapped, int accumulate, float alpha, float* __restrict__ dw, float* __restrict__ dbias) {
  const int total = Cout_w * Cin_w * 9, total_b = total + (dbias ? Cout_w : 0);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, sub = t & 7;
  const bool live = i < total_b, is_bias = i >= total;
  const int ii = (live && !is_bias) ? i : 0;
  const int tap = ii % 9, ci = (ii / 9) % Cin_w, co = is_bias ? (live ? i - total : 0) : ii / (9 * Cin_w);
  const int row = is_bias ? 72 : (swapped ? (8 - tap) * 8 + co : tap * 8 + ci), wc = (swapped && !is_bias) ? ci : co;
  const int col = ((wc >> 3) << 4) + (wc & 7);      // virtual column of the wide channel's hi piece; its lo piece 8 further
  float big = 0.f, small = 0.f;
  for (int b = sub; b < nblk; b += 8) {
    const float* ph = part_h + ((int64_t)b * 96 + row) * CV + col;
    big += ph[0];
    if (is_bias) small += ph[8];                    // (the "ones" slot of the hi launch: sum of the wide tensor's two pieces)
    else { const float* pl = part_l + ((int64_t)b * 96 + row) * CV + col; small += (ph[8] + pl[0]) + pl[8]; }
  }
  big += __shfl_xor(big, 1); big += __shfl_xor(big, 2); big += __shfl_xor(big, 4);
  small += __shfl_xor(small, 1); small += __shfl_xor(small, 2); small += __shfl_xor(small, 4);
  const float r = (big + small) * alpha;
  if (live && sub == 0) {
    float* dst = is_bias ? dbias + co : dw + ((int64_t)co * Cin_w + ci) * 9 + tap;
    *dst = accumulate ? (*dst + r) : r;
  }
}
static bool c8_x2_swapped(const VqConvDesc* d) { return d->Cout == 8 && c8_wide_ok(2 * d->Cin); }
bool vq_wgrad_c8_x2_eligible(const VqConvDesc* d) {       // d: the caller's (real-channel) descriptor
  return d->dtype == VQ_F16X2 && d->split == 1 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && d->up == 1 && d->pad_t == 1 &&
         d->pad_l == 1 && d->Ho == d->H && d->Wo == d->W && d->Wo % 64 == 0 && d->alpha_dev == nullptr &&
         ((d->Cin == 8 && c8_wide_ok(2 * d->Cout)) || c8_x2_swapped(d));
}
size_t vq_wgrad_c8_x2_workspace(const VqConvDesc* d) {
  return (size_t)2 * c8_wgrad_blocks(d) * 96 * 2 * (c8_x2_swapped(d) ? d->Cin : d->Cout) * sizeof(float);
}
int vq_launch_wgrad_c8_x2(const VqConvDesc* d, const void* x, const void* dy, float* dw, float* dbias, int* dbias_done, int accumulate,
                          float alpha, void* workspace, hipStream_t stream) {
  const bool sw = c8_x2_swapped(d);
  SmallWgradParams p;
  p.narrow = (const vq_bf16*)(sw ? dy : x); p.wide = (const vq_bf16*)(sw ? x : dy);
  p.H = d->H; p.W = d->W; p.C = 2 * (sw ? d->Cin : d->Cout);
  p.M = d->N * d->Ho * d->Wo;
  p.nstride = 16;
  p.nruns = p.M / 64;
  const int nblk = c8_wgrad_blocks(d);
  p.runs_per_block = (p.nruns + nblk - 1) / nblk;
  const int used = (p.nruns + p.runs_per_block - 1) / p.runs_per_block;
  const dim3 grid(used, p.C >= 128 ? p.C / 128 : 1);
  float* part_h = (float*)workspace;
  float* part_l = part_h + (size_t)nblk * 96 * p.C;
  for (int pl = 0; pl < 2; ++pl) {
    p.noff = pl * 8; p.part = pl ? part_l : part_h;
    if (p.C >= 128) hipLaunchKernelGGL((wgrad_c8_kernel<VQ_F16, 128>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((wgrad_c8_kernel<VQ_F16, 64>), grid, dim3(256), 0, stream, p);
  }
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad(c8, VQ_F16X2)");
  float* db = sw ? nullptr : dbias;
  *dbias_done = db != nullptr;
  const int total = d->Cout_w * d->Cin_w * 9 + (db ? d->Cout_w : 0);
  hipLaunchKernelGGL(wgrad_c8_reduce_x2_kernel, dim3((total * 8 + 255) / 256), dim3(256), 0, stream, (const float*)part_h, (const float*)part_l,
                     used, p.C, d->Cout_w, d->Cin_w, sw ? 1 : 0, accumulate, alpha, dw, db);
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad(c8 reduce, VQ_F16X2)");
  return VQ_OK;
}
