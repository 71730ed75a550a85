// Input preparation either side of the hot path, all HBM-bound streaming kernels on NCHW
// fp32 images / latents as the reference's trainer holds them:
//   * wavelet front-end (utils.py:206-247, ae.py:189-194,240): zero-pad 2, four fixed separable 6x6 filters, stride 2,
//     per input channel -> [B, 4C, H/2, W/2]; written straight into the NHWC (padded C) activation layout the
//     encoder's conv_in consumes (or NCHW fp32 for the stand-alone utils.wavelet_transform_multi_channel API);
//   * flips with latent sign changes (vae_trainer.py:534-536, 567-575, 664-671): an involution, so the same kernel
//     is its own backward;
//   * F.interpolate(mode="area") for integer ratios (vae_trainer.py:531-533) = k x k mean.
#include "vq_common.h"

// utils.py:206-209 (the literals are the reference's own 4-decimal constants)
VQ_CONSTANT float c_dec_lo[6] = {-0.1768f, 0.3536f, 1.0607f, 0.3536f, -0.1768f, 0.0000f};
VQ_CONSTANT float c_dec_hi[6] = {0.0000f, -0.0000f, 0.3536f, -0.7071f, 0.3536f, -0.0000f};

// One thread per (n, c, oy, ox): the 6x6 input window is read once and reduced against the four filters
//   f0 = lo(j) lo(i), f1 = lo(j) hi(i), f2 = hi(j) lo(i), f3 = hi(j) hi(i)      (i = row, j = column; utils.py:211-219)
// in the cross-correlation order of F.conv2d (row-major over the window).  Output channel = c*4 + f (utils.py:243-246).
template <int DT, int NHWC>
__global__ void wavelet_kernel(const float* __restrict__ x, void* __restrict__ y, int N, int C, int H, int W, int Cpad) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)N * C * Ho * Wo;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(t % Wo);
    int64_t r = t / Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const float* src = x + ((int64_t)n * C + c) * H * W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int iy = 2 * oy + i - 2;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ix = 2 * ox + j - 2;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = src[(int64_t)iy * W + ix];
        acc[0] = fmaf(c_dec_lo[j] * c_dec_lo[i], v, acc[0]);
        acc[1] = fmaf(c_dec_lo[j] * c_dec_hi[i], v, acc[1]);
        acc[2] = fmaf(c_dec_hi[j] * c_dec_lo[i], v, acc[2]);
        acc[3] = fmaf(c_dec_hi[j] * c_dec_hi[i], v, acc[3]);
      }
    }
    if (NHWC) {
      Store<DT>::store4(y, ((int64_t)(n * Ho + oy) * Wo + ox) * Cpad + c * 4, acc);
    } else {
      float* dst = (float*)y;
#pragma unroll
      for (int f = 0; f < 4; ++f) dst[(((int64_t)n * 4 * C + c * 4 + f) * Ho + oy) * Wo + ox] = acc[f];
    }
  }
}

// zero the padding channels [4C, Cpad) of the NHWC output
template <int DT>
__global__ void wavelet_pad_kernel(void* __restrict__ y, int64_t pixels, int C4, int Cpad) {
  const int npad = (Cpad - C4) >> 2;
  const int64_t total = pixels * npad;
  const float z[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
    Store<DT>::store4(y, (t / npad) * Cpad + C4 + (t % npad) * 4, z);
}

static int img_grid(int64_t total) {
  int64_t b = vq_ceil_div(total, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vq_wavelet_fwd(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, int out_nhwc,
                              void* stream) {
  VQ_REQUIRE(x && y, VQ_ERR_INVALID, "vq_wavelet_fwd: null pointer");
  VQ_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, VQ_ERR_INVALID,
             "vq_wavelet_fwd: H and W must be positive and even (got %dx%d)", H, W);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)N * C * (H / 2) * (W / 2);
  if (!out_nhwc) {
    hipLaunchKernelGGL((wavelet_kernel<VQ_F32, 0>), dim3(img_grid(total)), dim3(256), 0, s, x, y, N, C, H, W, 4 * C);
    VQ_CHECK_LAUNCH("vq_wavelet_fwd");
    return VQ_OK;
  }
  VQ_REQUIRE(Cpad % 8 == 0 && Cpad >= 4 * C, VQ_ERR_INVALID, "vq_wavelet_fwd: Cpad=%d must be a multiple of 8 >= 4*C=%d", Cpad, 4 * C);
  VQ_REQUIRE(dtype == VQ_BF16 || dtype == VQ_F32 || dtype == VQ_F16 || dtype == VQ_F16X2, VQ_ERR_INVALID, "vq_wavelet_fwd: unknown dtype %d", dtype);
  const int64_t pixels = (int64_t)N * (H / 2) * (W / 2);
  if (dtype == VQ_BF16) {
    hipLaunchKernelGGL((wavelet_kernel<VQ_BF16, 1>), dim3(img_grid(total)), dim3(256), 0, s, x, y, N, C, H, W, Cpad);
    if (Cpad > 4 * C)
      hipLaunchKernelGGL((wavelet_pad_kernel<VQ_BF16>), dim3(img_grid(pixels)), dim3(256), 0, s, y, pixels, 4 * C, Cpad);
  } else if (dtype == VQ_F16) {
    hipLaunchKernelGGL((wavelet_kernel<VQ_F16, 1>), dim3(img_grid(total)), dim3(256), 0, s, x, y, N, C, H, W, Cpad);
    if (Cpad > 4 * C)
      hipLaunchKernelGGL((wavelet_pad_kernel<VQ_F16>), dim3(img_grid(pixels)), dim3(256), 0, s, y, pixels, 4 * C, Cpad);
  } else if (dtype == VQ_F16X2) {
    hipLaunchKernelGGL((wavelet_kernel<VQ_F16X2, 1>), dim3(img_grid(total)), dim3(256), 0, s, x, y, N, C, H, W, Cpad);
    if (Cpad > 4 * C)
      hipLaunchKernelGGL((wavelet_pad_kernel<VQ_F16X2>), dim3(img_grid(pixels)), dim3(256), 0, s, y, pixels, 4 * C, Cpad);
  } else {
    hipLaunchKernelGGL((wavelet_kernel<VQ_F32, 1>), dim3(img_grid(total)), dim3(256), 0, s, x, y, N, C, H, W, Cpad);
    if (Cpad > 4 * C)
      hipLaunchKernelGGL((wavelet_pad_kernel<VQ_F32>), dim3(img_grid(pixels)), dim3(256), 0, s, y, pixels, 4 * C, Cpad);
  }
  VQ_CHECK_LAUNCH("vq_wavelet_fwd");
  return VQ_OK;
}

// y[n,c,h,w] = sgn(c) * x[n,c,h',w'],  h' = H-1-h if flip_h, w' = W-1-w if flip_w,  sgn = -1 on channels [neg_c0, neg_c1)
__global__ void flip_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W, int flip_h,
                            int flip_w, int neg_c0, int neg_c1) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(t % W);
    int64_t r = t / W;
    const int h = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int hs = flip_h ? H - 1 - h : h, ws = flip_w ? W - 1 - w : w;
    const float v = x[(r * H + hs) * W + ws];
    y[t] = (c >= neg_c0 && c < neg_c1) ? -v : v;
  }
}

extern "C" int vq_flip_nchw(const float* x, float* y, int N, int C, int H, int W, int flip_h, int flip_w, int neg_c0,
                            int neg_c1, void* stream) {
  VQ_REQUIRE(x && y && x != y, VQ_ERR_INVALID, "vq_flip_nchw: null or aliased pointers");
  VQ_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0, VQ_ERR_INVALID, "vq_flip_nchw: bad shape");
  const int64_t total = (int64_t)N * C * H * W;
  if (total == 0) return VQ_OK;
  hipLaunchKernelGGL(flip_kernel, dim3(img_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H, W, flip_h, flip_w,
                     neg_c0, neg_c1);
  VQ_CHECK_LAUNCH("vq_flip_nchw");
  return VQ_OK;
}

// y[nc, oy, ox] = mean of the k x k window (row-major sum, then one division by k*k, as adaptive average pooling does)
__global__ void area_down_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t NC, int H, int W, int k) {
  const int Ho = H / k, Wo = W / k;
  const int64_t total = NC * Ho * Wo;
  const float cnt = (float)(k * k);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(t % Wo);
    int64_t r = t / Wo;
    const int oy = (int)(r % Ho);
    const int64_t nc = r / Ho;
    const float* src = x + (nc * H + (int64_t)oy * k) * W + (int64_t)ox * k;
    float s = 0.f;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) s += src[(int64_t)i * W + j];
    y[t] = s / cnt;
  }
}

extern "C" int vq_area_downsample_nchw(const float* x, float* y, int N, int C, int H, int W, int k, void* stream) {
  VQ_REQUIRE(x && y, VQ_ERR_INVALID, "vq_area_downsample_nchw: null pointer");
  VQ_REQUIRE(k >= 1 && H % k == 0 && W % k == 0 && N >= 0 && C > 0, VQ_ERR_UNSUPPORTED,
             "vq_area_downsample_nchw: only integer ratios are supported (H=%d W=%d k=%d)", H, W, k);
  const int64_t total = (int64_t)N * C * (H / k) * (W / k);
  if (total == 0) return VQ_OK;
  hipLaunchKernelGGL(area_down_kernel, dim3(img_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, (int64_t)N * C, H, W, k);
  VQ_CHECK_LAUNCH("vq_area_downsample_nchw");
  return VQ_OK;
}
