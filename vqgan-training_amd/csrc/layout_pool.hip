// Layout conversion at the NCHW image / latent boundary, 2x2 max / sum pooling, column sums.
// All HBM-bound streaming kernels: 8 channels (16 B bf16 / 32 B fp32) per lane where possible.
//
// Reference call sites: the reference is NCHW everywhere (ae.py, utils.py); ScalingLayer
// utils.py:60-71 is folded into the NCHW->NHWC conversion; nn.MaxPool2d(2,2) = torchvision VGG16
// features idx 4,9,16,23 inside utils.py:104-111,150-154; the 2x2 sum pool is autograd's backward of
// F.interpolate(scale_factor=2, mode="nearest") (ae.py:165); column sums are the conv bias
// gradients.
#include "vq_common.h"

// ---- NCHW fp32 -> NHWC (padded C) ---------------------------------------------------------
template <int DT>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, void* __restrict__ dst, int N, int C, int64_t HW,
                                    int Cpad, const float* __restrict__ shift, const float* __restrict__ scale, float alpha,
                                    int* __restrict__ range_events) {
  typedef Store<DT> St;
  const int groups = Cpad >> 3;
  const int64_t total = (int64_t)N * HW * groups;
  unsigned rng = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // consecutive threads walk consecutive pixels of one channel group => coalesced NCHW reads
    const int64_t pix = i % HW;
    const int64_t t = i / HW;
    const int grp = (int)(t % groups);
    const int n = (int)(t / groups);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = grp * 8 + e;
      float val = 0.f;
      if (c < C) {
        val = src[((int64_t)n * C + c) * HW + pix];
        if (shift) val = (val - shift[c]) / scale[c];
      }
      v[e] = val * alpha;
    }
    if constexpr (IsHalfRange<DT>::value) rng = vq_absmax_bits(rng, v);
    St::store8(dst, ((int64_t)n * HW + pix) * Cpad + grp * 8, v);
  }
  if constexpr (IsHalfRange<DT>::value) { if (range_events) vq_range_events(range_events, rng, rng); }
}

template <int DT>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int N, int C, int64_t HW,
                                    int Cpad, const float* __restrict__ scale_inv, float alpha) {
  typedef Store<DT> St;
  const int groups = Cpad >> 3;
  const int64_t total = (int64_t)N * HW * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % HW;
    const int64_t t = i / HW;
    const int grp = (int)(t % groups);
    const int n = (int)(t / groups);
    float v[8];
    St::load8(src, ((int64_t)n * HW + pix) * Cpad + grp * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = grp * 8 + e;
      if (c < C) dst[((int64_t)n * C + c) * HW + pix] = (scale_inv ? v[e] / scale_inv[c] : v[e]) * alpha;
    }
  }
}

static int stream_grid(int64_t total) {
  int64_t b = vq_ceil_div(total, 256);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vq_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype,
                               const float* shift, const float* scale, float alpha, int32_t* range_events, void* stream) {
  VQ_REQUIRE(src && dst, VQ_ERR_INVALID, "vq_nchw_to_nhwc: null pointer");
  VQ_REQUIRE(Cpad % 8 == 0 && Cpad >= C && C > 0, VQ_ERR_INVALID, "vq_nchw_to_nhwc: Cpad=%d must be a multiple of 8 >= C=%d", Cpad, C);
  VQ_REQUIRE((shift == nullptr) == (scale == nullptr), VQ_ERR_INVALID, "vq_nchw_to_nhwc: shift and scale go together");
  const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW * (Cpad / 8);
  if (total == 0) return VQ_OK;
  if (dtype == VQ_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<VQ_BF16>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, shift, scale, alpha, (int*)nullptr);
  else if (dtype == VQ_F16)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<VQ_F16>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, shift, scale, alpha, (int*)range_events);
  else if (dtype == VQ_F32)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<VQ_F32>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, shift, scale, alpha, (int*)nullptr);
  else if (dtype == VQ_F16X2)
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<VQ_F16X2>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, shift, scale, alpha, (int*)range_events);
  else { vq_set_error("vq_nchw_to_nhwc: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH("vq_nchw_to_nhwc");
  return VQ_OK;
}

extern "C" int vq_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cpad, int dtype,
                               const float* scale_inv, float alpha, void* stream) {
  VQ_REQUIRE(src && dst, VQ_ERR_INVALID, "vq_nhwc_to_nchw: null pointer");
  VQ_REQUIRE(Cpad % 8 == 0 && Cpad >= C && C > 0, VQ_ERR_INVALID, "vq_nhwc_to_nchw: Cpad=%d must be a multiple of 8 >= C=%d", Cpad, C);
  const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW * (Cpad / 8);
  if (total == 0) return VQ_OK;
  if (dtype == VQ_BF16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<VQ_BF16>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, scale_inv, alpha);
  else if (dtype == VQ_F16)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<VQ_F16>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, scale_inv, alpha);
  else if (dtype == VQ_F32)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<VQ_F32>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, scale_inv, alpha);
  else if (dtype == VQ_F16X2)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<VQ_F16X2>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, HW, Cpad, scale_inv, alpha);
  else { vq_set_error("vq_nhwc_to_nchw: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH("vq_nhwc_to_nchw");
  return VQ_OK;
}

// ---- 2x2 pooling -------------------------------------------------------------------------------
// MODE 0: max fwd; MODE 1: max bwd (dy -> dx, first max in row-major order wins; + `add`, the gradient of x's other consumer);
// MODE 2: sum pool
template <int DT, int MODE>
__global__ void pool2_kernel(const void* __restrict__ x, const void* __restrict__ dy, const void* __restrict__ add,
                             void* __restrict__ out, int N, int H, int W, int C, int* __restrict__ range_events) {
  typedef Store<DT> St;
  unsigned rng = 0u;
  const int Ho = H >> 1, Wo = W >> 1, groups = C >> 3;
  const int64_t total = (int64_t)N * Ho * Wo * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int grp = (int)(i % groups);
    int64_t t = i / groups;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int64_t in00 = (((int64_t)n * H + 2 * oy) * W + 2 * ox) * C + grp * 8;
    const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C + grp * 8;
    float a[8], b[8], c[8], d[8];
    St::load8(x, in00, a);
    St::load8(x, in00 + C, b);
    St::load8(x, in00 + (int64_t)W * C, c);
    St::load8(x, in00 + (int64_t)W * C + C, d);
    if (MODE == 0) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
      St::store8(out, o, r);
    } else if (MODE == 2) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (a[e] + b[e]) + (c[e] + d[e]);
      St::store8(out, o, r);
    } else {
      float g[8], ra[8], rb[8], rc[8], rd[8];
      St::load8(dy, o, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // strict '>' keeps the first maximum in scan order a,b,c,d (PyTorch CPU max_pool2d)
        int w = 0; float m = a[e];
        if (b[e] > m) { m = b[e]; w = 1; }
        if (c[e] > m) { m = c[e]; w = 2; }
        if (d[e] > m) { m = d[e]; w = 3; }
        ra[e] = w == 0 ? g[e] : 0.f; rb[e] = w == 1 ? g[e] : 0.f;
        rc[e] = w == 2 ? g[e] : 0.f; rd[e] = w == 3 ? g[e] : 0.f;
      }
      if (add) {                                     // grid-uniform
        St::load8(add, in00, a); St::load8(add, in00 + C, b);
        St::load8(add, in00 + (int64_t)W * C, c); St::load8(add, in00 + (int64_t)W * C + C, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { ra[e] += a[e]; rb[e] += b[e]; rc[e] += c[e]; rd[e] += d[e]; }
        if constexpr (IsHalfRange<DT>::value) { rng = vq_absmax_bits(rng, ra); rng = vq_absmax_bits(rng, rb); rng = vq_absmax_bits(rng, rc); rng = vq_absmax_bits(rng, rd); }
      }
      St::store8(out, in00, ra);
      St::store8(out, in00 + C, rb);
      St::store8(out, in00 + (int64_t)W * C, rc);
      St::store8(out, in00 + (int64_t)W * C + C, rd);
    }
  }
  if constexpr (IsHalfRange<DT>::value && MODE == 1) { if (range_events && add) vq_range_events(range_events, rng, rng); }
}

template <int MODE>
static int pool_launch(const void* x, const void* dy, const void* add, void* out, int N, int H, int W, int C, int dtype,
                       int32_t* range_events, void* stream, const char* name) {
  VQ_REQUIRE(x && out && (MODE != 1 || dy), VQ_ERR_INVALID, "%s: null pointer", name);
  VQ_REQUIRE(add == nullptr || add != out, VQ_ERR_INVALID, "%s: `add` must not alias the output", name);
  // nn.MaxPool2d(2, 2) floors odd extents (the last row / column is dropped: crop-invariance batches reach VGG's
  // pool4 with odd sizes, vae_trainer.py:577-621); its backward leaves zero gradient there.  The sum pool is the
  // backward of a 2x upsample and only ever sees even extents.
  VQ_REQUIRE(C % 8 == 0 && N > 0 && H >= 2 && W >= 2 && (MODE != 2 || (H % 2 == 0 && W % 2 == 0)), VQ_ERR_INVALID,
             "%s: need C%%8==0, H,W >= 2 (even for the sum pool) (H=%d W=%d C=%d)", name, H, W, C);
  if (MODE == 1 && ((H | W) & 1)) {                  // the dropped last row / column: zero gradient from the pool, `add` alone
    const size_t bytes = (size_t)N * H * W * C * ((dtype == VQ_F32 || dtype == VQ_F16X2) ? 4 : 2);
    hipError_t e = add ? hipMemcpyAsync(out, add, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream)
                       : hipMemsetAsync(out, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) { vq_set_error("%s: hipMemset/MemcpyAsync: %s", name, hipGetErrorString(e)); return VQ_ERR_HIP; }
  }
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  if (dtype == VQ_BF16)
    hipLaunchKernelGGL((pool2_kernel<VQ_BF16, MODE>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, add, out, N, H, W, C, (int*)nullptr);
  else if (dtype == VQ_F16)
    hipLaunchKernelGGL((pool2_kernel<VQ_F16, MODE>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, add, out, N, H, W, C, (int*)range_events);
  else if (dtype == VQ_F32)
    hipLaunchKernelGGL((pool2_kernel<VQ_F32, MODE>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, add, out, N, H, W, C, (int*)nullptr);
  else if (dtype == VQ_F16X2)
    hipLaunchKernelGGL((pool2_kernel<VQ_F16X2, MODE>), dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, add, out, N, H, W, C, (int*)range_events);
  else { vq_set_error("%s: unknown dtype %d", name, dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH(name);
  return VQ_OK;
}
extern "C" int vq_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  return pool_launch<0>(x, nullptr, nullptr, y, N, H, W, C, dtype, nullptr, stream, "vq_maxpool2_fwd");
}
extern "C" int vq_maxpool2_bwd(const void* x, const void* dy, const void* add, void* dx, int N, int H, int W, int C, int dtype,
                               int32_t* range_events, void* stream) {
  return pool_launch<1>(x, dy, add, dx, N, H, W, C, dtype, range_events, stream, "vq_maxpool2_bwd");
}
extern "C" int vq_sumpool2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  return pool_launch<2>(x, nullptr, nullptr, y, N, H, W, C, dtype, nullptr, stream, "vq_sumpool2");
}

// ---- per-channel column sum over pixels (bias gradients) ---------------------------------------
static constexpr int CS_PIX_PER_BLOCK = 1024;   // 2048 left a 1M-pixel tensor with 2 blocks per CU
template <int DT>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ t, int64_t pixels, int C,
                                                      float* __restrict__ part) {
  typedef Store<DT> St;
  __shared__ float red[256 * 8];
  const int slots = C >> 3, tid = threadIdx.x;
  const int nps = 256 / slots;  // pixel lanes; threads of an incomplete last lane stay idle
  const int slot = tid % slots, pl = tid / slots;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  int64_t pbeg = (int64_t)blockIdx.x * CS_PIX_PER_BLOCK, pend = pbeg + CS_PIX_PER_BLOCK;
  if (pend > pixels) pend = pixels;
  if (pl < nps) {
    // four pixels per trip, their loads issued as one batch (vq_gload16_issue): with one load in flight per lane and
    // two blocks per CU this pass ran at ~3 TB/s.  Accumulation order unchanged.
    constexpr int U = 4;
    int64_t pix = pbeg + pl;
    for (; pix + (int64_t)(U - 1) * nps < pend; pix += (int64_t)U * nps) {
      typename St::Raw r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) St::load8_issue(r[u], t, (pix + (int64_t)u * nps) * C + slot * 8);
      vq_raw_wait(r);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[8];
        St::unpack8(r[u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
      }
    }
    for (; pix < pend; pix += nps) {
      float v[8];
      St::load8(t, pix * C + slot * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int q = 0; q < nps; ++q) a += red[(q * slots + (c >> 3)) * 8 + (c & 7)];
    part[(int64_t)blockIdx.x * C + c] = a;
  }
}
// CS_LPI lanes per channel: each sums every CS_LPI-th block partial in fp64, then a shuffle reduction (fixed order).  One
// lane per channel walked up to a thousand partials serially (15-55 us per call).
static constexpr int CS_LPI = 16;
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, int n_out,
                                                               int accumulate, float* __restrict__ out, float alpha,
                                                               const float* __restrict__ alpha_dev) {
  if (alpha_dev) alpha *= *alpha_dev;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = t / CS_LPI, sub = t % CS_LPI;
  const bool live = c < n_out;
  const int cc = live ? c : 0;
  double s = 0.0;
  for (int b = sub; b < nblk; b += CS_LPI) s += (double)part[(int64_t)b * C + cc];
#pragma unroll
  for (int m = 1; m < CS_LPI; m <<= 1) s += __shfl_xor(s, m);
  if (live && sub == 0) out[c] = accumulate ? out[c] + (float)s * alpha : (float)s * alpha;
}
extern "C" size_t vq_colsum_workspace(int64_t pixels, int C) {
  return (size_t)vq_ceil_div(pixels, CS_PIX_PER_BLOCK) * C * sizeof(float) + 64;
}
extern "C" int vq_colsum(const void* t, int64_t pixels, int C, int dtype, float* out, int n_out, int accumulate, float alpha,
                         const float* alpha_dev, void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(t && out && workspace, VQ_ERR_INVALID, "vq_colsum: null pointer");
  const int slots = C / 8;
  VQ_REQUIRE(C % 8 == 0 && C > 0 && slots <= 256 && n_out <= C, VQ_ERR_UNSUPPORTED,
             "vq_colsum: unsupported C=%d n_out=%d", C, n_out);
  VQ_REQUIRE(ws_bytes >= vq_colsum_workspace(pixels, C), VQ_ERR_WORKSPACE, "vq_colsum: workspace too small");
  const int nblk = (int)vq_ceil_div(pixels, CS_PIX_PER_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == VQ_BF16)
    hipLaunchKernelGGL((colsum_kernel<VQ_BF16>), dim3(nblk), dim3(256), 0, s, t, pixels, C, (float*)workspace);
  else if (dtype == VQ_F16)
    hipLaunchKernelGGL((colsum_kernel<VQ_F16>), dim3(nblk), dim3(256), 0, s, t, pixels, C, (float*)workspace);
  else if (dtype == VQ_F32)
    hipLaunchKernelGGL((colsum_kernel<VQ_F32>), dim3(nblk), dim3(256), 0, s, t, pixels, C, (float*)workspace);
  else if (dtype == VQ_F16X2)
    hipLaunchKernelGGL((colsum_kernel<VQ_F16X2>), dim3(nblk), dim3(256), 0, s, t, pixels, C, (float*)workspace);
  else { vq_set_error("vq_colsum: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH("vq_colsum");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((n_out * CS_LPI + 255) / 256), dim3(256), 0, s, (const float*)workspace, nblk, C,
                     n_out, accumulate, out, alpha, alpha_dev);
  VQ_CHECK_LAUNCH("vq_colsum(finalize)");
  return VQ_OK;
}

// ---- max |t| of a tensor (calibration of the VQ_F16 loss scales; never on the step's critical path) ----------------
template <int DT>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ t, int64_t n8, float* __restrict__ out) {
  typedef Store<DT> St;
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    St::load8(t, i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax((unsigned*)out, __float_as_uint(m));   // non-negative floats order like their bit patterns
  }
}
extern "C" int vq_absmax(const void* t, int64_t n, int dtype, float* out, void* stream) {
  VQ_REQUIRE(t && out && n > 0 && n % 8 == 0, VQ_ERR_INVALID, "vq_absmax: null pointer, or n=%lld not a positive multiple of 8", (long long)n);
  const int64_t n8 = n / 8;
  const dim3 grid(stream_grid(n8));
  if (dtype == VQ_BF16) hipLaunchKernelGGL((absmax_kernel<VQ_BF16>), grid, dim3(256), 0, (hipStream_t)stream, t, n8, out);
  else if (dtype == VQ_F16) hipLaunchKernelGGL((absmax_kernel<VQ_F16>), grid, dim3(256), 0, (hipStream_t)stream, t, n8, out);
  else if (dtype == VQ_F32) hipLaunchKernelGGL((absmax_kernel<VQ_F32>), grid, dim3(256), 0, (hipStream_t)stream, t, n8, out);
  else if (dtype == VQ_F16X2) hipLaunchKernelGGL((absmax_kernel<VQ_F16X2>), grid, dim3(256), 0, (hipStream_t)stream, t, n8, out);
  else { vq_set_error("vq_absmax: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH("vq_absmax");
  return VQ_OK;
}
