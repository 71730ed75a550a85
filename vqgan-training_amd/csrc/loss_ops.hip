// Loss-layer kernels: LPIPS tap, GradNorm norm/scale, z statistics, GAN hinge/BCE statistics.
// HBM- or latency-bound; every cross-block reduction goes through fixed-order partials.
//
// Reference: utils.py:39-57,134-140 (LPIPS tail), utils.py:76-89 (Dropout + 1x1 lin conv),
// vae_trainer.py:27-53 (GradNormFunction), vae_trainer.py:63-90 (gan_disc_loss),
// vae_trainer.py:179-217 (vae_loss_function statistics).
#include "vq_common.h"

// counter-based hash -> Bernoulli(0.5) keep bits, identical in fwd and bwd for the same (seed, index): ONE splitmix64 round per
// group of 8 consecutive channels (bit e of the result's high half belongs to channel e of the group).  One round per ELEMENT
// (three 64-bit multiplies each) made the tap kernels VALU-bound at 1.06 TB/s = 13 % of the HBM peak (profiles/r2j_*).
__device__ __forceinline__ uint32_t dropout_bits8(uint64_t seed, uint64_t group) {
  uint64_t z = seed + group * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

// Pixels per block: 256 on large feature maps; halved down to 32 (one trip of a 512-channel tap) until the launch has ~1024 blocks —
// at 256 the 512-channel taps of a 256 x 256 batch of 16 were 64 and 16 blocks on 256 CUs: 44-56 us per call for 4-17 MB tensors
// (0.55 TB/s, profiles/r6zo_kernel_stats.csv), two thirds of the family's time.  A function of (N, HW) only: vq_lpips_workspace
// sizes the partial rows with it.
static int lp_ppb(int N, int64_t HW) {
  int ppb = 256;
  while (ppb > 32 && (int64_t)N * vq_ceil_div(HW, (int64_t)ppb) < 1024) ppb >>= 1;
  return ppb;
}

// One sub-group of LANES = min(C/8, 8) lanes per pixel, PASSES = C / (8 LANES) 16-byte pieces per lane and tensor, U pixels per
// trip (all their loads issued before the first reduction: 2 U PASSES 16-byte loads in flight per lane).  BWD=0: partial sums of
// the tap value; BWD=1: df0.  (Round 1 kept 8 passes' worth of registers for every C, one pixel per trip and summed the block's
// 256 partials serially in one thread: latency-bound at 1.05 TB/s = 13 % of the HBM peak, profiles/r2j_bench_c3_ref.json.log.)
template <int DT, int LANES, int PASSES, int BWD>
__global__ __launch_bounds__(256) void lpips_tap_kernel(const void* __restrict__ f0, const void* __restrict__ f1,
                                                         const float* __restrict__ w, const float* __restrict__ mask,
                                                         uint64_t seed, const float* __restrict__ gval, int64_t HW, int C,
                                                         float* __restrict__ part, void* __restrict__ df0, int relu_inputs,
                                                         float alpha, int* __restrict__ range_events, int ppb) {
  typedef Store<DT> St;
  unsigned rng = 0u;                                 // VQ_F16 range events of df0 (vq_common.h)
  constexpr int U = PASSES >= 8 ? 1 : PASSES == 4 ? 2 : 4;
  __shared__ float red[4];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int sub = tid % LANES, grp = tid / LANES, ngrp = 256 / LANES;
  int64_t pbeg = (int64_t)blockIdx.x * ppb, pend = pbeg + ppb;
  if (pend > HW) pend = HW;
  float total = 0.f;
  const float ginv = BWD ? gval[n] * 2.0f / (float)HW * alpha : 0.f;   // alpha: loss scale of a VQ_F16 feature stack
  float wv[PASSES][8];                               // the lin weights of this lane's channels
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[ps][e] = w[(ps * LANES + sub) * 8 + e];
  // block-uniform trip count (wave collectives inside); out-of-range pixels contribute zeros
  const int iters = (int)((pend - pbeg + ngrp * U - 1) / (ngrp * U));
  for (int it = 0; it < iters; ++it) {
    float a[U][PASSES][8], b[U][PASSES][8];
    int64_t base[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = pbeg + ((int64_t)it * U + u) * ngrp + grp;
      valid[u] = pix < pend;
      base[u] = ((int64_t)n * HW + (valid[u] ? pix : pbeg)) * C;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int64_t off = base[u] + (ps * LANES + sub) * 8;
        St::load8(f0, off, a[u][ps]);
        St::load8(f1, off, b[u][ps]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        if (!valid[u]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { a[u][ps][e] = 0.f; b[u][ps][e] = 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa += a[u][ps][e] * a[u][ps][e]; sb += b[u][ps][e] * b[u][ps][e]; }
      }
      sa = subgroup_sum<LANES>(sa);
      sb = subgroup_sum<LANES>(sb);
      const float na = sqrtf(sa), nb = sqrtf(sb);
      const float ia = 1.0f / (na + 1e-10f), ib = 1.0f / (nb + 1e-10f);
      float acc = 0.f, dot = 0.f;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int c0 = (ps * LANES + sub) * 8;
        const uint32_t keep = (!mask && seed) ? dropout_bits8(seed, (uint64_t)(base[u] + c0) >> 3) : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float wm = wv[ps][e];
          if (mask) wm *= mask[base[u] + c0 + e];
          else if (seed) wm *= ((keep >> e) & 1u) ? 2.0f : 0.0f;
          const float d = a[u][ps][e] * ia - b[u][ps][e] * ib;
          if (!BWD) acc += wm * d * d;
          else {
            const float q = wm * d * ginv;  // d val / d u_c (times upstream grad)
            dot += q * a[u][ps][e];
            b[u][ps][e] = q;               // reuse storage
          }
        }
      }
      if (!BWD) {
        total += acc;
      } else {
        dot = subgroup_sum<LANES>(dot);
        // du/df: df_j = q_j/n - (sum_c q_c f_c) f_j / (|f| n^2), n = |f| + eps
        const float k2 = na > 0.f ? dot * ia * ia / na : 0.f;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (relu_inputs && a[u][ps][e] <= 0.f) ? 0.f : b[u][ps][e] * ia - k2 * a[u][ps][e];
          if constexpr (IsHalfRange<DT>::value && BWD) { if (valid[u]) rng = vq_absmax_bits(rng, o); }
          if (valid[u]) St::store8(df0, base[u] + (ps * LANES + sub) * 8, o);
        }
      }
    }
  }
  if constexpr (IsHalfRange<DT>::value && BWD) { if (range_events) vq_range_events(range_events, rng, rng); }
  if (!BWD) {
    // fixed-order block sum: butterfly over the 64 lanes of each wave, then the four wave totals in order
    total = wave_sum(total);
    if ((tid & 63) == 0) red[tid >> 6] = total;
    __syncthreads();
    if (tid == 0) part[(int64_t)n * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ void lpips_finalize_kernel(const float* __restrict__ part, int N, int nblk, double inv_hw, float* __restrict__ val) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)part[(int64_t)n * nblk + b];
  val[n] += (float)(s * inv_hw);
}

extern "C" size_t vq_lpips_workspace(int N, int64_t HW) {
  return (size_t)N * (size_t)vq_ceil_div(HW, (int64_t)lp_ppb(N, HW)) * sizeof(float) + 64;
}

template <int BWD>
static int lpips_launch(const void* f0, const void* f1, const float* w, const float* mask, uint64_t seed,
                        const float* gval, int N, int64_t HW, int C, int dtype, float* part, void* df0, int relu_inputs,
                        float alpha, int32_t* range_events, hipStream_t s) {
  // lanes per pixel: C/8 capped at 8 so that a pixel's channels are held in <= 8 register passes
  VQ_REQUIRE(C % 8 == 0 && C >= 8 && C <= 512, VQ_ERR_UNSUPPORTED, "vq_lpips_tap: unsupported C=%d", C);
  int lanes = C / 8;
  if (lanes > 8) lanes = 8;
  VQ_REQUIRE((lanes & (lanes - 1)) == 0 && C % (lanes * 8) == 0 && C / (lanes * 8) <= 8, VQ_ERR_UNSUPPORTED,
             "vq_lpips_tap: unsupported C=%d", C);
  const int ppb = lp_ppb(N, HW);
  dim3 grid((unsigned)vq_ceil_div(HW, (int64_t)ppb), N);
  const int passes = C / (lanes * 8);
  VQ_REQUIRE(passes == 1 || passes == 2 || passes == 4 || passes == 8 || lanes < 8, VQ_ERR_UNSUPPORTED,
             "vq_lpips_tap: C=%d (8 lanes x 1 / 2 / 4 / 8 pieces, or fewer than 64 channels)", C);
  VQ_REQUIRE(lanes == 8 || passes == 1, VQ_ERR_UNSUPPORTED, "vq_lpips_tap: unsupported C=%d", C);
#define VQ_LP(DTv, LN, PS) hipLaunchKernelGGL((lpips_tap_kernel<DTv, LN, PS, BWD>), grid, dim3(256), 0, s, f0, f1, w, mask, seed, gval, HW, C, part, df0, relu_inputs, alpha, (int*)range_events, ppb)
#define VQ_LPD(DTv) do { if (lanes == 8) { if (passes == 1) VQ_LP(DTv, 8, 1); else if (passes == 2) VQ_LP(DTv, 8, 2); else if (passes == 4) VQ_LP(DTv, 8, 4); \
                                           else if (passes == 8) VQ_LP(DTv, 8, 8); else { vq_set_error("vq_lpips_tap: unsupported C=%d", C); return VQ_ERR_UNSUPPORTED; } } \
                         else if (lanes == 4) VQ_LP(DTv, 4, 1); else if (lanes == 2) VQ_LP(DTv, 2, 1); else VQ_LP(DTv, 1, 1); } while (0)
  if (dtype == VQ_BF16) VQ_LPD(VQ_BF16);
  else if (dtype == VQ_F16) VQ_LPD(VQ_F16);
  else if (dtype == VQ_F32) VQ_LPD(VQ_F32);
  else if (dtype == VQ_F16X2) VQ_LPD(VQ_F16X2);
  else { vq_set_error("vq_lpips_tap: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
#undef VQ_LPD
#undef VQ_LP
  VQ_CHECK_LAUNCH("vq_lpips_tap");
  return VQ_OK;
}

extern "C" int vq_lpips_tap_fwd(const void* f0, const void* f1, const float* w, const float* mask, uint64_t seed, int N,
                                int64_t HW, int C, int dtype, float* val, void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(f0 && f1 && w && val && workspace, VQ_ERR_INVALID, "vq_lpips_tap_fwd: null pointer");
  VQ_REQUIRE(ws_bytes >= vq_lpips_workspace(N, HW), VQ_ERR_WORKSPACE, "vq_lpips_tap_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int rc = lpips_launch<0>(f0, f1, w, mask, seed, nullptr, N, HW, C, dtype, (float*)workspace, nullptr, 0, 1.f, nullptr, s);
  if (rc) return rc;
  const int nblk = (int)vq_ceil_div(HW, (int64_t)lp_ppb(N, HW));
  hipLaunchKernelGGL(lpips_finalize_kernel, dim3((N + 63) / 64), dim3(64), 0, s, (const float*)workspace, N, nblk, 1.0 / (double)HW, val);
  VQ_CHECK_LAUNCH("vq_lpips_tap_fwd(finalize)");
  return VQ_OK;
}
extern "C" int vq_lpips_tap_bwd(const void* f0, const void* f1, const float* w, const float* mask, uint64_t seed,
                                const float* gval, int N, int64_t HW, int C, int dtype, int relu_inputs, float alpha,
                                void* df0, int32_t* range_events, void* stream) {
  VQ_REQUIRE(f0 && f1 && w && gval && df0, VQ_ERR_INVALID, "vq_lpips_tap_bwd: null pointer");
  return lpips_launch<1>(f0, f1, w, mask, seed, gval, N, HW, C, dtype, nullptr, df0, relu_inputs, alpha,
                         (dtype == VQ_F16 || dtype == VQ_F16X2) ? range_events : nullptr, (hipStream_t)stream);
}

// ---- single-block-finalised scalar reductions ------------------------------------------------------
// Stage 1 writes per-block partials into `scratch`, the last stage is a 1-block kernel: no atomics.
static constexpr int RED_BLOCKS = 256;

template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&v)[NV], float* dst) {
  __shared__ float sm[4][NV];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = wave_sum(v[i]);
    if (lane == 0) sm[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) dst[threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// sum x^2, sum |x| and the centred second moment of |x| (torch's z.abs().std(), vae_trainer.py:214, is a Welford pass): |x| is
// accumulated about the pivot |x[0]| and everything in fp64 — per lane, across the block and in the partial rows — so that
// std|x| does not come out of  E[x^2] - E[|x|]^2  of rounded fp32 sums when |mean| >> std.
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
  __shared__ double sm[4][4];
  const double piv = (double)fabsf(x[0]);
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double t = (double)x[i], a = fabs(t), d = a - piv;
    v[0] += d * d; v[1] += t * t; v[2] += a; v[3] += d;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if (lane == 0) sm[wv][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
__global__ void moments_finalize_kernel(const double* __restrict__ part, int nblk, int64_t n, float* __restrict__ out4) {
  if (threadIdx.x == 0) {
    double q = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < nblk; ++i) { q += part[i * 4]; b += part[i * 4 + 1]; c += part[i * 4 + 2]; d += part[i * 4 + 3]; }
    double m2 = q - d * d / (double)n;              // sum (|x| - mean|x|)^2, from sums about the pivot
    if (m2 < 0.0) m2 = 0.0;
    out4[0] = (float)m2; out4[1] = (float)b; out4[2] = (float)c; out4[3] = (float)n;
  }
}
// scratch requirement for vq_moments / vq_l2norm: 1024 floats

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  float v[1] = {0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[0] += x[i] * x[i];
  block_reduce_store<1>(v, part + blockIdx.x);
}
__global__ void l2norm_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    double a = 0;
    for (int i = 0; i < nblk; ++i) a += part[i];
    out[0] = (float)sqrt(a);
  }
}
__global__ void scale_by_norm_kernel(const float* __restrict__ g, const float* __restrict__ norm, float weight, int64_t n,
                                     float* __restrict__ dx) {
  const float k = weight / (norm[0] + 1e-8f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = g[i] * k;
}

__global__ void scale_kernel(const float* __restrict__ x, float alpha, const float* __restrict__ alpha_dev, int64_t n,
                             float* __restrict__ out) {
  const float k = alpha_dev ? alpha * alpha_dev[0] : alpha;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] * k;
}
extern "C" int vq_scale(const float* x, float alpha, const float* alpha_dev, int64_t n, float* out, void* stream) {
  VQ_REQUIRE(x && out, VQ_ERR_INVALID, "vq_scale: null pointer");
  if (n <= 0) return VQ_OK;
  int64_t b = vq_ceil_div(n, 256 * 4);
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, alpha, alpha_dev, n, out);
  VQ_CHECK_LAUNCH("vq_scale");
  return VQ_OK;
}

static int red_blocks(int64_t n) {
  int64_t b = vq_ceil_div(n, 256 * 8);
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vq_l2norm(const float* g, int64_t n, float* norm_out, float* scratch, void* stream) {
  VQ_REQUIRE(g && norm_out && scratch, VQ_ERR_INVALID, "vq_l2norm: null pointer (scratch must hold 1024 floats)");
  const int nb = red_blocks(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
  VQ_CHECK_LAUNCH("vq_l2norm");
  hipLaunchKernelGGL(l2norm_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)scratch, nb, norm_out);
  VQ_CHECK_LAUNCH("vq_l2norm(finalize)");
  return VQ_OK;
}
extern "C" int vq_moments(const float* x, int64_t n, float* out4, float* scratch, void* stream) {
  VQ_REQUIRE(x && out4 && scratch && n > 0, VQ_ERR_INVALID, "vq_moments: null pointer or empty tensor (scratch must hold 1024 floats)");
  VQ_REQUIRE(((uintptr_t)scratch & 7) == 0, VQ_ERR_INVALID, "vq_moments: scratch must be 8-byte aligned (fp64 partial rows)");
  const int nb = std::min(red_blocks(n), 1024 / 8);     // four fp64 sums per block in 1024 floats of scratch
  hipLaunchKernelGGL(moments_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (double*)scratch);
  VQ_CHECK_LAUNCH("vq_moments");
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch, nb, n, out4);
  VQ_CHECK_LAUNCH("vq_moments(finalize)");
  return VQ_OK;
}
extern "C" int vq_scale_by_norm(const float* g, const float* norm, float weight, int64_t n, float* dx, void* stream) {
  VQ_REQUIRE(g && norm && dx, VQ_ERR_INVALID, "vq_scale_by_norm: null pointer");
  int64_t b = vq_ceil_div(n, 256 * 4);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(scale_by_norm_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, g, norm, weight, n, dx);
  VQ_CHECK_LAUNCH("vq_scale_by_norm");
  return VQ_OK;
}

// ---- discriminator loss statistics (single block: logits are [B, 256..1024]) ----------------------
__global__ __launch_bounds__(256) void gan_disc_loss_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                             int64_t n, int disc_type, float* __restrict__ out6,
                                                             float* __restrict__ d_real, float* __restrict__ d_fake) {
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const float inv = 0.5f / (float)n;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float r = real[i], f = fake[i];
    float lr, lf, gr, gf;
    if (disc_type == 1) {  // hinge
      lr = fmaxf(1.f - r, 0.f); lf = fmaxf(1.f + f, 0.f);
      gr = (1.f - r) > 0.f ? -1.f : 0.f; gf = (1.f + f) > 0.f ? 1.f : 0.f;
    } else {               // BCE with logits: target 1 for real, 0 for fake
      lr = fmaxf(-r, 0.f) + log1pf(expf(-fabsf(r)));   // softplus(-r)
      lf = fmaxf(f, 0.f) + log1pf(expf(-fabsf(f)));    // softplus(f)
      gr = vq_sigmoid(r) - 1.f; gf = vq_sigmoid(f);
    }
    v[0] += lr; v[1] += lf; v[2] += r; v[3] += f;
    v[4] += (r > 0.f ? 1.f : 0.f) + (f < 0.f ? 1.f : 0.f);
    if (d_real) d_real[i] = gr * inv;
    if (d_fake) d_fake[i] = gf * inv;
  }
  __shared__ float res[5];
  block_reduce_store<5>(v, res);
  __syncthreads();
  if (threadIdx.x == 0) {
    out6[0] = res[0] / (float)n; out6[1] = res[1] / (float)n;
    out6[2] = res[2] / (float)n; out6[3] = res[3] / (float)n;
    out6[4] = res[4]; out6[5] = (float)(2 * n);
  }
}
extern "C" int vq_gan_disc_loss(const float* real, const float* fake, int64_t n, int disc_type, float* out6, float* d_real,
                                float* d_fake, void* stream) {
  VQ_REQUIRE(real && fake && out6 && n > 0, VQ_ERR_INVALID, "vq_gan_disc_loss: null pointer / empty");
  VQ_REQUIRE(disc_type == 0 || disc_type == 1, VQ_ERR_INVALID, "vq_gan_disc_loss: disc_type must be 0 (bce) or 1 (hinge)");
  hipLaunchKernelGGL(gan_disc_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, real, fake, n, disc_type, out6, d_real, d_fake);
  VQ_CHECK_LAUNCH("vq_gan_disc_loss");
  return VQ_OK;
}
