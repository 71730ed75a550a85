// FP32GroupNorm (+ swish) forward / backward for NHWC tensors on gfx950.
//
// Replaces ae.py:41-53 (FP32GroupNorm: F.group_norm in fp32, 32 groups, eps 1e-6, affine) and
// ae.py:13-14 (swish) at their 50 call sites (ae.py:131-135, 254-255, 330-331).
// HBM-bound: every pass moves 16 B per lane (8 channels), all cross-block reductions go through fixed-order
// partial buffers (deterministic, no float atomics).
//
// Statistics (round 6): SHIFTED moments merged Chan-style, never `E[x^2] - E[x]^2` of raw fp32 sums — F.group_norm, which the
// reference runs (ae.py:45-53), keeps its accuracy when |mean| >> std (biased layers, smooth images) and so must this.  Every
// producer of partial rows (gn_reduce_kernel<., 0, .> here, the conv epilogues in conv_igemm.hip) accumulates sum(x - p) and
// sum((x - p)^2) about a pivot p that is one of the row's own values, and writes per (row, group) the pair
// (mean_row, M2_row = sum (x - mean_row)^2); gn_stats_finalize_kernel merges the rows in fp64:
//   mean = sum cnt_r mean_r / count,  M2 = sum M2_r + sum cnt_r (mean_r - mean)^2,  var = M2 / count.
//
//   fwd : mu, rstd per (n,g);  y = (x-mu)*rstd*gamma + beta;  s = y*sigmoid(y)
//   bwd : dy = ds * sig(y)*(1 + y*(1-sig(y)));  dgamma_c = sum dy*xhat;  dbeta_c = sum dy
//         a = mean_g(dy*gamma), b = mean_g(dy*gamma*xhat);  dx = rstd*(dy*gamma - a - xhat*b)
#include "vq_common.h"

// Pixels per reduction block: the largest of 256 / 128 / 64 / 32 that still yields >= 1024 blocks.  Big tensors
// stream best with 256 (fewer partials); a fixed 256 left the 512-channel 32x32 layers with 64 blocks of 64 serial
// 16-B loads per lane: 12 us per reduction on a 17 MB tensor (measured 3x faster with 32).
// Pixels per trip of the reduction passes: their loads are issued as a batch (vq_gload16_issue).  Measured on MI355X
// (profiles/r1_gn_batch_v31.txt): statistics pass -13 %, backward reduction -10 %; the same batching left the forward apply
// pass unchanged and made the backward apply pass 7 % slower (150 VGPRs), so those two keep the plain loop.
constexpr int GN_U = 4;    // statistics pass (one tensor)
constexpr int GN_UB = 2;   // backward reduction (two tensors)
static int gn_ppb(int N, int64_t HW, int C) {
  (void)C;
  int p = 256;
  while (p > 32 && (int64_t)N * vq_ceil_div(HW, p) < 1024) p >>= 1;
  return p;
}

// Per-(n, channel) two-moment reduction over a range of pixels.
//   MODE 0: (x, x^2)                       -> statistics
//   MODE 1: (dy, dy*xhat) with dy through silu'   -> backward sums
template <int DT, int MODE, int SILU>
__global__ __launch_bounds__(256) void gn_reduce_kernel(const void* __restrict__ x, const void* __restrict__ dsp,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int64_t HW, int C, int G, int ppb, float* __restrict__ part) {
  typedef Store<DT> St;
  __shared__ __attribute__((aligned(16))) float red[256 * 16];
  // (walking the images in reverse here and forward in the apply pass — to meet the tail of dy the data-gradient conv has just
  // written — was measured equal, 6.1 ms/step either way, and removed: DESIGN.md section 6)
  const int n = (int)blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int slots = C >> 3;
  const int tid = threadIdx.x;
  const int slot = tid % slots, pl = tid / slots, npl = 256 / slots;
  const int Cg = C / G;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  float ga[8], be[8], mu[8], rs[8];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = slot * 8 + e, g = c / Cg;
      ga[e] = gamma[c]; be[e] = beta[c]; mu[e] = mean[n * G + g]; rs[e] = rstd[n * G + g];
    }
  }
  int64_t pbeg = (int64_t)blk * ppb, pend = pbeg + ppb;
  if (pend > HW) pend = HW;
  // MODE 0: moments of every channel about ITS value at the block's first pixel (see the file header)
  float pv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pv[e] = 0.f;
  if (MODE == 0) St::load8(x, ((int64_t)n * HW + pbeg) * C + slot * 8, pv);
  if (pl < npl) {
    // U pixels per trip, all their 16-byte loads issued before the first use; the tail runs pixel by pixel.  The
    // accumulation order is unchanged.
    auto accum = [&](const float (&xv)[8], const float (&dv)[8]) {
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[e] - pv[e]; s1[e] += d; s2[e] += d * d; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xv[e] - mu[e]) * rs[e];
          float dy = dv[e];
          if (SILU) {
            const float y = xh * ga[e] + be[e];
            const float sg = vq_sigmoid(y);
            dy *= sg * (1.f + y * (1.f - sg));
          }
          s1[e] += dy; s2[e] += dy * xh;
        }
      }
    };
    constexpr int U = MODE == 0 ? GN_U : GN_UB;
    int64_t pix = pbeg + pl;
    for (; pix + (int64_t)(U - 1) * npl < pend; pix += (int64_t)npl * U) {
      typename St::Raw xr[U], dr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t off = ((int64_t)n * HW + pix + (int64_t)u * npl) * C + slot * 8;
        St::load8_issue(xr[u], x, off);
        if (MODE == 1) St::load8_issue(dr[u], dsp, off);
      }
      vq_raw_wait(xr);
      if (MODE == 1) vq_raw_wait(dr);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float xv[8], dv[8];
        St::unpack8(xr[u], xv);
        if (MODE == 1) St::unpack8(dr[u], dv);
        accum(xv, dv);
      }
    }
    for (; pix < pend; pix += npl) {
      const int64_t off = ((int64_t)n * HW + pix) * C + slot * 8;
      float xv[8], dv[8];
      St::load8(x, off, xv);
      if (MODE == 1) St::load8(dsp, off, dv);
      accum(xv, dv);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  if constexpr (MODE == 1) {
    for (int c = tid; c < C; c += 256) {
      const int sl = c >> 3, e = c & 7;
      float a = 0.f, b = 0.f;
      for (int q = 0; q < npl; ++q) {
        a += red[(q * slots + sl) * 16 + e];
        b += red[(q * slots + sl) * 16 + 8 + e];
      }
      float* dst = part + (((int64_t)n * nblk + blk) * C + c) * 2;
      dst[0] = a; dst[1] = b;
    }
  } else {
    // per channel (mean_c, M2_c) in fp64 from the shifted sums, then the group's channels merged (equal counts): the pair leaves as
    // fp32.  The fp64 arithmetic is a few operations per CHANNEL of a block, not per element.
    const double cnt = (double)(pend - pbeg);
    double cm[4], cq[4];                             // C <= 1024: at most four channels per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      cm[i] = cq[i] = 0.0;
      if (c < C) {
        const int sl = c >> 3, e = c & 7;
        float a = 0.f, b = 0.f;
        for (int q = 0; q < npl; ++q) {
          a += red[(q * slots + sl) * 16 + e];
          b += red[(q * slots + sl) * 16 + 8 + e];
        }
        const double pc = (double)St::load1(x, ((int64_t)n * HW + pbeg) * C + c), sh = (double)a / cnt;
        cm[i] = pc + sh;
        cq[i] = (double)b - (double)a * sh;
      }
    }
    __syncthreads();                                 // `red` is dead: its 16 KiB now hold the per-channel pairs
    double* dsum = (double*)red;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < C) { dsum[c] = cm[i]; dsum[1024 + c] = cq[i]; }
    }
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
      double ms = 0.0;
      for (int c = g * Cg; c < (g + 1) * Cg; ++c) ms += dsum[c];
      const double mg = ms / (double)Cg;
      double m2 = 0.0;
      for (int c = g * Cg; c < (g + 1) * Cg; ++c) { const double d = dsum[c] - mg; m2 += dsum[1024 + c] + cnt * d * d; }
      float* dst = part + (((int64_t)n * nblk + blk) * G + g) * 2;
      dst[0] = (float)mg; dst[1] = (float)m2;
    }
  }
}

// LPI lanes per (n,g): each lane sums every LPI-th block partial in fp64 (independent chains), then a shuffle
// reduction — the serial per-thread loop over ~256 partials was latency-bound (22 us per call).  LPI = 8 for the image
// tensors (<= 512 partials per sample), a whole wave for video tensors (thousands of partials: 140 us per call with 8).
template <int LPI>
__device__ __forceinline__ double sub_sum(double v) {
#pragma unroll
  for (int m = 1; m < LPI; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
static int gn_finalize_lanes(int nblk) { return nblk > 64 ? 64 : 8; }   // <= 8 serial fp64 adds per lane either way
// Rows (mean_b, M2_b) of `cnt_row` elements each (the last one: whatever is left of `count`) -> mean, rstd of the (n, g) item, merged
// in fp64 about the first row's mean (a shift, so that the merge itself cancels nothing).
template <int LPI>
__global__ void gn_stats_finalize_kernel(const float* __restrict__ part, int N, int nblk, int G, double count, double cnt_row, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / LPI, sub = t % LPI;
  const bool live = i < N * G;
  const int n = live ? i / G : 0, g = live ? i - n * G : 0;
  const double m0 = (double)part[(((int64_t)n * nblk) * G + g) * 2];
  double sa = 0.0, sb = 0.0, sm = 0.0;
  for (int b = sub; b < nblk; b += LPI) {
    const float* src = part + (((int64_t)n * nblk + b) * G + g) * 2;
    double cb = count - (double)b * cnt_row;
    if (cb > cnt_row) cb = cnt_row;
    const double d = (double)src[0] - m0;
    sa += cb * d; sb += cb * d * d; sm += (double)src[1];
  }
  sa = sub_sum<LPI>(sa); sb = sub_sum<LPI>(sb); sm = sub_sum<LPI>(sm);
  if (live && sub == 0) {
    const double sh = sa / count;
    double var = (sm + sb - sa * sh) / count;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)(m0 + sh);
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <int DT, int SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t HW, int C, int G,
                                                        void* __restrict__ y) {
  typedef Store<DT> St;
  // images in REVERSE order: the reduction pass that precedes this kernel streamed them 0..N-1, so the last ones are
  // still in the 256 MiB Infinity Cache when this pass starts (tensors of 270-540 MB do not fit entirely)
  const int n = (int)gridDim.y - 1 - (int)blockIdx.y;
  const int slots = C >> 3, tid = threadIdx.x;
  const int slot = tid % slots, pl = tid / slots, npl = 256 / slots;
  const int Cg = C / G;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = slot * 8 + e, g = c / Cg;
    const float r = rstd[n * G + g], m = mean[n * G + g];
    a[e] = r * gamma[c];
    b[e] = beta[c] - m * a[e];
  }
  if (pl >= npl) return;
  const int64_t stride = (int64_t)gridDim.x * npl;
  for (int64_t pix = (int64_t)blockIdx.x * npl + pl; pix < HW; pix += stride) {
    const int64_t off = ((int64_t)n * HW + pix) * C + slot * 8;
    float v[8];
    St::load8(x, off, v);               // (streaming loads / stores here: measured 2 % slower, removed)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e] * a[e] + b[e];
      v[e] = SILU ? t * vq_sigmoid(t) : t;
    }
    St::store8(y, off, v);
  }
}

// Per-(n, c) totals `nc` from the per-(n, blk, c) partial sums, and — FUSE_COEF — the per-(n, g) coefficients a, b as well: a block
// holds 256 / LPI consecutive channels of one sample, so when the group size divides that, whole groups are block-local and the
// second level needs no launch of its own (it was one: 50 launches per step of pure latency).
template <int LPI, int FUSE_COEF>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int N,
                                                              int nblk, int C, int G, double count, float* __restrict__ coef /* [N][G][2] */,
                                                              float* __restrict__ nc /* [N][C][2] */) {
  // LPI lanes per item (see gn_stats_finalize_kernel)
  __shared__ double wsum[2 * (256 / LPI)];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / LPI, sub = t % LPI;
  const bool live = i < N * C;
  const int n = live ? i / C : 0, c = live ? i - n * C : 0;
  double s1 = 0.0, s2 = 0.0;
  for (int b = sub; b < nblk; b += LPI) {
    const float* src = part + (((int64_t)n * nblk + b) * C + c) * 2;
    s1 += (double)src[0]; s2 += (double)src[1];
  }
  s1 = sub_sum<LPI>(s1); s2 = sub_sum<LPI>(s2);
  if (live && sub == 0) { nc[i * 2] = (float)s1; nc[i * 2 + 1] = (float)s2; }
  if (FUSE_COEF) {
    constexpr int IPB = 256 / LPI;                 // items (channels) per block; C % IPB == 0 and IPB % Cg == 0 (host)
    const int il = threadIdx.x / LPI;
    if (sub == 0) {                                // the same fp32-rounded totals the separate second level would read back
      const double g = live ? (double)gamma[c] : 0.0;
      wsum[il * 2] = g * (double)(float)s1; wsum[il * 2 + 1] = g * (double)(float)s2;
    }
    __syncthreads();
    const int Cg = C / G, gpb = IPB / Cg;
    if ((int)threadIdx.x < gpb) {
      const int i0 = blockIdx.x * IPB + threadIdx.x * Cg;      // first item of this group
      if (i0 < N * C) {
        const int nn = i0 / C, g = (i0 - nn * C) / Cg;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < Cg; ++k) { a += wsum[(threadIdx.x * Cg + k) * 2]; b += wsum[(threadIdx.x * Cg + k) * 2 + 1]; }
        coef[(nn * G + g) * 2] = (float)(a / count);
        coef[(nn * G + g) * 2 + 1] = (float)(b / count);
      }
    }
  }
}
// fallback second level (group sizes that do not tile a finalize block): coefficients only
__global__ void gn_bwd_coef_kernel(const float* __restrict__ nc, const float* __restrict__ gamma, int N, int C, int G, double count,
                                   float* __restrict__ coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int Cg = C / G;
  if (i < N * G) {
    const int n = i / G, g = i - n * G;
    double a = 0.0, b = 0.0;
    for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
      a += (double)gamma[c] * (double)nc[((int64_t)n * C + c) * 2];
      b += (double)gamma[c] * (double)nc[((int64_t)n * C + c) * 2 + 1];
    }
    coef[i * 2] = (float)(a / count);
    coef[i * 2 + 1] = (float)(b / count);
  }
}

// dgamma / dbeta = sum over the samples of the per-(n, c) totals: block (0, 0) of this kernel writes them (C threads x N values)
// instead of a launch of their own.
template <int DT, int SILU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void* __restrict__ x, const void* __restrict__ dsp,
                                                            const void* __restrict__ add, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ coef,
                                                            const float* __restrict__ nc, int64_t HW, int C, int G,
                                                            void* __restrict__ dx, float dx_scale,
                                                            const float* __restrict__ dx_scale_dev, int accumulate,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float pg_scale,
                                                            const float* __restrict__ pg_scale_dev, int* __restrict__ range_events) {
  typedef Store<DT> St;
  unsigned rng = 0u;
  if (dx_scale_dev) dx_scale *= *dx_scale_dev;      // re-bases the branch gradient onto the scale of `add` (1 outside VQ_F16)
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && (dgamma || dbeta)) {
    if (pg_scale_dev) pg_scale *= *pg_scale_dev;    // parameter gradients leave the loss-scaled domain here
    const int N = (int)gridDim.y;
    for (int c = tid; c < C; c += 256) {
      double s1 = 0.0, s2 = 0.0;
      for (int n = 0; n < N; ++n) { s1 += (double)nc[((int64_t)n * C + c) * 2]; s2 += (double)nc[((int64_t)n * C + c) * 2 + 1]; }
      const float g1 = (float)s1 * pg_scale, g2 = (float)s2 * pg_scale;
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + g1 : g1;
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + g2 : g2;
    }
  }
  // images in REVERSE order: the reduction pass that precedes this kernel streamed them 0..N-1, so the last ones are
  // still in the 256 MiB Infinity Cache when this pass starts (tensors of 270-540 MB do not fit entirely)
  const int n = (int)gridDim.y - 1 - (int)blockIdx.y;
  const int slots = C >> 3;
  const int slot = tid % slots, pl = tid / slots, npl = 256 / slots;
  const int Cg = C / G;
  float ga[8], be[8], mu[8], rs[8], ca[8], cb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = slot * 8 + e, g = c / Cg;
    ga[e] = gamma[c]; be[e] = beta[c]; mu[e] = mean[n * G + g]; rs[e] = rstd[n * G + g];
    ca[e] = coef[(n * G + g) * 2]; cb[e] = coef[(n * G + g) * 2 + 1];
  }
  float rsd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) rsd[e] = rs[e] * dx_scale;
  const int64_t stride = (int64_t)gridDim.x * npl;
  for (int64_t pix = (int64_t)blockIdx.x * npl + pl; pl < npl && pix < HW; pix += stride) {
    const int64_t off = ((int64_t)n * HW + pix) * C + slot * 8;
    float xv[8], dv[8], av[8];
    St::load8(x, off, xv);
    St::load8(dsp, off, dv);
    if (add) St::load8(add, off, av);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      float dy = dv[e];
      if (SILU) {
        const float y = xh * ga[e] + be[e];
        const float sg = vq_sigmoid(y);
        dy *= sg * (1.f + y * (1.f - sg));
      }
      float r = rsd[e] * (dy * ga[e] - ca[e] - xh * cb[e]);
      if (add) r += av[e];
      xv[e] = r;
    }
    if constexpr (IsHalfRange<DT>::value) rng = vq_absmax_bits(rng, xv);
    St::store8(dx, off, xv);
  }
  if constexpr (IsHalfRange<DT>::value) { if (range_events) vq_range_events(range_events, rng, rng); }   // (every lane of the block gets here)
}

// ------------------------------------------------------------------------------------ host
static bool gn_shape_ok(int C, int G) {
  if (C <= 0 || C % 8 != 0 || C > 1024 || G <= 0 || C % G != 0) return false;
  return true;   // a block's 256 threads cover floor(256 / (C/8)) pixels at a time; the remainder threads idle
}
static int gn_nblk(int N, int64_t HW, int C) { return (int)vq_ceil_div(HW, gn_ppb(N, HW, C)); }

extern "C" size_t vq_gn_workspace(int N, int64_t HW, int C) {
  // bwd needs the most: part [N][nblk][C][2] + nc [N][C][2] + coef [N][32+..][2]
  const size_t nblk = (size_t)gn_nblk(N, HW, C);
  return ((size_t)N * nblk * C * 2 + (size_t)N * C * 2 + (size_t)N * C * 2) * sizeof(float) + 256;
}

extern "C" int vq_gn_stats(const void* x, int N, int64_t HW, int C, int G, float eps, int dtype, float* mean,
                           float* rstd, void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(x && mean && rstd && workspace, VQ_ERR_INVALID, "vq_gn_stats: null pointer");
  VQ_REQUIRE(gn_shape_ok(C, G), VQ_ERR_UNSUPPORTED, "vq_gn_stats: unsupported C=%d G=%d (need C%%8==0, C <= 1024, C%%G==0)", C, G);
  VQ_REQUIRE(ws_bytes >= vq_gn_workspace(N, HW, C), VQ_ERR_WORKSPACE, "vq_gn_stats: workspace too small");
  const int nblk = gn_nblk(N, HW, C);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  dim3 grid(nblk, N);
  if (dtype == VQ_BF16)
    hipLaunchKernelGGL((gn_reduce_kernel<VQ_BF16, 0, 0>), grid, dim3(256), 0, s, x, (const void*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, HW, C, G, gn_ppb(N, HW, C), part);
  else if (dtype == VQ_F16)
    hipLaunchKernelGGL((gn_reduce_kernel<VQ_F16, 0, 0>), grid, dim3(256), 0, s, x, (const void*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, HW, C, G, gn_ppb(N, HW, C), part);
  else if (dtype == VQ_F32)
    hipLaunchKernelGGL((gn_reduce_kernel<VQ_F32, 0, 0>), grid, dim3(256), 0, s, x, (const void*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, HW, C, G, gn_ppb(N, HW, C), part);
  else if (dtype == VQ_F16X2)
    hipLaunchKernelGGL((gn_reduce_kernel<VQ_F16X2, 0, 0>), grid, dim3(256), 0, s, x, (const void*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, HW, C, G, gn_ppb(N, HW, C), part);
  else { vq_set_error("vq_gn_stats: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
  VQ_CHECK_LAUNCH("vq_gn_stats");
  const double count = (double)HW * (C / G), cnt_row = (double)gn_ppb(N, HW, C) * (C / G);
  if (gn_finalize_lanes(nblk) == 64)
    hipLaunchKernelGGL(gn_stats_finalize_kernel<64>, dim3((N * G * 64 + 255) / 256), dim3(256), 0, s, (const float*)part, N, nblk, G,
                       count, cnt_row, eps, mean, rstd);
  else
    hipLaunchKernelGGL(gn_stats_finalize_kernel<8>, dim3((N * G * 8 + 255) / 256), dim3(256), 0, s, (const float*)part, N, nblk, G,
                       count, cnt_row, eps, mean, rstd);
  VQ_CHECK_LAUNCH("vq_gn_stats(finalize)");
  return VQ_OK;
}

// mean / rstd from partial sums produced elsewhere — the epilogue of the convolution that wrote the tensor (vq_conv2d_fwd's
// gn_partials: [N][tiles][G][2]) — instead of a statistics pass over it
extern "C" int vq_gn_stats_finalize(const float* partials, int N, int tiles, int64_t HW, int C, int G, float eps, float* mean,
                                    float* rstd, void* stream) {
  VQ_REQUIRE(partials && mean && rstd && N > 0 && tiles > 0, VQ_ERR_INVALID, "vq_gn_stats_finalize: null pointer or empty problem");
  VQ_REQUIRE(gn_shape_ok(C, G), VQ_ERR_UNSUPPORTED, "vq_gn_stats_finalize: unsupported C=%d G=%d", C, G);
  VQ_REQUIRE(HW % tiles == 0, VQ_ERR_INVALID, "vq_gn_stats_finalize: %d partial rows do not divide %lld pixels", tiles, (long long)HW);
  const double count = (double)HW * (C / G), cnt_row = count / tiles;     // conv-epilogue rows are all the same length
  hipStream_t s = (hipStream_t)stream;
  if (gn_finalize_lanes(tiles) == 64)
    hipLaunchKernelGGL(gn_stats_finalize_kernel<64>, dim3((N * G * 64 + 255) / 256), dim3(256), 0, s, partials, N, tiles, G, count, cnt_row, eps, mean, rstd);
  else
    hipLaunchKernelGGL(gn_stats_finalize_kernel<8>, dim3((N * G * 8 + 255) / 256), dim3(256), 0, s, partials, N, tiles, G, count, cnt_row, eps, mean, rstd);
  VQ_CHECK_LAUNCH("vq_gn_stats_finalize");
  return VQ_OK;
}

static int gn_apply_grid(int64_t HW, int C) {
  const int npl = 256 / (C / 8);
  int64_t b = vq_ceil_div(HW, (int64_t)npl * 4);  // ~4 pixels per thread
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vq_gn_silu_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              int N, int64_t HW, int C, int G, int C_w, int dtype, int silu, void* y, void* stream) {
  VQ_REQUIRE(x && mean && rstd && gamma && beta && y, VQ_ERR_INVALID, "vq_gn_silu_fwd: null pointer");
  VQ_REQUIRE(gn_shape_ok(C, G) && C_w == C, VQ_ERR_UNSUPPORTED, "vq_gn_silu_fwd: unsupported C=%d C_w=%d G=%d", C, C_w, G);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(gn_apply_grid(HW, C), N);
#define VQ_GA(DTv, SLv) hipLaunchKernelGGL((gn_apply_kernel<DTv, SLv>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, HW, C, G, y)
  if (dtype == VQ_BF16) { if (silu) VQ_GA(VQ_BF16, 1); else VQ_GA(VQ_BF16, 0); }
  else if (dtype == VQ_F16) { if (silu) VQ_GA(VQ_F16, 1); else VQ_GA(VQ_F16, 0); }
  else if (dtype == VQ_F32) { if (silu) VQ_GA(VQ_F32, 1); else VQ_GA(VQ_F32, 0); }
  else if (dtype == VQ_F16X2) { if (silu) VQ_GA(VQ_F16X2, 1); else VQ_GA(VQ_F16X2, 0); }
  else { vq_set_error("vq_gn_silu_fwd: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
#undef VQ_GA
  VQ_CHECK_LAUNCH("vq_gn_silu_fwd");
  return VQ_OK;
}

extern "C" int vq_gn_silu_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const void* add, int N, int64_t HW, int C, int G, int C_w, int dtype,
                              int silu, void* dx, float* dgamma, float* dbeta, int accumulate, float dx_scale,
                              const float* dx_scale_dev, float pg_scale, const float* pg_scale_dev, int32_t* range_events,
                              const float* part_in, int part_rows, void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(x && dy && mean && rstd && gamma && beta && dx && workspace, VQ_ERR_INVALID, "vq_gn_silu_bwd: null pointer");
  VQ_REQUIRE(gn_shape_ok(C, G) && C_w == C, VQ_ERR_UNSUPPORTED, "vq_gn_silu_bwd: unsupported C=%d C_w=%d G=%d", C, C_w, G);
  VQ_REQUIRE(ws_bytes >= vq_gn_workspace(N, HW, C), VQ_ERR_WORKSPACE, "vq_gn_silu_bwd: workspace too small");
  VQ_REQUIRE(part_in == nullptr || part_rows > 0, VQ_ERR_INVALID, "vq_gn_silu_bwd: part_in without part_rows");
  // part_in: the per-channel sums were formed by the data-gradient conv that produced dy (vq_conv2d_fwd, VqGnBwdFuse): no reduction pass
  const int nblk = part_in ? part_rows : gn_nblk(N, HW, C);
  hipStream_t s = (hipStream_t)stream;
  float* part = part_in ? const_cast<float*>(part_in) : (float*)workspace;
  float* nc = (float*)workspace + (size_t)N * gn_nblk(N, HW, C) * C * 2;
  float* coef = nc + (size_t)N * C * 2;
  dim3 grid(nblk, N);
  if (!part_in) {
#define VQ_GR(DTv, SLv) hipLaunchKernelGGL((gn_reduce_kernel<DTv, 1, SLv>), grid, dim3(256), 0, s, x, dy, mean, rstd, gamma, beta, HW, C, G, gn_ppb(N, HW, C), part)
  if (dtype == VQ_BF16) { if (silu) VQ_GR(VQ_BF16, 1); else VQ_GR(VQ_BF16, 0); }
  else if (dtype == VQ_F16) { if (silu) VQ_GR(VQ_F16, 1); else VQ_GR(VQ_F16, 0); }
  else if (dtype == VQ_F32) { if (silu) VQ_GR(VQ_F32, 1); else VQ_GR(VQ_F32, 0); }
  else if (dtype == VQ_F16X2) { if (silu) VQ_GR(VQ_F16X2, 1); else VQ_GR(VQ_F16X2, 0); }
  else { vq_set_error("vq_gn_silu_bwd: unknown dtype %d", dtype); return VQ_ERR_INVALID; }
#undef VQ_GR
  VQ_CHECK_LAUNCH("vq_gn_silu_bwd(reduce)");
  }
  const double count = (double)HW * (C / G);
  {
    // lanes per (n, c) item: a whole wave when there are many partials, but never more than lets whole groups sit in one block
    const int Cg = C / G;
    int lpi = gn_finalize_lanes(nblk);
    bool fuse = false;
    for (int l = lpi; l >= 8; l >>= 1) {
      const int ipb = 256 / l;
      if (ipb % Cg == 0 && C % ipb == 0) { lpi = l; fuse = true; break; }
    }
    const dim3 fgrid((unsigned)vq_ceil_div((int64_t)N * C * lpi, 256));
#define VQ_GF(LPIv, FCv) hipLaunchKernelGGL((gn_bwd_finalize_kernel<LPIv, FCv>), fgrid, dim3(256), 0, s, (const float*)part, gamma, N, nblk, C, G, count, coef, nc)
    if (fuse) { if (lpi == 64) VQ_GF(64, 1); else if (lpi == 32) VQ_GF(32, 1); else if (lpi == 16) VQ_GF(16, 1); else VQ_GF(8, 1); }
    else { if (lpi == 64) VQ_GF(64, 0); else VQ_GF(8, 0); }
#undef VQ_GF
    VQ_CHECK_LAUNCH("vq_gn_silu_bwd(finalize)");
    if (!fuse) {
      hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((N * G + 255) / 256), dim3(256), 0, s, (const float*)nc, gamma, N, C, G, count, coef);
      VQ_CHECK_LAUNCH("vq_gn_silu_bwd(coef)");
    }
  }
  dim3 grid2(gn_apply_grid(HW, C), N);
#define VQ_GB(DTv, SLv) hipLaunchKernelGGL((gn_bwd_apply_kernel<DTv, SLv>), grid2, dim3(256), 0, s, x, dy, add, mean, rstd, gamma, beta, (const float*)coef, (const float*)nc, HW, C, G, dx, dx_scale, dx_scale_dev, accumulate, dgamma, dbeta, pg_scale, pg_scale_dev, (int*)((dtype == VQ_F16 || dtype == VQ_F16X2) ? range_events : nullptr))
  if (dtype == VQ_BF16) { if (silu) VQ_GB(VQ_BF16, 1); else VQ_GB(VQ_BF16, 0); }
  else if (dtype == VQ_F16) { if (silu) VQ_GB(VQ_F16, 1); else VQ_GB(VQ_F16, 0); }
  else if (dtype == VQ_F16X2) { if (silu) VQ_GB(VQ_F16X2, 1); else VQ_GB(VQ_F16X2, 0); }
  else { if (silu) VQ_GB(VQ_F32, 1); else VQ_GB(VQ_F32, 0); }
#undef VQ_GB
  VQ_CHECK_LAUNCH("vq_gn_silu_bwd(apply)");
  return VQ_OK;
}
