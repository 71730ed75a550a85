// Self-attention of AttnBlock (ae.py:74-90): F.scaled_dot_product_attention over the H*W tokens of the bottleneck,
// heads of 64 channels ("b (h d) x y -> b h (x y) d"), scale 1/sqrt(head_dim), no mask, no dropout.  The 3-D TVAE's
// block (tae.py:13-57) is the same computation over T*H*W tokens with 8 heads of C/8 channels, hence the head-dim template
// (8, 16, 32 or 64 channels per head).
//
// the block is unreachable in the reference at HEAD (F4) and sits at 32x32 tokens (1024 per image,
// 8 heads at C = 512): 34 GFLOP forward per step at B = 16 — four orders of magnitude below the convolutions — so
// these are plain fp32 VALU kernels (north_star: "MFMA only for the 3x3/1x1 ... contractions"), exact fp32 math
// whatever the storage dtype:
//   fwd      : one thread per query row (q and the output accumulator live in registers), K/V tiles of 32 keys
//              broadcast from LDS, online softmax per tile; writes out[N,T,C] and the log-sum-exp per (head, query);
//   bwd dq   : one thread per query: D = dO.O, p = exp(s - lse), dq += p (dO.v - D) k / 8
//   bwd dk/dv: one thread per key, Q/dO tiles broadcast from LDS; p recomputed from lse (two kernels: 256 VGPRs would
//              not hold k, v, dk and dv of a thread at once).
// qkv is the NHWC output of the 1x1 qkv conv: [N, T, 3C] with the q | k | v channel blocks of qkv.chunk(3, dim=1).
#include "vq_common.h"

// AT_D (a template parameter of everything below) = head_dim: 64 in ae.py:61, in_channels / 8 in tae.py:17-18
static constexpr int AT_TILE = 32;  // keys (or queries) staged per LDS tile

struct AttnParams {
  const void* qkv;    // [N][T][3C]
  const void* out;    // [N][T][C]   (bwd)
  const void* dout;   // [N][T][C]   (bwd)
  void* dst;          // fwd: out [N][T][C];  bwd: dqkv [N][T][3C]
  float* lse;         // [N*heads][T]
  float* dsum;        // [N*heads][T]  D = sum_d dO*O  (written by the dq kernel, read by the dk kernel)
  int N, T, C, heads;
  float scale;        // 1 / sqrt(head_dim)
};

// stage rows [t0, t0+32) of two channel blocks (a: offset ca of tensor A with row stride sa; b likewise) into LDS as fp32
template <int DT, int AT_D>
__device__ __forceinline__ void stage_pair(const void* A, int64_t rowA0, int sa, int ca, const void* B, int64_t rowB0, int sb,
                                           int cb, int t0, int T, float* la, float* lb) {
  constexpr int OCT = AT_D / 8;                              // 32 rows x OCT octets <= 256 threads
  const int r = threadIdx.x / OCT, oct = threadIdx.x % OCT;
  if (r >= AT_TILE) return;
  float va[8], vb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { va[e] = 0.f; vb[e] = 0.f; }
  if (t0 + r < T) {
    Store<DT>::load8(A, (rowA0 + t0 + r) * sa + ca + oct * 8, va);
    Store<DT>::load8(B, (rowB0 + t0 + r) * sb + cb + oct * 8, vb);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { la[r * AT_D + oct * 8 + e] = va[e]; lb[r * AT_D + oct * 8 + e] = vb[e]; }
}

template <int DT, int AT_D>
__device__ __forceinline__ void load_row(const void* base, int64_t elem, float (&v)[AT_D]) {
#pragma unroll
  for (int o = 0; o < AT_D / 8; ++o) {
    float t[8];
    Store<DT>::load8(base, elem + o * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[o * 8 + e] = t[e];
  }
}
template <int DT, int AT_D>
__device__ __forceinline__ void store_row(void* base, int64_t elem, const float (&v)[AT_D]) {
#pragma unroll
  for (int o = 0; o < AT_D / 8; ++o) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = v[o * 8 + e];
    Store<DT>::store8(base, elem + o * 8, t);
  }
}

template <int DT, int AT_D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  __shared__ float ks[AT_TILE * AT_D], vs[AT_TILE * AT_D];
  const int g = blockIdx.y, n = g / p.heads, h = g - n * p.heads;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < p.T;
  const int C3 = 3 * p.C;
  const int64_t row0 = (int64_t)n * p.T;
  float q[AT_D], o[AT_D];
#pragma unroll
  for (int d = 0; d < AT_D; ++d) { q[d] = 0.f; o[d] = 0.f; }
  if (live) load_row<DT, AT_D>(p.qkv, (row0 + t) * C3 + h * AT_D, q);
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < p.T; k0 += AT_TILE) {
    __syncthreads();
    stage_pair<DT, AT_D>(p.qkv, row0, C3, p.C + h * AT_D, p.qkv, row0, C3, 2 * p.C + h * AT_D, k0, p.T, ks, vs);
    __syncthreads();
    float s[AT_TILE];
    float tm = -INFINITY;
#pragma unroll
    for (int j = 0; j < AT_TILE; ++j) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < AT_D; ++d) a = fmaf(q[d], ks[j * AT_D + d], a);
      a = (k0 + j < p.T) ? a * p.scale : -INFINITY;
      s[j] = a;
      tm = fmaxf(tm, a);
    }
    const float mn = fmaxf(m, tm);
    const float corr = expf(m - mn);          // exp(-inf) = 0 on the first tile
    l *= corr;
#pragma unroll
    for (int d = 0; d < AT_D; ++d) o[d] *= corr;
#pragma unroll
    for (int j = 0; j < AT_TILE; ++j) {
      const float pj = expf(s[j] - mn);
      l += pj;
#pragma unroll
      for (int d = 0; d < AT_D; ++d) o[d] = fmaf(pj, vs[j * AT_D + d], o[d]);
    }
    m = mn;
  }
  if (live) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < AT_D; ++d) o[d] *= inv;
    store_row<DT, AT_D>(p.dst, (row0 + t) * p.C + h * AT_D, o);
    p.lse[(int64_t)g * p.T + t] = m + logf(l);
  }
}

template <int DT, int AT_D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnParams p) {
  __shared__ float ks[AT_TILE * AT_D], vs[AT_TILE * AT_D];
  const int g = blockIdx.y, n = g / p.heads, h = g - n * p.heads;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < p.T;
  const int C3 = 3 * p.C;
  const int64_t row0 = (int64_t)n * p.T;
  float q[AT_D], go[AT_D], dq[AT_D];
#pragma unroll
  for (int d = 0; d < AT_D; ++d) { q[d] = 0.f; go[d] = 0.f; dq[d] = 0.f; }
  float lse = 0.f, D = 0.f;
  if (live) {
    load_row<DT, AT_D>(p.qkv, (row0 + t) * C3 + h * AT_D, q);
    load_row<DT, AT_D>(p.dout, (row0 + t) * p.C + h * AT_D, go);
    float ov[AT_D];
    load_row<DT, AT_D>(p.out, (row0 + t) * p.C + h * AT_D, ov);
#pragma unroll
    for (int d = 0; d < AT_D; ++d) D = fmaf(go[d], ov[d], D);
    lse = p.lse[(int64_t)g * p.T + t];
    p.dsum[(int64_t)g * p.T + t] = D;
  }
  for (int k0 = 0; k0 < p.T; k0 += AT_TILE) {
    __syncthreads();
    stage_pair<DT, AT_D>(p.qkv, row0, C3, p.C + h * AT_D, p.qkv, row0, C3, 2 * p.C + h * AT_D, k0, p.T, ks, vs);
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < AT_TILE; ++j) {
      if (k0 + j >= p.T) break;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < AT_D; ++d) { a = fmaf(q[d], ks[j * AT_D + d], a); dp = fmaf(go[d], vs[j * AT_D + d], dp); }
      const float pj = expf(a * p.scale - lse);
      const float ds = pj * (dp - D) * p.scale;
#pragma unroll
      for (int d = 0; d < AT_D; ++d) dq[d] = fmaf(ds, ks[j * AT_D + d], dq[d]);
    }
  }
  if (live) store_row<DT, AT_D>(p.dst, (row0 + t) * C3 + h * AT_D, dq);
}

// WHICH = 0: dv[key] = sum_q p dO_q ;  WHICH = 1: dk[key] = sum_q p (dO_q . v - D_q) q / 8
template <int DT, int AT_D, int WHICH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnParams p) {
  __shared__ float qs[AT_TILE * AT_D], gs[AT_TILE * AT_D];
  __shared__ float ls[AT_TILE], dsm[AT_TILE];
  const int g = blockIdx.y, n = g / p.heads, h = g - n * p.heads;
  const int t = blockIdx.x * 256 + threadIdx.x;      // key index
  const bool live = t < p.T;
  const int C3 = 3 * p.C;
  const int64_t row0 = (int64_t)n * p.T;
  float k[AT_D], v[AT_D], acc[AT_D];
#pragma unroll
  for (int d = 0; d < AT_D; ++d) { k[d] = 0.f; v[d] = 0.f; acc[d] = 0.f; }
  if (live) {
    load_row<DT, AT_D>(p.qkv, (row0 + t) * C3 + p.C + h * AT_D, k);
    if (WHICH == 1) load_row<DT, AT_D>(p.qkv, (row0 + t) * C3 + 2 * p.C + h * AT_D, v);
  }
  for (int q0 = 0; q0 < p.T; q0 += AT_TILE) {
    __syncthreads();
    stage_pair<DT, AT_D>(p.qkv, row0, C3, h * AT_D, p.dout, row0, p.C, h * AT_D, q0, p.T, qs, gs);
    if (threadIdx.x < AT_TILE) {
      const int qi = q0 + threadIdx.x;
      ls[threadIdx.x] = qi < p.T ? p.lse[(int64_t)g * p.T + qi] : 0.f;
      dsm[threadIdx.x] = (WHICH == 1 && qi < p.T) ? p.dsum[(int64_t)g * p.T + qi] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < AT_TILE; ++j) {
      if (q0 + j >= p.T) break;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < AT_D; ++d) a = fmaf(qs[j * AT_D + d], k[d], a);
      const float pj = expf(a * p.scale - ls[j]);
      if (WHICH == 0) {
#pragma unroll
        for (int d = 0; d < AT_D; ++d) acc[d] = fmaf(pj, gs[j * AT_D + d], acc[d]);
      } else {
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < AT_D; ++d) dp = fmaf(gs[j * AT_D + d], v[d], dp);
        const float ds = pj * (dp - dsm[j]) * p.scale;
#pragma unroll
        for (int d = 0; d < AT_D; ++d) acc[d] = fmaf(ds, qs[j * AT_D + d], acc[d]);
      }
    }
  }
  if (live) store_row<DT, AT_D>(p.dst, (row0 + t) * C3 + (WHICH == 0 ? 2 * p.C : p.C) + h * AT_D, acc);
}

static int attn_check(const char* name, const void* qkv, int N, int T, int C, int head_dim, int dtype) {
  VQ_REQUIRE(qkv, VQ_ERR_INVALID, "%s: null pointer", name);
  VQ_REQUIRE(head_dim == 8 || head_dim == 16 || head_dim == 32 || head_dim == 64, VQ_ERR_UNSUPPORTED,
             "%s: head_dim %d is not one of 8, 16, 32, 64", name, head_dim);
  VQ_REQUIRE(N > 0 && T > 0 && C > 0 && C % head_dim == 0, VQ_ERR_UNSUPPORTED,
             "%s: channels must be a positive multiple of the head dim %d (N=%d T=%d C=%d)", name, head_dim, N, T, C);
  VQ_REQUIRE(dtype == VQ_BF16 || dtype == VQ_F16 || dtype == VQ_F32 || dtype == VQ_F16X2, VQ_ERR_INVALID, "%s: unknown dtype %d", name, dtype);
  VQ_REQUIRE((int64_t)N * (C / head_dim) < 65536, VQ_ERR_UNSUPPORTED, "%s: too many (image, head) pairs for one grid", name);
  return VQ_OK;
}

extern "C" size_t vq_attention_workspace(int N, int T, int C, int head_dim) {
  return head_dim > 0 ? (size_t)N * (C / head_dim) * T * sizeof(float) : 0;
}

template <int DT, int AT_D>
static void attn_launch_fwd(const AttnParams& p, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((attn_fwd_kernel<DT, AT_D>), grid, dim3(256), 0, s, p);
}
template <int DT, int AT_D>
static void attn_launch_bwd(const AttnParams& p, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DT, AT_D>), grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DT, AT_D, 0>), grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DT, AT_D, 1>), grid, dim3(256), 0, s, p);
}
// (dtype, head_dim) -> instantiation
#define ATTN_DISPATCH(FN, ...)                                                        \
  do {                                                                                \
    if (dtype == VQ_BF16) {                                                           \
      if (head_dim == 64) FN<VQ_BF16, 64>(__VA_ARGS__);                               \
      else if (head_dim == 32) FN<VQ_BF16, 32>(__VA_ARGS__);                          \
      else if (head_dim == 16) FN<VQ_BF16, 16>(__VA_ARGS__);                          \
      else FN<VQ_BF16, 8>(__VA_ARGS__);                                               \
    } else if (dtype == VQ_F16) {   /* binary16 storage of the "ref" policy: fp32 arithmetic inside, like the other dtypes; the */ \
      if (head_dim == 64) FN<VQ_F16, 64>(__VA_ARGS__);   /* backward is linear in dout, so dqkv carries dout's loss scale   */ \
      else if (head_dim == 32) FN<VQ_F16, 32>(__VA_ARGS__);                           \
      else if (head_dim == 16) FN<VQ_F16, 16>(__VA_ARGS__);                           \
      else FN<VQ_F16, 8>(__VA_ARGS__);                                                \
    } else if (dtype == VQ_F16X2) {   /* the f16x3 policy's storage (hi + lo binary16 pieces); fp32 arithmetic inside like the rest */ \
      if (head_dim == 64) FN<VQ_F16X2, 64>(__VA_ARGS__);                              \
      else if (head_dim == 32) FN<VQ_F16X2, 32>(__VA_ARGS__);                         \
      else if (head_dim == 16) FN<VQ_F16X2, 16>(__VA_ARGS__);                         \
      else FN<VQ_F16X2, 8>(__VA_ARGS__);                                              \
    } else {                                                                          \
      if (head_dim == 64) FN<VQ_F32, 64>(__VA_ARGS__);                                \
      else if (head_dim == 32) FN<VQ_F32, 32>(__VA_ARGS__);                           \
      else if (head_dim == 16) FN<VQ_F32, 16>(__VA_ARGS__);                           \
      else FN<VQ_F32, 8>(__VA_ARGS__);                                                \
    }                                                                                 \
  } while (0)

extern "C" int vq_attention_fwd(const void* qkv, void* out, float* lse, int N, int T, int C, int head_dim, int dtype,
                                void* stream) {
  int rc = attn_check("vq_attention_fwd", qkv, N, T, C, head_dim, dtype);
  if (rc) return rc;
  VQ_REQUIRE(out && lse, VQ_ERR_INVALID, "vq_attention_fwd: null pointer");
  AttnParams p;
  p.qkv = qkv; p.out = nullptr; p.dout = nullptr; p.dst = out; p.lse = lse; p.dsum = nullptr;
  p.N = N; p.T = T; p.C = C; p.heads = C / head_dim; p.scale = 1.f / sqrtf((float)head_dim);
  dim3 grid((unsigned)vq_ceil_div(T, 256), (unsigned)(N * p.heads));
  ATTN_DISPATCH(attn_launch_fwd, p, grid, (hipStream_t)stream);
  VQ_CHECK_LAUNCH("vq_attention_fwd");
  return VQ_OK;
}

extern "C" int vq_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int N, int T,
                                int C, int head_dim, int dtype, void* workspace, size_t ws_bytes, void* stream) {
  int rc = attn_check("vq_attention_bwd", qkv, N, T, C, head_dim, dtype);
  if (rc) return rc;
  VQ_REQUIRE(out && dout && lse && dqkv && workspace, VQ_ERR_INVALID, "vq_attention_bwd: null pointer");
  VQ_REQUIRE(ws_bytes >= vq_attention_workspace(N, T, C, head_dim), VQ_ERR_WORKSPACE, "vq_attention_bwd: workspace too small");
  AttnParams p;
  p.qkv = qkv; p.out = out; p.dout = dout; p.dst = dqkv; p.lse = const_cast<float*>(lse); p.dsum = (float*)workspace;
  p.N = N; p.T = T; p.C = C; p.heads = C / head_dim; p.scale = 1.f / sqrtf((float)head_dim);
  dim3 grid((unsigned)vq_ceil_div(T, 256), (unsigned)(N * p.heads));
  ATTN_DISPATCH(attn_launch_bwd, p, grid, (hipStream_t)stream);
  VQ_CHECK_LAUNCH("vq_attention_bwd");
  return VQ_OK;
}
