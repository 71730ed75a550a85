// Fused multi-tensor AdamW and the VQ codebook nearest-neighbour lookup (gfx950).
//
// AdamW: vae_trainer.py:455-475 (optimizer_G two param groups, optimizer_D), stepped at
// vae_trainer.py:659,702-704 — torch.optim.AdamW semantics, one launch for all tensors
// (28 B/param of HBM traffic: read p,g,m,v, write p,m,v).
// VQ: not in the reference; the algorithm is pinned by oracle/vq_oracle.c.
#include "vq_common.h"

__global__ __launch_bounds__(256) void adamw_multi_kernel(const VqAdamTensor* __restrict__ table,
                                                           const int64_t* __restrict__ chunk_offsets, int n_tensors,
                                                           int chunk, float lr, float wd, float beta1, float beta2,
                                                           float eps, float bc1, float bc2_sqrt, float grad_scale,
                                                           const int* __restrict__ skip_flags, int n_flags, int skip_stride) {
  // a step whose gradients were clipped by a saturating VQ_F16 store (range events, include/vqhip.h) changes nothing: block-uniform
  for (int f = 0; f < n_flags; ++f)
    if (skip_flags[(int64_t)f * skip_stride] != 0) return;
  const int64_t cid = blockIdx.x;
  // binary search: largest t with chunk_offsets[t] <= cid
  int lo = 0, hi = n_tensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunk_offsets[mid] <= cid) lo = mid; else hi = mid - 1;
  }
  const VqAdamTensor t = table[lo];
  const int64_t beg = (cid - chunk_offsets[lo]) * chunk;
  int64_t end = beg + chunk;
  if (end > t.n) end = t.n;
  const float decay = 1.f - lr * wd, step = lr / bc1;
  auto update = [&](float g, float& p, float& m, float& v) {
    g *= grad_scale;
    p *= decay;
    m = m + (g - m) * (1.f - beta1);                              // lerp_, as torch does
    v = v * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step * (m / denom);
  };
  // 16 bytes per lane and array, two quads per trip (8 loads in flight per lane): scalar 4-byte accesses ran this pass at 3.5-3.8 TB/s
  // (r2 profiles).  Chunks start at multiples of 4 elements of 16-byte aligned buffers (vq_adamw_multi checks); the tail is scalar.
  const bool aligned = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0) && (beg & 3) == 0;
  int64_t i = beg;
  if (aligned) {
    const int64_t nq = (end - beg) >> 2;
    vq_f4* __restrict__ p4 = (vq_f4*)(t.p + beg);
    const vq_f4* __restrict__ g4 = (const vq_f4*)(t.g + beg);
    vq_f4* __restrict__ m4 = (vq_f4*)(t.m + beg);
    vq_f4* __restrict__ v4 = (vq_f4*)(t.v + beg);
    int64_t q = threadIdx.x;
    for (; q + 256 < nq; q += 512) {
      vq_f4 p0 = p4[q], g0 = g4[q], m0 = m4[q], v0 = v4[q], p1 = p4[q + 256], g1 = g4[q + 256], m1 = m4[q + 256], v1 = v4[q + 256];
      update(g0.x, p0.x, m0.x, v0.x); update(g0.y, p0.y, m0.y, v0.y); update(g0.z, p0.z, m0.z, v0.z); update(g0.w, p0.w, m0.w, v0.w);
      update(g1.x, p1.x, m1.x, v1.x); update(g1.y, p1.y, m1.y, v1.y); update(g1.z, p1.z, m1.z, v1.z); update(g1.w, p1.w, m1.w, v1.w);
      p4[q] = p0; m4[q] = m0; v4[q] = v0; p4[q + 256] = p1; m4[q + 256] = m1; v4[q + 256] = v1;
    }
    for (; q < nq; q += 256) {
      vq_f4 p0 = p4[q], g0 = g4[q], m0 = m4[q], v0 = v4[q];
      update(g0.x, p0.x, m0.x, v0.x); update(g0.y, p0.y, m0.y, v0.y); update(g0.z, p0.z, m0.z, v0.z); update(g0.w, p0.w, m0.w, v0.w);
      p4[q] = p0; m4[q] = m0; v4[q] = v0;
    }
    i = beg + (nq << 2);
  }
  for (i += threadIdx.x; i < end; i += 256) {
    float p = t.p[i], m = t.m[i], v = t.v[i];
    update(t.g[i], p, m, v);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
  }
}

extern "C" int vq_adamw_multi(const VqAdamTensor* table, const int64_t* chunk_offsets, int n_tensors, int64_t total_chunks,
                              int chunk, float lr, float wd, float beta1, float beta2, float eps, float bc1, float bc2,
                              float grad_scale, const int32_t* skip_flags, int n_flags, int skip_stride, void* stream) {
  VQ_REQUIRE(table && chunk_offsets && n_tensors > 0 && chunk > 0, VQ_ERR_INVALID, "vq_adamw_multi: bad arguments");
  VQ_REQUIRE(n_flags >= 0 && n_flags <= 64 && (n_flags == 0 || (skip_flags && skip_stride > 0)), VQ_ERR_INVALID,
             "vq_adamw_multi: bad skip flags (n_flags=%d stride=%d)", n_flags, skip_stride);
  VQ_REQUIRE(total_chunks > 0 && total_chunks < (1ll << 31), VQ_ERR_INVALID, "vq_adamw_multi: bad chunk count");
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_offsets,
                     n_tensors, chunk, lr, wd, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale, (const int*)skip_flags,
                     skip_flags ? n_flags : 0, skip_stride);
  VQ_CHECK_LAUNCH("vq_adamw_multi");
  return VQ_OK;
}

// ------------------------------------------------------------------------------------------ VQ
// One thread per token (z row in registers), codebook streamed through LDS in tiles and read
// by broadcast; code range split over blockIdx.y; fixed-order fmaf chains (see oracle/vq_oracle.c):
//   zz = fma-chain_k z_k*z_k ; ee likewise ; dot = fma-chain_k z_k*e_k   (k ascending, start 0)
//   d  = (zz - 2*dot) + ee ;   strict '<' while scanning j ascending => lowest index wins ties.
static constexpr int VQ_TILE = 128;
template <int D>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                          int64_t n_tokens, int n_codes, int codes_per_split,
                                                          float* __restrict__ pmin, int* __restrict__ pidx) {
  __shared__ __attribute__((aligned(16))) float tile[VQ_TILE * D];
  __shared__ float tee[VQ_TILE];
  const int64_t tok = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = tok < n_tokens;
  float zr[D];
  float zz = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) { zr[k] = live ? z[tok * D + k] : 0.f; zz = fmaf(zr[k], zr[k], zz); }
  const int jbeg = blockIdx.y * codes_per_split;
  int jend = jbeg + codes_per_split;
  if (jend > n_codes) jend = n_codes;
  float best = __uint_as_float(0x7f800000u);  // +inf
  int besti = jbeg;
  for (int j0 = jbeg; j0 < jend; j0 += VQ_TILE) {
    const int nt = (jend - j0) < VQ_TILE ? (jend - j0) : VQ_TILE;
    __syncthreads();
    for (int i = threadIdx.x; i < nt * D; i += 256) tile[i] = cb[(int64_t)j0 * D + i];
    __syncthreads();
    if (threadIdx.x < nt) {
      float ee = 0.f;
      for (int k = 0; k < D; ++k) ee = fmaf(tile[threadIdx.x * D + k], tile[threadIdx.x * D + k], ee);
      tee[threadIdx.x] = ee;
    }
    __syncthreads();
    for (int jj = 0; jj < nt; ++jj) {
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) dot = fmaf(zr[k], tile[jj * D + k], dot);
      const float d = (zz - 2.f * dot) + tee[jj];
      if (d < best) { best = d; besti = j0 + jj; }
    }
  }
  if (live) {
    pmin[(int64_t)blockIdx.y * n_tokens + tok] = best;
    pidx[(int64_t)blockIdx.y * n_tokens + tok] = besti;
  }
}

// The same search on the fp32 MFMA (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, the fp32 vector rate with none of the VALU's
// broadcast-read traffic).  It is EXACT f32 — per output element an fmaf chain over k onto the accumulator — so chaining D / 2
// instructions over k = (0,1), (2,3), ... reproduces `dot = fmaf(z_k, e_k, dot)`, k ascending, bit for bit (pinned to silicon by
// tests/test_hw_layout.py probe 5; the indices by tests/test_vq.py against oracle/vq_oracle.c at 8192 x 16384 x 32).
// One wave = 32 tokens (A operand: lane l holds token l & 31, components k = 2 s + (l >> 5): D / 2 registers, resident); the codebook
// goes through LDS in tiles of 128 codes, de-interleaved on the way in ([code][parity][D / 2]) so that a lane's D / 2 B-operand values are
// contiguous; per 32-code block D / 2 MFMAs leave dot(token, code) for 16 tokens x 1 code per lane; d = (zz - 2 dot) + ee and the
// strict '<' scan run on the VALU under the next block's MFMAs (two waves per SIMD); a (d, index) butterfly over the 32 lanes that
// hold one token's codes — smaller d, lower index on ties — ends the split.  zz / ee are the oracle's fmaf chains, on the VALU.
// |e_j|^2 of every code, the oracle's fmaf chain (k ascending): once per lookup, so the search blocks do not each redo it
template <int D>
__global__ __launch_bounds__(256) void vq_code_norms_kernel(const float* __restrict__ cb, int n_codes, float* __restrict__ ee) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_codes) return;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) { const float v = cb[(int64_t)j * D + k]; acc = fmaf(v, v, acc); }
  ee[j] = acc;
}

template <int D>
__global__ __launch_bounds__(256) void vq_nearest_mfma_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                               const float* __restrict__ ee_all, int64_t n_tokens, int n_codes,
                                                               int codes_per_split, float* __restrict__ pmin, int* __restrict__ pidx) {
  constexpr int KS = D / 2, CT = D <= 32 ? 128 : 64;  // codes per tile: two buffers stay within 37 KiB of LDS
  constexpr int RS = D + 4;                           // row stride in LDS: + 4 floats, so that the lanes' 16-byte reads spread over the banks
  constexpr int NF4 = CT * D / 4 / 256;               // float4 pieces of a tile per thread
  static_assert(D % 8 == 0 && D >= 8 && D <= 64 && NF4 >= 1, "code dimension");
  __shared__ __attribute__((aligned(16))) float tile[2][CT * RS];   // [code][parity][KS] (+ pad), two buffers
  __shared__ float tee[2][CT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int64_t tok0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const int64_t tok = tok0 + fr;
  const bool live = tok < n_tokens;
  // A operand + the token's |z|^2 (full chain, k ascending: both halves of the wave compute it)
  float az[KS];
  float zz = 0.f;
  {
    const float* zr = z + (live ? tok : 0) * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float v = live ? zr[k] : 0.f;
      zz = fmaf(v, v, zz);
      if ((k & 1) == fh) az[k >> 1] = v;
    }
  }
  // accumulator element e of a lane = token row (e & 3) + 8 (e >> 2) + 4 fh of the wave's 32
  float zzr[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) zzr[e] = __shfl(zz, (e & 3) + 8 * (e >> 2) + 4 * fh);
  const float inf = __uint_as_float(0x7f800000u);
  float best[16];
  int besti[16];                                      // first code of the 32-code block that holds the best one (+ fr = the code)
#pragma unroll
  for (int e = 0; e < 16; ++e) { best[e] = inf; besti[e] = 0; }
  const int jbeg = blockIdx.y * codes_per_split;
  int jend = jbeg + codes_per_split;
  if (jend > n_codes) jend = n_codes;
  // tile staging through registers: the next tile's global loads are in flight under this tile's MFMAs
  float4 pf[NF4];
  float pe = 0.f;
  auto fetch = [&](int j0) {
    const int nt = (jend - j0) < CT ? (jend - j0) : CT;
#pragma unroll
    for (int u = 0; u < NF4; ++u) {
      const int i = tid + u * 256, code = i / (D / 4), q = i - code * (D / 4);
      pf[u] = code < nt ? *(const float4*)(cb + (int64_t)(j0 + code) * D + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    pe = (tid < nt) ? ee_all[j0 + tid] : inf;         // rows past the split's end: |e|^2 = +inf, never the minimum
  };
  auto stash = [&](int buf) {                       // de-interleave: k = 4q, 4q + 2 -> parity 0; 4q + 1, 4q + 3 -> parity 1
#pragma unroll
    for (int u = 0; u < NF4; ++u) {
      const int i = tid + u * 256, code = i / (D / 4), q = i - code * (D / 4);
      float* dst = &tile[buf][code * RS];
      *(float2*)(dst + 2 * q) = make_float2(pf[u].x, pf[u].z);
      *(float2*)(dst + KS + 2 * q) = make_float2(pf[u].y, pf[u].w);
    }
    if (tid < CT) tee[buf][tid] = pe;
  };
  auto load_b = [&](float (&bz)[KS], int buf, int cb0) {
    const float* bsrc = &tile[buf][(cb0 + fr) * RS + fh * KS];
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) {
      const float4 v = *(const float4*)(bsrc + q * 4);
      bz[q * 4] = v.x; bz[q * 4 + 1] = v.y; bz[q * 4 + 2] = v.z; bz[q * 4 + 3] = v.w;
    }
  };
  if (jbeg < jend) { fetch(jbeg); stash(0); }
  __syncthreads();
  int buf = 0;
  for (int j0 = jbeg; j0 < jend; j0 += CT, buf ^= 1) {
    const bool more = j0 + CT < jend;
    if (more) fetch(j0 + CT);
    float bz[2][KS];
    load_b(bz[0], buf, 0);
#pragma unroll
    for (int c = 0; c < CT / 32; ++c) {                // (all CT rows: rows past the split's end are zeros with |e|^2 = +inf)
      if (c + 1 < CT / 32) load_b(bz[(c + 1) & 1], buf, (c + 1) * 32);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int sidx = 0; sidx < KS; ++sidx) acc = mfma_32x32x2_f32(az[sidx], bz[c & 1][sidx], acc);
      const float ee = tee[buf][c * 32 + fr];
      const int blk = j0 + c * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float d = (zzr[e] - 2.f * acc[e]) + ee;
        const bool lt = d < best[e];
        best[e] = lt ? d : best[e];
        besti[e] = lt ? blk : besti[e];
      }
    }
    if (more) stash(buf ^ 1);                        // (buffer buf ^ 1 was last read one tile ago: behind the barrier below)
    __syncthreads();
  }
  // One token's candidates sit in the 32 lanes of a half wave (one code residue each): smaller d wins, the lower index on ties.
  // Halving exchange: at distance 16, 8, 4, 2 a lane keeps the half of its elements its lane bit selects and merges the partner's
  // copy of them — 8 + 4 + 2 + 1 merges instead of 16 per distance — then one merge at distance 1.
  float b[16];
  int bi[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { b[e] = best[e]; bi[e] = besti[e] + fr; }
#pragma unroll
  for (int lvl = 0; lvl < 4; ++lvl) {
    const int m = 16 >> lvl, n = 8 >> lvl;          // lane distance, elements kept
    const bool up = (fr & m) != 0;
#pragma unroll
    for (int e = 0; e < n; ++e) {
      const float keep = up ? b[e + n] : b[e], send = up ? b[e] : b[e + n];
      const int keepi = up ? bi[e + n] : bi[e], sendi = up ? bi[e] : bi[e + n];
      const float ob = __shfl_xor(send, m);
      const int oi = __shfl_xor(sendi, m);
      const bool take = ob < keep || (ob == keep && oi < keepi);
      b[e] = take ? ob : keep;
      bi[e] = take ? oi : keepi;
    }
  }
  {
    const float ob = __shfl_xor(b[0], 1);
    const int oi = __shfl_xor(bi[0], 1);
    if (ob < b[0] || (ob == b[0] && oi < bi[0])) { b[0] = ob; bi[0] = oi; }
  }
  // the element a lane ends with: bit 3 <- lane bit 4, bit 2 <- bit 3, bit 1 <- bit 2, bit 0 <- bit 1
  const int e_mine = fr >> 1;
  const int64_t t = tok0 + (e_mine & 3) + 8 * (e_mine >> 2) + 4 * fh;
  if ((fr & 1) == 0 && t < n_tokens) {
    pmin[(int64_t)blockIdx.y * n_tokens + t] = b[0];
    pidx[(int64_t)blockIdx.y * n_tokens + t] = bi[0];
  }
}

template <int D>
__global__ void vq_finalize_kernel(const float* __restrict__ pmin, const int* __restrict__ pidx, const float* __restrict__ cb,
                                   int64_t n_tokens, int nsplit, int64_t* __restrict__ idx, float* __restrict__ zq,
                                   float* __restrict__ min_dist) {
  const int64_t tok = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= n_tokens) return;
  float best = pmin[tok];
  int bi = pidx[tok];
  for (int s = 1; s < nsplit; ++s) {
    const float d = pmin[(int64_t)s * n_tokens + tok];
    if (d < best) { best = d; bi = pidx[(int64_t)s * n_tokens + tok]; }
  }
  idx[tok] = bi;
  if (min_dist) min_dist[tok] = best;
  if (zq)
    for (int k = 0; k < D; ++k) zq[tok * D + k] = cb[(int64_t)bi * D + k];
}

static int vq_nsplit(int64_t n_tokens, int n_codes) {
  const int64_t tb = vq_ceil_div(n_tokens, 256);
  int64_t want = vq_ceil_div(1024, tb);
  const int64_t maxs = vq_ceil_div(n_codes, VQ_TILE);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}

static int vq_mfma_blocks() {                      // blocks the fp32-MFMA search aims for
#ifdef VQ_ABLATION_KERNELS                         // (VQ_MFMA_BLOCKS: tools' A/B only — `make ablate` libraries and the emulator)
  static const int v = [] { const char* e = getenv("VQ_MFMA_BLOCKS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 512; }();
  return v;
#else
  return 512;
#endif
}

extern "C" size_t vq_vq_workspace(int64_t n_tokens, int n_codes) {      // split minima + indices, then the codes' squared norms
  return (size_t)vq_nsplit(n_tokens, n_codes) * (size_t)n_tokens * 8 + (size_t)n_codes * 4 + 64;
}

extern "C" int vq_vq_nearest_fwd(const float* z, const float* codebook, int64_t n_tokens, int n_codes, int dim, int64_t* idx,
                                 float* zq, float* min_dist, void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(z && codebook && idx && workspace, VQ_ERR_INVALID, "vq_vq_nearest_fwd: null pointer");
  VQ_REQUIRE(n_tokens > 0 && n_codes > 0, VQ_ERR_INVALID, "vq_vq_nearest_fwd: empty problem");
  VQ_REQUIRE(ws_bytes >= vq_vq_workspace(n_tokens, n_codes), VQ_ERR_WORKSPACE, "vq_vq_nearest_fwd: workspace too small");
  const int nsplit = vq_nsplit(n_tokens, n_codes);
  const int cps = (int)(vq_ceil_div(vq_ceil_div(n_codes, nsplit), VQ_TILE) * VQ_TILE);
  const int ns = (int)vq_ceil_div(n_codes, cps);
  float* pmin = (float*)workspace;
  int* pidx = (int*)(pmin + (size_t)nsplit * n_tokens);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)vq_ceil_div(n_tokens, 256), ns);
  const unsigned fb = (unsigned)vq_ceil_div(n_tokens, 256);
  // code dimensions 8 ... 64 (multiples of 4) on the fp32 MFMA (128 tokens per block, same code splits); 4 keeps the VALU kernel
  // (two blocks per CU — each a long run of code tiles, so the per-block prologue and merge are amortised — and never more splits
  //  than the workspace was sized for)
  int64_t msplit = vq_ceil_div(vq_mfma_blocks(), vq_ceil_div(n_tokens, 128));
  if (msplit > nsplit) msplit = nsplit;
  const int mcps = (int)(vq_ceil_div(vq_ceil_div(n_codes, msplit), VQ_TILE) * VQ_TILE);
  const int mns = (int)vq_ceil_div(n_codes, mcps);
  const dim3 mgrid((unsigned)vq_ceil_div(n_tokens, 128), mns);
  float* ee = (float*)(pidx + (size_t)nsplit * n_tokens);
#define VQ_NM(Dv)                                                                                                  \
  do {                                                                                                             \
    hipLaunchKernelGGL((vq_code_norms_kernel<Dv>), dim3((unsigned)vq_ceil_div(n_codes, 256)), dim3(256), 0, s, codebook, n_codes, ee); \
    hipLaunchKernelGGL((vq_nearest_mfma_kernel<Dv>), mgrid, dim3(256), 0, s, z, codebook, (const float*)ee, n_tokens, n_codes, mcps, pmin, pidx); \
    VQ_CHECK_LAUNCH("vq_vq_nearest_fwd(mfma)");                                                                    \
    hipLaunchKernelGGL((vq_finalize_kernel<Dv>), dim3(fb), dim3(256), 0, s, (const float*)pmin, (const int*)pidx, codebook, \
                       n_tokens, mns, idx, zq, min_dist);                                                          \
    VQ_CHECK_LAUNCH("vq_vq_nearest_fwd(finalize)");                                                                \
    return VQ_OK;                                                                                                  \
  } while (0)
  if (dim == 32) VQ_NM(32);
  if (dim == 16) VQ_NM(16);
  if (dim == 8) VQ_NM(8);
  if (dim == 64) VQ_NM(64);
#undef VQ_NM
#define VQ_NN(Dv)                                                                                                  \
  do {                                                                                                             \
    hipLaunchKernelGGL((vq_nearest_kernel<Dv>), grid, dim3(256), 0, s, z, codebook, n_tokens, n_codes, cps, pmin, pidx); \
    VQ_CHECK_LAUNCH("vq_vq_nearest_fwd");                                                                          \
    hipLaunchKernelGGL((vq_finalize_kernel<Dv>), dim3(fb), dim3(256), 0, s, (const float*)pmin, (const int*)pidx, codebook, \
                       n_tokens, ns, idx, zq, min_dist);                                                           \
    VQ_CHECK_LAUNCH("vq_vq_nearest_fwd(finalize)");                                                                \
  } while (0)
  if (dim == 32) VQ_NN(32);
  else if (dim == 16) VQ_NN(16);
  else if (dim == 8) VQ_NN(8);
  else if (dim == 4) VQ_NN(4);
  else if (dim == 64) VQ_NN(64);
  else { vq_set_error("vq_vq_nearest_fwd: unsupported code dim %d (4,8,16,32,64)", dim); return VQ_ERR_UNSUPPORTED; }
#undef VQ_NN
  return VQ_OK;
}

// dcodebook[idx_i][:] += gq_i[:]   (fp32 atomics; the summation order of tokens that share a
// code is not fixed — indices, not this gradient, carry the bit-exactness requirement)
// Order-independent (hence deterministic) scatter-add: the contributions are accumulated as 64-bit FIXED-POINT integers — integer
// addition is associative, so the atomics may land in any order — at a power-of-two scale chosen from the measured max |gq| such
// that n_tokens of them cannot overflow 2^60; an fp32 value within 2^-23 of the maximum converts exactly, smaller ones are rounded
// at 2^-47 of the maximum (far below fp32's own resolution of the sum).  fp32 atomics made config 5 the one path of the step
// whose bits depended on the scheduling.
__device__ __forceinline__ double vq_fixed_scale(float amax, int64_t n_tokens) {
  int e_amax = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127 + 1;    // |g| < 2^e_amax
  int lg = 0;
  while (((int64_t)1 << lg) < n_tokens) ++lg;
  int e = 60 - e_amax - lg;
  if (!(amax > 0.f) || e_amax == 129) e = 0;
  if (e > 1000) e = 1000;
  if (e < -1000) e = -1000;
  return ldexp(1.0, e);
}
__global__ __launch_bounds__(256) void vq_scatter_fixed_kernel(const float* __restrict__ gq, const int64_t* __restrict__ idx,
                                                                int64_t n_tokens, int n_codes, int dim,
                                                                const float* __restrict__ amax, unsigned long long* __restrict__ acc) {
  const double scale = vq_fixed_scale(*amax, n_tokens);
  const int64_t total = n_tokens * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / dim;
    const int k = (int)(i - t * dim);
    const int64_t code = idx[t];
    if (code >= 0 && code < n_codes) {
      const long long q = (long long)rint((double)gq[i] * scale);
      atomicAdd(acc + code * dim + k, (unsigned long long)q);              // two's complement: wraps like signed addition
    }
  }
}
__global__ __launch_bounds__(256) void vq_scatter_finish_kernel(const unsigned long long* __restrict__ acc, int64_t n, int64_t n_tokens,
                                                                 const float* __restrict__ amax, float* __restrict__ dcb) {
  const double inv = 1.0 / vq_fixed_scale(*amax, n_tokens);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dcb[i] += (float)((double)(long long)acc[i] * inv);
}
extern "C" size_t vq_vq_scatter_workspace(int n_codes, int dim) { return (size_t)n_codes * dim * 8 + 64; }
extern "C" int vq_vq_scatter_add(const float* gq, const int64_t* idx, int64_t n_tokens, int n_codes, int dim, float* dcodebook,
                                 void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(gq && idx && dcodebook && workspace, VQ_ERR_INVALID, "vq_vq_scatter_add: null pointer");
  VQ_REQUIRE(dim > 0 && n_tokens > 0 && n_codes > 0 && (n_tokens * dim) % 8 == 0, VQ_ERR_INVALID,
             "vq_vq_scatter_add: empty problem, or n_tokens * dim not a multiple of 8");
  VQ_REQUIRE(ws_bytes >= vq_vq_scatter_workspace(n_codes, dim), VQ_ERR_WORKSPACE, "vq_vq_scatter_add: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* amax = (float*)workspace;                                          // [0, 64): the measured max |gq|
  unsigned long long* acc = (unsigned long long*)((char*)workspace + 64);
  hipError_t e = hipMemsetAsync(workspace, 0, vq_vq_scatter_workspace(n_codes, dim), s);
  if (e != hipSuccess) { vq_set_error("vq_vq_scatter_add: hipMemsetAsync: %s", hipGetErrorString(e)); return VQ_ERR_HIP; }
  int rc = vq_absmax(gq, n_tokens * dim, VQ_F32, amax, stream);
  if (rc) return rc;
  int64_t b = vq_ceil_div(n_tokens * dim, 256);
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(vq_scatter_fixed_kernel, dim3((unsigned)b), dim3(256), 0, s, gq, idx, n_tokens, n_codes, dim, (const float*)amax, acc);
  int64_t b2 = vq_ceil_div((int64_t)n_codes * dim, 256);
  if (b2 > 2048) b2 = 2048;
  hipLaunchKernelGGL(vq_scatter_finish_kernel, dim3((unsigned)b2), dim3(256), 0, s, (const unsigned long long*)acc,
                     (int64_t)n_codes * dim, n_tokens, (const float*)amax, dcodebook);
  VQ_CHECK_LAUNCH("vq_vq_scatter_add");
  return VQ_OK;
}
