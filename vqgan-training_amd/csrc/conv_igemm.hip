// Implicit-GEMM convolution on MFMA for gfx950 — forward and data-gradient.
//
// Replaces nn.Conv2d forward / autograd-dgrad at every call site listed in include/vqhip.h
// (ae.py:105-117,133-139,146-154,160-166,197-199,230-232,282-284,307-309; utils.py:95-111,
// 148-185).  One kernel family covers 3x3 s1 p1, 3x3 s2 asymmetric pad, 1x1, nearest-2x
// upsample folded into the gather, k=stride patch convs and all data gradients (as a conv over the
// zero-dilated output gradient with 180°-rotated, channel-transposed weights).
//
// GEMM view (per block):   D[cout][pixel] = sum_k  Wt[cout][k] * Xg[pixel][k]
//   k = (r*S + s)*Cin + c  — the gather Xg is never materialised; 8 consecutive channels
//   (16 B of bf16) are the unit of every global load and LDS store.
//   MFMA a-operand = weights (rows = cout), b-operand = pixels  =>  every lane ends up with
//   4 consecutive output channels of one pixel per accumulator quad — a natural NHWC store.
// LDS: two K-contiguous tiles [BC][BK] and [BP][BK] per buffer, 16-byte slots XOR-swizzled so
// that the 16 lanes of a ds_read_b128 group hit 16 distinct slots; register-staged double
// buffering, one barrier per K-chunk.
// Precision: bf16 operands / fp32 accumulate (split=1) or the 3-term bf16 split
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (split=3, fp32 storage) used for the parity mode.
#include "vq_common.h"

struct ConvParams {
  VqConvDesc d;
  const void* x;
  const vq_bf16* w;   // packed [Cout][Kp] (+ lo plane)
  const float* bias;
  const void* residual;
  const void* relu_mask;
  void* y;
  int M;        // N*Ho*Wo
  int HoWo;
  int Kp;       // padded reduction length (multiple of BK)
  int RS;
  int G8;       // Cin / 8
  int dsh;      // log2(dil_in)
  int ush;      // log2(up)
  int n_ctiles, n_ptiles;
  int64_t lo_off;  // element offset of the lo plane in w
};

template <int BK> struct Swz {
  static constexpr int SLOTS = BK / 8;
  static constexpr int RPB = 16 / SLOTS;  // rows per 256-byte bank row
  __device__ static __forceinline__ int elem(int row, int slot) {
    return row * BK + ((slot ^ ((row / RPB) % SLOTS)) << 3);
  }
};

template <int DT, int SPLIT> struct XRegs;
template <> struct XRegs<VQ_BF16, 1> { vq_u4 q; };
template <int SPLIT> struct XRegs<VQ_F32, SPLIT> { vq_f4 a, b; };

template <int DT, int SPLIT, int BC, int BP, int WC, int WP, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  static_assert(SPLIT == 1 || DT == VQ_F32, "split mode needs fp32 storage");
  constexpr int SLOTS = BK / 8;
  constexpr int RPP = 256 / SLOTS;                 // rows per loader pass
  constexpr int XPASS = (BP + RPP - 1) / RPP;
  constexpr int WPASS = (BC + RPP - 1) / RPP;
  constexpr int PLANES = (SPLIT == 3) ? 2 : 1;
  constexpr int TILE = (BC + BP) * BK;             // elements per plane per buffer
  constexpr int FC = WC / 32, FP = WP / 32;
  constexpr int NWP = BP / WP;                     // waves along pixels
  static_assert((BC / WC) * (BP / WP) == 4, "4 waves per block");

  __shared__ __attribute__((aligned(16))) vq_bf16 lds[2 * PLANES * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc0 = (wave / NWP) * WC, wp0 = (wave % NWP) * WP;

  // XCD-aware block -> tile map: blocks that land on one XCD (bid % 8) get a contiguous run of
  // tiles, so the pixel tile shared by the n_ctiles cout-tiles and the weight panel stay in
  // that XCD's L2 (guide §5.5 T1, bijective form).
  const int nblk = p.n_ctiles * p.n_ptiles;
  int t;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ctile = t % p.n_ctiles, ptile = t / p.n_ctiles;
  const int c0 = ctile * BC, p0 = ptile * BP;

  const int slot = tid % SLOTS, lrow = tid / SLOTS;

  // ---- per-row gather state (pixel rows of the X tile) ------------------------------------
  int xn[XPASS], xby[XPASS], xbx[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int row = lrow + i * RPP;
    const int m = p0 + row;
    if (row < BP && m < p.M) {
      const int n = m / p.HoWo, rem = m - n * p.HoWo;
      const int oy = rem / p.d.Wo, ox = rem - oy * p.d.Wo;
      xn[i] = n;
      xby[i] = oy * p.d.stride - p.d.pad_t;
      xbx[i] = ox * p.d.stride - p.d.pad_l;
    } else {
      xn[i] = -1; xby[i] = 0; xbx[i] = 0;
    }
  }
  // incremental (tap, channel-group) walker of this thread's 8-channel slot
  int kc8 = slot % p.G8, ktap = slot / p.G8;
  int kr = ktap / p.d.S, ks = ktap - kr * p.d.S;
  const int dmask = (1 << p.dsh) - 1;
  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;

  // weight rows
  int wrow_g[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    int row = c0 + lrow + i * RPP;
    wrow_g[i] = row < p.d.Cout ? row : p.d.Cout - 1;
  }

  XRegs<DT, SPLIT> xr[XPASS];
  vq_u4 wr[WPASS][PLANES];

  auto issue_loads = [&](int chunk) {
    const bool tap_ok = ktap < p.RS;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      int vy = xby[i] + kr, vx = xbx[i] + ks;
      bool ok = tap_ok && xn[i] >= 0 && vy >= 0 && vx >= 0 && ((vy & dmask) == 0) && ((vx & dmask) == 0);
      vy >>= p.dsh; vx >>= p.dsh;
      ok = ok && vy < Hv && vx < Wv;
      const int iy = vy >> p.ush, ix = vx >> p.ush;
      const int64_t off = ((int64_t)(xn[i] * p.d.H + iy) * p.d.W + ix) * p.d.Cin + (kc8 << 3);
      if constexpr (DT == VQ_BF16) {
        vq_u4 z; z.x = z.y = z.z = z.w = 0u;
        xr[i].q = ok ? *(const vq_u4*)((const vq_bf16*)p.x + off) : z;
      } else {
        vq_f4 z; z.x = z.y = z.z = z.w = 0.f;
        const vq_f4* src = (const vq_f4*)((const float*)p.x + off);
        xr[i].a = ok ? src[0] : z;
        xr[i].b = ok ? src[1] : z;
      }
    }
    const int64_t kcol = (int64_t)chunk * BK + (slot << 3);
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      const vq_bf16* src = p.w + (int64_t)wrow_g[i] * p.Kp + kcol;
      wr[i][0] = *(const vq_u4*)src;
      if constexpr (PLANES == 2) wr[i][1] = *(const vq_u4*)(src + p.lo_off);
    }
    // advance the walker by one chunk
    kc8 += SLOTS;
    while (kc8 >= p.G8) {
      kc8 -= p.G8; ktap++; ks++;
      if (ks == p.d.S) { ks = 0; kr++; }
    }
  };

  auto store_lds = [&](int buf) {
    vq_bf16* base = lds + buf * PLANES * TILE;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
      const int row = lrow + i * RPP;
      if (row < BC) {
        *(vq_u4*)(base + Swz<BK>::elem(row, slot)) = wr[i][0];
        if constexpr (PLANES == 2) *(vq_u4*)(base + TILE + Swz<BK>::elem(row, slot)) = wr[i][1];
      }
    }
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int row = lrow + i * RPP;
      if (row < BP) {
        vq_bf16* dst = base + Swz<BK>::elem(BC + row, slot);
        if constexpr (DT == VQ_BF16) {
          *(vq_u4*)dst = xr[i].q;
        } else {
          const float v[8] = {xr[i].a.x, xr[i].a.y, xr[i].a.z, xr[i].a.w, xr[i].b.x, xr[i].b.y, xr[i].b.z, xr[i].b.w};
          vq_bf16 h[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = f2bf(v[e]);
          vq_u4 q;
          q.x = h[0] | ((unsigned)h[1] << 16); q.y = h[2] | ((unsigned)h[3] << 16);
          q.z = h[4] | ((unsigned)h[5] << 16); q.w = h[6] | ((unsigned)h[7] << 16);
          *(vq_u4*)dst = q;
          if constexpr (PLANES == 2) {
            vq_u4 ql;
            ql.x = pack_bf2(v[0] - bf2f(h[0]), v[1] - bf2f(h[1]));
            ql.y = pack_bf2(v[2] - bf2f(h[2]), v[3] - bf2f(h[3]));
            ql.z = pack_bf2(v[4] - bf2f(h[4]), v[5] - bf2f(h[5]));
            ql.w = pack_bf2(v[6] - bf2f(h[6]), v[7] - bf2f(h[7]));
            *(vq_u4*)(dst + TILE) = ql;
          }
        }
      }
    }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  auto compute = [&](int buf) {
    const vq_bf16* base = lds + buf * PLANES * TILE;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      s16x8 a_hi[FC], b_hi[FP];
      s16x8 a_lo[FC], b_lo[FP];
#pragma unroll
      for (int a = 0; a < FC; ++a) {
        const int e = Swz<BK>::elem(wc0 + a * 32 + fr, kk * 2 + fh);
        a_hi[a] = *(const s16x8*)(base + e);
        if constexpr (PLANES == 2) a_lo[a] = *(const s16x8*)(base + TILE + e);
      }
#pragma unroll
      for (int b = 0; b < FP; ++b) {
        const int e = Swz<BK>::elem(BC + wp0 + b * 32 + fr, kk * 2 + fh);
        b_hi[b] = *(const s16x8*)(base + e);
        if constexpr (PLANES == 2) b_lo[b] = *(const s16x8*)(base + TILE + e);
      }
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (PLANES == 2) {
            acc[a][b] = mfma_32x32x16_bf16(a_lo[a], b_hi[b], acc[a][b]);
            acc[a][b] = mfma_32x32x16_bf16(a_hi[a], b_lo[b], acc[a][b]);
          }
          acc[a][b] = mfma_32x32x16_bf16(a_hi[a], b_hi[b], acc[a][b]);
        }
    }
  };

  const int nchunks = p.Kp / BK;
  issue_loads(0);
  store_lds(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = (c + 1) < nchunks;
    if (more) issue_loads(c + 1);
    compute(c & 1);
    if (more) store_lds((c + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: + bias, + residual, relu, relu-mask, NHWC store (4 channels per lane) ------
  typedef Store<DT> St;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int m = p0 + wp0 + b * 32 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < FC; ++a) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = c0 + wc0 + a * 32 + q * 8 + fh * 4;
        if (co >= p.d.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][q * 4 + e];
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.d.Cout_w) v[e] += p.bias[co + e];
        }
        const int64_t off = (int64_t)m * p.d.Cout + co;
        if (p.residual) {
          float rv[4];
          St::load4(p.residual, off, rv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        if (p.d.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (p.relu_mask) {
          float mv[4];
          St::load4(p.relu_mask, off, mv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
        }
        St::store4(p.y, off, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------ LDS-DMA variant
// Throughput path (bf16 storage, Cin % 64 == 0): both tiles go HBM -> LDS with global_load_lds
// (16 B per lane, no VGPR staging, no ds_write), double-buffered, one barrier per K-chunk.
//   * K runs tap-major: per tap every lane recomputes the gather pointer of its 4 pixel rows once
//     (general stride / dilation / nearest-2x), then only adds 128 B per 64-channel chunk;
//   * out-of-image taps read a 128-byte zero page instead of branching;
//   * the XOR swizzle lives on the SOURCE address (the LDS image of an LDS-DMA is lane-linear):
//     the lane that owns physical slot p of row r fetches logical slot p ^ ((r>>1)&7) — same 128-byte
//     line, so no extra HBM sectors (guide §5.4 rule 21).
__device__ __attribute__((aligned(128))) unsigned int g_vq_zero_page[64];

// NSTAGE = 2: __syncthreads() per chunk (drains the DMA);  NSTAGE = 3: prefetch distance 2 with a
// counted `s_waitcnt vmcnt(N)` + raw s_barrier, so one chunk's DMA stays in flight across the barrier
// (guide §5 "Pipelining across barriers"); needs the dynamic-LDS launch (144 KiB for 128x256x64).
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#ifndef VQ_EMU
  static_assert(N == 0 || N == 5 || N == 6 || N == 8, "add the literal below");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
}
__device__ __forceinline__ void raw_barrier() {
#ifdef VQ_EMU
  __syncthreads();
#else
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}

template <int BC, int BP, int WC, int WP, int NSTAGE, int DBG = 0>
__global__ __launch_bounds__((BC / WC) * (BP / WP) * 64) void conv_igemm_glds_kernel(const ConvParams p) {
  constexpr int BK = 64;
  constexpr int TILE = (BC + BP) * BK;            // bf16 elements per buffer
  constexpr int FC = WC / 32, FP = WP / 32;
  constexpr int NWP = BP / WP;
  constexpr int NW = (BC / WC) * (BP / WP);       // waves per block (4 or 8)
  constexpr int NA = BP / 8 / NW, NB = BC / 8 / NW;   // 1-KiB DMA pieces (8 rows) per wave per chunk
  static_assert(NA >= 1 && NB >= 1 && NA * NW * 8 == BP && NB * NW * 8 == BC, "tile / wave mismatch");

  VQ_DYN_LDS(vq_bf16, lds);                       // NSTAGE * TILE elements, all LDS in ONE array

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc0 = (wave / NWP) * WC, wp0 = (wave % NWP) * WP;

  const int nblk = p.n_ctiles * p.n_ptiles;
  int t;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ctile = t % p.n_ctiles, ptile = t / p.n_ctiles;
  const int c0 = ctile * BC, p0 = ptile * BP;

  const int lr = lane >> 3, lp = lane & 7;        // row within the 8-row piece, physical 16-B slot

  // ---- pixel rows owned by this lane ---------------------------------------------------------
  int xn[NA], xby[NA], xbx[NA];
  const vq_bf16* pa[NA];
  int inca[NA];
  int lsa[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave * NA + i) * 8 + lr;
    lsa[i] = (lp ^ ((row >> 1) & 7)) << 3;        // logical slot (in elements) this lane fetches
    const int m = p0 + row;
    if (m < p.M) {
      const int n = m / p.HoWo, rem = m - n * p.HoWo;
      const int oy = rem / p.d.Wo, ox = rem - oy * p.d.Wo;
      xn[i] = n; xby[i] = oy * p.d.stride - p.d.pad_t; xbx[i] = ox * p.d.stride - p.d.pad_l;
    } else {
      xn[i] = -1; xby[i] = 0; xbx[i] = 0;
    }
  }
  const vq_bf16* pb[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + lr;
    int grow = c0 + row;
    if (grow >= p.d.Cout) grow = p.d.Cout - 1;
    pb[i] = p.w + (int64_t)grow * p.Kp + ((lp ^ ((row >> 1) & 7)) << 3);
  }
  const int dmask = (1 << p.dsh) - 1;
  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const int Hvd = Hv << p.dsh, Wvd = Wv << p.dsh, sh_y = p.dsh + p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;

  const int cpt = p.d.Cin >> 6;                   // chunks per tap
  int tap_r = 0, tap_s = 0, cit = 0;

  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      // branch-free: unsigned compares fold the >= 0 tests, the pointer is a select of two values
      const int vy = xby[i] + tap_r, vx = xbx[i] + tap_s;
      const int ok = (int)((unsigned)vy < (unsigned)Hvd) & (int)((unsigned)vx < (unsigned)Wvd) &
                     (int)(((vy | vx) & dmask) == 0) & (int)(xn[i] >= 0);
      const int iy = vy >> sh_y, ix = vx >> sh_y;
      const int64_t off = (int64_t)((xn[i] * p.d.H + iy) * p.d.W + ix) * p.d.Cin + lsa[i];
      const uintptr_t a_ok = (uintptr_t)(xbase + off), a_zero = (uintptr_t)(zero + lsa[i]);
      pa[i] = (const vq_bf16*)(ok ? a_ok : a_zero);
      inca[i] = ok ? BK : 0;
    }
  };

  bool first_stage = true;
  auto stage = [&](int buf) {
    vq_bf16* base = lds + buf * TILE;
    if (cit == 0) set_tap();
    const bool dma = !(DBG & 1) || first_stage;   // DBG is a compile-time ablation switch (0 in the product)
    first_stage = false;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (dma) glds16(pb[i], base + (wave * NB + i) * 8 * BK);
      pb[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (dma) glds16(pa[i], base + (BC + (wave * NA + i) * 8) * BK);
      pa[i] += inca[i];
    }
    if (++cit == cpt) {
      cit = 0;
      if (++tap_s == p.d.S) { tap_s = 0; ++tap_r; }
    }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  // Software-pipelined fragment schedule: the ds_read_b128s of k-step kk+1 are issued BEFORE the MFMAs of
  // k-step kk (fenced so hipcc keeps that order); the MFMA block then needs only a counted lgkmcnt wait and
  // its 4 x 32 cycles cover the LDS latency of the next fragments.  frag_load(buf, 0) of a chunk is issued
  // right after the barrier, ahead of the next chunk's address math + DMA issue.
  s16x8 af[2][FC], bfr[2][FP];
  auto frag_load = [&](int buf, int kk, int slot) {
    const vq_bf16* base = lds + buf * TILE;
#pragma unroll
    for (int a = 0; a < FC; ++a) af[slot][a] = *(const s16x8*)(base + Swz<BK>::elem(wc0 + a * 32 + fr, kk * 2 + fh));
#pragma unroll
    for (int b = 0; b < FP; ++b) bfr[slot][b] = *(const s16x8*)(base + Swz<BK>::elem(BC + wp0 + b * 32 + fr, kk * 2 + fh));
  };
  auto compute = [&](int buf) {   // expects frag_load(buf, 0, 0) to have been issued
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      if (kk + 1 < BK / 16) frag_load(buf, kk + 1, (kk + 1) & 1);
      vq_sched_fence();
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (!(DBG & 2)) acc[a][b] = mfma_32x32x16_bf16(af[kk & 1][a], bfr[kk & 1][b], acc[a][b]);
#ifndef VQ_EMU
          else asm volatile("" ::"v"(af[kk & 1][a]), "v"(bfr[kk & 1][b]));   // ablation: keep the reads alive
#endif
        }
      vq_sched_fence();
    }
  };

  const int nchunks = p.RS * cpt;
  if constexpr (NSTAGE == 2) {
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      frag_load(c & 1, 0, 0);
      vq_sched_fence();
      if (c + 1 < nchunks) stage((c + 1) & 1);
      compute(c & 1);
      __syncthreads();
    }
  } else {
    // 3-slot ring, prefetch distance 2.  Invariants: the slot staged in iteration c was last read in
    // iteration c-1 (all waves are past that iteration's barrier); chunk c+1 is waited for (own pieces,
    // counted vmcnt leaves only chunk c+2 in flight) before the barrier that precedes its first read.
    stage(0);
    if (nchunks > 1) stage(1);
    if (nchunks > 1) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
    raw_barrier();
    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
      const bool more2 = (c + 2) < nchunks;
      frag_load(cur, 0, 0);
      vq_sched_fence();
      if (more2) stage(cur == 0 ? 2 : cur - 1);
      compute(cur);
      if (more2) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
      raw_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
    }
  }

  typedef Store<VQ_BF16> St;
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int m = p0 + wp0 + b * 32 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < FC; ++a) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = c0 + wc0 + a * 32 + q * 8 + fh * 4;
        if (co >= p.d.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][q * 4 + e];
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.d.Cout_w) v[e] += p.bias[co + e];
        }
        const int64_t off = (int64_t)m * p.d.Cout + co;
        if (p.residual) {
          float rv[4];
          St::load4(p.residual, off, rv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        if (p.d.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (p.relu_mask) {
          float mv[4];
          St::load4(p.relu_mask, off, mv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
        }
        St::store4(p.y, off, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------ weight packing
// fwd: packed[row=co][k=(r*S+s)*Cin_pad+ci] = w[co][ci][r][s]
// dgrad: packed[row=ci][k=(r*S+s)*Cout_pad+co] = w[co][ci][R-1-r][S-1-s]
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout_w, int Cin_w, int R, int S,
                                   int rows_pad, int kch_pad, int Kp, int split, int dgrad,
                                   vq_bf16* __restrict__ out) {
  const int64_t total = (int64_t)rows_pad * Kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kp), k = (int)(i - (int64_t)row * Kp);
    const int tap = k / kch_pad, ch = k - tap * kch_pad;
    float v = 0.f;
    if (tap < R * S) {
      int r = tap / S, s = tap - r * S;
      if (!dgrad) {
        if (row < Cout_w && ch < Cin_w) v = w[(((int64_t)row * Cin_w + ch) * R + r) * S + s];
      } else {
        if (row < Cin_w && ch < Cout_w) v = w[(((int64_t)ch * Cin_w + row) * R + (R - 1 - r)) * S + (S - 1 - s)];
      }
    }
    const vq_bf16 h = f2bf(v);
    out[i] = h;
    if (split == 3) out[total + i] = f2bf(v - bf2f(h));
  }
}

static int kp_of(int R, int S, int kch_pad) { return vq_round_up(R * S * kch_pad, 64); }

extern "C" size_t vq_packed_weight_elems(int rows_pad, int R, int S, int cin_pad, int split) {
  return (size_t)rows_pad * kp_of(R, S, cin_pad) * (split == 3 ? 2 : 1);
}

static int pack_common(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                       int split, void* packed, void* stream, int dgrad) {
  VQ_REQUIRE(w && packed, VQ_ERR_INVALID, "vq_pack_weight: null pointer");
  VQ_REQUIRE(split == 1 || split == 3, VQ_ERR_INVALID, "vq_pack_weight: split must be 1 or 3 (got %d)", split);
  VQ_REQUIRE(Cout_pad % 8 == 0 && Cin_pad % 8 == 0 && Cout_pad >= Cout_w && Cin_pad >= Cin_w, VQ_ERR_INVALID,
             "vq_pack_weight: padded channel counts must be multiples of 8 and >= true counts");
  const int rows = dgrad ? Cin_pad : Cout_pad, kch = dgrad ? Cout_pad : Cin_pad;
  const int Kp = kp_of(R, S, kch);
  const int64_t total = (int64_t)rows * Kp;
  int blocks = (int)vq_ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout_w, Cin_w, R, S,
                     rows, kch, Kp, split, dgrad, (vq_bf16*)packed);
  VQ_CHECK_LAUNCH("vq_pack_weight");
  return VQ_OK;
}
extern "C" int vq_pack_weight_fwd(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                                  int split, void* packed, void* stream) {
  return pack_common(w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, packed, stream, 0);
}
extern "C" int vq_pack_weight_dgrad(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                                    int split, void* packed, void* stream) {
  return pack_common(w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, packed, stream, 1);
}

// ------------------------------------------------------------------------------ dispatch
static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

template <int DT, int SPLIT, int BC, int BP, int WC, int WP, int BK>
static int launch_conv(ConvParams& p, hipStream_t stream) {
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = (int)vq_ceil_div(p.M, BP);
  p.Kp = vq_round_up(p.RS * p.d.Cin, 64);
  const int grid = p.n_ctiles * p.n_ptiles;
  hipLaunchKernelGGL((conv_igemm_kernel<DT, SPLIT, BC, BP, WC, WP, BK>), dim3(grid), dim3(256), 0, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd");
  return VQ_OK;
}

template <int DT, int SPLIT, int BK>
static int dispatch_tile(ConvParams& p, hipStream_t stream) {
  if (p.d.Cout > 64) return launch_conv<DT, SPLIT, 128, 128, 64, 64, BK>(p, stream);
  if (p.d.Cout > 32) return launch_conv<DT, SPLIT, 64, 128, 32, 64, BK>(p, stream);
  return launch_conv<DT, SPLIT, 32, 128, 32, 32, BK>(p, stream);
}

template <int BC, int BP, int WC, int WP, int NSTAGE, int DBG = 0>
static int launch_glds(ConvParams& p, hipStream_t stream) {
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr size_t LDS_BYTES = (size_t)NSTAGE * (BC + BP) * 64 * sizeof(vq_bf16);
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = (int)vq_ceil_div(p.M, BP);
  const int grid = p.n_ctiles * p.n_ptiles;
#ifndef VQ_EMU
  static bool attr_set = false;   // benign race: the attribute call is idempotent
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_glds_kernel<BC, BP, WC, WP, NSTAGE, DBG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) { vq_set_error("vq_conv2d_fwd: cannot reserve %zu B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return VQ_ERR_HIP; }
    attr_set = true;
  }
#endif
  hipLaunchKernelGGL((conv_igemm_glds_kernel<BC, BP, WC, WP, NSTAGE, DBG>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(glds)");
  return VQ_OK;
}
static int g_vq_force_tile = 0;   // test/bench knob (vq_debug_set_conv_tile): low nibble 0 auto, 1 = 128x128x2-stage, 2 = 128x256x3-stage
static int g_vq_dbg = 0;          // bits 4.. of the knob: profiling ablations (ConvParams::dbg)
extern "C" void vq_debug_set_conv_tile(int mode) { g_vq_force_tile = mode & 15; g_vq_dbg = mode >> 4; }
static int dispatch_glds(ConvParams& p, hipStream_t stream) {
  if (p.d.Cout > 64) {
    // 256x256 tile (8 waves x 128x64, 128 KiB LDS): halves the L2->LDS bytes per flop — the 128x128 kernel
    // is L2-bandwidth bound (ablation: DMA-only time == MFMA-only time, ~21 TB/s of tile refills)
    const bool t256 = (g_vq_force_tile == 3) || (g_vq_force_tile == 0 && p.d.Cout % 256 == 0 && p.M >= 32768);
    if (t256) return launch_glds<256, 256, 128, 64, 2>(p, stream);
    const bool big = (g_vq_force_tile == 2);
    if (big) return launch_glds<128, 256, 64, 64, 3>(p, stream);
#ifdef VQ_ABLATION_KERNELS   // profiling-only builds (make ABLATE=1): compile-time ablated copies of the 128x128 kernel
    if (g_vq_dbg == 1) return launch_glds<128, 128, 64, 64, 2, 1>(p, stream);
    if (g_vq_dbg == 2) return launch_glds<128, 128, 64, 64, 2, 2>(p, stream);
    if (g_vq_dbg == 3) return launch_glds<128, 128, 64, 64, 2, 3>(p, stream);
#endif
    return launch_glds<128, 128, 64, 64, 2>(p, stream);
  }
  if (p.d.Cout > 32) return launch_glds<64, 128, 32, 64, 2>(p, stream);
  return launch_glds<32, 128, 32, 32, 2>(p, stream);
}

extern "C" int vq_conv2d_fwd(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias,
                             const void* residual, const void* relu_mask, void* y, void* stream) {
  VQ_REQUIRE(d && x && w_packed && y, VQ_ERR_INVALID, "vq_conv2d_fwd: null pointer");
  VQ_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->Cin > 0 && d->Cout > 0, VQ_ERR_INVALID,
             "vq_conv2d_fwd: channel counts must be positive multiples of 8 (Cin=%d Cout=%d)", d->Cin, d->Cout);
  VQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->R > 0 && d->S > 0, VQ_ERR_INVALID,
             "vq_conv2d_fwd: empty tensor");
  const int dsh = ilog2_exact(d->dil_in), ush = ilog2_exact(d->up);
  VQ_REQUIRE(dsh >= 0 && ush >= 0 && ush <= 1 && d->stride >= 1, VQ_ERR_UNSUPPORTED,
             "vq_conv2d_fwd: dil_in must be a power of two, up in {1,2} (dil_in=%d up=%d)", d->dil_in, d->up);
  VQ_REQUIRE(!(dsh > 0 && ush > 0), VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: dil_in and up cannot be combined");
  VQ_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31), VQ_ERR_UNSUPPORTED,
             "vq_conv2d_fwd: pixel count exceeds int32");
  VQ_REQUIRE(d->Cin_w <= d->Cin && d->Cout_w <= d->Cout, VQ_ERR_INVALID, "vq_conv2d_fwd: true channels exceed padded");
  ConvParams p;
  p.d = *d;
  p.x = x; p.w = (const vq_bf16*)w_packed; p.bias = bias; p.residual = residual; p.relu_mask = relu_mask; p.y = y;
  p.M = d->N * d->Ho * d->Wo;
  p.HoWo = d->Ho * d->Wo;
  p.RS = d->R * d->S;
  p.G8 = d->Cin / 8;
  p.dsh = dsh; p.ush = ush;
  p.Kp = vq_round_up(p.RS * d->Cin, 64);
  p.lo_off = (int64_t)d->Cout * p.Kp;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == VQ_BF16) {
    VQ_REQUIRE(d->split == 1, VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: bf16 storage supports split=1 only");
    const int rc8 = vq_launch_conv_c8(d, x, w_packed, bias, residual, relu_mask, y, s);   // 3-channel image layers
    if (rc8 <= 0) return rc8;
    if (d->Cin % 64 == 0) return dispatch_glds(p, s);
    return dispatch_tile<VQ_BF16, 1, 64>(p, s);
  } else if (d->dtype == VQ_F32) {
    if (d->split == 1) return dispatch_tile<VQ_F32, 1, 64>(p, s);
    if (d->split == 3) return dispatch_tile<VQ_F32, 3, 32>(p, s);
    vq_set_error("vq_conv2d_fwd: split must be 1 or 3 (got %d)", d->split);
    return VQ_ERR_UNSUPPORTED;
  }
  vq_set_error("vq_conv2d_fwd: unknown dtype %d", d->dtype);
  return VQ_ERR_INVALID;
}
