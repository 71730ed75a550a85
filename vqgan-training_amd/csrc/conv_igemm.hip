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
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (split=3, fp32 storage) used for the parity mode; split=6: three
// bf16 pieces per fp32 operand (hi + mid + lo = all 24 mantissa bits) and the six products down to 2^-16
// (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped ones are <= 2^-24 relative) = fp32-exact products.
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
  int d2s, d2s_c;  // depth-to-space epilogue (transposed patch conv, sub-pixel conv): patch size (0 = off), channels per tap
  int sub;         // sub-pixel conv (VqConvDesc.subpix): the window of row block (a,b) = c0 / d2s_c is moved by (a,b)
  int pt_tx, pt_tpi;  // nine-tap kernel: a pixel tile is a (BP/16) x 16 patch of ONE image; patches per image row / per image (0 = linear tiles)
  int wo_shift;    // log2(Wo) when Wo is a power of two (tap3 kernel), else -1
  int pix_wsh, pix_hwsh;   // log2(Wo), log2(Ho * Wo) when both extents are powers of two (depth-to-space epilogue), else -1
  float alpha;             // accumulator scale: VqConvDesc.alpha (0 -> 1) ...
  const float* alpha_dev;  // ... times this device scalar when non-null (1/s_w of a VQ_F16 packed weight)
  // GroupNorm statistics of the OUTPUT from the epilogue (the consumer's gn_reduce pass becomes unnecessary): per (image, pixel
  // tile, group) the sums of y and y^2 over the tile's pixels and the group's channels, in the [N][tiles][G][2] layout of the
  // statistics pass's own partials (gn_silu.hip) — vq_gn_stats_finalize turns them into mean / rstd.
  float* gn_part;          // null: off
  int gn_G, gn_cg;         // groups, channels per group (4, 8, 16 or 32)
  int gn_bp, gn_tiles;     // pixels per tile (= the launched kernel's BP: checked), tiles per image
  int gn_nw;               // partial rows per tile (= the launched kernel's wave count: checked)
  int skip_epilogue;       // ABLATE builds only (kernel_hint dbg 8192..8196), 0 in a release library: 1 = return before the epilogue,
                           // 2 = no global store, 3 = no LDS transposition writes (1-3: WRONG results, they price the epilogue's
                           // parts); 4 = ordinary instead of streaming output stores; 5 = streaming loads of the residual / mask operands
  int* range_events;       // VQ_F16 storage: {saturated stores, fully flushed waves} counters of the tensor's stack (null: not counted)
  // fused GroupNorm-backward sums (VqGnBwdFuse, include/vqhip.h): `residual` then holds the GroupNorm's input x
  int gnb;                 // 0 = off
  const float* gnb_mean; const float* gnb_rstd; const float* gnb_gamma; const float* gnb_beta;
  float* gnb_part;         // [N][gnb_rows][Cout][2]
  int gnb_G, gnb_silu, gnb_rows;   // groups; rows per image = Ho * Wo / 32 (every eligible tile is 32 pixels per wave)
};
#ifdef VQ_ABLATION_KERNELS
#define VQ_SKIP_EPI(p) ((p).skip_epilogue)
#define VQ_GNB(p) ((p).gnb)
#ifdef VQ_STAMPS_ON
// cycle stamps of the LAST block / thread 0 (tools only: `make ablate`, read back with vq_debug_stamps): where a tile's time goes.  (The last block, not
// the first: every first block of a CU runs the once-per-block code — prologue, epilogue — on a cold instruction cache.)
__device__ long long g_vq_stamps[512];
__device__ int g_vq_stamp_n;
#define VQ_STAMP(id) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); \
    const int k_ = g_vq_stamp_n; if (k_ < 512) { g_vq_stamps[k_] = ((long long)(id) << 56) | (t_ & 0x00ffffffffffffffll); g_vq_stamp_n = k_ + 1; } } } while (0)
extern "C" int vq_debug_stamps(long long* out, int max_n) {     // -> number of stamps copied; resets the log
  int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_vq_stamp_n), sizeof(int)) != hipSuccess) return -1;
  if (n > max_n) n = max_n;
  if (n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vq_stamps), sizeof(long long) * n) != hipSuccess) return -1;
  const int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_vq_stamp_n), &zero, sizeof(int)) != hipSuccess) return -1;
  return n;
}
#else
#define VQ_STAMP(id) ((void)0)
#endif
#else
#define VQ_SKIP_EPI(p) 0
// fused GroupNorm-backward sums (VqGnBwdFuse): measured cheaper per layer and 0.9 % SLOWER in the step — and the mere presence of the path
// in this epilogue cost the default step 1.5 % (profiles/r3ac_*, r3ad_*): compiled only into `make ABLATE=1` / `make ablate` builds
#define VQ_GNB(p) 0
#define VQ_STAMP(id) ((void)0)
#endif
__device__ __forceinline__ float conv_alpha(const ConvParams& p) { return p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha; }
// The device-side factor (binary16 weights: 1 / s_w, published by the pack kernels) read at the START of a kernel and turned into a
// wave-uniform value after the kernel's first wait for its tiles.  Read where it is used — at the head of the epilogue, as rounds
// 1-2 did — it is a dependent global load behind an `s_waitcnt vmcnt(0)`: ~1 us of exposed latency per tile on every binary16
// layer (a 128 x 128 tile of the 128-channel layers lasts ~11 us).
__device__ __forceinline__ float conv_alpha_request(const ConvParams& p) { return p.alpha_dev ? *p.alpha_dev : 1.f; }
// (returns the DEVICE factor only, as a wave-uniform value the compiler can keep in a scalar register through the main loop;
// the product with the host factor p.alpha is a vector instruction — gfx950 has no scalar float multiply — and formed in the
// epilogue: done here, the product sat in a VGPR and, in the 256-register kernels, in scratch)
__device__ __forceinline__ float conv_alpha_finish(const ConvParams&, float raw) { return vq_wave_uniform(raw); }
__host__ __device__ constexpr int ilog2_ce(int v) { return v <= 1 ? 0 : 1 + ilog2_ce(v >> 1); }

// NHWC element offset of output (pixel m, channel co).  With the depth-to-space epilogue, "channel"
// co = tap * d2s_c + ci of pixel (oy, ox) lands at channel ci of pixel (oy*d2s + r, ox*d2s + s) of the d2s-times
// larger image (4 consecutive channels never straddle a tap: d2s_c % 8 == 0).
__device__ __forceinline__ int64_t conv_out_offset(const ConvParams& p, int m, int co) {
  if (p.d2s == 0) return (int64_t)m * p.d.Cout + co;
  const int tap = co / p.d2s_c, ci = co - tap * p.d2s_c;
  const int r = tap / p.d2s, s = tap - r * p.d2s;
  const int n = m / p.HoWo, rem = m - n * p.HoWo, oy = rem / p.d.Wo, ox = rem - oy * p.d.Wo;
  return ((int64_t)((n * p.d.Ho + oy) * p.d2s + r) * (p.d.Wo * p.d2s) + ox * p.d2s + s) * p.d2s_c + ci;
}

template <int BK> struct Swz {
  static constexpr int SLOTS = BK / 8;
  static constexpr int RPB = 16 / SLOTS;  // rows per 256-byte bank row
  __device__ static __forceinline__ int elem(int row, int slot) {
    return row * BK + ((slot ^ ((row / RPB) % SLOTS)) << 3);
  }
};

// 16-byte slot (before the swizzle) of k-step kk's fragment half fh (lanes 0-31 / 32-63) inside a 64-wide K chunk.  16-bit storage: the
// chunk is 64 consecutive channels, k-step kk = channels 16 kk .. 16 kk + 15.  VQ_F16X2: the chunk is 32 real channels as
// [h0 l0 h1 l1 h2 l2 h3 l3] (8-channel groups, hi piece then lo piece); "k-step" kk = 2 j + plane reads plane `plane` of the groups
// 2 j (lower lanes) and 2 j + 1 (upper lanes) — the weight fragments are packed to match (pack_x2_frag_offset).
template <bool X2> __host__ __device__ constexpr int frag_slot(int kk, int fh) {
  return X2 ? (((kk >> 1) << 2) | (fh << 1) | (kk & 1)) : ((kk << 1) | fh);
}
// the same as byte-address bits for the kernels that keep fragment addresses in registers: address(kk) = address(0) ^ frag_xor(kk)
template <bool X2> __host__ __device__ constexpr unsigned frag_xor(int kk) {
  return X2 ? (unsigned)(((kk >> 1) << 6) | ((kk & 1) << 4)) : (unsigned)(kk << 5);
}
// The MFMAs of one k-step pair of a VQ_F16X2 chunk: a[0] / b[0] = hi fragments, a[1] / b[1] = lo fragments of the same 16 real channels;
// hi*lo and lo*hi first, hi*hi last (the dropped lo*lo is <= 2^-22 of the product)
template <int FC, int FP>
__device__ __forceinline__ void mfma_x2_pair(f32x16 (&acc)[FC][FP], const s16x8 (&ah)[FC], const s16x8 (&al)[FC], const s16x8 (&bh)[FP],
                                             const s16x8 (&bl)[FP]) {
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b) {
      acc[a][b] = mfma_32x32x16_f16(ah[a], bl[b], acc[a][b]);
      acc[a][b] = mfma_32x32x16_f16(al[a], bh[b], acc[a][b]);
      acc[a][b] = mfma_32x32x16_f16(ah[a], bh[b], acc[a][b]);
    }
}

template <int DT, int SPLIT> struct XRegs;
template <> struct XRegs<VQ_BF16, 1> { vq_u4 q; };
template <> struct XRegs<VQ_F16, 1> { vq_u4 q; };
template <> struct XRegs<VQ_F16X2, 1> { vq_u4 q; };
template <int SPLIT> struct XRegs<VQ_F32, SPLIT> { vq_f4 a, b; };

template <int DT, int BC, int BP, int WC, int WP, int PERM = 0, int MAXU = 4>
__device__ __forceinline__ void igemm_epilogue(const ConvParams& p, vq_bf16* lds, f32x16 (&acc)[WC / 32][WP / 32], int c0, int p0,
                                               int wc0, int wp0, float alpha_in = -0.f, int mbase_in = -1);   // defined with the LDS-DMA kernels below
template <int BC, int BP, int WC, int WP, int PERM = 0>
__device__ __forceinline__ void igemm_epilogue_x2(const ConvParams& p, vq_bf16* lds, f32x16 (&acc)[WC / 32][WP / 32], int c0, int p0,
                                                  int wc0, int wp0, float alpha_in = -0.f, int mbase_in = -1);
// Which pixel of its 32-pixel fragment MFMA column `fr` (= lane & 31) stands for in the nine-tap kernel's WA = 3 variant.  A
// ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (MI355X_MICROARCH.md §LDS): with the
// linear map a group reads halo rows r..r+3, r+12..r+15 and r+22..r+29 (the second patch row starts 18 rows on), two of which
// share a 16-byte slot under the (row >> 1) & 7 swizzle — SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE.  Giving each group
// the 16 pixels of ONE patch row makes its 16 halo rows consecutive, hence conflict-free.
__host__ __device__ constexpr int tap9_perm(int fr) {
  return fr < 4 ? fr : fr < 12 ? 12 + fr : fr < 16 ? fr - 8 : fr < 20 ? 8 + fr : fr < 28 ? fr - 12 : fr;
}

template <int DT, int SPLIT, int BC, int BP, int WC, int WP, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  static_assert(SPLIT == 1 || DT == VQ_F32, "split mode needs fp32 storage");
  constexpr bool X2 = DT == VQ_F16X2;                  // (p.d.Cin, p.G8, p.Kp then count VIRTUAL channels: see vq_conv2d_fwd)
  constexpr int OP = (DT == VQ_F16 || X2) ? VQ_F16 : VQ_BF16;   // MFMA operand type: binary16 for binary16 storage, else bf16
  constexpr int SLOTS = BK / 8;
  constexpr int RPP = 256 / SLOTS;                 // rows per loader pass
  constexpr int XPASS = (BP + RPP - 1) / RPP;
  constexpr int WPASS = (BC + RPP - 1) / RPP;
  constexpr int PLANES = (SPLIT == 6) ? 3 : (SPLIT == 3) ? 2 : 1;
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
  const int sub_ph = p.sub ? c0 / p.d2s_c : 0, sub_a = sub_ph >> 1, sub_b = sub_ph & 1;   // block-uniform phase (sub-pixel conv)

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
      xby[i] = oy * p.d.stride - p.d.pad_t + sub_a;
      xbx[i] = ox * p.d.stride - p.d.pad_l + sub_b;
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
      if constexpr (DT != VQ_F32) {
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
      if constexpr (PLANES >= 2) wr[i][1] = *(const vq_u4*)(src + p.lo_off);
      if constexpr (PLANES == 3) wr[i][2] = *(const vq_u4*)(src + 2 * p.lo_off);
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
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) *(vq_u4*)(base + pl * TILE + Swz<BK>::elem(row, slot)) = wr[i][pl];
      }
    }
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int row = lrow + i * RPP;
      if (row < BP) {
        vq_bf16* dst = base + Swz<BK>::elem(BC + row, slot);
        if constexpr (DT != VQ_F32) {
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
          if constexpr (PLANES >= 2) {
            float r1[8];                           // v - hi: exact in fp32
            vq_bf16 m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { r1[e] = v[e] - bf2f(h[e]); m[e] = f2bf(r1[e]); }
            vq_u4 ql;
            ql.x = m[0] | ((unsigned)m[1] << 16); ql.y = m[2] | ((unsigned)m[3] << 16);
            ql.z = m[4] | ((unsigned)m[5] << 16); ql.w = m[6] | ((unsigned)m[7] << 16);
            *(vq_u4*)(dst + TILE) = ql;
            if constexpr (PLANES == 3) {           // the last 8 mantissa bits
              vq_u4 qm;
              qm.x = pack_bf2(r1[0] - bf2f(m[0]), r1[1] - bf2f(m[1]));
              qm.y = pack_bf2(r1[2] - bf2f(m[2]), r1[3] - bf2f(m[3]));
              qm.z = pack_bf2(r1[4] - bf2f(m[4]), r1[5] - bf2f(m[5]));
              qm.w = pack_bf2(r1[6] - bf2f(m[6]), r1[7] - bf2f(m[7]));
              *(vq_u4*)(dst + 2 * TILE) = qm;
            }
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
    if constexpr (X2) {                                // per pair of k-steps: hi and lo fragments of 16 real channels, three products
#pragma unroll
      for (int j = 0; j < BK / 32; ++j) {
        s16x8 ax[2][FC], bx[2][FP];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int a = 0; a < FC; ++a) ax[pl][a] = *(const s16x8*)(base + Swz<BK>::elem(wc0 + a * 32 + fr, frag_slot<true>(2 * j + pl, fh)));
#pragma unroll
          for (int b = 0; b < FP; ++b) bx[pl][b] = *(const s16x8*)(base + Swz<BK>::elem(BC + wp0 + b * 32 + fr, frag_slot<true>(2 * j + pl, fh)));
        }
        mfma_x2_pair<FC, FP>(acc, ax[0], ax[1], bx[0], bx[1]);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      s16x8 af[PLANES][FC], bf[PLANES][FP];
#pragma unroll
      for (int a = 0; a < FC; ++a) {
        const int e = Swz<BK>::elem(wc0 + a * 32 + fr, kk * 2 + fh);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) af[pl][a] = *(const s16x8*)(base + pl * TILE + e);
      }
#pragma unroll
      for (int b = 0; b < FP; ++b) {
        const int e = Swz<BK>::elem(BC + wp0 + b * 32 + fr, kk * 2 + fh);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) bf[pl][b] = *(const s16x8*)(base + pl * TILE + e);
      }
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          // split modes: every product of pieces (pa, pb) with pa + pb < PLANES, smallest terms first (plane 0 = hi)
#pragma unroll
          for (int sum = PLANES - 1; sum >= 1; --sum)
#pragma unroll
            for (int pa = sum; pa >= 0; --pa) acc[a][b] = mfma_32x32x16_bf16(af[pa][a], bf[sum - pa][b], acc[a][b]);
          acc[a][b] = mfma16<OP>(af[0][a], bf[0][b], acc[a][b]);
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

  if constexpr (X2) {
    igemm_epilogue_x2<BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0);
    return;
  } else if constexpr (DT != VQ_F32) {   // 16-bit storage: the coalesced LDS-transposed epilogue of the LDS-DMA kernels
    igemm_epilogue<DT, BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0);
    return;
  }
  // ---- epilogue (fp32 storage): + bias, + residual, relu, relu-mask, NHWC store (4 channels per lane) ------
  typedef Store<DT> St;
  const int fr = lane & 31, fh = lane >> 5;
  const float alpha = conv_alpha(p);
  const float* bias = p.bias;                      // sub-pixel conv: the Cout/4 bias entries serve all four phase blocks
  if (bias && p.sub) bias -= (c0 / p.d2s_c) * p.d2s_c;
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
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][q * 4 + e] * alpha;
        if (bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.d.Cout_w) v[e] += bias[co + e];
        }
        const int64_t off = conv_out_offset(p, m, co);
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

// ---- epilogue of the LDS-DMA kernels ---------------------------------------------------------------------------
// The accumulators hold 4 consecutive couts of ONE pixel per lane: stored directly, a wave instruction touches 32
// different pixel rows with 16 B each (and a residual / mask read does the same).  Instead the tile (+ bias) is
// transposed through the now idle LDS as bf16 [pixel][cout] with the 16-byte slot index XOR-swizzled by the pixel
// row, and re-read so that consecutive lanes own consecutive 16-byte pieces of a pixel row: residual, mask and
// output move in fully coalesced 16 B/lane accesses.  (With a residual the sum is rounded twice, bf16(bf16(acc +
// bias) + res): one extra bf16 ulp at most, throughput mode only — the parity mode runs conv_igemm_kernel.)
// Precondition: every wave of the block is past the last barrier of the main loop (the tiles in `lds` are dead).
template <int DT, int BC, int BP, int WC, int WP, int PERM, int MAXU>
__device__ __forceinline__ void igemm_epilogue(const ConvParams& p, vq_bf16* lds, f32x16 (&acc)[WC / 32][WP / 32], int c0, int p0,
                                               int wc0, int wp0, float alpha_in, int mbase_in) {
  constexpr int FC = WC / 32, FP = WP / 32, NW = (BC / WC) * (BP / WP);
  const int tid = threadIdx.x, lane = tid & 63;
  const int fr = lane & 31, fh = lane >> 5;
  typedef Store<DT> St;
  if (VQ_SKIP_EPI(p) == 1) {                       // measurement knob: keep the accumulators alive, write nothing
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
      for (int b = 0; b < FP; ++b) s += acc[a][b][0] + acc[a][b][15];
    if (s == 123456.789f) ((float*)p.y)[0] = s;
    return;
  }
  // (-0.f = "not supplied": the kernels that matter read the DEVICE factor at their start, conv_alpha_request / conv_alpha_finish)
  const float alpha = __float_as_uint(alpha_in) == 0x80000000u ? conv_alpha(p) : p.alpha * alpha_in;
  // VQ_F16 range events (vq_common.h): what this wave stored, seen on the PACKED binary16 results of the final stores (one and + one
  // v_pk_max_u16 per pair) plus an OR over a quarter of the accumulators ("was anything non-zero to begin with"); one wave-level
  // test at the very end, no atomics in a healthy step.  (Rounds 1-2 tracked fp32 magnitudes at both rounding points: 4 VALU
  // instructions per element, a quarter of this epilogue's arithmetic on the binary16 layers.)
  const bool count_range = DT == VQ_F16 && p.range_events != nullptr;      // block-uniform
  unsigned rng_or = 0u, rng_pk = 0u;
  constexpr int SPRW = BC / 8;                     // 16-byte slots per tile row
  constexpr int NT = NW * 64;
  vq_bf16* ot = lds;                               // [BP][BC], all waves are past the last barrier: the tiles are dead
  const float* bias = p.bias;                      // sub-pixel conv: the Cout/4 bias entries serve all four phase blocks
  if (bias && p.sub) bias -= (c0 / p.d2s_c) * p.d2s_c;
  constexpr int ITEMS = BP * SPRW / NT, U = (ITEMS % 8 == 0 && MAXU >= 8) ? 8 : (ITEMS % 4 == 0 && MAXU >= 4) ? 4 : ((ITEMS % 2 == 0 && MAXU >= 2) ? 2 : 1), ROUNDS = ITEMS / U;
  static_assert(ITEMS * NT == BP * SPRW, "tile / thread-count mismatch");
  static_assert(NT % SPRW == 0, "a thread keeps one 8-channel slot across its items (bias and GroupNorm partials rely on it)");
  // ---- everything the second phase reads from global memory is requested HERE, before the transposition: the bias of this thread's
  // 8-channel slot and the residual / mask pieces of its first round of items.  Requested where they are used (bias inside the
  // accumulator loop, residual after the barrier) their latency was exposed once per tile: measured on 128 -> 128 at 256x256,
  // bias +8 %, bias + residual +15 % over the plain kernel (profiles/r2i_epilogue_micro.txt).
  const int sl = tid % SPRW, co = c0 + sl * 8;
  // (Applying bias / ReLU to the ACCUMULATORS — one fma, the lane's bias quad a broadcast 16-byte load — and making the second phase a
  // pure 16-byte copy for tiles without residual / mask / GroupNorm sums was built and measured in round 3: the nine-tap 128-row
  // kernel +2-3 % on binary16, the patch-staged 256 x 256 tile -5 %, the step -1.3 % (profiles/r3q_*); removed, it last existed in
  // commit "conv epilogue: simple tiles ...".  SQ counters of the 256 x 256 tile: ~1700 VALU + ~1100 SALU instructions per thread in
  // this epilogue, 17 % of that kernel's time with nothing to hide under: profiles/r3p_p9_sq.txt.)
  float b8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) b8[e] = 0.f;
  if (bias) {
    if (co + 8 <= p.d.Cout_w) {                    // two 16-byte loads (parameters are slices of a flat buffer: 4-byte alignment only)
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      const f4u lo = *(const f4u*)(bias + co), hi = *(const f4u*)(bias + co + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { b8[e] = lo[e]; b8[4 + e] = hi[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (co + e < p.d.Cout_w) b8[e] = bias[co + e];
    }
  }
  // fused GroupNorm-backward sums: this thread's eight channels' gamma / beta and their groups' statistics of the tile's image
  float fga[8], fbe[8], fmu[8], frs[8], fs1[8], fs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { fga[e] = fbe[e] = fmu[e] = frs[e] = 0.f; fs1[e] = fs2[e] = 0.f; }
  if (VQ_GNB(p)) {                                 // block-uniform
    const int n_img = (p0 / BP) / (p.HoWo / BP), cg = p.d.Cout / p.gnb_G;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (co + e < p.d.Cout) {
        const int g = (co + e) / cg;
        fga[e] = p.gnb_gamma[co + e]; fbe[e] = p.gnb_beta[co + e];
        fmu[e] = p.gnb_mean[n_img * p.gnb_G + g]; frs[e] = p.gnb_rstd[n_img * p.gnb_G + g];
      }
  }
  // pixel p_l of the tile -> output pixel m: consecutive pixels, or (nine-tap kernel) a 16-wide patch of one image
  const bool pt = p.pt_tpi > 0;
  int mbase = p0;
  if (mbase_in >= 0) mbase = mbase_in;             // the patch kernels located their patch at their start: no divisions here
  else if (pt) {
    const int ptile = p0 / BP, n = ptile / p.pt_tpi, rem = ptile - n * p.pt_tpi, tyi = rem / p.pt_tx;
    mbase = (n * p.d.Ho + tyi * (BP / 16)) * p.d.Wo + (rem - tyi * p.pt_tx) * 16;
  }
  // Item k of a thread (k = 0 .. ITEMS-1) is pixel p_l = tid / SPRW + k * PSTEP of the tile, always in the thread's own 8-channel slot.
  // PSTEP is a multiple of 16, so both pixel maps advance by a CONSTANT number of output pixels per item (linear: PSTEP; patch:
  // PSTEP / 16 image rows) and the plain NHWC offset by a constant number of elements: one 64-bit add per item instead of the
  // multiply chain per item that rounds 1-2 ran (depth-to-space outputs keep the per-item form).
  constexpr int PSTEP = NT / SPRW;
  static_assert(PSTEP % 16 == 0, "items of a thread are whole patch rows apart");
  const int pl0 = tid / SPRW;
  const int m0 = mbase + (pt ? (pl0 >> 4) * p.d.Wo + (pl0 & 15) : pl0);
  const int mstep = pt ? (PSTEP / 16) * p.d.Wo : PSTEP;
  const bool plain = p.d2s == 0;                   // block-uniform
  const int64_t off0 = (int64_t)m0 * p.d.Cout + co, ostep = (int64_t)mstep * p.d.Cout;
  typename St::Raw rraw[2][U], mraw[2][U];
  int64_t off[2][U];
  bool live[2][U];
  // depth-to-space stores (sub-pixel Upsample, patch-conv data gradients): the (tap, channel) part of the address is a constant
  // of the thread (its 8-channel slot never changes), the pixel part needs no division when the extents are powers of two.
  // conv_out_offset() per item was five 32-bit divisions = ~175 VALU instructions x 16 items per thread: a third of the
  // sub-pixel forward kernel (profiles/r2zz: 840 -> 564 us without the epilogue, 776 without its stores).
  int64_t d2s_add = 0;
  if (p.d2s) {
    const int tap = co / p.d2s_c, ci = co - tap * p.d2s_c, r = tap / p.d2s, s2 = tap - r * p.d2s;
    d2s_add = ((int64_t)r * (p.d.Wo * p.d2s) + s2) * p.d2s_c + ci;
  }
  auto d2s_offset = [&](int m) -> int64_t {
    int n, oy, ox;
    if (p.pix_hwsh >= 0) { n = m >> p.pix_hwsh; const int rem = m & (p.HoWo - 1); oy = rem >> p.pix_wsh; ox = rem & (p.d.Wo - 1); }
    else { n = m / p.HoWo; const int rem = m - n * p.HoWo; oy = rem / p.d.Wo; ox = rem - oy * p.d.Wo; }
    return ((int64_t)((n * p.d.Ho + oy) * p.d2s) * (p.d.Wo * p.d2s) + ox * p.d2s) * p.d2s_c + d2s_add;
  };
  // The read-once operands of a round are requested for DEAD items too (at element 0): no branch between the requests, so they
  // and the LDS reads of a round are in flight together; only the final store is predicated.
  auto request = [&](int round, int slot) {        // both compile-time after unrolling
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = round * U + u;
      const int m = m0 + k * mstep;
      live[slot][u] = m < p.M && co < p.d.Cout;
      int64_t o;
      if (plain) o = off0 + k * ostep;             // (block-uniform branch)
      else o = d2s_offset(live[slot][u] ? m : 0);
      off[slot][u] = live[slot][u] ? o : 0;
      if (VQ_SKIP_EPI(p) == 5) {                     // A/B candidate: the read-once operands as streaming loads too
        if (p.residual) St::load8_raw_nt(rraw[slot][u], p.residual, off[slot][u]);
        if (p.relu_mask) St::load8_raw_nt(mraw[slot][u], p.relu_mask, off[slot][u]);
      } else {
        if (p.residual) St::load8_raw(rraw[slot][u], p.residual, off[slot][u]);
        if (p.relu_mask) St::load8_raw(mraw[slot][u], p.relu_mask, off[slot][u]);
      }
    }
  };
  request(0, 0);
  VQ_STAMP(3);
  // (a 16-byte form of these writes — v_permlane32_swap of the packed accumulators, two ds_write_b128 per fragment — was measured
  // equal and removed: profiles/r2w_epilogue_swap_*; commit 2da2460)
  // (Round 3: all packed pairs and LDS addresses of a row block pinned in registers of their own before its eight writes — hipcc reuses
  // ONE pair for every group, so each group's first multiply has a write-after-read on the previous ds_write's operands: bf16 phase
  // 2.7k -> 1.7k cycles per row block on the 256 x 256 tile, binary16 3.0k -> 3.4k, step +-0: profiles/r3aa_*; not kept.)
  auto transpose_out = [&](auto unit_tag) {        // unit_tag: alpha == 1 (bf16 storage, no weight scale): nothing to multiply by
#pragma unroll
    for (int a = 0; a < FC; ++a) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co_l = wc0 + a * 32 + q * 8 + fh * 4;
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          const int p_l = wp0 + b * 32 + (PERM ? tap9_perm(fr) : fr);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = decltype(unit_tag)::value ? acc[a][b][q * 4 + e] : acc[a][b][q * 4 + e] * alpha;
          if constexpr (DT == VQ_F16) { if (count_range) rng_or |= __float_as_uint(acc[a][b][q * 4]); }
          if (VQ_SKIP_EPI(p) != 3 || v[0] == 123456.789f)    // measurement (3): no LDS transposition writes
            St::store4(ot, p_l * BC + (((co_l >> 3) ^ (p_l & (SPRW - 1))) << 3) + (co_l & 4), v);
        }
      }
      VQ_STAMP(30 + a);
    }
  };
  if (alpha == 1.f) transpose_out(std::true_type{});   // (block-uniform)
  else transpose_out(std::false_type{});
  VQ_STAMP(4);
  __syncthreads();
  VQ_STAMP(5);
  // GroupNorm partials of this thread's slot, channels 0-3 | 4-7: (sum, sum of squares) of v - pivot, the pivot being the first value of
  // the lane that starts the group in this wave (gn_silu.hip's header: shifted moments, never E[x^2] - E[x]^2 of raw fp32 sums)
  float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gpiv[2] = {0.f, 0.f};
  float fuse_c[16];                                // (pricing knob 6 only: per-channel sums of the fused GroupNorm backward)
#pragma unroll
  for (int e = 0; e < 16; ++e) fuse_c[e] = 0.f;
  // The common case — a whole tile, plain NHWC output, no residual / mask / GroupNorm sums — as straight-line code: no per-item
  // uniform branches (each `if (p.residual)`, `if (p.d.relu)`, `if (live)` ... of the general loop is a taken or not-taken branch
  // per ITEM: ~100 branches and ~1100 scalar instructions per thread on the 256 x 256 tile, profiles/r3p_p9_sq.txt); ReLU is a
  // max against 0 or -inf.  (The same straight-line form instantiated for GroupNorm sums / residual / ReLU mask too: +-0 in the step,
  // profiles/r3s_bench_ab.txt; not kept.)
  const bool whole = (pt || p0 + BP <= p.M) && c0 + BC <= p.d.Cout;
  if (whole && plain && !p.residual && !p.relu_mask && !p.gn_part && VQ_SKIP_EPI(p) == 0) {      // (block-uniform)
    const float floor_ = p.d.relu ? 0.f : -__builtin_inff();
    // NOT unrolled over the rounds: +0.25 % on the step, 4 KB less code per kernel (profiles/r3t_bench_ab.txt).  (The idea behind it —
    // once-per-tile code running from a cold instruction cache — did not survive its own test: 128 KiB of straight-line VALU code
    // in a loop runs within 4 % of 4 KiB, tools/micro/icache.hip, profiles/r3u_icache.txt.)
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
      float v[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p_l = pl0 + (r * U + u) * PSTEP;
        St::load8(ot, p_l * BC + ((sl ^ (p_l & (SPRW - 1))) << 3), v[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t o = off0 + (r * U + u) * ostep;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = fmaxf(v[u][e] + b8[e], floor_);
        if constexpr (DT == VQ_F16) {
          vq_u32x4 q;
          q.x = St::pack2(v[u][0], v[u][1]); q.y = St::pack2(v[u][2], v[u][3]); q.z = St::pack2(v[u][4], v[u][5]); q.w = St::pack2(v[u][6], v[u][7]);
          if (count_range) rng_pk = vq_pkmax16(vq_pkmax16(rng_pk, q.x & 0x7fff7fffu, q.y & 0x7fff7fffu), q.z & 0x7fff7fffu, q.w & 0x7fff7fffu);
          vq_store16_nt((vq_f16*)p.y + o, q);
        } else St::store8_nt(p.y, o, v[u]);
      }
      VQ_STAMP(40 + r);
    }
  } else
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {               // U items per round; the next round's global reads are in flight under this one
    if (r + 1 < ROUNDS) request(r + 1, (r + 1) & 1);
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {                  // all LDS reads of the round first
      const int p_l = pl0 + (r * U + u) * PSTEP;
      St::load8(ot, p_l * BC + ((sl ^ (p_l & (SPRW - 1))) << 3), v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] += b8[e];
      if (p.residual && VQ_SKIP_EPI(p) == 6) {
        // PRICING ONLY (ablate builds, hint 8197): what folding the GroupNorm BACKWARD sums into the data-gradient conv that produces
        // dy would add to this epilogue (the round-2 verdict's "5 -> 4 passes").  The residual operand stands for the GroupNorm input
        // x; per element the normalised value, the affine output, silu'(.) and the four sums (group: sum gamma dg, sum gamma dg xh;
        // channel: sum dg, sum dg xh) with stand-in statistics from the bias slot; the sums leave through the GroupNorm-partial rows.
        float rv[8];
        St::unpack8(rraw[r & 1][u], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (rv[e] - b8[e]) * 1.25f, yv = xh * (1.f + b8[e]) + b8[e], sg = vq_sigmoid(yv);
          const float dg = v[u][e] * (sg * (1.f + yv * (1.f - sg)));
          fuse_c[e] += dg; fuse_c[8 + e] += dg * xh;
          gsum[e < 4 ? 0 : 2] += (1.f + b8[e]) * dg; gsum[e < 4 ? 1 : 3] += (1.f + b8[e]) * dg * xh;
        }
      } else if (VQ_GNB(p)) {
        // dy = this conv's output v; x = the GroupNorm's input (in the residual slot): per channel sum dg and sum dg * xhat, exactly
        // what gn_reduce_kernel<DT, 1, SILU> forms from the stored tensors (here from the fp32 values, before dy is rounded)
        float rv[8];
        St::unpack8(rraw[r & 1][u], rv);
        if (live[r & 1][u]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (rv[e] - fmu[e]) * frs[e];
            float dg = v[u][e];
            if (p.gnb_silu) {
              const float yv = xh * fga[e] + fbe[e], sg = vq_sigmoid(yv);
              dg *= sg * (1.f + yv * (1.f - sg));
            }
            fs1[e] += dg; fs2[e] += dg * xh;
          }
        }
      } else if (p.residual) {
        float rv[8];
        St::unpack8(rraw[r & 1][u], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] += rv[e];
      }
      if (p.d.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = v[u][e] > 0.f ? v[u][e] : 0.f;
      }
      if (p.relu_mask) {
        float mv[8];
        St::unpack8(mraw[r & 1][u], mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = mv[e] > 0.f ? v[u][e] : 0.f;
      }
      if (p.gn_part && r == 0 && u == 0) {           // (block-uniform; every lane is live with the partials on: gn_tile_impl)
        const int spg = p.gn_cg >= 8 ? p.gn_cg >> 3 : 1, src = (lane % SPRW) & ~(spg - 1);
        gpiv[0] = __shfl(v[0][0], src);
        gpiv[1] = p.gn_cg == 4 ? __shfl(v[0][4], src) : gpiv[0];
      }
      if (live[r & 1][u]) {
        // streaming store: the output is not touched again by this kernel, and written through L2 in the ordinary way it evicted
        // the weight / halo lines the main loop keeps re-reading (measured: +13 % / +8 % on 128 ch @256^2, +2.5 % on the 256 tile)
        if (VQ_SKIP_EPI(p) == 2) { if (v[u][0] == 123456.789f) St::store8(p.y, off[r & 1][u], v[u]); }    // measurement: no store
        else if (VQ_SKIP_EPI(p) == 4) St::store8(p.y, off[r & 1][u], v[u]);                             // A/B: ordinary stores
        else if constexpr (DT == VQ_F16) {
          vq_u32x4 q;
          q.x = St::pack2(v[u][0], v[u][1]); q.y = St::pack2(v[u][2], v[u][3]); q.z = St::pack2(v[u][4], v[u][5]); q.w = St::pack2(v[u][6], v[u][7]);
          if (count_range) rng_pk = vq_pkmax16(vq_pkmax16(rng_pk, q.x & 0x7fff7fffu, q.y & 0x7fff7fffu), q.z & 0x7fff7fffu, q.w & 0x7fff7fffu);
          vq_store16_nt((vq_f16*)p.y + off[r & 1][u], q);
        } else St::store8_nt(p.y, off[r & 1][u], v[u]);
        if (p.gn_part) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dv = v[u][e] - gpiv[0]; gsum[0] += dv; gsum[1] += dv * dv; }
#pragma unroll
          for (int e = 4; e < 8; ++e) { const float dv = v[u][e] - gpiv[1]; gsum[2] += dv; gsum[3] += dv * dv; }
        }
      }
    }
    VQ_STAMP(40 + r);
  }
  VQ_STAMP(6);
  if (VQ_SKIP_EPI(p) == 6) {                       // the sixteen per-channel sums through the same wave butterfly as the group sums
#pragma unroll
    for (int m = SPRW; m < 64; m <<= 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) fuse_c[e] += __shfl_xor(fuse_c[e], m);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) gsum[e & 3] += 1e-30f * fuse_c[e];
  }
  if constexpr (DT == VQ_F16) { if (count_range) vq_range_events16(p.range_events, rng_pk, rng_or); }
  if (VQ_GNB(p)) {                                 // block-uniform: one row per wave, like the GroupNorm statistics below
    static_assert(64 % SPRW == 0, "slot <-> lane map");
#pragma unroll
    for (int m = SPRW; m < 64; m <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { fs1[e] += __shfl_xor(fs1[e], m); fs2[e] += __shfl_xor(fs2[e], m); }
    }
    // (every kernel the dispatcher reaches WITH the sums on has 32 pixels per wave: the 64 x 64 short-M tile is switched off for it,
    // hint-forced experimental tiles are refused by vq_conv2d_fwd)
    const int wave = tid >> 6, tpi = p.HoWo / BP, tile_lin = p0 / BP, n_img = tile_lin / tpi, tile = tile_lin - n_img * tpi;
    float* row = p.gnb_part + (((int64_t)n_img * p.gnb_rows + tile * NW + wave) * p.d.Cout + co) * 2;
    if (lane < SPRW && co < p.d.Cout) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { row[e * 2] = fs1[e]; row[e * 2 + 1] = fs2[e]; }
    }
  }
  if (p.gn_part) {                                 // block-uniform
    // One partial row per WAVE, no LDS and no barrier: lanes sl + SPRW * j of a wave hold the same 8-channel slot (NT and 64 are
    // multiples of SPRW), a fixed butterfly over j leaves the wave's totals of that slot in every lane; groups wider than a slot
    // (16 / 32 channels) are adjacent slots = adjacent lanes.  (The first version reduced the whole block through LDS behind two
    // extra barriers: ~8 % of the 128-channel 256x256 layers, profiles/r2f_conv_table_c3_ref.txt.)
    static_assert(64 % SPRW == 0, "slot <-> lane map");
#pragma unroll
    for (int m = SPRW; m < 64; m <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) gsum[e] += __shfl_xor(gsum[e], m);
    }
    const int cg = p.gn_cg, wave = tid >> 6;
    const int tile_lin = p0 / BP;                  // pixel tile index over the whole batch (patch tiles and linear tiles alike)
    const int n = tile_lin / p.gn_tiles, tile = tile_lin - n * p.gn_tiles;
    float* row = p.gn_part + (((int64_t)n * p.gn_tiles + tile) * NW + wave) * p.gn_G * 2;
    // the row leaves as (mean, M2 = sum (v - mean)^2) per group; BP / NW pixels x cg channels each (a power of two: exact reciprocal).
    // The writing lane is the one whose first value was the pivot.
    const float icnt = 1.f / (float)((BP / NW) * cg);
    if (cg == 4) {
      const int g = (c0 + sl * 8) >> 2;
      if (lane < SPRW && g < p.gn_G) {
        row[g * 2] = gpiv[0] + gsum[0] * icnt; row[g * 2 + 1] = gsum[1] - gsum[0] * (gsum[0] * icnt);
        if (g + 1 < p.gn_G) { row[g * 2 + 2] = gpiv[1] + gsum[2] * icnt; row[g * 2 + 3] = gsum[3] - gsum[2] * (gsum[2] * icnt); }
      }
    } else {
      float a = gsum[0] + gsum[2], b = gsum[1] + gsum[3];
      const int spg = cg >> 3;                     // slots per group: 1, 2 or 4
      for (int m = 1; m < spg; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
      const int g = (c0 + sl * 8) / cg;
      if (lane < SPRW && (sl & (spg - 1)) == 0 && g < p.gn_G) { row[g * 2] = gpiv[0] + a * icnt; row[g * 2 + 1] = b - a * (a * icnt); }
    }
  }
}

// ---- epilogue of the VQ_F16X2 kernels ----------------------------------------------------------------------------
// Same two phases as igemm_epilogue, with the tile crossing the LDS as FP32 (the 22 bits a stored value keeps must not be lost in a
// binary16 transposition): the accumulators (times alpha) of a CB-column slice of the tile go to LDS as [pixel][CB] floats — 16-byte
// units XOR-swizzled by the pixel row — and come back so that a lane owns 8 consecutive channels of one pixel: + bias, + residual,
// ReLU / ReLU mask, GroupNorm partial sums, hi / lo split, two adjacent 16-byte streaming stores (32 contiguous bytes per lane).
// The slice width CB (32 / 64 / 128 columns) is what BP x CB floats fit into the LDS the main loop had: BC / CB passes per tile.
// Precondition: every wave of the block is past the last barrier of the main loop.
template <int BC, int BP, int WC, int WP, int PERM>
__device__ __forceinline__ void igemm_epilogue_x2(const ConvParams& p, vq_bf16* lds, f32x16 (&acc)[WC / 32][WP / 32], int c0, int p0,
                                                  int wc0, int wp0, float alpha_in, int mbase_in) {
  constexpr int FC = WC / 32, FP = WP / 32, NW = (BC / WC) * (BP / WP), NT = NW * 64;
  constexpr int CB = BC >= 256 ? 128 : (BC >= 64 ? 64 : 32), NPASS = BC / CB;
  constexpr int SPRW = CB / 8;                     // 8-channel slots per pixel row of a pass
  constexpr int UPR = CB / 4;                      // 16-byte units per row
  constexpr int ITEMS = BP * SPRW / NT, PSTEP = NT / SPRW;
  static_assert(ITEMS * NT == BP * SPRW && NT % SPRW == 0 && 64 % SPRW == 0, "tile / thread-count mismatch");
  static_assert(PSTEP % 16 == 0 || ITEMS == 1, "items of a thread are whole patch rows apart");
  typedef Store<VQ_F16X2> St;
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  const float alpha = __float_as_uint(alpha_in) == 0x80000000u ? conv_alpha(p) : p.alpha * alpha_in;
  float* ot = (float*)lds;
  const bool count_range = p.range_events != nullptr;
  unsigned rng = 0u;
  const float* bias = p.bias;                      // sub-pixel conv: the Cout/4 bias entries serve all four phase blocks
  if (bias && p.sub) bias -= (c0 / p.d2s_c) * p.d2s_c;
  const bool pt = p.pt_tpi > 0;
  int mbase = p0;
  if (mbase_in >= 0) mbase = mbase_in;
  else if (pt) {
    const int ptile = p0 / BP, n = ptile / p.pt_tpi, rem = ptile - n * p.pt_tpi, tyi = rem / p.pt_tx;
    mbase = (n * p.d.Ho + tyi * (BP / 16)) * p.d.Wo + (rem - tyi * p.pt_tx) * 16;
  }
  const int sl = tid % SPRW, pl0 = tid / SPRW;
  const int m0 = mbase + (pt ? (pl0 >> 4) * p.d.Wo + (pl0 & 15) : pl0);
  const int mstep = pt ? (PSTEP / 16) * p.d.Wo : PSTEP;
  const bool plain = p.d2s == 0;
#pragma unroll 1
  for (int h = 0; h < NPASS; ++h) {
    const int co = c0 + h * CB + sl * 8;
    float b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = (bias && co + e < p.d.Cout_w) ? bias[co + e] : 0.f;
    // ---- phase 1: the fragments whose 32 columns lie in this slice
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      if ((wc0 + a * 32) / CB != h) continue;          // (wave-uniform)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = wc0 + a * 32 + q * 8 + fh * 4 - h * CB;
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          const int p_l = wp0 + b * 32 + (PERM ? tap9_perm(fr) : fr);
          vq_f4 v;
          v.x = acc[a][b][q * 4] * alpha; v.y = acc[a][b][q * 4 + 1] * alpha; v.z = acc[a][b][q * 4 + 2] * alpha; v.w = acc[a][b][q * 4 + 3] * alpha;
          *(vq_f4*)(ot + p_l * CB + (((cl >> 2) ^ (p_l & (UPR - 1))) << 2)) = v;
        }
      }
    }
    __syncthreads();
    // ---- phase 2
    int64_t d2s_add = 0;
    if (p.d2s) {
      const int tap = co / p.d2s_c, ci = co - tap * p.d2s_c, r = tap / p.d2s, s2 = tap - r * p.d2s;
      d2s_add = ((int64_t)r * (p.d.Wo * p.d2s) + s2) * p.d2s_c + ci;
    }
    float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gpiv[2] = {0.f, 0.f};     // (shifted moments: see igemm_epilogue)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int p_l = pl0 + k * PSTEP, m = m0 + k * mstep;
      const bool live = m < p.M && co < p.d.Cout;
      const int key = p_l & (UPR - 1);
      const vq_f4 lo4 = *(const vq_f4*)(ot + p_l * CB + (((2 * sl) ^ key) << 2));
      const vq_f4 hi4 = *(const vq_f4*)(ot + p_l * CB + (((2 * sl + 1) ^ key) << 2));
      if (!live) continue;
      int64_t off;
      if (plain) off = (int64_t)m * p.d.Cout + co;
      else {
        int n, oy, ox;
        if (p.pix_hwsh >= 0) { n = m >> p.pix_hwsh; const int rem = m & (p.HoWo - 1); oy = rem >> p.pix_wsh; ox = rem & (p.d.Wo - 1); }
        else { n = m / p.HoWo; const int rem = m - n * p.HoWo; oy = rem / p.d.Wo; ox = rem - oy * p.d.Wo; }
        off = ((int64_t)((n * p.d.Ho + oy) * p.d2s) * (p.d.Wo * p.d2s) + ox * p.d2s) * p.d2s_c + d2s_add;
      }
      float v[8] = {lo4.x + b8[0], lo4.y + b8[1], lo4.z + b8[2], lo4.w + b8[3], hi4.x + b8[4], hi4.y + b8[5], hi4.z + b8[6], hi4.w + b8[7]};
      if (p.residual) {
        float rv[8];
        St::load8(p.residual, off, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.d.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (p.relu_mask) {
        float mv[8];
        St::load8(p.relu_mask, off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      if (count_range) rng = vq_absmax_bits(rng, v);
      St::store8_nt(p.y, off, v);
      if (p.gn_part) {                               // (block-uniform, and every lane is live with the partials on: gn_tile_impl)
        if (k == 0) {
          const int spg = p.gn_cg >= 8 ? p.gn_cg >> 3 : 1, src = (lane % SPRW) & ~(spg - 1);
          gpiv[0] = __shfl(v[0], src);
          gpiv[1] = p.gn_cg == 4 ? __shfl(v[4], src) : gpiv[0];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dv = v[e] - gpiv[0]; gsum[0] += dv; gsum[1] += dv * dv; }
#pragma unroll
        for (int e = 4; e < 8; ++e) { const float dv = v[e] - gpiv[1]; gsum[2] += dv; gsum[3] += dv * dv; }
      }
    }
    if (p.gn_part) {                                 // block-uniform: one partial row per wave (see igemm_epilogue), this slice's groups
#pragma unroll
      for (int m = SPRW; m < 64; m <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gsum[e] += __shfl_xor(gsum[e], m);
      }
      const int cg = p.gn_cg, wave = tid >> 6;
      const int tile_lin = p0 / BP;
      const int n = tile_lin / p.gn_tiles, tile = tile_lin - n * p.gn_tiles;
      float* row = p.gn_part + (((int64_t)n * p.gn_tiles + tile) * NW + wave) * p.gn_G * 2;
      const float icnt = 1.f / (float)((BP / NW) * cg);
      if (cg == 4) {
        const int g = co >> 2;
        if (lane < SPRW && g < p.gn_G) {
          row[g * 2] = gpiv[0] + gsum[0] * icnt; row[g * 2 + 1] = gsum[1] - gsum[0] * (gsum[0] * icnt);
          if (g + 1 < p.gn_G) { row[g * 2 + 2] = gpiv[1] + gsum[2] * icnt; row[g * 2 + 3] = gsum[3] - gsum[2] * (gsum[2] * icnt); }
        }
      } else {
        float a = gsum[0] + gsum[2], b = gsum[1] + gsum[3];
        const int spg = cg >> 3;                     // slots per group: 1, 2 or 4
        for (int m = 1; m < spg; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        const int g = co / cg;
        if (lane < SPRW && (sl & (spg - 1)) == 0 && g < p.gn_G) { row[g * 2] = gpiv[0] + a * icnt; row[g * 2 + 1] = b - a * (a * icnt); }
      }
    }
    if (h + 1 < NPASS) __syncthreads();              // the slab is rewritten by the next slice
  }
  if (count_range) vq_range_events(p.range_events, rng, rng);
}

// Two-buffer LDS pipeline over 64-wide K chunks.  The tile DMAs of chunk c+1 are spread over the k-steps of
// chunk c (a quarter of the 1-KiB pieces after each k-step's MFMAs) instead of being issued as one burst.
// WREG = 1: the weight operand never touches LDS.  Weights are packed in MFMA-fragment order (layout 1 of
// pack_weight_kernel) and every wave fetches its own a-fragments with coalesced 16-B-per-lane global loads one
// chunk ahead (the load for chunk c+1, k-step kk is issued right after the MFMAs of chunk c, k-step kk
// released that register).  Measured (profiles/r1_conv_tile_ab.txt): it pays (+5..10 %) only when no two waves
// of a block share weight rows (WC = 32: the 128x128 tile as 4 waves x 32c x 128p, and the Cout <= 64 tiles);
// with shared rows the redundant L2->register traffic costs more than the LDS traffic it saves, so the
// 256x256 tile keeps its weights in LDS.
template <int DT, int BC, int BP, int WC, int WP, int WREG, int DBG = 0, int PP = 0>
__global__ __launch_bounds__((BC / WC) * (BP / WP) * 64) void conv_igemm_glds_kernel(const ConvParams p) {
  constexpr int BK = 64;
  constexpr bool X2 = DT == VQ_F16X2;             // K counts virtual channels (p.d.Cin doubled by vq_conv2d_fwd), three products per pair
  constexpr int XOFF = WREG ? 0 : BC;             // first row of the pixel tile inside a buffer
  constexpr int TILE = (XOFF + BP) * BK;          // bf16 elements per buffer
  constexpr int FC = WC / 32, FP = WP / 32;
  constexpr int NWP = BP / WP;
  constexpr int NW = (BC / WC) * (BP / WP);       // waves per block (4 or 8)
  constexpr int NA = BP / 8 / NW, NB = WREG ? 0 : BC / 8 / NW;   // 1-KiB DMA pieces (8 rows) per wave per chunk
  static_assert(NA >= 1 && NA * NW * 8 == BP && (WREG || NB * NW * 8 == BC), "tile / wave mismatch");

  VQ_DYN_LDS(vq_bf16, lds);                       // 2 * TILE elements, all LDS in ONE array

  const int tid = threadIdx.x;
  const float alpha_raw = conv_alpha_request(p);   // (consumed after the first tile wait: conv_alpha_finish)
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
  const int sub_ph = p.sub ? c0 / p.d2s_c : 0, sub_a = sub_ph >> 1, sub_b = sub_ph & 1;   // block-uniform phase (sub-pixel conv)

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
      xn[i] = n; xby[i] = oy * p.d.stride - p.d.pad_t + sub_a; xbx[i] = ox * p.d.stride - p.d.pad_l + sub_b;
    } else {
      xn[i] = -1; xby[i] = 0; xbx[i] = 0;
    }
  }
  const vq_bf16* pb[NB > 0 ? NB : 1];
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

  // DBG (compile-time, 0 in the product; ABLATE builds only): 1 = no DMA after the first chunk, 2 = no MFMA,
  // 4 = weights neither staged nor read from LDS
  auto stage = [&](int buf) {   // whole chunk in one burst: prologue only
    vq_bf16* base = lds + buf * TILE;
    if (cit == 0) set_tap();
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if constexpr (!(DBG & 4)) glds16(pb[i], base + (wave * NB + i) * 8 * BK);
      pb[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      glds16(pa[i], base + (XOFF + (wave * NA + i) * 8) * BK);
      pa[i] += inca[i];
    }
    if (++cit == cpt) {
      cit = 0;
      if (++tap_s == p.d.S) { tap_s = 0; ++tap_r; }
    }
  };

  // the same work as stage(buf), cut in BK/16 parts: part q issues pieces [q*PPQ, (q+1)*PPQ) of the NB + NA list
  constexpr int NPIECE = NA + NB, PPQ = (NPIECE + BK / 16 - 1) / (BK / 16);
  auto stage_part = [&](int buf, int q) {
    vq_bf16* base = lds + buf * TILE;
    if (q == 0 && cit == 0) set_tap();
#pragma unroll
    for (int j = q * PPQ; j < (q + 1) * PPQ && j < NPIECE; ++j) {
      if (j < NB) {
        const int i = j;
        if constexpr (!(DBG & 5)) glds16(pb[i], base + (wave * NB + i) * 8 * BK);
        pb[i] += BK;
      } else {
        const int i = j - NB;
        if constexpr (!(DBG & 1)) glds16(pa[i], base + (XOFF + (wave * NA + i) * 8) * BK);
        pa[i] += inca[i];
      }
    }
    if (q == BK / 16 - 1) {
      if (++cit == cpt) {
        cit = 0;
        if (++tap_s == p.d.S) { tap_s = 0; ++tap_r; }
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

  const int fr = lane & 31, fh = lane >> 5;
  // Software-pipelined fragment schedule: the ds_read_b128s of k-step kk+1 are issued BEFORE the MFMAs of
  // k-step kk (fenced so hipcc keeps that order); the MFMA block then needs only a counted lgkmcnt wait and
  // its 4 x 32 cycles cover the LDS latency of the next fragments.  frag_load(buf, 0) of a chunk is issued
  // right after the barrier, ahead of the next chunk's address math + DMA issue.
  constexpr int NS = X2 ? 4 : 2;                  // fragment slots (VQ_F16X2: a k-step pair's hi and lo fragments are live together)
  s16x8 af[NS][FC], bfr[NS][FP];
  if constexpr ((DBG & 4) != 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) af[q][a][e] = (short)(0x3F80 + lane);
  }
  auto frag_load = [&](int buf, int kk, int slot) {
    const vq_bf16* base = lds + buf * TILE;
#pragma unroll
    for (int a = 0; a < FC; ++a)
      if constexpr (!(DBG & 4) && !WREG) af[slot][a] = *(const s16x8*)(base + Swz<BK>::elem(wc0 + a * 32 + fr, frag_slot<X2>(kk, fh)));
#pragma unroll
    for (int b = 0; b < FP; ++b) bfr[slot][b] = *(const s16x8*)(base + Swz<BK>::elem(XOFF + wp0 + b * 32 + fr, frag_slot<X2>(kk, fh)));
  };
  // weight fragments in registers (WREG): wf[kk][a] holds k-step kk of the chunk about to be / being computed
  s16x8 wf[BK / 16][FC];
  const vq_bf16* wptr[FC];
  if constexpr (WREG) {
    const int ncb = (p.d.Cout + 31) >> 5;
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      int cb = ((c0 + wc0) >> 5) + a;
      if (cb >= ncb) cb = ncb - 1;
      wptr[a] = p.w + ((int64_t)cb * (p.Kp >> 4)) * 512 + lane * 8;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int a = 0; a < FC; ++a) wf[kk][a] = *(const s16x8*)(wptr[a] + kk * 512);
  }
  constexpr int WL = WREG ? (BK / 16) * FC : 0;   // weight-fragment loads a wave issues per chunk
  auto compute = [&](int buf, bool more, int nbuf) {   // expects frag_load(buf, 0, 0) to have been issued
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      if (kk + 1 < BK / 16) frag_load(buf, kk + 1, (kk + 1) % NS);
      vq_sched_fence();
      if constexpr (X2) {
        // kk even: hi x hi;  kk odd: hi x lo and lo x hi of the same pair (the pair's hi fragments are still in their slots / registers)
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
          for (int b = 0; b < FP; ++b) {
            const int kh = kk & ~1, kl = kk | 1;   // the pair's hi / lo k-steps
            if (kk & 1) {
              acc[a][b] = mfma16<DT>(WREG ? wf[kh][a] : af[kh][a], bfr[kl][b], acc[a][b]);
              acc[a][b] = mfma16<DT>(WREG ? wf[kl][a] : af[kl][a], bfr[kh][b], acc[a][b]);
            } else acc[a][b] = mfma16<DT>(WREG ? wf[kh][a] : af[kh][a], bfr[kh][b], acc[a][b]);
          }
        vq_sched_fence();
        if (more) stage_part(nbuf, kk);
        if constexpr (WREG) {
          if (more && (kk & 1)) {   // both registers of the pair are free again
#pragma unroll
            for (int a = 0; a < FC; ++a) {
              wf[kk & ~1][a] = *(const s16x8*)(wptr[a] + (BK / 16 + (kk & ~1)) * 512);
              wf[kk | 1][a] = *(const s16x8*)(wptr[a] + (BK / 16 + (kk | 1)) * 512);
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (!(DBG & 2) && WREG) acc[a][b] = mfma16<DT>(wf[kk][a], bfr[kk & 1][b], acc[a][b]);
          else if constexpr (!(DBG & 2)) acc[a][b] = mfma16<DT>(af[kk & 1][a], bfr[kk & 1][b], acc[a][b]);
          else VQ_KEEP_ALIVE2(af[kk & 1][a], bfr[kk & 1][b]);                    // ablation: keep the reads alive
        }
      vq_sched_fence();
      if (more) stage_part(nbuf, kk);
      if constexpr (WREG) {
        if (more) {   // the registers of k-step kk are free again: refill them for the next chunk
#pragma unroll
          for (int a = 0; a < FC; ++a) wf[kk][a] = *(const s16x8*)(wptr[a] + (BK / 16 + kk) * 512);
        }
      }
    }
    if constexpr (WREG) {
#pragma unroll
      for (int a = 0; a < FC; ++a) wptr[a] += (BK / 16) * 512;
    }
  };

  // vmcnt bookkeeping (VMEM returns in order): after each k-step a wave issues a quarter of its tile DMAs and
  // then the FC weight-fragment loads of that k-step, so "at most FC outstanding" at the end of the chunk means
  // every DMA has landed while the last weight registers may still be in flight (hipcc inserts the waits for
  // those itself, before their first use).
  const int nchunks = p.RS * cpt;
  if constexpr (PP != 0) {
    // "Ping-pong" schedule of the 8-wave tiles (+12-15 % over the free-running loop below on the 256x256 tile, profiles/
    // r1_pingpong_ab_v24.txt: MFMA blocks its wave, so two uncoordinated waves per SIMD leave the matrix pipe idle
    // whenever both are in their load phase).  The two waves of a SIMD (wave w and w + 4) run
    // one barrier apart: while one executes the 16 MFMAs of a half chunk at raised priority, the other issues the
    // fragment reads of its next half chunk and, once per chunk, its whole share of the next chunk's tile DMA.
    //   slot (barrier interval):   4c      4c+1     4c+2     4c+3
    //   group 0 (waves 0-3):      MEM c.0  MMA c.0  MEM c.1  MMA c.1
    //   group 1 (waves 4-7):      MMA ..   MEM c.0  MMA c.0  MEM c.1
    // Buffer (c+1)&1 was last read in slot 4c-1 (group 1, reads retired by lgkmcnt(0) before that slot's barrier) and
    // is restaged from slot 4c (group 0) / 4c+1 (group 1); every wave retires its own DMA (vmcnt(0)) before the
    // barrier that ends its "MEM c.1" slot, i.e. before slot 4c+4 in which chunk c+1 is first read.
    static_assert(!WREG && NW == 8, "ping-pong schedule: 8-wave LDS-weight tiles");
    const int grp = wave >> 2;
    stage(0);
    wait_vmcnt<0>();
    const float alpha_s = conv_alpha_finish(p, alpha_raw);
    raw_barrier();
    if (grp == 1) raw_barrier();
    for (int c = 0; c < nchunks; ++c) {
      const bool more = (c + 1) < nchunks;
      constexpr int NPH = 2, KPP = (BK / 16) / NPH;     // phases per chunk (4 measured equal), k-steps per phase
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
#pragma unroll
        for (int kq = 0; kq < KPP; ++kq) frag_load(c & 1, KPP * ph + kq, kq);
        if (ph == 0 && more) stage((c + 1) & 1);
        wait_lgkmcnt<0>();
        if (ph == NPH - 1) wait_vmcnt<0>();
        vq_sched_fence();
        raw_barrier();
        vq_sched_fence();
        vq_setprio(1);
        if constexpr (X2) mfma_x2_pair<FC, FP>(acc, af[0], af[1], bfr[0], bfr[1]);     // (a phase = one k-step pair: KPP == 2)
        else
#pragma unroll
        for (int kq = 0; kq < KPP; ++kq)
#pragma unroll
          for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FP; ++b) acc[a][b] = mfma16<DT>(af[kq][a], bfr[kq][b], acc[a][b]);
        vq_setprio(0);
        vq_sched_fence();
        raw_barrier();
        vq_sched_fence();
      }
    }
    if (grp == 0) raw_barrier();
    if constexpr (X2) igemm_epilogue_x2<BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
    else igemm_epilogue<DT, BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
    return;
  }
  stage(0);
  wait_vmcnt<0>();
  const float alpha_s = conv_alpha_finish(p, alpha_raw);
  raw_barrier();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = (c + 1) < nchunks;
    frag_load(c & 1, 0, 0);
    vq_sched_fence();
    compute(c & 1, more, (c + 1) & 1);
    // the last k-step's weight loads may stay in flight (VQ_F16X2: the last PAIR's, issued together after the last DMA pieces)
    if constexpr (!(DBG & 8)) { if (more) wait_vmcnt<(X2 ? 2 : 1) * WL / (BK / 16)>(); else wait_vmcnt<0>(); }
    raw_barrier();
  }

  if constexpr (X2) igemm_epilogue_x2<BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
  else igemm_epilogue<DT, BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
}

// ------------------------------------------------------------------------------ three taps per staged pixel tile
// 3x3 / stride 1 / pad 1 convolutions and their data gradients with register-resident weights (see WREG above).
// The one-tap kernel stages the same 64-channel pixel tile three times per kernel row, shifted by one pixel each
// time; here it is staged ONCE with a halo — one extra column either side of every image-row segment of the tile
// (BP + 2 * segments rows of LDS) — and the three taps read it at row offsets 0 / +1 / +2, the way conv_wgrad3_kernel
// shares its X tile: a third of the LDS-DMA traffic and of the barriers per MFMA.  K order: (kernel row, 64-channel
// chunk, tap within the row); the weight fragments of the next (tap, chunk) are loaded while the current one is
// multiplied, as in the one-tap kernel.
template <int DT, int BC, int BP, int WC, int WP>
__global__ __launch_bounds__((BC / WC) * (BP / WP) * 64, 2) void conv_igemm_tap3_kernel(const ConvParams p) {
  constexpr int BK = 64;
  constexpr bool X2 = DT == VQ_F16X2;                  // (see conv_igemm_glds_kernel)
  constexpr int FC = WC / 32, FP = WP / 32;
  constexpr int NWP = BP / WP;
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr int PMAX = (BP + 2 * (BP / 16) + 7) / 8;   // 8-row DMA pieces of the largest halo tile (16-pixel segments)
  constexpr int PPW = (PMAX + NW - 1) / NW;            // pieces per wave
  constexpr int XT = PMAX * 8 * BK;                    // elements per buffer
  static_assert(PPW <= 12, "one DMA piece per (tap, k-step)");

  VQ_DYN_LDS(vq_bf16, lds);                            // 2 * XT elements (>= BP * BC for the epilogue transpose)

  const int tid = threadIdx.x;
  const float alpha_raw = conv_alpha_request(p);   // (consumed after the first tile wait: conv_alpha_finish)
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

  // image-row segments of the tile: Wo is a power of two, so a segment is min(Wo, BP) consecutive pixels of one row
  const int wsh = p.wo_shift, segsh = wsh < ilog2_ce(BP) ? wsh : ilog2_ce(BP);
  const int wseg = 1 << segsh, nslots = (BP >> segsh) * (wseg + 2);
  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;

  // ---- halo slots owned by this lane: piece (wave + NW * i), row lr of the piece, physical 16-byte slot lp -------
  // (slot -> image position is re-derived in set_row, three times per kernel, rather than kept in registers)
  const int lr = lane >> 3, lp = lane & 7;
  const int cpt = p.d.Cin >> 6;
  const vq_bf16* pa[PPW];
  int inca[PPW];
  auto set_row = [&](int kr) {                         // gather pointers of kernel row kr, channel chunk 0
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int slot = (wave + NW * i) * 8 + lr;
      const int lsa = (lp ^ ((slot >> 1) & 7)) << 3;
      const int q = slot / (wseg + 2), jj = slot - q * (wseg + 2);
      const int m = p0 + (q << segsh);
      const int n = m / p.HoWo, rem = m - n * p.HoWo;
      const int oy = rem >> wsh, ox0 = rem & (p.d.Wo - 1);
      const int ix = ox0 - 1 + jj, iy = oy - 1 + kr;
      const int ok = (int)(slot < nslots) & (int)(m < p.M) & (int)((unsigned)ix < (unsigned)Wv) & (int)((unsigned)iy < (unsigned)Hv);
      const int64_t off = (int64_t)((n * p.d.H + (iy >> p.ush)) * p.d.W + (ix >> p.ush)) * p.d.Cin + lsa;
      const uintptr_t a_ok = (uintptr_t)(xbase + off), a_zero = (uintptr_t)(zero + lsa);
      pa[i] = (const vq_bf16*)(ok ? a_ok : a_zero);
      inca[i] = ok ? BK : 0;
    }
  };
  int st_kr = 0, st_cc = 0;                            // (kernel row, chunk) the next staged buffer holds
  auto stage_piece = [&](int buf, int i) {             // i compile-time after unrolling
    if (wave + NW * i < PMAX) {
      glds16(pa[i], lds + buf * XT + (wave + NW * i) * 8 * BK);
      pa[i] += inca[i];
    }
  };
  auto stage_advance = [&]() {
    if (++st_cc == cpt) { st_cc = 0; ++st_kr; if (st_kr < 3) set_row(st_kr); }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- pixel fragments: pixel p_l of the tile, tap ks -> halo row p_l + 2 * (p_l >> segsh) + ks --------------------
  const int fr = lane & 31, fh = lane >> 5;
  int rowb[FP];
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int p_l = wp0 + b * 32 + fr;
    rowb[b] = p_l + 2 * (p_l >> segsh);
  }
  s16x8 bfr[2][FP];
  auto frag_load = [&](int buf, int ks, int kk, int slot) {
    const vq_bf16* base = lds + buf * XT;
#pragma unroll
    for (int b = 0; b < FP; ++b) {
      int row = rowb[b];
      VQ_OPAQUE_VGPR(row);                              // keeps the 12 x FP addresses out of registers (re-derived per read)
      row += ks;
      bfr[slot][b] = *(const s16x8*)(base + row * BK + ((frag_slot<X2>(kk, fh) ^ ((row >> 1) & 7)) << 3));
    }
  };

  // ---- weight fragments (fragment-order packed layout, see pack_weight_kernel layout 1) --------------------------
  s16x8 wf[BK / 16][FC];
  const vq_bf16* wrow[FC];
  {
    const int ncb = (p.d.Cout + 31) >> 5;
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      int cb = ((c0 + wc0) >> 5) + a;
      if (cb >= ncb) cb = ncb - 1;
      wrow[a] = p.w + ((int64_t)cb * (p.Kp >> 4)) * 512 + lane * 8;
    }
  }
  auto kb_of = [&](int kr, int ks, int cc) -> int { return (((kr * 3 + ks) * p.d.Cin) >> 4) + cc * (BK / 16); };
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
    for (int a = 0; a < FC; ++a) wf[kk][a] = *(const s16x8*)(wrow[a] + (int64_t)(kb_of(0, 0, 0) + kk) * 512);

  const int nsc = 3 * cpt;                             // staged buffers: (kernel row, channel chunk)
  set_row(0);
#pragma unroll
  for (int i = 0; i < PPW; ++i) stage_piece(0, i);
  stage_advance();
  wait_vmcnt<0>();
  const float alpha_s = conv_alpha_finish(p, alpha_raw);
  raw_barrier();
  int kr = 0, cc = 0;
  for (int sc = 0; sc < nsc; ++sc) {
    const int buf = sc & 1;
    const bool more_x = sc + 1 < nsc;
    frag_load(buf, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 12; ++v) {                     // v = ks * 4 + kk
      constexpr int dummy = 0; (void)dummy;
      const int ks = v >> 2, kk = v & 3;
      if (v + 1 < 12) frag_load(buf, (v + 1) >> 2, (v + 1) & 3, (v + 1) & 1);
      vq_sched_fence();
      // VQ_F16X2: the pixel fragment of an even step is the hi piece (x weights hi and lo), of an odd step the lo piece (x weights hi)
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (X2) {
            if (!(kk & 1)) acc[a][b] = mfma16<DT>(wf[kk | 1][a], bfr[v & 1][b], acc[a][b]);
            acc[a][b] = mfma16<DT>(wf[kk & ~1][a], bfr[v & 1][b], acc[a][b]);
          } else acc[a][b] = mfma16<DT>(wf[kk][a], bfr[v & 1][b], acc[a][b]);
        }
      vq_sched_fence();
      if (v < PPW && more_x) stage_piece(buf ^ 1, v);  // next buffer's DMA, one piece per step
      // refill the weight registers of this k-step for the next (tap, chunk) (VQ_F16X2: the register whose last use this step was —
      // the lo weights after the even step, the hi weights after the odd one)
      int nkr = kr, ncc = cc, nks = ks + 1;
      if (nks == 3) { nks = 0; if (++ncc == cpt) { ncc = 0; ++nkr; } }
      if (nkr < 3) {
        const int rk = X2 ? (kk ^ 1) : kk;
#pragma unroll
        for (int a = 0; a < FC; ++a) wf[rk][a] = *(const s16x8*)(wrow[a] + (int64_t)(kb_of(nkr, nks, ncc) + rk) * 512);
      }
    }
    if (more_x) stage_advance();
    if (++cc == cpt) { cc = 0; ++kr; }
    if (more_x) wait_vmcnt<FC>(); else wait_vmcnt<0>();   // the last k-step's weight loads may stay in flight
    raw_barrier();
  }
  if constexpr (X2) igemm_epilogue_x2<BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
  else igemm_epilogue<DT, BC, BP, WC, WP>(p, lds, acc, c0, p0, wc0, wp0, alpha_s);
}

// ------------------------------------------------------------------------------ nine taps per staged pixel tile
// The same idea one step further: the pixel tile is a (BP/16) x 16 PATCH of one image, staged once per 64-channel chunk with a one-pixel halo all around
// ((BP/16 + 2) x 18 rows of LDS: 180 for BP = 128), and all nine taps read it at row offsets kr * 18 + ks — 184 DMA rows
// per chunk instead of 432 (three-tap) or 1152 (one-tap), one barrier per 36 k-steps.  K order: (64-channel chunk, tap, k-step);
// the packed weights keep their tap-major layout, only the walk over them changes.
// WA bit 0: the tile DMA of the next chunk is issued from inline asm (glds16_asm), invisible to hipcc's s_waitcnt bookkeeping.
//   With the builtin, hipcc answers every wait for a weight-fragment load that has an LDS-DMA behind it with `s_waitcnt vmcnt(0)`
//   — a full drain (the DMA pieces AND the weight requests issued one k-step earlier) three times at the head of every chunk;
//   without it the weight loads get the counted waits (vmcnt(6..7)) hipcc emits in the DMA-free part of the chunk.  The wave's
//   own DMA is awaited explicitly before the chunk's barrier: it is older than the last four k-steps' weight requests.
//   (Hiding the WEIGHT loads instead was tried first and is unsafe under this kernel's register pressure: hipcc treats an asm
//   load's destination as written at once and moved an address through it — memory faults at full size, r2 notes.)
// WA bit 1: fragment addresses in registers, 32-KiB buffer stride, conflict-free lane -> pixel map (tap9_perm).
// (A three-blocks-per-CU form — adjacent buffers, 32-bit piece offsets, 168 VGPRs — was measured neutral in round 2 and removed:
//   profiles/r2x_tap9_three_blocks_*; it last existed in commit 2da2460.)
// (Round 6: TWO taps of weight fragments in flight — a register ring of 8 k-steps, slot block = tap & 1, the two blocks changing hands
// at a chunk boundary; 210 / 155 VGPRs, no scratch, bit-exact — measured +-1 % on 128 -> 128 @256^2, 256 -> 128, 128 -> 256, 64 -> 64
// @256^2 / @512^2 in bf16 and binary16, profiles/r6f_tap9_wd_ab.txt: the weight stream's latency is covered already.  Not kept.)
template <int DT, int BC, int BP, int WC, int WP, int WA = 0>
__global__ __launch_bounds__((BC / WC) * (BP / WP) * 64, 2) void conv_igemm_tap9_kernel(const ConvParams p) {
  constexpr int BK = 64;
  constexpr bool X2 = DT == VQ_F16X2;                  // (see conv_igemm_glds_kernel)
  constexpr int FC = WC / 32, FP = WP / 32;
  constexpr int NWP = BP / WP;
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr int TW = 16, TH = BP / TW, HWD = TW + 2, NSLOT = (TH + 2) * HWD;
  constexpr int PMAX = (NSLOT + 7) / 8;                // 8-row DMA pieces of the halo tile
  constexpr int PPW = (PMAX + NW - 1) / NW;            // pieces per wave
  constexpr int XT = PMAX * 8 * BK;                    // elements per buffer
  // WA = 3: second buffer at a power-of-two distance, so that (buffer, k-step) enter a fragment address by ONE xor
  constexpr bool ASMDMA = (WA & 1) != 0, REGADDR = (WA & 2) != 0;
  static_assert((WA & ~3) == 0, "WA: bit 0 = asm tile DMA, bit 1 = register fragment addresses");
  constexpr int XTS = REGADDR ? 16384 : XT;            // buffer stride in elements (32 KiB for the one-xor form)
  static_assert(PPW <= 36, "one DMA piece per (tap, k-step)");
  static_assert(!REGADDR || XT <= XTS, "halo tile larger than the 32-KiB buffer stride");

  VQ_DYN_LDS(vq_bf16, lds);                            // XTS + XT elements (>= BP * BC for the epilogue transpose)

  const int tid = threadIdx.x;
  const float alpha_raw = conv_alpha_request(p);   // (consumed after the first tile wait: conv_alpha_finish)
  VQ_STAMP(10);
  const int lane = tid & 63, wave = tid >> 6;
  const int wc0 = (wave / NWP) * WC, wp0 = (wave % NWP) * WP;
  // (2-4 consecutive tiles per block — stores draining under the next tile's first DMA — were measured: +-0 at 128 channels,
  // +5-8 % at 64, worse where blocks are scarce, and 64 more VGPRs: profiles/r2v_tap9_tiles_per_block.txt; not kept.  Round 5, on the
  // tuned kernel: two persistent blocks per CU walking contiguous tile ranges, the next tile's halo addresses + first-chunk DMA issued
  // BEFORE this tile's epilogue (into buffer 1; the epilogue transposes through buffer 0), weight registers carried over, halo and
  // fragment addresses re-derived per tile to stay at 240 VGPRs without scratch: bit-exact, and +-0 .. -2 % at 128 -> 128, 256 -> 128,
  // 64 -> 64 @256^2 in bf16 / fp16 / f16x3 (profiles/r5n_tap9_persistent_ab.txt).  With two blocks per CU the other block's k-loop
  // already covers a tile's prologue; not kept, it last existed in commit 8a73385;
  // nor do the two blocks of a CU idle the matrix pipe by running in LOCKSTEP (both in their k-loops, then both in their epilogues):
  // delaying the first wave's second block per CU by 3-10 us so that the phases alternate moves nothing beyond noise
  // (profiles/r5r_tap9_block_stagger_ab.txt; last in commit d950821).)
  const int nblk = p.n_ctiles * p.n_ptiles;
  int t;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ctile = t % p.n_ctiles, ptile = t / p.n_ctiles;
  const int c0 = ctile * BC, p0 = ptile * BP;
  const int pn = ptile / p.pt_tpi, prem = ptile - pn * p.pt_tpi, ptyi = prem / p.pt_tx;
  const int ty0 = ptyi * TH, tx0 = (prem - ptyi * p.pt_tx) * TW;    // top-left output pixel of the patch

  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;

  // ---- halo slots owned by this lane: piece (wave + NW * i), row lr of the piece, physical 16-byte slot lp -------
  const int lr = lane >> 3, lp = lane & 7;
  const int cpt = p.d.Cin >> 6;
  const vq_bf16* pa[PPW];
  int inca[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int slot = (wave + NW * i) * 8 + lr;
    const int lsa = (lp ^ ((slot >> 1) & 7)) << 3;
    const int hy = slot / HWD, hx = slot - hy * HWD;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const int ok = (int)(slot < NSLOT) & (int)((unsigned)ix < (unsigned)Wv) & (int)((unsigned)iy < (unsigned)Hv);
    const int64_t off = (int64_t)((pn * p.d.H + (iy >> p.ush)) * p.d.W + (ix >> p.ush)) * p.d.Cin + lsa;
    const uintptr_t a_ok = (uintptr_t)(xbase + off), a_zero = (uintptr_t)(zero + lsa);
    pa[i] = (const vq_bf16*)(ok ? a_ok : a_zero);
    inca[i] = ok ? BK : 0;
  }
  auto stage_piece = [&](int buf, int i) {             // i compile-time after unrolling
    if (wave + NW * i < PMAX) {
      if constexpr (ASMDMA) glds16_asm(pa[i], lds + buf * XTS + (wave + NW * i) * 8 * BK);
      else glds16(pa[i], lds + buf * XTS + (wave + NW * i) * 8 * BK);
      pa[i] += inca[i];
    }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- pixel fragments: pixel p_l = (ty, tx) of the patch, tap (kr, ks) -> halo row (ty + kr) * 18 + tx + ks ----------
  const int fr = lane & 31, fh = lane >> 5;
  int rowb[FP];
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int p_l = wp0 + b * 32 + (REGADDR ? tap9_perm(fr) : fr);
    rowb[b] = (p_l / TW) * HWD + (p_l % TW);
  }
  s16x8 bfr[2][FP];
  // REGADDR: byte address of (tap, fragment b) at k-step 0 in buffer 0, kept in registers: 9 * FP VGPRs instead of ~7 VALU
  // operations per read.  The slot index of k-step kk is ((2 kk) | fh) ^ key = (2 kk) ^ (fh ^ key) (2 kk has no bit 0), i.e.
  // byte bits 5-6, and the second buffer is 2^15 bytes away: address = abase ^ ((kk << 5) | (buf << 15)).
  unsigned abase[REGADDR ? 9 : 1][FP];
  if constexpr (REGADDR) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int b = 0; b < FP; ++b) {
        const int row = rowb[b] + (tap / 3) * HWD + (tap % 3);
        abase[tap][b] = (unsigned)(row * BK * 2 + ((frag_slot<X2>(0, fh) ^ ((row >> 1) & 7)) << 4));
      }
  }
  auto frag_load = [&](int buf, int tap, int kk, int slot) {
    if constexpr (REGADDR) {
      const unsigned x = frag_xor<X2>(kk) | (unsigned)(buf << 15);
#pragma unroll
      for (int b = 0; b < FP; ++b) bfr[slot][b] = *(const s16x8*)((const char*)lds + (abase[tap][b] ^ x));
      return;
    }
    const vq_bf16* base = lds + buf * XT;
    const int toff = (tap / 3) * HWD + (tap % 3);
#pragma unroll
    for (int b = 0; b < FP; ++b) {
      int row = rowb[b];
      VQ_OPAQUE_VGPR(row);                              // keeps the 36 x FP addresses out of registers (re-derived per read)
      row += toff;
      bfr[slot][b] = *(const s16x8*)(base + row * BK + ((frag_slot<X2>(kk, fh) ^ ((row >> 1) & 7)) << 3));
    }
  };

  // ---- weight fragments (fragment-order packed layout, see pack_weight_kernel layout 1) --------------------------
  s16x8 wf[BK / 16][FC];
  const vq_bf16* wrow[FC];
  {
    const int ncb = (p.d.Cout + 31) >> 5;
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      int cb = ((c0 + wc0) >> 5) + a;
      if (cb >= ncb) cb = ncb - 1;
      wrow[a] = p.w + ((int64_t)cb * (p.Kp >> 4)) * 512 + lane * 8;
    }
  }
  auto kb_of = [&](int tap, int cc) -> int { return ((tap * p.d.Cin) >> 4) + cc * (BK / 16); };
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
    for (int a = 0; a < FC; ++a) wf[kk][a] = *(const s16x8*)(wrow[a] + (int64_t)(kb_of(0, 0) + kk) * 512);

#pragma unroll
  for (int i = 0; i < PPW; ++i) stage_piece(0, i);
  wait_vmcnt<0>();
  const float alpha_s = conv_alpha_finish(p, alpha_raw);
  VQ_STAMP(11);
  raw_barrier();
  for (int cc = 0; cc < cpt; ++cc) {
    const int buf = cc & 1;
    const bool more_x = cc + 1 < cpt;
    frag_load(buf, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 36; ++v) {                     // v = tap * 4 + kk
      const int tap = v >> 2, kk = v & 3;
      if (v + 1 < 36) frag_load(buf, (v + 1) >> 2, (v + 1) & 3, (v + 1) & 1);
      vq_sched_fence();
      // VQ_F16X2: even step = hi pixel piece x (lo, hi) weights, odd step = lo pixel piece x hi weights (see conv_igemm_tap3_kernel)
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (X2) {
            if (!(kk & 1)) acc[a][b] = mfma16<DT>(wf[kk | 1][a], bfr[v & 1][b], acc[a][b]);
            acc[a][b] = mfma16<DT>(wf[kk & ~1][a], bfr[v & 1][b], acc[a][b]);
          } else acc[a][b] = mfma16<DT>(wf[kk][a], bfr[v & 1][b], acc[a][b]);
        }
      vq_sched_fence();
      const int rk = X2 ? (kk ^ 1) : kk;             // the weight register whose last use this step was
      if (v < PPW && more_x) stage_piece(buf ^ 1, v);  // next chunk's DMA, one piece per step
      // refill the weight registers of this k-step for the next (tap, chunk)
      int ntap = tap + 1, ncc = cc;
      if (ntap == 9) { ntap = 0; ++ncc; }
      if constexpr (ASMDMA) {
        // unconditional (the block's last four requests re-read chunk 0 and are never used): a request under a run-time
        // condition makes hipcc assume the worst at every later wait — the last k-steps of every chunk drained the queue
        if (ncc >= cpt) ncc = 0;
#pragma unroll
        for (int a = 0; a < FC; ++a) wf[rk][a] = *(const s16x8*)(wrow[a] + (int64_t)(kb_of(ntap, ncc) + rk) * 512);
      } else if (ncc < cpt) {
#pragma unroll
        for (int a = 0; a < FC; ++a) wf[rk][a] = *(const s16x8*)(wrow[a] + (int64_t)(kb_of(ntap, ncc) + rk) * 512);
      }
    }
    // every DMA piece of this chunk is older than the weight requests of its last four k-steps, which may stay in flight
    if constexpr (ASMDMA) wait_vmcnt<4 * FC>();
    else { if (more_x) wait_vmcnt<FC>(); else wait_vmcnt<0>(); }
    raw_barrier();
  }
  VQ_STAMP(12);
  // (all 8 items of a thread in ONE round — MAXU = 8 — was measured: +-0, profiles/r3k_*)
  if constexpr (X2) igemm_epilogue_x2<BC, BP, WC, WP, REGADDR>(p, lds, acc, c0, p0, wc0, wp0, alpha_s, (pn * p.d.Ho + ty0) * p.d.Wo + tx0);
  else igemm_epilogue<DT, BC, BP, WC, WP, REGADDR>(p, lds, acc, c0, p0, wc0, wp0, alpha_s, (pn * p.d.Ho + ty0) * p.d.Wo + tx0);
  VQ_STAMP(13);
}

// ------------------------------------------------------------------------------ patch-conv data gradient, persistent
// The data gradient of a patch conv (kernel == stride: the PatchDiscriminator heads, utils.py:156-185) as vq_conv2d_fwd runs it — a
// 1x1 conv  dy[K = Cout of the head]  ->  rows (tap, ci)  with a depth-to-space store — is a pure STORE problem: K is 32, a
// 128 x 128 output tile is 32 KiB of output for 8-16 MFMAs per wave.  Through the generic tile kernels every such tile was a block
// of its own (prologue, LDS staging of both operands, barriers: 8192 blocks for the 64 -> 32 head at 256 x 256, 1.1 TB/s of stores,
// 0.57 ms per step over the heads).  Here a block keeps the weight fragments of its 128 rows in registers (KS x 4 VGPRs per wave)
// and walks a range of pixel tiles: the dy fragments come straight from global memory in MFMA operand order (lane = pixel, 8
// consecutive channels: one 16-byte load), the next tile's are requested before this tile's epilogue, nothing is staged.
// 4 waves x 32 rows x 128 pixels; the shared epilogue does the rest (alpha, ReLU mask, residual, depth-to-space addresses).
template <int DT, int KS>
__global__ __launch_bounds__(256, 2) void conv_patch_dgrad_kernel(const ConvParams p) {
  constexpr int BC = 128, BP = 128, WC = 32, WP = 128, FP = WP / 32, NW = 4;
  VQ_DYN_LDS(vq_bf16, lds);                            // the epilogue's 32-KiB transposition slab
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int n_ct = p.n_ctiles, groups = gridDim.x / n_ct;
  const int ctile = blockIdx.x % n_ct, grp = blockIdx.x / n_ct;
  if (grp >= groups) return;
  const int t_begin = (int)((int64_t)grp * p.n_ptiles / groups), t_end = (int)((int64_t)(grp + 1) * p.n_ptiles / groups);
  if (t_begin >= t_end) return;
  const int c0 = ctile * BC, wc0 = wave * WC;
  const int K = p.d.Cin;                               // channels of dy (padded), KS * 16
  // weights (pack layout 2): row (tap, ci) = K-contiguous run of Kp elements
  s16x8 wf[KS];
  {
    const vq_bf16* wr = p.w + (int64_t)(c0 + wc0 + fr) * p.Kp + fh * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) wf[kk] = *(const s16x8*)(wr + kk * 16);
  }
  const vq_bf16* xb = (const vq_bf16*)p.x + (int64_t)fr * K + fh * 8;
  s16x8 bf[KS][FP];
  auto load_tile = [&](int t) {
    const vq_bf16* src = xb + (int64_t)t * BP * K;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int b = 0; b < FP; ++b) bf[kk][b] = *(const s16x8*)(src + (int64_t)(b * 32) * K + kk * 16);
  };
  load_tile(t_begin);
  const float alpha_s = conv_alpha_finish(p, conv_alpha_request(p));     // once per block, not once per tile
  for (int t = t_begin; t < t_end; ++t) {
    f32x16 acc[1][FP];
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][b][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int b = 0; b < FP; ++b) acc[0][b] = mfma16<DT>(wf[kk], bf[kk][b], acc[0][b]);
    if (t + 1 < t_end) load_tile(t + 1);               // in flight under the epilogue
    igemm_epilogue<DT, BC, BP, WC, WP, 0, 4>(p, lds, acc, c0, t * BP, wc0, 0, alpha_s);
    raw_barrier();                                     // the slab is free again
  }
}

// ------------------------------------------------------------------------------ 256 x 256 tile over a staged 16 x 16 patch
// The 8-wave 256 x 256 tile of conv_igemm_glds_kernel (weights through LDS, ping-pong schedule) with the pixel operand of the
// nine-tap kernel: the tile's 256 pixels are a 16 x 16 PATCH of one image, staged once per 64-channel chunk with a one-pixel
// halo (18 x 18 = 324 LDS rows, two buffers) and read by all nine taps at row offsets kr * 18 + ks, while the 256 x 64 weight
// tile of every (chunk, tap) stage still arrives by LDS-DMA (two buffers).  K order (chunk, tap, k-step) over the unchanged
// tap-major packed weights.  Per stage a wave now issues its 4 weight pieces + at most ONE patch piece instead of 4 + 4: the
// one-tap tile re-staged the pixel tile for each of the nine taps — 32 KiB of LDS writes per 32 MFMAs per wave on top of the
// fragment reads that already take 75 % of the LDS cycles (r1 ablation: no DMA at all = +32 %), and, the tiles of an XCD's 32
// CUs being 8 MiB against 4 MiB of L2, 3-5x fetch amplification at the fabric (profiles/r2j_traffic_ref.txt).
// LDS: W0 | W1 (2 x 32 KiB) | X0 | X1 (2 x 41 KiB) = 146 KiB, one block per CU as before.
// S = 2 (round 3): the sub-pixel (phase-decomposed) Upsample forward — four 2x2 convs of the low-resolution input in one launch,
// rows = (phase, cout) — over the SAME staged patch: phase (a, b) of a block moves its 2x2 window by (a, b), so tap (r, s) reads halo
// row (ty + r + a) * 18 + tx + s + b, inside the 18 x 18 halo a 3x3 kernel needs.  These launches ran on the one-tap tile, which
// re-staged the pixel tile for each of the four taps of every phase block: 691 MB fetched per launch against 270 MB algorithmic
// (profiles/r2zz_traffic_ref.txt), the whole excess of the family's traffic.
template <int DT, int BC = 256, int S = 3>
__global__ __launch_bounds__(512) void conv_igemm_p9_kernel(const ConvParams p) {
  constexpr int BK = 64, BP = 256, WC = BC / 2, WP = 64;
  constexpr bool X2 = DT == VQ_F16X2;                  // (see conv_igemm_glds_kernel)
  constexpr int NTAP = S * S;
  constexpr int FC = WC / 32, FP = WP / 32, NWP = BP / WP, NW = 8;
  constexpr int TW = 16, TH = 16, HWD = TW + 2, NSLOT = (TH + 2) * HWD, PMAX = (NSLOT + 7) / 8;   // 324 halo rows, 41 pieces
  constexpr int WT = BC * BK, XT = PMAX * 8 * BK;      // elements per weight / patch buffer
  constexpr int XBASE = 2 * WT;                        // first element of X0
  constexpr int NBW = BC / 8 / NW;                     // weight pieces per wave per stage (4)
  static_assert(PMAX <= 6 * NW, "six patch pieces per wave and chunk");
  constexpr int XPT = (6 + NTAP - 1) / NTAP;           // patch pieces a wave issues per tap slot (1 with nine taps, 2 with four)

  VQ_DYN_LDS(vq_bf16, lds);

  const int tid = threadIdx.x;
  const float alpha_raw = conv_alpha_request(p);   // (consumed after the first tile wait: conv_alpha_finish)
  VQ_STAMP(10);
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
  const int pn = ptile / p.pt_tpi, prem = ptile - pn * p.pt_tpi, ptyi = prem / p.pt_tx;
  const int ty0 = ptyi * TH, tx0 = (prem - ptyi * p.pt_tx) * TW;    // top-left output pixel of the patch
  // sub-pixel conv: block-uniform phase (a, b) of this row tile (the tile height divides Cout / 4), window moved by it
  const int sub_ph = (S == 2 && p.sub) ? c0 / p.d2s_c : 0, sub_off = (sub_ph >> 1) * HWD + (sub_ph & 1);

  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;
  const int lr = lane >> 3, lp = lane & 7;             // row within an 8-row DMA piece, physical 16-byte slot
  const int cpt = p.d.Cin >> 6;                        // 64-channel chunks

  // ---- weight rows owned by this lane (piece wave * NBW + i of the 256-row tile) ---------------------------------------
  const vq_bf16* pb[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int row = (wave * NBW + i) * 8 + lr;
    int grow = c0 + row;
    if (grow >= p.d.Cout) grow = p.d.Cout - 1;
    pb[i] = p.w + (int64_t)grow * p.Kp + ((lp ^ ((row >> 1) & 7)) << 3);
  }
  auto stage_w = [&](int wbuf, int tap, int cc) {      // the 256 x 64 weight tile of (chunk cc, tap)
    const int koff = tap * p.d.Cin + cc * BK;
#pragma unroll
    for (int i = 0; i < NBW; ++i) glds16(pb[i] + koff, lds + wbuf * WT + (wave * NBW + i) * 8 * BK);
  };
  // patch pieces of this wave: piece i * NW + wave (8 halo slots each), i = 0..5 — its element offset in x at chunk 0, or -1 for
  // slots outside the image / beyond the 324 halo rows (zero page).  Six registers: re-deriving the position per stage put
  // ~300 cycles of quarter-rate integer math into the load slot of the ping-pong schedule, which then outlasted the other
  // group's 16 MFMAs (measured -17..27 % against the one-tap tile it was meant to beat).
  constexpr int XPW = (PMAX + NW - 1) / NW;            // 6
  int xo[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int slot = (i * NW + wave) * 8 + lr;
    const int lsa = (lp ^ ((slot >> 1) & 7)) << 3;
    const int hy = slot / HWD, hx = slot - hy * HWD;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = slot < NSLOT && (unsigned)ix < (unsigned)Wv && (unsigned)iy < (unsigned)Hv;
    // (element offsets of one tensor fit 31 bits: the caller's tensors are < 2^31 elements, checked by the launcher)
    xo[i] = ok ? ((pn * p.d.H + (iy >> p.ush)) * p.d.W + (ix >> p.ush)) * p.d.Cin + lsa : -1;
  }
  auto stage_x = [&](int xbuf, int i, int cc) {        // i compile-time after unrolling
    const int j = i * NW + wave;
    if (j < PMAX) {                                    // wave-uniform
      const int slot = j * 8 + lr;
      const int lsa = (lp ^ ((slot >> 1) & 7)) << 3;
      const vq_bf16* src = xo[i] >= 0 ? xbase + (int64_t)xo[i] + cc * BK : zero + lsa;
      glds16((const void*)src, lds + XBASE + xbuf * XT + j * 8 * BK);
    }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- fragment byte addresses at k-step 0: k-step kk enters by XOR (kk << 5) (16-byte slot (2 kk | fh) ^ key), the buffer by ADD
  // (buffer strides are multiples of the 128-byte row, so they commute with that XOR)
  const int fr = lane & 31, fh = lane >> 5;
  unsigned wab[FC];                                    // weight fragment a in W0
#pragma unroll
  for (int a = 0; a < FC; ++a) wab[a] = (unsigned)(Swz<BK>::elem(wc0 + a * 32 + fr, frag_slot<X2>(0, fh)) * 2);
  int row0[FP];                                        // halo row of pixel fragment b at tap (0, 0)
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int p_l = wp0 + b * 32 + tap9_perm(fr);
    row0[b] = (p_l / TW) * HWD + (p_l % TW);
  }
  unsigned xab[FP];                                    // (current tap, pixel fragment b) in X0: re-derived per stage (12 VALU
  auto set_tap = [&](int tap) {                        // per 32 MFMAs) rather than 18 registers on a 256-VGPR budget
#pragma unroll
    for (int b = 0; b < FP; ++b) {
      int row = row0[b];
      VQ_OPAQUE_VGPR(row);                              // opaque: nine taps' addresses must not be hoisted into registers
      row += (tap / S) * HWD + (tap % S) + sub_off;
      xab[b] = (unsigned)(XBASE * 2 + row * BK * 2 + ((frag_slot<X2>(0, fh) ^ ((row >> 1) & 7)) << 4));
    }
  };
  s16x8 af[2][FC], bfr[2][FP];
  auto frag_load = [&](unsigned woff, unsigned xoff, int kk, int slot) {   // kk, slot compile-time after unrolling
    const unsigned x = frag_xor<X2>(kk);
#pragma unroll
    for (int a = 0; a < FC; ++a) af[slot][a] = *(const s16x8*)((const char*)lds + ((wab[a] ^ x) + woff));
#pragma unroll
    for (int b = 0; b < FP; ++b) bfr[slot][b] = *(const s16x8*)((const char*)lds + ((xab[b] ^ x) + xoff));
  };

  // ---- prologue: weight tile of stage (0, 0) and the whole patch of chunk 0 --------------------------------------------------
  stage_w(0, 0, 0);
#pragma unroll
  for (int i = 0; i < XPW; ++i) stage_x(0, i, 0);
  wait_vmcnt<0>();
  const float alpha_s = conv_alpha_finish(p, alpha_raw);
  VQ_STAMP(11);
  raw_barrier();
  // Ping-pong schedule of conv_igemm_glds_kernel (PP): the two waves of a SIMD run one barrier apart, one in its MFMA slot at
  // raised priority while the other reads fragments / issues DMA.  A stage (chunk, tap) is what a chunk is there: its weight
  // buffer (s + 1) & 1 and — from tap 0 of a chunk on — the patch buffer (chunk + 1) & 1 were last read in the previous stage.
  const int grp = wave >> 2;
  if (grp == 1) raw_barrier();
  for (int cc = 0; cc < cpt; ++cc) {
    const bool more_c = cc + 1 < cpt;
    const unsigned xoff = (unsigned)((cc & 1) * XT * 2);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int wpar = (cc * NTAP + tap) & 1;          // parity of the stage index NTAP cc + tap
      const unsigned woff = (unsigned)(wpar * WT * 2);
      const bool more = more_c || tap < NTAP - 1;
      set_tap(tap);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        frag_load(woff, xoff, 2 * ph, 0);
        frag_load(woff, xoff, 2 * ph + 1, 1);
        if (ph == 0) {
          if (more) {
            if (tap < NTAP - 1) stage_w(wpar ^ 1, tap + 1, cc); else stage_w(wpar ^ 1, 0, cc + 1);
          }
          if (more_c) {
#pragma unroll
            for (int j = 0; j < XPT; ++j)
              if (tap * XPT + j < XPW) stage_x((cc + 1) & 1, tap * XPT + j, cc + 1);
          }
        }
        wait_lgkmcnt<0>();
        if (ph == 1) wait_vmcnt<0>();
        vq_sched_fence();
        raw_barrier();
        vq_sched_fence();
        vq_setprio(1);
        if constexpr (X2) mfma_x2_pair<FC, FP>(acc, af[0], af[1], bfr[0], bfr[1]);     // (a phase = one k-step pair)
        else
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
          for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FP; ++b) acc[a][b] = mfma16<DT>(af[kq][a], bfr[kq][b], acc[a][b]);
        vq_setprio(0);
        vq_sched_fence();
        raw_barrier();
        vq_sched_fence();
      }
    }
  }
  if (grp == 0) raw_barrier();
  VQ_STAMP(12);
  if constexpr (X2) igemm_epilogue_x2<BC, BP, WC, WP, 1>(p, lds, acc, c0, p0, wc0, wp0, alpha_s, (pn * p.d.Ho + ty0) * p.d.Wo + tx0);
  else igemm_epilogue<DT, BC, BP, WC, WP, 1>(p, lds, acc, c0, p0, wc0, wp0, alpha_s, (pn * p.d.Ho + ty0) * p.d.Wo + tx0);
  VQ_STAMP(13);
}

// ------------------------------------------------------------------------------ weight packing
// fwd: packed[row=co][k=(r*S+s)*Cin_pad+ci] = w[co][ci][r][s]
// dgrad: packed[row=ci][k=(r*S+s)*Cout_pad+co] = w[co][ci][R-1-r][S-1-s]
// layout 0: [row][Kp];  layout 1 ("fragment order", rows padded to 32): the 1-KiB block of (32-row block cb,
// 16-k block kb) holds, for lane l = (row & 31) + 32 * ((k & 15) >> 3), the 8 k-values of its MFMA a-operand,
// so a wave fetches one weight fragment with a single perfectly coalesced 16-B-per-lane global load;
// layout 2 (dgrad of a patch conv, kernel == stride): [R*S*rows_pad][roundup(kch_pad, 64)] — the transposed patch
// conv as a 1x1 conv: row = tap * Cin_pad + ci, k = co (taps not rotated).
// VQ_F16 operands are stored times s_w = 2^(14 - floor(log2 |w|max)): |w|max * s_w lies in [2^14, 2^15), 2^-14 .. 2^-24 of it are
// still normal binary16 numbers; 1/s_w goes to the consumer's epilogue through the job's scale slot {|w|max, s_w, 1/s_w, 0}.
__host__ __device__ __forceinline__ bool pack_half_range(int op_dtype) { return op_dtype == VQ_F16 || op_dtype == VQ_F16X2; }
__device__ __forceinline__ float pack_scale(const VqPackJob& j) {
  if (!pack_half_range(j.op_dtype)) return 1.f;
  const float amax = j.scale[0];
  const int E = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  if (!(amax > 0.f) || E == 128) return 1.f;       // all-zero (or non-finite) tensor
  int e = 14 - E;
  if (e > 100) e = 100;                            // sub-normal |w|max: keep s_w and 1/s_w finite
  return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ void pack_publish_scale(const VqPackJob& j, float sc) {
  if (pack_half_range(j.op_dtype)) { j.scale[1] = sc; j.scale[2] = 1.f / sc; j.scale[3] = 0.f; }   // powers of two: exact
}
__device__ __forceinline__ vq_bf16 pack_cvt(const VqPackJob& j, float v, float sc) {
  return pack_half_range(j.op_dtype) ? f2h(v * sc) : f2bf(v);
}
// VQ_F16X2 operands (include/vqhip.h): the reduction index runs over VIRTUAL channels — per 8 real channels the 8 hi values, then the
// 8 lo values — so a packed row is K_v = taps x 2 kch binary16 numbers.  In fragment order (layout 1) fragment f = 2 j + plane of a
// 64-wide virtual chunk holds, for the lower / upper 32 lanes, plane `plane` of the chunk's real 8-groups 2 j / 2 j + 1: the kernels
// issue A(j,hi) B(j,hi) + A(j,hi) B(j,lo) + A(j,lo) B(j,hi) per pair j.  -> element offset of the 8-value piece (row, virtual k of its
// first element) in fragment order:
__device__ __forceinline__ int64_t pack_x2_frag_offset(int row, int kv, int Kp) {
  const int chunk = kv >> 6, g = (kv >> 4) & 3, pl = (kv >> 3) & 1;
  return ((((int64_t)(row >> 5) * (Kp >> 4) + chunk * 4 + 2 * (g >> 1) + pl) * 64) + (row & 31) + 32 * (g & 1)) * 8;
}
// |w|max of unit `unit` of `n_units` equal slices of the master weight -> atomic max on the bit pattern (non-negative floats
// order like unsigned integers); the slot was zeroed before the launch
__device__ __forceinline__ void pack_amax_unit(const VqPackJob& j, int64_t unit, int64_t n_units, float* red) {
  const int64_t n = (int64_t)j.Cout_w * j.Cin_w * j.R * j.S;
  const int64_t per = (n + n_units - 1) / n_units;
  const int64_t beg = unit * per;
  int64_t end = beg + per;
  if (end > n) end = n;
  float m = 0.f;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) m = fmaxf(m, fabsf(j.w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) m = fmaxf(m, red[wv]);
    if (m > 0.f) atomicMax((unsigned*)j.scale, __float_as_uint(m));
  }
}

__device__ __forceinline__ void pack_one(const VqPackJob& j, int64_t i, float sc) {
  const float* __restrict__ w = j.w;
  vq_bf16* __restrict__ out = (vq_bf16*)j.out;
  const int Kp = j.Kp, R = j.R, S = j.S;
  const int row = (int)(i / Kp), k = (int)(i - (int64_t)row * Kp);
  if (j.op_dtype == VQ_F16X2) {                      // k = virtual reduction index: (tap, 8-group, plane, element)
    const int kch2 = 2 * j.kch_pad;
    int tap, rem, r_row = row;
    if (j.layout == 2) { tap = row / j.rows_pad; r_row = row - tap * j.rows_pad; rem = k; }
    else { tap = k / kch2; rem = k - tap * kch2; }
    const int ch = ((rem >> 4) << 3) + (rem & 7), pl = (rem >> 3) & 1;
    float v = 0.f;
    if (j.layout == 2) {
      const int r = tap / S, sx = tap - r * S;
      if (r_row < j.Cin_w && ch < j.Cout_w && rem < kch2) v = w[(((int64_t)ch * j.Cin_w + r_row) * R + r) * S + sx];
    } else if (tap < R * S) {
      const int r = tap / S, sx = tap - r * S;
      if (!j.dgrad) { if (row < j.Cout_w && ch < j.Cin_w) v = w[(((int64_t)row * j.Cin_w + ch) * R + r) * S + sx]; }
      else if (row < j.Cin_w && ch < j.Cout_w) v = w[(((int64_t)ch * j.Cin_w + row) * R + (R - 1 - r)) * S + (S - 1 - sx)];
    }
    v *= sc;
    const vq_f16 h = f2h(v);
    const vq_f16 val = pl ? f2h(v - h2f(h)) : h;
    out[j.layout == 1 ? pack_x2_frag_offset(row, k & ~7, Kp) + (k & 7) : i] = val;
    return;
  }
  if (j.layout == 2) {
    const int tap = row / j.rows_pad, ci = row - tap * j.rows_pad, r = tap / S, sx = tap - r * S;
    float v = 0.f;
    if (ci < j.Cin_w && k < j.Cout_w) v = w[(((int64_t)k * j.Cin_w + ci) * R + r) * S + sx];
    out[i] = pack_cvt(j, v, sc);
    return;
  }
  const int tap = k / j.kch_pad, ch = k - tap * j.kch_pad;
  float v = 0.f;
  if (tap < R * S) {
    int r = tap / S, s = tap - r * S;
    if (!j.dgrad) {
      if (row < j.Cout_w && ch < j.Cin_w) v = w[(((int64_t)row * j.Cin_w + ch) * R + r) * S + s];
    } else {
      if (row < j.Cin_w && ch < j.Cout_w) v = w[(((int64_t)ch * j.Cin_w + row) * R + (R - 1 - r)) * S + (S - 1 - s)];
    }
  }
  const vq_bf16 h = pack_cvt(j, v, sc);
  int64_t o = i;
  if (j.layout == 1) {
    const int cb = row >> 5, ri = row & 31, kb = k >> 4, ko = k & 15;
    o = (((int64_t)cb * (Kp >> 4) + kb) * 64 + ri + 32 * (ko >> 3)) * 8 + (ko & 7);
  }
  out[o] = h;
  if (j.split >= 3) {
    const float r1 = v - bf2f(h);
    const vq_bf16 m = f2bf(r1);
    out[j.total + o] = m;
    if (j.split == 6) out[2 * j.total + o] = f2bf(r1 - bf2f(m));
  }
}

// Tiled re-pack (j.tiled): one block owns the weight sub-tensor w[co0:co0+32][ci0:ci0+32][R*S] — read as 32 runs of
// 32*R*S contiguous floats, staged in LDS — and emits every packed 8-element (16 B) piece that depends on it:
// 64-byte (layout 0/2) or 512-byte (layout 1, lanes walk the 32 rows of one fragment block) contiguous stores.
// The element-wise pack_one path reads with a stride of R*S floats and writes 2 bytes per lane: ~10x slower on the
// 512-channel weights.  Host sets j.tiled only when both padded channel counts are multiples of 32, R*S <= 9 and the
// K padding is empty, so no pad region is left unwritten.
constexpr int PK_T = 32;
// RSC > 0: the tap count as a compile-time constant (3x3 weights are almost all of a model's bytes: the index arithmetic below
// divides by it and by 32 * RS per element — runtime 32-bit divisions are ~35 VALU instructions each on this part)
template <int RSC>
__device__ __forceinline__ void pack_tile_t(const VqPackJob& j, int64_t t, float* lds, float sc) {
  const int RS = RSC > 0 ? RSC : j.R * j.S;
  const int CoP = j.dgrad ? j.kch_pad : j.rows_pad, CiP = j.dgrad ? j.rows_pad : j.kch_pad;
  const int n_ci_t = CiP / PK_T;
  const int co0 = (int)(t / n_ci_t) * PK_T, ci0 = (int)(t % n_ci_t) * PK_T;
  const int run = PK_T * RS;                       // floats per cout row of the tile
  // whole tiles of 16-byte-aligned rows (every weight of the models but the 3-channel ends and what follows a 3-element bias in the
  // optimizer's flat buffer): 16 bytes per lane — 9 staging trips per thread instead of 36 (round 6: the re-pack ran at 0.29 of the HBM
  // peak on bytes that mostly sit in the Infinity Cache: issue-bound, not bandwidth-bound)
  const bool vec4 = co0 + PK_T <= j.Cout_w && ci0 + PK_T <= j.Cin_w && (run & 3) == 0 && (((int64_t)j.Cin_w * RS) & 3) == 0 &&
                    (((int64_t)ci0 * RS) & 3) == 0 && ((uintptr_t)j.w & 15) == 0;      // (block-uniform)
  if (vec4) {
    const int run4 = run >> 2;
    for (int e = threadIdx.x; e < PK_T * run4; e += blockDim.x) {
      const int co_l = e / run4, q = e - co_l * run4;
      const vq_f4 v = *(const vq_f4*)(j.w + ((int64_t)(co0 + co_l) * j.Cin_w + ci0) * RS + 4 * q);
      *(vq_f4*)(lds + co_l * run + 4 * q) = v;
    }
  } else
  for (int e = threadIdx.x; e < PK_T * run; e += blockDim.x) {
    const int co_l = e / run, rem = e - co_l * run;
    const int ci = ci0 + rem / RS, co = co0 + co_l;
    float v = 0.f;
    if (co < j.Cout_w && ci < j.Cin_w) v = j.w[((int64_t)co * j.Cin_w + ci0) * RS + rem];
    lds[e] = v;
  }
  __syncthreads();
  vq_bf16* __restrict__ out = (vq_bf16*)j.out;
  const int n_oct = PK_T * RS * (PK_T / 8);        // 16-byte pieces produced from this tile
  for (int q = threadIdx.x; q < n_oct; q += blockDim.x) {
    int row_l, tap, oct;                           // row_l: local row of the packed operand; oct: 8-group along k
    if (j.layout == 1) { row_l = q % PK_T; const int r2 = q / PK_T; oct = r2 % (PK_T / 8); tap = r2 / (PK_T / 8); }
    else { oct = q % (PK_T / 8); const int r2 = q / (PK_T / 8); tap = r2 % RS; row_l = r2 / RS; }
    const int tap_src = (j.dgrad && j.layout != 2) ? RS - 1 - tap : tap;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co_l = j.dgrad ? oct * 8 + e : row_l, ci_l = j.dgrad ? row_l : oct * 8 + e;
      v[e] = lds[co_l * run + ci_l * RS + tap_src];
    }
    const int row = (j.dgrad ? ci0 : co0) + row_l;
    const int kc = (j.dgrad ? co0 : ci0) + oct * 8;            // channel index along k
    int64_t o;
    if (j.layout == 2) o = ((int64_t)tap * j.rows_pad + row) * j.Kp + kc;
    else {
      const int k = tap * j.kch_pad + kc;
      if (j.layout == 0) o = (int64_t)row * j.Kp + k;
      else o = ((((int64_t)(row >> 5) * (j.Kp >> 4) + (k >> 4)) * 64) + (row & 31) + 32 * ((k & 15) >> 3)) * 8;
    }
    vq_bf16 h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = pack_cvt(j, v[e], sc);
    if (j.op_dtype == VQ_F16X2) {                    // hi piece, then the lo piece of the same 8 channels (virtual channels 16 g + 0..15)
      vq_bf16 l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = f2h(v[e] * sc - h2f(h[e]));
      int64_t oh, ol;
      if (j.layout == 2) { oh = ((int64_t)tap * j.rows_pad + row) * j.Kp + 2 * kc; ol = oh + 8; }
      else {
        const int kv = tap * 2 * j.kch_pad + 2 * kc;
        if (j.layout == 0) { oh = (int64_t)row * j.Kp + kv; ol = oh + 8; }
        else { oh = pack_x2_frag_offset(row, kv, j.Kp); ol = pack_x2_frag_offset(row, kv + 8, j.Kp); }
      }
      *(vq_u4*)(out + oh) = *(const vq_u4*)h;
      *(vq_u4*)(out + ol) = *(const vq_u4*)l;
      continue;
    }
    *(vq_u4*)(out + o) = *(const vq_u4*)h;
    if (j.split >= 3) {
      vq_bf16 l[8], l2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float r1 = v[e] - bf2f(h[e]);
        l[e] = f2bf(r1);
        l2[e] = f2bf(r1 - bf2f(l[e]));
      }
      *(vq_u4*)(out + j.total + o) = *(const vq_u4*)l;
      if (j.split == 6) *(vq_u4*)(out + 2 * j.total + o) = *(const vq_u4*)l2;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void pack_tile(const VqPackJob& j, int64_t t, float* lds, float sc) {
  const int RS = j.R * j.S;
  if (RS == 9) pack_tile_t<9>(j, t, lds, sc);
  else if (RS == 1) pack_tile_t<1>(j, t, lds, sc);
  else if (RS == 4) pack_tile_t<4>(j, t, lds, sc);
  else pack_tile_t<0>(j, t, lds, sc);
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const VqPackJob j) {
  __shared__ __attribute__((aligned(16))) float lds[PK_T * PK_T * 9];
  const float sc = pack_scale(j);
  if (blockIdx.x == 0 && threadIdx.x == 0) pack_publish_scale(j, sc);
  if (j.tiled) {
    for (int64_t t = blockIdx.x; t < j.n_units; t += gridDim.x) pack_tile(j, t, lds, sc);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < j.total; i += (int64_t)gridDim.x * blockDim.x)
    pack_one(j, i, sc);
}
__global__ __launch_bounds__(256) void pack_amax_kernel(const VqPackJob j) {
  __shared__ float red[4];
  pack_amax_unit(j, blockIdx.x, gridDim.x, red);
}

// All conv weights of an optimizer in ONE launch (after its step): block b serves the job whose block range holds b
// (binary search in the device table): one tile, or VQ_PACK_ELEMS_PER_BLOCK elements of an element-wise job.
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const VqPackJob* __restrict__ jobs, int n_jobs) {
  __shared__ __attribute__((aligned(16))) float lds[PK_T * PK_T * 9];
  int lo = 0, hi = n_jobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= b) lo = mid; else hi = mid - 1;
  }
  const VqPackJob j = jobs[lo];
  const float sc = pack_scale(j);
  if (b == j.block_start && threadIdx.x == 0) pack_publish_scale(j, sc);
  if (j.tiled) { pack_tile(j, b - j.block_start, lds, sc); return; }
  const int64_t beg = (b - j.block_start) * VQ_PACK_ELEMS_PER_BLOCK;
  int64_t end = beg + VQ_PACK_ELEMS_PER_BLOCK;
  if (end > j.total) end = j.total;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) pack_one(j, i, sc);
}
// |w|max of every VQ_F16 job of the table (same block -> job map as the pack launch): zero the slots, then the slices
__global__ void pack_amax_zero_multi_kernel(const VqPackJob* __restrict__ jobs, int n_jobs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_jobs && pack_half_range(jobs[i].op_dtype)) jobs[i].scale[0] = 0.f;
}
__global__ __launch_bounds__(256) void pack_amax_multi_kernel(const VqPackJob* __restrict__ jobs, int n_jobs) {
  __shared__ float red[4];
  int lo = 0, hi = n_jobs - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= b) lo = mid; else hi = mid - 1;
  }
  const VqPackJob j = jobs[lo];
  if (!pack_half_range(j.op_dtype)) return;         // block-uniform
  pack_amax_unit(j, b - j.block_start, j.n_units, red);
}

static int kp_of(int R, int S, int kch_pad) { return vq_round_up(R * S * kch_pad, 64); }

extern "C" size_t vq_packed_weight_elems(int rows_pad, int R, int S, int cin_pad, int split, int layout) {
  if (layout == 2) return (size_t)R * S * rows_pad * vq_round_up(cin_pad, 64);
  if (layout == 1) rows_pad = vq_round_up(rows_pad, 32);
  return (size_t)rows_pad * kp_of(R, S, cin_pad) * (split == 6 ? 3 : split == 3 ? 2 : 1);
}

static int pack_fill_job(VqPackJob* j, const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                         int split, int layout, void* packed, int dgrad, int op_dtype, float* scale) {
  VQ_REQUIRE(op_dtype == VQ_BF16 || (pack_half_range(op_dtype) && split == 1 && scale != nullptr), VQ_ERR_INVALID,
             "vq_pack_weight: op_dtype must be VQ_BF16, or VQ_F16 / VQ_F16X2 with split 1 and a scale slot (got dtype %d split %d scale %p)",
             op_dtype, split, (void*)scale);
  VQ_REQUIRE(layout == 0 || ((layout == 1 || (layout == 2 && dgrad)) && split == 1), VQ_ERR_INVALID,
             "vq_pack_weight: layout must be 0, or (split 1 only) 1, or 2 for dgrad");
  VQ_REQUIRE(w && packed, VQ_ERR_INVALID, "vq_pack_weight: null pointer");
  VQ_REQUIRE(split == 1 || split == 3 || split == 6, VQ_ERR_INVALID, "vq_pack_weight: split must be 1, 3 or 6 (got %d)", split);
  VQ_REQUIRE(Cout_pad % 8 == 0 && Cin_pad % 8 == 0 && Cout_pad >= Cout_w && Cin_pad >= Cin_w, VQ_ERR_INVALID,
             "vq_pack_weight: padded channel counts must be multiples of 8 and >= true counts");
  int rows = dgrad ? Cin_pad : Cout_pad;
  const int kch = dgrad ? Cout_pad : Cin_pad;
  if (layout == 1) rows = vq_round_up(rows, 32);
  j->w = w; j->out = packed;
  j->Cout_w = Cout_w; j->Cin_w = Cin_w; j->R = R; j->S = S;
  j->rows_pad = rows; j->kch_pad = kch;
  const int kch_v = op_dtype == VQ_F16X2 ? 2 * kch : kch;     // VQ_F16X2: K runs over virtual channels (hi and lo pieces)
  j->Kp = layout == 2 ? vq_round_up(kch_v, 64) : kp_of(R, S, kch_v);
  j->split = split; j->dgrad = dgrad; j->layout = layout;
  j->op_dtype = op_dtype; j->reserved0 = 0; j->scale = pack_half_range(op_dtype) ? scale : nullptr;
  j->total = (int64_t)rows * j->Kp * (layout == 2 ? R * S : 1);
  j->block_start = 0;
  j->tiled = (rows % 32 == 0 && kch % 32 == 0 && R * S <= 9 &&
              (layout == 2 ? kch_v % 64 == 0 : (R * S * kch_v) % 64 == 0)) ? 1 : 0;
  j->n_units = j->tiled ? (int64_t)(rows / 32) * (kch / 32) : vq_ceil_div(j->total, VQ_PACK_ELEMS_PER_BLOCK);
  return VQ_OK;
}
extern "C" int vq_pack_job(VqPackJob* job, const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                           int split, int layout, int dgrad, int op_dtype, float* scale, void* packed) {
  VQ_REQUIRE(job, VQ_ERR_INVALID, "vq_pack_job: null job");
  return pack_fill_job(job, w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, layout, packed, dgrad ? 1 : 0, op_dtype, scale);
}
extern "C" int64_t vq_pack_job_blocks(const VqPackJob* job) { return job ? job->n_units : 0; }
extern "C" int vq_pack_weights_multi(const VqPackJob* jobs_dev, int n_jobs, int64_t total_blocks, int with_scales, void* stream) {
  VQ_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), VQ_ERR_INVALID,
             "vq_pack_weights_multi: empty or oversized job table");
  if (with_scales) {
    hipLaunchKernelGGL(pack_amax_zero_multi_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    hipLaunchKernelGGL(pack_amax_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    VQ_CHECK_LAUNCH("vq_pack_weights_multi(amax)");
  }
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
  VQ_CHECK_LAUNCH("vq_pack_weights_multi");
  return VQ_OK;
}
static int pack_common(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                       int split, int layout, int op_dtype, float* scale, void* packed, void* stream, int dgrad) {
  VqPackJob j;
  int rc = pack_fill_job(&j, w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, layout, packed, dgrad, op_dtype, scale);
  if (rc) return rc;
  int64_t blocks = j.tiled ? j.n_units : vq_ceil_div(j.total, 256);
  if (blocks > 4096) blocks = 4096;
  if (pack_half_range(op_dtype)) {   // measure |w|max first (the slot is zeroed on the stream, the slices race with an atomic max)
    hipError_t e = hipMemsetAsync(scale, 0, 4 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { vq_set_error("vq_pack_weight: hipMemsetAsync: %s", hipGetErrorString(e)); return VQ_ERR_HIP; }
    const int64_t n = (int64_t)Cout_w * Cin_w * R * S;
    int64_t ab = vq_ceil_div(n, 4096);
    if (ab > 1024) ab = 1024;
    hipLaunchKernelGGL(pack_amax_kernel, dim3((unsigned)ab), dim3(256), 0, (hipStream_t)stream, j);
    VQ_CHECK_LAUNCH("vq_pack_weight(amax)");
  }
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, j);
  VQ_CHECK_LAUNCH("vq_pack_weight");
  return VQ_OK;
}
extern "C" int vq_pack_weight_fwd(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                                  int split, int layout, int op_dtype, float* scale, void* packed, void* stream) {
  return pack_common(w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, layout, op_dtype, scale, packed, stream, 0);
}
extern "C" int vq_pack_weight_dgrad(const float* w, int Cout_w, int Cin_w, int R, int S, int Cout_pad, int Cin_pad,
                                    int split, int layout, int op_dtype, float* scale, void* packed, void* stream) {
  return pack_common(w, Cout_w, Cin_w, R, S, Cout_pad, Cin_pad, split, layout, op_dtype, scale, packed, stream, 1);
}

// ------------------------------------------------------------------------------ sub-pixel weights
// Tap sums of a 3x3 weight for the phase-decomposed convolutions (include/vqhip.h, vq_subpixel_weights).  Row / column
// tap sets as 3-bit masks (bit r = tap r takes part); sums run r, then s ascending in fp32 (deterministic).
__global__ __launch_bounds__(256) void subpixel_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I,
                                                                int mode, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int o, i, mr, ms;
  if (mode == 1) {          // [I][O][4][4]
    const int kx = (int)(idx & 3), ky = (int)((idx >> 2) & 3);
    const int64_t q = idx >> 4;
    o = (int)(q % O); i = (int)(q / O);
    const int T[4] = {4, 6, 3, 1};
    mr = T[ky]; ms = T[kx];
  } else {                  // [4 * rows][kch][2][2]
    const int v = (int)(idx & 1), u = (int)((idx >> 1) & 1);
    const int64_t q = idx >> 2;
    const int kch = mode == 0 ? I : O, rows = mode == 0 ? O : I;
    const int kc = (int)(q % kch);
    const int64_t row = q / kch;
    const int ph = (int)(row / rows), rr = (int)(row - (int64_t)ph * rows), a = ph >> 1, b = ph & 1;
    if (mode == 0) { o = rr; i = kc; } else { o = kc; i = rr; }
    const int UP[2][2] = {{1, 6}, {3, 4}};     // R_a(u): Upsample forward
    const int DN[2][2] = {{4, 1}, {2, 0}};     // D_a(u): Downsample data gradient
    mr = mode == 0 ? UP[a][u] : DN[a][u];
    ms = mode == 0 ? UP[b][v] : DN[b][v];
  }
  const float* src = w + ((int64_t)o * I + i) * 9;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
      if (((mr >> r) & 1) && ((ms >> sx) & 1)) acc += src[r * 3 + sx];
  out[idx] = acc;
}
extern "C" int vq_subpixel_weights(const float* w, float* out, int O, int I, int mode, void* stream) {
  VQ_REQUIRE(w && out && O > 0 && I > 0, VQ_ERR_INVALID, "vq_subpixel_weights: null pointer or empty weight");
  VQ_REQUIRE(mode >= 0 && mode <= 2, VQ_ERR_INVALID, "vq_subpixel_weights: mode must be 0, 1 or 2 (got %d)", mode);
  const int64_t total = (int64_t)O * I * 16;
  hipLaunchKernelGGL(subpixel_weights_kernel, dim3((unsigned)vq_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, out,
                     O, I, mode, total);
  VQ_CHECK_LAUNCH("vq_subpixel_weights");
  return VQ_OK;
}

// dW (3x3, OIHW) of the Upsample conv from the weight gradient dW4 [I][O][4][4] of its transposed form (the 4x4 / stride-2
// conv over dy): dW[o][i][r][s] (+)= sum of dW4[i][o][ky][kx] over ky in K(r), kx in K(s); K(0) = {2,3}, K(1) = {1,2}, K(2) = {0,1}
// (the inverse of the tap sets T of vq_subpixel_weights mode 1).  Fixed summation order.
__global__ __launch_bounds__(256) void subpixel_wgrad_fold_kernel(const float* __restrict__ dw4, float* __restrict__ dw, int O, int I,
                                                                   int accumulate, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [I][O][3][3]: reads stay contiguous in (o, taps)
  if (idx >= total) return;
  const int sx = (int)(idx % 3), r = (int)((idx / 3) % 3);
  const int64_t q = idx / 9;
  const int o = (int)(q % O), i = (int)(q / O);
  const float* src = dw4 + ((int64_t)i * O + o) * 16;
  const int ky0 = 2 - r, kx0 = 2 - sx;
  float acc = src[ky0 * 4 + kx0];
  acc += src[ky0 * 4 + kx0 + 1];
  acc += src[(ky0 + 1) * 4 + kx0];
  acc += src[(ky0 + 1) * 4 + kx0 + 1];
  float* dst = dw + (((int64_t)o * I + i) * 3 + r) * 3 + sx;
  *dst = accumulate ? *dst + acc : acc;
}
extern "C" int vq_subpixel_wgrad_fold(const float* dw4, float* dw, int O, int I, int accumulate, void* stream) {
  VQ_REQUIRE(dw4 && dw && O > 0 && I > 0, VQ_ERR_INVALID, "vq_subpixel_wgrad_fold: null pointer or empty weight");
  const int64_t total = (int64_t)O * I * 9;
  hipLaunchKernelGGL(subpixel_wgrad_fold_kernel, dim3((unsigned)vq_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dw4, dw,
                     O, I, accumulate, total);
  VQ_CHECK_LAUNCH("vq_subpixel_wgrad_fold");
  return VQ_OK;
}

// ------------------------------------------------------------------------------ dispatch
// VQ_F16X2 (include/vqhip.h): the kernels of this file see such a tensor as a binary16 tensor of 2C VIRTUAL channels — hi and lo
// pieces interleaved per 8 real channels — so the descriptor they run on has Cin doubled (Cout stays real: the rows of the GEMM).
// The extern "C" entry points virtualise ONCE and hand the copy to the internal helpers below.
static inline VqConvDesc x2_virtual(const VqConvDesc* d) {
  VqConvDesc v = *d;
  if (v.dtype == VQ_F16X2) { v.Cin *= 2; v.Cin_w = v.Cin; }
  return v;
}
static inline bool dt16(int dt) { return dt == VQ_BF16 || dt == VQ_F16 || dt == VQ_F16X2; }
// LDS the VQ_F16X2 epilogue needs for a BC x BP tile: BP x CB floats (igemm_epilogue_x2)
constexpr size_t x2_epi_bytes(int BC, int BP) { return (size_t)BP * (BC >= 256 ? 128 : (BC >= 64 ? 64 : 32)) * sizeof(float); }
static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

template <int DT, int SPLIT, int BC, int BP, int WC, int WP, int BK>
static int launch_conv(ConvParams& p, hipStream_t stream) {
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != (BC / WC) * (BP / WP))) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, (BC / WC) * (BP / WP)); return VQ_ERR_UNSUPPORTED; }
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = (int)vq_ceil_div(p.M, BP);
  p.Kp = vq_round_up(p.RS * p.d.Cin, 64);
  const int grid = p.n_ctiles * p.n_ptiles;
  hipLaunchKernelGGL((conv_igemm_kernel<DT, SPLIT, BC, BP, WC, WP, BK>), dim3(grid), dim3(256), 0, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd");
  return VQ_OK;
}

// Tallest row tile a descriptor admits: in sub-pixel mode the rows of a block must lie in ONE phase block (the block's
// window shift is uniform), so the tile height has to divide Cout/4 (a multiple of 32, checked in vq_conv2d_fwd).
static int max_ctile(const VqConvDesc* d) {
  if (!d->subpix) return 256;
  const int c = d->Cout / 4;
  return c % 256 == 0 ? 256 : (c % 128 == 0 ? 128 : (c % 64 == 0 ? 64 : 32));
}

template <int DT, int SPLIT, int BK>
static int dispatch_tile(ConvParams& p, hipStream_t stream) {
  const int mct = max_ctile(&p.d);
  if (p.d.Cout > 64 && mct >= 128) return launch_conv<DT, SPLIT, 128, 128, 64, 64, BK>(p, stream);
  if (p.d.Cout > 32 && mct >= 64) return launch_conv<DT, SPLIT, 64, 128, 32, 64, BK>(p, stream);
  return launch_conv<DT, SPLIT, 32, 128, 32, 32, BK>(p, stream);
}

template <int DT, int BC, int BP, int WC, int WP, int WREG, int DBG = 0, int PP = 0>
static int launch_glds(ConvParams& p, hipStream_t stream) {
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != (BC / WC) * (BP / WP))) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, (BC / WC) * (BP / WP)); return VQ_ERR_UNSUPPORTED; }
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr size_t LDS_BYTES = (size_t)2 * ((WREG ? 0 : BC) + BP) * 64 * sizeof(vq_bf16);
  static_assert(DT != VQ_F16X2 || LDS_BYTES >= x2_epi_bytes(BC, BP), "the VQ_F16X2 epilogue transposes fp32 slices through the same LDS");
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = (int)vq_ceil_div(p.M, BP);
  const int grid = p.n_ctiles * p.n_ptiles;
  VQ_RESERVE_LDS((conv_igemm_glds_kernel<DT, BC, BP, WC, WP, WREG, DBG, PP>), LDS_BYTES, "vq_conv2d_fwd");
  hipLaunchKernelGGL((conv_igemm_glds_kernel<DT, BC, BP, WC, WP, WREG, DBG, PP>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(glds)");
  return VQ_OK;
}
// Kernel-selection hint of a descriptor (VqConvDesc.kernel_hint, include/vqhip.h): 0 = the library's own choice — what the product
// always passes.  Non-zero values travel IN the descriptor (no process-global dispatch state): tests use them to reach every shipped
// instantiation at small shapes, tools to A/B two shipped kernels on one shape.
//   bits 0-2 (tile): 1 = the 128x128 tiles, 2 = 32x128 tiles, 3 = the 256x256 tile, 5 = nine-tap kernel wherever the shape allows,
//                    6 = no three-tap / nine-tap kernel, 7 = three-tap kernel wherever eligible;  bit 3 (+8) = weights through LDS
//   bits 4.. (dbg):  512 = the one-tap 256x256 tile where the patch-staged one would run, 16 = 128-pixel tiles where the short-M
//                    rule picks 64-pixel ones;  everything else selects compile-time ablations / epilogue pricing knobs that
//                    exist only in `make ABLATE=1` builds: a release library refuses those values with VQ_ERR_UNSUPPORTED.
//                    (The measured-and-not-adopted KERNELS of rounds 2-3 — 128-row and 128 x 512 patch tiles, the resident-weight
//                    64-channel kernel; dbg 1024 / 2048 / 4096 / 24 / 8200-8203 — were removed in round 5: they last existed in
//                    commit a426375, csrc/experimental/conv_igemm_experimental.hip; their measurements are in HISTORY.md.)
static inline int hint_tile(const VqConvDesc* d) { return d->kernel_hint & 15; }
static inline int hint_dbg(const VqConvDesc* d) { return (d->kernel_hint >> 4) & 0xfffff; }
static bool hint_supported(const VqConvDesc* d) {
  const int t = hint_tile(d) & 7, g = hint_dbg(d);
#ifdef VQ_ABLATION_KERNELS
  (void)t;
  return !(g == 1024 || g == 2048 || g == 4096 || g == 24 || (g >= 8200 && g <= 8203));      // the removed kernels' hints
#else
  return t != 4 && (g == 0 || g == 512 || g == 16 || g == 40 || g == 48 || g == 56 || g == 72);
#endif
}

// data gradient of a patch conv (kernel == stride, no padding: the PatchDiscriminator heads, utils.py:156-185), as
// ops.conv_dgrad_raw describes it: a stride-1 conv over the R-fold zero-dilated dy with full padding.  Every output
// pixel has exactly ONE live tap, so it is run as a 1x1 conv dy[Cout] -> [R*S*Cin] with a depth-to-space store
// instead of R*S taps of which all but one multiply zeros.
static bool is_patch_dgrad(const VqConvDesc* d) {
  return d->dil_in > 1 && d->dil_in == d->R && d->R == d->S && d->stride == 1 && d->up == 1 && d->pad_t == d->R - 1 &&
         d->pad_l == d->S - 1 && d->Ho == d->H * d->R && d->Wo == d->W * d->S && d->split == 1;
}
static bool glds_eligible(const VqConvDesc* d) { return dt16(d->dtype) && d->split == 1 && d->Cin % 64 == 0; }
static bool glds_t256(const VqConvDesc* d) {
  const int tile = hint_tile(d) & 7;
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  if (tile == 5 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->dil_in == 1 && d->subpix == 0 &&
      d->Ho == d->H * d->up && d->Wo == d->W * d->up && d->Wo % 16 == 0 && d->Ho % 8 == 0)
    return false;   // nine-tap kernel forced (128-row tiles, register weights)
  return d->Cout > 64 && max_ctile(d) >= 256 &&
         (tile == 3 || ((tile == 0 || tile == 4) && d->Cout % 256 == 0 && (M >= 32768 || tile == 4)));
}
// direct-to-register weights: the 128x128 and 64x128 tiles (waves own disjoint, or at most pairwise shared,
// weight rows), except 1x1 convs (measured slower).  The 32x128 tile (4 waves on the same 32 rows) and the
// 256x256 tile keep the LDS path.
static bool tap9_shape_ok(const VqConvDesc* d);
// 3x3 layers with at most 32 (padded) output channels on large images — decoder.conv_out (128 -> 3), the data gradient of VGG
// conv1_1 (64 -> 3): HBM-bound layers that the one-tap 32-row tile ran at 1.5-2.4 TB/s of INPUT traffic because it staged the
// pixel tile once per tap (nine times the input through L2 -> LDS).  The nine-tap kernel stages it once.  hint 5 = at any size,
// dbg 40 = A/B against the one-tap tile.
static bool tap9_rows32(const VqConvDesc* d) {
  const int knob = hint_tile(d) & 7;
  return d->Cout <= 32 && d->Cin % 64 == 0 && tap9_shape_ok(d) && !is_patch_dgrad(d) &&
         (knob == 5 || (knob == 0 && hint_dbg(d) != 40 && (int64_t)d->N * d->Ho * d->Wo >= (int64_t)512 * 128));
}
static bool glds_wreg(const VqConvDesc* d) {
  return (hint_tile(d) & 8) == 0 && !glds_t256(d) && d->R * d->S > 1 &&
         ((d->Cout > 32 && max_ctile(d) >= 64) || tap9_rows32(d));
}
extern "C" int vq_conv_weight_layout(const VqConvDesc* d0) {
  if (!d0) return 0;
  const VqConvDesc dv = x2_virtual(d0);
  const VqConvDesc* d = &dv;
  if (is_patch_dgrad(d)) return 2;
  return (glds_eligible(d) && glds_wreg(d)) ? 1 : 0;
}

template <int DT, int BC, int BP, int WC, int WP>
static int launch_tap3(ConvParams& p, hipStream_t stream) {
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != (BC / WC) * (BP / WP))) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, (BC / WC) * (BP / WP)); return VQ_ERR_UNSUPPORTED; }
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr int PMAX = (BP + 2 * (BP / 16) + 7) / 8;
  constexpr size_t LDS_BYTES = (size_t)2 * PMAX * 8 * 64 * sizeof(vq_bf16);
  static_assert(LDS_BYTES >= (size_t)BP * BC * sizeof(vq_bf16), "the epilogue transposes the output tile through the same LDS");
  static_assert(DT != VQ_F16X2 || LDS_BYTES >= x2_epi_bytes(BC, BP), "the VQ_F16X2 epilogue transposes fp32 slices through the same LDS");
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = (int)vq_ceil_div(p.M, BP);
  const int grid = p.n_ctiles * p.n_ptiles;
  VQ_RESERVE_LDS((conv_igemm_tap3_kernel<DT, BC, BP, WC, WP>), LDS_BYTES, "vq_conv2d_fwd");
  hipLaunchKernelGGL((conv_igemm_tap3_kernel<DT, BC, BP, WC, WP>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(tap3)");
  return VQ_OK;
}
template <int DT, int BC, int BP, int WC, int WP, int WA = 0>
static int launch_tap9(ConvParams& p, hipStream_t stream) {
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != (BC / WC) * (BP / WP))) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, (BC / WC) * (BP / WP)); return VQ_ERR_UNSUPPORTED; }
  constexpr int NW = (BC / WC) * (BP / WP);
  constexpr int PMAX = ((BP / 16 + 2) * 18 + 7) / 8;
  constexpr size_t LDS_BYTES = ((WA & 2) ? (size_t)32768 : (size_t)PMAX * 8 * 64 * sizeof(vq_bf16)) + (size_t)PMAX * 8 * 64 * sizeof(vq_bf16);
  static_assert(LDS_BYTES >= (size_t)BP * BC * sizeof(vq_bf16), "the epilogue transposes the output tile through the same LDS");
  static_assert(DT != VQ_F16X2 || LDS_BYTES >= x2_epi_bytes(BC, BP), "the VQ_F16X2 epilogue transposes fp32 slices through the same LDS");
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = p.M / BP;
  p.pt_tx = p.d.Wo / 16;
  p.pt_tpi = p.pt_tx * (p.d.Ho / (BP / 16));
  const int grid = p.n_ctiles * p.n_ptiles;
  VQ_RESERVE_LDS((conv_igemm_tap9_kernel<DT, BC, BP, WC, WP, WA>), LDS_BYTES, "vq_conv2d_fwd");
  // A single 64-channel chunk (Cin = 64: VGG conv1_2, the 64-channel levels of the reference's own launch line) never touches the
  // second halo buffer: launched with the first one alone (>= the epilogue's transposition slab), so that the LDS no longer caps
  // the kernel at two blocks per CU (four fit its 123 VGPRs).  A K = 576 tile is short — 72 MFMAs per wave between a halo DMA that
  // must land first and an epilogue — and more resident blocks are what covers those two ends: forward +10-18 %, data gradient +5-8 %
  // (profiles/r6e_c64_onebuf_ab.txt; a deeper weight-fragment ring instead: +-1 %, r6f_tap9_wd_ab.txt).
  size_t lds_bytes = LDS_BYTES;
  if (DT != VQ_F16X2 && p.d.Cin == 64 && hint_dbg(&p.d) != 72) {      // (the VQ_F16X2 epilogue's fp32 slab is larger; dbg 72 = A/B)
    constexpr size_t ONE = (size_t)PMAX * 8 * 64 * sizeof(vq_bf16), EPI = (size_t)BP * BC * sizeof(vq_bf16);
    lds_bytes = ONE > EPI ? ONE : EPI;
  }
  hipLaunchKernelGGL((conv_igemm_tap9_kernel<DT, BC, BP, WC, WP, WA>), dim3(grid), dim3(NW * 64), lds_bytes, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(tap9)");
  return VQ_OK;
}
template <int DT, int BC = 256, int S = 3>
static int launch_p9(ConvParams& p, hipStream_t stream) {
  constexpr int BP = 256, NW = 8;
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != NW)) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, NW); return VQ_ERR_UNSUPPORTED; }
  constexpr int PMAX = (18 * 18 + 7) / 8;
  constexpr size_t LDS_BYTES = (size_t)2 * BC * 64 * sizeof(vq_bf16) + (size_t)2 * PMAX * 8 * 64 * sizeof(vq_bf16);
  static_assert(LDS_BYTES >= (size_t)BP * BC * sizeof(vq_bf16) && LDS_BYTES <= 160 * 1024, "epilogue transpose / LDS capacity");
  static_assert(DT != VQ_F16X2 || LDS_BYTES >= x2_epi_bytes(BC, BP), "the VQ_F16X2 epilogue transposes fp32 slices through the same LDS");
  if ((int64_t)p.d.N * p.d.H * p.d.W * p.d.Cin >= ((int64_t)1 << 31)) { vq_set_error("vq_conv2d_fwd(p9): input of 2^31 elements or more"); return VQ_ERR_UNSUPPORTED; }
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = p.M / BP;
  p.pt_tx = p.d.Wo / 16;
  p.pt_tpi = p.pt_tx * (p.d.Ho / 16);
  const int grid = p.n_ctiles * p.n_ptiles;
  VQ_RESERVE_LDS((conv_igemm_p9_kernel<DT, BC, S>), LDS_BYTES, "vq_conv2d_fwd");
  hipLaunchKernelGGL((conv_igemm_p9_kernel<DT, BC, S>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(p9)");
  return VQ_OK;
}
// conv_patch_dgrad_kernel: patch-conv data gradients (ConvParams already rewritten to the 1x1 form by vq_conv2d_fwd) with 32 or 64
// channels of dy, whole 128-row / 128-pixel tiles (template parameter KS = K / 16)
static bool patch_dgrad_persistent_ok(const ConvParams& p) {
  // (K = 64 — the 128 -> 64 head — was measured too: 69 vs 55 us on the LDS-DMA tile kernel, whose staging reads whole 128-byte
  // rows where this kernel's fragment loads touch a quarter of every row per instruction: profiles/r3e_patch_dgrad_micro.txt)
  return p.d.dtype != VQ_F16X2 && p.d2s > 0 && !p.sub && p.d.Cin == 32 && p.d.Cout % 128 == 0 && p.M % 128 == 0 && !p.gn_part &&
         (hint_tile(&p.d) & 7) == 0 && hint_dbg(&p.d) != 48 && (p.M >= 128 * 64 || hint_dbg(&p.d) == 56);    // dbg 56: at any size (tests)
}
template <int DT>
static int launch_patch_dgrad(ConvParams& p, hipStream_t stream) {
  constexpr size_t LDS_BYTES = (size_t)128 * 128 * sizeof(vq_bf16);
  p.n_ctiles = p.d.Cout / 128;
  p.n_ptiles = p.M / 128;
  p.pt_tx = 0; p.pt_tpi = 0;
  // two blocks per CU, every row tile the same number of pixel-tile ranges
  const int groups = std::max(1, std::min(512 / p.n_ctiles, p.n_ptiles / 4));
  const int grid = groups * p.n_ctiles;
  hipLaunchKernelGGL((conv_patch_dgrad_kernel<DT, 2>), dim3(grid), dim3(256), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(patch dgrad)");
  return VQ_OK;
}
// conv_igemm_tap9_kernel: 3x3 / stride 1 / pad 1 convs (also behind the nearest-2x gather, also as data gradients) whose
// output splits into 8 x 16 patches.  Measured (profiles/r1_tap9_v35.txt, B = 16): as 2 x 2 waves of 64c x 64p it beats
// the 128x128 register-weight tile (128 channels at 256x256: 765 -> 836 TFLOP/s fwd, 649 -> 695 dgrad) and the three-tap
// kernel (512 channels at 32x32: 811 -> 950), and loses to the 256x256 tile (256 ch at 128x128: 950 vs 934; 512 ch at 64x64:
// 1105 vs 1016) — so it takes over exactly where those two ran.  As 4 waves x 32c x 128p (every wave reads all 128 pixels'
// fragments from LDS) the 6x smaller DMA volume bought nothing (737 / 843 / 893 / 864): LDS fragment reads, not the fill,
// bound these tiles.
static bool tap9_shape_ok(const VqConvDesc* d) {
  return d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Ho == d->H * d->up &&
         d->Wo == d->W * d->up && d->Wo % 16 == 0 && d->Ho % 8 == 0 && d->subpix == 0;
}
// the phase-decomposed Upsample forward (VqConvDesc.subpix: 2x2 / stride 1 / pad 1 windows, 4 phase blocks of Cout / 4 rows) on
// images that split into 16 x 16 patches
static bool subpix_patch_ok(const VqConvDesc* d) {
  return d->subpix == 2 && d->R == 2 && d->S == 2 && d->stride == 1 && d->dil_in == 1 && d->up == 1 && d->pad_t == 1 && d->pad_l == 1 &&
         d->Ho == d->H && d->Wo == d->W && d->Wo % 16 == 0 && d->Ho % 16 == 0 && d->Cin % 64 == 0;
}
// conv_igemm_tap3_kernel: register-weight tiles of 3x3 / stride 1 / pad 1 convs (also behind a nearest-2x upsample,
// also as the data gradient of such a conv) whose output rows are a power of two >= 16 pixels long
static bool tap3_eligible(const VqConvDesc* d) {
  return (hint_tile(d) & 7) != 6 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && d->pad_t == 1 &&
         d->pad_l == 1 && d->Ho == d->H * d->up && d->Wo == d->W * d->up && d->Wo >= 16 && ilog2_exact(d->Wo) >= 0;
}

template <int DT>
static int dispatch_glds(ConvParams& p, hipStream_t stream) {
  const VqConvDesc* d = &p.d;
  const int knob = hint_tile(d) & 7, dbg = hint_dbg(d);
  (void)dbg;
  const bool wreg = glds_wreg(d);
  // measured (profiles/r1_tap3_ab_v22.txt): pays on the short-M layers (32x32 and 16x16 images: +11..34 %), not at 256x256
  const bool tap3 = wreg && p.d2s == 0 && tap3_eligible(d) && (p.M <= 16384 || knob == 7);
  p.wo_shift = ilog2_exact(p.d.Wo);
  const int mct = max_ctile(d);
  if (p.d.Cout > 64 && mct >= 128) {
    // 256x256 tile (8 waves x 128c x 64p, 128 KiB LDS): half the L2->LDS bytes per flop of the 128x128 tile
#ifdef VQ_ABLATION_KERNELS
    if (glds_t256(d) && dbg == 8) return launch_glds<DT, 256, 256, 128, 64, 0, 8>(p, stream);
    if (glds_t256(d) && dbg == 1) return launch_glds<DT, 256, 256, 128, 64, 0, 1>(p, stream);
    if (glds_t256(d) && knob == 4) return launch_glds<DT, 256, 256, 128, 64, 0, 0, 0>(p, stream);   // A/B: free-running loop
#endif
    // 3x3 / stride 1 / pad 1 layers whose images split into 16 x 16 patches: the same tile over a staged patch (nine taps per
    // staging); dbg 512 = A/B against the one-tap form
    if (glds_t256(d) && dbg != 512 && p.d2s == 0 && tap9_shape_ok(d) && p.d.Ho % 16 == 0 && p.d.Cin % 64 == 0)
      return launch_p9<DT>(p, stream);
    // ... and the sub-pixel Upsample forward (2x2 windows moved by the block's phase) over the same staged patch
    if (glds_t256(d) && dbg != 512 && subpix_patch_ok(d)) return launch_p9<DT, 256, 2>(p, stream);
    if (glds_t256(d)) return launch_glds<DT, 256, 256, 128, 64, 0, 0, 1>(p, stream);                                  // ping-pong schedule
#ifdef VQ_ABLATION_KERNELS   // profiling-only builds (make ABLATE=1): compile-time ablated copies of the 128x128 kernel
    if (dbg == 8) return launch_glds<DT, 128, 128, 64, 64, 0, 8>(p, stream);   // DMA issued, never waited for (wrong results)
    if (dbg == 1) return launch_glds<DT, 128, 128, 64, 64, 0, 1>(p, stream);
    if (dbg == 2) return launch_glds<DT, 128, 128, 64, 64, 0, 2>(p, stream);
    if (dbg == 3) return launch_glds<DT, 128, 128, 64, 64, 0, 3>(p, stream);
    if (dbg == 4) return launch_glds<DT, 128, 128, 64, 64, 0, 4>(p, stream);
#endif
    // small images (VGG conv5_x at 16x16: M = 4096): 128x128 tiles would leave half of the 256 CUs without a block
    if (knob == 2) return launch_glds<DT, 32, 128, 32, 32, 0>(p, stream);   // hint: 32x128 tiles
    const bool small = knob == 0 && vq_ceil_div(p.M, 128) * vq_ceil_div(p.d.Cout, 128) < 256;
    // nine-tap kernel: automatically where the 128x128 register-weight tile / the three-tap kernel would run with at least
    // one block per CU; hint 5 forces it wherever the shape allows, hint 6 switches it (and the three-tap kernel) off
    if (wreg && p.d2s == 0 && tap9_shape_ok(d) && (knob == 5 || (knob == 0 && !small))) {
      // measured on MI355X (profiles/r2_tap9_variants.txt, B = 16, bf16): WA = 3 vs the round-1 form +4..6 % at 128 channels /
      // 256x256, +11..15 % at 512 channels / 32x32; WA = 1 alone: no gain (202 VGPRs: one wave per SIMD fewer)
#ifdef VQ_ABLATION_KERNELS
      if (dbg == 64) return launch_tap9<DT, 128, 128, 32, 128>(p, stream);
      if (dbg == 128) return launch_tap9<DT, 128, 128, 64, 64, 0>(p, stream);      // the round-1 form
#endif
      // (Round 6: this tile's weight stream — 2 KB per k-step and wave from L2 = the CU's 64 B/clk vector-memory path (its width on CDNA parts as far as we know) for as long as the k-step's four
      // MFMAs last — halved by 8 waves x 64c x 128p over 32 x 16 patches, launch_tap9<DT, 128, 512, 64, 128, 1>: 158 KB of LDS = one
      // block per CU, 256 VGPRs + 228 B of scratch, no register addresses: 26-34 % SLOWER at 128 -> 128 and 256 -> 128 @256^2,
      // profiles/r6t_tap9_128x512_ab.txt.  Not kept.)
      return launch_tap9<DT, 128, 128, 64, 64, 3>(p, stream);
    }
    if (!small) {
      if (tap3) return launch_tap3<DT, 128, 128, 32, 128>(p, stream);
#ifdef VQ_ABLATION_KERNELS
      // A/B candidate (dbg 32): the register-weight one-tap tile as 2 x 2 waves of 64c x 64p
      if (wreg && dbg == 32) return launch_glds<DT, 128, 128, 64, 64, 1>(p, stream);
#endif
      if (wreg) return launch_glds<DT, 128, 128, 32, 128, 1>(p, stream);
      return launch_glds<DT, 128, 128, 64, 64, 0>(p, stream);
    }
  }
  // the nine-tap kernel as a 64-row tile, 4 waves x 64c x 32p (VGG conv1_2, 64 -> 64 at 256x256): measured +17..18 % forward and
  // data gradient over the one-tap register-weight tile (profiles/r2_tap9_variants.txt); hint 5 forces it
  if (p.d.Cout > 32 && mct >= 64 && wreg && p.d2s == 0 && tap9_shape_ok(d) &&
      (knob == 5 || (knob == 0 && dbg != 128 && vq_ceil_div(p.M, 128) >= 512))) {
#ifdef VQ_ABLATION_KERNELS
    if (dbg == 256) return launch_tap9<DT, 64, 128, 64, 32, 0>(p, stream);
#endif
    // (Round 6: the same tile over 16 x 16 patches as 4 waves x 64c x 64p — launch_tap9<DT, 64, 256, 64, 64, 1>, half the weight bytes
    // every wave pulls from L2 per pixel, 84 KB of LDS = one block per CU — measured 3-6 % SLOWER forward and data gradient at
    // 64 -> 64 @512^2 (B = 12) in bf16 and binary16, profiles/r6d_c64_ab.txt: the weight stream is not what bounds this tile.  Not kept.)
    return launch_tap9<DT, 64, 128, 64, 32, 3>(p, stream);
  }
  // Short-M layers (VGG conv5_x: 512 channels at 16 x 16, M = 4096 at B = 16): 64 x 128 tiles are 256 four-wave blocks — ONE wave per
  // SIMD, nothing to hide an LDS or weight-fetch latency under (measured 290-300 TFLOP/s).  64 x 64 tiles (4 waves x 32c x 32p) put
  // two blocks on every CU.  dbg 16 = A/B against the 128-pixel tile.  (Not with GroupNorm partials: their row length is the tile's.)
  if (p.d.Cout > 32 && mct >= 64 && tap3 && !p.gn_part && !p.gnb && knob == 0 && dbg != 16 &&
      vq_ceil_div(p.M, 128) * vq_ceil_div(p.d.Cout, 64) <= 256)
    return launch_tap3<DT, 64, 64, 32, 32>(p, stream);
  if (p.d.Cout > 32 && mct >= 64 && tap3) return launch_tap3<DT, 64, 128, 32, 64>(p, stream);
  if (p.d.Cout > 32 && mct >= 64)
    return wreg ? launch_glds<DT, 64, 128, 32, 64, 1>(p, stream) : launch_glds<DT, 64, 128, 32, 64, 0>(p, stream);
  if (wreg && tap9_rows32(d)) return launch_tap9<DT, 32, 128, 32, 32, 3>(p, stream);   // (register weights: see glds_wreg)
  return launch_glds<DT, 32, 128, 32, 32, 0>(p, stream);   // (Cout <= 32, or phase blocks of 32 / 96 / ... channels)
}

// pixel tile / wave count of the kernel a GroupNorm-partial-capable descriptor is dispatched to: the 8-wave 256 x 256 tile or one
// of the 4-wave 128-pixel tiles (the launchers re-check both against their template parameters)
static int gn_kernel_bp(const VqConvDesc* d) {
  return (glds_eligible(d) && d->Cout > 64 && max_ctile(d) >= 128 && glds_t256(d)) ? 256 : 128;
}
static int gn_kernel_waves(int bp) { return bp >= 256 ? 8 : 4; }
// Pixels per GroupNorm partial ROW (one row per wave of a tile) of the kernel this descriptor is dispatched to, or 0 when its epilogue cannot produce the
// partials (fp32 storage, the 8-channel image kernels, depth-to-space stores, tiles that straddle images, group sizes other than
// 4 / 8 / 16 / 32 channels).  MUST mirror dispatch_glds / dispatch_tile: the launchers re-check it.
static int gn_tile_impl(const VqConvDesc* d, int groups) {       // d: virtualised (x2_virtual)
  if (!d || groups <= 0 || !dt16(d->dtype) || d->split != 1) return 0;
  if (d->subpix || is_patch_dgrad(d) || d->Cout != d->Cout_w || d->Cout % groups) return 0;
  const int cg = d->Cout / groups;
  if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return 0;
  if (d->dtype != VQ_F16X2 && d->Cin == 8 && d->R == 3 && d->S == 3) return 0;     // conv_small.hip
  if (vq_conv_c8_x2_shape(d)) return 0;                                            //   ... its VQ_F16X2 twin
  const int bp = gn_kernel_bp(d);
  if (((int64_t)d->Ho * d->Wo) % bp) return 0;
  return bp / gn_kernel_waves(bp);
}
extern "C" int vq_conv2d_gn_tile(const VqConvDesc* d, int groups) {
  if (!d) return 0;
  const VqConvDesc dv = x2_virtual(d);
  return gn_tile_impl(&dv, groups);
}

// Rows per image of the fused GroupNorm-backward sums: every kernel a descriptor without depth-to-space output can reach through
// dispatch_glds / dispatch_tile writes one row per wave = per 32 output pixels (the 64 x 64 short-M tile, 16 pixels per wave, is not
// dispatched with the sums on); tiles must not straddle images (256 pixels is the largest tile).
extern "C" int vq_conv2d_gnb_rows(const VqConvDesc* d) {
#ifndef VQ_ABLATION_KERNELS
  (void)d;
  return 0;            // a release library does not carry the fused path (see VQ_GNB above): callers fall back to the reduction pass
#endif
  if (!d || (d->dtype != VQ_BF16 && d->dtype != VQ_F16) || d->split != 1) return 0;
  if (d->subpix || is_patch_dgrad(d) || d->Cout != d->Cout_w || d->Cout % 8) return 0;
  if (d->Cin == 8 && d->R == 3 && d->S == 3) return 0;                          // conv_small.hip
  const int64_t hw = (int64_t)d->Ho * d->Wo;
  if (hw % 256) return 0;
  return (int)(hw / 32);
}

extern "C" int vq_conv2d_fwd(const VqConvDesc* d0, const void* x, const void* w_packed, const float* bias,
                             const void* residual, const void* relu_mask, void* y, float* gn_partials, int gn_groups,
                             void* stream) {
  VQ_REQUIRE(d0 && x && w_packed && y, VQ_ERR_INVALID, "vq_conv2d_fwd: null pointer");
  VQ_REQUIRE(d0->Cin_w <= d0->Cin && d0->Cout_w <= d0->Cout, VQ_ERR_INVALID, "vq_conv2d_fwd: true channels exceed padded");
  const VqConvDesc dvirt = x2_virtual(d0);             // VQ_F16X2: Cin counts virtual channels from here on
  const VqConvDesc* d = &dvirt;
  VQ_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->Cin > 0 && d->Cout > 0, VQ_ERR_INVALID,
             "vq_conv2d_fwd: channel counts must be positive multiples of 8 (Cin=%d Cout=%d)", d->Cin, d->Cout);
  VQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->R > 0 && d->S > 0, VQ_ERR_INVALID,
             "vq_conv2d_fwd: empty tensor");
  {
    const int dsh0 = ilog2_exact(d->dil_in), ush0 = ilog2_exact(d->up);
    VQ_REQUIRE(dsh0 >= 0 && ush0 >= 0 && ush0 <= 1 && d->stride >= 1, VQ_ERR_UNSUPPORTED,
               "vq_conv2d_fwd: dil_in must be a power of two, up in {1,2} (dil_in=%d up=%d)", d->dil_in, d->up);
    VQ_REQUIRE(!(dsh0 > 0 && ush0 > 0), VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: dil_in and up cannot be combined");
  }
  VQ_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31), VQ_ERR_UNSUPPORTED,
             "vq_conv2d_fwd: pixel count exceeds int32");
  VQ_REQUIRE(d->Cin_w <= d->Cin && d->Cout_w <= d->Cout, VQ_ERR_INVALID, "vq_conv2d_fwd: true channels exceed padded");
  ConvParams p;
  p.d = *d;
  p.d2s = 0; p.d2s_c = 0; p.sub = 0; p.pt_tx = 0; p.pt_tpi = 0;
  if (d->subpix) {   // phase-decomposed conv (include/vqhip.h): 4 row blocks, window moved by the phase, depth-to-space store
    VQ_REQUIRE(d->subpix == 2 && d->up == 1 && d->dil_in == 1 && d->Cout % 128 == 0 && d->Cout_w == d->Cout, VQ_ERR_UNSUPPORTED,
               "vq_conv2d_fwd: subpix must be 2 with up = dil_in = 1 and Cout = Cout_w = 4 * (a multiple of 32) (subpix=%d up=%d "
               "dil_in=%d Cout=%d Cout_w=%d)", d->subpix, d->up, d->dil_in, d->Cout, d->Cout_w);
    VQ_REQUIRE((int64_t)d->N * d->Ho * d->Wo * 4 < (1ll << 31), VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: pixel count exceeds int32");
    p.d2s = 2; p.d2s_c = d->Cout / 4; p.sub = 1;
  }
  VqConvDesc pd;
  if (is_patch_dgrad(d)) {   // -> 1x1 conv over the (small) dy image, rows = (tap, ci), depth-to-space store
    VQ_REQUIRE(bias == nullptr && !d->relu, VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: patch data-gradient takes no bias / relu");
    pd = *d;
    pd.Ho = d->H; pd.Wo = d->W; pd.Cout = d->R * d->S * d->Cout; pd.Cout_w = pd.Cout;
    pd.R = pd.S = 1; pd.dil_in = 1; pd.pad_t = pd.pad_l = 0;
    p.d = pd; p.d2s = d->R; p.d2s_c = d->Cout;
    d = &pd;
  }
  const int dsh = ilog2_exact(d->dil_in), ush = ilog2_exact(d->up);
  p.x = x; p.w = (const vq_bf16*)w_packed; p.bias = bias; p.residual = residual; p.relu_mask = relu_mask; p.y = y;
  p.M = d->N * d->Ho * d->Wo;
  p.HoWo = d->Ho * d->Wo;
  p.RS = d->R * d->S;
  p.G8 = d->Cin / 8;
  p.dsh = dsh; p.ush = ush;
  p.Kp = vq_round_up(p.RS * d->Cin, 64);
  p.lo_off = (int64_t)d->Cout * p.Kp;
  p.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  p.alpha_dev = d->alpha_dev;
  p.gn_part = nullptr; p.gn_G = p.gn_cg = p.gn_bp = p.gn_tiles = p.gn_nw = 0;
  p.gnb = 0; p.gnb_mean = p.gnb_rstd = p.gnb_gamma = p.gnb_beta = nullptr; p.gnb_part = nullptr; p.gnb_G = p.gnb_silu = p.gnb_rows = 0;
  if (d->gn_bwd) {
    const VqGnBwdFuse* f = d->gn_bwd;
    VQ_REQUIRE(vq_conv2d_gnb_rows(d) > 0, VQ_ERR_UNSUPPORTED,
               "vq_conv2d_fwd: this descriptor cannot form the fused GroupNorm-backward sums (ask vq_conv2d_gnb_rows first)");
    VQ_REQUIRE(f->x && f->mean && f->rstd && f->gamma && f->beta && f->part && f->groups > 0 && d->Cout % f->groups == 0, VQ_ERR_INVALID,
               "vq_conv2d_fwd: incomplete VqGnBwdFuse");
    VQ_REQUIRE(!residual && !relu_mask && !gn_partials && !d->relu, VQ_ERR_UNSUPPORTED,
               "vq_conv2d_fwd: gn_bwd excludes residual / relu_mask / gn_partials / relu on the same call");
    VQ_REQUIRE(hint_dbg(d) == 0 && (hint_tile(d) & 7) != 4, VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: gn_bwd with a kernel_hint that forces an experimental tile");
    p.gnb = 1; p.gnb_mean = f->mean; p.gnb_rstd = f->rstd; p.gnb_gamma = f->gamma; p.gnb_beta = f->beta; p.gnb_part = f->part;
    p.gnb_G = f->groups; p.gnb_silu = f->silu; p.gnb_rows = vq_conv2d_gnb_rows(d);
    p.residual = f->x;                                 // read through the residual operand's request path
  }
  {
    const int ws = ilog2_exact(d->Wo), hs = ilog2_exact(d->Ho);
    p.pix_wsh = (ws >= 0 && hs >= 0) ? ws : -1;
    p.pix_hwsh = (ws >= 0 && hs >= 0) ? ws + hs : -1;
  }
  VQ_REQUIRE(hint_supported(d), VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: kernel_hint %d selects a kernel that only exists in `make ABLATE=1` builds",
             d->kernel_hint);
  {
    const int g = hint_dbg(d);
    p.skip_epilogue = (g == 8192 || g == 8201 || g == 8203) ? 1 : g == 8193 ? 2 : g == 8194 ? 3 : g == 8195 ? 4 : g == 8196 ? 5 : 0;
  }
  p.range_events = (d->dtype == VQ_F16 || d->dtype == VQ_F16X2) ? d->range_events : nullptr;
  if (gn_partials) {
    VQ_REQUIRE(gn_tile_impl(d, gn_groups) > 0, VQ_ERR_UNSUPPORTED,
               "vq_conv2d_fwd: this descriptor cannot produce GroupNorm partials (ask vq_conv2d_gn_tile first)");
    const int bp = gn_kernel_bp(d);
    p.gn_part = gn_partials; p.gn_G = gn_groups; p.gn_cg = d->Cout / gn_groups; p.gn_bp = bp; p.gn_tiles = (d->Ho * d->Wo) / bp;
    p.gn_nw = gn_kernel_waves(bp);
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == VQ_BF16 || d->dtype == VQ_F16) {
    VQ_REQUIRE(d->split == 1, VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: 16-bit storage supports split=1 only");
    const int rc8 = vq_launch_conv_c8(d, x, w_packed, bias, residual, relu_mask, y, p.alpha, p.alpha_dev, s);   // 3-channel image layers
    if (rc8 <= 0) return rc8;
    if (patch_dgrad_persistent_ok(p)) return d->dtype == VQ_F16 ? launch_patch_dgrad<VQ_F16>(p, s) : launch_patch_dgrad<VQ_BF16>(p, s);
    if (d->dtype == VQ_F16) return glds_eligible(d) ? dispatch_glds<VQ_F16>(p, s) : dispatch_tile<VQ_F16, 1, 64>(p, s);
    if (glds_eligible(d)) return dispatch_glds<VQ_BF16>(p, s);
    return dispatch_tile<VQ_BF16, 1, 64>(p, s);
  } else if (d->dtype == VQ_F16X2) {
    VQ_REQUIRE(d->split == 1 && d->gn_bwd == nullptr, VQ_ERR_UNSUPPORTED, "vq_conv2d_fwd: VQ_F16X2 storage takes split = 1 and no gn_bwd");
    if (!gn_partials) {                                // 3-channel image layers (the 8-channel kernel forms no GroupNorm partials)
      const int rc8 = vq_launch_conv_c8_x2(d, x, w_packed, bias, residual, relu_mask, y, p.alpha, p.alpha_dev, s);
      if (rc8 <= 0) return rc8;
    }
    return glds_eligible(d) ? dispatch_glds<VQ_F16X2>(p, s) : dispatch_tile<VQ_F16X2, 1, 64>(p, s);
  } else if (d->dtype == VQ_F32) {
    if (d->split == 1) return dispatch_tile<VQ_F32, 1, 64>(p, s);
    if (d->split == 3) return dispatch_tile<VQ_F32, 3, 32>(p, s);     // (16-wide chunks measured: 33.8 vs 43.9 img/s, profiles/r4h_*)
    if (d->split == 6) return dispatch_tile<VQ_F32, 6, 16>(p, s);     // three planes per operand: 16-wide chunks keep the tile in 48 KiB
    vq_set_error("vq_conv2d_fwd: split must be 1, 3 or 6 (got %d)", d->split);
    return VQ_ERR_UNSUPPORTED;
  }
  vq_set_error("vq_conv2d_fwd: unknown dtype %d", d->dtype);
  return VQ_ERR_INVALID;
}
