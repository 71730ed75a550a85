// Weight gradient of the convolutions as a split-K MFMA GEMM (gfx950).
//
// Replaces autograd's wgrad for every nn.Conv2d on the train-step path (same call sites as
// conv_igemm.hip; reduction length N*Ho*Wo is up to 2^20 at config 2).
//
//   dW[co][tap][ci] = sum_p dY[p][co] * Xg[p][tap][ci]       (p = output pixel)
//
// Both operands are pixel-major in HBM (channels contiguous) while MFMA wants the reduction
// index (pixels) contiguous per lane, so the tiles are kept in their HBM order in LDS
// ([pixel][channel], rows padded by 64 B so 4 consecutive pixel rows tile the 256 B bank row)
// and the fragments are fetched with the gfx950 LDS transpose read ds_read_b64_tr_b16.
// Grid: (cout-tile, cin-tile, tap) x split-K; each block writes an fp32 partial tile into the
// caller's workspace; wgrad_reduce_kernel sums the splits in a fixed order (deterministic) and
// scatters into the OIHW fp32 gradient.
#include "vq_common.h"

struct WgradParams {
  VqConvDesc d;
  const void* x;
  const void* dy;
  float* part;      // [split][tap][Cout][Cin]
  float* bias_part; // [split][Cout] or nullptr (LDS-DMA kernels fuse the bias gradient)
  int dbg;          // ABLATE builds only
  int M, HoWo, RS;
  int n_ct, n_cit;
  int dsh, ush;
  int pix_per_split;  // multiple of BKP
  int nsplit;
  int xcd_tiles;    // three-tap kernel: 0 = an XCD owns splits (all tiles; split counts multiples of 8), 1 = an XCD owns tiles / 8 tiles
                    // with all their splits, 2 = an XCD owns a contiguous range of the split-major (split, tile) list
  int wo_shift, ho_shift;  // log2(Wo), log2(Ho) when both are powers of two, else -1
  int seg_shift;           // three-tap kernel: log2 of the row-segment length (largest power of two <= 64 dividing Wo)
  int x3;                  // VQ_F16X2 operands, native three-product form of the three-tap kernel: `d` holds VIRTUAL channel counts (2 x real),
                           // the partial slabs / bias partials are indexed by REAL channels
};

#ifdef VQ_STAMPS_ON
// cycle stamps of block 0 / thread 0 (tools only: `make ablate`, read back with vq_debug_stamps_wgrad)
__device__ long long g_vq_wstamps[64];
__device__ int g_vq_wstamp_n;
#define VQ_WSTAMP(id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); \
    const int k_ = g_vq_wstamp_n; if (k_ < 64) { g_vq_wstamps[k_] = ((long long)(id) << 56) | (t_ & 0x00ffffffffffffffll); g_vq_wstamp_n = k_ + 1; } } } while (0)
extern "C" int vq_debug_stamps_wgrad(long long* out, int max_n) {
  int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_vq_wstamp_n), sizeof(int)) != hipSuccess) return -1;
  if (n > max_n) n = max_n;
  if (n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vq_wstamps), sizeof(long long) * n) != hipSuccess) return -1;
  const int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_vq_wstamp_n), &zero, sizeof(int)) != hipSuccess) return -1;
  return n;
}
#else
#define VQ_WSTAMP(id) ((void)0)
#endif
// partial-slab store (a streaming form was measured equal and removed: profiles/r2u_wgrad_nt_micro.txt)
// (Round 3 also measured the MFMAs with their operands swapped — D^T, so that a lane's accumulator quad is four consecutive cin of
// one cout and the slab is written with 24 sixteen-byte stores per lane instead of 96 four-byte ones: 1-5 % SLOWER on every layer and
// in the step, profiles/r3j_wgrad_store16_micro.txt — a four-byte store instruction of this layout fills two whole 128-byte lines,
// a sixteen-byte one touches 32 lines with 32 bytes each.  Removed; it last existed in commit "weight gradients: MFMA operands swapped".)
__device__ __forceinline__ void wg_store(const WgradParams&, float* dst, float v) { *dst = v; }
// The scale of VqConvDesc.alpha / alpha_dev (the inverse loss scale of a VQ_F16 dY) is applied by the reduce kernels, once per
// output element, not by the split-K blocks.

template <int DT, int SPLIT, int BT, int BKP, int NBUF>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  // BT x BT output tile (cout x cin) per tap; 4 waves as 2x2, each (BT/2)x(BT/2)
  constexpr int WT = BT / 2, FR = WT / 32;
  constexpr int OP = DT == VQ_F16 ? VQ_F16 : VQ_BF16;   // MFMA operand type
  constexpr int PLANES = (SPLIT == 6) ? 3 : (SPLIT == 3) ? 2 : 1;   // bf16 pieces per fp32 operand (conv_igemm.hip: split modes)
  constexpr int RSTR = BT + 32;                 // row stride in elements (BT*2 + 64 bytes)
  constexpr int TILE = BKP * RSTR;              // one operand tile, one plane
  constexpr int SLOTS = BT / 8;                 // 16-byte slots per row
  constexpr int RPP = 256 / SLOTS;
  constexpr int PASS = BKP / RPP;
  static_assert(BKP % RPP == 0, "");

  __shared__ __attribute__((aligned(16))) vq_bf16 lds[NBUF * 2 * PLANES * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = (wave >> 1) * WT, wci = (wave & 1) * WT;

  int t = blockIdx.x;
  const int tap = t % p.RS; t /= p.RS;
  const int cit = t % p.n_cit; const int ct = t / p.n_cit;
  const int co0 = ct * BT, ci0 = cit * BT;
  const int kr = tap / p.d.S, ks = tap - kr * p.d.S;
  const int split = blockIdx.y;
  const int pbeg = split * p.pix_per_split;
  int pend = pbeg + p.pix_per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (pend - pbeg + BKP - 1) / BKP : 0;

  const int slot = tid % SLOTS, lrow = tid / SLOTS;
  const bool co_ok = (co0 + slot * 8) < p.d.Cout;
  const bool ci_ok = (ci0 + slot * 8) < p.d.Cin;
  const int dmask = (1 << p.dsh) - 1;
  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;

  // per-row pixel walkers
  int pm[PASS], pn[PASS], poy[PASS], pox[PASS];
#pragma unroll
  for (int i = 0; i < PASS; ++i) {
    const int m = pbeg + lrow + i * RPP;
    pm[i] = m;
    const int n = m / p.HoWo, rem = m - n * p.HoWo;
    pn[i] = n; poy[i] = rem / p.d.Wo; pox[i] = rem - poy[i] * p.d.Wo;
  }

  typedef Store<DT> St;
  float yv[PASS][8], xv[PASS][8];

  auto issue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const bool row_ok = pm[i] < pend;
#pragma unroll
      for (int e = 0; e < 8; ++e) { yv[i][e] = 0.f; xv[i][e] = 0.f; }
      if (row_ok && co_ok) St::load8(p.dy, (int64_t)pm[i] * p.d.Cout + co0 + slot * 8, yv[i]);
      int vy = poy[i] * p.d.stride - p.d.pad_t + kr, vx = pox[i] * p.d.stride - p.d.pad_l + ks;
      bool ok = row_ok && ci_ok && vy >= 0 && vx >= 0 && ((vy & dmask) == 0) && ((vx & dmask) == 0);
      vy >>= p.dsh; vx >>= p.dsh;
      ok = ok && vy < Hv && vx < Wv;
      const int iy = vy >> p.ush, ix = vx >> p.ush;
      if (ok) St::load8(p.x, ((int64_t)(pn[i] * p.d.H + iy) * p.d.W + ix) * p.d.Cin + ci0 + slot * 8, xv[i]);
      // advance walker by one chunk
      pm[i] += BKP; pox[i] += BKP;
      while (pox[i] >= p.d.Wo) { pox[i] -= p.d.Wo; poy[i]++; }
      while (poy[i] >= p.d.Ho) { poy[i] -= p.d.Ho; pn[i]++; }
    }
  };

  auto pack8 = [](const float (&v)[8], vq_u4 (&pc)[PLANES]) {
    vq_bf16 h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = f2op<OP>(v[e]);   // (binary16 storage: an exact round trip)
    pc[0].x = h[0] | ((unsigned)h[1] << 16); pc[0].y = h[2] | ((unsigned)h[3] << 16);
    pc[0].z = h[4] | ((unsigned)h[5] << 16); pc[0].w = h[6] | ((unsigned)h[7] << 16);
    if constexpr (PLANES >= 2) {
      float r1[8];
      vq_bf16 m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { r1[e] = v[e] - bf2f(h[e]); m[e] = f2bf(r1[e]); }
      pc[1].x = m[0] | ((unsigned)m[1] << 16); pc[1].y = m[2] | ((unsigned)m[3] << 16);
      pc[1].z = m[4] | ((unsigned)m[5] << 16); pc[1].w = m[6] | ((unsigned)m[7] << 16);
      if constexpr (PLANES == 3) {
        pc[2].x = pack_bf2(r1[0] - bf2f(m[0]), r1[1] - bf2f(m[1]));
        pc[2].y = pack_bf2(r1[2] - bf2f(m[2]), r1[3] - bf2f(m[3]));
        pc[2].z = pack_bf2(r1[4] - bf2f(m[4]), r1[5] - bf2f(m[5]));
        pc[2].w = pack_bf2(r1[6] - bf2f(m[6]), r1[7] - bf2f(m[7]));
      }
    }
  };

  auto store_lds = [&](int buf) {
    vq_bf16* base = lds + buf * 2 * PLANES * TILE;
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int row = lrow + i * RPP;
      vq_u4 pc[PLANES];
      pack8(yv[i], pc);
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) *(vq_u4*)(base + pl * TILE + row * RSTR + slot * 8) = pc[pl];
      pack8(xv[i], pc);
#pragma unroll
      for (int pl = 0; pl < PLANES; ++pl) *(vq_u4*)(base + (PLANES + pl) * TILE + row * RSTR + slot * 8) = pc[pl];
    }
  };

  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // transposed fragment: lanes of 16-lane group gg = lane>>4 cover channels (gg&1)*16 + 0..15
  // and pixels 8*(gg>>1) + 0..7 (two tr reads of 4 pixels each).
  const int gg = lane >> 4, tl = lane & 15;
  const int frag_row = 8 * (gg >> 1) + (tl >> 2);
  const int frag_col = (gg & 1) * 16 + (tl & 3) * 4;
  auto read_frag = [&](const vq_bf16* tile, int kk, int chan0) -> s16x8 {
    const vq_bf16* ptr = tile + (kk * 16 + frag_row) * RSTR + chan0 + frag_col;
    s16x4 lo4 = lds_read_tr16_b64((const short*)ptr);
    s16x4 hi4 = lds_read_tr16_b64((const short*)(ptr + 4 * RSTR));
    s16x8 r;
    r[0] = lo4[0]; r[1] = lo4[1]; r[2] = lo4[2]; r[3] = lo4[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  };

  auto compute = [&](int buf) {
    const vq_bf16* ybase = lds + buf * 2 * PLANES * TILE;
    const vq_bf16* xbase = ybase + PLANES * TILE;
#pragma unroll
    for (int kk = 0; kk < BKP / 16; ++kk) {
      s16x8 af[PLANES][FR], bf[PLANES][FR];
#pragma unroll
      for (int a = 0; a < FR; ++a)
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
          af[pl][a] = read_frag(ybase + pl * TILE, kk, wco + a * 32);
          bf[pl][a] = read_frag(xbase + pl * TILE, kk, wci + a * 32);
        }
#pragma unroll
      for (int a = 0; a < FR; ++a)
#pragma unroll
        for (int b = 0; b < FR; ++b) {
#pragma unroll
          for (int sum = PLANES - 1; sum >= 1; --sum)       // smallest terms first (plane 0 = hi), cf. conv_igemm_kernel
#pragma unroll
            for (int pa = sum; pa >= 0; --pa) acc[a][b] = mfma_32x32x16_bf16(af[pa][a], bf[sum - pa][b], acc[a][b]);
          acc[a][b] = mfma16<OP>(af[0][a], bf[0][b], acc[a][b]);
        }
    }
  };

  if (nchunks > 0) {
    issue_loads();
    store_lds(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const bool more = (c + 1) < nchunks;
      if constexpr (NBUF == 2) {
        if (more) issue_loads();
        compute(c & 1);
        if (more) store_lds((c + 1) & 1);
        __syncthreads();
      } else {
        compute(0);
        __syncthreads();
        if (more) { issue_loads(); store_lds(0); }
        __syncthreads();
      }
    }
  }

  // partial tile -> workspace [split][tap][Cout][Cin]
  float* out = p.part + ((int64_t)(split * p.RS + tap) * p.d.Cout) * p.d.Cin;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b) {
      const int ci = ci0 + wci + b * 32 + fr;
      if (ci >= p.d.Cin) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wco + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (co < p.d.Cout) out[(int64_t)co * p.d.Cin + ci] = acc[a][b][e];
      }
    }
}

// ------------------------------------------------------------------------------ LDS-DMA variant
// bf16 storage, Cout % BT == 0 and Cin % BT == 0: both [64 pixels][BT channels] tiles are copied
// HBM -> LDS with global_load_lds (lane-linear 1-KiB pieces, double-buffered, one barrier per
// 64-pixel chunk).  Bank conflicts of the transposed fragment reads are removed by XOR-ing the
// 64-byte segment index with the pixel row on the SOURCE address (4 consecutive pixel rows of a
// ds_read_b64_tr_b16 lane group then cover the 256-byte bank row exactly once).
__device__ __attribute__((aligned(256))) unsigned int g_vq_wg_zero_page[128];

// BT x BT (cout x cin) tile per tap; NW waves: 4 = 2x2 (BT 64/128), 8 = 2(cout) x 4(cin) for BT = 256
// (per-wave 128 x 64, 128 KiB LDS) — the large tile halves both the L2->LDS bytes and the address math per MFMA.
// X3 = 1 (round 6, BT >= 128): VQ_F16X2 operands in the native three-product form — see conv_wgrad3_kernel: fragment 2 i + plane of a
// wave's operand side is the hi / lo piece of its i-th block of 32 REAL channels (64 virtual ones), hi*hi + hi*lo + lo*hi into one
// accumulator per pair of real blocks, slabs and bias partials indexed by real channels.
template <int DT, int BT, int NW, int X3 = 0>
__global__ __launch_bounds__(NW * 64) void conv_wgrad_glds_kernel(const WgradParams p) {
  static_assert(!X3 || (BT >= 128 && DT == VQ_F16), "native three-product form: 128 / 256 virtual channels per side, binary16 pieces");
  constexpr int BKP = 64;                       // pixels per chunk
  constexpr int NWI = NW / 2;                   // waves along cin
  constexpr int WTC = BT / 2, WTI = BT / NWI;   // per-wave tile
  constexpr int FRC = WTC / 32, FRI = WTI / 32;
  constexpr int RB = BT * 2;                    // row bytes (128 / 256 / 512)
  constexpr int SPR = RB / 16;                  // 16-byte slots per row
  constexpr int RPP = 1024 / RB;                // rows per 1-KiB DMA piece (8 / 4 / 2)
  constexpr int TILE = BKP * BT;                // elements per operand tile
  constexpr int NPC = (BKP / RPP) / NW;         // pieces per wave per operand per chunk
  static_assert(NPC >= 1 && NPC * NW * RPP == BKP && FRC >= 1 && FRI >= 1, "tile / wave mismatch");
  // swizzle key of a pixel row: 4 consecutive rows of a transposed-read lane group must land in 4 different
  // 64-byte segments of one 256-byte bank row
  auto seg_key = [](int row) -> int { return RB >= 256 ? (row & 3) : ((row >> 1) & 1); };

  VQ_DYN_LDS(vq_bf16, lds);                     // 2 stages x {dY tile, X tile}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = (wave / NWI) * WTC, wci = (wave % NWI) * WTI;
  // XCD-aware decode (1-D grid of 8 * ceil(nsplit/8) * tiles blocks): block b runs on XCD b % 8 (observed
  // dispatch rule, speed only).  All tiles — in particular the R*S taps — of one pixel split are given to ONE
  // XCD and are adjacent in its dispatch order, so the dY / X chunks they all stream are fetched from HBM once
  // and then hit in that XCD's L2 (without this the C=128 layers re-read both tensors 9x from HBM).
  const int tiles = p.RS * p.n_cit * p.n_ct;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  int split, t;
  if (p.xcd_tiles == 2) {                        // range-owning XCDs (see conv_wgrad3_kernel): any split count
    const int per = (int)(gridDim.x >> 3), idx = xcd * per + jb;
    if (idx >= p.nsplit * tiles) return;
    split = idx / tiles;
    t = idx - split * tiles;
  } else {
    split = (jb / tiles) * 8 + xcd;
    t = jb % tiles;
  }
  if (split >= p.nsplit) return;                 // whole block exits: no barrier has been reached yet
  const int tap = t % p.RS; t /= p.RS;
  const int cit = t % p.n_cit; const int ct = t / p.n_cit;
  const int co0 = ct * BT, ci0 = cit * BT;
  const int kr = tap / p.d.S, ks = tap - kr * p.d.S;
  const int pbeg = split * p.pix_per_split;
  int pend = pbeg + p.pix_per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (pend - pbeg + BKP - 1) / BKP : 0;

  const int lrow = lane / SPR, lp = lane % SPR;
  const int dmask = (1 << p.dsh) - 1;
  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_wg_zero_page;
  const vq_bf16* dyb = (const vq_bf16*)p.dy;
  const vq_bf16* xb = (const vq_bf16*)p.x;

  // Every lane owns one pixel row per piece.  Host guarantees (see vq_conv2d_wgrad): Ho, Wo are powers of
  // two and M % 64 == 0, so (n, oy, ox) of pixel m are shifts/masks, all dY rows are in range and the dY
  // pointer is linear in the chunk index; the gather pointer is a branch-free select with a zero page.
  const int wsh = p.wo_shift, hsh = p.ho_shift;
  const int wmask = p.d.Wo - 1, hmask = p.d.Ho - 1;
  const int Hvd = Hv << p.dsh, Wvd = Wv << p.dsh, sh_y = p.dsh + p.ush;
  int pm[NPC], lsl[NPC];
  const vq_bf16* pdy[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int row = (wave * NPC + i) * RPP + lrow;     // row inside the 64-pixel chunk
    const int seg = (lp >> 2) ^ seg_key(row);
    lsl[i] = ((seg << 2) | ((lp & 3) ^ (X3 ? (seg & 1) : 0))) << 3;   // logical element offset this lane fetches (X3: the 16-byte granules
                                                                     // of odd segments swapped: bank-conflict-free hi / lo reads)
    pm[i] = pbeg + row;
    pdy[i] = dyb + (int64_t)pm[i] * p.d.Cout + co0 + lsl[i];
  }
  const int64_t dy_step = (int64_t)BKP * p.d.Cout;

  auto stage = [&](int buf) {
    vq_bf16* ybase = lds + buf * 2 * TILE;
    vq_bf16* xbase = ybase + TILE;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      glds16(pdy[i], ybase + (wave * NPC + i) * RPP * BT);
      pdy[i] += dy_step;
      const int m = pm[i];
      const int ox = m & wmask, oy = (m >> wsh) & hmask, n = m >> (wsh + hsh);
      const int vy = oy * p.d.stride - p.d.pad_t + kr, vx = ox * p.d.stride - p.d.pad_l + ks;
      const int ok = (int)((unsigned)vy < (unsigned)Hvd) & (int)((unsigned)vx < (unsigned)Wvd) & (int)(((vy | vx) & dmask) == 0);
      const int iy = vy >> sh_y, ix = vx >> sh_y;
      const int64_t off = (int64_t)((n * p.d.H + iy) * p.d.W + ix) * p.d.Cin + ci0 + lsl[i];
      const uintptr_t a_ok = (uintptr_t)(xb + off), a_zero = (uintptr_t)(zero + lsl[i]);
      glds16((const void*)(ok ? a_ok : a_zero), xbase + (wave * NPC + i) * RPP * BT);
      pm[i] = m + BKP;
    }
  };

  constexpr int AFR = X3 ? FRC / 2 : FRC, BFR = X3 ? FRI / 2 : FRI;   // accumulator blocks (X3: blocks of 32 real channels)
  static_assert(AFR >= 1 && BFR >= 1, "native form: at least 64 virtual channels per wave and side");
  f32x16 acc[AFR][BFR];
#pragma unroll
  for (int a = 0; a < AFR; ++a)
#pragma unroll
    for (int b = 0; b < BFR; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // Bias gradient = dY^T * 1: one extra MFMA per k-step in the first FRC blocks of a cout tile's (tap, cin tile)
  // list, block i taking the cout fragment i of its waves — spread out so that no block carries more than one extra
  // accumulator (host: bias_part is null when R*S*n_cit < FRC, the column-sum kernel does the bias then).
  const int bias_frag = tap * p.n_cit + cit;
  const bool do_bias = p.bias_part != nullptr && bias_frag < AFR;
  f32x16 bacc;
  s16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)One16<DT>::BITS;   // 1.0 in the operand type
#pragma unroll
  for (int e = 0; e < 16; ++e) bacc[e] = 0.f;

  // Transposed fragment reads (ds_read_b64_tr_b16, untracked: see lds_read_tr16_b64_async).  A 32-channel fragment at
  // k-step kk is two 8-byte reads per lane, at pixel rows r0 = kk*16 + frow and r0 + 4.  seg_key(r0) does not depend
  // on kk (16 % 4 == 0) and is the same for r0 + 4, so a fragment is ONE lane address plus immediates.
  const int gg = lane >> 4, tl = lane & 15;
  const int frow = 8 * (gg >> 1) + (tl >> 2);            // pixel row of the first transposed read
  const int fcol = (gg & 1) * 16 + (tl & 3) * 4;         // channel offset inside the 32-wide fragment
  auto frag_addr = [&](int chan0) -> int {               // byte offset inside a tile of (row frow, channel chan0 + fcol)
    const int c = chan0 + fcol;
    const int seg = (c * 2) >> 6, within = (c * 2) & 63;
    return frow * RB + ((seg ^ seg_key(frow)) << 6) + within;
  };
  // X3: the hi (plane 0) / lo (plane 1) piece of the 32 real channels that start at virtual channel chan0 (a multiple of 64)
  auto frag_addr_x3 = [&](int chan0, int plane) -> int {
    const int c = chan0 + ((fcol >> 3) << 4) + (fcol & 7);
    const int seg = (c * 2) >> 6, within = ((c * 2) & 63) ^ ((seg & 1) << 4);
    return (frow * RB + ((seg ^ seg_key(frow)) << 6) + within) ^ (plane << 4);
  };
  int ya[FRC], xa[FRI];
#pragma unroll
  for (int a = 0; a < FRC; ++a) ya[a] = X3 ? frag_addr_x3(wco + (a >> 1) * 64, a & 1) : frag_addr(wco + a * 32);
#pragma unroll
  for (int b = 0; b < FRI; ++b) xa[b] = TILE * 2 + (X3 ? frag_addr_x3(wci + (b >> 1) * 64, b & 1) : frag_addr(wci + b * 32));
  constexpr int NRD = 2 * (FRC + FRI);                   // LDS reads per k-step per wave
  static_assert(NRD <= 15, "lgkmcnt is a 4-bit counter");

  // k-step kk+1's fragments are requested before the MFMAs of k-step kk; the counted lgkmcnt wait leaves exactly
  // those NRD reads in flight.  `bias_tag` (block-uniform) compiles the bias MFMAs in or out of the loop.
  constexpr bool DB = BT < 256;                          // BT = 256: 128 accumulators leave no room for a second fragment set
  auto run = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    s16x4 fy[DB ? 2 : 1][FRC][2], fx[DB ? 2 : 1][FRI][2];
    auto issue = [&](const char* base, auto kk_tag, int slot) {
      constexpr int KOFF = decltype(kk_tag)::value * 16 * RB;
#pragma unroll
      for (int a = 0; a < FRC; ++a) {
        fy[slot][a][0] = lds_read_tr16_b64_async<KOFF>(base + ya[a]);
        fy[slot][a][1] = lds_read_tr16_b64_async<KOFF + 4 * RB>(base + ya[a]);
      }
#pragma unroll
      for (int b = 0; b < FRI; ++b) {
        fx[slot][b][0] = lds_read_tr16_b64_async<KOFF>(base + xa[b]);
        fx[slot][b][1] = lds_read_tr16_b64_async<KOFF + 4 * RB>(base + xa[b]);
      }
    };
    auto mma = [&](int slot) {
#pragma unroll
      for (int a = 0; a < FRC; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h) vq_tie(fy[slot][a][h]);
#pragma unroll
      for (int b = 0; b < FRI; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) vq_tie(fx[slot][b][h]);
      s16x8 af[FRC], bfr[FRI];
#pragma unroll
      for (int a = 0; a < FRC; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[a][e] = fy[slot][a][0][e]; af[a][4 + e] = fy[slot][a][1][e]; }
#pragma unroll
      for (int b = 0; b < FRI; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) { bfr[b][e] = fx[slot][b][0][e]; bfr[b][4 + e] = fx[slot][b][1][e]; }
      if constexpr (X3) {
#pragma unroll
        for (int a = 0; a < AFR; ++a)
#pragma unroll
          for (int b = 0; b < BFR; ++b) {                // hi x hi + hi x lo + lo x hi
            acc[a][b] = mfma16<DT>(af[2 * a], bfr[2 * b], acc[a][b]);
            acc[a][b] = mfma16<DT>(af[2 * a], bfr[2 * b + 1], acc[a][b]);
            acc[a][b] = mfma16<DT>(af[2 * a + 1], bfr[2 * b], acc[a][b]);
          }
        if constexpr (BIAS) {
          s16x8 selh = af[0], sell = af[1];
#pragma unroll
          for (int a = 1; a < AFR; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) { selh[e] = (bias_frag == a) ? af[2 * a][e] : selh[e]; sell[e] = (bias_frag == a) ? af[2 * a + 1][e] : sell[e]; }
          bacc = mfma16<DT>(selh, ones, bacc);
          bacc = mfma16<DT>(sell, ones, bacc);
        }
      } else {
#pragma unroll
        for (int a = 0; a < FRC; ++a)
#pragma unroll
          for (int b = 0; b < FRI; ++b) acc[a][b] = mfma16<DT>(af[a], bfr[b], acc[a][b]);
        if constexpr (BIAS) {
          s16x8 sel = af[0];
#pragma unroll
          for (int a = 1; a < FRC; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) sel[e] = (bias_frag == a) ? af[a][e] : sel[e];
          bacc = mfma16<DT>(sel, ones, bacc);
        }
      }
    };
    stage(0);
    wait_vmcnt<0>();
    raw_barrier();
    for (int c = 0; c < nchunks; ++c) {
      const char* base = (const char*)(lds + (c & 1) * 2 * TILE);
      issue(base, std::integral_constant<int, 0>{}, 0);
#ifdef VQ_ABLATION_KERNELS
      if (c + 1 < nchunks && !(p.dbg & 1)) stage((c + 1) & 1);   // ablation: no DMA after the first chunk (wrong results)
#else
      if (c + 1 < nchunks) stage((c + 1) & 1);           // next chunk's DMA flies under this chunk's MFMAs
#endif
      if constexpr (DB) {
        issue(base, std::integral_constant<int, 1>{}, 1);
        wait_lgkmcnt<NRD>(); mma(0);
        issue(base, std::integral_constant<int, 2>{}, 0);
        wait_lgkmcnt<NRD>(); mma(1);
        issue(base, std::integral_constant<int, 3>{}, 1);
        wait_lgkmcnt<NRD>(); mma(0);
        wait_lgkmcnt<0>(); mma(1);
      } else {
        wait_lgkmcnt<0>(); mma(0);
        issue(base, std::integral_constant<int, 1>{}, 0);
        wait_lgkmcnt<0>(); mma(0);
        issue(base, std::integral_constant<int, 2>{}, 0);
        wait_lgkmcnt<0>(); mma(0);
        issue(base, std::integral_constant<int, 3>{}, 0);
        wait_lgkmcnt<0>(); mma(0);
      }
      wait_vmcnt<0>();
      raw_barrier();
    }
  };
  if (nchunks > 0) {
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});
  }

  const int fr = lane & 31, fh = lane >> 5;
  const int CoutS = X3 ? p.d.Cout / 2 : p.d.Cout, CinS = X3 ? p.d.Cin / 2 : p.d.Cin;       // slab extents (X3: real channels)
  const int cob = X3 ? (co0 + wco) / 2 : co0 + wco, cib = X3 ? (ci0 + wci) / 2 : ci0 + wci;
  if (do_bias && (wave % NWI) == 0 && fr == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      p.bias_part[(int64_t)split * CoutS + cob + bias_frag * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh] = bacc[e];
  }
  float* out = p.part + ((int64_t)(split * p.RS + tap) * CoutS) * CinS;
#pragma unroll
  for (int a = 0; a < AFR; ++a)
#pragma unroll
    for (int b = 0; b < BFR; ++b) {
      const int ci = cib + b * 32 + fr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = cob + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        wg_store(p, &out[(int64_t)co * CinS + ci], acc[a][b][e]);
      }
    }
}

// ------------------------------------------------------------------------------ three taps per block
// 3x3 / stride 1 / pad 1 convolutions (every ResnetBlock conv): one block owns a 128x128 (cout x cin) tile for the
// THREE taps of one kernel row.  Ablation (profiles/r1_wgrad_ablation_v20.txt): without the tile DMA and its address
// math the one-tap kernel runs at 1.0-1.3 PFLOP/s instead of 0.64-0.74, so the dY tile is staged once per three
// taps and the X tile once with a halo — 64 pixels + one extra column either side of every image-row segment of the
// chunk (up to 96 rows of LDS: segments of 4..64 pixels) — the taps being row offsets 0 / +1 / +2 into it, and the dY fragments
// are read from LDS once per three taps.  Fragment reads are software-pipelined one (k-step, tap) ahead with counted lgkmcnt waits.
// 8 waves as 2 (cout) x 4 (cin): 64 x 32 per wave and tap, 96 accumulator registers, 166-170 VGPRs in all (226 before round 3), ONE
// block per CU (the 68 KiB of LDS would admit two; 8 waves x 2 blocks need <= 128 registers).  Round 3 took the lane-invariant work out
// of the loop: the segment shift is a template parameter (halo slot <-> segment maps and the fragments' row offsets are immediates,
// four swizzled lane addresses instead of 24), one dY pointer + a uniform stride for the wave's pieces, 32-bit input offsets —
// +10-14 % on every layer (profiles/r3a_wgrad4_micro.txt: 128 ch @256^2 894 -> 1016 TFLOP/s, 512 ch @32^2 806 -> 931).
// A 4-wave form (2 x 2 waves of 64 x 64 per tap: 0.67 instead of 0.83 transposed fragments per MFMA, two blocks per CU, bias
// gradient by packed dot products) was built on the same skeleton and LOST 30 % (718 vs 1016 TFLOP/s, same file): 192 accumulators
// leave ~40 registers for everything else, hipcc spills the fragment addresses (156-236 B of scratch, 7 reloads per chunk in the
// loop) and the reloads share vmcnt with the tile DMA.  Removed; it last existed in commit b58ce8b.
// GEN = 0: power-of-two output extents with rows of >= 16 pixels (shift/mask pixel decode, 72 halo rows);  GEN = 1: any extent
// whose rows are a multiple of 4 pixels (crop-invariance batches: division decode, up to 96 halo rows).
// (A three-buffer ring with the staging pieces issued between the MFMA steps, and a two-buffer form with 32-bit halo addresses, were
// measured -5 % / +-0 in round 2 and removed: profiles/r2n_wgrad3_forms_micro.txt; they last existed in commit 2da2460.)
// SEG (GEN = 0 only): log2 of the image-row segment length (4, 5 or 6: rows of 16, 32, >= 64 pixels) as a compile-time constant —
// the halo slot <-> segment maps and the fragment row offsets become immediates instead of registers (the 4-wave form has none to spare).
// STAG = 1 (round 3): the two waves of a SIMD (w and w + 4) issue their share of the next chunk's staging — five LDS-DMA pieces
// and ~100 VALU instructions of address arithmetic, during which a wave feeds no MFMA — at DIFFERENT times: waves 0-3 right after
// the chunk barrier, waves 4-7 half a chunk later.  Unstaggered, both waves of every SIMD ran that burst together, straight after
// the barrier that had just drained their fragment pipelines: the matrix pipe idled through all of it, once per 24 MFMAs.
// (A three-buffer ring with the barrier moved to mid-chunk — so that the fragment pipeline never drains — was also measured in
// round 3: -2 % per layer, +0.3 % on the step, profiles/r3b_wgrad_ring_micro.txt; removed, it last existed in commit e5a0b3f.
// So was requesting the fragments two (X) / three (dY) steps ahead of their MFMAs instead of one: -0.7 % wgrad.frac in the step
// (0.3440 vs 0.3468, both repeats), profiles/r3d_bench_ab.txt; removed.  Back-to-back micro-benchmarks rank these variants the
// other way round (the plain rounds-1-2 form is fastest there); the step, with other kernels' tails and a colder L2 between
// launches, is what is optimised.)
// (Round 5: ONE pixel window per k-step instead of one X fragment per (k-step, tap) — a lane's operand for tap ks is pixels ks .. ks + 7 of
// a 12-pixel window, three transposed reads (rows +0, +4, +8) and four v_alignbit_b32 for tap 1; X fragment reads per k-step 6 -> 3, all
// LDS reads of a chunk 40 -> 28 per wave, the reads of a k-step requested one k-step ahead.  By the byte count the LDS data path is as busy
// as the matrix pipe in this kernel (8 waves x 40 x 512 B + 34 KiB of DMA per 24 MFMAs per wave against 128 B / cycle — round 6: the guide gives
// ds_read_b64_tr_b16 two LDS cycles per wave-instruction, 256 B / cycle, so the reads are ~40 % of the chunk, not all of it); bit-exact, 174-180
// VGPRs, and 5-8 % SLOWER on every layer in bf16 / fp16 / f16x3 (profiles/r5q_wgrad_window_ab.txt: 128 ch @256^2 974 -> 925 TFLOP/s, 512 ch
// @64^2 1149 -> 1132).  Not the LDS bytes, then; removed, it last existed in commit dcb5e85.)
// X3 = 1 (round 6): VQ_F16X2 operands, hi*hi + hi*lo + lo*hi formed HERE instead of on the virtual 2Cout x 2Cin problem.  The tiles are
// staged exactly as before (128 virtual binary16 channels = 64 real ones per side, [h0..7 l0..7 h8..15 l8..15 ...]); a transposed
// fragment read picks its 32 channels per lane address, so a fragment can be the hi (or the lo) piece of 32 REAL channels — virtual
// channel 16 (r >> 3) + (r & 7) [+ 8] of real channel r — and the lo fragment is the hi fragment's address XOR 16 bytes.  4 waves as
// 2 x 2, each 32 x 32 real channels per tap: the SAME fragment reads as the 4-wave virtual form (two per operand side: hi and lo
// instead of two 32-channel blocks), three MFMAs into ONE accumulator instead of four into four — per real MAC 0.75x the MFMAs and
// 0.8x the LDS reads of the 8-wave virtual form, 48 accumulator registers (two blocks per CU), slabs a quarter the size, and the
// ordinary split reduction instead of the quadrant sum (wgrad_reduce_x2_kernel).
template <int DT, int GEN, int NW, int SEG = 0, int STAG = 0, int X3 = 0>
__global__ __launch_bounds__(NW * 64, 2) void conv_wgrad3_kernel(const WgradParams p) {
  constexpr int BT = 128, BKP = 64, RB = BT * 2, XROWS = GEN ? 96 : 72;   // 64 pixels + 2 halo columns per row segment
  static_assert(GEN ? SEG == 0 : (SEG >= 4 && SEG <= 6), "SEG: compile-time segment shift of the power-of-two form");
  static_assert(!X3 || (NW == 4 && DT == VQ_F16), "the native three-product form: 2 x 2 waves over binary16 pieces");
  constexpr int TILE_Y = BKP * BT, TILE_X = XROWS * BT, STAGE = TILE_Y + TILE_X;   // elements
  constexpr int NWI = NW / 2, WTI = BT / NWI;       // waves along cin, cin channels per wave (32 or 64)
  constexpr int FRC = 2, FRI = WTI / 32;            // X3: the two fragments per side are the hi and the lo piece of the same 32 real channels
  constexpr int AFR = X3 ? 1 : FRC, BFR = X3 ? 1 : FRI;   // accumulator blocks per tap
  constexpr int BXOR = X3 ? 4 : 6;                  // second fragment of a side = first one's address XOR (1 << BXOR): +32 channels, or the lo piece
  static_assert(!X3 || FRI == 2, "native form: 64 virtual = 32 real channels per wave and side");
  constexpr int YP = (BKP / 4) / NW;                // dY pieces (4 rows = 1 KiB) per wave per chunk
  constexpr int XPT = XROWS / 4, XP = (XPT + NW - 1) / NW;   // X halo pieces per chunk / per wave
  static_assert(NW == 4 || NW == 8, "2 x 2 or 2 x 4 waves");
  VQ_DYN_LDS(vq_bf16, lds);                     // 2 x {dY [64][128], X [XROWS][128]}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  VQ_WSTAMP(20);
  const int wco = (wave / NWI) * 64, wci = (wave % NWI) * WTI;
  const int tiles = 3 * p.n_cit * p.n_ct;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  // default: all tiles of a pixel split on one XCD (see conv_wgrad_glds_kernel) — needs a multiple of 8 splits to keep the XCDs
  // balanced.  xcd_tiles (tiles % 8 == 0, short reductions): an XCD owns tiles / 8 tiles with ALL their splits, so any split
  // count fills the chip evenly — 512 channels at 32x32 (48 tiles): 5 splits = 240 blocks in one round instead of 8 = 384 in 1.5
  int split, t;
  if (p.xcd_tiles == 2) {
    // any tile count: the split-major list of (split, tile) items is cut in 8 contiguous ranges, one per XCD (the grid is padded
    // to a multiple of 8): consecutive splits stay together on an XCD, at most one split straddles a range boundary
    const int per = (int)(gridDim.x >> 3), idx = xcd * per + jb;
    if (idx >= p.nsplit * tiles) return;
    split = idx / tiles;
    t = idx - split * tiles;
  } else if (p.xcd_tiles) {
    const int tpx = tiles >> 3;
    t = xcd * tpx + jb % tpx;
    split = jb / tpx;
  } else {
    split = (jb / tiles) * 8 + xcd;
    t = jb % tiles;
  }
  if (split >= p.nsplit) return;
  const int kr = t % 3; t /= 3;
  const int cit = t % p.n_cit; const int ct = t / p.n_cit;
  const int co0 = ct * BT, ci0 = cit * BT;
  const int pbeg = split * p.pix_per_split;
  int pend = pbeg + p.pix_per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (pend - pbeg + BKP - 1) / BKP : 0;

  const int W = p.d.Wo, H = p.d.Ho;               // output extent = extent of the (nearest-2x upsampled, if up == 2) input
  const int wsh = p.wo_shift, hsh = p.ho_shift, wmask = W - 1, hmask = H - 1;
  constexpr bool pow2 = !GEN;                   // else (crop-invariance batches, e.g. 208x272): decode pixels by division
  const int segsh = GEN ? p.seg_shift : SEG;    // log2 of the image-row segment length inside a 64-pixel chunk: 2^segsh | Wo
  const int wseg = 1 << segsh, nslots = (BKP >> segsh) * (wseg + 2);
  const vq_bf16* zero = (const vq_bf16*)g_vq_wg_zero_page;
  const vq_bf16* dyb = (const vq_bf16*)p.dy;
  const vq_bf16* xb = (const vq_bf16*)p.x;

  // ---- staging: 1-KiB pieces of 4 rows; lane -> (row lrow of the piece, 16-byte slot lp) ------------------------
  const int lrow = lane >> 4, lp = lane & 15;
  // X3: the 16-byte granules of the ODD 64-byte segments are stored pairwise swapped (granule ^ 1).  A hi (or lo) fragment read takes
  // granules {0, 2} (or {1, 3}) of an even segment in one half of its lanes and of the odd segment next to it in the other half: stored
  // alike they meet in the same banks — the SQ pass of the first native form showed half of its LDS cycles as bank conflicts
  // (profiles/r6z_sq_summary.json) — swapped, the two halves of a read cover the four granules of every 64-byte quarter once.
  auto lsl_of = [&](int row) -> int {
    const int seg = (lp >> 2) ^ (row & 3), gr = (lp & 3) ^ (X3 ? (seg & 1) : 0);
    return ((seg << 2) | gr) << 3;
  };
  // dY piece i of this wave = rows (wave + NW i) * 4 + lrow: 4 NW rows apart, so one pointer + a uniform stride serves them all
  // (the swizzle key row & 3 is the same for every i)
  const vq_bf16* pdy0 = dyb + (int64_t)(pbeg + wave * 4 + lrow) * p.d.Cout + co0 + lsl_of(lrow);
  const int dy_piece = 4 * NW * p.d.Cout;        // elements between consecutive pieces of a wave
  // halo slot of this lane in X piece (wave + NW i) -> (segment q, column jj): kept in registers by the general form, re-derived
  // per chunk (a division by the compile-time segment length) by the power-of-two form
  int xqj[GEN ? XP : 1];                         // (q << 8) | jj, -1 = no such slot
  const int xlsl = lsl_of(lrow);                 // its swizzled 16-byte slot (slots are 4 NW apart: the same key for every i)
  if constexpr (GEN) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int slot = (wave + NW * i) * 4 + lrow;
      const int q = slot / (wseg + 2);
      xqj[i] = slot < nslots ? ((q << 8) | (slot - q * (wseg + 2))) : -1;
    }
  }
  const int cin0 = ci0 + xlsl;
  int m0 = pbeg;
  auto stage = [&](int buf) {
    vq_bf16* ybase = lds + buf * STAGE;
    vq_bf16* xbase = ybase + TILE_Y;
#pragma unroll
    for (int i = 0; i < YP; ++i) glds16(pdy0 + i * dy_piece, ybase + (wave + NW * i) * 4 * BT);
    pdy0 += (int64_t)BKP * p.d.Cout;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (wave + NW * i < XPT) {                  // wave-uniform
        int q, jj;
        bool have;
        if constexpr (GEN) { q = xqj[i] >> 8; jj = xqj[i] & 255; have = xqj[i] >= 0; }
        else {
          constexpr int WS2 = (1 << SEG) + 2;
          const int slot = (wave + NW * i) * 4 + lrow;
          q = slot / WS2; jj = slot - q * WS2; have = slot < (BKP >> SEG) * WS2;
        }
        const int ms = m0 + (q << segsh);
        int ox0, oy, n;
        if constexpr (pow2) { ox0 = ms & wmask; oy = (ms >> wsh) & hmask; n = ms >> (wsh + hsh); }
        else { n = ms / p.HoWo; const int rem = ms - n * p.HoWo; oy = rem / W; ox0 = rem - oy * W; }
        const int iy = oy + kr - 1, ix = ox0 - 1 + jj;
        const bool ok = have && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        // 32-bit element offset: the launcher checks that the input has fewer than 2^31 elements
        const int off = ((n * p.d.H + (iy >> p.ush)) * p.d.W + (ix >> p.ush)) * p.d.Cin + cin0;
        const vq_bf16* src = ok ? xb + off : zero + xlsl;
        glds16((const void*)src, xbase + (wave + NW * i) * 4 * BT);
      }
    }
    m0 += BKP;
  };

  // ---- fragment addressing (see conv_wgrad_glds_kernel) ----------------------------------------------------------
  const int gg = lane >> 4, tl = lane & 15;
  const int frow = 8 * (gg >> 1) + (tl >> 2);
  const int fcol = (gg & 1) * 16 + (tl & 3) * 4;
  const int fcolv = X3 ? ((fcol >> 3) << 4) + (fcol & 7) : fcol;     // virtual channel offset of this lane's 4 channels inside the wave's block
  int ya[FRC];
#pragma unroll
  for (int a = 0; a < FRC; ++a) {
    const int c = wco + (X3 ? 0 : a * 32) + fcolv, seg = (c * 2) >> 6, within = ((c * 2) & 63) ^ (X3 ? ((seg & 1) << 4) : 0);
    ya[a] = (frow * RB + ((seg ^ (frow & 3)) << 6) + within) ^ (X3 ? (a << 4) : 0);
  }
  // X: pixel row r of the chunk shifted by tap ks lives in halo slot r + 2 * (r >> segsh) + ks.  The second cin fragment of a
  // 64-channel wave (FRI = 2) is 32 channels = one 64-byte segment further: wci is a multiple of 64, so its swizzled segment
  // index differs in bit 0 only — the same address XOR 64 (bit 6 of the byte offset belongs to the segment field alone).
  // General form: one address per (k-step, tap, half).  Power-of-two form (rows of >= 16 pixels): a k-step's 16 rows never
  // straddle a segment, so halo row = frow + U(kk, ks) [+ 4 for the second half] with U = 16 kk + 2 (kk >> (SEG - 4)) + ks a
  // compile-time constant; the swizzle key (frow + U) & 3 takes four values, hence FOUR lane addresses xsw[j] — row frow with key
  // (frow + j) & 3 — and everything else is the instruction's immediate offset U * RB (+ 4 RB).
  int xoff[GEN ? 4 : 1][GEN ? 3 : 1][GEN ? 2 : 1];
  int xsw[GEN ? 1 : 4];
  {
    const int c = wci + fcolv, seg = (c * 2) >> 6, within = ((c * 2) & 63) ^ (X3 ? ((seg & 1) << 4) : 0);
    if constexpr (GEN) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = kk * 16 + frow + 4 * h;
            const int row = r + 2 * (r >> segsh) + ks;
            xoff[kk][ks][h] = TILE_Y * 2 + row * RB + ((seg ^ (row & 3)) << 6) + within;
          }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) xsw[j] = TILE_Y * 2 + frow * RB + ((seg ^ ((frow + j) & 3)) << 6) + within;
    }
  }

  f32x16 acc[3][AFR][BFR];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int a = 0; a < AFR; ++a)
#pragma unroll
      for (int b = 0; b < BFR; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ks][a][b][e] = 0.f;
  const bool do_bias = p.bias_part != nullptr && cit == 0 && kr < AFR;   // blocks (cin tile 0, kernel row r < 2) carry bias fragment r
  // (the bias accumulator and its all-ones operand live inside the BIAS instantiation of the pipeline only)
  auto store_bias = [&](const f32x16& bacc) {
    const int fr_ = lane & 31, fh_ = lane >> 5;
    if ((wave % NWI) == 0 && fr_ == 0) {
      const int cob = X3 ? (co0 + wco) / 2 : co0 + wco + kr * 32, cstride = X3 ? p.d.Cout / 2 : p.d.Cout;     // (X3: real channels)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        p.bias_part[(int64_t)split * cstride + cob + (e & 3) + 8 * (e >> 2) + 4 * fh_] = bacc[e];
    }
  };

  auto run = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    f32x16 bacc;
    s16x8 ones;
    if constexpr (BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) ones[e] = (short)One16<DT>::BITS;
#pragma unroll
      for (int e = 0; e < 16; ++e) bacc[e] = 0.f;
    }
    constexpr int NX = 2;
    s16x4 fy[2][FRC][2], fx[NX][FRI][2];
    auto issue_y = [&](const char* base, auto kk_tag) {
      constexpr int KK = decltype(kk_tag)::value, KOFF = KK * 16 * RB;
#pragma unroll
      for (int a = 0; a < FRC; ++a) {
        fy[KK & 1][a][0] = lds_read_tr16_b64_async<KOFF>(base + ya[a]);
        fy[KK & 1][a][1] = lds_read_tr16_b64_async<KOFF + 4 * RB>(base + ya[a]);
      }
    };
    auto issue_x = [&](const char* base, auto u_tag) {    // the X fragments of step u = kk * 3 + ks
      constexpr int U = decltype(u_tag)::value, KK = U / 3, KS = U % 3;
#pragma unroll
      for (int b = 0; b < FRI; ++b) {
        if constexpr (GEN) {
          fx[U % NX][b][0] = lds_read_tr16_b64_async<0>(base + (xoff[KK][KS][0] ^ (b << BXOR)));
          fx[U % NX][b][1] = lds_read_tr16_b64_async<0>(base + (xoff[KK][KS][1] ^ (b << BXOR)));
        } else {
          constexpr int UR = 16 * KK + 2 * (KK >> (SEG - 4)) + KS;      // halo row offset of this (k-step, tap)
          fx[U % NX][b][0] = lds_read_tr16_b64_async<UR * RB>(base + (xsw[UR & 3] ^ (b << BXOR)));
          fx[U % NX][b][1] = lds_read_tr16_b64_async<(UR + 4) * RB>(base + (xsw[UR & 3] ^ (b << BXOR)));
        }
      }
    };
    auto step = [&](const char* base, auto u_tag) {       // u = kk * 3 + ks: prefetch step u + 1, then the MFMAs of step u
      constexpr int U = decltype(u_tag)::value, KK = U / 3, KS = U % 3;
      constexpr int NU = U + 1, NKK = NU / 3, NKS = NU % 3;
      if constexpr (NU < 12) {
        if constexpr (NKS == 0) issue_y(base, std::integral_constant<int, NKK>{});
        issue_x(base, std::integral_constant<int, NU>{});
        wait_lgkmcnt<(NKS == 0 ? 2 * FRC + 2 * FRI : 2 * FRI)>();
      } else {
        wait_lgkmcnt<0>();
      }
#pragma unroll
      for (int b = 0; b < FRI; ++b) vq_tie(fx[U % NX][b][0], fx[U % NX][b][1]);
      if constexpr (KS == 0) {
#pragma unroll
        for (int a = 0; a < FRC; ++a) vq_tie(fy[KK & 1][a][0], fy[KK & 1][a][1]);
      }
      s16x8 af[FRC], bfr[FRI];
#pragma unroll
      for (int a = 0; a < FRC; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[a][e] = fy[KK & 1][a][0][e]; af[a][4 + e] = fy[KK & 1][a][1][e]; }
#pragma unroll
      for (int b = 0; b < FRI; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) { bfr[b][e] = fx[U % NX][b][0][e]; bfr[b][4 + e] = fx[U % NX][b][1][e]; }
      if constexpr (X3) {                               // hi x hi + hi x lo + lo x hi (the dropped lo x lo is <= 2^-22 of the product)
        acc[KS][0][0] = mfma16<DT>(af[0], bfr[0], acc[KS][0][0]);
        acc[KS][0][0] = mfma16<DT>(af[0], bfr[1], acc[KS][0][0]);
        acc[KS][0][0] = mfma16<DT>(af[1], bfr[0], acc[KS][0][0]);
        if constexpr (BIAS && KS == 0) { bacc = mfma16<DT>(af[0], ones, bacc); bacc = mfma16<DT>(af[1], ones, bacc); }
      } else {
#pragma unroll
        for (int a = 0; a < FRC; ++a)
#pragma unroll
          for (int b = 0; b < FRI; ++b) acc[KS][a][b] = mfma16<DT>(af[a], bfr[b], acc[KS][a][b]);
        if constexpr (BIAS && KS == 0) bacc = mfma16<DT>(kr == 0 ? af[0] : af[1], ones, bacc);
      }
    };
    stage(0);
    wait_vmcnt<0>();
    raw_barrier();
    VQ_WSTAMP(21);
    for (int c = 0; c < nchunks; ++c) {
      const char* base = (const char*)(lds + (c & 1) * STAGE);
      issue_y(base, std::integral_constant<int, 0>{});
      issue_x(base, std::integral_constant<int, 0>{});
      const bool more = c + 1 < nchunks;
      const bool late = (STAG & 1) && wave >= NW / 2;      // (wave-uniform) the SIMD's second wave stages half a chunk later
      if (more && !late) stage((c + 1) & 1);             // next chunk's DMA flies under this chunk's MFMAs
      step(base, std::integral_constant<int, 0>{});  step(base, std::integral_constant<int, 1>{});
      step(base, std::integral_constant<int, 2>{});  step(base, std::integral_constant<int, 3>{});
      step(base, std::integral_constant<int, 4>{});  step(base, std::integral_constant<int, 5>{});
      if (more && late) stage((c + 1) & 1);
      step(base, std::integral_constant<int, 6>{});  step(base, std::integral_constant<int, 7>{});
      step(base, std::integral_constant<int, 8>{});  step(base, std::integral_constant<int, 9>{});
      step(base, std::integral_constant<int, 10>{}); step(base, std::integral_constant<int, 11>{});
      wait_vmcnt<0>();
      raw_barrier();
    }
    if constexpr (BIAS) store_bias(bacc);
  };
  if (nchunks > 0) {
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});
  }
  VQ_WSTAMP(22);

  const int fr = lane & 31, fh = lane >> 5;
  const int CoutS = X3 ? p.d.Cout / 2 : p.d.Cout, CinS = X3 ? p.d.Cin / 2 : p.d.Cin;       // slab extents (X3: real channels)
  const int cob = X3 ? (co0 + wco) / 2 : co0 + wco, cib = X3 ? (ci0 + wci) / 2 : ci0 + wci;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    float* out = p.part + ((int64_t)(split * p.RS + kr * 3 + ks) * CoutS) * CinS;
#pragma unroll
    for (int b = 0; b < BFR; ++b) {
      const int ci = cib + b * 32 + fr;
#pragma unroll
      for (int a = 0; a < AFR; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = cob + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          wg_store(p, &out[(int64_t)co * CinS + ci], acc[ks][a][b][e]);
        }
    }
  }
  VQ_WSTAMP(23);
}

// dw[co][ci][tap] (+)= sum_split part[split][tap][co][ci]   (fixed summation order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int RS, int Cout, int Cin,
                                    int Cout_w, int Cin_w, int accumulate, float* __restrict__ dw,
                                    const float* __restrict__ bias_part, float* __restrict__ dbias, int dw_blocks,
                                    float alpha, const float* __restrict__ alpha_dev) {
  if (alpha_dev) alpha *= *alpha_dev;
  if ((int)blockIdx.x >= dw_blocks) {   // trailing blocks: bias gradient partials (same launch, no extra kernel)
    const int c = ((int)blockIdx.x - dw_blocks) * blockDim.x + threadIdx.x;
    if (c < Cout_w) {
      float s = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) s += bias_part[(int64_t)sp * Cout + c];
      s *= alpha;
      dbias[c] = accumulate ? dbias[c] + s : s;
    }
    return;
  }
  // one thread per (tap, co, ci): reads coalesce along ci, writes scatter with stride RS floats (the
  // gradient is 9x smaller than what is read, so the scattered 4-byte stores are not the bottleneck)
  const int64_t per_tap = (int64_t)Cout_w * Cin_w, total = per_tap * RS;
  const int64_t plane = (int64_t)Cout * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)dw_blocks * blockDim.x) {
    const int tap = (int)(i / per_tap);
    const int64_t j = i - (int64_t)tap * per_tap;
    const int co = (int)(j / Cin_w), ci = (int)(j - (int64_t)co * Cin_w);
    const float* src = part + (int64_t)tap * plane + (int64_t)co * Cin + ci;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += src[(int64_t)sp * RS * plane];
    s *= alpha;
    float* dst = dw + j * RS + tap;
    *dst = accumulate ? (*dst + s) : s;
  }
}

// The same sum with 16 B per lane: LPI lanes per (tap, co, 4 consecutive ci) item, lane `sub` summing splits sub, sub + LPI, ... in
// order (up to eight in flight), then a fixed-order butterfly over the LPI lanes.  LPI = 1 was the round-2 form: a 128-channel layer
// at 256x256 has 84 slabs of 590 KB but only 36,864 items — 144 blocks of serial readers on 256 CUs, 2 TB/s on slabs that sit in the
// Infinity Cache, and this launch is 5-6 % of the weight-gradient family's time.  The summation order is a function of (nsplit, LPI)
// only: deterministic, like everything else here.
template <int LPI>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ part, int nsplit, int RS, int Cout, int Cin,
                                                            int Cout_w, int Cin_w, int accumulate, float* __restrict__ dw,
                                                            const float* __restrict__ bias_part, float* __restrict__ dbias,
                                                            int dw_blocks, float alpha, const float* __restrict__ alpha_dev) {
  if (alpha_dev) alpha *= *alpha_dev;
  if ((int)blockIdx.x >= dw_blocks) {
    const int c = ((int)blockIdx.x - dw_blocks) * blockDim.x + threadIdx.x;
    if (c < Cout_w) {
      float s = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) s += bias_part[(int64_t)sp * Cout + c];
      s *= alpha;
      dbias[c] = accumulate ? dbias[c] + s : s;
    }
    return;
  }
  const int cq = Cin_w >> 2;
  const int64_t per_tap = (int64_t)Cout_w * cq, total = per_tap * RS;
  const int64_t plane = (int64_t)Cout * Cin, stride = (int64_t)RS * plane;
  const int sub = threadIdx.x % LPI;
  constexpr int IPB = 256 / LPI;                       // items per block
  // every lane of an LPI-group runs the same trip count (the butterfly below needs all of them): items beyond `total` are clamped
  const int64_t rounds = (total + (int64_t)dw_blocks * IPB - 1) / ((int64_t)dw_blocks * IPB);
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i0 = (r * dw_blocks + blockIdx.x) * IPB + threadIdx.x / LPI;
    const bool live = i0 < total;
    const int64_t i = live ? i0 : total - 1;
    const int tap = (int)(i / per_tap);
    const int64_t j = i - (int64_t)tap * per_tap;
    const int co = (int)(j / cq), ci = (int)(j - (int64_t)co * cq) << 2;
    const float* src = part + (int64_t)tap * plane + (int64_t)co * Cin + ci;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = sub;
    for (; sp + 7 * LPI < nsplit; sp += 8 * LPI) {     // fixed order per lane: splits sub, sub + LPI, ... (the loads of a trip are independent)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(src + (int64_t)(sp + u * LPI) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sp < nsplit; sp += LPI) {
      const float4 v = *(const float4*)(src + (int64_t)sp * stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int m = 1; m < LPI; m <<= 1) {
      s.x += __shfl_xor(s.x, m); s.y += __shfl_xor(s.y, m); s.z += __shfl_xor(s.z, m); s.w += __shfl_xor(s.w, m);
    }
    if (live && sub == 0) {
      float* dst = dw + ((int64_t)co * Cin_w + ci) * RS + tap;
      const float rr[4] = {s.x * alpha, s.y * alpha, s.z * alpha, s.w * alpha};
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[(int64_t)k * RS] = accumulate ? dst[(int64_t)k * RS] + rr[k] : rr[k];
    }
  }
}

// 3x3 weights with enough (co, ci) pairs to fill the chip (512 x 512 and up; measured: 256 x 256 weights with 64 splits lose 15 %
// with only 64 blocks of serial readers): one thread per (co, 4 consecutive ci) sums ALL nine taps, so its 36
// results are 144 contiguous bytes of dw[co][ci][tap] and leave as nine 16-byte stores.  (With one thread per tap every 64-byte
// line of dw was completed by nine partial writes from nine different blocks.)
__global__ __launch_bounds__(256) void wgrad_reduce9_kernel(const float* __restrict__ part, int nsplit, int Cout, int Cin, int Cout_w,
                                                            int Cin_w, int accumulate, float* __restrict__ dw,
                                                            const float* __restrict__ bias_part, float* __restrict__ dbias,
                                                            int dw_blocks, float alpha, const float* __restrict__ alpha_dev) {
  if (alpha_dev) alpha *= *alpha_dev;
  if ((int)blockIdx.x >= dw_blocks) {
    const int c = ((int)blockIdx.x - dw_blocks) * blockDim.x + threadIdx.x;
    if (c < Cout_w) {
      float s = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) s += bias_part[(int64_t)sp * Cout + c];
      s *= alpha;
      dbias[c] = accumulate ? dbias[c] + s : s;
    }
    return;
  }
  const int cq = Cin_w >> 2;
  const int64_t total = (int64_t)Cout_w * cq;
  const int64_t plane = (int64_t)Cout * Cin, stride = 9 * plane;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)dw_blocks * blockDim.x) {
    const int co = (int)(j / cq), ci = (int)(j - (int64_t)co * cq) << 2;
    const float* src = part + (int64_t)co * Cin + ci;
    float4 s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 2 <= nsplit; sp += 2) {          // fixed order: split 0, 1, 2, ...; 18 independent 16-byte loads per trip
      float4 v[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t) v[u][t] = *(const float4*)(src + (int64_t)(sp + u) * stride + (int64_t)t * plane);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 9; ++t) { s[t].x += v[u][t].x; s[t].y += v[u][t].y; s[t].z += v[u][t].z; s[t].w += v[u][t].w; }
    }
    for (; sp < nsplit; ++sp) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 v = *(const float4*)(src + (int64_t)sp * stride + (int64_t)t * plane);
        s[t].x += v.x; s[t].y += v.y; s[t].z += v.z; s[t].w += v.w;
      }
    }
    // dw[co][ci + k][tap], k = 0..3: 36 consecutive floats, 16-byte aligned (ci is a multiple of 4)
    float r[36];
#pragma unroll
    for (int t = 0; t < 9; ++t) { r[t] = s[t].x * alpha; r[9 + t] = s[t].y * alpha; r[18 + t] = s[t].z * alpha; r[27 + t] = s[t].w * alpha; }
    float4* dst = (float4*)(dw + ((int64_t)co * Cin_w + ci) * 9);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float4 o = make_float4(r[q * 4], r[q * 4 + 1], r[q * 4 + 2], r[q * 4 + 3]);
      if (accumulate) { const float4 old = dst[q]; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      dst[q] = o;
    }
  }
}

// VQ_F16X2 operands (include/vqhip.h): the split-K kernels above run UNCHANGED on the virtual problem — dY and X read as binary16
// tensors of 2 Cout / 2 Cin virtual channels (row / column 16 g + e = hi piece, 16 g + 8 + e = lo piece of real channel 8 g + e) —
// so a partial slab holds hi*hi, hi*lo, lo*hi and lo*lo of every (co, ci) pair in four places, and this reduction adds the four
// (small terms first, separately accumulated over the splits) while it sums the splits in their fixed order.  Four MFMAs per product
// where the forward / data-gradient kernels issue three: the price of not touching the transposed-fragment pipelines.
// LPI lanes per (tap, co, 4 consecutive ci) item, as in wgrad_reduce4_kernel.
template <int LPI>
__global__ __launch_bounds__(256) void wgrad_reduce_x2_kernel(const float* __restrict__ part, int nsplit, int RS, int CoutV, int CinV,
                                                              int Cout_w, int Cin_w, int accumulate, float* __restrict__ dw,
                                                              const float* __restrict__ bias_part, float* __restrict__ dbias,
                                                              int dw_blocks, float alpha, const float* __restrict__ alpha_dev) {
  if (alpha_dev) alpha *= *alpha_dev;
  if ((int)blockIdx.x >= dw_blocks) {
    const int c = ((int)blockIdx.x - dw_blocks) * blockDim.x + threadIdx.x;
    if (c < Cout_w) {
      const int vh = ((c >> 3) << 4) + (c & 7);
      float s = 0.f, t = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) { s += bias_part[(int64_t)sp * CoutV + vh]; t += bias_part[(int64_t)sp * CoutV + vh + 8]; }
      s = (s + t) * alpha;
      dbias[c] = accumulate ? dbias[c] + s : s;
    }
    return;
  }
  const int cq = (Cin_w + 3) >> 2;
  const int64_t per_tap = (int64_t)Cout_w * cq, total = per_tap * RS;
  const int64_t plane = (int64_t)CoutV * CinV, stride = (int64_t)RS * plane;
  const int sub = threadIdx.x % LPI;
  constexpr int IPB = 256 / LPI;
  const int64_t rounds = (total + (int64_t)dw_blocks * IPB - 1) / ((int64_t)dw_blocks * IPB);
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i0 = (r * dw_blocks + blockIdx.x) * IPB + threadIdx.x / LPI;
    const bool live = i0 < total;
    const int64_t i = live ? i0 : total - 1;
    const int tap = (int)(i / per_tap);
    const int64_t j = i - (int64_t)tap * per_tap;
    const int co = (int)(j / cq), ci = (int)(j - (int64_t)co * cq) << 2;
    const int vr = ((co >> 3) << 4) + (co & 7), vc = ((ci >> 3) << 4) + (ci & 7);
    const float* src = part + (int64_t)tap * plane + (int64_t)vr * CinV + vc;
    const int64_t lo_row = (int64_t)8 * CinV;
    float4 big = make_float4(0.f, 0.f, 0.f, 0.f), small = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = sub; sp < nsplit; sp += LPI) {
      const float* q = src + (int64_t)sp * stride;
      const float4 hh = *(const float4*)q, hl = *(const float4*)(q + 8), lh = *(const float4*)(q + lo_row), ll = *(const float4*)(q + lo_row + 8);
      big.x += hh.x; big.y += hh.y; big.z += hh.z; big.w += hh.w;
      small.x += (hl.x + lh.x) + ll.x; small.y += (hl.y + lh.y) + ll.y; small.z += (hl.z + lh.z) + ll.z; small.w += (hl.w + lh.w) + ll.w;
    }
#pragma unroll
    for (int m = 1; m < LPI; m <<= 1) {
      big.x += __shfl_xor(big.x, m); big.y += __shfl_xor(big.y, m); big.z += __shfl_xor(big.z, m); big.w += __shfl_xor(big.w, m);
      small.x += __shfl_xor(small.x, m); small.y += __shfl_xor(small.y, m); small.z += __shfl_xor(small.z, m); small.w += __shfl_xor(small.w, m);
    }
    if (live && sub == 0) {
      float* dst = dw + ((int64_t)co * Cin_w + ci) * RS + tap;
      const float rr[4] = {(big.x + small.x) * alpha, (big.y + small.y) * alpha, (big.z + small.z) * alpha, (big.w + small.w) * alpha};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ci + k < Cin_w) dst[(int64_t)k * RS] = accumulate ? dst[(int64_t)k * RS] + rr[k] : rr[k];
    }
  }
}
static inline VqConvDesc wg_x2_virtual(const VqConvDesc* d) {
  VqConvDesc v = *d;
  if (v.dtype == VQ_F16X2) { v.dtype = VQ_F16; v.Cin *= 2; v.Cout *= 2; v.Cin_w = v.Cin; v.Cout_w = v.Cout; }
  return v;
}

static int ilog2_exact_w(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

// LDS-DMA kernel preconditions: bf16, single-term MFMA, power-of-two output extent, 64 | M
static bool wgrad_glds_eligible(const VqConvDesc* d) {
  return (d->dtype == VQ_BF16 || d->dtype == VQ_F16) && d->split == 1 && ilog2_exact_w(d->Wo) >= 0 && ilog2_exact_w(d->Ho) >= 0 &&
         ((int64_t)d->N * d->Ho * d->Wo) % 64 == 0 && d->Cout % 64 == 0 && d->Cin % 64 == 0;
}

// VqConvDesc.kernel_hint as vq_conv2d_wgrad / vq_conv2d_wgrad_workspace read it (include/vqhip.h; 0 = the plan's own choice, what the
// product passes): 64 / 128 / 256 = force that one-tap LDS-DMA tile, +4 = never the three-tap kernel, +1 = the 4 B/lane split
// reduction (and, in ABLATE builds, the no-DMA ablation of the one-tap kernel), +16 = the three-tap kernel without the staging
// stagger, +8 = VQ_F16X2 weight gradients on the virtual 2C x 2C problem (A/B of the native three-product form); bits 16-31 = forced
// split-K count.  Part of the descriptor: no process-global state.
static inline int wg_hint_tile(const VqConvDesc* d) { return d->kernel_hint & (64 | 128 | 256); }
static inline bool wg_hint_no3(const VqConvDesc* d) { return (d->kernel_hint & 4) != 0; }
static inline bool wg_hint_slow_reduce(const VqConvDesc* d) { return (d->kernel_hint & 1) != 0; }
static inline bool wg_hint_unstaggered(const VqConvDesc* d) { return (d->kernel_hint & 16) != 0; }
static inline int wg_hint_split(const VqConvDesc* d) { return (d->kernel_hint >> 16) & 0xffff; }
static inline bool wg_hint_x2_virtual(const VqConvDesc* d) { return (d->kernel_hint & 8) != 0; }   // +8 = VQ_F16X2 on the virtual problem (rounds 5 form)
static bool wg_hint_supported(const VqConvDesc* d) { return (d->kernel_hint & 0xffff & ~(1 | 4 | 8 | 16 | 64 | 128 | 256)) == 0; }

// conv_wgrad3_kernel: 3x3 / stride 1 / pad 1 (also behind a nearest-2x upsample), 128-multiples of channels, output rows
// that are a multiple of 4 pixels (<= 96 halo slots)
// output rows need not be powers of two: a multiple of 4 pixels is enough (crop-invariance batches at every level of the pyramid)
static bool wgrad3_eligible(const VqConvDesc* d) {
  return (d->dtype == VQ_BF16 || d->dtype == VQ_F16) && d->split == 1 && ((int64_t)d->N * d->Ho * d->Wo) % 64 == 0 && !wg_hint_no3(d) && !wg_hint_tile(d) &&
         d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && (d->up == 1 || d->up == 2) && d->pad_t == 1 &&
         d->pad_l == 1 && d->Ho == d->H * d->up && d->Wo == d->W * d->up && d->Wo % 4 == 0 && d->Cout % 128 == 0 &&
         d->Cin % 128 == 0;
}

// CUs the split plan fills: 256, or VQ_WGRAD_CUS (tools' A/B of a CU-masked weight-gradient stream, ops.VQ_SIDE_CU_MASK: a plan made for
// 256 CUs runs two rounds on any smaller mask)
static int wgrad_cus() {
#ifdef VQ_ABLATION_KERNELS      // (the knob changes split plans and workspace sizes: `make ablate` libraries and the emulator only)
  static const int v = [] { const char* e = getenv("VQ_WGRAD_CUS"); const int x = e ? atoi(e) : 0; return (x >= 8 && x <= 256) ? x / 8 * 8 : 256; }();
  return v;
#else
  return 256;
#endif
}
// x3: the native three-product form of the three-tap kernel on a virtualised VQ_F16X2 descriptor (wg_x3): four-wave blocks, two per
// CU, slabs of REAL channels (a quarter of the virtual problem's)
static void wgrad_plan(const VqConvDesc* d, int& BT, int& n_ct, int& n_cit, int& nsplit, int& pix_per_split, int* xcd_tiles = nullptr,
                       bool x3 = false) {
  if (xcd_tiles) *xcd_tiles = 0;
  BT = (d->Cout >= 128 && d->Cin >= 128) ? 128 : 64;
  if (wgrad_glds_eligible(d)) {
    // the 256 tile needs a long reduction per block to pay off (measured: wins from ~128k output pixels up)
    // (16-tap descriptors = the transposed form of the Upsample conv, ops.conv_wgrad_raw: 16 tiles per channel-tile pair
    // keep the chip full with long reductions much earlier — measured 997 -> 874 us at 512 channels, 64x64, B = 16)
    if (d->Cout % 256 == 0 && d->Cin % 256 == 0 &&
        (int64_t)d->N * d->Ho * d->Wo >= (d->R * d->S >= 16 ? 16384 : 131072)) BT = 256;
    else if (d->Cout % 128 == 0 && d->Cin % 128 == 0) BT = 128;
    else BT = 64;
    if (wg_hint_tile(d) && d->Cout % wg_hint_tile(d) == 0 && d->Cin % wg_hint_tile(d) == 0) BT = wg_hint_tile(d);
  }
  const bool three = wgrad3_eligible(d);
  if (three) BT = 128;
  n_ct = (int)vq_ceil_div(d->Cout, BT);
  n_cit = (int)vq_ceil_div(d->Cin, BT);
  const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
  const int tiles = n_ct * n_cit * (three ? 3 : d->R * d->S);
  // ~1.5 waves of 2 blocks/CU (1 block/CU for the 8-wave 256 tile and the three-tap kernel); more splits only feed the reduce kernel
  const bool one_per_cu = (BT == 256 || three) && !x3;
  const int cus = wgrad_cus();
  int64_t want = vq_ceil_div((one_per_cu ? 2 : 3) * cus, tiles);
  int64_t max_split = vq_ceil_div(M, 512);   // at least 8 chunks of 64 pixels per split
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  // the LDS-DMA kernels hand whole splits to XCDs (block % 8): keep all 8 busy and balanced
  if ((wgrad_glds_eligible(d) || three) && max_split >= 8) want = vq_ceil_div(want, 8) * 8;
  if ((wgrad_glds_eligible(d) || three) && max_split >= 8) {
    // The grid runs in rounds of `slots` resident blocks (LDS-limited blocks per CU x 256 CUs): pick the split count
    // (multiple of 8) that minimises  kernel time x (rounds * slots / blocks)  +  partial-sum traffic (written once,
    // read once by the reduce).  A plain "blocks >= target" rule left e.g. the 128-channel layers with 528 blocks on
    // 512 slots: a third round for 16 blocks (measured: 712 -> 947 TFLOP/s on that layer).
    const int slots = cus * (one_per_cu ? 1 : (BT == 128 ? 2 : 4));
    const double t_kernel = 2.0 * (double)M * d->Cout * d->Cin * d->R * d->S / 7.0e14;
    const double t_split = 2.0 * d->R * d->S * d->Cout * d->Cin * 4.0 / 4.0e12 * (x3 ? 0.25 : 1.0);
    double best = 1e30;
    for (int64_t ns = 8; ns <= max_split && ns <= 256; ns += 8) {
      const int64_t blocks = ns * tiles, rounds = vq_ceil_div(blocks, slots);
      const double cost = t_kernel * (double)(rounds * slots) / (double)blocks + t_split * (double)ns;
      if (cost < best) { best = cost; want = ns; }
    }
    // three-tap kernel, tiles a multiple of 8: with an XCD owning TILES (all their splits) every split count balances the XCDs
    if (xcd_tiles) {
      const int mode = (three && tiles % 8 == 0) ? 1 : 2;
      if (mode == 2 || tiles % 8 == 0) {
        for (int64_t ns = 1; ns <= max_split && ns <= 256; ++ns) {
          const int64_t blocks = ns * tiles, rounds = vq_ceil_div(blocks, slots);
          const double cost = t_kernel * (double)(rounds * slots) / (double)blocks + t_split * (double)ns;
          if (cost < best * 0.97) { best = cost; want = ns; *xcd_tiles = mode; }
        }
      }
    }
  }
  // a split count that is not a multiple of 8 (short reductions: fewer than 8 splits possible) under the split-owning map would
  // leave XCDs idle: the range map has no such constraint
  if (xcd_tiles && *xcd_tiles == 0 && (three || wgrad_glds_eligible(d)) && want % 8 != 0) *xcd_tiles = 2;
  if (wg_hint_split(d) > 0) {
    want = wg_hint_split(d) < max_split ? wg_hint_split(d) : max_split;
    if (xcd_tiles) *xcd_tiles = ((three || wgrad_glds_eligible(d)) && want % 8 != 0) ? ((three && tiles % 8 == 0) ? 1 : 2) : 0;
  }
  int64_t pps = vq_ceil_div(vq_ceil_div(M, want), 64) * 64;
  nsplit = (int)vq_ceil_div(M, pps);
  pix_per_split = (int)pps;
}

static size_t wgrad_part_bytes(const VqConvDesc* d, int nsplit) {
  return (size_t)nsplit * d->R * d->S * d->Cout * d->Cin * sizeof(float);
}
static size_t wgrad_bias_bytes(const VqConvDesc* d, int nsplit) {
  return ((size_t)nsplit * d->Cout * sizeof(float) + 255) / 256 * 256;
}
// VQ_F16X2 weight gradients in the native three-product form: whatever the three-tap kernel takes on the virtual descriptor
static bool wg_x3(const VqConvDesc* d0, const VqConvDesc* dvirt) {
  return d0->dtype == VQ_F16X2 && !wg_hint_x2_virtual(d0) && wgrad3_eligible(dvirt);
}
// ... and whatever the one-tap LDS-DMA kernel takes on it with tiles of 128 / 256 virtual channels (the plan's BT; the 64-wide tile —
// 16 real channels per wave and side — keeps the virtual form)
static bool wg_x3_glds(const VqConvDesc* d0, const VqConvDesc* dvirt, int BT) {
  return d0->dtype == VQ_F16X2 && !wg_hint_x2_virtual(d0) && !wgrad3_eligible(dvirt) && wgrad_glds_eligible(dvirt) && BT >= 128;
}

template <int DT, int BT, int NW, int X3 = 0>
static int launch_wgrad_glds(const WgradParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t LDS_BYTES = (size_t)2 * 2 * 64 * BT * sizeof(vq_bf16);
  VQ_RESERVE_LDS((conv_wgrad_glds_kernel<DT, BT, NW, X3>), LDS_BYTES, "vq_conv2d_wgrad");
  hipLaunchKernelGGL((conv_wgrad_glds_kernel<DT, BT, NW, X3>), grid, dim3(NW * 64), LDS_BYTES, s, p);
  return VQ_OK;
}

template <int DT, int GEN, int NW, int SEG, int STAG, int X3 = 0>
static int launch_wgrad3_form(const WgradParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t LDS_BYTES = (size_t)2 * (64 + (GEN ? 96 : 72)) * 128 * sizeof(vq_bf16);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS capacity");
  VQ_RESERVE_LDS((conv_wgrad3_kernel<DT, GEN, NW, SEG, STAG, X3>), LDS_BYTES, "vq_conv2d_wgrad");
  hipLaunchKernelGGL((conv_wgrad3_kernel<DT, GEN, NW, SEG, STAG, X3>), grid, dim3(NW * 64), LDS_BYTES, s, p);
  return VQ_OK;
}
template <int DT, int GEN, int NW, int STAG, int X3 = 0>
static int launch_wgrad3_nw(const WgradParams& p, dim3 grid, hipStream_t s) {
  if constexpr (GEN) return launch_wgrad3_form<DT, 1, NW, 0, STAG, X3>(p, grid, s);
  else {
    if (p.seg_shift == 4) return launch_wgrad3_form<DT, 0, NW, 4, STAG, X3>(p, grid, s);
    if (p.seg_shift == 5) return launch_wgrad3_form<DT, 0, NW, 5, STAG, X3>(p, grid, s);
    return launch_wgrad3_form<DT, 0, NW, 6, STAG, X3>(p, grid, s);
  }
}
template <int DT, int GEN>
static int launch_wgrad3(const WgradParams& p, dim3 grid, hipStream_t s) {
  // the kernel addresses the input with 32-bit element offsets
  if ((int64_t)p.d.N * p.d.H * p.d.W * p.d.Cin >= ((int64_t)1 << 31)) { vq_set_error("vq_conv2d_wgrad(three-tap): input of 2^31 elements or more"); return VQ_ERR_UNSUPPORTED; }
  if constexpr (DT == VQ_F16) {
    if (p.x3) return launch_wgrad3_nw<DT, GEN, 4, 1, 1>(p, grid, s);    // VQ_F16X2, native three-product form (four waves, staggered staging)
  }
  // hint +16: every wave stages right after the chunk barrier (rounds 1-2) instead of the staggered form
  if (!wg_hint_unstaggered(&p.d)) return launch_wgrad3_nw<DT, GEN, 8, 1>(p, grid, s);
  return launch_wgrad3_nw<DT, GEN, 8, 0>(p, grid, s);
}

extern "C" size_t vq_conv2d_wgrad_workspace(const VqConvDesc* d0) {
  if (!d0) return 0;
  if (vq_wgrad_c8_x2_eligible(d0))
    return (vq_wgrad_c8_x2_workspace(d0) + 255) / 256 * 256 + vq_colsum_workspace((int64_t)d0->N * d0->Ho * d0->Wo, d0->Cout);
  const VqConvDesc dvirt = wg_x2_virtual(d0);
  const VqConvDesc* d = &dvirt;
  int BT, n_ct, n_cit, nsplit, pps;
  int xt;
  bool x3 = wg_x3(d0, d);
  wgrad_plan(d, BT, n_ct, n_cit, nsplit, pps, &xt, x3);    // the same plan vq_conv2d_wgrad makes (incl. the tile-owning split counts)
  x3 = x3 || wg_x3_glds(d0, d, BT);
  // [ dW partials | bias partials (LDS-DMA kernels) | column-sum scratch (other kernels) ]
  size_t main_bytes = x3 ? wgrad_part_bytes(d0, nsplit) + wgrad_bias_bytes(d0, nsplit)      // (slabs of real channels)
                         : wgrad_part_bytes(d, nsplit) + wgrad_bias_bytes(d, nsplit);
  if (vq_wgrad_c8_eligible(d)) main_bytes = (vq_wgrad_c8_workspace(d) + 255) / 256 * 256;
  return main_bytes + vq_colsum_workspace((int64_t)d->N * d->Ho * d->Wo, d->Cout);
}

extern "C" int vq_conv2d_wgrad(const VqConvDesc* d0, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                               void* workspace, size_t ws_bytes, void* stream) {
  VQ_REQUIRE(d0 && x && dy && dw, VQ_ERR_INVALID, "vq_conv2d_wgrad: null pointer");
  const bool x2 = d0->dtype == VQ_F16X2;
  VQ_REQUIRE(!x2 || d0->split == 1, VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: VQ_F16X2 storage takes split = 1");
  const VqConvDesc dvirt = wg_x2_virtual(d0);          // VQ_F16X2: the kernels run on the virtual binary16 problem (see wgrad_reduce_x2_kernel)
  const VqConvDesc* d = &dvirt;
  VQ_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, VQ_ERR_INVALID, "vq_conv2d_wgrad: channels must be multiples of 8");
  const int dsh = ilog2_exact_w(d->dil_in), ush = ilog2_exact_w(d->up);
  VQ_REQUIRE(dsh >= 0 && ush >= 0 && ush <= 1, VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: bad dil_in/up");
  VQ_REQUIRE(d->subpix == 0, VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: sub-pixel descriptors are forward-only");
  VQ_REQUIRE(wg_hint_supported(d), VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: unknown kernel_hint bits (%d)", d->kernel_hint);
  VQ_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) - 4096, VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: pixel count exceeds int32");
  const size_t need = vq_conv2d_wgrad_workspace(d0);
  VQ_REQUIRE(workspace && ws_bytes >= need, VQ_ERR_WORKSPACE, "vq_conv2d_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
  const float alpha = d->alpha == 0.f ? 1.f : d->alpha;
  if (vq_wgrad_c8_x2_eligible(d0)) {   // the 3-channel image layers in VQ_F16X2 storage: two passes of the one-pass kernel (conv_small.hip)
    int bias_done = 0;
    int rc = vq_launch_wgrad_c8_x2(d0, x, dy, dw, dbias, &bias_done, accumulate, alpha, workspace, (hipStream_t)stream);
    if (rc) return rc;
    if (dbias && !bias_done) {
      const int64_t pixels = (int64_t)d0->N * d0->Ho * d0->Wo;
      void* cws = (char*)workspace + (vq_wgrad_c8_x2_workspace(d0) + 255) / 256 * 256;
      return vq_colsum(dy, pixels, d0->Cout, d0->dtype, dbias, d0->Cout_w, accumulate, alpha, nullptr, cws,
                       vq_colsum_workspace(pixels, d0->Cout), stream);
    }
    return VQ_OK;
  }
  if (vq_wgrad_c8_eligible(d)) {   // 3-channel image layers: one pass over dY for all 9 taps (conv_small.hip)
    VQ_REQUIRE(d->alpha_dev == nullptr, VQ_ERR_UNSUPPORTED, "vq_conv2d_wgrad: the 8-channel kernels take a host alpha only");
    int bias_done = 0;
    int rc = vq_launch_wgrad_c8(d, x, dy, dw, dbias, &bias_done, accumulate, alpha, workspace, (hipStream_t)stream);
    if (rc) return rc;
    if (dbias && !bias_done) {
      const int64_t pixels = (int64_t)d->N * d->Ho * d->Wo;
      void* cws = (char*)workspace + (vq_wgrad_c8_workspace(d) + 255) / 256 * 256;
      return vq_colsum(dy, pixels, d->Cout, d->dtype, dbias, d->Cout_w, accumulate, alpha, nullptr, cws,
                       vq_colsum_workspace(pixels, d->Cout), stream);
    }
    return VQ_OK;
  }
  WgradParams p;
  p.d = *d; p.x = x; p.dy = dy; p.part = (float*)workspace;
  p.M = d->N * d->Ho * d->Wo; p.HoWo = d->Ho * d->Wo; p.RS = d->R * d->S;
  p.dsh = dsh; p.ush = ush;
  int BT, nsplit;
  bool x3 = wg_x3(d0, d);
  wgrad_plan(d, BT, p.n_ct, p.n_cit, nsplit, p.pix_per_split, &p.xcd_tiles, x3);
  x3 = x3 || wg_x3_glds(d0, d, BT);
  p.x3 = x3 ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.n_ct * p.n_cit * p.RS, nsplit);
  p.wo_shift = ilog2_exact_w(d->Wo); p.ho_shift = ilog2_exact_w(d->Ho);
  const bool glds_ok = wgrad_glds_eligible(d) || wgrad3_eligible(d);
  p.seg_shift = 0;
  while (p.seg_shift < 6 && d->Wo % (2 << p.seg_shift) == 0) ++p.seg_shift;
  const VqConvDesc* ds = x3 ? d0 : d;                  // the descriptor whose channel counts index the slabs and the split reduction
  float* bias_part = (float*)((char*)workspace + wgrad_part_bytes(ds, nsplit));
  void* colsum_ws = (char*)bias_part + wgrad_bias_bytes(ds, nsplit);
  p.bias_part = (glds_ok && dbias && (wgrad3_eligible(d) || p.RS * p.n_cit >= BT / 64)) ? bias_part : nullptr;   // FRC blocks per cout tile carry the bias fragments
#define VQ_WG(DTv, SPv, BTv, NB) \
  hipLaunchKernelGGL((conv_wgrad_kernel<DTv, SPv, BTv, 32, NB>), grid, dim3(256), 0, s, p)
  p.nsplit = nsplit;
#ifdef VQ_ABLATION_KERNELS
  p.dbg = wg_hint_slow_reduce(d) ? 1 : 0;
#else
  p.dbg = 0;
#endif
  if (glds_ok) {
    int rc = VQ_OK;
    const bool three = wgrad3_eligible(d);
    const dim3 grid1(p.xcd_tiles == 2 ? 8u * (unsigned)vq_ceil_div((int64_t)nsplit * (p.n_ct * p.n_cit * (three ? 3 : p.RS)), 8)
                     : p.xcd_tiles ? (unsigned)nsplit * (unsigned)(p.n_ct * p.n_cit * 3)
                                 : 8u * (unsigned)vq_ceil_div(nsplit, 8) * (unsigned)(p.n_ct * p.n_cit * (three ? 3 : p.RS)));
    const bool pow2 = p.wo_shift >= 0 && p.ho_shift >= 0 && d->Wo >= 16;
    if (d->dtype == VQ_F16) {
      if (three) rc = pow2 ? launch_wgrad3<VQ_F16, 0>(p, grid1, s) : launch_wgrad3<VQ_F16, 1>(p, grid1, s);
      else if (BT == 256) rc = p.x3 ? launch_wgrad_glds<VQ_F16, 256, 8, 1>(p, grid1, s) : launch_wgrad_glds<VQ_F16, 256, 8>(p, grid1, s);
      else if (BT == 128) rc = p.x3 ? launch_wgrad_glds<VQ_F16, 128, 4, 1>(p, grid1, s) : launch_wgrad_glds<VQ_F16, 128, 4>(p, grid1, s);
      else rc = launch_wgrad_glds<VQ_F16, 64, 4>(p, grid1, s);
    } else {
      if (three) rc = pow2 ? launch_wgrad3<VQ_BF16, 0>(p, grid1, s) : launch_wgrad3<VQ_BF16, 1>(p, grid1, s);
      else if (BT == 256) rc = launch_wgrad_glds<VQ_BF16, 256, 8>(p, grid1, s);
      else if (BT == 128) rc = launch_wgrad_glds<VQ_BF16, 128, 4>(p, grid1, s);
      else rc = launch_wgrad_glds<VQ_BF16, 64, 4>(p, grid1, s);
    }
    if (rc) return rc;
  } else if (d->dtype == VQ_BF16 && d->split == 1) {
    if (BT == 128) VQ_WG(VQ_BF16, 1, 128, 2); else VQ_WG(VQ_BF16, 1, 64, 2);
  } else if (d->dtype == VQ_F16 && d->split == 1) {
    if (BT == 128) VQ_WG(VQ_F16, 1, 128, 2); else VQ_WG(VQ_F16, 1, 64, 2);
  } else if (d->dtype == VQ_F32 && d->split == 1) {
    if (BT == 128) VQ_WG(VQ_F32, 1, 128, 2); else VQ_WG(VQ_F32, 1, 64, 2);
  } else if (d->dtype == VQ_F32 && d->split == 3) {
    if (BT == 128) VQ_WG(VQ_F32, 3, 128, 1); else VQ_WG(VQ_F32, 3, 64, 1);
  } else if (d->dtype == VQ_F32 && d->split == 6) {
    if (BT == 128) VQ_WG(VQ_F32, 6, 128, 1); else VQ_WG(VQ_F32, 6, 64, 1);
  } else {
    vq_set_error("vq_conv2d_wgrad: unsupported dtype/split combination (%d/%d)", d->dtype, d->split);
    return VQ_ERR_UNSUPPORTED;
  }
#undef VQ_WG
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad");
  d = ds;                                              // (from here on: the reduction's view — real channels for the native form)
  const int64_t total = (int64_t)d->Cout_w * d->Cin_w * p.RS;
  int blocks = (int)vq_ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  const int bias_blocks = (dbias && p.bias_part) ? (d->Cout_w + 255) / 256 : 0;
  if (x2 && !x3) {
    const int64_t items = (int64_t)d0->Cout_w * ((d0->Cin_w + 3) / 4) * p.RS;
    int lpi = 1;
    while (lpi < 8 && items * lpi < 131072 && lpi * 2 <= nsplit) lpi *= 2;
    blocks = (int)vq_ceil_div(items * lpi, 256);
    if (blocks > 4096) blocks = 4096;
    const int bb = (dbias && p.bias_part) ? (d0->Cout_w + 255) / 256 : 0;
#define VQ_RX(L) hipLaunchKernelGGL(wgrad_reduce_x2_kernel<L>, dim3(blocks + bb), dim3(256), 0, s, (const float*)workspace, nsplit, p.RS, \
                       d->Cout, d->Cin, d0->Cout_w, d0->Cin_w, accumulate, dw, (const float*)bias_part, dbias, blocks, alpha, d->alpha_dev)
    if (lpi == 8) VQ_RX(8); else if (lpi == 4) VQ_RX(4); else if (lpi == 2) VQ_RX(2); else VQ_RX(1);
#undef VQ_RX
  } else
  if (p.RS == 9 && d->Cin % 4 == 0 && d->Cin_w % 4 == 0 && (int64_t)d->Cout_w * d->Cin_w >= 512 * 512 && !wg_hint_slow_reduce(d) &&
      ((uintptr_t)dw & 15) == 0) {
    blocks = (int)vq_ceil_div((int64_t)d->Cout_w * (d->Cin_w / 4), 256);
    hipLaunchKernelGGL(wgrad_reduce9_kernel, dim3(blocks + bias_blocks), dim3(256), 0, s, (const float*)workspace, nsplit, d->Cout,
                       d->Cin, d->Cout_w, d->Cin_w, accumulate, dw, (const float*)bias_part, dbias, blocks, alpha, d->alpha_dev);
  } else if (d->Cin % 4 == 0 && d->Cin_w % 4 == 0 && !wg_hint_slow_reduce(d)) {
    // lanes per item: enough threads for >= 2 blocks per CU (131072), never more lanes than splits
    const int64_t items = total / 4;
    int lpi = 1;
    while (lpi < 8 && items * lpi < 131072 && lpi * 2 <= nsplit) lpi *= 2;
    blocks = (int)vq_ceil_div(items * lpi, 256);
    if (blocks > 4096) blocks = 4096;
#define VQ_R4(L) hipLaunchKernelGGL(wgrad_reduce4_kernel<L>, dim3(blocks + bias_blocks), dim3(256), 0, s, (const float*)workspace, nsplit, p.RS, \
                       d->Cout, d->Cin, d->Cout_w, d->Cin_w, accumulate, dw, (const float*)bias_part, dbias, blocks, alpha, d->alpha_dev)
    if (lpi == 8) VQ_R4(8); else if (lpi == 4) VQ_R4(4); else if (lpi == 2) VQ_R4(2); else VQ_R4(1);
#undef VQ_R4
  } else
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks + bias_blocks), dim3(256), 0, s, (const float*)workspace, nsplit, p.RS,
                     d->Cout, d->Cin, d->Cout_w, d->Cin_w, accumulate, dw, (const float*)bias_part, dbias, blocks, alpha, d->alpha_dev);
  VQ_CHECK_LAUNCH("vq_conv2d_wgrad(reduce)");
  if (dbias) {
    if (p.bias_part) {
    } else {
      const int64_t pixels = (int64_t)d->N * d->Ho * d->Wo;
      int rc = vq_colsum(dy, pixels, d0->Cout, d0->dtype, dbias, d0->Cout_w, accumulate, alpha, d->alpha_dev, colsum_ws,
                         vq_colsum_workspace(pixels, d0->Cout), stream);
      if (rc) return rc;
    }
  }
  return VQ_OK;
}
