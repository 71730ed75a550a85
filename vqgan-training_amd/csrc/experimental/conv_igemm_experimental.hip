// Measured-and-NOT-adopted implicit-GEMM kernels (round 2), kept for A/B runs only: textually included by conv_igemm.hip in
// `make ABLATE=1` builds (VQ_ABLATION_KERNELS) and reachable there through VqConvDesc.kernel_hint values >= VQ_HINT_EXPERIMENTAL.
// A release libvqhip.so contains none of this.  Numbers: profiles/r2m_patch128_micro.txt, r2p_patch128_single_phase_micro.txt,
// r2r_patch128x512_micro.txt ("not adopted").
#ifndef VQ_ABLATION_KERNELS
#error "experimental kernels are built with make ABLATE=1 only"
#endif
// The same tile for Cout = 128 layers (A/B candidate, dbg 2048): 128 weight rows (8 waves x 64c x 64p) make a ping-pong slot of the
// two-phase schedule only 8 MFMAs long (measured: the barriers weigh double, profiles/r2m_patch128_micro.txt), so here a stage
// is ONE slot pair — all four k-steps' fragments read in the load slot, 16 MFMAs in the matrix slot — and the weight tiles run
// through THREE buffers: the tile of stage s + 2 is requested in the load slot of stage s and only stage s + 1's must have landed
// at its end (counted vmcnt).  LDS: W0 | W1 | W2 (3 x 16 KiB) | X0 | X1 (2 x 41 KiB) = 130 KiB.
template <int DT>
__global__ __launch_bounds__(512) void conv_igemm_p9s_kernel(const ConvParams p) {
  constexpr int BC = 128;
  constexpr int BK = 64, BP = 256, WC = BC / 2, WP = 64, NWB = 3;
  constexpr int FC = WC / 32, FP = WP / 32, NWP = BP / WP, NW = 8;
  constexpr int TW = 16, TH = 16, HWD = TW + 2, NSLOT = (TH + 2) * HWD, PMAX = (NSLOT + 7) / 8;   // 324 halo rows, 41 pieces
  constexpr int WT = BC * BK, XT = PMAX * 8 * BK;      // elements per weight / patch buffer
  constexpr int XBASE = NWB * WT;                      // first element of X0
  constexpr int NBW = BC / 8 / NW;                     // weight pieces per wave per stage (4)
  static_assert(PMAX <= 6 * NW, "one patch piece per wave and tap, taps 0-5");

  VQ_DYN_LDS(vq_bf16, lds);

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
  const int pn = ptile / p.pt_tpi, prem = ptile - pn * p.pt_tpi, ptyi = prem / p.pt_tx;
  const int ty0 = ptyi * TH, tx0 = (prem - ptyi * p.pt_tx) * TW;    // top-left output pixel of the patch

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
  for (int a = 0; a < FC; ++a) wab[a] = (unsigned)(Swz<BK>::elem(wc0 + a * 32 + fr, fh) * 2);
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
#ifndef VQ_EMU
      asm volatile("" : "+v"(row));                    // opaque: nine taps' addresses must not be hoisted into registers
#endif
      row += (tap / 3) * HWD + (tap % 3);
      xab[b] = (unsigned)(XBASE * 2 + row * BK * 2 + ((fh ^ ((row >> 1) & 7)) << 4));
    }
  };
  // ---- prologue: weight tiles of stages 0 and 1, the whole patch of chunk 0 -------------------------------------------------
  const int nst = 9 * cpt;
  stage_w(0, 0, 0);
  stage_w(1, 1, 0);
#pragma unroll
  for (int i = 0; i < XPW; ++i) stage_x(0, i, 0);
  wait_vmcnt<0>();
  raw_barrier();
  const int grp = wave >> 2;
  if (grp == 1) raw_barrier();
  s16x8 af4[4][FC], bf4[4][FP];
  int ws = 0;                                          // weight buffer of the current stage (stage index mod 3)
  for (int cc = 0; cc < cpt; ++cc) {
    const bool more_c = cc + 1 < cpt;
    const unsigned xoff = (unsigned)((cc & 1) * XT * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const unsigned woff = (unsigned)(ws * WT * 2);
      set_tap(tap);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const unsigned x = (unsigned)(kk << 5);
#pragma unroll
        for (int a = 0; a < FC; ++a) af4[kk][a] = *(const s16x8*)((const char*)lds + ((wab[a] ^ x) + woff));
#pragma unroll
        for (int b = 0; b < FP; ++b) bf4[kk][b] = *(const s16x8*)((const char*)lds + ((xab[b] ^ x) + xoff));
      }
      // requests of this slot: the weight tile two stages ahead, one patch piece of the next chunk
      int tap2 = tap + 2, cc2 = cc;
      if (tap2 >= 9) { tap2 -= 9; ++cc2; }
      const bool w_new = cc2 < cpt;
      const int wn = ws == 0 ? 2 : ws - 1;             // (ws + 2) % 3
      if (w_new) stage_w(wn, tap2, cc2);
      const bool x_new = more_c && tap < XPW && tap * NW + wave < PMAX;
      if (more_c && tap < XPW) stage_x((cc + 1) & 1, tap, cc + 1);
      wait_lgkmcnt<0>();
      // everything older than this slot's requests has landed (the next stage's weight tile, earlier patch pieces)
      if (w_new) { if (x_new) wait_vmcnt<NBW + 1>(); else wait_vmcnt<NBW>(); }
      else { if (x_new) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
      vq_sched_fence();
      raw_barrier();
      vq_sched_fence();
      vq_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
          for (int b = 0; b < FP; ++b) acc[a][b] = mfma16<DT>(af4[kk][a], bf4[kk][b], acc[a][b]);
      vq_setprio(0);
      vq_sched_fence();
      raw_barrier();
      vq_sched_fence();
      ws = ws == 2 ? 0 : ws + 1;
    }
  }
  if (grp == 0) raw_barrier();
  igemm_epilogue<DT, BC, BP, WC, WP, 1>(p, lds, acc, c0, p0, wc0, wp0);
}

// ------------------------------------------------------------------------------ 128 x 512 tile over a staged 32 x 16 patch
// The patch-staged tile for Cout = 128 layers with the WAVE SHAPE that made conv_igemm_p9_kernel fast: 8 waves x (128c x 64p) — four
// weight + two pixel fragment reads per 8 MFMAs.  128 rows x 512 pixels only fits LDS with 32-channel chunks: rows are 64 bytes,
// four 16-byte slots, swizzled by (row >> 2) & 3 (16 consecutive rows of one logical slot cover the 16 bank groups: conflict-free
// ds_read_b128; an LDS-DMA piece is 16 rows).  The tile's pixels are a 32 x 16 patch of one image + halo (34 x 18 = 612 rows, two
// buffers); a stage is (32-channel chunk, tap): 128 x 32 weights (8 KiB, three buffers: the tile of stage s + 2 is requested in
// the load slot of stage s, counted vmcnt) and 2 k-steps = 16 MFMAs per wave, one load slot + one matrix slot of the ping-pong
// schedule.  LDS: 3 x 8 + 2 x 39 KiB (128 KiB reserved for the epilogue transposition).
template <int DT>
__global__ __launch_bounds__(512) void conv_igemm_p12_kernel(const ConvParams p) {
  constexpr int BK = 32, BC = 128, BP = 512, WC = 128, WP = 64, NWB = 3;
  constexpr int FC = WC / 32, FP = WP / 32, NW = 8;
  constexpr int TW = 16, TH = BP / TW, HWD = TW + 2, NSLOT = (TH + 2) * HWD;     // 612 halo rows
  constexpr int PMAX = (NSLOT + 15) / 16;              // 39 pieces of 16 rows (1 KiB)
  constexpr int WT = BC * BK, XT = PMAX * 16 * BK;     // elements per weight / patch buffer
  constexpr int XBASE = NWB * WT;
  constexpr int XPW = (PMAX + NW - 1) / NW;            // patch pieces per wave and chunk (5)
  static_assert(BC / 16 == NW, "one weight piece per wave and stage");

  VQ_DYN_LDS(vq_bf16, lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc0 = 0, wp0 = wave * WP;
  const int nblk = p.n_ctiles * p.n_ptiles;
  int t;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ctile = t % p.n_ctiles, ptile = t / p.n_ctiles;
  const int c0 = ctile * BC, p0 = ptile * BP;
  const int pn = ptile / p.pt_tpi, prem = ptile - pn * p.pt_tpi, ptyi = prem / p.pt_tx;
  const int ty0 = ptyi * TH, tx0 = (prem - ptyi * p.pt_tx) * TW;

  const int Hv = p.d.H << p.ush, Wv = p.d.W << p.ush;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;
  const int lr = lane >> 2, lp = lane & 3;             // row within a 16-row DMA piece, physical 16-byte slot
  const int cpt = p.d.Cin >> 5;                        // 32-channel chunks

  // ---- the weight row of this lane in piece `wave` of the 128-row tile ------------------------------------------------
  const vq_bf16* pb;
  {
    const int row = wave * 16 + lr;
    int grow = c0 + row;
    if (grow >= p.d.Cout) grow = p.d.Cout - 1;
    pb = p.w + (int64_t)grow * p.Kp + ((lp ^ ((row >> 2) & 3)) << 3);
  }
  auto stage_w = [&](int wbuf, int tap, int cc) {
    glds16(pb + (tap * p.d.Cin + cc * BK), lds + wbuf * WT + wave * 16 * BK);
  };
  // patch pieces of this wave: piece i * NW + wave, element offset in x at chunk 0 or -1 (zero page), see conv_igemm_p9_kernel
  int xo[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int slot = (i * NW + wave) * 16 + lr;
    const int lsa = (lp ^ ((slot >> 2) & 3)) << 3;
    const int hy = slot / HWD, hx = slot - hy * HWD;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = slot < NSLOT && (unsigned)ix < (unsigned)Wv && (unsigned)iy < (unsigned)Hv;
    xo[i] = ok ? ((pn * p.d.H + (iy >> p.ush)) * p.d.W + (ix >> p.ush)) * p.d.Cin + lsa : -1;
  }
  auto stage_x = [&](int xbuf, int i, int cc) {        // i compile-time after unrolling
    const int j = i * NW + wave;
    if (j < PMAX) {                                    // wave-uniform
      const int slot = j * 16 + lr;
      const int lsa = (lp ^ ((slot >> 2) & 3)) << 3;
      const vq_bf16* src = xo[i] >= 0 ? xbase + (int64_t)xo[i] + cc * BK : zero + lsa;
      glds16((const void*)src, lds + XBASE + xbuf * XT + j * 16 * BK);
    }
  };

  f32x16 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int b = 0; b < FP; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- fragment byte addresses at k-step 0: k-step 1 enters by XOR 32 (slot (2 kk | fh) ^ key), buffers by ADD ------------------
  const int fr = lane & 31, fh = lane >> 5;
  unsigned wab[FC];
#pragma unroll
  for (int a = 0; a < FC; ++a) {
    const int row = wc0 + a * 32 + fr;
    wab[a] = (unsigned)(row * BK * 2 + ((fh ^ ((row >> 2) & 3)) << 4));
  }
  int row0[FP];
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int p_l = wp0 + b * 32 + tap9_perm(fr);
    row0[b] = (p_l / TW) * HWD + (p_l % TW);
  }
  unsigned xab[FP];
  auto set_tap = [&](int tap) {
#pragma unroll
    for (int b = 0; b < FP; ++b) {
      int row = row0[b];
#ifndef VQ_EMU
      asm volatile("" : "+v"(row));                    // opaque: nine taps' addresses must not be hoisted into registers
#endif
      row += (tap / 3) * HWD + (tap % 3);
      xab[b] = (unsigned)(XBASE * 2 + row * BK * 2 + ((fh ^ ((row >> 2) & 3)) << 4));
    }
  };

  // ---- prologue: weight tiles of stages 0 and 1, the whole patch of chunk 0 -------------------------------------------------
  stage_w(0, 0, 0);
  stage_w(1, 1, 0);
#pragma unroll
  for (int i = 0; i < XPW; ++i) stage_x(0, i, 0);
  wait_vmcnt<0>();
  raw_barrier();
  const int grp = wave >> 2;
  if (grp == 1) raw_barrier();
  s16x8 af[2][FC], bfr[2][FP];
  int ws = 0;                                          // weight buffer of the current stage (stage index mod 3)
  for (int cc = 0; cc < cpt; ++cc) {
    const bool more_c = cc + 1 < cpt;
    const unsigned xoff = (unsigned)((cc & 1) * XT * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const unsigned woff = (unsigned)(ws * WT * 2);
      set_tap(tap);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const unsigned x = (unsigned)(kk << 5);
#pragma unroll
        for (int a = 0; a < FC; ++a) af[kk][a] = *(const s16x8*)((const char*)lds + ((wab[a] ^ x) + woff));
#pragma unroll
        for (int b = 0; b < FP; ++b) bfr[kk][b] = *(const s16x8*)((const char*)lds + ((xab[b] ^ x) + xoff));
      }
      int tap2 = tap + 2, cc2 = cc;
      if (tap2 >= 9) { tap2 -= 9; ++cc2; }
      const bool w_new = cc2 < cpt;
      const int wn = ws == 0 ? 2 : ws - 1;             // (ws + 2) % 3
      if (w_new) stage_w(wn, tap2, cc2);
      const bool x_new = more_c && tap < XPW && tap * NW + wave < PMAX;
      if (more_c && tap < XPW) stage_x((cc + 1) & 1, tap, cc + 1);
      wait_lgkmcnt<0>();
      // everything older than this slot's requests has landed (the next stage's weight tile, earlier patch pieces)
      if (w_new) { if (x_new) wait_vmcnt<2>(); else wait_vmcnt<1>(); }
      else { if (x_new) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
      vq_sched_fence();
      raw_barrier();
      vq_sched_fence();
      vq_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
          for (int b = 0; b < FP; ++b) acc[a][b] = mfma16<DT>(af[kk][a], bfr[kk][b], acc[a][b]);
      vq_setprio(0);
      vq_sched_fence();
      raw_barrier();
      vq_sched_fence();
      ws = ws == 2 ? 0 : ws + 1;
    }
  }
  if (grp == 0) raw_barrier();
  igemm_epilogue<DT, BC, BP, WC, WP, 1>(p, lds, acc, c0, p0, wc0, wp0);
}

template <int DT>
static int launch_p9s(ConvParams& p, hipStream_t stream) {
  constexpr int BC = 128, BP = 256, NW = 8;
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != NW)) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, NW); return VQ_ERR_UNSUPPORTED; }
  constexpr int PMAX = (18 * 18 + 7) / 8;
  constexpr size_t LDS_BYTES = (size_t)3 * BC * 64 * sizeof(vq_bf16) + (size_t)2 * PMAX * 8 * 64 * sizeof(vq_bf16);
  static_assert(LDS_BYTES >= (size_t)BP * BC * sizeof(vq_bf16) && LDS_BYTES <= 160 * 1024, "epilogue transpose / LDS capacity");
  if ((int64_t)p.d.N * p.d.H * p.d.W * p.d.Cin >= ((int64_t)1 << 31)) { vq_set_error("vq_conv2d_fwd(p9s): input of 2^31 elements or more"); return VQ_ERR_UNSUPPORTED; }
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = p.M / BP;
  p.pt_tx = p.d.Wo / 16;
  p.pt_tpi = p.pt_tx * (p.d.Ho / 16);
  const int grid = p.n_ctiles * p.n_ptiles;
#ifndef VQ_EMU
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_p9s_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) { vq_set_error("vq_conv2d_fwd: cannot reserve %zu B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return VQ_ERR_HIP; }
    attr_set = true;
  }
#endif
  hipLaunchKernelGGL((conv_igemm_p9s_kernel<DT>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(p9s)");
  return VQ_OK;
}
template <int DT>
static int launch_p12(ConvParams& p, hipStream_t stream) {
  constexpr int BC = 128, BP = 512, NW = 8;
  if (p.gn_part && (p.gn_bp != BP || p.gn_nw != NW)) { vq_set_error("vq_conv2d_fwd: GroupNorm partial tile %d x %d rows != kernel tile %d pixels x %d waves", p.gn_bp, p.gn_nw, BP, NW); return VQ_ERR_UNSUPPORTED; }
  constexpr size_t LDS_BYTES = (size_t)BP * BC * sizeof(vq_bf16);      // the epilogue transposition; the main loop uses 102 KiB of it
  static_assert(LDS_BYTES >= (size_t)3 * BC * 32 * 2 + (size_t)2 * 39 * 16 * 32 * 2 && LDS_BYTES <= 160 * 1024, "LDS budget");
  if ((int64_t)p.d.N * p.d.H * p.d.W * p.d.Cin >= ((int64_t)1 << 31)) { vq_set_error("vq_conv2d_fwd(p12): input of 2^31 elements or more"); return VQ_ERR_UNSUPPORTED; }
  p.n_ctiles = (int)vq_ceil_div(p.d.Cout, BC);
  p.n_ptiles = p.M / BP;
  p.pt_tx = p.d.Wo / 16;
  p.pt_tpi = p.pt_tx * (p.d.Ho / 32);
  const int grid = p.n_ctiles * p.n_ptiles;
#ifndef VQ_EMU
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_p12_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) { vq_set_error("vq_conv2d_fwd: cannot reserve %zu B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return VQ_ERR_HIP; }
    attr_set = true;
  }
#endif
  hipLaunchKernelGGL((conv_igemm_p12_kernel<DT>), dim3(grid), dim3(NW * 64), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(p12)");
  return VQ_OK;
}
