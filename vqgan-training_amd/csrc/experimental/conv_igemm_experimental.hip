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

// ------------------------------------------------------------------------------ 64 -> 64 channels: resident weights
// The 64-channel 3x3 layers (VGG conv1_2 under LPIPS and the discriminator, forward and data gradient: 1.3 ms of the step at
// 0.21-0.23 of the MFMA peak on the 64-row nine-tap tile) are bound by their WEIGHT stream, not by pixels: all of K = 576 is one
// 64-channel chunk, so a 128-pixel tile fetches 73.7 KB of weight fragments for 23 KB of halo patch, every wave its own copy —
// 128 B/clk/CU at full MFMA rate through a vector-memory return path of 64 B/clk/CU (the 2.4 GB per launch of §6).  Here the
// weights never move again after the first microsecond: one block per CU, ONE wave per SIMD, each wave holds all 72 weight
// fragments of the layer (64 cout x 576: 288 of the 512 registers a lone wave owns) and the block walks a contiguous range of
// 16 x 16-pixel patches: per patch one 41-KiB halo tile by LDS-DMA (two buffers: the next patch lands under this one's MFMAs),
// 36 (tap, k-step) steps of 4 MFMAs per wave (64 cout x 64 pixels) over 2 pixel-fragment reads — half an LDS fragment per MFMA,
// nothing from global memory inside the loop — then the shared epilogue through a 32-KiB slab of its own.
// LDS: X0 at 0, X1 at 64 KiB (one xor switches buffers), slab behind X1: 137 KiB.
template <int DT, int DBG = 0>   // DBG (make ABLATE=1 only, wrong results): 1 = the halo tile of the first patch serves all, 2 = and no MFMAs
__global__ __launch_bounds__(256) void conv_igemm_c64_kernel(const ConvParams p) {
  constexpr int BK = 64, BC = 64, BP = 256, WC = 64, WP = 64, FC = 2, FP = 2, NW = 4;
  constexpr int TW = 16, HWD = TW + 2, NSLOT = HWD * HWD;
  constexpr int PMAX = (NSLOT + 7) / 8;                // 41 eight-row DMA pieces
  constexpr int PPW = (PMAX + NW - 1) / NW;            // 11 per wave
  constexpr int XSTRIDE = 32768;                       // elements between the two halo buffers (64 KiB)
  constexpr int SLAB = XSTRIDE + PMAX * 8 * BK;        // first element of the epilogue slab
  constexpr int NREG = 7;                              // taps whose weights stay in registers
  constexpr int WLDS = SLAB + BP * BC;                 // the other taps' fragments
  static_assert(PPW <= 36, "one DMA piece per (tap, k-step)");
  VQ_DYN_LDS(vq_bf16, lds);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float alpha_raw = conv_alpha_request(p);   // (consumed after the first tile wait: conv_alpha_finish)
  const int wp0 = wave * WP;
  // contiguous patch ranges per block, the blocks of one XCD (blockIdx % 8) next to each other: neighbouring patches share halo
  // rows in that XCD's L2
  const int G = gridDim.x, lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int t_begin = (int)((int64_t)lb * p.n_ptiles / G), t_end = (int)((int64_t)(lb + 1) * p.n_ptiles / G);
  if (t_begin >= t_end) return;

  const int Hv = p.d.H, Wv = p.d.W;
  const vq_bf16* zero = (const vq_bf16*)g_vq_zero_page;
  const vq_bf16* xbase = (const vq_bf16*)p.x;
  const int lr = lane >> 3, lp = lane & 7;

  // piece i of patch `ptile` (this wave's share: pieces wave + NW * i): lane (lr, lp) fetches 16 bytes of halo row slot
  // patch -> first pixel of its halo tile: (image * H + ty0 - 1) * W + tx0 - 1 as a pixel index, and (ty0 - 1, tx0 - 1) for the
  // border tests; once per patch (two divisions by run-time extents), not once per DMA piece
  int nx_ty = 0, nx_tx = 0, nx_img = 0;
  auto locate = [&](int ptile) {
    const int pn = ptile / p.pt_tpi, prem = ptile - pn * p.pt_tpi, ptyi = prem / p.pt_tx;
    nx_ty = ptyi * TW - 1; nx_tx = (prem - ptyi * p.pt_tx) * TW - 1; nx_img = pn * p.d.H;
  };
  auto stage_piece = [&](int buf, int i) {
    if (wave + NW * i >= PMAX) return;
    const int slot = (wave + NW * i) * 8 + lr;
    const int lsa = (lp ^ ((slot >> 1) & 7)) << 3;
    const int hy = slot / HWD, hx = slot - hy * HWD;
    const int iy = nx_ty + hy, ix = nx_tx + hx;
    const int ok = (int)(slot < NSLOT) & (int)((unsigned)ix < (unsigned)Wv) & (int)((unsigned)iy < (unsigned)Hv);
    const int64_t off = (int64_t)((nx_img + iy) * p.d.W + ix) * BK + lsa;
    const uintptr_t a_ok = (uintptr_t)(xbase + off), a_zero = (uintptr_t)(zero + lsa);
    glds16_asm((const void*)(ok ? a_ok : a_zero), lds + buf * XSTRIDE + (wave + NW * i) * 8 * BK);
  };

  // ---- the layer's weights: fragment-order packed layout (pack_weight_kernel layout 1), [cout block of 32][k block of 16][lane][8].
  // Taps 0..NREG-1 live in registers; the last 9 - NREG taps in LDS (fragment-linear: conflict-free 16 B/lane reads), because 288
  // weight registers + 64 accumulators + the epilogue's temporaries do not fit in 512 (hipcc spilled 50-70 registers to scratch,
  // weight fragments among them, and re-loaded those INSIDE the loop).
  s16x8 wf[NREG][BK / 16][FC];
  auto wsrc = [&](int tap, int kk, int a) -> const s16x8* {
    return (const s16x8*)(p.w + ((int64_t)(a * (p.Kp >> 4) + tap * (BK / 16) + kk)) * 512 + lane * 8);
  };
#pragma unroll
  for (int tap = 0; tap < NREG; ++tap)
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int a = 0; a < FC; ++a) wf[tap][kk][a] = *wsrc(tap, kk, a);
  for (int f = wave; f < (9 - NREG) * (BK / 16) * FC; f += NW) {          // fragment f = ((tap - NREG) * 4 + kk) * FC + a
    const int a = f % FC, kk = (f / FC) % (BK / 16), tap = NREG + f / (FC * (BK / 16));
    *(s16x8*)(lds + WLDS + f * 512 + lane * 8) = *wsrc(tap, kk, a);
  }
  s16x8 wl[FC];
  auto wl_load = [&](int tap, int kk) {
#pragma unroll
    for (int a = 0; a < FC; ++a) wl[a] = *(const s16x8*)(lds + WLDS + (((tap - NREG) * (BK / 16) + kk) * FC + a) * 512 + lane * 8);
  };

  // ---- pixel fragments: pixel p_l = (ty, tx) of the patch, tap (kr, ks) -> halo row (ty + kr) * 18 + tx + ks; byte address of
  // (tap, fragment b) at k-step 0 in buffer 0; k-step kk and the buffer enter by one xor (see conv_igemm_tap9_kernel, WA bit 1)
  const int fr = lane & 31, fh = lane >> 5;
  unsigned abase[9][FP];
#pragma unroll
  for (int b = 0; b < FP; ++b) {
    const int p_l = wp0 + b * 32 + tap9_perm(fr);
    const int rowb = (p_l / TW) * HWD + (p_l % TW);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int row = rowb + (tap / 3) * HWD + (tap % 3);
      abase[tap][b] = (unsigned)(row * BK * 2 + ((fh ^ ((row >> 1) & 7)) << 4));
    }
  }
  s16x8 bfr[2][FP];
  auto frag_load = [&](int buf, int tap, int kk, int slot) {
    const unsigned x = (unsigned)((kk << 5) | (buf << 16));
#pragma unroll
    for (int b = 0; b < FP; ++b) bfr[slot][b] = *(const s16x8*)((const char*)lds + (abase[tap][b] ^ x));
  };

  locate(t_begin);
#pragma unroll
  for (int i = 0; i < PPW; ++i) stage_piece(0, i);
  wait_vmcnt<0>();
  const float alpha_s = conv_alpha_finish(p, alpha_raw);
  raw_barrier();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = DBG ? 0 : (t - t_begin) & 1;
    const bool more = t + 1 < t_end;
    if (more) locate(t + 1);
    f32x16 acc[FC][FP];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
      for (int b = 0; b < FP; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    VQ_STAMP(0);
    frag_load(buf, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 36; ++v) {                     // v = tap * 4 + kk
      const int tap = v >> 2, kk = v & 3;
      if (v + 1 < 36) frag_load(buf, (v + 1) >> 2, (v + 1) & 3, (v + 1) & 1);
      vq_sched_fence();
#pragma unroll
      for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) {
          if constexpr (DBG == 2) { acc[a][b][0] += (float)bfr[v & 1][b][0] + (float)(tap < NREG ? wf[tap < NREG ? tap : 0][kk][a] : wl[a])[1]; }
          else acc[a][b] = mfma16<DT>(tap < NREG ? wf[tap < NREG ? tap : 0][kk][a] : wl[a], bfr[v & 1][b], acc[a][b]);
        }
      vq_sched_fence();
      // the LDS-resident fragments of the next step, requested once this step's MFMAs (which read the same registers) are issued
      if (v + 1 < 36 && ((v + 1) >> 2) >= NREG) wl_load((v + 1) >> 2, (v + 1) & 3);
      if (DBG == 0 && v < PPW && more) stage_piece(buf ^ 1, v);   // the next patch's halo tile, one piece per step
    }
    // this wave's pieces of the next patch (requested >= 25 steps ago) and the previous patch's output stores: long done.  The
    // epilogue's barrier then publishes them to the other waves and tells this one that nobody reads the current buffer any more.
    VQ_STAMP(1);
    wait_vmcnt<0>();
    VQ_STAMP(2);
    igemm_epilogue<DT, BC, BP, WC, WP, 1, 2>(p, lds + SLAB, acc, 0, t * BP, 0, wp0, alpha_s);
    VQ_STAMP(7);
    raw_barrier();                                     // the slab is free again; buffer `buf` may be overwritten
    VQ_STAMP(8);
  }
}

// conv_igemm_c64_kernel: 3x3 / stride 1 / pad 1, exactly 64 -> 64 channels, images that split into 16 x 16 patches, at least two
// patches per CU (a persistent block wants a range to walk)
static bool c64_ok(const VqConvDesc* d, bool forced) {
  return d->Cin == 64 && d->Cout == 64 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil_in == 1 && d->pad_t == 1 && d->pad_l == 1 &&
         d->up == 1 && d->Ho == d->H && d->Wo == d->W && d->Wo % 16 == 0 && d->Ho % 16 == 0 && d->subpix == 0 &&
         (forced || (int64_t)d->N * d->Ho * d->Wo >= (int64_t)512 * 256);
}
template <int DT, int DBG = 0>
static int launch_c64(ConvParams& p, hipStream_t stream) {
  constexpr int BP = 256, PMAX = (18 * 18 + 7) / 8;
  constexpr size_t LDS_BYTES = (size_t)65536 + (size_t)PMAX * 8 * 64 * sizeof(vq_bf16) + (size_t)BP * 64 * sizeof(vq_bf16) + (size_t)2 * 8 * 1024;   // + two taps of weights
  static_assert(LDS_BYTES <= 160 * 1024, "LDS capacity");
  if ((int64_t)p.d.N * p.d.H * p.d.W * p.d.Cin >= ((int64_t)1 << 31)) { vq_set_error("vq_conv2d_fwd(c64): input of 2^31 elements or more"); return VQ_ERR_UNSUPPORTED; }
  p.n_ctiles = 1;
  p.n_ptiles = p.M / BP;
  p.pt_tx = p.d.Wo / 16;
  p.pt_tpi = p.pt_tx * (p.d.Ho / 16);
  // one block per CU, 8 | grid; small launches (tests) still give every block a range of patches to walk
  const int grid = std::min(256, std::max(8, p.n_ptiles / 16 * 8));
#ifndef VQ_EMU
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_c64_kernel<DT, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) { vq_set_error("vq_conv2d_fwd: cannot reserve %zu B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return VQ_ERR_HIP; }
    attr_set = true;
  }
#endif
  hipLaunchKernelGGL((conv_igemm_c64_kernel<DT, DBG>), dim3(grid), dim3(256), LDS_BYTES, stream, p);
  VQ_CHECK_LAUNCH("vq_conv2d_fwd(c64)");
  return VQ_OK;
}
