// Hardware-layout probes: one wave executes a single MFMA / LDS-transpose-read with lane-linear
// operands and dumps the raw per-lane results.  tests/test_hw_layout.py runs the same probe through
// the host emulator (tests/emu) and on the GPU and requires identical output — that pins the
// register-layout reading (guide §3) that conv_igemm.hip / conv_wgrad.hip are written against.
#include "vq_common.h"

// which = 0: mfma_f32_32x32x16_bf16   in: a[64][8] bf16, b[64][8] bf16      out: c[64][16] f32
// which = 1: mfma_f32_16x16x32_bf16   in: same                               out: c[64][4]  f32
// which = 3: mfma_f32_32x32x16_f16    in: a[64][8] binary16, b[64][8] binary16  out: c[64][16] f32
// which = 4: v_permlane32_swap_b32 (vq_swap32)   in: a[64], b[64] uint32       out: a'[64], b'[64]
// which = 5: mfma_f32_32x32x2_f32      in: a[64], b[64], c[64][16] f32             out: d[64][16] f32
// which = 2: ds_read_b64_tr_b16       in: lds image short[1024], then per-lane element offsets
//                                         int[64] (as 2 shorts each, appended)  out: short[64][4]
__global__ __launch_bounds__(64) void debug_probe_kernel(int which, const short* __restrict__ in, float* __restrict__ out) {
  const int lane = threadIdx.x;
  if (which == 0 || which == 1 || which == 3) {
    s16x8 a, b;
#pragma unroll
    for (int t = 0; t < 8; ++t) { a[t] = in[lane * 8 + t]; b[t] = in[512 + lane * 8 + t]; }
    if (which == 0 || which == 3) {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      c = which == 3 ? mfma_32x32x16_f16(a, b, c) : mfma_32x32x16_bf16(a, b, c);
#pragma unroll
      for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
    } else {
      f32x4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = 0.f;
      c = mfma_16x16x32_bf16(a, b, c);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
    }
  } else if (which == 5) {                           // mfma_f32_32x32x2_f32: in = a[64], b[64], c[64][16] f32; out = d[64][16] f32
    const float* f = (const float*)in;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = f[128 + lane * 16 + r];
    c = mfma_32x32x2_f32(f[lane], f[64 + lane], c);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
  } else if (which == 4) {                           // v_permlane32_swap_b32: in = a[64], b[64] as uint32; out = a'[64], b'[64]
    const unsigned* u = (const unsigned*)in;
    unsigned a = u[lane], b = u[64 + lane];
    vq_swap32(a, b);
    unsigned* o = (unsigned*)out;
    o[lane] = a; o[64 + lane] = b;
  } else {
    __shared__ __attribute__((aligned(16))) short img[1024];
    for (int i = lane; i < 1024; i += 64) img[i] = in[i];
    __syncthreads();
    const int off = ((const int*)(in + 1024))[lane];
    s16x4 v = lds_read_tr16_b64(img + off);
    short* o = (short*)out;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[lane * 4 + j] = v[j];
  }
}

extern "C" int vq_debug_probe(int which, const void* in, void* out, void* stream) {
  VQ_REQUIRE(in && out && which >= 0 && which <= 5, VQ_ERR_INVALID, "vq_debug_probe: bad arguments");
  hipLaunchKernelGGL(debug_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, which, (const short*)in, (float*)out);
  VQ_CHECK_LAUNCH("vq_debug_probe");
  return VQ_OK;
}
