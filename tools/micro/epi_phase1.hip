// Stand-alone timing of the conv epilogue's first phase (accumulators x scale -> saturate -> binary16 -> 8-byte LDS writes) as the
// patch-staged 256x256 tile runs it: 8 waves per block, one block per CU, 128 accumulator registers per lane, 32 groups of
// (4 mul, 4 med3, 2 cvt_pk, address, ds_write_b64).  Variants drop one ingredient at a time.  Prints cycles (s_memtime) of wave 0.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/epi_phase1.hip -o /tmp/epi1 && /tmp/epi1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b, bool sat) {
  if (sat) { a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f); }
  h2 v = {(_Float16)a, (_Float16)b};
  unsigned r; __builtin_memcpy(&r, &v, 4); return r;
}
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>   // 0 full, 1 no ds_write, 2 no med3, 3 no mul, 4 linear (conflict-free) LDS addresses, 5 accumulators produced by MFMAs right
                      // before every pass (128 MFMAs per wave, like a main loop's tail), 6 each 8-byte write its own ds_write_b64 (inline asm: no merging)
__global__ __launch_bounds__(512) void k(const float* in, long long* out, float alpha, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  float acc[128];
#pragma unroll
  for (int i = 0; i < 128; ++i) acc[i] = in[(tid * 128 + i) & 4095] * (1.f + i);
  const int wc0 = (wave >> 2) * 128, wp0 = (wave & 3) * 64;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter(), tfirst = 0;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 5 || MODE == 7) {
      f16v c[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[i][e] = acc[i * 16 + e];
      s8 fa, fb;
#pragma unroll
      for (int e = 0; e < 8; ++e) { fa[e] = (short)(0x3c00 + lane + e); fb[e] = (short)(0x3800 + e); }
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, fa), __builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, fb), c[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i * 16 + e] = c[i][e] * 1e-3f;
      t0 += 0;
    }
    if (MODE != 7)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co_l = wc0 + a * 32 + q * 8 + fh * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int p_l = wp0 + b * 32 + fr;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = MODE == 3 ? acc[(a * 2 + b) * 16 + q * 4 + e] : acc[(a * 2 + b) * 16 + q * 4 + e] * alpha;
          const unsigned lo = pk(v[0], v[1], MODE != 2), hi = pk(v[2], v[3], MODE != 2);
          int idx = p_l * 256 + (((co_l >> 3) ^ (p_l & 31)) << 3) + (co_l & 4);
          if (MODE == 4) idx = ((a * 4 + q) * 2 + b) * 2048 + tid * 4;
          if (MODE == 1) { if (lo == 0x12345678u) *(uint2*)(lds + idx) = make_uint2(lo, hi); }
          else if (MODE == 6) {
            const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned short*)(lds + idx);
            uint2 vv = make_uint2(lo, hi);
            asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(vv) : "memory");
          }
          else *(uint2*)(lds + idx) = make_uint2(lo, hi);
        }
      }
    __syncthreads();
    if (r == 0) tfirst = __builtin_readcyclecounter() - t0;
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) { out[blockIdx.x] = t1 - t0; out[256 + blockIdx.x] = tfirst; }
  float keep = 0.f;
#pragma unroll
  for (int i = 0; i < 128; ++i) keep += acc[i];
  if ((lds[tid] == 0x7777 && in[0] == 123.f) || keep == 123.456f) out[0] = 0;
}
template <int MODE> void run(const char* name, const float* in, long long* out) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int it = 0; it < 1; ++it) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, in, out, 0.37f, 8); hipDeviceSynchronize(); }
  long long h[512]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  long long s = 0, f = 0; for (int i = 0; i < 256; ++i) { s += h[i]; f += h[256 + i]; }
  printf("%-28s %8.0f cycles per phase (block average over 8 passes; 32 groups per lane, 8 waves); first pass of the launch %8.0f\n", name, (double)s / 256 / 8, (double)f / 256);
}
int main() {
  float* in; long long* out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 8);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 97) * 0.01f - 0.3f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("full", in, out); run<1>("no ds_write", in, out); run<2>("no med3", in, out); run<3>("no mul", in, out);
  run<4>("linear LDS addresses", in, out);
  run<5>("acc from 128 MFMAs per pass", in, out);
  run<6>("unmerged ds_write_b64", in, out);
  run<7>("128 MFMAs per pass only", in, out);
  return 0;
}
