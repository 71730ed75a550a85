// How much straight-line code does a CU run at full speed?  A loop whose body is N KiB of independent 4-byte VALU instructions
// (v_add_f32), 8 waves per block, one block per CU, every CU busy; cycles per instruction vs N.  The cliff is the instruction
// cache capacity as a kernel sees it (conv kernels of this repository are 30-46 KiB of code).
// hipcc --offload-arch=gfx950 -O1 tools/micro/icache.hip -o /tmp/icache && /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define I4 asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
#define I16 I4 I4 I4 I4
#define I64 I16 I16 I16 I16
#define I256 I64 I64 I64 I64          // 1 KiB of code
#define K4 I256 I256 I256 I256        // 4 KiB
template <int KB4> __device__ __forceinline__ void body(float& a, float& b, float& c, float& d, float e) {
  if constexpr (KB4 > 0) { K4 body<KB4 - 1>(a, b, c, d, e); }
}
template <int KB4>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f, e = 1e-3f;
  body<KB4>(a, b, c, d, e);                                   // warm
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) body<KB4>(a, b, c, d, e);
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
}
template <int KB4> void run(float* out, long long* cyc) {
  const int iters = 64;
  hipLaunchKernelGGL(k<KB4>, dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize();
  hipLaunchKernelGGL(k<KB4>, dim3(256), dim3(512), 0, 0, out, cyc, iters); hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  printf("%4d KiB of loop body: %6.2f cycles per instruction per wave (8 waves per CU)\n", KB4 * 4, s / 256 / iters / (KB4 * 1024.0));
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc); run<10>(out, cyc); run<12>(out, cyc);
  run<14>(out, cyc); run<16>(out, cyc); run<20>(out, cyc); run<24>(out, cyc); run<32>(out, cyc);
  return 0;
}
