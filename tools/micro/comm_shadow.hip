// TOOLS ONLY (tools/comm_shadow.py): a stand-in for RCCL's CU footprint on ONE GPU.  A ring all-reduce over xGMI keeps a fixed, small
// number of workgroups resident for as long as the links need (387 MB of gradients per step, ~1.75 S through one
// ~76.8 GB/s link direction => ~9 ms), each streaming its slice of the buffer through the CU.  This kernel does the part of that a
// single GPU can show: `blocks` persistent workgroups of 512 threads copy `bytes` from src to dst (read + write through HBM, like the
// reduce-scatter / all-gather halves), and — `pace_ns` > 0 — spread the copy over that many nanoseconds by sleeping between 64-KiB
// chunks, the way a link-bound ring waits for its peer.  What it cannot show: link contention and the receive path of the fabric.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/comm_shadow.hip -o build/tools/libcomm_shadow.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(512) void comm_shadow_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                           int64_t pace_ticks) {
  const int64_t per = (n16 + gridDim.x - 1) / gridDim.x;
  int64_t lo = per * blockIdx.x, hi = lo + per;
  if (hi > n16) hi = n16;
  const int64_t chunk = 4096;                      // 16-byte units per block per round: 64 KiB
  const int64_t rounds = (hi - lo + chunk - 1) / chunk;
  const int64_t t0 = wall_clock64();               // 100 MHz constant clock
  for (int64_t r = 0; r < rounds; ++r) {
    for (int64_t i = lo + r * chunk + threadIdx.x; i < lo + (r + 1) * chunk && i < hi; i += 512) {
      uint4 v = src[i];
      v.x += 1u;                                   // (a reduction stand-in: the copy must not be elided or turned into a DMA)
      dst[i] = v;
    }
    if (pace_ticks > 0) {                          // wait until this round's share of the paced duration has passed
      const int64_t due = t0 + pace_ticks * (r + 1) / rounds;
      while ((int64_t)wall_clock64() < due) __builtin_amdgcn_s_sleep(32);
    }
  }
}

extern "C" int comm_shadow_launch(const void* src, void* dst, int64_t bytes, int blocks, int64_t pace_ns, void* stream) {
  if (!src || !dst || bytes < 16 || blocks < 1) return -1;
  hipLaunchKernelGGL(comm_shadow_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes / 16,
                     pace_ns / 10);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
