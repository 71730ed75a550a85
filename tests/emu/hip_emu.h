// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// A tiny single-header "HIP on the CPU" shim used by the `-m "not gpu"` tests to run
// the *same* kernel sources (vqgan-training_amd/csrc/*.hip) on host cores so that the
// index math, the LDS choreography and the host-side logic can be checked in a
// container that has no GPU.  Every HIP thread is a fiber; a workgroup's fibers run
// round-robin on one OS thread; __syncthreads()/wave collectives are cooperative
// yields.  MFMA / cross-lane / LDS-transpose builtins are emulated as wave collectives
// following the gfx950 register layouts given in /opt/skills/guides (cdna_hip_programming
// §3).  Whether those layouts match the silicon is checked on the GPU by
// tests/test_hw_layout.py — the emulator only checks OUR code against OUR reading.
#pragma once
#define VQ_EMU 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

namespace emu {
struct Fiber;
struct WaveState {
  unsigned arrived = 0, gen = 0, alive = 0;
  alignas(16) unsigned char scratch[64][64];
};
struct BlockState;
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  emu_uint3 tid{0, 0, 0};
  unsigned linear = 0;
  BlockState* blk = nullptr;
};
Fiber* cur();
emu_uint3& block_idx();
emu_uint3& block_dim();
emu_uint3& grid_dim();
void syncthreads();
void wave_barrier();
WaveState* wave();
unsigned lane();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::block_idx())
#define blockDim (emu::block_dim())
#define gridDim (emu::grid_dim())
#define __syncthreads() emu::syncthreads()
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(grid, block, [&]() { kernel(__VA_ARGS__); })

// ---------------------------------------------------------------- vector types
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }

// ---------------------------------------------------------------- math
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }

// ---------------------------------------------------------------- atomics
static inline float atomicAdd(float* p, float v) {
  unsigned* up = (unsigned*)p;
  unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED), neu;
  float f;
  do {
    memcpy(&f, &old, 4);
    float nf = f + v;
    memcpy(&neu, &nf, 4);
  } while (!__atomic_compare_exchange_n(up, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

// ---------------------------------------------------------------- cross-lane
template <typename T>
static inline T emu_exchange(T v, unsigned src_lane) {
  static_assert(sizeof(T) <= 64, "");
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &v, sizeof(T));
  emu::wave_barrier();
  T r;
  memcpy(&r, w->scratch[src_lane & 63], sizeof(T));
  emu::wave_barrier();
  return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  unsigned l = emu::lane();
  unsigned src = l ^ (unsigned)mask;
  if ((src / width) != (l / width)) src = l;
  return emu_exchange(v, src);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  unsigned l = emu::lane();
  unsigned src = l + delta;
  if ((src / width) != (l / width)) src = l;
  return emu_exchange(v, src);
}
template <typename T>
static inline T __shfl(T v, int srcLane, int width = 64) {
  unsigned l = emu::lane();
  unsigned src = (l / width) * width + ((unsigned)srcLane % width);
  return emu_exchange(v, src);
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu_exchange(v, 0); }

static inline float emu_bf16_to_f32(short s) {
  unsigned u = ((unsigned)(unsigned short)s) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// D[32x32] += A[32x16] * B[16x32]; lane l holds A[l&31][8*(l>>5)+t], B[8*(l>>5)+t][l&31];
// D reg r of lane l: row (r&3)+8*(r>>2)+4*(l>>5), col l&31.   (guide §3 "Fragment layout")
static inline f32x16 emu_mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &a, 16);
  memcpy(w->scratch[l] + 16, &b, 16);
  emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int j = l & 31;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      short av, bv;
      memcpy(&av, w->scratch[i + 32 * (k >> 3)] + 2 * (k & 7), 2);
      memcpy(&bv, w->scratch[j + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
      acc += emu_bf16_to_f32(av) * emu_bf16_to_f32(bv);
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
static inline float emu_f16_to_f32(short s) {
  _Float16 h;
  memcpy(&h, &s, 2);
  return (float)h;
}
// the same shape and layouts with binary16 operands (guide §3: layouts are dtype-independent)
static inline f32x16 emu_mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &a, 16);
  memcpy(w->scratch[l] + 16, &b, 16);
  emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int j = l & 31;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      short av, bv;
      memcpy(&av, w->scratch[i + 32 * (k >> 3)] + 2 * (k & 7), 2);
      memcpy(&bv, w->scratch[j + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
      acc += emu_f16_to_f32(av) * emu_f16_to_f32(bv);
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
// D[16x16] += A[16x32] * B[32x16]; lane l holds A[l&15][8*(l>>4)+t], B[8*(l>>4)+t][l&15];
// D reg r: row 4*(l>>4)+r, col l&15.
static inline f32x4 emu_mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &a, 16);
  memcpy(w->scratch[l] + 16, &b, 16);
  emu::wave_barrier();
  for (int r = 0; r < 4; ++r) {
    int i = 4 * (l >> 4) + r;
    int j = l & 15;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      short av, bv;
      memcpy(&av, w->scratch[i + 16 * (k >> 3)] + 2 * (k & 7), 2);
      memcpy(&bv, w->scratch[j + 16 * (k >> 3)] + 16 + 2 * (k & 7), 2);
      acc += emu_bf16_to_f32(av) * emu_bf16_to_f32(bv);
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
// f32-input MFMA 16x16x4: lane l holds A[l&15][l>>4], B[l>>4][l&15]; D as the 16x16 form.
static inline f32x4 emu_mfma_16x16x4_f32(float a, float b, f32x4 c) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &a, 4);
  memcpy(w->scratch[l] + 4, &b, 4);
  emu::wave_barrier();
  for (int r = 0; r < 4; ++r) {
    int i = 4 * (l >> 4) + r;
    int j = l & 15;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, w->scratch[i + 16 * k], 4);
      memcpy(&bv, w->scratch[j + 16 * k] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
// f32-input MFMA 32x32x2: lane l holds A[l&31][l>>5], B[l>>5][l&31]; D as the 32x32 forms.  Exact f32: per output element an fmaf
// chain over k ascending onto the accumulator (the guides: "exact f32 (≡ an fmaf chain, bitwise)"; pinned to silicon by
// tests/test_hw_layout.py probe 5 on operands whose sum depends on the order and on the fusing).
static inline f32x16 emu_mfma_32x32x2_f32(float a, float b, f32x16 c) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], &a, 4);
  memcpy(w->scratch[l] + 4, &b, 4);
  emu::wave_barrier();
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int j = l & 31;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, w->scratch[i + 32 * k], 4);
      memcpy(&bv, w->scratch[j + 32 * k] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
// ds_read_b64_tr_b16: every lane fetches 8 bytes (4 x b16) at its own LDS address; within
// each 16-lane group the 16x4 block is transposed: result lane c, element j =
// element (c & 3) of the chunk fetched by lane (4*j + (c >> 2)) of the group.
static inline s16x4 emu_ds_read_tr16_b64(const short* p) {
  emu::WaveState* w = emu::wave();
  unsigned l = emu::lane();
  memcpy(w->scratch[l], p, 8);
  emu::wave_barrier();
  unsigned g = l & ~15u, c = l & 15u;
  s16x4 r;
  for (int j = 0; j < 4; ++j) {
    short v;
    memcpy(&v, w->scratch[g + 4 * j + (c >> 2)] + 2 * (c & 3), 2);
    r[j] = v;
  }
  emu::wave_barrier();
  return r;
}
