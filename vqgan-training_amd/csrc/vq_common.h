// Shared device/host helpers for the libvqhip kernels (gfx950 / CDNA4 only).
#pragma once
#include <initializer_list>
#include <type_traits>
#ifndef VQ_EMU
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 vq_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 vq_f16x8 __attribute__((ext_vector_type(8)));
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vqhip.h"

#define VQ_WAVE 64

// ------------------------------------------------------------------ error plumbing (host)
void vq_set_error(const char* fmt, ...);
#define VQ_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      vq_set_error(__VA_ARGS__);     \
      return (code);                 \
    }                                \
  } while (0)
#define VQ_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      vq_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return VQ_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

// ---- the three things the kernel sources need from "am I the host emulation of tests/emu?" live HERE, so that the .hip files carry no
// VQ_EMU conditionals of their own (the emulator is test infrastructure; the product is the gfx950 branch of each helper)
// VQ_RESERVE_LDS(kernel, bytes, what): opt a kernel into more than 64 KiB of dynamic LDS, once per instantiation (idempotent; the
// static flag's race is benign).  Statement macro for launchers that return an int status.
#ifdef VQ_EMU
#define VQ_RESERVE_LDS(kernel, bytes, what) ((void)0)
#define VQ_OPAQUE_VGPR(v) ((void)0)
#define VQ_CONSTANT static const
#else
#define VQ_RESERVE_LDS(kernel, bytes, what)                                                                              \
  do {                                                                                                                   \
    static bool attr_set__ = false;                                                                                      \
    if (!attr_set__) {                                                                                                   \
      hipError_t e__ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
      if (e__ != hipSuccess) {                                                                                           \
        vq_set_error("%s: cannot reserve %zu B of LDS: %s", what, (size_t)(bytes), hipGetErrorString(e__));              \
        return VQ_ERR_HIP;                                                                                               \
      }                                                                                                                  \
      attr_set__ = true;                                                                                                 \
    }                                                                                                                    \
  } while (0)
// an integer the optimiser must treat as unknown from here on: keeps per-tap fragment addresses from being hoisted into registers
#define VQ_OPAQUE_VGPR(v) asm volatile("" : "+v"(v))
#define VQ_CONSTANT __device__ __constant__
#endif
// profiling ablations only: a pair of fragment registers whose loads must survive although nothing consumes them
#ifdef VQ_EMU
#define VQ_KEEP_ALIVE2(a, b) ((void)0)
#else
#define VQ_KEEP_ALIVE2(a, b) asm volatile("" ::"v"(a), "v"(b))
#endif
// cycle stamps (tools only): ablation builds for the GPU
#if defined(VQ_ABLATION_KERNELS) && !defined(VQ_EMU)
#define VQ_STAMPS_ON 1
#endif

// ------------------------------------------------------------------ bf16 helpers
typedef unsigned short vq_bf16;  // raw bfloat16 bits

__device__ __forceinline__ float bf2f(vq_bf16 h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN payloads are not preserved; inputs here are finite).  On gfx950 this is ONE instruction
// per PAIR (v_cvt_pk_bf16_f32, which hipcc selects for a float2 -> __bf16 x 2 vector conversion); rounds 1-2 did the rounding
// with integer arithmetic — add3 / bfe / and / or: ~4 VALU instructions per element, a third of the conv epilogues' VALU work
// (profiles/r3f_c64_ablations.txt: the epilogue of a 64-channel tile costs as much as its MFMAs).  The emulator keeps the integer form.
#ifdef VQ_EMU
__device__ __forceinline__ vq_bf16 f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (vq_bf16)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
#else
typedef __bf16 vq_hwbf2 __attribute__((ext_vector_type(2)));
typedef float vq_hwf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const vq_hwf2 v = {lo, hi};
  const vq_hwbf2 r = __builtin_convertvector(v, vq_hwbf2);
  unsigned u; __builtin_memcpy(&u, &r, 4); return u;
}
__device__ __forceinline__ vq_bf16 f2bf(float f) { return (vq_bf16)(pack_bf2(f, 0.f) & 0xffffu); }
#endif

// ------------------------------------------------------------------ fp16 helpers (VQ_F16 storage: the "ref" precision)
// IEEE binary16, round-to-nearest-even, SATURATING at +-65504 (an overflowing value must not become inf and poison a
// GroupNorm statistic; the grad-scale calibration of ops.py watches the tensor maxima).  v_cvt_pk_f16_f32 / v_cvt_f32_f16.
typedef unsigned short vq_f16;   // raw binary16 bits
typedef _Float16 vq_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vq_sat16(float f) {
#ifdef VQ_EMU
  return f != f ? f : (f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f));
#else
  return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);
#endif
}
__device__ __forceinline__ float h2f(vq_f16 h) { _Float16 v; __builtin_memcpy(&v, &h, 2); return (float)v; }
__device__ __forceinline__ vq_f16 f2h(float f) { const _Float16 v = (_Float16)vq_sat16(f); vq_f16 r; __builtin_memcpy(&r, &v, 2); return r; }
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {
  vq_h2 v = {(_Float16)vq_sat16(lo), (_Float16)vq_sat16(hi)};
  unsigned r; __builtin_memcpy(&r, &v, 4); return r;
}
__device__ __forceinline__ void unpack_h2(unsigned u, float& lo, float& hi) {
  vq_h2 v; __builtin_memcpy(&v, &u, 4); lo = (float)v[0]; hi = (float)v[1];
}

struct __attribute__((aligned(16))) vq_u4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) vq_u2 { unsigned x, y; };
struct __attribute__((aligned(16))) vq_f4 { float x, y, z, w; };

// Asynchronous 16-byte global load: issued as inline asm the compiler does not track, so a BATCH of them is really in
// flight together (compiler-visible loads from `const __restrict__` memory get re-issued / narrowed next to each use, which
// left the streaming GroupNorm passes with one load in flight per lane).  The destination is valid only after vq_raw_wait().
typedef unsigned vq_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte streaming store / load (global_store/load_dwordx4 ... nt)
__device__ __forceinline__ void vq_store16_nt(void* p, vq_u32x4 v) {
#ifdef VQ_EMU
  *(vq_u32x4*)p = v;
#else
  __builtin_nontemporal_store(v, (vq_u32x4*)p);
#endif
}
__device__ __forceinline__ vq_u32x4 vq_load16_nt(const void* p) {
#ifdef VQ_EMU
  return *(const vq_u32x4*)p;
#else
  return __builtin_nontemporal_load((const vq_u32x4*)p);
#endif
}
__device__ __forceinline__ void vq_gload16_issue(vq_u32x4& dst, const void* p) {
#ifdef VQ_EMU
  dst = *(const vq_u32x4*)p;
#else
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
#endif
}

// Storage-type traits: 8 consecutive channels are the unit of every vectorised access.
// a wave-uniform float the compiler may keep in a scalar register
__device__ __forceinline__ float vq_wave_uniform(float v) {
#ifdef VQ_EMU
  return v;
#else
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
#endif
}
template <int DT> struct Store;
template <> struct Store<VQ_BF16> {
  typedef vq_bf16 T;
  static constexpr int BYTES = 2;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    vq_u4 q = *(const vq_u4*)((const vq_bf16*)base + elem);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
    v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
  }
  // asynchronous raw load (see vq_gload16_issue) and its unpacking
  static constexpr int RAWQ = 1;
  struct Raw { vq_u32x4 q[1]; };
  __device__ static __forceinline__ void load8_issue(Raw& r, const void* base, int64_t elem) {
    vq_gload16_issue(r.q[0], (const vq_bf16*)base + elem);
  }
  // the same request as an ordinary (compiler-tracked) load: for kernels whose register pressure makes asm destinations unsafe
  __device__ static __forceinline__ void load8_raw(Raw& r, const void* base, int64_t elem) {
    r.q[0] = *(const vq_u32x4*)((const vq_bf16*)base + elem);
  }
  __device__ static __forceinline__ void unpack8(const Raw& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.q[0].x << 16); v[1] = __uint_as_float(r.q[0].x & 0xffff0000u);
    v[2] = __uint_as_float(r.q[0].y << 16); v[3] = __uint_as_float(r.q[0].y & 0xffff0000u);
    v[4] = __uint_as_float(r.q[0].z << 16); v[5] = __uint_as_float(r.q[0].z & 0xffff0000u);
    v[6] = __uint_as_float(r.q[0].w << 16); v[7] = __uint_as_float(r.q[0].w & 0xffff0000u);
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    vq_u4 q;
    q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]);
    q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
    *(vq_u4*)((vq_bf16*)base + elem) = q;
  }
  // streaming ("non-temporal") forms: data this kernel never touches again must not evict what it re-reads from L2
  __device__ static __forceinline__ unsigned pack2(float a, float b) { return pack_bf2(a, b); }
  __device__ static __forceinline__ void store8_nt(void* base, int64_t elem, const float (&v)[8]) {
    vq_u32x4 q;
    q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]); q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
    vq_store16_nt((vq_bf16*)base + elem, q);
  }
  __device__ static __forceinline__ void load8_raw_nt(Raw& r, const void* base, int64_t elem) {
    r.q[0] = vq_load16_nt((const vq_bf16*)base + elem);
  }
  __device__ static __forceinline__ void load4(const void* base, int64_t elem, float (&v)[4]) {
    vq_u2 q = *(const vq_u2*)((const vq_bf16*)base + elem);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store4(void* base, int64_t elem, const float (&v)[4]) {
    vq_u2 q;
    q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]);
    *(vq_u2*)((vq_bf16*)base + elem) = q;
  }
  __device__ static __forceinline__ float load1(const void* base, int64_t elem) {
    return bf2f(((const vq_bf16*)base)[elem]);
  }
  __device__ static __forceinline__ void store1(void* base, int64_t elem, float v) {
    ((vq_bf16*)base)[elem] = f2bf(v);
  }
};
template <> struct Store<VQ_F16> {
  typedef vq_f16 T;
  static constexpr int BYTES = 2;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    vq_u4 q = *(const vq_u4*)((const vq_f16*)base + elem);
    unpack_h2(q.x, v[0], v[1]); unpack_h2(q.y, v[2], v[3]); unpack_h2(q.z, v[4], v[5]); unpack_h2(q.w, v[6], v[7]);
  }
  static constexpr int RAWQ = 1;
  struct Raw { vq_u32x4 q[1]; };
  __device__ static __forceinline__ void load8_issue(Raw& r, const void* base, int64_t elem) {
    vq_gload16_issue(r.q[0], (const vq_f16*)base + elem);
  }
  // the same request as an ordinary (compiler-tracked) load: for kernels whose register pressure makes asm destinations unsafe
  __device__ static __forceinline__ void load8_raw(Raw& r, const void* base, int64_t elem) {
    r.q[0] = *(const vq_u32x4*)((const vq_f16*)base + elem);
  }
  __device__ static __forceinline__ void unpack8(const Raw& r, float (&v)[8]) {
    unpack_h2(r.q[0].x, v[0], v[1]); unpack_h2(r.q[0].y, v[2], v[3]); unpack_h2(r.q[0].z, v[4], v[5]); unpack_h2(r.q[0].w, v[6], v[7]);
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    vq_u4 q;
    q.x = pack_h2(v[0], v[1]); q.y = pack_h2(v[2], v[3]); q.z = pack_h2(v[4], v[5]); q.w = pack_h2(v[6], v[7]);
    *(vq_u4*)((vq_f16*)base + elem) = q;
  }
  __device__ static __forceinline__ unsigned pack2(float a, float b) { return pack_h2(a, b); }
  __device__ static __forceinline__ void store8_nt(void* base, int64_t elem, const float (&v)[8]) {
    vq_u32x4 q;
    q.x = pack_h2(v[0], v[1]); q.y = pack_h2(v[2], v[3]); q.z = pack_h2(v[4], v[5]); q.w = pack_h2(v[6], v[7]);
    vq_store16_nt((vq_f16*)base + elem, q);
  }
  __device__ static __forceinline__ void load8_raw_nt(Raw& r, const void* base, int64_t elem) {
    r.q[0] = vq_load16_nt((const vq_f16*)base + elem);
  }
  __device__ static __forceinline__ void load4(const void* base, int64_t elem, float (&v)[4]) {
    vq_u2 q = *(const vq_u2*)((const vq_f16*)base + elem);
    unpack_h2(q.x, v[0], v[1]); unpack_h2(q.y, v[2], v[3]);
  }
  __device__ static __forceinline__ void store4(void* base, int64_t elem, const float (&v)[4]) {
    vq_u2 q;
    q.x = pack_h2(v[0], v[1]); q.y = pack_h2(v[2], v[3]);
    *(vq_u2*)((vq_f16*)base + elem) = q;
  }
  __device__ static __forceinline__ float load1(const void* base, int64_t elem) { return h2f(((const vq_f16*)base)[elem]); }
  __device__ static __forceinline__ void store1(void* base, int64_t elem, float v) { ((vq_f16*)base)[elem] = f2h(v); }
};
template <> struct Store<VQ_F32> {
  typedef float T;
  static constexpr int BYTES = 4;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    const vq_f4* p = (const vq_f4*)((const float*)base + elem);
    vq_f4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static constexpr int RAWQ = 2;
  struct Raw { vq_u32x4 q[2]; };
  __device__ static __forceinline__ void load8_issue(Raw& r, const void* base, int64_t elem) {
    vq_gload16_issue(r.q[0], (const float*)base + elem);
    vq_gload16_issue(r.q[1], (const float*)base + elem + 4);
  }
  __device__ static __forceinline__ void load8_raw(Raw& r, const void* base, int64_t elem) {
    r.q[0] = *(const vq_u32x4*)((const float*)base + elem);
    r.q[1] = *(const vq_u32x4*)((const float*)base + elem + 4);
  }
  __device__ static __forceinline__ void unpack8(const Raw& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.q[0].x); v[1] = __uint_as_float(r.q[0].y); v[2] = __uint_as_float(r.q[0].z); v[3] = __uint_as_float(r.q[0].w);
    v[4] = __uint_as_float(r.q[1].x); v[5] = __uint_as_float(r.q[1].y); v[6] = __uint_as_float(r.q[1].z); v[7] = __uint_as_float(r.q[1].w);
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    vq_f4* p = (vq_f4*)((float*)base + elem);
    vq_f4 a, b;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3]; b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
    p[0] = a; p[1] = b;
  }
  __device__ static __forceinline__ unsigned pack2(float a, float) { return __float_as_uint(a); }   // (never used: 16-bit epilogue only)
  __device__ static __forceinline__ void store8_nt(void* base, int64_t elem, const float (&v)[8]) {
    vq_u32x4 a, b;
    a.x = __float_as_uint(v[0]); a.y = __float_as_uint(v[1]); a.z = __float_as_uint(v[2]); a.w = __float_as_uint(v[3]);
    b.x = __float_as_uint(v[4]); b.y = __float_as_uint(v[5]); b.z = __float_as_uint(v[6]); b.w = __float_as_uint(v[7]);
    vq_store16_nt((float*)base + elem, a);
    vq_store16_nt((float*)base + elem + 4, b);
  }
  __device__ static __forceinline__ void load8_raw_nt(Raw& r, const void* base, int64_t elem) {
    r.q[0] = vq_load16_nt((const float*)base + elem);
    r.q[1] = vq_load16_nt((const float*)base + elem + 4);
  }
  __device__ static __forceinline__ void load4(const void* base, int64_t elem, float (&v)[4]) {
    vq_f4 a = *(const vq_f4*)((const float*)base + elem);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  __device__ static __forceinline__ void store4(void* base, int64_t elem, const float (&v)[4]) {
    vq_f4 a; a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3];
    *(vq_f4*)((float*)base + elem) = a;
  }
  __device__ static __forceinline__ float load1(const void* base, int64_t elem) {
    return ((const float*)base)[elem];
  }
  __device__ static __forceinline__ void store1(void* base, int64_t elem, float v) {
    ((float*)base)[elem] = v;
  }
};

// VQ_F16X2 (include/vqhip.h): v = hi + lo, two binary16 numbers; per group of 8 channels 16 bytes of hi, then 16 bytes of lo — the
// 32-byte footprint of fp32 storage, and byte for byte a binary16 tensor of 2C virtual channels for the LDS-DMA conv kernels.
// Element offsets are those of a 4-byte type; every vector access is one whole group (elem % 8 == 0).
__device__ __forceinline__ void vq_x2_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_h2(a, b);
  float fa, fb;
  unpack_h2(hi, fa, fb);
  lo = pack_h2(a - fa, b - fb);                      // the residual is exact in fp32; rounded to binary16 (11 more bits)
}
template <> struct Store<VQ_F16X2> {
  typedef unsigned T;
  static constexpr int BYTES = 4;
  __device__ static __forceinline__ const char* at(const void* base, int64_t elem) { return (const char*)base + (elem << 2); }
  __device__ static __forceinline__ void join(const vq_u32x4& h, const vq_u32x4& l, float (&v)[8]) {
    float a[8], b[8];
    unpack_h2(h.x, a[0], a[1]); unpack_h2(h.y, a[2], a[3]); unpack_h2(h.z, a[4], a[5]); unpack_h2(h.w, a[6], a[7]);
    unpack_h2(l.x, b[0], b[1]); unpack_h2(l.y, b[2], b[3]); unpack_h2(l.z, b[4], b[5]); unpack_h2(l.w, b[6], b[7]);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[e] + b[e];
  }
  __device__ static __forceinline__ void split(const float (&v)[8], vq_u32x4& h, vq_u32x4& l) {
    unsigned hx, hy, hz, hw, lx, ly, lz, lw;
    vq_x2_split2(v[0], v[1], hx, lx); vq_x2_split2(v[2], v[3], hy, ly);
    vq_x2_split2(v[4], v[5], hz, lz); vq_x2_split2(v[6], v[7], hw, lw);
    h.x = hx; h.y = hy; h.z = hz; h.w = hw; l.x = lx; l.y = ly; l.z = lz; l.w = lw;
  }
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    const vq_u32x4* p = (const vq_u32x4*)at(base, elem);
    const vq_u32x4 h = p[0], l = p[1];
    join(h, l, v);
  }
  static constexpr int RAWQ = 2;
  struct Raw { vq_u32x4 q[2]; };
  __device__ static __forceinline__ void load8_issue(Raw& r, const void* base, int64_t elem) {
    vq_gload16_issue(r.q[0], at(base, elem));
    vq_gload16_issue(r.q[1], at(base, elem) + 16);
  }
  __device__ static __forceinline__ void load8_raw(Raw& r, const void* base, int64_t elem) {
    const vq_u32x4* p = (const vq_u32x4*)at(base, elem);
    r.q[0] = p[0]; r.q[1] = p[1];
  }
  __device__ static __forceinline__ void unpack8(const Raw& r, float (&v)[8]) { join(r.q[0], r.q[1], v); }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    vq_u32x4 h, l;
    split(v, h, l);
    vq_u32x4* p = (vq_u32x4*)((char*)base + (elem << 2));
    p[0] = h; p[1] = l;
  }
  __device__ static __forceinline__ unsigned pack2(float a, float) { return __float_as_uint(a); }   // (never used: 16-bit epilogue only)
  __device__ static __forceinline__ void store8_nt(void* base, int64_t elem, const float (&v)[8]) {
    vq_u32x4 h, l;
    split(v, h, l);
    char* p = (char*)base + (elem << 2);
    vq_store16_nt(p, h);
    vq_store16_nt(p + 16, l);
  }
  __device__ static __forceinline__ void load8_raw_nt(Raw& r, const void* base, int64_t elem) {
    r.q[0] = vq_load16_nt(at(base, elem));
    r.q[1] = vq_load16_nt(at(base, elem) + 16);
  }
  // 4 elements (elem % 4 == 0): half a group — 8 bytes of hi at (elem & 4) * 2 inside the group, 8 bytes of lo 16 bytes on
  __device__ static __forceinline__ void load4(const void* base, int64_t elem, float (&v)[4]) {
    const char* p = (const char*)base + ((elem >> 3) << 5) + ((elem & 4) << 1);
    const vq_u2 h = *(const vq_u2*)p, l = *(const vq_u2*)(p + 16);
    float a[4], b[4];
    unpack_h2(h.x, a[0], a[1]); unpack_h2(h.y, a[2], a[3]); unpack_h2(l.x, b[0], b[1]); unpack_h2(l.y, b[2], b[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e] + b[e];
  }
  __device__ static __forceinline__ void store4(void* base, int64_t elem, const float (&v)[4]) {
    char* p = (char*)base + ((elem >> 3) << 5) + ((elem & 4) << 1);
    vq_u2 h, l;
    vq_x2_split2(v[0], v[1], h.x, l.x); vq_x2_split2(v[2], v[3], h.y, l.y);
    *(vq_u2*)p = h; *(vq_u2*)(p + 16) = l;
  }
  __device__ static __forceinline__ float load1(const void* base, int64_t elem) {
    const char* p = (const char*)base + ((elem >> 3) << 5) + ((elem & 7) << 1);
    return h2f(*(const vq_f16*)p) + h2f(*(const vq_f16*)(p + 16));
  }
  __device__ static __forceinline__ void store1(void* base, int64_t elem, float v) {
    char* p = (char*)base + ((elem >> 3) << 5) + ((elem & 7) << 1);
    const vq_f16 h = f2h(v);
    *(vq_f16*)p = h; *(vq_f16*)(p + 16) = f2h(v - h2f(h));
  }
};
// storage types whose stored values live in binary16's range (loss-scaled gradients, range events)
template <int DT> struct IsHalfRange { static constexpr bool value = DT == VQ_F16 || DT == VQ_F16X2; };

// vq_raw_wait(r): all asynchronous raw loads issued so far have landed; r[0..U) become usable.  The registers are
// in/out operands of the wait, so no use of them can be scheduled above it.
template <typename R, int U> __device__ __forceinline__ void vq_raw_wait(R (&r)[U]) {
#ifndef VQ_EMU
  static_assert(U == 1 || U == 2 || U == 4, "batch sizes in use");
  constexpr int Q = (int)(sizeof(R) / sizeof(vq_u32x4));
  static_assert(Q == 1 || Q == 2, "one or two 16-byte pieces per raw item");
  if constexpr (Q == 1) {
    if constexpr (U == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]), "+v"(r[1].q[0]), "+v"(r[2].q[0]), "+v"(r[3].q[0]));
    else if constexpr (U == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]), "+v"(r[1].q[0]));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]));
  } else {
    if constexpr (U == 4)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]), "+v"(r[0].q[1]), "+v"(r[1].q[0]), "+v"(r[1].q[1]), "+v"(r[2].q[0]),
                   "+v"(r[2].q[1]), "+v"(r[3].q[0]), "+v"(r[3].q[1]));
    else if constexpr (U == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]), "+v"(r[0].q[1]), "+v"(r[1].q[0]), "+v"(r[1].q[1]));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0].q[0]), "+v"(r[0].q[1]));
  }
#else
  (void)r;
#endif
}

// ------------------------------------------------------------------ wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// v_permlane32_swap_b32: the upper 32 lanes of `a` change places with the lower 32 lanes of `b`.  Afterwards a lane of the lower half
// holds (its own a, the a of lane + 32), a lane of the upper half (the b of lane - 32, its own b).  Pinned to silicon by
// tests/test_hw_layout.py (probe 4).
__device__ __forceinline__ void vq_swap32(unsigned& a, unsigned& b) {
#ifdef VQ_EMU
  const unsigned ta = __shfl_xor(a, 32), tb = __shfl_xor(b, 32);
  const bool hi = (threadIdx.x & 63) >= 32;
  const unsigned na = hi ? tb : a, nb = hi ? b : ta;
  a = na; b = nb;
#else
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}

// ------------------------------------------------------------------ VQ_F16 range events (include/vqhip.h, "range events")
// Every kernel that writes a binary16 tensor of a loss-scaled stack can report what the saturating store did to it:
//   ev[0] += 1  per wave that stored at least one value beyond +-65504 (or an inf / NaN): the tensor is clipped;
//   ev[1] += 1  per wave whose stored values were ALL flushed to zero although some were non-zero in fp32: a region of the tensor
//               vanished (isolated small elements flushing next to live ones is ordinary rounding and is not counted).
//   ev[2] += 1  per wave that stored a magnitude of 2^13 or more (round 6, "headroom" events): nothing is lost yet — three bits below
//               the limit — but a loss scale calibrated to put a stack's largest gradient at 2^10 has been outgrown 8x.  The trainer
//               lowers the scale at its next poll, BEFORE a store clips and an optimizer step has to be dropped.
// A healthy step executes no atomic at all.  The running maxima are kept on the BIT PATTERN of |v| (one v_and + v_max_u32 per value,
// or a v_max3): inf and NaN order above every finite value, so neither can hide behind fmaxf's NaN-dropping rule.
__device__ __forceinline__ unsigned vq_absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned vq_umax(unsigned a, unsigned b) { return a > b ? a : b; }
template <int N> __device__ __forceinline__ unsigned vq_absmax_bits(unsigned m, const float (&v)[N]) {
#pragma unroll
  for (int e = 0; e < N; ++e) m = vq_umax(m, vq_absbits(v[e]));
  return m;
}
// true in every lane iff `pred` holds in some lane of the wave (all lanes must call it)
__device__ __forceinline__ bool vq_wave_any(bool pred) {
#ifdef VQ_EMU
  int v = pred ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v != 0;
#else
  return __ballot(pred) != 0ull;
#endif
}
// m_sat: max |v| bits over everything this lane stored (any rounding point); m_final: the same over its FINAL stored values only
__device__ __forceinline__ void vq_range_events(int* ev, unsigned m_sat, unsigned m_final) {
  constexpr unsigned F16_MAX = 0x477fe000u;        // 65504.0f
  constexpr unsigned F16_TINY = 0x33800000u;       // 2^-24: the smallest binary16 subnormal; anything below half of it stores as 0
  constexpr unsigned F16_HOT = 0x46000000u;        // 8192.0f = 2^13
  const bool sat = m_sat > F16_MAX, live = m_final >= F16_TINY, flushed = m_final != 0u && !live;
  const bool any_sat = vq_wave_any(sat), any_live = vq_wave_any(live), any_flushed = vq_wave_any(flushed);
  const bool any_hot = vq_wave_any(m_sat >= F16_HOT);
  if ((threadIdx.x & 63) == 0) {
    if (any_sat) atomicAdd(ev, 1);
    if (any_flushed && !any_live) atomicAdd(ev + 1, 1);
    if (any_hot) atomicAdd(ev + 2, 1);
  }
}

// The same test on PACKED binary16 results: pk = running per-half maximum of (stored pair & 0x7fff7fff) over this lane's final stores
// (vq_pkmax16), orbits = OR of the raw fp32 bits of (a sample of) the values that went in.  Saturated: a stored magnitude of 65504
// (0x7bff: vq_sat16 clamps there) or an inf / NaN pattern; flushed: something non-zero went in and every stored half is zero.
__device__ __forceinline__ unsigned vq_pkmax16(unsigned m, unsigned a, unsigned b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 x, y, z;
  __builtin_memcpy(&x, &m, 4); __builtin_memcpy(&y, &a, 4); __builtin_memcpy(&z, &b, 4);
  x = __builtin_elementwise_max(__builtin_elementwise_max(x, y), z);
  unsigned r; __builtin_memcpy(&r, &x, 4); return r;
}
__device__ __forceinline__ void vq_range_events16(int* ev, unsigned pk, unsigned orbits) {
  const unsigned m16 = vq_umax(pk & 0xffffu, pk >> 16);
  const bool sat = m16 >= 0x7bffu, live = m16 != 0u, flushed = !live && (orbits & 0x7fffffffu) != 0u;
  const bool any_sat = vq_wave_any(sat), any_live = vq_wave_any(live), any_flushed = vq_wave_any(flushed);
  const bool any_hot = vq_wave_any(m16 >= 0x7000u);  // binary16 2^13
  if ((threadIdx.x & 63) == 0) {
    if (any_sat) atomicAdd(ev, 1);
    if (any_flushed && !any_live) atomicAdd(ev + 1, 1);
    if (any_hot) atomicAdd(ev + 2, 1);
  }
}

// sum over the `width` (power of two, <= 64) lanes of an aligned sub-group
template <int WIDTH>
__device__ __forceinline__ float subgroup_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sigmoid on the hardware transcendental units: v_exp_f32 (2^x, ~1 ulp) + v_rcp_f32 (~1 ulp); the
// GroupNorm+swish kernels are HBM-bound only if these stay at two quarter-rate instructions per element
#ifdef VQ_EMU
__device__ __forceinline__ float vq_sigmoid(float y) { return 1.0f / (1.0f + expf(-y)); }
#else
__device__ __forceinline__ float vq_sigmoid(float y) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * y));
}
#endif

// ------------------------------------------------------------------ MFMA wrappers
#ifdef VQ_EMU
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) { return emu_mfma_32x32x16_bf16(a, b, c); }
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) { return emu_mfma_16x16x32_bf16(a, b, c); }
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const short* p) { return emu_ds_read_tr16_b64(p); }
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) { return emu_mfma_32x32x16_f16(a, b, c); }
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) { return emu_mfma_32x32x2_f32(a, b, c); }
#else
// f32 in / f32 accumulate at the fp32 VECTOR rate (64 cycles per instruction per SIMD), exact f32: an fmaf chain over k (the nearest-code
// search of optim_vq.hip is bit-exact against oracle/vq_oracle.c through it).  Lane l: A[l & 31][l >> 5], B[l >> 5][l & 31].
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vq_f16x8, a), __builtin_bit_cast(vq_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vq_bf16x8, a), __builtin_bit_cast(vq_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vq_bf16x8, a), __builtin_bit_cast(vq_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
#endif

// The 16-bit operand type of a kernel: VQ_BF16 (8-bit mantissa) or VQ_F16 (10-bit mantissa = TF32's; scaled operands).  Same
// instruction shape, rate and register layouts (guide §3: C/D layout is dtype-independent; tests/test_hw_layout.py probes both).
template <int DT> __device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c) {
  static_assert(DT == VQ_BF16 || DT == VQ_F16 || DT == VQ_F16X2, "16-bit MFMA operand types");
  if constexpr (DT == VQ_F16 || DT == VQ_F16X2) return mfma_32x32x16_f16(a, b, c);
  else return mfma_32x32x16_bf16(a, b, c);
}
// bit pattern of 1.0 in the operand type (the "times ones" bias-gradient MFMAs)
template <int DT> struct One16 { static constexpr unsigned short BITS = DT == VQ_F16 ? 0x3C00 : 0x3F80; };
// fp32 -> the 16-bit operand type (register-staged loaders of fp32-free kernels never need it; weight packing does)
template <int DT> __device__ __forceinline__ unsigned short f2op(float f) {
  if constexpr (DT == VQ_F16) return f2h(f);
  else return f2bf(f);
}

// Direct global -> LDS copy (LDS-DMA): every lane fetches 16 bytes from its own global address; the
// wave's 1 KiB lands at `lds_wave_base + lane*16` (wave-uniform base, lane-linear image).  Completion is
// tracked by vmcnt; __syncthreads() after it drains the DMA (guide §5 "Async global->LDS").
#ifdef VQ_EMU
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  memcpy((char*)lds_wave_base + (threadIdx.x & 63) * 16, gsrc, 16);
}
#else
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#endif

// The same copy issued from inline asm: hipcc does not see it, so it neither counts it in its s_waitcnt bookkeeping nor drains
// the queue because of it (guide §5.7: an asm LDS-DMA has no register destination — register-safe; its completion is the
// caller's: a counted vmcnt wait, then a barrier).  M0 (the LDS base of the transfer) is saved and restored inside the statement.
#ifdef VQ_EMU
__device__ __forceinline__ void glds16_asm(const void* gsrc, void* lds_wave_base) { glds16(gsrc, lds_wave_base); }
#else
__device__ __forceinline__ void glds16_asm(const void* gsrc, void* lds_wave_base) {
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
#endif

// scheduling fence: no instruction is moved across it by the compiler's scheduler
// vq_wave_sync(): ordering point between LDS accesses of different lanes of ONE wave (wave-private LDS slabs).  The hardware
// runs a wave's LDS instructions in order, so only the compiler must not reorder; the fiber emulator needs a real rendezvous.
#ifdef VQ_EMU
#define vq_sched_fence() ((void)0)
#define vq_setprio(x) ((void)0)
#define vq_wave_sync() emu::wave_barrier()
#else
#define vq_sched_fence() __builtin_amdgcn_sched_barrier(0)
#define vq_setprio(x) __builtin_amdgcn_s_setprio(x)
#define vq_wave_sync() __builtin_amdgcn_wave_barrier()
#endif

// ---- explicit pipeline control (LDS-DMA kernels) -------------------------------------------------------------
// counted waits on the gfx9 s_waitcnt immediate: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#ifndef VQ_EMU
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field on gfx9");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
#endif
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
#ifndef VQ_EMU
  static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit field on gfx9");
  __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (N << 8) | (3 << 14));
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
// Transposed LDS read the compiler does not track (asm volatile): hipcc puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 that follows an LDS-DMA issue (it cannot tell the buffers apart), which serialises
// the DMA of the next chunk with the fragment reads of the current one.  The caller owns the completion wait:
// wait_lgkmcnt<N>() followed by vq_tie(regs...) before the first use of the registers (LDS reads return in order).
template <int OFF> __device__ __forceinline__ s16x4 lds_read_tr16_b64_async(const char* p) {
#ifdef VQ_EMU
  return emu_ds_read_tr16_b64((const short*)(p + OFF));
#else
  s16x4 r;
  const unsigned a = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
#endif
}
// makes every later use of the registers depend on this point of the asm-volatile order (i.e. on the preceding wait)
template <typename T> __device__ __forceinline__ void vq_tie1(T& r) {
#ifndef VQ_EMU
  asm volatile("" : "+v"(r));
#endif
}
template <typename... T> __device__ __forceinline__ void vq_tie(T&... regs) { (vq_tie1(regs), ...); }

// All LDS of a kernel in ONE dynamically sized array (a second __shared__ object makes hipcc drain
// vmcnt(0) before every ds_read of an LDS-DMA pipeline — guide §5, trap (a)).
#ifdef VQ_EMU
#define VQ_DYN_LDS(T, name) static thread_local __attribute__((aligned(16))) T name[163840 / sizeof(T)]
#else
#define VQ_DYN_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

// conv_small.hip: dedicated kernels for 8 (padded) input channels; launch_conv_c8 returns 1 if the shape is not its own
int vq_launch_conv_c8(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual,
                      const void* relu_mask, void* y, float alpha, const float* alpha_dev, hipStream_t stream);
// the same layers in VQ_F16X2 storage (d = the virtualised descriptor of vq_conv2d_fwd: Cin == 16 virtual channels)
bool vq_conv_c8_x2_shape(const VqConvDesc* d);
int vq_launch_conv_c8_x2(const VqConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual,
                         const void* relu_mask, void* y, float alpha, const float* alpha_dev, hipStream_t stream);
bool vq_wgrad_c8_eligible(const VqConvDesc* d);
size_t vq_wgrad_c8_workspace(const VqConvDesc* d);
int vq_launch_wgrad_c8(const VqConvDesc* d, const void* x, const void* dy, float* dw, float* dbias, int* dbias_done, int accumulate, float alpha,
                       void* workspace, hipStream_t stream);
// ... and for VQ_F16X2 storage (d = the caller's descriptor, real channel counts): two passes of the same kernel + a four-product reduction
bool vq_wgrad_c8_x2_eligible(const VqConvDesc* d);
size_t vq_wgrad_c8_x2_workspace(const VqConvDesc* d);
int vq_launch_wgrad_c8_x2(const VqConvDesc* d, const void* x, const void* dy, float* dw, float* dbias, int* dbias_done, int accumulate, float alpha,
                          void* workspace, hipStream_t stream);

// compile-time unrolled loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
#include <utility>
template <typename F, int... I> __device__ __forceinline__ void vq_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void vq_static_for(F&& f) {
  vq_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

static inline int64_t vq_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int vq_round_up(int a, int b) { return (a + b - 1) / b * b; }
