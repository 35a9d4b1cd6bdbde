// Shared declarations for the gfx950 (MI355X / CDNA4) kernels of the TubeDETR hot path.
// wave = 64 lanes everywhere; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tubedetr_hip.h"

namespace td {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// thread-local last-error string (td_last_error)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define TD_REQUIRE(cond, ...)           \
  do {                                  \
    if (!(cond)) {                      \
      td::set_error(__VA_ARGS__);       \
      return TD_ERR_INVALID;            \
    }                                   \
  } while (0)

// optional device-side dropout step counter (the `dropout_counter` argument of the dropout-capable entry points): the
// kernels re-key their seed with a HASH of (seed, *ptr), so a captured HIP graph draws fresh, independent masks on
// every replay without re-recording launches.  (A linear re-keying - seed + ctr * golden - would make the mask of step
// c the step-0 mask shifted by c elements, because hash32() below also enters the element index through
// idx * golden + seed.)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t effective_seed(uint32_t seed, const uint32_t* ctr) {
  return ctr ? mix32(mix32(seed ^ 0xA511E9B3u) + ctr[0] * 0xC2B2AE3Du) : seed;
}

// Deterministic parity mode (td_set_deterministic; seeded ONCE from TD_DETERMINISTIC=1 in the environment): reductions that are normally
// split over workgroups and combined with fp32 atomics - weight gradients over M, LayerNorm's dgamma / dbeta, bias column sums - run as
// ONE sequential reduction per output element, so that two runs of the same step are bit-identical (the exact-fp32 parity mode's
// regression anchor; slow).  One atomic flag in api.cpp: no getenv() on the launch path (another thread may be changing the environment).
bool deterministic();

// bench-only launch timing (api.cpp)
bool prof_on();
void prof_begin(int family, int dtype, double flops, hipStream_t st, int M = 0, int N = 0, int K = 0, int R = 0, int stride = 0, int mode = 0);
void prof_end(hipStream_t st);
void prof_set_bytes(double algorithmic_bytes);  // of the launch recorded by the last prof_begin

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (round-to-nearest-even), bit exact with torch's float->bfloat16 ----
__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kDtype = TD_F32;
  __device__ static __forceinline__ float load(const void* p, size_t i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void store(void* p, size_t i, float v) { ((float*)p)[i] = v; }
};
template <>
struct Elem<u16> {
  static constexpr int kDtype = TD_BF16;
  __device__ static __forceinline__ float load(const void* p, size_t i) { return bf16_to_f32(((const u16*)p)[i]); }
  __device__ static __forceinline__ void store(void* p, size_t i, float v) { ((u16*)p)[i] = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-based dropout RNG: keep-probability test on a 32-bit hash of (seed, element index).
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t idx) {
  return mix32(idx * 0x9E3779B1u + seed);
}
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t idx, uint32_t thresh) { return hash32(seed, idx) >= thresh; }

}  // namespace td
