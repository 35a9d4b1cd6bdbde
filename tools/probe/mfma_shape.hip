// Sustained MFMA throughput of the two bf16 shapes on random operands (power-limited clock included): which instruction gives more
// FLOP/s when the chip runs at its power cap?  hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_shape.hip -o tools/probe/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ in, float* out, int iters) {
  uint4 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = in[(threadIdx.x * 8 + i) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = in[(threadIdx.x * 4 + i + 1111) & 4095];
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf8*)&b[i & 3], *(bf8*)&a[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f16v acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf8*)&b[(i + r) & 3], *(bf8*)&a[i], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  uint4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 8 * 512 * 4);
  uint32_t* h = (uint32_t*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 4; ++i) { float x = (rand() / (float)RAND_MAX) * 2.f - 1.f, y = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t xu, yu; memcpy(&xu, &x, 4); memcpy(&yu, &y, 4); h[i] = (xu >> 16) | (yu & 0xffff0000u); }
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256;
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {16, 32, 16, 32}) {
      hipEventRecord(e0);
      if (shape == 16) k<16><<<grid, 512>>>(in, out, iters); else k<32><<<grid, 512>>>(in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per lane-iteration: SHAPE 16: 32 MFMAs x 16384 FLOP; SHAPE 32: 16 MFMAs x 32768 FLOP  (per wavefront)
      double fl = (double)grid * 8 * iters * 32.0 * 16384.0;
      printf("shape %dx%d: %.2f ms  %.1f TFLOP/s\n", shape, shape, ms, fl / ms / 1e9);
    }
  return 0;
}
