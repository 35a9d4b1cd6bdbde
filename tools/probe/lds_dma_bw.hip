// How fast can buffer_load ... lds (LDS-DMA) stream into the CUs?  One workgroup of 8 wavefronts per CU; every wavefront keeps
// DEPTH 1-KiB DMA instructions in flight (counted vmcnt) into a 64-KiB LDS ring and walks a window of `span` bytes that all
// workgroups of an XCD share (span 2 MiB: L2 hits; 128 MiB: Infinity Cache; 8 GiB: HBM).  No MFMA, no LDS reads.
// hipcc --offload-arch=gfx950 -O3 tools/probe/lds_dma_bw.hip -o tools/probe/lds_dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_p;
template <int DEPTH>
__global__ __launch_bounds__(512, 1) void k(const char* __restrict__ src, uint32_t span_mask, int iters, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xFFFFFFFFu, 0x00020000);
  // workgroup b of XCD x = b % 8 starts at a different 8-KiB block; the eight wavefronts of a workgroup read consecutive KiB
  uint32_t off = (uint32_t)((blockIdx.x >> 3) * 8192 + wave * 1024 + lane * 16);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)(lds + ((wave * DEPTH + d) & 63) * 1024), 16, off & span_mask, 0, 0, 0);
      off += 32u * 8192u;  // the next block of this workgroup (32 workgroups per XCD)
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)lds[blockIdx.x & 65535];
}
int main() {
  const size_t total = 4ull << 30;  // the 32-bit buffer range
  char* src; float* out;
  hipMalloc(&src, total); hipMalloc(&out, 256 * 4);
  hipMemset(src, 1, total);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (uint32_t span : {1u << 21, 1u << 24, 1u << 27, 0u}) {
    const uint32_t mask = span ? span - 1 : 0xFFFFFFFFu;
    for (int depth : {4, 8}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (depth == 4) k<4><<<256, 512>>>(src, mask, iters, out); else k<8><<<256, 512>>>(src, mask, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = 256.0 * 8 * iters * depth * 1024.0;
      printf("window %8.0f MiB, %d KiB in flight per wavefront: %7.2f ms  %6.2f TB/s into LDS\n", span ? span / 1048576.0 : 4096.0, depth, best, bytes / best / 1e9);
    }
  }
  return 0;
}
