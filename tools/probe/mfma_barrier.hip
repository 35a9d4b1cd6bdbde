// What a workgroup barrier per MFMA block costs on gfx950, and how much of the LDS fragment reads hides behind the MFMAs of the
// other wavefronts of a SIMD.  One workgroup per CU (256 workgroups), NW wavefronts; per "stage" a wavefront issues PER MFMAs
// (16x16x32 bf16, 16 independent accumulators, operands resident in registers) and optionally RD transposing LDS reads
// (ds_read_b64_tr_b16, results discarded) in front of them, then optionally meets the others at s_barrier.
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_barrier.hip -o tools/probe/mfma_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int NW, int PER, int RD, bool BAR>
__global__ __launch_bounds__(NW * 64, 1) void k(const uint4* __restrict__ in, float* out, int stages) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 16; i += NW * 64) ((uint4*)lds)[i] = in[i & 4095];
  __syncthreads();
  u4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { uint4 t = in[(threadIdx.x * 4 + i) & 4095]; a[i] = u4{t.x, t.y, t.z, t.w}; t = in[(threadIdx.x * 4 + i + 1111) & 4095]; b[i] = u4{t.x, t.y, t.z, t.w}; }
  f4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)((8 * lg + (lr >> 2)) * 256 + (((lr >> 2) | ((lg & 1) << 2)) << 5) + (lr & 3) * 8) + (threadIdx.x >> 6) * 1024u % 32768u;
  uint2 sink[RD > 0 ? RD : 1];
  for (int s = 0; s < stages; ++s) {
#pragma unroll
    for (int r = 0; r < RD; ++r) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(sink[r]) : "v"(addr), "n"((r & 7) * 2048 + (r >> 3) * 32));
    if (RD > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < PER; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a[m & 3]), "v"(b[(m >> 2) & 3]));
#pragma unroll
    for (int r = 0; r < RD; ++r) asm volatile("" ::"v"(sink[r]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s_ = 0.f;
  for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(acc[i])); s_ += acc[i][0] + acc[i][3]; }
  out[blockIdx.x * NW * 64 + threadIdx.x] = s_;
}
template <int NW, int PER, int RD, bool BAR>
static void run(const uint4* in, float* out, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int total_mfma_per_wave = 1 << 20, stages = total_mfma_per_wave / PER;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<NW, PER, RD, BAR><<<256, NW * 64>>>(in, out, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = 256.0 * NW * total_mfma_per_wave * 16384.0;
  printf("%2d waves, %2d MFMAs per stage, %2d reads, barrier %d : %8.2f ms  %7.1f TFLOP/s   %s\n", NW, PER, RD, (int)BAR, best, fl / best / 1e9, what);
}

// Software-pipelined stage of ONE wavefront per SIMD (4 per workgroup, 128 x 128 wave tiles: 64 MFMAs per 32-row k-step): the RD
// transposing reads of the NEXT stage are issued between the MFMAs of this one (one read per 64 / RD MFMAs), DMA optional
// (ND buffer_load ... lds of 1 KiB per wavefront and stage from an L2-resident buffer), one barrier per stage.
typedef __attribute__((address_space(3))) void* lds_p;
template <int NW, int PER, int RD, int ND>
__global__ __launch_bounds__(NW * 64, 1) void kp(const uint4* __restrict__ in, float* out, int stages) {
  __shared__ __attribute__((aligned(16))) char lds[131072];
  for (int i = threadIdx.x; i < 131072 / 16; i += NW * 64) ((uint4*)lds)[i] = in[i & 4095];
  __syncthreads();
  u4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { uint4 t = in[(threadIdx.x * 4 + i) & 4095]; a[i] = u4{t.x, t.y, t.z, t.w}; t = in[(threadIdx.x * 4 + i + 1111) & 4095]; b[i] = u4{t.x, t.y, t.z, t.w}; }
  f4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_p)lds;
  uint32_t addr = base + (uint32_t)((8 * lg + (lr >> 2)) * 256 + (((lr >> 2) | ((lg & 1) << 2)) << 5) + (lr & 3) * 8);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 65536, 0x00020000);
  uint2 sink[RD];
  int slot = 0;
  for (int s = 0; s < stages; ++s) {
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int d = 0; d < ND; ++d) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)(lds + slot * 32768 + (wave * ND + d) * 1024), 16, (uint32_t)(lane * 16 + d * 1024), 0, 0, 0);
    const uint32_t rd_base = addr + (uint32_t)(((slot + 1) & 3) * 32768);
#pragma unroll
    for (int m = 0; m < PER; ++m) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 15]) : "v"(a[m & 3]), "v"(b[(m >> 2) & 3]));
      if (RD > 0 && (m % (PER / RD)) == 0) {
        constexpr int dummy = 0; (void)dummy;
        const int r = m / (PER / RD);
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(sink[r]) : "v"(rd_base), "n"(0));
      }
    }
    if (ND > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ND) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < RD; ++r) asm volatile("" ::"v"(sink[r]));
    slot = (slot + 1) & 3;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s_ = 0.f;
  for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(acc[i])); s_ += acc[i][0] + acc[i][3]; }
  out[blockIdx.x * NW * 64 + threadIdx.x] = s_;
}
template <int NW, int PER, int RD, int ND>
static void runp(const uint4* in, float* out, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int total_mfma_per_wave = 1 << 20, stages = total_mfma_per_wave / PER;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    kp<NW, PER, RD, ND><<<256, NW * 64>>>(in, out, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = 256.0 * NW * total_mfma_per_wave * 16384.0;
  printf("%2d waves, %2d MFMAs per stage, %2d reads interleaved, %d DMA KiB per wave, barrier : %8.2f ms  %7.1f TFLOP/s   %s\n", NW, PER, RD, ND, best, fl / best / 1e9, what);
}
int main() {
  uint4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 16 * 64 * 4);
  uint32_t* h = (uint32_t*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 4; ++i) { float x = (rand() / (float)RAND_MAX) * 2.f - 1.f, y = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t xu, yu; memcpy(&xu, &x, 4); memcpy(&yu, &y, 4); h[i] = (xu >> 16) | (yu & 0xffff0000u); }
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  run<16, 16, 0, false>(in, out, "MFMAs only");
  run<16, 16, 0, true>(in, out, "barrier per 16");
  run<16, 32, 0, true>(in, out, "barrier per 32");
  run<16, 64, 0, true>(in, out, "barrier per 64");
  run<16, 16, 16, false>(in, out, "16 reads + wait + 16 MFMAs, no barrier");
  run<16, 16, 16, true>(in, out, "the 16-wavefront wgrad stage");
  run<16, 32, 32, true>(in, out, "its 64-row stage");
  run<8, 32, 0, false>(in, out, "MFMAs only");
  run<8, 16, 0, true>(in, out, "barrier per 16 (the phased 256-row kernel has two per 16)");
  run<8, 32, 0, true>(in, out, "barrier per 32");
  run<8, 64, 0, true>(in, out, "barrier per 64");
  run<8, 32, 24, false>(in, out, "24 reads + wait + 32 MFMAs, no barrier");
  run<8, 32, 24, true>(in, out, "same with a barrier");
  run<8, 64, 48, true>(in, out, "twice the stage");
  run<4, 64, 0, false>(in, out, "MFMAs only, one wavefront per SIMD");
  run<4, 64, 0, true>(in, out, "barrier per 64");
  runp<4, 64, 32, 0>(in, out, "128 x 128 wave tiles, reads pipelined, no DMA");
  runp<4, 64, 32, 8>(in, out, "same + the stage's DMA");
  runp<8, 32, 16, 4>(in, out, "eight wavefronts, 32 MFMAs, 16 reads (a 128 x 64 tile needs 24)");
  runp<8, 32, 32, 4>(in, out, "eight wavefronts, 32 MFMAs, 32 reads");
  runp<16, 16, 16, 2>(in, out, "sixteen wavefronts, 16 MFMAs, 16 reads, pipelined");
  return 0;
}
