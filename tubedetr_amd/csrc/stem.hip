// ResNet stem in ONE pass: 7x7 stride-2 convolution (pixel-pair form, FrozenBN folded) + bias + ReLU + 3x3 stride-2 max-pool.
// Replaces conv1 / bn1 / relu / maxpool of the torchvision trunk reached through models/backbone.py:94-98 for the frozen
// stem (no gradient ever flows here, backbone.py:82-89), on every frame of the step (1 000 per 8-clip step at res 352).
//
// Why its own kernel: as an implicit GEMM the stem is M = N*176*176 rows x 64 channels x K = 224 - 242 000 workgroups of a
// 3.5-tile K loop each (prologue / epilogue bound: 0.15 of the MFMA roof and 0.14 of the HBM roof measured), its 64-channel
// 176 x 176 output (4 GB per 1 000 frames) is written, re-read 1.5x by the pooling kernel and never used again.  Here a
// workgroup owns a tile of POOLED pixels: it stages the input patch once in LDS (the 28 taps of an output pixel are
// re-read from there, not through 28 gathers), keeps the whole 64 x 224 weight matrix as MFMA fragments in registers for
// the launch (persistent workgroups), leaves the convolution outputs of its tile (+ the one-pixel pooling halo) in LDS as
// bf16 and pools from there: HBM sees the 4-channel frames once and the pooled tensor once.
//
// Pixel-pair form (td_stem_pair_weights): frames are stored with 4 channels per pixel, two horizontally adjacent pixels form
// one 16-byte element; out(cy, cx) = sum_{r<7, t<4} W[r][t][8] . X[2cy + r - 3][cx + t - 2][8].  One k-step of the MFMA
// (v_mfma_f32_16x16x32_bf16: 32 k = 4 elements) is exactly one filter row r; lane group lg supplies tap t = lg.
#include <stdlib.h>

#include <algorithm>

#include "td_common.h"

namespace td {

typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_v __attribute__((ext_vector_type(2)));
typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));
// two fp32 -> packed bf16 (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  bf16x2_v v = {(__bf16)lo, (__bf16)hi};
  return *(uint32_t*)&v;
}
// ReLU of two packed bf16: max(x, 0) on the 16-bit integers (v_pk_max_i16): negative values, -0.0 included, become +0.0 - one operation
// per pair behind the conversion (fmaxf(x, 0.f) in front of it is two per value).  As asm because the compiler splits the conversion
// when it sees the vector form; the conversion itself must stay the compiler's instruction (it is the first reader of the MFMA result,
// and the wait states that read needs are inserted by the compiler's hazard pass, which does not look into asm operands).
__device__ __forceinline__ uint32_t relu_pk_bf16(uint32_t v) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
  return r;
}
// element-wise maximum of two packed pairs of NON-NEGATIVE bf16 values: their bit patterns order like unsigned integers (v_pk_max_u16)
__device__ __forceinline__ uint32_t max_pk_nonneg_bf16(uint32_t a, uint32_t b) {
  const u16x2_v r = __builtin_elementwise_max(*(const u16x2_v*)&a, *(const u16x2_v*)&b);
  return *(const uint32_t*)&r;
}

// workgroup barrier that orders LDS traffic only: the patch prefetch (global loads into registers) stays in flight across it
// (__syncthreads() would drain vmcnt and stall every tile on the HBM latency the prefetch exists to hide)
#define TD_LDS_BARRIER()                                   \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
  } while (0)

struct StemParams {
  const char* x;      // [N][H][W/2] elements of 8 bf16
  const char* w;      // [64][7][4][8] bf16
  const float* bias;  // [64]
  char* y;            // pooled [N][PH][PW][64] bf16
  int N, H, WP;       // WP = W / 2 (pairs per row)
  int CH, CW;         // convolution output size
  int PH, PW;         // pooled size
  int tiles_y, tiles_x, n_tiles;
};

// pooled tile TY x TX -> convolution region (2TY+1) x (2TX+1) (one halo row / column towards -1) -> input patch
// (2(2TY+1)+5) rows x ((2TX+1)+3) pairs
template <int TY, int TX>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(StemParams p) {
  constexpr int CR = 2 * TY + 1, CC = 2 * TX + 1;      // convolution rows / columns of a tile
  constexpr int MROWS = CR * CC, MB = (MROWS + 15) / 16;  // GEMM rows of a tile, 16-row blocks
  constexpr int PR = 2 * CR + 5, PC = CC + 3;          // input patch: rows x pairs
  constexpr int CPITCH = 144;                           // bytes per convolution-output row in LDS (64 bf16 + pad: 16-byte aligned, row r at bank 4r)
  __shared__ __attribute__((aligned(16))) char patch[PR * PC * 16];
  __shared__ __attribute__((aligned(16))) char ctile[MB * 16 * CPITCH];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (uint32_t)((size_t)p.N * p.PH * p.PW * 128), 0x00020000);
  // the weight matrix as MFMA A-operand fragments, resident for the whole launch: wreg[r][i] = rows (channels) i*16 + lr,
  // k = r*32 + lg*8 .. +8  (= filter row r, tap lg)
  uint4 wreg[7][4];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) wreg[r][i] = *(const uint4*)(p.w + ((size_t)(i * 16 + lr) * 224 + r * 32 + lg * 8) * 2);
  float bias4[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[i][q] = p.bias[i * 16 + 4 * lg + q];

  // The input patch of tile i + 1 is requested (into registers) before the convolution of tile i and written to LDS behind its
  // pooling: the HBM latency of a patch is covered by a whole tile of work instead of stalling the workgroup at every tile head.
  constexpr int PLD = (PR * PC + 255) / 256;  // 16-byte patch elements per thread
  uint4 pre[PLD];
  auto fetch_patch = [&](int tile) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int trem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int iy0 = 2 * (2 * ty * TY - 1) - 3, ix0 = (2 * tx * TX - 1) - 2;  // first input row / pair of the patch
    const char* ximg = p.x + (size_t)img * p.H * p.WP * 16;
#pragma unroll
    for (int j = 0; j < PLD; ++j) {
      const int e = t + j * 256;
      const int r = e / PC, c = e - r * PC;
      const int iy = iy0 + r, ix = ix0 + c;
      pre[j] = make_uint4(0, 0, 0, 0);  // zeros outside the frame
      if (e < PR * PC && tile < p.n_tiles && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.WP)
        pre[j] = *(const uint4*)(ximg + ((size_t)iy * p.WP + ix) * 16);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < PLD; ++j) {
      const int e = t + j * 256;
      if (e < PR * PC) *(uint4*)(patch + e * 16) = pre[j];
    }
  };
  fetch_patch(blockIdx.x);
  store_patch();
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int trem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int py0 = ty * TY, px0 = tx * TX;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;    // first convolution row / column of the tile (may be -1: pooling pad)
    fetch_patch(tile + gridDim.x);                     // (1) next tile's patch: in flight during this tile's work
    TD_LDS_BARRIER();
    // (2) convolution of the tile's rows, 16 at a time per wavefront; result (+ bias, ReLU; 0 outside the image = pooling pad,
    //     valid because every window holds at least one real, non-negative value) -> LDS as bf16
    for (int mb = wave; mb < MB; mb += 4) {
      const int m = mb * 16 + lr;
      const int mm = m < MROWS ? m : 0;
      const int cyl = mm / CC, cxl = mm - cyl * CC;
      const char* a0 = patch + ((2 * cyl) * PC + cxl + lg) * 16;  // tap (r = 0, t = lg) of this lane's output pixel
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{bias4[i][0], bias4[i][1], bias4[i][2], bias4[i][3]};  // the bias seeds the accumulator
      uint4 af[7];
#pragma unroll
      for (int r = 0; r < 7; ++r) af[r] = *(const uint4*)(a0 + r * (PC * 16));
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&wreg[r][i], *(const bf16x8*)&af[r], acc[i], 0, 0, 0);
      // D[n = i*16 + 4*lg + q][m = lr]: this lane holds 4 consecutive channels of output pixel m
      const int cy = cy0 + cyl, cx = cx0 + cxl;
      const bool inside = m < MROWS && (unsigned)cy < (unsigned)p.CH && (unsigned)cx < (unsigned)p.CW;
      const uint32_t keep = inside ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint2 o;  // (the integer ReLU leaves no -0.0, which would not order as an unsigned pattern in the pooling below)
        o.x = relu_pk_bf16(cvt_pk_bf16(acc[i][0], acc[i][1])) & keep;
        o.y = relu_pk_bf16(cvt_pk_bf16(acc[i][2], acc[i][3])) & keep;
        *(uint2*)(ctile + m * CPITCH + (i * 16 + 4 * lg) * 2) = o;
      }
    }
    TD_LDS_BARRIER();
    // (3) 3x3 stride-2 max-pool from LDS: pooled (pyl, pxl) covers convolution rows 2pyl .. 2pyl+2, columns 2pxl .. 2pxl+2 of the tile
    // (branch-free: a pooled pixel outside the map gets an out-of-range buffer offset and the hardware drops its store - stores
    //  behind a branch cannot be counted by the compiler, and the wait for the prefetched patch below would then also wait for them)
#pragma unroll
    for (int it = 0; it < (TY * TX * 8 + 255) / 256; ++it) {
      // A wavefront pools 8 consecutive pooled pixels x 8 chunks of 16 bytes; WHICH lane takes which (pixel, chunk) is chosen for the LDS
      // banks: ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), and with the 144-byte pitch
      // pooled pixel q starts 2q sixteen-byte slots into the 256-byte bank row - lane = 8 * pixel + chunk put pixels 1, 2, 3 of a group
      // on top of each other (12 cycles per read instead of 4, tools/lds_bank_model.py; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47).
      // With quad k = (lane >> 2) & 7: chunk half = bit 1 of k, pixel = (k0 ^ k1 ^ k2) + 4 k2 + 2 (lane >> 5) a group reads both halves of
      // pixels q and q + 4 (8 slots apart): one cycle per group, except where the 8 pixels wrap around the end of a tile row.
      const int e = t + it * 256;
      const int kq = (e >> 2) & 7;
      const int c8 = ((kq & 2) << 1) + (e & 3);
      const int qe = ((e >> 6) << 3) + (((kq ^ (kq >> 1) ^ (kq >> 2)) & 1) + (kq & 4) + ((e >> 4) & 2));  // pooled pixel of the tile, row-major
      const int px_ = qe % TX, py_ = min(qe / TX, TY - 1);
      const int py = py0 + py_, px = px0 + px_;
      uint4 o = make_uint4(0, 0, 0, 0);  // non-negative bf16 values order like their bit patterns: packed unsigned maxima, two channels per operation
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const uint4 v = *(const uint4*)(ctile + ((2 * py_ + dy) * CC + 2 * px_ + dx) * CPITCH + c8 * 16);
          o.x = max_pk_nonneg_bf16(o.x, v.x);
          o.y = max_pk_nonneg_bf16(o.y, v.y);
          o.z = max_pk_nonneg_bf16(o.z, v.z);
          o.w = max_pk_nonneg_bf16(o.w, v.w);
        }
      const bool live = qe < TY * TX && py < p.PH && px < p.PW;
      const uint32_t off = live ? (uint32_t)((((size_t)img * p.PH + py) * p.PW + px) * 128 + c8 * 16) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b128(st_u32x4{o.x, o.y, o.z, o.w}, rs_y, (int)off, 0, 0);
    }
    store_patch();    // every wavefront is done reading this tile's patch since the barrier behind the convolution
    TD_LDS_BARRIER();  // the next tile's convolution rows overwrite what the pooling just read; its patch is complete
  }
}

}  // namespace td
using namespace td;

extern "C" int td_stem_pool(const void* x_pairs, const void* w_pairs, const float* bias, void* pooled, int N, int H, int W, int dtype,
                            td_stream_t stream) {
  TD_REQUIRE(x_pairs && w_pairs && bias && pooled, "td_stem_pool: null pointer");
  TD_REQUIRE(dtype == TD_BF16, "td_stem_pool: the fused stem is a bf16 kernel (the exact-fp32 mode runs conv + pool separately)");
  TD_REQUIRE(N >= 1 && H >= 2 && W >= 2 && (W & 1) == 0, "td_stem_pool: frames must have an even width (pixel pairs)");
  StemParams p;
  p.x = (const char*)x_pairs;
  p.w = (const char*)w_pairs;
  p.bias = bias;
  p.y = (char*)pooled;
  p.N = N;
  p.H = H;
  p.WP = W / 2;
  p.CH = (H + 6 - 7) / 2 + 1;
  p.CW = (W + 6 - 7) / 2 + 1;
  p.PH = (p.CH + 2 - 3) / 2 + 1;
  p.PW = (p.CW + 2 - 3) / 2 + 1;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_FUSED, dtype, 2.0 * N * p.CH * p.CW * 64.0 * 224.0, st, N * p.CH * p.CW, 64, 224, 7, 2, 0);
    prof_set_bytes(((double)N * H * W * 4 + (double)N * p.PH * p.PW * 64 + 64.0 * 224) * 2.0);
  }
  // 22-wide tiles when they divide the pooled width (res 352 -> 88 = 4 x 22: no half-empty tile column), else 16-wide
  const bool wide = p.PW % 22 == 0;
  const int TX = wide ? 22 : 16, TY = 4;
  p.tiles_y = cdiv(p.PH, TY);
  p.tiles_x = cdiv(p.PW, TX);
  const long long nt = (long long)N * p.tiles_y * p.tiles_x;
  TD_REQUIRE(nt < 2000000000LL && (double)N * p.PH * p.PW * 128.0 < 4294967000.0, "td_stem_pool: pooled tensor exceeds the 4 GiB buffer-descriptor range");
  p.n_tiles = (int)nt;
  static const int per_cu = [] { const char* e = getenv("TD_STEM_WG_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();  // (A/B: persistent workgroups per CU)
  const int grid = (int)std::min<long long>(nt, (long long)per_cu * n_cu);
  if (wide) stem_pool_kernel<4, 22><<<grid, 256, 0, st>>>(p);
  else stem_pool_kernel<4, 16><<<grid, 256, 0, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_stem_pool");
}
