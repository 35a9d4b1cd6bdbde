// A whole FROZEN bottleneck of layer1 in one pass:
//     out = relu( conv3( relu( conv2_3x3( relu( conv1(x) ) ) ) ) + identity ),   identity = x  or  downsample_1x1(x)
// (torchvision Bottleneck.forward with FrozenBatchNorm folded into weights / biases, reached through models/backbone.py:94-98;
// layer1 and the stem never train, backbone.py:82-89: no gradient passes through and none of the block's inner tensors is ever
// needed again - for the slow frames no more than for the no-grad fast frames.)
//
// Why: at res 352 the three layer1 blocks run on 88 x 88 maps of 256 channels - 4 GB per 1 000 frames and tensor.  Launched
// layer by layer a block moves ~13 GB through HBM (the 64-channel inner tensors written and read back, the 256-channel input
// read by conv1 AND as the residual) for 1.1 TFLOP: every launch is HBM-bound, 3.8 ms per block.  Fused, HBM sees the block's
// input once and its output once (7.9 GB): the 64-channel tensors live in LDS, the residual is the input tile itself.
//
// One workgroup (4 wavefronts, two workgroups per CU) owns an 8 x 16 tile of output pixels:
//   phase 1  conv1 (1x1, CIN -> 64) on the tile + its one-pixel halo (10 x 18 = 180 pixels), the input streamed through LDS in
//            64-channel chunks (double buffer, the next chunk requested into registers before the current one is multiplied);
//            result (bias, ReLU, ZERO outside the image = conv2's padding) -> LDS as bf16
//   phase 2  conv2 (3x3, 64 -> 64) on the 128 centre pixels, its 9 taps read from the LDS halo tile by address arithmetic -> LDS
//   phase 3  conv3 (1x1, 64 -> 256) (+ the downsample 1x1 of block 0 as two more k-steps over the input tile still in LDS),
//            + bias + identity + ReLU -> HBM
// The output channels are split over the wavefronts (16 / 16 / 64 per wavefront in the three phases): a wavefront's weight
// fragments are 8 - 18 registers' worth per phase, loaded from L2 at the phase head; every activation fragment is read from
// LDS by all four wavefronts (4x LDS traffic - the block stays HBM-bound: ~14 000 cycles of HBM time per tile and CU against
// ~5 000 cycles of MFMA issue and ~4 000 LDS cycles).
// LDS rows are 128 bytes (64 bf16), 16-byte chunks XOR-swizzled by (row & 7) like everywhere in this library.
#include <stdlib.h>

#include <algorithm>

#include "td_common.h"

namespace td {

typedef __bf16 bn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bn_cvt_pk(float lo, float hi) {
  bn_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return *(uint32_t*)&v;
}
typedef unsigned int bn_u32x2 __attribute__((ext_vector_type(2)));
// 8-byte output store through a buffer descriptor: a lane whose pixel lies outside the image passes an out-of-range offset and the
// hardware drops the store.  No branch around the stores: behind a branch the compiler cannot count them in its vmcnt bookkeeping
// and every later wait for an OLDER load (bias, identity rows, the next tile's prefetch) also waits for the stores' write latency.
__device__ __forceinline__ void bn_store8(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(bn_u32x2{v.x, v.y}, rs, (int)off, 0, 0);
}
typedef unsigned int bn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bn_store16(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(bn_u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 0);
}
#define TD_BN_OOB 0xFFFFFFF0u
#ifndef TD_BN_ABL
#define TD_BN_ABL 0  // timing ablations of bottleneck_resident_kernel (tools/build_variant.sh; wrong results): 1 no input loads, 2 no output stores
#endif
// Output rows leave through a wavefront-private 2-KiB LDS transposition: the MFMA layout gives a lane 4 consecutive channels of
// one pixel (8-byte pieces, a store instruction touching 16 cache lines by 32 bytes); read back as 16 bytes per lane with 8 lanes
// per pixel, a store instruction writes 8 whole 128-byte lines.
#define TD_BN_STAGE_BYTES 2048
#define TD_BN_BARRIER()                                  \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
  } while (0)

struct BneckParams {
  const char* x;   // [N][H][W][CIN] bf16
  char* out;       // [N][H][W][256] bf16
  const char* w1;  // [64][CIN]
  const char* w2;  // [64][3][3][64]  (K = (r*3 + s)*64 + c)
  const char* w3;  // [256][64]
  const char* wd;  // [256][CIN] (block 0: CIN = 64) or null
  const float *b1, *b2, *b3, *bd;
  int N, H, W;
  int tiles_y, tiles_x, n_tiles;
};

template <int CIN, bool DS>
__global__ __launch_bounds__(256, 2) void bottleneck_fused_kernel(BneckParams p) {
  static_assert(CIN == 64 || CIN == 256, "layer1 blocks");
  static_assert(!DS || CIN == 64, "the downsample branch belongs to block 0 (64 input channels)");
  constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;
  constexpr int NHALO = HH * HW, HROWS = 192, NCEN = TH * TW;  // 180 halo pixels (padded to 12 blocks of 16), 128 centre pixels
  constexpr int NC = CIN / 64;                                 // 64-channel input chunks
  constexpr int XBUFS = NC > 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char xbuf[XBUFS][HROWS * 128];
  __shared__ __attribute__((aligned(16))) char h1[HROWS * 128];
  __shared__ __attribute__((aligned(16))) char h2own[NC > 1 ? 16 : NCEN * 128];
  __shared__ __attribute__((aligned(16))) char ostage[4][TD_BN_STAGE_BYTES];
  char* h2 = NC > 1 ? xbuf[1] : h2own;  // CIN = 256: the second chunk buffer is dead after phase 1
  const int t0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
  // t / lr / lg are re-"defined" at the head of every tile (an empty asm the optimiser cannot see through): all the per-lane
  // address arithmetic below is then recomputed per tile - a handful of VALU operations - instead of being hoisted out of the
  // persistent loop, kept live across all three phases and spilled (every scratch access would wait behind the phase-3 stores)
  int t = t0, lr = t0 & 15, lg = (t0 & 63) >> 4, lane = t0 & 63;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (uint32_t)((size_t)p.N * p.H * p.W * 512), 0x00020000);

  // ---- input chunk loader: 192 rows x 8 chunks of 16 bytes = 6 per thread ----
  uint4 pre[6];
  bool ok[6];
  auto fetch_chunk = [&](int tile, int kc) {
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = t + j * 256;
      const int row = e >> 3, c = e & 7;
      const int hy = row / HW, hx = row - hy * HW;
      const int y = y0 + hy, x = x0 + hx;
      // branch-free: every lane always issues its 6 loads (from the tensor's first bytes when the pixel lies outside the frame /
      // the tile does not exist) and the zero is selected afterwards - a load under a branch cannot be counted by the compiler's
      // vmcnt bookkeeping, which then waits for EVERYTHING in flight (this prefetch included) at the next weight-fragment use
      ok[j] = tile < p.n_tiles && row < NHALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const size_t off = ok[j] ? ((((size_t)img * p.H + y) * p.W + x) * CIN + kc * 64 + c * 8) * 2 : (size_t)0;
      pre[j] = *(const uint4*)(p.x + off);
    }
  };
  auto store_chunk = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = t + j * 256;
      const int row = e >> 3, c = e & 7;
      *(uint4*)(buf + row * 128 + ((c ^ (row & 7)) << 4)) = ok[j] ? pre[j] : make_uint4(0, 0, 0, 0);
    }
  };
  auto frag = [&](const char* buf, int row, int c16) -> uint4 { return *(const uint4*)(buf + row * 128 + ((c16 ^ (row & 7)) << 4)); };

  fetch_chunk(blockIdx.x, 0);
  store_chunk(xbuf[0]);
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    asm volatile("" : "+v"(t), "+v"(lr), "+v"(lg), "+v"(lane));
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;  // first centre pixel
    // ================= phase 1: conv1 on the halo tile, channels 16*wave .. +15 =================
    {
      // (requested behind the previous tile's output stores in the in-order VMEM queue: the first MFMA below may wait for their
      //  write latency - the other workgroup of the CU computes meanwhile; keeping these 32 registers resident instead spills)
      uint4 w1r[NC][2];
#pragma unroll
      for (int kc = 0; kc < NC; ++kc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w1r[kc][ks] = *(const uint4*)(p.w1 + ((size_t)(16 * wave + lr) * CIN + kc * 64 + ks * 32 + lg * 8) * 2);
      float b1v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b1v[q] = p.b1[16 * wave + 4 * lg + q];
      f32x4 acc[12];
#pragma unroll
      for (int mb = 0; mb < 12; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      TD_BN_BARRIER();  // chunk 0 is complete in xbuf[0] (stored at the end of the previous tile / before the loop)
#pragma unroll
      for (int kc = 0; kc < NC; ++kc) {
        if (kc + 1 < NC) fetch_chunk(tile, kc + 1);
        const char* xb = xbuf[kc & (XBUFS - 1)];
        // groups of 4 fragments, double-buffered: group g + 1 is requested before the MFMAs of group g, and the scheduler may not
        // move anything across a group boundary (left alone it hoists every fragment read of the phase and spills)
        uint4 fr[2][4];
        auto load4 = [&](int g, uint4 (&dst)[4]) {  // g = ks * 3 + (which third of the 12 row blocks)
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[j] = frag(xb, ((g % 3) * 4 + j) * 16 + lr, (g / 3) * 4 + lg);
        };
        load4(0, fr[0]);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          if (g + 1 < 6) load4(g + 1, fr[(g + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[(g % 3) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w1r[kc][g / 3], *(const bf16x8*)&fr[g & 1][j], acc[(g % 3) * 4 + j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kc + 1 < NC) {
          store_chunk(xbuf[(kc + 1) & (XBUFS - 1)]);  // that buffer was last read two chunks ago: every wavefront has passed a barrier since
          TD_BN_BARRIER();
        }
      }
#pragma unroll
      for (int mb = 0; mb < 12; ++mb) {
        const int row = mb * 16 + lr;
        const int hy = row / HW, hx = row - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inside = row < NHALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;  // outside: conv2's zero padding
        uint2 o;
        o.x = inside ? bn_cvt_pk(fmaxf(acc[mb][0] + b1v[0], 0.f), fmaxf(acc[mb][1] + b1v[1], 0.f)) : 0u;
        o.y = inside ? bn_cvt_pk(fmaxf(acc[mb][2] + b1v[2], 0.f), fmaxf(acc[mb][3] + b1v[3], 0.f)) : 0u;
        *(uint2*)(h1 + row * 128 + (((2 * wave + (lg >> 1)) ^ (row & 7)) << 4) + (lg & 1) * 8) = o;
      }
    }
    TD_BN_BARRIER();  // h1 complete
    // ================= phase 2: conv2 3x3 on the centre pixels, channels 16*wave .. +15 =================
    {
      uint4 w2r[18];
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) w2r[ks] = *(const uint4*)(p.w2 + ((size_t)(16 * wave + lr) * 576 + ks * 32 + lg * 8) * 2);
      // next tile's first input chunk: in flight during phases 2 and 3.  Requested BEHIND this phase's weights: loads return in
      // order, so the first MFMA (which waits for w2r[0]) would otherwise also wait for the HBM latency of the prefetch
      float b2v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b2v[q] = p.b2[16 * wave + 4 * lg + q];
      __builtin_amdgcn_sched_barrier(0);  // (keeps the issue order: the compiler would otherwise be free to put the prefetch first)
      fetch_chunk(tile + gridDim.x, 0);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[TH];
#pragma unroll
      for (int mb = 0; mb < TH; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      // 18 k-steps (tap, channel half) x 2 groups of 4 fragments, double-buffered like phase 1
      uint4 fr[2][4];
      auto load4 = [&](int g, uint4 (&dst)[4]) {  // g = ks * 2 + (rows 0-3 | rows 4-7 of the tile)
        const int ks = g >> 1, tap = ks >> 1, r = tap / 3, s = tap - 3 * r;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = frag(h1, ((g & 1) * 4 + j + r) * HW + lr + s, (ks & 1) * 4 + lg);  // centre pixel (mb, lr) -> halo pixel (mb + r, lr + s)
      };
      load4(0, fr[0]);
#pragma unroll
      for (int g = 0; g < 36; ++g) {
        if (g + 1 < 36) load4(g + 1, fr[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[(g & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w2r[g >> 1], *(const bf16x8*)&fr[g & 1][j], acc[(g & 1) * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < TH; ++mb) {
        const int row = mb * 16 + lr;
        uint2 o;
        o.x = bn_cvt_pk(fmaxf(acc[mb][0] + b2v[0], 0.f), fmaxf(acc[mb][1] + b2v[1], 0.f));
        o.y = bn_cvt_pk(fmaxf(acc[mb][2] + b2v[2], 0.f), fmaxf(acc[mb][3] + b2v[3], 0.f));
        *(uint2*)(h2 + row * 128 + (((2 * wave + (lg >> 1)) ^ (row & 7)) << 4) + (lg & 1) * 8) = o;
      }
    }
    TD_BN_BARRIER();  // h2 complete
    // ================= phase 3: conv3 (+ downsample) + identity + ReLU, channels 64*wave .. +63 =================
    {
      uint4 w3r[4][2], wdr[DS ? 4 : 1][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          w3r[i][ks] = *(const uint4*)(p.w3 + ((size_t)(64 * wave + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
          if constexpr (DS) wdr[i][ks] = *(const uint4*)(p.wd + ((size_t)(64 * wave + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
        }
      float b3v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 64 * wave + 16 * i + 4 * lg + q;
          b3v[i][q] = p.b3[n] + (DS ? p.bd[n] : 0.f);
        }
      const int x = x0 + lr;
      // identity rows of the whole tile (8 centre rows x 4 x 8 bytes per lane; the input tile was read moments ago: L2 / Infinity
      // Cache), ALL requested before the first output store: vmcnt counts loads and stores in one in-order queue on this ISA, so a
      // load issued behind a store cannot be waited for without waiting for the store's (long) write latency as well
      uint2 res[DS ? 1 : TH][4];
      if constexpr (!DS) {
#pragma unroll
        for (int mb = 0; mb < TH; ++mb) {
          const int y = y0 + mb;
          const bool ok = y < p.H && x < p.W;
          const char* base = p.x + ((((size_t)img * p.H + (ok ? y : 0)) * p.W + (ok ? x : 0)) * CIN + 64 * wave + 4 * lg) * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) res[mb][i] = *(const uint2*)(base + 32 * i);
        }
      }
#pragma unroll
      for (int mb = 0; mb < TH; ++mb) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{b3v[i][0], b3v[i][1], b3v[i][2], b3v[i][3]};  // the bias seeds the accumulator
        const int row = mb * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint4 a = frag(h2, row, ks * 4 + lg);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w3r[i][ks], *(const bf16x8*)&a, acc[i], 0, 0, 0);
        }
        if constexpr (DS) {  // identity = downsample(x): two more k-steps over the input tile (halo pixel (mb + 1, lr + 1))
          const int hp = (mb + 1) * HW + lr + 1;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const uint4 a = frag(xbuf[0], hp, ks * 4 + lg);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&wdr[i][ks], *(const bf16x8*)&a, acc[i], 0, 0, 0);
          }
        }
        char* stg = ostage[wave];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = acc[i][q];  // (bias already in the accumulator)
          if constexpr (!DS) {
            const uint2 r2 = res[mb][i];
            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
          }
          uint2 o;
          o.x = bn_cvt_pk(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
          o.y = bn_cvt_pk(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
          *(uint2*)(stg + lr * 128 + (((2 * i + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int y = y0 + mb;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {  // 8 pixels x 128 bytes per store instruction: lane = (pixel, 16-byte chunk)
          const int px_ = ps * 8 + (lane >> 3), c = lane & 7;
          const uint4 o16 = *(const uint4*)(stg + px_ * 128 + ((c ^ (px_ & 7)) << 4));
          const int xo = x0 + px_;
          const uint32_t off = (y < p.H && xo < p.W) ? (uint32_t)(((((size_t)img * p.H + y) * p.W + xo) * 256 + 64 * wave + 8 * c) * 2) : TD_BN_OOB;
          bn_store16(rs_out, off, o16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next row block overwrites the region
      }
    }
    TD_BN_BARRIER();  // every wavefront is done with h2 / the input tile: the next tile's first chunk may land
    store_chunk(xbuf[0]);
  }
}

// ------------------------------------------------------------------------------------------------
// Variant for the 256-channel blocks (layer1.1, layer1.2): 8 x 8 tiles with the WHOLE input tile resident in LDS.
// The streaming kernel above moves 220 KB per 128-pixel tile for them (input halo 92 KB + identity re-read 64 KB + output 64 KB,
// 1.72 KB per pixel against 1.0 KB algorithmic) and runs at the HBM rate of THOSE bytes (3.8 ms per 1 000 frames, no better than
// the layer-by-layer path).  Here the 10 x 10 halo tile of all 256 channels (51 KB) is loaded once, all four 64-channel chunks in
// flight together, conv1 runs over it, and conv3's identity is read back from the same LDS bytes: 83 KB per 64-pixel tile
// (1.3 KB per pixel), and 51 KB per workgroup in flight during its load phase instead of 24.  No cross-tile prefetch (no
// registers or LDS left for it): the two workgroups of a CU alternate between their load and compute phases.
__global__ __launch_bounds__(256, 2) void bottleneck_resident_kernel(BneckParams p) {
  constexpr int CIN = 256, NC = 4;
  constexpr int TH = 8, TW = 8, HW = TW + 2;
  constexpr int NHALO = (TH + 2) * HW, HROWS = 112, MB1 = HROWS / 16;  // 100 halo pixels -> 7 blocks of 16 rows
  constexpr int NCEN = TH * TW, MB2 = NCEN / 16;                       // 64 centre pixels -> 4 blocks
  constexpr int LD = HROWS * 8 * NC / 256;                             // 16-byte elements per thread for the whole tile: 14
  __shared__ __attribute__((aligned(16))) char xall[NC][HROWS * 128];
  __shared__ __attribute__((aligned(16))) char h1[HROWS * 128];
  __shared__ __attribute__((aligned(16))) char ostage[4][TD_BN_STAGE_BYTES];
  char* h2 = h1;  // conv2's output overwrites conv1's (behind a barrier: every wavefront has finished reading h1)
  const int t0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
  int t = t0, lr = t0 & 15, lg = (t0 & 63) >> 4, lane = t0 & 63;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (uint32_t)((size_t)p.N * p.H * p.W * 512), 0x00020000);
  auto frag = [&](const char* buf, int row, int c16) -> uint4 { return *(const uint4*)(buf + row * 128 + ((c16 ^ (row & 7)) << 4)); };

  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    asm volatile("" : "+v"(t), "+v"(lr), "+v"(lg), "+v"(lane));
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    // ---- the whole input halo tile -> LDS: 14 branch-free 16-byte loads per thread, all in flight together ----
    {
      uint4 v[LD];
      bool ok[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int e = t + j * 256;            // element (row, channel chunk of 8): 32 chunks per pixel row of 256 channels
        const int row = e >> 5, c32 = e & 31;
        const int hy = row / HW, hx = row - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        ok[j] = row < NHALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const size_t off = ok[j] ? ((((size_t)img * p.H + y) * p.W + x) * CIN + c32 * 8) * 2 : (size_t)0;
#if TD_BN_ABL & 1
        v[j] = make_uint4(e, off, 0, 0);
#else
        v[j] = *(const uint4*)(p.x + off);
#endif
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const int e = t + j * 256;
        const int row = e >> 5, c32 = e & 31;
        *(uint4*)(xall[c32 >> 3] + row * 128 + (((c32 & 7) ^ (row & 7)) << 4)) = ok[j] ? v[j] : make_uint4(0, 0, 0, 0);
      }
    }
    // ================= phase 1: conv1 on the halo tile, channels 16*wave .. +15 =================
    {
      uint4 w1r[NC][2];
#pragma unroll
      for (int kc = 0; kc < NC; ++kc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w1r[kc][ks] = *(const uint4*)(p.w1 + ((size_t)(16 * wave + lr) * CIN + kc * 64 + ks * 32 + lg * 8) * 2);
      float b1v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b1v[q] = p.b1[16 * wave + 4 * lg + q];
      f32x4 acc[MB1];
#pragma unroll
      for (int mb = 0; mb < MB1; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      TD_BN_BARRIER();  // the tile is complete in LDS
      // 8 k-steps x 7 row blocks, double-buffered groups: (k-step ks8, rows 0-3) and (ks8, rows 4-6)
      uint4 fr[2][4];
      auto load_g = [&](int g, uint4 (&dst)[4]) {
        const int ks8 = g >> 1, hi = g & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (hi * 4 + j < MB1) dst[j] = frag(xall[ks8 >> 1], (hi * 4 + j) * 16 + lr, (ks8 & 1) * 4 + lg);
      };
      load_g(0, fr[0]);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g + 1 < 16) load_g(g + 1, fr[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((g & 1) * 4 + j < MB1)
            acc[(g & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w1r[g >> 2][(g >> 1) & 1], *(const bf16x8*)&fr[g & 1][j], acc[(g & 1) * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < MB1; ++mb) {
        const int row = mb * 16 + lr;
        const int hy = row / HW, hx = row - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inside = row < NHALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;  // outside: conv2's zero padding
        uint2 o;
        o.x = inside ? bn_cvt_pk(fmaxf(acc[mb][0] + b1v[0], 0.f), fmaxf(acc[mb][1] + b1v[1], 0.f)) : 0u;
        o.y = inside ? bn_cvt_pk(fmaxf(acc[mb][2] + b1v[2], 0.f), fmaxf(acc[mb][3] + b1v[3], 0.f)) : 0u;
        *(uint2*)(h1 + row * 128 + (((2 * wave + (lg >> 1)) ^ (row & 7)) << 4) + (lg & 1) * 8) = o;
      }
    }
    TD_BN_BARRIER();  // h1 complete
    // ================= phase 2: conv2 3x3 on the 64 centre pixels, channels 16*wave .. +15 =================
    {
      uint4 w2r[18];
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) w2r[ks] = *(const uint4*)(p.w2 + ((size_t)(16 * wave + lr) * 576 + ks * 32 + lg * 8) * 2);
      float b2v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b2v[q] = p.b2[16 * wave + 4 * lg + q];
      f32x4 acc[MB2];
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int cyl = lr >> 3, cxl = lr & 7;  // centre pixel of row block mb, lane lr: (2 * mb + cyl, cxl)
      uint4 fr[2][MB2];
      auto load_k = [&](int ks, uint4 (&dst)[MB2]) {
        const int tap = ks >> 1, r = tap / 3, s = tap - 3 * r;
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) dst[mb] = frag(h1, (2 * mb + cyl + r) * HW + cxl + s, (ks & 1) * 4 + lg);
      };
      load_k(0, fr[0]);
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + 1 < 18) load_k(ks + 1, fr[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w2r[ks], *(const bf16x8*)&fr[ks & 1][mb], acc[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      TD_BN_BARRIER();  // every wavefront has read its last h1 fragment: h2 may overwrite it
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb) {
        const int row = mb * 16 + lr;
        uint2 o;
        o.x = bn_cvt_pk(fmaxf(acc[mb][0] + b2v[0], 0.f), fmaxf(acc[mb][1] + b2v[1], 0.f));
        o.y = bn_cvt_pk(fmaxf(acc[mb][2] + b2v[2], 0.f), fmaxf(acc[mb][3] + b2v[3], 0.f));
        *(uint2*)(h2 + row * 128 + (((2 * wave + (lg >> 1)) ^ (row & 7)) << 4) + (lg & 1) * 8) = o;
      }
    }
    TD_BN_BARRIER();  // h2 complete
    // ================= phase 3: conv3 + identity (from the LDS tile) + ReLU, channels 64*wave .. +63 =================
    {
      uint4 w3r[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w3r[i][ks] = *(const uint4*)(p.w3 + ((size_t)(64 * wave + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
      float b3v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) b3v[i][q] = p.b3[64 * wave + 16 * i + 4 * lg + q];
      const int cyl = lr >> 3, cxl = lr & 7;
      const int x = x0 + cxl;
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{b3v[i][0], b3v[i][1], b3v[i][2], b3v[i][3]};  // the bias seeds the accumulator
        const int row = mb * 16 + lr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint4 a = frag(h2, row, ks * 4 + lg);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w3r[i][ks], *(const bf16x8*)&a, acc[i], 0, 0, 0);
        }
        const int cy = 2 * mb + cyl;
        const int hp = (cy + 1) * HW + cxl + 1;  // this pixel's row in the input tile: channels 64*wave .. of chunk `wave`
        char* stg = ostage[wave];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint2 r2 = *(const uint2*)(xall[wave] + hp * 128 + (((2 * i + (lg >> 1)) ^ (hp & 7)) << 4) + (lg & 1) * 8);
          float v[4];
          v[0] = acc[i][0] + __uint_as_float(r2.x << 16);
          v[1] = acc[i][1] + __uint_as_float(r2.x & 0xffff0000u);
          v[2] = acc[i][2] + __uint_as_float(r2.y << 16);
          v[3] = acc[i][3] + __uint_as_float(r2.y & 0xffff0000u);
          uint2 o;
          o.x = bn_cvt_pk(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
          o.y = bn_cvt_pk(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
          *(uint2*)(stg + lr * 128 + (((2 * i + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {  // 8 pixels x 128 bytes per store instruction: lane = (pixel, 16-byte chunk)
          const int px_ = ps * 8 + (lane >> 3), c = lane & 7;  // pixel px_ of the row block: tile row 2 * mb + ps, column lane >> 3
          const uint4 o16 = *(const uint4*)(stg + px_ * 128 + ((c ^ (px_ & 7)) << 4));
          const int yo = y0 + 2 * mb + ps, xo = x0 + (lane >> 3);
          const uint32_t off = (yo < p.H && xo < p.W) ? (uint32_t)(((((size_t)img * p.H + yo) * p.W + xo) * 256 + 64 * wave + 8 * c) * 2) : TD_BN_OOB;
#if TD_BN_ABL & 2
          asm volatile("" ::"v"(off), "v"(o16.x), "v"(o16.y), "v"(o16.z), "v"(o16.w));
#else
          bn_store16(rs_out, off, o16);
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next row block overwrites the region
      }
    }
    TD_BN_BARRIER();  // every wavefront is done with the tile: the next one may overwrite it
  }
}

}  // namespace td
using namespace td;

extern "C" int td_bottleneck_fused(const void* x, void* out, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                                   const float* b3, const void* wd, const float* bd, int N, int H, int W, int Cin, int dtype, td_stream_t stream) {
  TD_REQUIRE(x && out && w1 && b1 && w2 && b2 && w3 && b3, "td_bottleneck_fused: null pointer");
  TD_REQUIRE(dtype == TD_BF16, "td_bottleneck_fused: bf16 only (the exact-fp32 mode runs the block layer by layer)");
  TD_REQUIRE((Cin == 64 && wd && bd) || (Cin == 256 && !wd), "td_bottleneck_fused: a layer1 block (64 -> 64 -> 256 with downsample, or 256 -> 64 -> 256)");
  TD_REQUIRE(N >= 1 && H >= 1 && W >= 1 && (double)N * H * W * 256 < 2147483647.0, "td_bottleneck_fused: bad geometry");
  BneckParams p;
  p.x = (const char*)x; p.out = (char*)out;
  p.w1 = (const char*)w1; p.w2 = (const char*)w2; p.w3 = (const char*)w3; p.wd = (const char*)wd;
  p.b1 = b1; p.b2 = b2; p.b3 = b3; p.bd = bd;
  p.N = N; p.H = H; p.W = W;
  static const int resident = [] { const char* e = getenv("TD_BNECK_RESIDENT"); return e ? atoi(e) : 1; }();
  const bool res256 = Cin == 256 && resident;  // 8 x 8 tiles, whole input tile in LDS (bottleneck_resident_kernel)
  p.tiles_y = cdiv(H, 8);
  p.tiles_x = cdiv(W, res256 ? 8 : 16);
  const long long nt = (long long)N * p.tiles_y * p.tiles_x;
  TD_REQUIRE(nt < 2000000000LL, "td_bottleneck_fused: too many tiles");
  p.n_tiles = (int)nt;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  const double rows = (double)N * H * W;
  if (prof) {
    prof_begin(TD_PROF_FUSED, dtype, 2.0 * rows * (64.0 * Cin + 64.0 * 576 + 256.0 * 64 + (wd ? 256.0 * 64 : 0.0)), st, (int)std::min(rows, 2147483647.0), 256, Cin, 3, 1, 0);
    prof_set_bytes((rows * (Cin + 256.0) + 64.0 * Cin + 64.0 * 576 + 256.0 * 64) * 2.0);
  }
  static const int per_cu = [] { const char* e = getenv("TD_BNECK_WG_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();  // (A/B: persistent workgroups per CU)
  const int grid = (int)std::min<long long>(nt, (long long)per_cu * n_cu);
  if (Cin == 64) bottleneck_fused_kernel<64, true><<<grid, 256, 0, st>>>(p);
  else if (res256) bottleneck_resident_kernel<<<grid, 256, 0, st>>>(p);
  else bottleneck_fused_kernel<256, false><<<grid, 256, 0, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_bottleneck_fused");
}
