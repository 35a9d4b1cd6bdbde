// A whole FROZEN bottleneck of layer1 in one pass:
//     out = relu( conv3( relu( conv2_3x3( relu( conv1(x) ) ) ) ) + identity ),   identity = x  or  downsample_1x1(x)
// (torchvision Bottleneck.forward with FrozenBatchNorm folded into weights / biases, reached through models/backbone.py:94-98;
// layer1 and the stem never train, backbone.py:82-89: no gradient passes through and none of the block's inner tensors is ever
// needed again - for the slow frames no more than for the no-grad fast frames.)
//
// Why: at res 352 the three layer1 blocks run on 88 x 88 maps of 256 channels - 4 GB per 1 000 frames and tensor.  Launched
// layer by layer a block moves ~13 GB through HBM (the 64-channel inner tensors written and read back, the 256-channel input
// read by conv1 AND as the residual) for 1.1 TFLOP: every launch is HBM-bound, 3.8 ms per block.  Fused, HBM sees the block's
// input once (plus the halo) and its output once: the 64-channel tensors live in LDS, the residual is the input tile itself.
//
// One persistent workgroup of eight wavefronts per CU owns an output tile (8 x 8 pixels for the 256-channel blocks, 8 x 16 for
// block 0) and runs three phases on it:
//   phase 1  conv1 (1x1, CIN -> 64) on the tile + its one-pixel halo; result (bias, ReLU, ZERO outside the image = conv2's
//            padding) -> LDS as bf16
//   phase 2  conv2 (3x3, 64 -> 64) on the centre pixels, its 9 taps read from the LDS halo tile by address arithmetic -> LDS
//   phase 3  conv3 (1x1, 64 -> 256) (+ the downsample 1x1 of block 0 over the input tile still in LDS) + bias + identity + ReLU -> HBM
// Two kernels: bottleneck_resident3_kernel (256 input channels, layer1.1 / layer1.2) and bottleneck_first3_kernel (64 input channels
// + downsample branch, layer1.0).  (Rounds 2-3 had a four-wavefront, register-staged form of each - bottleneck_fused_kernel /
// bottleneck_resident_kernel, 3.3 / 2.1 ms per 1 000 frames; removed in round 5, their measurements are quoted below where they
// explain a design decision.)
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "td_common.h"

namespace td {

typedef __bf16 bn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bn_cvt_pk(float lo, float hi) {
  bn_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return *(uint32_t*)&v;
}
// ReLU of two packed bf16: as 16-bit integers, max(x, 0) (v_pk_max_i16) - a negative value (sign bit set, -0.0 included) becomes +0.0,
// a non-negative one is unchanged.  ONE operation per two values behind the conversion; fmaxf(x, 0.f) before it is two per value
// (the compiler quiets a possible signalling NaN with a v_max x, x first).
// (As asm: written with vector types the compiler splits the conversion in front of it into two half conversions and a v_perm.  The
// CONVERSION must stay the compiler's own instruction: it is the first reader of an MFMA result, and the wait states an MFMA -> VALU
// read needs are inserted by the compiler's hazard pass, which does not look into asm operands - a v_cvt_pk_bf16_f32 written as
// asm read accumulators the matrix pipe had not written yet.)
__device__ __forceinline__ uint32_t bn_relu_pk(uint32_t v) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ uint2 bn_relu_pack4(const f32x4& a) { return make_uint2(bn_relu_pk(bn_cvt_pk(a[0], a[1])), bn_relu_pk(bn_cvt_pk(a[2], a[3]))); }
#ifndef TD_BN_XCD_CONTIG
#define TD_BN_XCD_CONTIG 1  // an XCD's workgroups walk a contiguous range of tiles (0: tile = blockIdx + k * gridDim; A/B builds)
#endif
// The walk over tiles (tile += gridDim.x) without a division per tile: (image, tile row, tile column) advanced with carries.
struct BnTile {
  int img, ty, tx;
};
// Which tiles a workgroup takes.  Workgroups go to the 8 XCDs round-robin (XCD = blockIdx & 7), each XCD has its own L2, and neighbouring
// tiles share their halo (an 8 x 8 tile reads 10 x 10 pixels): with tile = blockIdx + k * gridDim no two neighbours ever met in one L2
// and every halo pixel was fetched from beyond it once per tile that needs it (traffic 1.22x of the algorithmic bytes).  Now XCD x takes the
// contiguous range [x * ceil(n / 8), ...) and its 32 workgroups walk it 32 tiles at a time: at res 352 that window is three tile rows of
// one frame.  (Grids that are not a multiple of 8 - fewer tiles than CUs - keep the old order.)
struct BnWalk {
  int start, stride, count;
};
__device__ __forceinline__ BnWalk bn_walk(int n_tiles) {
  const int b = (int)blockIdx.x, g = (int)gridDim.x;
  BnWalk w;
  if ((g & 7) == 0 && TD_BN_XCD_CONTIG) {
    const int x = b & 7, l = b >> 3, per = (n_tiles + 7) >> 3;
    const int lim = min(per, n_tiles - x * per);
    w.stride = g >> 3;
    w.start = x * per + l;
    w.count = lim > l ? (lim - l + w.stride - 1) / w.stride : 0;
  } else {
    w.stride = g;
    w.start = b;
    w.count = b < n_tiles ? (n_tiles - b + g - 1) / g : 0;
  }
  return w;
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
// the same with a per-launch lane offset and a SCALAR offset for everything that changes (tile, row block): no vector arithmetic per store
__device__ __forceinline__ void bn_store16s(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, uint4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(bn_u32x4{v.x, v.y, v.z, v.w}, rs, (int)voff, (int)soff, 0);
}
// 8-byte LDS store as inline asm.  A DS WRITE the compiler can see is preceded by s_waitcnt vmcnt(0) whenever an LDS-DMA may be in
// flight (its wait-count pass applies alias information to DS reads only) - in the double-buffered kernel that would wait for the
// NEXT tile's pieces at the first result written.  Ordering is what the surrounding code provides anyway: DS operations of a wavefront
// execute in order, and every cross-wavefront hand-over goes through TD_BN_BARRIER (lgkmcnt(0) + s_barrier).
__device__ __forceinline__ void bn_lds_store8(char* p, uint2 v) {
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)p;
  asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}
// LDS banks (round 6).  ds_read_b128 is served in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -
// and a group takes one cycle only if its 16 addresses fall on 16 different 16-byte slots of the 256-byte bank row.  With the MFMA operand
// layout (lane = 16 * k-chunk + row) that is two k-chunks x 8 + 8 rows per group:
//   * 128-byte rows with chunk ^ (row & 7) (the input tiles): conflict-free for 16 CONSECUTIVE rows, and for any 16 rows whose (row & 7)
//     take every value once per k-chunk of a group;
//   * padded rows without a swizzle (h1 / h2, so that a tap is "lane base + immediate"): consecutive rows need a pitch of 160 bytes
//     (10 slots: rows r and r + 8 land 80 = 5 x 16 slots apart, i.e. on the other half of the bank row); the 144 bytes of rounds 4 - 5
//     were two-way for consecutive rows and THREE-way for the 2 x 8-pixel row blocks of the 256-channel kernel (pixels 0 and 14 of a
//     block share a slot for every pitch).  rocprofv3 had it in plain sight: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 / 0.44
//     (profiles/r06_pmc_LDS_per_kernel_before_layer1_rework.csv), the sum of exactly these patterns;
//   * the 256-channel kernel's row blocks are therefore 8 rows x 2 COLUMNS of the tile: halo rows 10 apart x 2 adjacent - with the 160-byte
//     pitch every tap's fragment read, h2's reads and the identity reads from the swizzled input tile are all one cycle per group.
// The 8-byte result stores pay for it (pitch 160: four rows per 128-byte window, 16 instead of 8 cycles), 6 per wavefront and tile
// against 48 fragment reads.  tools/lds_bank_model.py is the model these statements were checked with.
#define TD_BN_OOB 0xFFFFFFF0u
#ifndef TD_BN_ABL
#define TD_BN_ABL 0  // timing ablations of the bottleneck_resident kernels (tools/build_variant.sh; wrong results): 1 no input loads, 2 no output stores
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

// ------------------------------------------------------------------------------------------------
// The 256-channel blocks (layer1.1, layer1.2): 8 x 8 tiles with the WHOLE 10 x 10 halo tile of all 256 input channels resident in
// LDS (51 KiB; conv3's identity is read back from the same bytes: 1.3 KB of HBM traffic per pixel against 1.0 KB algorithmic).
// Round 4 form: 3.30 -> 2.68 ms per 1 000 frames.  With its loads AND its stores compiled out the round-3 form (four wavefronts, two
// workgroups per CU, input staged through registers) still needed 2.57 of its 3.29 ms (TD_BN_ABL): it was bound by ~1 700 VALU instructions per
// wavefront and 64-pixel tile around 160 MFMAs - per-element index arithmetic of the register-staged input pass (a division by 10
// per 16-byte element), a swizzle recomputed for each of the 72 conv2 fragment reads (the halo row changes with tap and row
// block), `inside` tests with another division per row block, 35 weight-fragment loads per tile, the identity unpacked and added
// on the VALU - and each of a CU's two 80-KiB workgroups waits out its own load phase.  Same tiling and phases, but:
//   * ONE workgroup of eight wavefronts per CU with the input tile DOUBLE-BUFFERED (2 x 56 KiB + 16 KiB h1 / h2 + 16 KiB output
//     staging = 144 KiB): the DMA pieces of tile t + 1 (7 x 1 KiB per wavefront: 8 halo rows x one 64-channel chunk each, the
//     XOR swizzle applied on the source side, out-of-image rows = out-of-range offset = zero fill) are issued before tile t is
//     computed and waited for by count; no staging registers, no ds_write pass;
//   * conv1 / conv2 outputs use a PADDED row pitch (144 bytes) instead of the XOR swizzle: every fragment address of phase 2 is
//     "lane base + compile-time constant" (ds_read immediates), no per-read arithmetic;
//   * every weight fragment and bias of the block lives in registers for the whole launch (136 + 24): nothing but DMA pieces and
//     output stores is ever in the vector-memory queue; each phase's work is split eight ways: wavefront w -> channel group
//     cg = w & 3 (16 / 16 / 64 channels in the three phases) x row-block half w >> 2;
//   * biases seed the accumulators, and the identity is one more MFMA per fragment against a unit-matrix block;
//   * tiles whose halo lies inside the image (81 of 121 per frame at res 352) take a path without any validity arithmetic.
// Round 6 (2.63 -> 2.35 ms per 1 000 frames, block 0: 1.61 -> 1.37; tools/fused_l1_time.py, old and new library in one call): the
// kernel issues instructions, it does not wait for memory - ~960 per wavefront and tile around 92 MFMAs.  (1) h1 / h2 at a 160-byte
// pitch and row blocks of 8 rows x 2 columns: every fragment read is one LDS cycle per lane group (see "LDS banks" above; -5 %);
// (2) the tile walk carries (image, row, column) instead of dividing, and an interior tile's seven DMA pieces take a per-launch
// register + a scalar offset (310 -> ~60 instructions in front of the first barrier); (3) ReLU on the packed pair behind the
// conversion (10 -> 4 VALU operations per accumulator fragment).  ~640 instructions per wavefront and tile now.
// What was left in round 4 (ablations of that form): 1.83 ms with no memory operation at all, +0.1 for the stores, +0.5 for the loads -
// a tile's compute (3.9 us) does not cover the landing time of the next tile's 51 KiB under load, and there is no LDS for a
// third tile.  Measured and dropped: touching tile t + 2's lines a tile earlier to pull them into L2 (2.98 instead of 2.69 ms:
// the extra requests cost more than the latency they hide); the same restructuring with two 4-wavefront workgroups per CU and a
// single input buffer (2.74 ms).
__global__ __launch_bounds__(512, 2) void bottleneck_resident3_kernel(BneckParams p) {
  constexpr int CIN = 256, NC = 4;
  constexpr int TH = 8, TW = 8, HW = TW + 2;
  constexpr int HROWS = 112;   // 100 halo pixels -> 7 blocks of 16 rows
  constexpr int NP = 7;        // DMA pieces per wavefront and tile: row groups half*7 .. half*7 + 6 of channel chunk cg
  constexpr int HP = 160;      // row pitch of h1 / h2: 128 bytes + 32 (see "LDS banks" above: conflict-free ds_read_b128 without a swizzle)
  constexpr int XBUF = NC * HROWS * 128;
  // two distinct LDS objects: the compiler's wait-count pass then knows that a fragment read of one buffer cannot alias the DMA that
  // is filling the other (one array would put a vmcnt(0) in front of the first read of every phase)
  __shared__ __attribute__((aligned(16))) char xall0[XBUF];
  __shared__ __attribute__((aligned(16))) char xall1[XBUF];
  __shared__ __attribute__((aligned(16))) char h1[HROWS * HP];
  __shared__ __attribute__((aligned(16))) char h2[TH * TW * HP];  // (its own 10 KiB: no barrier between the last h1 fragment read and the h2 stores)
  __shared__ __attribute__((aligned(16))) char ostage[8][TD_BN_STAGE_BYTES];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cg = wave & 3, half = wave >> 2;
  const int lane0 = t & 63;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const uint32_t x_bytes = (uint32_t)((size_t)p.N * p.H * p.W * CIN * 2);  // (host: N*H*W*256 < 2^31)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (uint32_t)((size_t)p.N * p.H * p.W * 512), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_p;

  // ---- resident for the whole launch: every weight fragment and bias of the block.  Everything else that depends on the lane is
  // recomputed per tile behind an empty asm the optimiser cannot see through (a dozen VALU operations per phase): hoisted out of
  // the tile loop it is spilled, and a spill reload is waited for with vmcnt(0) - which drains the next tile's DMA ----
  const int lr = lane0 & 15, lg = lane0 >> 4;
  uint4 w1r[NC][2], w2r[18], w3r[4][2];
#pragma unroll
  for (int kc = 0; kc < NC; ++kc)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w1r[kc][ks] = *(const uint4*)(p.w1 + ((size_t)(16 * cg + lr) * CIN + kc * 64 + ks * 32 + lg * 8) * 2);
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) w2r[ks] = *(const uint4*)(p.w2 + ((size_t)(16 * cg + lr) * 576 + ks * 32 + lg * 8) * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w3r[i][ks] = *(const uint4*)(p.w3 + ((size_t)(64 * cg + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
  float b1v[4], b2v[4], b3v[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b1v[q] = p.b1[16 * cg + 4 * lg + q];
    b2v[q] = p.b2[16 * cg + 4 * lg + q];
#pragma unroll
    for (int i = 0; i < 4; ++i) b3v[i][q] = p.b3[64 * cg + 16 * i + 4 * lg + q];
  }
  // ---- the tile walk: tile += gridDim.x as (image, tile row, tile column) with carries (one division per launch, none per tile) ----
  const BnWalk walk = bn_walk(p.n_tiles);
  if (walk.count <= 0) return;  // (uniform, before any barrier)
  const int g_img = walk.stride / tiles_per_img, g_rem = walk.stride - g_img * tiles_per_img;
  const int g_ty = g_rem / p.tiles_x, g_tx = g_rem - g_ty * p.tiles_x;
  auto advance = [&](BnTile c) {
    c.tx += g_tx;
    const int kx = c.tx >= p.tiles_x ? 1 : 0;
    c.tx -= kx ? p.tiles_x : 0;
    c.ty += g_ty + kx;
    const int ky = c.ty >= p.tiles_y ? 1 : 0;
    c.ty -= ky ? p.tiles_y : 0;
    c.img += g_img + ky;
    return c;
  };
  // DMA piece k of this wavefront: halo rows (half*7 + k)*8 .. +7 of channel chunk cg; the lane's source 16 bytes are chunk
  // (lane & 7) ^ lrow of its row (the swizzle of the LDS image, applied on the source side: the DMA destination is lane-linear).
  // The lane's byte offset RELATIVE TO THE HALO'S FIRST PIXEL does not depend on the tile: seven registers for the whole launch, and a
  // tile whose halo lies inside the image (81 of 121 per frame at res 352) issues its pieces with that register as the vector offset
  // and the halo origin as the instruction's SCALAR offset - no vector arithmetic at all.  (Rounds 4 - 5 recomputed row / 10, the
  // offset and a validity select per piece and tile: ~25 instructions per piece, a third of the instructions a wavefront issued per
  // tile.)  Rows 100 .. 111 of the last pieces are padding: out-of-range offset, zero fill.
  uint32_t voff[NP];
  {
    const int lrow = lane0 >> 3;
    const uint32_t lane_part = (uint32_t)(cg * 128 + (((lane0 & 7) ^ lrow) << 4));
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int row = (half * NP + k) * 8 + lrow;
      const int hy = (row * 205) >> 11, hx = row - hy * HW;  // row / 10 for row < 1029
      voff[k] = row < 100 ? (uint32_t)(hy * p.W + hx) * (CIN * 2) + lane_part : TD_BN_OOB;
    }
  }
  // also per launch: the unit-matrix block of phase 3's identity MFMA (row (channel) lr, columns 8*lg .. + 7: 1.0 = 0x3F80 at column lr)
  // and the lane's part of an output store's offset (staging row lane >> 3 = pixel (tile row (lane >> 4) [+ 4 ps], column (lane >> 3) & 1 [+ 2 mb]))
  uint4 eye = make_uint4(0u, 0u, 0u, 0u);
  if ((lr >> 3) == lg) {
    const uint32_t one = 0x3F80u << (16 * (lr & 1));
    const int d2 = (lr & 7) >> 1;
    eye.x = d2 == 0 ? one : 0u; eye.y = d2 == 1 ? one : 0u; eye.z = d2 == 2 ? one : 0u; eye.w = d2 == 3 ? one : 0u;
  }
  const uint32_t st_lane = (uint32_t)(((lane0 >> 4) * p.W + ((lane0 >> 3) & 1)) * 512 + cg * 128 + (lane0 & 7) * 16);
  // the 7 pieces of tile c (past the end: out-of-range offsets, zero fill) into buffer `xbuf`
  auto issue_tile = [&](BnTile c, char* xbuf) {
    const int y0 = c.ty * TH, x0 = c.tx * TW;
    const bool exists = c.img < p.N;
    const bool interior = exists && y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;  // wave-uniform: the whole halo lies inside the image
    const uint32_t px = (uint32_t)((c.img * p.H + y0) * p.W + x0);
    char* dst = xbuf + cg * (HROWS * 128) + half * NP * 1024;
    if (interior) {
      // the halo's first pixel (>= 0 for an interior tile); readfirstlane: the compiler does not prove the walk uniform and would wrap every piece in a loop
      const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((px - (uint32_t)(p.W + 1)) * (CIN * 2)));
#pragma unroll
      for (int k = 0; k < NP; ++k) {
#if TD_BN_ABL & 1
        asm volatile("" ::"v"(voff[k]), "s"(soff));
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_p)(dst + k * 1024), 16, voff[k], soff, 0, 0);
#endif
      }
    } else {
      int ln = lane0;
      asm volatile("" : "+v"(ln));
      const int lrow = ln >> 3;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int row = (half * NP + k) * 8 + lrow;
        const int hy = (row * 205) >> 11, hx = row - hy * HW;
        // (modulo 2^32: where the halo lies before the buffer's first byte the piece is masked anyway)
        uint32_t off = (px - (uint32_t)(p.W + 1)) * (CIN * 2) + voff[k];
        const bool ok = row < 100 && exists && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
        off = ok ? off : TD_BN_OOB;
#if TD_BN_ABL & 1
        asm volatile("" ::"v"(off));
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_p)(dst + k * 1024), 16, off, 0, 0, 0);
#endif
      }
    }
  };

  // one tile, input in buffer B (compile-time: the two buffers are different LDS objects)
  auto process = [&](BnTile tc, BnTile tnext, auto B_) {
    constexpr int B = decltype(B_)::value;
    char* const xb_w = B ? xall1 : xall0;
    char* const xo_w = B ? xall0 : xall1;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    const int cyl = lr >> 1, cxl = lr & 1;  // centre pixel of row block mb, lane lr: (cyl, 2 * mb + cxl) - a block is two tile COLUMNS
    const int y0 = tc.ty * TH, x0 = tc.tx * TW;
    const uint32_t tile_px = (uint32_t)((tc.img * p.H + y0) * p.W + x0);
    const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;
    // the other buffer was last read in phase 3 of the previous tile, behind that tile's closing barrier
    issue_tile(tnext, xo_w);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");  // this tile's pieces (issued a tile ago) have landed; the next tile's stay in flight
    TD_BN_BARRIER();
    const char* const xb = xb_w;
    // ================= phase 1: conv1 on the halo tile: channels 16*cg .. +15, row blocks 4*half .. (4 / 3 of the 7) =================
    {
      const char* const xa0 = xb + lr * 128 + ((lg ^ (lr & 7)) << 4) + half * (4 * 2048);  // k-step parity 0
      const char* const xa1 = xb + lr * 128 + (((4 + lg) ^ (lr & 7)) << 4) + half * (4 * 2048);
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{b1v[0], b1v[1], b1v[2], b1v[3]};  // the bias seeds the accumulator
      // 8 k-steps x 4 row blocks, fragment reads double-buffered one k-step (4 reads, 4 MFMAs) ahead: with two reads per group the
      // ~100-cycle LDS latency of each group was exposed behind 32 cycles of MFMA issue
      uint4 fr[2][4];
      auto load_g = [&](int ks8, uint4 (&dst)[4]) {  // (the second half's 4th block is rows 112 .. 127 of the chunk = LDS bytes of the next chunk: read, never used)
        const char* base = ((ks8 & 1) ? xa1 : xa0) + (ks8 >> 1) * (HROWS * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *(const uint4*)(base + j * 2048);
      };
      load_g(0, fr[0]);
#pragma unroll
      for (int ks8 = 0; ks8 < 8; ++ks8) {
        if (ks8 + 1 < 8) load_g(ks8 + 1, fr[(ks8 + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w1r[ks8 >> 1][ks8 & 1], *(const bf16x8*)&fr[ks8 & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      char* const hw = h1 + (half * 64 + lr) * HP + (16 * cg + 4 * lg) * 2;
      uint2 o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bn_relu_pack4(acc[j]);
      if (!interior) {  // outside the image: conv2's zero padding (an interior tile has no such pixel; rows >= 100 are never read)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = (half * 4 + j) * 16 + lr;
          const int hy = (row * 205) >> 11, hx = row - hy * HW;
          const bool inside = (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
          o[j].x = inside ? o[j].x : 0u;
          o[j].y = inside ? o[j].y : 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) bn_lds_store8(hw + j * 16 * HP, o[j]);
      if (half == 0) bn_lds_store8(hw + 3 * 16 * HP, o[3]);  // (wave-uniform; the 8th block does not exist)
    }
    TD_BN_BARRIER();  // h1 complete
    // ================= phase 2: conv2 3x3: channels 16*cg .. +15, centre row blocks 2*half, 2*half + 1 =================
    {
      const char* const hr2 = h1 + (cyl * HW + 4 * half + cxl) * HP + lg * 16;  // + (r * HW + 2j + s) * HP + parity * 64
      f32x4 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = f32x4{b2v[0], b2v[1], b2v[2], b2v[3]};
      // 9 taps x (2 channel halves x 2 row blocks), fragment reads double-buffered one tap (4 reads, 4 MFMAs) ahead
      uint4 fr[2][4];
      auto load_t = [&](int tap, uint4 (&dst)[4]) {
        const int r = tap / 3, s_ = tap - 3 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 2; ++j) dst[h * 2 + j] = *(const uint4*)(hr2 + (r * HW + 2 * j + s_) * HP + h * 64);
      };
      load_t(0, fr[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_t(tap + 1, fr[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w2r[tap * 2 + h], *(const bf16x8*)&fr[tap & 1][h * 2 + j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      char* const hw = h2 + (half * 32 + lr) * HP + (16 * cg + 4 * lg) * 2;  // (last read in phase 3 of the previous tile, three barriers back)
#pragma unroll
      for (int j = 0; j < 2; ++j) bn_lds_store8(hw + j * 16 * HP, bn_relu_pack4(acc[j]));
    }
    TD_BN_BARRIER();  // h2 complete
    // ================= phase 3: conv3 + identity + ReLU: channels 64*cg .. +63, centre row blocks 2*half, 2*half + 1 =================
    {
      // The identity goes through the matrix pipe: one more MFMA per N fragment with a 16 x 32 "weight" block that is the unit
      // matrix in its first 16 columns - the activation operand is the pixel's 16 channels 64*cg + 16i .. of the input tile still
      // in LDS (lane group lg < 2 reads channels 16i + 8lg .. + 7, groups 2, 3 read the same bytes against zero weights).
      // bf16 x 1.0 accumulated in fp32 is exact, and it replaces an 8-byte LDS read + 10 VALU operations (unpack, add) per fragment.
      const int hp0 = (cyl + 1) * HW + 4 * half + cxl + 1;                       // halo row of the lane's centre pixel in its first row block (+ 2 for the second)
      const char* const idb = xb + cg * (HROWS * 128) + hp0 * 128;
      const uint32_t idk0 = (uint32_t)(((lg & 1) ^ (hp0 & 7)) << 4);             // 16-byte chunk (2i | lg & 1) ^ (row & 7) = 32i ^ idk
      const uint32_t idk1 = (uint32_t)(((lg & 1) ^ ((hp0 + 2) & 7)) << 4);
      const char* const hr3 = h2 + (half * 32 + lr) * HP + lg * 16;              // + j * 16 * HP + ks * 64
      char* const stw = ostage[wave] + lr * 128 + (lg & 1) * 8;
      const uint32_t stk = (uint32_t)(((lg >> 1) ^ (lr & 7)) << 4);
      const char* const strd = ostage[wave] + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);  // + ps * 1024
      const bool full = y0 + TH <= p.H && x0 + TW <= p.W;  // wave-uniform; ONE branch per tile selects the phase's code (no branch per store)
      auto phase3 = [&](auto FULL_) {
      constexpr bool FULL = decltype(FULL_)::value;
      // all twelve fragment reads of the phase first (the 24 MFMAs then wait by count, not for a round trip each)
      uint4 a[2][2], xi[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[j][ks] = *(const uint4*)(hr3 + j * 16 * HP + ks * 64);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xi[j][i] = *(const uint4*)(idb + j * (2 * 128) + ((j ? idk1 : idk0) ^ (uint32_t)(32 * i)));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int mb = 2 * half + j;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{b3v[i][0], b3v[i][1], b3v[i][2], b3v[i][3]};  // the bias seeds the accumulator
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w3r[i][ks], *(const bf16x8*)&a[j][ks], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&eye, *(const bf16x8*)&xi[j][i], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) bn_lds_store8(stw + (stk ^ (uint32_t)(32 * i)), bn_relu_pack4(acc[i]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 8 pixels x 128 bytes per store instruction: lane = (pixel, 16-byte chunk)
        const uint4 o16a = *(const uint4*)(strd), o16b = *(const uint4*)(strd + 1024);
#if TD_BN_ABL & 2
        asm volatile("" ::"v"(o16a.x), "v"(o16a.y), "v"(o16a.z), "v"(o16a.w), "v"(o16b.x), "v"(o16b.y), "v"(o16b.z), "v"(o16b.w));
#else
        if constexpr (FULL) {  // the lane's offset is per launch, tile and row block are the scalar offset
          const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)((tile_px + (uint32_t)(2 * mb)) * 512u));
          bn_store16s(rs_out, st_lane, so, o16a);
          bn_store16s(rs_out, st_lane, so + (uint32_t)(4 * p.W) * 512u, o16b);
        } else {
          const int st_y = lane >> 4, st_x = (lane >> 3) & 1;
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            uint32_t off = (tile_px + (uint32_t)(4 * ps * p.W + 2 * mb)) * 512u + st_lane;
            off = (y0 + 4 * ps + st_y < p.H && x0 + 2 * mb + st_x < p.W) ? off : TD_BN_OOB;
            bn_store16(rs_out, off, ps ? o16b : o16a);
          }
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next row block overwrites the region
      }
      };
      if (full) phase3(std::true_type{});
      else phase3(std::false_type{});
    }
    TD_BN_BARRIER();  // every wavefront is done with the tile: h1 and this input buffer may be overwritten
  };
  BnTile cur;
  cur.img = walk.start / tiles_per_img;
  {
    const int trem = walk.start - cur.img * tiles_per_img;
    cur.ty = trem / p.tiles_x;
    cur.tx = trem - cur.ty * p.tiles_x;
  }
  int left = walk.count;  // tiles of this workgroup from `cur` on; a tile past the last one is marked by img = N ("does not exist": zero-fill pieces)
  auto after = [&](BnTile c) {
    BnTile n = advance(c);
    if (left <= 1) n.img = p.N;
    return n;
  };
  BnTile nxt = after(cur);
  issue_tile(cur, xall0);
  for (;;) {
    process(cur, nxt, std::integral_constant<int, 0>{});
    cur = nxt;
    --left;
    nxt = after(cur);
    if (cur.img >= p.N) break;
    process(cur, nxt, std::integral_constant<int, 1>{});
    cur = nxt;
    --left;
    nxt = after(cur);
    if (cur.img >= p.N) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (out-of-range) pieces must not outlive the workgroup's LDS
}

// ------------------------------------------------------------------------------------------------
// Block 0 of layer1 (64 -> 64 -> 256 with the 1x1 downsample branch as its identity) in the form of bottleneck_resident3_kernel:
// one workgroup of eight wavefronts per CU, 8 x 16 output tiles, the 10 x 18 halo tile of the 64-channel input (23 KiB) DMA'd into
// one of two LDS buffers a tile ahead (3 pieces per wavefront), conv1 / conv2 results at a padded 144-byte pitch (fragment reads of
// phase 2 are lane base + immediate), ALL weight fragments and biases resident in registers (8 + 72 + 32 + 32 + 24), biases seeding
// the accumulators, wavefront w -> channel group w & 3 x row-block half w >> 2, no validity arithmetic for tiles whose halo lies
// inside the image.  (The round-3 form spent ~1 400 VALU instructions per wavefront and tile on element index arithmetic, register
// staging, per-read swizzles and per-tile weight reloads: 2.09 ms per 1 000 frames; this one 1.60.)
__global__ __launch_bounds__(512, 2) void bottleneck_first3_kernel(BneckParams p) {
  constexpr int CIN = 64;
  constexpr int TH = 8, TW = 16, HW = TW + 2;          // halo 10 x 18 = 180 pixels
  constexpr int NHALO = (TH + 2) * HW, HROWS = 192;     // -> 12 blocks of 16 rows
  constexpr int NP = 3;                                 // DMA pieces per wavefront and tile: row groups 3w .. 3w + 2 (8 rows x 128 B each)
  constexpr int HP = 160;                               // row pitch of h1 / h2 (conflict-free ds_read_b128 of 16 consecutive rows; 144 is two-way)
  constexpr int XBUF = HROWS * 128;
  __shared__ __attribute__((aligned(16))) char x0buf[XBUF];  // (two LDS objects: see bottleneck_resident3_kernel)
  __shared__ __attribute__((aligned(16))) char x1buf[XBUF];
  __shared__ __attribute__((aligned(16))) char h1[HROWS * HP];
  __shared__ __attribute__((aligned(16))) char h2[TH * TW * HP];  // (its own 20 KiB: no barrier between the last h1 fragment read and the h2 stores)
  __shared__ __attribute__((aligned(16))) char ostage[8][TD_BN_STAGE_BYTES];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cg = wave & 3, half = wave >> 2;
  const int lane0 = t & 63;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const uint32_t x_bytes = (uint32_t)((size_t)p.N * p.H * p.W * CIN * 2);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (uint32_t)((size_t)p.N * p.H * p.W * 512), 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_p;

  // ---- resident for the whole launch: every weight fragment and bias (lane-derived addresses are recomputed per tile) ----
  const int lr = lane0 & 15, lg = lane0 >> 4;
  uint4 w1r[2], w2r[18], w3r[4][2], wdr[4][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) w1r[ks] = *(const uint4*)(p.w1 + ((size_t)(16 * cg + lr) * CIN + ks * 32 + lg * 8) * 2);
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) w2r[ks] = *(const uint4*)(p.w2 + ((size_t)(16 * cg + lr) * 576 + ks * 32 + lg * 8) * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      w3r[i][ks] = *(const uint4*)(p.w3 + ((size_t)(64 * cg + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
      wdr[i][ks] = *(const uint4*)(p.wd + ((size_t)(64 * cg + 16 * i + lr) * 64 + ks * 32 + lg * 8) * 2);
    }
  float b1v[4], b2v[4], b3v[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b1v[q] = p.b1[16 * cg + 4 * lg + q];
    b2v[q] = p.b2[16 * cg + 4 * lg + q];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 64 * cg + 16 * i + 4 * lg + q;
      b3v[i][q] = p.b3[n] + p.bd[n];
    }
  }

  // the tile walk and the per-launch piece offsets: see bottleneck_resident3_kernel
  const BnWalk walk = bn_walk(p.n_tiles);
  if (walk.count <= 0) return;  // (uniform, before any barrier)
  const int g_img = walk.stride / tiles_per_img, g_rem = walk.stride - g_img * tiles_per_img;
  const int g_ty = g_rem / p.tiles_x, g_tx = g_rem - g_ty * p.tiles_x;
  auto advance = [&](BnTile c) {
    c.tx += g_tx;
    const int kx = c.tx >= p.tiles_x ? 1 : 0;
    c.tx -= kx ? p.tiles_x : 0;
    c.ty += g_ty + kx;
    const int ky = c.ty >= p.tiles_y ? 1 : 0;
    c.ty -= ky ? p.tiles_y : 0;
    c.img += g_img + ky;
    return c;
  };
  uint32_t voff[NP];
  {
    const int lrow = lane0 >> 3;
    const uint32_t lane_part = (uint32_t)(((lane0 & 7) ^ lrow) << 4);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int row = (wave * NP + k) * 8 + lrow;
      const int hy = (row * 57) >> 10, hx = row - hy * HW;  // row / 18 for row < 1024
      voff[k] = row < NHALO ? (uint32_t)(hy * p.W + hx) * (CIN * 2) + lane_part : TD_BN_OOB;
    }
  }
  const uint32_t st_lane = (uint32_t)((lane0 >> 3) * 512 + cg * 128 + (lane0 & 7) * 16);  // the lane's part of an output store's offset
  auto issue_tile = [&](BnTile c, char* xbuf) {
    const int y0 = c.ty * TH, x0 = c.tx * TW;
    const bool exists = c.img < p.N;
    const bool interior = exists && y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;
    const uint32_t px = (uint32_t)((c.img * p.H + y0) * p.W + x0);
    if (interior) {
      const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((px - (uint32_t)(p.W + 1)) * (CIN * 2)));
#pragma unroll
      for (int k = 0; k < NP; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_p)(xbuf + (wave * NP + k) * 1024), 16, voff[k], soff, 0, 0);
    } else {
      int ln = lane0;
      asm volatile("" : "+v"(ln));
      const int lrow = ln >> 3;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int row = (wave * NP + k) * 8 + lrow;
        const int hy = (row * 57) >> 10, hx = row - hy * HW;
        uint32_t off = (px - (uint32_t)(p.W + 1)) * (CIN * 2) + voff[k];
        const bool ok = row < NHALO && exists && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
        off = ok ? off : TD_BN_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_p)(xbuf + (wave * NP + k) * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  auto process = [&](BnTile tc, BnTile tnext, auto B_) {
    constexpr int B = decltype(B_)::value;
    char* const xb_w = B ? x1buf : x0buf;
    char* const xo_w = B ? x0buf : x1buf;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    const int y0 = tc.ty * TH, x0 = tc.tx * TW;
    const uint32_t tile_px = (uint32_t)((tc.img * p.H + y0) * p.W + x0);
    const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= p.H && x0 + TW + 1 <= p.W;
    issue_tile(tnext, xo_w);  // (that buffer was last read in phase 3 of the previous tile, behind its closing barrier)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");  // this tile's pieces have landed; the next tile's stay in flight
    TD_BN_BARRIER();
    const char* const xb = xb_w;
    // ================= phase 1: conv1 (64 -> 64) on the halo tile: channels 16*cg .. +15, row blocks 6*half .. + 5 =================
    {
      const char* const xa0 = xb + (half * 96 + lr) * 128 + ((lg ^ (lr & 7)) << 4);
      const char* const xa1 = xb + (half * 96 + lr) * 128 + (((4 + lg) ^ (lr & 7)) << 4);
      f32x4 acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] = f32x4{b1v[0], b1v[1], b1v[2], b1v[3]};
      uint4 fr[2][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        fr[0][j] = *(const uint4*)(xa0 + j * 2048);
        fr[1][j] = *(const uint4*)(xa1 + j * 2048);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 6; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w1r[ks], *(const bf16x8*)&fr[ks][j], acc[j], 0, 0, 0);
      char* const hw = h1 + (half * 96 + lr) * HP + (16 * cg + 4 * lg) * 2;
      uint2 o[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = bn_relu_pack4(acc[j]);
      if (!interior) {  // outside the image: conv2's zero padding (rows >= 180 are never read)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int row = (half * 6 + j) * 16 + lr;
          const int hy = (row * 57) >> 10, hx = row - hy * HW;
          const bool inside = (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
          o[j].x = inside ? o[j].x : 0u;
          o[j].y = inside ? o[j].y : 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) bn_lds_store8(hw + j * 16 * HP, o[j]);
    }
    TD_BN_BARRIER();  // h1 complete
    // ================= phase 2: conv2 3x3: channels 16*cg .. +15, tile rows 4*half .. + 3 (row block = one tile row of 16 pixels) =================
    {
      const char* const hr2 = h1 + ((4 * half) * HW + lr) * HP + lg * 16;  // centre (row j, column lr) -> halo pixel (j + r, lr + s): + ((j + r) * HW + s) * HP + parity * 64
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{b2v[0], b2v[1], b2v[2], b2v[3]};
      uint4 fr[2][4];
      auto load_k = [&](int ks, uint4 (&dst)[4]) {
        const int tap = ks >> 1, r = tap / 3, s_ = tap - 3 * r;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *(const uint4*)(hr2 + ((j + r) * HW + s_) * HP + (ks & 1) * 64);
      };
      load_k(0, fr[0]);
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + 1 < 18) load_k(ks + 1, fr[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w2r[ks], *(const bf16x8*)&fr[ks & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      char* const hw = h2 + (half * 64 + lr) * HP + (16 * cg + 4 * lg) * 2;  // (last read in phase 3 of the previous tile, three barriers back)
#pragma unroll
      for (int j = 0; j < 4; ++j) bn_lds_store8(hw + j * 16 * HP, bn_relu_pack4(acc[j]));
    }
    TD_BN_BARRIER();  // h2 complete
    // ================= phase 3: conv3 + downsample(x) + ReLU: channels 64*cg .. +63, tile rows 4*half .. + 3 =================
    {
      const char* const hr3 = h2 + (half * 64 + lr) * HP + lg * 16;  // + j * 16 * HP + ks * 64
      char* const stw = ostage[wave] + lr * 128 + (lg & 1) * 8;
      const uint32_t stk = (uint32_t)(((lg >> 1) ^ (lr & 7)) << 4);
      const char* const strd = ostage[wave] + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);  // + ps * 1024
      const bool full = y0 + TH <= p.H && x0 + TW <= p.W;  // wave-uniform; ONE branch per tile selects the phase's code (no branch per store)
      auto phase3 = [&](auto FULL_) {
      constexpr bool FULL = decltype(FULL_)::value;
      // the four fragment reads of a tile row in front of its 16 MFMAs (a row ahead they cost 16 more registers than the kernel has)
      uint4 fa[1][2], fx[1][2];
      auto load_row = [&](int j, uint4 (&a)[2], uint4 (&xd)[2]) {
        const int hp = (4 * half + j + 1) * HW + lr + 1;  // this pixel in the input halo tile (the downsample branch reads it)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          a[ks] = *(const uint4*)(hr3 + j * 16 * HP + ks * 64);
          xd[ks] = *(const uint4*)(xb + hp * 128 + (((ks * 4 + lg) ^ (hp & 7)) << 4));
        }
      };
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mb = 4 * half + j;  // tile row
        load_row(j, fa[0], fx[0]);
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{b3v[i][0], b3v[i][1], b3v[i][2], b3v[i][3]};  // b3 + bd seed the accumulator
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w3r[i][ks], *(const bf16x8*)&fa[0][ks], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&wdr[i][ks], *(const bf16x8*)&fx[0][ks], acc[i], 0, 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bn_lds_store8(stw + (stk ^ (uint32_t)(32 * i)), bn_relu_pack4(acc[i]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 8 pixels x 128 bytes per store instruction: lane = (pixel, 16-byte chunk)
        const uint4 o16a = *(const uint4*)(strd), o16b = *(const uint4*)(strd + 1024);
        if constexpr (FULL) {  // the lane's offset is per launch, tile and tile row are the scalar offset
          const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)((tile_px + (uint32_t)(mb * p.W)) * 512u));
          bn_store16s(rs_out, st_lane, so, o16a);
          bn_store16s(rs_out, st_lane, so + 8u * 512u, o16b);
        } else {
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            uint32_t off = (tile_px + (uint32_t)(mb * p.W + ps * 8)) * 512u + st_lane;
            off = (y0 + mb < p.H && x0 + ps * 8 + (lane >> 3) < p.W) ? off : TD_BN_OOB;
            bn_store16(rs_out, off, ps ? o16b : o16a);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next row overwrites the region
      }
      };
      if (full) phase3(std::true_type{});
      else phase3(std::false_type{});
    }
    TD_BN_BARRIER();  // every wavefront is done with the tile: h1 and this input buffer may be overwritten
  };
  BnTile cur;
  cur.img = walk.start / tiles_per_img;
  {
    const int trem = walk.start - cur.img * tiles_per_img;
    cur.ty = trem / p.tiles_x;
    cur.tx = trem - cur.ty * p.tiles_x;
  }
  int left = walk.count;  // tiles of this workgroup from `cur` on; a tile past the last one is marked by img = N ("does not exist": zero-fill pieces)
  auto after = [&](BnTile c) {
    BnTile n = advance(c);
    if (left <= 1) n.img = p.N;
    return n;
  };
  BnTile nxt = after(cur);
  issue_tile(cur, x0buf);
  for (;;) {
    process(cur, nxt, std::integral_constant<int, 0>{});
    cur = nxt;
    --left;
    nxt = after(cur);
    if (cur.img >= p.N) break;
    process(cur, nxt, std::integral_constant<int, 1>{});
    cur = nxt;
    --left;
    nxt = after(cur);
    if (cur.img >= p.N) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (out-of-range) pieces must not outlive the workgroup's LDS
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
  const bool res256 = Cin == 256;  // 8 x 8 tiles, whole input tile in LDS (bottleneck_resident3_kernel); block 0: 8 x 16
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
  const int grid = (int)std::min<long long>(nt, (long long)n_cu);  // one persistent workgroup of eight wavefronts per CU
  if (Cin == 64) bottleneck_first3_kernel<<<grid, 512, 0, st>>>(p);
  else bottleneck_resident3_kernel<<<grid, 512, 0, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_bottleneck_fused");
}
