// Implicit-GEMM convolution / linear kernels for gfx950 (MI355X).
//
//   conv_gemm :  out[m][n] = epi( sum_k gather(src)[m][k] * W[n][k] )      forward (mode 0) and dgrad (mode 1)
//   conv_wgrad:  dW[n][k] += sum_m g[m][n] * gather(src)[m][k]             split over m, fp32 atomics
//
// Layout: activations NHWC (rows = (n,ho,wo), channels contiguous), weights [Cout][R*S*Cin] with K
// contiguous, so both MFMA operands are read from LDS as 16-byte K-contiguous fragments.  One source,
// two element types: TD_BF16 -> v_mfma_f32_16x16x32_bf16, TD_F32 -> v_mfma_f32_16x16x4_f32 (exact fp32,
// the parity mode).  Tiles are described in BYTES along K (128 B per row per stage) so the load path is
// identical for both types.  MFMA roles are swapped (A = weights, B = activations) so every lane ends up
// with 4 consecutive output channels of one output row -> 8/16-byte NHWC stores.
//
// Replaces the torch Conv2d/Linear (+FrozenBatchNorm2d/ReLU/residual) calls of the reference hot path:
// models/backbone.py:60-70,97-98 (torchvision resnet101 body), models/tubedetr.py:80,131,134 (input_proj),
// models/transformer.py:124-125,387,441-445,613-617,643,661-667,748,764-773 and models/tubedetr.py:37-42.
#include <stdlib.h>

#include <type_traits>

#include "td_common.h"
#include <algorithm>
#include <vector>

#ifndef TD_WGRAD_EARLY_ISSUE
#define TD_WGRAD_EARLY_ISSUE 1  // wide weight gradients: a stage buffer is refilled as soon as it is free, one whole stage ahead of its wait (0: half a stage; A/B builds)
#endif
#ifndef TD_BIG_XCD_CONTIG
#define TD_BIG_XCD_CONTIG 1  // conv_gemm_big8_kernel: an XCD walks a contiguous range of row tiles (0: row tile = 8 * seq + XCD; A/B builds)
#endif
namespace td {

struct GemmParams {
  const char* src;
  const char* w;
  char* out;
  td_conv_desc d;
  int M, K;
  uint32_t src_bytes, w_bytes;
  unsigned long long* dbg;  // optional per-workgroup cycle stamps (tools/stamp_conv.py)
  const float* bias;
  const char* residual;
  const char* mask_src;
  int relu, sigmoid;
  float alpha;
  uint32_t drop_thresh;
  float drop_scale;
  uint32_t seed;
  const uint32_t* seed_dev;
  // td_linear_ex (GX instances): the activation operand is [ A1[a1_map[m]] (K1 columns) | A2[a2_map[m]] (K - K1 columns) ]
  const char* src2;
  uint32_t src2_bytes;
  int K1, lda1, lda2, ldr;
  int w_shared;  // 1: wmat is [N][K1] and multiplies BOTH sources (= (A1 + A2) W^T without a [W | W] copy)
  const int* a1_map;
  const int* a2_map;
  const int* out_map;
  const int* res_map;
  int tap_inner;  // 256-row tile kernel, spatial layers: walk the K axis channel-chunk-major (all taps of one 64-channel chunk, then the next chunk)
  int w_ld;       // row stride of wmat in elements when it is not K (a launch that uses a subset of the columns of a larger matrix); 0 = K
  int wtap[4];    // tap-uniform instances with use_wtap: tap t of this launch reads weight columns wtap[t] * C .. (the tap's block in the larger matrix)
  int use_wtap;
};

template <typename T>
struct Mfma;
template <>
struct Mfma<u16> {
  static __device__ __forceinline__ void run(const uint4& wf, const uint4& af, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&wf, *(const bf16x8*)&af, acc, 0, 0, 0);
  }
};
template <>
struct Mfma<float> {
  // 16 fp32 of K per lane-group row chunk: lane (i, g) holds k = 4g..4g+3; MFMA step s consumes element s
  // of both operands (same k permutation on both sides, the dot product is unchanged).
  static __device__ __forceinline__ void run(const uint4& wf, const uint4& af, f32x4& acc) {
    const float* a = (const float*)&wf;
    const float* b = (const float*)&af;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ void load4(const char* base, size_t off, float (&o)[4]);
template <>
__device__ __forceinline__ void load4<float>(const char* base, size_t off, float (&o)[4]) {
  float4 v = *(const float4*)(base + off * 4);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <>
__device__ __forceinline__ void load4<u16>(const char* base, size_t off, float (&o)[4]) {
  uint2 v = *(const uint2*)(base + off * 2);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T, typename V>
__device__ __forceinline__ void unpack4(const V& v, float (&o)[4]);
template <>
__device__ __forceinline__ void unpack4<float, float4>(const float4& v, float (&o)[4]) {
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <>
__device__ __forceinline__ void unpack4<u16, uint2>(const uint2& v, float (&o)[4]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
// 16 bytes of T <-> floats (8 bf16 or 4 fp32); bf16 packing uses the gfx950 v_cvt_pk_bf16_f32 (round-to-nearest-even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return *(uint32_t*)&v;
}
template <typename T>
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[16 / sizeof(T)]);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& v, float (&o)[4]) {
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void unpack16<u16>(const uint4& v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ uint4 pack16(const float (&v)[16 / sizeof(T)]);
template <>
__device__ __forceinline__ uint4 pack16<float>(const float (&v)[4]) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <>
__device__ __forceinline__ uint4 pack16<u16>(const float (&v)[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
template <typename T>
__device__ __forceinline__ void store4(char* base, size_t off, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(char* base, size_t off, const float (&v)[4]) {
  *(float4*)(base + off * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<u16>(char* base, size_t off, const float (&v)[4]) {
  uint2 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *(uint2*)(base + off * 2) = o;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// 16-byte epilogue operand loads / result stores, non-temporal (TD_NT bit 0: loads, bit 1: stores; 0 = plain, for A/B builds):
// these tensors are 0.1 - 1 GB streams that are touched once per launch; measured on the K >= 512 pointwise launches
// -12 .. -18 %, on the persistent 1x1 instance 5.05 -> 5.41 TB/s.
#ifndef TD_NT
#define TD_NT 3
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld16(const char* p) {
#if TD_NT & 1
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *(const uint4*)p;
#endif
}
__device__ __forceinline__ void st16(char* p, const uint4& v) {
#if TD_NT & 2
  __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, (u32x4_t*)p);
#else
  *(uint4*)p = v;
#endif
}

// Main loop: operands go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB = 8 rows x 128 B per wave
// instruction, no VGPR staging).  The LDS image is row-major with the 16-byte chunk index XOR-swizzled by
// (row & 7): the DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address and
// again on the fragment read.  Out-of-image taps, rows >= M, channels >= Nc and the K tail are given an
// out-of-range buffer offset, which the hardware returns as zeros - no divergent control flow in the loop.
//
// NST = 2: two stages, one tile in flight (compute-bound shapes).  NST = 3: three stages and COUNTED waits - at each
// step only the oldest tile is waited for (s_waitcnt vmcnt(#DMA per tile) + raw s_barrier), the next one stays in
// flight across the barrier, and the residual / mask operands of the epilogue are requested before the K loop - for
// the short-K, HBM-bound 1x1 layers whose per-workgroup latency chain would otherwise be 5-6 exposed round trips.
// PW = pointwise (1x1, stride 1, no padding, dense output rows): the gather degenerates to row m at offset m*K, so all
// the (n,ho,wo)/(r,s,c) index arithmetic is compiled out.
// TU = tap-uniform tiles: a spatial conv whose channel count is a multiple of the K tile (C % BK == 0; every 3x3 of the
// trunk) - all 8 chunks of a tile row then belong to ONE filter tap, so the tap, its validity and its pixel offset are
// wave-uniform scalars that advance once per tile, and a DMA address is (row base + tap offset) * C + channel: a shift, an
// and, a multiply-add and a select per instruction instead of ~20 VALU operations of per-lane (r, s, c) decomposition and
// bounds tests.  Which taps fall inside the image is a per-row bit mask computed once per workgroup.  Forward geometry with
// any stride, or input-gradient geometry with stride 1.
// GX = gather-extended pointwise operand (td_linear_ex): the K axis is the concatenation of TWO row-major sources - columns
// < K1 from A1, the rest from A2 (K1 a multiple of the K tile: a tile never straddles the seam, the choice is wave-uniform) -
// each read through an optional row index (a1_map / a2_map), the result row and the residual row through out_map / res_map.
// That is how "x + pos" enters the Q/K projections without being formed ([x | pos] [W | W]^T = (x + pos) W^T,
// models/transformer.py:637-640, 735-737) and how the temporal replication (:393-427) is an index inside its consumers.
template <typename T, int BM, int BN, int NST, bool PW, bool TU = false, bool GX = false>
// (three 24-KiB stages of a 64 x 128 tile = 72 KiB: two workgroups per CU, so that is what the NST == 3 instances declare)
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 || NST == 3) ? 2 : 3) void conv_gemm_kernel(GemmParams p) {
  static_assert(!(PW && TU), "pointwise layers have no taps");
  static_assert(!GX || (PW && NST == 2), "the gather-extended operand is a pointwise, two-stage instance");
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;   // elements per 16-byte chunk
  constexpr int BK = 128 / ES;   // K elements per tile (128 bytes per row)
  constexpr int AI = BM / 32, BI = BN / 32;   // 8-row groups per wave
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  // two distinct LDS objects (one per pipeline stage): LDS lowering then tags their accesses with disjoint alias
  // scopes, so fragment reads of one stage do not wait (vmcnt) for the DMA that is filling the other stage.
  __shared__ __attribute__((aligned(16))) char smem0[(BM + BN) * 128];
  __shared__ __attribute__((aligned(16))) char smem1[(BM + BN) * 128];
  __shared__ __attribute__((aligned(16))) char smem2[NST == 3 ? (BM + BN) * 128 : 16];

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  // 1-D grid; consecutive workgroup ids go round-robin over the 8 XCDs (one L2 each), so M tiles are dealt to XCDs
  // and every XCD walks all N tiles of its M tile back to back: the activation tile is re-read from that XCD's L2.
  const int NT = (d.Nc + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mt = (seq / NT) * 8 + xcd, nt = seq - (seq / NT) * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= p.M) return;
  const int lrow = lane >> 3;                 // row inside an 8-row group
  const int chunk = (lane & 7) ^ lrow;        // source chunk that lands in LDS slot (lane & 7) of that row
  const int HoWo = d.Ho * d.Wo;

  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_src2 = __builtin_amdgcn_make_buffer_rsrc((void*)(GX ? p.src2 : p.src), 0, GX ? p.src2_bytes : p.src_bytes, 0x00020000);

  // per-lane row bookkeeping (fixed for the whole K loop)
  int a_img[AI], a_hb[AI], a_wb[AI];
  bool a_ok[AI];
  uint32_t a_off[AI];
  uint32_t a_off2[GX ? AI : 1];  // GX: byte offset of this lane's row in the second source
  int a_base[AI];        // TU: pixel index of tap (0, 0) of this row (may be negative: only used for taps inside the image)
  uint32_t a_mask[AI];   // TU: bit (r*S + s) set = tap (r, s) of this row lies inside the image
  const int RS = d.R * d.S;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + (i * 4 + wave) * 8 + lrow;
    a_ok[i] = m < p.M;
    if constexpr (GX) {
      const int mm = a_ok[i] ? m : 0;
      const int r1 = p.a1_map ? p.a1_map[mm] : mm;
      const int r2 = p.a2_map ? p.a2_map[mm] : mm;
      a_off[i] = a_ok[i] ? (uint32_t)r1 * (uint32_t)p.lda1 * ES : OOB;
      a_off2[i] = (a_ok[i] && p.src2) ? (uint32_t)r2 * (uint32_t)p.lda2 * ES : OOB;
    } else if constexpr (PW) {
      a_off[i] = a_ok[i] ? (uint32_t)m * (uint32_t)p.K * ES : OOB;
    } else {
      int mm = a_ok[i] ? m : 0;
      int n = mm / HoWo;
      int rem = mm - n * HoWo;
      int ho = rem / d.Wo, wo = rem - ho * d.Wo;
      a_img[i] = n * d.Hs * d.Ws;
      if (d.mode == 0) {
        a_hb[i] = ho * d.stride - d.pad;
        a_wb[i] = d.aniso ? wo * d.stride_w - d.pad_w : wo * d.stride - d.pad;  // (aniso: forward geometry of the generic path only)
      } else {
        a_hb[i] = ho + d.pad;
        a_wb[i] = wo + d.pad;
      }
      if constexpr (TU) {
        a_base[i] = a_img[i] + a_hb[i] * d.Ws + a_wb[i];
        uint32_t msk = 0;
        for (int r = 0; r < d.R; ++r) {
          const int hs = d.mode == 0 ? a_hb[i] + r : a_hb[i] - r;
          for (int sx = 0; sx < d.S; ++sx) {
            const int ws = d.mode == 0 ? a_wb[i] + sx : a_wb[i] - sx;
            const bool in = (unsigned)hs < (unsigned)d.Hs && (unsigned)ws < (unsigned)d.Ws;
            msk |= (in ? 1u : 0u) << (r * d.S + sx);
          }
        }
        a_mask[i] = a_ok[i] ? msk : 0u;
      }
    }
  }
  uint32_t b_off[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + (i * 4 + wave) * 8 + lrow;
    b_off[i] = n < d.Nc ? (uint32_t)n * (uint32_t)(p.w_ld ? p.w_ld : ((GX && p.w_shared) ? p.K1 : p.K)) * ES : OOB;
  }
  // running decomposition of this lane's k index into (r, s, c)
  int kk = chunk * VEC;
  int kr = 0, ks_ = 0, kc = kk;
  if constexpr (!PW && !TU) {
    if (d.R * d.S > 1) {
      int tap = kk / d.C;
      kc = kk - tap * d.C;
      kr = tap / d.S;
      ks_ = tap - kr * d.S;
    }
  }
  // TU: wave-uniform tile position (tap, channel base, pixel offset of the tap); the lane only contributes its chunk
  int t_tap = 0, t_ks = 0, t_kc = 0, t_pix = 0;
  int k_tile0 = 0;  // first column of the tile being issued (wave-uniform; GX picks the source by it)
  const int lane_c = chunk * VEC;

  auto issue_tile = [&](char* stage) {
    char* stA = stage;
    char* stB = stage + BM * 128;
    const bool kvalid = TU ? (t_tap < RS) : (kk < p.K);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      uint32_t off;
      if constexpr (TU) {
        const bool ok = kvalid && ((a_mask[i] >> (t_tap & 31)) & 1u);
        off = ok ? (uint32_t)((a_base[i] + t_pix) * d.C + t_kc + lane_c) * ES : OOB;
      } else if constexpr (GX) {
        // k_tile0 (wave-uniform) is the tile's first column: below K1 the whole tile comes from A1, else from A2
        if (k_tile0 >= p.K1) {
          off = (kvalid && a_off2[i] != OOB) ? a_off2[i] + (uint32_t)(kk - p.K1) * ES : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src2, (lds_ptr_t)(stA + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
          continue;
        }
        off = (kvalid && a_off[i] != OOB) ? a_off[i] + (uint32_t)kk * ES : OOB;
      } else if constexpr (PW) {
        off = (kvalid && a_ok[i]) ? a_off[i] + (uint32_t)kk * ES : OOB;
      } else {
        int hs, ws;
        bool ok = a_ok[i] && kvalid;
        if (d.mode == 0) {
          hs = a_hb[i] + kr;
          ws = a_wb[i] + ks_;
        } else {
          int th = a_hb[i] - kr, tw = a_wb[i] - ks_;
          ok = ok && th >= 0 && tw >= 0;
          if (d.stride == 1) {
            hs = th; ws = tw;
          } else if (d.stride == 2) {
            ok = ok && (((th | tw) & 1) == 0);
            hs = th >> 1; ws = tw >> 1;
          } else {
            ok = ok && (th % d.stride == 0) && (tw % d.stride == 0);
            hs = th / d.stride; ws = tw / d.stride;
          }
        }
        ok = ok && (unsigned)hs < (unsigned)d.Hs && (unsigned)ws < (unsigned)d.Ws;
        off = ok ? ((uint32_t)(a_img[i] + hs * d.Ws + ws) * (uint32_t)d.C + (uint32_t)kc) * ES : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(stA + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
    int kw = (GX && p.w_shared && k_tile0 >= p.K1) ? kk - p.K1 : kk;  // shared weight: the second source re-reads the same columns
    if constexpr (TU) {
      if (p.use_wtap) kw = p.wtap[t_tap & 3] * d.C + t_kc + lane_c;  // the tap's column block in the larger matrix (wave-uniform choice)
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      uint32_t off = (kvalid && b_off[i] != OOB) ? b_off[i] + (uint32_t)kw * ES : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(stB + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
    // advance this lane's k by one tile
    kk += BK;
    k_tile0 += BK;
    if constexpr (TU) {
      t_kc += BK;
      if (t_kc >= d.C) {  // next tap (C is a multiple of BK: a tile never straddles two taps)
        t_kc = 0;
        ++t_tap;
        if (++t_ks == d.S) { t_ks = 0; t_pix += d.mode == 0 ? d.Ws - (d.S - 1) : -(d.Ws - (d.S - 1)); }
        else t_pix += d.mode == 0 ? 1 : -1;
      }
    } else if constexpr (!PW) {
      kc += BK;
      if (d.R * d.S > 1) {
        while (kc >= d.C) {
          kc -= d.C;
          if (++ks_ == d.S) { ks_ = 0; ++kr; }
        }
      }
    }
  };

  const int wy = wave >> 1, wx = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute_tile = [&](const char* stage) {
    const char* stA = stage;
    const char* stB = stage + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cidx = ks * 4 + lg;
      uint4 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        int row = wy * WM + j * 16 + lr;
        af[j] = *(const uint4*)(stA + row * 128 + ((cidx ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int row = wx * WN + i * 16 + lr;
        wf[i] = *(const uint4*)(stB + row * 128 + ((cidx ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mfma<T>::run(wf[i], af[j], acc[i][j]);
    }
  };

  // ---- epilogue operands (defined before the K loop so NST == 3 can request them early) ----
  // The accumulators (4 consecutive channels of 16 different rows per lane) are transposed through LDS - each wave
  // owns a [WM][WN] fp32 region of the idle stage buffers, 16-byte chunks XOR-swizzled by the row - so that every
  // lane then owns 16 bytes of ONE output row (8 bf16 / 4 fp32 channels): residual / mask loads and the output
  // stores are dwordx4 over whole contiguous row segments (stores are issue-bound per instruction, not per byte).
  constexpr int CPRW = WN / 4;          // 16-byte fp32 chunks per staged row
  constexpr int EPL = 16 / ES;          // output elements per lane per row: 8 (bf16) / 4 (fp32)
  constexpr int LPR = WN / EPL;         // lanes per output row segment
  constexpr int RPI = 64 / LPR;         // rows handled per wave instruction
  constexpr int NIT = WM / RPI;
  static_assert(2 * WM * WN * 4 <= (BM + BN) * 128, "staging region does not fit the stage buffer");
  const int cc = lane % LPR, rsub = lane / LPR;
  const int n = n0 + wx * WN + cc * EPL;
  const bool vec_ok = ((d.ldc % EPL) == 0) && (!GX || (p.ldr % EPL) == 0) && (n + EPL - 1 < d.Nc);
  size_t offs[NIT];
  size_t roffs[GX ? NIT : 1];  // GX: element offset of the residual row (its own row map and row stride)
  bool live[NIT];
  uint4 res[NIT], msk[NIT];
  // every residual / mask operand of this lane is requested in one go - 2*NIT independent 16-byte loads in flight
  // instead of a load->store chain (the output may alias the residual, so the compiler cannot hoist them itself)
  auto fetch_epilogue_operands = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wy * WM + it * RPI + rsub;
      live[it] = (m < p.M) && (n < d.Nc);
      size_t orow = live[it] ? m : 0;
      if (!PW && d.out_sp > 1) {
        int ni = (int)orow / HoWo;
        int rem = (int)orow - ni * HoWo;
        int ho = rem / d.Wo, wo = rem - ho * d.Wo;
        orow = ((size_t)ni * d.out_H + (size_t)ho * d.out_sp) * d.out_W + (size_t)wo * d.out_sp;
      }
      if constexpr (GX) {
        const size_t mm = orow;
        if (p.out_map) orow = (size_t)p.out_map[mm];
        roffs[it] = (p.res_map ? (size_t)p.res_map[mm] : orow) * (size_t)p.ldr + n;
      }
      offs[it] = orow * d.ldc + n;
      if (vec_ok && live[it]) {
        if (p.residual) res[it] = ld16(p.residual + (GX ? roffs[it] : offs[it]) * ES);
        if (p.mask_src) msk[it] = ld16(p.mask_src + offs[it] * ES);
      }
    }
  };

  const int nk = (p.K + BK - 1) / BK;
#define TD_STAMP(i) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
  TD_STAMP(0);
  if constexpr (NST == 2) {
    issue_tile(smem0);
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();  // tile kt has landed in stage 0 (vmcnt(0) + barrier); all waves are done reading stage 1
      if (kt == 0) TD_STAMP(1);
      if (kt + 1 < nk) issue_tile(smem1);
      compute_tile(smem0);
      if (kt + 1 >= nk) break;
      __syncthreads();
      if (kt + 2 < nk) issue_tile(smem0);
      compute_tile(smem1);
    }
    TD_STAMP(2);
    fetch_epilogue_operands();
  } else {
    // three stages, two tiles in flight.  Every step issues exactly one tile (k tiles past the end are all-OOB = zero
    // fill, no traffic) so "tile j has landed" is always "at most AI+BI DMA instructions of this wave outstanding"; the
    // pipeline warm-up is folded into the loop (steps -2, -1 only issue) so that each stage buffer has a single static
    // DMA site and the compiler's own LDS-DMA wait counts stay exact.
#define TD_CG_STEP(cur, nxt, j)                                                     \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");                    \
  __builtin_amdgcn_s_barrier();                                                     \
  issue_tile(nxt);                                                                  \
  if ((j) >= 0) compute_tile(cur);
    for (int it = -2; it < nk; it += 3) {
      TD_CG_STEP(smem1, smem0, it)
      if (it + 1 >= nk) break;
      TD_CG_STEP(smem2, smem1, it + 1)
      if (it + 2 >= nk) break;
      TD_CG_STEP(smem0, smem2, it + 2)
    }
#undef TD_CG_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill DMAs must not land in the epilogue staging area
    TD_STAMP(2);
    fetch_epilogue_operands();
  }

  // (2) transpose the accumulators through LDS.  Raw barrier: the operand loads above stay in flight (a
  //     __syncthreads() would drain vmcnt); the fragment reads of the K loop are complete once their MFMAs issued.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  TD_STAMP(3);
  float* stg = (float*)((wave < 2 ? smem0 : smem1) + (wave & 1) * (WM * WN * 4));
  const float alpha = p.alpha;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int row = j * 16 + lr;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int c = (i * 4 + lg) ^ (row & (CPRW - 1));
      f32x4 a = acc[i][j];
      *(float4*)(stg + row * WN + c * 4) = make_float4(a[0] * alpha, a[1] * alpha, a[2] * alpha, a[3] * alpha);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float bias[EPL];
#pragma unroll
  for (int r = 0; r < EPL; ++r) bias[r] = (p.bias && n + r < d.Nc) ? p.bias[n + r] : 0.f;
  TD_STAMP(4);
  const uint32_t eff_seed = effective_seed(p.seed, p.drop_thresh ? p.seed_dev : nullptr);
  // (3) row-contiguous epilogue + 16-byte stores
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (!live[it]) continue;
    const int row = it * RPI + rsub;
    const int sw = row & (CPRW - 1);
    float v[EPL];
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) {
      float4 f = *(const float4*)(stg + row * WN + (((cc * (EPL / 4) + q) ^ sw) * 4));
      v[4 * q + 0] = f.x + bias[4 * q + 0]; v[4 * q + 1] = f.y + bias[4 * q + 1];
      v[4 * q + 2] = f.z + bias[4 * q + 2]; v[4 * q + 3] = f.w + bias[4 * q + 3];
    }
    const size_t off = offs[it];
    if (vec_ok) {
      if (p.residual) {
        float r8[EPL];
        unpack16<T>(res[it], r8);
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] += r8[r];
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.sigmoid) {
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = sigmoidf_(v[r]);
      }
      if (p.mask_src) {
        float m8[EPL];
        unpack16<T>(msk[it], m8);
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = m8[r] > 0.f ? v[r] : 0.f;
      }
      if (p.drop_thresh) {
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = dropout_keep(eff_seed, (uint32_t)(off + r), p.drop_thresh) ? v[r] * p.drop_scale : 0.f;
      }
      st16(p.out + off * ES, pack16<T>(v));
    } else {
      const int cnt = min(EPL, d.Nc - n);
      for (int r = 0; r < cnt; ++r) {
        float x = v[r];
        if (p.residual) x += Elem<T>::load(p.residual, (GX ? roffs[it] : off) + r);
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.sigmoid) x = sigmoidf_(x);
        if (p.mask_src) x = Elem<T>::load(p.mask_src, off + r) > 0.f ? x : 0.f;
        if (p.drop_thresh) x = dropout_keep(eff_seed, (uint32_t)(off + r), p.drop_thresh) ? x * p.drop_scale : 0.f;
        Elem<T>::store(p.out, off + r, x);
      }
    }
  }
  TD_STAMP(5);
#undef TD_STAMP
}

// ------------------------------------------------------------------------------------------------
// 256-row tiles for the MFMA-bound layers (bf16; every 3x3 of layer2..4, the 1x1 layers with K >= 512), forward and input
// gradient.  Why a second instance next to conv_gemm_kernel: a 128 x 128 x 64 tile moves 32 KiB HBM/L2 -> LDS per 512 MFMA cycles of
// the CU, i.e. it needs the full 64 B/clk/CU of the vector-memory path to keep the matrix pipes busy, and measured it sits at a
// third of both.  Here one workgroup of EIGHT wavefronts (two per SIMD, the only workgroup of its CU: 2 x 64 KiB stages) owns a
// 256 x 256 (conv_gemm_big8_kernel) or 256 x 128 (conv_gemm_big8n_kernel) tile: the same DMA instruction count per wavefront and
// tile feeds twice the MFMAs (32 B/clk/CU at peak), the activation rows of a 256-channel layer are read exactly once, and each
// wavefront's 128 x 64 sub-tile needs 12 KiB of LDS fragment reads per 32 MFMAs instead of 16.  Accumulators: 128 VGPRs per lane.
// (Measured and dropped: FOUR wavefronts with 128 x 128 sub-tiles and 256 AGPR-pinned accumulators each - 192 instead of 256 KiB of
// LDS traffic per K tile - ran the layer3 3x3 forward in 602 us against 493 us: one wavefront per SIMD cannot cover its own LDS
// and barrier latencies.)  TU / pointwise addressing exactly as in conv_gemm_kernel.
//
// The main loop is PHASED (round 4).  Its round-3 predecessor kept the eight wavefronts in lock step: once per K tile every
// wavefront issued its 8 DMA pieces and their address arithmetic back to back, then drained its own DMA (vmcnt(0)) in front of the
// tile's one barrier - both SIMD-resident wavefronts away from the matrix pipe at the same moment (2840 cycles per K tile measured
// against 2 x 64 MFMAs x 17 = 2176; removed in round 5, the phased kernels were bit-identical to it).  Here
//   * the wavefronts form two groups (0-3 / 4-7: one wavefront of each group per SIMD) that run the same instruction stream ONE
//     BARRIER APART: a K tile is four phases of { load section | barrier | 16 MFMAs | barrier }, so while one group multiplies the
//     other one reads its fragments and issues DMA - matrix beside memory on every SIMD, never matrix beside matrix;
//   * DMA pieces are spread over the phases, two per wavefront and phase, and run 4-6 phases (about two K tiles) ahead of their
//     first use; a load section ends with a COUNTED s_waitcnt vmcnt(7..10) that retires exactly the pieces the next phase reads
//     and leaves everything younger in flight across the barriers - vmcnt(0) does not occur in the loop;
//   * what is in flight lands in LDS regions whose last reader finished at least two barrier intervals earlier: the weight
//     fragments of a tile are read once (phase 0) into 32 registers, so the weight half of a stage is free for tile kt + 2 from
//     phase 2 of tile kt on; the activation rows are consumed a quarter per phase (the wavefront's M fragments 2q, 2q + 1), and
//     tile kt + 1's quarters go into the other stage during phases 0 and 1;
//   * s_setprio(1) around each MFMA cluster (the groups are in different roles at any time, so the arbiter has something to prefer).
// Fragment reads are inline asm: the compiler's wait-count pass would put a vmcnt in front of any LDS read that may alias an
// LDS-DMA in flight (here: always), i.e. re-serialise the loop; with asm reads it sees no LDS read in the loop and the
// ordering is the counted waits + barriers written out below.  Ordering rules used (MI355X_MICROARCH.md, "Two waves per SIMD" 7):
// RAW - the issuing wavefront's vmcnt, then a barrier, then the read (one phase later for either group); WAR - a region is
// re-staged no earlier than two barrier intervals after its last read was issued (the reads are retired by the lgkmcnt(0) right
// behind the next barrier).  Tiles past the end of K are issued with out-of-range offsets (zero fill, no traffic) so that the
// counts are the same in every iteration.
#ifndef TD_ABL
#define TD_ABL 0  // timing ablations of conv_gemm_big8_kernel (tools/build_variant.sh; results are WRONG with any bit set): 1 (was: no K walk), 2 no DMA, 4 no fragment reads, 8 no s_setprio, 16 stage buffers pre-filled with pseudo-random values (with 2: MFMAs on toggling operands without any L2 -> LDS traffic), 32 activation pieces of taps 1.. out of range
#endif
template <int OFF>
__device__ __forceinline__ u32x4_t lds_read16(uint32_t addr) {
  u32x4_t r;
#if TD_ABL & 4
  asm volatile("" : "=v"(r) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
#endif
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int I>
using ic = std::integral_constant<int, I>;

#ifndef TD_BIG8_AUX_W
#define TD_BIG8_AUX_W 0  // cache-policy bits of the weight / activation LDS-DMA loads of conv_gemm_big8_kernel (A/B builds: 2 = nt)
#endif
#ifndef TD_BIG8_AUX_X
#define TD_BIG8_AUX_X 0
#endif
template <bool TU, bool RES>  // RES: a residual operand in the epilogue (rare on this instance - the K = 512 expansions of layer4 -: its own instantiation keeps the loads out of everybody else's epilogue)
__global__ __launch_bounds__(512, 1) void conv_gemm_big8_kernel(GemmParams p) {
  using T = u16;
  constexpr int ES = 2, BM = 256, BN = 256, BK = 64, NW = 8, VEC = 8;
  constexpr int WM = 128, WN = 64, TM = 8, TN = 4;
  constexpr int STAGE = (BM + BN) * 128, WREG = BM * 128;  // one stage: activation rows, then weight rows (128 B per row)
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  constexpr int MAXT = 160;  // K tiles + 4 look-ahead / round-up entries (host: K <= 64 * (MAXT - 4))
  constexpr int STG = 16 * WN * ES;  // output transposition staging per wavefront: 16 rows x 64 channels of bf16 (2 KiB)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  __shared__ __attribute__((aligned(16))) uint4 ktab[MAXT];  // per K tile: {activation byte offset of the tile's tap / columns, tap, weight byte offset, in-range mask}
  __shared__ __attribute__((aligned(16))) char stg_all[NW * STG];  // (its own 16 KiB: the stage buffers take the NEXT tile's first pieces while this one is stored)

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int NT = d.Nc / BN;
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const int HoWo = d.Ho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const uint32_t out_bytes = (uint32_t)p.M * (uint32_t)d.ldc * ES;  // (host: M * ldc < 2^31)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, out_bytes, 0x00020000);
  const int wy = wave >> 2, wx = wave & 3;  // wy is also the wavefront's group
  const int RS = d.R * d.S;
  const int lane_c = chunk * VEC;
  // The K walk is a TABLE in LDS, built once per workgroup: entry t = where tile t's 64 columns come from.  (As a running
  // (tap, channel, pixel) state advanced in the loop it was ~35 dependent scalar instructions per advance, two advances per tile:
  // 2697 -> 2239 cycles per K tile with the advance compiled out, tools/build_variant.sh abl1 - the scalar unit issues into the
  // same in-order stream as the wavefront's MFMAs.)  Tile order: tap-inner = all R*S taps of one 64-channel chunk back to back
  // (the workgroup's window stays L2-resident), else tap-major; pointwise: columns t*64.  Entries past K are marked invalid.
  const int nk = p.K / BK;
  const bool tap_inner = TU && p.tap_inner;
  if (t < nk + 4) {  // (the loop runs an even number of tiles and looks two ahead)
    uint32_t xo, tap = 0, wo;
    if constexpr (TU) {
      const int cpt = d.C / BK;
      const int chunkc = tap_inner ? t / RS : t % cpt;
      tap = tap_inner ? t - chunkc * RS : t / cpt;
      const int r = (int)tap / d.S, sx = (int)tap - r * d.S;
      const int pix = d.mode == 0 ? r * d.Ws + sx : -(r * d.Ws + sx);
      xo = (uint32_t)(pix * d.C + chunkc * BK) * ES;
      wo = (uint32_t)((int)tap * d.C + chunkc * BK) * ES;
    } else {
      xo = wo = (uint32_t)t * BK * ES;
    }
    ktab[t] = make_uint4(xo, t < nk ? tap : 0u, wo, t < nk ? 0xFFFFFFFFu : 0u);
  }
  // PERSISTENT, with the next tile's first pieces in flight under this tile's epilogue (round 5).  The workgroup walks output tiles
  // vb = blockIdx.x, + gridDim.x, ... (the host launches one workgroup per CU when there are more tiles than CUs; gridDim.x is a
  // multiple of 8, so a workgroup stays on its XCD's share of the rows).  What the 100 MHz wall clock said about the one-tile-per-
  // workgroup form (tools/stamp_occupancy.py, tools/stamp_big.py; layer3 3x3, 800 frames, 62 us per tile and CU): K loop 50 us; per-
  // lane row decoding with integer divisions and a test per tap ~4 us; landing time of the first pieces 2.2 us; epilogue 4.9 us
  // (fp32 transposition in 16-row chunks through the stage buffers, two lgkmcnt(0) round trips per chunk) - 19 % of a tile with
  // the matrix pipe idle.  Now: rows decoded with reciprocal multiplications; the accumulators leave through a wavefront-private
  // 2-KiB bf16 staging region (bias / ReLU / rounding applied in the MFMA layout, 8-byte writes, 16-byte reads, both bank-conflict
  // free under the chunk ^ row swizzle; DS operations of a wavefront execute in order, so no wait separates a chunk's reads from
  // the next chunk's writes), and the twelve pieces of the NEXT tile's prologue are issued BEFORE the epilogue starts.
#if TD_ABL & 16  // (with 2: no LDS-DMA, but the stage buffers hold pseudo-random bf16 values around 1.0 - the matrix pipes then toggle as on real data)
  for (int i = t; i < 2 * STAGE / 4; i += 512) ((uint32_t*)smem)[i] = 0x3F803F80u ^ (mix32((uint32_t)i * 0x9E3779B1u + blockIdx.x) & 0x007F007Fu);
#endif
  const int nvb = 8 * ((cdiv(p.M, BM) + 7) / 8) * NT;
  const float rcp_howo = 1.f / (float)HoWo, rcp_wo = 1.f / (float)d.Wo;
  auto divmod = [](int a, int dv, float rcp, int& q, int& r) {
    q = (int)((float)a * rcp);
    r = a - q * dv;
    const int up = r >= dv, dn = r < 0;
    q += up - dn;
    r += (dn - up) * dv;
  };
  // tile (rows m0.., columns n0..) of virtual block vb; the blocks past the last row tile (grid rounded up to 8) can only be a
  // workgroup's LAST ones
#if TD_BIG_XCD_CONTIG
  // XCD x (= vb & 7: workgroups go to XCDs round-robin, the grid is a multiple of 8) walks a CONTIGUOUS range of row tiles: neighbouring
  // row tiles of a 3x3 layer share their halo rows (2 x (W + 1) of 256 rows at W = 22: 18 %), and with row tile = 8 * seq + x every
  // one of them was fetched into two L2s (traffic 1.19x of the algorithmic bytes); now the 32 tiles an XCD has in flight are neighbours.
  const int mtx = (cdiv(p.M, BM) + 7) / 8;
  auto tile_m0 = [&](int vb) { const int seq = vb >> 3; return ((vb & 7) * mtx + seq / NT) * BM; };
  auto tile_n0 = [&](int vb) { const int seq = vb >> 3; return (seq - (seq / NT) * NT) * BN; };
  auto tile_ok = [&](int vb) { return vb < nvb && (vb >> 3) / NT < mtx && tile_m0(vb) < p.M; };
#else
  auto tile_m0 = [&](int vb) { const int seq = vb >> 3; return ((seq / NT) * 8 + (vb & 7)) * BM; };
  auto tile_n0 = [&](int vb) { const int seq = vb >> 3; return (seq - (seq / NT) * NT) * BN; };
  auto tile_ok = [&](int vb) { return vb < nvb && tile_m0(vb) < p.M; };
#endif

  // DMA piece q (0..3) of this wavefront on the activation side: tile rows xrow(q) .. + 8 = the part of QUARTER q (the rows the
  // readers consume in phase q: M fragments 2q, 2q + 1 of both row groups) that this wavefront stages
  uint32_t a_off[4];   // byte offset of the row (pointwise) / of tap (0, 0) of the row (TU; modulo 2^32)
  uint32_t a_mask[4];  // bit (tap) set = the tap lies inside the image (pointwise: bit 0 = row below M)
  uint32_t b_off[4];
  const bool small_filter = d.R <= 3 && d.S <= 3;
  const int sgn = d.mode == 0 ? 1 : -1;  // forward: tap (r, s) reads pixel (hb + r, wb + s); input gradient: (hb - r, wb - s)
  auto tile_addr = [&](int m0, int n0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + (wave >> 2) * WM + q * 32 + (wave & 3) * 8 + lrow;
      const bool ok = m < p.M;
      if constexpr (!TU) {
        a_off[q] = (uint32_t)m * (uint32_t)p.K * ES + (uint32_t)lane_c * ES;
        a_mask[q] = ok ? 1u : 0u;  // (the table's tap is 0 for pointwise tiles)
      } else {
        // (reciprocal multiplication + one correction step: exact for rows below 2^24, the host's condition for this instance)
        const int mm = ok ? m : 0;
        int n, rem, ho, wo;
        divmod(mm, HoWo, rcp_howo, n, rem);
        divmod(rem, d.Wo, rcp_wo, ho, wo);
        const int hb = d.mode == 0 ? ho * d.stride - d.pad : ho + d.pad;
        const int wb = d.mode == 0 ? wo * d.stride - d.pad : wo + d.pad;
        a_off[q] = (uint32_t)(n * d.Hs * d.Ws + hb * d.Ws + wb) * (uint32_t)d.C * ES + (uint32_t)lane_c * ES;
        uint32_t cols = 0, msk = 0;
        if (small_filter) {  // (uniform) at most 3 x 3 taps: no loop, no branch per tap
#pragma unroll
          for (int sx = 0; sx < 3; ++sx) cols |= (sx < d.S && (unsigned)(wb + sgn * sx) < (unsigned)d.Ws ? 1u : 0u) << sx;
#pragma unroll
          for (int r = 0; r < 3; ++r) msk |= (r < d.R && (unsigned)(hb + sgn * r) < (unsigned)d.Hs ? cols : 0u) << (r * d.S);
        } else {
          for (int sx = 0; sx < d.S; ++sx) cols |= ((unsigned)(wb + sgn * sx) < (unsigned)d.Ws ? 1u : 0u) << sx;
          for (int r = 0; r < d.R; ++r) msk |= ((unsigned)(hb + sgn * r) < (unsigned)d.Hs ? cols : 0u) << (r * d.S);
        }
        a_mask[q] = ok ? msk : 0u;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) b_off[i] = ((uint32_t)(n0 + (i * NW + wave) * 8 + lrow) * (uint32_t)p.K + (uint32_t)lane_c) * ES;  // Nc % BN == 0: always inside
  };
  // one activation piece (quarter q) / one weight piece (i) of the tile described by table entry e into `stage`; branch-free: an
  // out-of-image tap, a row past M or a tile past K becomes the out-of-range offset through an all-ones / all-zeros mask
  auto issue_x = [&](char* stage, int q, const u32x4_t& e) {
    uint32_t ok = (0u - ((a_mask[q] >> (e.y & 31)) & 1u)) & e.w;
#if TD_ABL & 32  // (timing only) activation rows fetched for tap 0 of every channel chunk only, the other taps' pieces are out-of-range (zero fill, no L2 traffic): what staging a chunk's rows ONCE for all nine taps could save at most
    ok &= e.y == 0 ? 0xFFFFFFFFu : 0u;
#endif
    uint32_t off = a_off[q] + e.x;  // a_off (TU) = byte offset of tap (0, 0) of the row (mod 2^32: may be "negative")
    off = (off & ok) | (OOB & ~ok);
#if TD_ABL & 2
    asm volatile("" ::"v"(off));
    return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(stage + ((wave >> 2) * WM + q * 32 + (wave & 3) * 8) * 128), 16, off, 0, 0, TD_BIG8_AUX_X);
  };
  auto issue_w = [&](char* stage, int i, const u32x4_t& e) {
    const uint32_t off = ((b_off[i] + e.z) & e.w) | (OOB & ~e.w);
#if TD_ABL & 2
    asm volatile("" ::"v"(off));
    return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(stage + WREG + (i * NW + wave) * 1024), 16, off, 0, 0, TD_BIG8_AUX_W);
  };

  const int lr = lane & 15, lg = lane >> 4;
  f32x4 acc[TN][TM];

  // fragment read addresses (LDS byte offsets; the 16-byte chunk index is XOR-swizzled by row & 7 = lr & 7, the k-step flips bit 6)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t xa0 = lds0 + (uint32_t)((wy * WM + lr) * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t wa0 = lds0 + (uint32_t)(WREG + (wx * WN + lr) * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t xa[2][2] = {{xa0, xa0 ^ 64u}, {xa0 + STAGE, (xa0 ^ 64u) + STAGE}};  // [stage][k-step]: the offset field of a DS instruction holds 16 bits
  const uint32_t wa[2][2] = {{wa0, wa0 ^ 64u}, {wa0 + STAGE, (wa0 ^ 64u) + STAGE}};
  u32x4_t wreg[2][TN], xreg[2][2];  // [k-step][N fragment], [M fragment of the phase][k-step]

  // table entries: tile kt in stage 0 -> ea = entry kt + 1 (its activations are staged in phases 0, 1), eb = entry kt + 2 (its weights in
  // phases 2, 3; eb is read in phase 0 of the same tile and returns with that phase's fragments); tile kt + 1 in stage 1 swaps the roles
  const uint32_t tab0 = (uint32_t)(uintptr_t)(lds_ptr_t)ktab;
  uint32_t taddr = tab0 + 2 * 16;  // entry the next tile reads
  u32x4_t ea, eb;
  // one phase of the tile in stage S
  auto phase = [&](auto S_, auto P_) {
    constexpr int S = decltype(S_)::value, P = decltype(P_)::value;
    u32x4_t& ex = S == 0 ? ea : eb;  // entry kt + 1
    u32x4_t& ew = S == 0 ? eb : ea;  // entry kt + 2
    char* const cur = smem + S * STAGE;
    char* const oth = smem + (1 - S) * STAGE;
    // ---- load section ----
    if constexpr (P == 0) {
      wreg[0][0] = lds_read16<0 * 2048>(wa[S][0]); wreg[0][1] = lds_read16<1 * 2048>(wa[S][0]);
      wreg[0][2] = lds_read16<2 * 2048>(wa[S][0]); wreg[0][3] = lds_read16<3 * 2048>(wa[S][0]);
    }
    xreg[0][0] = lds_read16<(2 * P + 0) * 2048>(xa[S][0]);
    xreg[1][0] = lds_read16<(2 * P + 1) * 2048>(xa[S][0]);
    if constexpr (P == 0) {
      wreg[1][0] = lds_read16<0 * 2048>(wa[S][1]); wreg[1][1] = lds_read16<1 * 2048>(wa[S][1]);
      wreg[1][2] = lds_read16<2 * 2048>(wa[S][1]); wreg[1][3] = lds_read16<3 * 2048>(wa[S][1]);
    }
    xreg[0][1] = lds_read16<(2 * P + 0) * 2048>(xa[S][1]);
    xreg[1][1] = lds_read16<(2 * P + 1) * 2048>(xa[S][1]);
    if constexpr (P == 0) { ew = lds_read16<0>(taddr); taddr += 16; issue_x(oth, 0, ex); issue_x(oth, 1, ex); wait_vmcnt<8>(); }
    if constexpr (P == 1) { issue_x(oth, 2, ex); issue_x(oth, 3, ex); wait_vmcnt<9>(); }
    if constexpr (P == 2) { issue_w(cur, 0, ew); issue_w(cur, 1, ew); wait_vmcnt<10>(); }
    if constexpr (P == 3) { issue_w(cur, 2, ew); issue_w(cur, 3, ew); wait_vmcnt<7>(); }
    __builtin_amdgcn_s_barrier();
    // ---- MFMA section ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#if !(TD_ABL & 8)
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][2 * P + j]) : "v"(wreg[ks][i]), "v"(xreg[j][ks]));  // accumulate in place
#if !(TD_ABL & 8)
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  // prologue of a tile = the issue order of the steady state: weights of tile 0, activations of tile 0, weights of tile 1 (twelve pieces)
  auto prologue_issue = [&]() {
    u32x4_t e0 = lds_read16<0>(tab0);
    ea = lds_read16<16>(tab0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(ea) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(smem, i, e0);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_x(smem, q, e0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(smem + STAGE, i, ea);
  };

  // ---- epilogue addressing.  Write side: the MFMA layout - lane (lr, lg) holds channels i * 16 + lg * 4 .. + 4 of row lr of fragment
  // (i, j): 8 bytes of bf16 at 16-byte chunk 2 i + (lg >> 1), half lg & 1.  Read side: 8 lanes x 16 bytes per 128-byte row, 8 rows per
  // pass.  Chunk index XOR (row & 7) on both sides: the 8-byte writes of a wavefront spread over all 64 banks twice (rows r / r + 8),
  // each 16-lane group of the 16-byte reads over the sixteen 16-byte slots of a 256-byte bank row. ----
  const uint32_t stg0 = (uint32_t)(uintptr_t)(lds_ptr_t)stg_all + (uint32_t)wave * STG;
  uint32_t swa[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) swa[i] = stg0 + (uint32_t)(lr * 128 + (((2 * i + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8);
  const int cc = lane & 7, rsub = lane >> 3;
  const uint32_t sra = stg0 + (uint32_t)(rsub * 128 + ((cc ^ rsub) << 4));  // pass h: + h * 1024 (row h * 8 + rsub: the same row & 7)
  const float alpha = p.alpha;

#define TD_STAMP(i) do { if (p.dbg && t == 0) p.dbg[(size_t)vb * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
  int vb = blockIdx.x;
  if (!tile_ok(vb)) return;  // (uniform)
  tile_addr(tile_m0(vb), tile_n0(vb));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the K table is complete
  prologue_issue();
#pragma unroll 1
  for (;;) {
    const int m0 = tile_m0(vb), n0 = tile_n0(vb);
    TD_STAMP(0);
    if (p.dbg && t == 0) p.dbg[(size_t)vb * 8 + 6] = wall_clock64();  // (100 MHz, the same base on every CU: workgroup timeline, tools/stamp_occupancy.py)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    taddr = tab0 + 2 * 16;
    f32x4 bs[TN];  // bias of this lane's channels in the MFMA layout: requested now, used after the K loop
#pragma unroll
    for (int i = 0; i < TN; ++i) bs[i] = p.bias ? *(const f32x4*)(p.bias + n0 + wx * WN + i * 16 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // weights of K tile 0 and activation quarter 0 have landed (with stores of the previous tile's epilogue still in the queue the count
    // is only stricter: loads retire in order among themselves, whatever else is outstanding)
    wait_vmcnt<7>();
    __builtin_amdgcn_s_barrier();
    if (wy == 1) __builtin_amdgcn_s_barrier();  // the second group runs one barrier behind the first from here on
    TD_STAMP(1);
    // ONE loop over tile pairs and nothing else (an odd tile count is rounded up: the extra tile is zero fill without traffic) - an
    // exit in the middle of the body makes the register allocator rename the accumulators across the two halves and spill them
    const int npair = (nk + 1) / 2;
#pragma unroll 1
    for (int it = 0; it < npair; ++it) {
      phase(ic<0>{}, ic<0>{}); phase(ic<0>{}, ic<1>{}); phase(ic<0>{}, ic<2>{}); phase(ic<0>{}, ic<3>{});
      phase(ic<1>{}, ic<0>{}); phase(ic<1>{}, ic<1>{}); phase(ic<1>{}, ic<2>{}); phase(ic<1>{}, ic<3>{});
    }
    if (wy == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's last barrier: every wavefront has read its last fragments
    TD_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the trailing zero-fill DMAs have landed: the stage buffers are free
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the asm MFMAs' results are read below: the hazard the compiler would pad for
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(acc[i][j]));
    // (the bias registers are defined HERE, before the next tile's LDS-DMA pieces are issued: a register load still outstanding beside
    //  them would be waited for with vmcnt(0) by the compiler, i.e. together with them)
#pragma unroll
    for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(bs[i]));
    const int vb_next = vb + (int)gridDim.x;
    const bool more = tile_ok(vb_next);
    if (more) {  // (uniform) the next tile's rows, and its first twelve pieces into the stage buffers every wavefront has left
      tile_addr(tile_m0(vb_next), tile_n0(vb_next));
      prologue_issue();
    }
    TD_STAMP(3);
    // ---- epilogue: 16 rows (one M fragment) at a time ----
    uint4 mk[2][2];
    auto seg_off = [&](int j, int h) -> uint32_t {  // byte offset of this lane's 16-byte output segment of row h * 8 + rsub of fragment row j
      const int m = m0 + wy * WM + j * 16 + h * 8 + rsub;
      return m < p.M ? ((uint32_t)m * (uint32_t)d.ldc + (uint32_t)(n0 + wx * WN + cc * 8)) * ES : OOB;
    };
    auto fetch_mask = [&](int j, int b) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t off = seg_off(j, h);
        mk[b][h] = off != OOB ? ld16(p.mask_src + off) : make_uint4(0, 0, 0, 0);
      }
    };
    // fragment row j: bias / residual / ReLU / rounding in the MFMA layout, then four 8-byte writes into the staging rows
    auto put = [&](int j) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const f32x4 a = acc[i][j];
        float v[4] = {a[0] * alpha + bs[i][0], a[1] * alpha + bs[i][1], a[2] * alpha + bs[i][2], a[3] * alpha + bs[i][3]};
        if constexpr (RES) {  // 8 bytes per lane in the MFMA layout
          const int m = m0 + wy * WM + j * 16 + lr;
          if (m < p.M) {
            const uint2 rr = *(const uint2*)(p.residual + ((size_t)m * d.ldc + n0 + wx * WN + i * 16 + lg * 4) * ES);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          }
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
        asm volatile("ds_write_b64 %0, %1" ::"v"(swa[i]), "v"(u32x2_t{lo, hi}) : "memory");
      }
    };
    // Pipelined over the fragment rows with ONE staging buffer: DS operations of a wavefront execute in order, so the writes of row
    // j + 1 may be issued right behind the reads of row j (which then return the old bytes), and lgkmcnt(4) - the four younger writes
    // may still be outstanding - says that those reads have returned.  The arithmetic of row j + 1 covers the LDS round trip of row j.
    if (p.mask_src) fetch_mask(0, 0);
    put(0);
    u32x4_t o0, o1;
    asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(sra) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(o1) : "v"(sra) : "memory");
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int b = j & 1;
      if (j + 1 < TM) {
        if (p.mask_src) fetch_mask(j + 1, b ^ 1);
        put(j + 1);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(o0), "+v"(o1) : : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1) : : "memory");
      }
      u32x4_t s0 = o0, s1 = o1;
      if (j + 1 < TM) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(sra) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(o1) : "v"(sra) : "memory");
      }
      if (p.mask_src) {  // ReLU mask of the producing layer (input gradients): a select on the rounded values
        float x8[8], m8[8];
        unpack16<T>(make_uint4(s0.x, s0.y, s0.z, s0.w), x8);
        unpack16<T>(mk[b][0], m8);
#pragma unroll
        for (int r = 0; r < 8; ++r) x8[r] = m8[r] > 0.f ? x8[r] : 0.f;
        const uint4 q0 = pack16<T>(x8);
        s0 = u32x4_t{q0.x, q0.y, q0.z, q0.w};
        unpack16<T>(make_uint4(s1.x, s1.y, s1.z, s1.w), x8);
        unpack16<T>(mk[b][1], m8);
#pragma unroll
        for (int r = 0; r < 8; ++r) x8[r] = m8[r] > 0.f ? x8[r] : 0.f;
        const uint4 q1 = pack16<T>(x8);
        s1 = u32x4_t{q1.x, q1.y, q1.z, q1.w};
      }
      __builtin_amdgcn_raw_buffer_store_b128(s0, rs_out, (int)seg_off(j, 0), 0, 0);  // rows past M: out-of-range offset, dropped
      __builtin_amdgcn_raw_buffer_store_b128(s1, rs_out, (int)seg_off(j, 1), 0, 0);
    }
    TD_STAMP(5);
    if (p.dbg && t == 0) p.dbg[(size_t)vb * 8 + 7] = wall_clock64();
    if (!more) break;
    vb = vb_next;
  }
#undef TD_STAMP
}

// ------------------------------------------------------------------------------------------------
// The phased main loop for 128 output channels (layer2's 3x3 layers and its conv1): 256 x 128 tiles, THREE 48-KiB stages.
// Same two wavefront groups one barrier apart, same counted waits and table walk as conv_gemm_big8_kernel; what differs:
//   * wavefront tile 64 x 64 (4 M x 2 N wavefronts), so a K tile is TWO phases of 16 MFMAs (M fragments 2p, 2p + 1);
//   * with three stages tile kt + 2 goes into the stage tile kt - 1 has left: all of its six pieces per wavefront (2 weight,
//     2 + 2 activation) are issued during tile kt - four in phase 0, two in phase 1 - into regions whose last read lies at least
//     three barrier intervals back; every piece has four phases of lead;
//   * waits: before phase 0 of a tile vmcnt(8), before phase 1 vmcnt(10) (the pieces issued since the needed one);
//   * the loop is unrolled over three tiles (stage = compile-time constant); the table entry that drives the pieces of tile
//     T + 2 is read in phase 1 of tile T - 1 and returns with that phase's fragments.
template <bool TU>
__global__ __launch_bounds__(512, 1) void conv_gemm_big8n_kernel(GemmParams p) {
  using T = u16;
  constexpr int ES = 2, BM = 256, BN = 128, BK = 64, NW = 8, VEC = 8;
  constexpr int WM = 64, WN = 64, TM = 4, TN = 4;
  constexpr int WREG = BM * 128, STAGE = (BM + BN) * 128;
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  constexpr int MAXT = 160;
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
  __shared__ __attribute__((aligned(16))) uint4 ktab[MAXT];

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int NT = d.Nc / BN;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
#if TD_BIG_XCD_CONTIG
  const int mtx = (cdiv(p.M, BM) + 7) / 8;  // an XCD takes a contiguous range of row tiles (halo rows shared in ITS L2: see conv_gemm_big8_kernel)
  const int mt = xcd * mtx + seq / NT, nt = seq - (seq / NT) * NT;
#else
  const int mt = (seq / NT) * 8 + xcd, nt = seq - (seq / NT) * NT;
#endif
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= p.M) return;
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const int lane_c = chunk * VEC;
  const int HoWo = d.Ho * d.Wo;
  const int RS = d.R * d.S;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int wy = wave >> 1, wx = wave & 1;
  const int grp = wave >> 2;  // the wavefront's group (one wavefront of each group per SIMD)

  // activation piece (h, k): half h (the rows phase h consumes: M fragments 2h, 2h + 1 of every row group), this wavefront's
  // rows (wave >> 1) * 64 + h * 32 + ((wave & 1) * 2 + k) * 8 .. + 8
  uint32_t a_off[4], a_mask[4];
  const float rcp_howo = 1.f / (float)HoWo, rcp_wo = 1.f / (float)d.Wo;
  auto divmod = [](int a, int dv, float rcp, int& q, int& r) {
    q = (int)((float)a * rcp);
    r = a - q * dv;
    const int up = r >= dv, dn = r < 0;
    q += up - dn;
    r += (dn - up) * dv;
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + (wave >> 1) * WM + (q >> 1) * 32 + ((wave & 1) * 2 + (q & 1)) * 8 + lrow;
    const bool ok = m < p.M;
    if constexpr (!TU) {
      a_off[q] = ((uint32_t)m * (uint32_t)p.K + (uint32_t)lane_c) * ES;
      a_mask[q] = ok ? 1u : 0u;
    } else {
      const int mm = ok ? m : 0;
      int n, rem, ho, wo;
      divmod(mm, HoWo, rcp_howo, n, rem);
      divmod(rem, d.Wo, rcp_wo, ho, wo);
      const int hb = d.mode == 0 ? ho * d.stride - d.pad : ho + d.pad;
      const int wb = d.mode == 0 ? wo * d.stride - d.pad : wo + d.pad;
      a_off[q] = ((uint32_t)(n * d.Hs * d.Ws + hb * d.Ws + wb) * (uint32_t)d.C + (uint32_t)lane_c) * ES;
      uint32_t cols = 0, msk = 0;
      for (int sx = 0; sx < d.S; ++sx) cols |= ((unsigned)(d.mode == 0 ? wb + sx : wb - sx) < (unsigned)d.Ws ? 1u : 0u) << sx;
      for (int r = 0; r < d.R; ++r) msk |= ((unsigned)(d.mode == 0 ? hb + r : hb - r) < (unsigned)d.Hs ? cols : 0u) << (r * d.S);
      a_mask[q] = ok ? msk : 0u;
    }
  }
  uint32_t b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = ((uint32_t)(n0 + (i * NW + wave) * 8 + lrow) * (uint32_t)p.K + (uint32_t)lane_c) * ES;
  const int nk = p.K / BK;
  const bool tap_inner = TU && p.tap_inner;
  if (t < nk + 6) {  // (the loop runs a multiple of three tiles and looks two ahead)
    uint32_t xo, tap = 0, wo;
    if constexpr (TU) {
      const int cpt = d.C / BK;
      const int chunkc = tap_inner ? t / RS : t % cpt;
      tap = tap_inner ? t - chunkc * RS : t / cpt;
      const int r = (int)tap / d.S, sx = (int)tap - r * d.S;
      const int pix = d.mode == 0 ? r * d.Ws + sx : -(r * d.Ws + sx);
      xo = (uint32_t)(pix * d.C + chunkc * BK) * ES;
      wo = (uint32_t)((int)tap * d.C + chunkc * BK) * ES;
    } else {
      xo = wo = (uint32_t)t * BK * ES;
    }
    ktab[t] = make_uint4(xo, t < nk ? tap : 0u, wo, t < nk ? 0xFFFFFFFFu : 0u);
  }
  auto issue_x = [&](char* stage, int q, const u32x4_t& e) {
    const uint32_t ok = (0u - ((a_mask[q] >> (e.y & 31)) & 1u)) & e.w;
    uint32_t off = a_off[q] + e.x;
    off = (off & ok) | (OOB & ~ok);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(stage + ((wave >> 1) * WM + (q >> 1) * 32 + ((wave & 1) * 2 + (q & 1)) * 8) * 128), 16, off, 0, 0, 0);
  };
  auto issue_w = [&](char* stage, int i, const u32x4_t& e) {
    const uint32_t off = ((b_off[i] + e.z) & e.w) | (OOB & ~e.w);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(stage + WREG + (i * NW + wave) * 1024), 16, off, 0, 0, 0);
  };

  const int lr = lane & 15, lg = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t xa0 = lds0 + (uint32_t)((wy * WM + lr) * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t wa0 = lds0 + (uint32_t)(WREG + (wx * WN + lr) * 128 + ((lg ^ (lr & 7)) << 4));
  uint32_t xa[3][2], wa[3][2];  // [stage][k-step]
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) {
    xa[s_][0] = xa0 + s_ * STAGE; xa[s_][1] = (xa0 ^ 64u) + s_ * STAGE;
    wa[s_][0] = wa0 + s_ * STAGE; wa[s_][1] = (wa0 ^ 64u) + s_ * STAGE;
  }
  u32x4_t wreg[2][TN], xreg[2][2];
  const uint32_t tab0 = (uint32_t)(uintptr_t)(lds_ptr_t)ktab;
  uint32_t taddr = tab0 + 3 * 16;
  u32x4_t e[3];
  // one phase of tile T in stage S: pieces of tile T + 2 (table entry e[(S + 2) % 3]) into stage (S + 2) % 3
  auto phase = [&](auto S_, auto P_) {
    constexpr int S = decltype(S_)::value, P = decltype(P_)::value, S2 = (S + 2) % 3;
    char* const nxt = smem + S2 * STAGE;
    if constexpr (P == 0) {
      wreg[0][0] = lds_read16<0 * 2048>(wa[S][0]); wreg[0][1] = lds_read16<1 * 2048>(wa[S][0]);
      wreg[0][2] = lds_read16<2 * 2048>(wa[S][0]); wreg[0][3] = lds_read16<3 * 2048>(wa[S][0]);
    }
    xreg[0][0] = lds_read16<(2 * P + 0) * 2048>(xa[S][0]);
    xreg[1][0] = lds_read16<(2 * P + 1) * 2048>(xa[S][0]);
    if constexpr (P == 0) {
      wreg[1][0] = lds_read16<0 * 2048>(wa[S][1]); wreg[1][1] = lds_read16<1 * 2048>(wa[S][1]);
      wreg[1][2] = lds_read16<2 * 2048>(wa[S][1]); wreg[1][3] = lds_read16<3 * 2048>(wa[S][1]);
    }
    xreg[0][1] = lds_read16<(2 * P + 0) * 2048>(xa[S][1]);
    xreg[1][1] = lds_read16<(2 * P + 1) * 2048>(xa[S][1]);
    if constexpr (P == 0) {
      issue_w(nxt, 0, e[S2]); issue_w(nxt, 1, e[S2]); issue_x(nxt, 0, e[S2]); issue_x(nxt, 1, e[S2]);
      wait_vmcnt<10>();  // activation half 1 of THIS tile has landed (issued since: 4 + 2 pieces of tile T + 1, 4 of tile T + 2)
    } else {
      e[S] = lds_read16<0>(taddr);  // entry T + 3: drives the pieces tile T + 1 issues
      taddr += 16;
      issue_x(nxt, 2, e[S2]); issue_x(nxt, 3, e[S2]);
      wait_vmcnt<8>();   // weights and activation half 0 of tile T + 1 have landed (issued since: its half 1, 6 pieces of tile T + 2)
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][2 * P + j]) : "v"(wreg[ks][i]), "v"(xreg[j][ks]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- epilogue bookkeeping ----
  constexpr int CR = 16, NCH = WM / CR, EPL = 8, LPR = WN / EPL, RPI = 64 / LPR, NIT = CR / RPI, CPRW = WN / 4;
  const int cc = lane % LPR, rsub = lane / LPR;
  const int n = n0 + wx * WN + cc * EPL;
  uint32_t offs[2][NIT];
  bool live[2][NIT];
  uint4 res[2][NIT], msk[2][NIT];
  auto fetch_chunk = [&](int c, int b) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wy * WM + c * CR + it * RPI + rsub;
      live[b][it] = m < p.M;
      offs[b][it] = (uint32_t)min(m, p.M - 1) * (uint32_t)d.ldc + (uint32_t)n;
      if (p.residual) res[b][it] = ld16(p.residual + offs[b][it] * ES);
      if (p.mask_src) msk[b][it] = ld16(p.mask_src + offs[b][it] * ES);
    }
  };

#define TD_STAMP(i) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
  TD_STAMP(0);
  __syncthreads();  // the K table is complete
  {
    const u32x4_t e0 = lds_read16<0>(tab0);
    const u32x4_t e1 = lds_read16<16>(tab0);
    e[2] = lds_read16<32>(tab0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the issue order of the steady state: per tile 2 weight pieces, activation half 0, activation half 1
    issue_w(smem, 0, e0); issue_w(smem, 1, e0); issue_x(smem, 0, e0); issue_x(smem, 1, e0); issue_x(smem, 2, e0); issue_x(smem, 3, e0);
    issue_w(smem + STAGE, 0, e1); issue_w(smem + STAGE, 1, e1); issue_x(smem + STAGE, 0, e1); issue_x(smem + STAGE, 1, e1);
    issue_x(smem + STAGE, 2, e1); issue_x(smem + STAGE, 3, e1);
  }
  wait_vmcnt<8>();  // weights and activation half 0 of tile 0 have landed
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // the second group runs one barrier behind the first from here on
  TD_STAMP(1);
  const int ntri = (nk + 2) / 3;  // ONE loop over tile triples and nothing else (extra tiles: zero fill without traffic)
#pragma unroll 1
  for (int it = 0; it < ntri; ++it) {
    phase(ic<0>{}, ic<0>{}); phase(ic<0>{}, ic<1>{});
    phase(ic<1>{}, ic<0>{}); phase(ic<1>{}, ic<1>{});
    phase(ic<2>{}, ic<0>{}); phase(ic<2>{}, ic<1>{});
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's last barrier
  TD_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero-fill DMAs must not land in the staging regions below
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the asm MFMAs' results are read below
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(acc[i][j]));
  fetch_chunk(0, 0);
  __builtin_amdgcn_s_barrier();  // every wavefront is done with the stage buffers: they become the staging regions
  TD_STAMP(3);
  float* stg = (float*)(smem + wave * (CR * WN * 4));
  const float alpha = p.alpha;
  float bias[EPL];
#pragma unroll
  for (int r = 0; r < EPL; ++r) bias[r] = p.bias ? p.bias[n + r] : 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int b = c & 1;
    {
      const int row = lr;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int cx = (i * 4 + lg) ^ (row & (CPRW - 1));
        const f32x4 a = acc[i][c];
        *(float4*)(stg + row * WN + cx * 4) = make_float4(a[0] * alpha, a[1] * alpha, a[2] * alpha, a[3] * alpha);
      }
    }
    if (c + 1 < NCH) fetch_chunk(c + 1, b ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + rsub;
      const int sw = row & (CPRW - 1);
      float v[EPL];
#pragma unroll
      for (int q = 0; q < EPL / 4; ++q) {
        const float4 f = *(const float4*)(stg + row * WN + (((cc * (EPL / 4) + q) ^ sw) * 4));
        v[4 * q + 0] = f.x + bias[4 * q + 0]; v[4 * q + 1] = f.y + bias[4 * q + 1];
        v[4 * q + 2] = f.z + bias[4 * q + 2]; v[4 * q + 3] = f.w + bias[4 * q + 3];
      }
      if (p.residual) {
        float r8[EPL];
        unpack16<T>(res[b][it], r8);
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] += r8[r];
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.mask_src) {
        float m8[EPL];
        unpack16<T>(msk[b][it], m8);
#pragma unroll
        for (int r = 0; r < EPL; ++r) v[r] = m8[r] > 0.f ? v[r] : 0.f;
      }
      if (live[b][it]) st16(p.out + offs[b][it] * ES, pack16<T>(v));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next chunk overwrites the region
  }
  TD_STAMP(5);
#undef TD_STAMP
}

// ------------------------------------------------------------------------------------------------
// Persistent, weight-stationary instance for the HBM-bound pointwise layers with K <= 256 (bottleneck conv3 forward, conv1 input
// gradient, the encoder's 256 -> 2048 FFN layer: [M][K] x [Nc][K]^T with M in the 10^4..10^6 range).  In the tiled kernel above two
// thirds of such a workgroup's HBM -> LDS traffic is the weight tile it re-reads for every 64 rows, and with one K tile in flight per
// workgroup the chip holds too few bytes in flight to cover the HBM latency (2.9 TB/s measured).  Here two workgroups per CU keep
// their 128 x K weight tile as MFMA fragments IN REGISTERS for the whole launch and walk the 64-row M tiles of their group; the
// 128-row-of-N tiles of one M tile run on CUs of the same XCD at the same time, so the activation tile comes from HBM once per XCD.
// bf16, fused bias + residual + ReLU + mask (+ dropout) epilogue.  What bounds such a kernel is not bytes but requests in flight
// (the round-3 form, pw_resident_kernel, issued one activation tile + this tile's residual / mask rows per step and then spent the
// rest of the step - MFMAs, an accumulator transposition through the activation buffer itself, three more barriers, the stores -
// with nothing new on its way: 5.0 - 5.4 TB/s of its algorithmic bytes; removed in round 5, this kernel was bit-identical to it):
//   * the accumulator transposition has its own 4 KiB per wavefront (16 rows at a time; 2 x 32 KiB slots + 16 KiB = 80 KiB: still
//     two workgroups per CU), so the activation slot of tile t is free right after the barrier that follows its MFMAs: tile t + 2
//     is requested into it BEFORE the epilogue of tile t - two activation tiles in flight on a two-slot ring;
//   * residual / mask rows are requested ONE TILE AHEAD into a second register set (inline-asm loads with counted waits; the
//     compiler's own bookkeeping across the loop back edge would drain the queue), except for the residual + mask instances with
//     K = 256, whose 128 weight-fragment registers leave no room: they request this tile's rows before the MFMAs;
//   * two barriers per step.
// (Measured and dropped: the LDS-free epilogue of v_permlane16_swap - it stores 16 rows x 64 bytes per instruction instead of
// 8 rows x 128: 4.0 instead of 5.3 TB/s on the layer3 conv3 shape.  Half-line requests are what an HBM-bound kernel cannot afford.)
// Counted waits are derived from LOADS only (loads retire in issue order among themselves; stores in flight can only make a
// wait stricter).
template <int NKT, bool RES, bool MSK>
__global__ __launch_bounds__(256, 2) void pw_resident2_kernel(GemmParams p, int MT, int P) {
  using T = u16;
  constexpr int ES = 2;
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  constexpr int WM = 32, WN = 64, TM = 2, TN = 4;
  constexpr int CPRW = WN / 4, EPL = 8, LPR = WN / EPL, RPI = 64 / LPR, NIT = WM / RPI;   // row-contiguous epilogue: 8 lanes x 16 bytes per 64-channel row segment
  constexpr int NOPS = NIT * ((RES ? 1 : 0) + (MSK ? 1 : 0));      // operand loads per lane and step
  constexpr bool AHEAD = NOPS > 0 && !(RES && MSK && NKT == 4);    // operand rows requested one tile ahead (register budget)
  constexpr int NA = 2 * NKT;                                      // DMA pieces per wavefront and activation tile
  constexpr int AUX_NT = (TD_NT & 2) ? 2 : 0;
  __shared__ __attribute__((aligned(16))) char sA0[32768];
  __shared__ __attribute__((aligned(16))) char sA1[32768];
  __shared__ __attribute__((aligned(16))) char sT[4 * 16 * WN * 4];  // transposition staging: 16 rows x 64 fp32 per wavefront
  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int NT = d.Nc >> 7;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int GPX = P / NT;                      // M-tile groups per XCD
  const int nt = local % NT, gl = local / NT;
  if (gl >= GPX) return;
  const int gid = xcd * GPX + gl, G = 8 * GPX;
  const int lrow = lane >> 3, chunk = (lane & 7) ^ lrow;
  const uint32_t out_bytes = (uint32_t)p.M * (uint32_t)d.ldc * ES;  // (host: M * ldc < 2^31)
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.residual : p.out), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)(MSK ? p.mask_src : p.out), 0, out_bytes, 0x00020000);
  const int n0 = nt * 128;
  const int HoWo = d.Ho * d.Wo;
  const bool strided = d.stride != 1;
  auto issue_A = [&](char* buf, int mt) {
    const int m0 = mt * 64;
    uint32_t row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (i * 4 + wave) * 8 + lrow;
      int src_row = m;
      if (strided) {
        const int img = m / HoWo, rem = m - img * HoWo;
        const int ho = rem / d.Wo, wo = rem - ho * d.Wo;
        src_row = (img * d.Hs + ho * d.stride) * d.Ws + wo * d.stride;
      }
      row[i] = (mt < MT && m < p.M) ? (uint32_t)src_row * (uint32_t)p.K * ES : OOB;
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t off = row[i] != OOB ? row[i] + (uint32_t)(kt * 64 + chunk * 8) * ES : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_ptr_t)(buf + kt * 8192 + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
      }
  };
  const int wy = wave >> 1, wx = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int cc = lane % LPR, rsub = lane / LPR;
  const int n = n0 + wx * WN + cc * EPL;
  float bias[EPL];
#pragma unroll
  for (int r = 0; r < EPL; ++r) bias[r] = p.bias ? p.bias[n + r] : 0.f;
  uint4 wfr[NKT][2][TN];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wfr[kt][ks][i] = *(const uint4*)(p.w + ((size_t)(n0 + wx * WN + i * 16 + lr) * p.K + kt * 64 + ks * 32 + lg * 8) * ES);
  const uint32_t drop_seed = p.drop_thresh ? effective_seed(p.seed, p.seed_dev) : 0u;
  float* const stg = (float*)(sT + wave * (16 * WN * 4));

  // byte offset of this lane's 16-byte segment of row-iteration `it` of tile mt; rows past M / tiles past MT: out of range
  auto seg_off = [&](int mt, int it) -> uint32_t {
    const int m = mt * 64 + wy * WM + it * RPI + rsub;
    return (mt < MT && m < p.M) ? ((uint32_t)m * (uint32_t)d.ldc + (uint32_t)n) * ES : OOB;
  };
  u32x4_t res[2][NIT], msk[2][NIT];  // [register set][row iteration]
  auto fetch_ops = [&](int mt, int b) {  // inline asm: invisible to the compiler's wait-count pass, waited for by count below
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const uint32_t off = seg_off(mt, it);
      if constexpr (RES) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(res[b][it]) : "v"(off), "s"(rs_res) : "memory");
      if constexpr (MSK) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(msk[b][it]) : "v"(off), "s"(rs_msk) : "memory");
    }
  };

  // One step = tile mt in `cur`.  On entry: A(mt) and A(mt + G) are requested (in that order, into cur / nxt); with AHEAD the
  // operand rows of tile mt sit behind A(mt + G) in the queue, in register set B.
  auto step = [&](char* cur, char* nxt, int mt, auto B_) {
    constexpr int B = decltype(B_)::value;
    if constexpr (NOPS > 0 && !AHEAD) fetch_ops(mt, 0);
    // A(mt) has landed: loads younger than its last piece = A(mt + G) [NA], + the operand rows requested since
    wait_vmcnt<NA + NOPS>();
    __builtin_amdgcn_s_barrier();
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int cidx = ks * 4 + lg;
        uint4 af[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int row = wy * WM + j * 16 + lr;
          af[j] = *(const uint4*)(cur + kt * 8192 + row * 128 + ((cidx ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) Mfma<T>::run(wfr[kt][ks][i], af[j], acc[i][j]);
      }
    // every wavefront is done with the activation tile: its slot takes tile mt + 2G now, ahead of this tile's epilogue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_A(cur, mt + 2 * G);
    if constexpr (AHEAD) fetch_ops(mt + G, B ^ 1);
    // this tile's operand rows: younger loads = A(mt + 2G) [NA] (+ the rows of tile mt + G [NOPS]); without AHEAD they were
    // requested at the top of the step, ahead of nothing but A(mt + 2G)
    if constexpr (NOPS > 0) {
      wait_vmcnt<NA + (AHEAD ? NOPS : 0)>();
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int RB = AHEAD ? B : 0;
#pragma unroll
    for (int j = 0; j < TM; ++j) {  // 16 rows at a time through the wavefront's private staging region (no workgroup barrier)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int c = (i * 4 + lg) ^ (lr & (CPRW - 1));
        const f32x4 a = acc[i][j];
        *(float4*)(stg + lr * WN + c * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 0; h < 16 / RPI; ++h) {
        const int it = j * (16 / RPI) + h;
        const int row = h * RPI + rsub;
        const int sw = row & (CPRW - 1);
        float v[EPL];
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
          const float4 f = *(const float4*)(stg + row * WN + (((cc * (EPL / 4) + q) ^ sw) * 4));
          v[4 * q + 0] = f.x + bias[4 * q + 0]; v[4 * q + 1] = f.y + bias[4 * q + 1];
          v[4 * q + 2] = f.z + bias[4 * q + 2]; v[4 * q + 3] = f.w + bias[4 * q + 3];
        }
        const uint32_t off = seg_off(mt, it);
        if constexpr (RES) {
          float r8[EPL];
          unpack16<T>(make_uint4(res[RB][it].x, res[RB][it].y, res[RB][it].z, res[RB][it].w), r8);
#pragma unroll
          for (int r = 0; r < EPL; ++r) v[r] += r8[r];
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < EPL; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (MSK) {
          float m8[EPL];
          unpack16<T>(make_uint4(msk[RB][it].x, msk[RB][it].y, msk[RB][it].z, msk[RB][it].w), m8);
#pragma unroll
          for (int r = 0; r < EPL; ++r) v[r] = m8[r] > 0.f ? v[r] : 0.f;
        }
        if (p.drop_thresh) {  // (same element index as conv_gemm_kernel's epilogue and td_dropout: the backward regenerates this mask)
#pragma unroll
          for (int r = 0; r < EPL; ++r) v[r] = dropout_keep(drop_seed, off / ES + (uint32_t)r, p.drop_thresh) ? v[r] * p.drop_scale : 0.f;
        }
        const uint4 o = pack16<T>(v);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o.x, o.y, o.z, o.w}, rs_out, (int)off, 0, AUX_NT);  // rows past M: discarded
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads retired before the next 16 rows overwrite the region
    }
  };

  int mt = gid;
  issue_A(sA0, mt);
  issue_A(sA1, mt + G);
  if constexpr (AHEAD) fetch_ops(mt, 0);
  while (mt < MT) {
    step(sA0, sA1, mt, ic<0>{});
    mt += G;
    if (mt >= MT) break;
    step(sA1, sA0, mt, ic<1>{});
    mt += G;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (out-of-range) requests must not outlive the workgroup's LDS / registers
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][kk] += sum_{m in split} g[m][co] * gather(src)[m][kk]
// Both operands are reduction-major in memory ([m][channel]); LDS keeps them that way and the bf16
// fragments are produced by the gfx950 transposing LDS read (ds_read_b64_tr_b16).
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const char* g;
  const char* src;
  float* dw;
  const float* scale;  // out_mode 1/2: per-output-channel factor folded into the result (FrozenBN scale), may be null
  float* dbias;        // optional: dbias[co] += sum_m g[m][co] (column sums of the gradient), by the k-tile-0 workgroups
  td_conv_desc d;
  int M, K, ldg, mper;
  uint32_t g_bytes, src_bytes;
  int out_mode;  // 0: dw = [Nc][K] (K = R*S*C), fp32 atomics.  1: dw = the parameter's own [Nc][ci_real][R][S], atomics.
                 // 2: as 1 with plain stores (the job has a single split: nobody else touches the tile)
  int ci_real;   // input channels of the parameter (C may be padded)
  int tn, tk, first;  // batched launch: tile grid of this job and its first workgroup index
  int cls;            // wide-tile instance (conv_wgrad_wide_batch_kernel): bit 0 = 256 output channels per tile (else 128),
                      // bit 1 = 256 k columns per tile (else 128), bit 2 = pointwise
  unsigned long long* stamps;  // debug: 40 cycle stamps per workgroup (tools/stamp_wgrad.py)
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// LDS image of one stage: G tile [MK][128 channels] and X tile [MK][128 k] - reduction-major rows exactly as they
// lie in HBM, filled by buffer_load ... lds (lane-linear destination).  The transposing fragment reads
// (ds_read_b64_tr_b16) of a 32-lane group touch 8 different rows at one column: rows are a full bank period apart, so
// the 32-byte (bf16) / 64-byte (fp32) column blocks are XOR-swizzled by swz(row) - on the DMA source address and on
// the read, never on the DMA destination.
template <int ES>
__device__ __forceinline__ int wg_swz(int row) {
  return ES == 2 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (row & 7);
}

// NSTG = 2: two 32-KiB stages, one in flight, two workgroups per CU.  NSTG = 4: four stages (128 KiB of the CU's 160 KiB
// LDS, one workgroup per CU), three in flight behind counted s_waitcnt vmcnt - the launch has about one workgroup per
// CU anyway (fp32 atomics per output tile limit the split count), so bytes in flight per CU are what is left to raise.
template <typename T, int NSTG, bool PW>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int bx, const int by, const int bz, const int lin) {
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;
  constexpr int MK = 128 / ES;             // reduction rows per stage: 64 (bf16) / 32 (fp32)
  constexpr int ROWB = 128 * ES;           // bytes per row: 256 / 512
  constexpr int CPR = ROWB / 16;           // 16-byte chunks per row: 16 / 32
  constexpr int RPI = 64 / CPR;            // rows per wave DMA instruction: 4 / 2
  constexpr int LI = MK / (4 * RPI);       // DMA instructions per wave per operand per stage: 4 / 4
  constexpr int SWS = ES == 2 ? 1 : 2;     // log2(16-byte chunks per swizzle block): 32 B / 64 B blocks
  constexpr int TILEB = MK * ROWB;         // 16 KiB
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  __shared__ __attribute__((aligned(16))) char stage0[2 * TILEB];
  __shared__ __attribute__((aligned(16))) char stage1[2 * TILEB];
  __shared__ __attribute__((aligned(16))) char stage2[NSTG == 4 ? 2 * TILEB : 16];
  __shared__ __attribute__((aligned(16))) char stage3[NSTG == 4 ? 2 * TILEB : 16];

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const unsigned long long t_start = __builtin_readcyclecounter();
  const int co0 = bx * 128, kk0 = by * 128;
  const int mbeg = bz * p.mper;
  const int mend = min(p.M, mbeg + p.mper);
  if (mbeg >= mend) return;
  const int HoWo = d.Ho * d.Wo;
  constexpr bool pointwise = PW;  // 1x1, stride 1, no padding: im2col row == source row
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);

  // per-lane DMA bookkeeping: instruction i of this wave fills rows (i*4 + wave)*RPI + lane/CPR, LDS chunk lane%CPR
  int rloc[LI], gcol[LI], xr[LI], xs[LI], xc[LI];
  bool gok[LI], xok[LI];
#pragma unroll
  for (int i = 0; i < LI; ++i) {
    const int row = (i * 4 + wave) * RPI + lane / CPR;
    const int c16 = lane % CPR;
    const int blk = (c16 >> SWS) ^ wg_swz<ES>(row);                    // logical swizzle block stored at this slot
    const int col = ((blk << SWS) | (c16 & ((1 << SWS) - 1))) * VEC;   // logical column (elements) inside the tile
    rloc[i] = row;
    gcol[i] = co0 + col;
    gok[i] = gcol[i] < d.Nc;
    const int kk = kk0 + col;
    xok[i] = kk < p.K;
    xr[i] = 0; xs[i] = 0; xc[i] = kk;
    if (d.R * d.S > 1) {
      int tap = kk / d.C;
      xc[i] = kk - tap * d.C;
      xr[i] = tap / d.S;
      xs[i] = tap - xr[i] * d.S;
    }
  }
  // running (row, frame, ho, wo) of each DMA row of this lane; stages are issued strictly in order, MK rows apart, so the
  // im2col coordinates advance by a constant (dN, dH, dW) with one carry each - no divisions, no divergent branches in
  // the loop (the compiler can then interleave the address arithmetic with the MFMAs of the stage being consumed)
  int mcur[LI], rn[LI], rho[LI], rwo[LI];
  const int dN = MK / HoWo, dH = (MK - dN * HoWo) / d.Wo, dW = MK - dN * HoWo - dH * d.Wo;
#pragma unroll
  for (int i = 0; i < LI; ++i) {
    mcur[i] = mbeg + rloc[i];
    rn[i] = mcur[i] / HoWo;
    const int rem = mcur[i] - rn[i] * HoWo;
    rho[i] = rem / d.Wo;
    rwo[i] = rem - rho[i] * d.Wo;
  }
  auto issue_stage = [&](char* st) {
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      const int m = mcur[i];
      const bool ok = m < mend;
      const uint32_t og = (ok && gok[i]) ? ((uint32_t)m * (uint32_t)p.ldg + (uint32_t)gcol[i]) * ES : OOB;
      uint32_t ox;
      if constexpr (pointwise) {
        ox = (ok && xok[i]) ? ((uint32_t)m * (uint32_t)d.C + (uint32_t)xc[i]) * ES : OOB;
      } else {
        const int hs = rho[i] * d.stride - d.pad + xr[i], ws = rwo[i] * d.stride - d.pad + xs[i];
        const bool in = ok && xok[i] && (unsigned)hs < (unsigned)d.Hs && (unsigned)ws < (unsigned)d.Ws;
        ox = in ? ((uint32_t)((rn[i] * d.Hs + hs) * d.Ws + ws) * (uint32_t)d.C + (uint32_t)xc[i]) * ES : OOB;
        rwo[i] += dW;
        const int c1 = rwo[i] >= d.Wo;
        rwo[i] -= c1 ? d.Wo : 0;
        rho[i] += dH + c1;
        const int c2 = rho[i] >= d.Ho;
        rho[i] -= c2 ? d.Ho : 0;
        rn[i] += dN + c2;
      }
      mcur[i] += MK;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(st + (i * 4 + wave) * 1024), 16, og, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st + TILEB + (i * 4 + wave) * 1024), 16, ox, 0, 0, 0);
    }
  };

  const int wy = wave >> 1, wx = wave & 1;  // wy: co direction, wx: kk direction
  const int lr = lane & 15, lg = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient = column sums of g: one more MFMA per G fragment against an all-ones operand, only in the
  // workgroups of the first k tile (every column of the product holds the same sums)
  const bool do_bias = p.dbias != nullptr && by == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute_stage = [&](const char* st) {
    const char* sG = st;
    const char* sX = st + TILEB;
    if constexpr (ES == 2) {
      // all fragment reads of the stage are issued before the first MFMA: one wavefront per SIMD has nobody to hide the
      // LDS latency behind, so the second half's reads must fly while the first half multiplies
      constexpr int KS = MK / 32;
      const int jrow = lr >> 2, q = lr & 3;
      const int f = jrow | ((lg & 1) << 2);  // = wg_swz(r0) = wg_swz(r1)
      uint4 gf[KS][4], xf[KS][4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int r0 = ks * 32 + 8 * lg + jrow, r1 = r0 + 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cb = (((wy * 4 + i) ^ f) << 5) + q * 8;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sG + r0 * ROWB + cb));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sG + r1 * ROWB + cb));
          uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
          gf[ks][i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cb = (((wx * 4 + j) ^ f) << 5) + q * 8;
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sX + r0 * ROWB + cb));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sX + r1 * ROWB + cb));
          uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
          xf[ks][j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&gf[ks][i], *(const bf16x8*)&xf[ks][j], acc[i][j], 0, 0, 0);
      if (do_bias) {
        const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);  // 8 x bf16(1.0)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&gf[ks][i], *(const bf16x8*)&ones, accb[i], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < MK / 4; ++s4) {
        float gf[4], xf[4];
        const int krow = lg + 4 * s4;
        const int f = krow & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) gf[i] = *(const float*)(sG + krow * ROWB + (((wy * 4 + i) ^ f) << 6) + lr * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *(const float*)(sX + krow * ROWB + (((wx * 4 + j) ^ f) << 6) + lr * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[i], xf[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[i], 1.0f, accb[i], 0, 0, 0);
        }
      }
    }
  };

  const int nit = (mend - mbeg + MK - 1) / MK;
  unsigned long long* stp = p.stamps ? p.stamps + (size_t)lin * 40 : nullptr;
#define TD_WSTAMP(i) do { if (stp && t == 0) stp[i] = __builtin_readcyclecounter(); } while (0)
  if (stp && t == 0) stp[0] = t_start;
  TD_WSTAMP(1);
  if constexpr (NSTG == 2) {
    issue_stage(stage0);
    for (int it = 0; it < nit; it += 2) {
      __syncthreads();
      if (it + 1 < nit) issue_stage(stage1);
      compute_stage(stage0);
      if (it + 1 >= nit) break;
      __syncthreads();
      if (it + 2 < nit) issue_stage(stage0);
      compute_stage(stage1);
    }
  } else {
    // every step issues exactly one stage (rows past mend are all-OOB = zero fill, no traffic), so "the stage I am about
    // to read has landed" is always "at most two stages = 4*LI DMA instructions of this wave still outstanding"
#define TD_WG_STEP(cur, nxt, j)                                                        \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LI) : "memory");                      \
  __builtin_amdgcn_s_barrier();                                                       \
  issue_stage(nxt);                                                                   \
  if ((j) >= 0) { compute_stage(cur); if ((j) < 32) TD_WSTAMP(4 + (j)); }
    // software-pipeline warm-up folded into the loop (steps -3..-1 only issue): every stage buffer has exactly one
    // static DMA site, which keeps the compiler's own LDS-DMA wait counts exact
    for (int it = -3; it < nit; it += 4) {
      TD_WG_STEP(stage1, stage0, it)
      if (it + 1 >= nit) break;
      TD_WG_STEP(stage2, stage1, it + 1)
      if (it + 2 >= nit) break;
      TD_WG_STEP(stage3, stage2, it + 2)
      if (it + 3 >= nit) break;
      TD_WG_STEP(stage0, stage3, it + 3)
    }
#undef TD_WG_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill DMAs must not outlive the workgroup's LDS
  }
  // D[i=co][j=kk]: lane holds co = base + 4*lg + r, kk = base + lr
  TD_WSTAMP(2);
  if (do_bias && wx == 0 && lr == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int coo = co0 + wy * 64 + i * 16 + 4 * lg + rr;
        if (coo >= d.Nc) continue;
        if (p.out_mode == 2) p.dbias[coo] = accb[i][rr];  // single split: this workgroup owns the column sums of its channels
        else atomicAdd(p.dbias + coo, accb[i][rr]);
      }
  }
  if (p.out_mode == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kko = kk0 + wx * 64 + j * 16 + lr;
        if (kko >= p.K) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          int coo = co0 + wy * 64 + i * 16 + 4 * lg + rr;
          if (coo < d.Nc) atomicAdd(p.dw + (size_t)coo * p.K + kko, acc[i][j][rr]);
        }
      }
  } else {
    // straight into the parameter's [Nc][ci_real][R][S] layout with the FrozenBN scale folded (what wgrad_finalize did)
    const int RS = d.R * d.S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kko = kk0 + wx * 64 + j * 16 + lr;
      if (kko >= p.K) continue;
      const int tap = kko / d.C, ci = kko - tap * d.C;
      if (ci >= p.ci_real) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int coo = co0 + wy * 64 + i * 16 + 4 * lg + rr;
          if (coo >= d.Nc) continue;
          const float v = acc[i][j][rr] * (p.scale ? p.scale[coo] : 1.f);
          float* dst = p.dw + ((size_t)coo * p.ci_real + ci) * RS + tap;
          if (p.out_mode == 2) *dst = v;
          else atomicAdd(dst, v);
        }
    }
  }
  TD_WSTAMP(3);
#undef TD_WSTAMP
}

template <typename T, int NSTG, bool PW>
__global__ __launch_bounds__(256, NSTG == 2 ? 2 : 1) void conv_wgrad_kernel(WgradParams p) {
  wgrad_body<T, NSTG, PW>(p, blockIdx.x, blockIdx.y, blockIdx.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// Batched form: one launch covers the weight gradients of many layers.  With every trainable conv of the trunk in one
// launch there are thousands of tiles, so a job needs no (or few) splits: no atomics, no accumulator memset, no separate
// finalize pass - and no per-layer tail where 256 CUs wait on the slowest workgroup.
// Workgroup w runs on XCD w % 8 (round-robin dispatch): each job is owned by ONE XCD (host-side longest-first
// balancing), so the 32 CUs that share an L2 stream the same gradient / activation rows at about the same time and every
// operand row is fetched into that L2 once instead of once per tile.  Within its XCD a workgroup finds its job by
// binary search over the job's first slot.
struct WgradXcdIndex {
  int start[9];  // job table range of XCD x: [start[x], start[x+1])
  int slots[8];  // work items of XCD x
};

template <typename T, int NSTG, bool PW>
__global__ __launch_bounds__(256, NSTG == 2 ? 2 : 1) void conv_wgrad_batch_kernel(const WgradParams* __restrict__ jobs, WgradXcdIndex xi) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= xi.slots[xcd]) return;
  int lo = xi.start[xcd], hi = xi.start[xcd + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= slot) lo = mid;
    else hi = mid - 1;
  }
  const WgradParams p = jobs[lo];
  int local = slot - p.first;
  const int bx = local % p.tn;
  local /= p.tn;
  const int by = local % p.tk;
  wgrad_body<T, NSTG, PW>(p, bx, by, local / p.tk, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Wide-tile weight gradients (bf16, batched launches, large M): the 128 x 128 instance above moves 32 KiB HBM/L2 -> LDS per
// 128 MFMAs with ONE wavefront per SIMD - 0.17 of the MFMA peak measured over the trunk's 93 jobs.  Here a workgroup of
// SIXTEEN wavefronts (four per SIMD: the LDS latency of one is covered by the MFMAs of the others, no fragment double
// buffering, 128 registers each) owns a 256 (output channels) x 256 (k) tile of dW - or 128 x 256 / 256 x 128 for the
// 128-wide layers - and walks its slice of the reduction in stages of 64 rows: 64 KiB per 512 MFMAs, the gradient rows of a
// 256-channel layer read once per k tile.  LDS image: each 128-column sub-tile exactly as in the instance above
// (reduction-major rows as they lie in HBM, swizzled 32-byte blocks, ds_read_b64_tr_b16 fragment reads); one DMA
// instruction per wavefront, sub-tile and stage.  The barrier that publishes stage s + 1 sits between the fragment reads and
// the MFMAs of the last k-step of stage s, so its skew is covered by 16 MFMAs per wavefront.  Output straight into the
// parameter's [Nc][ci][R][S] layout with the FrozenBN scale folded (plain stores for unsplit jobs, fp32 atomics otherwise).
// Nc and C are multiples of 128 (host-checked): a sub-tile is inside the matrix or outside as a whole and lies within ONE
// filter tap - validity, tap and channel base are wave-uniform scalars, the lane only adds its column.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NW = 8 (round 6, TD_WGRAD_WIDE8): the same tile on EIGHT wavefronts, two per SIMD - wave tiles of 128 x 64 (256-channel layers) instead
// of 64 x 64: 12 instead of 16 transposing fragment reads per 32 MFMAs, two DMA instructions per wavefront, sub-tile and stage.
template <int GS, int XS, bool PW, int NW = 16>
__device__ __forceinline__ void wgrad_wide_body(const WgradParams& p, const int bx, const int by, const int bz, char* st0, char* st1) {
  constexpr int ES = 2, MK = 64, ROWB = 256, SUB = MK * ROWB;  // one 128-column sub-tile = 16 KiB
  static_assert(NW == 16 || NW == 8, "four or two wavefronts per SIMD");
  constexpr int RPW = 16 / NW;                 // DMA instructions (4 rows each) per wavefront, sub-tile and stage
  constexpr int WVK = XS * 2, WVC = NW / WVK;  // wavefronts along k / along the output channels
  constexpr int WCO = GS * 128 / WVC;
  constexpr int FI = WCO / 16, FJ = 4;
  static_assert(FI >= 1 && WCO <= 128, "wave tile");
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int co0 = bx * (GS * 128), kk0 = by * (XS * 128);
  const int mbeg = bz * p.mper;
  const int mend = min(p.M, mbeg + p.mper);
  if (mbeg >= mend) return;
  const int HoWo = d.Ho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);

  // DMA bookkeeping: this wavefront's instruction h fills rows (wave * RPW + h) * 4 + lane/16 of every sub-tile, LDS chunk lane%16
  // (the swizzle of a row depends on its bits 0, 1 and 3: the same for both instructions of a wavefront)
  const int drow = wave * (4 * RPW) + (lane >> 4);
  const int c16 = lane & 15;
  const int col = ((((c16 >> 1) ^ wg_swz<ES>(drow)) << 1) | (c16 & 1)) * 8;  // logical column of this lane's chunk
  int mcur = mbeg + drow;
  int rn[RPW], rho[RPW], rwo[RPW];
#pragma unroll
  for (int h = 0; h < RPW; ++h) {
    const int m_ = mcur + 4 * h;
    rn[h] = m_ / HoWo;
    rho[h] = (m_ - rn[h] * HoWo) / d.Wo;
    rwo[h] = m_ - rn[h] * HoWo - rho[h] * d.Wo;
  }
  bool g_in[GS], x_in[XS];
  int g_c0[GS], x_c0[XS], x_r[XS], x_s[XS];
#pragma unroll
  for (int s_ = 0; s_ < GS; ++s_) {
    g_c0[s_] = co0 + s_ * 128;
    g_in[s_] = g_c0[s_] < d.Nc;
  }
#pragma unroll
  for (int s_ = 0; s_ < XS; ++s_) {
    const int kk = kk0 + s_ * 128;
    x_in[s_] = kk < p.K;
    x_r[s_] = 0; x_s[s_] = 0; x_c0[s_] = kk;
    if (!PW && d.R * d.S > 1) {
      const int tap = kk / d.C;
      x_c0[s_] = kk - tap * d.C;
      x_r[s_] = tap / d.S;
      x_s[s_] = tap - x_r[s_] * d.S;
    }
  }
  const int dN = MK / HoWo, dH = (MK - dN * HoWo) / d.Wo, dW = MK - dN * HoWo - dH * d.Wo;
  auto issue_stage = [&](char* st) {
#pragma unroll
    for (int h = 0; h < RPW; ++h) {
      const int m = mcur + 4 * h;
      const bool ok = m < mend;
      const uint32_t grow = ((uint32_t)m * (uint32_t)p.ldg + (uint32_t)col) * ES;
#pragma unroll
      for (int s_ = 0; s_ < GS; ++s_) {
        const uint32_t og = (ok && g_in[s_]) ? grow + (uint32_t)g_c0[s_] * ES : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(st + s_ * SUB + (wave * RPW + h) * 1024), 16, og, 0, 0, 0);
      }
#pragma unroll
      for (int s_ = 0; s_ < XS; ++s_) {
        uint32_t ox;
        if constexpr (PW) {
          ox = (ok && x_in[s_]) ? ((uint32_t)m * (uint32_t)d.C + (uint32_t)(x_c0[s_] + col)) * ES : OOB;
        } else {
          const int hs = rho[h] * d.stride - d.pad + x_r[s_], ws = rwo[h] * d.stride - d.pad + x_s[s_];
          const bool in = ok && x_in[s_] && (unsigned)hs < (unsigned)d.Hs && (unsigned)ws < (unsigned)d.Ws;
          ox = in ? ((uint32_t)((rn[h] * d.Hs + hs) * d.Ws + ws) * (uint32_t)d.C + (uint32_t)(x_c0[s_] + col)) * ES : OOB;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st + (GS + s_) * SUB + (wave * RPW + h) * 1024), 16, ox, 0, 0, 0);
      }
      if constexpr (!PW) {
        rwo[h] += dW;
        const int c1 = rwo[h] >= d.Wo;
        rwo[h] -= c1 ? d.Wo : 0;
        rho[h] += dH + c1;
        const int c2 = rho[h] >= d.Ho;
        rho[h] -= c2 ? d.Ho : 0;
        rn[h] += dN + c2;
      }
    }
    mcur += MK;
  };

  const int wyc = wave / WVK, wxk = wave - wyc * WVK;
  const int lr = lane & 15, lg = lane >> 4;
  const int jrow = lr >> 2, q = lr & 3;
  int f = jrow | ((lg & 1) << 2);  // = wg_swz(r0) = wg_swz(r1)
  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 gf[FI], xf[FJ];
  auto frag = [&](const char* sub, int ks, int blk16) -> u32x4 {
    const int r0 = ks * 32 + 8 * lg + jrow;
    const int cb = ((blk16 ^ f) << 5) + q * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sub + r0 * ROWB + cb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sub + (r0 + 4) * ROWB + cb));
    const uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
    return u32x4{l2.x, l2.y, h2.x, h2.y};
  };
  auto load_frags = [&](const char* st, int ks) {
    const int xoff = wxk * 64, goff = wyc * WCO;  // both inside one sub-tile (WCO <= 64 or 128-aligned)
    // the swizzle operand is laundered per call: the 2 x (FI + FJ) fragment addresses are then recomputed where they are
    // used (an xor and a shift-add in the shadow of the MFMAs) instead of living in 16 registers across the loop
    asm volatile("" : "+v"(f));
#pragma unroll
    for (int i = 0; i < FI; ++i) gf[i] = frag(st + ((goff + i * 16) / 128) * SUB, ks, ((goff + i * 16) % 128) / 16);
#pragma unroll
    for (int j = 0; j < FJ; ++j) xf[j] = frag(st + (GS + xoff / 128) * SUB, ks, (xoff % 128) / 16 + j);
  };
  // The accumulator operand is TIED (inline asm "+v"): with the builtin the register allocator renames the accumulators
  // through the two-stage loop body (destination != source C on most MFMAs), which doubles their footprint and spills.
  // Hazards the compiler cannot see inside the asm: an accumulator is touched again 15 MFMAs (>= 240 cycles) later, and the
  // epilogue below waits explicitly before its first VALU read.
  auto mfmas = [&]() {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(gf[i]), "v"(xf[j]));
  };
#if TD_WGRAD_EARLY_ISSUE
  // Round 6: the refill of a stage buffer is issued the moment the buffer is free - behind the barrier that follows its last fragment
  // read, i.e. in the middle of the stage that consumed it, for the stage AFTER the next one - and waited for one whole stage later.
  // (Rounds 4 - 5 issued it at the head of the next stage and waited for it in that stage's middle: half a stage, ~0.5 us, of lead for a
  // load that takes longer than that under load.)
  auto stage_body = [&](char* cur, char* nxt) {
    (void)nxt;
    load_frags(cur, 0);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist the second k-step's reads: 2 x 32 fragment registers)
    mfmas();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 1);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): own DMA of the next stage (issued a stage ago) landed, own reads of `cur` returned
    __builtin_amdgcn_s_barrier();
    issue_stage(cur);  // every wavefront has read its last fragment of `cur`: refill it with the rows of the stage after next (past the slice: all-OOB, no traffic)
    mfmas();
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto stage_body = [&](char* cur, char* nxt) {
    issue_stage(nxt);  // every wavefront finished reading `nxt` before the last barrier; past the slice: all-OOB, no traffic
    load_frags(cur, 0);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist the second k-step's reads: 2 x 32 fragment registers)
    mfmas();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 1);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): own DMA of the next stage landed, own reads of `cur` returned
    __builtin_amdgcn_s_barrier();
    mfmas();
    __builtin_amdgcn_sched_barrier(0);
  };
#endif

  // ONE loop over stage pairs and nothing else: an odd stage count is rounded up (rows past the slice are zero-filled
  // without traffic).  A separate tail would bring register spills of the accumulators, and a spill store right behind an
  // inline-asm MFMA lacks the MFMA -> VMEM wait states the compiler adds for MFMAs it knows about.
  const int npair = ((mend - mbeg + MK - 1) / MK + 1) / 2;
  issue_stage(st0);
#if TD_WGRAD_EARLY_ISSUE
  issue_stage(st1);
#endif
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < npair; ++it) {
    stage_body(st0, st1);
    stage_body(st1, st0);
  }
#if TD_WGRAD_EARLY_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two trailing (all-OOB) refills must not outlive the workgroup's LDS
#endif
  // last MFMA results -> first read: 2 x 16 idle cycles, and every accumulator passes through an asm "modification" placed
  // behind them, so no compiler-generated use (VALU read, spill store) of an accumulator can be scheduled before the wait
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) asm volatile("" : "+v"(acc[i][j]));
  // D[i = co][j = kk]: lane holds co = base + 4*lg + rr, kk = base + lr
  const int RS = d.R * d.S;
#pragma unroll
  for (int j = 0; j < FJ; ++j) {
    const int kko = kk0 + wxk * 64 + j * 16 + lr;
    if (kko >= p.K) continue;
    int tap = 0, ci = kko;
    if (!PW && RS > 1) {
      tap = kko / d.C;
      ci = kko - tap * d.C;
    }
    if (ci >= p.ci_real) continue;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int coo = co0 + wyc * WCO + i * 16 + 4 * lg + rr;
        if (coo >= d.Nc) continue;
        const float v = acc[i][j][rr] * (p.scale ? p.scale[coo] : 1.f);
        float* dst = p.dw + ((size_t)coo * p.ci_real + ci) * RS + tap;
        if (p.out_mode == 2) *dst = v;
        else atomicAdd(dst, v);
      }
  }
}

__global__ __launch_bounds__(1024, 1) void conv_wgrad_wide_batch_kernel(const WgradParams* __restrict__ jobs, WgradXcdIndex xi) {
  __shared__ __attribute__((aligned(16))) char wst0[65536];
  __shared__ __attribute__((aligned(16))) char wst1[65536];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= xi.slots[xcd]) return;
  int lo = xi.start[xcd], hi = xi.start[xcd + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= slot) lo = mid;
    else hi = mid - 1;
  }
  const WgradParams p = jobs[lo];
  int local = slot - p.first;
  const int bx = local % p.tn;
  local /= p.tn;
  const int by = local % p.tk, bz = local / p.tk;
  switch (p.cls) {
    case 3: wgrad_wide_body<2, 2, false>(p, bx, by, bz, wst0, wst1); break;
    case 7: wgrad_wide_body<2, 2, true>(p, bx, by, bz, wst0, wst1); break;
    case 2: wgrad_wide_body<1, 2, false>(p, bx, by, bz, wst0, wst1); break;
    case 6: wgrad_wide_body<1, 2, true>(p, bx, by, bz, wst0, wst1); break;
    case 1: wgrad_wide_body<2, 1, false>(p, bx, by, bz, wst0, wst1); break;
    default: wgrad_wide_body<2, 1, true>(p, bx, by, bz, wst0, wst1); break;
  }
}

// the same launch on eight wavefronts per workgroup (TD_WGRAD_WIDE8=1; see wgrad_wide_body)
__global__ __launch_bounds__(512, 1) void conv_wgrad_wide8_batch_kernel(const WgradParams* __restrict__ jobs, WgradXcdIndex xi) {
  __shared__ __attribute__((aligned(16))) char wst0[65536];
  __shared__ __attribute__((aligned(16))) char wst1[65536];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= xi.slots[xcd]) return;
  int lo = xi.start[xcd], hi = xi.start[xcd + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= slot) lo = mid;
    else hi = mid - 1;
  }
  const WgradParams p = jobs[lo];
  int local = slot - p.first;
  const int bx = local % p.tn;
  local /= p.tn;
  const int by = local % p.tk, bz = local / p.tk;
  switch (p.cls) {
    case 3: wgrad_wide_body<2, 2, false, 8>(p, bx, by, bz, wst0, wst1); break;
    case 7: wgrad_wide_body<2, 2, true, 8>(p, bx, by, bz, wst0, wst1); break;
    case 2: wgrad_wide_body<1, 2, false, 8>(p, bx, by, bz, wst0, wst1); break;
    case 6: wgrad_wide_body<1, 2, true, 8>(p, bx, by, bz, wst0, wst1); break;
    case 1: wgrad_wide_body<2, 1, false, 8>(p, bx, by, bz, wst0, wst1); break;
    default: wgrad_wide_body<2, 1, true, 8>(p, bx, by, bz, wst0, wst1); break;
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 weight-gradient tiles on FOUR wavefronts with 128 x 128 wave tiles (round 6; the 256-channel / K >= 256 jobs of the batched launch:
// most of the trunk's weight-gradient FLOPs).  Why: both operands of dW = g^T x are reduction-major in HBM, so every MFMA fragment is a pair
// of transposing LDS reads (ds_read_b64_tr_b16), and those sustain 85 - 119 B/clk per CU here (round 4).  A 64 x 64 wave tile (the sixteen-
// wavefront body above) needs 16 such reads per 16 MFMAs = 128 B/clk at the full matrix rate - the kernel sat at mfma_util 0.39.  A 128 x 128
// wave tile reads 16 fragments (32 reads) per 64 MFMAs: 64 B/clk.  Price: one wavefront per SIMD (256 accumulator AGPRs, allocated by hand as in
// chain.hip), so the next stage's fragments are read UNDER this stage's MFMAs into a second register set - two fragments behind every eight MFMAs.
// Stages are 32 reduction rows (one K-step; 4 sub-tiles x 8 KiB) on a ring of four: the barrier at the top of stage s publishes stage s + 1
// (whose fragments are read during stage s) and frees the buffer of stage s - 1 for stage s + 3.  Same LDS image per sub-tile and the same
// summation order per output element as the sixteen-wavefront body: results are bit-identical to it for unsplit jobs.
template <bool PW>
__device__ __forceinline__ void wgrad_wide4_body(const WgradParams& p, const int bx, const int by, const int bz, char* r0_, char* r1_, char* r2_, char* r3_) {
  constexpr int ES = 2, MK = 32, ROWB = 256, SUB = MK * ROWB;  // one 128-column sub-tile of a stage = 8 KiB; a stage = g0 g1 x0 x1
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int co0 = bx * 256, kk0 = by * 256;
  const int mbeg = bz * p.mper;
  const int mend = min(p.M, mbeg + p.mper);
  if (mbeg >= mend) return;
  const int HoWo = d.Ho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  asm volatile("" ::: "a255");  // the whole accumulator file is this kernel's: acc(i, j) = a[4 (8 i + j) .. + 4], i = output-channel fragment, j = k fragment

  // DMA bookkeeping: piece e (0, 1) of this wavefront fills rows (e * 4 + wave) * 4 + lane / 16 of every sub-tile, LDS chunk lane % 16
  const int c16 = lane & 15;
  int mcur[2], rn[2], rho[2], rwo[2], col[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int drow = (e * 4 + wave) * 4 + (lane >> 4);
    col[e] = ((((c16 >> 1) ^ wg_swz<ES>(drow)) << 1) | (c16 & 1)) * 8;  // logical column of this lane's chunk
    mcur[e] = mbeg + drow;
    rn[e] = mcur[e] / HoWo;
    rho[e] = (mcur[e] - rn[e] * HoWo) / d.Wo;
    rwo[e] = mcur[e] - rn[e] * HoWo - rho[e] * d.Wo;
  }
  bool g_in[2], x_in[2];
  int g_c0[2], x_c0[2], x_r[2], x_s[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    g_c0[s_] = co0 + s_ * 128;
    g_in[s_] = g_c0[s_] < d.Nc;
    const int kk = kk0 + s_ * 128;
    x_in[s_] = kk < p.K;
    x_r[s_] = 0; x_s[s_] = 0; x_c0[s_] = kk;
    if (!PW && d.R * d.S > 1) {
      const int tap = kk / d.C;
      x_c0[s_] = kk - tap * d.C;
      x_r[s_] = tap / d.S;
      x_s[s_] = tap - x_r[s_] * d.S;
    }
  }
  const int dN = MK / HoWo, dH = (MK - dN * HoWo) / d.Wo, dW = MK - dN * HoWo - dH * d.Wo;
  // 8 pieces per stage, BRANCH-FREE (all-ones / all-zeros masks, no bool select the compiler could turn into divergent control flow: a piece
  // issued once per side of a branch would break the counted waits and leave LDS slots unwritten).  Rows past the slice, columns past the
  // matrix and out-of-image taps: out-of-range offset = zero fill, no traffic.
  auto neg = [](int v) -> uint32_t { return (uint32_t)(v >> 31); };  // all ones if v < 0
  uint32_t gm[2], xm[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    gm[s_] = g_in[s_] ? 0xFFFFFFFFu : 0u;
    xm[s_] = x_in[s_] ? 0xFFFFFFFFu : 0u;
  }
  auto issue_stage = [&](char* st) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int m = mcur[e];
      const uint32_t okm = neg(m - mend);  // m < mend
      const uint32_t grow = ((uint32_t)m * (uint32_t)p.ldg + (uint32_t)col[e]) * ES;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const uint32_t k_ = okm & gm[s_];
        const uint32_t og = ((grow + (uint32_t)g_c0[s_] * ES) & k_) | (OOB & ~k_);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(st + s_ * SUB + (e * 4 + wave) * 1024), 16, og, 0, 0, 0);
      }
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        uint32_t ox, k_;
        if constexpr (PW) {
          k_ = okm & xm[s_];
          ox = ((uint32_t)m * (uint32_t)d.C + (uint32_t)(x_c0[s_] + col[e])) * ES;
        } else {
          const int hs = rho[e] * d.stride - d.pad + x_r[s_], ws = rwo[e] * d.stride - d.pad + x_s[s_];
          k_ = okm & xm[s_] & ~neg(hs | (d.Hs - 1 - hs) | ws | (d.Ws - 1 - ws));  // 0 <= hs < Hs and 0 <= ws < Ws
          ox = ((uint32_t)((rn[e] * d.Hs + hs) * d.Ws + ws) * (uint32_t)d.C + (uint32_t)(x_c0[s_] + col[e])) * ES;
        }
        ox = (ox & k_) | (OOB & ~k_);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st + (2 + s_) * SUB + (e * 4 + wave) * 1024), 16, ox, 0, 0, 0);
      }
      if constexpr (!PW) {
        rwo[e] += dW;
        const int c1 = (int)(~neg(rwo[e] - d.Wo) & 1u);  // rwo >= Wo
        rwo[e] -= c1 * d.Wo;
        rho[e] += dH + c1;
        const int c2 = (int)(~neg(rho[e] - d.Ho) & 1u);
        rho[e] -= c2 * d.Ho;
        rn[e] += dN + c2;
      }
      mcur[e] += MK;
    }
  };

  const int wyc = wave >> 1, wxk = wave & 1;  // this wavefront's 128 output channels / 128 k columns = sub-tile wyc of g, sub-tile wxk of x
  const int lr = lane & 15, lg = lane >> 4;
  const int jrow = lr >> 2, q = lr & 3;
  const int f = jrow | ((lg & 1) << 2);  // = wg_swz(r0) = wg_swz(r0 + 4)
  const int rowoff = (8 * lg + jrow) * ROWB + q * 8;
  // fragment F (0..7 = g blocks, 8..15 = x blocks) of the stage in buffer `st`: two transposing reads (rows r0, r0 + 4)
  auto frag = [&](const char* st, int F) -> u32x4 {
    const char* sub = st + (F < 8 ? wyc : 2 + wxk) * SUB;
    const int cb = (((F & 7) ^ f) << 5);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sub + rowoff + cb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(sub + rowoff + 4 * ROWB + cb));
    const uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
    return u32x4{l2.x, l2.y, h2.x, h2.y};
  };
  u32x4 fa[16], fb[16];  // the two fragment sets: [0..7] g (output channels), [8..15] x (k columns)

  // one stage: 64 MFMAs on set `cur`; the next stage's 16 fragments are read into set `nxt` two at a time behind every eight MFMAs
#define TD_W4_MFMA(I, Jx, cur) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(4 * (8 * (I) + (Jx))), "n"(4 * (8 * (I) + (Jx)) + 3), "v"(cur[I]), "v"(cur[8 + (Jx)]));
#define TD_W4_ROWM(I, cur) TD_W4_MFMA(I, 0, cur) TD_W4_MFMA(I, 1, cur) TD_W4_MFMA(I, 2, cur) TD_W4_MFMA(I, 3, cur) TD_W4_MFMA(I, 4, cur) TD_W4_MFMA(I, 5, cur) TD_W4_MFMA(I, 6, cur) TD_W4_MFMA(I, 7, cur)
// (the x fragments - needed by EVERY row of the next stage - are read first, under rows 0..3; g fragment I, needed by row I, under rows 4..7;
//  the pieces of stage s + 3 are issued behind row 3: the top of a stage is a wait and a barrier only)
#define TD_W4_STAGE(cur, nxt, nbuf, buf_sp3)                                                            \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                      \
  __builtin_amdgcn_s_barrier();                                                                         \
  TD_W4_ROWM(0, cur) nxt[8] = frag(nbuf, 8);   nxt[9] = frag(nbuf, 9);   __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(1, cur) nxt[10] = frag(nbuf, 10); nxt[11] = frag(nbuf, 11); __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(2, cur) nxt[12] = frag(nbuf, 12); nxt[13] = frag(nbuf, 13); __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(3, cur) nxt[14] = frag(nbuf, 14); nxt[15] = frag(nbuf, 15); __builtin_amdgcn_sched_barrier(0); \
  issue_stage(buf_sp3);                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  TD_W4_ROWM(4, cur) nxt[0] = frag(nbuf, 0);  nxt[1] = frag(nbuf, 1);   __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(5, cur) nxt[2] = frag(nbuf, 2);  nxt[3] = frag(nbuf, 3);   __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(6, cur) nxt[4] = frag(nbuf, 4);  nxt[5] = frag(nbuf, 5);   __builtin_amdgcn_sched_barrier(0); \
  TD_W4_ROWM(7, cur) nxt[6] = frag(nbuf, 6);  nxt[7] = frag(nbuf, 7);   __builtin_amdgcn_sched_barrier(0);
  // top of stage s: this wavefront's pieces of stage s + 1 have landed (behind them: the 8 pieces of stage s + 2), then everybody's; the
  // buffer of stage s - 1 (read during stage s - 2, consumed in stage s - 1) takes stage s + 3 from the middle of the stage on

  // zero the accumulator file (64 fragments): C = 0 forms would need a first-stage copy of the whole stage macro
#define TD_W4_Z(N) asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\tv_accvgpr_write_b32 a%c3, 0" ::"n"(4 * (N)), "n"(4 * (N) + 1), "n"(4 * (N) + 2), "n"(4 * (N) + 3));
#define TD_W4_Z8(B) TD_W4_Z(B) TD_W4_Z(B + 1) TD_W4_Z(B + 2) TD_W4_Z(B + 3) TD_W4_Z(B + 4) TD_W4_Z(B + 5) TD_W4_Z(B + 6) TD_W4_Z(B + 7)
  TD_W4_Z8(0) TD_W4_Z8(8) TD_W4_Z8(16) TD_W4_Z8(24) TD_W4_Z8(32) TD_W4_Z8(40) TD_W4_Z8(48) TD_W4_Z8(56)
#undef TD_W4_Z8
#undef TD_W4_Z

  const int nst = (mend - mbeg + MK - 1) / MK;
  const int nquad = (nst + 3) / 4;  // ONE loop over stage quadruples and nothing else (extra stages: zero fill without traffic)
  issue_stage(r0_);
  issue_stage(r1_);
  issue_stage(r2_);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 has landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int F = 0; F < 16; ++F) fa[F] = frag(r0_, F);
#pragma unroll 1
  for (int it = 0; it < nquad; ++it) {
    TD_W4_STAGE(fa, fb, r1_, r3_)
    TD_W4_STAGE(fb, fa, r2_, r0_)
    TD_W4_STAGE(fa, fb, r3_, r1_)
    TD_W4_STAGE(fb, fa, r0_, r2_)
  }
#undef TD_W4_STAGE
#undef TD_W4_ROWM
#undef TD_W4_MFMA
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero-fill DMAs / fragment reads must not outlive the workgroup's LDS
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the asm MFMAs' results are read below
  // D[i = co][j = kk]: lane holds co = base + 4 lg + rr, kk = base + lr
  const int RS = d.R * d.S;
  // (a wide job has Nc % 256 == 0 - host-checked - so every output channel of the tile exists; k columns past K and padded input channels do not)
  float sc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sc[i][rr] = p.scale ? p.scale[co0 + wyc * 128 + i * 16 + 4 * lg + rr] : 1.f;
  const bool plain = p.out_mode == 2;  // (uniform) a single split: this workgroup owns the tile
#define TD_W4_OUT(I, Jx)                                                                                                                     \
  if (jok[Jx]) {                                                                                                                            \
    float v_[4];                                                                                                                            \
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"     \
                 : "=v"(v_[0]), "=v"(v_[1]), "=v"(v_[2]), "=v"(v_[3])                                                                        \
                 : "n"(4 * (8 * (I) + (Jx))), "n"(4 * (8 * (I) + (Jx)) + 1), "n"(4 * (8 * (I) + (Jx)) + 2), "n"(4 * (8 * (I) + (Jx)) + 3));      \
    float* dst = jdst[Jx] + (size_t)((I) * 16) * cstride;                                                                                   \
    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                                                      \
      const float v = v_[rr] * sc[I][rr];                                                                                                   \
      if (plain) dst[(size_t)rr * cstride] = v;                                                                                             \
      else atomicAdd(dst + (size_t)rr * cstride, v);                                                                                        \
    }                                                                                                                                       \
  }
  const size_t cstride = (size_t)p.ci_real * RS;  // elements between consecutive output channels of dW [Nc][ci_real][R][S]
  bool jok[8];
  float* jdst[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kko = kk0 + wxk * 128 + j * 16 + lr;
    int tap = 0, ci = kko;
    if (!PW && RS > 1) {
      tap = kko / d.C;
      ci = kko - tap * d.C;
    }
    jok[j] = kko < p.K && ci < p.ci_real;
    jdst[j] = p.dw + ((size_t)(co0 + wyc * 128 + 4 * lg) * p.ci_real + ci) * RS + tap;
  }
#define TD_W4_OUTROW(I) TD_W4_OUT(I, 0) TD_W4_OUT(I, 1) TD_W4_OUT(I, 2) TD_W4_OUT(I, 3) TD_W4_OUT(I, 4) TD_W4_OUT(I, 5) TD_W4_OUT(I, 6) TD_W4_OUT(I, 7)
  TD_W4_OUTROW(0) TD_W4_OUTROW(1) TD_W4_OUTROW(2) TD_W4_OUTROW(3) TD_W4_OUTROW(4) TD_W4_OUTROW(5) TD_W4_OUTROW(6) TD_W4_OUTROW(7)
#undef TD_W4_OUTROW
#undef TD_W4_OUT
}

__global__ __launch_bounds__(256, 1) void conv_wgrad_wide4_batch_kernel(const WgradParams* __restrict__ jobs, WgradXcdIndex xi) {
  // four 32-KiB stage buffers as four LDS objects: disjoint alias scopes, so the fragment reads of one stage do not wait for the LDS-DMA filling another
  __shared__ __attribute__((aligned(1024))) char w4s0[32768];
  __shared__ __attribute__((aligned(1024))) char w4s1[32768];
  __shared__ __attribute__((aligned(1024))) char w4s2[32768];
  __shared__ __attribute__((aligned(1024))) char w4s3[32768];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= xi.slots[xcd]) return;
  int lo = xi.start[xcd], hi = xi.start[xcd + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first <= slot) lo = mid;
    else hi = mid - 1;
  }
  const WgradParams p = jobs[lo];
  int local = slot - p.first;
  const int bx = local % p.tn;
  local /= p.tn;
  const int by = local % p.tk, bz = local / p.tk;
  if (p.cls & 4) wgrad_wide4_body<true>(p, bx, by, bz, w4s0, w4s1, w4s2, w4s3);
  else wgrad_wide4_body<false>(p, bx, by, bz, w4s0, w4s1, w4s2, w4s3);
}

static int validate(const td_conv_desc* d, int dtype, const char* who) {
  const int vec = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "%s: bad dtype %d", who, dtype);
  TD_REQUIRE(d->C % vec == 0, "%s: source channels C=%d must be a multiple of %d (pad them)", who, d->C, vec);
  TD_REQUIRE(d->N > 0 && d->Hs > 0 && d->Ws > 0 && d->Ho > 0 && d->Wo > 0 && d->R > 0 && d->S > 0 && d->stride > 0,
             "%s: bad geometry", who);
  TD_REQUIRE((double)d->N * d->Hs * d->Ws * d->C < 2147483647.0, "%s: source tensor exceeds 2^31 elements", who);
  TD_REQUIRE(!d->aniso || (d->mode == 0 && d->stride_w >= 1 && d->pad_w >= 0 && d->out_sp <= 1), "%s: anisotropic stride / padding is a forward-only geometry", who);
  return TD_OK;
}

}  // namespace td

using namespace td;

static thread_local unsigned long long* g_dbg = nullptr;  // debug hook (tools/stamp_*.py): per calling thread
static int persist_min_tiles() {
  static const int v = [] { const char* e = getenv("TD_PW_PERSIST_MIN"); return e ? atoi(e) : 4; }();
  return v;
}
extern "C" int td_debug_set_stamp_buffer(unsigned long long* buf) { g_dbg = buf; return TD_OK; }

// a launch that uses a subset of the column blocks (filter taps) of a larger weight matrix
struct WeightSubset {
  int w_ld;          // row stride of the matrix in elements
  int ntap;          // taps of this launch, in its own (r', s') order
  int wtap[4];       // their tap indices in the matrix
  size_t w_bytes;    // addressable bytes from the pointer handed in
};
static int conv_gemm_launch(const void* src, const void* wmat, void* out, const td_conv_desc* d, const td_epilogue* e, int dtype,
                            td_stream_t stream, const WeightSubset* sub);

// Input gradient of a 3x3 / stride 2 / pad 1 convolution (the second conv of a stage's first block), by output parity.  dx pixel
// (2a + ph, 2b + pw) only receives taps r with ph + 1 - r even: 1 tap per axis for an even coordinate, 2 for an odd one - 9 taps
// over 4 pixels.  Walked as ONE gather over all 9 taps per output pixel (the generic input-gradient path) three quarters of the
// MFMAs multiply zero rows: 4.0 ms per 16-clip step for the three such layers.  Here each parity class is a dense stride-1
// FORWARD-geometry convolution over g with a 1x1 / 1x2 / 2x1 / 2x2 filter whose taps are column blocks of the same w_dgrad matrix
// (no re-packed weights: the tap-uniform K walk takes a tap -> column-block map), written with output stride 2 at offset (ph, pw).
static int conv_dgrad_s2_by_parity(const void* g, const void* w_dgrad, void* dx, const td_conv_desc* d, const td_epilogue* e, int dtype,
                                   td_stream_t stream) {
  const int Hg = d->Hs, Wg = d->Ws, Co = d->C, H = d->Ho, W = d->Wo;
  const size_t es = 2, wbytes = (size_t)d->Nc * 9 * Co * es;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      // axis taps in the sub-filter's own order r' = 0, 1 (source offset + r'): even coordinate: r = 1; odd: r' = 0 <-> r = 2, r' = 1 <-> r = 0
      const int nr = ph ? 2 : 1, ns = pw ? 2 : 1;
      const int rmap[2] = {ph ? 2 : 1, 0}, smap[2] = {pw ? 2 : 1, 0};
      td_conv_desc c = {d->N, Hg, Wg, Co, Hg, Wg, nr, ns, 1, 0, 0, d->Nc, d->ldc, 2, H, W};
      WeightSubset sub;
      memset(&sub, 0, sizeof(sub));
      sub.w_ld = 9 * Co;
      sub.ntap = nr * ns;
      for (int r = 0; r < nr; ++r)
        for (int s_ = 0; s_ < ns; ++s_) sub.wtap[r * ns + s_] = rmap[r] * 3 + smap[s_];
      const size_t ooff = ((size_t)ph * W + pw) * d->ldc * es;
      td_epilogue ec;
      memset(&ec, 0, sizeof(ec));
      if (e) ec = *e;
      if (ec.residual) ec.residual = (const char*)ec.residual + ooff;
      if (ec.mask_src) ec.mask_src = (const char*)ec.mask_src + ooff;
      const char* wp = (const char*)w_dgrad;
      if (sub.ntap == 1) {  // one tap: its column block by pointer offset (a 1x1 launch has no tap walk)
        wp += (size_t)sub.wtap[0] * Co * es;
        sub.wtap[0] = 0;
      }
      sub.w_bytes = wbytes - (size_t)(wp - (const char*)w_dgrad);
      const int rc = conv_gemm_launch(g, wp, (char*)dx + ooff, &c, &ec, dtype, stream, &sub);
      if (rc) return rc;
    }
  return TD_OK;
}

extern "C" int td_conv_gemm(const void* src, const void* wmat, void* out, const td_conv_desc* d, const td_epilogue* e,
                            int dtype, td_stream_t stream) {
  TD_REQUIRE(src && wmat && out && d, "td_conv_gemm: null pointer");
  static const int parity = [] { const char* e_ = getenv("TD_DGRAD_S2_PARITY"); return e_ ? atoi(e_) : 1; }();  // (A/B: 0 = one gather over all 9 taps)
  if (parity && dtype == TD_BF16 && d->mode == 1 && d->stride == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && !d->aniso && d->out_sp <= 1 &&
      d->Ho == 2 * d->Hs && d->Wo == 2 * d->Ws && d->C % 64 == 0 && d->ldc >= d->Nc && !(e && (e->dropout_p > 0.f || e->sigmoid))) {
    int rc = validate(d, dtype, "td_conv_gemm");
    if (rc) return rc;
    return conv_dgrad_s2_by_parity(src, wmat, out, d, e, dtype, stream);
  }
  return conv_gemm_launch(src, wmat, out, d, e, dtype, stream, nullptr);
}

static int conv_gemm_launch(const void* src, const void* wmat, void* out, const td_conv_desc* d, const td_epilogue* e, int dtype,
                            td_stream_t stream, const WeightSubset* sub) {
  TD_REQUIRE(src && wmat && out && d, "td_conv_gemm: null pointer");
  int rc = validate(d, dtype, "td_conv_gemm");
  if (rc) return rc;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = (const char*)src;
  p.w = (const char*)wmat;
  p.out = (char*)out;
  p.d = *d;
  if (p.d.out_sp < 1) p.d.out_sp = 1;
  p.M = d->N * d->Ho * d->Wo;
  p.K = d->R * d->S * d->C;
  {
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    const double sb = (double)d->N * d->Hs * d->Ws * d->C * es, wb = (double)d->Nc * p.K * es;
    TD_REQUIRE(sb < 4294967000.0 && wb < 4294967000.0, "td_conv_gemm: operand exceeds the 4 GiB buffer-descriptor range");
    p.src_bytes = (uint32_t)sb;
    p.w_bytes = (uint32_t)wb;
  }
  if (sub) {
    p.w_ld = sub->w_ld;
    p.w_bytes = (uint32_t)sub->w_bytes;
    p.use_wtap = sub->ntap > 1;
    for (int i = 0; i < 4; ++i) p.wtap[i] = sub->wtap[i];
  }
  p.alpha = 1.f;
  p.dbg = g_dbg;
  if (e) {
    p.bias = e->bias;
    p.residual = (const char*)e->residual;
    p.mask_src = (const char*)e->mask_src;
    p.relu = e->relu;
    p.sigmoid = e->sigmoid;
    if (e->alpha != 0.f) p.alpha = e->alpha;
    if (e->dropout_p > 0.f) {
      TD_REQUIRE(e->dropout_p < 1.f, "td_conv_gemm: dropout_p must be < 1");
      p.drop_thresh = (uint32_t)((double)e->dropout_p * 4294967296.0);
      if (p.drop_thresh == 0) p.drop_thresh = 1;
      p.drop_scale = 1.f / (1.f - e->dropout_p);
      p.seed = e->dropout_seed;
      p.seed_dev = e->dropout_counter;
    }
  }
  TD_REQUIRE(d->ldc >= d->Nc, "td_conv_gemm: ldc < Nc");
  hipStream_t st = (hipStream_t)stream;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  {
    // persistent weight-stationary instance (see pw_resident2_kernel): bf16 pointwise layers with short K and many rows
    static const int persist = [] { const char* e_ = getenv("TD_PW_PERSIST"); return e_ ? atoi(e_) : 1; }();
    static const int persist_dropout = [] { const char* e_ = getenv("TD_PW_PERSIST_DROPOUT"); return e_ ? atoi(e_) : 1; }();  // (A/B: 0 = dropout epilogues on the tiled kernel)
    // dense rows, or the strided 1x1 of a downsample branch in forward geometry (the kernel derives the source pixel per row)
    const bool pw0 = d->R == 1 && d->S == 1 && d->pad == 0 && p.d.out_sp == 1 && !d->aniso &&
                     ((d->stride == 1 && d->Hs == d->Ho && d->Ws == d->Wo) ||
                      (d->stride > 1 && d->mode == 0 && (d->Ho - 1) * d->stride < d->Hs && (d->Wo - 1) * d->stride < d->Ws));
    const int NTp = d->Nc / 128, Pp = 2 * n_cu / 8;
    const bool shape_ok = pw0 && !p.w_ld && dtype == TD_BF16 && p.K % 64 == 0 && p.K <= 256 && d->Nc % 128 == 0 &&
                          (NTp == 1 || NTp == 2 || NTp == 4 || NTp == 8 || NTp == 16) && Pp >= NTp && d->ldc % 8 == 0 && !p.sigmoid &&
                          p.alpha == 1.f && n_cu % 8 == 0 && (!p.drop_thresh || persist_dropout) && (double)p.M * d->ldc < 2147483000.0;
    const int MTp = cdiv(p.M, 64);
    const int groups = shape_ok ? 8 * (Pp / NTp) : 1;
    if (persist && shape_ok && MTp >= persist_min_tiles() * groups && !(persist == 2 && p.mask_src) && !(persist == 3 && d->mode != 0)) {
      const bool prof_ = prof_on();
      if (prof_) {
        prof_begin(TD_PROF_PW_RESIDENT, dtype, 2.0 * p.M * d->Nc * p.K, st, p.M, d->Nc, p.K, d->R, d->stride, d->mode);
        const double es_ = 2.0;
        double by = ((double)p.M * p.K + (double)d->Nc * p.K + (double)p.M * d->Nc) * es_;
        if (p.residual) by += (double)p.M * d->Nc * es_;
        if (p.mask_src) by += (double)p.M * d->Nc * es_;
        prof_set_bytes(by);
      }
      const int nkt = p.K / 64;
      const bool hr = p.residual != nullptr, hm = p.mask_src != nullptr;
      dim3 g2(2 * n_cu);  // two resident workgroups per CU (64 KiB of LDS each)
#define TD_PWR(NK)                                                                               \
  do {                                                                                           \
    if (hr && hm) pw_resident2_kernel<NK, true, true><<<g2, 256, 0, st>>>(p, MTp, Pp);          \
    else if (hr) pw_resident2_kernel<NK, true, false><<<g2, 256, 0, st>>>(p, MTp, Pp);          \
    else if (hm) pw_resident2_kernel<NK, false, true><<<g2, 256, 0, st>>>(p, MTp, Pp);          \
    else pw_resident2_kernel<NK, false, false><<<g2, 256, 0, st>>>(p, MTp, Pp);                 \
  } while (0)
      if (nkt == 1) TD_PWR(1);
      else if (nkt == 2) TD_PWR(2);
      else if (nkt == 3) TD_PWR(3);
      else TD_PWR(4);
#undef TD_PWR
      if (prof_) prof_end(st);
      return check_launch("td_conv_gemm(pw_resident)");
    }
  }
  // tile choice: 128x64 for <=64 output channels; 64x128 when 128x128 would leave the 256 CUs under-filled
  const bool narrow = d->Nc <= 64;
  const int tiles128 = cdiv(p.M, 128) * cdiv(d->Nc, 128);
  // 64x128 (3 workgroups / CU) also for the short-K, HBM-bound 1x1 layers: more loads in flight per CU
  const bool small_m = !narrow && (tiles128 < 512 || p.K <= 512);
  const int BMsel = small_m ? 64 : 128, BNsel = narrow ? 64 : 128;
  const int MT = cdiv(p.M, BMsel), NTl = cdiv(d->Nc, BNsel);
  dim3 grid(8 * cdiv(MT, 8) * NTl);
  const bool prof = prof_on();
  const bool pw = d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && p.d.out_sp == 1 && d->Hs == d->Ho && d->Ws == d->Wo;
  // tap-uniform addressing (see conv_gemm_kernel): spatial convs whose channel count is a multiple of the K tile
  static const int tu_on = [] { const char* e_ = getenv("TD_CONV_TAP_UNIFORM"); return e_ ? atoi(e_) : 1; }();
  const int bk = dtype == TD_BF16 ? 64 : 32;
  // (a strided 1x1 - the downsample branch of a stage's first block - is the one-tap case of the same addressing)
  const bool tu = (tu_on || p.use_wtap) && !pw && !d->aniso && (d->R * d->S > 1 || (d->stride > 1 && d->mode == 0)) && d->R * d->S <= 32 && d->C % bk == 0 &&
                  (p.d.out_sp == 1 || d->mode == 0) && (d->mode == 0 || d->stride == 1);
  // a tap -> column-block map (WeightSubset: the parity classes of a stride-2 input gradient) is honoured by the tap-uniform K walk only
  TD_REQUIRE(tu || !p.use_wtap, "td_conv_gemm: a weight-column subset needs the tap-uniform addressing (C %% %d == 0, at most 32 taps)", bk);
  {
    // 256-row tiles (conv_gemm_big8_kernel / conv_gemm_big8n_kernel): MFMA-bound bf16 layers with enough workgroups to matter
    static const int big_on = [] { const char* e_ = getenv("TD_CONV_BIG"); return e_ ? atoi(e_) : 1; }();
    static const int big_min = [] { const char* e_ = getenv("TD_CONV_BIG_MIN_WG"); return e_ ? atoi(e_) : 160; }();
    const int bnb = d->Nc % 256 == 0 ? 256 : 128;
    const int wgs = cdiv(p.M, 256) * (d->Nc / bnb);
    if (big_on && dtype == TD_BF16 && (pw || tu) && p.d.out_sp == 1 && !p.use_wtap && !p.w_ld && d->Nc % 128 == 0 && p.K % 64 == 0 && p.K >= 512 && d->ldc % 8 == 0 && !p.sigmoid &&
        !p.drop_thresh && wgs >= big_min && (double)p.M * d->ldc < 2147483000.0 &&  // (rows past M leave through an out-of-range offset: the tensor must end below it)
        p.K <= 64 * 154 && p.M < (1 << 24)) {  // (the K table's 160 entries; rows decoded with float reciprocals.  Anything larger runs on 128 x 128 tiles)
      if (prof) {
        prof_begin(tu ? TD_PROF_GEMM_256 : TD_PROF_GEMM_256_PW, dtype, 2.0 * p.M * d->Nc * p.K, st, p.M, d->Nc, p.K, d->R, d->stride, d->mode);
        double by = ((double)d->N * d->Hs * d->Ws * d->C + (double)d->Nc * p.K + (double)p.M * d->Nc) * 2.0;
        if (p.residual) by += (double)p.M * d->Nc * 2.0;
        if (p.mask_src) by += (double)p.M * d->Nc * 2.0;
        prof_set_bytes(by);
      }
      static const int tap_inner = [] { const char* e_ = getenv("TD_CONV_TAP_INNER"); return e_ ? atoi(e_) : 1; }();
      p.tap_inner = tu && tap_inner && d->R * d->S > 1;
      dim3 gb(8 * cdiv(cdiv(p.M, 256), 8) * (d->Nc / bnb));
      static const int persist8 = [] { const char* e_ = getenv("TD_CONV_BIG_PERSIST"); return e_ ? atoi(e_) : 1; }();  // (A/B: 0 = one workgroup per tile)
      if (bnb == 256) {
        const dim3 gp(persist8 && n_cu % 8 == 0 ? std::min((unsigned)n_cu, gb.x) : gb.x);
        if (tu && p.residual) conv_gemm_big8_kernel<true, true><<<gp, 512, 0, st>>>(p);
        else if (tu) conv_gemm_big8_kernel<true, false><<<gp, 512, 0, st>>>(p);
        else if (p.residual) conv_gemm_big8_kernel<false, true><<<gp, 512, 0, st>>>(p);
        else conv_gemm_big8_kernel<false, false><<<gp, 512, 0, st>>>(p);
      } else {
        if (tu) conv_gemm_big8n_kernel<true><<<gb, 512, 0, st>>>(p);
        else conv_gemm_big8n_kernel<false><<<gb, 512, 0, st>>>(p);
      }
      if (prof) prof_end(st);
      return check_launch("td_conv_gemm(256-row tiles)");
    }
  }
  if (prof) prof_begin(narrow ? TD_PROF_GEMM_128x64 : (small_m ? TD_PROF_GEMM_64x128 : TD_PROF_GEMM_128x128), dtype, 2.0 * p.M * d->Nc * p.K, st, p.M, d->Nc, p.K, d->R, d->stride, d->mode);
  if (prof) {
    // algorithmic HBM bytes: every operand / result tensor once (the gathered source counted as the tensor it is read from)
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    const double rows_out = d->out_sp > 1 ? (double)d->N * d->out_H * d->out_W : (double)p.M;
    double by = ((double)d->N * d->Hs * d->Ws * d->C + (double)d->Nc * p.K + (double)p.M * d->Nc) * es;
    if (e && e->residual) by += rows_out * d->Nc * es;
    if (e && e->mask_src) by += rows_out * d->Nc * es;
    prof_set_bytes(by);
  }
#define TD_LAUNCH(TT, BMv, BNv)                                                                   \
  do {                                                                                            \
    if (pw) conv_gemm_kernel<TT, BMv, BNv, 2, true><<<grid, 256, 0, st>>>(p);                    \
    else if (tu) conv_gemm_kernel<TT, BMv, BNv, 2, false, true><<<grid, 256, 0, st>>>(p);        \
    else conv_gemm_kernel<TT, BMv, BNv, 2, false><<<grid, 256, 0, st>>>(p);                      \
  } while (0)
  static const int deep = [] { const char* e_ = getenv("TD_CONV_DEEP"); return e_ ? atoi(e_) : 1; }();
  // under-filled launch (at most two 64x128 workgroups per CU anyway) with a long K loop: the third stage costs nothing
  // in occupancy and halves the exposed DMA round trips
  const bool deep_ok = deep && small_m && (int)grid.x <= 2 * 256 && p.K >= 1024;  // measured: +6 % at K >= 1024, neutral-to-negative at K = 256
  if (dtype == TD_BF16) {
    if (narrow) TD_LAUNCH(u16, 128, 64);
    else if (deep_ok && pw) conv_gemm_kernel<u16, 64, 128, 3, true><<<grid, 256, 0, st>>>(p);
    else if (deep_ok && tu) conv_gemm_kernel<u16, 64, 128, 3, false, true><<<grid, 256, 0, st>>>(p);
    else if (deep_ok) conv_gemm_kernel<u16, 64, 128, 3, false><<<grid, 256, 0, st>>>(p);
    else if (small_m) TD_LAUNCH(u16, 64, 128);
    else TD_LAUNCH(u16, 128, 128);
  } else {
    if (narrow) TD_LAUNCH(float, 128, 64);
    else if (small_m) TD_LAUNCH(float, 64, 128);
    else TD_LAUNCH(float, 128, 128);
  }
#undef TD_LAUNCH
  if (prof) prof_end(st);
  return check_launch("td_conv_gemm");
}

// out[out_map[m]][n] = epilogue( [A1[a1_map[m]] | A2[a2_map[m]]] @ wmat[n][K1 + K2]^T (+ residual[res_map[m]][n]) )
extern "C" int td_linear_ex(const void* a1, const void* a2, const void* wmat, void* out, const td_linear_ex_desc* x, const td_epilogue* e,
                            int dtype, td_stream_t stream) {
  TD_REQUIRE(a1 && wmat && out && x, "td_linear_ex: null pointer");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_linear_ex: bad dtype %d", dtype);
  const int vec = dtype == TD_BF16 ? 8 : 4, bk = dtype == TD_BF16 ? 64 : 32;
  const int es = dtype == TD_BF16 ? 2 : 4;
  const int K2 = a2 ? x->K2 : 0;
  TD_REQUIRE(x->M >= 1 && x->N >= 1 && x->K1 >= vec && K2 >= 0, "td_linear_ex: bad sizes");
  TD_REQUIRE(x->K1 % vec == 0 && K2 % vec == 0 && x->lda1 % vec == 0 && (!a2 || x->lda2 % vec == 0), "td_linear_ex: K1 / K2 / lda must be multiples of %d", vec);
  TD_REQUIRE(!a2 || x->K1 % bk == 0, "td_linear_ex: with a second source K1=%d must be a multiple of the K tile (%d)", x->K1, bk);
  TD_REQUIRE(x->lda1 >= x->K1 && (!a2 || x->lda2 >= K2) && x->ldc >= x->N, "td_linear_ex: row stride smaller than the row");
  TD_REQUIRE(x->rows1 >= 1 && (!a2 || x->rows2 >= 1) && (x->a1_map || x->rows1 >= x->M) && (!a2 || x->a2_map || x->rows2 >= x->M),
             "td_linear_ex: source has fewer rows than M and no row map");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = (const char*)a1;
  p.src2 = (const char*)a2;
  p.w = (const char*)wmat;
  p.out = (char*)out;
  p.M = x->M;
  p.K = x->K1 + K2;
  p.K1 = x->K1;
  p.lda1 = x->lda1;
  p.lda2 = a2 ? x->lda2 : 0;
  p.ldr = x->ldr > 0 ? x->ldr : x->ldc;
  p.w_shared = (a2 && x->w_shared) ? 1 : 0;
  TD_REQUIRE(!p.w_shared || K2 == x->K1, "td_linear_ex: a shared weight needs K2 == K1");
  p.a1_map = x->a1_map;
  p.a2_map = a2 ? x->a2_map : nullptr;
  p.out_map = x->out_map;
  p.res_map = x->res_map;
  td_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.N = 1; d.Hs = x->M; d.Ws = 1; d.C = p.K; d.Ho = x->M; d.Wo = 1; d.R = d.S = 1; d.stride = 1; d.Nc = x->N; d.ldc = x->ldc; d.out_sp = 1;
  p.d = d;
  {
    const double b1 = (double)x->rows1 * x->lda1 * es, b2 = a2 ? (double)x->rows2 * x->lda2 * es : 16.0, wb = (double)x->N * (p.w_shared ? x->K1 : p.K) * es;
    TD_REQUIRE(b1 < 4294967000.0 && b2 < 4294967000.0 && wb < 4294967000.0, "td_linear_ex: operand exceeds the 4 GiB buffer-descriptor range");
    p.src_bytes = (uint32_t)b1;
    p.src2_bytes = (uint32_t)b2;
    p.w_bytes = (uint32_t)wb;
  }
  p.alpha = 1.f;
  if (e) {
    p.bias = e->bias;
    p.residual = (const char*)e->residual;
    p.mask_src = (const char*)e->mask_src;
    p.relu = e->relu;
    p.sigmoid = e->sigmoid;
    if (e->alpha != 0.f) p.alpha = e->alpha;
    if (e->dropout_p > 0.f) {
      TD_REQUIRE(e->dropout_p < 1.f, "td_linear_ex: dropout_p must be < 1");
      p.drop_thresh = (uint32_t)((double)e->dropout_p * 4294967296.0);
      if (p.drop_thresh == 0) p.drop_thresh = 1;
      p.drop_scale = 1.f / (1.f - e->dropout_p);
      p.seed = e->dropout_seed;
      p.seed_dev = e->dropout_counter;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const bool narrow = x->N <= 64;
  const int tiles128 = cdiv(p.M, 128) * cdiv(x->N, 128);
  const bool small_m = !narrow && (tiles128 < 512 || p.K <= 512);
  const int BMsel = small_m ? 64 : 128, BNsel = narrow ? 64 : 128;
  dim3 grid(8 * cdiv(cdiv(p.M, BMsel), 8) * cdiv(x->N, BNsel));
  const bool prof = prof_on();
  if (prof) {
    prof_begin(narrow ? TD_PROF_GEMM_128x64 : (small_m ? TD_PROF_GEMM_64x128 : TD_PROF_GEMM_128x128), dtype, 2.0 * p.M * x->N * p.K, st, p.M, x->N, p.K, 1, 1, 2);
    double by = ((double)p.M * p.K + (double)x->N * p.K + (double)p.M * x->N) * es;  // every gathered row counted once per use
    if (p.residual) by += (double)p.M * x->N * es;
    if (p.mask_src) by += (double)p.M * x->N * es;
    prof_set_bytes(by);
  }
#define TD_LX(TT)                                                                                     \
  do {                                                                                                \
    if (narrow) conv_gemm_kernel<TT, 128, 64, 2, true, false, true><<<grid, 256, 0, st>>>(p);         \
    else if (small_m) conv_gemm_kernel<TT, 64, 128, 2, true, false, true><<<grid, 256, 0, st>>>(p);   \
    else conv_gemm_kernel<TT, 128, 128, 2, true, false, true><<<grid, 256, 0, st>>>(p);               \
  } while (0)
  if (dtype == TD_BF16) TD_LX(u16);
  else TD_LX(float);
#undef TD_LX
  if (prof) prof_end(st);
  return check_launch("td_linear_ex");
}

static int wgrad_stages() {
  static const int nstg = [] { const char* e = getenv("TD_WGRAD_STAGES"); return (e && atoi(e) == 2) ? 2 : 4; }();
  return nstg;
}

// fills everything of p except the output fields; returns the split count through *splits_io.
// auto_mode: 0 = single launch (about one workgroup per CU), > 0 = batched launch with work items of that many stages
static int wgrad_fill(WgradParams& p, const void* g, const void* src, const td_conv_desc* d, int ldg, int dtype, int* splits_io,
                      int auto_mode, const char* who) {
  int rc = validate(d, dtype, who);
  if (rc) return rc;
  const int vec = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(d->mode == 0 && !d->aniso, "%s: isotropic forward geometry expected", who);
  TD_REQUIRE(d->Nc % vec == 0 && ldg % vec == 0, "%s: Nc=%d / ldg=%d must be multiples of %d", who, d->Nc, ldg, vec);
  memset(&p, 0, sizeof(p));
  p.g = (const char*)g;
  p.src = (const char*)src;
  p.d = *d;
  p.M = d->N * d->Ho * d->Wo;
  p.K = d->R * d->S * d->C;
  p.ldg = ldg;
  p.ci_real = d->C;
  {
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    const double gb = (double)p.M * ldg * es, sb = (double)d->N * d->Hs * d->Ws * d->C * es;
    TD_REQUIRE(gb < 4294967000.0 && sb < 4294967000.0, "%s: operand exceeds the 4 GiB buffer-descriptor range", who);
    p.g_bytes = (uint32_t)gb;
    p.src_bytes = (uint32_t)sb;
  }
  const int mk = dtype == TD_BF16 ? 64 : 32;
  p.tn = cdiv(d->Nc, 128);
  p.tk = cdiv(p.K, 128);
  int splits = *splits_io;
  if (splits < 1) {
    if (auto_mode == 0) {
      // single launch: about one workgroup per CU (TD_WGRAD_BLOCKS, default 256), at least 8 reduction stages per split.
      // measured: 192-256 workgroups beat 512-1536 (fewer fp32 atomics per output tile)
      static const int target_blocks = [] { const char* e = getenv("TD_WGRAD_BLOCKS"); return e ? atoi(e) : 256; }();
      const int tiles = p.tn * p.tk;
      splits = wgrad_stages() == 4 ? target_blocks / tiles : cdiv(target_blocks, tiles);  // 4 stages: one workgroup per CU, never a second round
      const int maxs = cdiv(p.M, 8 * mk);
      if (splits > maxs) splits = maxs;
    } else {
      // batched launch: the tiles of all jobs fill the chip; split only to bound the longest work item.  Every split pays an
      // fp32-atomic epilogue (65 536 atomics per 256 x 256 tile, 36-byte strided for the 3x3 layers: ~30 % of a 192-stage
      // item), an unsplit trunk leaves 3 000-stage items next to 190-stage ones: measured over the trunk's 93 jobs at 8 clips
      // 8.5 ms at 192 stages per item, 7.1 ms at 512 .. 1024, 10.4 ms unsplit.
      //   The length is chosen per launch (td_conv_wgrad_batch): total work / (3.4 items per CU), within 192 .. 2048 stages -
      //   512 at 4 clips, ~1000 at 8 (in the network: 80.5 clips/s at 512, 81.1 at 1024); auto_mode carries it.
      splits = cdiv(p.M, std::max(1, auto_mode) * mk);
    }
    if (splits < 1) splits = 1;
  }
  if (deterministic()) splits = 1;  // one workgroup per output tile walks the whole reduction: no atomics between splits, a fixed summation order
  p.mper = cdiv(cdiv(p.M, splits), mk) * mk;
  *splits_io = cdiv(p.M, p.mper);
  return TD_OK;
}

extern "C" int td_conv_wgrad(const void* g, const void* src, float* dw, const td_conv_desc* d, int ldg, int dtype,
                             int splits, td_stream_t stream) {
  return td_conv_wgrad_bias(g, src, dw, nullptr, d, ldg, dtype, splits, stream);
}

extern "C" int td_conv_wgrad_bias(const void* g, const void* src, float* dw, float* dbias, const td_conv_desc* d, int ldg,
                                  int dtype, int splits, td_stream_t stream) {
  TD_REQUIRE(g && src && dw && d, "td_conv_wgrad: null pointer");
  WgradParams p;
  int rc = wgrad_fill(p, g, src, d, ldg, dtype, &splits, 0, "td_conv_wgrad");
  if (rc) return rc;
  p.dw = dw;
  p.dbias = dbias;
  p.out_mode = 0;
  p.stamps = g_dbg;
  dim3 grid(p.tn, p.tk, splits);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) prof_begin(TD_PROF_WGRAD_SINGLE, dtype, 2.0 * p.M * d->Nc * p.K, st, p.M, d->Nc, p.K, d->R, d->stride, splits);
  if (prof) prof_set_bytes(((double)p.M * ldg + (double)d->N * d->Hs * d->Ws * d->C) * (dtype == TD_BF16 ? 2.0 : 4.0) + (double)d->Nc * p.K * 4.0);
  const bool pw = (d->R * d->S == 1) && d->stride == 1 && d->pad == 0;
  const int nstg = wgrad_stages();
#define TD_WG_LAUNCH(TT, NS)                                                    \
  do {                                                                          \
    if (pw) conv_wgrad_kernel<TT, NS, true><<<grid, 256, 0, st>>>(p);           \
    else conv_wgrad_kernel<TT, NS, false><<<grid, 256, 0, st>>>(p);             \
  } while (0)
  if (nstg == 4) {
    if (dtype == TD_BF16) TD_WG_LAUNCH(u16, 4);
    else TD_WG_LAUNCH(float, 4);
  } else {
    if (dtype == TD_BF16) TD_WG_LAUNCH(u16, 2);
    else TD_WG_LAUNCH(float, 2);
  }
#undef TD_WG_LAUNCH
  if (prof) prof_end(st);
  return check_launch("td_conv_wgrad");
}

// ---- batched weight gradients ----
// The job table of a launch lives in caller-provided memory: the library writes it into `table_host` (page-locked),
// enqueues ONE hipMemcpyAsync into `table_dev` on the caller's stream and launches; no allocation, no synchronisation.
static size_t wg_table_half(int n_jobs) { return (((size_t)n_jobs * sizeof(WgradParams)) + 255) & ~(size_t)255; }
extern "C" size_t td_conv_wgrad_batch_table_bytes(int n_jobs) { return n_jobs > 0 ? 8 * wg_table_half(n_jobs) : 0; }  // (general / pointwise / wide-tile / four-wavefront wide-tile tables) x (overwriting / accumulating jobs)

static int wgrad_batch_phase(const td_wgrad_job* jobs, int n_jobs, int dtype, void* table_host, void* table_dev, size_t half, bool accumulate, td_stream_t stream);

extern "C" int td_conv_wgrad_batch(const td_wgrad_job* jobs, int n_jobs, int dtype, void* table_host, void* table_dev,
                                   size_t table_bytes, td_stream_t stream) {
  TD_REQUIRE(jobs && n_jobs >= 1, "td_conv_wgrad_batch: no jobs");
  TD_REQUIRE(table_host && table_dev && table_bytes >= td_conv_wgrad_batch_table_bytes(n_jobs),
             "td_conv_wgrad_batch: job-table workspace missing or smaller than td_conv_wgrad_batch_table_bytes(%d)", n_jobs);
  const size_t half = wg_table_half(n_jobs);
  // jobs flagged `accumulate` add to what the overwriting jobs of the same call wrote: they run as a second launch set behind them
  std::vector<td_wgrad_job> first, second;
  for (int i = 0; i < n_jobs; ++i) (jobs[i].accumulate ? second : first).push_back(jobs[i]);
  int rc = TD_OK;
  if (!first.empty()) rc = wgrad_batch_phase(first.data(), (int)first.size(), dtype, table_host, table_dev, half, false, stream);
  if (rc == TD_OK && !second.empty())
    rc = wgrad_batch_phase(second.data(), (int)second.size(), dtype, (char*)table_host + 4 * half, (char*)table_dev + 4 * half, half, true, stream);
  return rc;
}

static int wgrad_batch_phase(const td_wgrad_job* jobs, int n_jobs, int dtype, void* table_host, void* table_dev, size_t half, bool accumulate,
                             td_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  std::vector<WgradParams> tab[4];  // [0] general geometry, [1] pointwise, [2] wide tiles (conv_wgrad_wide_batch_kernel), [3] 256 x 256 tiles on four wavefronts (conv_wgrad_wide4_batch_kernel)
  static const int wide8_on = [] { const char* e = getenv("TD_WGRAD_WIDE8"); return e ? atoi(e) : 0; }();  // the wide launch on eight wavefronts per workgroup (128 x 64 wave tiles)
  static const int wide4_on = [] { const char* e = getenv("TD_WGRAD_WIDE4"); return e ? atoi(e) : 0; }();  // (measured 11 % SLOWER than the sixteen-wavefront tiles on the trunk's table, profiles/r06_wgrad_wide4.log: off by default, kept for the A/B and its bit-identity test)
  double flops = 0, abytes = 0;
  static const int wide_on = [] { const char* e = getenv("TD_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
  static const int wide_min_m = [] { const char* e = getenv("TD_WGRAD_WIDE_MIN_M"); return e ? atoi(e) : 4096; }();
  // stages (64 / 32 reduction rows) per work item: total work of the launch in 256 x 256 tile-stages / (3.4 items per CU)
  int item_stages;
  {
    static const int fixed = [] { const char* e = getenv("TD_WGRAD_SPLIT_STAGES"); return e ? atoi(e) : 0; }();
    static const int n_cu = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return cus;
    }();
    double units = 0;
    const int mk = dtype == TD_BF16 ? 64 : 32;
    for (int i = 0; i < n_jobs; ++i) {
      const td_conv_desc& d = jobs[i].d;
      units += (double)cdiv(d.Nc, 128) * cdiv(d.R * d.S * d.C, 128) / 4.0 * cdiv(d.N * d.Ho * d.Wo, mk);
    }
    item_stages = fixed > 0 ? fixed : (int)std::min(2048.0, std::max(192.0, units / (3.4 * n_cu)));
  }
  for (int i = 0; i < n_jobs; ++i) {
    const td_wgrad_job& j = jobs[i];
    TD_REQUIRE(j.g && j.src && j.dW, "td_conv_wgrad_batch: job %d has a null pointer", i);
    WgradParams p;
    int splits = 0;
    int rc = wgrad_fill(p, j.g, j.src, &j.d, j.ldg, dtype, &splits, item_stages, "td_conv_wgrad_batch");
    if (rc) return rc;
    TD_REQUIRE(j.ci_real >= 1 && j.ci_real <= j.d.C, "td_conv_wgrad_batch: job %d: ci_real=%d out of range", i, j.ci_real);
    p.dw = j.dW;
    p.dbias = j.dbias;
    p.scale = j.scale;
    p.ci_real = j.ci_real;
    p.out_mode = (splits == 1 && !accumulate) ? 2 : 1;
    TD_REQUIRE(!accumulate || !j.dbias, "td_conv_wgrad_batch: job %d: an accumulating job cannot carry a bias gradient", i);
    p.first = splits;  // (temporarily: the split count, replaced by the first workgroup index below)
    if (splits > 1 && !accumulate && !j.prezeroed &&
        (hipMemsetAsync(j.dW, 0, (size_t)j.d.Nc * j.ci_real * j.d.R * j.d.S * sizeof(float), st) != hipSuccess ||
         (j.dbias && hipMemsetAsync(j.dbias, 0, (size_t)j.d.Nc * sizeof(float), st) != hipSuccess))) {
      set_error("td_conv_wgrad_batch: memset failed");
      return TD_ERR_LAUNCH;
    }
    const bool pw = (j.d.R * j.d.S == 1) && j.d.stride == 1 && j.d.pad == 0;
    // wide tiles: bf16, many reduction rows, whole 128-column sub-tiles (see wgrad_wide_body), no fused bias gradient
    const bool wide = wide_on && dtype == TD_BF16 && p.M >= wide_min_m && j.d.Nc % 128 == 0 && j.d.C % 128 == 0 && !j.dbias &&
                      j.d.R * j.d.S <= 255 && (j.d.Nc % 256 == 0 || p.K >= 256);
    if (wide) {
      const int gs = j.d.Nc % 256 == 0 ? 2 : 1, xs = p.K >= 256 ? 2 : 1;
      p.cls = (gs - 1) | ((xs - 1) << 1) | (pw ? 4 : 0);
      p.tn = j.d.Nc / (gs * 128);
      p.tk = cdiv(p.K, xs * 128);
    }
    tab[wide ? ((wide4_on && (p.cls & 3) == 3) ? 3 : 2) : (pw ? 1 : 0)].push_back(p);
    flops += 2.0 * p.M * j.d.Nc * p.K;
    abytes += ((double)p.M * j.ldg + (double)j.d.N * j.d.Hs * j.d.Ws * j.d.C) * (dtype == TD_BF16 ? 2.0 : 4.0) + (double)j.d.Nc * j.ci_real * j.d.R * j.d.S * 4.0;
  }
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_WGRAD, dtype, flops, st, 0, 0, 0, 0, 0, n_jobs);
    prof_set_bytes(abytes);
  }
  const int nstg = wgrad_stages();
  for (int pw = 0; pw < 4; ++pw) {
    std::vector<WgradParams>& t = tab[pw];
    if (t.empty()) continue;
    // one XCD per job, longest job first onto the least loaded XCD; inside an XCD the long work items come first so
    // that the tail of the launch is made of short ones
    const int nj = (int)t.size();
    std::vector<int> order(nj), owner(nj);
    std::vector<double> cost(nj);
    for (int i = 0; i < nj; ++i) {
      order[i] = i;
      cost[i] = (double)t[i].tn * t[i].tk * t[i].first /*splits*/ * (double)t[i].mper * ((pw == 1 || (t[i].cls & 4)) ? 1.0 : 1.5) *
                (pw >= 2 && (t[i].cls & 3) != 3 ? 0.6 : 1.0);  // half-size wide tiles: half the MFMAs, same DMA count
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : order) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      owner[i] = best;
      load[best] += cost[i];
    }
    std::vector<WgradParams> sorted;
    sorted.reserve(nj);
    WgradXcdIndex xi;
    long long max_slots = 0;
    for (int x = 0; x < 8; ++x) {
      xi.start[x] = (int)sorted.size();
      std::vector<int> mine;
      for (int i = 0; i < nj; ++i)
        if (owner[i] == x) mine.push_back(i);
      std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) {
        return (double)t[a].mper * t[a].d.R * t[a].d.S > (double)t[b].mper * t[b].d.R * t[b].d.S;
      });
      long long slots = 0;
      for (int i : mine) {
        WgradParams p = t[i];
        const int splits = p.first;
        p.first = (int)slots;
        p.stamps = g_dbg;
        slots += (long long)p.tn * p.tk * splits;
        sorted.push_back(p);
      }
      TD_REQUIRE(slots < 200000000LL, "td_conv_wgrad_batch: too many work items");
      xi.slots[x] = (int)slots;
      max_slots = std::max(max_slots, slots);
    }
    xi.start[8] = (int)sorted.size();
    // instance `pw` owns half `pw` of the caller's table; the async copy reads table_host when the stream gets there
    // (or, inside a captured graph, at every replay): the caller keeps both buffers untouched until then
    WgradParams* host = (WgradParams*)((char*)table_host + pw * half);
    const WgradParams* dev = (const WgradParams*)((char*)table_dev + pw * half);
    memcpy(host, sorted.data(), sorted.size() * sizeof(WgradParams));
    if (hipMemcpyAsync((void*)dev, host, sorted.size() * sizeof(WgradParams), hipMemcpyHostToDevice, st) != hipSuccess) {
      set_error("td_conv_wgrad_batch: job table upload failed");
      return TD_ERR_LAUNCH;
    }
    int rc;
    const unsigned grid = (unsigned)(8 * max_slots);
#define TD_WGB_LAUNCH(TT, NS)                                                                \
  do {                                                                                       \
    if (pw) conv_wgrad_batch_kernel<TT, NS, true><<<grid, 256, 0, st>>>(dev, xi);            \
    else conv_wgrad_batch_kernel<TT, NS, false><<<grid, 256, 0, st>>>(dev, xi);              \
  } while (0)
    if (pw == 3) {
      conv_wgrad_wide4_batch_kernel<<<grid, 256, 0, st>>>(dev, xi);
    } else if (pw == 2) {
      if (wide8_on) conv_wgrad_wide8_batch_kernel<<<grid, 512, 0, st>>>(dev, xi);
      else conv_wgrad_wide_batch_kernel<<<grid, 1024, 0, st>>>(dev, xi);
    } else if (nstg == 4) {
      if (dtype == TD_BF16) TD_WGB_LAUNCH(u16, 4);
      else TD_WGB_LAUNCH(float, 4);
    } else {
      if (dtype == TD_BF16) TD_WGB_LAUNCH(u16, 2);
      else TD_WGB_LAUNCH(float, 2);
    }
#undef TD_WGB_LAUNCH
    rc = check_launch("td_conv_wgrad_batch");
    if (rc) return rc;
  }
  if (prof) prof_end(st);
  return TD_OK;
}
