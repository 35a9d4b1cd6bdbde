// Time-aligned cross-attention of the space-time decoder (models/transformer.py:725-745) for its actual shape: ONE query
// per frame against the S = hw + L memory rows of that frame, 8 heads of 32 channels.
//
// The reference projects every memory row to a key and a value in every one of the six layers
// (nn.MultiheadAttention's in_proj, 2 x 2 * (b*t*S) * 256 * 256 FLOP per layer: 93 % of the decoder's arithmetic, SURVEY.md
// section 8a row D3) and then uses each projected row for exactly one dot product and one weighted sum.  With a single query
// per frame both projections commute to the query side:
//
//   score[h][s] = q_h . (W_k,h (x_s + pos_s) + b_k,h) = (W_k,h^T q_h) . (x_s + pos_s) + q_h . b_k,h
//                                                        `--- u_h ---'                  `- constant over s: cancels in the softmax
//   out_h       = sum_s pd[h][s] (W_v,h x_s + b_v,h)   = W_v,h (sum_s pd[h][s] x_s) + b_v,h sum_s pd[h][s]
//                                                              `------ z_h ------'
//
// (pd = the probabilities after dropout).  u (one 256-vector per frame and head) and the product with W_v,h are plain GEMMs
// over the b*t query rows (block-structured weights, td_head_blocks_expand); what is left per frame is what this file does:
// S x 8 dot products of length 256 against the memory rows as they lie in HBM, a softmax, and 8 weighted row sums - the
// memory is read once per layer (twice: the second pass hits L2), nothing of size rows x 256 is written, and the key / value
// projections, their 2 x (b*t*S) x 1536 activations, their input-gradient GEMMs and their weight-gradient jobs do not exist.
// The kernels are HBM-bound by the memory rows.  Two families: the templates right below do the arithmetic in fp32 on the VALU in
// both dtypes (24 FLOP per loaded byte; the exact-fp32 parity mode, S > 256); in bf16 both products of a frame - and the gradient of
// the shared memory over all layers - are v_mfma_f32_16x16x32_bf16 tiles (cross_q1_fwd_mfma_kernel, cross_q1_bwd_mfma_kernel,
// cross_q1_dmem_kernel further down): the rows come from HBM once per layer and direction.
//
// Layout of a VALU workgroup (256 threads, one frame): lane l of every wavefront owns channels 4l .. 4l+3; wavefront w walks the
// blocks of 8 (4) consecutive rows - their loads are issued back to back -; the 8 per-head partial dot products of a row are
// reduced across the 64 lanes by a butterfly that halves the number of live values per step (ten DPP / permlane-swap exchanges
// instead of 48 shuffles through the LDS crossbar).
#include "td_common.h"
#include <math.h>
#include <stdlib.h>

namespace td {

constexpr int QE = 256, QH = 8;  // model width, heads

struct CrossQ1Params {
  const void *u, *mem, *pos, *dz;
  const uint8_t* kpm;
  float* probs;
  float* wavg;
  const float* dwavg;
  void *zext, *du;
  float* dmem;
  u16* coef;  // deferred d(memory) (bf16 mode): row f*S + s receives [ds[0..8) | pd[0..8)] as bf16 at column coef_col (row stride coef_ld)
  int coef_ld, coef_col;
  int F, S, ldz, accumulate;
  uint32_t drop_thresh;
  float drop_scale;
  uint32_t seed;
  const uint32_t* seed_dev;
};

template <typename T>
__device__ __forceinline__ float4 load4(const void* p, size_t i);
template <>
__device__ __forceinline__ float4 load4<float>(const void* p, size_t i) { return *(const float4*)((const float*)p + i); }
template <>
__device__ __forceinline__ float4 load4<u16>(const void* p, size_t i) {
  const uint2 r = *(const uint2*)((const u16*)p + i);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
}
template <typename T>
__device__ __forceinline__ void store8(void* p, size_t i, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(void* p, size_t i, const float (&v)[8]) {
  float* d = (float*)p + i;
  *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<u16>(void* p, size_t i, const float (&v)[8]) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  r.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  r.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  r.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *(uint4*)((u16*)p + i) = r;
}

// 8 per-lane partial sums -> the total over the 64 lanes of value head_of_lane(lane), valid in every lane.  Butterfly that halves
// the live values per step inside each group of 8 lanes (row_half_mirror: partner 7 - i; quad_perm: partners i ^ 1, i ^ 2), then
// sums the eight groups (row_ror:8, v_permlane16_swap, v_permlane32_swap): ten VALU-rate lane exchanges, no LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ int head_of_lane(int lane) { return ((lane & 4) ? 4 : 0) + ((lane & 1) ? 2 : 0) + ((lane & 2) ? 1 : 0); }
__device__ __forceinline__ float head_reduce8(const float (&v)[QH], int lane) {
  const bool b2 = lane & 4, b0 = lane & 1, b1 = lane & 2;
  float w[4], x[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (b2 ? v[i + 4] : v[i]) + dpp_f<0x141>(b2 ? v[i] : v[i + 4]);  // row_half_mirror
#pragma unroll
  for (int i = 0; i < 2; ++i) x[i] = (b0 ? w[i + 2] : w[i]) + dpp_f<0xB1>(b0 ? w[i] : w[i + 2]);   // quad_perm [1,0,3,2]
  float y = (b1 ? x[1] : x[0]) + dpp_f<0x4E>(b1 ? x[0] : x[1]);                                      // quad_perm [2,3,0,1]
  y += dpp_f<0x128>(y);                                                                              // row_ror:8
  {
    const int yi = __float_as_int(y);
    const auto r = __builtin_amdgcn_permlane16_swap(yi, yi, false, false);
    y = __int_as_float(r[0]) + __int_as_float(r[1]);
  }
  {
    const int yi = __float_as_int(y);
    const auto r = __builtin_amdgcn_permlane32_swap(yi, yi, false, false);
    y = __int_as_float(r[0]) + __int_as_float(r[1]);
  }
  return y;
}

__device__ __forceinline__ float dot4(const float4& a, const float (&b)[4]) { return a.x * b[0] + a.y * b[1] + a.z * b[2] + a.w * b[3]; }

// Sum the four wavefronts' [QH][QE] partial accumulators (LDS) and store row f of `dst` (row stride ld, element type T)
template <typename T>
__device__ __forceinline__ void reduce_store_rows(const float* part, void* dst, size_t row_off, int t) {
  const int e0 = t * 8;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float4 a = *(const float4*)(part + w * QH * QE + e0), b = *(const float4*)(part + w * QH * QE + e0 + 4);
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
  }
  store8<T>(dst, row_off + e0, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void cross_q1_fwd_kernel(CrossQ1Params p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = p.S, SP = (S + 3) & ~3;
  float* sS = sm;                 // [QH][SP]  scores, then exp(score - max)
  float* sP = sS + QH * SP;       // [SP][QH]  probabilities after dropout
  float* sZ = sP + SP * QH;       // [4][QH * QE]  per-wavefront partial weighted sums
  float* sSp = sZ + 4 * QH * QE;  // [QH]  sum_s pd
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int f = blockIdx.x;
  const size_t row0 = (size_t)f * S;
  float ur[QH][4];
#pragma unroll
  for (int h = 0; h < QH; ++h) {
    const float4 v = load4<T>(p.u, (size_t)f * QH * QE + h * QE + 4 * lane);
    ur[h][0] = v.x; ur[h][1] = v.y; ur[h][2] = v.z; ur[h][3] = v.w;
  }
  // KB rows per wavefront and iteration: their loads are issued back to back, the KB reduction chains interleave
  constexpr int KB = 8;
  const int hl = head_of_lane(lane);
  for (int s0 = wave * KB; s0 < S; s0 += 4 * KB) {
    float4 m[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) m[j] = load4<T>(p.mem, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
    if (p.pos) {
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const float4 q = load4<T>(p.pos, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
        m[j].x += q.x; m[j].y += q.y; m[j].z += q.z; m[j].w += q.w;
      }
    }
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      float part[QH];
#pragma unroll
      for (int h = 0; h < QH; ++h) part[h] = dot4(m[j], ur[h]);
      const float tot = head_reduce8(part, lane);
      if (lane < QH && s0 + j < S) sS[hl * SP + s0 + j] = tot;
    }
  }
  __syncthreads();
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave + 4 * hh;
    float* row = sS + h * SP;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) {
      float v = row[s];
      if (p.kpm && p.kpm[row0 + s]) v = -INFINITY;
      row[s] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float e = __expf(row[s] - mx);
      row[s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float spd = 0.f;
    const size_t prow = ((size_t)f * QH + h) * S;
    for (int s = lane; s < S; s += 64) {
      float pr = row[s] * inv;
      p.probs[prow + s] = pr;
      if (p.drop_thresh) pr = dropout_keep(seed, (uint32_t)(prow + s), p.drop_thresh) ? pr * p.drop_scale : 0.f;
      sP[s * QH + h] = pr;
      spd += pr;
    }
    spd = wave_sum(spd);
    if (lane == 0) sSp[h] = spd;
  }
  __syncthreads();
  if (p.wavg)
    for (int s = t; s < S; s += 256) {
      const float4 a = *(const float4*)(sP + s * QH), b = *(const float4*)(sP + s * QH + 4);
      p.wavg[row0 + s] = (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * (1.f / QH);
    }
  float z[QH][4];
#pragma unroll
  for (int h = 0; h < QH; ++h) z[h][0] = z[h][1] = z[h][2] = z[h][3] = 0.f;
  for (int s0 = wave * KB; s0 < S; s0 += 4 * KB) {
    float4 m[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) m[j] = load4<T>(p.mem, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      if (s0 + j >= S) break;  // wave-uniform
      const float4 a = *(const float4*)(sP + (s0 + j) * QH), b = *(const float4*)(sP + (s0 + j) * QH + 4);
      const float pd[QH] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int h = 0; h < QH; ++h) {
        z[h][0] += pd[h] * m[j].x; z[h][1] += pd[h] * m[j].y; z[h][2] += pd[h] * m[j].z; z[h][3] += pd[h] * m[j].w;
      }
    }
  }
#pragma unroll
  for (int h = 0; h < QH; ++h) *(float4*)(sZ + wave * QH * QE + h * QE + 4 * lane) = make_float4(z[h][0], z[h][1], z[h][2], z[h][3]);
  __syncthreads();
  reduce_store_rows<T>(sZ, p.zext, (size_t)f * p.ldz, t);
  if (t < QH) Elem<T>::store(p.zext, (size_t)f * p.ldz + QH * QE + t, sSp[t]);
}

// The frame core on the matrix pipe (bf16 mode, S <= SMAX): both products of a frame are tiles of v_mfma_f32_16x16x32_bf16.
//   scores  D[head][row]   = U [16 x 256] . X^T   (A = u, heads 8..15 zero; B = 16 memory rows, 16 contiguous bytes of a row per lane:
//           the natural operand, straight from HBM; the positional rows are a second product into the same accumulator - never added)
//   sums    D[head][chan]  = P [16 x S] . X       (A = the probabilities after dropout in bf16; B = the SAME rows, now needed with the
//           row index along the lane's eight values: the wavefront that fetched a row tile also left it in an LDS image, read back
//           with the transposing ds_read_b64_tr_b16 - the memory rows come from HBM once, the VALU kernel's second pass re-read them)
// Eight wavefronts per frame: wavefront w fetches row tiles w, w + 8 (16 rows x 512 bytes of memory + as much of pos in flight per
// tile), normalises head w and owns channels 32 w .. 32 w + 31 of the sums.  Softmax, dropout mask, probs / wavg outputs: exactly the VALU kernel's code on fp32 scores.
typedef __bf16 cq_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 cq_bf4 __attribute__((ext_vector_type(4)));
typedef float cq_f4 __attribute__((ext_vector_type(4)));

template <int SMAX>
__global__ __launch_bounds__(512) void cross_q1_fwd_mfma_kernel(CrossQ1Params p) {
  constexpr int HALF = SMAX * 256;  // one 128-channel half of the row image: SMAX rows of 256 bytes
  __shared__ __attribute__((aligned(16))) char sX[2 * HALF];
  __shared__ __attribute__((aligned(16))) float sS[QH * SMAX];   // scores, then exp(score - max)
  __shared__ __attribute__((aligned(16))) float sP[SMAX * QH];   // probabilities after dropout, [s][h] (wavg)
  __shared__ __attribute__((aligned(16))) u16 sPb[QH * SMAX];    // the same in bf16, [h][s]: A operand of the sums
  __shared__ float sSp[QH];
  const int S = p.S, SP32 = (S + 31) & ~31;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, lg = lane >> 4, jrow = lr >> 2, q = lr & 3;
  const int f = blockIdx.x;
  const size_t row0 = (size_t)f * S;
  const u16* mem = (const u16*)p.mem;
  const u16* pos = (const u16*)p.pos;
  uint4 ua[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    ua[ks] = make_uint4(0u, 0u, 0u, 0u);
    if (lr < QH) ua[ks] = *(const uint4*)((const u16*)p.u + ((size_t)f * QH + lr) * QE + ks * 32 + lg * 8);
  }
  for (int m0 = wave * 16; m0 < SP32; m0 += 128) {
    const int row = m0 + lr;
    const bool valid = row < S;
    uint4 xm[8], xp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      xm[ks] = make_uint4(0u, 0u, 0u, 0u);
      if (valid) xm[ks] = *(const uint4*)(mem + (row0 + row) * QE + ks * 32 + lg * 8);
    }
    if (pos) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        xp[ks] = make_uint4(0u, 0u, 0u, 0u);
        if (valid) xp[ks] = *(const uint4*)(pos + (row0 + row) * QE + ks * 32 + lg * 8);
      }
    }
    cq_f4 acc = cq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cq_bf8*)&ua[ks], *(const cq_bf8*)&xm[ks], acc, 0, 0, 0);
    if (pos) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cq_bf8*)&ua[ks], *(const cq_bf8*)&xp[ks], acc, 0, 0, 0);
    }
    // D[i = head][j = row]: lane holds heads 4 lg + rr of row m0 + lr
    if (lg < 2 && valid) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) sS[(4 * lg + rr) * SMAX + row] = acc[rr];
    }
    // the rows themselves into the image (rows >= S: zeros): chunk c = 4 ks + lg of half c >> 4 at slot (((c16 >> 1) ^ swz(row)) << 1) | (c16 & 1)
    const int swz = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int c = 4 * ks + lg, hf = c >> 4, c16 = c & 15;
      const int slot = (((c16 >> 1) ^ swz) << 1) | (c16 & 1);
      *(uint4*)(sX + hf * HALF + row * 256 + slot * 16) = xm[ks];
    }
  }
  __syncthreads();
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  {
    const int h = wave;  // one head per wavefront
    float* row = sS + h * SMAX;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) {
      float v = row[s];
      if (p.kpm && p.kpm[row0 + s]) v = -INFINITY;
      row[s] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float e = __expf(row[s] - mx);
      row[s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float spd = 0.f;
    const size_t prow = ((size_t)f * QH + h) * S;
    for (int s = lane; s < SP32; s += 64) {
      float pr = 0.f;
      if (s < S) {
        pr = row[s] * inv;
        p.probs[prow + s] = pr;
        if (p.drop_thresh) pr = dropout_keep(seed, (uint32_t)(prow + s), p.drop_thresh) ? pr * p.drop_scale : 0.f;
        sP[s * QH + h] = pr;
        spd += pr;
      }
      sPb[h * SMAX + s] = f32_to_bf16(pr);
    }
    spd = wave_sum(spd);
    if (lane == 0) sSp[h] = spd;
  }
  __syncthreads();
  if (p.wavg)
    for (int s = t; s < S; s += 512) {
      const float4 a = *(const float4*)(sP + s * QH), b = *(const float4*)(sP + s * QH + 4);
      p.wavg[row0 + s] = (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * (1.f / QH);
    }
  // sums: this wavefront's 64 channels = four 16-column blocks of half wave >> 1
  cq_f4 z[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) z[n] = cq_f4{0.f, 0.f, 0.f, 0.f};
  const int fsw = jrow | ((lg & 1) << 2);
  const char* half = sX + (wave >> 2) * HALF;
  for (int ks = 0; ks < SP32 / 32; ++ks) {
    uint4 pa = make_uint4(0u, 0u, 0u, 0u);
    if (lr < QH) pa = *(const uint4*)(sPb + lr * SMAX + ks * 32 + lg * 8);
    const int r0 = ks * 32 + 8 * lg + jrow;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int blk16 = (wave & 3) * 2 + n;
      const int cb = ((blk16 ^ fsw) << 5) + q * 8;
      const cq_bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) cq_bf4*)(half + r0 * 256 + cb));
      const cq_bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) cq_bf4*)(half + (r0 + 4) * 256 + cb));
      const cq_bf8 b = cq_bf8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      z[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cq_bf8*)&pa, b, z[n], 0, 0, 0);
    }
  }
  // D[i = head][j = channel]: lane holds heads 4 lg + rr, channel 64 wave + 16 n + lr
  if (lg < 2) {
    u16* zrow = (u16*)p.zext + (size_t)f * p.ldz;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) zrow[(4 * lg + rr) * QE + wave * 32 + 16 * n + lr] = f32_to_bf16(z[n][rr]);
  }
  if (t < QH) ((u16*)p.zext)[(size_t)f * p.ldz + QH * QE + t] = f32_to_bf16(sSp[t]);
}

// Backward of the frame core.  Inputs: d(zext) (d_z [h][c] and d(sum_s pd) [h]), the gradient of the head-averaged weights;
// outputs: d_u [h][c] = sum_s ds[h][s] (x_s + pos_s) and, accumulated over the layers in fp32,
// d(memory row s) = sum_h ds[h][s] u[h] + pd[h][s] d_z[h].
template <typename T>
__global__ __launch_bounds__(256) void cross_q1_bwd_kernel(CrossQ1Params p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = p.S, SP = (S + 3) & ~3;
  float* sS = sm;                 // [QH][SP]  d_z . x_s, then dP
  float* sDS = sS + QH * SP;      // [SP][QH]
  float* sPD = sDS + SP * QH;     // [SP][QH]
  float* sU = sPD + SP * QH;      // [4][QH * QE]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int f = blockIdx.x;
  const size_t row0 = (size_t)f * S;
  float ur[QH][4], dzr[QH][4];
#pragma unroll
  for (int h = 0; h < QH; ++h) {
    const float4 v = load4<T>(p.u, (size_t)f * QH * QE + h * QE + 4 * lane);
    ur[h][0] = v.x; ur[h][1] = v.y; ur[h][2] = v.z; ur[h][3] = v.w;
    const float4 g = load4<T>(p.dz, (size_t)f * p.ldz + h * QE + 4 * lane);
    dzr[h][0] = g.x; dzr[h][1] = g.y; dzr[h][2] = g.z; dzr[h][3] = g.w;
  }
  constexpr int KB = 8, KB2 = 4;
  const int hl = head_of_lane(lane);
  for (int s0 = wave * KB; s0 < S; s0 += 4 * KB) {
    float4 m[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) m[j] = load4<T>(p.mem, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      float part[QH];
#pragma unroll
      for (int h = 0; h < QH; ++h) part[h] = dot4(m[j], dzr[h]);
      const float tot = head_reduce8(part, lane);
      if (lane < QH && s0 + j < S) sS[hl * SP + s0 + j] = tot;
    }
  }
  __syncthreads();
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave + 4 * hh;
    float* row = sS + h * SP;
    const float dsp = Elem<T>::load(p.dz, (size_t)f * p.ldz + QH * QE + h);
    const size_t prow = ((size_t)f * QH + h) * S;
    float delta = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float pr = p.probs[prow + s];
      const bool keep = !p.drop_thresh || dropout_keep(seed, (uint32_t)(prow + s), p.drop_thresh);
      float g = row[s] + dsp;
      if (p.dwavg) g += p.dwavg[row0 + s] * (1.f / QH);
      const float dp = keep ? g * p.drop_scale : 0.f;
      row[s] = dp;
      delta += pr * dp;
      sPD[s * QH + h] = keep ? pr * p.drop_scale : 0.f;
    }
    delta = wave_sum(delta);
    for (int s = lane; s < S; s += 64) sDS[s * QH + h] = p.probs[prow + s] * (row[s] - delta);
  }
  __syncthreads();
  if (p.coef) {
    // deferred d(memory): d(mem row s) = sum over the layers and heads of ds[h][s] u[h] + pd[h][s] d_z[h] is ONE product
    // [S x 16 nl] . [16 nl x E] per frame (cross_q1_dmem_kernel, on the matrix pipe) once every layer has left its 16
    // coefficients per row here - instead of an fp32 read-modify-write of the whole [F*S][E] gradient per layer
    for (int s = t; s < S; s += 256) {
      const float4 a = *(const float4*)(sDS + s * QH), b = *(const float4*)(sDS + s * QH + 4);
      const float4 c = *(const float4*)(sPD + s * QH), d = *(const float4*)(sPD + s * QH + 4);
      uint4 lo, hi;
      lo.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
      lo.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
      lo.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
      lo.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
      hi.x = (uint32_t)f32_to_bf16(c.x) | ((uint32_t)f32_to_bf16(c.y) << 16);
      hi.y = (uint32_t)f32_to_bf16(c.z) | ((uint32_t)f32_to_bf16(c.w) << 16);
      hi.z = (uint32_t)f32_to_bf16(d.x) | ((uint32_t)f32_to_bf16(d.y) << 16);
      hi.w = (uint32_t)f32_to_bf16(d.z) | ((uint32_t)f32_to_bf16(d.w) << 16);
      uint4* dst = (uint4*)(p.coef + (row0 + s) * (size_t)p.coef_ld + p.coef_col);
      dst[0] = lo;
      dst[1] = hi;
    }
  }
  float du[QH][4];
#pragma unroll
  for (int h = 0; h < QH; ++h) du[h][0] = du[h][1] = du[h][2] = du[h][3] = 0.f;
  for (int s0 = wave * KB2; s0 < S; s0 += 4 * KB2) {
    float4 m[KB2], o[KB2];
#pragma unroll
    for (int j = 0; j < KB2; ++j) m[j] = load4<T>(p.mem, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
    if (p.pos) {
#pragma unroll
      for (int j = 0; j < KB2; ++j) {
        const float4 q = load4<T>(p.pos, (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
        m[j].x += q.x; m[j].y += q.y; m[j].z += q.z; m[j].w += q.w;
      }
    }
    if (p.accumulate) {
#pragma unroll
      for (int j = 0; j < KB2; ++j) o[j] = *(const float4*)(p.dmem + (row0 + min(s0 + j, S - 1)) * QE + 4 * lane);
    }
#pragma unroll
    for (int j = 0; j < KB2; ++j) {
      if (s0 + j >= S) break;  // wave-uniform
      const int s = s0 + j;
      const float4 a = *(const float4*)(sDS + s * QH), b = *(const float4*)(sDS + s * QH + 4);
      const float4 c = *(const float4*)(sPD + s * QH), d = *(const float4*)(sPD + s * QH + 4);
      const float ds[QH] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const float pd[QH] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
      float4 g = p.accumulate ? o[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int h = 0; h < QH; ++h) {
        du[h][0] += ds[h] * m[j].x; du[h][1] += ds[h] * m[j].y; du[h][2] += ds[h] * m[j].z; du[h][3] += ds[h] * m[j].w;
      }
      if (!p.dmem) continue;  // the memory needs no gradient (frozen encoder / no-grad memory): nothing is formed or stored
#pragma unroll
      for (int h = 0; h < QH; ++h) {
        g.x += ds[h] * ur[h][0] + pd[h] * dzr[h][0];
        g.y += ds[h] * ur[h][1] + pd[h] * dzr[h][1];
        g.z += ds[h] * ur[h][2] + pd[h] * dzr[h][2];
        g.w += ds[h] * ur[h][3] + pd[h] * dzr[h][3];
      }
      *(float4*)(p.dmem + (row0 + s) * QE + 4 * lane) = g;
    }
  }
#pragma unroll
  for (int h = 0; h < QH; ++h) *(float4*)(sU + wave * QH * QE + h * QE + 4 * lane) = make_float4(du[h][0], du[h][1], du[h][2], du[h][3]);
  __syncthreads();
  reduce_store_rows<T>(sU, p.du, (size_t)f * QH * QE, t);
}

// Backward of the frame core on the matrix pipe (bf16 mode, S <= SMAX, the memory gradient deferred or not wanted): the same two
// products as cross_q1_fwd_mfma_kernel with d_z in the place of u (no positional rows: d_z . x_s) and ds in the place of the
// probabilities (d_u = sum_s ds[h][s] (x_s + pos_s): the image holds the SUM of the memory and positional rows, added in fp32 and
// rounded once to bf16 - the rows are fetched from HBM once).  The softmax backward in between is the VALU kernel's.
template <int SMAX>
__global__ __launch_bounds__(512) void cross_q1_bwd_mfma_kernel(CrossQ1Params p) {
  constexpr int HALF = SMAX * 256;
  __shared__ __attribute__((aligned(16))) char sX[2 * HALF];
  __shared__ __attribute__((aligned(16))) float sS[QH * SMAX];    // d_z . x_s, then dP
  __shared__ __attribute__((aligned(16))) float sDS[SMAX * QH];   // [s][h]
  __shared__ __attribute__((aligned(16))) float sPD[SMAX * QH];   // [s][h]
  __shared__ __attribute__((aligned(16))) u16 sDSb[QH * SMAX];    // ds in bf16, [h][s]: A operand of d_u
  const int S = p.S, SP32 = (S + 31) & ~31;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, lg = lane >> 4, jrow = lr >> 2, q = lr & 3;
  const int f = blockIdx.x;
  const size_t row0 = (size_t)f * S;
  const u16* mem = (const u16*)p.mem;
  const u16* pos = (const u16*)p.pos;
  uint4 da[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    da[ks] = make_uint4(0u, 0u, 0u, 0u);
    if (lr < QH) da[ks] = *(const uint4*)((const u16*)p.dz + (size_t)f * p.ldz + lr * QE + ks * 32 + lg * 8);
  }
  for (int m0 = wave * 16; m0 < SP32; m0 += 128) {
    const int row = m0 + lr;
    const bool valid = row < S;
    uint4 xm[8], xp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      xm[ks] = make_uint4(0u, 0u, 0u, 0u);
      if (valid) xm[ks] = *(const uint4*)(mem + (row0 + row) * QE + ks * 32 + lg * 8);
    }
    if (pos) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        xp[ks] = make_uint4(0u, 0u, 0u, 0u);
        if (valid) xp[ks] = *(const uint4*)(pos + (row0 + row) * QE + ks * 32 + lg * 8);
      }
    }
    cq_f4 acc = cq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cq_bf8*)&da[ks], *(const cq_bf8*)&xm[ks], acc, 0, 0, 0);
    if (lg < 2 && valid) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) sS[(4 * lg + rr) * SMAX + row] = acc[rr];
    }
    const int swz = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 v = xm[ks];
      if (pos) {
        const uint32_t a_[4] = {xm[ks].x, xm[ks].y, xm[ks].z, xm[ks].w}, b_[4] = {xp[ks].x, xp[ks].y, xp[ks].z, xp[ks].w};
        uint32_t o_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(a_[e] << 16) + __uint_as_float(b_[e] << 16);
          const float hi = __uint_as_float(a_[e] & 0xFFFF0000u) + __uint_as_float(b_[e] & 0xFFFF0000u);
          o_[e] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
        }
        v = make_uint4(o_[0], o_[1], o_[2], o_[3]);
      }
      const int c = 4 * ks + lg, hf = c >> 4, c16 = c & 15;
      const int slot = (((c16 >> 1) ^ swz) << 1) | (c16 & 1);
      *(uint4*)(sX + hf * HALF + row * 256 + slot * 16) = v;
    }
  }
  __syncthreads();
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  {
    const int h = wave;  // one head per wavefront
    float* row = sS + h * SMAX;
    const float dsp = Elem<u16>::load(p.dz, (size_t)f * p.ldz + QH * QE + h);
    const size_t prow = ((size_t)f * QH + h) * S;
    float delta = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float pr = p.probs[prow + s];
      const bool keep = !p.drop_thresh || dropout_keep(seed, (uint32_t)(prow + s), p.drop_thresh);
      float g = row[s] + dsp;
      if (p.dwavg) g += p.dwavg[row0 + s] * (1.f / QH);
      const float dp = keep ? g * p.drop_scale : 0.f;
      row[s] = dp;
      delta += pr * dp;
      sPD[s * QH + h] = keep ? pr * p.drop_scale : 0.f;
    }
    delta = wave_sum(delta);
    for (int s = lane; s < SP32; s += 64) {
      float ds = 0.f;
      if (s < S) {
        ds = p.probs[prow + s] * (row[s] - delta);
        sDS[s * QH + h] = ds;
      }
      sDSb[h * SMAX + s] = f32_to_bf16(ds);
    }
  }
  __syncthreads();
  if (p.coef) {
    for (int s = t; s < S; s += 512) {
      const float4 a = *(const float4*)(sDS + s * QH), b = *(const float4*)(sDS + s * QH + 4);
      const float4 c = *(const float4*)(sPD + s * QH), d = *(const float4*)(sPD + s * QH + 4);
      uint4 lo, hi;
      lo.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
      lo.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
      lo.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
      lo.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
      hi.x = (uint32_t)f32_to_bf16(c.x) | ((uint32_t)f32_to_bf16(c.y) << 16);
      hi.y = (uint32_t)f32_to_bf16(c.z) | ((uint32_t)f32_to_bf16(c.w) << 16);
      hi.z = (uint32_t)f32_to_bf16(d.x) | ((uint32_t)f32_to_bf16(d.y) << 16);
      hi.w = (uint32_t)f32_to_bf16(d.z) | ((uint32_t)f32_to_bf16(d.w) << 16);
      uint4* dst = (uint4*)(p.coef + (row0 + s) * (size_t)p.coef_ld + p.coef_col);
      dst[0] = lo;
      dst[1] = hi;
    }
  }
  cq_f4 z[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) z[n] = cq_f4{0.f, 0.f, 0.f, 0.f};
  const int fsw = jrow | ((lg & 1) << 2);
  const char* half = sX + (wave >> 2) * HALF;
  for (int ks = 0; ks < SP32 / 32; ++ks) {
    uint4 pa = make_uint4(0u, 0u, 0u, 0u);
    if (lr < QH) pa = *(const uint4*)(sDSb + lr * SMAX + ks * 32 + lg * 8);
    const int r0 = ks * 32 + 8 * lg + jrow;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int blk16 = (wave & 3) * 2 + n;
      const int cb = ((blk16 ^ fsw) << 5) + q * 8;
      const cq_bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) cq_bf4*)(half + r0 * 256 + cb));
      const cq_bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) cq_bf4*)(half + (r0 + 4) * 256 + cb));
      const cq_bf8 b = cq_bf8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      z[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const cq_bf8*)&pa, b, z[n], 0, 0, 0);
    }
  }
  if (lg < 2) {
    u16* drow = (u16*)p.du + (size_t)f * QH * QE;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) drow[(4 * lg + rr) * QE + wave * 32 + 16 * n + lr] = f32_to_bf16(z[n][rr]);
  }
}

// d(memory) of all the layers at once (bf16 mode): per frame  D [S][E] = C [S][KP] . B [KP][E]  with C the coefficient rows the
// layers' backward kernels left (k = 16 * layer + {ds[h], 8 + pd[h]}) and B row k = u_layer[h] or d_z_layer[h] - 10 x 16 output
// tiles of v_mfma_f32_16x16x32_bf16 over KP / 32 k-steps.  C fragments are 16 contiguous bytes of a row (k along the lane's eight
// values: the natural A operand); B is staged in LDS k-major as it lies in HBM and read with the transposing ds_read_b64_tr_b16
// (the image and its 32-byte-block swizzle are the weight-gradient kernels': gemm_conv.hip wgrad_wide_body).  Wavefront w owns
// channels 64 w .. 64 w + 63: its twelve B fragments live in registers across the row tiles; results go through a per-wavefront
// LDS transposition so that every row is stored as 128 contiguous bytes.  One fp32-free pass writes the gradient in bf16.
constexpr int DM_MAXL = 8;
struct CrossDmemParams {
  const u16* coef;
  const u16* u[DM_MAXL];
  const u16* dz[DM_MAXL];
  u16* dmem;
  int F, S, NL, ldz, coef_ld;
};
typedef __bf16 dm_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 dm_bf4 __attribute__((ext_vector_type(4)));
typedef float dm_f4 __attribute__((ext_vector_type(4)));

#ifndef TD_DM_ABL
#define TD_DM_ABL 0  // timing ablations (tools/build_variant.sh; results WRONG): 1 no stores, 2 no coefficient loads, 4 no u / d_z loads
#endif
template <int KT>  // k-steps of 32 (KP = 32 KT)
__global__ __launch_bounds__(256) void cross_q1_dmem_kernel(CrossDmemParams p) {
  constexpr int KP = 32 * KT, HALF = KP * 256;   // one 128-channel half of the B image: KP rows of 256 bytes
  __shared__ __attribute__((aligned(16))) char sB[2 * HALF];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int f = blockIdx.x, S = p.S;
  const int lr = lane & 15, lg = lane >> 4, jrow = lr >> 2, q = lr & 3;
  const size_t row0 = (size_t)f * S;
  auto load_a = [&](int m0, int ks) {
    const int ra = m0 + lr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
#if !(TD_DM_ABL & 2)
    if (ra < S) v = *(const uint4*)(p.coef + (row0 + ra) * (size_t)p.coef_ld + ks * 32 + lg * 8);
#endif
    return v;
  };
  // coefficient fragments two row tiles ahead (requested before the image is built: they fly during the fill)
  uint4 a0[KT], a1[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks) {
    a0[ks] = load_a(0, ks);
    a1[ks] = load_a(16, ks);
  }
  // B image: row k, 32 chunks of 8 channels; chunk c of half hf sits at 16-byte slot (((c >> 1) ^ swz(k)) << 1) | (c & 1)
  {
    constexpr int NCH = KP * 32 / 256;  // chunks per thread
    uint4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = t + i * 256, k = idx >> 5, ch = idx & 31;
      const int l = k >> 4, kind = (k >> 3) & 1, h = k & 7;
      v[i] = make_uint4(0u, 0u, 0u, 0u);
#if !(TD_DM_ABL & 4)
      if (l < p.NL && p.u[l]) {
        const u16* src = kind ? p.dz[l] + (size_t)f * p.ldz + h * QE : p.u[l] + ((size_t)f * QH + h) * QE;
        v[i] = *(const uint4*)(src + ch * 8);
      }
#endif
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = t + i * 256, k = idx >> 5, ch = idx & 31;
      const int hf = ch >> 4, c16 = ch & 15;
      const int swz = (k & 3) | (((k >> 3) & 1) << 2);
      const int slot = (((c16 >> 1) ^ swz) << 1) | (c16 & 1);
      *(uint4*)(sB + hf * HALF + k * 256 + slot * 16) = v[i];
    }
  }
  __syncthreads();
  const int fsw = jrow | ((lg & 1) << 2);  // the swizzle of rows 8 lg + jrow (+ 4) of any k-step
  dm_bf8 bf[4][KT];
  {
    const char* half = sB + (wave >> 1) * HALF;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int blk16 = (wave & 1) * 4 + n;
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        const int r0 = ks * 32 + 8 * lg + jrow;
        const int cb = ((blk16 ^ fsw) << 5) + q * 8;
        const dm_bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) dm_bf4*)(half + r0 * 256 + cb));
        const dm_bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) dm_bf4*)(half + (r0 + 4) * 256 + cb));
        bf[n][ks] = dm_bf8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  }
  __syncthreads();  // every wavefront holds its fragments: the image is dead, its first 16 KiB become the four transposition buffers
  float* st = (float*)sB + wave * (16 * 64);
  auto tile = [&](int m0, const uint4 (&a4)[KT]) {
    dm_bf8 af[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) af[ks] = *(const dm_bf8*)&a4[ks];
    dm_f4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      acc[n] = dm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bf[n][ks], acc[n], 0, 0, 0);
    }
    // D[i][j]: lane holds rows i = 4 lg + rr, column j = lr of tile n  ->  st[row][16 n + lr] (row pitch 64 floats, rows
    // rotated by 16 floats per row group of four so that the four lane groups hit different banks)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) st[(4 * lg + rr) * 64 + ((16 * n + lr + 16 * lg) & 63)] = acc[n][rr];
    // (same wavefront reads what it wrote: LDS operations of a wavefront complete in order)
    // lane -> row lane / 4, channels 8 (lane % 4) .. + 8 of the first and of the second 32: an instruction stores 64 contiguous bytes per row
    const int row = lane >> 2, c0 = (lane & 3) * 8, rot = 16 * (row >> 2);
    const float4 x0 = *(const float4*)(st + row * 64 + ((c0 + rot) & 63)), x1 = *(const float4*)(st + row * 64 + ((c0 + 4 + rot) & 63));
    const float4 y0 = *(const float4*)(st + row * 64 + ((c0 + 32 + rot) & 63)), y1 = *(const float4*)(st + row * 64 + ((c0 + 36 + rot) & 63));
#if TD_DM_ABL & 1
    if (m0 + row < S && x0.x == 123.456f) {
#else
    if (m0 + row < S) {
#endif
      uint4 o0, o1;
      o0.x = (uint32_t)f32_to_bf16(x0.x) | ((uint32_t)f32_to_bf16(x0.y) << 16);
      o0.y = (uint32_t)f32_to_bf16(x0.z) | ((uint32_t)f32_to_bf16(x0.w) << 16);
      o0.z = (uint32_t)f32_to_bf16(x1.x) | ((uint32_t)f32_to_bf16(x1.y) << 16);
      o0.w = (uint32_t)f32_to_bf16(x1.z) | ((uint32_t)f32_to_bf16(x1.w) << 16);
      o1.x = (uint32_t)f32_to_bf16(y0.x) | ((uint32_t)f32_to_bf16(y0.y) << 16);
      o1.y = (uint32_t)f32_to_bf16(y0.z) | ((uint32_t)f32_to_bf16(y0.w) << 16);
      o1.z = (uint32_t)f32_to_bf16(y1.x) | ((uint32_t)f32_to_bf16(y1.y) << 16);
      o1.w = (uint32_t)f32_to_bf16(y1.z) | ((uint32_t)f32_to_bf16(y1.w) << 16);
      u16* dst = p.dmem + (row0 + m0 + row) * QE + wave * 64 + c0;
      *(uint4*)dst = o0;
      *(uint4*)(dst + 32) = o1;
    }
  };
  // ONE loop over row-tile pairs (the two prefetch sets alternate); tiles past S are skipped by the wave-uniform tests
  for (int m0 = 0; m0 < S; m0 += 32) {
    tile(m0, a0);
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) a0[ks] = load_a(m0 + 32, ks);
    if (m0 + 16 < S) tile(m0 + 16, a1);
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) a1[ks] = load_a(m0 + 48, ks);
  }
}

// Block-structured forms of a per-head projection W [E][E] (row j = output channel j of head j / hd): column block h of
// row j holds W[j][:] * alpha if j belongs to head h, zeros otherwise; `nb` more columns hold bias[j] in column (head of j).
// w_n [E][H*E + nb] and its transpose w_t [H*E + nb][E], both in the compute dtype.
template <typename T>
__global__ void head_blocks_expand_kernel(const float* W, const float* bias, float alpha, void* w_n, void* w_t, int E, int H, int nb) {
  const int ld = H * E + nb;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)E * ld) return;
  const int j = (int)(idx / ld), col = (int)(idx - (size_t)j * ld);
  const int hj = j / (E / H);
  float v = 0.f;
  if (col < H * E) {
    const int h = col / E, c = col - h * E;
    if (h == hj) v = W[(size_t)j * E + c] * alpha;
  } else if (col - H * E == hj) {
    v = bias[j];
  }
  Elem<T>::store(w_n, idx, v);
  Elem<T>::store(w_t, (size_t)col * E + j, v);
}

// The diagonal blocks of a dense gradient G [E][H*E + nb] (fp32): dW[j][c] = G[j][head(j)*E + c] * alpha, db[j] = G[j][H*E + head(j)]
__global__ void head_blocks_extract_kernel(const float* G, float alpha, float* dW, float* db, int E, int H, int nb) {
  const int ld = H * E + nb;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)E * E) return;
  const int j = (int)(idx / E), c = (int)(idx - (size_t)j * E);
  const int hj = j / (E / H);
  dW[idx] = G[(size_t)j * ld + hj * E + c] * alpha;
  if (db && c == 0) db[j] = G[(size_t)j * ld + H * E + hj];
}

static int fill_q1(CrossQ1Params& p, int F, int S, int H, int E, int ldz, float dropout_p, uint32_t seed, const uint32_t* ctr, const char* who) {
  TD_REQUIRE(F >= 1 && S >= 1, "%s: bad sizes F=%d S=%d", who, F, S);
  TD_REQUIRE(E == QE && H == QH, "%s: built for %d channels in %d heads (got E=%d H=%d)", who, QE, QH, E, H);
  TD_REQUIRE(ldz >= QH * QE + QH && ldz % 8 == 0, "%s: ldz=%d must be a multiple of 8 and >= %d", who, ldz, QH * QE + QH);
  TD_REQUIRE((double)F * QH * S < 4294967295.0, "%s: more than 2^32 probabilities", who);
  TD_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "%s: dropout_p must be in [0, 1)", who);
  p.F = F; p.S = S; p.ldz = ldz;
  p.drop_scale = 1.f;
  if (dropout_p > 0.f) {
    p.drop_thresh = (uint32_t)((double)dropout_p * 4294967296.0);
    if (p.drop_thresh == 0) p.drop_thresh = 1;
    p.drop_scale = 1.f / (1.f - dropout_p);
    p.seed = seed;
    p.seed_dev = ctr;
  }
  return TD_OK;
}

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace td
using namespace td;

extern "C" int td_cross_q1_fwd(const void* u, const void* mem, const void* pos, const uint8_t* key_pad, float* probs, float* wavg, void* zext,
                               int F, int S, int H, int E, int ldz, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter,
                               int dtype, td_stream_t stream) {
  TD_REQUIRE(u && mem && probs && zext, "td_cross_q1_fwd: null pointer");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_cross_q1_fwd: bad dtype %d", dtype);
  TD_REQUIRE(al16(u) && al16(mem) && al16(pos) && al16(zext), "td_cross_q1_fwd: rows must be 16-byte aligned");
  CrossQ1Params p;
  memset(&p, 0, sizeof(p));
  int rc = fill_q1(p, F, S, H, E, ldz, dropout_p, dropout_seed, dropout_counter, "td_cross_q1_fwd");
  if (rc) return rc;
  p.u = u; p.mem = mem; p.pos = pos; p.kpm = key_pad; p.probs = probs; p.wavg = wavg; p.zext = zext;
  const int SP = (S + 3) & ~3;
  const size_t lds = (size_t)(2 * QH * SP + 4 * QH * QE + QH) * sizeof(float);
  TD_REQUIRE(lds <= 64 * 1024, "td_cross_q1_fwd: S=%d too large for LDS", S);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_CROSS_Q1, dtype, 4.0 * F * S * QH * QE, st, F, S, QE, 0, 0, 0);
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    prof_set_bytes((double)F * S * QE * es * (pos ? 2.0 : 1.0) + (double)F * QH * S * 4.0 + (double)F * (QH * QE + ldz) * es);
  }
  // TD_CROSS_Q1_MFMA=0: the VALU kernel in bf16 too (A/B; the fp32 mode always runs it: exact-fp32 parity)
  static const int mfma_on = [] { const char* e_ = getenv("TD_CROSS_Q1_MFMA"); return e_ ? atoi(e_) : 1; }();
  if (dtype == TD_BF16 && mfma_on && S <= 128) cross_q1_fwd_mfma_kernel<128><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16 && mfma_on && S <= 160) cross_q1_fwd_mfma_kernel<160><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16 && mfma_on && S <= 256) cross_q1_fwd_mfma_kernel<256><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16) cross_q1_fwd_kernel<u16><<<F, 256, lds, st>>>(p);
  else cross_q1_fwd_kernel<float><<<F, 256, lds, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_cross_q1_fwd");
}

static int cross_q1_bwd_launch(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg, void* d_u,
                               float* d_mem, int accumulate, void* coef, int coef_ld, int coef_col, int F, int S, int H, int E, int ldz, float dropout_p,
                               uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream, const char* who) {
  TD_REQUIRE(u && mem && probs && d_zext && d_u, "%s: null pointer", who);
  TD_REQUIRE(d_mem || !accumulate, "%s: accumulate without d_mem", who);
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "%s: bad dtype %d", who, dtype);
  TD_REQUIRE(al16(u) && al16(mem) && al16(pos) && al16(d_zext) && al16(d_u) && al16(d_mem) && al16(coef), "%s: rows must be 16-byte aligned", who);
  CrossQ1Params p;
  memset(&p, 0, sizeof(p));
  int rc = fill_q1(p, F, S, H, E, ldz, dropout_p, dropout_seed, dropout_counter, who);
  if (rc) return rc;
  p.u = u; p.mem = mem; p.pos = pos; p.probs = (float*)probs; p.dz = d_zext; p.dwavg = dwavg; p.du = d_u; p.dmem = d_mem;
  p.accumulate = (accumulate && d_mem) ? 1 : 0;
  p.coef = (u16*)coef; p.coef_ld = coef_ld; p.coef_col = coef_col;
  const int SP = (S + 3) & ~3;
  const size_t lds = (size_t)(3 * QH * SP + 4 * QH * QE) * sizeof(float);
  TD_REQUIRE(lds <= 64 * 1024, "%s: S=%d too large for LDS", who, S);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_CROSS_Q1, dtype, 8.0 * F * S * QH * QE, st, F, S, QE, 0, 0, 1);
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    prof_set_bytes((double)F * S * QE * es * (pos ? 2.0 : 1.0) + (d_mem ? (double)F * S * QE * 4.0 * (accumulate ? 2.0 : 1.0) : 0.0) + (double)F * QH * S * 4.0 +
                   (double)F * (2 * QH * QE + ldz) * es + (coef ? (double)F * S * 32.0 : 0.0));
  }
  // the matrix-pipe kernel serves the bf16 launches that do not touch an fp32 d(memory) (deferred, or not wanted): 92 us against 129 us
  // for the VALU kernel in the same mode at 1 600 frames x 151 rows (212 us with the fp32 read-modify-write).  TD_CROSS_Q1_MFMA=0: off
  static const int mfma_on = [] { const char* e_ = getenv("TD_CROSS_Q1_MFMA"); return e_ ? atoi(e_) : 1; }();
  if (dtype == TD_BF16 && mfma_on && !d_mem && S <= 128) cross_q1_bwd_mfma_kernel<128><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16 && mfma_on && !d_mem && S <= 160) cross_q1_bwd_mfma_kernel<160><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16 && mfma_on && !d_mem && S <= 256) cross_q1_bwd_mfma_kernel<256><<<F, 512, 0, st>>>(p);
  else if (dtype == TD_BF16) cross_q1_bwd_kernel<u16><<<F, 256, lds, st>>>(p);
  else cross_q1_bwd_kernel<float><<<F, 256, lds, st>>>(p);
  if (prof) prof_end(st);
  return check_launch(who);
}

extern "C" int td_cross_q1_bwd(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg,
                               void* d_u, float* d_mem, int accumulate, int F, int S, int H, int E, int ldz, float dropout_p,
                               uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream) {
  return cross_q1_bwd_launch(u, mem, pos, probs, d_zext, dwavg, d_u, d_mem, accumulate, nullptr, 0, 0, F, S, H, E, ldz, dropout_p, dropout_seed, dropout_counter,
                             dtype, stream, "td_cross_q1_bwd");
}

extern "C" int td_cross_q1_bwd_coef(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg,
                                    void* d_u, void* coef, int coef_ld, int coef_col, int F, int S, int H, int E, int ldz, float dropout_p,
                                    uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream) {
  TD_REQUIRE(coef && dtype == TD_BF16, "td_cross_q1_bwd_coef: the deferred d(memory) is a bf16-mode path (fp32 accumulates in td_cross_q1_bwd)");
  TD_REQUIRE(coef_ld % 32 == 0 && coef_col % 16 == 0 && coef_col >= 0 && coef_col + 16 <= coef_ld, "td_cross_q1_bwd_coef: coef_ld=%d must be a multiple of 32, coef_col=%d a multiple of 16 inside it", coef_ld, coef_col);
  return cross_q1_bwd_launch(u, mem, pos, probs, d_zext, dwavg, d_u, nullptr, 0, coef, coef_ld, coef_col, F, S, H, E, ldz, dropout_p, dropout_seed,
                             dropout_counter, dtype, stream, "td_cross_q1_bwd_coef");
}

extern "C" int td_cross_q1_dmem(const void* coef, int coef_ld, const void* const* u, const void* const* d_zext, int n_layers, void* d_mem, int F, int S,
                                int H, int E, int ldz, int dtype, td_stream_t stream) {
  TD_REQUIRE(coef && u && d_zext && d_mem, "td_cross_q1_dmem: null pointer");
  TD_REQUIRE(dtype == TD_BF16, "td_cross_q1_dmem: bf16 only");
  TD_REQUIRE(E == QE && H == QH, "td_cross_q1_dmem: built for %d channels in %d heads (got E=%d H=%d)", QE, QH, E, H);
  TD_REQUIRE(F >= 1 && S >= 1 && n_layers >= 1 && n_layers <= DM_MAXL, "td_cross_q1_dmem: bad sizes F=%d S=%d layers=%d (at most %d)", F, S, n_layers, DM_MAXL);
  TD_REQUIRE(coef_ld % 32 == 0 && coef_ld >= 16 * n_layers && coef_ld <= 32 * 4, "td_cross_q1_dmem: coef_ld=%d must be a multiple of 32 holding 16 columns per layer", coef_ld);
  TD_REQUIRE(ldz >= QH * QE + QH && ldz % 8 == 0, "td_cross_q1_dmem: ldz=%d", ldz);
  CrossDmemParams p;
  memset(&p, 0, sizeof(p));
  p.coef = (const u16*)coef; p.dmem = (u16*)d_mem; p.F = F; p.S = S; p.NL = n_layers; p.ldz = ldz; p.coef_ld = coef_ld;
  for (int l = 0; l < n_layers; ++l) {
    TD_REQUIRE((u[l] == nullptr) == (d_zext[l] == nullptr), "td_cross_q1_dmem: layer %d: u and d_zext must both be given or both be NULL", l);
    TD_REQUIRE(al16(u[l]) && al16(d_zext[l]), "td_cross_q1_dmem: rows must be 16-byte aligned");
    p.u[l] = (const u16*)u[l];
    p.dz[l] = (const u16*)d_zext[l];
  }
  TD_REQUIRE(al16(coef) && al16(d_mem), "td_cross_q1_dmem: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_CROSS_Q1, dtype, 2.0 * F * S * coef_ld * QE, st, F, S, QE, 0, 0, 2);
    prof_set_bytes((double)F * S * (coef_ld + QE) * 2.0 + (double)F * n_layers * (QH * QE + ldz) * 2.0);
  }
  switch (coef_ld / 32) {
    case 1: cross_q1_dmem_kernel<1><<<F, 256, 0, st>>>(p); break;
    case 2: cross_q1_dmem_kernel<2><<<F, 256, 0, st>>>(p); break;
    case 3: cross_q1_dmem_kernel<3><<<F, 256, 0, st>>>(p); break;
    default: cross_q1_dmem_kernel<4><<<F, 256, 0, st>>>(p); break;
  }
  if (prof) prof_end(st);
  return check_launch("td_cross_q1_dmem");
}

extern "C" int td_head_blocks_expand(const float* W, const float* bias, float alpha, void* w_n, void* w_t, int E, int H, int dtype,
                                     td_stream_t stream) {
  TD_REQUIRE(W && w_n && w_t && E >= 1 && H >= 1 && E % H == 0, "td_head_blocks_expand: bad arguments");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_head_blocks_expand: bad dtype %d", dtype);
  const int nb = bias ? H : 0;
  const size_t n = (size_t)E * ((size_t)H * E + nb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) head_blocks_expand_kernel<u16><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(W, bias, alpha, w_n, w_t, E, H, nb);
  else head_blocks_expand_kernel<float><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(W, bias, alpha, w_n, w_t, E, H, nb);
  return check_launch("td_head_blocks_expand");
}

extern "C" int td_head_blocks_extract(const float* G, float alpha, float* dW, float* db, int E, int H, td_stream_t stream) {
  TD_REQUIRE(G && dW && E >= 1 && H >= 1 && E % H == 0, "td_head_blocks_extract: bad arguments");
  const size_t n = (size_t)E * E;
  head_blocks_extract_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(G, alpha, dW, db, E, H, db ? H : 0);
  return check_launch("td_head_blocks_extract");
}
