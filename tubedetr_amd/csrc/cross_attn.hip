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
// The kernels are HBM-bound by the memory rows; the arithmetic is fp32 on the VALU in both dtypes (24 FLOP per loaded byte).
//
// Layout of a workgroup (256 threads, one frame): lane l of every wavefront owns channels 4l .. 4l+3; wavefront w walks the
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
  if (dtype == TD_BF16) cross_q1_fwd_kernel<u16><<<F, 256, lds, st>>>(p);
  else cross_q1_fwd_kernel<float><<<F, 256, lds, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_cross_q1_fwd");
}

extern "C" int td_cross_q1_bwd(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg,
                               void* d_u, float* d_mem, int accumulate, int F, int S, int H, int E, int ldz, float dropout_p,
                               uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream) {
  TD_REQUIRE(u && mem && probs && d_zext && d_u, "td_cross_q1_bwd: null pointer");
  TD_REQUIRE(d_mem || !accumulate, "td_cross_q1_bwd: accumulate without d_mem");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_cross_q1_bwd: bad dtype %d", dtype);
  TD_REQUIRE(al16(u) && al16(mem) && al16(pos) && al16(d_zext) && al16(d_u) && al16(d_mem), "td_cross_q1_bwd: rows must be 16-byte aligned");
  CrossQ1Params p;
  memset(&p, 0, sizeof(p));
  int rc = fill_q1(p, F, S, H, E, ldz, dropout_p, dropout_seed, dropout_counter, "td_cross_q1_bwd");
  if (rc) return rc;
  p.u = u; p.mem = mem; p.pos = pos; p.probs = (float*)probs; p.dz = d_zext; p.dwavg = dwavg; p.du = d_u; p.dmem = d_mem;
  p.accumulate = (accumulate && d_mem) ? 1 : 0;
  const int SP = (S + 3) & ~3;
  const size_t lds = (size_t)(3 * QH * SP + 4 * QH * QE) * sizeof(float);
  TD_REQUIRE(lds <= 64 * 1024, "td_cross_q1_bwd: S=%d too large for LDS", S);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_CROSS_Q1, dtype, 8.0 * F * S * QH * QE, st, F, S, QE, 0, 0, 1);
    const double es = dtype == TD_BF16 ? 2.0 : 4.0;
    prof_set_bytes((double)F * S * QE * es * (pos ? 2.0 : 1.0) + (d_mem ? (double)F * S * QE * 4.0 * (accumulate ? 2.0 : 1.0) : 0.0) + (double)F * QH * S * 4.0 +
                   (double)F * (2 * QH * QE + ldz) * es);
  }
  if (dtype == TD_BF16) cross_q1_bwd_kernel<u16><<<F, 256, lds, st>>>(p);
  else cross_q1_bwd_kernel<float><<<F, 256, lds, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_cross_q1_bwd");
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
