// Multi-head attention core (softmax(scale*QK^T + key padding) V) forward / backward for the small
// attention problems of TubeDETR: per-frame visual-text self-attention (S = hw+L = 151 keys, batch =
// slow clips), temporal self-attention (T = 100), time-aligned cross-attention (1 query per frame vs
// its 151 keys).  Head dim 32.  K/V of one (batch, head) live in LDS; one wavefront owns one query
// row: lanes span keys, softmax statistics by wavefront shuffles, PV with lanes re-mapped to
// (channel, key-half).  fp32 math on the VALU in both dtypes (inputs/outputs are T).
// Reference: torch nn.MultiheadAttention at models/transformer.py:613,638-640,661-662,713-740.
#include "td_common.h"
#include <stdlib.h>

namespace td {

// Workgroup -> (batch element, head).  Consecutive workgroup ids go round the eight XCDs, and each XCD has its own L2: with
// heads as the fast index the eight heads of a batch element - eight 64-byte column blocks of the SAME 512-byte rows - land on eight
// different L2s, every one of which fetches whole 128-byte lines for its half.  Here the ids an XCD sees (id % 8 fixed) walk the
// heads of ONE element before moving on: a row's lines are fetched into one L2 and hit there by the other heads.  Needs the batch
// to be a multiple of eight (else the plain order); TD_MHA_XCD_MAP=0: the plain order (A/B).
__device__ __forceinline__ int xcd_bh(int id, int BH, int H, int on) {
  const int B = BH / H;
  if (!on || (B & 7)) return id;
  const int x = id & 7, j = id >> 3;
  return ((j / H) * 8 + x) * H + (j % H);
}


constexpr int HD = 32;
constexpr int KJ = 8;    // keys per lane: Lk <= 64*KJ
constexpr int QT = 32;   // queries per forward workgroup

struct MhaParams {
  const void *q, *k, *v, *dout;
  const uint8_t* kpm;
  void *out, *dq, *dk, *dv;
  float* probs;
  const float* dwavg;
  float* ds_ws;
  float* stats;  // lean path: [B*H][Lq][4] = (row max of the scaled masked scores, 1 / sum exp, delta = dO . O, unused)
  int B, H, Lq, Lk, ldq, ldk, ldv, ldo;
  int xcd_map;  // workgroup ids walk the heads of one batch element per XCD (xcd_bh)
  float scale;
  uint32_t drop_thresh;
  float drop_scale;
  uint32_t seed;
  const uint32_t* seed_dev;
};

#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename T, int HDV>
__global__ __launch_bounds__(256) void mha_fwd_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LS = HDV + 1;  // odd row stride: conflict-free column walks
  const int Lk = p.Lk, Lq = p.Lq;
  float* sK = sm;
  float* sV = sK + Lk * LS;
  float* sP = sV + Lk * LS;  // [4][Lk]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lk * HDV; idx += 256) {
    int kk = idx / HDV, d = idx - kk * HDV;
    sK[kk * LS + d] = Elem<T>::load(p.k, (size_t)(b * Lk + kk) * p.ldk + h * HDV + d);
    sV[kk * LS + d] = Elem<T>::load(p.v, (size_t)(b * Lk + kk) * p.ldv + h * HDV + d);
  }
  __syncthreads();
  const int q0 = blockIdx.y * QT;
  const int q1 = min(Lq, q0 + QT);
  float* myP = sP + wave * Lk;
  for (int qi = q0 + wave; qi < q1; qi += 4) {
    float qv[HDV];
    const size_t qoff = (size_t)(b * Lq + qi) * p.ldq + h * HDV;
#pragma unroll
    for (int d = 0; d < HDV; ++d) qv[d] = Elem<T>::load(p.q, qoff + d);
    float s[KJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      int kk = lane + 64 * j;
      s[j] = -INFINITY;
      if (kk < Lk) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < HDV; ++d) dot += qv[d] * sK[kk * LS + d];
        dot *= p.scale;
        if (p.kpm && p.kpm[b * Lk + kk]) dot = -INFINITY;
        s[j] = dot;
      }
      mx = fmaxf(mx, s[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      s[j] = (lane + 64 * j < Lk) ? __expf(s[j] - mx) : 0.f;
      sum += s[j];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const size_t prow = ((size_t)bh * Lq + qi) * Lk;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      int kk = lane + 64 * j;
      if (kk < Lk) {
        float pr = s[j] * inv;
        p.probs[prow + kk] = pr;
        if (p.drop_thresh) pr = dropout_keep(effective_seed(p.seed, p.seed_dev), (uint32_t)(prow + kk), p.drop_thresh) ? pr * p.drop_scale : 0.f;
        myP[kk] = pr;
      }
    }
    LDS_FENCE();
    // P V: lanes = (channel, key part); 64 / HDV parts, combined by shuffles
    constexpr int PARTS = 64 / HDV;
    const int d = lane % HDV, part = lane / HDV;
    float acc = 0.f;
    for (int kk = part; kk < Lk; kk += PARTS) acc += myP[kk] * sV[kk * LS + d];
    if (PARTS == 2) acc += __shfl_xor(acc, 32, 64);
    if (part == 0) Elem<T>::store(p.out, (size_t)(b * Lq + qi) * p.ldo + h * HDV + d, acc);
    LDS_FENCE();
  }
}

__global__ void avg_heads_kernel(const float* probs, float* wavg, int B, int H, int Lq, int Lk, uint32_t drop_thresh,
                                 float drop_scale, uint32_t seed, const uint32_t* seed_dev) {
  const size_t n = (size_t)B * Lq * Lk;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const size_t per = (size_t)Lq * Lk;
  const size_t b = idx / per, rem = idx - b * per;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    size_t pi = (b * H + h) * per + rem;
    float pr = probs[pi];
    if (drop_thresh) pr = dropout_keep(effective_seed(seed, seed_dev), (uint32_t)pi, drop_thresh) ? pr * drop_scale : 0.f;
    acc += pr;
  }
  wavg[idx] = acc / H;
}

// Backward, kernel A: grid (batch*head, query tiles).  One wavefront per query row: dP = dO V^T (+ dwavg/H),
// dS = P * (dP - sum(P dP)) written to ds_ws, dQ = scale * dS K.
template <typename T, int HDV>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LS = HDV + 1;
  const int Lk = p.Lk, Lq = p.Lq;
  float* sK = sm;
  float* sV = sK + Lk * LS;
  float* sP = sV + Lk * LS;  // [4][Lk]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lk * HDV; idx += 256) {
    int kk = idx / HDV, d = idx - kk * HDV;
    sK[kk * LS + d] = Elem<T>::load(p.k, (size_t)(b * Lk + kk) * p.ldk + h * HDV + d);
    sV[kk * LS + d] = Elem<T>::load(p.v, (size_t)(b * Lk + kk) * p.ldv + h * HDV + d);
  }
  __syncthreads();
  float* myP = sP + wave * Lk;
  const float invH = 1.f / p.H;
  const int q0 = blockIdx.y * QT;
  const int q1 = min(Lq, q0 + QT);
  for (int qi = q0 + wave; qi < q1; qi += 4) {
    float dov[HDV];
    const size_t ooff = (size_t)(b * Lq + qi) * p.ldo + h * HDV;
#pragma unroll
    for (int d = 0; d < HDV; ++d) dov[d] = Elem<T>::load(p.dout, ooff + d);
    const size_t prow = ((size_t)bh * Lq + qi) * Lk;
    float dp[KJ], pr[KJ];
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      int kk = lane + 64 * j;
      dp[j] = pr[j] = 0.f;
      if (kk < Lk) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < HDV; ++d) dot += dov[d] * sV[kk * LS + d];
        if (p.dwavg) dot += p.dwavg[((size_t)b * Lq + qi) * Lk + kk] * invH;
        if (p.drop_thresh) dot = dropout_keep(effective_seed(p.seed, p.seed_dev), (uint32_t)(prow + kk), p.drop_thresh) ? dot * p.drop_scale : 0.f;
        dp[j] = dot;
        pr[j] = p.probs[prow + kk];
        delta += pr[j] * dot;
      }
    }
    delta = wave_sum(delta);
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      int kk = lane + 64 * j;
      if (kk < Lk) {
        float ds = pr[j] * (dp[j] - delta);
        p.ds_ws[prow + kk] = ds;
        myP[kk] = ds;
      }
    }
    LDS_FENCE();
    constexpr int PARTS = 64 / HDV;
    const int d = lane % HDV, part = lane / HDV;
    float acc = 0.f;
    for (int kk = part; kk < Lk; kk += PARTS) acc += myP[kk] * sK[kk * LS + d];
    if (PARTS == 2) acc += __shfl_xor(acc, 32, 64);
    if (part == 0) Elem<T>::store(p.dq, (size_t)(b * Lq + qi) * p.ldq + h * HDV + d, acc * p.scale);
    LDS_FENCE();
  }
}

// Backward, kernel B: grid (batch*head, tiles of 64 keys).  Thread = (key, 8-channel slice); loops over all queries:
// dK = scale * dS^T Q, dV = dropout(P)^T dO.  Q / dO rows are wave-uniform LDS broadcasts, dS / P columns are
// coalesced over keys.
template <typename T, int HDV>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int CH = HDV / 4;  // channels per thread: the 4 wavefronts of the workgroup split the head dim
  const int Lk = p.Lk, Lq = p.Lq;
  float* sQ = sm;              // [Lq][HDV]
  float* sO = sm + Lq * HDV;   // [Lq][HDV]
  const int t = threadIdx.x;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lq * HDV; idx += 256) {
    int qq = idx / HDV, d = idx - qq * HDV;
    sQ[idx] = Elem<T>::load(p.q, (size_t)(b * Lq + qq) * p.ldq + h * HDV + d);
    sO[idx] = Elem<T>::load(p.dout, (size_t)(b * Lq + qq) * p.ldo + h * HDV + d);
  }
  __syncthreads();
  const int kk = blockIdx.y * 64 + (t & 63);
  const int part = t >> 6;  // wave-uniform: channels part*CH .. part*CH+CH-1
  if (kk >= Lk) return;
  float dK[CH], dV[CH];
#pragma unroll
  for (int d = 0; d < CH; ++d) dK[d] = dV[d] = 0.f;
  for (int qq = 0; qq < Lq; ++qq) {
    const size_t pi = ((size_t)bh * Lq + qq) * Lk + kk;
    float ds = p.ds_ws[pi];
    float pr = p.probs[pi];
    if (p.drop_thresh) pr = dropout_keep(effective_seed(p.seed, p.seed_dev), (uint32_t)pi, p.drop_thresh) ? pr * p.drop_scale : 0.f;
    const float* qrow = sQ + qq * HDV + part * CH;
    const float* orow = sO + qq * HDV + part * CH;
#pragma unroll
    for (int d = 0; d < CH; ++d) {
      dK[d] += ds * qrow[d];
      dV[d] += pr * orow[d];
    }
  }
  const size_t ko = (size_t)(b * Lk + kk) * p.ldk + h * HDV + part * CH, vo = (size_t)(b * Lk + kk) * p.ldv + h * HDV + part * CH;
#pragma unroll
  for (int d = 0; d < CH; ++d) {
    Elem<T>::store(p.dk, ko + d, dK[d] * p.scale);
    Elem<T>::store(p.dv, vo + d, dV[d]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 MFMA path (v_mfma_f32_16x16x32_bf16), used for every bf16 problem with Lk <= 256.  One wavefront owns 16 query
// rows.  Scores are produced TRANSPOSED, S^T[key][query] = K Q^T (head dim 32 = exactly one MFMA K step), so that the
// accumulator layout (lane = query column, 4 consecutive keys per register quad) is already the B-operand layout of
// the second product O^T = V^T P^T / dQ^T = K^T dS^T: probabilities never leave registers.  Two 16-key tiles form one
// 32-deep reduction step in the permuted key order {tile0: 4g..4g+3, tile1: 4g..4g+3}; the A operand (V^T or K^T, kept
// transposed in LDS as [channel][key], row stride = odd multiple of 16 B -> conflict-free ds_read_b64) is read in the
// same order.  Softmax statistics need two cross-lane steps (xor 16, 32).  fp32 accumulation; P / dS are rounded to
// bf16 only as MFMA operands, the stored probabilities stay fp32.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, c, 0, 0, 0);
}
__device__ __forceinline__ uint4 ldg16(const u16* p) { return *(const uint4*)p; }

// rows [n_rows][32] of one head (global, row stride ld) -> LDS transposed [32][stride], rows >= n_valid zero-filled
__device__ __forceinline__ void stage_transposed(u16* dst, int stride, const u16* src, size_t ld, int n_valid, int n_rows, int t, int nthr) {
  for (int idx = t; idx < n_rows * 4; idx += nthr) {
    const int row = idx >> 2, c = idx & 3;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (row < n_valid) val = ldg16(src + (size_t)row * ld + c * 8);
    const u16* e = (const u16*)&val;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[(c * 8 + i) * stride + row] = e[i];
  }
}

// LEAN: nothing of size Lq x Lk leaves the kernel - the probabilities are not stored (the backward recomputes them from
// q, k and the two softmax statistics per row written to p.stats), for attention whose weights nobody reads (the encoder).
template <int NT, bool LEAN = false>
__global__ __launch_bounds__(256) void mha_fwd_mfma_kernel(MhaParams p) {
  constexpr int KP = NT * 16, VS = KP + 8;
  __shared__ __attribute__((aligned(16))) u16 sVt[32 * VS];
  __shared__ __attribute__((aligned(16))) float sBias[KP];
  const int Lk = p.Lk, Lq = p.Lq, nthr = blockDim.x, nw = nthr >> 6;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  const u16* Q = (const u16*)p.q;
  const u16* K = (const u16*)p.k;
  const u16* V = (const u16*)p.v;
  const int qb = (blockIdx.y * nw + wave) * 16, qj = qb + li;
  const bool qv = qj < Lq;
  uint4 bq = make_uint4(0, 0, 0, 0);
  if (qv) bq = ldg16(Q + (size_t)(b * Lq + qj) * p.ldq + h * HD + 8 * g);
  uint4 ak[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int key = tt * 16 + li;
    ak[tt] = make_uint4(0, 0, 0, 0);
    if (key < Lk) ak[tt] = ldg16(K + (size_t)(b * Lk + key) * p.ldk + h * HD + 8 * g);
  }
  stage_transposed(sVt, VS, V + (size_t)b * Lk * p.ldv + h * HD, p.ldv, Lk, KP, t, nthr);
  for (int key = t; key < KP; key += nthr) sBias[key] = (key < Lk && !(p.kpm && p.kpm[b * Lk + key])) ? 0.f : -INFINITY;
  __syncthreads();
  if (qb >= Lq) return;
  f32x4 s[NT];
  float mx = -INFINITY;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    s[tt] = mfma_bf16(ak[tt], bq, z);
    const float4 bias = *(const float4*)&sBias[tt * 16 + 4 * g];
    s[tt][0] = s[tt][0] * p.scale + bias.x;
    s[tt][1] = s[tt][1] * p.scale + bias.y;
    s[tt][2] = s[tt][2] * p.scale + bias.z;
    s[tt][3] = s[tt][3] * p.scale + bias.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tt][r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[tt][r] = __expf(s[tt][r] - mx);
      sum += s[tt][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  const size_t prow = ((size_t)bh * Lq + qj) * Lk;
  if (LEAN && qv && g == 0) *(float2*)(p.stats + ((size_t)bh * Lq + qj) * 4) = make_float2(mx, inv);
  f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    bf16x8 bp;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = (2 * u + hh) * 16 + 4 * g + r;
        float pr = s[2 * u + hh][r] * inv;
        if (!LEAN && qv && key < Lk) p.probs[prow + key] = pr;
        if (p.drop_thresh) pr = dropout_keep(seed, (uint32_t)(prow + key), p.drop_thresh) ? pr * p.drop_scale : 0.f;
        bp[hh * 4 + r] = (__bf16)pr;
      }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      uint4 av;
      const uint2 lo = *(const uint2*)&sVt[(m * 16 + li) * VS + (2 * u) * 16 + 4 * g];
      const uint2 hi = *(const uint2*)&sVt[(m * 16 + li) * VS + (2 * u + 1) * 16 + 4 * g];
      av.x = lo.x; av.y = lo.y; av.z = hi.x; av.w = hi.y;
      o[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&av, bp, o[m], 0, 0, 0);
    }
  }
  if (qv) {
    u16* O = (u16*)p.out + (size_t)(b * Lq + qj) * p.ldo + h * HD + 4 * g;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (__bf16)o[m][r];
      *(bf16x4*)(O + m * 16) = w;
    }
  }
}

// backward A (MFMA): dP^T = V dO^T, dS^T = P^T * (dP^T - delta), dQ^T = K^T dS^T; writes dS (fp32) for kernel B.
template <int NT>
__global__ __launch_bounds__(256) void mha_bwd_dq_mfma_kernel(MhaParams p) {
  constexpr int KP = NT * 16, VS = KP + 8;
  __shared__ __attribute__((aligned(16))) u16 sKt[32 * VS];
  const int Lk = p.Lk, Lq = p.Lq, nthr = blockDim.x, nw = nthr >> 6;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  const u16* K = (const u16*)p.k;
  const u16* V = (const u16*)p.v;
  const u16* DO = (const u16*)p.dout;
  const int qb = (blockIdx.y * nw + wave) * 16, qj = qb + li;
  const bool qv = qj < Lq;
  uint4 bdo = make_uint4(0, 0, 0, 0);
  if (qv) bdo = ldg16(DO + (size_t)(b * Lq + qj) * p.ldo + h * HD + 8 * g);
  uint4 av[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int key = tt * 16 + li;
    av[tt] = make_uint4(0, 0, 0, 0);
    if (key < Lk) av[tt] = ldg16(V + (size_t)(b * Lk + key) * p.ldv + h * HD + 8 * g);
  }
  stage_transposed(sKt, VS, K + (size_t)b * Lk * p.ldk + h * HD, p.ldk, Lk, KP, t, nthr);
  __syncthreads();
  if (qb >= Lq) return;
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  const size_t prow = ((size_t)bh * Lq + qj) * Lk;
  const size_t wrow = ((size_t)b * Lq + qj) * Lk;
  const float invH = 1.f / p.H;
  f32x4 dp[NT], pr[NT];
  float delta = 0.f;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    dp[tt] = mfma_bf16(av[tt], bdo, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = tt * 16 + 4 * g + r;
      float prv = 0.f, d = 0.f;
      if (qv && key < Lk) {
        prv = p.probs[prow + key];
        d = dp[tt][r];
        if (p.dwavg) d += p.dwavg[wrow + key] * invH;
        if (p.drop_thresh) d = dropout_keep(seed, (uint32_t)(prow + key), p.drop_thresh) ? d * p.drop_scale : 0.f;
      }
      pr[tt][r] = prv;
      dp[tt][r] = d;
      delta += prv * d;
    }
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    bf16x8 bs;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tt = 2 * u + hh, key = tt * 16 + 4 * g + r;
        const float ds = pr[tt][r] * (dp[tt][r] - delta);
        if (qv && key < Lk) p.ds_ws[prow + key] = ds;
        bs[hh * 4 + r] = (__bf16)ds;
      }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      uint4 ak;
      const uint2 lo = *(const uint2*)&sKt[(m * 16 + li) * VS + (2 * u) * 16 + 4 * g];
      const uint2 hi = *(const uint2*)&sKt[(m * 16 + li) * VS + (2 * u + 1) * 16 + 4 * g];
      ak.x = lo.x; ak.y = lo.y; ak.z = hi.x; ak.w = hi.y;
      dq[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&ak, bs, dq[m], 0, 0, 0);
    }
  }
  if (qv) {
    u16* DQ = (u16*)p.dq + (size_t)(b * Lq + qj) * p.ldq + h * HD + 4 * g;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (__bf16)(dq[m][r] * p.scale);
      *(bf16x4*)(DQ + m * 16) = w;
    }
  }
}

// backward B (MFMA): one wavefront per 16-key tile; dV^T = dO^T dropout(P), dK^T = scale * Q^T dS, reduction over queries
// in chunks of 32 (A = dO^T / Q^T from transposed LDS, B = P / dS columns read straight from the fp32 tensors).
__global__ __launch_bounds__(256) void mha_bwd_dkv_mfma_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lk = p.Lk, Lq = p.Lq, nthr = blockDim.x, nw = nthr >> 6;
  const int QP = cdiv(Lq, 32) * 32, QS = QP + 8;
  u16* sQt = (u16*)sm;
  u16* sOt = sQt + 32 * QS;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  stage_transposed(sQt, QS, (const u16*)p.q + (size_t)b * Lq * p.ldq + h * HD, p.ldq, Lq, QP, t, nthr);
  stage_transposed(sOt, QS, (const u16*)p.dout + (size_t)b * Lq * p.ldo + h * HD, p.ldo, Lq, QP, t, nthr);
  __syncthreads();
  const int kt = blockIdx.y * nw + wave;
  if (kt * 16 >= Lk) return;
  const int key = kt * 16 + li;
  const bool kv = key < Lk;
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  f32x4 dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int qc = 0; qc < QP; qc += 32) {
    bf16x8 bp, bs;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int qq = qc + 8 * g + i;
      float pr = 0.f, ds = 0.f;
      if (kv && qq < Lq) {
        const size_t pi = ((size_t)bh * Lq + qq) * Lk + key;
        pr = p.probs[pi];
        ds = p.ds_ws[pi];
        if (p.drop_thresh) pr = dropout_keep(seed, (uint32_t)pi, p.drop_thresh) ? pr * p.drop_scale : 0.f;
      }
      bp[i] = (__bf16)pr;
      bs[i] = (__bf16)ds;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bf16x8 ao = *(const bf16x8*)&sOt[(m * 16 + li) * QS + qc + 8 * g];
      const bf16x8 aq = *(const bf16x8*)&sQt[(m * 16 + li) * QS + qc + 8 * g];
      dv[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao, bp, dv[m], 0, 0, 0);
      dk[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bs, dk[m], 0, 0, 0);
    }
  }
  if (kv) {
    u16* DK = (u16*)p.dk + (size_t)(b * Lk + key) * p.ldk + h * HD + 4 * g;
    u16* DV = (u16*)p.dv + (size_t)(b * Lk + key) * p.ldv + h * HD + 4 * g;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x4 wk, wv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wk[r] = (__bf16)(dk[m][r] * p.scale);
        wv[r] = (__bf16)dv[m][r];
      }
      *(bf16x4*)(DK + m * 16) = wk;
      *(bf16x4*)(DV + m * 16) = wv;
    }
  }
}

// ---- lean backward (no Lq x Lk tensor is read or written) ----------------------------------------------------------
// kernel A: P recomputed from q, k and the row statistics; delta = sum_k P dP = dO . O (also under dropout: O = Pd V),
// written to p.stats[..][2] for kernel B; dQ^T = K^T dS^T as in the probs-based kernel.
template <int NT>
__global__ __launch_bounds__(256) void mha_bwd_dq_lean_kernel(MhaParams p) {
  constexpr int KP = NT * 16, VS = KP + 8;
  __shared__ __attribute__((aligned(16))) u16 sKt[32 * VS];
  __shared__ __attribute__((aligned(16))) float sBias[KP];
  const int Lk = p.Lk, Lq = p.Lq, nthr = blockDim.x, nw = nthr >> 6;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  const u16* Q = (const u16*)p.q;
  const u16* K = (const u16*)p.k;
  const u16* V = (const u16*)p.v;
  const u16* DO = (const u16*)p.dout;
  const u16* O = (const u16*)p.out;
  const int qb = (blockIdx.y * nw + wave) * 16, qj = qb + li;
  const bool qv = qj < Lq;
  uint4 bdo = make_uint4(0, 0, 0, 0), bq = make_uint4(0, 0, 0, 0), bo = make_uint4(0, 0, 0, 0);
  float2 st2 = make_float2(0.f, 0.f);
  if (qv) {
    bdo = ldg16(DO + (size_t)(b * Lq + qj) * p.ldo + h * HD + 8 * g);
    bo = ldg16(O + (size_t)(b * Lq + qj) * p.ldo + h * HD + 8 * g);
    bq = ldg16(Q + (size_t)(b * Lq + qj) * p.ldq + h * HD + 8 * g);
    st2 = *(const float2*)(p.stats + ((size_t)bh * Lq + qj) * 4);
  }
  stage_transposed(sKt, VS, K + (size_t)b * Lk * p.ldk + h * HD, p.ldk, Lk, KP, t, nthr);
  for (int key = t; key < KP; key += nthr) sBias[key] = (key < Lk && !(p.kpm && p.kpm[b * Lk + key])) ? 0.f : -INFINITY;
  __syncthreads();
  if (qb >= Lq) return;
  // delta = dO . O over the 32 channels of this head: 8 per lane group, summed over the four groups
  float delta = 0.f;
  {
    const u16* a = (const u16*)&bdo;
    const u16* c = (const u16*)&bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) delta += bf16_to_f32(a[i]) * bf16_to_f32(c[i]);
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  if (qv && g == 0) p.stats[((size_t)bh * Lq + qj) * 4 + 2] = delta;
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  const size_t prow = ((size_t)bh * Lq + qj) * Lk;
  f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    bf16x8 bs;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int tt = 2 * u + hh, key0 = tt * 16 + li;
      uint4 ak = make_uint4(0, 0, 0, 0), av = make_uint4(0, 0, 0, 0);
      if (key0 < Lk) {
        ak = ldg16(K + (size_t)(b * Lk + key0) * p.ldk + h * HD + 8 * g);
        av = ldg16(V + (size_t)(b * Lk + key0) * p.ldv + h * HD + 8 * g);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 sc = mfma_bf16(ak, bq, z);   // S^T[key 4g+r][query li]
      const f32x4 dp = mfma_bf16(av, bdo, z);  // dP^T
      const float4 bias = *(const float4*)&sBias[tt * 16 + 4 * g];
      const float bz[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = tt * 16 + 4 * g + r;
        float ds = 0.f;
        if (qv && key < Lk) {
          const float pr = __expf(sc[r] * p.scale + bz[r] - st2.x) * st2.y;
          float d = dp[r];
          if (p.drop_thresh) d = dropout_keep(seed, (uint32_t)(prow + key), p.drop_thresh) ? d * p.drop_scale : 0.f;
          ds = pr * (d - delta);
        }
        bs[hh * 4 + r] = (__bf16)ds;
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      uint4 ak;
      const uint2 lo = *(const uint2*)&sKt[(m * 16 + li) * VS + (2 * u) * 16 + 4 * g];
      const uint2 hi = *(const uint2*)&sKt[(m * 16 + li) * VS + (2 * u + 1) * 16 + 4 * g];
      ak.x = lo.x; ak.y = lo.y; ak.z = hi.x; ak.w = hi.y;
      dq[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&ak, bs, dq[m], 0, 0, 0);
    }
  }
  if (qv) {
    u16* DQ = (u16*)p.dq + (size_t)(b * Lq + qj) * p.ldq + h * HD + 4 * g;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (__bf16)(dq[m][r] * p.scale);
      *(bf16x4*)(DQ + m * 16) = w;
    }
  }
}

// kernel B: one wavefront per 16-key tile, reduction over the queries in chunks of 32.  Per chunk the score and dP tiles
// are recomputed as D[query][key] (A = 16 rows of q / dO, B = this tile's k / v rows, both straight 16-byte row loads),
// which leaves every lane with 8 queries of its key column in the permuted order {qc+4g..+3, qc+16+4g..+3}; the second
// products (dV^T = dO^T Pd, dK^T = Q^T dS) read their A operand from the transposed LDS copies in the same order.
__global__ __launch_bounds__(256) void mha_bwd_dkv_lean_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lk = p.Lk, Lq = p.Lq, nthr = blockDim.x, nw = nthr >> 6;
  const int QP = cdiv(Lq, 32) * 32, QS = QP + 8;
  u16* sQt = (u16*)sm;
  u16* sOt = sQt + 32 * QS;
  float4* sSt = (float4*)(sOt + 32 * QS);  // [QP] (max, 1/sum, delta, -)
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int bh = xcd_bh(blockIdx.x, gridDim.x, p.H, p.xcd_map), b = bh / p.H, h = bh - b * p.H;
  const u16* Q = (const u16*)p.q + (size_t)b * Lq * p.ldq + h * HD;
  const u16* DO = (const u16*)p.dout + (size_t)b * Lq * p.ldo + h * HD;
  stage_transposed(sQt, QS, Q, p.ldq, Lq, QP, t, nthr);
  stage_transposed(sOt, QS, DO, p.ldo, Lq, QP, t, nthr);
  for (int q = t; q < QP; q += nthr) sSt[q] = q < Lq ? *(const float4*)(p.stats + ((size_t)bh * Lq + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int kt = blockIdx.y * nw + wave;
  if (kt * 16 >= Lk) return;
  const int key = kt * 16 + li;
  const bool kv = key < Lk;
  const float kbias = (kv && !(p.kpm && p.kpm[b * Lk + key])) ? 0.f : -INFINITY;
  uint4 bk = make_uint4(0, 0, 0, 0), bv = make_uint4(0, 0, 0, 0);
  if (kv) {
    bk = ldg16((const u16*)p.k + (size_t)(b * Lk + key) * p.ldk + h * HD + 8 * g);
    bv = ldg16((const u16*)p.v + (size_t)(b * Lk + key) * p.ldv + h * HD + 8 * g);
  }
  const uint32_t seed = effective_seed(p.seed, p.seed_dev);
  f32x4 dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int qc = 0; qc < QP; qc += 32) {
    bf16x8 bp, bs;
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      const int qrow = qc + tq * 16 + li;  // A-operand row of this lane
      uint4 aq = make_uint4(0, 0, 0, 0), ado = make_uint4(0, 0, 0, 0);
      if (qrow < Lq) {
        aq = ldg16(Q + (size_t)qrow * p.ldq + 8 * g);
        ado = ldg16(DO + (size_t)qrow * p.ldo + 8 * g);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 sc = mfma_bf16(aq, bk, z);   // S[query 4g+r][key li]
      const f32x4 dp = mfma_bf16(ado, bv, z);  // dP
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qc + tq * 16 + 4 * g + r;
        float pr = 0.f, ds = 0.f;
        if (kv && qq < Lq) {
          const float4 st = sSt[qq];
          const float pv = __expf(sc[r] * p.scale + kbias - st.x) * st.y;
          const size_t pi = ((size_t)bh * Lq + qq) * Lk + key;
          const bool keep = !p.drop_thresh || dropout_keep(seed, (uint32_t)pi, p.drop_thresh);
          pr = keep ? pv * p.drop_scale : 0.f;
          const float d = keep ? dp[r] * p.drop_scale : 0.f;
          ds = pv * (d - st.z);
        }
        bp[tq * 4 + r] = (__bf16)pr;
        bs[tq * 4 + r] = (__bf16)ds;
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      uint4 ao, aq;
      const uint2 olo = *(const uint2*)&sOt[(m * 16 + li) * QS + qc + 4 * g], ohi = *(const uint2*)&sOt[(m * 16 + li) * QS + qc + 16 + 4 * g];
      const uint2 qlo = *(const uint2*)&sQt[(m * 16 + li) * QS + qc + 4 * g], qhi = *(const uint2*)&sQt[(m * 16 + li) * QS + qc + 16 + 4 * g];
      ao.x = olo.x; ao.y = olo.y; ao.z = ohi.x; ao.w = ohi.y;
      aq.x = qlo.x; aq.y = qlo.y; aq.z = qhi.x; aq.w = qhi.y;
      dv[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&ao, bp, dv[m], 0, 0, 0);
      dk[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&aq, bs, dk[m], 0, 0, 0);
    }
  }
  if (kv) {
    u16* DK = (u16*)p.dk + (size_t)(b * Lk + key) * p.ldk + h * HD + 4 * g;
    u16* DV = (u16*)p.dv + (size_t)(b * Lk + key) * p.ldv + h * HD + 4 * g;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x4 wk, wv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wk[r] = (__bf16)(dk[m][r] * p.scale);
        wv[r] = (__bf16)dv[m][r];
      }
      *(bf16x4*)(DK + m * 16) = wk;
      *(bf16x4*)(DV + m * 16) = wv;
    }
  }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool mfma_path_ok(const MhaParams& p, int dtype, bool bwd, int hd) {
  static const bool off = [] { const char* e = getenv("TD_MHA_VALU"); return e && e[0] == '1'; }();
  if (off || hd != HD || dtype != TD_BF16 || p.Lk > 256 || p.Lq > 448) return false;
  if ((p.ldq | p.ldk | p.ldv | p.ldo) & 7) return false;
  if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.v)) return false;
  if (!bwd && ((uintptr_t)p.out & 7)) return false;
  if (bwd && (!aligned16(p.dout) || ((uintptr_t)p.dq & 7) || ((uintptr_t)p.dk & 7) || ((uintptr_t)p.dv & 7))) return false;
  return true;
}
// wavefronts per workgroup: 4 when that still leaves >= 256 workgroups, else fewer (small problems want more, smaller
// workgroups: the temporal self-attention is 8 heads x 100 queries in total)
static int pick_waves(int BH, int tiles) {
  int nw = 4;
  while (nw > 1 && (BH * cdiv(tiles, nw) < 256 || nw > tiles)) nw >>= 1;
  return nw;
}
static int fill(MhaParams& p, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
                float dropout_p, uint32_t seed, const uint32_t* counter, const char* who) {
  TD_REQUIRE(hd == 32 || hd == 64, "%s: head dim %d unsupported (32: TubeDETR's transformer, 64: RoBERTa)", who, hd);
  TD_REQUIRE(Lk >= 1 && Lk <= 64 * KJ, "%s: Lk=%d out of range (1..%d)", who, Lk, 64 * KJ);
  TD_REQUIRE(B >= 1 && H >= 1 && Lq >= 1, "%s: bad sizes", who);
  TD_REQUIRE((double)B * H * Lq * Lk < 4294967295.0, "%s: probs tensor too large for the dropout index", who);
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale = scale;
  {
    static const int xm = [] { const char* e_ = getenv("TD_MHA_XCD_MAP"); return e_ ? atoi(e_) : 1; }();
    p.xcd_map = xm;
  }
  p.drop_thresh = 0; p.drop_scale = 1.f; p.seed = seed; p.seed_dev = nullptr;
  if (dropout_p > 0.f) {
    TD_REQUIRE(dropout_p < 1.f, "%s: dropout_p must be < 1", who);
    p.drop_thresh = (uint32_t)((double)dropout_p * 4294967296.0);
    if (!p.drop_thresh) p.drop_thresh = 1;
    p.drop_scale = 1.f / (1.f - dropout_p);
    p.seed_dev = counter;
  }
  return TD_OK;
}

// dynamic LDS above 64 KiB needs the attribute once per kernel (done once: not a stream operation, kept out of any
// graph capture that may be recording the launches)
static void mha_allow_big_lds() {
  static const bool done = [] {
    const int big = 160 * 1024;
#define TD_BIG(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, big)
    TD_BIG((mha_fwd_kernel<u16, 32>)); TD_BIG((mha_fwd_kernel<float, 32>)); TD_BIG((mha_fwd_kernel<u16, 64>)); TD_BIG((mha_fwd_kernel<float, 64>));
    TD_BIG((mha_bwd_dq_kernel<u16, 32>)); TD_BIG((mha_bwd_dq_kernel<float, 32>)); TD_BIG((mha_bwd_dq_kernel<u16, 64>)); TD_BIG((mha_bwd_dq_kernel<float, 64>));
    TD_BIG((mha_bwd_dkv_kernel<u16, 32>)); TD_BIG((mha_bwd_dkv_kernel<float, 32>)); TD_BIG((mha_bwd_dkv_kernel<u16, 64>)); TD_BIG((mha_bwd_dkv_kernel<float, 64>));
#undef TD_BIG
    return true;
  }();
  (void)done;
}

}  // namespace td
using namespace td;

extern "C" int td_mha_fwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, void* out, float* probs,
                          float* wavg, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo,
                          float scale, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype,
                          td_stream_t stream) {
  TD_REQUIRE(q && k && v && out && probs, "td_mha_fwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, dropout_counter, "td_mha_fwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.kpm = key_pad; p.out = out; p.probs = probs;
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)(2 * Lk * (hd + 1) + 4 * Lk) * sizeof(float);
  TD_REQUIRE(lds <= 160 * 1024, "td_mha_fwd: Lk too large for LDS");
  dim3 grid(B * H, (Lq + QT - 1) / QT);
  mha_allow_big_lds();
  if (mfma_path_ok(p, dtype, false, hd)) {
    const int qt = cdiv(Lq, 16), nw = pick_waves(B * H, qt);
    dim3 g2(B * H, cdiv(qt, nw));
    if (Lk <= 64) mha_fwd_mfma_kernel<4><<<g2, 64 * nw, 0, st>>>(p);
    else if (Lk <= 128) mha_fwd_mfma_kernel<8><<<g2, 64 * nw, 0, st>>>(p);
    else if (Lk <= 160) mha_fwd_mfma_kernel<10><<<g2, 64 * nw, 0, st>>>(p);
    else mha_fwd_mfma_kernel<16><<<g2, 64 * nw, 0, st>>>(p);
  } else if (dtype == TD_BF16 && hd == 32) mha_fwd_kernel<u16, 32><<<grid, 256, lds, st>>>(p);
  else if (dtype == TD_BF16) mha_fwd_kernel<u16, 64><<<grid, 256, lds, st>>>(p);
  else if (dtype == TD_F32 && hd == 32) mha_fwd_kernel<float, 32><<<grid, 256, lds, st>>>(p);
  else if (dtype == TD_F32) mha_fwd_kernel<float, 64><<<grid, 256, lds, st>>>(p);
  else TD_REQUIRE(false, "td_mha_fwd: bad dtype");
  rc = check_launch("td_mha_fwd");
  if (rc) return rc;
  if (wavg) {
    size_t n = (size_t)B * Lq * Lk;
    avg_heads_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(probs, wavg, B, H, Lq, Lk, p.drop_thresh, p.drop_scale, p.seed, p.seed_dev);
    rc = check_launch("td_mha_fwd(avg)");
  }
  return rc;
}

extern "C" int td_mha_bwd(const void* q, const void* k, const void* v, const void* dout, const float* probs,
                          const float* dwavg, void* dq, void* dk, void* dv, float* ds_ws, int B, int H, int Lq, int Lk,
                          int hd, int ldq, int ldk, int ldv, int ldo, float scale, float dropout_p,
                          uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream) {
  TD_REQUIRE(q && k && v && dout && probs && dq && dk && dv && ds_ws, "td_mha_bwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, dropout_counter, "td_mha_bwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.dout = dout; p.probs = (float*)probs; p.dwavg = dwavg;
  p.dq = dq; p.dk = dk; p.dv = dv; p.ds_ws = ds_ws;
  hipStream_t st = (hipStream_t)stream;
  size_t ldsA = (size_t)(2 * Lk * (hd + 1) + 4 * Lk) * sizeof(float);
  size_t ldsB = (size_t)(2 * Lq * hd) * sizeof(float);
  TD_REQUIRE(ldsA <= 160 * 1024 && ldsB <= 160 * 1024, "td_mha_bwd: Lq/Lk too large for LDS");
  dim3 gridA(B * H, (Lq + QT - 1) / QT), gridB(B * H, (Lk + 63) / 64);
  mha_allow_big_lds();
  if (mfma_path_ok(p, dtype, true, hd)) {
    const int qt = cdiv(Lq, 16), nwq = pick_waves(B * H, qt);
    dim3 gA(B * H, cdiv(qt, nwq));
    if (Lk <= 64) mha_bwd_dq_mfma_kernel<4><<<gA, 64 * nwq, 0, st>>>(p);
    else if (Lk <= 128) mha_bwd_dq_mfma_kernel<8><<<gA, 64 * nwq, 0, st>>>(p);
    else if (Lk <= 160) mha_bwd_dq_mfma_kernel<10><<<gA, 64 * nwq, 0, st>>>(p);
    else mha_bwd_dq_mfma_kernel<16><<<gA, 64 * nwq, 0, st>>>(p);
    const int ktl = cdiv(Lk, 16), nwk = pick_waves(B * H, ktl);
    const int QS = cdiv(Lq, 32) * 32 + 8;
    mha_bwd_dkv_mfma_kernel<<<dim3(B * H, cdiv(ktl, nwk)), 64 * nwk, (size_t)2 * 32 * QS * sizeof(u16), st>>>(p);
  } else if (dtype == TD_BF16 && hd == 32) {
    mha_bwd_dq_kernel<u16, 32><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<u16, 32><<<gridB, 256, ldsB, st>>>(p);
  } else if (dtype == TD_BF16) {
    mha_bwd_dq_kernel<u16, 64><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<u16, 64><<<gridB, 256, ldsB, st>>>(p);
  } else if (dtype == TD_F32 && hd == 32) {
    mha_bwd_dq_kernel<float, 32><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<float, 32><<<gridB, 256, ldsB, st>>>(p);
  } else if (dtype == TD_F32) {
    mha_bwd_dq_kernel<float, 64><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<float, 64><<<gridB, 256, ldsB, st>>>(p);
  } else TD_REQUIRE(false, "td_mha_bwd: bad dtype");
  return check_launch("td_mha_bwd");
}

// ---- lean entry points: bf16, head dim 32, Lk <= 256; no head-averaged weights (see include/tubedetr_hip.h) ----
static int lean_check(const MhaParams& p, int dtype, int hd, const char* who) {
  TD_REQUIRE(dtype == TD_BF16 && hd == HD && p.Lk <= 256 && p.Lq <= 448, "%s: the lean path takes bf16, head dim 32, Lk <= 256, Lq <= 448", who);
  TD_REQUIRE(((p.ldq | p.ldk | p.ldv | p.ldo) & 7) == 0 && aligned16(p.q) && aligned16(p.k) && aligned16(p.v) && aligned16(p.out) && aligned16(p.stats),
             "%s: rows must be 16-byte aligned", who);
  return TD_OK;
}

extern "C" size_t td_mha_lean_stats_bytes(int B, int H, int Lq) { return (size_t)B * H * Lq * 4 * sizeof(float); }

extern "C" int td_mha_lean_fwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, void* out, float* stats, int B,
                               int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale, float dropout_p,
                               uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream) {
  TD_REQUIRE(q && k && v && out && stats, "td_mha_lean_fwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  TD_REQUIRE(hd == HD, "td_mha_lean_fwd: the lean path takes head dim 32 (got %d)", hd);
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, dropout_counter, "td_mha_lean_fwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.kpm = key_pad; p.out = out; p.stats = stats;
  rc = lean_check(p, dtype, hd, "td_mha_lean_fwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int qt = cdiv(Lq, 16), nw = pick_waves(B * H, qt);
  dim3 g2(B * H, cdiv(qt, nw));
  if (Lk <= 64) mha_fwd_mfma_kernel<4, true><<<g2, 64 * nw, 0, st>>>(p);
  else if (Lk <= 128) mha_fwd_mfma_kernel<8, true><<<g2, 64 * nw, 0, st>>>(p);
  else if (Lk <= 160) mha_fwd_mfma_kernel<10, true><<<g2, 64 * nw, 0, st>>>(p);
  else mha_fwd_mfma_kernel<16, true><<<g2, 64 * nw, 0, st>>>(p);
  return check_launch("td_mha_lean_fwd");
}

extern "C" int td_mha_lean_bwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, const void* out, const void* dout,
                               float* stats, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk,
                               int ldv, int ldo, float scale, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter,
                               int dtype, td_stream_t stream) {
  TD_REQUIRE(q && k && v && out && dout && stats && dq && dk && dv, "td_mha_lean_bwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  TD_REQUIRE(hd == HD, "td_mha_lean_bwd: the lean path takes head dim 32 (got %d)", hd);
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, dropout_counter, "td_mha_lean_bwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.kpm = key_pad; p.out = (void*)out; p.dout = dout; p.stats = stats; p.dq = dq; p.dk = dk; p.dv = dv;
  rc = lean_check(p, dtype, hd, "td_mha_lean_bwd");
  if (rc) return rc;
  TD_REQUIRE(aligned16(dout), "td_mha_lean_bwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int qt = cdiv(Lq, 16), nwq = pick_waves(B * H, qt);
  dim3 gA(B * H, cdiv(qt, nwq));
  if (Lk <= 64) mha_bwd_dq_lean_kernel<4><<<gA, 64 * nwq, 0, st>>>(p);
  else if (Lk <= 128) mha_bwd_dq_lean_kernel<8><<<gA, 64 * nwq, 0, st>>>(p);
  else if (Lk <= 160) mha_bwd_dq_lean_kernel<10><<<gA, 64 * nwq, 0, st>>>(p);
  else mha_bwd_dq_lean_kernel<16><<<gA, 64 * nwq, 0, st>>>(p);
  const int ktl = cdiv(Lk, 16), nwk = pick_waves(B * H, ktl);
  const int QP = cdiv(Lq, 32) * 32, QS = QP + 8;
  mha_bwd_dkv_lean_kernel<<<dim3(B * H, cdiv(ktl, nwk)), 64 * nwk, (size_t)2 * 32 * QS * sizeof(u16) + (size_t)QP * sizeof(float4), st>>>(p);
  return check_launch("td_mha_lean_bwd");
}
