// Multi-head attention core (softmax(scale*QK^T + key padding) V) forward / backward for the small
// attention problems of TubeDETR: per-frame visual-text self-attention (S = hw+L = 151 keys, batch =
// slow clips), temporal self-attention (T = 100), time-aligned cross-attention (1 query per frame vs
// its 151 keys).  Head dim 32.  K/V of one (batch, head) live in LDS; one wavefront owns one query
// row: lanes span keys, softmax statistics by wavefront shuffles, PV with lanes re-mapped to
// (channel, key-half).  fp32 math on the VALU in both dtypes (inputs/outputs are T).
// Reference: torch nn.MultiheadAttention at models/transformer.py:613,638-640,661-662,713-740.
#include "td_common.h"

namespace td {

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
  int B, H, Lq, Lk, ldq, ldk, ldv, ldo;
  float scale;
  uint32_t drop_thresh;
  float drop_scale;
  uint32_t seed;
  const uint32_t* seed_dev;
};

#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename T>
__global__ __launch_bounds__(256) void mha_fwd_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lk = p.Lk, Lq = p.Lq;
  float* sK = sm;
  float* sV = sK + Lk * 33;
  float* sP = sV + Lk * 33;  // [4][Lk]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lk * HD; idx += 256) {
    int kk = idx >> 5, d = idx & 31;
    sK[kk * 33 + d] = Elem<T>::load(p.k, (size_t)(b * Lk + kk) * p.ldk + h * HD + d);
    sV[kk * 33 + d] = Elem<T>::load(p.v, (size_t)(b * Lk + kk) * p.ldv + h * HD + d);
  }
  __syncthreads();
  const int q0 = blockIdx.y * QT;
  const int q1 = min(Lq, q0 + QT);
  float* myP = sP + wave * Lk;
  for (int qi = q0 + wave; qi < q1; qi += 4) {
    float qv[HD];
    const size_t qoff = (size_t)(b * Lq + qi) * p.ldq + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = Elem<T>::load(p.q, qoff + d);
    float s[KJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      int kk = lane + 64 * j;
      s[j] = -INFINITY;
      if (kk < Lk) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) dot += qv[d] * sK[kk * 33 + d];
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
    const int d = lane & 31, half = lane >> 5;
    float acc = 0.f;
    for (int kk = half; kk < Lk; kk += 2) acc += myP[kk] * sV[kk * 33 + d];
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) Elem<T>::store(p.out, (size_t)(b * Lq + qi) * p.ldo + h * HD + d, acc);
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
template <typename T>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lk = p.Lk, Lq = p.Lq;
  float* sK = sm;
  float* sV = sK + Lk * 33;
  float* sP = sV + Lk * 33;  // [4][Lk]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lk * HD; idx += 256) {
    int kk = idx >> 5, d = idx & 31;
    sK[kk * 33 + d] = Elem<T>::load(p.k, (size_t)(b * Lk + kk) * p.ldk + h * HD + d);
    sV[kk * 33 + d] = Elem<T>::load(p.v, (size_t)(b * Lk + kk) * p.ldv + h * HD + d);
  }
  __syncthreads();
  float* myP = sP + wave * Lk;
  const float invH = 1.f / p.H;
  const int q0 = blockIdx.y * QT;
  const int q1 = min(Lq, q0 + QT);
  for (int qi = q0 + wave; qi < q1; qi += 4) {
    float dov[HD];
    const size_t ooff = (size_t)(b * Lq + qi) * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) dov[d] = Elem<T>::load(p.dout, ooff + d);
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
        for (int d = 0; d < HD; ++d) dot += dov[d] * sV[kk * 33 + d];
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
    const int d = lane & 31, half = lane >> 5;
    float acc = 0.f;
    for (int kk = half; kk < Lk; kk += 2) acc += myP[kk] * sK[kk * 33 + d];
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) Elem<T>::store(p.dq, (size_t)(b * Lq + qi) * p.ldq + h * HD + d, acc * p.scale);
    LDS_FENCE();
  }
}

// Backward, kernel B: grid (batch*head, tiles of 64 keys).  Thread = (key, 8-channel slice); loops over all queries:
// dK = scale * dS^T Q, dV = dropout(P)^T dO.  Q / dO rows are wave-uniform LDS broadcasts, dS / P columns are
// coalesced over keys.
template <typename T>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lk = p.Lk, Lq = p.Lq;
  float* sQ = sm;             // [Lq][32]
  float* sO = sm + Lq * HD;   // [Lq][32]
  const int t = threadIdx.x;
  const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
  for (int idx = t; idx < Lq * HD; idx += 256) {
    int qq = idx >> 5, d = idx & 31;
    sQ[idx] = Elem<T>::load(p.q, (size_t)(b * Lq + qq) * p.ldq + h * HD + d);
    sO[idx] = Elem<T>::load(p.dout, (size_t)(b * Lq + qq) * p.ldo + h * HD + d);
  }
  __syncthreads();
  const int kk = blockIdx.y * 64 + (t & 63);
  const int part = t >> 6;  // wave-uniform: channels part*8 .. part*8+7
  if (kk >= Lk) return;
  float dK[8], dV[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dK[d] = dV[d] = 0.f;
  for (int qq = 0; qq < Lq; ++qq) {
    const size_t pi = ((size_t)bh * Lq + qq) * Lk + kk;
    float ds = p.ds_ws[pi];
    float pr = p.probs[pi];
    if (p.drop_thresh) pr = dropout_keep(effective_seed(p.seed, p.seed_dev), (uint32_t)pi, p.drop_thresh) ? pr * p.drop_scale : 0.f;
    const float4* q4 = (const float4*)(sQ + qq * HD + part * 8);
    const float4* o4 = (const float4*)(sO + qq * HD + part * 8);
    float4 a0 = q4[0], a1 = q4[1], o0 = o4[0], o1 = o4[1];
    dK[0] += ds * a0.x; dK[1] += ds * a0.y; dK[2] += ds * a0.z; dK[3] += ds * a0.w;
    dK[4] += ds * a1.x; dK[5] += ds * a1.y; dK[6] += ds * a1.z; dK[7] += ds * a1.w;
    dV[0] += pr * o0.x; dV[1] += pr * o0.y; dV[2] += pr * o0.z; dV[3] += pr * o0.w;
    dV[4] += pr * o1.x; dV[5] += pr * o1.y; dV[6] += pr * o1.z; dV[7] += pr * o1.w;
  }
  const size_t ko = (size_t)(b * Lk + kk) * p.ldk + h * HD + part * 8, vo = (size_t)(b * Lk + kk) * p.ldv + h * HD + part * 8;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    Elem<T>::store(p.dk, ko + d, dK[d] * p.scale);
    Elem<T>::store(p.dv, vo + d, dV[d]);
  }
}

static int fill(MhaParams& p, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
                float dropout_p, uint32_t seed, const char* who) {
  TD_REQUIRE(hd == HD, "%s: head dim %d unsupported (only 32)", who, hd);
  TD_REQUIRE(Lk >= 1 && Lk <= 64 * KJ, "%s: Lk=%d out of range (1..%d)", who, Lk, 64 * KJ);
  TD_REQUIRE(B >= 1 && H >= 1 && Lq >= 1, "%s: bad sizes", who);
  TD_REQUIRE((double)B * H * Lq * Lk < 4294967295.0, "%s: probs tensor too large for the dropout index", who);
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.scale = scale;
  p.drop_thresh = 0; p.drop_scale = 1.f; p.seed = seed; p.seed_dev = nullptr;
  if (dropout_p > 0.f) {
    TD_REQUIRE(dropout_p < 1.f, "%s: dropout_p must be < 1", who);
    p.drop_thresh = (uint32_t)((double)dropout_p * 4294967296.0);
    if (!p.drop_thresh) p.drop_thresh = 1;
    p.drop_scale = 1.f / (1.f - dropout_p);
    p.seed_dev = dropout_counter();
  }
  return TD_OK;
}

// dynamic LDS above 64 KiB needs the attribute once per kernel (done once: not a stream operation, kept out of any
// graph capture that may be recording the launches)
static void mha_allow_big_lds() {
  static const bool done = [] {
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)mha_fwd_kernel<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)mha_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)mha_bwd_dq_kernel<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)mha_bwd_dq_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)mha_bwd_dkv_kernel<u16>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)mha_bwd_dkv_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    return true;
  }();
  (void)done;
}

}  // namespace td
using namespace td;

extern "C" int td_mha_fwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, void* out, float* probs,
                          float* wavg, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo,
                          float scale, float dropout_p, uint32_t dropout_seed, int dtype, td_stream_t stream) {
  TD_REQUIRE(q && k && v && out && probs, "td_mha_fwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, "td_mha_fwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.kpm = key_pad; p.out = out; p.probs = probs;
  hipStream_t st = (hipStream_t)stream;
  size_t lds = (size_t)(2 * Lk * 33 + 4 * Lk) * sizeof(float);
  dim3 grid(B * H, (Lq + QT - 1) / QT);
  mha_allow_big_lds();
  if (dtype == TD_BF16) mha_fwd_kernel<u16><<<grid, 256, lds, st>>>(p);
  else if (dtype == TD_F32) mha_fwd_kernel<float><<<grid, 256, lds, st>>>(p);
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
                          uint32_t dropout_seed, int dtype, td_stream_t stream) {
  TD_REQUIRE(q && k && v && dout && probs && dq && dk && dv && ds_ws, "td_mha_bwd: null pointer");
  MhaParams p;
  memset(&p, 0, sizeof(p));
  int rc = fill(p, B, H, Lq, Lk, hd, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, "td_mha_bwd");
  if (rc) return rc;
  p.q = q; p.k = k; p.v = v; p.dout = dout; p.probs = (float*)probs; p.dwavg = dwavg;
  p.dq = dq; p.dk = dk; p.dv = dv; p.ds_ws = ds_ws;
  hipStream_t st = (hipStream_t)stream;
  size_t ldsA = (size_t)(2 * Lk * 33 + 4 * Lk) * sizeof(float);
  size_t ldsB = (size_t)(2 * Lq * HD) * sizeof(float);
  TD_REQUIRE(ldsA <= 160 * 1024 && ldsB <= 160 * 1024, "td_mha_bwd: Lq/Lk too large for LDS");
  dim3 gridA(B * H, (Lq + QT - 1) / QT), gridB(B * H, (Lk + 63) / 64);
  mha_allow_big_lds();
  if (dtype == TD_BF16) {
    mha_bwd_dq_kernel<u16><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<u16><<<gridB, 256, ldsB, st>>>(p);
  } else if (dtype == TD_F32) {
    mha_bwd_dq_kernel<float><<<gridA, 256, ldsA, st>>>(p);
    mha_bwd_dkv_kernel<float><<<gridB, 256, ldsB, st>>>(p);
  } else TD_REQUIRE(false, "td_mha_bwd: bad dtype");
  return check_launch("td_mha_bwd");
}
