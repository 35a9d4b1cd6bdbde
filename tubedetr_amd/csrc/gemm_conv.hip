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
#include "td_common.h"

namespace td {

struct GemmParams {
  const char* src;
  const char* w;
  char* out;
  td_conv_desc d;
  int M, K;
  const float* bias;
  const char* residual;
  const char* mask_src;
  int relu, sigmoid;
  float alpha;
  uint32_t drop_thresh;
  float drop_scale;
  uint32_t seed;
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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmParams p) {
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;   // elements per 16-byte chunk
  constexpr int BK = 128 / ES;   // K elements per tile (128 bytes per row)
  constexpr int AI = BM / 32, BI = BN / 32;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 16, TN = WN / 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];
  auto sA = [&](int buf) -> char* { return smem + buf * ((BM + BN) * 128); };
  auto sB = [&](int buf) -> char* { return smem + buf * ((BM + BN) * 128) + BM * 128; };

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int chunk = t & 7, rowt = t >> 3;
  const int HoWo = d.Ho * d.Wo;

  // per-thread row bookkeeping (fixed for the whole K loop)
  int a_img[AI], a_hb[AI], a_wb[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + rowt + 32 * i;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    int n = mm / HoWo;
    int rem = mm - n * HoWo;
    int ho = rem / d.Wo, wo = rem - ho * d.Wo;
    a_img[i] = n * d.Hs * d.Ws;
    if (d.mode == 0) {
      a_hb[i] = ho * d.stride - d.pad;
      a_wb[i] = wo * d.stride - d.pad;
    } else {
      a_hb[i] = ho + d.pad;
      a_wb[i] = wo + d.pad;
    }
  }
  const int RS = d.R * d.S;
  uint4 ra[AI], rb[BI];

  auto load_tile = [&](int kt) {
    int kk = kt * BK + chunk * VEC;
    bool kvalid = kk < p.K;
    int r = 0, s = 0, c = kk;
    if (RS > 1) {
      int tap = kk / d.C;
      c = kk - tap * d.C;
      r = tap / d.S;
      s = tap - r * d.S;
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int hs, ws;
      bool ok = a_ok[i] && kvalid;
      if (d.mode == 0) {
        hs = a_hb[i] + r;
        ws = a_wb[i] + s;
      } else {
        int th = a_hb[i] - r, tw = a_wb[i] - s;
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
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        size_t off = ((size_t)(a_img[i] + hs * d.Ws + ws) * d.C + c) * ES;
        v = *(const uint4*)(p.src + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + rowt + 32 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kvalid && n < d.Nc) v = *(const uint4*)(p.w + ((size_t)n * p.K + kk) * ES);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int row = rowt + 32 * i;
      *(uint4*)(sA(buf) + row * 128 + ((chunk ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int row = rowt + 32 * i;
      *(uint4*)(sB(buf) + row * 128 + ((chunk ^ (row & 7)) << 4)) = rb[i];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wy = wave >> 1, wx = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cidx = ks * 4 + lg;
      uint4 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        int row = wy * WM + j * 16 + lr;
        af[j] = *(const uint4*)(sA(buf) + row * 128 + ((cidx ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        int row = wx * WN + i * 16 + lr;
        wf[i] = *(const uint4*)(sB(buf) + row * 128 + ((cidx ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mfma<T>::run(wf[i], af[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds out[m][nb..nb+3] per (i,j) tile ----
  const float alpha = p.alpha;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    int m = m0 + wy * WM + j * 16 + lr;
    if (m >= p.M) continue;
    size_t orow = m;
    if (d.out_sp > 1) {
      int n = m / HoWo;
      int rem = m - n * HoWo;
      int ho = rem / d.Wo, wo = rem - ho * d.Wo;
      orow = ((size_t)n * d.out_H + (size_t)ho * d.out_sp) * d.out_W + (size_t)wo * d.out_sp;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      int nb = n0 + wx * WN + i * 16 + 4 * lg;
      if (nb >= d.Nc) continue;
      size_t off = orow * d.ldc + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha;
      const bool full = (nb + 3 < d.Nc) && ((d.ldc & 3) == 0);
      const int cnt = full ? 4 : min(4, d.Nc - nb);
      if (p.bias)
        for (int r = 0; r < cnt; ++r) v[r] += p.bias[nb + r];
      if (p.residual)
        for (int r = 0; r < cnt; ++r) v[r] += Elem<T>::load(p.residual, off + r);
      if (p.relu)
        for (int r = 0; r < cnt; ++r) v[r] = fmaxf(v[r], 0.f);
      if (p.sigmoid)
        for (int r = 0; r < cnt; ++r) v[r] = sigmoidf_(v[r]);
      if (p.mask_src)
        for (int r = 0; r < cnt; ++r) v[r] = Elem<T>::load(p.mask_src, off + r) > 0.f ? v[r] : 0.f;
      if (p.drop_thresh)
        for (int r = 0; r < cnt; ++r) v[r] = dropout_keep(p.seed, (uint32_t)(off + r), p.drop_thresh) ? v[r] * p.drop_scale : 0.f;
      if (full) {
        if (ES == 2) {
          uint2 o;
          o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
          o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
          *(uint2*)(p.out + off * 2) = o;
        } else {
          *(float4*)(p.out + off * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        for (int r = 0; r < cnt; ++r) Elem<T>::store(p.out, off + r, v[r]);
      }
    }
  }
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
  td_conv_desc d;
  int M, K, ldg, mper;
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;
  constexpr int MK = 64 / ES;              // reduction rows per stage: 32 (bf16) / 16 (fp32)
  constexpr int CPR = 128 * ES / 16;       // 16-byte chunks per 128-channel row: 16 / 32
  constexpr int RSB = 128 * ES + 16 * ES;  // padded row stride in bytes: 288 / 576
  constexpr int TILEB = MK * RSB;          // 9216 both
  constexpr int LI = MK * CPR / 256;       // chunks per thread per operand = 2
  __shared__ __attribute__((aligned(16))) char smem[4 * TILEB];
  auto sG = [&](int buf) -> char* { return smem + buf * (2 * TILEB); };
  auto sX = [&](int buf) -> char* { return smem + buf * (2 * TILEB) + TILEB; };

  const td_conv_desc& d = p.d;
  const int t = threadIdx.x;
  const int co0 = blockIdx.x * 128, kk0 = blockIdx.y * 128;
  const int mbeg = blockIdx.z * p.mper;
  const int mend = min(p.M, mbeg + p.mper);
  if (mbeg >= mend) return;
  const int HoWo = d.Ho * d.Wo;
  const int ch = t % CPR;       // constant chunk column of this thread
  const int row0 = t / CPR;     // first row; second row = row0 + 256/CPR
  constexpr int RSTEP = 256 / CPR;

  // gather column info (fixed per thread)
  const int kk = kk0 + ch * VEC;
  const bool kvalid = kk < p.K;
  int r = 0, s = 0, c = kk;
  if (d.R * d.S > 1) {
    int tap = kk / d.C;
    c = kk - tap * d.C;
    r = tap / d.S;
    s = tap - r * d.S;
  }
  const int co = co0 + ch * VEC;
  const bool covalid = co < d.Nc;

  uint4 rg[LI], rx[LI];
  auto load_tile = [&](int mb) {
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      int m = mb + row0 + RSTEP * i;
      bool ok = m < mend;
      uint4 vg = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
      if (ok && covalid) vg = *(const uint4*)(p.g + ((size_t)m * p.ldg + co) * ES);
      if (ok && kvalid) {
        int n = m / HoWo;
        int rem = m - n * HoWo;
        int ho = rem / d.Wo, wo = rem - ho * d.Wo;
        int hs = ho * d.stride - d.pad + r, ws = wo * d.stride - d.pad + s;
        if ((unsigned)hs < (unsigned)d.Hs && (unsigned)ws < (unsigned)d.Ws)
          vx = *(const uint4*)(p.src + ((size_t)((n * d.Hs + hs) * d.Ws + ws) * d.C + c) * ES);
      }
      rg[i] = vg;
      rx[i] = vx;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      int row = row0 + RSTEP * i;
      *(uint4*)(sG(buf) + row * RSB + ch * 16) = rg[i];
      *(uint4*)(sX(buf) + row * RSB + ch * 16) = rx[i];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wy = wave >> 1, wx = wave & 1;  // wy: co direction, wx: kk direction
  const int lr = lane & 15, lg = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nit = (mend - mbeg + MK - 1) / MK;
  load_tile(mbeg);
  store_tile(0);
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) load_tile(mbeg + (it + 1) * MK);
    if constexpr (ES == 2) {
      // lane p = 4*j+q of each 16-lane group addresses row (8*lg + 4*h + j), columns 4q..4q+3 of the 16-wide
      // channel tile; the transposing read hands lane lr the 4 rows {8lg+4h+0..3} of column lr.
      const int jrow = lr >> 2, q = lr & 3;
      uint4 gf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* base = sG(buf) + (wy * 64 + i * 16 + 4 * q) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + (8 * lg + jrow) * RSB));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + (8 * lg + 4 + jrow) * RSB));
        uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
        gf[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* base = sX(buf) + (wx * 64 + j * 16 + 4 * q) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + (8 * lg + jrow) * RSB));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + (8 * lg + 4 + jrow) * RSB));
        uint2 l2 = *(uint2*)&lo, h2 = *(uint2*)&hi;
        xf[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&gf[i], *(const bf16x8*)&xf[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int s4 = 0; s4 < MK / 4; ++s4) {
        float gf[4], xf[4];
        const int krow = lg + 4 * s4;
#pragma unroll
        for (int i = 0; i < 4; ++i) gf[i] = *(const float*)(sG(buf) + krow * RSB + (wy * 64 + i * 16 + lr) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *(const float*)(sX(buf) + krow * RSB + (wx * 64 + j * 16 + lr) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (it + 1 < nit) store_tile(buf ^ 1);
    __syncthreads();
  }
  // D[i=co][j=kk]: lane holds co = base + 4*lg + r, kk = base + lr
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
}

static int validate(const td_conv_desc* d, int dtype, const char* who) {
  const int vec = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "%s: bad dtype %d", who, dtype);
  TD_REQUIRE(d->C % vec == 0, "%s: source channels C=%d must be a multiple of %d (pad them)", who, d->C, vec);
  TD_REQUIRE(d->N > 0 && d->Hs > 0 && d->Ws > 0 && d->Ho > 0 && d->Wo > 0 && d->R > 0 && d->S > 0 && d->stride > 0,
             "%s: bad geometry", who);
  TD_REQUIRE((double)d->N * d->Hs * d->Ws * d->C < 2147483647.0, "%s: source tensor exceeds 2^31 elements", who);
  return TD_OK;
}

}  // namespace td

using namespace td;

extern "C" int td_conv_gemm(const void* src, const void* wmat, void* out, const td_conv_desc* d, const td_epilogue* e,
                            int dtype, td_stream_t stream) {
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
  p.alpha = 1.f;
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
    }
  }
  TD_REQUIRE(d->ldc >= d->Nc, "td_conv_gemm: ldc < Nc");
  hipStream_t st = (hipStream_t)stream;
  const bool narrow = d->Nc <= 64;
  dim3 grid(cdiv(p.M, 128), cdiv(d->Nc, narrow ? 64 : 128));
  const bool prof = prof_on();
  if (prof) prof_begin(narrow ? TD_PROF_GEMM_128x64 : TD_PROF_GEMM_128x128, dtype, 2.0 * p.M * d->Nc * p.K, st);
  if (dtype == TD_BF16) {
    if (narrow) conv_gemm_kernel<u16, 128, 64><<<grid, 256, 0, st>>>(p);
    else conv_gemm_kernel<u16, 128, 128><<<grid, 256, 0, st>>>(p);
  } else {
    if (narrow) conv_gemm_kernel<float, 128, 64><<<grid, 256, 0, st>>>(p);
    else conv_gemm_kernel<float, 128, 128><<<grid, 256, 0, st>>>(p);
  }
  if (prof) prof_end(st);
  return check_launch("td_conv_gemm");
}

extern "C" int td_conv_wgrad(const void* g, const void* src, float* dw, const td_conv_desc* d, int ldg, int dtype,
                             int splits, td_stream_t stream) {
  TD_REQUIRE(g && src && dw && d, "td_conv_wgrad: null pointer");
  int rc = validate(d, dtype, "td_conv_wgrad");
  if (rc) return rc;
  const int vec = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(d->mode == 0, "td_conv_wgrad: forward geometry expected");
  TD_REQUIRE(d->Nc % vec == 0 && ldg % vec == 0, "td_conv_wgrad: Nc=%d / ldg=%d must be multiples of %d", d->Nc, ldg, vec);
  WgradParams p;
  p.g = (const char*)g;
  p.src = (const char*)src;
  p.dw = dw;
  p.d = *d;
  p.M = d->N * d->Ho * d->Wo;
  p.K = d->R * d->S * d->C;
  p.ldg = ldg;
  const int mk = dtype == TD_BF16 ? 32 : 16;
  if (splits < 1) {
    // aim for ~1024 workgroups (4 per CU) but keep >= 8 reduction stages per split
    int tiles = cdiv(d->Nc, 128) * cdiv(p.K, 128);
    splits = cdiv(1024, tiles);
    int maxs = cdiv(p.M, 8 * mk);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  p.mper = cdiv(cdiv(p.M, splits), mk) * mk;
  splits = cdiv(p.M, p.mper);
  dim3 grid(cdiv(d->Nc, 128), cdiv(p.K, 128), splits);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) prof_begin(TD_PROF_WGRAD, dtype, 2.0 * p.M * d->Nc * p.K, st);
  if (dtype == TD_BF16) conv_wgrad_kernel<u16><<<grid, 256, 0, st>>>(p);
  else conv_wgrad_kernel<float><<<grid, 256, 0, st>>>(p);
  if (prof) prof_end(st);
  return check_launch("td_conv_wgrad");
}
