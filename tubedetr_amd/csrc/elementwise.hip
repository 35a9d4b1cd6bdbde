// HBM-bound elementwise / reduction kernels of the TubeDETR hot path (gfx950): stem max-pool, fused
// residual-add + LayerNorm forward/backward (one wave per row, wavefront shuffle reductions), column
// sums for bias gradients, elementwise add, sine position encoding from the pad mask.
// Reference call sites: torchvision resnet stem maxpool (models/backbone.py:98), nn.LayerNorm at
// models/transformer.py:619-620,641-645,669-672,721-750,89,581,765-771, PositionEmbeddingSine
// (models/position_encoding.py:71-94).
#include "td_common.h"

namespace td {

template <typename T>
__global__ void maxpool3x3s2_kernel(const T* x, T* y, int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int VEC = 16 / sizeof(T);
  const int cv = C / VEC;
  const size_t n = (size_t)N * Ho * Wo * cv;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int c = (idx % cv) * VEC;
  size_t t = idx / cv;
  int wo = t % Wo; t /= Wo;
  int ho = t % Ho;
  int img = t / Ho;
  float m[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) m[i] = -INFINITY;
  for (int r = 0; r < 3; ++r) {
    int h = ho * 2 - 1 + r;
    if ((unsigned)h >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      int w = wo * 2 - 1 + s;
      if ((unsigned)w >= (unsigned)W) continue;
      uint4 v = *(const uint4*)(x + ((size_t)(img * H + h) * W + w) * C + c);
      const T* e = (const T*)&v;
#pragma unroll
      for (int i = 0; i < VEC; ++i) m[i] = fmaxf(m[i], Elem<T>::load(e, i));
    }
  }
  uint4 o;
  T* eo = (T*)&o;
#pragma unroll
  for (int i = 0; i < VEC; ++i) Elem<T>::store(eo, i, m[i]);
  *(uint4*)(y + ((size_t)(img * Ho + ho) * Wo + wo) * C + c) = o;
}

// one wave per row; cols <= 64*MAXJ
constexpr int LN_MAXJ = 16;

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const T* x, const T* r, const float* gamma, const float* beta,
                                                                T* y, T* s_out, float* mean, float* rstd, int rows, int cols,
                                                                float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * cols;
  float v[LN_MAXJ];
  float sum = 0.f;
  const int nj = (cols + 63) / 64;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    v[j] = 0.f;
    int c = lane + 64 * j;
    if (j < nj && c < cols) {
      float a = Elem<T>::load(x, base + c);
      if (r) a += Elem<T>::load(r, base + c);
      v[j] = a;
      sum += a;
    }
  }
  const float mu = wave_sum(sum) / cols;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int c = lane + 64 * j;
    if (j < nj && c < cols) {
      float dlt = v[j] - mu;
      var += dlt * dlt;
    }
  }
  var = wave_sum(var) / cols;
  const float rs = rsqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    int c = lane + 64 * j;
    if (j < nj && c < cols) {
      if (s_out) Elem<T>::store(s_out, base + c, v[j]);
      Elem<T>::store(y, base + c, (v[j] - mu) * rs * gamma[c] + beta[c]);
    }
  }
}

__device__ __forceinline__ void unpack4v(const uint2& v, float (&o)[4]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack4_bf16(const float (&v)[4]) {  // same rounding as Elem<u16>::store
  return make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16), (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
}
static bool aligned8(const void* a, const void* b, const void* c, const void* d) {  // null pointers are fine
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 7) == 0;
}

// bf16 rows whose length is a multiple of 256 (d_model = 256, RoBERTa's 768): a lane owns 4 consecutive columns per
// 256-column block - 8-byte loads / stores (the scalar kernels move 128 B per wave instruction and leave the launch
// latency-bound: 35 us per LayerNorm backward over 30 200 x 256 rows, 2 % of the step at 8 clips).
template <int NJ>
__global__ __launch_bounds__(256) void add_layernorm_fwd_v4_kernel(const u16* __restrict__ x, const u16* __restrict__ r, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, u16* __restrict__ y, u16* __restrict__ s_out,
                                                                   float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps) {
  constexpr int cols = NJ * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * cols + lane * 4;
  float v[NJ][4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    unpack4v(*(const uint2*)(x + base + 256 * j), v[j]);
    if (r) {
      float a[4];
      unpack4v(*(const uint2*)(r + base + 256 * j), a);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] += a[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += v[j][e];
  }
  const float mu = wave_sum(sum) * (1.f / cols);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[j][e] - mu;
      var += d * d;
    }
  var = wave_sum(var) * (1.f / cols);
  const float rs = rsqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 gm = *(const float4*)(gamma + lane * 4 + 256 * j), bt = *(const float4*)(beta + lane * 4 + 256 * j);
    if (s_out) *(uint2*)(s_out + base + 256 * j) = pack4_bf16(v[j]);
    float o[4] = {(v[j][0] - mu) * rs * gm.x + bt.x, (v[j][1] - mu) * rs * gm.y + bt.y, (v[j][2] - mu) * rs * gm.z + bt.z,
                  (v[j][3] - mu) * rs * gm.w + bt.w};
    *(uint2*)(y + base + 256 * j) = pack4_bf16(o);
  }
}

template <int NJ>
__global__ __launch_bounds__(256) void add_layernorm_bwd_v4_kernel(const u16* __restrict__ dy, const u16* __restrict__ s, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const u16* __restrict__ extra, u16* __restrict__ ds, float* dgamma, float* dbeta,
                                                                   int rows) {
  constexpr int cols = NJ * 256;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float pg[NJ][4], pb[NJ][4], gm[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 g4 = *(const float4*)(gamma + lane * 4 + 256 * j);
    gm[j][0] = g4.x; gm[j][1] = g4.y; gm[j][2] = g4.z; gm[j][3] = g4.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) pg[j][e] = pb[j][e] = 0.f;
  }
  for (int row = wid; row < rows; row += nw) {
    const size_t base = (size_t)row * cols + lane * 4;
    const float mu = mean[row], rs = rstd[row];
    float xh[NJ][4], dg[NJ][4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float d[4], sv[4];
      unpack4v(*(const uint2*)(dy + base + 256 * j), d);
      unpack4v(*(const uint2*)(s + base + 256 * j), sv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[j][e] = (sv[e] - mu) * rs;
        dg[j][e] = d[e] * gm[j][e];
        c1 += dg[j][e];
        c2 += dg[j][e] * xh[j][e];
        pg[j][e] += d[e] * xh[j][e];
        pb[j][e] += d[e];
      }
    }
    c1 = wave_sum(c1) * (1.f / cols);
    c2 = wave_sum(c2) * (1.f / cols);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (dg[j][e] - c1 - xh[j][e] * c2);
      if (extra) {
        float ex[4];
        unpack4v(*(const uint2*)(extra + base + 256 * j), ex);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += ex[e];
      }
      *(uint2*)(ds + base + 256 * j) = pack4_bf16(o);
    }
  }
  // combine the 4 waves of the workgroup in LDS, then one atomic per column per workgroup
  __shared__ float redg[4][cols], redb[4][cols];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      redg[wv][lane * 4 + 256 * j + e] = pg[j][e];
      redb[wv][lane * 4 + 256 * j + e] = pb[j][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    if (dgamma) atomicAdd(dgamma + c, redg[0][c] + redg[1][c] + redg[2][c] + redg[3][c]);
    if (dbeta) atomicAdd(dbeta + c, redb[0][c] + redb[1][c] + redb[2][c] + redb[3][c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const T* dy, const T* s, const float* mean, const float* rstd,
                                                                const float* gamma, const T* extra, T* ds, float* dgamma,
                                                                float* dbeta, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  const int nj = (cols + 63) / 64;
  float pg[LN_MAXJ], pb[LN_MAXJ];
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) pg[j] = pb[j] = 0.f;
  for (int row = wid; row < rows; row += nw) {
    const size_t base = (size_t)row * cols;
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXJ], dg[LN_MAXJ];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      xh[j] = dg[j] = 0.f;
      int c = lane + 64 * j;
      if (j < nj && c < cols) {
        float d = Elem<T>::load(dy, base + c);
        xh[j] = (Elem<T>::load(s, base + c) - mu) * rs;
        dg[j] = d * gamma[c];
        c1 += dg[j];
        c2 += dg[j] * xh[j];
        pg[j] += d * xh[j];
        pb[j] += d;
      }
    }
    c1 = wave_sum(c1) / cols;
    c2 = wave_sum(c2) / cols;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      int c = lane + 64 * j;
      if (j < nj && c < cols) {
        float o = rs * (dg[j] - c1 - xh[j] * c2);
        if (extra) o += Elem<T>::load(extra, base + c);
        Elem<T>::store(ds, base + c, o);
      }
    }
  }
  // combine the 4 waves of the workgroup in LDS, then one atomic per column per workgroup
  __shared__ float redg[4][64 * LN_MAXJ], redb[4][64 * LN_MAXJ];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    if (j < nj) {
      redg[wv][lane + 64 * j] = pg[j];
      redb[wv][lane + 64 * j] = pb[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    if (dgamma) atomicAdd(dgamma + c, redg[0][c] + redg[1][c] + redg[2][c] + redg[3][c]);
    if (dbeta) atomicAdd(dbeta + c, redb[0][c] + redb[1][c] + redb[2][c] + redb[3][c]);
  }
}

template <typename T>
__global__ void colsum_kernel(const T* g, float* out, int rows, int cols, int ld, int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += Elem<T>::load(g, (size_t)r * ld + c);
  atomicAdd(out + c, acc);
}

// vectorised column sums: thread = (column group of 16 bytes, row lane); 16-byte loads, LDS reduction over the 8
// row lanes, one atomic per column per workgroup
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* g, float* out, int rows, int cols, int ld, int rows_per_block) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[8][32 * VEC + 1];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + cg) * VEC;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (c0 < cols) {
    for (int r = r0 + rl; r < r1; r += 8) {
      uint4 v = *(const uint4*)(g + (size_t)r * ld + c0);
      const T* e = (const T*)&v;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += Elem<T>::load(e, i);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[rl][cg * VEC + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
    const int col = blockIdx.x * 32 * VEC + c;
    if (col < cols) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += red[j][c];
      atomicAdd(out + col, sum);
    }
  }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// d/dx of x * Phi(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// elementwise kernels move 16 bytes per lane per access (8 bf16 / 4 fp32); OP selects the operation:
// 0: y = a + b (b optional)   1: y = a * scale where b > 0 else 0 (ReLU backward)   2: y = dropout(a)
// 3: y = gelu(a) (exact erf form, HF RoBERTa's intermediate activation)   4: y = a * gelu'(b) (its backward, b = pre-activation)
template <typename T, int OP>
__global__ __launch_bounds__(256) void ew_kernel(const T* a, const T* b, T* y, size_t n, float scale, uint32_t thresh, uint32_t seed_in, const uint32_t* seed_dev) {
  const uint32_t seed = (OP == 2) ? effective_seed(seed_in, seed_dev) : seed_in;
  constexpr int VEC = 16 / sizeof(T);
  const size_t nv = n / VEC;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t v = i; v < nv; v += stride) {
    uint4 va = ((const uint4*)a)[v], vb = make_uint4(0, 0, 0, 0), vo;
    if (b) vb = ((const uint4*)b)[v];
    const T* ea = (const T*)&va;
    const T* eb = (const T*)&vb;
    T* eo = (T*)&vo;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float x = Elem<T>::load(ea, k), r;
      if (OP == 0) r = x + (b ? Elem<T>::load(eb, k) : 0.f);
      else if (OP == 1) r = Elem<T>::load(eb, k) > 0.f ? x * scale : 0.f;
      else if (OP == 3) r = gelu_f(x);
      else if (OP == 4) r = x * gelu_grad_f(Elem<T>::load(eb, k));
      else r = dropout_keep(seed, (uint32_t)(v * VEC + k), thresh) ? x * scale : 0.f;
      Elem<T>::store(eo, k, r);
    }
    ((uint4*)y)[v] = vo;
  }
  for (size_t k = nv * VEC + i; k < n; k += stride) {  // tail
    float x = Elem<T>::load(a, k), r;
    if (OP == 0) r = x + (b ? Elem<T>::load(b, k) : 0.f);
    else if (OP == 1) r = Elem<T>::load(b, k) > 0.f ? x * scale : 0.f;
    else if (OP == 3) r = gelu_f(x);
    else if (OP == 4) r = x * gelu_grad_f(Elem<T>::load(b, k));
    else r = dropout_keep(seed, (uint32_t)k, thresh) ? x * scale : 0.f;
    Elem<T>::store(y, k, r);
  }
}

// One workgroup per image.  Phase 1: the normalised cumulative coordinates of every token (position_encoding.py:71-83) into
// LDS, the 1 / temperature^(2i/npf) table beside them.  Phase 2: one thread per 16-byte run of channels (8 bf16 / 4 fp32) ->
// 16-byte stores over whole rows (the first version stored 2-byte elements at a 512-byte stride per thread: 19x the result's
// bytes reached HBM).  `rows` >= h*w rows are written per image; the rows behind the h*w tokens are ZERO: the encoder's
// positional operand covers the text tokens with zeros (transformer.py:323-326), produced here instead of a torch.cat.
template <typename T>
__global__ __launch_bounds__(256) void pos_sine_kernel(const uint8_t* mask, T* pos, int h, int w, int npf, float temperature, int rows) {
  extern __shared__ float sh[];  // [h*w] y coordinate, [h*w] x coordinate, [npf] dim_t
  const int img = blockIdx.x;
  const int hw = h * w;
  const uint8_t* m = mask + (size_t)img * hw;
  const float two_pi = 6.283185307179586f;
  float* dimt = sh + 2 * hw;
  for (int i = threadIdx.x; i < npf; i += blockDim.x) dimt[i] = powf(temperature, (float)(2 * (i / 2)) / (float)npf);
  for (int tkn = threadIdx.x; tkn < hw; tkn += blockDim.x) {
    const int y = tkn / w, x = tkn - y * w;
    float ye = 0.f, xe = 0.f, ylast = 0.f, xlast = 0.f;
    for (int yy = 0; yy < h; ++yy) {
      float nm = m[yy * w + x] ? 0.f : 1.f;
      ylast += nm;
      if (yy <= y) ye += nm;
    }
    for (int xx = 0; xx < w; ++xx) {
      float nm = m[y * w + xx] ? 0.f : 1.f;
      xlast += nm;
      if (xx <= x) xe += nm;
    }
    sh[tkn] = ye / (ylast + 1e-6f) * two_pi;
    sh[hw + tkn] = xe / (xlast + 1e-6f) * two_pi;
  }
  __syncthreads();
  constexpr int EPL = 16 / sizeof(T);
  const int C = 2 * npf, chunks = C / EPL;  // npf % EPL == 0 (host-checked): a run never straddles the y / x halves
  char* base = (char*)pos + (size_t)img * rows * C * sizeof(T);
  for (int idx = threadIdx.x; idx < rows * chunks; idx += blockDim.x) {
    const int row = idx / chunks, c0 = (idx - row * chunks) * EPL;
    float v[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) v[r] = 0.f;
    if (row < hw) {
      const bool xhalf = c0 >= npf;
      const float e = xhalf ? sh[hw + row] : sh[row];
      const int i0 = xhalf ? c0 - npf : c0;
#pragma unroll
      for (int r = 0; r < EPL; ++r) {
        const float val = e / dimt[i0 + r];
        v[r] = ((i0 + r) & 1) ? cosf(val) : sinf(val);
      }
    }
    uint4 o;
    if constexpr (sizeof(T) == 2) {
      o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
      o.z = (uint32_t)f32_to_bf16(v[4 % EPL]) | ((uint32_t)f32_to_bf16(v[5 % EPL]) << 16);
      o.w = (uint32_t)f32_to_bf16(v[6 % EPL]) | ((uint32_t)f32_to_bf16(v[7 % EPL]) << 16);
    } else {
      o = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
    *(uint4*)(base + ((size_t)row * C + c0) * sizeof(T)) = o;
  }
}

// dst[dst_map[i]] = src[src_map[i]] (+ add[i]): one thread per 16-byte run of a row
template <typename T>
__global__ __launch_bounds__(256) void rows_copy_kernel(const char* __restrict__ src, const int* __restrict__ src_map, const char* __restrict__ add,
                                                        char* __restrict__ dst, const int* __restrict__ dst_map, int n_rows, int chunks, int ld_src,
                                                        int ld_add, int ld_dst) {
  constexpr int EPL = 16 / sizeof(T);
  const size_t total = (size_t)n_rows * chunks;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / chunks), c = (int)(idx - (size_t)i * chunks);
    const size_t rs = src_map ? (size_t)src_map[i] : (size_t)i, rd = dst_map ? (size_t)dst_map[i] : (size_t)i;
    uint4 v = *(const uint4*)(src + (rs * ld_src + (size_t)c * EPL) * sizeof(T));
    if (add) {
      const uint4 a = *(const uint4*)(add + ((size_t)i * ld_add + (size_t)c * EPL) * sizeof(T));
      if constexpr (sizeof(T) == 2) {
        const uint32_t* pv = (const uint32_t*)&v;
        const uint32_t* pa = (const uint32_t*)&a;
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(pv[q] << 16) + __uint_as_float(pa[q] << 16);
          const float hi = __uint_as_float(pv[q] & 0xffff0000u) + __uint_as_float(pa[q] & 0xffff0000u);
          o[q] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
        }
        v = make_uint4(o[0], o[1], o[2], o[3]);
      } else {
        v = make_uint4(__float_as_uint(__uint_as_float(v.x) + __uint_as_float(a.x)), __float_as_uint(__uint_as_float(v.y) + __uint_as_float(a.y)),
                       __float_as_uint(__uint_as_float(v.z) + __uint_as_float(a.z)), __float_as_uint(__uint_as_float(v.w) + __uint_as_float(a.w)));
      }
    }
    *(uint4*)(dst + (rd * ld_dst + (size_t)c * EPL) * sizeof(T)) = v;
  }
}

// out[r] = sum over its segment of input rows, fp32 accumulation: one thread per 16-byte run of an output row
template <typename T>
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(const char* __restrict__ in, const int* __restrict__ idx, const int* __restrict__ ptr,
                                                               char* __restrict__ out, const int* __restrict__ out_map, int n_out, int chunks, int ld_in,
                                                               int ld_out) {
  constexpr int EPL = 16 / sizeof(T);
  const size_t total = (size_t)n_out * chunks;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(t / chunks), c = (int)(t - (size_t)r * chunks);
    float acc[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) acc[q] = 0.f;
    const int j0 = ptr[r], j1 = ptr[r + 1];
    for (int j = j0; j < j1; ++j) {
      const uint4 v = *(const uint4*)(in + ((size_t)idx[j] * ld_in + (size_t)c * EPL) * sizeof(T));
      if constexpr (sizeof(T) == 2) {
        const uint32_t* pv = (const uint32_t*)&v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[2 * q] += __uint_as_float(pv[q] << 16);
          acc[2 * q + 1] += __uint_as_float(pv[q] & 0xffff0000u);
        }
      } else {
        acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y); acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
      }
    }
    uint4 o;
    if constexpr (sizeof(T) == 2) {
      o.x = (uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16);
      o.y = (uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16);
      o.z = (uint32_t)f32_to_bf16(acc[4 % EPL]) | ((uint32_t)f32_to_bf16(acc[5 % EPL]) << 16);
      o.w = (uint32_t)f32_to_bf16(acc[6 % EPL]) | ((uint32_t)f32_to_bf16(acc[7 % EPL]) << 16);
    } else {
      o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
    }
    *(uint4*)(out + ((size_t)(out_map ? out_map[r] : r) * ld_out + (size_t)c * EPL) * sizeof(T)) = o;
  }
}

}  // namespace td
using namespace td;

static inline unsigned nblk(size_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }
#define TD_DISPATCH(dtype, CALL_BF16, CALL_F32, who)  \
  if (dtype == TD_BF16) { CALL_BF16; }                \
  else if (dtype == TD_F32) { CALL_F32; }             \
  else TD_REQUIRE(false, who ": bad dtype")

extern "C" int td_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, td_stream_t stream) {
  TD_REQUIRE(x && y, "td_maxpool3x3s2: null pointer");
  const int vec = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(C % vec == 0, "td_maxpool3x3s2: C must be a multiple of %d", vec);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  size_t n = (size_t)N * Ho * Wo * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  TD_DISPATCH(dtype, (maxpool3x3s2_kernel<u16><<<nblk(n), 256, 0, st>>>((const u16*)x, (u16*)y, N, H, W, C, Ho, Wo)),
              (maxpool3x3s2_kernel<float><<<nblk(n), 256, 0, st>>>((const float*)x, (float*)y, N, H, W, C, Ho, Wo)),
              "td_maxpool3x3s2");
  return check_launch("td_maxpool3x3s2");
}

extern "C" int td_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* y,
                                    void* s_out, float* mean, float* rstd, int rows, int cols, float eps, int dtype,
                                    td_stream_t stream) {
  TD_REQUIRE(x && gamma && beta && y, "td_add_layernorm_fwd: null pointer");
  TD_REQUIRE(cols <= 64 * LN_MAXJ, "td_add_layernorm_fwd: cols > %d", 64 * LN_MAXJ);
  if (rows == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned g = (rows + 3) / 4;
  if (dtype == TD_BF16 && (cols == 256 || cols == 768) && aligned8(x, r, y, s_out)) {
    if (cols == 256) add_layernorm_fwd_v4_kernel<1><<<g, 256, 0, st>>>((const u16*)x, (const u16*)r, gamma, beta, (u16*)y, (u16*)s_out, mean, rstd, rows, eps);
    else add_layernorm_fwd_v4_kernel<3><<<g, 256, 0, st>>>((const u16*)x, (const u16*)r, gamma, beta, (u16*)y, (u16*)s_out, mean, rstd, rows, eps);
    return check_launch("td_add_layernorm_fwd");
  }
  TD_DISPATCH(dtype,
              (add_layernorm_fwd_kernel<u16><<<g, 256, 0, st>>>((const u16*)x, (const u16*)r, gamma, beta, (u16*)y, (u16*)s_out, mean, rstd, rows, cols, eps)),
              (add_layernorm_fwd_kernel<float><<<g, 256, 0, st>>>((const float*)x, (const float*)r, gamma, beta, (float*)y, (float*)s_out, mean, rstd, rows, cols, eps)),
              "td_add_layernorm_fwd");
  return check_launch("td_add_layernorm_fwd");
}

extern "C" int td_add_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd,
                                    const float* gamma, const void* extra, void* ds, float* dgamma, float* dbeta,
                                    int rows, int cols, int dtype, td_stream_t stream) {
  TD_REQUIRE(dy && s && mean && rstd && gamma && ds, "td_add_layernorm_bwd: null pointer");
  TD_REQUIRE(cols <= 64 * LN_MAXJ, "td_add_layernorm_bwd: cols > %d", 64 * LN_MAXJ);
  if (rows == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned g = (rows + 3) / 4;
  if (dtype == TD_BF16 && (cols == 256 || cols == 768) && aligned8(dy, s, ds, extra)) {
    // ~8 rows per wavefront, at most 1024 workgroups (each ends with 2 * cols atomics)
    unsigned gv = (rows + 31) / 32;
    if (gv > 1024) gv = 1024;
    if (deterministic()) gv = 1;  // one workgroup: dgamma / dbeta get ONE atomic per column (into the caller's zeros), in a fixed order
    if (cols == 256) add_layernorm_bwd_v4_kernel<1><<<gv, 256, 0, st>>>((const u16*)dy, (const u16*)s, mean, rstd, gamma, (const u16*)extra, (u16*)ds, dgamma, dbeta, rows);
    else add_layernorm_bwd_v4_kernel<3><<<gv, 256, 0, st>>>((const u16*)dy, (const u16*)s, mean, rstd, gamma, (const u16*)extra, (u16*)ds, dgamma, dbeta, rows);
    return check_launch("td_add_layernorm_bwd");
  }
  if (g > 256) g = 256;
  if (deterministic()) g = 1;
  TD_DISPATCH(dtype,
              (add_layernorm_bwd_kernel<u16><<<g, 256, 0, st>>>((const u16*)dy, (const u16*)s, mean, rstd, gamma, (const u16*)extra, (u16*)ds, dgamma, dbeta, rows, cols)),
              (add_layernorm_bwd_kernel<float><<<g, 256, 0, st>>>((const float*)dy, (const float*)s, mean, rstd, gamma, (const float*)extra, (float*)ds, dgamma, dbeta, rows, cols)),
              "td_add_layernorm_bwd");
  return check_launch("td_add_layernorm_bwd");
}

extern "C" int td_colsum(const void* g, float* out, int rows, int cols, int ld, int dtype, td_stream_t stream) {
  TD_REQUIRE(g && out, "td_colsum: null pointer");
  if (rows == 0 || cols == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == TD_BF16 ? 8 : 4;
  if (cols % vec == 0 && ld % vec == 0) {
    const int gx = (cols + 32 * vec - 1) / (32 * vec);
    int rpb = (int)(((long long)rows * gx + 511) / 512);  // ~512 workgroups
    if (rpb < 64) rpb = 64;
    if (deterministic()) rpb = rows;  // one workgroup per column block
    dim3 grid(gx, (rows + rpb - 1) / rpb);
    TD_DISPATCH(dtype, (colsum_vec_kernel<u16><<<grid, 256, 0, st>>>((const u16*)g, out, rows, cols, ld, rpb)),
                (colsum_vec_kernel<float><<<grid, 256, 0, st>>>((const float*)g, out, rows, cols, ld, rpb)), "td_colsum");
    return check_launch("td_colsum");
  }
  // enough row chunks to fill the chip: ~1024 workgroups, at least 8 rows each
  int rpb = (int)(((long long)rows * ((cols + 255) / 256) + 1023) / 1024);
  if (rpb < 8) rpb = 8;
  if (deterministic()) rpb = rows;
  dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb);
  TD_DISPATCH(dtype, (colsum_kernel<u16><<<grid, 256, 0, st>>>((const u16*)g, out, rows, cols, ld, rpb)),
              (colsum_kernel<float><<<grid, 256, 0, st>>>((const float*)g, out, rows, cols, ld, rpb)), "td_colsum");
  return check_launch("td_colsum");
}

extern "C" int td_add(const void* a, const void* b, void* y, size_t n, int dtype, td_stream_t stream) {
  TD_REQUIRE(a && y, "td_add: null pointer");
  if (n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned g = nblk(n / 4 + 1);
  if (g > 4096) g = 4096;
  TD_REQUIRE(((uintptr_t)a | (uintptr_t)y | (uintptr_t)b) % 16 == 0, "td_add: pointers must be 16-byte aligned");
  TD_DISPATCH(dtype, (ew_kernel<u16, 0><<<g, 256, 0, st>>>((const u16*)a, (const u16*)b, (u16*)y, n, 1.f, 0, 0, nullptr)),
              (ew_kernel<float, 0><<<g, 256, 0, st>>>((const float*)a, (const float*)b, (float*)y, n, 1.f, 0, 0, nullptr)), "td_add");
  return check_launch("td_add");
}

extern "C" int td_relu_bwd(const void* dy, const void* y, void* g, size_t n, float scale, int dtype, td_stream_t stream) {
  TD_REQUIRE(dy && y && g, "td_relu_bwd: null pointer");
  if (n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned gr = nblk(n);
  if (gr > 4096) gr = 4096;
  TD_REQUIRE(((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g) % 16 == 0, "td_relu_bwd: pointers must be 16-byte aligned");
  TD_DISPATCH(dtype, (ew_kernel<u16, 1><<<gr, 256, 0, st>>>((const u16*)dy, (const u16*)y, (u16*)g, n, scale, 0, 0, nullptr)),
              (ew_kernel<float, 1><<<gr, 256, 0, st>>>((const float*)dy, (const float*)y, (float*)g, n, scale, 0, 0, nullptr)), "td_relu_bwd");
  return check_launch("td_relu_bwd");
}

extern "C" int td_gelu_fwd(const void* x, void* y, size_t n, int dtype, td_stream_t stream) {
  TD_REQUIRE(x && y, "td_gelu_fwd: null pointer");
  if (n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned gr = nblk(n);
  if (gr > 4096) gr = 4096;
  TD_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "td_gelu_fwd: pointers must be 16-byte aligned");
  TD_DISPATCH(dtype, (ew_kernel<u16, 3><<<gr, 256, 0, st>>>((const u16*)x, nullptr, (u16*)y, n, 1.f, 0, 0, nullptr)),
              (ew_kernel<float, 3><<<gr, 256, 0, st>>>((const float*)x, nullptr, (float*)y, n, 1.f, 0, 0, nullptr)), "td_gelu_fwd");
  return check_launch("td_gelu_fwd");
}

extern "C" int td_gelu_bwd(const void* dy, const void* x, void* dx, size_t n, int dtype, td_stream_t stream) {
  TD_REQUIRE(dy && x && dx, "td_gelu_bwd: null pointer");
  if (n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned gr = nblk(n);
  if (gr > 4096) gr = 4096;
  TD_REQUIRE(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) % 16 == 0, "td_gelu_bwd: pointers must be 16-byte aligned");
  TD_DISPATCH(dtype, (ew_kernel<u16, 4><<<gr, 256, 0, st>>>((const u16*)dy, (const u16*)x, (u16*)dx, n, 1.f, 0, 0, nullptr)),
              (ew_kernel<float, 4><<<gr, 256, 0, st>>>((const float*)dy, (const float*)x, (float*)dx, n, 1.f, 0, 0, nullptr)), "td_gelu_bwd");
  return check_launch("td_gelu_bwd");
}

extern "C" int td_dropout(const void* x, void* y, size_t n, float p, uint32_t seed, const uint32_t* dropout_counter, int dtype,
                          td_stream_t stream) {
  TD_REQUIRE(x && y, "td_dropout: null pointer");
  TD_REQUIRE(p >= 0.f && p < 1.f, "td_dropout: p out of range");
  if (n == 0) return TD_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned gr = nblk(n);
  if (gr > 4096) gr = 4096;
  uint32_t thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  if (p > 0.f && !thresh) thresh = 1;
  float scale = 1.f / (1.f - p);
  TD_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "td_dropout: pointers must be 16-byte aligned");
  TD_DISPATCH(dtype, (ew_kernel<u16, 2><<<gr, 256, 0, st>>>((const u16*)x, nullptr, (u16*)y, n, scale, thresh, seed, thresh ? dropout_counter : nullptr)),
              (ew_kernel<float, 2><<<gr, 256, 0, st>>>((const float*)x, nullptr, (float*)y, n, scale, thresh, seed, thresh ? dropout_counter : nullptr)), "td_dropout");
  return check_launch("td_dropout");
}

extern "C" int td_pos_sine(const uint8_t* mask, void* pos, int N, int h, int w, int npf, float temperature, int rows_per_image, int dtype,
                           td_stream_t stream) {
  TD_REQUIRE(mask && pos, "td_pos_sine: null pointer");
  if (N == 0) return TD_OK;
  const int rows = rows_per_image > 0 ? rows_per_image : h * w;
  const int epl = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(h >= 1 && w >= 1 && rows >= h * w, "td_pos_sine: rows_per_image=%d < h*w=%d", rows, h * w);
  TD_REQUIRE(npf >= epl && npf % epl == 0, "td_pos_sine: num_pos_feats=%d must be a multiple of %d", npf, epl);
  const size_t lds = (size_t)(2 * h * w + npf) * sizeof(float);
  TD_REQUIRE(lds <= 64 * 1024, "td_pos_sine: feature map of %d x %d tokens is too large", h, w);
  hipStream_t st = (hipStream_t)stream;
  TD_DISPATCH(dtype, (pos_sine_kernel<u16><<<N, 256, lds, st>>>(mask, (u16*)pos, h, w, npf, temperature, rows)),
              (pos_sine_kernel<float><<<N, 256, lds, st>>>(mask, (float*)pos, h, w, npf, temperature, rows)), "td_pos_sine");
  return check_launch("td_pos_sine");
}

extern "C" int td_rows_copy(const void* src, const int* src_map, const void* add, void* dst, const int* dst_map, int n_rows, int cols, int ld_src,
                            int ld_add, int ld_dst, int dtype, td_stream_t stream) {
  TD_REQUIRE(src && dst, "td_rows_copy: null pointer");
  if (n_rows <= 0) return TD_OK;
  const int epl = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(cols >= epl && cols % epl == 0 && ld_src % epl == 0 && ld_dst % epl == 0 && (!add || ld_add % epl == 0) && ld_src >= cols && ld_dst >= cols,
             "td_rows_copy: cols / row strides must be multiples of %d", epl);
  TD_REQUIRE(((uintptr_t)src | (uintptr_t)dst | (uintptr_t)add) % 16 == 0, "td_rows_copy: pointers must be 16-byte aligned");
  const int chunks = cols / epl;
  hipStream_t st = (hipStream_t)stream;
  unsigned g = nblk((size_t)n_rows * chunks);
  if (g > 16384) g = 16384;
  TD_DISPATCH(dtype, (rows_copy_kernel<u16><<<g, 256, 0, st>>>((const char*)src, src_map, (const char*)add, (char*)dst, dst_map, n_rows, chunks, ld_src, ld_add, ld_dst)),
              (rows_copy_kernel<float><<<g, 256, 0, st>>>((const char*)src, src_map, (const char*)add, (char*)dst, dst_map, n_rows, chunks, ld_src, ld_add, ld_dst)),
              "td_rows_copy");
  return check_launch("td_rows_copy");
}

extern "C" int td_rows_segment_sum(const void* in, const int* idx, const int* ptr, void* out, const int* out_map, int n_out, int cols, int ld_in,
                                   int ld_out, int dtype, td_stream_t stream) {
  TD_REQUIRE(in && idx && ptr && out, "td_rows_segment_sum: null pointer");
  if (n_out <= 0) return TD_OK;
  const int epl = dtype == TD_BF16 ? 8 : 4;
  TD_REQUIRE(cols >= epl && cols % epl == 0 && ld_in % epl == 0 && ld_out % epl == 0 && ld_in >= cols && ld_out >= cols,
             "td_rows_segment_sum: cols / row strides must be multiples of %d", epl);
  TD_REQUIRE(((uintptr_t)in | (uintptr_t)out) % 16 == 0, "td_rows_segment_sum: pointers must be 16-byte aligned");
  const int chunks = cols / epl;
  hipStream_t st = (hipStream_t)stream;
  unsigned g = nblk((size_t)n_out * chunks);
  if (g > 16384) g = 16384;
  TD_DISPATCH(dtype, (rows_segment_sum_kernel<u16><<<g, 256, 0, st>>>((const char*)in, idx, ptr, (char*)out, out_map, n_out, chunks, ld_in, ld_out)),
              (rows_segment_sum_kernel<float><<<g, 256, 0, st>>>((const char*)in, idx, ptr, (char*)out, out_map, n_out, chunks, ld_in, ld_out)), "td_rows_segment_sum");
  return check_launch("td_rows_segment_sum");
}
