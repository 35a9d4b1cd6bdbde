// Layout / precision preparation kernels: FrozenBatchNorm folding into conv weights, weight-gradient
// un-folding, NCHW<->NHWC conversion, dtype casts.  All HBM-bound, one pass each.
// Reference call sites: models/backbone.py:60-70 (FrozenBatchNorm2d.forward), engine.py:55 (NCHW fp32
// frames handed over by the data loader), models/backbone.py:98 (features returned as NCHW).
#include "td_common.h"

namespace td {

template <typename T>
__global__ void weight_prep_kernel(const float* W, const float* bn_w, const float* bn_b, const float* bn_rm,
                                   const float* bn_rv, const float* bias, int Co, int Ci, int R, int S, int Cpad,
                                   T* w_fwd, T* w_dgrad, float* bias_out, float* scale_out) {
  const size_t nf = (size_t)Co * R * S * Cpad;
  const size_t nd = w_dgrad ? (size_t)Ci * R * S * Co : 0;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < nf) {
    int ci = idx % Cpad;
    size_t t = idx / Cpad;
    int s = t % S; t /= S;
    int r = t % R;
    int co = t / R;
    float sc = bn_w ? bn_w[co] * rsqrtf(bn_rv[co] + 1e-5f) : 1.f;
    float v = ci < Ci ? W[(((size_t)co * Ci + ci) * R + r) * S + s] * sc : 0.f;
    Elem<T>::store(w_fwd, idx, v);
  } else if (idx - nf < nd) {
    size_t j = idx - nf;
    int co = j % Co;
    size_t t = j / Co;
    int s = t % S; t /= S;
    int r = t % R;
    int ci = t / R;
    float sc = bn_w ? bn_w[co] * rsqrtf(bn_rv[co] + 1e-5f) : 1.f;
    Elem<T>::store(w_dgrad, j, W[(((size_t)co * Ci + ci) * R + r) * S + s] * sc);
  }
  if (idx < (size_t)Co) {
    int co = (int)idx;
    float sc = bn_w ? bn_w[co] * rsqrtf(bn_rv[co] + 1e-5f) : 1.f;
    if (scale_out) scale_out[co] = sc;
    if (bias_out) bias_out[co] = bn_w ? (bn_b[co] - bn_rm[co] * sc) : (bias ? bias[co] : 0.f);
  }
}

// One launch for all layers: workgroup -> (item, 16-channel co tile, 32-channel ci tile); the W slab of the tile
// (16 x [32 ci x RS] contiguous floats) goes through LDS so reads and both writes run over contiguous segments.
// NRS = taps per group as a compile-time constant (1: linears / 1x1 convs, 9: 3x3; 0: generic, e.g. the 7x7 stem) so
// the index arithmetic folds to shifts and constant divisions; bf16 results leave as packed pairs (4-byte stores).
template <typename T, int NRS>
__device__ __forceinline__ void prep_group(const td_prep_item& I, float (&tile)[16][32 * 9 + 1], const float* sc16, int co0, int ci0,
                                           int rs0, int nrs_rt, int t) {
  const int nrs = NRS ? NRS : nrs_rt;
  const int RS = I.RS;
  T* wf = (T*)I.w_fwd;
  T* wd = (T*)I.w_dgrad;
  // load: 16 co x (32 ci x nrs taps); for a full-width tile the 32*nrs floats of one co are contiguous in W
  for (int idx = t; idx < 16 * 32 * nrs; idx += 256) {
    const int c = idx / (32 * nrs), k = idx - c * (32 * nrs);
    const int cil = k / nrs, r = k - cil * nrs;
    const int ci = ci0 + cil, co = co0 + c;
    float v = 0.f;
    if (co < I.Co && ci < I.Ci) v = I.W[((size_t)co * I.Ci + ci) * RS + rs0 + r] * sc16[c];
    tile[c][cil * 9 + r] = v;
  }
  __syncthreads();
  if constexpr (sizeof(T) == 2) {
    // forward layout [Co_alloc][RS][Cpad]: pairs of consecutive ci
    for (int idx = t; idx < 16 * nrs * 16; idx += 256) {
      const int cp = idx & 15, r = (idx >> 4) % nrs, c = idx / (16 * nrs);
      const int co = co0 + c, ci = ci0 + 2 * cp;
      if (co < I.Co_alloc && ci + 1 < I.Cpad) {
        const uint32_t v = (uint32_t)f32_to_bf16(tile[c][(2 * cp) * 9 + r]) | ((uint32_t)f32_to_bf16(tile[c][(2 * cp + 1) * 9 + r]) << 16);
        *(uint32_t*)((u16*)wf + ((size_t)co * RS + rs0 + r) * I.Cpad + ci) = v;
      } else if (co < I.Co_alloc && ci < I.Cpad) {
        Elem<T>::store(wf, ((size_t)co * RS + rs0 + r) * I.Cpad + ci, tile[c][(2 * cp) * 9 + r]);
      }
    }
    // dgrad layout [Ci][RS][Co_alloc]: pairs of consecutive co
    if (wd) {
      for (int idx = t; idx < 32 * nrs * 8; idx += 256) {
        const int cp = idx & 7, r = (idx >> 3) % nrs, cil = idx / (8 * nrs);
        const int co = co0 + 2 * cp, ci = ci0 + cil;
        if (ci < I.Ci && co + 1 < I.Co_alloc) {
          const uint32_t v = (uint32_t)f32_to_bf16(tile[2 * cp][cil * 9 + r]) | ((uint32_t)f32_to_bf16(tile[2 * cp + 1][cil * 9 + r]) << 16);
          *(uint32_t*)((u16*)wd + ((size_t)ci * RS + rs0 + r) * I.Co_alloc + co) = v;
        } else if (ci < I.Ci && co < I.Co_alloc) {
          Elem<T>::store(wd, ((size_t)ci * RS + rs0 + r) * I.Co_alloc + co, tile[2 * cp][cil * 9 + r]);
        }
      }
    }
  } else {
    for (int idx = t; idx < 16 * nrs * 32; idx += 256) {
      const int cil = idx & 31, r = (idx >> 5) % nrs, c = idx / (32 * nrs);
      const int co = co0 + c, ci = ci0 + cil;
      if (co < I.Co_alloc && ci < I.Cpad) Elem<T>::store(wf, ((size_t)co * RS + rs0 + r) * I.Cpad + ci, tile[c][cil * 9 + r]);
    }
    if (wd) {
      for (int idx = t; idx < 32 * nrs * 16; idx += 256) {
        const int c = idx & 15, r = (idx >> 4) % nrs, cil = idx / (16 * nrs);
        const int co = co0 + c, ci = ci0 + cil;
        if (co < I.Co_alloc && ci < I.Ci) Elem<T>::store(wd, ((size_t)ci * RS + rs0 + r) * I.Co_alloc + co, tile[c][cil * 9 + r]);
      }
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void prep_batch_kernel(const td_prep_item* items, int n_items) {
  __shared__ float tile[16][32 * 9 + 1];
  __shared__ float sc16[16];
  // locate the item of this workgroup (binary search over the items' first workgroup index)
  int it = 0;
  {
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (items[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    it = lo;
  }
  const td_prep_item I = items[it];
  const int local = blockIdx.x - I.blk0;
  const int n_cit = (I.Cpad + 31) / 32;
  const int cot = local / n_cit, cit = local - cot * n_cit;
  const int co0 = cot * 16, ci0 = cit * 32;
  const int RS = I.RS;
  const int t = threadIdx.x;
  if (t < 16) {
    const int co = co0 + t;
    sc16[t] = (co < I.Co && I.bn_w) ? I.bn_w[co] * rsqrtf(I.bn_rv[co] + 1e-5f) : 1.f;
  }
  __syncthreads();
  if (RS == 1) {
    prep_group<T, 1>(I, tile, sc16, co0, ci0, 0, 1, t);
  } else if (RS == 9) {
    prep_group<T, 9>(I, tile, sc16, co0, ci0, 0, 9, t);
  } else {
    for (int rs0 = 0; rs0 < RS; rs0 += 9)  // taps in groups of <= 9 (7x7 stem: 6 groups)
      prep_group<T, 0>(I, tile, sc16, co0, ci0, rs0, min(9, RS - rs0), t);
  }
  if (cit == 0 && t < 16) {
    const int co = co0 + t;
    if (co < I.Co_alloc) {
      const bool real = co < I.Co;
      const float sc = (real && I.bn_w) ? I.bn_w[co] * rsqrtf(I.bn_rv[co] + 1e-5f) : 1.f;
      if (I.scale_out) I.scale_out[co] = sc;
      if (I.bias_out) I.bias_out[co] = !real ? 0.f : (I.bn_w ? (I.bn_b[co] - I.bn_rm[co] * sc) : (I.bias ? I.bias[co] : 0.f));
    }
  }
}

__global__ void wgrad_finalize_kernel(const float* dw_k, const float* scale, float* dW, int Co, int Ci, int R, int S,
                                      int Cpad, int accumulate) {
  const size_t n = (size_t)Co * Ci * R * S;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int s = idx % S;
  size_t t = idx / S;
  int r = t % R; t /= R;
  int ci = t % Ci;
  int co = t / Ci;
  float v = dw_k[(((size_t)co * R + r) * S + s) * Cpad + ci] * (scale ? scale[co] : 1.f);
  dW[idx] = accumulate ? dW[idx] + v : v;
}

// frames of one source (NCHW, fp32 or uint8 pixels) -> NHWC T with channels zero-padded to Cpad, optionally picked by a
// device index list and normalised on the fly ((x * in_scale - mean[c]) * inv_std[c]; uint8: in_scale = 1/255).  One thread
// per pixel: C coalesced channel-plane reads, ONE 16-byte store of the Cpad packed channels (8 bf16 / 4 fp32).
struct FrameNorm {
  float mean[4], inv_std[4], in_scale;
};
template <typename T, typename S, int CP>
__global__ __launch_bounds__(256) void frames_to_nhwc_kernel(const S* __restrict__ x, const int* __restrict__ index, const int* __restrict__ valid_hw,
                                                             T* __restrict__ y, int n_frames, int C, int H, int W, FrameNorm nm) {
  const size_t hw = (size_t)H * W;
  const size_t n = (size_t)n_frames * hw;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const size_t img = idx / hw, p = idx - img * hw;
  const size_t src = index ? (size_t)index[img] : img;
  // pixels outside the frame's own extent (a batch of videos padded to a common H x W) are exactly 0 AFTER normalisation,
  // like NestedTensor.from_tensor_list pads the already normalised frames (util/misc.py:158-170)
  bool inside = true;
  if (valid_hw) {
    const int py = (int)(p / W), px = (int)(p - (size_t)py * W);
    inside = py < valid_hw[2 * src] && px < valid_hw[2 * src + 1];
  }
  float v[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    v[c] = 0.f;
    if (c < C && inside) v[c] = ((float)x[(src * C + c) * hw + p] * nm.in_scale - nm.mean[c]) * nm.inv_std[c];
  }
  if constexpr (sizeof(T) == 2 && CP == 4) {  // 4-channel bf16 pixels: the pixel-pair form of the stem (td_resnet_fwd stem_pairs)
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    ((uint2*)y)[idx] = o;
  } else if constexpr (sizeof(T) == 2) {
    static_assert(CP == 8, "bf16 rows are padded to 8 (or 4) channels");
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    ((uint4*)y)[idx] = o;
  } else {
    static_assert(CP == 4, "fp32 rows are padded to 4 channels");
    ((float4*)y)[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// The benchmarked case - uint8 pixels -> 4-channel bf16 pixels (the pixel-pair stem's input), W a multiple of 4 - with FOUR pixels per
// thread: one 4-byte load per colour plane (a wavefront instruction moves 256 bytes instead of 64) and two 16-byte stores.  The
// one-pixel-per-thread form above ran this at 2.4 TB/s of its 1.36 MB per frame (byte loads: 1.33 ms per 16-clip step).
__global__ __launch_bounds__(256) void frames_u8_to_bf16x4_kernel(const uint8_t* __restrict__ x, const int* __restrict__ index, const int* __restrict__ valid_hw,
                                                                  u16* __restrict__ y, int n_frames, int C, int H, int W, FrameNorm nm) {
  const uint32_t hw4 = (uint32_t)(H * W) >> 2, w4 = (uint32_t)W >> 2;
  const uint32_t img = blockIdx.y;
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 pixels inside the frame
  if (q >= hw4) return;
  const size_t src = index ? (size_t)index[img] : img;
  const uint32_t py = q / w4, px = (q - py * w4) * 4;
  int vh = H, vw = W;
  if (valid_hw) {
    vh = valid_hw[2 * src];
    vw = valid_hw[2 * src + 1];
  }
  const size_t plane = (size_t)H * W;
  uint32_t raw[3] = {0u, 0u, 0u};
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (c < C) raw[c] = *(const uint32_t*)(x + (src * C + c) * plane + (size_t)q * 4);
  uint4 o[2];
  uint32_t* ow = (uint32_t*)o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool inside = (int)py < vh && (int)(px + k) < vw;  // outside the frame's own extent: exactly 0 AFTER normalisation
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (c < C && inside) ? ((float)((raw[c] >> (8 * k)) & 0xffu) * nm.in_scale - nm.mean[c]) * nm.inv_std[c] : 0.f;
    ow[2 * k] = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    ow[2 * k + 1] = (uint32_t)f32_to_bf16(v[2]);
  }
  uint4* dst = (uint4*)(y + ((size_t)img * plane + (size_t)q * 4) * 4);
  dst[0] = o[0];
  dst[1] = o[1];
}

// generic fallback (any C / Cpad)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int N, int C, int H, int W, int Cpad) {
  const size_t n = (size_t)N * H * W;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const size_t hw = (size_t)H * W;
  const size_t img = idx / hw, p = idx - img * hw;
  for (int c = 0; c < Cpad; ++c) {
    float v = c < C ? x[(img * C + c) * hw + p] : 0.f;
    Elem<T>::store(y, idx * Cpad + c, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int N, int C, int H, int W) {
  const size_t n = (size_t)N * C * H * W;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const size_t hw = (size_t)H * W;
  size_t p = idx % hw;
  size_t t = idx / hw;
  int c = t % C;
  size_t img = t / C;
  y[idx] = Elem<T>::load(x, (img * hw + p) * C + c);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* x, TD* y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Elem<TD>::store(y, i, Elem<TS>::load(x, i));
}

}  // namespace td
using namespace td;

static inline unsigned nblk(size_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

extern "C" int td_weight_prep(const float* W, const float* bn_w, const float* bn_b, const float* bn_rm,
                              const float* bn_rv, const float* bias, int Co, int Ci, int R, int S, int Cpad,
                              void* w_fwd, void* w_dgrad, float* bias_out, float* scale_out, int dtype,
                              td_stream_t stream) {
  TD_REQUIRE(W && w_fwd, "td_weight_prep: null pointer");
  TD_REQUIRE(Cpad >= Ci, "td_weight_prep: Cpad < Ci");
  TD_REQUIRE(!bn_w || (bn_b && bn_rm && bn_rv), "td_weight_prep: incomplete FrozenBN buffers");
  size_t n = (size_t)Co * R * S * Cpad + (w_dgrad ? (size_t)Ci * R * S * Co : 0);
  if (n < (size_t)Co) n = Co;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16)
    weight_prep_kernel<u16><<<nblk(n), 256, 0, st>>>(W, bn_w, bn_b, bn_rm, bn_rv, bias, Co, Ci, R, S, Cpad, (u16*)w_fwd,
                                                     (u16*)w_dgrad, bias_out, scale_out);
  else if (dtype == TD_F32)
    weight_prep_kernel<float><<<nblk(n), 256, 0, st>>>(W, bn_w, bn_b, bn_rm, bn_rv, bias, Co, Ci, R, S, Cpad,
                                                       (float*)w_fwd, (float*)w_dgrad, bias_out, scale_out);
  else TD_REQUIRE(false, "td_weight_prep: bad dtype");
  return check_launch("td_weight_prep");
}

extern "C" int td_weight_prep_batch(const td_prep_item* items_dev, int n, int total_blocks, int dtype, td_stream_t stream) {
  TD_REQUIRE(items_dev && n > 0 && total_blocks > 0, "td_weight_prep_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) prep_batch_kernel<u16><<<total_blocks, 256, 0, st>>>(items_dev, n);
  else if (dtype == TD_F32) prep_batch_kernel<float><<<total_blocks, 256, 0, st>>>(items_dev, n);
  else TD_REQUIRE(false, "td_weight_prep_batch: bad dtype");
  return check_launch("td_weight_prep_batch");
}

extern "C" int td_wgrad_finalize(const float* dw_k, const float* scale, float* dW, int Co, int Ci, int R, int S,
                                 int Cpad, int accumulate, td_stream_t stream) {
  TD_REQUIRE(dw_k && dW, "td_wgrad_finalize: null pointer");
  size_t n = (size_t)Co * Ci * R * S;
  wgrad_finalize_kernel<<<nblk(n), 256, 0, (hipStream_t)stream>>>(dw_k, scale, dW, Co, Ci, R, S, Cpad, accumulate);
  return check_launch("td_wgrad_finalize");
}

extern "C" int td_frames_to_nhwc(const td_frame_source* srcs, int n_srcs, int C, int H, int W, int Cpad, const float* mean,
                                 const float* inv_std, void* y, int dtype, td_stream_t stream) {
  TD_REQUIRE(srcs && n_srcs >= 1 && y, "td_frames_to_nhwc: bad arguments");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_frames_to_nhwc: bad dtype");
  TD_REQUIRE(C >= 1 && C <= 4 && (Cpad == 4 || (Cpad == 8 && dtype == TD_BF16)), "td_frames_to_nhwc: C must be 1..4 and Cpad 4 (or 8 for bf16)");
  hipStream_t st = (hipStream_t)stream;
  const size_t es = dtype == TD_BF16 ? 2 : 4;
  size_t done = 0;
  for (int s = 0; s < n_srcs; ++s) {
    const td_frame_source& f = srcs[s];
    TD_REQUIRE(f.data && f.n >= 0 && (f.dtype == TD_F32 || f.dtype == TD_U8), "td_frames_to_nhwc: source %d: null data or bad dtype", s);
    if (f.n == 0) continue;
    FrameNorm nm;
    for (int c = 0; c < 4; ++c) {
      nm.mean[c] = (mean && c < C) ? mean[c] : 0.f;
      nm.inv_std[c] = (inv_std && c < C) ? inv_std[c] : 1.f;
    }
    nm.in_scale = f.dtype == TD_U8 ? 1.f / 255.f : 1.f;
    if (f.dtype == TD_F32 && !mean) nm.in_scale = 1.f;
    const size_t n = (size_t)f.n * H * W;
    char* out = (char*)y + done * (size_t)H * W * Cpad * es;
    const unsigned g = nblk(n);
    if (dtype == TD_BF16 && Cpad == 4 && f.dtype == TD_U8 && C <= 3 && W % 4 == 0 && ((uintptr_t)f.data & 3) == 0 && f.n <= 65535) {
      dim3 g4((unsigned)(((size_t)H * W / 4 + 255) / 256), (unsigned)f.n);
      frames_u8_to_bf16x4_kernel<<<g4, 256, 0, st>>>((const uint8_t*)f.data, f.index, f.valid_hw, (u16*)out, f.n, C, H, W, nm);
    } else if (dtype == TD_BF16 && Cpad == 4) {
      if (f.dtype == TD_U8) frames_to_nhwc_kernel<u16, uint8_t, 4><<<g, 256, 0, st>>>((const uint8_t*)f.data, f.index, f.valid_hw, (u16*)out, f.n, C, H, W, nm);
      else frames_to_nhwc_kernel<u16, float, 4><<<g, 256, 0, st>>>((const float*)f.data, f.index, f.valid_hw, (u16*)out, f.n, C, H, W, nm);
    } else if (dtype == TD_BF16) {
      if (f.dtype == TD_U8) frames_to_nhwc_kernel<u16, uint8_t, 8><<<g, 256, 0, st>>>((const uint8_t*)f.data, f.index, f.valid_hw, (u16*)out, f.n, C, H, W, nm);
      else frames_to_nhwc_kernel<u16, float, 8><<<g, 256, 0, st>>>((const float*)f.data, f.index, f.valid_hw, (u16*)out, f.n, C, H, W, nm);
    } else {
      if (f.dtype == TD_U8) frames_to_nhwc_kernel<float, uint8_t, 4><<<g, 256, 0, st>>>((const uint8_t*)f.data, f.index, f.valid_hw, (float*)out, f.n, C, H, W, nm);
      else frames_to_nhwc_kernel<float, float, 4><<<g, 256, 0, st>>>((const float*)f.data, f.index, f.valid_hw, (float*)out, f.n, C, H, W, nm);
    }
    done += f.n;
  }
  return check_launch("td_frames_to_nhwc");
}

extern "C" int td_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype,
                               td_stream_t stream) {
  TD_REQUIRE(x && y && Cpad >= C, "td_nchw_to_nhwc: bad arguments");
  if (C <= 4 && (Cpad == 4 || (Cpad == 8 && dtype == TD_BF16))) {
    td_frame_source f = {x, TD_F32, N, nullptr, nullptr};
    return td_frames_to_nhwc(&f, 1, C, H, W, Cpad, nullptr, nullptr, y, dtype, stream);
  }
  size_t n = (size_t)N * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) nchw_to_nhwc_kernel<u16><<<nblk(n), 256, 0, st>>>(x, (u16*)y, N, C, H, W, Cpad);
  else nchw_to_nhwc_kernel<float><<<nblk(n), 256, 0, st>>>(x, (float*)y, N, C, H, W, Cpad);
  return check_launch("td_nchw_to_nhwc");
}

extern "C" int td_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, td_stream_t stream) {
  TD_REQUIRE(x && y, "td_nhwc_to_nchw: null pointer");
  size_t n = (size_t)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TD_BF16) nhwc_to_nchw_kernel<u16><<<nblk(n), 256, 0, st>>>((const u16*)x, y, N, C, H, W);
  else nhwc_to_nchw_kernel<float><<<nblk(n), 256, 0, st>>>((const float*)x, y, N, C, H, W);
  return check_launch("td_nhwc_to_nchw");
}

extern "C" int td_cast(const void* x, void* y, size_t n, int src_dtype, int dst_dtype, td_stream_t stream) {
  TD_REQUIRE(x && y, "td_cast: null pointer");
  hipStream_t st = (hipStream_t)stream;
  unsigned g = nblk(n);
  if (g > 4096) g = 4096;
  if (g == 0) return TD_OK;
  if (src_dtype == TD_F32 && dst_dtype == TD_BF16) cast_kernel<float, u16><<<g, 256, 0, st>>>((const float*)x, (u16*)y, n);
  else if (src_dtype == TD_BF16 && dst_dtype == TD_F32) cast_kernel<u16, float><<<g, 256, 0, st>>>((const u16*)x, (float*)y, n);
  else if (src_dtype == TD_F32 && dst_dtype == TD_F32) cast_kernel<float, float><<<g, 256, 0, st>>>((const float*)x, (float*)y, n);
  else if (src_dtype == TD_BF16 && dst_dtype == TD_BF16) cast_kernel<u16, u16><<<g, 256, 0, st>>>((const u16*)x, (u16*)y, n);
  else TD_REQUIRE(false, "td_cast: bad dtype");
  return check_launch("td_cast");
}


// Pixel-pair stem weights (td_resnet_fwd, stem_pairs): prepared [Co][7][7][8] (3 real channels, K contiguous) ->
// [Co][7][4][8]: element (r, t, sub*4 + c) = w[r][s = 2t + sub - 1][c] for c < 3 and s >= 0, else 0.
template <typename T>
__global__ void stem_pair_weights_kernel(const T* __restrict__ w8, T* __restrict__ wp, int Co) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Co * 7 * 4 * 8) return;
  const int j = idx & 7, t = (idx >> 3) & 3, r = (idx >> 5) % 7, co = idx / (7 * 32);
  const int sub = j >> 2, c = j & 3, s_ = 2 * t + sub - 1;
  T v = T(0);
  if (s_ >= 0 && c < 3) v = w8[((co * 7 + r) * 7 + s_) * 8 + c];
  wp[idx] = v;
}

extern "C" int td_stem_pair_weights(const void* w_fwd_8, void* w_pairs, int Co, int dtype, td_stream_t stream) {
  TD_REQUIRE(w_fwd_8 && w_pairs && Co >= 1, "td_stem_pair_weights: bad arguments");
  TD_REQUIRE(dtype == TD_BF16, "td_stem_pair_weights: the pixel-pair stem is a bf16 layout");
  const int n = Co * 7 * 4 * 8;
  stem_pair_weights_kernel<u16><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const u16*)w_fwd_8, (u16*)w_pairs, Co);
  return check_launch("td_stem_pair_weights");
}
