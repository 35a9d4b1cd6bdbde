// Optimizer-side tail of a training step as three HBM-bound launches over FLAT fp32 buffers (all trainable
// parameters of the model back to back, gradients in the same order - the buffer the data-parallel exchange already
// produces): gradient-norm partial sums, norm finalize (clip coefficient + step counter), fused clip + AdamW + EMA.
// Replaces, per step, ~3 elementwise torch launches for each of the 923 state-dict entries:
//   torch.nn.utils.clip_grad_norm_ + optimizer.step()            engine.py:147-151
//   torch.optim.AdamW with three name-based parameter groups      main.py:381-413
//   update_ema(model, model_ema, decay)                           util/optim.py:8-25
// Learning rates (adjust_learning_rate, util/optim.py:28-95, stays host code: three scalars) and the step counter live in
// device memory, so the whole tail can be captured in a HIP graph behind the backward pass.
#include "td_common.h"

namespace td {

struct Segs {
  td_optim_segment s[TD_OPTIM_MAX_SEGMENTS];
  int n;
};

__device__ __forceinline__ int find_seg(const Segs& sg, long long i) {
  int k = 0;
#pragma unroll 4
  for (int j = 1; j < sg.n; ++j) k = (i >= sg.s[j].begin) ? j : k;
  return k;
}

// partial[block] = sum of g^2 over the block's grid-stride share (inactive segments = parameters without a gradient are
// skipped like clip_grad_norm_ skips p.grad is None)
__global__ __launch_bounds__(256) void grad_sq_partial_kernel(const float* __restrict__ g, long long n, Segs sg, double* partial) {
  const long long nv = n >> 2;
  double acc = 0.0;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long long)gridDim.x * 256) {
    const float4 x = ((const float4*)g)[v];
    const long long i0 = v << 2;
    const int k0 = find_seg(sg, i0), k1 = find_seg(sg, i0 + 3);
    if (k0 == k1) {
      if (sg.s[k0].active) acc += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
    } else {
      const float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (sg.s[find_seg(sg, i0 + q)].active) acc += (double)e[q] * e[q];
    }
  }
  if (blockIdx.x == 0)
    for (long long i = (nv << 2) + threadIdx.x; i < n; i += 256)
      if (sg.s[find_seg(sg, i)].active) acc += (double)g[i] * g[i];
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one workgroup: total norm, clip coefficient min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping), step += 1
__global__ __launch_bounds__(256) void grad_norm_finalize_kernel(const double* partial, int nblocks, float max_norm, float* norm_clip, int* step) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    norm_clip[0] = norm;
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(max_norm / (norm + 1e-6f), 1.f);
    norm_clip[1] = c;
    if (step) step[0] += 1;
  }
}

struct AdamArgs {
  float b1, b2, eps, wd, ema_decay;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float* ema, float lr, float clip, float bc1, float bc2s, const AdamArgs& a) {
  g *= clip;
  p *= 1.f - lr * a.wd;                 // decoupled weight decay
  m += (g - m) * (1.f - a.b1);          // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (1.f - a.b2) * g * g;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / bc2s + a.eps;
  p -= (lr / bc1) * (m / denom);
  if (ema) *ema = *ema * a.ema_decay + (1.f - a.ema_decay) * p;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, long long n, Segs sg,
                                                        const float* __restrict__ lr_dev, const float* __restrict__ norm_clip,
                                                        const int* __restrict__ step_dev, AdamArgs a) {
  const float clip = norm_clip ? norm_clip[1] : 1.f;
  const int step = step_dev[0];  // already incremented for this update
  const float bc1 = 1.f - powf(a.b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(a.b2, (float)step));
  float lrs[TD_OPTIM_MAX_GROUPS];
#pragma unroll
  for (int q = 0; q < TD_OPTIM_MAX_GROUPS; ++q) lrs[q] = lr_dev[q];
  const long long nv = n >> 2;
  for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nv; vi += (long long)gridDim.x * 256) {
    const long long i0 = vi << 2;
    const int k0 = find_seg(sg, i0), k1 = find_seg(sg, i0 + 3);
    if (k0 == k1 && !sg.s[k0].active) continue;
    float4 P = ((float4*)p)[vi], G = ((const float4*)g)[vi], M = ((float4*)m)[vi], V = ((float4*)v)[vi], E = make_float4(0, 0, 0, 0);
    if (ema) E = ((float4*)ema)[vi];
    float* pe = (float*)&P; float* ge = (float*)&G; float* me = (float*)&M; float* ve = (float*)&V; float* ee = (float*)&E;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = (k0 == k1) ? k0 : find_seg(sg, i0 + q);
      if (!sg.s[k].active) continue;
      adam_elem(pe[q], ge[q], me[q], ve[q], ema ? &ee[q] : nullptr, lrs[sg.s[k].group], clip, bc1, bc2s, a);
    }
    ((float4*)p)[vi] = P; ((float4*)m)[vi] = M; ((float4*)v)[vi] = V;
    if (ema) ((float4*)ema)[vi] = E;
  }
  if (blockIdx.x == 0)
    for (long long i = (nv << 2) + threadIdx.x; i < n; i += 256) {
      const int k = find_seg(sg, i);
      if (!sg.s[k].active) continue;
      adam_elem(p[i], g[i], m[i], v[i], ema ? &ema[i] : nullptr, lrs[sg.s[k].group], clip, bc1, bc2s, a);
    }
}

static int fill_segs(Segs& sg, const td_optim_segment* segs, int n_segs, size_t n, const char* who) {
  TD_REQUIRE(segs && n_segs >= 1 && n_segs <= TD_OPTIM_MAX_SEGMENTS, "%s: 1..%d segments expected", who, TD_OPTIM_MAX_SEGMENTS);
  long long at = 0;
  for (int i = 0; i < n_segs; ++i) {
    TD_REQUIRE(segs[i].begin == at && segs[i].end > segs[i].begin, "%s: segments must tile [0, n) in order", who);
    TD_REQUIRE(segs[i].group >= 0 && segs[i].group < TD_OPTIM_MAX_GROUPS, "%s: group out of range", who);
    sg.s[i] = segs[i];
    at = segs[i].end;
  }
  TD_REQUIRE((size_t)at == n, "%s: segments cover %lld of %zu elements", who, at, n);
  sg.n = n_segs;
  return TD_OK;
}

static unsigned opt_grid(size_t n) {
  size_t b = (n / 4 + 255) / 256;
  if (b < 1) b = 1;
  if (b > TD_OPTIM_NORM_BLOCKS) b = TD_OPTIM_NORM_BLOCKS;
  return (unsigned)b;
}

}  // namespace td
using namespace td;

extern "C" size_t td_grad_norm_ws_bytes(void) { return (size_t)TD_OPTIM_NORM_BLOCKS * sizeof(double); }

extern "C" int td_grad_norm_clip(const float* grad, size_t n, const td_optim_segment* segs, int n_segs, float max_norm, void* ws,
                                 size_t ws_bytes, float* norm_clip, int* step, td_stream_t stream) {
  TD_REQUIRE(grad && ws && norm_clip && n > 0, "td_grad_norm_clip: null pointer / empty buffer");
  TD_REQUIRE(ws_bytes >= td_grad_norm_ws_bytes(), "td_grad_norm_clip: workspace smaller than td_grad_norm_ws_bytes()");
  TD_REQUIRE(((uintptr_t)grad & 15) == 0, "td_grad_norm_clip: the flat buffer must be 16-byte aligned");
  Segs sg;
  int rc = fill_segs(sg, segs, n_segs, n, "td_grad_norm_clip");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = opt_grid(n);
  grad_sq_partial_kernel<<<g, 256, 0, st>>>(grad, (long long)n, sg, (double*)ws);
  grad_norm_finalize_kernel<<<1, 256, 0, st>>>((const double*)ws, (int)g, max_norm, norm_clip, step);
  return check_launch("td_grad_norm_clip");
}

extern "C" int td_adamw_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n,
                                 const td_optim_segment* segs, int n_segs, const float* lr_dev, const float* norm_clip,
                                 const int* step_dev, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                                 td_stream_t stream) {
  TD_REQUIRE(param && grad && exp_avg && exp_avg_sq && lr_dev && step_dev && n > 0, "td_adamw_ema_step: null pointer / empty buffer");
  TD_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema) & 15) == 0,
             "td_adamw_ema_step: flat buffers must be 16-byte aligned");
  Segs sg;
  int rc = fill_segs(sg, segs, n_segs, n, "td_adamw_ema_step");
  if (rc) return rc;
  AdamArgs a = {beta1, beta2, eps, weight_decay, ema_decay};
  size_t b = (n / 4 + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  adamw_ema_kernel<<<(unsigned)b, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, ema, (long long)n, sg, lr_dev, norm_clip,
                                                               step_dev, a);
  return check_launch("td_adamw_ema_step");
}
