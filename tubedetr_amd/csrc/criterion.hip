// SetCriterion (models/tubedetr.py:270-372, 397-460) for the main output and the five auxiliary decoder layers as ONE
// launch: the 24 losses (L1 + GIoU on the annotated frames' boxes, start/end KL divergence against Gaussian targets,
// guided-attention loss on the temporal self-attention weights) and, in the same pass, the derivative of every loss
// with respect to its inputs; a second tiny launch scales those by the upstream gradient of the 24 scalars.  The
// keep-index gather of engine.py:83-97 is an index read here.  Replaces ~250 torch micro-kernels per step.
// All tensors fp32; one workgroup per decoder layer.
#include "td_common.h"

namespace td {

struct CritParams {
  const float* boxes;       // [nl][bt][4] predicted boxes (cxcywh, after sigmoid) of every frame
  const float* tgt;         // [n][4] target boxes of the annotated frames
  const long long* keep;    // [n] frame index (into bt) of every annotated frame
  const float* sted;        // [nl][b][T][2] start / end logits, or null
  const float* weights;     // [nl][b][T][T] temporal self-attention weights, or null
  const uint8_t* time_mask; // [b][T] 1 = frame exists
  const uint8_t* positive;  // [b][T] 1 = frame inside the annotated interval
  const int* inter;         // [b][2] annotated (start, end) frame of every video
  const float* num_boxes_dev;  // device scalar, or null -> num_boxes_host
  float num_boxes_host;
  float inv_two_sigma_sq;
  int nl, b, T, bt, n;
  float* losses;   // [nl][4]: bbox, giou, sted, guided_attn
  float* g_l1;     // [nl][bt][4]  d loss_bbox / d boxes
  float* g_giou;   // [nl][bt][4]  d loss_giou / d boxes
  float* g_sted;   // [nl][b][T][2]
  float* g_w;      // [nl][b][T][T]
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void criterion_kernel(CritParams p) {
  __shared__ float red[4];
  const int l = blockIdx.x, t = threadIdx.x;
  const float eps = 1e-6f;
  float nb = p.num_boxes_dev ? p.num_boxes_dev[0] : p.num_boxes_host;
  nb = fmaxf(nb, 1.f);
  // ---- boxes: L1 + GIoU over the annotated frames (tubedetr.py:270-291, util/box_ops.py) ----
  float* gl1 = p.g_l1 + (size_t)l * p.bt * 4;
  float* ggi = p.g_giou + (size_t)l * p.bt * 4;
  for (int i = t; i < p.bt * 4; i += 256) gl1[i] = ggi[i] = 0.f;
  __syncthreads();
  float s_l1 = 0.f, s_gi = 0.f;
  for (int i = t; i < p.n; i += 256) {
    const long long f = p.keep[i];
    const float4 s = *(const float4*)(p.boxes + ((size_t)l * p.bt + f) * 4);
    const float4 g = *(const float4*)(p.tgt + (size_t)i * 4);
    const float sv[4] = {s.x, s.y, s.z, s.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float d1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = sv[c] - gv[c];
      s_l1 += fabsf(d);
      d1[c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / nb;
    }
    const float ax1 = s.x - 0.5f * s.z, ay1 = s.y - 0.5f * s.w, ax2 = s.x + 0.5f * s.z, ay2 = s.y + 0.5f * s.w;
    const float bx1 = g.x - 0.5f * g.z, by1 = g.y - 0.5f * g.w, bx2 = g.x + 0.5f * g.z, by2 = g.y + 0.5f * g.w;
    const float area_a = (ax2 - ax1) * (ay2 - ay1), area_b = (bx2 - bx1) * (by2 - by1);
    const float iw = fminf(ax2, bx2) - fmaxf(ax1, bx1), ih = fminf(ay2, by2) - fmaxf(ay1, by1);
    const float iwc = fmaxf(iw, 0.f), ihc = fmaxf(ih, 0.f);
    const float inter = iwc * ihc;
    const float uni = area_a + area_b - inter;
    const float ew = fmaxf(fmaxf(ax2, bx2) - fminf(ax1, bx1), 0.f), eh = fmaxf(fmaxf(ay2, by2) - fminf(ay1, by1), 0.f);
    const float hull = ew * eh;
    const float giou = inter / uni - (hull - uni) / hull;
    s_gi += 1.f - giou;
    // reverse mode through giou = inter/uni - 1 + uni/hull; upstream of giou is -1/nb
    const float up = -1.f / nb;
    float g_inter = up / uni;
    const float g_uni = up * (-inter / (uni * uni) + 1.f / hull);
    const float g_hull = up * (-uni / (hull * hull));
    const float g_area = g_uni;
    g_inter -= g_uni;
    const float g_iw = iw >= 0.f ? g_inter * ihc : 0.f, g_ih = ih >= 0.f ? g_inter * iwc : 0.f;
    const float g_ew = g_hull * eh, g_eh = g_hull * ew;
    // min / max pick one argument (ties split evenly, torch.minimum / maximum)
    auto lt = [](float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); };
    float gx1 = -g_iw * lt(bx1, ax1) - g_ew * lt(ax1, bx1) - g_area * (ay2 - ay1);
    float gx2 = g_iw * lt(ax2, bx2) + g_ew * lt(bx2, ax2) + g_area * (ay2 - ay1);
    float gy1 = -g_ih * lt(by1, ay1) - g_eh * lt(ay1, by1) - g_area * (ax2 - ax1);
    float gy2 = g_ih * lt(ay2, by2) + g_eh * lt(by2, ay2) + g_area * (ax2 - ax1);
    float* o1 = gl1 + f * 4;
    float* o2 = ggi + f * 4;
    o1[0] = d1[0]; o1[1] = d1[1]; o1[2] = d1[2]; o1[3] = d1[3];
    o2[0] = gx1 + gx2; o2[1] = gy1 + gy2; o2[2] = 0.5f * (gx2 - gx1); o2[3] = 0.5f * (gy2 - gy1);
  }
  s_l1 = block_sum(s_l1, red);
  s_gi = block_sum(s_gi, red);
  // ---- start / end: KL(softmax over time || Gaussian target), tubedetr.py:293-351 ----
  float s_sted = 0.f;
  if (p.sted) {
    const float inv_bt = 1.f / (float)(p.b * p.T);
    for (int v = 0; v < p.b; ++v) {
      const uint8_t* tm = p.time_mask + (size_t)v * p.T;
      for (int c = 0; c < 2; ++c) {
        const float* z = p.sted + (((size_t)l * p.b + v) * p.T) * 2 + c;
        float* gz = p.g_sted + (((size_t)l * p.b + v) * p.T) * 2 + c;
        const int t0 = p.inter[v * 2 + c];
        // softmax over the valid frames (padded logits are filled with -1e32 in the reference: probability exactly 0)
        float mx = -INFINITY;
        for (int j = t; j < p.T; j += 256) mx = fmaxf(mx, tm[j] ? z[(size_t)j * 2] : -1e32f);
        mx = block_max(mx, red);
        float se = 0.f, sg = 0.f;
        for (int j = t; j < p.T; j += 256) {
          se += __expf((tm[j] ? z[(size_t)j * 2] : -1e32f) - mx);
          const float dt = (float)(j - t0);
          sg += __expf(-dt * dt * p.inv_two_sigma_sq) + eps;
        }
        se = block_sum(se, red);
        sg = block_sum(sg, red);
        sg = fmaxf(sg, 1e-12f);  // F.normalize(p=1) divides by max(norm, 1e-12)
        // loss_c = mean_{b,T} tm * pr * log((pr + eps) / gt);  dL/dpr = tm * (log(..) + pr / (pr + eps)) / (bT)
        float lsum = 0.f, dot = 0.f;
        for (int j = t; j < p.T; j += 256) {
          const float pr = __expf((tm[j] ? z[(size_t)j * 2] : -1e32f) - mx) / se;
          const float dt = (float)(j - t0);
          const float gt = (__expf(-dt * dt * p.inv_two_sigma_sq) + eps) / sg;
          const float lg = __logf((pr + eps) / gt);
          const float m = tm[j] ? 1.f : 0.f;
          lsum += m * pr * lg;
          dot += pr * (m * (lg + pr / (pr + eps)) * inv_bt);
        }
        lsum = block_sum(lsum, red);
        dot = block_sum(dot, red);
        s_sted += lsum * inv_bt;
        for (int j = t; j < p.T; j += 256) {
          const float pr = __expf((tm[j] ? z[(size_t)j * 2] : -1e32f) - mx) / se;
          const float dt = (float)(j - t0);
          const float gt = (__expf(-dt * dt * p.inv_two_sigma_sq) + eps) / sg;
          const float m = tm[j] ? 1.f : 0.f;
          const float dpr = m * (__logf((pr + eps) / gt) + pr / (pr + eps)) * inv_bt;
          gz[(size_t)j * 2] = tm[j] ? pr * (dpr - dot) : 0.f;  // masked_fill: no gradient into the padded logits
        }
      }
    }
  }
  // ---- guided attention (tubedetr.py:353-372): -log(1 - w + eps) on the rows outside the annotated interval ----
  float s_ga = 0.f;
  if (p.weights) {
    for (int v = 0; v < p.b; ++v) {
      const uint8_t* tm = p.time_mask + (size_t)v * p.T;
      const uint8_t* pm = p.positive + (size_t)v * p.T;
      float cnt = 0.f;
      for (int j = t; j < p.T; j += 256) cnt += (pm[j] || !tm[j]) ? 0.f : 1.f;
      cnt = block_sum(cnt, red) + eps;
      const float sc = 1.f / (cnt * (float)p.b);
      const float* w = p.weights + ((size_t)l * p.b + v) * p.T * p.T;
      float* gw = p.g_w + ((size_t)l * p.b + v) * p.T * p.T;
      float acc = 0.f;
      for (int idx = t; idx < p.T * p.T; idx += 256) {
        const int i = idx / p.T;
        const bool excl = pm[i] || !tm[i];
        const float om = 1.f - w[idx] + eps;
        acc += excl ? 0.f : -__logf(om);
        gw[idx] = excl ? 0.f : sc / om;
      }
      s_ga += block_sum(acc, red) * sc;
    }
  }
  if (t == 0) {
    float* o = p.losses + l * 4;
    o[0] = s_l1 / nb;
    o[1] = s_gi / nb;
    o[2] = s_sted;
    o[3] = s_ga;
  }
}

// d_in = sum over the losses an input feeds of (upstream gradient of that loss) * (stored derivative)
__global__ void criterion_bwd_kernel(const float* dl /*[nl][4]*/, const float* g_l1, const float* g_giou, const float* g_sted, const float* g_w,
                                     float* d_boxes, float* d_sted, float* d_w, long long n_box, long long n_sted, long long n_w, int nl) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_box) {
    const int l = (int)(i / (n_box / nl));
    d_boxes[i] = dl[l * 4 + 0] * g_l1[i] + dl[l * 4 + 1] * g_giou[i];
  } else if (i < n_box + n_sted) {
    const long long j = i - n_box;
    d_sted[j] = dl[(int)(j / (n_sted / nl)) * 4 + 2] * g_sted[j];
  } else if (i < n_box + n_sted + n_w) {
    const long long j = i - n_box - n_sted;
    d_w[j] = dl[(int)(j / (n_w / nl)) * 4 + 3] * g_w[j];
  }
}

}  // namespace td
using namespace td;

extern "C" int td_criterion_fwd(const float* boxes, const float* tgt, const long long* keep, const float* sted, const float* weights,
                                const uint8_t* time_mask, const uint8_t* positive, const int* inter, const float* num_boxes_dev,
                                float num_boxes_host, float sigma, int nl, int b, int T, int n, float* losses, float* g_l1, float* g_giou,
                                float* g_sted, float* g_w, td_stream_t stream) {
  TD_REQUIRE(boxes && tgt && keep && losses && g_l1 && g_giou, "td_criterion_fwd: null pointer");
  TD_REQUIRE(nl >= 1 && b >= 1 && T >= 1 && n >= 0, "td_criterion_fwd: bad sizes");
  TD_REQUIRE(!sted || (time_mask && inter && g_sted), "td_criterion_fwd: pred_sted needs time_mask, inter and g_sted");
  TD_REQUIRE(!weights || (time_mask && positive && g_w), "td_criterion_fwd: weights need time_mask, positive and g_w");
  TD_REQUIRE(sigma > 0.f, "td_criterion_fwd: sigma must be positive");
  CritParams p;
  memset(&p, 0, sizeof(p));
  p.boxes = boxes; p.tgt = tgt; p.keep = keep; p.sted = sted; p.weights = weights; p.time_mask = time_mask; p.positive = positive;
  p.inter = inter; p.num_boxes_dev = num_boxes_dev; p.num_boxes_host = num_boxes_host; p.inv_two_sigma_sq = 1.f / (2.f * sigma * sigma);
  p.nl = nl; p.b = b; p.T = T; p.bt = b * T; p.n = n;
  p.losses = losses; p.g_l1 = g_l1; p.g_giou = g_giou; p.g_sted = g_sted; p.g_w = g_w;
  criterion_kernel<<<nl, 256, 0, (hipStream_t)stream>>>(p);
  return check_launch("td_criterion_fwd");
}

extern "C" int td_criterion_bwd(const float* dlosses, const float* g_l1, const float* g_giou, const float* g_sted, const float* g_w,
                                float* d_boxes, float* d_sted, float* d_weights, int nl, int b, int T, td_stream_t stream) {
  TD_REQUIRE(dlosses && g_l1 && g_giou && d_boxes, "td_criterion_bwd: null pointer");
  const long long n_box = (long long)nl * b * T * 4, n_sted = (g_sted && d_sted) ? (long long)nl * b * T * 2 : 0,
                  n_w = (g_w && d_weights) ? (long long)nl * b * T * T : 0;
  const long long tot = n_box + n_sted + n_w;
  criterion_bwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(dlosses, g_l1, g_giou, g_sted, g_w, d_boxes, d_sted, d_weights,
                                                                                      n_box, n_sted, n_w, nl);
  return check_launch("td_criterion_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Evaluation path: PostProcessSTVG (models/postprocessors.py:13-84) - the most likely (start, end) pair
// with end > start under the product of the start / end softmax distributions.  One workgroup per video; same arithmetic
// order as the reference (log_softmax, then (log p_start[s] + log p_end[e]) maximised over s < e, first index on ties),
// so the returned indices are bit-identical to torch's.
namespace td {

__global__ __launch_bounds__(256) void sted_decode_kernel(const float* __restrict__ steds, long long* __restrict__ out, int T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ls = smem;          // log p_start
  float* le = smem + T;      // log p_end
  float* best = smem + 2 * T;  // per end index: best score
  int* arg = (int*)(smem + 3 * T);
  __shared__ float red[4];
  const int v = blockIdx.x, t = threadIdx.x;
  const float* z = steds + (size_t)v * T * 2;
  for (int c = 0; c < 2; ++c) {
    float mx = -INFINITY;
    for (int j = t; j < T; j += 256) mx = fmaxf(mx, z[(size_t)j * 2 + c]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int j = t; j < T; j += 256) se += expf(z[(size_t)j * 2 + c] - mx);
    se = block_sum(se, red);
    const float lse = logf(se);
    float* dst = c == 0 ? ls : le;
    for (int j = t; j < T; j += 256) dst[j] = (z[(size_t)j * 2 + c] - mx) - lse;  // log_softmax: (x - max) - log(sum exp(x - max))
  }
  __syncthreads();
  for (int e = t; e < T; e += 256) {
    float bv = -INFINITY;
    int bs = 0;
    const float lee = le[e];
    for (int s = 0; s < e; ++s) {
      const float val = ls[s] + lee;
      if (val > bv) { bv = val; bs = s; }
    }
    best[e] = bv;
    arg[e] = bs;
  }
  __syncthreads();
  if (t == 0) {
    float bv = -INFINITY;
    int be = 0;
    for (int e = 0; e < T; ++e)
      if (best[e] > bv) { bv = best[e]; be = e; }
    out[v * 2 + 0] = arg[be];
    out[v * 2 + 1] = be;
  }
}

}  // namespace td

extern "C" int td_sted_decode(const float* steds, long long* start_end, int n_videos, int T, td_stream_t stream) {
  TD_REQUIRE(steds && start_end && n_videos >= 1 && T >= 1, "td_sted_decode: bad arguments");
  const size_t lds = (size_t)4 * T * sizeof(float);
  TD_REQUIRE(lds <= 60 * 1024, "td_sted_decode: T=%d too long (max 3840 frames per video)", T);
  td::sted_decode_kernel<<<n_videos, 256, lds, (hipStream_t)stream>>>(steds, start_end, T);
  return td::check_launch("td_sted_decode");
}
