// Native executor of the bottleneck-ResNet trunk (torchvision resnet50/101 body with FrozenBatchNorm folded in):
// one C call walks the static layer plan and enqueues every kernel of a forward or backward pass on the given
// stream - no per-layer host round trips through Python (104 convs forward, ~210 GEMM launches backward).
// Replaces the module-graph execution of models/backbone.py:97-98 (IntermediateLayerGetter(resnet101) forward) and
// its autograd backward.  Conv order everywhere: stem, then per block conv1, conv2, conv3[, downsample].
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "td_common.h"

namespace td {

struct ConvSpec {
  int cin, cout, k, stride, pad;
};
struct Tens {
  size_t off;  // byte offset in the workspace
  int H, W, C;
};
struct BlockPlan {
  int conv[4];  // indices into the conv list: conv1, conv2, conv3, downsample(-1)
  int stride, stage;
  Tens in, h1, h2, idt, out;
};
struct Plan {
  std::vector<ConvSpec> convs;
  std::vector<BlockPlan> blocks;
  Tens x, stem, pool;
  size_t total, slot;
  int es;
};

static inline int co(int h, int k, int s, int p) { return (h + 2 * p - k) / s + 1; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Forward activation layout.  save=1: every tensor has its own region (kept for backward).  save=0: a ring of 6
// slots of the largest activation (a block keeps at most {in, h1, h2, idt, out} alive).
static Plan make_plan(int N, int H, int W, const int* nb, int dtype, int save) {
  Plan P;
  P.es = dtype == TD_BF16 ? 2 : 4;
  const int cpad = dtype == TD_BF16 ? 8 : 4;
  std::vector<Tens*> order;
  P.convs.push_back({cpad, 64, 7, 2, 3});
  P.x = {0, H, W, cpad};
  int h = co(H, 7, 2, 3), w = co(W, 7, 2, 3);
  P.stem = {0, h, w, 64};
  h = co(h, 3, 2, 1);
  w = co(w, 3, 2, 1);
  P.pool = {0, h, w, 64};
  int inpl = 64;
  Tens cur = P.pool;
  for (int s = 0; s < 4; ++s) {
    const int planes = 64 << s;
    for (int j = 0; j < nb[s]; ++j) {
      BlockPlan b;
      b.stride = (j == 0 && s > 0) ? 2 : 1;
      b.stage = s;
      b.conv[0] = (int)P.convs.size();
      P.convs.push_back({inpl, planes, 1, 1, 0});
      b.conv[1] = (int)P.convs.size();
      P.convs.push_back({planes, planes, 3, b.stride, 1});
      b.conv[2] = (int)P.convs.size();
      P.convs.push_back({planes, planes * 4, 1, 1, 0});
      b.conv[3] = -1;
      if (j == 0) {
        b.conv[3] = (int)P.convs.size();
        P.convs.push_back({inpl, planes * 4, 1, b.stride, 0});
      }
      const int ho = co(cur.H, 3, b.stride, 1), wo = co(cur.W, 3, b.stride, 1);
      b.in = cur;
      b.h1 = {0, cur.H, cur.W, planes};
      b.h2 = {0, ho, wo, planes};
      b.idt = {0, ho, wo, planes * 4};
      b.out = {0, ho, wo, planes * 4};
      cur = b.out;
      inpl = planes * 4;
      P.blocks.push_back(b);
    }
  }
  auto bytes = [&](const Tens& t) { return align256((size_t)N * t.H * t.W * t.C * P.es); };
  // assign offsets in execution order
  size_t maxb = 0;
  std::vector<Tens*> seq = {&P.x, &P.stem, &P.pool};
  for (auto& b : P.blocks) {
    seq.push_back(&b.h1);
    seq.push_back(&b.h2);
    if (b.conv[3] >= 0) seq.push_back(&b.idt);
    seq.push_back(&b.out);
  }
  for (auto* t : seq) maxb = std::max(maxb, bytes(*t));
  P.slot = maxb;
  size_t off = 0;
  int ring = 0;
  for (auto* t : seq) {
    if (save) {
      t->off = off;
      off += bytes(*t);
    } else {
      t->off = (size_t)(ring % 6) * maxb;
      ++ring;
    }
  }
  P.total = save ? off : 6 * maxb;
  // block inputs alias the previous output
  Tens prev = P.pool;
  for (auto& b : P.blocks) {
    b.in = prev;
    prev = b.out;
  }
  return P;
}

// Frames one launch may cover: every operand is addressed through a 32-bit buffer descriptor (< 2^31 elements and < 4 GiB per
// tensor).  A launch whose largest operand would leave that range over N frames is issued over equal frame groups instead -
// frames are the leading dimension of every activation, so a group is the same launch on rebased pointers.  (Only the no-grad
// pass uses this: at res 352 its layer1 tensors pass 4 GiB beyond 1 083 bf16 frames.)  TD_TRUNK_MAX_FRAMES=n (tests): pretend
// the range ends at n frames of the largest activation (16 * H * W elements per frame).
struct FrameGroups {
  double lim_elems;
  int N;
  FrameGroups(int N_, int H, int W, int es) : N(N_) {
    lim_elems = std::min(2147483647.0, 4294963200.0 / es);
    if (const char* e = getenv("TD_TRUNK_MAX_FRAMES")) lim_elems = std::min(lim_elems, atof(e) * 16.0 * H * W);
  }
  // frames per group for a launch whose largest operand has `per_frame` elements per frame
  int step(size_t per_frame) const {
    const long long nmax = std::max(1LL, (long long)(lim_elems / (double)per_frame));
    if (N <= nmax) return N;
    const long long groups = (N + nmax - 1) / nmax;
    return (int)((N + groups - 1) / groups);
  }
};
static inline size_t frame_elems(const Tens& t) { return (size_t)t.H * t.W * t.C; }

static int run_conv(const char* ws, const Tens& in, const Tens& out, int N, const ConvSpec& c, const void* w, const float* bias,
                    const char* residual, int relu, int dtype, td_stream_t st, const FrameGroups* fg = nullptr) {
  const int es = dtype == TD_BF16 ? 2 : 4;
  const int step = fg ? fg->step(std::max(frame_elems(in), frame_elems(out))) : N;
  for (int f0 = 0; f0 < N; f0 += step) {
    const int n = std::min(step, N - f0);
    td_conv_desc d = {n, in.H, in.W, in.C, out.H, out.W, c.k, c.k, c.stride, c.pad, 0, c.cout, c.cout, 1, 0, 0};
    td_epilogue e;
    memset(&e, 0, sizeof(e));
    e.bias = bias;
    e.residual = residual ? residual + (size_t)f0 * frame_elems(out) * es : nullptr;  // (a residual has the output's shape)
    e.relu = relu;
    int rc = td_conv_gemm(ws + in.off + (size_t)f0 * frame_elems(in) * es, w, (void*)(ws + out.off + (size_t)f0 * frame_elems(out) * es), &d, &e, dtype, st);
    if (rc) return rc;
  }
  return TD_OK;
}

}  // namespace td
using namespace td;

extern "C" size_t td_resnet_fwd_ws_bytes(int N, int H, int W, const int* nblocks, int dtype, int save) {
  return make_plan(N, H, W, nblocks, dtype, save).total;
}

extern "C" int td_resnet_num_convs(const int* nblocks) {
  int n = 1;
  for (int s = 0; s < 4; ++s) n += 3 * nblocks[s] + (nblocks[s] > 0 ? 1 : 0);
  return n;
}

extern "C" int td_resnet_fwd(const td_frame_source* srcs, int n_srcs, const float* mean, const float* inv_std, int N, int H, int W,
                             const int* nblocks, const void* const* w_fwd, const float* const* bias, int save, void* ws,
                             size_t ws_bytes, void** feat, int* feat_hw, int stem_pairs, int first_train_stage, int dtype, td_stream_t stream) {
  TD_REQUIRE(srcs && n_srcs >= 1 && nblocks && w_fwd && bias && ws && feat, "td_resnet_fwd: null pointer");
  TD_REQUIRE(dtype == TD_F32 || dtype == TD_BF16, "td_resnet_fwd: bad dtype");
  {
    long long tot = 0;
    for (int i = 0; i < n_srcs; ++i) tot += srcs[i].n;
    TD_REQUIRE(tot == N, "td_resnet_fwd: the sources contribute %lld frames, N = %d", tot, N);
  }
  Plan P = make_plan(N, H, W, nblocks, dtype, save);
  TD_REQUIRE(ws_bytes >= P.total, "td_resnet_fwd: workspace too small (%zu < %zu)", ws_bytes, P.total);
  char* base = (char*)ws;
  int rc;
  // a pass that keeps its activations for backward is walked by td_resnet_bwd as ONE workspace of whole tensors: it must fit the
  // 32-bit range as it is (td_conv_gemm reports it if not); the no-grad pass splits oversized launches into frame groups
  const FrameGroups groups(N, H, W, P.es);
  const FrameGroups* fg = save ? nullptr : &groups;
  bool stem_done = false;  // the fused stem wrote the pooled tensor already
  if (stem_pairs) {
    // pixel-pair stem (see tubedetr_hip.h): 4-channel pixels in the first half of x's region = [H][W/2] elements of 8 channels
    TD_REQUIRE(dtype == TD_BF16 && (W & 1) == 0, "td_resnet_fwd: the pixel-pair stem needs bf16 and an even frame width");
    rc = td_frames_to_nhwc(srcs, n_srcs, 3, H, W, 4, mean, inv_std, base + P.x.off, dtype, stream);
    if (rc) return rc;
    // conv + bias + ReLU + max-pool in one pass (stem.hip): the 64-channel stem output never exists in HBM.  TD_STEM_FUSED=0:
    // the pixel-pair convolution and the pooling kernel as two launches (A/B).
    static const int fused = [] { const char* e_ = getenv("TD_STEM_FUSED"); return e_ ? atoi(e_) : 1; }();
    if (fused) {
      rc = td_stem_pool(base + P.x.off, w_fwd[0], bias[0], base + P.pool.off, N, H, W, dtype, stream);
      stem_done = true;
    } else {
      const int step = fg ? fg->step(frame_elems(P.stem)) : N;
      for (int f0 = 0; f0 < N && !rc; f0 += step) {
        td_conv_desc d = {std::min(step, N - f0), H, W / 2, 8, P.stem.H, P.stem.W, 7, 4, 2, 3, 0, 64, 64, 1, 0, 0, 1, 1, 2};
        td_epilogue e;
        memset(&e, 0, sizeof(e));
        e.bias = bias[0];
        e.relu = 1;
        rc = td_conv_gemm(base + P.x.off + (size_t)f0 * H * W * 4 * P.es, w_fwd[0], base + P.stem.off + (size_t)f0 * frame_elems(P.stem) * P.es, &d, &e, dtype, stream);
      }
    }
  } else {
    rc = td_frames_to_nhwc(srcs, n_srcs, 3, H, W, P.x.C, mean, inv_std, base + P.x.off, dtype, stream);
    if (rc) return rc;
    rc = run_conv(base, P.x, P.stem, N, P.convs[0], w_fwd[0], bias[0], nullptr, 1, dtype, stream, fg);
  }
  if (rc) return rc;
  if (!stem_done) {
    const int step = fg ? fg->step(frame_elems(P.stem)) : N;
    for (int f0 = 0; f0 < N; f0 += step)
      if ((rc = td_maxpool3x3s2(base + P.stem.off + (size_t)f0 * frame_elems(P.stem) * P.es, base + P.pool.off + (size_t)f0 * frame_elems(P.pool) * P.es,
                                std::min(step, N - f0), P.stem.H, P.stem.W, 64, dtype, stream)))
        return rc;
  }
  // frozen 64-plane bottlenecks (layer1: stages below first_train_stage are never back-propagated, so none of their inner
  // tensors is needed again) run as ONE launch each (bottleneck.hip).  Measured at 1 000 frames of 88 x 88 (tools/fused_l1_time.py):
  // block 0 (64 input channels + downsample) 2.09 ms fused vs 4.24 ms layer by layer; the 256-channel blocks 3.29 ms (resident-tile
  // variant) vs 3.72 ms.  TD_L1_FUSED: 0 = never (exact-fp32 mode and A/B: layer by layer), 1 = block 0 only, 2 (default) = every
  // frozen 64-plane block.
  static const int l1_fused = [] { const char* e = getenv("TD_L1_FUSED"); return e ? atoi(e) : 2; }();
  static const int chain_on = [] { const char* e = getenv("TD_CHAIN"); return e ? atoi(e) : 1; }();
  bool h1_done = false;  // this block's conv1 output was written by the previous block's chained launch
  for (size_t bi = 0; bi < P.blocks.size(); ++bi) {
    auto& b = P.blocks[bi];
    const int c1 = b.conv[0], c2 = b.conv[1], c3 = b.conv[2], cd = b.conv[3];
    if (l1_fused && (l1_fused >= 2 || cd >= 0) && dtype == TD_BF16 && (!save || b.stage < first_train_stage) && b.stride == 1 && P.convs[c1].cout == 64 &&
        P.convs[c3].cout == 256 && ((P.convs[c1].cin == 64 && cd >= 0) || (P.convs[c1].cin == 256 && cd < 0)) &&
        (fg || (double)N * b.out.H * b.out.W * 256 < 2147483647.0)) {
      const int step = fg ? fg->step(frame_elems(b.out)) : N;
      for (int f0 = 0; f0 < N; f0 += step)
        if ((rc = td_bottleneck_fused(base + b.in.off + (size_t)f0 * frame_elems(b.in) * P.es, base + b.out.off + (size_t)f0 * frame_elems(b.out) * P.es, w_fwd[c1],
                                      bias[c1], w_fwd[c2], bias[c2], w_fwd[c3], bias[c3], cd >= 0 ? w_fwd[cd] : nullptr, cd >= 0 ? bias[cd] : nullptr,
                                      std::min(step, N - f0), b.in.H, b.in.W, P.convs[c1].cin, dtype, stream)))
          return rc;
      continue;
    }
    if (!h1_done && (rc = run_conv(base, b.in, b.h1, N, P.convs[c1], w_fwd[c1], bias[c1], nullptr, 1, dtype, stream, fg))) return rc;
    h1_done = false;
    if ((rc = run_conv(base, b.h1, b.h2, N, P.convs[c2], w_fwd[c2], bias[c2], nullptr, 1, dtype, stream, fg))) return rc;
    const char* idt = base + b.in.off;
    if (cd >= 0) {
      if ((rc = run_conv(base, b.in, b.idt, N, P.convs[cd], w_fwd[cd], bias[cd], nullptr, 0, dtype, stream, fg))) return rc;
      idt = base + b.idt.off;
    }
    // conv3 + identity + ReLU of this block CHAINED with conv1 + ReLU of the next one (chain.hip: the block output is written - it is the
    // next identity and, in a saved pass, a saved activation - but not read back from HBM).  Where: 256-plane blocks (layer3) with a
    // successor in the same stage, bf16; both tensors of the pair are written exactly where the two separate launches write them, so
    // td_resnet_bwd walks the same workspace.  TD_CHAIN=0: the two launches (A/B; bit-identical results).
    if (chain_on && dtype == TD_BF16 && bi + 1 < P.blocks.size() && P.blocks[bi + 1].stage == b.stage && P.convs[c3].cin == 256 && P.convs[c3].cout == 1024 &&
        P.blocks[bi + 1].stride == 1 && (double)N * b.out.H * b.out.W * 1024.0 * 2.0 < 4294967000.0) {
      const BlockPlan& nb_ = P.blocks[bi + 1];
      const int n1 = nb_.conv[0];
      if ((rc = td_pw_chain2(base + b.h2.off, w_fwd[c3], bias[c3], idt, base + b.out.off, w_fwd[n1], bias[n1], base + nb_.h1.off, N * b.out.H * b.out.W, 256, dtype, stream)))
        return rc;
      h1_done = true;
      continue;
    }
    if ((rc = run_conv(base, b.h2, b.out, N, P.convs[c3], w_fwd[c3], bias[c3], idt, 1, dtype, stream, fg))) return rc;
  }
  const Tens& last = P.blocks.empty() ? P.pool : P.blocks.back().out;
  *feat = base + last.off;
  if (feat_hw) {
    feat_hw[0] = last.H;
    feat_hw[1] = last.W;
    feat_hw[2] = last.C;
  }
  return TD_OK;
}

// ---- backward ----
// ws layout: [ dw_k accumulators (fp32, all trainable convs) | ring of 6 gradient-activation slots ]
static size_t dwk_bytes(const Plan& P, int first_stage, std::vector<size_t>* offs) {
  size_t off = 0;
  if (offs) offs->assign(P.convs.size(), (size_t)-1);
  for (auto& b : P.blocks) {
    if (b.stage < first_stage) continue;
    for (int q = 0; q < 4; ++q) {
      int ci = b.conv[q];
      if (ci < 0) continue;
      const ConvSpec& c = P.convs[ci];
      if (offs) (*offs)[ci] = off;
      off += align256((size_t)c.cout * c.k * c.k * c.cin * sizeof(float));
    }
  }
  return off;
}
static size_t grad_slot_bytes(const Plan& P, int N, int first_stage) {
  size_t m = 0;
  for (auto& b : P.blocks) {
    if (b.stage < first_stage) continue;
    for (const Tens* t : {&b.in, &b.h1, &b.h2, &b.out}) m = std::max(m, align256((size_t)N * t->H * t->W * t->C * P.es));
  }
  return m;
}

// batched weight gradients (default): every gradient activation stays alive until the single td_conv_wgrad_batch launch
// at the end of the pass, so they are bump-allocated instead of cycling through the ring
static bool wgrad_batched() {
  static const bool on = [] { const char* e = getenv("TD_WGRAD_BATCH"); return !(e && e[0] == '0'); }();
  return on;
}
static size_t all_grad_bytes(const Plan& P, int N, int first_stage) {
  size_t tot = 0;
  bool top = true;
  for (int bi = (int)P.blocks.size() - 1; bi >= 0; --bi) {
    const BlockPlan& b = P.blocks[bi];
    if (b.stage < first_stage) break;
    auto sz = [&](const Tens& t) { return align256((size_t)N * t.H * t.W * t.C * P.es); };
    if (top) tot += sz(b.out);
    top = false;
    tot += sz(b.h2) + sz(b.h1) + sz(b.in);
  }
  return tot;
}

extern "C" size_t td_resnet_bwd_ws_bytes(int N, int H, int W, const int* nblocks, int first_train_stage, int dtype) {
  Plan P = make_plan(N, H, W, nblocks, dtype, 1);  // (only tensor shapes matter here, not offsets)
  if (wgrad_batched()) return all_grad_bytes(P, N, first_train_stage);
  return dwk_bytes(P, first_train_stage, nullptr) + 6 * grad_slot_bytes(P, N, first_train_stage);
}

// jobs of one stage: three convs per bottleneck + the stage's downsample
static int stage_jobs(const int* nblocks, int st) { return nblocks[st] > 0 ? 3 * nblocks[st] + 1 : 0; }
// The table is laid out per stage (stage 3 first): a pass that is issued stage by stage (only_stage >= 0) launches one batch per call, and a
// batch's staging memory must stay untouched until the stream has passed its launch - every stage owns its part.
static size_t stage_table_offset(const int* nblocks, int first_train_stage, int st) {
  size_t off = 0;
  for (int s = 3; s > st; --s)
    if (s >= first_train_stage) off += td_conv_wgrad_batch_table_bytes(stage_jobs(nblocks, s));
  return off;
}
extern "C" size_t td_resnet_bwd_table_bytes(const int* nblocks, int first_train_stage) {
  int n = 0;
  for (int st = 0; st < 4; ++st)
    if (st >= first_train_stage) n += stage_jobs(nblocks, st);
  // (one batch over all jobs, or one per stage: the larger of the two layouts)
  return std::max(td_conv_wgrad_batch_table_bytes(n), stage_table_offset(nblocks, first_train_stage, first_train_stage - 1));
}

extern "C" int td_resnet_bwd(const void* dfeat, int N, int N_fwd, int H, int W, const int* nblocks, int first_train_stage,
                             const void* const* w_dgrad, const float* const* scale, float* const* dW, const void* fwd_ws,
                             void* ws, size_t ws_bytes, void* table_host, void* table_dev, size_t table_bytes, int dW_prezeroed, int dtype,
                             int only_stage, td_stream_t stream) {
  TD_REQUIRE(dfeat && nblocks && w_dgrad && scale && dW && fwd_ws && ws, "td_resnet_bwd: null pointer");
  TD_REQUIRE(N >= 1 && N <= N_fwd, "td_resnet_bwd: N=%d must be in 1..N_fwd=%d", N, N_fwd);
  // only_stage = -1: the whole pass.  only_stage = s (first_train_stage..3): ONLY the launches of stage s - the walk below is the same
  // (the gradient activations are bump-allocated in walk order, so the gradient a stage receives sits where the previous call left it);
  // calls must come in the order 3, 2, .., first_train_stage on one stream with the same workspace.  That is how the trunk's weight
  // gradients leave in pieces (one batched launch per stage) for a data-parallel exchange that overlaps the remaining backward.
  TD_REQUIRE(only_stage == -1 || (only_stage >= first_train_stage && only_stage <= 3), "td_resnet_bwd: only_stage=%d out of range", only_stage);
  TD_REQUIRE(only_stage == -1 || wgrad_batched(), "td_resnet_bwd: a stage-by-stage pass needs the batched weight gradients");
  // activation offsets are those of the forward pass over N_fwd frames; the first N frames of every tensor (a
  // contiguous prefix, frames are the leading dimension) are the ones that carry gradient
  Plan P = make_plan(N_fwd, H, W, nblocks, dtype, 1);
  std::vector<size_t> dwoff;
  const size_t dwb = dwk_bytes(P, first_train_stage, &dwoff);
  const size_t slot = grad_slot_bytes(P, N, first_train_stage);
  const bool batched = wgrad_batched();
  TD_REQUIRE(ws_bytes >= (batched ? all_grad_bytes(P, N, first_train_stage) : dwb + 6 * slot), "td_resnet_bwd: workspace too small");
  const char* acts = (const char*)fwd_ws;
  char* base = (char*)ws;
  char* ring = batched ? base : base + dwb;
  size_t bump = 0;
  std::vector<td_wgrad_job> jobs;
  hipStream_t st = (hipStream_t)stream;
  int rix = 0;
  auto galloc = [&](const Tens& t) {
    if (batched) {
      char* q = base + bump;
      bump += align256((size_t)N * t.H * t.W * t.C * P.es);
      return q;
    }
    const int s_ = rix++ % 6;
    return ring + (size_t)s_ * slot;
  };
  if (!batched && hipMemsetAsync(base, 0, dwb, st) != hipSuccess) {
    set_error("td_resnet_bwd: memset failed");
    return TD_ERR_LAUNCH;
  }
  int rc;
  bool live = true;  // the block being walked belongs to the requested stage
  auto wgrad = [&](const void* g, const Tens& gt, const Tens& xin, int ci) -> int {
    if (!live) return TD_OK;
    const ConvSpec& c = P.convs[ci];
    td_conv_desc d = {N, xin.H, xin.W, xin.C, gt.H, gt.W, c.k, c.k, c.stride, c.pad, 0, c.cout, c.cout, 1, 0, 0};
    if (batched) {
      td_wgrad_job j;
      j.g = g;
      j.src = acts + xin.off;
      j.dW = dW[ci];
      j.scale = scale[ci];
      j.d = d;
      j.ldg = c.cout;
      j.ci_real = c.cin;
      j.dbias = nullptr;
      j.accumulate = 0;
      j.prezeroed = dW_prezeroed;
      jobs.push_back(j);
      return TD_OK;
    }
    float* dwk = (float*)(base + dwoff[ci]);
    int r = td_conv_wgrad(g, acts + xin.off, dwk, &d, c.cout, dtype, 0, stream);
    if (r) return r;
    return td_wgrad_finalize(dwk, scale[ci], dW[ci], c.cout, c.cin, c.k, c.k, c.cin, 0, stream);
  };
  auto dgrad = [&](const void* g, const Tens& gt, const Tens& xin, int ci, const void* residual, const void* mask, void* out) -> int {
    if (!live) return TD_OK;
    const ConvSpec& c = P.convs[ci];
    td_conv_desc d = {N, gt.H, gt.W, gt.C, xin.H, xin.W, c.k, c.k, c.stride, c.pad, 1, c.cin, c.cin, 1, 0, 0};
    td_epilogue e;
    memset(&e, 0, sizeof(e));
    e.residual = residual;
    e.mask_src = mask;
    return td_conv_gemm(g, w_dgrad[ci], out, &d, &e, dtype, stream);
  };
  int last = (int)P.blocks.size() - 1;
  int first = 0;
  while (first <= last && P.blocks[first].stage < first_train_stage) ++first;
  if (first > last) return TD_OK;
  const Tens& fo = P.blocks[last].out;
  char* g_out = galloc(fo);
  if ((only_stage == -1 || only_stage == P.blocks[last].stage) &&
      (rc = td_relu_bwd(dfeat, acts + fo.off, g_out, (size_t)N * fo.H * fo.W * fo.C, 1.f, dtype, stream)))
    return rc;
  for (int bi = last; bi >= first; --bi) {
    const BlockPlan& b = P.blocks[bi];
    live = only_stage == -1 || b.stage == only_stage;
    const int c1 = b.conv[0], c2 = b.conv[1], c3 = b.conv[2], cd = b.conv[3];
    if ((rc = wgrad(g_out, b.out, b.h2, c3))) return rc;
    char* g_h2 = galloc(b.h2);
    if ((rc = dgrad(g_out, b.out, b.h2, c3, nullptr, acts + b.h2.off, g_h2))) return rc;
    if ((rc = wgrad(g_h2, b.h2, b.h1, c2))) return rc;
    char* g_h1 = galloc(b.h1);
    if ((rc = dgrad(g_h2, b.h2, b.h1, c2, nullptr, acts + b.h1.off, g_h1))) return rc;
    if ((rc = wgrad(g_h1, b.h1, b.in, c1))) return rc;
    if (cd >= 0 && (rc = wgrad(g_out, b.out, b.in, cd))) return rc;
    if (bi == first) break;  // the first trainable block's input comes from frozen layers
    char* dx = galloc(b.in);
    const void* xin = acts + b.in.off;
    if (cd >= 0) {
      if ((rc = dgrad(g_h1, b.h1, b.in, c1, nullptr, xin, dx))) return rc;
      const ConvSpec& c = P.convs[cd];
      if (c.stride == 1) {
        if ((rc = dgrad(g_out, b.out, b.in, cd, dx, xin, dx))) return rc;
      } else {  // strided 1x1: scatter-accumulate into the positions the stride touches
        td_conv_desc d = {N, b.out.H, b.out.W, b.out.C, b.out.H, b.out.W, 1, 1, 1, 0, 0, c.cin, c.cin, c.stride, b.in.H, b.in.W};
        td_epilogue e;
        memset(&e, 0, sizeof(e));
        e.residual = dx;
        e.mask_src = xin;
        if (live && (rc = td_conv_gemm(g_out, w_dgrad[cd], dx, &d, &e, dtype, stream))) return rc;
      }
    } else {
      if ((rc = dgrad(g_h1, b.h1, b.in, c1, g_out, xin, dx))) return rc;
    }
    g_out = dx;
  }
  if (batched && !jobs.empty()) {
    const size_t toff = only_stage == -1 ? 0 : stage_table_offset(nblocks, first_train_stage, only_stage);
    TD_REQUIRE(table_bytes >= toff + td_conv_wgrad_batch_table_bytes((int)jobs.size()), "td_resnet_bwd: job-table workspace too small");
    return td_conv_wgrad_batch(jobs.data(), (int)jobs.size(), dtype, (char*)table_host + toff, (char*)table_dev + toff, table_bytes - toff, stream);
  }
  return TD_OK;
}
