/* C ABI of libtubedetr_hip.so - the MI355X (gfx950) kernels behind the TubeDETR hot path.
 *
 * The reference (antoyang/TubeDETR) is pure Python: its hot path has no FFI of its own, it calls
 * torch / torchvision operators from nn.Modules.  Each entry below therefore names the reference
 * call site (file:line under the reference tree) whose arithmetic it replaces; the Python modules
 * in tubedetr_amd/models keep the reference's nn.Module surface and bind these symbols with ctypes
 * (tubedetr_amd/_hip.py; the stub a reference maintainer would add is in INTEGRATION.md).
 *
 * Conventions: raw device pointers borrowed for the call (never retained, never freed); caller
 * allocates outputs and workspaces; no hidden allocation, no synchronisation, every launch goes to
 * the `stream` argument (a hipStream_t passed as void*); re-entrant, callable from any thread
 * (autograd workers).  Return 0 or a negative TD_ERR_* code; td_last_error() gives a thread-local
 * message.  dtype: TD_F32 (exact fp32 MFMA, parity mode) or TD_BF16 (bf16 MFMA, fp32 accumulate).
 * Activations are NHWC / row-major [rows][channels]; "T" below = the dtype's element type.
 */
#ifndef TUBEDETR_HIP_H
#define TUBEDETR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TD_F32 0
#define TD_BF16 1
#define TD_U8 2 /* input pixels only (td_frame_source) */
#define TD_OK 0
#define TD_ERR_INVALID (-1)
#define TD_ERR_LAUNCH (-2)

typedef void* td_stream_t; /* hipStream_t */

const char* td_last_error(void);
int td_abi_version(void);

/* Deterministic parity mode: 1 = every reduction that is normally split over workgroups and combined with fp32 atomics (weight gradients,
 * LayerNorm dgamma / dbeta, bias column sums) runs as one sequential reduction per output element: two runs of a step are bit-identical.
 * Process-wide atomic flag, seeded once from TD_DETERMINISTIC=1 in the environment, read at LAUNCH time: grid and split choices are frozen
 * into a captured HIP graph, so a graph captured under the other setting must be re-captured. */
int td_set_deterministic(int on);
int td_get_deterministic(void);

/* Dropout RNG: counter-based, keep(seed, element index) = hash32(element index * golden + seed) >= p * 2^32.  Every
 * dropout-capable entry point takes an optional `dropout_counter`: a DEVICE pointer to one uint32 step counter.  When
 * non-NULL the kernels re-key the seed with a hash of (seed, *dropout_counter), read at execution time - a captured HIP
 * graph then draws fresh, statistically independent masks on each replay (the caller increments the counter between
 * replays); forward and backward of one step must see the same counter value.  NULL = the seed is used as passed.
 * There is no process-global dropout state. */

/* Optional per-kernel-family timing with HIP events recorded on the launch stream around every MFMA kernel
 * launch (bench.py's roofline leg).  td_prof_enable(1) starts collecting (records are kept under a mutex; off by
 * default and then one relaxed atomic load per launch), td_prof_collect synchronises the recorded events and returns, per family (TD_PROF_*), the number of
 * launches, the summed duration in ms and the summed ALGORITHMIC flops (2*M*N*K of the un-padded problem). */
#define TD_PROF_GEMM_128x128 0
#define TD_PROF_GEMM_128x64 1
#define TD_PROF_WGRAD 2
#define TD_PROF_GEMM_64x128 3
#define TD_PROF_PW_RESIDENT 4 /* persistent weight-stationary pointwise instance */
#define TD_PROF_GEMM_256 5    /* 256-row tiles, eight wavefronts (conv_gemm_big_kernel), spatial (3x3) layers: the MFMA-bound members */
#define TD_PROF_GEMM_256_PW 6 /* the same kernel on pointwise layers with K >= 512: HBM-bound (4.2 TB/s of algorithmic bytes) */
#define TD_PROF_FUSED 7       /* LDS-resident chains: fused stem (stem.hip), fused frozen bottlenecks (bottleneck.hip) */
#define TD_PROF_CROSS_Q1 8    /* time-aligned cross-attention frame core (cross_attn.hip): HBM-bound by the memory rows */
#define TD_PROF_WGRAD_SINGLE 9 /* one weight gradient per launch (td_conv_wgrad[_bias]); TD_PROF_WGRAD = the batched launches */
#define TD_PROF_CHAIN 10 /* chained conv3 -> next-block conv1 pairs of layer3 (chain.hip): HBM-bound, the block output never read back */
#define TD_PROF_FAMILIES 11
int td_prof_enable(int on);
int td_prof_collect(int family, int dtype, long long* launches, double* ms, double* flops);
/* Sum of the ALGORITHMIC HBM bytes of the same launches (each operand / result tensor counted once per launch). */
int td_prof_collect_bytes(int family, int dtype, double* bytes);
/* CSV (family,dtype,M,N,K,R,stride,mode|splits,ms) of every recorded launch since td_prof_enable(1). */
int td_prof_dump(const char* path);
/* Debug: when non-NULL, td_conv_gemm workgroups launched by THIS thread write 6 cycle stamps each into
 * buf[workgroup*8 + i] (thread-local setting). */
int td_debug_set_stamp_buffer(unsigned long long* buf);

/* Geometry of one implicit-GEMM convolution / linear layer.  rows m enumerate (n, ho, wo);
 * k enumerates (r, s, c) with c fastest; the gathered source is NHWC [N][Hs][Ws][C].
 * mode 0 (forward):  hs = ho*stride - pad + r
 * mode 1 (dgrad):    hs = (ho + pad - r) / stride when divisible (source = grad of conv output)
 * A linear layer is R=S=1, stride=1, pad=0, Hs=Ho, Ws=Wo.                                        */
typedef struct td_conv_desc {
  int N, Hs, Ws, C;
  int Ho, Wo;
  int R, S, stride, pad;
  int mode;
  int Nc;     /* output channels = rows of the weight matrix [Nc][R*S*C] */
  int ldc;    /* leading dimension (elements) of the output / residual / mask rows */
  int out_sp; /* 1 = dense rows; >1 = row (n,ho,wo) is written at (n, ho*out_sp, wo*out_sp) of an */
  int out_H, out_W; /*        [N][out_H][out_W][ldc] tensor (strided 1x1 dgrad scatter)           */
  int aniso;            /* 1: `stride` / `pad` describe the vertical direction only, the horizontal one uses the two fields  */
  int stride_w, pad_w;  /*    below (forward geometry only).  The pixel-pair form of the stem: a 7x7 stride-2 convolution of */
                        /*    3-channel pixels = a 7x4 convolution, stride (2,1), pad (3,2), of 8-channel pixel PAIRS        */
} td_conv_desc;

/* Fused epilogue: v = acc (+ bias[n]) (+ residual[m][n]); relu; sigmoid; mask (v if mask_src>0 else 0);
 * dropout (inverted, counter-based hash RNG on element index). */
typedef struct td_epilogue {
  const float* bias;
  const void* residual;
  const void* mask_src;
  int relu;
  int sigmoid;
  float dropout_p;
  uint32_t dropout_seed;
  float alpha; /* scales acc before bias; 0 is treated as 1 */
  const uint32_t* dropout_counter; /* optional device step counter (see "Dropout RNG" above) */
} td_epilogue;

/* out[m][n] = epilogue(sum_k gather(src)[m][k] * wmat[n][k]).
 * Replaces: torch Conv2d/Linear forward + FrozenBatchNorm2d + ReLU + residual
 * (models/backbone.py:60-70,97-98; torchvision resnet Bottleneck; models/transformer.py:643,748,
 * 769; models/tubedetr.py:39,80) and, in mode 1 / with transposed weights, their input gradients. */
int td_conv_gemm(const void* src, const void* wmat, void* out, const td_conv_desc* d, const td_epilogue* e,
                 int dtype, td_stream_t stream);

/* Linear layer whose input rows are formed on the fly:
 *   out[out_map[m]][n] = epilogue( sum_k X[m][k] * wmat[n][k] (+ residual[res_map[m]][n]) ),   X[m] = [ A1[a1_map[m]] | A2[a2_map[m]] ]
 * (K1 columns from A1, K2 from A2; every map is a device int32[M] or NULL = identity; res_map NULL = the output row).
 * With a shared weight (w_shared) the product is (A1 + A2) W^T: "src + pos" enters the Q / K projections as a second operand stream
 * instead of a materialised sum (with_pos_embed + in_proj of nn.MultiheadAttention, models/transformer.py:637-640,
 * 735-737), and the temporal replication of the clip memory to its frames (models/transformer.py:393-427) becomes a row
 * index inside its consumers: the slow-fast aggregation (:441-445) reads the clip rows through a1_map / res_map and writes
 * the frame rows through out_map.  a2 may be NULL (one source).  K1 must be a multiple of the K tile (64 bf16 / 32 fp32
 * elements) when a2 is given; lda / ldc / ldr are row strides in elements (ldr <= 0: ldc). */
typedef struct td_linear_ex_desc {
  int M, N, K1, K2;
  int lda1, lda2, ldc, ldr;
  int w_shared;           /* 1: wmat is [N][K1] and multiplies both sources (K2 == K1): out = (A1 + A2) W^T, no [W | W] copy; 0: wmat is [N][K1 + K2] */
  long long rows1, rows2; /* rows held by the A1 / A2 buffers (bounds of the gather) */
  const int* a1_map;
  const int* a2_map;
  const int* out_map;
  const int* res_map;
} td_linear_ex_desc;
int td_linear_ex(const void* a1, const void* a2, const void* wmat, void* out, const td_linear_ex_desc* x, const td_epilogue* e,
                 int dtype, td_stream_t stream);

/* dw[n][k] += sum_m g[m][n] * gather(src)[m][k]   (fp32 atomics, `splits` partitions of m).
 * Replaces: autograd weight gradients of Conv2d / Linear on the same call sites. */
int td_conv_wgrad(const void* g, const void* src, float* dw, const td_conv_desc* d, int ldg, int dtype, int splits,
                  td_stream_t stream);
/* Same, and dbias[n] += sum_m g[m][n] (the bias gradient of a Linear / Conv2d with bias, models/transformer.py:643,748,769)
 * from the gradient fragments the kernel already holds; dbias (fp32, zero-initialised by the caller) may be NULL. */
int td_conv_wgrad_bias(const void* g, const void* src, float* dw, float* dbias, const td_conv_desc* d, int ldg, int dtype,
                       int splits, td_stream_t stream);

/* Batched weight gradients: ONE launch (two when 1x1 and 3x3 jobs are mixed) computes the weight gradients of many
 * conv layers.  dW of every job is written in the parameter's own [Nc][ci_real][R][S] fp32 layout with `scale[co]`
 * (the FrozenBN factor, may be NULL) folded in - overwritten, not accumulated.  With thousands of output tiles in flight
 * a job needs no reduction splits (no atomics, no accumulator memset, no finalize pass) unless its M is very long.
 * A linear layer is a job with R=S=1 (dW = the parameter's [out][in]); its bias gradient rides along (`dbias`).  The
 * transformer's ~94 weight gradients of one step are deferred to the end of backward and run as ONE such launch.
 * `jobs` is a host array, consumed before the call returns.  The per-launch job table lives in caller-provided
 * memory of td_conv_wgrad_batch_table_bytes(n_jobs) bytes each: `table_host` (page-locked host memory, written by this
 * call) and `table_dev` (device memory, filled by ONE hipMemcpyAsync on `stream`).  Both must stay untouched until the
 * stream has passed this call - for a captured stream, for the lifetime of the graph (the copy node re-reads
 * table_host at every replay).  No allocation, no synchronisation inside.
 * Replaces the same autograd call sites as td_conv_wgrad,
 * for all convs of the trunk at once (torchvision resnet Bottleneck backward, models/backbone.py:94-98). */
typedef struct td_wgrad_job {
  const void* g;      /* [M][ldg] output gradient rows */
  const void* src;    /* the layer's input activation (NHWC) */
  float* dW;          /* [Nc][ci_real][R][S] fp32 */
  const float* scale; /* [Nc] or NULL */
  td_conv_desc d;     /* forward geometry (mode 0) */
  int ldg;
  int ci_real;        /* input channels of the parameter (d.C may be padded) */
  float* dbias;       /* optional [Nc] fp32: column sums of g (bias gradient of a Linear / Conv2d with bias), overwritten */
  int accumulate;     /* 1: dW += (instead of =), run behind the overwriting jobs of the same call: the second operand stream of a
                       * two-source layer (td_linear_ex: dW = g^T A1 + g^T A2, the positional operand of the Q / K projections); no dbias */
  int prezeroed;      /* 1: the caller zero-filled dW (and dbias) already - e.g. ONE fill over a flat buffer that holds the dW of all
                       * jobs: the library then enqueues no per-job fill for jobs it splits along M (fp32 atomics into dW) */
} td_wgrad_job;
size_t td_conv_wgrad_batch_table_bytes(int n_jobs);
int td_conv_wgrad_batch(const td_wgrad_job* jobs, int n_jobs, int dtype, void* table_host, void* table_dev, size_t table_bytes,
                        td_stream_t stream);

/* Input frames of the trunk: one or more sources of NCHW frames - fp32 (the reference's normalised samples.tensors /
 * samples_fast.tensors, engine.py:55-57) or uint8 pixels (normalised here: (x/255 - mean[c]) * inv_std[c], the
 * datasets' T.Normalize done on the device, so the host sends a quarter of the bytes) - concatenated in order into one
 * NHWC T tensor with channels zero-padded to Cpad (8 or 4 for bf16, 4 for fp32).  `index` (device int32[n], may be NULL) picks the
 * source frame of every contributed frame: the slow clip is video[::k] of the SAME buffer as the fast frames
 * (datasets/vidstg.py:250-251) - no second copy of the pixels, no concatenation.  mean / inv_std: host arrays of C floats
 * or NULL (no normalisation). */
typedef struct td_frame_source {
  const void* data; /* [n_src][C][H][W] */
  int dtype;        /* TD_F32 or TD_U8 */
  int n;            /* frames contributed */
  const int* index; /* device int32[n] or NULL (= 0..n-1) */
  const int* valid_hw; /* device int32[n_src][2] = (rows, columns) of every SOURCE frame that hold pixels, or NULL (= H x W): a
                        * ragged batch padded to a common H x W; pixels outside are written as exactly 0 (after normalisation), like
                        * NestedTensor.from_tensor_list pads the normalised frames, util/misc.py:158-170 */
} td_frame_source;
int td_frames_to_nhwc(const td_frame_source* srcs, int n_srcs, int C, int H, int W, int Cpad, const float* mean,
                      const float* inv_std, void* y, int dtype, td_stream_t stream);

/* Native executor of the bottleneck-ResNet trunk (replaces the module-graph execution of torchvision resnet101 through
 * IntermediateLayerGetter, models/backbone.py:94-98, and its autograd backward).  Conv order in every array: stem,
 * then per block conv1, conv2, conv3[, downsample] (td_resnet_num_convs entries).  srcs: the N = sum of srcs[i].n input
 * frames (td_frames_to_nhwc: fp32 or uint8 + normalisation, optional frame index lists);
 * w_fwd/bias: prepared (FrozenBN-folded) weights of td_weight_prep; ws: caller-allocated workspace.  save=1 keeps every
 * activation for td_resnet_bwd; save=0 (the no_grad "fast" pass, models/tubedetr.py:128-129) runs in a 6-slot ring.
 * *feat points into ws: layer4 output NHWC [N][feat_hw[0]][feat_hw[1]][feat_hw[2]]. */
size_t td_resnet_fwd_ws_bytes(int N, int H, int W, const int* nblocks, int dtype, int save);
int td_resnet_num_convs(const int* nblocks);
int td_resnet_fwd(const td_frame_source* srcs, int n_srcs, const float* mean, const float* inv_std, int N, int H, int W,
                  const int* nblocks, const void* const* w_fwd, const float* const* bias, int save, void* ws, size_t ws_bytes,
                  void** feat, int* feat_hw, int stem_pairs, int first_train_stage, int dtype, td_stream_t stream);
/* first_train_stage (0..4, as for td_resnet_bwd): the stages below it are frozen - td_resnet_bwd never reads their inner
 * activations, so with save = 1 they may still run fused (td_bottleneck_fused); with save = 0 everything may. */
/* One whole frozen 64-plane bottleneck (layer1 of the torchvision trunk, models/backbone.py:82-98) in one pass:
 * out[N][H][W][256] = relu(conv3(relu(conv2_3x3(relu(conv1(x))))) + identity), identity = x (Cin = 256, wd = NULL) or the
 * block's downsample 1x1 (Cin = 64, wd / bd given); prepared (FrozenBN-folded, K-contiguous) bf16 weights: w1 [64][Cin],
 * w2 [64][3*3*64], w3 [256][64], wd [256][64].  The 64-channel tensors between the three convolutions stay in LDS.
 * Used by td_resnet_fwd; exported for the parity test. */
int td_bottleneck_fused(const void* x, void* out, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                        const float* b3, const void* wd, const float* bd, int N, int H, int W, int Cin, int dtype, td_stream_t stream);
/* Chained pointwise pair across two consecutive bottlenecks of one stage (bf16, planes = 256: layer3):
 *   out[M][4P] = relu(y2[M][P] w3[4P][P]^T + b3 + residual[M][4P])   conv3 + FrozenBN + identity + ReLU of block j
 *   h1[M][P]   = relu(out w1[P][4P]^T + b1)                           conv1 + FrozenBN + ReLU of block j + 1
 * in ONE launch: `out` is written (it is the next identity and, in a saved pass, the saved activation) but never read back.
 * Bit-identical to td_conv_gemm(conv3, residual, relu) followed by td_conv_gemm(conv1, relu).  Replaces the two calls
 * torchvision's Bottleneck.forward makes on either side of a block boundary (models/backbone.py:97-98) in both trunk passes
 * (models/tubedetr.py:127-134).  Used by td_resnet_fwd; exported for the parity test. */
int td_pw_chain2(const void* y2, const void* w3, const float* b3, const void* residual, void* out, const void* w1, const float* b1,
                 void* h1, int M, int planes, int dtype, td_stream_t stream);
/* stem_pairs = 1 (bf16, even W): the stem runs in its pixel-pair form - the frames are laid down with 4 channels per
 * pixel (3 + one zero), two horizontally adjacent pixels form one 8-channel element, and the 7x7 stride-2 convolution
 * becomes a 7x4 convolution with stride (2, 1) and padding (3, 2) over them: K = 224 instead of 392 (3 channels padded to 8),
 * half the input bytes.  w_fwd[0] must then be the [64][7][4][8] weight td_stem_pair_weights derives from the prepared
 * [64][7][7][8] one. */
int td_stem_pair_weights(const void* w_fwd_8, void* w_pairs, int Co, int dtype, td_stream_t stream);
/* The whole frozen stem in one pass (torchvision resnet conv1 + bn1 + relu + maxpool, reached through models/backbone.py:94-98):
 * pooled[N][PH][PW][64] = maxpool3x3s2p1( relu( conv7x7s2p3(x) + bias ) ), x_pairs = the frames as 4-channel pixels (two per 16-byte
 * element: [N][H][W/2][8], what td_frames_to_nhwc writes with Cpad = 4), w_pairs = td_stem_pair_weights' [64][7][4][8], bf16.  The
 * 64-channel convolution output stays in LDS (it is 4 GB per 1 000 frames of res 352 otherwise, written once and read 1.5 times).
 * td_resnet_fwd uses it when stem_pairs = 1; exported for the parity test. */
int td_stem_pool(const void* x_pairs, const void* w_pairs, const float* bias, void* pooled, int N, int H, int W, int dtype,
                 td_stream_t stream);
/* Backward through the stages >= first_train_stage (0..3; the reference trains layer2-4 = 1, backbone.py:82-89):
 * dfeat = gradient of *feat; fwd_ws = the save=1 workspace of the forward; dW[i] receives the gradient of conv i in
 * the parameter's own [Co][Ci][R][S] fp32 layout (FrozenBN scale un-folded); entries of frozen convs are ignored.
 * The forward may have run over N_fwd >= N frames (slow frames first, then the no_grad "fast" frames in the same
 * launch sequence); only the first N frames are back-propagated. */
size_t td_resnet_bwd_ws_bytes(int N, int H, int W, const int* nblocks, int first_train_stage, int dtype);
/* bytes of the weight-gradient job table (see td_conv_wgrad_batch): table_host = page-locked host memory, table_dev =
 * device memory, both caller-allocated and left untouched until the stream has passed the call. */
size_t td_resnet_bwd_table_bytes(const int* nblocks, int first_train_stage);
/* dW_prezeroed = 1: every dW[i] was zero-filled by the caller (one fill over the flat buffer they are views of): no per-job fills. */
int td_resnet_bwd(const void* dfeat, int N, int N_fwd, int H, int W, const int* nblocks, int first_train_stage,
                  const void* const* w_dgrad, const float* const* scale, float* const* dW, const void* fwd_ws, void* ws,
                  size_t ws_bytes, void* table_host, void* table_dev, size_t table_bytes, int dW_prezeroed, int dtype, int only_stage, td_stream_t stream);
/* only_stage = -1: the whole pass (one batched weight-gradient launch at its end).  only_stage = s: only the launches of stage s (3 = layer4 ..
 * first_train_stage), its weight gradients in a batched launch of their own; the caller issues s = 3, 2, .. on one stream with the same
 * workspace.  A data-parallel caller can then start the exchange of layer4's gradients while layer3 / layer2 are still computed
 * (DistributedDataParallel's bucketed reducer does that for the reference, main.py:372-376). */

/* Fold FrozenBatchNorm2d (models/backbone.py:60-70) into a conv: w_fwd[co][r][s][ci] = W[co][ci][r][s]*scale[co]
 * (ci zero-padded to Cpad), w_dgrad[ci][r][s][co] likewise (may be NULL), bias_out[co] = b - rm*scale,
 * scale_out[co] = w*rsqrt(rv+eps).  bn_* may be NULL (plain layer: scale 1, bias copied from `bias`). */
int td_weight_prep(const float* W, const float* bn_w, const float* bn_b, const float* bn_rm, const float* bn_rv,
                   const float* bias, int Co, int Ci, int R, int S, int Cpad, void* w_fwd, void* w_dgrad,
                   float* bias_out, float* scale_out, int dtype, td_stream_t stream);

/* Batched form of td_weight_prep: ONE launch prepares every layer of a model.  `items_dev` is a device copy of an
 * array of n td_prep_item (the caller uploads it once; it only holds pointers and sizes, which are stable across
 * optimizer steps).  blk0 = exclusive prefix sum of the items' workgroup counts ceil(Co_alloc/16)*ceil(Cpad/32);
 * total_blocks = their sum.  Co_alloc >= Co rows are written (rows >= Co are zero: padded output layers). */
typedef struct td_prep_item {
  const float* W;
  const float *bn_w, *bn_b, *bn_rm, *bn_rv, *bias;
  void *w_fwd, *w_dgrad;
  float *bias_out, *scale_out;
  int Co, Ci, RS, Cpad, Co_alloc, blk0;
} td_prep_item;
int td_weight_prep_batch(const td_prep_item* items_dev, int n, int total_blocks, int dtype, td_stream_t stream);

/* dW[co][ci][r][s] (+)= dw_k[co][r][s][ci] * scale[co]  (scale may be NULL). */
int td_wgrad_finalize(const float* dw_k, const float* scale, float* dW, int Co, int Ci, int R, int S, int Cpad,
                      int accumulate, td_stream_t stream);

/* NCHW fp32 frames -> NHWC T with channels zero-padded to Cpad (engine.py:55 samples.tensors). */
int td_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, td_stream_t stream);
/* NHWC T -> NCHW fp32 (features returned through the nn.Module boundary). */
int td_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, td_stream_t stream);
/* NCHW fp32 gradient -> NHWC T. */
int td_cast(const void* x, void* y, size_t n, int src_dtype, int dst_dtype, td_stream_t stream);

/* 3x3 stride-2 pad-1 max pooling, NHWC (torchvision resnet stem; forward only: the stem is frozen,
 * models/backbone.py:82-89). */
int td_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, td_stream_t stream);

/* y = LayerNorm(x + r) * gamma + beta over the last dim (cols); s_out (optional) receives x + r;
 * mean/rstd fp32 per row.  models/transformer.py:641-645,721-722,744-750,581,771. */
int td_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* y, void* s_out,
                         float* mean, float* rstd, int rows, int cols, float eps, int dtype, td_stream_t stream);
/* ds = LN backward wrt (x + r) (+ extra, an optional additional upstream gradient of s);
 * dgamma/dbeta accumulated with fp32 atomics. */
int td_add_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd, const float* gamma,
                         const void* extra, void* ds, float* dgamma, float* dbeta, int rows, int cols, int dtype,
                         td_stream_t stream);

/* out[n] += sum_m g[m][n]  (bias gradients). */
int td_colsum(const void* g, float* out, int rows, int cols, int ld, int dtype, td_stream_t stream);
/* y = a + b (b may be NULL -> copy); elementwise on T. */
int td_add(const void* a, const void* b, void* y, size_t n, int dtype, td_stream_t stream);

/* g = dy * scale where y > 0 else 0   (ReLU / ReLU+dropout backward from the saved output y). */
int td_relu_bwd(const void* dy, const void* y, void* g, size_t n, float scale, int dtype, td_stream_t stream);

/* GELU, exact erf form (the intermediate activation of HF RobertaModel, models/transformer.py:130-135,252-263):
 * y = 0.5 x (1 + erf(x / sqrt 2));  backward dx = dy * (Phi(x) + x phi(x)) from the saved pre-activation x. */
int td_gelu_fwd(const void* x, void* y, size_t n, int dtype, td_stream_t stream);
int td_gelu_bwd(const void* dy, const void* x, void* dx, size_t n, int dtype, td_stream_t stream);

/* y[i] = keep(seed, i) ? x[i] / (1-p) : 0 - the same counter-based mask as the fused epilogues
 * (element index i = row*ld + col), used standalone and to re-apply the mask in backward. */
int td_dropout(const void* x, void* y, size_t n, float p, uint32_t seed, const uint32_t* dropout_counter, int dtype,
               td_stream_t stream);

/* PositionEmbeddingSine (models/position_encoding.py:71-94, normalize=True, scale=2pi) from a
 * (N,h,w) uint8 pad mask -> pos [N][rows_per_image][2*npf] T  (token-major, the layout the encoder consumes).
 * rows_per_image >= h*w (<= 0: h*w): the rows behind the h*w tokens are written as zeros - the positional operand of the
 * text tokens the reference appends with torch.cat([pos_embed, zeros_like(text)]) (models/transformer.py:323-326). */
int td_pos_sine(const uint8_t* mask, void* pos, int N, int h, int w, int npf, float temperature, int rows_per_image, int dtype,
                td_stream_t stream);

/* Row gather / scatter: dst[dst_map[i]][0:cols] = src[src_map[i]][0:cols] (+ add[i][0:cols]) for i < n_rows (maps: device
 * int32[n_rows] or NULL = i).  The temporal replication of clip rows to frame rows (models/transformer.py:393-427: the text
 * rows of every frame's memory; the whole memory under --no_fast) as ONE pass of 16-byte accesses instead of a Python loop
 * of slice assignments; with `add` the same pass forms gathered + plain rows (the slow-fast mix, :441, re-formed in backward
 * instead of stored).  ld_* = row strides in elements, cols a multiple of 8 (bf16) / 4 (fp32). */
int td_rows_copy(const void* src, const int* src_map, const void* add, void* dst, const int* dst_map, int n_rows, int cols, int ld_src,
                 int ld_add, int ld_dst, int dtype, td_stream_t stream);
/* Segment sums over rows (the backward of that replication): out[out_map[r]][0:cols] = sum_{j in [ptr[r], ptr[r+1])} in[idx[j]][0:cols],
 * fp32 accumulation, r < n_out (ptr: device int32[n_out + 1], idx: device int32[ptr[n_out]], out_map: device int32[n_out] or NULL = r). */
int td_rows_segment_sum(const void* in, const int* idx, const int* ptr, void* out, const int* out_map, int n_out, int cols, int ld_in,
                        int ld_out, int dtype, td_stream_t stream);

/* Multi-head attention core on already projected q,k,v (nn.MultiheadAttention internals,
 * models/transformer.py:613,638-640 encoder; 661,713-719 temporal self-attention; 662,734-740
 * time-aligned cross-attention with Lq=1).  Batch-major rows: q [B][Lq] rows of stride ldq, head h in
 * columns h*hd..h*hd+hd-1 (hd = 32: TubeDETR's transformer, MFMA kernels in bf16; hd = 64: RoBERTa's 12 x 64 heads, fp32-math
 * kernels in both dtypes); k,v [B][Lk] likewise; key_pad [B][Lk] uint8 (1 = ignore) or NULL.
 * scores = scale * q.k; probs [B][H][Lq][Lk] fp32 = softmax (pre-dropout, saved for backward);
 * out [B][Lq] rows of stride ldo = dropout(probs) @ v; wavg [B][Lq][Lk] fp32 = head average of the
 * post-dropout probabilities (what nn.MultiheadAttention returns) or NULL. */
int td_mha_fwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, void* out, float* probs,
               float* wavg, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
               float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream);
/* Backward of td_mha_fwd.  dwavg [B][Lq][Lk] fp32 = gradient of the head-averaged weights
 * (guided-attention loss, models/tubedetr.py:357-369) or NULL; ds_ws: fp32 workspace [B][H][Lq][Lk].
 * dq, dk, dv are written with the row strides of q, k, v (ldq, ldk, ldv); dout has row stride ldo. */
int td_mha_bwd(const void* q, const void* k, const void* v, const void* dout, const float* probs, const float* dwavg,
               void* dq, void* dk, void* dv, float* ds_ws, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk,
               int ldv, int ldo, float scale, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter,
               int dtype, td_stream_t stream);

/* Time-aligned cross-attention of the decoder (models/transformer.py:725-745; nn.MultiheadAttention with ONE query per
 * frame, E = 256 channels in H = 8 heads) with both memory-side projections moved to the query side:
 *   score[h][s] = u[f][h] . (mem[f*S+s] + pos[f*S+s])      u[f][h] = scale * W_k,h^T q[f],  q.b_k is constant over s and cancels
 *   out[f]      = W_v,blk zext[f],  zext[f] = [ z[f][0..H) | sum_s pd[f][h][s] ],  z[f][h] = sum_s pd[f][h][s] mem[f*S+s]
 * (pd = probabilities after dropout; same mask index and the same `probs` / `wavg` outputs as td_mha_fwd with Lq = 1).  The key
 * and value projections of the memory rows, 93 % of the decoder's FLOPs in the reference formulation, do not exist; the two
 * remaining products with W_k / W_v are GEMMs over the F query rows against the block-structured weights below.
 * u [F][H*E] T, mem / pos [F*S][E] T (pos may be NULL), key_pad [F][S] uint8 or NULL, probs [F][H][S] fp32, wavg [F][S] fp32 or
 * NULL, zext [F][ldz] T with ldz >= H*E + H (a multiple of 8). */
int td_cross_q1_fwd(const void* u, const void* mem, const void* pos, const uint8_t* key_pad, float* probs, float* wavg, void* zext,
                    int F, int S, int H, int E, int ldz, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter,
                    int dtype, td_stream_t stream);
/* Backward: d_zext [F][ldz] T, dwavg [F][S] fp32 or NULL  ->  d_u [F][H*E] T and d_mem [F*S][E] FP32, overwritten when
 * accumulate == 0, else added to (the memory is shared by the six layers: the first layer to run writes, the others add);
 * d_mem == NULL: the memory needs no gradient, nothing is formed or stored.
 * pos receives no gradient here (sine encodings; learned ones use the projected-memory path). */
int td_cross_q1_bwd(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg,
                    void* d_u, float* d_mem, int accumulate, int F, int S, int H, int E, int ldz, float dropout_p, uint32_t dropout_seed,
                    const uint32_t* dropout_counter, int dtype, td_stream_t stream);
/* bf16 mode: the memory gradient of ALL layers in one pass instead of an fp32 read-modify-write of [F*S][E] per layer.
 * td_cross_q1_bwd_coef = td_cross_q1_bwd without d_mem: row f*S + s of `coef` (bf16, row stride coef_ld, a multiple of 32)
 * receives this layer's sixteen coefficients [ds[0..H) | pd[0..H)] at column coef_col (a multiple of 16; one block per layer).
 * td_cross_q1_dmem then forms  d_mem[f*S + s] = sum over layers l, heads h of  ds_l[h][s] u_l[f][h] + pd_l[h][s] d_z_l[f][h]
 * as one [S][coef_ld] x [coef_ld][E] product per frame on the matrix pipe and writes it in T (bf16): u[l] / d_zext[l] are the
 * tensors layer l handed to td_cross_q1_bwd_coef with coef_col = 16 l (a layer that did not run: both NULL, its columns of a
 * zero-filled `coef` are never written).  models/transformer.py:725-745 (backward of the six layers' shared memory). */
int td_cross_q1_bwd_coef(const void* u, const void* mem, const void* pos, const float* probs, const void* d_zext, const float* dwavg,
                         void* d_u, void* coef, int coef_ld, int coef_col, int F, int S, int H, int E, int ldz, float dropout_p,
                         uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream);
int td_cross_q1_dmem(const void* coef, int coef_ld, const void* const* u, const void* const* d_zext, int n_layers, void* d_mem, int F, int S,
                     int H, int E, int ldz, int dtype, td_stream_t stream);
/* Block-structured forms of one [E][E] slice of nn.MultiheadAttention.in_proj_weight (fp32, [out][in]; out channel j belongs to
 * head j / (E/H)) for those GEMMs: w_n [E][H*E + nb] holds W[j][:] * alpha in column block head(j) of row j (zeros elsewhere)
 * and, if bias != NULL (nb = H, else 0), bias[j] in column H*E + head(j); w_t [H*E + nb][E] is its transpose.  Both in `dtype`.
 *   keys:   u = q w_t^T (alpha = 1/sqrt(E/H)),  d_q = d_u w_n^T          values: out = zext w_n^T,  d_zext = d_out w_t^T */
int td_head_blocks_expand(const float* W, const float* bias, float alpha, void* w_n, void* w_t, int E, int H, int dtype, td_stream_t stream);
/* ...and back: from the dense fp32 gradient G [E][H*E + nb] of w_n, dW[j][c] = G[j][head(j)*E + c] * alpha and (db != NULL: nb = H)
 * db[j] = G[j][H*E + head(j)]. */
int td_head_blocks_extract(const float* G, float alpha, float* dW, float* db, int E, int H, td_stream_t stream);

/* Lean form of the same attention core for callers that do not need the weights (the per-frame visual-text encoder,
 * models/transformer.py:638-640: `weights` is dead there, SURVEY.md 8a'): nothing of size Lq x Lk is stored.  The forward
 * writes two softmax statistics per (batch, head, query) into `stats` (td_mha_lean_stats_bytes(B, H, Lq) bytes: 4 floats per
 * row); the backward recomputes the probabilities from q, k, key_pad and those statistics, takes sum_k P dP from dout . out
 * (`out` = the forward's output), and uses the third float of each row as scratch.  bf16, hd = 32, Lk <= 256, Lq <= 448,
 * 16-byte aligned rows; anything else returns TD_ERR_INVALID (use td_mha_fwd / td_mha_bwd).  Same dropout mask as td_mha_fwd. */
size_t td_mha_lean_stats_bytes(int B, int H, int Lq);
int td_mha_lean_fwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, void* out, float* stats, int B, int H,
                    int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale, float dropout_p,
                    uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype, td_stream_t stream);
int td_mha_lean_bwd(const void* q, const void* k, const void* v, const uint8_t* key_pad, const void* out, const void* dout,
                    float* stats, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int hd, int ldq, int ldk, int ldv,
                    int ldo, float scale, float dropout_p, uint32_t dropout_seed, const uint32_t* dropout_counter, int dtype,
                    td_stream_t stream);


/* ---- optimizer-side tail of a training step over FLAT fp32 buffers -------------------------------
 * All trainable parameters of the model lie back to back in one buffer (`param`), their gradients in the same order in
 * another (the flat buffer of the data-parallel exchange).  The buffers are tiled by at most TD_OPTIM_MAX_SEGMENTS
 * segments; a segment belongs to one parameter group (learning rate index: 0 = transformer + heads, 1 = backbone,
 * 2 = text encoder, main.py:381-405) and is `active` unless its parameters received no gradient (RoBERTa's pooler:
 * torch skips p.grad is None in clip_grad_norm_ and in optimizer.step()). */
#define TD_OPTIM_MAX_SEGMENTS 32
#define TD_OPTIM_MAX_GROUPS 4
#define TD_OPTIM_NORM_BLOCKS 2048
typedef struct td_optim_segment {
  long long begin, end; /* element range [begin, end) */
  int group;            /* index into lr_dev[] */
  int active;           /* 0: left untouched */
} td_optim_segment;
/* Replaces torch.nn.utils.clip_grad_norm_ (engine.py:149-150): norm_clip[0] = L2 norm of all active gradients,
 * norm_clip[1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0); *step (device int, may be NULL) += 1.
 * ws: td_grad_norm_ws_bytes() bytes of device scratch.  Two launches. */
size_t td_grad_norm_ws_bytes(void);
int td_grad_norm_clip(const float* grad, size_t n, const td_optim_segment* segs, int n_segs, float max_norm, void* ws,
                      size_t ws_bytes, float* norm_clip, int* step, td_stream_t stream);
/* Replaces optimizer.step() of torch.optim.AdamW with three parameter groups (main.py:406-413, engine.py:151) and
 * update_ema (util/optim.py:8-25) in ONE launch: g = grad * norm_clip[1] (norm_clip may be NULL); param *= 1 - lr*wd;
 * exp_avg = lerp(exp_avg, g, 1-beta1); exp_avg_sq = beta2*exp_avg_sq + (1-beta2) g^2; param -= lr/(1-beta1^t) *
 * exp_avg / (sqrt(exp_avg_sq)/sqrt(1-beta2^t) + eps); ema = ema*ema_decay + (1-ema_decay)*param (ema may be NULL).
 * lr_dev: TD_OPTIM_MAX_GROUPS device floats (written by the host's adjust_learning_rate, util/optim.py:28-95);
 * step_dev: device int holding t (already incremented by td_grad_norm_clip). */
int td_adamw_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n,
                      const td_optim_segment* segs, int n_segs, const float* lr_dev, const float* norm_clip,
                      const int* step_dev, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                      td_stream_t stream);

/* ---- SetCriterion as one launch ------------------------------------------------------------------
 * The 24 losses of models/tubedetr.py:270-372,397-460 (L1 + GIoU over the annotated frames' boxes, start / end KL
 * divergence against Gaussian targets, guided-attention loss) for the main output and the auxiliary decoder layers,
 * stacked on a leading layer axis, all fp32:
 *   boxes [nl][b*T][4] predicted cxcywh of EVERY frame; keep [n] = frame indices of the annotated frames (the
 *   keep-gather of engine.py:83-97), tgt [n][4] their target boxes; sted [nl][b][T][2] logits (or NULL);
 *   weights [nl][b][T][T] temporal self-attention weights (or NULL); time_mask / positive [b][T] uint8;
 *   inter [b][2] annotated (start, end); num_boxes: device scalar if non-NULL (kept by the data-parallel harness) else
 *   num_boxes_host; both are clamped to >= 1.
 * losses [nl][4] = (loss_bbox, loss_giou, loss_sted, loss_guided_attn) per layer.  The same launch stores the
 * derivative of every loss with respect to its inputs (g_l1, g_giou: [nl][b*T][4]; g_sted like sted; g_w like weights);
 * td_criterion_bwd combines them with the upstream gradient dlosses [nl][4] into d_boxes / d_sted / d_weights. */
int td_criterion_fwd(const float* boxes, const float* tgt, const long long* keep, const float* sted, const float* weights,
                     const uint8_t* time_mask, const uint8_t* positive, const int* inter, const float* num_boxes_dev,
                     float num_boxes_host, float sigma, int nl, int b, int T, int n, float* losses, float* g_l1, float* g_giou,
                     float* g_sted, float* g_w, td_stream_t stream);
int td_criterion_bwd(const float* dlosses, const float* g_l1, const float* g_giou, const float* g_sted, const float* g_w,
                     float* d_boxes, float* d_sted, float* d_weights, int nl, int b, int T, td_stream_t stream);

/* PostProcessSTVG (models/postprocessors.py:13-84): per video the (start, end) index pair with end > start that maximises
 * log_softmax(steds[:, :, 0])[start] + log_softmax(steds[:, :, 1])[end]; first index on ties, like torch.max.
 * steds [n_videos][T][2] fp32 logits (-inf on padded positions), start_end [n_videos][2] int64. */
int td_sted_decode(const float* steds, long long* start_end, int n_videos, int T, td_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
