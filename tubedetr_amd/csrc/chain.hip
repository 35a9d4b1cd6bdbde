// Chained pointwise pair across two consecutive bottlenecks of one stage (bf16, gfx950):
//
//   out[m][:] = relu(y2[m][:] W3^T + b3 + res[m][:])        conv3 + FrozenBN + identity + ReLU of block j     (P -> 4P channels)
//   h1[m][:]  = relu(out[m][:] W1^T + b1)                    conv1 + FrozenBN + ReLU of block j + 1            (4P -> P channels)
//
// Run apart (pw_resident2_kernel, then conv_gemm_big8_kernel<false>) the 4P-channel block output is written by the first launch and
// read back from HBM by the second: 3 584 B/px-units for the pair of a layer3 block (P = 256) where the chain moves 2 560 - both
// members are HBM-bound, so time follows bytes.  Replaces the pair of calls models/backbone.py:97-98 makes through torchvision's
// Bottleneck.forward (conv3 / bn3 / += identity / relu of block j, conv1 / bn1 / relu of block j + 1) in BOTH trunk passes of a step
// (models/tubedetr.py:127-134: the no-grad fast pass and the saved slow pass - `out` and `h1` are written where the unfused pair
// writes them, so the backward walks the same workspace).
//
// Structure: WAVE-PRIVATE CHAIN.  A workgroup of FOUR wavefronts (one per SIMD, 512 registers each) owns 128 rows; a wavefront owns 32
// of them for BOTH products and walks the 4P output channels of conv3 in chunks of 128:
//   phase A  acc3[32 x 128] = y2[32 x P] . W3[chunk]^T            (y2 as MFMA B fragments in AGPRs for the whole tile)
//   epilogue + bias + residual, ReLU, bf16 rounding IN THE MFMA LAYOUT; the rounded chunk is (a) written to HBM through a
//            wavefront-private 8-KiB LDS transposition (16-byte stores of whole 256-byte row segments) and (b) ALREADY the B operand of
//   phase B  acc1[32 x P] += chunk[32 x 128] . W1[:, chunk]^T     (no LDS hand-over, no barrier between the two products)
// That works because the MFMA roles are A = weights, B = activations: D[channel][pixel] leaves 4 consecutive D rows of one pixel per
// lane, and which CHANNEL a D row is, is decided by which weight row the LDS-DMA put into that row of the stage buffer.  Rows are
// fetched in the order sigma(16 i + a) = 32 (i >> 1) + 8 (a >> 2) + 4 (i & 1) + (a & 3), so the eight values a lane holds of fragments
// (2 s, 2 s + 1) are the eight CONSECUTIVE channels 32 s + 8 lg .. + 8 = exactly the K slots (lg, 0..7) of K-step s of phase B in natural
// order: results are bit-identical to the unfused pair (same products, same summation order per output element).
// The [rows x 4P] tile never exists on chip; the two weight matrices (2 x 4P x P bf16 = 1 MiB for layer3) stream from L2 through an
// eight-slot ring of 16-KiB stages (each stage = 128 weight rows x 64 k = 16 fragment reads = 32 MFMAs per wavefront), LDS-DMA issued
// seven stages ahead and retired by COUNTED s_waitcnt vmcnt (never 0 in the loop).  Residual rows arrive by LDS-DMA in the staging
// region one chunk ahead.  LDS: 128 KiB ring + 4 x 8 KiB staging = all 160 KiB.
//
// Wait accounting (loads retire in issue order among themselves; stores in flight only make a counted wait stricter).  Per chunk the
// VMEM loads of a wavefront are, in program order:  D(0) D(1) D(2) D(3) | LB | D(4) D(5) D(6) D(7)   with D(q) = the four 1-KiB weight
// pieces issued in the middle of stage q (for the stage seven ahead) and LB = the loads issued behind epilogue A:
//   normal chunk: 8 residual pieces + 8 bias loads (for the NEXT chunk);
//   last chunk of a tile: 16 bias1 loads (this tile's final epilogue), 16 y2 loads + 8 residual pieces + 8 bias loads (next tile).
//   mid-stage q waits for the pieces of stage q + 1 = D(q - 6): younger loads = five D's (20) + LB when it lies in between (q = 0, 1, 4..7).
// TD_CHAIN_SAFE=1 (A/B build) replaces every counted wait by vmcnt(0): results must not change.
#include <type_traits>

#include "td_common.h"

namespace td {

typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* clds_t;
template <int I>
using cic = std::integral_constant<int, I>;
template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (B < E) {
    f(cic<B>{});
    sfor<B + 1, E>(f);
  }
}

#ifndef TD_CHAIN_SAFE
#define TD_CHAIN_SAFE 0
#endif
#ifndef TD_CHAIN_NT
#define TD_CHAIN_NT 2  // bit 0: residual pieces nt, bit 1: stores nt
#endif
#ifndef TD_CHAIN_ABL
#define TD_CHAIN_ABL 0  // timing ablations (results WRONG with any bit set): 1 no epilogue A, 2 no weight DMA, 4 no fragment reads, 8 no barriers, 16 no residual pieces / output stores, 32 no MFMAs
#endif

struct ChainParams {
  const char* y2;
  const char* w3;
  const float* b3;
  const char* res;
  char* out;
  const char* w1;
  const float* b1;
  char* h1;
  int M;
  uint32_t y2_bytes, out_bytes, w_bytes;
};

// (asm operands inside a lambda nested in another lambda cannot name the kernel's locals: constant-index unrolling is by macro)
#define TD_REP4(M) M(0) M(1) M(2) M(3)
#define TD_REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define TD_REP16(M) TD_REP8(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int N>
__device__ __forceinline__ void cwait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TD_CHAIN_SAFE ? 0 : (N > 63 ? 63 : N)) : "memory");
}

template <int P>
__global__ __launch_bounds__(256, 1) void pw_chain2_kernel(ChainParams p, int MT) {
  static_assert(P == 256, "layer3 geometry: four K tiles of conv3, two K tiles x two row halves of the next conv1 per chunk");
  constexpr int ES = 2, N3 = 4 * P, NCH = N3 / 128;
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  constexpr int SLOT = 16384;
  __shared__ __attribute__((aligned(1024))) char ring[8 * SLOT];
  __shared__ __attribute__((aligned(1024))) char stg_all[4 * 8192];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y2, 0, p.y2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.h1, 0, p.y2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.b3, 0, N3 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.b1, 0, P * 4, 0x00020000);

  // ---- weight pieces: piece k of this wavefront fills stage rows 8 (4 k + wave) .. + 8; LDS slot (lane & 7) of a row takes source
  // chunk (lane & 7) ^ (row & 7) (the swizzle sits on the source address); stage row rho holds weight row sigma(rho) of the block ----
  uint32_t v3[4], v1[4];
  {
    const int drow = lane >> 3, dch = (lane & 7) ^ drow;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rho = 8 * (4 * k + wave) + drow, i = rho >> 4, a = rho & 15;
      const int sig = 32 * (i >> 1) + 8 * (a >> 2) + 4 * (i & 1) + (a & 3);
      v3[k] = (uint32_t)(sig * P + dch * 8) * ES;
      v1[k] = (uint32_t)(sig * N3 + dch * 8) * ES;
    }
  }
  // stage position PQ of chunk cn: 0..3 = K tile PQ of W3[chunk cn]; 4..7 = (K tile (PQ - 4) >> 1, row half (PQ - 4) & 1) of W1[:, chunk cn]
  auto issue_D = [&](auto PQ_, int cn) {
    constexpr int PQ = decltype(PQ_)::value;
    if constexpr ((TD_CHAIN_ABL & 2) != 0) return;
    char* const dst = ring + PQ * SLOT;
    if constexpr (PQ < 4) {
      const int soff = cn * (128 * P * ES) + PQ * 128;
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (clds_t)(dst + (4 * k + wave) * 1024), 16, v3[k], soff, 0, 0);
    } else {
      constexpr int kt = (PQ - 4) >> 1, nh = (PQ - 4) & 1;
      const int soff = nh * (128 * N3 * ES) + cn * 256 + kt * 128;
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (clds_t)(dst + (4 * k + wave) * 1024), 16, v1[k], soff, 0, 0);
    }
  };

  // ---- fragment reads: stage row 16 f + lr, 16-byte chunk (ks * 4 + lg) ^ (lr & 7); the DS offset field holds 16 bits: slots 4..7 get their own base ----
  const uint32_t ring0 = (uint32_t)(uintptr_t)(clds_t)ring;
  const uint32_t rb0 = ring0 + (uint32_t)(lr * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t rbase[2][2] = {{rb0, rb0 ^ 64u}, {rb0 + 65536u, (rb0 ^ 64u) + 65536u}};  // [slot >= 4][k-step]

  // ---- staging region of this wavefront: 32 rows x 256 B, 16-byte chunk index XOR (row & 15).  MFMA-layout side: row 16 j + lr, chunk 4 s + lg;
  // row-major side: pass u = row 4 u + lg, chunk lr ----
  const uint32_t stg0 = (uint32_t)(uintptr_t)(clds_t)stg_all + (uint32_t)wave * 8192u;
  uint32_t sma[4];  // MFMA-layout address of (j = 0, s): + j * 4096
#pragma unroll
  for (int s = 0; s < 4; ++s) sma[s] = stg0 + (uint32_t)(lr * 256 + (((4 * s + lg) ^ lr) << 4));
  uint32_t sra[4];  // row-major address of pass u & 3: + (u >> 2) * 4096
#pragma unroll
  for (int u = 0; u < 4; ++u) sra[u] = stg0 + (uint32_t)((4 * u + lg) * 256 + ((lr ^ (4 * u + lg)) << 4));

  f32x4 acc3[8][2], acc1[16][2];  // AGPRs
  cu32x4 y2f[2][8];               // AGPRs: B fragments of this wavefront's 32 rows, [row fragment][k-step]
  cu32x4 W[8];                    // rolling window of weight fragments
  cu32x4 oc[2][4];                // the rounded chunk = B fragments of phase B, [row fragment][k-step]
  f32x4 b3r[4][2], b1r[8][2];

  // per-tile row bookkeeping
  auto rowc = [&](int m) { return (uint32_t)min(m, p.M - 1); };
  auto issue_y2 = [&](int m0w) {  // 16 loads
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t vo = rowc(m0w + 16 * j + lr) * (uint32_t)(P * ES) + (uint32_t)lg * 16u;
#define TD_Y2(KS) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=a"(y2f[j][KS]) : "v"(vo), "s"(rs_y2), "n"(KS * 64) : "memory");
      TD_REP8(TD_Y2)
#undef TD_Y2
    }
  };
  auto issue_res = [&](int m0w, int cn) {  // 8 pieces: piece u = rows 4 u .. + 4 of the staging region, 16 lanes x 16 B per row
    const int soff = cn * 256;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = 4 * u + lg;
      const uint32_t vo = rowc(m0w + row) * (uint32_t)(N3 * ES) + (uint32_t)((lr ^ (row & 15)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (clds_t)(stg_all + wave * 8192 + u * 1024), 16, (TD_CHAIN_ABL & 16) ? OOB : vo, soff, 0, (TD_CHAIN_NT & 1) ? 2 : 0);
    }
  };
  auto issue_b3 = [&](int cn) {  // 8 loads
    const uint32_t vo = (uint32_t)lg * 32u;
    const int soff = cn * 512;
#define TD_B3(S)                                                                                                                                          \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][0]) : "v"(vo), "s"(rs_b3), "s"(soff), "n"(S * 128) : "memory");      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][1]) : "v"(vo), "s"(rs_b3), "s"(soff), "n"(S * 128 + 16) : "memory");
    TD_REP4(TD_B3)
#undef TD_B3
  };
  auto issue_b1 = [&]() {  // 16 loads
    const uint32_t vo = (uint32_t)lg * 32u;
#define TD_B1(S)                                                                                                                             \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(b1r[S][0]) : "v"(vo), "s"(rs_b1), "n"(S * 128) : "memory");      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(b1r[S][1]) : "v"(vo), "s"(rs_b1), "n"(S * 128 + 16) : "memory");
    TD_REP8(TD_B1)
#undef TD_B1
  };

  // ---- one stage: 16 fragments (k-step ks, row fragment f), two MFMAs each; fragment t lives in W[t & 7] and is re-requested for
  // t + 8 right behind the MFMAs that consumed it (DS operations return in order: lgkmcnt(7) = "the oldest of eight has returned") ----
#define TD_MFMA_PAIR_(Q, KS, F)                                                                                                             \
  if constexpr ((Q) < 4) {                                                                                                                 \
    if constexpr ((Q) == 0 && (KS) == 0) {                                                                                                 \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc3[F][0]) : "v"(W[F]), "a"(y2f[0][((Q) < 4 ? 2 * (Q) + (KS) : 0)]));     \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc3[F][1]) : "v"(W[F]), "a"(y2f[1][((Q) < 4 ? 2 * (Q) + (KS) : 0)]));     \
    } else {                                                                                                                               \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc3[F][0]) : "v"(W[F]), "a"(y2f[0][((Q) < 4 ? 2 * (Q) + (KS) : 0)]));    \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc3[F][1]) : "v"(W[F]), "a"(y2f[1][((Q) < 4 ? 2 * (Q) + (KS) : 0)]));    \
    }                                                                                                                                      \
  } else {                                                                                                                                 \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + (F) : 0)][0]) : "v"(W[F]), "v"(oc[0][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + (KS) : 0)])); \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + (F) : 0)][1]) : "v"(W[F]), "v"(oc[1][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + (KS) : 0)])); \
  }
#if TD_CHAIN_ABL & 32
#define TD_MFMA_PAIR(Q, KS, F) asm volatile("" : "+v"(W[F]));
#else
#define TD_MFMA_PAIR(Q, KS, F) TD_MFMA_PAIR_(Q, KS, F)
#endif
#if TD_CHAIN_ABL & 4
#define TD_FRAG_READ(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr));
#else
#define TD_FRAG_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
#endif
  auto stage = [&](auto Q_, auto LAST_, int c, int cnext) {
    constexpr int Q = decltype(Q_)::value;
    constexpr bool LAST = decltype(LAST_)::value;
    constexpr int QN = (Q + 1) & 7;
#define TD_H0(F)                                                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(W[F]));                                                                                         \
  TD_MFMA_PAIR(Q, 0, F)                                                                                                                      \
  TD_FRAG_READ(W[F], rbase[Q >= 4][1], (Q & 3) * SLOT + F * 2048)
    TD_REP8(TD_H0)
#undef TD_H0
    // the pieces of stage Q + 1 (this wavefront's, then - behind the barrier - everybody's) have landed; the slot of stage Q - 1 is free
    constexpr int CQ = (Q == 2 || Q == 3) ? 20 : ((LAST && Q >= 4) ? 68 : 36);
    if constexpr (!(TD_CHAIN_ABL & 2)) cwait_vm<CQ>();
    if constexpr (!(TD_CHAIN_ABL & 8)) __builtin_amdgcn_s_barrier();
    issue_D(cic<(Q + 7) & 7>{}, Q == 0 ? c : cnext);
#define TD_H1(F)                                                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(W[F]));                                                                                         \
  TD_MFMA_PAIR(Q, 1, F)                                                                                                                      \
  TD_FRAG_READ(W[F], rbase[QN >= 4][0], (QN & 3) * SLOT + F * 2048)
    TD_REP8(TD_H1)
#undef TD_H1
  };

  const int G = (int)gridDim.x;
  int tile = blockIdx.x;
  if (tile >= MT) return;  // (uniform)
  int m0w = tile * 128 + wave * 32;

  // ---- prologue: the steady state's in-flight set, drained once ----
  issue_y2(m0w);
  issue_res(m0w, 0);
  issue_b3(0);
  issue_D(cic<0>{}, 0); issue_D(cic<1>{}, 0); issue_D(cic<2>{}, 0); issue_D(cic<3>{}, 0); issue_D(cic<4>{}, 0); issue_D(cic<5>{}, 0); issue_D(cic<6>{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#define TD_W0(F) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(W[F]) : "v"(rbase[0][0]), "n"(F * 2048));
  TD_REP8(TD_W0)
#undef TD_W0
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    acc1[n][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[n][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto chunk = [&](auto LAST_, int c, int m0w_next) {
    constexpr bool LAST = decltype(LAST_)::value;
    const int cnext = LAST ? 0 : c + 1;
    if (c == 0) {  // (uniform) the y2 fragments of this tile: requested behind epilogue A of the previous tile's last chunk
      cwait_vm<32>();
#define TD_T(KS) asm volatile("" : "+a"(y2f[0][KS]), "+a"(y2f[1][KS]));
      TD_REP8(TD_T)
#undef TD_T
    }
    stage(cic<0>{}, LAST_, c, cnext);
    stage(cic<1>{}, LAST_, c, cnext);
    stage(cic<2>{}, LAST_, c, cnext);
    stage(cic<3>{}, LAST_, c, cnext);
#if !(TD_CHAIN_ABL & 1)
    // ---- epilogue A ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs' results are read below
#define TD_T(F) asm volatile("" : "+a"(acc3[F][0]), "+a"(acc3[F][1]));
    TD_REP8(TD_T)
#undef TD_T
    // residual pieces + bias of this chunk have landed (younger: the D's of stages 4..7 of the previous chunk and 0..3 of this one)
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(b3r[0][0]), "+v"(b3r[0][1]), "+v"(b3r[1][0]), "+v"(b3r[1][1]), "+v"(b3r[2][0]), "+v"(b3r[2][1]), "+v"(b3r[3][0]), "+v"(b3r[3][1]),
                   "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7])
                 : "n"(TD_CHAIN_SAFE ? 0 : 32)
                 : "memory");
    cu32x4 rr[2][4];
#define TD_RR(S)                                                                                              \
  asm volatile("ds_read_b128 %0, %1" : "=v"(rr[0][S]) : "v"(sma[S]) : "memory");                             \
  asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(rr[1][S]) : "v"(sma[S]) : "memory");
    TD_REP4(TD_RR)
#undef TD_RR
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(rr[0][0]), "+v"(rr[0][1]), "+v"(rr[0][2]), "+v"(rr[0][3]), "+v"(rr[1][0]), "+v"(rr[1][1]), "+v"(rr[1][2]), "+v"(rr[1][3]),
                   "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * h + r] = acc3[2 * s + h][j][r] + b3r[s][h][r];
        const cu32x4 q = rr[j][s];
        const float r8[8] = {__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u),
                             __uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u)};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + r8[e], 0.f);
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        cu32x4 o;
        {
          b2 a0 = {(__bf16)v[0], (__bf16)v[1]}, a1 = {(__bf16)v[2], (__bf16)v[3]}, a2 = {(__bf16)v[4], (__bf16)v[5]}, a3 = {(__bf16)v[6], (__bf16)v[7]};
          o.x = *(uint32_t*)&a0; o.y = *(uint32_t*)&a1; o.z = *(uint32_t*)&a2; o.w = *(uint32_t*)&a3;
        }
        oc[j][s] = o;
      }
    // the rounded chunk through the staging region (its residual image has been read): MFMA layout in, whole rows out
#define TD_OW(S)                                                                                  \
  asm volatile("ds_write_b128 %0, %1" ::"v"(sma[S]), "v"(oc[0][S]) : "memory");                  \
  asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(sma[S]), "v"(oc[1][S]) : "memory");
    TD_REP4(TD_OW)
#undef TD_OW
    cu32x4 ot[8];
#define TD_OR(U) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ot[U]) : "v"(sra[U & 3]), "n"((U >> 2) * 4096) : "memory");
    TD_REP8(TD_OR)
#undef TD_OR
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ot[0]), "+v"(ot[1]), "+v"(ot[2]), "+v"(ot[3]), "+v"(ot[4]), "+v"(ot[5]), "+v"(ot[6]), "+v"(ot[7]) : : "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = m0w + 4 * u + lg;
      const uint32_t off = (m < p.M && !(TD_CHAIN_ABL & 16)) ? ((uint32_t)m * (uint32_t)N3 + (uint32_t)(c * 128 + lr * 8)) * ES : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(ot[u], rs_out, (int)off, 0, (TD_CHAIN_NT & 2) ? 2 : 0);
    }
#else
    asm volatile("" : "+v"(oc[0][0]), "+v"(oc[0][1]), "+v"(oc[0][2]), "+v"(oc[0][3]), "+v"(oc[1][0]), "+v"(oc[1][1]), "+v"(oc[1][2]), "+v"(oc[1][3]));
#endif
    // ---- LB: loads for the next chunk (and, from a tile's last chunk, for its final epilogue and the next tile) ----
    if constexpr (LAST) {
      issue_b1();
      issue_y2(m0w_next);
      issue_res(m0w_next, 0);
      issue_b3(0);
    } else {
      issue_res(m0w, cnext);
      issue_b3(cnext);
    }
    stage(cic<4>{}, LAST_, c, cnext);
    stage(cic<5>{}, LAST_, c, cnext);
    stage(cic<6>{}, LAST_, c, cnext);
    stage(cic<7>{}, LAST_, c, cnext);
    if constexpr (LAST) {
      // ---- final epilogue: h1 = relu(acc1 + b1), straight from the MFMA layout (eight consecutive channels per lane; a tenth of the tile's bytes) ----
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#define TD_T(N) asm volatile("" : "+a"(acc1[N][0]), "+a"(acc1[N][1]));
      TD_REP16(TD_T)
#undef TD_T
      // bias1 landed (younger: 16 y2 + 8 residual + 8 bias + D(4..7))
      asm volatile("s_waitcnt vmcnt(%16)"
                   : "+v"(b1r[0][0]), "+v"(b1r[0][1]), "+v"(b1r[1][0]), "+v"(b1r[1][1]), "+v"(b1r[2][0]), "+v"(b1r[2][1]), "+v"(b1r[3][0]), "+v"(b1r[3][1]),
                     "+v"(b1r[4][0]), "+v"(b1r[4][1]), "+v"(b1r[5][0]), "+v"(b1r[5][1]), "+v"(b1r[6][0]), "+v"(b1r[6][1]), "+v"(b1r[7][0]), "+v"(b1r[7][1])
                   : "n"(TD_CHAIN_SAFE ? 0 : 48)
                   : "memory");
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0w + 16 * j + lr;
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // channels 32 g + 8 lg .. + 8: fragments 2 g, 2 g + 1
          float v[8];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * h + r] = fmaxf(acc1[2 * g + h][j][r] + b1r[g][h][r], 0.f);
          typedef __bf16 b2 __attribute__((ext_vector_type(2)));
          b2 a0 = {(__bf16)v[0], (__bf16)v[1]}, a1 = {(__bf16)v[2], (__bf16)v[3]}, a2 = {(__bf16)v[4], (__bf16)v[5]}, a3 = {(__bf16)v[6], (__bf16)v[7]};
          const cu32x4 o = {*(uint32_t*)&a0, *(uint32_t*)&a1, *(uint32_t*)&a2, *(uint32_t*)&a3};
          const uint32_t off = m < p.M ? ((uint32_t)m * (uint32_t)P + (uint32_t)(32 * g + 8 * lg)) * ES : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(o, rs_h1, (int)off, 0, 0);
        }
      }
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        acc1[n][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[n][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

#pragma unroll 1
  for (;;) {
    const int m0w_next = (tile + G) * 128 + wave * 32;
#pragma unroll 1
    for (int c = 0; c < NCH - 1; ++c) chunk(std::false_type{}, c, m0w_next);
    chunk(std::true_type{}, NCH - 1, m0w_next);
    tile += G;
    if (tile >= MT) break;
    m0w = m0w_next;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing of the run-ahead may outlive the workgroup's LDS / registers
}

}  // namespace td

using namespace td;

extern "C" int td_pw_chain2(const void* y2, const void* w3, const float* b3, const void* residual, void* out, const void* w1, const float* b1,
                            void* h1, int M, int planes, int dtype, td_stream_t stream) {
  TD_REQUIRE(y2 && w3 && b3 && residual && out && w1 && b1 && h1 && M >= 1, "td_pw_chain2: bad arguments");
  TD_REQUIRE(dtype == TD_BF16 && planes == 256, "td_pw_chain2: bf16, planes = 256 (a layer3 bottleneck pair) only");
  TD_REQUIRE((double)M * planes * 4 * 2 < 4294967000.0, "td_pw_chain2: tensor exceeds the 4 GiB buffer-descriptor range");
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  ChainParams p;
  p.y2 = (const char*)y2;
  p.w3 = (const char*)w3;
  p.b3 = b3;
  p.res = (const char*)residual;
  p.out = (char*)out;
  p.w1 = (const char*)w1;
  p.b1 = b1;
  p.h1 = (char*)h1;
  p.M = M;
  p.y2_bytes = (uint32_t)((size_t)M * planes * 2);
  p.out_bytes = (uint32_t)((size_t)M * planes * 4 * 2);
  p.w_bytes = (uint32_t)((size_t)planes * planes * 4 * 2);
  const int MT = cdiv(M, 128);
  hipStream_t st = (hipStream_t)stream;
  const bool prof = prof_on();
  if (prof) {
    prof_begin(TD_PROF_CHAIN, dtype, 4.0 * M * (4.0 * planes) * planes, st, M, 4 * planes, planes, 1, 1, 0);
    prof_set_bytes(((double)M * planes * 2 + (double)M * planes * 4 * 2 + 2.0 * planes * planes * 4) * 2.0);
  }
  pw_chain2_kernel<256><<<dim3(std::min(n_cu, MT)), 256, 0, st>>>(p, MT);
  if (prof) prof_end(st);
  return check_launch("td_pw_chain2");
}
