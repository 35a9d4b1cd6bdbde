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
// fetched in an order (tau / sigma in the kernel) that makes the eight values a lane holds of fragments (2 s, 2 s + 1) the eight
// CONSECUTIVE channels 32 s + 8 lg .. + 8 = exactly the K slots (lg, 0..7) of K-step s of phase B in natural order: results are
// bit-identical to the unfused pair (same products, same summation order per output element, same (acc + bias) + residual order).
// The [rows x 4P] tile never exists on chip; the two weight matrices (2 x 4P x P bf16 = 1 MiB for layer3) stream from L2 through an
// eight-slot ring of 16-KiB stages (a stage = 16 fragment reads = 32 MFMAs per wavefront), LDS-DMA issued seven stages ahead and retired by
// COUNTED s_waitcnt vmcnt (never 0 in the loop).  A conv3 stage is 32 weight ROWS over all of K, so the two fragments of K-step s of
// phase B are final when stage s ends: their epilogue runs in SLICES under the MFMAs of stage s + 1 (two accumulator reads, the adds,
// the ReLU and one v_cvt_pk per slice, pinned between MFMA pairs by sched_barrier), the rounded rows go back through the staging region
// under stage 5, and only the tile's final epilogue (h1, a tenth of the bytes) stands alone.  Residual rows are coalesced row-major loads
// requested a whole chunk ahead and re-laid through the staging region; bias1 waits in the staging region's idle window.  LDS: 128 KiB
// ring + 4 x 8 KiB staging = all 160 KiB.  The accumulator file (256 AGPRs: y2 fragments, both accumulators) is allocated by hand.
//
// Wait accounting (loads retire in issue order among themselves; stores in flight only make a counted wait stricter).  Per chunk the
// VMEM loads of a wavefront are, in program order:
//   D0 RL B3(1) | D1 B3(2) | D2 B3(3) | D3 | [Y2] D4 | D5 | [B1] D6 | D7 B3(0')        (one group per stage)
// D(q) = the four 1-KiB weight pieces issued behind the barrier in the middle of stage q (for the stage seven ahead), RL = the 8 residual
// loads of the next chunk, B3(s) = the 2 bias loads of K-step s (two stages ahead of their epilogue), and in a tile's last chunk Y2 = the
// next tile's 16 y2 loads, B1 = the one piece that brings bias1 into the staging region.  The wait in the middle of stage q is for the
// pieces of stage q + 1 = D(q - 6): its count = the loads listed behind D(q - 6) up to there (CQ in the kernel; the other waits likewise).
// TD_CHAIN_SAFE=1 (A/B build) replaces every counted wait by 0: results must not change (tools/chain_probe.py runs both).
//
// What it buys (profiles/r06_chain_probe.log): HBM bytes of the pair -28.6 %; time per launch 0.94 - 1.00 of the pair's, equal in a sustained
// loop - both forms hold the package at its 1 400 W cap, and the saved bytes are a few per cent of a launch's energy.  Every part of the
// side work (epilogue arithmetic, staging traffic, loads, stores) adds its share to the launch time whether it is placed under the MFMAs
// or not: the additive behaviour of a power-capped launch, not of a critical path.
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
#ifndef TD_CHAIN_LEAD4
#define TD_CHAIN_LEAD4 7
#endif
#ifndef TD_CHAIN_BP
#define TD_CHAIN_BP 7
#endif
#ifndef TD_CHAIN_ABL
#define TD_CHAIN_ABL 0  // timing ablations (results WRONG with any bit set): 1 no side work (epilogue slices, staging traffic, output stores), 2 no weight DMA, 4 no fragment reads, 8 no barriers, 16 residual loads and output stores out of range (no HBM traffic), 32 no MFMAs, 64 output stores out of range, 128 no epilogue arithmetic, 256 residual loads out of range
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

// NW = wavefronts per workgroup.  Shipped: 4 (one per SIMD, 32 rows each, 512 registers).  The body also instantiates for 8 (two per SIMD, 16
// rows each, 256 registers, one weight fragment per MFMA): built and measured in round 6 - 3.5 % faster than NW = 4 on the layer3 shape, and
// WRONG on some wavefronts of some launches (accumulators of whole wavefronts off from stage 0 on, with every counted wait replaced by 0,
// without any side work, with 32 wait states behind every MFMA and with the weight window pinned to fixed registers): not instantiated.
template <int P, int NW>
__device__ __forceinline__ void pw_chain2_body(const ChainParams& p, const int MT) {
  static_assert(P == 256, "layer3 geometry: four 32-row stages of conv3, two K tiles x two row halves of the next conv1 per chunk");
  static_assert(NW == 4 || NW == 8, "one or two wavefronts per SIMD");
  constexpr int ES = 2, N3 = 4 * P, NCH = N3 / 128;
  constexpr int PW = 128 / NW, J = PW / 16, U = PW / 4, ND = 16 / NW;  // rows per wavefront, row fragments, row-major passes, weight pieces per wavefront and stage
  constexpr uint32_t OOB = 0xFFFFFFF0u;
  constexpr int SLOT = 16384;
  constexpr int LEAD = TD_CHAIN_LEAD4, WN = LEAD < 8 ? 8 : 16, BP = TD_CHAIN_BP;  // window registers; the step whose tail holds the stage's barrier
  static_assert(LEAD < WN && 16 - LEAD > BP && BP >= 2 && BP < 16, "the next stage's first fragment is requested behind the barrier that publishes it");  // a weight fragment is requested LEAD steps ahead of its MFMAs, into the registers step T + LEAD - 8 consumed
  __shared__ __attribute__((aligned(1024))) char ring[8 * SLOT];
  __shared__ __attribute__((aligned(1024))) char stg_all[128 * 256];

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

  // ---- weight pieces.  A stage is 128 LDS rows of 128 B = 16 pieces of 8 rows; this wavefront fills pieces k NW + wave; LDS slot (lane & 7) of a
  // row takes source chunk (lane & 7) ^ (row & 7).  conv3 stage q of a chunk = weight rows 32 q .. + 32 over ALL of K: LDS row 32 kt + r holds K
  // tile kt of weight row 32 q + tau(r), tau(16 i2 + a) = 8 (a >> 2) + 4 i2 + (a & 3).  conv1 stage (kt, nh) = K tile kt of the chunk for weight
  // rows 128 nh + sigma(rho), sigma(16 i + a) = 32 (i >> 1) + 8 (a >> 2) + 4 (i & 1) + (a & 3). ----
  uint32_t v3[ND], v1[ND];
  {
    const int drow = lane >> 3, dch = (lane & 7) ^ drow;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int pi = k * NW + wave;
      const int kt = pi >> 2, r32 = 8 * (pi & 3) + drow, i2 = r32 >> 4, a3 = r32 & 15;
      const int tau = 8 * (a3 >> 2) + 4 * i2 + (a3 & 3);
      v3[k] = (uint32_t)(tau * P + dch * 8) * ES + (uint32_t)kt * 128u;
      const int rho = 8 * pi + drow, i = rho >> 4, a = rho & 15;
      const int sig = 32 * (i >> 1) + 8 * (a >> 2) + 4 * (i & 1) + (a & 3);
      v1[k] = (uint32_t)(sig * N3 + dch * 8) * ES;
    }
  }
  auto issue_D = [&](auto PQ_, int cn) {
    constexpr int PQ = decltype(PQ_)::value;
    if constexpr ((TD_CHAIN_ABL & 2) != 0) return;
    char* const dst = ring + PQ * SLOT;
    if constexpr (PQ < 4) {
      const int soff = (cn * 128 + PQ * 32) * (P * ES);
#pragma unroll
      for (int k = 0; k < ND; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (clds_t)(dst + (k * NW + wave) * 1024), 16, v3[k], soff, 0, 0);
    } else {
      constexpr int kt = (PQ - 4) >> 1, nh = (PQ - 4) & 1;
      const int soff = nh * (128 * N3 * ES) + cn * 256 + kt * 128;
#pragma unroll
      for (int k = 0; k < ND; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (clds_t)(dst + (k * NW + wave) * 1024), 16, v1[k], soff, 0, 0);
    }
  };

  // ---- fragment reads: LDS row + lr, 16-byte chunk (par * 4 + lg) ^ (lr & 7); the DS offset field holds 16 bits: slots 4..7 get their own base ----
  const uint32_t ring0 = (uint32_t)(uintptr_t)(clds_t)ring;
  const uint32_t rb0 = ring0 + (uint32_t)(lr * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t rbase[2][2] = {{rb0, rb0 ^ 64u}, {rb0 + 65536u, (rb0 ^ 64u) + 65536u}};  // [slot >= 4][k-step parity]

  // ---- staging region of this wavefront: PW rows x 256 B, 16-byte chunk index XOR (row & 15).  MFMA-layout side: row 16 j + lr, chunk 4 s + lg;
  // row-major side: pass u = row 4 u + lg, chunk lr ----
  const uint32_t stg0 = (uint32_t)(uintptr_t)(clds_t)stg_all + (uint32_t)wave * (uint32_t)(PW * 256);
  uint32_t sma[4];  // MFMA-layout address of (j = 0, s): + j * 4096
#pragma unroll
  for (int s = 0; s < 4; ++s) sma[s] = stg0 + (uint32_t)(lr * 256 + (((4 * s + lg) ^ lr) << 4));
  uint32_t sra[4];  // row-major address of pass u & 3: + (u >> 2) * 4096
#pragma unroll
  for (int u = 0; u < 4; ++u) sra[u] = stg0 + (uint32_t)((4 * u + lg) * 256 + ((lr ^ (4 * u + lg)) << 4));

  // The accumulator file is allocated BY HAND (literal register numbers in every statement that touches it): compiler-owned "a" operands
  // were moved between registers at loop boundaries - ahead of the counted wait of an asm load still in flight, and right behind asm MFMAs
  // whose latency the compiler cannot see (wrong results on some wavefronts of some launches with two wavefronts per SIMD).
  //   a[AY + 4 (8 j + ks) ..]      y2 B fragments of row fragment j, k-step ks          (32 J registers)
  //   a[A3 + 4 (J f + j) ..]       conv3 accumulator of weight-row fragment f (0..7)     (32 J)
  //   a[A1 + 4 (J n + j) ..]       conv1 accumulator of weight-row fragment n (0..15)    (64 J)
  constexpr int AY = 0, A3 = 32 * J, A1 = 64 * J, NAGPR = 128 * J;
  if constexpr (NAGPR == 256) asm volatile("" ::: "a255");  // (the kernel descriptor allocates the whole file; no compiler value lives in it)
  else asm volatile("" ::: "a127");
  cu32x4 W[WN];                   // rolling window of weight fragments (each element keeps ONE physical register quadruple for the whole kernel)
#pragma unroll
  for (int k = 0; k < WN; ++k) asm volatile("" : "=v"(W[k]));
  cu32x4 oc[J][4];                // the rounded chunk = B fragments of phase B, [row fragment][k-step]
  cu32x4 rl[U];                   // residual rows of the NEXT chunk, row-major (pass u), in flight for a whole chunk
  cu32x4 rr[J], ot[U];
  f32x4 b3r[4][2];

  // Loads address rows past M as they come: the buffer descriptors end at the tensors' last byte, out-of-range loads return zeros.
  auto issue_y2 = [&](int m0w) {  // 8 J loads
#define TD_Y2(KS) asm volatile("buffer_load_dwordx4 a[%c0:%c1], %2, %3, 0 offen offset:%4" ::"n"(AY + 4 * (8 * j + KS)), "n"(AY + 4 * (8 * j + KS) + 3), "v"(vo), "s"(rs_y2), "n"(KS * 64) : "memory");
    {
      constexpr int j = 0;
      const uint32_t vo = (uint32_t)(m0w + lr) * (uint32_t)(P * ES) + (uint32_t)lg * 16u;
      TD_REP8(TD_Y2)
    }
    if constexpr (J > 1) {
      constexpr int j = J - 1;
      const uint32_t vo = (uint32_t)(m0w + 16 + lr) * (uint32_t)(P * ES) + (uint32_t)lg * 16u;
      TD_REP8(TD_Y2)
    }
#undef TD_Y2
  };
  auto issue_rl = [&](int m0w, int cn) {  // U loads: pass u = rows 4 u + lg, 16 lanes x 16 B = the chunk's 256 B of a row
    const uint32_t vo = (TD_CHAIN_ABL & (16 | 256)) ? OOB : (uint32_t)(m0w + lg) * (uint32_t)(N3 * ES) + (uint32_t)lr * 16u;
#define TD_RL(UU)                                                                                                                                 \
  if constexpr ((UU) < U) {                                                                                                                       \
    const int soff = cn * 256 + (UU) * (4 * N3 * ES);                                                                                             \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rl[(UU) < U ? (UU) : 0]) : "v"(vo), "s"(rs_res), "s"(soff) : "memory");        \
  }
    TD_REP8(TD_RL)
#undef TD_RL
  };
  auto issue_b1 = [&]() {  // ONE piece: the 256 fp32 of bias1 into the first KiB of this wavefront's staging region (free between a chunk's last row read, stage 5, and the next chunk's residual rows, stage 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b1, (clds_t)(stg_all + wave * (PW * 256)), 16, (uint32_t)lane * 16u, 0, 0, 0);
  };
#define TD_B3(S, cn)                                                                                                                                     \
  {                                                                                                                                                      \
    const uint32_t vo_ = (uint32_t)lg * 32u;                                                                                                             \
    const int so_ = (cn) * 512;                                                                                                                          \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][0]) : "v"(vo_), "s"(rs_b3), "s"(so_), "n"((S) * 128) : "memory");     \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][1]) : "v"(vo_), "s"(rs_b3), "s"(so_), "n"((S) * 128 + 16) : "memory"); \
  }

  // ---- one stage = 16 steps: step T waits for fragment T (in W[T & 7]), issues its J MFMAs and requests fragment T + 7 into the registers
  // of fragment T - 1 (steps 9..15: the next stage's first seven).  DS operations return in order: lgkmcnt(n) with n = the DS operations issued behind
  // the wanted read.  Phase A (Q < 4), fragment T = (k-step T >> 1, row fragment T & 1) of weight rows 32 Q ..: LDS row 32 (T >> 2) + 16 (T & 1),
  // parity (T >> 1) & 1.  Phase B, fragment T = (k-step T >> 3, row fragment T & 7). ----
#define TD_RD_HI(Q) ((Q) >= 4)
#define TD_RD_PAR(Q, T) ((Q) < 4 ? (((T) >> 1) & 1) : ((T) >> 3))
#define TD_RD_OFF(Q, T) (((Q) & 3) * SLOT + ((Q) < 4 ? ((T) >> 2) * 4096 + ((T) & 1) * 2048 : ((T) & 7) * 2048))
#define TD_A3(Q, T) ((Q) < 4 ? 2 * (Q) + ((T) & 1) : 0)
#define TD_A1(Q, T) ((Q) >= 4 ? 8 * (((Q) - 4) & 1) + ((T) & 7) : 0)
#define TD_OS(Q, T) ((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + ((T) >> 3) : 0)
#define TD_MFMA_J(Q, T, JJ)                                                                                                                \
  if constexpr ((JJ) < J) {                                                                                                                \
    constexpr int j_ = (JJ) < J ? (JJ) : 0;                                                                                                \
    if constexpr ((Q) < 4) {                                                                                                               \
      constexpr int d_ = A3 + 4 * (J * TD_A3(Q, T) + j_), y_ = AY + 4 * (8 * j_ + ((T) >> 1));                                             \
      if constexpr (((T) >> 1) == 0) {                                                                                                     \
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, a[%c3:%c4], 0" ::"n"(d_), "n"(d_ + 3), "v"(W[(T) & (WN - 1)]), "n"(y_), "n"(y_ + 3)); \
      } else {                                                                                                                             \
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, a[%c3:%c4], a[%c0:%c1]" ::"n"(d_), "n"(d_ + 3), "v"(W[(T) & (WN - 1)]), "n"(y_), "n"(y_ + 3)); \
      }                                                                                                                                    \
    } else {                                                                                                                               \
      constexpr int d_ = A1 + 4 * (J * TD_A1(Q, T) + j_);                                                                                  \
      if constexpr (FIRST && (Q) <= 5 && ((T) >> 3) == 0) { /* a tile's first touch of this accumulator: C = 0 */                          \
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, 0" ::"n"(d_), "n"(d_ + 3), "v"(W[(T) & (WN - 1)]), "v"(oc[j_][TD_OS(Q, T)]));    \
      } else {                                                                                                                             \
        asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(d_), "n"(d_ + 3), "v"(W[(T) & (WN - 1)]), "v"(oc[j_][TD_OS(Q, T)])); \
      }                                                                                                                                    \
    }                                                                                                                                      \
  }
#if TD_CHAIN_ABL & 32
#define TD_MFMA_PAIR(Q, T) asm volatile("" : "+v"(W[(T) & (WN - 1)]));
#else
#define TD_MFMA_PAIR(Q, T) TD_MFMA_J(Q, T, 0) TD_MFMA_J(Q, T, 1)
#endif
#if TD_CHAIN_ABL & 4
#define TD_FRAG_READ(dst, addr, off) asm volatile("" : "+v"(dst) : "v"(addr));
#else
// ("+v": the destination stays the register the variable lives in - with "=v" the allocator is free to hand out the registers an MFMA issued
//  just before was given, and with two wavefronts per SIMD that MFMA may not have read them yet: wrong results on some wavefronts)
#define TD_FRAG_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(off));
#endif
#define TD_LGKM(n) (TD_CHAIN_SAFE ? 0 : ((n) > 15 ? 15 : (n)))
#define TD_VM(n) (TD_CHAIN_SAFE ? 0 : ((n) > 63 ? 63 : (n)))
  // load counts between a request and the wait for it (header comment): D = ND pieces per stage, RL = U residual loads, B3 = 2 bias loads
  // bias of K-step s: requested two stages ahead (end of stage s - 1; s = 0: end of stage 7 of the previous chunk)
  constexpr int VM_RL = 7 * ND + 8, VM_Y2 = 4 * ND + 3, VM_B1 = 2 * ND + 2, NY2 = 8 * J;

  // side work of step T of stage Q (everything that is not the weight stream): the chunk's epilogue in slices under the MFMAs
  auto side = [&](auto Q_, auto T_, auto LAST_, int c, int m0w) {
    constexpr int Q = decltype(Q_)::value, T = decltype(T_)::value;
    constexpr bool LAST = decltype(LAST_)::value;
#if !(TD_CHAIN_ABL & 1)
    if constexpr (Q >= 1 && Q <= 4) {
      constexpr int S = Q - 1;  // epilogue of K-step S of phase B = fragments 2 S, 2 S + 1 of phase A, final since the end of stage S
      if constexpr (T == 1) {
        // the residual fragments (requested at the end of the previous stage; behind them: the W reads of steps 0, 1) and this K-step's bias
        // (requested at the end of stage S + 1 of the previous chunk) are here
        // loads behind the bias request: S = 0: D(0), the residual rows, the bias of S = 1; S = 1, 2: D(S), the next bias; S = 3: D(3) (+ the next tile's y2 in a tile's last chunk)
        constexpr int VM_B3 = S == 0 ? ND + U + 2 : S <= 2 ? ND + 2 : ND + (LAST ? NY2 : 0);
        if constexpr (J == 2) asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(%5)" : "+v"(rr[0]), "+v"(rr[J - 1]), "+v"(b3r[S][0]), "+v"(b3r[S][1]) : "n"(TD_VM(VM_B3)), "n"(TD_LGKM(2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%3) lgkmcnt(%4)" : "+v"(rr[0]), "+v"(b3r[S][0]), "+v"(b3r[S][1]) : "n"(TD_VM(VM_B3)), "n"(TD_LGKM(2)) : "memory");
      }
      if constexpr (T >= 2 && T < 2 + 4 * J && !(TD_CHAIN_ABL & 128)) {
        constexpr int K = T - 2, JJ = K >> 2, PP = K & 3, H = PP >> 1, R0 = 2 * (PP & 1);
        const uint32_t wd = rr[JJ][PP];
        float a0_, a1_;  // (the MFMAs that wrote these accumulators were issued in stage S: at least two steps back)
        asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_accvgpr_read_b32 %1, a%c3" : "=v"(a0_), "=v"(a1_) : "n"(A3 + 4 * (J * (2 * S + H) + JJ) + R0), "n"(A3 + 4 * (J * (2 * S + H) + JJ) + R0 + 1));
        float x0 = a0_ + b3r[S][H][R0];
        float x1 = a1_ + b3r[S][H][R0 + 1];
        x0 += __uint_as_float(wd << 16);
        x1 += __uint_as_float(wd & 0xffff0000u);
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 pk = {(__bf16)x0, (__bf16)x1};
        // ReLU BEHIND the rounding, on the packed pair (v_pk_max_i16 against 0: a negative bf16, -0.0 included, becomes +0.0): one operation
        // per pair - fmaxf(x, 0.f) in front of the conversion is two per VALUE (the compiler quiets a possible signalling NaN first).  Same
        // values as the two-launch form, which clamps first: rounding is monotonic and keeps the sign.
        asm("v_pk_max_i16 %0, %1, 0" : "=v"(oc[JJ][S][PP]) : "v"(*(const uint32_t*)&pk));
      }
    }
    if constexpr (Q == 5) {
      if constexpr (T == 1) {
        if constexpr (U == 8) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(ot[0]), "+v"(ot[1]), "+v"(ot[2]), "+v"(ot[3]), "+v"(ot[U - 4]), "+v"(ot[U - 3]), "+v"(ot[U - 2]), "+v"(ot[U - 1]) : "n"(TD_LGKM(2)) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ot[0]), "+v"(ot[1]), "+v"(ot[2]), "+v"(ot[3]) : "n"(TD_LGKM(2)) : "memory");
      }
      if constexpr (T >= 2 && T < 2 + U) {
        constexpr int UU = T - 2;
        const int m = m0w + 4 * UU + lg;
        const uint32_t off = (m < p.M && !(TD_CHAIN_ABL & (16 | 64))) ? ((uint32_t)m * (uint32_t)N3 + (uint32_t)(c * 128 + lr * 8)) * ES : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(ot[UU], rs_out, (int)off, 0, (TD_CHAIN_NT & 2) ? 2 : 0);
      }
    }
#endif
  };

  auto stage = [&](auto Q_, auto FIRST_, auto LAST_, int c, int cnext, int m0w, int m0w_next) {
    constexpr int Q = decltype(Q_)::value;
    constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
    constexpr int QN = (Q + 1) & 7;
    // DS operations issued at the end of stage Q - 1 and at the start of stage Q: they sit behind the reads of fragments 0..7
    constexpr int NE_PREV = (TD_CHAIN_ABL & 1) ? 0 : (Q == 1 ? J : (Q >= 2 && Q <= 4) ? 2 * J : (Q == 5 ? J : 0));
    constexpr int NS = (TD_CHAIN_ABL & 1) ? 0 : ((Q == 0 || Q == 5) ? U : 0);
    constexpr int LG0 = LEAD - 1 + NE_PREV + NS;  // (steps 0..LEAD-1: behind the wanted read sit LEAD - 1 fragment reads and those operations)
    // ---- stage start ----
#if !(TD_CHAIN_ABL & 1)
    if constexpr (Q == 0) {
      // the residual rows of this chunk (requested at the end of stage 0 of the previous chunk) go to the staging region, row-major
      if constexpr (U == 8) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(rl[0]), "+v"(rl[1]), "+v"(rl[2]), "+v"(rl[3]), "+v"(rl[U - 4]), "+v"(rl[U - 3]), "+v"(rl[U - 2]), "+v"(rl[U - 1]) : "n"(TD_VM(VM_RL)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rl[0]), "+v"(rl[1]), "+v"(rl[2]), "+v"(rl[3]) : "n"(TD_VM(VM_RL)) : "memory");
#define TD_RW(UU) if constexpr ((UU) < U) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sra[(UU) & 3]), "v"(rl[(UU) < U ? (UU) : 0]), "n"(((UU) >> 2) * 4096) : "memory");
      TD_REP8(TD_RW)
#undef TD_RW
    }
    if constexpr (Q == 5) {
#define TD_OR(UU) if constexpr ((UU) < U) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ot[(UU) < U ? (UU) : 0]) : "v"(sra[(UU) & 3]), "n"(((UU) >> 2) * 4096) : "memory");
      TD_REP8(TD_OR)
#undef TD_OR
    }
#endif
    if constexpr (Q == 4 && LAST) issue_y2(m0w_next);  // (AGPR destinations, dead since the end of stage 3)
    if constexpr (Q == 6 && LAST) issue_b1();          // (into the registers the output rows have left)
    __builtin_amdgcn_sched_barrier(0);
#define TD_STEP(T)                                                                                                                           \
  if constexpr (!(TD_CHAIN_ABL & 1024)) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(W[(T) & (WN - 1)]) : "n"(TD_LGKM((T) < LEAD ? LG0 : LEAD - 1))); \
  TD_MFMA_PAIR(Q, T)                                                                                                                        \
  /* fragment T + 7 into the registers step T - 1 has consumed - NOT the ones this step's MFMAs were issued with: with two wavefronts */   \
  /* per SIMD an issued MFMA may still be queued for the matrix pipe when a DS read issued right behind it returns */                       \
  if constexpr ((T) + LEAD < 16) {                                                                                                          \
    TD_FRAG_READ(W[((T) + LEAD) & (WN - 1)], rbase[TD_RD_HI(Q)][TD_RD_PAR(Q, ((T) + LEAD) & 15)], TD_RD_OFF(Q, ((T) + LEAD) & 15))                 \
  } else {                                                                                                                                  \
    TD_FRAG_READ(W[((T) + LEAD) & (WN - 1)], rbase[TD_RD_HI(QN)][TD_RD_PAR(QN, ((T) + LEAD) & 15)], TD_RD_OFF(QN, ((T) + LEAD) & 15))              \
  }                                                                                                                                         \
  if constexpr ((T) == BP) {                                                                                                                \
    /* the pieces of stage Q + 1 (this wavefront's, then - behind the barrier - everybody's) have landed; the slot of stage Q - 1 is free */ \
    constexpr int CQ = 5 * ND + (Q == 0 ? 4 : Q == 1 ? 4 + U : Q == 2 ? 6 + U : Q <= 5 ? 8 + U : Q == 6 ? 6 + U : 4) + ((LAST && Q >= 4) ? NY2 : 0) + ((LAST && Q >= 6) ? 1 : 0); \
    if constexpr (!(TD_CHAIN_ABL & 2)) cwait_vm<CQ>();                                                                                       \
    if constexpr (!(TD_CHAIN_ABL & 8)) __builtin_amdgcn_s_barrier();                                                                         \
    issue_D(cic<(Q + 7) & 7>{}, Q == 0 ? c : cnext);                                                                                        \
  }                                                                                                                                         \
  side(Q_, cic<(T)>{}, LAST_, c, m0w);                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);
    TD_REP16(TD_STEP)
#undef TD_STEP
    // ---- stage end ----
#if !(TD_CHAIN_ABL & 1)
    if constexpr (Q >= 1 && Q <= 4) {
      asm volatile("ds_write_b128 %0, %1" ::"v"(sma[Q - 1]), "v"(oc[0][Q - 1]) : "memory");
      if constexpr (J > 1) asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(sma[Q - 1]), "v"(oc[J - 1][Q - 1]) : "memory");
    }
    if constexpr (Q <= 3) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(rr[0]) : "v"(sma[Q]) : "memory");
      if constexpr (J > 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(rr[J - 1]) : "v"(sma[Q]) : "memory");
    }
#endif
    if constexpr (Q == 0) {
      if constexpr (LAST) issue_rl(m0w_next, 0);
      else issue_rl(m0w, cnext);
    }
    if constexpr (Q <= 2) TD_B3(Q + 1, c)  // bias of K-step Q + 1: its epilogue runs under stage Q + 2
    if constexpr (Q == 7) TD_B3(0, cnext)
    __builtin_amdgcn_sched_barrier(0);
  };

  const int G = (int)gridDim.x;
  int tile = blockIdx.x;
  if (tile >= MT) return;  // (uniform)
  int m0w = tile * 128 + wave * PW;

  // ---- prologue: the steady state's in-flight set, drained once ----
  issue_y2(m0w);
  issue_rl(m0w, 0);
  TD_B3(0, 0) TD_B3(1, 0) TD_B3(2, 0) TD_B3(3, 0)
  issue_D(cic<0>{}, 0); issue_D(cic<1>{}, 0); issue_D(cic<2>{}, 0); issue_D(cic<3>{}, 0); issue_D(cic<4>{}, 0); issue_D(cic<5>{}, 0); issue_D(cic<6>{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#define TD_W0(F) if constexpr ((F) < LEAD) TD_FRAG_READ(W[(F) < LEAD ? (F) : 0], rbase[0][TD_RD_PAR(0, F)], TD_RD_OFF(0, F))
  TD_REP16(TD_W0)
#undef TD_W0
#if TD_CHAIN_ABL & 1
#pragma unroll
  for (int j = 0; j < J; ++j) asm volatile("" : "=v"(oc[j][0]), "=v"(oc[j][1]), "=v"(oc[j][2]), "=v"(oc[j][3]));
#endif

  auto chunk = [&](auto FIRST_, auto LAST_, int c, int m0w_next) {
    constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
    const int cnext = LAST ? 0 : c + 1;
    if constexpr (FIRST) {  // the y2 fragments of this tile: requested at the start of stage 4 of the previous tile's last chunk (behind them: D(4), 2 bias loads, D(5), 16 bias1 loads, D(6), D(7))
      cwait_vm<VM_Y2>();
    }
    stage(cic<0>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<1>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<2>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<3>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<4>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<5>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<6>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    stage(cic<7>{}, FIRST_, LAST_, c, cnext, m0w, m0w_next);
    if constexpr (LAST) {
      // ---- final epilogue: h1 = relu(acc1 + b1), straight from the MFMA layout (eight consecutive channels per lane; a tenth of the tile's bytes) ----
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      // bias1 has landed in the staging region (behind its piece: D(6), D(7), 2 bias loads); it is read from there a group ahead
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TD_VM(VM_B1)) : "memory");
      cu32x4 bb[2][2];
      const uint32_t ba = stg0 + (uint32_t)lg * 32u;
      asm volatile("ds_read_b128 %0, %1" : "=v"(bb[0][0]) : "v"(ba) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(bb[0][1]) : "v"(ba) : "memory");
#define TD_ACC1_READ(G, JJ)                                                                                                                \
  if constexpr ((JJ) < J) {                                                                                                                \
    constexpr int b0_ = A1 + 4 * (J * (2 * (G)) + ((JJ) < J ? (JJ) : 0)), b1_ = A1 + 4 * (J * (2 * (G) + 1) + ((JJ) < J ? (JJ) : 0));      \
    asm volatile("v_accvgpr_read_b32 %0, a%c8\n\tv_accvgpr_read_b32 %1, a%c9\n\tv_accvgpr_read_b32 %2, a%c10\n\tv_accvgpr_read_b32 %3, a%c11\n\t" \
                 "v_accvgpr_read_b32 %4, a%c12\n\tv_accvgpr_read_b32 %5, a%c13\n\tv_accvgpr_read_b32 %6, a%c14\n\tv_accvgpr_read_b32 %7, a%c15" \
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])                          \
                 : "n"(b0_), "n"(b0_ + 1), "n"(b0_ + 2), "n"(b0_ + 3), "n"(b1_), "n"(b1_ + 1), "n"(b1_ + 2), "n"(b1_ + 3));                \
  }
#define TD_FIN(G)                                                                                                                          \
  {                                                                                                                                        \
    if constexpr ((G) < 7) {                                                                                                               \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bb[((G) + 1) & 1][0]) : "v"(ba), "n"((((G) + 1) & 7) * 128) : "memory");          \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bb[((G) + 1) & 1][1]) : "v"(ba), "n"((((G) + 1) & 7) * 128 + 16) : "memory");     \
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bb[(G) & 1][0]), "+v"(bb[(G) & 1][1]) : "n"(TD_LGKM(2)) : "memory");                     \
    } else {                                                                                                                               \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[(G) & 1][0]), "+v"(bb[(G) & 1][1]) : : "memory");                                       \
    }                                                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < J; ++j) { /* channels 32 G + 8 lg .. + 8: fragments 2 G, 2 G + 1 */                              \
      const int m = m0w + 16 * j + lr;                                                                                                     \
      float v[8];                                                                                                                          \
      if (j == 0) { TD_ACC1_READ(G, 0) } else { TD_ACC1_READ(G, 1) }                                                                       \
      _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) v[4 * h + r] = v[4 * h + r] + __uint_as_float(bb[(G) & 1][h][r]);                    \
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));                                                                               \
      b2 a0 = {(__bf16)v[0], (__bf16)v[1]}, a1 = {(__bf16)v[2], (__bf16)v[3]}, a2 = {(__bf16)v[4], (__bf16)v[5]}, a3 = {(__bf16)v[6], (__bf16)v[7]}; \
      cu32x4 o = {*(uint32_t*)&a0, *(uint32_t*)&a1, *(uint32_t*)&a2, *(uint32_t*)&a3};                                                    \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) asm("v_pk_max_i16 %0, %1, 0" : "=v"(o[r]) : "v"(o[r])); /* ReLU on the rounded pairs */ \
      const uint32_t off = m < p.M ? ((uint32_t)m * (uint32_t)P + (uint32_t)(32 * (G) + 8 * lg)) * ES : OOB;                               \
      __builtin_amdgcn_raw_buffer_store_b128(o, rs_h1, (int)off, 0, 0);                                                                    \
    }                                                                                                                                      \
  }
      TD_REP8(TD_FIN)
#undef TD_FIN
#undef TD_ACC1_READ
    }
  };

#pragma unroll 1
  for (;;) {
    const int m0w_next = (tile + G) * 128 + wave * PW;
    chunk(std::true_type{}, std::false_type{}, 0, m0w_next);
#pragma unroll 1
    for (int c = 1; c < NCH - 1; ++c) chunk(std::false_type{}, std::false_type{}, c, m0w_next);
    chunk(std::false_type{}, std::true_type{}, NCH - 1, m0w_next);
    tile += G;
    if (tile >= MT) break;
    m0w = m0w_next;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing of the run-ahead may outlive the workgroup's LDS / registers
}

// (a plain kernel around the template body: with template-dependent launch bounds the host-side stub of a kernel template was not emitted)
__global__ __launch_bounds__(256, 1) void pw_chain2_kernel_w4(ChainParams p, int MT) { pw_chain2_body<256, 4>(p, MT); }

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
  pw_chain2_kernel_w4<<<dim3(std::min(n_cu, MT)), 256, 0, st>>>(p, MT);
  if (prof) prof_end(st);
  return check_launch("td_pw_chain2");
}
