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

template <int P>
__global__ __launch_bounds__(256, 1) void pw_chain2_kernel(ChainParams p, int MT) {
  static_assert(P == 256, "layer3 geometry: four 32-row stages of conv3, two K tiles x two row halves of the next conv1 per chunk");
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

  // ---- weight pieces.  A stage is 128 LDS rows of 128 B; piece k of this wavefront fills rows 8 (4 k + wave) .. + 8; LDS slot (lane & 7) of a
  // row takes source chunk (lane & 7) ^ (row & 7).  conv3 stage q of a chunk = weight rows 32 q .. + 32 over ALL of K: LDS row 32 kt + r holds K
  // tile kt of weight row 32 q + tau(r), tau(16 i2 + a) = 8 (a >> 2) + 4 i2 + (a & 3).  conv1 stage (kt, nh) = K tile kt of the chunk for weight
  // rows 128 nh + sigma(rho), sigma(16 i + a) = 32 (i >> 1) + 8 (a >> 2) + 4 (i & 1) + (a & 3). ----
  uint32_t v3[4], v1[4];
  {
    const int drow = lane >> 3, dch = (lane & 7) ^ drow;
    const int r32 = 8 * wave + drow, i2 = r32 >> 4, a3 = r32 & 15;
    const int tau = 8 * (a3 >> 2) + 4 * i2 + (a3 & 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v3[k] = (uint32_t)(tau * P + dch * 8) * ES + (uint32_t)k * 128u;
      const int rho = 8 * (4 * k + wave) + drow, i = rho >> 4, a = rho & 15;
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
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (clds_t)(dst + (4 * k + wave) * 1024), 16, v3[k], soff, 0, 0);
    } else {
      constexpr int kt = (PQ - 4) >> 1, nh = (PQ - 4) & 1;
      const int soff = nh * (128 * N3 * ES) + cn * 256 + kt * 128;
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (clds_t)(dst + (4 * k + wave) * 1024), 16, v1[k], soff, 0, 0);
    }
  };

  // ---- fragment reads: LDS row + lr, 16-byte chunk (par * 4 + lg) ^ (lr & 7); the DS offset field holds 16 bits: slots 4..7 get their own base ----
  const uint32_t ring0 = (uint32_t)(uintptr_t)(clds_t)ring;
  const uint32_t rb0 = ring0 + (uint32_t)(lr * 128 + ((lg ^ (lr & 7)) << 4));
  const uint32_t rbase[2][2] = {{rb0, rb0 ^ 64u}, {rb0 + 65536u, (rb0 ^ 64u) + 65536u}};  // [slot >= 4][k-step parity]

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
  cu32x4 rl[8];                   // residual rows of the NEXT chunk, row-major (pass u), in flight for a whole chunk
  cu32x4 rr[2], ot[8];
  f32x4 b3r[4][2], b1r[8][2];

  // Loads address rows past M as they come: the buffer descriptors end at the tensors' last byte, out-of-range loads return zeros.
  auto issue_y2 = [&](int m0w) {  // 16 loads
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t vo = (uint32_t)(m0w + 16 * j + lr) * (uint32_t)(P * ES) + (uint32_t)lg * 16u;
#define TD_Y2(KS) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=a"(y2f[j][KS]) : "v"(vo), "s"(rs_y2), "n"(KS * 64) : "memory");
      TD_REP8(TD_Y2)
#undef TD_Y2
    }
  };
  auto issue_rl = [&](int m0w, int cn) {  // 8 loads: pass u = rows 4 u + lg, 16 lanes x 16 B = the chunk's 256 B of a row
    const uint32_t vo = (TD_CHAIN_ABL & (16 | 256)) ? OOB : (uint32_t)(m0w + lg) * (uint32_t)(N3 * ES) + (uint32_t)lr * 16u;
#define TD_RL(U)                                                                                                                                  \
  {                                                                                                                                               \
    const int soff = cn * 256 + (U) * (4 * N3 * ES);                                                                                              \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rl[U]) : "v"(vo), "s"(rs_res), "s"(soff) : "memory");                          \
  }
    TD_REP8(TD_RL)
#undef TD_RL
  };
  auto issue_b1 = [&]() {  // 16 loads
    const uint32_t vo = (uint32_t)lg * 32u;
#define TD_B1(S)                                                                                                                             \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(b1r[S][0]) : "v"(vo), "s"(rs_b1), "n"(S * 128) : "memory");      \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(b1r[S][1]) : "v"(vo), "s"(rs_b1), "n"(S * 128 + 16) : "memory");
    TD_REP8(TD_B1)
#undef TD_B1
  };
#define TD_B3(S, cn)                                                                                                                                     \
  {                                                                                                                                                      \
    const uint32_t vo_ = (uint32_t)lg * 32u;                                                                                                             \
    const int so_ = (cn) * 512;                                                                                                                          \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][0]) : "v"(vo_), "s"(rs_b3), "s"(so_), "n"((S) * 128) : "memory");     \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(b3r[S][1]) : "v"(vo_), "s"(rs_b3), "s"(so_), "n"((S) * 128 + 16) : "memory"); \
  }

  // ---- one stage = 16 steps: step T waits for fragment T (in W[T & 7]), issues its two MFMAs and requests fragment T + 8 into the same
  // registers (steps 8..15: the next stage's first eight).  DS operations return in order: lgkmcnt(n) with n = the DS operations issued behind
  // the wanted read.  Phase A (Q < 4), fragment T = (k-step T >> 1, row fragment T & 1) of weight rows 32 Q ..: LDS row 32 (T >> 2) + 16 (T & 1),
  // parity (T >> 1) & 1.  Phase B, fragment T = (k-step T >> 3, row fragment T & 7). ----
#define TD_RD_HI(Q) ((Q) >= 4)
#define TD_RD_PAR(Q, T) ((Q) < 4 ? (((T) >> 1) & 1) : ((T) >> 3))
#define TD_RD_OFF(Q, T) (((Q) & 3) * SLOT + ((Q) < 4 ? ((T) >> 2) * 4096 + ((T) & 1) * 2048 : ((T) & 7) * 2048))
#define TD_MFMA_PAIR_(Q, T)                                                                                                                \
  if constexpr ((Q) < 4) {                                                                                                                 \
    if constexpr (((T) >> 1) == 0) {                                                                                                       \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc3[((Q) < 4 ? 2 * (Q) + ((T) & 1) : 0)][0]) : "v"(W[(T) & 7]), "a"(y2f[0][(T) >> 1])); \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc3[((Q) < 4 ? 2 * (Q) + ((T) & 1) : 0)][1]) : "v"(W[(T) & 7]), "a"(y2f[1][(T) >> 1])); \
    } else {                                                                                                                               \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc3[((Q) < 4 ? 2 * (Q) + ((T) & 1) : 0)][0]) : "v"(W[(T) & 7]), "a"(y2f[0][(T) >> 1])); \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc3[((Q) < 4 ? 2 * (Q) + ((T) & 1) : 0)][1]) : "v"(W[(T) & 7]), "a"(y2f[1][(T) >> 1])); \
    }                                                                                                                                      \
  } else if constexpr (FIRST && (Q) <= 5 && ((T) >> 3) == 0) { /* a tile's first touch of this accumulator: C = 0 */                        \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + ((T) & 7) : 0)][0]) : "v"(W[(T) & 7]), "v"(oc[0][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + ((T) >> 3) : 0)])); \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + ((T) & 7) : 0)][1]) : "v"(W[(T) & 7]), "v"(oc[1][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + ((T) >> 3) : 0)])); \
  } else {                                                                                                                                 \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + ((T) & 7) : 0)][0]) : "v"(W[(T) & 7]), "v"(oc[0][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + ((T) >> 3) : 0)])); \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc1[((Q) >= 4 ? 8 * (((Q) - 4) & 1) + ((T) & 7) : 0)][1]) : "v"(W[(T) & 7]), "v"(oc[1][((Q) >= 4 ? 2 * (((Q) - 4) >> 1) + ((T) >> 3) : 0)])); \
  }
#if TD_CHAIN_ABL & 32
#define TD_MFMA_PAIR(Q, T) asm volatile("" : "+v"(W[(T) & 7]));
#else
#define TD_MFMA_PAIR(Q, T) TD_MFMA_PAIR_(Q, T)
#endif
#if TD_CHAIN_ABL & 4
#define TD_FRAG_READ(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr));
#else
#define TD_FRAG_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
#endif
#define TD_LGKM(n) (TD_CHAIN_SAFE ? 0 : ((n) > 15 ? 15 : (n)))
#define TD_VM(n) (TD_CHAIN_SAFE ? 0 : ((n) > 63 ? 63 : (n)))

  // side work of step T of stage Q (everything that is not the weight stream): the chunk's epilogue in slices under the MFMAs
  auto side = [&](auto Q_, auto T_, auto LAST_, int c, int m0w) {
    constexpr int Q = decltype(Q_)::value, T = decltype(T_)::value;
#if !(TD_CHAIN_ABL & 1)
    if constexpr (Q >= 1 && Q <= 4) {
      constexpr int S = Q - 1;  // epilogue of K-step S of phase B = fragments 2 S, 2 S + 1 of phase A, final since the end of stage S
      if constexpr (T == 1) {
        // the residual fragments (requested at the end of the previous stage; behind them: the W reads of steps 0, 1) and this K-step's bias
        // (requested at the end of stage S + 1 of the previous chunk) are here
        asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(%5)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(b3r[S][0]), "+v"(b3r[S][1]) : "n"(TD_VM(42)), "n"(TD_LGKM(2)) : "memory");
      }
      if constexpr (T >= 2 && T <= 9 && !(TD_CHAIN_ABL & 128)) {
        constexpr int K = T - 2, J = K >> 2, PP = K & 3, H = PP >> 1, R0 = 2 * (PP & 1);
        const uint32_t wd = rr[J][PP];
        float x0 = acc3[2 * S + H][J][R0] + b3r[S][H][R0];
        float x1 = acc3[2 * S + H][J][R0 + 1] + b3r[S][H][R0 + 1];
        x0 = fmaxf(x0 + __uint_as_float(wd << 16), 0.f);
        x1 = fmaxf(x1 + __uint_as_float(wd & 0xffff0000u), 0.f);
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 pk = {(__bf16)x0, (__bf16)x1};
        oc[J][S][PP] = *(const uint32_t*)&pk;
      }
    }
    if constexpr (Q == 5) {
      if constexpr (T == 1) {
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(ot[0]), "+v"(ot[1]), "+v"(ot[2]), "+v"(ot[3]), "+v"(ot[4]), "+v"(ot[5]), "+v"(ot[6]), "+v"(ot[7]) : "n"(TD_LGKM(2)) : "memory");
      }
      if constexpr (T >= 2 && T <= 9) {
        constexpr int U = T - 2;
        const int m = m0w + 4 * U + lg;
        const uint32_t off = (m < p.M && !(TD_CHAIN_ABL & (16 | 64))) ? ((uint32_t)m * (uint32_t)N3 + (uint32_t)(c * 128 + lr * 8)) * ES : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(ot[U], rs_out, (int)off, 0, (TD_CHAIN_NT & 2) ? 2 : 0);
      }
    }
#endif
  };

  auto stage = [&](auto Q_, auto FIRST_, auto LAST_, int c, int cnext, int m0w, int m0w_next) {
    constexpr int Q = decltype(Q_)::value;
    constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
    constexpr int QN = (Q + 1) & 7;
    // DS operations issued at the end of stage Q - 1 and at the start of stage Q: they sit behind the reads of fragments 0..7
    constexpr int NE_PREV = (TD_CHAIN_ABL & 1) ? 0 : (Q == 1 ? 2 : (Q >= 2 && Q <= 4) ? 4 : (Q == 5 ? 2 : 0));
    constexpr int NS = (TD_CHAIN_ABL & 1) ? 0 : ((Q == 0 || Q == 5) ? 8 : 0);
    constexpr int LG0 = 7 + NE_PREV + NS;
    // ---- stage start ----
#if !(TD_CHAIN_ABL & 1)
    if constexpr (Q == 0) {
      // the residual rows of this chunk (requested at the end of stage 0 of the previous chunk) go to the staging region, row-major
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(rl[0]), "+v"(rl[1]), "+v"(rl[2]), "+v"(rl[3]), "+v"(rl[4]), "+v"(rl[5]), "+v"(rl[6]), "+v"(rl[7]) : "n"(TD_VM(36)) : "memory");
#define TD_RW(U) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sra[(U) & 3]), "v"(rl[U]), "n"(((U) >> 2) * 4096) : "memory");
      TD_REP8(TD_RW)
#undef TD_RW
    }
    if constexpr (Q == 5) {
#define TD_OR(U) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ot[U]) : "v"(sra[(U) & 3]), "n"(((U) >> 2) * 4096) : "memory");
      TD_REP8(TD_OR)
#undef TD_OR
    }
#endif
    if constexpr (Q == 4 && LAST) issue_y2(m0w_next);  // (AGPR destinations, dead since the end of stage 3)
    if constexpr (Q == 6 && LAST) issue_b1();          // (into the registers the output rows have left)
    __builtin_amdgcn_sched_barrier(0);
#define TD_STEP(T)                                                                                                                           \
  if constexpr (!(TD_CHAIN_ABL & 1024)) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(W[(T) & 7]) : "n"(TD_LGKM((T) < 8 ? LG0 : 7)));        \
  TD_MFMA_PAIR(Q, T)                                                                                                                        \
  if constexpr ((T) < 8) {                                                                                                                  \
    TD_FRAG_READ(W[(T) & 7], rbase[TD_RD_HI(Q)][TD_RD_PAR(Q, (T) + 8)], TD_RD_OFF(Q, (T) + 8))                                              \
  } else {                                                                                                                                  \
    TD_FRAG_READ(W[(T) & 7], rbase[TD_RD_HI(QN)][TD_RD_PAR(QN, (T) - 8)], TD_RD_OFF(QN, (T) - 8))                                           \
  }                                                                                                                                         \
  if constexpr ((T) == 7) {                                                                                                                 \
    /* the pieces of stage Q + 1 (this wavefront's, then - behind the barrier - everybody's) have landed; the slot of stage Q - 1 is free */ \
    constexpr int CQ = (Q == 0 ? 26 : Q <= 3 ? 32 : Q == 4 ? 34 : Q <= 6 ? 36 : 28) + ((LAST && Q >= 4) ? 16 : 0) + ((LAST && Q >= 6) ? 16 : 0); \
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
      asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(sma[Q - 1]), "v"(oc[1][Q - 1]) : "memory");
    }
    if constexpr (Q <= 3) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(rr[0]) : "v"(sma[Q]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(rr[1]) : "v"(sma[Q]) : "memory");
    }
#endif
    if constexpr (Q == 0) {
      if constexpr (LAST) issue_rl(m0w_next, 0);
      else issue_rl(m0w, cnext);
    }
    if constexpr (Q >= 1 && Q <= 4) TD_B3(Q - 1, cnext)
    __builtin_amdgcn_sched_barrier(0);
  };

  const int G = (int)gridDim.x;
  int tile = blockIdx.x;
  if (tile >= MT) return;  // (uniform)
  int m0w = tile * 128 + wave * 32;

  // ---- prologue: the steady state's in-flight set, drained once ----
  issue_y2(m0w);
  issue_rl(m0w, 0);
  TD_B3(0, 0) TD_B3(1, 0) TD_B3(2, 0) TD_B3(3, 0)
  issue_D(cic<0>{}, 0); issue_D(cic<1>{}, 0); issue_D(cic<2>{}, 0); issue_D(cic<3>{}, 0); issue_D(cic<4>{}, 0); issue_D(cic<5>{}, 0); issue_D(cic<6>{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#define TD_W0(F) TD_FRAG_READ(W[F], rbase[0][TD_RD_PAR(0, F)], TD_RD_OFF(0, F))
  TD_REP8(TD_W0)
#undef TD_W0
#if TD_CHAIN_ABL & 1
  asm volatile("" : "=v"(oc[0][0]), "=v"(oc[0][1]), "=v"(oc[0][2]), "=v"(oc[0][3]), "=v"(oc[1][0]), "=v"(oc[1][1]), "=v"(oc[1][2]), "=v"(oc[1][3]));
#endif

  auto chunk = [&](auto FIRST_, auto LAST_, int c, int m0w_next) {
    constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
    const int cnext = LAST ? 0 : c + 1;
    if constexpr (FIRST) {  // (uniform) the y2 fragments of this tile: requested at the start of stage 4 of the previous tile's last chunk (behind them: D(4), 2 bias loads, D(5), 16 bias1 loads, D(6), D(7))
      cwait_vm<34>();
#define TD_T(KS) asm volatile("" : "+a"(y2f[0][KS]), "+a"(y2f[1][KS]));
      TD_REP8(TD_T)
#undef TD_T
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
#define TD_T(N) asm volatile("" : "+a"(acc1[N][0]), "+a"(acc1[N][1]));
      TD_REP16(TD_T)
#undef TD_T
      // bias1 landed (younger: D(6), D(7))
      asm volatile("s_waitcnt vmcnt(%16)"
                   : "+v"(b1r[0][0]), "+v"(b1r[0][1]), "+v"(b1r[1][0]), "+v"(b1r[1][1]), "+v"(b1r[2][0]), "+v"(b1r[2][1]), "+v"(b1r[3][0]), "+v"(b1r[3][1]),
                     "+v"(b1r[4][0]), "+v"(b1r[4][1]), "+v"(b1r[5][0]), "+v"(b1r[5][1]), "+v"(b1r[6][0]), "+v"(b1r[6][1]), "+v"(b1r[7][0]), "+v"(b1r[7][1])
                   : "n"(TD_VM(8))
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
    }
  };

#pragma unroll 1
  for (;;) {
    const int m0w_next = (tile + G) * 128 + wave * 32;
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
