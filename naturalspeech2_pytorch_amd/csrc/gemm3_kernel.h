// Lean kernel for the MIXED linear products of the step (round 6): every nn.Linear / 1x1 conv that multiplies FMT_H8 operands on full
// 256-row tiles -- q | k | v (NS2:1051-1053), the attention out-projection (1069), FF-in + GEGLU (1021-1024), FF-out (1024), the
// Wavenet's skip-sum and final 1x1 convs (NS2:639-640, 725), to_pred (NS2:783) -- 5.9 ms of the 16.3 ms step on gemm2_kernel<2, *> at
// 0.165-0.19 of the 16-bit MFMA peak, whose steady-state loop (run_k8) issued 327 instructions per 24 MFMAs.
//
// Same recipe as ffconv_kernel.h: the schedule is gemm2.hip's phased loop unchanged -- a K tile of 32 logical columns (one 128-B
// FMT_H8 line per row) consumed in four accumulator-quadrant phases, half tiles A0 / A1 / B0 / B1 requested one per phase five phases
// ahead, counted vmcnt, raw barriers, the two wave groups one barrier apart -- and everything an address depends on is fixed before
// the loop:
//   * W is read as pre-tiled LDS images (gemm3_tile_kernel): [column tile][K tile][32 KiB], rows in the order the waves read them,
//     chunks swizzled: a wave's two pieces of a W half tile are `global_load_lds_dwordx4 v, s[base] offset:0 / 1024`;
//   * A rows are whole 128-B lines at a fixed row stride: four per-lane source offsets + two scalar bases advanced by 128 B per tile;
//   * fragment reads are ds_read_b128 with immediate offsets off 2 x (4 + 4) address VGPRs (one set per LDS stage: the stage stride
//     of 64 KiB does not fit the offset field); M0 values sit in twelve SGPRs;
//   * the K loop is unrolled over the two stages; an odd tile count peels one steady tile in front (tile 0 then starts in stage 1),
//     so the last two tiles -- the ones with fewer requests and tighter waits -- are always (stage 0, stage 1).
// Per accumulator: half product of k chunk 0, of k chunk 1, then the fp8 correction-term MFMA -- mma_quadrant's order: results are
// bit-identical to gemm2_kernel<2, EPI, true>.  Epilogues: gemm2_epilogue.h, untouched.
// Not handled here (gemm2_kernel / gemm_kernel keep them): M % 256 != 0, conv taps, grid-z batching, split-K, the other arithmetics.
#pragma once
#include <algorithm>

#include "gemm2_epilogue.h"

namespace ns2 {
namespace mx3 {

constexpr int RB = 128;
constexpr int HALF = 16384;                   // one half tile (128 rows)
constexpr int REGION = 32768;                 // one operand of one K tile
constexpr int STAGE = 65536;                  // A at stage + 0, W at stage + 32768
constexpr int LDS_BYTES = 8 * EPI_LDS_WAVE_BYTES;
static_assert(2 * STAGE <= LDS_BYTES, "two stages fit the epilogue's LDS");

template <int IMM>
NS2_DEVINL void dma_s(unsigned voff, const unsigned char* sbase, unsigned m0s) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(m0s), "n"(IMM) : "memory");
}
template <int N> NS2_DEVINL void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
NS2_DEVINL void bar() { asm volatile("s_barrier" ::: "memory"); }

struct Ctx {
  f32x16 acc[4][2];
  bf16x8 A[2][4], W0[4], W1[4];            // [row tile of the pair][read]: reads 0, 1 = half k chunks, 2, 3 = the 32 fp8 bytes of the lane's k half
  unsigned vA[2][4], vW[2][4];             // fragment read addresses per stage
  unsigned voA[4];                         // DMA source offsets of this wave's A pieces: A0 e0, A0 e1, A1 e0, A1 e1 (bytes from the tile's base)
  unsigned voW0, voW1;                     // ... inside a W image: half 0 / half 1
  const unsigned char* sA0; const unsigned char* sA1;   // A base of the tile whose A0 (tile u + 2) / A1 (tile u + 1) half is requested next
  const unsigned char* sW0; const unsigned char* sW1;   // W image whose B0 (u + 2) / B1 (u + 1) half is requested next
  unsigned m0A[2][4], m0W[2][2];           // M0 per stage: this wave's A pieces / W halves (second piece of a W half: + 1024 = the immediate)
};

NS2_DEVINL bf16x8 lds16(unsigned addr, int imm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  return *reinterpret_cast<const bf16x8*>(smem + addr + imm);
}
template <int S, int a> NS2_DEVINL void load_a(Ctx& c) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    c.A[i][0] = lds16(c.vA[S][0], (2 * a + i) * 4096);
    c.A[i][1] = lds16(c.vA[S][1], (2 * a + i) * 4096);
    c.A[i][2] = lds16(c.vA[S][2], (2 * a + i) * 4096);
    c.A[i][3] = lds16(c.vA[S][3], (2 * a + i) * 4096);
  }
}
template <int S, int b> NS2_DEVINL void load_w(Ctx& c, bf16x8 (&W)[4]) {
  W[0] = lds16(c.vW[S][0], b * HALF);
  W[1] = lds16(c.vW[S][1], b * HALF);
  W[2] = lds16(c.vW[S][2], b * HALF);
  W[3] = lds16(c.vW[S][3], b * HALF);
}
template <int a, int b> NS2_DEVINL void mma_q(Ctx& c, const bf16x8 (&W)[4]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    c.acc[2 * a + i][b] = mma16<true>(c.A[i][0], W[0], c.acc[2 * a + i][b]);
    c.acc[2 * a + i][b] = mma16<true>(c.A[i][1], W[1], c.acc[2 * a + i][b]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {              // the correction terms: lanes 0-31 a_h8 . w_l8, lanes 32-63 a_l8 . w_h8 (ns2_common.h)
    const int4 a0 = __builtin_bit_cast(int4, c.A[i][2]), a1 = __builtin_bit_cast(int4, c.A[i][3]);
    const int4 w0 = __builtin_bit_cast(int4, W[2]), w1 = __builtin_bit_cast(int4, W[3]);
    const i32x8 a8 = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const i32x8 w8 = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    c.acc[2 * a + i][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, c.acc[2 * a + i][b], /*A e5m2*/ 1, /*B e5m2*/ 1, 0, H8_E8M0_LO, 0,
                                                                         H8_E8M0_ONE);
  }
  __builtin_amdgcn_s_setprio(0);
}
template <int S, int h> NS2_DEVINL void issue_a(Ctx& c) {            // this wave's two pieces of half tile A<h> into stage S
  const unsigned char* base = h ? c.sA1 : c.sA0;
  dma_s<0>(c.voA[2 * h], base, c.m0A[S][2 * h]);
  dma_s<0>(c.voA[2 * h + 1], base, c.m0A[S][2 * h + 1]);
}
template <int S, int b> NS2_DEVINL void issue_w(Ctx& c) {
  const unsigned char* base = b ? c.sW1 : c.sW0;
  dma_s<0>(b ? c.voW1 : c.voW0, base, c.m0W[S][b]);
  dma_s<1024>(b ? c.voW1 : c.voW0, base, c.m0W[S][b]);
}

// One K tile u in stage S.  R1: tile u + 1 exists (its B1 / A1 halves are requested here, into the other stage), R2: tile u + 2 exists
// (A0 / B0, into this stage); R1P / R2P: the same for tile u - 1 (what is still in flight when this tile's waits are counted).
// Reads: A0, B0 in phase 0, B1 in phase 1, A1 in phase 2.  Waits (pieces younger than the half tile the NEXT phase reads):
//   phase 0 (B1(u), requested in phase 0 of u - 1): A1(u) | A0, B0(u + 1) | B1(u + 1)
//   phase 1 (A1(u), phase 1 of u - 1):              A0, B0(u + 1) | B1, A1(u + 1)
//   phase 3 (A0, B0(u + 1), phases 2 / 3 of u - 1): B1, A1(u + 1) | A0, B0(u + 2)
template <int S, bool R1, bool R2, bool R1P, bool R2P> NS2_DEVINL void tile(Ctx& c) {
  // ---- phase 0: quadrant (0, 0)
  load_a<S, 0>(c); load_w<S, 0>(c, c.W0);
  if constexpr (R1) issue_w<S ^ 1, 1>(c);
  vmwait<2 * ((R1P ? 1 : 0) + (R2P ? 2 : 0) + (R1 ? 1 : 0))>();
  bar();
  mma_q<0, 0>(c, c.W0);
  bar();
  // ---- phase 1: quadrant (0, 1)
  load_w<S, 1>(c, c.W1);
  if constexpr (R1) issue_a<S ^ 1, 1>(c);
  vmwait<2 * ((R2P ? 2 : 0) + (R1 ? 2 : 0))>();
  bar();
  mma_q<0, 1>(c, c.W1);
  bar();
  // ---- phase 2: quadrant (1, 1); the A0 rows of this stage were last read in phase 0
  load_a<S, 1>(c);
  if constexpr (R2) issue_a<S, 0>(c);
  bar();
  mma_q<1, 1>(c, c.W1);
  bar();
  // ---- phase 3: quadrant (1, 0)
  if constexpr (R2) issue_w<S, 0>(c);
  vmwait<(R1 ? 4 : 0) + (R2 ? 4 : 0)>();
  bar();
  mma_q<1, 0>(c, c.W0);
  bar();
  c.sA0 += RB; c.sA1 += RB; c.sW0 += REGION; c.sW1 += REGION;
}

// addresses, offsets and M0 values of one lean K loop: A = FMT_H8 rows of `lda_b` bytes starting at a_base (the row tile's first row, K
// offset 0), W = the tiled images starting at w_base
NS2_DEVINL void setup(Ctx& c, const unsigned char* a_base, const unsigned char* w_base, const long lda_b, const int wave, const int lane) {
  const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, hi = lane >> 5;
  // fragment reads (gemm2.hip compute_tile NS == 2): half k chunk kc of lane (l31, hi) = 16-B chunk 2 kc + hi of the line; fp8 bytes: A chunks
  // 4 + 2 hi, 5 + 2 hi ([h8 | l8][hi]), W chunks 6 - 2 hi, 7 - 2 hi; chunk q of LDS row r sits at position q ^ ((r >> 1) & 7)
  {
    const int sw = (l31 >> 1) & 7;
    const int qa[4] = {hi, 2 + hi, 4 + 2 * hi, 5 + 2 * hi}, qw[4] = {hi, 2 + hi, 6 - 2 * hi, 7 - 2 * hi};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c.vA[s][j] = s * STAGE + (wm * 128 + l31) * RB + ((qa[j] ^ sw) << 4);
        c.vW[s][j] = s * STAGE + REGION + wn * 4096 + l31 * RB + ((qw[j] ^ sw) << 4);
      }
  }
  // DMA pieces (8 rows x 128 B).  A half tile A<h> = the tile rows with bit 6 == h (row tiles 2h, 2h + 1 of both row halves): 16 row groups,
  // this wave's two: k = 2 wave + e -> rg = (k & 7) + 16 (k >> 3) + 8 h.  The LDS image of a piece is lane-linear: position p of row r is
  // fetched from chunk p ^ swz(r).
  const int lrow = lane >> 3, pch = lane & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = 2 * wave + e, rg = (k & 7) + 16 * (k >> 3) + 8 * h, row = 8 * rg + lrow;
      c.voA[2 * h + e] = (unsigned)(row * lda_b) + ((pch ^ ((row >> 1) & 7)) << 4);
#pragma unroll
      for (int s = 0; s < 2; ++s) c.m0A[s][2 * h + e] = s * STAGE + rg * 1024;
    }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int b = 0; b < 2; ++b) c.m0W[s][b] = s * STAGE + REGION + b * HALF + wave * 2048;
  c.voW0 = lane * 16 + wave * 2048;
  c.voW1 = c.voW0 + HALF;
  c.sA0 = a_base; c.sA1 = a_base; c.sW0 = w_base; c.sW1 = w_base;
}

// the K loop over T >= 2 tiles: prologue (tile 0 whole, A0 / B0 of tile 1; with an odd tile count tile 0 starts in stage 1, see the
// header), steady tiles, the last two.  Entered and left with the two wave groups in step (every wave past every LDS read on return).
NS2_DEVINL void kloop(Ctx& c, const int T, const int wave) {
  if (T & 1) { issue_a<1, 0>(c); issue_w<1, 0>(c); issue_w<1, 1>(c); issue_a<1, 1>(c); }
  else { issue_a<0, 0>(c); issue_w<0, 0>(c); issue_w<0, 1>(c); issue_a<0, 1>(c); }
  c.sA0 += RB; c.sW0 += REGION;
  if (T & 1) { issue_a<0, 0>(c); issue_w<0, 0>(c); }
  else { issue_a<1, 0>(c); issue_w<1, 0>(c); }
  c.sA0 += RB; c.sW0 += REGION; c.sA1 += RB; c.sW1 += REGION;       // A0 / B0 requests continue at tile 2, A1 / B1 at tile 1
  vmwait<8>();
  bar();
  if (wave >= 4) bar();                                         // the second group runs half a phase behind
  if (T & 1) tile<1, true, true, true, true>(c);                // (an odd count needs T >= 3: checked by the launchers)
  for (int i = (T - 2) >> 1; i > 0; --i) { tile<0, true, true, true, true>(c); tile<1, true, true, true, true>(c); }
  tile<0, true, false, true, true>(c);
  tile<1, false, false, true, false>(c);
  if (wave < 4) bar();                                          // ... and the first waits for its last MFMA phase
}

template <int EPI>
struct Kern {
  static NS2_DEVINL void run(const GemmArgs& g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = (g.N + G2_BN - 1) / G2_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = __builtin_amdgcn_readfirstlane(bid % ntn), tm = __builtin_amdgcn_readfirstlane(bid / ntn);
    const int T = g.nkt;
    const long lda_b = 4L * g.lda;                               // FMT_H8 lines: 4 bytes per logical column
    Ctx c;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) c.acc[mi][ni][r] = 0.f;
    setup(c, reinterpret_cast<const unsigned char*>(g.a_hi) + (long)tm * G2_BM * lda_b,
          reinterpret_cast<const unsigned char*>(g.w_tl) + (long)tn * T * REGION, lda_b, wave, lane);
    kloop(c, T, wave);
    g2_block_epilogue<2, EPI, true>(c.acc, g, 0, tm, tn, wave, lane, smem);
  }
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(const GemmArgs g) { Kern<EPI>::run(g); }

// ---- weights: the row-major FMT_H8 pack [rows_p][ldk / 32 lines of 128 B] -> tiled LDS images.  Image of (column tile tn, K tile t): row n
// of the tile lives in half b = (n >> 5) & 1, quarter wn = n >> 6, row r = n & 31 at byte b * 16384 + wn * 4096 + r * 128, chunk q of its
// line at position q ^ ((r >> 1) & 7).  A permutation of the packed bytes.  One thread per 16-byte chunk.
__global__ void gemm3_tile_kernel(const unsigned char* __restrict__ w, long row_bytes, int T, int ntn, uint4* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)ntn * T * (REGION / 16);
  if (idx >= total) return;
  const int ch = (int)(idx % (REGION / 16));
  const long img = idx / (REGION / 16);
  const int t = (int)(img % T), tn = (int)(img / T);
  const int byte = ch * 16;
  const int b = byte >> 14, wn = (byte >> 12) & 3, r = (byte >> 7) & 31, pos = (byte >> 4) & 7;
  const int q = pos ^ ((r >> 1) & 7);
  const long row = (long)tn * 256 + wn * 64 + b * 32 + r;                 // < rows_p = round_up(N, 256)
  out[idx] = *reinterpret_cast<const uint4*>(w + row * row_bytes + (long)t * RB + q * 16);
}

}  // namespace mx3

inline size_t gemm3_tiled_bytes(int rows_p, int nkt) { return (size_t)(rows_p / 256) * nkt * mx3::REGION; }
inline hipError_t launch_gemm3_tile(const bf16_t* w_hi, int ldk, int rows_p, bf16_t* out, hipStream_t s) {
  if (!w_hi || !out || (ldk & 31) || (rows_p & 255) || rows_p <= 0) return hipErrorInvalidValue;
  const int T = ldk / 32, ntn = rows_p / 256;
  const long total = (long)ntn * T * (mx3::REGION / 16);
  hipLaunchKernelGGL(mx3::gemm3_tile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned char*>(w_hi),
                     4L * ldk, T, ntn, reinterpret_cast<uint4*>(out));
  return hipGetLastError();
}

// Does this product take the lean kernel?  (precision 4 = FMT_H8 operands, plain linear, full row tiles, one launch)
inline bool gemm3_eligible(const GemmArgs& g, int precision) {
  if (!g.w_tl || precision != 4 || g.conv_taps != 0 || g.nkt != g.kt_per_tap || g.nkt < 3 || g.nz > 1 || g.dil_z || g.ksplit != 0) return false;
  if (g.epi != EPI_F32 && g.epi != EPI_SPLIT && g.epi != EPI_QKV && g.epi != EPI_GEGLU) return false;
  if (g.M <= 0 || (g.M & 255) || g.lda < g.nkt * 32 || g.ldw != g.nkt * 32 || g.nrm_hi) return false;
  return g.a_lo == g.a_hi + 32 && (reinterpret_cast<uintptr_t>(g.a_hi) & 15) == 0;
}

inline hipError_t launch_gemm3(const GemmArgs& g, hipStream_t s) {
  const int ntn = (g.N + G2_BN - 1) / G2_BN, grid = ntn * (g.M / G2_BM);
  static DynLdsAttr attr[4];
  const void* fn; int slot;
  switch (g.epi) {
    case EPI_F32: fn = reinterpret_cast<const void*>(&mx3::gemm3_kernel<EPI_F32>); slot = 0; break;
    case EPI_SPLIT: fn = reinterpret_cast<const void*>(&mx3::gemm3_kernel<EPI_SPLIT>); slot = 1; break;
    case EPI_QKV: fn = reinterpret_cast<const void*>(&mx3::gemm3_kernel<EPI_QKV>); slot = 2; break;
    case EPI_GEGLU: fn = reinterpret_cast<const void*>(&mx3::gemm3_kernel<EPI_GEGLU>); slot = 3; break;
    default: return hipErrorInvalidValue;
  }
  hipError_t e = attr[slot].ensure(fn, mx3::LDS_BYTES);
  if (e != hipSuccess) return e;
  switch (g.epi) {
    case EPI_F32: hipLaunchKernelGGL(mx3::gemm3_kernel<EPI_F32>, dim3(grid), dim3(512), mx3::LDS_BYTES, s, g); break;
    case EPI_SPLIT: hipLaunchKernelGGL(mx3::gemm3_kernel<EPI_SPLIT>, dim3(grid), dim3(512), mx3::LDS_BYTES, s, g); break;
    case EPI_QKV: hipLaunchKernelGGL(mx3::gemm3_kernel<EPI_QKV>, dim3(grid), dim3(512), mx3::LDS_BYTES, s, g); break;
    default: hipLaunchKernelGGL(mx3::gemm3_kernel<EPI_GEGLU>, dim3(grid), dim3(512), mx3::LDS_BYTES, s, g); break;
  }
  return hipGetLastError();
}

}  // namespace ns2
