// The WavenetResBlock of the hybrid plan as a lean kernel (round 6; NS2:597-642): per column z of a stack, one launch of
//   h = DilConv3(x) + b ; h = h * gamma_t + beta_t ; h = tanh(h) * sigmoid(h) ; out = h + Conv1x1(x) + b_res
// with the dilated k = 3 conv as ONE IEEE-half product on the half parts of the FMT_H8 operand lines and the residual 1x1 conv in the
// mixed arithmetic (model precision 5) -- what gemm2_kernel<2, EPI_WAVENET, true, P1 = 1> computes, 3.6 ms of the
// step at 0.245 of the 16-bit MFMA peak: its one-barrier loop spends ~30 scalar / vector instructions per MFMA on tap arithmetic, zero-
// page selects and K-tile coordinates.  Here (the recipe of ffconv_kernel.h / gemm3_kernel.h):
//   * both K phases run gemm3's phased loop on pre-tiled weight images ([z][column tile][K tile][32 KiB]: ns2_wavenet tiles below);
//     phase 1 visits (64-column chunk, tap) tiles -- tap-minor, as before -- whose LDS rows are gathered from the half parts of two
//     lines per row; phase 2 = gemm3's tile() unchanged on the unshifted rows;
//   * the three taps of a chunk read the rows m0 - (2 - tap) dil ...: one scalar base per tap (advanced by 256 B per chunk) and twelve
//     per-lane source offsets fixed before the loop.  Rows in front of the utterance exist only in the FIRST row tile of an utterance:
//     there the offsets are clamped to rows that exist and the fragments of those rows are zeroed after their LDS read (a wave-uniform
//     branch; 0 x w = 0 exactly) -- no zero page, no per-lane test at issue time;
//   * the K loops are unrolled so that stage parity AND tap are immediates (6 tile bodies per iteration).
// Summation order per accumulator = the old kernel's (chunk-major, tap-minor, k chunks in order; then the res-conv tiles): results are
// bit-identical.  Mid-gate and epilogue: gemm_epi.h / gemm2_epilogue.h, untouched.
#pragma once
#include "gemm3_kernel.h"

namespace ns2 {
namespace wn3 {

using mx3::bar; using mx3::dma_s; using mx3::vmwait;
constexpr int RB = mx3::RB, HALF = mx3::HALF, REGION = mx3::REGION, STAGE = mx3::STAGE;

struct Ctx : mx3::Ctx {
  unsigned vA1[2][4], vW1[2][4];           // phase 1: fragment read addresses per stage (k chunk kc of a 64-deep row: 16-B chunk 2 kc + hi)
  unsigned voT[3][4];                      // phase 1: DMA source offsets of this wave's A pieces per tap (bytes from the tap's base)
  const unsigned char* sT[3];              // phase 1: A base per tap = (row m0 - 2 dil, the chunk that tap requests next)
  unsigned zbits;                          // first row tile of an utterance: bit 4 t + mi = this lane's row of row tile mi lies in front of the utterance for tap t (0 / 1)
  bool first;
};

template <int S, int a> NS2_DEVINL void load_a1(Ctx& c) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    c.A[i][0] = mx3::lds16(c.vA1[S][0], (2 * a + i) * 4096);
    c.A[i][1] = mx3::lds16(c.vA1[S][1], (2 * a + i) * 4096);
    c.A[i][2] = mx3::lds16(c.vA1[S][2], (2 * a + i) * 4096);
    c.A[i][3] = mx3::lds16(c.vA1[S][3], (2 * a + i) * 4096);
  }
}
template <int S, int b> NS2_DEVINL void load_w1(Ctx& c, bf16x8 (&W)[4]) {
  W[0] = mx3::lds16(c.vW1[S][0], b * HALF);
  W[1] = mx3::lds16(c.vW1[S][1], b * HALF);
  W[2] = mx3::lds16(c.vW1[S][2], b * HALF);
  W[3] = mx3::lds16(c.vW1[S][3], b * HALF);
}
// rows in front of the utterance contribute nothing: their fragments become zeros (first row tile of an utterance, taps 0 and 1 only)
template <int TAP, int a> NS2_DEVINL void zero_front(Ctx& c) {
  if constexpr (TAP < 2) {
    if (c.first) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool z = (c.zbits >> (4 * TAP + 2 * a + i)) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int4 v = __builtin_bit_cast(int4, c.A[i][j]);
          v.x = z ? 0 : v.x; v.y = z ? 0 : v.y; v.z = z ? 0 : v.z; v.w = z ? 0 : v.w;
          c.A[i][j] = __builtin_bit_cast(bf16x8, v);
        }
      }
    }
  }
}
template <int a, int b> NS2_DEVINL void mma_q1(Ctx& c, const bf16x8 (&W)[4]) {       // one half product: 4 k chunks of 16
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    c.acc[2 * a][b] = mma16<true>(c.A[0][kc], W[kc], c.acc[2 * a][b]);
    c.acc[2 * a + 1][b] = mma16<true>(c.A[1][kc], W[kc], c.acc[2 * a + 1][b]);
  }
  __builtin_amdgcn_s_setprio(0);
}
template <int S, int h, int TAP> NS2_DEVINL void issue_a1(Ctx& c) {     // this wave's two pieces of half tile A<h> of a tile with tap TAP into stage S
  dma_s<0>(c.voT[TAP][2 * h], c.sT[TAP], c.m0A[S][2 * h]);
  dma_s<0>(c.voT[TAP][2 * h + 1], c.sT[TAP], c.m0A[S][2 * h + 1]);
  if constexpr (h == 1) c.sT[TAP] += 2 * RB;                            // A1 is the later request of a tile: the tap's next tile is the next chunk
}

// One tile u = (chunk, tap TAP) of phase 1 in stage S: gemm3's tile() with the half product and the per-tap A requests.  Tile u + 1 has tap
// TAP + 1, tile u + 2 tap TAP + 2 (mod 3).
template <int S, int TAP, bool R1, bool R2, bool R1P, bool R2P> NS2_DEVINL void tile1(Ctx& c) {
  constexpr int T1 = (TAP + 1) % 3, T2 = (TAP + 2) % 3;
  load_a1<S, 0>(c); load_w1<S, 0>(c, c.W0);
  if constexpr (R1) mx3::issue_w<S ^ 1, 1>(c);
  zero_front<TAP, 0>(c);
  vmwait<2 * ((R1P ? 1 : 0) + (R2P ? 2 : 0) + (R1 ? 1 : 0))>();
  bar();
  mma_q1<0, 0>(c, c.W0);
  bar();
  load_w1<S, 1>(c, c.W1);
  if constexpr (R1) issue_a1<S ^ 1, 1, T1>(c);
  vmwait<2 * ((R2P ? 2 : 0) + (R1 ? 2 : 0))>();
  bar();
  mma_q1<0, 1>(c, c.W1);
  bar();
  load_a1<S, 1>(c);
  if constexpr (R2) issue_a1<S, 0, T2>(c);
  zero_front<TAP, 1>(c);
  bar();
  mma_q1<1, 1>(c, c.W1);
  bar();
  if constexpr (R2) mx3::issue_w<S, 0>(c);
  vmwait<(R1 ? 4 : 0) + (R2 ? 4 : 0)>();
  bar();
  mma_q1<1, 0>(c, c.W0);
  bar();
  c.sW0 += REGION; c.sW1 += REGION;
}

NS2_DEVINL void run(const GemmArgs& g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, hi = lane >> 5;
  const int ntn = (g.N + G2_BN - 1) / G2_BN, ntm = g.M / G2_BM;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = __builtin_amdgcn_readfirstlane(bid % ntn);
  bid /= ntn;
  const int tm = __builtin_amdgcn_readfirstlane(bid % ntm), z = __builtin_amdgcn_readfirstlane(bid / ntm);
  const int dil = g.dil_z ? (g.dil << z) : g.dil;
  const int dp = g.kt_per_tap * 32;                              // input channels per tap (a multiple of 128)
  const int C = dp >> 6;                                         // 64-column chunks per tap (even)
  const int m0 = tm * G2_BM;
  const long lda_b = 4L * g.lda;
  Ctx c;
  c.first = __builtin_amdgcn_readfirstlane(m0 % g.seq_len) == 0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) c.acc[mi][ni][r] = 0.f;
  // ---- phase 1 set-up
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        c.vA1[s][kc] = s * STAGE + (wm * 128 + l31) * RB + (((2 * kc + hi) ^ sw) << 4);
        c.vW1[s][kc] = s * STAGE + REGION + wn * 4096 + l31 * RB + (((2 * kc + hi) ^ sw) << 4);
      }
  }
  const int lrow = lane >> 3, pch = lane & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = 2 * wave + e, rg = (k & 7) + 16 * (k >> 3) + 8 * h, row = 8 * rg + lrow;
      const int q = pch ^ ((row >> 1) & 7);                      // logical 16-B chunk of the 64-deep row: half part of line q >> 2, chunk q & 3
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        // tap t reads input row m0 - (2 - t) dil + row = (base row m0 - 2 dil) + row + t dil; in the first row tile of an utterance the rows in
        // front of it (row + t dil < 2 dil) are clamped to row m0 (they exist; their fragments are zeroed after the LDS read)
        int srow = row + t * dil;
        if (c.first && srow < 2 * dil) srow = 2 * dil;
        c.voT[t][2 * h + e] = (unsigned)(srow * lda_b) + (q >> 2) * RB + ((q & 3) << 4);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) c.m0A[s][2 * h + e] = s * STAGE + rg * 1024;
    }
  c.zbits = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) c.zbits |= ((wm * 128 + mi * 32 + l31) < (2 - t) * dil ? 1u : 0u) << (4 * t + mi);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int b = 0; b < 2; ++b) c.m0W[s][b] = s * STAGE + REGION + b * HALF + wave * 2048;
  c.voW0 = lane * 16 + wave * 2048;
  c.voW1 = c.voW0 + HALF;
  const unsigned char* const a_col = reinterpret_cast<const unsigned char*>(g.a_hi) + 4L * z * g.a_zs;   // column block z of the input planes
  {
    const unsigned char* const b0 = a_col + ((long)m0 - 2L * dil) * lda_b;
    c.sT[0] = b0; c.sT[1] = b0; c.sT[2] = b0;
  }
  const int T1 = 3 * C;
  c.sW0 = reinterpret_cast<const unsigned char*>(g.w_tw1) + ((long)z * ntn + tn) * T1 * REGION;
  c.sW1 = c.sW0;
  // ---- phase 1: prologue (tile 0 = chunk 0 / tap 0 whole, A0 / B0 of tile 1), steady tiles, the last two
  issue_a1<0, 0, 0>(c); mx3::issue_w<0, 0>(c); mx3::issue_w<0, 1>(c); issue_a1<0, 1, 0>(c);
  c.sW0 += REGION;
  issue_a1<1, 0, 1>(c); mx3::issue_w<1, 0>(c);
  c.sW0 += REGION; c.sW1 += REGION;
  vmwait<8>();
  bar();
  if (wave >= 4) bar();
  for (int i = (C >> 1) - 1; i > 0; --i) {
    tile1<0, 0, true, true, true, true>(c); tile1<1, 1, true, true, true, true>(c); tile1<0, 2, true, true, true, true>(c);
    tile1<1, 0, true, true, true, true>(c); tile1<0, 1, true, true, true, true>(c); tile1<1, 2, true, true, true, true>(c);
  }
  tile1<0, 0, true, true, true, true>(c); tile1<1, 1, true, true, true, true>(c); tile1<0, 2, true, true, true, true>(c);
  tile1<1, 0, true, true, true, true>(c);
  tile1<0, 1, true, false, true, true>(c);
  tile1<1, 2, false, false, true, false>(c);
  if (wave < 4) bar();
  // ---- FiLM + gate on the accumulators (NS2:629-636), then phase 2: res_conv on the unshifted rows, mixed arithmetic
  const int row_base = m0 + wm * 128, col_base = tn * G2_BN + wn * 64;
  wavenet_midgate<4, 2>(c.acc, g, z, row_base, col_base, l31, hi, (g.seq_len & 127) == 0);
  const int T2 = C * 2;
  mx3::setup(c, a_col + (long)m0 * lda_b, reinterpret_cast<const unsigned char*>(g.w_tw2) + ((long)z * ntn + tn) * T2 * REGION, lda_b, wave, lane);
  mx3::kloop(c, T2, wave);
  g2_block_epilogue<2, EPI_WAVENET, true>(c.acc, g, z, tm, tn, wave, lane, smem);
}

__global__ __launch_bounds__(512, 2) void wavenet3_kernel(const GemmArgs g) { run(g); }

// ---- weights: pack [nz][rows_p][4 dp] FMT_H8 (3 dilated taps + the 1x1 res conv along K) -> per z and column tile
//   phase-1 images: (chunk c, tap) tiles of 64 columns of HALF values: LDS row = [half part of line (tap dp + 64 c) / 32 | of the next line]
//   phase-2 images: the res conv's lines 3 dp / 32 ... (whole FMT_H8 lines), as gemm3's
__global__ void wavenet3_tile1_kernel(const unsigned char* __restrict__ w, long row_bytes, long z_bytes, int dp, int ntn, int nz, uint4* __restrict__ out) {
  const int C = dp >> 6, T1 = 3 * C;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nz * ntn * T1 * (REGION / 16);
  if (idx >= total) return;
  const int ch = (int)(idx % (REGION / 16));
  long img = idx / (REGION / 16);
  const int u = (int)(img % T1); img /= T1;
  const int tn = (int)(img % ntn), z = (int)(img / ntn);
  const int cc = u / 3, tap = u - 3 * cc;
  const int byte = ch * 16;
  const int b = byte >> 14, wn = (byte >> 12) & 3, r = (byte >> 7) & 31, pos = (byte >> 4) & 7;
  const int q = pos ^ ((r >> 1) & 7);
  const long row = (long)tn * 256 + wn * 64 + b * 32 + r;
  const long line = (long)(tap * dp + cc * 64) / 32 + (q >> 2);
  out[idx] = *reinterpret_cast<const uint4*>(w + z * z_bytes + row * row_bytes + line * RB + (q & 3) * 16);
}
__global__ void wavenet3_tile2_kernel(const unsigned char* __restrict__ w, long row_bytes, long z_bytes, int dp, int ntn, int nz, uint4* __restrict__ out) {
  const int T2 = dp >> 5;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nz * ntn * T2 * (REGION / 16);
  if (idx >= total) return;
  const int ch = (int)(idx % (REGION / 16));
  long img = idx / (REGION / 16);
  const int t = (int)(img % T2); img /= T2;
  const int tn = (int)(img % ntn), z = (int)(img / ntn);
  const int byte = ch * 16;
  const int b = byte >> 14, wn = (byte >> 12) & 3, r = (byte >> 7) & 31, pos = (byte >> 4) & 7;
  const int q = pos ^ ((r >> 1) & 7);
  const long row = (long)tn * 256 + wn * 64 + b * 32 + r;
  out[idx] = *reinterpret_cast<const uint4*>(w + z * z_bytes + row * row_bytes + (long)(3 * dp / 32 + t) * RB + q * 16);
}

}  // namespace wn3

// tiled images of a Wavenet stack's weights: nz matrices [rows_p][4 dp] FMT_H8 back to back
inline size_t wavenet3_tiles1_bytes(int rows_p, int dp, int nz) { return (size_t)nz * (rows_p / 256) * 3 * (dp / 64) * wn3::REGION; }
inline size_t wavenet3_tiles2_bytes(int rows_p, int dp, int nz) { return (size_t)nz * (rows_p / 256) * (dp / 32) * wn3::REGION; }
inline hipError_t launch_wavenet3_tiles(const bf16_t* w_hi, int rows_p, int dp, int nz, bf16_t* t1, bf16_t* t2, hipStream_t s) {
  if (!w_hi || !t1 || !t2 || (rows_p & 255) || (dp & 127) || nz < 1) return hipErrorInvalidValue;
  const long row_bytes = 4L * 4 * dp, z_bytes = row_bytes * rows_p;
  const int ntn = rows_p / 256;
  const long n1 = (long)nz * ntn * 3 * (dp / 64) * (wn3::REGION / 16), n2 = (long)nz * ntn * (dp / 32) * (wn3::REGION / 16);
  hipLaunchKernelGGL(wn3::wavenet3_tile1_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned char*>(w_hi), row_bytes,
                     z_bytes, dp, ntn, nz, reinterpret_cast<uint4*>(t1));
  hipLaunchKernelGGL(wn3::wavenet3_tile2_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned char*>(w_hi), row_bytes,
                     z_bytes, dp, ntn, nz, reinterpret_cast<uint4*>(t2));
  return hipGetLastError();
}

// the hybrid plan's Wavenet block on full row tiles of utterances aligned to them
inline bool wavenet3_eligible(const GemmArgs& g, int precision) {
  if (!g.w_tw1 || !g.w_tw2 || precision != 4 || g.epi != EPI_WAVENET || !g.p1_half || g.conv_taps != 3 || g.pad_left >= 0) return false;
  const int dp = g.kt_per_tap * 32;
  if ((dp & 127) || g.nkt != 4 * g.kt_per_tap || g.mid_kt != 3 * g.kt_per_tap || g.N != dp || (g.N & 255) || g.ksplit != 0) return false;
  if (g.M <= 0 || (g.M & 255) || g.seq_len <= 0 || (g.seq_len & 255) || g.dil < 1) return false;
  const int nz = g.nz > 0 ? g.nz : 1, dmax = g.dil_z ? (g.dil << (nz - 1)) : g.dil;
  if (2 * dmax > 256) return false;                                     // rows in front of the utterance only in its first row tile
  if (g.ldw != 4 * dp || g.w_zs != (long)g.N * g.ldw || g.lda < dp || (reinterpret_cast<uintptr_t>(g.a_hi) & 15)) return false;
  return g.a_lo == g.a_hi + 32 && g.film && g.bias && g.bias2;
}
inline hipError_t launch_wavenet3(const GemmArgs& g, hipStream_t s) {
  const int nz = g.nz > 0 ? g.nz : 1;
  const int grid = (g.N / G2_BN) * (g.M / G2_BM) * nz;
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(reinterpret_cast<const void*>(&wn3::wavenet3_kernel), mx3::LDS_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wn3::wavenet3_kernel, dim3(grid), dim3(512), mx3::LDS_BYTES, s, g);
  return hipGetLastError();
}

}  // namespace ns2
