// The feed-forward causal conv as a kernel of its own (round 6):  out[m, n] = bias[n] + sum_{tap, c} A[m - (2 - tap), c] W[n, c, tap]
// with rows in front of the utterance read as zeros -- FeedForward's CausalConv1d(f, f, 3), NS2:1016 / 583-595 -- for the plans whose
// conv runs as ONE IEEE-half product (model precisions 2, 5, 6): dense half activations [M, lda] x half weights -> FMT_H8 lines or
// dense half planes.  43 % of the step's FLOPs go through it.
//
// Why not another instantiation of gemm2_kernel: that kernel's steady-state loop carried its generality inside the loop (680
// instructions per 96 MFMAs at 253 VGPRs and 137 spilled SGPRs: tap arithmetic, zero-page selects, 64-bit per-lane addresses,
// run-time K-tile coordinates).  Here everything an address depends on is fixed before the loop:
//   * W is stored PRE-TILED (ffconv3_tile_kernel): the 32 KiB LDS image of every (column tile, K tile, tap) -- rows permuted into the
//     order the waves read them, 16-byte chunks already XOR-swizzled -- contiguous in memory in the order the loop visits them.  A
//     wave's LDS-DMA of a W piece is `global_load_lds_dwordx4 v(lane * 16 + wave * 2048), s[base] offset:0 / 1024` with one 64-bit scalar
//     add per step; M0 comes from eight precomputed SGPRs.
//   * A rows are 128-byte aligned (lda % 64 == 0; the executor pads the GEGLU output to a multiple of 128 columns): the three taps of a
//     K tile share ONE 264-row tile A'(it) = input rows m0 - 8 ... m0 + 255 (tap t of output row o reads LDS row o + 6 + t), 33 whole
//     8-row pieces whose per-lane source offsets are four VGPRs + one scalar base advanced by 128 B per K tile.  The piece in front of
//     the first tile of an utterance is redirected once, before the loop, to a dump area (its LDS rows are zeroed once): no lane ever
//     needs a zero page, a bounds test or a tap shift.
//   * fragment reads are ds_read_b128 with immediate offsets off 12 + 4 address VGPRs; the loop is unrolled over two K tiles so that
//     both buffer parities are immediates.
// Steady state (disassembly of the build in profiles/r06_ffconv_*.txt): 192 MFMAs, 144 ds_read_b128, 33 LDS-DMA triples
// (s_mov m0, s_nop, global_load_lds), 48 barriers, counted vmcnt waits, two branches; no VALU, no spills.
//
// Schedule = gemm2.hip's phased loop (the guide's 256^2 8-phase template): a step = one (K tile, tap) = four accumulator-quadrant
// phases [fragment reads + requests] barrier [8 MFMAs] barrier; wave groups 0-3 / 4-7 run one barrier apart so that one group's reads
// and DMA issue sit beside the other's MFMAs; requests: phases 0 / 1 of taps 0 and 1 one piece of A'(it + 1) each, phase 2 half B0 and
// phase 3 half B1 of W(step + 2) (two pieces per wave and half); one counted vmcnt per landing (phase 3: B0 of the next step and, at
// tap 2, A'(it + 1); phase 0: B1 of this step).  Hazards as derived in gemm2.hip: wait in phase w, read in phase w + 1; re-request a
// buffer two phases after its last read.  Accumulation order per accumulator = run_k8_conv3's: results are bit-identical to it.
//
// Column tiles that are at most half valid (N = 1365 = 5 tiles + 85 columns) fetch only the half of each W image their active waves
// read and their idle waves 4-7 run a loop of requests, waits and barriers only (gemm2.hip's helper loop).  Each kind of block / wave
// is a code path of its own with its own accumulators and epilogue: alternative loop bodies that rewrite the same 128 accumulator
// registers meet in tuple copies at their merge point (measured here: 1500 spilled VGPRs).
//
// Measured (tools/probe/ffconv_probe.hip, one MI355X, 32 x 1024 frames, f = 1365): 316 us = 1160 TF against 379 us = 967 TF for
// gemm2_kernel<1, EPI_SPLIT, true, 0> on the same box; K loop 1950 shader cycles per step (2 x 32 MFMAs x 32 cycles = 2048 per SIMD:
// the matrix pipe is what the loop waits for) at an effective clock of 1.57 GHz -- the chip gives back clock for the LDS-DMA traffic
// (the same loop without DMA: 1.94 GHz, without fragment reads: 1.66 GHz).  DESIGN.md section 4, round 6.
#pragma once
#include <algorithm>

#include "gemm_epi_fast.h"

namespace ns2 {
namespace ffc3 {

constexpr int RB = 128;                       // LDS row bytes = 64 halves = one K tile of one row
constexpr int A_ROWS = 264;                   // A'(it): input rows m0 - 8 ... m0 + 255 (33 pieces of 8 rows)
constexpr int A_BUF = A_ROWS * RB;            // 33792
constexpr int W_BUF = 256 * RB;               // 32768: one (column tile, K tile, tap) image
constexpr int SA = 0, SW = 2 * A_BUF;         // LDS offsets (the dynamic LDS segment starts at 0: this kernel has no static LDS)
constexpr int S_DUMP = SW + 2 * W_BUF;        // 1 KiB nobody reads: destination of the redirected piece of first-tile blocks
constexpr int LDS_BYTES = 8 * EPI_LDS_WAVE_BYTES;
static_assert(S_DUMP + 1024 <= LDS_BYTES, "K-loop buffers fit the epilogue's LDS");

struct Args {                    // 16 dwords
  const unsigned char* A;        // dense IEEE half [M, lda]
  const unsigned char* Wt;       // tiled weights: [column tiles][3 * tpt][32 KiB LDS image]
  const float* bias;             // [N] or null
  bf16_t* out;                   // FMT_H8 lines or dense half planes [M, ldo]
  int M, N, lda, ldo;            // ldo, out_ncols: logical columns
  int seq_len, tpt;              // tpt = lda / 64 K tiles per tap (even)
  int kv;                        // columns per tap that hold data (a multiple of 32, lda - 127 ... lda): K tiles beyond are not multiplied
  int out_ncols;                 // columns [N, out_ncols) are written as zeros
};

// ---- LDS-DMA as inline assembly: the compiler must not count these loads (its waitcnt pass drains every outstanding one with
// vmcnt(0) in front of the next LDS read) and need not know M0.  `s_mov_b32 m0` + one wait state + the load (guide 5.7).  The
// immediate offset applies to BOTH the global and the LDS address.
template <int IMM>
NS2_DEVINL void dma_s(unsigned voff, const unsigned char* sbase, unsigned m0s) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(m0s), "n"(IMM) : "memory");
}
template <int N> NS2_DEVINL void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
NS2_DEVINL void bar() { asm volatile("s_barrier" ::: "memory"); }

template <int PF>
struct Kern {
  struct Ctx {
    f32x16 acc[4][2];
    bf16x8 A[2][4], W0[4], W1[4];
    unsigned vA[3][4];         // fragment read addresses of A' per (tap, k chunk); buffer and row tile are immediates
    unsigned vW[4];
    unsigned voA[4], voAx;     // DMA source offsets of this wave's A' pieces j = wave + 8 e (bytes from sA); voAx: piece 32 (wave 7)
    unsigned voW0, voW1;       // DMA source offsets inside a W image: half 0 / half 1
    const unsigned char* sA;   // A' source base of the NEXT K tile to request: A + (m0 - 8) * lda * 2 + it * 128
    const unsigned char* sW;   // W image of the NEXT step to request
    unsigned m0A[2][4];        // M0 of this wave's A' pieces per buffer (piece 0 of wave 0 of a first-tile block: the dump area)
    unsigned m0W[2][2];        // M0 of this wave's pieces of W half b per buffer
    int wave;
    int kcA, kcB;              // 16-deep k chunks that hold data in the last two K tiles (0, 2 or 4; wave-uniform)
  };

  static NS2_DEVINL bf16x8 lds16(unsigned addr, int imm) {           // imm: a constant after unrolling -> the ds_read's offset field
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    return *reinterpret_cast<const bf16x8*>(smem + addr + imm);
  }
  template <int TAP, int ABUF, int a> static NS2_DEVINL void load_a(Ctx& c) {      // fragments of A' half a (row tiles 2a, 2a + 1), tap TAP
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      c.A[i][0] = lds16(c.vA[TAP][0], SA + ABUF * A_BUF + (2 * a + i) * 4096);
      c.A[i][1] = lds16(c.vA[TAP][1], SA + ABUF * A_BUF + (2 * a + i) * 4096);
      c.A[i][2] = lds16(c.vA[TAP][2], SA + ABUF * A_BUF + (2 * a + i) * 4096);
      c.A[i][3] = lds16(c.vA[TAP][3], SA + ABUF * A_BUF + (2 * a + i) * 4096);
    }
  }
  template <int WBUF, int b> static NS2_DEVINL void load_w(Ctx& c, bf16x8 (&W)[4]) {
    W[0] = lds16(c.vW[0], WBUF * W_BUF + b * 16384);
    W[1] = lds16(c.vW[1], WBUF * W_BUF + b * 16384);
    W[2] = lds16(c.vW[2], WBUF * W_BUF + b * 16384);
    W[3] = lds16(c.vW[3], WBUF * W_BUF + b * 16384);
  }
  // KM: 0 = every k chunk; 1 / 2 = the K tile before the last / the last: kcA / kcB chunks hold data (the rest is padding that is
  // fetched but never multiplied, so its content does not matter)
  template <int a, int b, int KM> static NS2_DEVINL void mma_q(Ctx& c, const bf16x8 (&W)[4]) {
    const int kcn = KM == 1 ? c.kcA : c.kcB;
    __builtin_amdgcn_s_setprio(1);
    if (KM == 0 || kcn >= 2) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        c.acc[2 * a][b] = mma16<true>(c.A[0][kc], W[kc], c.acc[2 * a][b]);
        c.acc[2 * a + 1][b] = mma16<true>(c.A[1][kc], W[kc], c.acc[2 * a + 1][b]);
      }
    }
    if (KM == 0 || kcn == 4) {
#pragma unroll
      for (int kc = 2; kc < 4; ++kc) {
        c.acc[2 * a][b] = mma16<true>(c.A[0][kc], W[kc], c.acc[2 * a][b]);
        c.acc[2 * a + 1][b] = mma16<true>(c.A[1][kc], W[kc], c.acc[2 * a + 1][b]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // this wave's pieces of half b of the W image at c.sW into W buffer WBUF.  NW = 2: every column quarter is needed; 1: a column tile at
  // most half valid -- only the first 8 KiB of each half of the image (column quarters 0 and 1), one piece per wave
  template <int WBUF, int b, int NW> static NS2_DEVINL void issue_w(Ctx& c) {
    dma_s<0>(b ? c.voW1 : c.voW0, c.sW, c.m0W[WBUF][b]);
    if constexpr (NW == 2) dma_s<1024>(b ? c.voW1 : c.voW0, c.sW, c.m0W[WBUF][b]);
  }
  template <int ABUF, int e> static NS2_DEVINL void issue_a(Ctx& c) { dma_s<0>(c.voA[e], c.sA, c.m0A[ABUF][e]); }
  template <int ABUF> static NS2_DEVINL void issue_ax(Ctx& c) {            // the 33rd piece (wave-uniform branch)
    if (c.wave == 7) dma_s<0>(c.voAx, c.sA, SA + ABUF * A_BUF + 32 * 1024);
  }

  // One step = one (K tile, tap).  P = it & 1 = the A' buffer; the W buffer of step s = 3 it + tap is (P + TAP) & 1.
  //   ACTIVE: this wave computes (false: waves 4-7 of a half-valid column tile)   AI: request A'(it + 1)   WI: request W(step + 2)
  //   WIP: the previous step requested W (false in the very last step only)
  template <int P, int TAP, int NW, bool ACTIVE, bool AI, bool WI, int KM, bool WIP = true> static NS2_DEVINL void step(Ctx& c) {
    constexpr int WB = (P + TAP) & 1;
    // pieces of A' this wave requests in this step / requested in the previous one (wave 7's 33rd piece is not counted: the smaller count
    // is the safe one -- that wave then waits for one piece more than it has to)
    constexpr int aS = AI ? (TAP < 2 ? 2 : 0) : 0;
    constexpr int aPrev = (TAP == 0) ? 0 : (AI ? 2 : 0);
    constexpr int a0 = AI ? (TAP < 2 ? 1 : 0) : 0;
    // ---- phase 0: quadrant (0, 0)
    if constexpr (ACTIVE) { load_a<TAP, P, 0>(c); load_w<WB, 0>(c, c.W0); }
    if constexpr (AI && TAP == 0) issue_a<P ^ 1, 0>(c);
    if constexpr (AI && TAP == 1) issue_a<P ^ 1, 2>(c);
    vmwait<(WIP ? 2 * NW : 0) + aPrev + a0>();         // half B1 of THIS step (requested in phase 3 two steps ago) is read in phase 1
    bar();
    if constexpr (ACTIVE) mma_q<0, 0, KM>(c, c.W0);
    bar();
    // ---- phase 1: quadrant (0, 1)
    if constexpr (ACTIVE) load_w<WB, 1>(c, c.W1);
    if constexpr (AI && TAP == 0) issue_a<P ^ 1, 1>(c);
    if constexpr (AI && TAP == 1) { issue_a<P ^ 1, 3>(c); issue_ax<P ^ 1>(c); }
    bar();
    if constexpr (ACTIVE) mma_q<0, 1, KM>(c, c.W1);
    bar();
    // ---- phase 2: quadrant (1, 1); request B0 of W(step + 2) into this step's W buffer (its B0 half was last read in phase 0)
    if constexpr (ACTIVE) load_a<TAP, P, 1>(c);
    if constexpr (WI) issue_w<WB, 0, NW>(c);
    bar();
    if constexpr (ACTIVE) mma_q<1, 1, KM>(c, c.W1);
    bar();
    // ---- phase 3: quadrant (1, 0); request B1 of W(step + 2); B0 of step + 1 (and, before tap 0, all of A'(it + 1)) must have landed
    if constexpr (WI) { issue_w<WB, 1, NW>(c); c.sW += W_BUF; }
    vmwait<(WI ? 3 * NW : NW) + aS>();                 // younger than B0(step + 1): B1(step + 1), this step's A' pieces, B0 / B1(step + 2)
    bar();
    if constexpr (ACTIVE) mma_q<1, 0, KM>(c, c.W0);
    bar();
    if constexpr (AI && TAP == 2) c.sA += 128;
  }

  template <int NW, bool ACTIVE> static NS2_DEVINL void kloop(Ctx& c, const int tpt) {
    for (int it = 0; it + 2 < tpt; it += 2) {          // steady state: pairs of K tiles (both buffer parities are immediates)
      step<0, 0, NW, ACTIVE, true, true, 0>(c); step<0, 1, NW, ACTIVE, true, true, 0>(c); step<0, 2, NW, ACTIVE, true, true, 0>(c);
      step<1, 0, NW, ACTIVE, true, true, 0>(c); step<1, 1, NW, ACTIVE, true, true, 0>(c); step<1, 2, NW, ACTIVE, true, true, 0>(c);
    }
    // the last pair: K tiles that may be (partly) padding, no A' beyond the last tile, no W beyond the last step
    step<0, 0, NW, ACTIVE, true, true, 1>(c); step<0, 1, NW, ACTIVE, true, true, 1>(c); step<0, 2, NW, ACTIVE, true, true, 1>(c);
    step<1, 0, NW, ACTIVE, false, true, 2>(c); step<1, 1, NW, ACTIVE, false, false, 2>(c); step<1, 2, NW, ACTIVE, false, false, 2, false>(c);
  }

  template <int NW, bool ACTIVE> static NS2_DEVINL void body(const Args& g, const int tn, const int m0, const bool first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, hi = lane >> 5;
    const long lda2 = 2L * g.lda;
    Ctx c;
    c.wave = wave;
    c.kcA = min(max((g.kv - (g.tpt - 2) * 64) >> 4, 0), 4);
    c.kcB = min(max((g.kv - (g.tpt - 1) * 64) >> 4, 0), 4);
    if constexpr (ACTIVE) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) c.acc[mi][ni][r] = 0.f;
      // fragment read addresses.  A' row of output row o and tap t: o + 6 + t; 16-B chunk q of LDS row r sits at position q ^ ((r >> 1) & 7)
      // (a 16-lane ds_read_b128 group covers 16 consecutive rows = every (row parity, position) pair once: no bank conflict)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int r = wm * 128 + l31 + 6 + t;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) c.vA[t][kc] = r * RB + (((2 * kc + hi) ^ ((r >> 1) & 7)) << 4);
      }
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) c.vW[kc] = SW + wn * 4096 + l31 * RB + (((2 * kc + hi) ^ ((l31 >> 1) & 7)) << 4);
    }
    // DMA source offsets.  A' piece j = wave + 8 e: LDS rows 8 j + lrow <- input rows m0 - 8 + 8 j + lrow; LDS position p of a row holds
    // chunk p ^ swz(row) (the LDS image of a DMA instruction is lane-linear, so the swizzle is applied to the SOURCE address)
    const int lrow = lane >> 3, pch = lane & 7;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 8 * (wave + 8 * e) + lrow;
      c.voA[e] = (unsigned)(row * lda2) + ((pch ^ ((row >> 1) & 7)) << 4);
    }
    { const int row = 256 + lrow; c.voAx = (unsigned)(row * lda2) + ((pch ^ ((row >> 1) & 7)) << 4); }
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c.m0A[bf][e] = SA + bf * A_BUF + (wave + 8 * e) * 1024;
#pragma unroll
      for (int b = 0; b < 2; ++b) c.m0W[bf][b] = SW + bf * W_BUF + b * 16384 + wave * (NW * 1024);
    }
    if (first && wave == 0) {                                     // (wave-uniform) the piece in front of the utterance goes to the dump area,
      c.voA[0] += (unsigned)(8 * lda2);                           // read from rows that exist
      c.m0A[0][0] = S_DUMP; c.m0A[1][0] = S_DUMP;
    }
    c.sA = g.A + ((long)m0 - 8) * lda2;
    c.sW = g.Wt + (long)tn * (3 * g.tpt) * W_BUF;
    c.voW0 = lane * 16 + wave * (NW * 1024);
    c.voW1 = c.voW0 + 16384;
    if (first && tid < 64) {                                      // LDS rows 0-7 of both A' buffers: the zeros in front of the utterance
      *reinterpret_cast<uint4*>(smem + SA + tid * 16) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(smem + SA + A_BUF + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- prologue: A'(0), W(0), W(1) whole; everything has landed before the first read
    issue_a<0, 0>(c); issue_a<0, 1>(c); issue_a<0, 2>(c); issue_a<0, 3>(c); issue_ax<0>(c);
    c.sA += 128;
    issue_w<0, 0, NW>(c); issue_w<0, 1, NW>(c); c.sW += W_BUF; issue_w<1, 0, NW>(c); issue_w<1, 1, NW>(c); c.sW += W_BUF;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bar();
    if (wave >= 4) bar();                                         // the second group runs half a phase behind
    kloop<NW, ACTIVE>(c, g.tpt);
    if (wave < 4) bar();                                          // ... and the first waits for its last MFMA phase: LDS is free after this
    if constexpr (!ACTIVE) return;
    // ---- epilogue: bias + plane conversion through the wave's private 18 KiB of LDS (gemm_epi_fast.h), edge tiles on the generic path
    const int row_base = m0 + wm * 128, col_base = tn * 256 + wn * 64;
    if (col_base >= max(g.N, g.out_ncols)) return;
    GemmArgs ga{};
    ga.M = g.M; ga.N = g.N; ga.bias = g.bias; ga.out_hi = g.out; ga.out_lo = (PF == PF_H8) ? g.out + 32 : nullptr;
    ga.ldo_s = g.ldo; ga.out_ncols = g.out_ncols; ga.out_fmt = (PF == PF_H8) ? FMT_H8 : FMT_F16; ga.epi = EPI_SPLIT;
    if (col_base + 64 <= g.N) epi_planes_fast<PF, true>(c.acc, ga, 0, row_base, col_base, lane, smem + wave * EPI_LDS_WAVE_BYTES);
    else gemm_epilogue<EPI_SPLIT, 4, 2>(c.acc, ga, 0, row_base, col_base, 0, lane);
  }

  static NS2_DEVINL void run(const Args& g) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncols = max(g.N, g.out_ncols);
    const int ntn = (ncols + 255) >> 8;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);             // column tile fastest: the six blocks of a row tile share A' in one XCD's L2
    const int tn = __builtin_amdgcn_readfirstlane(bid % ntn), tm = __builtin_amdgcn_readfirstlane(bid / ntn);
    const int m0 = tm * 256;
    const bool first = __builtin_amdgcn_readfirstlane(m0 % g.seq_len) == 0;   // the 8 rows in front of the tile belong to the previous utterance
    const bool last_half = tn * 256 + 128 >= ncols;               // at most half of the column tile is valid
    if (!last_half) body<2, true>(g, tn, m0, first);
    else if (wave < 4) body<1, true>(g, tn, m0, first);
    else body<1, false>(g, tn, m0, first);
  }
};

template <int PF>
__global__ __launch_bounds__(512, 2) void ffconv3_kernel(const Args g) { Kern<PF>::run(g); }

// ---- weights: the row-major dense-half pack [rows_p][3 * Cp] (model_exec.cpp pack_linear) -> tiled LDS images.  Image of (column tile
// tn, step s = 3 it + tap): row n of the tile (output column tn * 256 + n) lives in half b = (n >> 5) & 1, quarter wn = n >> 6, row
// r = n & 31 at byte b * 16384 + wn * 4096 + r * 128; chunk q (8 halves) of its 64-deep K tile at position q ^ ((r >> 1) & 7).  A pure
// permutation of the packed values (+ zero padding): the operands are the old kernel's bit for bit.  One thread per 16-byte chunk.
__global__ void ffconv3_tile_kernel(const bf16_t* __restrict__ w, int ldw, int Cp, int rows_p, int tpt, int ntn, uint4* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)ntn * 3 * tpt * (W_BUF / 16);
  if (idx >= total) return;
  const int ch = (int)(idx % (W_BUF / 16));
  const long img = idx / (W_BUF / 16);
  const int s = (int)(img % (3 * tpt)), tn = (int)(img / (3 * tpt));
  const int it = s / 3, tap = s - 3 * it;
  const int byte = ch * 16;
  const int b = byte >> 14, wn = (byte >> 12) & 3, r = (byte >> 7) & 31, pos = (byte >> 4) & 7;
  const int q = pos ^ ((r >> 1) & 7);
  const int col = tn * 256 + wn * 64 + b * 32 + r;
  const int k0 = it * 64 + q * 8;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (col < rows_p && k0 + 8 <= Cp) v = *reinterpret_cast<const uint4*>(w + (long)col * ldw + (long)tap * Cp + k0);
  out[idx] = v;
}

}  // namespace ffc3

// K tiles of 64 per tap the tiled image of a conv weight with Cp packed columns per tap uses (even; the activations' row length / 64)
inline int ffconv3_tiles_per_tap(int Cp) { return ((Cp + 127) / 128) * 2; }
inline size_t ffconv3_tiled_bytes(int N, int Cp) { return (size_t)((N + 255) / 256) * 3 * ffconv3_tiles_per_tap(Cp) * ffc3::W_BUF; }

// build the tiled image of a dense IEEE-half conv weight (k = 3): w.hi = [rows_p][3 * Cp], rows beyond N and columns beyond C are zeros
inline hipError_t launch_ffconv3_tile(const bf16_t* w_hi, int ldw, int Cp, int rows_p, int N, bf16_t* out, hipStream_t s) {
  if (!w_hi || !out || ldw != 3 * Cp || (Cp & 31) || N <= 0 || rows_p < N) return hipErrorInvalidValue;
  const int tpt = ffconv3_tiles_per_tap(Cp), ntn = (N + 255) / 256;
  const long total = (long)ntn * 3 * tpt * (ffc3::W_BUF / 16);
  hipLaunchKernelGGL(ffc3::ffconv3_tile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_hi, ldw, Cp, rows_p, tpt, ntn,
                     reinterpret_cast<uint4*>(out));
  return hipGetLastError();
}

// Does this product take the dedicated kernel?  (launch_gemm asks; everything else keeps gemm2_kernel / gemm_kernel)
inline bool ffconv3_eligible(const GemmArgs& g, int precision) {
  if (!g.w_t3 || precision != 2 || g.epi != EPI_SPLIT || g.conv_taps != 3 || g.nkt != 3 * g.kt_per_tap || g.dil != 1 || g.dil_z) return false;
  if (g.nz > 1 || g.pad_left >= 0 || g.act != 0 || g.ksplit != 0 || g.a_lo || g.w_lo) return false;
  if (g.seq_len <= 0 || (g.seq_len & 255) || g.M <= 0 || (g.M & 255)) return false;
  const int Cp = g.kt_per_tap * 32, tpt = ffconv3_tiles_per_tap(Cp);
  if (g.lda != tpt * 64 || (reinterpret_cast<uintptr_t>(g.a_hi) & 15)) return false;
  if (!g.out_hi || (reinterpret_cast<uintptr_t>(g.out_hi) & 15) || (g.ldo_s & 31) || g.out_ncols > g.ldo_s) return false;
  if ((std::max(g.N, g.out_ncols) + 255) / 256 != (g.N + 255) / 256) return false;      // the zero columns beyond N stay inside the image's last column tile
  if (g.out_fmt == FMT_H8) return g.out_lo == g.out_hi + 32;
  return g.out_fmt == FMT_F16 && !g.out_lo;
}

inline hipError_t launch_ffconv3(const GemmArgs& g, hipStream_t s) {
  ffc3::Args a;
  a.A = reinterpret_cast<const unsigned char*>(g.a_hi); a.Wt = reinterpret_cast<const unsigned char*>(g.w_t3);
  a.bias = g.bias; a.out = g.out_hi;
  a.M = g.M; a.N = g.N; a.lda = g.lda; a.ldo = g.ldo_s; a.seq_len = g.seq_len; a.tpt = g.lda / 64; a.kv = g.kt_per_tap * 32;
  a.out_ncols = g.out_ncols;
  const int ntn = (std::max(g.N, g.out_ncols) + 255) / 256, grid = ntn * (g.M / 256);
  const bool h8 = g.out_fmt == FMT_H8;
  static DynLdsAttr attr_h8, attr_f16;
  const void* fn = h8 ? reinterpret_cast<const void*>(&ffc3::ffconv3_kernel<PF_H8>) : reinterpret_cast<const void*>(&ffc3::ffconv3_kernel<PF_F16>);
  hipError_t e = (h8 ? attr_h8 : attr_f16).ensure(fn, ffc3::LDS_BYTES);
  if (e != hipSuccess) return e;
  if (h8) hipLaunchKernelGGL(ffc3::ffconv3_kernel<PF_H8>, dim3(grid), dim3(512), ffc3::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(ffc3::ffconv3_kernel<PF_F16>, dim3(grid), dim3(512), ffc3::LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace ns2
