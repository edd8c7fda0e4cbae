// Large-tile GEMM / causal-conv kernel for gfx950: same contract and epilogues as gemm.hip (see there), built for the
// big token-major contractions of the denoiser (M = B*N = 32 768 rows).
//
// Why a second kernel: the 128x128 register-staged kernel measured 33 % of the bf16 MFMA peak and was bound by
// (a) L2->LDS traffic (every CU re-reads 40 KB of split-plane operands per 3.1 MFLOP) and (b) exposed load latency
// (one K-tile of look-ahead, ~770 MFMA cycles per wave to cover a ~2k-cycle load).  This kernel:
//   * 256x256 block tile, 8 waves (2 x 4), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 accumulators:
//     half the operand bytes per FLOP and twice the MFMA work per K-tile (48 MFMAs/wave in exact mode);
//   * operands go global -> LDS by **LDS-DMA** (`global_load_lds_dwordx4`, 1 KiB per wave-instruction), no VGPR staging,
//     no ds_write pass; two 64 KiB stages (128 KiB of the 160 KiB LDS), the next K-tile's DMA is issued before the
//     current tile's MFMAs and has a full iteration (>= 3k cycles at 2 waves/SIMD) to land;
//   * every LDS row is 128 B = one cache line of the source: exact mode (BK=32) [hi(32) | lo(32)] of the interleaved
//     split-plane layout (ns2_common.h), fast mode (BK=64) 64 hi values; one DMA wave-instruction moves 8 full lines
//     (the planar layout this replaced moved 16 half lines and measured 2.5x the DMA time);
//   * LDS image is lane-linear per DMA instruction, so bank conflicts are removed by an XOR swizzle applied to the
//     per-lane SOURCE address and to the fragment read address (guide rule 21): 16-B chunk c of row r is stored at
//     chunk c ^ ((r>>1)&7);
//   * causal-conv zero fill (rows before the utterance start) and M-edge rows are lanes whose source pointer is
//     redirected to a 16-B zero page -- the DMA needs no predication.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemm2_epilogue.h"
#include "ffconv_kernel.h"
#include "gemm3_kernel.h"
#include "wavenet3_kernel.h"

namespace ns2 {

// (Round 3's schedule variants, timing-diagnostic builds and the buffer-descriptor DMA experiment -- G2_VAR, G2_DIAG, G2_BUFLDS --
// live in tools/experiments/gemm2_r3_with_diag_variants.hip; the one-barrier-per-tile loops the phased loops were A/B'd against
// (round 3: -DG2_PHASED=0, -DG2_PHASED_CONV3=0, -DG2_CONV3=false) in tools/experiments/gemm2_r4_one_barrier_loops.hip.)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

NS2_DEVINL void glds16(const void* gsrc, unsigned char* ldst) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)ldst, 16, 0, 0);
}

#ifdef G2_TRACE      // per-wave phase timeline of one block (tools/trace_gemm2.py): [wave][tile][stamp] shader-clock values
__device__ unsigned long long g2_trace[8 * 64 * 10];
#define STAMP(n) do { if (tr_on) ts[n] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(n)
#endif

#ifdef G2_BLKTRACE   // block-level timeline (tools/trace_blocks.py): per block and wave, 100 MHz s_memrealtime stamps at entry / first
                     // tile landed / K loop done / epilogue stores issued / stores drained, plus HW_ID and XCC_ID
constexpr int G2_BLK_MAX = 4096;
__device__ unsigned long long g2_blk[G2_BLK_MAX * 8 * 8];
#define BSTAMP(n) do { bts[n] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BSTAMP(n)
#endif

// 256 B of zeros in device memory: the DMA source of every padded / out-of-sequence lane.  A module-scope __device__ array
// exists once per device and needs no host-side allocation or bookkeeping (the library keeps no mutable host state).
__device__ __attribute__((aligned(256))) bf16_t g2_zero_page[128];

// arithmetic of a K range.  NS: 3 = bf16 hi/lo planes, three products; 1 = one 16-bit product (bf16 or IEEE half); 2 = FMT_H8
// "mixed": one half product + both correction terms in one fp8 MFMA per 32-deep k block (ns2_common.h)
template <int NS>
struct KMode {
  static constexpr int ns = NS;
  static constexpr int np = (NS == 3) ? 2 : 1;           // 16-bit planes per operand (both inside one 128-B LDS row in exact mode)
  static constexpr bool line32 = NS != 1;                // a K tile is one interleaved 128-B line per row: 32 logical columns
  static constexpr int bk = line32 ? 32 : 64;            // K-tile depth (logical elements)
  static constexpr int kch = bk / 16;                    // 16-deep MFMA K chunks per tile (2 / 4)
};

// LDS transpose reads of the TR kernel as inline assembly.  Through the builtins (or any load the compiler knows to be an LDS read)
// the waitcnt pass puts `s_waitcnt vmcnt(0)` in front of reads that follow LDS-DMA instructions -- it cannot tell which DMA wrote
// what -- and that drains the prefetched half tiles of the NEXT K tiles every phase: the kernel measured 1.4-1.55 x the time of its
// non-transposed twin (tools/bench_wgrad.py; same time with plain ds_read_b64 in place of the transpose reads).  As assembly the
// reads are opaque: ordering against the DMA is the phased loop's own counted vmcnt + barrier discipline, and the wait for the
// reads themselves is one explicit `s_waitcnt lgkmcnt(0)` tied to the accumulators of the quadrant (tr_wait below), so no MFMA can
// be scheduled above it.
typedef __attribute__((ext_vector_type(2))) int tr_i2;
template <int OFF>
NS2_DEVINL bf16x8 tr16_pair(unsigned a0, unsigned a1) {            // tokens 4 q .. of two 4-token blocks -> one 8-deep MFMA fragment
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  tr_i2 lo, hi2;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi2) : "v"(a1), "n"(OFF) : "memory");
  return __builtin_bit_cast(bf16x8, make_int4(lo[0], lo[1], hi2[0], hi2[1]));
}
template <int OFF>
NS2_DEVINL bf16x8 tr8_pair(unsigned a) {                           // 2 x 8 tokens of fp8 bytes
  static_assert(OFF >= 0 && OFF + 1024 < 65536, "ds offset field");
  tr_i2 lo, hi2;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(hi2) : "v"(a), "n"(OFF + 1024) : "memory");
  return __builtin_bit_cast(bf16x8, make_int4(lo[0], lo[1], hi2[0], hi2[1]));
}
// fragment J (0..3) of the 32-channel group RT groups (4 KiB each) above the lane addresses a16[plane][q] / a8
template <int NS, int RT, int J>
NS2_DEVINL bf16x8 tr_frag_asm(const unsigned (&a16)[2][2], unsigned a8) {
  if constexpr (NS == 2 && J >= 2) {
    return tr8_pair<RT * 4096 + (J - 2) * 2048>(a8);
  } else {
    constexpr int p = NS == 3 ? (J >> 1) : 0, kc = NS == 3 ? (J & 1) : J;
    return tr16_pair<RT * 4096 + kc * 2048>(a16[p][0], a16[p][1]);
  }
}
template <int NS, int RT>
NS2_DEVINL void tr_frag4(bf16x8 (&f)[4], const unsigned (&a16)[2][2], unsigned a8) {
  f[0] = tr_frag_asm<NS, RT, 0>(a16, a8); f[1] = tr_frag_asm<NS, RT, 1>(a16, a8);
  f[2] = tr_frag_asm<NS, RT, 2>(a16, a8); f[3] = tr_frag_asm<NS, RT, 3>(a16, a8);
}
NS2_DEVINL void tr_wait(f32x16& c0, f32x16& c1) {                 // every transpose read of this wave has landed.  The wait names the two
  // accumulators of the quadrant, which every MFMA that consumes the fragments reads and writes: none of them can be scheduled above
  // it.  (Naming the twelve fragment tuples instead cost ~30 register copies per quadrant: the tie of each 4-register tuple fights the
  // 8-register tuple the fp8 MFMA wants its two halves in.)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1) :: "memory");
}

// P1 != 0 (EPI_WAVENET only): the first K phase -- the dilated conv taps -- runs in arithmetic P1 instead of NSPLIT, reading
// the SAME operands: P1 = 1 under NSPLIT = 2 multiplies the IEEE-half parts of the FMT_H8 lines as one product per
// contraction (64-deep tiles gathered from two lines), the second phase (res_conv) keeps the correction terms.
//
// TR (round 5, weight gradients): BOTH operands are read TRANSPOSED.  dW[r, n] = sum_m dY[m, r] X[m - shift(n), k(n)] contracts over
// the TOKENS, and the operand planes of the training path are token-major ([M tokens, channels], one 128-B line per token and 32
// channels) -- rounds 4's wgrad ran this kernel on transposed COPIES written by tplanes passes (18 ms of a 117 ms step at the
// HBM roof).  Here the source lines go into LDS as they are (LDS row = one token's line of one 32-channel group, row index =
// channel group * 32 + token, so the half-tile / DMA-piece structure of the phased loop is unchanged) and the fragments are formed
// by gfx950's LDS transpose reads: ds_read_b64_tr_b16 hands a lane 4 tokens of ITS channel from a [4 tokens][16 channels] block of
// 16-bit values, ds_read_b64_tr_b8 8 tokens from an [8 tokens][16 channels] block of bytes (semantics probed on the device:
// tools/probe/tr_probe.hip) -- two reads per 16-deep MFMA fragment, four per 32 fp8 bytes; A and B use the same token -> k
// assignment, so the permutation inside a k block cancels.  The conv taps of a weight gradient are column blocks of N (tap =
// n / tr_kp) whose source rows are shifted by (tr_taps - 1 - tap) * dil TOKENS inside the utterance: no shifted copies either.
template <int NSPLIT, int EPI, bool F16, int P1 = 0, bool TR = false>
__global__ __launch_bounds__(512, 2) void gemm2_kernel(const GemmArgs g) {
  const bf16_t* zero_page = g2_zero_page;
  static_assert(!TR || (EPI == EPI_F32 && P1 == 0 && NSPLIT != 1), "transposed operands: weight gradients (fp32 slots) on interleaved lines");
  static_assert(NSPLIT != 2 || F16, "the mixed mode multiplies IEEE-half operands");
  static_assert(P1 == 0 || (EPI == EPI_WAVENET && P1 == 1 && NSPLIT == 2), "phase-1 override: half product under the mixed mode");
  using ModeMain = KMode<NSPLIT>;
  using ModeP1 = KMode<P1 ? P1 : NSPLIT>;
  constexpr int RB = 128;                            // LDS row bytes: [hi32|lo32] (exact), [half32|h8 32|l8 32] (mixed) or hi64
  constexpr int CPR = RB / 16;                       // 16-B chunks per row (8)
  constexpr int RPI = 64 / CPR;                      // tile rows moved by one DMA wave-instruction (8)
  constexpr int REGION = G2_BM * RB;                 // one operand of one K tile: 32 KiB
  constexpr int STAGE = 2 * REGION;                  // 64 KiB
  constexpr int IPO = G2_BM / RPI;                   // DMA instructions per operand (32)
  static_assert(2 * IPO == 64, "8 waves x 8 DMA instructions per K-tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef G2_BLKTRACE
  unsigned long long bts[6] = {0, 0, 0, 0, 0, 0};
  BSTAMP(0);
  const unsigned long long cyc0 = __builtin_readcyclecounter();
#endif
  // wave -> (row half, column quarter): waves w and w+4 share a SIMD (dispatch order 0,2,1,3,0,2,1,3), so column
  // quarters {0,1} and {2,3} are paired on every SIMD: when the last column tile is at most half valid (FF conv:
  // N = 1365 = 5.33 tiles) the idle quarters leave each SIMD's MFMA pipe to one active wave instead of idling two SIMDs.
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int ntn = (g.N + G2_BN - 1) / G2_BN;
  const int ntm = (g.M + G2_BM - 1) / G2_BM;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % ntn;
  bid /= ntn;
  const int tm = bid % ntm;
  const int z = bid / ntm;
  const int dil = g.dil_z ? (g.dil << z) : g.dil;

  // operand layouts (ns2_common.h): exact mode requires interleaved operands (checked by launch_gemm); the fast kernel
  // reads the hi plane of either layout
  const bool ail = g.a_lo != nullptr, wil = g.w_lo != nullptr;
  const long a_rs = pld(g.lda, ail), w_rs = pld(g.ldw, wil);     // physical row strides

  // ---- DMA roles: instruction j = (wave&3)*8 + i of the operand; waves 0-3 stream A, waves 4-7 stream W
  const bool a_wave = wave < 4;
  const int lrow = lane / CPR, pchunk = lane % CPR;
  const bf16_t* src[8];        // per-instruction source ROW pointer at K offset 0 (A: unshifted row), without the chunk offset
  int nseq[8];                 // A only: position inside the utterance (for the causal zero fill); -2^30 = row >= M (no tap shift brings it back into range)
  int ldst[8];                 // LDS byte offset inside a stage (wave-uniform)
  // logical 16-B chunk this lane fetches: pchunk ^ swizzle(row); row = 8 rg + lrow and rg = 8 (wave & 3) + i, so the
  // swizzle (row >> 1) & 7 = 4 (i & 1) + (lrow >> 1) takes two values per lane, for even and odd i
  const int lchunk_par[2] = {pchunk ^ (lrow >> 1), pchunk ^ (4 + (lrow >> 1))};
  const bool opil = a_wave ? ail : wil;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int rg = (wave & 3) * 8 + i;               // row group inside the operand: [0, IPO)
    const int row = rg * RPI + lrow;                 // tile row
    ldst[i] = (a_wave ? 0 : REGION) + rg * 1024;
    if (a_wave) {
      const long m = (long)tm * G2_BM + row;
      src[i] = g.a_hi + pcol((int)(z * g.a_zs), ail) + m * a_rs;
      nseq[i] = (m < g.M) ? ((g.seq_len > 0) ? (int)(m % g.seq_len) : 0x3fffffff) : -0x40000000;
    } else {
      src[i] = g.w_hi + (((long)z * g.w_zs) << (wil ? 1 : 0)) + ((long)tn * G2_BN + row) * w_rs;
      nseq[i] = 0;
    }
  }

  // K tiling in BK units: every tap spans tpt tiles; with BK = 64 an odd 32-multiple tap ends in a half tile whose
  // upper 32 columns are zero-filled (A and W lanes of those chunks read the zero page)
  const int tap_k = g.kt_per_tap * 32;                        // logical elements per tap
  const int ntaps = g.nkt / g.kt_per_tap;
  auto tiles_per_tap = [&](auto mode) __attribute__((always_inline)) { return (tap_k + decltype(mode)::bk - 1) / decltype(mode)::bk; };

  auto issue_tile = [&](auto mode, int kt, int stage) __attribute__((always_inline)) {
    using M = decltype(mode);
    constexpr int BK = M::bk;
    const int tpt = tiles_per_tap(mode);
    const bool half_tail = !M::line32 && (g.kt_per_tap & 1);
    const int nconv = g.conv_taps * tpt;                      // tiles of the row-shifted taps (come first)
    unsigned char* sbase = smem + stage * STAGE;
    // K order: the shifted conv taps are visited tap-minor (k chunk 0: taps 0..T-1, k chunk 1: ...).  Consecutive taps
    // read the same A lines shifted by `dil` rows, so every re-read comes straight after the first touch and hits L2;
    // tap-major order streamed the whole A row range (1.4 MB per block, ~7.5 MB per XCD) between two visits.
    int tap, it;
    if (kt < nconv) { it = kt / g.conv_taps; tap = kt - it * g.conv_taps; }
    else { tap = kt / tpt; it = kt - tap * tpt; }
    const bool half = half_tail && (it == tpt - 1);
    // chunk offset inside the tile (elements): a line32 tile is ONE 128-B line, chunk c = 8 elements at c * 8 (hi chunks
    // 0-3, lo / fp8 chunks 4-7); a 64-deep tile of one plane takes logical columns c * 8 of either operand layout
    int coff[2];
    bool chi[2];                                              // (64-deep tiles) this lane fetches one of the upper 32 columns
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      coff[e] = M::line32 ? lchunk_par[e] * 8 : pcol(lchunk_par[e] * 8, opil);
      chi[e] = lchunk_par[e] >= 4;
    }
    if (a_wave) {
      const int pl = g.pad_left < 0 ? g.conv_taps - 1 : g.pad_left;      // causal: all padding on the left (NS2:583-595)
      const int shift = (tap < g.conv_taps) ? (pl - tap) * dil : 0;
      const unsigned slim = g.seq_len > 0 ? (unsigned)g.seq_len : 0x7fffffffu;
      const long off = pcol(it * BK, ail) - (long)shift * a_rs;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = ((unsigned)(nseq[i] - shift) < slim) && !(half && chi[i & 1]);   // nseq = -2^30 marks rows >= M
        const bf16_t* p = ok ? (src[i] + off + coff[i & 1]) : zero_page;
        glds16(p, sbase + ldst[i]);
      }
    } else {
      const long off = pcol(tap * tap_k + it * BK, wil);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16_t* p = (half && chi[i & 1]) ? zero_page : (src[i] + off + coff[i & 1]);
        glds16(p, sbase + ldst[i]);
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int row_base = tm * G2_BM + wm * 128;
  const int col_base = tn * G2_BN + wn * 64;
  // nothing to compute or store for this wave (it still streams its share of the operands and joins the barriers)
  const int ncols_needed = (EPI == EPI_GEGLU || EPI == EPI_F32 || EPI == EPI_QKV) ? g.N : max(g.N, g.out_ncols);
  const bool wave_active = col_base < ncols_needed;

  // fragment read addressing: row = wave base + 32*i + l31 ; physical chunk = (4*plane + 2*kc + hi) ^ swz(row); the swizzle
  // depends on l31 only, and a 16-lane ds_read_b128 group covers 16 distinct (row&1, (row>>1)&7) pairs = all 64 banks
  const int fswz = (l31 >> 1) & 7;
  const int a_row_off = (wm * 128 + l31) * RB;
  const int w_row_off = REGION + (wn * 64 + l31) * RB;

#ifdef G2_TRACE
  unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool tr_on = false;
#endif

  // ---- one K tile of MFMAs.  A fragments come from `sa` (row offset a_off, swizzle fz_a), W fragments from `sw`; the two
  // differ only for the tap-shared conv path, where the A rows are shifted by the tap.  `after_first` runs after the first
  // group of MFMAs (the late DMA issue of waves 4-7).
  auto compute_tile = [&](auto mode, const unsigned char* sa, const int a_off, const int fz_a, const unsigned char* sw,
                          const int w_off, const int fz_w, auto&& after_first) __attribute__((always_inline)) {
    using M = decltype(mode);
    constexpr int NS = M::ns, NP = M::np, KCH = M::kch;
    if constexpr (NS == 2) {
      // mixed mode: per 32-deep tile 2 x (4x2) half MFMAs + (4x2) fp8 MFMAs of K = 64.  fp8 operands of lane (l31, hi):
      // A = 32 bytes [h8 | l8][hi] of row l31 (chunks 4+2hi, 5+2hi of the line), B = [l8 | h8][hi] (chunks 6-2hi, 7-2hi):
      // lanes 0-31 contribute a_h8 . w_l8, lanes 32-63 a_l8 . w_h8; every product carries exactly one 2^12-scaled factor,
      // undone by the block scale 2^-12 on A.
      bf16x8 af[2][4], wf[2][2];
      i32x8 a8[4], w8[2];
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const int ca = ((2 * kc + hi) ^ fz_a) * 16, cw = ((2 * kc + hi) ^ fz_w) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[kc][i] = *reinterpret_cast<const bf16x8*>(sa + a_off + i * 32 * RB + ca);
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[kc][i] = *reinterpret_cast<const bf16x8*>(sw + w_off + i * 32 * RB + cw);
      }
      {
        const int ca0 = ((4 + 2 * hi) ^ fz_a) * 16, ca1 = ((5 + 2 * hi) ^ fz_a) * 16;
        const int cw0 = ((6 - 2 * hi) ^ fz_w) * 16, cw1 = ((7 - 2 * hi) ^ fz_w) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int4 lo4 = *reinterpret_cast<const int4*>(sa + a_off + i * 32 * RB + ca0);
          const int4 hi4 = *reinterpret_cast<const int4*>(sa + a_off + i * 32 * RB + ca1);
          a8[i] = i32x8{lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int4 lo4 = *reinterpret_cast<const int4*>(sw + w_off + i * 32 * RB + cw0);
          const int4 hi4 = *reinterpret_cast<const int4*>(sw + w_off + i * 32 * RB + cw1);
          w8[i] = i32x8{lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = mma16<true>(af[0][mi], wf[0][ni], acc[mi][ni]);
          acc[mi][ni] = mma16<true>(af[1][mi], wf[1][ni], acc[mi][ni]);
        }
      after_first();
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[mi], w8[ni], acc[mi][ni], /*A e5m2*/ 1, /*B e5m2*/ 1,
                                                                        0, H8_E8M0_LO, 0, H8_E8M0_ONE);
    } else {
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        bf16x8 af[NP][4], wf[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int ca = ((4 * p + 2 * kc + hi) ^ fz_a) * 16, cw = ((4 * p + 2 * kc + hi) ^ fz_w) * 16;
#pragma unroll
          for (int i = 0; i < 4; ++i) af[p][i] = *reinterpret_cast<const bf16x8*>(sa + a_off + i * 32 * RB + ca);
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[p][i] = *reinterpret_cast<const bf16x8*>(sw + w_off + i * 32 * RB + cw);
        }
#ifdef G2_TRACE
        if (kc < 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (kc == 0) STAMP(2); else STAMP(5); }
#endif
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr (NS == 3) {
              acc[mi][ni] = mma16<F16>(af[1][mi], wf[0][ni], acc[mi][ni]);
              acc[mi][ni] = mma16<F16>(af[0][mi], wf[1][ni], acc[mi][ni]);
            }
            acc[mi][ni] = mma16<F16>(af[0][mi], wf[0][ni], acc[mi][ni]);
          }
#ifdef G2_TRACE
        if (kc == 0) STAMP(3); else if (kc == 1) STAMP(6);
#endif
        if (kc == 0) after_first();
#ifdef G2_TRACE
        if (kc == 0) STAMP(4);
#endif
      }
    }
  };

  auto run_k = [&](auto mode, const int kt0, const int kt1) __attribute__((always_inline)) {
    issue_tile(mode, kt0, kt0 & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // tile kt0 landed for every wave
#ifdef G2_BLKTRACE
    if (bts[1] == 0) BSTAMP(1);
#endif
    for (int kt = kt0; kt < kt1; ++kt) {
#ifdef G2_TRACE
#pragma unroll
      for (int n = 0; n < 10; ++n) ts[n] = 0;
      tr_on = (EPI == EPI_SPLIT) && blockIdx.x == 8 * 37 && kt >= 16 && kt < 80;
#endif
      STAMP(0);
#ifdef G2_TRACE
      if (tr_on) ts[9] = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz: calibrates the shader clock
#endif
      // Anti-phase DMA issue: an LDS-DMA instruction blocks its wave for ~60-180 clocks at issue.  The A-streaming
      // waves 0-3 (one per SIMD) issue theirs now, while their SIMD partners 4-7 already run MFMAs; waves 4-7 issue
      // the W half after their first K step, when waves 0-3 are in their MFMA phase (measured +5 % on the FF conv).
      if (kt + 1 < kt1 && a_wave) issue_tile(mode, kt + 1, (kt + 1) & 1);
      STAMP(1);
      const unsigned char* sb = smem + (kt & 1) * STAGE;
      auto late = [&]() __attribute__((always_inline)) { if (kt + 1 < kt1 && !a_wave) issue_tile(mode, kt + 1, (kt + 1) & 1); };
      if (wave_active) compute_tile(mode, sb, a_row_off, fswz, sb, w_row_off, fswz, late);
      else late();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile has landed
      STAMP(7);
      __syncthreads();                                // ... everybody's has, and this stage is free to overwrite
#ifdef G2_TRACE
      STAMP(8);
      if (tr_on && lane == 0) {
#pragma unroll
        for (int n = 0; n < 10; ++n) g2_trace[(wave * 64 + (kt - 16)) * 10 + n] = ts[n];
      }
#endif
    }
  };


  // ================================================================================================================
  // Phased K loop: the SAME LDS image, DMA pieces and fragment addressing as run_k above, re-scheduled on the
  // guide's "256^2 8-phase" structure.  run_k waits `vmcnt(0)` + one workgroup barrier per K tile: the tile requested at the
  // start of tile t must have landed by its end, and the block timeline (profiles/r03_block_timeline_*.txt) shows 54-67 %
  // MFMA duty -- the K loop waits for DMA latency, not for DMA throughput.  Here:
  //   * a K tile is consumed in four phases, one accumulator quadrant each: (a, b) = (0,0) (0,1) (1,1) (1,0), quadrant =
  //     row-tile pair mi in {2a, 2a+1} x column tile ni = b of the wave tile: 8-12 MFMAs = 256-384 matrix-pipe cycles;
  //   * the tile's operands are four HALF TILES: A0 / A1 = the tile rows with bit 6 clear / set (= the mi < 2 / mi >= 2 rows
  //     of every wave), B0 / B1 = the W rows with bit 5 clear / set (= ni 0 / 1 of every wave); 16 DMA pieces (16 KiB) each,
  //     two per wave.  A0 and B0 are read (into registers) in phase 0, B1 in phase 1, A1 in phase 2: a half tile's LDS rows
  //     are free two phases after its last read and are refilled with the data of tile t + 2 straight away -- half tile X of
  //     tile u is requested at phase 4u - 6 (A0), 4u - 5 (B0), 4u - 4 (B1), 4u - 3 (A1), five to six phases (>= 1300 matrix
  //     cycles) before its first read, one half tile per phase, four in flight;
  //   * one counted `s_waitcnt vmcnt(8)` per phase (everything older than the last four half tiles has landed), raw
  //     `s_barrier`s (a __syncthreads() would drain the LDS-DMA queue), never vmcnt(0) before the last two tiles;
  //   * waves 0-3 and 4-7 (one of each per SIMD) run half a phase apart: while one group is in its MFMA cluster (priority 1)
  //     the other reads fragments and issues DMA, so the SIMD's matrix pipe always has a wave to serve.
  // Hazards (MI355X_MICROARCH.md "Two waves per SIMD" 7, guide "256^2 8-phase template"): LDS-DMA data is ordered for a
  // ds_read only by the issuing wave's vmcnt plus a barrier; every wave waits in the load interval of phase p - 1 for the
  // half tile read in phase p (two barriers in between, also for the lagging group); a half tile is refilled >= 2 phases after
  // its last read (the reads have retired at the lgkmcnt before that phase's MFMAs, again two barriers earlier).
  // Accumulation order per accumulator is unchanged: results are bit-identical to run_k.
  const bf16_t* psrc[4][2];      // [A0, A1, B0, B1][piece]: source row pointer at K offset 0
  int pnseq[2][2];               // A pieces: position inside the utterance (-2^30 = row >= M)
  int pldst[4][2];               // LDS byte offset inside a stage (wave-uniform)
  // TR: a piece = 8 tokens of ONE 32-channel group: row group rg = 4 * channel group + token block.  psrc = this lane's 16 bytes of
  // that group in ITS token row of the slice's first K tile (tap shift applied; the zero page when the group lies beyond the operand),
  // ptok = the lane's token inside a K tile, ptshift = the tap shift of a B piece's channel group in tokens
  int ptok[4][2], ptshift[2][2];
  bool pvalid[4][2];             // (wave-uniform) the piece's channel group exists in the operand
  // TR LDS image of a piece (1 KiB = 8 tokens x the 128-B line of one channel group): eight 128-B BLOCKS, each exactly what one
  // 16-lane transpose group reads as 16 x 8 contiguous bytes, and the two groups of a 32-lane half side by side (256 contiguous
  // bytes = every bank once: the layout MI355X_MICROARCH.md lists as conflict free for these reads):
  //   bf16 x3:  block 4 p + 2 tb + tg = [4 tokens 4 tb ..][16 channels 16 tg ..] of plane p (hi / lo)
  //   FMT_H8:   blocks 0-3 = 2 tb + tg of the half part as above;  blocks 4 + 2 part + tg = [8 tokens][16 channels] bytes of h8 / l8
  // A DMA lane (block b = lane >> 3, 16-byte unit w = lane & 7) therefore fetches token tr_tokp, logical chunk tr_lchunk of the line:
  const int tr_b = lane >> 3, tr_w = lane & 7;
  const bool tr_bytes = (NSPLIT == 2) && tr_b >= 4;
  const int tr_tokp = tr_bytes ? tr_w : 4 * ((tr_b >> 1) & 1) + (tr_w >> 1);
  const int tr_lchunk = tr_bytes ? tr_b : ((NSPLIT == 3 ? 4 * (tr_b >> 2) : 0) + 2 * (tr_b & 1) + (tr_w & 1));
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = 2 * wave + e;                      // piece of the half tile: [0, 16)
      const int rg = (h < 2) ? ((k & 7) + 16 * (k >> 3) + 8 * h) : ((k & 3) + 8 * (k >> 2) + 4 * (h - 2));   // row group: parity == e
      const int row = rg * RPI + lrow;
      pldst[h][e] = (h < 2 ? 0 : REGION) + rg * 1024;
      if constexpr (TR) {
        const int cg = rg >> 2;                        // 32-channel group inside the tile: rows cg * 32 ... of the output tile
        ptok[h][e] = (rg & 3) * 8 + tr_tokp;
        const long tok = (long)z * g.nkt * 32 + ptok[h][e];      // this lane's token in the slice's first K tile
        if (h < 2) {
          const int c0 = (tm * 8 + cg) * 32;           // output row (channel of dY)
          pvalid[h][e] = c0 < g.lda;
          psrc[h][e] = pvalid[h][e] ? g.a_hi + pcol(c0, true) + tok * a_rs + tr_lchunk * 8 : zero_page;
        } else {
          const int n0 = (tn * 8 + cg) * 32;           // output column: tap * tr_kp + channel of X
          const int tap = n0 / g.tr_kp, ch = n0 - tap * g.tr_kp;
          const int sh = g.tr_taps > 0 ? (g.tr_taps - 1 - tap) * dil : 0;
          ptshift[h - 2][e] = sh;
          pvalid[h][e] = n0 < g.N;
          psrc[h][e] = pvalid[h][e] ? g.w_hi + pcol(ch, true) + (tok - sh) * w_rs + tr_lchunk * 8 : zero_page;
          if (!pvalid[h][e]) ptshift[h - 2][e] = 0;
        }
        continue;
      }
      if (h < 2) {
        const long m = (long)tm * G2_BM + row;
        psrc[h][e] = g.a_hi + pcol((int)(z * g.a_zs), ail) + m * a_rs;
        pnseq[h][e] = (m < g.M) ? ((g.seq_len > 0) ? (int)(m % g.seq_len) : 0x3fffffff) : -0x40000000;
      } else {
        psrc[h][e] = g.w_hi + (((long)z * g.w_zs) << (wil ? 1 : 0)) + ((long)tn * G2_BN + row) * w_rs;
      }
    }

  int tr_tok0[4], tr_n0[4];      // TR: per half tile, the first token of the NEXT tile to request and its position inside the utterance
  long tr_off[4];                //     ... and the element offset of that tile from the slice's first (added to psrc)
  if constexpr (TR) {
    const int t0 = z * g.nkt * 32;
    const int nb = g.seq_len > 0 ? t0 % g.seq_len : 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) { tr_tok0[h] = t0; tr_n0[h] = nb; tr_off[h] = 0; }
  }
  // (not in the single-product EPI_SPLIT instantiation = the FF causal conv of the hybrid plan, the dominant kernel: it runs the
  // tap-shared loop below, and a second issue path in its unused general loop cost it 2.6 % through register allocation)
  constexpr bool A_LIN = !(EPI == EPI_SPLIT && NSPLIT == 1);
  const bool a_lin_full = A_LIN && g.conv_taps == 0 && tm * G2_BM + G2_BM <= g.M;      // see issue_half
  struct TileCoord { int tap, it; };
  auto tile_coord = [&](auto mode, int kt) __attribute__((always_inline)) {
    const int tpt = tiles_per_tap(mode);
    const int nconv = g.conv_taps * tpt;
    TileCoord c;
    if (kt < nconv) { c.it = kt / g.conv_taps; c.tap = kt - c.it * g.conv_taps; }      // tap-minor over the shifted taps (see issue_tile)
    else { c.tap = kt / tpt; c.it = kt - c.tap * tpt; }
    return c;
  };
  // this wave's two pieces of half tile H (0 = A0, 1 = A1, 2 = B0, 3 = B1) of the tile at `c`, into stage `stage`
  auto issue_half = [&](auto mode, const TileCoord c, int stage, auto hsel) __attribute__((always_inline)) {
    using M = decltype(mode);
    constexpr int H = decltype(hsel)::value;
    constexpr int BK = M::bk;
    const int tpt = tiles_per_tap(mode);
    const bool half = !M::line32 && (g.kt_per_tap & 1) && (c.it == tpt - 1);
    const bool il = (H < 2) ? ail : wil;
    unsigned char* sbase = smem + stage * STAGE;
    if constexpr (TR) {
      // Half tile H is requested once per K tile, in tile order: its tokens [tok0, tok0 + 32) and their position n0 inside the
      // utterance are running (wave-uniform) counters -- no division in the loop -- and the piece pointers advance by 32 token rows.
      // Almost every tile is "full" (all 32 tokens exist and none of them precedes its utterance's start by less than the tap
      // shift): its lanes need no test; the others take the per-lane path to the zero page.
      const int tok0 = tr_tok0[H], n0 = tr_n0[H];
      const long off = tr_off[H];
      // wave-uniform: every token of the tile exists and (B side) none of them sits closer to its utterance's start than the largest
      // tap shift of the two pieces
      bool full = tok0 + 32 <= g.tr_tokens;
      if constexpr (H >= 2) {
        const int shmax = max(ptshift[H - 2][0], ptshift[H - 2][1]);
        full = full && (shmax == 0 || (n0 >= shmax && n0 + 32 <= g.seq_len));
      }
      const bf16_t* p[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) p[e] = psrc[H][e] + (pvalid[H][e] ? off : 0L);      // psrc of a missing channel group = the zero page
      if (!full) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          bool ok = tok0 + ptok[H][e] < g.tr_tokens;
          if constexpr (H >= 2) {
            int n = n0 + ptok[H][e];                                 // position inside the utterance (seq_len >= 32: wraps at most once)
            if (g.seq_len > 0 && n >= g.seq_len) n -= g.seq_len;
            ok = ok && (ptshift[H - 2][e] == 0 || n >= ptshift[H - 2][e]);
          }
          if (!ok) p[e] = zero_page;
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) glds16(p[e], sbase + pldst[H][e]);
      tr_tok0[H] = tok0 + 32;
      tr_off[H] = off + 32 * (H < 2 ? a_rs : w_rs);
      if (g.seq_len > 0) { const int n1 = n0 + 32; tr_n0[H] = n1 >= g.seq_len ? n1 - g.seq_len : n1; }
      return;
    }
    int coff[2];
    bool chi[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      coff[e] = M::line32 ? lchunk_par[e] * 8 : pcol(lchunk_par[e] * 8, il);
      chi[e] = lchunk_par[e] >= 4;
    }
    if constexpr (H < 2) {
      if (a_lin_full && !half) {
        // plain linear product on a row tile that lies inside the operand (wave-uniform): no lane can need the zero page and there is
        // no tap shift to apply -- the pieces cost one 64-bit add each instead of the shift arithmetic + compare + two selects (round 5:
        // the K loops are sensitive to the scalar / vector work around their DMA issue, see the TR kernel's notes)
        const long off = pcol(c.it * BK, ail);
#pragma unroll
        for (int e = 0; e < 2; ++e) glds16(psrc[H][e] + off + coff[e], sbase + pldst[H][e]);
        return;
      }
      const int pl = g.pad_left < 0 ? g.conv_taps - 1 : g.pad_left;
      const int shift = (c.tap < g.conv_taps) ? (pl - c.tap) * dil : 0;
      const unsigned slim = g.seq_len > 0 ? (unsigned)g.seq_len : 0x7fffffffu;
      const long off = pcol(c.it * BK, ail) - (long)shift * a_rs;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = ((unsigned)(pnseq[H][e] - shift) < slim) && !(half && chi[e]);
        glds16(ok ? (psrc[H][e] + off + coff[e]) : zero_page, sbase + pldst[H][e]);
      }
    } else {
      const long off = pcol(c.tap * tap_k + c.it * BK, wil);
#pragma unroll
      for (int e = 0; e < 2; ++e)
        glds16((half && chi[e]) ? zero_page : (psrc[H][e] + off + coff[e]), sbase + pldst[H][e]);
    }
  };
  using HA0 = std::integral_constant<int, 0>; using HA1 = std::integral_constant<int, 1>;
  using HB0 = std::integral_constant<int, 2>; using HB1 = std::integral_constant<int, 3>;

  // fragments of one half tile.  FR = 16-byte fragment reads per row tile: single product 4 (k chunks), bf16 x3 4 (2 planes x 2 k
  // chunks), mixed 4 (2 half k chunks + the 32 fp8 bytes of the lane's k half)
  struct AHalf { bf16x8 f[2][4]; };      // [row tile of the pair][read]
  struct WHalf { bf16x8 f[4]; };
  auto frag_chunk = [&](auto mode, int j, bool w_side) __attribute__((always_inline)) {     // logical 16-B chunk of read j for this lane
    using M = decltype(mode);
    if constexpr (M::ns == 1) return 2 * j + hi;                       // k chunk j of the 64-deep row
    else if constexpr (M::ns == 3) return 4 * (j >> 1) + 2 * (j & 1) + hi;   // plane j >> 1, k chunk j & 1
    else return (j < 2) ? (2 * j + hi) : (w_side ? (6 - 2 * hi + (j - 2)) : (4 + 2 * hi + (j - 2)));   // half k chunks, then [h8|l8] / [l8|h8]
  };
  // ---- TR fragments.  i16 = lane & 15 = the lane's place in its 16-lane transpose group, tg = (lane >> 4) & 1 = which 16 channels of
  // the 32-channel group (lanes l31 < 16 / >= 16 = MFMA rows 0-15 / 16-31), hi = the MFMA's k half.
  //   16-bit read q (0, 1) of k chunk kc of plane p: the group reads block 4 p + 2 q + tg of piece 2 kc + hi (tokens 16 kc + 8 hi + 4 q ..+3);
  //     the lane receives those 4 tokens of channel 16 tg + i16;
  //   byte read q (0 .. 3) of part P: block 4 + 2 P + tg of piece q (tokens 8 q .. +7); the lane receives 8 tokens of its channel.
  // Offsets inside a channel group's 4 pieces (4 KiB): tr16_off[p][q] + kc * 2048, tr8_off[w_side] + q * 1024.
  int tr16_off[2][2], tr8_off[2];
  if constexpr (TR) {
    const int i16 = lane & 15, tg = (lane >> 4) & 1;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) tr16_off[p][q] = hi * 1024 + (4 * p + 2 * q + tg) * 128 + i16 * 8;
#pragma unroll
    for (int ws = 0; ws < 2; ++ws) {
      const int part = ws ? 1 - hi : hi;                                // A: lanes 0-31 h8, 32-63 l8;  W: lanes 0-31 l8, 32-63 h8
      tr8_off[ws] = (4 + 2 * part + tg) * 128 + i16 * 8;
    }
  }
  auto load_a = [&](auto mode, const unsigned char* sa, const int a_off, const int fz, int a, AHalf& A) __attribute__((always_inline)) {
    if constexpr (TR) {
      constexpr int NS = decltype(mode)::ns;
      const unsigned base = (unsigned)(size_t)sa + wm * (4 * 32 * RB);           // LDS byte address of this wave's first channel group
      const unsigned a16[2][2] = {{base + tr16_off[0][0], base + tr16_off[0][1]}, {base + tr16_off[1][0], base + tr16_off[1][1]}};
      const unsigned a8 = base + tr8_off[0];
      if (a == 0) { tr_frag4<NS, 0>(A.f[0], a16, a8); tr_frag4<NS, 1>(A.f[1], a16, a8); }
      else { tr_frag4<NS, 2>(A.f[0], a16, a8); tr_frag4<NS, 3>(A.f[1], a16, a8); }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        A.f[i][j] = *reinterpret_cast<const bf16x8*>(sa + a_off + (2 * a + i) * 32 * RB + ((frag_chunk(mode, j, false) ^ fz) * 16));
      }
  };
  auto load_w = [&](auto mode, const unsigned char* sw, const int w_off, const int fz, int b, WHalf& W) __attribute__((always_inline)) {
    if constexpr (TR) {
      constexpr int NS = decltype(mode)::ns;
      const unsigned base = (unsigned)(size_t)sw + REGION + wn * (2 * 32 * RB);
      const unsigned a16[2][2] = {{base + tr16_off[0][0], base + tr16_off[0][1]}, {base + tr16_off[1][0], base + tr16_off[1][1]}};
      const unsigned a8 = base + tr8_off[1];
      if (b == 0) tr_frag4<NS, 0>(W.f, a16, a8);
      else tr_frag4<NS, 1>(W.f, a16, a8);
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      W.f[j] = *reinterpret_cast<const bf16x8*>(sw + w_off + b * 32 * RB + ((frag_chunk(mode, j, true) ^ fz) * 16));
    }
  };
  auto mma_quadrant = [&](auto mode, int a, int b, const AHalf& A, const WHalf& W) __attribute__((always_inline)) {
    using M = decltype(mode);
    if constexpr (TR) tr_wait(acc[2 * a][b], acc[2 * a + 1][b]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16& c = acc[2 * a + i][b];
      if constexpr (M::ns == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mma16<F16>(A.f[i][j], W.f[j], c);
      } else if constexpr (M::ns == 3) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {               // per k chunk: lo.hi, hi.lo, hi.hi (the order of compute_tile)
          c = mma16<F16>(A.f[i][2 + kc], W.f[kc], c);
          c = mma16<F16>(A.f[i][kc], W.f[2 + kc], c);
          c = mma16<F16>(A.f[i][kc], W.f[kc], c);
        }
      } else {
        c = mma16<true>(A.f[i][0], W.f[0], c);
        c = mma16<true>(A.f[i][1], W.f[1], c);
      }
    }
    if constexpr (M::ns == 2) {                         // the correction terms after both half products of the pair (compute_tile's order per accumulator)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int4 a0 = __builtin_bit_cast(int4, A.f[i][2]), a1 = __builtin_bit_cast(int4, A.f[i][3]);
        const int4 w0 = __builtin_bit_cast(int4, W.f[2]), w1 = __builtin_bit_cast(int4, W.f[3]);
        const i32x8 a8 = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const i32x8 w8 = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        acc[2 * a + i][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, acc[2 * a + i][b], /*A e5m2*/ 1, /*B e5m2*/ 1,
                                                                            0, H8_E8M0_LO, 0, H8_E8M0_ONE);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
#define G2_VMWAIT(fill) do { if (fill) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)

  auto run_k8 = [&](auto mode, const int kt0, const int kt1) __attribute__((always_inline)) {
    const int T = kt1 - kt0;
    if (T <= 0) return;
    // ---- prologue: tile 0 whole, A0 / B0 of tile 1; A0(0) and B0(0) have landed for every wave at the barrier
    TileCoord c1 = tile_coord(mode, kt0);               // coordinates of tile t + 1 (in the loop), here: tile 0
    {
      const int st = kt0 & 1;
      issue_half(mode, c1, st, HA0{}); issue_half(mode, c1, st, HB0{}); issue_half(mode, c1, st, HB1{}); issue_half(mode, c1, st, HA1{});
    }
    if (T > 1) {
      c1 = tile_coord(mode, kt0 + 1);
      issue_half(mode, c1, (kt0 + 1) & 1, HA0{}); issue_half(mode, c1, (kt0 + 1) & 1, HB0{});
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#ifdef G2_BLKTRACE
    if (bts[1] == 0) BSTAMP(1);
#endif
    if (wave >= 4) __builtin_amdgcn_s_barrier();        // the second group runs half a phase behind
    AHalf A;
    WHalf W0, W1;
    for (int t = 0; t < T; ++t) {
      const int kt = kt0 + t;
      const unsigned char* sb = smem + (kt & 1) * STAGE;
      const bool more1 = t + 1 < T, more2 = t + 2 < T;
      const TileCoord c2 = more2 ? tile_coord(mode, kt + 2) : c1;
      // ---- phase 0: quadrant (0, 0); request B1 of tile t + 1
      if (wave_active) { load_a(mode, sb, a_row_off, fswz, 0, A); load_w(mode, sb, w_row_off, fswz, 0, W0); }
      if (more1) issue_half(mode, c1, (kt + 1) & 1, HB1{});
      G2_VMWAIT(more1);
      __builtin_amdgcn_s_barrier();
      if (wave_active) mma_quadrant(mode, 0, 0, A, W0);
      __builtin_amdgcn_s_barrier();
      // ---- phase 1: quadrant (0, 1); request A1 of tile t + 1
      if (wave_active) load_w(mode, sb, w_row_off, fswz, 1, W1);
      if (more1) issue_half(mode, c1, (kt + 1) & 1, HA1{});
      G2_VMWAIT(more1);
      __builtin_amdgcn_s_barrier();
      if (wave_active) mma_quadrant(mode, 0, 1, A, W1);
      __builtin_amdgcn_s_barrier();
      // ---- phase 2: quadrant (1, 1); request A0 of tile t + 2 (the rows of A0(t) were last read in phase 0)
      if (wave_active) load_a(mode, sb, a_row_off, fswz, 1, A);
      if (more2) issue_half(mode, c2, kt & 1, HA0{});
      G2_VMWAIT(more2);
      __builtin_amdgcn_s_barrier();
      if (wave_active) mma_quadrant(mode, 1, 1, A, W1);
      __builtin_amdgcn_s_barrier();
      // ---- phase 3: quadrant (1, 0); request B0 of tile t + 2
      if (more2) issue_half(mode, c2, kt & 1, HB0{});
      G2_VMWAIT(more2);
      __builtin_amdgcn_s_barrier();
      if (wave_active) mma_quadrant(mode, 1, 0, A, W0);
      __builtin_amdgcn_s_barrier();
      c1 = c2;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();         // wait for the lagging group's last MFMA phase: LDS is free after this
  };

  // ---- tap-shared causal conv (3 taps, dilation 1, utterances aligned to the row tile: the FF conv and the Wavenet init conv) on
  // the phased schedule.  The three taps of a K chunk read the SAME 258 input rows shifted by one: instead of streaming three
  // 256-row A tiles per chunk, one 264-row tile A'(it) (rows m0 - 2 ...) is loaded once per chunk and the fragment reads address
  // rows r + tap; only the W tile changes per step.  LDS: two A' buffers (33 KiB each) + two W buffers (32 KiB each); DMA per
  // three steps: 33 + 96 wave-instructions instead of 192 (+10 % on the FF conv, bit-identical).  Rows before the utterance start
  // are zero-page lanes, as in the standard path.  One step = one (k chunk, tap): A fragments come from A'(it) (rows shifted by
  // the tap), W from W(it, tap).  Per step four phases as in run_k8; requests:
  //   phase 0: W.B1 of step + 1      phase 3: W.B0 of step + 2      (a W half tile is requested 5 phases before its first read)
  //   phases 1, 2 of tap 0 and phase 1 of tap 1: the three 11-piece parts of A'(it + 1) (a whole `it` ahead, other buffer)
  // A wave issues 2 pieces per W request and 1 or 2 per A' part (33 = 3 x 11 row groups over 8 waves), so the counted waits
  // differ per wave and tap: before phase 0 of step s + 1 and before phase 1 of step s the wave allows exactly the requests
  // younger than the half tile it is about to read: 4 (two W halves) + its A' pieces of the previous step.
  // `helper` = this wave is one of waves 4-7 of a column tile that is at most half valid (the last tile of the FF conv, N = 1365 =
  // 5 tiles + 85 columns).  W rows 128 ... 255 feed only the column quarters 2 and 3 = those waves, which have nothing to compute --
  // and exactly those waves request those rows.  So they take a loop of their own: their share of the A' pieces, the counted waits
  // for them and every barrier of the schedule below, but no W request (11 % of the FF conv's operand line requests were for
  // columns nobody stores), no fragment read, no MFMA.  LDS rows that are never written are never read: results are unchanged.
  // A separate loop because the same test as a run-time flag inside the common loop split its phases into more basic blocks and cost
  // every other GEMM 2 %, and a second instantiation of the whole loop spilled (profiles/r05_wavenet_dense_and_wskip_ab.txt).
  auto run_k8_conv3 = [&](auto mode, const bool helper) __attribute__((always_inline)) {
    using M = decltype(mode);
    constexpr int BK = M::bk;
    constexpr int A_BUF = 264 * RB, W_BUF = REGION;
    unsigned char* const sA = smem;
    unsigned char* const sW = smem + 2 * A_BUF;
    const int tpt = tiles_per_tap(mode);
    const int nst = 3 * tpt;
    const bool half_tail = !M::line32 && (g.kt_per_tap & 1);
    const int m0 = tm * G2_BM;
    const int n0 = m0 % g.seq_len;
    const bf16_t* a_base = g.a_hi + pcol((int)(z * g.a_zs), ail) + (long)(m0 - 2) * a_rs;
    const bf16_t* w_base = g.w_hi + (((long)z * g.w_zs) << (wil ? 1 : 0)) + ((long)tn * G2_BN) * w_rs;
    const int lc0 = pchunk ^ (lrow >> 1), lc1 = pchunk ^ (4 + (lrow >> 1));
    const int cofA0 = M::line32 ? lc0 * 8 : pcol(lc0 * 8, ail), cofA1 = M::line32 ? lc1 * 8 : pcol(lc1 * 8, ail);
    const int cofW0 = M::line32 ? lc0 * 8 : pcol(lc0 * 8, wil), cofW1 = M::line32 ? lc1 * 8 : pcol(lc1 * 8, wil);
    const bool chi0 = lc0 >= 4, chi1 = lc1 >= 4;
    const int nA = wave < 3 ? 2 : 1;                          // A' pieces of this wave per 11-piece part

    auto issue_wh = [&](int st, int b) __attribute__((always_inline)) {        // this wave's 2 pieces of half tile B<b> of W(step st)
      const int it = st / 3, tap = st - 3 * it;
      const bool half = half_tail && (it == tpt - 1);
      const long off = pcol(tap * tap_k + it * BK, wil);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * wave + e;
        const int rg = (k & 3) + 8 * (k >> 2) + 4 * b;       // parity == e
        const bf16_t* p = (half && (e ? chi1 : chi0)) ? zero_page : (w_base + (long)(8 * rg + lrow) * w_rs + off + (e ? cofW1 : cofW0));
        glds16(p, sW + (st & 1) * W_BUF + rg * 1024);
      }
    };
    auto issue_ap = [&](int it, int part) __attribute__((always_inline)) {      // this wave's 1-2 of the 11 pieces of part `part` of A'(it)
      const bool half = half_tail && (it == tpt - 1);
      const long off = pcol(it * BK, ail);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && wave >= 3) break;
        const int j = 11 * part + 8 * e + wave;              // row group [0, 33)
        const bool par = j & 1;
        const int row = 8 * j + lrow;                        // A' row: input row m0 - 2 + row
        const bool ok = (n0 + row - 2 >= 0) && ((long)m0 - 2 + row < g.M) && !(half && (par ? chi1 : chi0));
        glds16(ok ? (a_base + (long)row * a_rs + off + (par ? cofA1 : cofA0)) : zero_page, sA + (it & 1) * A_BUF + j * 1024);
      }
    };
    auto vmwait = [&](int n) __attribute__((always_inline)) {                   // wave-uniform n in {0, 4, 5, 6, 8}; {0, 1, 2} in the helper loop
      if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if (helper) {
      // ---- the helper loop (see above): the same requests of A' pieces in the same phases, the same barriers, nothing else
      issue_ap(0, 0); issue_ap(0, 1); issue_ap(0, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();                             // waves 4-7 run half a phase behind
      for (int it = 0; it < tpt; ++it) {
        const bool next_it = it + 1 < tpt;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          const int st = 3 * it + tap;
          const bool more1 = st + 1 < nst, more2 = st + 2 < nst;
          const int a_prev = (tap == 0) ? 0 : (next_it ? (tap == 1 ? 2 * nA : nA) : 0);
          const int a_this = next_it ? (tap == 0 ? 2 * nA : (tap == 1 ? nA : 0)) : 0;
          vmwait(more1 ? a_prev : 0);                            // phase 0
          __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
          if (next_it && tap == 0) issue_ap(it + 1, 0);           // phase 1
          if (next_it && tap == 1) issue_ap(it + 1, 2);
          __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
          if (next_it && tap == 0) issue_ap(it + 1, 1);           // phase 2
          __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
          vmwait(more2 ? a_this : 0);                            // phase 3
          __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
        }
      }
      return;
    }
    // ---- prologue: A'(0), W(0), B0 of W(1)
    issue_ap(0, 0); issue_ap(0, 1); issue_ap(0, 2);
    issue_wh(0, 0); issue_wh(0, 1);
    if (nst > 1) { issue_wh(1, 0); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef G2_BLKTRACE
    if (bts[1] == 0) BSTAMP(1);
#endif
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    const int w_off = (wn * 64 + l31) * RB;
    AHalf A;
    WHalf W0, W1;
    for (int it = 0; it < tpt; ++it) {
      const bool next_it = it + 1 < tpt;
      const unsigned char* sa = sA + (it & 1) * A_BUF;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int st = 3 * it + tap;
        const unsigned char* sw = sW + (st & 1) * W_BUF;
        const bool more1 = st + 1 < nst, more2 = st + 2 < nst;
        const int a_off = (wm * 128 + l31 + tap) * RB;
        const int fz_a = ((l31 + tap) >> 1) & 7;
        // A' pieces this wave issued in the previous step / issues in this one (younger than the W half tiles it waits for)
        const int a_prev = (tap == 0) ? 0 : (next_it ? (tap == 1 ? 2 * nA : nA) : 0);
        const int a_this = next_it ? (tap == 0 ? 2 * nA : (tap == 1 ? nA : 0)) : 0;
        // ---- phase 0: quadrant (0, 0); request B1 of W(step + 1); B1 of this step must have landed for phase 1
        if (wave_active) { load_a(mode, sa, a_off, fz_a, 0, A); load_w(mode, sw, w_off, fswz, 0, W0); }
        if (more1) issue_wh(st + 1, 1);
        vmwait(more1 ? 4 + a_prev : 0);
        __builtin_amdgcn_s_barrier();
        if (wave_active) mma_quadrant(mode, 0, 0, A, W0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: quadrant (0, 1); request a part of A'(it + 1)
        if (wave_active) load_w(mode, sw, w_off, fswz, 1, W1);
        if (next_it && tap == 0) issue_ap(it + 1, 0);
        if (next_it && tap == 1) issue_ap(it + 1, 2);
        __builtin_amdgcn_s_barrier();
        if (wave_active) mma_quadrant(mode, 0, 1, A, W1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (1, 1)
        if (wave_active) load_a(mode, sa, a_off, fz_a, 1, A);
        if (next_it && tap == 0) issue_ap(it + 1, 1);
        __builtin_amdgcn_s_barrier();
        if (wave_active) mma_quadrant(mode, 1, 1, A, W1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (1, 0); request B0 of W(step + 2); B0 of step + 1 (and, before tap 0, A'(it + 1)) must have landed
        if (more2) issue_wh(st + 2, 0);
        vmwait(more2 ? 4 + a_this : 0);
        __builtin_amdgcn_s_barrier();
        if (wave_active) mma_quadrant(mode, 1, 0, A, W0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
  };

  if constexpr (EPI == EPI_WAVENET) {
    // phase 1: the taps before mid_kt (dilated conv), phase 2: the rest (res_conv on the unshifted input)
    const int mid_tap = (g.mid_kt > 0) ? g.mid_kt / g.kt_per_tap : 0;
    const bool mid_uni = g.seq_len > 0 && (g.seq_len & 127) == 0;      // row_base % 128 == 0: the wave tile lies inside one utterance
    // the one-barrier-per-tile loop: the phased schedule measured 9 % SLOWER on this kernel (two short K phases = two pipeline
    // ramps per block, and the K loops of this kernel family are bound by LDS-DMA throughput, not by its latency: DESIGN.md)
    run_k(ModeP1{}, 0, mid_tap * tiles_per_tap(ModeP1{}));
    wavenet_midgate<4, 2>(acc, g, z, row_base, col_base, l31, hi, mid_uni);
    run_k(ModeMain{}, mid_tap * tiles_per_tap(ModeMain{}), ntaps * tiles_per_tap(ModeMain{}));
  } else if constexpr (EPI == EPI_SPLIT) {
    const bool conv3 = g.conv_taps == 3 && ntaps == 3 && !g.dil_z && g.dil == 1 && g.pad_left < 0 && g.seq_len > 0 &&
                       (g.seq_len % G2_BM) == 0;
    if (conv3) {
      run_k8_conv3(ModeMain{}, /*helper=*/wave >= 4 && tn * G2_BN + G2_BN / 2 >= ncols_needed);
    } else {
      run_k8(ModeMain{}, 0, ntaps * tiles_per_tap(ModeMain{}));
    }
  } else {
    run_k8(ModeMain{}, 0, ntaps * tiles_per_tap(ModeMain{}));
  }
  BSTAMP(2);
  g2_block_epilogue<NSPLIT, EPI, F16>(acc, g, z, tm, tn, wave, lane, smem);
#ifdef G2_BLKTRACE
  BSTAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BSTAMP(4);
  if (lane == 0 && blockIdx.x < G2_BLK_MAX) {
    unsigned long long* o = g2_blk + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = bts[0]; o[1] = bts[1]; o[2] = bts[2]; o[3] = bts[3]; o[4] = bts[4];
    o[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    o[6] = __builtin_readcyclecounter() - cyc0;
    o[7] = (unsigned long long)wave_active;
  }
#endif
}

#ifdef G2_TRACE
extern "C" int ns2_debug_read_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2_trace), sizeof(unsigned long long) * 8 * 64 * 10);
}
#endif
#ifdef G2_BLKTRACE
extern "C" int ns2_debug_read_blocks(unsigned long long* out, int nblk) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g2_blk), sizeof(unsigned long long) * 64 * (size_t)nblk);
}
#endif

template <int NSPLIT, int EPI, bool F16, int P1 = 0, bool TR = false>
static hipError_t launch2_one(const GemmArgs& g, hipStream_t s) {
  const int ntn = (g.N + G2_BN - 1) / G2_BN, ntm = (g.M + G2_BM - 1) / G2_BM;
  const int nz = g.nz > 0 ? g.nz : 1;
  const size_t lds = 8 * EPI_LDS_WAVE_BYTES;          // 144 KiB: 2 x 64 KiB K stages, reused as 8 x 18 KiB epilogue regions
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(reinterpret_cast<const void*>(&gemm2_kernel<NSPLIT, EPI, F16, P1, TR>), (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((gemm2_kernel<NSPLIT, EPI, F16, P1, TR>), dim3(ntn * ntm * nz), dim3(512), lds, s, g);
  return hipGetLastError();
}

// Weight gradient straight from the token-major operand planes (the TR kernel above): out_f[z][r][n] = sum over the tokens of K
// slice z of dY[m, r] X[m - shift(n), k(n)].  g: a = dY planes [tr_tokens, lda], w = X planes [tr_tokens, ldw], M = R output rows,
// N = taps * tr_kp output columns, nkt = 32-token K tiles PER SLICE, nz = slices, out_f / out_f_zs / ldo_f = the fp32 slots.
hipError_t launch_gemm_tr(const GemmArgs& g, int precision, hipStream_t s) {
  if (precision != 3 && precision != 4) return hipErrorInvalidValue;
  if (!g.a_hi || g.a_lo != g.a_hi + 32 || !g.w_hi || g.w_lo != g.w_hi + 32 || !g.out_f) return hipErrorInvalidValue;
  if (g.M <= 0 || g.N <= 0 || g.nkt <= 0 || g.kt_per_tap != g.nkt || g.conv_taps != 0 || g.epi != EPI_F32 || g.nz < 1) return hipErrorInvalidValue;
  if (g.tr_tokens <= 0 || g.tr_kp <= 0 || (g.tr_kp & 31) || (g.lda & 31) || (g.ldw & 31) || g.tr_taps < 0) return hipErrorInvalidValue;
  if (g.tr_taps > 1 && (g.seq_len < 32 || g.dil < 1)) return hipErrorInvalidValue;      // the in-tile utterance position wraps at most once
  if ((long)g.nz * g.nkt * 32 < g.tr_tokens) return hipErrorInvalidValue;                // the slices cover every token
  return precision == 3 ? launch2_one<3, EPI_F32, false, 0, true>(g, s) : launch2_one<2, EPI_F32, true, 0, true>(g, s);
}

template <int NSPLIT, bool F16>
static hipError_t launch2_epi(const GemmArgs& g, hipStream_t s) {
  switch (g.epi) {
    case EPI_F32: return launch2_one<NSPLIT, EPI_F32, F16>(g, s);
    case EPI_SPLIT: return launch2_one<NSPLIT, EPI_SPLIT, F16>(g, s);
    case EPI_QKV: return launch2_one<NSPLIT, EPI_QKV, F16>(g, s);
    case EPI_GEGLU: return launch2_one<NSPLIT, EPI_GEGLU, F16>(g, s);
    case EPI_WAVENET:
      if constexpr (NSPLIT == 2) { if (g.p1_half) return launch2_one<2, EPI_WAVENET, true, 1>(g, s); }
      return launch2_one<NSPLIT, EPI_WAVENET, F16>(g, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm1(const GemmArgs& g, int precision, hipStream_t s);   // gemm.hip (128x128 register-staged kernel)
hipError_t launch_gemm_splitk(const GemmArgs& g, int precision, int S, int c, hipStream_t s);   // gemm.hip: K slices + finish launch

// Test hook (ns2_debug_force_gemm / NS2_GEMM): the only switch of the GEMM family, process-wide by design, atomic so that two
// host threads (one per device) may read it while a test flips it.  -1: read NS2_GEMM once; 0 auto; 1 / 2 force a kernel.
static std::atomic<int> g_forced_kernel{-1};
void force_gemm_kernel(int k) { g_forced_kernel.store(k, std::memory_order_relaxed); }
static int forced_kernel() {
  int f = g_forced_kernel.load(std::memory_order_relaxed);
  if (f < 0) {
    const char* e = getenv("NS2_GEMM");
    f = e ? atoi(e) : 0;
    g_forced_kernel.store(f, std::memory_order_relaxed);
  }
  return f;
}

// Split-K plan of a product M x N over nkt K tiles of 32 (kt_per_tap per tap): S slices of c K tiles of EVERY tap, S = 1 = do not
// split.  A product is split when it has fewer 128 x 128 output tiles than the chip has CUs -- a handful of tiles, each a long
// serial K loop on one CU while the others idle -- and a K loop long enough to pay for the second launch: K >= 512, or K >= 352
// when the epilogue is fp32 (a flat 16-byte-vector finishing kernel of ~5 us).  Measured at 1 x 1024 frames
// (tools/exp_small_m_kernel.py): the FF causal conv (K = 4128) 114 -> 55 us, FF-out (K = 1376) 71 -> 22 us; the K = 512 products are
// neutral to slightly ahead.  Slices: enough for ~2 blocks per CU, at least 4 K tiles of every tap each, never an empty one, and
// all slots inside the lent scratch.  Pure host arithmetic (ns2_debug_splitk_plan exposes it to the CPU tests).
void splitk_plan(int M, int N, int nkt, int kt_per_tap, bool epi_f32, long scratch_floats, int* S_out, int* c_out) {
  *S_out = 1; *c_out = kt_per_tap;
  if (M <= 0 || N <= 0 || kt_per_tap <= 0 || nkt < kt_per_tap) return;
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles >= 256 || kt_per_tap < 8 || nkt < (epi_f32 ? 11 : 16)) return;
  const int want = (int)std::min<long>(512 / tiles, kt_per_tap / 4);
  if (want < 2) return;
  const int c = (kt_per_tap + want - 1) / want, S = (kt_per_tap + c - 1) / c;
  if (S < 2 || (long)S * M * ((N + 63) & ~63) > scratch_floats) return;
  *S_out = S; *c_out = c;
}

// Dispatch: the 256x256 LDS-DMA kernel for wide outputs, the 128x128 kernel when N <= 128 (half of a 256-wide tile
// would be padding, e.g. the dim=128 model's d x d projections).  W rows are padded to 256 by the packers.
// Does this product take the one-launch 128 x 128 kernel whose workgroups own whole rows (N == 128), so that its fp32 epilogue can run
// the RMSNorm that follows (GemmArgs::nrm_*)?  Not when K is split (the finishing launch runs the epilogue) or the big kernel is forced.
// Two cases: (a) K is split (small batches): the flat finishing launch owns whole rows whatever the tile -- N = 128, 256 or 512;
// (b) one launch of the 128 x 128 kernel with N == 128 == BN and M % 128 == 0 (the dim = 128 model at full batch).
bool gemm_fuses_norm(const GemmArgs& g, int precision) {
  if (g.epi != EPI_F32 || g.M <= 0 || g.act != 0 || g.nz > 1 || g.ksplit != 0 || g.dil_z) return false;
  if ((g.ldo_f & 3) || (reinterpret_cast<uintptr_t>(g.out_f) & 15) || (g.resid && ((g.ldr & 3) || (reinterpret_cast<uintptr_t>(g.resid) & 15)))) return false;
  const int f = forced_kernel();
  if ((f == 0 || f == 4) && g.sk_ws) {
    int S, c;
    splitk_plan(g.M, g.N, g.nkt, g.kt_per_tap, true, g.sk_ws_floats, &S, &c);
    if (S >= 2) return g.N == 128 || g.N == 256 || g.N == 512;
  }
  (void)precision;
  return f != 2 && g.N == 128 && (g.M % 128) == 0;
}

hipError_t launch_gemm(const GemmArgs& g_in, int precision, hipStream_t s) {
  GemmArgs g = g_in;
  if (precision < 1 || precision > 4) return hipErrorInvalidValue;
  if (g.nrm_hi) {
    const bool nil = g.nrm_lo != nullptr;
    if (!gemm_fuses_norm(g, precision) || !planes_ok(g.nrm_hi, g.nrm_lo) || (g.nrm_ld & (nil ? 31 : 3)) || g.nrm_ld < g.N ||
        (g.nrm_fmt == FMT_H8 && !nil) || (g.nrm_fmt == FMT_F16 && nil) || (g.nrm_cond && ((g.nrm_cond_ld & 3) || (reinterpret_cast<uintptr_t>(g.nrm_cond) & 15))) ||
        (g.nrm_gamma && (reinterpret_cast<uintptr_t>(g.nrm_gamma) & 15)))
      return hipErrorInvalidValue;
  }
  const int op_fmt = precision == 2 ? FMT_F16 : (precision == 4 ? FMT_H8 : FMT_BF16);
  if (g.out_fmt < 0) g.out_fmt = op_fmt;
  if (g.vt_fmt < 0) g.vt_fmt = (precision == 2 || precision == 4) ? FMT_F16 : FMT_BF16;
  // operand / output plane pointers must match the formats: IEEE-half planes are dense (no lo pointer), FMT_H8 and the
  // bf16 x3 operands are interleaved lines (lo == hi + 32)
  if (precision == 2 && (g.a_lo || g.w_lo)) return hipErrorInvalidValue;
  if ((precision == 3 || precision == 4) && (!g.a_lo || !g.w_lo)) return hipErrorInvalidValue;
  if (g.out_hi && ((g.out_fmt == FMT_F16 && g.out_lo) || (g.out_fmt == FMT_H8 && !g.out_lo))) return hipErrorInvalidValue;
  if (g.vt_hi && g.vt_fmt == FMT_F16 && g.vt_lo) return hipErrorInvalidValue;
  if (g.M <= 0 || g.N <= 0 || g.nkt <= 0 || g.kt_per_tap <= 0 || (g.nkt % g.kt_per_tap)) return hipErrorInvalidValue;
  // a lo plane means the interleaved layout: lo = hi + 32 (ns2_common.h)
  if (!planes_ok(g.a_hi, g.a_lo) || !planes_ok(g.w_hi, g.w_lo) || !planes_ok(g.out_hi, g.out_lo) ||
      !planes_ok(g.vt_hi, g.vt_lo))
    return hipErrorInvalidValue;
  const int f = forced_kernel();
  if (f == 5 && ffconv3_eligible(g, precision)) return launch_ffconv3(g, s);      // test hook: the dedicated kernels whatever the size
  if (f == 5 && gemm3_eligible(g, precision)) return launch_gemm3(g, s);
  if (f == 5 && wavenet3_eligible(g, precision)) return launch_wavenet3(g, s);
  // Small products (a batch of 1 ... 4 utterances) split K when the caller lent scratch (splitk_plan above).
  // (f == 3: automatic kernel choice, never split -- A/B hook.)
  if ((f == 0 || f == 4) && g.sk_ws && g.epi != EPI_WAVENET && g.nz <= 1 && !g.dil_z && g.ksplit == 0) {
    int S, c;
    splitk_plan(g.M, g.N, g.nkt, g.kt_per_tap, g.epi == EPI_F32, g.sk_ws_floats, &S, &c);
    if (S >= 2) return launch_gemm_splitk(g, precision, S, c, s);
  }
  // a product that would put at most 64 blocks of 256 x 256 on the 256 CUs runs on the 128 x 128 kernel (4 x the blocks, a
  // quarter of the serial work each): 1 x 1024-frame steps measured 4 % faster with it, 4 x 1024 slower (exp_small_m_kernel.py)
  const long blocks256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.nz > 0 ? g.nz : 1);
  const bool big = (f == 2) || (f != 1 && g.N > 128 && !((f == 0 || f >= 3) && blocks256 <= 64));
  if (!big) return launch_gemm1(g, precision, s);
  // the FF causal conv of the one-half-product plans on full row tiles: its own kernel (ffconv_kernel.h)
  if ((f == 0 || f == 3) && ffconv3_eligible(g, precision)) return launch_ffconv3(g, s);
  // the mixed linear products on full row tiles: the lean kernel (gemm3_kernel.h)
  if ((f == 0 || f == 3) && gemm3_eligible(g, precision)) return launch_gemm3(g, s);
  if ((f == 0 || f == 3) && wavenet3_eligible(g, precision)) return launch_wavenet3(g, s);
  switch (precision) {
    case 3: return launch2_epi<3, false>(g, s);
    case 4: return launch2_epi<2, true>(g, s);
    case 2: return launch2_epi<1, true>(g, s);
    default: return launch2_epi<1, false>(g, s);
  }
}

size_t ffconv3_tiled_bytes_of(int N, int Cp) { return ffconv3_tiled_bytes(N, Cp); }
hipError_t ffconv3_build_tiles(const bf16_t* w_hi, int ldw, int Cp, int rows_p, int N, bf16_t* out, hipStream_t s) {
  return launch_ffconv3_tile(w_hi, ldw, Cp, rows_p, N, out, s);
}
int ffconv3_lda(int Cp) { return ffconv3_tiles_per_tap(Cp) * 64; }
size_t gemm3_tiled_bytes_of(int rows_p, int nkt) { return gemm3_tiled_bytes(rows_p, nkt); }
hipError_t gemm3_build_tiles(const bf16_t* w_hi, int ldk, int rows_p, bf16_t* out, hipStream_t s) { return launch_gemm3_tile(w_hi, ldk, rows_p, out, s); }
size_t wavenet3_tiles_bytes(int rows_p, int dp, int nz, int phase) { return phase == 1 ? wavenet3_tiles1_bytes(rows_p, dp, nz) : wavenet3_tiles2_bytes(rows_p, dp, nz); }
hipError_t wavenet3_build_tiles(const bf16_t* w_hi, int rows_p, int dp, int nz, bf16_t* t1, bf16_t* t2, hipStream_t s) { return launch_wavenet3_tiles(w_hi, rows_p, dp, nz, t1, t2, s); }

NS2_DEFINE_SATURATION_READER(gemm2)

}  // namespace ns2
