// Software-pipelined 256x256 GEMM / causal-conv kernel for gfx950: one wave per SIMD, MFMA and LDS overlapped in-wave.
//
// Same contract, LDS image, DMA addressing and epilogues as gemm2.hip.  What changes is who hides what:
// gemm2 runs 8 waves (2 per SIMD) that all read fragments, wait, then all issue MFMAs -- measured on MI355X its time is
// the SUM of the MFMA pipe time (640 us on the FF conv), the LDS fragment reads and the DMA issue (ablations in
// DESIGN.md), i.e. nothing overlaps.  Here:
//   * 4 waves (2 x 2), wave tile 128x128 = 4x4 v_mfma_f32_32x32x16_bf16 accumulators (256 AGPRs; one wave per SIMD owns
//     all 512 registers): 16 fragment reads feed 48 MFMAs per 16-deep K step (gemm2: 12 reads per 24 MFMAs), so the CU's
//     LDS read traffic per FLOP drops by a third;
//   * fragments are double-buffered in registers: the ds_reads of K step s+1 are interleaved between the MFMAs of step s
//     (`sched_group_barrier` pins the order: 3 MFMAs, 1 read, ...), never a read burst followed by a wait;
//   * one barrier per K tile, placed a third of the way into the tile's last K step: by then every wave has finished
//     reading the current stage, so the DMA of tile t+2 is issued right there (into the stage being retired) and is
//     interleaved with the remaining MFMAs together with the first fragment reads of tile t+1 -- a DMA has a whole tile
//     (~3k MFMA cycles) to land, and its issue slots hide under MFMA execution as well.
#include <cstdlib>
#include <type_traits>

#include "gemm_epi.h"

namespace ns2 {

constexpr int G3_BM = 256, G3_BN = 256;

typedef __attribute__((address_space(3))) void lds3_void_t;
typedef const __attribute__((address_space(1))) void gbl3_void_t;

NS2_DEVINL void dma16(const void* gsrc, unsigned char* ldst) {
  __builtin_amdgcn_global_load_lds((gbl3_void_t*)gsrc, (lds3_void_t*)ldst, 16, 0, 0);
}

// One MFMA as volatile asm with the accumulator pinned to AGPRs ("+a"): (1) the 16 accumulators own all 256 AGPRs and are
// updated in place -- the register allocator cannot do that reliably when it is also free to reorder (it spilled 1.2k
// VGPRs on the builtin version); (2) volatile asm statements keep their program order and order all memory operations
// around them, so the C++ fragment reads and LDS-DMA calls written between two MFMAs are issued exactly there, while the
// compiler still tracks their results (s_waitcnt lgkmcnt / M0 handling stay automatic).
NS2_DEVINL void mfma_acc(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// MFMA number idx of a K step.  Product-major: 16 independent accumulators between two MFMAs on the same one; per
// accumulator the order is a_lo*w_hi, a_hi*w_lo, a_hi*w_hi -- as in gemm.hip / gemm2.hip, so the kernels agree bit for bit.
template <int NSPLIT, int NP>
NS2_DEVINL void mfma_step(const int idx, const bf16x8 (&af)[NP][4], const bf16x8 (&wf)[NP][4], f32x16 (&acc)[4][4]) {
  const int prod = idx >> 4, mi = (idx & 15) >> 2, ni = idx & 3;
  if constexpr (NSPLIT == 3) {
    if (prod == 0) mfma_acc(acc[mi][ni], af[1][mi], wf[0][ni]);
    else if (prod == 1) mfma_acc(acc[mi][ni], af[0][mi], wf[1][ni]);
    else mfma_acc(acc[mi][ni], af[0][mi], wf[0][ni]);
  } else {
    mfma_acc(acc[mi][ni], af[0][mi], wf[0][ni]);
  }
}

template <int NSPLIT, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm3_kernel(const GemmArgs g, const bf16_t* __restrict__ zero_page) {
  constexpr int NP = (NSPLIT == 3) ? 2 : 1;          // planes per operand
  constexpr int BK = (NSPLIT == 3) ? 32 : 64;        // K-tile depth (elements)
  constexpr int RB = BK * 2;                         // LDS row bytes (64 / 128)
  constexpr int CPR = RB / 16;                       // 16-B chunks per row (4 / 8)
  constexpr int RPI = 64 / CPR;                      // tile rows moved by one DMA wave-instruction (16 / 8)
  constexpr int PLANE = G3_BM * RB;                  // 16 KiB / 32 KiB
  constexpr int STAGE = 2 * NP * PLANE;              // 64 KiB
  constexpr int KCH = BK / 16;                       // 16-deep MFMA K steps per tile (2 / 4)
  constexpr int IPP = G3_BM / RPI;                   // DMA instructions per plane (16 / 32)
  constexpr int NDMA = 16;                           // DMA instructions per wave per K tile
  constexpr int NM = 16 * NSPLIT;                    // MFMAs per wave per K step (48 / 16)
  constexpr int NR = 8 * NP;                         // fragment reads per wave per K step (16 / 8)
  constexpr int PA = (NSPLIT == 3) ? 16 : 4;         // MFMAs of the last K step issued before the tile barrier
  constexpr int MPR = (NSPLIT == 3) ? 2 : 1;         // steady K step: one fragment read after every MPR-th MFMA
  constexpr int RPM = (NSPLIT == 3) ? 1 : 2;         // last K step, after the barrier: fragment reads per MFMA ...
  constexpr int TM = KCH * NM;                       // MFMAs per wave per K tile (96 / 64)
  constexpr int B0 = (KCH - 1) * NM + PA;            // index of the first MFMA after the tile barrier (64 / 52)
  // DMA pacing.  The CU's vector-memory front end retires one cache-line request every ~2 clocks, i.e. ~36 clocks per
  // 16-line DMA instruction (exact mode: 64-B row segments) or ~18 (fast mode: 128-B segments), and a wave that issues
  // faster than that blocks AT ISSUE -- with its MFMAs queued behind the blocked instruction (measured: a burst of 64
  // DMA instructions per tile costs ~2300 clocks of MFMA time).  So DMA instruction i of the tile after next is issued
  // after MFMA B0 + SP*i, running on into the next tile: 4 waves x 1 per (SP x 32) clocks stays below the retire rate.
#ifdef G3_SP
  constexpr int SP = G3_SP;
#else
  constexpr int SP = (NSPLIT == 3) ? 5 : 3;
#endif
  static_assert(NR * MPR <= NM - 8 && NR / RPM <= NM - PA, "fragment read schedule");
  static_assert(B0 + SP * (NDMA - 1) - TM < B0 - 8, "DMA of tile t+1 must be issued well before tile t's barrier");
  static_assert(2 * NP * IPP == 4 * NDMA, "4 waves x 16 DMA instructions per K-tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int ntn = (g.N + G3_BN - 1) / G3_BN;
  const int ntm = (g.M + G3_BM - 1) / G3_BM;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % ntn;
  bid /= ntn;
  const int tm = bid % ntm;
  const int z = bid / ntm;
  const int dil = g.dil_z ? (g.dil << z) : g.dil;

  // ---- DMA roles: waves 0,1 stream A, waves 2,3 stream W; instruction j = (wave&1)*16 + i inside the operand
  const bool a_wave = wave < 2;
  const int lrow = lane / CPR, pchunk = lane % CPR;
  const bf16_t* src[NDMA];     // per-instruction source pointer at K offset 0 (A: unshifted row)
  int nseq[NDMA];              // A only: position inside the utterance (causal zero fill); -1 = row >= M
  int ldst[NDMA];              // LDS byte offset inside a stage (wave-uniform)
  unsigned chunk_hi_mask = 0;  // bit i: this lane fetches one of the upper 32 columns of a 64-deep tile
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int j = (wave & 1) * NDMA + i;             // [0, NP*IPP)
    const int plane = j / IPP, rg = j % IPP;
    const int row = rg * RPI + lrow;                 // tile row
    const int swz = (NSPLIT == 3) ? ((row >> 2) & 3) : ((row >> 1) & 7);
    const int lchunk = pchunk ^ swz;                 // logical 16-B chunk this lane fetches
    ldst[i] = (a_wave ? 0 : NP * PLANE) + plane * PLANE + rg * 1024;
    if (lchunk >= 4) chunk_hi_mask |= 1u << i;
    if (a_wave) {
      const long m = (long)tm * G3_BM + row;
      const bf16_t* base = (plane == 0 ? g.a_hi : g.a_lo) + (long)z * g.a_zs;
      src[i] = base + m * g.lda + lchunk * 8;
      nseq[i] = (m < g.M) ? ((g.seq_len > 0) ? (int)(m % g.seq_len) : 0x3fffffff) : -1;
    } else {
      const bf16_t* base = (plane == 0 ? g.w_hi : g.w_lo) + (long)z * g.w_zs;
      src[i] = base + ((long)tn * G3_BN + row) * g.ldw + lchunk * 8;
      nseq[i] = 0;
    }
  }

  // K tiling in BK units (see gemm2.hip): with BK = 64 an odd 32-multiple tap ends in a half tile (upper half zero-filled)
  const int tap_k = g.kt_per_tap * 32;
  const int tpt = (tap_k + BK - 1) / BK;
  const bool half_tail = (NSPLIT != 3) && (g.kt_per_tap & 1);
  const int ntaps = g.nkt / g.kt_per_tap;
  const int ntiles = ntaps * tpt;
  const int mid_tile = (g.mid_kt > 0) ? (g.mid_kt / g.kt_per_tap) * tpt : 0;
  const int pl = g.pad_left < 0 ? g.conv_taps - 1 : g.pad_left;          // causal: all padding on the left (NS2:583-595)
  const unsigned slim = g.seq_len > 0 ? (unsigned)g.seq_len : 0x7fffffffu;

  // per-tile DMA context (wave-uniform), then one instruction at a time so the K loop can interleave them with MFMAs
  struct DmaCtx { long off; int sbase; int shift; int half; };      // sbase = LDS byte offset of the stage (no padding:
                                                                    // a loop-carried copy must stay in SGPRs)
  auto dma_ctx = [&](int kt, int stage) {
    DmaCtx c;
    c.sbase = stage * STAGE;
    const int tap = kt / tpt;
    const int it = kt - tap * tpt;
    c.half = half_tail && (it == tpt - 1);
    if (a_wave) {
      c.shift = (tap < g.conv_taps) ? (pl - tap) * dil : 0;
      c.off = (long)it * BK - (long)c.shift * g.lda;
    } else {
      c.shift = 0;
      c.off = (long)tap * tap_k + (long)it * BK;
    }
    return c;
  };
  auto dma_one = [&](const DmaCtx& c, const int i) {
    bool ok = !(c.half && ((chunk_hi_mask >> i) & 1));
    if (a_wave) ok = ok && ((unsigned)(nseq[i] - c.shift) < slim);
    const bf16_t* p = ok ? (src[i] + c.off) : zero_page;
#if defined(G3_ABL) && (G3_ABL & 16)                 // timing only: every DMA instruction reads one contiguous 1 KiB
    p = (a_wave ? g.a_hi + (long)tm * G3_BM * g.lda : g.w_hi + (long)tn * G3_BN * g.ldw) +
        (((int)(c.off & 0xfff) * 64 + i * 512 + (wave & 1) * 8192) & 0x3ffff) + lane * 8;
#endif
    dma16(p, smem + c.sbase + ldst[i]);
  };
  auto issue_tile = [&](int kt, int stage) {
    const DmaCtx c = dma_ctx(kt, stage);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_one(c, i);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int row_base = tm * G3_BM + wm * 128;
  const int col_base = tn * G3_BN + wn * 128;
  const int ncols_needed = (EPI == EPI_GEGLU || EPI == EPI_F32 || EPI == EPI_QKV) ? g.N : max(g.N, g.out_ncols);
  const bool wave_active = col_base < ncols_needed;   // else: stream operands and join barriers only

  // fragment read addressing: row = wave base + 32*i + l31 ; physical chunk = (2*kc + hi) ^ swz(row)
  const int fswz = (NSPLIT == 3) ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
  const int a_row_off = (wm * 128 + l31) * RB;
  const int w_row_off = NP * PLANE + (wn * 128 + l31) * RB;

  bf16x8 af[2][NP][4], wf[2][NP][4];                 // register double buffer of the fragments of one K step

  // fragment read number r of a K step, in the order the MFMAs consume them (a_lo, w_hi first in exact mode)
  auto read_one = [&](const unsigned char* sb, const int kc, const int r, bf16x8 (&a)[NP][4], bf16x8 (&w)[NP][4]) {
    const int coff = ((2 * kc + hi) ^ fswz) * 16;
    const int grp = r >> 2, i = r & 3;                // exact: groups a_lo, w_hi, w_lo, a_hi ; fast: a, w
    const bool is_a = (NP == 2) ? (grp == 0 || grp == 3) : (grp == 0);
    const int p = (NP == 2) ? ((grp == 0 || grp == 2) ? 1 : 0) : 0;
    if (is_a) a[p][i] = *reinterpret_cast<const bf16x8*>(sb + p * PLANE + a_row_off + i * 32 * RB + coff);
    else w[p][i] = *reinterpret_cast<const bf16x8*>(sb + p * PLANE + w_row_off + i * 32 * RB + coff);
  };
  auto read_frags = [&](const unsigned char* sb, const int kc, bf16x8 (&a)[NP][4], bf16x8 (&w)[NP][4]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) read_one(sb, kc, r, a, w);
  };

  // One K tile.  The barrier sits PA MFMAs into the last K step; after it come the first fragment reads of the next
  // stage.  DMA: c1 = context of tile kt+1 (its instructions >= (TM-B0)/SP are still to be issued, `cont`),
  // c2 = context of tile kt+2 (`more2`).  On the final tile the fragment reads fetch stale data that nobody uses.
  auto tile = [&](const int kt, const bool cont_in, bool more2, long& c1_off, int& c1_sbase, int& c1_shift, int& c1_half) {
#ifdef G3_ABL                                     // compile-time ablations (tools/ablate_gemm3.sh): 1 no DMA, 2 no MFMA,
    constexpr bool do_sync = !(G3_ABL & 64);        // 8 no fragment reads, 64 no barrier
    const bool cont = (G3_ABL & 1) ? false : cont_in;
    if (G3_ABL & 1) more2 = false;
#define RD(...) do { if (!(G3_ABL & 8)) { __VA_ARGS__; } } while (0)
#define MF(...) do { if (!(G3_ABL & 2)) { __VA_ARGS__; } } while (0)
#else
    constexpr bool do_sync = true;
    const bool cont = cont_in;
#define RD(...) __VA_ARGS__
#define MF(...) __VA_ARGS__
#endif
    const unsigned char* sb = smem + (kt & 1) * STAGE;
    const unsigned char* sn = smem + ((kt + 1) & 1) * STAGE;
    DmaCtx c1, c2;
    c1.off = c1_off; c1.sbase = c1_sbase; c1.shift = c1_shift; c1.half = c1_half;
    c2 = c1;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int m = kc * NM + i;
        if (m == B0) {
          if (do_sync) {
            // this wave's share of tile kt+1 has landed and its reads of tile kt are complete ...
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // ... and so have everybody else's: stage kt&1 is free
          }
          c2 = dma_ctx(kt + 2, kt & 1);
        }
        MF(mfma_step<NSPLIT, NP>(i, af[kc & 1], wf[kc & 1], acc));
        if (kc + 1 < KCH) {
          if ((i % MPR) == MPR - 1 && i / MPR < NR) RD(read_one(sb, kc + 1, i / MPR, af[(kc + 1) & 1], wf[(kc + 1) & 1]));
        } else if (i >= PA && i - PA < NR / RPM) {
#pragma unroll
          for (int q = 0; q < RPM; ++q) RD(read_one(sn, 0, (i - PA) * RPM + q, af[0], wf[0]));
        }
        if (m >= B0) {
          if ((m - B0) % SP == 0 && (m - B0) / SP < NDMA) {
            if (more2) dma_one(c2, (m - B0) / SP);
          }
        } else if ((m + TM - B0) % SP == 0 && (m + TM - B0) / SP < NDMA) {
          if (cont) dma_one(c1, (m + TM - B0) / SP);
        }
      }
    }
    c1_off = c2.off; c1_sbase = c2.sbase; c1_shift = c2.shift; c1_half = c2.half;
  };

  auto run_k = [&](const int kt0, const int kt1) {
    if (kt1 <= kt0) return;
    __syncthreads();                                  // nobody still reads the ring (previous K phase)
    issue_tile(kt0, kt0 & 1);
    if (kt0 + 1 < kt1) {
      issue_tile(kt0 + 1, (kt0 + 1) & 1);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile kt0 landed (tile kt0+1 may still be in flight)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    DmaCtx c1 = dma_ctx(kt0 + 1, (kt0 + 1) & 1);
    if (wave_active) {
      read_frags(smem + (kt0 & 1) * STAGE, 0, af[0], wf[0]);
      long c_off = c1.off;
      int c_sbase = c1.sbase, c_shift = c1.shift, c_half = c1.half;
      for (int kt = kt0; kt < kt1; ++kt) tile(kt, kt > kt0 && kt + 1 < kt1, kt + 2 < kt1, c_off, c_sbase, c_shift, c_half);
    } else {
      for (int kt = kt0; kt < kt1; ++kt) {
        if (kt > kt0 && kt + 1 < kt1) {
#pragma unroll
          for (int i = (TM - B0 + SP - 1) / SP; i < NDMA; ++i) dma_one(c1, i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        c1 = dma_ctx(kt + 2, kt & 1);
        if (kt + 2 < kt1) {
#pragma unroll
          for (int i = 0; i < (TM - B0 + SP - 1) / SP; ++i) dma_one(c1, i);
        }
      }
    }
  };

  if constexpr (EPI == EPI_WAVENET) {
    run_k(0, mid_tile);
    if (wave_active) wavenet_midgate<4, 4>(acc, g, z, row_base, col_base, l31, hi);
    run_k(mid_tile, ntiles);
  } else {
    run_k(0, ntiles);
  }
  __syncthreads();                                   // the LDS ring is free: every wave takes a private 18 KiB region
  if (!wave_active) return;
  if constexpr (EPI == EPI_F32) {
    unsigned char* wbuf = smem + wave * EPI_LDS_WAVE_BYTES;
    gemm_epilogue_lds<EPI, 4, 0>(acc, g, z, row_base, col_base, 0, lane, wbuf);
    if (col_base + 64 < g.N) gemm_epilogue_lds<EPI, 4, 2>(acc, g, z, row_base, col_base + 64, 0, lane, wbuf);
    return;
  }
  gemm_epilogue<EPI, 4, 4>(acc, g, z, row_base, col_base, tn * 128 + wn * 64, lane);
}

const bf16_t* gemm_zero_page();      // gemm2.hip

template <int NSPLIT, int EPI>
static hipError_t launch3_one(const GemmArgs& g, hipStream_t s) {
  const int ntn = (g.N + G3_BN - 1) / G3_BN, ntm = (g.M + G3_BM - 1) / G3_BM;
  const int nz = g.nz > 0 ? g.nz : 1;
  const size_t lds = 128 * 1024;                      // 2 x 64 KiB K stages (the 4 x 18 KiB epilogue regions reuse them)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<NSPLIT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const bf16_t* zp = gemm_zero_page();
  if (!zp) return hipErrorOutOfMemory;
  hipLaunchKernelGGL((gemm3_kernel<NSPLIT, EPI>), dim3(ntn * ntm * nz), dim3(256), lds, s, g, zp);
  return hipGetLastError();
}

template <int NSPLIT>
static hipError_t launch3_epi(const GemmArgs& g, hipStream_t s) {
  switch (g.epi) {
    case EPI_F32: return launch3_one<NSPLIT, EPI_F32>(g, s);
    case EPI_SPLIT: return launch3_one<NSPLIT, EPI_SPLIT>(g, s);
    case EPI_QKV: return launch3_one<NSPLIT, EPI_QKV>(g, s);
    case EPI_GEGLU: return launch3_one<NSPLIT, EPI_GEGLU>(g, s);
    case EPI_WAVENET: return launch3_one<NSPLIT, EPI_WAVENET>(g, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm3(const GemmArgs& g, int nsplit, hipStream_t s) {
  return nsplit == 3 ? launch3_epi<3>(g, s) : launch3_epi<1>(g, s);
}

}  // namespace ns2
