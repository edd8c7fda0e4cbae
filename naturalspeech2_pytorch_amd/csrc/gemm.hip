// GEMM / causal-conv family for the NaturalSpeech2 denoiser on gfx950 (CDNA4).
//
//   C[M, N] = epilogue( sum_k A[m - shift(k), k] * W[n, k] )
//
// One kernel covers every dense contraction of Model.forward:
//   * nn.Linear                         (NS2:1051-1053, 1069, 1021, 1024, 783)        shift = 0
//   * CausalConv1d k=3, dilation 2^i    (NS2:583-595, 615, 1016)  as a shifted-row implicit GEMM:
//       K = [tap0 | tap1 | tap2] x Cin, tap t reads activation row n - (2 - t) * dilation of the SAME
//       utterance and zero-fills across the utterance start (the reference left-pads with zeros).
//   * WavenetResBlock (NS2:597-642) fused into ONE launch: K-phase 1 = dilated conv, then the
//       accumulator is transformed in registers  h = tanh(g) * sigmoid(g), g = (acc + b) * gamma_t + beta_t,
//       then K-phase 2 accumulates the 1x1 res_conv of the same input on top (out = h + res).
//   * GEGLU (NS2:1004-1007) in the FF-in epilogue (weight rows are packed so that a wave owns a
//       32-column "x" tile and the matching 32-column "gate" tile).
//
// Operands are bf16 split planes (ns2_common.h).  NSPLIT = 3: hi*hi + hi*lo + lo*hi ("exact" mode, fp32-class
// accuracy on the bf16 MFMA pipe); NSPLIT = 1: hi*hi only ("fast" mode).
//
// Tiling (MI355X-first, 64-wide waves): 128x128x32 block tile, 256 threads = 4 waves in 2x2, each wave a
// 64x64 tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs).  Global -> registers -> LDS
// staging with the next K-tile's loads issued before the current tile's MFMAs (guide T14), LDS
// double-buffered (one barrier per K-tile), rows padded to 80 B so that both the 16-B ds_write of the
// staging pass and the ds_read_b128 fragment reads are bank-conflict free (80 = 5 x 16 B, 5 coprime to the
// 16 slots of a 256-B bank row).  Tile ids are remapped so that consecutive ids share an XCD L2 (T1).
#include "gemm_epi_fast.h"

namespace ns2 {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                      // padded LDS row stride in bytes (64 B payload)
constexpr int PLANE = BM * ROWB;              // 10240 B

template <int NSPLIT>
struct Stage {                                // registers holding one prefetched K-tile slice per thread
  uint4 a[NSPLIT == 1 ? 1 : 2][2];            // plane 1: bf16 lo (NSPLIT 3) or the [h8 x 32 | l8 x 32] byte half (NSPLIT 2, FMT_H8)
  uint4 w[NSPLIT == 1 ? 1 : 2][2];
};

NS2_DEVINL uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
NS2_DEVINL uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

// Interior 64 x 64 wave tile (every row and column valid): the streamlined epilogues of gemm_epi_fast.h through `wbuf`, the wave's
// private LDS region of WBUF bytes.  Returns false when the tile, the alignment or the format asks for the generic path.
// HALF / BF: which plane formats this caller can be asked for (a kernel on IEEE-half operands writes F16 / H8, one on bf16 operands
// bf16 planes; the split-K finishing kernel serves both).
template <int EPI, int WBUF, bool HALF, bool BF, bool BF_DENSE>
NS2_DEVINL bool small_tile_fast_epilogue(f32x16 (&acc)[2][2], const GemmArgs& g, int z, int row_base, int col_base, int ocol_base, int lane,
                                         unsigned char* wbuf) {
  if constexpr (EPI == EPI_F32) return false;
  if (row_base + 64 > g.M) return false;
  const bool al = (reinterpret_cast<uintptr_t>(g.out_hi) & 15) == 0 && (g.ldo_s & 31) == 0;
  auto planes = [&](auto&& fn) __attribute__((always_inline)) {
    if (!al) return false;
    if constexpr (HALF) {
      if (g.out_fmt == FMT_F16 && !g.out_lo) { fn(std::integral_constant<int, PF_F16>{}); return true; }
      if (g.out_fmt == FMT_H8) { fn(std::integral_constant<int, PF_H8>{}); return true; }
    }
    if constexpr (BF) {
      if (g.out_fmt == FMT_BF16 && g.out_lo) { fn(std::integral_constant<int, PF_BF16IL>{}); return true; }
    }
    if constexpr (BF_DENSE) {
      if (g.out_fmt == FMT_BF16 && !g.out_lo) { fn(std::integral_constant<int, PF_BF16>{}); return true; }
    }
    return false;
  };
  if constexpr (EPI == EPI_SPLIT) {
    if (col_base + 64 <= g.N && g.act == 0)
      return planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, true, 2, WBUF>(acc, g, z, row_base, col_base, lane, wbuf); });
  } else if constexpr (EPI == EPI_WAVENET) {
    if (col_base + 64 <= g.N)
      return planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, false, 2, WBUF>(acc, g, z, row_base, col_base, lane, wbuf); });
  } else if constexpr (EPI == EPI_GEGLU) {
    static_assert(EPI != EPI_GEGLU || WBUF >= 9216, "64 staged rows of a 128-byte output line + pad");
    if (ocol_base + 32 <= g.out_ncols)
      return planes([&](auto pf) __attribute__((always_inline)) { epi_geglu_fast<decltype(pf)::value, 2>(acc, g, row_base, col_base, ocol_base, lane, wbuf); });
  } else if constexpr (EPI == EPI_QKV) {
    static_assert(EPI != EPI_QKV || WBUF >= 9216, "64 feature rows of 64 tokens + pad");
    if (col_base + 64 <= g.N && !g.bias) {
      if (col_base + 64 <= g.split_col)
        return planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, false, 2, WBUF>(acc, g, 0, row_base, col_base, lane, wbuf); });
      if (col_base >= g.split_col && !g.vt_lo && g.seq_len > 0 && (g.seq_len & 63) == 0 && (g.vt_ld & 7) == 0 &&
          (reinterpret_cast<uintptr_t>(g.vt_hi) & 15) == 0) {
        if (HALF && g.vt_fmt == FMT_F16) { epi_vt_fast<true, 2>(acc, g, row_base, col_base, lane, wbuf); return true; }
        if ((BF || BF_DENSE) && g.vt_fmt == FMT_BF16) { epi_vt_fast<false, 2>(acc, g, row_base, col_base, lane, wbuf); return true; }
      }
    }
  }
  return false;
}

template <int NSPLIT, int EPI, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
  static_assert(NSPLIT != 2 || F16, "the mixed mode multiplies IEEE-half operands");
  constexpr int NP = (NSPLIT == 1) ? 1 : 2;   // 64-B halves of the interleaved line staged per operand
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE_BYTES = 2 * NP * PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int ntn = (g.N + BN - 1) / BN;
  const int ntm = (g.M + BM - 1) / BM;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % ntn;
  bid /= ntn;
  const int tm = bid % ntm;
  const int z = bid / ntm;

  // interleaved [hi32|lo32] rows when the operand has a lo plane (ns2_common.h); a_zs = logical column offset per z
  const bool ail = g.a_lo != nullptr, wil = g.w_lo != nullptr;
  const long azo = pcol((int)(z * g.a_zs), ail), wzo = ((long)z * g.w_zs) << (wil ? 1 : 0);
  const bf16_t* a_pl[2] = {g.a_hi + azo, g.a_lo + azo};
  const bf16_t* w_pl[2] = {g.w_hi + wzo, g.w_lo + wzo};
  const int dil = g.dil_z ? (g.dil << z) : g.dil;
  // split-K slice (GemmArgs::ksplit; operands are not offset per z then): K tiles [kin0, kin0 + kpt) of every tap
  const int kin0 = g.ksplit > 0 ? z * g.ksplit : 0;
  const int kpt = g.ksplit > 0 ? min(g.ksplit, g.kt_per_tap - kin0) : g.kt_per_tap;
  const int nkt = g.ksplit > 0 ? (g.nkt / g.kt_per_tap) * kpt : g.nkt;

  // ---- staging coordinates (2 chunks of 16 B per plane per thread)
  int srow[2], skc[2], nseq[2];
  bool arow_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = tid + 256 * i;
    srow[i] = c >> 2;
    skc[i] = c & 3;
    int m = tm * BM + srow[i];
    arow_ok[i] = m < g.M;
    nseq[i] = (g.seq_len > 0) ? (m % g.seq_len) : 0x3fffffff;   // 0x3fffffff: no sequence structure, never shifted
  }

  auto load_stage = [&](Stage<NSPLIT>& st, int kt, auto npl) {      // npl: planes staged (1 = the half plane only, p1_half)
    // same K order as gemm2.hip: shifted conv taps tap-minor (L2 reuse of the shifted A rows), then the unshifted taps
    int tap, kin;
    if (kt < g.conv_taps * kpt) { kin = kt / g.conv_taps; tap = kt - kin * g.conv_taps; }
    else { tap = kt / kpt; kin = kt - tap * kpt; }
    kin += kin0;
    const int kcol = kin * BK;
    const int pl = g.pad_left < 0 ? g.conv_taps - 1 : g.pad_left;        // causal: all padding on the left (NS2:583-595)
    const int shift = (tap < g.conv_taps) ? (pl - tap) * dil : 0;
    const unsigned slim = g.seq_len > 0 ? (unsigned)g.seq_len : 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = (long)tm * BM + srow[i];
      const bool ok = arow_ok[i] && ((unsigned)(nseq[i] - shift) < slim);   // source row inside the same utterance
      const long aoff = (m - shift) * pld(g.lda, ail) + pcol(kcol + skc[i] * 8, ail);
      const long woff = ((long)tn * BN + srow[i]) * pld(g.ldw, wil) + pcol((tap * g.kt_per_tap + kin) * BK + skc[i] * 8, wil);
#pragma unroll
      for (int p = 0; p < decltype(npl)::value; ++p) {
        st.a[p][i] = ok ? ld16(a_pl[p] + aoff) : zero16();
        st.w[p][i] = ld16(w_pl[p] + woff);
      }
    }
  };
  auto store_stage = [&](const Stage<NSPLIT>& st, int s, auto npl) {
    unsigned char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = srow[i] * ROWB + skc[i] * 16;
#pragma unroll
      for (int p = 0; p < decltype(npl)::value; ++p) {
        *reinterpret_cast<uint4*>(base + p * PLANE + off) = st.a[p][i];
        *reinterpret_cast<uint4*>(base + (NP + p) * PLANE + off) = st.w[p][i];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // epilogue coordinates of this lane
  const int row_base = tm * BM + wm * 64;     // + mi*32 + (r&3) + 8*(r>>2) + 4*hi
  const int col_base = tn * BN + wn * 64;     // + ni*32 + l31

  // 32-column sub-tiles of this wave that hold real output columns (wave-uniform): the SEANet codec's 1 ... 32-channel
  // convolutions over 10 M rows would otherwise spend 4 x their MFMA time on columns nobody stores
#ifndef NS2_GEMM_NO_NSKIP                        // A/B switch of experiment builds
  const int nv = g.N - col_base > 32 ? 2 : (g.N - col_base > 0 ? 1 : 0);
#else
  constexpr int nv = 2;
#endif

  const int a_frag_off = (wm * 64 + l31) * ROWB + hi * 16;
  const int w_frag_off = (wn * 64 + l31) * ROWB + hi * 16;

  // software-pipelined K loop over tiles [kt0, kt1): stage kt+1 is in flight (global -> VGPR) while tile kt
  // is multiplied out of LDS; one barrier per tile.
  // half_only (mixed mode, EPI_WAVENET with g.p1_half: the dilated-conv taps of the hybrid plan): ONE IEEE-half product, the
  // byte half of the operand lines is neither loaded nor staged -- the same arithmetic as gemm2.hip's p1_half phase
  auto run_k = [&](const int kt0, const int kt1, auto half_only) {
    constexpr bool HALF1 = decltype(half_only)::value;
    using NPL = std::integral_constant<int, HALF1 ? 1 : NP>;
    Stage<NSPLIT> st;
    load_stage(st, kt0, NPL{});
    store_stage(st, kt0 & 1, NPL{});
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = (kt + 1) < kt1;
      if (more) load_stage(st, kt + 1, NPL{});
      const unsigned char* sb = smem + (kt & 1) * STAGE_BYTES;
      if constexpr (NSPLIT == 2 && HALF1) {
        bf16x8 af[2][2], wf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            af[kc][i] = *reinterpret_cast<const bf16x8*>(sb + a_frag_off + i * 32 * ROWB + kc * 32);
            wf[kc][i] = *reinterpret_cast<const bf16x8*>(sb + 2 * PLANE + w_frag_off + i * 32 * ROWB + kc * 32);
          }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (ni >= nv) continue;
            acc[mi][ni] = mma16<true>(af[0][mi], wf[0][ni], acc[mi][ni]);
            acc[mi][ni] = mma16<true>(af[1][mi], wf[1][ni], acc[mi][ni]);
          }
      } else if constexpr (NSPLIT == 2) {
        // mixed mode (see gemm2.hip): 2 x (2x2) half MFMAs + (2x2) fp8 MFMAs of K = 64 per 32-deep tile; plane 1 of the
        // LDS image is the byte half of the line, [h8 x 32 | l8 x 32]
        bf16x8 af[2][2], wf[2][2];
        i32x8 a8[2], w8[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            af[kc][i] = *reinterpret_cast<const bf16x8*>(sb + a_frag_off + i * 32 * ROWB + kc * 32);
            wf[kc][i] = *reinterpret_cast<const bf16x8*>(sb + 2 * PLANE + w_frag_off + i * 32 * ROWB + kc * 32);
          }
          const unsigned char* pa = sb + PLANE + (wm * 64 + i * 32 + l31) * ROWB + 32 * hi;
          const unsigned char* pw = sb + 3 * PLANE + (wn * 64 + i * 32 + l31) * ROWB + 32 * (1 - hi);
          const int4 a0 = *reinterpret_cast<const int4*>(pa), a1 = *reinterpret_cast<const int4*>(pa + 16);
          const int4 w0 = *reinterpret_cast<const int4*>(pw), w1 = *reinterpret_cast<const int4*>(pw + 16);
          a8[i] = i32x8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          w8[i] = i32x8{w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (ni >= nv) continue;
            acc[mi][ni] = mma16<true>(af[0][mi], wf[0][ni], acc[mi][ni]);
            acc[mi][ni] = mma16<true>(af[1][mi], wf[1][ni], acc[mi][ni]);
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[mi], w8[ni], acc[mi][ni], 1, 1, 0, H8_E8M0_LO, 0,
                                                                          H8_E8M0_ONE);
          }
      } else {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        bf16x8 af[NP][2], wf[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[p][i] = *reinterpret_cast<const bf16x8*>(sb + p * PLANE + a_frag_off + i * 32 * ROWB + kc * 32);
            wf[p][i] = *reinterpret_cast<const bf16x8*>(sb + (NP + p) * PLANE + w_frag_off + i * 32 * ROWB + kc * 32);
          }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (ni >= nv) continue;
            if constexpr (NSPLIT == 3) {
              acc[mi][ni] = mma16<F16>(af[1][mi], wf[0][ni], acc[mi][ni]);
              acc[mi][ni] = mma16<F16>(af[0][mi], wf[1][ni], acc[mi][ni]);
            }
            acc[mi][ni] = mma16<F16>(af[0][mi], wf[0][ni], acc[mi][ni]);
          }
      }
      }
      if (more) store_stage(st, (kt + 1) & 1, NPL{});
      __syncthreads();
    }
  };

  if constexpr (EPI == EPI_WAVENET) {
    if constexpr (NSPLIT == 2) {
      if (g.p1_half) run_k(0, g.mid_kt, std::true_type{});
      else run_k(0, g.mid_kt, std::false_type{});
    } else {
      run_k(0, g.mid_kt, std::false_type{});
    }
    // row_base % 64 == 0: with seq_len % 64 == 0 the wave tile lies inside one utterance (gamma / beta once per column)
    wavenet_midgate<2, 2>(acc, g, z, row_base, col_base, l31, hi, g.seq_len > 0 && (g.seq_len & 63) == 0);
    run_k(g.mid_kt, g.nkt, std::false_type{});
  } else {
    run_k(0, nkt, std::false_type{});
  }

  // Round 3: interior wave tiles of the fp32 epilogue (the SEANet codec's 16 ... 128-channel convolutions run here, 80 k blocks
  // per launch on the early layers) leave through the wave's share of the now idle LDS ring as 16-byte stores of whole row
  // segments, with the bounds tested once per wave -- the generic epilogue below tests and branches per stored value.
  if constexpr (EPI == EPI_F32) {
    const int nvc = min(64, g.N - col_base);                     // valid columns of this wave tile
    if (nvc <= 0) return;
    if (row_base + 64 <= g.M && nvc >= 16 && (nvc & (nvc - 1)) == 0 && g.act == 0 && (g.ldo_f & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(g.out_f) & 15) == 0 &&
        (!g.resid || ((g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.resid) & 15) == 0))) {
      // two passes of 32 rows: 32 x (64 fp32 + 16 B pad) = 8.5 KiB fits the wave's share of the ring in every arithmetic mode
      // (10 KiB with one operand plane staged, 20 KiB with two; round 4: the single-plane kernels took the per-value path before)
      constexpr int RS = 272;
      unsigned char* wbuf = smem + wave * (STAGE_BYTES / 2);
      const float bc0 = (g.bias && col_base + l31 < g.N) ? g.bias[col_base + l31] : 0.f;
      const float bc1 = (g.bias && col_base + 32 + l31 < g.N) ? g.bias[col_base + 32 + l31] : 0.f;
      const int lcpr = 31 - __builtin_clz(nvc >> 2);              // log2(16-byte chunks per row): 2, 3 or 4
      const int lr0 = lane >> lcpr, ch = lane & ((1 << lcpr) - 1), rpi = 64 >> lcpr;
      // Round 5 -- the RMSNorm behind a residual update, in the same epilogue (launch_gemm guarantees N == 128 == BN and M % 128 == 0,
      // so this block owns 128 whole rows and all four waves are here): the lane keeps its 16 float4 of x, the row's sum of squares
      // is 16 lanes of this wave + the partner wave's half (wn ^ 1) through 64 floats of LDS per wave, added in a fixed order.
      if (g.nrm_hi) {
        float4 keep[2][8];
        float* s_part = reinterpret_cast<float*>(smem + wave * (STAGE_BYTES / 2) + 32 * RS);      // 64 floats in the slack behind the staged rows
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
              *reinterpret_cast<float*>(wbuf + lr * RS + (ni * 32 + l31) * 4) = acc[mi][ni][r] + (ni ? bc1 : bc0);
            }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 8; ++it) {                          // 64 columns: 16 lanes per row, 4 rows per iteration
            const int lr = it * 4 + (lane >> 4);
            float4 v = *reinterpret_cast<const float4*>(wbuf + lr * RS + (lane & 15) * 16);
            const long row = row_base + mi * 32 + lr;
            if (g.resid) {
              const float4 rr = *reinterpret_cast<const float4*>(g.resid + row * g.ldr + col_base + (lane & 15) * 4);
              v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            *reinterpret_cast<float4*>(g.out_f + row * g.ldo_f + col_base + (lane & 15) * 4) = v;
            keep[mi][it] = v;
            float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
            if ((lane & 15) == 0) s_part[mi * 32 + lr] = ss;
          }
          __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        const float* p0 = reinterpret_cast<const float*>(smem + (wm * 2 + 0) * (STAGE_BYTES / 2) + 32 * RS);
        const float* p1 = reinterpret_cast<const float*>(smem + (wm * 2 + 1) * (STAGE_BYTES / 2) + 32 * RS);
        const float scale = sqrtf((float)g.N);
        const bool nil = g.nrm_lo != nullptr;
        const int c = col_base + (lane & 15) * 4;
        float gm[4] = {1.f, 1.f, 1.f, 1.f};
        if (g.nrm_gamma) { const float4 t = *reinterpret_cast<const float4*>(g.nrm_gamma + c); gm[0] = t.x; gm[1] = t.y; gm[2] = t.z; gm[3] = t.w; }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + (lane >> 4);
            const long row = row_base + mi * 32 + lr;
            const float tot = p0[mi * 32 + lr] + p1[mi * 32 + lr];
            const float inv = scale / fmaxf(sqrtf(tot), 1e-12f);   // F.normalize eps (NS2:727-746)
            float o[4] = {keep[mi][it].x * inv, keep[mi][it].y * inv, keep[mi][it].z * inv, keep[mi][it].w * inv};
            if (g.nrm_gamma) { o[0] *= gm[0]; o[1] *= gm[1]; o[2] *= gm[2]; o[3] *= gm[3]; }
            if (g.nrm_cond) {
              const float* gc = g.nrm_cond + (g.nrm_seq_len > 0 ? row / g.nrm_seq_len : 0) * (long)g.nrm_cond_ld;
              const float4 t = *reinterpret_cast<const float4*>(gc + c), u = *reinterpret_cast<const float4*>(gc + g.N + c);
              o[0] = o[0] * t.x + u.x; o[1] = o[1] * t.y + u.y; o[2] = o[2] * t.z + u.z; o[3] = o[3] * t.w + u.w;
            }
            store_cols4(g.nrm_hi + row * pld(g.nrm_ld, nil), c, o[0], o[1], o[2], o[3], g.nrm_fmt, nil);
          }
        return;
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
            *reinterpret_cast<float*>(wbuf + lr * RS + (ni * 32 + l31) * 4) = acc[mi][ni][r] + (ni ? bc1 : bc0);
          }
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < (32 >> (6 - lcpr)); ++it) {         // 32 rows / rpi
          const int lr = it * rpi + lr0;
          float4 v = *reinterpret_cast<const float4*>(wbuf + lr * RS + ch * 16);
          const long row = row_base + mi * 32 + lr;
          if (g.resid) {
            const float4 rr = *reinterpret_cast<const float4*>(g.resid + row * g.ldr + col_base + ch * 4);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          *reinterpret_cast<float4*>(g.out_f + z * g.out_f_zs + row * g.ldo_f + col_base + ch * 4) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
  // Round 4: interior wave tiles of the plane-writing epilogues take gemm_epi_fast.h's route too (the dim = 128 model's Wavenet
  // blocks run here: 2048 blocks per launch whose generic epilogue -- a bounds test, a run-time format switch and a 2 ... 4 byte
  // store per value -- cost more than their 16 K tiles; small batches put QKV / GEGLU here as well)
  if constexpr (EPI != EPI_F32) {
    constexpr int WBUF = STAGE_BYTES / 2;
    if (small_tile_fast_epilogue<EPI, WBUF, F16, !F16, (!F16 && NSPLIT == 1)>(acc, g, z, row_base, col_base, tn * 64 + wn * 32, lane, smem + wave * WBUF))
      return;
  }
  gemm_epilogue<EPI, 2, 2>(acc, g, z, row_base, col_base, tn * 64 + wn * 32, lane);
}

template <int NSPLIT, int EPI, bool F16>
static hipError_t launch_one(const GemmArgs& g, hipStream_t s) {
  const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
  const int nz = g.nz > 0 ? g.nz : 1;
  const size_t lds = 2 * 2 * (NSPLIT == 1 ? 1 : 2) * PLANE;
  static DynLdsAttr attr;
  {
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&gemm_kernel<NSPLIT, EPI, F16>), (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((gemm_kernel<NSPLIT, EPI, F16>), dim3(ntn * ntm * nz), dim3(256), lds, s, g);
  return hipGetLastError();
}

template <int NSPLIT, bool F16>
static hipError_t launch_epi(const GemmArgs& g, hipStream_t s) {
  switch (g.epi) {
    case EPI_F32: return launch_one<NSPLIT, EPI_F32, F16>(g, s);
    case EPI_SPLIT: return launch_one<NSPLIT, EPI_SPLIT, F16>(g, s);
    case EPI_QKV: return launch_one<NSPLIT, EPI_QKV, F16>(g, s);
    case EPI_GEGLU: return launch_one<NSPLIT, EPI_GEGLU, F16>(g, s);
    case EPI_WAVENET: return launch_one<NSPLIT, EPI_WAVENET, F16>(g, s);
  }
  return hipErrorInvalidValue;
}

// ---- split-K, second launch: the slots of the first launch (raw fp32 partial sums, [S][M][ldp]) are added in slot order in
// the accumulator layout of a 64 x 64 wave tile, then the REQUESTED epilogue runs on the sums -- the same device code the
// one-launch product would have run (gemm_epi.h), so every format, bias, residual, V^T and range-guard rule holds unchanged.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const GemmArgs g, const float* part, int S, long slot, int ldp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int ntn = (g.N + BN - 1) / BN;
  const int tn = blockIdx.x % ntn, tm = blockIdx.x / ntn;
  const int row_base = tm * BM + wm * 64, col_base = tn * BN + wn * 64;
  // Unconditional loads from clamped addresses, all 64 of a slot in flight together (a bounds branch per value serialised them:
  // 64 x S dependent round trips, 85-200 us per launch; one (mi, ni) quarter at a time still left 4 x S dependent batches);
  // invalid positions are zeroed afterwards.  Slot order per value is fixed: s = 0, 1, ...
  f32x16 acc[2][2];
  int off[2][2][16];                      // < 2^31: a slot is at most SPLITK_SCRATCH_FLOATS / 2 elements
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = min(col_base + ni * 32 + l31, g.N - 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, g.M - 1);
        off[mi][ni][r] = row * ldp + col;
        acc[mi][ni][r] = 0.f;
      }
    }
  for (int sl = 0; sl < S; ++sl) {
    const float* ps = part + sl * slot;
    f32x16 l[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) l[mi][ni][r] = ps[off[mi][ni][r]];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += l[mi][ni][r];
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const bool cok = col_base + ni * 32 + l31 < g.N;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!(cok && row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < g.M)) acc[mi][ni][r] = 0.f;
    }
  if constexpr (EPI != EPI_F32) {
    // interior wave tiles: the LDS-staged epilogues (gemm_epi_fast.h), as in the one-launch kernel
    __shared__ __attribute__((aligned(16))) unsigned char fin_lds[4 * 9216];
    if (small_tile_fast_epilogue<EPI, 9216, true, true, true>(acc, g, 0, row_base, col_base, tn * 64 + wn * 32, lane, fin_lds + wave * 9216)) return;
  }
  gemm_epilogue<EPI, 2, 2>(acc, g, 0, row_base, col_base, tn * 64 + wn * 32, lane);
}

// EPI_F32 needs no accumulator layout: one thread per 4 adjacent columns, 16-byte loads of every slot / the residual, 16-byte
// store.  The same operation order per value as gemm_epilogue<EPI_F32>: (sum + bias) -> activation -> + residual.
__global__ __launch_bounds__(256) void splitk_finish_f32_kernel(const float* part, int S, long slot, int ldp, int M, int N, const float* bias,
                                                                const float* resid, int ldr, int act, float* out, int ldo) {
  const int n4 = N >> 2;                                       // N % 4 == 0 (checked by the launcher)
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * n4) return;
  const int row = (int)(idx / n4), c = (int)(idx - (long)row * n4) * 4;
  const float* p = part + (long)row * ldp + c;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sl = 0; sl < S; ++sl) {
    const float4 t = *reinterpret_cast<const float4*>(p + sl * slot);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (bias) { v.x += bias[c]; v.y += bias[c + 1]; v.z += bias[c + 2]; v.w += bias[c + 3]; }
  if (act) { v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act); }
  if (resid) {
    const float4 r = *reinterpret_cast<const float4*>(resid + (long)row * ldr + c);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  *reinterpret_cast<float4*>(out + (long)row * ldo + c) = v;
}

// The same finishing pass with the RMSNorm that follows the residual update (GemmArgs::nrm_*; round 5): N = 128, 256 or 512, so a row
// is 32 / 64 / 128 threads of a 256-thread block (N4 = N / 4 threads, 256 / N4 rows per block): the row's sum of squares is a wave
// reduction plus, for N = 512, one exchange between the two waves of the row through LDS, added in a fixed order.  Small batches (1 .. 8
// utterances: every N = dim product is split over K) lose their rmsnorm launches this way -- 13 of the ~58 launches of a dim-128 step.
template <int N4>
__global__ __launch_bounds__(256) void splitk_finish_f32_norm_kernel(const float* part, int S, long slot, int ldp, int M, const float* bias,
                                                                     const float* resid, int ldr, float* out, int ldo, GemmArgs g) {
  constexpr int RPB = 256 / N4;                                  // rows per block
  __shared__ float s_sum[4];
  const int tid = threadIdx.x;
  const int rloc = tid / N4, c = (tid - rloc * N4) * 4;
  const long row_raw = (long)blockIdx.x * RPB + rloc;
  const bool ok = row_raw < M;
  const long row = ok ? row_raw : M - 1;                         // every lane computes (the reductions need them); only valid rows store
  const float* p = part + row * ldp + c;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sl = 0; sl < S; ++sl) {
    const float4 t = *reinterpret_cast<const float4*>(p + sl * slot);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (bias) { v.x += bias[c]; v.y += bias[c + 1]; v.z += bias[c + 2]; v.w += bias[c + 3]; }
  if (resid) {
    const float4 r = *reinterpret_cast<const float4*>(resid + row * ldr + c);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (ok) *reinterpret_cast<float4*>(out + row * ldo + c) = v;
  float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
  for (int d = 1; d < (N4 < 64 ? N4 : 64); d <<= 1) ss += __shfl_xor(ss, d, 64);
  if constexpr (N4 > 64) {                                       // N = 512: waves 2 r, 2 r + 1 hold the two halves of row r
    if ((tid & 63) == 0) s_sum[tid >> 6] = ss;
    __syncthreads();
    ss = s_sum[2 * rloc] + s_sum[2 * rloc + 1];
  }
  const float inv = sqrtf((float)(4 * N4)) / fmaxf(sqrtf(ss), 1e-12f);
  float o[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
  if (g.nrm_gamma) { const float4 t = *reinterpret_cast<const float4*>(g.nrm_gamma + c); o[0] *= t.x; o[1] *= t.y; o[2] *= t.z; o[3] *= t.w; }
  if (g.nrm_cond) {
    const float* gc = g.nrm_cond + (g.nrm_seq_len > 0 ? row / g.nrm_seq_len : 0) * (long)g.nrm_cond_ld;
    const float4 t = *reinterpret_cast<const float4*>(gc + c), u = *reinterpret_cast<const float4*>(gc + 4 * N4 + c);
    o[0] = o[0] * t.x + u.x; o[1] = o[1] * t.y + u.y; o[2] = o[2] * t.z + u.z; o[3] = o[3] * t.w + u.w;
  }
  const bool nil = g.nrm_lo != nullptr;
  if (ok) store_cols4(g.nrm_hi + row * pld(g.nrm_ld, nil), c, o[0], o[1], o[2], o[3], g.nrm_fmt, nil);
}

hipError_t launch_gemm1(const GemmArgs& g, int precision, hipStream_t s);
// g: validated by launch_gemm, formats resolved; S slices of c K tiles per tap
hipError_t launch_gemm_splitk(const GemmArgs& g, int precision, int S, int c, hipStream_t s) {
  const int ldp = (g.N + 63) & ~63;
  const long slot = (long)g.M * ldp;
  if (!g.sk_ws || S < 2 || c < 1 || S * slot > g.sk_ws_floats || g.epi == EPI_WAVENET || g.nz > 1) return hipErrorInvalidValue;
  GemmArgs p = g;
  p.epi = EPI_F32; p.bias = nullptr; p.bias2 = nullptr; p.resid = nullptr; p.act = 0; p.film = nullptr;
  p.out_f = g.sk_ws; p.ldo_f = ldp; p.out_f_zs = slot;
  p.out_hi = nullptr; p.out_lo = nullptr; p.vt_hi = nullptr; p.vt_lo = nullptr; p.nrm_hi = nullptr; p.nrm_lo = nullptr;
  p.nz = S; p.a_zs = 0; p.w_zs = 0; p.dil_z = 0; p.ksplit = c;
  // the slots are ldp = round_up(N, 64) wide and the packed weight's padding rows are zeros: let the slices write whole 64-column
  // wave tiles (the vectorised fp32 epilogue; a ragged last tile took the per-value one: 41 us instead of ~25 for the FF conv's slices)
  p.N = ldp;
  hipError_t e = launch_gemm1(p, precision, s);
  if (e != hipSuccess) return e;
  const int grid = ((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM);
  switch (g.epi) {
    case EPI_F32:
      if ((g.N & 3) == 0 && (g.ldo_f & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out_f) & 15) == 0 &&
          (!g.resid || ((g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.resid) & 15) == 0))) {
        if (g.nrm_hi) {                                          // (gemm_fuses_norm: N in {128, 256, 512}, no activation)
          const int rpb = 256 / (g.N >> 2);
          const dim3 grid((unsigned)((g.M + rpb - 1) / rpb));
          if (g.N == 128) hipLaunchKernelGGL(splitk_finish_f32_norm_kernel<32>, grid, dim3(256), 0, s, g.sk_ws, S, slot, ldp, g.M, g.bias, g.resid, g.ldr, g.out_f, g.ldo_f, g);
          else if (g.N == 256) hipLaunchKernelGGL(splitk_finish_f32_norm_kernel<64>, grid, dim3(256), 0, s, g.sk_ws, S, slot, ldp, g.M, g.bias, g.resid, g.ldr, g.out_f, g.ldo_f, g);
          else hipLaunchKernelGGL(splitk_finish_f32_norm_kernel<128>, grid, dim3(256), 0, s, g.sk_ws, S, slot, ldp, g.M, g.bias, g.resid, g.ldr, g.out_f, g.ldo_f, g);
          break;
        }
        const long n = (long)g.M * (g.N >> 2);
        hipLaunchKernelGGL(splitk_finish_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g.sk_ws, S, slot, ldp, g.M, g.N,
                           g.bias, g.resid, g.ldr, g.act, g.out_f, g.ldo_f);
      } else {
        hipLaunchKernelGGL(splitk_finish_kernel<EPI_F32>, dim3(grid), dim3(256), 0, s, g, g.sk_ws, S, slot, ldp);
      }
      break;
    case EPI_SPLIT: hipLaunchKernelGGL(splitk_finish_kernel<EPI_SPLIT>, dim3(grid), dim3(256), 0, s, g, g.sk_ws, S, slot, ldp); break;
    case EPI_QKV: hipLaunchKernelGGL(splitk_finish_kernel<EPI_QKV>, dim3(grid), dim3(256), 0, s, g, g.sk_ws, S, slot, ldp); break;
    case EPI_GEGLU: hipLaunchKernelGGL(splitk_finish_kernel<EPI_GEGLU>, dim3(grid), dim3(256), 0, s, g, g.sk_ws, S, slot, ldp); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gemm1(const GemmArgs& g, int precision, hipStream_t s) {      // formats already validated by launch_gemm
  if (g.M <= 0 || g.N <= 0 || g.nkt <= 0) return hipErrorInvalidValue;
  switch (precision) {
    case 3: return launch_epi<3, false>(g, s);
    case 4: return launch_epi<2, true>(g, s);
    case 2: return launch_epi<1, true>(g, s);
    default: return launch_epi<1, false>(g, s);
  }
}

NS2_DEFINE_SATURATION_READER(gemm)

}  // namespace ns2
