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
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                      // padded LDS row stride in bytes (64 B payload)
constexpr int PLANE = BM * ROWB;              // 10240 B

template <int NSPLIT>
struct Stage {                                // registers holding one prefetched K-tile slice per thread
  uint4 a[NSPLIT == 3 ? 2 : 1][2];
  uint4 w[NSPLIT == 3 ? 2 : 1][2];
};

NS2_DEVINL uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
NS2_DEVINL uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

template <int NSPLIT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
  constexpr int NP = (NSPLIT == 3) ? 2 : 1;   // planes per operand
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

  const bf16_t* a_pl[2] = {g.a_hi + (long)z * g.a_zs, g.a_lo + (long)z * g.a_zs};
  const bf16_t* w_pl[2] = {g.w_hi + (long)z * g.w_zs, g.w_lo + (long)z * g.w_zs};
  const int dil = g.dil_z ? (g.dil << z) : g.dil;

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
    nseq[i] = (g.seq_len > 0) ? (m % g.seq_len) : 0x3fffffff;
  }

  auto load_stage = [&](Stage<NSPLIT>& st, int kt) {
    const int tap = kt / g.kt_per_tap;
    const int kcol = (kt - tap * g.kt_per_tap) * BK;
    const int shift = (tap < g.conv_taps) ? (g.conv_taps - 1 - tap) * dil : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long m = (long)tm * BM + srow[i];
      const bool ok = arow_ok[i] && (nseq[i] >= shift);
      const long aoff = (m - shift) * (long)g.lda + kcol + skc[i] * 8;
      const long woff = ((long)tn * BN + srow[i]) * (long)g.ldw + (long)kt * BK + skc[i] * 8;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        st.a[p][i] = ok ? ld16(a_pl[p] + aoff) : zero16();
        st.w[p][i] = ld16(w_pl[p] + woff);
      }
    }
  };
  auto store_stage = [&](const Stage<NSPLIT>& st, int s) {
    unsigned char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = srow[i] * ROWB + skc[i] * 16;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
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

  const int a_frag_off = (wm * 64 + l31) * ROWB + hi * 16;
  const int w_frag_off = (wn * 64 + l31) * ROWB + hi * 16;

  // software-pipelined K loop over tiles [kt0, kt1): stage kt+1 is in flight (global -> VGPR) while tile kt
  // is multiplied out of LDS; one barrier per tile.
  auto run_k = [&](const int kt0, const int kt1) {
    Stage<NSPLIT> st;
    load_stage(st, kt0);
    store_stage(st, kt0 & 1);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = (kt + 1) < kt1;
      if (more) load_stage(st, kt + 1);
      const unsigned char* sb = smem + (kt & 1) * STAGE_BYTES;
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
            if constexpr (NSPLIT == 3) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mi], wf[0][ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], wf[1][ni], acc[mi][ni], 0, 0, 0);
            }
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], wf[0][ni], acc[mi][ni], 0, 0, 0);
          }
      }
      if (more) store_stage(st, (kt + 1) & 1);
      __syncthreads();
    }
  };

  if constexpr (EPI == EPI_WAVENET) {
    run_k(0, g.mid_kt);
    // WavenetResBlock NS2:629-636: h = conv(x)+b ; h = h*gamma_t+beta_t ; h = tanh(h)*sigmoid(h) ; then += res_conv(x)
    // tanh(h)*sigmoid(h) = sign(h) * (1-u) * (h<0 ? u : 1) / (1+u^2),  u = exp(-|h|)   (one exp, no overflow)
    {
      const float* film = g.film + (long)z * g.film_zs;
      const float* bias = g.bias + (long)z * g.bias_zs;
      const float* bias2 = g.bias2 + (long)z * g.bias_zs;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = col_base + ni * 32 + l31;
          const bool cok = col < g.N;
          const float bc = cok ? bias[col] : 0.f;
          const float b2 = cok ? bias2[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            float v = 0.f;
            if (cok && row < g.M) {
              const int b = row / g.seq_len;
              const float gam = film[(long)b * g.film_ld + col];
              const float bet = film[(long)b * g.film_ld + g.N + col];
              const float h = (acc[mi][ni][r] + bc) * gam + bet;
              const float u = expf(-fabsf(h));
              const float t = (1.f - u) * (h < 0.f ? u : 1.f) * __frcp_rn(1.f + u * u);
              v = copysignf(t, h) + b2;
            }
            acc[mi][ni][r] = v;
          }
        }
    }
    run_k(g.mid_kt, g.nkt);
  } else {
    run_k(0, g.nkt);
  }

  // ------------------------------------------------------------------ epilogues
  const bool odd = lane & 1;

  if constexpr (EPI == EPI_F32) {
    // out = acc + bias (+ residual)      (to_out / FF-out / final_conv / to_pred)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = col_base + ni * 32 + l31;
        if (col >= g.N) continue;
        const float bc = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row >= g.M) continue;
          float v = acc[mi][ni][r] + bc;
          if (g.resid) v += g.resid[(long)row * g.ldr + col];
          g.out_f[(long)row * g.ldo_f + col] = v;
        }
      }
  } else if constexpr (EPI == EPI_GEGLU) {
    // wave tile = [x(32 cols) | gate(32 cols)] ; out[:, tn*64 + wn*32 + l31] = gelu(gate) * x   (NS2:1006-1007)
    const int ocol = tn * 64 + wn * 32 + l31;
    const int cx = col_base + l31, cg = col_base + 32 + l31;
    const float bx = g.bias[cx], bg = g.bias[cg];      // packed (padded) bias: always in range
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        float v0, v1;
        {
          const float x0 = acc[mi][0][2 * rp] + bx, g0 = acc[mi][1][2 * rp] + bg;
          const float x1 = acc[mi][0][2 * rp + 1] + bx, g1 = acc[mi][1][2 * rp + 1] + bg;
          v0 = gelu_erf(g0) * x0;
          v1 = gelu_erf(g1) * x1;
        }
        // pair adjacent columns: even lane stores row 2rp, odd lane stores row 2rp+1 (two bf16 per 4-B store)
        const float send = odd ? v0 : v1;
        const float recv = __shfl_xor(send, 1, 64);
        const float c_lo = odd ? recv : v0, c_hi = odd ? v1 : recv;
        const int r = 2 * rp + (odd ? 1 : 0);
        const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int col = ocol & ~1;
        if (row < g.M && col < g.out_ncols) {
          bf16_t h0, l0, h1, l1;
          split_bf16(c_lo, h0, l0);
          split_bf16(c_hi, h1, l1);
          const long o = (long)row * g.ldo_s + col;
          *reinterpret_cast<uint32_t*>(g.out_hi + o) = pack2(h0, h1);
          if (g.out_lo) *reinterpret_cast<uint32_t*>(g.out_lo + o) = pack2(l0, l1);
        }
      }
    }
  } else {
    // EPI_SPLIT / EPI_QKV / EPI_WAVENET: split planes, optionally the tail columns transposed (V^T for attention)
    const float* bias = g.bias ? g.bias + (long)z * g.bias_zs : nullptr;
    bf16_t* out_hi = g.out_hi + (long)z * g.out_zs;
    bf16_t* out_lo = g.out_lo ? g.out_lo + (long)z * g.out_zs : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = col_base + ni * 32 + l31;
        float bc = 0.f;
        if constexpr (EPI != EPI_WAVENET) bc = (bias && col < g.N) ? bias[col] : 0.f;   // wavenet biases were applied mid-loop
        const bool transposed = (EPI == EPI_QKV) && (col_base + ni * 32 >= g.split_col);   // wave-uniform
        if (!transposed) {
#pragma unroll
          for (int rp = 0; rp < 8; ++rp) {
            const float v0 = acc[mi][ni][2 * rp] + bc, v1 = acc[mi][ni][2 * rp + 1] + bc;
            const float send = odd ? v0 : v1;
            const float recv = __shfl_xor(send, 1, 64);
            // columns (col&~1, col|1): even lane holds its own col then the neighbour's, odd lane the reverse
            float c_lo = odd ? recv : v0, c_hi = odd ? v1 : recv;
            const int r = 2 * rp + (odd ? 1 : 0);
            const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int c0 = col & ~1;
            if (row < g.M && c0 < g.out_ncols) {
              if (c0 >= g.N) c_lo = 0.f;             // zero the K-padding columns of the next GEMM's operand
              if (c0 + 1 >= g.N) c_hi = 0.f;
              bf16_t h0, l0, h1, l1;
              split_bf16(c_lo, h0, l0);
              split_bf16(c_hi, h1, l1);
              const long o = (long)row * g.ldo_s + c0;
              *reinterpret_cast<uint32_t*>(out_hi + o) = pack2(h0, h1);
              if (out_lo) *reinterpret_cast<uint32_t*>(out_lo + o) = pack2(l0, l1);
            }
          }
        } else {
          // V^T[b][feature][n]: this lane owns feature `col - split_col` and 4 consecutive tokens per register group
          const int feat = col - g.split_col;
          if (col < g.N) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int row0 = row_base + mi * 32 + 8 * gq + 4 * hi;
              if (row0 >= g.M) continue;
              const int b = row0 / g.seq_len, n0 = row0 - b * g.seq_len;
              bf16_t h[4], l[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) split_bf16(acc[mi][ni][4 * gq + e] + bc, h[e], l[e]);
              const long o = ((long)b * g.vt_rows + feat) * g.vt_ld + n0;
              if ((g.seq_len & 3) == 0) {            // 4 tokens stay inside one utterance and are 8-B aligned
                *reinterpret_cast<uint2*>(g.vt_hi + o) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
                if (g.vt_lo) *reinterpret_cast<uint2*>(g.vt_lo + o) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int row = row0 + e;
                  if (row >= g.M) break;
                  const int bb = row / g.seq_len, nn = row - bb * g.seq_len;
                  const long oo = ((long)bb * g.vt_rows + feat) * g.vt_ld + nn;
                  g.vt_hi[oo] = h[e];
                  if (g.vt_lo) g.vt_lo[oo] = l[e];
                }
              }
            }
          }
        }
      }
  }
}

template <int NSPLIT, int EPI>
static hipError_t launch_one(const GemmArgs& g, hipStream_t s) {
  const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM;
  const int nz = g.nz > 0 ? g.nz : 1;
  const size_t lds = 2 * 2 * (NSPLIT == 3 ? 2 : 1) * PLANE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<NSPLIT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<NSPLIT, EPI>), dim3(ntn * ntm * nz), dim3(256), lds, s, g);
  return hipGetLastError();
}

template <int NSPLIT>
static hipError_t launch_epi(const GemmArgs& g, hipStream_t s) {
  switch (g.epi) {
    case EPI_F32: return launch_one<NSPLIT, EPI_F32>(g, s);
    case EPI_SPLIT: return launch_one<NSPLIT, EPI_SPLIT>(g, s);
    case EPI_QKV: return launch_one<NSPLIT, EPI_QKV>(g, s);
    case EPI_GEGLU: return launch_one<NSPLIT, EPI_GEGLU>(g, s);
    case EPI_WAVENET: return launch_one<NSPLIT, EPI_WAVENET>(g, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm(const GemmArgs& g, int nsplit, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.nkt <= 0) return hipErrorInvalidValue;
  if (nsplit == 3) {
    if (!g.a_lo || !g.w_lo) return hipErrorInvalidValue;
    return launch_epi<3>(g, s);
  }
  return launch_epi<1>(g, s);
}

}  // namespace ns2
