// Shared epilogues of the GEMM kernels (gemm.hip: 128x128 tile; gemm2.hip: 256x256 tile).
// A wave owns MI x NI accumulator tiles of v_mfma_f32_32x32x16_bf16 (C layout: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)), rows [row_base, row_base + 32*MI), cols [col_base, col_base + 32*NI).
#pragma once
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

// WavenetResBlock NS2:629-636: h = conv(x)+b ; h = h*gamma_t+beta_t ; h = tanh(h)*sigmoid(h) ; (then += res_conv(x)).
// tanh(h)*sigmoid(h) = sign(h) * (1-u) * (h<0 ? u : 1) / (1+u^2),  u = exp(-|h|)   (one exp, no overflow)
// `uni` (wave-uniform): the wave tile lies inside ONE utterance, so gamma / beta are per column and are loaded once per column tile
// instead of once per element (with a row / seq_len division each).  The per-element structure (one small diamond per value) is
// kept on purpose: a straight-line variant of this loop made the register allocator spill 100-300 VGPRs (gemm_epi_fast.h note).
template <int MI, int NI>
NS2_DEVINL void wavenet_midgate(f32x16 (&acc)[MI][NI], const GemmArgs& g, int z, int row_base, int col_base, int l31, int hi, bool uni = false) {
  const float* film = g.film + (long)z * g.film_zs;
  const float* bias = g.bias + (long)z * g.bias_zs;
  const float* bias2 = g.bias2 + (long)z * g.bias_zs;
  const float* film_u = film + (long)(uni ? row_base / g.seq_len : 0) * g.film_ld;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = col_base + ni * 32 + l31;
      const bool cok = col < g.N;
      const float bc = cok ? bias[col] : 0.f;
      const float b2 = cok ? bias2[col] : 0.f;
      const float gam_u = (uni && cok) ? film_u[col] : 0.f;
      const float bet_u = (uni && cok) ? film_u[g.N + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = 0.f;
        if (cok && row < g.M) {
          float gam = gam_u, bet = bet_u;
          if (!uni) {
            const int b = row / g.seq_len;
            gam = film[(long)b * g.film_ld + col];
            bet = film[(long)b * g.film_ld + g.N + col];
          }
          const float h = (acc[mi][ni][r] + bc) * gam + bet;
          const float u = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(h));     // v_exp_f32: <= 1 ulp, |arg| error ~1e-7*|h|
          const float t = (1.f - u) * (h < 0.f ? u : 1.f) * __builtin_amdgcn_rcpf(1.f + u * u);
          v = copysignf(t, h) + b2;
        }
        acc[mi][ni][r] = v;
      }
    }
}

// ocol_base: first output column of this wave for EPI_GEGLU (= half of the packed column index)
template <int EPI, int MI, int NI>
NS2_DEVINL void gemm_epilogue(f32x16 (&acc)[MI][NI], const GemmArgs& g, int z, int row_base, int col_base, int ocol_base,
                              int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  const bool odd = lane & 1;
  const bool il = g.out_lo != nullptr;               // interleaved 128-B output lines: bf16 [hi32|lo32] or FMT_H8 (ns2_common.h)

  if constexpr (EPI == EPI_F32) {
    // out = acc + bias (+ residual)      (to_out / FF-out / final_conv / to_pred)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = col_base + ni * 32 + l31;
        if (col >= g.N) continue;
        const float bc = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_base + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row >= g.M) continue;
          float v = acc[mi][ni][r] + bc;
          if (g.act) v = apply_act(v, g.act);
          if (g.resid) v += g.resid[(long)row * g.ldr + col];
          g.out_f[z * g.out_f_zs + (long)row * g.ldo_f + col] = v;
        }
      }
  } else if constexpr (EPI == EPI_GEGLU) {
    // wave tile = NI/2 x [x(32 cols) | gate(32 cols)] ; out[:, ocol] = gelu(gate) * x   (NS2:1006-1007)
    static_assert(EPI != EPI_GEGLU || (NI % 2) == 0, "GEGLU needs 64-column [x|gate] groups");
#pragma unroll
    for (int np = 0; np < NI / 2; ++np) {
    const int ocol = ocol_base + np * 32 + l31;
    const int cx = col_base + np * 64 + l31, cg = cx + 32;
    const float bx = g.bias[cx], bg = g.bias[cg];      // packed (padded) bias: always in range
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        float v0, v1;
        {
          const float x0 = acc[mi][2 * np][2 * rp] + bx, g0 = acc[mi][2 * np + 1][2 * rp] + bg;
          const float x1 = acc[mi][2 * np][2 * rp + 1] + bx, g1 = acc[mi][2 * np + 1][2 * rp + 1] + bg;
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
        if (row < g.M && col < g.out_ncols) store_cols2(g.out_hi + (long)row * pld(g.ldo_s, il), col, c_lo, c_hi, g.out_fmt, il);
      }
    }
    }
  } else {
    // EPI_SPLIT / EPI_QKV / EPI_WAVENET: split planes, optionally the tail columns transposed (V^T for attention)
    const float* bias = g.bias ? g.bias + (long)z * g.bias_zs : nullptr;
    const long zo = pcol((int)(z * g.out_zs), il);    // out_zs = logical column offset of slice z
    bf16_t* out_hi = g.out_hi + zo;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = col_base + ni * 32 + l31;
        float bc = 0.f;
        if constexpr (EPI != EPI_WAVENET) bc = (bias && col < g.N) ? bias[col] : 0.f;   // wavenet biases were applied mid-loop
        const bool transposed = (EPI == EPI_QKV) && (col_base + ni * 32 >= g.split_col);   // wave-uniform
        if (!transposed) {
#pragma unroll
          for (int rp = 0; rp < 8; ++rp) {
            float v0 = acc[mi][ni][2 * rp] + bc, v1 = acc[mi][ni][2 * rp + 1] + bc;
            if (EPI == EPI_SPLIT && g.act) { v0 = apply_act(v0, g.act); v1 = apply_act(v1, g.act); }
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
              store_cols2(out_hi + (long)row * pld(g.ldo_s, il), c0, c_lo, c_hi, g.out_fmt, il);
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
              // transposed values are attention operands: bf16 (with or without lo plane) or dense IEEE half, never FMT_H8
              uint32_t h01, l01, h23, l23;
              split2f(acc[mi][ni][4 * gq + 0] + bc, acc[mi][ni][4 * gq + 1] + bc, h01, l01, g.vt_fmt == FMT_F16);
              split2f(acc[mi][ni][4 * gq + 2] + bc, acc[mi][ni][4 * gq + 3] + bc, h23, l23, g.vt_fmt == FMT_F16);
              const bf16_t h[4] = {(bf16_t)(h01 & 0xffffu), (bf16_t)(h01 >> 16), (bf16_t)(h23 & 0xffffu), (bf16_t)(h23 >> 16)};
              const bf16_t l[4] = {(bf16_t)(l01 & 0xffffu), (bf16_t)(l01 >> 16), (bf16_t)(l23 & 0xffffu), (bf16_t)(l23 >> 16)};
              const bool vil = g.vt_lo != nullptr;
              const long o = ((long)b * g.vt_rows + feat) * pld(g.vt_ld, vil) + pcol(n0, vil);
              if ((g.seq_len & 3) == 0) {            // 4 tokens stay inside one utterance (and one 32-block), 8-B aligned
                *reinterpret_cast<uint2*>(g.vt_hi + o) = make_uint2(h01, h23);
                if (vil) *reinterpret_cast<uint2*>(g.vt_lo + o) = make_uint2(l01, l23);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int row = row0 + e;
                  if (row < g.M) {
                    const int bb = row / g.seq_len, nn = row - bb * g.seq_len;
                    const long oo = ((long)bb * g.vt_rows + feat) * pld(g.vt_ld, vil) + pcol(nn, vil);
                    g.vt_hi[oo] = h[e];
                    if (vil) g.vt_lo[oo] = l[e];
                  }
                }
              }
            }
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-staged epilogue for the 256x256 kernel (wave tile 128x64, MI = 4, NI = 2).  The direct epilogue above issues one
// 4-byte store per lane per two accumulator registers (64-B row segments): on short-K GEMMs (QKV, FF-in, out-proj) it
// measured ~100 us of a 145 us launch.  Here every wave transposes its tile through a private 18 KiB LDS region
// (free after the K loop) and writes full rows with 16-B stores per lane (128-B / 256-B contiguous segments).
constexpr int EPI_LDS_WAVE_BYTES = 18432;     // 128 rows x (128 B + 16 B pad)

NS2_DEVINL uint32_t pk2(float a, float b) { return cvt2(a, b); }

// Measured (MI355X, M = 32768): the fp32 + residual epilogue gains 1.45x-1.65x on the whole launch (FF-out 218 -> 150 us);
// LDS-staged variants of the bf16 split-plane epilogues were built too and were neutral (their cost is the store burst
// itself, not store issue) -- only EPI_F32 has an LDS path.
template <int EPI>
NS2_DEVINL bool epi_lds_supported(const GemmArgs&, int) { return EPI == EPI_F32; }

// NIT = accumulator column tiles of the wave, NI0 = first of the two column tiles this call stores
template <int EPI, int NIT, int NI0>
NS2_DEVINL void gemm_epilogue_lds(f32x16 (&acc)[4][NIT], const GemmArgs& g, int z, int row_base, int col_base, int ocol_base,
                                  int lane, unsigned char* wbuf) {
  const int l31 = lane & 31, hi = lane >> 5;
  const bool odd = lane & 1;

  if constexpr (EPI == EPI_F32) {
    // two halves of 64 rows x 64 cols fp32, LDS rows of 272 B
    constexpr int RS = 272;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = col_base + ni * 32 + l31;
          const float bc = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            float t = acc[half * 2 + mh][NI0 + ni][r] + bc;
            if (g.act) t = apply_act(t, g.act);
            *reinterpret_cast<float*>(wbuf + lr * RS + (ni * 32 + l31) * 4) = t;
          }
        }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int lr = it * 4 + (lane >> 4), ch = lane & 15;
        const int row = row_base + half * 64 + lr, col = col_base + ch * 4;
        const float4 v = *reinterpret_cast<const float4*>(wbuf + lr * RS + ch * 16);
        if (row < g.M && col < g.N) {
          float o[4] = {v.x, v.y, v.z, v.w};
          if (col + 3 < g.N) {
            if (g.resid) {
              const float4 rr = *reinterpret_cast<const float4*>(g.resid + (long)row * g.ldr + col);
              o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
            }
            *reinterpret_cast<float4*>(g.out_f + z * g.out_f_zs + (long)row * g.ldo_f + col) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) {
                float t = o[e];
                if (g.resid) t += g.resid[(long)row * g.ldr + col + e];
                g.out_f[z * g.out_f_zs + (long)row * g.ldo_f + col + e] = t;
              }
          }
        }
      }
    }
  }
}

}  // namespace ns2
