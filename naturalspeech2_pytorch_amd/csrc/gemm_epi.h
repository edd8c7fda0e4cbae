// Shared epilogues of the GEMM kernels (gemm.hip: 128x128 tile; gemm2.hip: 256x256 tile).
// A wave owns MI x NI accumulator tiles of v_mfma_f32_32x32x16_bf16 (C layout: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)), rows [row_base, row_base + 32*MI), cols [col_base, col_base + 32*NI).
#pragma once
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

// WavenetResBlock NS2:629-636: h = conv(x)+b ; h = h*gamma_t+beta_t ; h = tanh(h)*sigmoid(h) ; (then += res_conv(x)).
// tanh(h)*sigmoid(h) = sign(h) * (1-u) * (h<0 ? u : 1) / (1+u^2),  u = exp(-|h|)   (one exp, no overflow)
template <int MI, int NI>
NS2_DEVINL void wavenet_midgate(f32x16 (&acc)[MI][NI], const GemmArgs& g, int z, int row_base, int col_base, int l31, int hi) {
  const float* film = g.film + (long)z * g.film_zs;
  const float* bias = g.bias + (long)z * g.bias_zs;
  const float* bias2 = g.bias2 + (long)z * g.bias_zs;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
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

// ocol_base: first output column of this wave for EPI_GEGLU (= half of the packed column index)
template <int EPI, int MI, int NI>
NS2_DEVINL void gemm_epilogue(f32x16 (&acc)[MI][NI], const GemmArgs& g, int z, int row_base, int col_base, int ocol_base,
                              int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  const bool odd = lane & 1;

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
          if (g.resid) v += g.resid[(long)row * g.ldr + col];
          g.out_f[(long)row * g.ldo_f + col] = v;
        }
      }
  } else if constexpr (EPI == EPI_GEGLU) {
    // wave tile = [x(32 cols) | gate(32 cols)] ; out[:, ocol] = gelu(gate) * x   (NS2:1006-1007)
    static_assert(EPI != EPI_GEGLU || NI == 2, "GEGLU needs a 64-column wave tile");
    const int ocol = ocol_base + l31;
    const int cx = col_base + l31, cg = col_base + 32 + l31;
    const float bx = g.bias[cx], bg = g.bias[cg];      // packed (padded) bias: always in range
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
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
                  if (row < g.M) {
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
}

}  // namespace ns2
