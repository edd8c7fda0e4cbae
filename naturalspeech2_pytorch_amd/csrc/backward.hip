// Backward (training) kernels of the NaturalSpeech2 denoiser on gfx950 -- SURVEY §8f-4: the arithmetic behind
// `loss.backward()` of NS2:1635-1666, 1886.
//
// How the backward pass maps onto the machine:
//   * every contraction of the backward pass is a call of the FORWARD GEMM family (gemm.hip / gemm2.hip):
//       dgrad  dX = dY W          -> the same kernel on a second pack of the weight (transposed, taps flipped), pad_left = 0
//       wgrad  dW = dY^T X        -> the same kernel on TRANSPOSED operand planes (contraction over the M = B*N tokens),
//                                    split-K over grid-z into fixed slots, summed in a fixed order (deterministic, no atomics)
//     so what this file adds around them is data movement and pointwise calculus:
//   * tplanes_kernel: fp32 gradient (or operand planes) -> row planes + transposed planes (+ column sums = bias gradients) in
//     one pass; the transposed copy optionally row-shifted per utterance (tap t of a causal conv reads x[n - (k-1-t) dil]);
//   * the pointwise derivatives: FiLM + tanh*sigmoid gate (NS2:629-636), GEGLU (NS2:1004-1007), RMSNorm (NS2:727-746) with
//     the per-utterance reductions for the conditioning gradients left in fixed slots;
//   * flash-attention backward (ATT:77-155): P is recomputed from the forward's log-sum-exp; one kernel template, two roles
//     (dQ: a workgroup owns 128 queries and walks the keys; dK/dV: owns 128 keys and walks the queries), the same swapped
//     MFMA products and LDS tile shapes as attention.hip.
// Arithmetic: bf16 hi/lo planes, hi*hi + hi*lo + lo*hi (precision 3, "exact"): gradients are fp32-class like the forward.
#include <cstdlib>
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

// ================================================================================================ tplanes
// One 64 x 64 tile per workgroup: load (fp32 -> split, or planes), optional row-plane store, transpose through LDS, store
// transposed interleaved lines along the token axis.  H8 = false: bf16 lines [hi32 | lo32] (the exact arithmetic); H8 = true: FMT_H8
// lines [half32 | e5m2(x) 32 B | e5m2((x - half(x)) 2^12) 32 B] (the mixed training arithmetic).
// In LDS an element is ONE 32-bit word -- hi | lo << 16, resp. half | e5m2(x) << 16 | remainder << 24 -- so the tile is written with
// ds_write_b32 and read back down its columns with ds_read_b32 (row stride 65 words: conflict-free both ways), and the thread that
// gathers 8 tokens of a column writes BOTH parts of the output line from the same 8 words.  (Round 4 kept two 16-bit arrays: twice
// the LDS instructions, 2-byte accesses; the kernels ran at 2.7-3.2 TB/s of their traffic and were 17 ms of a training step.)
template <bool IN_F32, bool H8>
__global__ __launch_bounds__(256) void tplanes_kernel(const TPlanesArgs a) {
  __shared__ uint32_t tw[64][65];
  __shared__ float cs[16][64];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * 64;
  // rows of this tile: per_batch -> utterance b, positions n0 .. n0 + 63; else global rows m0 .. m0 + 63
  int b = 0;
  long n0;
  if (a.per_batch) {
    const int ntn = (int)((a.ld_t + 63) / 64);
    b = blockIdx.y / ntn;
    n0 = (long)(blockIdx.y % ntn) * 64;
  } else {
    n0 = (long)blockIdx.y * 64;
  }
  // source row of tile row i (or -1): the output position takes input row (m - shift) of the same utterance
  auto src_row = [&](int i) -> long {
    long m, n;
    if (a.per_batch) { n = n0 + i; if (n >= a.seq_len) return -1; m = (long)b * a.seq_len + n; }
    else { m = n0 + i; if (m >= a.M) return -1; n = a.seq_len > 0 ? m % a.seq_len : m; }
    const long ns = n - a.shift;
    if (a.seq_len > 0 ? (ns < 0 || ns >= a.seq_len) : (m - a.shift < 0 || m - a.shift >= a.M)) return -1;
    return m - a.shift;
  };

  if constexpr (IN_F32) {
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int ch = tid & 15;                    // 4 columns c0 + 4 ch ..
    const bool vec = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.xf) & 15) == 0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int i = (tid >> 4) + 16 * pass;
      const long sr = src_row(i);
      const int c = c0 + 4 * ch;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (sr >= 0 && c < a.C) {
        if (vec && c + 3 < a.C) {
          const float4 t = *reinterpret_cast<const float4*>(a.xf + sr * a.ldx + c);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.C) v[e] = a.xf[sr * a.ldx + c + e];
        }
      }
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        uint32_t w0, w1;
        if constexpr (H8) {
          uint32_t h16, h8, l8;
          cvt2_h8(v[2 * e2], v[2 * e2 + 1], h16, h8, l8);          // (counts values beyond the half range: the loss-scale overflow check)
          w0 = (h16 & 0xffffu) | ((h8 & 0xffu) << 16) | ((l8 & 0xffu) << 24);
          w1 = (h16 >> 16) | (((h8 >> 8) & 0xffu) << 16) | (((l8 >> 8) & 0xffu) << 24);
        } else {
          uint32_t ph, pl;
          split2(v[2 * e2], v[2 * e2 + 1], ph, pl);
          w0 = (ph & 0xffffu) | (pl << 16);
          w1 = (ph >> 16) | (pl & 0xffff0000u);
        }
        tw[i][4 * ch + 2 * e2] = w0; tw[i][4 * ch + 2 * e2 + 1] = w1;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += v[e];
      // row planes of the UNSHIFTED rows (shift must be 0 when they are requested; checked by the launcher)
      if (a.row_hi) {
        const long m = a.per_batch ? (long)b * a.seq_len + n0 + i : n0 + i;
        const bool rok = a.per_batch ? (n0 + i < a.seq_len) : (m < a.M);
        if (rok && c < a.ld_row) store_cols4(a.row_hi + m * 2L * a.ld_row, c, v[0], v[1], v[2], v[3], H8 ? FMT_H8 : FMT_BF16, true);
      }
    }
    if (a.colsum_partial) {                     // fixed-order column sums of this 64-row tile (bias gradients)
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[tid >> 4][4 * ch + e] = csum[e];
    }
  } else {
    // plane input: a thread takes 8 columns of one row -- the 16-byte chunk of the first part and the matching bytes of the second
    const int g8 = tid & 7;                     // columns c0 + 8 g8 .. (+ 7)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int i = (tid >> 3) + 32 * pass;
      const long sr = src_row(i);
      const int c = c0 + 8 * g8;
      uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      if (sr >= 0 && c < a.C) {                  // C is a multiple of 8 for plane inputs (checked by the launcher)
        const bf16_t* line = a.in_hi + sr * 2L * a.ld_in + pcol((a.in_col0 + c) & ~31, true);      // the 128-byte line of this column group
        const int e0 = (a.in_col0 + c) & 31;
        const uint4 p0 = *reinterpret_cast<const uint4*>(line + e0);
        const uint32_t hw[4] = {p0.x, p0.y, p0.z, p0.w};
        if constexpr (H8) {
          const unsigned char* bytes = reinterpret_cast<const unsigned char*>(line);
          const uint2 b8 = *reinterpret_cast<const uint2*>(bytes + 64 + e0), l8 = *reinterpret_cast<const uint2*>(bytes + 96 + e0);
          const uint32_t bw[2] = {b8.x, b8.y}, lw[2] = {l8.x, l8.y};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            w[e] = ((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu) | (((bw[e >> 2] >> (8 * (e & 3))) & 0xffu) << 16) |
                   (((lw[e >> 2] >> (8 * (e & 3))) & 0xffu) << 24);
        } else {
          const uint4 p1 = *reinterpret_cast<const uint4*>(line + 32 + e0);
          const uint32_t lw[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            w[e] = ((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu) | (((lw[e >> 1] >> (16 * (e & 1))) & 0xffffu) << 16);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) tw[i][8 * g8 + e] = w[e];
    }
  }
  __syncthreads();
  if (IN_F32 && a.colsum_partial && tid < 64) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += cs[r][tid];
    if (c0 + tid < a.C) a.colsum_partial[(long)blockIdx.y * a.C + c0 + tid] = s;
  }
  if (!a.t_hi) return;
  // ---- transposed store: output row = column c of the tile; a thread gathers 8 token positions of one column (8 words) and writes
  // both parts of them: bf16 16 + 16 bytes, FMT_H8 16 + 8 + 8 bytes
  const int g8 = tid & 7;                       // tokens n0 + 8 g8 .. (+ 7)
  const int rows_out = a.per_batch ? a.t_rows_per_batch : a.t_rows;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cr = (tid >> 3) + 32 * pass;
    const int c = c0 + cr;
    const long col = n0 + 8 * g8;               // first token position
    if (c >= rows_out || (col & ~31L) >= a.ld_t) continue;
    uint32_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = tw[8 * g8 + e][cr];
    const long orow = a.per_batch ? (long)b * a.t_rows_per_batch + c : c;
    bf16_t* line = a.t_hi + orow * 2L * a.ld_t + 2L * (col & ~31L);          // the 128-byte line of these token positions
    const int e0 = (int)(col & 31);
    uint32_t p0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) p0[e] = (w[2 * e] & 0xffffu) | (w[2 * e + 1] << 16);
    *reinterpret_cast<uint4*>(line + e0) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
    if constexpr (H8) {
      unsigned char* bytes = reinterpret_cast<unsigned char*>(line);
      uint32_t hb[2], lb[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        hb[e] = ((w[4 * e] >> 16) & 0xffu) | (((w[4 * e + 1] >> 16) & 0xffu) << 8) | (((w[4 * e + 2] >> 16) & 0xffu) << 16) | (((w[4 * e + 3] >> 16) & 0xffu) << 24);
        lb[e] = (w[4 * e] >> 24) | ((w[4 * e + 1] >> 24) << 8) | ((w[4 * e + 2] >> 24) << 16) | ((w[4 * e + 3] >> 24) << 24);
      }
      *reinterpret_cast<uint2*>(bytes + 64 + e0) = make_uint2(hb[0], hb[1]);
      *reinterpret_cast<uint2*>(bytes + 96 + e0) = make_uint2(lb[0], lb[1]);
    } else {
      uint32_t p1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p1[e] = (w[2 * e] >> 16) | (w[2 * e + 1] & 0xffff0000u);
      *reinterpret_cast<uint4*>(line + 32 + e0) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    }
  }
}

hipError_t launch_tplanes(const TPlanesArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.C <= 0) return hipErrorInvalidValue;
  const bool in_f32 = a.xf != nullptr;
  if (in_f32 == (a.in_hi != nullptr)) return hipErrorInvalidValue;                 // exactly one input form
  if (!in_f32 && (!planes_ok(a.in_hi, a.in_lo) || !a.in_lo || (a.C & 7) || (a.in_col0 & 31) || (a.ld_in & 31))) return hipErrorInvalidValue;
  if (a.row_hi && (!in_f32 || a.shift != 0 || !a.row_lo || a.row_lo != a.row_hi + 32 || (a.ld_row & 31) || a.ld_row < a.C))
    return hipErrorInvalidValue;
  if (a.t_hi && (!a.t_lo || a.t_lo != a.t_hi + 32 || (a.ld_t & 31))) return hipErrorInvalidValue;
  if (a.colsum_partial && (!in_f32 || a.per_batch || a.shift != 0)) return hipErrorInvalidValue;
  if (a.per_batch && (a.seq_len <= 0 || a.M % a.seq_len || a.ld_t < a.seq_len || a.t_rows_per_batch < a.C)) return hipErrorInvalidValue;
  if (!a.per_batch && a.t_hi && (a.ld_t < a.M || a.t_rows < a.C)) return hipErrorInvalidValue;
  if (a.seq_len > 0 && a.M % a.seq_len) return hipErrorInvalidValue;
  int ccover = a.C;
  if (a.t_hi) ccover = max(ccover, a.per_batch ? a.t_rows_per_batch : a.t_rows);
  if (a.row_hi) ccover = max(ccover, a.ld_row);
  long ntile_m;
  if (a.per_batch) ntile_m = (long)(a.M / a.seq_len) * ((a.ld_t + 63) / 64);
  else ntile_m = tplanes_slices(a.M, a.t_hi ? a.ld_t : 0);
  const dim3 grid((ccover + 63) / 64, (unsigned)ntile_m);
  if (a.fmt != FMT_BF16 && a.fmt != FMT_H8) return hipErrorInvalidValue;
  if (a.fmt == FMT_H8) {
    if (in_f32) hipLaunchKernelGGL((tplanes_kernel<true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tplanes_kernel<false, true>), grid, dim3(256), 0, s, a);
  } else {
    if (in_f32) hipLaunchKernelGGL((tplanes_kernel<true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tplanes_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}
long tplanes_slices(int M, long ld_t) { return (max((long)M, ld_t) + 63) / 64; }

// ================================================================================================ fixed-order reductions
// out[o * inner + j] (+)= sum_s partial[(o * S + s) * inner + j].  A workgroup owns 32 columns of one `o`; its 8 row groups each
// sum every 8th slot, then the 8 partial sums are added in group order: a fixed order whatever the launch -- deterministic.
// (The first version gave each output ONE thread looping over all S slots: 512 slots x a few hundred columns = a handful of
// waves doing 512 dependent passes, 88 us per call and 9 % of a training step.)
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* partial, long outer, int S, long inner, float* out, int accumulate) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long j = (long)blockIdx.x * 32 + tx;
  const long o = blockIdx.y;
  float v = 0.f;
  if (j < inner)
    for (int s = ty; s < S; s += 8) v += partial[(o * S + s) * inner + j];
  red[ty][tx] = v;
  __syncthreads();
  if (ty == 0 && j < inner) {
    float t = red[0][tx];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][tx];
    out[o * inner + j] = accumulate ? out[o * inner + j] + t : t;
  }
}
hipError_t launch_reduce_slices(const float* partial, long outer, int S, long inner, float* out, int accumulate, hipStream_t s) {
  if (outer <= 0 || S <= 0 || inner <= 0 || outer > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((inner + 31) / 32), (unsigned)outer), dim3(256), 0, s, partial, outer, S, inner, out, accumulate);
  return hipGetLastError();
}
// weight gradient from the split-K slots of the wgrad GEMM: partial [S][R][ldp], column t * Kp + k  ->  out [R, K, T] (the
// nn.Conv1d / nn.Linear weight layout), summed over s in a fixed order
__global__ void wgrad_reduce_kernel(const float* partial, int S, int R, long ldp, int T, int Kp, int K, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;       // k fastest: the slot reads of a wave are contiguous
  const long n = (long)R * K * T;
  if (i >= n) return;
  const int k = (int)(i % K);
  const long rt = i / K;
  const int t = (int)(rt % T);
  const long r = rt / T;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += partial[((long)s * R + r) * ldp + (long)t * Kp + k];
  out[(r * K + k) * T + t] = v;
}
hipError_t launch_wgrad_reduce(const float* partial, int S, int R, long ldp, int T, int Kp, int K, float* out, hipStream_t s) {
  if (S <= 0 || R <= 0 || T <= 0 || K <= 0 || Kp < K || ldp < (long)T * Kp) return hipErrorInvalidValue;
  const long n = (long)R * K * T;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, S, R, ldp, T, Kp, K, out);
  return hipGetLastError();
}

// ================================================================================================ FiLM + gate (NS2:629-636)
// z = h * gamma_b + beta_b ; g = tanh(z) * sigmoid(z)
NS2_DEVINL float gate_fn(float z) {
  const float u = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(z));
  const float t = (1.f - u) * (z < 0.f ? u : 1.f) * __builtin_amdgcn_rcpf(1.f + u * u);      // the forward's formula (gemm_epi.h)
  return copysignf(t, z);
}
NS2_DEVINL float gate_grad(float z) {            // d/dz tanh(z) sigmoid(z) = (1 - tanh^2) sig + tanh sig (1 - sig)
  const float sg = 1.0f / (1.0f + expf(-z));
  const float th = tanhf(z);
  return (1.f - th * th) * sg + th * sg * (1.f - sg);
}
__global__ __launch_bounds__(256) void film_gate_fwd_kernel(const float* h, long ldh, const float* film, int film_ld, int seq_len,
                                                            long M, int d, float* out, long ldo) {
  const int chunks = d >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * chunks) return;
  const long row = idx / chunks;
  const int c = (int)(idx - row * chunks) * 4;
  const float* fb = film + (row / seq_len) * film_ld;
  const float4 hv = *reinterpret_cast<const float4*>(h + row * ldh + c);
  const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = gate_fn(hh[e] * fb[c + e] + fb[d + c + e]);
  *reinterpret_cast<float4*>(out + row * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
}
hipError_t launch_film_gate_fwd(const float* h, long ldh, const float* film, int film_ld, int seq_len, long M, int d, float* out,
                                long ldo, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 3) || (ldh & 3) || (ldo & 3) || seq_len <= 0) return hipErrorInvalidValue;
  const long n = M * (d >> 2);
  hipLaunchKernelGGL(film_gate_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h, ldh, film, film_ld, seq_len, M, d, out, ldo);
  return hipGetLastError();
}
// backward: dh = dg * g'(z) * gamma ; dgamma[b, c] = sum_n dg g'(z) h ; dbeta[b, c] = sum_n dg g'(z).  A workgroup owns 64 columns
// x FG_ROWS positions of ONE utterance and leaves its two column sums in slot (b * nchunk + chunk) of `partial` ([.., 2 d]).
constexpr int FG_ROWS = 256;
__global__ __launch_bounds__(256) void film_gate_bwd_kernel(const float* dg, long lddg, const float* h, long ldh, const float* film,
                                                            int film_ld, int seq_len, int d, float* dh, long lddh, float* partial) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int nchunk = (seq_len + FG_ROWS - 1) / FG_ROWS;
  const int b = blockIdx.y / nchunk, chunk = blockIdx.y % nchunk;
  const float* fb = film + (long)b * film_ld;
  float sg = 0.f, sb = 0.f;
  if (c < d) {
    const float gam = fb[c], bet = fb[d + c];
    for (int i = ty; i < FG_ROWS; i += 4) {
      const int n = chunk * FG_ROWS + i;
      if (n >= seq_len) break;
      const long row = (long)b * seq_len + n;
      const float hv = h[row * ldh + c];
      const float dz = dg[row * lddg + c] * gate_grad(hv * gam + bet);
      dh[row * lddh + c] = dz * gam;
      sg += dz * hv;
      sb += dz;
    }
  }
  red[0][ty][tx] = sg; red[1][ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && c < d) {
    float* slot = partial + (long)blockIdx.y * 2 * d;
    slot[c] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    slot[d + c] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
  }
}
int film_gate_slices(int seq_len) { return (seq_len + FG_ROWS - 1) / FG_ROWS; }
hipError_t launch_film_gate_bwd(const float* dg, long lddg, const float* h, long ldh, const float* film, int film_ld, int B, int seq_len,
                                int d, float* dh, long lddh, float* partial, hipStream_t s) {
  if (B <= 0 || seq_len <= 0 || d <= 0 || !partial) return hipErrorInvalidValue;
  hipLaunchKernelGGL(film_gate_bwd_kernel, dim3((d + 63) / 64, B * film_gate_slices(seq_len)), dim3(256), 0, s, dg, lddg, h, ldh, film,
                     film_ld, seq_len, d, dh, lddh, partial);
  return hipGetLastError();
}

// ================================================================================================ GEGLU (NS2:1004-1007)
// pre [M, ldp] = [x (f) | gate (f)] -> h = gelu(gate) * x as operand planes [M, ldo] (zero beyond f)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* pre, long ldp, long M, int f, bf16_t* out_hi, bf16_t* out_lo, int ldo, int fmt) {
  const int chunks = ldo >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * chunks) return;
  const long row = idx / chunks;
  const int c = (int)(idx - row * chunks) * 4;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  const float* p = pre + row * ldp;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c + e < f) o[e] = gelu_erf(p[f + c + e]) * p[c + e];
  store_cols4(out_hi + row * 2L * ldo, c, o[0], o[1], o[2], o[3], fmt, out_lo != nullptr);
}
hipError_t launch_geglu_fwd(const float* pre, long ldp, long M, int f, bf16_t* out_hi, bf16_t* out_lo, int ldo, hipStream_t s, int fmt) {
  if (M <= 0 || f <= 0 || ldp < 2L * f || (ldo & 31) || ldo < f || !out_lo || out_lo != out_hi + 32) return hipErrorInvalidValue;
  if (fmt != FMT_BF16 && fmt != FMT_H8) return hipErrorInvalidValue;
  const long n = M * (ldo >> 2);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, ldp, M, f, out_hi, out_lo, ldo, fmt);
  return hipGetLastError();
}
// dpre[:, c] = dh * gelu(gate) ; dpre[:, f + c] = dh * x * (Phi(gate) + gate * phi(gate))
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* dh, long lddh, const float* pre, long ldp, long M, int f, float* dpre,
                                                        long lddp) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * f) return;
  const long row = idx / f;
  const int c = (int)(idx - row * f);
  const float x = pre[row * ldp + c], g = pre[row * ldp + f + c], dy = dh[row * lddh + c];
  const float Phi = 0.5f * erfc_fast(-0.70710678118654752440f * g);
  const float phi = 0.39894228040143267794f * __expf(-0.5f * g * g);
  dpre[row * lddp + c] = dy * g * Phi;
  dpre[row * lddp + f + c] = dy * x * (Phi + g * phi);
}
hipError_t launch_geglu_bwd(const float* dh, long lddh, const float* pre, long ldp, long M, int f, float* dpre, long lddp, hipStream_t s) {
  if (M <= 0 || f <= 0 || ldp < 2L * f || lddp < 2L * f || lddh < f) return hipErrorInvalidValue;
  const long n = M * f;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dh, lddh, pre, ldp, M, f, dpre, lddp);
  return hipGetLastError();
}

// ================================================================================================ RMSNorm backward (NS2:727-746)
// y = nh * gp * gc + bc,  nh = x * r,  r = sqrt(d) / max(|x|, eps)   (gp = learned gamma or 1, gc / bc = adaptive or 1 / 0)
//   dx = r * (dn - nh * (nh . dn) / d),  dn = dy * gc * gp
//   dgc[b, c] = sum_n dy nh gp ; dbc[b, c] = sum_n dy ; dgp[c] = sum_m dy gc nh
// One wave per row; a workgroup walks NB_ROWS rows of one utterance and leaves its column sums in its slot.
constexpr int NB_ROWS = 64;
template <int CH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const NormBwdArgs a) {
  extern __shared__ float red[];                  // [3][4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = (a.seq_len + NB_ROWS - 1) / NB_ROWS;
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
  const float* gc = a.cond ? a.cond + (long)b * a.cond_ld : nullptr;
  float gm[CH][4], gg[CH][4];
  float agc[CH][4], abc[CH][4], agp[CH][4];
#pragma unroll
  for (int j = 0; j < CH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = lane * 4 + 256 * j + e;
      gm[j][e] = (a.gamma && c < a.d) ? a.gamma[c] : 1.f;
      gg[j][e] = (gc && c < a.d) ? gc[c] : 1.f;
      agc[j][e] = abc[j][e] = agp[j][e] = 0.f;
    }
  const float scale = sqrtf((float)a.d), invd = 1.0f / (float)a.d;
  for (int i = wave; i < NB_ROWS; i += 4) {
    const int n = chunk * NB_ROWS + i;
    if (n >= a.seq_len) break;
    const long row = (long)b * a.seq_len + n;
    float4 xv[CH], dv[CH];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = lane * 4 + 256 * j;
      const bool ok = c < a.d;
      xv[j] = ok ? *reinterpret_cast<const float4*>(a.x + row * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      dv[j] = ok ? *reinterpret_cast<const float4*>(a.dy + row * a.lddy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += xv[j].x * xv[j].x + xv[j].y * xv[j].y + xv[j].z * xv[j].z + xv[j].w * xv[j].w;
    }
    ss = wave_sum(ss);
    const float r = scale / fmaxf(sqrtf(ss), 1e-12f);
    float dot = 0.f;
    float nh[CH][4], dn[CH][4];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float xx[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, dd[4] = {dv[j].x, dv[j].y, dv[j].z, dv[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        nh[j][e] = xx[e] * r;
        dn[j][e] = dd[e] * gg[j][e] * gm[j][e];
        dot += nh[j][e] * dn[j][e];
        agc[j][e] += dd[e] * nh[j][e] * gm[j][e];
        abc[j][e] += dd[e];
        agp[j][e] += dd[e] * gg[j][e] * nh[j][e];
      }
    }
    dot = wave_sum(dot) * invd;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = lane * 4 + 256 * j;
      if (c >= a.d) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = r * (dn[j][e] - nh[j][e] * dot);
      if (a.dx_add) {
        const float4 t = *reinterpret_cast<const float4*>(a.dx_add + row * a.lddx + c);
        o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
      }
      *reinterpret_cast<float4*>(a.dx + row * a.lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  // column sums of this workgroup's rows: waves -> LDS -> slot
  float* r0 = red, *r1 = red + 4 * a.d, *r2 = red + 8 * a.d;
#pragma unroll
  for (int j = 0; j < CH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = lane * 4 + 256 * j + e;
      if (c < a.d) { r0[wave * a.d + c] = agc[j][e]; r1[wave * a.d + c] = abc[j][e]; r2[wave * a.d + c] = agp[j][e]; }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < a.d; c += 256) {
    if (a.cond_partial) {
      float* slot = a.cond_partial + (long)blockIdx.x * 2 * a.d;
      slot[c] = r0[c] + r0[a.d + c] + r0[2 * a.d + c] + r0[3 * a.d + c];
      slot[a.d + c] = r1[c] + r1[a.d + c] + r1[2 * a.d + c] + r1[3 * a.d + c];
    }
    if (a.gamma_partial) a.gamma_partial[(long)blockIdx.x * a.d + c] = r2[c] + r2[a.d + c] + r2[2 * a.d + c] + r2[3 * a.d + c];
  }
}
int rmsnorm_bwd_slices(int seq_len) { return (seq_len + NB_ROWS - 1) / NB_ROWS; }
hipError_t launch_rmsnorm_bwd(const NormBwdArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.seq_len <= 0 || a.d <= 0 || (a.d & 3) || a.d > 2048 || (a.ldx & 3) || (a.lddy & 3) || (a.lddx & 3)) return hipErrorInvalidValue;
  if (a.cond && !a.cond_partial) return hipErrorInvalidValue;
  const dim3 grid(a.B * rmsnorm_bwd_slices(a.seq_len));
  const size_t lds = (size_t)12 * a.d * sizeof(float);
  if (a.d <= 256) hipLaunchKernelGGL(rmsnorm_bwd_kernel<1>, grid, dim3(256), lds, s, a);
  else if (a.d <= 512) hipLaunchKernelGGL(rmsnorm_bwd_kernel<2>, grid, dim3(256), lds, s, a);
  else if (a.d <= 1024) hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

// ================================================================================================ attention backward (ATT:77-155)
// delta[b, h, q] = sum_d dO[q, 64 h + d] * O[q, 64 h + d]   (O from its operand planes, hi + lo)
// One wave per token row, a lane owns 8 consecutive features (lane = 8 * head-in-group + part; 64 lanes = 512 features, wider rows
// take several passes): two float4 of dO and the matching 16-byte chunks of the o line per lane, 3 xor-shuffles inside the 8 lanes of
// a head.  (Round 4's kernel gave every (row, head) ONE thread reading 64 strided floats: 475 us per call, 5.7 ms of a training
// step, for 128 MB of traffic; this one streams.)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* dO, long lddo, const bf16_t* o_hi, int ldo, int B, int H, int Nq,
                                                         float* delta, int o_fmt) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long M = (long)B * Nq;
  if (m >= M) return;
  const long b = m / Nq, q = m - b * Nq;
  for (int f0 = 8 * lane; f0 < 64 * H; f0 += 512) {           // wave-uniform trip count (64 H is a multiple of 64)
    const float4 g0 = *reinterpret_cast<const float4*>(dO + m * lddo + f0), g1 = *reinterpret_cast<const float4*>(dO + m * lddo + f0 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const bf16_t* line = o_hi + m * 2L * ldo + ((f0 & ~31) << 1);     // the 128-byte line of 32 features
    const int e0 = f0 & 31;
    const uint4 hv = *reinterpret_cast<const uint4*>(line + e0);
    const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
    float o[8];
    if (o_fmt == FMT_H8) {
      // the value the out-projection multiplies: half + the e5m2 remainder (2^12-scaled), FMT_H8 line = [half32 | e5m2(x) | remainder]
      const uint2 lv = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(line) + 96 + e0);
      const uint32_t lw[2] = {lv.x, lv.y};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = h2f((bf16_t)((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu)) + bf8_to_f((lw[e >> 2] >> (8 * (e & 3))) & 0xffu) * (1.0f / H8_LO_SCALE);
    } else {
      const uint4 lv = *reinterpret_cast<const uint4*>(line + 32 + e0);
      const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = bf2f((bf16_t)((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu)) + bf2f((bf16_t)((lw[e >> 1] >> (16 * (e & 1))) & 0xffffu));
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(g[e], o[e], s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if ((lane & 7) == 0) delta[(b * H + (f0 >> 6)) * Nq + q] = s;
  }
}
hipError_t launch_attn_delta(const float* dO, long lddo, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, int B, int H, int Nq, float* delta,
                             hipStream_t s, int o_fmt) {
  if (B <= 0 || H <= 0 || Nq <= 0 || !o_lo || o_lo != o_hi + 32 || (ldo & 31) || ldo < 64 * H) return hipErrorInvalidValue;
  if (o_fmt != FMT_BF16 && o_fmt != FMT_H8) return hipErrorInvalidValue;
  if ((lddo & 3) || (reinterpret_cast<uintptr_t>(dO) & 15) || (reinterpret_cast<uintptr_t>(o_hi) & 15)) return hipErrorInvalidValue;
  const long n = (long)B * Nq;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dO, lddo, o_hi, ldo, B, H, Nq, delta, o_fmt);
  return hipGetLastError();
}

// Flash backward.  ROLE 0 (dQ): the workgroup OWNS 128 queries (4 waves x 32), walks 64-key tiles:
//     S^T = K Q^T ; dP^T = V dO^T ; P = exp2(S sl2 - lse) ; dS = P (dP - delta) ; dQ^T += K^T dS^T
//   ROLE 1 (dK, dV): owns 128 keys, walks 64-query tiles:
//     S = Q K^T ; dP = dO V^T ; P, dS as above with lse / delta per query (per register) ; dV^T += dO^T P ; dK^T += Q^T dS
// In both roles the first two products put the OWN row in the lane (col = lane & 31 of the MFMA C tile) and 16 walked rows in
// the registers -- the layout of attention.hip -- so P / dS feed the second pair of products as B operands straight from the
// registers, against TRANSPOSED fragments of the walked tiles (K^T, Q^T, dO^T: [64 d][64 walked rows]).
// Round 4 staged FOUR tiles per 64 walked rows through registers into a single LDS buffer -- Y, Yg row-major and Y1T (, Y2T) from
// TRANSPOSED copies of the same tensors that tplanes passes had written per utterance -- between two barriers: nothing overlapped
// the loads but the other workgroup of the CU, and the kernel ran at 5.2 x the forward (tools/experiments/
// r4_attention_backward_on_transposed_copies.hip; A/B profiles/r05_attention_backward_ab.json: same bits, 1.18 -> 0.655 ms per layer).
// Round 5: (1) only the row-major tiles are fetched; the transposed fragments (K^T for dQ; Q^T, dO^T for dK / dV) are read out of the SAME
// LDS tiles with ds_read_b64_tr_b16 (a lane receives 4 walked rows of ITS d column from a [4 rows][16 d] block; two reads = one 8-deep
// MFMA fragment) -- half the bytes per tile, and the per-utterance transposes of q, k and dO disappear from the step; (2) the tiles
// go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers) into TWO stages: the tile t + 1 is requested before
// the products of tile t and has their whole duration to land; one barrier per tile.
// LDS image of a tile: 64 rows x 256 B, a row = the head's 64 columns of one token exactly as they lie in memory
// [hi 0-31 | lo 0-31 | hi 32-63 | lo 32-63] = 16 chunks of 16 B; chunk c of row r is stored at c ^ f(r), f(r) = ((r & 3) << 2) |
// ((r >> 2) & 3): the 16 rows of a ds_read_b128 group (same chunk) spread over all 16 positions, and the 4 rows of a transpose
// block (4 adjacent chunks) over the four 64-B quarters -- both conflict free.  A DMA instruction moves 4 rows (8 full lines).
__device__ __attribute__((aligned(256))) bf16_t ab2_zero_page[128];
__device__ float ab2_inf_zero[2] = {INFINITY, 0.f};      // the statistics of a query beyond Nq: lse = +inf (P = 0), delta = 0
typedef __attribute__((address_space(3))) void ab2_lds_void_t;
typedef const __attribute__((address_space(1))) void ab2_gbl_void_t;
typedef __attribute__((ext_vector_type(4))) short ab2_v4s;
typedef __attribute__((address_space(3))) ab2_v4s ab2_lds_v4s_t;
constexpr int AB2_ROW = 256, AB2_MAT = 64 * AB2_ROW, AB2_STAGE = 2 * AB2_MAT;
NS2_DEVINL int ab2_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

template <int ROLE>
__global__ __launch_bounds__(256, 2) void attn_bwd2_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_stat = reinterpret_cast<float*>(smem + 2 * AB2_STAGE);       // [stage][lse 64 | delta 64] (role 1)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int Nown = ROLE == 0 ? a.Nq : a.Nk, Nwalk = ROLE == 0 ? a.Nk : a.Nq;
  const int nown_t = (Nown + 127) / 128;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ot = bid % nown_t;
  bid /= nown_t;
  const int h = bid % a.H, b = bid / a.H;
  const int orow = ot * 128 + wave * 32 + l31;
  const bool own_ok = orow < Nown;

  // own-side fragments (B operands): X = Q (role 0) / K (role 1); G = dO (role 0) / V (role 1)
  const bf16_t* xb = ROLE == 0 ? a.q_hi : a.k_hi;
  const int ldx = ROLE == 0 ? a.ldq : a.ldk, xcol = ROLE == 0 ? a.q_col0 : a.k_col0;
  const bf16_t* gb = ROLE == 0 ? a.do_hi : a.v_hi;
  const int ldg = ROLE == 0 ? a.lddo : a.ldv, gcol = ROLE == 0 ? 0 : a.v_col0;
  bf16x8 xf[2][4], gf[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint4 vx = make_uint4(0u, 0u, 0u, 0u), vg = make_uint4(0u, 0u, 0u, 0u);
      if (own_ok) {
        vx = *reinterpret_cast<const uint4*>(xb + ((long)b * Nown + orow) * 2L * ldx + pcol(xcol + h * 64 + 16 * c + 8 * hi, true) + 32 * p);
        vg = *reinterpret_cast<const uint4*>(gb + ((long)b * Nown + orow) * 2L * ldg + pcol(gcol + h * 64 + 16 * c + 8 * hi, true) + 32 * p);
      }
      xf[p][c] = *reinterpret_cast<bf16x8*>(&vx);
      gf[p][c] = *reinterpret_cast<bf16x8*>(&vg);
    }
  float lse_own = INFINITY, del_own = 0.f;
  if (ROLE == 0 && own_ok) {
    lse_own = a.lse[((long)b * a.H + h) * a.Nq + orow];
    del_own = a.delta[((long)b * a.H + h) * a.Nq + orow];
  }

  // walked-side sources (row-major): Y (scores; transposed for acc1), Yg (dP; role 1: transposed for acc2)
  const bf16_t* yb = ROLE == 0 ? a.k_hi : a.q_hi;
  const int ldy = ROLE == 0 ? a.ldk : a.ldq, ycol = ROLE == 0 ? a.k_col0 : a.q_col0;
  const bf16_t* ygb = ROLE == 0 ? a.v_hi : a.do_hi;
  const int ldyg = ROLE == 0 ? a.ldv : a.lddo, ygcol = ROLE == 0 ? a.v_col0 : 0;

  // ---- DMA pieces: instruction j = 8 wave + i of the tile's 32 (matrix j >> 4, tile rows 4 (j & 15) ...); the lane's row = .. + (lane >> 4),
  // stored position lane & 15, i.e. it fetches logical chunk (lane & 15) ^ f(row); f(row) = ((lane >> 4) << 2) | (i & 3)
  const int drow = lane >> 4, dpos = lane & 15;
  const bf16_t* dsrc[8];           // source of tile 0 (advanced by 64 rows per tile)
  int dtr[8];                      // tile row of the lane
  const long ystep = 64L * 2 * ldy, ygstep = 64L * 2 * ldyg;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = 8 * wave + i, mat = j >> 4;
    const int tr = 4 * (j & 15) + drow;
    const int lc = dpos ^ ((drow << 2) | (i & 3));
    dtr[i] = tr;
    dsrc[i] = mat == 0 ? yb + ((long)b * Nwalk + tr) * 2L * ldy + pcol(ycol + h * 64, true) + lc * 8
                       : ygb + ((long)b * Nwalk + tr) * 2L * ldyg + pcol(ygcol + h * 64, true) + lc * 8;
  }
  auto issue_tile = [&](int t, int stage) __attribute__((always_inline)) {
    const int r0 = t * 64;
    const bool full = r0 + 64 <= Nwalk;
    unsigned char* sb = smem + stage * AB2_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = 8 * wave + i;
      const bf16_t* p = dsrc[i] + (long)t * ((j >> 4) == 0 ? ystep : ygstep);
      if (!full && r0 + dtr[i] >= Nwalk) p = ab2_zero_page;                       // rows beyond Nwalk are zeros (as in the staged kernel)
      // As inline assembly: behind the builtin the waitcnt pass assumes every later LDS read may alias the DMA's destination and puts
      // `s_waitcnt vmcnt(0)` in front of the first read of the tile being multiplied -- the prefetch would be waited for at once.
      // The ordering that matters (this wave's vmcnt(0) + the barrier at the top of the next iteration) is explicit below.
      const unsigned dst = (unsigned)(size_t)(sb + (j >> 4) * AB2_MAT + (j & 15) * 1024);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
    }
    if constexpr (ROLE == 1) {
      // the 64 queries' lse (wave 0) and delta (wave 1) travel the same way, 4 bytes per lane: an ordinary load here would sit in the
      // same in-order queue BEHIND the pieces above, and the compiler's wait for it would wait for them too
      if (wave < 2) {
        const int q = r0 + lane;
        const float* p = q < a.Nq ? (wave == 0 ? a.lse : a.delta) + ((long)b * a.H + h) * a.Nq + q : ab2_inf_zero + wave;
        const unsigned dst = (unsigned)(size_t)(s_stat + stage * 128 + wave * 64);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(p), "s"(dst) : "memory", "m0");
      }
    }
  };

  // ---- fragment addressing
  const int pi_row = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int fy = ab2_f(pi_row);                    // (js * 32 does not touch bits 0-3)
  const int y_row_off = pi_row * AB2_ROW;
  // transpose reads: i16 = the lane's place in its 16-lane group, tg = which 16 d columns of the 32 (l31 = 16 tg + i16 = the d it receives).
  // read q (0, 1) of (js, g1): supplies row 32 js + 16 g1 + 8 hi + 4 q + (i16 >> 2), d columns 16 tg + 4 (i16 & 3) .. + 3 of plane p, 32-column half dt
  const int i16 = lane & 15, tg = (lane >> 4) & 1;
  int t_off[2][2][2];                              // [q][dt][p]: byte offset inside a matrix, rows 32 js + 16 g1 to be added
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * hi + 4 * q + (i16 >> 2);
    const int f = ab2_f(row);                      // (32 js + 16 g1 does not touch bits 0-3)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int lc = dt * 8 + p * 4 + 2 * tg + ((i16 >> 1) & 1);
        t_off[q][dt][p] = row * AB2_ROW + ((lc ^ f) << 4) + (i16 & 1) * 8;
      }
  }
  auto tfrag = [&](const unsigned char* mat, int js, int g1, int dt, int p) __attribute__((always_inline)) {
    const unsigned char* base = mat + (32 * js + 16 * g1) * AB2_ROW;
    const ab2_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab2_lds_v4s_t*)(base + t_off[0][dt][p]));
    const ab2_v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab2_lds_v4s_t*)(base + t_off[1][dt][p]));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  };

  f32x16 acc1[2], acc2[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[dt][r] = 0.f; acc2[dt][r] = 0.f; }
  const float sl2 = a.scale * 1.4426950408889634f;
  const int ntiles = (Nwalk + 63) / 64;

  // ---- prologue: tile 0 (and its per-query statistics, role 1)
  // The own-side fragments are complete BEFORE the loop, as far as the compiler can see: a wait of its own for one of these loads
  // inside the loop (it places them lazily, in front of the first use) would be `vmcnt(small)` -- and would drain the prefetched tile
  // in every iteration, since the DMA pieces are younger entries of the same in-order queue that it does not know about.
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("" :: "v"(xf[p][c]), "v"(gf[p][c]));
  issue_tile(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    const int r0 = t * 64, st = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed
    __syncthreads();                                       // ... everybody's have; everybody is done with tile t - 1 (stage st ^ 1 is free)
    if (t + 1 < ntiles) issue_tile(t + 1, st ^ 1);
    const unsigned char* my = smem + st * AB2_STAGE;      // Y
    const unsigned char* myg = my + AB2_MAT;               // Yg
    const float* c_lse = s_stat + st * 128;
    const float* c_del = c_lse + 64;

#pragma unroll
    for (int js = 0; js < 2; ++js) {
      f32x16 stt, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { stt[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 yf[2], ygf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int off = js * 32 * AB2_ROW + y_row_off + ((((c >> 1) * 8 + p * 4 + 2 * (c & 1) + hi) ^ fy) << 4);
          yf[p] = *reinterpret_cast<const bf16x8*>(my + off);
          ygf[p] = *reinterpret_cast<const bf16x8*>(myg + off);
        }
        stt = mma16<false>(yf[1], xf[0][c], stt);
        stt = mma16<false>(yf[0], xf[1][c], stt);
        stt = mma16<false>(yf[0], xf[0][c], stt);
        dp = mma16<false>(ygf[1], gf[0][c], dp);
        dp = mma16<false>(ygf[0], gf[1][c], dp);
        dp = mma16<false>(ygf[0], gf[0][c], dp);
      }
      // ---- P and dS for (own row = lane, walked row = register): register r <-> walked row r0 + 32 js + 16 (r >> 3) + 8 hi + (r & 7)
      float pv[16], dsv[16];
#pragma unroll
      for (int g1 = 0; g1 < 2; ++g1) {
        float ls[8], dl[8];
        if constexpr (ROLE == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { ls[e] = lse_own; dl[e] = del_own; }
        } else {
          const int w0 = 32 * js + 16 * g1 + 8 * hi;
          const float4 l0 = *reinterpret_cast<const float4*>(c_lse + w0), l1 = *reinterpret_cast<const float4*>(c_lse + w0 + 4);
          const float4 d0 = *reinterpret_cast<const float4*>(c_del + w0), d1 = *reinterpret_cast<const float4*>(c_del + w0 + 4);
          ls[0] = l0.x; ls[1] = l0.y; ls[2] = l0.z; ls[3] = l0.w; ls[4] = l1.x; ls[5] = l1.y; ls[6] = l1.z; ls[7] = l1.w;
          dl[0] = d0.x; dl[1] = d0.y; dl[2] = d0.z; dl[3] = d0.w; dl[4] = d1.x; dl[5] = d1.y; dl[6] = d1.z; dl[7] = d1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * g1 + e;
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(stt[r], sl2, -ls[e]));
          if (ROLE == 0 && r0 + 32 * js + 16 * g1 + 8 * hi + e >= Nwalk) p = 0.f;      // keys beyond Nk (role 1: lse = +inf did it)
          pv[r] = p;
          dsv[r] = p * (dp[r] - dl[e]);
        }
      }
      // ---- accumulate: acc1 += Y^T dS^T (dQ^T or dK^T), acc2 += Yg^T P^T (dV^T, role 1); the transposed fragments come out of the row-major tiles
#pragma unroll
      for (int g1 = 0; g1 < 2; ++g1) {
        bf16x8 dsf[2], pf[2];
        {
          uint32_t ph[4], pl[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2(dsv[8 * g1 + 2 * e], dsv[8 * g1 + 2 * e + 1], ph[e], pl[e]);
          const uint4 uh = make_uint4(ph[0], ph[1], ph[2], ph[3]), ul = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          dsf[0] = *reinterpret_cast<const bf16x8*>(&uh);
          dsf[1] = *reinterpret_cast<const bf16x8*>(&ul);
        }
        if constexpr (ROLE == 1) {
          uint32_t ph[4], pl[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2(pv[8 * g1 + 2 * e], pv[8 * g1 + 2 * e + 1], ph[e], pl[e]);
          const uint4 uh = make_uint4(ph[0], ph[1], ph[2], ph[3]), ul = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          pf[0] = *reinterpret_cast<const bf16x8*>(&uh);
          pf[1] = *reinterpret_cast<const bf16x8*>(&ul);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 t1h = tfrag(my, js, g1, dt, 0), t1l = tfrag(my, js, g1, dt, 1);
          acc1[dt] = mma16<false>(t1l, dsf[0], acc1[dt]);
          acc1[dt] = mma16<false>(t1h, dsf[1], acc1[dt]);
          acc1[dt] = mma16<false>(t1h, dsf[0], acc1[dt]);
          if constexpr (ROLE == 1) {
            const bf16x8 t2h = tfrag(myg, js, g1, dt, 0), t2l = tfrag(myg, js, g1, dt, 1);
            acc2[dt] = mma16<false>(t2l, pf[0], acc2[dt]);
            acc2[dt] = mma16<false>(t2h, pf[1], acc2[dt]);
            acc2[dt] = mma16<false>(t2h, pf[0], acc2[dt]);
          }
        }
      }
    }
  }

  // ---- store: lane holds d = 32 dt + 8 gq + 4 hi + e of its own row
  if (!own_ok) return;
  if (ROLE == 0 ? a.gp_q : a.gp_kv) {
    // operand planes: the lane's 4-column groups of its own row (lanes l31 and l31 + 32 interleave the groups of a 32-column line)
    bf16_t* prow = a.gp_hi + ((long)b * Nown + orow) * 2L * a.gp_ld;
    const int c1 = (ROLE == 0 ? a.dq_col0 : a.dk_col0) + h * 64, c2 = a.dv_col0 + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int dcol = 32 * dt + 8 * gq + 4 * hi;
        store_cols4(prow, c1 + dcol, acc1[dt][4 * gq] * a.scale, acc1[dt][4 * gq + 1] * a.scale, acc1[dt][4 * gq + 2] * a.scale,
                    acc1[dt][4 * gq + 3] * a.scale, a.gp_fmt, true);
        if constexpr (ROLE == 1)
          store_cols4(prow, c2 + dcol, acc2[dt][4 * gq], acc2[dt][4 * gq + 1], acc2[dt][4 * gq + 2], acc2[dt][4 * gq + 3], a.gp_fmt, true);
      }
    return;
  }
  float* o1 = ROLE == 0 ? a.dq + ((long)b * a.Nq + orow) * a.lddq + a.dq_col0 + h * 64
                        : a.dk + ((long)b * a.Nk + orow) * a.lddk + a.dk_col0 + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int dcol = 32 * dt + 8 * gq + 4 * hi;
      *reinterpret_cast<float4*>(o1 + dcol) = make_float4(acc1[dt][4 * gq] * a.scale, acc1[dt][4 * gq + 1] * a.scale,
                                                          acc1[dt][4 * gq + 2] * a.scale, acc1[dt][4 * gq + 3] * a.scale);
      if constexpr (ROLE == 1) {
        float* o2 = a.dv + ((long)b * a.Nk + orow) * a.lddv + a.dv_col0 + h * 64;
        *reinterpret_cast<float4*>(o2 + dcol) = make_float4(acc2[dt][4 * gq], acc2[dt][4 * gq + 1], acc2[dt][4 * gq + 2], acc2[dt][4 * gq + 3]);
      }
    }
}

template <int ROLE>
static hipError_t launch_attn_bwd2_role(const AttnBwdArgs& a, hipStream_t s) {
  const size_t lds = 2 * AB2_STAGE + 2 * 128 * sizeof(float);
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(reinterpret_cast<const void*>(&attn_bwd2_kernel<ROLE>), (int)lds);
  if (e != hipSuccess) return e;
  const int nown = ROLE == 0 ? a.Nq : a.Nk;
  dim3 grid(((nown + 127) / 128) * a.H * a.B);
  hipLaunchKernelGGL((attn_bwd2_kernel<ROLE>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.H <= 0 || a.Nq <= 0 || a.Nk <= 0 || !a.lse || !a.delta) return hipErrorInvalidValue;
  auto il = [](const bf16_t* hi_, const bf16_t* lo_) { return hi_ && lo_ == hi_ + 32; };
  if (!il(a.q_hi, a.q_lo) || !il(a.k_hi, a.k_lo) || !il(a.v_hi, a.v_lo) || !il(a.do_hi, a.do_lo)) return hipErrorInvalidValue;
  if (((a.ldq | a.ldk | a.ldv | a.lddo) & 31) || ((a.q_col0 | a.k_col0 | a.v_col0) & 31)) return hipErrorInvalidValue;
  const bool want_q = a.dq != nullptr || a.gp_q, want_kv = a.dk != nullptr || a.dv != nullptr || a.gp_kv;
  if (!want_q && !want_kv) return hipErrorInvalidValue;
  if (a.gp_q || a.gp_kv) {        // planes: interleaved lines, 32-column aligned head blocks, one of the two line formats
    if (!a.gp_hi || a.gp_lo != a.gp_hi + 32 || (a.gp_ld & 31) || (a.gp_fmt != FMT_BF16 && a.gp_fmt != FMT_H8)) return hipErrorInvalidValue;
    if ((a.gp_q && (a.dq_col0 & 31)) || (a.gp_kv && ((a.dk_col0 | a.dv_col0) & 31))) return hipErrorInvalidValue;
  }
  if (want_kv && !a.gp_kv && (!a.dk || !a.dv || ((a.lddk | a.lddv | a.dk_col0 | a.dv_col0) & 3))) return hipErrorInvalidValue;
  if (want_q && !a.gp_q && ((a.lddq & 3) || (a.dq_col0 & 3))) return hipErrorInvalidValue;
  if (want_q) { hipError_t e = launch_attn_bwd2_role<0>(a, s); if (e != hipSuccess) return e; }
  if (want_kv) { hipError_t e = launch_attn_bwd2_role<1>(a, s); if (e != hipSuccess) return e; }
  return hipSuccess;
}

NS2_DEFINE_SATURATION_READER(backward)

}  // namespace ns2
