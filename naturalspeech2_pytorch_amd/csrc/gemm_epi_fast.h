// Fast-path epilogues of the 256x256 kernel (gemm2.hip) for INTERIOR wave tiles (128 rows x 64 columns, all valid).
//
// Why: the generic epilogues of gemm_epi.h test `row < M && col < N` per store, pick the plane format per store at run time,
// bump the saturation counter with one branch + atomic per converted pair, exchange lanes through ds_bpermute and store
// 2-4 bytes per lane.  The block timeline (tools/trace_blocks.py, profiles/r03_block_timeline_*.txt) showed every epilogue
// type at 15-19 us per block whatever it wrote (64 KiB ... 256 KiB): ~20k mostly-scalar instructions with ~600 branches --
// as long as the whole K loop of a K = 512 GEMM in the single-product modes.  Here:
//   * the plane format is a template parameter, bounds are checked once per wave (edge tiles keep the generic path);
//   * the range guard of the IEEE-half formats is a running max + NaN flag per lane, ONE atomic per lane at the end;
//   * adjacent columns are paired with a DPP quad permute (no LDS crossbar round trip);
//   * the tile is transposed through the wave's private LDS region (free after the K loop) and leaves as 16-byte stores:
//     one wave-instruction writes whole 128-B / 256-B row segments (the vector-memory path costs ~40-60 cycles per
//     wave-instruction whatever its width; 4-byte stores were 4-12x as many instructions);
//   * V^T (EPI_QKV) is transposed in LDS too: 256 contiguous bytes per feature row instead of 8-byte scattered stores;
//   * the FiLM gate between the Wavenet block's K phases reads gamma / beta once per column (wavenet_midgate's `uni`, gemm_epi.h);
//   * GEGLU evaluates erfc with a branch-free rational-exponential form (relative error 1.2e-7 everywhere) instead of the
//     device library's two-branch erff (both branches execute in a divergent wave).
// Results are bit-identical to the generic path for the plane formats; GEGLU differs by <= ~2e-7 relative (gelu's erf).
#pragma once
#include "gemm_epi.h"

namespace ns2 {

enum FastPlaneFmt : int { PF_F16 = 0, PF_BF16IL = 1, PF_H8 = 2, PF_BF16 = 3 };
template <int PF> struct PlaneGeom {
  static constexpr bool il = (PF == PF_BF16IL || PF == PF_H8);     // interleaved 128-B lines of 32 logical columns
  static constexpr int bytes_per_col32 = il ? 128 : 64;            // global bytes of 32 logical columns of one row
  static constexpr float limit = (PF == PF_H8) ? H8_MAX : 65504.f;
  static constexpr bool guarded = (PF == PF_F16 || PF == PF_H8);   // formats with the IEEE-half range
};

// value of lane ^ 1 (DPP quad_perm [1,0,3,2]): one VALU move, no LDS crossbar
NS2_DEVINL float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// range guard of the IEEE-half formats (ns2_common.h): a running max and a NaN flag per lane, one atomic at the end
struct RangeTrack {
  float mx = 0.f;
  bool nan = false;
  NS2_DEVINL void see(float a, float b) {
    mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
    nan = nan || (a != a) || (b != b);
  }
  NS2_DEVINL void flush(float limit) {
    if (nan || !(mx <= limit)) atomicAdd(&ns2_sat_counter, 1u);
  }
};

NS2_DEVINL uint32_t cvt2h_q(float a, float b) {            // cvt2h without the range note (the caller tracks it)
  f32x2_t v = {fminf(fmaxf(a, -65504.f), 65504.f), fminf(fmaxf(b, -65504.f), 65504.f)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
NS2_DEVINL void cvt2_h8_q(float a, float b, uint32_t& h16, uint32_t& h8, uint32_t& l8) {   // cvt2_h8 without the range note
  a = fminf(fmaxf(a, -H8_MAX), H8_MAX);
  b = fminf(fmaxf(b, -H8_MAX), H8_MAX);
  f32x2_t v = {a, b};
  f16x2_t h = __builtin_convertvector(v, f16x2_t);
  f32x2_t r = (v - __builtin_convertvector(h, f32x2_t)) * H8_LO_SCALE;
  h16 = __builtin_bit_cast(uint32_t, h);
  h8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
  l8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(r.x, -H8_MAX), H8_MAX), fminf(fmaxf(r.y, -H8_MAX), H8_MAX), 0, false) & 0xffffu;
}

// two adjacent logical columns (c even, 0 <= c < 64 inside the wave tile) of one staged LDS row, in global byte order
template <int PF>
NS2_DEVINL void lds_put2(unsigned char* rowp, int c, float v0, float v1, RangeTrack& rt) {
  if constexpr (PF == PF_F16) {
    rt.see(v0, v1);
    *reinterpret_cast<uint32_t*>(rowp + c * 2) = cvt2h_q(v0, v1);
  } else if constexpr (PF == PF_BF16) {
    *reinterpret_cast<uint32_t*>(rowp + c * 2) = cvt2(v0, v1);
  } else if constexpr (PF == PF_BF16IL) {
    uint32_t ph, pl;
    split2(v0, v1, ph, pl);
    unsigned char* line = rowp + (c >> 5) * 128 + (c & 31) * 2;
    *reinterpret_cast<uint32_t*>(line) = ph;
    *reinterpret_cast<uint32_t*>(line + 64) = pl;
  } else {
    rt.see(v0, v1);
    uint32_t h16, h8, l8;
    cvt2_h8_q(v0, v1, h16, h8, l8);
    unsigned char* line = rowp + (c >> 5) * 128;
    *reinterpret_cast<uint32_t*>(line + (c & 31) * 2) = h16;
    *reinterpret_cast<uint16_t*>(line + 64 + (c & 31)) = (uint16_t)h8;
    *reinterpret_cast<uint16_t*>(line + 96 + (c & 31)) = (uint16_t)l8;
  }
}

// staged rows -> global: ROWS rows of ROWB bytes (LDS row stride ROWB + 16), 16 bytes per lane, whole row segments per instruction
template <int ROWB, int ROWS>
NS2_DEVINL void lds_flush_rows(const unsigned char* wbuf, unsigned char* gbase, long row_stride_bytes, int lane) {
  constexpr int LPR = ROWB / 16, RPI = 64 / LPR, RS = ROWB + 16;
  static_assert(ROWS % RPI == 0, "row count must be a multiple of the rows per store instruction");
  const int lr0 = lane / LPR, ch = lane % LPR;
#pragma unroll
  for (int it = 0; it < ROWS / RPI; ++it) {
    const int lr = it * RPI + lr0;
    const uint4 v = *reinterpret_cast<const uint4*>(wbuf + lr * RS + ch * 16);
    *reinterpret_cast<uint4*>(gbase + (long)lr * row_stride_bytes + ch * 16) = v;
  }
}

// ---- split planes: EPI_SPLIT (bias; calls with an activation keep the generic path), EPI_WAVENET (biases were applied mid-loop), the q / k part of EPI_QKV
// MI = 32-row accumulator tiles of the wave (4: gemm2.hip's 128 x 64 wave tile, 2: gemm.hip's 64 x 64), WBUF = bytes of the wave's
// private LDS region: the tile leaves in passes of the most rows (a power of two, at least 32) that fit it.
template <int PF, bool BIAS_ACT, int MI = 4, int WBUF = 18432>
NS2_DEVINL void epi_planes_fast(f32x16 (&acc)[MI][2], const GemmArgs& g, int z, int row_base, int col_base, int lane, unsigned char* wbuf) {
  using G = PlaneGeom<PF>;
  constexpr int ROWB = 2 * G::bytes_per_col32;            // 64 columns of one row
  constexpr int RS = ROWB + 16;
  constexpr int RPP = (32 * MI * RS <= WBUF) ? 32 * MI : ((16 * MI * RS <= WBUF) ? 16 * MI : 32);   // rows per pass
  static_assert(RPP * RS <= WBUF && RPP >= 32 && (32 * MI) % RPP == 0, "the wave's LDS region holds at least one 32-row pass");
  const int l31 = lane & 31, hi = lane >> 5;
  const bool odd = lane & 1;
  float bc[2] = {0.f, 0.f};
  if constexpr (BIAS_ACT) {
    if (g.bias) {
      const float* bias = g.bias + (long)z * g.bias_zs;
      bc[0] = bias[col_base + l31];
      bc[1] = bias[col_base + 32 + l31];
    }
  }
  const long rsb = pld(g.ldo_s, G::il) * 2;
  unsigned char* gbase = reinterpret_cast<unsigned char*>(g.out_hi + pcol((int)(z * g.out_zs), G::il)) + (long)row_base * rsb +
                         (long)(col_base >> 5) * G::bytes_per_col32;
  RangeTrack rt;
#pragma unroll
  for (int pass = 0; pass < 32 * MI / RPP; ++pass) {
#pragma unroll
    for (int mh = 0; mh < RPP / 32; ++mh) {
      const int mi = pass * (RPP / 32) + mh;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
          const float v0 = acc[mi][ni][2 * rp] + bc[ni], v1 = acc[mi][ni][2 * rp + 1] + bc[ni];
          // columns (l31 & ~1, l31 | 1): the even lane keeps row 2rp, the odd lane row 2rp + 1
          const float recv = lane_xor1(odd ? v0 : v1);
          const float c_lo = odd ? recv : v0, c_hi = odd ? v1 : recv;
          const int r = 2 * rp + (odd ? 1 : 0);
          const int lr = mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          lds_put2<PF>(wbuf + lr * RS, ni * 32 + (l31 & ~1), c_lo, c_hi, rt);
          if ((rp & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the live set small: this code runs at the VGPR cap
        }
    }
    __builtin_amdgcn_wave_barrier();
    lds_flush_rows<ROWB, RPP>(wbuf, gbase + (long)pass * RPP * rsb, rsb, lane);
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (G::guarded) rt.flush(G::limit);
}

// ---- V^T of EPI_QKV: the wave tile's 64 value features x 128 tokens, transposed in LDS, 256 contiguous bytes per feature row.
// Dense 16-bit formats only (F16: IEEE half with the range guard, else bf16); needs seq_len % 128 == 0 (one utterance per tile).
// (MI = 32-token accumulator tiles of the wave: 4 in gemm2.hip, needs seq_len % 128 == 0; 2 in gemm.hip, seq_len % 64 == 0)
template <bool F16, int MI = 4>
NS2_DEVINL void epi_vt_fast(f32x16 (&acc)[MI][2], const GemmArgs& g, int row_base, int col_base, int lane, unsigned char* wbuf) {
  constexpr int TOKB = 64 * MI;                      // bytes of the wave tile's 32 * MI tokens in one feature row
  constexpr int RS = TOKB + 16;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = row_base / g.seq_len, n0 = row_base - b * g.seq_len;
  const int feat0 = col_base - g.split_col;
  RangeTrack rt;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float a0 = acc[mi][ni][4 * gq + 0], a1 = acc[mi][ni][4 * gq + 1], a2 = acc[mi][ni][4 * gq + 2], a3 = acc[mi][ni][4 * gq + 3];
        uint32_t p01, p23;
        if constexpr (F16) { rt.see(a0, a1); rt.see(a2, a3); p01 = cvt2h_q(a0, a1); p23 = cvt2h_q(a2, a3); }
        else { p01 = cvt2(a0, a1); p23 = cvt2(a2, a3); }
        const int tok = mi * 32 + 8 * gq + 4 * hi;
        *reinterpret_cast<uint2*>(wbuf + (ni * 32 + l31) * RS + tok * 2) = make_uint2(p01, p23);
        if (gq == 3) __builtin_amdgcn_sched_barrier(0);
      }
  __builtin_amdgcn_wave_barrier();
  unsigned char* gbase = reinterpret_cast<unsigned char*>(g.vt_hi + ((long)b * g.vt_rows + feat0) * g.vt_ld + n0);
  lds_flush_rows<TOKB, 64>(wbuf, gbase, (long)g.vt_ld * 2, lane);
  __builtin_amdgcn_wave_barrier();
  if constexpr (F16) rt.flush(65504.f);
}

// ---- GEGLU: wave tile = [x(32 cols) | gate(32 cols)] -> 32 output columns gelu(gate) * x
template <int PF, int MI = 4>
NS2_DEVINL void epi_geglu_fast(f32x16 (&acc)[MI][2], const GemmArgs& g, int row_base, int col_base, int ocol_base, int lane, unsigned char* wbuf) {
  using G = PlaneGeom<PF>;
  constexpr int ROWB = G::bytes_per_col32;
  constexpr int RS = ROWB + 16;
  const int l31 = lane & 31, hi = lane >> 5;
  const bool odd = lane & 1;
  const float bx = g.bias[col_base + l31], bg = g.bias[col_base + 32 + l31];      // packed (padded) bias: always in range
  const long rsb = pld(g.ldo_s, G::il) * 2;
  unsigned char* gbase = reinterpret_cast<unsigned char*>(g.out_hi) + (long)row_base * rsb + (long)(ocol_base >> 5) * G::bytes_per_col32;
  RangeTrack rt;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
      float v0 = gelu_erf(acc[mi][1][2 * rp] + bg) * (acc[mi][0][2 * rp] + bx);
      float v1 = gelu_erf(acc[mi][1][2 * rp + 1] + bg) * (acc[mi][0][2 * rp + 1] + bx);
      asm volatile("" : "+v"(v0), "+v"(v1));      // finish this pair before the next one starts (see wavenet_midgate_fast)
      const float recv = lane_xor1(odd ? v0 : v1);
      const float c_lo = odd ? recv : v0, c_hi = odd ? v1 : recv;
      const int r = 2 * rp + (odd ? 1 : 0);
      const int lr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      lds_put2<PF>(wbuf + lr * RS, l31 & ~1, c_lo, c_hi, rt);
      if ((rp & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  __builtin_amdgcn_wave_barrier();
  lds_flush_rows<ROWB, 32 * MI>(wbuf, gbase, rsb, lane);
  __builtin_amdgcn_wave_barrier();
  if constexpr (G::guarded) rt.flush(G::limit);
}

// ---- fp32 (+ bias, residual; calls with an activation keep the generic path): two passes of 64 rows x 64 columns through LDS, float4 loads / stores
NS2_DEVINL void epi_f32_fast(f32x16 (&acc)[4][2], const GemmArgs& g, int z, int row_base, int col_base, int lane, unsigned char* wbuf) {
  constexpr int RS = 272;
  const int l31 = lane & 31, hi = lane >> 5;
  const float bc0 = g.bias ? g.bias[col_base + l31] : 0.f, bc1 = g.bias ? g.bias[col_base + 32 + l31] : 0.f;
  const int lr0 = lane >> 4, ch = lane & 15;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    // issue the residual loads of this pass first: they fly while the tile is staged
    float4 rr[16];
    if (g.resid) {
      const float* rbase = g.resid + (long)(row_base + pass * 64 + lr0) * g.ldr + col_base + ch * 4;
#pragma unroll
      for (int it = 0; it < 16; ++it) rr[it] = *reinterpret_cast<const float4*>(rbase + (long)it * 4 * g.ldr);
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          *reinterpret_cast<float*>(wbuf + lr * RS + (ni * 32 + l31) * 4) = acc[pass * 2 + mh][ni][r] + (ni ? bc1 : bc0);
        }
    __builtin_amdgcn_wave_barrier();
    float* obase = g.out_f + z * g.out_f_zs + (long)(row_base + pass * 64 + lr0) * g.ldo_f + col_base + ch * 4;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      float4 v = *reinterpret_cast<const float4*>(wbuf + (it * 4 + lr0) * RS + ch * 16);
      if (g.resid) { v.x += rr[it].x; v.y += rr[it].y; v.z += rr[it].z; v.w += rr[it].w; }
      *reinterpret_cast<float4*>(obase + (long)it * 4 * g.ldo_f) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace ns2
