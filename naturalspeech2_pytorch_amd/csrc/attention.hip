// Flash-attention forward for the NaturalSpeech2 denoiser on gfx950 (CDNA4): softmax(q k^T / sqrt(64)) v,
// non-causal, no dropout, optional key-length tail (ATT:77-155; hot path = Attend.flash_attn / math path).
// Covers self-attention (Nk = N), prompt cross-attention (Nk = 32 perceiver latents) and the
// PerceiverResampler's own attention (Nq = 32, Nk = 32 + n_prompt).
//
// Design (64-wide waves, MFMA 32x32x16 bf16, split-plane operands like the GEMMs):
//   * one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows;
//   * K tile [64 keys][64 d] and V^T tile [64 d][64 keys] are staged global -> VGPR -> LDS (double-buffered,
//     next tile's loads in flight during the current tile's MFMAs), rows padded to 144 B (9 x 16 B slots,
//     9 coprime to 16 => conflict-free ds_read_b128 fragments);
//   * S^T = K Q^T ("swapped" product): the C fragment then holds, per lane, 16 keys of ONE query
//     (query = lane & 31), so the online softmax runs in registers with a single cross-half exchange;
//     key rows are read through the bit-swap permutation pi (bits 2<->3) so that each half-wave's
//     registers r = 8*g1 .. 8*g1+7 are 8 CONSECUTIVE keys: P feeds the second MFMA's B operand directly
//     and the matching V^T fragment is one 16-B LDS read;
//   * O^T = V^T P^T accumulates with query = lane & 31 again, so the running rescale is a per-lane scalar;
//   * V arrives already transposed ([b][h*64+d][n]) from the QKV GEMM epilogue (gemm.hip EPI_QKV).
// NSPLIT = 3 evaluates both products as hi*hi + hi*lo + lo*hi (fp32-class accuracy), NSPLIT = 1 hi only.
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

constexpr int AT_ROWB = 144;               // V^T tile row: 64 keys = 128 B payload + 16 B pad
// head dimension D (32 / 64 / 128; round 6): K tile rows hold D dims = 2 D bytes + 16 B pad (5 / 9 / 17 sixteen-byte slots: coprime to 16,
// conflict-free ds_read_b128 fragments), the V^T tile has D rows
template <int D> struct AtGeom {
  static constexpr int ROWB_K = 2 * D + 16;
  static constexpr int KPLANE = 64 * ROWB_K;
  static constexpr int VPLANE = D * AT_ROWB;
};

NS2_DEVINL uint4 ld16g(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// zero the bf16 elements e >= nvalid of an 8-element chunk
NS2_DEVINL uint4 mask_chunk(uint4 v, int nvalid) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (2 * i >= nvalid) w[i] = 0u;
    else if (2 * i + 1 >= nvalid) w[i] &= 0xffffu;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// NW waves per workgroup = 32 NW query rows share every staged K / V^T tile.  NW = 8 (256-query workgroups, half the K / V^T
// re-reads) was built and measured neutral (0.1434 vs 0.1439 ms): the re-reads are not what bounds this kernel; NW = 4 is used.
// Occupancy: the single-product kernel compiled to 169 VGPRs -- one register over the 168 that let three waves share a SIMD.  Asked
// for three (one 4-byte spill outside the loop) it runs 9.5 % faster at the headline shape (0.1439 -> 0.1302 ms, 478 -> 528 TF):
// the loop is VALU-bound (softmax) and lock-stepped by one barrier per tile, so a third wave per SIMD is what overlaps one
// wave's exp / max / convert work with another's MFMAs.  Four waves (<= 128 VGPRs) spills 60+ registers.
// WLSE: also write the log-sum-exp the backward kernels recompute P from (training); a separate instantiation, so the inference
// kernels keep their register allocation (three waves per SIMD is a one-register margin, see above)
template <int NSPLIT, bool F16, int NW, bool WLSE, int D = 64>
__global__ __launch_bounds__(64 * NW, ((NSPLIT == 3 && D == 128) ? 1 : (NSPLIT == 3 || D == 128) ? 2 : 3)) void attn_kernel(const AttnArgs a) {
  using G = AtGeom<D>;
  constexpr int DC = D / 16;                         // 16-deep k chunks of the q . k contraction
  constexpr int DT = D / 32;                         // 32-row output tiles of O^T
  constexpr int NP = (NSPLIT == 3) ? 2 : 1;
  constexpr int QB = 32 * NW;                        // query rows per workgroup
  constexpr int NT = 64 * NW;                        // threads
  constexpr int CPT = 8 * D / NT;                    // 16-B chunks of one plane per thread: the K tile has 64 x D / 8, the V^T tile D x 8
  static_assert(CPT >= 1 && CPT * NT == 8 * D, "whole chunks per thread");
  constexpr int STAGE_BYTES = NP * (G::KPLANE + G::VPLANE);     // K planes then V^T planes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // 1-D grid, XCD-aware: the query tiles of one (batch, head) get consecutive remapped ids and therefore share one
  // XCD's L2 for their K / V^T re-reads (rocprofv3 FETCH_SIZE showed ~3x over-fetch with the default round-robin).
  const int nqt = (a.Nq + QB - 1) / QB;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % nqt;
  bid /= nqt;
  const int h = bid % a.H, b = bid / a.H;
  const int qrow = qt * QB + wave * 32 + l31;
  const bool q_ok = qrow < a.Nq;

  const bf16_t* q_pl[2] = {a.q_hi, a.q_lo};
  const bf16_t* k_pl[2] = {a.k_hi, a.k_lo};
  const bf16_t* v_pl[2] = {a.vt_hi, a.vt_lo};
  // operands with a lo plane are interleaved [hi32|lo32] per 32 columns (ns2_common.h); ld* are logical
  const bool qil = a.q_lo != nullptr, kil = a.k_lo != nullptr, vil = a.vt_lo != nullptr, oil = a.o_lo != nullptr;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = l31][d = 16c + 8hi .. +7]
  bf16x8 qf[NP][DC];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (q_ok) v = ld16g(q_pl[p] + ((long)b * a.Nq + qrow) * pld(a.ldq, qil) + pcol(a.q_col0 + h * D + 16 * c + 8 * hi, qil));
      qf[p][c] = *reinterpret_cast<bf16x8*>(&v);
    }

  // ---- staging coordinates: CPT chunks per plane per thread
  int srow[CPT], sch[CPT];          // V^T tile: feature row / 8-key chunk
  int krow[CPT], kch[CPT];          // K tile: key row / 8-dim chunk
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + NT * i;
    srow[i] = c >> 3;
    sch[i] = c & 7;
    krow[i] = c / (D / 8);
    kch[i] = c % (D / 8);
  }
  struct Regs { uint4 k[NP][CPT]; uint4 v[NP][CPT]; };

  auto load_tile = [&](Regs& rg, int key0) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int key = key0 + krow[i];
      const bool kok = key < a.Nk;
      const long koff = ((long)b * a.Nk + key) * pld(a.ldk, kil) + pcol(a.k_col0 + h * D + kch[i] * 8, kil);
      const int vkey = key0 + sch[i] * 8;
      const int nvalid = a.Nk - vkey;
      const long voff = ((long)b * a.H * D + h * D + srow[i]) * pld(a.vt_ld, vil) + pcol(vkey, vil);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        rg.k[p][i] = kok ? ld16g(k_pl[p] + koff) : make_uint4(0u, 0u, 0u, 0u);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (nvalid > 0) {
          v = ld16g(v_pl[p] + voff);                  // vt_ld is a multiple of 8: the chunk is inside the row
          if (nvalid < 8) v = mask_chunk(v, nvalid);  // finite zeros where P is exactly 0
        }
        rg.v[p][i] = v;
      }
    }
  };
  auto store_tile = [&](const Regs& rg, int s) {
    unsigned char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int offk = krow[i] * G::ROWB_K + kch[i] * 16, offv = srow[i] * AT_ROWB + sch[i] * 16;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        *reinterpret_cast<uint4*>(base + p * G::KPLANE + offk) = rg.k[p][i];
        *reinterpret_cast<uint4*>(base + NP * G::KPLANE + p * G::VPLANE + offv) = rg.v[p][i];
      }
    }
  };

  // pi: swap bits 2 and 3 of the MFMA row index -> key row inside a 32-key sub-tile
  const int pi_row = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int k_frag_off = pi_row * G::ROWB_K + hi * 16;        // + js*32*ROWB_K + c*32
  const int v_frag_off = l31 * AT_ROWB + hi * 16;             // + dt*32*ROWB + js*64 + g1*32

  f32x16 ot[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = a.scale * 1.4426950408889634f;

  const int ntiles = (a.Nk + 63) / 64;
  Regs rg;
  load_tile(rg, 0);
  store_tile(rg, 0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) load_tile(rg, (t + 1) * 64);
    const unsigned char* sb = smem + (t & 1) * STAGE_BYTES;
    const int key0 = t * 64;

    // ---- S^T = K Q^T  (2 sub-tiles of 32 keys)
    f32x16 st[2];
#pragma unroll
    for (int js = 0; js < 2; ++js) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[js][r] = 0.f;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        bf16x8 kf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
          kf[p] = *reinterpret_cast<const bf16x8*>(sb + p * G::KPLANE + k_frag_off + js * 32 * G::ROWB_K + c * 32);
        if constexpr (NSPLIT == 3) {
          st[js] = mma16<F16>(kf[1], qf[0][c], st[js]);
          st[js] = mma16<F16>(kf[0], qf[1][c], st[js]);
        }
        st[js] = mma16<F16>(kf[0], qf[0][c], st[js]);
      }
    }

    // ---- online softmax for query l31; register r of sub-tile js is key key0 + 32js + 16(r>>3) + 8hi + (r&7).
    // st holds RAW scores; the scale (and log2 e) is folded into the exponent's fma.  Key masking is a wave-uniform
    // slow path: only the last tile of a ragged key length, or a call with a key-padding mask, takes it.
    const unsigned char* km = a.kmask ? a.kmask + (long)b * a.Nk : nullptr;
    if (key0 + 64 > a.Nk || km) {
#pragma unroll
      for (int js = 0; js < 2; ++js)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + 32 * js + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= a.Nk) st[js][r] = -INFINITY;
          else if (km && !km[key]) st[js][r] = -3.0e38f;   // ATT:136-138 masked_fill(~mask, -finfo.max): finite, like the reference
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int js = 0; js < 2; ++js)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[js][r]), st[js][r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sl2);           // sl2 > 0: the scaled maximum is the maximum of the scaled scores
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int js = 0; js < 2; ++js)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[js][r], sl2, -m_new));
        st[js][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {                           // the running maximum rarely moves after the first tiles
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int js = 0; js < 2; ++js)
#pragma unroll
      for (int g1 = 0; g1 < 2; ++g1) {
        bf16x8 pf[NP];
        {
          uint32_t ph[4], pl[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (F16) { ph[e] = cvt2h_inrange(st[js][8 * g1 + 2 * e], st[js][8 * g1 + 2 * e + 1]); pl[e] = 0u; }   // p in [0, 1]
            else split2(st[js][8 * g1 + 2 * e], st[js][8 * g1 + 2 * e + 1], ph[e], pl[e]);
          }
          const uint4 uh = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          pf[0] = *reinterpret_cast<const bf16x8*>(&uh);
          if constexpr (NSPLIT == 3) {
            const uint4 ul = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            pf[1] = *reinterpret_cast<const bf16x8*>(&ul);
          }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          bf16x8 vf[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p)
            vf[p] = *reinterpret_cast<const bf16x8*>(sb + NP * G::KPLANE + p * G::VPLANE + v_frag_off + dt * 32 * AT_ROWB +
                                                    js * 64 + g1 * 32);
          if constexpr (NSPLIT == 3) {
            ot[dt] = mma16<F16>(vf[1], pf[0], ot[dt]);
            ot[dt] = mma16<F16>(vf[0], pf[1], ot[dt]);
          }
          ot[dt] = mma16<F16>(vf[0], pf[0], ot[dt]);
        }
      }

    if (more) store_tile(rg, (t + 1) & 1);
    __syncthreads();
  }

  // ---- normalise and write O[q][h*64 + d]: lane holds d = 32dt + 8g + 4hi + e for its query
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if constexpr (WLSE) if (q_ok && hi == 0) a.lse[((long)b * a.H + h) * a.Nq + qrow] = m_run + log2f(l_tot);   // training: P is recomputed from this
  if (q_ok) {
    bf16_t* orow = a.o_hi + ((long)b * a.Nq + qrow) * pld(a.ldo, oil);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        store_cols4(orow, h * D + 32 * dt + 8 * gq + 4 * hi, ot[dt][4 * gq + 0] * inv, ot[dt][4 * gq + 1] * inv,
                    ot[dt][4 * gq + 2] * inv, ot[dt][4 * gq + 3] * inv, a.o_fmt, oil);
  }
}

template <int NSPLIT, bool F16, int NW, bool WLSE, int D = 64>
static hipError_t launch_attn_w(const AttnArgs& a, hipStream_t s) {
  const size_t lds = 2 * (NSPLIT == 3 ? 2 : 1) * (AtGeom<D>::KPLANE + AtGeom<D>::VPLANE);
  static DynLdsAttr attr;
  {
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&attn_kernel<NSPLIT, F16, NW, WLSE, D>), (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid(((a.Nq + 32 * NW - 1) / (32 * NW)) * a.H * a.B);
  hipLaunchKernelGGL((attn_kernel<NSPLIT, F16, NW, WLSE, D>), grid, dim3(64 * NW), lds, s, a);
  return hipGetLastError();
}
template <int NSPLIT, bool F16>
static hipError_t launch_attn_t(const AttnArgs& a, hipStream_t s) {
  const int D = a.D > 0 ? a.D : 64;
  if (D == 32) return a.lse ? hipErrorInvalidValue : launch_attn_w<NSPLIT, F16, 4, false, 32>(a, s);      // (the backward kernels have a head dim of 64:
  if (D == 128) return a.lse ? hipErrorInvalidValue : launch_attn_w<NSPLIT, F16, 4, false, 128>(a, s);    //  training.unsupported_reason)
  if (D != 64) return hipErrorInvalidValue;
  if constexpr (NSPLIT == 3) { if (a.lse) return launch_attn_w<NSPLIT, F16, 4, true>(a, s); }   // training runs in precision 3
  else if (a.lse) return hipErrorInvalidValue;
  return launch_attn_w<NSPLIT, F16, 4, false>(a, s);
}

hipError_t launch_attention(const AttnArgs& a_in, int nsplit, hipStream_t s) {
  AttnArgs a = a_in;
  if (nsplit < 1 || nsplit > 4) return hipErrorInvalidValue;
  // precision 4 ("mixed"): the attention products themselves run as ONE IEEE-half product (they contribute nothing
  // measurable to the end-to-end error, tools/precision_study.py); only the output takes the FMT_H8 form its consumer
  // (the out-projection GEMM) multiplies in
  if (a.o_fmt < 0) a.o_fmt = nsplit == 4 ? FMT_H8 : (nsplit == 2 ? FMT_F16 : FMT_BF16);
  if ((a.o_fmt == FMT_H8 && !a.o_lo) || (a.o_fmt == FMT_F16 && a.o_lo)) return hipErrorInvalidValue;
  if (nsplit == 4) nsplit = 2;
  if (a.B <= 0 || a.H <= 0 || a.Nq <= 0 || a.Nk <= 0 || (a.vt_ld & 7)) return hipErrorInvalidValue;
  if (!planes_ok(a.q_hi, a.q_lo) || !planes_ok(a.k_hi, a.k_lo) || !planes_ok(a.vt_hi, a.vt_lo) || !planes_ok(a.o_hi, a.o_lo))
    return hipErrorInvalidValue;
  if (a.vt_lo && (a.vt_ld & 31)) return hipErrorInvalidValue;   // interleaved rows come in 32-column blocks
  if (nsplit == 3) {
    if (!a.q_lo || !a.k_lo || !a.vt_lo) return hipErrorInvalidValue;
    return launch_attn_t<3, false>(a, s);
  }
  if (nsplit == 2) {                                  // "half" precision: fp16 hi-only planes, one product
    if (a.q_lo || a.k_lo || a.vt_lo) return hipErrorInvalidValue;
    return launch_attn_t<1, true>(a, s);
  }
  if (a.o_fmt != FMT_BF16) return hipErrorInvalidValue;
  return launch_attn_t<1, false>(a, s);
}

NS2_DEFINE_SATURATION_READER(attention)

}  // namespace ns2
