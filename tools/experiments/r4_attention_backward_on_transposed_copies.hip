// The round-4 attention backward (attn_bwd_kernel): four tiles per 64 walked rows staged through registers into one LDS buffer, the
// transposed ones from per-utterance transposed copies (kt / qt / dot) -- replaced in round 5 by attn_bwd2_kernel (backward.hip);
// A/B: profiles/r05_attention_backward_ab.json (tools/runs/r5_attention_backward_ab.sh at commit 'Attention backward on LDS-DMA ...').
// Flash backward.  ROLE 0 (dQ): the workgroup OWNS 128 queries (4 waves x 32), walks 64-key tiles:
//     S^T = K Q^T ; dP^T = V dO^T ; P = exp2(S sl2 - lse) ; dS = P (dP - delta) ; dQ^T += K^T dS^T
//   ROLE 1 (dK, dV): owns 128 keys, walks 64-query tiles:
//     S = Q K^T ; dP = dO V^T ; P, dS as above with lse / delta per query (per register) ; dV^T += dO^T P ; dK^T += Q^T dS
// In both roles the first two products put the OWN row in the lane (col = lane & 31 of the MFMA C tile) and 16 walked rows in
// the registers -- the layout of attention.hip -- so P / dS feed the second pair of products as B operands straight from the
// registers, against tiles of the TRANSPOSED planes (K^T, Q^T, dO^T: [64 d][64 walked rows]) read from LDS.
constexpr int AB_ROWB = 144;
constexpr int AB_PLANE = 64 * AB_ROWB;
template <int ROLE>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const AttnBwdArgs a) {
  constexpr int NT = ROLE == 0 ? 3 : 4;            // LDS tiles: [Y, Yg, Y1T (, Y2T)], 2 planes each
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_lse = reinterpret_cast<float*>(smem + NT * 2 * AB_PLANE);
  float* s_del = s_lse + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  float lse_own = INFINITY, del_own = 0.f;        // role 0: per-lane statistics of the own query
  if (ROLE == 0 && own_ok) {
    lse_own = a.lse[((long)b * a.H + h) * a.Nq + orow];
    del_own = a.delta[((long)b * a.H + h) * a.Nq + orow];
  }

  // walked-side sources
  const bf16_t* yb = ROLE == 0 ? a.k_hi : a.q_hi;      // row-major, scores
  const int ldy = ROLE == 0 ? a.ldk : a.ldq, ycol = ROLE == 0 ? a.k_col0 : a.q_col0;
  const bf16_t* ygb = ROLE == 0 ? a.v_hi : a.do_hi;    // row-major, dP
  const int ldyg = ROLE == 0 ? a.ldv : a.lddo, ygcol = ROLE == 0 ? a.v_col0 : 0;
  const bf16_t* t1b = ROLE == 0 ? a.kt_hi : a.qt_hi;   // transposed [B][H*64][ld]
  const int ldt1 = ROLE == 0 ? a.kt_ld : a.qt_ld;
  const bf16_t* t2b = a.dot_hi;                        // role 1 only
  const int ldt2 = a.dot_ld;

  const int pi_row = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int y_frag_off = pi_row * AB_ROWB + hi * 16;
  const int t_frag_off = l31 * AB_ROWB + hi * 16;

  f32x16 acc1[2], acc2[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[dt][r] = 0.f; acc2[dt][r] = 0.f; }
  const float sl2 = a.scale * 1.4426950408889634f;
  const int ntiles = (Nwalk + 63) / 64;

  for (int t = 0; t < ntiles; ++t) {
    const int r0 = t * 64;
    __syncthreads();                                   // the previous tile's reads are done
    // ---- stage the tile: 2 chunks of 16 B per plane per thread and matrix
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cidx = tid + 256 * i;
      const int srow = cidx >> 3, sch = cidx & 7;
      const int off = srow * AB_ROWB + sch * 16;
      {                                                // row-major tiles: row = walked row r0 + srow, chunk = d 8 sch ..
        const int wr = r0 + srow;
        const bool ok = wr < Nwalk;
        const long base_y = ((long)b * Nwalk + wr) * 2L * ldy + pcol(ycol + h * 64 + sch * 8, true);
        const long base_g = ((long)b * Nwalk + wr) * 2L * ldyg + pcol(ygcol + h * 64 + sch * 8, true);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(smem + (0 * 2 + p) * AB_PLANE + off) = ok ? *reinterpret_cast<const uint4*>(yb + base_y + 32 * p) : z4;
          *reinterpret_cast<uint4*>(smem + (1 * 2 + p) * AB_PLANE + off) = ok ? *reinterpret_cast<const uint4*>(ygb + base_g + 32 * p) : z4;
        }
      }
      {                                                // transposed tiles: row = d srow, chunk = walked rows r0 + 8 sch ..
        const int wc = r0 + sch * 8;
        const bool ok = wc < Nwalk;                      // the transposed planes are zero beyond Nwalk up to their ld (tplanes)
        const long base1 = ((long)(b * a.H + h) * 64 + srow) * 2L * ldt1 + pcol(wc, true);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(smem + (2 * 2 + p) * AB_PLANE + off) = ok ? *reinterpret_cast<const uint4*>(t1b + base1 + 32 * p) : z4;
          if constexpr (ROLE == 1) {
            const long base2 = ((long)(b * a.H + h) * 64 + srow) * 2L * ldt2 + pcol(wc, true);
            *reinterpret_cast<uint4*>(smem + (3 * 2 + p) * AB_PLANE + off) = ok ? *reinterpret_cast<const uint4*>(t2b + base2 + 32 * p) : z4;
          }
        }
      }
    }
    if (ROLE == 1 && tid < 64) {
      const int q = r0 + tid;
      s_lse[tid] = q < a.Nq ? a.lse[((long)b * a.H + h) * a.Nq + q] : INFINITY;      // exp2(-inf) = 0: rows beyond Nq contribute nothing
      s_del[tid] = q < a.Nq ? a.delta[((long)b * a.H + h) * a.Nq + q] : 0.f;
    }
    __syncthreads();

#pragma unroll
    for (int js = 0; js < 2; ++js) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 yf[2], ygf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          yf[p] = *reinterpret_cast<const bf16x8*>(smem + (0 * 2 + p) * AB_PLANE + y_frag_off + js * 32 * AB_ROWB + c * 32);
          ygf[p] = *reinterpret_cast<const bf16x8*>(smem + (1 * 2 + p) * AB_PLANE + y_frag_off + js * 32 * AB_ROWB + c * 32);
        }
        st = mma16<false>(yf[1], xf[0][c], st);
        st = mma16<false>(yf[0], xf[1][c], st);
        st = mma16<false>(yf[0], xf[0][c], st);
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
          const float4 l0 = *reinterpret_cast<const float4*>(s_lse + w0), l1 = *reinterpret_cast<const float4*>(s_lse + w0 + 4);
          const float4 d0 = *reinterpret_cast<const float4*>(s_del + w0), d1 = *reinterpret_cast<const float4*>(s_del + w0 + 4);
          ls[0] = l0.x; ls[1] = l0.y; ls[2] = l0.z; ls[3] = l0.w; ls[4] = l1.x; ls[5] = l1.y; ls[6] = l1.z; ls[7] = l1.w;
          dl[0] = d0.x; dl[1] = d0.y; dl[2] = d0.z; dl[3] = d0.w; dl[4] = d1.x; dl[5] = d1.y; dl[6] = d1.z; dl[7] = d1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * g1 + e;
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sl2, -ls[e]));
          if (ROLE == 0 && r0 + 32 * js + 16 * g1 + 8 * hi + e >= Nwalk) p = 0.f;      // keys beyond Nk (role 1: lse = +inf did it)
          pv[r] = p;
          dsv[r] = p * (dp[r] - dl[e]);
        }
      }
      // ---- accumulate: acc1 += Y1T dS^T (dQ^T or dK^T), acc2 += Y2T P^T (dV^T, role 1)
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
          bf16x8 t1f[2];
#pragma unroll
          for (int p = 0; p < 2; ++p)
            t1f[p] = *reinterpret_cast<const bf16x8*>(smem + (2 * 2 + p) * AB_PLANE + t_frag_off + dt * 32 * AB_ROWB + js * 64 + g1 * 32);
          acc1[dt] = mma16<false>(t1f[1], dsf[0], acc1[dt]);
          acc1[dt] = mma16<false>(t1f[0], dsf[1], acc1[dt]);
          acc1[dt] = mma16<false>(t1f[0], dsf[0], acc1[dt]);
          if constexpr (ROLE == 1) {
            bf16x8 t2f[2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
              t2f[p] = *reinterpret_cast<const bf16x8*>(smem + (3 * 2 + p) * AB_PLANE + t_frag_off + dt * 32 * AB_ROWB + js * 64 + g1 * 32);
            acc2[dt] = mma16<false>(t2f[1], pf[0], acc2[dt]);
            acc2[dt] = mma16<false>(t2f[0], pf[1], acc2[dt]);
            acc2[dt] = mma16<false>(t2f[0], pf[0], acc2[dt]);
          }
        }
      }
    }
  }

  // ---- store: lane holds d = 32 dt + 8 gq + 4 hi + e of its own row
  if (!own_ok) return;
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

