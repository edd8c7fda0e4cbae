// Epilogue dispatch of a 256 x 256 output block whose eight waves hold 128 x 64 accumulator tiles: shared by gemm2_kernel (gemm2.hip) and
// the lean linear kernel (gemm3_kernel.h).  (Moved out of gemm2_kernel in round 6; the code is unchanged.)
#pragma once
#include <type_traits>

#include "gemm_epi_fast.h"

namespace ns2 {

constexpr int G2_BM = 256, G2_BN = 256;

// ---- epilogue of a 256 x 256 block whose waves hold 128 x 64 accumulator tiles (wave -> (wm, wn) = (wave & 1, wave >> 1)): shared by
// gemm2_kernel and the lean linear kernel (gemm3_kernel.h).  All waves are past the K loop's last barrier: the LDS ring is free, every
// wave takes a private 18 KiB region.
template <int NSPLIT, int EPI, bool F16>
NS2_DEVINL void g2_block_epilogue(f32x16 (&acc)[4][2], const GemmArgs& g, const int z, const int tm, const int tn, const int wave, const int lane,
                                  unsigned char* smem) {
  const int wm = wave & 1, wn = wave >> 1;
  const int row_base = tm * G2_BM + wm * 128;
  const int col_base = tn * G2_BN + wn * 64;
  const int ncols_needed = (EPI == EPI_GEGLU || EPI == EPI_F32 || EPI == EPI_QKV) ? g.N : max(g.N, g.out_ncols);
  const bool wave_active = col_base < ncols_needed;
  if (wave_active) {
    bool done = false;
    unsigned char* const wbuf = smem + wave * EPI_LDS_WAVE_BYTES;
    const int ocol_base = tn * 128 + wn * 32;
#ifndef G2_SLOW_EPILOGUE
    // Interior wave tiles (all 128 rows and 64 columns valid) take the streamlined epilogues of gemm_epi_fast.h; edge tiles
    // and the formats a kernel of this arithmetic does not normally write keep the generic path.
    if (row_base + 128 <= g.M) {
      // plane format of the output: kernels on IEEE-half operands write F16 / H8, kernels on bf16 operands bf16 planes
      auto planes = [&](auto&& fn) __attribute__((always_inline)) {
        const bool al = ((reinterpret_cast<uintptr_t>(g.out_hi) & 15) == 0) && (g.ldo_s & 31) == 0;
        if (!al) return false;
        if constexpr (F16) {
          if (g.out_fmt == FMT_F16 && !g.out_lo) { fn(std::integral_constant<int, PF_F16>{}); return true; }
          if (g.out_fmt == FMT_H8) { fn(std::integral_constant<int, PF_H8>{}); return true; }
          // bf16 hi / lo lines from the mixed product: q | k | v of the mixed TRAINING arithmetic, whose attention stays bf16 x3
          if constexpr (NSPLIT == 2 && EPI == EPI_SPLIT) { if (g.out_fmt == FMT_BF16 && g.out_lo) { fn(std::integral_constant<int, PF_BF16IL>{}); return true; } }
        } else {
          if (g.out_fmt == FMT_BF16 && g.out_lo) { fn(std::integral_constant<int, PF_BF16IL>{}); return true; }
          if constexpr (NSPLIT == 1) { if (g.out_fmt == FMT_BF16 && !g.out_lo) { fn(std::integral_constant<int, PF_BF16>{}); return true; } }
        }
        return false;
      };
      if constexpr (EPI == EPI_F32) {
        if (col_base + 64 <= g.N && g.act == 0 && (g.ldo_f & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out_f) & 15) == 0 &&
            (!g.resid || ((g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.resid) & 15) == 0))) {
          epi_f32_fast(acc, g, z, row_base, col_base, lane, wbuf);
          done = true;
        }
      } else if constexpr (EPI == EPI_GEGLU) {
        if (ocol_base + 32 <= g.out_ncols)
          done = planes([&](auto pf) __attribute__((always_inline)) { epi_geglu_fast<decltype(pf)::value>(acc, g, row_base, col_base, ocol_base, lane, wbuf); });
      } else if constexpr (EPI == EPI_SPLIT) {
        if (col_base + 64 <= g.N && g.act == 0)
          done = planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, true>(acc, g, z, row_base, col_base, lane, wbuf); });
      } else if constexpr (EPI == EPI_WAVENET) {
        if (col_base + 64 <= g.N)
          done = planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, false>(acc, g, z, row_base, col_base, lane, wbuf); });
      } else if constexpr (EPI == EPI_QKV) {
        if (col_base + 64 <= g.N && !g.bias) {
          if (col_base + 64 <= g.split_col) {
            done = planes([&](auto pf) __attribute__((always_inline)) { epi_planes_fast<decltype(pf)::value, false>(acc, g, 0, row_base, col_base, lane, wbuf); });
          } else if (col_base >= g.split_col && !g.vt_lo && g.vt_fmt == (F16 ? FMT_F16 : FMT_BF16) && g.seq_len > 0 &&
                     (g.seq_len & 127) == 0 && (g.vt_ld & 7) == 0 && (reinterpret_cast<uintptr_t>(g.vt_hi) & 15) == 0) {
            epi_vt_fast<F16>(acc, g, row_base, col_base, lane, wbuf);
            done = true;
          }
        }
      }
    }
#endif
    if (!done) {
      if constexpr (EPI == EPI_F32) {
        if (epi_lds_supported<EPI>(g, row_base)) {
          gemm_epilogue_lds<EPI, 2, 0>(acc, g, z, row_base, col_base, ocol_base, lane, wbuf);
          done = true;
        }
      }
      if (!done) gemm_epilogue<EPI, 4, 2>(acc, g, z, row_base, col_base, ocol_base, lane);
    }
  }
}

}  // namespace ns2
