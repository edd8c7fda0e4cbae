// Host-side shared declarations of libns2hip (capi.cpp <-> model_exec.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/ns2hip.h"
#include "ns2_kernels.h"

namespace ns2 {

void set_error(const char* fmt, ...);

struct PackedW {                 // bf16 split-plane weight, rows padded to 128, K contiguous
  bf16_t* hi = nullptr; bf16_t* lo = nullptr;
  int rows_p = 0, ldk = 0, N = 0, nkt = 0, kt_per_tap = 0;
  int fmt = 0;                   // PlaneFmt: bf16 planes (precisions 1 / 3), dense IEEE half (2), FMT_H8 lines (4)
  bf16_t* tl = nullptr;          // FMT_H8 linear weights: the tiled LDS images of the lean mixed linear kernel (gemm3_kernel.h), or null
  bf16_t* tw1 = nullptr; bf16_t* tw2 = nullptr;   // a Wavenet stack's FMT_H8 weights: tiled images of the lean block kernel (wavenet3_kernel.h), or null
  bf16_t* t3 = nullptr;          // k = 3 conv weights in dense IEEE half: the tiled LDS images of the dedicated FF-conv kernel (ffconv_kernel.h), or null
};

int gemm_f32(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int conv_taps, int dil, int seq_len,
             const float* bias, const float* resid, int ldr, float* out, int ldo, int prec, hipStream_t s, int pad_left = -1,
             int act = 0);
int gemm_f32_norm(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, const float* bias, const float* resid, int ldr,
                  float* out, int ldo, int prec, int seq_len, const float* gamma, const float* cond, int cond_ld, bf16_t* n_hi, bf16_t* n_lo,
                  int n_ld, int n_fmt, hipStream_t s, bool* fused);
int gemm_split(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int conv_taps, int dil, int seq_len,
               const float* bias, bf16_t* o_hi, bf16_t* o_lo, int ldo, int prec, hipStream_t s, int pad_left = -1, int act = 0,
               int out_fmt = -1);
int gemm_geglu(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, const float* pbias, bf16_t* o_hi,
               bf16_t* o_lo, int ldo, int prec, hipStream_t s, int out_fmt = -1, int out_ncols = 0);   // out_ncols: columns written (zeros beyond f); 0 = ldo
int gemm_qkv(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int seq_len, int split_col,
             bf16_t* o_hi, bf16_t* o_lo, int ldo, bf16_t* vt_hi, bf16_t* vt_lo, int vt_ld, int prec, hipStream_t s, int att_fmt = -1);
int gemm_wavenet(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, long a_zs, int M, int seq_len, int dil,
                 int dil_z, int nz, const float* b_conv, const float* b_res, long bias_zs, const float* film, int film_ld,
                 long film_zs, bf16_t* o_hi, bf16_t* o_lo, int ldo, long out_zs, int out_ncols, int prec, hipStream_t s, int p1_half = 0);

int pack_weight_public(const float* w, int rows, int cols, int taps, int geglu, const float* extra, int precision, PackedW* out,
                       std::vector<void*>* owned, hipStream_t s);
std::vector<int> geglu_row_map(int f, int rows_p);
int build_conv3_tiles(std::vector<void*>* owned, PackedW* w, hipStream_t s);
int build_lin_tiles(std::vector<void*>* owned, PackedW* w, hipStream_t s);     // model_exec.cpp: (re)build w->tl (FMT_H8 packs; a no-op for other formats)   // model_exec.cpp: (re)build w->t3 from the row-major pack

}  // namespace ns2

struct ns2_weight {
  ns2::PackedW w;
  int taps, geglu, has_extra, cols_p;
  int cols = 0;                  // source columns (per tap)
  int* d_map = nullptr;          // device copy of the row map the weight was packed with (ns2_weight_update re-packs in place)
  std::vector<void*> owned;
};
