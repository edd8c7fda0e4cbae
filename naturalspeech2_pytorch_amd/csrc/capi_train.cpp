// extern "C" surface of the backward (training) pass: declared in include/ns2hip.h under "training".  Kernels: backward.hip; the
// contractions are calls of the forward GEMM family (gemm.hip / gemm2.hip) on re-packed weights / transposed operand planes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "ns2_host.h"

using namespace ns2;

#define HIPRET(expr)                                                                 \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return NS2_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)
#define ARGCHK(cond, msg)            \
  do {                               \
    if (!(cond)) {                   \
      set_error("%s", msg);          \
      return NS2_ERR_ARG;            \
    }                                \
  } while (0)

extern "C" int ns2_weight_update(ns2_weight* w, const float* w_src, const float* extra1x1, void* stream) {
  ARGCHK(w && w_src && w->d_map, "ns2_weight_update: null weight / source");
  ARGCHK((extra1x1 != nullptr) == (w->has_extra != 0), "ns2_weight_update: extra1x1 must be given iff the weight was packed with one");
  hipStream_t s = (hipStream_t)stream;
  // same kernel and row map as ns2_weight_pack, in place: stream-ordered, no allocation, no synchronisation
  HIPRET(launch_pack_weight(w_src, w->cols, w->taps, w->cols_p, w->d_map, w->w.rows_p, w->w.hi, w->w.lo, w->w.ldk, 0, s, w->w.fmt));
  if (extra1x1)
    HIPRET(launch_pack_weight(extra1x1, w->cols, 1, w->cols_p, w->d_map, w->w.rows_p, w->w.hi, w->w.lo, w->w.ldk, w->taps * w->cols_p, s,
                              w->w.fmt));
  if (w->w.t3) return build_conv3_tiles(&w->owned, &w->w, s);      // the tiled copies follow the pack (no allocation: they exist)
  if (w->w.tl) return build_lin_tiles(&w->owned, &w->w, s);
  if (w->w.tw1) HIPRET(wavenet3_build_tiles(w->w.hi, w->w.rows_p, w->cols_p, 1, w->w.tw1, w->w.tw2, s));
  return NS2_OK;
}

// ---- all packs of a training pass in one launch (elementwise.hip repack_kernel).  The table lives in CALLER-owned device memory (the
// library keeps no state): ns2_weights_repack_build fills it from a host array of parts (once; synchronising copy), ns2_weights_repack
// replays it (stream-ordered, no allocation, no synchronisation).  A part is valid while its weight handle and its source storage live.
extern "C" int64_t ns2_weights_repack_table_bytes(int n) { return n > 0 ? (int64_t)n * (int64_t)sizeof(RepackDesc) : 0; }
extern "C" int ns2_weights_repack_build(const ns2_repack_part* parts, int n, void* table_device, int64_t table_bytes, int64_t* total_blocks, void* stream) {
  ARGCHK(parts && n > 0 && table_device && total_blocks, "ns2_weights_repack_build: null argument");
  ARGCHK(table_bytes >= ns2_weights_repack_table_bytes(n), "ns2_weights_repack_build: table too small (ns2_weights_repack_table_bytes)");
  std::vector<RepackDesc> tab((size_t)n);
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    const ns2_repack_part& p = parts[i];
    ARGCHK(p.w && p.src, "ns2_weights_repack_build: null weight / source");
    const ns2_weight* w = p.w;
    ARGCHK(!w->geglu && !w->has_extra, "ns2_weights_repack_build: plain packs only (no GEGLU row permutation, no extra 1x1 block)");
    ARGCHK(p.rows > 0 && p.cols > 0 && p.row0 >= 0 && p.col0 >= 0 && p.row0 + p.rows <= w->w.N && p.col0 + p.cols <= w->cols,
           "ns2_weights_repack_build: part outside its weight");
    RepackDesc& d = tab[(size_t)i];
    const bool il = w->w.lo != nullptr;
    d.src = p.src; d.sr = (long)p.sr; d.sc = (long)p.sc; d.st = (long)p.st;
    d.dst_hi = w->w.hi; d.drs = il ? 2L * w->w.ldk : (long)w->w.ldk;
    d.row0 = p.row0; d.rows = p.rows; d.col0 = p.col0; d.cols = p.cols;
    d.Cp = w->cols_p; d.T = w->taps; d.fmt = w->w.fmt; d.il = il ? 1 : 0;
    d.chunks = (w->taps * ((p.cols + 3) / 4) + 255) / 256; d.pad_ = 0;          // a thread packs 4 columns
    d.block0 = blocks;
    blocks += (long)p.rows * d.chunks;
  }
  ARGCHK(blocks <= 0x7fffffffL, "ns2_weights_repack_build: too many blocks for one launch");
  HIPRET(hipMemcpyAsync(table_device, tab.data(), (size_t)n * sizeof(RepackDesc), hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPRET(hipStreamSynchronize((hipStream_t)stream));         // the host vector dies here
  *total_blocks = blocks;
  return NS2_OK;
}
// the tiled images (ffconv_kernel.h / gemm3_kernel.h / wavenet3_kernel.h) of weights whose packs ns2_weights_repack has just refreshed:
// one small launch per weight that has images, none for the others; stream-ordered, no allocation (the images exist)
extern "C" int ns2_weights_retile(ns2_weight* const* ws, int n, void* stream) {
  ARGCHK(ws && n > 0, "ns2_weights_retile: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    ns2_weight* w = ws[i];
    ARGCHK(w, "ns2_weights_retile: null weight");
    int r = NS2_OK;
    if (w->w.t3) r = build_conv3_tiles(&w->owned, &w->w, s);
    else if (w->w.tl) r = build_lin_tiles(&w->owned, &w->w, s);
    else if (w->w.tw1) HIPRET(wavenet3_build_tiles(w->w.hi, w->w.rows_p, w->cols_p, 1, w->w.tw1, w->w.tw2, s));
    if (r != NS2_OK) return r;
  }
  return NS2_OK;
}
extern "C" int ns2_weights_repack(const void* table_device, int n, int64_t total_blocks, void* stream) {
  ARGCHK(table_device && n > 0 && total_blocks > 0, "ns2_weights_repack: bad arguments");
  HIPRET(launch_repack(static_cast<const RepackDesc*>(table_device), n, (long)total_blocks, (hipStream_t)stream));
  return NS2_OK;
}

// the training kernels' share of the range guard (ns2_saturation_count sums it in; this is its stream-ordered, non-synchronising
// read for a training loop under the mixed arithmetic: one word into pinned host memory)
extern "C" int ns2_saturation_peek_train_async(unsigned int* host1, void* stream) {
  ARGCHK(host1 != nullptr, "ns2_saturation_peek_train_async: null pointer");
  HIPRET(saturation_peek_backward(host1, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int64_t ns2_grad_prep_slices(int M, int64_t ld_t) { return M > 0 ? (int64_t)tplanes_slices(M, (long)ld_t) : 0; }

extern "C" int ns2_grad_prep(const float* x, int64_t ldx, int M, int C, int seq_len, int shift, uint16_t* row_hi, uint16_t* row_lo,
                             int ld_row, uint16_t* t_hi, uint16_t* t_lo, int64_t ld_t, int t_rows, int per_batch, float* colsum_partial,
                             int precision, void* stream) {
  ARGCHK(x && (row_hi || t_hi || colsum_partial), "ns2_grad_prep: nothing to do");
  ARGCHK(precision == 3 || precision == 4, "ns2_grad_prep: precision 3 (bf16 hi / lo planes) or 4 (FMT_H8 lines)");
  TPlanesArgs a;
  memset(&a, 0, sizeof a);
  a.xf = x; a.ldx = (long)ldx; a.M = M; a.C = C; a.seq_len = seq_len; a.shift = shift;
  a.row_hi = row_hi; a.row_lo = row_lo; a.ld_row = ld_row;
  a.t_hi = t_hi; a.t_lo = t_lo; a.ld_t = (long)ld_t; a.per_batch = per_batch; a.t_rows_per_batch = t_rows; a.t_rows = t_rows;
  a.colsum_partial = colsum_partial;
  a.fmt = precision == 4 ? FMT_H8 : FMT_BF16;
  HIPRET(launch_tplanes(a, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_planes_transpose(const uint16_t* in_hi, const uint16_t* in_lo, int ld_in, int in_col0, int M, int C, int seq_len,
                                    int shift, uint16_t* t_hi, uint16_t* t_lo, int64_t ld_t, int t_rows, int per_batch, int precision,
                                    void* stream) {
  ARGCHK(in_hi && in_lo && t_hi && t_lo, "ns2_planes_transpose: null pointer (operands are interleaved lines: lo = hi + 32)");
  ARGCHK(precision == 3 || precision == 4, "ns2_planes_transpose: precision 3 (bf16 hi / lo planes) or 4 (FMT_H8 lines)");
  TPlanesArgs a;
  memset(&a, 0, sizeof a);
  a.in_hi = in_hi; a.in_lo = in_lo; a.ld_in = ld_in; a.in_col0 = in_col0; a.M = M; a.C = C; a.seq_len = seq_len; a.shift = shift;
  a.t_hi = t_hi; a.t_lo = t_lo; a.ld_t = (long)ld_t; a.per_batch = per_batch; a.t_rows_per_batch = t_rows; a.t_rows = t_rows;
  a.fmt = precision == 4 ? FMT_H8 : FMT_BF16;
  HIPRET(launch_tplanes(a, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_reduce_slices(const float* partial, int64_t outer, int S, int64_t inner, float* out, int accumulate, void* stream) {
  ARGCHK(partial && out, "ns2_reduce_slices: null pointer");
  HIPRET(launch_reduce_slices(partial, (long)outer, S, (long)inner, out, accumulate, (hipStream_t)stream));
  return NS2_OK;
}

// ---- weight gradient: dW[r, k, t] = sum_m dY[m, r] * X_t[m, k]  as ONE GEMM over transposed planes, split-K into fixed slots
// Slices of the token axis: every slice is a grid-z layer of output tiles writing its own slot.  The 256 x 256 kernel runs ONE workgroup
// per CU, so a launch costs ceil(tiles * S / 256) rounds of (nkt / S) K tiles each; among the divisors of nkt that leave >= 8 K tiles
// per slice take the cheapest -- a slice also pays for storing its fp32 slot and for the fixed-order sum reading it (~12 K-tile
// times) -- and of equally cheap ones the smallest.  (Round 4 aimed at ~2.5 workgroups per CU whatever the shape: a 512 x 512 gradient
// got 128 slots = 128 MB of partial sums for two rounds of 8-tile blocks; now 64 slots, one round of 16-tile blocks.)
static int wgrad_split(int R, int ncols, int64_t ld_t) {
  const int nkt = (int)(ld_t / 32);
  const bool big = ncols > 128;
  const long tiles = big ? (long)((R + 255) / 256) * ((ncols + 255) / 256) : (long)((R + 127) / 128);
  const long per_round = big ? 256 : 512;                // workgroups the chip runs at once (the 128 x 128 kernel: two per CU)
  int S = 1;
  long best = -1;
  for (int d = 1; d <= nkt && d <= std::max(1, nkt / 8); ++d) {
    if (nkt % d) continue;                               // slices must cover whole K tiles evenly
    const long cost = ((tiles * d + per_round - 1) / per_round) * (nkt / d + 12);     // + a slot's fp32 store and its later read, in K-tile times
    if (best < 0 || cost < best) { best = cost; S = d; }
  }
  return S;
}
extern "C" int64_t ns2_wgrad_workspace_bytes(int R, int ncols, int64_t ld_t) {
  if (R <= 0 || ncols <= 0 || ld_t <= 0 || (ld_t & 31)) return 0;
  return (int64_t)wgrad_split(R, ncols, ld_t) * R * ncols * (int64_t)sizeof(float);
}
extern "C" int ns2_wgrad(const uint16_t* dyt_hi, const uint16_t* dyt_lo, const uint16_t* xt_hi, const uint16_t* xt_lo, int64_t ld_t, int R,
                         int T, int Kp, int K, float* dw, void* workspace, int64_t workspace_bytes, int precision, void* stream) {
  ARGCHK(dyt_hi && dyt_lo == dyt_hi + 32 && xt_hi && xt_lo == xt_hi + 32 && dw && workspace, "ns2_wgrad: null pointer / operands must be interleaved lines (lo = hi + 32)");
  ARGCHK(precision == 3 || precision == 4, "ns2_wgrad: precision 3 (bf16 x3) or 4 (half product + fp8 correction terms on FMT_H8 lines)");
  ARGCHK(R > 0 && T > 0 && K > 0 && Kp >= K && (Kp & 3) == 0 && ld_t > 0 && (ld_t & 31) == 0 && ld_t < (1LL << 30), "ns2_wgrad: bad shapes");
  const int ncols = T * Kp;
  const int S = wgrad_split(R, ncols, ld_t);
  ARGCHK(workspace_bytes >= (int64_t)S * R * ncols * (int64_t)sizeof(float), "ns2_wgrad: workspace too small (ns2_wgrad_workspace_bytes)");
  GemmArgs g;
  memset(&g, 0, sizeof g);
  const int nkt = (int)(ld_t / 32) / S;
  g.a_hi = dyt_hi; g.a_lo = dyt_lo; g.lda = (int)ld_t;
  g.w_hi = xt_hi; g.w_lo = xt_lo; g.ldw = (int)ld_t;        // rows of X^T beyond T * Kp up to the next multiple of 256 must exist (zeros)
  g.M = R; g.N = ncols; g.nkt = nkt; g.kt_per_tap = nkt; g.conv_taps = 0; g.dil = 1; g.mid_kt = -1;
  g.nz = S; g.a_zs = (long)nkt * 32; g.w_zs = (long)nkt * 32; g.out_f_zs = (long)R * ncols;
  g.pad_left = -1; g.out_fmt = -1; g.vt_fmt = -1;
  g.epi = EPI_F32; g.out_f = (float*)workspace; g.ldo_f = ncols;
  HIPRET(launch_gemm(g, precision, (hipStream_t)stream));
  HIPRET(launch_wgrad_reduce((const float*)workspace, S, R, ncols, T, Kp, K, dw, (hipStream_t)stream));
  return NS2_OK;
}

// The same gradient straight from the TOKEN-MAJOR planes (gemm2.hip TR: LDS transpose reads form the fragments; no transposed
// copies, no shifted copies of a conv's input).  Same slots, same fixed-order sum.
// The TR kernel is the 256 x 256 kernel: gradients it would run on a handful of half-empty tiles (the dim = 128 model, short batches) keep
// the transposed route, where launch_gemm picks the 128 x 128 kernel by the same rule (gemm2.hip launch_gemm: <= 64 blocks of 256 x 256).
extern "C" int ns2_wgrad_rows_preferred(int R, int ncols, int64_t M) {
  if (R <= 128 || ncols <= 128 || M <= 0) return 0;
  const int64_t ld_t = (M + 31) / 32 * 32;
  const long blocks256 = (long)((R + 255) / 256) * ((ncols + 255) / 256) * wgrad_split(R, ncols, ld_t);
  return blocks256 > 64;
}
extern "C" int ns2_wgrad_rows(const uint16_t* dy_hi, const uint16_t* dy_lo, int ld_dy, const uint16_t* x_hi, const uint16_t* x_lo, int ld_x,
                              int64_t M, int R, int T, int Kp, int K, int dil, int seq_len, float* dw, void* workspace, int64_t workspace_bytes,
                              int precision, void* stream) {
  ARGCHK(dy_hi && dy_lo == dy_hi + 32 && x_hi && x_lo == x_hi + 32 && dw && workspace, "ns2_wgrad_rows: null pointer / operands must be interleaved lines (lo = hi + 32)");
  ARGCHK(precision == 3 || precision == 4, "ns2_wgrad_rows: precision 3 (bf16 x3) or 4 (half product + fp8 correction terms on FMT_H8 lines)");
  ARGCHK(M > 0 && M < (1LL << 30) && R > 0 && T > 0 && K > 0 && Kp >= K && (Kp & 31) == 0 && (ld_dy & 31) == 0 && (ld_x & 31) == 0 && ld_dy >= R && ld_x >= Kp,
         "ns2_wgrad_rows: bad shapes (Kp, ld_dy, ld_x multiples of 32; ld_dy >= R; ld_x >= Kp)");
  ARGCHK(T == 1 || (seq_len >= 32 && dil >= 1 && M % seq_len == 0), "ns2_wgrad_rows: a conv's gradient needs utterances of seq_len >= 32 tokens");
  const int ncols = T * Kp;
  const int64_t ld_t = (M + 31) / 32 * 32;
  const int S = wgrad_split(R, ncols, ld_t);
  ARGCHK(workspace_bytes >= (int64_t)S * R * ncols * (int64_t)sizeof(float), "ns2_wgrad_rows: workspace too small (ns2_wgrad_workspace_bytes(R, T * Kp, round_up(M, 32)))");
  GemmArgs g;
  memset(&g, 0, sizeof g);
  const int nkt = (int)(ld_t / 32) / S;
  g.a_hi = dy_hi; g.a_lo = dy_lo; g.lda = ld_dy;
  g.w_hi = x_hi; g.w_lo = x_lo; g.ldw = ld_x;
  g.M = R; g.N = ncols; g.nkt = nkt; g.kt_per_tap = nkt; g.conv_taps = 0; g.dil = dil > 0 ? dil : 1; g.seq_len = T > 1 ? seq_len : 0; g.mid_kt = -1;
  g.nz = S; g.out_f_zs = (long)R * ncols;
  g.pad_left = -1; g.out_fmt = -1; g.vt_fmt = -1;
  g.epi = EPI_F32; g.out_f = (float*)workspace; g.ldo_f = ncols;
  g.tr_tokens = (int)M; g.tr_kp = Kp; g.tr_taps = T > 1 ? T : 0;
  HIPRET(launch_gemm_tr(g, precision, (hipStream_t)stream));
  HIPRET(launch_wgrad_reduce((const float*)workspace, S, R, ncols, T, Kp, K, dw, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_film_gate_fwd(const float* h, int64_t ldh, const float* film, int film_ld, int seq_len, int64_t M, int d, float* out,
                                 int64_t ldo, void* stream) {
  ARGCHK(h && film && out, "ns2_film_gate_fwd: null pointer");
  HIPRET(launch_film_gate_fwd(h, (long)ldh, film, film_ld, seq_len, (long)M, d, out, (long)ldo, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_film_gate_slices(int seq_len) { return seq_len > 0 ? film_gate_slices(seq_len) : 0; }
extern "C" int ns2_film_gate_bwd(const float* dg, int64_t lddg, const float* h, int64_t ldh, const float* film, int film_ld, int B,
                                 int seq_len, int d, float* dh, int64_t lddh, float* partial, void* stream) {
  ARGCHK(dg && h && film && dh && partial, "ns2_film_gate_bwd: null pointer");
  HIPRET(launch_film_gate_bwd(dg, (long)lddg, h, (long)ldh, film, film_ld, B, seq_len, d, dh, (long)lddh, partial, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_geglu_fwd(const float* pre, int64_t ldp, int64_t M, int f, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision,
                             void* stream) {
  ARGCHK(pre && out_hi && out_lo, "ns2_geglu_fwd: null pointer");
  ARGCHK(precision == 3 || precision == 4, "ns2_geglu_fwd: precision 3 (bf16 hi / lo planes) or 4 (FMT_H8 lines)");
  HIPRET(launch_geglu_fwd(pre, (long)ldp, (long)M, f, out_hi, out_lo, ldo, (hipStream_t)stream, precision == 4 ? FMT_H8 : FMT_BF16));
  return NS2_OK;
}
extern "C" int ns2_geglu_bwd(const float* dh, int64_t lddh, const float* pre, int64_t ldp, int64_t M, int f, float* dpre, int64_t lddp,
                             void* stream) {
  ARGCHK(dh && pre && dpre, "ns2_geglu_bwd: null pointer");
  HIPRET(launch_geglu_bwd(dh, (long)lddh, pre, (long)ldp, (long)M, f, dpre, (long)lddp, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_rmsnorm_bwd_slices(int seq_len) { return seq_len > 0 ? rmsnorm_bwd_slices(seq_len) : 0; }
extern "C" int ns2_rmsnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma, const float* cond,
                               int cond_ld, int B, int seq_len, int d, const float* dx_add, float* dx, int64_t lddx, float* cond_partial,
                               float* gamma_partial, void* stream) {
  ARGCHK(x && dy && dx, "ns2_rmsnorm_bwd: null pointer");
  NormBwdArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.ldx = (long)ldx; a.dy = dy; a.lddy = (long)lddy; a.gamma = gamma; a.cond = cond; a.cond_ld = cond_ld;
  a.dx_add = dx_add; a.dx = dx; a.lddx = (long)lddx; a.cond_partial = cond_partial; a.gamma_partial = gamma_partial;
  a.B = B; a.seq_len = seq_len; a.d = d;
  HIPRET(launch_rmsnorm_bwd(a, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_attention_lse(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi, const uint16_t* k_lo,
                                 int ldk, int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld, uint16_t* o_hi, uint16_t* o_lo,
                                 int ldo, int B, int H, int Nq, int Nk, float scale, float* lse, int precision, int o_precision, void* stream) {
  ARGCHK(q_hi && k_hi && vt_hi && o_hi && lse && precision >= 1 && precision <= 4, "ns2_attention_lse: bad arguments");
  ARGCHK(o_precision == 0 || o_precision == 3 || o_precision == 4, "ns2_attention_lse: o_precision 0 (= precision), 3 (bf16 hi / lo) or 4 (FMT_H8)");
  AttnArgs a;
  a.D = 64;
  a.q_hi = q_hi; a.q_lo = q_lo; a.ldq = ldq; a.q_col0 = q_col0;
  a.k_hi = k_hi; a.k_lo = k_lo; a.ldk = ldk; a.k_col0 = k_col0;
  a.vt_hi = vt_hi; a.vt_lo = vt_lo; a.vt_ld = vt_ld;
  a.o_hi = o_hi; a.o_lo = o_lo; a.ldo = ldo; a.o_fmt = o_precision == 4 ? FMT_H8 : (o_precision == 3 ? FMT_BF16 : -1);
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale; a.kmask = nullptr; a.lse = lse;
  HIPRET(launch_attention(a, precision, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_attention_delta(const float* d_out, int64_t ld_dout, const uint16_t* o_hi, const uint16_t* o_lo, int ldo, int B, int H,
                                   int Nq, float* delta, int o_precision, void* stream) {
  ARGCHK(d_out && o_hi && delta, "ns2_attention_delta: null pointer");
  ARGCHK(o_precision == 3 || o_precision == 4, "ns2_attention_delta: o_precision 3 (bf16 hi / lo planes) or 4 (FMT_H8 lines)");
  HIPRET(launch_attn_delta(d_out, (long)ld_dout, o_hi, o_lo, ldo, B, H, Nq, delta, (hipStream_t)stream, o_precision == 4 ? FMT_H8 : FMT_BF16));
  return NS2_OK;
}
extern "C" int ns2_attention_bwd(const ns2_attn_bwd_args* p, void* stream) {
  ARGCHK(p != nullptr, "ns2_attention_bwd: null argument block");
  AttnBwdArgs a;
  memset(&a, 0, sizeof a);
  a.q_hi = p->q_hi; a.q_lo = p->q_lo; a.ldq = p->ldq; a.q_col0 = p->q_col0;
  a.k_hi = p->k_hi; a.k_lo = p->k_lo; a.ldk = p->ldk; a.k_col0 = p->k_col0;
  a.v_hi = p->v_hi; a.v_lo = p->v_lo; a.ldv = p->ldv; a.v_col0 = p->v_col0;
  a.do_hi = p->do_hi; a.do_lo = p->do_lo; a.lddo = p->lddo;
  a.lse = p->lse; a.delta = p->delta;
  a.dq = p->dq; a.lddq = p->lddq; a.dq_col0 = p->dq_col0;
  a.dk = p->dk; a.lddk = p->lddk; a.dk_col0 = p->dk_col0;
  a.dv = p->dv; a.lddv = p->lddv; a.dv_col0 = p->dv_col0;
  a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.scale = p->scale;
  a.gp_hi = p->gp_hi; a.gp_lo = p->gp_lo; a.gp_ld = p->gp_ld; a.gp_q = p->gp_q; a.gp_kv = p->gp_kv;
  ARGCHK(p->gp_precision == 0 || p->gp_precision == 3 || p->gp_precision == 4, "ns2_attention_bwd: gp_precision 3 (bf16 hi / lo planes) or 4 (FMT_H8 lines)");
  a.gp_fmt = p->gp_precision == 4 ? FMT_H8 : FMT_BF16;
  HIPRET(launch_attention_bwd(a, (hipStream_t)stream));
  return NS2_OK;
}
