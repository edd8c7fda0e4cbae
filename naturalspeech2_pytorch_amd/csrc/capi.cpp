// extern "C" surface of libns2hip (declared in include/ns2hip.h): error plumbing + op-level entry points.
// The Model executor entry points live in model_exec.cpp.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ns2_host.h"

namespace ns2 {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

}  // namespace ns2

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
static inline int prec_ok(int p) { return p >= 1 && p <= 4; }
static inline int op_fmt(int p) { return p == 2 ? FMT_F16 : (p == 4 ? FMT_H8 : FMT_BF16); }

extern "C" const char* ns2_last_error(void) { return g_err; }
extern "C" int ns2_version(void) { return 115; }   // 115: ns2_weights_retile; 114: ns2_attention_hd (head dims 32 / 64 / 128), ns2_model_create takes dim_head 32 / 128; 113: ns2_weight_tile_conv3 + ns2_conv3_input_ld (the dedicated FF causal conv kernel, ffconv_kernel.h), ns2_weight_tile_linear (gemm3_kernel.h), ns2_weight_tile_wavenet (wavenet3_kernel.h), ns2_debug_force_gemm(4 / 5); 112: ns2_seanet_resblock_narrow; 111: training entry points take a precision (3 = bf16 x3, 4 = mixed on FMT_H8 lines), ns2_linear_split_as; 110: backward pass (capi_train.cpp: ns2_wgrad, ns2_attention_bwd, ...), ns2_weight_update; 109: ns2_seanet_conv_narrow; 108: ns2_seanet_prep2; 107: ns2_lstm2 (two LSTM layers, one launch); 106: ns2_saturation_peek_async; 105: ns2_lstm_layer takes the scratch size (persistent recurrence); 104: model precision 5 (per-site plan); 103: precision 2 / 4 at op level, caller-owned skinny-linear scratch
extern "C" int ns2_debug_force_gemm(int kernel) {
  ARGCHK(kernel >= 0 && kernel <= 5, "ns2_debug_force_gemm: 0 auto, 1 = 128x128 kernel, 2 = 256x256 kernel, 3 = auto without split-K, 4 = auto without the dedicated FF-conv kernel, 5 = auto, the FF-conv kernel whenever eligible");
  force_gemm_kernel(kernel);
  return NS2_OK;
}

extern "C" int ns2_weight_pack(const float* w, int rows, int cols, int taps, int geglu, const float* extra1x1, int precision,
                               ns2_weight** out, void* stream) {
  ARGCHK(w && out && rows > 0 && cols > 0 && taps >= 1 && prec_ok(precision), "ns2_weight_pack: bad arguments");
  ARGCHK(!(geglu && (taps != 1 || extra1x1 || (rows & 1))), "ns2_weight_pack: geglu needs taps=1, no extra, even rows");
  ns2_weight* h = new ns2_weight();
  h->taps = taps; h->geglu = geglu; h->has_extra = extra1x1 != nullptr; h->cols_p = (cols + 31) / 32 * 32;
  int r = pack_weight_public(w, rows, cols, taps, geglu, extra1x1, precision, &h->w, &h->owned, (hipStream_t)stream);
  if (r != NS2_OK) { ns2_weight_free(h); return r; }
  {   // keep the row map on the device: ns2_weight_update (training: the weights change every step) re-packs without allocating
    h->cols = cols;
    const std::vector<int> m = geglu ? geglu_row_map(rows / 2, h->w.rows_p) : [&] {
      std::vector<int> id(h->w.rows_p, -1);
      for (int i = 0; i < rows && i < h->w.rows_p; ++i) id[i] = i;
      return id;
    }();
    if (hipMalloc((void**)&h->d_map, m.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(h->d_map, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("ns2_weight_pack: could not keep the row map");
      ns2_weight_free(h);
      return NS2_ERR_HIP;
    }
    h->owned.push_back(h->d_map);
  }
  *out = h;
  return NS2_OK;
}
// Give a k = 3 conv weight packed for precision 2 (dense IEEE half) the tiled images of the dedicated FF causal conv kernel
// (ffconv_kernel.h).  ns2_linear_split / ns2_linear_split_as then take that kernel whenever the call is eligible: dilation 1, causal
// padding, no activation, M and seq_len multiples of 256, activations with rows of ns2_conv3_input_ld(cols) elements, FMT_H8 or dense
// half output.  One-time set-up like ns2_weight_pack (allocates on the first call); ns2_weight_update refreshes the images with the pack.
extern "C" int ns2_weight_tile_conv3(ns2_weight* w, void* stream) {
  ARGCHK(w && w->taps == 3 && !w->geglu && !w->has_extra && w->w.fmt == FMT_F16 && !w->w.lo, "ns2_weight_tile_conv3: a k = 3 conv weight packed for precision 2");
  return build_conv3_tiles(&w->owned, &w->w, (hipStream_t)stream);
}
// Give a linear weight (taps = 1; also the GEGLU packing) packed for precision 4 (FMT_H8 lines) the tiled images of the lean mixed linear
// kernel (gemm3_kernel.h): ns2_linear_f32 / _split / _qkv / _geglu then take that kernel when M % 256 == 0 and K >= 96.  One-time set-up
// (allocates on the first call); ns2_weight_update refreshes the images with the pack; after ns2_weights_repack, ns2_weights_retile does.
extern "C" int ns2_weight_tile_linear(ns2_weight* w, void* stream) {
  ARGCHK(w && w->taps == 1 && !w->has_extra && w->w.fmt == FMT_H8 && w->w.nkt >= 3, "ns2_weight_tile_linear: a linear weight (taps = 1, K >= 96) packed for precision 4");
  return build_lin_tiles(&w->owned, &w->w, (hipStream_t)stream);
}
// Give a WavenetResBlock weight (taps = 3 + the 1x1 res conv, packed for precision 4) the tiled images of the lean block kernel of the
// hybrid plan (wavenet3_kernel.h): ns2_wavenet_block at precision 5 then takes that kernel when channels % 256 == 0 (square), M and
// seq_len are multiples of 256 and dilation <= 128.  Same life-cycle rules as ns2_weight_tile_conv3.
extern "C" int ns2_weight_tile_wavenet(ns2_weight* w, void* stream) {
  ARGCHK(w && w->taps == 3 && w->has_extra && !w->geglu && w->w.fmt == FMT_H8 && (w->cols_p % 128) == 0 && w->w.N == w->cols_p && (w->w.N % 256) == 0,
         "ns2_weight_tile_wavenet: a square WavenetResBlock weight (taps = 3, extra1x1) packed for precision 4, channels a multiple of 256");
  PackedW& W = w->w;
  if (!W.tw1) {
    void *t1 = nullptr, *t2 = nullptr;
    HIPRET(hipMalloc(&t1, wavenet3_tiles_bytes(W.rows_p, w->cols_p, 1, 1))); w->owned.push_back(t1);
    HIPRET(hipMalloc(&t2, wavenet3_tiles_bytes(W.rows_p, w->cols_p, 1, 2))); w->owned.push_back(t2);
    W.tw1 = static_cast<bf16_t*>(t1); W.tw2 = static_cast<bf16_t*>(t2);
  }
  HIPRET(wavenet3_build_tiles(W.hi, W.rows_p, w->cols_p, 1, W.tw1, W.tw2, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_conv3_input_ld(int cols) { return cols > 0 ? ffconv3_lda((cols + 31) / 32 * 32) : 0; }
extern "C" void ns2_weight_free(ns2_weight* w) {
  if (!w) return;
  for (void* p : w->owned) (void)hipFree(p);
  delete w;
}

// the weight's element format must be the one the requested precision multiplies in (fp16 for 2, bf16 planes for 1 / 3)
#define WFMT(w, precision, who) ARGCHK(op_fmt(precision) == (w)->w.fmt, who ": weight was packed for a different precision")

extern "C" int ns2_split_f32(const float* x, int ldx, int M, int d, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision,
                             void* stream) {
  ARGCHK(x && out_hi && prec_ok(precision), "ns2_split_f32: bad arguments");
  ARGCHK(precision != 2 || !out_lo, "ns2_split_f32: precision 2 (fp16) has no lo plane");
  HIPRET(launch_split(x, ldx, nullptr, 0, 0, 0, out_hi, out_lo, ldo, M, d, 0, (hipStream_t)stream, op_fmt(precision)));
  return NS2_OK;
}
extern "C" int ns2_join_f32(const uint16_t* hi, const uint16_t* lo, int ld, float* out, int ldo, int64_t M, int d, int precision,
                            void* stream) {
  ARGCHK(hi && out && prec_ok(precision), "ns2_join_f32: bad arguments");
  ARGCHK(precision != 2 || !lo, "ns2_join_f32: precision 2 (fp16) has no lo plane");
  HIPRET(launch_join(hi, lo, ld, out, ldo, (long)M, d, (hipStream_t)stream, op_fmt(precision)));
  return NS2_OK;
}

extern "C" int ns2_linear_f32(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                              int dilation, int seq_len, const float* bias, const float* resid, int ldr, float* out, int ldo,
                              int pad_left, int act, int precision, void* stream) {
  ARGCHK(w && a_hi && out && prec_ok(precision), "ns2_linear_f32: bad arguments");
  WFMT(w, precision, "ns2_linear_f32");
  ARGCHK(!w->geglu && !w->has_extra && (conv_taps == 0 ? w->taps == 1 : w->taps == conv_taps), "ns2_linear_f32: weight packing does not match");
  ARGCHK(lda >= w->cols_p, "ns2_linear_f32: lda smaller than the padded K");
  ARGCHK(pad_left >= -1 && pad_left < (conv_taps > 0 ? conv_taps : 1) && (act >= 0 && act <= 2), "ns2_linear_f32: bad pad_left / act");
  return gemm_f32(w->w, a_hi, a_lo, lda, M, conv_taps, dilation, seq_len, bias, resid, ldr, out, ldo, precision, (hipStream_t)stream,
                  pad_left, act);
}
extern "C" int ns2_linear_split(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                                int dilation, int seq_len, const float* bias, uint16_t* out_hi, uint16_t* out_lo, int ldo,
                                int pad_left, int act, int precision, void* stream) {
  ARGCHK(w && a_hi && out_hi && prec_ok(precision), "ns2_linear_split: bad arguments");
  WFMT(w, precision, "ns2_linear_split");
  ARGCHK(!w->geglu && !w->has_extra && (conv_taps == 0 ? w->taps == 1 : w->taps == conv_taps), "ns2_linear_split: weight packing does not match");
  ARGCHK(lda >= w->cols_p && (ldo & 1) == 0, "ns2_linear_split: bad leading dimensions");
  ARGCHK(pad_left >= -1 && pad_left < (conv_taps > 0 ? conv_taps : 1) && (act >= 0 && act <= 2), "ns2_linear_split: bad pad_left / act");
  return gemm_split(w->w, a_hi, a_lo, lda, M, conv_taps, dilation, seq_len, bias, out_hi, out_lo, ldo, precision, (hipStream_t)stream,
                    pad_left, act);
}
extern "C" int ns2_linear_split_as(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                                   int dilation, int seq_len, const float* bias, uint16_t* out_hi, uint16_t* out_lo, int ldo,
                                   int pad_left, int act, int precision, int out_precision, void* stream) {
  ARGCHK(w && a_hi && out_hi && prec_ok(precision) && prec_ok(out_precision), "ns2_linear_split_as: bad arguments");
  WFMT(w, precision, "ns2_linear_split_as");
  ARGCHK(!w->geglu && !w->has_extra && (conv_taps == 0 ? w->taps == 1 : w->taps == conv_taps), "ns2_linear_split_as: weight packing does not match");
  ARGCHK(lda >= w->cols_p && (ldo & 1) == 0, "ns2_linear_split_as: bad leading dimensions");
  ARGCHK(pad_left >= -1 && pad_left < (conv_taps > 0 ? conv_taps : 1) && (act >= 0 && act <= 2), "ns2_linear_split_as: bad pad_left / act");
  ARGCHK(out_precision == 1 || (out_precision == 2) == (out_lo == nullptr), "ns2_linear_split_as: out_lo must be null for dense IEEE-half output, hi + 32 for interleaved lines");
  return gemm_split(w->w, a_hi, a_lo, lda, M, conv_taps, dilation, seq_len, bias, out_hi, out_lo, ldo, precision, (hipStream_t)stream,
                    pad_left, act, op_fmt(out_precision));
}
extern "C" int ns2_linear_geglu(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M,
                                const float* packed_bias, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream) {
  ARGCHK(w && a_hi && out_hi && packed_bias && prec_ok(precision) && w->geglu, "ns2_linear_geglu: bad arguments");
  WFMT(w, precision, "ns2_linear_geglu");
  ARGCHK(ldo * 2 == w->w.N, "ns2_linear_geglu: ldo must be round_up(f, 32)");
  return gemm_geglu(w->w, a_hi, a_lo, lda, M, packed_bias, out_hi, out_lo, ldo, precision, (hipStream_t)stream);
}
extern "C" int ns2_geglu_pack_bias(const float* bias, int f, float* packed, int packed_len, void* stream) {
  ARGCHK(bias && packed && f > 0, "ns2_geglu_pack_bias: bad arguments");
  const int rows_p = ((2 * ((f + 31) / 32 * 32)) + 255) / 256 * 256;
  ARGCHK(packed_len >= rows_p, "ns2_geglu_pack_bias: packed_len too small");
  std::vector<float> hb(2 * f), pb(packed_len, 0.f);
  HIPRET(hipMemcpy(hb.data(), bias, 2 * f * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<int> m = geglu_row_map(f, rows_p);
  for (int i = 0; i < rows_p; ++i)
    if (m[i] >= 0) pb[i] = hb[m[i]];
  HIPRET(hipMemcpy(packed, pb.data(), packed_len * sizeof(float), hipMemcpyHostToDevice));
  (void)stream;
  return NS2_OK;
}
extern "C" int ns2_linear_qkv(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int seq_len,
                              int split_col, uint16_t* out_hi, uint16_t* out_lo, int ldo, uint16_t* vt_hi, uint16_t* vt_lo,
                              int vt_ld, int precision, void* stream) {
  ARGCHK(w && a_hi && out_hi && vt_hi && prec_ok(precision), "ns2_linear_qkv: bad arguments");
  WFMT(w, precision, "ns2_linear_qkv");
  ARGCHK(seq_len > 0 && M % seq_len == 0 && (split_col % 32) == 0 && split_col < w->w.N && vt_ld >= seq_len && (vt_ld & 7) == 0,
         "ns2_linear_qkv: bad shapes");
  return gemm_qkv(w->w, a_hi, a_lo, lda, M, seq_len, split_col, out_hi, out_lo, ldo, vt_hi, vt_lo, vt_ld, precision, (hipStream_t)stream);
}
extern "C" int ns2_wavenet_block(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int seq_len,
                                 int dilation, const float* conv_bias, const float* res_bias, const float* film, int film_ld,
                                 uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream) {
  // precision 5 (this entry point only) = precision 4 operands, the dilated conv as one half product, res_conv with the correction terms
  const int p1_half = precision == 5;
  if (p1_half) precision = 4;
  ARGCHK(w && a_hi && out_hi && conv_bias && res_bias && film && prec_ok(precision), "ns2_wavenet_block: bad arguments");
  WFMT(w, precision, "ns2_wavenet_block");
  ARGCHK(w->taps == 3 && w->has_extra && seq_len > 0, "ns2_wavenet_block: weight must be packed with taps=3 and extra1x1");
  return gemm_wavenet(w->w, a_hi, a_lo, lda, 0, M, seq_len, dilation, 0, 1, conv_bias, res_bias, 0, film, film_ld, 0, out_hi, out_lo,
                      ldo, 0, ldo, precision, (hipStream_t)stream, p1_half);
}

extern "C" int ns2_attention_hd(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi,
                                const uint16_t* k_lo, int ldk, int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld,
                                uint16_t* o_hi, uint16_t* o_lo, int ldo, int B, int H, int Nq, int Nk, float scale,
                                const uint8_t* key_mask, int precision, int head_dim, void* stream) {
  ARGCHK(q_hi && k_hi && vt_hi && o_hi && prec_ok(precision), "ns2_attention: bad arguments");
  ARGCHK(head_dim == 32 || head_dim == 64 || head_dim == 128, "ns2_attention: head_dim must be 32, 64 or 128");
  AttnArgs a;
  a.D = head_dim;
  a.lse = nullptr;
  a.q_hi = q_hi; a.q_lo = q_lo; a.ldq = ldq; a.q_col0 = q_col0;
  a.k_hi = k_hi; a.k_lo = k_lo; a.ldk = ldk; a.k_col0 = k_col0;
  a.vt_hi = vt_hi; a.vt_lo = vt_lo; a.vt_ld = vt_ld;
  a.o_hi = o_hi; a.o_lo = o_lo; a.ldo = ldo; a.o_fmt = -1;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = scale; a.kmask = key_mask;
  HIPRET(launch_attention(a, precision, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_attention(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi,
                             const uint16_t* k_lo, int ldk, int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld,
                             uint16_t* o_hi, uint16_t* o_lo, int ldo, int B, int H, int Nq, int Nk, float scale,
                             const uint8_t* key_mask, int precision, void* stream) {
  return ns2_attention_hd(q_hi, q_lo, ldq, q_col0, k_hi, k_lo, ldk, k_col0, vt_hi, vt_lo, vt_ld, o_hi, o_lo, ldo, B, H, Nq, Nk, scale,
                          key_mask, precision, 64, stream);
}

extern "C" int ns2_rmsnorm(const float* x, int ldx, int M, int d, int seq_len, const float* gamma, const float* cond, int cond_ld,
                           uint16_t* out_hi, uint16_t* out_lo, int ldo, float* out_f32, int ldo_f, int precision, void* stream) {
  ARGCHK(x && (out_hi || out_f32) && prec_ok(precision), "ns2_rmsnorm: bad arguments");
  ARGCHK(precision != 2 || !out_lo, "ns2_rmsnorm: precision 2 (fp16) has no lo plane");
  ARGCHK(!cond || seq_len > 0, "ns2_rmsnorm: adaptive norm needs seq_len");
  NormArgs n;
  n.x = x; n.ldx = ldx; n.gamma = gamma; n.cond = cond; n.cond_ld = cond_ld;
  n.out_hi = out_hi; n.out_lo = out_lo; n.ldo = out_hi ? ldo : d; n.out_f = out_f32; n.ldo_f = ldo_f; n.fmt = op_fmt(precision);
  n.M = M; n.d = d; n.seq_len = seq_len;
  HIPRET(launch_rmsnorm(n, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int64_t ns2_skinny_linear_workspace_bytes(int B, int K, int J) { return (int64_t)skinny_linear_workspace_bytes(B, K, J); }
extern "C" int ns2_skinny_linear(const float* in, int ld_in, const float* wt, const float* bias, float* out, int ld_out, int B,
                                 int K, int J, int act, void* workspace, int64_t workspace_bytes, void* stream) {
  ARGCHK(in && wt && out, "ns2_skinny_linear: null pointer");
  HIPRET(launch_skinny_linear(in, ld_in, wt, bias, out, ld_out, B, K, J, act, (float*)workspace,
                              workspace ? (size_t)workspace_bytes : 0, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_time_embed(const float* times, const float* freqs, const float* wt, const float* bias, float* feat_ws,
                              float* out, int ld_out, int B, int dim, int dt, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  ARGCHK(times && freqs && wt && feat_ws && out, "ns2_time_embed: null pointer");
  HIPRET(launch_time_embed(times, freqs, wt, bias, feat_ws, out, ld_out, B, dim, dt, (float*)workspace,
                           workspace ? (size_t)workspace_bytes : 0, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_embedding(const int64_t* ids, const float* table, float* out, int64_t n, int dim, int64_t pad_id, void* stream) {
  ARGCHK(ids && table && out && n > 0 && dim > 0, "ns2_embedding: bad arguments");
  HIPRET(launch_embedding(ids, table, out, (long)n, dim, (long)pad_id, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_transpose_f32(const float* in, int batch, int R, int C, float* out, void* stream) {
  ARGCHK(in && out, "ns2_transpose_f32: null pointer");
  HIPRET(launch_transpose_f32(in, batch, R, C, out, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_ddim_step(const float* audio, const float* model_out, float* out, const float* times, const float* times_next,
                             int B, int64_t per_batch, int objective, int schedule, float scale, void* stream) {
  ARGCHK(audio && model_out && out && times && times_next, "ns2_ddim_step: null pointer");
  ARGCHK(objective >= 0 && objective <= 2 && schedule >= 0 && schedule <= 2, "ns2_ddim_step: bad objective/schedule");
  DdimArgs a;
  a.audio = const_cast<float*>(audio); a.model_out = model_out; a.out = out; a.times = times; a.times_next = times_next;
  a.B = B; a.per_batch = (long)per_batch; a.objective = objective; a.schedule = schedule; a.scale = scale;
  HIPRET(launch_ddim(a, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_cfg_mix(const float* cond_out, const float* null_out, float* out, int64_t n, float cond_scale, void* stream) {
  ARGCHK(cond_out && null_out && out, "ns2_cfg_mix: null pointer");
  HIPRET(launch_cfg_mix(cond_out, null_out, out, (long)n, cond_scale, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int ns2_seanet_prep(const float* x, int ldx, int in_prefix, const float* add, int ldadd, int B, int64_t T, int C, int elu,
                               int prefix, int im2col_k, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream) {
  ARGCHK(x && out_hi && prec_ok(precision), "ns2_seanet_prep: bad arguments");
  HIPRET(launch_seanet_prep(x, ldx, in_prefix, add, ldadd, B, (long)T, C, elu, prefix, im2col_k, out_hi, out_lo, ldo,
                            op_fmt(precision), (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_seanet_prep2(const float* x, int ldx, int in_prefix, int B, int64_t T, int C, int prefix, uint16_t* elu_hi,
                                uint16_t* elu_lo, int elu_ld, int elu_col0, int elu_cols, uint16_t* raw_hi, uint16_t* raw_lo, int raw_ld,
                                int raw_col0, int raw_cols, int precision, void* stream) {
  ARGCHK(x && (elu_hi || raw_hi) && prec_ok(precision), "ns2_seanet_prep2: bad arguments");
  HIPRET(launch_seanet_prep2(x, ldx, in_prefix, B, (long)T, C, prefix, elu_hi, elu_lo, elu_ld, elu_col0, elu_cols, raw_hi, raw_lo, raw_ld,
                             raw_col0, raw_cols, op_fmt(precision), (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_seanet_conv_narrow(const float* x, int64_t ldx, int in_prefix, int B, int64_t T, int ci, int co, int k, int elu,
                                      const float* w, const float* bias, float* out, int64_t ldo, void* stream) {
  ARGCHK(x && w && out && B > 0 && T > 0 && ci > 0 && co > 0 && k > 0 && in_prefix >= 0, "ns2_seanet_conv_narrow: bad arguments");
  const hipError_t e = launch_seanet_conv_narrow(x, (long)ldx, in_prefix, B, (long)T, ci, co, k, elu, w, bias, out, (long)ldo,
                                                 (hipStream_t)stream);
  if (e == hipErrorNotReady) return NS2_UNAVAILABLE;
  HIPRET(e);
  return NS2_OK;
}
extern "C" int ns2_seanet_resblock_narrow(const float* x, int64_t ldx, int in_prefix, int B, int64_t T, int C, const float* w1p,
                                          const float* b1, const float* w2p, const float* wsp, const float* b2s, float* out, int64_t ldo,
                                          void* stream) {
  ARGCHK(x && w1p && b1 && w2p && wsp && b2s && out && B > 0 && T > 0 && C > 0 && in_prefix >= 0, "ns2_seanet_resblock_narrow: bad arguments");
  const hipError_t e = launch_seanet_resblock_narrow(x, (long)ldx, in_prefix, B, (long)T, C, w1p, b1, w2p, wsp, b2s, out, (long)ldo,
                                                     (hipStream_t)stream);
  if (e == hipErrorNotReady) return NS2_UNAVAILABLE;
  HIPRET(e);
  return NS2_OK;
}
extern "C" int ns2_seanet_unpad(const float* src, int64_t ld_src, int prefix, float* dst, int64_t ld_dst, int B, int64_t T, int C,
                                void* stream) {
  ARGCHK(src && dst && B > 0 && T > 0 && C > 0 && prefix >= 0, "ns2_seanet_unpad: bad arguments");
  HIPRET(launch_seanet_unpad(src, (long)ld_src, prefix, dst, (long)ld_dst, B, (long)T, C, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int64_t ns2_lstm_state_floats(int B, int H) { return (B > 0 && H > 0) ? (int64_t)lstm_state_floats(B, H) : 0; }
extern "C" int ns2_lstm_layer(const float* xproj, int64_t ld_x, const float* w_hh, const float* b_hh, float* state, int64_t state_floats,
                              const float* resid, int64_t ld_r, float* out, int64_t ld_o, int B, int64_t T, int H, void* stream) {
  ARGCHK(xproj && w_hh && b_hh && state && out, "ns2_lstm_layer: null pointer");
  ARGCHK(B > 0 && T > 0 && H > 0 && H <= 512 && (H % 4) == 0, "ns2_lstm_layer: hidden size must be a multiple of 4, at most 512");
  ARGCHK(state_floats >= 3 * (int64_t)B * H, "ns2_lstm_layer: state scratch too small (ns2_lstm_state_floats)");
  float* h_a = state;                                  // caller-owned scratch (h ping, h pong, c | h exchange + step barrier)
  float* h_b = state + (size_t)B * H;
  float* c = state + 2 * (size_t)B * H;
  HIPRET(launch_lstm_layer(xproj, (long)ld_x, w_hh, b_hh, h_a, h_b, c, (long)state_floats, resid, (long)ld_r, out, (long)ld_o, B,
                           (long)T, H, (hipStream_t)stream));
  return NS2_OK;
}

extern "C" int64_t ns2_lstm2_state_floats(void) { return (int64_t)lstm2_state_floats(); }
extern "C" int ns2_lstm2(const float* xproj1, int64_t ld_x, const float* w_hh1, const float* b_hh1, const float* w_ih2, const float* b_ih2,
                         const float* w_hh2, const float* b_hh2, float* state, int64_t state_floats, const float* resid, int64_t ld_r,
                         float* out, int64_t ld_o, int B, int64_t T, void* stream) {
  ARGCHK(xproj1 && w_hh1 && b_hh1 && w_ih2 && b_ih2 && w_hh2 && b_hh2 && state && out, "ns2_lstm2: null pointer");
  ARGCHK(B > 0 && T > 0, "ns2_lstm2: empty batch or sequence");
  ARGCHK(state_floats >= (int64_t)lstm2_state_floats(), "ns2_lstm2: state scratch too small (ns2_lstm2_state_floats)");
  const hipError_t e = launch_lstm2(xproj1, (long)ld_x, w_hh1, b_hh1, w_ih2, b_ih2, w_hh2, b_hh2, state, (long)state_floats, resid, (long)ld_r,
                                    out, (long)ld_o, B, (long)T, (hipStream_t)stream);
  if (e == hipErrorNotReady) return NS2_UNAVAILABLE;
  HIPRET(e);
  return NS2_OK;
}

extern "C" int ns2_saturation_count(int reset, int64_t* count) {
  ARGCHK(count != nullptr, "ns2_saturation_count: null pointer");
  HIPRET(hipDeviceSynchronize());                    // a diagnostic read between sampling runs, never on a launch path
  const unsigned int parts[5] = {saturation_read_gemm(reset != 0), saturation_read_gemm2(reset != 0),
                                 saturation_read_attention(reset != 0), saturation_read_elementwise(reset != 0),
                                 saturation_read_backward(reset != 0)};
  int64_t tot = 0;
  for (unsigned int p : parts) {
    ARGCHK(p != ~0u, "ns2_saturation_count: could not read the device counter");
    tot += p;
  }
  *count = tot;
  return NS2_OK;
}

extern "C" int ns2_debug_lstm_inject_abort(int n) {
  ARGCHK(n >= 0, "ns2_debug_lstm_inject_abort: negative count");
  HIPRET(hipDeviceSynchronize());
  HIPRET(lstm_abort_inject((unsigned int)n));
  return NS2_OK;
}
extern "C" int ns2_lstm_abort_count(int reset, int64_t* count) {
  ARGCHK(count != nullptr, "ns2_lstm_abort_count: null pointer");
  HIPRET(hipDeviceSynchronize());
  const unsigned int v = lstm_abort_read(reset != 0);
  ARGCHK(v != ~0u, "ns2_lstm_abort_count: could not read the device counter");
  *count = v;
  return NS2_OK;
}

extern "C" int ns2_saturation_peek_async(unsigned int* host4, void* stream) {
  ARGCHK(host4 != nullptr, "ns2_saturation_peek_async: null pointer");
  hipStream_t s = (hipStream_t)stream;
  HIPRET(saturation_peek_gemm(host4 + 0, s));
  HIPRET(saturation_peek_gemm2(host4 + 1, s));
  HIPRET(saturation_peek_attention(host4 + 2, s));
  HIPRET(saturation_peek_elementwise(host4 + 3, s));
  return NS2_OK;
}

extern "C" int ns2_rvq_prepare(const float* codebooks, float* cb_norm, int Q, int C, int D, void* stream) {
  ARGCHK(codebooks && cb_norm, "ns2_rvq_prepare: null pointer");
  HIPRET(launch_rvq_prepare(codebooks, cb_norm, Q, C, D, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_rvq_encode(const float* x, const float* codebooks, const float* cb_norm, int64_t* codes, float* emb,
                              float* residual, int* near_tie_count, int M, int Q, int C, int D, float tie_eps, void* stream) {
  ARGCHK(x && codebooks && cb_norm && codes, "ns2_rvq_encode: null pointer");
  ARGCHK(D == 128 && C % 64 == 0, "ns2_rvq_encode: needs codebook_dim 128 and codebook_size % 64 == 0 (EnCodec: 128 / 1024)");
  RvqArgs a;
  a.x = x; a.codebooks = codebooks; a.cb_norm = cb_norm; a.codes = codes; a.emb = emb; a.residual = residual;
  a.near_tie_count = near_tie_count; a.M = M; a.Q = Q; a.C = C; a.D = D; a.tie_eps = tie_eps;
  HIPRET(launch_rvq_encode(a, (hipStream_t)stream));
  return NS2_OK;
}
extern "C" int ns2_rvq_decode(const int64_t* codes, const float* codebooks, float* emb, int M, int Q, int C, int D, void* stream) {
  ARGCHK(codes && codebooks && emb, "ns2_rvq_decode: null pointer");
  HIPRET(launch_rvq_decode(codes, codebooks, emb, M, Q, C, D, (hipStream_t)stream));
  return NS2_OK;
}
