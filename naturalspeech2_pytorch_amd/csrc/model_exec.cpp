// Host-side executor of the NaturalSpeech2 denoiser (`Model`, NS2:811-1000) for MI355X: owns the packed weights,
// carves the caller's workspace, and enqueues the whole forward as a fixed kernel sequence on the caller's stream
// (stream-ordered, no allocation, no synchronisation => capturable into a hipGraph by the caller).
//
// Layout decisions (MI355X-first, not the reference's):
//   * activations are token-major [B*N, C] everywhere (the reference flips to channel-first for the Wavenet,
//     NS2:972/997); causal convolutions become shifted-row GEMMs (gemm.hip);
//   * the 8 Wavenet columns of a stack (NS2:645-688) run as ONE batched launch (grid-z), reading/writing
//     column slices of a [M, 8*dim] buffer; the last stack's 8 skip convs + their sum are one K=8*dim GEMM;
//   * all time/prompt conditioning projections of a step (32 FiLM + 24/36 adaptive-norm Linears, NS2:623, 744)
//     are one weight-streaming skinny GEMM against a concatenated, K-major fp32 weight;
//   * everything that depends only on (prompt, cond) -- to_prompt_cond, the perceiver resampler, the per-layer
//     cross-attention K/V^T, cond_to_model_dim -- is computed once by prepare_cond() into a caller-owned state.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "ns2_host.h"

namespace ns2 {

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return NS2_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
#define NSCHK(expr)                  \
  do {                               \
    int _r = (expr);                 \
    if (_r != NS2_OK) return _r;     \
  } while (0)

// kernel categories for ns2_model_profile_* (== kernel symbols in a rocprofv3 trace)
enum { PC_GEMM_F32 = 0, PC_GEMM_SPLIT = 1, PC_GEMM_QKV = 2, PC_GEMM_GEGLU = 3, PC_GEMM_WAVENET = 4, PC_ATTENTION = 5, PC_NORM = 6,
       PC_GEMM_FFCONV = 7 /* the feed-forward causal conv alone (same kernel family as PC_GEMM_SPLIT) */ };

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t rup64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct Param { const float* p; std::vector<int64_t> dims; };

}  // namespace ns2

using namespace ns2;


struct ns2_model {
  ns2_model_config cfg;
  std::map<std::string, Param> params;
  std::vector<void*> owned;      // hipMalloc'ed blobs
  bool finalized = false;
  // derived
  int dim, a, f, fp, dp, dt, Tc, L, S, Lm, dpp, nnorm, Jtot;
  int fpc;                       // row length of the FF conv's INPUT planes: fp, or (plans whose conv is one half product on dense planes) the
                                 // 128-multiple the dedicated conv kernel wants (ffconv_kernel.h: 128-byte aligned rows, an even number of K tiles)
  // packed
  const float* freqs; float* wt_time; const float* b_time;
  float* wt_cond; float* b_cond;
  PackedW w_init; const float* b_init;
  std::vector<PackedW> w_wn;           // per stack, batched over z
  std::vector<float*> b_wn_conv, b_wn_res;
  PackedW w_skip; float* b_skip;
  PackedW w_final; const float* b_final;
  struct Layer {
    PackedW qkv, out, ffin, conv, ffout, cq, ckv, cout;
    float* b_ffin; const float* b_conv; const float* b_ffout;
  };
  std::vector<Layer> layers;
  const float* g_pred; PackedW w_pred;
  // conditional-only
  float* wt_prompt; const float* b_prompt; const float* null_prompt_cond; const float* null_prompt_tokens;
  const float* null_cond; PackedW w_cond2model; const float* b_cond2model;
  bool has_proj; PackedW w_proj; const float* b_proj; const float* latents;
  struct RLayer { PackedW q, kv, out, ffin, ffout; float* b_ffin; const float* b_ffout; };
  std::vector<RLayer> rlayers;
  const float* g_resampler;
  // sampled content checksum of the registered parameters (ns2_model_param_checksum)
  const float** d_param_ptrs = nullptr; long* d_param_numels = nullptr; float* d_param_sums = nullptr; int n_params = 0;
  // debug taps
  std::map<std::string, std::pair<float*, int64_t>> taps;
  // live kernel timing (bench.py roofline): HIP events around the launches of the selected kernel categories
  unsigned prof_mask = 0; size_t prof_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
};

namespace ns2 {

// ------------------------------------------------------------------------------------------------ packing helpers
static int dev_alloc(std::vector<void*>* owned, void** p, size_t bytes) {
  HIPCHK(hipMalloc(p, bytes ? bytes : 16));
  if (owned) owned->push_back(*p);
  return NS2_OK;
}

// Packing context: who owns the allocations and the layout of the weights being packed (ns2_common.h): interleaved
// [hi32|lo32] rows with a lo plane (exact models and ns2_weight_pack) or dense hi-only rows (precision-1 "fast" models).
struct PackCtx { std::vector<void*>* owned; bool il; int fmt; };   // il: interleaved 128-B lines (bf16 hi/lo or FMT_H8); fmt: PlaneFmt
// Model precision 5 ("hybrid") is a per-site plan on top of precision 4: every contraction keeps the fp8 correction terms
// except (a) the feed-forward causal conv (43 % of the FLOPs), which runs as ONE IEEE-half product: its input arrives as dense
// half planes from the GEGLU epilogue and it writes FMT_H8 lines for FF-out; (b) the dilated k = 3 convs of the Wavenet
// blocks (16 % of the FLOPs), which multiply the half parts of the same FMT_H8 operands as one product while the block's
// res_conv keeps the correction terms (gemm2.hip, P1).  tools/precision_study.py: these are the two sites whose rounding
// error reaches the output least (4.1e-5 -> 9.4e-5 for both at d128; every other site costs 1.6e-4 ... 3.5e-4).
// Model precision 6 ("hybrid_ff") extends the plan to the whole feed-forward branch: FF-in (+ GEGLU) and FF-out run as one IEEE-half
// product as well -- the adaptive RMSNorm in front of the branch writes dense half planes, the conv writes dense half planes for
// FF-out -- so the branch moves 2 bytes per operand element instead of 4 and spends 1 MFMA unit per FLOP instead of 2.  Every other
// site (q / k / v, the attention out-projection, the Wavenet's 1x1 convs, to_pred: the sites whose rounding reaches the output most)
// keeps the correction terms.  Error measured on the MI355X: tests/test_parity_r2_gpu.py sweeps, profiles/r04_parity.json.
static inline int op_precision(int model_precision) { return (model_precision == 5 || model_precision == 6) ? 4 : model_precision; }
static inline bool hybrid_plan(int model_precision) { return model_precision == 5 || model_precision == 6; }
static inline bool ff_half_plan(int model_precision) { return model_precision == 6; }
// The step-invariant conditioning (ns2_model_prepare_cond: perceiver resampler, cond_to_model_dim, per-layer cross-attention
// K / V) runs once per sampling run on a few rows, and every frame of every step consumes its 32 resampled tokens: their
// rounding is a systematic error of the whole run (measured on the conditioned d512/L12 model: tokens 2.1e-4 -> output
// 2.2e-4 at precision 4 against 4e-5 without conditioning).  So the IEEE-half modes compute it with the bf16 x3 products.
static inline int cond_precision(int op_prec) { return (op_prec == 2 || op_prec == 4) ? 3 : op_prec; }

static PackCtx pack_ctx_for(std::vector<void*>* owned, int precision) {
  // precision 3: interleaved bf16 hi/lo rows; 1: dense bf16 hi-only weights; 2: dense IEEE-half weights; 4: FMT_H8 lines
  return PackCtx{owned, precision == 3 || precision == 4, precision == 2 ? FMT_F16 : (precision == 4 ? FMT_H8 : FMT_BF16)};
}

// allocate a packed weight of rows_p x ldk logical columns (zero-filled)
static int alloc_packed(const PackCtx& pc, PackedW* w, int N, int ldk, int kt_per_tap) {
  std::vector<void*>* owned = pc.owned;
  const bool il = pc.il;
  w->N = N;
  w->rows_p = rup(N, 256);
  w->ldk = ldk;
  w->nkt = ldk / 32;
  w->kt_per_tap = kt_per_tap;
  size_t bytes = (size_t)w->rows_p * ldk * sizeof(bf16_t) * (il ? 2 : 1);
  NSCHK(dev_alloc(owned, (void**)&w->hi, bytes));
  w->lo = il ? w->hi + 32 : nullptr;
  w->fmt = pc.fmt;
  HIPCHK(hipMemset(w->hi, 0, bytes));
  return NS2_OK;
}

// pack src [R, C, T] into rows [row0, row0 + nrows_p) / columns [k_off, k_off + T*Cp) of w; row_map (host) gives the
// source row of each destination row (-1 = zero row)
static int pack_into(PackedW* w, const float* src, int C, int T, int Cp, const std::vector<int>& row_map, int row0, int k_off,
                     hipStream_t s) {
  int* d_map = nullptr;
  HIPCHK(hipMalloc((void**)&d_map, row_map.size() * sizeof(int)));
  HIPCHK(hipMemcpy(d_map, row_map.data(), row_map.size() * sizeof(int), hipMemcpyHostToDevice));
  const size_t roff = (size_t)row0 * w->ldk * (w->lo ? 2 : 1);
  hipError_t e = launch_pack_weight(src, C, T, Cp, d_map, (int)row_map.size(), w->hi + roff, w->lo ? w->lo + roff : nullptr,
                                    w->ldk, k_off, s, w->fmt);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d_map);
  HIPCHK(e);
  return NS2_OK;
}

static std::vector<int> identity_map(int R, int rows_p) {
  std::vector<int> m(rows_p, -1);
  for (int i = 0; i < R && i < rows_p; ++i) m[i] = i;
  return m;
}

// GEGLU row permutation: packed row g*64 + j -> x row g*32+j (j<32) | gate row f + g*32 + (j-32)
std::vector<int> geglu_row_map(int f, int rows_p) {
  std::vector<int> m(rows_p, -1);
  for (int rp = 0; rp < rows_p; ++rp) {
    int g = rp / 64, j = rp % 64;
    int fi = g * 32 + (j & 31);
    if (fi < f) m[rp] = (j < 32) ? fi : f + fi;
  }
  return m;
}

static int pack_linear(const PackCtx& pc, PackedW* w, const float* src, int R, int C, int T, hipStream_t s) {
  const int Cp = rup(C, 32);
  NSCHK(alloc_packed(pc, w, R, T * Cp, Cp / 32));
  return pack_into(w, src, C, T, Cp, identity_map(R, w->rows_p), 0, 0, s);
}

// k = 3 conv weights packed as dense IEEE half: (re)build the tiled LDS images the dedicated FF-conv kernel reads (ffconv_kernel.h) from the
// row-major pack -- a permutation of the packed values, so both kernels multiply the same operands
int build_conv3_tiles(std::vector<void*>* owned, PackedW* w, hipStream_t s) {
  if (w->fmt != FMT_F16 || w->lo || w->nkt != 3 * w->kt_per_tap) { set_error("build_conv3_tiles: not a dense IEEE-half k = 3 conv weight"); return NS2_ERR_ARG; }
  const int Cp = w->kt_per_tap * 32;
  if (!w->t3) NSCHK(dev_alloc(owned, (void**)&w->t3, ffconv3_tiled_bytes_of(w->N, Cp)));
  HIPCHK(ffconv3_build_tiles(w->hi, w->ldk, Cp, w->rows_p, w->N, w->t3, s));
  return NS2_OK;
}

// FMT_H8 linear weights: (re)build the tiled LDS images the lean mixed linear kernel reads (gemm3_kernel.h) -- a permutation of the packed bytes.
// Other formats, convolutions and K < 96: nothing to do (those products keep gemm2_kernel / gemm_kernel).
int build_lin_tiles(std::vector<void*>* owned, PackedW* w, hipStream_t s) {
  if (w->fmt != FMT_H8 || !w->lo || w->nkt != w->kt_per_tap || w->nkt < 3 || (w->rows_p & 255)) return NS2_OK;
  if (!w->tl) NSCHK(dev_alloc(owned, (void**)&w->tl, gemm3_tiled_bytes_of(w->rows_p, w->nkt)));
  HIPCHK(gemm3_build_tiles(w->hi, w->ldk, w->rows_p, w->tl, s));
  return NS2_OK;
}

static int pack_geglu(const PackCtx& pc, PackedW* w, const float* src, int f, int C, hipStream_t s) {
  const int Cp = rup(C, 32), fpad = rup(f, 32);
  NSCHK(alloc_packed(pc, w, 2 * fpad, Cp, Cp / 32));
  return pack_into(w, src, C, 1, Cp, geglu_row_map(f, w->rows_p), 0, 0, s);
}

static int pack_geglu_bias(std::vector<void*>* owned, float** out, const float* bias, int f, int rows_p) {
  std::vector<float> hb(2 * f), pb(rows_p, 0.f);
  HIPCHK(hipMemcpy(hb.data(), bias, 2 * f * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<int> m = geglu_row_map(f, rows_p);
  for (int i = 0; i < rows_p; ++i)
    if (m[i] >= 0) pb[i] = hb[m[i]];
  NSCHK(dev_alloc(owned, (void**)out, rows_p * sizeof(float)));
  HIPCHK(hipMemcpy(*out, pb.data(), rows_p * sizeof(float), hipMemcpyHostToDevice));
  return NS2_OK;
}

// op-level packing used by ns2_weight_pack (tests / non-Python hosts)
int pack_weight_public(const float* w, int rows, int cols, int taps, int geglu, const float* extra, int precision, PackedW* out,
                       std::vector<void*>* owned, hipStream_t s) {
  const int Cp = rup(cols, 32);
  // op-level weights: interleaved bf16 (serves precisions 1 and 3), dense IEEE half (2) or FMT_H8 lines (4)
  const PackCtx pc = pack_ctx_for(owned, precision == 1 ? 3 : precision);
  if (geglu) return pack_geglu(pc, out, w, rows / 2, cols, s);
  const int T = taps + (extra ? 1 : 0);
  NSCHK(alloc_packed(pc, out, rows, T * Cp, Cp / 32));
  NSCHK(pack_into(out, w, cols, taps, Cp, identity_map(rows, out->rows_p), 0, 0, s));
  if (extra) NSCHK(pack_into(out, extra, cols, 1, Cp, identity_map(rows, out->rows_p), 0, taps * Cp, s));
  return NS2_OK;
}

// ------------------------------------------------------------------------------------------------ GEMM call helpers
// Split-K scratch of the forward that is being enqueued on THIS host thread (ns2_kernels.h GemmArgs::sk_ws): a region of the
// caller's workspace, lent to every product of the pass by SplitKScope and gone when the entry point returns -- not library state.
static thread_local float* tl_sk_ws = nullptr;
struct SplitKScope {
  float* prev;
  explicit SplitKScope(float* ws) : prev(tl_sk_ws) { tl_sk_ws = ws; }
  ~SplitKScope() { tl_sk_ws = prev; }
  SplitKScope(const SplitKScope&) = delete;
  SplitKScope& operator=(const SplitKScope&) = delete;
};
extern "C" int64_t ns2_splitk_scratch_bytes(void) { return SPLITK_SCRATCH_FLOATS * (int64_t)sizeof(float); }
extern "C" int ns2_debug_splitk_plan(int M, int N, int k_tiles, int k_tiles_per_tap, int fp32_epilogue, int* slices, int* k_tiles_per_slice) {
  if (!slices || !k_tiles_per_slice) { set_error("ns2_debug_splitk_plan: null pointer"); return NS2_ERR_ARG; }
  splitk_plan(M, N, k_tiles, k_tiles_per_tap, fp32_epilogue != 0, SPLITK_SCRATCH_FLOATS, slices, k_tiles_per_slice);
  return NS2_OK;
}
extern "C" int ns2_debug_lend_splitk_scratch(void* scratch, int64_t bytes) {
  if (scratch && bytes < ns2_splitk_scratch_bytes()) { set_error("ns2_debug_lend_splitk_scratch: scratch smaller than ns2_splitk_scratch_bytes()"); return NS2_ERR_ARG; }
  tl_sk_ws = static_cast<float*>(scratch);
  return NS2_OK;
}
static GemmArgs base_args(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.sk_ws = tl_sk_ws; g.sk_ws_floats = tl_sk_ws ? SPLITK_SCRATCH_FLOATS : 0;
  g.a_hi = a_hi; g.a_lo = a_lo; g.lda = lda;
  g.w_hi = w.hi; g.w_lo = w.lo; g.ldw = w.ldk; g.w_t3 = w.t3; g.w_tl = w.tl; g.w_tw1 = w.tw1; g.w_tw2 = w.tw2;
  g.M = M; g.N = w.N; g.nkt = w.nkt; g.kt_per_tap = w.nkt; g.conv_taps = 0; g.dil = 1; g.mid_kt = -1;
  g.nz = 1; g.pad_left = -1; g.act = 0; g.out_fmt = -1; g.vt_fmt = -1;
  return g;
}
static void set_conv(GemmArgs& g, const PackedW& w, int taps, int dil, int seq_len) {
  g.kt_per_tap = w.kt_per_tap; g.conv_taps = taps; g.dil = dil; g.seq_len = seq_len;
}

int gemm_f32(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int conv_taps, int dil, int seq_len,
             const float* bias, const float* resid, int ldr, float* out, int ldo, int prec, hipStream_t s, int pad_left, int act) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  if (conv_taps) set_conv(g, w, conv_taps, dil, seq_len);
  g.pad_left = pad_left; g.act = act;
  g.epi = EPI_F32; g.bias = bias; g.resid = resid; g.ldr = ldr; g.out_f = out; g.ldo_f = ldo;
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}
// The residual stream's update followed by the RMSNorm that reads it (NS2:794-807): ONE launch where the GEMM's workgroups own whole
// rows (dim = 128: the 128 x 128 kernel, gemm.hip), else the GEMM and rmsnorm_kernel.  `fused` tells the caller what happened (profile
// categories).  cond_ld == 0 with cond != null: every utterance reads the same (gamma, beta) row (the time table).
int gemm_f32_norm(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, const float* bias, const float* resid, int ldr,
                  float* out, int ldo, int prec, int seq_len, const float* gamma, const float* cond, int cond_ld, bf16_t* n_hi, bf16_t* n_lo,
                  int n_ld, int n_fmt, hipStream_t s, bool* fused) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  g.epi = EPI_F32; g.bias = bias; g.resid = resid; g.ldr = ldr; g.out_f = out; g.ldo_f = ldo;
  *fused = gemm_fuses_norm(g, prec);
  if (*fused) {
    g.nrm_hi = n_hi; g.nrm_lo = n_lo; g.nrm_ld = n_ld; g.nrm_fmt = n_fmt;
    g.nrm_gamma = gamma; g.nrm_cond = cond; g.nrm_cond_ld = cond_ld; g.nrm_seq_len = seq_len;
  }
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}
int gemm_split(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int conv_taps, int dil, int seq_len,
               const float* bias, bf16_t* o_hi, bf16_t* o_lo, int ldo, int prec, hipStream_t s, int pad_left, int act, int out_fmt) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  if (conv_taps) set_conv(g, w, conv_taps, dil, seq_len);
  g.pad_left = pad_left; g.act = act; g.out_fmt = out_fmt;
  g.epi = EPI_SPLIT; g.bias = bias; g.out_hi = o_hi; g.out_lo = o_lo; g.ldo_s = ldo; g.out_ncols = ldo;
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}
int gemm_geglu(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, const float* pbias, bf16_t* o_hi,
               bf16_t* o_lo, int ldo, int prec, hipStream_t s, int out_fmt, int out_ncols) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  g.out_fmt = out_fmt;
  g.epi = EPI_GEGLU; g.bias = pbias; g.out_hi = o_hi; g.out_lo = o_lo; g.ldo_s = ldo; g.out_ncols = out_ncols > 0 ? out_ncols : ldo;
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}
int gemm_qkv(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, int M, int seq_len, int split_col,
             bf16_t* o_hi, bf16_t* o_lo, int ldo, bf16_t* vt_hi, bf16_t* vt_lo, int vt_ld, int prec, hipStream_t s, int att_fmt) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  g.epi = EPI_QKV; g.seq_len = seq_len; g.split_col = split_col;
  g.out_fmt = (prec == 2 || prec == 4) ? FMT_F16 : FMT_BF16;        // q / k are attention operands: IEEE half also at precision 4
  if (att_fmt >= 0) { g.out_fmt = att_fmt; g.vt_fmt = att_fmt; }   // ... unless the caller's attention runs in another format
  g.out_hi = o_hi; g.out_lo = o_lo; g.ldo_s = ldo; g.out_ncols = split_col;
  g.vt_hi = vt_hi; g.vt_lo = vt_lo; g.vt_ld = vt_ld; g.vt_rows = w.N - split_col;
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}
int gemm_wavenet(const PackedW& w, const bf16_t* a_hi, const bf16_t* a_lo, int lda, long a_zs, int M, int seq_len, int dil,
                 int dil_z, int nz, const float* b_conv, const float* b_res, long bias_zs, const float* film, int film_ld,
                 long film_zs, bf16_t* o_hi, bf16_t* o_lo, int ldo, long out_zs, int out_ncols, int prec, hipStream_t s, int p1_half) {
  GemmArgs g = base_args(w, a_hi, a_lo, lda, M);
  set_conv(g, w, 3, dil, seq_len);
  g.p1_half = (prec == 4) ? p1_half : 0;
  g.dil_z = dil_z; g.nz = nz; g.a_zs = a_zs; g.w_zs = (long)w.rows_p * w.ldk; g.bias_zs = bias_zs; g.film_zs = film_zs;
  g.out_zs = out_zs;
  g.mid_kt = 3 * w.kt_per_tap;
  g.epi = EPI_WAVENET; g.bias = b_conv; g.bias2 = b_res; g.film = film; g.film_ld = film_ld;
  g.out_hi = o_hi; g.out_lo = o_lo; g.ldo_s = ldo; g.out_ncols = out_ncols;
  HIPCHK(launch_gemm(g, prec, s));
  return NS2_OK;
}

}  // namespace ns2

// ================================================================================================ model

static const Param* find_param(const ns2_model* m, const std::string& k) {
  auto it = m->params.find(k);
  return it == m->params.end() ? nullptr : &it->second;
}
#define GETP(var, key)                                                              \
  const Param* var = find_param(m, key);                                            \
  if (!var) { set_error("missing parameter '%s'", std::string(key).c_str()); return NS2_ERR_STATE; }

extern "C" int ns2_model_create(const ns2_model_config* cfg, ns2_model** out) {
  if (!cfg || !out) { set_error("null argument"); return NS2_ERR_ARG; }
  if (cfg->dim_head != 32 && cfg->dim_head != 64 && cfg->dim_head != 128) { set_error("dim_head must be 32, 64 or 128 (the head dims the attention kernel is built for), got %d", cfg->dim_head); return NS2_ERR_ARG; }
  if (cfg->dim % 32) { set_error("dim must be a multiple of 32, got %d", cfg->dim); return NS2_ERR_ARG; }
  if (cfg->precision < 1 || cfg->precision > 6) { set_error("precision must be 1 (bf16), 2 (fp16), 3 (bf16 x3), 4 (fp16 + fp8 correction terms), 5 (4, with the FF causal conv and the Wavenet's dilated convs as one fp16 product) or 6 (5, with the whole feed-forward branch as fp16 products)"); return NS2_ERR_ARG; }
  if (cfg->wavenet_layers < 1 || cfg->wavenet_layers > 16 || cfg->wavenet_stacks < 1) { set_error("bad wavenet shape"); return NS2_ERR_ARG; }
  ns2_model* m = new ns2_model();
  m->cfg = *cfg;
  m->dim = cfg->dim; m->a = cfg->heads * cfg->dim_head;
  m->f = (int)((double)cfg->dim * cfg->ff_mult * 2 / 3);           // int(dim * mult * 2 / 3)  NS2:1010
  m->fp = rup(m->f, 32); m->dp = cfg->dim;
  m->fpc = (cfg->precision == 2 || hybrid_plan(cfg->precision)) ? ffconv3_lda(m->fp) : m->fp;
  m->dt = cfg->dim * cfg->dim_cond_mult;
  m->Tc = m->dt * (cfg->condition_on_prompt ? 2 : 1);              // NS2:884
  m->L = cfg->wavenet_layers; m->S = cfg->wavenet_stacks;
  m->Lm = cfg->num_latents_m; m->dpp = rup(cfg->dim_prompt > 0 ? cfg->dim_prompt : cfg->dim, 32);
  m->nnorm = cfg->condition_on_prompt ? 3 : 2;
  m->Jtot = (m->S * m->L + cfg->depth * m->nnorm) * 2 * m->dim;
  *out = m;
  return NS2_OK;
}

extern "C" int ns2_model_set_param(ns2_model* m, const char* name, const float* data, int ndim, const int64_t* dims) {
  if (!m || !name || !data) { set_error("null argument"); return NS2_ERR_ARG; }
  Param p; p.p = data; p.dims.assign(dims, dims + ndim);
  m->params[name] = p;
  return NS2_OK;
}

extern "C" void ns2_model_destroy(ns2_model* m) {
  if (!m) return;
  for (auto& e : m->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (void* p : m->owned) (void)hipFree(p);
  delete m;
}

static int transpose_weight(ns2_model* m, const Param* w, float** out, hipStream_t s) {   // [R, C] -> K-major [C, R]
  const int R = (int)w->dims[0], C = (int)w->dims[1];
  NSCHK(dev_alloc(&m->owned, (void**)out, (size_t)R * C * sizeof(float)));
  HIPCHK(launch_transpose_into(w->p, R, C, *out, R, 0, s));
  return NS2_OK;
}

extern "C" int ns2_model_finalize(ns2_model* m, void* stream) {
  if (!m) { set_error("null model"); return NS2_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int dim = m->dim, a = m->a, f = m->f, L = m->L, S = m->S;
  const bool cond = m->cfg.condition_on_prompt;
  char key[256];
  const PackCtx pc = pack_ctx_for(&m->owned, op_precision(m->cfg.precision));
  const PackCtx pc_conv = hybrid_plan(m->cfg.precision) ? pack_ctx_for(&m->owned, 2) : pc;
  const PackCtx pc_ff = ff_half_plan(m->cfg.precision) ? pack_ctx_for(&m->owned, 2) : pc;       // FF-in / FF-out weights
  const PackCtx pc_cond = pack_ctx_for(&m->owned, cond_precision(op_precision(m->cfg.precision)));   // weights of prepare_cond
  const bool il = pc.il;

  // ---- time conditioning (NS2:839-843)
  { GETP(fw, "to_time_cond.0.weights"); m->freqs = fw->p; }
  { GETP(w, "to_time_cond.1.weight"); NSCHK(transpose_weight(m, w, &m->wt_time, s)); }
  { GETP(b, "to_time_cond.1.bias"); m->b_time = b->p; }

  // ---- concatenated conditioning projections: K-major [Tc, Jtot]
  NSCHK(dev_alloc(&m->owned, (void**)&m->wt_cond, (size_t)m->Tc * m->Jtot * sizeof(float)));
  NSCHK(dev_alloc(&m->owned, (void**)&m->b_cond, (size_t)m->Jtot * sizeof(float)));
  auto add_cond = [&](const char* prefix, int slot) -> int {
    GETP(w, std::string(prefix) + ".weight");
    GETP(b, std::string(prefix) + ".bias");
    if (w->dims[0] != 2 * dim || w->dims[1] != m->Tc) { set_error("%s.weight has unexpected shape", prefix); return NS2_ERR_STATE; }
    HIPCHK(launch_transpose_into(w->p, 2 * dim, m->Tc, m->wt_cond, m->Jtot, (long)slot * 2 * dim, s));
    HIPCHK(hipMemcpyAsync(m->b_cond + (size_t)slot * 2 * dim, b->p, 2 * dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return NS2_OK;
  };
  for (int st = 0; st < S; ++st)
    for (int i = 0; i < L; ++i) {
      snprintf(key, sizeof key, "wavenet.stacks.%d.blocks.%d.to_time_cond", st, i);
      NSCHK(add_cond(key, st * L + i));
    }
  for (int l = 0; l < m->cfg.depth; ++l)
    for (int j = 0; j < m->nnorm; ++j) {
      const int idx = (j == 0) ? 0 : (m->nnorm == 3 ? (j == 1 ? 2 : 4) : 4);
      snprintf(key, sizeof key, "transformer.layers.%d.%d.to_gamma_beta", l, idx);
      NSCHK(add_cond(key, S * L + l * m->nnorm + j));
    }

  // ---- wavenet (NS2:690-725)
  { GETP(w, "wavenet.init_conv.weight"); GETP(b, "wavenet.init_conv.bias");
    NSCHK(pack_linear(pc, &m->w_init, w->p, dim, dim, (int)w->dims[2], s)); m->b_init = b->p; }
  m->w_wn.resize(S); m->b_wn_conv.resize(S); m->b_wn_res.resize(S);
  for (int st = 0; st < S; ++st) {
    PackedW& W = m->w_wn[st];
    // batched: L matrices of [rows_p, 4*dp] back to back
    W.N = dim; W.rows_p = rup(dim, 256); W.ldk = 4 * m->dp; W.nkt = W.ldk / 32; W.kt_per_tap = m->dp / 32;
    const size_t per = (size_t)W.rows_p * W.ldk * (il ? 2 : 1);     // physical elements per matrix
    NSCHK(dev_alloc(&m->owned, (void**)&W.hi, per * L * sizeof(bf16_t)));
    W.lo = il ? W.hi + 32 : nullptr;
    W.fmt = pc.fmt;
    HIPCHK(hipMemset(W.hi, 0, per * L * sizeof(bf16_t)));
    NSCHK(dev_alloc(&m->owned, (void**)&m->b_wn_conv[st], (size_t)L * dim * sizeof(float)));
    NSCHK(dev_alloc(&m->owned, (void**)&m->b_wn_res[st], (size_t)L * dim * sizeof(float)));
    for (int i = 0; i < L; ++i) {
      snprintf(key, sizeof key, "wavenet.stacks.%d.blocks.%d", st, i);
      GETP(cw, std::string(key) + ".conv.weight"); GETP(cb, std::string(key) + ".conv.bias");
      GETP(rw, std::string(key) + ".res_conv.weight"); GETP(rb, std::string(key) + ".res_conv.bias");
      PackedW view = W; view.hi = W.hi + per * i; view.lo = W.lo ? W.lo + per * i : nullptr;
      NSCHK(pack_into(&view, cw->p, dim, 3, m->dp, identity_map(dim, W.rows_p), 0, 0, s));
      NSCHK(pack_into(&view, rw->p, dim, 1, m->dp, identity_map(dim, W.rows_p), 0, 3 * m->dp, s));
      HIPCHK(hipMemcpyAsync(m->b_wn_conv[st] + (size_t)i * dim, cb->p, dim * sizeof(float), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(m->b_wn_res[st] + (size_t)i * dim, rb->p, dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  { // skip convs of the last stack, concatenated along K; summed bias (NS2:639-640, 685-686, 725)
    NSCHK(alloc_packed(pc, &m->w_skip, dim, L * m->dp, L * m->dp / 32));
    std::vector<float> bsum(dim, 0.f), hb(dim);
    for (int i = 0; i < L; ++i) {
      snprintf(key, sizeof key, "wavenet.stacks.%d.blocks.%d.skip_conv", S - 1, i);
      GETP(w, std::string(key) + ".weight"); GETP(b, std::string(key) + ".bias");
      NSCHK(pack_into(&m->w_skip, w->p, dim, 1, m->dp, identity_map(dim, m->w_skip.rows_p), 0, i * m->dp, s));
      HIPCHK(hipMemcpy(hb.data(), b->p, dim * sizeof(float), hipMemcpyDeviceToHost));
      for (int c = 0; c < dim; ++c) bsum[c] += hb[c];
    }
    NSCHK(dev_alloc(&m->owned, (void**)&m->b_skip, dim * sizeof(float)));
    HIPCHK(hipMemcpy(m->b_skip, bsum.data(), dim * sizeof(float), hipMemcpyHostToDevice));
  }
  { GETP(w, "wavenet.final_conv.weight"); GETP(b, "wavenet.final_conv.bias");
    NSCHK(pack_linear(pc, &m->w_final, w->p, dim, dim, 1, s)); m->b_final = b->p; }

  // ---- transformer (NS2:748-809)
  m->layers.resize(m->cfg.depth);
  for (int l = 0; l < m->cfg.depth; ++l) {
    ns2_model::Layer& ly = m->layers[l];
    snprintf(key, sizeof key, "transformer.layers.%d", l);
    const std::string p(key);
    { GETP(q, p + ".1.to_q.weight"); GETP(kv, p + ".1.to_kv.weight"); GETP(o, p + ".1.to_out.weight");
      NSCHK(alloc_packed(pc, &ly.qkv, 3 * a, m->dp, m->dp / 32));
      NSCHK(pack_into(&ly.qkv, q->p, dim, 1, m->dp, identity_map(a, a), 0, 0, s));
      NSCHK(pack_into(&ly.qkv, kv->p, dim, 1, m->dp, identity_map(2 * a, ly.qkv.rows_p - a), a, 0, s));
      NSCHK(pack_linear(pc, &ly.out, o->p, dim, a, 1, s)); }
    if (cond) {
      GETP(q, p + ".3.to_q.weight"); GETP(kv, p + ".3.to_kv.weight"); GETP(o, p + ".3.to_out.weight");
      NSCHK(pack_linear(pc, &ly.cq, q->p, a, dim, 1, s));
      NSCHK(pack_linear(pc_cond, &ly.ckv, kv->p, 2 * a, dim, 1, s));
      NSCHK(pack_linear(pc, &ly.cout, o->p, dim, a, 1, s));
    }
    { GETP(w1, p + ".5.0.weight"); GETP(b1, p + ".5.0.bias");
      GETP(cw, p + ".5.2.1.weight"); GETP(cb, p + ".5.2.1.bias");
      GETP(w2, p + ".5.3.weight"); GETP(b2, p + ".5.3.bias");
      if (w1->dims[0] != 2 * f) { set_error("FF inner dim mismatch: expected %d got %lld", 2 * f, (long long)w1->dims[0]); return NS2_ERR_STATE; }
      NSCHK(pack_geglu(pc_ff, &ly.ffin, w1->p, f, dim, s));
      NSCHK(pack_geglu_bias(&m->owned, &ly.b_ffin, b1->p, f, ly.ffin.rows_p));
      NSCHK(pack_linear(pc_conv, &ly.conv, cw->p, f, f, 3, s)); ly.b_conv = cb->p;
      if (pc_conv.fmt == FMT_F16) NSCHK(build_conv3_tiles(&m->owned, &ly.conv, s));
      NSCHK(pack_linear(pc_ff, &ly.ffout, w2->p, dim, f, 1, s)); ly.b_ffout = b2->p; }
  }
  { GETP(g, "transformer.to_pred.0.gamma"); GETP(w, "transformer.to_pred.1.weight");
    m->g_pred = g->p; NSCHK(pack_linear(pc, &m->w_pred, w->p, dim, dim, 1, s)); }

  // ---- prompt conditioning (NS2:849-881)
  if (cond) {
    const int dprompt = m->cfg.dim_prompt;
    { GETP(w, "to_prompt_cond.1.weight"); GETP(b, "to_prompt_cond.1.bias");
      NSCHK(transpose_weight(m, w, &m->wt_prompt, s)); m->b_prompt = b->p; }
    { GETP(p1, "null_prompt_cond"); m->null_prompt_cond = p1->p; }
    { GETP(p2, "null_prompt_tokens"); m->null_prompt_tokens = p2->p; }
    { GETP(p3, "null_cond"); m->null_cond = p3->p; }
    { GETP(w, "cond_to_model_dim.weight"); GETP(b, "cond_to_model_dim.bias");
      NSCHK(pack_linear(pc_cond, &m->w_cond2model, w->p, dim, dprompt, 1, s)); m->b_cond2model = b->p; }
    m->has_proj = find_param(m, "perceiver_resampler.proj_context.weight") != nullptr;
    if (m->has_proj) {
      GETP(w, "perceiver_resampler.proj_context.weight"); GETP(b, "perceiver_resampler.proj_context.bias");
      NSCHK(pack_linear(pc_cond, &m->w_proj, w->p, dim, dprompt, 1, s)); m->b_proj = b->p;
    } else if (dprompt != dim) { set_error("dim_prompt != dim but proj_context is missing"); return NS2_ERR_STATE; }
    { GETP(lt, "perceiver_resampler.latents"); m->latents = lt->p; }
    { GETP(g, "perceiver_resampler.norm.gamma"); m->g_resampler = g->p; }
    m->rlayers.resize(m->cfg.resampler_depth);
    for (int l = 0; l < m->cfg.resampler_depth; ++l) {
      ns2_model::RLayer& r = m->rlayers[l];
      snprintf(key, sizeof key, "perceiver_resampler.layers.%d", l);
      const std::string p(key);
      GETP(q, p + ".0.to_q.weight"); GETP(kv, p + ".0.to_kv.weight"); GETP(o, p + ".0.to_out.weight");
      GETP(w1, p + ".1.0.weight"); GETP(b1, p + ".1.0.bias"); GETP(w2, p + ".1.2.weight"); GETP(b2, p + ".1.2.bias");
      NSCHK(pack_linear(pc_cond, &r.q, q->p, a, dim, 1, s));
      NSCHK(pack_linear(pc_cond, &r.kv, kv->p, 2 * a, dim, 1, s));
      NSCHK(pack_linear(pc_cond, &r.out, o->p, dim, a, 1, s));
      NSCHK(pack_geglu(pc_cond, &r.ffin, w1->p, f, dim, s));
      NSCHK(pack_geglu_bias(&m->owned, &r.b_ffin, b1->p, f, r.ffin.rows_p));
      NSCHK(pack_linear(pc_cond, &r.ffout, w2->p, dim, f, 1, s)); r.b_ffout = b2->p;
    }
  }
  // tiled images of the Wavenet stacks for the lean block kernel of the hybrid plan (wavenet3_kernel.h): 20 MB per stack at d512
  if (hybrid_plan(m->cfg.precision) && (m->dp % 128) == 0 && (dim % 256) == 0)
    for (int st = 0; st < S; ++st) {
      PackedW& W = m->w_wn[st];
      if (W.fmt != FMT_H8) continue;
      NSCHK(dev_alloc(&m->owned, (void**)&W.tw1, wavenet3_tiles_bytes(W.rows_p, m->dp, L, 1)));
      NSCHK(dev_alloc(&m->owned, (void**)&W.tw2, wavenet3_tiles_bytes(W.rows_p, m->dp, L, 2)));
      HIPCHK(wavenet3_build_tiles(W.hi, W.rows_p, m->dp, L, W.tw1, W.tw2, s));
    }
  // tiled images of the per-step mixed linear weights (gemm3_kernel.h): +1 copy of 12.6 MB per layer at d512
  NSCHK(build_lin_tiles(&m->owned, &m->w_skip, s)); NSCHK(build_lin_tiles(&m->owned, &m->w_final, s)); NSCHK(build_lin_tiles(&m->owned, &m->w_pred, s));
  for (auto& ly : m->layers) {
    NSCHK(build_lin_tiles(&m->owned, &ly.qkv, s)); NSCHK(build_lin_tiles(&m->owned, &ly.out, s));
    NSCHK(build_lin_tiles(&m->owned, &ly.ffin, s)); NSCHK(build_lin_tiles(&m->owned, &ly.ffout, s));
    if (cond) { NSCHK(build_lin_tiles(&m->owned, &ly.cq, s)); NSCHK(build_lin_tiles(&m->owned, &ly.cout, s)); }
  }
  {   // device-side list of the registered parameter tensors (read in place: they alias the module's parameters)
    std::vector<const float*> ptrs; std::vector<long> numels;
    for (auto& kv : m->params) {
      long n = 1;
      for (int64_t d : kv.second.dims) n *= (long)d;
      if (n > 0) { ptrs.push_back(kv.second.p); numels.push_back(n); }
    }
    m->n_params = (int)ptrs.size();
    NSCHK(dev_alloc(&m->owned, (void**)&m->d_param_ptrs, ptrs.size() * sizeof(float*)));
    NSCHK(dev_alloc(&m->owned, (void**)&m->d_param_numels, numels.size() * sizeof(long)));
    NSCHK(dev_alloc(&m->owned, (void**)&m->d_param_sums, 2 * ptrs.size() * sizeof(float)));
    HIPCHK(hipMemcpy(m->d_param_ptrs, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->d_param_numels, numels.data(), numels.size() * sizeof(long), hipMemcpyHostToDevice));
  }
  HIPCHK(hipStreamSynchronize(s));
  m->finalized = true;
  return NS2_OK;
}

// Sampled content checksum of every registered parameter, stream-ordered and non-synchronising: 2 * ns2_model_param_count(m) floats
// are copied into host_out (pinned memory for a truly asynchronous copy; valid once the stream has passed this point).  The
// parameters are read where ns2_model_set_param found them, so writes through `.data` show up.
extern "C" int ns2_model_param_count(const ns2_model* m) { return (m && m->finalized) ? m->n_params : 0; }
extern "C" int ns2_model_param_checksum(ns2_model* m, float* host_out, int capacity, void* stream) {
  if (!m || !m->finalized || !host_out) { set_error("ns2_model_param_checksum: model not finalized / null output"); return NS2_ERR_STATE; }
  if (capacity < 2 * m->n_params) { set_error("ns2_model_param_checksum: output holds fewer than 2 * ns2_model_param_count floats"); return NS2_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(launch_param_sample(m->d_param_ptrs, m->d_param_numels, m->n_params, m->d_param_sums, s));
  HIPCHK(hipMemcpyAsync(host_out, m->d_param_sums, 2 * (size_t)m->n_params * sizeof(float), hipMemcpyDeviceToHost, s));
  return NS2_OK;
}

// ------------------------------------------------------------------------------------------------ workspace carving
namespace {
struct Carver {
  char* base; int64_t off = 0; int64_t cap;
  Carver(void* b, int64_t c) : base((char*)b), cap(c) {}
  template <typename T> T* take(int64_t n) {
    off = rup64(off, 256);
    T* p = (T*)(base ? base + off : nullptr);
    off += n * (int64_t)sizeof(T);
    return p;
  }
};
struct Planes { bf16_t* hi; bf16_t* lo; int fmt; };
// Activation formats of a precision: `op` = what the GEMMs multiply in (and write for each other), `att` = what the
// attention kernel reads (q, k, V^T): 3 -> interleaved bf16 hi/lo for both; 1 -> dense bf16; 2 -> dense IEEE half;
// 4 -> FMT_H8 lines for GEMM operands, dense IEEE half for the attention operands.
// `xatt` = the operands of the CROSS attention to the 32 resampled prompt tokens (NS2:799-803).  At precision 4 they are
// bf16 hi/lo planes and the product runs as bf16 x3: every frame attends to the same 32 keys, so their rounding is not
// averaged out over the sequence like self-attention's (tools/precision_study.py on the conditioned model: 1.4e-4 with
// half-product cross attention, 3.8e-5 with it exact); its cost is 6 % of the self-attention's.
struct Fmts { bool op_il; int op; bool att_il; int att; bool xatt_il; int xatt; int xatt_prec; };
static Fmts fmts_for(int precision) {
  switch (precision) {
    case 3: return Fmts{true, FMT_BF16, true, FMT_BF16, true, FMT_BF16, 3};
    case 2: return Fmts{false, FMT_F16, false, FMT_F16, false, FMT_F16, 2};
    case 4: return Fmts{true, FMT_H8, false, FMT_F16, true, FMT_BF16, 3};
    default: return Fmts{false, FMT_BF16, false, FMT_BF16, false, FMT_BF16, 1};
  }
}
// n logical elements; il: interleaved 128-B lines in one buffer (4 bytes per element), else one dense 16-bit plane
static Planes take_planes(Carver& c, int64_t n, bool il, int fmt) {
  Planes p;
  p.hi = c.take<bf16_t>(il ? 2 * n : n);
  p.lo = il ? p.hi + 32 : nullptr;
  p.fmt = fmt;
  return p;
}

struct Work {
  float *tfeat, *t, *condall, *xres, *tmp_f;
  float* skinny_ws; size_t skinny_ws_bytes;     // split-K partial sums of the conditioning projections (caller-owned)
  float* sk_ws;                                 // split-K slots of the small-batch GEMMs (SPLITK_SCRATCH_FLOATS, ns2_kernels.h)
  Planes xs, h0, wA, wB, ssum, xn, qk, vt, o, ffh, ffc;
  Planes xq;               // cross-attention queries [M, a] in the cross-attention operand format: a view of qk's memory
  Planes ffh_conv;         // the FF conv's input: ffh itself, or (precision 5) a dense IEEE-half view of the same memory
  int Nkp;
  // prepare_cond scratch
  float *pmean, *ctxf, *latf, *condT, *projf;
  Planes ctxp, latp, cpl, rkv, rvt;
  int Nctx, Nctxp;
};
static int64_t carve_work(const ns2_model* m, Work* w, void* base, int64_t cap, int B, int N, int n_prompt, int n_cond) {
  Carver c(base, cap);
  const int64_t M = (int64_t)B * N;
  // prepare_cond reuses qk / o / ffh for its B x Lm rows, possibly in a wider format (cond_precision): twice the rows cover it
  const int64_t Mq = (int64_t)B * std::max(N, m->cfg.condition_on_prompt ? 2 * m->Lm : 0);
  const int dim = m->dim, a = m->a, dp = m->dp, fp = m->fp, L = m->L;
  const Fmts F = fmts_for(op_precision(m->cfg.precision));
  const bool il = F.op_il;
  const int f16 = F.op;                               // (historical name) PlaneFmt of the GEMM operands
  const bool ail = F.att_il;
  const int afmt = F.att;
  const int kpad = ail ? 32 : 8;                      // key-axis padding of the transposed V planes
  w->tfeat = c.take<float>((int64_t)B * (dim + 1));
  w->t = c.take<float>((int64_t)B * m->Tc);
  w->condall = c.take<float>((int64_t)B * m->Jtot);
  w->skinny_ws_bytes = std::max(std::max(skinny_linear_workspace_bytes(B, m->Tc, m->Jtot), skinny_linear_workspace_bytes(B, m->dt, m->Jtot)),
                                std::max(skinny_linear_workspace_bytes(B, dim + 1, m->dt),
                                         skinny_linear_workspace_bytes(B, std::max(m->cfg.dim_prompt, 1), m->dt)));
  w->skinny_ws = c.take<float>((int64_t)(w->skinny_ws_bytes / sizeof(float)));
  w->sk_ws = c.take<float>(SPLITK_SCRATCH_FLOATS);
  w->xres = c.take<float>(M * dim);
  w->xs = take_planes(c, M * dp, il, f16);
  w->h0 = take_planes(c, M * dp, il, f16);
  w->wA = take_planes(c, M * L * dp, il, f16);
  w->wB = take_planes(c, M * L * dp, il, f16);
  w->ssum = take_planes(c, M * dp, il, f16);
  w->xn = take_planes(c, M * dp, il, f16);
  w->qk = take_planes(c, Mq * 2 * a, ail, afmt);
  w->xq = w->qk;                                      // M x a interleaved (4 B / element) fits M x 2a dense 16-bit
  w->xq.fmt = F.xatt; w->xq.lo = F.xatt_il ? w->xq.hi + 32 : nullptr;
  w->Nkp = rup(N, kpad);
  w->vt = take_planes(c, (int64_t)B * a * w->Nkp, ail, afmt);
  w->o = take_planes(c, Mq * a, il, f16);
  w->ffh = take_planes(c, Mq * std::max(fp, m->fpc), il, f16);      // (the conv's dense-half input view has rows of fpc elements)
  w->ffh_conv = w->ffh;
  if (hybrid_plan(m->cfg.precision)) { w->ffh_conv.lo = nullptr; w->ffh_conv.fmt = FMT_F16; }
  w->ffc = take_planes(c, M * fp, il, f16);
  if (m->cfg.condition_on_prompt && n_prompt > 0) {
    const int Lm = m->Lm;
    const Fmts CF = fmts_for(cond_precision(op_precision(m->cfg.precision)));     // formats of the prepare_cond pass
    const bool il = CF.op_il, ail = CF.att_il;
    const int f16 = CF.op, afmt = CF.att;
    w->Nctx = Lm + n_prompt; w->Nctxp = rup(w->Nctx, ail ? 32 : 8);
    w->pmean = c.take<float>((int64_t)B * m->cfg.dim_prompt);
    w->ctxf = c.take<float>((int64_t)B * w->Nctx * dim);
    w->latf = c.take<float>((int64_t)B * Lm * dim);
    w->projf = c.take<float>((int64_t)B * n_prompt * dim);
    w->condT = c.take<float>((int64_t)B * n_cond * m->cfg.dim_prompt);
    w->ctxp = take_planes(c, (int64_t)B * std::max(w->Nctx, std::max(n_prompt, n_cond)) * std::max(dp, m->dpp), il, f16);
    w->latp = take_planes(c, (int64_t)B * Lm * std::max(dp, std::max(fp, a)), il, f16);
    w->cpl = take_planes(c, (int64_t)B * Lm * dp, il, f16);
    w->rkv = take_planes(c, (int64_t)B * w->Nctx * a, ail, afmt);
    w->rvt = take_planes(c, (int64_t)B * a * w->Nctxp, ail, afmt);
  }
  return rup64(c.off, 256);
}

struct CondState {       // lives in the caller-owned cond_state blob
  float* prompt_cond;    // [B, dt]
  float* pbias;          // [B, Jtot]: W[:, dt:] prompt_cond[b] + bias of every conditioning projection (the step-invariant half of
                         // NS2:623 / 744 for a conditioned model; ns2_model_forward_row adds the hoisted time half)
  float* condadd;        // [B, n_cond, dim]
  std::vector<Planes> ck, cvt;   // per layer: K planes [B*Lm, a], V^T planes [B][a][Lmp]
  int n_cond_valid; int Lmp;
};
static int64_t carve_cond(const ns2_model* m, CondState* cs, void* base, int64_t cap, int B, int N, int n_cond) {
  Carver c(base, cap);
  c.take<int64_t>(4);                              // header: {magic, B, N, n_cond_valid}
  cs->prompt_cond = c.take<float>((int64_t)B * m->dt);
  cs->pbias = c.take<float>((int64_t)B * m->Jtot);
  cs->condadd = c.take<float>((int64_t)B * n_cond * m->dim);
  const Fmts F = fmts_for(op_precision(m->cfg.precision));
  const bool il = F.xatt_il;                        // the cached cross-attention keys / values are attention operands
  const int f16 = F.xatt;
  cs->Lmp = rup(m->Lm, il ? 32 : 8);
  cs->ck.resize(m->cfg.depth); cs->cvt.resize(m->cfg.depth);
  for (int l = 0; l < m->cfg.depth; ++l) {
    cs->ck[l] = take_planes(c, (int64_t)B * m->Lm * m->a, il, f16);
    cs->cvt[l] = take_planes(c, (int64_t)B * m->a * cs->Lmp, il, f16);
  }
  cs->n_cond_valid = std::min(n_cond, N);
  return rup64(c.off, 256);
}
}  // namespace

extern "C" int64_t ns2_model_workspace_bytes(const ns2_model* m, int B, int N, int n_prompt, int n_cond) {
  Work w;
  return carve_work(m, &w, nullptr, 0, B, N, n_prompt, n_cond);
}
extern "C" int64_t ns2_model_cond_bytes(const ns2_model* m, int B, int N, int n_prompt, int n_cond) {
  (void)n_prompt;
  CondState cs;
  return carve_cond(m, &cs, nullptr, 0, B, N, n_cond);
}

extern "C" int ns2_model_debug_tap(ns2_model* m, const char* name, float* dst, int64_t dst_elems) {
  if (!m || !name) return NS2_ERR_ARG;
  if (!dst) m->taps.erase(name);
  else m->taps[name] = std::make_pair(dst, dst_elems);
  return NS2_OK;
}

static int tap_f32(ns2_model* m, const char* name, const float* src, int64_t n, hipStream_t s) {
  auto it = m->taps.find(name);
  if (it == m->taps.end()) return NS2_OK;
  if (it->second.second < n) { set_error("tap '%s' needs %lld elements", name, (long long)n); return NS2_ERR_ARG; }
  HIPCHK(hipMemcpyAsync(it->second.first, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return NS2_OK;
}
static int tap_planes(ns2_model* m, const char* name, Planes p, int ld, int64_t M, int d, hipStream_t s) {
  auto it = m->taps.find(name);
  if (it == m->taps.end()) return NS2_OK;
  if (it->second.second < M * d) { set_error("tap '%s' needs %lld elements", name, (long long)(M * d)); return NS2_ERR_ARG; }
  HIPCHK(launch_join(p.hi, p.lo, ld, it->second.first, d, M, d, s, p.fmt));
  return NS2_OK;
}

static int attention_call(const bf16_t* q_hi, const bf16_t* q_lo, int ldq, int q_col0, const bf16_t* k_hi, const bf16_t* k_lo,
                          int ldk, int k_col0, Planes vt, int vt_ld, Planes o, int ldo, int B, int H, int Nq, int Nk, int prec,
                          hipStream_t s, int dim_head) {
  AttnArgs a;
  a.D = dim_head;
  a.lse = nullptr;
  a.q_hi = q_hi; a.q_lo = q_lo; a.ldq = ldq; a.q_col0 = q_col0;
  a.k_hi = k_hi; a.k_lo = k_lo; a.ldk = ldk; a.k_col0 = k_col0;
  a.vt_hi = vt.hi; a.vt_lo = vt.lo; a.vt_ld = vt_ld;
  a.o_hi = o.hi; a.o_lo = o.lo; a.ldo = ldo; a.o_fmt = o.fmt;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = 1.0f / sqrtf((float)dim_head);       // dim_head ** -0.5  (ATT:128 / SDPA default; 0.125 exactly at 64)
  a.kmask = nullptr;
  HIPCHK(launch_attention(a, prec, s));
  return NS2_OK;
}

static int norm_call(const float* x, int ldx, int M, int d, int seq_len, const float* gamma, const float* cond, int cond_ld,
                     Planes out, int ldo, float* out_f, int ldo_f, hipStream_t s) {
  NormArgs n;
  n.x = x; n.ldx = ldx; n.gamma = gamma; n.cond = cond; n.cond_ld = cond_ld;
  n.out_hi = out.hi; n.out_lo = out.lo; n.ldo = ldo; n.out_f = out_f; n.ldo_f = ldo_f;
  n.M = M; n.d = d; n.seq_len = seq_len; n.fmt = out.fmt;
  HIPCHK(launch_rmsnorm(n, s));
  return NS2_OK;
}

// ------------------------------------------------------------------------------------------------ prepare_cond
// Classifier-free guidance as ONE batch of 2 B utterances (NS2:914-927): every array of a cond_state is batch-major, so the
// state of [the B conditioned utterances | the same B with the null substitutes] is the two prepared states laid end to end per array.
// stream-ordered D2D copies, no allocation; done once per (prompt, cond) by the caller, like the states themselves.
extern "C" int ns2_model_cond_stack(ns2_model* m, const void* state_a, const void* state_b, int B, int N, int n_cond, void* state_out, void* stream) {
  if (!m || !m->finalized || !m->cfg.condition_on_prompt) { set_error("ns2_model_cond_stack: a finalized conditional model"); return NS2_ERR_STATE; }
  if (!state_a || !state_b || !state_out || B <= 0 || N <= 0 || n_cond <= 0) { set_error("ns2_model_cond_stack: bad arguments"); return NS2_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  CondState a, b, o;
  carve_cond(m, &a, const_cast<void*>(state_a), 0, B, N, n_cond);
  carve_cond(m, &b, const_cast<void*>(state_b), 0, B, N, n_cond);
  carve_cond(m, &o, state_out, 0, 2 * B, N, n_cond);
  auto put = [&](void* dst, const void* sa, const void* sb, size_t bytes) -> hipError_t {
    hipError_t e = hipMemcpyAsync(dst, sa, bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(static_cast<char*>(dst) + bytes, sb, bytes, hipMemcpyDeviceToDevice, s);
  };
  HIPCHK(put(o.prompt_cond, a.prompt_cond, b.prompt_cond, sizeof(float) * (size_t)B * m->dt));
  HIPCHK(put(o.pbias, a.pbias, b.pbias, sizeof(float) * (size_t)B * m->Jtot));
  HIPCHK(put(o.condadd, a.condadd, b.condadd, sizeof(float) * (size_t)B * n_cond * m->dim));
  const bool il = fmts_for(op_precision(m->cfg.precision)).xatt_il;
  const size_t eb = sizeof(bf16_t) * (il ? 2 : 1);
  for (int l = 0; l < m->cfg.depth; ++l) {
    HIPCHK(put(o.ck[l].hi, a.ck[l].hi, b.ck[l].hi, eb * (size_t)B * m->Lm * m->a));
    HIPCHK(put(o.cvt[l].hi, a.cvt[l].hi, b.cvt[l].hi, eb * (size_t)B * m->a * a.Lmp));
  }
  // (the 32-byte header slot of a state is reserved and unread)
  return NS2_OK;
}

extern "C" int ns2_model_prepare_cond(ns2_model* m, const float* prompt, int n_prompt, const float* cond, int n_cond, int drop,
                                      int B, int N, void* cond_state, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!m || !m->finalized) { set_error("model not finalized"); return NS2_ERR_STATE; }
  if (!m->cfg.condition_on_prompt) { set_error("model is unconditional"); return NS2_ERR_STATE; }
  if (!prompt || !cond || !cond_state || n_prompt <= 0 || n_cond <= 0) { set_error("prompt and cond are required"); return NS2_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  Work w;
  if (carve_work(m, &w, workspace, workspace_bytes, B, N, n_prompt, n_cond) > workspace_bytes) { set_error("workspace too small"); return NS2_ERR_ARG; }
  SplitKScope sk_scope(w.sk_ws);
  CondState cs;
  carve_cond(m, &cs, cond_state, 0, B, N, n_cond);
  const int dim = m->dim, a = m->a, dp = m->dp, fp = m->fp, Lm = m->Lm, H = m->cfg.heads;
  const int prec = cond_precision(op_precision(m->cfg.precision));     // the arithmetic of THIS pass (see cond_precision)
  const int dprompt = m->cfg.dim_prompt, dpp = m->dpp;
  {   // the buffers shared with the forward pass (qk, o, ffh), viewed in this pass's formats
    const Fmts CF = fmts_for(prec);
    auto retype = [](Planes& p, bool il, int fmt) { p.fmt = fmt; p.lo = il ? p.hi + 32 : nullptr; };
    retype(w.qk, CF.att_il, CF.att); retype(w.o, CF.op_il, CF.op); retype(w.ffh, CF.op_il, CF.op);
  }

  float* ctok = w.latf;      // resampled prompt tokens c [B*Lm, dim] fp32 (final norm output or null tokens)
  if (drop) {
    // NS2:954-958, 964-968, 982-986: null substitutes
    HIPCHK(launch_bcast_rows(m->null_prompt_cond, cs.prompt_cond, B, m->dt, m->dt, s));
    HIPCHK(launch_bcast_rows(m->null_prompt_tokens, ctok, B, (long)Lm * dim, (long)Lm * dim, s));
    HIPCHK(launch_split(ctok, dim, nullptr, 0, 0, 0, w.cpl.hi, w.cpl.lo, dp, B * Lm, dim, 0, s, w.cpl.fmt));
    HIPCHK(launch_bcast_rows(m->null_cond, cs.condadd, B * n_cond, dim, dim, s));
  } else {
    // to_prompt_cond: mean over n -> Linear -> SiLU (NS2:858-862)
    HIPCHK(launch_mean_rows(prompt, B, n_prompt, dprompt, w.pmean, s));
    HIPCHK(launch_skinny_linear(w.pmean, dprompt, m->wt_prompt, m->b_prompt, cs.prompt_cond, m->dt, B, dprompt, m->dt, 1, w.skinny_ws,
                                w.skinny_ws_bytes, s));
    // perceiver resampler (NS2:532-579)
    const int Nctx = w.Nctx;
    if (m->has_proj) {
      HIPCHK(launch_split(prompt, dprompt, nullptr, 0, 0, 0, w.ctxp.hi, w.ctxp.lo, dpp, B * n_prompt, dprompt, 0, s, w.ctxp.fmt));
      NSCHK(gemm_f32(m->w_proj, w.ctxp.hi, w.ctxp.lo, dpp, B * n_prompt, 0, 1, 0, m->b_proj, nullptr, 0, w.projf, dim, prec, s));
      HIPCHK(hipMemcpy2DAsync(w.ctxf + (size_t)Lm * dim, (size_t)Nctx * dim * 4, w.projf, (size_t)n_prompt * dim * 4,
                              (size_t)n_prompt * dim * 4, B, hipMemcpyDeviceToDevice, s));
    } else {
      HIPCHK(hipMemcpy2DAsync(w.ctxf + (size_t)Lm * dim, (size_t)Nctx * dim * 4, prompt, (size_t)n_prompt * dim * 4,
                              (size_t)n_prompt * dim * 4, B, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(launch_bcast_rows(m->latents, w.latf, B, (long)Lm * dim, (long)Lm * dim, s));
    for (size_t l = 0; l < m->rlayers.size(); ++l) {
      const ns2_model::RLayer& r = m->rlayers[l];
      // context = cat(latents, x)  (NS2:1060-1061, cross_attn_include_queries)
      HIPCHK(hipMemcpy2DAsync(w.ctxf, (size_t)Nctx * dim * 4, w.latf, (size_t)Lm * dim * 4, (size_t)Lm * dim * 4, B,
                              hipMemcpyDeviceToDevice, s));
      HIPCHK(launch_split(w.ctxf, dim, nullptr, 0, 0, 0, w.ctxp.hi, w.ctxp.lo, dp, B * Nctx, dim, 0, s, w.ctxp.fmt));
      HIPCHK(launch_split(w.latf, dim, nullptr, 0, 0, 0, w.latp.hi, w.latp.lo, dp, B * Lm, dim, 0, s, w.latp.fmt));
      NSCHK(gemm_split(r.q, w.latp.hi, w.latp.lo, dp, B * Lm, 0, 1, 0, nullptr, w.qk.hi, w.qk.lo, a, prec, s, -1, 0, w.qk.fmt));
      NSCHK(gemm_qkv(r.kv, w.ctxp.hi, w.ctxp.lo, dp, B * Nctx, Nctx, a, w.rkv.hi, w.rkv.lo, a, w.rvt.hi, w.rvt.lo, w.Nctxp, prec, s));
      NSCHK(attention_call(w.qk.hi, w.qk.lo, a, 0, w.rkv.hi, w.rkv.lo, a, 0, w.rvt, w.Nctxp, w.o, a, B, H, Lm, Nctx, prec, s, m->cfg.dim_head));
      NSCHK(gemm_f32(r.out, w.o.hi, w.o.lo, a, B * Lm, 0, 1, 0, nullptr, w.latf, dim, w.latf, dim, prec, s));
      // FeedForward without conv (NS2:1009-1025)
      HIPCHK(launch_split(w.latf, dim, nullptr, 0, 0, 0, w.latp.hi, w.latp.lo, dp, B * Lm, dim, 0, s, w.latp.fmt));
      NSCHK(gemm_geglu(r.ffin, w.latp.hi, w.latp.lo, dp, B * Lm, r.b_ffin, w.ffh.hi, w.ffh.lo, fp, prec, s));
      NSCHK(gemm_f32(r.ffout, w.ffh.hi, w.ffh.lo, fp, B * Lm, 0, 1, 0, r.b_ffout, w.latf, dim, w.latf, dim, prec, s));
    }
    // final RMSNorm (NS2:579): tokens in fp32 (tap) and as planes for the per-layer K/V projections
    NSCHK(norm_call(w.latf, dim, B * Lm, dim, 0, m->g_resampler, nullptr, 0, w.cpl, dp, w.latf, dim, s));
    // cond_to_model_dim: 1x1 conv over channel-first cond [B, dprompt, n_cond] (NS2:978)
    HIPCHK(launch_transpose_f32(cond, B, dprompt, n_cond, w.condT, s));
    HIPCHK(launch_split(w.condT, dprompt, nullptr, 0, 0, 0, w.ctxp.hi, w.ctxp.lo, dpp, B * n_cond, dprompt, 0, s, w.ctxp.fmt));
    NSCHK(gemm_f32(m->w_cond2model, w.ctxp.hi, w.ctxp.lo, dpp, B * n_cond, 0, 1, 0, m->b_cond2model, nullptr, 0, cs.condadd, dim, prec, s));
  }
  NSCHK(tap_f32(m, "c", ctok, (int64_t)B * Lm * dim, s));
  // the prompt half of every conditioning projection of the step: rows dt .. 2 dt of the K-major weight (t = cat(time, prompt_cond), NS2:960)
  HIPCHK(launch_skinny_linear(cs.prompt_cond, m->dt, m->wt_cond + (size_t)m->dt * m->Jtot, m->b_cond, cs.pbias, m->Jtot, B, m->dt, m->Jtot, 0,
                              w.skinny_ws, w.skinny_ws_bytes, s));
  // per-layer cross-attention keys / values of the (step-invariant) context (NS2:1063 with context = c)
  for (int l = 0; l < m->cfg.depth; ++l)
    NSCHK(gemm_qkv(m->layers[l].ckv, w.cpl.hi, w.cpl.lo, dp, B * Lm, Lm, a, cs.ck[l].hi, cs.ck[l].lo, a, cs.cvt[l].hi, cs.cvt[l].lo,
                   cs.Lmp, prec, s, cs.ck[l].fmt));
  return NS2_OK;
}

// ------------------------------------------------------------------------------------------------ live profiling
static int prof_start(ns2_model* m, int cat, hipStream_t s) {
  if (!(m->prof_mask & (1u << cat))) return NS2_OK;
  if (m->prof_used == m->prof_events.size()) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    m->prof_events.push_back(std::make_pair(a, b));
  }
  HIPCHK(hipEventRecord(m->prof_events[m->prof_used].first, s));
  return NS2_OK;
}
static int prof_stop(ns2_model* m, int cat, hipStream_t s) {
  if (!(m->prof_mask & (1u << cat))) return NS2_OK;
  HIPCHK(hipEventRecord(m->prof_events[m->prof_used].second, s));
  m->prof_used++;
  return NS2_OK;
}
#define PROF(cat, call)                 \
  do {                                  \
    NSCHK(prof_start(m, cat, s));       \
    NSCHK(call);                        \
    NSCHK(prof_stop(m, cat, s));        \
  } while (0)

extern "C" int ns2_model_profile_begin(ns2_model* m, unsigned category_mask) {
  if (!m) return NS2_ERR_ARG;
  m->prof_mask = category_mask;
  m->prof_used = 0;
  return NS2_OK;
}
extern "C" int ns2_model_profile_end(ns2_model* m, double* total_ms, int64_t* launches) {
  if (!m || !total_ms || !launches) return NS2_ERR_ARG;
  double tot = 0.0;
  for (size_t i = 0; i < m->prof_used; ++i) {
    HIPCHK(hipEventSynchronize(m->prof_events[i].second));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, m->prof_events[i].first, m->prof_events[i].second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)m->prof_used;
  m->prof_mask = 0;
  m->prof_used = 0;
  return NS2_OK;
}

// ------------------------------------------------------------------------------------------------ forward
// cond_row != null: the step's time conditioning comes from a row of the table ns2_model_time_table built ahead of the run
// (the sampler's times are known up front and shared by the batch, NS2:1303-1308): no projection is launched in the step
static int forward_impl(ns2_model* m, const float* x, const float* times, const float* cond_row, const void* cond_state, int n_cond, float* out,
                        int B, int N, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!m || !m->finalized) { set_error("model not finalized"); return NS2_ERR_STATE; }
  if (!x || (!times && !cond_row) || !out || B <= 0 || N <= 0) { set_error("bad forward arguments"); return NS2_ERR_ARG; }
  const bool cond = m->cfg.condition_on_prompt;
  if (cond != (cond_state != nullptr)) {
    set_error(cond ? "conditional model needs a cond_state (ns2_model_prepare_cond)" : "unconditional model got a cond_state");
    return NS2_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  Work w;
  if (carve_work(m, &w, workspace, workspace_bytes, B, N, 0, 0) > workspace_bytes) { set_error("workspace too small"); return NS2_ERR_ARG; }
  SplitKScope sk_scope(w.sk_ws);
  CondState cs;
  if (cond) carve_cond(m, &cs, const_cast<void*>(cond_state), 0, B, N, n_cond);
  const int dim = m->dim, a = m->a, dp = m->dp, fp = m->fp, L = m->L, S = m->S, H = m->cfg.heads, prec = op_precision(m->cfg.precision);
  const int M = B * N, Jtot = m->Jtot, Lm = m->Lm, fpc = m->fpc;
  const int conv_prec = hybrid_plan(m->cfg.precision) ? 2 : prec;
  const int ff_prec = ff_half_plan(m->cfg.precision) ? 2 : prec;
  Planes xn_ff = w.xn, ffc_ff = w.ffc;                 // precision 6: dense IEEE-half views of the same memory
  if (ff_half_plan(m->cfg.precision)) { xn_ff.lo = nullptr; xn_ff.fmt = FMT_F16; ffc_ff.lo = nullptr; ffc_ff.fmt = FMT_F16; }
  const int xprec = fmts_for(prec).xatt_prec;
  char name[64];

  // ---- t = to_time_cond(times) [, prompt_cond]  (NS2:944-960), then every conditioning projection of the step at once
  const float* call = w.condall;     // [gamma | beta] blocks of all FiLM / adaptive-norm projections; row b at call + b * cld
  int cld = Jtot;
  if (cond_row) {
    if (cond) HIPCHK(launch_add_row(cond_row, cs.pbias, w.condall, B, Jtot, s));     // time half (hoisted) + prompt half (per utterance)
    else { call = cond_row; cld = 0; }                                                // every utterance reads the same row
  } else {
    HIPCHK(launch_time_embed(times, m->freqs, m->wt_time, m->b_time, w.tfeat, w.t, m->Tc, B, dim, m->dt, w.skinny_ws, w.skinny_ws_bytes, s));
    if (cond)
      HIPCHK(hipMemcpy2DAsync(w.t + m->dt, (size_t)m->Tc * 4, cs.prompt_cond, (size_t)m->dt * 4, (size_t)m->dt * 4, B,
                              hipMemcpyDeviceToDevice, s));
    NSCHK(tap_f32(m, "t", w.t, (int64_t)B * m->Tc, s));
    HIPCHK(launch_skinny_linear(w.t, m->Tc, m->wt_cond, m->b_cond, w.condall, Jtot, B, m->Tc, Jtot, 0, w.skinny_ws, w.skinny_ws_bytes, s));
  }

  // ---- x (+ aligned conditioning, NS2:976-992) -> split planes
  HIPCHK(launch_split(x, dim, cond ? cs.condadd : nullptr, dim, n_cond, cond ? cs.n_cond_valid : 0, w.xs.hi, w.xs.lo, dp, M, dim, N, s, w.xs.fmt));

  // ---- wavenet (NS2:718-725)
  PROF(PC_GEMM_SPLIT, gemm_split(m->w_init, w.xs.hi, w.xs.lo, dp, M, 3, 1, N, m->b_init, w.h0.hi, w.h0.lo, dp, prec, s));
  NSCHK(tap_planes(m, "wavenet.init", w.h0, dp, M, dim, s));
  Planes cur = w.wA, prev = w.wB;
  for (int st = 0; st < S; ++st) {
    const bf16_t* a_hi = (st == 0) ? w.h0.hi : prev.hi;
    const bf16_t* a_lo = (st == 0) ? w.h0.lo : prev.lo;
    const int lda = (st == 0) ? dp : L * dp;
    const long a_zs = (st == 0) ? 0 : dp;
    PROF(PC_GEMM_WAVENET, gemm_wavenet(m->w_wn[st], a_hi, a_lo, lda, a_zs, M, N, /*dil=*/1, /*dil_z=*/1, /*nz=*/L, m->b_wn_conv[st], m->b_wn_res[st],
                       dim, call + (size_t)st * L * 2 * dim, cld, 2 * dim, cur.hi, cur.lo, L * dp, dp, dp, prec, s,
                       /*p1_half=*/hybrid_plan(m->cfg.precision) ? 1 : 0));
    snprintf(name, sizeof name, "wavenet.stack%d", st);
    NSCHK(tap_planes(m, name, cur, L * dp, M, L * dp, s));
    Planes tmp = prev; prev = cur; cur = tmp;
  }
  // sum of the 8 skip convs == one GEMM over the concatenated columns (NS2:639-640, 685-686, 725), then final_conv
  PROF(PC_GEMM_SPLIT, gemm_split(m->w_skip, prev.hi, prev.lo, L * dp, M, 0, 1, 0, m->b_skip, w.ssum.hi, w.ssum.lo, dp, prec, s));
  // Every update of the residual stream is followed by exactly one RMSNorm that reads it (NS2:794-807, 781-784): `update_then_norm` runs
  // the pair as one launch where the GEMM's workgroups own whole rows (dim = 128; gemm_f32_norm), else as the GEMM + rmsnorm_kernel.
  const float* cbase = call + (size_t)S * L * 2 * dim;
  auto update_then_norm = [&](const PackedW& pw, const Planes& a, int lda, const float* bias, bool add_resid, int gprec, const float* gamma,
                              const float* ncond, const Planes& nout) -> int {
    bool fused = false;
    PROF(PC_GEMM_F32, gemm_f32_norm(pw, a.hi, a.lo, lda, M, bias, add_resid ? w.xres : nullptr, dim, w.xres, dim, gprec, N, gamma, ncond, cld,
                                    nout.hi, nout.lo, dp, nout.fmt, s, &fused));
    if (!fused) PROF(PC_NORM, norm_call(w.xres, dim, M, dim, N, gamma, ncond, cld, nout, dp, nullptr, 0, s));
    return NS2_OK;
  };
  auto layer_cond = [&](int l, int which) { return cbase + (size_t)l * m->nnorm * 2 * dim + (size_t)which * 2 * dim; };
  // final_conv of the Wavenet -> the residual stream, normed for layer 0's self attention
  NSCHK(update_then_norm(m->w_final, w.ssum, dp, m->b_final, false, prec, nullptr, layer_cond(0, 0), w.xn));
  NSCHK(tap_f32(m, "wavenet.out", w.xres, (int64_t)M * dim, s));

  // ---- transformer (NS2:786-809)
  for (int l = 0; l < m->cfg.depth; ++l) {
    const ns2_model::Layer& ly = m->layers[l];
    const bool last = l + 1 == m->cfg.depth;
    // self attention (its norm ran behind the previous update)
    PROF(PC_GEMM_QKV, gemm_qkv(ly.qkv, w.xn.hi, w.xn.lo, dp, M, N, 2 * a, w.qk.hi, w.qk.lo, 2 * a, w.vt.hi, w.vt.lo, w.Nkp, prec, s));
    PROF(PC_ATTENTION, attention_call(w.qk.hi, w.qk.lo, 2 * a, 0, w.qk.hi, w.qk.lo, 2 * a, a, w.vt, w.Nkp, w.o, a, B, H, N, N, prec, s, m->cfg.dim_head));
    // out-projection + residual, then the norm of what follows: the cross attention (conditioned) or the feed-forward
    NSCHK(update_then_norm(ly.out, w.o, a, nullptr, true, prec, nullptr, layer_cond(l, cond ? 1 : m->nnorm - 1), cond ? w.xn : xn_ff));
    snprintf(name, sizeof name, "layer%d.attn", l);
    NSCHK(tap_f32(m, name, w.xres, (int64_t)M * dim, s));
    if (cond) {   // cross attention to the resampled prompt tokens (NS2:799-803)
      PROF(PC_GEMM_SPLIT, gemm_split(ly.cq, w.xn.hi, w.xn.lo, dp, M, 0, 1, 0, nullptr, w.xq.hi, w.xq.lo, a, prec, s, -1, 0, w.xq.fmt));
      PROF(PC_ATTENTION, attention_call(w.xq.hi, w.xq.lo, a, 0, cs.ck[l].hi, cs.ck[l].lo, a, 0, cs.cvt[l], cs.Lmp, w.o, a, B, H, N, Lm, xprec, s, m->cfg.dim_head));
      NSCHK(update_then_norm(ly.cout, w.o, a, nullptr, true, prec, nullptr, layer_cond(l, m->nnorm - 1), xn_ff));
    }
    // feedforward: Linear -> GEGLU -> causal conv k3 -> Linear (NS2:1009-1025); precision 6: the whole branch on dense IEEE-half planes
    PROF(PC_GEMM_GEGLU, gemm_geglu(ly.ffin, xn_ff.hi, xn_ff.lo, dp, M, ly.b_ffin, w.ffh_conv.hi, w.ffh_conv.lo, fpc, ff_prec, s, w.ffh_conv.fmt, fp));
    PROF(PC_GEMM_FFCONV, gemm_split(ly.conv, w.ffh_conv.hi, w.ffh_conv.lo, fpc, M, 3, 1, N, ly.b_conv, ffc_ff.hi, ffc_ff.lo, fp,
                                    conv_prec, s, -1, 0, ffc_ff.fmt));
    // FF-out + residual, then the next layer's self-attention norm -- or to_pred's RMSNorm (learned gamma, NS2:781-784) after the last
    NSCHK(update_then_norm(ly.ffout, ffc_ff, fp, ly.b_ffout, true, ff_prec, last ? m->g_pred : nullptr, last ? nullptr : layer_cond(l + 1, 0), w.xn));
    snprintf(name, sizeof name, "layer%d", l);
    NSCHK(tap_f32(m, name, w.xres, (int64_t)M * dim, s));
  }
  // to_pred: Linear on the normed stream (NS2:781-784)
  PROF(PC_GEMM_F32, gemm_f32(m->w_pred, w.xn.hi, w.xn.lo, dp, M, 0, 1, 0, nullptr, nullptr, 0, out, dim, prec, s));
  return NS2_OK;
}

extern "C" int ns2_model_forward(ns2_model* m, const float* x, const float* times, const void* cond_state, int n_cond, float* out,
                                 int B, int N, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!times) { set_error("bad forward arguments"); return NS2_ERR_ARG; }
  return forward_impl(m, x, times, nullptr, cond_state, n_cond, out, B, N, workspace, workspace_bytes, stream);
}
extern "C" int ns2_model_forward_row(ns2_model* m, const float* x, const float* cond_row, const void* cond_state, int n_cond, float* out,
                                     int B, int N, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!cond_row) { set_error("ns2_model_forward_row: null conditioning row"); return NS2_ERR_ARG; }
  return forward_impl(m, x, nullptr, cond_row, cond_state, n_cond, out, B, N, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ hoisted time conditioning
// "all time-conditioning projections" are step-invariant work once the schedule is known: the sampler's times are
// linspace(1, 0, T + 1) and identical across the batch (NS2:1303-1308), so the 56-68 Linears of NS2:623 / 744 for the WHOLE run are
// this one table [T, Jtot] (229 MB at T = 1000, d512/L12) and a step reads row i.  Rows are produced 32 at a time by the very
// kernels a step would launch, with the K split of a batch of `plan_B` rows (the run's batch size): every row is bit-identical
// to what the step's own launches compute for a batch of that size.  Unconditional model: row = the complete projections (with
// bias); conditioned: the time half W[:, :dt] t (the prompt half + bias sits in the cond_state, ns2_model_prepare_cond).
extern "C" int ns2_model_table_cols(const ns2_model* m) { return m ? m->Jtot : 0; }
extern "C" int64_t ns2_model_time_table_workspace_bytes(const ns2_model* m, int plan_B) {
  if (!m) return 0;
  const int kt = m->cfg.condition_on_prompt ? m->dt : m->Tc;
  int64_t n = rup64((int64_t)32 * (m->dim + 1) * 4, 256) + rup64((int64_t)32 * m->dt * 4, 256);
  n += rup64((int64_t)std::max(skinny_linear_workspace_bytes(32, kt, m->Jtot, plan_B), skinny_linear_workspace_bytes(32, m->dim + 1, m->dt, plan_B)), 256);
  return n + 256;
}
extern "C" int ns2_model_time_table(ns2_model* m, const float* times, int T, int plan_B, float* table, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  if (!m || !m->finalized) { set_error("model not finalized"); return NS2_ERR_STATE; }
  if (!times || !table || !workspace || T <= 0 || plan_B <= 0) { set_error("ns2_model_time_table: bad arguments"); return NS2_ERR_ARG; }
  if (workspace_bytes < ns2_model_time_table_workspace_bytes(m, plan_B)) { set_error("ns2_model_time_table: workspace too small"); return NS2_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const bool cond = m->cfg.condition_on_prompt;
  const int kt = cond ? m->dt : m->Tc;               // unconditional: Tc == dt
  Carver c(workspace, workspace_bytes);
  float* feat = c.take<float>((int64_t)32 * (m->dim + 1));
  float* tt = c.take<float>((int64_t)32 * m->dt);
  const size_t wsb = std::max(skinny_linear_workspace_bytes(32, kt, m->Jtot, plan_B), skinny_linear_workspace_bytes(32, m->dim + 1, m->dt, plan_B));
  float* ws = c.take<float>((int64_t)(wsb / sizeof(float)));
  for (int t0 = 0; t0 < T; t0 += 32) {
    const int nb = std::min(32, T - t0);
    HIPCHK(launch_time_embed(times + t0, m->freqs, m->wt_time, m->b_time, feat, tt, m->dt, nb, m->dim, m->dt, ws, wsb, s, plan_B));
    HIPCHK(launch_skinny_linear(tt, m->dt, m->wt_cond, cond ? nullptr : m->b_cond, table + (size_t)t0 * m->Jtot, m->Jtot, nb, kt, m->Jtot, 0, ws,
                                wsb, s, plan_B));
  }
  return NS2_OK;
}
