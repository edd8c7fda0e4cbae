// Internal (C++) launcher interface of libns2hip: one launcher per HIP kernel family.
// The public C ABI is include/ns2hip.h; csrc/capi.cpp and csrc/model_exec.cpp sit on top of these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "ns2_fmt.h"

typedef uint16_t bf16_t;

namespace ns2 {

// Per-kernel "dynamic LDS size raised" flag, kept PER DEVICE (function attributes belong to the device's code object) and
// safe under concurrent first calls from several host threads (one thread per device is the supported model; a duplicated
// hipFuncSetAttribute is harmless).  This is the only host-side state a launcher keeps; it never allocates.
struct DynLdsAttr {
  static constexpr int kMaxDev = 64;
  std::atomic<int> bytes[kMaxDev];
  DynLdsAttr() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
  hipError_t ensure(const void* fn, int lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < kMaxDev;
    if (tracked && bytes[dev].load(std::memory_order_acquire) >= lds) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess && tracked) bytes[dev].store(lds, std::memory_order_release);
    return e;
  }
};

// split-plane operands: a lo plane implies the interleaved [hi32|lo32] row layout, i.e. lo == hi + 32 (ns2_common.h)
inline bool planes_ok(const bf16_t* hi, const bf16_t* lo) { return lo == nullptr || lo == hi + 32; }

enum GemmEpilogue : int {
  EPI_F32 = 0,      // out_f = acc + bias (+ resid)
  EPI_SPLIT = 1,    // split planes = acc + bias
  EPI_QKV = 2,      // columns < split_col -> split planes, columns >= split_col -> transposed planes V^T[b][feat][n]
  EPI_GEGLU = 3,    // packed [x|gate] wave tiles -> split planes gelu(gate)*x
  EPI_WAVENET = 4,  // mid-loop FiLM + tanh*sigmoid gate, second K phase = res conv, split planes out
};

struct GemmArgs {
  // A operand: activations, bf16 split planes [M, lda]; every ld / column count in this struct is LOGICAL (the
  // physical row stride is 2*ld when the lo plane exists, see ns2_common.h); a_zs / out_zs are logical column offsets
  const bf16_t* a_hi; const bf16_t* a_lo; int lda;
  // W operand: packed weights, bf16 split planes [ceil(N/128)*128, ldw], K contiguous
  const bf16_t* w_hi; const bf16_t* w_lo; int ldw;
  const bf16_t* w_tl;       // optional: the same FMT_H8 linear weight as tiled LDS images (gemm3_kernel.h); null = none
  const bf16_t* w_tw1; const bf16_t* w_tw2;   // optional (EPI_WAVENET, hybrid plan): tiled images of the dilated conv's half parts / of the res conv (wavenet3_kernel.h)
  const bf16_t* w_t3;       // optional: the same conv weight (k = 3, dense IEEE half) as tiled LDS images (ffconv_kernel.h); null = none
  int M, N;          // N = valid output columns
  int nkt;           // number of 32-wide K tiles (total, all taps/phases)
  int kt_per_tap;    // K tiles per tap (plain linear: == nkt)
  int conv_taps;     // taps [0, conv_taps) are causal-shifted by (conv_taps-1-tap)*dil rows; later taps unshifted
  int dil; int dil_z;// dilation; if dil_z the effective dilation is dil << blockIdx-z
  int seq_len;       // tokens per utterance (rows never read across an utterance start); 0 = no sequence structure
  int mid_kt;        // EPI_WAVENET: K tile index at which the gate transform runs
  int p1_half;       // EPI_WAVENET at precision 4: the taps before mid_kt run as ONE half product (no fp8 correction terms, the byte half of the lines is not fetched)
  int epi;
  // epilogue operands
  const float* bias; const float* bias2;
  const float* film; int film_ld;           // gamma at film[b*film_ld + col], beta at film[b*film_ld + N + col]
  const float* resid; int ldr;
  float* out_f; int ldo_f;
  long out_f_zs;     // EPI_F32 with nz > 1: slice z writes out_f + z * out_f_zs (split-K partial sums of the weight gradients)
  bf16_t* out_hi; bf16_t* out_lo; int ldo_s; int out_ncols;   // writes columns [0, out_ncols) (zero beyond N)
  bf16_t* vt_hi; bf16_t* vt_lo; int vt_ld; int vt_rows; int split_col;
  // batching over blockIdx-z (wavenet columns): element offsets per z
  int nz; long a_zs, w_zs, bias_zs, film_zs, out_zs;
  int pad_left;      // conv: zero rows in front of the sequence (-1 = causal: conv_taps-1); k=9 'same' padding = 4
  int act;           // 1 = SiLU after the bias (EPI_F32 / EPI_SPLIT)
  int out_fmt;       // PlaneFmt of the split-plane output (ns2_common.h).  -1 = "the operand format of the precision":
                     // bf16 planes for 1 / 3, dense IEEE half for 2, FMT_H8 for 4.  Attention operands (q, k: EPI_QKV
                     // columns < split_col, the cross-attention q projection) are IEEE half also at precision 4.
  int vt_fmt;        // PlaneFmt of the transposed value planes (FMT_BF16 or FMT_F16); -1 = the attention format of the precision
  // Split-K for products too small to fill the chip (a batch of 1 ... 4 utterances: 8 ... 100 output tiles on 256 CUs, each a
  // serial K loop).  The caller lends scratch (sk_ws: fp32, sk_ws_floats elements; null = never split); launch_gemm then runs
  // the 128x128 kernel over grid-z K slices writing raw fp32 partial sums into fixed slots and a second launch that adds the
  // slots in slot order and applies the requested epilogue -- deterministic.  ksplit is set by launch_gemm: K tiles of EVERY tap
  // one slice covers (slice z: input columns [z * ksplit * 32, ...) of each tap); 0 = the plain launch.
  float* sk_ws; long sk_ws_floats;
  int ksplit;
  // EPI_F32 on the 128 x 128 kernel with N == 128 (the dim = 128 model: a workgroup owns WHOLE rows): the RMSNorm that follows the
  // residual stream's update (NS2:727-746, 794-807) inside the same epilogue -- out_f = acc + bias + resid as always, then
  // nrm planes = F.normalize(out_f row) * sqrt(N) [* nrm_gamma] [* cond_g + cond_b] in format nrm_fmt, the next GEMM's operand.
  // nrm_hi == null: not requested.  gemm_fuses_norm() says whether launch_gemm will honour it for a given shape.
  bf16_t* nrm_hi; bf16_t* nrm_lo; int nrm_ld, nrm_fmt;
  const float* nrm_gamma; const float* nrm_cond; int nrm_cond_ld, nrm_seq_len;
  // launch_gemm_tr only (weight gradients from token-major planes, gemm2.hip TR): tokens the operands have, channels per tap block of
  // the N dimension (a multiple of 32), taps (0 / 1: no shift; tap t of a causal conv reads token m - (tr_taps - 1 - t) * dil)
  int tr_tokens, tr_kp, tr_taps;
};

// precision: 3 = bf16 x3 ("exact"), 1 = bf16 ("fast"), 2 = one IEEE-half product ("half"), 4 = half product + both
// first-order correction terms on the fp8 MFMA ("mixed", FMT_H8 operands).  Dispatches gemm.hip / gemm2.hip by shape.
hipError_t launch_gemm(const GemmArgs& g, int precision, hipStream_t s);
bool gemm_fuses_norm(const GemmArgs& g, int precision);                 // gemm2.hip: the shape takes the 128 x 128 kernel's whole-row epilogue (no split-K)
hipError_t launch_gemm_tr(const GemmArgs& g, int precision, hipStream_t s);   // gemm2.hip: both operands read transposed (precision 3 / 4)
void splitk_plan(int M, int N, int nkt, int kt_per_tap, bool epi_f32, long scratch_floats, int* S, int* c);   // gemm2.hip; S = 1: no split
void force_gemm_kernel(int k);                                          // 0 auto, 1 = 128x128, 2 = 256x256, 3 = auto without split-K, 4 = auto without the dedicated FF-conv kernel, 5 = auto with it whatever the size (test hook); a forced kernel never splits K
// the dedicated FF causal conv kernel (ffconv_kernel.h, compiled in gemm2.hip): tiled weight images
size_t ffconv3_tiled_bytes_of(int N, int Cp);
hipError_t ffconv3_build_tiles(const bf16_t* w_hi, int ldw, int Cp, int rows_p, int N, bf16_t* out, hipStream_t s);
int ffconv3_lda(int Cp);
// the lean mixed linear kernel (gemm3_kernel.h, compiled in gemm2.hip): tiled weight images of an FMT_H8 pack [rows_p][ldk]
size_t gemm3_tiled_bytes_of(int rows_p, int nkt);
hipError_t gemm3_build_tiles(const bf16_t* w_hi, int ldk, int rows_p, bf16_t* out, hipStream_t s);
// the lean Wavenet block kernel of the hybrid plan (wavenet3_kernel.h): tiled images of a stack's nz matrices [rows_p][4 dp]
size_t wavenet3_tiles_bytes(int rows_p, int dp, int nz, int phase);
hipError_t wavenet3_build_tiles(const bf16_t* w_hi, int rows_p, int dp, int nz, bf16_t* t1, bf16_t* t2, hipStream_t s);                                                // activations' row length (elements) the kernel wants for Cp packed columns per tap
constexpr long SPLITK_SCRATCH_FLOATS = 512L * 128 * 128;                // what any split needs at most: slices x output tiles <= 512 tiles of 128 x 128 (32 MiB)

// flash attention forward, head dim 64, non-causal (ATT:77-155 hot path)
struct AttnArgs {
  const bf16_t* q_hi; const bf16_t* q_lo; int ldq;     // [B*Nq, ldq], head h at columns q_col0 + 64h
  const bf16_t* k_hi; const bf16_t* k_lo; int ldk;     // [B*Nk, ldk], head h at columns k_col0 + 64h
  const bf16_t* vt_hi; const bf16_t* vt_lo; int vt_ld; // [B][H*64][vt_ld] transposed values
  bf16_t* o_hi; bf16_t* o_lo; int ldo;                 // [B*Nq, ldo], head h at columns 64h
  int o_fmt;                                           // PlaneFmt of o (-1: the operand format; FMT_H8 feeds a precision-4 GEMM)
  int q_col0, k_col0;
  int B, H, Nq, Nk;
  int D;                                               // head dimension: 32, 64 or 128 (0 = 64); head h at columns col0 + D h, V^T rows H D per utterance
  float scale;
  const unsigned char* kmask;                            // optional key-padding mask [B, Nk], 1 = attend (ATT:92-94, 136-138)
  float* lse;                                            // optional [B, H, Nq]: log2 of the softmax denominator of the SCALED scores
                                                         // (m + log2 l), what the backward kernels recompute P from; null = not wanted
};                                                       // precision: 3 bf16x3, 1 bf16, 2 and 4: one IEEE-half product
hipError_t launch_attention(const AttnArgs& a, int precision, hipStream_t s);

// RMSNorm (NS2:727-746): out = x / max(|x|, 1e-12) * sqrt(d) [* gamma] [* g_c + b_c]  -> split planes
struct NormArgs {
  const float* x; int ldx;           // [M, d]
  const float* gamma;                // [d] or null
  const float* cond; int cond_ld;    // adaptive: g_c = cond[b*cond_ld + c], b_c = cond[b*cond_ld + d + c]; null = plain
  bf16_t* out_hi; bf16_t* out_lo; int ldo;
  float* out_f; int ldo_f;           // optional fp32 copy (e.g. resampler output)
  int M, d, seq_len;
  int fmt;                           // PlaneFmt of the output planes
};
hipError_t launch_rmsnorm(const NormArgs& a, hipStream_t s);

// x (+ add) -> split planes, with optional zero padding to ldo columns
hipError_t launch_split(const float* x, int ldx, const float* add, int ldadd, int add_rows_per_batch, int add_valid_rows,
                        bf16_t* out_hi, bf16_t* out_lo, int ldo, int M, int d, int seq_len, hipStream_t s, int fmt = 0);

// LearnedSinusoidalPosEmb + Linear(d+1, dt) + SiLU (NS2:108-120, 839-843): times[B] -> out[B, ld_out] columns [0, dt)
// wt is the Linear weight stored K-major [dim+1, dt]; feat_ws is a [B, dim+1] fp32 scratch.
hipError_t launch_time_embed(const float* times, const float* freqs, const float* wt, const float* bias, float* feat_ws,
                             float* out, int ld_out, int B, int dim, int dt, float* ws, size_t ws_bytes, hipStream_t s, int plan_B = 0);

// out[b, j] = act( sum_k in[b,k] * wt[k, j] + bias[j] ), wt stored K-major ([K, J]); act 0 none, 1 SiLU
// ws: caller-owned scratch of skinny_linear_workspace_bytes(B, K, J) for the split-K partial sums (null: no K split)
// plan_B > 0: split K as a batch of plan_B rows would (same order of partial sums: ns2_model_time_table)
size_t skinny_linear_workspace_bytes(int B, int K, int J, int plan_B = 0);
hipError_t launch_skinny_linear(const float* in, int ld_in, const float* wt, const float* bias, float* out, int ld_out,
                                int B, int K, int J, int act, float* ws, size_t ws_bytes, hipStream_t s, int plan_B = 0);
hipError_t launch_param_sample(const float* const* ptrs, const long* numels, int n, float* out, hipStream_t s);
hipError_t launch_add_row(const float* row, const float* add, float* out, int B, long J, hipStream_t s);

// batched fp32 [R, C] -> [C, R]
hipError_t launch_transpose_f32(const float* in, int batch, int R, int C, float* out, hipStream_t s);

// mean over n of [B, n, d] -> [B, d]
hipError_t launch_mean_rows(const float* in, int B, int n, int d, float* out, hipStream_t s);

// out[b*ld_out + e] = src[e] for e < row_elems (broadcast a parameter row over the batch)
hipError_t launch_bcast_rows(const float* src, float* out, int B, long row_elems, long ld_out, hipStream_t s);

hipError_t launch_embedding(const int64_t* ids, const float* table, float* out, long n, int dim, long pad_id, hipStream_t s);
hipError_t launch_join(const bf16_t* hi, const bf16_t* lo, int ld, float* out, int ldo, long M, int d, hipStream_t s, int fmt = 0);
hipError_t launch_transpose_into(const float* src, int R, int C, float* dst, long ld_dst, long col_off, hipStream_t s);

// DDIM update (NS2:1396-1429), objective 'v'/'eps'/'x0', sigmoid/cosine/linear schedule evaluated on device
struct DdimArgs {
  float* audio; const float* model_out; float* out;   // [B, n*d]; out may alias audio
  const float* times; const float* times_next;        // [B]
  int B; long per_batch;
  int objective;     // 0 'v', 1 'eps', 2 'x0'
  int schedule;      // 0 sigmoid, 1 cosine, 2 linear
  float scale;
};
hipError_t launch_ddim(const DdimArgs& a, hipStream_t s);

// CFG mix: out = null + (cond - null) * scale   (NS2:927)
hipError_t launch_cfg_mix(const float* cond, const float* null, float* out, long n, float scale, hipStream_t s);

// weight packing: fp32 [rows, C, T] (T taps, 1 for linear) -> split planes [rows_p, T*Cp]; row_map[r] = source row or -1
// one rectangular part of a packed weight and where its fp32 values live (elementwise.hip repack_kernel; built by ns2_weights_repack_build)
struct RepackDesc {
  const float* src; long sr, sc, st;        // element (r, c, tap) of the part at src[r * sr + c * sc + tap * st] (element strides, may be negative)
  bf16_t* dst_hi; long drs;                 // packed weight: first plane, physical row stride
  int row0, rows, col0, cols;               // packed rows [row0, row0 + rows), packed columns (per tap) [col0, col0 + cols)
  int Cp, T, fmt, il;                       // packed columns per tap, taps, PlaneFmt, interleaved lines
  long block0; int chunks, pad_;            // first block of the part, 256-element chunks per packed row
};
hipError_t launch_repack(const RepackDesc* tab, int n, long total_blocks, hipStream_t s);
hipError_t launch_pack_weight(const float* src, int C, int T, int Cp, const int* row_map, int rows_p, bf16_t* dst_hi,
                              bf16_t* dst_lo, int ldk, int k_off, hipStream_t s, int fmt = 0);

// EnCodec residual VQ encode (HFENC:364-369, 424-447)
struct RvqArgs {
  const float* x;            // [M, D] latents
  const float* codebooks;    // [Q, C, D]
  const float* cb_norm;      // [Q, C] 0.5*|e|^2 (from launch_rvq_prepare)
  int64_t* codes;            // [M, Q]
  float* emb;                // [M, D] sum of selected codes (may be null)
  float* residual;           // [M, D] final residual (may be null)
  int* near_tie_count;       // optional counter of fp64 re-checks taken
  int M, Q, C, D;
  float tie_eps;
};
// EnCodec SEANet helpers (elementwise.hip): pre-convolution activation + reflect prefix + operand planes; prefix removal;
// one nn.LSTM layer (recurrent part; the input projections are a GEMM)
hipError_t launch_seanet_prep(const float* x, int ldx, int in_prefix, const float* add, int ldadd, int B, long T, int C, int elu,
                              int prefix, int im2col_k, bf16_t* out_hi, bf16_t* out_lo, int ldo, int fmt, hipStream_t s);
hipError_t launch_seanet_prep2(const float* x, int ldx, int in_prefix, int B, long T, int C, int prefix, bf16_t* elu_hi, bf16_t* elu_lo,
                               int elu_ld, int elu_col0, int elu_cols, bf16_t* raw_hi, bf16_t* raw_lo, int raw_ld, int raw_col0,
                               int raw_cols, int fmt, hipStream_t s);
hipError_t launch_seanet_conv_narrow(const float* x, long ldx, int in_prefix, int B, long T, int ci, int co, int k, int elu, const float* w,
                                     const float* bias, float* out, long ldo, hipStream_t s);
hipError_t launch_seanet_resblock_narrow(const float* x, long ldx, int in_prefix, int B, long T, int C, const float* w1p, const float* b1,
                                         const float* w2p, const float* wsp, const float* b2s, float* out, long ldo, hipStream_t s);
hipError_t launch_seanet_unpad(const float* src, long ld_src, int prefix, float* dst, long ld_dst, int B, long T, int C, hipStream_t s);
long lstm_state_floats(int B, int H);     // caller scratch of launch_lstm_layer (h exchange, cell state / barrier counter)
hipError_t launch_lstm_layer(const float* xproj, long ld_x, const float* w_hh, const float* b_hh, float* h_a, float* h_b,
                             float* c_state, long state_floats, const float* resid, long ld_r, float* out, long ld_o, int B,
                             long T, int H, hipStream_t s);

// both layers of a 2-layer LSTM (H = 512) in one launch, layer 2 overlapped with layer 1; hipErrorNotReady = not available on
// this device / switched off: run the layers one after the other
long lstm2_state_floats();
hipError_t launch_lstm2(const float* xproj, long ld_x, const float* w_hh1, const float* b_hh1, const float* w_ih2, const float* b_ih2,
                        const float* w_hh2, const float* b_hh2, float* state, long state_floats, const float* resid, long ld_r, float* out,
                        long ld_o, int B, long T, hipStream_t s);

hipError_t lstm_abort_inject(unsigned int n);   // test hook
unsigned int lstm_abort_read(bool reset);   // persistent-LSTM launches that gave up at their step barrier since the last reset (synchronising)

// values that left the IEEE-half range in this translation unit's kernels since the last reset (synchronising reads)
unsigned int saturation_read_gemm(bool reset);
unsigned int saturation_read_gemm2(bool reset);
unsigned int saturation_read_attention(bool reset);
unsigned int saturation_read_elementwise(bool reset);
unsigned int saturation_read_backward(bool reset);       // the FMT_H8 conversions of the training kernels (backward.hip)
// stream-ordered, non-synchronising copy of the same counter into (pinned) host memory
hipError_t saturation_peek_gemm(unsigned int* dst, hipStream_t s);
hipError_t saturation_peek_gemm2(unsigned int* dst, hipStream_t s);
hipError_t saturation_peek_attention(unsigned int* dst, hipStream_t s);
hipError_t saturation_peek_elementwise(unsigned int* dst, hipStream_t s);
hipError_t saturation_peek_backward(unsigned int* dst, hipStream_t s);

hipError_t launch_rvq_prepare(const float* codebooks, float* cb_norm, int Q, int C, int D, hipStream_t s);
hipError_t launch_rvq_encode(const RvqArgs& a, hipStream_t s);
// decode: emb[m] = sum_q codebooks[q][codes[m][q]]  (HFENC:440-447)
hipError_t launch_rvq_decode(const int64_t* codes, const float* codebooks, float* emb, int M, int Q, int C, int D,
                             hipStream_t s);

// ---------------------------------------------------------------------------------------------- backward pass (backward.hip)
// fp32 [M, C] (xf) or operand planes (in_hi / in_lo, interleaved bf16 hi/lo) -> row planes and / or TRANSPOSED planes
// T[c][m] (the operands of the weight-gradient GEMMs and of the attention backward), optionally shifted along the token axis
// inside each utterance: T[c][m] = in[m - shift][c] if m - shift lies in the utterance of m, else 0.
struct TPlanesArgs {
  const float* xf; long ldx;
  const bf16_t* in_hi; const bf16_t* in_lo; int ld_in; int in_col0;
  int M, C;
  int seq_len, shift;
  bf16_t* row_hi; bf16_t* row_lo; int ld_row;      // optional (fp32 input, shift 0): planes [M, ld_row], zero beyond C
  bf16_t* t_hi; bf16_t* t_lo; long ld_t;           // optional transposed planes, ld_t (logical, multiple of 32) token columns, zero beyond
  int per_batch;                                   // 0: row c, column m (all M tokens); 1: row b * t_rows_per_batch + c, column n
  int t_rows_per_batch; int t_rows;                // rows written (>= C, zeros beyond C): per utterance / in total
  float* colsum_partial;                           // optional (fp32 input): [slices][C] column sums of 64-row tiles (bias gradients)
  int fmt;                                         // FMT_BF16 (hi / lo lines) or FMT_H8 (the mixed training arithmetic): the format of the
                                                   // input planes, of the row planes and of the transposed planes alike
};
hipError_t launch_tplanes(const TPlanesArgs& a, hipStream_t s);
long tplanes_slices(int M, long ld_t);             // number of 64-row tiles (= colsum slots) of a non-per_batch launch
hipError_t launch_reduce_slices(const float* partial, long outer, int S, long inner, float* out, int accumulate, hipStream_t s);
hipError_t launch_wgrad_reduce(const float* partial, int S, int R, long ldp, int T, int Kp, int K, float* out, hipStream_t s);
hipError_t launch_film_gate_fwd(const float* h, long ldh, const float* film, int film_ld, int seq_len, long M, int d, float* out,
                                long ldo, hipStream_t s);
int film_gate_slices(int seq_len);
hipError_t launch_film_gate_bwd(const float* dg, long lddg, const float* h, long ldh, const float* film, int film_ld, int B, int seq_len,
                                int d, float* dh, long lddh, float* partial, hipStream_t s);
hipError_t launch_geglu_fwd(const float* pre, long ldp, long M, int f, bf16_t* out_hi, bf16_t* out_lo, int ldo, hipStream_t s, int fmt = 0);
hipError_t launch_geglu_bwd(const float* dh, long lddh, const float* pre, long ldp, long M, int f, float* dpre, long lddp, hipStream_t s);
struct NormBwdArgs {
  const float* x; long ldx;            // the norm's input [B * seq_len, d]
  const float* dy; long lddy;          // gradient of its output
  const float* gamma;                  // learned scale [d] or null
  const float* cond; int cond_ld;      // adaptive [gamma_c | beta_c] per utterance or null
  const float* dx_add; float* dx; long lddx;   // dx = (dx_add ? dx_add : 0) + d(loss)/dx ; dx may alias dx_add
  float* cond_partial;                 // [B * slices][2 d] (required with cond)
  float* gamma_partial;                // [B * slices][d] or null
  int B, seq_len, d;
};
int rmsnorm_bwd_slices(int seq_len);
hipError_t launch_rmsnorm_bwd(const NormBwdArgs& a, hipStream_t s);
hipError_t launch_attn_delta(const float* dO, long lddo, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, int B, int H, int Nq, float* delta,
                             hipStream_t s, int o_fmt = 0);
struct AttnBwdArgs {
  const bf16_t* q_hi; const bf16_t* q_lo; int ldq, q_col0;        // [B*Nq, ldq], head h at columns q_col0 + 64 h
  const bf16_t* k_hi; const bf16_t* k_lo; int ldk, k_col0;        // [B*Nk, ldk]
  const bf16_t* v_hi; const bf16_t* v_lo; int ldv, v_col0;        // [B*Nk, ldv] values, ROW-major
  const bf16_t* do_hi; const bf16_t* do_lo; int lddo;             // [B*Nq, lddo] gradient of the attention output, head h at 64 h
  const float* lse; const float* delta;                           // [B, H, Nq]
  float* dq; int lddq, dq_col0;                                   // fp32 outputs (null = not wanted; dk and dv come together)
  float* dk; int lddk, dk_col0;
  float* dv; int lddv, dv_col0;
  int B, H, Nq, Nk; float scale;
  // round 5: the gradients as OPERAND PLANES instead of fp32 (what consumes dq | dk | dv of a self attention is the q | k | v projection's
  // dgrad and wgrad GEMMs, nothing else): gp planes [rows, gp_ld] in format gp_fmt (FMT_BF16 hi / lo lines or FMT_H8), dq at columns
  // dq_col0 + 64 h of row b Nq + q, dk / dv at dk_col0 / dv_col0 of row b Nk + k; gp_q / gp_kv say which halves go there
  bf16_t* gp_hi; bf16_t* gp_lo; int gp_ld, gp_fmt, gp_q, gp_kv;
};
hipError_t launch_attention_bwd(const AttnBwdArgs& a, hipStream_t s);

}  // namespace ns2
