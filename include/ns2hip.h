/* libns2hip — C ABI of the MI355X-native NaturalSpeech2 denoising hot path.
 *
 * The reference (lucidrains/naturalspeech2-pytorch) has no FFI/plugin layer: its boundary for this path is the
 * Python class surface.  This header is the C-ABI a host binds instead; each entry point cites the
 * reference interface it replaces (NS2 = naturalspeech2_pytorch/naturalspeech2_pytorch.py, ATT = attend.py,
 * HFENC = transformers/models/encodec/modeling_encodec.py, the restatement of the un-vendored encodec RVQ).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is named host_*; the caller owns all buffers
 *     (inputs, outputs, workspaces); the library owns only packed-weight blobs (ns2_weight / ns2_model).
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are stream-ordered and never
 *     synchronise the device, except ns2_model_finalize / ns2_weight_pack which are one-time set-up calls.
 *   - return value 0 = ok; otherwise a negative code, message via ns2_last_error() (thread-local).
 *   - activations travel between kernels as 16-bit "split planes": hi = bf16(x), lo = bf16(x - hi).  precision
 *     3 = hi*hi+hi*lo+lo*hi on the bf16 MFMA (fp32-class, matches the fp32 reference to 1e-5), 1 = bf16 hi only,
 *     2 = ONE IEEE-half plane (x_lo == NULL, values saturate at +-65504) multiplied on the f16 MFMA, fp32 accumulate,
 *     4 = "mixed": the IEEE-half product plus BOTH first-order correction terms (a_hi.w_lo + a_lo.w_hi) evaluated in one
 *         block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e5m2 operands) per 32-deep k block: 4e-5-class accuracy
 *         at two thirds of the MFMA work of precision 3.
 *   - the library keeps no mutable host state on a launch path and never allocates there: scratch is caller-owned
 *     (ns2_*_workspace_bytes), constants live in per-device __device__ storage, so one host thread per device (or one
 *     process per device) may drive several devices concurrently, also under stream capture.  What it does keep, all of it
 *     write-once or atomic: per-device "attribute raised" / occupancy answers, the environment switches NS2_GEMM,
 *     NS2_LSTM_PERSISTENT and NS2_LSTM_FUSED (read once per process), and the test hooks ns2_debug_*.  One thread-local
 *     pointer exists for the duration of a ns2_model_forward* / ns2_model_prepare_cond call: the split-K region of the
 *     workspace that call was given (set on entry, restored on return).
 *   - layout of a split-plane matrix (x_hi, x_lo, ld): `ld` is the LOGICAL column count, a multiple of 32.
 *     x_lo != NULL: ONE bf16 buffer [rows, 2*ld]; every 32 logical columns occupy a 128-byte line [hi(32) | lo(32)],
 *       i.e. element (r, c) has hi at r*2*ld + ((c & ~31) << 1) + (c & 31) and lo 32 elements further; the caller
 *       passes x_lo == x_hi + 32 (anything else is NS2_ERR_HIP / invalid value).  precision 3 needs this form.
 *     x_lo == NULL: the dense [rows, ld] hi plane alone (precision 1: bf16, precision 2: IEEE half).
 *     precision 4 operands use the interleaved form with the line [half(32) | e5m2(x)(32 B) | e5m2((x-half(x))*2^12)(32 B)]
 *       (x_lo == x_hi + 32 as above); the operands of ns2_attention (q, k, transposed values) are dense IEEE half at
 *       precision 4, which is also what ns2_linear_qkv writes there.
 *     Transposed value planes (vt_hi, vt_lo, vt_ld) use the same rule along the key axis.
 */
#ifndef NS2HIP_H
#define NS2HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS2_OK 0
#define NS2_ERR_ARG -1
#define NS2_ERR_HIP -2
#define NS2_ERR_STATE -3
#define NS2_UNAVAILABLE 1 /* not an error: this fast path does not apply here, take the general one (ns2_lstm2) */

const char* ns2_last_error(void);
int ns2_version(void);
/* test hook: force the GEMM kernel variant (0 = dispatch by shape, 1 = 128x128 register-staged, 2 = 256x256 LDS-DMA,
 * 3 = dispatch by shape but never split K, 4 = dispatch by shape but never the round-6 kernels (ffconv_kernel.h, gemm3_kernel.h), 5 = dispatch by shape but
 * those kernels whenever a call is eligible, whatever its size) */
int ns2_debug_force_gemm(int kernel);
/* Split-K of small products.  ns2_model_forward* lend a region of their workspace to every GEMM of the pass: a product with too
 * few output tiles to fill the chip runs as K slices into fixed slots plus a second launch that adds the slots in order and
 * applies the epilogue (deterministic).  The stand-alone GEMM entry points below have no workspace argument and never split --
 * unless a test lends scratch to the calling host thread here: `scratch` = device memory of at least
 * ns2_splitk_scratch_bytes() bytes, used by every following stand-alone GEMM call of this thread until cleared with
 * (NULL, 0).  Test hook: the caller keeps the memory alive and the calls stream-ordered. */
int64_t ns2_splitk_scratch_bytes(void);
/* host arithmetic only (no device needed): the plan a product of M x N over k_tiles K tiles of 32 (k_tiles_per_tap per conv tap)
 * gets when scratch is lent -- `slices` (1 = not split) of `k_tiles_per_slice` K tiles of every tap */
int ns2_debug_splitk_plan(int M, int N, int k_tiles, int k_tiles_per_tap, int fp32_epilogue, int* slices, int* k_tiles_per_slice);
int ns2_debug_lend_splitk_scratch(void* scratch, int64_t bytes);

/* ------------------------------------------------------------------ packed weights (library-owned) */
typedef struct ns2_weight ns2_weight;
/* nn.Linear weight [rows, cols] (taps = 1) or Conv1d weight [rows, cols, taps] (NS2:583-595) -> K-contiguous bf16
 * split planes (precision 1 or 3: interleaved layout, usable at both; precision 2: dense IEEE-half rows; 4: the mixed-mode
 * lines; the latter two serve only their own precision), rows padded to 256, each tap's columns padded to 32.  geglu != 0 packs the rows of
 * FeedForward's first Linear (NS2:1021) so that GEGLU (NS2:1004-1007) fuses into the GEMM epilogue.
 * extra1x1 (may be null): a [rows, cols, 1] weight appended as one more, unshifted tap (WavenetResBlock.res_conv). */
int ns2_weight_pack(const float* w, int rows, int cols, int taps, int geglu, const float* extra1x1, int precision,
                    ns2_weight** out, void* stream);
void ns2_weight_free(ns2_weight* w);

/* ------------------------------------------------------------------ op-level entry points */
/* fp32 [M, d] (+ optional per-utterance addend) -> split planes [M, ldo] (zero padded) */
int ns2_split_f32(const float* x, int ldx, int M, int d, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision,
                  void* stream);

/* split planes -> fp32 (hi + lo); lo may be null (dense hi-only layout) */
int ns2_join_f32(const uint16_t* hi, const uint16_t* lo, int ld, float* out, int ldo, int64_t M, int d, int precision,
                 void* stream);

/* nn.Linear / CausalConv1d as one GEMM (NS2:1051-1069, 1021-1024, 583-595).
 * conv_taps = 0 for a Linear, k for a Conv1d(kernel k) with `dilation`; seq_len = tokens per utterance (rows never read
 * across utterances); pad_left = zero frames in front of the sequence: -1 = causal (k-1, CausalConv1d NS2:583-595),
 * (k-1)/2 = the "same" padding of SpeechPromptEncoder's convs (NS2:316).  act: 0 none, 1 SiLU, 2 ELU (after the bias).
 * out = act(A W^T + bias) (+ resid), fp32 */
int ns2_linear_f32(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                   int dilation, int seq_len, const float* bias, const float* resid, int ldr, float* out, int ldo,
                   int pad_left, int act, int precision, void* stream);
/* same, output as split planes [M, ldo] */
int ns2_linear_split(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                     int dilation, int seq_len, const float* bias, uint16_t* out_hi, uint16_t* out_lo, int ldo,
                     int pad_left, int act, int precision, void* stream);
/* The feed-forward causal conv (CausalConv1d(f, f, 3), NS2:1016 / 583-595) as one IEEE-half product has a kernel of its own
 * (csrc/ffconv_kernel.h).  It reads the weight as pre-tiled LDS images: ns2_weight_tile_conv3 builds them for a weight packed with
 * taps = 3 at precision 2 (one-time set-up, allocates once; ns2_weight_update keeps them current; after ns2_weights_repack call
 * ns2_weights_retile).  A later ns2_linear_split / ns2_linear_split_as call with that weight takes the kernel when: conv_taps = 3,
 * dilation = 1, pad_left = -1, act = 0, M % 256 == 0, seq_len % 256 == 0, lda == ns2_conv3_input_ld(cols) (dense half rows padded to
 * a multiple of 128 columns; the padding is never multiplied), output dense half or FMT_H8 lines.  Results are bit-identical to the
 * general kernel's (same products, same summation order).  ns2_model_finalize does all of this for precisions 2 / 5 / 6. */
int ns2_weight_tile_conv3(ns2_weight* w, void* stream);
int ns2_conv3_input_ld(int cols);
/* The mixed linear products (precision 4, FMT_H8 operands) on full 256-row tiles have a lean kernel too (csrc/gemm3_kernel.h): same
 * arithmetic and summation order as the general kernel (bit-identical results), weights read as pre-tiled LDS images.
 * ns2_weight_tile_linear builds them for a weight packed with taps = 1 at precision 4 (also with geglu = 1); ns2_linear_f32 / _split /
 * _split_as / _qkv / _geglu then take the kernel when M % 256 == 0 and K >= 96.  Same life-cycle rules as ns2_weight_tile_conv3. */
int ns2_weight_tile_linear(ns2_weight* w, void* stream);
/* ... and so has the WavenetResBlock of the hybrid plan (csrc/wavenet3_kernel.h; NS2:597-642): ns2_weight_tile_wavenet builds the tiled
 * images of a weight packed with taps = 3 and extra1x1 at precision 4 (square, channels % 256 == 0); ns2_wavenet_block at precision 5
 * then takes the lean kernel when M % 256 == 0, seq_len % 256 == 0 and dilation <= 128.  Bit-identical to the general kernel. */
int ns2_weight_tile_wavenet(ns2_weight* w, void* stream);
/* same, with the output planes in the format of ANOTHER precision (out_precision 3: bf16 hi / lo lines from a precision-4 product --
 * the q | k | v projection of the mixed training arithmetic, whose attention stays bf16 x3) */
int ns2_linear_split_as(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int conv_taps,
                        int dilation, int seq_len, const float* bias, uint16_t* out_hi, uint16_t* out_lo, int ldo,
                        int pad_left, int act, int precision, int out_precision, void* stream);
/* FeedForward first half: GEGLU(Linear(x)) (NS2:1004-1007, 1021); w packed with geglu=1; packed_bias from
 * ns2_geglu_pack_bias; out planes [M, ldo] with ldo = round_up(f, 32) */
int ns2_linear_geglu(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M,
                     const float* packed_bias, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream);
int ns2_geglu_pack_bias(const float* bias, int f, float* packed, int packed_len, void* stream);
/* fused q/k/v projection (NS2:1051-1053, 1063): columns < split_col -> planes [M, ldo]; columns >= split_col
 * (the values) -> transposed planes vt[b][col - split_col][n] with row stride vt_ld (for ns2_attention) */
int ns2_linear_qkv(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int seq_len,
                   int split_col, uint16_t* out_hi, uint16_t* out_lo, int ldo, uint16_t* vt_hi, uint16_t* vt_lo,
                   int vt_ld, int precision, void* stream);
/* WavenetResBlock (NS2:597-642) in one launch: out = tanh(g)*sigmoid(g) + res_conv(x), g = conv_dil(x)*gamma_t+beta_t.
 * w packed with taps=3 and extra1x1 = res_conv.weight; film[b] = [gamma(dim) | beta(dim)] = to_time_cond(t).
 * precision 5 (this entry point only): precision-4 operands and weight; the dilated conv multiplies their IEEE-half parts as
 * one product, res_conv keeps the fp8 correction terms (what model precision 5 runs; outputs wider than 128 columns) */
int ns2_wavenet_block(const ns2_weight* w, const uint16_t* a_hi, const uint16_t* a_lo, int lda, int M, int seq_len,
                      int dilation, const float* conv_bias, const float* res_bias, const float* film, int film_ld,
                      uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream);

/* Attend.forward (ATT:77-155), non-causal, head dim 64: o = softmax(q k^T * scale) v.
 * key_mask (may be null): key-padding mask [B, Nk] bytes, 1 = attend (ATT:92-94 / 136-138) */
int ns2_attention(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi,
                  const uint16_t* k_lo, int ldk, int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld,
                  uint16_t* o_hi, uint16_t* o_lo, int ldo, int B, int H, int Nq, int Nk, float scale,
                  const uint8_t* key_mask, int precision, void* stream);

/* the same with the head dimension named: 32, 64 or 128 (the reference's `dim_head` keyword, NS2:814-831 -> Attention ATT:77-155;
 * q / k columns of head h start at col0 + h * head_dim, vt rows at h * head_dim); scale is the caller's (dim_head ** -0.5) */
int ns2_attention_hd(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi,
                     const uint16_t* k_lo, int ldk, int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld,
                     uint16_t* o_hi, uint16_t* o_lo, int ldo, int B, int H, int Nq, int Nk, float scale,
                     const uint8_t* key_mask, int precision, int head_dim, void* stream);

/* RMSNorm.forward (NS2:727-746).  gamma may be null; cond (may be null) holds [gamma_c | beta_c] per batch row */
int ns2_rmsnorm(const float* x, int ldx, int M, int d, int seq_len, const float* gamma, const float* cond, int cond_ld,
                uint16_t* out_hi, uint16_t* out_lo, int ldo, float* out_f32, int ldo_f, int precision, void* stream);

/* out[b, j] = act(in[b, :] . wt[:, j] + bias[j]); wt K-major [K, J]; act: 0 none, 1 SiLU (conditioning projections:
 * to_time_cond / to_gamma_beta / to_prompt_cond Linears on [batch, dim_cond] rows, NS2:623, 744, 841, 860).
 * workspace (may be null: slower single pass) = ns2_skinny_linear_workspace_bytes(B, K, J) bytes of caller-owned scratch
 * for the deterministic split-K partial sums */
int64_t ns2_skinny_linear_workspace_bytes(int B, int K, int J);
int ns2_skinny_linear(const float* in, int ld_in, const float* wt, const float* bias, float* out, int ld_out, int B, int K,
                      int J, int act, void* workspace, int64_t workspace_bytes, void* stream);
/* to_time_cond (NS2:108-120, 839-843); wt = Linear weight K-major [dim+1, dt]; feat_ws [B, dim+1] scratch;
 * workspace as for ns2_skinny_linear(B, dim + 1, dt) */
int ns2_time_embed(const float* times, const float* freqs, const float* wt, const float* bias, float* feat_ws, float* out,
                   int ld_out, int B, int dim, int dt, void* workspace, int64_t workspace_bytes, void* stream);
int ns2_transpose_f32(const float* in, int batch, int R, int C, float* out, void* stream);
/* nn.Embedding gather, ids < 0 -> pad_id (PhonemeEncoder NS2:281-284) */
int ns2_embedding(const int64_t* ids, const float* table, float* out, int64_t n, int dim, int64_t pad_id, void* stream);

/* one DDIM update (NS2:1396-1430): audio <- f(audio, model_out, times, times_next).  objective 0 'v', 1 'eps', 2 'x0';
 * schedule 0 sigmoid, 1 cosine, 2 linear (NS2:1133-1148) */
int ns2_ddim_step(const float* audio, const float* model_out, float* out, const float* times, const float* times_next,
                  int B, int64_t per_batch, int objective, int schedule, float scale, void* stream);
/* classifier-free guidance mix (NS2:927) */
int ns2_cfg_mix(const float* cond_out, const float* null_out, float* out, int64_t n, float cond_scale, void* stream);

/* ------------------------------------------------------------------ EnCodec SEANet encoder / decoder pieces (HFENC:81-347)
 * The convolutions themselves are ns2_linear_* calls on channel-last rows (strided / transposed convolutions as 2-tap
 * convolutions over rows regrouped by the stride); these are the pieces around them.
 * ns2_seanet_prep: fp32 x [B, in_prefix + T, C] (row stride ldx, the first in_prefix rows of every utterance skipped;
 *   + optional add [B, T, C]) -> optional ELU (HFENC:285-347) -> operand
 *   planes [B, prefix + T, ldo] whose `prefix` leading rows per utterance hold the mirrored samples (row -j = row j): EnCodec's
 *   reflect padding of causal convolutions (HFENC:142-175).  im2col_k > 0 (C must be 1, prefix 0): output column j of row n is
 *   x_reflect[n - (k - 1) + j], the k taps of the first convolution as a K = k Linear.
 * ns2_seanet_unpad: dst[b][t][:] = src[b][prefix + t][:]  (drop the prefix rows of a convolution output)
 * ns2_lstm_layer: one nn.LSTM layer (HFENC:253-266; gate order i, f, g, o).  xproj [B*T, 4H] = x W_ih^T + b_ih (a GEMM, row
 *   b*T + t), w_hh [4H, H], b_hh [4H]; state = ns2_lstm_state_floats(B, H) floats of caller scratch (at least 3*B*H: with less
 *   than the full amount the recurrence runs as one launch per step instead of one persistent launch per layer);
 *   out[b*T + t, :H] = h_t (+ resid row). */
int ns2_seanet_prep(const float* x, int ldx, int in_prefix, const float* add, int ldadd, int B, int64_t T, int C, int elu, int prefix,
                    int im2col_k, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream);
/* ns2_seanet_prep2: one pass over fp32 x [B, in_prefix + T, C] for the two operands EnCodec's residual block (HFENC:268-301) needs
 *   of it -- ELU(x) and x -- each (either may be NULL) into the column window [col0, col0 + cols) of a plane buffer of row stride
 *   ld (logical columns; col0 a multiple of 32, cols >= C a multiple of 4, zero filled past C), rows laid out as ns2_seanet_prep
 *   does (`prefix` mirrored rows per utterance).  Lets x and the hidden activation share one operand, so that
 *   conv2(elu(h)) + shortcut(x) is ONE GEMM over the concatenated K. */
int ns2_seanet_prep2(const float* x, int ldx, int in_prefix, int B, int64_t T, int C, int prefix, uint16_t* elu_hi, uint16_t* elu_lo,
                     int elu_ld, int elu_col0, int elu_cols, uint16_t* raw_hi, uint16_t* raw_lo, int raw_ld, int raw_col0, int raw_cols,
                     int precision, void* stream);
/* ns2_seanet_conv_narrow: the two ends of the SEANet stacks as what they are -- the encoder's first convolution (ci = 1 -> co
 *   channels, HFENC:304-327) and the decoder's last (ci -> co = 1, HFENC:330-358): causal, reflect padded (HFENC:142-175), k = 7,
 *   fp32 on the vector ALUs in one pass over the wide side.  x fp32 [B, in_prefix + T, ci] (row stride ldx), optional ELU on the
 *   input, w [co, ci, k] as nn.Conv1d holds it, out fp32 [B * T, co] (row stride ldo).  Returns NS2_UNAVAILABLE (nothing launched)
 *   for any other shape: the caller takes the GEMM path. */
int ns2_seanet_conv_narrow(const float* x, int64_t ldx, int in_prefix, int B, int64_t T, int ci, int co, int k, int elu, const float* w,
                           const float* bias, float* out, int64_t ldo, void* stream);
/* ns2_seanet_resblock_narrow: EnCodec's residual block (HFENC:268-301) y = shortcut(x) + conv2(elu(conv1(elu(x)))) -- conv1 k = 3
 *   causal, dilation 1, reflect padded, C -> C / 2; conv2 and the shortcut 1 x 1 -- in ONE fp32 pass on the vector ALUs for the narrow
 *   end of the SEANet stacks (C = 32: the blocks that run at the full sample rate), instead of two operand-plane passes + two GEMMs.
 *   x fp32 [B, in_prefix + T, C] (row stride ldx) -> out fp32 [B * T, C] (row stride ldo).  Weights pre-arranged in consumption order:
 *   w1p [3][C][C / 2] = conv1.weight[h][c][t] at (t, c, h); w2p [C / 2][C] = conv2.weight[c][h] at (h, c); wsp [C][C] =
 *   shortcut.weight[o][c] at (c, o); b1 = conv1.bias; b2s = conv2.bias + shortcut.bias.  Returns NS2_UNAVAILABLE (nothing launched)
 *   for any other shape: the caller takes the GEMM path. */
int ns2_seanet_resblock_narrow(const float* x, int64_t ldx, int in_prefix, int B, int64_t T, int C, const float* w1p, const float* b1,
                               const float* w2p, const float* wsp, const float* b2s, float* out, int64_t ldo, void* stream);
int ns2_seanet_unpad(const float* src, int64_t ld_src, int prefix, float* dst, int64_t ld_dst, int B, int64_t T, int C, void* stream);
int64_t ns2_lstm_state_floats(int B, int H);
int ns2_lstm_layer(const float* xproj, int64_t ld_x, const float* w_hh, const float* b_hh, float* state, int64_t state_floats,
                   const float* resid, int64_t ld_r, float* out, int64_t ld_o, int B, int64_t T, int H, void* stream);

/* ns2_lstm2: BOTH layers of EnCodec's 2-layer nn.LSTM (HFENC:253-266, H = 512) in one launch, layer 2 running one frame behind
 *   layer 1 on its own workgroups and layer 1 forming layer 2's input projections (no GEMM between the layers).  xproj1 [B*T, 4H] =
 *   x W_ih1^T + b_ih1; w_hh1 / w_ih2 / w_hh2 [4H, H]; b_* [4H]; state = ns2_lstm2_state_floats() floats of caller scratch;
 *   out[b*T + t, :H] = h2_t (+ resid row).  Returns NS2_UNAVAILABLE -- nothing launched -- when the device cannot hold all the
 *   workgroups at once, B > 32, or NS2_LSTM_FUSED=0 / NS2_LSTM_PERSISTENT=0: call ns2_lstm_layer per layer instead. */
int64_t ns2_lstm2_state_floats(void);
int ns2_lstm2(const float* xproj1, int64_t ld_x, const float* w_hh1, const float* b_hh1, const float* w_ih2, const float* b_ih2,
              const float* w_hh2, const float* b_hh2, float* state, int64_t state_floats, const float* resid, int64_t ld_r, float* out,
              int64_t ld_o, int B, int64_t T, void* stream);

/* The one-launch LSTM recurrences synchronise their workgroups frame by frame through tagged values in global memory, which needs
 * every workgroup resident.  The launchers take that path only when the occupancy query says the device can hold them; should a
 * wait still time out (CU masking, a device saturated by other processes) the kernel gives up -- its output is then incomplete --
 * and counts the launch here instead of trapping.  Synchronises; call it after a codec run.  NS2_LSTM_PERSISTENT=0 forces the
 * per-step kernel. */
int ns2_lstm_abort_count(int reset, int64_t* count);
/* test hook: set that counter, as if n launches had given up (exercises the caller's fallback from the one-launch recurrences) */
int ns2_debug_lstm_inject_abort(int n);

/* Range guard of precisions 2 and 4 (IEEE-half operands stop at 65504 / 57344; beyond, values are clamped: finite but
 * wrong).  Counts, on the CURRENT device, the conversions that met an out-of-range (or NaN) value since the last reset.
 * Synchronises the device: call it between sampling runs, not per step.  A non-zero count means the model needs
 * precision 3 (bf16 planes: fp32 exponent range). */
int ns2_saturation_count(int reset, int64_t* count);
/* The same counters without a synchronisation: enqueues, on `stream`, copies of the four per-translation-unit counters (cumulative
 * since the last reset) into host4[0..3] -- host memory that stays valid until the stream reaches this point (pinned memory for a
 * truly asynchronous copy).  The caller records an event behind it and reads the values once the event has completed; the host-side
 * `Model` does so every few forwards, so that calls from ANY wrapper (this package's sampler, the reference's own
 * NaturalSpeech2 / Trainer around compat.HipBackedModel, a bare Model.forward) notice clamped activations. */
int ns2_saturation_peek_async(unsigned int* host4, void* stream);

/* EnCodec RVQ (HFENC:364-369, 424-447; reference call sites NS2:1445, NS2:1611, NS2:1496).
 * cb_norm: [Q, C] scratch filled by ns2_rvq_prepare (once per codebook set). codes: [M, Q] int64; emb/residual [M, D] or null */
int ns2_rvq_prepare(const float* codebooks, float* cb_norm, int Q, int C, int D, void* stream);
int ns2_rvq_encode(const float* x, const float* codebooks, const float* cb_norm, int64_t* codes, float* emb,
                   float* residual, int* near_tie_count, int M, int Q, int C, int D, float tie_eps, void* stream);
int ns2_rvq_decode(const int64_t* codes, const float* codebooks, float* emb, int M, int Q, int C, int D, void* stream);

/* ------------------------------------------------------------------ Model (NS2:811-1000) */
typedef struct ns2_model ns2_model;
typedef struct {
  int dim, depth, dim_head, heads, ff_mult, wavenet_layers, wavenet_stacks, dim_cond_mult;   /* NS2:814-823; dim_head 32, 64 or 128 */
  int condition_on_prompt, dim_prompt, num_latents_m, resampler_depth;                       /* NS2:826-831 */
  int precision;               /* 3 = bf16 x3 split "exact", 4 = fp16 + fp8 correction terms "mixed", 2 = fp16 single product
                                  "half", 1 = bf16 single product "fast"; 5 = "hybrid": the per-site plan of this model
                                  only: precision 4 everywhere except the feed-forward causal conv (NS2:1016) and the dilated
                                  convs of the Wavenet blocks (NS2:612), which run as one fp16 product on the same operands.
                                  6 = "hybrid_ff": 5, with the rest of the feed-forward branch (FF-in + GEGLU, NS2:1021, and
                                  FF-out, NS2:1024) as one fp16 product on dense half planes too.
                                  At precisions 2 / 4 / 5 / 6 the step-invariant pass (ns2_model_prepare_cond) computes in
                                  precision 3: its few rows condition every frame of every step */
} ns2_model_config;

int ns2_model_create(const ns2_model_config* cfg, ns2_model** out);
/* register one state_dict entry (reference key names); data = device fp32, contiguous */
int ns2_model_set_param(ns2_model* m, const char* name, const float* data, int ndim, const int64_t* dims);
/* pack all weights (one-time, synchronises); the registered parameter tensors must stay alive afterwards only for
 * the small fp32 vectors the executor reads in place (biases, gammas, sinusoid freqs) */
int ns2_model_finalize(ns2_model* m, void* stream);
int64_t ns2_model_workspace_bytes(const ns2_model* m, int B, int N, int n_prompt, int n_cond);
int64_t ns2_model_cond_bytes(const ns2_model* m, int B, int N, int n_prompt, int n_cond);
/* step-invariant conditioning (NS2:944-992: to_prompt_cond, perceiver_resampler, cond_to_model_dim, null substitutes)
 * -> `cond_state` (caller-owned, ns2_model_cond_bytes).  drop != 0 == cond_drop_prob 1 (the CFG null branch). */
int ns2_model_prepare_cond(ns2_model* m, const float* prompt, int n_prompt, const float* cond, int n_cond, int drop, int B,
                           int N, void* cond_state, void* workspace, int64_t workspace_bytes, void* stream);
/* Classifier-free guidance as ONE batch (NS2:914-927: cond and null forward of the same x): state_out (ns2_model_cond_bytes for 2 B
 * utterances) = the arrays of state_a (B utterances, e.g. drop = 0) followed by those of state_b (drop = 1).  ns2_model_forward with
 * batch 2 B, x and times repeated, then ns2_cfg_mix of the two halves.  Stream-ordered device copies, no allocation. */
int ns2_model_cond_stack(ns2_model* m, const void* state_a, const void* state_b, int B, int N, int n_cond, void* state_out, void* stream);
/* Model.forward (NS2:929-1000) for x [B, N, dim], times [B] -> out [B, N, dim]; cond_state null iff unconditional;
 * n_cond = the n_cond the cond_state was prepared with */
int ns2_model_forward(ns2_model* m, const float* x, const float* times, const void* cond_state, int n_cond, float* out, int B,
                      int N, void* workspace, int64_t workspace_bytes, void* stream);
/* Step-invariant TIME conditioning.  A sampler knows its times up front and they are shared by the batch
 * (get_sampling_timesteps, NS2:1303-1308), so every time-conditioning projection of the run (to_time_cond NS2:839-843 + the FiLM /
 * adaptive-norm Linears NS2:623, 744) is ONE table: ns2_model_time_table fills table[T, ns2_model_table_cols(m)] for times[T]
 * (device), and ns2_model_forward_row runs Model.forward for the step whose conditioning is `cond_row` = table + i * cols -- no
 * projection is launched inside the step.  Rows are computed with the K split of a batch of plan_B rows (pass the run's batch size):
 * for an unconditional model the step is then BIT-IDENTICAL to ns2_model_forward(times = t_i for every utterance); a conditioned
 * model adds the per-utterance prompt half kept in its cond_state (sum order differs: equal to fp32 rounding). */
int ns2_model_table_cols(const ns2_model* m);
int64_t ns2_model_time_table_workspace_bytes(const ns2_model* m, int plan_B);
int ns2_model_time_table(ns2_model* m, const float* times, int T, int plan_B, float* table, void* workspace, int64_t workspace_bytes,
                         void* stream);
int ns2_model_forward_row(ns2_model* m, const float* x, const float* cond_row, const void* cond_state, int n_cond, float* out, int B, int N,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* Sampled content checksum of every registered parameter (sum and absolute sum of ~2048 evenly spaced elements each), stream-ordered
 * and NON-synchronising: 2 * ns2_model_param_count(m) floats land in host_out (pinned memory) once the stream passes this point.
 * The parameters are read where ns2_model_set_param found them, so a caller that compares two checksums notices parameters
 * rewritten through `.data` (ema_pytorch, NS2:1793-1801) without a device synchronisation -- the host-side Model does, every few forwards. */
int ns2_model_param_count(const ns2_model* m);
int ns2_model_param_checksum(ns2_model* m, float* host_out, int capacity, void* stream);
/* optional intermediate taps for parity tests (fp32 copies made during forward / prepare_cond): "t", "c",
 * "wavenet.init", "wavenet.stack<s>", "wavenet.out", "layer<i>.attn", "layer<i>"; dst = null unregisters */
int ns2_model_debug_tap(ns2_model* m, const char* name, float* dst, int64_t dst_elems);
/* live kernel timing with HIP events on the caller's stream (bench.py roofline): category bits
 * 0 gemm<f32 epilogue> 1 gemm<split epilogue> 2 gemm<qkv> 3 gemm<geglu> 4 gemm<wavenet> 5 attention 6 rmsnorm.
 * _end synchronises on the recorded events and returns the summed kernel time and the number of launches. */
int ns2_model_profile_begin(ns2_model* m, unsigned category_mask);
int ns2_model_profile_end(ns2_model* m, double* total_ms, int64_t* launches);
void ns2_model_destroy(ns2_model* m);

/* ------------------------------------------------------------------ training: the backward pass
 * What `loss.backward()` runs for the denoiser (NS2:1635 `pred = self.model(...)` under autograd, NS2:1637-1666 the loss,
 * NS2:1886 `accelerator.backward`).  Two training arithmetics, selected per call by `precision`:
 *   3 = bf16 hi / lo planes, three bf16 MFMA products per contraction (the fp32 exponent range: no loss scaling);
 *   4 = "mixed": FMT_H8 lines, one IEEE-half product + both correction terms on the fp8 MFMA (2 MFMA units instead of 3).  IEEE half
 *       stops at 65504 and loses precision below 6e-5, so the caller scales the loss (a power of two; every gradient is linear in
 *       it) -- the reference trains under accelerate's fp16 mixed precision the same way (NS2:1710-1711, 1723-1726).  Values that
 *       leave the half range are counted (ns2_saturation_count): an overflowed step is detected, not silently clamped.
 *       The attention products stay bf16 x3 at both precisions (q / k / v / dO planes are bf16 hi / lo: ns2_linear_split_as).
 * The contractions reuse the forward GEMM family:
 *   dgrad  dX = dY W   : ns2_linear_f32 on a SECOND pack of the weight -- ns2_weight_pack of W^T ([in, out, taps] with the taps
 *                        flipped), conv_taps as in the forward, pad_left = 0 (the gradient of a causal conv looks ahead);
 *   wgrad  dW = dY^T X : ns2_wgrad on TRANSPOSED planes (contraction over the tokens), split over fixed slots + fixed-order sum.
 * Every reduction of this section is slot based and summed in a fixed order: gradients are deterministic, no atomics. */

/* the training kernels' share of the range guard, stream-ordered and non-synchronising (one word, pinned host memory); the gradient
 * GEMMs' own conversions are in ns2_saturation_peek_async's words.  ns2_saturation_count sums all of them. */
int ns2_saturation_peek_train_async(unsigned int* host1, void* stream);

/* re-pack a weight IN PLACE from new fp32 values (same shape / flags as the ns2_weight_pack call that made it): stream-ordered,
 * no allocation, no synchronisation -- the per-step refresh of an optimizer's weights */
int ns2_weight_update(ns2_weight* w, const float* w_src, const float* extra1x1, void* stream);

/* Every pack of a training pass in ONE launch.  A part = a rectangle of a packed weight (rows [row0, row0 + rows), columns per tap
 * [col0, col0 + cols) of a plain ns2_weight_pack weight: no GEGLU permutation, no extra 1x1 block) and where its fp32 values live:
 * element (r, c, tap) at src[r * sr + c * sc + tap * st], strides in elements and possibly negative -- the parameter's own storage
 * serves the forward pack (sr = C T, sc = T, st = 1), the dgrad pack W^T (sr = T, sc = C T) with flipped taps (st = -1 from src + T - 1)
 * and q | kv concatenations (two parts) without any copy.  ns2_weights_repack_build writes the device table (caller-owned memory of
 * ns2_weights_repack_table_bytes(n) bytes; synchronises once), ns2_weights_repack replays it: stream-ordered, no allocation. */
typedef struct {
  ns2_weight* w; const float* src;
  int64_t sr, sc, st;
  int row0, rows, col0, cols;
} ns2_repack_part;
int64_t ns2_weights_repack_table_bytes(int n);
int ns2_weights_repack_build(const ns2_repack_part* parts, int n, void* table_device, int64_t table_bytes, int64_t* total_blocks, void* stream);
int ns2_weights_repack(const void* table_device, int n, int64_t total_blocks, void* stream);
/* after ns2_weights_repack: rebuild the tiled images (ns2_weight_tile_conv3 / _linear / _wavenet) of those weights that have them --
 * one small launch per such weight, stream-ordered, no allocation.  With it packs that carry images may be part of a repack table. */
int ns2_weights_retile(ns2_weight* const* weights, int n, void* stream);

/* fp32 gradient x [M, C] (row stride ldx) -> any of
 *   row planes [M, ld_row] (zero beyond C)                                  -- the A operand of the dgrad GEMM;
 *   transposed planes T[c][m] with ld_t token columns (ld_t a multiple of 32, zero beyond M) and t_rows rows (zero beyond C; a
 *     W operand of ns2_wgrad needs its rows padded to a multiple of 256)    -- the operands of ns2_wgrad;
 *     per_batch != 0: T[b * t_rows + c][n] per utterance of seq_len tokens   -- the transposed operands of ns2_attention_bwd;
 *     shift: T[c][m] = x[m - shift][c] if m - shift lies in the utterance of m (seq_len tokens each), else 0;
 *   colsum_partial [ns2_grad_prep_slices(M, ld_t)][C]: column sums of 64-row tiles (sum the slots with ns2_reduce_slices:
 *     the bias gradient). */
int64_t ns2_grad_prep_slices(int M, int64_t ld_t);
int ns2_grad_prep(const float* x, int64_t ldx, int M, int C, int seq_len, int shift, uint16_t* row_hi, uint16_t* row_lo, int ld_row,
                  uint16_t* t_hi, uint16_t* t_lo, int64_t ld_t, int t_rows, int per_batch, float* colsum_partial, int precision, void* stream);
/* the same transposition for an activation that already exists as operand planes (columns [in_col0, in_col0 + C) of [M, ld_in]):
 * tap t of a causal conv (NS2:583-595) contributes x[n - (k - 1 - t) * dilation], i.e. shift = (k - 1 - t) * dilation */
int ns2_planes_transpose(const uint16_t* in_hi, const uint16_t* in_lo, int ld_in, int in_col0, int M, int C, int seq_len, int shift,
                         uint16_t* t_hi, uint16_t* t_lo, int64_t ld_t, int t_rows, int per_batch, int precision, void* stream);
/* out[o * inner + j] (+)= sum_s partial[(o * S + s) * inner + j], s in increasing order */
int ns2_reduce_slices(const float* partial, int64_t outer, int S, int64_t inner, float* out, int accumulate, void* stream);
/* weight gradient of nn.Linear / CausalConv1d (NS2:583-595, 1021-1024, 1051-1069): dw[r, k, t] = sum_m dY[m, r] * X_t[m, k].
 * dyt: transposed planes of dY, R rows; xt: transposed planes of the T (shifted) inputs stacked along the rows, tap t at rows
 * [t * Kp, t * Kp + K), allocated (zeros) up to the next multiple of 256 rows; both with ld_t token columns.  dw [R, K, T] fp32
 * (the nn.Conv1d layout; T = 1: nn.Linear).  workspace = ns2_wgrad_workspace_bytes(R, T * Kp, ld_t) bytes of caller scratch. */
int64_t ns2_wgrad_workspace_bytes(int R, int ncols, int64_t ld_t);
int ns2_wgrad(const uint16_t* dyt_hi, const uint16_t* dyt_lo, const uint16_t* xt_hi, const uint16_t* xt_lo, int64_t ld_t, int R, int T,
              int Kp, int K, float* dw, void* workspace, int64_t workspace_bytes, int precision, void* stream);

/* The same gradient from the TOKEN-MAJOR operand planes themselves: dy [M, ld_dy] (the row planes ns2_grad_prep writes for the dgrad
 * GEMM), x [M, ld_x] (the operand planes the forward GEMM read).  Tap t of a causal conv's gradient reads x[m - (T - 1 - t) * dil]
 * inside the utterance of m (seq_len tokens, >= 32, M a multiple of it); T = 1: nn.Linear (dil / seq_len ignored).  The kernel forms
 * its MFMA fragments with gfx950's LDS transpose reads, so neither transposed nor shifted copies of the operands exist in memory.
 * Any shape is computed; ns2_wgrad_rows_preferred says whether this route is the faster one (the 256 x 256 kernel must fill the chip:
 * the dim = 128 model's narrow gradients stay on ns2_wgrad, which picks the 128 x 128 kernel).  workspace = ns2_wgrad_workspace_bytes(R, T * Kp, round_up(M, 32)). */
int ns2_wgrad_rows_preferred(int R, int ncols, int64_t M);
int ns2_wgrad_rows(const uint16_t* dy_hi, const uint16_t* dy_lo, int ld_dy, const uint16_t* x_hi, const uint16_t* x_lo, int ld_x, int64_t M,
                   int R, int T, int Kp, int K, int dil, int seq_len, float* dw, void* workspace, int64_t workspace_bytes, int precision,
                   void* stream);

/* WavenetResBlock's FiLM + gate (NS2:629-636) for the unfused training forward: out = tanh(z) * sigmoid(z), z = h * gamma_b + beta_b,
 * film[b] = [gamma (d) | beta (d)]; and its backward: dh = dg g'(z) gamma, partial[(b * slices + s)] = [sum dg g'(z) h | sum dg g'(z)]
 * over the s-th group of tokens of utterance b (ns2_film_gate_slices(seq_len) groups; ns2_reduce_slices gives d film [B, 2 d]) */
int ns2_film_gate_fwd(const float* h, int64_t ldh, const float* film, int film_ld, int seq_len, int64_t M, int d, float* out, int64_t ldo,
                      void* stream);
int ns2_film_gate_slices(int seq_len);
int ns2_film_gate_bwd(const float* dg, int64_t lddg, const float* h, int64_t ldh, const float* film, int film_ld, int B, int seq_len, int d,
                      float* dh, int64_t lddh, float* partial, void* stream);
/* GEGLU (NS2:1004-1007) on the saved pre-activation pre [M, ldp] = [x (f) | gate (f)]: planes of gelu(gate) * x, and the backward */
int ns2_geglu_fwd(const float* pre, int64_t ldp, int64_t M, int f, uint16_t* out_hi, uint16_t* out_lo, int ldo, int precision, void* stream);
int ns2_geglu_bwd(const float* dh, int64_t lddh, const float* pre, int64_t ldp, int64_t M, int f, float* dpre, int64_t lddp, void* stream);
/* RMSNorm backward (NS2:727-746): dx = (dx_add ? dx_add : 0) + dL/dx (dx may alias dx_add: the residual stream's gradient);
 * cond_partial [B * slices][2 d] -> the adaptive (gamma_c, beta_c) gradients, gamma_partial [B * slices][d] -> the learned gamma's */
int ns2_rmsnorm_bwd_slices(int seq_len);
int ns2_rmsnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* gamma, const float* cond, int cond_ld, int B,
                    int seq_len, int d, const float* dx_add, float* dx, int64_t lddx, float* cond_partial, float* gamma_partial, void* stream);

/* ns2_attention that also returns lse [B, H, Nq] = log2 of the softmax denominator of the scaled scores (m + log2 l): what the
 * backward recomputes P from (ATT:77-155; no key-padding mask on this path: Model never passes one) */
int ns2_attention_lse(const uint16_t* q_hi, const uint16_t* q_lo, int ldq, int q_col0, const uint16_t* k_hi, const uint16_t* k_lo, int ldk,
                      int k_col0, const uint16_t* vt_hi, const uint16_t* vt_lo, int vt_ld, uint16_t* o_hi, uint16_t* o_lo, int ldo, int B,
                      int H, int Nq, int Nk, float scale, float* lse, int precision, int o_precision, void* stream);
/* (o_precision: the format of the output planes when it is not the operands': 4 = FMT_H8 lines for a precision-4 out-projection
 * behind a bf16 x3 attention; 0 = as `precision`) */
/* delta[b, h, q] = sum_d dO[q, 64 h + d] * O[q, 64 h + d] (O from its operand planes, o_precision 3 or 4) */
int ns2_attention_delta(const float* d_out, int64_t ld_dout, const uint16_t* o_hi, const uint16_t* o_lo, int ldo, int B, int H, int Nq,
                        float* delta, int o_precision, void* stream);
/* flash-attention backward, head dim 64: dq = scale * dS k, dk = scale * dS^T q, dv = P^T dO with P recomputed from lse and
 * dS = P (dO v^T - delta).  q / k / v / d_out: row-major planes (head h at columns col0 + 64 h) -- the kernels form K^T, Q^T and
 * dO^T themselves (LDS transpose reads of the row-major tiles: round 4 took per-utterance transposed copies here).  dq == NULL or
 * dk == dv == NULL skips that half (cross-attention to a context that needs no gradient). */
typedef struct {
  const uint16_t* q_hi; const uint16_t* q_lo; int ldq, q_col0;
  const uint16_t* k_hi; const uint16_t* k_lo; int ldk, k_col0;
  const uint16_t* v_hi; const uint16_t* v_lo; int ldv, v_col0;
  const uint16_t* do_hi; const uint16_t* do_lo; int lddo;
  const float* lse; const float* delta;
  float* dq; int lddq, dq_col0;
  float* dk; int lddk, dk_col0;
  float* dv; int lddv, dv_col0;
  int B, H, Nq, Nk; float scale;
  /* round 5: gp_q / gp_kv != 0 -- that half of the gradients leaves the kernel as OPERAND PLANES instead of fp32 (the q | k | v projection's
   * dgrad and wgrad GEMMs are their only consumers): planes [rows, gp_ld] of precision gp_precision (3: bf16 hi / lo lines, 4: FMT_H8),
   * dq at columns dq_col0 + 64 h of row b Nq + q, dk / dv at dk_col0 / dv_col0 + 64 h of row b Nk + k (column offsets multiples of 32) */
  uint16_t* gp_hi; uint16_t* gp_lo; int gp_ld, gp_precision, gp_q, gp_kv;
} ns2_attn_bwd_args;
int ns2_attention_bwd(const ns2_attn_bwd_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NS2HIP_H */
