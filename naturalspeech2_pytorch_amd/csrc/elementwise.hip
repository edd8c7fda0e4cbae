// HBM-bound and tiny kernels of the NaturalSpeech2 denoiser on gfx950: RMSNorm (wavefront reduction), fp32 ->
// split-plane conversion, time embedding, the batched skinny conditioning projections, DDIM update, CFG mix,
// weight packing.  All vectorised to 16 B per lane where the layout allows (guide G13).
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

// ---------------------------------------------------------------- RMSNorm (NS2:727-746)
// one wave per row, the row lives in registers (CH float4 per lane, d <= 1024*CH/4... ), fp32 math, split-plane output;
// all loads/stores are 16 B (x, gamma, adaptive gamma/beta) or 8 B (bf16 planes) per lane.
template <int CH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const NormArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* x = a.x + row * a.ldx;
  float4 v[CH];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = lane * 4 + 256 * j;
    v[j] = (c < a.d) ? *reinterpret_cast<const float4*>(x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ss += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);     // F.normalize eps
  const float scale = sqrtf((float)a.d);
  const int b = a.seq_len > 0 ? (int)(row / a.seq_len) : 0;
  const float* gc = a.cond ? a.cond + (long)b * a.cond_ld : nullptr;
  const bool vec_cond = gc && ((a.cond_ld | a.d) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.cond) & 15) == 0);
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = lane * 4 + 256 * j;
    if (c >= a.ldo) continue;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < a.d) {
      const float xv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      float gm[4] = {1.f, 1.f, 1.f, 1.f}, gg[4] = {1.f, 1.f, 1.f, 1.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.gamma) {
        const float4 t = *reinterpret_cast<const float4*>(a.gamma + c);
        gm[0] = t.x; gm[1] = t.y; gm[2] = t.z; gm[3] = t.w;
      }
      if (gc) {
        if (vec_cond) {
          const float4 t = *reinterpret_cast<const float4*>(gc + c), u = *reinterpret_cast<const float4*>(gc + a.d + c);
          gg[0] = t.x; gg[1] = t.y; gg[2] = t.z; gg[3] = t.w;
          bb[0] = u.x; bb[1] = u.y; bb[2] = u.z; bb[3] = u.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { gg[e] = gc[c + e]; bb[e] = gc[a.d + c + e]; }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = xv[e] * inv * scale;
        if (a.gamma) t *= gm[e];
        if (gc) t = t * gg[e] + bb[e];
        o[e] = t;
      }
    }
    if (a.out_hi) {
      const bool il = a.out_lo != nullptr;
      store_cols4(a.out_hi + row * pld(a.ldo, il), c, o[0], o[1], o[2], o[3], a.fmt, il);
    }
    if (a.out_f && c < a.d) *reinterpret_cast<float4*>(a.out_f + row * a.ldo_f + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

hipError_t launch_rmsnorm(const NormArgs& a, hipStream_t s) {
  if (a.M <= 0 || (a.d & 3) || (a.ldo & 3) || (a.ldx & 3) || a.ldo > 2048 || a.d > 2048) return hipErrorInvalidValue;
  if (!planes_ok(a.out_hi, a.out_lo) || (a.out_lo && (a.ldo & 31))) return hipErrorInvalidValue;
  if (a.out_hi && ((a.fmt == FMT_H8 && !a.out_lo) || (a.fmt == FMT_F16 && a.out_lo))) return hipErrorInvalidValue;
  const dim3 grid((a.M + 3) / 4), block(256);
  const int width = a.ldo > a.d ? a.ldo : a.d;
  if (width <= 256) hipLaunchKernelGGL(rmsnorm_kernel<1>, grid, block, 0, s, a);
  else if (width <= 512) hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, block, 0, s, a);
  else if (width <= 1024) hipLaunchKernelGGL(rmsnorm_kernel<4>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, block, 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- x (+ add) -> split planes
__global__ __launch_bounds__(256) void split_kernel(const float* x, int ldx, const float* add, int ldadd, int add_rows,
                                                    int add_valid, bf16_t* out_hi, bf16_t* out_lo, int ldo, long M, int d,
                                                    int seq_len, int fmt) {
  const int chunks = ldo >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * chunks) return;
  const long row = idx / chunks;
  const int c = (int)(idx - row * chunks) * 4;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = ((d | ldx | ldadd) & 3) == 0;
  if (c < d) {
    if (vec) {
      const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < d) o[e] = x[row * ldx + c + e];
    }
    if (add) {
      const long b = row / seq_len;
      const int n = (int)(row - b * seq_len);
      if (n < add_valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < d) o[e] += add[(b * add_rows + n) * ldadd + c + e];
      }
    }
  }
  const bool il = out_lo != nullptr;
  store_cols4(out_hi + row * pld(ldo, il), c, o[0], o[1], o[2], o[3], fmt, il);
}

hipError_t launch_split(const float* x, int ldx, const float* add, int ldadd, int add_rows_per_batch, int add_valid_rows,
                        bf16_t* out_hi, bf16_t* out_lo, int ldo, int M, int d, int seq_len, hipStream_t s, int fmt) {
  if (M <= 0 || (ldo & 3) || ldo < d || (add && seq_len <= 0)) return hipErrorInvalidValue;
  if (!planes_ok(out_hi, out_lo) || (out_lo && (ldo & 31)) || (fmt == FMT_F16 && out_lo) || (fmt == FMT_H8 && !out_lo))
    return hipErrorInvalidValue;
  const long total = (long)M * (ldo >> 2);
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, add, ldadd,
                     add_rows_per_batch, add_valid_rows, out_hi, out_lo, ldo, (long)M, d, seq_len, fmt);
  return hipGetLastError();
}

// ---------------------------------------------------------------- skinny linear: out[b,j] = act(in[b,:] . wt[:,j] + bias[j])
// M = batch rows (<= 32 per pass), weight stored K-major so a wave streams it fully coalesced with 16-B loads (a lane owns 4
// adjacent output columns); the batch rows of the activation chunk sit in LDS and are read as broadcast ds_read_b128.
// Weight-bandwidth bound: this is how all 56-68 time/prompt conditioning projections of one denoising step are produced at once
// (470 MB of fp32 weights at d=512).  The K range is split over blockIdx.z so that >= ~2 blocks per CU stream concurrently
// (also for the tiny time-embedding Linear, 513 x 2048, which used to run on 8 workgroups); partial sums go to a CALLER-OWNED
// scratch (ns2_skinny_linear_workspace_bytes) and are combined in a fixed order by a second pass: deterministic, no atomics,
// no library-owned buffer.
constexpr int SK_KC = 64;          // K rows staged in LDS per chunk
constexpr int SK_COLS = 1024;      // output columns per block (256 threads x 4)
// ROWS = 32: every batch row has its own accumulators; ROWS = 1: the rows of this block's K range are identical, one product
// serves them all
template <int ROWS>
__device__ __forceinline__ void skinny_body(const float* in, int ld_in, const float* wt, const float* bias, float* out, int ld_out, int B,
                                            int J, int act, float* partial, int j, int b0, int nb, bool jvec, int kbeg, int kend,
                                            float (*s_in)[SK_KC + 4]) {
  float acc[ROWS][4];
#pragma unroll
  for (int b = 0; b < ROWS; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[b][e] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += SK_KC) {
    __syncthreads();
    for (int i = threadIdx.x; i < SK_KC * ROWS; i += blockDim.x) {
      const int b = i / SK_KC, k = i - b * SK_KC;          // coalesced along k, conflict-free LDS writes
      float v = 0.f;
      if (b < nb && k0 + k < kend) v = in[(long)(b0 + b) * ld_in + k0 + k];
      s_in[b][k] = v;
    }
    __syncthreads();
    // 8 weight rows (8 KiB per wave) are fetched one batch AHEAD of the FMAs that consume them, so loads are in flight
    // all the time (the single-buffered loop measured 2 TB/s: every wave alternated between waiting and computing)
    float w[8][4], wn[8][4];
    auto load_rows = [&](float (&dst)[8][4], int k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const long kr = k0 + k + e;
        if (kr < kend && jvec) {
          const float4 t = *reinterpret_cast<const float4*>(wt + kr * J + j);
          dst[e][0] = t.x; dst[e][1] = t.y; dst[e][2] = t.z; dst[e][3] = t.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) dst[e][c] = (kr < kend && j + c < J) ? wt[kr * J + j + c] : 0.f;
        }
      }
    };
    load_rows(w, 0);
#pragma unroll 1
    for (int k = 0; k < SK_KC; k += 8) {
      if (k + 8 < SK_KC) load_rows(wn, k + 8);
#pragma unroll
      for (int b = 0; b < ROWS; ++b) {
        const float4 t = *reinterpret_cast<const float4*>(&s_in[b][k]);       // wave-uniform address: LDS broadcast
        const float4 u = *reinterpret_cast<const float4*>(&s_in[b][k + 4]);
        const float x[8] = {t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[b][c] = fmaf(x[e], w[e][c], acc[b][c]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[e][c] = wn[e][c];
    }
  }
  if (j >= J) return;
#pragma unroll
  for (int b = 0; b < (ROWS == 1 ? 32 : ROWS); ++b) {
    if (b >= nb) continue;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = acc[ROWS == 1 ? 0 : b][c];
    if (partial) {                                          // [split][B][J]
      float* dst = partial + ((long)blockIdx.z * B + b0 + b) * J + j;
      if (jvec) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      else
#pragma unroll
        for (int c = 0; c < 4; ++c) if (j + c < J) dst[c] = v[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (j + c < J) {
          float t = v[c] + (bias ? bias[j + c] : 0.f);
          if (act == 1) t = siluf(t);
          out[(long)(b0 + b) * ld_out + j + c] = t;
        }
    }
  }
}

// Launch shapes: "wide" = 256 threads x 4 columns, 32 batch rows per block (the batched conditioning projections: tens of
// thousands of columns); "narrow" = ONE wave x 4 columns = 256 columns, `rpb` = 8 batch rows per block, for products too small to
// fill the chip with wide blocks (a single 2048 -> 1024 projection of a training step is 32 wide blocks on 256 CUs, each
// VALU-bound for ~50 us; as 512 narrow blocks it is a quarter of the FMAs per wave on every CU).  The K split, and with it the
// order of every sum, is the same in both shapes: results are bit-identical.
__global__ __launch_bounds__(256) void skinny_linear_kernel(const float* in, int ld_in, const float* wt, const float* bias,
                                                            float* out, int ld_out, int B, int K, int J, int act,
                                                            int k_per_split, float* partial, int rpb) {
  __shared__ __attribute__((aligned(16))) float s_in[32][SK_KC + 4];
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int b0 = blockIdx.y * rpb;
  const int nb = min(rpb, B - b0);
  const bool jvec = (j + 3 < J) && ((J & 3) == 0);       // 16-B aligned full group
  const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
  // Round 3: are the batch rows of this block's K range all IDENTICAL?  (The sampler's time conditioning: every utterance of a
  // step has the same diffusion time, NS2:1303-1308.)  Then one product serves every row: 32 x fewer FMAs -- with 1024 FMAs per
  // 8 weight rows the per-row loop is VALU-bound at ~1.9 TB/s of weight streaming.  The FMA order per row is the same in both
  // bodies, so a uniform batch gives every row the single-row result bit for bit.  (NaN != NaN: a NaN row takes the per-row body.)
  int same = 1;
  for (int i = threadIdx.x; i < (kend - kbeg) * (nb - 1); i += blockDim.x) {
    const int b = 1 + i / (kend - kbeg), k = kbeg + i % (kend - kbeg);
    if (in[(long)(b0 + b) * ld_in + k] != in[(long)b0 * ld_in + k]) same = 0;
  }
  if (nb == 1 || __syncthreads_and(same)) skinny_body<1>(in, ld_in, wt, bias, out, ld_out, B, J, act, partial, j, b0, nb, jvec, kbeg, kend, s_in);
  else if (rpb == 8) skinny_body<8>(in, ld_in, wt, bias, out, ld_out, B, J, act, partial, j, b0, nb, jvec, kbeg, kend, s_in);
  else skinny_body<32>(in, ld_in, wt, bias, out, ld_out, B, J, act, partial, j, b0, nb, jvec, kbeg, kend, s_in);
}

__global__ void skinny_reduce_kernel(const float* partial, int nsplit, const float* bias, float* out, int ld_out, int B, int J,
                                     int act) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * J) return;
  const int b = (int)(i / J), j = (int)(i - (long)b * J);
  float v = 0.f;
  for (int s = 0; s < nsplit; ++s) v += partial[((long)s * B + b) * J + j];      // fixed order: deterministic
  v += bias ? bias[j] : 0.f;
  if (act == 1) v = siluf(v);
  out[(long)b * ld_out + j] = v;
}

// K split: enough blocks to keep ~2 per CU streaming, in whole SK_KC chunks
// plan_B (> 0): take the K split a batch of plan_B rows would get.  The split fixes the order of the partial sums, so a table of
// conditioning rows built ahead of a sampling run in 32-row chunks is bit-identical to what each step's own launch (batch = plan_B
// identical rows) would have produced (ns2_model_time_table).
static void skinny_plan(int B, int K, int J, int* nsplit, int* kps, int plan_B = 0) {
  const long blocks = (long)((J + SK_COLS - 1) / SK_COLS) * (((plan_B > 0 ? plan_B : B) + 31) / 32);
  int want = (int)((512 + blocks - 1) / blocks);
  const int max_split = (K + SK_KC - 1) / SK_KC;
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  *kps = ((K + want - 1) / want + SK_KC - 1) / SK_KC * SK_KC;
  *nsplit = (K + *kps - 1) / *kps;
}
size_t skinny_linear_workspace_bytes(int B, int K, int J, int plan_B) {
  if (B <= 0 || K <= 0 || J <= 0) return 0;
  int nsplit, kps;
  skinny_plan(B, K, J, &nsplit, &kps, plan_B);
  return nsplit > 1 ? (size_t)nsplit * B * J * sizeof(float) : 0;
}

hipError_t launch_skinny_linear(const float* in, int ld_in, const float* wt, const float* bias, float* out, int ld_out,
                                int B, int K, int J, int act, float* ws, size_t ws_bytes, hipStream_t s, int plan_B) {
  if (B <= 0 || K <= 0 || J <= 0) return hipErrorInvalidValue;
  int nsplit, kps;
  skinny_plan(B, K, J, &nsplit, &kps, plan_B);
  if (plan_B > 0 && nsplit > 1 && (!ws || ws_bytes < (size_t)nsplit * B * J * sizeof(float))) return hipErrorInvalidValue;   // a planned split must not degrade silently
  if (nsplit > 1 && (!ws || ws_bytes < (size_t)nsplit * B * J * sizeof(float))) { nsplit = 1; kps = (K + SK_KC - 1) / SK_KC * SK_KC; }
  float* partial = nsplit > 1 ? ws : nullptr;               // no scratch -> one pass over the whole K (slower, still correct)
  const long wide_blocks = (long)((J + SK_COLS - 1) / SK_COLS) * ((B + 31) / 32) * nsplit;
  if (wide_blocks >= 256)
    hipLaunchKernelGGL(skinny_linear_kernel, dim3((J + SK_COLS - 1) / SK_COLS, (B + 31) / 32, nsplit), dim3(256), 0, s, in, ld_in,
                       wt, bias, out, ld_out, B, K, J, act, kps, partial, 32);
  else
    hipLaunchKernelGGL(skinny_linear_kernel, dim3((J + 255) / 256, (B + 7) / 8, nsplit), dim3(64), 0, s, in, ld_in,
                       wt, bias, out, ld_out, B, K, J, act, kps, partial, 8);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nsplit == 1) return e;
  const long n = (long)B * J;
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, nsplit, bias, out, ld_out,
                     B, J, act);
  return hipGetLastError();
}

// ---------------------------------------------------------------- time features (NS2:108-120): [t, sin(2 pi t w), cos(2 pi t w)]
__global__ void time_feat_kernel(const float* times, const float* freqs, float* feat, int B, int dim) {
  const int half = dim / 2, K = dim + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  const float t = times[b];
  float v;
  if (k == 0) v = t;
  else {
    const int f = (k - 1) % half;
    const float fr = t * freqs[f] * 2.0f * 3.14159265358979323846f;   // x * w * 2 * pi, left to right in fp32
    v = (k - 1 < half) ? sinf(fr) : cosf(fr);
  }
  feat[i] = v;
}

hipError_t launch_time_embed(const float* times, const float* freqs, const float* wt, const float* bias, float* feat_ws,
                             float* out, int ld_out, int B, int dim, int dt, float* ws, size_t ws_bytes, hipStream_t s, int plan_B) {
  const int K = dim + 1;
  hipLaunchKernelGGL(time_feat_kernel, dim3((B * K + 255) / 256), dim3(256), 0, s, times, freqs, feat_ws, B, dim);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_skinny_linear(feat_ws, K, wt, bias, out, ld_out, B, K, dt, /*act=*/1, ws, ws_bytes, s, plan_B);
}

// ---------------------------------------------------------------- small data movers
__global__ void transpose_kernel(const float* in, int R, int C, float* out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const float* src = in + (long)blockIdx.z * R * C;
  float* dst = out + (long)blockIdx.z * R * C;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) dst[(long)c * R + r] = tile[threadIdx.x][i];
  }
}
hipError_t launch_transpose_f32(const float* in, int batch, int R, int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(32, 8), 0, s, in, R, C, out);
  return hipGetLastError();
}

__global__ void mean_rows_kernel(const float* in, int n, int d, float* out) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) acc += in[((long)b * n + i) * d + c];
  out[(long)b * d + c] = acc / (float)n;
}
hipError_t launch_mean_rows(const float* in, int B, int n, int d, float* out, hipStream_t s) {
  hipLaunchKernelGGL(mean_rows_kernel, dim3((d + 255) / 256, B), dim3(256), 0, s, in, n, d, out);
  return hipGetLastError();
}

// Sampled content checksum of a list of parameter tensors (one workgroup per tensor: the sum and the absolute sum of ~2048 evenly
// spaced elements, reduced in a fixed order).  What the host-side Model compares, one call later and without synchronising, to notice
// parameters rewritten through `.data` (an EMA update touches every element) between two version-counter checks.
__global__ __launch_bounds__(256) void param_sample_kernel(const float* const* ptrs, const long* numels, float* out) {
  __shared__ float r0[256], r1[256];
  const float* p = ptrs[blockIdx.x];
  const long n = numels[blockIdx.x];
  const long stride = n > 2048 ? n / 2048 : 1;
  float s = 0.f, a = 0.f;
  for (long i = (long)threadIdx.x * stride; i < n; i += 256 * stride) { const float v = p[i]; s += v; a += fabsf(v); }
  r0[threadIdx.x] = s; r1[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) { r0[threadIdx.x] += r0[threadIdx.x + w]; r1[threadIdx.x] += r1[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = r0[0]; out[2 * blockIdx.x + 1] = r1[0]; }
}
hipError_t launch_param_sample(const float* const* ptrs, const long* numels, int n, float* out, hipStream_t s) {
  if (n <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(param_sample_kernel, dim3(n), dim3(256), 0, s, ptrs, numels, out);
  return hipGetLastError();
}

// out[b, j] = row[j] + add[b, j]: the step's conditioning of a CONDITIONED model from the hoisted time table row and the
// per-utterance prompt part (ns2_model_forward_row)
__global__ void add_row_kernel(const float* row, const float* add, float* out, long J, long n_total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_total) return;
  out[i] = row[i % J] + add[i];
}
hipError_t launch_add_row(const float* row, const float* add, float* out, int B, long J, hipStream_t s) {
  const long n = (long)B * J;
  hipLaunchKernelGGL(add_row_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, row, add, out, J, n);
  return hipGetLastError();
}

// out[b, :] = bcast[:]  (broadcast a parameter row to every batch entry; used for the CFG "null" conditioning)
__global__ void bcast_rows_kernel(const float* src, float* out, long row_elems, long ld_out, long n_total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_total) return;
  const long b = i / row_elems, e = i - b * row_elems;
  out[b * ld_out + e] = src[e];
}
hipError_t launch_bcast_rows(const float* src, float* out, int B, long row_elems, long ld_out, hipStream_t s) {
  const long n = (long)B * row_elems;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, out, row_elems, ld_out, n);
  return hipGetLastError();
}

// nn.Embedding gather with the reference's padding rule (NS2:281-282: ids < 0 -> pad_id)
__global__ void embedding_kernel(const int64_t* ids, const float* table, float* out, long n, int dim, long pad_id) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * dim) return;
  const long r = i / dim;
  const int c = (int)(i - r * dim);
  long id = ids[r];
  if (id < 0) id = pad_id;
  out[i] = table[id * dim + c];
}
hipError_t launch_embedding(const int64_t* ids, const float* table, float* out, long n, int dim, long pad_id, hipStream_t s) {
  const long tot = n * dim;
  hipLaunchKernelGGL(embedding_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ids, table, out, n, dim, pad_id);
  return hipGetLastError();
}

// dst[c * ld_dst + col_off + r] = src[r * C + c]   (nn.Linear weight [R=out, C=in] -> K-major slice of a wider matrix)
__global__ void transpose_into_kernel(const float* src, int R, int C, float* dst, long ld_dst, long col_off) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) dst[(long)c * ld_dst + col_off + r] = tile[threadIdx.x][i];
  }
}
hipError_t launch_transpose_into(const float* src, int R, int C, float* dst, long ld_dst, long col_off, hipStream_t s) {
  hipLaunchKernelGGL(transpose_into_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, s, src, R, C, dst, ld_dst, col_off);
  return hipGetLastError();
}

// split planes -> fp32 (hi + lo), used by debug taps and tests
__global__ void join_kernel(const bf16_t* hi, const bf16_t* lo, int ld, float* out, int ldo, long M, int d, int fmt) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * d) return;
  const long r = i / d;
  const int c = (int)(i - r * d);
  const bool il = lo != nullptr;
  const long o = r * pld(ld, il) + pcol(c, il);
  float v = (fmt != FMT_BF16) ? h2f(hi[o]) : bf2f(hi[o]);
  if (fmt == FMT_H8) {               // half + l8 * 2^-12 (the h8 byte duplicates the half to 3 bits and is not added)
    const unsigned char* line = reinterpret_cast<const unsigned char*>(hi + r * pld(ld, true) + ((c & ~31) << 1));
    v += bf8_to_f(line[96 + (c & 31)]) * (1.0f / H8_LO_SCALE);
  } else if (il) v += bf2f(lo[o]);
  out[r * ldo + c] = v;
}
hipError_t launch_join(const bf16_t* hi, const bf16_t* lo, int ld, float* out, int ldo, long M, int d, hipStream_t s, int fmt) {
  if (!planes_ok(hi, lo) || (fmt == FMT_F16 && lo) || (fmt == FMT_H8 && !lo)) return hipErrorInvalidValue;
  const long n = M * d;
  hipLaunchKernelGGL(join_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hi, lo, ld, out, ldo, M, d, fmt);
  return hipGetLastError();
}

// ---------------------------------------------------------------- DDIM update (NS2:1396-1430)
NS2_DEVINL float gamma_of(float t, int schedule) {
  float g;
  if (schedule == 0) {        // sigmoid_schedule NS2:1144-1148, start=-3 end=3 tau=1
    const float vs = 1.0f / (1.0f + expf(3.0f)), ve = 1.0f / (1.0f + expf(-3.0f));
    const float sg = 1.0f / (1.0f + expf(-(t * 6.0f - 3.0f)));
    g = (-sg + ve) / (ve - vs);
    return fminf(fmaxf(g, 1e-9f), 1.0f);
  } else if (schedule == 1) { // cosine_schedule NS2:1136-1142 (start=0, end=1, tau=1): (v_end - cos^2)/(v_end - 1), v_end ~ 0
    const float c = cosf(t * 1.57079632679489661923f);
    return fmaxf(c * c, 1e-9f);
  }
  return fmaxf(1.0f - t, 1e-9f);   // simple_linear_schedule NS2:1133-1134
}

__global__ __launch_bounds__(256) void ddim_kernel(const DdimArgs a) {
  const int b = blockIdx.y;
  const float g = gamma_of(a.times[b], a.schedule), gn = gamma_of(a.times_next[b], a.schedule);
  const float alpha = sqrtf(g) * a.scale, sigma = sqrtf(1.0f - g);
  const float alpha_n = sqrtf(gn) * a.scale, sigma_n = sqrtf(1.0f - gn);
  const float inv_sigma = 1.0f / fmaxf(sigma, 1e-10f);      // safe_div NS2:1122-1123
  const float inv_alpha = 1.0f / fmaxf(alpha, 1e-10f);
  const long base = (long)b * a.per_batch;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.per_batch; i += (long)gridDim.x * 1024) {
    const float4 x4 = *reinterpret_cast<const float4*>(a.audio + base + i);
    const float4 v4 = *reinterpret_cast<const float4*>(a.model_out + base + i);
    const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x0;
      if (a.objective == 0) x0 = alpha * xs[e] - sigma * vs[e];            // 'v'   NS2:1422
      else if (a.objective == 1) x0 = (xs[e] - sigma * vs[e]) * inv_alpha;  // 'eps' NS2:1419
      else x0 = vs[e];                                                      // 'x0'  NS2:1416
      const float eps = (xs[e] - alpha * x0) * inv_sigma;                   // NS2:1426
      o[e] = x0 * alpha_n + eps * sigma_n;                                  // NS2:1430
    }
    *reinterpret_cast<float4*>(a.out + base + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
hipError_t launch_ddim(const DdimArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.per_batch <= 0 || (a.per_batch & 3)) return hipErrorInvalidValue;
  const long blocks = (a.per_batch / 4 + 255) / 256;
  hipLaunchKernelGGL(ddim_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512), a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ void cfg_mix_kernel(const float* cond, const float* null, float* out, long n, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float u = null[i];
    out[i] = u + (cond[i] - u) * scale;       // NS2:927
  }
}
hipError_t launch_cfg_mix(const float* cond, const float* null, float* out, long n, float scale, hipStream_t s) {
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(cfg_mix_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, cond, null, out, n, scale);
  return hipGetLastError();
}

// ---------------------------------------------------------------- weight packing (one-time, at model finalize)
__global__ void pack_weight_kernel(const float* src, int C, int T, int Cp, const int* row_map, int rows_p, bf16_t* dst_hi,
                                   bf16_t* dst_lo, int ldk, int k_off, int fmt) {
  const int kk = blockIdx.x * 256 + threadIdx.x;          // column inside [0, T*Cp)
  const int rp = blockIdx.y;
  if (kk >= T * Cp || rp >= rows_p) return;
  const int tap = kk / Cp, c = kk - tap * Cp;
  const int r = row_map ? row_map[rp] : rp;
  float v = 0.f;
  if (r >= 0 && c < C) v = src[((long)r * C + c) * T + tap];
  const bool il = dst_lo != nullptr;
  const int kc = k_off + kk;
  const long o = (long)rp * pld(ldk, il) + pcol(kc, il);
  if (fmt == FMT_H8) {
    uint32_t h16, h8, l8;
    cvt2_h8(v, 0.f, h16, h8, l8);
    dst_hi[o] = (bf16_t)(h16 & 0xffffu);
    unsigned char* line = reinterpret_cast<unsigned char*>(dst_hi + (long)rp * pld(ldk, true) + ((kc & ~31) << 1));
    line[64 + (kc & 31)] = (unsigned char)(h8 & 0xffu);
    line[96 + (kc & 31)] = (unsigned char)(l8 & 0xffu);
    return;
  }
  bf16_t h, l;
  split_bf16(v, h, l);
  if (fmt == FMT_F16) h = (bf16_t)(cvt2h(v, 0.f) & 0xffffu);
  dst_hi[o] = h;
  if (il) dst_lo[o] = l;
}
hipError_t launch_pack_weight(const float* src, int C, int T, int Cp, const int* row_map, int rows_p, bf16_t* dst_hi,
                              bf16_t* dst_lo, int ldk, int k_off, hipStream_t s, int fmt) {
  if (!planes_ok(dst_hi, dst_lo) || (fmt == FMT_F16 && dst_lo) || (fmt == FMT_H8 && !dst_lo)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((T * Cp + 255) / 256, rows_p), dim3(256), 0, s, src, C, T, Cp, row_map,
                     rows_p, dst_hi, dst_lo, ldk, k_off, fmt);
  return hipGetLastError();
}

// ---------------------------------------------------------------- re-pack of MANY weights in one launch (training: once per pass)
// A training pass multiplies ~270 packed weights (two packs per parameter: forward and dgrad) and an optimizer step changes all of
// them.  Refreshing them one ns2_weight_update at a time cost 269 launches of ~6 us + 1-3 small PyTorch kernels each to bring the
// source into the pack's orientation (W^T, taps flipped, q | kv concatenated): 3.5 ms of a 100 ms d512 step and 5 ms of the
// HOST-bound 15 ms d128 step.  Here every part is a descriptor -- where its values live (the PARAMETER's own storage, any strides:
// a transposition or a tap flip is a stride) and which rectangle of which packed weight they fill -- in a table in device memory
// that is built once; a pass is ONE launch, block -> descriptor by binary search over the block prefix.
__global__ __launch_bounds__(256) void repack_kernel(const RepackDesc* tab, int n) {
  const long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const RepackDesc d = tab[lo];
  const long lb = b - d.block0;
  const int r = (int)(lb / d.chunks), ch = (int)(lb - (long)r * d.chunks);
  // a thread takes 4 adjacent columns of one tap (one 8 + 4 + 4 byte store of an FMT_H8 line instead of twelve 1-2 byte stores)
  const int c4 = (d.cols + 3) >> 2;
  const int q = ch * 256 + threadIdx.x;
  if (q >= d.T * c4) return;
  const int tap = q / c4, c = (q - tap * c4) * 4;
  const float* sp = d.src + (long)r * d.sr + (long)c * d.sc + (long)tap * d.st;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (c + e < d.cols) ? sp[(long)e * d.sc] : 0.f;
  const int rp = d.row0 + r, kc = tap * d.Cp + d.col0 + c;
  bf16_t* row = d.dst_hi + (long)rp * d.drs;
  // whole group inside the part and 4-aligned in the pack (Cp and col0 multiples of 4 in practice): vector store; else element by element
  if ((kc & 3) == 0 && c + 3 < d.cols) { store_cols4(row, kc, v[0], v[1], v[2], v[3], d.fmt, d.il != 0); return; }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (c + e >= d.cols) break;
    const int k1 = kc + e;
    if (d.fmt == FMT_H8) {
      uint32_t h16, h8, l8;
      cvt2_h8(v[e], 0.f, h16, h8, l8);
      bf16_t* line = row + ((k1 & ~31) << 1);
      line[k1 & 31] = (bf16_t)(h16 & 0xffffu);
      reinterpret_cast<unsigned char*>(line)[64 + (k1 & 31)] = (unsigned char)(h8 & 0xffu);
      reinterpret_cast<unsigned char*>(line)[96 + (k1 & 31)] = (unsigned char)(l8 & 0xffu);
    } else {
      bf16_t h, l;
      split_bf16(v[e], h, l);
      if (d.fmt == FMT_F16) h = (bf16_t)(cvt2h(v[e], 0.f) & 0xffffu);
      const long o = pcol(k1, d.il != 0);
      row[o] = h;
      if (d.il) row[o + 32] = l;
    }
  }
}
hipError_t launch_repack(const RepackDesc* tab, int n, long total_blocks, hipStream_t s) {
  if (!tab || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(repack_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s, tab, n);
  return hipGetLastError();
}

// ================================================================ EnCodec SEANet helpers (HFENC:81-347)
// The SEANet convolutions run on the GEMM family (channel-last rows, shifted-row taps).  What they need around the GEMM:
//   * the activation in front of every convolution (ELU, HFENC:285-347 nn.ELU()) and the conversion to operand planes;
//   * EnCodec pads with REFLECTION (pad_mode = "reflect", causal: everything on the left, HFENC:142-175) where the GEMM's
//     conv loader zero-fills: every utterance gets `prefix` extra rows in front, filled with the mirrored first samples
//     (row -j = row j), so the causal taps of the rows >= prefix never reach the zero fill; the GEMM's outputs for the prefix
//     rows are garbage and are never read;
//   * the first convolution has one input channel: its 7 taps are gathered into 7 columns here (im2col), so that it is a
//     Linear with K = 7 instead of 7 taps of a 32-column-padded single channel.
// x: fp32 [B, in_prefix + T, ldx] (channel-last; the first in_prefix rows of every utterance are skipped: x may be the output of
// a convolution over prefixed rows), out planes [B, prefix + T, ldo]; add: optional fp32 [B, T, lda] added before the ELU.

__global__ __launch_bounds__(256) void seanet_prep_kernel(const float* x, int ldx, int in_prefix, const float* add, int ldadd, int B,
                                                          long T, int C, int elu, int prefix, int im2col_k, bf16_t* out_hi,
                                                          bf16_t* out_lo, int ldo, int fmt) {
  const int chunks = ldo >> 2;
  const long rows = (long)B * (prefix + T);
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * chunks) return;
  const long orow = idx / chunks;
  const int c = (int)(idx - orow * chunks) * 4;
  const long b = orow / (prefix + T);
  const long p = orow - b * (prefix + T);                 // position inside the padded utterance
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  const long xrow0 = b * (in_prefix + T) + in_prefix;      // the input may itself be a convolution output with garbage prefix rows
  if (im2col_k > 0) {
    // column j of output row n = x_reflect[n - (k - 1) + j]   (single input channel, causal left padding k - 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = c + e;
      if (j < im2col_k) {
        long t = p - (im2col_k - 1) + j;
        if (t < 0) t = -t;                                // reflection: x[-t] = x[t]
        float v = x[(xrow0 + t) * ldx];
        o[e] = elu ? eluf(v) : v;
      }
    }
  } else {
    const long t = p >= prefix ? p - prefix : prefix - p;  // prefix row p mirrors row (prefix - p)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) {
        float v = x[(xrow0 + t) * ldx + c + e];
        if (add) v += add[(b * T + t) * ldadd + c + e];
        o[e] = elu ? eluf(v) : v;
      }
  }
  const bool il = out_lo != nullptr;
  store_cols4(out_hi + orow * pld(ldo, il), c, o[0], o[1], o[2], o[3], fmt, il);
}
hipError_t launch_seanet_prep(const float* x, int ldx, int in_prefix, const float* add, int ldadd, int B, long T, int C, int elu,
                              int prefix, int im2col_k, bf16_t* out_hi, bf16_t* out_lo, int ldo, int fmt, hipStream_t s) {
  if (B <= 0 || T <= 0 || C <= 0 || (ldo & 3) || prefix < 0 || prefix >= T || in_prefix < 0) return hipErrorInvalidValue;
  if (im2col_k > 0 ? (ldo < im2col_k || prefix != 0 || add) : ldo < C) return hipErrorInvalidValue;
  if (im2col_k > 0 && T < im2col_k) return hipErrorInvalidValue;      // the reflected index |p - (k - 1) + j| must stay inside the utterance
  if (!planes_ok(out_hi, out_lo) || (out_lo && (ldo & 31)) || (fmt == FMT_F16 && out_lo) || (fmt == FMT_H8 && !out_lo))
    return hipErrorInvalidValue;
  const long total = (long)B * (prefix + T) * (ldo >> 2);
  hipLaunchKernelGGL(seanet_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, in_prefix, add, ldadd, B, T, C,
                     elu, prefix, im2col_k, out_hi, out_lo, ldo, fmt);
  return hipGetLastError();
}

// dst[b][t][c] = src[b][prefix + t][c]  (drop the prefix rows of a GEMM output), optionally + add
__global__ void seanet_unpad_kernel(const float* src, long ld_src, int prefix, float* dst, long ld_dst, int B, long T, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T * C) return;
  const int c = (int)(i % C);
  const long r = i / C;
  const long b = r / T, t = r - b * T;
  dst[r * ld_dst + c] = src[(b * (prefix + T) + prefix + t) * ld_src + c];
}
// One pass over an fp32 activation for the two operands EnCodec's residual block needs of it: ELU(x) (conv1's input) and x
// itself (the shortcut's), each into a column window [col0, col0 + cols) of its own plane buffer -- so that x and the hidden
// activation can sit side by side in ONE operand and conv2 + shortcut become one GEMM over the concatenated K.  Same row
// layout as seanet_prep_kernel: `prefix` mirrored rows in front of every utterance.
struct PrepOut {
  bf16_t* hi;
  bf16_t* lo;
  int ld, col0, cols;
};
__global__ __launch_bounds__(256) void seanet_prep2_kernel(const float* x, int ldx, int in_prefix, int B, long T, int C, int prefix,
                                                           PrepOut eo, PrepOut ro, int chunks, int fmt) {
  const long rows = (long)B * (prefix + T);
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * chunks) return;
  const long orow = idx / chunks;
  const int c = (int)(idx - orow * chunks) * 4;
  const long b = orow / (prefix + T);
  const long p = orow - b * (prefix + T);
  const long t = p >= prefix ? p - prefix : prefix - p;    // prefix row p mirrors row (prefix - p)
  const float* xr = x + (b * (in_prefix + T) + in_prefix + t) * ldx;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (c + 3 < C && (ldx & 3) == 0) {
    const float4 q = *reinterpret_cast<const float4*>(xr + c);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < C) v[e] = xr[c + e];
  }
  if (eo.hi && c < eo.cols) {
    const bool il = fmt_il(fmt, eo.lo);
    store_cols4(eo.hi + orow * pld(eo.ld, il), eo.col0 + c, eluf(v[0]), eluf(v[1]), eluf(v[2]), eluf(v[3]), fmt, il);
  }
  if (ro.hi && c < ro.cols) {
    const bool il = fmt_il(fmt, ro.lo);
    store_cols4(ro.hi + orow * pld(ro.ld, il), ro.col0 + c, v[0], v[1], v[2], v[3], fmt, il);
  }
}
static bool prep_out_ok(const PrepOut& o, int C, int fmt) {
  if (!o.hi) return o.lo == nullptr;
  if (o.ld <= 0 || (o.ld & 31) || (o.col0 & 31) || o.col0 < 0 || (o.cols & 3) || o.cols < C || o.col0 + o.cols > o.ld) return false;
  return planes_ok(o.hi, o.lo) && !(fmt == FMT_F16 && o.lo) && !(fmt == FMT_H8 && !o.lo);
}
hipError_t launch_seanet_prep2(const float* x, int ldx, int in_prefix, int B, long T, int C, int prefix, bf16_t* elu_hi, bf16_t* elu_lo,
                               int elu_ld, int elu_col0, int elu_cols, bf16_t* raw_hi, bf16_t* raw_lo, int raw_ld, int raw_col0,
                               int raw_cols, int fmt, hipStream_t s) {
  if (B <= 0 || T <= 0 || C <= 0 || prefix < 0 || prefix >= T || in_prefix < 0 || (!elu_hi && !raw_hi)) return hipErrorInvalidValue;
  const PrepOut eo{elu_hi, elu_lo, elu_ld, elu_col0, elu_cols}, ro{raw_hi, raw_lo, raw_ld, raw_col0, raw_cols};
  if (!prep_out_ok(eo, C, fmt) || !prep_out_ok(ro, C, fmt)) return hipErrorInvalidValue;
  const int cols = (elu_hi ? elu_cols : 0) > (raw_hi ? raw_cols : 0) ? elu_cols : raw_cols;
  const long total = (long)B * (prefix + T) * (cols >> 2);
  hipLaunchKernelGGL(seanet_prep2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, in_prefix, B, T, C, prefix, eo, ro,
                     cols >> 2, fmt);
  return hipGetLastError();
}

// ---- the two ends of the SEANet stacks: 1 -> co channels (the encoder's first convolution) and ci -> 1 (the decoder's last).
// As GEMMs they padded a K of 7 to 32 (plus an im2col pass) and an N of 1 to a 64-column wave tile with the operand re-read once
// per tap; here they are what they are -- 7 x co and 7 x ci multiply-adds per sample in fp32 on the vector ALUs, one pass over the
// wide side of the layer, causal with EnCodec's reflect padding (HFENC:142-175), optional ELU on the input (HFENC:285-347).
template <int K>
__global__ __launch_bounds__(256) void seanet_conv_in_kernel(const float* x, long ldx, int in_prefix, int B, long T, int co, int elu,
                                                             const float* w, const float* bias, float* out, long ldo, int iters) {
  const int cg = co >> 2;                          // threads per output row, 4 channels each: whole 16-byte stores, 128 B per row at co = 32
  const int c4 = (threadIdx.x % cg) * 4, rslot = threadIdx.x / cg, rpb = 256 / cg;
  float wr[4][K], b4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b4[e] = bias ? bias[c4 + e] : 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) wr[e][j] = w[(c4 + e) * K + j];
  }
  for (int it = 0; it < iters; ++it) {
    const long row = ((long)blockIdx.x * iters + it) * rpb + rslot;
    if (row >= (long)B * T) break;
    const long b = row / T, t = row - b * T;
    const float* xb = x + (b * (in_prefix + T) + in_prefix) * ldx;
    float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
    for (int j = 0; j < K; ++j) {
      long tt = t - (K - 1) + j;
      if (tt < 0) tt = -tt;                        // reflection: x[-t] = x[t]
      float v = xb[tt * ldx];
      if (elu) v = eluf(v);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(wr[e][j], v, acc[e]);
    }
    *reinterpret_cast<float4*>(out + row * ldo + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}
template <int K>
__global__ __launch_bounds__(256) void seanet_conv_out_kernel(const float* x, long ldx, int in_prefix, int B, long T, int ci, int elu,
                                                              const float* w, const float* bias, float* out, long ldo, int R) {
  const int cg = ci >> 2;                          // lanes per row (a power of two): 16 bytes each, whole 128-B rows at ci = 32
  const int c4 = (threadIdx.x % cg) * 4;
  const long grp = ((long)blockIdx.x * 256 + threadIdx.x) / cg;     // a run of R consecutive rows of one utterance
  const long gpu = (T + R - 1) / R;
  const long b = grp / gpu;
  if (b >= B) return;                              // whole groups leave together
  const long t0 = (grp - b * gpu) * R;
  const float* xb = x + (b * (in_prefix + T) + in_prefix) * ldx + c4;
  float wv[4][K];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int j = 0; j < K; ++j) wv[e][j] = w[(c4 + e) * K + j];
  const float b0 = bias ? bias[0] : 0.f;
  auto load = [&](long tt) {
    if (tt < 0) tt = -tt;
    float4 v = *reinterpret_cast<const float4*>(xb + tt * ldx);
    if (elu) { v.x = eluf(v.x); v.y = eluf(v.y); v.z = eluf(v.z); v.w = eluf(v.w); }
    return v;
  };
  float4 win[K];                                   // rows t - (K - 1) .. t of this lane's 4 channels, after the ELU
#pragma unroll
  for (int j = 0; j + 1 < K; ++j) win[j] = load(t0 - (K - 1) + j);
  for (int r = 0; r < R && t0 + r < T; ++r) {
    win[K - 1] = load(t0 + r);
    float p = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      p = fmaf(wv[0][j], win[j].x, p); p = fmaf(wv[1][j], win[j].y, p); p = fmaf(wv[2][j], win[j].z, p); p = fmaf(wv[3][j], win[j].w, p);
    }
    for (int m = cg >> 1; m > 0; m >>= 1) p += __shfl_xor(p, m, 64);
    if (c4 == 0) out[(b * T + t0 + r) * ldo] = p + b0;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) win[j] = win[j + 1];
  }
}
// hipErrorNotReady: not one of the two shapes (the caller takes the GEMM path)
hipError_t launch_seanet_conv_narrow(const float* x, long ldx, int in_prefix, int B, long T, int ci, int co, int k, int elu, const float* w,
                                     const float* bias, float* out, long ldo, hipStream_t s) {
  if (B <= 0 || T <= 0 || ci <= 0 || co <= 0 || in_prefix < 0 || !x || !w || !out) return hipErrorInvalidValue;
  if (k != 7 || T < k) return hipErrorNotReady;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (ci == 1 && (co & 3) == 0 && pow2(co >> 2) && co <= 256 && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int rpb = 256 / (co >> 2), iters = 8;
    const long blocks = ((long)B * T + (long)rpb * iters - 1) / ((long)rpb * iters);
    hipLaunchKernelGGL(seanet_conv_in_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, in_prefix, B, T, co, elu, w, bias, out, ldo,
                       iters);
    return hipGetLastError();
  }
  if (co == 1 && (ci & 3) == 0 && pow2(ci >> 2) && ci <= 256 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int R = 32, cg = ci >> 2;
    const long groups = (long)B * ((T + R - 1) / R);
    const long blocks = (groups * cg + 255) / 256;
    hipLaunchKernelGGL(seanet_conv_out_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, in_prefix, B, T, ci, elu, w, bias, out, ldo,
                       R);
    return hipGetLastError();
  }
  return hipErrorNotReady;
}

// ---- EnCodec's residual block at its narrow end (HFENC:268-301: y = shortcut(x) + conv2(elu(conv1(elu(x)))), conv1 k = 3 causal with
// reflect padding, conv2 and the shortcut 1 x 1; hidden = C / 2) as ONE pass on the vector ALUs.  At C = 32 the block runs at the
// full 24 kHz rate -- 10.5 M rows for 32 utterances of 13.6 s -- and as GEMMs it was four HBM round trips of 1.3 GB each (two
// operand-plane passes + two products that use 16 / 32 of the 64 columns of a wave tile): 3.4 ms.  Here a workgroup stages 256 + 2
// fp32 rows in LDS with coalesced loads (row stride 144 B: the per-thread 16-byte reads down a column are conflict-free), a thread
// computes one output row -- 3 C H + H C + C C = 3072 fp32 FMAs at C = 32 -- with the weights as SCALAR operands (they are the same
// for every lane: packed in consumption order, s_load straight into the FMA), the row goes back through LDS and leaves coalesced:
// x is read once and y written once.  Plain fp32 FMAs: closer to HF's own arithmetic than the bf16 x3 products.
// w1p [3][C][H] (tap, in, hidden), w2p [H][C] (hidden, out), wsp [C][C] (in, out); rows of x: [B, in_prefix + T, ldx]
// ELU for the VALU-bound kernel below: v_exp_f32 instead of the library's expm1f (a ~40-instruction polynomial: 112 of them per row
// would outweigh the 3072 FMAs).  exp(x) - 1 cancels for small |x|; there the cubic Taylor form takes over (|error| < 1e-9 below 2^-6),
// so the absolute error stays at fp32 rounding level everywhere.
NS2_DEVINL float elu_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.0f;
  const float p = x * fmaf(x, fmaf(x, 0.16666667f, 0.5f), 1.0f);
  const float neg = x > -0.015625f ? p : e;
  return x > 0.f ? x : neg;
}

template <int C, int RPT>
__global__ __launch_bounds__(256) void seanet_resblock_narrow_kernel(const float* __restrict__ x, long ldx, int in_prefix, int B, long T,
                                                                     const float* __restrict__ w1p, const float* __restrict__ b1,
                                                                     const float* __restrict__ w2p, const float* __restrict__ wsp,
                                                                     const float* __restrict__ b2s, float* __restrict__ out, long ldo) {
  // RPT rows per thread (rows tid, tid + 256, ...): every weight fetched from the scalar cache serves RPT FMAs
  constexpr int H = C / 2, RS = C + 4, NR = 256 * RPT; // LDS row stride in floats: 16-byte aligned, 4-bank skew per row
  extern __shared__ __attribute__((aligned(16))) float tile[];      // (NR + 2) * RS floats
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * NR;               // first output row (flattened b * T + n) of this workgroup
  const long rows_total = (long)B * T;
  auto src_row = [&](long r) -> const float* {         // flattened row -> its fp32 row in the input layout
    const long b = r / T, n = r - b * T;
    return x + (b * (in_prefix + T) + in_prefix + n) * ldx;
  };
  // ---- stage rows r0 - 2 ... r0 + NR - 1 (tile row j = row r0 - 2 + j), C / 4 float4 per row, coalesced
  for (int i = tid; i < (NR + 2) * (C / 4); i += 256) {
    const int j = i / (C / 4), q = i - j * (C / 4);
    const long r = r0 - 2 + j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 0 && r < rows_total) v = *reinterpret_cast<const float4*>(src_row(r) + 4 * q);
    *reinterpret_cast<float4*>(&tile[j * RS + 4 * q]) = v;
  }
  __syncthreads();
  float y[RPT][C], h[RPT][H];
  bool live[RPT];
  long nn[RPT], bb[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const long r = r0 + tid + 256 * k;
    live[k] = r < rows_total;
    const long rc = live[k] ? r : rows_total - 1;      // (a dead row computes on valid data and is not stored)
    bb[k] = rc / T; nn[k] = rc - bb[k] * T;
#pragma unroll
    for (int o = 0; o < H; ++o) h[k][o] = b1[o];
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    // tap t reads x[n - 2 + t]; before the utterance start EnCodec reflects (x[-i] = x[i], HFENC:142-175): those two rows per
    // utterance come from global memory, everything else from the staged tile
    float xv[RPT][C];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const long i = nn[k] - 2 + t;
      const float* p = (i >= 0 && live[k]) ? &tile[(tid + 256 * k + t) * RS] : x + (bb[k] * (in_prefix + T) + in_prefix + (i >= 0 ? i : -i)) * ldx;
      if (i >= 0 && live[k]) {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);                 // (LDS)
          xv[k][4 * q] = v.x; xv[k][4 * q + 1] = v.y; xv[k][4 * q + 2] = v.z; xv[k][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);                 // (global)
          xv[k][4 * q] = v.x; xv[k][4 * q + 1] = v.y; xv[k][4 * q + 2] = v.z; xv[k][4 * q + 3] = v.w;
        }
      }
    }
    if (t == 2) {                                      // the unshifted row: also the shortcut's input (raw)
#pragma unroll
      for (int k = 0; k < RPT; ++k)
#pragma unroll
        for (int o = 0; o < C; ++o) y[k][o] = b2s[o];
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int o = 0; o < C; ++o) {
          const float w = wsp[c * C + o];
#pragma unroll
          for (int k = 0; k < RPT; ++k) y[k][o] = fmaf(w, xv[k][c], y[k][o]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float e[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) e[k] = elu_fast(xv[k][c]);
#pragma unroll
      for (int o = 0; o < H; ++o) {
        const float w = w1p[(t * C + c) * H + o];
#pragma unroll
        for (int k = 0; k < RPT; ++k) h[k][o] = fmaf(w, e[k], h[k][o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < H; ++o) {
    float e[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) e[k] = elu_fast(h[k][o]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float w = w2p[o * C + c];
#pragma unroll
      for (int k = 0; k < RPT; ++k) y[k][c] = fmaf(w, e[k], y[k][c]);
    }
  }
  __syncthreads();                                     // every read of the staged input is done: the tile becomes the output stage
#pragma unroll
  for (int k = 0; k < RPT; ++k)
#pragma unroll
    for (int q = 0; q < C / 4; ++q)
      *reinterpret_cast<float4*>(&tile[(tid + 256 * k) * RS + 4 * q]) = make_float4(y[k][4 * q], y[k][4 * q + 1], y[k][4 * q + 2], y[k][4 * q + 3]);
  __syncthreads();
  for (int i = tid; i < NR * (C / 4); i += 256) {
    const int j = i / (C / 4), q = i - j * (C / 4);
    if (r0 + j < rows_total) *reinterpret_cast<float4*>(out + (r0 + j) * ldo + 4 * q) = *reinterpret_cast<const float4*>(&tile[j * RS + 4 * q]);
  }
}
// hipErrorNotReady: not a shape this kernel serves (the caller takes the GEMM path)
hipError_t launch_seanet_resblock_narrow(const float* x, long ldx, int in_prefix, int B, long T, int C, const float* w1p, const float* b1,
                                         const float* w2p, const float* wsp, const float* b2s, float* out, long ldo, hipStream_t s) {
  if (B <= 0 || T <= 0 || in_prefix < 0 || !x || !w1p || !b1 || !w2p || !wsp || !b2s || !out) return hipErrorInvalidValue;
  if (C != 32 || T < 3 || (ldx & 3) || (ldo & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15)) return hipErrorNotReady;
  constexpr int RPT = 1;                             // (2 rows per thread measured no fewer instructions per row: the packed FMAs become plain ones)
  const long blocks = ((long)B * T + 256 * RPT - 1) / (256 * RPT);
  const size_t lds = (size_t)(256 * RPT + 2) * (32 + 4) * sizeof(float);       // 36.3 KiB
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(reinterpret_cast<const void*>(&seanet_resblock_narrow_kernel<32, RPT>), (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((seanet_resblock_narrow_kernel<32, RPT>), dim3((unsigned)blocks), dim3(256), lds, s, x, ldx, in_prefix, B, T, w1p, b1, w2p, wsp, b2s, out, ldo);
  return hipGetLastError();
}

hipError_t launch_seanet_unpad(const float* src, long ld_src, int prefix, float* dst, long ld_dst, int B, long T, int C, hipStream_t s) {
  const long n = (long)B * T * C;
  hipLaunchKernelGGL(seanet_unpad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, prefix, dst, ld_dst, B, T, C);
  return hipGetLastError();
}

// ---- LSTM (HFENC:253-266: nn.LSTM(dim, dim, num_layers), PyTorch gate order i, f, g, o).  The input projections of all time
// steps are one GEMM; this kernel is ONE recurrent step of one layer:  gates = xproj[b, t] + h_prev[b] . W_hh^T + b_hh ;
// c = sigma(f) c + sigma(i) tanh(g) ; h = sigma(o) tanh(c).  A workgroup owns 16 hidden units x 16 batch rows;
// W_hh rows and h_prev come from L2 (the step is latency-bound: 2 x T dependent launches per utterance batch).  This is the
// fallback (H != 512, B > 32, or a caller that passes only the minimal scratch); the persistent kernel below is the default.
constexpr int LSTM_UNITS = 16;                    // hidden units per workgroup
constexpr int LSTM_ROWS = 16;                     // batch rows per workgroup
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* xproj, long ld_x, long row_stride_t, long t, const float* w_hh,
                                                        const float* b_hh, const float* h_prev, float* h_next, float* c_state,
                                                        const float* resid, long ld_r, float* out, long ld_o, int B, int H) {
  __shared__ __attribute__((aligned(16))) float s_h[LSTM_ROWS][512 + 4];       // +16 B per row: conflict-free 16-B reads down a column
  const int b0 = blockIdx.y * LSTM_ROWS;
  const int nb = min(LSTM_ROWS, B - b0);
  for (int i = threadIdx.x; i < LSTM_ROWS * H; i += 256) {
    const int b = i / H, k = i - b * H;
    s_h[b][k] = b < nb ? h_prev[(long)(b0 + b) * H + k] : 0.f;
  }
  __syncthreads();
  // thread -> (batch row b = tid & 15, unit u = tid >> 4): 4 gate dot products of length H, accumulated as 4 independent chains
  const int b = threadIdx.x & 15, u = threadIdx.x >> 4;
  const int j = blockIdx.x * LSTM_UNITS + u;
  if (j >= H || b >= nb) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < H; k += 4) {
    const float4 hv = *reinterpret_cast<const float4*>(&s_h[b][k]);
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
      const float4 wv = *reinterpret_cast<const float4*>(w_hh + ((long)gate * H + j) * H + k);   // shared by the 16 rows of a unit
      acc[gate] = fmaf(wv.x, hv.x, acc[gate]); acc[gate] = fmaf(wv.y, hv.y, acc[gate]);
      acc[gate] = fmaf(wv.z, hv.z, acc[gate]); acc[gate] = fmaf(wv.w, hv.w, acc[gate]);
    }
  }
  const long row = (long)(b0 + b) * row_stride_t + t;
  float g4[4];
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) g4[gate] = acc[gate] + xproj[row * ld_x + (long)gate * H + j] + b_hh[gate * H + j];
  const long sidx = (long)(b0 + b) * H + j;
  const float ig = sigmoidf_acc(g4[0]), fg = sigmoidf_acc(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_acc(g4[3]);
  const float c = fg * c_state[sidx] + ig * gg;
  const float h = og * tanhf(c);
  c_state[sidx] = c;
  h_next[sidx] = h;
  out[row * ld_o + j] = h + (resid ? resid[row * ld_r + j] : 0.f);
}
// ---- persistent recurrence: ONE launch per layer instead of T.  H = 512 only (EnCodec's LSTM width); 64 workgroups per group
// of 8 batch rows, each owning 8 hidden units = 32 gate rows of W_hh, which live in REGISTERS for the whole sequence: thread
// (u = tid >> 5, kq = tid & 31) holds the four gate rows of unit u, columns 4 kq + 128 i .. + 3 (i < 4).  Per step a workgroup
// stages h_{t-1} [8, 512] into LDS, every thread forms its 4 x 8 partial dot products (32 LDS reads of 16 bytes: round 3's first
// mapping, one gate row and 64 columns per thread, read 128 and spent 1.9 us of every step on LDS bandwidth), a five-stage
// transpose-reduction over the 32 kq lanes leaves lane kq with the complete sum of gate kq >> 3, batch row kq & 7, the four gates
// meet through three lane exchanges, and the gate-0 lanes apply the cell update (the cell state never leaves their registers).
// The XCDs' L2s are not coherent with each other, so h travels through global memory with agent-scope (sc1) loads and stores
// -- individually coherent accesses of the few KB that are shared -- instead of release / acquire fences, whose L2 write-back
// and invalidate cost 10 us per step (measured: 16.6 us per step with fences).  The next step's input projections are
// fetched before the wait: they do not depend on the recurrence.
// Step synchronisation: every exchanged value is an 8-byte {h, step tag} pair written and read as ONE 64-bit agent-scope access,
// and a consumer simply re-reads the pairs whose tag is not yet the step it needs.  A step then costs one store -> load trip
// through the memory side instead of three dependent ones (store + wait for the write acknowledge, atomic arrival, counter
// poll): the counter barrier this replaces measured 7.5 us per frame, 36.0 ms per encode against 32.8 (tools/gpu_r3_k.sh).
// Two exchange buffers suffice: a workgroup can write h_{t+2} into the buffer that held h_t only after it has read every
// h_{t+1}, and those are written by workgroups that have finished reading h_t.  Tags are t + 1 (never 0 = the host's memset).
constexpr int LP_UNITS = 8;
constexpr int LP_LDH = 512 + 4;                   // LDS row stride of the staged h (floats)
constexpr int LP_MAXB = 32;
constexpr unsigned LP_SPIN_LIMIT = 1u << 20;      // ~ seconds of polling: a lost workgroup turns into a reported abort, not a hang
// launches whose step wait timed out (a workgroup that never became resident: CU masking, a partitioned device, a GPU
// saturated by other processes).  The kernel then gives up -- every workgroup leaves at its next wait -- instead of trapping
// (a trap kills the HIP context and the process); the host reads this counter after a codec run (ns2_lstm_abort_count).
__device__ unsigned int ns2_lstm_aborts;

NS2_DEVINL unsigned long long ld_agent_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NS2_DEVINL void st_agent_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- pieces shared by the one-layer and the two-layer kernel.  A "block" of h is 8 batch rows x 512 {h, tag} pairs,
// contiguous in global memory; pair q of a thread is element tid + 256 q of it (row q >> 1).
struct LstmWait {
  unsigned* abort_flag;
  int gave_up = 0;
};
// Stage NS blocks into LDS (block n -> s_h + n * 8 * LP_LDH), each once every lane of the wave sees tag want[n] in it;
// zero[n]: the block is h_{-1} = 0, nothing is read.  Bit 16 n + q of `pend` (wave-uniform) = that pair has not been staged
// yet: every round reads the pending pairs, and a pair goes to LDS in the round in which all 64 lanes see the wanted tag --
// nothing but the mask is carried between rounds.
template <int NS>
NS2_DEVINL void lstm_stage(const unsigned long long* const (&src)[NS], const unsigned (&want)[NS], const bool (&zero)[NS], float* s_h,
                           int rows, int tid, LstmWait& wt) {
  unsigned pend = 0;
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if ((q >> 1) >= rows) continue;              // rows past the batch are never read by the dot products
      if (zero[n]) s_h[(8 * n + (q >> 1)) * LP_LDH + 256 * (q & 1) + tid] = 0.f;
      else pend |= 1u << (16 * n + q);
    }
  unsigned spins = 0;
  while (pend) {
    unsigned long long v[16 * NS];
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (pend & (1u << (16 * n + q))) v[16 * n + q] = ld_agent_u64(src[n] + tid + 256 * q);
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if ((pend & (1u << (16 * n + q))) && !__builtin_amdgcn_ballot_w64((unsigned)(v[16 * n + q] >> 32) != want[n])) {
          s_h[(8 * n + (q >> 1)) * LP_LDH + 256 * (q & 1) + tid] = __uint_as_float((unsigned)v[16 * n + q]);
          pend &= ~(1u << (16 * n + q));
        }
    if (!pend) break;
    if (((++spins) & 255u) == 0) {
      if (__hip_atomic_load(wt.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { wt.gave_up = 1; break; }
      if (spins > LP_SPIN_LIMIT) {                 // first to time out: tell the others, count the aborted launch once
        if (__hip_atomic_exchange(wt.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicAdd(&ns2_lstm_aborts, 1u);
        wt.gave_up = 1;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(2);
  }
}
struct LstmRows { float4 w[4][4]; };              // [gate][i]: columns 4 kq + 128 i .. + 3 of row gate * 512 + j of a [2048, 512] matrix
NS2_DEVINL void lstm_load_rows(LstmRows& r, const float* w, int j, int kq) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[g][i] = *reinterpret_cast<const float4*>(w + ((long)g * 512 + j) * 512 + 4 * kq + 128 * i);
}
NS2_DEVINL void lstm_dot(const LstmRows& r, const float* s_blk, int rows, int kq, float (&acc)[4][8]) {     // acc += rows of r . block
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) {
    if (bb >= rows) continue;
    const float* hb = s_blk + bb * LP_LDH + 4 * kq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 h4 = *reinterpret_cast<const float4*>(hb + 128 * i);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a = acc[g][bb];
        a = fmaf(r.w[g][i].x, h4.x, a); a = fmaf(r.w[g][i].y, h4.y, a); a = fmaf(r.w[g][i].z, h4.z, a); a = fmaf(r.w[g][i].w, h4.w, a);
        acc[g][bb] = a;
      }
    }
  }
}
// Transpose-reduction over the 32 kq lanes of a unit through LDS: lane kq ends with the complete sum of value m = kq, i.e. gate
// kq >> 3, batch row kq & 7.  s_red = this unit's [32 values][LP_RED lanes] floats, rows padded to 33: the writes (fixed m, 32
// lanes) and the reads (lane m walks its row) both touch 32 different banks, and every address is one base register plus an
// immediate.  The two units of a wave use their own regions and a wave's LDS operations execute in order: no workgroup barrier.
// (The register-only version -- five rounds of select + lane exchange -- cost ~100 VGPRs next to the packed accumulators and
// pushed the two-layer kernel into scratch.)
constexpr int LP_RED = 33;
NS2_DEVINL float lstm_reduce(const float (&acc)[4][8], float* s_red, int kq) {
  float* wr = s_red + kq;
#pragma unroll
  for (int m = 0; m < 32; ++m) wr[m * LP_RED] = acc[m >> 3][m & 7];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* rd = s_red + kq * LP_RED;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i & 3] += rd[i];
  __builtin_amdgcn_wave_barrier();                 // the region is rewritten by the next reduction
  return (r[0] + r[1]) + (r[2] + r[3]);
}
// lane kq holds x = pre-activation of gate kq >> 3 (i, f, g, o) for batch row kq & 7; lanes kq ^ 8, ^ 16, ^ 24 hold the other
// three gates.  Every lane must call (lane exchanges); the result is meaningful on the gate-0 lanes, which own the cell state.
NS2_DEVINL float lstm_cell(float x, float& c) {
  const float x1 = __shfl_xor(x, 8, 64), x2 = __shfl_xor(x, 16, 64), x3 = __shfl_xor(x, 24, 64);
  const float ig = sigmoidf_acc(x), fg = sigmoidf_acc(x1), gg = tanhf(x2), og = sigmoidf_acc(x3);
  c = fg * c + ig * gg;
  return og * tanhf(c);
}
NS2_DEVINL unsigned long long lstm_pair(float h, long t) { return ((unsigned long long)(unsigned)(t + 1) << 32) | __float_as_uint(h); }

// Batch rows are independent through the recurrence: blockIdx.y selects a group of 8 BG rows that synchronises only with itself
// (64 workgroups each).  B = 32 runs as four 8-row groups side by side (256 workgroups, one per CU) instead of one group of
// 32 rows -- the per-step cost of a 32-row group (h staging 64 KiB per workgroup + 2048 FMAs per thread) was four times the 8-row step.
template <int BG>                                 // batch rows per group in units of 8
__global__ __launch_bounds__(256, 2) void lstm_persistent_kernel(const float* xproj, long ld_x, long T, const float* w_hh, const float* b_hh,
                                                              unsigned long long* hx, const float* resid, long ld_r, float* out, long ld_o,
                                                              int B_all, unsigned* flags_all) {
  constexpr int H = 512;
  extern __shared__ __attribute__((aligned(16))) float s_h[];      // [8 BG][LP_LDH] staged h, then [8 units][32][32] reduction scratch
  const int tid = threadIdx.x, kq = tid & 31, u = tid >> 5;
  const int gate = kq >> 3, bl = kq & 7;                           // this lane's role after the reduction
  float* s_red = s_h + 8 * BG * LP_LDH + u * (32 * LP_RED);
  const int j = blockIdx.x * LP_UNITS + u;                         // hidden unit
  // this group's rows [b_off, b_off + 8 BG): shift every per-row pointer, keep the group-local indexing below
  const int b_off = blockIdx.y * 8 * BG;
  const int B = min(B_all - b_off, 8 * BG);
  LstmWait wt{flags_all + 16 * blockIdx.y + 1};                     // one 64-byte line per group
  xproj += (long)b_off * T * ld_x;
  hx += (long)b_off * H;                                           // {h, tag} pairs, [2][LP_MAXB][H]
  out += (long)b_off * T * ld_o;
  if (resid) resid += (long)b_off * T * ld_r;

  LstmRows wr;
  lstm_load_rows(wr, w_hh, j, kq);
  const float bias = b_hh[gate * H + j];
  float c[BG], xp[BG];
#pragma unroll
  for (int g = 0; g < BG; ++g) {
    c[g] = 0.f;
    const int b = 8 * g + bl;
    xp[g] = (b < B) ? xproj[((long)b * T) * ld_x + (long)gate * H + j] : 0.f;      // step 0's input projection
  }

  for (long t = 0; t < T; ++t) {
    // ---- stage h_{t-1} (zeros at t = 0): written at step t - 1 with tag t
    const unsigned long long* hp = hx + (t & 1) * (long)LP_MAXB * H;
#pragma unroll
    for (int g = 0; g < BG; ++g) {
      const unsigned long long* const src[1] = {hp + (long)(8 * g) * H};
      const unsigned want[1] = {(unsigned)t};
      const bool zero[1] = {t == 0};
      lstm_stage<1>(src, want, zero, s_h + 8 * g * LP_LDH, B - 8 * g, tid, wt);
    }
    if (__syncthreads_or(wt.gave_up)) return;      // uniform: the whole workgroup leaves
    // ---- partial dot products over this thread's 16 columns: four gate rows x all batch rows, then the transpose-reduction
    float pre[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) {
      float acc[4][8] = {};
      lstm_dot(wr, s_h + 8 * g * LP_LDH, B - 8 * g, kq, acc);
      pre[g] = lstm_reduce(acc, s_red, kq);
    }
    __syncthreads();                               // s_h is free for the next step's staging
    unsigned long long* hn = hx + ((t + 1) & 1) * (long)LP_MAXB * H;
#pragma unroll
    for (int g = 0; g < BG; ++g) {
      const int b = 8 * g + bl;
      const float x = pre[g] + bias + xp[g];
      if (b < B && t + 1 < T) xp[g] = xproj[((long)b * T + t + 1) * ld_x + (long)gate * H + j];   // in flight across the wait
      const float h = lstm_cell(x, c[g]);
      if (gate == 0 && b < B) {
        st_agent_u64(hn + (long)b * H + j, lstm_pair(h, t));
        const long row = (long)b * T + t;
        out[row * ld_o + j] = h + (resid ? resid[row * ld_r + j] : 0.f);
      }
    }
  }
}

// ---- both layers of a 2-layer nn.LSTM in ONE launch: the layers overlap instead of following each other (the recurrences are
// latency-bound: a step is mostly the store -> load trip of h), and the [B T, 512] x [512, 2048] GEMM between them disappears.
// blockIdx.z = 0, layer 1: as above, and -- from the h1_{t-1} block it has staged anyway -- the layer-2 input projection
// W_ih2 h1_{t-1} + b_ih2 of ITS OWN 8 units x 4 gates x 8 batch rows (a second set of 64 weights per thread, the same LDS reads),
// one value per thread after the reduction.  blockIdx.z = 1, layer 2: the one-layer recurrence on W_hh2, its input projections
// coming lane-to-lane from the layer-1 workgroup of the same units (blockIdx.x, blockIdx.y) through a ring of LP_XDEPTH tagged
// slots per thread.  Layer 1 does not depend on layer 2 and could run ahead without bound: before it reuses a ring slot it checks
// the consumer workgroup's acknowledge word (the last frame that workgroup has taken in; read at the top of the step, normally
// far ahead of what is required).  All 128 workgroups of a group must be resident.
constexpr int LP_XDEPTH = 32;
__global__ __launch_bounds__(256, 2) void lstm2_persistent_kernel(const float* xproj, long ld_x, long T, const float* w_hh1,
                                                                  const float* b_hh1, const float* w_ih2, const float* b_ih2,
                                                                  const float* w_hh2, const float* b_hh2, unsigned long long* h1,
                                                                  unsigned long long* h2, unsigned long long* xs, unsigned* acks,
                                                                  const float* resid, long ld_r, float* out, long ld_o, int B_all,
                                                                  unsigned* flags_all) {
  constexpr int H = 512;
  extern __shared__ __attribute__((aligned(16))) float s_h[];      // [8][LP_LDH] staged h, then [8 units][32][32] reduction scratch
  const int tid = threadIdx.x, kq = tid & 31, u = tid >> 5;
  const int gate = kq >> 3, bl = kq & 7;
  float* s_red = s_h + 8 * LP_LDH + u * (32 * LP_RED);
  const int j = blockIdx.x * LP_UNITS + u;
  const int b_off = blockIdx.y * 8;
  const int B = min(B_all - b_off, 8);
  const bool mine = gate == 0 && bl < B;                            // this lane owns (batch row b_off + bl, unit j)
  LstmWait wt{flags_all + 16 * blockIdx.y + 1};
  const long pair_wg = blockIdx.y * gridDim.x + blockIdx.x;         // producer / consumer workgroup pair
  xs += pair_wg * ((long)LP_XDEPTH * 256) + tid;                    // this thread's ring: slot s at + 256 s
  unsigned* ack = acks + pair_wg;
  float c = 0.f;

  if (blockIdx.z == 0) {
    h1 += (long)b_off * H;                                         // ring [2][LP_MAXB][H]
    xproj += (long)b_off * T * ld_x;
    LstmRows wr, wi;
    lstm_load_rows(wr, w_hh1, j, kq);
    lstm_load_rows(wi, w_ih2, j, kq);
    const float bias = b_hh1[gate * H + j], bias2 = b_ih2[gate * H + j];
    float xp = (bl < B) ? xproj[((long)bl * T) * ld_x + (long)gate * H + j] : 0.f;
    for (long t = 0; t <= T; ++t) {                // step T only forms the last input projection of layer 2
      const unsigned acked = t >= LP_XDEPTH ? __hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      const unsigned long long* const src[1] = {h1 + (t & 1) * (long)LP_MAXB * H};
      const unsigned want[1] = {(unsigned)t};
      const bool zero[1] = {t == 0};
      lstm_stage<1>(src, want, zero, s_h, B, tid, wt);
      if (__syncthreads_or(wt.gave_up)) return;
      // the recurrence first -- h1_t is what the other workgroups are waiting for -- then, off that path, layer 2's projection
      if (t < T) {
        float acc[4][8] = {};
        lstm_dot(wr, s_h, B, kq, acc);
        const float x = lstm_reduce(acc, s_red, kq) + bias + xp;
        if (bl < B && t + 1 < T) xp = xproj[((long)bl * T + t + 1) * ld_x + (long)gate * H + j];
        const float h = lstm_cell(x, c);
        if (mine) st_agent_u64(h1 + ((t + 1) & 1) * (long)LP_MAXB * H + (long)bl * H + j, lstm_pair(h, t));
      }
      if (t > 0) {                                 // W_ih2 h1_{t-1} + b_ih2 -> frame t - 1 of layer 2, ring slot (t - 1) % LP_XDEPTH
        float acc2[4][8] = {};
        lstm_dot(wi, s_h, B, kq, acc2);
        const float x2 = lstm_reduce(acc2, s_red, kq) + bias2;
        if (t - 1 >= LP_XDEPTH) {                  // the slot still holds frame t - 1 - LP_XDEPTH until the consumer has taken it in
          unsigned a = acked, spins = 0;
          while (a < (unsigned)(t - LP_XDEPTH) && !wt.gave_up) {
            __builtin_amdgcn_s_sleep(8);
            a = __hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > LP_SPIN_LIMIT || __hip_atomic_load(wt.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) wt.gave_up = 1;
          }
        }
        st_agent_u64(xs + 256 * ((t - 1) % LP_XDEPTH), lstm_pair(x2, t - 1));
      }
      __syncthreads();                             // s_h is free for the next step's staging
    }
  } else {
    h2 += (long)b_off * H;                                         // ring [2][LP_MAXB][H]
    out += (long)b_off * T * ld_o;
    if (resid) resid += (long)b_off * T * ld_r;
    LstmRows wr;
    lstm_load_rows(wr, w_hh2, j, kq);
    const float bias = b_hh2[gate * H + j];
    for (long t = 0; t < T; ++t) {
      // this lane's input projection of frame t (tag t + 1: normally long there) rides along with the wait for h2_{t-1}
      const unsigned long long* xsrc = xs + 256 * (t % LP_XDEPTH);
      unsigned long long xv = ld_agent_u64(xsrc);
      const unsigned long long* const src[1] = {h2 + (t & 1) * (long)LP_MAXB * H};
      const unsigned want[1] = {(unsigned)t};
      const bool zero[1] = {t == 0};
      lstm_stage<1>(src, want, zero, s_h, B, tid, wt);
      for (unsigned spins = 0; (unsigned)(xv >> 32) != (unsigned)(t + 1) && !wt.gave_up;) {
        __builtin_amdgcn_s_sleep(2);
        xv = ld_agent_u64(xsrc);
        if (++spins > LP_SPIN_LIMIT || (((spins & 255u) == 0) && __hip_atomic_load(wt.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))
          wt.gave_up = 1;
      }
      if (__syncthreads_or(wt.gave_up)) return;
      if (tid == 0) __hip_atomic_store(ack, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // frames 0 .. t are taken in
      float acc[4][8] = {};
      lstm_dot(wr, s_h, B, kq, acc);
      const float x = lstm_reduce(acc, s_red, kq) + bias + __uint_as_float((unsigned)xv);
      __syncthreads();
      const float h = lstm_cell(x, c);
      if (mine) {
        st_agent_u64(h2 + ((t + 1) & 1) * (long)LP_MAXB * H + (long)bl * H + j, lstm_pair(h, t));
        const long row = (long)bl * T + t;
        out[row * ld_o + j] = h + (resid ? resid[row * ld_r + j] : 0.f);
      }
    }
  }
}

// The step waits need every workgroup resident at once: ask the occupancy API (per device and kernel, once).
struct LstmCapacity {
  std::atomic<int> cap[DynLdsAttr::kMaxDev];
  int get(const void* fn, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DynLdsAttr::kMaxDev) return -1;
    int v = cap[dev].load(std::memory_order_acquire);
    if (v == 0) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return -1;
      v = per_cu * cus > 0 ? per_cu * cus : -1;
      cap[dev].store(v, std::memory_order_release);
    }
    return v;
  }
};

template <int BG>
static hipError_t launch_lstm_persistent(const float* xproj, long ld_x, const float* w_hh, const float* b_hh, float* hbuf, unsigned* flags,
                                         const float* resid, long ld_r, float* out, long ld_o, int B, long T, hipStream_t s) {
  const size_t lds = ((size_t)8 * BG * LP_LDH + LP_UNITS * 32 * LP_RED) * sizeof(float);
  const int ngroups = (B + 8 * BG - 1) / (8 * BG);
  const int nwg = (512 / LP_UNITS) * ngroups;
  const void* fn = reinterpret_cast<const void*>(&lstm_persistent_kernel<BG>);
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(fn, (int)lds);
  if (e != hipSuccess) return e;
  // Require twice the grid: the API is known to be one block per CU optimistic in places (MI355X_MICROARCH.md), and a CU-masked or
  // partitioned device (32 CUs) must not take this path on a borderline count.  hipErrorNotReady = "use the per-step kernel".
  static LstmCapacity capacity;
  if (capacity.get(fn, lds) < 2 * nwg) return hipErrorNotReady;
  hipLaunchKernelGGL((lstm_persistent_kernel<BG>), dim3(512 / LP_UNITS, ngroups), dim3(256), lds, s, xproj, ld_x, T, w_hh, b_hh,
                     reinterpret_cast<unsigned long long*>(hbuf), resid, ld_r, out, ld_o, B, flags);
  return hipGetLastError();
}

static bool lstm_persistent_enabled() {              // NS2_LSTM_PERSISTENT=0: the per-step kernel (A/B, tests of both paths)
  static const bool on = [] { const char* e = getenv("NS2_LSTM_PERSISTENT"); return !(e && e[0] == '0'); }();
  return on;
}
constexpr long LP_STATE_FLOATS = 4L * LP_MAXB * 512 + 64;     // two exchange buffers of {h, tag} pairs + one abort-flag line per group
long lstm_state_floats(int B, int H) {
  const long step = 3L * B * H;
  return (H == 512 && B <= LP_MAXB) ? (step > LP_STATE_FLOATS ? step : LP_STATE_FLOATS) : step;
}
hipError_t launch_lstm_layer(const float* xproj, long ld_x, const float* w_hh, const float* b_hh, float* h_a, float* h_b,
                             float* c_state, long state_floats, const float* resid, long ld_r, float* out, long ld_o, int B,
                             long T, int H, hipStream_t s) {
  if (B <= 0 || T <= 0 || H <= 0 || H > 512 || (H & 3)) return hipErrorInvalidValue;
  if (H == 512 && B <= LP_MAXB && lstm_persistent_enabled()) {
    // the persistent kernel needs 2 x 32 x 512 {h, tag} pairs of exchange + the abort flags: the caller's scratch
    // (3 x B x H floats at least: h_a, h_b, c contiguous, include/ns2hip.h) is large enough when it was sized by
    // ns2_lstm_state_floats; a caller that handed over less takes the step kernel
    if (state_floats >= LP_STATE_FLOATS) {
      float* hbuf = h_a;
      unsigned* bar = reinterpret_cast<unsigned*>(h_a + 4L * LP_MAXB * 512);
      // tags of an earlier launch must not be mistaken for this one's: the whole exchange area starts as zeros (tag 0 is never written)
      hipError_t e = hipMemsetAsync(hbuf, 0, (size_t)LP_STATE_FLOATS * sizeof(float), s);
      if (e != hipSuccess) return e;
      // 8-row groups side by side (see the kernel); a single group of up to 32 rows (BG = 4) remains for devices too small
      // to hold 64 workgroups per group twice over
      e = launch_lstm_persistent<1>(xproj, ld_x, w_hh, b_hh, hbuf, bar, resid, ld_r, out, ld_o, B, T, s);
      if (e == hipErrorNotReady && B > 8) e = launch_lstm_persistent<4>(xproj, ld_x, w_hh, b_hh, hbuf, bar, resid, ld_r, out, ld_o, B, T, s);
      if (e != hipErrorNotReady) return e;         // NotReady: not enough resident workgroups on this device -> the per-step kernel below
    }
  }
  hipError_t e = hipMemsetAsync(h_a, 0, (size_t)B * H * sizeof(float), s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(c_state, 0, (size_t)B * H * sizeof(float), s);
  if (e != hipSuccess) return e;
  const dim3 grid((H + LSTM_UNITS - 1) / LSTM_UNITS, (B + LSTM_ROWS - 1) / LSTM_ROWS);
  for (long t = 0; t < T; ++t) {
    float* hp = (t & 1) ? h_b : h_a;
    float* hn = (t & 1) ? h_a : h_b;
    hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, s, xproj, ld_x, T, t, w_hh, b_hh, hp, hn, c_state, resid, ld_r, out, ld_o, B, H);
  }
  return hipGetLastError();
}

// scratch of launch_lstm2: h1 ring and h2 ring [2][LP_MAXB][512] pairs each, the projection rings [groups][64][LP_XDEPTH][256]
// pairs, one acknowledge word per workgroup pair, one flag line per group
static long lstm2_pairs() { return 4L * LP_MAXB * 512 + (LP_MAXB / 8) * 64L * LP_XDEPTH * 256; }
long lstm2_state_floats() { return 2 * lstm2_pairs() + (LP_MAXB / 8) * 64 + 64; }
// hipErrorNotReady: this device cannot hold all the workgroups at once (or the two-layer path is switched off, NS2_LSTM_FUSED=0)
// -- the caller runs the layers one after the other (launch_lstm_layer)
hipError_t launch_lstm2(const float* xproj, long ld_x, const float* w_hh1, const float* b_hh1, const float* w_ih2, const float* b_ih2,
                        const float* w_hh2, const float* b_hh2, float* state, long state_floats, const float* resid, long ld_r, float* out,
                        long ld_o, int B, long T, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("NS2_LSTM_FUSED"); return !(e && e[0] == '0'); }();
  if (B <= 0 || T <= 0 || state_floats < lstm2_state_floats()) return hipErrorInvalidValue;
  if (!on || !lstm_persistent_enabled() || B > LP_MAXB) return hipErrorNotReady;
  const size_t lds = ((size_t)8 * LP_LDH + LP_UNITS * 32 * LP_RED) * sizeof(float);
  const int ngroups = (B + 7) / 8;
  const int nwg = 2 * (512 / LP_UNITS) * ngroups;
  const void* fn = reinterpret_cast<const void*>(&lstm2_persistent_kernel);
  static DynLdsAttr attr;
  hipError_t e = attr.ensure(fn, (int)lds);
  if (e != hipSuccess) return e;
  // B = 32 fills the device exactly (two workgroups per CU, bounded by registers, where the occupancy API is exact): no factor
  // of two here; a launch that still cannot get resident ends through the abort flag and the caller is told (lstm_abort_read)
  static LstmCapacity capacity;
  if (capacity.get(fn, lds) < nwg) return hipErrorNotReady;
  e = hipMemsetAsync(state, 0, (size_t)lstm2_state_floats() * sizeof(float), s);       // no tag of an earlier launch survives
  if (e != hipSuccess) return e;
  unsigned long long* h1 = reinterpret_cast<unsigned long long*>(state);
  unsigned long long* h2 = h1 + 2L * LP_MAXB * 512;
  unsigned long long* xs = h2 + 2L * LP_MAXB * 512;
  unsigned* acks = reinterpret_cast<unsigned*>(h1 + lstm2_pairs());
  unsigned* flags = acks + (LP_MAXB / 8) * 64;
  hipLaunchKernelGGL(lstm2_persistent_kernel, dim3(512 / LP_UNITS, ngroups, 2), dim3(256), lds, s, xproj, ld_x, T, w_hh1, b_hh1, w_ih2, b_ih2,
                     w_hh2, b_hh2, h1, h2, xs, acks, resid, ld_r, out, ld_o, B, flags);
  return hipGetLastError();
}

hipError_t lstm_abort_inject(unsigned int n) {         // test hook: pretend n launches gave up (the host's fallback chain)
  return hipMemcpyToSymbol(HIP_SYMBOL(ns2_lstm_aborts), &n, sizeof n);
}
unsigned int lstm_abort_read(bool reset) {
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(ns2_lstm_aborts), sizeof v) != hipSuccess) return ~0u;
  if (reset && v) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(ns2_lstm_aborts), &z, sizeof z); }
  return v;
}

NS2_DEFINE_SATURATION_READER(elementwise)

}  // namespace ns2
