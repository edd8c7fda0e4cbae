// EnCodec residual-VQ encode on gfx950: for each of Q quantizers  idx = argmin_c |r - E_q[c]|^2 ; r -= E_q[idx]
// (HFENC:364-369 quantize, HFENC:424-438 encode, HFENC:440-447 decode -- the restatement of the un-vendored
// `encodec` core_vq that audiolm_pytorch.EncodecWrapper runs; reference call sites NS2:1445, NS2:1611).
//
// Index parity needs fp32-exact scores, so the distance contraction runs on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-exact fmaf chain, 157 TF peak) -- this is an fp32-MFMA-bound kernel, not an
// HBM-bound one (~3000 FLOP/B).  All Q stages are fused: a wave keeps its 32 residual rows in registers for
// the whole encode (64 VGPRs per lane), codebook tiles of 64 codes stream through a double-buffered LDS ring
// (an fp32 codebook is 512 KiB, larger than the 160 KiB LDS, so "LDS-resident" means tiled), and the running
// arg-max lives in registers: the product is computed as D[code][row] so that a lane owns ONE row and 16 codes
// per MFMA tile, visited in increasing code order (first-max tie-break like torch's).
// Near-ties (top-2 margin below the fp32 noise floor) are re-decided in fp64 on the two candidates.
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

constexpr int RV_D = 128;
constexpr int RV_ROWF = RV_D + 4;                 // padded LDS row (floats): 528 B = 33 x 16 B
constexpr int RV_TILE = 64;                       // codes per tile
constexpr int RV_STAGE_F = RV_TILE * RV_ROWF + RV_TILE;   // tile + its 64 half-norms

struct Cand { float v; int i; };
NS2_DEVINL bool better(const Cand& a, const Cand& b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

__global__ __launch_bounds__(256, 1) void rvq_encode_kernel(const RvqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long row = (long)blockIdx.x * 128 + wave * 32 + l31;
  const bool row_ok = row < a.M;

  // residual fragment: rf[s] = r[row][hi*64 + s]  (B operand slot (hi, s) of the fp32 MFMA)
  float rf[64];
#pragma unroll
  for (int s4 = 0; s4 < 16; ++s4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok) v = *reinterpret_cast<const float4*>(a.x + row * RV_D + hi * 64 + s4 * 4);
    rf[4 * s4] = v.x; rf[4 * s4 + 1] = v.y; rf[4 * s4 + 2] = v.z; rf[4 * s4 + 3] = v.w;
  }

  const int ntile = a.C / RV_TILE;
  // staging: 64 rows x 512 B = 2048 chunks of 16 B -> 8 per thread
  struct TileRegs { f32x4 v[8]; float nrm; };
  auto load_tile = [&](TileRegs& tr, int q, int ct) {
    const float* src = a.codebooks + ((long)q * a.C + (long)ct * RV_TILE) * RV_D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      tr.v[i] = *reinterpret_cast<const f32x4*>(src + (long)(c >> 5) * RV_D + (c & 31) * 4);
    }
    tr.nrm = (tid < RV_TILE) ? a.cb_norm[(long)q * a.C + ct * RV_TILE + tid] : 0.f;
  };
  auto store_tile = [&](const TileRegs& tr, int sidx) {
    float* base = lds + sidx * RV_STAGE_F;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<f32x4*>(base + (c >> 5) * RV_ROWF + (c & 31) * 4) = tr.v[i];
    }
    if (tid < RV_TILE) base[RV_TILE * RV_ROWF + tid] = tr.nrm;
  };

  for (int q = 0; q < a.Q; ++q) {
    Cand best = {-INFINITY, 0}, second = {-INFINITY, 0};
    TileRegs tr;
    load_tile(tr, q, 0);
    __syncthreads();                      // previous stage's readers are done with buffer 0
    store_tile(tr, 0);
    __syncthreads();
    for (int ct = 0; ct < ntile; ++ct) {
      const bool more = (ct + 1) < ntile;
      if (more) load_tile(tr, q, ct + 1);
      const float* tb = lds + (ct & 1) * RV_STAGE_F;
#pragma unroll
      for (int js = 0; js < 2; ++js) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* erow = tb + (js * 32 + l31) * RV_ROWF + hi * 64;     // A operand: E[code = l31][slot (hi, s)]
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
          const float4 e4 = *reinterpret_cast<const float4*>(erow + s4 * 4);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e4.x, rf[4 * s4 + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e4.y, rf[4 * s4 + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e4.z, rf[4 * s4 + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e4.w, rf[4 * s4 + 3], acc, 0, 0, 0);
        }
        // lane (row l31, half hi) register r is code  ct*64 + js*32 + (r&3) + 8*(r>>2) + 4*hi  (increasing in r)
        const float* nb = tb + RV_TILE * RV_ROWF + js * 32 + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = (r & 3) + 8 * (r >> 2);
          const float sc = acc[r] - nb[cl];            // r.e - |e|^2/2  (monotone in -distance)
          const int code = ct * RV_TILE + js * 32 + 4 * hi + cl;
          if (sc > best.v) { second = best; best.v = sc; best.i = code; }
          else if (sc > second.v) { second.v = sc; second.i = code; }
        }
      }
      if (more) store_tile(tr, (ct + 1) & 1);
      __syncthreads();
    }
    // merge the two half-waves (same row, interleaved code sets)
    Cand pb = {__shfl_xor(best.v, 32, 64), __shfl_xor(best.i, 32, 64)};
    Cand ps = {__shfl_xor(second.v, 32, 64), __shfl_xor(second.i, 32, 64)};
    Cand win = better(best, pb) ? best : pb;
    Cand lose = better(best, pb) ? pb : best;
    Cand s2 = better(second, ps) ? second : ps;
    Cand run = better(lose, s2) ? lose : s2;
    int idx = win.i;

    const float* cbq = a.codebooks + (long)q * a.C * RV_D;
    if (win.v - run.v < a.tie_eps * fmaxf(1.f, fabsf(win.v))) {
      // fp32 near-tie: decide the two candidates by their exact (fp64) squared distances; ties -> lower index
      const float* e0 = cbq + (long)win.i * RV_D + hi * 64;
      const float* e1 = cbq + (long)run.i * RV_D + hi * 64;
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int s = 0; s < 64; ++s) {
        const double t0 = (double)rf[s] - (double)e0[s], t1 = (double)rf[s] - (double)e1[s];
        d0 += t0 * t0;
        d1 += t1 * t1;
      }
      d0 += __shfl_xor(d0, 32, 64);
      d1 += __shfl_xor(d1, 32, 64);
      if (d1 < d0 || (d1 == d0 && run.i < win.i)) idx = run.i;
      if (a.near_tie_count && hi == 0 && row_ok) atomicAdd(a.near_tie_count, 1);
    }
    if (row_ok && hi == 0) a.codes[row * a.Q + q] = (int64_t)idx;

    // residual -= E[idx]   (HFENC:433-434); the summed embedding is produced by rvq_decode_kernel from the codes
    const float* esel = cbq + (long)idx * RV_D + hi * 64;
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      const float4 v = *reinterpret_cast<const float4*>(esel + s4 * 4);
      rf[4 * s4 + 0] -= v.x; rf[4 * s4 + 1] -= v.y; rf[4 * s4 + 2] -= v.z; rf[4 * s4 + 3] -= v.w;
    }
  }

  if (row_ok) {
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      if (a.residual)
        *reinterpret_cast<float4*>(a.residual + row * RV_D + hi * 64 + s4 * 4) =
            make_float4(rf[4 * s4], rf[4 * s4 + 1], rf[4 * s4 + 2], rf[4 * s4 + 3]);
    }
  }
}

hipError_t launch_rvq_encode(const RvqArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.Q <= 0 || a.D != RV_D || a.C <= 0 || (a.C % RV_TILE)) return hipErrorInvalidValue;
  const size_t lds = 2 * RV_STAGE_F * sizeof(float);
  static DynLdsAttr attr;
  {
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&rvq_encode_kernel), (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(rvq_encode_kernel, dim3((a.M + 127) / 128), dim3(256), lds, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !a.emb) return e;
  return launch_rvq_decode(a.codes, a.codebooks, a.emb, a.M, a.Q, a.C, a.D, s);
}

// cb_norm[q][c] = 0.5 * |E_q[c]|^2 : one wave per code
__global__ void rvq_prepare_kernel(const float* cb, float* out, long n_codes, int D) {
  const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_codes) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) { const float v = cb[c * D + k]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) out[c] = 0.5f * s;
}
hipError_t launch_rvq_prepare(const float* codebooks, float* cb_norm, int Q, int C, int D, hipStream_t s) {
  const long n = (long)Q * C;
  hipLaunchKernelGGL(rvq_prepare_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, codebooks, cb_norm, n, D);
  return hipGetLastError();
}

// emb[m] = 0.0 + E_0[c_0] + E_1[c_1] + ...   (HFENC:440-447)
__global__ void rvq_decode_kernel(const int64_t* codes, const float* cb, float* emb, long M, int Q, int C, int D) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * D) return;
  const long m = i / D;
  const int k = (int)(i - m * D);
  float acc = 0.f;
  for (int q = 0; q < Q; ++q) {
    const long c = codes[m * Q + q];
    acc += cb[((long)q * C + c) * D + k];
  }
  emb[i] = acc;
}
hipError_t launch_rvq_decode(const int64_t* codes, const float* codebooks, float* emb, int M, int Q, int C, int D,
                             hipStream_t s) {
  const long n = (long)M * D;
  hipLaunchKernelGGL(rvq_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, codes, codebooks, emb, (long)M,
                     Q, C, D);
  return hipGetLastError();
}

}  // namespace ns2
