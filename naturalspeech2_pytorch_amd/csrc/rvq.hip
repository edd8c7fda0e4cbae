// EnCodec residual-VQ encode on gfx950: for each of Q quantizers  idx = argmin_c |r - E_q[c]|^2 ; r -= E_q[idx]
// (HFENC:364-369 quantize, HFENC:424-438 encode, HFENC:440-447 decode -- the restatement of the un-vendored
// `encodec` core_vq that audiolm_pytorch.EncodecWrapper runs; reference call sites NS2:1445, NS2:1611).
//
// Index parity needs fp32-exact scores, so the distance contraction runs on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 products and sums, 157 TF peak) -- this is an fp32-MFMA-bound kernel, not an
// HBM-bound one (~3000 FLOP/B).  All Q stages are fused: a wave keeps its 16 residual rows in registers for
// the whole encode (32 VGPRs per lane), codebook tiles of 64 codes stream through a double-buffered LDS ring
// (an fp32 codebook is 512 KiB, larger than the 160 KiB LDS, so "LDS-resident" means tiled), and the running
// arg-max lives in registers: the product is computed as D[code][row] so that a lane owns ONE row and 4 codes
// per MFMA tile, visited in increasing code order (first-max tie-break like torch's).
// Near-ties (top-2 margin below the fp32 noise floor) are re-decided in fp64 on the two candidates.
#include "ns2_common.h"
#include "ns2_kernels.h"

namespace ns2 {

constexpr int RV_D = 128;
constexpr int RV_ROWF = RV_D + 4;                 // padded LDS row (floats): 528 B = 33 x 16 B
constexpr int RV_TILE = 64;                       // codes per tile
constexpr int RV_STAGE_F = RV_TILE * RV_ROWF + RV_TILE;   // tile + its 64 half-norms
constexpr int RV_ROWS = 128;                      // latent rows per workgroup: 8 waves x 16 rows

struct Cand { float v; int i; };
NS2_DEVINL bool better(const Cand& a, const Cand& b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

// Round-2 structure: v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency) on 16 latent rows per wave and
// EIGHT waves per workgroup of 128 rows, i.e. two waves per SIMD at one workgroup per CU: while one wave of a SIMD runs its
// arg-max VALU / LDS phase the other keeps the fp32 matrix pipe busy (the 32x32x2 version ran ONE wave per SIMD and idled the
// pipe during every VALU phase: 51 % of the 157 TF peak).  Product D[code][row] as before: a lane (l15 = lane & 15,
// g = lane >> 4) owns row l15 and, per 16-code group, the 4 codes 4 g .. 4 g + 3; two code groups are accumulated as
// independent chains to cover the dependent latency.  K assignment: lane group g multiplies k = 32 g .. 32 g + 31 (any
// assignment is a valid contraction order; this one makes both operands 16-B vector loads).
__global__ __launch_bounds__(512, 2) void rvq_encode_kernel(const RvqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const long row = (long)blockIdx.x * RV_ROWS + wave * 16 + l15;
  const bool row_ok = row < a.M;

  // residual fragment: rf[s] = r[row][32 g + s]  (B operand: k slot (g, s))
  float rf[32];
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok) v = *reinterpret_cast<const float4*>(a.x + row * RV_D + g * 32 + s4 * 4);
    rf[4 * s4] = v.x; rf[4 * s4 + 1] = v.y; rf[4 * s4 + 2] = v.z; rf[4 * s4 + 3] = v.w;
  }

  const int ntile = a.C / RV_TILE;
  // staging: 64 rows x 512 B = 2048 chunks of 16 B -> 4 per thread
  struct TileRegs { f32x4 v[4]; float nrm; };
  auto load_tile = [&](TileRegs& tr, int q, int ct) {
    const float* src = a.codebooks + ((long)q * a.C + (long)ct * RV_TILE) * RV_D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 512 * i;
      tr.v[i] = *reinterpret_cast<const f32x4*>(src + (long)(c >> 5) * RV_D + (c & 31) * 4);
    }
    tr.nrm = (tid < RV_TILE) ? a.cb_norm[(long)q * a.C + ct * RV_TILE + tid] : 0.f;
  };
  auto store_tile = [&](const TileRegs& tr, int sidx) {
    float* base = lds + sidx * RV_STAGE_F;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 512 * i;
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
      for (int cp = 0; cp < 2; ++cp) {                                       // two pairs of 16-code groups per tile
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* e0 = tb + ((2 * cp) * 16 + l15) * RV_ROWF + g * 32;     // A operand: E[code = l15][k slot (g, s)]
        const float* e1 = e0 + 16 * RV_ROWF;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const float4 x0 = *reinterpret_cast<const float4*>(e0 + s4 * 4);
          const float4 x1 = *reinterpret_cast<const float4*>(e1 + s4 * 4);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, rf[4 * s4 + 0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, rf[4 * s4 + 0], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, rf[4 * s4 + 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, rf[4 * s4 + 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, rf[4 * s4 + 2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, rf[4 * s4 + 2], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, rf[4 * s4 + 3], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, rf[4 * s4 + 3], acc1, 0, 0, 0);
        }
        // lane (row l15, group g) register i of code group cg is code  ct*64 + cg*16 + 4 g + i  (increasing in cg, i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int cg = 2 * cp + h;
          const float4 n4 = *reinterpret_cast<const float4*>(tb + RV_TILE * RV_ROWF + cg * 16 + 4 * g);
          const float nb[4] = {n4.x, n4.y, n4.z, n4.w};
          const f32x4 acc = h ? acc1 : acc0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sc = acc[i] - nb[i];             // r.e - |e|^2/2  (monotone in -distance)
            const int code = ct * RV_TILE + cg * 16 + 4 * g + i;
            if (sc > best.v) { second = best; best.v = sc; best.i = code; }
            else if (sc > second.v) { second.v = sc; second.i = code; }
          }
        }
      }
      if (more) store_tile(tr, (ct + 1) & 1);
      __syncthreads();
    }
    // merge the four lane groups holding the same row (disjoint code sets): two butterfly steps
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const Cand pb = {__shfl_xor(best.v, off, 64), __shfl_xor(best.i, off, 64)};
      const Cand ps = {__shfl_xor(second.v, off, 64), __shfl_xor(second.i, off, 64)};
      const bool mine = better(best, pb);
      const Cand win = mine ? best : pb;
      const Cand lose = mine ? pb : best;
      const Cand s2 = better(second, ps) ? second : ps;
      best = win;
      second = better(lose, s2) ? lose : s2;
    }
    int idx = best.i;

    const float* cbq = a.codebooks + (long)q * a.C * RV_D;
    if (best.v - second.v < a.tie_eps * fmaxf(1.f, fabsf(best.v))) {
      // fp32 near-tie: decide the two candidates by their exact (fp64) squared distances; ties -> lower index
      const float* e0 = cbq + (long)best.i * RV_D + g * 32;
      const float* e1 = cbq + (long)second.i * RV_D + g * 32;
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const double t0 = (double)rf[s] - (double)e0[s], t1 = (double)rf[s] - (double)e1[s];
        d0 += t0 * t0;
        d1 += t1 * t1;
      }
      d0 += __shfl_xor(d0, 16, 64); d1 += __shfl_xor(d1, 16, 64);
      d0 += __shfl_xor(d0, 32, 64); d1 += __shfl_xor(d1, 32, 64);
      if (d1 < d0 || (d1 == d0 && second.i < best.i)) idx = second.i;
      if (a.near_tie_count && g == 0 && row_ok) atomicAdd(a.near_tie_count, 1);
    }
    if (row_ok && g == 0) a.codes[row * a.Q + q] = (int64_t)idx;

    // residual -= E[idx]   (HFENC:433-434); the summed embedding is produced by rvq_decode_kernel from the codes
    const float* esel = cbq + (long)idx * RV_D + g * 32;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 v = *reinterpret_cast<const float4*>(esel + s4 * 4);
      rf[4 * s4 + 0] -= v.x; rf[4 * s4 + 1] -= v.y; rf[4 * s4 + 2] -= v.z; rf[4 * s4 + 3] -= v.w;
    }
  }

  if (row_ok && a.residual) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
      *reinterpret_cast<float4*>(a.residual + row * RV_D + g * 32 + s4 * 4) =
          make_float4(rf[4 * s4], rf[4 * s4 + 1], rf[4 * s4 + 2], rf[4 * s4 + 3]);
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
  hipLaunchKernelGGL(rvq_encode_kernel, dim3((a.M + RV_ROWS - 1) / RV_ROWS), dim3(512), lds, s, a);
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
