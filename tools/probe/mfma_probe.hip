// MFMA layout probe for gfx950: verifies the lane->element maps this repo's kernels assume.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libprobe.so mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A: [32][16] bf16 row-major, B: [16][32] bf16 row-major, D: [32][32] f32 row-major, raw: [64][16]
__global__ void probe_32x32x16(const uint16_t* A, const uint16_t* B, float* D, float* raw) {
  int l = threadIdx.x, hi = l >> 5, r32 = l & 31;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (short)A[r32 * 16 + 8 * hi + t];        // A[i=l&31][k=8*hi+t]
    b[t] = (short)B[(8 * hi + t) * 32 + r32];      // B[k=8*hi+t][j=l&31]
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi, col = r32;
    D[row * 32 + col] = c[r];
    raw[l * 16 + r] = c[r];
  }
}

// A: [16][32], B: [32][16], D: [16][16]
__global__ void probe_16x16x32(const uint16_t* A, const uint16_t* B, float* D, float* raw) {
  int l = threadIdx.x, g = l >> 4, r16 = l & 15;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (short)A[r16 * 32 + 8 * g + t];
    b[t] = (short)B[(8 * g + t) * 16 + r16];
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = g * 4 + r, col = r16;
    D[row * 16 + col] = c[r];
    raw[l * 4 + r] = c[r];
  }
}

// f32: A [32][2], B [2][32], D [32][32]
__global__ void probe_32x32x2f32(const float* A, const float* B, float* D, float* raw) {
  int l = threadIdx.x, hi = l >> 5, r32 = l & 31;
  float a = A[r32 * 2 + hi], b = B[hi * 32 + r32];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi, col = r32;
    D[row * 32 + col] = c[r];
    raw[l * 16 + r] = c[r];
  }
}

// f32: A [16][4], B [4][16], D [16][16]
__global__ void probe_16x16x4f32(const float* A, const float* B, float* D, float* raw) {
  int l = threadIdx.x, g = l >> 4, r16 = l & 15;
  float a = A[r16 * 4 + g], b = B[g * 16 + r16];
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = g * 4 + r, col = r16;
    D[row * 16 + col] = c[r];
    raw[l * 4 + r] = c[r];
  }
}

extern "C" int probe_run(int which, const void* A, const void* B, void* D, void* raw, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (which) {
    case 0: probe_32x32x16<<<1, 64, 0, s>>>((const uint16_t*)A, (const uint16_t*)B, (float*)D, (float*)raw); break;
    case 1: probe_16x16x32<<<1, 64, 0, s>>>((const uint16_t*)A, (const uint16_t*)B, (float*)D, (float*)raw); break;
    case 2: probe_32x32x2f32<<<1, 64, 0, s>>>((const float*)A, (const float*)B, (float*)D, (float*)raw); break;
    case 3: probe_16x16x4f32<<<1, 64, 0, s>>>((const float*)A, (const float*)B, (float*)D, (float*)raw); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
