// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (e5m2 x e5m2): raw operand bytes in, raw accumulators out, so that
// the (lane, byte) <-> k pairing and the per-lane E8M0 scale semantics the "mixed" precision GEMM assumes can be checked
// (and, if wrong, reverse-engineered from structured inputs) in one run.   a, b: [64 lanes][32 bytes]; sa, sb: [64] ints.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libscaleprobe.so mfma_scale_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_scale(const int* a, const int* b, const int* sa, const int* sb, float* d) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 1, 1, 0, sa[l], 0, sb[l]);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}
// throughput: n back-to-back dependent-free instructions per wave, 4 accumulators, one wave per SIMD x blocks
__global__ void k_rate(const int* a, const int* b, float* d, int iters) {
  const int l = threadIdx.x & 63;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c0, 1, 1, 0, 127, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c1, 1, 1, 0, 127, 0, 127);
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c2, 1, 1, 0, 127, 0, 127);
    c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c3, 1, 1, 0, 127, 0, 127);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 123.456f) d[0] = s;
}
extern "C" int scale_probe_run(const int* a, const int* b, const int* sa, const int* sb, float* d, void* stream) {
  hipLaunchKernelGGL(k_scale, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, sa, sb, d);
  return (int)hipGetLastError();
}
extern "C" int scale_probe_rate(const int* a, const int* b, float* d, int iters, int blocks, void* stream) {
  hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, d, iters);
  return (int)hipGetLastError();
}
