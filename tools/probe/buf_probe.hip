// probe: bounds-check semantics of buffer_load_dwordx4 ... offen lds (raw buffer, stride 0) on gfx950:
// does the SGPR offset count towards num_records, and do wrapped ("negative") VGPR offsets return zeros?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_t;
__global__ void k(const float* p, int nbytes, int soff, int vbias, float* out) {
  __shared__ float sm[256];
  sm[threadIdx.x] = -1.f; sm[threadIdx.x + 64] = -1.f; sm[threadIdx.x + 128] = -1.f; sm[threadIdx.x + 192] = -1.f;
  __syncthreads();
  auto r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_t*)sm, 16, threadIdx.x * 16 + vbias, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = sm[threadIdx.x * 4 + i];
}
int main() {
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1000.f + i;
  float *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 1024); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  struct { int nbytes, soff, vbias; const char* what; } cs[] = {
    {1024, 0, 0, "num_records 1024, no offsets: lanes 0-63 read bytes 0..1023"},
    {1024, 512, 0, "soffset 512: total offset >= 1024 for lanes >= 32"},
    {1024, 0, -256, "voffset biased by -256: lanes 0-15 wrap negative"},
    {512, 0, 0, "num_records 512: lanes >= 32 out of range"},
    {1024, 1024, -1024, "soffset 1024, voffset -1024 (sum in range, voffset wrapped)"},
  };
  for (auto& c : cs) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.nbytes, c.soff, c.vbias, o);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("%s\n   lane0=%g lane15=%g lane16=%g lane31=%g lane32=%g lane47=%g lane48=%g lane63=%g\n", c.what, r[0], r[60], r[64], r[124], r[128], r[188], r[192], r[252]);
  }
  return 0;
}
