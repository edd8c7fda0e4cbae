// Probe of gfx950's LDS transpose reads (ds_read_b64_tr_b16 / ds_read_b64_tr_b8): which LDS bytes does lane l receive, as a
// function of the per-lane addresses?  LDS holds its own offsets (16-bit words: word index; bytes: byte index split over two runs),
// every lane supplies `addr[lane]`, the result is printed per lane.  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned char* in, const int* addr, unsigned long long* out16, unsigned long long* out8) {
  __shared__ __attribute__((aligned(16))) unsigned char s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = in[i];
  __syncthreads();
  const unsigned a = (unsigned)(size_t)s + addr[threadIdx.x];
  unsigned long long r16, r8;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r16) : "v"(a) : "memory");
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r8) : "v"(a) : "memory");
  out16[threadIdx.x] = r16;
  out8[threadIdx.x] = r8;
}
int main() {
  std::vector<unsigned char> h(8192);
  unsigned char* din; int* daddr; unsigned long long *d16, *d8;
  hipMalloc(&din, 8192); hipMalloc(&daddr, 256); hipMalloc(&d16, 512); hipMalloc(&d8, 512);
  // address patterns: 0 = lane * 8 (contiguous); 1 = rows of 64 B: (lane % 16 / 4) * 64 + (lane % 4) * 8 + (lane / 16) * 1024
  //                   2 = tr_b8 guess: rows of 32 B: (lane % 16 / 2) * 32 + (lane % 2) * 8 + (lane / 16) * 1024
  for (int pat = 0; pat < 3; ++pat) {
    int addr[64];
    for (int l = 0; l < 64; ++l)
      addr[l] = pat == 0 ? l * 8 : pat == 1 ? ((l % 16) / 4) * 64 + (l % 4) * 8 + (l / 16) * 1024 : ((l % 16) / 2) * 32 + (l % 2) * 8 + (l / 16) * 1024;
    hipMemcpy(daddr, addr, 256, hipMemcpyHostToDevice);
    unsigned long long r16[64], r8lo[64], r8hi[64];
    for (int i = 0; i < 4096; ++i) { h[2 * i] = i & 255; h[2 * i + 1] = i >> 8; }           // 16-bit word i = i
    hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, daddr, d16, d8);
    hipMemcpy(r16, d16, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8192; ++i) h[i] = i & 255;
    hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, daddr, d16, d8);
    hipMemcpy(r8lo, d8, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8192; ++i) h[i] = i >> 8;
    hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, daddr, d16, d8);
    hipMemcpy(r8hi, d8, 512, hipMemcpyDeviceToHost);
    printf("== pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr %5d | tr_b16 word offsets:", l, addr[l]);
      for (int j = 0; j < 4; ++j) printf(" %5d", (int)((r16[l] >> (16 * j)) & 0xffff) * 2);
      printf(" | tr_b8 byte offsets:");
      for (int j = 0; j < 8; ++j) printf(" %5d", (int)(((r8hi[l] >> (8 * j)) & 0xff) << 8 | ((r8lo[l] >> (8 * j)) & 0xff)));
      printf("\n");
    }
  }
  return 0;
}
