// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libns2hip.
//
// Activation / weight storage format ("split planes"): every GEMM operand is held as two bf16 planes,
// hi = bf16_rne(x) and lo = bf16_rne(x - float(hi)).  A product of two split operands is evaluated as
// hi*hi + hi*lo + lo*hi on the bf16 MFMA pipe with fp32 accumulation (3 MFMAs, relative error ~2^-16 per
// product => fp32-class results); the "fast" mode uses the hi plane only (1 MFMA).
//
// Memory layout of a split-plane matrix [rows, ld] (ld = LOGICAL columns, a multiple of 32):
//   * with a lo plane ("interleaved"): ONE bf16 buffer [rows, 2*ld]; every 32 logical columns occupy one 128-byte
//     line [hi(32) | lo(32)], i.e. element (r, c) has hi at r*2*ld + pcol(c) and lo 32 elements further
//     (lo pointer == hi pointer + 32, checked by the launchers).  A K tile of 32 columns of both planes is therefore
//     ONE full cache line per row: the GEMM's LDS-DMA touches 8 full lines per 1 KiB wave-instruction instead of
//     16 half lines of two planar buffers -- the vector-memory front end retires requests per line, and the planar
//     layout measured 2.5x the DMA time (GEMM measurements);
//   * hi only (lo == nullptr, "fast" precision): dense [rows, ld].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                           // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;          // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;         // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NS2_DEVINL __device__ __forceinline__

// fp32 -> bf16 (round to nearest even); NaN kept quiet.
NS2_DEVINL bf16_t f2bf(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
NS2_DEVINL float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// split x into (hi, lo) bf16
NS2_DEVINL void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// physical column / row stride of a split-plane matrix (see the layout note above); il = "has a lo plane"
NS2_DEVINL int pcol(int c, bool il) { return il ? (((c & ~31) << 1) | (c & 31)) : c; }
NS2_DEVINL long pld(int ld, bool il) { return il ? 2L * ld : (long)ld; }

NS2_DEVINL uint32_t pack2(bf16_t a, bf16_t b) { return (uint32_t)a | ((uint32_t)b << 16); }

// gfx950 has a hardware fp32x2 -> packed bf16 conversion (v_cvt_pk_bf16_f32, RNE): 5 VALU instructions split two
// values into packed (hi, lo) words, against ~25 for the integer formulation above (measured: the GEMM epilogues and
// the attention P conversion were VALU-bound on it).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
NS2_DEVINL uint32_t cvt2(float a, float b) {          // {bf16(a) | bf16(b) << 16}
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
NS2_DEVINL void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  f32x2_t v = {a, b};
  bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
  f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));
}

// ---- IEEE half operands ("half" precision: ONE fp16 product per contraction, fp32 accumulate).  fp16 carries 11
// significand bits against bf16's 8, so a single product lands at ~5e-4 end to end where bf16 needs the 3-product split
// (tools/precision_study.py); the price is range: values are clamped to +-65504 on conversion instead of becoming inf.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
NS2_DEVINL uint32_t cvt2h(float a, float b) {         // {half(a) | half(b) << 16}, round to nearest even, saturating
  f32x2_t v = {fminf(fmaxf(a, -65504.f), 65504.f), fminf(fmaxf(b, -65504.f), 65504.f)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
NS2_DEVINL float h2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// element-format-aware pair conversion: f16 planes have no lo part
NS2_DEVINL void split2f(float a, float b, uint32_t& hi, uint32_t& lo, int f16) {
  if (f16) { hi = cvt2h(a, b); lo = 0u; }
  else split2(a, b, hi, lo);
}
// one 32x32x16 MFMA on 16-bit operands of either format
template <bool F16>
NS2_DEVINL f32x16 mma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

NS2_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

NS2_DEVINL float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
NS2_DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
NS2_DEVINL float siluf(float x) { return x / (1.0f + expf(-x)); }

// XCD-aware bijective remap of a linear workgroup id (guide T1): consecutive ids land on one XCD's L2.
NS2_DEVINL int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
