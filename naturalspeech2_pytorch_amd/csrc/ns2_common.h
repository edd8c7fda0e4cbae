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

#include "ns2_fmt.h"

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

// ---- Range guard of the IEEE-half based precisions.  Half (and e5m2) stop at 65504 (57344): a value beyond is clamped, which
// keeps everything finite but is WRONG, silently (measured: weights x8 on a random-init d128/L6 model -> relative error 1.0
// at precisions 2 and 4, 1e-5 at precision 3 whose bf16 planes have the fp32 exponent range).  Every conversion that clamps
// (or meets a NaN) bumps a per-device counter; the host reads it after a sampling run (ns2_saturation_count) and fails
// loudly.  One static counter per translation unit (no relocatable device code); capi.cpp sums them.
static __device__ unsigned int ns2_sat_counter;
NS2_DEVINL void note_out_of_range(float a, float b, float limit) {
  if (!(fabsf(a) <= limit) || !(fabsf(b) <= limit)) atomicAdd(&ns2_sat_counter, 1u);
}
#define NS2_DEFINE_SATURATION_READER(tu)                                                              \
  unsigned int saturation_read_##tu(bool reset) {                                                     \
    unsigned int v = 0;                                                                               \
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(ns2_sat_counter), sizeof v) != hipSuccess) return ~0u;     \
    if (reset && v) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(ns2_sat_counter), &z, sizeof z); } \
    return v;                                                                                         \
  }                                                                                                   \
  hipError_t saturation_peek_##tu(unsigned int* dst, hipStream_t s) {                                 \
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(ns2_sat_counter), sizeof(unsigned int), 0, hipMemcpyDeviceToHost, s); \
  }

// ---- IEEE half operands ("half" precision: ONE fp16 product per contraction, fp32 accumulate).  fp16 carries 11
// significand bits against bf16's 8, so a single product lands at ~5e-4 end to end where bf16 needs the 3-product split
// (tools/precision_study.py); the price is range: values are clamped to +-65504 on conversion instead of becoming inf.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
NS2_DEVINL uint32_t cvt2h(float a, float b) {         // {half(a) | half(b) << 16}, round to nearest even, saturating
  note_out_of_range(a, b, 65504.f);
  f32x2_t v = {fminf(fmaxf(a, -65504.f), 65504.f), fminf(fmaxf(b, -65504.f), 65504.f)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// the same for values known to lie inside the half range (softmax probabilities): no clamp, no range guard
NS2_DEVINL uint32_t cvt2h_inrange(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
NS2_DEVINL float h2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// element-format-aware pair conversion: f16 planes have no lo part
NS2_DEVINL void split2f(float a, float b, uint32_t& hi, uint32_t& lo, int f16) {
  if (f16) { hi = cvt2h(a, b); lo = 0u; }
  else split2(a, b, hi, lo);
}
// ---- "mixed" precision (FMT_H8): fp16 main product + BOTH first-order correction terms in one fp8 MFMA.
// x = h + l with h = half(x), l = x - h (|l| <= 2^-11 |x|).  a.w = a_h.w_h + (a_h.w_l + a_l.w_h) + O(2^-22): the bracket only has
// to be right to a few bits, so it runs on the gfx950 block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the
// 16-bit rate) with e5m2 operands: h8 = e5m2(x) (fp16's exponent range, so no data-dependent scale is needed) and
// l8 = e5m2(l * 2^12); the constant 2^-12 is the instruction's E8M0 block scale.  One K = 64 fp8 instruction covers a 32-deep
// k block: lanes 0-31 multiply a_h8 . w_l8, lanes 32-63 a_l8 . w_h8.  Measured on the CPU emulation of the whole Model
// (tools/precision_study.py): 4e-5 from the fp32 reference, against 5.5e-4 for the single fp16 product and 8e-6 for bf16 x3.
// Storage: the interleaved 128-B line of 32 logical columns is [half x 32 | h8 x 32 | l8 x 32]; pointer convention as for
// the bf16 split planes (the "lo" pointer is hi + 32 elements = the byte half of the line).
constexpr float H8_LO_SCALE = 4096.0f;            // 2^12
constexpr int H8_E8M0_LO = 127 - 12;              // E8M0 code of 2^-12
constexpr int H8_E8M0_ONE = 127;
constexpr float H8_MAX = 57344.0f;                // largest finite e5m2 (< 65504: one clamp serves the half and the e5m2 parts)
typedef __attribute__((ext_vector_type(8))) int i32x8;

// (a, b) -> packed halves, packed e5m2(a, b) and packed e5m2 of the scaled remainders (each in the low 16 bits)
NS2_DEVINL void cvt2_h8(float a, float b, uint32_t& h16, uint32_t& h8, uint32_t& l8) {
  note_out_of_range(a, b, H8_MAX);
  a = fminf(fmaxf(a, -H8_MAX), H8_MAX);
  b = fminf(fmaxf(b, -H8_MAX), H8_MAX);
  f32x2_t v = {a, b};
  f16x2_t h = __builtin_convertvector(v, f16x2_t);
  f32x2_t r = (v - __builtin_convertvector(h, f32x2_t)) * H8_LO_SCALE;
  h16 = __builtin_bit_cast(uint32_t, h);
  h8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
  l8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(r.x, -H8_MAX), H8_MAX), fminf(fmaxf(r.y, -H8_MAX), H8_MAX), 0, false) & 0xffffu;
}
NS2_DEVINL float bf8_to_f(uint32_t byte) {        // e5m2 -> fp32 (e5m2 is the top byte of an IEEE half)
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(byte << 8));
}

// Store two adjacent logical columns (c0 even) / four (c0 % 4 == 0) of ONE row of a split-plane matrix in format `fmt`.
// `row` = address of the row's first element (physical), il = the row is made of interleaved 128-B lines (bf16 with a lo
// plane, or FMT_H8).
NS2_DEVINL void store_cols2(bf16_t* row, int c0, float v0, float v1, int fmt, bool il) {
  if (fmt == FMT_H8) {
    uint32_t h16, h8, l8;
    cvt2_h8(v0, v1, h16, h8, l8);
    bf16_t* line = row + ((c0 & ~31) << 1);
    *reinterpret_cast<uint32_t*>(line + (c0 & 31)) = h16;
    unsigned char* bytes = reinterpret_cast<unsigned char*>(line) + 64 + (c0 & 31);
    *reinterpret_cast<uint16_t*>(bytes) = (uint16_t)h8;
    *reinterpret_cast<uint16_t*>(bytes + 32) = (uint16_t)l8;
  } else if (fmt == FMT_F16) {
    *reinterpret_cast<uint32_t*>(row + c0) = cvt2h(v0, v1);
  } else {
    uint32_t ph, pl;
    split2(v0, v1, ph, pl);
    bf16_t* p = row + (il ? (((c0 & ~31) << 1) | (c0 & 31)) : c0);
    *reinterpret_cast<uint32_t*>(p) = ph;
    if (il) *reinterpret_cast<uint32_t*>(p + 32) = pl;
  }
}
NS2_DEVINL void store_cols4(bf16_t* row, int c0, float v0, float v1, float v2, float v3, int fmt, bool il) {
  if (fmt == FMT_H8) {
    uint32_t ha, hb, h8a, h8b, l8a, l8b;
    cvt2_h8(v0, v1, ha, h8a, l8a);
    cvt2_h8(v2, v3, hb, h8b, l8b);
    bf16_t* line = row + ((c0 & ~31) << 1);
    *reinterpret_cast<uint2*>(line + (c0 & 31)) = make_uint2(ha, hb);
    unsigned char* bytes = reinterpret_cast<unsigned char*>(line) + 64 + (c0 & 31);
    *reinterpret_cast<uint32_t*>(bytes) = h8a | (h8b << 16);
    *reinterpret_cast<uint32_t*>(bytes + 32) = l8a | (l8b << 16);
  } else if (fmt == FMT_F16) {
    *reinterpret_cast<uint2*>(row + c0) = make_uint2(cvt2h(v0, v1), cvt2h(v2, v3));
  } else {
    uint32_t h01, l01, h23, l23;
    split2(v0, v1, h01, l01);
    split2(v2, v3, h23, l23);
    bf16_t* p = row + (il ? (((c0 & ~31) << 1) | (c0 & 31)) : c0);
    *reinterpret_cast<uint2*>(p) = make_uint2(h01, h23);
    if (il) *reinterpret_cast<uint2*>(p + 32) = make_uint2(l01, l23);
  }
}
// does a matrix of this format / lo pointer use interleaved 128-B lines?
NS2_DEVINL bool fmt_il(int fmt, const void* lo) { return fmt == FMT_H8 || lo != nullptr; }

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
// erfc(z), relative error < 1.2e-7 for every z (Chebyshev fit of erfc(z) * exp(z^2) in t = 1 / (1 + |z| / 2), Numerical Recipes
// "erfcc"); branch-free: one v_rcp_f32, nine fmas, one v_exp_f32
NS2_DEVINL float erfc_fast(float z) {
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, az, 1.0f));
  float p = 0.17087277f;
  p = fmaf(p, t, -0.82215223f);
  p = fmaf(p, t, 1.48851587f);
  p = fmaf(p, t, -1.13520398f);
  p = fmaf(p, t, 0.27886807f);
  p = fmaf(p, t, -0.18628806f);
  p = fmaf(p, t, 0.09678418f);
  p = fmaf(p, t, 0.37409196f);
  p = fmaf(p, t, 1.00002368f);
  p = fmaf(p, t, -1.26551223f);
  const float e = __builtin_amdgcn_exp2f((p - az * az) * 1.4426950408889634f);
  const float r = t * e;
  return z >= 0.f ? r : 2.0f - r;
}
// gelu(x) = x * Phi(x) = 0.5 * x * erfc(-x / sqrt(2))   (NS2:1006-1007, F.gelu's erf form).  ONE definition for every GEGLU epilogue
// (both GEMM kernels, fast and generic paths): a row's result must not depend on which path its tile took (batch independence
// is asserted to 1e-6).  The device library's erff is two divergent branches (~60 instructions in a mixed wave); this is 16.
NS2_DEVINL float gelu_erf(float x) { return 0.5f * x * erfc_fast(-0.70710678118654752440f * x); }

NS2_DEVINL float siluf(float x) { return x / (1.0f + expf(-x)); }
NS2_DEVINL float eluf(float x) { return x > 0.f ? x : expm1f(x); }
// epilogue activations of the linear entry points (include/ns2hip.h): 0 none, 1 SiLU, 2 ELU
NS2_DEVINL float apply_act(float v, int act) { return act == 1 ? siluf(v) : (act == 2 ? eluf(v) : v); }

// XCD-aware bijective remap of a linear workgroup id (guide T1): consecutive ids land on one XCD's L2.
NS2_DEVINL int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
