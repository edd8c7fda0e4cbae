// Element formats of a split-plane matrix (layouts: ns2_common.h).
#pragma once
enum PlaneFmt : int {
  FMT_BF16 = 0,   // bf16 hi plane, optional lo plane (interleaved [hi32|lo32] lines when present)
  FMT_F16 = 1,    // one dense IEEE-half plane
  FMT_H8 = 2,     // interleaved lines [half x 32 | e5m2(x) x 32 | e5m2((x - half(x)) * 2^12) x 32]  ("mixed" precision)
};
