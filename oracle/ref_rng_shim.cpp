// C-ABI wrapper around the REFERENCE's own rng.h, compiled from /root/reference/RayTracer/rng.h where
// it lies (see oracle/Makefile, target `ref`).  rng.h needs only std headers, so no stand-in for any
// absent header is involved.  The resulting oracle/_ref/librefrng.so pins oracle.cpp's restatement of
// PCG32 / Halton (tests/test_oracle_rng.py).  This file contains no reference source.
#include <cstdint>

#include "rng.h"

extern "C" {

void ref_pcg32_seq(uint64_t sequence, uint64_t seed, int use_seed, int n, uint32_t* out_u32, float* out_f32) {
  RNG r(sequence);
  if (use_seed) r = RNG(sequence, seed);
  for (int i = 0; i < n; i++) {
    if (out_u32) out_u32[i] = r.RandomInt();
    else if (out_f32) out_f32[i] = r.RandomFloat();
  }
}

float ref_halton(int index, int base) { return Halton(index, base); }

void ref_halton_seq128(int base, float* out) {
  HaltonSeq<128> s(base);
  for (int i = 0; i < 128; i++) out[i] = s[i];
}

}
