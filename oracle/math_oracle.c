/* math_oracle.c -- TEST INFRASTRUCTURE (oracle).  Host libm expf (the function the reference calls at
 * core/ctree/cnode.cpp:87) swept over blocks of 2^20 float bit patterns; the HIP library computes the same
 * checksum with its own device expf (hanabizero_amd/csrc/hz_common.h) and tests/test_hip_math.py compares. */
#include <math.h>
#include <stdint.h>
#include <string.h>

void hzo_expf_checksum_range(uint64_t* out, int block_begin, int block_end) {
  for (int b = block_begin; b < block_end; ++b) {
    uint32_t base = (uint32_t)b << 20;
    uint64_t acc = 0;
    for (uint32_t i = 0; i < (1u << 20); ++i) {
      uint32_t u = base + i, rb;
      float x, r;
      memcpy(&x, &u, 4);
      r = expf(x);
      memcpy(&rb, &r, 4);
      if (r != r) rb = 0x7fc00000u;
      acc += (uint64_t)rb * (2ull * (uint64_t)i + 1ull);
    }
    out[b] = acc;
  }
}

void hzo_expf_array(const float* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n; ++i) y[i] = expf(x[i]);
}
