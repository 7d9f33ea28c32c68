// hz_selfplay.hip -- batch form of the per-env Python glue of core/selfplay_worker.py:286-347 (gfx950).
#include <math.h>

#include "hz_common.h"
#include "hz_selfplay.h"

// one lane per env; rows are short (A <= 64) and the kernel is launch-bound, not bandwidth-bound
__global__ __launch_bounds__(256) void k_select_action(int N, int A, int32_t* __restrict__ counts,
                                                       const uint8_t* __restrict__ legal,
                                                       const double* __restrict__ uniform, float temperature,
                                                       int deterministic, int32_t* __restrict__ out_action,
                                                       double* __restrict__ out_entropy) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  int32_t* c = counts + (size_t)env * A;
  const uint8_t* lg = legal + (size_t)env * A;
  // utils.py:282-284: zero the counts of illegal actions
  double total = 0.0;
  int best = 0, best_count = INT32_MIN;
  const bool unit_t = (temperature == 1.0f);
  const double inv_t = 1.0 / (double)temperature;
  for (int a = 0; a < A; ++a) {
    int v = c[a];
    if (lg[a] == 0 && v >= 1) {
      v = 0;
      c[a] = 0;
    }
    if (v > best_count) {  // np.argmax: first maximum
      best_count = v;
      best = a;
    }
    total += unit_t ? (double)v : pow((double)v, inv_t);  // utils.py:286-287 (Python sum, left to right)
  }
  if (!(total > 0.0)) {
    out_action[env] = -1;
    if (out_entropy) out_entropy[env] = 0.0;
    return;
  }
  // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, side='right')
  double last = 0.0;
  for (int a = 0; a < A; ++a) {
    const double x = unit_t ? (double)c[a] : pow((double)c[a], inv_t);
    last += x / total;
  }
  int action = best;
  double ent = 0.0;
  if (!deterministic) {
    const double u = uniform[env];
    double acc = 0.0;
    int idx = 0;
    for (int a = 0; a < A; ++a) {
      const double x = unit_t ? (double)c[a] : pow((double)c[a], inv_t);
      acc += x / total;
      if (acc / last <= u) idx = a + 1;  // side='right': number of cdf entries <= u
    }
    action = idx < A ? idx : A - 1;
  }
  if (out_entropy) {
    // scipy.stats.entropy(pk, base=2): pk /= sum(pk); sum(-pk*log(pk)) / log(2)
    double psum = 0.0;
    for (int a = 0; a < A; ++a) psum += (unit_t ? (double)c[a] : pow((double)c[a], inv_t)) / total;
    for (int a = 0; a < A; ++a) {
      const double pk = ((unit_t ? (double)c[a] : pow((double)c[a], inv_t)) / total) / psum;
      if (pk > 0.0) ent -= pk * log(pk);
    }
    out_entropy[env] = ent / log(2.0);
  }
  out_action[env] = action;
}

extern "C" int hz_select_action(int N, int A, int32_t* counts, const uint8_t* legal, const double* uniform,
                                float temperature, int deterministic, int32_t* out_action, double* out_entropy,
                                void* stream) {
  HZ_REQUIRE(N > 0 && A > 0 && A <= 64, "hz_select_action: bad sizes N=%d A=%d", N, A);
  HZ_REQUIRE(counts && legal && out_action, "hz_select_action: NULL argument");
  HZ_REQUIRE(deterministic || uniform, "hz_select_action: uniform samples required when sampling");
  HZ_REQUIRE(temperature > 0.0f, "hz_select_action: temperature must be > 0");
  hipLaunchKernelGGL(k_select_action, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, A, counts, legal,
                     uniform, temperature, deterministic, out_action, out_entropy);
  HZ_HIP(hipGetLastError());
  return 0;
}

// ---- finished-game flush: masked row scatter ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_scatter(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      long long row_bytes, const int32_t* __restrict__ slot, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int s = slot[row];
  if (s < 0) return;
  const uint8_t* a = src + (size_t)row * (size_t)row_bytes;
  uint8_t* b = dst + (size_t)s * (size_t)row_bytes;
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes) & 15) == 0) {
    for (long long off = (long long)lane * 16; off < row_bytes; off += 64 * 16)
      *reinterpret_cast<uint4*>(b + off) = *reinterpret_cast<const uint4*>(a + off);
  } else if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes) & 3) == 0) {
    for (long long off = (long long)lane * 4; off < row_bytes; off += 64 * 4)
      *reinterpret_cast<uint32_t*>(b + off) = *reinterpret_cast<const uint32_t*>(a + off);
  } else {
    for (long long off = lane; off < row_bytes; off += 64) b[off] = a[off];
  }
}

extern "C" int hz_rows_scatter(const void* src, void* dst, int64_t row_bytes, const int32_t* slot, int num_rows,
                               void* stream) {
  HZ_REQUIRE(src && dst && slot, "hz_rows_scatter: NULL argument");
  HZ_REQUIRE(num_rows > 0 && row_bytes > 0, "hz_rows_scatter: num_rows and row_bytes must be positive");
  hipLaunchKernelGGL(k_rows_scatter, dim3((num_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                     (uint8_t*)dst, (long long)row_bytes, slot, num_rows);
  HZ_HIP(hipGetLastError());
  return 0;
}
