/* mt_discrete.c -- TEST INFRASTRUCTURE (oracle).  See mt_discrete.h for what is restated and why. */
#include "mt_discrete.h"

#include <math.h>

/* libstdc++ random.tcc: mersenne_twister_engine::seed(result_type) */
void hzo_mt_seed(hzo_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) {
    uint32_t x = g->mt[i - 1];
    x ^= x >> 30;
    g->mt[i] = 1812433253u * x + (uint32_t)i;
  }
  g->idx = 624;
}

/* libstdc++ random.tcc: mersenne_twister_engine::_M_gen_rand() */
static void hzo_mt_twist(hzo_mt19937* g) {
  const uint32_t upper = 0x80000000u, lower = 0x7fffffffu, a = 0x9908b0dfu;
  uint32_t* mt = g->mt;
  for (int k = 0; k < 624 - 397; ++k) {
    uint32_t y = (mt[k] & upper) | (mt[k + 1] & lower);
    mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? a : 0u);
  }
  for (int k = 624 - 397; k < 623; ++k) {
    uint32_t y = (mt[k] & upper) | (mt[k + 1] & lower);
    mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? a : 0u);
  }
  uint32_t y = (mt[623] & upper) | (mt[0] & lower);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? a : 0u);
  g->idx = 0;
}

/* libstdc++ random.tcc: mersenne_twister_engine::operator()() */
uint32_t hzo_mt_next(hzo_mt19937* g) {
  if (g->idx >= 624) hzo_mt_twist(g);
  uint32_t z = g->mt[g->idx++];
  z ^= (z >> 11) & 0xffffffffu;
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

/* random.tcc:3348-3380 with _RealType=double, bits=53, range 2^32 -> m = 2 draws */
double hzo_canonical53(hzo_mt19937* g) {
  double sum = 0.0, tmp = 1.0;
  for (int k = 2; k != 0; --k) {
    sum += (double)hzo_mt_next(g) * tmp;
    tmp *= 4294967296.0;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

/* random.tcc:2656-2677 + 2698-2713 */
int hzo_discrete(hzo_mt19937* g, const double* w, int n) {
  if (n < 2) return 0; /* _M_prob.clear(): _M_cp stays empty -> result 0, no draw */
  double prob[64], cp[64];
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += w[i]; /* std::accumulate(…, 0.0) */
  for (int i = 0; i < n; ++i) prob[i] = w[i] / sum;
  double acc = prob[0];
  cp[0] = acc;
  for (int i = 1; i < n; ++i) { /* std::partial_sum */
    acc = acc + prob[i];
    cp[i] = acc;
  }
  cp[n - 1] = 1.0;
  double p = hzo_canonical53(g);
  /* std::lower_bound: first i with !(cp[i] < p) */
  int lo = 0, len = n;
  while (len > 0) {
    int half = len >> 1;
    if (cp[lo + half] < p) {
      lo = lo + half + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  return lo;
}
