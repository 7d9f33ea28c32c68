/* hz_tiebreak.h -- the deterministic tie-break stream of the search tree.
 *
 * The reference picks among near-equal pUCT children with libc rand():
 *     core/ctree/cnode.cpp:369   rand() % max_index_lst.size()
 * after reseeding from gettimeofday() on every traverse call
 * (core/ctree/cnode.cpp:409-411), so the reference itself is not reproducible
 * and its stream is consumed tree after tree, which would serialise a GPU.
 *
 * This library defines the value that call returns as a counter-based hash of
 * (seed, tree id, simulation index, depth of the selecting node).  It is in
 * [0, 2^31) like glibc's rand().  The reference build used as the oracle
 * (oracle/ref_tree_harness.cpp) binds rand() to exactly this function, so the
 * reference, the C restatement and the HIP kernels consume one stream.
 */
#ifndef HZ_TIEBREAK_H
#define HZ_TIEBREAK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define HZ_HD __host__ __device__
#else
#define HZ_HD
#endif

/* depth: 0 for the selection made at the root, 1 for the next node down, ... */
static inline HZ_HD uint32_t hz_tiebreak_rand(uint64_t seed, uint32_t tree,
                                              uint32_t sim, uint32_t depth) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull *
                          ((((uint64_t)tree) << 32) | (((uint64_t)sim & 0xFFFFu) << 16) |
                           ((uint64_t)depth & 0xFFFFu));
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 33);
}

#endif /* HZ_TIEBREAK_H */
