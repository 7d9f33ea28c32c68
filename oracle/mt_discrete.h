/* mt_discrete.h -- TEST INFRASTRUCTURE (oracle): restatement of the libstdc++ <random> pieces the
 * reference's card dealing goes through.  The algorithm lives OUTSIDE /root/reference, in the
 * third-party dependency libstdc++ (GCC 11.4, /usr/include/c++/11/bits/random.{h,tcc}):
 *   std::mt19937                       random.h:  mersenne_twister_engine<uint32,32,624,397,31,0x9908b0df,11,
 *                                                 0xffffffff,7,0x9d2c5680,15,0xefc60000,18,1812433253>
 *   std::generate_canonical<double,53> random.tcc:3348-3380
 *   std::discrete_distribution<uint32> random.tcc:2656-2677 (_M_initialize), :2698-2713 (operator())
 * Reference call sites: envs/hanabi/hanabi_lib/hanabi_game.cc:48-51 (seed), :106-112 (PickRandomChance),
 * hanabi_state.cc:277-286, 313-325 (probabilities = count / deck_size as doubles, in chance-uid order).
 * Pinned by tests/golden/env_*.npz (deal sequences produced by the compiled reference).
 */
#ifndef HZO_MT_DISCRETE_H
#define HZO_MT_DISCRETE_H
#include <stdint.h>

typedef struct {
  uint32_t mt[624];
  int idx; /* 624 after seeding: the first draw regenerates the block */
} hzo_mt19937;

void hzo_mt_seed(hzo_mt19937* g, uint32_t seed);
uint32_t hzo_mt_next(hzo_mt19937* g);
/* generate_canonical<double,53>: two draws, low word first */
double hzo_canonical53(hzo_mt19937* g);
/* discrete_distribution over n weights (doubles).  n < 2 -> returns 0 WITHOUT drawing. */
int hzo_discrete(hzo_mt19937* g, const double* w, int n);

#endif
