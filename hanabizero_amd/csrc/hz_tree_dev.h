// hz_tree_dev.h -- device code of the search tree shared by hz_tree.hip (one launch per phase) and hz_search.hip (the
// persistent search kernel): handle, view, descent and backup bodies.  See hz_tree.hip for the algorithm notes.
#pragma once
#include <math.h>

#include "hz_common.h"
#include "hz_tree.h"

// ------------------------------------------------------------------------------------------ handle
struct hz_tree {
  int N, A, S, device;
  int pb_c_base;
  float pb_c_init, discount, delta;
  uint64_t seed;
  uint32_t id_base;
  int params_set;
  int next_entry;  // host-side guard: the entry the next backprop must create
  int predicted_lines;  // hz_search_run launches the kernels that walk predicted lines (hz_search_set_predicted_lines; default 1)
  float4* rec;
  float* qsa;
  int32_t* ref;
  int32_t* path;
  float4* prec;  // [N][S+1] the child record read at each depth of the last descent (what the backup will update)
  int32_t* path_len;
  int32_t* root_visit;
  float* root_vsum;
  float* mm_min;
  float* mm_max;
  int8_t* best_action;
  float* pbc_tab;  // [S+1]: logf((n + base + 1) / base) + pb_c_init  for parent visit count n
  int64_t bytes;
};

struct TreeView {
  int N, A, S;
  float discount, delta;
  uint64_t seed;
  uint32_t id_base;
  float4* rec;
  float* qsa;
  int32_t* ref;
  int32_t* path;
  float4* prec;
  int32_t* path_len;
  int32_t* root_visit;
  float* root_vsum;
  float* mm_min;
  float* mm_max;
  int8_t* best_action;
  const float* pbc_tab;
};

static inline TreeView view(const hz_tree* t) {
  TreeView v;
  v.N = t->N; v.A = t->A; v.S = t->S;
  v.discount = t->discount; v.delta = t->delta; v.seed = t->seed; v.id_base = t->id_base;
  v.rec = t->rec; v.qsa = t->qsa; v.ref = t->ref; v.path = t->path; v.prec = t->prec; v.path_len = t->path_len;
  v.root_visit = t->root_visit; v.root_vsum = t->root_vsum; v.mm_min = t->mm_min; v.mm_max = t->mm_max;
  v.best_action = t->best_action; v.pbc_tab = t->pbc_tab;
  return v;
}

__device__ __forceinline__ uint32_t pack_vc(int visit, int child) { return ((uint32_t)visit << 16) | (uint32_t)(child + 1); }

// masked softmax -> priors, the arithmetic of CNode::expand (cnode.cpp:49-114).
// lane a holds logit a; `legal_mask` bit a set = legal.  Returns this lane's prior.
__device__ __forceinline__ float expand_prior(float logit, uint64_t legal_mask, int lane, int A,
                                              const uint64_t* exp_tab = hz_exp2f_tab) {
  const bool legal = (legal_mask >> lane) & 1ull;
  // policy_max = max over legal, non-NaN logits, starting from FLOAT_MIN (cnode.cpp:59,68-78)
  float m = (legal && logit == logit) ? logit : -INFINITY;
  m = hz_wave_max(m);
  const float policy_max = fmaxf(m, HZ_FLOAT_MIN);
  const float tp = legal ? hz_expf(logit - policy_max, exp_tab) : 0.0f;  // cnode.cpp:87
  // policy_sum: 1e-4 + terms in action order over the legal children (cnode.cpp:57,88)
  const float policy_sum = legal_mask ? hz_ordered_sum(tp, 64 - __clzll((unsigned long long)legal_mask), 0.0001f) : 0.0001f;
  float prior = legal ? tp / policy_sum : 0.0f;  // cnode.cpp:98-103
  if (prior != prior) prior = 0.0f;              // cnode.cpp:107-109
  (void)A;
  return prior;
}

// ------------------------------------------------------------------------------------------ traverse
// cmulti_traverse (cnode.cpp:407-441) with get_mean_q (:144-164), cselect_child (:346-374), cucb_score (:376-405)
// and CMinMaxStats::normalize (cminimax.cpp:31-44).  Optionally fused with the hidden-state gather of
// core/mcts.py:31-36.
struct TraverseOut {
  int32_t* ix;
  int32_t* iy;
  int32_t* la;
  const uint8_t* pool;
  uint8_t* net_in;
  int row_bytes, net_in_stride_bytes, onehot_cols, dtype;
  int tree0;  // row of net_in that tree `tree0` owns is row 0 (0 for the whole-batch buffer)
};

#ifndef HZ_TREE_REPLAY
#define HZ_TREE_REPLAY 1
#endif
#ifndef HZ_REPLAY_G  // lanes per level / children per lane of a pass (A <= 20 needs G C >= 20): 4, 5 = sixteen levels per pass; 8, 3 = eight
#define HZ_REPLAY_G 4
#define HZ_REPLAY_C 5
#endif
#ifndef HZ_TREE_REPLAY_MIN  // levels of a descent from which on the tree's descents look for predicted lines
#define HZ_TREE_REPLAY_MIN 6
#endif
// one descent of one tree by one wave; mn / mx / root_visit are passed in registers so that the fused
// backup+descent kernel does not have to re-read what it has just computed
// Diagnostic build only (-DHZ_TREE_PROFILE, tools/tree_profile.py): s_memtime stamps of the wave that owns tree 400.
#ifdef HZ_TREE_PROFILE
__device__ unsigned long long hz_tree_prof[16];
__device__ int hz_tree_prof_on;  // 1 while the wave of tree 400 runs the fused kernel (k_backprop shares the body)
extern "C" int hz_tree_profile_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_tree_prof), sizeof(hz_tree_prof));
}
#define TP(i) do { if (tree == 400 && lane == 0 && hz_tree_prof_on) hz_tree_prof[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TP_ON(v) do { if (tree == 400 && lane == 0) hz_tree_prof_on = (v); } while (0)
// sums over every level of every descent of the wave that owns tree HZ_TREE_PROFILE_TREE, or (HZ_TREE_PROFILE_TREE < 0) of
// trees 17, 81, 145, ... in slots tree / 64 (tools/level_profile.py): fields 0..7 = the segments of a level between the TPL
// stamps, 8 = levels, 9 / 10 = cycles / levels of whole descents at least HZ_TREE_PROFILE_MINDEPTH levels long
#ifndef HZ_TREE_PROFILE_TREE
#define HZ_TREE_PROFILE_TREE 400
#endif
#ifndef HZ_TREE_PROFILE_MINDEPTH
#define HZ_TREE_PROFILE_MINDEPTH 0
#endif
__device__ unsigned long long hz_tree_lvl[64 * 32];
extern "C" int hz_tree_level_profile_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_tree_lvl), sizeof(hz_tree_lvl));
}
#define TPL_DECL unsigned long long tpl_prev = __builtin_amdgcn_s_memtime(), tpl_acc[9] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; \
                 const unsigned long long tpl_t0 = tpl_prev
#define TPL_WAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// (i: segment that ends here; at_depth: the level it belongs to)
#define TPL(i, at_depth) do { const unsigned long long tpl_now = __builtin_amdgcn_s_memtime();                         \
                    if ((at_depth) >= HZ_TREE_PROFILE_MINDEPTH) tpl_acc[(i)] += ((i) == 8) ? 1ull : tpl_now - tpl_prev;  \
                    tpl_prev = __builtin_amdgcn_s_memtime(); } while (0)
#define TPL_FLUSH do { if ((HZ_TREE_PROFILE_TREE < 0 ? (tree & 63) == 17 && tree < 4096 : tree == HZ_TREE_PROFILE_TREE) && lane == 0) { \
                         unsigned long long* slot = hz_tree_lvl + 32 * (HZ_TREE_PROFILE_TREE < 0 ? tree >> 6 : 0);     \
                         for (int q = 0; q < 9; ++q) slot[q] += tpl_acc[q];                                             \
                         if (depth >= HZ_TREE_PROFILE_MINDEPTH && depth >= 2) {                                         \
                           slot[9] += __builtin_amdgcn_s_memtime() - tpl_t0;                                            \
                           slot[10] += (unsigned long long)depth; slot[15] += (unsigned long long)tpr_levels; } } } while (0)
// replays (hz_tree_replay_dev.h): fields 11 = replays, 12 = levels they covered, 13 = their cycles, 14 = descents
#define TPR_T0 const unsigned long long tpr_t0 = __builtin_amdgcn_s_memtime()
#define TPR_LEVELS_DECL int tpr_levels = 0
#define TPR_LEVELS(n) tpr_levels = (n)
#define TPR(levels) do { if ((HZ_TREE_PROFILE_TREE < 0 ? (tree & 63) == 17 && tree < 4096 : tree == HZ_TREE_PROFILE_TREE) && lane == 0) { \
                      unsigned long long* slot = hz_tree_lvl + 32 * (HZ_TREE_PROFILE_TREE < 0 ? tree >> 6 : 0);        \
                      slot[11] += 1; slot[12] += (unsigned long long)(levels);                                          \
                      slot[13] += __builtin_amdgcn_s_memtime() - tpr_t0; } } while (0)
// inside a pass (hz_tree_replay_dev.h): fields 16.. = segments between the TPP stamps, 24 = passes
#define TPP_DECL unsigned long long tpp_prev = __builtin_amdgcn_s_memtime()
#define TPP(i) do { const unsigned long long tpp_now = __builtin_amdgcn_s_memtime();                                     \
                    if ((HZ_TREE_PROFILE_TREE < 0 ? (tree & 63) == 17 && tree < 4096 : tree == HZ_TREE_PROFILE_TREE) && lane == 0) \
                      hz_tree_lvl[32 * (HZ_TREE_PROFILE_TREE < 0 ? tree >> 6 : 0) + 16 + (i)] += ((i) == 8) ? 1ull : tpp_now - tpp_prev; \
                    tpp_prev = __builtin_amdgcn_s_memtime(); } while (0)
#define TPR_DESCENT do { if ((HZ_TREE_PROFILE_TREE < 0 ? (tree & 63) == 17 && tree < 4096 : tree == HZ_TREE_PROFILE_TREE) && lane == 0) \
                      hz_tree_lvl[32 * (HZ_TREE_PROFILE_TREE < 0 ? tree >> 6 : 0) + 14] += 1; } while (0)
#else
#define TPR_T0 do { } while (0)
#define TPP_DECL do { } while (0)
#define TPP(i) do { } while (0)
#define TPR_LEVELS_DECL do { } while (0)
#define TPR_LEVELS(n) do { } while (0)
#define TPR(levels) do { } while (0)
#define TPR_DESCENT do { } while (0)
#define TP(i) do { } while (0)
#define TP_ON(v) do { } while (0)
#define TPL_DECL do { } while (0)
#define TPL_WAIT do { } while (0)
#define TPL(i, at_depth) do { } while (0)
#define TPL_FLUSH do { } while (0)
#endif

// Per-tree search state a persistent kernel keeps on chip from simulation to simulation (hz_search.hip): the last
// descent's path and records and the q cache in LDS, the root's running sums in registers.  With LOCAL = false the
// bodies read all of it from the tree's arrays in HBM (one launch per phase) and this struct is ignored.
struct TreeLocal {
  int32_t* path;    // [S+1]  (LDS)
  int32_t* nextact;   // [64] (LDS) or null: per node its last selection, HZ_NEXTACT (hz_tree_replay_dev.h)
  float4* prec;     // [S+1]  (LDS)
  float root_vsum;  // the root's value_sum / visit_count (also stored to HBM for the read-outs)
  int root_visit;
  int path_len;     // nodes on the last descent's path
  bool publish;     // store this call's bookkeeping scalars (leaf entry, path length, root sums, min / max) to HBM: only
                    // the last descent's and the last backup's are ever read (the read-outs after the search)
  float pbc_reg, sqrt_reg;  // per-lane tables of the descent (pb_c's log factor, sqrt(n + 1)), loaded / computed once
  const float* lq;          // LDS: this tree's q cache (backprop_body), [S]
  const float* ptab;        // LDS, or null: pb_c(parent visits) * (sqrt(parent visits + 1) / (visits + 1)), the exploration
                            // factor of cnode.cpp:386, for every pair of counts a search can meet (hz_ptab_index) -- built once
                            // per launch with the very operations the descent would use, so a level reads one word instead of
                            // issuing two lane reads, a conversion, a correctly rounded division and a product
  const uint64_t* exp_tab;  // hz_exp2f_tab in LDS
  bool deep;                // a descent of this tree has been HZ_TREE_REPLAY_MIN levels long: it keeps `nextact` (non-null then) and its descents look for predicted lines
  float leaf_reward, leaf_value;  // the leaf's outputs for the coming backup (uniform), and lane a's policy logit
  float leaf_logit;
};

// The table of exploration factors (TreeLocal::ptab) is triangular: a child has been visited at most as often as its parent
// (every visit of the child is one of the parent; inactive lanes read visit 0), (S + 1)(S + 2) / 2 words.  (An index past a row's
// end -- impossible while the counts are what the backups made them -- would read a neighbouring LDS word, not fault.)
__device__ __forceinline__ int hz_ptab_index(int parent_visits, int visits) {
  return ((parent_visits * (parent_visits + 1)) >> 1) + visits;
}
__device__ __forceinline__ int hz_ptab_words(int S) { return ((S + 1) * (S + 2)) >> 1; }
__device__ __forceinline__ void hz_ptab_fill(float* ptab, const float* pbc_tab, int S, int tid, int nthreads) {
  // (the operations, operand for operand, of traverse_body's own computation of this factor)
  for (int pvc = 0; pvc <= S; ++pvc) {
    const float sq = sqrtf((float)pvc + 1.0f);
    const float pb = pbc_tab[pvc];
    for (int visit = tid; visit <= pvc; visit += nthreads) ptab[hz_ptab_index(pvc, visit)] = pb * (sq / (float)(visit + 1));
  }
}

#include "hz_tree_replay_dev.h"

// Issue priority of the tree phases inside the persistent search kernels.  The four waves of a SIMD share its issue port and
// the sequencer serves the oldest first: with equal priorities the youngest wave of each SIMD needs 21 k cycles for a tree
// phase the oldest gets through in 12 k (tools/search_profile.py), and the inference waits for the slowest tree.  Youngest
// first throughout was worth +0.9 % moves/s at 4096 envs (A/B on one box, tools/ab_multi.sh) but only moved the problem (now
// the oldest waves needed 21 k cycles at 8192 envs); priorities that grow with the depth of the descent did nothing.  What is
// in: youngest first for the first half of a tree phase (read-out, expansion, backup: hz_tree_phase_prio), equal priorities
// -- the hardware's oldest first -- for the descent (hz_tree_descent_prio), so that neither age group is last in both
// halves: another +1.0 % at 4096 envs, +1.5 % at 8192 (oldest-first made explicit in the descent, the halves the other way
// round, or the switch behind the read-out or behind the expansion instead: all less).
__device__ __forceinline__ void hz_tree_descent_prio() { __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void hz_tree_phase_prio() {
  switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8)) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

template <bool LOCAL = false, bool REPLAY = false>
__device__ __forceinline__ void traverse_body(const TreeView& tv, int tree, int lane, int sim, float mn, float mx,
                                              int root_visit, const TraverseOut& to, bool have_root, float4 root_row,
                                              int* out_entry = nullptr, TreeLocal* tl = nullptr) {
  const int A = tv.A, S = tv.S;
  const bool on = lane < A;
  const float discount = tv.discount;
  const float delta = mx - mn;
  const float4* rec = tv.rec + (size_t)tree * S * A;
  int32_t* path = LOCAL ? tl->path : tv.path + (size_t)tree * (S + 1);
  float4* prec = LOCAL ? tl->prec : tv.prec + (size_t)tree * (S + 1);
  // pb_c's first factor for every possible parent visit count, one per lane (S + 1 <= 64: no dependent table load
  // on the critical path of a level); larger S falls back to the table in memory
  const bool tab_in_regs = LOCAL || S < 64;  // (LOCAL: S < 64 is hz_search_run's condition -- no code for more in the persistent kernel)
  const float pbc_reg = LOCAL ? tl->pbc_reg : ((tab_in_regs && lane <= S) ? tv.pbc_tab[lane] : 0.0f);
  const float sqrt_reg = LOCAL ? tl->sqrt_reg : sqrtf((float)lane + 1.0f);  // sqrt(parent visits + 1), same trick

  int e = 0;
  int pvc = root_visit;
  bool is_root = true;
  float parent_q = 0.0f;  // cnode.cpp:414 (0 whenever it is read, see oracle/ref_tree_harness.cpp)
  int depth = 0;
  int action = 0;
  // the level's child records: the root's come in registers from the fused backup (have_root), a node's are requested at the
  // end of the level above
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  bool at_leaf = false;
  // LOCAL: in a tree that has grown deep, the predicted line below a node is walked sixteen levels at a time
  // (hz_tree_replay_dev.h) wherever one is known; one level at a time from where none is
  const bool replay = LOCAL && REPLAY && HZ_TREE_REPLAY && A <= 20 && tl->ptab != nullptr && tl->nextact != nullptr && hz_uniform(tl->deep ? 1 : 0) != 0;
  TPR_DESCENT;
  TPR_LEVELS_DECL;
  // ("unlikely": the register allocator then spills here rather than in the ordinary walk's loop.  Two copies of the walk -- one
  // for trees that have not grown deep, without any of this, as in the side-by-side kernel -- cost MORE here: 1.15 % instead of 0.6 %.)
  if (__builtin_expect(replay, 0)) {
    TPR_T0;
    ReplayIn in;
    in.A = A; in.S = S; in.tree = tree; in.sim = sim;
    in.mn = mn; in.mx = mx; in.discount = discount; in.delta_floor = tv.delta;
    in.seed = tv.seed; in.id_base = tv.id_base;
    in.rec = rec; in.best_action = tv.best_action + (size_t)tree * S;
    in.nextact = tl->nextact; in.path = path; in.prec = prec; in.lq = tl->lq; in.ptab = tl->ptab;
    // the last backup's stores to the records must have landed (the ordinary walk waits for them behind its root level)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    while (true) {
      const ReplayOut ro = traverse_replay<HZ_REPLAY_G, HZ_REPLAY_C>(in, e, depth, parent_q, pvc);
      // (said to the compiler in so many words: all of this is wave-uniform -- what it cannot prove it computes per lane,
      // and the ordinary walk below would inherit that)
      e = hz_uniform(ro.e);
      action = hz_uniform(ro.action);
      depth = hz_uniform(ro.depth);
      pvc = hz_uniform(ro.pvc);
      parent_q = __int_as_float(hz_uniform(__float_as_int(ro.parent_q)));
      is_root = false;
      at_leaf = hz_uniform(ro.leaf ? 1 : 0) != 0;
      if (at_leaf) break;
      const int na = hz_uniform(tl->nextact[e]);  // another pass if the node has been passed before and its choice led to an expanded node
      if (!((na & 0x10000) && ((na >> 8) & 0xff))) break;
    }
    TPR(depth);
    TPR_LEVELS(depth);
    if (!at_leaf && on) r = rec[(size_t)e * A + lane];
  } else if (have_root) {
    r = root_row;
  } else if (on) {
    r = rec[lane];
  }
  TPL_DECL;
  while (!at_leaf) {
    TPL_WAIT;
    TPL(0, depth);
    const uint32_t w = __float_as_uint(r.w);
    const int visit = (int)(w >> 16);
    const int child = (int)(w & 0xffffu) - 1;
    float prior = r.x;
    if (prior != prior) prior = 0.0f;  // cnode.cpp:379-381
    float qsa;
    if (LOCAL) {
      // reward + discount * value_sum / visits of a visited child is what the last backup over that edge left in the q cache
      // (backprop_body: lq[child], same expression, same operands): one LDS word instead of a correctly rounded division
      qsa = r.z + discount * 0.0f;
      if (visit > 0) qsa = tl->lq[child];
    } else {
      const float val = (visit == 0) ? 0.0f : r.y / (float)visit;  // CNode::value cnode.cpp:180-189
      qsa = r.z + discount * val;
    }
    // get_mean_q: sum over visited children in action order
    uint64_t vm = __ballot(on && visit > 0);
    const int nvis = __popcll((unsigned long long)vm);
    float total = 0.0f;
    if (nvis > 4) {  // many visited children (the root, late in the search): one systolic pass over the lanes
      total = hz_ordered_sum((on && visit > 0) ? qsa : 0.0f, 64 - __clzll((unsigned long long)vm), 0.0f);
    } else {
      while (vm) {
        const int a = __ffsll((unsigned long long)vm) - 1;
        vm &= vm - 1;
        total += hz_readlane_f(qsa, a);
      }
    }
    const bool root_mean = is_root && nvis > 0;  // cnode.cpp:228-236: the root leaves its own q out
    const float mean_q = (root_mean ? total : parent_q + total) / (float)(root_mean ? nvis : nvis + 1);
    is_root = false;
    parent_q = mean_q;
    TPL(1, depth);
    // cucb_score
    float pb_c;
    if (LOCAL && tl->ptab != nullptr) {
      pb_c = tl->ptab[hz_ptab_index(pvc, visit)];
    } else {
      pb_c = tab_in_regs ? hz_readlane_f(pbc_reg, pvc) : tv.pbc_tab[pvc];  // logf((n+base+1)/base) + pb_c_init
      const float sq = tab_in_regs ? hz_readlane_f(sqrt_reg, pvc) : sqrtf((float)pvc + 1.0f);
      pb_c = pb_c * (sq / (float)(visit + 1));  // cnode.cpp:386
    }
    const float prior_score = pb_c * prior;
    float vs = (visit == 0) ? mean_q : qsa;
    if (delta > 0.0f) vs = (vs - mn) / (delta < tv.delta ? tv.delta : delta);  // CMinMaxStats::normalize
    if (vs < 0.0f) vs = 0.0f;
    if (vs > 1.0f) vs = 1.0f;
    const float score = prior_score + vs;
    // cselect_child: final tie list = {first arg-max} U {later children within epsilon of the max}
    const bool valid = on && (score == score) && (score > HZ_FLOAT_MIN);
    const float M = hz_wave_max(valid ? score : -INFINITY);
    const uint64_t eq = __ballot(valid && score == M);
    TPL(2, depth);
    action = 0;
    if (eq != 0) {
      const int first = __ffsll((unsigned long long)eq) - 1;
      const float thr = M - 0.000001f;
      uint64_t cand = __ballot(valid && score >= thr);
      cand &= ~((1ull << first) - 1ull);
      const uint32_t cnt = (uint32_t)__popcll((unsigned long long)cand);
      if (cnt > 1) {  // (x % 1 == 0: the draw only matters when there is a tie)
        const uint32_t rnd = hz_tiebreak_rand(tv.seed, tv.id_base + (uint32_t)tree, (uint32_t)sim, (uint32_t)depth);
        uint32_t k = rnd % cnt;  // rand() % max_index_lst.size()  (cnode.cpp:369)
        while (k--) cand &= cand - 1;
      }
      action = __ffsll((unsigned long long)cand) - 1;
    }
    action = hz_uniform(action);
    TPL(3, depth);
    // LOCAL (persistent search kernel): the backup that ran just before this descent stored child records the next levels may
    // load.  Its stores are waited for HERE -- a level's worth of work after they were issued, so the wait is over before it
    // begins -- and before this level's own store goes out, which nothing below depends on.
    if (LOCAL && depth == 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (lane == 0) {
      tv.best_action[(size_t)tree * S + e] = (int8_t)action;  // node->best_action (cnode.cpp:426)
      path[depth] = (e << 8) | action;
    }
    if (lane == action) prec[depth] = r;  // the record the coming backup updates: saves it a dependent load
    TPL(4, depth);
    const int child_e = hz_readlane_i(child, action);
    const int child_visit = hz_readlane_i(visit, action);
    if (LOCAL && REPLAY && HZ_TREE_REPLAY && replay && lane == 0) tl->nextact[e] = HZ_NEXTACT(child_e, action);  // (a deep tree's table)
    ++depth;
    TP(5 + (depth < 7 ? depth : 7));
    if (child_e < 0 || depth >= S) {  // leaf reached (second clause: defensive bound, never true)
      TPL(5, depth - 1);
      TPL(8, depth - 1);
      break;
    }
    e = child_e;
    pvc = child_visit;
    r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) r = rec[(size_t)e * A + lane];
    TPL(5, depth - 1);
    TPL(8, depth - 1);
  }
  TPL_FLUSH;
  if (lane == 0) {
    to.la[tree] = action;
    if (!LOCAL || tl->publish) {
      to.ix[tree] = e;     // parent->hidden_state_index_x (entry index == hidden_state_index_x)
      to.iy[tree] = tree;  // parent->hidden_state_index_y
      tv.path_len[tree] = depth + 1;
    }
  }
  if (LOCAL) tl->path_len = depth + 1;
  if (LOCAL && REPLAY && HZ_TREE_REPLAY && !replay && depth >= HZ_TREE_REPLAY_MIN && A <= 20 && tl->ptab != nullptr && tl->nextact != nullptr) {
    // the tree has grown deep: from now on its descents keep the table of last choices (the ordinary walk above, the passes,
    // the backup for the entry it expands) -- which starts as this path: level k's choice led to level k + 1's node, the last
    // level's edge leads nowhere yet.  Nodes off this path have no entry until they are passed again (a pass stops there).
    if (lane < depth) {
      const int here = path[lane], below = lane + 1 < depth ? (path[lane + 1] >> 8) : -1;
      tl->nextact[here >> 8] = HZ_NEXTACT(below, here & 255);
    }
    tl->deep = true;
  }
  if (out_entry) *out_entry = e;
  if (to.pool != nullptr) {
    // hidden_states[i] = hidden_state_pool[ix][iy]  (core/mcts.py:31-32), 16 B per lane per trip
    const uint8_t* src = to.pool + ((size_t)e * tv.N + tree) * (size_t)to.row_bytes;
    uint8_t* dst = to.net_in + (size_t)(tree - to.tree0) * (size_t)to.net_in_stride_bytes;
    for (int off = lane * 16; off < to.row_bytes; off += 64 * 16)
      *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
    // action_one_hot of MuZeroNet.dynamics (config/hanabi_control/model.py:215-219), appended after the state
    for (int c = lane; c < to.onehot_cols; c += 64) {
      const bool hot = (c == action);
      if (to.dtype == HZ_F32) reinterpret_cast<float*>(dst + to.row_bytes)[c] = hot ? 1.0f : 0.0f;
      else reinterpret_cast<uint16_t*>(dst + to.row_bytes)[c] = hot ? (to.dtype == HZ_BF16 ? 0x3f80u : 0x3c00u) : 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------ backprop
// cmulti_back_propagate (cnode.cpp:337-344): expand (all legal) + cback_propagate (:317-335) + update_tree_q
// (:296-315, here a wave min/max over the cached per-entry q values instead of a DFS of the whole tree).
// Where the leaf's (reward, value, policy logits) come from.  Plain: three fp32 arrays (the cytree signature).
// Fused: straight from the network heads -- categorical reward/value logits and policy logits in the net's dtype;
// the kernel applies core/mcts.py:48-49 (NaN logits -> 0) and core/config.py:210-232 (softmax . support -> h^-1).
struct NetOut {
  const float* rewards;
  const float* values;
  const float* logits;
  const void* reward_logits;
  const void* value_logits;
  const void* policy_logits;
  long long reward_stride, value_stride, policy_stride;  // elements
  int support_size, support_min, dtype;
  float* out_rewards;
  float* out_values;
};

__device__ __forceinline__ float load_as_f32(const void* p, long long i, int dtype) {
  if (dtype == HZ_F32) return ((const float*)p)[i];
  const uint16_t h = ((const uint16_t*)p)[i];
  if (dtype == HZ_BF16) return __uint_as_float((uint32_t)h << 16);
  return (float)(*reinterpret_cast<const _Float16*>(&h));
}

__device__ __forceinline__ float hz_wave_sum(float v) {
  v = hz_row16_sum(v);
  return (hz_readlane_f(v, 0) + hz_readlane_f(v, 16)) + (hz_readlane_f(v, 32) + hz_readlane_f(v, 48));
}

// inverse_scalar_transform (core/config.py:210-232, delta = 1, epsilon = 0.001) of one row of V categorical
// logits over the integer support [support_min, support_min + V), by one wave.  Tolerance-level arithmetic (this is
// a network output, north-star tolerance 1e-3); the tree arithmetic that consumes the scalar stays exact.
__device__ __forceinline__ float support_to_scalar(const void* row, int V, int support_min, int dtype, int lane) {
  float m = -INFINITY, se = 0.0f, sw = 0.0f;
  if (V <= 256) {  // the Hanabi supports (51 / 201 bins): every logit is read once and kept in registers
    float x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = lane + 64 * k;
      x[k] = (i < V) ? load_as_f32(row, i, dtype) : -INFINITY;
      m = fmaxf(m, x[k]);
    }
    m = hz_wave_max(m);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = lane + 64 * k;
      const float e = (i < V) ? __expf(x[k] - m) : 0.0f;
      se += e;
      sw += e * (float)(support_min + i);
    }
  } else {
    for (int i = lane; i < V; i += 64) m = fmaxf(m, load_as_f32(row, i, dtype));
    m = hz_wave_max(m);
    for (int i = lane; i < V; i += 64) {
      const float e = __expf(load_as_f32(row, i, dtype) - m);
      se += e;
      sw += e * (float)(support_min + i);
    }
  }
  se = hz_wave_sum(se);
  sw = hz_wave_sum(sw);
  const float v = sw / se;
  const float eps = 0.001f;
  const float t = (sqrtf(1.0f + 4.0f * eps * (fabsf(v) + 1.0f + eps)) - 1.0f) / (2.0f * eps);
  float out = t * t - 1.0f;
  if (v < 0.0f) out = -out;
  if (out != out) out = 0.0f;  // nan_part -> 0 (config.py:229-232)
  return out;
}

// one tree's expand + backup + min-max by one wave; returns the new (min, max, root visit count) in registers
template <bool FUSED, bool LOCAL = false>
__device__ __forceinline__ void backprop_body(const TreeView& tv, int tree, int lane, int wave, float* lds_q, int e_new,
                                              const NetOut& no, float& out_mn, float& out_mx, int& out_root_visit,
                                              float4& out_first_rec, int& out_first_action, TreeLocal* tl = nullptr) {
  const int A = tv.A, S = tv.S;
  const bool on = lane < A;
  const float discount = tv.discount;
  float4* rec = tv.rec + (size_t)tree * S * A;
  const int32_t* path = LOCAL ? tl->path : tv.path + (size_t)tree * (S + 1);
  const float4* prec = LOCAL ? tl->prec : tv.prec + (size_t)tree * (S + 1);
  float* lq = lds_q + wave * S;  // LOCAL: persists across the simulations (never re-staged from tv.qsa)

  // the path's first 64 edges and the records the descent read there, fetched before path_len is known (the buffers
  // hold S+1 slots per tree, so this never leaves them): every load of this function is issued up front, independent
  int pr0 = 0;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane <= S) {
    pr0 = path[lane];
    r0 = prec[lane];  // == rec[entry][action] as the descent read it
  }
  const float old_root_vsum = LOCAL ? tl->root_vsum : tv.root_vsum[tree];
  const int old_root_visit = LOCAL ? tl->root_visit : tv.root_visit[tree];
  // stage the cached q of entries 1..e_new-1 in LDS (coalesced), entry e_new is produced below
  if (!LOCAL)
    for (int e = 1 + lane; e < e_new; e += 64) lq[e] = tv.qsa[(size_t)tree * S + e];

  // expand the leaf: priors of the new entry's children
  float logit = 0.0f;
  if (on) {
    if (LOCAL) {
      logit = tl->leaf_logit;
    } else if (FUSED) {
      logit = load_as_f32(no.policy_logits, (long long)tree * no.policy_stride + lane, no.dtype);
      if (logit != logit) logit = 0.0f;  // core/mcts.py:48-49
    } else {
      logit = no.logits[(size_t)tree * A + lane];
    }
  }
  const uint64_t all = (A >= 64) ? ~0ull : ((1ull << A) - 1ull);
  const float prior = expand_prior(logit, all, lane, A, LOCAL ? tl->exp_tab : hz_exp2f_tab);
  TP(1);
  if (on) {
    float4 r;
    r.x = prior; r.y = 0.0f; r.z = 0.0f; r.w = __uint_as_float(pack_vc(0, -1));
    rec[(size_t)e_new * A + lane] = r;
  }

  const int npairs = (LOCAL ? tl->path_len : tv.path_len[tree]) - 1;  // edges on the path; the node below edge k is at depth k+1
  float G, leaf_reward;                      // bootstrap_value (cnode.cpp:318) and the leaf's reward
  if (LOCAL) {
    G = tl->leaf_value;
    leaf_reward = tl->leaf_reward;
  } else if (FUSED) {
    const int es = (no.dtype == HZ_F32) ? 4 : 2;
    const uint8_t* vrow = (const uint8_t*)no.value_logits + (size_t)tree * (size_t)no.value_stride * es;
    const uint8_t* rrow = (const uint8_t*)no.reward_logits + (size_t)tree * (size_t)no.reward_stride * es;
    G = support_to_scalar(vrow, no.support_size, no.support_min, no.dtype, lane);
    leaf_reward = support_to_scalar(rrow, no.support_size, no.support_min, no.dtype, lane);
    if (lane == 0) {
      if (no.out_values) no.out_values[tree] = G;
      if (no.out_rewards) no.out_rewards[tree] = leaf_reward;
    }
  } else {
    G = no.values[tree];
    leaf_reward = no.rewards[tree];
  }
  TP(2);
  int pr = 0;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  // (LOCAL: S < 64, the path fits one chunk -- and the persistent kernel's simulation loop holds no loop that is not needed:
  // see row32_support_to_scalar in hz_mlp_dev.h for what a loop costs there)
  for (int base = LOCAL ? 0 : (((npairs - 1) >> 6) << 6); base >= 0; base -= 64) {
    const int k = base + lane;
    const bool act = k < npairs;
    if (base == 0) {
      pr = pr0;
      r = r0;
    } else if (act) {
      pr = path[k];
      r = prec[k];
    }
    uint32_t w = __float_as_uint(r.w);
    int visit = (int)(w >> 16);
    int child = (int)(w & 0xffffu) - 1;
    if (act && k == npairs - 1) {  // the leaf: CNode::expand sets reward and hidden index (cnode.cpp:50-53)
      r.z = leaf_reward;
      child = e_new;
      if (!LOCAL) tv.ref[(size_t)tree * S + e_new] = pr;
      if (LOCAL && tl->deep) tl->nextact[pr >> 8] = HZ_NEXTACT(e_new, pr & 255);  // (a deep tree's table: the edge now leads to an entry)
    }
    // the backup chain, deepest node first: value_sum += G; G = reward + discount * G   (cnode.cpp:320-331)
    float myG = 0.0f;
    const int hi = min(npairs - 1 - base, 63);
    for (int j = hi; j >= 0; --j) {
      const float rj = hz_readlane_f(r.z, j);
      if (lane == j) myG = G;
      G = rj + discount * G;
    }
    if (act) {
      r.y += myG;
      visit += 1;
      r.w = __uint_as_float(pack_vc(visit, child));
      rec[(size_t)(pr >> 8) * A + (pr & 255)] = r;
      const float q = r.z + discount * (r.y / (float)visit);  // update_tree_q's qsa (cnode.cpp:304)
      lq[child] = q;
      if (!LOCAL) tv.qsa[(size_t)tree * S + child] = q;
    }
  }
  TP(3);
  // the loop ends with the chunk that holds path edge 0 in lane 0: the root's child record as just stored
  out_first_rec.x = hz_readlane_f(r.x, 0);
  out_first_rec.y = hz_readlane_f(r.y, 0);
  out_first_rec.z = hz_readlane_f(r.z, 0);
  out_first_rec.w = hz_readlane_f(r.w, 0);
  out_first_action = __builtin_amdgcn_readfirstlane(pr) & 255;
  const float new_root_vsum = old_root_vsum + G;  // the root (search_path[0])
  out_root_visit = old_root_visit + 1;
  if (lane == 0 && (!LOCAL || tl->publish)) {
    tv.root_vsum[tree] = new_root_vsum;
    tv.root_visit[tree] = out_root_visit;
  }
  if (LOCAL) {
    tl->root_vsum = new_root_vsum;
    tl->root_visit = out_root_visit;
  }
  // min_max_stats.clear(); update_tree_q(root): every expanded non-root node contributes (cnode.cpp:332-334)
  float vmax = -INFINITY, vmin = INFINITY;
  if (LOCAL) {  // (e_new <= S < 64: one entry per lane)
    const float q = 1 + lane <= e_new ? lq[1 + lane] : __builtin_nanf("");  // same-wave LDS traffic is processed in order
    if (q == q) {
      vmax = q;
      vmin = q;
    }
  } else {
    for (int e = 1 + lane; e <= e_new; e += 64) {
      const float q = lq[e];  // same-wave LDS traffic is processed in order: sees the stores above
      if (q == q) {
        vmax = fmaxf(vmax, q);
        vmin = fminf(vmin, q);
      }
    }
  }
  out_mx = fmaxf(hz_wave_max(vmax), HZ_FLOAT_MIN);  // CMinMaxStats::update from the cleared state (cminimax.cpp:17-29)
  out_mn = fminf(hz_wave_min(vmin), HZ_FLOAT_MAX);
  if (lane == 0 && (!LOCAL || tl->publish)) {
    tv.mm_max[tree] = out_mx;
    tv.mm_min[tree] = out_mn;
  }
  TP(4);
}

