// ref_tree_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" driver around the *genuine* reference tree
// (/root/reference/core/ctree/{cnode,cminimax}.{h,cpp}), compiled in place by
// oracle/Makefile into oracle/_ref/libref_tree.so.  It replaces the Cython
// binding core/ctree/cytree.pyx (same calls, same argument order) with flat
// arrays so that ctypes/numpy can drive it, and it binds libc rand()/srand()
// (cnode.cpp:369,411) to the counter-based stream of include/hz_tiebreak.h.
//
// Two modes:
//   mode 0  one CRoots(N, A, pool) exactly as cytree.Roots builds it
//           (cytree.pyx:42-45); rand() returns 0 ("first of the ties").
//   mode 1  N independent CRoots(1, A, pool) driven tree by tree, rand()
//           returns hz_tiebreak_rand(seed, tree, sim, depth).  Trees are
//           independent in the reference (the only loop-carried value,
//           parent_q at cnode.cpp:414-423, is 0 whenever it is read), which
//           tools/gen_golden.py::check_tree_equivalence asserts whenever the
//           goldens are generated: mode 0 against mode 2 (= mode 1's N
//           single-root CRoots under the rand()==0 stream of mode 0).
//
// Only tests/, tools/gen_golden.py and bench.py's cpu_baseline leg load this.
#include <cstdint>
#include <cstring>
#include <vector>

#include "cnode.h"
#include "hz_tiebreak.h"

namespace {
struct Shim {
  int mode = 0;          // 0: rand()==0 ; 1: hashed
  uint64_t seed = 0;
  uint32_t tree = 0, sim = 0, depth = 0;
} g_shim;
}  // namespace

// Bound inside this .so by -Wl,-Bsymbolic-functions (oracle/Makefile).
extern "C" int rand(void) __THROW {
  if (g_shim.mode == 0) return 0;
  return (int)hz_tiebreak_rand(g_shim.seed, g_shim.tree, g_shim.sim, g_shim.depth++);
}
extern "C" void srand(unsigned int) __THROW {}

namespace {

struct RefTree {
  int N, A, S, mode;
  // mode 0
  tree::CRoots* roots = nullptr;
  tools::CMinMaxStatsList* mm = nullptr;
  tree::CSearchResults* results = nullptr;
  // mode 1 (and mode 2 = per-tree structure with rand()==0)
  std::vector<tree::CRoots*> roots1;
  std::vector<tools::CMinMaxStatsList*> mm1;
  std::vector<tree::CSearchResults*> results1;
};

std::vector<std::vector<float>> rows_f(const float* p, int n, int a) {
  std::vector<std::vector<float>> v(n);
  for (int i = 0; i < n; ++i) v[i].assign(p + (size_t)i * a, p + (size_t)(i + 1) * a);
  return v;
}
std::vector<std::vector<int>> rows_i(const int* p, int n, int a) {
  std::vector<std::vector<int>> v(n);
  for (int i = 0; i < n; ++i) v[i].assign(p + (size_t)i * a, p + (size_t)(i + 1) * a);
  return v;
}

}  // namespace

extern "C" {

// mode: 0 = one N-root CRoots, rand()==0; 1 = N single-root CRoots, hashed rand();
//       2 = N single-root CRoots, rand()==0 (equivalence check against mode 0).
void* ref_tree_new(int N, int A, int S, int mode, uint64_t seed) {
  RefTree* t = new RefTree();
  t->N = N; t->A = A; t->S = S; t->mode = mode;
  int pool = A * (S + 2);  // cytree.pyx:44
  g_shim.seed = seed;
  if (mode == 0) {
    t->roots = new tree::CRoots(N, A, pool);
    t->mm = new tools::CMinMaxStatsList(N);
  } else {
    for (int i = 0; i < N; ++i) {
      t->roots1.push_back(new tree::CRoots(1, A, pool));
      t->mm1.push_back(new tools::CMinMaxStatsList(1));
      t->results1.push_back(nullptr);
    }
  }
  return t;
}

void ref_tree_free(void* h) {
  RefTree* t = (RefTree*)h;
  delete t->roots; delete t->mm; delete t->results;
  for (auto p : t->roots1) delete p;
  for (auto p : t->mm1) delete p;
  for (auto p : t->results1) delete p;
  delete t;
}

void ref_tree_set_delta(void* h, float d) {
  RefTree* t = (RefTree*)h;
  if (t->mode == 0) t->mm->set_delta(d);
  else for (auto p : t->mm1) p->set_delta(d);
}

// with_noise != 0 -> CRoots::prepare, else prepare_no_noise (cnode.cpp:247-259)
void ref_tree_prepare(void* h, int with_noise, float frac, const float* noises,
                      const float* rewards, const float* logits, const int* legal) {
  RefTree* t = (RefTree*)h;
  int N = t->N, A = t->A;
  if (t->mode == 0) {
    std::vector<float> r(rewards, rewards + N);
    if (with_noise) t->roots->prepare(frac, rows_f(noises, N, A), r, rows_f(logits, N, A), rows_i(legal, N, A));
    else t->roots->prepare_no_noise(r, rows_f(logits, N, A), rows_i(legal, N, A));
  } else {
    for (int i = 0; i < N; ++i) {
      std::vector<float> r(1, rewards[i]);
      if (with_noise)
        t->roots1[i]->prepare(frac, rows_f(noises + (size_t)i * A, 1, A), r,
                              rows_f(logits + (size_t)i * A, 1, A), rows_i(legal + (size_t)i * A, 1, A));
      else
        t->roots1[i]->prepare_no_noise(r, rows_f(logits + (size_t)i * A, 1, A),
                                       rows_i(legal + (size_t)i * A, 1, A));
    }
  }
}

// cytree.multi_traverse (cytree.pyx:97-101): fresh ResultsWrapper, then cmulti_traverse.
void ref_tree_traverse(void* h, int sim, int pb_c_base, float pb_c_init, float discount,
                       int* ix, int* iy, int* last_action) {
  RefTree* t = (RefTree*)h;
  int N = t->N;
  if (t->mode == 0) {
    g_shim.mode = 0;
    delete t->results;
    t->results = new tree::CSearchResults(N);
    tree::cmulti_traverse(t->roots, pb_c_base, pb_c_init, discount, t->mm, *t->results);
    for (int i = 0; i < N; ++i) {
      ix[i] = t->results->hidden_state_index_x_lst[i];
      iy[i] = t->results->hidden_state_index_y_lst[i];
      last_action[i] = t->results->last_actions[i];
    }
  } else {
    for (int i = 0; i < N; ++i) {
      g_shim.mode = (t->mode == 1) ? 1 : 0;
      g_shim.tree = (uint32_t)i; g_shim.sim = (uint32_t)sim; g_shim.depth = 0;
      delete t->results1[i];
      t->results1[i] = new tree::CSearchResults(1);
      tree::cmulti_traverse(t->roots1[i], pb_c_base, pb_c_init, discount, t->mm1[i], *t->results1[i]);
      ix[i] = t->results1[i]->hidden_state_index_x_lst[0];
      iy[i] = i;  // a single-root CRoots reports 0; the N-root one reports the tree index
      last_action[i] = t->results1[i]->last_actions[0];
    }
  }
}

// path length (number of nodes incl. root and leaf) of the last traverse, per tree
void ref_tree_path_len(void* h, int* out) {
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i)
    out[i] = (t->mode == 0) ? (int)t->results->search_paths[i].size()
                            : (int)t->results1[i]->search_paths[0].size();
}

// cytree.multi_back_propagate (cytree.pyx:87-94)
void ref_tree_backprop(void* h, int hidden_state_index_x, float discount, const float* rewards,
                       const float* values, const float* logits) {
  RefTree* t = (RefTree*)h;
  int N = t->N, A = t->A;
  if (t->mode == 0) {
    std::vector<float> r(rewards, rewards + N), v(values, values + N);
    tree::cmulti_back_propagate(hidden_state_index_x, discount, r, v, rows_f(logits, N, A), t->mm, *t->results);
  } else {
    for (int i = 0; i < N; ++i) {
      std::vector<float> r(1, rewards[i]), v(1, values[i]);
      tree::cmulti_back_propagate(hidden_state_index_x, discount, r, v,
                                  rows_f(logits + (size_t)i * A, 1, A), t->mm1[i], *t->results1[i]);
    }
  }
}

void ref_tree_distributions(void* h, int* out) {  // [N][A]
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i) {
    std::vector<int> d = (t->mode == 0) ? t->roots->roots[i].get_children_distribution()
                                        : t->roots1[i]->roots[0].get_children_distribution();
    for (int a = 0; a < t->A; ++a) out[(size_t)i * t->A + a] = a < (int)d.size() ? d[a] : 0;
  }
}

void ref_tree_values(void* h, float* out) {  // [N]
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i)
    out[i] = (t->mode == 0) ? t->roots->roots[i].value() : t->roots1[i]->roots[0].value();
}

// trajectories: out[N][max_len] padded with -1; returns nothing
void ref_tree_trajectories(void* h, int* out, int max_len) {
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i) {
    std::vector<int> tr = (t->mode == 0) ? t->roots->roots[i].get_trajectory()
                                         : t->roots1[i]->roots[0].get_trajectory();
    for (int k = 0; k < max_len; ++k) out[(size_t)i * max_len + k] = k < (int)tr.size() ? tr[k] : -1;
  }
}

void ref_tree_minmax(void* h, float* mn, float* mx) {
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i) {
    tools::CMinMaxStats& s = (t->mode == 0) ? t->mm->stats_lst[i] : t->mm1[i]->stats_lst[0];
    mn[i] = s.minimum; mx[i] = s.maximum;
  }
}

// root children priors after prepare (pins expand + add_exploration_noise bit-exactly)
void ref_tree_root_priors(void* h, float* out) {  // [N][A]
  RefTree* t = (RefTree*)h;
  for (int i = 0; i < t->N; ++i) {
    tree::CNode& r = (t->mode == 0) ? t->roots->roots[i] : t->roots1[i]->roots[0];
    for (int a = 0; a < t->A; ++a) out[(size_t)i * t->A + a] = r.get_child(a)->prior;
  }
}

}  // extern "C"
