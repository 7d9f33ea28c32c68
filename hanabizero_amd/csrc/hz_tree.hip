// hz_tree.hip -- the pUCT search tree of HanabiZero as hand-written HIP for gfx950 (MI355X).
//
// What it replaces (reference, /root/reference): core/ctree/cnode.{h,cpp} + cminimax.{h,cpp} behind
// core/ctree/cytree.pyx.  Written from scratch for CDNA4 -- no shared code with the reference or the oracle.
//
// Data layout in HBM (per hz_tree_t; N trees, A actions, S = num_simulations expandable entries per tree):
//   rec   [N][S][A] float4   one 16-byte record per CHILD edge: {prior, value_sum, reward, bits(visit<<16 | child+1)}
//                            entry e of a tree is an expanded node (e == its hidden_state_index_x; entry 0 = root);
//                            its A children sit in rec[tree][e][0..A): one dwordx4 load per lane reads a whole level.
//   qsa   [N][S]    f32      cached reward + discount*value() of expanded entry e (e >= 1) for the min-max pass
//   ref   [N][S]    i32      (parent_entry << 8 | action) of entry e
//   path  [N][S+1]  i32      (entry << 8 | action) per depth of the last descent (the reference's CSearchResults)
//   root_visit/root_vsum/mm_min/mm_max/path_len [N];  best_action [N][S] i8
// Execution model: ONE 64-lane wavefront per tree, lane a owns child a (A <= 64), 4 trees per 256-thread block.
// Order-sensitive fp32 reductions of the reference (get_mean_q, policy_sum, legal_noise, the backup chain) are
// evaluated in the reference's order with v_readlane broadcasts; order-free ones (max, min) with wave butterflies.
// Compiled with -ffp-contract=off: one rounding per operation, like the reference's x86-64 build.
#include <math.h>
#include <stdarg.h>

#include "hz_common.h"
#include "hz_tree.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void hz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* hz_last_error(void) { return g_err; }
extern "C" int hz_version(void) { return 1; }

#include "hz_tree_dev.h"

// ------------------------------------------------------------------------------------------ prepare
// CRoots::prepare / prepare_no_noise (cnode.cpp:247-259): reset the tree, expand the root (hidden index (0, tree)),
// mix in exploration noise (cnode.cpp:116-142).
__global__ __launch_bounds__(256) void k_prepare(TreeView tv, float frac, const float* __restrict__ noises,
                                                 const float* __restrict__ rewards,
                                                 const float* __restrict__ logits,
                                                 const uint8_t* __restrict__ legal) {
  const int lane = threadIdx.x & 63;
  const int tree = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tree >= tv.N) return;
  const int A = tv.A, S = tv.S;
  const bool on = lane < A;
  const float logit = on ? logits[(size_t)tree * A + lane] : 0.0f;
  const int lg = on ? (int)legal[(size_t)tree * A + lane] : 0;
  const uint64_t legal_mask = __ballot(on && lg != 0);  // expand skips legal == 0 (cnode.cpp:71)
  float prior = expand_prior(logit, legal_mask, lane, A);
  if (noises != nullptr) {
    const float nz = on ? noises[(size_t)tree * A + lane] : 0.0f;
    // legal_noise: sum in action order over legal == 1 (cnode.cpp:120-129)
    float legal_noise = 0.0f;
    uint64_t mm = __ballot(on && lg == 1);
    while (mm) {
      const int a = __ffsll((unsigned long long)mm) - 1;
      mm &= mm - 1;
      legal_noise += hz_readlane_f(nz, a);
    }
    if (lg <= 0) {
      prior = 0.0f;  // cnode.cpp:131-135
    } else {
      const float noise = nz / legal_noise;          // cnode.cpp:136
      prior = prior * (1 - frac) + noise * frac;     // cnode.cpp:140 (three roundings, no FMA)
    }
  }
  if (on) {
    float4 r;
    r.x = prior; r.y = 0.0f; r.z = 0.0f; r.w = __uint_as_float(pack_vc(0, -1));
    tv.rec[((size_t)tree * S + 0) * A + lane] = r;
  }
  for (int e = lane; e < S; e += 64) tv.best_action[(size_t)tree * S + e] = -1;
  if (lane == 0) {
    tv.root_visit[tree] = 0;
    tv.root_vsum[tree] = 0.0f;
    tv.mm_min[tree] = HZ_FLOAT_MAX;  // CMinMaxStats ctor (cminimax.cpp:5-9)
    tv.mm_max[tree] = HZ_FLOAT_MIN;
    tv.path_len[tree] = 0;
  }
  (void)rewards;  // the root's own reward is never read by the search (cnode.cpp:303,330 skip the root)
}

__global__ __launch_bounds__(256) void k_traverse(TreeView tv, int sim, TraverseOut to) {
  const int lane = threadIdx.x & 63;
  const int tree = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SALU addressing
  if (tree >= tv.N) return;
  traverse_body(tv, tree, lane, sim, tv.mm_min[tree], tv.mm_max[tree], tv.root_visit[tree], to, false,
                make_float4(0.f, 0.f, 0.f, 0.f));
}

template <bool FUSED>
__global__ __launch_bounds__(256) void k_backprop(TreeView tv, int e_new, NetOut no) {
  extern __shared__ float lds_q[];  // [4 waves][S]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SALU addressing and hashing
  const int tree = blockIdx.x * 4 + wave;
  if (tree >= tv.N) return;
  TP_ON(0);
  float mn, mx;
  int rv, a0;
  float4 first;
  backprop_body<FUSED>(tv, tree, lane, wave, lds_q, e_new, no, mn, mx, rv, first, a0);
}

// multi_back_propagate of simulation k immediately followed by multi_traverse of simulation k+1 on the same tree by
// the same wave: one launch instead of two per simulation, and the new min/max and root count never leave registers.
// The descent reads child records this wave has just stored: a workgroup-scope fence orders them (same CU, same L1).
template <bool FUSED>
__global__ __launch_bounds__(256) void k_backprop_traverse(TreeView tv, int e_new, NetOut no, int sim_next, TraverseOut to) {
  extern __shared__ float lds_q[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SALU addressing and hashing
  const int tree = blockIdx.x * 4 + wave;
  if (tree >= tv.N) return;
  // the root's child row of the coming descent: fetched now, the one record the backup changes is patched in registers
  TP_ON(1);
  TP(0);
  float4 root_row = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < tv.A) root_row = tv.rec[(size_t)tree * tv.S * tv.A + lane];
  float mn, mx;
  int rv, a0;
  float4 first;
  backprop_body<FUSED>(tv, tree, lane, wave, lds_q, e_new, no, mn, mx, rv, first, a0);
  if (lane == a0) root_row = first;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  TP(5);
  traverse_body(tv, tree, lane, sim_next, mn, mx, rv, to, true, root_row);
  TP(13);
}

// ------------------------------------------------------------------------------------------ read-outs
__global__ void k_distributions(TreeView tv, int32_t* __restrict__ out) {  // cnode.cpp:205-214, 276-284
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tv.N * tv.A) return;
  const int tree = i / tv.A, a = i % tv.A;
  const float4 r = tv.rec[((size_t)tree * tv.S) * tv.A + a];
  out[i] = (int)(__float_as_uint(r.w) >> 16);
}

__global__ void k_root_stats(TreeView tv, int32_t* __restrict__ counts, float* __restrict__ values) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // get_distributions + get_values in one launch
  if (i >= tv.N * tv.A) return;
  const int tree = i / tv.A, a = i % tv.A;
  const float4 r = tv.rec[((size_t)tree * tv.S) * tv.A + a];
  counts[i] = (int)(__float_as_uint(r.w) >> 16);
  if (a == 0) {
    const int v = tv.root_visit[tree];
    values[tree] = v == 0 ? 0.0f : tv.root_vsum[tree] / (float)v;
  }
}

__global__ void k_values(TreeView tv, float* __restrict__ out) {  // cnode.cpp:286-292
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tv.N) return;
  const int v = tv.root_visit[i];
  out[i] = v == 0 ? 0.0f : tv.root_vsum[i] / (float)v;
}

__global__ void k_trajectories(TreeView tv, int32_t* __restrict__ out, int max_len) {  // cnode.cpp:191-203
  const int tree = blockIdx.x * blockDim.x + threadIdx.x;
  if (tree >= tv.N) return;
  int e = 0, k = 0;
  while (k < max_len && e >= 0) {
    const int ba = tv.best_action[(size_t)tree * tv.S + e];
    if (ba < 0) break;
    out[(size_t)tree * max_len + k++] = ba;
    const float4 r = tv.rec[((size_t)tree * tv.S + e) * tv.A + ba];
    e = (int)(__float_as_uint(r.w) & 0xffffu) - 1;
  }
  for (; k < max_len; ++k) out[(size_t)tree * max_len + k] = -1;
}

__global__ void k_root_priors(TreeView tv, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tv.N * tv.A) return;
  const int tree = i / tv.A, a = i % tv.A;
  out[i] = tv.rec[((size_t)tree * tv.S) * tv.A + a].x;
}

__global__ void k_copy_minmax(TreeView tv, float* __restrict__ mn, float* __restrict__ mx, int32_t* __restrict__ pl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tv.N) return;
  if (mn) mn[i] = tv.mm_min[i];
  if (mx) mx[i] = tv.mm_max[i];
  if (pl) pl[i] = tv.path_len[i];
}

// ------------------------------------------------------------------------------------------ C ABI
template <typename T>
static int dev_alloc(T** p, size_t n, int64_t* total) {
  HZ_HIP(hipMalloc((void**)p, n * sizeof(T)));
  *total += (int64_t)(n * sizeof(T));
  return 0;
}

extern "C" int hz_tree_create(hz_tree_t** out, int N, int A, int S, int device) {
  HZ_REQUIRE(out != nullptr, "hz_tree_create: out is NULL");
  HZ_REQUIRE(N > 0, "hz_tree_create: num_trees must be > 0 (got %d)", N);
  HZ_REQUIRE(A > 0 && A <= HZ_MAX_ACTIONS, "hz_tree_create: num_actions must be in [1, %d] (got %d)", HZ_MAX_ACTIONS, A);
  HZ_REQUIRE(S >= 2 && S <= HZ_MAX_SIMULATIONS, "hz_tree_create: num_simulations must be in [2, %d] (got %d)",
             HZ_MAX_SIMULATIONS, S);
  HZ_HIP(hipSetDevice(device));
  hz_tree* t = new hz_tree();
  memset(t, 0, sizeof(*t));
  t->predicted_lines = 1;
  t->N = N; t->A = A; t->S = S; t->device = device;
  int rc = 0;
  rc |= dev_alloc(&t->rec, (size_t)N * S * A, &t->bytes);
  rc |= dev_alloc(&t->qsa, (size_t)N * S, &t->bytes);
  rc |= dev_alloc(&t->ref, (size_t)N * S, &t->bytes);
  rc |= dev_alloc(&t->path, (size_t)N * (S + 1), &t->bytes);
  rc |= dev_alloc(&t->prec, (size_t)N * (S + 1), &t->bytes);
  rc |= dev_alloc(&t->path_len, (size_t)N, &t->bytes);
  rc |= dev_alloc(&t->root_visit, (size_t)N, &t->bytes);
  rc |= dev_alloc(&t->root_vsum, (size_t)N, &t->bytes);
  rc |= dev_alloc(&t->mm_min, (size_t)N, &t->bytes);
  rc |= dev_alloc(&t->mm_max, (size_t)N, &t->bytes);
  rc |= dev_alloc(&t->best_action, (size_t)N * S, &t->bytes);
  rc |= dev_alloc(&t->pbc_tab, (size_t)S + 1, &t->bytes);
  if (rc != 0) {
    hz_tree_destroy(t);
    return -2;
  }
  HZ_HIP(hipMemset(t->path_len, 0, sizeof(int32_t) * N));
  *out = t;
  // reference defaults (core/config.py:107-111, config/hanabi_control/__init__.py:25-27)
  return hz_tree_set_params(t, 19652, 1.25f, 0.999f, 0.0f, 0, 0);
}

extern "C" int hz_tree_destroy(hz_tree_t* t) {
  if (!t) return 0;
  (void)hipSetDevice(t->device);
  void* bufs[] = {t->rec, t->qsa, t->ref, t->path, t->prec, t->path_len, t->root_visit, t->root_vsum, t->mm_min,
                  t->mm_max, t->best_action, t->pbc_tab};
  for (void* b : bufs) (void)hipFree(b);
  delete t;
  return 0;
}

extern "C" int hz_tree_set_params(hz_tree_t* t, int pb_c_base, float pb_c_init, float discount,
                                  float value_delta_max, uint64_t tie_seed, uint32_t tree_id_base) {
  HZ_REQUIRE(t != nullptr, "hz_tree_set_params: NULL handle");
  HZ_REQUIRE(pb_c_base > 0, "hz_tree_set_params: pb_c_base must be > 0");
  // (a caller that only moves to another tie-break stream -- the reanalyze search takes a new seed per learner step -- must not
  // pay the table's synchronous upload below: on the null stream it would wait for every other stream of the device)
  const bool same_table = t->params_set && t->pb_c_base == pb_c_base && t->pb_c_init == pb_c_init;
  t->pb_c_base = pb_c_base; t->pb_c_init = pb_c_init; t->discount = discount; t->delta = value_delta_max;
  t->seed = tie_seed; t->id_base = tree_id_base;
  if (same_table) return 0;
  // pb_c's first factor depends only on the parent visit count n <= S: tabulate it with the host libm logf
  // the reference links against (cnode.cpp:385), same expression, same fp32 roundings.
  float* tab = new float[t->S + 1];
  const float base = (float)pb_c_base;
  for (int n = 0; n <= t->S; ++n) {
    const float pvc = (float)n;
    volatile float num = pvc + base;
    num = num + 1;
    volatile float ratio = num / base;
    tab[n] = logf(ratio) + pb_c_init;
  }
  HZ_HIP(hipSetDevice(t->device));
  hipError_t e = hipMemcpy(t->pbc_tab, tab, sizeof(float) * (t->S + 1), hipMemcpyHostToDevice);
  delete[] tab;
  HZ_HIP(e);
  t->params_set = 1;
  return 0;
}

static inline dim3 tree_grid(const hz_tree* t) { return dim3((unsigned)((t->N + 3) / 4)); }

extern "C" int hz_tree_prepare(hz_tree_t* t, float frac, const float* noises, const float* rewards,
                               const float* logits, const uint8_t* legal, void* stream) {
  HZ_REQUIRE(t != nullptr, "hz_tree_prepare: NULL handle");
  HZ_REQUIRE(logits != nullptr && legal != nullptr, "hz_tree_prepare: policy_logits and legal must not be NULL");
  hipLaunchKernelGGL(k_prepare, tree_grid(t), dim3(256), 0, (hipStream_t)stream, view(t), frac, noises, rewards,
                     logits, legal);
  HZ_HIP(hipGetLastError());
  t->next_entry = 1;
  return 0;
}

static int launch_traverse(hz_tree_t* t, int sim, int32_t* ix, int32_t* iy, int32_t* la, const void* pool,
                           int row_bytes, void* net_in, int stride_bytes, int onehot_cols, int dtype, void* stream) {
  HZ_REQUIRE(t != nullptr, "hz_tree_traverse: NULL handle");
  HZ_REQUIRE(t->next_entry >= 1, "hz_tree_traverse: call hz_tree_prepare first");
  HZ_REQUIRE(ix && iy && la, "hz_tree_traverse: output pointers must not be NULL");
  HZ_REQUIRE(sim >= 0 && sim < 65536, "hz_tree_traverse: sim out of range (%d)", sim);
  TraverseOut to;
  to.ix = ix; to.iy = iy; to.la = la; to.pool = (const uint8_t*)pool; to.net_in = (uint8_t*)net_in;
  to.row_bytes = row_bytes; to.net_in_stride_bytes = stride_bytes; to.onehot_cols = onehot_cols; to.dtype = dtype;
  to.tree0 = 0;
  hipLaunchKernelGGL(k_traverse, tree_grid(t), dim3(256), 0, (hipStream_t)stream, view(t), sim, to);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_traverse(hz_tree_t* t, int sim, int32_t* ix, int32_t* iy, int32_t* la, void* stream) {
  return launch_traverse(t, sim, ix, iy, la, nullptr, 0, nullptr, 0, 0, HZ_F32, stream);
}

extern "C" int hz_tree_traverse_gather(hz_tree_t* t, int sim, int32_t* ix, int32_t* iy, int32_t* la,
                                       const void* pool, int hidden, int dtype, void* net_in, int net_in_stride,
                                       int action_onehot_cols, void* stream) {
  HZ_REQUIRE(pool != nullptr && net_in != nullptr, "hz_tree_traverse_gather: pool and net_in must not be NULL");
  HZ_REQUIRE(dtype == HZ_F32 || dtype == HZ_BF16 || dtype == HZ_F16, "hz_tree_traverse_gather: bad dtype %d", dtype);
  const int es = dtype == HZ_F32 ? 4 : 2;
  HZ_REQUIRE(hidden > 0 && (hidden * es) % 16 == 0, "hz_tree_traverse_gather: hidden*elem_size must be a multiple of 16 B");
  HZ_REQUIRE(action_onehot_cols == 0 || action_onehot_cols >= t->A,
             "hz_tree_traverse_gather: action_onehot_cols must be 0 or >= num_actions");
  HZ_REQUIRE(net_in_stride >= hidden + action_onehot_cols && (net_in_stride * es) % 16 == 0,
             "hz_tree_traverse_gather: net_in_stride*elem_size must be a multiple of 16 B and >= hidden + one-hot columns");
  HZ_REQUIRE(((uintptr_t)pool % 16) == 0 && ((uintptr_t)net_in % 16) == 0, "hz_tree_traverse_gather: pointers must be 16-B aligned");
  return launch_traverse(t, sim, ix, iy, la, pool, hidden * es, net_in, net_in_stride * es, action_onehot_cols, dtype,
                         stream);
}

extern "C" int hz_tree_backprop(hz_tree_t* t, int hidden_state_index_x, const float* rewards, const float* values,
                                const float* logits, void* stream) {
  HZ_REQUIRE(t != nullptr, "hz_tree_backprop: NULL handle");
  HZ_REQUIRE(rewards && values && logits, "hz_tree_backprop: input pointers must not be NULL");
  HZ_REQUIRE(hidden_state_index_x >= 1 && hidden_state_index_x < t->S,
             "hz_tree_backprop: hidden_state_index_x=%d outside [1, num_simulations=%d)", hidden_state_index_x, t->S);
  HZ_REQUIRE(hidden_state_index_x == t->next_entry,
             "hz_tree_backprop: hidden_state_index_x must advance 1,2,3,... after prepare (expected %d, got %d)",
             t->next_entry, hidden_state_index_x);
  const size_t lds = (size_t)4 * t->S * sizeof(float);
  NetOut no;
  memset(&no, 0, sizeof(no));
  no.rewards = rewards; no.values = values; no.logits = logits;
  hipLaunchKernelGGL(k_backprop<false>, tree_grid(t), dim3(256), lds, (hipStream_t)stream, view(t),
                     hidden_state_index_x, no);
  HZ_HIP(hipGetLastError());
  t->next_entry = hidden_state_index_x + 1;
  return 0;
}

extern "C" int hz_tree_backprop_traverse(hz_tree_t* t, int hidden_state_index_x, const float* rewards, const float* values,
                                        const float* logits, int next_sim, int32_t* ix, int32_t* iy, int32_t* la,
                                        void* stream) {
  HZ_REQUIRE(t != nullptr, "hz_tree_backprop_traverse: NULL handle");
  HZ_REQUIRE(rewards && values && logits && ix && iy && la, "hz_tree_backprop_traverse: pointers must not be NULL");
  HZ_REQUIRE(hidden_state_index_x >= 1 && hidden_state_index_x < t->S,
             "hz_tree_backprop_traverse: hidden_state_index_x=%d outside [1, num_simulations=%d)", hidden_state_index_x, t->S);
  HZ_REQUIRE(hidden_state_index_x == t->next_entry,
             "hz_tree_backprop_traverse: hidden_state_index_x must advance 1,2,3,... after prepare (expected %d, got %d)",
             t->next_entry, hidden_state_index_x);
  HZ_REQUIRE(next_sim >= 0 && next_sim < 65536, "hz_tree_backprop_traverse: next_sim out of range (%d)", next_sim);
  NetOut no;
  memset(&no, 0, sizeof(no));
  no.rewards = rewards; no.values = values; no.logits = logits;
  TraverseOut to;
  memset(&to, 0, sizeof(to));
  to.ix = ix; to.iy = iy; to.la = la;
  const size_t lds = (size_t)4 * t->S * sizeof(float);
  hipLaunchKernelGGL(k_backprop_traverse<false>, tree_grid(t), dim3(256), lds, (hipStream_t)stream, view(t),
                     hidden_state_index_x, no, next_sim, to);
  HZ_HIP(hipGetLastError());
  t->next_entry = hidden_state_index_x + 1;
  return 0;
}

extern "C" int hz_tree_backprop_nets(hz_tree_t* t, int hidden_state_index_x, const void* reward_logits,
                                     int64_t reward_stride, const void* value_logits, int64_t value_stride,
                                     int support_size, int support_min, const void* policy_logits,
                                     int64_t policy_stride, int dtype, float* out_rewards, float* out_values,
                                     void* stream) {
  HZ_REQUIRE(t != nullptr, "hz_tree_backprop_nets: NULL handle");
  HZ_REQUIRE(reward_logits && value_logits && policy_logits, "hz_tree_backprop_nets: input pointers must not be NULL");
  HZ_REQUIRE(dtype == HZ_F32 || dtype == HZ_BF16 || dtype == HZ_F16, "hz_tree_backprop_nets: bad dtype %d", dtype);
  HZ_REQUIRE(support_size > 0 && reward_stride >= support_size && value_stride >= support_size && policy_stride >= t->A,
             "hz_tree_backprop_nets: strides shorter than the rows");
  HZ_REQUIRE(hidden_state_index_x >= 1 && hidden_state_index_x < t->S,
             "hz_tree_backprop_nets: hidden_state_index_x=%d outside [1, num_simulations=%d)", hidden_state_index_x, t->S);
  HZ_REQUIRE(hidden_state_index_x == t->next_entry,
             "hz_tree_backprop_nets: hidden_state_index_x must advance 1,2,3,... after prepare (expected %d, got %d)",
             t->next_entry, hidden_state_index_x);
  NetOut no;
  memset(&no, 0, sizeof(no));
  no.reward_logits = reward_logits; no.value_logits = value_logits; no.policy_logits = policy_logits;
  no.reward_stride = reward_stride; no.value_stride = value_stride; no.policy_stride = policy_stride;
  no.support_size = support_size; no.support_min = support_min; no.dtype = dtype;
  no.out_rewards = out_rewards; no.out_values = out_values;
  const size_t lds = (size_t)4 * t->S * sizeof(float);
  hipLaunchKernelGGL(k_backprop<true>, tree_grid(t), dim3(256), lds, (hipStream_t)stream, view(t), hidden_state_index_x,
                     no);
  HZ_HIP(hipGetLastError());
  t->next_entry = hidden_state_index_x + 1;
  return 0;
}

extern "C" int hz_tree_get_distributions(hz_tree_t* t, int32_t* out, void* stream) {
  HZ_REQUIRE(t && out, "hz_tree_get_distributions: NULL argument");
  const int n = t->N * t->A;
  hipLaunchKernelGGL(k_distributions, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_root_stats(hz_tree_t* t, int32_t* counts, float* values, void* stream) {
  HZ_REQUIRE(t && counts && values, "hz_tree_get_root_stats: NULL argument");
  const int n = t->N * t->A;
  hipLaunchKernelGGL(k_root_stats, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), counts, values);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_values(hz_tree_t* t, float* out, void* stream) {
  HZ_REQUIRE(t && out, "hz_tree_get_values: NULL argument");
  hipLaunchKernelGGL(k_values, dim3((t->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_trajectories(hz_tree_t* t, int32_t* out, int max_len, void* stream) {
  HZ_REQUIRE(t && out && max_len > 0, "hz_tree_get_trajectories: bad argument");
  hipLaunchKernelGGL(k_trajectories, dim3((t->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), out, max_len);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_minmax(hz_tree_t* t, float* mn, float* mx, void* stream) {
  HZ_REQUIRE(t && mn && mx, "hz_tree_get_minmax: NULL argument");
  hipLaunchKernelGGL(k_copy_minmax, dim3((t->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), mn, mx,
                     (int32_t*)nullptr);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_path_len(hz_tree_t* t, int32_t* out, void* stream) {
  HZ_REQUIRE(t && out, "hz_tree_get_path_len: NULL argument");
  hipLaunchKernelGGL(k_copy_minmax, dim3((t->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t),
                     (float*)nullptr, (float*)nullptr, out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_tree_get_root_priors(hz_tree_t* t, float* out, void* stream) {
  HZ_REQUIRE(t && out, "hz_tree_get_root_priors: NULL argument");
  const int n = t->N * t->A;
  hipLaunchKernelGGL(k_root_priors, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, view(t), out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t hz_tree_hbm_bytes(const hz_tree_t* t) { return t ? t->bytes : 0; }

extern "C" int hz_tree_copy(hz_tree_t* dst, const hz_tree_t* src, void* stream) {
  HZ_REQUIRE(dst && src, "hz_tree_copy: NULL handle");
  HZ_REQUIRE(dst->N == src->N && dst->A == src->A && dst->S == src->S && dst->device == src->device,
             "hz_tree_copy: handles must have the same shape and device");
  const size_t N = src->N, A = src->A, S = src->S;
  hipStream_t st = (hipStream_t)stream;
#define HZ_CP(f, n) HZ_HIP(hipMemcpyAsync(dst->f, src->f, (n) * sizeof(*src->f), hipMemcpyDeviceToDevice, st))
  HZ_CP(rec, N * S * A); HZ_CP(qsa, N * S); HZ_CP(ref, N * S); HZ_CP(path, N * (S + 1)); HZ_CP(prec, N * (S + 1)); HZ_CP(path_len, N);
  HZ_CP(root_visit, N); HZ_CP(root_vsum, N); HZ_CP(mm_min, N); HZ_CP(mm_max, N); HZ_CP(best_action, N * S);
  HZ_CP(pbc_tab, S + 1);
#undef HZ_CP
  dst->pb_c_base = src->pb_c_base; dst->pb_c_init = src->pb_c_init; dst->discount = src->discount;
  dst->delta = src->delta; dst->seed = src->seed; dst->id_base = src->id_base; dst->params_set = src->params_set;
  dst->next_entry = src->next_entry; dst->predicted_lines = src->predicted_lines;
  return 0;
}

// the scalar transform of hz_tree_backprop_nets on its own (same device function, hence bit-identical to it)
__global__ __launch_bounds__(256) void k_support_to_scalar(const uint8_t* __restrict__ logits, long long stride_bytes,
                                                           int V, int support_min, int dtype, float* __restrict__ out,
                                                           int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float v = support_to_scalar(logits + (size_t)row * (size_t)stride_bytes, V, support_min, dtype, lane);
  if (lane == 0) out[row] = v;
}

extern "C" int hz_support_to_scalar(const void* logits, int64_t stride, int support_size, int support_min, int dtype,
                                    float* out, int num_rows, void* stream) {
  HZ_REQUIRE(logits && out && num_rows > 0 && support_size > 0 && stride >= support_size, "hz_support_to_scalar: bad argument");
  HZ_REQUIRE(dtype == HZ_F32 || dtype == HZ_BF16 || dtype == HZ_F16, "hz_support_to_scalar: bad dtype %d", dtype);
  const int es = dtype == HZ_F32 ? 4 : 2;
  hipLaunchKernelGGL(k_support_to_scalar, dim3((num_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)logits, (long long)stride * es, support_size, support_min, dtype, out, num_rows);
  HZ_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------ test hooks
// Device expf over an array, and a blocked checksum over ALL 2^32 float bit patterns (tests/test_hip_math.py
// compares it with the same checksum of the host libm expf the reference links against).
__global__ void k_expf(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = hz_expf(x[i]);
}

__global__ __launch_bounds__(256) void k_expf_checksum(uint64_t* __restrict__ out) {
  // block b covers bit patterns [b << 20, (b+1) << 20); checksum = sum over patterns of (bits(expf) * odd multiplier)
  __shared__ uint64_t part[256];
  const uint32_t base = (uint32_t)blockIdx.x << 20;
  uint64_t acc = 0;
  for (uint32_t i = threadIdx.x; i < (1u << 20); i += 256) {
    const uint32_t u = base + i;
    const float r = hz_expf(__uint_as_float(u));
    uint32_t rb = __float_as_uint(r);
    if (r != r) rb = 0x7fc00000u;  // canonical NaN
    acc += (uint64_t)rb * (2ull * (uint64_t)i + 1ull);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

extern "C" int hz_test_expf(const float* x, float* y, int64_t n, void* stream) {
  hipLaunchKernelGGL(k_expf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_test_expf_checksum(uint64_t* out4096, void* stream) {
  hipLaunchKernelGGL(k_expf_checksum, dim3(4096), dim3(256), 0, (hipStream_t)stream, out4096);
  HZ_HIP(hipGetLastError());
  return 0;
}
