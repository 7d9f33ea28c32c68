// hz_search.hip -- the whole search of one move (every simulation of core/mcts.py:26-58) as ONE persistent kernel.
//
// The S-1 simulations of a tree are a strict chain -- descent -> recurrent inference of the leaf -> expand + backup ->
// next descent -- but chains of different trees never meet.  So a workgroup keeps 16 trees for the whole search: its
// 16 wavefronts each own one tree (descent and backup: hz_tree_dev.h, one wave per tree as in hz_tree.hip), and
// together they are the 16 x 2-tile workgroup of the fused MFMA recurrent inference (hz_mlp_dev.h) over exactly
// those 16 rows.  Between the phases there is only a workgroup barrier: no launch, no grid-wide dependency, no wait for
// the deepest of ALL trees before any inference can start, and the leaf outputs (reward, value, policy logits, next
// hidden state) go from the MFMA phase to the backup phase of the same compute unit.
// 256 workgroups x 16 trees = the 4096 envs of the benchmark on the 256 CUs of an MI355X; more envs simply queue.
#include "hz_mlp_dev.h"
#include "hz_search.h"
#include "hz_tree_dev.h"
#include "hz_tree_half_dev.h"

struct SearchArgs {
  const hz_mlp_job_t* jobs;
  const uint16_t* wstream;
  const float* bias;
  const float* act_tab;
  uint16_t* pool;          // [S][N][hidden] bf16 / fp16 (the MLP header's dtype); plane 0 = root hidden states
  long long plane_stride;  // elements
  long long row_stride;
  int32_t* ix;             // [N] scratch: leaf parent entry of the current simulation (= pool plane)
  int32_t* iy;             // [N]
  int32_t* la;             // [N] last action
  float* rew;              // [N], [N], [N][A]: reserved (the leaf outputs never leave the chip)
  float* val;
  float* pol;
  int sims;                // simulations to run (S - 1, as the reference)
  int ptab;                // 1: the launch reserved hz_ptab_words(S) floats of LDS behind the other arrays for SearchLds::ptab
  int nextact;             // 1: ... and behind that [16][64] words (k_search: the nodes' last selections, hz_tree_replay_dev.h)
};

// (The phases must be inlined into the kernel: through a real call the compiler loses the address space of every
// pointer -- flat loads, which count against both wait counters and break the MFMA loop's pipelining -- and the
// uniformity of every scalar.  The price is a few loop-invariant registers spilled across the phase boundaries.)
// What the phases hand to each other inside a workgroup: the descent ends by requesting its leaf's parent hidden state
// (pool[entry][tree], one 16-B load per lane) into registers and leaves the action in act_s; the inference writes those
// registers into its row of the image only after it has started its weight stream (the load's latency hides under it),
// and leaves the head logits in the image, where the tree's wave turns them into reward / value / policy logits in its
// registers.  The pointers handed to the shared bodies are biased by -row0 so that their indexing by the global tree
// number lands in these arrays.
struct SearchLds {
  uint16_t* image;  // [16][row_stride]
  float4* prec_s;   // [16][S+1]  the last descent's records, per tree
  int32_t* path_s;  // [16][S+1]  the last descent's path, per tree
  int32_t* nextact_s;  // [16][64] or null: TreeLocal::nextact
  float* lds_q;     // [16][S]    q cache, per tree, for the whole search
  int32_t* act_s;   // [16]
  uint64_t* exp_s;  // [32] hz_exp2f_tab
  float* ptab;      // TreeLocal::ptab where the workgroup's LDS has room for it (SearchArgs::ptab), else null
};

__device__ __forceinline__ TraverseOut search_traverse_out(const hz_mlp_header_t& H, const SearchArgs& a, const SearchLds& L,
                                                           int row0) {
  TraverseOut to;
  to.ix = a.ix; to.iy = a.iy; to.la = L.act_s - row0;
  to.pool = nullptr; to.net_in = nullptr; to.row_bytes = 0; to.net_in_stride_bytes = 0; to.onehot_cols = 0; to.dtype = 0;
  to.tree0 = 0;
  return to;
}

template <class EL>
__device__ __forceinline__ RowFrag search_request_row(const TreeView& tv, const hz_mlp_header_t& H, const SearchArgs& a,
                                                      int entry, int tree, int lane) {
  RowFrag f;
  constexpr int EB = EL::split ? 4 : 2;   // bytes per pool element (the fp16-pair build keeps its pool in fp32)
  const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.pool) + ((size_t)entry * tv.N + tree) * (size_t)H.hidden * EB);
#pragma unroll
  for (int u = 0; u < 2; ++u) f.v[u] = (lane + 64 * u) * (16 / EB) < H.hidden ? src[lane + 64 * u] : make_uint4(0u, 0u, 0u, 0u);
  return f;
}

template <class EL>
__device__ __forceinline__ RowFrag search_first_descent(const TreeView& tv, const hz_mlp_header_t& H, const SearchArgs& a,
                                                        const SearchLds& L, int row0, int tree, int lane, TreeLocal& tl,
                                                        float4& root_row) {
  const TraverseOut to = search_traverse_out(H, a, L, row0);
  int entry;
  tl.root_visit = tv.root_visit[tree];
  tl.root_vsum = tv.root_vsum[tree];
  tl.publish = a.sims == 1;
  root_row = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < tv.A) root_row = tv.rec[(size_t)tree * tv.S * tv.A + lane];  // from now on kept in registers, patched per backup
  traverse_body<true>(tv, tree, lane, 0, tv.mm_min[tree], tv.mm_max[tree], tl.root_visit, to, true, root_row, &entry, &tl);
  return search_request_row<EL>(tv, H, a, entry, tree, lane);
}

template <class EL, bool REPLAY>
__device__ __forceinline__ RowFrag search_backup_descent(const TreeView& tv, const hz_mlp_header_t& H, const SearchArgs& a,
                                                         const SearchLds& L, int row0, int tree, int lane, int srow, int sim,
                                                         bool more, TreeLocal& tl, float4& root_row) {
  const TraverseOut to = search_traverse_out(H, a, L, row0);
  NetOut no;
  no.rewards = nullptr; no.values = nullptr; no.logits = nullptr;
  no.reward_logits = nullptr; no.value_logits = nullptr; no.policy_logits = nullptr;
  no.reward_stride = 0; no.value_stride = 0; no.policy_stride = 0;
  no.support_size = 0; no.support_min = 0; no.dtype = 0; no.out_rewards = nullptr; no.out_values = nullptr;
  TP_ON(more ? 1 : 0);  // (diagnostic builds: the stamps of the last simulation that has a descent stay)
  TP(14);
  // the leaf's heads, straight from the row image the inference left behind into registers (the arithmetic of the
  // stand-alone kernel's final stage): lanes 0-31 turn the reward logits into a scalar, lanes 32-63 the value logits;
  // lane a takes policy logit a
  {
    const uint16_t* row = L.image + (size_t)srow * H.row_stride;
    const float x = row32_support_to_scalar(row, (lane >> 5) ? H.off_value : H.off_reward, (lane >> 5) ? H.off_value2 : H.off_reward2,
                                            H.logit_split, H.support_size, H.support_min, lane & 31);
    float pl = 0.0f;
    if (lane < tv.A) pl = row_policy_logit(row, H.off_policy, lane);
    tl.leaf_reward = hz_readlane_f(x, 0);
    tl.leaf_value = hz_readlane_f(x, 32);
    tl.leaf_logit = pl;
  }
  float mn, mx;
  int rv, a0;
  float4 first;
  TP(0);
  tl.publish = !more;  // the last backup's root sums and min / max are the ones the read-outs see
  backprop_body<false, true>(tv, tree, lane, srow, L.lds_q, sim + 1, no, mn, mx, rv, first, a0, &tl);
  RowFrag f;
  f.v[0] = f.v[1] = make_uint4(0u, 0u, 0u, 0u);
  if (lane == a0) root_row = first;  // the one record of the root's row this backup changed
  if (more) {
    // (the fence between this backup's stores and the descent's loads sits inside traverse_body, behind the root level)
    hz_tree_descent_prio();
    int entry;
    TP(5);
    tl.publish = sim + 2 == a.sims;  // the last descent
    traverse_body<true, REPLAY>(tv, tree, lane, sim + 1, mn, mx, rv, to, true, root_row, &entry, &tl);
    TP(13);
    f = search_request_row<EL>(tv, H, a, entry, tree, lane);
  }
  return f;
}

template <class EL, int RT>
__device__ __forceinline__ void search_inference(const hz_mlp_header_t& H, const SearchArgs& a, const SearchLds& L, int sim,
                                                 int n_rows, int row0, const RowFrag* rows, const int* jv) {
  uint16_t* hidden_out = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.pool) + (size_t)(sim + 1) * a.plane_stride * (EL::split ? 4 : 2));
  mlp_body<EL, RT, 16, 2, STAGE_REGS, false, RT == 1>(H, a.jobs, a.wstream, a.bias, a.act_tab, a.pool, a.row_stride, nullptr, a.plane_stride,
                                 L.act_s - row0, hidden_out, nullptr, nullptr, nullptr, n_rows, L.image, row0, rows, jv);
}

// Diagnostic build only (-DHZ_SEARCH_PROFILE, tools/search_profile.py): per-phase s_memtime sums of workgroup 100.
#ifdef HZ_SEARCH_PROFILE
__device__ unsigned long long hz_search_prof[16 * 4];
extern "C" int hz_search_profile_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_search_prof), sizeof(hz_search_prof));
}
#define SP_NOW() __builtin_amdgcn_s_memtime()
#else
#define SP_NOW() 0ull
#endif

// RT = 16-row tiles per workgroup: 1 = one tree per wave (a workgroup per CU covers 4096 trees on 256 CUs); 2 = two trees
// per wave, taken one after the other in the tree phases, and 32 rows per weight fragment in the inference -- for more
// trees than 16 x #CUs, where the workgroups would otherwise queue and stream the weights once per 16 rows.
// (amdgpu_num_vgpr: the compiler's registers end below the weight ring of the hand-scheduled k-loop, hz_mlp_dev.h)
template <class EL, int RT, bool RP>  // RP: the descent along predicted lines compiled in (hz_tree_replay_dev.h; RT == 1 only)
__device__ __forceinline__ void search_kernel_body(const TreeView& tv, const hz_mlp_header_t& H, const SearchArgs& a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = 16 * RT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * MT;
  SearchLds L;
  L.image = lds;
  L.exp_s = reinterpret_cast<uint64_t*>(lds + (size_t)MT * H.row_stride);  // (row_stride % 8 == 0: 16-B aligned)
  L.prec_s = reinterpret_cast<float4*>(L.exp_s + 32);
  L.path_s = reinterpret_cast<int32_t*>(L.prec_s + MT * (tv.S + 1));
  L.lds_q = reinterpret_cast<float*>(L.path_s + MT * (tv.S + 1));
  L.act_s = reinterpret_cast<int32_t*>(L.lds_q + MT * tv.S);
  L.ptab = (a.ptab && tv.S < 64) ? reinterpret_cast<float*>(L.act_s + MT + 2) + 128 : nullptr;  // (behind the half kernels' tab_s)
  L.nextact_s = (RT == 1 && RP && a.nextact && L.ptab != nullptr) ? reinterpret_cast<int32_t*>(L.ptab + hz_ptab_words(tv.S)) : nullptr;
  if (L.nextact_s != nullptr) L.nextact_s[threadIdx.x] = 0;
  if (threadIdx.x < 32) L.exp_s[threadIdx.x] = hz_exp2f_tab[threadIdx.x];
  if (L.ptab != nullptr) hz_ptab_fill(L.ptab, tv.pbc_tab, tv.S, (int)threadIdx.x, 1024);
  TreeLocal tl[RT];
  bool mine[RT];
  RowFrag rows[RT];
  float4 root_row[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) {
    const int srow = 16 * s + wave;
    mine[s] = row0 + srow < tv.N;
    tl[s].exp_tab = L.exp_s;
    tl[s].ptab = L.ptab;
    tl[s].lq = L.lds_q + srow * tv.S;
    tl[s].pbc_reg = (tv.S < 64 && lane <= tv.S) ? tv.pbc_tab[lane] : 0.0f;
    tl[s].sqrt_reg = sqrtf((float)lane + 1.0f);
    tl[s].path = L.path_s + srow * (tv.S + 1);
    tl[s].nextact = L.nextact_s != nullptr ? L.nextact_s + srow * 64 : nullptr;
    tl[s].prec = L.prec_s + srow * (tv.S + 1);
    tl[s].root_vsum = 0.0f; tl[s].root_visit = 0; tl[s].path_len = 0; tl[s].deep = false; tl[s].publish = false;
    rows[s].v[0] = rows[s].v[1] = make_uint4(0u, 0u, 0u, 0u);
    root_row[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  unsigned long long p_tree = 0, p_wait1 = 0, p_mlp = 0, p_wait2 = 0;
  (void)p_tree; (void)p_wait1; (void)p_mlp; (void)p_wait2;
  unsigned long long t0 = SP_NOW();
#pragma unroll
  for (int s = 0; s < RT; ++s)
    if (mine[s]) rows[s] = search_first_descent<EL>(tv, H, a, L, row0, row0 + 16 * s + wave, lane, tl[s], root_row[s]);
  // this wave's job entries of the inference: the same for every simulation, loaded once (two registers hold 16 jobs; the in-turn
  // 32-row kernel has none to spare and reloads them per inference)
  int jvc[4] = {0, 0, 0, 0};
  if (RT == 1) {
    jvc[0] = hz_mlp_job_entries(a.jobs, H.n_jobs, 16, wave, lane, 0);
    jvc[1] = hz_mlp_job_entries(a.jobs, H.n_jobs, 16, wave, lane, 1);
  }
  for (int sim = 0; sim < a.sims; ++sim) {
    unsigned long long t1 = SP_NOW();
    // (no barrier here: the inference's own barrier after staging orders the waves' rows and actions)
    unsigned long long t2 = SP_NOW();
    search_inference<EL, RT>(H, a, L, sim, tv.N, row0, rows, RT == 1 ? jvc : nullptr);
    unsigned long long t3 = SP_NOW();
    __syncthreads();  // leaf outputs visible; the row image is free again
    unsigned long long t4 = SP_NOW();
    hz_tree_phase_prio();
    int tid_t = threadIdx.x;  // (opaque: thread-derived values of the tree phase are recomputed per simulation, not spilled)
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63;
    // (in-turn kernel: the wave number too -- kept across the inference, the two trees' global indices were the kernel's only
    // values in scratch: copied into vector registers for want of scalar ones, spilled, reloaded for the last descent's publish)
    const int wave_t = RT == 1 ? wave : __builtin_amdgcn_readfirstlane(tid_t >> 6);
#pragma unroll
    for (int s = 0; s < RT; ++s)
      if (mine[s])
        rows[s] = search_backup_descent<EL, RT == 1 && RP>(tv, H, a, L, row0, row0 + 16 * s + wave_t, lane_t, 16 * s + wave_t, sim, sim + 1 < a.sims,
                                        tl[s], root_row[s]);
    p_tree += t1 - t0; p_wait1 += t2 - t1; p_mlp += t3 - t2; p_wait2 += t4 - t3;
    t0 = t4;
  }
#ifdef HZ_SEARCH_PROFILE
  if (blockIdx.x == 100 / RT && lane == 0) {
    unsigned long long* o = hz_search_prof + wave * 4;
    o[0] = p_tree + (SP_NOW() - t0); o[1] = p_wait1; o[2] = p_mlp; o[3] = p_wait2;
  }
#endif
}

// (RP = false: the same kernel without the descent along predicted lines -- its mere presence costs the plain walk 0.6 %, so a
// caller whose trees stay shallow asks for this one: hz_search_set_predicted_lines)
template <class EL, bool RP>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS))) void k_search(TreeView tv, hz_mlp_header_t H, SearchArgs a) {
  search_kernel_body<EL, 1, RP>(tv, H, a);
}
// the fp16-pair build (include/hz_mlp.h, HZ_F16X2; fp32 pool): one tree per wave, the inference's k-loop compiler-scheduled -- no
// weight ring in fixed registers, so no register budget of its own
template <bool RP>
__global__ __launch_bounds__(1024, 1) void k_search_pairs(TreeView tv, hz_mlp_header_t H, SearchArgs a) {
  search_kernel_body<ElF16x2, 1, RP>(tv, H, a);
}
// the two trees of a wave one after the other: its register pressure peaks above the others', and amdgpu_num_vgpr is a budget the
// allocator was seen to overdraw by four registers -- into the ring (tools/scan_ring_registers.py) -- so this one gets a lower one.
// It keeps workgroup barriers at the layer boundaries (mlp_body<.., BW = false>): with the arrival counters compiled in, the two
// trees' state that lives across the inference no longer fits -- 57 vector registers spilled, 34 scratch accesses spread over
// both phases, at any budget from 88 to 96 (r04, measured on the ISA; a scratch reload drains the weight ring)
template <class EL>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS - 8))) void k_search_turn(TreeView tv, hz_mlp_header_t H, SearchArgs a) {
  search_kernel_body<EL, 2, false>(tv, H, a);
}

// Two trees per tree-owning wave, side by side in its two 32-lane halves (hz_tree_half_dev.h; A <= 32, hidden <= 512): the tree
// phase of two trees costs one instruction stream.  TW = tree-owning waves: 16 -> 32 trees per workgroup (two 16-row tiles in
// the inference); 8 -> 16 trees per workgroup, waves 8-15 sit the tree phase out -- the tree phase is bound by instruction
// issue (the youngest of a SIMD's four waves takes twice as long as the oldest through the same work), so two streams per SIMD
// instead of four take about half the time.
template <class EL, int TW, bool RP>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS))) void k_search_half(TreeView tv, hz_mlp_header_t H, SearchArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = 2 * TW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * MT;
  SearchLds L;
  L.image = lds;
  L.exp_s = reinterpret_cast<uint64_t*>(lds + (size_t)MT * H.row_stride);
  L.prec_s = reinterpret_cast<float4*>(L.exp_s + 32);
  L.path_s = reinterpret_cast<int32_t*>(L.prec_s + MT * (tv.S + 1));
  L.lds_q = reinterpret_cast<float*>(L.path_s + MT * (tv.S + 1));
  L.act_s = reinterpret_cast<int32_t*>(L.lds_q + MT * tv.S);
  float* tab_s = reinterpret_cast<float*>(L.act_s + MT);  // the descent's tables: pb_c's log factor and sqrt(n + 1), n < 64
  L.ptab = (a.ptab && tv.S < 64) ? reinterpret_cast<float*>(L.act_s + MT + 2) + 128 : nullptr;
  if (L.ptab != nullptr) hz_ptab_fill(L.ptab, tv.pbc_tab, tv.S, (int)threadIdx.x, 1024);
  // [MT][S] 16-bit words behind the table of exploration factors: the nodes' last selections (HalfTree::nextact), all "never passed"
  uint16_t* nextact16_s = (TW == 16 && RP && a.nextact && L.ptab != nullptr) ? reinterpret_cast<uint16_t*>(L.ptab + hz_ptab_words(tv.S)) : nullptr;
  if (nextact16_s != nullptr)
    for (int i = (int)threadIdx.x; i < MT * tv.S; i += 1024) nextact16_s[i] = 0;
  if (threadIdx.x < 32) L.exp_s[threadIdx.x] = hz_exp2f_tab[threadIdx.x];
  if (threadIdx.x < 64) {
    tab_s[threadIdx.x] = (tv.S < 64 && (int)threadIdx.x <= tv.S) ? tv.pbc_tab[threadIdx.x] : 0.0f;
    tab_s[64 + threadIdx.x] = sqrtf((float)threadIdx.x + 1.0f);
  }
  __syncthreads();
  HalfLane q;
  HalfTree t;
  float4 root_row = make_float4(0.f, 0.f, 0.f, 0.f);
  RowFrag rows;
  rows.v[0] = rows.v[1] = make_uint4(0u, 0u, 0u, 0u);
  float mn = 0.0f, mx = 0.0f;
  // (everything lane-derived is rebuilt from an opaque lane index wherever a phase starts: see k_search)
#define HZ_HALF_SETUP(LANE)                                                              \
  {                                                                                       \
    q.l = (LANE) & 31; q.h = (LANE) >> 5; q.hbase = 32 * q.h;                            \
    const int srow = TW * q.h + (wave < TW ? wave : 0);                                   \
    t.tree = row0 + srow;                                                                 \
    t.mine = t.tree < tv.N && wave < TW;                                                  \
    t.path = L.path_s + srow * (tv.S + 1);                                                \
    t.prec = L.prec_s + srow * (tv.S + 1);                                                \
    t.lq = L.lds_q + srow * tv.S;                                                         \
    t.nextact = nextact16_s != nullptr ? nextact16_s + srow * tv.S : nullptr;             \
  }
  HZ_HALF_SETUP(lane)
  t.root_vsum = 0.0f; t.root_visit = 0; t.path_len = 1; t.deep = false;
  t.leaf_reward = t.leaf_value = t.leaf_logit = 0.0f;
  const bool any_mine = wave < TW && row0 + wave < tv.N;  // (rows are filled in order: the lower half's tree exists if any does)
  if (any_mine) {
    if (t.mine) {
      t.root_visit = tv.root_visit[t.tree];
      t.root_vsum = tv.root_vsum[t.tree];
      mn = tv.mm_min[t.tree];
      mx = tv.mm_max[t.tree];
      if (q.l < tv.A) root_row = tv.rec[(size_t)t.tree * tv.S * tv.A + q.l];
    }
    const int entry = traverse_half(tv, q, t, 0, mn, mx, root_row, tab_s, L.ptab, L.act_s + TW * q.h + wave, a.ix, a.iy,
                                    a.sims == 1);
    if (t.mine) {
      const uint4* src = reinterpret_cast<const uint4*>(a.pool + ((size_t)entry * tv.N + t.tree) * (size_t)H.hidden);
#pragma unroll
      for (int u = 0; u < 2; ++u) rows.v[u] = (q.l + 32 * u) * 8 < H.hidden ? src[q.l + 32 * u] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  unsigned long long p_tree = 0, p_mlp = 0, p_wait2 = 0;
  (void)p_tree; (void)p_mlp; (void)p_wait2;
  unsigned long long t0 = SP_NOW();
  int jvc[4] = {hz_mlp_job_entries(a.jobs, H.n_jobs, 16, wave, lane, 0), hz_mlp_job_entries(a.jobs, H.n_jobs, 16, wave, lane, 1), 0, 0};
  for (int sim = 0; sim < a.sims; ++sim) {
    const unsigned long long t2 = SP_NOW();
    mlp_body<EL, TW / 8, 16, 2, STAGE_REGS_HALF, false>(H, a.jobs, a.wstream, a.bias, a.act_tab, a.pool, a.row_stride, nullptr,
                                               a.plane_stride, L.act_s - row0, a.pool + (size_t)(sim + 1) * a.plane_stride,
                                               nullptr, nullptr, nullptr, tv.N, L.image, row0, &rows, jvc);
    const unsigned long long t3 = SP_NOW();
    __syncthreads();  // leaf outputs visible; the row image is free again
    const unsigned long long t4 = SP_NOW();
    hz_tree_phase_prio();
    p_tree += t2 - t0; p_mlp += t3 - t2; p_wait2 += t4 - t3;
    t0 = t4;
    if (!any_mine) continue;
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    HZ_HALF_SETUP(lane_t)
    const bool more = sim + 1 < a.sims;
    {  // the leaf's heads, from the row image into registers: each half's 32 lanes turn first its reward logits, then its
      // value logits into scalars (uniform within the half); lane l takes policy logit l
      const uint16_t* row = L.image + (size_t)(TW * q.h + wave) * H.row_stride;
      t.leaf_reward = row32_support_to_scalar(row, H.off_reward, H.off_reward2, H.logit_split, H.support_size, H.support_min, q.l);
      t.leaf_value = row32_support_to_scalar(row, H.off_value, H.off_value2, H.logit_split, H.support_size, H.support_min, q.l);
      t.leaf_logit = q.l < tv.A ? row_policy_logit(row, H.off_policy, q.l) : 0.0f;
    }
    int rv, a0;
    float4 first;
    backprop_half(tv, q, t, sim + 1, L.exp_s, !more, mn, mx, rv, first, a0);
    if (q.l == a0) root_row = first;
    rows.v[0] = rows.v[1] = make_uint4(0u, 0u, 0u, 0u);
    if (more) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      hz_tree_descent_prio();
      int entry;
      if (TW == 16 && RP && __ballot(t.deep) != 0)
        entry = traverse_half<2>(tv, q, t, sim + 1, mn, mx, root_row, tab_s, L.ptab, L.act_s + TW * q.h + wave, a.ix, a.iy, sim + 2 == a.sims);
      else
        entry = traverse_half<(TW == 16 && RP) ? 1 : 0>(tv, q, t, sim + 1, mn, mx, root_row, tab_s, L.ptab, L.act_s + TW * q.h + wave, a.ix, a.iy,
                                                sim + 2 == a.sims);
      if (t.mine) {
        const uint4* src = reinterpret_cast<const uint4*>(a.pool + ((size_t)entry * tv.N + t.tree) * (size_t)H.hidden);
#pragma unroll
        for (int u = 0; u < 2; ++u) rows.v[u] = (q.l + 32 * u) * 8 < H.hidden ? src[q.l + 32 * u] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
#undef HZ_HALF_SETUP
#ifdef HZ_SEARCH_PROFILE
  if (blockIdx.x == 800 / MT && lane == 0) {
    unsigned long long* o = hz_search_prof + wave * 4;
    o[0] = p_tree + (SP_NOW() - t0); o[1] = 0; o[2] = p_mlp; o[3] = p_wait2;
  }
#endif
}

extern "C" int hz_search_set_predicted_lines(hz_tree_t* t, int on) {
  HZ_REQUIRE(t != nullptr, "hz_search_set_predicted_lines: tree is NULL");
  t->predicted_lines = on ? 1 : 0;
  return 0;
}

// What this library remembers per DEVICE (ordinal of the tree handle, not the calling thread's current device): the compute-
// unit count and, per kernel variant, the dynamic-LDS limit already raised with hipFuncSetAttribute (a per-device attribute).
struct SearchDevice {
  int n_cu;
  size_t configured[16];
};
static SearchDevice g_search_dev[64];

// rows per workgroup: 0 = choose by the tree count (two trees per wave once one per wave would need more workgroups than the
// device has compute units), 16 / 32 = force, -32 = 32 with the two trees of a wave one after the other (tests, tools)
extern "C" int hz_search_poll_giveups(unsigned int* count) {
  HZ_REQUIRE(count != nullptr, "hz_search_poll_giveups: NULL argument");
  HZ_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(hz_poll_giveups_dev), sizeof(unsigned int)));
  return 0;
}

// the same count, copied on `stream` into pinned host memory: no device-wide synchronisation (hz_mlp_poll_giveups_async)
extern "C" int hz_search_poll_giveups_async(unsigned int* host_pinned, void* stream) {
  HZ_REQUIRE(host_pinned != nullptr, "hz_search_poll_giveups_async: NULL argument");
  HZ_HIP(hipMemcpyFromSymbolAsync(host_pinned, HIP_SYMBOL(hz_poll_giveups_dev), sizeof(unsigned int), 0, hipMemcpyDeviceToHost,
                                  (hipStream_t)stream));
  return 0;
}

extern "C" int hz_search_run(hz_tree_t* t, int num_simulations, const hz_mlp_header_t* H, const hz_mlp_job_t* jobs,
                             const void* wstream, const float* biases, const float* action_table, void* pool,
                             int64_t plane_stride, int64_t row_stride, int32_t* ix, int32_t* iy, int32_t* la,
                             float* rewards, float* values, float* policy, int rows_per_workgroup, void* stream) {
  HZ_REQUIRE(t && H && jobs && wstream && biases && action_table && pool && ix && iy && la && rewards && values && policy,
             "hz_search_run: NULL argument");
  HZ_REQUIRE(t->params_set, "hz_search_run: call hz_tree_set_params first");
  HZ_REQUIRE(num_simulations >= 1 && num_simulations < t->S,
             "hz_search_run: num_simulations=%d outside [1, tree capacity %d)", num_simulations, t->S);
  HZ_REQUIRE(t->next_entry == 1, "hz_search_run: the tree must be freshly prepared (hz_tree_prepare)");
  HZ_REQUIRE(rows_per_workgroup == 0 || rows_per_workgroup == 16 || rows_per_workgroup == 32 || rows_per_workgroup == -32 ||
                 rows_per_workgroup == -16,
             "hz_search_run: rows_per_workgroup %d (0 = auto, 16, 32; -16: 16 with two trees side by side on 8 waves; -32: 32 with the two trees of a wave one after the other)",
             rows_per_workgroup);
  HZ_REQUIRE(H->dtype == HZ_BF16 || H->dtype == HZ_F16 || H->dtype == HZ_F16X2,
             "hz_search_run: header dtype must be HZ_BF16, HZ_F16 or HZ_F16X2 (got %d)", H->dtype);
  HZ_REQUIRE(H->dtype != HZ_F16X2 || (H->lo_plane > 0 && H->lo_plane % 8 == 0 && 2 * H->lo_plane <= H->row_stride &&
                                      (rows_per_workgroup == 0 || rows_per_workgroup == 16)),
             "hz_search_run: the fp16-pair build keeps 16 trees per workgroup (one per wave); its image's lo plane lies lo_plane columns behind the hi plane");
  HZ_REQUIRE(H->num_waves == 16 && H->tiles_per_wave == 2, "hz_search_run: the MLP must be laid out for 16 waves x 2 tiles");
  HZ_REQUIRE(H->num_actions == t->A, "hz_search_run: the MLP has %d actions, the tree %d", H->num_actions, t->A);
  HZ_REQUIRE(H->n_jobs > 0 && H->n_jobs <= 32 && H->support_size > 0 && H->support_size <= 256 &&
                 H->off_reward % 8 == 0 && H->off_value % 8 == 0 && H->row_stride % 8 == 0 && H->hidden % 8 == 0 &&
                 H->state_off % 8 == 0 && H->hidden_off % 8 == 0 && H->action_table_stride % 4 == 0,
             "hz_search_run: malformed MLP header");
  HZ_REQUIRE(row_stride == H->hidden && plane_stride == (int64_t)t->N * row_stride,
             "hz_search_run: the pool must be contiguous [planes][N][hidden] (row_stride %lld, plane_stride %lld)",
             (long long)row_stride, (long long)plane_stride);
  HZ_REQUIRE(((uintptr_t)pool % 16) == 0 && ((uintptr_t)wstream % 16) == 0 && ((uintptr_t)biases % 16) == 0 &&
                 ((uintptr_t)action_table % 16) == 0,
             "hz_search_run: pointers must be 16-B aligned");
  for (int w = 0; w < 16; ++w)
    HZ_REQUIRE(H->wave_stream_off[w] % 8 == 0, "hz_search_run: weight streams must start on 16-B boundaries");
  HZ_REQUIRE(H->kstep_stride >= 512 * H->tiles_per_wave && H->kstep_stride % 8 == 0,
             "hz_search_run: kstep_stride must be a multiple of 8 and at least one k-step (512 * tiles_per_wave)");
  HZ_REQUIRE(H->in_width > 0 && H->in_width % 8 == 0, "hz_search_run: in_width must be a positive multiple of 8");
  HZ_REQUIRE(H->in_width == H->hidden && H->hidden <= 1024, "hz_search_run: the recurrent inference maps a hidden state (<= 1024 wide) to a hidden state");
  int rows_wg = rows_per_workgroup < 0 ? -rows_per_workgroup : rows_per_workgroup;
  HZ_REQUIRE(t->device >= 0 && t->device < 64, "hz_search_run: device ordinal %d outside [0, 64)", t->device);
  SearchDevice& dev = g_search_dev[t->device];
  if (dev.n_cu == 0) HZ_HIP(hipDeviceGetAttribute(&dev.n_cu, hipDeviceAttributeMultiprocessorCount, t->device));
  if (rows_wg == 0) rows_wg = ((t->N + 15) / 16 > dev.n_cu && H->dtype != HZ_F16X2) ? 32 : 16;
  auto lds_for = [&](int mt) {
    return (size_t)mt * H->row_stride * sizeof(uint16_t) + (size_t)mt * (t->S + 1) * (16 + 4) + (size_t)mt * t->S * sizeof(float) +
           (size_t)(mt + 2) * sizeof(float) + 32 * 8 + 128 * sizeof(float) + 128;  // (+128: slack behind the last array)
  };
  if (rows_wg == 32 && lds_for(32) > 160 * 1024 && rows_per_workgroup == 0) rows_wg = 16;
  size_t lds_bytes = lds_for(rows_wg);
  HZ_REQUIRE(lds_bytes <= 160 * 1024, "hz_search_run: %zu B of LDS per workgroup exceed 160 KiB", lds_bytes);
  HZ_REQUIRE(H->n_jobs <= 16, "hz_search_run: %d passes in the job table (the persistent kernels keep 16 in registers)", H->n_jobs);
  HZ_REQUIRE(t->S < 64 && H->hidden <= 512, "hz_search_run: the persistent kernels are written for < 64 simulations and hidden <= 512 "
                                            "(got %d, %d): run the launch-per-phase search", t->S, H->hidden);
  // the table of exploration factors (TreeLocal::ptab, triangular: 5.3 KB at S = 50) where the workgroup's LDS has room for it
  const size_t ptab_bytes = (size_t)(t->S + 1) * (t->S + 2) / 2 * sizeof(float);
  const bool use_ptab = t->S < 64 && lds_bytes + ptab_bytes <= 160 * 1024;
  if (use_ptab) lds_bytes += ptab_bytes;
  // the nodes' last selections for the descent along predicted lines (hz_tree_replay_dev.h): [16][64] words in the 16-tree
  // kernel, [32][S] half-words in the side-by-side 32-tree kernel (which has 3 KB left), where they fit
  const bool halves_early = ((rows_wg == 32 && rows_per_workgroup != -32) || rows_per_workgroup == -16) && t->A <= 32 && H->hidden <= 512;
  const size_t nextact_bytes = rows_wg == 16 ? (size_t)16 * 64 * sizeof(int32_t) : (((size_t)32 * t->S * sizeof(uint16_t) + 15) & ~(size_t)15);
  const bool use_nextact = t->predicted_lines != 0 && use_ptab && ((rows_wg == 16 && rows_per_workgroup != -16) || (rows_wg == 32 && halves_early)) &&
                           lds_bytes + nextact_bytes <= 160 * 1024;
  if (use_nextact) lds_bytes += nextact_bytes;
  // two trees per tree-owning wave, side by side: the default with 32 trees per workgroup; with 16 only on request (-16) --
  // measured at 4096 envs: +1.4 % moves/s with random-init nets (mean path 2.2 edges), -12 % with a sharp policy (5.3 edges: a
  // half that has reached its leaf idles through the other's remaining levels)
  const bool halves = ((rows_wg == 32 && rows_per_workgroup != -32) || rows_per_workgroup == -16) && t->A <= 32 && H->hidden <= 512;
  SearchArgs a;
  a.jobs = jobs; a.wstream = (const uint16_t*)wstream; a.bias = biases; a.act_tab = action_table;
  a.pool = (uint16_t*)pool; a.plane_stride = plane_stride; a.row_stride = row_stride;
  a.ix = ix; a.iy = iy; a.la = la; a.rew = rewards; a.val = values; a.pol = policy; a.sims = num_simulations;
  a.ptab = use_ptab ? 1 : 0;
  a.nextact = use_nextact ? 1 : 0;
#define HZ_SEARCH_LAUNCH(VARIANT, KERNEL, GRID)                                                                       \
  do {                                                                                                                \
    if (lds_bytes > dev.configured[VARIANT]) {                                                                        \
      HZ_HIP(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));    \
      dev.configured[VARIANT] = lds_bytes;                                                                            \
    }                                                                                                                 \
    hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(1024), lds_bytes, (hipStream_t)stream, view(t), *H, a);               \
  } while (0)
#define HZ_SEARCH_LAUNCH_EL(V0, EL)                                                                  \
  do {                                                                                               \
    if (halves && rows_wg == 32 && use_nextact) HZ_SEARCH_LAUNCH(V0 + 4, (k_search_half<EL, 16, true>), (t->N + 31) / 32); \
    else if (halves && rows_wg == 32) HZ_SEARCH_LAUNCH(V0 + 2, (k_search_half<EL, 16, false>), (t->N + 31) / 32); \
    else if (halves) HZ_SEARCH_LAUNCH(V0 + 3, (k_search_half<EL, 8, false>), (t->N + 15) / 16);      \
    else if (rows_wg == 32) HZ_SEARCH_LAUNCH(V0 + 1, (k_search_turn<EL>), (t->N + 31) / 32);         \
    else if (use_nextact) HZ_SEARCH_LAUNCH(V0 + 5, (k_search<EL, true>), (t->N + 15) / 16);          \
    else HZ_SEARCH_LAUNCH(V0, (k_search<EL, false>), (t->N + 15) / 16);                              \
  } while (0)
  int cur = -1;
  HZ_HIP(hipGetDevice(&cur));
  HZ_REQUIRE(cur == t->device, "hz_search_run: the calling thread's current device is %d, the tree lives on %d", cur, t->device);
  if (H->dtype == HZ_F16X2) {
    if (use_nextact) HZ_SEARCH_LAUNCH(7, (k_search_pairs<true>), (t->N + 15) / 16);
    else HZ_SEARCH_LAUNCH(6, (k_search_pairs<false>), (t->N + 15) / 16);
  } else if (H->dtype == HZ_F16) HZ_SEARCH_LAUNCH_EL(8, ElF16);
  else HZ_SEARCH_LAUNCH_EL(0, ElBf16);
#undef HZ_SEARCH_LAUNCH_EL
#undef HZ_SEARCH_LAUNCH
  HZ_HIP(hipGetLastError());
  t->next_entry = num_simulations + 1;
  return 0;
}
