// hz_tree_replay_dev.h -- the descent along the tree's PREDICTED line, sixteen levels at a time (included by hz_tree_dev.h).
//
// The tree walk of the reference (core/ctree/cnode.cpp:407-441) is a chain: a level's selection names the next node.  One wave
// doing one level at a time pays ~2 k cycles per level whatever else the workgroup does, and with a sharp policy -- what training
// produces -- a few lines of the tree are searched ever deeper: a workgroup's inference then waits for the one tree whose path
// is 20 or 30 levels long (tools/level_profile.py).  But a node mostly chooses what it chose the last time it was passed.
// The persistent kernel keeps that choice per node in LDS (TreeLocal::nextact: action and the child entry it leads to; the
// backup enters the entry it expands), which names a predicted line below any node: sixteen of its nodes are found with
// pointer doubling across the lanes (the table's entry n sits in lane n) and evaluated side by side -- four lanes per level, each
// owning C consecutive children -- with exactly the operations of traverse_body, operand for operand and in its order:
//   * the ordered sum over a node's visited children runs through the four lanes of the level in child order;
//   * mean_q is the one true recurrence (a level's value enters the next level's): one division per level, level after level;
//   * scores, the arg-max with its epsilon tie list and the tie-break draw are per level.
// The pass is valid down to the first level whose selection differs from the prediction (that level included: its inputs
// were the right ones).  Stores (best_action, path, the record the backup will update, the table) are made for the valid
// levels only; traverse_body goes on from the node that level chose -- with another pass if a line is known below it.
#pragma once

struct ReplayOut {
  int e;           // leaf: the node the leaf action was chosen at; otherwise the node the walk goes on from
  int action;      // the last valid level's selection
  int depth;       // levels done
  int pvc;         // visit count of the edge into `e` (when the walk goes on)
  float parent_q;  // mean_q of the last valid level
  bool leaf;
};

#define HZ_QUAD(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
__device__ __forceinline__ int hz_quad_or(int v) {  // OR over the four lanes of a quad, in all four
  v |= HZ_QUAD(v, 0xB1);  // quad_perm:[1,0,3,2]
  v |= HZ_QUAD(v, 0x4E);  // quad_perm:[2,3,0,1]
  return v;
}
// ... over the G = 4 | 8 lanes of a level's group (8: the two quads of a half row swap through row_half_mirror), in all of them
template <int G>
__device__ __forceinline__ int hz_group_or(int v) {
  v = hz_quad_or(v);
  if (G == 8) v |= HZ_QUAD(v, 0x141);
  return v;
}
template <int G>
__device__ __forceinline__ int hz_group_add(int v) {
  v += HZ_QUAD(v, 0xB1);
  v += HZ_QUAD(v, 0x4E);
  if (G == 8) v += HZ_QUAD(v, 0x141);
  return v;
}
template <int G>
__device__ __forceinline__ float hz_group_max(float v) {
  v = fmaxf(v, __int_as_float(HZ_QUAD(__float_as_int(v), 0xB1)));
  v = fmaxf(v, __int_as_float(HZ_QUAD(__float_as_int(v), 0x4E)));
  if (G == 8) v = fmaxf(v, __int_as_float(HZ_QUAD(__float_as_int(v), 0x141)));
  return v;
}

// What the replay needs of TreeView / TreeLocal, by value.  (Measured at 4096 envs, random-init nets, where replays are rare: with
// the two structs passed by reference the mere presence of the inlined code cost the ordinary descent 1.4 % -- 25 more
// scalar-register spills in the kernel; with this struct 0.6 %; as a real function, called: 4.4 %, and a third of the gain lost.
// Hence two builds of the search kernels, with and without all of this: hz_search.hip, hz_search_set_predicted_lines.)
struct ReplayIn {
  int A, S, tree, sim;
  float mn, mx, discount, delta_floor;
  unsigned long long seed;
  unsigned int id_base;
  const float4* rec;        // the tree's records
  int8_t* best_action;      // the tree's [S]
  int32_t* nextact;         // LDS [64]: HZ_NEXTACT(child entry, action) of a node's last selection, 0 = never passed
  int32_t* path;            // LDS
  float4* prec;             // LDS
  const float* lq;          // LDS
  const float* ptab;        // LDS
};

// a node's last selection: bit 16 = "has been passed", bits 8-15 = child entry + 1 (0: not expanded then), bits 0-7 = action
#define HZ_NEXTACT(child_e, action) (0x10000 | (((child_e) + 1) << 8) | (action))

// G lanes per level, C children per lane: lane G j + g of the wave owns children g C .. g C + C - 1 of level slot j (A <= G C);
// 64 / G levels per pass
template <int G, int C>
__device__ __forceinline__ ReplayOut traverse_replay(const ReplayIn in, int start, int depth0, float mq_in, int pvc_in) {
  const int lane = (int)(threadIdx.x & 63);
  const int A = in.A, S = in.S, tree = in.tree, sim = in.sim;
  constexpr int LG = G == 4 ? 2 : 3, SLOTS = 64 / G;
  const int g = lane & (G - 1), j = lane >> LG;
  const float discount = in.discount, mn = in.mn, mx = in.mx;
  const float delta = mx - mn;
  const float dn = delta < in.delta_floor ? in.delta_floor : delta;
  const float4* rec = in.rec;
  ReplayOut out;
  TPP_DECL;
  {
    // the predicted line below `start`: slot j's node is next^j(start), by pointer doubling (lane 63 = "no node", its own successor)
    const int na_mine = lane < S ? in.nextact[lane] : 0;
    const int c1 = (na_mine >> 8) & 0xff;
    const int J1 = ((na_mine & 0x10000) && c1) ? c1 - 1 : 63;
    const int J2 = __builtin_amdgcn_ds_bpermute(4 * J1, J1);
    const int J4 = __builtin_amdgcn_ds_bpermute(4 * J2, J2);
    const int J8 = SLOTS > 8 ? __builtin_amdgcn_ds_bpermute(4 * J4, J4) : 63;
    int n = start;
    { const int t = __builtin_amdgcn_ds_bpermute(4 * n, J1); if (j & 1) n = t; }
    { const int t = __builtin_amdgcn_ds_bpermute(4 * n, J2); if (j & 2) n = t; }
    { const int t = __builtin_amdgcn_ds_bpermute(4 * n, J4); if (j & 4) n = t; }
    if (SLOTS > 8) { const int t = __builtin_amdgcn_ds_bpermute(4 * n, J8); if (j & 8) n = t; }
    const bool lvl = n != 63;
    const int na_n = __builtin_amdgcn_ds_bpermute(4 * n, na_mine);
    const int ap = (lvl && (na_n & 0x10000)) ? (na_n & 255) : -1;  // the node's last selection (none: never passed)
    const int k = depth0 + j;  // this lane's level
    TPP(0);
    float4 R[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const int a = g * C + i;
      R[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lvl && a < A) R[i] = rec[(size_t)n * A + a];
    }
    float q[C];
    int nv = 0, wprev = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const int a = g * C + i;
      const uint32_t w = __float_as_uint(R[i].w);
      const int visit = (int)(w >> 16);
      const int child = (int)(w & 0xffffu) - 1;
      const bool vis = lvl && a < A && visit > 0;
      q[i] = R[i].z + discount * 0.0f;
      if (vis) q[i] = in.lq[child];
      nv += vis ? 1 : 0;
      if (a == ap) wprev = (int)w;
    }
    TPP(1);
    // get_mean_q's sum over the visited children in action order: through the level's four lanes, one after the other
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < G; ++t) {  // (every lane folds its children onto what the lane below holds; lane t keeps the result in round t)
      // the lane below: quad_perm:[0,0,1,2] inside a quad, row_shr:1 inside a group of eight (lane 0 of a group takes none)
      float f = t == 0 ? 0.0f : __int_as_float(G == 4 ? HZ_QUAD(__float_as_int(s), 0x90) : HZ_QUAD(__float_as_int(s), 0x111));
#pragma unroll
      for (int i = 0; i < C; ++i) {
        const uint32_t w = __float_as_uint(R[i].w);
        const float with = f + q[i];
        f = (lvl && g * C + i < A && (w >> 16) > 0) ? with : f;
      }
      s = g == t ? f : s;
    }
    const float total = G == 4 ? __int_as_float(HZ_QUAD(__float_as_int(s), 0xFF))  // quad_perm:[3,3,3,3]: the group's last lane
                               : __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (lane | (G - 1)), __float_as_int(s)));
    nv = hz_group_add<G>(nv);
    wprev = hz_group_or<G>(wprev);  // (visits << 16 | child + 1) of the edge the last descent took from here
    // visit count of the edge INTO this level's node: the level above holds it
    int pvc = __builtin_amdgcn_ds_bpermute(4 * (lane - G), wprev >> 16);
    if (j == 0) pvc = pvc_in;
    if (!lvl) pvc = 0;
    TPP(2);
    // mean_q, level after level (cnode.cpp:228-236 for the root, :414-420 below it)
    float mq = 0.0f;
    {
      const bool rootm = k == 0 && nv > 0;
      const float den = (float)(rootm ? nv : nv + 1);
      const int steps = __popcll((unsigned long long)__ballot(lvl)) >> LG;  // (a line's nodes fill the slots from 0 up)
      float pq = mq_in;
      for (int t = 0; t < steps; ++t) {
        const float cand = (rootm ? total : pq + total) / den;
        if (j == t) mq = cand;
        pq = hz_readlane_f(cand, G * t);
      }
    }
    TPP(3);
    // cucb_score per child, the level's maximum
    float score[C];
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const int a = g * C + i;
      const uint32_t w = __float_as_uint(R[i].w);
      const int visit = (int)(w >> 16);
      float prior = R[i].x;
      if (prior != prior) prior = 0.0f;
      const float pb_c = in.ptab[hz_ptab_index(pvc, lvl ? visit : 0)];
      const float prior_score = pb_c * prior;
      float vs = (visit == 0) ? mq : q[i];
      if (delta > 0.0f) vs = (vs - mn) / dn;
      if (vs < 0.0f) vs = 0.0f;
      if (vs > 1.0f) vs = 1.0f;
      const float sc = prior_score + vs;
      const bool valid = lvl && a < A && (sc == sc) && (sc > HZ_FLOAT_MIN);
      score[i] = valid ? sc : -INFINITY;
      M = fmaxf(M, score[i]);
    }
    M = hz_group_max<G>(M);
    TPP(4);
    // cselect_child: {first arg-max} U {later children within epsilon of the max}, the draw among them
    const float thr = M - 0.000001f;
    int eqb = 0, cb = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const bool valid = score[i] != -INFINITY;
      if (valid && score[i] == M) eqb |= 1 << (g * C + i);
      if (valid && score[i] >= thr) cb |= 1 << (g * C + i);
    }
    eqb = hz_group_or<G>(eqb);
    uint32_t cand = (uint32_t)hz_group_or<G>(cb);
    int action = 0;
    if (eqb != 0) {
      const int first = __ffs(eqb) - 1;
      cand &= ~((1u << first) - 1u);
      const uint32_t cnt = (uint32_t)__popc(cand);
      if (cnt > 1) {
        const uint32_t rnd = hz_tiebreak_rand(in.seed, in.id_base + (uint32_t)tree, (uint32_t)sim, (uint32_t)k);
        uint32_t kk = rnd % cnt;
        while (kk--) cand &= cand - 1;
      }
      action = __ffs(cand) - 1;
    }
    int wsel = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
      if (g * C + i == action) wsel = (int)__float_as_uint(R[i].w);
    wsel = hz_group_or<G>(wsel);
    TPP(5);
    // the pass ends at the first level that chose differently from the prediction (or at the line's last node)
    const uint64_t lb = __ballot(lvl);
    const int last = (64 - __clzll((unsigned long long)lb) - 1) >> LG;  // (slot 0 always holds `start`)
    const uint64_t sb = __ballot(lvl && action != ap);
    const int first_off = sb ? ((__ffsll((unsigned long long)sb) - 1) >> LG) : SLOTS - 1;
    const int mslot = min(first_off, last);
    const bool commit = lvl && j <= mslot;
    if (commit && g == 0) {
      in.best_action[n] = (int8_t)action;
      in.path[k] = (n << 8) | action;
      in.nextact[n] = 0x10000 | ((wsel & 0xffff) << 8) | action;
    }
#pragma unroll
    for (int i = 0; i < C; ++i)
      if (commit && g * C + i == action) in.prec[k] = R[i];
    const int src = G * mslot;
    const uint32_t wv = (uint32_t)hz_readlane_i(wsel, src);
    const int child_e = (int)(wv & 0xffffu) - 1;
    out.action = hz_readlane_i(action, src);
    out.parent_q = hz_readlane_f(mq, src);
    out.depth = depth0 + mslot + 1;
    out.pvc = (int)(wv >> 16);
    out.leaf = child_e < 0 || out.depth >= S;
    out.e = out.leaf ? hz_readlane_i(n, src) : child_e;
    TPP(6);
    TPP(8);
  }
  return out;
}
