// hz_tree_half_dev.h -- two search trees per wavefront, side by side: the tree of lanes 0-31 and the tree of lanes 32-63
// walk through the descent / expand + backup code of hz_tree_dev.h in ONE instruction stream (lane l of a half owns child
// l; needs A <= 32).  Used by the persistent search kernel when a workgroup owns 32 trees (hz_search.hip): the two trees of a
// wave then cost one tree phase instead of two.
//
// Same arithmetic, same order as traverse_body / backprop_body<false, true> (the bodies the goldens pin): what is
// wave-uniform there (entry, depth, action, parent visit count, running sums ...) is uniform per HALF here and lives in
// vector registers; cross-lane steps use ballots cut in two, DPP inside the 16-lane rows, and ds_bpermute for a lookup at
// a per-half index.  A half that has reached its leaf (or owns no tree) idles through the other half's remaining levels
// with every store and state update masked.
#pragma once
#include "hz_tree_dev.h"

struct HalfLane {
  int l;      // lane within the half = child index
  int h;      // 0 | 1
  int hbase;  // 32 * h
};

__device__ __forceinline__ float hh_bcast_f(float x, int idx, const HalfLane& q) {  // x of lane idx of this lane's half
  return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (q.hbase + idx), __float_as_int(x)));
}
__device__ __forceinline__ int hh_bcast_i(int x, int idx, const HalfLane& q) {
  return __builtin_amdgcn_ds_bpermute(4 * (q.hbase + idx), x);
}
__device__ __forceinline__ float hh_lookup_f(float tab, int idx) {  // a 64-entry table held one entry per lane of the wave
  return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * idx, __float_as_int(tab)));
}
__device__ __forceinline__ uint32_t hh_ballot(bool p, const HalfLane& q) {
  const uint64_t b = __ballot(p);
  return q.h ? (uint32_t)(b >> 32) : (uint32_t)b;
}
// max / min over the 32 lanes of each half (callers mask with +-inf; no NaNs): DPP-modified v_max / v_min -- four butterflies
// inside the 16-lane rows, row_bcast:15 into the odd rows -- after which lanes 31 and 63 hold the halves' results
// (hz_common.h::hz_wave_max has the reasons for writing it by hand).  All 64 lanes must be active.
#define HH_HALF_REDUCE(OP)                                                                                            \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"              \
               OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                            \
               OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                \
               OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                     \
               OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"                                                   \
               : "+v"(v))
__device__ __forceinline__ float hh_max(float v, const HalfLane& q) {
  HH_HALF_REDUCE("v_max_f32_dpp");
  const float lo = hz_readlane_f(v, 31), hi = hz_readlane_f(v, 63);
  return q.h ? hi : lo;
}
__device__ __forceinline__ float hh_min(float v, const HalfLane& q) {
  HH_HALF_REDUCE("v_min_f32_dpp");
  const float lo = hz_readlane_f(v, 31), hi = hz_readlane_f(v, 63);
  return q.h ? hi : lo;
}
__device__ __forceinline__ int hh_any_max_i(int v) {  // max of the two halves' (per-half uniform) values: wave-uniform
  const int a = hz_readlane_i(v, 0), b = hz_readlane_i(v, 32);
  return a > b ? a : b;
}

// hz_ordered_sum per half: after the pass lane l of a half holds init + x_0 + ... + x_l of ITS half (lane 0 of either half
// takes `init`: the value wave_shr:1 brings into lane 32 from lane 31 is replaced).  `steps` >= the highest lane that matters
// + 1 in either half (extra steps change nothing); returns the running sums, the caller picks its lane.
__device__ __forceinline__ float hh_ordered_scan(float x, int steps, float init, const HalfLane& q) {
  float acc = 0.0f;
  for (int s = 0; s < steps; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(init), __float_as_int(acc), 0x138, 0xf, 0xf, false));
      t = q.l == 0 ? init : t;
      acc = t + x;
    }
  }
  return acc;
}

// expand_prior with every child legal (the expansion of a leaf, cnode.cpp:49-114 through cmulti_back_propagate)
__device__ __forceinline__ float hh_expand_prior_all(float logit, int A, const uint64_t* exp_tab, const HalfLane& q) {
  const bool legal = q.l < A;
  float m = (legal && logit == logit) ? logit : -INFINITY;
  m = hh_max(m, q);
  const float policy_max = fmaxf(m, HZ_FLOAT_MIN);
  const float tp = legal ? hz_expf(logit - policy_max, exp_tab) : 0.0f;
  const float policy_sum = hh_bcast_f(hh_ordered_scan(tp, A, 0.0001f, q), A - 1, q);
  float prior = legal ? tp / policy_sum : 0.0f;
  if (prior != prior) prior = 0.0f;
  return prior;
}

// per-half search state (TreeLocal of hz_tree_dev.h, every member per half)
struct HalfTree {
  int tree;         // global tree index of this lane's half
  bool mine;        // the half owns a tree (tree < N)
  int32_t* path;    // [S+1] LDS
  float4* prec;     // [S+1] LDS
  float* lq;        // [S]   LDS q cache
  float root_vsum;
  int root_visit;
  int path_len;
  float leaf_reward, leaf_value, leaf_logit;
  uint16_t* nextact;  // [S] LDS or null: per node its last selection, HZ_NEXTACT16 (the predicted lines below)
  bool deep;          // a descent of this tree has been HZ_TREE_REPLAY_MIN levels long: it keeps `nextact` and walks predicted lines
};

// a node's last selection, 16 bits: bit 15 = "has been passed", bits 8-14 = child entry + 1 (0: not expanded then), bits 0-7 = action
#define HZ_NEXTACT16(child_e, action) ((uint16_t)(0x8000 | (((child_e) + 1) << 8) | (action)))
struct HalfReplayOut;
template <int C>
__device__ __forceinline__ HalfReplayOut traverse_replay_half(const TreeView& tv, const HalfLane& q, const HalfTree& t, int sim,
                                                              float mn, float mx, const float* ptab, bool want, int start,
                                                              int depth0, float mq_in, int pvc_in);
struct HalfReplayOut {  // per half
  int e, action, depth, pvc;
  float parent_q;
  bool leaf;
};

// traverse_body<true> for two trees.  `publish` (uniform): store the bookkeeping scalars the read-outs use.
// Returns this half's leaf parent entry; its action goes to *la_slot (LDS, per half).  A tree that has grown deep walks its
// predicted lines eight levels at a time first (traverse_replay_half below); the halves may then stand at different depths.
// MODE 0: the plain walk.  1: the plain walk in a kernel that keeps tables of last choices -- none of the wave's two trees
// has grown deep yet; one that does so now starts its table.  2: at least one has: passes first, the table kept up to date.
// (Two copies of the walk in the kernel: with the passes compiled in, depth and the tables' predicates are per half, and
// the plain walk of a wave that never needs them paid 2.4 % at 8192 envs for it.)
template <int MODE = 0>
__device__ __forceinline__ int traverse_half(const TreeView& tv, const HalfLane& q, HalfTree& t, int sim, float mn, float mx,
                                             float4 root_row, const float* tab /* LDS: [64] pb_c log factors, [64] sqrt(n + 1) */,
                                             const float* ptab /* LDS or null: hz_ptab_index */, int32_t* la_slot, int32_t* ix,
                                             int32_t* iy, bool publish) {
  const int A = tv.A, S = tv.S;
  const bool on = q.l < A;
  const float discount = tv.discount;
  const float delta = mx - mn;
  const float4* rec = tv.rec + (size_t)(t.mine ? t.tree : 0) * S * A;
  int e = 0, pvc = t.root_visit, depth = 0, action = 0;  // (all per half)
  bool active = t.mine;
  float parent_q = 0.0f;
  int leaf_e = 0, leaf_action = 0, leaf_depth = 1;
  const bool keeps_table = MODE >= 1 && HZ_TREE_REPLAY && A <= 20 && ptab != nullptr && t.nextact != nullptr;
  if (MODE == 2 && keeps_table) {
    bool want = active && t.deep;
    while (__ballot(want) != 0) {
      const HalfReplayOut o = traverse_replay_half<5>(tv, q, t, sim, mn, mx, ptab, want, e, depth, parent_q, pvc);
      if (want) {
        action = o.action;
        parent_q = o.parent_q;
        depth = o.depth;
        pvc = o.pvc;
        if (o.leaf) {
          active = false;
          leaf_e = o.e;
          leaf_action = o.action;
          leaf_depth = o.depth;
        } else {
          e = o.e;
        }
      }
      // another pass where the node has been passed before and its choice led to an expanded node
      const int na = (want && active) ? (int)t.nextact[e] : 0;
      want = want && active && (na & 0x8000) && ((na >> 8) & 0x7f);
    }
  }
  const bool table_on = MODE == 2 && keeps_table && t.deep;  // (per half: the ordinary walk below enters its choices too)
  if (__ballot(active) != 0) {
    // a level's child records: the root's in registers, a node's requested at the end of the level above
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (depth == 0) r = root_row;
    else if (on && active) r = rec[(size_t)e * A + q.l];
    while (true) {
      const uint32_t w = __float_as_uint(r.w);
      const int visit = (int)(w >> 16);
      const int child = (int)(w & 0xffffu) - 1;
      float prior = r.x;
      if (prior != prior) prior = 0.0f;
      // reward + discount * value_sum / visits of a visited child = what the last backup over that edge left in the q cache
      // (backprop_half: t.lq[child], same expression, same operands): an LDS word instead of a correctly rounded division
      float qsa = r.z + discount * 0.0f;
      if (visit > 0) qsa = t.lq[child];
      const bool vis = on && visit > 0;
      const uint32_t vm = hh_ballot(vis, q);
      const int nvis = __popc(vm);
      float total = 0.0f;
      if (hh_any_max_i(nvis) > 4) {
        const int n = 32 - __clz(vm);  // highest visited lane + 1 (0: none)
        const float scan = hh_ordered_scan(vis ? qsa : 0.0f, hh_any_max_i(n), 0.0f, q);
        const float tot = hh_bcast_f(scan, n > 0 ? n - 1 : 0, q);
        total = n > 0 ? tot : 0.0f;
      } else {
        uint32_t rem = vm;
        const int trips = hh_any_max_i(nvis);
        for (int k = 0; k < trips; ++k) {
          const int a = rem ? __ffs(rem) - 1 : 0;
          const float v = hh_bcast_f(qsa, a, q);
          if (rem) total += v;
          rem &= rem - 1;
        }
      }
      const bool root_mean = depth == 0 && nvis > 0;
      const float mean_q = (root_mean ? total : parent_q + total) / (float)(root_mean ? nvis : nvis + 1);
      parent_q = mean_q;
      const bool tab_in_regs = S < 64;
      float pb_c;
      if (ptab != nullptr) {
        pb_c = ptab[hz_ptab_index(pvc, visit)];
      } else {
        pb_c = tab_in_regs ? tab[pvc] : tv.pbc_tab[pvc];
        const float sq = tab_in_regs ? tab[64 + pvc] : sqrtf((float)pvc + 1.0f);
        pb_c = pb_c * (sq / (float)(visit + 1));
      }
      const float prior_score = pb_c * prior;
      float vs = (visit == 0) ? mean_q : qsa;
      if (delta > 0.0f) vs = (vs - mn) / (delta < tv.delta ? tv.delta : delta);
      if (vs < 0.0f) vs = 0.0f;
      if (vs > 1.0f) vs = 1.0f;
      const float score = prior_score + vs;
      const bool valid = on && (score == score) && (score > HZ_FLOAT_MIN);
      const float M = hh_max(valid ? score : -INFINITY, q);
      const uint32_t eq = hh_ballot(valid && score == M, q);
      const float thr = M - 0.000001f;
      uint32_t cand = hh_ballot(valid && score >= thr, q);
      action = 0;
      if (eq != 0) {
        const int first = __ffs(eq) - 1;
        cand &= ~((1u << first) - 1u);
        const uint32_t cnt = (uint32_t)__popc(cand);
        if (cnt > 1) {
          const uint32_t rnd = hz_tiebreak_rand(tv.seed, tv.id_base + (uint32_t)t.tree, (uint32_t)sim, (uint32_t)depth);
          uint32_t k = rnd % cnt;
          while (k--) cand &= cand - 1;
        }
        action = __ffs(cand) - 1;
      }
      if (q.l == 0 && active) {
        tv.best_action[(size_t)t.tree * S + e] = (int8_t)action;
        t.path[depth] = (e << 8) | action;
      }
      if (q.l == action && active) t.prec[depth] = r;
      const int child_e = hh_bcast_i(child, action, q);
      const int child_visit = hh_bcast_i(visit, action, q);
      if (table_on && q.l == 0 && active) t.nextact[e] = HZ_NEXTACT16(child_e, action);
      ++depth;
      if (active && (child_e < 0 || depth >= S)) {
        active = false;
        leaf_e = e;
        leaf_action = action;
        leaf_depth = depth;
      }
      if (active) {
        e = child_e;
        pvc = child_visit;
      }
      if (__ballot(active) == 0) break;
      r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on && active) r = rec[(size_t)e * A + q.l];
    }
  }
  if (q.l == 0 && t.mine) {
    *la_slot = leaf_action;
    if (publish) {
      ix[t.tree] = leaf_e;
      iy[t.tree] = t.tree;
      tv.path_len[t.tree] = leaf_depth + 1;
    }
  }
  t.path_len = leaf_depth + 1;
  if (keeps_table && t.mine && !t.deep && leaf_depth >= HZ_TREE_REPLAY_MIN) {
    // the tree has grown deep: from now on it keeps the table of last choices, which starts as this path (hz_tree_dev.h)
    for (int k = q.l; k < leaf_depth; k += 32) {
      const int here = t.path[k], below = k + 1 < leaf_depth ? (t.path[k + 1] >> 8) : -1;
      t.nextact[here >> 8] = HZ_NEXTACT16(below, here & 255);
    }
    t.deep = true;
  }
  return leaf_e;
}

// backprop_body<false, true> for two trees; `e_new` (uniform) = the entry this simulation creates.  Returns the new
// min / max / root visit count per half and the root's child record the backup changed (+ its action).
__device__ __forceinline__ void backprop_half(const TreeView& tv, const HalfLane& q, HalfTree& t, int e_new,
                                              const uint64_t* exp_tab, bool publish, float& out_mn, float& out_mx,
                                              int& out_root_visit, float4& out_first_rec, int& out_first_action) {
  const int A = tv.A, S = tv.S;
  const bool on = q.l < A;
  const float discount = tv.discount;
  float4* rec = tv.rec + (size_t)(t.mine ? t.tree : 0) * S * A;
  int pr0 = 0;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q.l <= S) {
    pr0 = t.path[q.l];
    r0 = t.prec[q.l];
  }
  const float old_root_vsum = t.root_vsum;
  const int old_root_visit = t.root_visit;
  const float prior = hh_expand_prior_all(on ? t.leaf_logit : 0.0f, A, exp_tab, q);
  if (on && t.mine) {
    float4 r;
    r.x = prior; r.y = 0.0f; r.z = 0.0f; r.w = __uint_as_float(pack_vc(0, -1));
    rec[(size_t)e_new * A + q.l] = r;
  }
  const int npairs = t.mine ? t.path_len - 1 : 0;
  float G = t.leaf_value;
  const float leaf_reward = t.leaf_reward;
  int pr = 0;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  const int np_max = hh_any_max_i(npairs);
  for (int base = np_max > 0 ? (((np_max - 1) >> 5) << 5) : 0; base >= 0; base -= 32) {
    const int k = base + q.l;
    const bool act = k < npairs;
    if (base == 0) {
      pr = pr0;
      r = r0;
    } else if (act) {
      pr = t.path[k];
      r = t.prec[k];
    }
    uint32_t w = __float_as_uint(r.w);
    int visit = (int)(w >> 16);
    int child = (int)(w & 0xffffu) - 1;
    if (act && k == npairs - 1) {
      r.z = leaf_reward;
      child = e_new;
      if (t.deep) t.nextact[pr >> 8] = HZ_NEXTACT16(e_new, pr & 255);  // (a deep tree's table: the edge now leads to an entry)
    }
    float myG = 0.0f;
    const int hi = min(npairs - 1 - base, 31);  // this half's last edge in the chunk (< 0: none)
    const int hi_max = hh_any_max_i(hi);
    for (int j = hi_max; j >= 0; --j) {
      const float rj = q.h ? hz_readlane_f(r.z, 32 + j) : hz_readlane_f(r.z, j);
      const bool take = j <= hi;
      if (q.l == j && take) myG = G;
      G = take ? rj + discount * G : G;
    }
    if (act) {
      r.y += myG;
      visit += 1;
      r.w = __uint_as_float(pack_vc(visit, child));
      rec[(size_t)(pr >> 8) * A + (pr & 255)] = r;
      const float qv = r.z + discount * (r.y / (float)visit);
      t.lq[child] = qv;
    }
  }
  out_first_rec.x = hh_bcast_f(r.x, 0, q);
  out_first_rec.y = hh_bcast_f(r.y, 0, q);
  out_first_rec.z = hh_bcast_f(r.z, 0, q);
  out_first_rec.w = hh_bcast_f(r.w, 0, q);
  out_first_action = hh_bcast_i(pr, 0, q) & 255;
  const float new_root_vsum = old_root_vsum + G;
  out_root_visit = old_root_visit + 1;
  if (q.l == 0 && t.mine && publish) {
    tv.root_vsum[t.tree] = new_root_vsum;
    tv.root_visit[t.tree] = out_root_visit;
  }
  t.root_vsum = new_root_vsum;
  t.root_visit = out_root_visit;
  float vmax = -INFINITY, vmin = INFINITY;
  for (int e = 1 + q.l; e <= e_new; e += 32) {
    const float qv = t.lq[e];
    if (qv == qv) {
      vmax = fmaxf(vmax, qv);
      vmin = fminf(vmin, qv);
    }
  }
  out_mx = fmaxf(hh_max(vmax, q), HZ_FLOAT_MIN);
  out_mn = fminf(hh_min(vmin, q), HZ_FLOAT_MAX);
  if (q.l == 0 && t.mine && publish) {
    tv.mm_max[t.tree] = out_mx;
    tv.mm_min[t.tree] = out_mn;
  }
}

// ------------------------------------------------------------------------------------------ predicted lines, two trees
// hz_tree_replay_dev.h for the side-by-side layout: eight levels of EACH half's tree per pass (four lanes per level, C
// children per lane), the same arithmetic in the same order.  What differs from the one-tree pass: a half has 32 lanes, so the
// predicted line is found by walking the table (eight dependent LDS reads, both halves at once) instead of pointer doubling;
// the mean-q chain and the bookkeeping take their operands per half; the table is 16 bits per node (the 32-tree workgroup has
// 3 KB of LDS left, not 8).  A half that wants no pass idles through the other's with every slot empty.
template <int C>
__device__ __forceinline__ HalfReplayOut traverse_replay_half(const TreeView& tv, const HalfLane& q, const HalfTree& t, int sim,
                                                              float mn, float mx, const float* ptab, bool want, int start,
                                                              int depth0, float mq_in, int pvc_in) {
  const int A = tv.A, S = tv.S;
  const int lane = q.hbase + q.l;
  const int g = q.l & 3, j = q.l >> 2;  // slot j of this half: level depth0 + j
  const float delta = mx - mn;
  const float dn = delta < tv.delta ? tv.delta : delta;
  const float4* rec = tv.rec + (size_t)(t.mine ? t.tree : 0) * S * A;
  // the predicted line below `start`: walk the table (all lanes of a half read the same word)
  int n = -1, ap = -1;
  {
    int node = want ? start : -1;
    for (int s = 0; s < 8; ++s) {
      const int na = node >= 0 ? (int)t.nextact[node] : 0;
      if (j == s) {
        n = node;
        ap = (na & 0x8000) ? (na & 255) : -1;  // the node's last selection (none: never passed)
      }
      const int c1 = (na >> 8) & 0x7f;
      node = ((na & 0x8000) && c1) ? c1 - 1 : -1;
      if (__ballot(node >= 0) == 0) break;
    }
  }
  const bool lvl = n >= 0;
  const int nn = lvl ? n : 0;
  const int k = depth0 + j;
  // (of a child's record the pass needs the prior and the packed visits / child word; the one record the backup will update
  // is read again at the end: the 32-tree kernel has no registers for five whole records per lane)
  float Rx[C];
  uint32_t Rw[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int a = g * C + i;
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lvl && a < A) r4 = rec[(size_t)nn * A + a];
    Rx[i] = r4.x;
    Rw[i] = __float_as_uint(r4.w);
  }
  float qv[C];
  int nv = 0, wprev = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int a = g * C + i;
    const uint32_t w = Rw[i];
    const int visit = (int)(w >> 16);
    const int child = (int)(w & 0xffffu) - 1;
    const bool vis = lvl && a < A && visit > 0;
    qv[i] = 0.0f;  // (an unvisited child's q enters neither the sum nor its score)
    if (vis) qv[i] = t.lq[child];
    nv += vis ? 1 : 0;
    if (a == ap) wprev = (int)w;
  }
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float f = r == 0 ? 0.0f : __int_as_float(HZ_QUAD(__float_as_int(s), 0x90));
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const uint32_t w = Rw[i];
      const float with = f + qv[i];
      f = (lvl && g * C + i < A && (w >> 16) > 0) ? with : f;
    }
    s = g == r ? f : s;
  }
  const float total = __int_as_float(HZ_QUAD(__float_as_int(s), 0xFF));
  nv += HZ_QUAD(nv, 0xB1);
  nv += HZ_QUAD(nv, 0x4E);
  wprev = hz_quad_or(wprev);
  int pvc = __builtin_amdgcn_ds_bpermute(4 * (lane - 4), wprev >> 16);  // (slot j - 1 of the same half for j >= 1)
  if (j == 0) pvc = pvc_in;
  if (!lvl) pvc = 0;
  float mq = 0.0f;
  {
    const bool rootm = k == 0 && nv > 0;
    const float den = (float)(rootm ? nv : nv + 1);
    const uint64_t lb = __ballot(lvl);
    const int steps = max(__popc((uint32_t)lb), __popc((uint32_t)(lb >> 32))) >> 2;  // the longer of the two lines
    float pq = mq_in;
    for (int r = 0; r < steps; ++r) {
      const float cand = (rootm ? total : pq + total) / den;
      if (j == r) mq = cand;
      const float lo = hz_readlane_f(cand, 4 * r), hi = hz_readlane_f(cand, 32 + 4 * r);
      pq = q.h ? hi : lo;
    }
  }
  float score[C];
  float M = -INFINITY;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int a = g * C + i;
    const uint32_t w = Rw[i];
    const int visit = (int)(w >> 16);
    float prior = Rx[i];
    if (prior != prior) prior = 0.0f;
    const float pb_c = ptab[hz_ptab_index(pvc, lvl ? visit : 0)];
    const float prior_score = pb_c * prior;
    float vs = (visit == 0) ? mq : qv[i];
    if (delta > 0.0f) vs = (vs - mn) / dn;
    if (vs < 0.0f) vs = 0.0f;
    if (vs > 1.0f) vs = 1.0f;
    const float sc = prior_score + vs;
    const bool valid = lvl && a < A && (sc == sc) && (sc > HZ_FLOAT_MIN);
    score[i] = valid ? sc : -INFINITY;
    M = fmaxf(M, score[i]);
  }
  M = fmaxf(M, __int_as_float(HZ_QUAD(__float_as_int(M), 0xB1)));
  M = fmaxf(M, __int_as_float(HZ_QUAD(__float_as_int(M), 0x4E)));
  const float thr = M - 0.000001f;
  int eqb = 0, cb = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const bool valid = score[i] != -INFINITY;
    if (valid && score[i] == M) eqb |= 1 << (g * C + i);
    if (valid && score[i] >= thr) cb |= 1 << (g * C + i);
  }
  eqb = hz_quad_or(eqb);
  uint32_t cand = (uint32_t)hz_quad_or(cb);
  int action = 0;
  if (eqb != 0) {
    const int first = __ffs(eqb) - 1;
    cand &= ~((1u << first) - 1u);
    const uint32_t cnt = (uint32_t)__popc(cand);
    if (cnt > 1) {
      const uint32_t rnd = hz_tiebreak_rand(tv.seed, tv.id_base + (uint32_t)t.tree, (uint32_t)sim, (uint32_t)k);
      uint32_t kk = rnd % cnt;
      while (kk--) cand &= cand - 1;
    }
    action = __ffs(cand) - 1;
  }
  int wsel = 0;
#pragma unroll
  for (int i = 0; i < C; ++i)
    if (g * C + i == action) wsel = (int)Rw[i];
  wsel = hz_quad_or(wsel);
  // per half: the pass ends at the first level that chose differently from the prediction (or at the line's last node)
  const uint32_t lbh = hh_ballot(lvl, q);
  const int last = lbh ? ((31 - __clz(lbh)) >> 2) : 0;
  const uint32_t sbh = hh_ballot(lvl && action != ap, q);
  const int first_off = sbh ? ((__ffs(sbh) - 1) >> 2) : 7;
  const int mslot = min(first_off, last);
  const bool commit = lvl && j <= mslot;
  if (commit && g == 0) {
    tv.best_action[(size_t)t.tree * S + nn] = (int8_t)action;
    t.path[k] = (nn << 8) | action;
    t.nextact[nn] = (uint16_t)(0x8000 | ((wsel & 0x7f) << 8) | action);
  }
  if (commit && g == 0) t.prec[k] = rec[(size_t)nn * A + action];  // == the record as the pass read it (nothing stores in between)
  HalfReplayOut out;
  const int src = 4 * mslot;
  const uint32_t wv = (uint32_t)hh_bcast_i(wsel, src, q);
  const int child_e = (int)(wv & 0xffffu) - 1;
  out.action = hh_bcast_i(action, src, q);
  out.parent_q = hh_bcast_f(mq, src, q);
  out.depth = depth0 + mslot + 1;
  out.pvc = (int)(wv >> 16);
  out.leaf = child_e < 0 || out.depth >= S;
  out.e = out.leaf ? hh_bcast_i(nn, src, q) : child_e;
  return out;
}
