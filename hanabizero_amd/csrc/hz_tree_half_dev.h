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
};

// traverse_body<true> for two trees.  `publish` (uniform): store the bookkeeping scalars the read-outs use.
// Returns this half's leaf parent entry; its action goes to *la_slot (LDS, per half).
__device__ __forceinline__ int traverse_half(const TreeView& tv, const HalfLane& q, HalfTree& t, int sim, float mn, float mx,
                                             float4 root_row, const float* tab /* LDS: [64] pb_c log factors, [64] sqrt(n + 1) */,
                                             const float* ptab /* LDS or null: hz_ptab_index */, int32_t* la_slot, int32_t* ix,
                                             int32_t* iy, bool publish) {
  const int A = tv.A, S = tv.S;
  const bool on = q.l < A;
  const float discount = tv.discount;
  const float delta = mx - mn;
  const float4* rec = tv.rec + (size_t)(t.mine ? t.tree : 0) * S * A;
  int e = 0, pvc = t.root_visit, depth = 0, action = 0;
  bool is_root = true, active = t.mine;
  float parent_q = 0.0f;
  int leaf_e = 0, leaf_action = 0, leaf_depth = 1;
  float4 r = root_row;  // a level's child records: the root's in registers, a node's requested at the end of the level above
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
    const bool root_mean = is_root && nvis > 0;
    const float mean_q = (root_mean ? total : parent_q + total) / (float)(root_mean ? nvis : nvis + 1);
    is_root = false;
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
  if (q.l == 0 && t.mine) {
    *la_slot = leaf_action;
    if (publish) {
      ix[t.tree] = leaf_e;
      iy[t.tree] = t.tree;
      tv.path_len[t.tree] = leaf_depth + 1;
    }
  }
  t.path_len = leaf_depth + 1;
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
