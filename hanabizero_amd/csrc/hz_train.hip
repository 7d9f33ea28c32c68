// hz_train.hip -- BatchNorm1d (training) + residual + ReLU behind the learner's GEMMs, forward and backward (include/hz_train.h).
// Latency-bound work: a learner batch is 256 rows, a layer 256..1024 columns.  One workgroup per 32 columns; column sums meet in LDS.
#include "hz_common.h"
#include "hz_tree.h"
#include "hz_train.h"

template <int DT>
__device__ __forceinline__ float tr_load(const uint16_t* p) {
  const uint32_t b = *p;
  if (DT == HZ_BF16) return __uint_as_float(b << 16);
  const uint16_t h = (uint16_t)b;
  return (float)*reinterpret_cast<const _Float16*>(&h);
}
template <int DT>
__device__ __forceinline__ uint16_t tr_round(float v) {
  if (DT == HZ_BF16) {  // round to nearest even, NaN kept quiet (torch's float -> bfloat16)
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  const _Float16 h = (_Float16)v;  // round to nearest even
  return *reinterpret_cast<const uint16_t*>(&h);
}

template <int DT>
__device__ __forceinline__ float tr_requant(float v) {  // through the element format and back: where PyTorch materialises a 16-bit tensor
  const uint16_t h = tr_round<DT>(v);
  return tr_load<DT>(&h);
}

// Thread (g, p) = (tid >> 2, tid & 3) of a 256-thread workgroup owns column pair p of the workgroup's 8 columns (one 4-B access per
// row) in rows g, g + 64, g + 128, ...: C / 8 workgroups (64 .. 128 for a layer), few dependent steps each.  Up to TR_HOLD rows per
// thread stay in registers between the passes (the learner's batch of 256: every row, so each input is read from memory ONCE, all
// loads of a thread in flight together); beyond that the later passes read again (L2).  Column sums: four xor-shuffles across a
// wave's 16 row groups, then the four waves meet in LDS.
constexpr int TR_COLS = 8, TR_GROUPS = 64, TR_HOLD = 4;

template <int DT>
__device__ __forceinline__ float2 tr_load2(const uint16_t* p, bool both) {  // two adjacent elements (4-B aligned when both exist)
  if (both) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
    const uint16_t lo = (uint16_t)(w & 0xffffu), hi = (uint16_t)(w >> 16);
    return make_float2(tr_load<DT>(&lo), tr_load<DT>(&hi));
  }
  return make_float2(tr_load<DT>(p), 0.0f);
}
template <int DT>
__device__ __forceinline__ void tr_store2(uint16_t* p, float a, float b, bool both) {
  if (both) *reinterpret_cast<uint32_t*>(p) = (uint32_t)tr_round<DT>(a) | ((uint32_t)tr_round<DT>(b) << 16);
  else *p = tr_round<DT>(a);
}
// sums over the 64 row groups of a workgroup, per column pair (lane & 3); every thread gets its column pair's totals
__device__ __forceinline__ float4 tr_reduce(float4 v, float4 (*s)[4]) {
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) {
    v.x += __shfl_xor(v.x, m);
    v.y += __shfl_xor(v.y, m);
    v.z += __shfl_xor(v.z, m);
    v.w += __shfl_xor(v.w, m);
  }
  const int wave = threadIdx.x >> 6, p = threadIdx.x & 3;
  if ((threadIdx.x & 63) < 4) s[wave][p] = v;
  __syncthreads();
  float4 t = s[0][p];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    t.x += s[k][p].x; t.y += s[k][p].y; t.z += s[k][p].z; t.w += s[k][p].w;
  }
  __syncthreads();
  return t;
}

template <int DT>
__global__ __launch_bounds__(256) void k_bn_act_forward(const uint16_t* __restrict__ x, long long xs, const uint16_t* __restrict__ res,
                                                        long long rs, uint16_t* __restrict__ out, long long os, int rows, int cols,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                                        float* __restrict__ smean, float* __restrict__ sinv, int relu, int vec,
                                                        float* __restrict__ scratch) {
  __shared__ float4 s_r[4][4];
  const int g = threadIdx.x >> 2, p = threadIdx.x & 3, c = blockIdx.x * TR_COLS + 2 * p;
  const bool on = c < cols, both = vec && c + 1 < cols, on1 = c + 1 < cols;
  // blockIdx.y = the row group: `rows` consecutive rows normalised by their own statistics (one BatchNorm call of the module each)
  const int gy = blockIdx.y, G = gridDim.y;
  x += (long long)gy * rows * xs;
  out += (long long)gy * rows * os;
  if (res != nullptr) res += (long long)gy * rows * rs;
  smean += (long long)gy * cols;
  sinv += (long long)gy * cols;
  float2 v[TR_HOLD];
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i) {
    const int r = g + TR_GROUPS * i;
    v[i] = make_float2(0.0f, 0.0f);
    if (on && r < rows) {
      v[i] = tr_load2<DT>(x + (long long)r * xs + c, both);
      if (!both && on1) v[i].y = tr_load<DT>(x + (long long)r * xs + c + 1);
    }
  }
  float2 sum = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i) { sum.x += v[i].x; sum.y += v[i].y; }
  for (int r = g + TR_GROUPS * TR_HOLD; on && r < rows; r += TR_GROUPS) {
    sum.x += tr_load<DT>(x + (long long)r * xs + c);
    if (on1) sum.y += tr_load<DT>(x + (long long)r * xs + c + 1);
  }
  {
    const float4 t = tr_reduce(make_float4(sum.x, sum.y, 0.0f, 0.0f), s_r);
    sum = make_float2(t.x, t.y);
  }
  const float2 mean = make_float2(sum.x / (float)rows, sum.y / (float)rows);
  float2 sq = make_float2(0.0f, 0.0f);  // (second moment about the mean: no cancellation)
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i)
    if (g + TR_GROUPS * i < rows) {
      const float dx = v[i].x - mean.x, dy = v[i].y - mean.y;
      sq.x += dx * dx;
      sq.y += dy * dy;
    }
  for (int r = g + TR_GROUPS * TR_HOLD; on && r < rows; r += TR_GROUPS) {
    const float dx = tr_load<DT>(x + (long long)r * xs + c) - mean.x;
    sq.x += dx * dx;
    if (on1) {
      const float dy = tr_load<DT>(x + (long long)r * xs + c + 1) - mean.y;
      sq.y += dy * dy;
    }
  }
  {
    const float4 t = tr_reduce(make_float4(sq.x, sq.y, 0.0f, 0.0f), s_r);
    sq = make_float2(t.x, t.y);
  }
  const float2 var = make_float2(sq.x / (float)rows, sq.y / (float)rows);
  const float2 inv = make_float2(rsqrtf(var.x + eps), rsqrtf(var.y + eps));
  const float ub = rows > 1 ? (float)rows / (float)(rows - 1) : 1.0f;
  if (on && g == 0) {
    smean[c] = mean.x; sinv[c] = inv.x;
    if (on1) { smean[c + 1] = mean.y; sinv[c + 1] = inv.y; }
    if (G == 1) {
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean.x;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (var.x * ub);
      if (on1) {
        rmean[c + 1] = (1.0f - momentum) * rmean[c + 1] + momentum * mean.y;
        rvar[c + 1] = (1.0f - momentum) * rvar[c + 1] + momentum * (var.y * ub);
      }
    } else {  // the running statistics take the groups' batches one after the other, as the module's G calls would: hz_bn_groups_finish
      scratch[((long long)gy * 2 + 0) * cols + c] = mean.x;
      scratch[((long long)gy * 2 + 1) * cols + c] = var.x * ub;
      if (on1) {
        scratch[((long long)gy * 2 + 0) * cols + c + 1] = mean.y;
        scratch[((long long)gy * 2 + 1) * cols + c + 1] = var.y * ub;
      }
    }
  }
  if (!on) return;
  const float2 ga = make_float2(gamma[c], on1 ? gamma[c + 1] : 0.0f), be = make_float2(beta[c], on1 ? beta[c + 1] : 0.0f);
  auto finish = [&](int r, float2 xv) {
    float a = (xv.x - mean.x) * inv.x * ga.x + be.x, b = (xv.y - mean.y) * inv.y * ga.y + be.y;
    if (res != nullptr) {  // (bn's output is a 16-bit tensor before the add, as in PyTorch)
      float2 rv = tr_load2<DT>(res + (long long)r * rs + c, both);
      if (!both && on1) rv.y = tr_load<DT>(res + (long long)r * rs + c + 1);
      a = tr_requant<DT>(a) + rv.x;
      b = tr_requant<DT>(b) + rv.y;
    }
    if (relu) {
      if (!(a > 0.0f)) a = a != a ? a : 0.0f;
      if (!(b > 0.0f)) b = b != b ? b : 0.0f;
    }
    tr_store2<DT>(out + (long long)r * os + c, a, b, both);
    if (!both && on1) out[(long long)r * os + c + 1] = tr_round<DT>(b);
  };
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i)
    if (g + TR_GROUPS * i < rows) finish(g + TR_GROUPS * i, v[i]);
  for (int r = g + TR_GROUPS * TR_HOLD; r < rows; r += TR_GROUPS)
    finish(r, make_float2(tr_load<DT>(x + (long long)r * xs + c), on1 ? tr_load<DT>(x + (long long)r * xs + c + 1) : 0.0f));
}

template <int DT>
__global__ __launch_bounds__(256) void k_bn_act_backward(const uint16_t* __restrict__ dout, long long ds, const uint16_t* __restrict__ out,
                                                         long long os, const uint16_t* __restrict__ x, long long xs,
                                                         uint16_t* __restrict__ dx, long long dxs, uint16_t* __restrict__ dres,
                                                         long long drs, int rows, int cols, const float* __restrict__ gamma,
                                                         const float* __restrict__ smean, const float* __restrict__ sinv,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int relu, int vec,
                                                         float* __restrict__ scratch) {
  __shared__ float4 s_r[4][4];
  const int g = threadIdx.x >> 2, p = threadIdx.x & 3, c = blockIdx.x * TR_COLS + 2 * p;
  const bool on = c < cols, both = vec && c + 1 < cols, on1 = c + 1 < cols;
  const int gy = blockIdx.y, G = gridDim.y;  // row groups as in the forward kernel
  dout += (long long)gy * rows * ds;
  x += (long long)gy * rows * xs;
  dx += (long long)gy * rows * dxs;
  if (out != nullptr) out += (long long)gy * rows * os;
  if (dres != nullptr) dres += (long long)gy * rows * drs;
  smean += (long long)gy * cols;
  sinv += (long long)gy * cols;
  const float2 mean = make_float2(on ? smean[c] : 0.0f, on1 ? smean[c + 1] : 0.0f);
  const float2 inv = make_float2(on ? sinv[c] : 0.0f, on1 ? sinv[c + 1] : 0.0f);
  auto fetch = [&](int r, float2& dz, float2& xh) {  // dz = dout masked by the ReLU, xh = the normalised input
    dz = tr_load2<DT>(dout + (long long)r * ds + c, both);
    float2 xv = tr_load2<DT>(x + (long long)r * xs + c, both);
    if (!both && on1) {
      dz.y = tr_load<DT>(dout + (long long)r * ds + c + 1);
      xv.y = tr_load<DT>(x + (long long)r * xs + c + 1);
    }
    if (relu) {
      float2 o = tr_load2<DT>(out + (long long)r * os + c, both);
      if (!both && on1) o.y = tr_load<DT>(out + (long long)r * os + c + 1);
      if (!(o.x > 0.0f)) dz.x = 0.0f;
      if (!(o.y > 0.0f)) dz.y = 0.0f;
    }
    xh = make_float2((xv.x - mean.x) * inv.x, (xv.y - mean.y) * inv.y);
  };
  float2 dzv[TR_HOLD], xhv[TR_HOLD];
  float2 s1 = make_float2(0.0f, 0.0f), s2 = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i) {
    dzv[i] = xhv[i] = make_float2(0.0f, 0.0f);
    if (on && g + TR_GROUPS * i < rows) fetch(g + TR_GROUPS * i, dzv[i], xhv[i]);
  }
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i) {
    s1.x += dzv[i].x; s1.y += dzv[i].y;
    s2.x += dzv[i].x * xhv[i].x; s2.y += dzv[i].y * xhv[i].y;
  }
  for (int r = g + TR_GROUPS * TR_HOLD; on && r < rows; r += TR_GROUPS) {
    float2 dz, xh;
    fetch(r, dz, xh);
    s1.x += dz.x; s1.y += dz.y;
    s2.x += dz.x * xh.x; s2.y += dz.y * xh.y;
  }
  {
    const float4 t = tr_reduce(make_float4(s1.x, s1.y, s2.x, s2.y), s_r);  // both sums in one trip
    s1 = make_float2(t.x, t.y);
    s2 = make_float2(t.z, t.w);
  }
  if (on && g == 0) {
    if (G == 1) {
      dbeta[c] += s1.x; dgamma[c] += s2.x;
      if (on1) { dbeta[c + 1] += s1.y; dgamma[c + 1] += s2.y; }
    } else {  // the groups' sums are added up in the order of the groups (the same bits whatever the schedule): hz_bn_groups_finish
      scratch[((long long)gy * 2 + 0) * cols + c] = s1.x;
      scratch[((long long)gy * 2 + 1) * cols + c] = s2.x;
      if (on1) {
        scratch[((long long)gy * 2 + 0) * cols + c + 1] = s1.y;
        scratch[((long long)gy * 2 + 1) * cols + c + 1] = s2.y;
      }
    }
  }
  if (!on) return;
  const float2 m1 = make_float2(s1.x / (float)rows, s1.y / (float)rows), m2 = make_float2(s2.x / (float)rows, s2.y / (float)rows);
  const float2 k = make_float2(gamma[c] * inv.x, on1 ? gamma[c + 1] * inv.y : 0.0f);
  auto finish = [&](int r, float2 dz, float2 xh) {
    const float a = k.x * (dz.x - m1.x - xh.x * m2.x), b = k.y * (dz.y - m1.y - xh.y * m2.y);
    tr_store2<DT>(dx + (long long)r * dxs + c, a, b, both);
    if (!both && on1) dx[(long long)r * dxs + c + 1] = tr_round<DT>(b);
    if (dres != nullptr) {
      tr_store2<DT>(dres + (long long)r * drs + c, dz.x, dz.y, both);
      if (!both && on1) dres[(long long)r * drs + c + 1] = tr_round<DT>(dz.y);
    }
  };
#pragma unroll
  for (int i = 0; i < TR_HOLD; ++i)
    if (g + TR_GROUPS * i < rows) finish(g + TR_GROUPS * i, dzv[i], xhv[i]);
  for (int r = g + TR_GROUPS * TR_HOLD; r < rows; r += TR_GROUPS) {
    float2 dz, xh;
    fetch(r, dz, xh);
    finish(r, dz, xh);
  }
}

extern "C" int hz_bn_act_forward(const void* x, int64_t x_stride, const void* res, int64_t res_stride, void* out, int64_t out_stride, int rows,
                                 int cols, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, float* save_mean, float* save_invstd, int relu, int dtype, void* stream) {
  return hz_bn_act_forward_groups(x, x_stride, res, res_stride, out, out_stride, rows, 1, cols, gamma, beta, running_mean, running_var, momentum,
                                  eps, save_mean, save_invstd, nullptr, relu, dtype, stream);
}

extern "C" int hz_bn_act_forward_groups(const void* x, int64_t x_stride, const void* res, int64_t res_stride, void* out, int64_t out_stride,
                                        int rows, int groups, int cols, const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, float momentum, float eps, float* save_mean, float* save_invstd, float* scratch,
                                        int relu, int dtype, void* stream) {
  HZ_REQUIRE(x && out && gamma && beta && running_mean && running_var && save_mean && save_invstd, "hz_bn_act_forward: null pointer");
  HZ_REQUIRE(groups >= 1 && groups <= 1024 && (groups == 1 || scratch), "hz_bn_act_forward_groups: groups=%d (more than one needs scratch)", groups);
  HZ_REQUIRE(rows >= 1 && cols >= 1 && x_stride >= cols && out_stride >= cols && (!res || res_stride >= cols),
             "hz_bn_act_forward: rows=%d cols=%d strides %lld / %lld / %lld", rows, cols, (long long)x_stride, (long long)out_stride, (long long)res_stride);
  HZ_REQUIRE(dtype == HZ_BF16 || dtype == HZ_F16, "hz_bn_act_forward: dtype %d (HZ_BF16 or HZ_F16)", dtype);
  HZ_REQUIRE(eps > 0.0f && momentum >= 0.0f && momentum <= 1.0f, "hz_bn_act_forward: eps=%g momentum=%g", (double)eps, (double)momentum);
  const dim3 grid((cols + TR_COLS - 1) / TR_COLS, groups);
  // (two adjacent columns per 4-B access where every row starts 4-B aligned)
  const int vec = (x_stride % 2 == 0 && out_stride % 2 == 0 && (!res || res_stride % 2 == 0) && ((uintptr_t)x % 4) == 0 && ((uintptr_t)out % 4) == 0 &&
                   (!res || ((uintptr_t)res % 4) == 0)) ? 1 : 0;
  if (dtype == HZ_BF16)
    hipLaunchKernelGGL(k_bn_act_forward<HZ_BF16>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (long long)x_stride, (const uint16_t*)res,
                       (long long)res_stride, (uint16_t*)out, (long long)out_stride, rows, cols, gamma, beta, running_mean, running_var, momentum, eps,
                       save_mean, save_invstd, relu, vec, scratch);
  else
    hipLaunchKernelGGL(k_bn_act_forward<HZ_F16>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (long long)x_stride, (const uint16_t*)res,
                       (long long)res_stride, (uint16_t*)out, (long long)out_stride, rows, cols, gamma, beta, running_mean, running_var, momentum, eps,
                       save_mean, save_invstd, relu, vec, scratch);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_bn_act_backward(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, const void* x, int64_t x_stride,
                                  void* dx, int64_t dx_stride, void* dres, int64_t dres_stride, int rows, int cols, const float* gamma,
                                  const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int relu, int dtype,
                                  void* stream) {
  return hz_bn_act_backward_groups(dout, dout_stride, out, out_stride, x, x_stride, dx, dx_stride, dres, dres_stride, rows, 1, cols, gamma, save_mean,
                                   save_invstd, dgamma, dbeta, nullptr, relu, dtype, stream);
}

extern "C" int hz_bn_act_backward_groups(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, const void* x,
                                         int64_t x_stride, void* dx, int64_t dx_stride, void* dres, int64_t dres_stride, int rows, int groups,
                                         int cols, const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                                         float* dbeta, float* scratch, int relu, int dtype, void* stream) {
  HZ_REQUIRE(dout && x && dx && gamma && save_mean && save_invstd && dgamma && dbeta && (out || !relu), "hz_bn_act_backward: null pointer");
  HZ_REQUIRE(groups >= 1 && groups <= 1024 && (groups == 1 || scratch), "hz_bn_act_backward_groups: groups=%d (more than one needs scratch)", groups);
  HZ_REQUIRE(rows >= 1 && cols >= 1 && dout_stride >= cols && x_stride >= cols && dx_stride >= cols && (!relu || out_stride >= cols) &&
                 (!dres || dres_stride >= cols),
             "hz_bn_act_backward: rows=%d cols=%d and a row stride below cols", rows, cols);
  HZ_REQUIRE(dtype == HZ_BF16 || dtype == HZ_F16, "hz_bn_act_backward: dtype %d (HZ_BF16 or HZ_F16)", dtype);
  const dim3 grid((cols + TR_COLS - 1) / TR_COLS, groups);
  auto al = [](const void* q, int64_t st) { return !q || (st % 2 == 0 && ((uintptr_t)q % 4) == 0); };
  const int vec = (al(dout, dout_stride) && al(relu ? out : nullptr, out_stride) && al(x, x_stride) && al(dx, dx_stride) && al(dres, dres_stride)) ? 1 : 0;
  if (dtype == HZ_BF16)
    hipLaunchKernelGGL(k_bn_act_backward<HZ_BF16>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dout, (long long)dout_stride,
                       (const uint16_t*)out, (long long)out_stride, (const uint16_t*)x, (long long)x_stride, (uint16_t*)dx, (long long)dx_stride,
                       (uint16_t*)dres, (long long)dres_stride, rows, cols, gamma, save_mean, save_invstd, dgamma, dbeta, relu, vec, scratch);
  else
    hipLaunchKernelGGL(k_bn_act_backward<HZ_F16>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dout, (long long)dout_stride,
                       (const uint16_t*)out, (long long)out_stride, (const uint16_t*)x, (long long)x_stride, (uint16_t*)dx, (long long)dx_stride,
                       (uint16_t*)dres, (long long)dres_stride, rows, cols, gamma, save_mean, save_invstd, dgamma, dbeta, relu, vec, scratch);
  HZ_HIP(hipGetLastError());
  return 0;
}


// What crosses the groups of hz_bn_act_*_groups, for any number of BatchNorm layers in one launch: thread = one column of one entry.
__global__ __launch_bounds__(256) void k_bn_groups_finish(const hz_bn_finish_t* __restrict__ entries, int backward) {
  const hz_bn_finish_t e = entries[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= e.cols) return;
  float a = e.dst0[c], b = e.dst1[c];
  if (backward) {
    float sa = 0.0f, sb = 0.0f;
    for (int k = 0; k < e.groups; ++k) {
      sa += e.scratch[((long long)k * 2 + 0) * e.cols + c];
      sb += e.scratch[((long long)k * 2 + 1) * e.cols + c];
    }
    a += sa;
    b += sb;
  } else {
    for (int k = 0; k < e.groups; ++k) {
      a = (1.0f - e.momentum) * a + e.momentum * e.scratch[((long long)k * 2 + 0) * e.cols + c];
      b = (1.0f - e.momentum) * b + e.momentum * e.scratch[((long long)k * 2 + 1) * e.cols + c];
    }
  }
  e.dst0[c] = a;
  e.dst1[c] = b;
}

extern "C" int hz_bn_groups_finish(const hz_bn_finish_t* entries, int num_entries, int max_cols, int backward, void* stream) {
  HZ_REQUIRE(entries && num_entries >= 1 && num_entries <= 65535 && max_cols >= 1, "hz_bn_groups_finish: entries=%p n=%d max_cols=%d", (const void*)entries,
             num_entries, max_cols);
  hipLaunchKernelGGL(k_bn_groups_finish, dim3((max_cols + 255) / 256, num_entries), dim3(256), 0, (hipStream_t)stream, entries, backward);
  HZ_HIP(hipGetLastError());
  return 0;
}


// ------------------------------------------------------------------------------------------------ the heads' losses of one inference
// One wavefront per batch row: lane l owns logits l, l + 64, ... of each head.  Everything in fp32 (the reference casts the
// logits to float before log_softmax: train.py:145-168 via .float()); the gradients leave in the logits' element format.
template <int DT>
__device__ __forceinline__ float hl_load(const void* base, long long i) {
  if (DT == HZ_F32) return reinterpret_cast<const float*>(base)[i];
  return tr_load<(DT == HZ_F32 ? HZ_BF16 : DT)>(reinterpret_cast<const uint16_t*>(base) + i);
}
template <int DT>
__device__ __forceinline__ void hl_store(void* base, long long i, float v) {
  if (DT == HZ_F32) reinterpret_cast<float*>(base)[i] = v;
  else reinterpret_cast<uint16_t*>(base)[i] = tr_round<(DT == HZ_F32 ? HZ_BF16 : DT)>(v);
}
__device__ __forceinline__ float hl_wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float hl_wave_max(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// log-softmax statistics of one row of n logits: returns (max, log of the sum of exp(x - max)); e[] keeps the lane's exponentials
template <int DT, int MAXPER>
__device__ __forceinline__ void hl_row_stats(const void* row, int n, int lane, float (&x)[MAXPER], float& mx, float& lse) {
  mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int j = lane + 64 * i;
    x[i] = j < n ? hl_load<DT>(row, j) : -INFINITY;
    mx = fmaxf(mx, x[i]);
  }
  mx = hl_wave_max(mx);
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i)
    if (lane + 64 * i < n) s += expf(x[i] - mx);
  lse = logf(hl_wave_sum(s));
}

// a categorical head against the two-hot of a scalar target: (loss, predicted scalar); writes the gradient row scaled by `fac`
template <int DT>
__device__ __forceinline__ void hl_support_head(const void* logits, void* grad, int V, int smin, float target, float fac, int lane,
                                                float& loss, float& pred) {
  float x[4], mx, lse;
  hl_row_stats<DT, 4>(logits, V, lane, x, mx, lse);
  // phi(h(target)): core/config.py:192-202, 240-253 (delta = 1)
  float h = (target < 0.0f ? -1.0f : 1.0f) * (sqrtf(fabsf(target) + 1.0f) - 1.0f) + 0.001f * target;
  h = fminf(fmaxf(h, (float)smin), (float)(smin + V - 1));
  const float lo_f = floorf(h), hi_f = ceilf(h);
  const float p_hi = h - lo_f;
  const int lo = (int)lo_f - smin, hi = (int)hi_f - smin;
  float l = 0.0f, ev = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + 64 * i;
    if (j < V) {
      const float logp = x[i] - mx - lse;
      const float p = expf(logp);
      float t = 0.0f;
      if (j == hi) t = p_hi;
      if (j == lo) t = 1.0f - p_hi;   // (lo == hi: the second scatter of the reference overwrites the first)
      l -= logp * t;
      ev += p * (float)(smin + j);
      if (grad != nullptr) hl_store<DT>(grad, j, fac * (p - t));
    }
  }
  loss = hl_wave_sum(l);
  const float v = hl_wave_sum(ev);    // inverse_scalar_transform: core/config.py:210-232
  const float a = fabsf(v);
  const float r = (sqrtf(1.0f + 4.0f * 0.001f * (a + 1.0f + 0.001f)) - 1.0f) / (2.0f * 0.001f);
  float out = (v < 0.0f ? -1.0f : 1.0f) * (r * r - 1.0f);
  pred = out != out ? 0.0f : out;
}

template <int DT>
__global__ __launch_bounds__(256) void k_head_losses(const void* __restrict__ vlog, long long vs, const void* __restrict__ rlog, long long rs,
                                                     const void* __restrict__ plog, long long ps, int rows, int V, int smin, int A,
                                                     const float* __restrict__ tv, long long tvs, const float* __restrict__ trw, long long trs,
                                                     const float* __restrict__ tp, long long tps, const float* __restrict__ weights,
                                                     float vc, float rc, float pc, void* dv, void* dr, void* dp,
                                                     float* __restrict__ losses, float* __restrict__ preds, int batch, long long tvk,
                                                     long long trk, long long tpk) {
  // rows = steps * batch: row R is position b = R % batch of inference k = R / batch of the unrolled step (steps == 1: one
  // inference, as hz_muzero_head_losses).  Logit rows are stacked inference by inference; the reward head has none for the
  // initial inference (its row of R is R - batch); targets are indexed [b][k] through their two strides; weights by b.
  const int lane = threadIdx.x & 63, R = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (R >= rows) return;
  constexpr long long ES = DT == HZ_F32 ? 4 : 2;
  const bool unrolled = rows != batch;
  const int k = R / batch, b = R - k * batch;
  const long long row = R, rrow = unrolled ? R - batch : R;
  const float wfac = weights[b] / (float)batch;
  auto at = [&](const void* p, long long r, long long stride) { return reinterpret_cast<const char*>(p) + r * stride * ES; };
  auto atw = [&](void* p, long long r, long long stride) { return p ? reinterpret_cast<char*>(p) + r * stride * ES : nullptr; };
  float vl, vpred, rl = 0.0f, rpred = 0.0f;
  hl_support_head<DT>(at(vlog, row, vs), atw(dv, row, V), V, smin, tv[(long long)b * tvs + (long long)k * tvk], wfac * vc, lane, vl, vpred);
  if (rlog != nullptr && !(unrolled && k == 0))
    hl_support_head<DT>(at(rlog, rrow, rs), atw(dr, rrow, V), V, smin, trw[(long long)b * trs + (long long)(unrolled ? k - 1 : 0) * trk], wfac * rc, lane, rl,
                        rpred);
  // policy: -(log_softmax . target); gradient softmax * sum(target) - target
  float x[1], mx, lse;
  hl_row_stats<DT, 1>(at(plog, row, ps), A, lane, x, mx, lse);
  const float t = lane < A ? tp[(long long)b * tps + (long long)k * tpk + lane] : 0.0f;
  const float tsum = hl_wave_sum(t);
  float pl = 0.0f;
  if (lane < A) {
    const float logp = x[0] - mx - lse;
    pl = -logp * t;
    if (dp != nullptr) hl_store<DT>(atw(dp, row, A), lane, wfac * pc * (expf(logp) * tsum - t));
  }
  pl = hl_wave_sum(pl);
  if (lane == 0) {
    float* o = losses + 4ll * row;
    o[0] = pl; o[1] = vl; o[2] = rl; o[3] = wfac * (pc * pl + vc * vl + rc * rl);
    preds[2ll * row] = vpred;
    preds[2ll * row + 1] = rpred;
  }
}

extern "C" int hz_muzero_head_losses(const void* value_logits, int64_t value_stride, const void* reward_logits, int64_t reward_stride,
                                     const void* policy_logits, int64_t policy_stride, int rows, int support_size, int support_min, int num_actions,
                                     int dtype, const float* target_value, int64_t target_value_stride, const float* target_reward,
                                     int64_t target_reward_stride, const float* target_policy, int64_t target_policy_stride, const float* weights,
                                     float value_coeff, float reward_coeff, float policy_coeff, void* d_value, void* d_reward, void* d_policy,
                                     float* losses, float* preds, void* stream) {
  return hz_muzero_unrolled_losses(value_logits, value_stride, reward_logits, reward_stride, policy_logits, policy_stride, rows, 1, support_size,
                                   support_min, num_actions, dtype, target_value, target_value_stride, 0, target_reward, target_reward_stride, 0,
                                   target_policy, target_policy_stride, 0, weights, value_coeff, reward_coeff, policy_coeff, d_value, d_reward,
                                   d_policy, losses, preds, stream);
}

extern "C" int hz_muzero_unrolled_losses(const void* value_logits, int64_t value_stride, const void* reward_logits, int64_t reward_stride,
                                         const void* policy_logits, int64_t policy_stride, int batch, int steps, int support_size, int support_min,
                                         int num_actions, int dtype, const float* target_value, int64_t target_value_stride,
                                         int64_t target_value_step_stride, const float* target_reward, int64_t target_reward_stride,
                                         int64_t target_reward_step_stride, const float* target_policy, int64_t target_policy_stride,
                                         int64_t target_policy_step_stride, const float* weights, float value_coeff, float reward_coeff,
                                         float policy_coeff, void* d_value, void* d_reward, void* d_policy, float* losses, float* preds,
                                         void* stream) {
  HZ_REQUIRE(batch >= 1 && steps >= 1 && (long long)batch * steps < (1ll << 30), "hz_muzero_unrolled_losses: batch=%d steps=%d", batch, steps);
  const int rows = batch * steps;
  HZ_REQUIRE(value_logits && policy_logits && target_value && target_policy && weights && losses && preds, "hz_muzero_head_losses: null pointer");
  HZ_REQUIRE(!reward_logits || target_reward, "hz_muzero_head_losses: reward logits without reward targets");
  HZ_REQUIRE(rows >= 1 && support_size >= 1 && support_size <= 256 && num_actions >= 1 && num_actions <= 64,
             "hz_muzero_head_losses: rows=%d support_size=%d (<= 256) num_actions=%d (<= 64)", rows, support_size, num_actions);
  HZ_REQUIRE(value_stride >= support_size && (!reward_logits || reward_stride >= support_size) && policy_stride >= num_actions,
             "hz_muzero_head_losses: a logits row stride below its width");
  HZ_REQUIRE(dtype == HZ_F32 || dtype == HZ_BF16 || dtype == HZ_F16, "hz_muzero_head_losses: dtype %d", dtype);
  const dim3 grid((rows + 3) / 4);
#define HZ_HL(DT)                                                                                                                      \
  hipLaunchKernelGGL(k_head_losses<DT>, grid, dim3(256), 0, (hipStream_t)stream, value_logits, (long long)value_stride, reward_logits,  \
                     (long long)reward_stride, policy_logits, (long long)policy_stride, rows, support_size, support_min, num_actions,  \
                     target_value, (long long)target_value_stride, target_reward, (long long)target_reward_stride, target_policy,      \
                     (long long)target_policy_stride, weights, value_coeff, reward_coeff, policy_coeff, d_value, d_reward, d_policy,   \
                     losses, preds, batch, (long long)target_value_step_stride, (long long)target_reward_step_stride,                  \
                     (long long)target_policy_step_stride)
  if (dtype == HZ_F32) HZ_HL(HZ_F32);
  else if (dtype == HZ_BF16) HZ_HL(HZ_BF16);
  else HZ_HL(HZ_F16);
#undef HZ_HL
  HZ_HIP(hipGetLastError());
  return 0;
}


// ------------------------------------------------------------------------------------------------ small glue of the unrolled step
// [state | one-hot(action)] rows, the dynamics net's input (config/hanabi_control/model.py:215-219: zeros + scatter_ + cat in PyTorch)
__global__ __launch_bounds__(256) void k_state_action_rows(const uint16_t* __restrict__ state, long long ss, const int64_t* __restrict__ action,
                                                           long long as, int H, int A, uint16_t* __restrict__ out, long long os, uint16_t one) {
  const int b = blockIdx.x;
  const uint16_t* s = state + (long long)b * ss;
  uint16_t* o = out + (long long)b * os;
  const int a = (int)action[(long long)b * as];
  for (int c = threadIdx.x; c < H + A; c += blockDim.x) o[c] = c < H ? s[c] : (c - H == a ? one : (uint16_t)0);
}

extern "C" int hz_state_action_rows(const void* state, int64_t state_stride, const int64_t* action, int64_t action_stride, int rows, int hidden,
                                    int num_actions, void* out, int64_t out_stride, int dtype, void* stream) {
  HZ_REQUIRE(state && action && out, "hz_state_action_rows: null pointer");
  HZ_REQUIRE(rows >= 1 && hidden >= 1 && num_actions >= 1 && state_stride >= hidden && out_stride >= hidden + num_actions && action_stride >= 1,
             "hz_state_action_rows: rows=%d hidden=%d num_actions=%d", rows, hidden, num_actions);
  HZ_REQUIRE(dtype == HZ_BF16 || dtype == HZ_F16, "hz_state_action_rows: dtype %d (HZ_BF16 or HZ_F16)", dtype);
  hipLaunchKernelGGL(k_state_action_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)state, (long long)state_stride, action,
                     (long long)action_stride, hidden, num_actions, (uint16_t*)out, (long long)out_stride,
                     (uint16_t)(dtype == HZ_BF16 ? 0x3f80u : 0x3c00u));
  HZ_HIP(hipGetLastError());
  return 0;
}

// out_i[r][c] = in_i[r][c] * round(g[r]) for up to three 16-bit matrices that share their rows' factor (the heads' logit gradients times the
// upstream gradient of a row's loss: PyTorch's `d * g.to(d.dtype)`, one rounding of the product); reward rows lag the others by `lag` rows
template <int DT>
__global__ __launch_bounds__(256) void k_scale_rows3(const uint16_t* __restrict__ a, int wa, const uint16_t* __restrict__ b, int wb,
                                                     const uint16_t* __restrict__ c, int wc, const float* __restrict__ g, int rows, int lag,
                                                     uint16_t* __restrict__ oa, uint16_t* __restrict__ ob, uint16_t* __restrict__ oc) {
  const int r = blockIdx.x;
  const float f = tr_requant<DT>(g[r]);
  for (int i = threadIdx.x; i < wa; i += blockDim.x) oa[(long long)r * wa + i] = tr_round<DT>(tr_load<DT>(a + (long long)r * wa + i) * f);
  for (int i = threadIdx.x; i < wc; i += blockDim.x) oc[(long long)r * wc + i] = tr_round<DT>(tr_load<DT>(c + (long long)r * wc + i) * f);
  if (b != nullptr && r >= lag)
    for (int i = threadIdx.x; i < wb; i += blockDim.x)
      ob[(long long)(r - lag) * wb + i] = tr_round<DT>(tr_load<DT>(b + (long long)(r - lag) * wb + i) * f);
}

extern "C" int hz_scale_rows3(const void* a, int width_a, const void* b, int width_b, const void* c, int width_c, const float* factor, int rows,
                              int lag, void* out_a, void* out_b, void* out_c, int dtype, void* stream) {
  HZ_REQUIRE(a && c && factor && out_a && out_c && (!b || out_b), "hz_scale_rows3: null pointer");
  HZ_REQUIRE(rows >= 1 && width_a >= 1 && width_c >= 1 && lag >= 0 && lag < rows, "hz_scale_rows3: rows=%d lag=%d", rows, lag);
  HZ_REQUIRE(dtype == HZ_BF16 || dtype == HZ_F16, "hz_scale_rows3: dtype %d (HZ_BF16 or HZ_F16)", dtype);
  if (dtype == HZ_BF16)
    hipLaunchKernelGGL(k_scale_rows3<HZ_BF16>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, width_a, (const uint16_t*)b, width_b,
                       (const uint16_t*)c, width_c, factor, rows, lag, (uint16_t*)out_a, (uint16_t*)out_b, (uint16_t*)out_c);
  else
    hipLaunchKernelGGL(k_scale_rows3<HZ_F16>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, width_a, (const uint16_t*)b, width_b,
                       (const uint16_t*)c, width_c, factor, rows, lag, (uint16_t*)out_a, (uint16_t*)out_b, (uint16_t*)out_c);
  HZ_HIP(hipGetLastError());
  return 0;
}
