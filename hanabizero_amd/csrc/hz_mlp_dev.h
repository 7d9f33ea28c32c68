// hz_mlp_dev.h -- device code of the fused recurrent-inference MLP (see hz_mlp.hip), shared by the stand-alone kernel
// and the persistent search kernel (hz_search.hip).  Everything is a template over the element format EL of weights and
// activations: ElBf16 (v_mfma_f32_16x16x32_bf16) or ElF16 (v_mfma_f32_16x16x32_f16 -- the reference's own autocast
// format, /root/reference/core/mcts.py:38-40); accumulation, bias, residual and ReLU are fp32 in both.
#pragma once
#include "hz_addrelu_dev.h"
#include "hz_common.h"
#include "hz_mlp.h"
#include "hz_tree.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct ElBf16 {
  typedef bf16x8 v8;
  static constexpr int code = HZ_BF16;
  static constexpr bool split = false;
  // the two elements of a packed pair -> fp32
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ float one(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
  // two fp32 -> packed pair: a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN kept)
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    const f32x2 f = {a, b};
    const bf16x2 h = __builtin_convertvector(f, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&h);
  }
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

struct ElF16 {
  typedef f16x8 v8;
  static constexpr int code = HZ_F16;
  static constexpr bool split = false;
  static __device__ __forceinline__ float one(uint16_t h) { return (float)*reinterpret_cast<const _Float16*>(&h); }
  static __device__ __forceinline__ float lo(uint32_t w) { return one((uint16_t)(w & 0xffffu)); }
  static __device__ __forceinline__ float hi(uint32_t w) { return one((uint16_t)(w >> 16)); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {  // v_cvt_pk_f16_f32: round-to-nearest-even
    const f32x2 f = {a, b};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    return *reinterpret_cast<const uint32_t*>(&h);
  }
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// Every fp32 number as a PAIR of fp16 -- hi = fp16(x), lo = fp16(x - hi): 22 bits of mantissa together -- for weights and
// activations alike, the product of two numbers as hi*hi + hi*lo + lo*hi (three MFMAs into the same fp32 accumulator; the
// dropped lo*lo term is 2^-22 of the product).  The image keeps the lo halves in a second plane `lo_plane` columns behind
// the hi halves, a wave's weight stream carries a tile's lo fragment behind its hi fragment, state rows come from and go
// to memory as fp32.  This is the build whose outputs stay within 1e-3 of the reference's fp32 nets (include/hz_mlp.h).
struct ElF16x2 : ElF16 {
  static constexpr int code = HZ_F16X2;
  static constexpr bool split = true;
  static __device__ __forceinline__ void halves(float a, float b, uint32_t& hi2, uint32_t& lo2) {  // two numbers -> packed hi pair, lo pair
    hi2 = pack(a, b);
    lo2 = pack(a - lo(hi2), b - hi(hi2));
  }
};

// inverse_scalar_transform of LDS rows of fp32 logits, one (row, head) pair per 32-lane half of the wave: lane l32
// owns logits [8*l32, 8*l32 + 8) (two ds_read_b128; Hanabi-Full's supports have 201 bins), max and sums over the half by
// DPP-modified v_max / v_add (hz_common.h::hz_wave_max has the reasons): butterflies inside the 16-lane rows, row_bcast:15
// into the odd rows, lanes 31 / 63 hold the halves' results.  V <= 256.  Same maths as hz_tree.hip support_to_scalar.
// (Tried: one logit per lane and trip, ceil(V / 32) trips -- fewer instructions for small V, 1.7 % slower end to end at
// V = 201: the loops.)
#define HZ_HALF32_REDUCE(OP)                                                                                          \
  asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"              \
               OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                            \
               OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                \
               OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                                     \
               OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"                                                   \
               : "+v"(v))
__device__ __forceinline__ float half32_max(float v) {
  HZ_HALF32_REDUCE("v_max_f32_dpp");
  const float lo = hz_readlane_f(v, 31), hi = hz_readlane_f(v, 63);
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float half32_sum(float v) {
  HZ_HALF32_REDUCE("v_add_f32_dpp");
  const float lo = hz_readlane_f(v, 31), hi = hz_readlane_f(v, 63);
  return (threadIdx.x & 32) ? hi : lo;
}
// `row`: the image row; the head's fp32 logits (HZ_MLP_F32_OUT) start at 16-bit column `off`, those from logit `split` on
// (a multiple of 32, so a lane's eight never straddle it) at column `off2`: include/hz_mlp.h::hz_mlp_header_t.
__device__ __forceinline__ float row32_support_to_scalar(const uint16_t* row, int off, int off2, int split, int V,
                                                         int support_min, int l32) {
  const int base = 8 * l32;
  float x[8];
  if (base < V) {
    const float4* p = reinterpret_cast<const float4*>(row + (base < split ? off + 2 * base : off2 + 2 * (base - split)));
    const float4 a = p[0], b = p[1];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (base + k < V) ? x[k] : -INFINITY;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) m = fmaxf(m, x[k]);
  m = half32_max(m);
  float se = 0.0f, sw = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float e = (base + k < V) ? __expf(x[k] - m) : 0.0f;
    se += e;
    sw += e * (float)(support_min + base + k);
  }
  se = half32_sum(se);
  sw = half32_sum(sw);
  // (a network output, tolerance 1e-3: the hardware reciprocal and square root -- 1 ulp -- instead of the correctly
  // rounded sequences the tree arithmetic needs)
  const float v = sw * __builtin_amdgcn_rcpf(se);
  const float eps = 0.001f;
  const float t = (__builtin_amdgcn_sqrtf(1.0f + 4.0f * eps * (fabsf(v) + 1.0f + eps)) - 1.0f) * (1.0f / (2.0f * eps));
  float out = t * t - 1.0f;
  if (v < 0.0f) out = -out;
  if (out != out) out = 0.0f;
  return out;
}
// policy logit a of an image row (fp32, HZ_MLP_F32_OUT), NaN -> 0 (core/mcts.py:48-49)
__device__ __forceinline__ float row_policy_logit(const uint16_t* row, int off_policy, int a) {
  const float x = reinterpret_cast<const float*>(row + off_policy)[a];
  return x != x ? 0.0f : x;
}

// Diagnostic build only (-DHZ_MLP_PROFILE, tools/mlp_profile.py): per-phase shader-cycle sums of workgroup 100.
#ifdef HZ_MLP_PROFILE
__device__ unsigned long long hz_mlp_prof[16 * 8];
__device__ unsigned long long hz_mlp_prof_pass[34];  // wave 0 of workgroup 100: s_memtime when it leaves job j's prologue
__device__ unsigned int hz_mlp_prof_tl[16 * 8 * 4];  // per wave, first 8 jobs: s_memtime (low word) past the barrier / k-loop start / k-loop end / epilogue end
extern "C" int hz_mlp_profile_read_timeline(unsigned int* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_mlp_prof_tl), sizeof(hz_mlp_prof_tl));
}
#define PROF_TL(J, K) do { if (blockIdx.x == 100 && lane == 0 && (J) < 8) prof_tl[(wave * 8 + (J)) * 4 + (K)] = (unsigned int)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int hz_mlp_profile_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_mlp_prof), sizeof(hz_mlp_prof));
}
extern "C" int hz_mlp_profile_read_passes(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_mlp_prof_pass), sizeof(hz_mlp_prof_pass));
}
#define PROF_NOW() __builtin_amdgcn_s_memtime()
#define PROF_ADD(var, t0) var += __builtin_amdgcn_s_memtime() - (t0)
#else
#define PROF_NOW() 0ull
#define PROF_ADD(var, t0) (void)(t0)
#define PROF_TL(J, K) (void)0
#endif

// Weight-fragment ring: 4 slots of one k-step each in every shape (8 slots for the workgroups of <= 8 waves, which have the
// registers for it, measured no faster: what in-flight depth can cover at a layer boundary is not what a boundary costs).

// The bias of two 16-column tiles (32 consecutive floats at a wave-uniform address) as the start values of two accumulators:
// lane L gets floats [4 (L >> 4), 4 (L >> 4) + 4) of each tile -- the four output columns its accumulator holds.  The 128 B
// come through the SCALAR cache (s_load into fixed scalar registers) and are handed out by four EXEC-masked rounds of
// moves, one per 16-lane row: no vector-memory instruction (a vector load of the same values would occupy the L1's return
// path as long as a whole weight fragment does), and nothing the compiler would wait for with vmcnt.  All lanes active.
__device__ __forceinline__ void hz_bias_start_values(const float* uniform_row, f32x4& t0, f32x4& t1) {
  float a0, a1, a2, a3, b0, b1, b2, b3;
  asm volatile(
      "s_load_dwordx16 s[68:83], %[p], 0x0\n\ts_load_dwordx16 s[84:99], %[p], 0x40\n\ts_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 exec_hi, 0\n\ts_mov_b32 exec_lo, 0xffff\n\t"
      "v_mov_b32 %[a0], s68\n\tv_mov_b32 %[a1], s69\n\tv_mov_b32 %[a2], s70\n\tv_mov_b32 %[a3], s71\n\t"
      "v_mov_b32 %[b0], s84\n\tv_mov_b32 %[b1], s85\n\tv_mov_b32 %[b2], s86\n\tv_mov_b32 %[b3], s87\n\t"
      "s_mov_b32 exec_lo, 0xffff0000\n\t"
      "v_mov_b32 %[a0], s72\n\tv_mov_b32 %[a1], s73\n\tv_mov_b32 %[a2], s74\n\tv_mov_b32 %[a3], s75\n\t"
      "v_mov_b32 %[b0], s88\n\tv_mov_b32 %[b1], s89\n\tv_mov_b32 %[b2], s90\n\tv_mov_b32 %[b3], s91\n\t"
      "s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, 0xffff\n\t"
      "v_mov_b32 %[a0], s76\n\tv_mov_b32 %[a1], s77\n\tv_mov_b32 %[a2], s78\n\tv_mov_b32 %[a3], s79\n\t"
      "v_mov_b32 %[b0], s92\n\tv_mov_b32 %[b1], s93\n\tv_mov_b32 %[b2], s94\n\tv_mov_b32 %[b3], s95\n\t"
      "s_mov_b32 exec_hi, 0xffff0000\n\t"
      "v_mov_b32 %[a0], s80\n\tv_mov_b32 %[a1], s81\n\tv_mov_b32 %[a2], s82\n\tv_mov_b32 %[a3], s83\n\t"
      "v_mov_b32 %[b0], s96\n\tv_mov_b32 %[b1], s97\n\tv_mov_b32 %[b2], s98\n\tv_mov_b32 %[b3], s99\n\t"
      "s_mov_b64 exec, -1"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
      : [p] "s"(uniform_row)
      : "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85",
        "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
  t0 = (f32x4){a0, a1, a2, a3};
  t1 = (f32x4){b0, b1, b2, b3};
}

// The weight ring of the hand-scheduled k-loop (16 waves x 2 tiles): 4 slots x 2 fragments of 16 B per lane in FIXED
// registers v[96:127], which no compiler-generated instruction ever touches: kernels that inline that k-loop are compiled
// with __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS))) -- the register allocator stays below v96 -- and every asm statement
// that names ring registers lists them as clobbers, which is what makes the kernel descriptor ask for all 128.  Loads into
// them stay in flight across compiler-generated code (epilogues, barriers, the next job's decode); had they been C++ values,
// a register copy at a loop back-edge or a spill would have read them before the data arrived (both were observed).
#define HZ_ASMK_VGPRS 96
#define HZ_W00 "v[96:99]"
#define HZ_W01 "v[100:103]"
#define HZ_W10 "v[104:107]"
#define HZ_W11 "v[108:111]"
#define HZ_W20 "v[112:115]"
#define HZ_W21 "v[116:119]"
#define HZ_W30 "v[120:123]"
#define HZ_W31 "v[124:127]"
#define HZ_RING_CLOBBER                                                                                                  \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
      "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// Issue priority of this wave for the next few k-steps (s_setprio takes an immediate).  The sequencer serves the oldest wave of
// a SIMD first; behind a workgroup barrier per layer that made the oldest waves idle at the barrier while the youngest finished
// on a memory pipe they could not fill alone, and rotating the priority over the four age groups every four k-steps was worth
// +1.2 % moves/s at 16 rows per workgroup (r01; the compiler-scheduled shapes still do that).  The 16 x 2 shape has no barrier
// between its passes any more and runs with the natural order made explicit (oldest group first, see the job loop).
__device__ __forceinline__ void hz_rotate_prio(int group_plus_phase) {
  switch (group_plus_phase & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// The body, called by every wave of the workgroup: NW waves, each producing NT 16-column tiles per job (NW x NT = 4 x 4
// for the stand-alone kernel, 16 x 2 inside the persistent search kernel, whose 16 waves also own one tree each).
// `lds`: the row image, MT * row_stride elements.
// The input rows of a workgroup: gathered here from state_src (STAGE_GATHER), or handed over by the caller's waves in
// registers (STAGE_REGS: wave w holds rows w, 16 + w, ..; lane l its l-th 16-B chunk in row_frag[rt].v[l / 64]; NW == 16) and
// written into the image here, after the weight ring has been started -- the rows' load latency hides under it.
// A wave that finds a counter short sleeps HZ_POLL_SLEEP x 64 cycles before it looks again.  Long on purpose: 1 (look again at
// once) costs 3.5 % moves/s at 4096 envs against 31..48, 63 is worse again; s_wakeup from every signalling wave (sleepers look
// again immediately) is worse than not waking them (2.13 M against 2.16 M moves/s): what the waiting waves issue -- LDS reads,
// branches -- is taken from the waves everybody is waiting for, and the chip runs this kernel at the power limit.
#define HZ_POLL_SLEEP "40"
#define HZ_POLL_TRIES (1u << 16)  // (then a wave gives up on a counter: a job table that breaks the contract must not hang the GPU)
// ... and says so: waits given up since the library was loaded, per translation unit (hz_mlp_poll_giveups adds them up); anything
// but 0 means results that cannot be trusted
static __device__ unsigned int hz_poll_giveups_dev;
#define HZ_POLL_GIVEUP(TRIES_LEFT) do { if ((TRIES_LEFT) == 0u && lane == 0) atomicAdd(&hz_poll_giveups_dev, 1u); } while (0)
enum { STAGE_GATHER = 0, STAGE_REGS = 1, STAGE_REGS_HALF = 2 };
// register i (of 4) of a wave's copy of its job entries: lane 8 * (j % 8) + f holds field f of job j = 8 i + j % 8
__device__ __forceinline__ int hz_mlp_job_entries(const hz_mlp_job_t* jobs, int n_jobs, int num_waves, int wave, int lane, int i) {
  const int jl = 8 * i + (lane >> 3);
  return jl < n_jobs ? reinterpret_cast<const int*>(jobs)[((size_t)jl * num_waves + wave) * 8 + (lane & 7)] : 0;
}
struct RowFrag {
  uint4 v[2];
};
// FINAL = false: stop after the last layer (logits stay in the image; the caller's waves read them there).
// BW = false: HZ_MLP_BLOCKWISE jobs wait at a workgroup barrier instead (the stronger condition; for the kernel that has no
// registers to spare for the counters' address).
// (Tried, r03: the ring primed across the caller's tree phase -- the next inference's first fragments requested at the end of the
// previous one, the prologue below skipped: -0.4 % moves/s at 4096 envs, -0.5 % at 8192 in an A/B on one box.  The prologue's
// round trip is already hidden behind the staging of the rows, and the extra requests are in the way of the tree phase's.)
template <class EL, int RT, int NW, int NT, int STAGE = STAGE_GATHER, bool FINAL = true, bool BW = true>
__device__ __forceinline__ void mlp_body(
    const hz_mlp_header_t& H, const hz_mlp_job_t* __restrict__ jobs, const uint16_t* __restrict__ wstream,
    const float* __restrict__ bias, const float* __restrict__ act_tab, const uint16_t* __restrict__ state_src,
    long long state_row_stride, const int32_t* __restrict__ plane_index, long long plane_stride,
    const int32_t* __restrict__ actions, uint16_t* __restrict__ hidden_out, float* __restrict__ out_reward,
    float* __restrict__ out_value, float* __restrict__ out_policy, int n_rows, uint16_t* lds, int row0,
    const RowFrag* row_frag, const int* jv_cached = nullptr, const uint16_t* __restrict__ state_res = nullptr,
    long long state_res_stride = 0) {
  typedef typename EL::v8 v8;
  constexpr int NTHR = 64 * NW;
  constexpr bool PRESTAGED = STAGE != STAGE_GATHER;
  constexpr int MT = 16 * RT;
  constexpr bool SPLIT = EL::split;   // ElF16x2: hi / lo planes, three MFMAs per fragment pair, fp32 state rows
  constexpr int WP = SPLIT ? 2 : 1;   // fragments per tile and k-step (weights) / per row tile and k-step (activations)
  static_assert(!SPLIT || STAGE != STAGE_REGS_HALF, "the fp16-pair build: stand-alone kernel or one tree per wave; compiler-scheduled k-loop");
  const int LP = H.lo_plane;          // SPLIT: columns between an element's hi half and its lo half
  // (opaque to the optimiser: inside a caller's loop -- the simulations of the persistent search kernel -- everything
  // derived from the thread index would otherwise be hoisted out of that loop, kept alive across the other phases and,
  // at 128 registers per lane, spilled to scratch: a memory round trip per use instead of a few ALU instructions)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: job fields stay in SGPRs, branches are scalar
  const int rs = H.row_stride;
  (void)NTHR;

#ifdef HZ_MLP_PROFILE
  unsigned int* prof_tl = reinterpret_cast<unsigned int*>(lds + (size_t)MT * H.row_stride);  // (the launcher adds 2 KiB behind the image)
#endif
  unsigned long long p_loop = 0, p_epi = 0, p_bar = 0, p_pre = 0;
  const unsigned long long p_t0 = PROF_NOW();
  (void)p_loop; (void)p_epi; (void)p_bar; (void)p_pre; (void)p_t0;
  // The parent hidden states (the gather of core/mcts.py:31-36) are one dependent pair of loads away: plane index,
  // then the row.  Issue the index loads first, the weight ring next (it does not depend on the inputs and keeps the
  // memory pipe busy meanwhile), then all row loads of this thread at once: two latencies in total, not two per trip.
  // this wave's job entries: 8 dwords per job, lane 8*(j % 8) + f of register j / 8 holds field f of job j.  One vector
  // load now instead of one scalar load (a dependent L2 round trip in front of every job) inside the job loop.
  // (the persistent search kernel loads them once per launch -- hz_mlp_job_entries -- and hands them in: jv_cached)
  int jv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) jv[i] = jv_cached ? jv_cached[i] : hz_mlp_job_entries(jobs, H.n_jobs, NW, wave, lane, i);
  const int chunks = H.in_width / 8;
  const int n_stage = MT * chunks;
  constexpr int SU = 4;  // rows-chunks per thread per trip (hidden = 512, 16 rows: exactly one trip)
  long long plane0[SU];
#pragma unroll
  for (int u = 0; u < SU; ++u) {
    const int i = tid + NTHR * u;
    const int row = row0 + i / chunks;
    plane0[u] = (!PRESTAGED && plane_index && i < n_stage && row < n_rows) ? (long long)plane_index[row] * plane_stride : 0;
  }
  __builtin_amdgcn_sched_barrier(0);
  // this wave's weight stream: a uniform base (scalar registers) + a per-lane byte offset that walks the stream
  const char* wbase = reinterpret_cast<const char*>(wstream + H.wave_stream_off[wave]);
  const unsigned int kss = (unsigned int)(H.kstep_stride * 2);  // bytes between consecutive k-steps of this wave's stream
  // Prefetch distance in k-steps.  16-row shapes: RING - 1 (2 costs 1.7 % moves/s at 4096 envs).  32-row shape: 2 -- one
  // ring slot stays spare, so the refill of a slot does not have to wait for the four MFMAs that have just read it
  // (3: -2.1 % moves/s at 8192 envs, 1: -1.6 %; A/B on one box, tools/ab_bench.sh).
  // (the fp16-pair build inside the persistent search kernel -- two fragments per tile and step, 128 registers per lane, sixteen
  // wavefronts to hide latency with: a ring of two)
#ifdef HZ_NO_ASMK   // (diagnostic build: the 16 x 2 shape on the compiler-scheduled k-loop, as the fp16-pair build runs it)
  constexpr bool NOASM = true;
#else
  constexpr bool NOASM = false;
#endif
  constexpr int RING = ((SPLIT || NOASM) && NT == 2) ? 2 : 4;
  constexpr int PF = ((SPLIT || NOASM) && NT == 2) ? 1 : (RT == 1 ? RING - 1 : RING - 2);
  // NT == 2 (the 16 x 2 shape of the persistent search kernel): the k-loop is hand-scheduled assembly (below) and its
  // loads are invisible to the compiler; the other shapes keep the compiler-scheduled loop.
  constexpr bool ASMK = NT == 2 && RING == 4 && !SPLIT && !NOASM;
  v8 wf[ASMK ? 1 : RING][NT * WP];                    // !ASMK: the ring as C++ values (ASMK: the fixed registers above)
  unsigned int voff = (unsigned int)lane * 16u;  // ASMK: byte offset of the next fragment this lane requests
  long long gstep = 0;                           // !ASMK: k-steps of this wave's stream consumed so far
#define wp(k, t) (*reinterpret_cast<const v8*>(wbase + (long long)(k) * kss + ((t) * 1024u + (unsigned int)lane * 16u)))
#define HZ_LD(WN0, WN1) \
  "global_load_dwordx4 " WN0 ", %[voff], %[sa]\n\tglobal_load_dwordx4 " WN1 ", %[voff], %[sa] offset:1024\n\tv_add_u32 %[voff], %[kss], %[voff]\n\t"
  if constexpr (ASMK) {
    static_assert(!ASMK || PF == 3 || PF == 2, "ring of 4 k-steps");
    if constexpr (PF == 3)
      asm volatile(HZ_LD(HZ_W00, HZ_W01) HZ_LD(HZ_W10, HZ_W11) HZ_LD(HZ_W20, HZ_W21)
                   : [voff] "+v"(voff) : [sa] "s"(wbase), [kss] "s"(kss) : HZ_RING_CLOBBER);
    else
      asm volatile(HZ_LD(HZ_W00, HZ_W01) HZ_LD(HZ_W10, HZ_W11)
                   : [voff] "+v"(voff) : [sa] "s"(wbase), [kss] "s"(kss) : HZ_RING_CLOBBER);
  } else {
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int t = 0; t < NT * WP; ++t) wf[d][t] = wp(d, t);
  }
  __builtin_amdgcn_sched_barrier(0);

  // stage the states into the image; rows past N read as zero
  if constexpr (SPLIT) {  // fp32 rows (strides in fp32 elements) -> hi plane, lo plane
    const float* src32 = reinterpret_cast<const float*>(state_src);
    for (int base = 0; base < n_stage; base += NTHR * SU) {
      float4 v[SU][2];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = base + tid + NTHR * u;
        const int r = i / chunks, c = i % chunks;
        const int row = row0 + r;
        v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_stage && row < n_rows) {
          const long long plane = base == 0 ? plane0[u] : (plane_index ? (long long)plane_index[row] * plane_stride : 0);
          const float4* p = reinterpret_cast<const float4*>(src32 + plane + (long long)row * state_row_stride + c * 8);
          v[u][0] = p[0];
          v[u][1] = p[1];
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = base + tid + NTHR * u;
        if (i < n_stage) {
          uint4 h, l;
          EL::halves(v[u][0].x, v[u][0].y, h.x, l.x);
          EL::halves(v[u][0].z, v[u][0].w, h.y, l.y);
          EL::halves(v[u][1].x, v[u][1].y, h.z, l.z);
          EL::halves(v[u][1].z, v[u][1].w, h.w, l.w);
          uint16_t* d = lds + (size_t)(i / chunks) * rs + H.state_off + (i % chunks) * 8;
          *reinterpret_cast<uint4*>(d) = h;
          *reinterpret_cast<uint4*>(d + LP) = l;
        }
      }
    }
  }
  for (int base = 0; !SPLIT && !PRESTAGED && base < n_stage; base += NTHR * SU) {
    uint4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = base + tid + NTHR * u;
      const int r = i / chunks, c = i % chunks;
      const int row = row0 + r;
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      if (i < n_stage && row < n_rows) {
        const long long plane = base == 0 ? plane0[u] : (plane_index ? (long long)plane_index[row] * plane_stride : 0);
        v[u] = *reinterpret_cast<const uint4*>(state_src + plane + (long long)row * state_row_stride + c * 8);
        if (state_res != nullptr) {  // the input rows are relu(state_src + state_res): hz_add_relu's arithmetic, applied on the way in
          const uint4 b = *reinterpret_cast<const uint4*>(state_res + (long long)row * state_res_stride + c * 8);
          if (EL::code == HZ_BF16) {
            v[u].x = hz_add_relu_word<HZ_BF16>(v[u].x, b.x); v[u].y = hz_add_relu_word<HZ_BF16>(v[u].y, b.y);
            v[u].z = hz_add_relu_word<HZ_BF16>(v[u].z, b.z); v[u].w = hz_add_relu_word<HZ_BF16>(v[u].w, b.w);
          } else {
            v[u].x = hz_add_relu_word<HZ_F16>(v[u].x, b.x); v[u].y = hz_add_relu_word<HZ_F16>(v[u].y, b.y);
            v[u].z = hz_add_relu_word<HZ_F16>(v[u].z, b.z); v[u].w = hz_add_relu_word<HZ_F16>(v[u].w, b.w);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = base + tid + NTHR * u;
      if (i < n_stage) *reinterpret_cast<uint4*>(lds + (size_t)(i / chunks) * rs + H.state_off + (i % chunks) * 8) = v[u];
    }
  }
  if (STAGE == STAGE_REGS) {
    static_assert(STAGE != STAGE_REGS || NW == 16, "wave w holds rows w, 16 + w, ...");
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (SPLIT) {  // the wave's row in fp32: lane l holds elements [4 (l + 64 u), + 4)
          if ((lane + 64 * u) * 4 < H.in_width) {
            const uint4 f = row_frag[rt].v[u];
            uint2 h, l;
            EL::halves(__uint_as_float(f.x), __uint_as_float(f.y), h.x, l.x);
            EL::halves(__uint_as_float(f.z), __uint_as_float(f.w), h.y, l.y);
            uint16_t* d = lds + (size_t)(16 * rt + wave) * rs + H.state_off + (lane + 64 * u) * 4;
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + LP) = l;
          }
        } else if (lane + 64 * u < chunks) {
          *reinterpret_cast<uint4*>(lds + (size_t)(16 * rt + wave) * rs + H.state_off + (lane + 64 * u) * 8) = row_frag[rt].v[u];
        }
      }
  }
  if (STAGE == STAGE_REGS_HALF) {  // the first 8 RT waves hold two rows each: wave w row w in lanes 0-31 and row 8 RT + w in
    static_assert(STAGE != STAGE_REGS_HALF || NW == 16, "two rows per tree-owning wave");  // lanes 32-63; lane l of a half its chunks l, 32 + l
    const int hl = lane & 31, hrow = 8 * RT * (lane >> 5) + wave;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (wave < 8 * RT && hl + 32 * u < chunks)
        *reinterpret_cast<uint4*>(lds + (size_t)hrow * rs + H.state_off + (hl + 32 * u) * 8) = row_frag[0].v[u];
  }
  // arrival counters of the HZ_MLP_SIGNAL jobs (include/hz_mlp.h): four per job, in the padding behind image row `job`
  if (ASMK && tid < 64) reinterpret_cast<unsigned int*>(lds + (size_t)(tid >> 2) * rs + (rs - 8))[tid & 3] = 0u;
  __syncthreads();
  int act[RT];  // (read after the barrier: in STAGE_REGS mode the caller's waves have only just written them)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = row0 + 16 * rt + (lane & 15);
    int a = row < n_rows ? actions[row] : 0;
    act[rt] = (a < 0 || a >= H.num_actions) ? 0 : a;
  }

  const unsigned long long p_staged = PROF_NOW();
  (void)p_staged;
  auto job_of = [&](int j) {
    const int jsel = j >> 3, jb = (j & 7) * 8;
    const int jr = jsel == 0 ? jv[0] : (jsel == 1 ? jv[1] : (jsel == 2 ? jv[2] : jv[3]));
    hz_mlp_job_t J;
    J.ks = __builtin_amdgcn_readlane(jr, jb + 0);
    J.src_off = __builtin_amdgcn_readlane(jr, jb + 1);
    J.dst_off = __builtin_amdgcn_readlane(jr, jb + 2);
    J.res_off = __builtin_amdgcn_readlane(jr, jb + 3);
    J.bias_off = __builtin_amdgcn_readlane(jr, jb + 4);
    J.flags = __builtin_amdgcn_readlane(jr, jb + 5);
    J.producer = __builtin_amdgcn_readlane(jr, jb + 7);
    return J;
  };
  // The accumulators start from the epilogue's additive term -- bias (+ the action's column of the first dynamics layer): one
  // row of the action table per batch row (row num_actions = the bias alone); the MFMAs add the products on top.
  // The bias row is the same for every batch row, so a lane needs 4 of each tile's 16 values, chosen by its column quad:
  // hz_bias_start_values (scalar cache + EXEC-masked moves; ASMK shapes) -- vector loads of them occupied the L1's 64 B/clk
  // return path as long as whole weight fragments do (measured: 4 % of the inference at 16 rows per workgroup, 9 % at 32).
  // Only the action-row job (the first dynamics layer) loads per-lane rows.
  hz_mlp_job_t J = job_of(0);
  for (int j = 0; j < H.n_jobs; ++j) {
    // this lane's place in the MFMA fragments, derived afresh per job from an opaque copy of the lane index: kept across
    // the jobs the derived LDS addresses are spilled at 128 registers per lane, and their reloads (scratch = vector
    // memory) would drain the weight ring in front of every k-loop
    int lane_j = lane;
    asm volatile("" : "+v"(lane_j));
    const int r0 = lane_j & 15, kq = (lane_j >> 4) * 8, c4 = 4 * (lane_j >> 4);
    if (j > 0) J = job_of(j);
    f32x4 acc[NT][RT];
    if (J.ks != 0) {
      if (J.flags & HZ_MLP_ACTION_ROW) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float* p = act_tab + (size_t)act[rt] * H.action_table_stride + J.bias_off + 16 * t + c4;
            if constexpr (ASMK)  // (a load the compiler knows of inside this loop makes it wait with vmcnt(0) -- for the weight ring
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(acc[t][rt]) : "v"(p));  //  too -- at EVERY job's start)
            else
              acc[t][rt] = *reinterpret_cast<const f32x4*>(p);
          }
        if constexpr (ASMK) {  // ... so this one job waits for its rows itself (it is the first: nothing else is in flight yet)
          if constexpr (RT == 1)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[1][0]));
          else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        }
      } else {
        const float* brow = act_tab + (size_t)H.num_actions * H.action_table_stride + J.bias_off;  // wave-uniform
        if constexpr (NT == 2) {
          f32x4 sv[2];
          hz_bias_start_values(brow, sv[0], sv[1]);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[t][rt] = sv[t];
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[t][rt] = *reinterpret_cast<const f32x4*>(brow + 16 * t + c4);
        }
      }
    }
    const unsigned long long p_j0 = PROF_NOW();
    if ((J.flags & HZ_MLP_BARRIER) || (!(BW && ASMK) && (J.flags & (HZ_MLP_BLOCKWISE | HZ_MLP_WAITS))))
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: loads stay in flight
    if constexpr (BW && ASMK) {
      if (J.flags & HZ_MLP_WAITS) {  // this job's own dependencies instead of a barrier (include/hz_mlp.h; hanabizero_amd/mlp_sync.py)
        unsigned int tok = (unsigned int)J.producer;
        const unsigned int img = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)lds;
        unsigned int pc2 = HZ_POLL_TRIES;  // (tries for all of this job's tokens together)
        for (int k = (J.flags >> 8) & 7; k > 0; --k, tok >>= 8) {
          unsigned int fa2 = img + 2u * (((tok >> 4) & 15u) * (unsigned int)rs + (unsigned int)(rs - 8)) + 4u * ((tok >> 2) & 3u);
          unsigned int want = (tok & 3u) + 1u, pt2, ps2;
          asm volatile("2:\n\tds_read_b32 %[pt], %[fa]\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %[ps], %[pt]\n\ts_nop 3\n\t"
                       "s_cmp_ge_u32 %[ps], %[want]\n\ts_cbranch_scc1 3f\n\ts_cmp_eq_u32 %[pc], 0\n\ts_cbranch_scc1 3f\n\t"
                       "s_sleep " HZ_POLL_SLEEP "\n\ts_sub_u32 %[pc], %[pc], 1\n\ts_branch 2b\n\t3:"
                       : [pt] "=&v"(pt2), [ps] "=&s"(ps2), [pc] "+s"(pc2)
                       : [fa] "v"(fa2), [want] "s"(want)
                       : "memory", "scc");
        }
        HZ_POLL_GIVEUP(pc2);
      }
    }
    if (J.flags & HZ_MLP_STORE_HIDDEN) {
      const int chunks = H.hidden / 8;
      for (int i = tid; i < MT * chunks; i += NTHR) {
        const int r = i / chunks, c = i % chunks;
        if (row0 + r >= n_rows) continue;
        const uint16_t* s = lds + (size_t)r * rs + H.hidden_off + c * 8;
        if constexpr (SPLIT) {  // hi + lo -> fp32 rows of `hidden` elements
          const uint4 h = *reinterpret_cast<const uint4*>(s), l = *reinterpret_cast<const uint4*>(s + LP);
          float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(hidden_out) + (size_t)(row0 + r) * H.hidden + c * 8);
          d[0] = make_float4(EL::lo(h.x) + EL::lo(l.x), EL::hi(h.x) + EL::hi(l.x), EL::lo(h.y) + EL::lo(l.y), EL::hi(h.y) + EL::hi(l.y));
          d[1] = make_float4(EL::lo(h.z) + EL::lo(l.z), EL::hi(h.z) + EL::hi(l.z), EL::lo(h.w) + EL::lo(l.w), EL::hi(h.w) + EL::hi(l.w));
        } else {
          *reinterpret_cast<uint4*>(hidden_out + (size_t)(row0 + r) * H.hidden + c * 8) = *reinterpret_cast<const uint4*>(s);
        }
      }
    }
    PROF_ADD(p_bar, p_j0);
#ifdef HZ_MLP_PROFILE
    if (blockIdx.x == 100 && wave == 0 && lane == 0 && j < 32) hz_mlp_prof_pass[j] = PROF_NOW();  // past job j's barrier
#endif
    PROF_TL(j, 0);
    if (J.ks == 0) continue;
    const unsigned long long p_j1 = PROF_NOW();
    const uint16_t* src = lds + (size_t)r0 * rs + J.src_off + kq;
    PROF_ADD(p_pre, p_j1);
    const unsigned long long p_j2 = PROF_NOW();
    PROF_TL(j, 1);
    unsigned int pc_left = 1u;  // (ASMK: tries the blockwise k-loop had left when it ended)
    if constexpr (ASMK) {
      // ---- the k-loop, hand-scheduled (the compiler's own schedule of the same loop drained the weight ring at the start
      // of every job and of every 8 k-steps: it packs address arithmetic into ring registers and serialises the refills
      // behind the MFMAs -- 80 GB/s per CU where tools/l2_stream_bench.hip reaches 125 with the same access pattern).
      // Per k-step: request the fragments PF k-steps ahead (ring slot (u + PF) % 4) and the activation fragments BQPF
      // k-steps ahead, wait until this step's own fragments have arrived (loads and LDS reads return in order: at most
      // 2 PF loads / the younger reads may remain), issue its MFMAs.  Blocks of 4 k-steps (ring positions are static),
      // J.ks / 4 of them; the last block requests no activation fragments past the K range; the weight requests run on
      // into the next job's first k-steps (the streams are contiguous) -- or into the zero padding behind the stream.
      // The ring lives in the fixed registers v[96:127] (see HZ_W00 ..): in flight across everything the compiler generates.
      unsigned int cnt = (unsigned int)J.ks >> 2;
      unsigned int la0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)src;
      // issue priority.  Full-width layers: oldest age group first -- the groups then finish a layer in the order of their columns,
      // which is the order in which a blockwise consumer wants the blocks (with a barrier behind every layer the priorities
      // rotated over the groups, job by job; with the counters a fixed order measures the same or better).  The staggered
      // passes (HZ_MLP_WAITS): the younger half first -- it runs the longer chain of dependent layers there (model.py puts
      // reward and actor heads on waves 8-15) and finished 3.5 k cycles behind the other half: +0.9 % moves/s at 4096 envs,
      // +1.9 % at 8192 (A/B on one box)
      hz_rotate_prio((J.flags & HZ_MLP_WAITS) ? (wave >> 2) : 3 - (wave >> 2));
      const bool blockwise = BW && (J.flags & HZ_MLP_BLOCKWISE);
      const unsigned int last = (unsigned int)J.flags & HZ_MLP_LAST;
      unsigned int pc = HZ_POLL_TRIES;
      unsigned int fa = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)(lds + (size_t)J.producer * rs + (rs - 8));
      unsigned int pt, ps;
      if constexpr (RT == 1) {
        // one row tile: weights 3 k-steps ahead (4 ring slots), activation fragments 3 k-steps ahead in 4 slots
        v8 b0, b1, b2, b3;
#define HZ_RD1(B, OFF) "ds_read_b128 %[" B "], %[la0] offset:" #OFF "\n\t"
#define HZ_K1(W0, W1, WN0, WN1, B, RD, LGKM)                                               \
  HZ_LD(WN0, WN1) RD "s_waitcnt vmcnt(6) lgkmcnt(" LGKM ")\n\t"                            \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a0], " W0 ", %[" B "], %[a0]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a1], " W1 ", %[" B "], %[a1]\n\t"
#define HZ_K1N(W0, W1, B, VM, LGKM)                                                       \
  "s_waitcnt vmcnt(" VM ") lgkmcnt(" LGKM ")\n\t"                                          \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a0], " W0 ", %[" B "], %[a0]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a1], " W1 ", %[" B "], %[a1]\n\t"
#define HZ_K1_BODY()                                                                                            \
        asm volatile(                                                                                           \
            HZ_RD1("b0", 0) HZ_RD1("b1", 64) HZ_RD1("b2", 128)                                                    \
            "1:\n\t"                                                                                            \
            HZ_K1(HZ_W00, HZ_W01, HZ_W30, HZ_W31, "b0", HZ_RD1("b3", 192), "3")                                    \
            HZ_K1(HZ_W10, HZ_W11, HZ_W00, HZ_W01, "b1", HZ_RD1("b0", 256), "3")                                    \
            HZ_K1(HZ_W20, HZ_W21, HZ_W10, HZ_W11, "b2", HZ_RD1("b1", 320), "3")                                    \
            HZ_K1(HZ_W30, HZ_W31, HZ_W20, HZ_W21, "b3", HZ_RD1("b2", 384), "3")                                    \
            "v_add_u32 %[la0], 0x100, %[la0]\n\t"                                                                \
            "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 1\n\ts_cbranch_scc1 1b\n\t"                     \
            HZ_K1(HZ_W00, HZ_W01, HZ_W30, HZ_W31, "b0", HZ_RD1("b3", 192), "3")                                    \
            "s_cmp_lg_u32 %[last], 0\n\ts_cbranch_scc1 4f\n\t"                                                  \
            HZ_K1(HZ_W10, HZ_W11, HZ_W00, HZ_W01, "b1", "", "2")                                                   \
            HZ_K1(HZ_W20, HZ_W21, HZ_W10, HZ_W11, "b2", "", "1")                                                   \
            HZ_K1(HZ_W30, HZ_W31, HZ_W20, HZ_W21, "b3", "", "0")                                                   \
            "s_branch 5f\n\t4:\n\t"  /* this wave's last job of the inference: nothing to request ahead, nothing left in flight */ \
            HZ_K1N(HZ_W10, HZ_W11, "b1", "4", "2") HZ_K1N(HZ_W20, HZ_W21, "b2", "2", "1") HZ_K1N(HZ_W30, HZ_W31, "b3", "0", "0") \
            "5:\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                                                                \
            : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [a0] "+v"(acc[0][0]), [a1] "+v"(acc[1][0]), \
              [voff] "+v"(voff), [la0] "+v"(la0), [cnt] "+s"(cnt)                                                  \
            : [sa] "s"(wbase), [kss] "s"(kss), [last] "s"(last)                                                   \
            : "memory", "scc", HZ_RING_CLOBBER)
        // blockwise: per block of 4 k-steps wait for the producers of its 128 input columns, then prime the activation
        // fragments and run the block without reading past it
#define HZ_POLL()                                                                                                       \
            "2:\n\tds_read_b32 %[pt], %[fa]\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %[ps], %[pt]\n\ts_nop 3\n\t"        \
            "s_cmp_ge_u32 %[ps], 4\n\ts_cbranch_scc1 3f\n\ts_cmp_eq_u32 %[pc], 0\n\ts_cbranch_scc1 3f\n\t"  /* (no tries left: not this block either) */ \
            "s_sleep " HZ_POLL_SLEEP "\n\ts_sub_u32 %[pc], %[pc], 1\n\ts_branch 2b\n\t3:\n\t"
#define HZ_K1B_BODY()                                                                                           \
        asm volatile(                                                                                           \
            "1:\n\t" HZ_POLL()                                                                                   \
            HZ_RD1("b0", 0) HZ_RD1("b1", 64) HZ_RD1("b2", 128)                                                    \
            HZ_K1(HZ_W00, HZ_W01, HZ_W30, HZ_W31, "b0", HZ_RD1("b3", 192), "3")                                    \
            HZ_K1(HZ_W10, HZ_W11, HZ_W00, HZ_W01, "b1", "", "2")                                                   \
            HZ_K1(HZ_W20, HZ_W21, HZ_W10, HZ_W11, "b2", "", "1")                                                   \
            HZ_K1(HZ_W30, HZ_W31, HZ_W20, HZ_W21, "b3", "", "0")                                                   \
            "v_add_u32 %[la0], 0x100, %[la0]\n\tv_add_u32 %[fa], 4, %[fa]\n\t"                                   \
            "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t"                     \
            "s_nop 7\n\ts_nop 7\n\ts_nop 7"                                                                      \
            : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [a0] "+v"(acc[0][0]), [a1] "+v"(acc[1][0]), \
              [voff] "+v"(voff), [la0] "+v"(la0), [cnt] "+s"(cnt), [fa] "+v"(fa), [pt] "=&v"(pt), [ps] "=&s"(ps),   \
              [pc] "+s"(pc)                                                                                       \
            : [sa] "s"(wbase), [kss] "s"(kss)                                                                     \
            : "memory", "scc", HZ_RING_CLOBBER)
        if constexpr (EL::code == HZ_BF16) {
#define HZ_EL_ASM "bf16"
          if constexpr (BW) { if (blockwise) HZ_K1B_BODY(); else HZ_K1_BODY(); } else HZ_K1_BODY();
#undef HZ_EL_ASM
        } else {
#define HZ_EL_ASM "f16"
          if constexpr (BW) { if (blockwise) HZ_K1B_BODY(); else HZ_K1_BODY(); } else HZ_K1_BODY();
#undef HZ_EL_ASM
        }
#undef HZ_K1B_BODY
#undef HZ_K1_BODY
#undef HZ_K1N
#undef HZ_K1
#undef HZ_RD1
      } else {
        // two row tiles: weights 2 k-steps ahead (one ring slot spare), activation fragments one k-step ahead in two slots per row tile
        v8 b00, b01, b10, b11;
        unsigned int la1 = la0 + (unsigned int)(16 * rs * 2);
#define HZ_RD2(B0, B1, OFF) "ds_read_b128 %[" B0 "], %[la0] offset:" #OFF "\n\tds_read_b128 %[" B1 "], %[la1] offset:" #OFF "\n\t"
#define HZ_K2(W0, W1, WN0, WN1, B0, B1, RD, LGKM)                                             \
  HZ_LD(WN0, WN1) RD "s_waitcnt vmcnt(4) lgkmcnt(" LGKM ")\n\t"                               \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a00], " W0 ", %[" B0 "], %[a00]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a01], " W0 ", %[" B1 "], %[a01]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a10], " W1 ", %[" B0 "], %[a10]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a11], " W1 ", %[" B1 "], %[a11]\n\t"
#define HZ_K2N(W0, W1, B0, B1, RD, VM, LGKM)                                                  \
  RD "s_waitcnt vmcnt(" VM ") lgkmcnt(" LGKM ")\n\t"                                          \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a00], " W0 ", %[" B0 "], %[a00]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a01], " W0 ", %[" B1 "], %[a01]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a10], " W1 ", %[" B0 "], %[a10]\n\t"                   \
  "v_mfma_f32_16x16x32_" HZ_EL_ASM " %[a11], " W1 ", %[" B1 "], %[a11]\n\t"
#define HZ_K2_BODY()                                                                                            \
        asm volatile(                                                                                           \
            HZ_RD2("b00", "b01", 0)                                                                               \
            "1:\n\t"                                                                                            \
            HZ_K2(HZ_W00, HZ_W01, HZ_W20, HZ_W21, "b00", "b01", HZ_RD2("b10", "b11", 64), "2")                     \
            HZ_K2(HZ_W10, HZ_W11, HZ_W30, HZ_W31, "b10", "b11", HZ_RD2("b00", "b01", 128), "2")                    \
            HZ_K2(HZ_W20, HZ_W21, HZ_W00, HZ_W01, "b00", "b01", HZ_RD2("b10", "b11", 192), "2")                    \
            HZ_K2(HZ_W30, HZ_W31, HZ_W10, HZ_W11, "b10", "b11", HZ_RD2("b00", "b01", 256), "2")                    \
            "v_add_u32 %[la0], 0x100, %[la0]\n\tv_add_u32 %[la1], 0x100, %[la1]\n\t"                              \
            "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 1\n\ts_cbranch_scc1 1b\n\t"                     \
            HZ_K2(HZ_W00, HZ_W01, HZ_W20, HZ_W21, "b00", "b01", HZ_RD2("b10", "b11", 64), "2")                     \
            HZ_K2(HZ_W10, HZ_W11, HZ_W30, HZ_W31, "b10", "b11", HZ_RD2("b00", "b01", 128), "2")                    \
            "s_cmp_lg_u32 %[last], 0\n\ts_cbranch_scc1 4f\n\t"                                                  \
            HZ_K2(HZ_W20, HZ_W21, HZ_W00, HZ_W01, "b00", "b01", HZ_RD2("b10", "b11", 192), "2")                    \
            HZ_K2(HZ_W30, HZ_W31, HZ_W10, HZ_W11, "b10", "b11", "", "0")                                           \
            "s_branch 5f\n\t4:\n\t"  /* this wave's last job of the inference: nothing to request ahead, nothing left in flight */ \
            HZ_K2N(HZ_W20, HZ_W21, "b00", "b01", HZ_RD2("b10", "b11", 192), "2", "2") HZ_K2N(HZ_W30, HZ_W31, "b10", "b11", "", "0", "0") \
            "5:\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                                                                \
            : [b00] "=&v"(b00), [b01] "=&v"(b01), [b10] "=&v"(b10), [b11] "=&v"(b11),                              \
              [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a10] "+v"(acc[1][0]), [a11] "+v"(acc[1][1]),          \
              [voff] "+v"(voff), [la0] "+v"(la0), [la1] "+v"(la1), [cnt] "+s"(cnt)                                 \
            : [sa] "s"(wbase), [kss] "s"(kss), [last] "s"(last)                                                   \
            : "memory", "scc", HZ_RING_CLOBBER)
#define HZ_K2B_BODY()                                                                                           \
        asm volatile(                                                                                           \
            "1:\n\t" HZ_POLL()                                                                                   \
            HZ_RD2("b00", "b01", 0)                                                                               \
            HZ_K2(HZ_W00, HZ_W01, HZ_W20, HZ_W21, "b00", "b01", HZ_RD2("b10", "b11", 64), "2")                     \
            HZ_K2(HZ_W10, HZ_W11, HZ_W30, HZ_W31, "b10", "b11", HZ_RD2("b00", "b01", 128), "2")                    \
            HZ_K2(HZ_W20, HZ_W21, HZ_W00, HZ_W01, "b00", "b01", HZ_RD2("b10", "b11", 192), "2")                    \
            HZ_K2(HZ_W30, HZ_W31, HZ_W10, HZ_W11, "b10", "b11", "", "0")                                           \
            "v_add_u32 %[la0], 0x100, %[la0]\n\tv_add_u32 %[la1], 0x100, %[la1]\n\tv_add_u32 %[fa], 4, %[fa]\n\t"  \
            "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t"                     \
            "s_nop 7\n\ts_nop 7\n\ts_nop 7"                                                                      \
            : [b00] "=&v"(b00), [b01] "=&v"(b01), [b10] "=&v"(b10), [b11] "=&v"(b11),                              \
              [a00] "+v"(acc[0][0]), [a01] "+v"(acc[0][1]), [a10] "+v"(acc[1][0]), [a11] "+v"(acc[1][1]),          \
              [voff] "+v"(voff), [la0] "+v"(la0), [la1] "+v"(la1), [cnt] "+s"(cnt), [fa] "+v"(fa), [pt] "=&v"(pt), \
              [ps] "=&s"(ps), [pc] "+s"(pc)                                                                       \
            : [sa] "s"(wbase), [kss] "s"(kss)                                                                     \
            : "memory", "scc", HZ_RING_CLOBBER)
        if constexpr (EL::code == HZ_BF16) {
#define HZ_EL_ASM "bf16"
          if constexpr (BW) { if (blockwise) HZ_K2B_BODY(); else HZ_K2_BODY(); } else HZ_K2_BODY();
#undef HZ_EL_ASM
        } else {
#define HZ_EL_ASM "f16"
          if constexpr (BW) { if (blockwise) HZ_K2B_BODY(); else HZ_K2_BODY(); } else HZ_K2_BODY();
#undef HZ_EL_ASM
        }
#undef HZ_K2B_BODY
#undef HZ_POLL
#undef HZ_K2_BODY
#undef HZ_K2N
#undef HZ_K2
#undef HZ_RD2
      }
      pc_left = pc;
    } else {
      // ---- the compiler-scheduled k-loop of the other shapes
      // activation fragments, BQPF k-steps ahead of their use (RT = 2 has the registers for one step ahead only -- two
      // measured no faster, with more spills; its four MFMAs per k-step cover the LDS round trip)
      constexpr int BQD = ((SPLIT || NOASM) && NT == 2) ? 2 : (RT == 1 ? 4 : 2);
      constexpr int BQPF = BQD - 1;
      v8 bq[BQD][RT * WP];
#pragma unroll
      for (int d = 0; d < BQPF; ++d)
        if (d < J.ks) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < WP; ++p)
              bq[d][WP * rt + p] = *reinterpret_cast<const v8*>(src + (size_t)(16 * rt) * rs + p * LP + 32 * d);
        }
      __builtin_amdgcn_sched_barrier(0);
      // one k-step: request the fragments PF steps ahead (weights) / BQPF steps ahead (activations; unconditional: the last
      // trips read past the K range, into fragments nobody uses), then this step's MFMAs
#define HZ_MLP_STEP(S, U)                                                                                            \
      {                                                                                                              \
        _Pragma("unroll") for (int t = 0; t < NT * WP; ++t) wf[((U) + PF) % RING][t] = wp(gstep + (S) + PF, t);       \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) _Pragma("unroll") for (int p = 0; p < WP; ++p)             \
            bq[((U) + BQPF) % BQD][WP * rt + p] =                                                                    \
                *reinterpret_cast<const v8*>(src + (size_t)(16 * rt) * rs + p * LP + 32 * ((S) + BQPF));             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {           \
          acc[t][rt] = EL::mfma(wf[(U) % RING][WP * t], bq[(U) % BQD][WP * rt], acc[t][rt]);                          \
          if constexpr (SPLIT) {  /* hi * lo, lo * hi */                                                             \
            acc[t][rt] = EL::mfma(wf[(U) % RING][WP * t], bq[(U) % BQD][WP * rt + WP - 1], acc[t][rt]);               \
            acc[t][rt] = EL::mfma(wf[(U) % RING][WP * t + WP - 1], bq[(U) % BQD][WP * rt], acc[t][rt]);               \
          }                                                                                                          \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }
      // all but the last 8 k-steps in a loop, the last 8 peeled: the fragment loads they issue sit between the epilogue
      // operands' loads above and their first use, so the compiler can wait with a counted vmcnt instead of draining the ring
      int s = 0;
      const int prio_grp = wave >> 2;
      // the priorities rotate every 4 k-steps, 16-row shapes only (see hz_rotate_prio)
#define HZ_PRIO_AT(S, U) \
  if (RT == 1 && ((U) & 3) == 0) hz_rotate_prio(prio_grp + (int)((gstep + (S)) >> 2));
      for (; s + 8 < J.ks; s += 8) {
        HZ_PRIO_AT(s, 0) HZ_MLP_STEP(s, 0)
        HZ_MLP_STEP(s + 1, 1)
        HZ_MLP_STEP(s + 2, 2)
        HZ_MLP_STEP(s + 3, 3)
        HZ_PRIO_AT(s + 4, 4) HZ_MLP_STEP(s + 4, 4)
        HZ_MLP_STEP(s + 5, 5)
        HZ_MLP_STEP(s + 6, 6)
        HZ_MLP_STEP(s + 7, 7)
      }
      HZ_PRIO_AT(s, 0) HZ_MLP_STEP(s, 0)
      HZ_MLP_STEP(s + 1, 1)
      HZ_MLP_STEP(s + 2, 2)
      HZ_MLP_STEP(s + 3, 3)
      HZ_PRIO_AT(s + 4, 4) HZ_MLP_STEP(s + 4, 4)
      HZ_MLP_STEP(s + 5, 5)
      HZ_MLP_STEP(s + 6, 6)
      HZ_MLP_STEP(s + 7, 7)
#undef HZ_PRIO_AT
#undef HZ_MLP_STEP
      gstep += J.ks;
    }
    PROF_ADD(p_loop, p_j2);
    PROF_TL(j, 2);
    if constexpr (ASMK && BW) HZ_POLL_GIVEUP(pc_left);
    const unsigned long long p_j3 = PROF_NOW();

    // epilogue: (+ residual) (+ ReLU) in fp32, round to EL, 4 consecutive columns per lane
    const bool relu = J.flags & HZ_MLP_RELU;
    // all residual fragments in one batch of LDS reads (one wait), not one round trip per column tile
    uint2 rr[NT][RT], rl[NT][RT];  // (rl: the residual's lo halves, SPLIT)
    if (J.res_off >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const uint16_t* rp = lds + (size_t)(16 * rt + r0) * rs + J.res_off + 16 * t + c4;
          rr[t][rt] = *reinterpret_cast<const uint2*>(rp);
          rl[t][rt] = SPLIT ? *reinterpret_cast<const uint2*>(rp + LP) : make_uint2(0u, 0u);
        }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rr[t][rt] = rl[t][rt] = make_uint2(0u, 0u);  // +0 in either format: adds nothing
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 16 * t + c4;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const size_t rowbase = (size_t)(16 * rt + r0) * rs;
        float v[4] = {acc[t][rt][0], acc[t][rt][1], acc[t][rt][2], acc[t][rt][3]};
        v[0] += EL::lo(rr[t][rt].x); v[1] += EL::hi(rr[t][rt].x);
        v[2] += EL::lo(rr[t][rt].y); v[3] += EL::hi(rr[t][rt].y);
        if constexpr (SPLIT) {
          v[0] += EL::lo(rl[t][rt].x); v[1] += EL::hi(rl[t][rt].x);
          v[2] += EL::lo(rl[t][rt].y); v[3] += EL::hi(rl[t][rt].y);
        }
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (v[r] > 0.0f || v[r] != v[r]) ? v[r] : 0.0f;
        }
        if (J.flags & HZ_MLP_F32_OUT) {  // the head's last layer: fp32 to the scalar transform / the tree (include/hz_mlp.h)
          *reinterpret_cast<float4*>(lds + rowbase + J.dst_off + 2 * col) = make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (SPLIT) {
          uint2 o, ol;
          EL::halves(v[0], v[1], o.x, ol.x);
          EL::halves(v[2], v[3], o.y, ol.y);
          *reinterpret_cast<uint2*>(lds + rowbase + J.dst_off + col) = o;
          *reinterpret_cast<uint2*>(lds + rowbase + J.dst_off + LP + col) = ol;
        } else {
          uint2 o;
          o.x = EL::pack(v[0], v[1]);
          o.y = EL::pack(v[2], v[3]);
          *reinterpret_cast<uint2*>(lds + rowbase + J.dst_off + col) = o;
        }
      }
    }
    if (ASMK && (J.flags & HZ_MLP_SIGNAL)) {
      // this wave's columns are in the image before its arrival shows: the epilogue's LDS stores are not to sink below this point
      // (compiler barrier) and have completed (lgkmcnt; an LDS-only wait: a release fence proper may also wait for vector memory,
      // i.e. for the weight ring)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned int* ready = reinterpret_cast<unsigned int*>(lds + (size_t)j * rs + (rs - 8)) + (wave >> 2);
      if (lane == 0) __hip_atomic_fetch_add(ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PROF_ADD(p_epi, p_j3);
    PROF_TL(j, 3);
  }
  if constexpr (ASMK)  // the last requests ran into the padding behind the stream: nobody waits for them, so wait here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", HZ_RING_CLOBBER);
#undef HZ_LD
#undef wp
  const unsigned long long p_jobs_done = PROF_NOW();
  (void)p_jobs_done;
  __builtin_amdgcn_s_setprio(0);
  if (FINAL) __syncthreads();  // (without the final stage the caller's own barrier follows)
  // heads -> scalars / policy logits: 32 lanes per (row, head) pair
  if (FINAL) {
    const int l32 = tid & 31, slot = tid >> 5;
    for (int pair = slot; pair < 2 * MT; pair += NTHR / 32) {
      const int r = pair >> 1, head = pair & 1;
      if (row0 + r < n_rows) {  // (uniform over the 32 lanes of a pair)
        const uint16_t* row = lds + (size_t)r * rs;
        const float x = row32_support_to_scalar(row, head ? H.off_value : H.off_reward, head ? H.off_value2 : H.off_reward2,
                                                H.logit_split, H.support_size, H.support_min, l32);
        if (l32 == 0) (head ? out_value : out_reward)[row0 + r] = x;
      }
    }
    for (int i = tid; i < MT * H.num_actions; i += NTHR) {
      const int r = i / H.num_actions, a = i % H.num_actions;
      if (row0 + r < n_rows) {
        out_policy[(size_t)(row0 + r) * H.num_actions + a] = row_policy_logit(lds + (size_t)r * rs, H.off_policy, a);
      }
    }
  }
#ifdef HZ_MLP_PROFILE
  if (blockIdx.x == 100 && lane == 0) {
    unsigned long long* o = hz_mlp_prof + wave * 8;
    o[0] = p_staged - p_t0; o[1] = p_bar; o[2] = p_pre; o[3] = p_loop; o[4] = p_epi;
    o[5] = PROF_NOW() - p_jobs_done; o[6] = PROF_NOW() - p_t0; o[7] = (unsigned long long)gstep;
    for (int i = 0; i < 32; ++i) hz_mlp_prof_tl[wave * 32 + i] = prof_tl[wave * 32 + i];
    if (wave == 0) {
      hz_mlp_prof_pass[32] = p_t0;
      hz_mlp_prof_pass[33] = PROF_NOW();
    }
  }
#endif
}
