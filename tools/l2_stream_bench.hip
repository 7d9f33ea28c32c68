// tools/l2_stream_bench.hip -- what the L2 -> CU weight stream of the fused MLP can reach at best on this chip.
// Every workgroup (one per CU) reads the SAME `bytes`-sized buffer (L2-resident: 3.2 MB of weights against 4 MiB of L2 per XCD)
// `iters` times, exactly as the MLP's k-loop does: each wave owns a share of every k-step (1 KiB per load = 16 B per lane),
// the waves' shares interleaved k-step by k-step, DEPTH loads in flight per wave, and nothing else -- no MFMAs, no LDS, no
// barriers.  The rate measured here is the ceiling for `roofline.l2_stream`; the gap to it is what the kernel's other work costs.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/l2_stream_bench tools/l2_stream_bench.hip && gpurun_out/l2_stream_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// NW waves per workgroup, NT fragments of 1 KiB per wave and k-step, DEPTH k-steps in flight per wave
template <int NW, int NT, int DEPTH, bool SYNC>
__global__ __launch_bounds__(64 * NW, 1) void k_stream(const u32x4* __restrict__ w, long long ksteps, int iters, int sync_every,
                                                       unsigned int* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long kss = (long long)NW * NT * 64;  // 16-B units per k-step (all waves)
  const u32x4* base = w + (long long)wave * NT * 64 + lane;
  u32x4 ring[DEPTH][NT];
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[d][t] = base[(long long)d * kss + t * 64];
    for (long long s = 0; s < ksteps; s += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[(u + DEPTH - 1) % DEPTH][t] = base[(s + u + DEPTH - 1) * kss + t * 64];  // (reads into the padding at the end)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc ^= ring[u][t];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (SYNC && ((s / DEPTH) % sync_every) == sync_every - 1) __syncthreads();
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1;
}

// The same stream feeding MFMAs the way the MLP's k-loop does (MODE 1: B operand constant; 2: B fragments read from LDS three
// k-steps ahead; 3: plus a layer boundary every 16 k-steps -- epilogue arithmetic, packed LDS stores, barrier, first LDS reads;
// 4: as 3 with two row tiles = 32 rows per workgroup).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NW, int NT, int DEPTH, int MODE, int PF = DEPTH - 1, int BA = 3>
__global__ __launch_bounds__(64 * NW, 1) void k_stream_mfma(const bf16x8* __restrict__ w, long long ksteps, int iters,
                                                            unsigned int* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int RT = MODE == 4 ? 2 : 1;
  constexpr int RS = 1800;  // image row stride in elements, as the MLP's
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long kss = (long long)NW * NT * 64;
  const bf16x8* base = w + (long long)wave * NT * 64 + lane;
  for (int i = threadIdx.x; i < 16 * RT * RS; i += 64 * NW) lds[i] = 0x3c00;
  __syncthreads();
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  bf16x8 ring[DEPTH][NT];
  bf16x8 bq[4][RT];
  f32x4 acc[NT][RT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned short* src = lds + (lane & 15) * RS + (lane >> 4) * 8;
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) bq[d][rt] = *reinterpret_cast<const bf16x8*>(src + 16 * rt * RS + 32 * d);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[d][t] = base[(long long)d * kss + t * 64];
    for (long long s = 0; s < ksteps; s += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[(u + PF) % DEPTH][t] = base[(s + u + PF) * kss + t * 64];
        if (MODE >= 2) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            bq[(u + BA) % 4][rt] = *reinterpret_cast<const bf16x8*>(src + 16 * rt * RS + 32 * ((u + BA) & 15));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[u % DEPTH][t], bq[u % 4][rt], acc[t][rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE >= 3) {  // a layer boundary: epilogue, packed stores into the image, barrier
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[t][rt][r] + 0.25f;
              v[r] = v[r] > 0.0f ? v[r] * 1e-3f : 0.0f;
              acc[t][rt][r] = 0.0f;
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 f0 = {v[0], v[1]}, f1 = {v[2], v[3]};
            const bf16x2 h0 = __builtin_convertvector(f0, bf16x2), h1 = __builtin_convertvector(f1, bf16x2);
            uint2 o;
            o.x = *reinterpret_cast<const unsigned int*>(&h0);
            o.y = *reinterpret_cast<const unsigned int*>(&h1);
            *reinterpret_cast<uint2*>(lds + (16 * rt + (lane & 15)) * RS + 512 + 32 * wave * NT / 2 * 0 + (wave * NT + t) * 16 % 512 + 4 * (lane >> 4)) = o;
          }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) bq[d][rt] = *reinterpret_cast<const bf16x8*>(src + 16 * rt * RS + 32 * d);
      }
    }
  }
  float x = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x += acc[t][rt][0] + acc[t][rt][1] + acc[t][rt][2] + acc[t][rt][3];
  if (x == 12345.678f) sink[threadIdx.x] = 1;
  if (blockIdx.x == 100 && threadIdx.x == 0) {  // shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of this workgroup
    reinterpret_cast<unsigned long long*>(sink)[256] = __builtin_amdgcn_s_memtime() - st0;
    reinterpret_cast<unsigned long long*>(sink)[257] = __builtin_amdgcn_s_memrealtime() - sr0;
  }
}

template <int NW, int NT, int DEPTH, int MODE, int PF = DEPTH - 1, int BA = 3>
static void run_mfma(const char* name, const void* w, size_t bytes, int grid, unsigned int* sink) {
  const long long ksteps = (long long)(bytes / ((size_t)NW * NT * 1024)) / 16 * 16;
  const int iters = 49;
  const size_t lds_bytes = (size_t)32 * 1800 * 2;
  CHECK(hipFuncSetAttribute((const void*)k_stream_mfma<NW, NT, DEPTH, MODE, PF, BA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stream_mfma<NW, NT, DEPTH, MODE, PF, BA>), dim3(grid), dim3(64 * NW), lds_bytes, 0, (const bf16x8*)w, ksteps, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const double per_wg = (double)ksteps * NW * NT * 1024 * iters;
  unsigned long long st[2] = {0, 0};
  CHECK(hipMemcpy(st, reinterpret_cast<unsigned long long*>(sink) + 256, 16, hipMemcpyDeviceToHost));
  printf("%-52s %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip | %5.1f B / shader cycle, shader clock %.2f GHz\n", name, best, per_wg / best / 1e6,
         per_wg * grid / best / 1e9, per_wg / (double)st[0], (double)st[0] / (double)st[1] * 0.1);
}

template <int NW, int NT, int DEPTH, bool SYNC>
static void run(const char* name, const u32x4* w, size_t bytes, int grid, unsigned int* sink, int sync_every) {
  const long long ksteps = (long long)(bytes / ((size_t)NW * NT * 1024)) / DEPTH * DEPTH;
  const int iters = 49;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_stream<NW, NT, DEPTH, SYNC>), dim3(grid), dim3(64 * NW), 0, 0, w, ksteps, iters, sync_every, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double per_wg = (double)ksteps * NW * NT * 1024 * iters;
  printf("%-44s %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip  (%.2f MB x %d per workgroup, %d workgroups)\n", name, best,
         per_wg / best / 1e6, per_wg * grid / best / 1e9, per_wg / iters / 1e6, iters, grid);
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : 3178496;
  int dev = 0, cus = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = argc > 2 ? atoi(argv[2]) : cus;
  u32x4* w;
  unsigned int* sink;
  CHECK(hipMalloc(&w, bytes + (1 << 20)));
  CHECK(hipMemset(w, 1, bytes + (1 << 20)));
  if (argc > 3 && argv[3][0] == 'r') {  // random bf16 weights ~ U(-1/16, 1/16) instead of one repeated byte: does the data (power) matter?
    const size_t n = (bytes + (1 << 20)) / 2;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    unsigned int x = 12345u;
    for (size_t i = 0; i < n; ++i) {
      x = x * 1664525u + 1013904223u;
      const float f = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 27));
      unsigned int u;
      memcpy(&u, &f, 4);
      h[i] = (unsigned short)(u >> 16);
    }
    CHECK(hipMemcpy(w, h, n * 2, hipMemcpyHostToDevice));
    free(h);
    printf("(random weights)\n");
  }
  CHECK(hipMalloc(&sink, 4096));
  printf("L2 -> CU stream of a %.2f MB buffer shared by %d workgroups (%d CUs)\n", bytes / 1e6, grid, cus);
  run<16, 2, 4, false>("16 waves x 2 KiB/k-step, 4 k-steps in flight", w, bytes, grid, sink, 1);
  run<16, 2, 8, false>("16 waves x 2 KiB/k-step, 8 k-steps in flight", w, bytes, grid, sink, 1);
  run<16, 1, 8, false>("16 waves x 1 KiB/k-step, 8 k-steps in flight", w, bytes, grid, sink, 1);
  run<8, 4, 4, false>(" 8 waves x 4 KiB/k-step, 4 k-steps in flight", w, bytes, grid, sink, 1);
  run<8, 4, 8, false>(" 8 waves x 4 KiB/k-step, 8 k-steps in flight", w, bytes, grid, sink, 1);
  run<4, 4, 8, false>(" 4 waves x 4 KiB/k-step, 8 k-steps in flight", w, bytes, grid, sink, 1);
  run<4, 8, 8, false>(" 4 waves x 8 KiB/k-step, 8 k-steps in flight", w, bytes, grid, sink, 1);
  // ... and with a workgroup barrier every 16 k-steps of 2 KiB x 16 waves (= one 512 x 512 layer), as the layer chain has
  run<16, 2, 4, true>("16 waves x 2 KiB, 4 in flight, barrier / 512 KiB", w, bytes, grid, sink, 4);
  run<16, 2, 8, true>("16 waves x 2 KiB, 8 in flight, barrier / 512 KiB", w, bytes, grid, sink, 2);
  run<8, 4, 8, true>(" 8 waves x 4 KiB, 8 in flight, barrier / 512 KiB", w, bytes, grid, sink, 2);
  printf("the same stream feeding v_mfma_f32_16x16x32_bf16 (A = the streamed fragments)\n");
  run_mfma<16, 2, 4, 1>("16 x 2, 4 in flight: MFMA, B constant", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 2>("16 x 2, 4 in flight: MFMA, B from LDS", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 3>("16 x 2, 4 in flight: + layer boundary / 16 k-steps", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 4>("16 x 2, 4 in flight: + boundary, 32 rows", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 4, 2, 1>("16 x 2, 32 rows, weights 2 ahead, B 1 ahead (product)", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 4, 2, 3>("16 x 2, 32 rows, weights 2 ahead, B 3 ahead", w, bytes, grid, sink);
  run_mfma<16, 2, 4, 4, 3, 1>("16 x 2, 32 rows, weights 3 ahead, B 1 ahead", w, bytes, grid, sink);
  run_mfma<8, 4, 4, 1>(" 8 x 4, 4 in flight: MFMA, B constant", w, bytes, grid, sink);
  run_mfma<8, 4, 4, 2>(" 8 x 4, 4 in flight: MFMA, B from LDS", w, bytes, grid, sink);
  run_mfma<8, 4, 4, 3>(" 8 x 4, 4 in flight: + layer boundary / 16 k-steps", w, bytes, grid, sink);
  run_mfma<8, 4, 8, 3>(" 8 x 4, 8 in flight: + layer boundary / 16 k-steps", w, bytes, grid, sink);
  run_mfma<8, 4, 4, 4>(" 8 x 4, 4 in flight: + boundary, 32 rows", w, bytes, grid, sink);
  run_mfma<8, 4, 8, 4>(" 8 x 4, 8 in flight: + boundary, 32 rows", w, bytes, grid, sink);
  run_mfma<4, 4, 8, 3>(" 4 x 4, 8 in flight: + layer boundary / 16 k-steps", w, bytes, grid, sink);
  return 0;
}
