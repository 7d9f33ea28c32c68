/* hz_mlp.h -- C ABI of the fused recurrent-inference kernel of the search loop (libhanabizero_hip.so).
 *
 * What it replaces: one call of BaseMuZeroNet.recurrent_inference (/root/reference/core/model.py:74-84) as issued by
 * core/mcts.py:31-50 -- the per-tree gather of the parent hidden state (mcts.py:31-36), dynamics
 * (config/hanabi_control/model.py:61-125, the one-hot action concat of :215-219 becomes a row of the first layer's
 * action block added in its epilogue), reward head + prediction heads (model.py:138-149 / 250-269), inverse scalar
 * transforms (core/config.py:204-232) and NaN clearing of the policy logits (core/mcts.py:48-49) -- for all N trees.
 * In PyTorch this is a chain of ~10 small GEMMs bound by launch latency and activation reloads; here ONE kernel keeps
 * each workgroup's rows' activations in LDS across all layers and streams the (eval-mode, BatchNorm-folded, bf16)
 * weights through v_mfma_f32_16x16x32_bf16 (fp32 accumulate): the matrix cores do exactly the net GEMMs, nothing else.
 *
 * The layer chain is DATA: a list of "jobs".  Job j of wave w (num_waves = 4, 8 or 16 waves per workgroup) computes
 * C = 16 * tiles_per_wave output columns (64, or 32 with 16 waves)
 *     out[:, dst_off : dst_off+C] = act( in[:, src_off : src_off + 32*ks] @ W_jw^T + b_jw [+ action row] [+ residual] )
 * over the workgroup's rows, reading and writing one LDS image row per batch row.  Every wave owns ONE weight stream
 * (its jobs' fragments in execution order; the streams are interleaved k-step by k-step), prefetched 3 k-steps ahead
 * in a register ring across job and layer boundaries.  hanabizero_amd/model.py::FusedRecurrent (recurrent inference)
 * and ::FusedInitialTail (the small-GEMM tail of the initial inference) build job tables and streams from a module.
 * Numerics: 16-bit (fp16 -- the reference's own autocast format, core/mcts.py:38-40 -- or bf16) weights and activations
 * between layers, fp32 accumulation and epilogue; the last layer of every head (value / reward / policy logits) stays fp32
 * through the scalar transform (HZ_MLP_F32_OUT).  Measured against the reference nets' fp32 outputs AND against the
 * reference run under fp16 autocast (tests/golden/nets_*.npz, nets_*_autocast.npz; tests/test_model.py).
 * HZ_F16X2 (4- and 8-wave shapes, hz_mlp_recurrent only) is the build inside the contract's 1e-3 of the reference's fp32 nets:
 * every fp32 weight and activation as a pair of fp16 (hi = fp16(x), lo = fp16(x - hi)), every product as hi*hi + hi*lo + lo*hi
 * on the same v_mfma_f32_16x16x32_f16 into the same fp32 accumulators (twice the weight stream, three times the MFMAs,
 * which the matrix cores have to spare).  Its state rows, hidden_out rows and strides are fp32; a wave's weight stream holds a
 * tile's lo fragment behind its hi fragment (kstep_stride counts both); the image keeps the lo halves `lo_plane` columns
 * behind the hi halves.
 * Conventions as include/hz_tree.h.
 */
#ifndef HZ_MLP_H
#define HZ_MLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  HZ_MLP_RELU = 1,          /* ReLU in the epilogue */
  HZ_MLP_ACTION_ROW = 2,    /* start from action_table[action[row]] instead of action_table[num_actions] (first dynamics layer) */
  HZ_MLP_BARRIER = 4,       /* workgroup barrier before the job (layer boundary); same for the 4 waves of a job */
  HZ_MLP_STORE_HIDDEN = 8,  /* after that barrier copy LDS columns [hidden_off, hidden_off+hidden) to hidden_out */
  /* 16 waves x 2 tiles only -- a layer boundary without a workgroup barrier.  A full-width layer (16 waves, wave w
   * producing columns [32 w, 32 w + 32) of a 512-column output) whose jobs carry HZ_MLP_SIGNAL counts, per group of four
   * waves, the waves that have written their columns: counter g of job j = the four waves 4 g .. 4 g + 3 = columns
   * [128 g, 128 g + 128).  A job with HZ_MLP_BLOCKWISE (instead of HZ_MLP_BARRIER) reads exactly that output (16
   * k-steps) and waits in front of every block of four k-steps until counter (block) of job `producer` says 4: a wave
   * that is done with a layer starts the next one on the columns that exist already while the slower waves finish, so
   * the weight stream does not stop at the boundary.  The counters live in the 16 B of padding behind image row j (row_stride
   * - used width >= 8 elements, j < 16) and are cleared at the start of every inference. */
  HZ_MLP_SIGNAL = 16,
  HZ_MLP_BLOCKWISE = 32,
  /* 16 waves x 2 tiles only -- a pass that begins with neither: before it starts, a job waits for up to four arrival counters
   * (tokens), the ones of the jobs its own columns depend on: writers of what it reads, readers and writers of what it
   * writes.  Token count = bits 8..10 of flags; token k = byte k of `producer`: pass << 4 | group << 2 | (count - 1) -- wait
   * until counter `group` of job `pass` says `count` (the jobs of that pass carry HZ_MLP_SIGNAL; an idle entry signals
   * nothing, so count = the group's entries with ks != 0).  The flag is the same for the 16 entries of a pass (a kernel
   * without the counters runs a workgroup barrier there instead); tokens are per entry, idle entries may have some (the
   * hidden-state store reads columns too).  hanabizero_amd/mlp_sync.py derives the tokens from the jobs' column ranges and
   * proves the table race-free. */
  HZ_MLP_WAITS = 64,
  /* 16 waves x 2 tiles only, per entry: this is the wave's last job of the inference -- its k-loop requests no weight
   * fragments past its own (the other jobs' loops run 3 k-steps ahead into the next job's), so nothing is in flight when
   * the wave leaves the chain. */
  HZ_MLP_LAST = 128,
  /* bits 8..10: token count of HZ_MLP_WAITS */
  /* The job's outputs leave the epilogue as fp32, not rounded to the element format: output column c of the job goes to
   * image columns [dst_off + 2 c, dst_off + 2 c + 2) (two 16-bit columns hold one float; dst_off % 8 == 0).  For the last
   * layer of a head: the categorical value / reward logits and the policy logits reach the scalar transform (softmax . support
   * through h^-1 amplifies their rounding) and the tree with the accumulators' precision.  No residual, no consumer job: only the
   * final stage (or the search kernel's tree waves) reads them. */
  HZ_MLP_F32_OUT = 2048
};

/* One (job, wave) entry; all offsets are bf16-element columns of the LDS row image.  ks == 0: this wave idles.
 * "64" below is 16 * tiles_per_wave: 64 columns per job with 4 waves x 4 tiles, 32 with 16 waves x 2 tiles. */
typedef struct {
  int32_t ks;        /* k-steps of 32 inputs; multiple of 8 (pad K with zero weights) */
  int32_t src_off;   /* first input column */
  int32_t dst_off;   /* first of the 64 output columns */
  int32_t res_off;   /* first residual column (added before the ReLU) or -1 */
  int32_t bias_off;  /* first of this job's columns in action_table rows (and in biases) */
  int32_t flags;
  int32_t reserved0;
  int32_t producer;  /* HZ_MLP_BLOCKWISE: index of the job whose counters guard this job's input blocks; HZ_MLP_WAITS: tokens */
} hz_mlp_job_t;

typedef struct {
  int32_t n_jobs;      /* <= 32 */
  int32_t row_stride;  /* LDS elements per row, multiple of 8, = 8 (mod 128) for conflict-free ds_read_b128 */
  int32_t hidden;      /* width of the hidden state written by HZ_MLP_STORE_HIDDEN (multiple of 8) */
  int32_t state_off;   /* LDS column where the input hidden state is staged */
  int32_t hidden_off;  /* LDS column of the next hidden state when HZ_MLP_STORE_HIDDEN fires */
  int32_t off_reward, off_value, off_policy; /* LDS columns (16-bit units, multiples of 8) where the final reward / value /
                                                policy logits start -- fp32 (HZ_MLP_F32_OUT): logit k of a head sits 2 k
                                                columns behind its offset, except ... */
  int32_t support_size, support_min, num_actions;
  int32_t action_table_stride;   /* fp32 elements per action row */
  int32_t in_width;              /* elements of an input row staged at state_off (multiple of 8; = hidden for the
                                    recurrent inference, the width of the last big representation layer for the tail
                                    of the initial inference) */
  int32_t dtype;                 /* element format of weights, activations and state rows: HZ_BF16, HZ_F16 or HZ_F16X2
                                    (include/hz_tree.h); accumulation and epilogues are fp32 in all */
  int32_t num_waves;             /* 4 (stand-alone kernel) or 16 (inside hz_search_run) */
  int32_t tiles_per_wave;        /* 16-column MFMA tiles per job: 4 with 4 waves, 2 with 16 waves */
  int32_t logit_split;           /* ... value / reward logits k >= logit_split (a multiple of 32; >= support_size: none), which sit
                                    2 (k - logit_split) columns behind off_value2 / off_reward2: a head's fp32 logits may
                                    occupy two dead regions of the image instead of one contiguous range */
  int32_t off_reward2, off_value2;
  int32_t lo_plane;              /* HZ_F16X2 only (0 otherwise): image columns between the hi half of an element and its lo half */
  int64_t kstep_stride;          /* elements between consecutive k-steps of one wave's stream: 512 * tiles_per_wave when
                                    each stream is contiguous, 512 * tiles_per_wave * num_waves when the streams are
                                    interleaved k-step by k-step (all waves of a workgroup then read one contiguous
                                    region at a time: every L2 channel serves an equal share) */
  int64_t wave_stream_off[16];   /* element offset of each wave's weight stream inside `wstream` */
} hz_mlp_header_t;

/* jobs        [n_jobs][num_waves] hz_mlp_job_t (DEVICE)
 * wstream     packed bf16 weight streams (DEVICE; each stream followed by >= 8 k-steps of zero padding)
 * action_table [num_actions + 1][action_table_stride] fp32 (DEVICE): the additive term of every output column by action --
 *             row a < num_actions = bias + the action's column of the first dynamics layer where a job has
 *             HZ_MLP_ACTION_ROW, row num_actions = the bias alone.  The accumulators start from a row of it (fp32), the
 *             MFMAs add the products on top, the epilogue adds only the residual.
 * biases      fp32 (DEVICE): the biases alone, in the same column layout (kept for inspection; the kernel reads action_table)
 * state rows  row i is read from state_src + plane_index[i]*plane_stride + i*row_stride (bf16 elements);
 *             plane_index may be NULL (= 0).  With the search's pool [S][N][H]: plane_index = hz_tree_traverse's
 *             out_ix, plane_stride = N*H, row_stride = H -> the gather of core/mcts.py:31-36 happens here.
 * actions     [N] i32 (hz_tree_traverse's out_last_action)
 * hidden_out  [N][hidden] bf16; out_reward / out_value [N] f32 (inverse scalar transform, NaN -> 0);
 * out_policy  [N][num_actions] f32 (NaN -> 0).  rows_per_wg: 16 or 32. */
int hz_mlp_recurrent(const hz_mlp_header_t* host_header, const hz_mlp_job_t* jobs, const void* wstream,
                     const float* biases, const float* action_table, const void* state_src, int64_t row_stride,
                     const int32_t* plane_index, int64_t plane_stride, const int32_t* actions, void* hidden_out,
                     float* out_reward, float* out_value, float* out_policy, int num_rows, int rows_per_wg,
                     void* stream);

/* The same with input rows relu(state_src[i] + state_res[i]) (state_res: [N] rows of res_stride elements in the header's dtype;
 * plane_index must be NULL): the residual-add + ReLU in front of the chain (hz_add_relu's arithmetic, include/hz_netglue.h)
 * applied while the rows are staged, instead of a launch of its own -- the tail of the root inference takes the output of the
 * representation net's first residual block this way (config/hanabi_control/model.py:43-57, 240-248). */
int hz_mlp_recurrent_res(const hz_mlp_header_t* host_header, const hz_mlp_job_t* jobs, const void* wstream,
                         const float* biases, const float* action_table, const void* state_src, int64_t row_stride,
                         const int32_t* plane_index, int64_t plane_stride, const int32_t* actions, void* hidden_out,
                         float* out_reward, float* out_value, float* out_policy, int num_rows, int rows_per_wg,
                         const void* state_res, int64_t res_stride, void* stream);

/* Waits on arrival counters (HZ_MLP_BLOCKWISE / HZ_MLP_WAITS) that gave up -- after 2^16 looks, about 70 ms -- since the
 * library was loaded, summed over hz_mlp_recurrent and hz_search_run launches on the current device (synchronises with it).  Must be 0: a
 * wave that gave up went on with inputs that may not have been there.  (bench.py prints it; the GPU tests assert it.) */
int hz_mlp_poll_giveups(unsigned int* count);

/* The same two counters (stand-alone kernel's, search kernels') copied on `stream` into host_pinned2[0], [1] (pinned host
 * memory): no device-wide synchronisation -- for callers that must not stall work queued on other streams.  The values are
 * there once `stream` has reached this point; their sum is hz_mlp_poll_giveups's count.  hanabizero_amd/selfplay.py reads them
 * with every drain of finished games and raises if they grew. */
int hz_mlp_poll_giveups_async(unsigned int* host_pinned2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_MLP_H */
