/* hz_rows.h -- a row scatter described as data, so that a launch of another entry point can carry it along.
 *
 * hz_rows_job_t: for every row index r in list[0 .. *count) with slot[r] >= 0 and every array k < num_arrays,
 * dst[k][slot[r]] = src[k][r] (rows of row_bytes[k] bytes): the semantics of hz_rows_scatter / hz_actor_flush
 * (include/hz_selfplay.h).  hz_actor_flush_job fills one from an actor's buffers; hz_env_reset_rows (include/hz_env.h)
 * executes one with additional workgroups of the reset launch.  All pointers are DEVICE pointers. */
#ifndef HZ_ROWS_H
#define HZ_ROWS_H

#include <stdint.h>

typedef struct {
  const int32_t* slot;  /* [rows] destination row per source row, < 0 = skip */
  const int32_t* list;  /* source rows to move */
  const int32_t* count; /* [1] how many entries of `list` */
  int32_t num_arrays;   /* <= 8 */
  int32_t max_rows;     /* upper bound of *count (sizes the launch) */
  const void* src[8];
  void* dst[8];
  int64_t row_bytes[8];
} hz_rows_job_t;

#endif /* HZ_ROWS_H */
