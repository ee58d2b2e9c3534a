/* palace_hip_diag.h -- diagnosis entry points of libpalace_hip.so.  NOT part of the drop-in C ABI (include/palace_hip.h): unstable,
 * may change or go away with the experiment they serve; nothing of the product path calls them (bench/step.py does, behind
 * PALACE_BENCH_DISTURB, to measure what a stream of random memory operations costs the counting kernels beside it). */
#ifndef PALACE_HIP_DIAG_H
#define PALACE_HIP_DIAG_H
#include "palace_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnosis (no reference counterpart): `launches` kernels of `blocks` x 256 threads on the context's stream that together do n_ops
 * random memory operations over d_buf[0 .. n_slots) (8-byte slots): mode 0 = 8-byte loads, 1 = 64-bit atomicMin, 2 = 8-byte stores,
 * 3 = 1-byte loads.  Used to measure what a stream of such operations costs a bandwidth-bound launch on another stream
 * (tools/archive/r04u.sh). */
int palace_diag_disturb(palace_ctx *ctx, void *d_buf, uint64_t n_slots, uint64_t n_ops, int mode, int launches, int blocks);

#ifdef __cplusplus
}
#endif
#endif
