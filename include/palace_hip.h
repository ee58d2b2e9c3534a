/*
 * palace_hip.h -- C ABI of the MI355X (gfx950) conjugate-graph hot path.
 *
 * The reference has no FFI/plugin boundary: its hot path is three executables coupled by files
 * (SURVEY.md section 8(b)).  This library is the layer the replacement executables
 * (palace_amd/host/{eref,generateGraph,matching}_main.cpp) call; each entry point names the
 * reference code it stands in for.  Plain pointers and sizes only; `d_` arguments are device
 * (HBM) pointers, everything else is host memory.  Every function returns 0 on success and a
 * negative PALACE_E* code on failure; palace_last_error() gives the message for the calling
 * thread.  No exceptions cross this boundary.  A context is bound to one device and one HIP
 * stream; calls on one context are ordered, distinct contexts are independent (one process per
 * GPU is the intended deployment).
 */
#ifndef PALACE_HIP_H
#define PALACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PALACE_OK 0
#define PALACE_EINVAL (-1)   /* bad argument (null pointer, negative size, shape mismatch) */
#define PALACE_EHIP (-2)     /* a HIP runtime call failed */
#define PALACE_ENOMEM (-3)   /* device or host allocation failed */
#define PALACE_ESTATE (-4)   /* call sequence error (e.g. coder not set) */

typedef struct palace_ctx palace_ctx;

const char *palace_last_error(void);
const char *palace_version(void);

/* ---- context, memory, stream plumbing --------------------------------------------------- */
int palace_ctx_create(int device, palace_ctx **out);
int palace_ctx_destroy(palace_ctx *ctx);
int palace_sync(palace_ctx *ctx);
/* raw hipStream_t of the context (for callers that want to order their own work / events) */
void *palace_stream(palace_ctx *ctx);
int palace_malloc(palace_ctx *ctx, size_t bytes, void **d_out);
int palace_free(palace_ctx *ctx, void *d_ptr);
int palace_memset(palace_ctx *ctx, void *d_ptr, int value, size_t bytes);
int palace_h2d(palace_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int palace_d2h(palace_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int palace_d2d(palace_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* HIP-event timing on the context's stream: begin/end bracket, elapsed in milliseconds. */
int palace_timer_begin(palace_ctx *ctx);
int palace_timer_end(palace_ctx *ctx, float *ms_out);
/* Non-blocking variant for timing kernels inside a longer timed region: mark(i) records event i
 * (0 <= i < 4096) on the stream; mark_elapsed(a, b) waits for event b and returns b - a in ms. */
int palace_mark(palace_ctx *ctx, int i);
int palace_mark_elapsed(palace_ctx *ctx, int a, int b, float *ms_out);

/* ---- eref: k-mer screening of reads against the phage DB (bin/extract_ref.cpp) ---------- */

/* E1. Install the per-position coder permutation from the 400-byte index header
 * (replaces generate_coder/generate_base/generate_complement/saved_random_coder,
 * extract_ref.cpp:1010-1080, 1104-1122). */
int palace_eref_set_coder(palace_ctx *ctx, const uint8_t header400[400]);

/* E2. Index build for `n_refs` sequences resident in HBM (ASCII, 1 B/base, concatenated;
 * d_offsets has n_refs+1 entries).  For ref r and position j < len-31 writes the three canonical
 * 32-mer indices (0 = k-mer holds an invalid base) at d_out[d_out_offsets[r] + 3*j + i]
 * (replaces the index loops of read_ref, extract_ref.cpp:711-738, 773-799). */
int palace_eref_index_refs(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                           int64_t n_refs, uint32_t *d_out, const int64_t *d_out_offsets);

/* E4. Count table.  reset zeroes it (extract_ref.cpp:1257); count_reads adds every 32-mer of
 * every read, all three channels, saturating at 3 (read_fastq, extract_ref.cpp:961-1000).
 * d_keep (optional, 1 B/read) carries the E3 subsampling decision (extract_ref.cpp:955-960).
 * The table is held as three 2^32-bit planes "count >= 1 / >= 2 / >= 3". */
int palace_eref_table_reset(palace_ctx *ctx);
int palace_eref_count_reads(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                            int64_t n_reads, const uint8_t *d_keep);

/* E5 + E6. For each ref: look the three indices of every position up in the table and run the
 * 500-base window scan (read_index + slide_window, extract_ref.cpp:813-903, 504-617).
 * one_min / three_min are int(500 * float(ratio)) as computed by the caller (extract_ref.cpp:
 * 513-514).  d_rows receives n_refs x 4 int32: n_intervals, el, ref_len, reserved(0). */
int palace_eref_scan_refs(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                          int64_t n_refs, int64_t total_bases, int one_min, int three_min,
                          int32_t *d_rows);

/* Multi-GPU exchange of the count table (no reference counterpart: the reference shares one
 * table between std::threads, extract_ref.cpp:1269-1291).  planes() exposes the three device
 * buffers (each 2^29 bytes); merge_slices() folds `n_parts` partial tables laid out as
 * [part][plane][slice_bytes] into the caller's planes at byte offset `slice_off`, with the
 * saturating add  (a + b >= t  for t = 1, 2, 3)  done bit-parallel on the planes. */
int palace_eref_table_planes(palace_ctx *ctx, void **d_planes3, size_t *bytes_per_plane);
int palace_eref_table_merge_slices(palace_ctx *ctx, const void *d_parts, int n_parts,
                                   size_t slice_off, size_t slice_bytes);

/* Test hooks: counts (0..3) of `n` indices; population count of each plane. */
int palace_eref_table_lookup(palace_ctx *ctx, const uint32_t *d_keys, int64_t n, uint8_t *d_counts);
int palace_eref_table_popcounts(palace_ctx *ctx, uint64_t out3[3]);

#ifdef __cplusplus
}
#endif
#endif /* PALACE_HIP_H */
