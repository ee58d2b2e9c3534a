/*
 * palace_rccl.h -- the multi-GPU exchange of the eref count table as a C entry point, for a C++ host that drives one
 * process per GPU itself (libpalace_rccl.so: links libpalace_hip.so and RCCL; kept out of libpalace_hip.so so that the
 * single-GPU executables do not load a collective library they never call).
 *
 * The reference shares ONE count table between std::threads of one process (bin/extract_ref.cpp:1269-1291); with the reads
 * of a sample sharded over the GPUs of a node every rank counts into a private table and the tables are merged
 * (SURVEY.md section 8(e)).  bench.py does this exchange through torch.distributed (palace_amd/multigpu.py); this is the same
 * exchange without Python: every rank owns 1/world of the key space, peers send it their slice of TWO planes (the unary planes
 * of a partial table carry two bits per key: palace_eref_table_pack_low), the owner folds the parts with the saturating
 * bit-plane add (palace_eref_table_merge_slices_packed) and the merged ">= 3" plane -- the only one Phase B reads -- is
 * all-gathered.  Point-to-point sends to every peer at once (all seven xGMI links of a GPU busy), no ring all-reduce of the
 * whole table.
 */
#ifndef PALACE_RCCL_H
#define PALACE_RCCL_H

#include "palace_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* comm: an ncclComm_t of `world` ranks on which this process is `rank` (RCCL over xGMI inside a node); the calls are
 * enqueued on the context's stream, behind the counting kernels, and nothing waits for the host.  On return (stream order)
 * plane 3 of the context's table is the global "count >= 3" plane on every rank; planes 1 and 2 hold the merged values for
 * this rank's slice only.  Needs 3 x 2^29 bytes of the context's scratch.  world must divide 2^25 (slices stay 16-byte
 * aligned); world == 1 is allowed (the calls degenerate to copies). */
int palace_eref_table_exchange(palace_ctx *ctx, void *comm, int rank, int world);

/* The other way to spread Phase A over the GPUs of a node, and the cheaper one when every rank can hold all reads (they are
 * 0.375 bytes per base packed): every rank counts ALL reads but only the keys of its share of the key space
 * (palace_eref_set_key_buckets with the mask palace_eref_key_share gives: mirrored pairs {r, 2W-1-r} of every 2W of the 128
 * buckets -- equal key mass, since canonical keys thin out linearly over the key space); its 4 MiB slices of the ">= 3"
 * plane are then exact, and palace_eref_key_share_gather completes the plane on every rank (each slice from its owner, in
 * place, stream-ordered).  No partial tables, no merge.  world must divide 64. */
int palace_eref_key_share(int rank, int world, uint32_t mask128[4]);
int palace_eref_key_share_gather(palace_ctx *ctx, void *comm, int rank, int world);
/* The same gather with the plane in SPARSE form (palace_eref_plane_pack / _unpack of palace_hip.h): every rank sends the number of
 * set bits of each of its fine buckets and their 16-bit offsets -- 48 MB for the whole plane of a 1M-contig sample instead of
 * 512 MiB of slices.  cap_keys = room for ONE rank's keys (the same on every rank).  h_max_keys, when not NULL, receives the
 * number of keys the largest share holds (the call then waits for the stream): a value above cap_keys means keys were cut off
 * -- call palace_eref_key_share_gather instead (the counting is not repeated: the rank's own slices are untouched), and give
 * the next sample that much room + a margin.  A first call with cap_keys = 0 only sizes. */
int palace_eref_key_share_gather_sparse(palace_ctx *ctx, void *comm, int rank, int world, int64_t cap_keys, unsigned long long *h_max_keys);

/* The reads sharded, partial COUNTS of the DB's probe-index entries exchanged instead of planes (palace_hip.h: palace_eref_entry_layout):
 * after a final count with option "probe_all_sets" 2 and the whole DB's index attached (a rank without reads makes the same call with
 * n = 0, which zeroes its block; a context that made no such call since its last reset is refused: PALACE_ESTATE, before anything is sent),
 * every rank sends each peer that peer's share of its count block, sums the `world` parts of its own share into hit bits, and the
 * shares of the hit-bit block are all-gathered; the block is then declared whole (keys_counted: key instances of ALL ranks, -1 =
 * unknown), and palace_eref_scan_refs_indexed starts from it (options "scan_ref_lo" / "scan_ref_hi": this rank's refs).  The blocks are
 * the index's own unless palace_eref_entry_buffers_attach put the caller's in their place; world must divide the count block into
 * 512-byte multiples (1 .. 8 always do). */
int palace_eref_entry_counts_exchange(palace_ctx *ctx, palace_eref_probe_index *ix, void *comm, int rank, int world, int64_t keys_counted);

/* Phase B rows: every rank scanned the refs [ref_lo[r], ref_hi[r]) and holds their rows (4 x int32 per ref) in d_rows;
 * afterwards every rank holds all n_refs rows.  ref_lo / ref_hi: host arrays of `world` entries, the same on every rank. */
int palace_eref_rows_allgather(palace_ctx *ctx, void *comm, int rank, int world, int32_t *d_rows, int64_t n_refs,
                               const int64_t *ref_lo, const int64_t *ref_hi);

#ifdef __cplusplus
}
#endif
#endif /* PALACE_RCCL_H */
