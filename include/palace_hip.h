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
/* A context owns one HIP stream, the eref count table, grow-only device / pinned / host scratch.  Calls on ONE context
 * must come from one thread at a time (they are ordered on its stream); DIFFERENT contexts may be used concurrently from
 * different threads and their work overlaps on the device.  palace_last_error() is per thread. */
int palace_ctx_create(int device, palace_ctx **out);
/* same, with the context's stream at the device's highest priority when high_priority != 0 (for
 * small latency-bound work that runs beside bulk kernels of another context) */
int palace_ctx_create_prio(int device, int high_priority, palace_ctx **out);
/* A context on the CALLER's stream (a hipStream_t): every call of the context enqueues there, so the caller's own work on that
 * stream -- collectives, copies -- is ordered with the library's kernels without events or waits.  The stream is not
 * destroyed with the context. */
int palace_ctx_create_on_stream(int device, void *hip_stream, palace_ctx **out);
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
/* Streaming ingest (the reference reads its inputs with T threads per phase, extract_ref.cpp:1267-1291,
 * and streams the BAM, generate_graph.cpp:644): page-locked host staging buffers, and a host-to-device copy that only
 * enqueues -- the source must stay untouched until a later palace_mark() on the stream has been waited for with
 * palace_mark_wait() (or palace_sync()). */
int palace_host_alloc(palace_ctx *ctx, size_t bytes, void **h_out);
int palace_host_free(palace_ctx *ctx, void *h_ptr);
int palace_h2d_async(palace_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
/* ... and the other way (results of one batch fetched while the next batch is being enqueued): h_dst page-locked, its
 * content is there once a later palace_mark() on the stream has been waited for. */
int palace_d2h_async(palace_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* HIP-event timing on the context's stream: begin/end bracket, elapsed in milliseconds. */
int palace_timer_begin(palace_ctx *ctx);
int palace_timer_end(palace_ctx *ctx, float *ms_out);
/* Non-blocking variant for timing kernels inside a longer timed region: mark(i) records event i
 * (0 <= i < 4096) on the stream; mark_elapsed(a, b) waits for event b and returns b - a in ms. */
int palace_mark(palace_ctx *ctx, int i);
int palace_mark_elapsed(palace_ctx *ctx, int a, int b, float *ms_out);
int palace_mark_wait(palace_ctx *ctx, int i);
/* the same with a deadline: polls the mark and gives up after `seconds` (PALACE_ESTATE; the work is still enqueued -- a caller
 * that cannot wait any longer for a device has to leave without touching what that work writes) */
int palace_mark_wait_for(palace_ctx *ctx, int i, double seconds);
/* Orders two contexts on the device without the host: work enqueued on `ctx` after this call starts only when mark i of
 * `other` (recorded before this call) has been reached on other's stream. */
int palace_wait_for_mark(palace_ctx *ctx, palace_ctx *other, int i);

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
 * total_bases = d_offsets[n_reads] - d_offsets[0] when the caller knows it (keeps the call
 * asynchronous), or -1 to have it read back.
 * The table is held as three 2^32-bit planes "count >= 1 / >= 2 / >= 3". */
int palace_eref_table_reset(palace_ctx *ctx);
/* Optional: allocate the table and the scratch memory a count_reads call over `total_bases` bases will need now (tens of
 * GB at a gigabase; the allocation alone can take from a millisecond to a second), e.g. while the caller is still parsing. */
int palace_eref_reserve(palace_ctx *ctx, int64_t total_bases);
int palace_eref_count_reads(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                            int64_t n_reads, const uint8_t *d_keep, int64_t total_bases);

/* E4 with the read set already packed (replaces the same loop, extract_ref.cpp:927-1004, for a caller that packs while it
 * parses: 0.375 bytes per base cross PCIe instead of 1, and the kernels that derive the streams from ASCII do not run).
 * Three bit streams over the positions 0 .. n_positions-1 of the read set (bit p & 31 of 32-bit word p >> 5, little endian):
 *   P0[p] = base p is A or T,  P1[p] = base p is A or C   (either case; together the base itself -- what the reference's
 *           generate_base tables project, extract_ref.cpp:1010-1046);
 *   U[p]  = a 32-mer is counted at p: positions p .. p+31 lie in ONE read that is counted (E3 subsampling) and all are
 *           A/C/G/T -- the `n`/read-end/short-read tests of extract_ref.cpp:963-996.
 * Positions between reads that belong to no read (pads, e.g. to start every parser thread's part on a word) are allowed:
 * U = 0 there and the other two streams are not looked at.  Each stream: palace_eref_packed_bytes(n_positions) bytes of
 * device memory, 8-byte aligned (two words of look-ahead behind the last position are read, their content is ignored).
 * n_reads_hint: number of reads if known (picks the tile shape for short-read sets), else 0.  Same table, same
 * asynchrony as palace_eref_count_reads. */
size_t palace_eref_packed_bytes(int64_t n_positions);
/* The same three streams made on the device from a read set that is in HBM as ASCII (the arguments of
 * palace_eref_count_reads; positions = bases, no gaps): for a caller that counts one read set more than once -- the streams
 * do not depend on the coder, so one packing serves every DB -- or keeps its samples resident in the packed form. */
int palace_eref_pack_reads(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_reads,
                           const uint8_t *d_keep, int64_t total_bases, uint32_t *d_p0, uint32_t *d_p1, uint32_t *d_u);
int palace_eref_count_reads_packed(palace_ctx *ctx, const uint32_t *d_p0, const uint32_t *d_p1, const uint32_t *d_u,
                                   int64_t n_positions, int64_t n_reads_hint);

/* Tuning knobs of count_reads (no reference counterpart; results are identical for every setting, which is what the
 * tests use them for).  set_count_mode: mode 0 = automatic (partition + LDS counting for large inputs, direct global
 * atomics for tiny ones), 1 = always direct, 2 = always partitioned; bucket_cap > 0 overrides the per-bucket capacity
 * of the partitioned path (keys beyond it take the direct path).  set_option(name, value):
 *   "slab_bases"  positions per slab that large read sets are processed in (multiple of 64; 0 = default 2^30 / 2^31)
 *   "final_count" 1: every count call from now on is the ONLY one between palace_eref_table_reset and the scan.  Phase B
 *                 reads nothing but the "count >= 3" plane (the slide tests `== least_depth`, extract_ref.cpp:23, :531, of a count that
 *                 saturates there, :995), so such a
 *                 call keeps the two lower planes in LDS only: they are not written (1 GB less per call) and stay zero, and
 *                 the next reset clears one plane instead of three.  Afterwards the table cannot take further counts,
 *                 merges or lookups until it is reset (those calls fail); popcounts report 0, 0, n.  Applies to binned
 *                 counts of one slab into a clean table, otherwise the call behaves as without the option.  0: off (default).
 *   "probe_all_sets" 1: a final count (see "final_count") with a probe index attached (palace_eref_attach_probe_index) tests ALL of the
 *                 index's entries -- the three channels of every DB position and the sentinels -- against each fine bucket's ">= 3"
 *                 slice while it is in LDS, and does not write the slice: the table's planes stay all zero (slices the overflow path of
 *                 the partition kernels had written into are zeroed again), so Phase B of that sample is the index's hit bits alone
 *                 (read_index's table look-ups, extract_ref.cpp:858-870, done where the counts are) and the next reset costs nothing.
 *                 Until that reset the table holds NOTHING: only palace_eref_scan_refs_indexed with the attached index works, every
 *                 other call that reads or extends the table fails.  2: the same, but what is left per entry is its partial COUNT (a rank
 *                 that counted a share of the reads: see palace_eref_entry_layout).  0: off (default).
 *   "scan_ref_lo", "scan_ref_hi"  palace_eref_scan_refs_indexed works on the refs [lo, hi) only (hi = 0: all); the rows of the others
 *                 read n_intervals = el = 0. */
int palace_eref_set_count_mode(palace_ctx *ctx, int mode, int64_t bucket_cap);
/* The count calls that follow take in only the keys whose top 7 bits -- one of 128 buckets of the key space -- are in the set
 * (bit b of mask128 = bucket b; default all).  For N GPUs that each hold all reads (the reference's threads share one table,
 * extract_ref.cpp:1269-1291): rank r counts the keys of its buckets -- its slices of the ">= 3" plane (4 MiB per bucket) are
 * then exact -- and the slices are gathered: no table exchange, no merge; the other keys are dropped where they are made, so
 * the partition kernels move the rank's share of the bytes.  Canonical keys thin out linearly over the key space (bucket b
 * holds (255 - 2b) / 16384 of them), so equal shares take buckets in mirrored pairs, e.g. {r, 2N-1-r} of every 2N. */
int palace_eref_set_key_buckets(palace_ctx *ctx, const uint32_t mask128[4]);
int palace_eref_set_option(palace_ctx *ctx, const char *name, int64_t value);

/* E5 + E6. For each ref: look the three indices of every position up in the table and run the
 * 500-base window scan (read_index + slide_window, extract_ref.cpp:813-903, 504-617).
 * one_min / three_min are int(500 * float(ratio)) as computed by the caller (extract_ref.cpp:
 * 513-514).  d_rows receives n_refs x 4 int32: n_intervals, el, ref_len, reserved(0). */
int palace_eref_scan_refs(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                          int64_t n_refs, int64_t total_bases, int one_min, int three_min,
                          int32_t *d_rows);

/* E5 with a per-DB probe index -- the analogue of the index file the reference builds once per DB
 * and then only reads (<fasta>.k32.index.dat, extract_ref.cpp:676-712, 1245-1251).  The index holds
 * the three channel indices of every valid ref position as 16-bit entries grouped by table slice, the
 * maps from a position to its entries, and a quarter of channel 0 once more as "sentinels" with their
 * positions: 19.5 B/position in device memory (3.9 GB for a 200 Mb DB); it depends on the ref set and
 * the coder only, not on the reads.  A DB of up to 2^32 positions.
 * palace_eref_scan_refs_indexed gives exactly the rows of palace_eref_scan_refs for the same table;
 * it replaces the per-position random probes by one sequential pass over 6.5 B/position that leaves a
 * hit BIT per entry (the plane slices in LDS), a scatter of the sentinels that hit, the exact pruning
 * of refs and 64-position chunks on those, and a gather of the three channels' bits for what is left.
 * Meant for a resident DB scanned against many samples; a one-shot run gains nothing from it. */
typedef struct palace_eref_probe_index palace_eref_probe_index;
int palace_eref_probe_index_build(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_refs,
                                  int64_t total_bases, palace_eref_probe_index **out);
int palace_eref_probe_index_free(palace_ctx *ctx, palace_eref_probe_index *ix);
int palace_eref_scan_refs_indexed(palace_ctx *ctx, const palace_eref_probe_index *ix, const uint8_t *d_bases,
                                  const int64_t *d_offsets, int64_t n_refs, int64_t total_bases, int one_min,
                                  int three_min, int32_t *d_rows);

/* Fuse Phase B's channel-0 probe of this DB into the count launch: while such an index is attached, a count call that runs as
 * the FINAL count (option final_count, one slab, the whole key space) tests the DB's positions of every fine bucket against
 * the bucket's final ">= 3" slice while that slice is still in LDS, and the next palace_eref_scan_refs_indexed with the same
 * index starts from those hits: no probe kernel, no second read of the plane.  Results are identical either way (any other
 * count call, a merge, an attach or a reset in between makes the scan probe for itself).  The index keeps the hit bits (one per
 * entry): attach it to ONE context at a time, and scan with it from one context at a time.  ix = NULL detaches. */
int palace_eref_attach_probe_index(palace_ctx *ctx, const palace_eref_probe_index *ix);

/* N GPUs that each counted a SHARE OF THE READS of one sample (no reference counterpart: its threads share one table,
 * extract_ref.cpp:1269-1291).  Phase B reads the table at the DB's keys only (read_index, :858-870), so what the ranks owe each other
 * is not their partial tables but their partial COUNTS of the DB's entries: with option "probe_all_sets" 2 the final count of a rank
 * leaves, for every entry of the attached index (built over the WHOLE DB on every rank), its count 0..3 in two bits -- 16 bits per
 * vector of eight entries, the four entry sets in one block of `counts_bytes` = 2 x `hits_bytes` (multiples of 256 x 840, so that 1 .. 8
 * ranks own equal, aligned shares).  The ranks exchange the block by shares (an all-to-all of counts_bytes / W per peer), each sums the
 * W parts of ITS share -- entry_hits_from_counts(d_parts, n_parts, part_stride, off, bytes): part p's counts of the block's bytes
 * [off, off + bytes) lie at d_parts + p * part_stride (the receive buffer of the all-to-all as it is); bit set iff the counts add up
 * to >= 3, exact because min(3, sum of min(3, c_r)) =
 * min(3, sum of c_r) -- into its share of the hit-bit block, the shares are all-gathered, and entry_hits_complete declares the block
 * whole: palace_eref_scan_refs_indexed then starts from it as from a count that tested every entry itself (options "scan_ref_lo" /
 * "scan_ref_hi": a rank scans its range of the refs).  No plane crosses a link: 162 + 81 MB per 200 Mb of DB instead of 2 x 512 MiB.
 * buffers_attach: the caller's device buffers (what its collectives address; NULL = the index's own, zeroed) stand in for the two blocks.
 * Under option "probe_all_sets" 2 a count call has no other form of result: it takes the binned, fused path whatever the size of the
 * share (one call per reset, one slab, whole key space, clean table) or fails with PALACE_ESTATE; with n = 0 (a rank without reads) it
 * zeroes the block.  entry_counts_valid: 1 when such a call of this context stands behind the block `ix` points at, else 0;
 * entry_hits_complete fails (PALACE_ESTATE) otherwise -- stale counts are never summed silently. */
int palace_eref_entry_layout(const palace_eref_probe_index *ix, size_t *counts_bytes, size_t *hits_bytes);
int palace_eref_entry_buffers_attach(palace_ctx *ctx, palace_eref_probe_index *ix, void *d_counts, void *d_hits);
int palace_eref_entry_buffers(const palace_eref_probe_index *ix, void **d_counts, void **d_hits);     /* where the two blocks lie now (counts: NULL before a first attach) */
int palace_eref_entry_hits_from_counts(palace_ctx *ctx, const palace_eref_probe_index *ix, const void *d_parts, int n_parts, size_t part_stride,
                                       size_t off, size_t bytes);
int palace_eref_entry_hits_complete(palace_ctx *ctx, const palace_eref_probe_index *ix, int64_t keys_counted);
int palace_eref_entry_counts_valid(const palace_ctx *ctx, const palace_eref_probe_index *ix);


/* Multi-GPU exchange of the count table (no reference counterpart: the reference shares one
 * table between std::threads, extract_ref.cpp:1269-1291).  planes() exposes the three device
 * buffers (each 2^29 bytes); merge_slices() folds `n_parts` partial tables laid out as
 * [plane][part][slice_bytes] into the caller's planes at byte offset `slice_off`, with the
 * saturating add  (a + b >= t  for t = 1, 2, 3)  done bit-parallel on the planes. */
int palace_eref_table_planes(palace_ctx *ctx, void **d_planes3, size_t *bytes_per_plane);
/* Use three caller-owned device buffers (each 2^29 bytes, 16-byte aligned) as the table from now on
 * (so a collective library can address them directly); the context no longer frees table memory. */
int palace_eref_table_attach(palace_ctx *ctx, void *const d_planes3[3]);
/* Contract for planes the caller can write (attached ones, or the pointers of palace_eref_table_planes): the library
 * remembers that a table_reset left every bit zero and lets the first count_reads after it skip reading the plane
 * slices.  Between a table_reset and the next count_reads the planes may therefore only be modified through library
 * calls -- or the caller says so: palace_eref_table_invalidate() makes the next count_reads read what is there.
 * (attach itself invalidates; merge_slices and count_reads do as well.) */
int palace_eref_table_invalidate(palace_ctx *ctx);
int palace_eref_table_merge_slices(palace_ctx *ctx, const void *d_parts, int n_parts,
                                   size_t slice_off, size_t slice_bytes);

/* The same exchange with two planes per peer instead of three: the unary planes of a partial table carry two bits per
 * key, the count's low bit (count>=1 ^ count>=2 ^ count>=3) and its high bit (count>=2).  palace_eref_table_pack_low
 * writes the low-bit plane of the context's table into d_low (2^29 bytes, 16-byte aligned); the sender ships d_low and
 * its count>=2 plane; palace_eref_table_merge_slices_packed folds parts laid out [2][n_parts][slice] = (low, high). */
int palace_eref_table_pack_low(palace_ctx *ctx, void *d_low);
int palace_eref_table_merge_slices_packed(palace_ctx *ctx, const void *d_parts, int n_parts, size_t slice_off,
                                          size_t slice_bytes);

/* The ">= 3" plane in sparse form, for exchanges between ranks that each hold a share of the key space (the level-1 buckets of
 * mask128, as in palace_eref_set_key_buckets): the plane of a sample is sparse (the 1M-contig sample sets 24 M of its 2^32 bits),
 * so a fine bucket (2^16 keys = 8 KiB of the plane) travels as the number of its set bits and their 16-bit offsets -- 48 MB for the
 * whole plane instead of 512 MiB.
 * pack:   d_counts[k] = set bits of the k-th fine bucket of the share (the share's level-1 buckets ascending, 512 fine buckets
 *         each: 512 * popcount(mask128) entries), d_first[k] = their exclusive prefix (k = 0 .. n: d_first[n] = total, one more
 *         entry than d_counts), d_keys[d_first[k] ..] = the offsets, ascending.  Keys beyond cap_keys are not written: the
 *         caller reads d_first[n] back (at its leisure) and repeats with more room, or ships the dense slices.
 * unpack: the reverse into THIS context's plane for the buckets of mask128 (every bit of those buckets is rewritten), from
 *         d_counts and d_keys as pack left them; cap_keys = the room d_keys has (keys the sender could not fit are not looked
 *         for: the caller, who sees the counts, discards such a result); d_first is scratch of n + 1 entries.
 * Both are enqueued on the context's stream; nothing is read back. */
int palace_eref_plane_pack(palace_ctx *ctx, const uint32_t mask128[4], uint32_t *d_counts, uint16_t *d_keys, int64_t cap_keys,
                           unsigned long long *d_first);
int palace_eref_plane_unpack(palace_ctx *ctx, const uint32_t mask128[4], const uint32_t *d_counts, const uint16_t *d_keys, int64_t cap_keys,
                             unsigned long long *d_first);

/* Test hooks: counts (0..3) of `n` indices; population count of each plane. */
int palace_eref_table_lookup(palace_ctx *ctx, const uint32_t *d_keys, int64_t n, uint8_t *d_counts);
int palace_eref_table_popcounts(palace_ctx *ctx, uint64_t out3[3]);

/* ---- generateGraph: BAM evidence -> conjugate graph (bin/generate_graph.cpp) ------------ */

/* Options of generate_graph.cpp:20-44 (defaults there; set by its getopt loop :573-593). */
typedef struct {
    int32_t max_end;        /* MAX_END        300 */
    int32_t min_mapq;       /* MIN_MAPQ         0 */
    int32_t max_nm;         /* MAX_NM           5 */
    int32_t enable_paired;  /* ENABLE_PAIRED    1 */
    int32_t both_order;     /* OUTPUT_BOTH_ORDER 0 */
    int32_t reserved;
    double max_span_frac;   /* MAX_SPAN_FRAC 0.80 */
} palace_graph_params;

/* Decoded primary-alignment columns, one entry per BAM record in file order (device pointers).
 * What htslib's bam1_t hands the reference at generate_graph.cpp:644-698, as structure of arrays:
 * ref_len = bam_cigar2rlen; read_len = getReadLength (:385-397); clip_s / clip_e = the soft clips
 * parseCigarReadInterval (:330-383) finds on the record's own CIGAR; nm = NM tag or 0; qkey = a
 * (clip_s = -1 marks a record without CIGAR ops, whose read interval is [0,0], :332); qkey = a
 * 64-bit key of the read name (equal names <=> equal keys, guaranteed by the caller);
 * sa_off[i]..sa_off[i+1] = the record's parsed SA items (empty when the tag is absent). */
typedef struct {
    int64_t n;
    const int32_t *tid, *pos, *mtid, *mpos, *nm, *ref_len, *read_len, *clip_s, *clip_e;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint64_t *qkey;
    const int32_t *sa_off;
} palace_bam_cols;

/* One parsed SA item (parseSAItem + parseCigarReadInterval on its CIGAR, :185-206, :744;
 * clip_s2 = -1 when the item's CIGAR text is empty).
 * tid2 < 0 when the item must be skipped (name equals the primary's contig :731, or is not in the
 * header :733-734); items that fail to parse are not listed at all. */
typedef struct {
    int32_t tid2, pos2, mapq2, nm2, clip_s2, clip_e2, len2, rev2;
} palace_sa_item;

/* One piece of candidate evidence (classify output / resolve input).  kind 0 = split read,
 * 1 = cross-contig pair.  cls: 0 score is 0 (a mapq is 0), 1 score > 0, 2 decided on the host by
 * libm (exp underflow region of computeLayoutScore, :432-461).  found: a layout exists (:916-938).
 * left/right/oL/oR are already canonical (:855-861); in_fastg is the :863 lookup.  sa_index: which item of
 * the record's SA list a split candidate comes from (0 for pairs) -- with `ord` the order in which the
 * reference meets the evidence (its --debug READS lists, :872, :1008, are in that order). */
typedef struct {
    int64_t ord;
    uint64_t qkey;
    int32_t left, right;
    int32_t mtid, ref_len;
    int32_t dL, dR;
    int32_t nmL, nmR;
    int16_t mapqL, mapqR;
    uint8_t kind, cls, found, in_fastg, oL, oR, pad0, pad1;
    int32_t sa_index;
} palace_graph_cand;

/* Aggregated edge (AggStats, :300-306): counts[0..3] = supplementCount, supplementCountNoFastg,
 * spanCount, spanCountNoFastg.  oL/oR: 0 = '+', 1 = '-'. */
typedef struct {
    int32_t left, right;
    uint32_t counts[4];
    uint8_t oL, oR, pad[6];
} palace_graph_edge;

/* G2-G5 + first half of G6.  Per record: filters (:647-649, :679), depth accumulation into
 * d_consumed[tid] (:654-662), split-read (:684-879) and read-pair (:887-1011) layout search.
 * Appends candidates to d_cands (capacity cand_cap) and returns their number in *n_cands_out.
 * d_tlen / d_trank: target lengths and the dense rank of each target name in byte order (used for
 * the `cR < cL` test :856 and for output order).  d_fastg: sorted keys
 * (tidA << 33 | tidB << 2 | (o1=='-') << 1 | (o2=='-')) of parseFastgFile's set (:119-169).
 * ord_base is the file ordinal of record 0 of this shard. */
int palace_graph_classify(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                          int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                          const uint64_t *d_fastg, int64_t n_fastg, const palace_graph_params *prm,
                          int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                          int64_t cand_cap, int64_t *n_cands_out);

/* The same, also returning how many of the candidates fall into the exp() underflow zone (cls == 2): the two counters
 * come back in one copy behind one wait, and palace_graph_resolve_ex needs no round trip of its own to learn the second. */
int palace_graph_classify_ex(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                             int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                             const uint64_t *d_fastg, int64_t n_fastg, const palace_graph_params *prm,
                             int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                             int64_t cand_cap, int64_t *n_cands_out, int64_t *n_border_out);

/* The same with the FASTG search narrowed: d_fastg_first[t] (n_targets + 1 entries, made once per sample by
 * palace_graph_fastg_offsets from the same sorted key array) = index of the first key whose left contig is >= t, so that a
 * candidate's look-up starts inside its left contig's two or three links instead of bisecting the whole set (:863-864 is a
 * std::set find per evidence; here ~22 dependent loads per candidate were most of the kernel's time).  NULL = bisect. */
int palace_graph_fastg_offsets(palace_ctx *ctx, const uint64_t *d_fastg, int64_t n_fastg, int32_t n_targets, uint32_t *d_first);
int palace_graph_classify_ix(palace_ctx *ctx, const palace_bam_cols *cols, const palace_sa_item *d_sa,
                             int32_t n_targets, const int32_t *d_tlen, const int32_t *d_trank,
                             const uint64_t *d_fastg, int64_t n_fastg, const uint32_t *d_fastg_first, const palace_graph_params *prm,
                             int64_t ord_base, uint64_t *d_consumed, palace_graph_cand *d_cands,
                             int64_t cand_cap, int64_t *n_cands_out, int64_t *n_border_out);

/* Second half: decide host-side borderline scores, apply the order-dependent rules
 * (hasSupplementEvidence gating :881-888, processedPairedReads "first in file order wins" and its
 * mate-contig depth quirk :890-893, :938), aggregate per canonical edge (:866-872, :1002-1008).
 * n_records_total bounds candidate ordinals.  Writes up to edge_cap edges (unsorted). */
int palace_graph_resolve(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands,
                         int64_t n_records_total, const palace_graph_params *prm, uint64_t *d_consumed,
                         palace_graph_edge *d_edges, int64_t edge_cap, int64_t *n_edges_out);

/* The same without the host in the loop.  n_border: number of candidates with cls == 2 (from palace_graph_classify_ex; summed
 * over the ranks whose candidates were gathered), 0 = none, so nothing is copied to the host; < 0 = unknown, look.
 * d_n_edges (device, 8 bytes, optional) receives the edge count in stream order; n_edges_out may be NULL, and then the call
 * only enqueues: no synchronisation (a count above edge_cap is then the reader's to detect: edges beyond it are dropped). */
int palace_graph_resolve_ex(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands, int64_t n_border,
                            int64_t n_records_total, const palace_graph_params *prm, uint64_t *d_consumed,
                            palace_graph_edge *d_edges, int64_t edge_cap, int64_t *d_n_edges, int64_t *n_edges_out);

/* The host-libm decision alone (computeLayoutScore's exp() underflow gate, :432-461) on the candidates of ONE classify call:
 * every found candidate with cls == 2 becomes cls 0 or 1 in place.  A rank of a multi-GPU run calls it on its own candidates
 * (n_border from palace_graph_classify_ex; 0 = nothing to do, nothing is copied) before they are gathered, so that what every
 * rank resolves carries no undecided candidate and palace_graph_resolve_ex can be given n_border = 0 without an exchange of
 * counts.  Synchronises the stream when n_border != 0. */
int palace_graph_score_border(palace_ctx *ctx, palace_graph_cand *d_cands, int64_t n_cands, int64_t n_border,
                              const palace_graph_params *prm);

/* G6 epilogue numbers (generate_graph.cpp:1029-1031): depth = consumed / max(1, len) and
 * cn = (int)floor(depth / avg_depth + 0.5) (0 when avg_depth <= 0), per target, in IEEE double. */
int palace_graph_copy_numbers(palace_ctx *ctx, const uint64_t *d_consumed, const int32_t *d_tlen,
                              int32_t n_targets, double avg_depth, int32_t *d_cn);

/* ---- N4: BGZF members inflated on the device ------------------------------------------------------------------------------ */

/* What htslib's bgzf layer does inside sam_read1 (generate_graph.cpp:644), for n_members BGZF members at once, one wavefront
 * each: member m's raw DEFLATE data are d_in[d_in_off[m] .. + d_in_len[m]) (behind the member's 18-byte header, in front of
 * its CRC32 / ISIZE trailer), its d_out_len[m] (= ISIZE, <= 65536) bytes go to d_out + d_out_off[m].  d_status[m] = 0: exactly
 * those bytes were written; non-zero: the decoder refused the member (malformed, or a size that does not fit) and the caller
 * lets zlib decide it on the host, as the loader's CPU decoder does.  CRC32 is not checked (the loader never did; htslib does).
 * The buffer behind d_in must extend at least 3 bytes past the last member's data (reads are whole dwords).  Enqueues only. */
int palace_bgzf_inflate(palace_ctx *ctx, const uint8_t *d_in, int64_t n_members, const int64_t *d_in_off, const int32_t *d_in_len,
                        const int64_t *d_out_off, const int32_t *d_out_len, uint8_t *d_out, int32_t *d_status);

/* ---- depth stage: `samtools depth <bam> | awk '{sum+=$3} END {print sum/NR}'` (palace:538-552) ------------------ */

/* The two numbers of that mean.  A match segment is one M / = / X CIGAR operation of a record whose UNMAP, SECONDARY,
 * QCFAIL and DUP flags are clear (samtools depth, default options: deletions and reference skips do not count): target,
 * 0-based reference position, length.  sum_out = total length of the segments (cut at the end of their contig),
 * covered_out = number of distinct reference positions they cover = the NR of the awk line.  d_tbase[t] = sum of the
 * lengths of targets 0..t-1 (int64, n_targets entries), total_len = sum of all lengths. */
int palace_depth_sum_covered(palace_ctx *ctx, int64_t n_segs, const int32_t *d_seg_tid, const int32_t *d_seg_pos,
                             const int32_t *d_seg_len, int32_t n_targets, const int32_t *d_tlen, const int64_t *d_tbase,
                             int64_t total_len, uint64_t *sum_out, uint64_t *covered_out);

/* The same per contig as well: d_contig_sum[t] / d_contig_covered[t] (device, n_targets entries each) = what
 * `tabix fetch(contig)` on the depth file yields, reduced to sum and count (create_sub_graph.py:186-234 reads exactly
 * that: mean = sum / count, weight = count). */
int palace_depth_per_contig(palace_ctx *ctx, int64_t n_segs, const int32_t *d_seg_tid, const int32_t *d_seg_pos,
                            const int32_t *d_seg_len, int32_t n_targets, const int32_t *d_tlen, const int64_t *d_tbase,
                            int64_t total_len, uint64_t *sum_out, uint64_t *covered_out, uint64_t *d_contig_sum,
                            uint64_t *d_contig_covered);

/* ---- matching: path / cycle decomposition of the conjugate graph ------------------------- */

/* M1. One greedy matching over the arcs of the conjugate graph, computed as rounds of locally
 * dominant arcs (an arc is taken when it is the best remaining arc of both its tail's out-slot
 * and its head's in-slot), which yields exactly the sequential greedy matching in rank order.
 * The reference's `matching` binary is absent; the call it serves is
 * palace:587-590 / 684-688 and the algorithm is this repository's own.
 * Vertices are oriented segments (2*seg + (orient=='-')); arcs are given in rank order
 * (arc id == rank, lower is better; an arc and its conjugate are adjacent), with CSR lists of
 * arc ids per tail (out_off/out_arcs) and per head (in_off/in_arcs).  d_alive: 1 B per vertex.
 * Outputs per vertex: d_next / d_prev (-1 = none) and d_next_arc (id of the arc leaving it). */
int palace_match_greedy(palace_ctx *ctx, int32_t n_vertices, int64_t n_arcs, const int32_t *d_src,
                        const int32_t *d_dst, const int64_t *d_out_off, const int32_t *d_out_arcs,
                        const int64_t *d_in_off, const int32_t *d_in_arcs, const uint8_t *d_alive,
                        int32_t *d_next, int32_t *d_prev, int32_t *d_next_arc, int32_t *rounds_out);

/* Host glue between generateGraph's numbers and palace_match_decompose, i.e. what the matching
 * executable does while it reads the SEG/JUNC text (palace_amd/host/matching_main.cpp): copies[s] =
 * max(1, cn[s]); every edge whose four counters sum to >= min_count (JUNC filter,
 * generateGraph.cpp:1056-1061) becomes the arc (left,oL)->(right,oR) plus its conjugate
 * (right,!oR)->(left,!oL) (make_final_fa.py:20-34), equal arcs merged with their weights added;
 * arcs come out in rank order (weight descending, class {arc, conjugate} ascending, (u, v) ascending).
 * Host arrays; src/dst/weight need room for 2 * n_edges arcs.  Pure host code (no GPU work). */
int palace_match_arcs_from_edges(const int32_t *cn, int32_t n_segs, const palace_graph_edge *edges, int64_t n_edges,
                                 int32_t min_count, int64_t *copies, int32_t *src, int32_t *dst, int64_t *weight,
                                 int64_t *n_arcs_out);

/* M1, whole decomposition: `iterations` rounds of {greedy matching on the GPU, read the paths and
 * cycles off the successor links, charge copy numbers, drop exhausted segments} (+ one copy-number
 * blind round when `aggressive`).  Host arrays in: copies[n_segs] (>= 1), arcs in rank order
 * (src/dst oriented-vertex ids, an arc and its conjugate adjacent).  Result: components in emission
 * order; component c holds verts[off[c] .. off[c+1]) in path order (cycles rotated to their
 * smallest vertex, conjugate representative chosen), kind[c] = 0 path / 1 cycle, iter[c] = round,
 * open_at[c] = position after the cycle's weakest arc (where -b opens it; 0 for paths).
 * Duplicate and later-round singleton components are NOT filtered here (the caller formats). */
typedef struct palace_match_result palace_match_result;
/* Tuning knob (results are identical for every setting).  The decomposition runs on the device, one launch per phase, enqueued
 * a group of rounds at a time with a fixed number of matching iterations per round (7 in the first round, 4 later; the
 * iterations behind a round's fixed point return at once); the host looks at the state after each group, stops as soon as no
 * segment keeps a copy, and redoes the decomposition with a check after every batch of iterations should a round not have
 * settled.  "iters_per_round" overrides the number of iterations (0 = defaults, at most 64; 1 forces the checked path);
 * "decomp_grid" workgroups of the decomposition's arc- and vertex-sized phases (0 = default 2048: chains of dependent random
 * look-ups, bounded by how many are in flight; 256 costs a saturating kernel on another stream less, see bench/step.py).
 * "one_word_keys" 0: palace_stage04_match ranks its arcs by the two-word key in every case (default 1: by a one-word form of
 * the same order -- weight | path-backed | class of (tail, head) -- whenever the sample's arcs fit it, which saves the second
 * proposal pass of every matching iteration: half of the atomics and a third of the look-ups). */
int palace_match_set_option(palace_ctx *ctx, const char *name, int64_t value);
int palace_match_decompose(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, int64_t n_arcs,
                           const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive,
                           palace_match_result **out);
/* The same with `compact` != 0: only components that hold a segment with at least one arc are listed (same order, same
 * fields); the segments without any arc -- each a one-vertex path of round 0, and again of the extra round when
 * `aggressive`, that a full result lists in first-vertex order between the others -- are given as one bit per segment
 * (palace_match_result_bare: ceil(n_segs / 64) words, bit s of word s / 64; palace_match_result_bare_count of them).  A
 * graph of a million segments of which a few percent touch a junction is the normal case: the full listing is almost
 * all single-vertex entries. */
int palace_match_decompose_ex(palace_ctx *ctx, int32_t n_segs, const int64_t *copies, int64_t n_arcs,
                              const int32_t *src, const int32_t *dst, int32_t iterations, int32_t aggressive,
                              int32_t compact, palace_match_result **out);
const uint64_t *palace_match_result_bare(const palace_match_result *r);
int64_t palace_match_result_bare_count(const palace_match_result *r);
int64_t palace_match_result_count(const palace_match_result *r);
const int64_t *palace_match_result_offsets(const palace_match_result *r);
const int32_t *palace_match_result_verts(const palace_match_result *r);
const uint8_t *palace_match_result_kind(const palace_match_result *r);
const int32_t *palace_match_result_iter(const palace_match_result *r);
const int32_t *palace_match_result_open_at(const palace_match_result *r);
void palace_match_result_free(palace_match_result *r);

/* ---- stage 04 resident in HBM: filter_graph.py + matching without the text files in between ----------------------- */

/* What the pipeline does between generateGraph's numbers and `all_result` (palace:566-600) is a selection on the graph
 * (share/palace/scripts/filter_graph.py) and `matching` on what the selection leaves.  For a sample whose edges are already
 * in HBM (palace_graph_resolve) both run on the device; the host formats.  Per-sample inputs, parsed once like the BAM
 * columns (host arrays; create copies them to the device): */
typedef struct {
    int32_t n_segs;            /* contigs = BAM targets */
    int32_t min_count;         /* MIN_COUNT (generate_graph.cpp:40): a JUNC line exists for an edge whose counters sum to >= it (:1061) */
    const uint8_t *seed;       /* per contig: bit 0 in blast_segs (filter_graph.py:66-94), bit 1 in gene_res (:99-102), bit 2 score > threshold (:104-112) */
    const int32_t *tlen;       /* target lengths: a contig has a SEG line iff its length is > 0 (generate_graph.cpp:1019-1050) */
    const int32_t *rank;       /* dense rank of the names in byte order = order of the SEG lines of `_graph.txt` */
    const int32_t *name_len;   /* the length token of each name (get_edge_len, filter_graph.py:50-52) */
    int64_t n_paths;           /* lines of contigs.paths that are not NODE headers */
    const int64_t *path_off;   /* n_paths + 1 offsets into path_tok */
    const int32_t *path_tok;   /* 2 * contig + (orientation == '-'); -1 = an id no contig has */
} palace_stage04_inputs;

typedef struct palace_stage04 palace_stage04;
int palace_stage04_create(palace_ctx *ctx, const palace_stage04_inputs *in, palace_stage04 **out);
int palace_stage04_destroy(palace_ctx *ctx, palace_stage04 *s);

/* Optional: allocate now what a filter call with this edge bound will need (hundreds of MB for a million contigs; the
 * allocation alone takes tens of milliseconds), e.g. while the caller is still decoding its BAM.  The object's memory belongs
 * to the device: it may be created and reserved through one context and used through another. */
int palace_stage04_reserve(palace_ctx *ctx, palace_stage04 *s, int64_t edge_bound);

/* B2 in memory (filter_graph.py:201-264): which junctions and which SEG lines `_filtered_graph.txt` holds.  d_edges and
 * d_n_edges (device, 8 bytes) are what palace_graph_resolve_ex leaves; edge_bound is a bound on the count the host knows
 * (the number of candidates): it sizes the tables.  Only enqueues.  Per edge a flag byte: 1 = the JUNC line exists,
 * 2 = kept by pass 2 (:223-233), 4 = kept by pass 3 (:237-245); per contig: 1 = its SEG line is selected (seed, or end
 * of a kept junction), 2 = rescued through contigs.paths only (:126-151, written with the ` 0 1.0 0` tail), 4 = core seed.
 * The file lists the selected SEG lines, then the rescued ones, then the pass-2 junctions and the pass-3 junctions that are
 * not pass-2 ones, each group in `_graph.txt` order. */
int palace_stage04_filter(palace_ctx *ctx, palace_stage04 *s, const palace_graph_edge *d_edges, const int64_t *d_n_edges,
                          int64_t edge_bound);
/* the flags on the host (waits for the stream); n_edges <= edge_bound entries of the edge flags */
int palace_stage04_flags(palace_ctx *ctx, palace_stage04 *s, uint8_t *h_seg_flags, uint8_t *h_edge_flags, int64_t n_edges);
/* counts[8] = edges, JUNC lines, junctions kept by pass 2, further junctions kept by pass 3, selected SEG lines, rescued SEG
 * lines, merged arcs (-1 before palace_stage04_match), segments of the filtered graph (waits for the stream); also reports
 * what the reference script dies on: a contigs.paths id that names no contig, a rescued contig without SEG line */
int palace_stage04_counts(palace_ctx *ctx, palace_stage04 *s, int64_t counts[8]);

/* M1 on the filtered graph: `matching -g <filtered graph> -i <iterations> [-l contigs.paths] [--aggressive]` (palace:587-590)
 * on the device.  Segments are numbered as the filtered file lists them; copies = max(1, d_cn[contig]); arcs = kept junctions
 * (weight n1 + n2) + conjugates, and with use_paths the path-backed arcs.  Only enqueues (at most 10 iterations +
 * aggressive per filter call's reservation). */
int palace_stage04_match(palace_ctx *ctx, palace_stage04 *s, const palace_graph_edge *d_edges, const int32_t *d_cn,
                         int32_t iterations, int32_t aggressive, int32_t use_paths);
/* Waits and hands out the result in the compact form of palace_match_decompose_ex (vertices 2 * segment + orientation,
 * segments = ids of the filtered graph; bare segments as bits), owned by `s` (valid until the next match call or destroy;
 * palace_match_result_free on it does nothing), and contig_of[filtered segment] -> contig (n_segs_filtered entries). */
int palace_stage04_result(palace_ctx *ctx, palace_stage04 *s, palace_match_result **out, const int32_t **contig_of_out,
                          int64_t *n_segs_filtered_out);

#ifdef __cplusplus
}
#endif
#endif /* PALACE_HIP_H */
