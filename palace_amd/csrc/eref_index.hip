// eref on gfx950: the per-DB probe index (the analogue of <fasta>.k32.index.dat kept in HBM) and the entry-count exchange of N ranks
#include "eref_common.hpp"

namespace palace {

// ------------------------------------------------------------------------------------------
// E5 with a probe index: the reference reads the three indices of every ref position from a file it
// built once per DB (<fasta>.k32.index.dat, 12 B/position, extract_ref.cpp:676-712) instead of
// recomputing them.  The analogue here is built once per DB and kept in HBM, and is laid out for what
// the scan does with it -- test EVERY position's index against the ">= 3" plane, then look at the few
// refs that can pass:
//   entry sets   four lists of 16-bit entries (index & 0xffff) grouped by the count kernel's fine buckets
//                (index >> 16; a bucket's entries start on a multiple of 8): channel 0, 1 and 2 of every
//                valid position, and the SENTINELS -- channel 0 of the positions = 0 (mod 4) of every ref.
//                A probe tests each group of four buckets against its 32 KiB slice of plane 3 in LDS --
//                2 B per position and channel read sequentially instead of one random 128-byte line each --
//                and leaves one hit BIT per entry, in entry order (eref_probe_sets_kernel; for channel 0
//                the count launch can do it, palace_eref_attach_probe_index).
//   sentinels    `pos_s` (entry -> sentinel ordinal = position id / 4): the sentinels' hits (a quarter of
//                channel 0's, 2.3 M at the 1M-contig sample) are scattered to position order.  A window
//                passes only with >= three_min of its 500 positions hit in ALL channels, so it misses at most
//                500 - three_min channel-0 hits, so of its >= 124 sentinels at least three_min - 376 hit:
//                eref_need_kernel with that threshold on the sentinel bits marks, exactly as before, the refs
//                and 64-position chunks that can lie in a passing window (96 % of the refs have none).
//   entry maps   `eix[c]` (position id -> entry of channel c, ~0 = none): for the needed chunks only, the hit
//                bits of the three channels are GATHERED from the entry-order bit arrays (25 MB each:
//                cache resident) -- eref_gather_hits_kernel -- where round 4 / early round 5 scattered all
//                9 M channel-0 hits into a byte per position (0.45 ms) and probed channels 1 and 2 of the needed
//                chunks at random in the 512 MB plane (0.47 ms).
// Entry-order hit bits are also what ranks could exchange when the key space is split between GPUs.
// ------------------------------------------------------------------------------------------
struct IndexBuild {
    unsigned long long *count;                  // [kSets][65536]
    const unsigned long long *first;            // [kSets][65537] (PASS 1)
    uint16_t *keys16[kSets];
    uint32_t *eix[3];                           // position id -> entry of the channel
    uint32_t *pos_s;                            // sentinel entry -> position id / 4
    uint32_t *epos[kSets];                      // (build only) entry -> position id: what eref_probe_index_canon_kernel orders a bucket's entries by
};
template <int PASS>   // 0: count positions per set and fine bucket, 1: place them
__global__ __launch_bounds__(256) void eref_probe_index_kernel(const uint8_t *__restrict__ bases,
                                                               const int64_t *__restrict__ offsets, int64_t n_refs,
                                                               const int64_t *__restrict__ tile_pre,
                                                               const int64_t *__restrict__ word_pre, CoderMasks masks, IndexBuild ib)
{
    const int64_t tile = blockIdx.x;
    if (tile >= tile_pre[n_refs]) return;
    const int64_t r = find_seq(tile_pre, n_refs, tile);
    const int64_t beg = offsets[r], len = offsets[r + 1] - beg;
    const int64_t npos = len - 31;
    const int64_t n_chunks = (len + 63) / 64;
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    constexpr int per_wave = kTileChunks / 4;
    const int64_t c0 = (tile - tile_pre[r]) * kTileChunks + static_cast<int64_t>(wv_id) * per_wave;
    if (c0 >= n_chunks) return;
    const int64_t c1 = min(n_chunks, c0 + per_wave);
    const uint8_t *s = bases + beg;
    const int64_t wbase = word_pre[r];
    Streams lo = ballot_streams(s, c0 * 64 + lane, len);
    for (int64_t c = c0; c < c1; c++) {
        Streams hi = ballot_streams(s, (c + 1) * 64 + lane, len);
        const int64_t j = c * 64 + lane;
        const uint32_t ok = window32(lo.ok, hi.ok, lane);
        if (j < npos && ok == 0xffffffffu) {
            uint32_t key[3];
            kmer_keys(masks, window32(lo.p0, hi.p0, lane), window32(lo.p1, hi.p1, lane), window32(lo.p2, hi.p2, lane), key);
            const uint32_t posid = static_cast<uint32_t>((wbase + c) * 64 + lane);
#pragma unroll
            for (int set = 0; set < kSets; set++) {
                const uint32_t k = key[set == kSentinelSet ? 0 : set];
                if (k == 0) continue;                             // index 0 means "none" (extract_ref.cpp:861)
                if (set == kSentinelSet && (lane & (kSentinelStride - 1))) continue;      // (refs start on word boundaries: lane = position mod 64)
                const uint32_t b = k >> 16;                       // fine bucket of the count kernel; four of them are one probe group
                const unsigned long long at = atomicAdd(&ib.count[static_cast<size_t>(set) * kIndexGroups + b], 1ull);
                if (PASS == 1) {
                    const unsigned long long e = ib.first[static_cast<size_t>(set) * (kIndexGroups + 1) + b] + at;
                    ib.keys16[set][e] = static_cast<uint16_t>(k);
                    if (ib.epos[set]) ib.epos[set][e] = posid;
                    if (set == kSentinelSet) ib.pos_s[e] = posid / kSentinelStride;
                    else ib.eix[set][posid] = static_cast<uint32_t>(e);
                }
            }
        }
        lo = hi;
    }
}

// The placement above hands out a bucket's slots by atomicAdd: WHICH slot a position gets depends on the order its thread got there,
// i.e. two builds of one DB agree on the buckets and disagree inside them.  For everything one GPU does that is immaterial; ranks that
// sum partial counts entry by entry (palace_eref_entry_hits_from_counts) need the same entry to mean the same DB position everywhere.
// So every bucket's entries are put into position order afterwards: one workgroup per (set, bucket), a bitonic sort of
// (position id << 16 | key) in LDS, keys / maps rewritten.  Buckets of more than kCanonMax entries (a DB of gigabases) are left as
// they are and counted: the index then refuses the partial-count mode.
constexpr int kCanonMax = 8192, kCanonThreads = 1024;
__global__ __launch_bounds__(kCanonThreads) void eref_probe_index_canon_kernel(IndexBuild ib, const unsigned long long *__restrict__ count,
                                                                                 unsigned int *__restrict__ not_canon)
{
    __shared__ unsigned long long e[kCanonMax];
    const uint32_t set = blockIdx.y, b = blockIdx.x;
    const unsigned long long n = count[static_cast<size_t>(set) * kIndexGroups + b];
    if (n <= 1) return;
    if (n > kCanonMax) { if (threadIdx.x == 0) atomicAdd(not_canon, 1u); return; }
    const unsigned long long f0 = ib.first[static_cast<size_t>(set) * (kIndexGroups + 1) + b];
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t i = threadIdx.x; i < N; i += kCanonThreads)
        e[i] = i < n ? (static_cast<unsigned long long>(ib.epos[set][f0 + i]) << 16) | ib.keys16[set][f0 + i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < N; i += kCanonThreads) {
                const uint32_t p = i ^ j;
                if (p > i) {
                    const unsigned long long a = e[i], c = e[p];
                    if (((i & k) == 0) == (a > c)) { e[i] = c; e[p] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += kCanonThreads) {
        const unsigned long long v = e[i];
        const uint32_t posid = static_cast<uint32_t>(v >> 16);
        ib.keys16[set][f0 + i] = static_cast<uint16_t>(v);
        if (set == kSentinelSet) ib.pos_s[f0 + i] = posid / kSentinelStride;
        else ib.eix[set][posid] = static_cast<uint32_t>(f0 + i);
    }
}

// exclusive prefix of the 65536 fine-bucket counts, each rounded up to a multiple of 8 (one workgroup, 64 buckets per thread);
// first[65536] = total (padded)
__global__ __launch_bounds__(1024) void eref_bucket_prefix_kernel(const unsigned long long *__restrict__ count_all,
                                                                  unsigned long long *__restrict__ first_all)
{
    const unsigned long long *count = count_all + static_cast<size_t>(blockIdx.x) * kIndexGroups;       // one workgroup per entry set
    unsigned long long *first = first_all + static_cast<size_t>(blockIdx.x) * (kIndexGroups + 1);
    __shared__ unsigned long long part[1024];
    constexpr int kPer = kIndexGroups / 1024;
    unsigned long long sum = 0;
    for (int i = 0; i < kPer; i++) sum += (count[threadIdx.x * kPer + i] + 7ull) & ~7ull;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
    for (int i = 0; i < kPer; i++) { first[threadIdx.x * kPer + i] = run; run += (count[threadIdx.x * kPer + i] + 7ull) & ~7ull; }
    if (threadIdx.x == 1023) first[kIndexGroups] = run;
}

}  // namespace palace

using namespace palace;

extern "C" {

int palace_eref_probe_index_build(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_refs,
                                  int64_t total_bases, palace_eref_probe_index **out)
{
    PALACE_REQUIRE(ctx && out && n_refs >= 0 && total_bases >= 0, "bad argument");
    if (!ctx->coder_set) { set_error("palace_eref_probe_index_build: coder not set"); return PALACE_ESTATE; }
    PALACE_REQUIRE(n_refs == 0 || (d_bases && d_offsets), "null device pointer");
    PALACE_REQUIRE(n_refs < (1ll << 31), "too many refs for one launch");
    PALACE_REQUIRE(total_bases + 64 * (n_refs + 1) < (1ll << 32) - 1, "position ids must fit in 32 bits");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    palace_eref_probe_index *ix = new palace_eref_probe_index();
    ix->n_refs = n_refs; ix->total_bases = total_bases; ix->masks = ctx->masks;
    unsigned long long *count = nullptr;                  // a counter per fine bucket, only during the build
    uint32_t *epos[kSets] = {nullptr, nullptr, nullptr, nullptr};
    auto done = [&](int rc) {
        (void)hipStreamSynchronize(ctx->stream);
        if (count) (void)hipFree(count);
        for (uint32_t *p : epos) if (p) (void)hipFree(p);
        if (rc) palace_eref_probe_index_free(ctx, ix); else *out = ix;
        return rc;
    };
#define TRY_OR_DONE(expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e__)); return done(PALACE_EHIP); } \
    } while (0)
    TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&ix->first), static_cast<size_t>(kSets) * (kIndexGroups + 1) * 8));
    TRY_OR_DONE(hipMemsetAsync(ix->first, 0, static_cast<size_t>(kSets) * (kIndexGroups + 1) * 8, ctx->stream));
    if (n_refs == 0) return done(PALACE_OK);
    ScanBuffers b;
    int rc = scan_buffers(ctx, d_offsets, n_refs, total_bases, &b);     // tile_pre / word_pre exactly as the scans lay them out
    if (rc) return done(rc);
    const size_t count_bytes = static_cast<size_t>(kSets) * kIndexGroups * 8;
    TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&count), count_bytes));
    TRY_OR_DONE(hipMemsetAsync(count, 0, count_bytes, ctx->stream));
    IndexBuild ib{};
    ib.count = count;
    hipLaunchKernelGGL(eref_probe_index_kernel<0>, dim3(static_cast<unsigned>(b.max_tiles)), dim3(256), 0, ctx->stream,
                       d_bases, d_offsets, n_refs, b.tile_pre, b.word_pre, ctx->masks, ib);
    hipLaunchKernelGGL(eref_bucket_prefix_kernel, dim3(kSets), dim3(1024), 0, ctx->stream, count, ix->first);
    TRY_OR_DONE(hipGetLastError());
    for (int k = 0; k < kSets; k++)
        TRY_OR_DONE(hipMemcpyAsync(&ix->n_entries[k], ix->first + static_cast<size_t>(k) * (kIndexGroups + 1) + kIndexGroups, 8, hipMemcpyDeviceToHost, ctx->stream));
    TRY_OR_DONE(hipStreamSynchronize(ctx->stream));
    ix->hit_bytes_size = static_cast<size_t>(b.max_words) * 64;                                     // (position ids run over the words of the hit bitmap)
    for (int k = 0; k < kSets; k++) {
        if (ix->n_entries[k] >= (1ull << 32) - 256) { set_error("palace_eref_probe_index_build: too many entries for 32-bit entry ids"); return done(PALACE_EINVAL); }
        const unsigned long long n128 = (ix->n_entries[k] + 127) / 128 * 128;
        TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&ix->keys16[k]), (n128 + 8) * 2));
        TRY_OR_DONE(hipMemsetAsync(ix->keys16[k], 0, (n128 + 8) * 2, ctx->stream));
        ix->ehits_bytes[k] = static_cast<size_t>(n128 / 8);
        ib.keys16[k] = ix->keys16[k];
    }
    for (int c = 0; c < 3; c++) {
        TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&ix->eix[c]), ix->hit_bytes_size * 4 + 64));
        TRY_OR_DONE(hipMemsetAsync(ix->eix[c], 0xff, ix->hit_bytes_size * 4 + 64, ctx->stream));
        ib.eix[c] = ix->eix[c];
    }
    {
        const unsigned long long n128 = (ix->n_entries[kSentinelSet] + 127) / 128 * 128;
        TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&ix->pos_s), (n128 + 8) * 4));
        TRY_OR_DONE(hipMemsetAsync(ix->pos_s, 0xff, (n128 + 8) * 4, ctx->stream));
        ib.pos_s = ix->pos_s;
    }
    {   // the hit bits a count launch leaves (channel 0's, or every set's): one block, each set's part 256-byte aligned with 16 spare bytes
        size_t at[kSets], total = 0;
        for (int k = 0; k < kSets; k++) { at[k] = total; total += align_up(ix->ehits_bytes[k] + 16, 256); }
        total = align_up(total, kEntryBlockAlign);          // (so that 1 .. 8 ranks can each own an equal, 256-byte aligned share of the block)
        ix->entry_hits_bytes = total;
        const size_t at_sent = total;
        total += align_up(ix->hit_bytes_size / kSentinelStride + 16, 256);
        uint8_t *blk = nullptr;
        TRY_OR_DONE(hipMalloc(reinterpret_cast<void **>(&blk), total));
        TRY_OR_DONE(hipMemsetAsync(blk, 0, total, ctx->stream));
        for (int k = 0; k < kSets; k++) { ix->ehits_own[k] = blk + at[k]; ix->set_at[k] = at[k]; }
        ix->sent_bytes_own = blk + at_sent;
        ix->ehits_own_bytes = total;
        ix->hits_block = blk;
    }
    TRY_OR_DONE(hipMemsetAsync(count, 0, count_bytes, ctx->stream));
    ib.first = ix->first;
    for (int k = 0; k < kSets; k++) {                      // entry -> position id, for the ordering pass only (4 B per entry: 2.6 GB for a 200 Mb DB)
        if (hipMalloc(reinterpret_cast<void **>(&epos[k]), (ix->n_entries[k] + 8) * 4) != hipSuccess) { epos[k] = nullptr; (void)hipGetLastError(); }
        ib.epos[k] = epos[k];
    }
    const bool can_order = epos[0] && epos[1] && epos[2] && epos[3];
    if (!can_order) for (int k = 0; k < kSets; k++) ib.epos[k] = nullptr;
    hipLaunchKernelGGL(eref_probe_index_kernel<1>, dim3(static_cast<unsigned>(b.max_tiles)), dim3(256), 0, ctx->stream,
                       d_bases, d_offsets, n_refs, b.tile_pre, b.word_pre, ctx->masks, ib);
    TRY_OR_DONE(hipGetLastError());
    if (can_order) {
        TRY_OR_DONE(hipMemsetAsync(ctx->d_small, 0, 8, ctx->stream));
        hipLaunchKernelGGL(eref_probe_index_canon_kernel, dim3(kIndexGroups, kSets), dim3(kCanonThreads), 0, ctx->stream, ib, count,
                           reinterpret_cast<unsigned int *>(ctx->d_small));
        TRY_OR_DONE(hipGetLastError());
        unsigned int not_canon = 1;
        TRY_OR_DONE(hipMemcpyAsync(&not_canon, ctx->d_small, 4, hipMemcpyDeviceToHost, ctx->stream));
        TRY_OR_DONE(hipStreamSynchronize(ctx->stream));
        ix->canonical = not_canon == 0;
    }
#undef TRY_OR_DONE
    return done(PALACE_OK);
}

int palace_eref_probe_index_free(palace_ctx *ctx, palace_eref_probe_index *ix)
{
    if (!ix) return PALACE_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    if (ctx && ctx->probe_ix == ix) ctx->probe_ix = nullptr;
    if (ctx && ctx->c0_hits_ix == ix) ctx->c0_hits_ix = nullptr;
    if (ix->first) (void)hipFree(ix->first);
    for (int k = 0; k < palace::kSets; k++) if (ix->keys16[k]) (void)hipFree(ix->keys16[k]);
    for (int c = 0; c < 3; c++) if (ix->eix[c]) (void)hipFree(ix->eix[c]);
    if (ix->pos_s) (void)hipFree(ix->pos_s);
    if (ix->hits_block) (void)hipFree(ix->hits_block);
    if (ix->counts_block) (void)hipFree(ix->counts_block);
    delete ix;
    return PALACE_OK;
}

int palace_eref_attach_probe_index(palace_ctx *ctx, const palace_eref_probe_index *ix)
{
    PALACE_REQUIRE(ctx, "ctx is null");
    if (ix) PALACE_REQUIRE(std::memcmp(&ix->masks, &ctx->masks, sizeof(CoderMasks)) == 0 && ctx->coder_set, "probe index was built with another coder");
    ctx->probe_ix = ix;
    if (!ix) ctx->c0_hits_ix = nullptr;
    return PALACE_OK;
}

/* ---- N GPUs that each counted a share of the READS: partial counts of the DB's entries instead of partial planes ---- */
namespace {
// parts[p][j] (u16 = eight 2-bit partial counts of the entries 8 j .. 8 j + 7), p < n_parts -> hit byte j: bit e set iff the counts of
// entry e add up to 3 or more.  Eight u16 (16 bytes) per thread and part.
__global__ __launch_bounds__(256) void entry_sum_kernel(const uint4 *__restrict__ parts, int n_parts, size_t part_stride16, size_t n16,
                                                        unsigned long long *__restrict__ hits)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint32_t sum[8][8];                                              // [u16 of the vector][entry]
#pragma unroll
        for (int h = 0; h < 8; h++)
#pragma unroll
            for (int e = 0; e < 8; e++) sum[h][e] = 0;
        for (int p = 0; p < n_parts; p++) {
            const uint4 v = parts[static_cast<size_t>(p) * part_stride16 + i];
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const uint32_t w = (d[h >> 1] >> (16 * (h & 1))) & 0xffffu;
#pragma unroll
                for (int e = 0; e < 8; e++) sum[h][e] += (w >> (2 * e)) & 3u;
            }
        }
        unsigned long long out = 0;
#pragma unroll
        for (int h = 0; h < 8; h++) {
            uint32_t byte = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) byte |= (sum[h][e] >= 3u ? 1u : 0u) << e;
            out |= static_cast<unsigned long long>(byte) << (8 * h);
        }
        hits[i] = out;
    }
}
}  // namespace

int palace_eref_entry_layout(const palace_eref_probe_index *ix, size_t *counts_bytes, size_t *hits_bytes)
{
    PALACE_REQUIRE(ix && counts_bytes && hits_bytes, "null argument");
    *hits_bytes = ix->entry_hits_bytes;
    *counts_bytes = 2 * ix->entry_hits_bytes;
    return PALACE_OK;
}

int palace_eref_entry_buffers_attach(palace_ctx *ctx, palace_eref_probe_index *ix, void *d_counts, void *d_hits)
{
    PALACE_REQUIRE(ctx && ix, "null argument");
    PALACE_REQUIRE((reinterpret_cast<uintptr_t>(d_counts) | reinterpret_cast<uintptr_t>(d_hits)) % 256 == 0, "buffers must be 256-byte aligned");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->c0_hits_ix == ix) ctx->c0_hits_ix = nullptr;
    uint8_t *counts = static_cast<uint8_t *>(d_counts);
    if (!counts) {                                                     // the index's own count block (made on first use: 2 x the hit bits), zero
        if (!ix->counts_block) PALACE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ix->counts_block), 2 * ix->entry_hits_bytes));
        counts = ix->counts_block;
        PALACE_HIP_TRY(hipMemsetAsync(counts, 0, 2 * ix->entry_hits_bytes, ctx->stream));
    }
    ctx->counts_ptr = nullptr;                                         // (whatever this context counted lies in the blocks attached before)
    uint8_t *hits = d_hits ? static_cast<uint8_t *>(d_hits) : ix->hits_block;
    for (int k = 0; k < kSets; k++) { ix->ecnt_own[k] = counts + 2 * ix->set_at[k]; ix->ehits_own[k] = hits + ix->set_at[k]; }
    return PALACE_OK;
}

int palace_eref_entry_buffers(const palace_eref_probe_index *ix, void **d_counts, void **d_hits)
{
    PALACE_REQUIRE(ix && d_counts && d_hits, "null argument");
    *d_counts = ix->ecnt_own[0];
    *d_hits = ix->ehits_own[0];
    return PALACE_OK;
}

int palace_eref_entry_hits_from_counts(palace_ctx *ctx, const palace_eref_probe_index *ix, const void *d_parts, int n_parts, size_t part_stride,
                                       size_t off, size_t bytes)
{
    PALACE_REQUIRE(ctx && ix && d_parts && n_parts > 0, "bad argument");
    PALACE_REQUIRE(off % 16 == 0 && bytes % 16 == 0 && part_stride % 16 == 0 && reinterpret_cast<uintptr_t>(d_parts) % 16 == 0, "16-byte granules");
    PALACE_REQUIRE(off + bytes <= 2 * ix->entry_hits_bytes && bytes <= part_stride, "range outside the count block");
    if (bytes == 0) return PALACE_OK;
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(entry_sum_kernel, dim3(kCUs * 8), dim3(256), 0, ctx->stream, static_cast<const uint4 *>(d_parts), n_parts, part_stride / 16,
                       bytes / 16, reinterpret_cast<unsigned long long *>(ix->ehits_own[0] + off / 2));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_eref_entry_hits_complete(palace_ctx *ctx, const palace_eref_probe_index *ix, int64_t keys_counted)
{
    PALACE_REQUIRE(ctx && ix, "null argument");
    PALACE_REQUIRE(ctx->probe_ix == ix && ctx->probe_all_sets == 2, "the index is not attached to this context with option probe_all_sets 2");
    if (!ctx->counts_ptr || ctx->counts_ptr != ix->ecnt_own[0]) {
        set_error("palace_eref_entry_hits_complete: no count call of this context has left its partial counts in the index's count block since the last reset");
        return PALACE_ESTATE;
    }
    ctx->c0_hits_ix = ix;
    ctx->hits_mask = (1u << kSets) - 1;
    ctx->sent_scattered = false;                                       // (the scan carries the sentinels' hits to position order)
    ctx->keys_counted = keys_counted;                                  // key instances of ALL ranks (what the scan's pruning goes by); -1: unknown
    return PALACE_OK;
}

int palace_eref_entry_counts_valid(const palace_ctx *ctx, const palace_eref_probe_index *ix)
{
    return ctx && ix && ctx->counts_ptr && ctx->counts_ptr == ix->ecnt_own[0] ? 1 : 0;
}

}  // extern "C"
