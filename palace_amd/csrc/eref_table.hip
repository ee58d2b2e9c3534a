// eref on gfx950: the count table as three bit planes -- coder, reset, planes of the caller, merges of partial tables, the sparse form of the ">= 3" plane, look-ups
#include "eref_common.hpp"

namespace palace {

// ------------------------------------------------------------------------------------------
// table utilities
// ------------------------------------------------------------------------------------------
__global__ void table_lookup_kernel(const uint32_t *__restrict__ keys, int64_t n,
                                    const uint32_t *__restrict__ p1, const uint32_t *__restrict__ p2,
                                    const uint32_t *__restrict__ p3, uint8_t *__restrict__ out)
{
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = keys[i], w = k >> 5, b = k & 31;
    out[i] = ((p1[w] >> b) & 1) + ((p2[w] >> b) & 1) + ((p3[w] >> b) & 1);
}

__global__ __launch_bounds__(256) void plane_popcount_kernel(const uint4 *__restrict__ plane, size_t n16,
                                                             unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        uint4 v = plane[i];
        acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (int d = 32; d; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// The unary planes of a partial table (p3 subset of p2 subset of p1) hold two bits of information per key: the
// count's low bit p1 ^ p2 ^ p3 and its high bit p2.  Peers are sent those two planes instead of three.
__global__ __launch_bounds__(256) void pack_low_kernel(const uint4 *__restrict__ p1, const uint4 *__restrict__ p2,
                                                       const uint4 *__restrict__ p3, size_t n16, uint4 *__restrict__ low)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const uint4 a = p1[i], b = p2[i], c = p3[i];
        low[i] = make_uint4(a.x ^ b.x ^ c.x, a.y ^ b.y ^ c.y, a.z ^ b.z ^ c.z, a.w ^ b.w ^ c.w);
    }
}

// saturating unary add of n_parts partial tables into the context's planes (16 B per lane); PACKED: the parts come
// as (low bit, high bit) planes, layout [2][part][slice], otherwise as the three unary planes, layout [3][part][slice]
template <bool PACKED>
__global__ __launch_bounds__(256) void merge_slices_kernel(const uint4 *__restrict__ parts, int n_parts,
                                                           size_t slice16, uint4 *__restrict__ d1,
                                                           uint4 *__restrict__ d2, uint4 *__restrict__ d3)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < slice16;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        uint32_t a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, a3[4] = {0, 0, 0, 0};
        for (int p = 0; p < n_parts; p++) {
            const size_t np = static_cast<size_t>(n_parts);
            uint4 v1 = parts[(0 * np + p) * slice16 + i], v2 = parts[(1 * np + p) * slice16 + i],
                  v3 = PACKED ? v2 : parts[(2 * np + p) * slice16 + i];
            if (PACKED) {                             // (low, high) -> count >= 1, >= 2, >= 3
                const uint4 lo = v1, hi = v2;
                v1 = make_uint4(lo.x | hi.x, lo.y | hi.y, lo.z | hi.z, lo.w | hi.w);
                v3 = make_uint4(lo.x & hi.x, lo.y & hi.y, lo.z & hi.z, lo.w & hi.w);
            }
            uint32_t b1[4] = {v1.x, v1.y, v1.z, v1.w}, b2[4] = {v2.x, v2.y, v2.z, v2.w},
                     b3[4] = {v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t r3 = a3[k] | b3[k] | (a2[k] & b1[k]) | (a1[k] & b2[k]);
                uint32_t r2 = a2[k] | b2[k] | (a1[k] & b1[k]);
                uint32_t r1 = a1[k] | b1[k];
                a1[k] = r1; a2[k] = r2; a3[k] = r3;
            }
        }
        d1[i] = make_uint4(a1[0], a1[1], a1[2], a1[3]);
        d2[i] = make_uint4(a2[0], a2[1], a2[2], a2[3]);
        d3[i] = make_uint4(a3[0], a3[1], a3[2], a3[3]);
    }
}

// ------------------------------------------------------------------------------------------
// sparse form of the ">= 3" plane (what ranks exchange instead of plane slices when the key space is split between GPUs).  The
// plane is sparse -- a 1M-contig sample sets 24 M of its 2^32 bits -- so a fine bucket (2^16 keys, 8 KiB of the plane) travels
// as its count and the 16-bit offsets of its set bits: 2 B per key at >= 3 instead of 8 KiB per bucket.
//   entry k of `counts` / `first` = the k-th fine bucket of the level-1 buckets in `share`, ascending (512 fine buckets each)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int share_slot(const KeyBuckets &share, uint32_t b1)         // ordinal of level-1 bucket b1 among the share's buckets
{
    int n = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t lo = 32 * w;
        if (b1 >= lo + 32) n += __popc(share.m[w]);
        else if (b1 > lo) n += __popc(share.m[w] & ((1u << (b1 - lo)) - 1u));
    }
    return n;
}

__global__ __launch_bounds__(256) void plane_sparse_count_kernel(const uint32_t *__restrict__ p3, KeyBuckets share, uint32_t *__restrict__ counts)
{
    const uint32_t b = blockIdx.x, b1 = b / kL2Rows;
    if (!share.bucket(b1)) return;
    const uint4 *g = reinterpret_cast<const uint4 *>(p3 + static_cast<size_t>(b) * kFineWords);
    uint32_t c = 0;
    for (int i = threadIdx.x; i < kFineWords / 4; i += 256) { const uint4 v = g[i]; c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
    __shared__ uint32_t part[4];
#pragma unroll
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[share_slot(share, b1) * kL2Rows + b % kL2Rows] = part[0] + part[1] + part[2] + part[3];
}

// exclusive prefix of n <= 65536 counts (one workgroup of 1024 threads); first[n] = total
__global__ __launch_bounds__(1024) void plane_sparse_prefix_kernel(const uint32_t *__restrict__ counts, int n, unsigned long long *__restrict__ first)
{
    __shared__ unsigned long long part[1024];
    const int per = (n + 1023) / 1024, a = min(n, static_cast<int>(threadIdx.x) * per), e = min(n, a + per);
    unsigned long long sum = 0;
    for (int i = a; i < e; i++) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = static_cast<int>(threadIdx.x) >= d ? part[threadIdx.x - d] : 0ull;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
    for (int i = a; i < e; i++) { first[i] = run; run += counts[i]; }
    if (threadIdx.x == 1023) first[n] = part[1023];
}

// the set bits of every fine bucket of the share as ascending 16-bit offsets at first[slot] (keys beyond `cap` are not written:
// the caller sees first[n] > cap)
__global__ __launch_bounds__(256) void plane_sparse_pack_kernel(const uint32_t *__restrict__ p3, KeyBuckets share,
                                                                const unsigned long long *__restrict__ first, uint16_t *__restrict__ keys,
                                                                unsigned long long cap)
{
    const uint32_t b = blockIdx.x, b1 = b / kL2Rows;
    if (!share.bucket(b1)) return;
    const unsigned long long at0 = first[share_slot(share, b1) * kL2Rows + b % kL2Rows];
    const uint32_t *g = p3 + static_cast<size_t>(b) * kFineWords;
    constexpr int kPer = kFineWords / 256;                           // 8 consecutive words per thread: ascending keys overall
    uint32_t w[kPer], c = 0;
    const uint4 *g4 = reinterpret_cast<const uint4 *>(g + threadIdx.x * kPer);
#pragma unroll
    for (int k = 0; k < kPer / 4; k++) { const uint4 v = g4[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
#pragma unroll
    for (int k = 0; k < kPer; k++) c += __popc(w[k]);
    // exclusive prefix of c over the workgroup
    __shared__ uint32_t wave_sum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (int k = 0; k < wv; k++) before += wave_sum[k];
    unsigned long long at = at0 + before;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        uint32_t x = w[k];
        while (x) {
            const int bit = __ffs(static_cast<int>(x)) - 1;
            x &= x - 1;
            if (at < cap) keys[at] = static_cast<uint16_t>((threadIdx.x * kPer + k) * 32 + bit);
            at++;
        }
    }
}

// the reverse: every fine bucket of the share rebuilt from its keys in LDS and written to the plane (all 8 KiB of it)
__global__ __launch_bounds__(256) void plane_sparse_unpack_kernel(uint32_t *__restrict__ p3, KeyBuckets share,
                                                                  const unsigned long long *__restrict__ first, const uint16_t *__restrict__ keys,
                                                                  unsigned long long cap)
{
    __shared__ uint32_t l3[kFineWords];
    const uint32_t b = blockIdx.x, b1 = b / kL2Rows;
    if (!share.bucket(b1)) return;
    const int slot = share_slot(share, b1) * kL2Rows + b % kL2Rows;
    const unsigned long long a = min(first[slot], cap), e = min(first[slot + 1], cap);      // (a sender whose keys did not fit its room: what is there)
    for (int i = threadIdx.x; i < kFineWords; i += 256) l3[i] = 0;
    __syncthreads();
    for (unsigned long long i = a + threadIdx.x; i < e; i += 256) { const uint32_t k = keys[i]; atomicOr(&l3[k >> 5], 1u << (k & 31)); }
    __syncthreads();
    uint4 *o = reinterpret_cast<uint4 *>(p3 + static_cast<size_t>(b) * kFineWords);
    for (int i = threadIdx.x; i < kFineWords / 4; i += 256) o[i] = reinterpret_cast<const uint4 *>(l3)[i];
}

}  // namespace palace

using namespace palace;

extern "C" {

int palace_eref_set_coder(palace_ctx *ctx, const uint8_t header400[400])
{
    PALACE_REQUIRE(ctx && header400, "null argument");
    CoderMasks m;
    PALACE_REQUIRE(masks_from_header(header400, &m) == 0,
                   "index header does not hold a permutation of (0,1,2) at every k-mer position");
    ctx->masks = m;
    ctx->coder_set = true;
    return PALACE_OK;
}

int palace_eref_table_reset(palace_ctx *ctx)
{
    PALACE_REQUIRE(ctx, "ctx is null");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    bool fresh = ctx->plane[0] == nullptr;
    int rc = ensure_table(ctx);
    if (rc) return rc;
    if (!fresh && !ctx->planeless)                             // (a count that probed every entry set itself left all three planes zero)
        for (int p = ctx->final_only ? 2 : 0; p < 3; p++)       // (after a final count the two lower planes are zero already)
            PALACE_HIP_TRY(hipMemsetAsync(ctx->plane[p], 0, kPlaneBytes, ctx->stream));
    ctx->planeless = false;
    ctx->table_clean = true;
    ctx->keys_counted = 0;
    ctx->final_only = false;
    ctx->c0_hits_ix = nullptr;
    ctx->counts_ptr = nullptr;
    return PALACE_OK;
}

int palace_eref_table_planes(palace_ctx *ctx, void **d_planes3, size_t *bytes_per_plane)
{
    PALACE_REQUIRE(ctx && d_planes3 && bytes_per_plane, "null argument");
    PALACE_REQUIRE(!ctx->planeless, "the table holds nothing (option probe_all_sets: its last count tested the attached index and wrote no plane): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    for (int p = 0; p < 3; p++) d_planes3[p] = ctx->plane[p];
    *bytes_per_plane = kPlaneBytes;
    return PALACE_OK;
}

int palace_eref_table_attach(palace_ctx *ctx, void *const d_planes3[3])
{
    PALACE_REQUIRE(ctx && d_planes3 && d_planes3[0] && d_planes3[1] && d_planes3[2], "null argument");
    for (int p = 0; p < 3; p++)
        PALACE_REQUIRE(reinterpret_cast<uintptr_t>(d_planes3[p]) % 16 == 0, "planes must be 16-byte aligned");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < 3; p++) {
        if (ctx->plane[p] && !ctx->planes_external) PALACE_HIP_TRY(hipFree(ctx->plane[p]));
        ctx->plane[p] = static_cast<uint32_t *>(d_planes3[p]);
    }
    ctx->planes_external = true;
    ctx->planeless = false;
    ctx->table_clean = false;                              // caller-owned memory: contents unknown
    ctx->keys_counted = -1;
    ctx->final_only = false;
    ctx->c0_hits_ix = nullptr;
    ctx->counts_ptr = nullptr;
    return PALACE_OK;
}

int palace_eref_table_invalidate(palace_ctx *ctx)
{
    PALACE_REQUIRE(ctx, "ctx is null");
    ctx->table_clean = false;
    ctx->planeless = false;
    ctx->keys_counted = -1;
    ctx->final_only = false;
    ctx->c0_hits_ix = nullptr;
    ctx->counts_ptr = nullptr;
    return PALACE_OK;
}

static int merge_slices_impl(palace_ctx *ctx, const void *d_parts, int n_parts, size_t slice_off, size_t slice_bytes, bool packed)
{
    PALACE_REQUIRE(ctx && d_parts && n_parts > 0, "bad argument");
    PALACE_REQUIRE(slice_off % 16 == 0 && slice_bytes % 16 == 0 && slice_off + slice_bytes <= kPlaneBytes,
                   "slice must be 16-byte aligned and inside the plane");
    PALACE_REQUIRE(!ctx->final_only, "the table holds only its \">= 3\" plane (option final_count): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    size_t n16 = slice_bytes / 16;
    if (n16 == 0) return PALACE_OK;
    ctx->table_clean = false;
    ctx->keys_counted = -1;                                // (partial tables of other ranks folded in: how many keys stand behind the planes is not known here)
    char *b1 = reinterpret_cast<char *>(ctx->plane[0]) + slice_off;
    char *b2 = reinterpret_cast<char *>(ctx->plane[1]) + slice_off;
    char *b3 = reinterpret_cast<char *>(ctx->plane[2]) + slice_off;
    unsigned blocks = static_cast<unsigned>(std::min<size_t>((n16 + 255) / 256, kCUs * 8));
    if (packed)
        hipLaunchKernelGGL(merge_slices_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint4 *>(d_parts), n_parts, n16, reinterpret_cast<uint4 *>(b1),
                           reinterpret_cast<uint4 *>(b2), reinterpret_cast<uint4 *>(b3));
    else
        hipLaunchKernelGGL(merge_slices_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint4 *>(d_parts), n_parts, n16, reinterpret_cast<uint4 *>(b1),
                           reinterpret_cast<uint4 *>(b2), reinterpret_cast<uint4 *>(b3));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_eref_table_merge_slices(palace_ctx *ctx, const void *d_parts, int n_parts, size_t slice_off,
                                   size_t slice_bytes)
{
    return merge_slices_impl(ctx, d_parts, n_parts, slice_off, slice_bytes, false);
}

int palace_eref_table_merge_slices_packed(palace_ctx *ctx, const void *d_parts, int n_parts, size_t slice_off,
                                          size_t slice_bytes)
{
    return merge_slices_impl(ctx, d_parts, n_parts, slice_off, slice_bytes, true);
}

int palace_eref_table_pack_low(palace_ctx *ctx, void *d_low)
{
    PALACE_REQUIRE(ctx && d_low, "null argument");
    PALACE_REQUIRE(reinterpret_cast<uintptr_t>(d_low) % 16 == 0, "buffer must be 16-byte aligned");
    PALACE_REQUIRE(!ctx->final_only, "the table holds only its \">= 3\" plane (option final_count): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    const size_t n16 = kPlaneBytes / 16;
    hipLaunchKernelGGL(pack_low_kernel, dim3(kCUs * 8), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint4 *>(ctx->plane[0]), reinterpret_cast<const uint4 *>(ctx->plane[1]),
                       reinterpret_cast<const uint4 *>(ctx->plane[2]), n16, static_cast<uint4 *>(d_low));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

static int sparse_share(const uint32_t mask128[4], KeyBuckets *kb, int *n_fine)
{
    int n1 = 0;
    for (int k = 0; k < 4; k++) { kb->m[k] = mask128[k]; n1 += __builtin_popcount(mask128[k]); }
    *n_fine = n1 * kL2Rows;
    return n1;
}

int palace_eref_plane_pack(palace_ctx *ctx, const uint32_t mask128[4], uint32_t *d_counts, uint16_t *d_keys, int64_t cap_keys,
                           unsigned long long *d_first)
{
    PALACE_REQUIRE(ctx && mask128 && d_counts && d_keys && d_first && cap_keys >= 0, "bad argument");
    PALACE_REQUIRE(!ctx->planeless, "the table holds nothing (option probe_all_sets: its last count tested the attached index and wrote no plane): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    KeyBuckets kb;
    int n_fine = 0;
    PALACE_REQUIRE(sparse_share(mask128, &kb, &n_fine) > 0, "empty share");
    hipLaunchKernelGGL(plane_sparse_count_kernel, dim3(kFine), dim3(256), 0, ctx->stream, ctx->plane[2], kb, d_counts);
    hipLaunchKernelGGL(plane_sparse_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_counts, n_fine, d_first);
    hipLaunchKernelGGL(plane_sparse_pack_kernel, dim3(kFine), dim3(256), 0, ctx->stream, ctx->plane[2], kb, d_first, d_keys,
                       static_cast<unsigned long long>(cap_keys));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_eref_plane_unpack(palace_ctx *ctx, const uint32_t mask128[4], const uint32_t *d_counts, const uint16_t *d_keys, int64_t cap_keys,
                             unsigned long long *d_first)
{
    PALACE_REQUIRE(ctx && mask128 && d_counts && d_keys && d_first && cap_keys >= 0, "bad argument");
    PALACE_REQUIRE(!ctx->planeless, "the table holds nothing (option probe_all_sets: its last count tested the attached index and wrote no plane): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    KeyBuckets kb;
    int n_fine = 0;
    PALACE_REQUIRE(sparse_share(mask128, &kb, &n_fine) > 0, "empty share");
    hipLaunchKernelGGL(plane_sparse_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_counts, n_fine, d_first);
    hipLaunchKernelGGL(plane_sparse_unpack_kernel, dim3(kFine), dim3(256), 0, ctx->stream, ctx->plane[2], kb, d_first, d_keys,
                       static_cast<unsigned long long>(cap_keys));
    PALACE_HIP_TRY(hipGetLastError());
    ctx->table_clean = false;
    ctx->c0_hits_ix = nullptr;
    return PALACE_OK;
}

int palace_eref_table_lookup(palace_ctx *ctx, const uint32_t *d_keys, int64_t n, uint8_t *d_counts)
{
    PALACE_REQUIRE(ctx && n >= 0, "bad argument");
    if (n == 0) return PALACE_OK;
    PALACE_REQUIRE(d_keys && d_counts, "null device pointer");
    PALACE_REQUIRE(!ctx->final_only, "the table holds only its \">= 3\" plane (option final_count): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(table_lookup_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_keys, n, ctx->plane[0], ctx->plane[1], ctx->plane[2], d_counts);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_eref_table_popcounts(palace_ctx *ctx, uint64_t out3[3])
{
    PALACE_REQUIRE(ctx && out3, "null argument");
    PALACE_REQUIRE(!ctx->planeless, "the table holds nothing (option probe_all_sets: its last count tested the attached index and wrote no plane): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    PALACE_HIP_TRY(hipMemsetAsync(ctx->d_small, 0, 3 * sizeof(uint64_t), ctx->stream));
    for (int p = 0; p < 3; p++) {
        hipLaunchKernelGGL(plane_popcount_kernel, dim3(kCUs * 8), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const uint4 *>(ctx->plane[p]), kPlaneBytes / 16,
                           reinterpret_cast<unsigned long long *>(ctx->d_small) + p);
        PALACE_HIP_TRY(hipGetLastError());
    }
    PALACE_HIP_TRY(hipMemcpyAsync(out3, ctx->d_small, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PALACE_OK;
}

}  // extern "C"
