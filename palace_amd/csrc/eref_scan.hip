// eref on gfx950, Phase B: look-ups and the 500-position windows over every ref (rows E2, E5, E6 of SURVEY.md section 8)
#include "eref_common.hpp"

namespace palace {

// prefix sums of ceil(len/kTilePos) and ceil(len/64) over the sequences (single block).
__global__ __launch_bounds__(1024) void seq_prefix_kernel(const int64_t *__restrict__ offsets,
                                                          int64_t n, int64_t *__restrict__ tile_pre,
                                                          int64_t *__restrict__ word_pre)
{
    __shared__ int64_t s_t[1024], s_w[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t a = min(n, t * per), b = min(n, a + per);
    int64_t st = 0, sw = 0;
    for (int64_t r = a; r < b; r++) {
        int64_t len = offsets[r + 1] - offsets[r];
        st += (len + kTilePos - 1) / kTilePos;
        sw += (len + 63) / 64;
    }
    s_t[t] = st;
    s_w[t] = sw;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t vt = (t >= d) ? s_t[t - d] : 0, vw = (t >= d) ? s_w[t - d] : 0;
        __syncthreads();
        s_t[t] += vt;
        s_w[t] += vw;
        __syncthreads();
    }
    int64_t rt = s_t[t] - st, rw = s_w[t] - sw;      // exclusive
    for (int64_t r = a; r < b; r++) {
        int64_t len = offsets[r + 1] - offsets[r];
        tile_pre[r] = rt;
        word_pre[r] = rw;
        rt += (len + kTilePos - 1) / kTilePos;
        rw += (len + 63) / 64;
    }
    if (t == 1023) {
        tile_pre[n] = s_t[1023];
        word_pre[n] = s_w[1023];
    }
}

// ------------------------------------------------------------------------------------------
// E5: per-position hit bits of every ref (lookup in plane 3)
// MODE 0: write any/all hit words;  MODE 1: write the three indices (E2 index build)
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void eref_ref_kernel(const uint8_t *__restrict__ bases,
                                                       const int64_t *__restrict__ offsets,
                                                       int64_t n_refs,
                                                       const int64_t *__restrict__ tile_pre,
                                                       const int64_t *__restrict__ word_pre,
                                                       CoderMasks masks,
                                                       const uint32_t *__restrict__ p3,
                                                       uint64_t *__restrict__ any_words,
                                                       uint64_t *__restrict__ all_words,
                                                       uint32_t *__restrict__ idx_out,
                                                       const int64_t *__restrict__ idx_offsets,
                                                       const uint8_t *__restrict__ need,
                                                       const uint8_t *__restrict__ active)
{
    const int64_t tile = blockIdx.x;
    if (tile >= tile_pre[n_refs]) return;
    const int64_t r = find_seq(tile_pre, n_refs, tile);
    if (MODE == 0 && active && !active[r]) return;       // inactive ref (eref_need_kernel): nobody reads its words
    const int64_t beg = offsets[r], len = offsets[r + 1] - beg;
    const int64_t npos = len - 31;                       // may be <= 0
    const int64_t n_chunks = (len + 63) / 64;
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    constexpr int per_wave = kTileChunks / 4;
    const int64_t c0 = (tile - tile_pre[r]) * kTileChunks + static_cast<int64_t>(wv_id) * per_wave;
    if (c0 >= n_chunks) return;
    const int64_t c1 = min(n_chunks, c0 + per_wave);
    const uint8_t *s = bases + beg;
    if (MODE == 0 || MODE == 2) {
        // MODE 2: probe channel 0 only (writes the channel-0 hit bits into any_words).
        // MODE 0: full three-channel probe; with `need`, chunks whose flag is clear keep their channel-0
        //         bits as `any` and get all = 0 (they cannot lie in a window that passes, see scan_refs).
        // All byte loads of the wave's 8 chunks, then all probes, are issued before the first use.
        constexpr int NCH = MODE == 2 ? 1 : 3;
        const int64_t wbase = word_pre[r];
        bool todo[per_wave];
        bool any_todo = false;
#pragma unroll
        for (int q = 0; q < per_wave; q++) {
            todo[q] = (c0 + q < c1) && (MODE == 2 || !need || need[wbase + c0 + q]);
            any_todo |= todo[q];
        }
        if (MODE == 0 && need) {
#pragma unroll
            for (int q = 0; q < per_wave; q++)
                if (c0 + q < c1 && !todo[q] && lane == 0) all_words[wbase + c0 + q] = 0;
            if (!any_todo) return;                                         // wave-uniform
        }
        uint32_t ch[per_wave + 1];
#pragma unroll
        for (int q = 0; q <= per_wave; q++) {
            const int64_t idx = (c0 + q) * 64 + lane;
            ch[q] = (idx < len) ? s[idx] : 0u;
        }
        uint32_t word[per_wave][NCH], sh[per_wave];
        BaseBits b0 = classify(ch[0]);
        Streams lo{__ballot(b0.p0), __ballot(b0.p1), __ballot(b0.p2), __ballot(b0.ok)};
#pragma unroll
        for (int q = 0; q < per_wave; q++) {
            BaseBits bn = classify(ch[q + 1]);
            Streams hi{__ballot(bn.p0), __ballot(bn.p1), __ballot(bn.p2), __ballot(bn.ok)};
            const int64_t j = (c0 + q) * 64 + lane;
            const uint32_t ok = window32(lo.ok, hi.ok, lane);
            const bool valid = todo[q] && (j < npos) && ok == 0xffffffffu;
            uint32_t key[3] = {0, 0, 0};
            if (valid) {
                const uint32_t w0 = window32(lo.p0, hi.p0, lane), w1 = window32(lo.p1, hi.p1, lane),
                               w2 = window32(lo.p2, hi.p2, lane);
                if (MODE == 2) key[0] = canonical(masks, 0, w0, w1, w2, __brev(w0), __brev(w1), __brev(w2));
                else kmer_keys(masks, w0, w1, w2, key);
            }
            sh[q] = (key[0] & 31) | ((key[1] & 31) << 8) | ((key[2] & 31) << 16);
#pragma unroll
            for (int i = 0; i < NCH; i++)         // index 0 means "none" (extract_ref.cpp:861)
                word[q][i] = (valid && key[i] != 0) ? p3[key[i] >> 5] : 0u;
            lo = hi;
        }
#pragma unroll
        for (int q = 0; q < per_wave; q++) {
            if (!todo[q]) continue;                                    // wave-uniform
            int h = (word[q][0] >> (sh[q] & 31)) & 1u;
            if (MODE == 0)
                h += ((word[q][NCH > 1 ? 1 : 0] >> ((sh[q] >> 8) & 31)) & 1u) +
                     ((word[q][NCH > 2 ? 2 : 0] >> ((sh[q] >> 16) & 31)) & 1u);
            const uint64_t any = __ballot(h > 0), all = __ballot(h == 3);
            if (lane == 0) {
                any_words[wbase + c0 + q] = any;
                if (MODE == 0) all_words[wbase + c0 + q] = all;
            }
        }
        return;
    }
    Streams lo = ballot_streams(s, c0 * 64 + lane, len);
    for (int64_t c = c0; c < c1; c++) {
        Streams hi = ballot_streams(s, (c + 1) * 64 + lane, len);
        const int64_t j = c * 64 + lane;
        uint32_t ok = window32(lo.ok, hi.ok, lane);
        bool valid = (j < npos) && ok == 0xffffffffu;
        uint32_t key[3] = {0, 0, 0};
        if (valid)
            kmer_keys(masks, window32(lo.p0, hi.p0, lane), window32(lo.p1, hi.p1, lane),
                      window32(lo.p2, hi.p2, lane), key);
        if (j < npos) {
            uint32_t *o = idx_out + idx_offsets[r] + 3 * j;
            o[0] = key[0]; o[1] = key[1]; o[2] = key[2];
        }
        lo = hi;
    }
}

// ------------------------------------------------------------------------------------------
// E6: window scan + interval merge, one block per ref (slide_window, extract_ref.cpp:504-617)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t prefix_count(const uint64_t *__restrict__ words,
                                                 const uint32_t *__restrict__ pre, int64_t j)
{
    int64_t w = j >> 6;
    int b = static_cast<int>(j & 63);
    uint64_t mask = (b == 63) ? ~0ull : ((2ull << b) - 1);
    return pre[w] + __popcll(words[w] & mask);           // hits at positions <= j
}


// one workgroup per group of four fine buckets: its 32 KiB slice of plane 3 in LDS, the group's entries of every entry set in
// `sets.mask` (16-byte vectors of eight 16-bit keys; a vector lies in ONE fine bucket, buckets start on multiples of 8) tested
// against it, a byte of hit bits per vector.  The plane is read ONCE for all sets.  The first set's first batch of loads is
// issued before the slice is waited for; the stores of a batch follow its tests.
struct ProbeSet { const unsigned long long *first; const uint16_t *keys16; uint8_t *ehits; };
struct ProbeSets { ProbeSet s[kSets]; uint32_t mask; };
constexpr int kProbeThreads = 512;
__global__ __launch_bounds__(kProbeThreads) void eref_probe_sets_kernel(ProbeSets sets, const uint32_t *__restrict__ p3)
{
    __shared__ uint32_t l3[kSliceWords];
    const uint32_t g = blockIdx.x;
    const uint4 *g3 = reinterpret_cast<const uint4 *>(p3 + static_cast<size_t>(g) * kSliceWords);
    for (int i = threadIdx.x; i < kSliceWords / 4; i += kProbeThreads) reinterpret_cast<uint4 *>(l3)[i] = g3[i];
    __syncthreads();
    for (int set = 0; set < kSets; set++) {                            // uniform
        if (!((sets.mask >> set) & 1u)) continue;
        const ProbeSet &ps = sets.s[set];
        const unsigned long long f0 = ps.first[g * kGroupsPerProbe] / 8;
        unsigned long long fk[kGroupsPerProbe];                        // start vector of each fine bucket behind the first, end of the group
#pragma unroll
        for (int k = 0; k < kGroupsPerProbe; k++) fk[k] = ps.first[g * kGroupsPerProbe + k + 1] / 8;
        const unsigned long long hi = fk[kGroupsPerProbe - 1];
        constexpr int kBatch = 3;                                      // a group's ~12 000 entries of a channel = ~1 500 vectors: one batch of 512 x 3
        const uint4 *pv = reinterpret_cast<const uint4 *>(ps.keys16);
        for (unsigned long long b0 = f0; b0 < hi; b0 += static_cast<unsigned long long>(kBatch) * kProbeThreads) {     // uniform
            uint4 cur[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const unsigned long long i = b0 + threadIdx.x + static_cast<unsigned long long>(u) * kProbeThreads;
                cur[u] = i < hi ? pv[i] : uint4{0, 0, 0, 0};
            }
            uint32_t m[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const unsigned long long i = b0 + threadIdx.x + static_cast<unsigned long long>(u) * kProbeThreads;
                uint32_t sub = 0;                                      // the fine bucket's 2^16-bit part of the slice
#pragma unroll
                for (int k = 0; k + 1 < kGroupsPerProbe; k++) sub += i >= fk[k] ? 1u : 0u;
                m[u] = i < hi ? probe_vector(l3 + sub * kFineWords, cur[u]) : 0u;
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const unsigned long long i = b0 + threadIdx.x + static_cast<unsigned long long>(u) * kProbeThreads;
                if (i < hi) ps.ehits[i] = static_cast<uint8_t>(m[u]);
            }
        }
    }
}

// the sentinels' hit bytes (one per sentinel = per 4 positions, in position order) -> a bit word per 64 positions with the bits
// of the sentinel positions (0, 4, ..., 60) set: what eref_need_kernel reads as "channel-0 hits" with the sentinel threshold
__global__ __launch_bounds__(256) void eref_sentinel_words_kernel(const uint4 *__restrict__ sent_bytes, int64_t n_words, uint64_t *__restrict__ words)
{
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint4 v = sent_bytes[w];                                 // 16 sentinels = 64 positions
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        uint64_t out = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) out |= static_cast<uint64_t>((d[k >> 2] >> (8 * (k & 3))) & 1u) << (kSentinelStride * k);
        words[w] = out;
    }
}

// the three channels' hit bits of the needed chunks of the active refs, gathered from the entry-order bit arrays through the
// entry maps: any / all words as eref_ref_kernel<0> writes them; chunks that are not needed get zeros (no window that can pass
// touches them: eref_need_kernel), the words of inactive refs are nobody's to read.  Tiling as eref_ref_kernel.
struct GatherArgs { const uint32_t *eix[3]; const uint8_t *ehits[3]; };
// MODE 0: all three channels (the pruning on the sentinels is the only one);
// MODE 1: channel 0 alone into any_words -- on which eref_need_kernel prunes a SECOND time, with the exact threshold;
// MODE 2: channels 1 and 2, joined with the channel-0 bits MODE 1 left in any_words (chunks not needed any more: zeros).
template <int MODE>
__global__ __launch_bounds__(256) void eref_gather_hits_kernel(const int64_t *__restrict__ offsets, int64_t n_refs,
                                                               const int64_t *__restrict__ tile_pre, const int64_t *__restrict__ word_pre,
                                                               GatherArgs ga, const uint8_t *__restrict__ need, const uint8_t *__restrict__ active,
                                                               uint64_t *__restrict__ any_words, uint64_t *__restrict__ all_words)
{
    const int64_t tile = blockIdx.x;
    if (tile >= tile_pre[n_refs]) return;
    const int64_t r = find_seq(tile_pre, n_refs, tile);
    if (!active[r]) return;
    const int64_t len = offsets[r + 1] - offsets[r];
    const int64_t npos = len - 31;
    const int64_t n_chunks = (len + 63) / 64;
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    constexpr int per_wave = kTileChunks / 4;
    const int64_t c0 = (tile - tile_pre[r]) * kTileChunks + static_cast<int64_t>(wv_id) * per_wave;
    if (c0 >= n_chunks) return;
    const int64_t c1 = min(n_chunks, c0 + per_wave), wbase = word_pre[r];
    constexpr int C0 = MODE == 2 ? 1 : 0, C1 = MODE == 1 ? 1 : 3;         // channels [C0, C1) are gathered
    uint32_t e[per_wave][3];
    bool todo[per_wave];
#pragma unroll
    for (int q = 0; q < per_wave; q++) {                                  // every entry look-up of the wave's chunks, then every bit look-up
        todo[q] = c0 + q < c1 && need[wbase + c0 + q];
        const int64_t j = (c0 + q) * 64 + lane;
        const int64_t posid = (wbase + c0 + q) * 64 + lane;
#pragma unroll
        for (int c = C0; c < C1; c++) e[q][c] = (todo[q] && j < npos) ? ga.eix[c][posid] : ~0u;
    }
    uint32_t byte_of[per_wave][3];
#pragma unroll
    for (int q = 0; q < per_wave; q++)
#pragma unroll
        for (int c = C0; c < C1; c++) byte_of[q][c] = e[q][c] != ~0u ? ga.ehits[c][e[q][c] >> 3] : 0u;
#pragma unroll
    for (int q = 0; q < per_wave; q++) {
        if (c0 + q >= c1) continue;                                       // uniform
        int h = 0;
#pragma unroll
        for (int c = C0; c < C1; c++) h += (byte_of[q][c] >> (e[q][c] & 7u)) & 1u;
        if (MODE == 2) h += todo[q] ? static_cast<int>((any_words[wbase + c0 + q] >> lane) & 1ull) : 0;
        const uint64_t any = __ballot(h > 0), all = __ballot(h == 3);
        if (lane == 0) {
            any_words[wbase + c0 + q] = todo[q] ? any : 0ull;
            if (MODE != 1) all_words[wbase + c0 + q] = todo[q] ? all : 0ull;
        }
    }
}

// entry-order hit bits of the SENTINEL set -> a byte per sentinel in position order: the entries that hit (a few per cent) are listed per workgroup in LDS,
// then every thread takes hits of the list -- the look-ups of `pos` and the byte stores of a thread are independent of each
// other and issued together.  n16: 16-byte vectors of `ehits` (128 entries each).
// A hit is a BYTE store into a byte array that eref_sentinel_words_kernel packs afterwards: as atomicOr into the bit words
// themselves (no memset of the bytes, no packing pass) the ~9 M random hits of a step -- when ALL of channel 0 was scattered, before
// the sentinels -- cost 0.45 ms MORE (scan 1.81 against 1.36 ms, round 5, tools/ab.sh r05g; round 3 had found the same).
constexpr int kScatterThreads = 256, kScatterList = 4096;
// word_pre / r_lo / r_hi (options scan_ref_lo / _hi: a rank of N scans its range of the refs): only the sentinels of those refs are
// carried over -- the byte stores, the expensive half of a hit (a random 64-byte granule each), shrink with the rank's share of the DB.
__global__ __launch_bounds__(kScatterThreads) void eref_ehits_scatter_kernel(const uint4 *__restrict__ ehits, unsigned long long n16,
                                                                             const uint32_t *__restrict__ pos, uint8_t *__restrict__ hit_bytes,
                                                                             const int64_t *__restrict__ word_pre, int64_t r_lo, int64_t r_hi)
{
    // sentinel ordinals of the refs [r_lo, r_hi): 64 / kSentinelStride per word of the hit bitmap
    const uint32_t s_lo = static_cast<uint32_t>(word_pre[r_lo] * (64 / kSentinelStride)), s_n = static_cast<uint32_t>(word_pre[r_hi] * (64 / kSentinelStride)) - s_lo;
    auto hit = [&](uint32_t p) { if (p - s_lo < s_n) hit_bytes[p] = 1; };
    __shared__ uint32_t list[kScatterList];
    __shared__ uint32_t n_list;
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * kScatterThreads;
    for (unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kScatterThreads; base < n16; base += stride) {   // uniform
        if (threadIdx.x == 0) n_list = 0;
        __syncthreads();
        const unsigned long long i = base + threadIdx.x;
        const uint4 v = i < n16 ? ehits[i] : uint4{0, 0, 0, 0};
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const uint32_t mine = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
        uint32_t at = mine ? atomicAdd(&n_list, mine) : 0u;
        const unsigned long long e0 = i * 128;                         // (entries of a launch fit 32 bits relative to the vector's start: e0 + 127)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = w[k];
            while (x) {
                const int bit = __ffs(static_cast<int>(x)) - 1;
                x &= x - 1;
                const uint32_t rel = static_cast<uint32_t>(threadIdx.x) * 128u + 32u * k + bit;          // entry relative to `base * 128`
                if (at < kScatterList) list[at] = rel;
                else { const uint32_t p = pos[e0 + 32 * k + bit]; if (p != ~0u) hit(p); }      // list full: directly
                at++;
            }
        }
        __syncthreads();
        const uint32_t n = min(n_list, static_cast<uint32_t>(kScatterList));
        const unsigned long long eb = base * 128;
        constexpr int kUn = 4;
        for (uint32_t j0 = threadIdx.x; j0 < n; j0 += kUn * kScatterThreads) {
            uint32_t p[kUn];
#pragma unroll
            for (int u = 0; u < kUn; u++) {
                const uint32_t j = j0 + u * kScatterThreads;
                p[u] = j < n ? pos[eb + list[j]] : ~0u;
            }
#pragma unroll
            for (int u = 0; u < kUn; u++)
                if (p[u] != ~0u) hit(p[u]);
        }
        __syncthreads();
    }
}

// Phase B probe pruning (exact).  A window can only pass if it holds >= three_min positions where ALL
// three channels hit (extract_ref.cpp:561), hence >= three_min channel-0 hits.  So channel 0 is probed
// everywhere first; this kernel marks the 64-position chunks that overlap at least one window with
// enough channel-0 hits, and only those chunks get the other two probes.  Everywhere else `all` is 0
// and `any` keeps the channel-0 bits: every window touching such a chunk fails the three_min test
// with the true bits already, so the substitution cannot change any good[j].
constexpr int kRefThreads = 1024;      // per-ref kernels: one workgroup walks a whole ref, so its latency is the kernel's
__global__ __launch_bounds__(kRefThreads) void eref_need_kernel(const int64_t *__restrict__ offsets, int64_t n_refs,
                                                        const int64_t *__restrict__ word_pre,
                                                        const uint64_t *__restrict__ c0_words,
                                                        uint32_t *__restrict__ c0_pre, uint64_t *__restrict__ cand_words,
                                                        uint32_t *__restrict__ cand_pre, int three_min,
                                                        uint8_t *__restrict__ need, uint8_t *__restrict__ active,
                                                        int64_t r_lo, int64_t r_hi)       // refs outside [r_lo, r_hi) are not this call's: inactive
{
    const int64_t r = blockIdx.x;
    if (r >= n_refs) return;
    if (r < r_lo || r >= r_hi) { if (threadIdx.x == 0) active[r] = 0; return; }
    const int64_t len = offsets[r + 1] - offsets[r];
    const int64_t n_words = (len + 63) / 64, w0 = word_pre[r];
    const uint64_t *A = c0_words + w0;
    uint32_t *PA = c0_pre + w0, *PC = cand_pre + w0;
    uint64_t *C = cand_words + w0;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int kWaves = kRefThreads / 64;
    __shared__ uint32_t s_part[kWaves];
    __shared__ uint32_t carry;
    auto block_prefix = [&](const uint64_t *W, uint32_t *P) {     // exclusive prefix popcount per word
        if (t == 0) carry = 0;
        __syncthreads();
        for (int64_t base = 0; base < n_words; base += kRefThreads) {
            const int64_t w = base + t;
            const uint32_t c = (w < n_words) ? __popcll(W[w]) : 0;
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t u = __shfl_up(inc, d);
                if (lane >= d) inc += u;
            }
            if (lane == 63) s_part[wv] = inc;
            __syncthreads();
            uint32_t o = carry;
            for (int k = 0; k < wv; k++) o += s_part[k];
            if (w < n_words) P[w] = o + inc - c;
            __syncthreads();
            if (t == kRefThreads - 1) carry = o + inc;
            __syncthreads();
        }
    };
    // Cheap exclusion first.  A 500-position window with >= three_min channel-0 hits overlaps at most two aligned
    // 512-position groups (8 words), so one of them holds >= three_min / 2 of its hits.  Chance hits are spread
    // thin (a few per hundred positions), so for most refs of a DB no group comes close: the ref is marked
    // inactive -- no window of it can pass -- and neither the other two channels nor the window scan look at it.
    {
        bool dense = false;
        for (int64_t g = t; g * 8 < n_words; g += kRefThreads) {
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) c += (g * 8 + k < n_words) ? __popcll(A[g * 8 + k]) : 0;
            dense |= 2 * static_cast<int>(c) >= three_min;
        }
        const bool live = __syncthreads_or(dense);
        if (!live) {                                               // uniform for the workgroup
            if (t == 0) active[r] = 0;
            return;
        }
        if (t == 0) active[r] = 1;
    }
    block_prefix(A, PA);
    __threadfence_block();
    __syncthreads();
    for (int64_t w = wv; w < n_words; w += kWaves) {               // cand[j]: channel-0 hits in (j-500, j] >= three_min
        const int64_t j = w * 64 + lane;
        bool cand = false;
        if (j < len) {
            uint32_t c = prefix_count(A, PA, j);
            if (j >= 500) c -= prefix_count(A, PA, j - 500);
            cand = static_cast<int>(c) >= three_min;
        }
        const uint64_t g = __ballot(cand);
        if (lane == 0) C[w] = g;
    }
    __threadfence_block();
    __syncthreads();
    block_prefix(C, PC);
    __threadfence_block();
    __syncthreads();
    // no window with enough channel-0 hits anywhere in the ref (the usual case: chance hits are spread thin) -> inactive
    if (carry == 0) {                                              // carry = number of candidate positions; uniform
        if (t == 0) active[r] = 0;
        return;
    }
    for (int64_t w = t; w < n_words; w += kRefThreads) {           // chunk w is needed iff a cand j lies in [64w, 64w+562]
        const int64_t hi = min(len - 1, w * 64 + 63 + 499);
        uint32_t upto = prefix_count(C, PC, hi);
        uint32_t before = w ? prefix_count(C, PC, w * 64 - 1) : 0u;
        need[w0 + w] = upto > before;
    }
}

__global__ __launch_bounds__(kRefThreads) void eref_window_kernel(const int64_t *__restrict__ offsets,
                                                                  int64_t n_refs,
                                                                  const int64_t *__restrict__ word_pre,
                                                                  const uint64_t *__restrict__ any_words,
                                                                  const uint64_t *__restrict__ all_words,
                                                                  uint32_t *__restrict__ any_pre,
                                                                  uint32_t *__restrict__ all_pre,
                                                                  uint64_t *__restrict__ good_words,
                                                                  int one_min, int three_min,
                                                                  const uint8_t *__restrict__ active,
                                                                  int32_t *__restrict__ rows)
{
    const int64_t r = blockIdx.x;
    if (r >= n_refs) return;
    const int64_t len = offsets[r + 1] - offsets[r];
    if (!active[r]) {                                              // see eref_need_kernel: no window can pass
        if (threadIdx.x == 0) { rows[4 * r + 0] = 0; rows[4 * r + 1] = 0; rows[4 * r + 2] = static_cast<int>(len); rows[4 * r + 3] = 0; }
        return;
    }
    const int64_t n_words = (len + 63) / 64, w0 = word_pre[r];
    const uint64_t *A = any_words + w0, *T = all_words + w0;
    uint32_t *PA = any_pre + w0, *PT = all_pre + w0;
    uint64_t *G = good_words + w0;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int kWaves = kRefThreads / 64;

    // (a) exclusive prefix population counts per 64-position word, kRefThreads words per sweep
    __shared__ uint32_t s_a[kWaves], s_t[kWaves];
    __shared__ uint32_t carry_a, carry_t;
    if (t == 0) { carry_a = 0; carry_t = 0; }
    __syncthreads();
    for (int64_t base = 0; base < n_words; base += kRefThreads) {
        int64_t w = base + t;
        uint32_t ca = (w < n_words) ? __popcll(A[w]) : 0, ct = (w < n_words) ? __popcll(T[w]) : 0;
        uint32_t ia = ca, it = ct;                     // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t ua = __shfl_up(ia, d), ut = __shfl_up(it, d);
            if (lane >= d) { ia += ua; it += ut; }
        }
        if (lane == 63) { s_a[wv] = ia; s_t[wv] = it; }
        __syncthreads();
        uint32_t oa = carry_a, ot = carry_t;
        for (int k = 0; k < wv; k++) { oa += s_a[k]; ot += s_t[k]; }
        if (w < n_words) { PA[w] = oa + ia - ca; PT[w] = ot + it - ct; }
        __syncthreads();
        if (t == kRefThreads - 1) { carry_a = oa + ia; carry_t = ot + it; }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();

    // (b) good[j]: >= one_min any-hits and >= three_min all-hits among positions (j-500, j]
    for (int64_t w = wv; w < n_words; w += kWaves) {
        int64_t j = w * 64 + lane;
        bool good = false;
        if (j < len) {
            uint32_t one = prefix_count(A, PA, j), three = prefix_count(T, PT, j);
            if (j >= 500) { one -= prefix_count(A, PA, j - 500); three -= prefix_count(T, PT, j - 500); }
            good = static_cast<int>(one) >= one_min && static_cast<int>(three) >= three_min;
        }
        uint64_t g = __ballot(good);
        if (lane == 0) G[w] = g;
    }
    __threadfence_block();
    __syncthreads();

    // (c) rising edge -> start = max(1, j-1000); falling edge (or end of ref) -> end =
    //     min(len, j+1000); merge into the previous interval when start - prev_end < 500.
    //     The edges (a handful per ref) are collected by all threads, ordered and merged by one; a ref with
    //     more edges than the list holds is walked serially.
    constexpr int kMaxEdges = 1024;
    __shared__ uint32_t edge[kMaxEdges];               // position << 1 | rising
    __shared__ unsigned int n_edge;
    if (t == 0) n_edge = 0;
    __syncthreads();
    for (int64_t w = t; w <= n_words; w += kRefThreads) {          // one virtual zero word closes an open run
        const uint64_t g = (w < n_words) ? G[w] : 0;
        const uint64_t prev_bit = w ? (G[w - 1] >> 63) : 0;
        uint64_t x = g ^ ((g << 1) | prev_bit);
        while (x) {
            const int b = __ffsll(static_cast<long long>(x)) - 1;
            x &= x - 1;
            const unsigned int at = atomicAdd(&n_edge, 1u);
            if (at < kMaxEdges) edge[at] = (static_cast<uint32_t>(w * 64 + b) << 1) | static_cast<uint32_t>((g >> b) & 1);
        }
    }
    __syncthreads();
    if (t == 0) {
        int frag = 0, el = 0, start = 0, prev_end = 0;
        const int ilen = static_cast<int>(len);
        auto on_edge = [&](int j, bool rising) {
            if (rising) {
                start = max(1, j - 1000);
            } else {
                int end = min(ilen, j + 1000);
                if (frag > 0 && start - prev_end < 500) { el += end - prev_end; }
                else { frag++; el += end - start; }
                prev_end = end;
            }
        };
        if (n_edge <= kMaxEdges) {
            const int n = static_cast<int>(n_edge);
            for (int i = 1; i < n; i++) {                          // insertion sort: a handful of entries
                const uint32_t e = edge[i];
                int k = i - 1;
                while (k >= 0 && edge[k] > e) { edge[k + 1] = edge[k]; k--; }
                edge[k + 1] = e;
            }
            for (int i = 0; i < n; i++) on_edge(static_cast<int>(edge[i] >> 1), edge[i] & 1u);
        } else {
            uint64_t prev_bit = 0;
            for (int64_t w = 0; w <= n_words; w++) {
                uint64_t g = (w < n_words) ? G[w] : 0;
                uint64_t x = g ^ ((g << 1) | prev_bit);
                while (x) {
                    int b = __ffsll(static_cast<long long>(x)) - 1;
                    x &= x - 1;
                    on_edge(static_cast<int>(w * 64 + b), (g >> b) & 1);
                }
                prev_bit = g >> 63;
            }
        }
        rows[4 * r + 0] = frag;
        rows[4 * r + 1] = el;
        rows[4 * r + 2] = ilen;
        rows[4 * r + 3] = 0;
    }
}

}  // namespace palace

using namespace palace;

extern "C" {

static int launch_prefix(palace_ctx *ctx, const int64_t *d_offsets, int64_t n, int64_t *tile_pre, int64_t *word_pre)
{
    hipLaunchKernelGGL(seq_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_offsets, n, tile_pre, word_pre);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int palace_eref_index_refs(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                           int64_t n_refs, uint32_t *d_out, const int64_t *d_out_offsets)
{
    PALACE_REQUIRE(ctx && n_refs >= 0, "bad argument");
    if (!ctx->coder_set) { set_error("palace_eref_index_refs: coder not set"); return PALACE_ESTATE; }
    if (n_refs == 0) return PALACE_OK;
    PALACE_REQUIRE(d_bases && d_offsets && d_out && d_out_offsets, "null device pointer");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int64_t h_off[2];
    PALACE_HIP_TRY(hipMemcpyAsync(&h_off[0], d_offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipMemcpyAsync(&h_off[1], d_offsets + n_refs, 8, hipMemcpyDeviceToHost, ctx->stream));
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int64_t total = h_off[1] - h_off[0];
    PALACE_REQUIRE(total >= 0, "offsets not ascending");
    size_t pre_bytes = align_up((n_refs + 1) * 8, 256);
    int rc = ensure_workspace(ctx, 2 * pre_bytes);
    if (rc) return rc;
    char *ws = static_cast<char *>(ctx->ws.ptr);
    int64_t *tile_pre = reinterpret_cast<int64_t *>(ws), *word_pre = reinterpret_cast<int64_t *>(ws + pre_bytes);
    rc = launch_prefix(ctx, d_offsets, n_refs, tile_pre, word_pre);
    if (rc) return rc;
    int64_t max_tiles = total / kTilePos + n_refs;
    PALACE_REQUIRE(max_tiles < (1ll << 31), "too many tiles for one launch");
    hipLaunchKernelGGL(eref_ref_kernel<1>, dim3(static_cast<unsigned>(max_tiles)), dim3(256), 0, ctx->stream,
                       d_bases, d_offsets, n_refs, tile_pre, word_pre, ctx->masks,
                       static_cast<const uint32_t *>(nullptr), static_cast<uint64_t *>(nullptr),
                       static_cast<uint64_t *>(nullptr), d_out, d_out_offsets, static_cast<const uint8_t *>(nullptr),
                       static_cast<const uint8_t *>(nullptr));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

}  // extern "C"

namespace palace {
int scan_buffers(palace_ctx *ctx, const int64_t *d_offsets, int64_t n_refs, int64_t total_bases, ScanBuffers *b, bool with_hit_bytes,
                 const size_t *ehits_bytes)
{
    b->max_tiles = total_bases / kTilePos + n_refs;
    b->max_words = total_bases / 64 + n_refs + 1;
    PALACE_REQUIRE(b->max_tiles < (1ll << 31), "too many tiles for one launch");
    const size_t pre_bytes = align_up((n_refs + 1) * 8, 256);
    const size_t w64 = align_up(b->max_words * 8, 256), w32 = align_up(b->max_words * 4, 256);
    const size_t w8 = align_up(b->max_words, 256);
    const size_t hb = with_hit_bytes ? align_up(static_cast<size_t>(b->max_words) * (64 / kSentinelStride), 256) : 0;
    size_t eb[kSets] = {0, 0, 0, 0}, eb_all = 0;
    for (int k = 0; k < kSets; k++) { eb[k] = ehits_bytes && ehits_bytes[k] ? align_up(ehits_bytes[k] + 16, 256) : 0; eb_all += eb[k]; }
    int rc = ensure_workspace(ctx, 2 * pre_bytes + 3 * w64 + 2 * w32 + w8 + align_up(n_refs + 1, 256) + hb + eb_all);
    if (rc) return rc;
    char *ws = static_cast<char *>(ctx->ws.ptr);
    b->tile_pre = reinterpret_cast<int64_t *>(ws); ws += pre_bytes;
    b->word_pre = reinterpret_cast<int64_t *>(ws); ws += pre_bytes;
    b->any_w = reinterpret_cast<uint64_t *>(ws); ws += w64;
    b->all_w = reinterpret_cast<uint64_t *>(ws); ws += w64;
    b->good_w = reinterpret_cast<uint64_t *>(ws); ws += w64;
    b->any_p = reinterpret_cast<uint32_t *>(ws); ws += w32;
    b->all_p = reinterpret_cast<uint32_t *>(ws); ws += w32;
    b->need = reinterpret_cast<uint8_t *>(ws); ws += w8;
    b->active = reinterpret_cast<uint8_t *>(ws); ws += align_up(n_refs + 1, 256);
    b->hit_bytes = with_hit_bytes ? reinterpret_cast<uint8_t *>(ws) : nullptr; ws += hb;
    for (int k = 0; k < kSets; k++) { b->ehits[k] = eb[k] ? reinterpret_cast<uint8_t *>(ws) : nullptr; ws += eb[k]; }
    return launch_prefix(ctx, d_offsets, n_refs, b->tile_pre, b->word_pre);
}
}  // namespace palace

namespace {
// channel-0 hit bits are in any_w: chunks that can matter -> channels 1 and 2 only there (exact; see
// eref_need_kernel) -> windows
int scan_tail(palace_ctx *ctx, const ScanBuffers &b, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_refs,
              int one_min, int three_min, int32_t *d_rows)
{
    hipLaunchKernelGGL(eref_need_kernel, dim3(static_cast<unsigned>(n_refs)), dim3(kRefThreads), 0, ctx->stream, d_offsets,
                       n_refs, b.word_pre, b.any_w, b.any_p, b.good_w, b.all_p, three_min, b.need, b.active, static_cast<int64_t>(0), n_refs);
    PALACE_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(eref_ref_kernel<0>, dim3(static_cast<unsigned>(b.max_tiles)), dim3(256), 0, ctx->stream,
                       d_bases, d_offsets, n_refs, b.tile_pre, b.word_pre, ctx->masks, ctx->plane[2], b.any_w,
                       b.all_w, static_cast<uint32_t *>(nullptr), static_cast<const int64_t *>(nullptr),
                       static_cast<const uint8_t *>(b.need), static_cast<const uint8_t *>(b.active));
    PALACE_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(eref_window_kernel, dim3(static_cast<unsigned>(n_refs)), dim3(kRefThreads), 0, ctx->stream,
                       d_offsets, n_refs, b.word_pre, b.any_w, b.all_w, b.any_p, b.all_p, b.good_w, one_min, three_min,
                       static_cast<const uint8_t *>(b.active), d_rows);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

int scan_args_ok(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_refs, int64_t total_bases,
                 const int32_t *d_rows)
{
    PALACE_REQUIRE(ctx && n_refs >= 0 && total_bases >= 0, "bad argument");
    if (!ctx->coder_set) { set_error("scan_refs: coder not set"); return PALACE_ESTATE; }
    PALACE_REQUIRE(n_refs == 0 || (d_bases && d_offsets && d_rows), "null device pointer");
    PALACE_REQUIRE(n_refs < (1ll << 31), "too many refs for one launch");
    return PALACE_OK;
}

}  // namespace

extern "C" {

int palace_eref_scan_refs(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                          int64_t n_refs, int64_t total_bases, int one_min, int three_min,
                          int32_t *d_rows)
{
    int rc = scan_args_ok(ctx, d_bases, d_offsets, n_refs, total_bases, d_rows);
    if (rc || n_refs == 0) return rc;
    PALACE_REQUIRE(!ctx->planeless, "the table holds nothing (option probe_all_sets: its last count tested the attached index and wrote no plane): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure_table(ctx);
    if (rc) return rc;
    ScanBuffers b;
    rc = scan_buffers(ctx, d_offsets, n_refs, total_bases, &b);
    if (rc) return rc;
    // channel 0 everywhere, recomputed from the bases
    hipLaunchKernelGGL(eref_ref_kernel<2>, dim3(static_cast<unsigned>(b.max_tiles)), dim3(256), 0, ctx->stream,
                       d_bases, d_offsets, n_refs, b.tile_pre, b.word_pre, ctx->masks, ctx->plane[2], b.any_w,
                       b.all_w, static_cast<uint32_t *>(nullptr), static_cast<const int64_t *>(nullptr),
                       static_cast<const uint8_t *>(nullptr), static_cast<const uint8_t *>(nullptr));
    PALACE_HIP_TRY(hipGetLastError());
    return scan_tail(ctx, b, d_bases, d_offsets, n_refs, one_min, three_min, d_rows);
}

int palace_eref_scan_refs_indexed(palace_ctx *ctx, const palace_eref_probe_index *ix, const uint8_t *d_bases,
                                  const int64_t *d_offsets, int64_t n_refs, int64_t total_bases, int one_min,
                                  int three_min, int32_t *d_rows)
{
    PALACE_REQUIRE(ix, "null probe index");
    int rc = scan_args_ok(ctx, d_bases, d_offsets, n_refs, total_bases, d_rows);
    if (rc) return rc;
    PALACE_REQUIRE(ix->n_refs == n_refs && ix->total_bases == total_bases, "probe index was built for another ref set");
    PALACE_REQUIRE(std::memcmp(&ix->masks, &ctx->masks, sizeof(CoderMasks)) == 0, "probe index was built with another coder");
    if (n_refs == 0) return PALACE_OK;
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure_table(ctx);
    if (rc) return rc;
    // channel 0's hit bits: the count launch has left them when this index was attached to it and nothing has touched the planes
    // since; every other entry set (and channel 0 otherwise) is probed now, the plane read once for all of them
    const uint32_t fused = ctx->c0_hits_ix == ix ? ctx->hits_mask : 0u;      // bit k: set k's hit bits are the count launch's
    PALACE_REQUIRE(!ctx->planeless || fused == (1u << kSets) - 1,
                   "the table holds nothing (option probe_all_sets): only the index that rode along in the count can be scanned through; reset the table first");
    size_t eb[kSets];
    for (int k = 0; k < kSets; k++) eb[k] = ((fused >> k) & 1u) ? 0 : ix->ehits_bytes[k];
    ScanBuffers b;
    rc = scan_buffers(ctx, d_offsets, n_refs, total_bases, &b, true, eb);
    if (rc) return rc;
    PALACE_REQUIRE(static_cast<size_t>(b.max_words) * 64 == ix->hit_bytes_size, "probe index was built for another layout of the hit words");
    const bool sent_done = fused == (1u << kSets) - 1 && ctx->sent_scattered;   // the count launch carried the sentinels' hits to position order as well
    if (!sent_done) PALACE_HIP_TRY(hipMemsetAsync(b.hit_bytes, 0, static_cast<size_t>(b.max_words) * (64 / kSentinelStride), ctx->stream));
    ProbeSets sets{};
    for (int k = 0; k < kSets; k++) {
        const bool have = (fused >> k) & 1u;
        uint8_t *eh = have ? ix->ehits_own[k] : b.ehits[k];          // (this context's: several contexts may scan through one index)
        sets.s[k] = ProbeSet{ix->first + static_cast<size_t>(k) * (kIndexGroups + 1), ix->keys16[k], eh};
        if (!have) {
            sets.mask |= 1u << k;
            if (ix->ehits_bytes[k] >= 16) PALACE_HIP_TRY(hipMemsetAsync(eh + ix->ehits_bytes[k] - 16, 0, 16, ctx->stream));     // (bytes behind the last entry)
        }
    }
    if (sets.mask) hipLaunchKernelGGL(eref_probe_sets_kernel, dim3(kBuckets), dim3(kProbeThreads), 0, ctx->stream, sets, ctx->plane[2]);
    // the sentinels that hit -> position order -> the bit words eref_need_kernel reads
    const int64_t r_lo = std::min(ctx->scan_ref_lo, n_refs), r_hi = ctx->scan_ref_hi > 0 ? std::max(r_lo, std::min(ctx->scan_ref_hi, n_refs)) : n_refs;   // options scan_ref_lo / _hi
    if (!sent_done)
        hipLaunchKernelGGL(eref_ehits_scatter_kernel, dim3(kCUs * 8), dim3(kScatterThreads), 0, ctx->stream,
                           reinterpret_cast<const uint4 *>(sets.s[kSentinelSet].ehits), static_cast<unsigned long long>(ix->ehits_bytes[kSentinelSet] / 16),
                           ix->pos_s, b.hit_bytes, static_cast<const int64_t *>(b.word_pre), r_lo, r_hi);
    hipLaunchKernelGGL(eref_sentinel_words_kernel, dim3(kCUs * 8), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint4 *>(sent_done ? ix->sent_bytes_own : b.hit_bytes), b.max_words, b.any_w);
    PALACE_HIP_TRY(hipGetLastError());
    // A window that passes holds >= three_min positions hit in all channels, i.e. misses at most 500 - three_min channel-0 hits;
    // it holds at least 500 / 4 - 1 sentinels (the cumulative windows at a ref's start, which must hold three_min positions
    // to pass at all, hold more in proportion), so at least this many of its sentinels hit:
    const int sentinel_min = std::max(0, 500 / kSentinelStride - 1 - (500 - three_min));
    hipLaunchKernelGGL(eref_need_kernel, dim3(static_cast<unsigned>(n_refs)), dim3(kRefThreads), 0, ctx->stream, d_offsets,
                       n_refs, b.word_pre, b.any_w, b.any_p, b.good_w, b.all_p, sentinel_min, b.need, b.active, r_lo, r_hi);
    GatherArgs ga{};
    for (int c = 0; c < 3; c++) { ga.eix[c] = ix->eix[c]; ga.ehits[c] = sets.s[c].ehits; }
    const dim3 tiles(static_cast<unsigned>(b.max_tiles));
    const uint8_t *need = b.need, *active = b.active;
    // How sharp the sentinel pruning is depends on how full the table is: a passing window needs 39 % of its sentinels hit where the
    // exact rule needs 85 % of its positions, and in a table that many reads have filled (5M contigs: 12 G key instances for 4.3 G
    // slots, half of all keys at >= 3) chance alone gives that -- every ref would be gathered in full.  So unless the table is known
    // to be sparse (fewer key instances counted since the reset than 0.9 x 2^32: the 1M-contig sample has 2.4 G), channel 0 is
    // gathered first, the exact rule prunes once more, and channels 1 and 2 are gathered for what is left.
    const bool sparse_table = ctx->keys_counted >= 0 && ctx->keys_counted < static_cast<int64_t>(0.9 * 4294967296.0);
    if (sparse_table) {
        hipLaunchKernelGGL(eref_gather_hits_kernel<0>, tiles, dim3(256), 0, ctx->stream, d_offsets, n_refs, b.tile_pre, b.word_pre, ga, need, active,
                           b.any_w, b.all_w);
    } else {
        hipLaunchKernelGGL(eref_gather_hits_kernel<1>, tiles, dim3(256), 0, ctx->stream, d_offsets, n_refs, b.tile_pre, b.word_pre, ga, need, active,
                           b.any_w, b.all_w);
        hipLaunchKernelGGL(eref_need_kernel, dim3(static_cast<unsigned>(n_refs)), dim3(kRefThreads), 0, ctx->stream, d_offsets,
                           n_refs, b.word_pre, b.any_w, b.any_p, b.good_w, b.all_p, three_min, b.need, b.active, r_lo, r_hi);
        hipLaunchKernelGGL(eref_gather_hits_kernel<2>, tiles, dim3(256), 0, ctx->stream, d_offsets, n_refs, b.tile_pre, b.word_pre, ga, need, active,
                           b.any_w, b.all_w);
    }
    hipLaunchKernelGGL(eref_window_kernel, dim3(static_cast<unsigned>(n_refs)), dim3(kRefThreads), 0, ctx->stream,
                       d_offsets, n_refs, b.word_pre, b.any_w, b.all_w, b.any_p, b.all_p, b.good_w, one_min, three_min,
                       static_cast<const uint8_t *>(b.active), d_rows);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

}  // extern "C"
