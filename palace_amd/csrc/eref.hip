// eref on gfx950, Phase A: the reads' keys partitioned in two levels and counted in LDS (rows E3-E5 of SURVEY.md section 8); design notes in eref_common.hpp
#include "eref_common.hpp"

namespace palace {

__global__ __launch_bounds__(256) void eref_count_kernel(const uint8_t *__restrict__ bases,
                                                         const int64_t *__restrict__ offsets,
                                                         int64_t n_reads,
                                                         const uint8_t *__restrict__ keep,
                                                         CoderMasks masks, KeyBuckets range, uint32_t *__restrict__ p1,
                                                         uint32_t *__restrict__ p2,
                                                         uint32_t *__restrict__ p3)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_reads; r += n_waves) {
        if (keep && !keep[r]) continue;
        const int64_t beg = offsets[r];
        const int64_t len = offsets[r + 1] - beg;
        const int64_t npos = len - 31;
        if (npos <= 0) continue;
        const uint8_t *s = bases + beg;
        Streams lo = ballot_streams(s, lane, len);
        for (int64_t base = 0; base < npos; base += 64) {
            Streams hi = ballot_streams(s, base + 64 + lane, len);
            const int64_t j = base + lane;
            uint32_t wv = window32(lo.ok, hi.ok, lane);
            if (j < npos && wv == 0xffffffffu) {
                uint32_t key[3];
                kmer_keys(masks, window32(lo.p0, hi.p0, lane), window32(lo.p1, hi.p1, lane),
                          window32(lo.p2, hi.p2, lane), key);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (range.has(key[i])) count_key(key[i], p1, p2, p3);
            }
            lo = hi;
        }
    }
}

// Small packed read sets (palace_eref_count_reads_packed below the binning threshold): a lane per position, the windows of
// the two projection streams straight from two words each, three atomics per marked position.
__global__ __launch_bounds__(256) void eref_count_packed_kernel(const uint32_t *__restrict__ s0, const uint32_t *__restrict__ s1,
                                                                const uint32_t *__restrict__ su, int64_t n, CoderMasks masks,
                                                                KeyBuckets range, uint32_t *__restrict__ p1, uint32_t *__restrict__ p2,
                                                                uint32_t *__restrict__ p3)
{
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int64_t g = p >> 5;
        const int sh = static_cast<int>(p & 31);
        if (!((su[g] >> sh) & 1u)) continue;
        const uint32_t a0 = __builtin_amdgcn_alignbit(s0[g + 1], s0[g], sh), a1 = __builtin_amdgcn_alignbit(s1[g + 1], s1[g], sh);
        uint32_t key[3];
        kmer_keys(masks, a0, a1, ~(a0 ^ a1), key);
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (range.has(key[i])) count_key(key[i], p1, p2, p3);
    }
}

// Read ends as a bit per base position (bit p set <=> position p is the last base of a read), so
// the kernels below need no per-read offset lookups: a 32-mer starting at p is inside one read
// iff no end bit lies in [p, p+30].
__global__ void mark_read_ends_kernel(const int64_t *__restrict__ offsets, int64_t n_reads,
                                      unsigned long long *__restrict__ ends)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t a = offsets[r] - offsets[0], b = offsets[r + 1] - offsets[0];
    if (b > a) atomicOr(&ends[(b - 1) >> 6], 1ull << ((b - 1) & 63));
}

// E3 keep mask as a bit per base position: every base of a read with keep[r] == 0 is marked dropped
// (the stream kernel below then treats it as an invalid base, so none of its 32-mers is counted --
// exactly the reads the reference skips at extract_ref.cpp:955-960).
__global__ void mark_dropped_kernel(const int64_t *__restrict__ offsets, int64_t n_reads,
                                    const uint8_t *__restrict__ keep, unsigned long long *__restrict__ dropped)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n_reads || keep[r]) return;
    const int64_t a = offsets[r] - offsets[0], b = offsets[r + 1] - offsets[0];      // [a, b)
    for (int64_t w = a >> 6; w <= (b - 1) >> 6 && b > a; w++) {
        const int64_t lo = max(a, w << 6), hi = min(b, (w + 1) << 6);            // bits [lo, hi) of word w
        const unsigned long long m = ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1)) << (lo & 63);
        atomicOr(&dropped[w], m);
    }
}

// The read set packed to two bits per base plus a start mask, as three bit streams (u32 word w = positions 32w .. 32w+31):
// the projections P0 = {A,T} and P1 = {A,C} of every base -- together the base itself; the third projection the coder
// uses, P2 = {A,G}, is ~(P0 ^ P1) -- and U, "a 32-mer may start here": its 32 bases are valid (and not dropped), it does
// not run over a read end or over the end of the set.  One pass over the bases (16 B per lane), 0.375 B/base written;
// everything downstream -- the partition kernel -- then gets a 32-mer's projection windows with one v_alignbit each
// instead of a byte load, a classification, four ballots and ten 64-bit shifts per position.
//
// Per dword of four ASCII bases the class bits are computed on all four bytes at once (bit k of a byte is brought to
// bit 0 of that byte by x >> k; what the shift drags in from the next byte lands in bits 1..7 and is masked off):
//   A 0x41, C 0x43, G 0x47, T 0x54 (bit 5 = case, ignored):  c = x >> 1:  {A,T} = !(c & 1), {A,C} = !(c & 2),
//   {A,G} = !((c ^ c >> 1) & 1) as in classify();  valid = b6 & !b7 & !b3 & (b4 ? b2 & !b1 & !b0 : b0 & (b1 | !b2)).
__device__ __forceinline__ uint32_t gather4(uint32_t y)        // bit 0 of bytes 0..3 -> bits 0..3
{
    y &= 0x01010101u;
    y |= y >> 7;
    return (y | (y >> 14)) & 15u;
}

__device__ __forceinline__ void class_bits4(uint32_t x, uint32_t &p0, uint32_t &p1, uint32_t &ok)
{
    const uint32_t b0 = x, b1 = x >> 1, b2 = x >> 2, b3 = x >> 3, b4 = x >> 4, b6 = x >> 6, b7 = x >> 7;
    p0 = gather4(~b1);
    p1 = gather4(~b2);
    const uint32_t t_like = b2 & ~b1 & ~b0, acg_like = b0 & (b1 | ~b2);
    ok = gather4(b6 & ~b7 & ~b3 & ((b4 & t_like) | (~b4 & acg_like)));
}

constexpr int kStreamGroups = 4;                       // 16-base groups per lane (their loads are in flight together)
constexpr int kStreamTile = kStreamGroups * 256;       // groups per workgroup

__device__ __forceinline__ void load_group(const uint8_t *__restrict__ bases, int64_t p, int64_t total, bool aligned, uint32_t x[4])
{
    if (p + 16 <= total && aligned) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + p);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {                                               // unaligned read set, the last partial group, or nothing
#pragma unroll
        for (int d = 0; d < 4; d++) {
            x[d] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (p + 4 * d + j < total) x[d] |= static_cast<uint32_t>(bases[p + 4 * d + j]) << (8 * j);
        }
    }
}

// One workgroup: 1024 groups of 16 positions.  Every lane classifies its groups (P0, P1, P2 and the validity bits); the
// validity bits of the whole tile, its read-end bits and its dropped bits (+ two groups of look-ahead each) meet in LDS
// as 16-bit pieces, and every lane then derives U for its own groups from three consecutive pieces of each: bit t of a
// group = the 32 positions from t on are valid and not dropped (AND over a 32-bit window, log steps) and no read ends
// among the first 31 of them (OR over a 31-bit window).
__global__ __launch_bounds__(256) void eref_streams_kernel(const uint8_t *__restrict__ all_bases,
                                                           const int64_t *__restrict__ offsets, int64_t total,
                                                           const uint16_t *__restrict__ ends16,
                                                           const uint16_t *__restrict__ dropped16,
                                                           uint16_t *__restrict__ s0, uint16_t *__restrict__ s1,
                                                           uint16_t *__restrict__ su)
{
    __shared__ uint16_t ok_lds[kStreamTile + 2], en_lds[kStreamTile + 2];
    const uint8_t *bases = all_bases + offsets[0];
    const bool aligned = (reinterpret_cast<uintptr_t>(bases) & 15) == 0;
    const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kStreamTile;
    const int64_t n_groups = (total + 15) >> 4;
    uint32_t x[kStreamGroups + 1][4];
    uint16_t en[kStreamGroups + 1], dr[kStreamGroups + 1];
#pragma unroll
    for (int k = 0; k <= kStreamGroups; k++) {
        if (k == kStreamGroups && threadIdx.x >= 2) break;         // the last round is the two look-ahead groups
        const int64_t i = g0 + (k < kStreamGroups ? k * 256 : kStreamTile) + threadIdx.x;
        load_group(bases, i * 16, total, aligned, x[k]);
        en[k] = i < n_groups ? ends16[i] : static_cast<uint16_t>(0);          // (the bit arrays are padded: see the caller)
        dr[k] = (dropped16 && i < n_groups) ? dropped16[i] : static_cast<uint16_t>(0);
    }
#pragma unroll
    for (int k = 0; k <= kStreamGroups; k++) {
        if (k == kStreamGroups && threadIdx.x >= 2) break;
        uint32_t o0 = 0, o1 = 0, ok = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t a, b, v;
            class_bits4(x[k][d], a, b, v);
            o0 |= a << (4 * d); o1 |= b << (4 * d); ok |= v << (4 * d);
        }
        const int local = (k < kStreamGroups ? k * 256 : kStreamTile) + static_cast<int>(threadIdx.x);
        ok_lds[local] = static_cast<uint16_t>(ok & ~static_cast<uint32_t>(dr[k]));
        en_lds[local] = en[k];
        const int64_t i = g0 + local;
        if (k < kStreamGroups && i < n_groups) {
            s0[i] = static_cast<uint16_t>(o0); s1[i] = static_cast<uint16_t>(o1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kStreamGroups; k++) {
        const int local = k * 256 + static_cast<int>(threadIdx.x);
        const int64_t i = g0 + local;
        if (i >= n_groups) break;
        unsigned long long a = static_cast<unsigned long long>(ok_lds[local]) | (static_cast<unsigned long long>(ok_lds[local + 1]) << 16) |
                               (static_cast<unsigned long long>(ok_lds[local + 2]) << 32);
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) a &= a >> s;             // bit t: positions t .. t+31 all valid
        unsigned long long e = static_cast<unsigned long long>(en_lds[local]) | (static_cast<unsigned long long>(en_lds[local + 1]) << 16) |
                               (static_cast<unsigned long long>(en_lds[local + 2]) << 32);
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) e |= e >> s;
        e |= e >> 15;                                             // bit t: a read end in [t, t+30]
        su[i] = static_cast<uint16_t>(a & ~e);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Level 1 as a counting sort in LDS (third version of this kernel; the first two staged a row per bucket with slack and
// wrote one partial-wave store per row).
//   1. the lane computes its keys (registers) and counts them per bucket in an LDS histogram; the returning add gives
//      the key its rank in the row;
//   2. one wave turns the histogram into row starts, every row padded to a multiple of 5 slots (one 16-byte group of five
//      25-bit records and a count, see pack_group: the pad slots are never written, the group's count says how many are keys);
//   3. meanwhile 128 lanes of two other waves reserve the runs (in groups) in the bucket regions: these global atomics are
//      in flight during 2. and 4.;
//   4. every lane places its keys at row start + rank;
//   5. the compact, bucket-sorted tile is swept group by group: five LDS reads, the row from the first key's own top bits,
//      its destination one LDS read, one 16-byte store.
// What bounds it (kernel cut short after each step, 1M-contig set): launch + stream loads + keys 1.5 ms, + histogram and
// row starts 1.7, + placement 2.6, + reservations and sweep without the stores 3.4, all of it 4.1 -- each workgroup is a
// chain of latencies of ~11 us whatever its tile size (the same with 256 threads, with 10 positions per lane), so the
// throughput is the key slots the LDS of a CU holds (4 workgroups) divided by that latency; HBM moves 10.7 GB in that
// time, half of what it could.  No change from: one LDS atomic less per key, earlier reservations, unrolled sweep
// (no store waits for the previous one), 95- to 380-byte runs, 8 to 64 replicas, contiguous instead of scattered stores,
// touching the stream lines of a later tile of the same XCD (-2 %).  Non-temporal stores: +25 %.
// ------------------------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS accesses only.  __syncthreads() also fences global memory, i.e. waits (vmcnt) for every
// global atomic and store the wave has in flight -- which is exactly what the kernels below want to keep in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (VALU only).  __shfl_up is a ds_bpermute: six of them in
// a row queue up behind the LDS traffic of the whole CU -- in the level-1 kernel the scan of ONE wave, which the other
// seven wait for at a barrier, took 2.2 us that way.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);        // row_shr:1  (within rows of 16 lanes)
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);        // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);        // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);        // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2 and 3
    return v;
}

// Level-1 records: the bucket implies the top 7 bits of a key, so a record keeps 25.  FIVE records and a 3-bit count share one
// 16-byte group -- bits [25 i, 25 i + 25) = key i & 0x1ffffff, bits [125, 128) = how many of the five are keys (the rest of a
// run's last group is padding) -- 3.2 bytes per key instead of 4, every group, run and region still 16-byte aligned, and no
// sentinel keys: level 2 reads the count.  Region capacities, cursors and run sizes are counted in groups.
constexpr uint32_t kGroupKeys = 5;
constexpr uint32_t kKeyMask25 = (1u << kL1Shift) - 1;
__device__ __forceinline__ uint4 pack_group(const uint32_t k[5], uint32_t count)
{
    const unsigned long long k0 = k[0] & kKeyMask25, k1 = k[1] & kKeyMask25, k2 = k[2] & kKeyMask25, k3 = k[3] & kKeyMask25, k4 = k[4] & kKeyMask25;
    const unsigned long long lo = k0 | (k1 << 25) | (k2 << 50);
    const unsigned long long hi = (k2 >> 14) | (k3 << 11) | (k4 << 36) | (static_cast<unsigned long long>(count) << 61);
    return uint4{static_cast<uint32_t>(lo), static_cast<uint32_t>(lo >> 32), static_cast<uint32_t>(hi), static_cast<uint32_t>(hi >> 32)};
}
__device__ __forceinline__ uint32_t unpack_group(const uint4 &g, uint32_t k[5])          // -> count
{
    const unsigned long long lo = g.x | (static_cast<unsigned long long>(g.y) << 32), hi = g.z | (static_cast<unsigned long long>(g.w) << 32);
    k[0] = static_cast<uint32_t>(lo) & kKeyMask25;
    k[1] = static_cast<uint32_t>(lo >> 25) & kKeyMask25;
    k[2] = static_cast<uint32_t>((lo >> 50) | (hi << 14)) & kKeyMask25;
    k[3] = static_cast<uint32_t>(hi >> 11) & kKeyMask25;
    k[4] = static_cast<uint32_t>(hi >> 36) & kKeyMask25;
    return static_cast<uint32_t>(hi >> 61);
}

// SHARE: the call counts a share of the key space (KeyBuckets): keys of other buckets are dropped before the histogram.  A
// template parameter, not a test of o.keys in the loop: the bucket test is ~6 instructions per key, +0.9 ms on the whole set.
template <int P, int THREADS, bool SHARE>
__global__ __launch_bounds__(THREADS) void eref_bin1_sort_kernel(const uint32_t *__restrict__ s0,
                                                                     const uint32_t *__restrict__ s1,
                                                                     const uint32_t *__restrict__ su,
                                                                     int64_t pos_lo, int64_t pos_hi,
                                                                     CoderMasks masks, BinOut o)
{
    constexpr int kMaxKeys = THREADS * P * 3 + kL1Buckets * (kGroupKeys - 1);    // every key of the tile + the pad slots of every row
    __shared__ __attribute__((aligned(16))) uint32_t tile[kMaxKeys];
    __shared__ uint32_t hist[kL1Buckets], room[kL1Buckets];
    __shared__ uint16_t start[kL1Buckets + 2];           // (16 bits: with 32 the P = 6 tile is 40 976 bytes, 16 more than a quarter of the CU's LDS)
    static_assert(kMaxKeys < 65536, "row starts are kept in 16 bits");
    static_assert(P != 6 || THREADS != 512 || sizeof(uint32_t) * (kMaxKeys + 3 * kL1Buckets + 1) + sizeof(uint16_t) * (kL1Buckets + 2) <= 160 * 1024 / 4,
                  "the 6-position tile must fit four times into the LDS of a CU");
    __shared__ uint32_t dst[kL1Buckets];                 // group index in o.buf of the row's first group minus the row's first group in the
                                                         // tile, modulo 2^32 (row starts are multiples of 5 slots; a slab's regions hold < 2^32 groups)
    __shared__ uint32_t any_partial;                     // some run of this tile did not fit its region whole (rare): check `room`
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef PALACE_STAMPS
    unsigned long long stamp_arr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long *stamps = (threadIdx.x == 0 && blockIdx.x % 7 == 0 && blockIdx.x / 7 < 65536) ? stamp_arr : nullptr;
#endif
    STAMP(stamps, 0);
    const int64_t p = pos_lo + (static_cast<int64_t>(blockIdx.x) * THREADS + threadIdx.x) * P;
    // every load of the lane is issued before anything else (the streams are padded: see the caller)
    const int64_t g = min(p, pos_hi) >> 5;
    const int sh = static_cast<int>(p & 31);
    uint32_t w[2][3], uw[2];
#pragma unroll
    for (int q = 0; q < 3; q++) { w[0][q] = s0[g + q]; w[1][q] = s1[g + q]; }
    uw[0] = su[g]; uw[1] = su[g + 1];
    if (threadIdx.x < kL1Buckets) hist[threadIdx.x] = 0;
    if (threadIdx.x == kL1Buckets) any_partial = 0;
    lds_barrier();
    // ---- 1. keys and histogram ----
    uint32_t u = __builtin_amdgcn_alignbit(uw[1], uw[0], sh) & ((1u << P) - 1);
    if (p >= pos_hi) u = 0;
    else if (p + P > pos_hi) u &= (1u << (pos_hi - p)) - 1;     // the slab's last lane
    uint32_t key[P][3], rank[P][3];
    if (u) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            lo[q] = __builtin_amdgcn_alignbit(w[q][1], w[q][0], sh);
            hi[q] = __builtin_amdgcn_alignbit(w[q][2], w[q][1], sh);
        }
#pragma unroll
        for (int t = 0; t < P; t++) {
            const uint32_t a0 = __builtin_amdgcn_alignbit(hi[0], lo[0], t), a1 = __builtin_amdgcn_alignbit(hi[1], lo[1], t);
            kmer_keys(masks, a0, a1, ~(a0 ^ a1), key[t]);                       // {A,G} = not ({A,T} xor {A,C})
            if ((u >> t) & 1u) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const uint32_t bk = key[t][i] >> kL1Shift;
                    rank[t][i] = (!SHARE || o.keys.bucket(bk)) ? atomicAdd(&hist[bk], 1u) : ~0u;        // its rank in the row; ~0: not this call's key
                }
            }
        }
    }
    STAMP(stamps, 1);
    lds_barrier();
    STAMP(stamps, 2);
    // ---- 2. row starts (wave 0: two rows per lane), pad slots; 3. meanwhile waves 1-2 reserve the padded runs in the
    // bucket regions (a run's size needs its own count only): these global atomics are in flight during 2. and 4. ----
    const uint32_t replica = blockIdx.x % kL1Replicas;
    const bool reserver = threadIdx.x >= 64 && threadIdx.x < 64 + kL1Buckets;
    const uint32_t my_row = threadIdx.x - 64;
    uint32_t my_pc = 0, my_g = 0;
    if (wave == 0) {                                     // rows start on multiples of 5 slots (a group); the pad slots stay unwritten
        const uint32_t c0 = hist[2 * lane], c1 = hist[2 * lane + 1];
        const uint32_t pc0 = (c0 + kGroupKeys - 1) / kGroupKeys * kGroupKeys, pc1 = (c1 + kGroupKeys - 1) / kGroupKeys * kGroupKeys;
        const uint32_t incl = wave_inclusive_scan(pc0 + pc1);
        const uint32_t a0 = incl - pc0 - pc1, a1 = a0 + pc0;
        start[2 * lane] = a0; start[2 * lane + 1] = a1;
        if (lane == 63) start[kL1Buckets] = incl;
    } else if (reserver) {
        my_pc = (hist[my_row] + kGroupKeys - 1) / kGroupKeys;                    // groups of the run
        if (my_pc) my_g = atomicAdd(&o.cursor[l1_cursor(my_row, replica)], my_pc);
    }
    lds_barrier();
    STAMP(stamps, 3);
    // ---- 4. placement ----
    if (u) {
#pragma unroll
        for (int t = 0; t < P; t++) {
            if ((u >> t) & 1u) {
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (!SHARE || rank[t][i] != ~0u) tile[start[key[t][i] >> kL1Shift] + rank[t][i]] = key[t][i];
            }
        }
    }
    if (reserver) {
        const uint32_t cap = o.caps.cap(my_row);
        const bool whole = static_cast<uint64_t>(my_g) + my_pc <= cap;
        room[my_row] = cap - min(cap, my_g);                      // groups of the run that fit
        dst[my_row] = static_cast<uint32_t>(l1_region_base(o.caps, my_row, replica) + my_g - start[my_row] / kGroupKeys);
        if (!whole) any_partial = 1;
    }
    lds_barrier();
    STAMP(stamps, 4);
    // ---- 5. sweep: a lane packs the five slots of a group into 16 bytes; unrolled, every LDS read of the lane first, then its
    // 16-byte stores back to back (as a loop the compiler made every iteration wait for the previous iteration's store to
    // be acknowledged: three HBM round trips in a row per tile) ----
    const uint32_t total = start[kL1Buckets];
    const bool check = any_partial != 0;                           // uniform
    constexpr int kSweeps = (kMaxKeys / static_cast<int>(kGroupKeys) + THREADS - 1) / THREADS;
    uint32_t slow = 0;
    uint4 k[kSweeps];
    uint32_t to[kSweeps], live = 0;                                // group index in o.buf; bit it of live: group it is stored
#pragma unroll
    for (int it = 0; it < kSweeps; it++) {
        const uint32_t gi = it * THREADS + threadIdx.x, x = gi * kGroupKeys;
        to[it] = 0;
        k[it] = uint4{0, 0, 0, 0};
        if (x < total) {
            uint32_t s5[kGroupKeys];
#pragma unroll
            for (uint32_t e = 0; e < kGroupKeys; e++) s5[e] = tile[x + e];
            const uint32_t row = s5[0] >> kL1Shift;                // the group's first slot is always a key: the row is its top bits
            const uint32_t in_row = x - start[row];
            k[it] = pack_group(s5, min(kGroupKeys, hist[row] - in_row));
            if (!check || in_row / kGroupKeys < room[row]) { to[it] = dst[row] + gi; live |= 1u << it; }
            else slow |= 1u << it;
        }
    }
#pragma unroll
    for (int it = 0; it < kSweeps; it++)
        if (live & (1u << it)) reinterpret_cast<uint4 *>(o.buf)[to[it]] = k[it];
    if (slow) {                                                    // the region is full: exact slow path
#pragma unroll 1
        for (int it = 0; it < kSweeps; it++) {
            if (!((slow >> it) & 1u)) continue;
            const uint32_t x = (it * THREADS + threadIdx.x) * kGroupKeys;
            const uint32_t row = tile[x] >> kL1Shift;
            const uint32_t n = min(kGroupKeys, hist[row] - (x - start[row]));
#pragma unroll 1
            for (uint32_t e = 0; e < n; e++) count_key_marked(tile[x + e], o.p1, o.p2, o.p3, o.touched);
        }
    }
    STAMP(stamps, 5);
#ifdef PALACE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(stamps, 6);
    if (stamps)
        for (int i = 0; i < 8; i++) palace_stamp_buf[(blockIdx.x / 7) * 8 + i] = stamp_arr[i];
#endif
}

__global__ __launch_bounds__(kBin2Threads) void eref_bin2_kernel(const unsigned int *__restrict__ cursor1,
                                                                 const uint32_t *__restrict__ buf1, DensityCaps caps1,
                                                                 Bin2Grid grid, Bin2Out o)
{
    __shared__ Stage2 st;
    // The grid is sized by the CAPACITY of every region, which follows the key density (bucket 0: twice the mean, bucket
    // 127: almost nothing), not by the largest region times the region count: half as many workgroups start only to
    // find nothing to do.  blockIdx.x -> (bucket, replica, tile): binary search in the 129-entry prefix.
    uint32_t lo = 0, hi = kL1Buckets;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (grid.first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const uint32_t b1 = lo, within = blockIdx.x - grid.first[b1], per_region = tiles_of_bucket(caps1, b1);
    const uint32_t replica = within / per_region, tile = within % per_region;
    const uint32_t n1 = min(cursor1[l1_cursor(b1, replica)], caps1.cap(b1));       // groups
    const uint32_t start = tile * kTile2Groups;
    if (start >= n1) return;                               // uniform for the workgroup
    const uint32_t end = min(n1, start + kTile2Groups);
    // all of a thread's groups are loaded (16 bytes each) before the first append
    const uint4 *src = reinterpret_cast<const uint4 *>(buf1) + l1_region_base(caps1, b1, replica);
    uint4 v[kGroups2PerThread];
#pragma unroll
    for (int it = 0; it < kGroups2PerThread; it++) {
        const uint32_t i = start + it * kBin2Threads + threadIdx.x;
        v[it] = i < end ? src[i] : uint4{0, 0, 0, 0};     // (count 0: nothing to append)
    }
    if (threadIdx.x < kL2Rows) st.rows[threadIdx.x] = threadIdx.x * kRowSlots;
    const uint32_t cap = fine_sub_cap(o.caps, b1), xcd = xcc_id();          // this XCD's share of every fine region
    __syncthreads();
    // The appends carry no branches: the kernel is bound by its vector and scalar instruction issue (~45 vector instructions
    // per key before this form, 77 % of the kernel's time at full issue rate), and every `if` around an LDS operation costs an
    // exec-mask round.  A slot of a group that holds no key (only a run's last group has such) adds 0 to a row; a key
    // that finds no room, and a slot that is none, write to the spare slot behind the rows.
    using Homeless = std::conditional_t<(kGroups2PerThread * kGroupKeys > 32), unsigned long long, uint32_t>;
    Homeless homeless = 0;                                 // bit 5 * it + e: that key found its row full
#pragma unroll
    for (int it = 0; it < kGroups2PerThread; it++) {
        uint32_t k[kGroupKeys];
        const uint32_t n = unpack_group(v[it], k);
        uint32_t at[kGroupKeys];
#pragma unroll
        for (uint32_t e = 0; e < kGroupKeys; e++) at[e] = atomicAdd(&st.rows[k[e] >> kFineBits], e < n ? 1u : 0u);   // (25-bit record: bits 24..16 are the fine row)
#pragma unroll
        for (uint32_t e = 0; e < kGroupKeys; e++) {
            const uint32_t row = k[e] >> kFineBits;
            const bool is_key = e < n, fits = at[e] < (row + 1) * kRowSlots;
            st.slot[is_key && fits ? at[e] : kStage2Slots] = static_cast<uint16_t>(k[e]);
            homeless |= static_cast<Homeless>(is_key && !fits ? 1u : 0u) << (kGroupKeys * it + e);
        }
    }
    if (homeless) {
        // Row full (a row holds 72 of the tile's keys, mean 50: about one key in a thousand): the key goes to its fine region
        // as a run of its own.  It used to take the direct-atomic path, which marks the fine bucket `touched` -- and with
        // ~50 000 such keys per launch 38 % of the 65 536 count workgroups then began by seeding their 24 KiB of plane slices
        // from HBM (0.6 GB per launch, PMC) for the sake of one or two keys.
        // (unrolled: v[] and k[] indexed by a loop variable would live in scratch memory)
#pragma unroll
        for (int it = 0; it < kGroups2PerThread; it++) {
            if (!((homeless >> (kGroupKeys * it)) & 31u)) continue;
            uint32_t k[kGroupKeys];
            unpack_group(src[start + it * kBin2Threads + threadIdx.x], k);        // (loaded again: keeping v[] alive until here costs 20 registers,
                                                                                  //  and with them the second workgroup of the CU)
#pragma unroll
            for (uint32_t e = 0; e < kGroupKeys; e++) {
                if (!((homeless >> (kGroupKeys * it + e)) & 1u)) continue;
                const uint32_t row = k[e] >> kFineBits;
                const uint32_t gg = atomicAdd(&o.cursor[(b1 * kL2Rows + row) * kXcds + xcd], 1u);
                if (gg < cap) o.buf[fine_region_base(o.caps, b1, row) + static_cast<uint64_t>(xcd) * cap + gg] = static_cast<uint16_t>(k[e]);
                else count_key_marked((b1 << kL1Shift) | k[e], o.p1, o.p2, o.p3, o.touched);   // region full as well: exact slow path
            }
        }
    }
    __syncthreads();
    // flush: a wave owns 32 rows; the first 32 lanes reserve the runs (one 128-byte piece of the cursor array) and work
    // out the destination pointers, then the wave walks the rows with count / source / pointer in SGPRs
    constexpr int rows_per_wave = kL2Rows / (kBin2Threads / 64);
    const int lane = threadIdx.x & 63;
    const int row0 = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) * rows_per_wave;
    uint32_t c = 0, g = 0, p_lo = 0, p_hi = 0;
    bool over = false;
    if (lane < rows_per_wave) {
        const uint32_t row = row0 + lane;
        c = min(st.rows[row], (row + 1) * kRowSlots) - row * kRowSlots;
        if (c) g = atomicAdd(&o.cursor[(b1 * kL2Rows + row) * kXcds + xcd], c);
        over = static_cast<uint64_t>(g) + c > cap;
        const uint64_t ptr = reinterpret_cast<uint64_t>(o.buf + fine_region_base(o.caps, b1, row) + static_cast<uint64_t>(xcd) * cap + min(g, cap));
        p_lo = static_cast<uint32_t>(ptr); p_hi = static_cast<uint32_t>(ptr >> 32);
    }
    const unsigned long long over_rows = __ballot(over);
    const uint32_t c_fast = over ? 0u : c;                  // (a run that does not fit is left to the pass below: no lane stores it here --
                                                            //  cheaper than a scalar test per row: the flush is bound by the CU's scalar unit)
    // rows in batches of 16: all LDS reads of a batch first, then its stores back to back (a row holds at most 72 slots:
    // one more, short pass for the fullest rows)
#pragma unroll
    for (int j0 = 0; j0 < rows_per_wave; j0 += 16) {
        uint16_t k[16];
#pragma unroll
        for (int j = 0; j < 16; j++) k[j] = st.slot[(row0 + j0 + j) * kRowSlots + lane];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t cj = __builtin_amdgcn_readlane(c_fast, j0 + j);
            const uint32_t bl = __builtin_amdgcn_readlane(p_lo, j0 + j), bh = __builtin_amdgcn_readlane(p_hi, j0 + j);
            global_u16 *dst = reinterpret_cast<global_u16 *>((static_cast<uint64_t>(bh) << 32) | bl);
            if (lane < cj) dst[lane] = k[j];
        }
    }
    unsigned long long more = __ballot(lane < rows_per_wave && (c > 64 || over));
    while (more) {
        const int j = __builtin_ctzll(more);
        more &= more - 1;
        const uint32_t cj = __builtin_amdgcn_readlane(c, j);
        const uint32_t bl = __builtin_amdgcn_readlane(p_lo, j), bh = __builtin_amdgcn_readlane(p_hi, j);
        global_u16 *dst = reinterpret_cast<global_u16 *>((static_cast<uint64_t>(bh) << 32) | bl);
        const uint16_t *row_slots = st.slot + (row0 + j) * kRowSlots;
        if ((over_rows >> j) & 1ull) {                             // rare: the run does not fit its region
            const uint32_t gj = __builtin_amdgcn_readlane(g, j);
            const uint32_t room = cap - (gj < cap ? gj : cap);
            const uint32_t fine = b1 * kL2Rows + row0 + j;
#pragma unroll 1
            for (uint32_t q = 0; q < cj; q += 64) {
                if (q + lane < cj) {
                    const uint32_t kk = row_slots[q + lane];
                    if (q + lane < room) dst[q + lane] = static_cast<uint16_t>(kk);
                    else count_key_marked((fine << kFineBits) | kk, o.p1, o.p2, o.p3, o.touched);
                }
            }
            continue;
        }
        for (uint32_t q = 64; q < cj; q += 64)
            if (lane + q < cj) dst[lane + q] = row_slots[lane + q];
    }
}

// one workgroup per fine bucket: its 2^16-key slice of the three planes (3 x 8 KiB) lives in LDS, is seeded from the
// global planes, takes the bucket's 16-bit keys with LDS atomicOr climbing 1 -> 2 -> 3, and is written back
constexpr int kCountThreads = 256;
// CLEAN: the planes were all zero when this launch began (first count after a reset), so a slice is only read when the
// overflow path of the partition kernels has written into it (`touched`); otherwise it starts from zero in LDS.
// FINAL (implies CLEAN): the caller reads nothing but the ">= 3" plane afterwards (Phase B does not) -- the two lower planes
// are not written at all, and where the overflow path of the partition kernels put bits into them (touched buckets) they are
// zeroed again: after the launch they are all zero, as after a reset.
// PROBE (with FINAL, whole key space): Phase B's channel-0 probe of one DB rides along -- while the bucket's final ">= 3" slice
// is in LDS, the DB positions whose channel-0 index falls into the bucket (the per-DB probe index, grouped by these very
// buckets) are tested against it and their hit bytes set: the scan that follows needs neither the probe kernel nor its read
// of the plane (palace_eref_attach_probe_index).
// PROBE = 2 (round 5, with FINAL, whole key space; option probe_all_sets): ALL FOUR entry sets of the DB's probe index (channels 0, 1, 2
// and the sentinels) are tested here, so nothing reads the ">= 3" plane afterwards -- and the slice is not written: the plane never
// exists in HBM (0.54 GB of write-back, the 0.5 GB reset of the next step and the probe kernel's 0.5 GB read of it fall away).  Where
// the overflow path of the partition kernels put bits into the global planes (touched buckets) all three slices are zeroed again:
// after the launch the table is all zero, as after a reset, and holds nothing (palace_ctx::planeless).
constexpr int kProbeSetsMax = 4;
struct ProbeArgs {
    const unsigned long long *first[kProbeSetsMax];         // [65537] start of every fine bucket's entries of the set (multiples of 8)
    const uint16_t *keys16[kProbeSetsMax];                  // index & 0xffff of the set's entries, grouped by fine bucket
    uint8_t *ehits[kProbeSetsMax];                          // a BIT per entry, in entry order (byte i = entries 8 i .. 8 i + 7), zero before the launch
    // PROBE == 2: the sentinels that hit are carried to position order right here (entry -> sentinel ordinal, a byte per sentinel, zero
    // before the launch) -- what eref_ehits_scatter_kernel does for a scan that probes for itself; their entry-order bits are not stored
    const uint32_t *pos_s;
    uint8_t *sent_bytes;
};
// the sentinels of one 16-byte vector (entries 8 i .. 8 i + 7) that hit: every look-up of `pos_s` first, then the byte stores
__device__ __forceinline__ void scatter_sentinels(const ProbeArgs &pr, unsigned long long i, uint32_t m)
{
    uint32_t p[8];
#pragma unroll
    for (int e = 0; e < 8; e++) p[e] = ((m >> e) & 1u) ? pr.pos_s[i * 8 + e] : ~0u;
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (p[e] != ~0u) pr.sent_bytes[p[e]] = 1;
}
// PROBE = 3 (option probe_all_sets 2: a rank that counted a SHARE OF THE READS of a sample): as PROBE = 2, but what is left per
// entry is not the hit bit but the partial COUNT, 0..3 in two bits (the three unary slices added up; 16 bits per vector of eight
// entries, in `ehits`, which then points at the index's count block) -- ranks sum the counts of an entry range
// (palace_eref_entry_hits_from_counts: min(3, sum of min(3, c_r)) = min(3, sum of c_r), exact) and gather the hit bits, so that no
// plane crosses a link; the sentinels' hits are carried to position order by the scan (their counts are partial here).
// the eight 16-bit entries of one 16-byte vector against a 2^16-bit slice in LDS -> a byte of hit bits
__device__ __forceinline__ uint32_t count_vector(const uint32_t *__restrict__ l1, const uint32_t *__restrict__ l2, const uint32_t *__restrict__ l3, const uint4 &v)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t k0 = d[e] & 0xffffu, k1 = d[e] >> 16;
        const uint32_t c0 = ((l1[k0 >> 5] >> (k0 & 31)) & 1u) + ((l2[k0 >> 5] >> (k0 & 31)) & 1u) + ((l3[k0 >> 5] >> (k0 & 31)) & 1u);
        const uint32_t c1 = ((l1[k1 >> 5] >> (k1 & 31)) & 1u) + ((l2[k1 >> 5] >> (k1 & 31)) & 1u) + ((l3[k1 >> 5] >> (k1 & 31)) & 1u);
        m |= c0 << (4 * e);
        m |= c1 << (4 * e + 2);
    }
    return m;
}

template <bool CLEAN, bool FINAL = false, int PROBE = 0>
__global__ __launch_bounds__(kCountThreads) void eref_lds_count_kernel(const unsigned int *__restrict__ cursor,
                                                                       const uint16_t *__restrict__ binned,
                                                                       DensityCaps caps, uint32_t *__restrict__ p1,
                                                                       uint32_t *__restrict__ p2,
                                                                       uint32_t *__restrict__ p3,
                                                                       const unsigned int *__restrict__ touched, KeyBuckets share,
                                                                       ProbeArgs pr)
{
    __shared__ uint32_t l1[kFineWords], l2[kFineWords], l3[kFineWords];
    const uint32_t b = blockIdx.x, b1 = b / kL2Rows;
    if (!share.bucket(b1)) return;                          // a call that counts a share of the key space: not its bucket
    constexpr int kProbeBatch = 2;                          // 16-byte vectors (8 entries each) per thread and batch: a bucket's ~3000 entries are ONE batch
    unsigned long long pe0 = 0, phi = 0;                    // the bucket's entries, in vectors of 8
    if (PROBE) { pe0 = pr.first[0][b] / 8; phi = pr.first[0][b + 1] / 8; }
    // PROBE == 2: the other three sets' entries of the bucket (vectors [q0, q1) of each); channels 1 and 2 take up to kLateCh vectors per
    // thread in one batch (the densest buckets hold twice the mean of ~380), the sentinels one
    constexpr int kLateCh = 3, kLateSent = 1;
#ifndef PALACE_PROBE_GROUP
#define PALACE_PROBE_GROUP 2
#endif
    constexpr int kProbeGroup = PALACE_PROBE_GROUP;
    constexpr bool ALL = PROBE >= 2, COUNTS = PROBE == 3;   // every entry set is tested here / what is left per entry is its partial count
    auto code = [&](const uint4 &vec) -> uint32_t { return COUNTS ? count_vector(l1, l2, l3, vec) : probe_vector(l3, vec); };
    auto put_code = [&](int k, unsigned long long i, uint32_t m) {
        if (COUNTS) reinterpret_cast<uint16_t *>(pr.ehits[k])[i] = static_cast<uint16_t>(m);
        else pr.ehits[k][i] = static_cast<uint8_t>(m);
    };
    unsigned long long q0[kProbeSetsMax - 1] = {0, 0, 0}, q1[kProbeSetsMax - 1] = {0, 0, 0};
    if (ALL) {
#pragma unroll
        for (int k = 1; k < kProbeSetsMax; k++) { q0[k - 1] = pr.first[k][b] / 8; q1[k - 1] = pr.first[k][b + 1] / 8; }
    }
    // the bucket's keys lie in eight sub-regions (one per XCD that wrote them); as one sequence of 16-byte vectors of eight
    // keys: vector j belongs to sub-region x with first[x] <= j < first[x + 1]
    const uint32_t sub_cap = fine_sub_cap(caps, b1);
    uint32_t n_sub[kXcds], first[kXcds + 1];
    first[0] = 0;
#pragma unroll
    for (int x = 0; x < kXcds; x++) {
        n_sub[x] = min(cursor[b * kXcds + x], sub_cap);
        first[x + 1] = first[x] + (n_sub[x] + 7) / 8;
    }
    const uint32_t n8 = first[kXcds];
    const size_t w0 = static_cast<size_t>(b) * kFineWords;
    if (n8 == 0) {                                         // uniform for the whole workgroup
        if (FINAL && ((touched[b >> 5] >> (b & 31)) & 1u)) {       // only overflow keys: plane 3 is right as it is, the lower two go back to zero
            uint4 *z1 = reinterpret_cast<uint4 *>(p1 + w0), *z2 = reinterpret_cast<uint4 *>(p2 + w0);
            if (COUNTS) {                                          // (partial counts need the lower slices as well: read before they are zeroed)
                for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) {
                    reinterpret_cast<uint4 *>(l1)[i] = z1[i];
                    reinterpret_cast<uint4 *>(l2)[i] = z2[i];
                }
            }
            for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) z1[i] = z2[i] = uint4{0, 0, 0, 0};
            if (ALL || (PROBE && phi > pe0)) {                     // ... and is what the DB's positions of this bucket are tested against
                const uint4 *s3 = reinterpret_cast<const uint4 *>(p3 + w0);
                for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) reinterpret_cast<uint4 *>(l3)[i] = s3[i];
                __syncthreads();
                const uint4 *pv = reinterpret_cast<const uint4 *>(pr.keys16[0]);
                for (unsigned long long i = pe0 + threadIdx.x; i < phi; i += kCountThreads) put_code(0, i, code(pv[i]));
                if (ALL) {
#pragma unroll
                    for (int k = 1; k < kProbeSetsMax; k++) {
                        const uint4 *qv = reinterpret_cast<const uint4 *>(pr.keys16[k]);
                        for (unsigned long long i = q0[k - 1] + threadIdx.x; i < q1[k - 1]; i += kCountThreads) {
                            const uint32_t m = code(qv[i]);
                            if (k == 3 && !COUNTS) scatter_sentinels(pr, i, m);
                            else put_code(k, i, m);
                        }
                    }
                    uint4 *z3 = reinterpret_cast<uint4 *>(p3 + w0);        // the plane goes back to zero as well: nothing reads it any more
                    for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) z3[i] = uint4{0, 0, 0, 0};
                }
            }
        }
        return;                                            // (no key at all: the slice is zero, no position of the DB hits)
    }
    const uint4 *g1 = reinterpret_cast<const uint4 *>(p1 + w0), *g2 = reinterpret_cast<const uint4 *>(p2 + w0),
                *g3 = reinterpret_cast<const uint4 *>(p3 + w0);
    // (sub-regions start on 16-byte boundaries and their capacity is a multiple of 8 keys, so the last vector of one may run
    // past its count but not past the sub-region); the first batch of key loads is issued together with the seeds
    const uint4 *keys = reinterpret_cast<const uint4 *>(binned + fine_region_base(caps, b1, b % kL2Rows));
    auto locate = [&](uint32_t j, uint32_t &valid) -> const uint4 * {      // vector j and how many of its 8 keys count
        // first[x] and n_sub[x] by multiply-adds over the eight (uniform) entries: indexed with x, or picked by a chain of
        // selects, the compiler puts the arrays into scratch memory and every vector's address waits for two scratch loads
        uint32_t x = 0, f = 0, ns = n_sub[0];
#pragma unroll
        for (int k = 1; k < kXcds; k++) {
            const uint32_t ge = j >= first[k] ? 1u : 0u;
            x += ge; f += ge * (first[k] - first[k - 1]); ns += ge * (n_sub[k] - n_sub[k - 1]);
        }
        const uint32_t local = j - f;
        valid = min(8u, ns - 8 * local);
        return keys + (static_cast<size_t>(x) * sub_cap) / 8 + local;
    };
    constexpr int kBatch = 4;
    uint4 v[kBatch];
    uint32_t ok[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; u++) {
        const uint32_t i = threadIdx.x + u * kCountThreads;
        ok[u] = 0; v[u] = uint4{0, 0, 0, 0};
        if (i < n8) v[u] = *locate(i, ok[u]);
    }
    // PROBE: the bucket's entries of the DB's probe index are requested now and are in flight during the whole count phase
    uint4 pcur[kProbeBatch];
    if (PROBE) {
        const uint4 *pv = reinterpret_cast<const uint4 *>(pr.keys16[0]);
#pragma unroll
        for (int u = 0; u < kProbeBatch; u++) {
            const unsigned long long i = pe0 + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
            pcur[u] = i < phi ? pv[i] : uint4{0, 0, 0, 0};
        }
    }
    const bool seed = !CLEAN || ((touched[b >> 5] >> (b & 31)) & 1u);      // uniform for the workgroup
    // (two loops, not `seed ? g[i] : zero` in one: for that the compiler selects between the global ADDRESS and the address of
    // a zero it keeps in scratch memory, and the clean case pays three flat loads per lane and round all the same)
    if (seed) {
        for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) {
            reinterpret_cast<uint4 *>(l1)[i] = g1[i];
            reinterpret_cast<uint4 *>(l2)[i] = g2[i];
            reinterpret_cast<uint4 *>(l3)[i] = g3[i];
        }
    } else {
        for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads)
            reinterpret_cast<uint4 *>(l1)[i] = reinterpret_cast<uint4 *>(l2)[i] = reinterpret_cast<uint4 *>(l3)[i] = uint4{0, 0, 0, 0};
    }
    __syncthreads();
    auto apply = [&](uint32_t k) {
        const uint32_t w = k >> 5, bit = 1u << (k & 31);
        if (atomicOr(&l1[w], bit) & bit)
            if (atomicOr(&l2[w], bit) & bit) atomicOr(&l3[w], bit);
    };
    for (uint32_t i0 = threadIdx.x; i0 < n8; i0 += kBatch * kCountThreads) {
        uint4 nx[kBatch];
        uint32_t nok[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {                 // the next batch is in flight while this one is applied
            const uint32_t i = i0 + (kBatch + u) * kCountThreads;
            nok[u] = 0; nx[u] = uint4{0, 0, 0, 0};
            if (i < n8) nx[u] = *locate(i, nok[u]);
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (2 * e < static_cast<int>(ok[u])) apply(d[e] & 0xffffu);
                if (2 * e + 1 < static_cast<int>(ok[u])) apply(d[e] >> 16);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) { v[u] = nx[u]; ok[u] = nok[u]; }
    }
    // PROBE == 2: the entries of the other three sets are requested now, in the registers the key vectors have left (behind the count
    // phase, not in flight during it like set 0's: seven more vectors per lane then would cost the kernel a workgroup per CU)
    uint4 lch[2][kLateCh], lsn[kLateSent];
    if (ALL) {
        asm volatile("" ::: "memory");                     // (not hoisted above the count loop: 28 registers live through it would halve the occupancy)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint4 *qv = reinterpret_cast<const uint4 *>(pr.keys16[1 + c]);
#pragma unroll
            for (int u = 0; u < kLateCh; u++) {
                const unsigned long long i = q0[c] + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
                lch[c][u] = i < q1[c] ? qv[i] : uint4{0, 0, 0, 0};
            }
        }
        const uint4 *qs = reinterpret_cast<const uint4 *>(pr.keys16[3]);
#pragma unroll
        for (int u = 0; u < kLateSent; u++) {
            const unsigned long long i = q0[2] + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
            lsn[u] = i < q1[2] ? qs[i] : uint4{0, 0, 0, 0};
        }
    }
    __syncthreads();
    // PROBE: the entries are tested against the final slice, eight to a byte of hit bits IN ENTRY ORDER (what position an entry
    // belongs to is the business of the scatter kernel of the scan).  No global store is issued before the last entry load has
    // come back: a wave's vmcnt counts loads and stores together and the compiler has to wait for ALL of them once both kinds are
    // in flight.  A bucket with more entries than one batch holds (> 4096) tests the rest behind the slice's write-back.
    uint32_t pm[kProbeBatch], lm[2][kLateCh], sm[kLateSent];
    if (PROBE) {
#pragma unroll
        for (int u = 0; u < kProbeBatch; u++) {
            pm[u] = code(pcur[u]);
            if (ALL) asm volatile("" : "+v"(pm[u]) : : "memory");
        }
    }
    if (ALL) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int u = 0; u < kLateCh; u++) {
                lm[c][u] = code(lch[c][u]);
                // (kProbeGroup vectors at a time: left alone the compiler issues the LDS reads of all nine vectors first -- 72 registers
                // of results, 136 in all, three workgroups per CU instead of six)
                if (COUNTS || (c * kLateCh + u) % kProbeGroup == kProbeGroup - 1) asm volatile("" : "+v"(lm[c][u]) : : "memory");
            }
#pragma unroll
        for (int u = 0; u < kLateSent; u++) { sm[u] = code(lsn[u]); asm volatile("" : "+v"(sm[u]) : : "memory"); }
    }
    if (ALL && !COUNTS) {                                  // (the position look-ups of the sentinels that hit: before any store of this wave is in flight)
#pragma unroll
        for (int u = 0; u < kLateSent; u++) {
            const unsigned long long i = q0[2] + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
            if (i < q1[2] && sm[u]) scatter_sentinels(pr, i, sm[u]);
        }
    }
    uint4 *o1 = reinterpret_cast<uint4 *>(p1 + w0), *o2 = reinterpret_cast<uint4 *>(p2 + w0),
          *o3 = reinterpret_cast<uint4 *>(p3 + w0);
    if (ALL) {                                             // the slice stays in LDS; a seeded one (overflow keys) goes back to zero in all planes
        if (seed)
            for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) o1[i] = o2[i] = o3[i] = uint4{0, 0, 0, 0};
    } else {
        for (int i = threadIdx.x; i < kFineWords / 4; i += kCountThreads) {
            if (!FINAL) {
                o1[i] = reinterpret_cast<const uint4 *>(l1)[i];
                o2[i] = reinterpret_cast<const uint4 *>(l2)[i];
            } else if (seed) {
                o1[i] = o2[i] = uint4{0, 0, 0, 0};
            }
            o3[i] = reinterpret_cast<const uint4 *>(l3)[i];
        }
    }
    if (PROBE) {
#pragma unroll
        for (int u = 0; u < kProbeBatch; u++) {
            const unsigned long long i = pe0 + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
            if (i < phi) put_code(0, i, pm[u]);
        }
    }
    if (ALL) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int u = 0; u < kLateCh; u++) {
                const unsigned long long i = q0[c] + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
                if (i < q1[c]) put_code(1 + c, i, lm[c][u]);
            }
        if (COUNTS) {
#pragma unroll
            for (int u = 0; u < kLateSent; u++) {
                const unsigned long long i = q0[2] + threadIdx.x + static_cast<unsigned long long>(u) * kCountThreads;
                if (i < q1[2]) put_code(3, i, sm[u]);
            }
        }
    }
    if (PROBE) {                                           // buckets with more entries than the batches hold (another DB's density): the rest, one by one
        const uint4 *pv = reinterpret_cast<const uint4 *>(pr.keys16[0]);
        for (unsigned long long i = pe0 + threadIdx.x + static_cast<unsigned long long>(kProbeBatch) * kCountThreads; i < phi; i += kCountThreads)
            put_code(0, i, code(pv[i]));
    }
    if (ALL) {
#pragma unroll
        for (int k = 1; k < kProbeSetsMax; k++) {
            const uint4 *qv = reinterpret_cast<const uint4 *>(pr.keys16[k]);
            const int held = k == 3 ? kLateSent : kLateCh;
            for (unsigned long long i = q0[k - 1] + threadIdx.x + static_cast<unsigned long long>(held) * kCountThreads; i < q1[k - 1]; i += kCountThreads) {
                const uint32_t m = code(qv[i]);
                if (k == 3 && !COUNTS) scatter_sentinels(pr, i, m);
                else put_code(k, i, m);
            }
        }
    }
}

}  // namespace palace

using namespace palace;

static_assert(kSets == kProbeSetsMax, "the count kernel's probe arguments hold every entry set");

namespace {
int plan_count(palace_ctx *ctx, int64_t total_bases, CountPlan *pl)
{
    // Large read sets are processed in slabs of at most slab_bases_max positions (the planes accumulate across slabs),
    // which bounds the workspace whatever the input size.  Slab size: 2^30 positions (workspace ~25 GB); 2^31 when the
    // read set is larger than that AND the device has the room (~50 GB of workspace) -- every slab rewrites all plane
    // slices once, so fewer slabs mean less traffic.
    int64_t default_slab = 1ll << 30;
    if (ctx->slab_override == 0 && total_bases > default_slab) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + ctx->ws.bytes >= (160ull << 30)) default_slab = 1ll << 31;
    }
    pl->slab_bases_max = ctx->slab_override > 0 ? ctx->slab_override : default_slab;   // multiple of 64
    pl->n_slabs = (total_bases + pl->slab_bases_max - 1) / pl->slab_bases_max;
    const int64_t slab_bases = std::min(total_bases, pl->slab_bases_max);
    // capacities: the key upper bound of one slab (a position range) shared out by the key density with 20 % head
    // room, plus a flat pad of 1/8 of the mean and a constant
    // (Level 1 of a slab in parts on a second stream beside level 2 of the part before -- two sets of level-1 regions -- was an option in
    // rounds 3-5 and always lost, 9.24 ms for one part against 9.49 / 9.87 / 13.1 for 2 / 4 / 8: side by side the two kernels split the CUs'
    // LDS and every part adds a kernel tail.  Removed in round 6.)
    const int64_t max_keys = 3 * slab_bases, max_keys1 = max_keys;
    // (level-1 runs are padded to 4 keys: on average 1.5 pad keys per run of ~48)
    // (level-1 regions hold GROUPS of five keys; a run's last group is partly filled: ~2 pad slots per run of ~72)
    const int64_t mean1 = max_keys1 / kRegions / kGroupKeys * 26 / 25 + 1, mean2 = max_keys / kFine / 2;    // mean1: groups; mean2: pairs of 16-bit keys
    pl->caps1 = DensityCaps{static_cast<uint64_t>(mean1 + mean1 / 5), static_cast<uint32_t>(mean1 / 8 + 1024), 1};   // per level-1 region
    pl->caps2 = DensityCaps{static_cast<uint64_t>(mean2 + mean2 / 5) / 4, static_cast<uint32_t>(mean2 / 8 + 2048) / 4};   // per fine bucket
    if (ctx->bin_cap_override > 0) {                       // test hook: uniform, deliberately small regions
        pl->caps2 = DensityCaps{0, static_cast<uint32_t>((ctx->bin_cap_override + 3) / 4)};
        pl->caps1 = DensityCaps{0, static_cast<uint32_t>((ctx->bin_cap_override + kGroupKeys - 1) / kGroupKeys), 1};
    }
    PALACE_REQUIRE(pl->caps1.cap(0) < (1u << 31) && pl->caps2.cap(0) < (1u << 30), "slab too large for 32-bit region cursors");
    pl->cur1_bytes = align_up(kRegions * sizeof(unsigned int), 256);
    pl->cur2_bytes = align_up(static_cast<size_t>(kFine) * kXcds * sizeof(unsigned int), 256);       // a cursor per fine bucket and XCD
    pl->buf1_bytes = align_up(static_cast<size_t>(pl->caps1.prefix(kL1Buckets)) * kL1Replicas * 16, 256);
    PALACE_REQUIRE(pl->buf1_bytes < (1ull << 36), "slab too large: level 1 addresses its regions as 2^32 groups of 16 bytes");
    pl->buf2_bytes = align_up(static_cast<size_t>(pl->caps2.prefix(kL1Buckets)) * kL2Rows * 4, 256);    // pairs of 2-byte keys
    pl->n_chunks = (total_bases + 63) / 64;
    pl->words_bytes = align_up(static_cast<size_t>(pl->n_chunks + 2) * 8, 256);       // one u64 per 64 positions (+ pad)
    // The invariants the level-1 kernel relies on, stated where the sizes are made (round 2 lost an afternoon's variant to
    // an out-of-range access whose source was not kept -- DESIGN.md section 4 item 6):
    //  (a) a lane reads the u32 stream words g .. g+2 with g <= pos_hi >> 5 <= total_bases >> 5: the pad behind the last
    //      word of a stream must cover two more words, for the last tile of EVERY slab (inner slabs read real words);
    PALACE_REQUIRE(pl->words_bytes >= (static_cast<size_t>(total_bases >> 5) + 3) * 4, "stream pad does not cover the last tile's look-ahead");
    //  (b) destinations are 32-bit indices of 16-byte groups into buf1: base of the last region + its capacity < 2^32 groups;
    PALACE_REQUIRE(pl->caps1.prefix(kL1Buckets) * kL1Replicas < (1ull << 32), "level-1 regions exceed 2^32 groups of 16 bytes");
    //  (c) a region cursor keeps counting when its region is full (the excess takes the exact path): it must not wrap even
    //      if every key of the slab's tiles of one replica lands in one bucket.
    PALACE_REQUIRE(3ull * static_cast<uint64_t>(slab_bases) / kL1Replicas + (1ull << 20) < (1ull << 32), "slab too large for 32-bit region cursors");
    return PALACE_OK;
}
}  // namespace

extern "C" {

int palace_eref_reserve(palace_ctx *ctx, int64_t total_bases)
{
    PALACE_REQUIRE(ctx && total_bases >= 0, "bad argument");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    const bool binned = ctx->probe_all_sets == 2 || ctx->count_mode == 2 || (ctx->count_mode == 0 && total_bases >= (1ll << 22));
    if (!binned) return PALACE_OK;
    CountPlan pl;
    rc = plan_count(ctx, total_bases, &pl);
    if (rc) return rc;
    return ensure_workspace(ctx, pl.total());
}

}  // extern "C"

// The read set as bit streams, once for the whole set: P0, P1, validity; read ends (and dropped reads) -> U.
// ends / dropped: scratch of words_bytes each; strm[q]: (n_chunks + 2) 64-bit words each.
static int build_streams(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_reads, const uint8_t *d_keep,
                         int64_t total_bases, unsigned long long *ends, unsigned long long *dropped, unsigned long long *const strm[3],
                         size_t words_bytes)
{
    const int64_t n_chunks = (total_bases + 63) / 64;
    PALACE_HIP_TRY(hipMemsetAsync(ends, 0, words_bytes, ctx->stream));
    if (d_keep) PALACE_HIP_TRY(hipMemsetAsync(dropped, 0, words_bytes, ctx->stream));
    hipLaunchKernelGGL(mark_read_ends_kernel, dim3(static_cast<unsigned>((n_reads + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_offsets, n_reads, ends);
    if (d_keep)
        hipLaunchKernelGGL(mark_dropped_kernel, dim3(static_cast<unsigned>((n_reads + 255) / 256)), dim3(256), 0,
                           ctx->stream, d_offsets, n_reads, d_keep, dropped);
    PALACE_HIP_TRY(hipGetLastError());
    for (int q = 0; q < 3; q++)                           // the last word of each stream may be partly written, and the
        PALACE_HIP_TRY(hipMemsetAsync(strm[q] + n_chunks - 1, 0, 24, ctx->stream));   // two pad words behind it are read
    const int64_t groups = (total_bases + 15) / 16, blocks = (groups + kStreamTile - 1) / kStreamTile;
    PALACE_REQUIRE(blocks < (1ll << 31), "too many tiles for one launch");
    hipLaunchKernelGGL(eref_streams_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream, d_bases,
                       d_offsets, total_bases, reinterpret_cast<const uint16_t *>(ends),
                       d_keep ? reinterpret_cast<const uint16_t *>(dropped) : nullptr, reinterpret_cast<uint16_t *>(strm[0]),
                       reinterpret_cast<uint16_t *>(strm[1]), reinterpret_cast<uint16_t *>(strm[2]));
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

static void carve_count(const CountPlan &pl, char *ws, bool with_words, CountBufs *b)
{
    b->cursor2 = reinterpret_cast<unsigned int *>(ws); ws += pl.cur2_bytes;
    b->touched = reinterpret_cast<unsigned int *>(ws); ws += kTouchedBytes;
    b->cursor1 = reinterpret_cast<unsigned int *>(ws); ws += pl.cur1_bytes;
    if (with_words) { b->words = reinterpret_cast<unsigned long long *>(ws); ws += 5 * pl.words_bytes; }
    b->buf1 = reinterpret_cast<uint32_t *>(ws); ws += pl.buf1_bytes;
    b->buf2 = reinterpret_cast<uint16_t *>(ws);
}

// the final count kernel of a launch with Phase B's channel-0 probe riding along (eref_lds_count_kernel<true, true, true>)
// (all_sets: eref_lds_count_kernel<true, true, 2> -- every entry set is tested and the ">= 3" plane is not written at all)
static int probe_index_launch_fused(palace_ctx *ctx, const palace_eref_probe_index *ix, const CountBufs &b, const CountPlan &pl, const KeyBuckets &keys,
                                    int mode)                  // 0: channel 0; 1: every entry set, hit bits; 2: every entry set, partial counts
{
    const bool all_sets = mode >= 1;
    // (buckets without keys leave their bytes alone)
    if (mode == 2) {
        PALACE_REQUIRE(ix->ecnt_own[0], "the probe index has no count block (palace_eref_entry_layout / _buffers_attach)");
        PALACE_REQUIRE(ix->canonical, "the probe index's entries are not in position order (a bucket of more than 8192 entries, or no room for the ordering pass): "
                                      "ranks could not add their partial counts entry by entry");
        PALACE_HIP_TRY(hipMemsetAsync(ix->ecnt_own[0], 0, 2 * ix->entry_hits_bytes, ctx->stream));
    } else if (all_sets && ix->ehits_own[0] == ix->hits_block) {
        PALACE_HIP_TRY(hipMemsetAsync(ix->hits_block, 0, ix->ehits_own_bytes, ctx->stream));     // hit bits and sentinel bytes: one block
    } else {
        PALACE_HIP_TRY(hipMemsetAsync(ix->ehits_own[0], 0, all_sets ? ix->entry_hits_bytes : ix->ehits_bytes[0], ctx->stream));
        if (all_sets) PALACE_HIP_TRY(hipMemsetAsync(ix->sent_bytes_own, 0, ix->hit_bytes_size / kSentinelStride, ctx->stream));
    }
    ProbeArgs pr{};
    for (int k = 0; k < kSets; k++) {
        pr.first[k] = ix->first + static_cast<size_t>(k) * (kIndexGroups + 1);
        pr.keys16[k] = ix->keys16[k];
        pr.ehits[k] = mode == 2 ? ix->ecnt_own[k] : ix->ehits_own[k];
    }
    pr.pos_s = ix->pos_s;
    pr.sent_bytes = ix->sent_bytes_own;
    if (mode == 2)
        hipLaunchKernelGGL((eref_lds_count_kernel<true, true, 3>), dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                           ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, pr);
    else if (all_sets)
        hipLaunchKernelGGL((eref_lds_count_kernel<true, true, 2>), dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                           ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, pr);
    else
        hipLaunchKernelGGL((eref_lds_count_kernel<true, true, 1>), dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                           ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, pr);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

// Option probe_all_sets 2 (a rank that counts a share of a sample's reads): a count call has ONE form of result, the partial counts of
// the attached index's entries in its count block, and only the fused count kernel writes it -- so the call must take the binned path
// into a clean table, in one slab, over the whole key space, or fail; what it must never do is succeed some other way and leave the
// block as it was (ADVICE of round 5: the exchange would then add stale counts, and rows come out wrong without an error).
static int partial_counts_ready(palace_ctx *ctx)
{
    const palace_eref_probe_index *ix = ctx->probe_ix;
    if (!ix || !ctx->want_final) { set_error("option probe_all_sets 2 needs option final_count and an attached probe index"); return PALACE_ESTATE; }
    if (!probe_index_usable(ctx, ix)) { set_error("probe_all_sets 2: the attached probe index was built with another coder or holds no entries"); return PALACE_ESTATE; }
    if (!ix->ecnt_own[0]) { set_error("probe_all_sets 2: the probe index has no count block (palace_eref_entry_buffers_attach)"); return PALACE_ESTATE; }
    const bool whole = (ctx->key_buckets[0] & ctx->key_buckets[1] & ctx->key_buckets[2] & ctx->key_buckets[3]) == ~0u;
    if (!whole) { set_error("probe_all_sets 2: a share of the reads is counted over the whole key space (palace_eref_set_key_buckets is set)"); return PALACE_ESTATE; }
    if (!ctx->table_clean || ctx->counts_ptr) { set_error("probe_all_sets 2: one count call per reset (the table is not clean)"); return PALACE_ESTATE; }
    return PALACE_OK;
}
// ... a rank without reads: its counts are zero
static int partial_counts_zero(palace_ctx *ctx)
{
    const palace_eref_probe_index *ix = ctx->probe_ix;
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    PALACE_HIP_TRY(hipMemsetAsync(ix->ecnt_own[0], 0, 2 * ix->entry_hits_bytes, ctx->stream));
    ctx->counts_ptr = ix->ecnt_own[0];
    ctx->c0_hits_ix = nullptr;
    ctx->hits_mask = 0;
    return PALACE_OK;
}

static int launch_bin1(palace_ctx *ctx, hipStream_t stream, int ppl, const uint32_t *w0, const uint32_t *w1, const uint32_t *wu, int64_t p_lo,
                       int64_t p_hi, const BinOut &o1)
{
    const int64_t tile_pos = static_cast<int64_t>(kBinThreads) * ppl;
    const int64_t tiles = (p_hi - p_lo + tile_pos - 1) / tile_pos;
    PALACE_REQUIRE(tiles < (1ll << 31), "too many tiles for one launch");
    // one tile per workgroup, 8 waves.  Measured and dropped: workgroups that take several tiles with the next tile's
    // words in flight -- five variants; the last one (a loader wave with direct-to-LDS loads,
    // LDS-only barriers, hand-placed waits: nothing of the previous tile is waited for) 4.4 ms against 4.07 at the same
    // tile size; 256-thread workgroups (+3 %); 10 / 16 positions per lane (the same / +45 %).
    const dim3 grid(static_cast<unsigned>(tiles)), block(kBinThreads);
#define PALACE_BIN1(P_, SHARE_) hipLaunchKernelGGL((eref_bin1_sort_kernel<P_, kBinThreads, SHARE_>), grid, block, 0, stream, w0, w1, wu, p_lo, p_hi, ctx->masks, o1)
    if (o1.keys.all()) {
        if (ppl == 6) PALACE_BIN1(6, false); else PALACE_BIN1(8, false);
    } else {
        if (ppl == 6) PALACE_BIN1(6, true); else PALACE_BIN1(8, true);
    }
#undef PALACE_BIN1
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}

// The read set as bit streams (P0, P1, U: see eref_streams_kernel) -> level-1 partition -> level-2 partition -> count in LDS,
// slab by slab.  Shared by the ASCII entry (which builds the streams first) and the packed entry (whose caller did).
//
static int bin_and_count(palace_ctx *ctx, const CountPlan &pl, const CountBufs &b, const uint32_t *w0, const uint32_t *w1,
                         const uint32_t *wu, int64_t total_bases, double keys_per_pos)
{
    ctx->c0_hits_ix = nullptr;
    const int64_t kSlabBases = pl.slab_bases_max, n_slabs = pl.n_slabs;
    // positions per lane of the level-1 kernel.  Its throughput is (key slots the CU's LDS holds) / (latency of a tile,
    // ~11 us whatever the tile size): 6 positions x 3 keys x 512 lanes + pads = 39.8 KiB, the most that still fits four
    // times into 160 KiB (5: +4 %, 4: +8 %, 8 -- three workgroups per CU --: +2 %).  Sparse sets (short reads) take 8.
    const int ppl = keys_per_pos > 1.6 ? 6 : 8;
    const KeyBuckets keys = ctx_buckets(ctx);
    Bin2Grid g2;
    g2.first[0] = 0;
    for (uint32_t bk = 0; bk < kL1Buckets; bk++)                                                 // (level 2 only where level 1 wrote)
        g2.first[bk + 1] = g2.first[bk] + (keys.bucket(bk) ? tiles_of_bucket(pl.caps1, bk) * kL1Replicas : 0);
    const Bin2Out o2{b.cursor2, b.buf2, pl.caps2, ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched};
    const BinOut o1{b.cursor1, b.buf1, pl.caps1, ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys};
    for (int64_t slab = 0; slab < n_slabs; slab++) {
        PALACE_HIP_TRY(hipMemsetAsync(b.cursor2, 0, pl.cur2_bytes + kTouchedBytes, ctx->stream));
        PALACE_HIP_TRY(hipMemsetAsync(b.cursor1, 0, pl.cur1_bytes, ctx->stream));
        const bool clean = ctx->table_clean && slab == 0;        // every plane bit is still zero: slices need no reading
        const int64_t s_lo = slab * kSlabBases, s_hi = std::min(total_bases, (slab + 1) * kSlabBases);
        int rc1 = launch_bin1(ctx, ctx->stream, ppl, w0, w1, wu, s_lo, s_hi, o1);
        if (rc1) return rc1;
        hipLaunchKernelGGL(eref_bin2_kernel, dim3(g2.first[kL1Buckets]), dim3(kBin2Threads), 0, ctx->stream, b.cursor1, b.buf1, pl.caps1, g2, o2);
        PALACE_HIP_TRY(hipGetLastError());
        const ProbeArgs no_probe{};
        if (clean && n_slabs == 1 && ctx->want_final) {
            const palace_eref_probe_index *ix = ctx->probe_ix;
            const bool whole = (ctx->key_buckets[0] & ctx->key_buckets[1] & ctx->key_buckets[2] & ctx->key_buckets[3]) == ~0u;
            if (ix && whole && probe_index_usable(ctx, ix)) {
                // the final count of a whole key space with a probe index attached: channel 0 of Phase B rides along -- or (option
                // probe_all_sets) all of Phase B's look-ups, and the ">= 3" plane is never written
                int rc = probe_index_launch_fused(ctx, ix, b, pl, keys, ctx->probe_all_sets);
                if (rc) return rc;
                if (ctx->probe_all_sets == 2) {              // partial counts: nothing to scan from until the ranks' counts are summed
                    ctx->c0_hits_ix = nullptr;
                    ctx->hits_mask = 0;
                    ctx->counts_ptr = ix->ecnt_own[0];
                } else {
                    ctx->c0_hits_ix = ix;
                    ctx->hits_mask = ctx->probe_all_sets ? (1u << kSets) - 1 : 1u;
                }
                ctx->sent_scattered = ctx->probe_all_sets == 1;
                ctx->planeless = ctx->probe_all_sets != 0;
            } else {
                if (ctx->probe_all_sets == 2) { set_error("probe_all_sets 2: the count could not be fused with the probe index"); return PALACE_ESTATE; }
                hipLaunchKernelGGL((eref_lds_count_kernel<true, true>), dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                                   ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, no_probe);
            }
            ctx->final_only = true;
        } else if (clean)
            hipLaunchKernelGGL(eref_lds_count_kernel<true>, dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                               ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, no_probe);
        else
            hipLaunchKernelGGL(eref_lds_count_kernel<false>, dim3(kFine), dim3(kCountThreads), 0, ctx->stream, b.cursor2, b.buf2, pl.caps2,
                               ctx->plane[0], ctx->plane[1], ctx->plane[2], b.touched, keys, no_probe);
        PALACE_HIP_TRY(hipGetLastError());
        ctx->table_clean = false;
    }
    return PALACE_OK;
}

extern "C" {

int palace_eref_count_reads(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets,
                            int64_t n_reads, const uint8_t *d_keep, int64_t total_bases)
{
    PALACE_REQUIRE(ctx && n_reads >= 0, "bad argument");
    if (!ctx->coder_set) { set_error("palace_eref_count_reads: coder not set"); return PALACE_ESTATE; }
    const bool partial = ctx->probe_all_sets == 2;       // the call leaves partial entry counts or fails (partial_counts_ready)
    if (partial) { int rc0 = partial_counts_ready(ctx); if (rc0) return rc0; }
    if (n_reads == 0) return partial ? partial_counts_zero(ctx) : PALACE_OK;
    PALACE_REQUIRE(d_bases && d_offsets, "null device pointer");
    PALACE_REQUIRE(!ctx->final_only, "the table holds only its \">= 3\" plane (option final_count): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    // total bases bound the number of keys; tiny inputs keep the direct path (a 16 Ki-workgroup launch
    // per call would dominate them), everything else is binned
    if (total_bases < 0) {                                  // caller does not know: read the two end offsets back
        int64_t h_off[2];
        PALACE_HIP_TRY(hipMemcpyAsync(&h_off[0], d_offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipMemcpyAsync(&h_off[1], d_offsets + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
        total_bases = h_off[1] - h_off[0];
        PALACE_REQUIRE(total_bases >= 0, "offsets not ascending");
    }
    if (ctx->keys_counted >= 0) ctx->keys_counted += 3 * total_bases;
    const bool binned = partial || ctx->count_mode == 2 || (ctx->count_mode == 0 && total_bases >= (1ll << 22));
    if (!binned) {
        int64_t blocks = (n_reads + 3) / 4;                 // 4 waves (reads) per 256-thread block
        int64_t cap = static_cast<int64_t>(kCUs) * 8 * 8;   // grid-stride beyond 16 Ki blocks
        if (blocks > cap) blocks = cap;
        ctx->c0_hits_ix = nullptr;
        hipLaunchKernelGGL(eref_count_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream,
                           d_bases, d_offsets, n_reads, d_keep, ctx->masks, ctx_buckets(ctx), ctx->plane[0], ctx->plane[1],
                           ctx->plane[2]);
        PALACE_HIP_TRY(hipGetLastError());
        ctx->table_clean = false;
        return PALACE_OK;
    }
    CountPlan pl;
    rc = plan_count(ctx, total_bases, &pl);
    if (rc) return rc;
    if (partial && pl.n_slabs != 1) { set_error("probe_all_sets 2: the read share does not fit one slab (%lld positions)", static_cast<long long>(total_bases)); return PALACE_ESTATE; }
    const size_t words_bytes = pl.words_bytes;
    rc = ensure_workspace(ctx, pl.total());
    if (rc) return rc;
    CountBufs cb;
    carve_count(pl, static_cast<char *>(ctx->ws.ptr), true, &cb);
    unsigned long long *ends = cb.words, *dropped = cb.words + words_bytes / 8;
    unsigned long long *strm[3];                           // P0, P1, U
    for (int q = 0; q < 3; q++) strm[q] = cb.words + (2 + q) * (words_bytes / 8);
    rc = build_streams(ctx, d_bases, d_offsets, n_reads, d_keep, total_bases, ends, dropped, strm, words_bytes);
    if (rc) return rc;
    const double keys_per_pos = 3.0 * std::max(0.02, 1.0 - 31.0 * static_cast<double>(n_reads) / std::max<double>(1.0, static_cast<double>(total_bases)));
    return bin_and_count(ctx, pl, cb, reinterpret_cast<const uint32_t *>(strm[0]), reinterpret_cast<const uint32_t *>(strm[1]),
                         reinterpret_cast<const uint32_t *>(strm[2]), total_bases, keys_per_pos);
}

/* E4, packed input (include/palace_hip.h): the three bit streams come from the caller, the partition kernels read them where
 * they lie -- no stream kernel, no read-end marks, no ASCII in HBM. */
size_t palace_eref_packed_bytes(int64_t n_positions)
{
    return n_positions < 0 ? 0 : (static_cast<size_t>((n_positions + 63) / 64) + 2) * 8;
}

int palace_eref_pack_reads(palace_ctx *ctx, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_reads, const uint8_t *d_keep,
                           int64_t total_bases, uint32_t *d_p0, uint32_t *d_p1, uint32_t *d_u)
{
    PALACE_REQUIRE(ctx && n_reads >= 0 && total_bases >= 0, "bad argument");
    if (n_reads == 0 || total_bases == 0) return PALACE_OK;
    PALACE_REQUIRE(d_bases && d_offsets && d_p0 && d_p1 && d_u, "null device pointer");
    PALACE_REQUIRE(((reinterpret_cast<uintptr_t>(d_p0) | reinterpret_cast<uintptr_t>(d_p1) | reinterpret_cast<uintptr_t>(d_u)) & 7) == 0,
                   "the streams must be 8-byte aligned");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const size_t words_bytes = align_up(palace_eref_packed_bytes(total_bases), 256);
    int rc = ensure_workspace(ctx, 2 * words_bytes);
    if (rc) return rc;
    unsigned long long *ends = static_cast<unsigned long long *>(ctx->ws.ptr), *dropped = ends + words_bytes / 8;
    unsigned long long *const strm[3] = {reinterpret_cast<unsigned long long *>(d_p0), reinterpret_cast<unsigned long long *>(d_p1),
                                         reinterpret_cast<unsigned long long *>(d_u)};
    return build_streams(ctx, d_bases, d_offsets, n_reads, d_keep, total_bases, ends, dropped, strm, words_bytes);
}

int palace_eref_count_reads_packed(palace_ctx *ctx, const uint32_t *d_p0, const uint32_t *d_p1, const uint32_t *d_u,
                                   int64_t n_positions, int64_t n_reads_hint)
{
    PALACE_REQUIRE(ctx && n_positions >= 0 && n_reads_hint >= 0, "bad argument");
    if (!ctx->coder_set) { set_error("palace_eref_count_reads_packed: coder not set"); return PALACE_ESTATE; }
    const bool partial = ctx->probe_all_sets == 2;       // the call leaves partial entry counts or fails (partial_counts_ready)
    if (partial) { int rc0 = partial_counts_ready(ctx); if (rc0) return rc0; }
    if (n_positions == 0) return partial ? partial_counts_zero(ctx) : PALACE_OK;
    PALACE_REQUIRE(d_p0 && d_p1 && d_u, "null device pointer");
    PALACE_REQUIRE(((reinterpret_cast<uintptr_t>(d_p0) | reinterpret_cast<uintptr_t>(d_p1) | reinterpret_cast<uintptr_t>(d_u)) & 7) == 0,
                   "the streams must be 8-byte aligned");
    PALACE_REQUIRE(!ctx->final_only, "the table holds only its \">= 3\" plane (option final_count): reset it first");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    if (ctx->keys_counted >= 0) ctx->keys_counted += 3 * n_positions;
    const bool binned = partial || ctx->count_mode == 2 || (ctx->count_mode == 0 && n_positions >= (1ll << 22));
    if (!binned) {
        const int64_t blocks = std::min<int64_t>((n_positions + 255) / 256, static_cast<int64_t>(kCUs) * 8 * 8);
        ctx->c0_hits_ix = nullptr;
        hipLaunchKernelGGL(eref_count_packed_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream, d_p0, d_p1, d_u,
                           n_positions, ctx->masks, ctx_buckets(ctx), ctx->plane[0], ctx->plane[1], ctx->plane[2]);
        PALACE_HIP_TRY(hipGetLastError());
        ctx->table_clean = false;
        return PALACE_OK;
    }
    CountPlan pl;
    rc = plan_count(ctx, n_positions, &pl);
    if (rc) return rc;
    if (partial && pl.n_slabs != 1) { set_error("probe_all_sets 2: the read share does not fit one slab (%lld positions)", static_cast<long long>(n_positions)); return PALACE_ESTATE; }
    // (palace_eref_packed_bytes covers what plan_count checks the look-ahead of the last tile against: (n >> 5) + 3 words of 4 bytes)
    rc = ensure_workspace(ctx, pl.total() - 5 * pl.words_bytes);
    if (rc) return rc;
    CountBufs cb;
    carve_count(pl, static_cast<char *>(ctx->ws.ptr), false, &cb);
    const double keys_per_pos = n_reads_hint ? 3.0 * std::max(0.02, 1.0 - 31.0 * static_cast<double>(n_reads_hint) / static_cast<double>(n_positions)) : 3.0;
    return bin_and_count(ctx, pl, cb, d_p0, d_p1, d_u, n_positions, keys_per_pos);
}

int palace_eref_set_key_buckets(palace_ctx *ctx, const uint32_t mask128[4])
{
    PALACE_REQUIRE(ctx && mask128, "null argument");
    PALACE_REQUIRE((mask128[0] | mask128[1] | mask128[2] | mask128[3]) != 0, "the set of key buckets is empty");
    for (int i = 0; i < 4; i++) ctx->key_buckets[i] = mask128[i];
    return PALACE_OK;
}

/* Tuning knobs of count_reads (see include/palace_hip.h). */
int palace_eref_set_count_mode(palace_ctx *ctx, int mode, int64_t bucket_cap)
{
    PALACE_REQUIRE(ctx && mode >= 0 && mode <= 2 && bucket_cap >= 0 && bucket_cap < (1ll << 31), "bad argument");
    ctx->count_mode = mode;
    ctx->bin_cap_override = bucket_cap;
    return PALACE_OK;
}

int palace_eref_set_option(palace_ctx *ctx, const char *name, int64_t value)
{
    PALACE_REQUIRE(ctx && name, "null argument");
    if (!std::strcmp(name, "slab_bases")) {
        PALACE_REQUIRE(value >= 0 && value % 64 == 0, "slab size must be a non-negative multiple of 64");
        ctx->slab_override = value;
    } else if (!std::strcmp(name, "final_count")) {          // the count calls that follow are each the only one between a reset and Phase B
        PALACE_REQUIRE(value == 0 || value == 1, "final_count must be 0 or 1");
        ctx->want_final = value != 0;
    } else if (!std::strcmp(name, "probe_all_sets")) {       // with final_count and an attached probe index: the final count tests ALL of the
        PALACE_REQUIRE(value >= 0 && value <= 2, "probe_all_sets must be 0, 1 or 2");   // index's entry sets and writes no plane (see ProbeArgs);
        ctx->probe_all_sets = static_cast<int>(value);                                   // 2: leaves partial counts (a share of the reads)
    } else if (!std::strcmp(name, "scan_ref_lo") || !std::strcmp(name, "scan_ref_hi")) {   // palace_eref_scan_refs_indexed works on refs [lo, hi) only
        PALACE_REQUIRE(value >= 0, "a ref ordinal");                                       // (hi = 0: all); rows of other refs: n_intervals = el = 0
        (name[9] == 'l' ? ctx->scan_ref_lo : ctx->scan_ref_hi) = value;
    } else {
        set_error("palace_eref_set_option: unknown option '%s'", name);
        return PALACE_EINVAL;
    }
    return PALACE_OK;
}

}  // extern "C"

#ifdef PALACE_STAMPS
__device__ unsigned long long palace_stamp_buf[8 * 65536];
extern "C" int palace_debug_stamps(palace_ctx *ctx, unsigned long long *h_out, int64_t n_words)
{
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PALACE_HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(palace_stamp_buf), static_cast<size_t>(n_words) * 8));
    return PALACE_OK;
}
#endif
