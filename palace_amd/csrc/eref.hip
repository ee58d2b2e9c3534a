// eref on gfx950: k-mer screening of reads against the phage DB.
// Functional spec: bin/extract_ref.cpp of the reference (rows E1-E6 of SURVEY.md section 8).
//
// Design (MI355X-first, not a translation):
//  * A sequence is turned into three projection bit-streams + one validity stream with wave-wide
//    ballots: lane l of a 64-lane wave classifies base l, __ballot() delivers 64 bases of one
//    projection as one 64-bit scalar.  A 32-mer at offset j is then the 32-bit window
//    w_q = stream_q >> j (bit t = base j+t), obtained with one 64-bit funnel shift per stream.
//  * The reference's 32-step loop per (position, channel) collapses to mask algebra.  Channel i
//    reads projection cc[3z+i] at k-mer offset z with weight 2^(31-z) (extract_ref.cpp:717-725).
//    With M[i][q] = { bit t : cc[3(31-t)+i] == q } the forward index is
//        fwd_i = OR_q ( brev(w_q) & M[i][q] )
//    and, because the complement leaves projection 0 unchanged and inverts projections 1 and 2
//    (A<->T, C<->G; extract_ref.cpp:1012-1051, 1071-1078), the reverse-complement index is
//        rc_i  = (w_0 & M[i][0]) | (~w_1 & M[i][1]) | (~w_2 & M[i][2]).
//    canonical = min(fwd, rc) (extract_ref.cpp:727-732, 989-994).
//  * The 4 GiB saturating byte table (extract_ref.cpp:25-26, 995-996) becomes three 512 MiB bit
//    planes "count>=1", ">=2", ">=3".  An occurrence does atomicOr on plane 1 and climbs to the
//    next plane only if the bit was already set, so n occurrences set exactly min(n,3) planes in
//    any interleaving: the result equals the reference's threads=1 table, with no CAS loop.
//    Phase B only ever asks "count == 3" (extract_ref.cpp:531), i.e. it reads plane 3 alone.
//  * Phase B recomputes the ref-side indices from the ref bases (1 B/base) instead of streaming
//    the 12 B/position index file, writes 2 bits per position (any-channel / all-channel hit),
//    and does the 500-base window test with prefix population counts.
#include "common.hpp"

#ifdef PALACE_STAMPS        // diagnostic build (tools/dbg/stamps.py): per-workgroup phase stamps of the partition kernels
__device__ unsigned long long palace_stamp_buf[8 * 65536];
#define STAMP(arr, i) do { if (arr) (arr)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(arr, i) do { } while (0)      // (the argument is not even named in product builds)
#endif


namespace palace {

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
struct BaseBits {
    bool p0, p1, p2, ok;
};

// One base -> its three projection bits and validity.  Upper-cased ASCII: A 0x41, C 0x43, G 0x47,
// T 0x54; (x >> 1) & 3 maps A,C,T,G to 0,1,2,3, so {A,T} = !(c & 1), {A,C} = !(c & 2), {A,G} = bits equal.
__device__ __forceinline__ BaseBits classify(uint32_t ch)
{
    const uint32_t x = ch & 0xDFu;                 // fold case (only letters can land on A/C/G/T)
    const uint32_t d = x - 0x41u;                  // A,C,G,T -> 0,2,6,19
    const bool ok = d < 20u && ((0x80045u >> d) & 1u);
    const uint32_t c = x >> 1;
    return BaseBits{!(c & 1u), !(c & 2u), !((c ^ (c >> 1)) & 1u), ok};
}

struct Streams {
    uint64_t p0, p1, p2, ok;
};

__device__ __forceinline__ Streams ballot_streams(const uint8_t *__restrict__ s, int64_t idx, int64_t len)
{
    uint32_t ch = (idx < len) ? s[idx] : 0u;
    BaseBits b = classify(ch);
    return Streams{__ballot(b.p0), __ballot(b.p1), __ballot(b.p2), __ballot(b.ok)};   // p* of invalid bases are never used
}

// bits [lane, lane+31] of the 128-bit value hi:lo, for lane in 0..63.  lo and hi are wave-uniform
// (ballots, SGPR pairs): two 64-bit shifts with the scalar pair as source and one OR (selecting the
// two 32-bit words per lane for a v_alignbit costs three SGPR->VGPR moves and two selects more).
__device__ __forceinline__ uint32_t window32(uint64_t lo, uint64_t hi, int lane)
{
    return static_cast<uint32_t>(lo >> lane) | static_cast<uint32_t>((hi << 1) << (63 - lane));
}

__device__ __forceinline__ uint32_t canonical(const CoderMasks &m, int i, uint32_t w0, uint32_t w1,
                                              uint32_t w2, uint32_t f0, uint32_t f1, uint32_t f2)
{
    // the three masks of a channel partition the 32 bits (set_coder checks the header for that), so each
    // index is two bit-selects (v_bfi / v_bitop3) instead of three ANDs and two ORs
    // v_bitop3_b32 truth tables: 0xCA = a ? b : c (bit select), 0xC5 = a ? b : ~c
    const uint32_t m0 = m.m[i][0], m1 = m.m[i][1];
    const uint32_t fwd = __builtin_amdgcn_bitop3_b32(m0, f0, __builtin_amdgcn_bitop3_b32(m1, f1, f2, 0xCA), 0xCA);
    const uint32_t rc = __builtin_amdgcn_bitop3_b32(m0, w0, __builtin_amdgcn_bitop3_b32(m1, w1, w2, 0xCA), 0xC5);
    return fwd < rc ? fwd : rc;
}

// Three canonical indices of the 32-mer whose projection windows are w0..w2.
__device__ __forceinline__ void kmer_keys(const CoderMasks &m, uint32_t w0, uint32_t w1, uint32_t w2,
                                          uint32_t key[3])
{
    uint32_t f0 = __brev(w0), f1 = __brev(w1), f2 = __brev(w2);
#pragma unroll
    for (int i = 0; i < 3; i++) key[i] = canonical(m, i, w0, w1, w2, f0, f1, f2);
}

// ------------------------------------------------------------------------------------------
// E4: count reads -- one wave per read, grid-stride over reads
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void count_key(uint32_t key, uint32_t *__restrict__ p1,
                                          uint32_t *__restrict__ p2, uint32_t *__restrict__ p3)
{
    uint32_t word = key >> 5, bit = 1u << (key & 31);
    if (atomicOr(&p1[word], bit) & bit)
        if (atomicOr(&p2[word], bit) & bit) atomicOr(&p3[word], bit);
}

// Option key buckets (multi-GPU: every rank counts ALL reads but only the keys of its share of the key space, the ">= 3" plane
// is gathered): keys whose top 7 bits -- the level-1 bucket -- are not in the set are dropped where they are made.  A set, not
// a range: the key density falls linearly over the key space (DensityCaps), so equal shares pair a dense bucket with a sparse one.
struct KeyBuckets {
    uint32_t m[4];                                    // bit b: level-1 bucket b is counted; all ones: the whole space
    __host__ __device__ __forceinline__ bool bucket(uint32_t b) const        // (two 64-bit words: one select, one shift; b < 128)
    {
        const unsigned long long lo = m[0] | (static_cast<unsigned long long>(m[1]) << 32), hi = m[2] | (static_cast<unsigned long long>(m[3]) << 32);
        return (((b & 64u) ? hi : lo) >> (b & 63u)) & 1ull;
    }
    __host__ __device__ __forceinline__ bool all() const { return (m[0] & m[1] & m[2] & m[3]) == ~0u; }
    __device__ __forceinline__ bool has(uint32_t key) const { return bucket(key >> 25); }
};

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


// ------------------------------------------------------------------------------------------
// E4, partitioned path: no global atomics on the table.  The 2.4e9 keys of a gigabase of reads are radix-partitioned in
// two levels into 2^16 fine buckets (key >> 16), then one workgroup per fine bucket counts in LDS.
//   stream kernel  bases -> packed bit streams P0, P1, P2 (projections) and U (a 32-mer may start here)
//   level 1        a workgroup owns a tile of positions: keys from the bit streams (one v_alignbit per window), sorted by
//                  their top 7 bits in LDS (counting sort), every bucket's run written to that bucket's region with
//                  16-byte stores; one global atomicAdd per bucket and tile reserves the run
//   level 2        the same for a tile of one level-1 region, on key bits 24..16 (512 fine rows), writing only the low
//                  16 bits of every key: below this level a key costs 2 bytes
//   count kernel   one workgroup per fine bucket: its 2^16-key slice of the three planes (3 x 8 KiB) lives in LDS,
//                  takes the bucket's keys with LDS atomicOr climbing 1 -> 2 -> 3, and is written back
// The canonical index is min(forward, reverse complement) of two hash-like 32-bit words, so for ANY input its density
// over the key space is 2(1-x): level-1 bucket 0 receives twice the mean, bucket 127 almost nothing.  Level-1 regions
// and fine-bucket regions are therefore sized by that density (a constant pad plus a share proportional to 255-2b).
// A key that finds its region (or, in level 2, its staging row) full goes straight to the planes with global atomics,
// so the result stays exact for any input.
// Traffic per key: 3.2 B written + 3.2 B read (level-1 records of 25 bits, five to a 16-byte group) + 2 B written + 2 B read,
// instead of ~52 B of memory-side atomic requests.
// ------------------------------------------------------------------------------------------
constexpr int kBucketBits = 14;                       // probe index of Phase B: 2^14 groups of 2^18 keys
constexpr int kBuckets = 1 << kBucketBits;
constexpr int kBucketShift = 32 - kBucketBits;
constexpr int kSliceWords = 1 << (kBucketShift - 5);  // 8192 u32 of a plane per probe group
constexpr int kL1Buckets = 128, kL1Shift = 25;        // level 1: key >> 25
constexpr int kL1Replicas = 32;                       // level-1 bucket regions are split 32 ways so that the per-tile
                                                      // reservations do not pile onto 128 addresses (8 ... 64: no difference)
constexpr int kBinThreads = 512;                      // level 1: 8 waves, <= 39.5 KiB of LDS -> 4 workgroups per CU
constexpr int kRowSlots = 72;                         // level 2: slots of a staging row (mean fill 48: +3.5 sigma)

// Capacity of the slot range that belongs to level-1 bucket b when a total is shared out by the key
// density: prefix(b) = pad*b + share*b*(256-b)/128, capacity(b) = prefix(b+1) - prefix(b)
//        = pad + share*(255-2b)/128 (up to rounding); prefix(128) = 128*(pad + share).
struct DensityCaps {
    uint64_t share;      // mean capacity handed out by density, in units of `unit` keys
    uint32_t pad;        // flat capacity every bucket gets, in units of `unit` keys
    uint32_t unit = 4;   // capacities and region starts are multiples of this many keys (4 keys = 16 bytes)
    __host__ __device__ uint64_t prefix(uint32_t b) const { return unit * (static_cast<uint64_t>(pad) * b + ((share * (b * (256u - b))) >> 7)); }
    __host__ __device__ uint32_t cap(uint32_t b) const { return static_cast<uint32_t>(prefix(b + 1) - prefix(b)); }
};

// the exact slow path of the partition kernels: the key goes straight to the planes, and its fine bucket is marked so
// that the count kernel knows this slice of the planes is not what it was when the launch began
__device__ __forceinline__ void count_key_marked(uint32_t key, uint32_t *__restrict__ p1, uint32_t *__restrict__ p2,
                                                 uint32_t *__restrict__ p3, unsigned int *__restrict__ touched)
{
    atomicOr(&touched[key >> 21], 1u << ((key >> 16) & 31));
    count_key(key, p1, p2, p3);
}

struct BinOut {
    unsigned int *cursor;          // per destination region: keys reserved so far
    uint32_t *buf;                 // destination regions, laid out by `caps`
    DensityCaps caps;              // capacity of a destination region of level-1 bucket b
    uint32_t *p1, *p2, *p3;        // overflow path
    unsigned int *touched;         // one bit per fine bucket: the overflow path wrote into its plane slices
    KeyBuckets keys;               // level-1 buckets this call counts
};

// Level-1 cursors are laid out replica-major: the 128 reservations of a tile (one per bucket, lanes 0..127) fall into
// 512 consecutive bytes instead of 128 different cache lines.
__host__ __device__ constexpr uint32_t l1_cursor(uint32_t b, uint32_t replica) { return replica * kL1Buckets + b; }
// Level-1 regions: the kL1Replicas regions of bucket b lie side by side, buckets in order.
__device__ __forceinline__ uint64_t l1_region_base(const DensityCaps &c, uint32_t b, uint32_t replica)
{
    return c.prefix(b) * kL1Replicas + static_cast<uint64_t>(replica) * c.cap(b);
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

// level 2: a workgroup takes a tile of kTile2Groups groups of one level-1 region (bucket b1, replica).  The 25 low
// bits of a key split into a fine row (bits 24..16: 512 rows) and a 16-bit payload, and only the payload is staged and
// written: below this level a key costs 2 bytes, not 4.  The grid covers the largest region's capacity, so most
// workgroups of the sparser buckets leave at once (a device-built list of the non-empty tiles was measured and cost
// more than it saved).
constexpr int kFineBits = 16;                         // fine bucket = key >> 16: 65536 slices of 2^16 keys
constexpr int kFine = 1 << kFineBits;
constexpr int kL2Rows = kFine / kL1Buckets;           // 512 fine rows per level-1 bucket
constexpr int kBin2Threads = 1024;                    // 16 waves; 78 KiB of LDS -> 2 workgroups per CU
constexpr int kGroups2PerThread = 5;
constexpr int kTile2Groups = kBin2Threads * kGroups2PerThread;   // 5120 groups <= 25600 keys (runs fill their groups to ~96 %): row mean 48 of 72 slots
constexpr int kStage2Slots = kL2Rows * kRowSlots;

struct Stage2 {
    uint16_t slot[kStage2Slots + 2];                  // 72 KiB; [kStage2Slots]: where the appends that are none land
    uint32_t rows[kL2Rows];                           // next free slot of the row (row r owns slots [r * kRowSlots, (r + 1) * kRowSlots))
};

struct Bin2Out {
    unsigned int *cursor;          // per fine bucket: keys reserved so far
    uint16_t *buf;                 // fine-bucket regions (16-bit payloads)
    DensityCaps caps;              // capacity of a fine region of level-1 bucket b1, in PAIRS of keys
    uint32_t *p1, *p2, *p3;        // overflow path
    unsigned int *touched;
};

// Fine-bucket regions: the 512 fine buckets of level-1 bucket b1 lie side by side, equal capacity; capacities are
// counted in pairs of 16-bit keys, so regions start on 16-byte boundaries (caps are multiples of 4 pairs).
__device__ __forceinline__ uint64_t fine_region_base(const DensityCaps &c, uint32_t b1, uint32_t sub)
{
    return 2 * (c.prefix(b1) * kL2Rows + static_cast<uint64_t>(sub) * c.cap(b1));
}

// A fine region is split eight ways, one sub-region per XCD: a workgroup appends its runs (~48 payloads, 2-byte granular)
// to the sub-region of the XCD it runs on (read from the hardware, HW_REG_XCC_ID; any value 0..7 is correct).  Measured:
// the kernel 4.09 -> 3.73 ms (eight times as many cursors share the reservations, and a sub-region's run ends meet in one
// L2); the bytes written do NOT drop (6.0 -> 6.6 GB for 4.76 GB of payloads): the memory side writes 64-byte granules, and a
// ~96-byte run at a 2-byte offset touches 2.4 of them wherever its neighbours come from.  Runs padded to whole 16-byte pieces
// (pad value 0xffff, keys with that payload counted in a side array) were built, parity-green, and dropped: 6.9 GB written,
// the same 3.75 ms, the count kernel 1.32 -> 1.50 ms for the pads it skips.  Only longer runs would help, i.e. more LDS.
constexpr int kXcds = 8;
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
__host__ __device__ inline uint32_t fine_sub_cap(const DensityCaps &c, uint32_t b1) { return (2 * c.cap(b1) / kXcds) & ~7u; }   // keys; multiple of 8

typedef uint16_t __attribute__((address_space(1))) global_u16;

// tiles of kTile2Groups groups that cover the capacity of one region of level-1 bucket b
__host__ __device__ inline uint32_t tiles_of_bucket(const DensityCaps &c, uint32_t b) { return (c.cap(b) + kTile2Groups - 1) / kTile2Groups; }

struct Bin2Grid { uint32_t first[kL1Buckets + 1]; };   // first[b] = workgroups in front of bucket b (tiles x replicas, prefix)

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
constexpr int kFineWords = kFine / 32;               // 2048 u32 per plane per fine bucket
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
__device__ __forceinline__ uint32_t probe_vector(const uint32_t *__restrict__ l3, const uint4 &v)
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t k0 = d[e] & 0xffffu, k1 = d[e] >> 16;
        m |= ((l3[k0 >> 5] >> (k0 & 31)) & 1u) << (2 * e);
        m |= ((l3[k1 >> 5] >> (k1 & 31)) & 1u) << (2 * e + 1);
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

// ------------------------------------------------------------------------------------------
// tiling of a set of sequences: tile = kTileChunks x 64 positions of one sequence
// ------------------------------------------------------------------------------------------
constexpr int kTileChunks = 32;                 // 2048 positions per 256-thread block
constexpr int kTilePos = kTileChunks * 64;

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

__device__ __forceinline__ int64_t find_seq(const int64_t *__restrict__ pre, int64_t n, int64_t tile)
{
    int64_t lo = 0, hi = n;            // largest r with pre[r] <= tile
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (pre[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
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
constexpr int kIndexGroups = 1 << 16, kGroupsPerProbe = kIndexGroups / kBuckets;
constexpr int kSets = 4, kSentinelSet = 3, kSentinelStride = 4;
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

// host: E1 masks from the header (extract_ref.cpp:1104-1122 for the header layout)
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

static int masks_from_header(const uint8_t *hdr, CoderMasks *out)
{
    std::memset(out, 0, sizeof *out);
    for (int z = 0; z < 32; z++) {
        int seen = 0;
        for (int i = 0; i < 3; i++) {
            int q = static_cast<int16_t>(hdr[4 * (3 * z + i)] | (hdr[4 * (3 * z + i) + 1] << 8));
            if (q < 0 || q > 2) return -1;
            seen |= 1 << q;
            out->m[i][q] |= 1u << (31 - z);
        }
        if (seen != 7) return -1;                  // each position must hold a permutation of 0,1,2
    }
    return 0;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

namespace {
// Workspace of one count_reads call over `total_bases` positions: slab size, region capacities, byte counts.
constexpr size_t kTouchedBytes = 8192 + 256;             // one bit per fine bucket (2^16 bits), padded
struct CountPlan {
    int64_t slab_bases_max = 0, n_slabs = 0, n_chunks = 0;
    DensityCaps caps1{}, caps2{};
    size_t cur1_bytes = 0, cur2_bytes = 0, buf1_bytes = 0, buf2_bytes = 0, words_bytes = 0;
    size_t total() const { return cur1_bytes + buf1_bytes + cur2_bytes + kTouchedBytes + 5 * words_bytes + buf2_bytes; }
};
constexpr int64_t kRegions = static_cast<int64_t>(kL1Buckets) * kL1Replicas;

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

static KeyBuckets ctx_buckets(const palace_ctx *ctx)
{
    KeyBuckets k;
    for (int i = 0; i < 4; i++) k.m[i] = ctx->key_buckets[i];
    return k;
}

// Scratch of one count call, carved out of the context's workspace (sizes: CountPlan).
struct CountBufs {
    unsigned int *cursor2 = nullptr, *touched = nullptr, *cursor1 = nullptr;
    unsigned long long *words = nullptr;                 // 5 x words_bytes (ASCII entry: read ends, dropped, three streams) or nothing
    uint32_t *buf1 = nullptr;
    uint16_t *buf2 = nullptr;
};
static void carve_count(const CountPlan &pl, char *ws, bool with_words, CountBufs *b)
{
    b->cursor2 = reinterpret_cast<unsigned int *>(ws); ws += pl.cur2_bytes;
    b->touched = reinterpret_cast<unsigned int *>(ws); ws += kTouchedBytes;
    b->cursor1 = reinterpret_cast<unsigned int *>(ws); ws += pl.cur1_bytes;
    if (with_words) { b->words = reinterpret_cast<unsigned long long *>(ws); ws += 5 * pl.words_bytes; }
    b->buf1 = reinterpret_cast<uint32_t *>(ws); ws += pl.buf1_bytes;
    b->buf2 = reinterpret_cast<uint16_t *>(ws);
}

struct palace_eref_probe_index {
    int64_t n_refs = 0, total_bases = 0;
    palace::CoderMasks masks{};               // the coder the indices were computed with
    // entry sets 0..2 = channels 0..2 of every valid position, 3 = the sentinels (channel 0 at positions = 0 mod 4)
    unsigned long long n_entries[palace::kSets] = {0, 0, 0, 0};      // incl. the pads that bring every fine bucket's start to a multiple of 8
    unsigned long long *first = nullptr;      // [kSets][kIndexGroups + 1]: entries grouped by index >> 16 (the count kernel's fine buckets;
                                              //  four consecutive groups are one 2^18-key group of the probe kernel)
    uint16_t *keys16[palace::kSets] = {nullptr, nullptr, nullptr, nullptr};       // [n_entries rounded up to 128 (+ 8)] index & 0xffff (pads: 0)
    uint32_t *eix[3] = {nullptr, nullptr, nullptr};                    // [hit_bytes_size] position id -> entry of channel c (~0: none)
    uint32_t *pos_s = nullptr;                // sentinel entry -> position id / 4 (pads and the tail: ~0)
    uint8_t *ehits_own[palace::kSets] = {nullptr, nullptr, nullptr, nullptr};    // the sets' hit bits (one per entry) when a count launch this index is
                                              //  attached to leaves them: channel 0's, or (option probe_all_sets) all four; ONE allocation, [0] heads it
    size_t ehits_own_bytes = 0;
    uint8_t *sent_bytes_own = nullptr;        // ... and, behind them in the same block, the byte per sentinel in position order that launch sets for the hits
    // (option probe_all_sets 2) the sets' partial COUNTS, 16 bits per vector of eight entries: same layout as the hit bits, twice the bytes
    uint8_t *ecnt_own[palace::kSets] = {nullptr, nullptr, nullptr, nullptr};
    size_t entry_hits_bytes = 0;              // bytes of the four sets' hit-bit parts together (without the sentinel bytes); counts: twice that
    size_t set_at[palace::kSets] = {0, 0, 0, 0};    // where a set's part starts in the hit-bit block
    uint8_t *hits_block = nullptr, *counts_block = nullptr;       // the index's own allocations (the pointers above may be re-pointed at a caller's)
    bool canonical = false;                   // every bucket's entries are in position order: two builds of one DB are the same index (eref_probe_index_canon_kernel)
    size_t ehits_bytes[palace::kSets] = {0, 0, 0, 0};                  // bytes of a set's hit bits (multiple of 16; the tail stays zero)
    size_t hit_bytes_size = 0;                // position ids run over [0, hit_bytes_size)
};

constexpr size_t kEntryBlockAlign = 256 * 840;          // 840 = lcm(1 .. 8)
static_assert(kIndexGroups == kFine, "the probe index is grouped by the count kernel's fine buckets");
static_assert(kSets == kProbeSetsMax, "the count kernel's probe arguments hold every entry set");

static bool probe_index_usable(const palace_ctx *ctx, const palace_eref_probe_index *ix)
{
    return ix->ehits_own[0] && ix->keys16[0] && std::memcmp(&ix->masks, &ctx->masks, sizeof(CoderMasks)) == 0;
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


namespace {

struct ScanBuffers {
    int64_t *tile_pre, *word_pre;
    uint64_t *any_w, *all_w, *good_w;
    uint32_t *any_p, *all_p;
    uint8_t *need, *active;                 // per chunk / per ref flags of eref_need_kernel
    uint8_t *hit_bytes;                     // indexed scan: a byte per SENTINEL (16 per word of any_w), see eref_ehits_scatter_kernel
    uint8_t *ehits[kSets];                  // indexed scan: a hit bit per index entry and entry set (eref_probe_sets_kernel)
    int64_t max_tiles, max_words;
};

int scan_buffers(palace_ctx *ctx, const int64_t *d_offsets, int64_t n_refs, int64_t total_bases, ScanBuffers *b, bool with_hit_bytes = false,
                 const size_t *ehits_bytes = nullptr)
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

#ifdef PALACE_STAMPS
extern "C" int palace_debug_stamps(palace_ctx *ctx, unsigned long long *h_out, int64_t n_words)
{
    PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PALACE_HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(palace_stamp_buf), static_cast<size_t>(n_words) * 8));
    return PALACE_OK;
}
#endif
