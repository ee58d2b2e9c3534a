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
#pragma once
#include "common.hpp"

#ifdef PALACE_STAMPS        // diagnostic build (tools/dbg/stamps.py): per-workgroup phase stamps of the partition kernels
extern __device__ unsigned long long palace_stamp_buf[8 * 65536];      // (defined in eref.hip)
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

constexpr int kFineWords = kFine / 32;               // 2048 u32 per plane per fine bucket

// ------------------------------------------------------------------------------------------
// tiling of a set of sequences: tile = kTileChunks x 64 positions of one sequence
// ------------------------------------------------------------------------------------------
constexpr int kTileChunks = 32;                 // 2048 positions per 256-thread block
constexpr int kTilePos = kTileChunks * 64;

__device__ __forceinline__ int64_t find_seq(const int64_t *__restrict__ pre, int64_t n, int64_t tile)
{
    int64_t lo = 0, hi = n;            // largest r with pre[r] <= tile
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (pre[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// the eight 16-bit entries of one 16-byte vector against a 2^16-bit slice in LDS -> a byte of hit bits (count kernel, probe kernel)
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

// the per-DB probe index (eref_index.hip): entry sets and grouping
constexpr int kIndexGroups = 1 << 16, kGroupsPerProbe = kIndexGroups / kBuckets;
constexpr int kSets = 4, kSentinelSet = 3, kSentinelStride = 4;

static inline int masks_from_header(const uint8_t *hdr, CoderMasks *out)
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

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline KeyBuckets ctx_buckets(const palace_ctx *ctx)
{
    KeyBuckets k;
    for (int i = 0; i < 4; i++) k.m[i] = ctx->key_buckets[i];
    return k;
}

// Workspace of one count_reads call over `total_bases` positions: slab size, region capacities, byte counts.
constexpr size_t kTouchedBytes = 8192 + 256;             // one bit per fine bucket (2^16 bits), padded
struct CountPlan {
    int64_t slab_bases_max = 0, n_slabs = 0, n_chunks = 0;
    DensityCaps caps1{}, caps2{};
    size_t cur1_bytes = 0, cur2_bytes = 0, buf1_bytes = 0, buf2_bytes = 0, words_bytes = 0;
    size_t total() const { return cur1_bytes + buf1_bytes + cur2_bytes + kTouchedBytes + 5 * words_bytes + buf2_bytes; }
};
constexpr int64_t kRegions = static_cast<int64_t>(kL1Buckets) * kL1Replicas;

// Scratch of one count call, carved out of the context's workspace (sizes: CountPlan).
struct CountBufs {
    unsigned int *cursor2 = nullptr, *touched = nullptr, *cursor1 = nullptr;
    unsigned long long *words = nullptr;                 // 5 x words_bytes (ASCII entry: read ends, dropped, three streams) or nothing
    uint32_t *buf1 = nullptr;
    uint16_t *buf2 = nullptr;
};

// scan workspace (eref_scan.hip): tile / word prefixes of the refs, hit words, prefix counts, flags; also what the probe-index build lays its
// position ids out by (eref_index.hip)
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
                 const size_t *ehits_bytes = nullptr);

}  // namespace palace

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

namespace palace {
constexpr size_t kEntryBlockAlign = 256 * 840;          // 840 = lcm(1 .. 8)
static_assert(kIndexGroups == kFine, "the probe index is grouped by the count kernel's fine buckets");
static inline bool probe_index_usable(const palace_ctx *ctx, const palace_eref_probe_index *ix)
{
    return ix->ehits_own[0] && ix->keys16[0] && std::memcmp(&ix->masks, &ctx->masks, sizeof(CoderMasks)) == 0;
}
}  // namespace palace
