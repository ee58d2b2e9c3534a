// A DEFLATE (RFC 1951) decoder for BGZF members, written for this loader: whole member in, whole member out, sizes known
// beforehand (the BGZF trailer gives ISIZE), no streaming state.  zlib's inflate() is a general streaming decoder and spends
// most of its time per symbol on state it does not need here; on BAM payloads (packed bases and qualities: literal-heavy,
// poorly compressible) it delivers ~185 MB/s per thread, this one ~2.5x that.  generateGraph's wall time at the 1M-contig
// configuration was 0.8 s of inflate on 16 threads out of 1.33 s.
//
// Contract: inflate_fast() either returns true with exactly out_len bytes written -- the bytes zlib would produce -- or returns
// false having written nothing outside [out, out + out_len); on false the caller runs zlib on the member, which stays the
// authority on malformed input (its error behaviour is the loader's).  It is deliberately stricter than zlib nowhere it
// matters and never more lenient: code sets zlib rejects (over-subscribed, incomplete other than a single 1-bit code) are
// rejected here too.  Memory safety does not depend on the input: every read of the input is bounded by in_len + in_slack,
// every write by out_len.
//
// Technique (standard for fast inflaters): a 64-bit bit buffer refilled by one unaligned 8-byte load per symbol, tables
// indexed by the next 11 (literal/length) or 8 (distance) bits whose entries carry the decoded value, its extra-bit count
// and the code length, second-level tables for longer codes, up to three literals per refill, matches copied 8 bytes at a time.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace palace_host {
namespace inflate_detail {

constexpr int kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr uint32_t kFlagLit2 = 1u << 12, kFlagLit = 1u << 13, kFlagEob = 1u << 14, kFlagSub = 1u << 15;
// entry: bits 0..7 code bits to drop, bits 8..11 extra bits (or sub-table index bits), flags, bits 16..31 value
// (a literal; with kFlagLit2 two literals -- second in bits 24..31 -- whose codes together fit the primary index)
constexpr int kLitCap = (1 << kLitBits) + 288 * 16, kDistCap = (1 << kDistBits) + 32 * 128;

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                       4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum Kind { kPre, kLit, kDist };

inline uint32_t reverse_bits(uint32_t code, int len)      // the low `len` (<= 15) bits of code, reversed
{
    uint32_t v = code;
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0f0fu) << 4) | ((v >> 4) & 0x0f0fu);
    v = ((v & 0x00ffu) << 8) | ((v >> 8) & 0x00ffu);
    return v >> (16 - len);
}

inline uint32_t symbol_entry(Kind kind, int sym)       // value / extra / flags of a symbol; 0 = a symbol that must not occur
{
    if (kind == kPre) return static_cast<uint32_t>(sym) << 16;
    if (kind == kLit) {
        if (sym < 256) return (static_cast<uint32_t>(sym) << 16) | kFlagLit;
        if (sym == 256) return kFlagEob;
        if (sym > 285) return 0;
        return (static_cast<uint32_t>(kLenBase[sym - 257]) << 16) | (static_cast<uint32_t>(kLenExtra[sym - 257]) << 8);
    }
    if (sym > 29) return 0;
    return (static_cast<uint32_t>(kDistBase[sym]) << 16) | (static_cast<uint32_t>(kDistExtra[sym]) << 8);
}

// canonical Huffman code of `n` lengths -> lookup table; false = a set zlib's inflate_table() rejects, or no room
inline bool build_table(Kind kind, const uint8_t *lens, int n, int primary, uint32_t *table, int cap)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int max_len = 15;
    while (max_len > 0 && count[max_len] == 0) max_len--;
    const int size1 = 1 << primary;
    std::memset(table, 0, sizeof(uint32_t) * static_cast<size_t>(size1));
    if (max_len == 0) return kind != kPre;                         // no codes at all: every look-up fails (zlib: allowed for lens / dists)
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;                                // over-subscribed
    }
    if (left > 0 && (kind == kPre || max_len != 1)) return false;  // incomplete (inftrees.c: only a lone 1-bit code may be)
    const int sub_bits = max_len > primary ? max_len - primary : 0;
    int next_sub = size1;
    // symbols in code order (by length, then by value): a counting sort instead of one pass over the symbols per length
    uint16_t sorted[320];
    int at[17];
    at[1] = 0;
    for (int l = 1; l <= 15; l++) at[l + 1] = at[l] + count[l];
    for (int sym = 0; sym < n; sym++)
        if (lens[sym]) sorted[at[lens[sym]]++] = static_cast<uint16_t>(sym);
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= max_len; len++) {
        for (int c = 0; c < count[len]; c++, k++) {
            const int sym = sorted[k];
            const uint32_t rev = reverse_bits(code, len);
            uint32_t e = symbol_entry(kind, sym);
            if (len <= primary) {
                if (e || kind == kPre) e |= static_cast<uint32_t>(len);            // (a forbidden symbol keeps the all-zero entry)
                for (uint32_t i = rev; i < static_cast<uint32_t>(size1); i += 1u << len) table[i] = e;
            } else {
                const uint32_t prefix = rev & static_cast<uint32_t>(size1 - 1);
                if (!(table[prefix] & kFlagSub)) {
                    if (next_sub + (1 << sub_bits) > cap) return false;
                    std::memset(table + next_sub, 0, sizeof(uint32_t) << sub_bits);
                    table[prefix] = (static_cast<uint32_t>(next_sub) << 16) | kFlagSub | (static_cast<uint32_t>(sub_bits) << 8) | static_cast<uint32_t>(primary);
                    next_sub += 1 << sub_bits;
                }
                uint32_t *sub = table + (table[prefix] >> 16);
                if (e) e |= static_cast<uint32_t>(len - primary);
                for (uint32_t i = rev >> primary; i < (1u << sub_bits); i += 1u << (len - primary)) sub[i] = e;
            }
            code++;
        }
        code <<= 1;
    }
    if (kind == kLit) {
        // two literals per look-up where both codes fit into the primary index: BAM payloads are literal-heavy with short codes
        // (binned qualities: 2-3 bits, packed bases: ~4), and a symbol costs its table look-up, not its bits
        uint32_t single[1 << kLitBits];
        std::memcpy(single, table, sizeof single);
        for (int i = 0; i < size1; i++) {
            const uint32_t e1 = single[i];
            if (!(e1 & kFlagLit)) continue;
            const int l1 = static_cast<int>(e1 & 0xff), room = primary - l1;
            if (room < 1) continue;
            const uint32_t e2 = single[i >> l1];                   // the bits behind code 1; valid if code 2 needs no more than `room` of them
            if (!(e2 & kFlagLit) || static_cast<int>(e2 & 0xff) > room) continue;
            table[i] = ((e1 >> 16) << 16) | ((e2 >> 16) << 24) | kFlagLit | kFlagLit2 | static_cast<uint32_t>(l1 + static_cast<int>(e2 & 0xff));
        }
    }
    return true;
}

struct Tables {
    uint32_t lit[kLitCap], dist[kDistCap];
};

inline const Tables *fixed_tables()
{
    static const Tables *t = [] {
        Tables *f = new Tables;
        uint8_t l[288], d[32];
        for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(kLit, l, 288, kLitBits, f->lit, kLitCap);
        build_table(kDist, d, 32, kDistBits, f->dist, kDistCap);
        return f;
    }();
    return t;
}

}  // namespace inflate_detail

namespace inflate_detail {
// the decoder proper; compiled twice below (plain x86-64, and with BMI2 where its variable shifts and bit-field extracts need
// no count register: +7 % measured)
static inline __attribute__((always_inline)) bool inflate_body(const uint8_t *in, size_t in_len, size_t in_slack, uint8_t *out, size_t out_len)
{
    static const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const size_t fast_end = in_len + in_slack >= 8 ? in_len + in_slack - 8 : 0;   // ipos <= fast_end: an 8-byte load at ipos is in bounds
    const bool can_fast = in_len + in_slack >= 8;
    uint64_t bitbuf = 0;
    unsigned bitcnt = 0;
    size_t ipos = 0;                       // bytes of input moved into the bit buffer (may run past in_len with zeros: checked at the end)
    size_t opos = 0;
    Tables own;
    // at least 56 valid bits afterwards
    auto refill = [&]() {
        if (can_fast && ipos <= fast_end) {
            uint64_t w;
            std::memcpy(&w, in + ipos, 8);
            bitbuf |= w << bitcnt;
            ipos += (63 - bitcnt) >> 3;
            bitcnt |= 56;
        } else {
            while (bitcnt <= 56) {
                const uint64_t byte = ipos < in_len ? in[ipos] : 0;
                bitbuf |= byte << bitcnt;
                ipos++;
                bitcnt += 8;
            }
        }
    };
    auto drop = [&](unsigned n) { bitbuf >>= n; bitcnt -= n; };
    auto take = [&](unsigned n) { const uint32_t v = static_cast<uint32_t>(bitbuf & ((1ull << n) - 1)); drop(n); return v; };
    for (;;) {
        refill();
        const uint32_t bfinal = take(1), btype = take(2);
        const uint32_t *lt, *dt;
        if (btype == 0) {
            // stored: back to a byte boundary of the INPUT (whole unread bytes in the buffer are given back)
            drop(bitcnt & 7);
            // (after the branch-free refill the buffer may hold bits above bitcnt; they are discarded with it)
            const size_t unread = bitcnt >> 3;
            if (ipos < unread) return false;
            size_t p = ipos - unread;
            if (p + 4 > in_len) return false;
            const uint32_t len = in[p] | (static_cast<uint32_t>(in[p + 1]) << 8), nlen = in[p + 2] | (static_cast<uint32_t>(in[p + 3]) << 8);
            if ((len ^ 0xffffu) != nlen) return false;
            p += 4;
            if (len > in_len - p || len > out_len - opos) return false;
            std::memcpy(out + opos, in + p, len);
            opos += len;
            ipos = p + len;
            bitbuf = 0;
            bitcnt = 0;
            if (bfinal) break;
            continue;
        }
        if (btype == 3) return false;
        if (btype == 1) {
            lt = fixed_tables()->lit;
            dt = fixed_tables()->dist;
        } else {
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            uint8_t pre_lens[19] = {0};
            refill();
            for (uint32_t i = 0; i < hclen; i++) {
                if (bitcnt < 3) refill();
                pre_lens[kPreOrder[i]] = static_cast<uint8_t>(take(3));
            }
            uint32_t pre[1 << kPreBits];
            if (!build_table(kPre, pre_lens, 19, kPreBits, pre, 1 << kPreBits)) return false;
            uint8_t lens[286 + 30 + 138];
            uint32_t n = 0;
            while (n < hlit + hdist) {
                refill();
                const uint32_t e = pre[bitbuf & ((1u << kPreBits) - 1)];
                if ((e & 0xff) == 0) return false;
                drop(e & 0xff);
                const uint32_t sym = e >> 16;
                if (sym < 16) { lens[n++] = static_cast<uint8_t>(sym); continue; }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (n == 0) return false;
                    val = lens[n - 1];
                    rep = 3 + take(2);
                } else if (sym == 17) rep = 3 + take(3);
                else rep = 11 + take(7);
                if (n + rep > hlit + hdist) return false;
                std::memset(lens + n, static_cast<int>(val), rep);
                n += rep;
            }
            if (lens[256] == 0) return false;                               // no end-of-block code
            if (!build_table(kLit, lens, static_cast<int>(hlit), kLitBits, own.lit, kLitCap)) return false;
            if (!build_table(kDist, lens + hlit, static_cast<int>(hdist), kDistBits, own.dist, kDistCap)) return false;
            lt = own.lit;
            dt = own.dist;
        }
        // ---- symbols of the block ----
        // one or two literals of an entry; `roomy`: at least 8 bytes of output are left, so the pair is stored as it is
        // (an entry with one literal has 0 in the second byte, which the next symbol overwrites)
        auto put_lits = [&](uint32_t e, bool roomy) -> bool {
            const uint32_t two = (e >> 12) & 1u;
            if (roomy) {
                const uint16_t v = static_cast<uint16_t>(e >> 16);
                std::memcpy(out + opos, &v, 2);
                opos += 1 + two;
                return true;
            }
            if (out_len - opos < 1 + two) return false;
            out[opos++] = static_cast<uint8_t>(e >> 16);
            if (two) out[opos++] = static_cast<uint8_t>(e >> 24);
            return true;
        };
        // One refill serves up to three literal entries or one match.  After a match the NEXT entry is looked up before the
        // copy is done, so that its load overlaps the copy (matches dominate BAM payloads: ~5 bytes each on average).
        refill();                                                           // >= 56 bits
        uint32_t e = lt[bitbuf & ((1u << kLitBits) - 1)];                   // raw primary entry of the bits at hand
        for (;;) {
            const bool roomy = out_len - opos >= 8;                         // up to three entries = six literal bytes follow unchecked
            int lits = 0;
            for (;;) {                                                      // entries of <= 15 bits each
                if (e & kFlagSub) {
                    drop(kLitBits);
                    e = lt[(e >> 16) + (bitbuf & ((1u << ((e >> 8) & 15)) - 1))];
                }
                drop(e & 0xff);
                if (!(e & kFlagLit)) break;
                if (!put_lits(e, roomy)) return false;
                if (++lits == 3) break;
                e = lt[bitbuf & ((1u << kLitBits) - 1)];
            }
            if (e & kFlagLit) {                                             // three literal entries: next round
                refill();
                e = lt[bitbuf & ((1u << kLitBits) - 1)];
                continue;
            }
            if (lits) refill();                                             // literals ate into the budget of the match
            if (e & kFlagEob) break;
            if ((e & 0xff) == 0) return false;                              // unused code, or a symbol that must not occur
            // a length: <= 5 extra bits, then a distance code of <= 15 bits with <= 13 extra bits (>= 41 bits are there)
            const uint32_t len = (e >> 16) + take((e >> 8) & 15);
            uint32_t d = dt[bitbuf & ((1u << kDistBits) - 1)];
            if (d & kFlagSub) {
                drop(kDistBits);
                d = dt[(d >> 16) + (bitbuf & ((1u << ((d >> 8) & 15)) - 1))];
            }
            if ((d & 0xff) == 0) return false;
            drop(d & 0xff);
            const size_t dist = (d >> 16) + take((d >> 8) & 15);
            if (dist > opos || len > out_len - opos) return false;
            refill();
            e = lt[bitbuf & ((1u << kLitBits) - 1)];                        // the next entry's load runs beside the copy
            uint8_t *dst = out + opos;
            const uint8_t *src = dst - dist;
            if (dist >= 8 && out_len - opos >= static_cast<size_t>(len) + 8) {          // whole words; may overshoot the match by <= 7 bytes, inside `out`
                uint8_t *const end = dst + len;
                do {
                    uint64_t w;
                    std::memcpy(&w, src, 8);
                    std::memcpy(dst, &w, 8);
                    src += 8; dst += 8;
                } while (dst < end);
            } else if (dist == 1) {
                std::memset(dst, *src, len);
            } else {
                for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
            }
            opos += len;
        }
        if (bfinal) break;
    }
    // the stream must have ended inside the member, and filled the output exactly
    const size_t unread = bitcnt >> 3;
    return opos == out_len && ipos >= unread && ipos - unread <= in_len;
}

__attribute__((target("bmi2"))) inline bool inflate_bmi2(const uint8_t *in, size_t in_len, size_t in_slack, uint8_t *out, size_t out_len)
{
    return inflate_body(in, in_len, in_slack, out, out_len);
}
inline bool inflate_plain(const uint8_t *in, size_t in_len, size_t in_slack, uint8_t *out, size_t out_len)
{
    return inflate_body(in, in_len, in_slack, out, out_len);
}
}  // namespace inflate_detail

// in_slack: bytes behind in + in_len that may be READ (their content is irrelevant); 8 or more makes every refill one load
inline bool inflate_fast(const uint8_t *in, size_t in_len, size_t in_slack, uint8_t *out, size_t out_len)
{
    static const bool bmi2 = __builtin_cpu_supports("bmi2") != 0;
    return bmi2 ? inflate_detail::inflate_bmi2(in, in_len, in_slack, out, out_len) : inflate_detail::inflate_plain(in, in_len, in_slack, out, out_len);
}

}  // namespace palace_host
