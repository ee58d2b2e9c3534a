// N4 on the device: the BGZF members of a BAM file inflated by the GPU, one wavefront per member.
//
// A BGZF member is an independent raw DEFLATE stream (RFC 1951) of at most 64 KiB of output; a 0.9 GB BAM holds ~36 000 of
// them.  DEFLATE decoding is serial inside a stream -- a symbol's position in the bit stream is known only when the one
// before it is decoded -- so the parallelism is across members: every member gets a wavefront whose 64 lanes all run the SAME
// decode (no divergence, nothing to broadcast) and split what is parallel inside a member: building the code tables,
// copying matches.  DEFLATE's window (the last 32 KiB of output) is the output buffer itself: a match reads the bytes it repeats
// back from where the wave wrote them (L2), which costs a wave a round trip per match -- and leaves a wave 4 KiB of LDS (its code
// tables) instead of 36, so that a CU holds 32 of them instead of 4.  The decode of ONE stream is a chain of dependent steps that
// issues an instruction every ~10 cycles; with one wave per SIMD (window in LDS: the first version, 5.7 GB/s of output for a
// whole BAM) the chip idled behind those latencies, with eight they overlap.
//
// Input words reach the lanes through a register window: lane l holds dword (base + l) of the member, the decode takes dword
// j with one v_readlane, and the following 64 dwords are always already requested -- the bit reader never waits for HBM.
//
// Memory safety does not depend on the input: input dwords are fetched only inside [first, last] dword of the member (zeros
// beyond), every LDS index is masked, every write to the output is bounded by the member's out_len, and every loop consumes
// input or produces output, both of which are bounded.  A member the decoder refuses (malformed, or a size mismatch) gets a
// non-zero status and the host decides it with zlib, which stays the authority on malformed input (as for the CPU decoder,
// host/inflate_fast.hpp).
#include "common.hpp"

namespace palace {

constexpr int kLitBits = 10, kDistBits = 8;

// base value and number of extra bits of a length code (257 .. 285 -> c = 0 .. 28) and of a distance code (0 .. 29), RFC 1951 3.2.5,
// as arithmetic (a table in memory is a dependent load per symbol)
__device__ __forceinline__ void len_code(int c, int32_t &base, int &extra)
{
    if (c < 8) { base = 3 + c; extra = 0; }
    else if (c == 28) { base = 258; extra = 0; }
    else { extra = (c >> 2) - 1; base = 3 + ((4 + (c & 3)) << extra); }
}
__device__ __forceinline__ void dist_code(int c, int32_t &base, int &extra)
{
    if (c < 4) { base = 1 + c; extra = 0; }
    else { extra = (c >> 1) - 1; base = 1 + ((2 + (c & 1)) << extra); }
}
__device__ const uint8_t kPreOrderD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum : int32_t { kInfOk = 0, kInfBadBlock = 1, kInfBadCode = 2, kInfBadDistance = 3, kInfOverrun = 4, kInfSize = 5, kInfInput = 6 };

// one canonical Huffman code in LDS: a primary table indexed by the next PRIMARY bits of the stream (entry = symbol | length << 9;
// 0 = the code is longer, or unused) and the canonical description (symbols in code order, count per length) for the rest
struct Code {
    uint16_t *primary;          // [1 << bits]
    uint16_t *sorted;           // symbols in code order
    uint16_t *count;            // [16]
    int bits;
};

struct BitReader {
    const uint32_t *base;       // 4-byte aligned start of the member's first dword
    int64_t last;               // index of the last dword that holds bytes of the member
    uint32_t win, win_next;     // lane l: dwords wbase + l and wbase + 64 + l
    int64_t wbase, next;        // next: index of the next dword to enter the bit buffer
    uint64_t buf;
    int cnt;

    __device__ __forceinline__ uint32_t load(int64_t j) const { return (j >= 0 && j <= last) ? base[j] : 0u; }
    __device__ __forceinline__ void seek(int64_t bit)            // position the reader at bit `bit` (from base)
    {
        const int lane = threadIdx.x & 63;
        wbase = bit >> 5;
        win = load(wbase + lane);
        win_next = load(wbase + 64 + lane);
        next = wbase;
        buf = 0; cnt = 0;
        refill();
        buf >>= (bit & 31); cnt -= static_cast<int>(bit & 31);
    }
    __device__ __forceinline__ uint32_t dword(int64_t j)         // j ascends: inside the window, or the first of the next one
    {
        if (j - wbase >= 64) {                                   // (uniform) slide: the words requested long ago become current
            win = win_next;
            wbase += 64;
            win_next = load(wbase + 64 + (threadIdx.x & 63));
        }
        return __builtin_amdgcn_readlane(win, __builtin_amdgcn_readfirstlane(static_cast<int>(j - wbase)));
    }
    __device__ __forceinline__ void refill()                     // >= 33 valid bits afterwards
    {
        if (cnt <= 32) {
            buf |= static_cast<uint64_t>(dword(next)) << cnt;
            next++;
            cnt += 32;
        }
    }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = static_cast<uint32_t>(buf) & ((1u << n) - 1); drop(n); return v; }
    __device__ __forceinline__ int64_t bit_pos() const { return next * 32 - cnt; }      // of the next unread bit
};

// lengths[0 .. n) -> the code's tables.  The lanes share the symbols (lane l: symbols l, l + 64, ...); a symbol's rank inside its
// length class comes from ballots, so codes are assigned in symbol order as the canonical construction demands.
// Returns false for a set zlib's inflate_table() rejects: over-subscribed, or incomplete other than a single 1-bit code
// (`lone_ok`: lengths / distances may be incomplete that way, the code-length code may not).
__device__ bool build_code(const Code &c, const uint8_t *lens, int n, bool lone_ok)
{
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 16; i += 64) c.count[i] = 0;
    for (int i = lane; i < (1 << c.bits); i += 64) c.primary[i] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the zeros are in LDS before the adds below
    int cnt[16];
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (int base = 0; base < n; base += 64) {                  // (uniform) class sizes by ballots
        const int sym = base + lane, l = sym < n ? lens[sym] : 0;
#pragma unroll
        for (int L = 1; L < 16; L++) cnt[L] += __popcll(__ballot(l == L));
    }
    int left = 1, max_len = 0, total = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
        left = (left << 1) - cnt[L];
        if (cnt[L]) max_len = L;
        total += cnt[L];
    }
    {                                                            // over-subscribed at some length
        int lf = 1;
#pragma unroll
        for (int L = 1; L < 16; L++) { lf = (lf << 1) - cnt[L]; if (lf < 0) return false; }
    }
    if (total == 0) return lone_ok;                              // no codes: every look-up fails (allowed for lengths / distances)
    if (left > 0 && !(lone_ok && max_len == 1)) return false;    // incomplete
#pragma unroll
    for (int L = 1; L < 16; L++)
        if (lane == L) c.count[L] = static_cast<uint16_t>(cnt[L]);         // (count[0] stays 0)
    int offs[16], code0[16];                                     // first index in `sorted` / first code of every length
    offs[1] = 0; code0[1] = 0; offs[0] = 0; code0[0] = 0;
#pragma unroll
    for (int L = 1; L < 15; L++) { offs[L + 1] = offs[L] + cnt[L]; code0[L + 1] = (code0[L] + cnt[L]) << 1; }
    for (int base = 0; base < n; base += 64) {
        const int sym = base + lane, l = sym < n ? lens[sym] : 0;
        int idx = 0, code = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            const unsigned long long m = __ballot(l == L);
            const int before = __popcll(m & ((1ull << lane) - 1));
            if (l == L) { idx = offs[L] + before; code = code0[L] + before; }
            offs[L] += __popcll(m); code0[L] += __popcll(m);
        }
        if (l) {
            c.sorted[idx] = static_cast<uint16_t>(sym);
            if (l <= c.bits) {                                   // every primary slot whose low l bits are the reversed code
                const uint32_t rev = __brev(static_cast<uint32_t>(code)) >> (32 - l);
                const uint16_t e = static_cast<uint16_t>(sym | (l << 9));
                for (uint32_t t = rev; t < (1u << c.bits); t += 1u << l) c.primary[t] = e;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    return true;
}

// the next symbol of code `c` (>= 0), or -1 for a bit pattern that is no code.  At least 15 bits are in the buffer.
__device__ __forceinline__ int decode_sym(const Code &c, BitReader &br)
{
    const uint32_t e = c.primary[static_cast<uint32_t>(br.buf) & ((1u << c.bits) - 1)];
    if (e) { br.drop(static_cast<int>(e >> 9)); return static_cast<int>(e & 511u); }
    // a longer code (rare): bit by bit against the canonical description
    int code = 0, first = 0, index = 0;
    uint32_t bits = static_cast<uint32_t>(br.buf);
    for (int len = 1; len <= 15; len++) {
        code |= static_cast<int>(bits & 1u);
        bits >>= 1;
        const int count = c.count[len];
        if (code - count < first) { br.drop(len); return c.sorted[index + (code - first)]; }
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return -1;
}

struct InflateArgs {
    const uint8_t *in;
    const int64_t *in_off;
    const int32_t *in_len;
    const int64_t *out_off;
    const int32_t *out_len;
    uint8_t *out;
    int32_t *status;
    int64_t n_members;
};

__global__ __launch_bounds__(64) void bgzf_inflate_kernel(InflateArgs a)
{
    __shared__ uint16_t lit_primary[1 << kLitBits], lit_sorted[288], lit_count[16];
    __shared__ uint16_t dist_primary[1 << kDistBits], dist_sorted[32], dist_count[16];
    __shared__ uint16_t pre_primary[1 << 7], pre_sorted[19], pre_count[16];
    __shared__ uint8_t lens[352];                                            // [0, 19) code-length code; [20, 20 + 316) the block's lengths
    const int lane = threadIdx.x & 63;
    const int64_t m = blockIdx.x;
    if (m >= a.n_members) return;
    const int64_t in_off = a.in_off[m], out_off = a.out_off[m];
    const int32_t in_len = a.in_len[m], out_len = a.out_len[m];
    if (in_len < 0 || out_len < 0 || out_len > 65536 + 0) { if (lane == 0) a.status[m] = kInfSize; return; }
    if (out_len == 0 && in_len == 0) { if (lane == 0) a.status[m] = kInfOk; return; }
    const Code lit{lit_primary, lit_sorted, lit_count, kLitBits}, dist{dist_primary, dist_sorted, dist_count, kDistBits},
               pre{pre_primary, pre_sorted, pre_count, 7};
    BitReader br;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(a.in + in_off);
    br.base = reinterpret_cast<const uint32_t *>(addr & ~static_cast<uintptr_t>(3));
    const int skip = static_cast<int>(addr & 3);
    br.last = in_len > 0 ? (skip + static_cast<int64_t>(in_len) - 1) >> 2 : -1;
    br.seek(skip * 8);
    const int64_t end_bit = (skip + static_cast<int64_t>(in_len)) * 8;     // the member's DEFLATE data ends here
    uint8_t *const out = a.out + out_off;
    int32_t opos = 0, err = kInfOk;
    // what the wave wrote so far is in memory before anything reads it back (a match's source may be a literal lane 0 stored a moment
    // ago): a release fence of the wave's stores, then loads that are served where the stores went
    auto written = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); };
    auto back = [&](int32_t at) { return __hip_atomic_load(out + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    bool last_block = false;
    while (!last_block && err == kInfOk) {
        br.refill();
        last_block = br.take(1) != 0;
        const uint32_t btype = br.take(2);
        if (btype == 0) {
            // stored: LEN / NLEN at the next byte boundary, then LEN raw bytes
            const int64_t p_bit = (br.bit_pos() + 7) & ~7ll;
            br.seek(p_bit);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ 0xffffu) != nlen) { err = kInfBadBlock; break; }
            const int64_t data_bit = p_bit + 32;
            if (data_bit + static_cast<int64_t>(len) * 8 > end_bit || static_cast<int32_t>(len) > out_len - opos) { err = kInfOverrun; break; }
            const uint8_t *src = reinterpret_cast<const uint8_t *>(br.base) + (data_bit >> 3);
            for (uint32_t i = lane; i < len; i += 64) out[opos + static_cast<int32_t>(i)] = src[i];
            opos += static_cast<int32_t>(len);
            br.seek(data_bit + static_cast<int64_t>(len) * 8);
            continue;
        }
        if (btype == 3) { err = kInfBadBlock; break; }
        if (btype == 1) {                                                      // fixed code (RFC 1951 3.2.6)
            for (int i = lane; i < 288; i += 64) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            for (int i = lane; i < 32; i += 64) lens[288 + i] = 5;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (!build_code(lit, lens, 288, true) || !build_code(dist, lens + 288, 32, true)) { err = kInfBadCode; break; }   // (32 distance codes of 5 bits: 30 and 31 never occur in valid data)
        } else {
            br.refill();
            const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = kInfBadCode; break; }
            for (int i = lane; i < 19; i += 64) lens[i] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            for (uint32_t i = 0; i < hclen; i++) {
                br.refill();
                const uint32_t v = br.take(3);
                if (lane == 0) lens[kPreOrderD[i]] = static_cast<uint8_t>(v);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (!build_code(pre, lens, 19, false)) { err = kInfBadCode; break; }
            // the hlit + hdist code lengths, run-length coded with the code-length code; kept in registers of the decode (all
            // lanes alike) and stored by lane 0
            uint32_t n = 0, prev = 0;
            const uint32_t want = hlit + hdist;
            while (n < want && err == kInfOk) {
                br.refill();
                const int sym = decode_sym(pre, br);
                if (sym < 0) { err = kInfBadCode; break; }
                if (sym < 16) { if (lane == 0) lens[20 + n] = static_cast<uint8_t>(sym); prev = static_cast<uint32_t>(sym); n++; continue; }
                uint32_t rep, val = 0;
                br.refill();
                if (sym == 16) { if (n == 0) { err = kInfBadCode; break; } val = prev; rep = 3 + br.take(2); }
                else if (sym == 17) rep = 3 + br.take(3);
                else rep = 11 + br.take(7);
                if (n + rep > want) { err = kInfBadCode; break; }
                for (uint32_t i = lane; i < rep; i += 64) lens[20 + n + i] = static_cast<uint8_t>(val);
                n += rep; prev = val;
            }
            if (err != kInfOk) break;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lens[20 + 256] == 0) { err = kInfBadCode; break; }            // no end-of-block code
            if (!build_code(lit, lens + 20, static_cast<int>(hlit), true) || !build_code(dist, lens + 20 + hlit, static_cast<int>(hdist), true)) { err = kInfBadCode; break; }
        }
        // ---- the block's symbols ----
        for (;;) {
            br.refill();                                                       // >= 33 bits: a literal/length code and its extra bits
            if (br.bit_pos() > end_bit + 64) { err = kInfInput; break; }       // far past the member's data: a stream that does not end
            const int sym = decode_sym(lit, br);
            if (sym < 0) { err = kInfBadCode; break; }
            if (sym < 256) {
                if (opos >= out_len) { err = kInfOverrun; break; }
                if (lane == 0) out[opos] = static_cast<uint8_t>(sym);
                opos++;
                continue;
            }
            if (sym == 256) break;
            if (sym > 285) { err = kInfBadCode; break; }
            int32_t lbase, dbase;
            int lextra, dextra;
            len_code(sym - 257, lbase, lextra);
            const int32_t len = lbase + static_cast<int32_t>(br.take(lextra));
            br.refill();
            const int ds = decode_sym(dist, br);
            if (ds < 0 || ds > 29) { err = kInfBadDistance; break; }
            br.refill();                                                       // (a distance code may have used 15 of the 33 bits)
            dist_code(ds, dbase, dextra);
            const int32_t d = dbase + static_cast<int32_t>(br.take(dextra));
            if (d > opos) { err = kInfBadDistance; break; }
            if (len > out_len - opos) { err = kInfOverrun; break; }
            // the copy: sources lie below opos, destinations at and above it; an overlapping match (d < len) repeats its d bytes
            written();
            for (int32_t i = lane; i < len; i += 64) {
                const int32_t s = d >= len ? i : i % d;
                out[opos + i] = back(opos - d + s);
            }
            opos += len;
        }
    }
    if (err == kInfOk) {
        if (opos != out_len) err = kInfSize;
        else if (br.bit_pos() > end_bit + 7) err = kInfInput;                  // the stream ran past the member's data
    }
    if (lane == 0) a.status[m] = err;
}

}  // namespace palace

using namespace palace;

extern "C" int palace_bgzf_inflate(palace_ctx *ctx, const uint8_t *d_in, int64_t n_members, const int64_t *d_in_off, const int32_t *d_in_len,
                                   const int64_t *d_out_off, const int32_t *d_out_len, uint8_t *d_out, int32_t *d_status)
{
    PALACE_REQUIRE(ctx && n_members >= 0 && n_members < (1ll << 31), "bad argument");
    if (n_members == 0) return PALACE_OK;
    PALACE_REQUIRE(d_in && d_in_off && d_in_len && d_out_off && d_out_len && d_out && d_status, "null device pointer");
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const InflateArgs a{d_in, d_in_off, d_in_len, d_out_off, d_out_len, d_out, d_status, n_members};
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(static_cast<unsigned>(n_members)), dim3(64), 0, ctx->stream, a);
    PALACE_HIP_TRY(hipGetLastError());
    return PALACE_OK;
}
