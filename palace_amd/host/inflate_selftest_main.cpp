// inflate_selftest -- differential test of inflate_fast.hpp against zlib (CPU only, run by tests/test_host_parsers.py).
//   inflate_selftest            many generated inputs x every zlib strategy / level / window, truncations and bit flips:
//                               for every stream, inflate_fast either fails or gives exactly zlib's bytes; prints "ok <n>"
//   inflate_selftest time <bam> decodes every BGZF member of a file with both and prints MB/s (single thread)
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "inflate_fast.hpp"
#include "fastx.hpp"

using palace_host::inflate_fast;

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &src, int level, int strategy, int mem, size_t chunk)
{
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, mem, strategy) != Z_OK) std::abort();
    std::vector<uint8_t> out(deflateBound(&zs, src.size()) + 64 + (chunk ? 16 * (src.size() / chunk + 2) : 0));
    zs.next_out = out.data(); zs.avail_out = static_cast<uInt>(out.size());
    if (chunk == 0) {
        zs.next_in = const_cast<Bytef *>(src.data()); zs.avail_in = static_cast<uInt>(src.size());
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) std::abort();
    } else {                                   // several blocks (full flushes in between: stored empty blocks appear too)
        size_t p = 0;
        while (p < src.size()) {
            const size_t n = std::min(chunk, src.size() - p);
            zs.next_in = const_cast<Bytef *>(src.data() + p); zs.avail_in = static_cast<uInt>(n);
            if (deflate(&zs, Z_FULL_FLUSH) != Z_OK) std::abort();
            p += n;
        }
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) std::abort();
    }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

// zlib's verdict on a raw stream that must give exactly want_len bytes: true + bytes, or false
static bool zlib_inflate(const uint8_t *in, size_t in_len, std::vector<uint8_t> &out, size_t want_len)
{
    z_stream zs{};
    if (inflateInit2(&zs, -15) != Z_OK) std::abort();
    out.assign(want_len + 1, 0);
    zs.next_in = const_cast<Bytef *>(in); zs.avail_in = static_cast<uInt>(in_len);
    zs.next_out = out.data(); zs.avail_out = static_cast<uInt>(want_len);
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0;
    inflateEnd(&zs);
    out.resize(want_len);
    return ok;
}

static long n_checked = 0, n_fast_ok = 0;

// the contract: inflate_fast fails, or agrees with zlib (which then must have succeeded too); never writes outside `out`
static bool check(const std::vector<uint8_t> &stream, size_t want_len, size_t slack, bool must_succeed)
{
    std::vector<uint8_t> in(stream.size() + slack, 0xA7);
    std::memcpy(in.data(), stream.data(), stream.size());
    std::vector<uint8_t> guard(want_len + 64, 0x5C), ref;
    const bool fast = inflate_fast(in.data(), stream.size(), slack, guard.data() + 32, want_len);
    for (size_t i = 0; i < 32; i++)
        if (guard[i] != 0x5C || guard[32 + want_len + i] != 0x5C) { std::fprintf(stderr, "write outside the output buffer\n"); return false; }
    {                                          // the build without BMI2 is the same source: same verdict, same bytes
        std::vector<uint8_t> other(want_len + 1, 0x5C);
        const bool plain = palace_host::inflate_detail::inflate_plain(in.data(), stream.size(), slack, other.data(), want_len);
        if (plain != fast || (fast && std::memcmp(other.data(), guard.data() + 32, want_len) != 0)) { std::fprintf(stderr, "plain and BMI2 builds disagree\n"); return false; }
    }
    const bool z = zlib_inflate(stream.data(), stream.size(), ref, want_len);
    n_checked++;
    if (fast) {
        n_fast_ok++;
        if (!z) { std::fprintf(stderr, "inflate_fast accepted a stream zlib rejects (len %zu)\n", stream.size()); return false; }
        if (std::memcmp(guard.data() + 32, ref.data(), want_len) != 0) { std::fprintf(stderr, "bytes differ from zlib's\n"); return false; }
    } else if (must_succeed) {
        std::fprintf(stderr, "inflate_fast refused a valid stream (len %zu -> %zu, zlib %s)\n", stream.size(), want_len, z ? "ok" : "failed");
        return false;
    }
    return true;
}

static int self_test()
{
    std::mt19937_64 rng(12345);
    std::vector<std::vector<uint8_t>> inputs;
    inputs.push_back({});
    inputs.push_back({'a'});
    inputs.push_back(std::vector<uint8_t>(65280, 'x'));                            // one long run: distance 1, length 258 chains
    for (int period : {2, 3, 5, 7, 8, 9, 31, 258, 259, 32768}) {                   // every short distance, the window edge
        std::vector<uint8_t> v(70000);
        for (size_t i = 0; i < v.size(); i++) v[i] = static_cast<uint8_t>((i % static_cast<size_t>(period)) * 37 + (i / 9973));
        inputs.push_back(v);
    }
    for (int k = 0; k < 6; k++) {                                                  // random bytes over alphabets of 2 .. 256 symbols
        std::vector<uint8_t> v(1000 + static_cast<size_t>(rng() % 64000));
        const int alpha = 1 << (1 + k + (k > 3 ? k - 3 : 0));
        for (auto &b : v) b = static_cast<uint8_t>(rng() % static_cast<uint64_t>(std::min(alpha, 256)));
        inputs.push_back(v);
    }
    {                                                                              // BAM-like: packed bases, binned qualities, names, repeats of earlier records
        std::vector<uint8_t> v;
        while (v.size() < 65000) {
            for (int i = 0; i < 36; i++) v.push_back(static_cast<uint8_t>(rng() % 3 ? 0 : rng()));
            for (int i = 0; i < 12; i++) v.push_back("0123456789abcdef"[rng() & 15]);
            for (int i = 0; i < 75; i++) v.push_back(static_cast<uint8_t>(((1u << (rng() & 3)) << 4) | (1u << (rng() & 3))));
            for (int i = 0; i < 150; i++) v.push_back(static_cast<uint8_t>("\x02\x0b\x19\x25"[rng() & 3]));
            if (rng() % 5 == 0 && v.size() > 600) { const size_t from = rng() % (v.size() - 300); for (int i = 0; i < 273; i++) v.push_back(v[from + static_cast<size_t>(i)]); }
        }
        inputs.push_back(v);
    }
    for (const auto &src : inputs) {
        for (int strategy : {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}) {
            for (int level : {0, 1, 4, 6, 9}) {
                for (size_t chunk : {static_cast<size_t>(0), static_cast<size_t>(3001)}) {
                    const std::vector<uint8_t> st = deflate_raw(src, level, strategy, level == 1 ? 1 : 8, chunk);
                    for (size_t slack : {static_cast<size_t>(0), static_cast<size_t>(3), static_cast<size_t>(8), static_cast<size_t>(40)})
                        if (!check(st, src.size(), slack, true)) return 1;
                    // damaged streams: truncated, extended expectations, flipped bits -- fail or agree, never crash
                    if (st.size() > 4) {
                        std::vector<uint8_t> cut(st.begin(), st.begin() + static_cast<long>(st.size() - 1 - rng() % std::min<size_t>(st.size() - 1, 40)));
                        if (!check(cut, src.size(), 8, false)) return 1;
                    }
                    if (!check(st, src.size() + 1, 8, false) || (src.size() && !check(st, src.size() - 1, 8, false))) return 1;
                    for (int f = 0; f < 12 && !st.empty(); f++) {
                        std::vector<uint8_t> bad = st;
                        const size_t at = f < 6 ? rng() % std::min<size_t>(bad.size(), 40) : rng() % bad.size();     // headers get their share
                        bad[at] ^= static_cast<uint8_t>(1u << (rng() & 7));
                        if (!check(bad, src.size(), f & 1 ? 8 : 0, false)) return 1;
                    }
                }
            }
        }
    }
    for (int k = 0; k < 3000; k++) {                                               // noise as a stream
        std::vector<uint8_t> noise(1 + rng() % 300);
        for (auto &b : noise) b = static_cast<uint8_t>(rng());
        if (!check(noise, rng() % 2000, k & 1 ? 8 : 0, false)) return 1;
    }
    std::printf("ok %ld streams checked, %ld decoded by inflate_fast\n", n_checked, n_fast_ok);
    return 0;
}

static int time_file(const char *path)
{
    const std::vector<char> f = palace_host::read_file(path);
    const uint8_t *d = reinterpret_cast<const uint8_t *>(f.data());
    struct Member { size_t in, in_len, out_len; };
    std::vector<Member> ms;
    size_t total = 0;
    for (size_t p = 0; p + 18 <= f.size();) {
        const size_t xlen = d[p + 10] | (d[p + 11] << 8), bsize = (d[p + 16] | (d[p + 17] << 8)) + 1u;     // (BC is the first subfield in what htslib and synthbam write)
        const size_t isize = d[p + bsize - 4] | (d[p + bsize - 3] << 8) | (d[p + bsize - 2] << 16) | (static_cast<size_t>(d[p + bsize - 1]) << 24);
        ms.push_back({p + 12 + xlen, bsize - xlen - 20, isize});
        total += isize;
        p += bsize;
    }
    std::vector<uint8_t> a(65536 + 8), b;
    for (int which = 0; which < 2; which++) {
        const auto t0 = std::chrono::steady_clock::now();
        size_t refused = 0;
        for (const Member &m : ms) {
            if (!m.out_len) continue;
            if (which == 0) { if (!inflate_fast(d + m.in, m.in_len, f.size() - m.in - m.in_len, a.data(), m.out_len)) refused++; }
            else if (!zlib_inflate(d + m.in, m.in_len, b, m.out_len)) refused++;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%-12s %8.1f MB/s  (%zu members, %.1f MB, %zu refused)\n", which ? "zlib" : "inflate_fast", total / s / 1e6, ms.size(), total / 1e6, refused);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !std::strcmp(argv[1], "time")) return time_file(argv[2]);
    return self_test();
}
