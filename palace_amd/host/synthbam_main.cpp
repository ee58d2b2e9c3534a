// synthbam -- writes a synthetic BAM from decoded columns (bench / test tooling, not part of the drop-in surface).
// bench.py generates its BAM-side sample as columns in HBM; the end-to-end leg needs the same sample as the FILE the
// real pipeline hands to generateGraph (palace:557-560), and the Python writer (palace_amd/synth.write_bam) is far
// too slow for 6.7 M records.  Written against the SAM/BAM specification (sections 4.1, 4.2), multi-threaded BGZF.
//
//   synthbam <dir> <out.bam> <threads> [level]
// <dir> holds little-endian raw arrays, one file per column, n records each:
//   tid.i32 pos.i32 mtid.i32 mpos.i32 nm.i32 ref_len.i32 clip_e.i32 sa_off.i32 (n+1) flag.u16 mapq.u8 qkey.u64
//   sa.i32 (rows of 8: tid2 pos2 mapq2 nm2 clip_s2 clip_e2 len2 rev2)   targets.tsv (name \t length per line)
// Record i: qname "q<qkey & 2^48-1 in hex>", CIGAR <ref_len>M[<clip_e>S], 150 pseudo-random bases + binned qualities,
// NM:C, SA:Z when it has items.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

template <class T>
std::vector<T> slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "synthbam: cannot open " << path << "\n"; std::exit(1); }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v(static_cast<size_t>(n) / sizeof(T));
    f.read(reinterpret_cast<char *>(v.data()), n);
    return v;
}

void put32(std::vector<uint8_t> &b, uint32_t v) { for (int k = 0; k < 4; k++) b.push_back(static_cast<uint8_t>(v >> (8 * k))); }
void put16(std::vector<uint8_t> &b, uint16_t v) { b.push_back(static_cast<uint8_t>(v)); b.push_back(static_cast<uint8_t>(v >> 8)); }

template <class F>
void parallel_for(size_t n, int threads, F f)
{
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([=] { f(n * t / threads, n * (t + 1) / threads, t); });
    for (auto &th : pool) th.join();
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 4) { std::cerr << "usage: synthbam <dir> <out.bam> <threads> [level]\n"; return 2; }
    const std::string dir = std::string(argv[1]) + "/";
    const int threads = std::max(1, std::atoi(argv[3])), level = argc > 4 ? std::atoi(argv[4]) : 1;
    auto tid = slurp<int32_t>(dir + "tid.i32"), pos = slurp<int32_t>(dir + "pos.i32"), mtid = slurp<int32_t>(dir + "mtid.i32"),
         mpos = slurp<int32_t>(dir + "mpos.i32"), nm = slurp<int32_t>(dir + "nm.i32"), ref_len = slurp<int32_t>(dir + "ref_len.i32"),
         clip_e = slurp<int32_t>(dir + "clip_e.i32"), sa_off = slurp<int32_t>(dir + "sa_off.i32"), sa = slurp<int32_t>(dir + "sa.i32");
    auto flag = slurp<uint16_t>(dir + "flag.u16");
    auto mapq = slurp<uint8_t>(dir + "mapq.u8");
    auto qkey = slurp<uint64_t>(dir + "qkey.u64");
    const size_t n = tid.size();
    std::vector<std::string> names;
    std::vector<int32_t> lens;
    {
        std::ifstream f(dir + "targets.tsv");
        std::string line;
        while (std::getline(f, line)) {
            const size_t t = line.find('\t');
            names.push_back(line.substr(0, t));
            lens.push_back(std::atoi(line.c_str() + t + 1));
        }
    }
    // header
    std::vector<std::vector<uint8_t>> part(static_cast<size_t>(threads) + 1);
    {
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (size_t i = 0; i < names.size(); i++) text += "@SQ\tSN:" + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
        auto &h = part[0];
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, static_cast<uint32_t>(text.size()));
        h.insert(h.end(), text.begin(), text.end());
        put32(h, static_cast<uint32_t>(names.size()));
        for (size_t i = 0; i < names.size(); i++) {
            put32(h, static_cast<uint32_t>(names[i].size() + 1));
            h.insert(h.end(), names[i].begin(), names[i].end());
            h.push_back(0);
            put32(h, static_cast<uint32_t>(lens[i]));
        }
    }
    // records, serialised per thread range
    parallel_for(n, threads, [&](size_t a, size_t b, int t) {
        auto &o = part[static_cast<size_t>(t) + 1];
        o.reserve((b - a) * 340);
        char buf[256];
        for (size_t i = a; i < b; i++) {
            const int l_seq = 150;
            const int nl = std::snprintf(buf, sizeof buf, "q%llx", static_cast<unsigned long long>(qkey[i] & 0xffffffffffffull)) + 1;
            const int n_cig = clip_e[i] ? 2 : 1;
            const size_t at = o.size();
            put32(o, 0);                                           // block_size, patched below
            put32(o, static_cast<uint32_t>(tid[i])); put32(o, static_cast<uint32_t>(pos[i]));
            o.push_back(static_cast<uint8_t>(nl)); o.push_back(mapq[i]); put16(o, 4680);
            put16(o, static_cast<uint16_t>(n_cig)); put16(o, flag[i]); put32(o, l_seq);
            put32(o, static_cast<uint32_t>(mtid[i])); put32(o, static_cast<uint32_t>(mpos[i])); put32(o, 0);
            o.insert(o.end(), buf, buf + nl);
            if (clip_e[i]) { put32(o, (static_cast<uint32_t>(ref_len[i]) << 4) | 0); put32(o, (static_cast<uint32_t>(clip_e[i]) << 4) | 4); }
            else put32(o, (150u << 4) | 0);
            // pseudo-random packed bases and Illumina-like binned qualities, so that the file compresses about as a real BAM
            // does (constant bytes would make the BGZF inflate of the consumer unrealistically cheap)
            uint64_t h = qkey[i] * 0x9e3779b97f4a7c15ull + flag[i];
            for (int k = 0; k < (l_seq + 1) / 2; k++) {
                h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
                o.push_back(static_cast<uint8_t>((1u << (h & 3)) << 4 | (1u << ((h >> 2) & 3))));
            }
            static const uint8_t qbin[8] = {37, 37, 37, 37, 37, 25, 11, 2};
            for (int k = 0; k < l_seq; k++) {
                if ((k & 15) == 0) { h ^= h >> 31; h *= 0x94d049bb133111ebull; h ^= h >> 29; }
                o.push_back(qbin[(h >> (4 * (k & 15))) & 7]);
            }
            o.insert(o.end(), {'N', 'M', 'C', static_cast<uint8_t>(nm[i])});
            if (sa_off[i + 1] > sa_off[i]) {
                o.insert(o.end(), {'S', 'A', 'Z'});
                for (int32_t k = sa_off[i]; k < sa_off[i + 1]; k++) {
                    const int32_t *s = &sa[8 * static_cast<size_t>(k)];
                    const int m = std::snprintf(buf, sizeof buf, "%s,%d,%c,%dS%dM,%d,%d;", names[static_cast<size_t>(s[0])].c_str(), s[1],
                                                s[7] ? '-' : '+', s[4], s[6] - s[4], s[2], s[3]);
                    o.insert(o.end(), buf, buf + m);
                }
                o.push_back(0);
            }
            const uint32_t bs = static_cast<uint32_t>(o.size() - at - 4);
            std::memcpy(&o[at], &bs, 4);
        }
    });
    // BGZF blocks over the concatenation of the parts
    std::vector<size_t> start(part.size() + 1, 0);
    for (size_t k = 0; k < part.size(); k++) start[k + 1] = start[k] + part[k].size();
    const size_t total = start.back(), kBlock = 0xff00, n_blocks = (total + kBlock - 1) / kBlock;
    std::vector<std::vector<uint8_t>> z(n_blocks);
    std::atomic<bool> bad{false};
    parallel_for(n_blocks, threads, [&](size_t a, size_t b, int) {
        std::vector<uint8_t> raw(kBlock);
        z_stream zs{};
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = true; return; }
        for (size_t blk = a; blk < b; blk++) {
            const size_t lo = blk * kBlock, hi = std::min(total, lo + kBlock);
            size_t k = std::upper_bound(start.begin(), start.end(), lo) - start.begin() - 1, w = 0;
            for (size_t p = lo; p < hi;) {
                const size_t take = std::min(hi, start[k + 1]) - p;
                std::memcpy(raw.data() + w, part[k].data() + (p - start[k]), take);
                w += take; p += take; k++;
            }
            auto &out = z[blk];
            out.resize(18 + compressBound(static_cast<uLong>(w)) + 8);
            deflateReset(&zs);
            zs.next_in = raw.data(); zs.avail_in = static_cast<uInt>(w);
            zs.next_out = out.data() + 18; zs.avail_out = static_cast<uInt>(out.size() - 26);
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { bad = true; break; }
            const size_t clen = out.size() - 26 - zs.avail_out, bsize = 18 + clen + 8;
            const uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0,
                                      static_cast<uint8_t>((bsize - 1) & 255), static_cast<uint8_t>((bsize - 1) >> 8)};
            std::memcpy(out.data(), head, 18);
            const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0, nullptr, 0), raw.data(), static_cast<uInt>(w))), isz = static_cast<uint32_t>(w);
            std::memcpy(out.data() + 18 + clen, &crc, 4);
            std::memcpy(out.data() + 22 + clen, &isz, 4);
            out.resize(bsize);
            if (bsize > 65536) { bad = true; break; }
        }
        deflateEnd(&zs);
    });
    if (bad) { std::cerr << "synthbam: deflate failed\n"; return 1; }
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) { std::cerr << "synthbam: cannot write " << argv[2] << "\n"; return 1; }
    for (auto &b : z) std::fwrite(b.data(), 1, b.size(), f);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::fwrite(eof, 1, 28, f);
    std::fclose(f);
    return 0;
}
