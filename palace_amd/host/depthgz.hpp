// The per-base depth file of the driver's depth stage (palace:541-545):
//     samtools depth -@ T <bam> > <bam>.depth        one line `contig <TAB> 1-based position <TAB> depth` per covered position
//     bgzip -@ T -f <bam>.depth                       BGZF: gzip members of at most 0xff00 bytes of text each + the 28-byte EOF member
//     tabix -f -s 1 -b 2 -e 2 <bam>.depth.gz          <bam>.depth.gz.tbi: bins + 16 kb linear index of virtual file offsets, itself BGZF
// as one pass over the match segments the BAM loader collected (`bamdepth --depth-gz <out.gz> <bam>`), host only.  The reference's
// step 5 reads the result through pysam's TabixFile.fetch(contig) (create_sub_graph.py:206-234).
//
// Written from the SAM/BAM specification (section 4.1, BGZF) and the tabix format description (tabix.pdf: the TBI layout; bins by
// reg2bin with min_shift 14 and 5 levels as in section 5.3 of the SAM specification).  Neither samtools, bgzip nor tabix exist in the
// build image, so the files are NOT byte-identical to theirs by construction: the text inside is (samtools depth semantics as in
// oracle/graph_oracle.cpp orc_depth_mean), the members are cut at the same 0xff00 bytes, but the DEFLATE bytes depend on the zlib
// build and level, and the index keeps one leaf bin per 16 kb window where tabix merges sparsely filled bins into their parents
// (both are valid indexes of the same file; a reader looks through every level's bins).
//   record of the index = one text line: interval [pos - 1, pos) of its contig (tbx.c: -b and -e naming the same column)
//   bin of a record     = 4681 + ((pos - 1) >> 14)           (leaf level; a 1-base interval never spans two windows)
//   chunk of a bin      = [virtual offset of its first line, virtual offset behind its last line)   (lines are sorted: one chunk per bin)
//   linear index        = per 16 kb window the virtual offset of the first line that starts in it; empty windows take the next
//                         window's value (hts_idx_finish fills backwards)
//   pseudo-bin 37450    = {[first, behind last) of the contig's lines, (number of lines, 0)}, as htslib writes it
//   virtual offset      = (file offset of the BGZF member << 16) | offset of the byte inside the member's text
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bam.hpp"

namespace palace_host {

constexpr size_t kBgzfText = 0xff00;                   // bytes of text per member (bgzf.h BGZF_BLOCK_SIZE)

inline void put_le(std::vector<uint8_t> &b, uint64_t v, int bytes) { for (int k = 0; k < bytes; k++) b.push_back(static_cast<uint8_t>(v >> (8 * k))); }

// one BGZF member for `n` (<= 0xff00) bytes of text; throws when zlib fails or the member would not fit 64 KiB
inline void bgzf_member(const uint8_t *text, size_t n, int level, std::vector<uint8_t> &out)
{
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    const size_t at = out.size();
    out.insert(out.end(), head, head + 16);
    out.push_back(0); out.push_back(0);                                      // BSIZE, patched below
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    const size_t bound = deflateBound(&zs, static_cast<uLong>(n));
    out.resize(at + 18 + bound);
    zs.next_in = const_cast<Bytef *>(text); zs.avail_in = static_cast<uInt>(n);
    zs.next_out = out.data() + at + 18; zs.avail_out = static_cast<uInt>(bound);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = bound - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("deflate failed");
    out.resize(at + 18 + clen);
    put_le(out, crc32(crc32(0L, Z_NULL, 0), text, static_cast<uInt>(n)), 4);
    put_le(out, n, 4);
    const size_t total = out.size() - at;
    if (total > 0x10000) throw std::runtime_error("BGZF member larger than 64 KiB");
    out[at + 16] = static_cast<uint8_t>(total - 1); out[at + 17] = static_cast<uint8_t>((total - 1) >> 8);
}

inline const uint8_t *bgzf_eof_member()
{
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return eof;
}

// A BGZF file written from a stream of text: members of exactly kBgzfText bytes (the last one shorter), compressed on `threads`
// threads a batch at a time, the EOF member at close.  member_off[k] = file offset of member k (member n_members = the EOF member).
struct BgzfTextWriter {
    FILE *f = nullptr;
    int level, threads;
    std::vector<uint8_t> pending;                    // text not yet written (always < one batch)
    std::vector<uint64_t> member_off;
    uint64_t text_bytes = 0, file_bytes = 0;
    static constexpr size_t kBatch = 256;            // members per batch

    BgzfTextWriter(const std::string &path, int level_, int threads_) : level(level_), threads(std::max(1, threads_))
    {
        f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + path + " for writing");
        pending.reserve(kBatch * kBgzfText + (1u << 20));
    }
    ~BgzfTextWriter() { if (f) std::fclose(f); }
    uint64_t tell_text() const { return text_bytes; }                 // bytes of text handed over so far
    void write(const char *s, size_t n)
    {
        pending.insert(pending.end(), s, s + n);
        text_bytes += n;
        if (pending.size() >= kBatch * kBgzfText) flush(false);
    }
    void flush(bool all)
    {
        const size_t n_full = pending.size() / kBgzfText, n_mem = all ? (pending.size() + kBgzfText - 1) / kBgzfText : n_full;
        if (n_mem == 0) return;
        std::vector<std::vector<uint8_t>> z(n_mem);
        std::vector<std::string> err(static_cast<size_t>(threads));
        auto work = [&](int t) {
            try {
                for (size_t k = static_cast<size_t>(t); k < n_mem; k += static_cast<size_t>(threads))
                    bgzf_member(pending.data() + k * kBgzfText, std::min(kBgzfText, pending.size() - k * kBgzfText), level, z[k]);
            } catch (const std::exception &e) { err[static_cast<size_t>(t)] = e.what(); }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        for (auto &e : err) if (!e.empty()) throw std::runtime_error(e);
        for (size_t k = 0; k < n_mem; k++) {
            member_off.push_back(file_bytes);
            if (std::fwrite(z[k].data(), 1, z[k].size(), f) != z[k].size()) throw std::runtime_error("write failed");
            file_bytes += z[k].size();
        }
        pending.erase(pending.begin(), pending.begin() + static_cast<std::ptrdiff_t>(std::min(pending.size(), n_mem * kBgzfText)));
    }
    void close()
    {
        flush(true);
        member_off.push_back(file_bytes);                               // the EOF member
        if (std::fwrite(bgzf_eof_member(), 1, 28, f) != 28) throw std::runtime_error("write failed");
        file_bytes += 28;
        if (std::fclose(f) != 0) { f = nullptr; throw std::runtime_error("close failed"); }
        f = nullptr;
    }
    // virtual offset of text byte `abs` (after close): behind the last byte of a file whose last member is short it stays inside that
    // member, as bgzf_tell reports it before the member is flushed
    uint64_t voffset(uint64_t abs) const
    {
        // (abs == text_bytes: the short last member has index text_bytes / kBgzfText; after a full last member that index is the EOF member's)
        return (member_off[static_cast<size_t>(abs / kBgzfText)] << 16) | (abs % kBgzfText);
    }
};

struct TbiBin { uint32_t bin; uint64_t beg, end; };                      // one chunk per bin (text offsets until the file is closed)
struct TbiRef { std::string name; std::vector<TbiBin> bins; std::vector<uint64_t> ioff; uint64_t off_beg = 0, off_end = 0, n_lines = 0; };

// the .tbi of a file indexed with `-s 1 -b 2 -e 2`, refs in file order; offsets are converted with w.voffset
inline void write_tbi(const std::string &path, const std::vector<TbiRef> &refs, const BgzfTextWriter &w)
{
    std::vector<uint8_t> t;
    t.insert(t.end(), {'T', 'B', 'I', 1});
    put_le(t, refs.size(), 4);
    put_le(t, 0, 4);                    // format: generic
    put_le(t, 1, 4); put_le(t, 2, 4); put_le(t, 2, 4);      // col_seq, col_beg, col_end
    put_le(t, '#', 4); put_le(t, 0, 4);                     // meta, skip
    size_t l_nm = 0;
    for (auto &r : refs) l_nm += r.name.size() + 1;
    put_le(t, l_nm, 4);
    for (auto &r : refs) { t.insert(t.end(), r.name.begin(), r.name.end()); t.push_back(0); }
    for (auto &r : refs) {
        put_le(t, r.bins.size() + 1, 4);                    // + the pseudo-bin
        for (auto &b : r.bins) {
            put_le(t, b.bin, 4); put_le(t, 1, 4);
            put_le(t, w.voffset(b.beg), 8); put_le(t, w.voffset(b.end), 8);
        }
        put_le(t, 37450, 4); put_le(t, 2, 4);
        put_le(t, w.voffset(r.off_beg), 8); put_le(t, w.voffset(r.off_end), 8);
        put_le(t, r.n_lines, 8); put_le(t, 0, 8);
        put_le(t, r.ioff.size(), 4);
        for (uint64_t o : r.ioff) put_le(t, w.voffset(o), 8);
    }
    put_le(t, 0, 8);                    // n_no_coor
    BgzfTextWriter out(path, 6, 1);
    out.write(reinterpret_cast<const char *>(t.data()), t.size());
    out.close();
}

struct DepthGzResult { uint64_t sum = 0, lines = 0, text_bytes = 0, file_bytes = 0; };

// Writes <gz_path> and <gz_path>.tbi from the match segments of `c` (load_bam with want_match_segments).  Contigs in header
// order (the order of a coordinate-sorted BAM); a contig without a covered position has no line and no entry in the index.
inline DepthGzResult write_depth_gz(const BamColumns &c, const std::string &gz_path, int threads, int level = 6)
{
    const size_t nt = c.target_len.size(), ns = c.mseg_tid.size();
    // segments grouped by contig (counting sort; their order inside a contig does not matter)
    std::vector<uint64_t> first(nt + 1, 0);
    for (size_t k = 0; k < ns; k++) {
        const int32_t t = c.mseg_tid[k];
        if (t >= 0 && static_cast<size_t>(t) < nt) first[static_cast<size_t>(t) + 1]++;
    }
    for (size_t t = 0; t < nt; t++) first[t + 1] += first[t];
    std::vector<uint32_t> order(first[nt]);
    {
        std::vector<uint64_t> cur(first.begin(), first.end() - 1);
        for (size_t k = 0; k < ns; k++) {
            const int32_t t = c.mseg_tid[k];
            if (t >= 0 && static_cast<size_t>(t) < nt) order[cur[static_cast<size_t>(t)]++] = static_cast<uint32_t>(k);
        }
    }
    BgzfTextWriter w(gz_path, level, threads);
    std::vector<TbiRef> refs;
    DepthGzResult res;
    std::vector<int32_t> diff;
    std::string line;
    for (size_t t = 0; t < nt; t++) {
        if (first[t + 1] == first[t]) continue;
        const int64_t L = std::max(0, c.target_len[t]);
        diff.assign(static_cast<size_t>(L) + 1, 0);
        bool any = false;
        for (uint64_t q = first[t]; q < first[t + 1]; q++) {
            const int64_t p = c.mseg_pos[order[q]], e = std::min<int64_t>(L, p + c.mseg_len[order[q]]);
            if (p < 0 || p >= e) continue;                                 // (outside the contig: samtools counts nothing there)
            diff[static_cast<size_t>(p)]++; diff[static_cast<size_t>(e)]--;
            any = true;
        }
        if (!any) continue;
        TbiRef r;
        r.name = c.target_name[t];
        r.off_beg = w.tell_text();
        const std::string prefix = r.name + "\t";
        int64_t depth = 0;
        for (int64_t p = 0; p < L; p++) {
            depth += diff[static_cast<size_t>(p)];
            if (depth <= 0) continue;
            const uint64_t at = w.tell_text();
            const uint32_t win = static_cast<uint32_t>(p >> 14);
            if (r.bins.empty() || r.bins.back().bin != 4681u + win) r.bins.push_back(TbiBin{4681u + win, at, at});
            if (r.ioff.size() <= win) r.ioff.resize(static_cast<size_t>(win) + 1, ~0ull);
            if (r.ioff[win] == ~0ull) r.ioff[win] = at;
            char num[48], *q = num + sizeof num;                           // "<pos>\t<depth>\n", digits from the back
            *--q = '\n';
            for (uint64_t v = static_cast<uint64_t>(depth); ; v /= 10) { *--q = static_cast<char>('0' + v % 10); if (v < 10) break; }
            *--q = '\t';
            for (uint64_t v = static_cast<uint64_t>(p + 1); ; v /= 10) { *--q = static_cast<char>('0' + v % 10); if (v < 10) break; }
            line.assign(prefix); line.append(q, static_cast<size_t>(num + sizeof num - q));
            w.write(line.data(), line.size());
            r.bins.back().end = w.tell_text();
            r.n_lines++;
            res.sum += static_cast<uint64_t>(depth);
        }
        r.off_end = w.tell_text();
        for (size_t k = r.ioff.size(); k-- > 1;)                          // empty windows take the next window's offset
            if (r.ioff[k - 1] == ~0ull) r.ioff[k - 1] = r.ioff[k];
        res.lines += r.n_lines;
        refs.push_back(std::move(r));
    }
    w.close();
    res.text_bytes = w.text_bytes; res.file_bytes = w.file_bytes;
    write_tbi(gz_path + ".tbi", refs, w);
    return res;
}

}  // namespace palace_host
