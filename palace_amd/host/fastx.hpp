// FASTA / FASTQ text -> packed sequences (1 B/base ASCII, concatenated, int64 offsets), with the
// line semantics the reference's std::getline loops have (extract_ref.cpp:686-756, 940-1004):
// a line ends at '\n' only ('\r' stays in the sequence and is an invalid base), a final line
// without '\n' still counts, FASTQ sequence lines are those with (0-based line index) % 4 == 1.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace palace_host {

struct SeqSet {
    std::vector<uint8_t> bases;
    std::vector<int64_t> offsets{0};
    std::vector<std::string> names;      // FASTA only
    std::vector<int64_t> ordinal;        // FASTA only: 1-based record number in the file
    int64_t n() const { return static_cast<int64_t>(offsets.size()) - 1; }
    int64_t len(int64_t i) const { return offsets[i + 1] - offsets[i]; }
};

inline std::vector<char> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<char> buf(static_cast<size_t>(n));
    if (n && !f.read(buf.data(), n)) throw std::runtime_error("cannot read " + path);
    return buf;
}

// FASTQ: sequence lines only.  Chunked over threads at line boundaries.
inline void parse_fastq(const std::vector<char> &txt, int threads, SeqSet &out)
{
    const size_t N = txt.size();
    threads = std::max(1, threads);
    std::vector<size_t> cut(threads + 1, N);
    cut[0] = 0;
    for (int t = 1; t < threads; t++) {
        size_t p = std::max(cut[t - 1], N * t / threads);
        const void *nl = p < N ? std::memchr(txt.data() + p, '\n', N - p) : nullptr;
        cut[t] = nl ? static_cast<const char *>(nl) - txt.data() + 1 : N;
    }
    std::vector<int64_t> lines(threads + 1, 0);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                int64_t c = 0;
                for (size_t p = cut[t]; p < cut[t + 1];) {
                    const void *nl = std::memchr(txt.data() + p, '\n', cut[t + 1] - p);
                    c++;
                    p = nl ? static_cast<const char *>(nl) - txt.data() + 1 : cut[t + 1];
                }
                lines[t + 1] = c;
            });
        for (auto &th : pool) th.join();
    }
    for (int t = 0; t < threads; t++) lines[t + 1] += lines[t];
    std::vector<std::vector<uint8_t>> pb(threads);
    std::vector<std::vector<int64_t>> pl(threads);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                int64_t li = lines[t];
                for (size_t p = cut[t]; p < cut[t + 1]; li++) {
                    const void *nl = std::memchr(txt.data() + p, '\n', cut[t + 1] - p);
                    size_t e = nl ? static_cast<const char *>(nl) - txt.data() : cut[t + 1];
                    if ((li & 3) == 1) {
                        pb[t].insert(pb[t].end(), txt.data() + p, txt.data() + e);
                        pl[t].push_back(static_cast<int64_t>(e - p));
                    }
                    p = nl ? e + 1 : cut[t + 1];
                }
            });
        for (auto &th : pool) th.join();
    }
    size_t tb = 0, tn = 0;
    for (int t = 0; t < threads; t++) { tb += pb[t].size(); tn += pl[t].size(); }
    out.bases.clear(); out.bases.reserve(tb + 64);
    out.offsets.assign(1, 0); out.offsets.reserve(tn + 1);
    for (int t = 0; t < threads; t++) {
        out.bases.insert(out.bases.end(), pb[t].begin(), pb[t].end());
        for (int64_t l : pl[t]) out.offsets.push_back(out.offsets.back() + l);
    }
}

// get_read_ID (extract_ref.cpp:246-254) on a header line, then drop the leading '>' (:689)
inline std::string fasta_name(const std::string &line)
{
    std::string s = line.substr(0, line.find('/'));
    s = s.substr(0, s.find(' '));
    s = s.substr(0, s.find('\t'));
    return s.empty() ? s : s.substr(1);
}

// FASTA: every '>' record (also empty / short ones: the caller applies the len > 32 rule).
inline void parse_fasta(const std::vector<char> &txt, SeqSet &out)
{
    const size_t N = txt.size();
    out.bases.clear(); out.bases.reserve(N);
    out.offsets.assign(1, 0);
    // text ahead of the first header belongs to an implicit record "start" with ordinal 0
    // (extract_ref.cpp:672, 688: pre_name = "start", ref_index = 0)
    out.names.assign(1, "start"); out.ordinal.assign(1, 0);
    bool open = true;
    int64_t rec = 0;
    for (size_t p = 0; p < N;) {
        const void *nl = std::memchr(txt.data() + p, '\n', N - p);
        size_t e = nl ? static_cast<const char *>(nl) - txt.data() : N;
        if (e > p && txt[p] == '>') {
            if (open) out.offsets.push_back(static_cast<int64_t>(out.bases.size()));
            out.names.push_back(fasta_name(std::string(txt.data() + p, e - p)));
            out.ordinal.push_back(++rec);
            open = true;
        } else if (open) {
            out.bases.insert(out.bases.end(), txt.data() + p, txt.data() + e);
        }
        p = nl ? e + 1 : N;
    }
    if (open) out.offsets.push_back(static_cast<int64_t>(out.bases.size()));
}

}  // namespace palace_host
