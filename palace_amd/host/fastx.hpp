// FASTA / FASTQ text -> packed sequences (1 B/base ASCII, concatenated, int64 offsets), with the
// line semantics the reference's std::getline loops have (extract_ref.cpp:686-756, 940-1004):
// a line ends at '\n' only ('\r' stays in the sequence and is an invalid base), a final line
// without '\n' still counts, FASTQ sequence lines are those with (0-based line index) % 4 == 1.
#pragma once
#include <algorithm>
#include <cctype>
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
    std::vector<std::string> names;      // FASTA only: get_read_ID view of the header (extract_ref.cpp:246-254)
    std::vector<std::string> ids;        // FASTA only: header up to the first white space (what `samtools faidx` lists)
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
inline void parse_fasta(const char *txt, size_t N, SeqSet &out)
{
    out.bases.clear(); out.bases.reserve(N);
    out.offsets.assign(1, 0);
    // text ahead of the first header belongs to an implicit record "start" with ordinal 0
    // (extract_ref.cpp:672, 688: pre_name = "start", ref_index = 0)
    out.names.assign(1, "start"); out.ordinal.assign(1, 0); out.ids.assign(1, "");
    bool open = true;
    int64_t rec = 0;
    for (size_t p = 0; p < N;) {
        const void *nl = std::memchr(txt + p, '\n', N - p);
        size_t e = nl ? static_cast<const char *>(nl) - txt : N;
        if (e > p && txt[p] == '>') {
            if (open) out.offsets.push_back(static_cast<int64_t>(out.bases.size()));
            out.names.push_back(fasta_name(std::string(txt + p, e - p)));
            {
                size_t w = p + 1;
                while (w < e && !std::isspace(static_cast<unsigned char>(txt[w]))) w++;
                out.ids.emplace_back(txt + p + 1, w - p - 1);
            }
            out.ordinal.push_back(++rec);
            open = true;
        } else if (open) {
            out.bases.insert(out.bases.end(), txt + p, txt + e);
        }
        p = nl ? e + 1 : N;
    }
    if (open) out.offsets.push_back(static_cast<int64_t>(out.bases.size()));
}
inline void parse_fasta(const std::vector<char> &txt, SeqSet &out) { parse_fasta(txt.data(), txt.size(), out); }

// The same on threads: the text is cut at header lines, every part parsed on its own, the pieces joined (the copy of the bases
// into their final place runs on the threads as well).  A 200-Mbase DB: 190 ms -> ~40 ms on 16 threads.
inline void parse_fasta_mt(const char *txt, size_t N, SeqSet &out, int threads)
{
    if (threads <= 1 || N < (4u << 20)) { parse_fasta(txt, N, out); return; }
    std::vector<size_t> cut{0};
    for (int k = 1; k < threads; k++) {
        size_t p = std::max(cut.back(), N / static_cast<size_t>(threads) * static_cast<size_t>(k));
        size_t at = N;
        while (p < N) {                                     // next line that starts with '>'
            const void *nl = std::memchr(txt + p, '\n', N - p);
            if (!nl) break;
            p = static_cast<size_t>(static_cast<const char *>(nl) - txt) + 1;
            if (p < N && txt[p] == '>') { at = p; break; }
        }
        if (at >= N) break;
        if (at > cut.back()) cut.push_back(at);
    }
    cut.push_back(N);
    const size_t n_parts = cut.size() - 1;
    std::vector<SeqSet> piece(n_parts);
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < n_parts; k++) pool.emplace_back([&, k] { parse_fasta(txt + cut[k], cut[k + 1] - cut[k], piece[k]); });
        for (auto &t : pool) t.join();
    }
    // every piece begins with the implicit record "start" (text ahead of its first header): real in piece 0, empty in the others
    size_t n_rec = 0, n_bases = 0;
    std::vector<size_t> rec0(n_parts), base0(n_parts);
    for (size_t k = 0; k < n_parts; k++) {
        rec0[k] = n_rec; base0[k] = n_bases;
        n_rec += static_cast<size_t>(piece[k].n()) - (k ? 1 : 0);
        n_bases += piece[k].bases.size();
    }
    out.bases.resize(n_bases);
    out.offsets.assign(n_rec + 1, 0);
    out.names.resize(n_rec); out.ids.resize(n_rec); out.ordinal.resize(n_rec);
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < n_parts; k++)
            pool.emplace_back([&, k] {
                const SeqSet &pc = piece[k];
                if (!pc.bases.empty()) std::memcpy(out.bases.data() + base0[k], pc.bases.data(), pc.bases.size());
                const size_t skip = k ? 1 : 0;
                for (size_t i = skip; i < static_cast<size_t>(pc.n()); i++) {
                    const size_t r = rec0[k] + i - skip;
                    out.offsets[r + 1] = static_cast<int64_t>(base0[k]) + pc.offsets[i + 1];
                    out.names[r] = pc.names[i]; out.ids[r] = pc.ids[i];
                    out.ordinal[r] = static_cast<int64_t>(r);           // record 0 is "start" (ordinal 0), header n has ordinal n
                }
            });
        for (auto &t : pool) t.join();
    }
}

}  // namespace palace_host

// ------------------------------------------------------------------------------------------------
// Streaming FASTQ ingest for the executables: the file is mapped, cut into parts of a few MiB at line
// boundaries, and every part is scanned by a pool of threads -- first for its line / sequence-line / sequence-byte counts
// under each of the four possible line phases (so the parts can be placed without a serial pass), then to copy its
// sequence lines to their final place in a staging buffer.  Same getline semantics as parse_fastq above.
// ------------------------------------------------------------------------------------------------
#include <emmintrin.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <functional>

namespace palace_host {

struct MappedText {
    const char *data = nullptr;
    size_t size = 0;
    MappedText() = default;
    explicit MappedText(const std::string &path) { open(path); }
    void open(const std::string &path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot open " + path); }
        size = static_cast<size_t>(st.st_size);
        if (size) {
            void *m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("cannot read " + path); }
            data = static_cast<const char *>(m);
        }
        ::close(fd);
    }
    ~MappedText() { if (data) ::munmap(const_cast<char *>(data), size); }
    MappedText(const MappedText &) = delete;
    MappedText &operator=(const MappedText &) = delete;
};

// run f(i) for i in [0, n) on `threads` threads (dynamic hand-out, small n per call is fine)
inline void pool_for(size_t n, int threads, const std::function<void(size_t)> &f)
{
    threads = std::max(1, std::min<int>(threads, static_cast<int>(std::max<size_t>(1, n))));
    if (threads == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (auto &th : pool) th.join();
}

struct FastqPart {
    size_t a = 0, b = 0;                 // text range, starts at a line start
    int64_t lines = 0;
    int64_t n_by_phase[4] = {0, 0, 0, 0}, bytes_by_phase[4] = {0, 0, 0, 0};   // lines with (local index & 3) == c
    int64_t line0 = 0, read0 = 0, byte0 = 0;                                  // after place(): global bases of this part
    int64_t n_seq() const { return n_by_phase[(1 - line0) & 3]; }
    int64_t seq_bytes() const { return bytes_by_phase[(1 - line0) & 3]; }
};

struct FastqPlan {
    const MappedText *txt = nullptr;
    std::vector<FastqPart> parts;
    int64_t n_reads = 0, n_bases = 0;
};

// pass 1: cut + count.  part_bytes: target text size of a part.
inline void plan_fastq(const MappedText &txt, int threads, FastqPlan &plan, size_t part_bytes = 4u << 20)
{
    plan.txt = &txt;
    plan.parts.clear();
    const size_t N = txt.size;
    for (size_t p = 0; p < N;) {
        size_t q = std::min(N, p + part_bytes);
        if (q < N) {
            const void *nl = std::memchr(txt.data + q, '\n', N - q);
            q = nl ? static_cast<const char *>(nl) - txt.data + 1 : N;
        }
        FastqPart part;
        part.a = p; part.b = q;
        plan.parts.push_back(part);
        p = q;
    }
    pool_for(plan.parts.size(), threads, [&](size_t i) {
        FastqPart &pt = plan.parts[i];
        int64_t l = 0;
        for (size_t p = pt.a; p < pt.b; l++) {
            const void *nl = std::memchr(txt.data + p, '\n', pt.b - p);
            const size_t e = nl ? static_cast<const char *>(nl) - txt.data : pt.b;
            pt.n_by_phase[l & 3]++;
            pt.bytes_by_phase[l & 3] += static_cast<int64_t>(e - p);
            p = nl ? e + 1 : pt.b;
        }
        pt.lines = l;
    });
    int64_t line = 0, read = 0, byte = 0;
    for (FastqPart &pt : plan.parts) {
        pt.line0 = line; pt.read0 = read; pt.byte0 = byte;
        line += pt.lines; read += pt.n_seq(); byte += pt.seq_bytes();
    }
    plan.n_reads = read; plan.n_bases = byte;
}

// pass 2 for one part: its sequence lines go to bases_dst + (byte0 - bytes_base); offsets_dst[read0 + i + 1] gets the
// GLOBAL end offset (offset_base + byte0 + ...) of its i-th read.
inline void extract_fastq_part(const FastqPlan &plan, size_t i, uint8_t *bases_dst, int64_t bytes_base, int64_t *offsets_dst,
                               int64_t offset_base)
{
    const FastqPart &pt = plan.parts[i];
    const char *d = plan.txt->data;
    uint8_t *w = bases_dst + (pt.byte0 - bytes_base);
    int64_t at = offset_base + pt.byte0, r = pt.read0, l = pt.line0;
    for (size_t p = pt.a; p < pt.b; l++) {
        const void *nl = std::memchr(d + p, '\n', pt.b - p);
        const size_t e = nl ? static_cast<const char *>(nl) - d : pt.b;
        if ((l & 3) == 1) {
            std::memcpy(w, d + p, e - p);
            w += e - p;
            at += static_cast<int64_t>(e - p);
            offsets_dst[++r] = at;
        }
        p = nl ? e + 1 : pt.b;
    }
}


// ---- the packed form of a part (include/palace_hip.h: palace_eref_count_reads_packed) --------------------------------------
// Pass 2 without the ASCII copy: the sequence lines of a part become bits of three streams -- P0 = {A,T}, P1 = {A,C}, and
// U = "a 32-mer is counted here" -- which is what the device kernels read; 3 bits per base go over PCIe instead of 8.  Parts
// start on multiples of 64 positions (the gaps are positions of no read, U = 0), so no two threads share a word.
//
// Sixteen bases per step (SSE2, the x86-64 baseline): four byte-wise compares against A, C, G, T on the case-folded bytes,
// and a movemask turns each class into 16 bits.
inline int64_t packed_span(int64_t seq_bytes) { return (seq_bytes + 63) / 64 * 64; }     // positions a part occupies

// Words [0, packed_span / 64) of p0 / p1 / u are written, all of them.  keep: one byte per read of the whole set (E3
// subsampling, indexed read_base + the part's read numbers) or null.
inline void pack_fastq_part(const FastqPlan &plan, size_t i, uint64_t *p0, uint64_t *p1, uint64_t *u, const uint8_t *keep, int64_t read_base)
{
    const FastqPart &pt = plan.parts[i];
    const char *d = plan.txt->data;
    const size_t nw = static_cast<size_t>(packed_span(pt.seq_bytes()) / 64);
    std::vector<uint64_t> ok(nw + 1, 0), en(nw + 1, 0);
    uint64_t a0 = 0, a1 = 0, ak = 0;                       // accumulators of the word being filled
    int fill = 0;
    size_t w = 0;
    int64_t at = 0, r = pt.read0, l = pt.line0;            // position within the part
    auto put = [&](uint64_t b0, uint64_t b1, uint64_t bk, int n) {          // n <= 16 bits each
        a0 |= b0 << fill; a1 |= b1 << fill; ak |= bk << fill;
        fill += n;
        if (fill >= 64) {
            p0[w] = a0; p1[w] = a1; ok[w] = ak; w++;
            fill -= 64;
            const int used = n - fill;                     // bits of this put that went into the finished word
            a0 = fill ? b0 >> used : 0; a1 = fill ? b1 >> used : 0; ak = fill ? bk >> used : 0;
        }
    };
    for (size_t p = pt.a; p < pt.b; l++) {
        const void *nl = std::memchr(d + p, '\n', pt.b - p);
        const size_t e = nl ? static_cast<const char *>(nl) - d : pt.b;
        if ((l & 3) == 1) {
            const bool counted = !keep || keep[read_base + r];
            r++;
            const size_t len = e - p;
            const __m128i fold = _mm_set1_epi8(static_cast<char>(0xDF)), cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'),
                          cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
            for (size_t q = 0; q < len; q += 16) {
                const int n = static_cast<int>(std::min<size_t>(16, len - q));
                __m128i x;
                if (n == 16) x = _mm_loadu_si128(reinterpret_cast<const __m128i *>(d + p + q));
                else {                                                           // the line's tail: never read past it (the mapping may end there)
                    alignas(16) char tail[16] = {0};
                    std::memcpy(tail, d + p + q, static_cast<size_t>(n));
                    x = _mm_load_si128(reinterpret_cast<const __m128i *>(tail));
                }
                x = _mm_and_si128(x, fold);
                const __m128i isA = _mm_cmpeq_epi8(x, cA), isC = _mm_cmpeq_epi8(x, cC), isG = _mm_cmpeq_epi8(x, cG), isT = _mm_cmpeq_epi8(x, cT);
                const uint64_t at_ = static_cast<uint64_t>(_mm_movemask_epi8(_mm_or_si128(isA, isT))),
                               ac_ = static_cast<uint64_t>(_mm_movemask_epi8(_mm_or_si128(isA, isC))),
                               gt_ = static_cast<uint64_t>(_mm_movemask_epi8(_mm_or_si128(isG, isT)));
                put(at_, ac_, counted ? ((ac_ | gt_) & ((1ull << n) - 1)) : 0, n);      // (zero bytes of a tail match nothing anyway)
            }
            at += static_cast<int64_t>(len);
            if (len) en[static_cast<size_t>((at - 1) >> 6)] |= 1ull << ((at - 1) & 63);
        }
        p = nl ? e + 1 : pt.b;
    }
    if (w < nw) { p0[w] = a0; p1[w] = a1; ok[w] = ak; w++; }
    for (; w < nw; w++) { p0[w] = 0; p1[w] = 0; }
    // U: bit t = the 32 positions from t on are bases of a counted read, and no read ends among the first 31 of them
    for (size_t k = 0; k < nw; k++) {
        unsigned __int128 a = (static_cast<unsigned __int128>(ok[k + 1]) << 64) | ok[k];
        for (int sft = 1; sft < 32; sft <<= 1) a &= a >> sft;
        unsigned __int128 x = (static_cast<unsigned __int128>(en[k + 1]) << 64) | en[k];
        for (int sft = 1; sft < 16; sft <<= 1) x |= x >> sft;
        x |= x >> 15;
        u[k] = static_cast<uint64_t>(a) & ~static_cast<uint64_t>(x);
    }
}

}  // namespace palace_host
