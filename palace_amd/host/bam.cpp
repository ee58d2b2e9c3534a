#include "bam.hpp"
#include "inflate_fast.hpp"
#include "trace.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <memory>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace palace_host {

namespace {

uint32_t le32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
uint16_t le16(const uint8_t *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

// The compressed file, mapped read-only (no copy of it is made).
struct MappedFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    explicit MappedFile(const std::string &path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("Failed to open BAM " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("Failed to open BAM " + path); }
        size = static_cast<size_t>(st.st_size);
        if (size) {
            void *m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); throw std::runtime_error("Failed to read BAM " + path); }
            ::madvise(m, size, MADV_SEQUENTIAL);
            data = static_cast<const uint8_t *>(m);
        }
        ::close(fd);
    }
    ~MappedFile() { if (data) ::munmap(const_cast<uint8_t *>(data), size); }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

template <class F>
void parallel_for(size_t n, int threads, F f)
{
    threads = std::max(1, std::min<int>(threads, static_cast<int>(n ? n : 1)));
    if (threads == 1) { f(0, n, 0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        size_t a = n * t / threads, b = n * (t + 1) / threads;
        pool.emplace_back([=] { f(a, b, t); });
    }
    for (auto &th : pool) th.join();
}

// BGZF: gzip members with a BC extra subfield carrying the member size (SAM spec 4.1).  Every field that comes from the
// file is checked against the file before it is used: a member is 12 + XLEN header bytes, the deflate stream and an
// 8-byte trailer, BSIZE + 1 bytes in all, and holds at most 64 KiB of data.
struct Block { size_t in_off, in_len, out_off, out_len; };

std::vector<Block> index_bgzf(const uint8_t *file, size_t file_size, size_t *total_out)
{
    std::vector<Block> blocks;
    size_t p = 0, total = 0;
    while (p + 18 <= file_size) {
        const uint8_t *h = file + p;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("Failed to read BAM header");
        const size_t xlen = le16(h + 10), left = file_size - p;
        if (12 + xlen + 8 > left) throw std::runtime_error("truncated BGZF block");
        size_t q = 12, bsize = 0;
        while (q + 4 <= 12 + xlen) {
            const size_t slen = le16(h + q + 2);
            if (q + 4 + slen > 12 + xlen) throw std::runtime_error("malformed BGZF extra field");
            if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) bsize = static_cast<size_t>(le16(h + q + 4)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > left) throw std::runtime_error("truncated BGZF block");
        const size_t isize = le32(h + bsize - 4);
        if (isize > 65536) throw std::runtime_error("malformed BGZF block (ISIZE > 64 KiB)");
        blocks.push_back({p + 12 + xlen, bsize - xlen - 20, total, isize});
        total += isize;
        p += bsize;
    }
    *total_out = total;
    return blocks;
}

// One CIGAR as parseCigarReadInterval sees it (generate_graph.cpp:330-366): zero-length ops are
// dropped; clip_s = leading S, clip_e = trailing S when more than one op remains; len = query span.
struct ClipInfo { int32_t clip_s, clip_e, len; };

struct OpScan {
    int n_ops = 0; int first_len = 0, last_len = 0; char first = 0, last = 0; int32_t len = 0;
    void add(int n, char c)
    {
        if (n <= 0) return;
        if (!n_ops) { first = c; first_len = n; }
        last = c; last_len = n; n_ops++;
        if (c == 'M' || c == 'I' || c == 'S' || c == '=' || c == 'X') len += n;
    }
    ClipInfo done() const
    {
        ClipInfo r{0, 0, len};
        if (n_ops && first == 'S') r.clip_s = first_len;
        if (n_ops > 1 && last == 'S') r.clip_e = last_len;
        return r;
    }
};

ClipInfo clip_from_text(const char *s, size_t n)
{
    if (n == 0) return ClipInfo{-1, 0, 0};                 // empty text: interval stays [0,0] (:332)
    OpScan sc;
    int acc = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char ch = static_cast<unsigned char>(s[i]);
        if (std::isdigit(ch)) acc = acc * 10 + (ch - '0');
        else { sc.add(acc, static_cast<char>(ch)); acc = 0; }
    }
    return sc.done();
}

void trim_ws(const char *&b, const char *&e)
{
    while (b < e && std::isspace(static_cast<unsigned char>(*b))) ++b;
    while (e > b && std::isspace(static_cast<unsigned char>(e[-1]))) --e;
}

// parseSAItem (generate_graph.cpp:185-206): six comma fields must be extractable in getline's
// sense (a field exists iff at least one byte -- possibly just its delimiter -- is left).
bool parse_sa(const char *b, const char *e, const BamColumns &cols, int32_t own_tid, palace_sa_item &out)
{
    const char *fb[6], *fe[6];
    const char *p = b;
    for (int k = 0; k < 6; k++) {
        if (p >= e) return false;                          // nothing left: getline fails
        const char *c = static_cast<const char *>(std::memchr(p, ',', static_cast<size_t>(e - p)));
        fb[k] = p;
        fe[k] = c ? c : e;
        p = c ? c + 1 : e;
    }
    for (int k = 0; k < 6; k++) trim_ws(fb[k], fe[k]);
    if (fb[0] == fe[0] || fb[1] == fe[1]) return false;
    const std::string_view rname(fb[0], static_cast<size_t>(fe[0] - fb[0]));
    out.pos2 = std::atoi(std::string(fb[1], fe[1]).c_str());
    out.rev2 = (fe[2] - fb[2] == 1 && *fb[2] == '-') ? 1 : 0;
    ClipInfo ci = clip_from_text(fb[3], static_cast<size_t>(fe[3] - fb[3]));
    out.clip_s2 = ci.clip_s; out.clip_e2 = ci.clip_e; out.len2 = ci.len;
    out.mapq2 = std::atoi(std::string(fb[4], fe[4]).c_str());
    out.nm2 = std::atoi(std::string(fb[5], fe[5]).c_str());
    out.tid2 = -1;
    if (own_tid >= 0 && rname != cols.target_name[own_tid]) {       // r1 == r2 -> skip (:731)
        out.tid2 = cols.tid_of(rname);                                 // unknown name -> -1 -> skip (:733-734)
    }
    return true;
}

// size of one aux value at p (type byte already consumed); 0 on malformed input
size_t aux_size(uint8_t type, const uint8_t *p, const uint8_t *end)
{
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': { const void *z = std::memchr(p, 0, static_cast<size_t>(end - p)); return z ? static_cast<const uint8_t *>(z) - p + 1 : 0; }
    case 'B': {                                            // subtype, int32 count, count elements
        if (end - p < 5) return 0;
        size_t es;
        switch (p[0]) {
        case 'c': case 'C': es = 1; break;
        case 's': case 'S': es = 2; break;
        case 'i': case 'I': case 'f': es = 4; break;
        default: return 0;
        }
        return 5 + es * static_cast<size_t>(le32(p + 1));
    }
    default: return 0;
    }
}

}  // namespace

uint64_t name_key(const char *s, size_t n, uint64_t seed)
{
    uint64_t h = 0xcbf29ce484222325ull ^ (seed * 0x9e3779b97f4a7c15ull);
    for (size_t i = 0; i < n; i++) { h ^= static_cast<unsigned char>(s[i]); h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}

void RawBuf::release()
{
    if (p) ::munmap(p, mapped);
    p = nullptr; n = mapped = 0;
}
void RawBuf::alloc(size_t bytes)
{
    release();
    const size_t huge = size_t{2} << 20;
    mapped = (bytes + huge) / huge * huge;
    void *m = ::mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { mapped = 0; throw std::bad_alloc(); }
    p = static_cast<uint8_t *>(m);
    n = bytes;
    if (!std::getenv("PALACE_BAM_SMALL_PAGES")) ::madvise(p, mapped, MADV_HUGEPAGE);      // (a hint: refused or ignored, the stream is in 4 KiB pages as before)
}

void rekey(BamColumns &c, uint64_t seed)
{
    for (int64_t i = 0; i < c.n(); i++)
        c.qkey[i] = name_key(reinterpret_cast<const char *>(c.raw.data()) + c.qname_at[i], c.qname_len[i], seed);
}

// The loader as a pipeline (the reference streams the file through htslib, generate_graph.cpp:644):
//   inflate workers   take the BGZF members one by one off the front, so the inflated stream grows from the front; helpers
//                     (a device) take batches off the back;
//   the walker        one thread, started by load_bam_begin the moment the header's end is known: the record boundaries behind the
//                     inflate front (serial: a record's size is its first word), published in steps of 4 096 records.  Round 6: the
//                     inflate workers walk the stream SPECULATIVELY in segments of 1 MiB as the inflate front passes them (entry =
//                     the first offset from which four plausible records chain; from there the walker's own rules), and the walker
//                     adopts a segment's list from the moment its true position IS one of the list's record starts -- from the same
//                     start the walk is the same function of the stream, so the result is the serial walk's, record for record; a
//                     segment whose list the true walk never meets is walked serially as before;
//   the decode        the inflate workers, as they run out of members, take chunks of 32 768 walked records and write their
//                     columns (sized for the most records the stream can hold; pages behind the real ones are never touched);
//   load_bam_begin    returns as soon as the members that hold the header are there and the header is parsed -- the caller
//                     can start what depends on the target names only (name ranks, FASTG keys) beside the rest;
//   load_bam_finish   waits for the walker, helps with the chunks that are left, cuts the columns to the records found and
//                     puts the chunks' SA items and match segments behind one another.
struct BamLoad : BackMembers {
    std::unique_ptr<MappedFile> file;
    std::vector<Block> blocks;
    std::vector<BgzfMember> as_members;  // the same, in the helpers' type (filled when there are helpers)
    std::mutex claim_mu;                 // members [next_front, next_back) are nobody's yet
    size_t next_front = 0, next_back = 0;
    std::atomic<size_t> by_helpers{0};
    std::unique_ptr<std::atomic<uint8_t>[]> done;
    std::vector<std::thread> workers;
    std::atomic<bool> bad{false};
    BamColumns *c = nullptr;
    int threads = 1;
    size_t first_record = 0;             // offset of the first alignment record in the inflated stream
    int32_t n_ref = 0;

    // bytes of the inflated stream that are final: everything in front of the first member still missing.  Blocks
    // until at least `need` bytes are there (or everything that will ever come is).
    size_t wait_for(size_t need, size_t &ready_blocks)         // ready_blocks: the caller's own cursor (members [0, ready_blocks) seen inflated)
    {
        for (;;) {
            while (ready_blocks < blocks.size() && done[ready_blocks].load(std::memory_order_acquire)) ready_blocks++;
            const size_t have = ready_blocks < blocks.size() ? blocks[ready_blocks].out_off : c->raw.size();
            if (have >= need || ready_blocks == blocks.size()) return have;
            if (bad) throw std::runtime_error("BGZF inflate failed");
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    // the record walk (serial: a record's size is its first word), on a thread of its own from the moment the header's end is known
    std::thread walker;
    std::string walk_error;
    void walk();
    // ... and ahead of it, by the inflate workers: segment s = bytes [first_record + s * kSeg, + kSeg) of the inflated stream
    static constexpr size_t kSeg = size_t{1} << 20, kSegSlack = size_t{1} << 16;
    struct SpecSeg {
        std::vector<uint64_t> rec;       // record starts (offset of the byte behind block_size) found from the guessed entry on, ascending
        size_t exit = 0;                 // offset of the block_size word the walk of this segment ended in front of
        bool stopped = false;            // ... because the stream ends or is malformed THERE (the serial walk stops there too)
        std::atomic<int> state{0};       // 2: the list is complete
    };
    std::unique_ptr<SpecSeg[]> segs;
    std::atomic<size_t> n_segs_pub{SIZE_MAX};      // SIZE_MAX: the header's end (where the records begin) is not known yet
    std::atomic<size_t> next_seg{0};
    std::atomic<int> spec_drainers{0};
    size_t n_adopted = 0;                // records the walker took over from the segments' lists (the rest it walked itself)
    bool speculate = true;
    // bytes of the inflated stream that are final right now (never blocks)
    size_t ready_prefix(size_t &ready_blocks) const
    {
        while (ready_blocks < blocks.size() && done[ready_blocks].load(std::memory_order_acquire)) ready_blocks++;
        return ready_blocks < blocks.size() ? blocks[ready_blocks].out_off : c->raw.size();
    }
    // one step of THE walk at offset p, on the bytes [0, limit) of a stream of `total` bytes: 1 = a record (its start appended by
    // the caller, *next = the offset behind it), 0 = the stream ends or is malformed here (the walk is over), -1 = not decidable
    // on `limit` bytes yet.  The only place the record rules live: the walker and the speculation share it.
    static int walk_step(const uint8_t *d, size_t p, size_t limit, size_t total, size_t *next)
    {
        if (p + 4 > total) return 0;
        if (p + 4 > limit) return -1;
        const size_t bs = le32(d + p);
        if (bs < 32) return 0;                                        // truncated tail: stop like a failed sam_read1
        if (p + 4 + bs > total) return 0;
        if (p + 4 + bs > limit) return -1;
        // the variable-length fields must fit the record (htslib's bam_read1 fails on such a record, which ends the
        // reference's `while (sam_read1(...) >= 0)` loop at generate_graph.cpp:644): name, CIGAR, packed bases, qualities
        const uint8_t *r = d + p + 4;
        const size_t l_name = r[8], n_cig = le16(r + 12), l_seq = le32(r + 16);
        if (l_name < 1 || l_seq > 0x7fffffffu || 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq > bs) return 0;
        *next = p + 4 + bs;
        return 1;
    }
    bool plausible_chain(size_t p, size_t limit) const;
    void spec_segment(size_t s, size_t limit);
    void spec_some(size_t &cursor, bool drain);
    // the decode of the records into columns, pipelined behind the record walk (load_bam_finish): the walker publishes how many record
    // starts it has found, the loader's threads -- done with the inflate -- take chunks of records as they become known
    static constexpr size_t kChunk = 32768;
    std::vector<uint64_t> rec_at;        // reserved to the most records the stream can hold: never reallocated while it is read
    std::atomic<size_t> n_walked{0};
    std::atomic<bool> walk_done{false};
    std::atomic<int> decode_go{0};       // the name index is filled (SA items can be resolved): the decode may start
    std::atomic<size_t> next_chunk{0};
    uint64_t key_seed = 1;
    Column<int32_t> sa_cnt;
    std::vector<std::vector<palace_sa_item>> sa_part;      // per chunk, in record order
    std::vector<std::vector<int32_t>> ms_part;            // (tid, pos, len) triples per chunk
    void decode_range(size_t a, size_t b, size_t part);
    void decode_chunks();
    std::atomic<int> helpers_running{0};
    size_t hold_at = SIZE_MAX;           // tests (PALACE_BAM_HOST_SHARE=<per cent>): the threads stop there while a helper is at work
    bool claim_front(size_t *i)
    {
        for (;;) {
            {
                std::lock_guard<std::mutex> g(claim_mu);
                if (next_front >= next_back) return false;
                if (next_front < hold_at || helpers_running.load() == 0) { *i = next_front++; return true; }
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    // A helper's batch comes back after its transfer and decode latency (a device: ~20 ms for the slowest member of any batch, plus
    // the copies), in which the loader's threads get through some 1 500 members themselves: nothing of the last 1 500, and of what
    // is left two fifths per claim -- a device with two helper threads decodes ~3 x what sixteen host threads do, so it should end
    // up with about three quarters of what both start on (0.4 + 0.4 x 0.6 now, the same of what is left when a helper returns)
    bool claim(size_t max, size_t *first, size_t *n) override
    {
        std::lock_guard<std::mutex> g(claim_mu);
        const size_t left = next_back - next_front;
        if (bad || left < 1500) return false;
        *n = std::min(max, left * 2 / 5);
        next_back -= *n;
        *first = next_back;
        return true;
    }
    const BgzfMember &member(size_t i) const override { return as_members[i]; }
    void finished(size_t i, bool decoded) override
    {
        if (!decoded && !inflate_member(file->data, file->size, as_members[i], c->raw.data() + as_members[i].out_off)) bad = true;
        if (decoded) by_helpers.fetch_add(1, std::memory_order_relaxed);
        done[i].store(1, std::memory_order_release);
    }
    ~BamLoad() override
    {
        bad = true;                          // unwinding from a header / record error: the workers stop at their next member
        if (walker.joinable()) walker.join();
        for (auto &t : workers) if (t.joinable()) t.join();
    }
};

std::vector<BgzfMember> bgzf_members(const uint8_t *file, size_t size, size_t *total_out)
{
    std::vector<BgzfMember> out;
    for (const Block &b : index_bgzf(file, size, total_out)) out.push_back(BgzfMember{b.in_off, b.in_len, b.out_off, b.out_len});
    return out;
}

bool inflate_member(const uint8_t *file, size_t size, const BgzfMember &m, uint8_t *out)
{
    if (m.out_len == 0) return true;
    if (inflate_fast(file + m.in_off, m.in_len, size - (m.in_off + m.in_len), out, m.out_len)) return true;
    z_stream zs{};
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(file + m.in_off);
    zs.avail_in = static_cast<uInt>(m.in_len);
    zs.next_out = out;
    zs.avail_out = static_cast<uInt>(m.out_len);
    const bool ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
    inflateEnd(&zs);
    return ok;
}

BamLoad *load_bam_begin(const std::string &path, int threads, BamColumns &c, const std::vector<MemberHelper> &helpers)
{
    Trace trh("bam/header");
    std::unique_ptr<BamLoad> L(new BamLoad());
    L->c = &c;
    L->threads = threads = std::max(1, threads);
    L->file.reset(new MappedFile(path));
    size_t total = 0;
    L->blocks = index_bgzf(L->file->data, L->file->size, &total);
    c.raw.alloc(total);
    const size_t nb = L->blocks.size();
    L->done.reset(new std::atomic<uint8_t>[nb ? nb : 1]);
    for (size_t i = 0; i < nb; i++) L->done[i].store(0, std::memory_order_relaxed);
    BamLoad *ld = L.get();
    L->next_back = nb;
    // (a helper costs a device context and its buffers: not for files the threads are done with before the HIP runtime is even up)
    if (!helpers.empty() && (nb >= 4096 || std::getenv("PALACE_BAM_HOST_SHARE"))) {
        L->as_members.reserve(nb);
        for (const Block &b : L->blocks) L->as_members.push_back(BgzfMember{b.in_off, b.in_len, b.out_off, b.out_len});
        L->file_data = L->file->data;
        L->file_size = L->file->size;
        L->out = c.raw.data();
        if (const char *e = std::getenv("PALACE_BAM_HOST_SHARE")) L->hold_at = nb * static_cast<size_t>(std::max(0, std::min(100, std::atoi(e)))) / 100;
        L->helpers_running = static_cast<int>(helpers.size());
        for (const MemberHelper &h : helpers) L->workers.emplace_back([ld, h] { h(*ld); ld->helpers_running.fetch_sub(1); ld->decode_chunks(); });
    }
    for (int t = 0; t < threads; t++)
        L->workers.emplace_back([ld] {
            z_stream zs{};
            if (inflateInit2(&zs, -15) != Z_OK) { ld->bad = true; return; }
            uint8_t *out = ld->c->raw.data();
            size_t spec_cursor = 0;
            const bool use_fast = std::getenv("PALACE_BAM_ZLIB") == nullptr;          // PALACE_BAM_ZLIB=1: zlib for every member (A/B, tests)
            for (size_t i = 0; !ld->bad && ld->claim_front(&i);) {
                const Block &k = ld->blocks[i];
                // the member decoder written for this loader first (inflate_fast.hpp); whatever it does not take, zlib decides
                if (k.out_len && !(use_fast && inflate_fast(ld->file->data + k.in_off, k.in_len, ld->file->size - (k.in_off + k.in_len),
                                                            out + k.out_off, k.out_len))) {
                    if (inflateReset(&zs) != Z_OK) { ld->bad = true; break; }
                    zs.next_in = const_cast<Bytef *>(ld->file->data + k.in_off);
                    zs.avail_in = static_cast<uInt>(k.in_len);
                    zs.next_out = out + k.out_off;
                    zs.avail_out = static_cast<uInt>(k.out_len);
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { ld->bad = true; break; }
                }
                ld->done[i].store(1, std::memory_order_release);
                ld->spec_some(spec_cursor, false);                  // a segment the inflate front has passed, if nobody has taken it
            }
            inflateEnd(&zs);
            // what is left of the segments, as the members of the helpers arrive: two of the threads (a segment is ~0.1 ms of walking; the
            // others decode columns behind the walker meanwhile)
            if (ld->spec_drainers.fetch_add(1) < 2) ld->spec_some(spec_cursor, true);
            ld->decode_chunks();
        });
    trh.lap("file mapped, members indexed, inflate threads started");
    // ---- header (BAM spec 4.2): magic, l_text, text, n_ref, then (l_name, name, l_ref) per reference ----
    const uint8_t *d = c.raw.data();
    size_t header_cursor = 0;
    auto need = [&](size_t upto) {
        if (L->wait_for(upto, header_cursor) < upto) throw std::runtime_error("Failed to read BAM header");
    };
    need(12);
    if (std::memcmp(d, "BAM\1", 4) != 0) throw std::runtime_error("Failed to read BAM header");
    size_t p = 8 + static_cast<size_t>(le32(d + 4));
    need(p + 4);
    const int32_t n_ref = static_cast<int32_t>(le32(d + p));
    p += 4;
    // (n_ref is bounded by what the stream can hold: >= 9 bytes per reference)
    const size_t cap = n_ref > 0 ? std::min<size_t>(static_cast<size_t>(n_ref), c.raw.size() / 9 + 1) : 0;
    std::vector<size_t> name_at;                                 // offset of every l_name word: a serial walk, each size is in the stream
    name_at.reserve(cap);
    for (int32_t i = 0; i < n_ref; i++) {
        need(p + 4);
        const size_t l = le32(d + p);
        need(p + 4 + l + 4);
        name_at.push_back(p);
        p += 8 + l;
    }
    trh.lap("name offsets walked (behind the inflate front)");
    // The end of the header is known: the record walk starts now, on a thread of its own, beside the rest of the header work (names,
    // hashes, the name index) -- the walk is the longest serial piece of the load (6.7 M dependent steps at 1M contigs).  The columns
    // are sized for the most records the stream can hold (36 bytes each at least; the pages behind the ones that do not exist are never
    // touched) so that the threads can decode chunks of records while the walk is still finding the later ones.
    L->first_record = p;
    L->n_ref = n_ref;
    {
        const size_t ub = (c.raw.size() > p ? (c.raw.size() - p) / 36 : 0) + 16;
        for (auto *v : {&c.tid, &c.pos, &c.mtid, &c.mpos, &c.nm, &c.ref_len, &c.read_len, &c.clip_s, &c.clip_e})
            v->resize(ub);
        c.flag.resize(ub); c.mapq.resize(ub); c.qkey.resize(ub);
        c.qname_at.resize(ub); c.qname_len.resize(ub);
        L->sa_cnt.resize(ub);
        const size_t max_chunks = ub / BamLoad::kChunk + 1;
        L->sa_part.resize(max_chunks);
        L->ms_part.resize(max_chunks);
        L->rec_at.reserve(ub);
        L->speculate = std::getenv("PALACE_BAM_SERIAL_WALK") == nullptr;      // PALACE_BAM_SERIAL_WALK=1: the walker alone (A/B, tests)
        const size_t n_segs = L->speculate && c.raw.size() > p ? (c.raw.size() - p + BamLoad::kSeg - 1) / BamLoad::kSeg : 0;
        L->segs.reset(new BamLoad::SpecSeg[n_segs ? n_segs : 1]);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        L->n_segs_pub.store(n_segs, std::memory_order_release);     // (the workers have been running since before the header was parsed)
        L->walker = std::thread([ld] { ld->walk(); });
    }
    // names, lengths and name hashes on the threads; the index itself is filled by this thread (hashes in hand)
    const size_t nr = name_at.size();
    c.target_name.resize(nr);
    c.target_len.resize(nr);
    std::vector<uint64_t> hash(nr);
    {
        const size_t parts = nr < 4096 ? 1 : static_cast<size_t>(std::min(threads, 8));
        std::vector<std::thread> pool;
        for (size_t t = 0; t < parts; t++)
            pool.emplace_back([&, t] {
                for (size_t i = nr * t / parts; i < nr * (t + 1) / parts; i++) {
                    const size_t at = name_at[i], l = le32(d + at);
                    const std::string_view nm(reinterpret_cast<const char *>(d + at + 4), l ? l - 1 : 0);
                    c.target_name[i].assign(nm);
                    c.target_len[i] = static_cast<int32_t>(le32(d + at + 4 + l));
                    hash[i] = hash_bytes(nm);
                }
            });
        for (auto &th : pool) th.join();
    }
    trh.lap("names, lengths, hashes (threads)");
    c.tid_names.reserve(nr + 16);
    c.tid_of_name.reserve(nr + 16);
    for (size_t i = 0; i < nr; i++) {
        if (i + 8 < nr) c.tid_names.prefetch(hash[i + 8]);
        const size_t at = name_at[i], l = le32(d + at);
        const int k = c.tid_names.intern_hashed(std::string_view(reinterpret_cast<const char *>(d + at + 4), l ? l - 1 : 0), hash[i]);
        if (static_cast<size_t>(k) >= c.tid_of_name.size()) c.tid_of_name.resize(static_cast<size_t>(k) + 1);
        c.tid_of_name[static_cast<size_t>(k)] = static_cast<int32_t>(i);
    }
    trh.lap("name index filled");
    L->decode_go.store(1, std::memory_order_release);            // the name index is there: SA items can be resolved
    return L.release();
}

size_t load_bam_size_hint(const BamLoad *load) { return load->c->raw.size(); }

void load_bam(const std::string &path, int threads, uint64_t key_seed, BamColumns &c)
{
    load_bam_finish(load_bam_begin(path, threads, c), key_seed);
}

// a record could start at p: the fixed fields of four records in a row are what a record's are (stricter than the walk's own rules,
// which decide nothing here: a wrong guess only costs its segment the speculation)
bool BamLoad::plausible_chain(size_t p, size_t limit) const
{
    const uint8_t *d = c->raw.data();
    for (int k = 0; k < 4; k++) {
        if (p + 36 > limit) return false;
        const size_t bs = le32(d + p);
        if (bs < 32 || bs > (size_t{1} << 24) || p + 4 + bs > limit) return false;
        const uint8_t *r = d + p + 4;
        const int32_t tid = static_cast<int32_t>(le32(r)), pos = static_cast<int32_t>(le32(r + 4)), mtid = static_cast<int32_t>(le32(r + 20)),
                      mpos = static_cast<int32_t>(le32(r + 24));
        const size_t l_name = r[8], n_cig = le16(r + 12), l_seq = le32(r + 16);
        if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1) return false;
        if (l_name < 1 || l_seq > (size_t{1} << 28) || 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq > bs) return false;
        if (r[32 + l_name - 1] != 0) return false;                    // the read name is NUL-terminated
        p += 4 + bs;
    }
    return true;
}

// the walk of segment s from a guessed entry, on the first `limit` bytes of the stream (final when the segment was taken)
void BamLoad::spec_segment(size_t s, size_t limit)
{
    const uint8_t *d = c->raw.data();
    const size_t total = c->raw.size(), seg_lo = first_record + s * kSeg, seg_hi = std::min(total, seg_lo + kSeg);
    SpecSeg &sg = segs[s];
    size_t p = seg_lo;
    if (s > 0) {                                                      // (the first segment starts where the records do)
        while (p < seg_hi && !plausible_chain(p, limit)) p++;
    }
    sg.exit = p;
    if (p < seg_hi) {
        sg.rec.reserve(kSeg / 160);
        for (;;) {
            size_t next = 0;
            const int st = p < seg_hi ? walk_step(d, p, limit, total, &next) : -1;
            if (st <= 0) { sg.stopped = st == 0; break; }
            sg.rec.push_back(p + 4);
            p = next;
        }
        sg.exit = p;
    }
    sg.state.store(2, std::memory_order_release);
}

// an inflate worker between two members (drain = false: at most one segment, only if the inflate front has passed it) or out of
// members (drain = true: every segment nobody has taken, as the members of the helpers arrive)
void BamLoad::spec_some(size_t &cursor, bool drain)
{
    for (;;) {
        const size_t n = n_segs_pub.load(std::memory_order_acquire);
        if (n == SIZE_MAX) {                                          // the header is still being read
            if (!drain || bad) return;
            std::this_thread::sleep_for(std::chrono::microseconds(100));
            continue;
        }
        size_t s = next_seg.load(std::memory_order_relaxed);
        if (s >= n) return;
        const size_t need = std::min(c->raw.size(), first_record + (s + 1) * kSeg + kSegSlack);
        const size_t have = ready_prefix(cursor);
        if (have < need) {
            if (!drain || bad) return;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            continue;
        }
        if (!next_seg.compare_exchange_strong(s, s + 1)) continue;
        spec_segment(s, have);
        if (!drain) return;
    }
}

void BamLoad::walk()
{
    const uint8_t *d = c->raw.data();
    const size_t total = c->raw.size();
    size_t cursor = 0;
    try {
        size_t p = first_record, have = wait_for(std::min(total, p + 4), cursor), ahead = p & ~size_t{63};
        const size_t n = n_segs_pub.load(std::memory_order_acquire) == SIZE_MAX ? 0 : n_segs_pub.load(std::memory_order_acquire);
        size_t seg = 0;                                               // the segment p lies in (speculation on)
        int misses = 0;                                               // serial steps in this segment that did not meet its list
        for (;;) {
            if (n) {
                const size_t in = (p - first_record) / kSeg;
                if (in != seg) { seg = in; misses = 0; }
                // the true position is a record start of the segment's list: from here the list IS the walk
                if (seg < n && misses < 64 && segs[seg].state.load(std::memory_order_acquire) == 2) {
                    const SpecSeg &sg = segs[seg];
                    const auto it = std::lower_bound(sg.rec.begin(), sg.rec.end(), static_cast<uint64_t>(p + 4));
                    if (it != sg.rec.end() && *it == p + 4) {
                        rec_at.insert(rec_at.end(), it, sg.rec.end());
                        n_adopted += static_cast<size_t>(sg.rec.end() - it);
                        n_walked.store(rec_at.size(), std::memory_order_release);
                        p = sg.exit;
                        if (sg.stopped) break;
                        ahead = p & ~size_t{63};
                        continue;
                    }
                    misses++;
                }
            }
            size_t next = 0;
            int st = walk_step(d, p, have, total, &next);
            if (st < 0) {                                             // behind the inflate front: wait for the bytes the step needs
                have = wait_for(std::min(total, p + 4), cursor);
                if (p + 4 <= have) have = wait_for(std::min(total, p + 4 + static_cast<size_t>(le32(d + p))), cursor);
                st = walk_step(d, p, have, total, &next);
                if (st < 0) break;                                    // (everything that will ever come is there, and it is not enough)
            }
            if (st == 0) break;
            rec_at.push_back(p + 4);
            if ((rec_at.size() & 4095) == 0) n_walked.store(rec_at.size(), std::memory_order_release);
            p = next;
            // the next few record heads lie in the kilobyte behind this one, not at a fixed stride (the hardware does not see a
            // stream): every line of that kilobyte is asked for as the walk exposes it
            for (const size_t upto = std::min(have, p + 1024); ahead + 64 <= upto; ahead += 64) __builtin_prefetch(d + ahead);
            if (ahead < p) ahead = p & ~size_t{63};
        }
    } catch (const std::exception &e) { walk_error = e.what(); }
    n_walked.store(rec_at.size(), std::memory_order_release);
    walk_done.store(true, std::memory_order_release);
}

// records [a, b) of the walk into the columns (every element written exactly once, by the thread that has the chunk)
void BamLoad::decode_range(size_t a, size_t b, size_t part)
{
    BamColumns &c = *this->c;
    const uint8_t *d = c.raw.data();
    const int32_t n_ref = this->n_ref;
    const uint64_t key_seed = this->key_seed;
    static const char opchr[] = "MIDNSHP=XB??????";
        for (size_t i = a; i < b; i++) {
            const uint8_t *r = d + rec_at[i];
            const uint8_t *end = r + le32(r - 4);
            const int32_t tid = static_cast<int32_t>(le32(r));
            c.tid[i] = tid;
            c.nm[i] = 0;
            sa_cnt[i] = 0;
            c.pos[i] = static_cast<int32_t>(le32(r + 4));
            const size_t l_name = r[8];
            c.mapq[i] = r[9];
            const size_t n_cig = le16(r + 12);
            c.flag[i] = le16(r + 14);
            const size_t l_seq = le32(r + 16);
            c.mtid[i] = static_cast<int32_t>(le32(r + 20));
            c.mpos[i] = static_cast<int32_t>(le32(r + 24));
            const uint8_t *name = r + 32;
            size_t nlen = l_name ? l_name - 1 : 0;
            if (const void *z = std::memchr(name, 0, l_name)) nlen = static_cast<const uint8_t *>(z) - name;   // C-string view (:651)
            c.qname_at[i] = static_cast<uint64_t>(name - d);
            c.qname_len[i] = static_cast<uint8_t>(nlen);
            c.qkey[i] = name_key(reinterpret_cast<const char *>(name), nlen, key_seed);
            const uint8_t *cg = name + l_name;
            const uint8_t *x0 = cg + 4 * n_cig + (l_seq + 1) / 2 + l_seq;      // first aux field
            // A CIGAR of more than 65535 ops is stored in a CG:B,I tag behind a placeholder (SAM spec 4.2.2); htslib
            // puts it back in place inside bam_read1 (bam_tag2cigar), so that is what the reference sees.
            const uint8_t *ops = cg;
            size_t n_ops = n_cig;
            if (n_cig > 0 && tid >= 0 && static_cast<int32_t>(le32(r + 4)) >= 0 && (le32(cg) & 15) == 4 && (le32(cg) >> 4) == l_seq) {
                for (const uint8_t *x = x0; x + 3 <= end;) {
                    const uint8_t *v = x + 3;
                    const size_t sz = aux_size(x[2], v, end);
                    if (!sz || sz > static_cast<size_t>(end - v)) break;
                    if (x[0] == 'C' && x[1] == 'G') {              // first CG tag decides (bam_aux_get)
                        if (x[2] == 'B' && (v[0] == 'I' || v[0] == 'i') && le32(v + 1) >= n_cig && le32(v + 1) < (1u << 29)) {
                            ops = v + 5;
                            n_ops = le32(v + 1);
                        }
                        break;
                    }
                    x = v + sz;
                }
            }
            int32_t rl = 0, ql = 0;
            OpScan sc;
            const bool depth_counts = c.want_match_segments && !(le16(r + 14) & 0x704) && tid >= 0 && tid < n_ref && static_cast<int32_t>(le32(r + 4)) >= 0;
            for (size_t k = 0; k < n_ops; k++) {
                uint32_t v = le32(ops + 4 * k);
                int op = v & 15, len = static_cast<int>(v >> 4);
                if (depth_counts && len > 0 && (op == 0 || op == 7 || op == 8)) {        // a match segment at pos + (ref consumed so far)
                    auto &m = ms_part[part];
                    m.push_back(tid); m.push_back(static_cast<int32_t>(le32(r + 4)) + rl); m.push_back(len);
                }
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += len;   // bam_cigar2rlen
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += len;   // getReadLength (:385-397)
                sc.add(len, opchr[op]);
            }
            ClipInfo ci = sc.done();
            c.ref_len[i] = rl;
            c.read_len[i] = ql;
            c.clip_s[i] = n_ops ? ci.clip_s : -1;
            c.clip_e[i] = ci.clip_e;
            // aux: first NM (integer types only, like bam_aux2i) and first SA (Z)
            const uint8_t *x = x0;
            bool have_nm = false, have_sa = false;
            while (x + 3 <= end && !(have_nm && have_sa)) {
                uint8_t ty = x[2];
                const uint8_t *v = x + 3;
                size_t sz = aux_size(ty, v, end);
                if (!sz || sz > static_cast<size_t>(end - v)) break;
                if (!have_nm && x[0] == 'N' && x[1] == 'M') {
                    have_nm = true;
                    int64_t val = 0;
                    switch (ty) {
                    case 'c': val = static_cast<int8_t>(v[0]); break;
                    case 'C': val = v[0]; break;
                    case 's': val = static_cast<int16_t>(le16(v)); break;
                    case 'S': val = le16(v); break;
                    case 'i': val = static_cast<int32_t>(le32(v)); break;
                    case 'I': val = le32(v); break;
                    default: val = 0;
                    }
                    c.nm[i] = static_cast<int32_t>(val);
                } else if (!have_sa && x[0] == 'S' && x[1] == 'A' && ty == 'Z') {
                    have_sa = true;
                    if (tid >= 0 && tid < n_ref) {                         // :687
                        const char *s = reinterpret_cast<const char *>(v), *se = s + sz - 1;
                        while (s < se) {                                   // items split at ';' (:719-720)
                            const char *semi = static_cast<const char *>(std::memchr(s, ';', static_cast<size_t>(se - s)));
                            const char *ie = semi ? semi : se;
                            palace_sa_item it{};
                            if (ie > s && parse_sa(s, ie, c, tid, it)) { sa_part[part].push_back(it); sa_cnt[i]++; }
                            s = semi ? semi + 1 : se;
                        }
                    }
                }
                x = v + sz;
            }
        }
}

void BamLoad::decode_chunks()
{
    while (!decode_go.load(std::memory_order_acquire)) {          // (the header is being parsed, or the loader is being torn down)
        if (bad) return;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    for (;;) {
        const size_t part = next_chunk.fetch_add(1), a = part * kChunk;
        size_t have;
        for (;;) {
            const bool done = walk_done.load(std::memory_order_acquire);
            have = n_walked.load(std::memory_order_acquire);
            if (have >= a + kChunk || done) break;
            if (bad) return;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (a >= have || part >= sa_part.size()) return;
        decode_range(a, std::min(a + kChunk, have), part);
    }
}

void load_bam_finish(BamLoad *load, uint64_t key_seed)
{
    std::unique_ptr<BamLoad> L(load);
    BamColumns &c = *L->c;
    Trace tr("bam");
    L->walker.join();                                             // the record walk (started by load_bam_begin), behind the inflate front
    if (!L->walk_error.empty()) throw std::runtime_error(L->walk_error);
    std::vector<uint64_t> &rec_at = L->rec_at;
    tr.lap("record boundaries (behind the inflate front)");
    L->decode_chunks();                                           // this thread helps with what is left
    for (auto &t : L->workers) t.join();                          // (members behind a malformed record are still inflated)
    tr.lap("inflate threads joined, columns decoded");
    if (tr.on && L->speculate)
        std::fprintf(stderr, "[bam] %zu of %zu record boundaries came from the segments walked ahead by the inflate threads\n", L->n_adopted, rec_at.size());
    if (tr.on && !L->as_members.empty())
        std::fprintf(stderr, "[bam] %zu of %zu members were inflated by helpers (device)\n", L->by_helpers.load(), L->blocks.size());
    if (L->bad) throw std::runtime_error("BGZF inflate failed");
    L->file.reset();
    const size_t n = rec_at.size();
    for (auto *v : {&c.tid, &c.pos, &c.mtid, &c.mpos, &c.nm, &c.ref_len, &c.read_len, &c.clip_s, &c.clip_e})
        v->resize(n);
    c.flag.resize(n); c.mapq.resize(n); c.qkey.resize(n);
    c.qname_at.resize(n); c.qname_len.resize(n);
    c.sa_off.resize(n + 1);
    c.sa_off[0] = 0;
    const Column<int32_t> &sa_cnt = L->sa_cnt;
    for (size_t i = 0; i < n; i++) c.sa_off[i + 1] = c.sa_off[i] + sa_cnt[i];
    c.sa.clear();
    c.sa.reserve(static_cast<size_t>(c.sa_off[n]) + 1);
    for (auto &part : L->sa_part) c.sa.insert(c.sa.end(), part.begin(), part.end());   // chunks are in record order
    size_t n_ms = 0;
    for (auto &part : L->ms_part) n_ms += part.size() / 3;
    c.mseg_tid.reserve(n_ms); c.mseg_pos.reserve(n_ms); c.mseg_len.reserve(n_ms);
    for (auto &part : L->ms_part)
        for (size_t k = 0; k + 2 < part.size(); k += 3) { c.mseg_tid.push_back(part[k]); c.mseg_pos.push_back(part[k + 1]); c.mseg_len.push_back(part[k + 2]); }
    if (key_seed != L->key_seed) rekey(c, key_seed);             // (the decode keyed the read names with the default seed)
    tr.lap("SA items + match segments joined");
}

}  // namespace palace_host
