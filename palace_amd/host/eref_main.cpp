// eref -- drop-in for the reference executable (bin/extract_ref.cpp; call site palace:475-477):
//     eref <fq1> <fq2> <phagedb.fa> <tmp.txt> <hit_ratio> <perfect_ratio> <threads> [<refs_out.fasta> <percent_out.txt>]  > ref_names.txt
// The two optional arguments fold the next pipeline step in (palace:483-498, get_ref_by_index.py + the .fai it reads): the
// reported references as FASTA and the "name <TAB> ratio" table, written from the DB this run has parsed anyway.
// Host side: text parsing, the index file contract, stdout formatting.  The k-mer work (index
// build, read counting, reference scan) runs in HIP through libpalace_hip.so; no CPU path exists.
//
// Defined behaviour where the reference has none: results are those of the
// reference at threads=1 with a pre-existing index, for every <threads> value; lines are printed
// in index order.  <threads> only sizes the host-side text parsing.
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <future>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/palace_hip.h"
#include "fastx.hpp"
#include "trace.hpp"
#include "fast_exit.hpp"
#include "device_pick.hpp"

using namespace palace_host;

namespace {

#define CK(call)                                                                   \
    do {                                                                           \
        int rc__ = (call);                                                         \
        if (rc__ != 0) {                                                           \
            std::cerr << "eref: " #call " failed: " << palace_last_error() << "\n"; \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// glibc srand()/rand() (TYPE_3 additive feedback generator, as documented in random_r.c), so the
// E3 subsampling decisions (extract_ref.cpp:955-960) match the reference's seed-1 stream.
struct GlibcRand {
    uint32_t r[34];
    int f = 3, b = 0;
    explicit GlibcRand(unsigned seed)
    {
        int32_t w = seed ? static_cast<int32_t>(seed) : 1;
        r[0] = static_cast<uint32_t>(w);
        for (int i = 1; i < 31; i++) {
            long hi = w / 127773, lo = w % 127773;
            w = static_cast<int32_t>(16807 * lo - 2836 * hi);
            if (w < 0) w += 2147483647;
            r[i] = static_cast<uint32_t>(w);
        }
        for (int i = 0; i < 310; i++) next();
    }
    int next()
    {
        r[f] += r[b];
        int out = static_cast<int>(r[f] >> 1);
        f = (f + 1) % 31;
        b = (b + 1) % 31;
        return out;
    }
};

bool file_exists(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

// random_coder (extract_ref.cpp:1082-1102): one of six orders of (0,1,2) per k-mer offset.  The
// reference seeds this with time(0); any choice is valid, so a fixed default keeps runs
// reproducible (override with PALACE_CODER_SEED).
bool make_header(uint8_t hdr[400])
{
    // PALACE_CODER_HEADER=<file>: take the 400-byte coder header from a file (e.g. the first 400 bytes of an index built
    // elsewhere) instead of drawing one -- several machines can then build interchangeable indices of one DB without
    // shipping 12 bytes per reference position.
    if (const char *path = std::getenv("PALACE_CODER_HEADER")) {
        std::ifstream f(path, std::ios::binary);
        if (!f.read(reinterpret_cast<char *>(hdr), 400)) { std::cerr << "eref: cannot read 400 bytes from " << path << "\n"; return false; }
        return true;
    }
    static const int16_t orders[18] = {0, 1, 2, 0, 2, 1, 1, 2, 0, 1, 0, 2, 2, 0, 1, 2, 1, 0};
    const char *env = std::getenv("PALACE_CODER_SEED");
    GlibcRand g(env ? static_cast<unsigned>(std::strtoul(env, nullptr, 10)) : 20261003u);
    int16_t cc[97] = {0};
    for (int z = 0; z < 32; z++) {
        int pick = g.next() % 6;
        for (int i = 0; i < 3; i++) cc[3 * z + i] = orders[3 * pick + i];
    }
    std::memset(hdr, 0, 400);
    for (int j = 0; j < 96; j++) {                        // 4-byte writes from a 2-byte array (:680-682)
        uint32_t w = static_cast<uint16_t>(cc[j]) | (static_cast<uint32_t>(static_cast<uint16_t>(cc[j + 1])) << 16);
        std::memcpy(hdr + 4 * j, &w, 4);
    }
    return true;
}

template <class T>
int upload(palace_ctx *ctx, const T *src, size_t n, T **d)
{
    void *p = nullptr;
    int rc = palace_malloc(ctx, (n ? n : 1) * sizeof(T), &p);
    if (rc) return rc;
    *d = static_cast<T *>(p);
    return palace_h2d(ctx, p, src, n * sizeof(T));
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 8) {
        std::cerr << "Usage: " << argv[0] << " <fq1> <fq2> <phagedb.fa> <tmp.txt> <hit_ratio> <perfect_ratio> <threads>\n";
        return 1;
    }
    palace_host::FastExit fast_exit = palace_host::fast_exit_begin();   // from here on this is the worker process (fast_exit.hpp)
    const int device = palace_host::pick_device();                       // PALACE_DEVICE (device_pick.hpp): before anything touches HIP
    const std::string fq1 = argv[1], fq2 = argv[2], fasta = argv[3], interval_name = argv[4];
    const float hit_ratio = static_cast<float>(std::stod(argv[5]));            // :1228-1229
    const float perfect_ratio = static_cast<float>(std::stod(argv[6]));
    const int threads = std::max(1, std::min(64, static_cast<int>(std::stod(argv[7]))));
    const int window = 500;
    const int one_min = window * hit_ratio, three_min = window * perfect_ratio;   // :513-514

    Trace tr("eref");
    // The HIP runtime comes up (device, stream, 3 x 512 MiB of planes zeroed) on its own thread while this one maps and
    // scans the text inputs: neither waits for the other (the reference, too, reads with T threads per phase, :1267-1291).
    palace_ctx *ctx = nullptr;
    int ctx_rc = 0;
    std::string ctx_err, in_err;
    MappedText fq_txt[2];
    FastqPlan plan[2];
    std::shared_future<void> scan_done = std::async(std::launch::async, [&] {   // pass 1 over both FASTQ files: parts, line phases, sizes
        try {
            for (int side = 0; side < 2; side++) {
                fq_txt[side].open(side == 0 ? fq1 : fq2);
                plan_fastq(fq_txt[side], threads, plan[side]);
            }
        } catch (const std::exception &e) { in_err = e.what(); }
    }).share();
    std::thread hip_up([&] {
        Trace th("eref/hip");
        ctx_rc = palace_ctx_create(device, &ctx);
        th.lap("context");
        if (!ctx_rc) ctx_rc = palace_eref_table_reset(ctx);
        th.lap("table");
        // One-shot process: count in slabs of 2^28 positions (6.5 GB of scratch) instead of the library's 2^30 (26 GB).  The
        // driver hands out zeroed device memory, and zeroing what the previous process left behind was measured at up to
        // 1.1 s for 26 GB; the three extra passes over the planes cost ~3 ms.  PALACE_EREF_SLAB=<positions> overrides.
        if (!ctx_rc) {
            const char *e = std::getenv("PALACE_EREF_SLAB");
            ctx_rc = palace_eref_set_option(ctx, "slab_bases", e ? std::atoll(e) : (1ll << 28));
        }
        if (!ctx_rc) {                                      // ... and the scratch memory of the count, as soon as its size is known
            scan_done.wait();
            th.lap("fastq pass 1 (waited)");
            if (in_err.empty()) ctx_rc = palace_eref_reserve(ctx, plan[0].n_bases + plan[1].n_bases);
            th.lap("scratch reserved");
        }
        if (ctx_rc) ctx_err = palace_last_error();          // (the message is per thread)
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } join_hip{hip_up};

    // ---- references: FASTA -> records longer than k=32 (:697), index-file contract ----
    SeqSet db_all, db;
    try {
        MappedText fa_txt(fasta);
        parse_fasta_mt(fa_txt.data, fa_txt.size, db_all, static_cast<int>(std::max(1u, std::min(16u, std::thread::hardware_concurrency()))));
    } catch (const std::exception &e) { std::cerr << "eref: " << e.what() << "\n"; return 1; }
    std::vector<int64_t> cum(db_all.n() + 1, 0);
    bool all_long = db_all.n() > 0 && db_all.len(0) == 0;          // the usual DB: nothing ahead of the first header, every record > 32 bases
    for (int64_t i = 0; i < db_all.n(); i++) {
        cum[i + 1] = cum[i] + db_all.len(i);
        if (i > 0 && db_all.len(i) <= 32) all_long = false;
    }
    if (all_long) {                                                 // then the record set IS the reference set: no second copy of the bases
        db.bases = std::move(db_all.bases);                         // (record i of db_all starts at the same offset in it: record 0 is empty)
        db.offsets.assign(db_all.offsets.begin() + 1, db_all.offsets.end());
        db.names.assign(db_all.names.begin() + 1, db_all.names.end());
        db.ordinal.resize(static_cast<size_t>(db_all.n() - 1));
        for (int64_t i = 1; i < db_all.n(); i++) db.ordinal[static_cast<size_t>(i - 1)] = i;
    }
    for (int64_t i = 0; i < db_all.n() && !all_long; i++) {
        if (db_all.len(i) > 32) {
            db.bases.insert(db.bases.end(), db_all.bases.begin() + db_all.offsets[i], db_all.bases.begin() + db_all.offsets[i + 1]);
            db.offsets.push_back(static_cast<int64_t>(db.bases.size()));
            db.names.push_back(db_all.names[i]);
            db.ordinal.push_back(i);                     // position in db_all
        }
    }
    tr.lap("fasta parsed");
    const int64_t n_refs = db.n();
    std::vector<int64_t> idx_off(n_refs + 1, 0);
    for (int64_t r = 0; r < n_refs; r++) idx_off[r + 1] = idx_off[r] + 3 * (db.len(r) - 31);
    const uint64_t index_bytes = 400 + 4ull * n_refs + 4ull * static_cast<uint64_t>(idx_off[n_refs]);

    hip_up.join();
    tr.lap("hip runtime up (joined)");
    if (ctx_rc) { std::cerr << "eref: cannot set up the GPU: " << ctx_err << "\n"; return 1; }
    uint8_t *d_ref = nullptr; int64_t *d_ref_off = nullptr;
    CK(upload(ctx, db.bases.data(), db.bases.size(), &d_ref));
    CK(upload(ctx, db.offsets.data(), db.offsets.size(), &d_ref_off));

    const std::string index_name = fasta + ".k32.index.dat";                   // :1245
    uint8_t hdr[400];
    if (!file_exists(index_name)) {                                            // :1246-1251 -> read_ref
        if (!make_header(hdr)) return 1;
        CK(palace_eref_set_coder(ctx, hdr));
        uint32_t *d_idx = nullptr; int64_t *d_idx_off = nullptr;
        void *p = nullptr;
        CK(palace_malloc(ctx, static_cast<size_t>(idx_off[n_refs] + 1) * 4, &p));
        d_idx = static_cast<uint32_t *>(p);
        CK(upload(ctx, idx_off.data(), idx_off.size(), &d_idx_off));
        CK(palace_eref_index_refs(ctx, d_ref, d_ref_off, n_refs, d_idx, d_idx_off));
        std::vector<uint32_t> idx(static_cast<size_t>(idx_off[n_refs]));
        CK(palace_d2h(ctx, idx.data(), d_idx, idx.size() * 4));
        CK(palace_free(ctx, d_idx)); CK(palace_free(ctx, d_idx_off));
        std::ofstream fi(index_name, std::ios::binary), fl(fasta + ".genome.len.txt");
        if (!fi || !fl) { std::cerr << "eref: cannot write index beside " << fasta << "\n"; return 1; }
        fi.write(reinterpret_cast<const char *>(hdr), 400);
        for (int64_t r = 0; r < n_refs; r++) {
            uint32_t l32 = static_cast<uint32_t>(db.len(r));
            fi.write(reinterpret_cast<const char *>(&l32), 4);                  // :710
            fi.write(reinterpret_cast<const char *>(idx.data() + idx_off[r]), 4 * (idx_off[r + 1] - idx_off[r]));
            const int64_t a = db.ordinal[r];                                    // :698, :803
            fl << db.names[r] << "\t" << db_all.ordinal[a] << "\t" << db.len(r) << "\t" << cum[a + 1] << "\n";
        }
    } else {                                                                   // saved_random_coder (:1104-1122)
        std::ifstream fi(index_name, std::ios::binary);
        if (!fi.read(reinterpret_cast<char *>(hdr), 400)) { std::cerr << "eref: short index " << index_name << "\n"; return 1; }
        fi.seekg(0, std::ios::end);
        if (static_cast<uint64_t>(fi.tellg()) != index_bytes) {
            std::cerr << "eref: " << index_name << " does not belong to " << fasta << " (size " << fi.tellg()
                      << ", expected " << index_bytes << "); delete it to rebuild\n";
            return 1;
        }
        CK(palace_eref_set_coder(ctx, hdr));
    }

    // ---- reads: Phase A ----
    // Both FASTQ sides become ONE read set in HBM (side 2 behind side 1) and are counted by one launch: the partition
    // kernels then run once and every plane slice is loaded and stored once.  Pass 2 over the text copies the sequence
    // lines of a run of parts straight to their place in a page-locked staging buffer (all threads), which goes to the
    // device with an asynchronous copy while the other staging buffer is being filled.
    tr.lap("refs uploaded / index ready");
    scan_done.wait();
    tr.lap("fastq pass 1 (joined)");
    if (!in_err.empty()) { std::cerr << "eref: " << in_err << "\n"; return 1; }
    const int64_t n_reads = plan[0].n_reads + plan[1].n_reads, n_bases = plan[0].n_bases + plan[1].n_bases;
    long sample = static_cast<long>(plan[0].n_bases) * 2;                       // cal_sam_ratio (:1124-1148)
    long target = 2000000000L;                                                  // down_sampling_size (:1230)
#ifdef PALACE_TEST_HOOKS
    if (const char *t = std::getenv("PALACE_EREF_SAMPLE_TARGET")) target = std::atol(t);   // test builds only (bin/eref_testhooks)
#endif
    const int down_sam_ratio = sample > 0 ? static_cast<int>(100L * target / sample) : 100;
    // E3: one draw per sequence line, file order (:955-960, seeded :1239-1240)
    std::vector<uint8_t> keep;
    if (down_sam_ratio < 100) {
        GlibcRand rng(1);
        keep.resize(static_cast<size_t>(n_reads));
        for (int64_t i = 0; i < n_reads; i++) keep[static_cast<size_t>(i)] = (rng.next() % 100) < down_sam_ratio;
    }
    const char *in_mode = std::getenv("PALACE_EREF_INPUT");                      // "ascii": the byte-per-base entry (kept, tested equal)
    const bool packed = !(in_mode && std::strcmp(in_mode, "ascii") == 0);
    if (packed) {
        // The parser threads pack while they scan (fastx.hpp: pack_fastq_part): two bits per base and the 32-mer start mask
        // go to the device, 0.375 bytes per base; the stream kernel and the read-end marks do not run.
        std::vector<int64_t> pos0[2];
        int64_t n_pos = 0;
        for (int side = 0; side < 2; side++)
            for (const FastqPart &pt : plan[side].parts) { pos0[side].push_back(n_pos); n_pos += packed_span(pt.seq_bytes()); }
        const size_t stream_bytes = palace_eref_packed_bytes(n_pos);
        uint64_t *d_s[3] = {nullptr, nullptr, nullptr};
        for (int q = 0; q < 3; q++) { void *p = nullptr; CK(palace_malloc(ctx, stream_bytes, &p)); d_s[q] = static_cast<uint64_t *>(p); }
        tr.lap("device buffers");
        // per stream and buffer 8 MiB = 64 M positions (a dozen parts per thread); page-locking the ASCII path's 2 x 96 MiB
        // costs ~50 ms, these 2 x 24 MiB a quarter of it
        constexpr int64_t kStageWords = (8ll << 20) / 8;
        uint64_t *stage[2] = {nullptr, nullptr};
        for (int k = 0; k < 2; k++) { void *p = nullptr; CK(palace_host_alloc(ctx, static_cast<size_t>(3 * kStageWords * 8), &p)); stage[k] = static_cast<uint64_t *>(p); }
        tr.lap("pinned staging");
        int n_sent = 0;
        for (int side = 0; side < 2; side++) {
            const FastqPlan &pl = plan[side];
            const int64_t read_base = side == 0 ? 0 : plan[0].n_reads;
            const uint8_t *kp = keep.empty() ? nullptr : keep.data();
            for (size_t i0 = 0; i0 < pl.parts.size();) {
                size_t i1 = i0;                                                  // parts [i0, i1): as many as the buffer takes
                const int64_t w0 = pos0[side][i0] / 64;
                auto end_word = [&](size_t i) { return (pos0[side][i] + packed_span(pl.parts[i].seq_bytes())) / 64; };
                while (i1 < pl.parts.size() && end_word(i1) - w0 <= kStageWords) i1++;
                std::vector<uint64_t> big;                                       // one part larger than the buffer (huge lines)
                uint64_t *dst = stage[n_sent & 1];
                int64_t cap = kStageWords;
                if (i1 == i0) { i1 = i0 + 1; cap = end_word(i0) - w0; big.resize(static_cast<size_t>(3 * cap)); dst = big.data(); }
                else if (n_sent >= 2) CK(palace_mark_wait(ctx, 100 + ((n_sent - 2) & 1)));   // this buffer's previous copies have landed
                pool_for(i1 - i0, threads, [&](size_t k) {
                    const int64_t o = pos0[side][i0 + k] / 64 - w0;
                    pack_fastq_part(pl, i0 + k, dst + o, dst + cap + o, dst + 2 * cap + o, kp, read_base);
                });
                const int64_t nw = end_word(i1 - 1) - w0;
                for (int q = 0; q < 3; q++) {
                    if (big.empty()) CK(palace_h2d_async(ctx, d_s[q] + w0, dst + q * cap, static_cast<size_t>(nw) * 8));
                    else CK(palace_h2d(ctx, d_s[q] + w0, dst + q * cap, static_cast<size_t>(nw) * 8));
                }
                if (big.empty()) { CK(palace_mark(ctx, 100 + (n_sent & 1))); n_sent++; }
                i0 = i1;
            }
        }
        tr.lap("fastq pass 2 (packed) + staged h2d");
        if (n_pos) CK(palace_eref_count_reads_packed(ctx, reinterpret_cast<const uint32_t *>(d_s[0]), reinterpret_cast<const uint32_t *>(d_s[1]),
                                                     reinterpret_cast<const uint32_t *>(d_s[2]), n_pos, n_reads));
        tr.lap("count_reads_packed enqueued");
        CK(palace_sync(ctx));                                                    // the staging buffers are being read until the copies have landed
        for (int k = 0; k < 2; k++) CK(palace_host_free(ctx, stage[k]));
        for (int q = 0; q < 3; q++) CK(palace_free(ctx, d_s[q]));
    } else {
        uint8_t *d_b = nullptr, *d_k = nullptr; int64_t *d_o = nullptr;
        {
            void *p = nullptr;
            CK(palace_malloc(ctx, static_cast<size_t>(n_bases) + 64, &p)); d_b = static_cast<uint8_t *>(p);
            CK(palace_malloc(ctx, static_cast<size_t>(n_reads + 1) * 8, &p)); d_o = static_cast<int64_t *>(p);
        }
        tr.lap("device buffers");
        std::vector<int64_t> offsets(static_cast<size_t>(n_reads) + 1, 0);
        constexpr int64_t kStage = 96ll << 20;
        uint8_t *stage[2] = {nullptr, nullptr};
        for (int k = 0; k < 2; k++) { void *p = nullptr; CK(palace_host_alloc(ctx, static_cast<size_t>(kStage), &p)); stage[k] = static_cast<uint8_t *>(p); }
        tr.lap("pinned staging");
        int n_sent = 0;
        for (int side = 0; side < 2; side++) {
            const FastqPlan &pl = plan[side];
            const int64_t byte_base = side == 0 ? 0 : plan[0].n_bases, read_base = side == 0 ? 0 : plan[0].n_reads;
            for (size_t i0 = 0; i0 < pl.parts.size();) {
                size_t i1 = i0;                                                      // parts [i0, i1): as many as the buffer takes
                const int64_t b0 = pl.parts[i0].byte0;
                while (i1 < pl.parts.size() && pl.parts[i1].byte0 + pl.parts[i1].seq_bytes() - b0 <= kStage) i1++;
                uint8_t *dst = stage[n_sent & 1];
                std::vector<uint8_t> big;                                            // one part larger than the buffer (huge lines)
                if (i1 == i0) { i1 = i0 + 1; big.resize(static_cast<size_t>(pl.parts[i0].seq_bytes())); dst = big.data(); }
                else if (n_sent >= 2) CK(palace_mark_wait(ctx, 100 + ((n_sent - 2) & 1)));   // this buffer's previous copy has landed
                pool_for(i1 - i0, threads, [&](size_t k) {
                    extract_fastq_part(pl, i0 + k, dst, b0, offsets.data() + read_base, byte_base);
                });
                const int64_t nb = pl.parts[i1 - 1].byte0 + pl.parts[i1 - 1].seq_bytes() - b0;
                if (big.empty()) {
                    CK(palace_h2d_async(ctx, d_b + byte_base + b0, dst, static_cast<size_t>(nb)));
                    CK(palace_mark(ctx, 100 + (n_sent & 1)));
                    n_sent++;
                } else {
                    CK(palace_h2d(ctx, d_b + byte_base + b0, dst, static_cast<size_t>(nb)));
                }
                i0 = i1;
            }
        }
        tr.lap("fastq pass 2 + staged h2d");
        CK(palace_h2d(ctx, d_o, offsets.data(), offsets.size() * 8));               // (also waits for the staged copies)
        tr.lap("offsets h2d");
        if (!keep.empty()) CK(upload(ctx, keep.data(), keep.size(), &d_k));
        if (n_reads) CK(palace_eref_count_reads(ctx, d_b, d_o, n_reads, d_k, n_bases));
        tr.lap("count_reads enqueued");
        for (int k = 0; k < 2; k++) CK(palace_host_free(ctx, stage[k]));
        CK(palace_free(ctx, d_b)); CK(palace_free(ctx, d_o)); CK(palace_free(ctx, d_k));
    }
    for (int side = 0; side < 2; side++) plan[side].parts.clear();

    // ---- references: Phase B ----
    std::vector<int32_t> rows(static_cast<size_t>(4 * n_refs));
    if (n_refs) {
        void *p = nullptr;
        CK(palace_malloc(ctx, rows.size() * 4, &p));
        CK(palace_eref_scan_refs(ctx, d_ref, d_ref_off, n_refs, static_cast<int64_t>(db.bases.size()), one_min, three_min,
                                 static_cast<int32_t *>(p)));
        CK(palace_d2h(ctx, rows.data(), p, rows.size() * 4));
    }
    tr.lap("scan_refs + rows d2h");
    palace_ctx_destroy(ctx);
    tr.lap("ctx destroyed");

    { std::ofstream trunc(interval_name, std::ios::out | std::ios::trunc); }   // :825, :899 (left empty)
    std::string out;
    char line[160];
    for (int64_t r = 0; r < n_refs; r++) {
        const int n_int = rows[4 * r], el = rows[4 * r + 1], len = rows[4 * r + 2];
        const float ratio = static_cast<float>(el) / static_cast<float>(len);   // :615
        if (el > 0 && ratio > 0.75)
            out.append(line, std::snprintf(line, sizeof line, "ref_index\t%d\t%d\t%d\t%d\t%g\n", static_cast<int>(r + 1), n_int, el, len,
                                           static_cast<double>(ratio)));
    }
    bool wrote = std::fwrite(out.data(), 1, out.size(), stdout) == out.size();      // the ref list is a data channel (palace:477)
    if (argc >= 10) {
        // get_ref_by_index.py:6-89: the first integer of a stdout line is taken as the 1-based ROW of <db>.fai, i.e. the
        // record number among ALL records of the FASTA (a DB with records of <= 32 bases therefore names the wrong
        // record, as in the reference pipeline: SURVEY.md row E5); name = header up to the first white space, sequence with
        // the line ends removed, percentage = Python's str(float(<ratio text>)); ascending index order.
        std::ofstream fa(argv[8], std::ios::binary), pc(argv[9], std::ios::binary);
        if (!fa || !pc) { std::cerr << "eref: cannot write " << argv[8] << " / " << argv[9] << "\n"; return 1; }
        std::vector<int64_t> by_ordinal(static_cast<size_t>(db_all.n()) + 2, -1);
        for (int64_t i = 0; i < db_all.n(); i++)
            if (db_all.ordinal[i] >= 1 && db_all.ordinal[i] < static_cast<int64_t>(by_ordinal.size())) by_ordinal[static_cast<size_t>(db_all.ordinal[i])] = i;
        for (int64_t r = 0; r < n_refs; r++) {
            const int el = rows[4 * r + 1], len = rows[4 * r + 2];
            const float ratio = static_cast<float>(el) / static_cast<float>(len);
            if (!(el > 0 && ratio > 0.75)) continue;
            const int64_t row = r + 1;
            if (row >= static_cast<int64_t>(by_ordinal.size()) || by_ordinal[static_cast<size_t>(row)] < 0) continue;   // "Index not found in FAI file"
            const int64_t i = by_ordinal[static_cast<size_t>(row)];
            std::snprintf(line, sizeof line, "%g", static_cast<double>(ratio));
            std::string pct = line;
            if (pct.find_first_of(".en") == std::string::npos) pct += ".0";      // float("1") prints as 1.0
            fa << '>' << db_all.ids[static_cast<size_t>(i)] << '\n';
            const uint8_t *all_bases = all_long ? db.bases.data() : db_all.bases.data();
            std::string seq(reinterpret_cast<const char *>(all_bases + db_all.offsets[i]), static_cast<size_t>(db_all.len(i)));
            seq.erase(std::remove_if(seq.begin(), seq.end(), [](char ch) { return std::isspace(static_cast<unsigned char>(ch)); }), seq.end());
            fa << seq << '\n';                                                  // (Bio.SeqIO drops white space inside sequence lines)
            pc << db_all.ids[static_cast<size_t>(i)] << '\t' << pct << '\n';
        }
        fa.close(); pc.close();
        if (!fa.good() || !pc.good()) { std::cerr << "eref: write to " << argv[8] << " / " << argv[9] << " failed\n"; wrote = false; }
    }
    tr.lap("stdout");
    if (std::fflush(stdout) != 0 || std::ferror(stdout)) wrote = false;
    if (!wrote) {                                            // disk full, closed pipe: a cut-off ref list must not look like success
        std::cerr << "eref: writing the results failed: " << std::strerror(errno) << "\n";
        std::fflush(nullptr);
        _exit(1);
    }
    fast_exit.done(0);          // every output is complete: the caller goes on; runtime and gigabytes of mappings are torn down behind it
}
