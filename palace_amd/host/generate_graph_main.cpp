// generateGraph -- drop-in for the reference executable of the same name
// (bin/generate_graph.cpp; call site palace:557-560):
//     generateGraph [options] <BAM> <FASTG_FAI> <OUT> <AverageDepth>
// Host side: BGZF/BAM decode to columns, name tables, text output.  All per-record and
// per-evidence work runs in HIP through libpalace_hip.so (palace_graph_classify / _resolve);
// there is no CPU path for it.
#include <getopt.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <fstream>
#include <iostream>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "bam.hpp"
#include "bam_device.hpp"
#include "depth_host.hpp"
#include "fastx.hpp"
#include "stage04_fused.hpp"
#include "trace.hpp"
#include "fast_exit.hpp"
#include "device_pick.hpp"
#include "debug_trace.hpp"

using namespace palace_host;

namespace {

void usage(const char *prog)            // same option surface as generate_graph.cpp:542-556
{
    std::cerr << "Usage: " << prog << " [options] <BAM> <FASTG_FAI> <OUT> <AverageDepth>\n"
              << "Options:\n"
              << "  -e <int>      MAX_END (default: 300)\n"
              << "  -q <int>      MIN_MAPQ (default: 0)\n"
              << "  -n <int>      MAX_NM (default: 5)\n"
              << "  -p <double>   MIN_MATCH_FRAC (accepted, unused as in the reference)\n"
              << "  -P <0/1>      Enable paired-end evidence (default: 1)\n"
              << "  --max-span-frac <double>  (default: 0.80)\n"
              << "  --both-order <0/1>        Output both orders (default: 0)\n"
              << "  --lib <FR|RF|FF>          Library type (accepted, unused as in the reference)\n"
              << "  --min-count <int>         Minimum supporting reads (default: 5)\n"
              << "  --min-score <double>      (accepted, unused as in the reference)\n"
              << "  --debug                   JUNC lines carry their supporting reads (' READS: name(flag) ...') and the per-read\n"
              << "                            text of the reference's debug mode goes to stderr\n"
              << "Stage 04 in this process (optional; every file of palace:566-600, none read back):\n"
              << "  --hit-seqs F --node-scores F --blast F --fasta-fai F --paths F   inputs of filter_graph.py (+ --blast-ratio, --score-threshold: 0.7)\n"
              << "  --filtered-pre F --filtered F --all-hit-segs F                  its outputs (F after uniq)\n"
              << "  --linear F --cycle F --cycle-nodup F --all-result F              matching / remove_cycle_dup.py / cat outputs\n"
              << "  -s  -b  -i <int>  --aggressive                                   matching options (palace:587-590)\n";
}

#define CK(call)                                                                        \
    do {                                                                                \
        int rc__ = (call);                                                              \
        if (rc__ != 0) {                                                                \
            std::cerr << "generateGraph: " #call " failed: " << palace_last_error() << "\n"; \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

template <class T, class A>
int upload(palace_ctx *ctx, const std::vector<T, A> &v, T **d)
{
    void *p = nullptr;
    int rc = palace_malloc(ctx, std::max<size_t>(1, v.size()) * sizeof(T), &p);
    if (rc) return rc;
    *d = static_cast<T *>(p);
    return palace_h2d(ctx, p, v.data(), v.size() * sizeof(T));
}

void radix_sort_u64(std::vector<uint64_t> &v);

// ascending order of 64-bit keys by least-significant-digit radix passes of 11 bits (only over the bits in use)
void radix_sort_u64(std::vector<uint64_t> &v)
{
    uint64_t all = 0;
    for (uint64_t x : v) all |= x;
    std::vector<uint64_t> tmp(v.size());
    for (int shift = 0; shift < 64 && (all >> shift) != 0; shift += 11) {
        size_t count[2049] = {0};
        for (uint64_t x : v) count[((x >> shift) & 2047) + 1]++;
        for (int b = 0; b < 2048; b++) count[b + 1] += count[b];
        for (uint64_t x : v) tmp[count[(x >> shift) & 2047]++] = x;
        v.swap(tmp);
    }
}

// parseFastgFile (generate_graph.cpp:119-169) reduced to the pairs whose two names are BAM targets.  The file is mapped
// and cut into parts at line ends; every part is parsed by a thread with string views (no per-line allocation).
constexpr size_t kLineItems = 32;
std::vector<uint64_t> fastg_keys(const std::string &path, const BamColumns &c, int threads)
{
    MappedText txt;
    try { txt.open(path); } catch (const std::exception &) { return {}; }     // the reference reads an unopenable file as empty
    // (names are looked up with BamColumns::tid_of_hashed: the last duplicate of a target name wins, :624-627)
    const size_t N = txt.size;
    std::vector<size_t> cut{0};
    const size_t n_parts = static_cast<size_t>(std::max(1, threads)) * 4;
    for (size_t k = 1; k < n_parts; k++) {
        size_t p = std::max(cut.back(), N * k / n_parts);
        const void *nl = p < N ? std::memchr(txt.data + p, '\n', N - p) : nullptr;
        cut.push_back(nl ? static_cast<const char *>(nl) - txt.data + 1 : N);
    }
    cut.push_back(N);
    std::vector<std::vector<uint64_t>> part(cut.size() - 1);
    pool_for(part.size(), threads, [&](size_t k) {
        auto &keys = part[k];
        for (size_t p = cut[k]; p < cut[k + 1];) {
            const void *nl = std::memchr(txt.data + p, '\n', cut[k + 1] - p);
            const size_t e = nl ? static_cast<const char *>(nl) - txt.data : cut[k + 1];
            std::string_view line(txt.data + p, e - p);
            p = nl ? e + 1 : cut[k + 1];
            std::string_view head = line.substr(0, line.find(';'));           // getline(ss, fullName, ';')
            const size_t colon = head.find(':');
            if (colon == std::string_view::npos) continue;                     // no linked contigs
            // the names of the line first (hashes made, table slots asked for), then their look-ups: a table of a million names is a
            // cache miss per look-up, and a line has four of them on average
            struct Item { std::string_view n; uint64_t h; bool rev; };
            Item item[kLineItems];
            size_t n_items = 0;
            auto add = [&](std::string_view n) {
                const bool r = !n.empty() && n.back() == '\'';
                if (r) n.remove_suffix(1);
                item[n_items] = Item{n, hash_bytes(n), r};
                c.tid_names.prefetch(item[n_items].h);
                n_items++;
            };
            add(head.substr(0, colon));
            auto flush = [&](size_t from) {                                    // links item[from ..) of the line's own contig item[0]
                const int64_t a = c.tid_of_hashed(item[0].n, item[0].h);
                const bool rev = item[0].rev;
                for (size_t i = from; i < n_items; i++) {
                    const int64_t b = c.tid_of_hashed(item[i].n, item[i].h);
                    if (a < 0 || b < 0) continue;
                    const uint64_t o1 = rev ? 1 : 0, o2 = (rev != item[i].rev) ? 1 : 0;  // :151-157
                    keys.push_back((static_cast<uint64_t>(a) << 33) | (static_cast<uint64_t>(b) << 2) | (o1 << 1) | o2);
                    keys.push_back((static_cast<uint64_t>(b) << 33) | (static_cast<uint64_t>(a) << 2) | ((o1 ^ 1) << 1) | (o2 ^ 1));
                }
                n_items = 1;
            };
            for (size_t q = colon + 1; q < head.size();) {
                const size_t comma = head.find(',', q);
                std::string_view lk = head.substr(q, comma == std::string_view::npos ? std::string_view::npos : comma - q);
                q = comma == std::string_view::npos ? head.size() : comma + 1;
                if (lk.empty()) continue;
                add(lk);
                if (n_items == kLineItems) flush(1);
            }
            flush(1);
        }
    });
    // all keys in ascending order, each once: the parts' keys are dealt into 256 ranges of the key space (by their top bits: the left
    // contig, evenly spread), every range is sorted and made unique by a thread, the ranges are put behind one another
    uint64_t top = 0;
    size_t total = 0;
    for (auto &v : part) { total += v.size(); for (uint64_t x : v) top |= x; }
    int shift = 0;
    while ((top >> shift) >= 256) shift++;
    constexpr size_t kRanges = 256;
    std::vector<std::vector<size_t>> at(part.size(), std::vector<size_t>(kRanges + 1, 0));
    pool_for(part.size(), threads, [&](size_t k) { for (uint64_t x : part[k]) at[k][(x >> shift) + 1]++; });
    std::vector<size_t> first(kRanges + 1, 0);
    for (size_t r = 0; r < kRanges; r++) {
        size_t n = 0;
        for (size_t k = 0; k < part.size(); k++) { const size_t c = at[k][r + 1]; at[k][r + 1] = first[r] + n; n += c; }   // at[k][r + 1]: where part k writes its keys of range r
        first[r + 1] = first[r] + n;
    }
    std::vector<uint64_t> dealt(total);
    pool_for(part.size(), threads, [&](size_t k) { for (uint64_t x : part[k]) dealt[at[k][(x >> shift) + 1]++] = x; });
    std::vector<size_t> kept(kRanges, 0);
    pool_for(kRanges, threads, [&](size_t r) {
        std::vector<uint64_t> v(dealt.begin() + static_cast<std::ptrdiff_t>(first[r]), dealt.begin() + static_cast<std::ptrdiff_t>(first[r + 1]));
        radix_sort_u64(v);
        v.erase(std::unique(v.begin(), v.end()), v.end());
        std::copy(v.begin(), v.end(), dealt.begin() + static_cast<std::ptrdiff_t>(first[r]));
        kept[r] = v.size();
    });
    std::vector<uint64_t> keys;
    keys.reserve(total);
    for (size_t r = 0; r < kRanges; r++) keys.insert(keys.end(), dealt.begin() + static_cast<std::ptrdiff_t>(first[r]), dealt.begin() + static_cast<std::ptrdiff_t>(first[r] + kept[r]));
    return keys;
}

// dense rank of every target name in byte order (the `cR < cL` test :856 and the output order :1048): sort by the first
// eight bytes as one big-endian word, finish ties with the full comparison
void name_ranks(const std::vector<std::string> &names, std::vector<int32_t> &by_name, std::vector<int32_t> &rank)
{
    const int32_t nt = static_cast<int32_t>(names.size());
    struct Key { uint64_t hi, lo; int32_t id; };
    std::vector<Key> key(static_cast<size_t>(nt));
    for (int32_t i = 0; i < nt; i++) {
        uint64_t k[2] = {0, 0};
        const std::string &s = names[static_cast<size_t>(i)];
        for (size_t b = 0; b < 16; b++) k[b >> 3] = (k[b >> 3] << 8) | (b < s.size() ? static_cast<unsigned char>(s[b]) : 0);
        key[static_cast<size_t>(i)] = {k[0], k[1], i};
    }
    auto before = [&](const Key &a, const Key &b) {
        if (a.hi != b.hi) return a.hi < b.hi;
        if (a.lo != b.lo) return a.lo < b.lo;
        const int c = names[static_cast<size_t>(a.id)].compare(names[static_cast<size_t>(b.id)]);   // (a NUL inside a name also lands here)
        return c != 0 ? c < 0 : a.id < b.id;
    };
    // eight slices sorted side by side, then merged pairwise (a strict total order: the result does not depend on the slicing)
    const size_t parts = nt < (1 << 16) ? 1 : 8;
    auto cut = [&](size_t k) { return key.begin() + static_cast<std::ptrdiff_t>(static_cast<size_t>(nt) * k / parts); };
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < parts; k++) pool.emplace_back([&, k] { std::sort(cut(k), cut(k + 1), before); });
        for (auto &t : pool) t.join();
    }
    for (size_t width = 1; width < parts; width *= 2) {
        std::vector<std::thread> pool;
        for (size_t k = 0; k + width < parts; k += 2 * width)
            pool.emplace_back([&, k] { std::inplace_merge(cut(k), cut(k + width), cut(std::min(parts, k + 2 * width)), before); });
        for (auto &t : pool) t.join();
    }
    by_name.resize(static_cast<size_t>(nt));
    rank.assign(static_cast<size_t>(nt), 0);
    for (int32_t k = 0, r = -1; k < nt; k++) {
        by_name[static_cast<size_t>(k)] = key[static_cast<size_t>(k)].id;
        if (k == 0 || names[static_cast<size_t>(by_name[k])] != names[static_cast<size_t>(by_name[k - 1])]) r++;
        rank[static_cast<size_t>(by_name[k])] = r;
    }
}


}  // namespace

int main(int argc, char **argv)
{
    palace_graph_params prm{300, 0, 5, 1, 0, 0, 0.80};
    int min_count = 5;
    bool debug = false;
    static struct option long_opts[] = {{"max-span-frac", required_argument, 0, 1000},
                                        {"both-order", required_argument, 0, 1001},
                                        {"lib", required_argument, 0, 1002},
                                        {"min-count", required_argument, 0, 1003},
                                        {"min-score", required_argument, 0, 1004},
                                        {"debug", no_argument, 0, 1005},
                                        {"hit-seqs", required_argument, 0, 1100}, {"node-scores", required_argument, 0, 1101},
                                        {"blast", required_argument, 0, 1102}, {"fasta-fai", required_argument, 0, 1103},
                                        {"paths", required_argument, 0, 1104}, {"blast-ratio", required_argument, 0, 1105},
                                        {"score-threshold", required_argument, 0, 1106}, {"filtered-pre", required_argument, 0, 1107},
                                        {"filtered", required_argument, 0, 1108}, {"all-hit-segs", required_argument, 0, 1109},
                                        {"linear", required_argument, 0, 1110}, {"cycle", required_argument, 0, 1111},
                                        {"cycle-nodup", required_argument, 0, 1112}, {"all-result", required_argument, 0, 1113},
                                        {"aggressive", no_argument, 0, 1114},
                                        {0, 0, 0, 0}};
    Stage04Options s4o;
    int opt, li = 0;
    while ((opt = getopt_long(argc, argv, "e:q:n:p:P:sbi:", long_opts, &li)) != -1) {
        switch (opt) {                                       // clamps as generate_graph.cpp:575-590
        case 'e': prm.max_end = std::max(1, std::atoi(optarg)); break;
        case 'q': prm.min_mapq = std::max(0, std::atoi(optarg)); break;
        case 'n': prm.max_nm = std::max(0, std::atoi(optarg)); break;
        case 'p': break;
        case 'P': prm.enable_paired = std::atoi(optarg) != 0; break;
        case 1000: prm.max_span_frac = std::min(0.99, std::max(0.1, std::atof(optarg))); break;
        case 1001: prm.both_order = std::atoi(optarg) != 0; break;
        case 1002: {
            std::string v = optarg;
            if (!(v == "FR" || v == "RF" || v == "FF")) std::cerr << "Unknown --lib " << v << ", fallback FR\n";
            break;
        }
        case 1003: min_count = std::max(1, std::atoi(optarg)); break;
        case 1004: break;
        case 1005: debug = true; break;
        case 1100: s4o.gene_file = optarg; break;
        case 1101: s4o.score_file = optarg; break;
        case 1102: s4o.blast_file = optarg; break;
        case 1103: s4o.fasta_fai = optarg; break;
        case 1104: s4o.paths_file = optarg; break;
        case 1105: s4o.blast_ratio = std::atof(optarg); break;
        case 1106: s4o.score_threshold = std::atof(optarg); break;
        case 1107: s4o.pre_out = optarg; break;
        case 1108: s4o.filtered_out = optarg; break;
        case 1109: s4o.hit_segs_out = optarg; break;
        case 1110: s4o.linear_out = optarg; break;
        case 1111: s4o.cycle_out = optarg; break;
        case 1112: s4o.nodup_out = optarg; break;
        case 1113: s4o.result_out = optarg; break;
        case 1114: s4o.aggressive = true; break;
        case 's': s4o.self_loops = true; break;
        case 'b': s4o.break_cycles = true; break;
        case 'i': s4o.iterations = std::max(1, std::atoi(optarg)); break;
        default: usage(argv[0]); return 1;
        }
    }
    if (argc - optind < 4) { usage(argv[0]); return 1; }
    if (s4o.enabled() && (s4o.gene_file.empty() || s4o.score_file.empty() || s4o.blast_file.empty() || s4o.fasta_fai.empty() || s4o.paths_file.empty())) {
        std::cerr << "generateGraph: stage 04 needs --hit-seqs, --node-scores, --blast, --fasta-fai and --paths\n";
        return 1;
    }
    if (s4o.enabled() && debug) {                 // (filter_graph.py copies JUNC lines as they are: the read lists would have to travel through stage 04)
        std::cerr << "generateGraph: --debug goes with the plain four-argument call, not with the stage-04 outputs\n";
        return 1;
    }
    const std::string bam_path = argv[optind], fai_path = argv[optind + 1], out_path = argv[optind + 2];
    // <avgDepth> = "auto": the depth stage (palace:538-552) is done here, on the records this run decodes anyway; the value
    // goes through the same text the driver would have passed ("%.6g" of awk, then atof)
    const bool auto_depth = std::string(argv[optind + 3]) == "auto";
    double avg_depth = auto_depth ? 0.0 : std::atof(argv[optind + 3]);

    FastExit fast_exit = fast_exit_begin();                    // from here on this is the worker process (fast_exit.hpp)
    const int device = pick_device();                          // PALACE_DEVICE (device_pick.hpp): before anything touches HIP
    Trace tr("generateGraph");
    BamColumns c;
    c.want_match_segments = auto_depth;                         // (only the depth stage reads them: 6.7 M triples at 1M contigs)
    const int threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    uint64_t seed = 1;
    BamLoad *load = nullptr;
    try {
        // header parsed, the rest of the file is being inflated -- by the threads, and by the device once its runtime is up (bam_device.hpp)
        load = load_bam_begin(bam_path, threads, c, device_inflate_helpers(device));
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    tr.lap("bam header");
    const int32_t nt = static_cast<int32_t>(c.target_name.size());
    // what depends on the target names only runs beside the inflate / decode of the records: the name ranks, the FASTG
    // keys, and the HIP runtime coming up
    std::vector<int32_t> by_name, rank;
    std::vector<uint64_t> fkeys;
    palace_ctx *ctx = nullptr;
    int ctx_rc = 0;
    std::string ctx_err;
    Stage04Side s4side;
    std::thread side04;
    auto join04 = [&] { if (side04.joinable()) side04.join(); };
    std::thread side([&] { Trace t("generateGraph/names"); name_ranks(c.target_name, by_name, rank); t.lap("name ranks"); });
    std::thread side2([&] { Trace t("generateGraph/fastg"); fkeys = fastg_keys(fai_path, c, 8); t.lap("fastg keys"); });
    std::thread hip_up([&] {
        ctx_rc = palace_ctx_create(device, &ctx);
        if (ctx_rc) ctx_err = palace_last_error();
    });
    palace_stage04 *s4obj = nullptr;
    std::string s4err;
    if (s4o.enabled())                                            // beside the inflate, like the rest: the side files, then (once the
        side04 = std::thread([&] {                               // name ranks are there) the resident object and its scratch on the device
            stage04_read_side_files(s4o, c, s4side);
            side.join();
            bool unique_names = true;
            for (int32_t k = 1; k < nt && unique_names; k++) unique_names = rank[by_name[k]] != rank[by_name[k - 1]];
            if (!unique_names) { s4err = "stage 04 in this process needs distinct target names; run the stages separately"; return; }
            s4obj = stage04_prepare(s4o, c, s4side, rank, min_count, std::max<int64_t>(1 << 20, static_cast<int64_t>(load_bam_size_hint(load)) / 2400), s4err);
        });
    try {
        load_bam_finish(load, seed);
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        join04(); if (side.joinable()) side.join(); side2.join(); hip_up.join();
        return 1;
    }
    tr.lap("bam records");
    if (debug) {
        // the per-read text of the reference's --debug (:454-458, :607-609, :711-853): a diagnostic the host writes from the decoded
        // records (debug_trace.hpp); the graph below is the device's as without the option
        const std::string text = debug_trace(c, fai_path, prm);
        std::fwrite(text.data(), 1, text.size(), stderr);
        tr.lap("--debug: per-read text");
    }
    join04();                                                     // (it joined `side`)
    if (side.joinable()) side.join();
    side2.join();
    if (s4o.enabled() && !s4obj) { std::cerr << "generateGraph: stage 04: " << s4err << "\n"; hip_up.join(); return 1; }
    tr.lap("name ranks + fastg keys (joined)");
    hip_up.join();
    if (ctx_rc) { std::cerr << "generateGraph: cannot set up the GPU: " << ctx_err << "\n"; join04(); return 1; }
    tr.lap("hip runtime up (joined)");
    if (auto_depth) {
        std::string text;
        const int drc = first_depth(ctx, c, text);
        if (drc < 0) { std::cerr << "generateGraph: " << palace_last_error() << "\n"; return 1; }
        if (drc > 0) { std::cerr << "generateGraph: no position is covered, cannot derive avgDepth\n"; return 1; }
        avg_depth = std::atof(text.c_str());
        std::cerr << "Average sequencing depth: " << text << "\n";               // the driver logs the same line (palace:551)
        tr.lap("depth stage");
    }
    palace_bam_cols cols{};
    cols.n = c.n();
    int32_t *d_tid, *d_pos, *d_mtid, *d_mpos, *d_nm, *d_rl, *d_ql, *d_cs, *d_ce, *d_sao, *d_tlen, *d_rank;
    uint16_t *d_flag; uint8_t *d_mapq; uint64_t *d_qkey, *d_fk, *d_consumed; palace_sa_item *d_sa;
    CK(upload(ctx, c.tid, &d_tid)); CK(upload(ctx, c.pos, &d_pos)); CK(upload(ctx, c.mtid, &d_mtid));
    CK(upload(ctx, c.mpos, &d_mpos)); CK(upload(ctx, c.nm, &d_nm)); CK(upload(ctx, c.ref_len, &d_rl));
    CK(upload(ctx, c.read_len, &d_ql)); CK(upload(ctx, c.clip_s, &d_cs)); CK(upload(ctx, c.clip_e, &d_ce));
    CK(upload(ctx, c.sa_off, &d_sao)); CK(upload(ctx, c.flag, &d_flag)); CK(upload(ctx, c.mapq, &d_mapq));
    CK(upload(ctx, c.qkey, &d_qkey)); CK(upload(ctx, c.sa, &d_sa)); CK(upload(ctx, c.target_len, &d_tlen));
    CK(upload(ctx, rank, &d_rank)); CK(upload(ctx, fkeys, &d_fk));
    cols.tid = d_tid; cols.pos = d_pos; cols.mtid = d_mtid; cols.mpos = d_mpos; cols.nm = d_nm;
    cols.ref_len = d_rl; cols.read_len = d_ql; cols.clip_s = d_cs; cols.clip_e = d_ce; cols.sa_off = d_sao;
    cols.flag = d_flag; cols.mapq = d_mapq; cols.qkey = d_qkey;
    void *p = nullptr;
    CK(palace_malloc(ctx, std::max<size_t>(1, nt) * 8, &p));
    d_consumed = static_cast<uint64_t *>(p);
    const int64_t cand_cap = c.n() + static_cast<int64_t>(c.sa.size()) + 1;      // worst case: every record, every item
    CK(palace_malloc(ctx, static_cast<size_t>(cand_cap) * sizeof(palace_graph_cand), &p));
    palace_graph_cand *d_cands = static_cast<palace_graph_cand *>(p);

    // per-contig offsets into the sorted FASTG keys: a candidate's look-up starts inside its left contig's links
    CK(palace_malloc(ctx, (static_cast<size_t>(nt) + 1) * 4, &p));
    uint32_t *d_fk_first = static_cast<uint32_t *>(p);
    CK(palace_graph_fastg_offsets(ctx, d_fk, static_cast<int64_t>(fkeys.size()), nt, d_fk_first));
    tr.lap("uploads");
    std::vector<palace_graph_cand> cands;
    int64_t n_cands = 0;
    for (int attempt = 0;; attempt++) {
        CK(palace_memset(ctx, d_consumed, 0, std::max<size_t>(1, nt) * 8));
        CK(palace_graph_classify_ix(ctx, &cols, d_sa, nt, d_tlen, d_rank, d_fk, static_cast<int64_t>(fkeys.size()), d_fk_first, &prm,
                                    0, d_consumed, d_cands, cand_cap, &n_cands, nullptr));
        // exactness guard: among pair candidates equal keys must mean equal read names
        cands.resize(static_cast<size_t>(n_cands));
        CK(palace_d2h(ctx, cands.data(), d_cands, cands.size() * sizeof(palace_graph_cand)));
        // (a flat open-addressing table: the node-based map took ~30 ms for the 0.3 M pair candidates of a 1M-contig sample)
        size_t n_pair = 0;
        for (const auto &k : cands) n_pair += k.kind == 1;
        size_t cap_t = 1024;
        while (cap_t < 2 * n_pair) cap_t <<= 1;
        std::vector<uint64_t> g_key(cap_t);
        std::vector<int64_t> g_ord(cap_t, -1);                             // -1: empty slot
        bool collision = false;
        for (const auto &k : cands) {
            if (k.kind != 1) continue;
            size_t at = static_cast<size_t>((k.qkey * 0x9E3779B97F4A7C15ull) >> 20) & (cap_t - 1);
            while (g_ord[at] >= 0 && g_key[at] != k.qkey) at = (at + 1) & (cap_t - 1);
            if (g_ord[at] < 0) { g_key[at] = k.qkey; g_ord[at] = k.ord; }
            else if (g_ord[at] != k.ord && c.qname(g_ord[at]) != c.qname(k.ord)) { collision = true; break; }
        }
        if (!collision) break;
        if (attempt == 8) { std::cerr << "generateGraph: read-name key collisions persist\n"; return 1; }
        rekey(c, ++seed);
        CK(palace_h2d(ctx, d_qkey, c.qkey.data(), c.qkey.size() * 8));
    }
    tr.lap("classify + name guard");
    const int64_t n_records = c.n();
    CK(palace_malloc(ctx, static_cast<size_t>(std::max<int64_t>(1, n_cands)) * sizeof(palace_graph_edge), &p));
    palace_graph_edge *d_edges = static_cast<palace_graph_edge *>(p);
    int64_t n_edges = 0;
    tr.lap("edge buffer");
    CK(palace_graph_resolve(ctx, d_cands, n_cands, n_records, &prm, d_consumed, d_edges, std::max<int64_t>(1, n_cands), &n_edges));
    tr.lap("resolve");
    if (debug) CK(palace_d2h(ctx, cands.data(), d_cands, cands.size() * sizeof(palace_graph_cand)));   // classes are final now (the host-scored ones too)
    std::vector<uint64_t> consumed(static_cast<size_t>(nt));
    std::vector<palace_graph_edge> edges(static_cast<size_t>(n_edges));
    std::vector<int32_t> cn_dev(static_cast<size_t>(nt));
    CK(palace_malloc(ctx, std::max<size_t>(1, nt) * 4, &p));
    int32_t *d_cn = static_cast<int32_t *>(p);
    tr.lap("host buffers");
    CK(palace_graph_copy_numbers(ctx, d_consumed, d_tlen, nt, avg_depth, d_cn));
    CK(palace_d2h(ctx, cn_dev.data(), p, cn_dev.size() * 4));
    CK(palace_d2h(ctx, consumed.data(), d_consumed, consumed.size() * 8));
    CK(palace_d2h(ctx, edges.data(), d_edges, edges.size() * sizeof(palace_graph_edge)));
    tr.lap("copy numbers + d2h");
    // (The inflated stream -- gigabytes -- and the per-record columns have served by now.  Rounds 2-3 handed their pages back
    // from helper threads so that the process would exit faster; whatever ran beside those helpers paid for it (TLB shoot-downs,
    // the address-space lock: hipMalloc 25 ms, SEG formatting +45 ms).  Now nothing is handed back: the caller's process leaves
    // when the outputs are complete and this worker's address space is torn down behind it, fast_exit.hpp.)
    if (!s4o.enabled()) {
        palace_ctx_destroy(ctx);
        tr.lap("ctx destroy");
    }

    // --debug (:1068-1073): the reads behind every junction.  The counting is the device's (palace_graph_resolve); which
    // candidates it counted is restated here from its three rules (graph.hip, resolve_*_kernel) and checked against its counts.
    // Order: as the reference meets the evidence -- records in file order, the SA items of a record in list order.
    std::vector<std::string> reads_of(debug ? edges.size() : 0);
    if (debug) {
        std::vector<char> has_split(static_cast<size_t>(n_records) + 1, 0);
        for (const auto &k : cands)
            if (k.kind == 0 && k.cls == 1) has_split[static_cast<size_t>(k.ord)] = 1;          // :874
        std::unordered_map<uint64_t, int64_t> first_of;                                        // read name (exact key) -> first record with a layout (:938)
        for (const auto &k : cands)
            if (k.kind == 1 && k.found && !has_split[static_cast<size_t>(k.ord)]) {
                auto at = first_of.try_emplace(k.qkey, k.ord).first;
                at->second = std::min(at->second, k.ord);
            }
        std::vector<uint32_t> counted;
        for (size_t i = 0; i < cands.size(); i++) {
            const auto &k = cands[i];
            if (k.kind == 0) { if (k.cls == 1) counted.push_back(static_cast<uint32_t>(i)); continue; }
            if (has_split[static_cast<size_t>(k.ord)]) continue;                               // :887
            const auto at = first_of.find(k.qkey);
            if (at != first_of.end() && at->second < k.ord) continue;                          // :890-893
            if (k.found && k.cls == 1) counted.push_back(static_cast<uint32_t>(i));
        }
        std::sort(counted.begin(), counted.end(), [&](uint32_t a, uint32_t b) {
            return cands[a].ord != cands[b].ord ? cands[a].ord < cands[b].ord : cands[a].sa_index < cands[b].sa_index;
        });
        auto key_of = [](int32_t l, int32_t r, int oL, int oR) {
            return (static_cast<uint64_t>(static_cast<uint32_t>(l)) << 33) | (static_cast<uint64_t>(static_cast<uint32_t>(r)) << 2) | static_cast<uint64_t>((oL ? 2 : 0) | (oR ? 1 : 0));
        };
        std::unordered_map<uint64_t, uint32_t> edge_at;
        for (size_t e = 0; e < edges.size(); e++) edge_at[key_of(edges[e].left, edges[e].right, edges[e].oL, edges[e].oR)] = static_cast<uint32_t>(e);
        std::vector<uint32_t> listed(edges.size(), 0);
        for (uint32_t i : counted) {
            const auto &k = cands[i];
            const auto at = edge_at.find(key_of(k.left, k.right, k.oL, k.oR));
            if (at == edge_at.end()) { std::cerr << "generateGraph: --debug: evidence without an edge\n"; return 1; }
            reads_of[at->second] += " " + c.qname(k.ord) + "(" + std::to_string(c.flag[static_cast<size_t>(k.ord)]) + ")";
            listed[at->second]++;
        }
        for (size_t e = 0; e < edges.size(); e++)
            if (listed[e] != edges[e].counts[0] + edges[e].counts[1] + edges[e].counts[2] + edges[e].counts[3]) {
                std::cerr << "generateGraph: --debug: the read lists disagree with the device's counts\n";
                return 1;
            }
    }

    // ---- text output (generate_graph.cpp:1019-1076) ----
    FILE *out = std::fopen(out_path.c_str(), "w");
    if (!out) { std::cerr << "Failed to open output " << out_path << "\n"; return 1; }
    std::vector<char> big(1 << 22);
    std::setvbuf(out, big.data(), _IOFBF, big.size());
    // `_graph.txt` order of the edges, as an index into the device order (stage 04 flags edges in device order)
    std::vector<uint32_t> sorted_index(edges.size());
    std::iota(sorted_index.begin(), sorted_index.end(), 0u);
    std::sort(sorted_index.begin(), sorted_index.end(), [&](uint32_t ia, uint32_t ib) {
        const palace_graph_edge &a = edges[ia], &b = edges[ib];
        if (rank[a.left] != rank[b.left]) return rank[a.left] < rank[b.left];
        if (rank[a.right] != rank[b.right]) return rank[a.right] < rank[b.right];
        if (a.oL != b.oL) return a.oL < b.oL;                          // '+' (43) < '-' (45)
        return a.oR < b.oR;
    });
    // SEG lines in name order; formatted by all threads (a slice of the order each), written in order
    std::vector<std::string> seg_text(static_cast<size_t>(threads) * 4);
    struct SegAt { int32_t tid; size_t at, len; };
    std::vector<std::vector<SegAt>> seg_at(seg_text.size());
    std::vector<std::atomic<int>> seg_ready(seg_text.size());
    for (auto &f : seg_ready) f.store(0, std::memory_order_relaxed);
    // The file itself (60 MB at a million contigs) is written by a thread of its own, part by part as the formatting threads finish
    // them (and, with stage 04 in the process, while that goes on): edges, their order and the names are final
    bool graph_ok = false;
    std::thread graph_writer([&] {
        for (size_t k = 0; k < seg_text.size(); k++) {
            while (!seg_ready[k].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
            std::fwrite(seg_text[k].data(), 1, seg_text[k].size(), out);
        }
        for (uint32_t ei : sorted_index) {
            const palace_graph_edge &e = edges[ei];
            const uint32_t supp = e.counts[0], supp_nf = e.counts[1], span = e.counts[2], span_nf = e.counts[3];
            const uint32_t total = supp + supp_nf + span + span_nf;
            if (total == 0 || total < static_cast<uint32_t>(min_count)) continue;   // :1056-1061
            std::fprintf(out, "JUNC %s %c %s %c %u %u", c.target_name[e.left].c_str(), e.oL ? '-' : '+',
                         c.target_name[e.right].c_str(), e.oR ? '-' : '+', supp + span + supp_nf, span_nf);
            if (debug) { std::fputs(" READS:", out); std::fwrite(reads_of[ei].data(), 1, reads_of[ei].size(), out); }
            std::fputc('\n', out);
        }
        graph_ok = !(std::ferror(out) | std::fclose(out));   // a short write must not exit 0
    });
    pool_for(seg_text.size(), threads, [&](size_t part) {
        std::string &txt = seg_text[part];
        char line[512];
        const int32_t k0 = static_cast<int32_t>(static_cast<int64_t>(nt) * part / seg_text.size());
        const int32_t k1 = static_cast<int32_t>(static_cast<int64_t>(nt) * (part + 1) / seg_text.size());
        txt.reserve(static_cast<size_t>(k1 - k0) * 48);
        for (int32_t k = k0; k < k1; k++) {
            // (name order is not target order: every line's name, sums and lengths are cache misses in tables of a million entries --
            // the ones of the line sixteen ahead are asked for now, its name's characters eight ahead)
            if (k + 16 < k1) {
                const int32_t t = by_name[k + 16];
                __builtin_prefetch(&c.target_name[static_cast<size_t>(t)]);
                __builtin_prefetch(&consumed[static_cast<size_t>(t)]);
                __builtin_prefetch(&c.target_len[static_cast<size_t>(t)]);
                __builtin_prefetch(&cn_dev[static_cast<size_t>(t)]);
            }
            if (k + 8 < k1) __builtin_prefetch(c.target_name[static_cast<size_t>(by_name[k + 8])].data());
            // std::map keeps one entry per distinct name; a later duplicate overwrites an earlier one (:1033)
            if (k + 1 < nt && rank[by_name[k + 1]] == rank[by_name[k]]) continue;
            int32_t best = -1;
            for (int32_t j = k; j >= 0 && rank[by_name[j]] == rank[by_name[k]]; j--)
                if (c.target_len[by_name[j]] > 0) best = std::max(best, by_name[j]);
            if (best < 0) continue;                                       // L <= 0 targets are skipped (:1023)
            // refConsumed is keyed by NAME (:631, :659): duplicates share one sum
            double sum = 0.0;
            for (int32_t j = k; j >= 0 && rank[by_name[j]] == rank[by_name[k]]; j--) sum += static_cast<double>(consumed[by_name[j]]);
            const int32_t L = c.target_len[best];
            const double depth = sum / std::max(1, L);
            const bool unique_name = (k == 0 || rank[by_name[k - 1]] != rank[by_name[k]]);
            const double cnf = avg_depth > 0.0 ? depth / avg_depth : 0.0;
            const int cn = unique_name ? cn_dev[best] : static_cast<int>(std::floor(cnf + 0.5));   // device value; host only for duplicate names
            const std::string &nm = c.target_name[best];
            const size_t at = txt.size();
            // "SEG <name> <%g of depth> <cn>\n" without printf: a million lines, and %g of a double alone is ~0.4 us of snprintf
            // (format_g6, textio.hpp: the same characters, checked against snprintf on 3e7 values by `hostdump fmtg`)
            txt += "SEG ";
            txt += nm;
            size_t w = 0;
            line[w++] = ' ';
            w += format_g6(depth, line + w);
            line[w++] = ' ';
            w += static_cast<size_t>(std::to_chars(line + w, line + sizeof line - 2, cn).ptr - (line + w));
            line[w++] = '\n';
            txt.append(line, w);
            seg_at[part].push_back({best, at, txt.size() - at});
        }
        seg_ready[part].store(1, std::memory_order_release);               // the writer may take this part
    });
    auto graph_written = [&]() -> bool {
        if (graph_writer.joinable()) graph_writer.join();
        if (!graph_ok) std::cerr << "Error: failed writing " << out_path << "\n";
        return graph_ok;
    };
    if (!s4o.enabled() && !graph_written()) return 1;
    tr.lap(s4o.enabled() ? "text output (SEG text; the file is being written)" : "text output");
    if (s4o.enabled()) {
        std::vector<std::string_view> raw_seg(static_cast<size_t>(nt));
        for (size_t part = 0; part < seg_text.size(); part++)
            for (const SegAt &sg : seg_at[part]) raw_seg[static_cast<size_t>(sg.tid)] = std::string_view(seg_text[part]).substr(sg.at, sg.len);
        std::string err;
        if (stage04_run(ctx, s4obj, s4o, c, s4side, rank, raw_seg, edges, sorted_index, d_edges, n_cands, d_cn, threads, tr, err)) {
            std::cerr << "generateGraph: stage 04: " << err << "\n";
            graph_written();
            return 1;
        }
        if (!graph_written()) return 1;
        tr.lap("_graph.txt written (joined)");
    }
    fast_exit.done(0);          // the outputs are complete and closed: the caller goes on, gigabytes of host containers are torn down behind it
}
