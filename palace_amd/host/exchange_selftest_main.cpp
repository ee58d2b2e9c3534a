// exchange_selftest -- the C entry points of include/palace_rccl.h driven from C++ the way a one-process-per-GPU host would:
// count reads into the table, exchange it over an RCCL communicator, check the table.  With one GPU the communicator has one
// rank (the sends are copies to self): what this can check is the call sequence, the packing and the merge, not xGMI.
//     exchange_selftest [n_reads]      exit 0 and "ok ..." on success
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/palace_rccl.h"

#define CK(call)                                                                                   \
    do {                                                                                           \
        if ((call) != 0) { std::fprintf(stderr, "exchange_selftest: %s failed: %s\n", #call, palace_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const int64_t n_reads = argc > 1 ? std::atoll(argv[1]) : 20000;
    const int read_len = 100;
    std::mt19937_64 rng(7);
    std::vector<uint8_t> bases(static_cast<size_t>(n_reads) * read_len);
    std::vector<uint8_t> pool(200000);
    for (auto &b : pool) b = "ACGT"[rng() & 3];
    for (int64_t r = 0; r < n_reads; r++) {                   // reads from a small pool: many keys reach count 2 and 3
        const size_t at = rng() % (pool.size() - read_len);
        std::memcpy(bases.data() + r * read_len, pool.data() + at, read_len);
    }
    std::vector<int64_t> off(static_cast<size_t>(n_reads) + 1);
    for (int64_t r = 0; r <= n_reads; r++) off[static_cast<size_t>(r)] = r * read_len;
    uint8_t hdr[400] = {0};
    for (int z = 0; z < 32; z++) for (int i = 0; i < 3; i++) hdr[4 * (3 * z + i)] = static_cast<uint8_t>((i + z) % 3);
    palace_ctx *ctx = nullptr;
    CK(palace_ctx_create(0, &ctx));
    CK(palace_eref_set_coder(ctx, hdr));
    void *d_bases = nullptr, *d_off = nullptr;
    CK(palace_malloc(ctx, bases.size(), &d_bases)); CK(palace_malloc(ctx, off.size() * 8, &d_off));
    CK(palace_h2d(ctx, d_bases, bases.data(), bases.size())); CK(palace_h2d(ctx, d_off, off.data(), off.size() * 8));
    CK(palace_eref_table_reset(ctx));
    CK(palace_eref_count_reads(ctx, static_cast<const uint8_t *>(d_bases), static_cast<const int64_t *>(d_off), n_reads, nullptr, n_reads * read_len));
    uint64_t before[3], after[3];
    CK(palace_eref_table_popcounts(ctx, before));
    ncclComm_t comm;
    const int dev = 0;
    if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) { std::fprintf(stderr, "exchange_selftest: ncclCommInitAll failed\n"); return 1; }
    CK(palace_eref_table_exchange(ctx, comm, 0, 1));
    CK(palace_sync(ctx));
    CK(palace_eref_table_popcounts(ctx, after));
    // one rank: the merged table is the table (1 part: a + 0), and plane 3 is untouched by the gather
    if (std::memcmp(before, after, sizeof before) != 0 || before[2] == 0 || before[0] <= before[1]) {
        std::fprintf(stderr, "exchange_selftest: planes changed: %llu %llu %llu -> %llu %llu %llu\n", (unsigned long long)before[0],
                     (unsigned long long)before[1], (unsigned long long)before[2], (unsigned long long)after[0], (unsigned long long)after[1],
                     (unsigned long long)after[2]);
        return 1;
    }
    // the key space split one way: the share is everything, the gather leaves the plane as it is
    uint32_t share[4];
    CK(palace_eref_key_share(0, 1, share));
    if ((share[0] & share[1] & share[2] & share[3]) != ~0u) { std::fprintf(stderr, "exchange_selftest: a one-rank share is not the whole key space\n"); return 1; }
    CK(palace_eref_set_key_buckets(ctx, share));
    CK(palace_eref_key_share_gather(ctx, comm, 0, 1));
    CK(palace_sync(ctx));
    CK(palace_eref_table_popcounts(ctx, after));
    if (std::memcmp(before, after, sizeof before) != 0) { std::fprintf(stderr, "exchange_selftest: the share gather changed the planes\n"); return 1; }
    // ... and in sparse form: sized first (room 0: nothing is written, the total is reported), then with room
    unsigned long long need = 0, again = 0;
    CK(palace_eref_key_share_gather_sparse(ctx, comm, 0, 1, 0, &need));
    if (need != before[2]) { std::fprintf(stderr, "exchange_selftest: sparse gather sized %llu keys, plane 3 has %llu\n", need, (unsigned long long)before[2]); return 1; }
    CK(palace_eref_key_share_gather_sparse(ctx, comm, 0, 1, static_cast<int64_t>(need + 64), &again));
    CK(palace_eref_table_popcounts(ctx, after));
    if (again != need || std::memcmp(before, after, sizeof before) != 0) { std::fprintf(stderr, "exchange_selftest: the sparse share gather changed the planes\n"); return 1; }
    // rows: 5 refs, all owned by rank 0
    std::vector<int32_t> rows(20);
    for (int i = 0; i < 20; i++) rows[static_cast<size_t>(i)] = i * 3 + 1;
    void *d_rows = nullptr;
    CK(palace_malloc(ctx, 80, &d_rows)); CK(palace_h2d(ctx, d_rows, rows.data(), 80));
    const int64_t lo[1] = {0}, hi[1] = {5};
    CK(palace_eref_rows_allgather(ctx, comm, 0, 1, static_cast<int32_t *>(d_rows), 5, lo, hi));
    std::vector<int32_t> back(20);
    CK(palace_d2h(ctx, back.data(), d_rows, 80));
    if (back != rows) { std::fprintf(stderr, "exchange_selftest: rows changed\n"); return 1; }
    // the reads sharded, partial counts of the DB's entries exchanged: the pool as a one-ref DB, the reads counted with every entry of
    // its probe index tested (option probe_all_sets 2), a one-rank exchange (the sum of one part), the indexed scan -- against the
    // plain count and scan of the same reads
    {
        void *d_ref = nullptr, *d_roff = nullptr, *d_r1 = nullptr, *d_r2 = nullptr;
        const int64_t roff[2] = {0, static_cast<int64_t>(pool.size())};
        CK(palace_malloc(ctx, pool.size(), &d_ref)); CK(palace_malloc(ctx, 16, &d_roff)); CK(palace_malloc(ctx, 16, &d_r1)); CK(palace_malloc(ctx, 16, &d_r2));
        CK(palace_h2d(ctx, d_ref, pool.data(), pool.size())); CK(palace_h2d(ctx, d_roff, roff, 16));
        CK(palace_eref_set_count_mode(ctx, 2, 0));                                      // (the binned path, although the input is small)
        CK(palace_eref_table_reset(ctx));
        CK(palace_eref_count_reads(ctx, static_cast<const uint8_t *>(d_bases), static_cast<const int64_t *>(d_off), n_reads, nullptr, n_reads * read_len));
        CK(palace_eref_scan_refs(ctx, static_cast<const uint8_t *>(d_ref), static_cast<const int64_t *>(d_roff), 1, roff[1], 450, 425, static_cast<int32_t *>(d_r1)));
        palace_eref_probe_index *ix = nullptr;
        CK(palace_eref_probe_index_build(ctx, static_cast<const uint8_t *>(d_ref), static_cast<const int64_t *>(d_roff), 1, roff[1], &ix));
        CK(palace_eref_entry_buffers_attach(ctx, ix, nullptr, nullptr));                // the index's own blocks
        CK(palace_eref_attach_probe_index(ctx, ix));
        CK(palace_eref_set_option(ctx, "final_count", 1));
        CK(palace_eref_set_option(ctx, "probe_all_sets", 2));
        CK(palace_eref_table_reset(ctx));
        if (palace_eref_entry_counts_exchange(ctx, ix, comm, 0, 1, 0) != PALACE_ESTATE) {     // no count since the reset: refused before anything is sent
            std::fprintf(stderr, "exchange_selftest: an exchange without a count was not refused\n");
            return 1;
        }
        CK(palace_eref_set_count_mode(ctx, 0, 0));                                      // (auto: a small share must still take the fused path)
        CK(palace_eref_count_reads(ctx, static_cast<const uint8_t *>(d_bases), static_cast<const int64_t *>(d_off), n_reads, nullptr, n_reads * read_len));
        CK(palace_eref_entry_counts_exchange(ctx, ix, comm, 0, 1, 3 * n_reads * read_len));
        CK(palace_eref_scan_refs_indexed(ctx, ix, static_cast<const uint8_t *>(d_ref), static_cast<const int64_t *>(d_roff), 1, roff[1], 450, 425, static_cast<int32_t *>(d_r2)));
        int32_t r1[4], r2[4];
        CK(palace_d2h(ctx, r1, d_r1, 16)); CK(palace_d2h(ctx, r2, d_r2, 16));
        if (std::memcmp(r1, r2, 16) != 0 || r1[1] <= 0) {
            std::fprintf(stderr, "exchange_selftest: entry-count exchange: rows %d %d %d, plain scan %d %d %d\n", r2[0], r2[1], r2[2], r1[0], r1[1], r1[2]);
            return 1;
        }
        CK(palace_eref_set_option(ctx, "probe_all_sets", 0));
        CK(palace_eref_set_option(ctx, "final_count", 0));
        CK(palace_eref_attach_probe_index(ctx, nullptr));
        CK(palace_eref_probe_index_free(ctx, ix));
        CK(palace_eref_table_reset(ctx));
    }
    ncclCommDestroy(comm);
    palace_ctx_destroy(ctx);
    std::printf("ok: planes %llu >= %llu >= %llu unchanged by a one-rank exchange\n", (unsigned long long)before[0], (unsigned long long)before[1],
                (unsigned long long)before[2]);
    return 0;
}
