// libpalace_rccl.so: the count-table exchange of include/palace_rccl.h (RCCL over xGMI, one process per GPU).
// The arithmetic on the planes is libpalace_hip.so's (pack_low, merge_slices_packed); this file only moves bytes.
#include <rccl/rccl.h>

#include <algorithm>
#include <vector>

#include "../../include/palace_rccl.h"
#include "common.hpp"

using namespace palace;

#define PALACE_NCCL_TRY(expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) {                                                                          \
            palace::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, __LINE__); \
            return PALACE_EHIP;                                                                            \
        }                                                                                                  \
    } while (0)

extern "C" {

int palace_eref_table_exchange(palace_ctx *ctx, void *comm_, int rank, int world)
{
    PALACE_REQUIRE(ctx && comm_ && world >= 1 && rank >= 0 && rank < world, "bad argument");
    PALACE_REQUIRE(kPlaneBytes % (16 * static_cast<size_t>(world)) == 0, "world must divide the plane into 16-byte aligned slices");
    ncclComm_t comm = static_cast<ncclComm_t>(comm_);
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    void *planes[3];
    size_t plane_bytes = 0;
    int rc = palace_eref_table_planes(ctx, planes, &plane_bytes);
    if (rc) return rc;
    const size_t B = plane_bytes, S = B / static_cast<size_t>(world);
    rc = ensure_workspace(ctx, 3 * B);                       // low-bit plane to send + [2][world][S] received parts
    if (rc) return rc;
    char *low = static_cast<char *>(ctx->ws.ptr), *recv = low + B;
    rc = palace_eref_table_pack_low(ctx, low);
    if (rc) return rc;
    const char *send[2] = {low, static_cast<const char *>(planes[1])};      // (low bit, count >= 2): two bits per key
    // all-to-all of the slices, every peer at once: a plane already is [peer][slice]
    PALACE_NCCL_TRY(ncclGroupStart());
    for (int p = 0; p < 2; p++)
        for (int peer = 0; peer < world; peer++) {
            PALACE_NCCL_TRY(ncclSend(send[p] + static_cast<size_t>(peer) * S, S, ncclUint8, peer, comm, ctx->stream));
            PALACE_NCCL_TRY(ncclRecv(recv + (static_cast<size_t>(p) * world + peer) * S, S, ncclUint8, peer, comm, ctx->stream));
        }
    PALACE_NCCL_TRY(ncclGroupEnd());
    rc = palace_eref_table_merge_slices_packed(ctx, recv, world, static_cast<size_t>(rank) * S, S);
    if (rc) return rc;
    // the merged ">= 3" plane everywhere (in place: this rank's slice already sits where the gather puts it)
    char *p3 = static_cast<char *>(planes[2]);
    PALACE_NCCL_TRY(ncclAllGather(p3 + static_cast<size_t>(rank) * S, p3, S, ncclUint8, comm, ctx->stream));
    return PALACE_OK;
}

// ---- the key space split instead of the reads ----
int palace_eref_key_share(int rank, int world, uint32_t mask128[4])
{
    PALACE_REQUIRE(mask128 && world >= 1 && rank >= 0 && rank < world && 64 % world == 0, "world must divide 64");
    for (int i = 0; i < 4; i++) mask128[i] = 0;
    for (int base = 0; base < 128; base += 2 * world)
        for (int b : {base + rank, base + 2 * world - 1 - rank}) mask128[b >> 5] |= 1u << (b & 31);
    return PALACE_OK;
}

int palace_eref_key_share_gather(palace_ctx *ctx, void *comm_, int rank, int world)
{
    PALACE_REQUIRE(ctx && comm_ && world >= 1 && rank >= 0 && rank < world && 64 % world == 0, "world must divide 64");
    ncclComm_t comm = static_cast<ncclComm_t>(comm_);
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    void *planes[3];
    size_t plane_bytes = 0;
    int rc = palace_eref_table_planes(ctx, planes, &plane_bytes);
    if (rc) return rc;
    const size_t slice = plane_bytes / 128;                  // a level-1 bucket's 2^25 keys: 4 MiB of a plane
    char *p3 = static_cast<char *>(planes[2]);
    // every bucket's slice from its owner to everybody, in place, all of them in flight at once
    PALACE_NCCL_TRY(ncclGroupStart());
    for (int b = 0; b < 128; b++) {
        const int x = b % (2 * world), owner = x < world ? x : 2 * world - 1 - x;
        PALACE_NCCL_TRY(ncclBroadcast(p3 + b * slice, p3 + b * slice, slice, ncclUint8, owner, comm, ctx->stream));
    }
    PALACE_NCCL_TRY(ncclGroupEnd());
    return PALACE_OK;
}

// The same gather with the plane in sparse form (palace_eref_plane_pack / _unpack): counts + 16-bit keys instead of slices.
int palace_eref_key_share_gather_sparse(palace_ctx *ctx, void *comm_, int rank, int world, int64_t cap_keys, unsigned long long *h_max_keys)
{
    PALACE_REQUIRE(ctx && comm_ && world >= 1 && rank >= 0 && rank < world && 64 % world == 0 && cap_keys >= 0, "world must divide 64");
    ncclComm_t comm = static_cast<ncclComm_t>(comm_);
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_fine = 65536 / static_cast<size_t>(world);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t counts_b = up(n_fine * 4), keys_b = up(static_cast<size_t>(cap_keys) * 2 + 16), first_b = up((n_fine + 1) * 8);
    int rc = ensure_workspace(ctx, world * (counts_b + keys_b) + first_b);
    if (rc) return rc;
    char *ws = static_cast<char *>(ctx->ws.ptr);
    char *counts = ws, *keys = ws + world * counts_b, *first = keys + world * keys_b;
    uint32_t mine[4];
    rc = palace_eref_key_share(rank, world, mine);
    if (rc) return rc;
    // this rank's share packed straight into its slot of the gathered arrays, then both arrays gathered in place
    rc = palace_eref_plane_pack(ctx, mine, reinterpret_cast<uint32_t *>(counts + rank * counts_b), reinterpret_cast<uint16_t *>(keys + rank * keys_b),
                                cap_keys, reinterpret_cast<unsigned long long *>(first));
    if (rc) return rc;
    PALACE_NCCL_TRY(ncclGroupStart());
    PALACE_NCCL_TRY(ncclAllGather(counts + rank * counts_b, counts, counts_b, ncclUint8, comm, ctx->stream));
    PALACE_NCCL_TRY(ncclAllGather(keys + rank * keys_b, keys, keys_b, ncclUint8, comm, ctx->stream));
    PALACE_NCCL_TRY(ncclGroupEnd());
    for (int r = 0; r < world; r++) {
        if (r == rank) continue;
        uint32_t theirs[4];
        rc = palace_eref_key_share(r, world, theirs);
        if (rc) return rc;
        rc = palace_eref_plane_unpack(ctx, theirs, reinterpret_cast<const uint32_t *>(counts + r * counts_b), reinterpret_cast<const uint16_t *>(keys + r * keys_b),
                                      cap_keys, reinterpret_cast<unsigned long long *>(first));
        if (rc) return rc;
    }
    if (h_max_keys) {                                        // what the largest share needed: more than cap_keys means keys were cut off
        std::vector<uint32_t> h(world * n_fine);
        for (int r = 0; r < world; r++)
            PALACE_HIP_TRY(hipMemcpyAsync(h.data() + r * n_fine, counts + r * counts_b, n_fine * 4, hipMemcpyDeviceToHost, ctx->stream));
        PALACE_HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long mx = 0;
        for (int r = 0; r < world; r++) {
            unsigned long long t = 0;
            for (size_t k = 0; k < n_fine; k++) t += h[r * n_fine + k];
            mx = std::max(mx, t);
        }
        *h_max_keys = mx;
    }
    return PALACE_OK;
}

// ---- the reads sharded, partial counts of the DB's entries exchanged (no plane moves) ----
int palace_eref_entry_counts_exchange(palace_ctx *ctx, palace_eref_probe_index *ix, void *comm_, int rank, int world, int64_t keys_counted)
{
    PALACE_REQUIRE(ctx && ix && comm_ && world >= 1 && rank >= 0 && rank < world, "bad argument");
    ncclComm_t comm = static_cast<ncclComm_t>(comm_);
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    size_t cb = 0, hb = 0;
    int rc = palace_eref_entry_layout(ix, &cb, &hb);
    if (rc) return rc;
    PALACE_REQUIRE(cb % (512 * static_cast<size_t>(world)) == 0, "world must divide the count block into 512-byte multiples");
    void *cnt = nullptr, *hits = nullptr;
    rc = palace_eref_entry_buffers(ix, &cnt, &hits);
    if (rc) return rc;
    PALACE_REQUIRE(cnt && hits, "the index has no count block yet (palace_eref_entry_buffers_attach before the count)");
    if (!palace_eref_entry_counts_valid(ctx, ix)) {          // (before anything is sent: stale or foreign counts must not reach the peers)
        set_error("palace_eref_entry_counts_exchange: no count call (option probe_all_sets 2; n = 0 for a rank without reads) of this context "
                  "has written the index's count block since the last reset");
        return PALACE_ESTATE;
    }
    const size_t S = cb / static_cast<size_t>(world);
    rc = ensure_workspace(ctx, cb);                          // [world][S]: every peer's counts of this rank's share
    if (rc) return rc;
    char *recv = static_cast<char *>(ctx->ws.ptr);
    const char *send = static_cast<const char *>(cnt);
    PALACE_NCCL_TRY(ncclGroupStart());
    for (int peer = 0; peer < world; peer++) {
        PALACE_NCCL_TRY(ncclSend(send + static_cast<size_t>(peer) * S, S, ncclUint8, peer, comm, ctx->stream));
        PALACE_NCCL_TRY(ncclRecv(recv + static_cast<size_t>(peer) * S, S, ncclUint8, peer, comm, ctx->stream));
    }
    PALACE_NCCL_TRY(ncclGroupEnd());
    rc = palace_eref_entry_hits_from_counts(ctx, ix, recv, world, S, static_cast<size_t>(rank) * S, S);
    if (rc) return rc;
    char *h = static_cast<char *>(hits);
    PALACE_NCCL_TRY(ncclAllGather(h + static_cast<size_t>(rank) * (S / 2), h, S / 2, ncclUint8, comm, ctx->stream));
    return palace_eref_entry_hits_complete(ctx, ix, keys_counted);
}

int palace_eref_rows_allgather(palace_ctx *ctx, void *comm_, int rank, int world, int32_t *d_rows, int64_t n_refs,
                               const int64_t *ref_lo, const int64_t *ref_hi)
{
    PALACE_REQUIRE(ctx && comm_ && world >= 1 && rank >= 0 && rank < world && n_refs >= 0 && ref_lo && ref_hi, "bad argument");
    PALACE_REQUIRE(n_refs == 0 || d_rows, "null rows");
    ncclComm_t comm = static_cast<ncclComm_t>(comm_);
    PALACE_HIP_TRY(hipSetDevice(ctx->device));
    for (int r = 0; r < world; r++) PALACE_REQUIRE(0 <= ref_lo[r] && ref_lo[r] <= ref_hi[r] && ref_hi[r] <= n_refs, "ref range out of bounds");
    // ranges differ in length: a broadcast per owner, grouped (16 bytes per ref: tens of KB in all)
    PALACE_NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < world; r++) {
        const size_t n = static_cast<size_t>(ref_hi[r] - ref_lo[r]) * 4;
        if (n) PALACE_NCCL_TRY(ncclBroadcast(d_rows + 4 * ref_lo[r], d_rows + 4 * ref_lo[r], n, ncclInt32, r, comm, ctx->stream));
    }
    PALACE_NCCL_TRY(ncclGroupEnd());
    return PALACE_OK;
}

}  // extern "C"
