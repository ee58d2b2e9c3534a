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
