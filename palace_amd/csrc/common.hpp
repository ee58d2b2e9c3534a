// Shared internals of libpalace_hip.so (gfx950 only; no other back end exists or is planned).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/palace_hip.h"

namespace palace {

void set_error(const char *fmt, ...);

#define PALACE_HIP_TRY(expr)                                                                    \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            palace::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                              __LINE__);                                                        \
            return PALACE_EHIP;                                                                 \
        }                                                                                       \
    } while (0)

#define PALACE_REQUIRE(cond, msg)                                  \
    do {                                                           \
        if (!(cond)) {                                             \
            palace::set_error("%s: %s", __func__, msg);            \
            return PALACE_EINVAL;                                  \
        }                                                          \
    } while (0)

constexpr int kWave = 64;               // CDNA wavefront
constexpr int kCUs = 256;               // MI355X
constexpr size_t kPlaneBytes = 1ull << 29;   // 2^32 bits
constexpr size_t kPlaneWords = 1ull << 27;   // u32 words per plane

// E1: the nine 32-bit masks of the position-wise coder.  mask[i][q] has bit t set iff channel i
// reads projection q at k-mer offset 31-t (see eref.hip for the derivation).
struct CoderMasks {
    uint32_t m[3][3];
};

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace palace

namespace palace {
struct MatchScratch;                      // host temporaries of palace_match_decompose (match.hip)
void free_match_scratch(MatchScratch *m);
}  // namespace palace

struct palace_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;        // false: the caller's stream (palace_ctx_create_on_stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t key_buckets[4] = {~0u, ~0u, ~0u, ~0u};   // palace_eref_set_key_buckets: the level-1 buckets (key >> 25) count calls take in
    std::vector<hipEvent_t> marks;   // lazily created
    // eref
    bool coder_set = false;
    palace::CoderMasks masks{};
    uint32_t *plane[3] = {nullptr, nullptr, nullptr};
    bool planes_external = false;
    bool want_final = false;        // option final_count: a count into a clean table may keep only the ">= 3" plane
    bool final_only = false;        // ... and did: planes ">= 1" and ">= 2" are all zero, the table cannot take further counts
    bool table_clean = false;       // every plane bit is zero (set by reset, cleared by whatever writes the planes)
    int64_t keys_counted = 0;       // key instances counted into the table since the last reset (an upper bound; -1: unknown -- planes merged or written from outside)
    // Phase B's channel-0 probe fused into the count launch (palace_eref_attach_probe_index): the attached per-DB index, and the
    // index whose hit bytes the last count launch left complete (cleared by whatever changes the planes afterwards)
    const struct palace_eref_probe_index *probe_ix = nullptr;
    const struct palace_eref_probe_index *c0_hits_ix = nullptr;
    uint32_t hits_mask = 0;         // ... which of its entry sets' hit bits that launch left (bit 0: channel 0; 0xf: all four, option probe_all_sets)
    int probe_all_sets = 0;         // option: a final count with an attached index tests every entry set and writes no plane (2: leaves partial counts)
    bool sent_scattered = false;    // ... and that launch carried the sentinels' hits to position order itself
    int64_t scan_ref_lo = 0, scan_ref_hi = 0;   // options scan_ref_lo / _hi: the indexed scan works on refs [lo, hi) (hi = 0: all)
    const void *counts_ptr = nullptr;   // probe_all_sets 2: the count block (an index's, or the caller's) that holds this context's partial counts of
                                    // the sample counted since the last reset -- a fused count wrote them, or a call without reads zeroed them
    bool planeless = false;         // ... and did: the table holds NOTHING (all three planes are zero, as after a reset); Phase B is the attached
                                    // index's hit bits alone, and whatever else reads the table is refused until the next reset
    int count_mode = 0;             // 0 auto, 1 direct atomics, 2 binned
    int64_t bin_cap_override = 0;
    int64_t slab_override = 0;
    palace::Workspace ws;      // grow-only scratch
    bool ws_grown = false;
    palace::Workspace pin;     // grow-only pinned host staging
    palace::MatchScratch *match_scratch = nullptr;
    int64_t graph_border = -1;      // candidates of the last classify call that the host's libm has to score, and whose they are
    const void *graph_border_cands = nullptr;
    int64_t graph_border_n = -1;    // ... and how many candidates that call left there (a buffer that was appended to is not that call's)
    int match_grid = 0;             // workgroups of the decomposition's arc- and vertex-sized phases (0 = default; decomp.hip)
    bool match_two_word_keys = false; // option: the decomposition never uses the one-word form of the arc keys (A/B runs, tests)
    int match_iters = 0;            // matching iterations enqueued per round (0 = defaults; tests lower it to force the checked path)
    uint64_t *d_small = nullptr;   // 64 x u64 scratch for reductions
};

namespace palace {
int ensure_workspace(palace_ctx *ctx, size_t bytes);
int ensure_pinned(palace_ctx *ctx, size_t bytes);
int ensure_table(palace_ctx *ctx);
}  // namespace palace
