// Internal interface of the device-resident path / cycle decomposition (decomp.hip), shared by the two entry points that
// feed it: palace_match_decompose[_ex] (arcs ranked by the host, match.hip) and the resident stage-04 path (arcs built on the
// device from the filtered graph, filter.hip).  Not part of the C ABI.
#pragma once
#include <functional>

#include "common.hpp"

// the result object of the C ABI (palace_match_result_* accessors, match.hip); the resident stage-04 object (filter.hip) hands
// out one whose arrays it owns itself
struct palace_match_result {
    int64_t n = 0;
    int64_t *off = nullptr;
    int32_t *verts = nullptr, *iter = nullptr, *open_at = nullptr;
    uint8_t *kind = nullptr;
    uint64_t *bare = nullptr;            // compact results: bit s set = segment s has no arc (bare path of round 0 [+ the aggressive round])
    int64_t n_bare = 0;
    bool borrowed = false;               // the arrays (and this object) belong to somebody else: nothing to give back
    ~palace_match_result();              // match.hip (block pool)
};

namespace palace {

constexpr int kDecompGrid = 256;           // every kernel of the decomposition is grid-stride over a count it reads from device
constexpr int kDecompBlock = 256;          // memory: the launch sequence never depends on a number only the device knows
constexpr int kMaxRounds = 1024;           // iterations (+1 when aggressive) of a decomposition
constexpr int kMaxIters = 64;              // matching iterations enqueued per round at most

// scalars of one decomposition, device resident (the host reads them back once, at the end)
struct DecompState {
    int32_t S, V;                          // segments / oriented vertices of the arc-bearing sub-graph
    int64_t E;                             // arcs
    int64_t n_comp, n_vert;                // components / vertex slots written so far
    int64_t comp_cap, vert_cap;            // room in the output arrays
    uint32_t unsettled;                    // a round ran out of enqueued matching iterations before its fixed point
    uint32_t overflow;                     // output arrays too small
    uint32_t bad;                          // (unused)
    uint32_t dead;                         // no segment kept a copy in the last round: the kernels of further ordinary rounds return at once
    uint32_t packed;                       // the arcs' one-word keys (DecompBufs::kc) rank them: the second proposal pass has nothing to do
    uint64_t scan_total;                   // last scan: sum of all inputs
    int64_t alive_after;                   // vertices reported in the last round whose segment still has copies (0: later rounds are empty)
};

struct DecompBufs {
    DecompState *st = nullptr;
    // arcs [E]: tail, head (sub-graph vertex ids) and rank key -- lower (khi, klo) is better, keys are distinct
    int32_t *src = nullptr, *dst = nullptr;
    uint64_t *khi = nullptr, *klo = nullptr;
    uint8_t *done = nullptr;                            // the arc was seen closed in this round (it stays closed until the round ends)
    // Optional one-word form of the rank key (round 5): kc[e] < 2^52 orders the arcs of every slot exactly as (khi, klo) does,
    // so one 64-bit atomicMin per slot settles a proposal and the klo pass, its slot arrays and their resets fall away.  Whoever
    // builds the arcs fills kc and counts in *pack_bad the arcs that do not fit (filter.hip: weight | backed | class of (u, v)
    // in sub-graph ids); dec_init turns the mode on when none was counted and the decomposition enqueues fewer than 4096
    // iterations (12 bits of stamp above the key).  nullptr: keys are (khi, klo) or, with unique_hi, khi alone.
    uint64_t *kc = nullptr;
    const uint32_t *pack_bad = nullptr;
    // segments [S]: copies left, id of the segment in the caller's graph (vertices are reported as 2 * orig + orientation)
    int64_t *left = nullptr;
    int32_t *orig = nullptr;
    // vertices [V]
    int32_t *next = nullptr, *prev = nullptr, *on_path = nullptr, *open_at = nullptr;
    uint64_t *nhi = nullptr, *nlo = nullptr;            // key of the arc leaving the vertex
    uint64_t *bo_hi = nullptr, *bi_hi = nullptr;        // per out / in slot: its best proposal (stamped), or 0 = the slot is closed (decomp.hip)
    uint64_t *bo_lo = nullptr, *bi_lo = nullptr;        // [2][V], by iteration parity
    uint64_t *len_a = nullptr, *pos = nullptr;          // scan input / output: (1 << 40 | length) at emitting first vertices
    int64_t *pay = nullptr;
    uint8_t *alive = nullptr, *kind = nullptr;
    uint64_t *partials = nullptr;                       // [kDecompGrid + 1]
    uint32_t *changed = nullptr;                        // [rounds][iters]: a matching iteration took an arc
    // outputs (device)
    int64_t *o_off = nullptr;
    int32_t *o_verts = nullptr, *o_iter = nullptr, *o_open = nullptr;
    uint8_t *o_kind = nullptr;
};

// bytes of device memory carve() takes for a sub-graph of at most s_cap segments, e_cap arcs and the given output room
size_t decomp_bytes(int64_t s_cap, int64_t e_cap, int64_t comp_cap, int64_t vert_cap, int rounds, int iters);
// lays the arrays of `b` out in [base, base + decomp_bytes(...))
void decomp_carve(DecompBufs &b, char *base, int64_t s_cap, int64_t e_cap, int64_t comp_cap, int64_t vert_cap, int rounds, int iters);
// The decomposition is enqueued in pieces, none of which waits for the host: decomp_begin (state, slot arrays), then
// decomp_rounds for rounds [t0, t1) with `iters` matching iterations each (the ones behind a round's fixed point return at
// once; a round that was still taking arcs in its last enqueued iteration sets st->unsettled).  b.st->S/V/E, the arcs, left
// and orig must be in place (stream order).  `count` carries the iteration stamp from call to call.  When `unique_hi`, khi
// alone ranks the arcs (the second proposal pass is skipped).  After a group of rounds the caller reads the state back:
// alive_after == 0 means every later round is empty (all but an `aggressive` last one can be skipped).
constexpr int kFirstRoundIters = 7, kLaterRoundIters = 4, kRoundsPerGroup = 5, kFirstGroupRounds = 5;
int decomp_begin(palace_ctx *ctx, const DecompBufs &b, int rounds, int64_t comp_cap, int64_t vert_cap);
int decomp_rounds(palace_ctx *ctx, const DecompBufs &b, int t0, int t1, int rounds, int aggressive, int iters, bool unique_hi,
                  uint64_t *count);

// The usual way through: decomp_begin, decomp_group (the next kRoundsPerGroup rounds; only enqueues), then decomp_finish, which
// reads the state back after every group, enqueues further groups while segments keep copies, and redoes the decomposition
// with decomp_run_checked (after `reset_left` has put the copy numbers back) should a round not have settled.  h_state: pinned.
struct DecompRun { int next_round = 0; uint64_t count = 0; };
int decomp_group(palace_ctx *ctx, const DecompBufs &b, DecompRun &run, int rounds, int aggressive, bool unique_hi);
int decomp_finish(palace_ctx *ctx, const DecompBufs &b, DecompRun &run, int rounds, int aggressive, bool unique_hi, int64_t comp_cap,
                  int64_t vert_cap, int64_t max_iterations, DecompState *h_state, const std::function<int()> &reset_left);

// The same with the host checking for each round's fixed point (any number of iterations); synchronises the stream.
int decomp_run_checked(palace_ctx *ctx, const DecompBufs &b, int rounds, int aggressive, bool unique_hi, int64_t comp_cap,
                       int64_t vert_cap, int64_t max_iterations);

// exclusive scan of u64 values whose count *n_dev lives on the device: 3 launches on the stream; total -> *total_dev
int scan_u64(palace_ctx *ctx, const uint64_t *in, uint64_t *out, const int32_t *n_dev, uint64_t *partials, uint64_t *total_dev);

}  // namespace palace
