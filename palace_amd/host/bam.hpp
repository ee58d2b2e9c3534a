// BGZF inflate + BAM record decode into the structure-of-arrays columns the HIP classify kernel
// consumes (include/palace_hip.h: palace_bam_cols / palace_sa_item).  Stands in for what htslib
// does inside sam_open / sam_hdr_read / sam_read1 / bam_aux_get for the reference
// (generate_graph.cpp:611-698); written against the SAM/BAM specification, not against htslib.
#pragma once
#include <sys/mman.h>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/palace_hip.h"
#include "textio.hpp"

namespace palace_host {

// the inflated stream: mapped once, not zero-filled by this process (the inflate threads are the first to touch its pages), and
// offered to the kernel as huge pages: two gigabytes in 4 KiB pages are half a million page faults taken by the inflate threads
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, mapped = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { release(); }
    void release();
    void alloc(size_t bytes);
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

// A vector whose resize() leaves its (trivial) elements uninitialised.  The per-record columns are written exactly once, in full,
// by the decode threads; zero-filling 350 MB of them on one thread first was ~0.1 s of generateGraph at 6.7 M records.
// Large columns are RESERVED, not committed (mmap with MAP_NORESERVE, as RawBuf): they are sized for the most records the inflated
// stream can hold (bytes / 36) before the record walk has counted them -- 3.7 GB of address space for a 2 GB stream, of which only
// the pages of the records that exist are ever touched -- and that must not fail under vm.overcommit_memory = 2 or an address-space
// limit that the exact sizing of earlier versions fitted.
template <class T>
struct NoInit {
    using value_type = T;
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    static constexpr size_t kMapFrom = size_t{1} << 20;                   // bytes from which an allocation is a mapping of its own
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kMapFrom) return static_cast<T *>(::operator new(bytes));
        void *m = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        return static_cast<T *>(m);
    }
    void deallocate(T *p, size_t n) noexcept
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kMapFrom) ::operator delete(p);
        else ::munmap(p, bytes);
    }
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInit<U> &) const { return true; }
    template <class U> bool operator!=(const NoInit<U> &) const { return false; }
};
template <class T> using Column = std::vector<T, NoInit<T>>;

struct BamColumns {
    // header
    std::vector<std::string> target_name;
    std::vector<int32_t> target_len;
    Names tid_names;                               // target name -> tid, keys are views of `raw` (the inflated header)
    std::vector<int32_t> tid_of_name;              // ... by tid_names' dense id; the last duplicate wins (:624-627)
    int32_t tid_of(std::string_view name) const
    {
        const int k = tid_names.find(name);
        return k < 0 ? -1 : tid_of_name[static_cast<size_t>(k)];
    }
    int32_t tid_of_hashed(std::string_view name, uint64_t h) const       // h = hash_bytes(name)
    {
        const int k = tid_names.find_hashed(name, h);
        return k < 0 ? -1 : tid_of_name[static_cast<size_t>(k)];
    }
    // one entry per record, file order
    Column<int32_t> tid, pos, mtid, mpos, nm, ref_len, read_len, clip_s, clip_e, sa_off;
    Column<uint16_t> flag;
    Column<uint8_t> mapq;
    Column<uint64_t> qkey;
    std::vector<palace_sa_item> sa;
    // depth stage (palace:538-552): one entry per M / = / X CIGAR operation of the records `samtools depth` counts
    // (UNMAP, SECONDARY, QCFAIL, DUP clear): target, 0-based reference position, length.  Order is irrelevant.
    std::vector<int32_t> mseg_tid, mseg_pos, mseg_len;
    bool want_match_segments = true;               // set to false before loading when the depth stage is not run (generateGraph with a numeric <avgDepth>)
    // read names stay in the inflated stream; (offset, length) per record for the exactness guard
    RawBuf raw;
    Column<uint64_t> qname_at;
    Column<uint8_t> qname_len;
    int64_t n() const { return static_cast<int64_t>(flag.size()); }
    std::string qname(int64_t i) const
    {
        return std::string(reinterpret_cast<const char *>(raw.data()) + qname_at[i], qname_len[i]);
    }
};

// 64-bit key of a read name (seeded so a collision can be escaped by re-keying).
uint64_t name_key(const char *s, size_t n, uint64_t seed);

// The BGZF members of a file: raw DEFLATE data [in_off, in_off + in_len) of the file, ISIZE = out_len bytes at out_off of the
// inflated stream.
struct BgzfMember { uint64_t in_off, in_len, out_off, out_len; };

// A helper that inflates members somewhere else (a device: bam_device.hpp) while the loader's threads inflate from the front of
// the file: it is handed members from the BACK, runs on a thread the loader starts for it, and returns when claim() says no.
struct BackMembers {
    const uint8_t *file_data = nullptr;            // the mapped file
    size_t file_size = 0;
    uint8_t *out = nullptr;                        // the inflated stream
    virtual ~BackMembers() = default;
    // members [first, first + n), 0 < n <= max, taken off the back of what nobody has claimed yet; false: what is left is for
    // the loader's own threads (they would be done with it before a helper's batch came back)
    virtual bool claim(size_t max, size_t *first, size_t *n) = 0;
    virtual const BgzfMember &member(size_t i) const = 0;
    // member i of a claimed range is in place (decoded) or was refused by the helper (then it is inflated here and now, by the
    // loader's decoder with zlib behind it).  Every claimed member must be finished, whatever happens to the helper.
    virtual void finished(size_t i, bool decoded) = 0;
};
using MemberHelper = std::function<void(BackMembers &)>;

// Reads, inflates (threads) and decodes a whole BAM file.  Throws std::runtime_error.
void load_bam(const std::string &path, int threads, uint64_t key_seed, BamColumns &out);
// The same in two steps: begin() returns once the header (target names and lengths, the name index) is in `out`, with the
// inflate threads, the record walk (a thread of its own from the moment the header's end is known) and the decode of the records
// found so far (the inflate threads, as they run out of members) still going on; finish() waits for them and delivers the records.
// `out` must stay where it is in between; its per-record columns are sized for the most records the stream can hold until finish()
// cuts them to the records found.
struct BamLoad;
BamLoad *load_bam_begin(const std::string &path, int threads, BamColumns &out, const std::vector<MemberHelper> &helpers = {});
void load_bam_finish(BamLoad *load, uint64_t key_seed);
size_t load_bam_size_hint(const BamLoad *load);          // bytes of the inflated stream (known from the BGZF member trailers)

// The BGZF members of a mapped file (raw DEFLATE data and ISIZE of each; every field checked against the file) and the
// loader's own inflate of one of them -- for tools that inflate members themselves (bin/gpuinflate).  Throw std::runtime_error.
std::vector<BgzfMember> bgzf_members(const uint8_t *file, size_t size, size_t *total_out);
bool inflate_member(const uint8_t *file, size_t size, const BgzfMember &m, uint8_t *out);   // inflate_fast, then zlib; false: neither took it

// Re-key every read name with another seed (collision escape hatch).
void rekey(BamColumns &cols, uint64_t key_seed);

}  // namespace palace_host
