// Small text helpers shared by the host executables that read line-oriented files (filter_graph, matching): Python-like
// strip/split on string_views over mapped files, and a flat string -> dense id index.
#pragma once
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string_view>
#include <thread>
#include <vector>

namespace palace_host {

using sv = std::string_view;

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline sv rstrip(sv s) { while (!s.empty() && is_space(s.back())) s.remove_suffix(1); return s; }
inline sv strip(sv s) { s = rstrip(s); while (!s.empty() && is_space(s.front())) s.remove_prefix(1); return s; }

// str.split(sep): every separator counts, empty fields kept
inline void split_on(sv s, char sep, std::vector<sv> &out)
{
    out.clear();
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(sep, a);
        if (b == sv::npos) { out.push_back(s.substr(a)); return; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
}
// str.split(): runs of whitespace, no empty fields
inline void split_ws(sv s, std::vector<sv> &out)
{
    out.clear();
    size_t a = 0;
    while (a < s.size()) {
        while (a < s.size() && is_space(s[a])) a++;
        size_t b = a;
        while (b < s.size() && !is_space(s[b])) b++;
        if (b > a) out.push_back(s.substr(a, b - a));
        a = b;
    }
}

// calls f(line) for every line of [p, p + n), the line INCLUDING its '\n' when it has one (Python's iteration over a file)
template <class F>
inline void for_each_line(const char *p, size_t n, F f)
{
    size_t a = 0;
    while (a < n) {
        const void *e = std::memchr(p + a, '\n', n - a);
        const size_t b = e ? static_cast<size_t>(static_cast<const char *>(e) - p) + 1 : n;
        f(sv(p + a, b - a));
        a = b;
    }
}

// cuts [0, n) into about `parts` ranges that start at line starts
inline std::vector<size_t> line_cuts(const char *p, size_t n, size_t parts)
{
    std::vector<size_t> cut{0};
    for (size_t k = 1; k < parts; k++) {
        size_t at = n / parts * k;
        if (at <= cut.back()) continue;
        const void *e = std::memchr(p + at, '\n', n - at);
        if (!e) break;
        at = static_cast<size_t>(static_cast<const char *>(e) - p) + 1;
        if (at > cut.back() && at < n) cut.push_back(at);
    }
    cut.push_back(n);
    return cut;
}

// f(part, begin, end) for the parts of line_cuts(), one thread per part
template <class F>
inline void for_parts(const std::vector<size_t> &cut, F f)
{
    const size_t n = cut.size() - 1;
    if (n <= 1) { if (n) f(size_t{0}, cut[0], cut[1]); return; }
    std::vector<std::thread> pool;
    for (size_t k = 0; k < n; k++) pool.emplace_back([&, k] { f(k, cut[k], cut[k + 1]); });
    for (auto &t : pool) t.join();
}

inline uint64_t hash_bytes(sv s)                          // FNV-1a over 8-byte steps, finished with a multiply-shift mix
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ s.size();
    size_t i = 0;
    for (; i + 8 <= s.size(); i += 8) {
        uint64_t w;
        std::memcpy(&w, s.data() + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    uint64_t w = 0;
    if (i < s.size()) std::memcpy(&w, s.data() + i, s.size() - i);
    h = (h ^ w) * 0xbf58476d1ce4e5b9ull;
    return h ^ (h >> 32);
}

// string -> dense id, open addressing (a node-based map spends most of this program's time in malloc and cache misses)
struct Names {
    std::vector<uint64_t> slots;                   // (hash & ~mask_low32) | (id + 1); 0 = empty
    std::vector<sv> names;
    size_t mask = 0;
    void reserve(size_t n)
    {
        size_t cap = 1024;
        while (cap < 2 * n) cap <<= 1;
        if (cap <= slots.size()) return;
        slots.assign(cap, 0);
        mask = cap - 1;
        for (size_t id = 0; id < names.size(); id++) place(hash_bytes(names[id]), static_cast<int>(id));
    }
    void place(uint64_t h, int id)
    {
        size_t at = h & mask;
        while (slots[at]) at = (at + 1) & mask;
        slots[at] = (h & 0xffffffff00000000ull) | static_cast<uint32_t>(id + 1);
    }
    int find_hashed(sv s, uint64_t h) const
    {
        if (slots.empty()) return -1;
        for (size_t at = h & mask; slots[at]; at = (at + 1) & mask)
            if ((slots[at] ^ h) >> 32 == 0) {
                const int id = static_cast<int>(static_cast<uint32_t>(slots[at])) - 1;
                if (names[static_cast<size_t>(id)] == s) return id;
            }
        return -1;
    }
    // the slot a look-up of hash h starts at, requested ahead of time: a table of a million names is tens of megabytes, every
    // look-up a cache miss -- loops that know their next hashes ask for the slots a few rows early
    void prefetch(uint64_t h) const { if (!slots.empty()) __builtin_prefetch(&slots[h & mask]); }
    int find(sv s) const { return find_hashed(s, hash_bytes(s)); }
    int intern(sv s) { return intern_hashed(s, hash_bytes(s)); }
    int intern_hashed(sv s, uint64_t h)
    {
        if (2 * (names.size() + 1) > slots.size()) reserve(2 * names.size() + 512);
        const int got = find_hashed(s, h);
        if (got >= 0) return got;
        names.push_back(s);
        place(h, static_cast<int>(names.size()) - 1);
        return static_cast<int>(names.size()) - 1;
    }
};

// printf("%g") of a double into buf (at least 32 bytes; returns the length, no terminator), ~20 x faster than snprintf for the values a
// graph's SEG lines carry (generate_graph.cpp:1036 prints a million depths through ostream << double, which is %g).  %g = six
// significant digits, fixed notation while the decimal exponent X of the ROUNDED value is in [-4, 6), trailing zeros dropped.  The fast path
// takes 1e-4 <= v < 1e6 only: there v * 10^(5 - X) is ONE rounding away from the exact scaled value (the power is exact in double),
// so its nearest integer is the correctly rounded digit string unless the scaled value lies within 1e-6 of a tie -- those, and everything
// else (0, negatives, tiny, huge, inf, nan), go through snprintf.
inline size_t format_g6(double v, char *buf)
{
    static const double p10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    if (!(v >= 1e-4 && v < 1e6)) return static_cast<size_t>(std::snprintf(buf, 32, "%g", v));
    int X = v >= 1e5 ? 5 : v >= 1e4 ? 4 : v >= 1e3 ? 3 : v >= 1e2 ? 2 : v >= 1e1 ? 1 : v >= 1e0 ? 0 : v >= 1e-1 ? -1 : v >= 1e-2 ? -2 : v >= 1e-3 ? -3 : -4;
    double x = v * p10[5 - X];                               // 1e5 <= x < 1e6, give or take the inexact thresholds below 1
    if (x < 1e5) { X--; x = v * p10[5 - X]; }
    else if (x >= 1e6) { X++; x = v * p10[5 - X]; }
    if (X < -4 || X > 5 || !(x >= 1e5 && x < 1e6)) return static_cast<size_t>(std::snprintf(buf, 32, "%g", v));
    const double fl = std::floor(x), frac = x - fl;
    if (frac > 0.5 - 1e-6 && frac < 0.5 + 1e-6) return static_cast<size_t>(std::snprintf(buf, 32, "%g", v));
    long n = static_cast<long>(fl) + (frac > 0.5 ? 1 : 0);
    if (n == 1000000) { n = 100000; X++; if (X > 5) return static_cast<size_t>(std::snprintf(buf, 32, "%g", v)); }
    char d[6];
    for (int i = 5; i >= 0; i--) { d[i] = static_cast<char>('0' + n % 10); n /= 10; }
    int last = 5;
    while (last > 0 && d[last] == '0') last--;             // significant digits d[0 .. last]
    char *o = buf;
    if (X >= 0) {
        for (int i = 0; i <= X; i++) *o++ = d[i];            // (digits behind `last` up to the point are zeros of the value itself)
        if (last > X) { *o++ = '.'; for (int i = X + 1; i <= last; i++) *o++ = d[i]; }
    } else {
        *o++ = '0'; *o++ = '.';
        for (int i = -1; i > X; i--) *o++ = '0';
        for (int i = 0; i <= last; i++) *o++ = d[i];
    }
    return static_cast<size_t>(o - buf);
}

}  // namespace palace_host
