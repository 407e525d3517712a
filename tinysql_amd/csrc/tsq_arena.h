// tsq_arena.h — the free ranges of the context's arena (tsq_ctx_reserve), host only: no HIP.
//
// One slab, free ranges kept by offset and merged with their neighbours on release; a request takes the smallest free range that
// holds it (256-byte granules; blocks of 1 MiB or more on 2 MiB boundaries).  tests/hostsim drives it with random allocate / release sequences: no two live blocks overlap,
// everything released = one free range again.
#ifndef TSQ_ARENA_H
#define TSQ_ARENA_H

#include <cstddef>
#include <iterator>
#include <map>

struct tsq_arena_ranges {
    size_t size = 0, used = 0, peak = 0;
    std::map<size_t, size_t> free_;  // offset -> length
    void reset(size_t sz) {
        size = sz;
        used = peak = 0;
        free_.clear();
        if (sz) free_[0] = sz;
    }
    // Blocks of 1 MiB or more start on a 2 MiB boundary of the slab, as a hipMalloc of their own would (round 6: three 800 MB columns
    // carved back to back at 256-byte granules streamed 2.5 % slower than the same columns from hipMalloc — (a+b)*3-a over 1e8 rows
    // 0.464 against 0.453 ms); the bytes in front of the boundary stay a free range of their own.
    bool get(size_t bytes, size_t* off_out, size_t* got) {
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (need == 0 || need < bytes) return false;
        const size_t align = need >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 256;
        auto best = free_.end();
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            const size_t a = (it->first + align - 1) & ~(align - 1);
            if (a - it->first <= it->second && it->second - (a - it->first) >= need && (best == free_.end() || it->second < best->second)) best = it;
        }
        if (best == free_.end()) return false;
        const size_t off = best->first, len = best->second;
        const size_t a = (off + align - 1) & ~(align - 1), head = a - off;
        free_.erase(best);
        if (head) free_[off] = head;
        if (len - head > need) free_[a + need] = len - head - need;
        used += need;
        if (used > peak) peak = used;
        *off_out = a;
        *got = need;
        return true;
    }
    void put(size_t off, size_t len) {
        used -= len;
        auto next = free_.lower_bound(off);
        if (next != free_.end() && off + len == next->first) {  // merge with the range behind
            len += next->second;
            next = free_.erase(next);
        }
        if (next != free_.begin()) {  // ... and with the one in front
            auto prev = std::prev(next);
            if (prev->first + prev->second == off) {
                prev->second += len;
                return;
            }
        }
        free_[off] = len;
    }
};

#endif
