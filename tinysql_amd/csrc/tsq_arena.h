// tsq_arena.h — the free ranges of the context's arena (tsq_ctx_reserve), host only: no HIP.
//
// One slab, free ranges kept by offset and merged with their neighbours on release; a request takes the smallest free range that
// holds it (256-byte granules).  tests/hostsim drives it with random allocate / release sequences: no two live blocks overlap,
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
    bool get(size_t bytes, size_t* off_out, size_t* got) {
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (need == 0 || need < bytes) return false;
        auto best = free_.end();
        for (auto it = free_.begin(); it != free_.end(); ++it)
            if (it->second >= need && (best == free_.end() || it->second < best->second)) best = it;
        if (best == free_.end()) return false;
        const size_t off = best->first, len = best->second;
        free_.erase(best);
        if (len > need) free_[off + need] = len - need;
        used += need;
        if (used > peak) peak = used;
        *off_out = off;
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
