// shared_sim.cpp — TEST-ONLY walk of the SHARED-IMAGES join plan (tsq_join_build_finish_shared, tinysql_amd/csrc/tsq_join.hip:
// da_prepare with a communicator) on the CPU, for any world size, with an element-wise sum standing in for ncclAllReduce.
//
// Everything the plan decides on the host runs through the SAME headers as the product: the merged key range -> tsq_da_plan
// (domain bits, byte or bit cells, partitions), the word of a key -> tsq_da_mix, what a summed image may look like ->
// tsq_da_shared_images_ok, the wire bytes -> tsq_shared_plan_wire_bytes.  Only the kernels are replaced by loops: a rank's images
// are assembled cell by cell (uint8 cells that WRAP like the device's bytes do under ncclSum, or one bit per cell), the ranks' images
// are summed (uint8 / uint32 arithmetic, i.e. with the wrap and the carries the device sum has), and every rank probes its OWN probe
// keys against the sum.  The result must equal the whole-table join count computed by a hash map.  Never loaded by the product.
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../tinysql_amd/csrc/tsq_device.h"
#include "../../tinysql_amd/csrc/tsq_dapack.h"
#include "../../tinysql_amd/csrc/tsq_comm_plan.h"

extern "C" {

// kind: 0 = keys uniform in [key_lo, key_lo + span) (duplicates inside and across ranks), 1 = a UNIQUE build side (a bijection of
// [0, total build rows) scaled by `stride` and shifted by key_lo, dealt to the ranks round robin), 2 = kind 1 plus ONE key that two ranks hold.
// hot: every rank also holds `hot` rows of the key key_lo + 3.  force: tsq_join_set_key_packing(FORCE) (no density test).
// out: [0] shared (0/1)  [1] count by the plan (-1 when not shared)  [2] count by the hash map  [3] bit cells  [4] image bytes
//      [5] domain bits  [6] wire bytes per rank of the build (ring all-reduce)  [7] wire bytes of the probe phase
int32_t sim_shared_join(int32_t world, const int64_t* build_rows, const int64_t* probe_rows, int32_t kind, int64_t key_lo, int64_t span, int64_t stride,
                        int32_t hot, int32_t force, uint64_t seed, int64_t* out, char* err, int32_t cap) {
    auto fail = [&](int code, const std::string& m) {
        if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m.c_str());
        return code;
    };
    if (world < 1 || world > 64) return fail(1, "world");
    std::vector<std::vector<int64_t>> bk((size_t)world), pk((size_t)world);
    int64_t total_b = 0;
    for (int r = 0; r < world; r++) total_b += build_rows[r];
    uint64_t s = seed;
    auto rnd = [&]() { return s = tsq_splitmix64(s); };
    if (kind == 0) {
        for (int r = 0; r < world; r++)
            for (int64_t i = 0; i < build_rows[r]; i++) bk[(size_t)r].push_back(key_lo + (int64_t)(rnd() % (uint64_t)span));
    } else {  // an affine bijection of [0, total_b): key i = ((a * i + c) mod total_b) * stride + key_lo
        const uint64_t a = 2654435761ull;
        int64_t at = 0;
        for (int r = 0; r < world; r++)
            for (int64_t i = 0; i < build_rows[r]; i++, at++) bk[(size_t)r].push_back(key_lo + (int64_t)(((a * (uint64_t)at + 12345) % (uint64_t)total_b)) * stride);
        if (kind == 2 && world > 1 && !bk[0].empty()) bk[(size_t)world - 1].push_back(bk[0][0]);
    }
    for (int r = 0; r < world; r++)
        for (int i = 0; i < hot; i++) bk[(size_t)r].push_back(key_lo + 3);
    const int64_t pspan = kind == 0 ? span * 2 : (total_b + total_b / 2) * stride;  // half / two thirds of the probe keys can match
    for (int r = 0; r < world; r++)
        for (int64_t i = 0; i < probe_rows[r]; i++) {
            int64_t k = key_lo - (kind == 0 ? span / 2 : 0) + (int64_t)(rnd() % (uint64_t)pspan);
            if (kind != 0) k = key_lo + ((k - key_lo) / stride) * stride;  // on the lattice of the build keys
            pk[(size_t)r].push_back(k);
        }
    // ---- the reference answer: one hash map over the whole build side
    std::unordered_map<int64_t, int64_t> mult;
    for (auto& v : bk)
        for (int64_t k : v) mult[k]++;
    int64_t want = 0;
    for (auto& v : pk)
        for (int64_t k : v) {
            auto it = mult.find(k);
            if (it != mult.end()) want += it->second;
        }
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[1] = -1;
    out[2] = want;
    // ---- step 1: the key range over all ranks (the product: k_da_minmax per rank, then two 8-byte all-reduces)
    const uint64_t flip = 0x8000000000000000ull;  // BIGINT = BIGINT: signed order as unsigned order of the images
    uint64_t lo = ~0ull, hi = 0, usable = 0;
    for (auto& v : bk) {
        uint64_t l = ~0ull, h = 0, n = 0;  // what a rank without rows contributes
        for (int64_t k : v) {
            const uint64_t x = (uint64_t)k ^ flip;
            l = x < l ? x : l;
            h = x > h ? x : h;
            n++;
        }
        // the all-reduce is done on int64 with the top bit flipped back (tsq_join.hip): min of lo, min of ~hi
        const int64_t ml = (int64_t)(l ^ flip), mh = ~(int64_t)(h ^ flip);
        const int64_t cl = (int64_t)(lo ^ flip), ch = ~(int64_t)(hi ^ flip);
        lo = (uint64_t)(ml < cl ? ml : cl) ^ flip;
        hi = (uint64_t)(~(mh < ch ? mh : ch)) ^ flip;
        usable += n;
    }
    if (usable == 0) return 0;
    // ---- step 2: the plan (host arithmetic of the product)
    const DaPlan pl = tsq_da_plan(lo ^ flip, hi ^ flip, usable, 0, true, force != 0);
    if (!pl.ok) return 0;
    const uint64_t img_bytes = tsq_da_image_bytes(pl);
    out[3] = pl.bit_cells;
    out[4] = (int64_t)img_bytes;
    out[5] = pl.dm.b;
    if (img_bytes > (1ull << 28)) return fail(2, "image too large for the simulation");
    // ---- step 3: every rank's images over the GLOBAL range; local verdicts (a byte cell beyond 255, a bit set twice)
    std::vector<std::vector<uint8_t>> img((size_t)world, std::vector<uint8_t>((size_t)img_bytes, 0));
    bool any_fail = false;
    for (int r = 0; r < world; r++) {
        uint8_t* im = img[(size_t)r].data();
        for (int64_t k : bk[(size_t)r]) {
            const uint64_t d = (uint64_t)k - pl.dm.kmin;
            if (d > pl.dm.range) return fail(3, "a build key outside the merged range");
            const uint32_t u = tsq_da_mix((uint32_t)d, pl.dm.s, pl.dm.mask);
            if (pl.bit_cells) {
                if (im[u >> 3] & (1u << (u & 7u))) any_fail = true;  // k_da_build_bits: flags[1]
                im[u >> 3] |= (uint8_t)(1u << (u & 7u));
            } else {
                if (im[u] == 255) any_fail = true;  // k_da_build_images: the bytes no longer add up to the rows (flags[0])
                im[u]++;
            }
        }
    }
    if (any_fail) return 0;  // (the flags all-reduce: every rank learns it)
    // ---- step 4: the sum across the ranks, with the arithmetic of ncclSum on uint8 / uint32 elements
    std::vector<uint8_t> sum((size_t)img_bytes, 0);
    if (pl.bit_cells) {
        for (size_t w = 0; w < (size_t)img_bytes / 4; w++) {
            uint32_t acc = 0;
            for (int r = 0; r < world; r++) {
                uint32_t x;
                memcpy(&x, img[(size_t)r].data() + 4 * w, 4);
                acc += x;
            }
            memcpy(sum.data() + 4 * w, &acc, 4);
        }
    } else {
        for (size_t i = 0; i < (size_t)img_bytes; i++) {
            uint8_t acc = 0;
            for (int r = 0; r < world; r++) acc = (uint8_t)(acc + img[(size_t)r][i]);
            sum[i] = acc;
        }
    }
    // ---- step 5: the population of the sum (k_da_image_check) against the usable rows of all ranks
    uint64_t pop = 0;
    for (size_t i = 0; i < (size_t)img_bytes; i++) pop += pl.bit_cells ? (uint64_t)__builtin_popcount(sum[i]) : sum[i];
    if (!tsq_da_shared_images_ok(pop, usable)) return 0;
    // ---- step 6: every rank probes its OWN rows (da_word: outside the range = no match), the counts are added up
    int64_t got = 0;
    for (int r = 0; r < world; r++)
        for (int64_t k : pk[(size_t)r]) {
            const uint64_t d = (uint64_t)k - pl.dm.kmin;
            if (d > pl.dm.range) continue;
            const uint32_t u = tsq_da_mix((uint32_t)d, pl.dm.s, pl.dm.mask);
            got += pl.bit_cells ? ((sum[u >> 3] >> (u & 7u)) & 1u) : sum[u];
        }
    out[0] = 1;
    out[1] = got;
    out[6] = (int64_t)tsq_shared_plan_wire_bytes(world, img_bytes);
    out[7] = 0;
    return 0;
}

// the arithmetic of DESIGN.md §6 for a scaling projection: per-rank wire bytes of both plans and the plan a build side takes, without
// touching a row.  out: [0] plan ok  [1] bit cells  [2] domain bits  [3] image bytes  [4] build wire bytes per rank (shared)
// [5] probe wire bytes per rank and step (shared: 0)  [6] probe wire bytes per rank and step of the exchange plan (8-byte keys)
int32_t sim_shared_projection(int32_t world, int64_t build_rows_per_rank, int64_t probe_rows_per_rank, int32_t unique, int64_t* out) {
    const uint64_t usable = (uint64_t)build_rows_per_rank * (uint64_t)world;
    const DaPlan pl = tsq_da_plan(0, usable - 1, usable, 0, true, false);
    for (int i = 0; i < 7; i++) out[i] = 0;
    out[6] = (int64_t)tsq_exchange_plan_wire_bytes(world, (uint64_t)probe_rows_per_rank, 8);
    if (!pl.ok || (pl.bit_cells && !unique)) return 0;
    out[0] = 1;
    out[1] = pl.bit_cells;
    out[2] = pl.dm.b;
    out[3] = (int64_t)tsq_da_image_bytes(pl);
    out[4] = (int64_t)tsq_shared_plan_wire_bytes(world, tsq_da_image_bytes(pl));
    out[5] = 0;
    return 0;
}

// ---- the same plan rank by rank, for a run with REAL processes (tests/dist_shared_worker.py: two processes over gloo).  A process
// calls sim_shared_rank_plan with the ALL-REDUCED range and row count (the product all-reduces the same three numbers), assembles its
// image with sim_shared_rank_image, the processes sum their images with an all-reduce (uint8 / int32 elements: the arithmetic of
// ncclSum), every process checks the population and probes its own rows.
// plan[]: [0] ok  [1] bit cells  [2] domain bits b  [3] mix shift s  [4] mask  [5] kmin (as int64 bits)  [6] range  [7] image bytes
int32_t sim_shared_rank_plan(int64_t kmin, int64_t kmax, int64_t usable, int32_t force, int64_t* plan) {
    const DaPlan pl = tsq_da_plan((uint64_t)kmin, (uint64_t)kmax, (uint64_t)usable, 0, true, force != 0);
    plan[0] = pl.ok ? 1 : 0;
    plan[1] = pl.bit_cells;
    plan[2] = pl.dm.b;
    plan[3] = pl.dm.s;
    plan[4] = pl.dm.mask;
    plan[5] = (int64_t)pl.dm.kmin;
    plan[6] = (int64_t)pl.dm.range;
    plan[7] = pl.ok ? (int64_t)tsq_da_image_bytes(pl) : 0;
    return 0;
}
// this rank's build keys into its (zeroed) image; returns 1 when a cell overflowed locally (byte beyond 255 / bit set twice)
int32_t sim_shared_rank_image(const int64_t* keys, int64_t n, const int64_t* plan, uint8_t* img) {
    int32_t fail = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t d = (uint64_t)keys[i] - (uint64_t)plan[5];
        if (d > (uint64_t)plan[6]) return -1;  // outside the merged range: the range all-reduce was wrong
        const uint32_t u = tsq_da_mix((uint32_t)d, (uint32_t)plan[3], (uint32_t)plan[4]);
        if (plan[1]) {
            if (img[u >> 3] & (1u << (u & 7u))) fail = 1;
            img[u >> 3] |= (uint8_t)(1u << (u & 7u));
        } else {
            if (img[u] == 255) fail = 1;
            img[u]++;
        }
    }
    return fail;
}
// the population of a (summed) image: k_da_image_check
int64_t sim_shared_population(const uint8_t* img, int64_t bytes, int32_t bit_cells) {
    uint64_t pop = 0;
    for (int64_t i = 0; i < bytes; i++) pop += bit_cells ? (uint64_t)__builtin_popcount(img[i]) : img[i];
    return (int64_t)pop;
}
int32_t sim_shared_images_ok(int64_t population, int64_t usable) { return tsq_da_shared_images_ok((uint64_t)population, (uint64_t)usable) ? 1 : 0; }
// this rank's probe keys against the summed image: joined rows
int64_t sim_shared_rank_probe(const int64_t* keys, int64_t n, const int64_t* plan, const uint8_t* sum) {
    int64_t got = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t d = (uint64_t)keys[i] - (uint64_t)plan[5];
        if (d > (uint64_t)plan[6]) continue;
        const uint32_t u = tsq_da_mix((uint32_t)d, (uint32_t)plan[3], (uint32_t)plan[4]);
        got += plan[1] ? ((sum[u >> 3] >> (u & 7u)) & 1u) : sum[u];
    }
    return got;
}

}  // extern "C"
