// comm_sim.cpp — TEST-ONLY walk of the multi-GPU exchange's bookkeeping (tinysql_amd/csrc/tsq_comm_plan.h) on the CPU.
//
// `world` ranks live in one process; memcpy stands in for ncclSend / ncclRecv (sends and receives between two ranks pair up in
// issue order, as RCCL matches them inside a group).  Everything tsq_redistribute computes on the host — the count vectors, the
// world x L matrix, which bytes of which buffer go to whom and where they land, the rebasing of received var-len offsets, the
// NOT-NULL bytes of nullable columns — runs exactly as in tsq_comm.hip (same header); only the split kernel is replaced by a
// stable partition by tsq_key_rank and the wire by memcpy.  Never loaded by the product.
#include <cstring>
#include <string>
#include <vector>

#include "../../tinysql_amd/csrc/tsq_device.h"
#include "../../tinysql_amd/csrc/tsq_comm_plan.h"
#include "../../tinysql_amd/csrc/tsq_arena.h"
#include "../../tinysql_amd/csrc/tsq_dapack.h"

namespace {

struct SimCol {
    int32_t es = 8;  // 0: var-len
    bool nullable = false;
    std::vector<uint8_t> data;    // fixed: rows * es; var: bytes
    std::vector<int64_t> offs;    // var: rows + 1
    std::vector<uint8_t> notnull; // rows (1 byte each)
};
struct SimRank {
    int64_t rows = 0;
    std::vector<SimCol> cols;
};

uint64_t rnd(uint64_t& s) { return s = tsq_splitmix64(s); }

SimCol take_rows(const SimCol& c, const std::vector<int64_t>& rows) {  // the split: rows gathered in the given order
    SimCol o;
    o.es = c.es;
    o.nullable = c.nullable;
    if (c.es == 0) o.offs.push_back(0);
    for (int64_t r : rows) {
        if (c.es) o.data.insert(o.data.end(), c.data.begin() + r * c.es, c.data.begin() + (r + 1) * c.es);
        else {
            o.data.insert(o.data.end(), c.data.begin() + c.offs[(size_t)r], c.data.begin() + c.offs[(size_t)r + 1]);
            o.offs.push_back((int64_t)o.data.size());
        }
        o.notnull.push_back(c.notnull[(size_t)r]);
    }
    return o;
}

}  // namespace

extern "C" {

// returns 0 when every rank received exactly the rows it owns, in source-rank order, with their cells, strings and NULL flags;
// otherwise a code (and a message in err, cap bytes).  col_kinds[i]: 8 / 4 = fixed width, 0 = var-len; nullable_mask: per
// (rank, column) bit r * n_cols + i (a column nullable on ONE rank must travel with NOT-NULL bytes on all of them).
int32_t sim_comm_exchange(int32_t world, int32_t n_cols, const int32_t* col_kinds, uint64_t nullable_mask, const int64_t* rows_per_rank, uint64_t seed,
                          int32_t skew, char* err, int32_t cap) {
    auto fail = [&](int code, const std::string& m) {
        if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m.c_str());
        return code;
    };
    int n_var = 0;
    std::vector<int> var_of((size_t)n_cols, -1);
    for (int i = 0; i < n_cols; i++)
        if (col_kinds[i] == 0) var_of[(size_t)i] = n_var++;
    const size_t L = tsq_comm_lwords(world, n_var);
    const bool broadcast = skew == 2;  // key_mode 2 of tsq_redistribute: an all-gather of the columns
    if (broadcast) skew = 0;
    // ---- the tables: column 0 is the BIGINT key (NULL keys go to rank 0, tsq_comm.hip)
    std::vector<SimRank> in((size_t)world);
    uint64_t s = seed;
    for (int r = 0; r < world; r++) {
        in[(size_t)r].rows = rows_per_rank[r];
        in[(size_t)r].cols.resize((size_t)n_cols);
        for (int i = 0; i < n_cols; i++) {
            SimCol& c = in[(size_t)r].cols[(size_t)i];
            c.es = col_kinds[i];
            c.nullable = (nullable_mask >> (r * n_cols + i)) & 1;
            if (c.es == 0) c.offs.push_back(0);
            for (int64_t k = 0; k < rows_per_rank[r]; k++) {
                const bool isnull = c.nullable && rnd(s) % 7 == 0;
                c.notnull.push_back(isnull ? 0 : 1);
                if (c.es == 0) {
                    const size_t len = isnull ? 0 : (size_t)(rnd(s) % 19);
                    for (size_t b = 0; b < len; b++) c.data.push_back((uint8_t)rnd(s));
                    c.offs.push_back((int64_t)c.data.size());
                } else {
                    uint64_t v = isnull ? 0 : rnd(s);
                    if (i == 0 && skew && rnd(s) % 3) v = 42;  // a hot key: one rank receives most rows
                    c.data.insert(c.data.end(), (uint8_t*)&v, (uint8_t*)&v + c.es);
                }
            }
        }
    }
    // ---- split (stable partition by destination), count vectors
    std::vector<std::vector<SimCol>> send((size_t)world);                          // [rank][col]: run after run
    std::vector<std::vector<std::vector<int64_t>>> dest_rows((size_t)world);       // [rank][dest] -> source rows in order
    std::vector<uint64_t> M((size_t)world * L, 0);
    for (int r = 0; r < world; r++) {
        dest_rows[(size_t)r].resize((size_t)world);
        const SimCol& key = in[(size_t)r].cols[0];
        for (int64_t k = 0; k < in[(size_t)r].rows; k++) {
            uint64_t kw = 0;
            memcpy(&kw, key.data.data() + k * key.es, (size_t)key.es);
            if (broadcast) {  // key_mode 2: every row goes to every rank
                for (int d = 0; d < world; d++) dest_rows[(size_t)r][(size_t)d].push_back(k);
                continue;
            }
            const uint32_t d = key.notnull[(size_t)k] ? tsq_key_rank(kw, (uint32_t)world) : 0u;
            if (d >= (uint32_t)world) return fail(10, "tsq_key_rank out of range");
            dest_rows[(size_t)r][d].push_back(k);
        }
        std::vector<int64_t> order;
        for (int p = 0; p < world; p++) {
            M[(size_t)r * L + (size_t)p] = (uint64_t)dest_rows[(size_t)r][(size_t)p].size();
            if (!broadcast || p == 0) order.insert(order.end(), dest_rows[(size_t)r][(size_t)p].begin(), dest_rows[(size_t)r][(size_t)p].end());  // broadcast: the columns once
        }
        uint64_t mask = 0;
        for (int i = 0; i < n_cols; i++) {
            send[(size_t)r].push_back(take_rows(in[(size_t)r].cols[(size_t)i], order));
            if (in[(size_t)r].cols[(size_t)i].nullable) mask |= 1ull << i;
            if (var_of[(size_t)i] >= 0) {  // byte boundaries of the runs: offsets[first row of run p]
                const SimCol& sc = send[(size_t)r][(size_t)i];
                size_t row = 0;
                for (int p = 0; p < world; p++) {
                    const size_t n = dest_rows[(size_t)r][(size_t)p].size();
                    M[(size_t)r * L + (size_t)world + 1 + (size_t)var_of[(size_t)i] * world + (size_t)p] = (uint64_t)(sc.offs[row + n] - sc.offs[row]);
                    if (!broadcast) row += n;
                }
            }
        }
        M[(size_t)r * L + (size_t)world] = mask;
    }
    // ---- every rank's plan and receive buffers
    std::vector<tsq_comm_plan> plan;
    struct Recv { std::vector<uint8_t> data, nn; std::vector<int64_t> tmp, offs; };
    std::vector<std::vector<Recv>> recv((size_t)world, std::vector<Recv>((size_t)n_cols));
    for (int r = 0; r < world; r++) {
        plan.push_back(tsq_comm_make_plan(r, world, n_cols, col_kinds, M.data(), broadcast));
        const tsq_comm_plan& pl = plan.back();
        for (int i = 0; i < n_cols; i++) {
            Recv& rc = recv[(size_t)r][(size_t)i];
            rc.data.assign(var_of[(size_t)i] >= 0 ? (size_t)pl.recv_bytes[(size_t)i] : (size_t)pl.total_rows * col_kinds[i], 0xEE);
            rc.nn.assign((size_t)pl.total_rows, 0xEE);
            rc.tmp.assign((size_t)pl.total_rows + world + 1, -7777);
            rc.offs.assign((size_t)pl.total_rows + 1, -7777);
        }
    }
    // ---- the wire: the k-th send of a to b meets the k-th receive of b from a (issue order inside the group)
    auto sbuf = [&](int r, const tsq_comm_xfer& x) -> const uint8_t* {
        const SimCol& c = send[(size_t)r][(size_t)x.col];
        return x.kind == TSQ_XFER_DATA ? c.data.data() : (x.kind == TSQ_XFER_OFFS ? (const uint8_t*)c.offs.data() : c.notnull.data());
    };
    auto slen = [&](int r, const tsq_comm_xfer& x) -> size_t {
        const SimCol& c = send[(size_t)r][(size_t)x.col];
        return x.kind == TSQ_XFER_DATA ? c.data.size() : (x.kind == TSQ_XFER_OFFS ? c.offs.size() * 8 : c.notnull.size());
    };
    auto rbuf = [&](int r, const tsq_comm_xfer& x, size_t* len) -> uint8_t* {
        Recv& rc = recv[(size_t)r][(size_t)x.col];
        if (x.kind == TSQ_XFER_DATA) { *len = rc.data.size(); return rc.data.data(); }
        if (x.kind == TSQ_XFER_OFFS) { *len = rc.tmp.size() * 8; return (uint8_t*)rc.tmp.data(); }
        *len = rc.nn.size();
        return rc.nn.data();
    };
    for (int a = 0; a < world; a++) {
        for (int b = 0; b < world; b++) {
            std::vector<const tsq_comm_xfer*> sends, recvs;
            for (const tsq_comm_xfer& x : plan[(size_t)a].xfers)
                if (x.peer == b && x.send_len) sends.push_back(&x);
            for (const tsq_comm_xfer& x : plan[(size_t)b].xfers)
                if (x.peer == a && x.recv_len) recvs.push_back(&x);
            if (a == b) {  // local copies: the same xfer carries both sides
                for (const tsq_comm_xfer* x : sends) {
                    if (x->send_len != x->recv_len) return fail(20, "own run: send_len != recv_len");
                    recvs.clear();
                }
                for (const tsq_comm_xfer& x : plan[(size_t)a].xfers)
                    if (x.peer == a && x.recv_len) recvs.push_back(&x);
            }
            if (sends.size() != recvs.size()) return fail(21, "rank " + std::to_string(a) + " -> " + std::to_string(b) + ": " + std::to_string(sends.size()) + " sends meet " + std::to_string(recvs.size()) + " receives");
            for (size_t k = 0; k < sends.size(); k++) {
                const tsq_comm_xfer &sx = *sends[k], &rx = *recvs[k];
                if (sx.send_len != rx.recv_len || sx.col != rx.col || sx.kind != rx.kind) return fail(22, "a send and its receive disagree (column / kind / length)");
                size_t rl = 0;
                uint8_t* rp = rbuf(b, rx, &rl);
                if (sx.send_off + sx.send_len > slen(a, sx)) return fail(23, "send beyond the send buffer");
                if (rx.recv_off + rx.recv_len > rl) return fail(24, "receive beyond the receive buffer");
                memcpy(rp + rx.recv_off, sbuf(a, sx) + sx.send_off, sx.send_len);
            }
        }
    }
    // ---- rebase the offsets, then compare with what every rank must hold: the runs of rank 0, 1, ... for it, in order
    for (int r = 0; r < world; r++) {
        const tsq_comm_plan& pl = plan[(size_t)r];
        for (int i = 0; i < n_cols; i++)
            if (var_of[(size_t)i] >= 0) recv[(size_t)r][(size_t)i].offs[0] = 0;
        for (const tsq_comm_shift& sh : pl.shifts) {
            Recv& rc = recv[(size_t)r][(size_t)sh.col];
            if (sh.src_entry + sh.rows > rc.tmp.size() || sh.dst_entry + sh.rows > rc.offs.size()) return fail(30, "offset shift out of range");
            for (uint64_t k = 0; k < sh.rows; k++) rc.offs[sh.dst_entry + k] = rc.tmp[sh.src_entry + k] + sh.delta;
        }
        uint64_t want_mask = 0;
        for (int q = 0; q < world; q++) want_mask |= M[(size_t)q * L + (size_t)world];
        if (pl.mask != want_mask) return fail(31, "nullable mask");
        int64_t total = 0;
        for (int q = 0; q < world; q++) total += (int64_t)dest_rows[(size_t)q][(size_t)r].size();
        if (pl.total_rows != total) return fail(32, "total rows");
        for (int i = 0; i < n_cols; i++) {
            SimCol want;
            want.es = col_kinds[i];
            if (want.es == 0) want.offs.push_back(0);
            for (int q = 0; q < world; q++) {
                const SimCol part = take_rows(in[(size_t)q].cols[(size_t)i], dest_rows[(size_t)q][(size_t)r]);
                want.data.insert(want.data.end(), part.data.begin(), part.data.end());
                for (size_t k = 1; k < part.offs.size() && want.es == 0; k++) want.offs.push_back(want.offs[0] + (int64_t)(want.data.size() - part.data.size()) + part.offs[k]);
                want.notnull.insert(want.notnull.end(), part.notnull.begin(), part.notnull.end());
            }
            const Recv& rc = recv[(size_t)r][(size_t)i];
            if (rc.data != want.data) return fail(40, "rank " + std::to_string(r) + " column " + std::to_string(i) + ": data bytes differ");
            if (want.es == 0 && rc.offs != want.offs) return fail(41, "rank " + std::to_string(r) + " column " + std::to_string(i) + ": offsets differ");
            if (((pl.mask >> i) & 1) && rc.nn != want.notnull) return fail(42, "rank " + std::to_string(r) + " column " + std::to_string(i) + ": NOT-NULL bytes differ");
        }
    }
    return 0;
}

// the plan of one rank through plain arrays (the world-size-2 gloo test executes it between two real processes).
// xfers_out: 7 words per transfer (col, kind, peer, send_off, send_len, recv_off, recv_len); shifts_out: 5 words per shift.
int32_t sim_comm_plan(int32_t rank, int32_t world, int32_t n_cols, const int32_t* elem_size, const uint64_t* M, int64_t* total_rows, uint64_t* mask,
                      int64_t* recv_bytes, int64_t* xfers_out, int32_t xfers_cap, int32_t* n_xfers, int64_t* shifts_out, int32_t shifts_cap, int32_t* n_shifts) {
    const tsq_comm_plan pl = tsq_comm_make_plan(rank, world, n_cols, elem_size, M);
    *total_rows = pl.total_rows;
    *mask = pl.mask;
    for (int i = 0; i < n_cols; i++) recv_bytes[i] = pl.recv_bytes[(size_t)i];
    *n_xfers = (int32_t)pl.xfers.size();
    *n_shifts = (int32_t)pl.shifts.size();
    if ((int32_t)pl.xfers.size() > xfers_cap || (int32_t)pl.shifts.size() > shifts_cap) return 1;
    for (size_t k = 0; k < pl.xfers.size(); k++) {
        const tsq_comm_xfer& x = pl.xfers[k];
        const int64_t w[7] = {x.col, x.kind, x.peer, (int64_t)x.send_off, (int64_t)x.send_len, (int64_t)x.recv_off, (int64_t)x.recv_len};
        memcpy(xfers_out + 7 * k, w, sizeof w);
    }
    for (size_t k = 0; k < pl.shifts.size(); k++) {
        const tsq_comm_shift& x = pl.shifts[k];
        const int64_t w[5] = {x.col, (int64_t)x.src_entry, (int64_t)x.dst_entry, (int64_t)x.rows, x.delta};
        memcpy(shifts_out + 5 * k, w, sizeof w);
    }
    return 0;
}

// the count exchange of several pieces at once: every rank packs its pieces' vectors, the all-gather concatenates the ranks'
// contributions, every piece's matrix must come back as if it had been exchanged alone.  Ls[k]: words of piece k's vector.
int32_t sim_comm_counts(int32_t world, int32_t n_pieces, const int32_t* Ls, uint64_t seed) {
    uint64_t s = seed;
    std::vector<size_t> L((size_t)n_pieces);
    size_t Lsum = 0;
    for (int k = 0; k < n_pieces; k++) Lsum += (L[(size_t)k] = (size_t)Ls[k]);
    std::vector<std::vector<std::vector<uint64_t>>> vec((size_t)world, std::vector<std::vector<uint64_t>>((size_t)n_pieces));  // [rank][piece]
    std::vector<uint64_t> G((size_t)world * Lsum);
    for (int r = 0; r < world; r++) {
        std::vector<const std::vector<uint64_t>*> mine;
        for (int k = 0; k < n_pieces; k++) {
            for (size_t w = 0; w < L[(size_t)k]; w++) vec[(size_t)r][(size_t)k].push_back(rnd(s));
            mine.push_back(&vec[(size_t)r][(size_t)k]);
        }
        if (tsq_comm_pack_counts(mine, G.data() + (size_t)r * Lsum) != Lsum) return 1;  // (the all-gather: rank r's words land at r * Lsum)
    }
    for (int k = 0; k < n_pieces; k++) {
        const std::vector<uint64_t> M = tsq_comm_unpack_counts(G.data(), world, L, (size_t)k);
        if (M.size() != (size_t)world * L[(size_t)k]) return 2;
        for (int r = 0; r < world; r++)
            for (size_t w = 0; w < L[(size_t)k]; w++)
                if (M[(size_t)r * L[(size_t)k] + w] != vec[(size_t)r][(size_t)k][w]) return 3;
    }
    return 0;
}

// the context arena's range bookkeeping (tsq_arena.h) under a random allocate / release sequence: live blocks never overlap, stay
// inside the slab, `used` is their sum, and once everything is released the slab is ONE free range again.  0 = all of that held.
int32_t sim_arena(uint64_t slab, int32_t steps, uint64_t seed) {
    tsq_arena_ranges a;
    a.reset((size_t)slab);
    uint64_t s = seed;
    std::vector<std::pair<size_t, size_t>> live;  // (offset, length)
    size_t sum = 0;
    for (int i = 0; i < steps; i++) {
        if (live.empty() || rnd(s) % 3) {
            const size_t want = (size_t)(rnd(s) % (slab / 8 + 1)) + 1;
            size_t off = 0, got = 0;
            if (a.get(want, &off, &got)) {
                if (got < want || (got & 255) || off + got > slab) return 1;
                for (const auto& b : live)
                    if (off < b.first + b.second && b.first < off + got) return 2;  // overlap
                live.emplace_back(off, got);
                sum += got;
            }
        } else {
            const size_t k = (size_t)(rnd(s) % live.size());
            a.put(live[k].first, live[k].second);
            sum -= live[k].second;
            live.erase(live.begin() + (long)k);
        }
        if (a.used != sum || a.peak < a.used) return 3;
        size_t free_sum = 0, prev_end = (size_t)-1;
        for (const auto& f : a.free_) {
            if (prev_end != (size_t)-1 && f.first <= prev_end) return 4;  // two free ranges touch or overlap: not merged
            prev_end = f.first + f.second;
            free_sum += f.second;
        }
        if (free_sum + sum != slab) return 5;
    }
    for (const auto& b : live) a.put(b.first, b.second);
    if (a.used != 0 || a.free_.size() != 1 || a.free_.begin()->first != 0 || a.free_.begin()->second != slab) return 6;
    return 0;
}

// the arithmetic of packed keys (tsq_dapack.h).  b <= exhaustive_bits: every d in [0, 2^b) — mix stays inside the range, unmix
// inverts it (so mix is a bijection); larger b: `samples` random d.  0 = held for every b in [13, 31].
int32_t sim_da_mix(int32_t exhaustive_bits, int64_t samples, uint64_t seed) {
    uint64_t s = seed;
    for (uint32_t b = 13; b <= 31; b++) {
        const uint32_t mask = (uint32_t)((1ull << b) - 1), sh = (b + 1) / 2;
        if ((int32_t)b <= exhaustive_bits) {
            for (uint32_t d = 0; d <= mask; d++) {
                const uint32_t u = tsq_da_mix(d, sh, mask);
                if (u > mask) return (int32_t)b;
                if (tsq_da_unmix(u, sh, mask) != d) return 100 + (int32_t)b;
            }
        } else {
            for (int64_t i = 0; i < samples; i++) {
                const uint32_t d = (uint32_t)rnd(s) & mask;
                const uint32_t u = tsq_da_mix(d, sh, mask);
                if (u > mask) return (int32_t)b;
                if (tsq_da_unmix(u, sh, mask) != d) return 100 + (int32_t)b;
                if (tsq_da_mix(tsq_da_unmix(d, sh, mask), sh, mask) != d) return 200 + (int32_t)b;  // (and the other way round: onto)
            }
        }
    }
    return 0;
}
// composite keys: n_keys columns with random fields; rows drawn inside and slightly outside the fields.  Two rows must get the same
// composite exactly when every cell is inside its field and all cells are equal; a row with a cell outside gets ~0.
int32_t sim_da_compose(int32_t n_keys, int32_t rows, uint64_t seed) {
    uint64_t s = seed;
    DaFields f;
    memset(&f, 0, sizeof f);
    f.n = n_keys;
    uint32_t total = 0;
    for (int k = 0; k < n_keys; k++) {
        const uint32_t w = 1 + (uint32_t)(rnd(s) % 6);
        f.kmin[k] = rnd(s) % 3 == 0 ? (uint64_t)(-(int64_t)(rnd(s) % 1000)) : rnd(s) % 100000;  // (a negative BIGINT minimum wraps: exact)
        f.maxd[k] = (1ull << w) - 1 - rnd(s) % 2;
        f.shift[k] = total;
        f.skip_high[k] = 0;
        total += w;
    }
    std::vector<std::vector<uint64_t>> cells((size_t)rows, std::vector<uint64_t>((size_t)n_keys));
    std::vector<uint64_t> comp((size_t)rows);
    std::vector<char> inside((size_t)rows);
    for (int r = 0; r < rows; r++) {
        bool in = true;
        for (int k = 0; k < n_keys; k++) {
            const int64_t off = (int64_t)(rnd(s) % (f.maxd[k] + 4)) - 1;  // -1 .. maxd + 2
            cells[(size_t)r][(size_t)k] = f.kmin[k] + (uint64_t)off;
            in = in && off >= 0 && (uint64_t)off <= f.maxd[k];
        }
        inside[(size_t)r] = in;
        comp[(size_t)r] = tsq_da_compose_cells(f, cells[(size_t)r].data());
        if (in == (comp[(size_t)r] == ~0ull)) return 1;
        if (in && (comp[(size_t)r] >> total)) return 2;  // the composite fits the fields' bits
    }
    for (int a = 0; a < rows; a++)
        for (int b = a + 1; b < rows; b++) {
            if (!inside[(size_t)a] || !inside[(size_t)b]) continue;
            const bool same_cells = cells[(size_t)a] == cells[(size_t)b];
            if (same_cells != (comp[(size_t)a] == comp[(size_t)b])) return 3;
        }
    return 0;
}

}  // extern "C"
