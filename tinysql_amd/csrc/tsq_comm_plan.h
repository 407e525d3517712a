// tsq_comm_plan.h — the BOOKKEEPING of the multi-GPU exchange (tsq_comm.hip), host only: no HIP, no RCCL.
//
// tsq_redistribute is an all-to-all(v) of columns: every rank splits its rows into `world` runs (run p = the rows rank p owns),
// the ranks exchange their run sizes, and every (column, peer) pair becomes one send and one receive.  What can go wrong there is
// arithmetic — which bytes of which buffer go to whom, where they land, how the offsets of a received var-len run are rebased —
// and none of it needs a GPU.  This header turns the gathered count matrix into the list of transfers and offset shifts one
// rank performs; tsq_comm.hip executes that list with ncclSend / ncclRecv / hipMemcpyAsync, tests/hostsim walks it for world sizes
// 2, 4 and 8 with memcpy standing in for the wire (test infrastructure only: the product has no other transport than RCCL).
//
// Replaces (reference): the bookkeeping of the worker dispatch / shuffle across workers — fetchOuterSideChunks handing chunks to
// join workers (executor/join.go:160-231), shuffleIntermData (executor/aggregate.go:352-356) — across GPUs.
#ifndef TSQ_COMM_PLAN_H
#define TSQ_COMM_PLAN_H

#include <cstdint>
#include <vector>

// The count vector of one rank, L = world + 1 + n_var * world words:
//   [0, world)                      rows it sends to every rank
//   [world]                         bit i set: its column i carries a null bitmap
//   [world + 1 + v * world + p]     data BYTES of var-len column v (the v-th var-len column) it sends to rank p
inline size_t tsq_comm_lwords(int world, int n_var) { return (size_t)world + 1 + (size_t)n_var * world; }

// The count exchange of SEVERAL pieces at once (tsq_redistribute_counts): every rank contributes the vectors of its pieces one
// after the other (Lsum words), ONE all-gather returns G[q * Lsum + ...] (rank q's contribution); piece k's world x L_k matrix is
// the k-th slice of every rank's contribution.
inline size_t tsq_comm_pack_counts(const std::vector<const std::vector<uint64_t>*>& vecs, uint64_t* out) {
    size_t o = 0;
    for (const std::vector<uint64_t>* v : vecs) {
        for (uint64_t w : *v) out[o++] = w;
    }
    return o;
}
inline std::vector<uint64_t> tsq_comm_unpack_counts(const uint64_t* G, int world, const std::vector<size_t>& Ls, size_t k) {
    size_t Lsum = 0, off = 0;
    for (size_t i = 0; i < Ls.size(); i++) {
        if (i < k) off += Ls[i];
        Lsum += Ls[i];
    }
    std::vector<uint64_t> M((size_t)world * Ls[k]);
    for (int q = 0; q < world; q++)
        for (size_t w = 0; w < Ls[k]; w++) M[(size_t)q * Ls[k] + w] = G[(size_t)q * Lsum + off + w];
    return M;
}

// ---- wire arithmetic of the two distributed join plans, per rank
// shared images (tsq_join_build_finish_shared): ONE all-reduce of the images per build side — a ring moves 2 (W - 1) / W of the
// buffer through every rank — and nothing per probe row.
inline uint64_t tsq_shared_plan_wire_bytes(int world, uint64_t image_bytes) { return world <= 1 ? 0 : 2 * image_bytes * (uint64_t)(world - 1) / (uint64_t)world; }
// hash-radix exchange (tsq_redistribute): every row leaves its rank unless it is owned by it — (W - 1) / W of the rows, row_bytes each, per step
inline uint64_t tsq_exchange_plan_wire_bytes(int world, uint64_t rows, uint64_t row_bytes) { return world <= 1 ? 0 : rows * row_bytes * (uint64_t)(world - 1) / (uint64_t)world; }

enum { TSQ_XFER_DATA = 0, TSQ_XFER_OFFS = 1, TSQ_XFER_NOTNULL = 2 };
struct tsq_comm_xfer {  // one piece: send_len bytes at send_off of the (column, kind) send buffer go to `peer`, recv_len bytes from `peer`
    int32_t col, kind, peer;  // land at recv_off of the (column, kind) receive buffer.  peer == own rank: a local copy (send_len == recv_len)
    uint64_t send_off, send_len, recv_off, recv_len;
};
struct tsq_comm_shift {  // received offsets of a var-len run -> the column's offsets: dst[dst_entry + k] = tmp[src_entry + k] + delta, k < rows
    int32_t col;
    uint64_t src_entry, dst_entry, rows;
    int64_t delta;
};
struct tsq_comm_plan {
    int64_t total_rows = 0;             // rows this rank receives
    uint64_t mask = 0;                  // columns that are nullable on ANY rank (they travel with one NOT-NULL byte per row)
    std::vector<int64_t> recv_bytes;    // per column: data bytes received (var-len columns; 0 otherwise)
    std::vector<int64_t> send_rows, recv_rows;  // per peer
    std::vector<tsq_comm_xfer> xfers;   // in issue order: sends and receives between two ranks pair up in this order
    std::vector<tsq_comm_shift> shifts;
};

// elem_size[i]: bytes of a fixed-width cell, 0 for a var-len column.  M: the gathered matrix, rank q's vector at M + q * L.
// Layout of the buffers the transfers refer to (per column):
//   DATA     send: the split's cells, run after run          recv: the runs of rank 0, 1, ... one after the other
//   OFFS     send: the split's offsets[nrows + 1] (8 B each)  recv: a scratch of (total rows + world) entries — the run from rank q lands
//            at entry (rows before it) + q and brings rows + 1 entries; the shifts rebase it into the column's offsets[total + 1]
//   NOTNULL  send / recv: one byte per row, run after run
// broadcast (key_mode 2 of tsq_redistribute = an all-gather of the columns): every rank sends ALL its rows to every rank, the send
// buffers hold the columns once (no runs): every send starts at offset 0 and a received var-len run's offsets count from 0.
inline tsq_comm_plan tsq_comm_make_plan(int rank, int world, int n_cols, const int32_t* elem_size, const uint64_t* M, bool broadcast = false) {
    tsq_comm_plan pl;
    int n_var = 0;
    std::vector<int> var_of((size_t)n_cols, -1);
    for (int i = 0; i < n_cols; i++)
        if (elem_size[i] == 0) var_of[(size_t)i] = n_var++;
    const size_t L = tsq_comm_lwords(world, n_var);
    pl.send_rows.resize((size_t)world);
    pl.recv_rows.resize((size_t)world);
    pl.recv_bytes.assign((size_t)n_cols, 0);
    for (int p = 0; p < world; p++) {
        pl.send_rows[(size_t)p] = (int64_t)M[(size_t)rank * L + (size_t)p];
        pl.recv_rows[(size_t)p] = (int64_t)M[(size_t)p * L + (size_t)rank];  // what rank p sends to this rank
        pl.total_rows += pl.recv_rows[(size_t)p];
        pl.mask |= M[(size_t)p * L + (size_t)world];
    }
    auto var_bytes = [&](int from, int v, int to) -> uint64_t { return M[(size_t)from * L + (size_t)world + 1 + (size_t)v * world + (size_t)to]; };
    for (int i = 0; i < n_cols; i++) {
        const int v = var_of[(size_t)i];
        const bool nn = (pl.mask >> i) & 1;
        uint64_t so = 0, ro = 0;  // rows before run p on the send / receive side
        uint64_t sb = 0, rb = 0;  // var-len: bytes before it
        for (int p = 0; p < world; p++) {
            const uint64_t sr = (uint64_t)pl.send_rows[(size_t)p], rr = (uint64_t)pl.recv_rows[(size_t)p];
            if (v >= 0) {
                const uint64_t sby = var_bytes(rank, v, p), rby = var_bytes(p, v, rank);
                // the run's slice of the offsets: rows + 1 entries (none for an empty run)
                pl.xfers.push_back({i, TSQ_XFER_OFFS, p, so * 8, sr ? (sr + 1) * 8 : 0, (ro + (uint64_t)p) * 8, rr ? (rr + 1) * 8 : 0});
                pl.xfers.push_back({i, TSQ_XFER_DATA, p, sb, sby, rb, rby});
                // on the wire the run's offsets count from the bytes rank p sent to the ranks before this one
                int64_t first = 0;
                for (int q = 0; q < rank && !broadcast; q++) first += (int64_t)var_bytes(p, v, q);
                if (rr) pl.shifts.push_back({i, ro + (uint64_t)p + 1, ro + 1, rr, (int64_t)rb - first});
                if (!broadcast) sb += sby;
                rb += rby;
            } else {
                const uint64_t es = (uint64_t)elem_size[i];
                pl.xfers.push_back({i, TSQ_XFER_DATA, p, so * es, sr * es, ro * es, rr * es});
            }
            if (nn) pl.xfers.push_back({i, TSQ_XFER_NOTNULL, p, so, sr, ro, rr});
            if (!broadcast) so += sr;
            ro += rr;
        }
        if (v >= 0) pl.recv_bytes[(size_t)i] = (int64_t)rb;
    }
    return pl;
}

#endif
