// tsq_boundary_bench.cpp — the drop-in boundary driven the way the cgo shim drives it: HOST chunks of
// tidb_max_chunk_size rows (sessionctx/variable/tidb_vars.go:242) pushed through tsq_join_build_push / tsq_join_probe_push, HOST chunks
// pulled through tsq_join_pull (Executor.Next, executor/executor.go:146-152) — from C, so that the per-call cost measured is the
// library's (lock, validation, copy into pinned staging) and not a Python interpreter's (ctypes: ~6 us per call, more than the call
// itself).  bench.py (tools/bench_sides.py: extra_pcie) calls tsq_boundary_join with the context it already holds.
// pull_every < 0: the pulls BORROW (TSQ_COL_BORROW on host columns: pointers into the operator's pinned result batch instead of a copy into
// the caller's chunk — what a shim that wraps the pinned memory as a chunk.Column for the duration of the parent's Next does); every
// borrowed cell is still read once (a sum per column) so that the number does not hide the parent's first touch of the rows.
// Build: g++ -shared (host/Makefile) -> tinysql_amd/host/libtsq_boundary.so; links libtsq.so from the package directory.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/tsq.h"

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void host_col(tsq_col& c, const int64_t* p, int64_t n) {
    memset(&c, 0, sizeof c);
    c.data = (void*)p;
    c.length = n;
    c.elem_size = 8;
    c.type = TSQ_I64;
}
}  // namespace

// 1 = a (k, v) x (k, v) inner join of n_build x n_probe rows in chunks of `chunk` rows; out[0..4] = build seconds, probe + pull seconds,
// of which pull seconds, joined rows, pull calls.  Pulls happen every `pull_every` pushed chunks (the parent's Next loop) and at the end.
extern "C" __attribute__((visibility("default"))) int32_t tsq_boundary_join(tsq_ctx* ctx, int64_t n_build, int64_t n_probe, int64_t chunk, int64_t pull_every,
                                                                            const int64_t* bk, const int64_t* bv, const int64_t* pk, const int64_t* pv,
                                                                            double* out) {
    const bool borrow = pull_every < 0;
    if (borrow) pull_every = -pull_every;
    uint64_t touched = 0;
    tsq_join_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.join_type = TSQ_JOIN_INNER;
    cfg.build_is_right = 1;
    cfg.n_keys = 1;
    cfg.n_build_cols = cfg.n_probe_cols = 2;
    for (int i = 0; i < 2; i++) cfg.build_types[i] = cfg.probe_types[i] = TSQ_I64;
    cfg.max_chunk_size = (int32_t)(chunk < (1 << 30) ? chunk : (1 << 30));
    cfg.concurrency = 5;
    tsq_join* j = nullptr;
    tsq_status s = tsq_join_create(ctx, &cfg, &j);
    if (s != TSQ_OK) return s;
    std::vector<int64_t> ob[4];
    std::vector<uint8_t> obm[4];
    tsq_col oc[4];
    for (int i = 0; i < 4; i++) {
        ob[i].resize((size_t)chunk);
        obm[i].resize((size_t)chunk / 8 + 16);
        host_col(oc[i], ob[i].data(), chunk);
        oc[i].null_bitmap = obm[i].data();
    }
    double t0 = now_s();
    tsq_col c[2];
    for (int64_t lo = 0; lo < n_build && s == TSQ_OK; lo += chunk) {
        const int64_t m = n_build - lo < chunk ? n_build - lo : chunk;
        host_col(c[0], bk + lo, m);
        host_col(c[1], bv + lo, m);
        s = tsq_join_build_push(j, c, 2, m);
    }
    if (s == TSQ_OK) s = tsq_join_build_finish(j);
    out[0] = now_s() - t0;
    double t_pull = 0;
    int64_t rows = 0, pulls = 0;
    auto drain = [&]() {
        while (s == TSQ_OK) {
            int64_t n = 0;
            int32_t eos = 0;
            const double t = now_s();
            if (borrow)
                for (int i = 0; i < 4; i++) oc[i].flags = TSQ_COL_BORROW;
            s = tsq_join_pull(j, oc, 4, chunk, &n, &eos);
            if (borrow && s == TSQ_OK)
                for (int i = 0; i < 4; i++) {
                    const uint64_t* p = (const uint64_t*)oc[i].data;
                    uint64_t a = 0;
                    for (int64_t r = 0; r < n; r++) a += p[r];
                    touched += a;
                }
            t_pull += now_s() - t;
            pulls++;
            if (n == 0) return;
            rows += n;
        }
    };
    t0 = now_s();
    int64_t pushed = 0;
    for (int64_t lo = 0; lo < n_probe && s == TSQ_OK; lo += chunk) {
        const int64_t m = n_probe - lo < chunk ? n_probe - lo : chunk;
        host_col(c[0], pk + lo, m);
        host_col(c[1], pv + lo, m);
        s = tsq_join_probe_push(j, c, 2, m, nullptr);
        if (++pushed % pull_every == 0) drain();
    }
    if (s == TSQ_OK) s = tsq_join_probe_finish(j);
    drain();
    out[1] = now_s() - t0;
    out[2] = t_pull;
    out[3] = (double)rows;
    out[4] = (double)pulls;
    memcpy(&out[5], &touched, 8);  // (the 64 bits of the wrapped sum, not a double)
    tsq_join_destroy(j);
    return s;
}
