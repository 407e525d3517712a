/*
 * tsq.h — C-ABI of libtsq: the MI355X (gfx950) hot path for TinySQL's chunked operators.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++/torch types.
 * A cgo shim inside TinySQL's package `executor` binds exactly these symbols
 * (see INTEGRATION.md).  Every entry point cites the reference interface it replaces
 * (paths relative to the TinySQL tree).
 *
 * Conventions
 *   - every function returns a tsq_status (int32_t); 0 == TSQ_OK.  Nothing aborts/exits.
 *   - tsq_last_error(handle) returns a human readable message for the last failure on
 *     that handle (or the last creation failure when handle == NULL).
 *   - handles are NOT thread-affine (executor/join.go:207, aggregate.go:512 call Next from
 *     background goroutines): every entry point sets its own device and uses the handle's
 *     own HIP stream; one caller at a time per handle, except tsq_*_cancel which may be
 *     called from any thread at any time.
 *   - host pointers passed in are only read/written for the duration of the call
 *     (cgo pointer rule); nothing is retained.
 *   - a column (tsq_col) mirrors util/chunk/column.go:28-34: fixed-width little-endian
 *     data, null bitmap with bit==1 meaning NOT NULL, LSB first (column.go:89-92);
 *     null_bitmap == NULL means "no NULLs".
 */
#ifndef TSQ_H
#define TSQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSQ_ABI_VERSION 7

/* ---------------------------------------------------------------- status codes */
typedef int32_t tsq_status;
enum {
    TSQ_OK = 0,
    TSQ_ERR_INVALID = 1,          /* bad argument / bad state (maps to a Go errors.New)          */
    TSQ_ERR_UNSUPPORTED = 2,      /* plan not eligible: caller must fall back to the Go operator */
    TSQ_ERR_OOM_DEVICE = 3,
    TSQ_ERR_HIP = 4,
    TSQ_ERR_OVERFLOW_BIGINT = 5,  /* types.ErrOverflow "BIGINT"           (types/overflow.go:33-40) */
    TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED = 6, /* types.ErrOverflow "BIGINT UNSIGNED"                  */
    TSQ_ERR_OVERFLOW_DOUBLE = 7,  /* types.ErrOverflow "DOUBLE" (builtin_arithmetic_vec.go:51)     */
    TSQ_ERR_CANCELLED = 8,        /* tsq_*_cancel was called (executor.go:158 kill flag)           */
    TSQ_ERR_NO_DEVICE = 9,        /* no HIP device visible: the product path never falls back     */
    TSQ_ERR_DIV_BY_ZERO = 10      /* only in strict INSERT/DELETE mode (expression/errors.go:65-77)*/
};

/* ---------------------------------------------------------------- column ABI */
/* element types (types/eval_type.go:21-28 has only Int/Real/String eval types) */
enum {
    TSQ_I64 = 0,   /* bigint and narrower ints, stored as 8 bytes (util/chunk/codec.go:171-181) */
    TSQ_U64 = 1,   /* same 8 bytes, UNSIGNED flag set in FieldType.Flag                         */
    TSQ_F32 = 2,   /* float, 4 bytes                                                            */
    TSQ_F64 = 3,   /* double, 8 bytes                                                           */
    TSQ_BYTES = 4  /* var-len (offsets[n+1] + data), binary collation: expression inputs (string builtins), join payload
                      and join keys, group keys and FIRST_ROW / MAX / MIN / COUNT arguments of the aggregate */
};

/* tsq_col.flags */
#define TSQ_COL_DEVICE 1u /* data/null_bitmap/offsets are device (HBM) pointers */
#define TSQ_COL_RETAIN 4u /* ABI 7, an INPUT column of tsq_join_build_push: the operator may KEEP this device buffer as its build-side storage
                           * instead of copying the rows — what the reference does with the chunks it is handed (hash_table.go:146-169 PutChunk ->
                           * chunk.List.Add keeps the chunk, util/chunk/list.go:96-110).  The caller leaves the buffer (and its null bitmap) unchanged and
                           * allocated until tsq_join_destroy.  Honoured for the FIRST push of a build side when every column is a fixed-width
                           * TSQ_COL_DEVICE | TSQ_COL_RETAIN column whose buffers are whole tsq_dev_alloc blocks; otherwise the rows are copied as always
                           * (later pushes copy the retained rows into the operator's own storage first). */
#define TSQ_COL_BORROW 2u /* an OUTPUT column of a pull (tsq_join_pull; device-resident pulls, and since round 6 host pulls too — the pointers then
                             lead into the operator's PINNED result batch, which the D2H copies filled): instead of copying into the caller's buffers the
                             operator hands out pointers into its own result batch — data / null_bitmap are SET by the call (null_bitmap =
                             NULL when the column holds no NULL) and stay valid until the next pull, peek, finish or destroy on the handle (tsq_join_peek
                             releases fully consumed result batches too).
                             The device-chunk hand-off between GPU operators (Chunk.SwapColumns, util/chunk/chunk.go:231-235, is the
                             same idea: ownership of the buffers moves, no row is copied).  Fixed-width columns only. */

/* mirrors util/chunk/column.go:28-34 */
typedef struct tsq_col {
    void*    data;         /* fixed: length*elem_size bytes, native little endian                */
    uint8_t* null_bitmap;  /* (length+7)/8 bytes, bit=1 => NOT NULL; NULL => column has no NULL */
    int64_t* offsets;      /* var-len only (length+1 entries, first 0); NULL for fixed           */
    int64_t  length;       /* rows                                                               */
    int32_t  elem_size;    /* 4, 8, or -1 for var-len                                            */
    int32_t  type;         /* TSQ_I64 ...                                                        */
    uint32_t flags;        /* TSQ_COL_DEVICE                                                     */
    uint32_t reserved;
} tsq_col;

/* ---------------------------------------------------------------- misc / context */
typedef struct tsq_ctx tsq_ctx;

int32_t     tsq_abi_version(void);
int32_t     tsq_device_count(void);                 /* 0 when no GPU is visible */
const char* tsq_last_error(const void* handle);     /* handle may be ctx/join/agg/expr or NULL */

/* One context per (process, device).  All operators created from it share its stream. */
tsq_status tsq_ctx_create(int32_t device, tsq_ctx** out);
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the ctx's own. */
tsq_status tsq_ctx_set_stream(tsq_ctx* ctx, void* hip_stream);
tsq_status tsq_ctx_sync(tsq_ctx* ctx);
void       tsq_ctx_destroy(tsq_ctx* ctx);
/* Reserve `bytes` of HBM as the context's ARENA: the buffers of every operator created from the context (hash tables, partition
 * stores, result batches, tsq_dev_alloc blocks) are carved from it, so the first query of a session does not pay hipMalloc's
 * first touch (~35 ms per GB: 310 ms for a 1e8-row build side, 2.7 ms once the memory is there).  Call it when the host process
 * creates the context (the reference recycles its chunks through outerChkResourceCh / joinChkResourceCh instead of allocating
 * per batch, executor/join.go:54-56,169-171, and a Go process keeps its heap across queries); a request the arena cannot hold
 * falls through to hipMalloc.  bytes = 0 gives the arena back; not allowed
 * while operators hold buffers of it.  tsq_ctx_arena_stats: slab size, bytes in use, high-water mark. */
tsq_status tsq_ctx_reserve(tsq_ctx* ctx, int64_t bytes);
/* TEST AND MEASUREMENT KNOBS.  A host never calls this: every knob defaults to what the library was measured and tested
 * with.  The parity tests use them to force the rarely taken variants (a heap compaction after every batch, truncated group
 * tags, the un-pipelined row decoder ...), tools/ and bench.py to A/B kernel variants on one box.  They replace the
 * environment variables earlier rounds read inside the product code.  value = TSQ_KNOB_DEFAULT restores the default.
 * A knob is read when the operator takes the decision it steers (usually at its first batch). */
#define TSQ_KNOB_DEFAULT INT64_MIN
enum {
    TSQ_KNOB_PACKED_KEYS = 0,        /* 0: no packed-key routes (join and aggregate) */
    TSQ_KNOB_DA_MIN_BUILD_ROWS = 1,  /* build rows from which AUTO tries the packed routes (default 1 Mi) */
    TSQ_KNOB_DA_PBITS = 2,           /* log2(partitions) of the packed routes (default: min(11, b - 10)) */
    TSQ_KNOB_PACKED_EMIT_PAIRS = 3,  /* 1: AUTO may take the pairs variant of the materialising packed route (K4d) */
    TSQ_KNOB_RADIX_KERNEL_L2 = 4,    /* 1: the 64-bit radix probe keeps round 1's L2 route */
    TSQ_KNOB_LDS_NF_MAX = 5,         /* cap on the table slices per LDS image (forces S > 1 on small tables) */
    TSQ_KNOB_RADIX_PB_MAX = 6,       /* cap on log2(partitions) of the 64-bit radix probe */
    TSQ_KNOB_TABLE_LF_PERMILLE = 7,  /* load factor of a sliced join table, in 1/1000 (default 750) */
    TSQ_KNOB_LDS_PROF = 8,           /* 1: per-phase shader cycles of the LDS probe on stderr (synchronises) */
    TSQ_KNOB_DA_TRACE = 9,           /* 1: host time points of the travelling-columns route on stderr */
    TSQ_KNOB_BUILD_IMAGES_CAS = 10,  /* 1: the first (compare-and-swap) slice-image kernel of the partitioned build */
    TSQ_KNOB_DAAGG_SIG = 11,         /* 0: no plan-specialised instantiations of k_agg_da; 1: SUM + COUNT(*) plans; 2 (default since round 6): also count and sum of 2-byte cells in one LDS word */
    TSQ_KNOB_DAAGG_LOG2C = 12,       /* log2(cells per LDS table) of the packed aggregate, 9..12 */
    TSQ_KNOB_AGG_HEAP_GC_BYTES = 13, /* string-heap size from which the aggregate compacts between batches (default 256 MiB) */
    TSQ_KNOB_AGG_TAG_BITS = 14,      /* truncate the 64-bit group tag of a several-column GROUP BY (collision tests) */
    TSQ_KNOB_AGG_BATCH_ROWS = 15,    /* device batch of host-pushed aggregate input (default 4 Mi rows) */
    TSQ_KNOB_ROWCODEC_LDS_KB = 16,   /* LDS tile of the stored-row decoder */
    TSQ_KNOB_ROWCODEC_FAST_LAYOUT = 17, /* 0: every wave takes the per-row column search */
    TSQ_KNOB_ROWCODEC_PIPELINE = 18, /* 0: the un-pipelined decoder kernel */
    TSQ_KNOB_DA_PARTITION = 19,      /* packed partition kernel: 0 = default per entry width, 1 = one 1024-thread workgroup per CU, 2 = two of 512 */
    TSQ_KNOB_DA_NT_LOADS = 20,       /* 0: plain instead of non-temporal key loads in k_da_partition2 */
    TSQ_KNOB_LAZY_TABLE = 21,        /* 0: tsq_join_build_finish always builds the 64-bit table (default: a build side the packed routes are likely to
                                        serve leaves it to the first probe batch that needs it) */
    TSQ_KNOB_DA_PAIRS_BELOW_PERMILLE = 22, /* AUTO: a probe batch whose sampled hit ratio lies below this (in 1/1000, default 350) takes the pairs
                                        variant of the materialising packed route (K4d) instead of the travelling columns (K5f + K4e) */
    TSQ_KNOB_AGG_WIDE_KEYS = 23,     /* 0: several integer group-key columns never become one 64-bit composite key (the several-column upsert keeps them) */
    TSQ_KNOB_AGG_DENSE = 24,         /* 0: the one-key packed aggregate appends partial groups after every batch instead of folding its LDS tables into the dense state; v > 1 (tests): the state is emptied into the table before more than v rows went into it (default 2^31) */
    TSQ_KNOB_AGG_NARROW_CELLS = 25,  /* 0: the argument column of the packed aggregate always travels as 8-byte cells */
    TSQ_KNOB_DAAGG_PART2 = 26,       /* partition kernel of the packed aggregate with a dense state: 0 = 1024 threads, one workgroup per CU; 1 = two 512-thread workgroups per CU for narrow argument cells (default); 2 = for 8-byte cells too */
    TSQ_KNOB_DAAGG_HOT = 27,         /* 0: the packed aggregate does not sample the batch for hot keys (their rows then travel through the partitioned store and its overflow store) */
    TSQ_KNOB_KEYREC = 28,            /* 0: joins on several key columns / string keys never take the key-record route (csrc/tsq_keyrec.h), aggregates never the dictionary of group keys (csrc/tsq_keydict.h); 2: the dictionary's scatter pass writes separate arrays instead of 64-byte slots; 3 (tests): the digest of a long string cell carries its length only, so that every two cells of one length are candidates and the byte comparison decides */
    TSQ_KNOB_STREAMAGG_LANES = 29,   /* 0: StreamAggExec always reduces every 64-row step across the lanes (k_sa_update) instead of keeping per-lane partial results of the open run (k_sa_update_lanes) */
    TSQ_KNOB_XCD_ATOMICS = 30,       /* retired in round 6 (was: workgroup-scope cursor atomics, an A/B that measured equal); setting it has no effect */
    TSQ_KNOB_DENSE_DIRECT = 31,      /* 0: the packed aggregate's dense state always leaves through partial groups and the hash table, also when the table is empty (k_dense_finalize off) */
    TSQ_KNOB_DA_LDS_BUILD = 32,      /* 0: the materialising packed join never keeps a partition's build rows in LDS (csrc/tsq_damat.h): unique build sides take the sorted-build-columns variant of round 4 like the others; 2 .. 5 (tests): the second partition level splits 1 / 2 / 4 / 8 ways whatever the build side's size */
    TSQ_KNOB_AGG_PG = 33,            /* 0: an aggregate with about as many groups as rows never keeps its groups in partitioned LDS-sized sub-tables (csrc/tsq_aggfast.h K7p): the row upsert serves it; v >= 2 (tests): the mode is taken whatever the estimate, with 2^(v - 2) sub-tables */
    TSQ_KNOB_AGG_OVERLAP = 34,       /* default 0: the packed aggregate with a dense state runs every batch on one stream (partition pass, then k_agg_da); 1: k_agg_da / k_daagg_ovf of a batch run on a side stream beside the partition pass of the next batch (two partitioned stores) for batches of 2^24 rows or more — an A/B that measured SLOWER (C3 7.0 -> 10.1 ms, profiles/r06_ab_measurements.txt); v >= 2 (tests): for batches of v rows or more */
    TSQ_KNOB_JIT_VARIANT = 35,       /* A/B bits of the hiprtc-specialised projection kernel (jit_expr), default 391 = 7 | 128 | 256 (measured 0.525 -> 0.453 ms per 1e8 rows of (a+b)*3-a against 0, the filter a<b AND c>0.5 0.463 -> 0.429 ms, profiles/r06_jit_sweep.txt): 1 = non-temporal loads of the input cells, 2 = non-temporal stores of the result, 4 = whole-wave coalesced 16-byte accesses (a lane takes rows 2 l, 2 l + 1 of each 128-row half of a 256-row step instead of four consecutive rows), 8 = two steps' loads in flight; bits 4-6: workgroups per CU = 8 (0), 4, 16, 32, 2; 128 = jit_filter reads its 8-byte cells with non-temporal loads, 256 = and writes the selected byte with a non-temporal store */
    TSQ_KNOB_HOST_OVERLAP = 36,      /* bits of the pipelining of a join fed with HOST chunks (default 3; 0 = rounds 1-5: one stream, every result batch copied to pinned memory and waited for): 1 = the D2H copies run on the operator's copy stream beside the staging, H2D and kernels of the next batch, a flush waits for its H2D copies only, and tsq_join_pull answers "no rows yet" while the front batch is still on its way and the probe side is not finished; 2 = the H2D copies of staged probe rows are queued every 256 Ki rows while the caller still pushes (a software prefetch of the next pull's first lines was measured without effect and removed) */
    TSQ_KNOB_HOST_NT_COPY = 37,      /* 0: host chunks enter the pinned staging buffers through memcpy instead of non-temporal stores (process-wide) */
    TSQ_KNOB_KR_WG = 38,             /* workgroups (contiguous row chunks) of the key-record hist / scatter passes, 8..256 (default 256): fewer workgroups keep fewer partition lines open at once (A/B, profiles/r06_keyrec_ab.txt) */
    TSQ_KNOB_COUNT = 48
};
tsq_status tsq_ctx_set_knob(tsq_ctx* ctx, int32_t knob, int64_t value);
tsq_status tsq_ctx_arena_stats(tsq_ctx* ctx, int64_t* size_out, int64_t* used_out, int64_t* peak_out);

/* Device memory + copies for harnesses that keep tables resident in HBM (bench, multi-GPU). */
tsq_status tsq_dev_alloc(tsq_ctx* ctx, int64_t bytes, void** out);
tsq_status tsq_dev_free(tsq_ctx* ctx, void* p);
tsq_status tsq_dev_memset(tsq_ctx* ctx, void* p, int32_t byte, int64_t bytes);
/* Pinned host memory (ABI 6) for the chunks a host pushes or pulls and for tsq_copy_h2d / tsq_copy_d2h: the DMA engines read and write
 * it directly, a pageable Go / numpy buffer goes through a bounce buffer at a fraction of the link rate (BASELINE north star: "chunk.Column
 * batches pinned and DMA'd to HBM").  The cgo shim keeps its chunk.Column data in such blocks; freed blocks return to
 * a process-wide pool. */
tsq_status tsq_host_alloc(tsq_ctx* ctx, int64_t bytes, void** out);
tsq_status tsq_host_free(tsq_ctx* ctx, void* p);
tsq_status tsq_copy_h2d(tsq_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
tsq_status tsq_copy_d2h(tsq_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
/* device to device, queued on the context's stream (not synchronised): an operator's output batch kept beyond its next Next() */
tsq_status tsq_copy_d2d(tsq_ctx* ctx, void* dst_dev, const void* src_dev, int64_t bytes);
/* rows [dst_rows, dst_rows + n) of a device null bitmap := the first n bits of src_bitmap (device; NULL: n NOT-NULL bits) — the bit
 * offset is arbitrary (Column.appendNullBitmap, util/chunk/column.go; chunk.Decoder.decodeColumn's shift-and-or, codec.go:325-343).
 * Queued on the context's stream.  The destination must hold (dst_rows + n + 7) / 8 + 8 bytes. */
tsq_status tsq_bitmap_append(tsq_ctx* ctx, uint8_t* dst_bitmap, int64_t dst_rows, const uint8_t* src_bitmap, int64_t n);

/* Timing helper: HIP events recorded on the ctx stream (bench.py's roofline leg). */
tsq_status tsq_timer_start(tsq_ctx* ctx);
tsq_status tsq_timer_stop_ms(tsq_ctx* ctx, double* ms_out); /* synchronises the stop event */

/* ---------------------------------------------------------------- synthetic tables
 * Counter-based generator, identical on host (oracle) and device:
 *   r(i,c) = splitmix64(seed ^ (table<<56) ^ (c<<48) ^ i)
 * Replaces the reference's mockDataSource generators (executor/benchmark_test.go:50-177). */
enum {
    TSQ_GEN_SEQ = 0,        /* v = start + i                                   (benchmark_test.go:396-407) */
    TSQ_GEN_AFFINE = 1,     /* v = (a*i + b) mod m : bijection on [0,m) when gcd(a,m)=1, m < 2^31        */
    TSQ_GEN_RAND_MOD = 2,   /* v = r(i,c) mod m                                                           */
    TSQ_GEN_RAND_F64 = 3,   /* v = (r(i,c)>>11) * 2^-53 in [0,1)  (rand.Float64, benchmark_test.go:118)   */
    TSQ_GEN_HASH_OF_COL = 4,/* v = splitmix64(src[i] ^ b): payload that is a function of another column  */
    TSQ_GEN_ZIPF_OCT = 5    /* skewed keys with the s = 1 harmonic envelope, piecewise constant per octave: an octave e is drawn uniformly
                               from [0, a), then a value uniformly from [2^e, 2^(e+1)); v = (that - 1) mod m.  P(v) ~ 1 / (a 2^floor(log2(v+1))):
                               key 0 takes 1 / a of the rows (C3 with Zipf s = 1.0 keys)                */
};
typedef struct tsq_gen_spec {
    int32_t  kind;
    int32_t  table;      /* t */
    int32_t  col;        /* c */
    int32_t  null_pct;   /* 0..100: row is NULL iff r(i,7) % 100 < null_pct */
    uint64_t seed;
    int64_t  start;      /* SEQ start / global row offset i0 for all kinds (row index = i0 + i) */
    uint64_t a, b, m;    /* AFFINE / RAND_MOD parameters */
} tsq_gen_spec;
/* dst/null_bitmap/src are device pointers; null_bitmap may be NULL when null_pct == 0. */
tsq_status tsq_gen_column(tsq_ctx* ctx, const tsq_gen_spec* spec, int64_t nrows,
                          void* dst, uint8_t* null_bitmap, const void* src);

/* ---------------------------------------------------------------- vectorized expressions
 * Replaces expression.VecExpr / VecEval{Int,Real} (expression/expression.go:43-55,329-341),
 * VectorizedFilter (chunk_executor.go:196-245) and EvaluatorSuite.Run (evaluator.go:121-133):
 * the Go side flattens an expression tree to postfix bytecode; ONE fused kernel evaluates the
 * whole tree per row instead of one pass per node.  Each opcode is one reference signature. */
enum {
    /* leaves */
    TSQ_OP_COL_INT = 1,     /* arg = column index            (expression/column.go:56-75)          */
    TSQ_OP_COL_REAL = 2,    /* arg = column index; F32 widened (column.go:81-104)                  */
    TSQ_OP_CONST_INT = 3,   /* arg = const index             (constant.go:76-88, vectorized.go:23) */
    TSQ_OP_CONST_REAL = 4,
    TSQ_OP_CONST_NULL_INT = 5,
    TSQ_OP_CONST_NULL_REAL = 6,
    /* arithmetic (builtin_arithmetic_vec.go) */
    TSQ_OP_PLUS_REAL = 10,  /* :275 */
    TSQ_OP_MINUS_REAL = 11, /* :62  */
    TSQ_OP_MUL_REAL = 12,   /* :29  */
    TSQ_OP_DIV_REAL = 13,   /* :348 */
    TSQ_OP_PLUS_INT = 14,   /* :389 (+plusUU/US/SU/SS :428-495), flags = unsigned bits           */
    TSQ_OP_MINUS_INT = 15,  /* :95  (+minusFUU..SS :142-271), flags = unsigned bits|FORCE_SIGNED  */
    TSQ_OP_MUL_INT = 16,    /* :308 */
    TSQ_OP_MUL_INT_UNSIGNED = 17, /* :501 */
    /* compare (builtin_compare_vec.go:26-292, builtin_compare_vec_generated.go) */
    TSQ_OP_LT_INT = 20, TSQ_OP_LE_INT = 21, TSQ_OP_GT_INT = 22, TSQ_OP_GE_INT = 23,
    TSQ_OP_EQ_INT = 24, TSQ_OP_NE_INT = 25,
    TSQ_OP_LT_REAL = 26, TSQ_OP_LE_REAL = 27, TSQ_OP_GT_REAL = 28, TSQ_OP_GE_REAL = 29,
    TSQ_OP_EQ_REAL = 30, TSQ_OP_NE_REAL = 31,
    /* logic / unary (builtin_op_vec.go) */
    TSQ_OP_LOGIC_AND = 40,  /* :173 */
    TSQ_OP_LOGIC_OR = 41,   /* :29  */
    TSQ_OP_NOT_INT = 42,    /* :249 */
    TSQ_OP_NOT_REAL = 43,   /* :141 */
    TSQ_OP_NEG_INT = 44,    /* :221, flags bit0 = arg unsigned */
    TSQ_OP_NEG_REAL = 45,   /* :74  */
    TSQ_OP_ISNULL_INT = 46, /* :92  */
    TSQ_OP_ISNULL_REAL = 47,/* :113 */
    /* control (builtin_control_vec_generated.go) */
    TSQ_OP_IFNULL_INT = 50, /* :23  */
    TSQ_OP_IFNULL_REAL = 51,/* :52  */
    TSQ_OP_IF_INT = 52,     /* :117 (cond is Int; value args Int)  */
    TSQ_OP_IF_REAL = 53,    /* :163 */
    /* IN (builtin_other_vec_generated.go); arg = number of list items n (stack: x, v1..vn) */
    TSQ_OP_IN_INT = 60,     /* :24, flags bit0 = x unsigned; unsigned-ness of item j in in_unsigned_mask bit j */
    TSQ_OP_IN_REAL = 61,    /* :151 */
    /* strings (types.EvalType ETString), binary collation: types.CompareString = bytes.Compare.  A string VALUE on the evaluation
     * stack is a reference (source column or the program's constant pool, byte offset, length): no bytes are copied. */
    TSQ_OP_COL_STR = 70,        /* arg = column index (TSQ_BYTES)        (expression/column.go:106-129)              */
    TSQ_OP_CONST_STR = 71,      /* arg = const index: consts[arg] = (offset into str_pool << 32) | length            */
    TSQ_OP_CONST_NULL_STR = 72,
    TSQ_OP_LT_STR = 73, TSQ_OP_LE_STR = 74, TSQ_OP_GT_STR = 75, TSQ_OP_GE_STR = 76,  /* builtin_compare_vec_generated.go:65,147,229,311 */
    TSQ_OP_EQ_STR = 77, TSQ_OP_NE_STR = 78,                                          /* :393, :475                                      */
    TSQ_OP_STRCMP = 79,         /* builtin_string_vec.go:52  -> -1 / 0 / 1                                          */
    TSQ_OP_LENGTH = 80,         /* builtin_string_vec.go:89  -> bytes                                               */
    TSQ_OP_ISNULL_STR = 81,     /* builtin_string_vec.go:21 (builtinStringIsNullSig)                                */
    TSQ_OP_IFNULL_STR = 82,     /* builtin_control_vec_generated.go:81                                              */
    TSQ_OP_IF_STR = 83,         /* builtin_control_vec_generated.go:209 (cond is Int; value args String)            */
    TSQ_OP_IN_STR = 84          /* builtin_other_vec_generated.go:97; arg = number of list items                    */
};
/* tsq_expr_op.flags */
#define TSQ_F_LHS_UNSIGNED 1u
#define TSQ_F_RHS_UNSIGNED 2u
#define TSQ_F_FORCE_SIGNED 4u /* SQLMode NO_UNSIGNED_SUBTRACTION (builtin_arithmetic_vec.go:119) */

typedef struct tsq_expr_op {
    uint8_t  opcode;
    uint8_t  flags;
    uint16_t arg;      /* column index / const index / IN item count */
    uint32_t aux;      /* IN_INT: bitmask of unsigned list items (bit j-1 for item j) */
} tsq_expr_op;

#define TSQ_EXPR_MAX_OPS 64
#define TSQ_EXPR_MAX_STACK 12
#define TSQ_EXPR_MAX_CONSTS 32
#define TSQ_EXPR_STR_POOL 256

/* One expression tree in postfix form.  result_type: TSQ_I64 (Int, also used for U64), TSQ_F64, or TSQ_BYTES for a STRING-valued
 * root — builtinIfStringSig / builtinIfNullStringSig.vecEvalString (builtin_control_vec_generated.go:209, :81), a string column
 * (Column.VecEvalString, column.go:111) or constant (constant.go:86) — which tsq_expr_eval_str evaluates into a var-len column;
 * tsq_expr_eval, tsq_filter_eval and the join's conditions / filters take Int and Real roots only. */
typedef struct tsq_expr_prog {
    int32_t     n_ops;
    int32_t     n_consts;
    int32_t     result_type;
    int32_t     result_unsigned; /* UNSIGNED flag of the result FieldType (informational) */
    tsq_expr_op ops[TSQ_EXPR_MAX_OPS];
    int64_t     consts[TSQ_EXPR_MAX_CONSTS]; /* int64, the bit pattern of a double, or (offset << 32 | length) of a string constant */
    int32_t     n_str_bytes;                 /* bytes used in str_pool */
    int32_t     reserved;
    uint8_t     str_pool[TSQ_EXPR_STR_POOL]; /* the bytes of the string constants */
} tsq_expr_prog;

typedef struct tsq_expr tsq_expr;

/* Compile = validate + upload.  n_progs > 1 is a CNF list (expression.CNFExprs) for filters. */
tsq_status tsq_expr_compile(tsq_ctx* ctx, const tsq_expr_prog* progs, int32_t n_progs, tsq_expr** out);
/* Projection form (evaluator.go:121 / expression.go:329 VecEval): evaluates progs[0] over
 * nrows logical rows; `sel` (optional, int32[nrows]) is chunk.Chunk.sel (chunk.go:319-331).
 * out->data/out->null_bitmap must be sized for nrows; out->type F64 or I64.
 * div_by_zero_warnings (optional) receives the number of x/0 rows (errors.go:65-77). */
tsq_status tsq_expr_eval(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows,
                         const int32_t* sel, tsq_col* out, int64_t* div_by_zero_warnings);
/* String-valued root (prog.result_type TSQ_BYTES): the result is a var-len column — ProjectionExec's evaluatorSuite on an ETString
 * expression (executor/projection.go:160, expression/evaluator.go:54-133 -> VecEvalString, expression.go:329-341).  Replaces
 * builtinIfStringSig / builtinIfNullStringSig.vecEvalString (expression/builtin_control_vec_generated.go:209, :81: per row
 * AppendNull or AppendString of the chosen argument), Column.VecEvalString (column.go:111: CopyReconstruct through sel) and
 * Constant.VecEvalString (constant.go:86: the value n times).  out->offsets holds nrows + 1 entries, out->data cap_bytes bytes,
 * out->null_bitmap (nrows + 7) / 8 bytes; same placement (host / TSQ_COL_DEVICE) as the inputs.  *bytes_out = the data bytes of
 * the result; cap_bytes < that (0: only ask) -> nothing but *bytes_out is written and the call returns TSQ_ERR_INVALID.  A NULL row
 * has no bytes (offsets repeat) and a clear bitmap bit, as AppendNull leaves it (util/chunk/column.go:150-158). */
tsq_status tsq_expr_eval_str(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows, const int32_t* sel, tsq_col* out,
                             int64_t cap_bytes, int64_t* bytes_out, int64_t* div_by_zero_warnings);
/* Filter form (chunk_executor.go:196 VectorizedFilter / expression.go:205 VecEvalBool):
 * selected_out[i] (one byte per row, Go []bool) = row passes every conjunct, non-NULL.
 * isnull_out (optional) mirrors VecEvalBool's `nulls` for Int-typed conjuncts. */
tsq_status tsq_filter_eval(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows,
                           const int32_t* sel, uint8_t* selected_out, uint8_t* isnull_out,
                           int64_t* div_by_zero_warnings);
/* Kernel selection.  TSQ_JIT_AUTO (default): once a handle has seen >= 256 Ki rows its programs are compiled into
 * specialised kernels with hiprtc (same source as the interpreter, programs as compile-time constants: the node loop
 * unrolls and every opcode switch folds) — on a helper thread: no call waits for the compile (~250 ms per distinct tree and
 * context, cached), the generic interpreter kernels serve the handle until the code object is there, and they keep serving
 * smaller inputs and any hiprtc failure (tsq_last_error(e) after tsq_expr_jit_launches tells why).  TSQ_JIT_FORCE compiles
 * inside the first call.  Results are identical by construction. */
#define TSQ_JIT_AUTO  (-1)
#define TSQ_JIT_OFF     0
#define TSQ_JIT_FORCE   1
tsq_status tsq_expr_set_jit(tsq_expr* e, int32_t mode);
int64_t    tsq_expr_jit_launches(tsq_expr* e);   /* number of launches served by specialised kernels so far */
double     tsq_expr_jit_compile_ms(tsq_expr* e); /* ABI 7: what hiprtc + the module load of this handle's programs took (0: not compiled yet, or found
                                                  * in the context's cache of program sets; a tree is compiled once per context) */
void       tsq_expr_destroy(tsq_expr* e);

/* ---------------------------------------------------------------- hash join
 * Replaces HashJoinExec (executor/join.go:31-60): fetchAndBuildHashTable (:148-158, STUB in the
 * reference), hashRowContainer.PutChunk/GetMatchedRows (hash_table.go:110-169), join2Chunk
 * (join.go:325-362) and the joiners (joiner.go:220-410). */
enum { TSQ_JOIN_INNER = 0, TSQ_JOIN_LEFT_OUTER = 1, TSQ_JOIN_RIGHT_OUTER = 2 }; /* joiner.go:105-116 */

#define TSQ_MAX_KEYS 4
#define TSQ_MAX_COLS 16

typedef struct tsq_join_cfg {
    int32_t join_type;
    /* which child is the build (inner/hashed) side: 1 = right child (probe = left).
     * Output column order is ALWAYS left-child cols || right-child cols (joiner.go:145-150). */
    int32_t build_is_right;
    int32_t n_keys;
    int32_t build_key_idx[TSQ_MAX_KEYS];
    int32_t probe_key_idx[TSQ_MAX_KEYS];
    int32_t n_build_cols;
    int32_t n_probe_cols;
    int32_t build_types[TSQ_MAX_COLS];  /* TSQ_I64.. per build-side column */
    int32_t probe_types[TSQ_MAX_COLS];
    int64_t est_build_rows;             /* innerSideEstCount (hash_table.go:84-96); 0 = unknown */
    int32_t max_chunk_size;             /* tidb_max_chunk_size (tidb_vars.go:242), default 1024 */
    int32_t concurrency;                /* tidb_hash_join_concurrency; unused on the GPU (kept for parity of cfg) */
    int64_t probe_batch_rows;           /* rows staged per device batch; 0 = default (4Mi) */
    /* OtherConditions evaluated on the joined row (lhs||rhs) (joiner.go:155-167); may be NULL */
    const tsq_expr_prog* other_conds;
    int32_t n_other_conds;
    /* outerSideFilter over the probe-side schema (join.go:328); may be NULL */
    const tsq_expr_prog* outer_filters;
    int32_t n_outer_filters;
} tsq_join_cfg;

typedef struct tsq_join tsq_join;

tsq_status tsq_join_create(tsq_ctx* ctx, const tsq_join_cfg* cfg, tsq_join** out);
/* Build side: one call per inner-child chunk (hashRowContainer.PutChunk, hash_table.go:146).
 * Rows whose key has a NULL are kept for outer-join output but never inserted (:161-163). */
tsq_status tsq_join_build_push(tsq_join* j, const tsq_col* cols, int32_t n_cols, int64_t nrows);
tsq_status tsq_join_build_finish(tsq_join* j);
/* The build side of a COUNT(*) join SHARDED over the GPUs of a node, without moving a single probe row.
 * COLLECTIVE: every rank of `comm` pushes ITS build rows (tsq_join_build_push) and then calls this instead of
 * tsq_join_build_finish.  The ranks agree on the key range of the whole build side (two 8-byte all-reduces), every rank
 * assembles the packed direct-address images (csrc/tsq_dajoin.h) of its rows over that range, and the images are summed across
 * the ranks with ONE all-reduce (2^b bytes, or 2^b / 8 for a build side without duplicate keys: 128 MiB for 8e8 keys).  Every
 * rank then holds the images of the WHOLE build side: tsq_join_probe_push takes the rank's OWN probe rows, tsq_join_count the
 * rank's joined rows (the plan adds them up, tsq_comm_allreduce_i64) — the probe phase puts nothing on xGMI and scales with the
 * number of GPUs.  Replaces the N join workers probing ONE shared, read-only hashRowContainer (executor/join.go:233-239,
 * hash_table.go:110-134): the images are that container, replicated because HBM is not shared between GPUs.
 * *shared_out = 1: done — the handle is count-only.  *shared_out = 0 (the same on every rank): the build side is not packable
 * (outer join / conditions / several or non-integer key columns, a key range beyond 31 bits or too sparse, more than 255 build
 * rows per key, duplicate keys in a range beyond 28 bits) — nothing was built, the handle still holds the pushed rows
 * (tsq_join_build_finish makes it a LOCAL join); a distributed plan destroys it and redistributes both sides by rank(key)
 * (tsq_redistribute), as tinysql_amd/parallel.py does. */
typedef struct tsq_comm tsq_comm;
tsq_status tsq_join_build_finish_shared(tsq_join* j, tsq_comm* comm, int32_t* shared_out);
/* Probe side: one call per outer-child chunk (join2Chunk, join.go:325). `selected` (optional,
 * one byte per row) is an externally evaluated outer-side filter; rows with selected==0 or a
 * NULL key go to onMissMatch (join.go:344-345). */
tsq_status tsq_join_probe_push(tsq_join* j, const tsq_col* cols, int32_t n_cols, int64_t nrows,
                               const uint8_t* selected);
tsq_status tsq_join_probe_finish(tsq_join* j);
/* HashJoinExec.Next (join.go:125-146): fills up to cap_rows joined rows into out_cols
 * (n_cols == n_probe_cols + n_build_cols, left||right order, host or device buffers).
 * *nrows_out == 0 && *eos == 0  -> needs more probe input;  *eos == 1 -> end of stream. */
tsq_status tsq_join_pull(tsq_join* j, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows,
                         int64_t* nrows_out, int32_t* eos);
/* Var-len (TSQ_BYTES) columns travel through the join as payload (util/chunk/column.go:28-34: offsets + data; Chunk.AppendRow,
 * chunk.go:334-356); join KEYS are fixed width.  A pull fills out_cols[c].offsets (cap_rows + 1 entries) and .data of such a
 * column; tsq_join_peek tells, for the next pull of up to cap_rows rows, how many rows it will deliver and how many data bytes
 * each var-len output column needs (bytes_out[c]; 0 for fixed-width columns), so the caller can size .data first. */
tsq_status tsq_join_peek(tsq_join* j, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols);
/* COUNT(*) fast path (config C1: SELECT count(*) FROM t1 JOIN t2 ON t1.k=t2.k): number of
 * joined rows produced so far by probe_push'd input, without materialising them.
 * Valid instead of (not mixed with) tsq_join_pull. */
/* Inline projection (column pruning, planner/core/rule_column_pruning.go): used[c] = 0 tells the operator that the parent never
 * reads output column c (left child's columns, then right child's): routes that gather their output through (probe row,
 * build row) pairs do not materialise it, tsq_join_pull does not touch its buffer (TSQ_COL_BORROW: data = NULL).  A route that
 * writes whole rows anyway may still fill it.  used = NULL: every column is used (default).  Before the first probe row. */
tsq_status tsq_join_set_used_columns(tsq_join* j, const uint8_t* used, int32_t n_cols);
tsq_status tsq_join_set_count_only(tsq_join* j, int32_t on);
tsq_status tsq_join_count(tsq_join* j, int64_t* rows_out);
/* Order-independent checksum of the joined rows seen so far in count-only mode:
 * sum and xor over joined rows of rowhash(all output columns) — parity at full size. */
tsq_status tsq_join_checksum(tsq_join* j, uint64_t* sum_out, uint64_t* xor_out);
tsq_status tsq_join_set_checksum(tsq_join* j, int32_t on);
/* Probe strategy of the COUNT(*) fast path.  TSQ_RADIX_AUTO (default): probe batches that are large
 * enough are radix partitioned by the top bits of the key hash (LDS-staged, write combined) and probed
 * partition by partition so that every XCD works inside a table slice that fits its L2; small batches
 * and small tables take the direct probe.  The same switch, when set before tsq_join_build_finish, selects
 * the BUILD strategy: AUTO assembles the table slice by slice in LDS (two radix passes over key words +
 * row ids, then one LDS image per slice) for single-key builds of >= 4 Mi rows and inserts row by row
 * (64-bit CAS) otherwise.  OFF / FORCE exist for tests and measurements; the joined rows are identical
 * either way.  Replaces the worker dispatch of executor/join.go:160-231 and PutChunk (hash_table.go:146-169). */
#define TSQ_RADIX_AUTO  (-1)
#define TSQ_RADIX_OFF     0
#define TSQ_RADIX_FORCE   1
tsq_status tsq_join_set_radix(tsq_join* j, int32_t mode);
/* Key packing of the radix probe (round 3; csrc/tsq_dajoin.h).  The build side is complete before the first probe row, so the
 * range [kmin, kmax] of its key column is known.  TSQ_RADIX_AUTO (default): when the key is ONE integer column on both sides,
 * the range fits 28 bits, holds >= 1 build row per 32 values and no key has more than 255 build rows, a COUNT(*) probe batch
 * travels as 2-byte entries (bijective mix of key - kmin; partition = its top bits) and meets a direct-address table of one
 * byte per value of the range, one 64 KB image per partition in LDS — equality of entries IS equality of keys
 * (util/codec/codec.go:363-382), a probe key outside the range joins nothing.  A build side WITHOUT duplicate keys may span up to
 * 31 bits for a COUNT(*) probe: one BIT per value (4-byte entries).  Materialising probes of inner / outer joins (8-byte columns,
 * OtherConditions of inner joins — and of outer joins over a build side without duplicate keys — included) take the same entries with the probe columns travelling next to them (<= 27 bits).
 * Several integer key columns (<= 4, fields adding up to <= 28 bits) are composed into one key column first and take the same routes.
 * OFF keeps 64-bit table words; FORCE drops the
 * size and density conditions (tests).  Must be chosen before the first probe batch.  The joined rows are identical either way.
 * Replaces join2Chunk + GetMatchedRows (executor/join.go:343-360, hash_table.go:110-134) for that shape. */
tsq_status tsq_join_set_key_packing(tsq_join* j, int32_t mode);
/* Ordered output: joined rows come out in probe-row order (the order of the probe pushes and of the rows inside them), the
 * matches of one probe row in build-row order (the order of the build pushes).  With both children sorted on the join keys
 * and the OUTER child as probe side this is exactly MergeJoinExec's output order (executor/merge_join.go:257-310: outer
 * rows in order, each with its inner group in order, NULL-key inner rows skipped :148-156, unmatched outer rows padded for
 * outer joins) — the GPU operator does not need the inputs to be sorted to produce it.  Pulls then also preserve it. */
tsq_status tsq_join_set_ordered(tsq_join* j, int32_t on);
tsq_status tsq_join_cancel(tsq_join* j);
void       tsq_join_destroy(tsq_join* j);

/* ---------------------------------------------------------------- hash aggregation
 * Replaces HashAggExec (executor/aggregate.go:134-155): partial workers' updatePartialResult
 * (:332-350), getGroupKey (:359-394), shuffleIntermData (:354, STUB), consumeIntermData (:424,
 * STUB), getFinalResult (:429-457) and the executor/aggfuncs package. */
enum {
    TSQ_AGG_COUNT = 0,      /* aggfuncs/func_count.go    */
    TSQ_AGG_SUM = 1,        /* aggfuncs/func_sum.go      */
    TSQ_AGG_AVG = 2,        /* aggfuncs/func_avg.go      */
    TSQ_AGG_MAX = 3,        /* aggfuncs/func_max_min.go  */
    TSQ_AGG_MIN = 4,
    TSQ_AGG_FIRSTROW = 5    /* aggfuncs/func_first_row.go */
};
/* expression/aggregation/aggregation.go:82-97 */
enum { TSQ_MODE_COMPLETE = 0, TSQ_MODE_FINAL = 1, TSQ_MODE_PARTIAL1 = 2, TSQ_MODE_PARTIAL2 = 3 };

#define TSQ_MAX_AGGS 16
#define TSQ_MAX_GROUP_KEYS 4

typedef struct tsq_agg_func {
    int32_t func;       /* TSQ_AGG_*                                                          */
    int32_t mode;       /* TSQ_MODE_*; COMPLETE/PARTIAL1 read raw args, FINAL/PARTIAL2 merge   */
    int32_t arg_col;    /* input column; -1 = constant non-NULL arg (COUNT(*) == count(1),
                           parser/parser.y:3258-3262)                                         */
    int32_t arg_col2;   /* AVG in FINAL/PARTIAL2 mode: arg_col = count column, arg_col2 = sum
                           column (func_avg.go:86-113, descriptor.go:70-81)                    */
    int32_t arg_type;   /* TSQ_I64/U64/F32/F64 of the value argument; TSQ_BYTES for COUNT / MAX / MIN /
                           FIRST_ROW of a string (func_max_min.go:312-378, func_first_row.go:193-230)  */
    int32_t reserved;
} tsq_agg_func;

typedef struct tsq_agg_cfg {
    int32_t n_group_keys;
    int32_t group_key_col[TSQ_MAX_GROUP_KEYS];  /* bare input columns (group-by expressions are
                                                   pre-evaluated with tsq_expr_eval)           */
    int32_t group_key_type[TSQ_MAX_GROUP_KEYS];
    int32_t n_aggs;
    tsq_agg_func aggs[TSQ_MAX_AGGS];
    int32_t n_input_cols;
    int32_t input_types[TSQ_MAX_COLS];
    int64_t est_groups;       /* 0 = unknown (table grows by rehash) */
    int32_t max_chunk_size;
    int32_t reserved;
} tsq_agg_cfg;

typedef struct tsq_agg tsq_agg;

tsq_status tsq_agg_create(tsq_ctx* ctx, const tsq_agg_cfg* cfg, tsq_agg** out);
/* One call per child chunk (HashAggPartialWorker.updatePartialResult, aggregate.go:332). */
tsq_status tsq_agg_push(tsq_agg* a, const tsq_col* cols, int32_t n_cols, int64_t nrows);
/* End of child input: finalise groups (consumeIntermData + getFinalResult, aggregate.go:424-457).
 * Returns TSQ_ERR_OVERFLOW_BIGINT if an int64 SUM/AVG left the BIGINT range (func_sum.go:133). */
tsq_status tsq_agg_finish(tsq_agg* a);
/* Update strategy.  TSQ_AGGFAST_AUTO (default): large batches of a single-key aggregate whose functions
 * fit (COUNT/SUM/AVG/MIN/MAX over <= 2 argument columns, firstrow(group key)) are pre-aggregated in LDS
 * — directly when few groups are expected, after a radix partition by group-key hash otherwise — and
 * only the partial groups are merged into the table in HBM (the reference's partial -> shuffle -> final
 * shape, executor/aggregate.go:96-133).  OFF / FORCE exist for tests and measurements; results are the
 * same up to the documented floating-point reordering of SUM/AVG(double). */
#define TSQ_AGGFAST_AUTO  (-1)
#define TSQ_AGGFAST_OFF     0
#define TSQ_AGGFAST_FORCE   1
tsq_status tsq_agg_set_fast(tsq_agg* a, int32_t mode);
/* StreamAggExec (BASELINE.json north star; the reference has only the plan name, planner/core/cbo_test.go:200-212 "StreamAgg"): on != 0
 * before the first row tells the operator that its child delivers rows ORDERED by the group-by items — equal keys are adjacent (NULL
 * equals NULL, +0.0 equals -0.0: the group key of util/codec/codec.go:713-746).  A group is closed by the first row with another
 * key; tsq_agg_pull then returns the groups IN INPUT ORDER, and FIRST_ROW is the first row of its group.  Aggregate functions,
 * modes, NULL / overflow rules, the empty-input default row and the output schema are HashAggExec's (executor/aggregate.go:307-350,
 * 559-588).  No hash table is probed: a row's group = the groups closed so far + the group heads up to the row (csrc/tsq_streamagg.h);
 * a key that comes back after another key opens a NEW group (the child's order is the caller's contract, as in the planner's
 * property enforcement).  Sort an unordered child with tsq_sort_* first (tinysql_amd/executor.py StreamAggExec does). */
tsq_status tsq_agg_set_stream(tsq_agg* a, int32_t on);
/* After tsq_agg_finish: the number of result rows.  Before: the groups the operator's table holds so far — a LOWER bound (the packed
 * route keeps the partial state of a one-key GROUP BY outside the table until finish, tsq_stats.dense_flushes). */
tsq_status tsq_agg_num_groups(tsq_agg* a, int64_t* out);
/* HashAggExec.Next (aggregate.go:559-588): output schema = one column per agg func, in cfg
 * order.  COMPLETE/FINAL emit final values; PARTIAL1/PARTIAL2 emit partial columns (AVG emits
 * two: count then sum).  Group order is unspecified (Go map order in the reference). */
tsq_status tsq_agg_pull(tsq_agg* a, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows,
                        int64_t* nrows_out, int32_t* eos);
/* Strings in the aggregate: a group key may be TSQ_BYTES (getGroupKey: compactBytesFlag + bytes, util/codec/codec.go:738-744;
 * NULL is its own group) and FIRST_ROW / MAX / MIN (binary collation, types.CompareString) / COUNT take a TSQ_BYTES argument;
 * SUM / AVG of a string answer TSQ_ERR_UNSUPPORTED (the planner casts to double first).  The operator keeps the bytes of the
 * var-len input columns it was pushed until it is destroyed.  A pull fills out_cols[c].offsets (cap_rows + 1 entries) and .data
 * of a var-len output column; tsq_agg_peek (after tsq_agg_finish) tells how many rows the next pull of up to cap_rows rows
 * delivers and how many data bytes each var-len output column needs (bytes_out[c]; 0 for fixed-width columns). */
tsq_status tsq_agg_peek(tsq_agg* a, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols);
tsq_status tsq_agg_cancel(tsq_agg* a);
void       tsq_agg_destroy(tsq_agg* a);

/* ---------------------------------------------------------------- device-chunk hand-off between GPU operators
 * Dense copy of the selected rows of a DEVICE-resident chunk (selected[]: one byte per row, device memory, e.g. the
 * output of tsq_filter_eval on device columns).  Replaces SelectionExec's copy of selected rows (executor/executor.go:
 * 393-438) / Column.CopyReconstruct (util/chunk/column.go:504-552) when parent and child are both GPU operators, so a
 * filtered chunk reaches the join / aggregate without leaving HBM.  out_cols must be sized for nrows rows; rows keep
 * their order up to a permutation inside 256-row tiles (downstream hash operators are order-insensitive). */
tsq_status tsq_chunk_compact(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int64_t nrows, const uint8_t* selected,
                             tsq_col* out_cols, int64_t* nrows_out);

/* ---------------------------------------------------------------- coprocessor response rows -> columns (SURVEY.md §8 f, rank 2)
 * Replaces selectResult.readRowsData (distsql/select_result.go:139-155) + codec.Decoder.DecodeOne (util/codec/codec.go:
 * 623-690) for fixed-width schemas: `rows_data` is the RowsData byte string of a coprocessor response chunk — rows one
 * after the other, every value = flag byte + varint (8/9) | 8 big-endian bytes (3/4/5) | nothing (0 = NULL); both the
 * EncodeValue and the EncodeKey forms are accepted, like DecodeOne.  Decodes at most cap_rows rows into out_cols (host
 * or TSQ_COL_DEVICE buffers sized for cap_rows rows; data + null_bitmap), *bytes_consumed = length of the decoded
 * prefix (the caller keeps the remainder, select_result.go:153).  data_flags: TSQ_COL_DEVICE when rows_data is in HBM.
 * Errors are the reference's, decided by the FIRST offending value in stream order; *nrows_out then holds the complete
 * rows before it (already in out_cols) and *bytes_consumed is 0: TSQ_ERR_INVALID with tsq_last_error =
 * "invalid encoded key" (a row ends early, codec.go:625) | "insufficient bytes to decode value" (number.go:46,122) |
 * "value larger than 64 bits" (number.go:120) | "invalid encoded key flag" (codec.go:683).
 * col_types may contain TSQ_BYTES (ABI 6): a compact-bytes (flag 2, bytes.go:141-160) or memcomparable (flag 1, bytes.go:35-118) datum
 * becomes a var-len cell — out_cols of such a column need offsets[cap_rows + 1] and a data buffer of n_bytes bytes, as for
 * tsq_rows_decode_chunks, whose errors apply ("datum kind does not match the column type", DecodeBytes' messages).  A bytes datum has
 * no bounded length, so the row boundaries of such a stream come from one sequential walk over flags and lengths on the host (a
 * stream in HBM is copied back for it); the values are decoded on the GPU, 64 rows per piece (csrc/tsq_decode.hip). */
tsq_status tsq_rows_decode(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, uint32_t data_flags, int32_t n_cols,
                           const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows, int64_t* nrows_out,
                           int64_t* bytes_consumed);

/* The same for a response whose CHUNKS are known, var-len columns included.  The storage side cuts the rows of a response into
 * tipb.Chunks of 64 rows (cop_handler_dag.go:510-519) — selectResult walks them one after the other (select_result.go:102-155).
 * Chunk k = bytes [chunk_offsets[k], chunk_offsets[k+1]) of rows_data (n_chunks + 1 non-decreasing entries); every chunk holds whole
 * rows.  col_types may contain TSQ_BYTES: a compact-bytes datum (flag 2: varint length + bytes, util/codec/bytes.go:141-160 — what
 * a varchar / blob column arrives as) becomes a var-len cell; such an output column needs offsets[cap_rows + 1] and a data buffer of
 * n_bytes bytes (a cell is a piece of the response).  All rows of all chunks are decoded: if they exceed cap_rows nothing is written
 * and the call returns TSQ_ERR_INVALID with *nrows_out = the rows needed.  Errors as for tsq_rows_decode, decided by the first
 * offending value in stream order (*nrows_out = the complete rows before it, already in out_cols); additionally TSQ_ERR_INVALID
 * "datum kind does not match the column type" (a bytes datum for a number column or the reverse).  A memcomparable bytes datum
 * (flag 1: groups of 8 bytes + a marker, util/codec/bytes.go:35-118 — the EncodeKey form index keys hold) becomes a var-len cell too;
 * its errors are DecodeBytes': "insufficient bytes to decode value" | "invalid marker byte" | "invalid padding byte". */
tsq_status tsq_rows_decode_chunks(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks,
                                  uint32_t data_flags, int32_t n_cols, const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows,
                                  int64_t* nrows_out);

/* ---------------------------------------------------------------- index scans: the pairs of an index range -> columns (SURVEY.md §8 f, rank 4)
 * Replaces mocktikv's indexScanExec loop (store/mockstore/mocktikv/executor.go:191-320) = tablecodec.DecodeIndexKV
 * (tablecodec/tablecodec.go:376-434) per pair + the DecodeOne of every cut value by the executors above it.  Key k = bytes
 * [key_offsets[k], key_offsets[k+1]) of `keys`:  't' | EncodeInt(tableID) | "_i" | EncodeInt(indexID)  (19 bytes, skipped like
 * CutIndexKeyNew does)  | n_index_cols datums in EncodeKey form (ints flag 3, uints 4, reals 5, strings flag 1 + memcomparable
 * groups, NULL 0; the value forms are accepted too, like DecodeOne)  | [the handle as an int datum — a non-unique index].
 * pk_status (tablecodec.PrimaryKeyStatus): 0 = no handle column; 1 / 2 = out column n_index_cols is the handle (TSQ_I64 / TSQ_U64):
 * taken from the key when bytes remain behind the index columns, otherwise from the pair's value (bytes [value_offsets[k],
 * value_offsets[k+1]) of `values`, 8 bytes big endian: DecodeIndexValueAsHandle, tablecodec.go:456-465).
 * col_types / out_cols: n_index_cols (+ 1) entries; out_cols sized for n_keys rows, a TSQ_BYTES column with offsets[n_keys + 1] and
 * a data buffer of n_bytes bytes.  data_flags: TSQ_COL_DEVICE when keys / key_offsets / values / value_offsets are in HBM.
 * Errors are decided by the first offending pair in key order (*nkeys_out = the pairs before it, already in out_cols):
 * TSQ_ERR_INVALID "invalid encoded key" (a key shorter than its prefix + columns) | the DecodeOne errors of tsq_rows_decode_chunks |
 * "no handle in index key or value". */
tsq_status tsq_indexkeys_decode(tsq_ctx* ctx, const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys,
                                const uint8_t* values, int64_t n_value_bytes, const int64_t* value_offsets, uint32_t data_flags,
                                int32_t n_index_cols, const int32_t* col_types, int32_t pk_status, tsq_col* out_cols, int64_t* nkeys_out);

/* ---------------------------------------------------------------- the chunk wire format (SURVEY.md §8 a/A "wire Codec", f rank 2)
 * Replaces chunk.Codec (util/chunk/codec.go:28-143) and chunk.Decoder (codec.go:233-353) on device chunks.  A wire chunk is its
 * columns one after the other, each  u32 length | u32 nullCount | [(length + 7) / 8 bitmap bytes, only when nullCount > 0] |
 * [(length + 1) int64 offsets, var-len only] | data  — little endian, no padding (codec.go:50-76).  This is the column-major response
 * format a coprocessor can answer with instead of datum rows (tsq_rows_decode*): decoding it is a copy.
 *
 * tsq_chunk_encode = Codec.Encode (codec.go:42-48): cols (host, or TSQ_COL_DEVICE) -> out (host, or device with out_flags =
 * TSQ_COL_DEVICE).  *bytes_out = the encoded length; cap_bytes = 0 only asks for it; a too small buffer -> TSQ_ERR_INVALID with
 * *bytes_out set.  A column whose bitmap has no zero bit in its first nrows bits travels without it (nullCount = 0).
 *
 * tsq_chunk_decode = Codec.DecodeToChunk (codec.go:88-93) followed by Decoder.Decode (codec.go:257-269, 298-353): rows
 * [first_row, first_row + max_rows) of the wire chunk (cut at its end) are APPENDED to out_cols behind the out_cols[c].length rows
 * they already hold — data copied, var-len offsets rebased onto offsets[length] (an empty destination gets offsets[0] = 0), bitmap
 * bits shifted to the destination's bit position with the bits beyond the last row cleared; a column without a wire bitmap appends
 * set bits.  first_row must be a multiple of 8 (the Decoder consumes its intermediate chunk in multiples of 8 rows, codec.go:259);
 * the buffers of out_cols must hold length + max_rows rows (tsq_chunk_decode_peek tells the data bytes of the var-len columns).
 * out_cols[c].length is updated, *nrows_out = rows appended, *bytes_consumed = the length of the wire chunk (n_cols columns; the
 * caller keeps the remainder, codec.go:93).  first_row = 0, max_rows >= length on empty out_cols is DecodeToChunk / ReuseIntermChk.
 * A HOST buffer is staged into HBM by every call (the whole wire chunk): a Decoder that takes its response window by window uploads it
 * once (tsq_dev_alloc + tsq_copy_h2d) and passes data_flags = TSQ_COL_DEVICE.
 * Errors: TSQ_ERR_INVALID when the buffer ends inside a column, the offsets of a var-len column are damaged, or the columns have
 * different lengths (the reference slices out of range and panics). */
tsq_status tsq_chunk_encode(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int64_t nrows, uint8_t* out, int64_t cap_bytes,
                            uint32_t out_flags, int64_t* bytes_out);
/* rows_total_out: Column.length of the wire chunk; nrows_out: rows the window holds; bytes_out[c]: data bytes of var-len column c in
 * that window (0 for fixed-width columns); any out pointer may be NULL */
tsq_status tsq_chunk_decode_peek(tsq_ctx* ctx, const uint8_t* buf, int64_t n_bytes, uint32_t data_flags, const int32_t* col_types,
                                 int32_t n_cols, int64_t first_row, int64_t max_rows, int64_t* rows_total_out, int64_t* nrows_out,
                                 int64_t* bytes_out, int64_t* bytes_consumed);
tsq_status tsq_chunk_decode(tsq_ctx* ctx, const uint8_t* buf, int64_t n_bytes, uint32_t data_flags, const int32_t* col_types,
                            int32_t n_cols, int64_t first_row, int64_t max_rows, tsq_col* out_cols, int64_t* nrows_out,
                            int64_t* bytes_consumed);

/* ---------------------------------------------------------------- stored rows (rowcodec v2) -> columns (SURVEY.md §8 f, rank 4)
 * Replaces the per-row loop around rowcodec.ChunkDecoder.DecodeToChunk (util/rowcodec/decoder.go:158-238; row.fromBytes /
 * findColID / getData, util/rowcodec/row.go:37-150) — equivalently the storage-side chain BytesDecoder.DecodeToBytes
 * (decoder.go:252-322, mocktikv tableScanExec, store/mockstore/mocktikv/executor.go:124-196) + readRowsData + DecodeOne —
 * a table scan's KV values, each one row in the new row format
 *     [128][flag: bit0 = large][numNotNull u16][numNull u16][colIDs: u8 | u32 each, not-null ids sorted then null ids sorted]
 *     [end offsets of the not-null values: u16 | u32 each][values: ints 1/2/4/8 little-endian bytes, reals 8 bytes memcomparable]
 * become chunk columns.  `values` (n_bytes bytes) holds the rows back to back, row r = bytes [offsets[r], offsets[r+1])
 * (offsets: nrows+1 non-decreasing entries, the last one <= n_bytes), handles[r] = the row's int64 handle from its key (may be NULL when no column is the handle).  data_flags:
 * TSQ_COL_DEVICE when values / offsets / handles are in HBM.  cols[c] describes output column c. */
#define TSQ_RC_HANDLE      1u /* the column is the handle column (col.ID == handleColID, decoder.go:165-168): value = handles[r]   */
#define TSQ_RC_HAS_DEFAULT 2u /* column absent from the row: def_bits instead of NULL (defDatum, decoder.go:186-194)              */
#define TSQ_RC_BIT         4u /* a TypeBit column (type TSQ_BYTES): stored as an unsigned int, the cell is its last byteSize bytes in
                                 big endian (decoder.go:229-231, types.NewBinaryLiteralFromUint); byteSize = (Flen + 7) / 8 in flags bits 8..11 */
#define TSQ_RC_BIT_SIZE(flags) (((flags) >> 8) & 15u)
typedef struct tsq_rowcodec_col {
    int64_t  col_id;    /* ColInfo.ID                                                                                          */
    int32_t  type;      /* TSQ_I64 (signed int types, year), TSQ_U64 (UnsignedFlag), TSQ_F32 (TypeFloat), TSQ_F64 (TypeDouble),
                           TSQ_BYTES (varchar / varstring / string / blobs: chk.AppendBytes of the value, decoder.go:226-228)     */
    uint32_t flags;     /* TSQ_RC_*                                                                                             */
    uint64_t def_bits;  /* default value as it is stored in the column (a float32 default in the low 4 bytes)                  */
    const uint8_t* def_bytes; /* TSQ_BYTES column with TSQ_RC_HAS_DEFAULT: the default string (host memory), def_len bytes       */
    int64_t  def_len;
} tsq_rowcodec_col;
/* out_cols: host or TSQ_COL_DEVICE buffers for nrows rows (data + null_bitmap); a TSQ_BYTES column: offsets[nrows + 1] and a data
 * buffer of n_bytes bytes (a cell is a piece of its row, so the column cannot be larger than `values`) — plus nrows * def_len bytes for
 * a column with a default string (every row may lack the column).  Errors are decided by the FIRST offending
 * row in scan order; *nrows_out then holds the rows before it (already in out_cols): TSQ_ERR_INVALID with tsq_last_error =
 * "invalid codec version" (row.go:54-56) | "insufficient bytes to decode value" (a real shorter than 8 bytes, codec
 * number.go:84-86) | "malformed row" (header / id / offset arrays or a value running past the row, an int value that is not
 * 1, 2, 4 or >= 8 bytes long: the reference panics with an index out of range there). */
tsq_status tsq_rowcodec_decode(tsq_ctx* ctx, const uint8_t* values, int64_t n_bytes, const int64_t* offsets,
                               const int64_t* handles, int64_t nrows, uint32_t data_flags, int32_t n_cols,
                               const tsq_rowcodec_col* cols, tsq_col* out_cols, int64_t* nrows_out);

/* ---------------------------------------------------------------- chunk rows -> coprocessor response bytes (SURVEY.md §8 f, rank 4)
 * The inverse of tsq_rows_decode, for the storage side of a pushed-down plan: replaces the per-datum codec.EncodeValue loop
 * that turns the output rows of the coprocessor's executors into RowsData (store/mockstore/mocktikv/aggregate.go:96-113 for
 * partial aggregates, util/rowcodec/decoder.go:252-322 for scanned rows; cop_handler_dag.go:414-425 concatenates the values of a
 * row).  Row r of the columns `cols` becomes its values' datums back to back — int64 -> varintFlag + zig-zag varint,
 * uint64 -> uvarintFlag + varint, float / double -> floatFlag + 8 memcomparable bytes of the double, a var-len (TSQ_BYTES) cell ->
 * compactBytesFlag + varint(length) + the bytes (codec.go:101-109, bytes.go:141-148), NULL -> NilFlag
 * (util/codec/codec.go:74-99 with comparable = false) — and rows follow each other in `out` (host, or TSQ_COL_DEVICE in
 * out_flags).  col_flags[c] & TSQ_ENC_COMPARABLE selects the EncodeKey form of an integer column (intFlag / uintFlag + 8
 * big-endian bytes): what BytesDecoder.DecodeToBytes emits for the handle column (decoder.go:263-273).  col_flags may be NULL.
 * row_offsets (NULL or nrows + 1 entries, same residency as out): row r = out[row_offsets[r], row_offsets[r + 1]) — the response
 * is cut into tipb.Chunks of 64 rows (cop_handler_dag.go:510-519).  *bytes_out = length of the byte string; when it exceeds
 * cap_bytes nothing is written and the call returns TSQ_ERR_INVALID with *bytes_out = the bytes needed (a caller that cannot bound
 * the size — string columns — asks with cap_bytes = 0 first).  TSQ_ENC_COMPARABLE on a var-len column writes the memcomparable
 * form index keys hold: bytesFlag + groups of 8 bytes, each followed by its marker 0xFF - pad count (codec.go:86-91,
 * bytes.go:35-67).  256 consecutive rows must encode to < 4 GiB. */
#define TSQ_ENC_COMPARABLE 1u
tsq_status tsq_rows_encode(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, const uint32_t* col_flags, int64_t nrows,
                           uint8_t* out, int64_t cap_bytes, uint32_t out_flags, int64_t* row_offsets, int64_t* bytes_out);

/* ---------------------------------------------------------------- record keys of a table scan <-> handles (SURVEY.md §8 f, rank 4)
 * Replaces tablecodec.DecodeRowKey (tablecodec/tablecodec.go:235-242; DecodeRecordKey :73-77 hands the table id back as well),
 * called once per scanned KV pair by mocktikv's tableScanExec (store/mockstore/mocktikv/executor.go:124-196) before
 * rd.DecodeToBytes — handles_out[] is the `handles` argument of tsq_rowcodec_decode — and EncodeRowKeyWithHandle (:65-70).
 *     record key = 't' | EncodeInt(tableID) | "_r" | EncodeInt(handle)           RecordRowKeyLen = 19 bytes
 * `keys` holds the keys back to back: key r = bytes [key_offsets[r], key_offsets[r+1]) (n_keys+1 non-decreasing entries), or
 * bytes [19 r, 19 r + 19) when key_offsets is NULL (then n_bytes = 19 * n_keys).  data_flags: TSQ_COL_DEVICE when keys /
 * key_offsets / the outputs are in HBM.  table_ids_out may be NULL.  The FIRST key in scan order that is not a record key
 * (length != 19, no 't' prefix, no "_r" separator) decides: TSQ_ERR_INVALID with tsq_last_error = "invalid key"
 * (errInvalidKey, tablecodec.go:237), *nkeys_out = the keys before it (their handles are out). */
tsq_status tsq_rowkeys_decode(tsq_ctx* ctx, const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys,
                              uint32_t data_flags, int64_t* handles_out, int64_t* table_ids_out, int64_t* nkeys_out);
/* keys_out: 19 * n_keys bytes, key r = EncodeRowKeyWithHandle(table_id, handles[r]) */
tsq_status tsq_rowkeys_encode(tsq_ctx* ctx, int64_t table_id, const int64_t* handles, int64_t n_keys, uint32_t data_flags,
                              uint8_t* keys_out);

/* ---------------------------------------------------------------- ORDER BY / TopN (SURVEY.md §8 f, rank 3)
 * Replaces SortExec (executor/sort.go:27-144) and TopNExec (sort.go:146-318): all child rows are pushed, tsq_sort_finish
 * orders them by the ByItems — bare columns (sort.go:107-113), each ascending or descending, compared like
 * chunk.GetCompareFunc's comparators (util/chunk/compare.go:27-103: NULL smaller than every value, unsigned / signed /
 * float32-widened / float64 order) — and tsq_sort_pull hands them out in that order.  limit_count >= 0 makes it a
 * TopN: only rows [limit_offset, limit_offset + limit_count) of the order are returned (sort.go:213-238).  Rows whose
 * keys all compare equal come out in input order (the reference's sort.Slice / heap leave that order unspecified). */
typedef struct tsq_sort_cfg {
    int32_t n_cols;
    int32_t col_types[TSQ_MAX_COLS];
    int32_t n_keys;                       /* ByItems */
    int32_t key_col[TSQ_MAX_KEYS];
    int32_t key_desc[TSQ_MAX_KEYS];       /* ByItems[i].Desc */
    int64_t limit_offset;                 /* PhysicalLimit.Offset (0 for a plain sort) */
    int64_t limit_count;                  /* PhysicalLimit.Count; < 0 = no limit (SortExec) */
    int32_t max_chunk_size;
    int32_t reserved;
} tsq_sort_cfg;
typedef struct tsq_sort tsq_sort;
tsq_status tsq_sort_create(tsq_ctx* ctx, const tsq_sort_cfg* cfg, tsq_sort** out);
tsq_status tsq_sort_push(tsq_sort* s, const tsq_col* cols, int32_t n_cols, int64_t nrows);   /* fetchRowChunks, sort.go:80-97 */
tsq_status tsq_sort_finish(tsq_sort* s);
/* out_cols: the input schema; host or TSQ_COL_DEVICE buffers (data + null_bitmap) for cap_rows rows */
tsq_status tsq_sort_pull(tsq_sort* s, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows, int64_t* nrows_out, int32_t* eos);
/* Var-len (TSQ_BYTES) columns travel through the sort as payload (Chunk.AppendRow of a var-len cell, util/chunk/chunk.go:334-356)
 * and may be ORDER BY items: chunk.GetCompareFunc's cmpString (util/chunk/compare.go:71-77: bytes, then the shorter string first;
 * binary collation, the only one TinySQL has).  A string item costs one radix pass per byte position that differs between rows
 * (plus the length); the TopN radix select takes a string first item by its first eight bytes.  A pull fills out_cols[c].offsets (cap_rows + 1 entries) and .data of a var-len column; tsq_sort_peek (after
 * tsq_sort_finish) tells how many rows the next pull of up to cap_rows rows delivers and how many data bytes each var-len column
 * needs (bytes_out[c]; 0 for fixed-width columns). */
tsq_status tsq_sort_peek(tsq_sort* s, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols);
/* rows = rows that went through the radix passes: all of them for a sort; for a TopN whose Offset + Count is small, only the
 * candidates kept by the radix select on the first ORDER BY item (ties at the threshold included) */
tsq_status tsq_sort_stats(tsq_sort* s, int64_t* rows, int32_t* passes, int32_t* passes_skipped, double* sort_kernel_ms);
tsq_status tsq_sort_cancel(tsq_sort* s);
void       tsq_sort_destroy(tsq_sort* s);

/* ---------------------------------------------------------------- multi-GPU radix redistribute
 * Splits rows by rank(key) = ((mix64(key) & 0xffff) * n_parts) >> 16 into n_parts contiguous
 * runs (CPU analogue: aggregate.go:352-356 shuffle / join.go:219 dispatch).  The exchange
 * itself (RCCL all-to-all over xGMI) is done by the caller on the returned device buffers.
 * key_mode 0 ranks by join-key equality (util/codec/codec.go:212-240), 1 by group-key equality
 * (codec.go:713-746: -0.0 and +0.0 share a group).  Rows with a NULL key are routed to part 0.
 * cols/out_cols are device resident; out_cols need capacity nrows; counts_out: int64[n_parts]
 * on the host; run p of every output column is rows [sum(counts[0..p)), +counts[p]).
 * Var-len (TSQ_BYTES) columns travel with their rows — as payload, or as the key (equal strings rank alike: a hash of the bytes):
 * their out column needs offsets[nrows + 1] and as many data bytes as the input column holds; the cells are laid out in the order
 * of the split rows. */
tsq_status tsq_radix_split(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int32_t key_col,
                           int32_t key_mode, int64_t nrows, int32_t n_parts, tsq_col* out_cols,
                           int64_t* counts_out);

/* ---------------------------------------------------------------- multi-GPU exchange (RCCL over xGMI), one process per GPU
 * What the reference does with goroutines and channels inside one process — HashJoinExec handing outer chunks to its join
 * workers (executor/join.go:160-231), HashAggExec shuffling partial results to its final workers by group key
 * (executor/aggregate.go:352-356) — across the GPUs of a node: the rank of a key (((mix64(key word) & 0xffff) * world) >> 16, as in tsq_radix_split) owns it.
 * Every rank runs the same call sequence.  Bootstrap: rank 0 calls tsq_comm_unique_id and hands the 128 bytes to the other
 * ranks by any side channel (the Go host: the coordinator's RPC; the harness: a file), then every rank calls tsq_comm_create.
 *
 * tsq_redistribute splits `cols` (device resident; var-len columns included, also as the key) by rank(key) on the
 * context's stream (tsq_radix_split), exchanges the run sizes (rows, and bytes of every var-len column), and queues ONE group of RCCL
 * sends / receives on the communicator's own stream.  A var-len column's run travels as its slice of the offsets plus its bytes; the
 * received column gets offsets[n + 1] rebased onto its own data (out_cols[c].offsets, owned by the slot).  A column with a null bitmap
 * on ANY rank travels with one NOT-NULL byte per row and arrives with a packed bitmap (out_cols[c].null_bitmap, owned by the
 * slot); rows with a NULL key go to rank 0 (they never join; GROUP BY makes them one group).  out_cols describe device
 * buffers owned by `slot` (0..7) of the communicator: they hold the received rows once tsq_redistribute_wait(comm, slot) has
 * made the context's stream wait for the exchange, and stay valid until the next tsq_redistribute on the same slot.  Queue
 * piece c + 1's redistribute before piece c's consumer (wait; tsq_join_probe_push / tsq_agg_push) and the wire time of
 * c + 1 hides behind the operator kernels of c.
 * The all-reduces take up to 8 host words (op 0 sum, 1 max, 2 min) and synchronise both streams: they are the barrier + the
 * COUNT(*) / timing reductions a distributed plan needs. */
#define TSQ_COMM_ID_BYTES 128
tsq_status tsq_comm_unique_id(uint8_t* id_out /* [TSQ_COMM_ID_BYTES] */);
tsq_status tsq_comm_create(tsq_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id, tsq_comm** out);
void       tsq_comm_destroy(tsq_comm* c);
tsq_status tsq_comm_allreduce_i64(tsq_comm* c, int64_t* inout, int32_t n, int32_t op);
tsq_status tsq_comm_allreduce_f64(tsq_comm* c, double* inout, int32_t n, int32_t op);
tsq_status tsq_comm_barrier(tsq_comm* c);
/* ABI 7: the communicator as the collective library sees it — ncclCommUserRank, ncclCommCount, ncclGetVersion (-1: the loaded librccl
 * lacks the call).  bench.py prints them next to a multi-GPU number. */
tsq_status tsq_comm_info(tsq_comm* c, int32_t* rank_out, int32_t* nranks_out, int32_t* rccl_version_out);
/* key_mode of tsq_redistribute: 0 / 1 as for tsq_radix_split; TSQ_KEYMODE_BROADCAST = an ALL-GATHER of the columns — every rank
 * receives every rank's rows, in rank order (key_col is ignored): the small side of a broadcast join (a filtered dimension table,
 * the result of an earlier join) goes to every GPU once, and the big side is never moved (tinysql_amd/parallel.py: dist_q3). */
#define TSQ_KEYMODE_JOIN      0
#define TSQ_KEYMODE_GROUP     1
#define TSQ_KEYMODE_BROADCAST 2
tsq_status tsq_redistribute(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode,
                            int64_t nrows, int32_t slot, tsq_col* out_cols, int64_t* nrows_out);
tsq_status tsq_redistribute_wait(tsq_comm* c, int32_t slot);
/* tsq_redistribute in three calls, for a plan that redistributes its input in PIECES (piece c -> slot c): prepare every piece
 * (the split, on the context's stream), exchange the run sizes of ALL prepared slots with one all-gather (the only call of the
 * three that waits for the other ranks on the host), then issue the pieces one after the other — each issue only queues the
 * sends / receives of its slot behind the previous one.  tsq_redistribute(slot) = prepare(slot) + counts({slot}) + issue(slot).
 * Every rank lists the same slots in the same order. */
tsq_status tsq_redistribute_prepare(tsq_comm* c, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode,
                                    int64_t nrows, int32_t slot);
tsq_status tsq_redistribute_counts(tsq_comm* c, const int32_t* slots, int32_t n_slots);
tsq_status tsq_redistribute_issue(tsq_comm* c, int32_t slot, tsq_col* out_cols, int32_t n_cols, int64_t* nrows_out);

/* ---------------------------------------------------------------- statistics (roofline reporting) */
typedef struct tsq_stats {
    int64_t build_rows;
    int64_t build_rows_inserted;   /* non-NULL keys */
    int64_t probe_rows;
    int64_t out_rows;
    int64_t table_bytes;
    int64_t table_buckets;
    double  build_kernel_ms;       /* HIP-event time of the last build / probe kernels */
    double  probe_kernel_ms;
    int64_t h2d_bytes;
    int64_t d2h_bytes;
    int64_t kernel_launches;
    double  partition_kernel_ms;   /* last radix batch: partition kernel (0 when the direct probe ran) */
    double  radix_probe_kernel_ms; /* last radix batch: partition-at-a-time probe (+ overflow list) kernels */
    double  partition_kernel_ms_sum;    /* HIP-event sums over the most recent radix_timed_batches (<= 32) batches */
    double  radix_probe_kernel_ms_sum;
    int64_t radix_timed_batches;
    int64_t radix_batches;         /* join: probe batches through the radix path; agg: batches pre-aggregated in LDS */
    int64_t radix_overflow_rows;   /* rows that did not fit their partition region (skew) in the last batch */
    int32_t radix_bits;            /* log2(partitions) of the last radix batch */
    int32_t build_partitioned;     /* join: 1: the table was assembled slice by slice in LDS (tsq_buildpart.h), 0: row-at-a-time CAS build;
                                      aggregate: 2: several integer key columns composed into one 64-bit key for a child aggregate, 3: the group
                                      keys (strings / wide key sets) went through the dictionary of key records (tsq_keydict.h) to a child
                                      aggregate by group id; build_handed_back_rows then counts the exception rows this operator kept; 4 (ABI 7): about
                                      as many groups as rows — the group table was a set of partitioned, LDS-sized sub-tables (csrc/tsq_aggfast.h K7p);
                                      when the composite-key child of (2) took that mode, dense_flushes reads -4 */
    int64_t build_handed_back_rows; /* partitioned build: rows inserted row by row afterwards (skewed pass-1 / pass-2 regions);
                                       aggregate: rows of a multi-key GROUP BY whose 64-bit tag belonged to another key (resolved) */
    int32_t table_slice_bits;      /* join: log2(slices) of the join table (0: one slice); aggregate on the packed route: bits of a travelling argument cell (16 / 32: narrow cells, 64) */
    int32_t build_slice_retries;   /* 1: a slice overflowed (skewed keys) and the table was rebuilt as one slice */
    int32_t probe_route;           /* route of the last probe batch: TSQ_ROUTE_* */
    int32_t packed_key_bits;       /* TSQ_ROUTE_PACKED: bits of the build side's key range (0 otherwise) */
    double  packed_build_ms;       /* kernels that made the packed-key images (once per build side) */
    int64_t heap_bytes;            /* aggregate: bytes of var-len input cells the operator holds (largest column heap) */
    int64_t heap_compactions;      /* aggregate: times a string heap was compacted to the strings the groups refer to */
    int32_t shared_build;          /* join: 1 = tsq_join_build_finish_shared replicated the packed images (probe rows never cross xGMI) */
    int32_t dense_flushes;         /* aggregate: times the dense partial state of the one-key packed route became groups of the table (1 = at finish only; 0 = route not taken) */
    int64_t shared_image_bytes;    /* ... bytes of the images every rank all-reduced, once per build side */
    double  shared_allreduce_ms;   /* ... wall time of that all-reduce on this rank */
    int64_t div_by_zero_warnings;  /* join: division-by-zero warnings the OtherConditions / outer filters raised so far (NULL result + warning,
                                      expression/errors.go:65-77): the shim appends that many ErrDivisionByZero warnings to the statement context
                                      (handleDivisionByZeroError via executor/joiner.go:155-167) */
    int32_t packed_lds_bits;       /* join, ABI 7: log2 of the FINAL partitions of the last materialising packed batch whose build side sat in LDS
                                      (csrc/tsq_damat.h: two partition levels, ranked payload tables); 0: the batch took another variant */
    int32_t keyrec_digests;        /* join, ABI 7: 1 when the key-record route keeps string cells that do not fit a record as (length, 64-bit digest)
                                      and compares the bytes of every candidate match (long string keys), 0 otherwise */
    int32_t side_stream_batches;   /* aggregate, ABI 7: batches of the packed route whose rows went from the partitioned store into the dense state on the
                                      operator's side stream, beside the partition pass of the next batch (TSQ_KNOB_AGG_OVERLAP) */
    int32_t reserved0;
} tsq_stats;
#define TSQ_ROUTE_DIRECT     0   /* k_probe_count / k_probe_emit on the table in HBM */
#define TSQ_ROUTE_RADIX_L2   1   /* radix partition, table slices through the XCD's L2 */
#define TSQ_ROUTE_RADIX_LDS  2   /* radix partition, LDS copies of the table slices (64-bit table words) */
#define TSQ_ROUTE_PACKED     3   /* packed keys: 2-byte entries against direct-address images in LDS */
#define TSQ_ROUTE_KEYREC     4   /* key records (ABI 6): several key columns / string keys as 32-byte records, hash-partitioned, matched in LDS (csrc/tsq_keyrec.h) */
tsq_status tsq_join_stats(tsq_join* j, tsq_stats* out);
tsq_status tsq_agg_stats(tsq_agg* a, tsq_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* TSQ_H */
