/*
 * oracle.h — CPU restatement of the TinySQL reference algorithms for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tinysql_amd/, libtsq) may include,
 * link, call or execute anything in this directory.  Allowed users: tests/, bench.py's
 * `cpu_baseline` leg, __graft_entry__.smoke().
 *
 * The reference cannot be built here (no Go toolchain; HashJoinExec/HashAggExec bodies are
 * course stubs), so this is a line-traceable restatement; every function cites
 * the reference file:line it follows.  It is pinned against the reference's own golden
 * vectors in tests/test_oracle_golden.py.  Parity status: pinned for all
 * integer/COUNT paths; SUM/AVG(double) pinned on the reference's tiny exact cases;
 * FIRST_ROW on non-key columns unpinned (nondeterministic in the reference itself).
 *
 * POD types (tsq_col, tsq_join_cfg, tsq_agg_cfg, tsq_expr_prog, opcodes) are shared with
 * include/tsq.h so both sides are driven by byte-identical inputs.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/tsq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_result orc_result; /* a materialised set of output columns */

int64_t orc_result_rows(const orc_result* r);
int32_t orc_result_cols(const orc_result* r);
/* copies column c: data (rows*8 bytes; F32 columns are returned as 4-byte) and a per-row
 * NOT-NULL byte array (1 = not null).  Either pointer may be NULL. */
void    orc_result_copy_col(const orc_result* r, int32_t c, void* data, uint8_t* notnull);
int32_t orc_result_col_type(const orc_result* r, int32_t c);
int64_t orc_result_col_bytes(const orc_result* r, int32_t c);   /* TSQ_BYTES column: data bytes */
void    orc_result_copy_varlen(const orc_result* r, int32_t c, int64_t* offsets, void* data, uint8_t* notnull);
void    orc_result_free(orc_result* r);
const char* orc_last_error(void);

/* hash/fnv New64 (FNV-1) over [flag][bytes] per key column — util/codec/codec.go:249-338,
 * executor/hash_table.go:47-72.  out_hash[nrows], out_has_null[nrows]. */
void orc_hash_keys(const tsq_col* cols, const int32_t* key_idx, int32_t n_keys, int64_t nrows,
                   const uint8_t* selected, uint64_t* out_hash, uint8_t* out_has_null);
/* FNV-1 64 of a raw byte string (KATs). */
uint64_t orc_fnv1_64(const uint8_t* p, int64_t n);

/* util/codec/codec.go:713-746 HashGroupKey for one row: appends the encoded key of column
 * `col` row `row` to buf, returns bytes written. */
int32_t orc_group_key_encode(const tsq_col* col, int64_t row, uint8_t* buf);

/* synthetic tables — same generator as tsq_gen_column */
void orc_gen_column(const tsq_gen_spec* spec, int64_t nrows, void* dst, uint8_t* null_bitmap,
                    const void* src);

/* HashJoinExec restated: executor/join.go:125-146,290-362, hash_table.go:110-169,181-276,
 * joiner.go:145-167,220-410; stubs filled per courses/proj5-part2-README-zh_CN.md.
 * Output: left-child cols || right-child cols.  Returns NULL on error (orc_last_error). */
orc_result* orc_hash_join(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build,
                          const tsq_col* probe_cols, int64_t n_probe, const uint8_t* selected,
                          tsq_status* status);

/* Timed CPU baseline: same algorithm, single-threaded build + `threads` probe workers,
 * row-at-a-time append into per-worker 1024-row result chunks which are then dropped.
 * Returns the number of joined rows; fills build_ms/probe_ms. */
int64_t orc_hash_join_timed_multi(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build, const tsq_col* probe_cols, int64_t n_probe,
                                  const int32_t* thread_counts, int32_t n_runs, double* build_ms, double* probe_ms /* [n_runs] */, uint64_t* sum_out,
                                  uint64_t* xor_out);
int64_t orc_hash_join_timed(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build,
                            const tsq_col* probe_cols, int64_t n_probe, int32_t threads,
                            double* build_ms, double* probe_ms,
                            uint64_t* sum_out, uint64_t* xor_out);

/* order-independent checksum of a result set: per row h = rowhash(values, null flags);
 * returns sum and xor over rows — the same function the GPU fuses into its probe. */
void orc_rows_checksum(const tsq_col* cols, int32_t n_cols, int64_t nrows,
                       uint64_t* sum_out, uint64_t* xor_out);

/* HashAggExec restated: executor/aggregate.go:307-350,359-410,429-457,559-588 with the two
 * stubs (shuffleIntermData :354, consumeIntermData :424) filled per
 * courses/proj5-part3-README-zh_CN.md; aggfuncs/func_{count,sum,avg,max_min,first_row}.go.
 * partial_workers/final_workers emulate M partial + N final workers deterministically
 * (chunk c goes to partial worker c % M; group goes to final worker fnv(groupkey) % N). */
orc_result* orc_hash_agg(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows,
                         int32_t partial_workers, int32_t final_workers, tsq_status* status);
/* timed multi-threaded variant (threads partial workers + threads final workers) */
orc_result* orc_hash_agg_timed(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows,
                               int32_t threads, double* ms, tsq_status* status);

/* expression.VecEval (expression.go:329) over builtin_*_vec.go signatures, node at a time.
 * sel: optional chunk.sel.  Output: out_data (8 bytes/row), out_notnull (1 byte/row). */
tsq_status orc_expr_eval(const tsq_expr_prog* prog, const tsq_col* cols, int32_t n_cols,
                         int64_t nrows, const int32_t* sel, void* out_data, uint8_t* out_notnull,
                         int64_t* div_by_zero_warnings);
/* expression.VecEvalString (expression.go:329-341) for a STRING-valued root: builtinIfStringSig / builtinIfNullStringSig
 * .vecEvalString (builtin_control_vec_generated.go:209, :81: result.ReserveString(n), then AppendNull or AppendString per row),
 * Column.VecEvalString (column.go:111-130: CopyReconstruct through sel), Constant.VecEvalString (constant.go:86).  Output = the
 * state of the result column: out_offsets[nrows + 1], its data bytes (at most cap_bytes are written; *bytes_out = all of them),
 * out_notnull (1 byte/row). */
tsq_status orc_expr_eval_str(const tsq_expr_prog* prog, const tsq_col* cols, int32_t n_cols, int64_t nrows, const int32_t* sel,
                             int64_t* out_offsets, uint8_t* out_data, int64_t cap_bytes, uint8_t* out_notnull, int64_t* bytes_out,
                             int64_t* div_by_zero_warnings);
/* expression.VecEvalBool / VectorizedFilter (expression.go:205-279, chunk_executor.go:196) */
tsq_status orc_filter_eval(const tsq_expr_prog* progs, int32_t n_progs, const tsq_col* cols,
                           int32_t n_cols, int64_t nrows, const int32_t* sel,
                           uint8_t* selected_out, uint8_t* isnull_out,
                           int64_t* div_by_zero_warnings);

/* rowHashMap unit (executor/hash_table.go:181-276): put n (key,ptr) pairs then Get(key):
 * writes matching ptrs in the order Get returns them; returns the count. */
int64_t orc_rowhashmap_put_get(const uint64_t* keys, const uint64_t* ptrs, int64_t n,
                               uint64_t probe_key, uint64_t* out_ptrs, int64_t cap);

/* MergeJoinExec (executor/merge_join.go:31-373) over sorted children: outer rows in order, each with its inner group in
 * order; `build` = inner table, `probe` = outer table of cfg.  OtherConditions -> TSQ_ERR_UNSUPPORTED. */
orc_result* orc_merge_join(const tsq_join_cfg* cfg, const tsq_col* inner_cols, int64_t n_inner, const tsq_col* outer_cols,
                           int64_t n_outer, tsq_status* status);

/* ---- coprocessor-response row codec (codec_rows.cpp; SURVEY.md §8 f rank 2) */
int32_t orc_value_size_signed(int64_t v);    /* valueSizeOfSignedInt, util/codec/codec.go:156-165 */
int32_t orc_value_size_unsigned(uint64_t v); /* valueSizeOfUnsignedInt, codec.go:178-187 */
/* encode (codec.go:74-99): EncodeValue (comparable = 0) / EncodeKey (1) of rows of fixed-width columns */
int64_t orc_encode_rows(const tsq_col* cols, int32_t n_cols, int64_t nrows, int32_t comparable, uint8_t* out, int64_t cap);
/* readRowsData (distsql/select_result.go:139-155) + Decoder.DecodeOne (codec.go:623-690) */
int32_t orc_decode_rows(const uint8_t* data, int64_t n_bytes, int32_t n_cols, const int32_t* types, int64_t cap_rows, void** out_data,
                        uint8_t** out_notnull, int64_t* nrows_out, int64_t* consumed);

/* the chunks of a response one after the other, bytes datums included (var-len columns) */
orc_result* orc_decode_rows_chunks(const uint8_t* data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks, int32_t n_cols,
                                   const int32_t* types, int32_t* status);

/* ---- stored rows, rowcodec v2 (rowcodec.cpp; SURVEY.md §8 f rank 4) */
/* Encoder.Encode (util/rowcodec/encoder.go:34-194) of every row of a fixed-width chunk (+ an optional KindBytes pad column) */
int64_t orc_rowcodec_encode(const tsq_col* cols, const int64_t* col_ids, int32_t n_cols, int64_t nrows, int64_t pad_col_id, const int64_t* pad_len,
                            uint8_t* out, int64_t cap, int64_t* offsets_out);
/* the scan loop around ChunkDecoder.DecodeToChunk (util/rowcodec/decoder.go:158-238) */
int32_t orc_rowcodec_decode(const uint8_t* values, const int64_t* offsets, const int64_t* handles, int64_t nrows, const tsq_rowcodec_col* cols,
                            int32_t n_cols, void** out_data, uint8_t** out_notnull, int64_t* nrows_out);
/* the same loop with var-len (TSQ_BYTES) columns, into a materialised result: the rows before the offending one */
orc_result* orc_rowcodec_decode_chunk(const uint8_t* values, const int64_t* offsets, const int64_t* handles, int64_t nrows, const tsq_rowcodec_col* cols,
                                      int32_t n_cols, int32_t* status);
/* BytesDecoder.DecodeToBytes (decoder.go:252-322) of one row, values concatenated in column order */
int64_t orc_rowcodec_to_old_bytes(const uint8_t* row_data, int64_t len, int64_t handle, const tsq_rowcodec_col* cols, int32_t n_cols, uint8_t* out,
                                  int64_t cap);
/* row.ColumnIsNull (util/rowcodec/row.go:152-165) */
int32_t orc_rowcodec_column_is_null(const uint8_t* row_data, int64_t len, int64_t col_id, int32_t has_default);

/* ---- the chunk wire format: chunk.Codec / chunk.Decoder (chunk_wire.cpp; SURVEY.md §8 a/A) — a chunk object holds every Column
 * as the reference does (length, nullBitmap, offsets, data as byte slices) */
typedef struct orc_wire_chunk orc_wire_chunk;
orc_wire_chunk* orc_wire_new(const int32_t* elem, int32_t n_cols);                                  /* chunk.New; elem = getFixedLen: 4, 8, -1 */
orc_wire_chunk* orc_wire_from_cols(const tsq_col* cols, int32_t n_cols, int64_t nrows);
void    orc_wire_free(orc_wire_chunk* k);
int64_t orc_wire_encode(const orc_wire_chunk* k, uint8_t* out, int64_t cap);                       /* Codec.Encode, codec.go:42-76 */
int64_t orc_wire_decode_to_chunk(orc_wire_chunk* k, const uint8_t* buffer, int64_t n);            /* Codec.DecodeToChunk, codec.go:88-143; -1: out of range */
int64_t orc_wire_decoder_reset(orc_wire_chunk* interm, const uint8_t* data, int64_t n);           /* Decoder.Reset, codec.go:272-275 */
int64_t orc_wire_decoder_remained(const orc_wire_chunk* interm);
int64_t orc_wire_decoder_decode(orc_wire_chunk* interm, orc_wire_chunk* chk, int64_t required);   /* Decoder.Decode, codec.go:257-269, 298-353 */
void    orc_wire_decoder_reuse(orc_wire_chunk* interm, orc_wire_chunk* chk);                      /* Decoder.ReuseIntermChk, codec.go:291-308 */
int64_t orc_wire_col_length(const orc_wire_chunk* k, int32_t c);
int64_t orc_wire_col_bitmap(const orc_wire_chunk* k, int32_t c, uint8_t* out, int64_t cap);
int64_t orc_wire_col_offsets(const orc_wire_chunk* k, int32_t c, int64_t* out, int64_t cap);
int64_t orc_wire_col_data(const orc_wire_chunk* k, int32_t c, uint8_t* out, int64_t cap);

/* ---- index keys: EncodeIndexSeekKey + EncodeKey, and indexScanExec's DecodeIndexKV (codec_rows.cpp; SURVEY.md §8 f rank 4) */
int64_t orc_encode_index_keys(const tsq_col* cols, int32_t n_cols, int64_t nrows, int64_t table_id, int64_t index_id, const int64_t* handles,
                              const uint8_t* handle_in_key, uint8_t* out, int64_t cap, int64_t* key_offsets_out);
orc_result* orc_decode_index_kv(const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys, const uint8_t* values,
                                const int64_t* value_offsets, int32_t n_index_cols, const int32_t* types, int32_t pk_status, int32_t* status);

/* ---- SortExec / TopNExec row order (sort_rows.cpp; SURVEY.md §8 f rank 3) */
int32_t orc_row_compare(const tsq_col* cols, const int32_t* key_col, const int32_t* key_desc, int32_t n_keys, int64_t i, int64_t j);
void    orc_sort_rows(const tsq_col* cols, int64_t nrows, const int32_t* key_col, const int32_t* key_desc, int32_t n_keys, int64_t* perm_out);

/* ---- tablecodec record keys + the storage side's datum-level aggregate (mocktikv.cpp; SURVEY.md §8 f rank 4) */
void    orc_encode_row_key(int64_t table_id, int64_t handle, uint8_t* out19);                       /* EncodeRowKeyWithHandle, tablecodec.go:65-70 */
int32_t orc_decode_row_key(const uint8_t* key, int64_t len, int64_t* handle);                       /* DecodeRowKey, :235-242 */
int32_t orc_decode_key_head(const uint8_t* key, int64_t len, int64_t* table_id, int64_t* index_id, int32_t* is_record);  /* DecodeKeyHead, :188-220 */
int32_t orc_decode_record_key(const uint8_t* key, int64_t len, int64_t* table_id, int64_t* handle); /* DecodeRecordKey, :73-77 (stub filled) */
/* hashAggExec (store/mockstore/mocktikv/aggregate.go:78-182) with expression/aggregation's functions, row at a time in scan order */
orc_result* orc_cop_hash_agg(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows, tsq_status* status);

#ifdef __cplusplus
}
#endif
#endif
