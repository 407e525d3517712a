// tsq_tablecodec_dp.h — the scalar core of tsq_rowkeys_decode / tsq_rowkeys_encode (tsq_tablecodec.hip): one record key of a
// table scan <-> (table id, handle).  TSQ_HD and templated on the byte reader, so the kernel runs it on a tile staged in LDS and the
// CPU test-suite runs the very same code through tests/hostsim against the restatement pinned on tablecodec_test.go.
// Reference: tablecodec.EncodeRowKeyWithHandle / appendTableRecordPrefix (tablecodec/tablecodec.go:57-70), DecodeRowKey (:235-242),
// DecodeRecordKey (:73-77, a course stub: the layout below is what EncodeRowKeyWithHandle writes and TestRecordKey reads back),
// hasTablePrefix / hasRecordPrefixSep (:157-163); codec.EncodeInt / DecodeInt = 8 big-endian bytes of v ^ signMask
// (util/codec/number.go:24-53).
//
//   record key = 't' | EncodeInt(tableID) (8 B) | "_r" | EncodeInt(handle) (8 B)          RecordRowKeyLen = 19
#ifndef TSQ_TABLECODEC_DP_H
#define TSQ_TABLECODEC_DP_H

#include "tsq_device.h"

#define TSQ_TC_ROWKEY_LEN 19u
enum { TC_OK = 0, TC_INVALID_KEY = 1 };

// 8 big-endian bytes <-> the host-order number (EncodeInt / DecodeInt without the sign flip)
TSQ_HD uint64_t tsq_tc_bswap64(uint64_t u) {
    return ((uint64_t)__builtin_bswap32((uint32_t)u) << 32) | (uint64_t)__builtin_bswap32((uint32_t)(u >> 32));
}

// DecodeRowKey (tablecodec.go:235-242) + the table id DecodeRecordKey / DecodeKeyHead hand back.  R::le(p, 8) = the 8 bytes at p
// as a little-endian number (the reader of tsq_rowcodec_dp.h: a staged tile or plain memory); `len` is the key's length.
template <class R>
TSQ_HD int tsq_tc_decode_row_key(const R& b, uint32_t len, int64_t* table_id, int64_t* handle) {
    *table_id = 0;
    *handle = 0;
    if (len != TSQ_TC_ROWKEY_LEN) return TC_INVALID_KEY;               // len(key) != RecordRowKeyLen
    const uint64_t w0 = b.le(0, 8), w1 = b.le(8, 8), w2 = b.le(11, 8);  // bytes 0..7, 8..15, 11..18
    if ((uint8_t)w0 != (uint8_t)'t') return TC_INVALID_KEY;           // hasTablePrefix
    if ((uint8_t)(w1 >> 8) != (uint8_t)'_' || (uint8_t)(w1 >> 16) != (uint8_t)'r') return TC_INVALID_KEY;  // hasRecordPrefixSep(key[prefixLen-2:])
    // table id = bytes 1..8 big endian: seven of them in w0, the eighth is byte 8 = the low byte of w1
    const uint64_t tid_le = (w0 >> 8) | (w1 << 56);
    *table_id = (int64_t)(tsq_tc_bswap64(tid_le) ^ 0x8000000000000000ULL);
    *handle = (int64_t)(tsq_tc_bswap64(w2) ^ 0x8000000000000000ULL);   // DecodeCmpUintToInt(binary.BigEndian.Uint64(key[prefixLen:]))
    return TC_OK;
}

// EncodeRowKeyWithHandle (tablecodec.go:65-70): the 19 bytes as three little-endian pieces — bytes 0..7, 8..15, 16..18
TSQ_HD void tsq_tc_encode_row_key(int64_t table_id, int64_t handle, uint64_t* p0, uint64_t* p1, uint32_t* p2) {
    const uint64_t t = tsq_tc_bswap64((uint64_t)table_id ^ 0x8000000000000000ULL);  // byte k of EncodeInt = t >> 8k
    const uint64_t h = tsq_tc_bswap64((uint64_t)handle ^ 0x8000000000000000ULL);
    *p0 = (uint64_t)(uint8_t)'t' | (t << 8);                                                   // 't', table id bytes 0..6
    *p1 = (t >> 56) | ((uint64_t)(uint8_t)'_' << 8) | ((uint64_t)(uint8_t)'r' << 16) | (h << 24);  // table id byte 7, "_r", handle bytes 0..4
    *p2 = (uint32_t)(h >> 40);                                                                 // handle bytes 5..7
}

#endif
