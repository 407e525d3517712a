/*
 * orc_result_internal.h — the materialised result set of the oracle's operators (shared by oracle.cpp and mocktikv.cpp).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#ifndef ORC_RESULT_INTERNAL_H
#define ORC_RESULT_INTERNAL_H
#include <cstdint>
#include <string>
#include <vector>

#include "oracle.h"

struct OutCol {
    int32_t type = TSQ_I64;
    std::vector<uint64_t> v;  // raw 64-bit (F32 stored in low 32 bits); TSQ_BYTES: the END offset of the cell in `bytes`
    std::vector<uint8_t> notnull;
    std::string bytes;        // TSQ_BYTES: concatenated data (util/chunk/column.go:28-34: a NULL cell has no bytes)
    void append_raw(uint64_t bits, bool nn) {
        if (type == TSQ_BYTES) bits = bytes.size();  // AppendNull on a var-len column repeats the last offset
        v.push_back(nn || type == TSQ_BYTES ? bits : 0);
        notnull.push_back(nn ? 1 : 0);
    }
    void append_bytes(const void* p, size_t n) {  // Column.AppendBytes (column.go:207-211)
        bytes.append((const char*)p, n);
        v.push_back(bytes.size());
        notnull.push_back(1);
    }
};

struct orc_result {
    std::vector<OutCol> cols;
    int64_t rows = 0;
};

void orc_set_error(const std::string& msg);  // what orc_last_error() returns (thread local, oracle.cpp)
#endif
