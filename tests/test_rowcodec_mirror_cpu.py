"""CPU: the host logic of the stored-row mirror (tinysql_amd/rowcodec.py) that needs no GPU — the mysql type / flag -> column type
mapping of rowcodec.ColInfo (util/rowcodec/decoder.go:201-236, parser/mysql/type.go) and the default-value bits handed to the
library — and the 64-row response chunking of tinysql_amd/distsql.py (cop_handler_dag.go:510-519)."""
import struct

import numpy as np

from tinysql_amd import _abi as abi
from tinysql_amd import distsql
from tinysql_amd import rowcodec as RC


def test_colinfo_type_mapping():
    for tp in (RC.TypeTiny, RC.TypeShort, RC.TypeInt24, RC.TypeLong, RC.TypeLonglong, RC.TypeYear):
        assert RC.ColInfo(1, tp).tsq_type() == abi.I64 and RC.ColInfo(1, tp, RC.UnsignedFlag).tsq_type() == abi.U64
    assert RC.ColInfo(1, RC.TypeFloat).tsq_type() == abi.F32 and RC.ColInfo(1, RC.TypeDouble).tsq_type() == abi.F64
    for tp in (RC.TypeVarchar, RC.TypeVarString, RC.TypeString, RC.TypeBlob, RC.TypeTinyBlob, RC.TypeMediumBlob, RC.TypeLongBlob):
        assert RC.ColInfo(1, tp).tsq_type() == abi.BYTES  # chk.AppendBytes of the value (decoder.go:226-228)
    assert RC.ColInfo(1, RC.TypeBit).tsq_type() is None and RC.ColInfo(1, RC.TypeBit, Flen=10).tsq_type() == abi.BYTES  # a binary literal of (Flen + 7) / 8 bytes (decoder.go:229-231)
    assert RC.ChunkDecoder(None, [RC.ColInfo(1, RC.TypeBit, Flen=10)]).cols[0].flags == abi.RC_BIT | (2 << 8)


def test_decoder_descriptor_flags_and_default_bits():
    cols = [RC.ColInfo(-1, RC.TypeLonglong, 0, True), RC.ColInfo(3, RC.TypeDouble), RC.ColInfo(4, RC.TypeFloat), RC.ColInfo(5, RC.TypeLonglong, RC.UnsignedFlag),
            RC.ColInfo(6, RC.TypeLong)]
    d = RC.ChunkDecoder(None, cols, -1, lambda i: {1: 2.5, 2: 1.5, 3: (1 << 64) - 1}.get(i))
    assert [d.cols[i].flags for i in range(5)] == [abi.RC_HANDLE, abi.RC_HAS_DEFAULT, abi.RC_HAS_DEFAULT, abi.RC_HAS_DEFAULT, 0]
    assert d.cols[1].def_bits == struct.unpack("<Q", struct.pack("<d", 2.5))[0]
    assert d.cols[2].def_bits == struct.unpack("<I", struct.pack("<f", 1.5))[0]
    assert d.cols[3].def_bits == (1 << 64) - 1
    assert [d.cols[i].col_id for i in range(5)] == [-1, 3, 4, 5, 6] and d.types == [abi.I64, abi.F64, abi.F32, abi.U64, abi.I64]
    neg = RC.ChunkDecoder(None, [RC.ColInfo(7, RC.TypeLonglong)], -1, lambda i: -2)
    assert neg.cols[0].def_bits == (1 << 64) - 2  # int64 -2 as the column stores it


def test_response_chunks_of_64_rows():
    lens = np.array([3] * 130)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    raw = np.arange(offs[-1], dtype=np.int64).astype(np.uint8)
    chunks = distsql.response_chunks(raw, offs)
    assert [len(c) for c in chunks] == [192, 192, 6] and b"".join(chunks) == bytes(raw)
    assert distsql.response_chunks(raw[:0], offs[:1]) == []


def test_string_default_goes_into_the_descriptor_as_bytes():
    d = RC.ChunkDecoder(None, [RC.ColInfo(3, RC.TypeVarchar), RC.ColInfo(4, RC.TypeBlob), RC.ColInfo(5, RC.TypeVarchar)], -1, lambda i: {0: "n/a", 1: b"\x00\x01"}.get(i))
    assert [d.cols[i].flags for i in range(3)] == [abi.RC_HAS_DEFAULT, abi.RC_HAS_DEFAULT, 0]
    assert [d.cols[i].def_len for i in range(3)] == [3, 2, 0] and d.def_len == [3, 2, 0]
    import ctypes as C
    assert C.string_at(d.cols[0].def_bytes, 3) == b"n/a" and C.string_at(d.cols[1].def_bytes, 2) == b"\x00\x01" and not d.cols[2].def_bytes
