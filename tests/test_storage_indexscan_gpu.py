"""GPU parity of tsq_indexkeys_decode / coprocessor.indexScanExec (SURVEY.md §8 f rank 4) against the oracle's restatement of
mocktikv's indexScanExec = tablecodec.DecodeIndexKV per pair (store/mockstore/mocktikv/executor.go:191-320,
tablecodec/tablecodec.go:376-465) + Decoder.DecodeOne of every cut value (util/codec/codec.go:623-690, bytes.go:69-118)."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import coprocessor as cop
from tinysql_amd.chunk import Chunk, Column, StrColumn, chunk_from_buffers, out_buffers
from tinysql_amd.gpu_pipeline import drain_device

pytestmark = pytest.mark.gpu

MSG = {1: "invalid encoded key", 2: "insufficient bytes to decode value", 3: "value larger than 64 bits", 4: "invalid encoded key flag",
       6: "datum kind does not match the column type", 7: "invalid marker byte", 8: "invalid padding byte", 9: "no handle in index key or value"}


def _index_rows(rng, n, long_strings=False):
    names = [None if rng.random() < 0.1 else bytes(rng.integers(0, 256, int(rng.integers(0, 120 if long_strings else 20)), dtype=np.uint8)) for _ in range(n)]
    return Chunk([Column(abi.I64, rng.integers(-1 << 40, 1 << 40, n), rng.random(n) >= 0.1), StrColumn(names), Column(abi.F64, np.round(rng.standard_normal(n), 3), rng.random(n) >= 0.1),
                  Column(abi.U64, rng.integers(0, 1 << 63, n).astype(np.uint64) * np.uint64(2), None)])


def _values(handles, in_key):
    """a unique index stores the handle as its value (8 bytes big endian); a key that carries the handle has the value '0'"""
    parts = [b"0" if k else int(h).to_bytes(8, "big", signed=True) for h, k in zip(handles, in_key)]
    return b"".join(parts), np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)


def _gpu_decode(ctx, keys, koffs, vals, voffs, colsLen, types, pk, cap=None):
    raw = np.frombuffer(bytes(keys) + b"\0" * 8, np.uint8)
    n = len(koffs) - 1
    keep = []
    out, bufs = out_buffers(types, max(cap or n, 1), keep, var_bytes=[len(keys) if t == abi.BYTES else 0 for t in types])
    tp = (C.c_int32 * len(types))(*types)
    v = None if vals is None else np.frombuffer(bytes(vals) + b"\0" * 8, np.uint8)
    got = C.c_int64(0)
    st = ctx.lib.tsq_indexkeys_decode(ctx.h, raw.ctypes.data_as(C.c_void_p), len(keys), np.ascontiguousarray(koffs).ctypes.data_as(C.c_void_p), n,
                                      None if v is None else v.ctypes.data_as(C.c_void_p), 0 if vals is None else len(vals),
                                      None if voffs is None else np.ascontiguousarray(voffs).ctypes.data_as(C.c_void_p), 0, colsLen, tp, pk, out, C.byref(got))
    return st, chunk_from_buffers(types, bufs, got.value)


@pytest.mark.parametrize("n", [1, 63, 1000, 40_000])
@pytest.mark.parametrize("form", ["unique", "non_unique", "mixed", "no_pk"])
def test_index_pairs_against_the_oracle(ctx, orc, n, form):
    rng = np.random.default_rng(n + len(form))
    t = _index_rows(rng, n, long_strings=(n == 1000))
    handles = rng.integers(-1 << 62, 1 << 62, n)
    in_key = {"unique": np.zeros(n, bool), "non_unique": np.ones(n, bool), "mixed": rng.random(n) < 0.5, "no_pk": rng.random(n) < 0.5}[form]
    keys, koffs = orc.encode_index_keys(t, 45, 3, handles, in_key)
    vals, voffs = _values(handles, in_key)
    pk = 0 if form == "no_pk" else 1
    types = t.types() + ([abi.I64] if pk else [])
    st, want = orc.decode_index_kv(keys.tobytes(), koffs, vals, voffs, 4, types, pk)
    gst, got = _gpu_decode(ctx, keys.tobytes(), koffs, vals, voffs, 4, types, pk)
    assert st == 0 and gst == abi.OK and got.NumRows() == n
    assert got.rows() == want.rows()
    if pk:
        assert [r[-1] for r in got.rows()] == handles.tolist() and [r[:-1] for r in got.rows()] == t.rows()


def test_unsigned_handles_and_value_form_datums(ctx, orc):
    # PrimaryKeyIsUnsigned: the value's 8 bytes read as uint64 (tablecodec.go:421-425); DecodeOne also takes the VALUE forms
    # (varint, compact bytes) inside a key — e.g. keys assembled by EncodeValue in a test
    rng = np.random.default_rng(3)
    n = 500
    t = Chunk([Column(abi.I64, rng.integers(-1000, 1000, n), None), StrColumn([b"v%d" % i for i in range(n)])])
    handles = rng.integers(0, 1 << 63, n).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    rows = orc.encode_rows  # value form rows behind a hand-made prefix
    prefix = b"t" + (45 ^ (1 << 63)).to_bytes(8, "big") + b"_i" + (7 ^ (1 << 63)).to_bytes(8, "big")
    parts = [prefix + bytes(rows(t.slice(i, i + 1))) for i in range(n)]
    keys, koffs = b"".join(parts), np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    vals = b"".join(int(h).to_bytes(8, "big") for h in handles)
    voffs = np.arange(n + 1, dtype=np.int64) * 8
    types = [abi.I64, abi.BYTES, abi.U64]
    st, want = orc.decode_index_kv(keys, koffs, vals, voffs, 2, types, 2)
    gst, got = _gpu_decode(ctx, keys, koffs, vals, voffs, 2, types, 2)
    assert st == 0 and gst == abi.OK and got.rows() == want.rows()
    assert [r[2] for r in got.rows()] == handles.tolist()


@pytest.mark.parametrize("case", ["short_key", "cut_column", "bad_marker", "bad_padding", "bad_flag", "no_value", "short_value", "string_handle", "kind"])
def test_first_offending_pair(ctx, orc, case):
    rng = np.random.default_rng(9)
    n = 300
    t = _index_rows(rng, n)
    handles = np.arange(n, dtype=np.int64) * 3 - 50
    in_key = np.arange(n) % 2 == 0
    keys, koffs = orc.encode_index_keys(t, 45, 3, handles, in_key)
    vals, voffs = _values(handles, in_key)
    keys = bytearray(keys.tobytes())
    koffs = koffs.copy()
    types = t.types() + [abi.I64]
    bad = 101  # (an odd pair: its handle is in the value)
    lo, hi = int(koffs[bad]), int(koffs[bad + 1])
    if case == "short_key":
        del keys[lo + 10:hi]
        koffs[bad + 1:] -= hi - lo - 10
    elif case == "cut_column":  # the key ends inside its last column (the uint: flag 4 + 8 bytes)
        del keys[hi - 4:hi]
        koffs[bad + 1:] -= 4
    elif case in ("bad_marker", "bad_padding"):
        t.columns[1]._vals[bad] = b"abc"
        t = Chunk([t.columns[0], StrColumn(t.columns[1]._vals), t.columns[2], t.columns[3]])
        k2, o2 = orc.encode_index_keys(t, 45, 3, handles, in_key)
        keys, koffs = bytearray(k2.tobytes()), o2.copy()
        lo = int(koffs[bad])
        at = lo + 19 + (9 if not t.columns[0].IsNull(bad) else 1)  # the string datum: flag, "abc", 5 pad bytes, marker 250
        assert keys[at] == 1 and keys[at + 9] == 250
        if case == "bad_marker":
            keys[at + 9] = 200
        else:
            keys[at + 5] = 7
    elif case == "bad_flag":
        keys[lo + 19] = 6
    elif case == "no_value":
        vals, voffs = None, None
    elif case == "short_value":
        vals = bytearray(vals)
        v0 = int(voffs[bad])
        del vals[v0:v0 + 3]
        voffs = voffs.copy()
        voffs[bad + 1:] -= 3
        vals = bytes(vals)
    elif case == "string_handle":  # a pair whose remainder behind the index columns is a string datum
        bad = 100
        hi = int(koffs[bad + 1])
        keys[hi - 9:hi] = b"\x02\x0e" + b"handle!"
    else:  # a string where the bigint column's value belongs
        types = [abi.BYTES] + types[1:]
        bad = 0
        while t.columns[0].IsNull(bad):
            bad += 1
    st, want = orc.decode_index_kv(bytes(keys), koffs, vals, voffs, 4, types, 1)
    gst, got = _gpu_decode(ctx, bytes(keys), koffs, vals, voffs, 4, types, 1)
    assert st != 0 and gst == abi.ERR_INVALID and _lib.last_error(ctx.h) == MSG[st], (st, _lib.last_error(ctx.h))
    if case == "no_value":
        bad = 1
    assert want.NumRows() == got.NumRows() == bad and got.rows() == want.rows()


def test_index_scan_feeds_the_pushed_down_chain(ctx, orc):
    # indexScanExec -> selectionExec -> limitExec, the device chunks staying in HBM; the same rows from the row-at-a-time restatement
    from tinysql_amd import expression as E
    rng = np.random.default_rng(12)
    n = 20_000
    t = Chunk([Column(abi.I64, rng.integers(0, 100, n), rng.random(n) >= 0.05), StrColumn([None if i % 13 == 0 else b"name-%05d" % (i % 977) for i in range(n)])])
    handles = np.arange(n, dtype=np.int64) + 1
    in_key = np.ones(n, bool)
    keys, koffs = orc.encode_index_keys(t, 45, 2, handles, in_key)
    types = [abi.I64, abi.BYTES, abi.I64]
    scan = cop.indexScanExec(ctx, types, 2, cop.indexScanExec.PrimaryKeyIsSigned, keys, koffs, batch_rows=4096)
    sel = cop.selectionExec(ctx, scan, [E.ScalarFunction("gt", E.Column(0, abi.I64), E.Constant(90))])
    lim = cop.limitExec(ctx, sel, 500)
    got = [r for c in drain_device(lim) for r in c.rows()]
    st, rows = orc.decode_index_kv(keys.tobytes(), koffs, None, None, 2, types, 1)
    want = [r for r in rows.rows() if r[0] is not None and r[0] > 90][:500]
    assert st == 0 and got == want and len(got) == 500


def test_full_size_fixed_width_index_device_resident(ctx):
    # 2e7 pairs of a non-unique (bigint, bigint) index: keys assembled on the host once (numpy), decoded on the device, compared with the source
    n = 20_000_000
    rng = np.random.default_rng(1)
    a, b, h = rng.integers(-1 << 62, 1 << 62, n), rng.integers(-1 << 62, 1 << 62, n), np.arange(n, dtype=np.int64)
    key = np.zeros((n, 46), np.uint8)  # 19 + 9 + 9 + 9
    key[:, 0] = ord("t")
    key[:, 1:9] = np.frombuffer((45 ^ (1 << 63)).to_bytes(8, "big"), np.uint8)
    key[:, 9], key[:, 10] = ord("_"), ord("i")
    key[:, 11:19] = np.frombuffer((2 ^ (1 << 63)).to_bytes(8, "big"), np.uint8)
    for j, v in enumerate((a, b, h)):
        at = 19 + 9 * j
        key[:, at] = 3
        key[:, at + 1:at + 9] = (v.view(np.uint64) ^ np.uint64(1 << 63)).astype(">u8").view(np.uint8).reshape(n, 8)
    koffs = np.arange(n + 1, dtype=np.int64) * 46
    dk, do = ctx.alloc(key.size + 64), ctx.alloc(koffs.nbytes + 64)
    ctx.h2d(dk, key.reshape(-1))
    ctx.h2d(do, koffs)
    from . import gpu_helpers as G
    outs = [G.DevCol(ctx, abi.I64, n, True) for _ in range(3)]
    oarr = G.dev_cols(outs)
    tp = (C.c_int32 * 3)(abi.I64, abi.I64, abi.I64)
    got = C.c_int64(0)
    ctx.timer_start()
    _lib.check(ctx.lib.tsq_indexkeys_decode(ctx.h, C.c_void_p(dk), key.size, C.c_void_p(do), n, None, 0, None, abi.COL_DEVICE, 2, tp, 1, oarr, C.byref(got)), ctx.h)
    ms = ctx.timer_stop_ms()
    assert got.value == n
    for col, v in zip(outs, (a, b, h)):
        hc = col.to_host()
        assert (hc.data == v).all() and (hc.notnull is None or hc.notnull.all())
    print("tsq_indexkeys_decode: %.2f ms for %d pairs (%.1f GB/s of keys)" % (ms, n, key.size / ms / 1e6))
    for c in outs:
        c.free()
    ctx.free(dk)
    ctx.free(do)
