#!/usr/bin/env python3
"""bench.py's `materialising` side measurement alone (the 1e8 x 1e8 join with its four output columns written to HBM)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import _lib  # noqa: E402


def main():
    nb = npr = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    with _lib.Context(0) as ctx:
        bk, bv, pk, pv = (ctx.alloc(nb * 8) for _ in range(4))
        ctx.gen_column(bench._spec(abi, abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=nb), nb, bk)
        ctx.gen_column(bench._spec(abi, abi.GEN_HASH_OF_COL, table=2, b=0xABCDEF), nb, bv, src=bk)
        ctx.gen_column(bench._spec(abi, abi.GEN_RAND_MOD, table=1, col=0, m=nb), npr, pk)
        ctx.gen_column(bench._spec(abi, abi.GEN_RAND_MOD, table=1, col=1, m=1 << 62), npr, pv)
        ctx.sync()
        print(json.dumps(bench.extra_materialising(ctx, abi, _lib, bk, bv, pk, pv, nb, npr)))


if __name__ == "__main__":
    main()
