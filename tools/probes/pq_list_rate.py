#!/usr/bin/env python3
"""Decode rate of a List<Int64> column chunk on the device (dbhip_pq_chunk_decode_device_list): `--rows` lists of 0..7 elements, 8 % NULL
lists, 10 % NULL elements, written by pyarrow with the reference writer's settings (V1 pages, PLAIN), per codec. Timed: the ABI call
alone (it synchronises its stream), chunk resident in HBM, output buffers allocated; checked against pyarrow on the first 20 000 rows.

    python tools/probes/pq_list_rate.py > gpurun_out/pq_list_rate.json
"""
import argparse
import ctypes as C
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=6_000_000)
    ap.add_argument("--codecs", default="none,zstd,lz4")
    args = ap.parse_args()
    import pyarrow as pa
    import pyarrow.parquet as pq
    from databend_amd import device as D
    from databend_amd import _lib as T
    from tests import parquet_util as PU
    D.init(0)
    L = T.lib()
    rng = np.random.default_rng(21)
    n = args.rows
    lens = rng.integers(0, 8, n).astype(np.int32)
    null_list = rng.random(n) < 0.08
    lens[null_list] = 0
    offsets = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=offsets[1:])
    m = int(offsets[-1])
    vals = rng.integers(-10**12, 10**12, m)
    arr = pa.ListArray.from_arrays(pa.array(offsets, pa.int32()), pa.array(vals, pa.int64(), mask=rng.random(m) < 0.1), mask=pa.array(null_list))
    table = pa.Table.from_arrays([arr], schema=pa.schema([pa.field("c", arr.type)]))
    out = {"rows": n, "elements": m, "cases": {}}
    for cname in args.codecs.split(","):
        data = PU.write_parquet(table, dictionary=False, v2=False, compression=cname)
        ch = PU.column_chunks(data)[0][0]
        pc = D.ParquetChunk(ch["chunk"], ch["physical"], T.T_I64, ch["type_length"], codec=ch["codec"], list_of=(1, 1))
        offs, lv, col = pc.decode_list()                                    # warm-up + the check
        back = pq.read_table(io.BytesIO(data)).column(0).slice(0, 20_000).to_pylist()
        ev = col.validity.to_numpy(np.uint8, (pc.elems + 7) // 8)
        evb = np.unpackbits(ev, bitorder="little")[:pc.elems].astype(bool)
        v = col.data.to_numpy(np.int64, pc.elems)
        got = [None if not lv[r] else [int(v[i]) if evb[i] else None for i in range(int(offs[r]), int(offs[r + 1]))] for r in range(20_000)]
        assert got == back and pc.rows == n and pc.elems == m, cname
        i = pc.info
        d_offs = D.DeviceBuffer((i.num_values + 1) * 8 + 16)
        d_lv = D.DeviceBuffer(i.validity_bytes + 8)
        d_out = D.DeviceBuffer(i.out_bytes + 16)
        d_ev = D.DeviceBuffer(i.validity_bytes + 8)
        rows, elems, nl = C.c_int64(), C.c_int64(), C.c_int64()
        best = 1e9
        for _ in range(4):
            T.check(L.dbhip_stream_sync(None))
            t0 = time.perf_counter()
            T.check(L.dbhip_pq_chunk_decode_device_list(pc.h, C.c_void_p(pc.chunk_dev.ptr), C.c_void_p(pc.image_dev.ptr) if pc.image_dev else None,
                                                        C.c_void_p(d_offs.ptr), C.c_void_p(d_lv.ptr), C.c_void_p(d_out.ptr), C.c_void_p(d_ev.ptr),
                                                        C.byref(rows), C.byref(elems), C.byref(nl), None))
            best = min(best, time.perf_counter() - t0)
        written = (n + 1) * 8 + m * 8 + (n + 7) // 8 + (m + 7) // 8
        out["cases"][cname] = {"chunk_bytes": len(ch["chunk"]), "image_bytes": int(i.image_bytes), "level_entries": int(i.num_values), "pages": int(i.n_pages),
                               "decode_ms": round(best * 1e3, 3), "stored_GBps": round(len(ch["chunk"]) / best / 1e9, 2),
                               "written_bytes": written, "alg_GBps": round((len(ch["chunk"]) + written) / best / 1e9, 2),
                               "rows_per_s": round(n / best), "elements_per_s": round(m / best)}
        pc.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
