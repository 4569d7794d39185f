"""What a scan of TPC-H Q1's columns hands the device: the column chunks of several blocks, written the way the reference's writer does
(storages/common/blocks/src/parquet_rs.rs:113-160 — one row group per block, DATA_PAGE_V2 + RLE_DICTIONARY with the dictionary switched
off for high-cardinality columns, pages of at most 20,000 rows (parquet-rs' default row-count limit), TableCompression Zstd at
ZstdLevel::default() = 1 / LZ4 / Snappy), decoded by ONE dbhip_pq_chunks_decode_device call.
    python tools/pq_scan_probe.py [--codec zstd|lz4|snappy|none] [--blocks 8] [--rows 6000000] [--reps 5] [--one-by-one] [--out f.json]
Prints one JSON line: stored / image / output bytes, ms per batch, GB/s of each."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from databend_amd import _lib as T          # noqa: E402
from databend_amd import device as D        # noqa: E402
from tests import parquet_util as PU        # noqa: E402


def lineitem_block(rng, n):
    """the seven columns Q1 reads, as (name, pyarrow array, out type, numpy source, dictionary?)"""
    import pyarrow as pa
    qty = rng.integers(1, 51, n) * 100
    price = rng.integers(90000, 10494951, n)
    disc = rng.integers(0, 11, n)
    tax = rng.integers(0, 9, n)
    ship = rng.integers(8036, 10561, n).astype(np.int32)
    rf = np.array([b"A", b"N", b"R"], dtype=object)[rng.integers(0, 3, n)]
    ls = np.array([b"F", b"O"], dtype=object)[rng.integers(0, 2, n)]
    return [("l_quantity", pa.array(qty, pa.int64()), T.T_DEC64, qty, True), ("l_extendedprice", pa.array(price, pa.int64()), T.T_DEC64, price, False),
            ("l_discount", pa.array(disc, pa.int64()), T.T_DEC64, disc, True), ("l_tax", pa.array(tax, pa.int64()), T.T_DEC64, tax, True),
            ("l_returnflag", pa.array(list(rf), pa.binary()), T.T_STRING, rf, True), ("l_linestatus", pa.array(list(ls), pa.binary()), T.T_STRING, ls, True),
            ("l_shipdate", pa.array(ship, pa.int32()).cast(pa.date32()), T.T_DATE, ship, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codec", default="zstd")
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=6_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--page-rows", type=int, default=20_000)
    ap.add_argument("--one-by-one", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-check", action="store_true", help="skip the identity check (experiments that decode wrongly on purpose)")
    args = ap.parse_args()
    res = run(args)
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


def run(args=None, **kw_args):
    """-> the result record; `args`: the parsed command line, or keywords (codec, blocks, rows, reps, page_rows, one_by_one, no_check) from bench.py"""
    import pyarrow as pa
    if args is None:
        args = argparse.Namespace(**dict(dict(codec="zstd", blocks=8, rows=6_000_000, reps=5, page_rows=20_000, one_by_one=False, no_check=False), **kw_args))
    rng = np.random.default_rng(7)
    D.init(0)
    pcs, srcs = [], []
    kw = dict(compression=args.codec, max_rows_per_page=args.page_rows, page_size=1 << 20)
    if args.codec == "zstd":
        kw["compression_level"] = 1        # ZstdLevel::default() of the parquet crate
    t0 = time.perf_counter()
    block = lineitem_block(rng, args.rows)      # (the same values in every block: what is timed does not depend on them)
    for name, arr, ot, src, dictionary in block:
        chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": arr}), dictionary=dictionary, v2=True, **kw))
        ch = chunks[0]
        for b in range(args.blocks):
            pc = D.ParquetChunk(ch["chunk"], ch["physical"], ot, ch["type_length"], ch["max_def"], 0, ch["codec"], precision=15, scale=2, device=True)
            pc.upload()
            pcs.append(pc)
            srcs.append((name, ot, src))
    write_s = time.perf_counter() - t0
    stored = sum(len(pc.host) for pc in pcs)
    image = sum(int(pc.info.image_bytes) for pc in pcs)
    out_bytes = sum(int(pc.info.out_bytes) for pc in pcs)
    pages = sum(int(pc.info.n_pages) for pc in pcs)

    def run():
        if args.one_by_one:
            return [pc.decode() for pc in pcs]
        return D.ParquetChunk.decode_many(pcs)
    cols = run()
    # the decode is the identity on what was written
    for (name, ot, src), col, pc in ([] if args.no_check else list(zip(srcs, cols, pcs))[:: args.blocks]):
        if ot == T.T_STRING:
            v = col.data.to_numpy(np.uint8, 16 * len(src)).reshape(-1, 16)
            assert (v[:, 0] == 1).all() and np.array_equal(v[:, 4], np.frombuffer(b"".join(src), np.uint8)), name
        elif ot == T.T_DATE:
            assert np.array_equal(col.data.to_numpy(np.int32, len(src)), src), name
        else:
            assert np.array_equal(col.data.to_numpy(np.int64, len(src)), src), name
    del cols
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        cols = run()          # (the call synchronises its stream)
        ts.append(time.perf_counter() - t0)
        del cols
    best = min(ts)
    res = dict(codec=args.codec, blocks=args.blocks, rows_per_block=args.rows, chunks=len(pcs), pages=pages, page_rows=args.page_rows,
               mode="one call per chunk" if args.one_by_one else "one dbhip_pq_chunks_decode_device call", stored_bytes=stored, image_bytes=image,
               out_bytes=out_bytes, ms=round(best * 1e3, 3), all_ms=[round(t * 1e3, 3) for t in ts], stored_GBps=round(stored / best / 1e9, 2),
               image_GBps=round((image or stored) / best / 1e9, 2), out_GBps=round(out_bytes / best / 1e9, 2),
               rows_per_s=round(args.rows * args.blocks / best), write_seconds=round(write_s, 1))
    for pc in pcs:
        pc.close()
    return res


if __name__ == "__main__":
    main()
