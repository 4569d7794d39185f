"""Decodes one BASELINE-sized compressed chunk through the device mode of the scan side a few times (for rocprofv3 runs):
    python tools/pq_device_probe.py [snappy|lz4] [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from databend_amd import _lib as T          # noqa: E402
from databend_amd import device as D        # noqa: E402
from tests import parquet_util as PU        # noqa: E402


def main():
    import pyarrow as pa
    codec = sys.argv[1] if len(sys.argv) > 1 else "snappy"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
    rng = np.random.default_rng(4)
    price = rng.integers(90000, 10494951, n)
    chunks, _ = PU.column_chunks(PU.write_parquet(pa.table({"c": pa.array(price, pa.int64())}), dictionary=False, compression=codec))
    ch = chunks[0]
    D.init(0)
    pc = D.ParquetChunk(ch["chunk"], ch["physical"], T.T_DEC64, ch["type_length"], ch["max_def"], 0, ch["codec"], precision=15, scale=2, device=True)
    for _ in range(4):
        col = pc.decode()
    assert np.array_equal(col.data.to_numpy(np.int64, n), price)
    print("ok", codec, n, "rows", len(ch["chunk"]), "bytes stored", pc.info.n_pages, "pages")


if __name__ == "__main__":
    main()
