"""Cuts the Zstandard frames the REFERENCE keeps under tests/data (with the plaintext it keeps beside one of them) into
tests/golden/zstd_ref/: <name>.zst = the frame as stored, index.json = decoded length + sha256 of what the system's libzstd makes of it.
ontime_200.csv.zst is a known-answer pair: the reference holds its plaintext (tests/data/ontime_200.csv), checked here byte for byte.
Run in the build container (needs /root/reference):  python tests/golden/make_zstd_ref_golden.py"""
import ctypes
import hashlib
import json
import os
import shutil

REF = "/root/reference/tests/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_ref")
FILES = [("ontime_200_csv", "ontime_200.csv.zst", "ontime_200.csv"), ("max_records_csv", "csv/max_records.zst", None),
         ("max_records_ndjson", "ndjson/max_records.zst", None), ("udf_wasm_gcd", "udf/test10_udf_wasm_gcd.wasm.zst", None)]


def main():
    Z = ctypes.CDLL("libzstd.so.1")
    Z.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
    Z.ZSTD_decompress.restype = ctypes.c_size_t
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for name, rel, plain in FILES:
        z = open(os.path.join(REF, rel), "rb").read()
        n = Z.ZSTD_getFrameContentSize(z, len(z))
        buf = ctypes.create_string_buffer(max(n, 1))
        r = Z.ZSTD_decompress(buf, n, z, len(z))
        assert r == n
        d = buf.raw[:n]
        if plain:
            assert d == open(os.path.join(REF, plain), "rb").read(), "the reference's own plaintext differs"
        shutil.copyfile(os.path.join(REF, rel), os.path.join(OUT, name + ".zst"))
        index[name] = dict(source="tests/data/" + rel, reference_plaintext=("tests/data/" + plain) if plain else None, frame_bytes=len(z),
                           decoded_bytes=n, sha256=hashlib.sha256(d).hexdigest())
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1)
    print(index)


if __name__ == "__main__":
    main()
