"""GPU: dbhip_cast (to_<number> / try_to_<number>) against the CPU restatement for every pair of number types, through the
C-ABI; plus the reference's cast.txt goldens."""
import json
import os

import numpy as np
import pytest

from tests.test_cast_cpu import CODE, HERE, NP, interesting, oracle_cast
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rounding", [False, True])
@pytest.mark.parametrize("is_try", [False, True])
def test_cast_every_number_pair_bit_exact(gpu, is_try, rounding):
    L = O.load()
    rng = np.random.default_rng(29)
    for src in NP:
        arr = interesting(src, rng)
        arr = np.concatenate([arr, arr[rng.integers(0, len(arr), 1000 - len(arr) % 1000 + 37)]])   # ragged length > 4 chunks
        valid = rng.integers(0, 6, len(arr)) > 0
        col = gpu.Column.from_numpy(arr, CODE[src], validity=valid)
        for dst in NP:
            with np.errstate(all="ignore"):
                eout, eok, enerr = oracle_cast(L, arr, src, dst, is_try, rounding, validity=valid)
            c, ok, nerr = gpu.cast(col, CODE[dst], is_try=is_try, rounding_mode=rounding)
            got = c.to_numpy()
            assert np.array_equal(got.view(np.uint8), eout.view(np.uint8)), (src, dst, arr[np.nonzero(got != eout)[0][:5]])
            assert np.array_equal(ok, eok), (src, dst)
            if not is_try:
                assert nerr == enerr


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 256, 257, 100_003])
def test_cast_sizes_and_scalar(gpu, n):
    L = O.load()
    rng = np.random.default_rng(n)
    arr = (rng.standard_normal(n) * 200).astype(np.float64)
    col = gpu.Column.from_numpy(arr, CODE["Float64"])
    c, ok, nerr = gpu.cast(col, CODE["Int8"], is_try=True, rounding_mode=True)
    eout, eok, _ = oracle_cast(L, arr, "Float64", "Int8", True, True)
    assert np.array_equal(c.to_numpy(), eout) and np.array_equal(ok, eok)


def test_cast_goldens_through_the_c_abi(gpu):
    g = json.load(open(os.path.join(HERE, "golden", "cast.json")))
    for case in g["cases"]:
        src = np.array([float(x) if "Float" in case["src_type"] else int(x) for x in case["src"]], dtype=NP[case["src_type"]])
        c, ok, nerr = gpu.cast(gpu.Column.from_numpy(src, CODE[case["src_type"]]), CODE[case["dst_type"]], is_try=case["try"], rounding_mode=False)
        exp = np.array([float(x) if "Float" in case["dst_type"] else int(x) for x in case["out"]], dtype=NP[case["dst_type"]])
        assert np.array_equal(c.to_numpy(), exp), case["ast"]
        if case["try"]:
            assert ok.tolist() == case["validity"]
        else:
            assert nerr == 0
    with pytest.raises(Exception):
        gpu.cast(gpu.Column.from_numpy(np.zeros(4, np.int32), 12), CODE["Int64"])   # DATE: not a number type here
