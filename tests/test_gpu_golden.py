"""GPU: every reference golden vector of the hot path (tests/golden/*.json) through the C-ABI — the same cases and the same
evaluator (tests/golden_eval.py) as tests/test_golden_cpu.py, with libdbhip.so as the backend."""
import ctypes as C

import numpy as np
import pytest

from databend_amd import _lib as T
from tests import golden_eval as G
from tests.test_golden_cpu import AGG_KIND, agg_argument, agg_column, chunk_pairs, compare_agg, expected_agg, golden

pytestmark = pytest.mark.gpu


class DeviceBackend:
    """tests/golden_eval.py backend over libdbhip.so (databend_amd.device)"""

    def __init__(self, gpu):
        self.D = gpu

    def col(self, v):
        D = self.D
        if v.dtype in (T.T_DEC128, T.T_DEC256):
            words = 2 if v.dtype == T.T_DEC128 else 4
            c = D.Column(v.dtype, len(v.arr) // words, D.DeviceBuffer.from_numpy(np.ascontiguousarray(v.arr)), precision=v.precision, scale=v.scale)
        elif v.dtype == T.T_BOOL:
            c = D.Column.boolean(np.asarray(v.arr, dtype=bool))
        else:
            c = D.Column.from_numpy(np.ascontiguousarray(v.arr), v.dtype, precision=v.precision, scale=v.scale)
        c.is_scalar = v.is_scalar
        return c

    def arith(self, op, a, b, n):
        out = self.D.arith(op, self.col(a), self.col(b), n)
        return G.Val(out.dtype, out.to_numpy())

    def decimal(self, op, a, b, n):
        out = self.D.decimal_arith(op, self.col(a), self.col(b), n)
        words = {T.T_DEC64: 0, T.T_DEC128: 2, T.T_DEC256: 4}[out.dtype]
        arr = out.data.to_numpy(np.uint64, words * n) if words else out.to_numpy()
        return G.Val(out.dtype, arr, None, out.precision, out.scale)

    def decimal_neg(self, a, n):
        out = self.D.decimal_neg(self.col(a), n)
        words = {T.T_DEC64: 0, T.T_DEC128: 2, T.T_DEC256: 4}[out.dtype]
        arr = out.data.to_numpy(np.uint64, words * n) if words else out.to_numpy()
        return G.Val(out.dtype, arr, None, out.precision, out.scale)

    def cmp(self, op, a, b, n):
        if a.dtype != b.dtype and not (a.is_decimal and b.is_decimal):
            raise G.Skip("comparison of different physical types without a CAST")
        return self.D.cmp(op, self.col(a), self.col(b), n).to_numpy()


def test_arithmetic_goldens_through_the_c_abi(gpu):
    """all of arithmetic.txt's hot-path cases, the Decimal(76,x) ones and unary minus on decimals included"""
    checked, skipped = G.run_cases(golden("arithmetic.json"), DeviceBackend(gpu))
    assert len(checked) >= 75, (len(checked), skipped)
    assert not any("Decimal256" in k or "unary minus" in k for k in skipped), skipped


def test_comparison_goldens_through_the_c_abi(gpu):
    checked, skipped = G.run_cases(golden("comparison.json"), DeviceBackend(gpu))
    assert len(checked) >= 21, (len(checked), skipped)


def test_aggregate_goldens_through_the_c_abi(gpu):
    """{sum,count,avg,min,max}[_group_by].txt through dbhip_groupby_*: values and NULL results."""
    D = gpu
    checked, skipped = [], {}
    for case in golden("aggregates.json"):
        try:
            func, spec, n = agg_argument(case)
            kind, exp_vals, exp_valid = expected_agg(case)
            plan = ["sum", "count"] if func == "avg" else [func]
            aggs, args = [], []
            scale = 0
            if spec is not None:
                arr, validity, code, prec, scale = agg_column(spec, n)
            for f in plan:
                if spec is None:
                    if f != "count":
                        raise G.Skip("argument-less " + f)
                    aggs.append((T.AGG_COUNT, 0, 0, 0, 0))
                    args.append(None)
                    continue
                aggs.append((AGG_KIND[f], code, prec, scale, 1 if validity is not None else 0))
                if code == T.T_STRING:
                    args.append(D.Column.strings(arr, validity=validity))
                elif code == T.T_DEC128:
                    args.append(D.Column.decimal128(arr, prec, scale, validity=validity))
                elif code == T.T_DEC64:
                    args.append(D.Column.from_numpy(np.array(arr, np.int64), T.T_DEC64, validity=validity, precision=prec, scale=scale))
                else:
                    args.append(D.Column.from_numpy(arr, code, validity=validity))
            groups = (np.arange(n) % 2).astype(np.uint8) if case["grouped"] else np.zeros(n, np.uint8)
            g = D.GroupBy([T.T_U8], aggs)
            g.add_block([D.Column.from_numpy(groups)], args, n)
            rows = sorted(g.result())
            compare_agg(func, rows, kind, exp_vals, exp_valid, scale, case["ast"])
            # the same golden through partial -> serialized-state block -> final (what a GPU partial aggregate hands the CPU final stage:
            # Payload::aggregate_flush / batch_merge; min(s) / max(s) travel as a Nullable(String) column, aggregate_min_max_any.rs:163-205)
            kcols, fcols = g.flush_state_block()
            final = D.GroupBy([T.T_U8], aggs)
            final.merge_state_block(kcols, fcols, kcols[0].n)
            del kcols, fcols
            g.destroy()
            compare_agg(func, sorted(final.result()), kind, exp_vals, exp_valid, scale, case["ast"] + " (through the state block)")
            checked.append(case["ast"])
        except G.Skip as e:
            skipped[e.args[0]] = skipped.get(e.args[0], 0) + 1
    assert len(checked) >= 80 and "min(s)" in checked and "max(s)" in checked, (len(checked), skipped)


def test_kernel_pass_filter_and_take_goldens_through_the_c_abi(gpu):
    D = gpu
    for case in [c for c in golden("kernel.json") if c["kind"] in ("filter", "take")]:
        src = case["source"]
        n = len(src)
        if case["kind"] == "filter":
            sel, k = D.filter_select(D.Column.boolean(np.array(case["arg"], bool)))
        else:
            k = len(case["arg"])
            sel = D.DeviceBuffer.from_numpy(np.array(case["arg"], np.uint32))
        for c in range(len(case["header"])):
            cells = [r[c] for r in src]
            valid = np.array([x != "NULL" for x in cells])
            if any(x.startswith("'") for x in cells):
                col = D.Column.strings([x.strip("'").encode() if x != "NULL" else b"" for x in cells], validity=valid)
                out = D.take(col, sel, k)
                vals = [("'" + s.decode() + "'") for s in D.view_strings(out.to_numpy())]
            else:
                col = D.Column.from_numpy(np.array([int(x) if x != "NULL" else 0 for x in cells], np.int32), validity=valid)
                out = D.take(col, sel, k)
                vals = [str(v) for v in out.to_numpy().tolist()]
            ov = out.validity_numpy()
            got = [v if ok else "NULL" for v, ok in zip(vals, ov)]
            assert got == [r[c] for r in case["result"]], (case["kind"], c, got)


def test_take_block_goldens_and_range_repeat_selections_through_the_c_abi(gpu, oracle):
    """kernel-pass.txt 'Take Block indices' / 'by slices' through dbhip_take_chunks (values and validity), and
    dbhip_sel_from_ranges / dbhip_sel_from_repeats against the oracle on random run-length inputs."""
    D = gpu
    for case in [c for c in golden("kernel.json") if c["kind"] in ("chunks", "slices")]:
        pairs = chunk_pairs(case)
        for c in range(2):
            cols = []
            for b in case["blocks"]:
                cells = [r[c] for r in b]
                valid = np.array([x != "NULL" for x in cells])
                cols.append(D.Column.from_numpy(np.array([int(x) if x != "NULL" else 0 for x in cells], np.int32),
                                                validity=None if valid.all() and c == 0 else valid))
            out = D.take_chunks(cols, pairs)
            vals, ov = out.to_numpy().tolist(), out.validity_numpy()
            assert [str(v) if ok else "NULL" for v, ok in zip(vals, ov)] == [r[c] for r in case["result"]], (case["kind"], c)
    rng = np.random.default_rng(4)
    for n_items in (1, 7, 3000):
        starts = rng.integers(0, 1 << 20, n_items).astype(np.uint32)
        lens = rng.integers(0, 300, n_items).astype(np.uint32)
        ranges = np.stack([starts, starts + lens], axis=1).astype(np.uint32)
        total = int(lens.sum())
        exp = np.zeros(max(total, 1), np.uint32)
        oracle.orc_sel_from_ranges.restype = C.c_int64
        oracle.orc_sel_from_repeats.restype = C.c_int64
        assert oracle.orc_sel_from_ranges(ranges.ctypes.data_as(C.c_void_p), n_items, exp.ctypes.data_as(C.c_void_p)) == total
        got = D.sel_from_ranges(ranges, total).to_numpy(np.uint32, total)
        assert np.array_equal(got, exp[:total])
        reps = np.stack([starts, lens], axis=1).astype(np.uint32)
        assert oracle.orc_sel_from_repeats(reps.ctypes.data_as(C.c_void_p), n_items, exp.ctypes.data_as(C.c_void_p)) == total
        got = D.sel_from_repeats(reps, total).to_numpy(np.uint32, total)
        assert np.array_equal(got, exp[:total])
    with pytest.raises(T.DbhipError):     # the items must cover exactly num_rows (debug_assert in the reference)
        D.sel_from_ranges([(0, 5)], 4)


def test_sort_goldens_through_the_c_abi(gpu):
    D = gpu
    for case in golden("sort.json"):
        cols = []
        for c in case["source"]:
            if c["kind"] == "String":
                cols.append(D.Column.strings([s.encode() for s in c["values"]]))
            elif c["kind"] == "Decimal128":
                cols.append(D.Column.decimal128(c["values"], 38, 0))
            else:
                cols.append(D.Column.from_numpy(np.array(c["values"], np.int64)))
        keys = [cols[d["offset"]] for d in case["sort"]]
        perm = D.sort_perm(keys, desc=[0 if d["asc"] else 1 for d in case["sort"]], nulls_first=[1 if d["nulls_first"] else 0 for d in case["sort"]],
                           limit=case["limit"])
        for c, e in zip(case["source"], case["expected"]):
            assert [c["values"][i] for i in perm] == e["values"], (case["sort"], case["limit"])


def test_hash_index_cases_through_merge_serialized(gpu):
    """hash_index/index.rs:385-404 on the device table: rows with INJECTED hashes (the serialized row carries its hash word)
    — keys 1 and 3 share a hash, the table already holds (4, hash, 77): three new groups, every key keeps its own state."""
    D = gpu
    for inc, pay in (([(1, 123), (2, 456), (3, 123), (4, 44)], [(4, 44, 77)]),
                     ([(1, 11 << 48), (2, 22 << 48), (3, 33 << 48), (4, 44 << 48)], [(4, 44 << 48, 77)])):
        g = D.GroupBy([T.T_U64], [(T.AGG_SUM, T.T_U64, 0, 0, 0)], capacity=16)
        g.merge_serialized(np.array([[k, h, v] for k, h, v in pay], np.uint64))
        before = g.num_groups()
        g.merge_serialized(np.array([[k, h, (k + 20 if k != 4 else 0)] for k, h in inc], np.uint64))
        assert g.num_groups() - before == 3
        assert dict((r[0], r[1]) for r in g.result()) == {1: 21, 2: 22, 3: 23, 4: 77}


def test_arithmetic_decimal_golden_through_the_c_abi(gpu):
    """arithmetic_decimal.txt: `l_extendedprice + (1 - l_discount) - l_quantity` over Decimal(15, 2) columns through dbhip_decimal_arith,
    node by node, with the DecimalSize of every node (tests/test_golden_cpu.py runs the same fixture through the oracle)"""
    import json
    import os
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arithmetic_decimal.json")))
    sizes = []

    class Recording(DeviceBackend):
        def decimal(self, op, a, b, n):
            out = DeviceBackend.decimal(self, op, a, b, n)
            sizes.append((out.precision, out.scale))
            return out
    checked, skipped = G.run_cases(data["cases"], Recording(gpu))
    assert len(checked) == 1 and not skipped, skipped
    assert sizes == [(16, 2), (17, 2), (18, 2)], sizes


def test_arithmetic_decimal_folded_constant_through_the_c_abi(gpu):
    """the constant the reference folds, `1964831797.0000 - 0.0214642400000` (Decimal(14, 4) - Decimal(13, 13) -> Decimal(24, 13)), as a
    one-row evaluation with two scalar operands"""
    import json
    import os
    from decimal import Decimal
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arithmetic_decimal.json")))
    be = DeviceBackend(gpu)
    for f in data["folded"]:
        got = G.evaluate(G.parse_expr(f["expr"]), {}, be, 1)
        kind = G.parse_type(f["output_type"])
        assert (got.precision, got.scale) == (kind[1], kind[2]), (f["ast"], got.precision, got.scale)
        assert got.ints()[:1] == [int(Decimal(f["output"]).scaleb(kind[2]))], (f["ast"], got.ints())
