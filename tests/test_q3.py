"""TPC-H Q3 (BASELINE.json configs[2]): the reference-shaped CPU restatement against an independent
numpy statement of the query, and the GPU operator plan against the restatement."""
import numpy as np
import pytest

from databend_amd import tpch
from tests import oracle_lib as O


def numpy_q3(host, segment, date, limit):
    """Plain set-based statement of benchmark/tpch/queries/03.sql with exact integer arithmetic."""
    c, o, li = host["customer"], host["orders"], host["lineitem"]
    seg = np.zeros(16, np.uint8)
    seg[0] = len(segment)
    seg[4:4 + len(segment)] = np.frombuffer(segment.encode(), np.uint8)
    cust = set(c["c_custkey"][(c["c_mktsegment"] == seg).all(axis=1)].tolist())
    okeep = (o["o_orderdate"] < date) & np.array([k in cust for k in o["o_custkey"].tolist()], dtype=bool)
    orders = {int(k): (int(d), int(s)) for k, d, s in zip(o["o_orderkey"][okeep], o["o_orderdate"][okeep], o["o_shippriority"][okeep])}
    lkeep = li["l_shipdate"] > date
    rev = {}
    for k, p, d in zip(li["l_orderkey"][lkeep].tolist(), li["l_extendedprice"][lkeep].tolist(), li["l_discount"][lkeep].tolist()):
        if k in orders:
            rev[k] = rev.get(k, 0) + p * (100 - d)
    rows = [(k, r, orders[k][0], orders[k][1]) for k, r in rev.items()]
    rows.sort(key=lambda t: (-t[1], t[2]))
    return rows[:limit] if limit else rows


def same_result(got, exp):
    """ORDER BY revenue DESC, o_orderdate: the (revenue, o_orderdate) sequence is defined, ties are unordered."""
    assert [(r[1], r[2]) for r in got] == [(r[1], r[2]) for r in exp]
    assert sorted(got) == sorted(exp)


@pytest.mark.parametrize("sf,threads", [(0.002, 1), (0.01, 3)])
def test_q3_oracle_matches_numpy_statement(sf, threads):
    host = tpch.gen_q3(sf, seed=3)
    st = {}
    got = O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=10, threads=threads, block_rows=4096, stages=st)
    exp = numpy_q3(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, 10)
    assert len(exp) == 10 and st["groups"] > 10
    same_result(got, exp)
    # no LIMIT: every group
    same_result(O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=0, threads=threads), numpy_q3(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, 0))


def test_q3_oracle_empty_results():
    host = tpch.gen_q3(0.001, seed=4)
    assert O.q3_run(host, "NOSUCHSEG", tpch.Q3_DATE) == []
    assert O.q3_run(host, tpch.Q3_SEGMENT, tpch.ORDER_LO) == []          # no order before the first order date
    assert O.q3_run(host, tpch.Q3_SEGMENT, tpch.ORDER_HI + 200) == []    # no line shipped after the last ship date


@pytest.mark.gpu
@pytest.mark.parametrize("sf,seed", [(0.001, 5), (0.02, 3), (0.2, 7)])
def test_q3_gpu_operator_plan_matches_oracle(gpu, sf, seed):
    host = tpch.gen_q3(sf, seed=seed)
    t = tpch.Q3Device(host)
    st, so = {}, {}
    got = tpch.q3_operator_at_a_time(t, stats=st)
    exp = O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=10, threads=4, stages=so)
    for k in ("customers_kept", "orders_kept", "orders_joined", "groups"):
        assert st[k] == so[k], (k, st, so)
    same_result(got, exp)
    # no LIMIT: the whole result, as sorted sets + key sequence
    same_result(tpch.q3_operator_at_a_time(t, limit=0), O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=0, threads=4))
    # the alternative filter->probe plan (filter_select + take, probe the compacted keys) gives the same rows
    same_result(tpch.q3_operator_at_a_time(t, bitmap_probe=False), exp)


@pytest.mark.gpu
def test_q3_gpu_empty_results(gpu):
    host = tpch.gen_q3(0.001, seed=4)
    t = tpch.Q3Device(host)
    assert tpch.q3_operator_at_a_time(t, segment="NOSUCHSEG") == []
    assert tpch.q3_operator_at_a_time(t, date=tpch.ORDER_LO) == []
    assert tpch.q3_operator_at_a_time(t, date=tpch.ORDER_HI + 200) == []


@pytest.mark.gpu
def test_q3_broadcast_join_plan_on_one_rank_over_rccl(gpu):
    """the distributed plan (databend_amd.dist.q3_broadcast_join: build sides all-gathered, probe sides stay, states routed by
    hash) with the device operators and the nccl (= RCCL) backend in a world of one: every collective really runs; the
    multi-rank logic is covered by tests/test_dist_gloo.py with world 2 / 3."""
    import socket

    import torch
    import torch.distributed as dist

    from databend_amd import dist as DX
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        host = tpch.gen_q3(0.05, seed=21)
        t = tpch.Q3Device(host)
        dev = torch.device("cuda", torch.cuda.current_device())
        got = DX.q3_broadcast_join(t, tpch.Q3DeviceOps(torch), dist, torch, dev, limit=10)
        same_result(got, O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=10, threads=4))
        got_all = DX.q3_broadcast_join(t, tpch.Q3DeviceOps(torch), dist, torch, dev, limit=0)
        same_result(got_all, O.q3_run(host, tpch.Q3_SEGMENT, tpch.Q3_DATE, limit=0, threads=4))
    finally:
        dist.destroy_process_group()
