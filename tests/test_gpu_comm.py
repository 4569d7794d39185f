"""GPU: the communicator behind the C-ABI (dbhip_comm_*, dbhip_groupby_exchange_*) on the one GPU a test box has — a world
of one, both as the local object (no RCCL) and as a REAL RCCL communicator of one rank (ncclGetUniqueId / ncclCommInitRank /
ncclAllGather / grouped ncclSend + ncclRecv / ncclAllReduce through librccl.so): the exchanges must leave the table's result
unchanged, report an overflowing block before touching the table, and the plain collectives must be identities. The N > 1
behaviour of the same block protocol is covered over gloo by tests/test_dist_gloo.py (world size 2)."""
import numpy as np
import pytest

from databend_amd import _lib as T

pytestmark = pytest.mark.gpu


def make_table(D, n_groups, n=200_000, seed=1):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, n_groups, n).astype(np.int64)
    a = rng.integers(-10**6, 10**6, n).astype(np.int64)
    g = D.GroupBy([T.T_I64], [(T.AGG_SUM, T.T_I64, 0, 0, 0), (T.AGG_COUNT, 0, 0, 0, 0)])
    g.add_block([D.Column.from_numpy(k)], [D.Column.from_numpy(a), None], n)
    return g


@pytest.mark.parametrize("real_rccl", [False, True])
def test_exchanges_in_a_world_of_one_leave_the_result_unchanged(gpu, real_rccl):
    D = gpu
    comm = D.Comm(0, 1, D.Comm.unique_id()) if real_rccl else D.Comm.local()
    for groups in (4, 150):
        g = make_table(D, groups)
        before = sorted(g.result())
        comm.exchange_allgather(g, max_rows=256)      # every rank ends with the global result: here, its own
        assert sorted(g.result()) == before
        comm.exchange_alltoall(g, max_rows=256)       # rank 0 of 1 owns every hash class
        assert sorted(g.result()) == before
    # an overflowing block is reported before the table is touched
    g = make_table(D, 1000)
    before = sorted(g.result())
    for fn in (comm.exchange_allgather, comm.exchange_alltoall):
        with pytest.raises(T.DbhipError) as e:
            fn(g, max_rows=256)
        assert e.value.code == T.ERR_CAPACITY
        assert sorted(g.result()) == before
    comm.exchange_alltoall(g, max_rows=2048)
    assert sorted(g.result()) == before
    # plain collectives: identities in a world of one
    x = np.arange(1000, dtype=np.uint64) * np.uint64(977)
    send, recv = D.DeviceBuffer.from_numpy(x), D.DeviceBuffer(8000 + 64)
    for op in ("allgather", "alltoall"):
        recv.zero()
        getattr(comm, op)(send.ptr, recv.ptr, 8000)
        assert np.array_equal(recv.to_numpy(np.uint64, 1000), x)
    recv.zero()
    comm.allreduce_sum_u64(send.ptr, recv.ptr, 1000)
    assert np.array_equal(recv.to_numpy(np.uint64, 1000), x)
    comm.destroy()
