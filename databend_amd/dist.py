"""Multi-GPU exchange for the hash-aggregation path.

One process per GPU. Each rank aggregates its row-range shard into its own device
table; the only data-path collective is the exchange of serialized partial-state rows
(`[keys | hash | state words]`, gb_layout.h) — the RCCL analogue of the reference's
AggregateMeta shuffle (aggregator/aggregate_exchange_injector.rs:57-147):
  * low cardinality (Q1: 4 groups): all-gather of every rank's rows + local merge on
    every rank (one hop over xGMI, a few hundred bytes);
  * high cardinality: rows are routed by `hash % world` (payload.rs:571-577 scatter
    semantics) with all_to_all, so each rank finalises a disjoint key range.
The exchange is expressed against a minimal `comm` interface so that the same code runs
over torch.distributed with the nccl (= RCCL) backend on GPUs and with gloo in the CPU
tests (which inject a stand-in table; the product path always uses the HIP table).
"""
import numpy as np


def route_rows_by_hash(rows, hash_word, world):
    """Split serialized rows [n, W] by hash % world (Payload::scan_hash_partition_transfer)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    if rows.size == 0:
        return [rows.reshape(0, rows.shape[1] if rows.ndim == 2 else 0) for _ in range(world)]
    dest = (rows[:, hash_word] % np.uint64(world)).astype(np.int64)
    return [rows[dest == r] for r in range(world)]


def allgather_rows(rows, dist, torch, device):
    """All-gather variable-length row sets; returns the concatenation over ranks (rank order)."""
    world = dist.get_world_size()
    W = rows.shape[1]
    cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, W), dtype=torch.int64, device=device)
    if rows.shape[0]:
        buf[: rows.shape[0]] = torch.from_numpy(rows.view(np.int64)).to(device)
    outs = [torch.zeros((mx, W), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [o[:c].cpu().numpy().view(np.uint64) for o, c in zip(outs, counts)]
    return np.concatenate(parts, axis=0) if parts else rows


def alltoall_rows(parts, dist, torch, device):
    """parts[r] = rows destined to rank r. Returns the rows this rank receives."""
    world = dist.get_world_size()
    W = parts[0].shape[1]
    send_cnt = torch.tensor([p.shape[0] for p in parts], dtype=torch.int64, device=device)
    recv_cnt = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(recv_cnt, send_cnt)
    rc = [int(x) for x in recv_cnt.tolist()]
    send = torch.from_numpy(np.concatenate(parts, axis=0).view(np.int64)).to(device) if sum(p.shape[0] for p in parts) else torch.zeros((0, W), dtype=torch.int64, device=device)
    recv = torch.zeros((sum(rc), W), dtype=torch.int64, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=[p.shape[0] for p in parts])
    return recv.cpu().numpy().view(np.uint64)


def exchange_partials(table, dist, torch, device, mode="allgather", hash_word=None):
    """Merge every rank's partial groups. `table` needs flush_serialized()/reset()/merge_serialized().
    allgather: every rank ends with the global result. alltoall: rank r ends with the groups whose
    hash % world == r."""
    rows = table.flush_serialized()
    if mode == "allgather":
        allrows = allgather_rows(rows, dist, torch, device)
    else:
        allrows = alltoall_rows(route_rows_by_hash(rows, hash_word, dist.get_world_size()), dist, torch, device)
    table.reset()
    table.merge_serialized(allrows)
    return table


def exchange_partials_fixed(table, dist, torch, device, max_rows=256):
    """Low-cardinality exchange in ONE fixed-size collective: every rank contributes a [max_rows + 1, W] block
    (row 0 carries its row count), all-gathered in a single call, merged locally — no size negotiation, no per-rank
    host synchronisation (the variable-length path above costs two collectives and a host sync per rank, which is
    what limits weak scaling of a 0.75 ms step). Falls back to `exchange_partials` when a rank holds more rows."""
    rows = np.ascontiguousarray(table.flush_serialized(), dtype=np.uint64)
    world = dist.get_world_size()
    n = rows.shape[0]
    W = rows.shape[1] if rows.ndim == 2 and rows.shape[1] else 1
    block = np.zeros((max_rows + 1, W), dtype=np.int64)
    block[0, 0] = min(n, max_rows + 1)
    if 0 < n <= max_rows:
        block[1:1 + n] = rows.view(np.int64)
    if n > max_rows:
        block[0, 0] = -1
    send = torch.from_numpy(block).to(device)
    recv = torch.empty((world * (max_rows + 1), W), dtype=torch.int64, device=device)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(recv, send)
    else:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        recv = torch.cat(parts, dim=0)
    got = recv.cpu().numpy().reshape(world, max_rows + 1, W)
    if (got[:, 0, 0] < 0).any():  # some rank overflowed the fixed block: every rank sees it and takes the general path together
        return _exchange_rows(table, rows, dist, torch, device)
    allrows = np.concatenate([got[r, 1:1 + int(got[r, 0, 0])] for r in range(world)], axis=0).view(np.uint64)
    table.reset()
    table.merge_serialized(allrows)
    return table


def _exchange_rows(table, rows, dist, torch, device):
    allrows = allgather_rows(rows, dist, torch, device)
    table.reset()
    table.merge_serialized(allrows)
    return table


def exchange_partials_device(table, dist, torch, device, max_rows=256, stream=None, lib_sync=None, capacity_error=None):
    """Device-resident low-cardinality exchange (bench.py --gpus N): the table writes its block (header row + serialized
    rows, dbhip_groupby_flush_block — no host synchronisation), ONE all_gather_into_tensor of equal-size blocks runs
    over RCCL, and every rank merges the other ranks' blocks straight from the gathered tensor
    (dbhip_groupby_merge_blocks; its own states never leave the table). No row crosses PCIe.
    `stream`: the stream handle passed to the library calls. With torch's current stream the step is ordered on one
    stream together with the collective; with None (the library's own stream) `lib_sync()` must drain that stream
    before the collective and torch's stream is drained before the merge.
    A block that overflowed `max_rows` is seen by EVERY rank in the gathered headers (merge_blocks reports it before
    touching the table, `capacity_error(exc)` recognises the error): all ranks then take the variable-length path."""
    world, rank = dist.get_world_size(), dist.get_rank()
    W = table.row_bytes() // 8
    send = torch.empty((max_rows + 1, W), dtype=torch.int64, device=device)
    table.flush_block(send.data_ptr(), max_rows, stream)
    if stream is None and lib_sync is not None:
        lib_sync()
    recv = torch.empty((world * (max_rows + 1), W), dtype=torch.int64, device=device)   # rank-major concatenation
    dist.all_gather_into_tensor(recv, send)
    if stream is None and device.type == "cuda":
        torch.cuda.current_stream().synchronize()
    try:
        table.merge_blocks(recv.data_ptr(), world, max_rows, rank, stream)
    except Exception as e:  # noqa: BLE001 — only the overflow report is handled, everything else propagates
        if capacity_error is None or not capacity_error(e):
            raise
        return exchange_partials(table, dist, torch, device, mode="allgather")
    return table


def exchange_partials_nccl(table, dist, torch, stream=None, lib_sync=None):
    from ._lib import DbhipError, ERR_CAPACITY
    return exchange_partials_device(table, dist, torch, torch.device("cuda", torch.cuda.current_device()), stream=stream,
                                    lib_sync=lib_sync, capacity_error=lambda e: isinstance(e, DbhipError) and e.code == ERR_CAPACITY)


def exchange_partials_alltoall_device(table, dist, torch, device, max_rows=256, stream=None, lib_sync=None, capacity_error=None):
    """Device-resident hash-partitioned exchange (BASELINE configs[3]; the RCCL analogue of the reference's
    scatter by `hash % n` into the Flight exchange, payload.rs:548-589 + aggregate_exchange_injector.rs:57-147):
    the table routes every group row to bucket hash % world on the device (dbhip_groupby_partition_blocks: `world`
    fixed-size blocks, no host synchronisation), ONE all_to_all_single with equal splits moves block r to rank r, and
    the receiver replaces its table with the merge of the blocks it got (dbhip_groupby_replace_with_blocks). Afterwards
    rank r holds exactly the groups with hash % world == r, each merged over all ranks. No row crosses PCIe.
    A sender that overflowed `max_rows` flags it in ALL its blocks, so every rank sees the same flags and all take the
    variable-length path together (`capacity_error(exc)` recognises the report)."""
    world = dist.get_world_size()
    W = table.row_bytes() // 8
    send = torch.empty((world * (max_rows + 1), W), dtype=torch.int64, device=device)
    table.partition_blocks(send.data_ptr(), world, max_rows, stream)
    if stream is None and lib_sync is not None:
        lib_sync()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    if stream is None and device.type == "cuda":
        torch.cuda.current_stream().synchronize()
    try:
        table.replace_with_blocks(recv.data_ptr(), world, max_rows, stream)
    except Exception as e:  # noqa: BLE001 — only the overflow report is handled, everything else propagates
        if capacity_error is None or not capacity_error(e):
            raise
        return exchange_partials_alltoall_variable(table, dist, torch, device, stream=stream, lib_sync=lib_sync)
    return table


def exchange_partials_alltoall_variable(table, dist, torch, device, stream=None, lib_sync=None):
    """High-cardinality form of the same exchange: rows grouped by destination on the device
    (dbhip_groupby_flush_partitioned -> the split sizes), one all_to_all_single for the counts and one for the rows, both on
    device tensors; the received rows are merged straight from the receive buffer."""
    world = dist.get_world_size()
    W = table.row_bytes() // 8
    g = table.num_groups()
    send = torch.empty((max(g, 1), W), dtype=torch.int64, device=device)
    counts = table.flush_partitioned(world, send.data_ptr(), g, stream)   # synchronises: the counts are host values
    send_cnt = torch.tensor(counts, dtype=torch.int64, device=device)
    recv_cnt = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(recv_cnt, send_cnt)
    rc = [int(x) for x in recv_cnt.tolist()]
    recv = torch.empty((max(sum(rc), 1), W), dtype=torch.int64, device=device)
    dist.all_to_all_single(recv[: sum(rc)], send[: sum(counts)], output_split_sizes=rc, input_split_sizes=counts)
    if device.type == "cuda":
        torch.cuda.current_stream().synchronize()
    table.reset(stream)
    table.merge_serialized_device(recv.data_ptr(), sum(rc), stream)
    return table


def exchange_partials_alltoall_nccl(table, dist, torch, stream=None, lib_sync=None, max_rows=256):
    from ._lib import DbhipError, ERR_CAPACITY
    return exchange_partials_alltoall_device(table, dist, torch, torch.device("cuda", torch.cuda.current_device()), max_rows=max_rows,
                                             stream=stream, lib_sync=lib_sync,
                                             capacity_error=lambda e: isinstance(e, DbhipError) and e.code == ERR_CAPACITY)


def _check_global_ids(row_offset, max_local_id):
    """global row ids are u32 and 0xFFFFFFFF is the 'empty' sentinel: the largest id a shard can produce must stay below it"""
    if int(row_offset) < 0 or int(row_offset) + int(max_local_id) >= 0xFFFFFFFF:
        raise ValueError(f"global row id {int(row_offset) + int(max_local_id)} does not fit the u32 id space below the empty sentinel "
                         f"(row_offset {int(row_offset)}): shard the index over more ranks or widen the ids")


def merge_shard_topk(idx, dst, row_offset, k, dist, torch, device, merge_fn):
    """ANN over a row-range sharded base: queries are replicated, every rank searched its shard and
    holds `idx` (u32 local row ids, 0xFFFFFFFF = empty) / `dst` (f32) of shape [nq, k]. One all-gather of the
    (k ids + k distances) per query over RCCL, then the k-way merge `merge_fn(dists [nq, world*k], ids) ->
    (idx, dist)` (the device select kernel, dbhip_vec_topk_merge). Returns global row ids."""
    world = dist.get_world_size()
    _check_global_ids(row_offset, idx[idx != 0xFFFFFFFF].max() if (idx != 0xFFFFFFFF).any() else 0)
    gid = np.where(idx == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), (idx.astype(np.int64) + int(row_offset)).astype(np.uint32))
    ti = torch.from_numpy(gid.view(np.int32)).to(device)
    td = torch.from_numpy(np.ascontiguousarray(dst, dtype=np.float32)).to(device)
    gi = [torch.empty_like(ti) for _ in range(world)]
    gd = [torch.empty_like(td) for _ in range(world)]
    dist.all_gather(gi, ti)
    dist.all_gather(gd, td)
    all_i = torch.cat(gi, dim=1).cpu().numpy().view(np.uint32)
    all_d = torch.cat(gd, dim=1).cpu().numpy()
    return merge_fn(all_d, all_i, k)


def merge_shard_topk_device(idx_t, dst_t, row_offset, k, dist, torch, lib_sync, merge_dev):
    """Device-resident variant of merge_shard_topk for the RCCL path: `idx_t` (int32 view of u32 ids, -1 = empty) and
    `dst_t` (f32) are CUDA tensors [nq, k] the library just wrote; ids are globalised with torch arithmetic, both are
    all-gathered with ONE fixed-size collective each and merged by dbhip_vec_topk_merge straight from the gathered
    tensors — no host round trip. `lib_sync()` drains the library's stream (the search wrote the inputs there),
    `merge_dev(d_ptr, i_ptr, nq, m, k, out_i_ptr, out_d_ptr)` calls the C-ABI."""
    world = dist.get_world_size()
    nq = idx_t.shape[0]
    lib_sync()
    # ids are u32 (0xFFFFFFFF = empty) carried in an int32 view: globalise in int64 (an offset >= 2^31 is a legal u32 id but not a
    # legal int32 scalar), check the u32 id space, and go back to the int32 view
    wide = idx_t.to(torch.int64) & 0xFFFFFFFF
    live = idx_t != -1
    _check_global_ids(row_offset, int(wide[live].max()) if bool(live.any()) else 0)
    g64 = wide + int(row_offset)
    gid = torch.where(live, torch.where(g64 >= (1 << 31), g64 - (1 << 32), g64).to(torch.int32), idx_t).contiguous()
    gi = torch.empty((world * nq, k), dtype=torch.int32, device=idx_t.device)   # rank-major concatenation
    gd = torch.empty((world * nq, k), dtype=torch.float32, device=idx_t.device)
    dist.all_gather_into_tensor(gi, gid)
    dist.all_gather_into_tensor(gd, dst_t.contiguous())
    all_i = gi.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).contiguous()
    all_d = gd.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).contiguous()
    out_i = torch.empty((nq, k), dtype=torch.int32, device=idx_t.device)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=idx_t.device)
    torch.cuda.current_stream().synchronize()   # the gathered tensors are complete before the library reads them
    merge_dev(all_d.data_ptr(), all_i.data_ptr(), nq, world * k, k, out_i.data_ptr(), out_d.data_ptr())
    lib_sync()
    return out_i, out_d


# ---------------------------------------------------------------------------------------------------------------------
# Broadcast hash join across ranks (TPC-H Q3, BASELINE configs[2] on N GPUs).
# The reference runs a distributed inner join either by scattering both sides by the hash of the join key
# (flight_scatter_hash.rs:57-120: hash % n per row, one flight per destination) or, when the build side is small,
# by BROADCASTING the build side to every node (flight_scatter_broadcast.rs:22-37: every destination gets the whole
# block; the planner picks it in hash_join.rs/`broadcast` exchanges) so that the probe side never moves. Q3's build
# sides — customers of one market segment, then the orders that joined them — are 50x / 40x smaller than the probe sides
# (orders, lineitem), which is the broadcast case: every rank filters its shard of the build side, the survivors are
# all-gathered (ONE variable-length all-gather of a packed image over RCCL / xGMI), every rank builds the SAME table and probes
# its own rows. Only the final aggregation exchanges states (route by hash % world, like configs[3]).
# ---------------------------------------------------------------------------------------------------------------------
def allgather_columns(cols, dist, torch):
    """Variable-length all-gather of equally long 1-D tensors (one build-side block per rank): returns the rank-major
    concatenation of every column. One small collective for the lengths, then ONE padded all_gather_into_tensor of a packed image
    (every column padded to the longest rank's rows and to 8 bytes, back to back) whatever the number of columns."""
    world = dist.get_world_size()
    n = int(cols[0].shape[0])
    dev = cols[0].device
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt)
    counts = [int(x) for x in cnts.tolist()]
    mx = max(max(counts), 1)
    isz = [c.element_size() for c in cols]
    col_bytes = [(mx * z + 7) & ~7 for z in isz]
    img = sum(col_bytes)
    send = torch.zeros(img, dtype=torch.uint8, device=dev)
    off = 0
    for c, z, cb in zip(cols, isz, col_bytes):
        assert int(c.shape[0]) == n and c.dim() == 1
        if n:
            send[off:off + n * z] = c.contiguous().view(torch.uint8)
        off += cb
    recv = torch.empty(world * img, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    out = []
    off = 0
    for c, z, cb in zip(cols, isz, col_bytes):
        parts = [recv[r * img + off: r * img + off + counts[r] * z].view(c.dtype) for r in range(world)]
        out.append(torch.cat(parts) if world > 1 else parts[0].clone())
        off += cb
    return out


def q3_broadcast_join(shard, ops, dist, torch, device, limit=10):
    """Distributed Q3 over row-range shards of the three tables (any partition of the rows: the lines of one order may
    live on different ranks). `ops` runs the single-node operators on this rank's shard (databend_amd.tpch.Q3DeviceOps on
    a GPU; a numpy stand-in in the gloo tests):
        ops.filter_customers(shard)                     -> [custkey]                       (1-D tensors on `device`)
        ops.join_orders(custkeys, shard)                -> [orderkey, orderdate, shippriority] of the orders that joined
        ops.aggregate_lineitem(okey, odate, oprio, shard) -> partial-aggregation table keyed by the three group columns
        ops.exchange(table, dist, device)               -> routes every group to rank hash % world and merges (the configs[3]
                                                          exchange: exchange_partials_alltoall_variable on a GPU)
        ops.top_rows(table, limit)                      -> [(l_orderkey, revenue, o_orderdate, o_shippriority)] sorted
    Every rank returns the same global top `limit` rows."""
    (all_ck,) = allgather_columns(ops.filter_customers(shard), dist, torch)                  # broadcast build side #1
    okey, odate, oprio = allgather_columns(ops.join_orders(all_ck, shard), dist, torch)      # broadcast build side #2
    table = ops.aggregate_lineitem(okey, odate, oprio, shard)
    ops.exchange(table, dist, device)                                                        # disjoint groups per rank
    mine = ops.top_rows(table, limit)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, mine)                                                  # <= limit small rows per rank
    rows = [r for part in gathered for r in part]
    rows.sort(key=lambda r: (-r[1], r[2], r[0]))
    return rows[:limit] if limit else rows


# ---------------------------------------------------------------------------------------------------------------------
# Distributed sort (SURVEY §8e "sort": sample -> range-partition -> all-to-all -> local sort). The reference sorts a table
# across nodes by cutting every node's rows at the same global Bounds (sorts/sort_broadcast.rs:150-197: each node merges its
# samples into Bounds, broadcasts them, merges + dedups what it received; sorts/sort_spill.rs:740-1040 cuts the sorted streams
# at each bound: rows <= bound[i] belong to range i; service/.../sort/sort_exchange_injector.rs SortBoundScatter sends range i
# to node i % n) and merges the ranges it receives. Device plan: the rows are NOT sorted before the exchange (one sort per row
# instead of sort + merge) — every rank takes a strided sample of its key columns, the samples are all-gathered, ordered and
# deduplicated into world - 1 bounds (sort_bounds.balanced_cuts: equal shares of the samples, so rank r receives exactly range
# r — the fewest, largest messages on xGMI; the reference keeps every distinct sample as a bound and deals the ranges round
# robin), one kernel gives every row its range (dbhip_sort_bound_partition), the columns are grouped by range and
# exchanged with ONE all_to_all_single of a packed row image (alltoall_columns), and each rank sorts what it received. The concatenation of the ranks'
# outputs in rank order is the sorted table.
# ---------------------------------------------------------------------------------------------------------------------
def alltoall_columns(cols, send_counts, dist, torch):
    """cols = equally long 1-D tensors whose rows are grouped by destination rank (send_counts[r] rows for rank r).
    ONE collective for the counts and ONE for the data of all columns: what goes to rank r is packed as a row image (every column's
    slice for r, each padded to 8 bytes, back to back), so the width of the block does not multiply the number of RCCL launches."""
    world = dist.get_world_size()
    dev = cols[0].device
    sc = [int(c) for c in send_counts]
    send_cnt = torch.tensor(sc, dtype=torch.int64, device=dev)
    recv_cnt = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_cnt, send_cnt)
    rc = [int(x) for x in recv_cnt.tolist()]
    isz = [c.element_size() for c in cols]
    pad8 = lambda nb: (nb + 7) & ~7
    seg_bytes = lambda rows: sum(pad8(rows * z) for z in isz)
    send = torch.zeros(max(sum(seg_bytes(r) for r in sc), 8), dtype=torch.uint8, device=dev)
    off, start = 0, 0
    for r in range(world):
        for c, z in zip(cols, isz):
            nb = sc[r] * z
            if nb:
                send[off:off + nb] = c[start:start + sc[r]].contiguous().view(torch.uint8)
            off += pad8(nb)
        start += sc[r]
    in_split = [seg_bytes(r) for r in sc]
    out_split = [seg_bytes(r) for r in rc]
    recv = torch.empty(max(sum(out_split), 8), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv[:sum(out_split)], send[:sum(in_split)], output_split_sizes=out_split, input_split_sizes=in_split)
    pieces = [[] for _ in cols]
    off = 0
    for r in range(world):
        for k, (c, z) in enumerate(zip(cols, isz)):
            nb = rc[r] * z
            pieces[k].append(recv[off:off + nb].view(c.dtype) if nb else torch.empty(0, dtype=c.dtype, device=dev))
            off += pad8(nb)
    return [torch.cat(p) if len(p) > 1 else p[0].clone() for p in pieces], rc


def range_partitioned_sort(cols, keys, ops, dist, torch, desc=None, nulls_first=None, valids=None, samples_per_rank=1024):
    """Sort the table whose rows are spread over the ranks. cols = this rank's rows as equally long 1-D tensors; keys = indices
    of the sort key columns, most significant first; desc / nulls_first per key (SortColumnDescription); valids[c] = None or a
    uint8 tensor (1 = the value is not NULL). `ops` runs the single-node operators (databend_amd.sort_ops.SortDeviceOps on a
    GPU; a numpy stand-in in the gloo tests):
        ops.ordered_rows(key_cols, key_valids, desc, nulls_first)          -> the rows as host tuples (None = NULL) in sort order
        ops.partition(flat, kpos, kvpos, bounds, desc, nulls_first)        -> (flat grouped by range, rows per range)
        ops.sort(flat, kpos, kvpos, desc, nulls_first)                     -> flat in sort order
    Returns (cols, valids, bounds): rank r's rows are range r of the global order, sorted; bounds = the global cut rows."""
    from .sort_bounds import balanced_cuts
    world = dist.get_world_size()
    nk = len(keys)
    desc = list(desc or [0] * nk)
    nulls_first = list(nulls_first or [0] * nk)
    valids = list(valids or [None] * len(cols))
    n = int(cols[0].shape[0])
    dev = cols[0].device
    flat, vpos = list(cols), [None] * len(cols)
    for c, v in enumerate(valids):
        if v is not None:
            vpos[c] = len(flat)
            flat.append(v)
    kpos, kvpos = list(keys), [vpos[k] for k in keys]
    cnt = min(n, int(samples_per_rank))
    if cnt:
        i = torch.arange(cnt, dtype=torch.int64, device=dev)
        ids = (i * n) // cnt + n // (2 * cnt)
    else:
        ids = torch.zeros(0, dtype=torch.int64, device=dev)
    sample_pos = kpos + [p for p in kvpos if p is not None]
    gathered = allgather_columns([flat[p][ids] for p in sample_pos], dist, torch)            # every node's samples, everywhere
    g_keys = gathered[:nk]
    g_valids, at = [], nk
    for p in kvpos:
        g_valids.append(gathered[at] if p is not None else None)
        at += 1 if p is not None else 0
    rows = ops.ordered_rows(g_keys, g_valids, desc, nulls_first)
    bound_rows = balanced_cuts(rows, world)                                                   # the same on every rank
    grouped, counts = ops.partition(flat, kpos, kvpos, bound_rows, desc, nulls_first)
    counts = [int(c) for c in counts] + [0] * (world - len(counts))                          # range i -> rank i
    received, _rc = alltoall_columns(grouped, counts, dist, torch)
    out = ops.sort(received, kpos, kvpos, desc, nulls_first)
    out_cols = out[:len(cols)]
    out_valids = [out[p] if p is not None else None for p in vpos]
    return out_cols, out_valids, bound_rows


# ---------------------------------------------------------------------------------------------------------------------
# Shuffle (co-partitioned) hash join (SURVEY §8e "hash join": "else co-partition both sides by hash(key) % n"). The reference
# scatters BOTH sides of a join by `siphash64(key) % n` (HashFlightScatter / OneHashKeyFlightScatter,
# servers/flight/v1/scatter/flight_scatter_hash.rs:57-330) so that equal keys meet on one node, and joins locally. Device plan:
# dbhip_scatter_indices gives every row its destination with the reference's own hash (bit-identical: a GPU rank routes rows
# exactly like a CPU node would), one radix pass + dbhip_take_block groups the columns by destination (DataBlock::scatter), one
# all_to_all_single of a packed row image (alltoall_columns) moves them, and each rank joins what it received. Used when the build side is too large to
# broadcast (dist.q3_broadcast_join is the other plan).
# ---------------------------------------------------------------------------------------------------------------------
def shuffle_hash_join(build_cols, build_key, probe_cols, probe_key, ops, dist, torch, build_valids=None, probe_valids=None):
    """Inner join of two tables whose rows are spread over the ranks. *_cols = this rank's rows as equally long 1-D tensors,
    *_key = index of the (integer) join key column, *_valids[c] = None or a uint8 tensor (1 = not NULL; NULL keys never match
    and travel to rank 0, the reference's default scatter index). `ops`:
        ops.scatter(flat, kpos, kvpos, world)         -> (flat grouped by destination, rows per destination)
        ops.join(build_flat, bk, bkv, probe_flat, pk, pkv) -> the joined rows: probe columns ++ build columns (flat lists)
    Returns (probe columns, build columns) of the pairs this rank produced; the union over the ranks is the join."""
    world = dist.get_world_size()

    def flatten(cols, valids):
        valids = list(valids or [None] * len(cols))
        flat, vpos = list(cols), [None] * len(cols)
        for c, v in enumerate(valids):
            if v is not None:
                vpos[c] = len(flat)
                flat.append(v)
        return flat, vpos

    bflat, bv = flatten(build_cols, build_valids)
    pflat, pv = flatten(probe_cols, probe_valids)
    sides = []
    for flat, key, vpos in ((bflat, build_key, bv), (pflat, probe_key, pv)):
        grouped, counts = ops.scatter(flat, key, vpos[key], world)
        received, _ = alltoall_columns(grouped, [int(c) for c in counts], dist, torch)
        sides.append(received)
    out_p, out_b = ops.join(sides[0], build_key, bv[build_key], sides[1], probe_key, pv[probe_key])
    return out_p, out_b


# ---------------------------------------------------------------------------------------------------------------------
# Aggregation WITHOUT keys across ranks (SURVEY §8e "`sum` without keys": BASELINE configs[0] `SELECT sum(a + b * c)` and every
# other single-state aggregate). The reference's PartialSingleStateAggregator hands one serialized state per thread / node to the
# FinalSingleStateAggregator, which merges them one after the other (transform_single_key.rs:43-279). Here every rank holds ONE
# state per aggregate (the device reduced its rows: dbhip_expr_eval's sum_out, dbhip_groupby_* with no key); the states of all
# ranks travel in ONE all-gather of three words per state and are folded IN RANK ORDER on every rank — not an all-reduce, whose
# summation order is the library's: f64 sums are order-dependent (§8a a10) and the fold below is the one order every rank and
# every run agrees on; i128 sums need their carry, which a per-word all-reduce would drop.
# ---------------------------------------------------------------------------------------------------------------------
def merge_single_states(states, dist, torch, device):
    """states = this rank's [(kind, cls, value, has)]: kind in ("sum", "count", "min", "max"); cls in ("i64", "u64", "i128", "f64");
    value = a Python int (two's complement of the class's width is applied) or float; has = saw a non-NULL row (count: ignored).
    Returns the merged [(value, has)] — identical on every rank; sums wrap at the class's width like the reference's states
    (aggregate_sum.rs:113-129; Decimal overflow is decided on the final total by the caller, as for the hash table)."""
    import struct
    world = dist.get_world_size()
    M64, M128 = (1 << 64) - 1, (1 << 128) - 1

    def s64(x):
        return x - (1 << 64) if x >> 63 else x

    words = []
    for kind, cls, value, has in states:
        if cls == "f64":
            lo, hi = struct.unpack("<Q", struct.pack("<d", float(value)))[0], 0
        else:
            v = int(value) & (M128 if cls == "i128" else M64)
            lo, hi = v & M64, v >> 64
        words += [s64(lo), s64(hi), 1 if (has or kind == "count") else 0]
    mine = torch.tensor(words, dtype=torch.int64, device=device)
    gathered = torch.empty(world * max(len(words), 1), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(gathered[: world * len(words)] if words else gathered, mine if words else torch.zeros(1, dtype=torch.int64, device=device))
    rows = gathered[: world * len(words)].view(world, -1).tolist() if words else []
    out = []
    for i, (kind, cls, _value, _has) in enumerate(states):
        acc, acc_has = None, False
        for r in range(world):
            lo, hi, has = rows[r][3 * i] & M64, rows[r][3 * i + 1] & M64, rows[r][3 * i + 2]
            if not has:
                continue
            if cls == "f64":
                v = struct.unpack("<d", struct.pack("<Q", lo))[0]
            else:
                v = lo | (hi << 64)
                width = 128 if cls == "i128" else 64
                if cls != "u64" and v >> (width - 1):
                    v -= 1 << width
            if acc is None:
                acc = v
            elif kind in ("sum", "count"):
                acc = acc + v
                if cls == "i64":
                    acc = s64(acc & M64)
                elif cls == "u64":
                    acc &= M64
                elif cls == "i128":
                    acc &= M128
                    acc = acc - (1 << 128) if acc >> 127 else acc
            elif kind == "min":
                acc = v if v < acc else acc
            else:
                acc = v if v > acc else acc
            acc_has = True
        out.append(((0.0 if cls == "f64" else 0) if acc is None else acc, acc_has))
    return out
