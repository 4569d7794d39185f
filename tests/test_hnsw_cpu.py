"""CPU: the HNSW / u8-quantiser restatement (oracle/hnsw_oracle.c) against the reference's own known answers and against
independent statements of the same algorithms."""
import json
import os

import numpy as np
import pytest

from tests import hnsw_oracle as H

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def L():
    return H.lib()


def block_results(L, golden, q, level_seed):
    """what the reference's read path produces for an indexed table, block by block (vector_index_reader.rs:55-71):
    ORDER BY distance ASC LIMIT k without a filter -> HNSWIndex::search(k) on the block's graph (m=10, ef_construct=40, ef=4k);
    otherwise HNSWIndex::generate_scores (every row). Graph levels: the reference draws them from thread_rng(), so its
    expected answers cannot depend on them — the caller tries several seeds."""
    out = []
    col = 0 if q["column"] in ("embedding", "embedding1") else 1
    use_search = q["order"] == "ASC" and q["where_gt"] is None
    rng = np.random.default_rng(level_seed)
    for block in golden["tables"][q["table"]]:
        vecs = np.array([r["vectors"][col] for r in block], dtype=np.float32)
        data = H.preprocess(L, vecs, q["distance"])
        quant = H.Quantised(L, data, q["distance"])
        query = H.preprocess(L, np.array([q["query"]], dtype=np.float32), q["distance"])[0]   # preprocess_query, hnsw.rs:307-312
        if use_search:
            levels = H.random_levels(len(block), 10, rng) if level_seed else np.zeros(len(block), dtype=np.int32)
            g = H.Graph(L, len(block), 10, 40, levels)
            g.build(vecs, q["distance"])            # the ORIGINAL column scores the build (hnsw.rs:226-232)
            ids, d = g.search(quant, query, q["limit"])
            g.free()
            out += [(block[i]["id"], np.float32(x)) for i, x in zip(ids.tolist(), d)]
        else:
            d = quant.distances(query)
            out += [(r["id"], np.float32(x)) for r, x in zip(block, d)]
    return out


@pytest.mark.parametrize("level_seed", [0, 1, 2, 3, 4])
def test_index_answers_equal_the_reference_sqllogictest(L, level_seed):
    """09_0000_vector_index_base.test: tables t / t_native / t2 answer through HNSWIndex; the printed distances are
    post-processed quantised scores. Printed f32 values round-trip: equality is exact (1 ulp allowed)."""
    g = json.load(open(os.path.join(HERE, "golden", "hnsw.json")))
    checked = 0
    for q in g["queries"]:
        if q["table"] not in g["indexed_tables"]:
            continue
        rows = block_results(L, g, q, level_seed)
        if q["where_gt"] is not None:
            rows = [r for r in rows if r[1] > np.float32(q["where_gt"])]
        rows.sort(key=lambda r: (r[1], r[0]), reverse=q["order"] == "DESC")
        got = rows[:q["limit"]]
        exp = [(i, np.float32(d)) for i, d in q["expected"]]
        assert [r[0] for r in got] == [r[0] for r in exp], (q, got, exp)
        for (gi, gd), (ei, ed) in zip(got, exp):
            assert gd == ed or abs(float(gd) - float(ed)) <= 1.2e-7 * max(1.0, abs(float(ed))), (q["distance"], gi, gd, ed)
        checked += 1
    assert checked == 11


def np_encode(data, distance):
    """an independent numpy statement of EncodedVectorsU8::encode (encoded_vectors_u8.rs:54-161)"""
    f = np.float32
    mn, mx = f(data.min()), f(data.max())
    alpha = f((mx - mn) / f(127.0))
    offset = mn
    dim = data.shape[1]
    ad = dim + (16 - dim % 16) % 16
    codes = np.clip(np.trunc(((data - offset) / alpha).astype(np.float32)), 0, 127).astype(np.uint8)
    pad = f(0.0) if distance == "cosine" else offset
    padcode = np.uint8(np.clip(np.trunc(f((pad - offset) / alpha)), 0, 127))
    codes = np.concatenate([codes, np.full((data.shape[0], ad - dim), padcode, dtype=np.uint8)], axis=1)
    s1 = codes.astype(np.float64).sum(axis=1).astype(np.float32)           # exact: integers < 2^24
    s2 = (codes.astype(np.float64) ** 2).sum(axis=1).astype(np.float32)
    if distance == "cosine":
        vo = (f(ad) * offset * offset + s1 * alpha * offset).astype(np.float32)
        mult = alpha * alpha
    elif distance == "l1":
        vo = np.zeros(data.shape[0], dtype=np.float32)
        mult = -alpha
    else:
        vo = -((f(ad) * offset * offset) + s2 * alpha * alpha).astype(np.float32)
        mult = f(2.0) * alpha * alpha
    return alpha, offset, f(mult), codes, vo


@pytest.mark.parametrize("distance", ["cosine", "l1", "l2"])
@pytest.mark.parametrize("dim", [8, 16, 37, 768])
def test_encode_against_a_numpy_statement(L, distance, dim):
    rng = np.random.default_rng(7 + dim)
    raw = rng.standard_normal((53, dim)).astype(np.float32)
    data = H.preprocess(L, raw, distance)
    if distance == "cosine":
        ref = (raw / np.sqrt((raw.astype(np.float64) ** 2).sum(axis=1, keepdims=True))).astype(np.float32)
        assert np.allclose(data, ref, rtol=3e-6, atol=1e-7)
    quant = H.Quantised(L, data, distance)
    alpha, offset, mult, codes, vo = np_encode(data, distance)
    m = quant.meta
    assert (np.float32(m.alpha), np.float32(m.offset), np.float32(m.multiplier)) == (alpha, offset, mult)
    enc = quant.encoded.reshape(53, m.actual_dim + 4)
    assert np.array_equal(enc[:, 4:], codes)
    got_vo = enc[:, :4].copy().view(np.float32).ravel()
    assert np.array_equal(got_vo, vo)
    # score_point vs the definition: multiplier * sum(q * v) + query offset + vector offset
    q = H.preprocess(L, rng.standard_normal((1, dim)).astype(np.float32), distance)[0]
    qc, qo = quant.encode_query(q)
    raw_s = (np.abs(qc.astype(np.int64) - codes.astype(np.int64)).sum(axis=1) if distance == "l1"
             else (qc.astype(np.int64) * codes.astype(np.int64)).sum(axis=1)).astype(np.float32)
    score = (mult * raw_s + qo + vo).astype(np.float32)
    post = {"cosine": lambda s: np.abs(np.float32(1) - s), "l1": np.abs, "l2": lambda s: np.sqrt(np.abs(s))}[distance](score).astype(np.float32)
    assert np.array_equal(quant.distances(q), post)
    # and the quantised distance approximates the exact one
    exact = {"cosine": lambda: 1.0 - data @ q, "l1": lambda: np.abs(data - q).sum(axis=1), "l2": lambda: np.sqrt(((data - q) ** 2).sum(axis=1))}[distance]()
    # (the reference's L2 vector offset carries a constant actual_dim * offset^2, encoded_vectors_u8.rs:125-133, which the
    # expansion of |q - v|^2 does not contain: its L2 "distance" is sqrt(d^2 + actual_dim * offset^2) — order-preserving)
    if distance == "l2":
        post = np.sqrt(np.maximum(post.astype(np.float64) ** 2 - m.actual_dim * float(offset) ** 2, 0.0))
    if distance == "cosine" and dim % 16:
        return   # padded dims decode to `offset`, not 0 (:107-113): the quantised dot is biased by the padding, as in the reference
    tol = {"cosine": 0.02, "l2": 0.35, "l1": 0.02 * dim + 0.5}[distance]
    assert np.abs(post - exact).max() < tol, np.abs(post - exact).max()


def exact_topk(data, q, k, distance):
    d = {"cosine": lambda: 1.0 - data @ q, "l1": lambda: np.abs(data - q).sum(axis=1), "l2": lambda: ((data - q) ** 2).sum(axis=1)}[distance]()
    return np.argsort(d, kind="stable")[:k]


@pytest.mark.parametrize("distance", ["cosine", "l2"])
def test_sequential_build_has_the_structure_and_recall_of_hnsw(L, distance):
    """GraphLayersBuilder restated (levels seeded): link counts within m / m0, no self links, every point reachable from the
    entry point on level 0, the entry point is the first point of the highest level, and searching the quantised index
    with ef = 4k finds the exact neighbours (recall@10 >= 0.9 on 2000 x 32 clustered vectors)."""
    rng = np.random.default_rng(3)
    n, dim, m = 2000, 32, 10
    centers = rng.standard_normal((20, dim)).astype(np.float32) * 2
    raw = (centers[rng.integers(0, 20, n)] + rng.standard_normal((n, dim))).astype(np.float32)
    if distance == "cosine":   # the build scores RAW dot products (hnsw.rs:226-232): keep the norms comparable, as BASELINE's N(0,1) data does
        raw = (raw / np.linalg.norm(raw, axis=1, keepdims=True) * rng.uniform(0.97, 1.03, (n, 1))).astype(np.float32)
    levels = H.random_levels(n, m, rng)
    g = H.Graph(L, n, m, 40, levels)
    g.build(raw, distance)                       # the reference scores the ORIGINAL column while building (hnsw.rs:226-232)
    ep, lv = g.entry()
    assert lv == levels.max() and ep == int(np.argmax(levels == levels.max()))
    seen = {ep}
    for p in range(n):
        for level in range(levels[p] + 1):
            l = g.links(p, level)
            assert len(l) <= (2 * m if level == 0 else m) and p not in l and len(set(l.tolist())) == len(l)
            assert all(levels[x] >= level for x in l)
    frontier = [ep]
    while frontier:
        nxt = []
        for p in frontier:
            for x in g.links(p, 0).tolist():
                if x not in seen:
                    seen.add(x)
                    nxt.append(x)
        frontier = nxt
    # (links are directed and pruned by the heuristic: HNSW does not promise that every point is reachable)
    assert len(seen) >= 0.97 * n, len(seen)
    data = H.preprocess(L, raw, distance)
    quant = H.Quantised(L, data, distance)
    hits = 0
    queries = H.preprocess(L, (centers[rng.integers(0, 20, 50)] + rng.standard_normal((50, dim))).astype(np.float32), distance)
    for q in queries:
        ids, dist = g.search(quant, q, 10)
        assert len(ids) == 10 and np.all(np.diff(dist) >= 0)
        hits += len(set(ids.tolist()) & set(exact_topk(data, q, 10, distance).tolist()))
    print(distance, 'reachable', len(seen), 'recall@10', hits / 500)
    assert hits / 500 >= 0.85, hits / 500
    g.free()


def test_search_on_a_hand_made_graph(L):
    """a 6-point line graph on level 0 with a 2-level entry: greedy descent, then the beam; ties in the quantised score are
    resolved by the heap order restated from std::collections::BinaryHeap"""
    data = np.array([[0.0, 0], [1, 0], [2, 0], [3, 0], [4, 0], [5, 0]], dtype=np.float32)
    levels = np.array([0, 0, 1, 0, 0, 1], dtype=np.int32)
    g = H.Graph(L, 6, 10, 40, levels)
    for p in range(6):
        g.set_links(p, 0, [x for x in (p - 1, p + 1) if 0 <= x < 6])
    g.set_links(2, 1, [5])
    g.set_links(5, 1, [2])
    g.set_entry(5, 1)
    quant = H.Quantised(L, data, "l2")
    ids, dist = g.search(quant, np.array([0.9, 0], dtype=np.float32), 3)
    assert ids.tolist() == [1, 0, 2]
    assert np.allclose(dist, [0.1, 0.9, 1.1], atol=0.05)
    g.free()
