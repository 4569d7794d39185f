"""ctypes view of oracle/hnsw_oracle.c (the reference's HNSW + u8 quantiser restated on the CPU) — tests only."""
import ctypes as C

import numpy as np

from tests import oracle_lib as O

DIST = {"cosine": 0, "dot": 0, "l1": 1, "l2": 2}


class Meta(C.Structure):
    _fields_ = [("dim", C.c_int), ("actual_dim", C.c_int), ("distance", C.c_int), ("invert", C.c_int), ("count", C.c_int64),
                ("alpha", C.c_float), ("offset", C.c_float), ("multiplier", C.c_float)]


def lib():
    L = O.load()
    L.orc_u8_encode_query.restype = C.c_float
    L.orc_u8_score_point.restype = C.c_float
    L.orc_hnsw_postprocess.restype = C.c_float
    L.orc_hnsw_graph_new.restype = C.c_void_p
    L.orc_hnsw_graph_new.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p]
    for f in ("orc_hnsw_graph_free", "orc_hnsw_build", "orc_hnsw_graph_links", "orc_hnsw_graph_set_links", "orc_hnsw_graph_set_entry",
              "orc_hnsw_graph_entry", "orc_hnsw_search", "orc_hnsw_link_new_point"):
        getattr(L, f).argtypes = None
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def preprocess(L, vecs, distance):
    """hnsw.rs:345-374: cosine -> every vector normalised (unless its squared length is ~0 or ~1)"""
    v = np.ascontiguousarray(vecs, dtype=np.float32)
    if DIST[distance] != 0:
        return v
    out = np.empty_like(v)
    for i in range(v.shape[0]):
        L.orc_cosine_preprocess(ptr(v[i]), C.c_int(v.shape[1]), ptr(out[i]))
    return out


class Quantised:
    """EncodedVectorsU8 of one block (vectors already pre-processed)"""

    def __init__(self, L, data, distance):
        self.L = L
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        n, dim = self.data.shape
        self.meta = Meta()
        L.orc_u8_params(ptr(self.data), C.c_int64(n), C.c_int(dim), C.c_int(DIST[distance]), C.byref(self.meta))
        self.rec = self.meta.actual_dim + 4
        self.encoded = np.zeros(max(n, 1) * self.rec, dtype=np.uint8)
        L.orc_u8_encode(ptr(self.data), C.byref(self.meta), ptr(self.encoded))

    def encode_query(self, q):
        q = np.ascontiguousarray(q, dtype=np.float32)
        codes = np.zeros(self.meta.actual_dim, dtype=np.uint8)
        off = self.L.orc_u8_encode_query(ptr(q), C.byref(self.meta), ptr(codes))
        return codes, np.float32(off)

    def distances(self, q):
        """generate_scores (hnsw.rs:120-140)"""
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros(max(self.meta.count, 1), dtype=np.float32)
        self.L.orc_hnsw_generate_scores(ptr(self.encoded), C.byref(self.meta), ptr(q), ptr(out))
        return out[:self.meta.count]


def random_levels(n, m, rng):
    """get_random_layer (graph_layers_builder.rs:246-255) with a seeded generator instead of thread_rng()"""
    u = rng.random(n)
    u[u == 0.0] = 0.5
    return np.round(-np.log(u) * (1.0 / np.log(max(m, 2)))).astype(np.int32)


class Graph:
    def __init__(self, L, n, m, ef_construct, levels):
        self.L = L
        self.n, self.m, self.m0 = n, m, 2 * m
        self.levels = np.ascontiguousarray(levels, dtype=np.int32)
        self.h = C.c_void_p(L.orc_hnsw_graph_new(C.c_int64(n), C.c_int(m), C.c_int(ef_construct), ptr(self.levels)))

    def build(self, column, distance):
        col = np.ascontiguousarray(column, dtype=np.float32)
        self.L.orc_hnsw_build(self.h, ptr(col), C.c_int(col.shape[1]), C.c_int(DIST[distance]))

    def links(self, p, level):
        out = np.zeros(self.m0, dtype=np.uint32)
        c = self.L.orc_hnsw_graph_links(self.h, C.c_uint32(p), C.c_int(level), ptr(out))
        return out[:max(c, 0)].copy()

    def set_links(self, p, level, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        self.L.orc_hnsw_graph_set_links(self.h, C.c_uint32(p), C.c_int(level), ptr(a), C.c_int(len(a)))

    def entry(self):
        p, lv = C.c_uint32(), C.c_int()
        has = self.L.orc_hnsw_graph_entry(self.h, C.byref(p), C.byref(lv))
        return (p.value, lv.value) if has else None

    def set_entry(self, p, level):
        self.L.orc_hnsw_graph_set_entry(self.h, C.c_uint32(p), C.c_int(level))

    def search(self, quant, query, limit):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = np.zeros(limit, dtype=np.uint32)
        dist = np.zeros(limit, dtype=np.float32)
        k = self.L.orc_hnsw_search(self.h, ptr(quant.encoded), C.byref(quant.meta), ptr(q), C.c_int(limit), ptr(ids), ptr(dist))
        return ids[:k], dist[:k]

    def free(self):
        self.L.orc_hnsw_graph_free(self.h)
        self.h = None
