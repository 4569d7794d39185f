"""Host side of the distributed sort: the Bounds that cut every node's rows into ranges.

Reference: src/query/pipeline/transforms/src/processors/transforms/sorts/core/bounds.rs (Bounds: from_column :41-53, merge :55-80,
next_bound :82-98, reduce :109-137, dedup_reduce :140-185, dedup :187-189) and sort_broadcast.rs:150-197 (every node merges its own
samples into Bounds, broadcasts them, merges what it received and dedups: the GLOBAL bounds, equal on every node). Bounds are a few
hundred rows, so this is plain host code; the rows themselves are cut on the device (dbhip_sort_bound_partition).

A bound is any Python value with == (the dedup) — an int in the reference's own tests, a tuple of key values (None = NULL) for
multi-column sorts; `before(a, b)` says whether a sorts strictly before b (default: a < b; pass `lambda a, b: a > b` for the
reference's SimpleRowsDesc). Like the reference, the blocks are STORED IN REVERSE ORDER (bounds.rs:27-30): blocks[-1] holds the
first bounds in sort order."""
import heapq
from functools import cmp_to_key


def _asc(a, b):
    return a < b


class Bounds:
    def __init__(self, blocks=()):
        self.blocks = [list(b) for b in blocks]

    def __eq__(self, other):
        return isinstance(other, Bounds) and self.blocks == other.blocks

    def __repr__(self):
        return "Bounds(%r)" % (self.blocks,)

    @classmethod
    def new_unchecked(cls, column):                      # bounds.rs:34-39
        return cls([column]) if len(column) else cls()

    @classmethod
    def from_column(cls, column, before=_asc):           # bounds.rs:41-53: DataBlock::sort of the samples
        key = cmp_to_key(lambda a, b: -1 if before(a, b) else (1 if before(b, a) else 0))
        return cls([sorted(column, key=key)])

    @classmethod
    def merge(cls, vector, batch_rows, before=_asc):     # bounds.rs:55-80: LoserTreeMerger over the streams, blocks of batch_rows
        if len(vector) == 0:
            return cls()
        if len(vector) == 1:
            return vector[0]
        key = cmp_to_key(lambda a, b: -1 if before(a, b) else (1 if before(b, a) else 0))
        rows = list(heapq.merge(*[list(v.rows()) for v in vector], key=key))
        blocks = [rows[i: i + batch_rows] for i in range(0, len(rows), batch_rows)]
        return cls(blocks[::-1])

    def rows(self):
        """every bound in sort order (the reversed storage unrolled, bounds.rs:120-124)"""
        for b in reversed(self.blocks):
            yield from b

    def next_bound(self):                                # bounds.rs:82-98
        if not self.blocks:
            return None
        last = self.blocks[-1]
        bound = last[0]
        if len(last) == 1:
            self.blocks.pop()
        else:
            self.blocks[-1] = last[1:]
        return bound

    def __len__(self):
        return sum(len(b) for b in self.blocks)

    def is_empty(self):
        return all(len(b) == 0 for b in self.blocks)

    def reduce(self, n):                                 # bounds.rs:109-137: n evenly spaced bounds, None when there are not more than n
        if n == 0:
            return Bounds()
        total = len(self)
        if n >= total:
            return None
        step = total // n
        offset = step // 2
        picked = [r for i, r in enumerate(self.rows()) if i < step * n and i % step == offset]
        return Bounds([picked])

    def dedup_reduce(self, n):                           # bounds.rs:140-185: at most n DISTINCT bounds, re-spaced after long runs of equals
        if n == 0:
            return Bounds()
        total = len(self)
        step = total / n
        target = step / 2.0
        picked = []
        have_last, last = False, None
        for i, r in enumerate(self.rows()):
            if len(picked) >= n:
                break
            if float(i) < target:
                continue
            if have_last and r == last:
                continue
            picked.append(r)
            target += step
            if float(i) > target and len(picked) < n:
                step = (total - i) / (n - len(picked))
                target = i + step / 2.0
            have_last, last = True, r
        return Bounds.new_unchecked(picked)

    def dedup(self):                                     # bounds.rs:187-189
        return self.dedup_reduce(len(self))


def balanced_cuts(rows, ranges):
    """The cut rows of the device plan: `rows` = every rank's samples in sort order, `ranges` = the number of ranks. Bound j is the
    LAST sample of the j-th of `ranges` equal slices (rows <= bound go left, sort_spill.rs:1008-1040), equal neighbours dropped —
    so the ranges hold equal shares of the samples. (Bounds::dedup_reduce(ranges - 1) is not used for this: it centres n bounds in
    n slices, which makes the outer ranges half as large as the inner ones — 25 / 50 / 25 % on three ranks.)"""
    total = len(rows)
    cuts = []
    for j in range(ranges - 1):
        at = ((j + 1) * total) // ranges - 1
        if at < 0:
            continue
        if not cuts or cuts[-1] != rows[at]:
            cuts.append(rows[at])
    return cuts
