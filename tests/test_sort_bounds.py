"""databend_amd.sort_bounds.Bounds against the reference's own known answers (core/bounds.rs:200-311: test_merge, test_reduce,
test_dedup_reduce) — the expected values below are the ones those tests assert."""
from databend_amd.sort_bounds import Bounds, balanced_cuts

DESC = lambda a, b: a > b   # SimpleRowsDesc


def test_merge_known_answers():
    b = Bounds.from_column([0, 7, 6, 6, 6])
    assert b == Bounds([[0, 6, 6, 6, 7]])
    merged = Bounds.merge([b, Bounds(), Bounds.from_column([0, 1, 2])], 3)
    assert merged == Bounds([[6, 7], [2, 6, 6], [0, 0, 1]])
    data = [Bounds.from_column(v, DESC) for v in ([77, -2, 7], [3, 8, 6, 1, 1], [2])]
    assert Bounds.merge(data, 2, DESC) == Bounds([[-2], [1, 1], [3, 2], [7, 6], [77, 8]])
    assert Bounds.merge([], 4) == Bounds() and Bounds.merge([b], 4) is b


def test_reduce_known_answers():
    data = [Bounds.from_column(v, DESC) for v in ([77, -2, 7], [3, 8, 6, 1, 1], [2])]
    bounds = Bounds.merge(data, 2, DESC)
    assert bounds.reduce(4) == Bounds([[8, 6, 2, 1]])
    assert bounds.reduce(3) == Bounds([[8, 3, 1]])
    assert bounds.reduce(2) == Bounds([[7, 1]])
    assert bounds.reduce(1) == Bounds([[3]])
    assert bounds.reduce(9) is None and bounds.reduce(0) == Bounds()


def test_dedup_reduce_known_answers():
    assert Bounds.new_unchecked([1, 2, 2, 3, 3, 3, 4, 5, 5]).dedup_reduce(3) == Bounds([[2, 3, 5]])
    assert Bounds.new_unchecked([5, 5, 4, 3, 3, 3, 2, 2, 1]).dedup_reduce(3) == Bounds([[4, 3, 1]])
    assert Bounds([[5, 6, 7, 7], [3, 3, 4, 5], [1, 2, 2, 3]]).dedup_reduce(5) == Bounds([[2, 3, 4, 6, 7]])
    assert Bounds([[1, 1, 1, 1, 1]]).dedup_reduce(3) == Bounds([[1]])
    # not a reference vector: worked by hand through bounds.rs:140-185 with n = len (the first target is step / 2 = 0.5, so row 0
    # is never a bound — harmless, bounds are cut points)
    assert Bounds([[5, 6, 7, 7], [3, 3, 4, 5], [1, 2, 2, 3]]).dedup() == Bounds([[2, 3, 4, 5, 6, 7]])


def test_next_bound_walks_the_reversed_storage():
    b = Bounds([[6, 7], [2, 6, 6], [0, 0, 1]])
    assert len(b) == 8 and not b.is_empty()
    assert [b.next_bound() for _ in range(8)] == [0, 0, 1, 2, 6, 6, 6, 7]
    assert b.next_bound() is None and b.is_empty() and len(b) == 0


def test_balanced_cuts_split_the_samples_evenly():
    rows = list(range(100))
    assert balanced_cuts(rows, 4) == [24, 49, 74] and balanced_cuts(rows, 1) == [] and balanced_cuts([], 3) == []
    assert balanced_cuts([1, 1, 1, 1, 1, 1], 3) == [1] and balanced_cuts([5], 2) == []
    assert balanced_cuts([1, 2], 2) == [1] and balanced_cuts([(None, 3), (None, 3), (2, 1), (2, 9)], 2) == [(None, 3)]
