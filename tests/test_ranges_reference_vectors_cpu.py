"""The range logic of the product -- make_examples_core.{merge,intersect,exclude}_ranges,
partition (what a nucleus RangeSet does for regions_to_process) and realigner.utils.
{overlap_len,find_max_overlapping} (read -> assembly-window assignment) -- on the vectors of
third_party/nucleus/util/ranges_test.py:111-124 (merging), :296-352 (intersection), :354-379
(exclusion), :410-450 (partition), :551-667 (overlap length, maximal overlap)."""
import itertools

import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_core as mec
from deepvariant_amd.realigner import utils as U


def lit(s):
  name, span = s.split(':')
  if '-' in span:
    a, b = span.split('-')
    return T.Range(name, int(a) - 1, int(b))
  return T.Range(name, int(span) - 1, int(span))


def tup(rs):
  return sorted((r.reference_name, r.start, r.end) for r in rs)


@pytest.mark.parametrize('regions,expected', [
    (['1:1-5', '1:3-8'], ['1:1-8']),
    (['1:1-5', '1:3-8', '1:6-9'], ['1:1-9']),
    (['1:1-5', '1:5-8'], ['1:1-8']),                        # adjacent intervals are merged
    (['1:1-5', '1:5-8', '1:8-10'], ['1:1-10']),
    (['1:1-5', '1:6-8'], ['1:1-5', '1:6-8']),
])
def test_overlapping_and_adjacent_ranges_are_merged(regions, expected):
  # NB: '1:1-5' and '1:6-8' are [0,5) and [5,8) -- adjacent -- and the reference's own expected
  # set, built through the same RangeSet, merges them too; compare through the merge as it does
  assert tup(mec.merge_ranges([lit(x) for x in regions])) == tup(mec.merge_ranges([lit(x) for x in expected]))


@pytest.mark.parametrize('regions,expected', [
    ([['1:1-10']], ['1:1-10']),
    ([['1:1-10'], ['1:1-10']], ['1:1-10']),
    ([['1:1-10'], ['1:1-10'], ['1:1-10']], ['1:1-10']),
    ([['1:1-10'], ['1:11-15']], []),
    ([['1:1-10'], ['1:10-15']], ['1:10']),
    ([['1:1-10'], ['1:9-15']], ['1:9-10']),
    ([['1:5-10'], ['1:1-15']], ['1:5-10']),
    ([['1:5-10'], ['1:1-4']], []),
    ([['1:5-10'], ['1:1-5']], ['1:5']),
    ([['1:5-15'], ['1:6-8', '1:10-12']], ['1:6-8', '1:10-12']),
    ([['1:5-15'], ['1:3-8', '1:10-12']], ['1:5-8', '1:10-12']),
    ([['1:5-15'], ['1:3-8', '1:10-20']], ['1:5-8', '1:10-15']),
    ([['1:5-15'], ['1:3-8', '1:6-10']], ['1:5-10']),
    ([['1:5-15'], ['1:3-8', '1:6-10', '1:13']], ['1:5-10', '1:13']),
    ([['1:5-15', '1:20-25'], ['1:3-8', '1:16-23']], ['1:5-8', '1:20-23']),
    ([['1:5-15', '1:20-25'], ['1:3-8', '1:50-60']], ['1:5-8']),
    ([['1:5-15', '1:20-25'], ['1:3-4', '1:16-23']], ['1:20-23']),
    ([['1:10-20'], ['1:5-15']], ['1:10-15']),
    ([['1:10-20'], ['1:5-15'], ['1:13-30']], ['1:13-15']),
    ([['1:10-20'], ['1:5-15'], ['1:25-30']], []),
    ([['1:10-20'], ['2:10-20']], []),
    ([['1:10-20', '2:11-14'], ['1:11-14']], ['1:11-14']),
    ([['1:10-20', '2:11-14'], ['2:10-20']], ['2:11-14']),
])
def test_intersection(regions, expected):
  sets = [[lit(x) for x in r] for r in regions]
  for order in (sets, sets[::-1]):           # "even if we do it in a different direction"
    acc = mec.merge_ranges(order[0])
    for other in order[1:]:
      acc = mec.intersect_ranges(acc, other)
    assert tup(acc) == tup(mec.merge_ranges([lit(x) for x in expected]))


@pytest.mark.parametrize('lhs,rhs,expected', [
    (['1:1-100'], ['1:10-20'], ['1:1-9', '1:21-100']),
    (['1:1-100'], [], ['1:1-100']),
    (['1:1-100', '2:1-10'], ['2:1-100'], ['1:1-100']),
    (['1:1-100'], ['1:10-20', '1:15-30'], ['1:1-9', '1:31-100']),
    (['1:1-100'], ['1:10-20', '1:30-40'], ['1:1-9', '1:21-29', '1:41-100']),
    (['1:1-100'], ['2:1-100'], ['1:1-100']),
    (['1:1-100'], ['1:1-100'], []),
    ([], ['1:1-100'], []),
])
def test_exclude_regions(lhs, rhs, expected):
  got = mec.exclude_ranges([lit(x) for x in lhs], [lit(x) for x in rhs])
  assert tup(got) == tup([lit(x) for x in expected])


@pytest.mark.parametrize('size,expected', [
    (50, [('chr1', 0, 50), ('chr1', 50, 76), ('chr2', 0, 50), ('chr2', 50, 100), ('chr2', 100, 121),
          ('chrM', 0, 50), ('chrM', 50, 100)]),
    (120, [('chr1', 0, 76), ('chr2', 0, 120), ('chr2', 120, 121), ('chrM', 0, 100)]),
    (500, [('chr1', 0, 76), ('chr2', 0, 121), ('chrM', 0, 100)]),
])
def test_partitions(size, expected):       # without contigs a RangeSet iterates in name order
  merged = mec.merge_ranges([T.Range('chrM', 0, 100), T.Range('chr1', 0, 76), T.Range('chr2', 0, 121)])
  got = [p for r in merged for p in mec.partition(r, size)]
  assert [(p.reference_name, p.start, p.end) for p in got] == expected


@pytest.mark.parametrize('size,expected', [
    (10, [('1', 0, 10), ('1', 20, 30), ('1', 30, 40), ('1', 45, 50)]),
    (7, [('1', 0, 7), ('1', 7, 10), ('1', 20, 27), ('1', 27, 34), ('1', 34, 40), ('1', 45, 50)]),
    (50, [('1', 0, 10), ('1', 20, 40), ('1', 45, 50)]),
])
def test_partition_of_multiple_intervals(size, expected):
  merged = mec.merge_ranges([T.Range('1', 0, 10), T.Range('1', 20, 40), T.Range('1', 45, 50)])
  got = [p for r in merged for p in mec.partition(r, size)]
  assert [(p.reference_name, p.start, p.end) for p in got] == expected
  for bad in (-10, 0):
    with pytest.raises(ValueError):
      list(mec.partition(T.Range('chrM', 0, 100), bad))


@pytest.mark.parametrize('a,b,expected', [
    (('1', 0, 10), ('2', 0, 10), 0), (('1', 0, 10), ('1', 10, 20), 0), (('1', 0, 10), ('1', 100, 200), 0),
    (('1', 10, 10), ('1', 0, 20), 0), (('1', 0, 100), ('1', 50, 99), 49), (('1', 0, 10), ('1', 0, 1), 1),
    (('1', 0, 10), ('1', 0, 2), 2), (('1', 1, 10), ('1', 0, 1), 0),
])
def test_overlap_len(a, b, expected):
  assert U.overlap_len(T.Range(*a), T.Range(*b)) == expected
  assert U.overlap_len(T.Range(*b), T.Range(*a)) == expected


@pytest.mark.parametrize('query,search,expected', [
    (('1', 20, 30), [], None),
    (('1', 20, 30), [('1', 0, 10), ('1', 5, 10)], None),
    (('1', 4, 10), [('1', 0, 10), ('1', 5, 10)], 0),
    (('1', 9, 20), [('1', 0, 10), ('1', 5, 15)], 1),
    (('1', 9, 20), [('1', 0, 10), ('1', 0, 15), ('1', 5, 20)], 2),
    (('1', 5, 13), [('1', 0, 10), ('1', 0, 15), ('1', 10, 20)], 1),
    (('2', 0, 10), [('1', 0, 10), ('2', 5, 15), ('3', 0, 10)], 1),
    (('1', 5, 15), [('1', 0, 10), ('1', 10, 20), ('1', 12, 20)], 0),      # equal overlap: the first
])
def test_find_max_overlapping(query, search, expected):
  assert U.find_max_overlapping(T.Range(*query), [T.Range(*s) for s in search]) == expected


def test_find_max_overlapping_order_and_ties():
  query = T.Range('1', 4, 12)
  search = [T.Range('1', 0, 10), T.Range('1', 10, 20), T.Range('1', 12, 20)]
  for perm in itertools.permutations(search):
    assert U.find_max_overlapping(query, list(perm)) == list(perm).index(search[0])
  q = T.Range('1', 0, 10)
  two = [T.Range('1', 0, 5), T.Range('1', 5, 10)]
  for s in (two, two[::-1]):
    assert U.find_max_overlapping(q, s) == 0


# ---- third_party/nucleus/util/utils_test.py:50-165: read_range / read_end / read_overlaps_region ----
def test_read_range_and_end():
  start = 10000001
  for cigar, span in (('2M1I3M', 5), ('2M16D3M', 5 + 16)):
    read = T.make_read('AAACAG', chrom='chrX', start=start, cigar=cigar, quals=list(range(10, 16)), name='read1')
    r = U.read_range(read)
    assert (r.reference_name, r.start, r.end) == ('chrX', start, start + span)


_OVERLAPS = [
    (0, 3, 'chr1', 4, 10, False), (0, 3, 'chr1', 3, 10, False), (0, 3, 'chr1', 2, 10, True),
    (0, 3, 'chr1', 1, 10, True), (0, 3, 'chr1', 0, 10, True), (0, 3, 'chr1', 0, 1, True),
    (0, 3, 'chr1', 0, 2, True), (0, 3, 'chr1', 0, 3, True), (0, 3, 'chr1', 1, 2, True),
    (0, 3, 'chr1', 1, 3, True), (0, 3, 'chr1', 2, 3, True), (0, 3, 'chr1', 0, 4, True),
    (0, 3, 'chr1', 1, 4, True), (0, 3, 'chr2', 1, 4, False),
]


@pytest.mark.parametrize('s1,e1,ref2,s2,e2,expected', _OVERLAPS)
def test_read_overlaps_region(s1, e1, ref2, s2, e2, expected):
  """The same answer from the three places that ask the question: the host utilities, the
  oracle's ReadOverlaps and the C ABI's dv_query_reads (InMemoryReader::Query; one contig
  per table, so the other-contig case is the host's)."""
  import ctypes as C
  import numpy as np
  from deepvariant_amd import _lib
  from oracle import oracle as O
  nbp = e1 - s1
  read = T.make_read('A' * nbp, chrom='chr1', start=s1, cigar='%dM' % nbp, quals=[30] * nbp)
  region = T.Range(ref2, s2, e2)
  assert U.ranges_overlap(U.read_range(read), region) is expected
  assert U.ranges_overlap(region, U.read_range(read)) is expected
  if ref2 == 'chr1':
    assert O.read_overlaps(read, s2, e2) is expected
    pos = np.array([s1], np.int32)
    off = np.array([0, 1], np.uint32)
    cig = np.array([(nbp << 4) | 1], np.uint32)
    q0, q1 = np.array([s2], np.int64), np.array([e2], np.int64)
    list_off = np.zeros(2, np.uint32)
    lib = _lib.lib()
    lib.dv_query_reads.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.check(lib.dv_query_reads(1, pos.ctypes.data, off.ctypes.data, cig.ctypes.data, 1, q0.ctypes.data,
                                  q1.ctypes.data, list_off.ctypes.data, None))
    assert bool(list_off[1]) is expected
