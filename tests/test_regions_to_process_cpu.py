"""make_examples_core.regions_to_process and the range-set logic under it, on the vectors of
deepvariant/make_examples_core_test.py:666-826 (calling regions, partition sizes, order within
and across contigs, sharding, bad shard arguments) and of
third_party/nucleus/util/ranges_test.py for the RangeSet operations it is built from."""
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_core as mec


def lit(s):
  """ranges.parse_literal: '1:1-10' (1-based, inclusive) or '1:7'."""
  name, span = s.split(':')
  if '-' in span:
    a, b = span.split('-')
    return T.Range(name, int(a) - 1, int(b))
  return T.Range(name, int(span) - 1, int(span))


def lits(xs):
  return None if xs is None else [lit(x) for x in xs]


def key(rs):
  return sorted((r.reference_name, r.start, r.end) for r in rs)


@pytest.mark.parametrize('calling_regions,expected', [
    (['1:1-10'], ['1:1-10']),
    (['1:1-100'], ['1:1-100']),
    (['1:50-150'], ['1:50-100']),
    (None, ['1:1-100', '2:1-200']),
    (['1:20-50'], ['1:20-50']),
    (['1:20-30', '1:40-60', '3:10-50'], ['1:20-30', '1:40-60']),     # chr3 is not one of the contigs
    (['1:25-30', '1:20-40'], ['1:20-40']),                           # overlapping calling regions
])
def test_regions_to_process(calling_regions, expected):       # make_examples_core_test.py:666-688
  contigs = [('1', 100), ('2', 200)]
  got = mec.regions_to_process(contigs, 1000, calling_regions=lits(calling_regions))
  assert key(got) == key(lits(expected))


@pytest.mark.parametrize('max_size,calling_regions,expected', [
    (50, None, ['1:1-50', '1:51-100', '2:1-50', '2:51-76', '3:1-50', '3:51-100', '3:101-121']),
    (120, None, ['1:1-100', '2:1-76', '3:1-120', '3:121']),
    (500, None, ['1:1-100', '2:1-76', '3:1-121']),
    (10, ['1:1-20', '1:30-35'], ['1:1-10', '1:11-20', '1:30-35']),
    (8, ['1:1-20', '1:30-35'], ['1:1-8', '1:9-16', '1:17-20', '1:30-35']),
])
def test_regions_to_process_partition(max_size, calling_regions, expected):   # :690-718
  contigs = [('1', 100), ('2', 76), ('3', 121)]
  got = mec.regions_to_process(contigs, max_size, calling_regions=lits(calling_regions))
  assert key(got) == key(lits(expected))


def test_regions_to_process_sorted_within_contig():            # :758-771 (exact order)
  got = mec.regions_to_process([('z', 100)], 100,
                               calling_regions=lits(['z:15', 'z:20', 'z:6', 'z:25-30', 'z:3-4']))
  assert [(r.reference_name, r.start, r.end) for r in got] == \
      [(r.reference_name, r.start, r.end) for r in lits(['z:3-4', 'z:6', 'z:15', 'z:20', 'z:25-30'])]


def test_regions_to_process_sorted_contigs():                  # :773-784: FASTA order, not names
  contigs = [('z', 100), ('a', 100), ('n', 100)]
  got = mec.regions_to_process(contigs, 100, calling_regions=lits(['a:10', 'n:1', 'z:20', 'z:5']))
  assert [(r.reference_name, r.start, r.end) for r in got] == \
      [(r.reference_name, r.start, r.end) for r in lits(['z:5', 'z:20', 'a:10', 'n:1'])]


@pytest.mark.parametrize('num_shards', [2, 3, 4, 5, 50])
@pytest.mark.parametrize('round_robin', [True, False])
def test_regions_to_process_sharding(num_shards, round_robin):  # :786-804
  contigs = [('z', 100), ('a', 100), ('n', 100)]
  unsharded = mec.regions_to_process(contigs, 5, task_id=0, num_shards=0)
  sharded = []
  for task in range(num_shards):
    part = mec.regions_to_process(contigs, 5, task_id=task, num_shards=num_shards,
                                  round_robin_sampling=round_robin)
    sharded.extend(part)
    if round_robin:   # the rule the ranks of a multi-GPU run use (deepvariant_amd/dist.py)
      from deepvariant_amd import dist
      assert part == dist.regions_for_rank(unsharded, task, num_shards)
  assert key(sharded) == key(unsharded) and len(sharded) == len(unsharded) == 60


@pytest.mark.parametrize('task,num_shards', [(None, 0), (None, 2), (2, None), (0, None), (-1, 2),
                                             (0, -2), (2, 2), (3, 2)])
def test_regions_to_process_fails_with_bad_shard_args(task, num_shards):      # :806-826
  with pytest.raises(ValueError):
    mec.regions_to_process([('z', 100), ('a', 100), ('n', 100)], 10, task_id=task, num_shards=num_shards)


def test_range_set_operations():
  """RangeSet semantics (third_party/nucleus/util/ranges.py:77-300): adjacent ranges merge,
  intersection keeps common bases only, exclusion chops, unknown contigs are an error for a
  set that has contigs."""
  R = T.Range
  assert key(mec.merge_ranges([R('1', 0, 10), R('1', 10, 20), R('1', 25, 30), R('1', 5, 12)])) == \
      [('1', 0, 20), ('1', 25, 30)]
  a = [R('chr1', 0, 10), R('chr2', 19, 30)]
  b = [R('chr1', 4, 8), R('chr3', 9, 40)]
  c = [R('chr1', 2, 7), R('chr3', 9, 30)]
  assert key(mec.intersect_ranges(mec.intersect_ranges(a, b), c)) == [('chr1', 4, 7)]   # ranges.py:213-225
  assert key(mec.exclude_ranges([R('1', 0, 100)], [R('1', 10, 20), R('1', 50, 60), R('2', 0, 5)])) == \
      [('1', 0, 10), ('1', 20, 50), ('1', 60, 100)]
  assert mec.exclude_ranges([R('1', 0, 10)], [R('1', 0, 10)]) == []
  with pytest.raises(ValueError):
    mec.merge_ranges([R('9', 0, 1)], ['1', '2'], known_contigs_only=True)
  got = mec.build_calling_regions([('1', 100), ('2', 50)], [R('1', 10, 90), R('2', 0, 50)], [R('1', 20, 30)])
  assert [(r.reference_name, r.start, r.end) for r in got] == [('1', 10, 20), ('1', 30, 90), ('2', 0, 50)]
