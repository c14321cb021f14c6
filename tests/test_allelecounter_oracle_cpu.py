"""The AlleleCounter oracle (oracle/allelecounter_ref.py) against the vectors of
deepvariant/allelecounter_test.cc (tests/allelecounter_vectors.py)."""
import pytest

from oracle import allelecounter_ref as R
from tests import allelecounter_vectors as V

CASES = V.cases()


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_reference_vectors(case):
  name, contig, start, end, reads, expected = case
  counter = R.AlleleCounter(V.TestRef(), contig, start, end, min_base_quality=V.MIN_BQ)
  for r in reads:
    counter.add(r)
  assert counter.n_reads_counted == len(reads)
  for i, (ac, want) in enumerate(zip(counter.counts, expected)):
    got = sorted(a.key() for a in R.sum_allele_counts(ac))
    assert got == sorted(want), (name, i)
    assert R.total_allele_counts(ac) == sum(n for _, _, n in want), (name, i)
    assert ac.ref_base == V.TestRef().get_bases(contig, start + i, start + i + 1)


def test_low_mapq_reads_are_ignored():
  counter = R.AlleleCounter(V.TestRef(), 'chr1', 0, 4, min_mapping_quality=10)
  counter.add(V.make_read('chr1', 0, 'ACGT', ['4M'], mapq=0))
  assert counter.n_reads_counted == 0
  assert all(R.total_allele_counts(ac) == 0 for ac in counter.counts)


def test_track_ref_reads_keeps_reference_reads_at_candidate_positions():
  """allelecounter.cc:504-512 + SumAlleleCounts' track_ref_reads branch (:100-113).  The reference
  has no count vector for this (its only track_ref_reads test is about methylation flags,
  allelecounter_test.cc:469-516), so the expectations are spelled out from the code."""
  from oracle import allelecounter_ref as R
  ref = V.TestRef()
  bases = ref.get_bases('chr1', 10, 15)
  sub = 'A' if bases[2] != 'A' else 'C'
  reads = [V.make_read('chr1', 10, bases, ['5M'], name='r1'),
           V.make_read('chr1', 10, bases[:2] + sub + bases[3:], ['5M'], name='r2')]
  counter = R.AlleleCounter(ref, 'chr1', 10, 15, track_ref_reads=True, candidate_positions=[11, 12, 99])
  for r in reads:
    counter.add(r)
  got = [{k: (a.bases, a.type) for k, a in c.read_alleles.items()} for c in counter.counts]
  assert got[0] == {} and got[3] == {} and got[4] == {}
  assert got[1] == {'r1/0': (bases[1], R.REFERENCE), 'r2/0': (bases[1], R.REFERENCE)}
  assert got[2] == {'r1/0': (bases[2], R.REFERENCE), 'r2/0': (sub, R.SUBSTITUTION)}
  assert [c.ref_supporting_read_count for c in counter.counts] == [2, 2, 1, 2, 2]
  sums = [sorted((a.bases, a.type, a.count) for a in R.sum_allele_counts(c)) for c in counter.counts]
  assert sums[0] == [] and sums[1] == [(bases[1], R.REFERENCE, 2)]          # no synthetic reference allele
  assert sums[2] == sorted([(bases[2], R.REFERENCE, 1), (sub, R.SUBSTITUTION, 1)])
  assert [R.total_allele_counts(c) for c in counter.counts] == [2, 2, 2, 2, 2]
