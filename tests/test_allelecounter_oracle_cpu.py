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
