"""Window selection (deepvariant_amd/realigner/window_selector.py) against the reference's
vectors, deepvariant/realigner/window_selector_test.py.  The HOST logic is what runs here;
allele counts come from the oracle counter (tests/realigner_fixture.OracleAlleleCounter).  The
same vectors run through the device counter in tests/test_hip_realigner.py."""
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd.realigner import window_selector as ws
from tests import realigner_fixture as RF
from tests import window_selector_vectors as V


@pytest.mark.parametrize('case', V.CASES, ids=lambda c: '%s-%s' % (c[0], '+'.join(r[2] for r in c[1])))
def test_candidates_from_reads(case):
  V.run_case(case, RF.OracleAlleleCounter)


@pytest.mark.parametrize('read_mapq', range(10, 15))
@pytest.mark.parametrize('min_mapq', range(8, 17))
def test_candidates_respect_mapq(read_mapq, min_mapq):        # :425-446
  case = ('threshold', [('AGA', 10, '3M', None, read_mapq)], [11] if read_mapq >= min_mapq else [], {})
  V.run_case(case, RF.OracleAlleleCounter, min_mapq=min_mapq)


R = lambda a, b: T.Range('ref', a, b)


@pytest.mark.parametrize('candidates,expected', [
    ([100, 200], [R(96, 104), R(196, 204)]),
    ([100, 200, 300], [R(96, 104), R(196, 204), R(296, 304)]),
    ([2, 8], [R(-2, 12)]),
    ([2, 14], [R(-2, 6), R(10, 18)]),
    ([2, 10], [R(-2, 14)]),          # boundary: merged
    ([2, 11], [R(-2, 6), R(7, 15)])  # boundary: not merged
])
def test_candidates_to_windows(candidates, expected):          # :448-503
  assert ws._candidates_to_windows(V.threshold_config(), candidates, 'ref') == expected


@pytest.mark.parametrize('d', range(1, 20))
def test_candidates_to_windows_distances(d):                   # :505-567
  config = V.threshold_config()
  config.min_windows_distance = d
  assert ws._candidates_to_windows(config, [100], 'ref') == [R(100 - d, 100 + d)]
  assert ws._candidates_to_windows(config, [100, 100 - 2 * d - 1, 100 + d], 'ref') == [
      R(100 - 3 * d - 1, 100 - d - 1), R(100 - d, 100 + 2 * d)]
  close = [100 + i * d for i in range(5)]
  assert ws._candidates_to_windows(config, close, 'ref') == [R(100 - d, max(close) + d)]


def test_select_windows():                                     # :569-585
  reads = [V.mk('AGA', 99, '3M', [q] * 3) for q in (64, 63, 62)]
  chrom = reads[0].alignment.position.reference_name
  with RF.oracle_allele_counter():
    got = ws.select_windows(V.threshold_config(), RF.StringRef(chrom, 'A' * 300), reads, T.Range(chrom, 0, 200))
  assert got == [T.Range(chrom, 96, 104)]


def test_select_windows_without_reads_or_with_realign_all():   # :587-597
  ref = RF.StringRef('chr1', 'A' * 500)
  assert ws.select_windows(V.threshold_config(), ref, [], T.Range('chr1', 1, 100)) == []
  config = V.threshold_config()
  config.realign_all = True
  region = T.Range('chr1', 1, 100)
  assert ws.select_windows(config, ref, [V.mk('AGA', 10, '3M')], region) == [region]


def test_min_allele_support_and_strict_insertion_filter():
  """AlleleFilter (window_selector.cc:64-80): an allele seen once is dropped at
  min_allele_support 2; a <= 2-base insertion below 8 % of the reads is dropped when the
  strict filter is on."""
  config = V.threshold_config()
  config.min_allele_support = 2
  assert V.candidates(config, [V.mk('AAGA', 10, '4M')], counter_cls=RF.OracleAlleleCounter) == []
  two = [V.mk('AAGA', 10, '4M'), V.mk('AAGA', 10, '4M')]
  assert V.candidates(config, two, counter_cls=RF.OracleAlleleCounter) == [12]
  config = V.threshold_config()
  config.window_selector_model.variant_reads_model.max_num_supporting_reads = 100
  reads = [V.mk('AAAAT', 10, '4M1I')] + [V.mk('AAAA', 10, '4M') for _ in range(12)]    # 1 of 13 = 7.7 %
  assert V.candidates(config, reads, counter_cls=RF.OracleAlleleCounter) == [13, 14]
  config.enable_strict_insertion_filter = True
  assert V.candidates(config, reads, counter_cls=RF.OracleAlleleCounter) == []
  assert V.candidates(config, reads[:12], counter_cls=RF.OracleAlleleCounter) == [13, 14]  # 1 of 12 = 8.3 %
