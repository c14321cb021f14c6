"""Read phasing (csrc/direct_phasing.cpp through dv_phase_reads) against the reference's own
vectors, deepvariant/direct_phasing_test.cc: graph construction, candidate filtering, phases of
reads through clean, noisy, broken and tied blocks, phased variants with block starts.
Host only."""
import re

import pytest

from deepvariant_amd import _lib
from deepvariant_amd import direct_phasing
from deepvariant_amd import dv_types as T


def cand(start, end, allele_support=None, ref_support=None, low_quality=()):
  """MakeCandidate (:79-106)."""
  c = T.DeepVariantCall(variant=T.Variant('chr1', start, end))
  for allele, names in (allele_support or {}).items():
    c.allele_support_ext[allele] = [T.ReadSupport(n, n in low_quality) for n in names]
  c.ref_support_ext = [T.ReadSupport(n, n in low_quality) for n in (ref_support or [])]
  return c


def reads(n):
  """CreateTestReads (:231-241): read<i>, read_number 0."""
  return [T.cc_make_read('chr1', 89 + i, 'ACGTTGACTTGC', ['12M'], 'read%d' % i) for i in range(1, n + 1)]


def r(*ids):
  return ['read%d/0' % i for i in ids]


def phaser():
  return direct_phasing.DirectPhasing(min_alleles_to_phase=2)     # CreateDefaultDirectPhasing


def graph(dp):
  text = dp.graphviz()
  labels = dict(re.findall(r'(\d+)\[label="([^"]+)"\]', text))
  edges = {(labels[a], labels[b]): float(w) for a, b, w in re.findall(r'(\d+)->(\d+) \[label=([0-9.]+)\]', text)}
  return set(labels.values()), edges


def test_build_graph_simple():                              # :243-340
  dp = phaser()
  dp.phase([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5, 6)}),
            cand(105, 106, {'C': r(1, 2, 3)}, r(4, 5, 6)),
            cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5)})], reads(6))
  vertices, edges = graph(dp)
  assert vertices == {'100 A', '100 C', '105 C', '105 REF', '110 T', '110 G'}
  assert set(edges) == {('100 A', '105 C'), ('100 C', '105 REF'), ('105 C', '110 T'), ('105 REF', '110 G')}
  assert edges[('100 A', '105 C')] == 3.0 and edges[('105 REF', '110 G')] == 2.0   # one per linking read


def test_low_quality_and_unknown_reads_do_not_support():    # ReadSupportFromProto*, :193-229
  dp = phaser()
  lq = set(r(3))
  cs = [cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5) + ['stranger/0']}, low_quality=lq),
        cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5)}, low_quality=lq)]
  assert dp.phase(cs, reads(5)) == [1, 1, 0, 2, 2]
  _, edges = graph(dp)
  assert edges[('100 A', '110 T')] == 2.0                   # read3 is low quality at both sites


@pytest.mark.parametrize('candidates,n_reads,expected', [
    # PhaseReadSimpleTest: the one-allele site at 105 is filtered
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 4, 5)}),
      cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5)})], 5, [1, 1, 1, 2, 2]),
    # PhaseReadWithErrorCorrection: read3 switches sides at 110 only
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 3, 4, 5)}),
      cand(110, 111, {'T': r(1, 2), 'G': r(3, 4, 5)}), cand(120, 121, {'T': r(1, 2, 3), 'G': r(4, 5)})],
     5, [1, 1, 1, 2, 2]),
    # PhaseReadChangedOrderOfAlleles
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 3, 4, 5)}),
      cand(110, 111, {'T': r(4, 5), 'G': r(1, 2, 3)}), cand(120, 121, {'G': r(4, 5), 'T': r(1, 2, 3)})],
     5, [1, 1, 1, 2, 2]),
    # PhaseReadUnphasedRead: read3 has one allele on each side
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 3, 4, 5)}),
      cand(110, 111, {'T': r(1, 2), 'G': r(4, 5, 3)})], 5, [1, 1, 0, 2, 2]),
    # PhaseReadBrokenPath
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(4, 5), 'G': r(6, 7)}),
      cand(110, 111, {'T': r(6, 7), 'G': r(4, 5)})], 7, [0, 0, 0, 2, 2, 1, 1]),
    # PhaseReadBrokenPathNoConnection: two blocks
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 3), 'G': r(4, 5)}),
      cand(110, 111, {'C': r(6, 7), 'G': r(8, 9)}), cand(120, 121, {'T': r(6, 7), 'G': r(8, 9)})],
     9, [1, 1, 1, 2, 2, 1, 1, 2, 2]),
    # PhaseReadFullyConnectedGraph
    ([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5, 6)}), cand(105, 106, {'C': r(4, 5, 1), 'G': r(2, 3, 6)}),
      cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})], 6, [1, 1, 1, 2, 2, 2]),
])
def test_phase_reads(candidates, n_reads, expected):        # :491-850
  assert phaser().phase(candidates, reads(n_reads)) == expected


@pytest.mark.parametrize('order', [(105, 100, 110), (100, 105, 104, 110)])
def test_unordered_candidates_are_refused(order):            # EXPECT_DEATH, :853-911
  cs = [cand(p, p + 1, {'A': r(1, 2, 3), 'C': r(4, 5, 6)}) for p in order]
  with pytest.raises(_lib.DvError, match='Check failed'):
    phaser().phase(cs, reads(6))


def test_two_blocks_with_score_tie():                        # :913-965
  dp = phaser()
  cs = [cand(100, 101, {'A': r(1, 2), 'C': r(3, 4)}), cand(110, 111, {'G': r(1, 2), 'T': r(3, 4)}),
        cand(120, 121, {'A': r(5, 6, 7, 8), 'C': r(9, 10, 11, 12)})]
  assert dp.phase(cs, reads(12)) == [1, 1, 2, 2] + [0] * 8
  assert dp.get_phased_variants() == [direct_phasing.PhasedVariant(100, 'A', 'C', True),
                                      direct_phasing.PhasedVariant(110, 'G', 'T', False)]


def test_filter_one_allele_candidate_and_indels():           # :967-1030
  dp = phaser()
  dp.phase([cand(100, 101, {'C': r(4, 5, 6)}, r(7)), cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})], reads(7))
  assert graph(dp)[0] == {'110 T', '110 G'}                 # one allele + one reference read: not a site
  dp.phase([cand(100, 102, {'CC': r(4, 5, 6), 'A': r(1, 2)}, r(7)),
            cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})], reads(7))
  assert graph(dp)[0] == {'110 T', '110 G'}                 # an indel allele removes the whole site
  # ... and so does lying inside an earlier indel's span
  dp.phase([cand(100, 105, {'A': r(1, 2), 'CCCCC': r(4, 5)}), cand(103, 104, {'T': r(1, 2, 3), 'G': r(4, 5, 6)}),
            cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})], reads(7))
  assert graph(dp)[0] == {'110 T', '110 G'}
  # three reference reads make a one-allele site usable
  dp.phase([cand(100, 101, {'C': r(4, 5, 6)}, r(1, 2, 3)), cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})],
           reads(7))
  assert graph(dp)[0] == {'100 REF', '100 C', '110 T', '110 G'}


def test_reuse_object():                                     # :1032-1073
  dp = phaser()
  first = [cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(105, 106, {'C': r(1, 2, 4, 5)}),
           cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5)})]
  assert dp.phase(first, reads(5)) == [1, 1, 1, 2, 2]
  second = [cand(120, 121, {'G': r(1, 2, 3), 'A': r(4, 5)}), cand(130, 131, {'T': r(1, 2, 3, 4, 5)})]
  assert dp.phase(second, reads(5)) == [0, 0, 0, 0, 0]


PV = direct_phasing.PhasedVariant


def test_phased_variants_sanity():                           # :1081-1125
  dp = phaser()
  dp.phase([cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5, 6)}), cand(105, 106, {'C': r(4, 5, 1), 'G': r(2, 3, 6)}),
            cand(110, 111, {'T': r(1, 2, 3), 'G': r(4, 5, 6)})], reads(6))
  assert dp.get_phased_variants() == [PV(100, 'A', 'C', True), PV(105, 'G', 'C', False), PV(110, 'T', 'G', False)]


def test_phased_variants_with_broken_phase():                # NotPhasablePosition + :1127-1183
  dp = phaser()
  cs = [cand(100, 101, {'A': r(1, 2, 3, 10), 'C': r(4, 5)}),
        cand(105, 106, {'C': r(1, 2, 3, 10, 11), 'G': r(4, 5, 12, 13)}),
        cand(110, 111, {'C': r(10, 13), 'G': r(11, 12)}),
        cand(120, 121, {'T': r(6, 7), 'G': r(8, 9)}), cand(125, 126, {'A': r(6, 7), 'T': r(8, 9)})]
  dp.phase(cs, reads(13))
  assert dp.get_phased_variants() == [PV(100, 'A', 'C', True), PV(105, 'C', 'G', False),
                                      PV(120, 'G', 'T', True), PV(125, 'T', 'A', False)]


def test_phased_variants_broken_phase_no_connection():       # :1185-1236
  dp = phaser()
  cs = [cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5, 6)}), cand(105, 106, {'C': r(4, 5, 1), 'G': r(2, 3, 6)}),
        cand(110, 111, {'C': r(7, 8, 9), 'G': r(10, 11, 12)}), cand(120, 121, {'T': r(10, 11, 9), 'G': r(7, 8, 12)})]
  dp.phase(cs, reads(12))
  assert dp.get_phased_variants() == [PV(100, 'A', 'C', True), PV(105, 'G', 'C', False),
                                      PV(110, 'C', 'G', True), PV(120, 'G', 'T', False)]


def test_min_alleles_to_phase_and_empty_inputs():
  cs = [cand(100, 101, {'A': r(1, 2, 3), 'C': r(4, 5)}), cand(110, 111, {'T': r(1, 2), 'G': r(4, 5)})]
  assert direct_phasing.DirectPhasing(2).phase(cs, reads(5)) == [1, 1, 0, 2, 2]   # read3 carries one allele
  assert direct_phasing.DirectPhasing(1).phase(cs, reads(5)) == [1, 1, 1, 2, 2]
  assert direct_phasing.DirectPhasing(1).phase([], reads(3)) == [0, 0, 0]
  assert direct_phasing.DirectPhasing(1).phase(cs, []) == []
