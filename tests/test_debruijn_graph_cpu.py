"""The local-assembly graph (csrc/debruijn_graph.cpp through dv_debruijn_*) against the
reference's own vectors: deepvariant/realigner/python/debruijn_graph_wrap_test.py.  Host only."""
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd.realigner import debruijn_graph
from tests import realigner_fixture as RF


def dbg_options():          # debruijn_graph_wrap_test.py:50-58
  return debruijn_graph.DeBruijnGraphOptions(min_k=12, max_k=50, step_k=2, min_mapq=20, min_base_quality=20,
                                             min_edge_weight=2, max_num_paths=10)


def single_k(k):
  o = dbg_options()
  o.min_k = o.max_k = k
  o.step_k = 1
  return o


def same_graph(expected: str, graph):
  assert ''.join(expected.split()) == ''.join(graph.graphviz().split())


def _read(bases, start=1, quals=None):
  return T.make_read(bases, chrom='chr20', start=start, cigar='%dM' % len(bases),
                     quals=quals or [30] * len(bases), name='read')


BASIC = """digraph G {
  0[label=GAT]; 1[label=ATT]; 2[label=TTA]; 3[label=TAC]; 4[label=ACA]; 5[label=ATG]; 6[label=TGA]; 7[label=GAC];
  0->1 [label=1 color=red]; 1->2 [label=1 color=red]; 2->3 [label=1 color=red]; 3->4 [label=1 color=red];
  0->5 [label=2]; 5->6 [label=2]; 6->7 [label=2]; 7->4 [label=2]; }"""


def test_basics():
  read = _read('GATGACA')
  g = debruijn_graph.build('GATTACA', [read, read], single_k(3))     # two reads: the path survives pruning
  assert sorted(g.candidate_haplotypes()) == ['GATGACA', 'GATTACA']
  assert g.kmer_size == 3
  same_graph(BASIC, g)


def test_pruning_removes_a_path_seen_once():
  g = debruijn_graph.build('GATTACA', [_read('GATGACA')], single_k(3))
  same_graph("""digraph G { 0[label=GAT]; 1[label=ATT]; 2[label=TTA]; 3[label=TAC]; 4[label=ACA];
    0->1 [label=1 color=red]; 1->2 [label=1 color=red]; 2->3 [label=1 color=red]; 3->4 [label=1 color=red]; }""", g)


def test_pruning_removes_edges_not_between_source_and_sink():
  read = _read('CCGATGACACC')
  same_graph(BASIC, debruijn_graph.build('GATTACA', [read, read], single_k(3)))


@pytest.mark.parametrize('bad_position,dropped', [
    (None, set()), (0, {'GA->AT'}), (1, {'GA->AT', 'AT->TT'}), (2, {'GA->AT', 'AT->TT', 'TT->TA'}),
    (3, {'AT->TT', 'TT->TA', 'TA->AC'}), (4, {'TT->TA', 'TA->AC', 'AC->CA'}), (5, {'TA->AC', 'AC->CA'}),
    (6, {'AC->CA'})])
def test_edges_with_bad_positions(bad_position, dropped):
  index = {'GA': 0, 'AT': 1, 'TT': 2, 'TA': 3, 'AC': 4, 'CA': 5}
  dropped = {'%d->%d' % tuple(index[k] for k in e.split('->')) for e in dropped}
  for bad_type in ('qual', 'base'):
    bases, quals = list('GATTACA'), [30] * 7
    if bad_position is not None:
      if bad_type == 'qual':
        quals[bad_position] = 1
      else:
        bases[bad_position] = 'N'
    read = T.make_read(''.join(bases), start=0, cigar='7M', quals=quals)
    g = debruijn_graph.build('GATTACA', [read, read], single_k(2))
    edges = '\n'.join('%s [label=%d color=red];' % (e, 1 if e in dropped else 3)
                      for e in ('0->1', '1->2', '2->3', '3->4', '4->5'))
    same_graph('digraph G { 0[label=GA]; 1[label=AT]; 2[label=TT]; 3[label=TA]; 4[label=AC]; 5[label=CA]; %s }'
               % edges, g)


def test_low_mapq_reads_are_left_out():
  read = _read('GATGACA')
  read.alignment.mapping_quality = 19
  assert debruijn_graph.build('GATTACA', [read, read], single_k(3)).candidate_haplotypes() == ['GATTACA']


def test_lowercase_read_bases_are_upper_cased():
  read = _read('gatgaca')
  assert debruijn_graph.build('GATTACA', [read, read], single_k(3)).candidate_haplotypes() == ['GATGACA', 'GATTACA']


def test_straightforward_region():
  ref, sets = RF.load()
  ref_seq = ref.get_bases('chr20', 9_999_999, 10_000_100)          # chr20:10,000,000-10,000,100
  g = debruijn_graph.build(ref_seq, sets['dbg0'], single_k(30))
  assert g is not None and g.candidate_haplotypes() == [ref_seq]


def test_complex_region():
  ref, sets = RF.load()                                            # het 9 bp deletion in a TGA repeat
  ref_seq = ref.get_bases('chr20', 10_095_378, 10_095_500)
  g = debruijn_graph.build(ref_seq, sets['ex1'], dbg_options())
  assert g is not None and g.kmer_size == 44
  haplotypes = g.candidate_haplotypes()
  assert len(haplotypes) == 2 and ref_seq in haplotypes and haplotypes == sorted(haplotypes)


def test_k_exceeds_read_length():
  read = _read('GATGACA')
  assert debruijn_graph.build('GATTACATG', [read, read], single_k(8)) is not None


def test_k_exceeds_ref_length():
  assert debruijn_graph.build('GATTACA', [], single_k(7)) is None
  assert debruijn_graph.build('GATTACA', [], single_k(8)) is None


@pytest.mark.parametrize('ref,smallest_good_k', [
    ('ACGTACGT', 5), ('ACGTAAACGT', 5), ('ACGTAAACGTAAA', 8), ('AAACGTAAACGT', 7), ('AAACGTAAACGTAAA', 10),
    ('TGGTAAGTTTATAAGGTTATAAGCTGAGAGGTTTTGCTGATCTTGGCTGAGCTCAGCTGGGCAGGTC'
     'TTCCGGTCTTGGCTGGGGTTCACTGACACACAAGCAGCTGACAGTTGGCTGATCTAGGATGGCCTCA'
     'GCTGGG', 11)])
def test_ref_cycle_detector(ref, smallest_good_k):
  for k in range(max(smallest_good_k - 5, 1), min(smallest_good_k + 5, len(ref))):
    g = debruijn_graph.build(ref, [], single_k(k))
    assert (g is None) == (k < smallest_good_k), k


def test_too_many_paths_gives_no_haplotypes():
  # 4 independent SNP bubbles -> 16 walks > max_num_paths 10: CandidatePaths returns nothing
  ref = 'ACGTTGCAAGCTTCGAATGCCGTAAGGCTTACGGATCCTAGGTACCATGG'
  reads = []
  for mask in range(16):
    b = list(ref)
    for bit, pos in enumerate((8, 18, 28, 38)):
      if mask >> bit & 1:
        b[pos] = 'A' if b[pos] != 'A' else 'C'
    reads += [_read(''.join(b), start=0)] * 2
  o = single_k(5)
  assert debruijn_graph.build(ref, reads, o).candidate_haplotypes() == []
  o.max_num_paths = 64
  assert len(debruijn_graph.build(ref, reads, o).candidate_haplotypes()) == 16


def test_disable_graph_pruning_keeps_light_edges():
  o = single_k(3)
  o.disable_graph_pruning = True
  g = debruijn_graph.build('GATTACA', [_read('GATGACA')], o)
  assert g.candidate_haplotypes() == ['GATGACA', 'GATTACA']


def test_bad_options_are_refused():
  from deepvariant_amd import _lib
  o = single_k(3)
  o.step_k = 0
  with pytest.raises(_lib.DvError):
    debruijn_graph.build('GATTACA', [], o)
