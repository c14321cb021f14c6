"""The window realigner's host side (deepvariant_amd/realigner/realigner.py + the native
graph and aligner) against deepvariant/realigner/realigner_test.py: read assignment, flag
handling, the two known-answer regions of chr20 (windows and haplotypes), the realigned
deletion, split_reads, align_to_haplotype, trimming, and the end-to-end invariants.

Window selection needs per-position allele counts; on this GPU-less leg they come from the
oracle counter (tests/realigner_fixture.OracleAlleleCounter) -- everything else is the product
code.  tests/test_hip_realigner.py runs the same regions with the device counter."""
import itertools

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd.realigner import realigner
from deepvariant_amd.realigner import utils
from deepvariant_amd.realigner import window_selector as ws
from tests import realigner_fixture as RF


def parse_literal(s):          # ranges.parse_literal: 1-based inclusive text -> 0-based half-open
  contig, rest = s.split(':')
  a, b = rest.replace(',', '').split('-')
  return T.Range(contig, int(a) - 1, int(b))


def cigar_text(cigar):
  return ''.join('%d%s' % (c.operation_length, 'XMIDNSHP=X'[c.operation]) for c in cigar)


def assembled(region, haplotypes=None):
  return realigner.AssemblyRegion(realigner.CandidateHaplotypes(parse_literal(region), haplotypes or []))


@pytest.fixture(scope='module')
def fixture():
  return RF.load()


@pytest.fixture(autouse=True)
def _oracle_counter():
  """No GPU here: the window selector's allele counts come from the oracle's counter."""
  with RF.oracle_allele_counter():
    yield


def make_realigner(ref, **flags):
  return realigner.Realigner(realigner.realigner_config(**flags), ref)


# ---------------------------------------------------------------- ReadAssignmentTests, :77-194
def _assignment_reads():
  reads = [T.make_read('ACG', start=1, cigar='3M', name='read1'), T.make_read('ACG', start=6, cigar='3M', name='read2'),
           T.make_read('ACG', start=9, cigar='3M', name='read3'), T.make_read('ACG', start=28, cigar='3M', name='read4'),
           T.make_read('A' * 10, start=3, cigar='10M', name='read5')]
  return {r.fragment_name: r for r in reads}


def test_assembly_region_construction_and_read_span():
  a = assembled('chr1:1-5', ['A', 'C'])
  assert a.region == parse_literal('chr1:1-5') and a.haplotypes == ['A', 'C'] and a.reads == []
  reads = _assignment_reads()
  a = assembled('chr1:3-15')
  assert a.read_span is None
  a.add_read(reads['read2'])
  assert a.read_span == parse_literal('chr1:7-9')
  a.add_read(reads['read1'])
  assert a.read_span == parse_literal('chr1:2-9')
  for n in ('read3', 'read4', 'read5'):
    a.add_read(reads[n])
  assert a.read_span == parse_literal('chr1:2-31') and len(a.reads) == 5


@pytest.mark.parametrize('names', list(itertools.permutations(['read1', 'read2', 'read3', 'read4', 'read5']))[::7])
def test_assign_reads_to_assembled_regions(names):
  reads = _assignment_reads()
  regions = {'r1': assembled('chr1:1-5'), 'r2': assembled('chr1:10-15'), 'r3': assembled('chr1:20-30')}
  unassigned = realigner.assign_reads_to_assembled_regions([regions[k] for k in sorted(regions)],
                                                           [reads[n] for n in names])
  got = {k: sorted(r.fragment_name for r in v.reads) for k, v in regions.items()}
  # read2 falls between r1 and r2; read5 overlaps r1 and r2 but r2 more
  assert got == {'r1': ['read1'], 'r2': ['read3', 'read5'], 'r3': ['read4']}
  assert [r.fragment_name for r in unassigned] == ['read2']


def test_find_max_overlapping_ties_go_to_the_first():
  q = T.Range('chr1', 0, 10)
  assert utils.find_max_overlapping(q, [T.Range('chr1', 8, 20), T.Range('chr1', 0, 2), T.Range('chr2', 0, 10)]) == 0
  assert utils.find_max_overlapping(q, [T.Range('chr1', 10, 20)]) is None
  assert utils.find_max_overlapping(q, []) is None


# ---------------------------------------------------------------- flags, :205-295
def test_window_selector_model_flags():
  c = realigner.realigner_config(ws_min_num_supporting_reads=2, ws_max_num_supporting_reads=300)
  m = c.ws_config.window_selector_model
  assert m.model_type == ws.VARIANT_READS
  assert (m.variant_reads_model.min_num_supporting_reads, m.variant_reads_model.max_num_supporting_reads) == (2, 300)
  assert realigner.realigner_config().ws_config.window_selector_model == m      # the defaults
  m = realigner.realigner_config(ws_use_window_selector_model=True).ws_config.window_selector_model
  assert m.model_type == ws.ALLELE_COUNT_LINEAR and m.allele_count_linear_model.decision_boundary == 3
  custom = ws.WindowSelectorModel(model_type=ws.VARIANT_READS,
                                  variant_reads_model=ws.VariantReadsThresholdModel(1, 5))
  c = realigner.realigner_config(ws_use_window_selector_model=True, ws_window_selector_model=custom)
  assert c.ws_config.window_selector_model is custom
  assert (c.ws_config.min_allele_support, c.ws_config.min_mapq, c.ws_config.min_windows_distance,
          c.ws_config.max_window_size, c.ws_config.region_expansion_in_bp) == (2, 20, 80, 1000, 20)
  assert (c.dbg_config.min_k, c.dbg_config.max_k, c.dbg_config.min_mapq, c.dbg_config.min_base_quality,
          c.dbg_config.min_edge_weight, c.dbg_config.max_num_paths) == (10, 101, 14, 15, 2, 256)
  assert (c.aln_config.match, c.aln_config.mismatch, c.aln_config.gap_open, c.aln_config.gap_extend,
          c.aln_config.kmer_size, c.aln_config.max_num_of_mismatches) == (4, 6, 8, 2, 32, 2)


@pytest.mark.parametrize('flags,message', [
    (dict(ws_min_num_supporting_reads=2, ws_max_num_supporting_reads=1), 'should be smaller'),
    (dict(ws_window_selector_model=ws.WindowSelectorModel()), 'Cannot specify a ws_window_selector_model'),
    (dict(ws_use_window_selector_model=True, ws_min_num_supporting_reads=1), 'Cannot use both ws_min_num'),
    (dict(ws_use_window_selector_model=True, ws_max_num_supporting_reads=1), 'Cannot use both ws_max_num'),
    (dict(no_such_flag=1), 'unknown realigner flag')])
def test_window_selector_model_flags_failures(flags, message):
  with pytest.raises(ValueError, match=message):
    realigner.realigner_config(**flags)


# ---------------------------------------------------------------- known-answer regions, :296-392
EXAMPLE_REGIONS = [
    ('ex1', 'chr20:10,095,379-10,095,500', 'chr20:10,095,352-10,095,553', {     # het 9 bp deletion in a TGA repeat
        'TAGTGATCTAGTCCTTTTTGTTGTGCAAAAGGAAGTGCTAAAATCAGAATGAGAACCATGGTCA'
        'CCTGACATAGACACAAGTGATGATGATGATGATGATGATGATGATGATGATGATATCCATGTTC'
        'AAGTACTAATTCTGGGCAAGACACTGTTCTAAGTGCTATGAATATATTACCTCATTTAATCATC'
        'T',
        'TAGTGATCTAGTCCTTTTTGTTGTGCAAAAGGAAGTGCTAAAATCAGAATGAGAACCATGGTCA'
        'CCTGACATAGACACAAGTGATGATGATGATGATGATGATGATGATGATGATGATGATGATGATA'
        'TCCATGTTCAAGTACTAATTCTGGGCAAGACACTGTTCTAAGTGCTATGAATATATTACCTCAT'
        'TTAATCATCT'}),
    ('ex2', 'chr20:10,046,080-10,046,307', 'chr20:10,046,096-10,046,267', {     # het 10 bp deletion
        'CCCAAAAAAAGAGTTAGGGATGCTGGAAAGGCAGAAAGAAAAGGGAAGGGAAGAGGAAGGGGAA'
        'AAGGAAAGAAAAAAAAGAAAGAAAGAAAGAGAAAGAAAGAGAAAGAGAAAGAAAGAGGAAAGAG'
        'AGAAAGAGAAAGAGAAGGAAAGAGAAAGAAAGAGAAGGAAAGAG',
        'CCCAAAAAAAGAGTTAGGGATGCTGGAAAGGCAGAAAGAAAAGGGAAGGGAAGAGGAAGGGGAA'
        'AAGGAAAGAAAAAAAAGAAAGAAAGAAAGAGAAAGAGAAAGAAAGAGGAAAGAGAGAAAGAGAA'
        'AGAGAAGGAAAGAGAAAGAAAGAGAAGGAAAGAG'})]


@pytest.mark.parametrize('name,region,window,haplotypes', EXAMPLE_REGIONS)
def test_realigner_example_region(fixture, name, region, window, haplotypes):
  ref, sets = fixture
  r = make_realigner(ref, ws_use_window_selector_model=True)      # the reference test's setUp
  windows_haplotypes, realigned = r.realign_reads(sets[name], parse_literal(region))
  assert len(realigned) == len(sets[name])
  assert windows_haplotypes[0].span == parse_literal(window)
  assert set(windows_haplotypes[0].haplotypes) == haplotypes


def test_realigner_example_variant(fixture):
  """Every realigned read that spans chr20:10,046,179-10,046,188 carries the 10 bp deletion."""
  ref, sets = fixture
  variant = parse_literal('chr20:10,046,179-10,046,188')
  r = make_realigner(ref, ws_use_window_selector_model=True)
  _, realigned = r.realign_reads(sets['ex2'], parse_literal('chr20:10,046,080-10,046,307'))
  spanning = 0
  for read in realigned:
    ref_pos = read.alignment.position.position
    has_variant = False
    for c in read.alignment.cigar:
      assert c.operation in utils.CIGAR_OPS
      if c.operation in utils.CIGAR_ALIGN_OPS:
        ref_pos += c.operation_length
      elif c.operation in utils.CIGAR_DELETE_OPS:
        if ref_pos == variant.start and c.operation_length == variant.end - ref_pos:
          has_variant = True
        ref_pos += c.operation_length
    if read.alignment.position.position <= variant.start and ref_pos >= variant.end:
      spanning += 1
      assert has_variant, read.fragment_name
  assert spanning > 20


def test_realigner_doesnt_create_invalid_intervals(fixture):
  """Reads at the very end of the contig (reference all N there), :394-428."""
  ref, _ = fixture
  r = make_realigner(ref, ws_use_window_selector_model=True)
  region = parse_literal('chr20:63,025,320-63,025,520')
  quals = list(np.tile(range(30, 35), 50))
  reads = [T.make_read('ACCGT' * 50, start=63025520 - 250, cigar='250M', quals=quals, chrom='chr20')
           for _ in range(20)]
  assert len(r.realign_reads(reads, region)[1]) == 20
  reads = [T.make_read('TTATA' * 50, start=63025520 - 200, cigar='200M50S', quals=quals, chrom='chr20')
           for _ in range(20)]
  assert len(r.realign_reads(reads, region)[1]) == 20


def test_realigner_end2end(fixture):
  """RealignerIntegrationTest (:728-760) over chr20:10,000,000-10,009,999 in 1000-base
  partitions: every read comes back, the reference is one of each window's haplotypes."""
  ref, sets = fixture
  r = make_realigner(ref)
  reads = sets['wgs']
  spans = [utils.read_range(x) for x in reads]
  n_windows = n_changed = 0
  for start in range(9_999_999, 10_009_999, 1000):
    region = T.Range('chr20', start, min(start + 1000, 10_009_999))
    in_reads = [x for x, s in zip(reads, spans) if utils.ranges_overlap(s, region)]
    windows, out_reads = r.realign_reads(in_reads, region)
    assert sorted(x.fragment_name for x in in_reads) == sorted(x.fragment_name for x in out_reads)
    for w in windows:
      assert ref.get_bases('chr20', w.span.start, w.span.end) in set(w.haplotypes)
      assert w.haplotypes == sorted(w.haplotypes) and len(w.haplotypes) > 1
    n_windows += len(windows)
    before = {(x.fragment_name, x.read_number): x for x in in_reads}
    for x in out_reads:
      old = before[(x.fragment_name, x.read_number)]
      assert x.aligned_sequence == old.aligned_sequence
      n_read = sum(c.operation_length for c in x.alignment.cigar if c.operation in utils.READ_ADVANCING_OPS)
      assert n_read == len(x.aligned_sequence)
      n_changed += x.alignment != old.alignment
  assert n_windows == 26 and n_changed > 100


def test_no_reads_and_oversized_windows(fixture):
  ref, sets = fixture
  r = make_realigner(ref)
  assert r.realign_reads([], T.Range('chr20', 10_000_000, 10_001_000)) == ([], [])
  # a window wider than ws_max_window_size is skipped: nothing is assembled, all reads come back unchanged
  r = make_realigner(ref, ws_use_window_selector_model=True, ws_max_window_size=50)
  windows, out = r.realign_reads(sets['ex1'], parse_literal('chr20:10,095,379-10,095,500'))
  assert windows == [] and out == sets['ex1']


# ---------------------------------------------------------------- align_to_haplotype, :497-603
@pytest.mark.parametrize('read_seq,prefix,suffix,haplotypes,expected', [
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCATGGATCCATGTTCAAGTACTAATTCTGGGC',
     'AGTGATCTAGTCCTTTTTGTTGTGCAAAAGGAAGTGCTAAAATCAGAATGAGAACCATGG',
     'ATCCATGTTCAAGTACTAATTCTGGGCAAGACACTGTTCTAAGTGCTATGAATATATTACC', ['CATCATCAT', ''], ['33M9D27M', '60M']),
    ('TTGCCCGGGCATAAGGTGTTTCGGAGAAGCCTAG' 'TATATATA' 'CTCCGGTTTTTAAGTAGGGTCGTAGCAG',
     'AACGGGTCTACAAGTCTCTGCGTGTTGCCCGGGCATAAGGTGTTTCGGAGAAGCCTAG',
     'CTCCGGTTTTTAAGTAGGGTCGTAGCAGCAAAGTAAGAGTGGAACGCGTGGGCGACTA', ['', 'TATATATA'], ['34M8I28M', '70M']),
    ('AAAAAAAAAAGGGGGGGGGGATTTTTTTTTTTTTCCCCCCCCCCCCCCC', 'AAAAAAAAAAGGGGGGGGGG', 'TTTTTTTTTTTTTCCCCCCCCCCCCCCC',
     ['A', ''], ['49M', '20M1I28M'])])
def test_align_to_haplotype(fixture, read_seq, prefix, suffix, haplotypes, expected):
  r = make_realigner(fixture[0], ws_use_window_selector_model=True)
  reads = [T.make_read(read_seq, start=1)]
  for h, want in zip(haplotypes, expected):
    aligned = r.align_to_haplotype(h, haplotypes, prefix, suffix, reads, 'test', 1)
    assert len(aligned) == 1 and cigar_text(aligned[0].alignment.cigar) == want


@pytest.mark.parametrize('alt_allele,ref_buffer,read_buffer', [('CATTACA', 70, 20), ('CATTACA', 20, 20), ('G', 70, 20)])
def test_align_to_haplotype_stress(fixture, alt_allele, ref_buffer, read_buffer):
  whole_prefix = 'AGTGATCTAGTCCTTTTTGTTGTGCAAAAGGAAGTGCTAAAATCAGAATGAGAACCATGGTCACCTGACATAGAC'
  whole_suffix = 'ATCCATGTTCAAGTACTAATTCTGGGCAAGACACTGTTCTAAGTGCTATGAATATATTACCTCATTTAATCATCT'
  ref_prefix, ref_suffix = whole_prefix[-ref_buffer:], whole_suffix[:ref_buffer]
  read_prefix, read_suffix = ref_prefix[-read_buffer:], ref_suffix[:read_buffer]
  haplotypes = ['', alt_allele]
  expected = ['%dM%dI%dM' % (len(read_prefix), len(alt_allele), len(read_suffix)),
              '%dM' % (len(read_prefix) + len(alt_allele) + len(read_suffix))]
  r = make_realigner(fixture[0], ws_use_window_selector_model=True)
  reads = [T.make_read(read_prefix + alt_allele + read_suffix, start=1)]
  for h, want in zip(haplotypes, expected):
    aligned = r.align_to_haplotype(h, haplotypes, ref_prefix, ref_suffix, reads, 'test', 1)
    assert cigar_text(aligned[0].alignment.cigar) == want


def test_align_to_haplotype_empty_reads(fixture):
  r = make_realigner(fixture[0])
  assert r.align_to_haplotype('G', ['G', ''], 'AAA', 'AAA', [], 'test', 1) == []


# ---------------------------------------------------------------- split_reads, :605-725
@pytest.mark.parametrize('read_seq,cigar,cigars,sequences,positions', [
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCA', '30M', ['30M'], ['AAGGAAGTGCTAAAATCAGAATGAGAACCA'], [1]),
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCA', '15M5000N15M', ['15M', '15M'], ['AAGGAAGTGCTAAAA', 'TCAGAATGAGAACCA'], [1, 5016]),
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCA', '10M10N20M', ['20M'], ['TAAAATCAGAATGAGAACCA'], [21]),
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCA', '5M5N5M5N5M5N5M5N5M5N5M', [], [], []),
    ('AAGGAAGTGCTAAAATCAGAATGAGAACCA', '2M5000N28M', ['28M'], ['GGAAGTGCTAAAATCAGAATGAGAACCA'], [5003]),
    ('AAGGAAGTGCTAATTTTTAATCAGAATGAGAACCA', '15M5I15M', ['15M5I15M'], ['AAGGAAGTGCTAATTTTTAATCAGAATGAGAACCA'], [1]),
    ('AAGGAAGTGCTAAAAGGGGGTCAGAATGAGAACCA', '15M5I50N15M', ['15M5I', '15M'],
     ['AAGGAAGTGCTAAAAGGGGG', 'TCAGAATGAGAACCA'], [1, 66]),
    ('AAGGAAGTGCTAATTTTTAATCAGAATGAGAACCA', '15M5D15M', ['15M5D15M'], ['AAGGAAGTGCTAATTTTTAATCAGAATGAGAACCA'], [1]),
    ('AAGGAAGTGCTAATTTCAGAATGAGAACCA', '15M5D50N15M', ['15M5D', '15M'], ['AAGGAAGTGCTAATT', 'TCAGAATGAGAACCA'], [1, 71]),
    ('CCCCGGACACTTCTAGTTTGTCGGAGCGAGTC', '15=1X1=20N15=', ['15=1X1=', '15='],
     ['CCCCGGACACTTCTAGT', 'TTGTCGGAGCGAGTC'], [1, 38]),
    ('TGAGCTAGTAGAATTTAGGGAGAAAGATTAATGCG', '15S5M50N15M', ['15S5M', '15M'],
     ['TGAGCTAGTAGAATTTAGGG', 'AGAAAGATTAATGCG'], [1, 56]),
    ('ATCCCGGCCACGTTAATCCCGGCCACGTTA', '15H15M50N15M15H', ['15H15M', '15M15H'],
     ['ATCCCGGCCACGTTA', 'ATCCCGGCCACGTTA'], [1, 66])])
def test_split_reads(read_seq, cigar, cigars, sequences, positions):
  read = T.make_read(read_seq, cigar=cigar, start=1, quals=list(range(len(read_seq))), name='r')
  parts = realigner.split_reads([read])
  assert [p.aligned_sequence for p in parts] == sequences
  assert [cigar_text(p.alignment.cigar) for p in parts] == cigars
  assert [p.alignment.position.position for p in parts] == positions
  for p in parts:
    assert len(p.aligned_quality) == len(p.aligned_sequence)
    if 'N' in cigar:
      assert p.fragment_name.startswith('r_p') and p.alignment.mapping_quality == read.alignment.mapping_quality
  assert cigar_text(read.alignment.cigar) == cigar      # the input is left alone


def test_split_skip_reads_flag_is_applied(fixture):
  ref, sets = fixture
  r = make_realigner(ref, split_skip_reads=True)
  read = T.make_read('A' * 30, cigar='15M5000N15M', start=10_000_100, chrom='chr20', quals=[30] * 30, name='s')
  _, out = r.realign_reads([read], T.Range('chr20', 10_000_000, 10_001_000))
  assert sorted(x.fragment_name for x in out) == ['s_p0', 's_p1']


# ---------------------------------------------------------------- TrimTest, :763-960
@pytest.mark.parametrize('cigar,ref_trim,ref_length,want_cigar,want_trim,want_length', [
    ('3M2D5M3I10M', 6, 9, '4M3I5M', 4, 12), ('30M', 5, 10, '10M', 5, 10), ('10D10M', 5, 10, '5D5M', 0, 5),
    ('10I10M', 5, 5, '5M', 15, 5), ('10M', 5, 10, '5M', 5, 5), ('10M', 20, 10, '', 10, 0),
    ('10M20D10M', 12, 5, '5D', 10, 0), ('10M20I10M', 10, 20, '20I10M', 10, 30), ('10M2I10M', 0, 20, '10M2I10M', 0, 22)])
def test_trim_cigar(cigar, ref_trim, ref_length, want_cigar, want_trim, want_length):
  read = T.make_read('AAAATAAAATAAAATAAAATA', start=100, cigar=cigar)
  out, trim, length = realigner.trim_cigar(read.alignment.cigar, ref_trim, ref_length)
  assert (cigar_text(out), trim, length) == (want_cigar, want_trim, want_length)
  assert cigar_text(read.alignment.cigar) == cigar


@pytest.mark.parametrize('cigar,start,n,want_cigar,want_pos,want_n', [
    ('9M', 8, 9, '7M', 10, 7), ('9M', 13, 9, '7M', 13, 7), ('5M', 12, 5, '5M', 12, 5), ('9M', 10, 9, '9M', 10, 9)])
def test_trim_read(cigar, start, n, want_cigar, want_pos, want_n):
  read = T.make_read('A' * n, start=start, cigar=cigar, quals=[30] * n)
  out = realigner.trim_read(read, parse_literal('chr1:11-20'))
  assert cigar_text(out.alignment.cigar) == want_cigar and out.alignment.position.position == want_pos
  assert len(out.aligned_sequence) == want_n == len(out.aligned_quality)
