"""deepvariant_amd/alt_aligned_pileup_lib.py against the reference's own vectors:
  TrimCigar / TrimRead            deepvariant/alt_aligned_pileup_lib_test.cc:125-247
  FillPileupArray (5 layouts)     deepvariant/pileup_image_native_test.cc:439-660
plus CreateHaplotype / NeedAltAlignment / GetAltImageRowIndices restated from
make_examples_native.cc:269-297,500-512 and pileup_image_native.cc:193-209."""
import numpy as np
import pytest

from deepvariant_amd import alt_aligned_pileup_lib as A
from deepvariant_amd import dv_types as T

_OPS = {'M': 1, 'I': 2, 'D': 3, 'N': 4, 'S': 5, 'H': 6, 'P': 7, '=': 8, 'X': 9}


def _cigar(elems):
  return [T.CigarUnit(_OPS[e[-1]], int(e[:-1])) for e in elems]


def _str(cigar):
  inv = {v: k for k, v in _OPS.items()}
  return ['%d%s' % (u.operation_length, inv[u.operation]) for u in cigar]


@pytest.mark.parametrize('ref_start,ref_length,cigar,want,read_start,read_length', [
    (10, 20, ['20M', '5I', '10M'], ['10M', '5I', '10M'], 10, 25),   # INS
    (10, 20, ['20M', '5D', '10M'], ['10M', '5D', '5M'], 10, 15),    # DEL
    (22, 10, ['20M', '5I', '20M'], ['10M'], 27, 10),                 # start falls into INS
    (22, 10, ['20M', '5D', '20M'], ['3D', '7M'], 20, 7),             # start falls into DEL
    (50, 20, ['20M', '5I', '10M'], [], 35, 0),                       # start beyond the read
    (10, 40, ['20M', '5I', '10M'], ['10M', '5I', '10M'], 10, 25),    # window beyond the read
])
def test_trim_cigar_reference_vectors(ref_start, ref_length, cigar, want, read_start, read_length):
  got, rs, rl = A.trim_cigar(_cigar(cigar), ref_start, ref_length)
  assert (_str(got), rs, rl) == (want, read_start, read_length)


_BASES = 'ACGTACGTAAAAAAGTGTGATC'
_QUALS = list(range(1, 23))


@pytest.mark.parametrize('trim_start,trim_len,cigar,want_pos,want_bases,want_cigar,want_quals', [
    (15, 5, ['22M'], 15, 'CGTAA', ['5M'], [6, 7, 8, 9, 10]),
    (15, 5, ['2M', '3I', '17M'], 15, 'AAAAA', ['5M'], [9, 10, 11, 12, 13]),
    (15, 5, ['2M', '3D', '20M'], 15, 'GTACG', ['5M'], [3, 4, 5, 6, 7]),
    (8, 5, ['22M'], 10, 'ACG', ['3M'], [1, 2, 3]),
    (10, 22, ['22M'], 10, _BASES, ['22M'], _QUALS),
])
def test_trim_read_reference_vectors(trim_start, trim_len, cigar, want_pos, want_bases,
                                     want_cigar, want_quals):
  read = T.Read(fragment_name='test_read', aligned_sequence=_BASES, aligned_quality=bytes(_QUALS),
                alignment=T.LinearAlignment(position=T.Position('chr1', 10, False),
                                            mapping_quality=90, cigar=_cigar(cigar)),
                base_modifications={'5mC': bytes(range(100, 122))})
  got = A.trim_read(read, trim_start, trim_start + trim_len)
  assert got.alignment.position.position == want_pos
  assert got.aligned_sequence == want_bases
  assert _str(got.alignment.cigar) == want_cigar
  assert list(got.aligned_quality) == want_quals
  assert got.fragment_name == 'test_read' and got.alignment.mapping_quality == 90
  # base modifications follow the same slice (alt_aligned_pileup_lib.cc:166-172)
  assert list(got.base_modifications['5mC']) == [q + 99 for q in want_quals]


def test_trim_reads_drops_short_overlaps_and_keeps_original_starts():
  def mk(pos, n, name):
    return T.Read(fragment_name=name, aligned_sequence='A' * n, aligned_quality=bytes([30] * n),
                  alignment=T.LinearAlignment(position=T.Position('chr1', pos, False),
                                              mapping_quality=60, cigar=_cigar(['%dM' % n])))
  reads = [mk(0, 110, 'long'), mk(90, 24, 'edge14'), mk(95, 40, 'edge15'), mk(120, 50, 'inside')]
  trimmed, starts = A.trim_reads(reads, 100, 200)
  assert [r.fragment_name for r in trimmed] == ['edge15', 'inside']    # 10 and 14 bp overlaps dropped
  assert starts == [95, 120]
  assert trimmed[0].alignment.position.position == 100 and len(trimmed[0].aligned_sequence) == 35
  with pytest.raises(ValueError, match='ref_length > 0'):
    A.trim_read(mk(300, 10, 'beyond'), 100, 200)
  # keep_only_window_spanning_reads: min_overlap = image width (make_examples_native.cc:667-671)
  spanning, _ = A.trim_reads(reads + [mk(50, 400, 'span')], 100, 200, min_overlap=100)
  assert [r.fragment_name for r in spanning] == ['span']


class _Ref:
  seq = 'TTTTTTTTTTACGTACGTAAAAAAGTGTGATCCCCCCCCCCCC'    # alt_aligned_pileup_lib_test.cc:278

  def n_bases(self, contig):
    return len(self.seq)

  def get_bases(self, contig, start, end):
    return self.seq[start:end]


def test_create_haplotype_and_alignment_region():
  v = T.Variant('chr1', 20, 21, 'C', ['CGGG'])
  hap, s, e = A.create_haplotype(_Ref(), v, 'CGGG', half_width=8)
  assert (s, e) == (12, 29)
  assert hap == _Ref.seq[12:20] + 'CGGG' + _Ref.seq[21:29]
  # clipped at both contig ends
  hap, s, e = A.create_haplotype(_Ref(), T.Variant('chr1', 3, 5, 'TT', ['T']), 'T', half_width=8)
  assert (s, e) == (0, 13) and hap == _Ref.seq[0:3] + 'T' + _Ref.seq[5:13]
  hap, s, e = A.create_haplotype(_Ref(), T.Variant('chr1', 40, 41, 'C', ['A']), 'A', half_width=8)
  assert (s, e) == (32, 43) and hap == _Ref.seq[32:40] + 'A' + _Ref.seq[41:43]
  assert A.calculate_alignment_region(v, 8, 43) == (12, 29)
  assert A.calculate_alignment_region(T.Variant('chr1', 3, 5, 'TT', ['T']), 8, 43) == (0, 13)


def test_need_alt_alignment():
  pic = T.default_options()
  snp, ins, mnp = (T.Variant('chr1', 5, 6, 'A', ['C']), T.Variant('chr1', 5, 6, 'A', ['ACC']),
                   T.Variant('chr1', 5, 7, 'AC', ['A']))
  assert not any(A.need_alt_alignment(pic, v) for v in (snp, ins, mnp))        # 'none'
  pic.alt_aligned_pileup = 'diff_channels'
  assert [A.need_alt_alignment(pic, v) for v in (snp, ins, mnp)] == [False, True, True]
  pic.types_to_alt_align = 'all'
  assert all(A.need_alt_alignment(pic, v) for v in (snp, ins, mnp))


def _images():
  """pileup_image_native_test.cc:439-520: three rows, seven channels, width five."""
  ref = np.zeros((3, 5, 7), np.uint8)
  for r in range(3):
    for c in range(7):
      ref[r, :, c] = 10 * (c + 1) + r + 1
  alts = []
  for delta in (-5, 5):
    a = np.ones((3, 5, 7), np.uint8)
    for r in range(3):
      a[r, :, 0] = 11 + r + delta
      a[r, :, 5] = 61 + r + delta
    alts.append(a)
  return ref, alts


def test_fill_pileup_array_reference_layouts():
  ref, alts = _images()
  np.testing.assert_array_equal(A.fill_pileup_array(ref, alts, A.NONE), ref)
  for mode, ch in ((A.BASE_CHANNELS, 0), (A.DIFF_CHANNELS, 5)):
    got = A.fill_pileup_array(ref, alts, mode)
    assert got.shape == (3, 5, 9)
    np.testing.assert_array_equal(got[:, :, :7], ref)
    for r in range(3):      # expected "base_channels" / "diff_channels" tables of the reference test
      assert (got[r, :, 7] == 10 * (ch + 1) + r + 1 - 5).all()
      assert (got[r, :, 8] == 10 * (ch + 1) + r + 1 + 5).all()
  np.testing.assert_array_equal(A.fill_pileup_array(ref, alts, A.ROWS, [0, 1]),
                                np.concatenate([ref, alts[0], alts[1]]))
  np.testing.assert_array_equal(A.fill_pileup_array(ref, alts, A.SINGLE_ROW, [0]),
                                np.concatenate([ref, alts[0]]))
  np.testing.assert_array_equal(A.fill_pileup_array(ref, alts, A.SINGLE_ROW, [1]),
                                np.concatenate([ref, alts[1]]))
  # missing alt images (pileup_image_native.h:246-271,275-288)
  got = A.fill_pileup_array(ref, [alts[0], None], A.DIFF_CHANNELS)
  assert (got[:, :, 8] == got[:, :, 7]).all() and (got[0, :, 7] == 56).all()
  got = A.fill_pileup_array(ref, [None, None], A.DIFF_CHANNELS)
  assert not got[:, :, 7:].any()
  got = A.fill_pileup_array(ref, [alts[0], None], A.ROWS, [0, 1])
  assert got.shape == (9, 5, 7) and not got[6:].any()


def test_alt_image_row_indices_and_by_sample():
  assert A.get_alt_image_row_indices(A.ROWS, ['A']) == [0, 1]
  assert A.get_alt_image_row_indices(A.SINGLE_ROW, ['A', 'ACG']) == [1]   # the longer allele
  assert A.get_alt_image_row_indices(A.SINGLE_ROW, ['ACG', 'A']) == [0]
  assert A.get_alt_image_row_indices(A.SINGLE_ROW, ['A']) == [0]
  assert A.get_alt_image_row_indices(A.DIFF_CHANNELS, ['A']) == []
  with pytest.raises(ValueError):
    A.get_alt_aligned_pileup('bogus')
  ref, alts = _images()
  opts = T.MakeExamplesOptions(pic_options=T.default_options(), sample_options=[
      T.SampleOptions(role='a', pileup_height=3), T.SampleOptions(role='b', pileup_height=3,
                                                                  alt_aligned_pileup='single_row')])
  opts.pic_options.alt_aligned_pileup = 'rows'
  got = A.fill_pileup_array_by_sample([ref, ref], [alts, alts], opts, ['A', 'ACG'])
  np.testing.assert_array_equal(got, np.concatenate([ref, alts[0], alts[1], ref, alts[1]]))
