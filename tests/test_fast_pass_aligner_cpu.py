"""The read realigner against the reference's own test vectors.

  deepvariant/realigner/fast_pass_aligner_test.cc   (every TEST_F, same inputs / expectations)
  deepvariant/realigner/python/ssw_wrap_test.py, ssw_misc_test.py, ssw_test.cc
The local aligner restates libssw v1.2.5 (a dependency that is not in the reference tree);
these vectors pin its tie-breaking: end / begin selection, gap placement, '=' / 'X' / 'S' text.
"""
import pytest

from deepvariant_amd import fast_pass_aligner as F

REF = 'ATCAAGGGAAAAAGTGCCCAGGGCCAAATATGTTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTGAGCTGAAGATATG'
REF2 = 'CTCTGTAATCGGATCATGTTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTGAGCTGAAGATATG'
RA = F.ReadAlignment
MATCH, MISMATCH = 4, 6     # class defaults (fast_pass_aligner.h:384-387)


def _aligner(reference=REF, **kw):
  a = F.FastPassAligner(**kw)
  a.set_reference(reference)
  return a


# ---------------------------------------------------------------- local aligner
def test_local_align_wrap_vectors():
  ref, query = 'CAGCCTTTCTGACCCGGAAATCAAAATAGGCACAACAAA', 'CTGAGCCGGTAAATC'
  r = F.local_align(ref, query)
  assert (r.score, r.ref_begin, r.ref_end, r.query_begin, r.query_end, r.mismatches) == (21, 8, 21, 0, 14, 2)
  assert r.cigar == b'4=1X4=1I5='
  r = F.local_align(query, ref)
  assert (r.score, r.query_begin, r.query_end, r.ref_begin, r.ref_end, r.mismatches) == (21, 8, 21, 0, 14, 2)
  assert r.cigar == b'8S4=1X4=1D5=17S'


def test_local_align_short_and_longer():
  assert F.local_align('tttt', 'ttAtt', 4, 2, 4, 2).cigar == b'2=1I2='
  assert F.local_align('TTTTGGGGGGGGGGGGG', 'TTATTGGGGGGGGGGGGG', 4, 2, 4, 2).cigar == b'2=1I15='


def test_ssw_aligner_sanity_check():
  r = F.local_align('TTTGCCGAAGTTAAACCC', 'GCCGAAGTTA', 4, 6, 8, 1)
  assert r.cigar == b'10=' and r.ref_begin == 3


# ---------------------------------------------------------------- index
def test_reads_index_integration():
  a = _aligner(kmer_size=3)
  a.set_reads(['AAACCC', 'CTCTCT', 'TGAGCTGAAG'])
  a.stage(F.BUILD_INDEX)
  want = {'AAA': [(0, 0)], 'AAC': [(0, 1)], 'ACC': [(0, 2)], 'CCC': [(0, 3)],
          'CTC': [(1, 0), (1, 2)], 'TCT': [(1, 1), (1, 3)], 'TGA': [(2, 0), (2, 5)],
          'GAG': [(2, 1)], 'AGC': [(2, 2)], 'GCT': [(2, 3)], 'CTG': [(2, 4)], 'GAA': [(2, 6)],
          'AAG': [(2, 7)]}
  assert a.index_size() == len(want)
  for kmer, occ in want.items():
    assert a.kmer_occurrences(kmer) == occ, kmer


def test_reads_index_ignores_reads_shorter_than_kmer():
  a = _aligner(kmer_size=4)
  a.set_reads(['AAC', 'TGAGCTG'])
  a.stage(F.BUILD_INDEX)
  assert a.index_size() == 4
  for i, kmer in enumerate(['TGAG', 'GAGC', 'AGCT', 'GCTG']):
    assert a.kmer_occurrences(kmer) == [(1, i)]


# ---------------------------------------------------------------- fast pass
def _fast(reads, haplotype, **kw):
  a = _aligner(kmer_size=3, **kw)
  a.set_reads(reads)
  a.stage(F.BUILD_INDEX)
  return a.fast_align_reads_to_haplotype(haplotype)


def test_fast_align_reads_to_haplotype():
  score, got = _fast(['AAACCC', 'CTCTCT', 'TGAGCTGAAG'], 'TGAGCTGAAGTTAAACCC')
  assert score == 10 * MATCH + 6 * MATCH
  assert got == [RA(12, '6=', 6 * MATCH), RA(), RA(0, '10=', 10 * MATCH)]


def test_fast_align_partial_read_overlap():
  score, got = _fast(['TGAGCTGAAGTT', 'AAACCC', 'AGTTAAAC'], 'TGAGCTGAAGTTAAAC')
  assert score == 12 * MATCH + 8 * MATCH
  assert got == [RA(0, '12=', 12 * MATCH), RA(), RA(8, '8=', 8 * MATCH)]


def test_fast_align_with_one_mismatch():
  score, got = _fast(['AAACCC', 'CTCTCT', 'TGAGCTGAAG'], 'TGAGCCGAAGTTAAACCC')
  assert score == 9 * MATCH - MISMATCH + 6 * MATCH
  assert got == [RA(12, '6=', 6 * MATCH), RA(), RA(0, '10=', 9 * MATCH - MISMATCH)]


def test_fast_align_with_more_than_allowed_mismatches():
  score, got = _fast(['TTTGCCGAAGTTAAACCC', 'CTCTCT', 'TGAGCTGAAG'], 'TTTGCCGAAGTTAAACCC',
                     max_num_of_mismatches=2)
  assert score == 18 * MATCH
  assert got == [RA(0, '18=', 18 * MATCH), RA(), RA()]


COVERAGE_HAP = 'ATCAAGGGAAAAAGTGCCCAGGGCCAAATATGTTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTGAGCT'


def test_haplotype_has_zero_coverage_outside_interval():
  score, got = _fast(['ATCAAGGGAAAAAGTGCCCA', 'GGGCCAAATATGTTTTG', 'ATATGTTATGGGTTATGCAGGA',
                      'GTTTTGGGTTTTGCAGGTCA', 'AGGACAAAGTATGGTT', 'CAAAGTATGGTTGTGAGCT'],
                     COVERAGE_HAP, max_num_of_mismatches=2, ref_prefix_len=11, ref_suffix_len=11)
  assert score == 350
  assert got == [RA(0, '20=', 80), RA(20, '17=', 68), RA(27, '22=', 68), RA(31, '20=', 70),
                 RA(45, '16=', 64), RA()]


def test_haplotype_has_zero_coverage_inside_interval():
  score, got = _fast(['ATCAAGGGAAAAAGTGCCCA', 'GGGAAACCAAATATGTTTTG', 'ATATGTTATGGGTTATGCAGGA',
                      'GTTTTGGGTTTTGCAGGTCA', 'AGGACAAAGTATGGTT', 'CAAAGTATGGTTGTGAGCT'],
                     COVERAGE_HAP, max_num_of_mismatches=2, ref_prefix_len=11, ref_suffix_len=11)
  assert score == 0
  assert got == [RA(0, '20=', 80), RA(), RA(), RA(), RA(), RA()]


# ---------------------------------------------------------------- haplotypes -> reference
def test_align_haplotypes_to_reference():
  a = _aligner('AGAAGGTCCCTTTGCCGAAGTTAAACCCTTTCGCGC')
  a.stage(F.INIT_LOCAL_ALIGNER)
  a.set_haplotypes(['GTCCCTTTGCCGAAGTTAAACCCTTT', 'GTCCCTTTGCCGAGTTAAACCCTTT', 'GTCCCTATGCCGAAGTTAAACCCTTT'])
  a.stage(F.ALIGN_HAPLOTYPES)
  got = [a.haplotype_alignment(k) for k in range(3)]
  assert got[0] == dict(haplotype_index=0, haplotype_score=-1, ref_pos=5, is_reference=True, cigar='26=')
  assert got[1] == dict(haplotype_index=1, haplotype_score=-1, ref_pos=5, is_reference=False, cigar='12=1D13=')
  assert got[2] == dict(haplotype_index=2, haplotype_score=-1, ref_pos=5, is_reference=False, cigar='6=1X19=')


@pytest.mark.parametrize('cigar,size,want', [
    ('10=1X3=', 24, [0] * 24),
    ('3=4I2=', 9, [0, 0, 0, 0, -1, -2, -3, -4, -4]),
    ('3=4D2=', 5, [0, 0, 0, 4, 4]),
    ('3=4D2=2I2=', 9, [0, 0, 0, 4, 4, 4, 3, 2, 2]),
    ('3=4I2=2D2=', 11, [0, 0, 0, 0, -1, -2, -3, -4, -4, -2, -2]),
])
def test_set_positions_map(cigar, size, want):
  assert F.positions_map(cigar, size) == want


def test_ssw_align_reads_to_haplotypes():
  a = _aligner(kmer_size=3)
  a.stage(F.INIT_LOCAL_ALIGNER)
  a.set_reads(['CAGGGCCAAATGTTT', 'GCCATATATGCACAGGGTTATG', 'TTGGGTTGCAGGACA', 'ACAGGGTTTTTTGCAGGACAA',
               'TGTTGGGTTCAGCAGTTTT'])
  a.set_haplotypes(['AAGTGCCCAGGGCCAAATGTTTTGGGTTTTGCAGGACAAAGTATGGTT',
                    'AAGTGCCCAGGGCCAAATATGCACAGGGTTTTGCAGGACAAAGTATGGTT'])
  a.stage(F.ALIGN_HAPLOTYPES)
  a.stage(F.LOCAL_ALIGN_READS, 40)
  hap1 = [a.read_alignment(0, r) for r in range(5)]
  hap2 = [a.read_alignment(1, r) for r in range(5)]
  assert hap1 == [RA(7, '15=', 60), RA(), RA(21, '5=2D10=', 51), RA(23, '3S3=2I13=', 55), RA()]
  assert hap2 == [RA(7, '11=4S', 44), RA(11, '4=1X14=1X2=', 68), RA(25, '2S3=2D10=', 43),
                  RA(22, '6=2I13=', 67), RA()]


# ---------------------------------------------------------------- CIGAR merging
def _merge(reference, haplotype, read, position, read_cigar):
  a = _aligner(reference)
  a.stage(F.INIT_LOCAL_ALIGNER)
  a.set_haplotypes([haplotype])
  a.stage(F.ALIGN_HAPLOTYPES)
  a.set_reads([read])
  return a.calculate_read_to_ref_alignment(0, position, read_cigar, a.haplotype_alignment(0)['cigar'])


def test_read_to_ref_match_mismatch():
  assert _merge(REF, 'TGTTTAGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'TGTTTAGGGTTTTGCAGGA', 7, '19=') == '19M'


def test_read_to_ref_haplotype_soft_clipped():
  ref = 'nnnnnnnnnnnTGTTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTGAGCTGAAGATATG'
  assert _merge(ref, 'GATCATGTTTAGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTAGGGTTTT', 0, '19=') == '5S13M'


@pytest.mark.parametrize('name,haplotype,read,cigar,want', [
    ('ins_snp_merge', 'CGGATCATGTTTTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGATTTTTGGGTTTTCAG', '7=1X15=',
     '7M2I11M1D3M'),
    ('ins_ins_merge', 'CGGATCATGTTTTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTTTTGGGTTTTCAG', '7=1I16=',
     '7M3I11M1D3M'),
    ('del_del_merge', 'CGGATCATGTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTGGGTTTTCAGGACAAA', '7=1D18=',
     '7M2D9M1D9M'),
    ('del_ins_merge', 'CGGATCATGTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTTGGGTTTTCAGGACAAA', '7=2I19=',
     '7M1I11M1D9M'),
    ('del_ins_merge2', 'CGGATCATGTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTGGGTTTTCAGGACAAA', '7=2I17=',
     '7M1D10M1D9M'),
    ('ins_del_merge', 'CGGATCATGTTTTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTTGGGTTTTCAGGACAAA', '7=1D21=',
     '7M1I11M1D9M'),
    ('2ins_3del_merge', 'CGGATCATGTTTTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTGGGTTTTCAGGACAAA', '7=3D19=',
     '7M1D10M1D9M'),
    ('1ins_1del_back_to_back', 'CGGATCATGTTTTGGGTTTTCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTGGGTTTTCCAGGACAAA',
     '18=1I9=', '28M'),
    ('1ins_1del_consecutive', 'CGGATCATGTTTTGGGTTTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTGGGTTTTGCAGGACAAA',
     '16=2D12=', '28M'),
    ('1del_1ins_consecutive2', 'CGGATCATGTTTTGGGTTTTGCGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTGGGTTGCGCAGGACAAA',
     '16=2D12=', '28M'),
    ('two_dels_different_positions', 'CGGATCATGTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTGGTTTT', '10=1D6=',
     '7M1D3M1D6M'),
    ('merged_dels', 'CGGATCATGTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTGGGTTTT', '7=1D9=', '7M2D9M'),
    ('ins', 'CGGATCATGTTTTAAGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTAAGGGCCTTTT', '16=2I4=', '11M2I3M2I4M'),
    ('merged_ins', 'CGGATCATGTTTTAAGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTTAAAAGGGTTTT', '13=2I7=', '11M4I7M'),
    ('del_ins', 'CGGATCATGTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGTTTGGGAATTTT', '13=2I4=', '7M1D6M2I4M'),
    ('ins_del', 'CGGATCATGAATTTTGGGTTTTGCAGGACAAAGTATGGTTGAAACTG', 'GATCATGAATTTTGGGTTT', '16=1D3=', '7M2I7M1D3M'),
])
def test_calculate_read_to_ref_alignment(name, haplotype, read, cigar, want):
  assert _merge(REF2, haplotype, read, 2, cigar) == want, name


@pytest.mark.parametrize('cigar,op,length,read_len,want', [
    ('', 'M', 3, 10, '3M'),
    ('3M5I', 'M', 2, 10, '3M5I2M'),
    ('3M5I', 'I', 2, 10, '3M7I'),
    ('3M5I', 'I', 20, 10, '3M7I'),
    ('3M5D', 'D', 20, 10, '3M25D'),
    ('3M5D5M', 'I', 20, 8, '3M5D5M'),
])
def test_merge_cigar_op(cigar, op, length, read_len, want):
  assert F.merge_cigar_op(cigar, op, length, read_len) == want


def test_calculate_score_threshold():
  a = _aligner(read_size=10, realignment_similarity_threshold=0.1)
  a.stage(F.SCORE_THRESHOLD)
  assert 0 <= a.score_threshold() <= 10 * MATCH


@pytest.mark.parametrize('cigar,read,want', [
    ('14=', 'ACTCTCTCTCAGCT', True),
    ('4=2D10=', 'ACTCTCTCAGCTGT', False),
    ('1=2D13=', 'ACTCTCTCAGCTGT', True),
    ('4=2I8=', 'ACTCTCTCTCTCAGCTGT', False),
    ('1=2I10=', 'ACTCTCTCTCTCAGCTGT', True),
])
def test_is_alignment_normalized(cigar, read, want):
  a = _aligner('ATGCTGCACTCTCTCTCAGCTGTCACC')
  assert a.is_alignment_normalized(cigar, 7, read) is want


def test_options_are_checked_like_the_reference():
  from deepvariant_amd import _lib
  with pytest.raises(_lib.DvError, match='kmer_size'):
    F.FastPassAligner(kmer_size=33)
  with pytest.raises(_lib.DvError, match='similarity'):
    F.FastPassAligner(realignment_similarity_threshold=1.5)


# ---------------------------------------------------------------- AlignReads end to end
def test_align_reads_end_to_end():
  """AlignReads: reads tiling two haplotypes (the reference and the reference with a 2-base
  deletion) come back in reference coordinates (window start 1000); the read that spans the
  deletion carries it, reads away from it are plain matches, an unrelated read keeps its
  original alignment."""
  dele = REF[:33] + REF[35:]
  a = _aligner(kmer_size=8, read_size=25, ref_prefix_len=11, ref_suffix_len=11)
  a.set_reference(REF, 1000)
  a.set_haplotypes([dele, REF])
  reads = [REF[5:30], REF[25:50], REF[45:75],          # reference haplotype
           dele[5:30], dele[22:48], dele[40:73],       # deletion haplotype (the first equals REF[5:30])
           'ACGTACGTACGTACGTACGTACGTA']
  out = a.align_reads(reads)
  assert [o[0] for o in out] == [1, 1, 1, 1, 1, 1, 0]
  assert out[0][1:] == (1005, [(1, 25)]) and out[3][1:] == (1005, [(1, 25)])
  assert out[1][1:] == (1025, [(1, 25)])
  assert out[2][1:] == (1045, [(1, 30)])
  assert REF[32:36] == 'TTTT'                                  # the gap is placed leftmost in the run
  assert out[4][1:] == (1022, [(1, 10), (3, 2), (1, 16)])     # 10M2D16M
  assert out[5][1:] == (1042, [(1, 33)])                       # behind the deletion: shifted by 2


def test_force_alignment_against_the_haplotype_as_reference():
  """The alt-aligned use (RealignReadsToHaplotype): one haplotype that IS the reference, every
  read either fast-passes, or goes through the local aligner whatever its score; reads the
  local aligner cannot place at all come back dropped (status 2)."""
  a = _aligner(kmer_size=8, read_size=25, force_alignment=True, ref_prefix_len=5, ref_suffix_len=5)
  a.set_reference(REF, 500)
  a.set_haplotypes([REF])
  reads = [REF[10:40],                                   # exact
           REF[10:25] + 'TT' + REF[25:45],               # 2-base insertion
           REF[30:42] + REF[45:70],                      # 3-base deletion
           'N' * 20]                                     # scores nothing anywhere
  out = a.align_reads(reads)
  assert out[0] == (1, 510, [(1, 30)])
  assert out[1][0] == 1 and out[1][1] == 510 and sum(n for op, n in out[1][2] if op == 2) == 2
  assert out[2][0] == 1 and out[2][1] == 530 and [op for op, _ in out[2][2]] == [1, 3, 1]
  assert out[3][0] == 2


# ---------------------------------------------------------------- SIMD batch path == scalar path
def _fields(a):
  return (a.score, a.ref_begin, a.ref_end, a.query_begin, a.query_end, a.mismatches, a.cigar)


@pytest.mark.parametrize('seed,scoring', [(1, (4, 6, 8, 2)), (2, (2, 2, 3, 1)), (3, (1, 4, 6, 1)), (4, (4, 6, 8, 1))])
def test_batched_alignment_equals_one_at_a_time(seed, scoring):
  """The realigner aligns 16 (reference, query) pairs per SIMD batch (int16 lanes, forward and
  reverse pass); every field must equal the scalar aligner's, which the libssw vectors pin.
  Queries: mutated copies of reference pieces (substitutions, indels, N, soft-clipped junk),
  pure junk, one-base and repeat sequences, lengths 1..400 in one batch."""
  import numpy as np
  rng = np.random.default_rng(seed)
  letters = np.array(list('ACGT'))
  reference = ''.join(rng.choice(letters, size=700))
  reference = reference[:300] + 'TGA' * 15 + reference[300:500] + 'N' * 3 + reference[500:]
  queries = []
  for _ in range(75):
    a = int(rng.integers(0, len(reference) - 50))
    piece = list(reference[a:a + int(rng.integers(20, 400))])
    for _ in range(int(rng.integers(0, 6))):
      k = int(rng.integers(0, len(piece)))
      kind = int(rng.integers(0, 4))
      if kind == 0:
        piece[k] = str(rng.choice(letters))
      elif kind == 1:
        piece[k:k] = list(rng.choice(letters, size=int(rng.integers(1, 12))))
      elif kind == 2:
        del piece[k:k + int(rng.integers(1, 12))]
      else:
        piece[k] = 'N'
    clip = ''.join(rng.choice(letters, size=int(rng.integers(0, 10))))
    queries.append(clip + ''.join(piece) + clip[::-1])
  queries += [''.join(rng.choice(letters, size=30)), 'A', 'TGA' * 20, 'N' * 10, reference, reference[100:140].lower()]
  got = F.local_align_many(reference, queries, *scoring)
  assert len(got) == len(queries)
  for q, g in zip(queries, got):
    want = F.local_align(reference, q, *scoring)
    assert g is not None and _fields(g) == _fields(want), q
  # and with a single query (the batch path needs >= 2 pairs; one pair takes the scalar route)
  assert _fields(F.local_align_many(reference, queries[:1], *scoring)[0]) == _fields(
      F.local_align(reference, queries[0], *scoring))


def test_batched_alignment_refuses_what_the_scalar_path_refuses():
  got = F.local_align_many('ACGTACGT', ['ACGT', '', 'TTTT'])
  assert got[1] is None and got[0].score == 8 and got[2].score == 2
