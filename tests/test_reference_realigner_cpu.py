"""The read realigner and the trimmed-read / alt-haplotype helpers against the REFERENCE's own sources
(oracle/_ref/libdvref.so: deepvariant/realigner/fast_pass_aligner.cc + ssw.cc and deepvariant/
alt_aligned_pileup_lib.cc compiled unmodified, oracle/ref_build/):

  * FastPassAligner::AlignReads -- k-mer index, fast pass, haplotype -> reference alignment, local alignment of
    the reads the fast pass could not place, best haplotype per read, CIGAR merging, the normalisation check --
    against the product's native aligner (deepvariant_amd/csrc/fast_pass_aligner.cpp) on seeded random windows:
    assembled-looking haplotypes (SNPs, insertions, deletions against the window), reads drawn from them with
    sequencing errors and indels, both force_alignment settings, reference padding as the window realigner uses it;
  * TrimReads / TrimRead / TrimCigar, CalculateAlignmentRegion, RealignReadsToHaplotype against
    deepvariant_amd/alt_aligned_pileup_lib.py / fast_pass_aligner.realign_reads_to_haplotype on reads with every
    CIGAR operator and methylation arrays.

libssw, which the reference links, is not vendored: in this build its C++ interface runs on the product's
restatement (csrc/local_align.cpp, pinned by the vectors of ssw_test.cc), so what is compared is the code ABOVE
the local aligner, with one local aligner under both sides.  CPU only; skipped without the reference build.
"""
import numpy as np
import pytest

from oracle import oracle as O

if not O.reference_available():
  pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here', allow_module_level=True)

from deepvariant_amd import alt_aligned_pileup_lib as AL      # noqa: E402
from deepvariant_amd import dv_types as T                     # noqa: E402
from deepvariant_amd import fast_pass_aligner as FPA          # noqa: E402
from tests.test_hip_allelecounter import _Ref, _fuzz_reads    # noqa: E402

ALN = dict(match=4, mismatch=6, gap_open=8, gap_extend=2, kmer_size=32, max_num_of_mismatches=2,
           realignment_similarity_threshold=0.16934)      # realigner.py's aln_* flag defaults


def _mutate(rng, seq, n_events):
  s = list(seq)
  for _ in range(n_events):
    p = int(rng.integers(20, len(s) - 20))
    kind = rng.random()
    if kind < 0.4:
      s[p] = 'ACGT'[('ACGT'.index(s[p]) + int(rng.integers(1, 4))) % 4] if s[p] in 'ACGT' else 'A'
    elif kind < 0.7:
      s[p:p] = ['ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 12)))]
    else:
      del s[p:p + int(rng.integers(1, 12))]
  return ''.join(s)


def _window_case(rng, read_len, n_reads):
  n = int(rng.integers(260, 520))
  window = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=n))
  if rng.random() < 0.3:      # a short tandem repeat in the window
    p = int(rng.integers(40, n - 80))
    window = window[:p] + 'CA' * int(rng.integers(6, 15)) + window[p:]
  haplotypes = sorted({window} | {_mutate(rng, window, int(rng.integers(1, 4))) for _ in range(int(rng.integers(1, 5)))})
  reads = []
  for i in range(n_reads):
    hap = haplotypes[int(rng.integers(0, len(haplotypes)))]
    L = min(read_len, len(hap) - 2)
    s = int(rng.integers(0, len(hap) - L + 1))
    seq = list(hap[s:s + L])
    u = rng.random()
    if u < 0.35:                      # sequencing errors
      for _ in range(int(rng.integers(1, 4))):
        seq[int(rng.integers(0, L))] = 'ACGT'[int(rng.integers(0, 4))]
    elif u < 0.5:                     # an indel against every haplotype: only the local aligner places it
      p = int(rng.integers(10, L - 10))
      if rng.random() < 0.5:
        del seq[p:p + int(rng.integers(1, 5))]
      else:
        seq[p:p] = ['ACGT'[int(j)] for j in rng.integers(0, 4, size=int(rng.integers(1, 5)))]
    elif u < 0.55:
      seq = ['ACGT'[int(j)] for j in rng.integers(0, 4, size=L)]      # unrelated read
    seq = ''.join(seq)
    reads.append(T.Read(
        fragment_name='r%d' % i, read_number=int(rng.integers(0, 2)), number_reads=2, aligned_sequence=seq,
        aligned_quality=bytes(rng.integers(10, 41, size=len(seq)).astype(np.uint8)),
        alignment=T.LinearAlignment(position=T.Position('chr', 5000 + s, bool(rng.integers(0, 2))),
                                    mapping_quality=int(rng.integers(0, 61)), cigar=[T.CigarUnit(1, len(seq))])))
  return window, haplotypes, reads


def _same_reads(mine, theirs, originals):
  assert len(mine) == len(theirs) == len(originals)
  changed = 0
  for m, t, o in zip(mine, theirs, originals):
    if t is None:
      assert m is None
      continue
    assert m is not None
    got = (m.alignment.position.position, [(c.operation, c.operation_length) for c in m.alignment.cigar], m.aligned_sequence,
           m.fragment_name, m.read_number, m.alignment.mapping_quality, m.alignment.position.reverse_strand)
    want = (t['position'], t['cigar'], t['seq'], t['name'], t['read_number'], t['mapq'], t['reverse'])
    assert got == want, (o.fragment_name, got[:2], want[:2])
    changed += (t['position'], t['cigar']) != (o.alignment.position.position, [(c.operation, c.operation_length) for c in o.alignment.cigar])
  return changed


@pytest.mark.parametrize('seed,read_len,force,padded', [(1, 100, False, False), (2, 100, False, True), (3, 150, True, False),
                                                       (4, 150, False, True), (5, 60, False, False), (6, 250, True, True)])
def test_fast_pass_aligner_align_reads(seed, read_len, force, padded):
  rng = np.random.default_rng(seed)
  total = changed = empties = 0
  for _ in range(12):
    window, haplotypes, reads = _window_case(rng, read_len, int(rng.integers(20, 70)))
    pre = int(rng.integers(1, 30)) if padded else 0
    suf = int(rng.integers(1, 30)) if padded else 0
    pad_l = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=pre))
    pad_r = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=suf))
    reference = pad_l + window + pad_r
    haps = [pad_l + h + pad_r for h in haplotypes]
    cfg = dict(ALN, read_size=read_len, force_alignment=force, ref_prefix_len=pre, ref_suffix_len=suf)
    theirs = O.reference_align_reads(reference, 'chr', 5000 - pre, haps, reads, **cfg)
    aligner = FPA.FastPassAligner(**cfg)
    aligner.set_reference(reference, 5000 - pre)
    aligner.set_haplotypes(haps)
    mine = aligner.realign_reads(reads)
    changed += _same_reads(mine, theirs, reads)
    total += len(reads)
    empties += sum(t is None for t in theirs)
  assert total > 300 and changed > 30
  assert force or empties == 0      # the empty Read only exists under force_alignment


@pytest.mark.parametrize('seed,long_reads', [(31, False), (32, True)])
def test_trim_reads_and_alignment_region(seed, long_reads):
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  ref = _Ref(seq)
  reads = _fuzz_reads(rng, ref, 300 if not long_reads else 120, 700, 2100, long_reads)
  for r in reads[::3]:
    n = len(r.aligned_sequence)
    r.base_modifications[T.K5MC] = bytes(rng.integers(0, 256, size=n).astype(np.uint8))
    if rng.random() < 0.5:
      r.base_modifications[T.K6MA] = bytes(rng.integers(0, 256, size=n).astype(np.uint8))
  kept_total = 0
  for (lo, hi) in ((1000, 1147), (1500, 1699), (690, 760), (2050, 2400), (1200, 1201)):
    # TrimRead CHECKs that a read starts before the region's end (callers hand over the reads of a region query);
    # both sides refuse the others
    late = [r for r in reads if r.alignment.position.position >= hi][:1]
    if late:
      with pytest.raises(ValueError, match='ref_length > 0'):
        AL.trim_reads(late, lo, hi)
      with pytest.raises(O.OracleError, match='ref_length > 0'):
        O.reference_trim_reads(late, 'c', lo, hi)
    inside = [r for r in reads if r.alignment.position.position < hi]
    for min_overlap in (15, 1, 60):
      theirs = O.reference_trim_reads(inside, 'c', lo, hi, min_overlap)
      mine, original = AL.trim_reads(inside, lo, hi, min_overlap)
      assert len(mine) == len(theirs) and original == [t['original_position'] for t in theirs]
      for m, t in zip(mine, theirs):
        assert (m.fragment_name, m.read_number, m.alignment.position.position,
                [(c.operation, c.operation_length) for c in m.alignment.cigar], m.aligned_sequence,
                bytes(m.aligned_quality)) == (t['name'], t['read_number'], t['position'], t['cigar'], t['seq'], t['qual'])
        assert m.base_modifications.get(T.K5MC) == t['mod_5mc'] and m.base_modifications.get(T.K6MA) == t['mod_6ma']
      kept_total += len(theirs)
  assert kept_total > 80
  for start, n_ref, hw, length in ((5, 1, 73, 6000), (5990, 4, 73, 6000), (3000, 12, 110, 6000), (0, 1, 5, 8), (50, 3, 99, 60)):
    assert AL.calculate_alignment_region(T.Variant('c', start, start + n_ref, 'A' * n_ref, ['C']), hw, length) == \
        O.reference_calculate_alignment_region(length, start, n_ref, hw)


@pytest.mark.parametrize('seed', [41, 42, 43])
def test_realign_reads_to_haplotype(seed):
  """RealignReadsToHaplotype: window-trimmed reads against prefix + alt + suffix (kRefAlignMargin = 0,
  force_alignment, read_size from the first read)."""
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=3000))
  ref = _Ref(seq)
  total = changed = 0
  for _ in range(10):
    start = int(rng.integers(300, 2500))
    ref_start, ref_end = start - 73, start + 74
    window = seq[ref_start:ref_end]
    alt_kind = rng.random()
    if alt_kind < 0.5:
      haplotype = window[:73] + window[73] + ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 9)))) + window[74:]
    else:
      haplotype = window[:74] + window[74 + int(rng.integers(1, 9)):]
    reads = []
    for i in range(int(rng.integers(10, 40))):
      src = haplotype if rng.random() < 0.6 else window
      L = int(rng.integers(40, len(src) - 1))
      s = int(rng.integers(0, len(src) - L + 1))
      body = list(src[s:s + L])
      if rng.random() < 0.3:
        body[int(rng.integers(0, L))] = 'ACGT'[int(rng.integers(0, 4))]
      body = ''.join(body)
      reads.append(T.Read(fragment_name='t%d' % i, read_number=0, number_reads=1, aligned_sequence=body,
                          aligned_quality=bytes([30] * len(body)),
                          alignment=T.LinearAlignment(position=T.Position('c', ref_start + s, False), mapping_quality=60,
                                                      cigar=[T.CigarUnit(1, len(body))])))
    cfg = dict(match=4, mismatch=6, gap_open=8, gap_extend=2, kmer_size=32, max_num_of_mismatches=2,
               realignment_similarity_threshold=0.16934)
    theirs = O.reference_realign_reads_to_haplotype(haplotype, reads, 'c', ref_start, ref_end, ref, len(seq), **cfg)
    mine = FPA.realign_reads_to_haplotype(haplotype, reads, 'c', ref_start, ref_end, ref, cfg)
    changed += _same_reads(mine, theirs, reads)
    total += len(reads)
  assert total > 100 and changed > 10
