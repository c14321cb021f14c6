"""Candidate generation against the REFERENCE's own code (oracle/_ref/libdvref.so, oracle/ref_build/:
deepvariant/allelecounter.cc and deepvariant/variant_calling_multisample.cc -- the caller make_examples runs --
compiled unmodified):

  * allele counts: oracle/allelecounter_ref.py (the checker of the device kernel, tests/test_hip_allelecounter.py)
    == the reference's AlleleCounter at every position -- reference base, reference-supporting read count and
    every read allele (key, bases, type, low-quality flag) -- on the raw NA12878 reads of the ten golden calling
    regions and on seeded fuzz reads (every CIGAR operator, N bases, qualities around the thresholds, duplicate
    read keys, reads hanging off the interval, long reads; both quality modes; `full_range`; track_ref_reads with
    candidate positions);
  * candidates: the product's caller (deepvariant_amd/variant_calling.py, a single-sample restatement) on those
    counts == the reference's MULTI-sample caller with one sample: positions, reference / alternate bases,
    allele_support read lists, AD / DP / VAF, the no-call genotype and the sample name; the first pass of the
    two-pass scheme (CallPositionsFromAlleleCounts); ref_support(_ext) under track_ref_reads.

CPU only; skipped where neither /root/reference nor a prebuilt oracle/_ref/libdvref.so exists.
"""
import numpy as np
import pytest

from oracle import oracle as O

if not O.reference_available():
  pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here', allow_module_level=True)

from deepvariant_amd import allelecounter as ac             # noqa: E402
from deepvariant_amd import dv_types as T                   # noqa: E402
from deepvariant_amd import variant_calling as vc           # noqa: E402
from oracle import allelecounter_ref as AR                  # noqa: E402
from tests.test_hip_allelecounter import _Ref, _fuzz_reads  # noqa: E402   (pure-Python generators)


def _oracle_counts(ref, contig, start, end, reads, candidate_positions=(), track_ref_reads=False, **kw):
  counter = AR.AlleleCounter(ref, contig, start, end, candidate_positions=candidate_positions,
                             track_ref_reads=track_ref_reads, **kw) if (candidate_positions or track_ref_reads) else \
      AR.AlleleCounter(ref, contig, start, end, **kw)
  for r in reads:
    counter.add(r)
  return counter


def _check_counts(counter, start, theirs):
  n_alleles = 0
  for i, c in enumerate(counter.counts):
    mine = (c.ref_base, c.ref_supporting_read_count, {k: (v.bases, v.type, bool(v.is_low_quality)) for k, v in c.read_alleles.items()})
    want = theirs.get(start + i, (c.ref_base, 0, {}))
    assert mine == want, (start + i, mine, want)
    n_alleles += len(mine[2])
  assert set(theirs) <= set(range(start, start + len(counter.counts)))
  return n_alleles


def _product_counts(counter, contig, track_ref_reads=False):
  out = []
  for c in counter.counts:
    a = ac.AlleleCount(contig, c.position, c.ref_base)
    a.ref_supporting_read_count = c.ref_supporting_read_count
    a.track_ref_reads = track_ref_reads
    a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
    out.append(a)
  return out


def _check_calls(mine, theirs, sample, track_ref_reads=False):
  assert [(c.variant.start, c.variant.end, c.variant.reference_bases, list(c.variant.alternate_bases)) for c in mine] == \
         [(c['start'], c['end'], c['reference_bases'], c['alternate_bases']) for c in theirs]
  for a, b in zip(mine, theirs):
    where = (a.variant.start, a.variant.reference_bases)
    assert {k: sorted(s.read_names) for k, s in a.allele_support.items()} == {k: sorted(v) for k, v in b['allele_support'].items()}, where
    call = a.variant.calls[0]
    assert call.call_set_name == b['call_set_name'] == sample and list(call.genotype) == b['genotype'] == [-1, -1]
    assert [v.int_value for v in call.info['AD'].values] == b['info']['AD'], where
    assert [v.int_value for v in call.info['DP'].values] == b['info']['DP'], where
    assert [float(v.number_value) for v in call.info['VAF'].values] == [float(x) for x in b['info']['VAF']], where
    if track_ref_reads:
      assert sorted(a.ref_support) == sorted(b['ref_support']), where
      assert sorted((r.read_name, bool(r.is_low_quality)) for r in a.ref_support_ext) == sorted(b['ref_support_ext']), where
      assert {k: sorted((r.read_name, bool(r.is_low_quality)) for r in v) for k, v in a.allele_support_ext.items()} == \
             {k: sorted(v) for k, v in b['allele_support_ext'].items()}, where


def test_na12878_golden_regions():
  """The raw reads of chr20:10,000,000-10,010,000 (BASELINE configs[0]), per 1000-base calling region as
  make_examples walks them, with make_examples' thresholds."""
  from deepvariant_amd.realigner import utils as U
  from tests import realigner_fixture as RF
  ref, sets = RF.load()
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  caller_kw = dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06)
  caller = vc.VariantCaller(vc.VariantCallerOptions(sample_name='NA12878', **caller_kw))
  n_calls = 0
  for start in range(9_999_999, 10_010_000, 1000):
    region = T.Range('chr20', start, min(start + 1000, 10_010_000))
    in_reads = [r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)]
    counts, calls, _ = O.reference_count_and_call(ref, 'chr20', region.start, region.end, in_reads, 'NA12878',
                                                  min_mapping_quality=5, min_base_quality=10, caller=caller_kw)
    counter = _oracle_counts(ref, 'chr20', region.start, region.end, in_reads, min_mapping_quality=5, min_base_quality=10)
    _check_counts(counter, region.start, counts)
    _check_calls(caller.calls_from_allele_counts(_product_counts(counter, 'chr20')), calls, 'NA12878')
    n_calls += len(calls)
  assert n_calls > 60


@pytest.mark.parametrize('seed,long_reads,legacy', [(1, False, False), (2, False, True), (3, True, False), (4, True, True)])
def test_fuzz_reads(seed, long_reads, legacy):
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  seq = seq[:700] + 'NN' + seq[702:2500] + 'N' + seq[2501:]
  ref = _Ref(seq)
  total_alleles = total_calls = 0
  for (start, end), (lo, hi) in (((1000, 2000), (700, 2100)), ((0, 400), (0, 420)), ((5600, 6000), (5300, 5990))):
    reads = _fuzz_reads(rng, ref, 700 if not long_reads else 200, lo, hi, long_reads)
    kw = dict(min_mapping_quality=10, min_base_quality=20, keep_legacy_behavior=legacy)
    # fuzz reads support many alleles per position thinly: thresholds low enough that multi-allelic sites, deletions
    # of different lengths at one position and insertions next to them all become candidates
    for caller_kw in (dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.02, min_fraction_indels=0.02),
                      dict(min_count_snps=3, min_count_indels=2, min_fraction_snps=0.1, min_fraction_indels=0.04)):
      counts, calls, positions = O.reference_count_and_call(ref, 'c', start, end, reads, 'fuzz', contig_length=len(seq),
                                                            caller=caller_kw, **kw)
      counter = _oracle_counts(ref, 'c', start, end, reads, **kw)
      total_alleles += _check_counts(counter, start, counts)
      caller = vc.VariantCaller(vc.VariantCallerOptions(sample_name='fuzz', **caller_kw))
      mine = caller.calls_from_allele_counts(_product_counts(counter, 'c'))
      _check_calls(mine, calls, 'fuzz')
      total_calls += len(calls)
      _, _, positions = O.reference_count_and_call(ref, 'c', start, end, reads, 'fuzz', contig_length=len(seq),
                                                   caller=caller_kw, call_positions_only=True, **kw)
      assert caller.call_positions_from_allele_counts(_product_counts(counter, 'c')) == positions
  assert total_alleles > 1000 and total_calls > 20


@pytest.mark.parametrize('seed,long_reads', [(11, False), (12, True)])
def test_track_ref_reads_two_pass(seed, long_reads):
  """make_examples_core.py:2880-2932: pass 1 finds the candidate positions, pass 2 counts again keeping the
  reference-supporting reads at those positions by name; the calls then carry ref_support(_ext)."""
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=4000))
  ref = _Ref(seq[:1500] + 'N' + seq[1501:])
  start, end = 1000, 2000
  reads = _fuzz_reads(rng, ref, 600 if not long_reads else 150, 700, 2100, long_reads)
  kw = dict(min_mapping_quality=10, min_base_quality=20)
  caller_kw = dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.03, min_fraction_indels=0.03)
  _, _, positions = O.reference_count_and_call(ref, 'c', start, end, reads, 's', contig_length=len(seq), caller=caller_kw,
                                               call_positions_only=True, **kw)
  assert len(positions) >= 3
  counts, calls, _ = O.reference_count_and_call(ref, 'c', start, end, reads, 's', contig_length=len(seq), caller=caller_kw,
                                                candidate_positions=positions, track_ref_reads=True, **kw)
  counter = _oracle_counts(ref, 'c', start, end, reads, candidate_positions=positions, track_ref_reads=True, **kw)
  _check_counts(counter, start, counts)
  caller = vc.VariantCaller(vc.VariantCallerOptions(sample_name='s', track_ref_reads=True, **caller_kw))
  _check_calls(caller.calls_from_allele_counts(_product_counts(counter, 'c', True)), calls, 's', track_ref_reads=True)
  assert any(c['ref_support'] for c in calls)


@pytest.mark.parametrize('seed,long_reads,legacy', [(21, False, False), (22, False, True), (23, True, False)])
def test_window_selector_models(seed, long_reads, legacy):
  """deepvariant/realigner/window_selector.cc (compiled unmodified) on the reference's AlleleCounter, against
  the product's window selector (deepvariant_amd/realigner/window_selector.py) on the oracle's counts: the
  per-position read-support profile of the VARIANT_READS model -- with and without min_allele_support and the
  strict insertion filter -- exactly, and the ALLELE_COUNT_LINEAR scores to float32 rounding."""
  from deepvariant_amd.realigner import window_selector as WS
  from tests import realigner_fixture as RF
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=5000))
  ref = _Ref(seq[:1800] + 'N' + seq[1801:])
  start, end = 1000, 2200
  reads = _fuzz_reads(rng, ref, 700 if not long_reads else 180, 700, 2300, long_reads)
  linear = (-0.68, 0.081, 0.073, 0.41, 0.29, -0.012, 3.0)
  for support, strict in ((0, False), (2, False), (2, True), (3, True)):
    counts, scores = O.reference_window_candidates(ref, 'c', start, end, reads, min_mapq=20, min_base_quality=20,
                                                   keep_legacy_behavior=legacy, min_allele_support=support,
                                                   enable_strict_insertion_filter=strict, linear_model=linear,
                                                   contig_length=len(seq))
    counter = RF.OracleAlleleCounter(ref, 'c', start, end, min_mapping_quality=20, min_base_quality=20,
                                     keep_legacy_behavior=legacy)
    for r in reads:
      counter.add(r)
    config = WS.WindowSelectorOptions(min_allele_support=support, enable_strict_insertion_filter=strict)
    mine = WS.variant_reads_candidates_from_allele_counter(counter, config)
    assert mine == counts.tolist() and (support > 0 or max(mine) > 3)
    model = WS.AlleleCountLinearModel(*linear)
    got = WS.allele_count_linear_candidates_from_allele_counter(counter, model)
    # float32 sums: the reference adds a position's read alleles in the iteration order of a protobuf Map -- hash
    # order there, key order in this build, arrival order in the product -- so the last bit is not defined by the
    # reference itself; everything above it is
    assert got.dtype == np.float32 and np.abs(got - scores).max() <= 2e-6, np.abs(got - scores).max()
    assert ((got > linear[6]) == (scores > linear[6])).mean() > 0.999


@pytest.mark.parametrize('seed', [51, 52, 53])
def test_normalize_cigar(seed):
  """AlleleCounter::NormalizeCigar (--normalize_reads): indels left-aligned against the reference, adjacent
  operations merged, a leading indel folded into the start -- deepvariant_amd.allelecounter.normalize_cigar against
  the reference's NormalizeAndAdd on reads whose indels sit in homopolymers and tandem repeats."""
  rng = np.random.default_rng(seed)
  seq = []
  while len(seq) < 4000:      # a reference rich in repeats: where left-alignment has something to do
    u = rng.random()
    if u < 0.35:
      seq += ['ACGT'[int(rng.integers(0, 4))]] * int(rng.integers(2, 9))
    elif u < 0.55:
      unit = ['ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(2, 4)))]
      seq += unit * int(rng.integers(2, 7))
    else:
      seq += ['ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 12)))]
  ref = _Ref(''.join(seq[:4000]))
  start, end = 500, 3500
  reads = []
  for i in range(400):
    pos = int(rng.integers(start + 5, end - 400))
    cigar, bases, p = [], [], pos
    for k in range(int(rng.integers(1, 6))):
      n = int(rng.integers(5, 60))
      cigar.append(T.CigarUnit(1, n))
      bases.append(ref.seq[p:p + n])
      p += n
      u = rng.random()
      m = int(rng.integers(1, 7))
      if u < 0.4:      # an insertion that repeats the bases in front of it: shiftable
        cigar.append(T.CigarUnit(2, m))
        bases.append(ref.seq[p - m:p] if rng.random() < 0.7 else ''.join('ACGT'[int(j)] for j in rng.integers(0, 4, size=m)))
      elif u < 0.8:
        cigar.append(T.CigarUnit(3, m))
        p += m
      elif u < 0.9:
        cigar.append(T.CigarUnit(1, 3))      # adjacent matches: merged
        bases.append(ref.seq[p:p + 3])
        p += 3
    if cigar[-1].operation != 1:
      cigar.append(T.CigarUnit(1, 12))
      bases.append(ref.seq[p:p + 12])
    # (no soft clips here: the reference CHECK-fails -- allelecounter.cc:664 -- when an indel shifts left until it
    # meets one, e.g. 4S5M2I in a repeat; the reference's NormalizeCigar vectors cover clipped reads)
    if rng.random() < 0.1:      # a read that starts with an indel
      cigar.insert(0, T.CigarUnit(int(rng.choice([2, 3])), 2))
      if cigar[0].operation == 2:
        bases.insert(0, 'GG')
    s = ''.join(bases)
    reads.append(T.Read(fragment_name='n%d' % i, read_number=0, number_reads=1, aligned_sequence=s,
                        aligned_quality=bytes([30] * len(s)),
                        alignment=T.LinearAlignment(position=T.Position('c', pos, False), mapping_quality=60, cigar=cigar)))
  theirs = O.reference_normalize_cigars(ref, 'c', start, end, reads, contig_length=len(ref.seq))
  # the counter's reads interval is the counting interval itself here (no full_range): its reference bases
  window = ref.seq[start:end]
  changed = 0
  for r, (modified, shift, cigar) in zip(reads, theirs):
    got_modified, got_cigar, got_shift = ac.normalize_cigar(r.aligned_sequence, r.alignment.position.position - start,
                                                            r.alignment.cigar, window)
    assert got_modified == modified and got_shift == shift, (r.fragment_name, modified, shift, got_modified, got_shift)
    if modified:
      assert [(c.operation, c.operation_length) for c in got_cigar] == cigar, r.fragment_name
      changed += 1
  assert changed > 60
