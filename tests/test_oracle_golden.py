"""Pins the CPU oracle to the reference's golden TFRecord images (CPU only).

Fixture: tests/golden/illumina_wgs_chr20.npz, made by tests/golden/make_golden.py
from deepvariant/testdata/golden.calling_examples.tfrecord.gz (84 x 100x221x7,
HG001 chr20:10,000,000-10,010,000 = BASELINE.json configs[0]).
"""
import os

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from oracle import oracle as O
from tests import golden_io
from tests.golden.make_golden import wgs_options

FIXTURE = os.path.join(os.path.dirname(__file__), 'golden',
                       'illumina_wgs_chr20.npz')


@pytest.fixture(scope='module')
def golden():
  return golden_io.load(FIXTURE)


def test_golden_illumina_images(golden):
  reads, examples, z = golden
  opts = wgs_options()
  hw = (opts.width - 1) // 2
  band = opts.reference_band_height
  n_rows = n_match = n_full = 0
  for i, ex in enumerate(examples):
    call, img = ex['call'], ex['image']
    assert img.shape == (100, 221, 7)
    got, kept, _ = O.build_pileup(
        opts, call, ex['ref_window'], [reads[k] for k in ex['read_idx']],
        call.variant.start - hw, ex['alt_alleles'], return_row_reads=True)
    # (1) reference band: pure function of the FASTA window -> exact, always.
    np.testing.assert_array_equal(got[:band], img[:band])
    # (2) golden read rows vs raw-BAM encodings.
    ours = {got[r].tobytes() for r in range(band, band + kept)}
    gold_rows = [r for r in range(band, 100) if img[r].any()]
    n_rows += len(gold_rows)
    n_match += sum(img[r].tobytes() in ours for r in gold_rows)
    # (3) images the realigner left untouched are bit-exact in full.
    if z['e_full'][i]:
      np.testing.assert_array_equal(got, img)
      n_full += 1
    # (4) structure: blank rows only at the bottom.
    nz = [bool(img[r].any()) for r in range(100)]
    assert nz == sorted(nz, reverse=True)
  assert len(examples) == 84
  assert n_rows == 4309
  # Acceptance threshold measured by the survey AND by make_golden.py: 80.5 %
  # of rows are untouched by the (not yet restated) realigner.
  assert n_match == 3467
  assert n_full == 7


# ---------------------------------------------------------------------------
# PacBio golden (BASELINE.json configs[3] shape: 100x147x10, realigner OFF).
# Fixture: tests/golden/pacbio_chr20.npz from golden.pacbio_examples.tfrecord.gz +
# test_pacbio.chr20_100kbp_at_9mb.bam + grch38.chr20_and_21_10M.fa.gz
# (tests/golden/make_golden.py pacbio).  Pinned: all 8 encoder channels of the reference
# band, and {read_base, base_quality, mapping_quality, strand, base_differs_from_ref} of
# every read row ('=' / 'X' / I / D CIGARs of real HiFi reads); haplotype / support /
# methylation / the two alt-aligned channels need inputs the testdata does not carry.
# ---------------------------------------------------------------------------
PACBIO_FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'pacbio_chr20.npz')


def pacbio_rows(opts, reads, ex, encode_read):
  from tests.golden.make_golden import PACBIO_CHECKED
  call = ex['call']
  start = call.variant.start - (opts.width - 1) // 2
  ours = set()
  for k in ex['read_idx']:
    row = encode_read(call, ex['ref_window'], reads[k], start)
    if row is not None:
      ours.add(np.ascontiguousarray(row[0][:, PACBIO_CHECKED]).tobytes())
  return ours


def check_pacbio_example(opts, ex, ours, ref_row):
  from tests.golden.make_golden import PACBIO_CHECKED
  img = ex['image']
  assert img.shape == (100, 147, 10)
  band = opts.reference_band_height
  for r in range(band):
    np.testing.assert_array_equal(img[r, :, :8], ref_row[0])
  gold = [r for r in range(band, 100) if img[r][:, PACBIO_CHECKED].any()]
  hits = sum(np.ascontiguousarray(img[r][:, PACBIO_CHECKED]).tobytes() in ours for r in gold)
  return len(gold), hits


def test_golden_pacbio_rows():
  from tests.golden.make_golden import pacbio_options
  reads, examples, z = golden_io.load(PACBIO_FIXTURE)
  # what make_golden.py measured over ALL 401 golden images (= SURVEY 8c's numbers)
  assert z['stats'].tolist() == [401, 401, 13689, 13689]
  opts = pacbio_options()
  n_rows = n_hit = 0
  for ex in examples:
    ours = pacbio_rows(opts, reads, ex,
                       lambda call, win, rd, start: O.encode_read(opts, call, win, rd, start, []))
    a, b = check_pacbio_example(opts, ex, ours, O.encode_reference(opts, ex['ref_window']))
    n_rows += a
    n_hit += b
  assert len(examples) == 134 and n_rows > 4000
  assert n_hit == n_rows


# ---------------------------------------------------------------------------
# PacBio golden, the two alt-aligned channels (8, 9).  Fixture: tests/golden/
# pacbio_alt_chr20.npz (make_golden.py pacbio_alt): every second indel candidate of
# golden.pacbio_examples with ITS window-trimmed reads, the alt haplotypes and the golden
# alt channels.  The product's realigner (libdvhip: FastPassAligner + the libssw
# restatement, over the C ABI) realigns the reads to each haplotype, the oracle draws the
# base_differs_from_ref row of every realigned read, and every golden alt-channel row must
# be one of them (row ORDER needs the phasing tags the testdata lacks).
# ---------------------------------------------------------------------------
PACBIO_ALT_FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'pacbio_alt_chr20.npz')


class _HaplotypeContig:
  """ref_reader stand-in: the realigner only asks for the contig length here (margin 0)."""

  def n_bases(self, contig):
    return 1 << 40

  def get_bases(self, contig, start, end):
    raise AssertionError('kRefAlignMargin is 0: no reference padding is fetched')


def test_golden_pacbio_alt_aligned_channels():
  from deepvariant_amd import fast_pass_aligner as fpa
  from deepvariant_amd import make_examples_native as men
  from tests.golden.make_golden import pacbio_options
  reads, examples, z = golden_io.load(PACBIO_ALT_FIXTURE)
  # measured by make_golden.py over ALL 401 golden images: 131 indel candidates, every one
  # of their 4,406 alt-1 rows and 1,123 alt-2 rows reproduced
  assert z['stats'].tolist() == [401, 131, 4406, 1123, 4406, 1123]
  opts = pacbio_options()
  hw = (opts.width - 1) // 2
  band = opts.reference_band_height
  hap2 = bytes(z['hap2']).decode().split('\n')
  n_rows = n_hit = n_alt2 = 0
  for k, ex in enumerate(examples):
    call, combo = ex['call'], ex['alt_alleles']
    v = call.variant
    trimmed = [reads[i] for i in ex['read_idx']]
    h0, h1 = (int(x) for x in z['hap_range'][k])
    assert h0 == v.start - hw
    img = ex['image']                                     # [100, 147, 2] = golden channels 8, 9
    for a, hap in enumerate([ex['ref_window'], hap2[k]][:len(combo)]):
      realigned = fpa.realign_reads_to_haplotype(hap, trimmed, v.reference_name, h0, h1,
                                                 _HaplotypeContig(), men.DEFAULT_ALN_CONFIG)
      ours = set()
      for r in realigned:
        if r is None:
          continue
        row = O.encode_read(opts, call, hap[:opts.width], r, v.start - hw, [])
        if row is not None:
          ours.add(np.ascontiguousarray(row[0][:, 5]).tobytes())
      gold = [r for r in range(band, img.shape[0]) if img[r, :, a].any()]
      n_rows += len(gold)
      n_alt2 += a
      n_hit += sum(np.ascontiguousarray(img[r, :, a]).tobytes() in ours for r in gold)
    if len(combo) == 1:
      np.testing.assert_array_equal(img[:, :, 1], img[:, :, 0])
  assert len(examples) == 65 and n_alt2 > 5 and n_rows > 2500
  assert n_hit == n_rows


# ---------------------------------------------------------------------------
# Candidates: the golden DeepVariantCalls of the Illumina fixture
# (golden.calling_candidates.tfrecord.gz: 78 calls with allele_support) against allele
# counting + the candidate caller run on the fixture's RAW BAM reads with make_examples'
# defaults (min_mapping_quality 5, min_base_quality 10, vsc_min_count 2 / 2,
# vsc_min_fraction 0.12 / 0.06).  The reference realigns reads before it counts alleles
# (realigner ON for this golden); THIS test skips the realigner on purpose and pins what the
# raw reads alone determine: 72 of 78 calls identical in (position, reference bases,
# alternate bases), 47 of them with identical allele_support read-name lists.  The fixture
# holds every read that overlaps a golden candidate, so the counts AT the golden positions
# are complete.  With the realigner in front (test_golden_illumina_chain_with_realigner at
# the end of this file) all 78 calls and their supporting reads are reproduced.
# ---------------------------------------------------------------------------
def golden_candidate_agreement(examples, counts_at):
  """-> (identical calls, identical calls with identical support) over the golden candidates."""
  from deepvariant_amd import variant_calling as vc
  caller = vc.VariantCaller(vc.VariantCallerOptions(
      min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06))
  gold = {}
  for ex in examples:
    v = ex['call'].variant
    gold[(v.start, v.reference_bases, tuple(v.alternate_bases))] = ex['call']
  same = same_support = 0
  for (start, ref, alts), g in gold.items():
    call = caller.call_variant(counts_at(start))
    if call is None or (call.variant.reference_bases, tuple(call.variant.alternate_bases)) != (ref, alts):
      continue
    same += 1
    a = {k: sorted(s.read_names) for k, s in call.allele_support.items()}
    b = {k: sorted(s.read_names) for k, s in g.allele_support.items()}
    same_support += a == b
  return len(gold), same, same_support


class _WindowRef:
  """Reference bases around the golden candidates, from the fixture's 221-base windows."""

  def __init__(self, examples):
    self.bases = {}
    for ex in examples:
      start = ex['call'].variant.start - 110
      for i, b in enumerate(ex['ref_window']):
        self.bases[start + i] = b

  def n_bases(self, contig):
    return 1 << 40

  def get_bases(self, contig, start, end):
    return ''.join(self.bases.get(p, 'N') for p in range(start, end))


def test_golden_candidates_from_raw_reads():
  from deepvariant_amd import allelecounter as ac
  from oracle import allelecounter_ref as AR
  reads, examples, _ = golden_io.load(FIXTURE)
  ref = _WindowRef(examples)
  lo = min(ex['call'].variant.start for ex in examples)
  hi = max(ex['call'].variant.end for ex in examples)
  counter = AR.AlleleCounter(ref, 'chr20', lo, hi, min_mapping_quality=5, min_base_quality=10)
  for r in reads:
    counter.add(r)

  def counts_at(pos):
    c = counter.counts[pos - lo]
    a = ac.AlleleCount('chr20', c.position, c.ref_base)
    a.ref_supporting_read_count = c.ref_supporting_read_count
    a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
    return a

  n, same, same_support = golden_candidate_agreement(examples, counts_at)
  assert (n, same, same_support) == (78, 72, 47)


# ---------------------------------------------------------------------------
# The whole make_examples chain behind golden.calling_examples / golden.calling_candidates
# (BASELINE.json configs[0]; make_examples_test.py:363-395: realigner ON, partition 1000):
#   reads of the region -> window realigner -> AlleleCounter -> candidate caller -> pileup images.
# Product code: the realigner (window selection logic, de Bruijn assembly, FastPassAligner) and
# the candidate caller.  Oracle code on this GPU-less leg: the per-position allele counts and
# the encoder.  tests/test_hip_realigner.py runs the chain with the device counter and encoder.
# Result: all 78 golden candidates (same alleles, same supporting reads) and all 84 golden
# images bit-exact.
# ---------------------------------------------------------------------------
def run_golden_chain(make_counter, realigner_counter_cls, build_image):
  """-> (found calls by (start, ref, alts), images by example index).  `realigner_counter_cls`: not None =
  the realigner's window selector counts with the oracle's counter (swapped in for the device one while
  the chain runs: no GPU in the CPU suite)."""
  import contextlib
  from tests import realigner_fixture as RF
  with (RF.oracle_allele_counter() if realigner_counter_cls is not None else contextlib.nullcontext()):
    return _run_golden_chain(make_counter, build_image)


def _run_golden_chain(make_counter, build_image):
  from deepvariant_amd import variant_calling as vc
  from deepvariant_amd.realigner import realigner as R
  from deepvariant_amd.realigner import utils as U
  from tests import realigner_fixture as RF
  from tests.golden import make_golden as MG
  ref, sets = RF.load()
  _, examples, _ = golden_io.load(FIXTURE)
  opts = MG.wgs_options()
  rl = R.Realigner(R.realigner_config(), ref)
  caller = vc.VariantCaller(vc.VariantCallerOptions(
      min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06,
      sample_name='NA12878'))
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  found, region_reads = {}, []
  for start in range(9_999_999, 10_010_000, 1000):            # --regions chr20:10,000,000-10,010,000
    region = T.Range('chr20', start, min(start + 1000, 10_010_000))
    in_reads = [r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)]
    _, realigned = rl.realign_reads(in_reads, region)
    in_region = [r for r in realigned if U.ranges_overlap(U.read_range(r), region)]
    for call in caller.calls_from_allele_counts(make_counter(ref, region, in_region)):
      v = call.variant
      found[(v.start, v.reference_bases, tuple(v.alternate_bases))] = call
    region_reads.append((region, realigned))
  hw = (opts.width - 1) // 2
  images = []
  for ex in examples:
    v = ex['call'].variant
    pool = next(rs for region, rs in region_reads if region.start <= v.start < region.end)
    q0, q1 = v.start - opts.read_overlap_buffer_bp, v.end + opts.read_overlap_buffer_bp
    overlapping = [r for r in pool if O.read_overlaps(r, q0, q1)]
    images.append(build_image(opts, ex, overlapping, v.start - hw))
  return found, examples, images


def check_golden_chain(found, examples, images):
  gold = {}
  for ex in examples:
    v = ex['call'].variant
    gold[(v.start, v.reference_bases, tuple(v.alternate_bases))] = ex['call']
  assert set(found) == set(gold) and len(gold) == 78
  from tests import realigner_fixture as RF
  facts = RF.golden_wgs_variants()
  for k, g in gold.items():
    a = {x: sorted(s.read_names) for x, s in found[k].allele_support.items()}
    b = {x: sorted(s.read_names) for x, s in g.allele_support.items()}
    assert a == b, k
    # the Variant inside the example: AD / DP / VAF, no-call genotype, sample name -- what
    # postprocess_variants reads back from CallVariantsOutput.variant
    assert RF.variant_facts(found[k].variant) == facts[(k[0], k[2])], k
  assert len(images) == 84
  for ex, image in zip(examples, images):
    assert np.array_equal(image, ex['image']), ex['call'].variant.start


def test_golden_illumina_chain_with_realigner():
  from deepvariant_amd import allelecounter as ac
  from oracle import allelecounter_ref as AR
  from tests import realigner_fixture as RF

  def make_counter(ref, region, reads):
    counter = AR.AlleleCounter(ref, region.reference_name, region.start, region.end, min_mapping_quality=5,
                               min_base_quality=10)
    for r in reads:
      counter.add(r)
    out = []
    for c in counter.counts:
      a = ac.AlleleCount(region.reference_name, c.position, c.ref_base)
      a.ref_supporting_read_count = c.ref_supporting_read_count
      a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
      out.append(a)
    return out

  def build_image(opts, ex, reads, image_start):
    return O.build_pileup(opts, ex['call'], ex['ref_window'], reads, image_start, ex['alt_alleles'])

  check_golden_chain(*run_golden_chain(make_counter, RF.OracleAlleleCounter, build_image))


# ---------------------------------------------------------------------------
# The long-read chain behind golden.pacbio_examples (BASELINE.json configs[3] shape;
# make_examples_test.py:794-818: realigner off, 25 kb calling regions, track_ref_reads,
# phase_reads with 20 % region padding, sort_by_haplotypes, trim_reads_for_pileup,
# alt_aligned_pileup=diff_channels, vsc_min_fraction_indels 0.12):
#   raw HiFi reads -> allele counts (two passes) -> candidate caller -> DirectPhasing -> HP tags ->
#   trimmed reads, alt haplotypes, realigned reads -> [100, 147, 10] images.
# Product code on this leg: candidate caller, read phasing (native), trimming / haplotypes /
# FastPassAligner, image layout.  Oracle code: allele counts and the encoder.  Result: the 341
# golden variants exactly, and every one of the 401 golden images bit-exact in all 10 channels.
# tests/test_hip_realigner.py runs it with the device counter and encoder.  CPU time: the
# oracle counter walks 4 M bases twice in Python, so this leg draws every 3rd image.
# ---------------------------------------------------------------------------
def test_golden_pacbio_chain_with_phasing():
  from deepvariant_amd import allelecounter as ac
  from deepvariant_amd import alt_aligned_pileup_lib as A
  from deepvariant_amd import direct_phasing
  from deepvariant_amd import fast_pass_aligner as fpa
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import variant_calling as vc
  from deepvariant_amd.realigner import utils as U
  from oracle import allelecounter_ref as AR
  from tests import pacbio_chain as PC
  ref, reads, meta, images = PC.load()
  pic = PC.pic_options(True)
  enc = PC.pic_options(False)
  hw = (pic.width - 1) // 2
  caller = vc.VariantCaller(vc.VariantCallerOptions(2, 2, 0.12, 0.12, sample_name='s', track_ref_reads=True))
  phaser = direct_phasing.DirectPhasing(1)
  n_contig = ref.n_bases('chr20')

  def counts(region, rs, positions=()):
    counter = AR.AlleleCounter(ref, 'chr20', region.start, region.end, min_mapping_quality=1, min_base_quality=10,
                               track_ref_reads=True, candidate_positions=positions)
    for r in rs:
      counter.add(r)
    out = []
    for c in counter.counts:
      a = ac.AlleleCount('chr20', c.position, c.ref_base)
      a.ref_supporting_read_count, a.track_ref_reads = c.ref_supporting_read_count, True
      a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
      out.append(a)
    return out

  spans = [U.read_range(r) for r in reads]
  found, pools = {}, []
  for start in range(PC.REGION.start, PC.REGION.end, PC.PARTITION):
    region = T.Range('chr20', start, min(start + PC.PARTITION, PC.REGION.end))
    padded = U.expand(region, int((region.end - region.start) * 20 / 100), n_contig)
    rs = [golden_io_copy(r) for r, s in zip(reads, spans) if U.ranges_overlap(s, region)]
    positions = caller.call_positions_from_allele_counts(counts(padded, rs))
    candidates = caller.calls_from_allele_counts(counts(padded, rs, positions))
    for r, phase in zip(rs, phaser.phase(candidates, rs)):
      r.info['HP'] = T.ListValue(values=[T.Value(int_value=phase)])
    for c in candidates:
      if region.start <= c.variant.start < region.end:
        v = c.variant
        found[(v.start, v.end, v.reference_bases, tuple(v.alternate_bases))] = (c, len(pools))
    pools.append(rs)
  assert set(found) == {m[:4] for m in meta} and len(found) == 341

  n_alt = 0
  for k in range(0, len(meta), 3):
    start, end, refb, alts, idx = meta[k]
    cand, pool = found[(start, end, refb, alts)]
    v = cand.variant
    combo = [alts[i] for i in idx]
    window = men.get_reference_bases_for_pileup(ref, v, pic.width)
    overlapping = [r for r in pools[pool] if O.read_overlaps(r, v.start - 5, v.end + 5)]
    r0, r1 = A.calculate_alignment_region(v, hw, n_contig)
    drawn, starts = A.trim_reads(overlapping, r0, r1)            # trim_reads_for_pileup: every candidate
    want = O.build_pileup(enc, cand, window, drawn, v.start - hw, combo, pileup_height=100,
                          alignment_positions=starts)
    alt_images = [None, None]
    if A.need_alt_alignment(pic, v):
      for a, alt in enumerate(combo[:2]):
        hap, h0, h1 = A.create_haplotype(ref, v, alt, hw)
        realigned = fpa.realign_reads_to_haplotype(hap, drawn, 'chr20', h0, h1, ref, men.DEFAULT_ALN_CONFIG)
        kept = [(r, s) for r, s in zip(realigned, starts) if r is not None]
        alt_images[a] = O.build_pileup(enc, cand, hap[:pic.width], [r for r, _ in kept], v.start - hw, combo,
                                       pileup_height=100, alignment_positions=[s for _, s in kept])
        n_alt += 1
    full = A.fill_pileup_array(want, alt_images, 'diff_channels', A.get_alt_image_row_indices('diff_channels', combo))
    if full.shape[2] < 10:
      full = np.concatenate([full, np.zeros(full.shape[:2] + (10 - full.shape[2],), np.uint8)], axis=2)
    assert np.array_equal(full, images[k]), (k, start, combo)
  assert n_alt > 40


def golden_io_copy(read):
  import dataclasses
  return dataclasses.replace(read, info=dict(read.info))


# ---------------------------------------------------------------------------
# The Illumina goldens with alt-aligned pileups (make_examples_test.py:736-792): the same region,
# reads and realigner as golden.calling_examples, six default channels, alt_aligned_pileup = rows
# ([300, 221, 6]) or diff_channels ([100, 221, 8]), indel candidates realigned to their alt
# haplotypes.  The golden files hold the 49 examples the training mode could label; every one of
# them must come out bit-exact (the labels themselves are the truth-VCF labeler's, out of scope).
# ---------------------------------------------------------------------------
ALT_FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'illumina_alt_aligned_chr20.npz')


def load_alt_goldens(mode):
  with np.load(ALT_FIXTURE) as f:
    images = f[mode + '_images']
    lines = bytes(f[mode + '_meta']).decode().split('\n')
  meta = []
  for line in lines:
    start, end, ref, alts, idx = line.split('\t')
    meta.append((int(start), int(end), ref, tuple(alts.split(',')), tuple(int(i) for i in idx.split(','))))
  return meta, images


def alt_pic_options(mode, with_alt):
  rr = T.ReadRequirements(min_mapping_quality=5, min_base_quality=10, min_base_quality_mode=1)
  o = T.default_options(rr)
  extra = ['diff_channels_alternate_allele_1', 'diff_channels_alternate_allele_2'] if mode == 'diff_channels' else []
  o.channels = list(T.PILEUP_DEFAULT_CHANNELS) + (extra if with_alt else [])
  o.num_channels = len(o.channels)
  if with_alt:
    o.alt_aligned_pileup = mode
    o.types_to_alt_align = 'indels'
  return o


@pytest.mark.parametrize('mode', ['rows', 'diff_channels'])
def test_golden_illumina_alt_aligned_chain(mode):
  from tests import realigner_fixture as RF
  with RF.oracle_allele_counter():
    _golden_illumina_alt_aligned_chain(mode)


def _golden_illumina_alt_aligned_chain(mode):
  from deepvariant_amd import allelecounter as ac
  from deepvariant_amd import alt_aligned_pileup_lib as A
  from deepvariant_amd import fast_pass_aligner as fpa
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import variant_calling as vc
  from deepvariant_amd.realigner import realigner as R
  from deepvariant_amd.realigner import utils as U
  from oracle import allelecounter_ref as AR
  from tests import realigner_fixture as RF
  ref, sets = RF.load()
  meta, images = load_alt_goldens(mode)
  pic, enc = alt_pic_options(mode, True), alt_pic_options(mode, False)
  hw = (pic.width - 1) // 2
  n_contig = ref.n_bases('chr20')
  rl = R.Realigner(R.realigner_config(), ref)
  caller = vc.VariantCaller(vc.VariantCallerOptions(2, 2, 0.12, 0.06))
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  wanted_regions = sorted({(m[0] - 9_999_999) // 1000 for m in meta})
  found = {}
  for k in wanted_regions:
    region = T.Range('chr20', 9_999_999 + 1000 * k, min(9_999_999 + 1000 * (k + 1), 10_010_000))
    _, realigned = rl.realign_reads([r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)], region)
    counter = AR.AlleleCounter(ref, 'chr20', region.start, region.end, min_mapping_quality=5, min_base_quality=10)
    for r in realigned:
      if U.ranges_overlap(U.read_range(r), region):
        counter.add(r)
    for c in counter.counts:
      if not c.read_alleles:
        continue
      a = ac.AlleleCount('chr20', c.position, c.ref_base)
      a.ref_supporting_read_count = c.ref_supporting_read_count
      a.read_alleles = {n: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for n, v in c.read_alleles.items()}
      call = caller.call_variant(a)
      if call is not None:
        v = call.variant
        found[(v.start, v.end, v.reference_bases, tuple(v.alternate_bases))] = (call, realigned)
  n_alt_images = 0
  for k, (start, end, refb, alts, idx) in enumerate(meta):
    cand, pool = found[(start, end, refb, alts)]
    v = cand.variant
    combo = [alts[i] for i in idx]
    window = men.get_reference_bases_for_pileup(ref, v, pic.width)
    overlapping = [r for r in pool if O.read_overlaps(r, v.start - 5, v.end + 5)]
    needs_alt = A.need_alt_alignment(pic, v)
    if needs_alt:
      r0, r1 = A.calculate_alignment_region(v, hw, n_contig)
      drawn, starts = A.trim_reads(overlapping, r0, r1)
    else:
      drawn, starts = overlapping, None
    want = O.build_pileup(enc, cand, window, drawn, v.start - hw, combo, pileup_height=100, alignment_positions=starts)
    alt_images = [None, None]
    if needs_alt:
      for a, alt in enumerate(combo[:2]):
        hap, h0, h1 = A.create_haplotype(ref, v, alt, hw)
        realigned = fpa.realign_reads_to_haplotype(hap, drawn, 'chr20', h0, h1, ref, men.DEFAULT_ALN_CONFIG)
        kept = [(r, s) for r, s in zip(realigned, starts) if r is not None]
        alt_images[a] = O.build_pileup(enc, cand, hap[:pic.width], [r for r, _ in kept], v.start - hw, combo,
                                       pileup_height=100, alignment_positions=[s for _, s in kept])
        n_alt_images += 1
    full = A.fill_pileup_array(want, alt_images, mode, A.get_alt_image_row_indices(mode, combo))
    if full.shape[2] < images.shape[3]:
      full = np.concatenate([full, np.zeros(full.shape[:2] + (images.shape[3] - full.shape[2],), np.uint8)], axis=2)
    assert full.shape == images[k].shape and np.array_equal(full, images[k]), (mode, k, start, combo)
  assert len(meta) == 49 and n_alt_images >= 4      # the labelled set has few indels


def test_golden_candidate_positions():
  """golden.candidate_positions (make_examples_test.py, --mode candidate_sweep): per 1 kb partition the
  positions where the RAW reads (no realigner) make the caller emit a candidate, -2 after each
  partition, -1 at the end of the region.  Oracle counts + the product's CallVariantPosition."""
  from deepvariant_amd import allelecounter as ac
  from deepvariant_amd import variant_calling as vc
  from deepvariant_amd.realigner import utils as U
  from oracle import allelecounter_ref as AR
  from tests import realigner_fixture as RF
  ref, sets = RF.load()
  with np.load(RF.FIXTURE) as f:
    golden = f['wgs_candidate_positions'].tolist()
  caller = vc.VariantCaller(vc.VariantCallerOptions(2, 2, 0.12, 0.06))
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  got = []
  for start in range(9_999_999, 10_010_000, 1000):
    region = T.Range('chr20', start, min(start + 1000, 10_010_000))
    counter = AR.AlleleCounter(ref, 'chr20', region.start, region.end, min_mapping_quality=5, min_base_quality=10)
    for r, s in zip(reads, spans):
      if U.ranges_overlap(s, region):
        counter.add(r)
    counts = []
    for c in counter.counts:
      a = ac.AlleleCount('chr20', c.position, c.ref_base)
      a.ref_supporting_read_count = c.ref_supporting_read_count
      a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
      counts.append(a)
    got += caller.call_positions_from_allele_counts(counts) + [-2]
  got.append(-1)
  assert got == golden and len(golden) == 94 and sum(p >= 0 for p in golden) == 82
