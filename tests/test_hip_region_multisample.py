"""Region-scale parity through the reference-shaped interfaces (GPU): many candidates
sharing one region's reads, two samples stacked per example (DeepTrio-style), one of
them deeper than the image (DownsampleReadIndices' shuffle-and-truncate on every item).

The HIP side is the product end to end -- ExamplesGenerator.encode_region: packing.py
(read table, per-candidate query, support codes, name ranks) + ONE dv_encode_batch launch.
The oracle side gets the PROTO-SHAPED inputs per (candidate, alt combination, sample):
the test restates the reference's own per-candidate steps -- InMemoryReader::Query
(make_examples_native.cc:802-810, nucleus/util/utils.cc:172-188), the window
(`:645-648`), AltAlleleCombinations (`:191-267`) -- and never touches the product's
packed batch, so packing errors cannot cancel out.
"""
import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from tests import fuzz_inputs as F

pytestmark = pytest.mark.gpu


class _Ref:
  def __init__(self, seq):
    self.seq = seq

  def n_bases(self, contig):
    return len(self.seq)

  def get_bases(self, contig, start, end):
    return self.seq[start:end]


def _read_end(read):
  """ReadEnd: start + reference-consuming CIGAR ops (M, D, N, =, X)."""
  return read.alignment.position.position + sum(
      c.operation_length for c in read.alignment.cigar if c.operation in (1, 3, 4, 8, 9))


def _query(reads, start, end):
  """utils.cc:172-188 ReadOverlapsRegion, in input order (InMemoryReader::Query)."""
  return [r for r in reads if end > r.alignment.position.position and start < _read_end(r)]


def _region_reads(rng, n, lo, hi, tag, max_ops=5):
  reads = []
  for i in range(n):
    cigar = F.random_cigar(rng, 1, max_ops)
    if F.query_len(cigar) == 0:
      cigar.append(T.CigarUnit(1, 1))
    qlen = F.query_len(cigar)
    quals = rng.integers(0, 60, size=qlen).astype(np.uint8)
    reads.append(T.Read(
        fragment_name='%s%d' % (tag, int(rng.integers(0, max(n // 2, 1)))),
        read_number=int(rng.integers(0, 2)), number_reads=2,
        fragment_length=int(rng.integers(-900, 900)),
        aligned_sequence=''.join('ACGT'[int(j)] for j in rng.integers(0, 4, size=qlen)),
        aligned_quality=bytes(quals),
        alignment=T.LinearAlignment(
            position=T.Position('chr1', int(rng.integers(lo, hi)), bool(rng.integers(0, 2))),
            mapping_quality=int(rng.integers(0, 70)), cigar=cigar)))
  # the reference's readers hand reads over sorted by position
  reads.sort(key=lambda r: r.alignment.position.position)
  return reads


@pytest.mark.parametrize('sort_by_support', [False, True])
def test_two_samples_many_candidates_one_launch(sort_by_support):
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from oracle import oracle as O
  rng = np.random.default_rng(20260921)
  width = 81
  hw = (width - 1) // 2
  pic = F.options(T.PILEUP_CHANNELS_WITH_INSERT_SIZE, width, 0,
                  sort_by_alt_allele_support=sort_by_support)
  pic.num_channels = len(pic.channels)
  heights = (40, 60)
  pic.height = sum(heights)
  options = T.MakeExamplesOptions(
      pic_options=pic,
      sample_options=[T.SampleOptions(role='child', name='c', pileup_height=heights[0]),
                      T.SampleOptions(role='parent', name='p', pileup_height=heights[1])])
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=4000)))
  # sample 0: ~25x, sample 1: ~110 reads over every site (more than its 55 read rows)
  reads = [_region_reads(rng, 420, 900, 2100, 'c'), _region_reads(rng, 2600, 900, 2100, 'p')]
  cands = []
  for pos in sorted(set(rng.integers(1000, 2000, size=34).tolist())):
    refb = ref.seq[pos]
    alts = [b for b in 'ACGT' if b != refb][:int(rng.integers(1, 3))]
    near = [r for s in reads for r in _query(s, pos - 5, pos + 6)]
    support = {}
    for a in alts:
      k = int(rng.integers(0, 12))
      pick = rng.integers(0, len(near), size=k)
      support[a] = T.SupportingReads(['%s/%d' % (near[int(j)].fragment_name,
                                                 near[int(j)].read_number) for j in pick])
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + 1, refb, alts),
                                   allele_support=support))
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  stats = {}
  examples, shape = gen.encode_region(cands, reads, [0, 1], [0.0, 0.0], stats)
  assert shape == [100, width, len(pic.channels)]
  # the reference's example order: candidates in order, alt combinations in order
  k = 0
  deep_items = 0
  for cand in cands:
    v = cand.variant
    window = men.get_reference_bases_for_pileup(ref, v, width)
    for combo in men.alt_allele_combinations(cand, pic.multi_allelic_mode):
      rec = pw.decode_example(examples[k])
      k += 1
      img = np.frombuffer(rec['image/encoded'][0], np.uint8).reshape(100, width, len(pic.channels))
      parts = []
      for s, h in enumerate(heights):
        overlapping = _query(reads[s], v.start - pic.read_overlap_buffer_bp,
                             v.end + pic.read_overlap_buffer_bp)
        deep_items += len(overlapping) > h - pic.reference_band_height
        parts.append(O.build_pileup(pic, cand, window, overlapping, v.start - hw, list(combo),
                                    pileup_height=h))
      np.testing.assert_array_equal(img, np.concatenate(parts, axis=0),
                                    err_msg='candidate at %d, alts %s' % (v.start, combo))
  assert k == len(examples) == stats['n_examples'] and k >= len(cands)
  assert deep_items >= len(cands)      # the shuffle path ran for (at least) every parent image


def test_trim_reads_for_pileup_long_reads():
  """trim_reads_for_pileup (the long-read configs: deepvariant/json/deepvariant.pacbio.*
  `trim_reads_for_pileup: true`): every candidate draws window-trimmed copies of its reads
  (TrimReads, deepvariant/alt_aligned_pileup_lib.cc:231-248), reads overlapping the window
  by less than 15 bp vanish, rows stay sorted by the UNTRIMMED starts
  (make_examples_native.cc:672-697).  The oracle is fed the trimmed proto-shaped reads per
  candidate; the trimming itself is pinned by the reference's vectors in
  tests/test_alt_aligned_pileup_lib_cpu.py."""
  from deepvariant_amd import alt_aligned_pileup_lib as A
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from oracle import oracle as O
  rng = np.random.default_rng(77)
  width = 61
  hw = (width - 1) // 2
  pic = F.options(T.PILEUP_DEFAULT_CHANNELS + ['haplotype'], width, 50, sort_by_haplotypes=True,
                  min_mapq=1)
  options = T.MakeExamplesOptions(
      pic_options=pic, trim_reads_for_pileup=True,
      sample_options=[T.SampleOptions(role='main', name='m', pileup_height=50)])
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000)))
  reads = _region_reads(rng, 260, 500, 2600, 'r', max_ops=40)       # up to ~1 kb alignments
  for r in reads:
    if rng.random() < 0.6:
      r.info['HP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(0, 3)))])
  cands = []
  for pos in sorted(set(rng.integers(1200, 2400, size=22).tolist())):
    refb = ref.seq[pos]
    alts = [b for b in 'ACGT' if b != refb][:1]
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + 1, refb, alts),
                                   allele_support={}))
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  stats = {}
  examples, shape = gen.encode_region(cands, [reads], [0], [0.0], stats, role='main')
  assert len(examples) == len(cands)
  dropped = moved = 0
  for cand, ex in zip(cands, examples):
    v = cand.variant
    img = np.frombuffer(pw.decode_example(ex)['image/encoded'][0], np.uint8).reshape(shape)
    overlapping = _query(reads, v.start - 5, v.end + 5)
    r0, r1 = A.calculate_alignment_region(v, hw, len(ref.seq))
    kept, original = A.trim_reads(overlapping, r0, r1)
    dropped += len(overlapping) - len(kept)
    moved += sum(1 for k, o in zip(kept, original) if k.alignment.position.position != o)
    want = O.build_pileup(pic, cand, men.get_reference_bases_for_pileup(ref, v, width), kept,
                          v.start - hw, list(v.alternate_bases), pileup_height=50,
                          alignment_positions=original)
    np.testing.assert_array_equal(img, want, err_msg='candidate at %d' % v.start)
  assert moved > 50          # the trimming did cut reads (their starts moved to the window)


def _haplotype_reads(rng, ref_seq, variants, n, lo, hi):
  """Reads sampled from the reference or from one of the alt haplotypes of `variants`
  [(pos, ref_bases, alt_bases)], with the matching CIGAR and ~1 % substitutions."""
  reads = []
  for i in range(n):
    start = int(rng.integers(lo, hi))
    length = int(rng.integers(150, 420))
    end = start + length
    carried = [v for v in variants if start + 12 <= v[0] and v[0] + len(v[1]) + 12 <= end and rng.random() < 0.55]
    seq, cigar, pos = [], [], start
    def add(op, ln):
      if ln <= 0:
        return
      if cigar and cigar[-1].operation == op:
        cigar[-1] = T.CigarUnit(op, cigar[-1].operation_length + ln)
      else:
        cigar.append(T.CigarUnit(op, ln))
    for vpos, refb, altb in sorted(carried):
      if vpos < pos:
        continue                                      # overlapping variants: keep the first
      seq.append(ref_seq[pos:vpos + 1])
      add(1, vpos + 1 - pos)                          # up to and including the anchor base
      if len(refb) == 1 and len(altb) == 1:           # SNP: replace the anchor itself
        seq[-1] = seq[-1][:-1] + altb
      elif len(altb) > len(refb):                     # insertion after the anchor
        seq.append(altb[1:])
        add(2, len(altb) - 1)
      else:                                           # deletion after the anchor
        add(3, len(refb) - 1)
      pos = vpos + len(refb)
    seq.append(ref_seq[pos:end])
    add(1, end - pos)
    bases = list(''.join(seq))
    for j in range(len(bases)):
      if rng.random() < 0.01:
        bases[j] = 'ACGT'[int(rng.integers(0, 4))]
    bases = ''.join(bases)
    reads.append(T.Read(
        fragment_name='m%d' % i, read_number=0, number_reads=1, fragment_length=0,
        aligned_sequence=bases, aligned_quality=bytes(rng.integers(8, 50, size=len(bases)).astype(np.uint8)),
        alignment=T.LinearAlignment(position=T.Position('chr1', start, bool(rng.integers(0, 2))),
                                    mapping_quality=int(rng.integers(1, 61)), cigar=cigar)))
    if rng.random() < 0.7:
      reads[-1].info['HP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(0, 3)))])
  return reads


def _alt_region(mode, types, pacbio, seed=101):
  """Reference, haplotype-carrying reads and candidates (SNPs, insertions, deletions, some
  with two alts) for the alt-aligned tests."""
  rng = np.random.default_rng(seed)
  # pacbio: exactly the released PacBio model's tensor (deepvariant/json/deepvariant.pacbio.savedmodel/
  # model.example_info.json: shape [100, 147, 10], channels [1..7, 26, 9, 10], diff_channels, indels)
  width, height = (147, 100) if pacbio else (99, 40)
  hw = (width - 1) // 2
  channels = list(T.PILEUP_DEFAULT_CHANNELS) + ['haplotype'] + (['supplementary_alignment'] if pacbio else [])
  extra = {'diff_channels': ['diff_channels_alternate_allele_1', 'diff_channels_alternate_allele_2'],
           'base_channels': ['base_channels_alternate_allele_1', 'base_channels_alternate_allele_2']}.get(mode, [])
  pic = F.options(channels + extra, width, height, sort_by_haplotypes=True, min_mapq=1,
                  alt_aligned_pileup=mode, types_to_alt_align=types)
  enc_pic = F.options(channels, width, height, sort_by_haplotypes=True, min_mapq=1)   # what the oracle draws
  options = T.MakeExamplesOptions(
      pic_options=pic, sample_options=[T.SampleOptions(role='main', name='m', pileup_height=height)])
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=5000)))
  variants = []
  for pos in sorted(set(rng.integers(900, 3600, size=26).tolist())):
    kind = int(rng.integers(0, 3))
    if kind == 0:
      refb = ref.seq[pos]
      altb = [b for b in 'ACGT' if b != refb][int(rng.integers(0, 3))]
    elif kind == 1:
      refb = ref.seq[pos]
      altb = refb + ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 9))))
    else:
      refb = ref.seq[pos:pos + 1 + int(rng.integers(1, 9))]
      altb = refb[0]
    if variants and pos < variants[-1][0] + 30:
      continue
    variants.append((pos, refb, altb))
  reads = _haplotype_reads(rng, ref.seq, variants, 420, 600, 3700)
  if pacbio:
    for r in reads:
      r.supplementary_alignment = bool(rng.random() < 0.2)
  cands = []
  for k, (pos, refb, altb) in enumerate(variants):
    alts = [altb]
    if k % 4 == 0:                                          # a second alt: three combinations
      alts.append(refb[0] + 'TTG' if len(altb) <= len(refb) else [b for b in 'ACGT' if b != refb[0]][0])
    support = {a: T.SupportingReads(read_names=[]) for a in alts}
    for r in _query(reads, pos, pos + len(refb)):
      if rng.random() < 0.5:
        support[alts[int(rng.integers(0, len(alts)))]].read_names.append(
            '%s/%d' % (r.fragment_name, r.read_number))
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + len(refb), refb, alts),
                                   allele_support=support))
  return dict(pic=pic, enc_pic=enc_pic, options=options, ref=ref, reads=reads, cands=cands, width=width,
              height=height, hw=hw, channels=channels, extra=extra)


@pytest.mark.parametrize('mode,types,pacbio', [('diff_channels', 'all', False), ('base_channels', 'indels', False),
                                               ('rows', 'all', False), ('single_row', 'indels', False),
                                               ('diff_channels', 'indels', True)])
def test_alt_aligned_pileups(mode, types, pacbio):
  """--alt_aligned_pileup (the PacBio / ONT models use diff_channels): per candidate and alt
  allele the reads are trimmed to the window, realigned to the alt haplotype
  (CreateHaplotype + RealignReadsToHaplotype, make_examples_native.cc:553-626) and drawn
  against it; the alt images become two extra channels or extra row blocks
  (FillPileupArray, pileup_image_native.h:214-307).

  The product draws reference and alt images of the whole region in ONE encoder launch and
  merges on the host; the oracle side restates the reference's per-candidate steps with
  proto-shaped inputs.  The aligner itself is shared by both sides: it is pinned by the
  reference's vectors in tests/test_fast_pass_aligner_cpu.py."""
  from deepvariant_amd import alt_aligned_pileup_lib as A
  from deepvariant_amd import fast_pass_aligner as fpa
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from oracle import oracle as O
  g = _alt_region(mode, types, pacbio)
  pic, enc_pic, options, ref, reads, cands = (g[k] for k in ('pic', 'enc_pic', 'options', 'ref', 'reads', 'cands'))
  width, height, hw, channels, extra = (g[k] for k in ('width', 'height', 'hw', 'channels', 'extra'))
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  stats = {}
  examples, shape = gen.encode_region(cands, [reads], [0], [0.0], stats, role='main')
  mult = {'rows': 3, 'single_row': 2}.get(mode, 1)
  assert shape == [height * mult, width, len(channels) + len(extra)]
  assert not pacbio or shape == [100, 147, 10]
  k = n_alt_images = n_plain = 0
  for cand in cands:
    v = cand.variant
    window = men.get_reference_bases_for_pileup(ref, v, width)
    needs_alt = A.need_alt_alignment(pic, v)
    overlapping = _query(reads, v.start - 5, v.end + 5)
    if needs_alt:
      r0, r1 = A.calculate_alignment_region(v, hw, len(ref.seq))
      drawn, starts = A.trim_reads(overlapping, r0, r1)
    else:
      drawn, starts = overlapping, None
    for combo in men.alt_allele_combinations(cand, pic.multi_allelic_mode):
      img = np.frombuffer(pw.decode_example(examples[k])['image/encoded'][0], np.uint8).reshape(shape)
      k += 1
      want_ref = O.build_pileup(enc_pic, cand, window, drawn, v.start - hw, list(combo),
                                pileup_height=height, alignment_positions=starts)
      alts = [None, None]
      if needs_alt:
        for a, alt in enumerate(combo[:2]):
          hap, h0, h1 = A.create_haplotype(ref, v, alt, hw)
          assert len(hap) >= width
          re = fpa.realign_reads_to_haplotype(hap, drawn, 'chr1', h0, h1, ref, men.DEFAULT_ALN_CONFIG)
          kept = [(r, s) for r, s in zip(re, starts) if r is not None]
          alts[a] = O.build_pileup(enc_pic, cand, hap[:width], [r for r, _ in kept], v.start - hw,
                                   list(combo), pileup_height=height,
                                   alignment_positions=[s for _, s in kept])
          n_alt_images += 1
      else:
        n_plain += 1
      want = A.fill_pileup_array(want_ref, alts, mode, A.get_alt_image_row_indices(mode, list(combo)))
      if want.shape[2] < shape[2]:      # channel modes without alt images keep the two channels at zero
        want = np.concatenate([want, np.zeros(want.shape[:2] + (shape[2] - want.shape[2],), np.uint8)], axis=2)
      np.testing.assert_array_equal(img, want, err_msg='%s: candidate at %d, alts %s' % (mode, v.start, combo))
  assert k == len(examples) and n_alt_images > 10
  assert types == 'all' or n_plain > 0       # 'indels': SNP candidates draw untrimmed reads, no alt images


@pytest.mark.parametrize('mode,types,pacbio', [('diff_channels', 'indels', True), ('rows', 'all', False),
                                               ('base_channels', 'all', True)])
def test_fused_device_path_with_alt_aligned_layouts(mode, types, pacbio):
  """call_variants_in_region with alt-aligned pileups: the alt images are items of the same
  launch, the channel merge runs on the device (dv_merge_alt_channels), nothing returns to the
  host before the CNN.  CallVariantsOutput must equal classifying the examples of
  encode_region (whose tensors test_alt_aligned_pileups checks against the oracle)."""
  import torch
  from deepvariant_amd import call_variants as cv
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from deepvariant_amd.inception_v3 import InceptionV3
  g = _alt_region(mode, types, pacbio)
  gen = men.ExamplesGenerator(g['options'], {}, test_mode=True, ref_reader=g['ref'])
  examples, shape = gen.encode_region(g['cands'], [g['reads']], [0], [0.0], {}, role='main')
  model = InceptionV3(tuple(shape), max_batch=64)
  model.init_random(seed=5)
  cvos = gen.call_variants_in_region(g['cands'], [g['reads']], [0], [0.0], model)
  assert len(cvos) == len(examples) > 20
  imgs = np.stack([np.frombuffer(pw.decode_example(e)['image/encoded'][0], np.uint8).reshape(shape)
                   for e in examples])
  want = cv.round_gls_batch(model(torch.from_numpy(imgs).cuda()).cpu().numpy(), 10)
  for k, (cvo, ex) in enumerate(zip(cvos, examples)):
    _, alt, probs = pw.decode_call_variants_output(cvo)
    assert alt == pw.decode_alt_allele_indices(pw.decode_example(ex)['alt_allele_indices/encoded'][0])
    assert probs == want[k].tolist(), k
