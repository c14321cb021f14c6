#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's own fixtures.

Runs ONLY in the build container (needs /root/reference); the output fixtures
are committed so that the tests never read /root/reference.

Illumina WGS golden (BASELINE.json configs[0]):
  deepvariant/testdata/golden.calling_examples.tfrecord.gz      84 x [100,221,7]
  deepvariant/testdata/golden.calling_candidates.tfrecord.gz    78 DeepVariantCall
  deepvariant/testdata/input/NA12878_S1.chr20.10_10p1mb.bam
  deepvariant/testdata/input/ucsc.hg19.chr20.unittest.fasta.gz
  flags: deepvariant/make_examples_test.py:363-395 (channel list =
  PILEUP_CHANNELS_WITH_INSERT_SIZE, realigner ON, min_mapping_quality 5,
  min_base_quality 10).

The golden images were produced AFTER the reference's realigner rewrote some
reads; this fixture records, next to each golden image, the reads of the *raw*
BAM that overlap the candidate (the realigner's own input is the `realigner`
fixture further down, with which tests/test_oracle_golden.py reproduces all 84
images and 78 candidates exactly).  The raw-read acceptance checks are:
  * reference-band rows bit-exact in 84/84 images,
  * every golden read row that equals the raw-BAM encoding of some read
    (>= 80 % of all rows; the rest are realigner-rewritten reads),
  * images whose reads the realigner left untouched are bit-exact in full.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from deepvariant_amd import dv_types as T  # noqa: E402
from deepvariant_amd import genomics_io, protowire as pw, tfrecord  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import golden_io  # noqa: E402

REF = '/root/reference/deepvariant/testdata'


def wgs_options():
  rr = T.ReadRequirements(min_mapping_quality=5, min_base_quality=10,
                          min_base_quality_mode=1)
  o = T.default_options(rr)
  o.channels = list(T.PILEUP_CHANNELS_WITH_INSERT_SIZE)
  o.num_channels = len(o.channels)
  return o


def reader_filter(reads, min_mapq=5):
  """sam_reader.cc:217-247 with make_examples' default ReadRequirements."""
  out = []
  for r in reads:
    if (r.duplicate_fragment or r.failed_vendor_quality_checks or
        r.secondary_alignment or r.supplementary_alignment):
      continue
    if not (r.number_reads < 2 or r.proper_placement or r._mate_ok):
      continue
    if r.alignment.mapping_quality < min_mapq:
      continue
    out.append(r)
  return out


def alt_combination(call, indices):
  return [call.variant.alternate_bases[i] for i in indices]


def main():
  opts = wgs_options()
  hw = (opts.width - 1) // 2
  fasta = genomics_io.FastaReader(
      os.path.join(REF, 'input/ucsc.hg19.chr20.unittest.fasta.gz'))
  _, reads = genomics_io.read_bam(
      os.path.join(REF, 'input/NA12878_S1.chr20.10_10p1mb.bam'), 'chr20',
      9_999_000, 10_012_000)
  reads = reader_filter(reads)
  cands = {}
  for rec in tfrecord.read_tfrecords(
      os.path.join(REF, 'golden.calling_candidates.tfrecord.gz')):
    c = pw.decode_deepvariant_call(rec)
    cands[(c.variant.start, tuple(c.variant.alternate_bases))] = c
  examples = []
  n_rows = n_match = n_full = n_ref_ok = 0
  for rec in tfrecord.read_tfrecords(
      os.path.join(REF, 'golden.calling_examples.tfrecord.gz'),
      verify_crc=True):
    ex = pw.decode_example(rec)
    shape = ex['image/shape']
    img = np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(shape)
    v = pw.decode_variant(ex['variant/encoded'][0])
    call = cands[(v.start, tuple(v.alternate_bases))]
    combo = alt_combination(call, pw.decode_alt_allele_indices(
        ex['alt_allele_indices/encoded'][0]))
    start = v.start - hw
    n_bases = fasta.n_bases(v.reference_name)
    window = fasta.get_bases(v.reference_name, max(start, 0),
                             min(start + opts.width, n_bases))
    window = 'N' * max(-start, 0) + window
    window += 'N' * (opts.width - len(window))
    q0, q1 = v.start - opts.read_overlap_buffer_bp, v.end + opts.read_overlap_buffer_bp
    idx = [i for i, r in enumerate(reads) if O.read_overlaps(r, q0, q1)]
    got, kept, row_read = O.build_pileup(
        opts, call, window, [reads[i] for i in idx], start, combo,
        return_row_reads=True)
    band = opts.reference_band_height
    n_ref_ok += int((got[:band] == img[:band]).all())
    ours = {got[r].tobytes() for r in range(band, band + kept)}
    gold_rows = [r for r in range(band, shape[0]) if img[r].any()]
    n_rows += len(gold_rows)
    n_match += sum(img[r].tobytes() in ours for r in gold_rows)
    full = bool((got == img).all())
    n_full += int(full)
    examples.append(dict(call=call, alt_alleles=combo, ref_window=window,
                         read_idx=idx, image=img, full=full))
  print('images', len(examples), 'ref-band exact', n_ref_ok,
        'read rows', n_rows, 'matched', n_match,
        '(%.1f%%)' % (100.0 * n_match / n_rows), 'fully exact', n_full)
  used = sorted({i for e in examples for i in e['read_idx']})
  remap = {old: new for new, old in enumerate(used)}
  for e in examples:
    e['read_idx'] = [remap[i] for i in e['read_idx']]
  golden_io.save(
      os.path.join(ROOT, 'tests/golden/illumina_wgs_chr20.npz'),
      [reads[i] for i in used], examples,
      e_full=np.array([e['full'] for e in examples], np.uint8),
      stats=np.array([len(examples), n_ref_ok, n_rows, n_match, n_full],
                     np.int64))


if __name__ == '__main__' and not any(a.startswith(('pacbio', 'realigner', 'illumina_alt', 'nucleus_sam', 'na12878_100kb')) for a in sys.argv[1:]):
  main()


# ---------------------------------------------------------------------------
# PacBio golden (BASELINE.json configs[3] shape)
#   deepvariant/testdata/golden.pacbio_examples.tfrecord.gz     401 x [100,147,10]
#   deepvariant/testdata/input/test_pacbio.chr20_100kbp_at_9mb.bam  (285 HiFi reads)
#   deepvariant/testdata/input/grch38.chr20_and_21_10M.fa.gz
#   flags: deepvariant/make_examples_test.py:794-818 (8 encoder channels = default 6 +
#   haplotype + base_methylation, + 2 alt-aligned diff channels; realigner OFF,
#   min_mapping_quality 1, width 147, trim_reads_for_pileup, direct phasing).
# The candidates (allele_support) and the phasing (HP) are not in the testdata, so the
# fixture pins what the raw BAM + FASTA determine: all 8 encoder channels of the
# reference-band rows, and channels {read_base, base_quality, mapping_quality, strand,
# base_differs_from_ref} of every read row.  Reads are clipped to the window +-20 bp here
# (the reference trims them too; in-window pixels do not change), which keeps the fixture
# small.
# ---------------------------------------------------------------------------
PACBIO_CHANNELS = list(T.PILEUP_DEFAULT_CHANNELS) + ['haplotype', 'base_methylation']
PACBIO_CHECKED = [0, 1, 2, 3, 5]   # indices of the pinned read-row channels


def pacbio_options():
  rr = T.ReadRequirements(min_mapping_quality=1, min_base_quality=10, min_base_quality_mode=1)
  o = T.default_options(rr)
  o.channels = list(PACBIO_CHANNELS)
  o.num_channels = len(o.channels)
  o.width = 147
  o.sort_by_haplotypes = True
  return o


def clip_read(read, lo, hi):
  """The part of `read` aligned to reference [lo, hi): new position, CIGAR, bases, quals.
  Indels are kept when they start inside the interval (their anchor pixel is drawn at the
  base before them, whatever follows)."""
  ref_i = read.alignment.position.position
  read_i = 0
  new_pos = None
  cigar, s_lo, s_hi = [], None, None

  def take(op, ln, r0, n_read):
    nonlocal s_lo, s_hi
    if n_read:
      s_lo = r0 if s_lo is None else s_lo
      s_hi = r0 + n_read
    if cigar and cigar[-1].operation == op:
      cigar[-1] = T.CigarUnit(op, cigar[-1].operation_length + ln)
    else:
      cigar.append(T.CigarUnit(op, ln))

  for cu in read.alignment.cigar:
    op, ln = cu.operation, cu.operation_length
    if op in (1, 8, 9):
      a, b = max(ref_i, lo), min(ref_i + ln, hi)
      if b > a:
        if new_pos is None:
          new_pos = a
        take(op, b - a, read_i + (a - ref_i), b - a)
      ref_i += ln
      read_i += ln
    elif op == 2:
      if new_pos is not None and lo < ref_i < hi:
        take(op, ln, read_i, ln)
      read_i += ln
    elif op in (3, 4):
      if new_pos is not None and lo < ref_i < hi:   # its anchor pixel sits at ref_i - 1
        take(op, ln, read_i, 0)
      ref_i += ln
    elif op == 5:
      read_i += ln
  if new_pos is None:
    return None
  out = T.Read(
      fragment_name=read.fragment_name, read_number=read.read_number,
      number_reads=read.number_reads, fragment_length=read.fragment_length,
      aligned_sequence=read.aligned_sequence[s_lo:s_hi],
      aligned_quality=bytes(bytearray(read.aligned_quality))[s_lo:s_hi],
      alignment=T.LinearAlignment(
          position=T.Position(read.alignment.position.reference_name, new_pos,
                              read.alignment.position.reverse_strand),
          mapping_quality=read.alignment.mapping_quality, cigar=cigar))
  return out


def main_pacbio(n_keep=120):
  opts = pacbio_options()
  hw = (opts.width - 1) // 2
  band = opts.reference_band_height
  fasta = genomics_io.FastaReader(os.path.join(REF, 'input/grch38.chr20_and_21_10M.fa.gz'))
  _, reads = genomics_io.read_bam(
      os.path.join(REF, 'input/test_pacbio.chr20_100kbp_at_9mb.bam'), 'chr20',
      8_900_000, 9_200_000)
  reads = [r for r in reads if not (r.duplicate_fragment or r.failed_vendor_quality_checks or
                                    r.secondary_alignment or r.supplementary_alignment)
           and r.alignment.mapping_quality >= 1]
  examples_all = list(tfrecord.read_tfrecords(
      os.path.join(REF, 'golden.pacbio_examples.tfrecord.gz'), verify_crc=True))
  n_rows = n_match = n_ref_ok = 0
  kept_examples, kept_reads = [], []
  step = max(1, len(examples_all) // n_keep)
  for k, rec in enumerate(examples_all):
    ex = pw.decode_example(rec)
    shape = ex['image/shape']
    img = np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(shape)
    v = pw.decode_variant(ex['variant/encoded'][0])
    start = v.start - hw
    window = fasta.get_bases(v.reference_name, start, start + opts.width)
    call = T.DeepVariantCall(variant=v)
    q0, q1 = v.start - opts.read_overlap_buffer_bp, v.end + opts.read_overlap_buffer_bp
    clipped = []
    for r in reads:
      if O.read_overlaps(r, q0, q1):
        c = clip_read(r, start - 20, start + opts.width + 20)
        if c is not None:
          clipped.append(c)
    ref_row = O.encode_reference(opts, window)
    n_ref_ok += int(all((img[r, :, :8] == ref_row[0]).all() for r in range(band)))
    ours = set()
    for r in clipped:
      row = O.encode_read(opts, call, window, r, start, [])
      if row is not None:
        ours.add(row[0][:, PACBIO_CHECKED].tobytes())
    gold = [r for r in range(band, shape[0]) if img[r, :, PACBIO_CHECKED].any()]
    n_rows += len(gold)
    n_match += sum(np.ascontiguousarray(img[r][:, PACBIO_CHECKED]).tobytes() in ours for r in gold)
    if k % step == 0:
      kept_examples.append(dict(call=call, alt_alleles=[], ref_window=window,
                                read_idx=list(range(len(kept_reads), len(kept_reads) + len(clipped))),
                                image=img, full=False))
      kept_reads.extend(clipped)
  print('pacbio images', len(examples_all), 'ref-band exact', n_ref_ok, 'read rows', n_rows,
        'matched', n_match, '(%.2f%%)' % (100.0 * n_match / max(n_rows, 1)),
        'kept', len(kept_examples), 'examples /', len(kept_reads), 'clipped reads')
  golden_io.save(
      os.path.join(ROOT, 'tests/golden/pacbio_chr20.npz'), kept_reads, kept_examples,
      stats=np.array([len(examples_all), n_ref_ok, n_rows, n_match], np.int64))


if __name__ == '__main__' and 'pacbio' in sys.argv[1:] and 'pacbio_full' not in sys.argv[1:]:
  main_pacbio()


# ---------------------------------------------------------------------------
# PacBio golden, alt-aligned channels (channels 8 and 9 of golden.pacbio_examples: the
# base_differs_from_ref channel of the reads realigned to the haplotype of alt 1 / alt 2,
# make_examples_native.cc:553-626, pileup_image_native.h:246-271).
# What the raw BAM + FASTA determine: for every indel candidate (types_to_alt_align =
# "indels") the SET of alt-image rows -- the reference sorts them by phasing tags the
# testdata does not carry.  This pins trim_reads + create_haplotype + the realigner
# (FastPassAligner and the libssw restatement) against the reference's own output.
# ---------------------------------------------------------------------------
def alt_rows_for_example(opts, fasta, v, combo, trimmed, realign):
  """-> [set of diff-channel rows of alt image k] for the (<= 2) alts of `combo`."""
  from deepvariant_amd import alt_aligned_pileup_lib as A
  hw = (opts.width - 1) // 2
  call = T.DeepVariantCall(variant=v)
  out = []
  for alt in combo[:2]:
    hap, h0, h1 = A.create_haplotype(fasta, v, alt, hw)
    rows = set()
    for r in realign(hap, trimmed, h0, h1):
      if r is None:
        continue
      row = O.encode_read(opts, call, hap[:opts.width], r, v.start - hw, [])
      if row is not None:
        rows.add(np.ascontiguousarray(row[0][:, 5]).tobytes())
    out.append(rows)
  return out


def main_pacbio_alt(keep_every=2):
  from deepvariant_amd import alt_aligned_pileup_lib as A
  from deepvariant_amd import fast_pass_aligner as fpa
  from deepvariant_amd import make_examples_native as men
  opts = pacbio_options()
  hw = (opts.width - 1) // 2
  band = opts.reference_band_height
  fasta = genomics_io.FastaReader(os.path.join(REF, 'input/grch38.chr20_and_21_10M.fa.gz'))
  _, reads = genomics_io.read_bam(
      os.path.join(REF, 'input/test_pacbio.chr20_100kbp_at_9mb.bam'), 'chr20',
      8_900_000, 9_200_000)
  reads = [r for r in reads if not (r.duplicate_fragment or r.failed_vendor_quality_checks or
                                    r.secondary_alignment or r.supplementary_alignment)
           and r.alignment.mapping_quality >= 1]
  pic = T.PileupImageOptions(alt_aligned_pileup='diff_channels', types_to_alt_align='indels')
  realign = lambda hap, trimmed, h0, h1: fpa.realign_reads_to_haplotype(
      hap, trimmed, 'chr20', h0, h1, fasta, men.DEFAULT_ALN_CONFIG)
  n_images = n_need = 0
  rows, hits = [0, 0], [0, 0]
  kept_examples, kept_reads = [], []
  for rec in tfrecord.read_tfrecords(os.path.join(REF, 'golden.pacbio_examples.tfrecord.gz'),
                                     verify_crc=True):
    ex = pw.decode_example(rec)
    shape = ex['image/shape']
    img = np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(shape)
    v = pw.decode_variant(ex['variant/encoded'][0])
    combo = alt_combination(T.DeepVariantCall(variant=v),
                            pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0]))
    n_images += 1
    if not A.need_alt_alignment(pic, v):
      assert not img[:, :, 8:].any()          # SNPs: the two channels stay zero
      continue
    n_need += 1
    overlapping = [r for r in reads if O.read_overlaps(
        r, v.start - opts.read_overlap_buffer_bp, v.end + opts.read_overlap_buffer_bp)]
    r0, r1 = A.calculate_alignment_region(v, hw, fasta.n_bases('chr20'))
    trimmed, _ = A.trim_reads(overlapping, r0, r1)
    ours = alt_rows_for_example(opts, fasta, v, combo, trimmed, realign)
    if len(combo) == 1:
      assert (img[:, :, 9] == img[:, :, 8]).all()   # a missing alt 2 repeats alt 1
    for a in range(len(ours)):
      gold = [r for r in range(band, shape[0]) if img[r, :, 8 + a].any()]
      rows[a] += len(gold)
      hits[a] += sum(np.ascontiguousarray(img[r, :, 8 + a]).tobytes() in ours[a] for r in gold)
    if n_need % keep_every == 0:
      haps = [A.create_haplotype(fasta, v, alt, hw) for alt in combo[:2]]
      kept_examples.append(dict(
          call=T.DeepVariantCall(variant=v), alt_alleles=combo, ref_window=haps[0][0],
          read_idx=list(range(len(kept_reads), len(kept_reads) + len(trimmed))),
          image=np.ascontiguousarray(img[:, :, 8:]), full=False,
          hap2=haps[1][0] if len(haps) > 1 else '', hap_range=(haps[0][1], haps[0][2])))
      kept_reads.extend(trimmed)
  print('pacbio alt: images', n_images, 'indel candidates', n_need, 'alt rows', rows, 'reproduced', hits,
        'kept', len(kept_examples), 'examples /', len(kept_reads), 'trimmed reads')
  hap2 = '\n'.join(e['hap2'] for e in kept_examples)
  golden_io.save(
      os.path.join(ROOT, 'tests/golden/pacbio_alt_chr20.npz'), kept_reads, kept_examples,
      hap2=np.frombuffer(hap2.encode(), np.uint8),
      hap_range=np.array([e['hap_range'] for e in kept_examples], np.int64),
      stats=np.array([n_images, n_need] + rows + hits, np.int64))


if __name__ == '__main__' and 'pacbio_alt' in sys.argv[1:]:
  main_pacbio_alt()


# ---------------------------------------------------------------------------
# Window realigner fixture
#   realigner_chr20.npz:
#     ref_bases / ref_start    chr20:9,995,000-10,100,600 of ucsc.hg19.chr20.unittest.fasta.gz
#     n_contig_bases           63,025,520
#     reads (golden_io)        every read of NA12878_S1.chr20.10_10p1mb.bam that the tests below use
#     sets_*                   named read subsets (indices into `reads`):
#       wgs     reads make_examples sees for --regions chr20:10,000,000-10,010,000 (reader filter as
#               in main()), i.e. the input of the realigner for golden.calling_examples
#       ex1/ex2 reads overlapping the two regions of realigner_test.py:296-360
#               (test_realigner_example_region; SamReader without read requirements)
#       dbg0    reads overlapping chr20:10,000,000-10,000,100 (debruijn_graph_wrap_test.py
#               test_straightforward_region); test_complex_region uses ex1's region
# ---------------------------------------------------------------------------
def main_realigner():
  fasta = genomics_io.FastaReader(os.path.join(REF, 'input/ucsc.hg19.chr20.unittest.fasta.gz'))
  bam = os.path.join(REF, 'input/NA12878_S1.chr20.10_10p1mb.bam')
  ref_start, ref_end = 9_995_000, 10_100_600
  wanted = dict(wgs=(10_000_000, 10_010_000, True), ex1=(10_095_378, 10_095_500, False),
                ex2=(10_046_079, 10_046_307, False), dbg0=(9_999_999, 10_000_100, False))
  reads, index_of, sets = [], {}, {}
  for name, (lo, hi, filtered) in wanted.items():
    _, rs = genomics_io.read_bam(bam, 'chr20', lo, hi)
    if filtered:
      rs = reader_filter(rs)
    ids = []
    for r in rs:
      key = (r.fragment_name, r.read_number, r.alignment.position.position)
      if key not in index_of:
        index_of[key] = len(reads)
        reads.append(r)
      ids.append(index_of[key])
    sets['sets_' + name] = np.array(ids, np.int32)
    print(name, len(ids), 'reads')
  d = golden_io.pack_reads(reads)
  d.update(sets)
  d['ref_bases'] = np.frombuffer(fasta.get_bases('chr20', ref_start, ref_end).encode(), np.uint8)
  d['ref_start'] = np.array([ref_start], np.int64)
  d['n_contig_bases'] = np.array([fasta.n_bases('chr20')], np.int64)
  # the Variant protos inside golden.calling_examples: what calls[0] must carry (AD / DP / VAF, the
  # no-call genotype, the sample name) -- one JSON line per distinct variant
  import json
  lines = {}
  for rec in tfrecord.read_tfrecords(os.path.join(REF, 'golden.calling_examples.tfrecord.gz')):
    v = pw.decode_variant(pw.decode_example(rec)['variant/encoded'][0])
    c = v.calls[0]
    lines[(v.start, tuple(v.alternate_bases))] = json.dumps(dict(
        start=v.start, end=v.end, ref=v.reference_bases, alts=list(v.alternate_bases), sample=c.call_set_name,
        genotype=list(c.genotype), AD=[x.int_value for x in c.info['AD'].values],
        DP=[x.int_value for x in c.info['DP'].values],
        VAF=[float(x.number_value).hex() for x in c.info['VAF'].values]))
  # golden.candidate_positions (make_examples_test.py candidate_sweep mode): int32 positions of the raw-read
  # candidates per 1 kb partition, -2 after each partition, -1 at the end of the region
  d['wgs_candidate_positions'] = np.fromfile(os.path.join(REF, 'golden.candidate_positions'), np.int32)
  d['wgs_variants'] = np.frombuffer('\n'.join(lines[k] for k in sorted(lines)).encode(), np.uint8)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/realigner_chr20.npz'), **d)


if __name__ == '__main__' and 'realigner' in sys.argv[1:]:
  main_realigner()


# ---------------------------------------------------------------------------
# PacBio golden, whole chain (BASELINE.json configs[3] shape; make_examples_test.py:794-818)
#   pacbio_full_chr20.npz:
#     reads (golden_io)       the 281 HiFi reads make_examples reads for chr20:9,000,000-9,100,000
#                             (min_mapping_quality 1, default read filter), UNCLIPPED
#     ref_bases / ref_start   chr20:8,984,001-9,115,000 of grch38.chr20_and_21_10M.fa.gz
#     g_images                golden.pacbio_examples.tfrecord.gz, all 401 x [100,147,10]
#     g_meta                  per example: start, end, reference_bases, alternate_bases, alt_allele_indices
# ---------------------------------------------------------------------------
def main_pacbio_full():
  fasta = genomics_io.FastaReader(os.path.join(REF, 'input/grch38.chr20_and_21_10M.fa.gz'))
  _, reads = genomics_io.read_bam(
      os.path.join(REF, 'input/test_pacbio.chr20_100kbp_at_9mb.bam'), 'chr20', 8_999_999, 9_100_000)
  reads = [r for r in reads if not (r.duplicate_fragment or r.failed_vendor_quality_checks or
                                    r.secondary_alignment or r.supplementary_alignment)
           and r.alignment.mapping_quality >= 1]
  for r in reads:
    r.info.pop('HP', None)
  images, meta = [], []
  for rec in tfrecord.read_tfrecords(os.path.join(REF, 'golden.pacbio_examples.tfrecord.gz'), verify_crc=True):
    ex = pw.decode_example(rec)
    v = pw.decode_variant(ex['variant/encoded'][0])
    idx = pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0])
    images.append(np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(ex['image/shape']))
    meta.append('\t'.join([str(v.start), str(v.end), v.reference_bases, ','.join(v.alternate_bases),
                           ','.join(str(i) for i in idx)]))
  ref_start, ref_end = 8_984_000, 9_115_000
  d = golden_io.pack_reads(reads)
  d['ref_bases'] = np.frombuffer(fasta.get_bases('chr20', ref_start, ref_end).encode(), np.uint8)
  d['ref_start'] = np.array([ref_start], np.int64)
  d['n_contig_bases'] = np.array([fasta.n_bases('chr20')], np.int64)
  d['g_images'] = np.stack(images)
  d['g_meta'] = np.frombuffer('\n'.join(meta).encode(), np.uint8)
  print(len(reads), 'reads', len(images), 'images', d['g_images'].shape)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/pacbio_full_chr20.npz'), **d)


if __name__ == '__main__' and 'pacbio_full' in sys.argv[1:]:
  main_pacbio_full()


# ---------------------------------------------------------------------------
# Illumina goldens with alt-aligned pileups (make_examples_test.py:736-792: training mode, 6 default
# channels, realigner on, alt_aligned_pileup = rows | diff_channels, types_to_alt_align = indels)
#   illumina_alt_aligned_chr20.npz: per mode the 49 labelled examples of
#   golden.alt_aligned_pileup_{rows,diff_channels}_examples.tfrecord.gz -- images, variants, alt indices.
#   Their inputs are the `wgs` reads / reference of realigner_chr20.npz.
# ---------------------------------------------------------------------------
def main_illumina_alt():
  d = {}
  for mode in ('rows', 'diff_channels'):
    images, meta = [], []
    for rec in tfrecord.read_tfrecords(
        os.path.join(REF, 'golden.alt_aligned_pileup_%s_examples.tfrecord.gz' % mode), verify_crc=True):
      ex = pw.decode_example(rec)
      v = pw.decode_variant(ex['variant/encoded'][0])
      idx = pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0])
      images.append(np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(ex['image/shape']))
      meta.append('\t'.join([str(v.start), str(v.end), v.reference_bases, ','.join(v.alternate_bases),
                             ','.join(str(i) for i in idx)]))
    d[mode + '_images'] = np.stack(images)
    d[mode + '_meta'] = np.frombuffer('\n'.join(meta).encode(), np.uint8)
    print(mode, d[mode + '_images'].shape)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/illumina_alt_aligned_chr20.npz'), **d)


if __name__ == '__main__' and 'illumina_alt' in sys.argv[1:]:
  main_illumina_alt()


# ---------------------------------------------------------------------------
# nucleus' SamReader test inputs (third_party/nucleus/testdata/): test.bam (+ .bai), test.sam and
# its golden Read protos, test_oq.sam -- the files sam_reader_test.cc runs on -- bundled as bytes.
#   nucleus_sam.npz: one uint8 array per file; tests/test_bam_reference_vectors_cpu.py writes
#   them back to a temporary directory.
# ---------------------------------------------------------------------------
def main_nucleus_sam():
  src = '/root/reference/third_party/nucleus/testdata'
  d = {}
  for name in ('test.bam', 'test.bam.bai', 'test.sam', 'test.sam.golden.tfrecord', 'test_oq.sam'):
    d[name.replace('.', '_')] = np.frombuffer(open(os.path.join(src, name), 'rb').read(), np.uint8)
    print(name, d[name.replace('.', '_')].size)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/nucleus_sam.npz'), **d)


if __name__ == '__main__' and 'nucleus_sam' in sys.argv[1:]:
  main_nucleus_sam()


# ---------------------------------------------------------------------------
# BASELINE.json configs[0] as FILES for `bench.py --mode bam`: the reference tree's NA12878 30x
# Illumina slice (deepvariant/testdata/input/NA12878_S1.chr20.10_10p1mb.bam + .bai, reads of
# chr20:10,000,000-10,100,000) and the hg19 chr20 stretch around it from
# ucsc.hg19.chr20.unittest.fasta.gz, bundled as bytes -- /root/reference does not exist on the GPU box.
#   na12878_100kb.npz: bam, bai (uint8), ref_bases (uint8, chr20:[ref_start, ref_start + len)),
#   ref_start, n_contig_bases.  bench.py writes them back into a temporary directory.
# ---------------------------------------------------------------------------
def main_na12878_100kb():
  bam = os.path.join(REF, 'input/NA12878_S1.chr20.10_10p1mb.bam')
  fasta = genomics_io.FastaReader(os.path.join(REF, 'input/ucsc.hg19.chr20.unittest.fasta.gz'))
  lo, hi = 9_990_000, 10_110_000
  n = fasta.n_bases('chr20')
  hi = min(hi, n)
  d = dict(bam=np.frombuffer(open(bam, 'rb').read(), np.uint8),
           bai=np.frombuffer(open(bam + '.bai', 'rb').read(), np.uint8),
           ref_bases=np.frombuffer(fasta.get_bases('chr20', lo, hi).encode(), np.uint8),
           ref_start=np.array([lo], np.int64), n_contig_bases=np.array([n], np.int64))
  for k, v in d.items():
    print(k, v.size)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/na12878_100kb.npz'), **d)


if __name__ == '__main__' and 'na12878_100kb' in sys.argv[1:]:
  main_na12878_100kb()


def main_cram():
  """CRAM fixtures: the reference tree's NA12878 slice as CRAM 3.0 (the file its make_examples
  test runs with USE_CRAM, deepvariant/make_examples_test.py:330-369; the same alignments as the BAM
  bundled in na12878_100kb.npz) and nucleus' own CRAM test files with their SAM text and FASTA
  (third_party/nucleus/io/sam_test.py:250-300)."""
  nucleus = os.path.join(os.path.dirname(REF), '..', 'third_party', 'nucleus', 'testdata')
  files = {
      'na12878_cram': os.path.join(REF, 'input/NA12878_S1.chr20.10_10p1mb.cram'),
      'na12878_crai': os.path.join(REF, 'input/NA12878_S1.chr20.10_10p1mb.cram.crai'),
      'nucleus_embed_ref_0': os.path.join(nucleus, 'test_cram.embed_ref_0_version_3.0.cram'),
      'nucleus_embed_ref_1': os.path.join(nucleus, 'test_cram.embed_ref_1_version_3.0.cram'),
      'nucleus_sam': os.path.join(nucleus, 'test_cram.sam'),
      'nucleus_fasta': os.path.join(nucleus, 'test.fasta'),
  }
  d = {k: np.frombuffer(open(v, 'rb').read(), np.uint8) for k, v in files.items()}
  for k, v in d.items():
    print(k, v.size)
  np.savez_compressed(os.path.join(ROOT, 'tests/golden/cram.npz'), **d)


if __name__ == '__main__' and 'cram' in sys.argv[1:]:
  main_cram()

