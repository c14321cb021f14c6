"""Loader and shared pieces for the PacBio golden chain (tests/golden/pacbio_full_chr20.npz, made
by tests/golden/make_golden.py pacbio_full): the 281 unclipped HiFi reads, the reference
stretch, and all 401 images of golden.pacbio_examples.tfrecord.gz with their variants.
Flags of the golden: deepvariant/make_examples_test.py:794-818."""
import os

import numpy as np

from deepvariant_amd import dv_types as T
from tests import golden_io
from tests import realigner_fixture as RF

FIXTURE = os.path.join(os.path.dirname(__file__), 'golden', 'pacbio_full_chr20.npz')
REGION = T.Range('chr20', 8_999_999, 9_100_000)        # --regions chr20:9,000,000-9,100,000
PARTITION = 25000
CHANNELS = list(T.PILEUP_DEFAULT_CHANNELS) + ['haplotype', 'base_methylation']


def load():
  """-> (ref_reader, reads, [(start, end, ref, alts, alt_allele_indices)], images[401,100,147,10])."""
  with np.load(FIXTURE) as f:
    z = {k: f[k] for k in f.files}
  reads = golden_io.unpack_reads(z)
  meta = []
  for line in bytes(z['g_meta']).decode().split('\n'):
    start, end, ref, alts, idx = line.split('\t')
    meta.append((int(start), int(end), ref, tuple(alts.split(',')), tuple(int(i) for i in idx.split(','))))
  return RF.FixtureRef(z), reads, meta, z['g_images']


def pic_options(with_alt_channels: bool):
  """PileupImageOptions of the golden run: 8 encoder channels (+ the two diff channels), width 147,
  min_mapping_quality 1, sort_by_haplotypes, alt_aligned_pileup diff_channels for indels."""
  rr = T.ReadRequirements(min_mapping_quality=1, min_base_quality=10, min_base_quality_mode=1)
  o = T.default_options(rr)
  o.channels = list(CHANNELS) + (['diff_channels_alternate_allele_1', 'diff_channels_alternate_allele_2']
                                 if with_alt_channels else [])
  o.num_channels = len(o.channels)
  o.width = 147
  o.sort_by_haplotypes = True
  if with_alt_channels:
    o.alt_aligned_pileup = 'diff_channels'
    o.types_to_alt_align = 'indels'
  return o
