"""Pins the CPU oracle to the reference's golden TFRecord images (CPU only).

Fixture: tests/golden/illumina_wgs_chr20.npz, made by tests/golden/make_golden.py
from deepvariant/testdata/golden.calling_examples.tfrecord.gz (84 x 100x221x7,
HG001 chr20:10,000,000-10,010,000 = BASELINE.json configs[0]).
"""
import os

import numpy as np
import pytest

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
