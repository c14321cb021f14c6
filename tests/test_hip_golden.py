"""HIP encoder vs the oracle and vs the reference's golden images (GPU).

All 84 golden inputs (tests/golden/illumina_wgs_chr20.npz = BASELINE.json
configs[0]) go through ONE dv_encode_batch call; the result must equal the
oracle bit for bit on every image, and reproduce the golden-image facts the
oracle was pinned with (reference bands, untouched images).
"""
import os

import numpy as np
import pytest

from tests import golden_io
from tests.golden.make_golden import wgs_options

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(__file__), 'golden',
                       'illumina_wgs_chr20.npz')


def test_golden_batch_bit_exact():
  from deepvariant_amd import packing
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  reads, examples, z = golden_io.load(FIXTURE)
  opts = wgs_options()
  hw = (opts.width - 1) // 2
  table = packing.ReadTable.from_reads(reads)
  batch = packing.PackedBatch(table=table, width=opts.width)
  img_bytes = 100 * 221 * 7
  for i, ex in enumerate(examples):
    call = ex['call']
    idx = np.array(ex['read_idx'], np.uint32)
    batch.add_item(call.variant.start, call.variant.start - hw,
                   batch.add_ref_window(ex['ref_window']), idx,
                   packing.support_codes(call, ex['alt_alleles'], table, idx),
                   height=100, out_off=i * img_bytes)
  out, rows = _Encoder(opts, opts.width).encode(batch, 7)
  got = out.reshape(len(examples), 100, 221, 7)
  want, want_rows = O.encode_packed(opts, batch, 7)
  np.testing.assert_array_equal(rows, want_rows)
  np.testing.assert_array_equal(got, want.reshape(got.shape))
  band = opts.reference_band_height
  n_full = 0
  for i, ex in enumerate(examples):
    np.testing.assert_array_equal(got[i, :band], ex['image'][:band])
    if z['e_full'][i]:
      np.testing.assert_array_equal(got[i], ex['image'])
      n_full += 1
  assert n_full == 7
