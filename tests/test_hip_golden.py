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


def test_pacbio_golden_rows_hip():
  """Real HiFi reads ('=' / 'X' / I / D CIGARs, W = 147, 8 channels): every read of the
  134 fixture examples goes through ONE dv_encode_batch call as an EncodeRead item; the
  rows must equal the oracle's bit for bit and contain every golden read row."""
  from deepvariant_amd import packing
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  from tests.golden.make_golden import pacbio_options
  from tests.test_oracle_golden import PACBIO_FIXTURE, check_pacbio_example
  from tests.golden.make_golden import PACBIO_CHECKED
  reads, examples, _ = golden_io.load(PACBIO_FIXTURE)
  opts = pacbio_options()
  band, w, c = opts.reference_band_height, opts.width, 8
  hw = (w - 1) // 2
  table = packing.ReadTable.from_reads(reads)
  batch = packing.PackedBatch(table=table, width=w)
  item_bytes = (band + 1) * w * c
  owner = []
  for e, ex in enumerate(examples):
    call = ex['call']
    ref_idx = batch.add_ref_window(ex['ref_window'])
    for k in ex['read_idx']:
      idx = np.array([k], np.uint32)
      batch.add_item(call.variant.start, call.variant.start - hw, ref_idx, idx,
                     np.zeros(1, np.uint8), height=band + 1, out_off=len(owner) * item_bytes)
      owner.append(e)
  out, rows = _Encoder(opts, w).encode(batch, c)
  want, want_rows = O.encode_packed(opts, batch, c, n_threads=8)
  np.testing.assert_array_equal(rows, want_rows)
  np.testing.assert_array_equal(out, want)
  got = out.reshape(len(owner), band + 1, w, c)
  per_example = [set() for _ in examples]
  for i, e in enumerate(owner):
    if rows[i]:
      per_example[e].add(np.ascontiguousarray(got[i, band][:, PACBIO_CHECKED]).tobytes())
  n_rows = n_hit = 0
  for e, ex in enumerate(examples):
    a, b = check_pacbio_example(opts, ex, per_example[e], got[[o for o in range(len(owner)) if owner[o] == e][0], 0:1]
                                if ex['read_idx'] else O.encode_reference(opts, ex['ref_window']))
    n_rows += a
    n_hit += b
  assert n_hit == n_rows and n_rows > 4000
