"""HIP encoder vs oracle on synthetic ILLUMINA30 batches (GPU), bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('channels,n,seed', [(7, 768, 11), (6, 300, 5)])
def test_synthetic_batch_bit_exact(channels, n, seed):
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  opts = synth.illumina_options(channels)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts)
  off = np.array(batch.item_list_off)
  assert ((off[1:] - off[:-1]) > 95).any(), 'shuffle path not exercised'
  out, rows = _Encoder(opts, opts.width).encode(batch, channels)
  want, want_rows = O.encode_packed(opts, batch, channels, n_threads=8)
  np.testing.assert_array_equal(rows, want_rows)
  assert (rows < (off[1:] - off[:-1])).any(), 'no rejected read in the batch'
  np.testing.assert_array_equal(out, want)


def test_padded_output_channels_and_empty_items():
  """out_channels > n_channels zero-fills the tail; items without reads."""
  from deepvariant_amd import synth, packing
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(40, seed=3, options=opts, out_channels=8)
  # an item with an empty read list
  batch.add_item(500, 500 - 110, 0, np.zeros(0, np.uint32),
                 np.zeros(0, np.uint8), height=100,
                 out_off=batch.n_items * 100 * 221 * 8)
  out, rows = _Encoder(opts, opts.width).encode(batch, 8)
  want, want_rows = O.encode_packed(opts, batch, 8)
  np.testing.assert_array_equal(rows, want_rows)
  np.testing.assert_array_equal(out, want)
  assert rows[-1] == 0
  img = out.reshape(-1, 100, 221, 8)
  assert not img[..., 7].any()


def test_device_resident_batch_matches_host_batch():
  """dv_batch.memory = DEVICE: pointers straight from torch tensors."""
  import ctypes as C
  import torch
  from deepvariant_amd import synth, _lib
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(128, seed=9, options=opts)
  enc = _Encoder(opts, opts.width)
  want, want_rows = enc.encode(batch, 7)
  dev = DeviceBatch(batch, torch.device('cuda:0'))
  out = torch.empty(batch.out_bytes(7), dtype=torch.uint8, device='cuda:0')
  rows = torch.empty(batch.n_items, dtype=torch.int32, device='cuda:0')
  dev.encode(enc, 7, out, rows)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(out.cpu().numpy(), want)
  np.testing.assert_array_equal(rows.cpu().numpy(), want_rows)


@pytest.mark.parametrize('kind,n', [('hifi', 160), ('ont', 160)])
def test_longread_shapes_bit_exact(kind, n):
  """BASELINE.json configs 4/5 shapes: W=147 (8 ch: haplotype + methylation, '='/'X'
  CIGARs) and W=199 (7 ch, ~14 CIGAR ops/read, 10 % of sites deeper than the image)."""
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  opts = synth.longread_options(kind)
  c = len(opts.channels)
  batch = synth.make_longread_batch(n, kind, seed=77)
  got, got_rows = _Encoder(opts, opts.width).encode(batch, c)
  want, want_rows = O.encode_packed(opts, batch, c, n_threads=8)
  np.testing.assert_array_equal(got_rows, want_rows)
  np.testing.assert_array_equal(got, want)
  if kind == 'ont':
    off = np.asarray(batch.item_list_off)
    assert (np.diff(off) > 95).any()      # the shuffle-and-truncate path ran


def test_pileups_deeper_than_the_dense_shuffle_table():
  """DownsampleReadIndices on amplicon-depth pile-ups (pileup_image_native.cc:153-165): the
  device looks the permutation up in a table that is dense up to 1024 reads and holds only
  the depths that occur above that.  Items of 1500 / 2600 / 5003 list entries (reads of
  their site repeated), host batch and device-resident batch, against the oracle's
  std::shuffle; then a second call with a NEW depth on the same encoder (the table grows)."""
  import torch
  from deepvariant_amd import synth
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  opts = synth.illumina_options(7)
  enc = _Encoder(opts, opts.width)
  rng = np.random.default_rng(5)

  def deep_batch(depths, seed):
    batch = synth.make_illumina_batch(24, seed=seed, options=opts)
    off = np.array(batch.item_list_off)
    reads = np.concatenate(batch.list_read_chunks)
    codes = np.concatenate(batch.list_code_chunks)
    n0 = batch.n_items
    for k, depth in enumerate(depths):
      i = k % n0
      pick = rng.integers(off[i], off[i + 1], size=depth)
      batch.add_item(batch.item_variant_start[i], batch.item_image_start[i],
                     batch.item_ref_idx[i], reads[pick], codes[pick], height=100,
                     out_off=batch.n_items * 100 * 221 * 7)
    return batch

  batch = deep_batch([1500, 2600, 5003, 1025], seed=21)
  out, rows = enc.encode(batch, 7)
  want, want_rows = O.encode_packed(opts, batch, 7, n_threads=8)
  np.testing.assert_array_equal(rows, want_rows)
  np.testing.assert_array_equal(out, want)
  assert (rows[-4:] == 95).all()

  batch2 = deep_batch([3001, 1500], seed=22)     # 3001 is new to the encoder's table
  want2, want_rows2 = O.encode_packed(opts, batch2, 7, n_threads=8)
  dev = DeviceBatch(batch2, torch.device('cuda:0'))
  out2 = torch.empty(batch2.out_bytes(7), dtype=torch.uint8, device='cuda:0')
  rows2 = torch.empty(batch2.n_items, dtype=torch.int32, device='cuda:0')
  dev.encode(enc, 7, out2, rows2)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(rows2.cpu().numpy(), want_rows2)
  np.testing.assert_array_equal(out2.cpu().numpy(), want2)


@pytest.mark.parametrize('kind,n', [('illumina30', 7700), ('hifi35', 4096), ('ont50', 2048)])
def test_bench_sized_batches_equal_the_reference_build(kind, n):
  """BASELINE.json's full sizes against the REFERENCE's own encoder (oracle/_ref/libdvref.so: its sources
  compiled unmodified, oracle/ref_build/): the 7700-site ILLUMINA30 step of bench.py (about 8100 pileups, 1.25 GB of
  pixels) and the two long-read workloads, every byte of every pileup and every row count."""
  import os
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  if not O.reference_available():
    pytest.skip('oracle/_ref/libdvref.so was not built (no reference tree where build() ran)')
  if kind == 'illumina30':
    opts = synth.illumina_options(7)
    batch = synth.make_illumina_batch(n, options=opts)
  else:
    lr = 'hifi' if kind == 'hifi35' else 'ont'
    opts = synth.longread_options(lr)
    batch = synth.make_longread_batch(n, lr, options=opts)
  C = len(opts.channels)
  out, rows = _Encoder(opts, opts.width).encode(batch, C)
  threads = max(1, min(32, len(os.sched_getaffinity(0))))
  with O.reference_backend():
    want, want_rows = O.encode_packed(opts, batch, C, n_threads=threads)
  np.testing.assert_array_equal(rows, want_rows)
  assert out.size == want.size > n * 100 * opts.width * C - 1
  assert np.array_equal(out, want), 'first differing byte: %d' % int(np.flatnonzero(out != want)[0])
