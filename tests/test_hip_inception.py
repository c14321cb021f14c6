"""HIP Inception-v3 (fp16 MFMA) vs the fp32 CPU restatement (GPU).

Tolerance: |softmax_hip - softmax_fp32| <= 1e-3 (BASELINE.json north_star).
The reference pins no CNN numerics ("parity unpinned"), so the
oracle is the fp32 torch restatement with seeded random weights, exactly like
deepvariant/call_variants_test.py:109-127.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _per_layer_model(shape, max_batch):
  """A model built with DV_NO_STEM_FUSE: one launch (and one HBM tensor) per stem layer."""
  import os
  from deepvariant_amd.inception_v3 import InceptionV3
  os.environ['DV_NO_STEM_FUSE'] = '1'
  try:
    return InceptionV3(shape, max_batch=max_batch)
  finally:
    del os.environ['DV_NO_STEM_FUSE']


def _pileups(n, channels, seed):
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(channels)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts,
                                    multi_allelic=False)
  out, _ = _Encoder(opts, opts.width).encode(batch, channels)
  return out.reshape(-1, 100, 221, channels)[:n]


@pytest.mark.parametrize('channels', [7, 6])
def test_softmax_within_1e3_of_fp32_oracle(channels):
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  ref = R.make_random_model(channels, seed=3)
  model = InceptionV3((100, 221, channels), max_batch=8)
  assert model.num_params == ref.num_keras_params()
  model.load_flat_weights(ref.export_flat())
  imgs = np.concatenate([
      _pileups(6, channels, seed=21),
      np.random.default_rng(0).integers(0, 256, (4, 100, 221, channels),
                                        dtype=np.uint8)])
  x = torch.from_numpy(imgs)
  got = model(x.cuda()).cpu()          # 10 examples, max_batch 8 -> two chunks
  with torch.no_grad():
    want = ref(x)
  assert got.shape == (10, 3)
  assert torch.allclose(got.sum(1), torch.ones(10), atol=1e-5)
  err = (got - want).abs().max().item()
  assert err <= TOL, err
  # the oracle's answers must differ between inputs for the test to mean much
  assert (want.max(0).values - want.min(0).values).max() > 1e-2


@pytest.mark.parametrize('shape', [(100, 147, 10), (100, 199, 9), (75, 75, 1), (120, 301, 16)])
def test_other_model_shapes_within_1e3(shape):
  """PACBIO (100x147x10), ONT (100x199x9) and the extremes dv_model_create accepts."""
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  h, w, c = shape
  ref = R.make_random_model(c, seed=5)
  model = InceptionV3(shape, max_batch=4)
  assert model.num_params == ref.num_keras_params()
  model.load_flat_weights(ref.export_flat())
  x = torch.from_numpy(np.random.default_rng(c).integers(0, 256, (5, h, w, c), dtype=np.uint8))
  got = model(x.cuda()).cpu()
  with torch.no_grad():
    want = ref(x)
  err = (got - want).abs().max().item()
  assert err <= TOL, (shape, err)


def test_first_conv_matches_activation_by_activation():
  """Localises errors: the first conv (fused with the uint8 preprocessing,
  (x-128)/128, deepvariant/dv_utils.py:343-366) vs torch fp32."""
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  ref = R.make_random_model(7, seed=5)
  model = _per_layer_model((100, 221, 7), 4)   # the fused stem never materialises this tensor
  model.load_flat_weights(ref.export_flat())
  x = torch.from_numpy(_pileups(2, 7, seed=4))
  model(x.cuda())
  want_pre = (x.float() - 128.0) / 128.0
  with torch.no_grad():
    want = ref.stem[0](want_pre.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
  got = model.debug_tensor(1, 2).astype(np.float32)
  # buffer 1 feeds a 'valid' conv: no halo, so the padded plane is the tensor
  assert got.shape == want.shape == (2, 49, 110, 32)
  np.testing.assert_allclose(got, want, atol=2e-2, rtol=2e-2)


def test_shape_mismatch_raises():
  from deepvariant_amd.inception_v3 import InceptionV3
  model = InceptionV3((100, 221, 7), max_batch=2)
  model.init_random(0)
  with pytest.raises(ValueError, match='input shape'):
    model(torch.zeros((1, 100, 199, 7), dtype=torch.uint8, device='cuda'))


def test_large_batch_addresses_past_4gib():
  """3000 examples put the stem tensors past 2^32 bytes: per-wave relative
  addressing must give the same answers as a small batch of the same images."""
  from deepvariant_amd.inception_v3 import InceptionV3
  model = InceptionV3((100, 221, 7), max_batch=3000)
  model.init_random(seed=7)
  rng = np.random.default_rng(5)
  base = torch.from_numpy(rng.integers(0, 256, (8, 100, 221, 7), dtype=np.uint8)).cuda()
  big = base.repeat(375, 1, 1, 1)                  # 3000 examples, known period 8
  got = model(big).cpu().reshape(375, 8, 3)
  small = InceptionV3((100, 221, 7), max_batch=8)
  small.init_random(seed=7)
  want = small(base).cpu()
  assert (got - want[None]).abs().max().item() <= 1e-6


def test_graph_replay_on_a_side_stream_matches_eager():
  """On a non-default stream dv_model_infer captures the forward once and replays it as a
  hipGraph; on the legacy default stream it launches eagerly.  Same bytes either way."""
  from deepvariant_amd.inception_v3 import InceptionV3
  model = InceptionV3((100, 221, 7), max_batch=16)
  model.init_random(seed=3)
  rng = np.random.default_rng(9)
  x1 = torch.from_numpy(rng.integers(0, 256, (16, 100, 221, 7), dtype=np.uint8)).cuda()
  eager = model(x1).clone()
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    first = model(x1).clone()       # capture + first launch
    x1.copy_(torch.from_numpy(rng.integers(0, 256, (16, 100, 221, 7), dtype=np.uint8)))
    second = model(x1).clone()      # replay of the same graph on new pixels
  s.synchronize()
  eager2 = model(x1).clone()
  torch.cuda.synchronize()
  assert torch.equal(first.cpu(), eager.cpu())
  assert torch.equal(second.cpu(), eager2.cpu())
  assert not torch.equal(first.cpu(), second.cpu())


def test_preprocess_known_answer_all_byte_values():
  """deepvariant/dv_utils_test.py:177-200: preprocess_images([0, 128, 255]) ==
  [-1, 0, 0.9921875] exactly.  Both device paths are checked on all 256 byte values:
  the staging kernel (inputs with > 8 channels; buffer 0 is the preprocessed image) and
  the conversion fused into conv_first_u8_kernel (a one-hot 3x3 filter copies the centre
  pixel's channel 0 to output channel 0)."""
  from deepvariant_amd.inception_v3 import InceptionV3
  want = ((np.arange(256, dtype=np.float32) - 128.0) / 128.0)
  assert want[0] == -1.0 and want[128] == 0.0 and want[255] == 0.9921875
  # (a) staging kernel (round 4: inputs with 9..16 channels read the uint8 image directly too --
  # part (d); DV_NO_U8_CONV1_WIDE keeps the staging path for them)
  os.environ['DV_NO_U8_CONV1_WIDE'] = '1'
  try:
    model = InceptionV3((75, 75, 9), max_batch=1)
  finally:
    os.environ.pop('DV_NO_U8_CONV1_WIDE', None)
  model.init_random(seed=1)
  x = np.zeros((1, 75, 75, 9), np.uint8)
  x[0, 0, :, 0] = np.arange(75)
  x[0, 1, :, 0] = np.arange(75, 150)
  x[0, 2, :, 0] = np.arange(150, 225)
  x[0, 3, :31, 0] = np.arange(225, 256)
  model(torch.from_numpy(x).cuda())
  got = model.debug_tensor(0, 1).astype(np.float32)[0, :, :, 0]
  flat = np.concatenate([got[0], got[1], got[2], got[3, :31]])
  np.testing.assert_array_equal(flat, want)
  # (b) fused uint8 first conv: relu((x-128)/128 * 1 + 2) - 2 recovers every value exactly
  from oracle import inception_ref as R
  ref = R.make_random_model(1, seed=2)
  flat_w = ref.export_flat()
  k = np.zeros((3, 3, 1, 32), np.float32)
  k[1, 1, 0, 0] = 1.0
  n_w = k.size
  # BN with beta=2, mean=0, var=1-eps: scale 1, shift 2
  flat_w[:n_w] = k.reshape(-1)
  flat_w[n_w:n_w + 32] = 2.0
  flat_w[n_w + 32:n_w + 64] = 0.0
  flat_w[n_w + 64:n_w + 96] = 1.0 - 1e-3
  model1 = _per_layer_model((75, 75, 1), 1)
  model1.load_flat_weights(flat_w)
  x1 = np.zeros((1, 75, 75, 1), np.uint8)
  vals = np.arange(256, dtype=np.uint8)
  # centre pixels of the stride-2 windows: (2*oh+1, 2*ow+1)
  for i, v in enumerate(vals):
    x1[0, 2 * (i // 37) + 1, 2 * (i % 37) + 1, 0] = v
  model1(torch.from_numpy(x1).cuda())
  out = model1.debug_tensor(1, 1).astype(np.float32)[0, :, :, 0]
  got1 = np.array([out[i // 37, i % 37] for i in range(256)]) - 2.0
  np.testing.assert_allclose(got1, want, atol=2e-3)   # fp16 output grid near 2.0 is 2^-9
  assert got1[128] == 0.0 and got1[0] == -1.0
  # (c) the conversion inside the fused stem kernel (stem.hip): the second conv copies the
  # first conv's channel 0 (one-hot centre tap, BN shift 0), so conv2[oy, ox, 0] =
  # conv1[oy+1, ox+1, 0] = (x - 128)/128 + 2 on the 2^-7 grid -- exact in fp16
  k2 = np.zeros((3, 3, 32, 32), np.float32)
  k2[1, 1, 0, 0] = 1.0
  o2 = n_w + 96
  flat_w[o2:o2 + k2.size] = k2.reshape(-1)
  flat_w[o2 + k2.size:o2 + k2.size + 32] = 0.0           # beta
  flat_w[o2 + k2.size + 32:o2 + k2.size + 64] = 0.0      # mean
  flat_w[o2 + k2.size + 64:o2 + k2.size + 96] = 1.0 - 1e-3
  model2 = InceptionV3((75, 75, 1), max_batch=1)
  model2.load_flat_weights(flat_w)
  model2(torch.from_numpy(x1).cuda())
  out2 = model2.debug_tensor(2, 1).astype(np.float32)[0, :, :, 0]
  halo = (out2.shape[0] - 35) // 2
  got2 = np.array([out2[halo + i // 37 - 1, halo + i % 37 - 1]
                   for i in range(256) if i // 37 >= 1 and 1 <= i % 37 <= 35]) - 2.0
  want2 = np.array([want[i] for i in range(256) if i // 37 >= 1 and 1 <= i % 37 <= 35])
  np.testing.assert_array_equal(got2, want2)
  # (d) the wide uint8 first conv (9..16 channels: one tap x 16 channels per K chunk): channel 8 sits in
  # the SECOND k-group (bytes 8..15 of the pixel), channel 0 in the first; a one-hot filter on either
  # recovers every byte value
  for ch in (0, 8):
    ref9 = R.make_random_model(9, seed=2)
    flat9 = ref9.export_flat()
    k9 = np.zeros((3, 3, 9, 32), np.float32)
    k9[1, 1, ch, 0] = 1.0
    flat9[:k9.size] = k9.reshape(-1)
    flat9[k9.size:k9.size + 32] = 2.0
    flat9[k9.size + 32:k9.size + 64] = 0.0
    flat9[k9.size + 64:k9.size + 96] = 1.0 - 1e-3
    model9 = _per_layer_model((75, 75, 9), 1)
    model9.load_flat_weights(flat9)
    x9 = np.random.default_rng(ch).integers(0, 256, (1, 75, 75, 9), dtype=np.uint8)
    for i, v in enumerate(vals):
      x9[0, 2 * (i // 37) + 1, 2 * (i % 37) + 1, ch] = v
    model9(torch.from_numpy(x9).cuda())
    out9 = model9.debug_tensor(1, 1).astype(np.float32)[0, :, :, 0]
    got9 = np.array([out9[i // 37, i % 37] for i in range(256)]) - 2.0
    np.testing.assert_allclose(got9, want, atol=2e-3)
    assert got9[128] == 0.0 and got9[0] == -1.0

def test_conv_macs_match_the_architecture_count():
  """dv_model_conv_macs (what bench.py prices the conv kernels with) equals the count of
  the fp32 restatement: 1.005 GMAC at 100x221x7 (2.011 GFLOP per example)."""
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  for shape in ((100, 221, 7), (100, 147, 10)):
    model = InceptionV3(shape, max_batch=1)
    assert model.conv_macs_per_example == R.macs_per_example(shape[2], shape[0], shape[1]) - 2048 * 3
  assert abs(2 * InceptionV3((100, 221, 7), max_batch=1).conv_macs_per_example - 2.0105e9) < 1e6


@pytest.mark.parametrize('shape,n', [((100, 221, 7), 70), ((100, 147, 10), 9), ((140, 221, 7), 5)])
def test_row_band_mode_is_bit_identical_to_the_full_filters(shape, n):
  """Filters taller than the map (7x1 on the 4-row maps, 3x3 / 3x1 on the 1-row maps) skip
  the tap rows that only meet the zero halo (ConvArgs::band).  Dropping exact-zero products
  from an fp32 accumulation changes nothing: logits must be IDENTICAL with DV_NO_BAND.
  (140 rows: the last maps are 3 rows high -- 3x3 no longer qualifies, 7x1 on 6 rows neither.)"""
  import os
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=11)
  flat = ref.export_flat()
  x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
  banded = InceptionV3(shape, max_batch=64)
  banded.load_flat_weights(flat)
  os.environ['DV_NO_BAND'] = '1'
  try:
    full = InceptionV3(shape, max_batch=64)
  finally:
    del os.environ['DV_NO_BAND']
  full.load_flat_weights(flat)
  pa, pb = banded(x).clone(), full(x).clone()
  a, b = banded.debug_tensor(-1, n), full.debug_tensor(-1, n)   # the head's input: [n, h, w, 2048]
  assert np.array_equal(a, b)
  assert torch.equal(pa, pb)
  assert np.abs(a.astype(np.float32)).max() > 0


def test_second_stem_pool_fused_into_mixed0_is_bit_identical():
  """The stem's second max-pool runs inside mixed0's grouped 1x1 launch (max of the nine
  pieces in registers -- exact) instead of as its own kernel (DV_NO_POOL2_FUSE)."""
  import os
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  shape, n = (100, 221, 7), 37
  flat = R.make_random_model(7, seed=13).export_flat()
  x = torch.from_numpy(np.random.default_rng(6).integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
  fused = InceptionV3(shape, max_batch=64)
  fused.load_flat_weights(flat)
  os.environ['DV_NO_POOL2_FUSE'] = '1'
  try:
    plain = InceptionV3(shape, max_batch=64)
  finally:
    del os.environ['DV_NO_POOL2_FUSE']
  plain.load_flat_weights(flat)
  pa, pb = fused(x).clone(), plain(x).clone()
  assert np.array_equal(fused.debug_tensor(-1, n), plain.debug_tensor(-1, n))
  assert torch.equal(pa, pb)


def test_fresh_image_tensors_replay_the_captured_graph():
  """The hipGraph cache is keyed by (n, stream): the caller's image / output pointers go through a
  device-side table, so a new tensor per call (what a per-region driver does) must not recapture,
  and every call must classify ITS images."""
  import torch
  from deepvariant_amd.inception_v3 import InceptionV3
  shape = (100, 221, 7)
  m = InceptionV3(shape, max_batch=32)
  m.init_random(3)
  rng = np.random.default_rng(5)
  stream = torch.cuda.Stream()
  keep, outs = [], []
  with torch.cuda.stream(stream):
    for i in range(6):
      x = torch.from_numpy(rng.integers(0, 256, (24,) + shape, dtype=np.uint8)).cuda()
      keep.append(x)                       # keep every tensor alive: six distinct device pointers
      outs.append(m(x).cpu().numpy())
    again = [m(x).cpu().numpy() for x in keep]
  assert len({x.data_ptr() for x in keep}) == 6
  captures, replays = m.graph_stats()
  assert captures == 1 and replays == 11
  for a, b in zip(outs, again):
    np.testing.assert_array_equal(a, b)
  assert max(np.abs(outs[0] - o).max() for o in outs[1:]) > 0      # different images, different answers
