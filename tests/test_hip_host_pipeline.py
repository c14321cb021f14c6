"""Host-inclusive pipeline (deepvariant_amd/host_pipeline.py): native region packing on a
worker thread, pinned-memory upload on a copy stream, encode + classify on the GPU -- against
the plain path (Python-packed batch, one blocking upload), bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_equals_direct_path():
  from deepvariant_amd import host_pipeline as hp
  from deepvariant_amd import packing, synth
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.inception_v3 import InceptionV3
  from deepvariant_amd.pileup_image_native import _Encoder
  from oracle import oracle as O
  opts = synth.illumina_options(7)
  H, W, C = opts.height, opts.width, 7
  base = synth.make_illumina_batch(300, seed=11, options=opts)
  table, cands, combos, windows = synth.region_inputs_from_batch(base, opts)
  dev = torch.device('cuda', 0)
  enc = _Encoder(opts, W)
  model = InceptionV3((H, W, C), max_batch=512)
  model.init_random(seed=5)
  inputs = hp.RegionInputs(table, cands, combos, windows, W, opts.read_overlap_buffer_bp, H, H * W * C)
  pipe = hp.HostPipeline(inputs, enc, model, C, dev, (H, W, C))
  probs = pipe.run(5)                       # both slots reused
  n = pipe.slots[0].n_items
  assert n == base.n_items and probs.shape == (n, 3)
  got_images = pipe.slots[0].images[:n].cpu().numpy()
  # direct path: the Python wrapper of the same native packer -> PackedBatch -> DeviceBatch
  batch, plan = packing.pack_region_native(table, cands, combos, windows, W,
                                           opts.read_overlap_buffer_bp, H, H * W * C)
  assert batch.n_items == n
  images = torch.empty((n, H, W, C), dtype=torch.uint8, device=dev)
  DeviceBatch(batch, dev).encode(enc, C, images)
  np.testing.assert_array_equal(got_images, images.cpu().numpy())
  assert torch.equal(probs.cpu(), model(images).cpu())
  # and the images are the oracle's for the same packed lists
  want, _ = O.encode_packed(opts, batch, C)
  np.testing.assert_array_equal(got_images.reshape(-1), want.reshape(-1))
