"""The three Ultima flow-space channels on the device (ABI v6: dv_batch::base_aux2, dv_base_aux_plane): pile-ups
drawn by the HIP encoder from host-computed per-base planes equal the oracle's and -- where it is built -- the
REFERENCE's own (channels/homopolymer_{insertion,deletion}_quality_channel.cc,
channels/inter_homopolymer_insertion_quality_channel.cc compiled unmodified into oracle/_ref/libdvref.so), byte for
byte, through both entries of dv_encode_batch (host batch staged by the library; device-resident batch)."""
import zlib

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from tests import fuzz_inputs as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,channels,width,height,okw,ckw', F.ULTIMA_CONFIGS, ids=[c[0] for c in F.ULTIMA_CONFIGS])
def test_flow_channels_equal_the_oracle_and_the_reference_build(name, channels, width, height, okw, ckw):
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  rng = np.random.default_rng(zlib.crc32(name.encode()))
  opts = F.options(channels, width, height, **dict(okw))
  enc = PileupImageEncoderNative(opts)
  so = T.SampleOptions(pileup_height=height)
  for trial in range(10):
    n_reads = int(rng.choice([0, 1, 3, 10, height, height + 30, 3 * height]))
    call, ref, reads, start, combo = F.make_case(rng, width, n_reads, **dict(ckw))
    blank = [int(T.CHANNEL_STR_TO_ENUM[channels[int(rng.integers(0, len(channels)))]])] if trial % 3 == 0 else None
    got = enc.build_pileup_for_one_sample(call, ref, reads, start, combo, so, channels_to_blank=blank)
    want = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height, channels_to_blank=blank)
    np.testing.assert_array_equal(got, want, err_msg='%s trial %d (oracle)' % (name, trial))
    if O.reference_available():
      with O.reference_backend():
        ref_img = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height, channels_to_blank=blank)
      np.testing.assert_array_equal(got, ref_img, err_msg='%s trial %d (reference build)' % (name, trial))


def test_device_resident_batch_carries_the_third_plane():
  import torch
  from deepvariant_amd import packing
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative, _Encoder
  name, channels, width, height, okw, ckw = F.ULTIMA_CONFIGS[0]
  opts = F.options(channels, width, height, **dict(okw))
  api = PileupImageEncoderNative(opts)
  rng = np.random.default_rng(9)
  call, ref, reads, start, combo = F.make_case(rng, width, 25, **dict(ckw))
  table = packing.ReadTable.from_reads(reads, need_seq_aux=api._need_seq_aux)      # pylint: disable=protected-access
  assert table.base_aux2 is not None
  batch = packing.PackedBatch(table=table, width=width)
  idx = np.arange(len(reads), dtype=np.uint32)
  batch.add_item(variant_start=call.variant.start, image_start=start, ref_idx=batch.add_ref_window(ref), read_idx=idx,
                 codes=packing.support_codes(call, combo, table, idx), height=height, out_off=0)
  enc = _Encoder(opts, width)
  host_img, _ = enc.encode(batch, len(channels))
  dev = torch.device('cuda:0')
  out = torch.zeros(height * width * len(channels), dtype=torch.uint8, device=dev)
  DeviceBatch(batch, dev).encode(enc, len(channels), out)
  torch.cuda.synchronize()
  assert np.array_equal(out.cpu().numpy(), host_img.reshape(-1)[:out.numel()])
  got = api.build_pileup_for_one_sample(call, ref, reads, start, combo, T.SampleOptions(pileup_height=height))
  assert np.array_equal(host_img.reshape(-1)[:got.size], got.reshape(-1))
