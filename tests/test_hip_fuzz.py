"""Randomised parity through the reference-shaped API (GPU): every CIGAR op,
partial window overlap, HP sorting/channel, methylation channels, pile-ups
deeper than the image, blanked channels, mean coverage -- HIP vs oracle,
bit-exact."""
import zlib

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from tests import fuzz_inputs as F
from tests import known_answers as KA

pytestmark = pytest.mark.gpu


CONFIGS = F.CONFIGS
_options = F.options


@pytest.mark.parametrize('name,channels,width,height,okw,ckw', CONFIGS)
def test_fuzz_build_pileup(name, channels, width, height, okw, ckw):
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  rng = np.random.default_rng(zlib.crc32(name.encode()))   # (str hashes differ per process)
  opts = _options(channels, width, height, **dict(okw))
  enc = PileupImageEncoderNative(opts)
  so = T.SampleOptions(pileup_height=height)
  for trial in range(12):
    n_reads = int(rng.choice([0, 1, 3, 10, height, height + 30, 3 * height]))
    call, ref, reads, start, combo = F.make_case(rng, width, n_reads, **dict(ckw))
    if 'avg_base_quality' in channels:  # reference aborts on quals > 93
      for r in reads:
        r.aligned_quality = bytes(min(q, 93) for q in r.aligned_quality)
    mc = float(rng.integers(0, height + 5)) if 'mean_coverage' in channels else 0.0
    blank = [int(T.CHANNEL_STR_TO_ENUM[channels[int(rng.integers(0, len(channels)))]])] \
        if trial % 3 == 0 else None
    got = enc.build_pileup_for_one_sample(call, ref, reads, start, combo, so,
                                          mean_coverage=mc, channels_to_blank=blank)
    want = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height,
                          mean_coverage=mc, channels_to_blank=blank)
    np.testing.assert_array_equal(got, want, err_msg='%s trial %d' % (name, trial))


@pytest.mark.parametrize('name,channels,width,height,okw,ckw', CONFIGS)
def test_fuzz_build_pileup_against_the_reference_build(name, channels, width, height, okw, ckw):
  """The same shapes against the REFERENCE's own encoder (oracle/_ref/libdvref.so: its pileup_image_native.cc /
  pileup_channel_lib.cc / channels/*.cc compiled unmodified, oracle/ref_build/) instead of the restatement: the HIP
  kernel's pixels equal the reference's, with nothing of ours in between but the input marshalling."""
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  if not O.reference_available():
    pytest.skip('oracle/_ref/libdvref.so was not built (no reference tree where build() ran)')
  rng = np.random.default_rng(zlib.crc32(name.encode()) ^ 0x5bd1e995)
  opts = _options(channels, width, height, **dict(okw))
  enc = PileupImageEncoderNative(opts)
  so = T.SampleOptions(pileup_height=height)
  for trial in range(8):
    n_reads = int(rng.choice([0, 2, 10, height, height + 30, 3 * height]))
    call, ref, reads, start, combo = F.make_case(rng, width, n_reads, **dict(ckw))
    if 'avg_base_quality' in channels:  # reference aborts on quals > 93
      for r in reads:
        r.aligned_quality = bytes(min(q, 93) for q in r.aligned_quality)
    mc = float(rng.integers(0, height + 5)) if 'mean_coverage' in channels else 0.0
    blank = [int(T.CHANNEL_STR_TO_ENUM[channels[int(rng.integers(0, len(channels)))]])] \
        if trial % 3 == 0 else None
    got = enc.build_pileup_for_one_sample(call, ref, reads, start, combo, so,
                                          mean_coverage=mc, channels_to_blank=blank)
    with O.reference_backend():
      want = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height,
                            mean_coverage=mc, channels_to_blank=blank)
    np.testing.assert_array_equal(got, want, err_msg='%s trial %d' % (name, trial))


def test_empty_batch_and_bad_arguments():
  import ctypes as C
  from deepvariant_amd import _lib, packing, synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  enc = _Encoder(opts, opts.width)
  batch = synth.make_illumina_batch(4, seed=1, options=opts)
  empty = packing.PackedBatch(table=batch.table, width=opts.width)
  out, rows = enc.encode(empty, 7)
  assert out.size == 0 and rows.size == 0
  # out_channels smaller than the channel list is an argument error
  b, keep = batch.to_ctypes()
  buf = np.zeros(batch.out_bytes(7), np.uint8)
  rc = _lib.lib().dv_encode_batch(enc.handle, C.byref(b), 3, buf.ctypes.data, None,
                                  _lib.DV_MEM_HOST, None)
  assert rc == _lib.DV_ERR_INVALID_ARGUMENT
  # a list entry pointing past the read table is rejected, not dereferenced
  bad = synth.make_illumina_batch(4, seed=1, options=opts)
  bad.list_read_chunks[0] = bad.list_read_chunks[0].copy()
  bad.list_read_chunks[0][0] = 10 ** 7
  bad._frozen = None
  with pytest.raises(_lib.DvError):
    enc.encode(bad, 7)


def test_maximum_pileup_height_and_depth():
  """The tallest image the encoder accepts (reference band + 256 read rows), with more
  reads than rows (shuffle-and-truncate over 700 reads), and one row too many."""
  from deepvariant_amd import _lib
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  rng = np.random.default_rng(99)
  width, height = 61, 5 + 256
  opts = _options(T.PILEUP_CHANNELS_WITH_INSERT_SIZE, width, height)
  enc = PileupImageEncoderNative(opts)
  for n_reads in (256, 257, 700):
    call, ref, reads, start, combo = F.make_case(rng, width, n_reads)
    got = enc.build_pileup_for_one_sample(call, ref, reads, start, combo,
                                          T.SampleOptions(pileup_height=height))
    want = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height)
    np.testing.assert_array_equal(got, want)
  with pytest.raises(_lib.DvError, match='item_height'):
    enc.build_pileup_for_one_sample(call, ref, reads, start, combo,
                                    T.SampleOptions(pileup_height=height + 1))
