"""The host side of the three Ultima flow-space channels (homopolymer_insertion_quality, homopolymer_deletion_quality,
inter_homopolymer_insertion_quality; include/dvhip.h ABI v6): the per-base pixels dv_flow_channel_pixels computes from
the tp / t0 tags are the bytes the reference draws (deepvariant/channels/homopolymer_indel_quality_channel.cc:123-183,
inter_homopolymer_insertion_quality_channel.cc:76-125) -- checked against the oracle restatement and, where it is
built, against the reference's own channel classes (oracle/_ref/libdvref.so) -- and they land in the base_aux plane
the device encoder reads for that channel (dv_base_aux_plane).  The device half is tests/test_hip_flow_channels.py."""
import ctypes as C

import numpy as np
import pytest

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import packing
from oracle import oracle as O
from tests import fuzz_inputs as FZ

FLOW = ['homopolymer_insertion_quality', 'homopolymer_deletion_quality', 'inter_homopolymer_insertion_quality']


def _plane(channels, index):
  arr = (C.c_int32 * len(channels))(*channels)
  return _lib.lib().dv_base_aux_plane(arr, len(channels), index)


def test_plane_assignment_rule():
  # is_homopolymer -> 0, homopolymer_weighted -> 1, wherever they stand; flow channels take 2, 1, 0 in list order
  assert packing.seq_aux_planes([1, 2, 3]) == ()
  assert packing.seq_aux_planes([1, 16, 17]) == (16, 17, 0)
  assert packing.seq_aux_planes([17, 1, 16]) == (16, 17, 0)
  assert packing.seq_aux_planes([1, 28, 29, 30]) == (30, 29, 28)
  assert packing.seq_aux_planes([30, 1, 28]) == (0, 28, 30)
  assert packing.seq_aux_planes([29, 16, 2, 30]) == (16, 30, 29)
  assert packing.seq_aux_planes([28, 17, 16]) == (16, 17, 28)
  assert _plane([1, 28, 2], 0) == 3 and _plane([1, 28, 2], 1) == 2          # DV_BASE_AUX_NONE for the others
  assert _plane([16, 17, 28, 29], 3) == _lib.DV_ERR_UNSUPPORTED              # four per-base channels: one too many
  assert 'per-base' in _lib.last_error()
  with pytest.raises(Exception):
    packing.seq_aux_planes([16, 17, 28, 29])
  assert _plane([1, 2], 5) == _lib.DV_ERR_INVALID_ARGUMENT


def _reads(seed, n=60):
  rng = np.random.default_rng(seed)
  _, _, reads, _, _ = FZ.make_case(rng, 71, n, with_ultima=True)
  return reads


def _drawn_by(backend_reference, channel, read):
  """What the checker draws for the read's bases: a one-read window of plain matches, every quality accepted."""
  n = len(read.aligned_sequence)
  width = n | 1
  if width < 3:
    width = 3
  opts = FZ.options(['read_base', channel], width, 4, min_bq=0, min_mapq=0)
  r = T.Read(fragment_name='r', read_number=0, aligned_sequence=read.aligned_sequence,
             aligned_quality=read.aligned_quality,
             alignment=T.LinearAlignment(position=T.Position('chr1', 100, False), mapping_quality=60,
                                         cigar=[T.CigarUnit(1, n)]))
  r.info = dict(read.info)
  call = T.DeepVariantCall(variant=T.Variant('chr1', 100, 101, 'A', ['C']), allele_support={})
  if backend_reference:
    with O.reference_backend():
      row = O.encode_read(opts, call, 'A' * width, r, 100, ['C'], None)
  else:
    row = O.encode_read(opts, call, 'A' * width, r, 100, ['C'], None)
  assert row is not None
  return row[0, :n, 1]


@pytest.mark.parametrize('against_reference', [False, True], ids=['oracle', 'reference_build'])
def test_pixels_are_what_the_reference_draws(against_reference):
  if against_reference and not O.reference_available():
    pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here')
  reads = _reads(11)
  table = packing.ReadTable.from_reads(reads, need_seq_aux=packing.seq_aux_planes([1, 28, 29, 30]))
  assert table.base_aux0 is not None and table.base_aux1 is not None and table.base_aux2 is not None
  seen = set()
  for i, r in enumerate(reads):
    s0, s1 = int(table.read_seq_off[i]), int(table.read_seq_off[i + 1])
    for plane, name in ((table.base_aux2, FLOW[0]), (table.base_aux1, FLOW[1]), (table.base_aux0, FLOW[2])):
      want = _drawn_by(against_reference, name, r)
      assert np.array_equal(plane[s0:s1], want), (i, name, r.aligned_sequence, plane[s0:s1].tolist(), want.tolist())
      seen.update(want.tolist())
  assert 255 in seen and 0 in seen and len(seen) > 20      # (the scale's top is 255 here, not 254)


def test_reads_without_tags_and_empty_tables():
  r = T.Read(fragment_name='x', aligned_sequence='AACCCGT', aligned_quality=bytes([30] * 7),
             alignment=T.LinearAlignment(position=T.Position('chr1', 5, False), mapping_quality=50,
                                         cigar=[T.CigarUnit(1, 7)]))
  t = packing.ReadTable.from_reads([r], need_seq_aux=(28, 30, 29))
  assert t.base_aux0.tolist() == [255] * 7 and t.base_aux2.tolist() == [255] * 7    # no tp: nothing to subtract
  assert t.base_aux1.tolist() == [0] * 7                                            # no t0: zeros
  e = packing.ReadTable.from_reads([], need_seq_aux=(28, 30, 29))
  assert e.base_aux0.size == 0 and e.base_aux2.size == 0
  # one error of each direction in the CCC run: insertion / deletion qualities differ, the other runs keep the top
  r = T.Read(fragment_name='y', aligned_sequence='AACCCGT', aligned_quality=bytes([30, 30, 20, 30, 10, 30, 30]),
             alignment=T.LinearAlignment(position=T.Position('chr1', 5, False), mapping_quality=50,
                                         cigar=[T.CigarUnit(1, 7)]))
  r.info['tp'] = T.ListValue(values=[T.Value(int_value=v) for v in (0, 0, 1, 0, -1, 0, 0)])
  t = packing.ReadTable.from_reads([r], need_seq_aux=(28, 0, 29))
  ins, dele = t.base_aux0.tolist(), t.base_aux2.tolist()
  assert ins == [255, 255] + [int(np.float32(255.0) * 20 / np.float32(93.0))] * 3 + [255, 255]
  assert dele[2:5] == [int(np.float32(255.0) * 10 / np.float32(93.0))] * 3 and dele[:2] == [255, 255]
  lib = _lib.lib()
  assert lib.dv_flow_channel_pixels(16, None, None, None, None, 0, None) == _lib.DV_ERR_INVALID_ARGUMENT


def test_rows_and_concatenation_keep_the_third_plane():
  reads = _reads(5, n=12)
  planes = packing.seq_aux_planes([28, 29, 30])
  t = packing.ReadTable.from_reads(reads, need_seq_aux=planes)
  rows = np.array([7, 2, 2, 11], np.int64)
  sub = t.take(rows)
  direct = packing.ReadTable.from_reads([reads[i] for i in rows.tolist()], need_seq_aux=planes)
  for name in ('base_aux0', 'base_aux1', 'base_aux2'):
    assert np.array_equal(getattr(sub, name), getattr(direct, name)), name
