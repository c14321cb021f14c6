"""HIP encoder vs the reference's known-answer vectors, through the C ABI.

Same vectors as tests/test_oracle_known_answers.py, but `make` is the product:
deepvariant_amd.pileup_image_native.PileupImageEncoderNative ->
packing -> dv_encode_batch (libdvhip.so) on the GPU.
"""
import numpy as np
import pytest

from tests import known_answers as KA
from deepvariant_amd import dv_types as T

pytestmark = pytest.mark.gpu


def make(options):
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  return PileupImageEncoderNative(options)


def test_reference_encoding():
  KA.check_reference_encoding(make)


def test_encode_read_matches():
  KA.check_encode_read_matches(make)


@pytest.mark.parametrize('hp_value,hp_color,polishing', KA.HP_CASES)
def test_encode_read_hp_channel(hp_value, hp_color, polishing):
  KA.check_encode_read_hp_channel(make, hp_value, hp_color, polishing)


def test_encode_read_allele_frequency():
  KA.check_encode_read_allele_frequency(make)


@pytest.mark.parametrize('s,e', KA.SPANS2_CASES)
def test_encode_read_spans2(s, e):
  KA.check_encode_read_spans2(make, s, e)


def test_encode_read_deletion():
  KA.check_encode_read_deletion(make)


def test_encode_read_insertion():
  KA.check_encode_read_insertion(make)


@pytest.mark.parametrize('bq,mq', KA.QUAL_GRID[::3])
def test_quality_gates(bq, mq):
  KA.check_ignores_low_quality_bases(make, bq, mq)
  KA.check_keeps_low_quality_bases(make, bq, mq)
  KA.check_ignores_low_mapping_quality(make, bq, mq)


@pytest.mark.parametrize('case', KA.READ_SUPPORT_CASES)
def test_read_support_is_respected(case):
  KA.check_read_support_is_respected(make, *case)


@pytest.mark.parametrize('case', KA.MULTIALLELIC_CASES)
def test_read_support_multiallelic(case):
  KA.check_read_support_multiallelic(make, *case)


@pytest.mark.parametrize('case', KA.CUSTOM_CHANNEL_CASES)
def test_custom_channels(case):
  KA.check_custom_channel(make, *case)


@pytest.mark.parametrize('case', KA.FUZZY_CASES)
def test_read_supports_variant_fuzzy(case):
  KA.check_fuzzy_channel(make, case)


def test_custom_multi():
  KA.check_custom_multi(make)


@pytest.mark.parametrize('name', sorted(KA.BUILD_PILEUP_CASES))
def test_build_pileup(name):
  KA.check_build_pileup_case(make, name)


def test_unsupported_channel_fails_loudly():
  from deepvariant_amd import _lib
  with pytest.raises(_lib.DvError) as e:
    # four per-base host-computed channels: one more than dv_batch has planes for (include/dvhip.h, ABI v6)
    make(KA.default_options(['is_homopolymer', 'homopolymer_weighted', 'homopolymer_insertion_quality',
                             'homopolymer_deletion_quality'])).encode_reference('ACGTA')
  assert e.value.status == _lib.DV_ERR_UNSUPPORTED


def test_sequence_context_channels_reference_rows():
  """gc_content / is_homopolymer / homopolymer_weighted reference rows
  (channels/*_channel.cc FillRefBase; vectors of pileup_channel_lib_test.cc:487-560:
  ATCGGGAG -> 00011100, ATCGGGAA -> 11133322)."""
  from oracle import oracle as O
  opts = KA.default_options(['gc_content', 'is_homopolymer', 'homopolymer_weighted'])
  ref = 'ATCGGGAAT'
  got = make(opts).encode_reference(ref)
  np.testing.assert_array_equal(got, O.encode_reference(opts, ref))
  assert got[0, :, 1].tolist() == [0, 0, 0, 254, 254, 254, 0, 0, 0]
  assert got[0, :, 2].tolist() == [8, 8, 8, 25, 25, 25, 16, 16, 8]
  assert set(got[0, :, 0].tolist()) == {int(254.0 * (44 / 100.0))}   # 4 of 9 bases are G/C


def test_unknown_cigar_op_is_an_error():
  o = KA.default_options()
  read = T.make_read('AAA', start=1, cigar='3M', quals=[30] * 3)
  read.alignment.cigar[0].operation = 0
  with pytest.raises(ValueError, match='CIGAR'):
    make(o).encode_read(KA.make_dv_call(), 'AACAG', read, 1, ['C'])
