"""Known-answer vectors lifted from the reference's own tests.

Each `check_*` function takes an encoder *factory* `make(options)` returning an
object with the reference's pybind interface
(`deepvariant/python/pileup_image_native_pybind.cc:82-129`):
  .encode_reference(ref_bases)                                 -> u8[1,W,C]
  .encode_read(dv_call, ref_bases, read, image_start_pos, alts) -> u8[1,W,C]|None
  .build_pileup_for_one_sample(dv_call, ref_bases, reads, image_start_pos,
                               alt_alleles, sample_options)    -> u8[H,W,C]
so that the same vectors pin (a) the CPU oracle and (b) the HIP product path.

Sources (file:line in /root/reference):
  deepvariant/pileup_image_test.py:138-785
  deepvariant/pileup_image_native_test.cc:277-413
  deepvariant/pileup_channel_lib_test.cc:74-851
"""
import itertools

import numpy as np

from deepvariant_amd import dv_types as T


def _supporting_reads(*names):
  return T.SupportingReads(read_names=list(names))


def make_dv_call(ref_bases='A', alt_bases='C'):
  # pileup_image_test.py:52-63
  return T.DeepVariantCall(
      variant=T.Variant(reference_name='chr1', start=10, end=11,
                        reference_bases=ref_bases, alternate_bases=[alt_bases]),
      allele_support={'C': _supporting_reads('read1/1', 'read2/1')})


def make_dv_call_with_allele_frequency(alt_frequency=0.1):
  # pileup_image_test.py:66-80
  call = make_dv_call()
  call.allele_frequency = {'A': 1 - alt_frequency, 'C': alt_frequency}
  return call


def default_options(channels=T.PILEUP_DEFAULT_CHANNELS, read_requirements=None,
                    **kwargs):
  # pileup_image_test.py:83-90 (_make_encoder)
  options = T.default_options(read_requirements)
  for ch in channels:
    options.channels.append(ch)
    options.num_channels += 1
  for k, v in kwargs.items():
    setattr(options, k, v)
  return options


def cc_options(width, height, ref_band_height, channels):
  # testing_utils.cc:141-164 (MakeDefaultPileupImageOptions): no
  # read_requirements, multi_allelic_mode etc. left at proto defaults.
  o = T.PileupImageOptions(
      reference_band_height=ref_band_height,
      base_color_offset_a_and_g=40, base_color_offset_t_and_c=30,
      base_color_stride=70, allele_supporting_read_alpha=1.0,
      allele_unsupporting_read_alpha=0.6,
      other_allele_supporting_read_alpha=0.6,
      reference_matching_read_alpha=0.2, reference_mismatching_read_alpha=1.0,
      indel_anchoring_base_char='*', reference_alpha=0.4,
      reference_base_quality=60, positive_strand_color=70,
      negative_strand_color=240, base_quality_cap=40, mapping_quality_cap=60,
      height=height, width=width, read_overlap_buffer_bp=5,
      random_seed=2101079370, min_non_zero_allele_frequency=0.00001)
  o.channels = list(channels)
  o.num_channels = len(channels)
  return o


FULL_EXPECTED = np.dstack([
    (250, 30, 30, 180, 100),   # base
    (63, 69, 76, 82, 88),      # base quality
    (211, 211, 211, 211, 211), # mapping quality
    (70, 70, 70, 70, 70),      # strand
    (254, 254, 254, 254, 254), # supports alt
    (50, 50, 254, 50, 50),     # matches ref
]).astype(np.uint8)


def check_reference_encoding(make):
  # pileup_image_test.py:138-155
  got = make(default_options()).encode_reference('ACGTN')
  exp = np.dstack([
      (250, 30, 180, 100, 0), (254,) * 5, (254,) * 5, (70,) * 5, (152,) * 5,
      (50,) * 5]).astype(np.uint8)
  np.testing.assert_equal(got, exp)


def check_encode_read_matches(make):
  # pileup_image_test.py:160-185
  dv_call = make_dv_call()
  read = T.make_read('ACCGT', start=10, cigar='5M', quals=range(10, 15),
                     name='read1')
  got = make(default_options()).encode_read(
      dv_call, 'ACAGT', read, 10, dv_call.variant.alternate_bases)
  np.testing.assert_equal(got, FULL_EXPECTED)


HP_CASES = [(None, 0, None), (0, 0, None), (1, 127, None), (2, 254, None),
            (None, 0, 2), (0, 0, 2), (1, 254, 2), (2, 127, 2)]


def check_encode_read_hp_channel(make, hp_value, hp_color, polishing):
  # pileup_image_test.py:187-239
  dv_call = make_dv_call_with_allele_frequency()
  read = T.make_read('ACCGT', start=10, cigar='5M', quals=range(10, 15),
                     name='read1')
  if hp_value is not None:
    read.info['HP'] = T.ListValue(values=[T.Value(int_value=hp_value)])
  opts = default_options(T.PILEUP_DEFAULT_CHANNELS + ['haplotype'])
  if polishing is not None:
    opts.hp_tag_for_assembly_polishing = polishing
  got = make(opts).encode_read(dv_call, 'ACAGT', read, 10,
                               dv_call.variant.alternate_bases)
  exp = np.concatenate(
      [FULL_EXPECTED, np.full((1, 5, 1), hp_color, np.uint8)], axis=2)
  np.testing.assert_equal(got, exp)


def check_encode_read_allele_frequency(make):
  # pileup_image_test.py:241-271
  dv_call = make_dv_call_with_allele_frequency()
  read = T.make_read('ACCGT', start=10, cigar='5M', quals=range(10, 15),
                     name='read1')
  opts = default_options(T.PILEUP_DEFAULT_CHANNELS + ['allele_frequency'])
  got = make(opts).encode_read(dv_call, 'ACAGT', read, 10,
                               dv_call.variant.alternate_bases)
  exp = np.concatenate(
      [FULL_EXPECTED, np.full((1, 5, 1), 203, np.uint8)], axis=2)
  np.testing.assert_equal(got, exp)


SPANS2_CASES = [(s, e) for s in range(0, 5) for e in range(6, 12)]


def check_encode_read_spans2(make, bases_start, bases_end):
  # pileup_image_test.py:274-326
  bases = 'AAAACCGTCCC'
  quals = [9, 9, 9, 10, 11, 12, 13, 14, 8, 8, 8]
  ref_start, ref_size = 10, 5
  read_bases = bases[bases_start:bases_end]
  read_quals = quals[bases_start:bases_end]
  read_start = 7 + bases_start
  expected = np.zeros((1, ref_size, 6), dtype=np.uint8)
  for i in range(read_start, read_start + len(read_bases)):
    if ref_start <= i < ref_start + ref_size:
      expected[0, i - ref_start] = FULL_EXPECTED[0, i - ref_start]
  read = T.make_read(read_bases, start=read_start,
                     cigar=str(len(read_bases)) + 'M', quals=read_quals,
                     name='read1')
  dv_call = make_dv_call()
  got = make(default_options()).encode_read(
      dv_call, 'ACAGT', read, ref_start, dv_call.variant.alternate_bases)
  np.testing.assert_equal(got, expected)


def check_encode_read_deletion(make):
  # pileup_image_test.py:328-354
  read = T.make_read('AAG', start=2, cigar='2M2D1M', quals=range(10, 13),
                     name='read1')
  dv_call = make_dv_call()
  exp = np.dstack([
      (250, 0, 0, 0, 180), (63, 69, 0, 0, 76), (211, 211, 0, 0, 211),
      (70, 70, 0, 0, 70), (254, 254, 0, 0, 254), (50, 254, 0, 0, 50),
  ]).astype(np.uint8)
  got = make(default_options()).encode_read(
      dv_call, 'AACAG', read, 2, dv_call.variant.alternate_bases)
  np.testing.assert_equal(got, exp)


def check_encode_read_insertion(make):
  # pileup_image_test.py:356-382
  read = T.make_read('AAACAG', start=2, cigar='2M1I3M', quals=range(10, 16),
                     name='read1')
  dv_call = make_dv_call()
  exp = np.dstack([
      (250, 0, 30, 250, 180), (63, 76, 82, 88, 95), (211,) * 5, (70,) * 5,
      (254,) * 5, (50, 254, 50, 50, 50),
  ]).astype(np.uint8)
  got = make(default_options()).encode_read(
      dv_call, 'AACAG', read, 2, dv_call.variant.alternate_bases)
  np.testing.assert_equal(got, exp)


QUAL_GRID = list(itertools.product(range(0, 5), range(0, 5)))


def _low_qual_call():
  return T.DeepVariantCall(variant=T.Variant(
      reference_name='chr1', start=2, end=3, reference_bases='A',
      alternate_bases=['C']))


def check_ignores_low_quality_bases(make, min_bq, min_mq):
  # pileup_image_test.py:384-434
  rr = T.ReadRequirements(min_base_quality=min_bq, min_mapping_quality=min_mq,
                          min_base_quality_mode=1)
  pie = make(default_options(read_requirements=rr))
  for base_qual in range(min_bq + 5):
    read = T.make_read('AAA', start=1, cigar='3M',
                       quals=[min_bq, base_qual, min_bq], mapq=min_mq)
    actual = pie.encode_read(_low_qual_call(), 'AACAG', read, 1, ['C'])
    if base_qual < min_bq:
      assert actual is None
    else:
      assert actual is not None


def check_keeps_low_quality_bases(make, min_bq, min_mq):
  # pileup_image_test.py:436-487
  rr = T.ReadRequirements(min_base_quality=min_bq, min_mapping_quality=min_mq,
                          min_base_quality_mode=1)
  pie = make(default_options(read_requirements=rr))
  for base_qual in range(1, min_bq + 5):
    read = T.make_read('AAA', start=1, cigar='3M',
                       quals=[base_qual - 1, min_bq, base_qual + 1],
                       mapq=min_mq)
    assert pie.encode_read(_low_qual_call(), 'AACAG', read, 1,
                           ['C']) is not None


def check_ignores_low_mapping_quality(make, min_bq, min_mq):
  # pileup_image_test.py:489-540
  rr = T.ReadRequirements(min_base_quality=min_bq, min_mapping_quality=min_mq,
                          min_base_quality_mode=1)
  pie = make(default_options(read_requirements=rr))
  for mapping_qual in range(min_mq + 5):
    read = T.make_read('AAA', start=1, cigar='3M', quals=[min_bq] * 3,
                       mapq=mapping_qual)
    actual = pie.encode_read(_low_qual_call(), 'AACAG', read, 1, ['C'])
    if mapping_qual < min_mq:
      assert actual is None
    else:
      assert actual is not None


READ_SUPPORT_CASES = [
    ('read1', 1, 'C', 'C', True), ('read1', 2, 'C', 'C', False),
    ('read2', 1, 'C', 'G', False), ('read2', 2, 'C', 'G', False),
    ('read3', 1, 'C', 'C', False), ('read3', 2, 'C', 'C', True),
    ('read1', 1, 'G', 'C', False), ('read1', 2, 'G', 'C', False),
    ('read2', 1, 'G', 'G', True), ('read2', 2, 'G', 'G', True),
    ('read3', 1, 'G', 'C', False), ('read3', 2, 'G', 'C', False),
]


def check_read_support_is_respected(make, read_name, read_number, alt_allele,
                                    read_base, supports_alt):
  # pileup_image_test.py:542-604
  dv_call = T.DeepVariantCall(
      variant=T.Variant(reference_name='chr1', start=10, end=11,
                        reference_bases='A', alternate_bases=['C', 'G']),
      allele_support={'C': _supporting_reads('read1/1', 'read3/2'),
                      'G': _supporting_reads('read2/1', 'read2/2')})
  read = T.make_read(read_base, start=10, cigar='1M', quals=[50],
                     name=read_name)
  read.read_number = read_number
  actual = make(default_options()).encode_read(dv_call, 'TAT', read, 9,
                                               [alt_allele])
  expected = [{'C': 30, 'G': 180}[read_base], 254, 211, 70,
              [152, 254][supports_alt], 254]
  assert list(actual[0, 1]) == expected


MULTIALLELIC_CASES = [
    ('read1', 1, 'C', 'C', True, int(254.0 * 1.0)),
    ('read1', 2, 'C', 'C', True, int(254.0 * 0.6)),
    ('read2', 1, 'C', 'G', True, int(254.0 * 0.3)),
    ('read1', 1, 'C', 'C', False, int(254.0 * 1.0)),
    ('read1', 2, 'C', 'C', False, int(254.0 * 0.6)),
    ('read2', 1, 'C', 'G', False, int(254.0 * 0.6)),
]


def check_read_support_multiallelic(make, read_name, read_number, alt_allele,
                                    read_base, other_color, expected_color):
  # pileup_image_test.py:606-660
  dv_call = T.DeepVariantCall(
      variant=T.Variant(reference_name='chr1', start=10, end=11,
                        reference_bases='A', alternate_bases=['C', 'G']),
      allele_support={'C': _supporting_reads('read1/1'),
                      'G': _supporting_reads('read2/1', 'read2/2')})
  read = T.make_read(read_base, start=10, cigar='1M', quals=[50],
                     name=read_name)
  read.read_number = read_number
  pie = make(default_options(
      other_allele_supporting_read_alpha=0.3 if other_color else 0.6))
  actual = pie.encode_read(dv_call, 'TAT', read, 9, [alt_allele])
  assert actual[0, 1, 4] == expected_color


_CUSTOM_SEQ = 'TTTTATGACAAAAAAGATGCGACGGTTCCGTAACCCATAAGAAAGAACGT'

CUSTOM_CHANNEL_CASES = [
    # (channels, cigar, fragment_length, expected unique values of channel 0)
    (['read_mapping_percent'], '20M5D20M5S', 10, {0, 203}),
    (['avg_base_quality'], '20M5D20M5S', 10, {0, 68}),
    (['identity'], '5M20D20M5S', 10, {0, 127}),
    (['gap_compressed_identity'], '5M20D20M5S', 10, {0, 243}),
    (['blank'], '20M5D20M5S', 10, {0}),
    (['insert_size'], '20M5D20M5S', 22, {0, 5}),
]


def get_encoded_custom(make, channel_set, cigar, fragment_length):
  # pileup_image_test.py:665-686
  dv_call = make_dv_call()
  read = T.make_read(_CUSTOM_SEQ, start=500, cigar=cigar, quals=range(1, 51),
                     name='read1', fragment_length=fragment_length)
  return make(default_options(channel_set)).encode_read(
      dv_call, _CUSTOM_SEQ, read, 500, [dv_call.variant.alternate_bases[0]])


def check_custom_channel(make, channels, cigar, fragment_length, expected):
  # pileup_image_test.py:688-719
  result = get_encoded_custom(make, channels, cigar, fragment_length)
  assert set(np.unique(result[:, :, 0]).tolist()) == expected


def check_custom_multi(make):
  # pileup_image_test.py:721-725
  result = get_encoded_custom(
      make, ['read_mapping_percent', 'gap_compressed_identity', 'blank'],
      '20M5D20M5S', 10)
  assert result.shape == (1, 50, 3)


def _make_pileup(make, seq, channels):
  # pileup_image_test.py:730-744
  dv_call = make_dv_call()
  read = T.make_read(seq, start=500, cigar='%dM' % len(seq),
                     quals=range(1, len(seq) + 1), name='read1')
  return make(default_options(channels)).encode_read(
      dv_call, seq, read, 500, [dv_call.variant.alternate_bases[0]])


GC_CASES = [('GC', 1.0), ('GAC', 0.66), ('GGAA', 0.50), ('ATTCTGTTAA', 0.20),
            ('TTTTTTTTTT', 0.00)]


def check_gc_content(make, seq, exp):
  # pileup_image_test.py:746-757
  result = _make_pileup(make, seq, ['gc_content'])
  assert abs(result[:, :, 0][0].max() / 254.0 - exp) < 0.005 + 1e-9


IS_HOMOPOLYMER_CASES = [
    ('AAATTCCC', [1, 1, 1, 0, 0, 1, 1, 1]),
    ('ATCGTTCCC', [0, 0, 0, 0, 0, 0, 1, 1, 1]),
    ('ATTCCCTTA', [0, 0, 0, 1, 1, 1, 0, 0, 0]),
    ('ATCG', [0, 0, 0, 0]),
    ('AATTCCGG', [0] * 8),
    ('AAAAAAAA', [1] * 8),
]


def check_is_homopolymer(make, seq, expected):
  # pileup_image_test.py:759-771.  NOTE the reads start with base quality 1 <
  # min_base_quality only matters at the variant start (col 10 - 500 < 0).
  result = _make_pileup(make, seq, ['is_homopolymer'])
  got = (result[:, :, 0][0] / 254.0).astype(int)
  assert (got == expected).all()


WEIGHTED_HOMOPOLYMER_CASES = [
    ('AAATTCCC', [3, 3, 3, 2, 2, 3, 3, 3]),
    ('ATCGTTCCC', [1, 1, 1, 1, 2, 2, 3, 3, 3]),
    ('ATTCCCTTA', [1, 2, 2, 3, 3, 3, 2, 2, 1]),
]


def check_weighted_homopolymer(make, seq, expected):
  # pileup_image_test.py:773-781
  result = _make_pileup(make, seq, ['homopolymer_weighted'])
  got = np.round((result[:, :, 0][0] / 254.0) * 30).astype(int)
  assert (got == expected).all()


# ---- pileup_image_native_test.cc:277-413 (BuildPileupForOneSampleTests) ----

def _cc_call(ref, alts, start):
  return T.DeepVariantCall(variant=T.Variant(
      reference_name='chr1', start=start, end=start + len(ref),
      reference_bases=ref, alternate_bases=list(alts)))


def _row(base, bq, mq):
  return np.stack([np.array(base), np.array(bq), np.array(mq)],
                  axis=1).astype(np.uint8)  # [W, 3]


_REF_ROW = _row([250, 30, 180, 100, 250, 30, 100, 30, 30, 30, 250],
                [254] * 11, [254] * 11)
_ZERO_ROW = _row([0] * 11, [0] * 11, [0] * 11)
_INS_BQ = [190] * 10 + [0]
_INS_MQ = [254] * 10 + [0]

BUILD_PILEUP_CASES = {
    'simple_case': dict(
        call=_cc_call('A', ['G'], 5),
        reads=[('ACGTGCTCCCA', ['11M'], 'read_2', -1),
               ('ACGTGCTCCCA', ['11M'], 'read_3', -1)],
        rows=[_REF_ROW,
              _row([250, 30, 180, 100, 180, 30, 100, 30, 30, 30, 250],
                   [190] * 11, [254] * 11),
              _row([250, 30, 180, 100, 180, 30, 100, 30, 30, 30, 250],
                   [190] * 11, [254] * 11),
              _ZERO_ROW]),
    'no_reads': dict(
        call=_cc_call('A', ['G'], 5), reads=[],
        rows=[_REF_ROW, _ZERO_ROW, _ZERO_ROW, _ZERO_ROW]),
    'numer_of_reads_greater_than_max_reads': dict(
        call=_cc_call('A', ['AGG'], 5),
        reads=[('ACGTAGGCTCCCA', ['5M', '2I', '5M'], 'read_2', -1),
               ('ACGTAGGCTCCCA', ['5M', '2I', '5M'], 'read_3', -1),
               ('ACGTAGGGCTCCCA', ['5M', '3I', '5M'], 'read_4', -1),
               ('ACGTAGGGCTCCCA', ['5M', '3I', '5M'], 'read_5', -1)],
        rows=[_REF_ROW] + [
            _row([250, 30, 180, 100, 0, 30, 100, 30, 30, 30, 0], _INS_BQ,
                 _INS_MQ)] * 3),
    'image_creation_with_haplotype_sorting': dict(
        call=_cc_call('A', ['AGG', 'AGGG'], 5),
        reads=[('ACGTAGGCTCCCA', ['5M', '2I', '5M'], 'read_2', 2),
               ('TCGTAGGCTCCCA', ['5M', '2I', '5M'], 'read_3', 0),
               ('CCGTAGGGCTCCCA', ['5M', '3I', '5M'], 'read_4', 1)],
        rows=[_REF_ROW,
              _row([250, 30, 180, 100, 0, 30, 100, 30, 30, 30, 0], _INS_BQ,
                   _INS_MQ),
              _row([100, 30, 180, 100, 0, 30, 100, 30, 30, 30, 0], _INS_BQ,
                   _INS_MQ),
              _row([30, 30, 180, 100, 0, 30, 100, 30, 30, 30, 0], _INS_BQ,
                   _INS_MQ)]),
}


def check_build_pileup_case(make, name):
  case = BUILD_PILEUP_CASES[name]
  opts = cc_options(11, 4, 1, ['read_base', 'base_quality', 'mapping_quality'])
  reads = [T.cc_make_read('chr1', 0, seq, cig, rname, hp)
           for seq, cig, rname, hp in case['reads']]
  img = make(opts).build_pileup_for_one_sample(
      case['call'], 'ACGTACTCCCA', reads, 0, ['G'], T.SampleOptions())
  img = np.asarray(img)
  assert img.shape == (4, 11, 3)
  # The reference asserts UnorderedElementsAreArray over rows.
  got = sorted(bytes(img[r].tobytes()) for r in range(4))
  exp = sorted(bytes(r.tobytes()) for r in case['rows'])
  assert got == exp
  # Reference band is always row 0 (pileup_image_native.cc:321-323).
  np.testing.assert_equal(img[0], case['rows'][0])


# ---- pileup_channel_lib_test.cc:696-849 (GetChannelDataTest) ---------------

GET_CHANNEL_DATA_CHANNELS = [
    'read_base', 'base_quality', 'mapping_quality', 'strand',
    'read_supports_variant', 'base_differs_from_ref', 'read_mapping_percent',
    'avg_base_quality', 'identity', 'gap_compressed_identity', 'gc_content',
    'is_homopolymer', 'homopolymer_weighted', 'blank', 'insert_size',
    'supplementary_alignment',
]


def get_channel_data_options():
  o = T.PileupImageOptions(
      mapping_quality_cap=1, positive_strand_color=20,
      allele_unsupporting_read_alpha=1.0, base_color_offset_a_and_g=1,
      base_color_offset_t_and_c=1, base_color_stride=1, base_quality_cap=20,
      reference_matching_read_alpha=1, reference_mismatching_read_alpha=0,
      width=13, height=4, reference_band_height=1)
  o.channels = list(GET_CHANNEL_DATA_CHANNELS)
  o.num_channels = len(o.channels)
  return o


def check_get_channel_data(encode_read_with_blank, blank):
  """encode_read_with_blank(options, dv_call, ref, read, start, alts, blank)."""
  o = get_channel_data_options()
  seq = 'GGGCGCTTTTAT'
  ref = seq + 'N'  # width must be odd for the encoder ctor; col 12 unused.
  read = T.cc_make_read('chr1', 1, seq, ['11M'], 'r')
  read.fragment_length = 1000
  read.aligned_quality = [33] * len(seq)
  dv_call = T.DeepVariantCall()
  data = encode_read_with_blank(o, dv_call, ref, read, 0, [], blank)[0]
  idx = {name: i for i, name in enumerate(GET_CHANNEL_DATA_CHANNELS)}
  ch = lambda name: data[:, idx[name]]
  if T.DeepVariantChannelEnum.CH_READ_BASE not in blank:
    assert [ch('read_base')[c] for c in (11, 9, 1, 4)] == [4, 2, 3, 1]
  else:
    assert [ch('read_base')[c] for c in (11, 9, 1, 4)] == [0, 0, 0, 0]
  assert ch('base_quality')[1] == 254
  if T.DeepVariantChannelEnum.CH_MAPPING_QUALITY not in blank:
    assert ch('mapping_quality')[1] == 254
  else:
    assert ch('mapping_quality')[1] == 0
  assert ch('strand')[1] == 20
  assert ch('read_supports_variant')[1] == 254
  assert ch('base_differs_from_ref')[1] == 254
  assert ch('read_mapping_percent')[3] == 231
  assert ch('avg_base_quality')[3] == 90
  assert ch('identity')[9] == 231
  assert ch('gap_compressed_identity')[9] == 254
  assert ch('gc_content')[3] == 127
  assert ch('is_homopolymer')[1] == 254
  assert ch('is_homopolymer')[4] == 0
  assert ch('homopolymer_weighted')[1] == 25
  assert ch('homopolymer_weighted')[9] == 33
  assert ch('blank')[1] == 0
  assert ch('insert_size')[1] == 254


# ---- read_supports_variant_fuzzy -------------------------------------------------------
# channels/read_supports_variant_fuzzy_channel_test.cc:98-158 (7 TEST_Fs, 11 expectations) and
# pileup_channel_lib_test.cc:288-368 (ReadSupportsAltFuzzy, 4 expectations).
# (ref, alts, allele_support, ALT_PS values or None, ref_support, read name, read number, HP or
#  None, alts of the image, expected ReadSupportsAlt)
def _fuzzy_lib_case(read, hp, expected):
  return ('GGGCGC', ['GGGCGCATT', 'GGGCGCAT', 'GGGCGCATTT', 'GGGCGCA'],
          {'GGGCGCATT': ['Read1/1'], 'GGGCGCAT': ['Read2/1'], 'GGGCGCATTT': ['Read3/1'],
           'GGGCGCA': ['Read4/1']}, [1, 1, 1, 2], [], read, 1, hp, ['GGGCGCATT'], expected)


FUZZY_CASES = [
    # ExactMatch
    ('A', ['AC', 'ACC'], {'AC': ['read1/0'], 'ACC': ['read2/0']}, [0, 1, 1], [], 'read1', 0, 1, ['AC'], 1),
    ('A', ['AC', 'ACC'], {'AC': ['read1/0'], 'ACC': ['read2/0']}, [0, 1, 1], [], 'read2', 0, 1, ['AC'], 10),
    ('A', ['AC', 'ACC'], {'AC': ['read1/0'], 'ACC': ['read2/0']}, [0, 1, 1], [], 'read1', 0, 1, ['ACC'], 10),
    ('A', ['AC', 'ACC'], {'AC': ['read1/0'], 'ACC': ['read2/0']}, [0, 1, 1], [], 'read2', 0, 1, ['ACC'], 1),
    # FuzzyMatch1bp, FuzzyMatch2bp
    ('A', ['AC', 'ACC'], {'ACC': ['read1/0']}, [0, 1, 1], [], 'read1', 0, 1, ['AC'], 10),
    ('A', ['AC', 'ACCC'], {'ACCC': ['read1/0']}, [0, 1, 1], [], 'read1', 0, 1, ['AC'], 9),
    # FuzzyMatchPhaseMismatch (read HP 2, allele phase 1), FuzzyMatchPhaseZero
    ('A', ['AC', 'ACC'], {'ACC': ['read1/0']}, [0, 1, 1], [], 'read1', 0, 2, ['AC'], 0),
    ('A', ['AC', 'ACC'], {'ACC': ['read1/0']}, [0, 0, 0], [], 'read1', 0, 1, ['AC'], 10),
    # RefSupport: a reference-supporting read is one base away from "AA", three from "ATGC"
    ('A', ['ATGC', 'AA'], {}, None, ['read1/0'], 'read1', 0, None, ['ATGC'], 0),
    ('A', ['ATGC', 'AA'], {}, None, ['read1/0'], 'read1', 0, None, ['AA'], 10),
    # pileup_channel_lib_test.cc ReadSupportsAltFuzzy
    _fuzzy_lib_case('Read1', 1, 1), _fuzzy_lib_case('Read2', 1, 10),
    _fuzzy_lib_case('Read3', 1, 10), _fuzzy_lib_case('Read4', 2, 0),
]


def fuzzy_inputs(case):
  ref, alts, support, alt_ps, ref_support, name, number, hp, image_alts, expected = case
  variant = T.Variant(reference_name='chr1', start=10, end=10 + len(ref), reference_bases=ref,
                      alternate_bases=list(alts))
  if alt_ps is not None:
    variant.info['ALT_PS'] = T.ListValue(values=[T.Value(int_value=v) for v in alt_ps])
  call = T.DeepVariantCall(
      variant=variant,
      allele_support={k: _supporting_reads(*v) for k, v in support.items()},
      ref_support=list(ref_support))
  read = T.make_read('A', start=10, cigar='1M', quals=[50], name=name)
  read.read_number = number
  if hp is not None:
    read.info['HP'] = T.ListValue(values=[T.Value(int_value=hp)])
  return call, read, image_alts, expected


def fuzzy_color(options, code):
  # ReadSupportsVariantFuzzyChannel::SupportsAltColor (:290-312) in fp32
  alpha = {0: options.allele_unsupporting_read_alpha, 1: options.allele_supporting_read_alpha,
           2: options.other_allele_supporting_read_alpha, 10: 0.90, 9: 0.80}[code]
  return int(np.float32(254.0) * np.float32(alpha))


def check_fuzzy_channel(make, case):
  """The fuzzy-support pixel of a one-base read drawn through the encoder's own interface,
  and the reference row (SupportsAltColor(0))."""
  call, read, image_alts, expected = fuzzy_inputs(case)
  options = default_options(channels=['read_base', 'read_supports_variant_fuzzy'],
                            other_allele_supporting_read_alpha=0.3)
  enc = make(options)
  actual = enc.encode_read(call, 'TAT', read, 9, image_alts)
  assert list(actual[0, 1]) == [250, fuzzy_color(options, expected)]
  assert list(actual[0, 0]) == [0, 0] and list(actual[0, 2]) == [0, 0]
  ref_row = enc.encode_reference('TAT')
  assert list(ref_row[0, :, 1]) == [fuzzy_color(options, 0)] * 3
