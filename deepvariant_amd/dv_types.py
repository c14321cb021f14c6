"""Plain-Python mirrors of the protos that cross the hot-path boundary.

The reference passes protobuf messages through its pybind boundary
(`deepvariant/python/pileup_image_native_pybind.cc:57-129`,
`deepvariant/python/make_examples_native_pybind.cc:56-108`).  protoc is not
available in this image, so the host side is written against *field names*
only: the dataclasses below carry the same attribute names as

  * nucleus.genomics.v1.Read / LinearAlignment / Position / CigarUnit
      (third_party/nucleus/protos/reads.proto, cigar.proto, position.proto)
  * nucleus.genomics.v1.Variant / VariantCall  (variants.proto)
  * DeepVariantCall, PileupImageOptions, SampleOptions, MakeExamplesOptions,
    ReadRequirements                         (deepvariant/protos/deepvariant.proto)

so real protobuf objects (which expose exactly these attributes) can be handed
to `deepvariant_amd.pileup_image_native` / `make_examples_native` unchanged.
"""
from __future__ import annotations

import dataclasses
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence


class CigarOp(enum.IntEnum):
  """CigarUnit.Operation, third_party/nucleus/protos/cigar.proto:38-82."""
  OPERATION_UNSPECIFIED = 0
  ALIGNMENT_MATCH = 1
  INSERT = 2
  DELETE = 3
  SKIP = 4
  CLIP_SOFT = 5
  CLIP_HARD = 6
  PAD = 7
  SEQUENCE_MATCH = 8
  SEQUENCE_MISMATCH = 9


CIGAR_CHAR_TO_OP = {
    'M': CigarOp.ALIGNMENT_MATCH, 'I': CigarOp.INSERT, 'D': CigarOp.DELETE,
    'N': CigarOp.SKIP, 'S': CigarOp.CLIP_SOFT, 'H': CigarOp.CLIP_HARD,
    'P': CigarOp.PAD, '=': CigarOp.SEQUENCE_MATCH,
    'X': CigarOp.SEQUENCE_MISMATCH,
}
# BAM's op numbering (MIDNSHP=X) -> nucleus numbering.
BAM_OP_TO_NUCLEUS = [1, 2, 3, 4, 5, 6, 7, 8, 9]


class DeepVariantChannelEnum(enum.IntEnum):
  """deepvariant/protos/deepvariant.proto:1287-1342."""
  CH_UNSPECIFIED = 0
  CH_READ_BASE = 1
  CH_BASE_QUALITY = 2
  CH_MAPPING_QUALITY = 3
  CH_STRAND = 4
  CH_READ_SUPPORTS_VARIANT = 5
  CH_BASE_DIFFERS_FROM_REF = 6
  CH_HAPLOTYPE_TAG = 7
  CH_ALLELE_FREQUENCY = 8
  CH_DIFF_CHANNELS_ALTERNATE_ALLELE_1 = 9
  CH_DIFF_CHANNELS_ALTERNATE_ALLELE_2 = 10
  CH_READ_MAPPING_PERCENT = 11
  CH_AVG_BASE_QUALITY = 12
  CH_IDENTITY = 13
  CH_GAP_COMPRESSED_IDENTITY = 14
  CH_GC_CONTENT = 15
  CH_IS_HOMOPOLYMER = 16
  CH_HOMOPOLYMER_WEIGHTED = 17
  CH_BLANK = 18
  CH_INSERT_SIZE = 19
  CH_BASE_CHANNELS_ALTERNATE_ALLELE_1 = 20
  CH_BASE_CHANNELS_ALTERNATE_ALLELE_2 = 21
  CH_MEAN_COVERAGE = 22
  CH_BASE_METHYLATION = 23
  CH_BASE_6MA = 24
  CH_READ_SUPPORTS_VARIANT_FUZZY = 25
  CH_SUPPLEMENTARY_ALIGNMENT = 26
  CH_ALLELE_SAMPLE_PROBABILITY = 27
  CH_HOMOPOLYMER_INSERTION_QUALITY = 28
  CH_HOMOPOLYMER_DELETION_QUALITY = 29
  CH_INTER_HOMOPOLYMER_INSERTION_QUALITY = 30


# Channels::ChannelStrToEnum, deepvariant/pileup_channel_lib.cc:421-512.
CHANNEL_STR_TO_ENUM: Dict[str, int] = {
    'read_base': 1, 'base_quality': 2, 'mapping_quality': 3, 'strand': 4,
    'read_supports_variant': 5, 'read_supports_variant_fuzzy': 25,
    'base_differs_from_ref': 6, 'read_mapping_percent': 11, 'haplotype': 7,
    'allele_frequency': 8,
    'diff_channels_alternate_allele_1': 0, 'diff_channels_alternate_allele_2': 0,
    'avg_base_quality': 12, 'identity': 13, 'gap_compressed_identity': 14,
    'gc_content': 15, 'is_homopolymer': 16, 'homopolymer_weighted': 17,
    'blank': 18, 'insert_size': 19,
    'base_channels_alternate_allele_1': 0, 'base_channels_alternate_allele_2': 0,
    'mean_coverage': 22, 'base_methylation': 23, 'base_6ma': 24,
    'supplementary_alignment': 26, 'allele_sample_probability': 27,
    'homopolymer_insertion_quality': 28, 'homopolymer_deletion_quality': 29,
    'inter_homopolymer_insertion_quality': 30,
}

# The enum each *name* gets in `<examples>.example_info.json` "channels"
# (make_examples_core.py:3766-3774): alt-aligned names keep their own enums.
CHANNEL_NAME_TO_INFO_ENUM = dict(CHANNEL_STR_TO_ENUM)
CHANNEL_NAME_TO_INFO_ENUM.update({
    'diff_channels_alternate_allele_1': 9, 'diff_channels_alternate_allele_2': 10,
    'base_channels_alternate_allele_1': 20, 'base_channels_alternate_allele_2': 21,
})

# deepvariant/dv_constants.py:41-62
PILEUP_DEFAULT_HEIGHT = 100
PILEUP_DEFAULT_WIDTH = 221
PILEUP_DEFAULT_CHANNELS = [
    'read_base', 'base_quality', 'mapping_quality', 'strand',
    'read_supports_variant', 'base_differs_from_ref',
]
PILEUP_CHANNELS_WITH_INSERT_SIZE = PILEUP_DEFAULT_CHANNELS + ['insert_size']

# nucleus base-modification keys (third_party/nucleus/util/utils.h).
K5MC = '5mC'
K6MA = '6mA'


@dataclass
class CigarUnit:
  operation: int = 0
  operation_length: int = 0


@dataclass
class Position:
  reference_name: str = ''
  position: int = 0
  reverse_strand: bool = False


@dataclass
class Range:
  """nucleus.genomics.v1.Range: [start, end) on reference_name."""
  reference_name: str = ''
  start: int = 0
  end: int = 0


@dataclass
class LinearAlignment:
  position: Position = field(default_factory=Position)
  mapping_quality: int = 0
  cigar: List[CigarUnit] = field(default_factory=list)


@dataclass
class Value:
  """nucleus.genomics.v1.Value (struct.proto): a oneof `kind`."""
  int_value: Optional[int] = None
  string_value: Optional[str] = None
  number_value: Optional[float] = None

  def WhichOneof(self, _name):
    if self.int_value is not None:
      return 'int_value'
    if self.string_value is not None:
      return 'string_value'
    if self.number_value is not None:
      return 'number_value'
    return None


@dataclass
class ListValue:
  values: List[Value] = field(default_factory=list)


@dataclass
class Read:
  fragment_name: str = ''
  read_number: int = 0
  number_reads: int = 0
  proper_placement: bool = False
  duplicate_fragment: bool = False
  failed_vendor_quality_checks: bool = False
  secondary_alignment: bool = False
  supplementary_alignment: bool = False
  fragment_length: int = 0
  aligned_sequence: str = ''
  aligned_quality: Sequence[int] = field(default_factory=list)
  alignment: LinearAlignment = field(default_factory=LinearAlignment)
  info: Dict[str, ListValue] = field(default_factory=dict)
  base_modifications: Dict[str, bytes] = field(default_factory=dict)


@dataclass
class VariantCall:
  call_set_name: str = ''
  genotype: List[int] = field(default_factory=list)
  info: Dict[str, ListValue] = field(default_factory=dict)


@dataclass
class Variant:
  reference_name: str = ''
  start: int = 0
  end: int = 0
  reference_bases: str = ''
  alternate_bases: List[str] = field(default_factory=list)
  calls: List[VariantCall] = field(default_factory=list)
  # variant.info (e.g. ALT_PS, the per-allele phase the fuzzy support channel reads) and the
  # alt alleles the caller rejected (variants.proto:75,90)
  info: Dict[str, ListValue] = field(default_factory=dict)
  alternate_bases_rejected: List[str] = field(default_factory=list)
  # Opaque serialized form when the variant was decoded from the wire; used to
  # re-emit `variant/encoded` byte-for-byte.
  serialized: Optional[bytes] = None


@dataclass
class SupportingReads:
  read_names: List[str] = field(default_factory=list)


@dataclass
class ReadSupport:
  """DeepVariantCall.ReadSupport (the fields phasing reads)."""
  read_name: str = ''
  is_low_quality: bool = False


@dataclass
class AltAlleleIndices:
  indices: List[int] = field(default_factory=list)


@dataclass
class DeepVariantCall:
  variant: Variant = field(default_factory=Variant)
  allele_support: Dict[str, SupportingReads] = field(default_factory=dict)
  allele_frequency: Dict[str, float] = field(default_factory=dict)
  rejected_allele_support: Dict[str, SupportingReads] = field(default_factory=dict)
  ref_support: List[str] = field(default_factory=list)
  # allele_support_ext[allele].read_infos / ref_support_ext.read_infos (track_ref_reads, phasing)
  allele_support_ext: Dict[str, List[ReadSupport]] = field(default_factory=dict)
  ref_support_ext: List[ReadSupport] = field(default_factory=list)
  make_examples_alt_allele_indices: List[AltAlleleIndices] = field(
      default_factory=list)


@dataclass
class ReadRequirements:
  min_mapping_quality: int = 0
  min_base_quality: int = 0
  min_base_quality_mode: int = 0


class MultiAllelicMode(enum.IntEnum):
  UNSPECIFIED = 0
  ADD_HET_ALT_IMAGES = 1
  NO_HET_ALT_IMAGES = 2


@dataclass
class PileupImageOptions:
  """deepvariant/protos/deepvariant.proto:500-638 (same field names)."""
  height: int = 0
  width: int = 0
  reference_band_height: int = 0
  base_color_offset_a_and_g: int = 0
  base_color_offset_t_and_c: int = 0
  base_color_stride: int = 0
  reference_alpha: float = 0.0
  reference_base_quality: int = 0
  allele_supporting_read_alpha: float = 0.0
  other_allele_supporting_read_alpha: float = 0.0
  allele_unsupporting_read_alpha: float = 0.0
  reference_matching_read_alpha: float = 0.0
  reference_mismatching_read_alpha: float = 0.0
  indel_anchoring_base_char: str = ''
  positive_strand_color: int = 0
  negative_strand_color: int = 0
  base_quality_cap: int = 0
  read_overlap_buffer_bp: int = 0
  read_requirements: ReadRequirements = field(default_factory=ReadRequirements)
  multi_allelic_mode: int = 0
  mapping_quality_cap: int = 0
  random_seed: int = 0
  num_channels: int = 0
  sequencing_type: int = 0
  alt_aligned_pileup: str = ''
  sort_by_haplotypes: bool = False
  reverse_haplotypes: bool = False
  min_non_zero_allele_frequency: float = 0.0
  use_allele_frequency: bool = False
  types_to_alt_align: str = ''
  hp_tag_for_assembly_polishing: int = 0
  channels: List[str] = field(default_factory=list)
  sort_by_alt_allele_support: bool = False

  def MergeFrom(self, other: 'PileupImageOptions'):
    """proto MergeFrom semantics: non-default scalar fields overwrite."""
    blank = PileupImageOptions()
    for f in dataclasses.fields(self):
      v = getattr(other, f.name)
      if f.name == 'channels':
        self.channels.extend(v)
      elif v != getattr(blank, f.name):
        setattr(self, f.name, v)


@dataclass
class SampleOptions:
  """deepvariant/protos/deepvariant.proto:642-725."""
  role: str = ''
  name: str = ''
  pileup_height: int = 0
  order: List[int] = field(default_factory=list)
  keep_only_window_spanning_reads: bool = False
  channels_enum_to_blank: List[int] = field(default_factory=list)
  variant_types_to_blank: List[int] = field(default_factory=list)
  use_non_uniform_downsampling: bool = False
  non_uniform_downsampling_threshold: int = 0
  alt_aligned_pileup: str = ''


@dataclass
class MakeExamplesOptions:
  """deepvariant/protos/deepvariant.proto:737-1076 (fields the path reads)."""
  pic_options: PileupImageOptions = field(default_factory=PileupImageOptions)
  sample_options: List[SampleOptions] = field(default_factory=list)
  reference_filename: str = ''
  trim_reads_for_pileup: bool = False
  stream_examples: bool = False
  denovo_regions_filename: str = ''
  mode: int = 0


def default_options(read_requirements: Optional[ReadRequirements] = None
                    ) -> PileupImageOptions:
  """deepvariant/pileup_image.py:36-74 (`default_options`)."""
  if not read_requirements:
    read_requirements = ReadRequirements(
        min_base_quality=10, min_mapping_quality=10, min_base_quality_mode=1)
  return PileupImageOptions(
      reference_band_height=5,
      base_color_offset_a_and_g=40,
      base_color_offset_t_and_c=30,
      base_color_stride=70,
      allele_supporting_read_alpha=1.0,
      allele_unsupporting_read_alpha=0.6,
      other_allele_supporting_read_alpha=0.6,
      reference_matching_read_alpha=0.2,
      reference_mismatching_read_alpha=1.0,
      indel_anchoring_base_char='*',
      reference_alpha=0.4,
      reference_base_quality=60,
      positive_strand_color=70,
      negative_strand_color=240,
      base_quality_cap=40,
      mapping_quality_cap=60,
      height=PILEUP_DEFAULT_HEIGHT,
      width=PILEUP_DEFAULT_WIDTH,
      read_overlap_buffer_bp=5,
      read_requirements=read_requirements,
      multi_allelic_mode=MultiAllelicMode.ADD_HET_ALT_IMAGES,
      random_seed=2101079370,
      sequencing_type=0,
      alt_aligned_pileup='none',
      types_to_alt_align='indels',
      min_non_zero_allele_frequency=0.00001,
      use_allele_frequency=False,
  )


def parse_cigar(cigar) -> List[CigarUnit]:
  """'20M5D20M5S' or ['5M', '2I'] -> CigarUnits (nucleus util/cigar.py)."""
  if isinstance(cigar, (list, tuple)):
    cigar = ''.join(cigar)
  units, num = [], ''
  for ch in cigar:
    if ch.isdigit():
      num += ch
    else:
      units.append(CigarUnit(int(CIGAR_CHAR_TO_OP[ch]), int(num)))
      num = ''
  if num:
    raise ValueError('malformed CIGAR: %r' % (cigar,))
  return units


def make_read(bases, start, quals=None, cigar=None, mapq=50, chrom='chr1',
              name=None, fragment_length=None, read_number=1,
              reverse_strand=False) -> Read:
  """third_party/nucleus/testing/test_utils.py:288-316 (`make_read`)."""
  if quals is not None and len(bases) != len(quals):
    raise ValueError('Incompatable bases and quals', bases, quals)
  make_read.counter += 1
  return Read(
      fragment_name=name if name else 'read_' + str(make_read.counter - 1),
      proper_placement=True,
      read_number=read_number,
      number_reads=2,
      aligned_sequence=bases,
      aligned_quality=list(quals) if quals is not None else [],
      fragment_length=fragment_length or 0,
      alignment=LinearAlignment(
          position=Position(reference_name=chrom, position=start,
                            reverse_strand=reverse_strand),
          mapping_quality=mapq,
          cigar=parse_cigar(cigar) if cigar else [],
      ),
  )


make_read.counter = 0


def cc_make_read(chrom, start, bases, cigar_elements, read_name,
                 hp_tag=-1) -> Read:
  """deepvariant/testing_utils.cc:77-90 + nucleus/testing/test_utils.cc:128-152.

  The C++ test factory: base quality 30 everywhere, MAPQ 90, read_number 0.
  """
  read = Read(
      fragment_name=read_name,
      read_number=0,
      number_reads=2,
      proper_placement=True,
      aligned_sequence=bases,
      aligned_quality=[30] * len(bases),
      alignment=LinearAlignment(
          position=Position(reference_name=chrom, position=start),
          mapping_quality=90,
          cigar=parse_cigar(list(cigar_elements)),
      ),
  )
  if hp_tag >= 0:
    read.info['HP'] = ListValue(values=[Value(int_value=hp_tag)])
  return read
