"""Gives the CPU oracle the reference's pybind interface (tests only)."""
import numpy as np

from oracle import oracle as O


class OracleEncoder:
  """Same surface as deepvariant.python.pileup_image_native.PileupImageEncoderNative."""

  def __init__(self, options):
    if not (options.width % 2 == 1 and options.width >= 3):
      raise ValueError('Width must be odd; found %d' % options.width)
    self.options = options

  def encode_reference(self, ref_bases):
    return O.encode_reference(self.options, ref_bases)

  def encode_read(self, dv_call, ref_bases, read, image_start_pos, alt_alleles,
                  channels_to_blank=None):
    return O.encode_read(self.options, dv_call, ref_bases, read,
                         image_start_pos, list(alt_alleles), channels_to_blank)

  def build_pileup_for_one_sample(self, dv_call, ref_bases, reads,
                                  image_start_pos, alt_alleles, sample_options,
                                  mean_coverage=0.0, alignment_positions=None,
                                  channels_to_blank=None):
    return O.build_pileup(
        self.options, dv_call, ref_bases, list(reads), image_start_pos,
        list(alt_alleles), pileup_height=sample_options.pileup_height,
        mean_coverage=mean_coverage, alignment_positions=alignment_positions,
        channels_to_blank=channels_to_blank)


def make(options):
  return OracleEncoder(options)
