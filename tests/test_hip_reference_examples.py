"""tests/test_reference_examples_cpu.py on the GPU with NOTHING substituted: the product's
ExamplesGenerator.encode_region (native packer, ONE dv_encode_batch launch per region, alt images as items of the
same launch, channel / row layouts) against the REFERENCE's own ExamplesGenerator::WriteExamplesInRegion
(oracle/_ref/libdvref.so: deepvariant/make_examples_native.cc and everything under it compiled unmodified,
oracle/ref_build/) -- every feature of every tf.Example, every pixel."""
import pytest

from tests import test_reference_examples_cpu as CPU      # (skips itself when the reference build is absent)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_encoder(monkeypatch):
  """The CPU module swaps the device encoder for the oracle; here encode_region runs as shipped."""
  from deepvariant_amd import make_examples_native as men

  def product_examples(options, ref, candidates, reads_per_sample, sample_order, role, coverage):
    gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
    stats = {}
    return gen.encode_region(candidates, reads_per_sample, sample_order, coverage, stats, role=role)
  monkeypatch.setattr(CPU, 'product_examples', product_examples)


def test_illumina_golden_region_examples():
  CPU.test_illumina_golden_region_examples()


@pytest.mark.parametrize('sort_by_support', [False, True])
def test_two_samples_stacked(sort_by_support):
  CPU.test_two_samples_stacked(sort_by_support)


@pytest.mark.parametrize('mode,types,pacbio', [('diff_channels', 'all', False), ('base_channels', 'indels', False),
                                               ('rows', 'all', False), ('single_row', 'indels', False),
                                               ('diff_channels', 'indels', True)])
def test_alt_aligned_pileups(mode, types, pacbio):
  CPU.test_alt_aligned_pileups(mode, types, pacbio)
