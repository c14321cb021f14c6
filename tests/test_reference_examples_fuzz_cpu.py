"""A slice of tools/ref_fuzz_examples.py in the suite: random ExamplesGenerator configurations (samples, orders, roles,
channel lists, blanked channels, window-spanning reads, multi-allelic modes, overlap buffers, trimming, candidates at the
contig ends, every --alt_aligned_pileup layout) -- the product's encode_region == the reference's
WriteExamplesInRegion (oracle/_ref/libdvref.so) in every feature and every pixel.  The full campaign (3,600 cases,
129,114 examples, 0 differing) is profiles/r04_reference_fuzz_examples.txt."""
import pytest

from oracle import oracle as O

if not O.reference_available():
  pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here', allow_module_level=True)

from tools import ref_fuzz_examples as RF      # noqa: E402


@pytest.mark.parametrize('first', [0, 40, 80])
def test_region_option_cases(first):
  n = sum(max(RF.run_case(seed), 0) for seed in range(first, first + 40))
  assert n > 500


def test_a_case_the_reference_refuses_is_refused():
  """trim_reads_for_pileup + a read without reference bases inside the window: the reference CHECK-fails, the product
  raises the same check (run_case returns -1 for such a pair, raises if only one side refuses)."""
  assert any(RF.run_case(seed) == -1 for seed in range(200, 420))


@pytest.mark.parametrize('seed', range(8))
def test_alt_aligned_cases(seed):
  assert RF.run_case(seed, alt=True) > 5
