"""dv_pack_region (deepvariant_amd/csrc/region_packer.cpp) vs the Python packer of
ExamplesGenerator._plan_region (packing.py: ReadTable.query, support_codes, allele_groups)
-- the same PackedBatch, array for array -- on the golden HG001 slice and on seeded random
regions (overlapping candidates, reads listed under several alts, multi-allelic sites, a
candidate without a reference window, unsorted reads).  No GPU needed: both are host code."""
import os

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_native as men
from tests import fuzz_inputs as F


class _Ref:
  def __init__(self, seq):
    self.seq = seq

  def n_bases(self, contig):
    return len(self.seq)

  def get_bases(self, contig, start, end):
    return self.seq[start:end]


def _plan(gen, cands, reads, python):
  old = os.environ.pop('DV_PY_PACKER', None)
  if python:
    os.environ['DV_PY_PACKER'] = '1'
  try:
    return gen._plan_region(cands, [reads], [0], [3.0])
  finally:
    os.environ.pop('DV_PY_PACKER', None)
    if old is not None:
      os.environ['DV_PY_PACKER'] = old


def _same(a, b, groups):
  batch_a, plan_a, shape_a = a
  batch_b, plan_b, shape_b = b
  assert plan_a == plan_b and shape_a == shape_b
  names = ['item_variant_start', 'item_image_start', 'item_ref_idx', 'item_list_off', 'item_height',
           'item_out_off', 'item_blank_mask', 'item_mean_coverage', 'ref_windows', 'list_read',
           'list_code']
  if groups:
    names.append('list_group')
  for name in names:
    np.testing.assert_array_equal(getattr(batch_a, name), getattr(batch_b, name), err_msg=name)
  assert batch_a.max_list_len == batch_b.max_list_len
  assert (batch_a.list_group is None) == (batch_b.list_group is None)


@pytest.mark.parametrize('sort_by_support,shuffle_reads,n_cands,threads', [
    (False, False, 60, 4), (True, False, 60, 4), (False, True, 60, 4), (True, True, 60, 4),
    (True, False, 700, 1), (True, True, 700, 8)])      # 700 candidates: slices on several host threads
def test_native_packer_equals_python_packer(sort_by_support, shuffle_reads, n_cands, threads, monkeypatch):
  monkeypatch.setenv('DV_PACK_THREADS', str(threads))
  rng = np.random.default_rng(5 + sort_by_support + 2 * shuffle_reads)
  width = 61
  pic = F.options(T.PILEUP_CHANNELS_WITH_INSERT_SIZE, width, 40,
                  sort_by_alt_allele_support=sort_by_support)
  options = T.MakeExamplesOptions(
      pic_options=pic, sample_options=[T.SampleOptions(role='main', pileup_height=40)])
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=3000)))
  reads = []
  for i in range(700):
    cigar = F.random_cigar(rng, 1, 6)
    if F.query_len(cigar) == 0:
      cigar.append(T.CigarUnit(1, 1))
    qlen = F.query_len(cigar)
    reads.append(T.Read(
        fragment_name='f%d' % int(rng.integers(0, 300)), read_number=int(rng.integers(0, 2)),
        aligned_sequence='A' * qlen, aligned_quality=bytes([30] * qlen),
        alignment=T.LinearAlignment(position=T.Position('chr1', int(rng.integers(0, 2900)), False),
                                    mapping_quality=60, cigar=cigar)))
  if not shuffle_reads:
    reads.sort(key=lambda r: r.alignment.position.position)
  cands = []
  for pos in sorted(rng.integers(0, 2990, size=n_cands).tolist()):   # duplicates and contig edges included
    refb = ref.seq[pos]
    alts = [b for b in 'ACGT' if b != refb][:int(rng.integers(1, 4))]
    support = {}
    for a in alts + ['NOT_AN_ALT']:
      pick = rng.integers(0, len(reads), size=int(rng.integers(0, 40)))
      support[a] = T.SupportingReads(['%s/%d' % (reads[int(j)].fragment_name, reads[int(j)].read_number)
                                      for j in pick] + ['ghost/0'])
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + 1, refb, alts),
                                   allele_support=support))
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  native = _plan(gen, cands, reads, python=False)
  python = _plan(gen, cands, reads, python=True)
  assert native[0].n_items > len(cands) // 2
  _same(native, python, groups=sort_by_support)


def test_native_packer_on_the_golden_slice():
  from tests import golden_io
  from tests.golden.make_golden import wgs_options
  from tests.test_oracle_golden import FIXTURE
  from tests.test_hip_pipeline import _WindowRef
  reads, examples, _ = golden_io.load(FIXTURE)
  pic = wgs_options()
  options = T.MakeExamplesOptions(
      pic_options=pic, sample_options=[T.SampleOptions(role='main', pileup_height=100)])
  cands, seen = [], set()
  for ex in examples:
    key = (ex['call'].variant.start, tuple(ex['call'].variant.alternate_bases))
    if key not in seen:
      seen.add(key)
      cands.append(ex['call'])
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=_WindowRef(examples, pic.width))
  native = _plan(gen, cands, reads, python=False)
  python = _plan(gen, cands, reads, python=True)
  assert native[0].n_items == 84
  _same(native, python, groups=False)
