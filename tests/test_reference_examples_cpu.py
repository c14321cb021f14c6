"""The top of the hot path against the REFERENCE's own code: ExamplesGenerator::WriteExamplesInRegion
(deepvariant/make_examples_native.cc compiled unmodified into oracle/_ref/libdvref.so, with the reference's
InMemoryReader, alt-allele combinations, per-sample pileups, alt-aligned images through its own TrimReads /
RealignReadsToHaplotype / FastPassAligner, and EncodeExample).

The product side is deepvariant_amd.make_examples_native.ExamplesGenerator.encode_region with ONE substitution, made
by the test: the device encoder (there is no GPU in the CPU suite) is replaced by the oracle restatement's packed
adapter.  Everything else is product code: the native region packer (dv_pack_region: read query per candidate, support
codes from allele_support name lists), alt-allele combinations, trimming, haplotypes and the native realigner for alt
images, row / channel layouts, tf.Example assembly.  Every example must agree in every feature: locus, variant_type,
alt_allele_indices/encoded, image/shape, sequencing_type, the variant (decoded) and EVERY PIXEL.
tests/test_hip_reference_examples.py repeats it on the GPU with nothing substituted.
"""
import numpy as np
import pytest

from oracle import oracle as O

if not O.reference_available():
  pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here', allow_module_level=True)

from deepvariant_amd import dv_types as T                   # noqa: E402
from deepvariant_amd import make_examples_native as men     # noqa: E402
from deepvariant_amd import protowire as pw                 # noqa: E402


class OracleDeviceEncoder:
  """Stands in for make_examples_native._Encoder in the CPU suite: same call, pixels from the oracle."""

  def __init__(self, pic):
    self._pic = pic

  def encode(self, batch, out_channels, min_bytes=0):
    images, rows = O.encode_packed(self._pic, batch, out_channels, n_threads=4)
    if images.size < min_bytes:
      images = np.concatenate([images, np.zeros(min_bytes - images.size, np.uint8)])
    return images, rows


def product_examples(options, ref, candidates, reads_per_sample, sample_order, role, coverage):
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  gen._device_encoder = OracleDeviceEncoder(options.pic_options)      # pylint: disable=protected-access
  stats = {}
  examples, shape = gen.encode_region(candidates, reads_per_sample, sample_order, coverage, stats, role=role)
  return examples, shape


def same_examples(mine, theirs, shape_mine, shape_theirs):
  assert list(shape_mine) == list(shape_theirs)
  assert len(mine) == len(theirs)
  for k, (a, b) in enumerate(zip(mine, theirs)):
    x, y = pw.decode_example(a), pw.decode_example(b)
    assert set(x) == set(y), (k, set(x) ^ set(y))
    for key in ('locus', 'variant_type', 'alt_allele_indices/encoded', 'image/shape', 'sequencing_type'):
      assert x[key] == y[key], (k, key, x[key], y[key])
    va, vb = pw.decode_variant(x['variant/encoded'][0]), pw.decode_variant(y['variant/encoded'][0])
    assert (va.reference_name, va.start, va.end, va.reference_bases, list(va.alternate_bases)) == \
           (vb.reference_name, vb.start, vb.end, vb.reference_bases, list(vb.alternate_bases)), k
    assert [(c.call_set_name, list(c.genotype), {i: [(v.int_value, v.number_value) for v in lv.values] for i, lv in c.info.items()})
            for c in va.calls] == \
           [(c.call_set_name, list(c.genotype), {i: [(v.int_value, v.number_value) for v in lv.values] for i, lv in c.info.items()})
            for c in vb.calls], k
    ia = np.frombuffer(x['image/encoded'][0], np.uint8).reshape(shape_mine)
    ib = np.frombuffer(y['image/encoded'][0], np.uint8).reshape(shape_theirs)
    assert np.array_equal(ia, ib), (k, x['locus'], np.argwhere(ia != ib)[:4].tolist())


def test_illumina_golden_region_examples():
  """BASELINE configs[0]: the 78 golden candidates of chr20:10,000,000-10,010,000 over the raw reads of the region:
  84 examples [100, 221, 7] (multi-allelic sites give three)."""
  from tests import golden_io
  from tests import realigner_fixture as RF
  from tests.golden.make_golden import wgs_options
  from tests.test_oracle_golden import FIXTURE
  reads, examples, _ = golden_io.load(FIXTURE)
  ref, _ = RF.load()
  pic = wgs_options()
  options = T.MakeExamplesOptions(pic_options=pic, sample_options=[
      T.SampleOptions(role='main', name='NA12878', pileup_height=100, order=[0])])
  cands, seen = [], set()
  for ex in examples:
    key = (ex['call'].variant.start, tuple(ex['call'].variant.alternate_bases))
    if key not in seen:
      seen.add(key)
      cands.append(ex['call'])
  theirs, shape_t = O.reference_write_examples_in_region(options, ref, 'chr20', 63025520, cands, [reads], [0], 'main', [0.0])
  mine, shape_m = product_examples(options, ref, cands, [reads], [0], 'main', [0.0])
  assert len(theirs) == 84 and shape_t == [100, 221, 7]
  same_examples(mine, theirs, shape_m, shape_t)


@pytest.mark.parametrize('sort_by_support', [False, True])
def test_two_samples_stacked(sort_by_support):
  """DeepTrio-shaped: two samples of different pileup heights in one example, one of them deeper than its rows
  (the shuffle on every item), 30-odd candidates sharing the region's reads, multi-allelic sites."""
  from tests import fuzz_inputs as F
  from tests.test_hip_region_multisample import _Ref, _query, _region_reads
  rng = np.random.default_rng(20260921)
  width = 81
  pic = F.options(T.PILEUP_CHANNELS_WITH_INSERT_SIZE, width, 0, sort_by_alt_allele_support=sort_by_support)
  pic.num_channels = len(pic.channels)
  heights = (40, 60)
  pic.height = sum(heights)
  options = T.MakeExamplesOptions(pic_options=pic, sample_options=[
      T.SampleOptions(role='child', name='c', pileup_height=heights[0], order=[0, 1]),
      T.SampleOptions(role='parent', name='p', pileup_height=heights[1], order=[0, 1])])
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=4000)))
  reads = [_region_reads(rng, 420, 900, 2100, 'c'), _region_reads(rng, 2600, 900, 2100, 'p')]
  cands = []
  for pos in sorted(set(rng.integers(1000, 2000, size=34).tolist())):
    refb = ref.seq[pos]
    alts = [b for b in 'ACGT' if b != refb][:int(rng.integers(1, 3))]
    near = [r for s in reads for r in _query(s, pos - 5, pos + 6)]
    support = {}
    for a in alts:
      pick = rng.integers(0, len(near), size=int(rng.integers(0, 12)))
      support[a] = T.SupportingReads(['%s/%d' % (near[int(j)].fragment_name, near[int(j)].read_number) for j in pick])
    v = T.Variant('chr1', pos, pos + 1, refb, alts, calls=[T.VariantCall(call_set_name='c', genotype=[-1, -1])])
    v.calls[0].info['AD'] = T.ListValue(values=[T.Value(int_value=int(x)) for x in rng.integers(0, 40, size=len(alts) + 1)])
    v.calls[0].info['DP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(1, 90)))])
    v.calls[0].info['VAF'] = T.ListValue(values=[T.Value(number_value=float(x)) for x in rng.random(len(alts))])
    cands.append(T.DeepVariantCall(variant=v, allele_support=support))
  theirs, shape_t = O.reference_write_examples_in_region(options, ref, 'chr1', len(ref.seq), cands, reads, [0, 1], 'child', [0.0, 0.0])
  mine, shape_m = product_examples(options, ref, cands, reads, [0, 1], 'child', [0.0, 0.0])
  assert shape_t == [100, width, len(pic.channels)] and len(theirs) >= len(cands)
  same_examples(mine, theirs, shape_m, shape_t)


@pytest.mark.parametrize('mode,types,pacbio', [('diff_channels', 'all', False), ('base_channels', 'indels', False),
                                               ('rows', 'all', False), ('single_row', 'indels', False),
                                               ('diff_channels', 'indels', True)])
def test_alt_aligned_pileups(mode, types, pacbio):
  """--alt_aligned_pileup in all four layouts (the PacBio / ONT models use diff_channels; the last case is the released
  PacBio model's [100, 147, 10] tensor): per candidate and alt allele the reference trims the reads to the window
  (TrimReads), builds the haplotype (CreateHaplotype), realigns (RealignReadsToHaplotype -> FastPassAligner) and
  draws them -- all of it the reference's own code here -- and lays the alt images out as channels or row blocks."""
  from tests.test_hip_region_multisample import _alt_region
  g = _alt_region(mode, types, pacbio)
  options, ref, reads, cands = g['options'], g['ref'], g['reads'], g['cands']
  for so in options.sample_options:
    so.order = [0]
  theirs, shape_t = O.reference_write_examples_in_region(options, ref, 'chr1', len(ref.seq), cands, [reads], [0], 'main', [0.0],
                                                         aln_config=men.DEFAULT_ALN_CONFIG)
  mine, shape_m = product_examples(options, ref, cands, [reads], [0], 'main', [0.0])
  mult = {'rows': 3, 'single_row': 2}.get(mode, 1)
  assert shape_t == [g['height'] * mult, g['width'], len(g['channels']) + len(g['extra'])] and len(theirs) > 20
  same_examples(mine, theirs, shape_m, shape_t)
