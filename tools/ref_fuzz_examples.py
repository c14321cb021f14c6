"""Random ExamplesGenerator configurations: deepvariant_amd.make_examples_native.ExamplesGenerator.encode_region against
the REFERENCE's own ExamplesGenerator::WriteExamplesInRegion (oracle/_ref/libdvref.so, make_examples_native.cc compiled
unmodified) -- every feature of every tf.Example, every pixel.  CPU tool (the device encoder is replaced by the
oracle's packed adapter exactly as in tests/test_reference_examples_cpu.py; the GPU form of those tests is
tests/test_hip_reference_examples.py).  TEST INFRASTRUCTURE.

  python tools/ref_fuzz_examples.py [n_cases] [first_seed] [alt]     -> a summary line; failures are printed with their seed

What varies per case: window width, 1-3 samples with their own heights / read sets / channels_enum_to_blank /
keep_only_window_spanning_reads / use_non_uniform_downsampling, the sample order and the role, the channel list (insert_size, haplotype + HP tags with
sort_by_haplotypes, mean_coverage, blank, is_homopolymer, base_methylation / base_6ma with per-base modification bytes ...), sort_by_alt_allele_support, multi_allelic_mode,
read_overlap_buffer_bp, trim_reads_for_pileup, SNP / insertion / deletion candidates with 1-3 alts (some with explicit
make_examples_alt_allele_indices), candidates at both ends of the contig (N padding), reads listed under several
alleles, under none, under alleles of other candidates.
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])

from deepvariant_amd import dv_types as T                     # noqa: E402
from oracle import oracle as O                                # noqa: E402
from tests import fuzz_inputs as F                            # noqa: E402
from tests import test_reference_examples_cpu as RE           # noqa: E402
from tests.test_hip_region_multisample import _Ref, _query, _region_reads    # noqa: E402


def make_case(seed):
  rng = np.random.default_rng(seed)
  width = int(rng.choice([21, 41, 81, 147, 221]))
  n_samples = int(rng.choice([1, 1, 2, 3]))
  channels = list(T.PILEUP_DEFAULT_CHANNELS)
  with_hp = False
  for extra, p in (('insert_size', .5), ('haplotype', .3), ('mean_coverage', .3), ('blank', .15),
                   ('is_homopolymer', .15), ('homopolymer_weighted', .15), ('gc_content', .15),
                   ('supplementary_alignment', .2), ('read_mapping_percent', .15), ('avg_base_quality', .15),
                   ('base_methylation', .15), ('base_6ma', .1)):
    if rng.random() < p and len(channels) < 12:
      channels.append(extra)
      with_hp |= extra == 'haplotype'
  sort_hp = with_hp and rng.random() < 0.7
  pic = F.options(channels, width, 0, sort_by_haplotypes=sort_hp,
                  sort_by_alt_allele_support=bool(rng.random() < 0.3),
                  min_mapq=int(rng.choice([0, 5, 10])), min_bq=int(rng.choice([0, 10, 20])))
  pic.num_channels = len(channels)
  pic.multi_allelic_mode = int(rng.choice([T.MultiAllelicMode.ADD_HET_ALT_IMAGES, T.MultiAllelicMode.ADD_HET_ALT_IMAGES,
                                           T.MultiAllelicMode.NO_HET_ALT_IMAGES]))
  pic.read_overlap_buffer_bp = int(rng.choice([0, 5, 5, 17]))
  heights = [int(rng.integers(8, 70)) for _ in range(n_samples)]
  roles = ['child', 'parent1', 'parent2'][:n_samples]
  enums = [int(T.CHANNEL_STR_TO_ENUM[c]) for c in channels]
  samples = []
  for s in range(n_samples):
    so = T.SampleOptions(role=roles[s], name='s%d' % s, pileup_height=heights[s])
    if rng.random() < 0.25:
      so.channels_enum_to_blank = sorted(set(int(enums[int(i)]) for i in rng.integers(0, len(enums), size=int(rng.integers(1, 3)))))
    so.keep_only_window_spanning_reads = bool(rng.random() < 0.15)
    if rng.random() < 0.2:      # every allele keeps a minimum of its supporters (a threshold too high: uniform again)
      so.use_non_uniform_downsampling = True
      so.non_uniform_downsampling_threshold = int(rng.choice([0, 1, 2, 5, 30]))
    samples.append(so)
  order = [int(x) for x in rng.permutation(n_samples)] if rng.random() < 0.5 else list(range(n_samples))
  for so in samples:
    so.order = list(order)
  role_idx = int(rng.integers(0, n_samples))
  L = int(rng.choice([600, 1500, 4000]))
  ref = _Ref(''.join('ACGTN'[int(i)] for i in rng.choice(5, size=L, p=[.245, .245, .245, .245, .02])))
  reads = []
  for s in range(n_samples):
    rs = _region_reads(rng, int(rng.choice([0, 40, 300, 1200])), -20, L - 10, 'q%d_' % s)
    for r in rs:
      if 'avg_base_quality' in channels:
        r.aligned_quality = bytes(min(q, 93) for q in r.aligned_quality)
      if 'base_methylation' in channels and rng.random() < 0.6:
        r.base_modifications[T.K5MC] = bytes(rng.integers(0, 256, size=len(r.aligned_sequence)).astype(np.uint8))
      if 'base_6ma' in channels and rng.random() < 0.6:
        r.base_modifications[T.K6MA] = bytes(rng.integers(0, 256, size=len(r.aligned_sequence)).astype(np.uint8))
      if with_hp and rng.random() < 0.6:
        r.info['HP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(0, 3)))])
      r.alignment.position.position = max(0, r.alignment.position.position)
    rs.sort(key=lambda r: r.alignment.position.position)
    reads.append(rs)
  pic.height = sum(heights)
  options = T.MakeExamplesOptions(pic_options=pic, sample_options=samples)
  options.trim_reads_for_pileup = bool(rng.random() < 0.2)
  cands, last = [], -100
  positions = sorted(set([0, 1, L - 1, L - 2] + rng.integers(0, L, size=int(rng.integers(3, 30))).tolist()))
  for pos in positions:
    if pos < last + 3 or ref.seq[pos] == 'N':
      continue
    kind = int(rng.integers(0, 4))
    refb = ref.seq[pos]
    if kind == 2 and pos + 6 < L and 'N' not in ref.seq[pos:pos + 6]:
      refb = ref.seq[pos:pos + int(rng.integers(2, 6))]
      alts = [refb[0]] + ([refb[:2]] if len(refb) > 2 and rng.random() < .4 else [])
    elif kind == 3:
      alts = [refb + ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=int(rng.integers(1, 5))))]
      if rng.random() < .4:
        alts.append([b for b in 'ACGT' if b != refb][0])
    else:
      alts = [b for b in 'ACGT' if b != refb][:int(rng.integers(1, 4))]
    near = [r for s in reads for r in _query(s, pos - 8, pos + len(refb) + 8)]
    support = {}
    for a in alts:
      if near and rng.random() < 0.9:
        pick = rng.integers(0, len(near), size=int(rng.integers(0, 14)))
        support[a] = T.SupportingReads(['%s/%d' % (near[int(j)].fragment_name, near[int(j)].read_number) for j in pick])
    if rng.random() < 0.2:
      support['NOT_AN_ALT'] = T.SupportingReads(['nobody/0'])
    v = T.Variant('chr1', pos, pos + len(refb), refb, alts,
                  calls=[T.VariantCall(call_set_name=samples[role_idx].name, genotype=[-1, -1])])
    v.calls[0].info['AD'] = T.ListValue(values=[T.Value(int_value=int(x)) for x in rng.integers(0, 40, size=len(alts) + 1)])
    v.calls[0].info['DP'] = T.ListValue(values=[T.Value(int_value=int(rng.integers(1, 90)))])
    v.calls[0].info['VAF'] = T.ListValue(values=[T.Value(number_value=float(x)) for x in rng.random(len(alts))])
    cands.append(T.DeepVariantCall(variant=v, allele_support=support))
    last = pos + len(refb)
  coverage = [float(rng.integers(0, 80)) for _ in range(n_samples)]
  return dict(options=options, ref=ref, reads=reads, cands=cands, order=order, role=roles[role_idx], coverage=coverage,
              L=L, width=width, channels=channels)


def make_alt_case(seed):
  """--alt_aligned_pileup: haplotype-carrying long reads, every layout, both types_to_alt_align, with and without
  trimming flags of their own (tests/test_hip_region_multisample._alt_region with a seed)."""
  from deepvariant_amd import make_examples_native as men
  from tests.test_hip_region_multisample import _alt_region
  rng = np.random.default_rng(seed)
  mode = ['diff_channels', 'base_channels', 'rows', 'single_row'][int(rng.integers(0, 4))]
  g = _alt_region(mode, ['all', 'indels'][int(rng.integers(0, 2))], bool(rng.random() < 0.3), seed=seed)
  for so in g['options'].sample_options:
    so.order = [0]
    if rng.random() < 0.3:      # the alt-aligned images sample their own (realigned) read lists
      so.use_non_uniform_downsampling = True
      so.non_uniform_downsampling_threshold = int(rng.choice([1, 3, 8]))
      so.pileup_height = int(rng.choice([so.pileup_height, 12, 20]))
  g['options'].pic_options.multi_allelic_mode = int(rng.choice([T.MultiAllelicMode.ADD_HET_ALT_IMAGES,
                                                               T.MultiAllelicMode.NO_HET_ALT_IMAGES]))
  return dict(options=g['options'], ref=g['ref'], reads=[g['reads']], cands=g['cands'], order=[0], role='main',
              coverage=[0.0], L=len(g['ref'].seq), aln_config=men.DEFAULT_ALN_CONFIG)


def run_case(seed, alt=False):
  g = make_alt_case(seed) if alt else make_case(seed)
  refused = None
  try:
    theirs, shape_t = O.reference_write_examples_in_region(g['options'], g['ref'], 'chr1', g['L'], g['cands'], g['reads'],
                                                           g['order'], g['role'], g['coverage'],
                                                           aln_config=g.get('aln_config'))
  except O.OracleError as e:
    if 'Check failed' not in str(e):
      raise
    refused = str(e)
  if refused is not None:
    # an input the reference CHECK-fails on (a read that consumes no reference inside the trimming window): the
    # product has to refuse it too, not draw something
    try:
      RE.product_examples(g['options'], g['ref'], g['cands'], g['reads'], g['order'], g['role'], g['coverage'])
    except ValueError:
      return -1
    raise AssertionError('the reference refuses (%s), the product does not' % refused[-60:])
  mine, shape_m = RE.product_examples(g['options'], g['ref'], g['cands'], g['reads'], g['order'], g['role'], g['coverage'])
  RE.same_examples(mine, theirs, shape_m, shape_t)
  return len(theirs)


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
  alt = len(sys.argv) > 3 and sys.argv[3] == 'alt'
  t, n_examples, n_refused, failures = time.time(), 0, 0, []
  for seed in range(first, first + n):
    try:
      k = run_case(seed, alt)
      n_examples += max(k, 0)
      n_refused += k < 0
    except Exception as e:      # pylint: disable=broad-except
      failures.append(seed)
      print('seed %d: %s: %s' % (seed, type(e).__name__, str(e)[:600]))
  print('%s: %d cases (seeds %d..%d), %d examples compared, %d cases refused by both sides, %d cases differ %s, %.1f s'
        % ('alt-aligned pileups' if alt else 'region options', n, first, first + n - 1, n_examples, n_refused,
           len(failures), failures[:20], time.time() - t))
  return 1 if failures else 0


if __name__ == '__main__':
  sys.exit(main())
