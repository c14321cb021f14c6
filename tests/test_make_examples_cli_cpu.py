"""`python -m deepvariant_amd.make_examples`: flag handling, region parsing, sharding and the
downsampler (host side; the run itself needs the device and lives in tests/test_hip_pipeline.py)."""
import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import genomics_io
from deepvariant_amd import make_examples as me
from tests import realigner_fixture as RF


def parse(*argv):
  return me.build_arg_parser().parse_args(['--ref', 'r.fa', '--reads', 'x.bam'] + list(argv))


def test_defaults_are_the_references():
  args = parse('--examples', 'e.tfrecord.gz')
  me.check_flags(args)
  options, po = me.options_from_flags(args)
  pic = options.pic_options
  assert pic.channels == list(T.PILEUP_DEFAULT_CHANNELS) and (pic.height, pic.width) == (100, 221)
  assert (pic.read_requirements.min_mapping_quality, pic.read_requirements.min_base_quality) == (5, 10)
  assert (po.realigner_enabled, po.partition_size, po.max_read_length_to_realign) == (True, 1000, 500)
  assert (po.vsc_min_count_snps, po.vsc_min_count_indels, po.vsc_min_fraction_snps, po.vsc_min_fraction_indels) == (
      2, 2, 0.12, 0.06)
  assert not po.phase_reads and not po.track_ref_reads
  ws = po.realigner_options.ws_config
  assert (ws.min_mapq, ws.min_windows_distance, ws.window_selector_model.variant_reads_model.min_num_supporting_reads) == (
      20, 80, 2)


def test_long_read_flags():
  args = parse('--examples', 'e.tfrecord.gz', '--norealign' if False else '--realign_reads=false', '--phase_reads',
               '--track_ref_reads', '--sort_by_haplotypes', '--trim_reads_for_pileup', '--alt_aligned_pileup',
               'diff_channels', '--pileup_image_width', '147', '--min_mapping_quality', '1',
               '--vsc_min_fraction_indels', '0.12', '--partition_size', '25000',
               '--channel_list', ','.join(T.PILEUP_DEFAULT_CHANNELS + ['haplotype', 'base_methylation']))
  me.check_flags(args)
  options, po = me.options_from_flags(args)
  pic = options.pic_options
  assert len(pic.channels) == 10 and pic.channels[-2:] == ['diff_channels_alternate_allele_1',
                                                           'diff_channels_alternate_allele_2']
  assert pic.width == 147 and pic.sort_by_haplotypes and options.trim_reads_for_pileup
  assert not po.realigner_enabled and po.phase_reads and po.track_ref_reads and po.partition_size == 25000


@pytest.mark.parametrize('argv,message', [
    (['--examples', 'e', '--mode', 'training'], 'calling and candidate_sweep are built'),
    (['--mode', 'candidate_sweep'], 'writes --candidate_positions'),
    (['--examples', 'e', '--gvcf', 'g.tfrecord.gz'], '--gvcf is not supported'),
    (['--examples', 'e', '--truth_variants', 't.vcf.gz'], '--truth_variants is not supported'),
    (['--examples', 'e', '--population_vcfs', 'p.vcf.gz'], '--population_vcfs is not supported'),
    (['--examples', 'e', '--normalize_reads'], '--normalize_reads is not supported'),
    (['--examples', 'e', '--stream_examples'], '--stream_examples is not supported'),
    (['--examples', 'e', '--phase_reads'], 'track_ref_reads must be set'),
    (['--examples', 'e', '--ws_window_selector_model', 'm.pbtxt'], 'not parsed here'),
    (['--call_variants_outfile', 'cvo.tfrecord.gz'], 'needs --checkpoint'),
    ([], '--examples'),
])
def test_unsupported_flags_are_refused(argv, message):
  with pytest.raises(ValueError, match=message):
    me.check_flags(parse(*argv))


def test_absl_negated_booleans():
  ap = me.build_arg_parser()
  argv = me.absl_booleans(ap, ['--ref', 'r', '--reads', 'b', '--examples', 'e', '--norealign_reads', '--phase_reads',
                               '--notrim_reads_for_pileup', '--nonsense'])
  assert argv[-4:] == ['--realign_reads=false', '--phase_reads', '--trim_reads_for_pileup=false', '--nonsense']
  args = ap.parse_args(argv[:-1] + ['--track_ref_reads'])
  _, po = me.options_from_flags(args)
  assert not po.realigner_enabled and po.phase_reads


def test_unknown_flags_are_refused():
  with pytest.raises(SystemExit):
    parse('--examples', 'e', '--no_such_flag', '1')


def test_parse_region():
  ref = RF.StringRef('chr20', 'A' * 5000)
  assert me.parse_region('chr20:1,001-2,000', ref) == T.Range('chr20', 1000, 2000)
  assert me.parse_region('chr20:10', ref) == T.Range('chr20', 9, 10)
  assert me.parse_region('chr20', ref) == T.Range('chr20', 0, 5000)
  assert me.parse_region('chr20:4,000-9,000', ref) == T.Range('chr20', 3999, 5000)     # clipped to the contig
  with pytest.raises(KeyError):
    me.parse_region('chrX:1-5', ref)
  with pytest.raises(ValueError):
    me.parse_region('chr20:5-1', ref)


def test_calling_regions_are_shared_round_robin():
  ref = RF.StringRef('chr20', 'A' * 10500)
  args = parse('--examples', 'e@3', '--regions', 'chr20:1-4,000 chr20:8,001-10,500', '--task', '1')
  pieces = me.calling_regions(args, ref, ['chr20'], 3)
  every = me.calling_regions(args, ref, ['chr20'], 0)
  assert [(p.start, p.end) for p in every] == [(0, 1000), (1000, 2000), (2000, 3000), (3000, 4000), (8000, 9000),
                                               (9000, 10000), (10000, 10500)]
  assert pieces == every[1::3]
  assert me._shard('x.tfrecord@3.gz', 1) == ('x.tfrecord-00001-of-00003.gz', 3)
  assert me._shard('x.tfrecord.gz', 0) == ('x.tfrecord.gz', 0)
  with pytest.raises(ValueError):
    me._shard('x.tfrecord@3.gz', 3)
  # no --regions: every contig of the BAM that the reference has
  args = parse('--examples', 'e', '--partition_size', '6000')
  assert [(p.reference_name, p.start, p.end) for p in me.calling_regions(args, ref, ['chrM', 'chr20'], 0)] == [
      ('chr20', 0, 6000), ('chr20', 6000, 10500)]


def test_regions_are_merged_clipped_and_excluded_like_the_reference():
  """build_calling_regions + regions_to_process (calling_regions_utils.py:48-98,
  make_examples_core.py:800-888): overlapping / adjacent literals merge, a literal past the contig
  end is clipped, --exclude_regions is chopped out, nothing left is an error."""
  ref = RF.StringRef('chr20', 'A' * 10500)
  args = parse('--examples', 'e', '--regions', 'chr20:3,001-5,000 chr20:1-3,000 chr20:9,001-99,000',
               '--exclude_regions', 'chr20:2,501-2,600 chr20:10,001-10,500', '--partition_size', '2000')
  assert [(p.start, p.end) for p in me.calling_regions(args, ref, ['chr20'], 0)] == [
      (0, 2000), (2000, 2500), (2600, 4600), (4600, 5000), (9000, 10000)]
  args = parse('--examples', 'e', '--regions', 'chr20:1-100', '--exclude_regions', 'chr20')
  with pytest.raises(ValueError, match='regions to call is empty'):
    me.calling_regions(args, ref, ['chr20'], 0)


def test_reservoir_sample():
  """nucleus utils_test.py:96-132: lengths, bad k, and uniform frequencies with the seeded RandomState."""
  rs = np.random.RandomState(123456789)
  assert len(me.reservoir_sample(range(10), 11, rs)) == 10
  assert me.reservoir_sample(range(10), 10, rs) == list(range(10))
  assert len(me.reservoir_sample(range(10), 9, rs)) == 9
  assert me.reservoir_sample(range(10), 0, rs) == []
  with pytest.raises(ValueError):
    me.reservoir_sample(range(10), -1, rs)
  for n, k in ((10, 1), (6, 3), (10, 3)):
    counts = np.zeros(n)
    for _ in range(20000):
      for item in me.reservoir_sample(range(n), k, rs):
        counts[item] += 1
    np.testing.assert_allclose(counts / 20000, min(k / n, 1.0), atol=0.015)
  # the draws are numpy's RandomState.randint(0, i + 1): pinned for the make_examples seed
  assert me.reservoir_sample(range(12), 4, np.random.RandomState(609314161)) == me.reservoir_sample(
      range(12), 4, np.random.RandomState(609314161))


def test_bam_round_trip(tmp_path):
  """write_bam (the way fixtures get back to a file) -> both BAM readers give the reads back."""
  from deepvariant_amd import packing
  _, sets = RF.load()
  reads = sets['ex1']
  path = str(tmp_path / 'ex1.bam')
  genomics_io.write_bam(path, [('chrM', 16571), ('chr20', 63025520)], reads, sample_name='NA12878')
  assert genomics_io.bam_contig_names(path) == ['chrM', 'chr20']
  _, back = genomics_io.read_bam(path, 'chr20', 0, 1 << 40)
  key = lambda r: (r.alignment.position.position, r.fragment_name, r.read_number)
  for a, b in zip(sorted(reads, key=key), sorted(back, key=key)):
    assert (a.fragment_name, a.read_number, a.aligned_sequence, bytes(bytearray(a.aligned_quality)), a.alignment,
            a.fragment_length) == (b.fragment_name, b.read_number, b.aligned_sequence,
                                   bytes(bytearray(b.aligned_quality)), b.alignment, b.fragment_length)
  table = packing.ReadTable.from_bam(path, 'chr20', 0, 1 << 40)
  assert table.n_reads == len(reads) == 116


PACBIO_MODEL_JSON = {     # deepvariant/json/deepvariant.pacbio.savedmodel/model.example_info.json
    'version': '1.10.0', 'shape': [100, 147, 10], 'channels': [1, 2, 3, 4, 5, 6, 7, 26, 9, 10],
    'flags_for_calling': {
        'alt_aligned_pileup': 'diff_channels', 'call_small_model_examples': True,
        'keep_supplementary_alignments': True, 'max_reads_per_partition': 600, 'min_mapping_quality': 1,
        'parse_sam_aux_fields': True, 'partition_size': 25000, 'phase_reads': True, 'pileup_image_height': 100,
        'pileup_image_width': 147, 'realign_reads': False, 'small_model_indel_gq_threshold': 16,
        'small_model_snp_gq_threshold': 15, 'small_model_vaf_context_window_size': 51, 'sort_by_haplotypes': True,
        'track_ref_reads': True, 'trained_small_model_path': '/opt/smallmodels/pacbio',
        'trim_reads_for_pileup': True, 'vsc_min_fraction_indels': 0.12}}


def test_flags_for_calling_come_from_the_model_json(tmp_path):
  """apply_flags_for_calling (make_examples_core.py:3825-3905): command line > model.example_info.json >
  defaults; the channel list is the model's; the released PacBio model's json gives its 10-channel tensor."""
  import io
  import json
  model_dir = tmp_path / 'model'
  model_dir.mkdir()
  (model_dir / 'saved_model.pb').write_bytes(b'')
  (model_dir / 'model.example_info.json').write_text(json.dumps(PACBIO_MODEL_JSON))
  ap = me.build_arg_parser()
  argv = ['--ref', 'r', '--reads', 'b', '--examples', 'e', '--checkpoint', str(model_dir), '--vsc_min_fraction_indels',
          '0.2']
  args = ap.parse_args(argv)
  log = io.StringIO()
  me.apply_flags_for_calling(ap, args, argv, log=log)
  me.check_flags(args)
  options, po = me.options_from_flags(args)
  pic = options.pic_options
  assert [T.CHANNEL_NAME_TO_INFO_ENUM[c] for c in pic.channels] == PACBIO_MODEL_JSON['channels']
  assert (pic.width, pic.height, pic.alt_aligned_pileup, pic.sort_by_haplotypes) == (147, 100, 'diff_channels', True)
  assert pic.read_requirements.min_mapping_quality == 1 and options.trim_reads_for_pileup
  assert (po.realigner_enabled, po.phase_reads, po.track_ref_reads, po.partition_size) == (False, True, True, 25000)
  assert args.max_reads_per_partition == 600 and me._true(args.keep_supplementary_alignments)
  assert po.vsc_min_fraction_indels == 0.2                      # the command line wins over the json's 0.12
  assert 'call_small_model_examples' in log.getvalue()          # skipped with a note, not silently
  # partition_size and max_reads_per_partition only together on the command line
  argv = argv + ['--partition_size', '1000']
  with pytest.raises(ValueError, match='must be set together'):
    me.apply_flags_for_calling(ap, ap.parse_args(argv), argv, log=log)
  # a json flag this program does not have at all is an error, as in the reference
  bad = dict(PACBIO_MODEL_JSON, flags_for_calling={'no_such_flag': 1})
  (model_dir / 'model.example_info.json').write_text(json.dumps(bad))
  argv = ['--ref', 'r', '--reads', 'b', '--examples', 'e', '--checkpoint', str(model_dir)]
  with pytest.raises(ValueError, match='not defined as an application flag'):
    me.apply_flags_for_calling(ap, ap.parse_args(argv), argv, log=log)
  # --checkpoint_json overrides the lookup; random:<seed> checkpoints have no json
  assert me.model_example_info_path('random:1') == ''
  assert me.model_example_info_path(str(model_dir), 'x.json') == 'x.json'
  assert me.model_example_info_path(str(model_dir / 'ckpt-1')) == str(model_dir / 'model.example_info.json')
  assert me.model_example_info_path(str(tmp_path / 'elsewhere' / 'ckpt-1')) == ''


def test_native_table_to_reads(tmp_path):
  """packing.ReadTable.from_bam(...).to_reads(): the native reader's table turned back into Read
  objects equals the Python BAM reader + read requirements, field by field (what the CLI feeds
  the region chain)."""
  from deepvariant_amd import packing
  _, sets = RF.load()
  reads = sets['ex2'] + sets['dbg0']
  reads[3].info['HP'] = T.ListValue(values=[T.Value(int_value=2)])
  path = str(tmp_path / 'r.bam')
  genomics_io.write_bam(path, [('chr20', 63025520)], reads)
  for lo, hi, mapq in ((0, 1 << 40, 0), (10_046_100, 10_046_200, 30)):
    got = packing.ReadTable.from_bam(path, 'chr20', lo, hi, min_mapping_quality=mapq).to_reads('chr20')
    _, want = genomics_io.read_bam(path, 'chr20', lo, hi)
    want = [r for r in want if genomics_io.read_satisfies_requirements(r, min_mapping_quality=mapq)]
    assert len(got) == len(want) > 20
    for a, b in zip(got, want):
      assert (a.fragment_name, a.read_number, a.aligned_sequence, bytes(a.aligned_quality), a.alignment,
              a.fragment_length, a.supplementary_alignment, {k: v.values[0].int_value for k, v in a.info.items()}) == (
                  b.fragment_name, b.read_number, b.aligned_sequence, bytes(bytearray(b.aligned_quality)), b.alignment,
                  b.fragment_length, b.supplementary_alignment, {k: v.values[0].int_value for k, v in b.info.items()})


def test_region_reads_equal_a_linear_scan(tmp_path, monkeypatch):
  """RegionReads (block-wise native decode + bisection on the sorted starts) returns, per calling
  region, exactly the reads a scan over the whole contig keeps -- also across block reloads."""
  from deepvariant_amd import packing
  _, sets = RF.load()
  reads = sorted(sets['ex2'] + sets['dbg0'] + sets['ex1'], key=lambda r: r.alignment.position.position)
  path = str(tmp_path / 'r.bam')
  genomics_io.write_bam(path, [('chr20', 63025520)], reads)
  args = me.build_arg_parser().parse_args(['--ref', 'x', '--reads', path, '--examples', 'e'])
  monkeypatch.setattr(me.RegionReads, 'BLOCK_BASES', 700)     # several reloads over the span
  table = packing.ReadTable.from_bam(path, 'chr20', 0, 1 << 40, min_mapping_quality=args.min_mapping_quality)
  every = table.to_reads('chr20')
  lo = min(r.alignment.position.position for r in every) - 50
  hi = max(int(e) for e in table.read_end) + 50
  reads_for = me.RegionReads(args)
  n_total = 0
  for start in range(lo, hi, 300):
    region = T.Range('chr20', start, start + 300)
    want = [r for r, s, e in zip(every, table.read_pos.tolist(), table.read_end.tolist())
            if s < region.end and e > region.start]
    got = reads_for(region)
    assert [(r.fragment_name, r.read_number, r.alignment.position.position) for r in got] == [
        (r.fragment_name, r.read_number, r.alignment.position.position) for r in want]
    n_total += len(got)
  assert n_total > len(every)            # reads that straddle region borders are returned by both sides
  # Read objects are built on demand and shared by neighbouring regions; rows behind the current
  # region are dropped from the cache.  Order of the calls must not matter: every second region
  # (a task of two), then the same regions backwards, give the same reads, and a read returned for
  # two regions in a row is ONE object (its packed record travels with it).
  def names(rs):
    return [(r.fragment_name, r.read_number, r.alignment.position.position) for r in rs]
  starts = list(range(lo, hi, 300))
  forward = {s0: names(reads_for(T.Range('chr20', s0, s0 + 300))) for s0 in starts[::2]}
  backward = {s0: names(reads_for(T.Range('chr20', s0, s0 + 300))) for s0 in reversed(starts[::2])}
  assert forward == backward
  straddled = [k for k in range(len(starts) - 1)
               if any(s < starts[k + 1] < e for s, e in zip(table.read_pos.tolist(), table.read_end.tolist()))]
  k = straddled[len(straddled) // 2]
  a = reads_for(T.Range('chr20', starts[k], starts[k] + 300))          # (re)loads a block that holds both regions
  b = reads_for(T.Range('chr20', starts[k + 1], starts[k + 1] + 300))
  shared = {id(r) for r in a} & {id(r) for r in b}
  assert shared and all(hasattr(r, '_dv_packed') for r in a)


def test_the_two_output_routes_exclude_each_other():
  args = me.build_arg_parser().parse_args(['--ref', 'x', '--reads', 'y', '--examples', 'e.tfrecord',
                                           '--call_variants_outfile', 'c.tfrecord', '--checkpoint', 'random:1'])
  with pytest.raises(ValueError, match='two routes'):
    me.check_flags(args)
