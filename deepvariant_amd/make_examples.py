"""`python -m deepvariant_amd.make_examples` -- calling-mode make_examples for one sample with
the reference's flag names (deepvariant/make_examples.py, make_examples_options.py:71-941):
BAM + FASTA + regions -> `tf.Example` TFRecords (+ `.example_info.json`) that
`deepvariant_amd.call_variants` or the reference's call_variants consume, or -- with
`--call_variants_outfile` and `--checkpoint` -- straight to CallVariantsOutput records without
tf.Examples in between (the reference's precedent: fast_pipeline).

Per calling region it runs make_examples_core.RegionProcessor: reads -> (downsample) -> window
realigner -> device allele counts -> candidate caller -> (read phasing) -> device pileup encoder.
Flags that belong to machinery outside this path (training labels, gVCF, population VCFs,
candidate import, multi-sample roles, sharded runtime profiles ...) are rejected when set,
never silently ignored.  Reads come through the native BAM reader (dv_bam_read_region: plain BAM,
.bai region queries) or, for a CRAM 3.0 file, through deepvariant_amd/cram_reader.py with --ref as
its reference (--use_ref_for_cram, the reference's default).
"""
from __future__ import annotations

import argparse
import re
import sys
from typing import List, Optional, Sequence

import numpy as np

from deepvariant_amd import dv_types as T
from deepvariant_amd import genomics_io
from deepvariant_amd import make_examples_core
from deepvariant_amd import make_examples_native
from deepvariant_amd import packing
from deepvariant_amd import sharded_file_utils
from deepvariant_amd import tfrecord
from deepvariant_amd.realigner import realigner as realigner_module
from deepvariant_amd.realigner import utils

_RANDOM_SEED = 609314161            # make_examples_options.py:981
_CLASSIFY_AT = 256                  # fused route, table path: examples collected on the device per CNN forward
_REGION_BATCH = 32                  # table path: calling regions whose realigner work goes through one native call
_REALIGNER_FLAGS = {k: v for k, v in realigner_module._FLAG_DEFAULTS.items()   # pylint: disable=protected-access
                    if k.startswith(('ws_', 'dbg_', 'aln_')) or k in (
                        'max_num_mismatches', 'realignment_similarity_threshold', 'kmer_size', 'split_skip_reads')}
_REJECTED_IF_SET = (   # flags of the reference whose machinery is outside this path
    'truth_variants', 'confident_regions', 'gvcf', 'candidates', 'proposed_variants', 'population_vcfs',
    'runtime_by_region', 'denovo_regions', 'small_model_path', 'read_phases_output',
    'allele_frequency_vcfs', 'customized_classes_labeler_classes_list', 'pangenome')


def parse_region(literal: str, ref_reader) -> T.Range:
  """ranges.parse_literal: 'chr20:10,000,000-10,010,000' (1-based, inclusive), 'chr20:5' or 'chr20'."""
  m = re.match(r'^([^:]+)(?::([\d,]+)(?:-([\d,]+))?)?$', literal.strip())
  if not m:
    raise ValueError('cannot parse region %r' % literal)
  contig = m.group(1)
  n = ref_reader.n_bases(contig)                    # KeyError for an unknown contig
  if m.group(2) is None:
    return T.Range(contig, 0, n)
  start = int(m.group(2).replace(',', ''))
  end = int(m.group(3).replace(',', '')) if m.group(3) else start
  if start < 1 or end < start:
    raise ValueError('bad region %r' % literal)
  return T.Range(contig, start - 1, min(end, n))


def reservoir_sample(items, k: int, random: np.random.RandomState) -> List:
  """utils.reservoir_sample (third_party/nucleus/util/utils.py:80-125), Algorithm R with numpy's
  RandomState -- the same draws as the reference for the same seed."""
  if k < 0:
    raise ValueError('k must be nonnegative, but got {}'.format(k))
  sample = []
  for i, item in enumerate(items):
    if len(sample) < k:
      sample.append(item)
    else:
      j = random.randint(0, i + 1)
      if j < k:
        sample[j] = item
  return sample


def build_arg_parser() -> argparse.ArgumentParser:
  ap = argparse.ArgumentParser(prog='make_examples', allow_abbrev=False,
                               description='MI355X make_examples (calling mode, one sample)')
  boolean = dict(nargs='?', const='true')
  ap.add_argument('--mode', default='calling')
  ap.add_argument('--ref', required=True)
  ap.add_argument('--reads', required=True)
  ap.add_argument('--examples', default='')
  ap.add_argument('--candidate_positions', default='')     # output of --mode candidate_sweep
  ap.add_argument('--regions', default='')
  ap.add_argument('--exclude_regions', default='')      # space-separated literals, chopped out of the calling regions
  ap.add_argument('--task', type=int, default=0)
  ap.add_argument('--sample_name', default='')
  ap.add_argument('--channel_list', default=','.join(T.PILEUP_DEFAULT_CHANNELS))
  ap.add_argument('--partition_size', type=int, default=1000)
  ap.add_argument('--max_reads_per_partition', type=int, default=1500)
  ap.add_argument('--realign_reads', default='true', **boolean)
  ap.add_argument('--max_read_length_to_realign', type=int, default=500)
  ap.add_argument('--min_mapping_quality', type=int, default=5)
  ap.add_argument('--min_base_quality', type=int, default=10)
  ap.add_argument('--vsc_min_count_snps', type=int, default=2)
  ap.add_argument('--vsc_min_count_indels', type=int, default=2)
  ap.add_argument('--vsc_min_fraction_snps', type=float, default=0.12)
  ap.add_argument('--vsc_min_fraction_indels', type=float, default=0.06)
  ap.add_argument('--pileup_image_width', type=int, default=221)
  ap.add_argument('--pileup_image_height', type=int, default=100)
  # make_examples_options.py:866-883: every allele keeps at least this many of its supporting reads in the image
  ap.add_argument('--use_non_uniform_downsampling', default='false', **boolean)
  ap.add_argument('--non_uniform_downsampling_threshold', type=int, default=3)
  ap.add_argument('--sort_by_haplotypes', default='false', **boolean)
  ap.add_argument('--reverse_haplotypes', default='false', **boolean)
  ap.add_argument('--phase_reads', default='false', **boolean)
  ap.add_argument('--track_ref_reads', default='false', **boolean)
  ap.add_argument('--min_alleles_to_phase', type=int, default=1)
  ap.add_argument('--phase_max_candidates', type=int, default=5000)
  ap.add_argument('--trim_reads_for_pileup', default='false', **boolean)
  ap.add_argument('--alt_aligned_pileup', default='none')
  ap.add_argument('--types_to_alt_align', default='indels')
  ap.add_argument('--parse_sam_aux_fields', default='false', **boolean)     # the HP tag is always read
  ap.add_argument('--keep_duplicates', default='false', **boolean)
  ap.add_argument('--use_ref_for_cram', default='true', **boolean)   # make_examples_options.py:80-89
  ap.add_argument('--use_original_quality_scores', default='false', **boolean)   # qualities from the OQ tag
  ap.add_argument('--keep_supplementary_alignments', default='false', **boolean)
  ap.add_argument('--keep_secondary_alignments', default='false', **boolean)
  ap.add_argument('--keep_legacy_allele_counter_behavior', default='false', **boolean)
  ap.add_argument('--normalize_reads', default='false', **boolean)
  ap.add_argument('--output_phase_info', default='false', **boolean)
  ap.add_argument('--call_small_model_examples', default='false', **boolean)
  ap.add_argument('--stream_examples', default='false', **boolean)
  for name in _REJECTED_IF_SET:
    ap.add_argument('--' + name, default='')
  for name, default in _REALIGNER_FLAGS.items():
    if isinstance(default, bool):
      ap.add_argument('--' + name, default='true' if default else 'false', **boolean)
    elif default is None:
      ap.add_argument('--' + name, default=None)
    else:
      ap.add_argument('--' + name, type=type(default), default=default)
  # --checkpoint / --checkpoint_json: where model.example_info.json (channels + flags_for_calling) is
  # looked up, as in the reference; with --call_variants_outfile (not in the reference) the weights
  # are loaded too and CallVariantsOutput records are written instead of tf.Examples
  ap.add_argument('--checkpoint', default='')
  ap.add_argument('--checkpoint_json', default='')
  ap.add_argument('--call_variants_outfile', default='')
  ap.add_argument('--device', type=int, default=0)
  # not a reference flag: the node-level launcher that stands in for run_deepvariant's N processes
  # (scripts/run_deepvariant.py:457-462) -- N ranks, rank r = task r of --call_variants_outfile name@N,
  # one GPU each, one final all-gather of the CallVariantsOutput records
  ap.add_argument('--gpus', type=int, default=1)
  # host processes per GPU: the region loop up to the packed batch is host work (BAM decode,
  # window realigner, candidate caller); R ranks share one GPU, as N make_examples processes share
  # the reference's call_variants GPU.  world = gpus * ranks_per_gpu, rank r runs on GPU r // R
  ap.add_argument('--ranks_per_gpu', type=int, default=1)
  # fused route: as call_variants' flag of the same name (size of the fixed synthetic calibration set)
  ap.add_argument('--calibration_examples', type=int, default=256)
  return ap


_SMALL_MODEL_FLAGS = ('call_small_model_examples', 'trained_small_model_path', 'small_model_indel_gq_threshold',
                      'small_model_snp_gq_threshold', 'small_model_vaf_context_window_size',
                      'small_model_emit_all_candidates')


def model_example_info_path(checkpoint: str, checkpoint_json: str = '') -> str:
  """get_model_example_info_json_path (make_examples_core.py:3780-3822); '' if there is none
  (a `random:<seed>` or flat-array checkpoint has no json)."""
  import os
  if checkpoint_json:
    return checkpoint_json
  if not checkpoint or checkpoint.startswith('random:'):
    return ''
  model_dir = checkpoint if os.path.exists(os.path.join(checkpoint, 'saved_model.pb')) else os.path.dirname(checkpoint)
  for name in ('model.example_info.json', 'example_info.json'):
    cand = os.path.join(model_dir, name)
    if os.path.exists(cand):
      return cand
  return ''


def apply_flags_for_calling(ap: argparse.ArgumentParser, args, argv: Sequence[str], log=sys.stderr) -> None:
  """apply_flags_for_calling (make_examples_core.py:3825-3905) + the channel list of the model
  (make_examples_options.py:1058-1078): command line > model.example_info.json > defaults.  The
  json's small-model flags are skipped with a note -- the small model is not built, every
  candidate goes to the CNN -- whereas on the command line they are an error."""
  import json
  path = model_example_info_path(args.checkpoint, args.checkpoint_json)
  if not path:
    return
  with open(path) as f:
    info = json.load(f)
  present = set()
  for arg in argv:
    if arg.startswith('--'):
      name = arg[2:].split('=', 1)[0]
      present.add(name)
      if name.startswith('no'):
        present.add(name[2:])
  known = {a.dest: a for a in ap._actions}                       # pylint: disable=protected-access
  for name, value in (info.get('flags_for_calling') or {}).items():
    if name in _SMALL_MODEL_FLAGS:
      print('make_examples: %s from %s skipped (no small model here: all candidates are classified by the CNN)'
            % (name, path), file=log)
      continue
    if name not in known:
      raise ValueError('Flag "%s" (from %s) is not defined as an application flag.' % (name, path))
    if name in present:
      continue
    setattr(args, name, str(value).lower() if isinstance(value, bool) else value)
  if ('partition_size' in present) != ('max_reads_per_partition' in present):
    raise ValueError('Both --partition_size and --max_reads_per_partition must be set together, or not at all.')
  channels = info.get('channels')
  if channels:
    by_enum = {v: k for k, v in T.CHANNEL_NAME_TO_INFO_ENUM.items()}
    names = []
    for c in channels:
      if c not in by_enum:
        raise ValueError('Channel "%s" does not map to an available opt channel' % c)
      names.append(by_enum[c])
    args.model_channels = names


def _true(v) -> bool:
  return str(v).lower() in ('1', 'true', 't', 'yes')


def check_flags(args) -> None:
  if args.mode not in ('calling', 'candidate_sweep'):
    raise ValueError('--mode=%s: calling and candidate_sweep are built (labelling needs the truth-VCF labeler)'
                     % args.mode)
  if args.mode == 'candidate_sweep':
    if not args.candidate_positions:
      raise ValueError('--mode candidate_sweep writes --candidate_positions')
    return
  for name in _REJECTED_IF_SET:
    if getattr(args, name):
      raise ValueError('--%s is not supported by the MI355X make_examples' % name)
  for name in ('normalize_reads', 'stream_examples', 'call_small_model_examples', 'output_phase_info'):
    if _true(getattr(args, name)):
      raise ValueError('--%s is not supported by the MI355X make_examples' % name)
  if args.ws_window_selector_model:
    raise ValueError('--ws_window_selector_model: text-proto window selector models are not parsed here '
                     '(--ws_use_window_selector_model selects the built-in linear model)')
  if args.call_variants_outfile and not args.checkpoint:
    raise ValueError('--call_variants_outfile needs --checkpoint (the fused route)')
  if not args.examples and not args.call_variants_outfile:
    raise ValueError('--examples (or --call_variants_outfile with --checkpoint) is required')
  if args.examples and args.call_variants_outfile:
    raise ValueError('--examples and --call_variants_outfile are two routes (tf.Examples, or CallVariantsOutput '
                     'records straight from the device): give one of them')
  if _true(args.phase_reads) and not _true(args.track_ref_reads):
    raise ValueError('--track_ref_reads must be set to True when --phase_reads is set.')
  if args.partition_size < 1:
    raise ValueError('--partition_size must be positive')
  if args.gpus < 1:
    raise ValueError('--gpus must be positive')
  if args.ranks_per_gpu < 1:
    raise ValueError('--ranks_per_gpu must be positive')
  if args.gpus * args.ranks_per_gpu > 1 and not args.call_variants_outfile:
    raise ValueError('--gpus N shards the fused route (--call_variants_outfile name@N --checkpoint ...); '
                     'for tf.Examples run N independent --task processes, as the reference does')


def _shard(spec: str, task: int):
  """'x.tfrecord@N.gz' + task -> (the task's file name, N); a plain name is one shard
  (sharded_file_utils.resolve_filespecs, make_examples_core.py:3349-3360)."""
  if not sharded_file_utils.is_sharded_file_spec(spec):
    if task != 0:
      raise ValueError('--task=%d needs a sharded output name (name@N)' % task)
    return spec, 0
  n = sharded_file_utils.parse_sharded_file_spec(spec)[1]
  if not 0 <= task < n:
    raise ValueError('task_id={} should be >= 0 and < num_shards={}'.format(task, n))
  return sharded_file_utils.sharded_filename(spec, task), n


def options_from_flags(args):
  """-> (MakeExamplesOptions, RegionProcessorOptions): default_options (make_examples_options.py:944-1380)
  for the flags this slice has."""
  rr = T.ReadRequirements(min_mapping_quality=args.min_mapping_quality, min_base_quality=args.min_base_quality,
                          min_base_quality_mode=1)
  pic = T.default_options(rr)
  alt_mode = args.alt_aligned_pileup
  if getattr(args, 'model_channels', None):        # the model's own list (it names the alt channels itself)
    channels = list(args.model_channels)
  else:
    channels = [c for c in re.split('[, ]+', args.channel_list.replace(
        'BASE_CHANNELS', ','.join(T.PILEUP_DEFAULT_CHANNELS))) if c]
    if alt_mode in ('diff_channels', 'base_channels'):
      channels += ['%s_alternate_allele_1' % alt_mode, '%s_alternate_allele_2' % alt_mode]
  pic.channels = channels
  pic.num_channels = len(channels)
  pic.width, pic.height = args.pileup_image_width, args.pileup_image_height
  pic.sort_by_haplotypes = _true(args.sort_by_haplotypes)
  pic.reverse_haplotypes = _true(args.reverse_haplotypes)
  pic.alt_aligned_pileup = alt_mode
  pic.types_to_alt_align = args.types_to_alt_align
  options = T.MakeExamplesOptions(
      pic_options=pic, trim_reads_for_pileup=_true(args.trim_reads_for_pileup),
      sample_options=[T.SampleOptions(role='main', name=args.sample_name or 'default',
                                      pileup_height=args.pileup_image_height,
                                      use_non_uniform_downsampling=_true(args.use_non_uniform_downsampling),
                                      non_uniform_downsampling_threshold=args.non_uniform_downsampling_threshold)])
  realigner_flags = {}
  for name, default in _REALIGNER_FLAGS.items():
    v = getattr(args, name)
    realigner_flags[name] = _true(v) if isinstance(default, bool) else v
  realigner_flags['keep_legacy_allele_counter_behavior'] = _true(args.keep_legacy_allele_counter_behavior)
  po = make_examples_core.RegionProcessorOptions(
      realigner_enabled=_true(args.realign_reads),
      realigner_options=realigner_module.realigner_config(**realigner_flags),
      max_read_length_to_realign=args.max_read_length_to_realign,
      vsc_min_count_snps=args.vsc_min_count_snps, vsc_min_count_indels=args.vsc_min_count_indels,
      vsc_min_fraction_snps=args.vsc_min_fraction_snps, vsc_min_fraction_indels=args.vsc_min_fraction_indels,
      keep_legacy_allele_counter_behavior=_true(args.keep_legacy_allele_counter_behavior),
      partition_size=args.partition_size, track_ref_reads=_true(args.track_ref_reads),
      phase_reads=_true(args.phase_reads), phase_max_candidates=args.phase_max_candidates,
      min_alleles_to_phase=args.min_alleles_to_phase)
  return options, po


def requested_regions(args, ref_reader, contig_names: Sequence[str]) -> List[T.Range]:
  if args.regions:
    return [parse_region(x, ref_reader) for x in args.regions.split()]     # space-separated literals
  regions = []
  for name in contig_names:
    try:
      regions.append(T.Range(name, 0, ref_reader.n_bases(name)))
    except KeyError:
      continue
  return regions


def calling_regions(args, ref_reader, contig_names: Sequence[str], num_shards: int) -> List[T.Range]:
  """processing_regions_from_options (make_examples_core.py:3380-3445): the contigs the BAM and the
  reference share, in the reference's order, intersected with --regions (overlapping and adjacent
  literals merged, clipped to the contigs, unknown contigs dropped), cut into partition_size pieces;
  this task's share round robin (the examples go to TFRecords)."""
  in_bam = set(contig_names)
  contigs = []
  for name in (ref_reader.contig_names() if hasattr(ref_reader, 'contig_names') else contig_names):
    if name in in_bam:
      try:
        contigs.append((name, ref_reader.n_bases(name)))
      except KeyError:
        continue
  include = requested_regions(args, ref_reader, contig_names) if args.regions else []
  exclude = [parse_region(x, ref_reader) for x in args.exclude_regions.split()]
  calling = make_examples_core.build_calling_regions(contigs, include, exclude)
  if not calling:
    raise ValueError(
        'The regions to call is empty. Check your --regions and --exclude_regions flags to make sure '
        'they are not resulting in set of empty region to process.')        # make_examples_core.py:3424-3431
  pieces = make_examples_core.regions_to_process(
      contigs, args.partition_size, calling_regions=calling,
      task_id=args.task if num_shards else None, num_shards=num_shards if num_shards else None)
  return pieces


class RegionReads:
  """The reads of one calling region, as the reference's per-region `SamReader.query` returns them
  (make_examples_core.py:2518-2560): every read whose alignment overlaps the region, in file order.

  The BAM is decoded natively (dv_bam_read_region: .bai region query, parallel BGZF inflate,
  nucleus' read requirements) in BLOCKS of neighbouring regions, so memory is bounded by one
  block whatever the contig's size, and a region's reads are found by bisection on the sorted
  alignment starts (window widened by the block's longest alignment) instead of a scan."""

  BLOCK_BASES = 1 << 20

  def __init__(self, args, ref_reader=None):
    self._args = args
    self._contig = None
    self._lo = self._hi = 0
    self._make = None
    self._ref = ref_reader      # the runner's FastaReader (CRAM decoding shares it: one reference per process)
    self._reads = {}
    self._starts = np.zeros(0, np.int64)
    self._ends = np.zeros(0, np.int64)
    self._max_span = 0
    # BGZF members of a block inflate on a few host threads (this rank's share of the node's cores)
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    self._threads = max(1, min(8, cores // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))))

  def _load(self, contig: str, lo: int, hi: int) -> None:
    a = self._args
    requirements = dict(
        min_mapping_quality=a.min_mapping_quality, keep_duplicates=_true(a.keep_duplicates),
        keep_supplementary=_true(a.keep_supplementary_alignments), keep_secondary=_true(a.keep_secondary_alignments),
        use_original_quality_scores=_true(a.use_original_quality_scores))
    # resolve_sam_aux_fields (make_examples_core.py:288-373): the aux tags the channel list needs -- MM / ML / MN for the
    # base-modification channels, tp / t0 for the Ultima flow-space channels; the native decoders turn them into
    # per-base planes (csrc/aux_planes.h)
    channels = set(getattr(a, 'model_channels', None) or re.split('[, ]+', getattr(a, 'channel_list', '') or ''))
    requirements['parse_base_modifications'] = bool(channels & {'base_methylation', 'base_6ma'})
    requirements['parse_flow_tags'] = bool(channels & {'homopolymer_insertion_quality', 'homopolymer_deletion_quality',
                                                       'inter_homopolymer_insertion_quality'})
    if genomics_io.is_cram(a.reads):
      # sam_reader.cc:560-640: htslib decodes against --ref (use_ref_for_cram) or the slices' own
      # embedded reference; without either the file cannot be parsed
      fetch = None
      if _true(a.use_ref_for_cram):
        if self._ref is None:
          self._ref = genomics_io.FastaReader(a.ref)
        fetch = self._ref.get_bases
      # native decoder (dv_cram_read_region): the slices of the block decode on this rank's host threads
      table = packing.ReadTable.from_cram(a.reads, fetch, contig, lo, hi, n_threads=self._threads, **requirements)
    else:
      table = packing.ReadTable.from_bam(a.reads, contig, lo, hi, n_threads=self._threads, **requirements)
    # Read objects are built on demand (a task of N touches 1/N of the block's reads) and kept
    # for the neighbouring region, which shares the reads that straddle the boundary
    self._make = table.read_factory(contig)
    self._table = table
    self._reads = {}
    self._starts = table.read_pos.astype(np.int64)
    self._ends = table.read_end.astype(np.int64)
    if self._starts.size > 1 and np.any(np.diff(self._starts) < 0):
      raise ValueError('%s is not coordinate-sorted on %s' % (a.reads, contig))
    self._max_span = int((self._ends - self._starts).max()) if self._starts.size else 0
    self._contig, self._lo, self._hi = contig, lo, hi

  def _rows(self, region: T.Range):
    if region.reference_name != self._contig or region.start < self._lo or region.end > self._hi:
      self._load(region.reference_name, region.start, max(region.end, region.start + self.BLOCK_BASES))
    first = int(np.searchsorted(self._starts, region.start - self._max_span, side='left'))
    last = int(np.searchsorted(self._starts, region.end, side='left'))      # start < region.end
    return first, np.nonzero(self._ends[first:last] > region.start)[0] + first      # end > region.start

  def table(self, region: T.Range) -> 'packing.ReadTable':
    """The same reads as a packed table (rows of the decoded block): what the region chain's table
    path takes (make_examples_core.RegionProcessor.*_table) -- no Read objects at all."""
    _, keep = self._rows(region)
    return self._table.take(keep)

  def __call__(self, region: T.Range) -> list:
    first, keep = self._rows(region)
    out, cache, make = [], self._reads, self._make
    for i in keep.tolist():
      r = cache.get(i)
      if r is None:
        r = cache[i] = make(i)
      out.append(r)
    if first > 0 and len(cache) > 4 * len(out) + 4096:   # rows before this region are never asked for again
      for i in [i for i in cache if i < first]:
        del cache[i]
    return out


class _BackgroundModel:
  """The classifier of the fused route, built on a worker thread (device allocations, the
  checkpoint read, BN folding + fp16 packing + upload: native code, the GIL is free) while the
  region loop decodes its first block of reads and realigns its first batch of regions.  The
  loop only needs the input shape until the first batch of examples is classified."""

  def __init__(self, build, input_shape):
    import threading
    self.input_shape = tuple(input_shape)
    self._model, self._error = None, None
    self._thread = threading.Thread(target=self._run, args=(build,), name='dv-model-setup', daemon=True)
    self._thread.start()

  def _run(self, build):
    try:
      self._model = build()
    except BaseException as e:      # pylint: disable=broad-except   (re-raised on the main thread)
      self._error = e

  def get(self):
    if self._thread is not None:
      self._thread.join()
      self._thread = None
    if self._error is not None:
      raise self._error
    built = getattr(self._model, 'input_shape', None)
    if built is not None and tuple(built) != self.input_shape:
      raise ValueError('model shape %s != example shape %s' % (tuple(built), self.input_shape))
    return self._model

  @property
  def max_batch(self) -> int:
    return self.get().max_batch

  def __call__(self, images):
    return self.get()(images)


class RunnerHooks:
  """What the runner builds per task.  The product uses these defaults; the multi-process CPU
  tests substitute a processor / model that need no GPU (tests/test_make_examples_dist_cpu.py)."""

  def make_processor(self, options, ref_reader, po, device):
    return make_examples_core.RegionProcessor(options, ref_reader, po, device=device)

  def make_model(self, args, options):
    from deepvariant_amd import call_variants
    from deepvariant_amd.inception_v3 import InceptionV3
    shape = (make_examples_native.calculate_pileup_image_height(options), options.pic_options.width,
             len(options.pic_options.channels))
    # activations are allocated for max_batch examples (6 MB each); R processes sharing a GPU
    # each own a model, and a 1 kb region rarely yields more than a few dozen examples
    model = InceptionV3(shape, max_batch=512 if args.ranks_per_gpu == 1 else 256, device=args.device)
    call_variants.load_flat_checkpoint(args.checkpoint, model)
    # the shifts are calibrated on the fixed synthetic set of this input shape: the same corrections on every rank
    # (computed, or read from the cache next to the checkpoint), whatever the rank layout and the regions a rank gets
    model.calibrate_for_checkpoint(getattr(args, 'calibration_examples', 256),
                                   cache_prefix=call_variants.calibration_cache_prefix(args.checkpoint))
    return model


def make_examples_runner(args, log=sys.stderr, hooks: Optional[RunnerHooks] = None, sink=None) -> dict:
  """-> stats; writes the task's example shard (and its example_info.json) or the CVO file.
  `sink` (write(bytes), close()) replaces the task's TFRecord file: the multi-GPU driver keeps
  a rank's records in memory for the final gather."""
  import time
  t_start = time.perf_counter()
  hooks = hooks or RunnerHooks()
  check_flags(args)
  ref_reader = genomics_io.FastaReader(args.ref)
  options, po = options_from_flags(args)
  sweep = args.mode == 'candidate_sweep'
  # the output follows the active route: the fused route writes CallVariantsOutput records to
  # --call_variants_outfile, the example route tf.Examples to --examples (check_flags refuses both)
  out_spec = args.candidate_positions if sweep else (args.call_variants_outfile or args.examples)
  out_path, num_shards = _shard(out_spec, args.task)
  contig_names = genomics_io.bam_contig_names(args.reads)
  pieces = calling_regions(args, ref_reader, contig_names, num_shards)
  reads_for = RegionReads(args, ref_reader)
  proc = hooks.make_processor(options, ref_reader, po, args.device)
  model = None
  if args.call_variants_outfile:
    shape = (make_examples_native.calculate_pileup_image_height(options), options.pic_options.width,
             len(options.pic_options.channels))
    model = _BackgroundModel(lambda: hooks.make_model(args, options), shape)
  stats = dict(n_regions=0, n_reads=0, n_candidates=0, n_examples=0)
  if sweep:
    # int32 positions per calling region, END_OF_PARTITION after each, END_OF_REGION where a
    # requested region ends (make_examples_core.py:3592-3605) -- input of a later run's
    # partitioning by candidates
    region_ends = {(r.reference_name, r.end) for r in requested_regions(args, ref_reader, contig_names)}
    with open(out_path, 'wb') as f:
      for region in pieces:
        in_reads = reads_for(region)
        if args.max_reads_per_partition > 0:
          in_reads = reservoir_sample(in_reads, args.max_reads_per_partition, np.random.RandomState(_RANDOM_SEED))
        positions = proc.find_candidate_positions(region, in_reads)
        if (region.reference_name, region.end) in region_ends:
          positions = positions + [make_examples_core.END_OF_REGION]
        f.write(np.array(positions, np.int32).tobytes())
        stats['n_regions'] += 1
        stats['n_reads'] += len(in_reads)
        stats['n_candidates'] += sum(p >= 0 for p in positions)
    print('make_examples task %d: %d regions, %d reads, %d candidate positions -> %s' % (
        args.task, stats['n_regions'], stats['n_reads'], stats['n_candidates'], out_path), file=log)
    return stats
  writer = sink if sink is not None else tfrecord.Writer(out_path)
  image_shape = None
  # the table path (no Read objects between the BAM decoder and the encoder) where the
  # configuration allows it; DV_REGION_OBJECTS=1 forces the object path (A/B, tests)
  import os
  use_tables = os.environ.get('DV_REGION_OBJECTS') is None and getattr(proc, 'table_path_ok', lambda: False)()
  stats['table_path'] = bool(use_tables)
  t_loop = time.perf_counter()
  stats['setup_s'] = t_loop - t_start        # flags, region list, processor (the model is set up on a worker thread)
  try:
    # the table path walks the regions in batches of _REGION_BATCH: the realigner's native work of a
    # whole batch (every window's assembly and alignment) is ONE threaded call, handed to a worker
    # thread for batch k + 1 before the candidates of batch k are called, drawn and classified on
    # the main thread -- the native call holds no Python lock.  (Moving the REST of a batch's
    # preparation -- rows, window selection, write-back -- to the worker as well was measured and
    # is slower: those are Python / numpy steps that contend for the interpreter lock, 900-918
    # against 973-997 examples/s on the NA12878 slice, same box.)
    def batch_tables(at):
      batch = pieces[at:at + _REGION_BATCH]
      tables = []
      for region in batch:
        in_table = reads_for.table(region)
        if 0 < args.max_reads_per_partition < in_table.n_reads:     # the same draws on row numbers
          in_table = in_table.take(np.array(reservoir_sample(range(in_table.n_reads), args.max_reads_per_partition,
                                                             np.random.RandomState(_RANDOM_SEED)), np.int64))
        tables.append(in_table)
      return batch, tables

    worker = None
    if use_tables and len(pieces) > _REGION_BATCH:
      import concurrent.futures
      worker = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='dv-realign-batch')

    def start_batch(at):                 # -> a callable that returns (batch, tables, realigned tables)
      batch, tables = batch_tables(at)
      finish = proc.start_realign_tables(tables, batch, executor=worker)
      return lambda: (batch, tables, finish())

    pending = start_batch(0) if use_tables and pieces else None
    for at in range(0, len(pieces) if use_tables else 0, _REGION_BATCH):
      t_wait = time.perf_counter()
      batch, tables, realigned_tables = pending()
      stats['wait_for_prepared_batches_s'] = stats.get('wait_for_prepared_batches_s', 0.0) + time.perf_counter() - t_wait
      pending = start_batch(at + _REGION_BATCH) if at + _REGION_BATCH < len(pieces) else None
      called = proc.process_tables(batch, tables, realigned_tables)       # the batch's allele counts: one device call
      for region, in_table, (candidates, realigned) in zip(batch, tables, called):
        stats['n_regions'] += 1
        stats['n_reads'] += in_table.n_reads
        stats['n_candidates'] += len(candidates)
        if model is not None:
          # drawn on the device now, classified with the regions around it (one CNN forward per
          # _CLASSIFY_AT examples); records leave in region order
          proc.queue_region_candidates(candidates, realigned, model)
          if proc.n_queued_examples >= _CLASSIFY_AT:
            for records in proc.flush_queue(model):
              for rec in records:
                writer.write(rec)
              stats['n_examples'] += len(records)
          continue
        _, records = proc.examples_in_region_table(region, in_table, called=(candidates, realigned))
        for rec in records:
          writer.write(rec)
        stats['n_examples'] += len(records)
    if worker is not None:
      worker.shutdown()
    for region in ([] if use_tables else pieces):
      in_reads = reads_for(region)
      if args.max_reads_per_partition > 0:
        in_reads = reservoir_sample(in_reads, args.max_reads_per_partition, np.random.RandomState(_RANDOM_SEED))
      stats['n_regions'] += 1
      stats['n_reads'] += len(in_reads)
      if model is not None:
        candidates, records = proc.call_variants_in_region(region, in_reads, model.get())
      else:
        candidates, records = proc.examples_in_region(region, in_reads)
      for rec in records:
        writer.write(rec)
      stats['n_candidates'] += len(candidates)
      stats['n_examples'] += len(records)
    if use_tables and model is not None:
      for records in proc.flush_queue(model):
        for rec in records:
          writer.write(rec)
        stats['n_examples'] += len(records)
    if model is not None:
      model.get()                 # a failed model set-up is an error even when no example asked for it
  finally:
    writer.close()
  stats['loop_s'] = time.perf_counter() - t_loop   # the region loop proper (BAM decode to the last record written)
  if model is None:
    pic = options.pic_options
    image_shape = [make_examples_native.calculate_pileup_image_height(options), pic.width, len(pic.channels)]
    if args.task == 0:
      make_examples_native.write_example_info_json(out_path, image_shape, pic.channels)
  print('make_examples task %d: %d regions, %d reads, %d candidates, %d %s -> %s' % (
      args.task, stats['n_regions'], stats['n_reads'], stats['n_candidates'], stats['n_examples'],
      'CallVariantsOutputs' if model is not None else 'examples', out_path), file=log)
  return stats


class _MemorySink:
  def __init__(self):
    self.records: List[bytes] = []

  def write(self, rec: bytes) -> None:
    self.records.append(rec)

  def close(self) -> None:
    pass


def distributed_runner(args, rank: int, world: int, log=sys.stderr, hooks: Optional[RunnerHooks] = None) -> dict:
  """One rank of `--gpus N`: the node-level form of what the reference runs as N make_examples
  processes + N call_variants writer shards glued by files (scripts/run_deepvariant.py:457-462,
  deepvariant/call_variants.py:934-951).  Rank r IS task r of N (regions i % N == r,
  make_examples_core.py:879-888), runs the fused route on its own GPU with no inter-GPU traffic,
  and the CallVariantsOutput records are exchanged once at the end (one all-gather: RCCL over
  xGMI on GPUs, gloo in the CPU tests); rank 0 writes the N shard files, each byte-identical to
  what an independent `--task r` run writes.  The process group must be initialised."""
  import torch
  from deepvariant_amd import dist as dvd
  spec = args.call_variants_outfile
  if not sharded_file_utils.is_sharded_file_spec(spec) or sharded_file_utils.parse_sharded_file_spec(spec)[1] != world:
    raise ValueError('--gpus x --ranks_per_gpu = %d ranks need --call_variants_outfile name@%d (one shard per rank)'
                     % (world, world))
  import torch.distributed as dist
  args.task = rank
  on_gpu = torch.cuda.is_available()
  if on_gpu:
    args.device = (rank // max(getattr(args, 'ranks_per_gpu', 1), 1)) % torch.cuda.device_count()
    torch.cuda.set_device(args.device)
  # the ranks of a node share its cores: each rank's realigner pool gets its share (16 threads per
  # rank, the single-process default, would oversubscribe a node that runs 8 or 16 ranks)
  import os
  if os.environ.get('DV_REALIGN_THREADS') is None:
    from deepvariant_amd.realigner import realigner as realigner_module
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    on_node = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    realigner_module._NATIVE_THREADS = max(1, min(16, cores // on_node))      # pylint: disable=protected-access
  sink = _MemorySink()
  # RCCL wants one rank per GPU; ranks that share a GPU exchange their (host) records over gloo
  device = torch.device('cuda', args.device) if on_gpu and dist.get_backend() == 'nccl' else None
  import time
  t_runner = time.perf_counter()
  try:
    stats = make_examples_runner(args, log=log, hooks=hooks, sink=sink)
  except BaseException as err:
    # tell the peers before leaving: they would otherwise sit in the all-gather until the backend's timeout.
    # The notification is itself a collective: it is skipped when there is no process group (any more) or
    # when the failure CAME from a collective (a second, mismatched all-gather could hang instead of
    # exiting), and nothing it raises may replace the original error -- the `raise` below always runs.
    notify = dist.is_available() and dist.is_initialized() and not isinstance(err, dvd.PeerFailed) and \
        not getattr(err, 'from_collective', False) and 'ProcessGroup' not in type(err).__name__ and \
        'DistBackendError' not in type(err).__name__
    if notify:
      try:
        dvd.gather_records([], device=device, failed=True)
      except BaseException:      # pylint: disable=broad-except
        pass
    raise
  t_gather = time.perf_counter()
  per_rank = dvd.gather_records(sink.records, device=device)
  t_write = time.perf_counter()
  if rank == 0:
    for r, records in enumerate(per_rank):
      writer = tfrecord.Writer(sharded_file_utils.sharded_filename(spec, r))
      try:
        for rec in records:
          writer.write(rec)
      finally:
        writer.close()
    print('make_examples --gpus %d: %s CallVariantsOutputs gathered from %d ranks -> %s' % (
        world, '+'.join(str(len(x)) for x in per_rank), world, spec), file=log)
  stats['n_gathered'] = sum(len(x) for x in per_rank)
  # where this rank's time went: its own region loop (incl. set-up), the record exchange (which also waits for
  # the slowest rank), rank 0's shard files
  stats['runner_s'], stats['gather_s'], stats['write_s'] = t_gather - t_runner, t_write - t_gather, time.perf_counter() - t_write
  return stats


def _backend(args) -> str:
  import torch
  return 'nccl' if torch.cuda.is_available() and args.ranks_per_gpu == 1 else 'gloo'


def _spawned_rank(rank: int, args, world: int, port: int) -> None:
  import os
  import torch.distributed as dist
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group(_backend(args), rank=rank, world_size=world)
  try:
    distributed_runner(args, rank, world)
  finally:
    dist.destroy_process_group()


def run_multi_gpu(args) -> None:
  """`--gpus N`: under a launcher (torch.distributed.run: RANK / WORLD_SIZE set) this process is
  one rank; otherwise the N ranks are spawned here, one process per GPU, 127.0.0.1 rendezvous."""
  import os
  import socket
  import torch
  import torch.distributed as dist
  world = args.gpus * args.ranks_per_gpu
  if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    if int(os.environ['WORLD_SIZE']) != world:
      raise ValueError('--gpus x --ranks_per_gpu must equal WORLD_SIZE')
    dist.init_process_group(_backend(args))
    try:
      distributed_runner(args, dist.get_rank(), world)
    finally:
      dist.destroy_process_group()
    return
  if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
    raise ValueError('--gpus %d but only %d GPU(s) are visible' % (args.gpus, torch.cuda.device_count()))
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  import torch.multiprocessing as mp
  mp.spawn(_spawned_rank, args=(args, world, port), nprocs=world, join=True)


def absl_booleans(ap: argparse.ArgumentParser, argv: Sequence[str]) -> List[str]:
  """absl's `--noflag` spelling of `--flag=false` (scripts/run_deepvariant.py passes
  `--norealign_reads`), for the flags that are boolean here."""
  booleans = {a.dest for a in ap._actions if a.nargs == '?' and a.const == 'true'}   # pylint: disable=protected-access
  out = []
  for arg in argv:
    if arg.startswith('--no') and '=' not in arg and arg[4:] in booleans:
      out.append('--%s=false' % arg[4:])
    else:
      out.append(arg)
  return out


def main(argv=None) -> int:
  ap = build_arg_parser()
  argv = absl_booleans(ap, sys.argv[1:] if argv is None else argv)
  args = ap.parse_args(argv)
  try:
    apply_flags_for_calling(ap, args, argv)
    if args.gpus * args.ranks_per_gpu > 1:
      check_flags(args)
      run_multi_gpu(args)
    else:
      make_examples_runner(args)
  except (ValueError, KeyError, IOError, OSError) as e:
    print('make_examples: %s' % e, file=sys.stderr)
    return 1
  except Exception as e:  # pylint: disable=broad-except
    # a spawned rank's ValueError / IOError comes back wrapped (torch.multiprocessing.ProcessRaisedException):
    # same message and exit code as the single-process route, not a traceback
    if type(e).__name__ not in ('ProcessRaisedException', 'ProcessExitedException'):
      raise
    lines = [ln for ln in str(e).strip().splitlines() if ln.strip()]
    print('make_examples: %s' % (lines[-1] if lines else e), file=sys.stderr)
    return 1
  return 0


if __name__ == '__main__':
  sys.exit(main())
