"""One region of make_examples, single sample: reads -> window realigner -> allele counts ->
candidate calls -> pileup examples (or, fused, genotype probabilities).  The part of
deepvariant/make_examples_core.py's RegionProcessor that joins the hot path's stages:

  RegionProcessor.realign_reads          make_examples_core.py:2479-2518
  RegionProcessor.candidates_in_region   make_examples_core.py:2840-3165 (one sample; incl. the
                                         two-pass track_ref_reads counting and read phasing with
                                         region padding; no gVCF, no normalize_reads, no
                                         methylation-aware phasing)
  RegionProcessor.process                make_examples_core.py:2215-2380
  partition                              ranges.RangeSet.partition (1000-base calling regions)

Every stage below is this package's own: realigner/ (device allele counts for window
selection, native assembly and alignment), allelecounter.AlleleCounter (device),
variant_calling.VariantCaller (host), direct_phasing.DirectPhasing (native, host),
make_examples_native.ExamplesGenerator (device encoder, optionally fused with the CNN).  File
handling, sharding, labelling, gVCF and multi-sample plumbing of the reference's make_examples are
outside SURVEY.md section 8.
"""
from __future__ import annotations

import dataclasses
import numpy as np
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

from deepvariant_amd import allelecounter
from deepvariant_amd import direct_phasing
from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_native
from deepvariant_amd import packing
from deepvariant_amd import variant_calling
from deepvariant_amd.realigner import realigner as realigner_module
from deepvariant_amd.realigner import utils


@dataclasses.dataclass
class RegionProcessorOptions:
  """The MakeExamplesOptions fields this slice reads (flag defaults of make_examples_options.py)."""
  realigner_enabled: bool = True
  realigner_options: Optional[realigner_module.RealignerOptions] = None    # None = realigner_config()
  max_read_length_to_realign: int = 500
  vsc_min_count_snps: int = 2
  vsc_min_count_indels: int = 2
  vsc_min_fraction_snps: float = 0.12
  vsc_min_fraction_indels: float = 0.06
  keep_legacy_allele_counter_behavior: bool = False
  partition_size: int = 1000
  # long-read models (make_examples_options.py:670-700, 1152-1168): phase reads on the fly and
  # tag them HP for the haplotype channel / sort_by_haplotypes; needs track_ref_reads
  track_ref_reads: bool = False
  phase_reads: bool = False
  phase_reads_region_padding_pct: int = 20        # dv_constants.PHASE_READS_REGION_PADDING_PCT
  phase_max_candidates: int = 5000
  min_alleles_to_phase: int = 1


END_OF_REGION = -1        # make_examples_core.py:125-129: markers in a candidate-positions file
END_OF_PARTITION = -2


def partition(region: T.Range, size: int) -> Iterator[T.Range]:
  """Calling regions of at most `size` bases, in order."""
  if size <= 0:
    raise ValueError('partition size must be positive')
  for start in range(region.start, region.end, size):
    yield T.Range(region.reference_name, start, min(start + size, region.end))


def _contig_tuples(contigs) -> List[Tuple[str, int]]:
  return [(c[0], int(c[1])) if isinstance(c, (tuple, list)) else (c.name, int(c.n_bases)) for c in contigs]


def merge_ranges(ranges: Iterable[T.Range], contig_order: Optional[Sequence[str]] = None,
                 known_contigs_only: bool = False) -> List[T.Range]:
  """What iterating a nucleus `RangeSet` yields (third_party/nucleus/util/ranges.py:77-147):
  overlapping AND adjacent ranges merged (`merge_overlaps(strict=False)`), sorted by contig --
  in `contig_order` (the FASTA's) when given, else by name -- then by start.  With
  `known_contigs_only` a range on a contig outside `contig_order` is an error, as it is for a
  RangeSet built with `contigs`."""
  by_contig: Dict[str, List[Tuple[int, int]]] = {}
  for r in ranges:
    by_contig.setdefault(r.reference_name, []).append((int(r.start), int(r.end)))
  if contig_order is not None:
    pos = {name: i for i, name in enumerate(contig_order)}
    if known_contigs_only:
      for name in by_contig:
        if name not in pos:
          raise ValueError('Range on an unrecognized contig: %s' % name)
    names = sorted(by_contig, key=lambda n: (pos.get(n, len(pos)), n))
  else:
    names = sorted(by_contig)
  out = []
  for name in names:
    merged: List[List[int]] = []
    for start, end in sorted(by_contig[name]):
      if merged and start <= merged[-1][1]:          # overlapping or touching
        merged[-1][1] = max(merged[-1][1], end)
      else:
        merged.append([start, end])
    out.extend(T.Range(name, a, b) for a, b in merged)
  return out


def intersect_ranges(a: Sequence[T.Range], b: Sequence[T.Range],
                     contig_order: Optional[Sequence[str]] = None) -> List[T.Range]:
  """`RangeSet.intersection` (ranges.py:204-279): the bases common to both sets; contigs present
  in only one of them drop out."""
  a, b = merge_ranges(a, contig_order), merge_ranges(b, contig_order)
  b_by: Dict[str, List[T.Range]] = {}
  for r in b:
    b_by.setdefault(r.reference_name, []).append(r)
  out = []
  for r in a:
    for o in b_by.get(r.reference_name, ()):
      lo, hi = max(r.start, o.start), min(r.end, o.end)
      if lo < hi:
        out.append(T.Range(r.reference_name, lo, hi))
  return merge_ranges(out, contig_order)


def exclude_ranges(a: Sequence[T.Range], b: Sequence[T.Range],
                   contig_order: Optional[Sequence[str]] = None) -> List[T.Range]:
  """`RangeSet.exclude_regions` (ranges.py:281-300): `a` with every base of `b` chopped out."""
  cut_by: Dict[str, List[T.Range]] = {}
  for r in merge_ranges(b, contig_order):
    cut_by.setdefault(r.reference_name, []).append(r)
  out = []
  for r in merge_ranges(a, contig_order):
    start = r.start
    for c in cut_by.get(r.reference_name, ()):
      if c.end <= start or c.start >= r.end:
        continue
      if c.start > start:
        out.append(T.Range(r.reference_name, start, c.start))
      start = max(start, c.end)
    if start < r.end:
      out.append(T.Range(r.reference_name, start, r.end))
  return out


def build_calling_regions(contigs, regions_to_include: Sequence[T.Range] = (),
                          regions_to_exclude: Sequence[T.Range] = ()) -> List[T.Range]:
  """calling_regions_utils.build_calling_regions (calling_regions_utils.py:48-98) without the
  reference-N exclusion (`--discard_non_dna_regions`, default off): every base of `contigs`,
  intersected with the regions to include if there are any, minus the regions to exclude."""
  contigs = _contig_tuples(contigs)
  order = [name for name, _ in contigs]
  regions = [T.Range(name, 0, n) for name, n in contigs]
  if regions_to_include:
    regions = intersect_ranges(regions, regions_to_include, order)
  if regions_to_exclude:
    regions = exclude_ranges(regions, regions_to_exclude, order)
  return merge_ranges(regions, order)


def regions_to_process(contigs, partition_size: int, calling_regions: Optional[Sequence[T.Range]] = None,
                       task_id: Optional[int] = None, num_shards: Optional[int] = None,
                       round_robin_sampling: bool = True) -> List[T.Range]:
  """make_examples_core.regions_to_process (make_examples_core.py:800-888): the contigs'
  bases (intersected with `calling_regions`), in FASTA contig order, cut into pieces of at
  most `partition_size`; with shards, task `task_id` takes the pieces i % num_shards ==
  task_id -- the rule the reference applies whenever examples go to TFRecords, and the rule
  the ranks of a multi-GPU run use (deepvariant_amd/dist.py) -- or, without round robin, the
  task_id-th block of ceil(n / num_shards) consecutive pieces."""
  if (task_id is None) != (num_shards is None):
    raise ValueError('Both task_id and num_shards must be present if either is', task_id, num_shards)
  if num_shards:
    if num_shards < 0:
      raise ValueError('num_shards={} must be >= 0'.format(num_shards))
    if task_id < 0 or task_id >= num_shards:
      raise ValueError('task_id={} should be >= 0 and < num_shards={}'.format(task_id, num_shards))
  contigs = _contig_tuples(contigs)
  order = [name for name, _ in contigs]
  regions = [T.Range(name, 0, n) for name, n in contigs]
  if calling_regions:
    regions = intersect_ranges(regions, calling_regions, order)
  else:
    regions = merge_ranges(regions, order)
  pieces = [p for r in regions for p in partition(r, partition_size)]
  if num_shards:
    if round_robin_sampling:
      return [p for i, p in enumerate(pieces) if i % num_shards == task_id]
    per_shard = -(-len(pieces) // num_shards)
    return pieces[task_id * per_shard:(task_id + 1) * per_shard]
  return pieces


class RegionProcessor:
  def __init__(self, options: T.MakeExamplesOptions, ref_reader, processor_options: Optional[RegionProcessorOptions] = None,
               example_filenames: Optional[Dict[str, str]] = None, device: int = 0):
    if len(options.sample_options) != 1:
      raise NotImplementedError('RegionProcessor handles one sample; multi-sample regions go through '
                                'ExamplesGenerator directly')
    self.options = options
    self.ref_reader = ref_reader
    self.processor_options = processor_options or RegionProcessorOptions()
    po = self.processor_options
    if po.phase_reads and not po.track_ref_reads:
      raise ValueError('--track_ref_reads must be set to True when --phase_reads is set.')
    self.direct_phasing = direct_phasing.DirectPhasing(po.min_alleles_to_phase) if po.phase_reads else None
    self.realigner = None
    if po.realigner_enabled:
      self.realigner = realigner_module.Realigner(po.realigner_options or realigner_module.realigner_config(),
                                                  ref_reader)
    sample = options.sample_options[0]
    self.variant_caller = variant_calling.VariantCaller(variant_calling.VariantCallerOptions(
        po.vsc_min_count_snps, po.vsc_min_count_indels, po.vsc_min_fraction_snps, po.vsc_min_fraction_indels,
        sample_name=sample.name, track_ref_reads=po.track_ref_reads))
    self._queue: List = []
    self.n_queued_examples = 0
    self.generator = make_examples_native.ExamplesGenerator(
        options, example_filenames or {}, test_mode=not example_filenames, device=device, ref_reader=ref_reader)

  def realign_reads(self, reads: Sequence, region: T.Range) -> List:
    """Reads longer than max_read_length_to_realign bypass the realigner and come first."""
    if self.realigner is None:
      return list(reads)
    limit = self.processor_options.max_read_length_to_realign
    if limit == 0:
      return self.realigner.realign_reads(reads, region)[1]
    long_reads = [r for r in reads if len(r.aligned_sequence) > limit]
    short_reads = [r for r in reads if len(r.aligned_sequence) <= limit]
    return long_reads + self.realigner.realign_reads(short_reads, region)[1]

  def _allele_counter(self, region: T.Range, table, candidate_positions=()):
    rr = self.options.pic_options.read_requirements
    counter = allelecounter.AlleleCounter(
        self.ref_reader, region.reference_name, region.start, region.end,
        candidate_positions=candidate_positions, min_mapping_quality=rr.min_mapping_quality,
        min_base_quality=rr.min_base_quality,
        keep_legacy_behavior=self.processor_options.keep_legacy_allele_counter_behavior,
        track_ref_reads=self.processor_options.track_ref_reads)
    counter.add_table(table)        # the region's reads are packed once for both passes
    return counter

  def candidates_in_region(self, region: T.Range, reads: Sequence,
                           padded_region: Optional[T.Range] = None) -> List[T.DeepVariantCall]:
    """Allele counts over the (padded) region from the reads that overlap `region` (one kernel
    launch; two with track_ref_reads: the first finds the positions that will be called, the
    second keeps their reference-supporting reads by name), the candidate caller, and -- with
    phase_reads -- the phasing of those reads (their HP tags are REPLACED in place)."""
    po = self.processor_options
    in_region = [r for r in reads if utils.ranges_overlap(utils.read_range(r), region)]
    if not in_region:
      return []
    effective = padded_region or region
    table = packing.ReadTable.from_reads(in_region)
    positions = ()
    if po.track_ref_reads:
      first_pass = self._allele_counter(effective, table)
      positions = self.variant_caller.call_positions_from_allele_counter(first_pass)
    candidates = self.variant_caller.calls_from_allele_counter(self._allele_counter(effective, table, positions))
    if self.direct_phasing is not None:
      to_phase = [r for r in in_region if utils.ranges_overlap(utils.read_range(r), effective)]
      for read in to_phase:
        read.info.pop('HP', None)                 # an existing phasing must not leak into the images
      if not (po.phase_max_candidates and len(candidates) > po.phase_max_candidates):
        for read, phase in zip(to_phase, self.direct_phasing.phase(candidates, to_phase)):
          if self.options.pic_options.reverse_haplotypes and phase in (1, 2):
            phase = 1 + (phase % 2)
          read.info['HP'] = T.ListValue(values=[T.Value(int_value=int(phase))])
    if padded_region is not None:                 # filter_candidates_by_region, :2579-2606
      candidates = [c for c in candidates if region.start <= c.variant.start < region.end]
    return candidates

  def find_candidate_positions(self, region: T.Range, reads: Sequence) -> List[int]:
    """candidate_sweep mode (make_examples_core.py:2117-2189): the positions at which the RAW reads
    of the region (no realigner) would make the caller emit a candidate, then END_OF_PARTITION."""
    in_region = [r for r in reads if utils.ranges_overlap(utils.read_range(r), region)]
    if not in_region:
      return [END_OF_PARTITION]
    counter = self._allele_counter(region, packing.ReadTable.from_reads(in_region))
    return self.variant_caller.call_positions_from_allele_counter(counter) + [END_OF_PARTITION]

  def process(self, region: T.Range, reads: Sequence) -> Tuple[List[T.DeepVariantCall], List]:
    """-> (candidates, the region's reads as the pileup images must see them)."""
    po = self.processor_options
    realigned = self.realign_reads(reads, region)
    padded = None
    if po.phase_reads:
      # HP tags are written on this region's own copies: a long read that overlaps two calling
      # regions is phased in each (the labels 1 / 2 are only meaningful within a region)
      realigned = [dataclasses.replace(r, info=dict(r.info)) for r in realigned]
      if po.phase_reads_region_padding_pct > 0:
        padding = int((region.end - region.start) * po.phase_reads_region_padding_pct / 100)
        padded = utils.expand(region, padding, self.ref_reader.n_bases(region.reference_name))
    return self.candidates_in_region(region, realigned, padded), realigned

  # ---- the table path: the region's reads stay a packed table from the BAM decoder to the
  # encoder (make_examples.RegionReads.table); no Read objects.  It covers the default short-read
  # calling configuration; everything that works on Read objects (read phasing, spliced-read
  # splitting, trimmed / alt-aligned pileups, channels with per-read aux pixels) takes the
  # object path above.
  def table_path_ok(self) -> bool:
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    po, gen = self.processor_options, self.generator
    sample = self.options.sample_options[0]
    return (not po.phase_reads and
            not (self.realigner is not None and self.realigner.config.split_skip_reads) and
            gen._alt_mode == aap.NONE and                                      # pylint: disable=protected-access
            not getattr(self.options, 'trim_reads_for_pileup', False) and
            not sample.keep_only_window_spanning_reads and
            not gen._encoder_api._need_aux and not gen._encoder_api._need_seq_aux)   # pylint: disable=protected-access

  def realign_table(self, table, region: T.Range):
    """realign_reads on a table: reads longer than max_read_length_to_realign bypass the
    realigner and come first."""
    return self.realign_tables([table], [region])[0]

  def realign_tables(self, tables: Sequence, regions: Sequence[T.Range]) -> List:
    """`realign_table` for a batch of calling regions: the realigner's assembly and alignment
    work of all of them goes through one native, threaded call (Realigner.realign_tables)."""
    return self.start_realign_tables(tables, regions)()

  def start_realign_tables(self, tables: Sequence, regions: Sequence[T.Range], executor=None):
    """-> a callable that returns the realigned tables.  The regions' windows are selected now;
    with an `executor` the native call is already running on one of its threads when this
    returns (Realigner.start_realign_tables)."""
    tables = list(tables)
    if self.realigner is None:
      return lambda: tables
    limit = self.processor_options.max_read_length_to_realign
    long_parts = [None] * len(tables)
    short = tables
    if limit:
      short = []
      for k, table in enumerate(tables):
        lengths = np.diff(table.read_seq_off.astype(np.int64))
        long_rows = np.nonzero(lengths > limit)[0]
        if len(long_rows):
          long_parts[k] = table.take(long_rows)
          table = table.take(np.nonzero(lengths <= limit)[0])
        short.append(table)
    job = self.realigner.start_realign_tables(short, regions, want_haplotypes=False, executor=executor)

    def finish():
      return [t if lp is None else packing.concat_tables([lp, t])
              for lp, (_, t) in zip(long_parts, job.result())]
    return finish

  def process_table(self, region: T.Range, table, realigned=None) -> Tuple[List[T.DeepVariantCall], 'packing.ReadTable']:
    """`realigned`: the region's table as `realign_tables` returned it (the runner realigns a
    batch of regions ahead); None = realign here."""
    return self.process_tables([region], [table], None if realigned is None else [realigned])[0]

  def process_tables(self, regions: Sequence[T.Range], tables: Sequence, realigned_tables: Optional[Sequence] = None
                     ) -> List[Tuple[List[T.DeepVariantCall], 'packing.ReadTable']]:
    """Candidates of a batch of calling regions: -> [(candidates, realigned table)] per region.  The
    regions' allele counters are filled in one device call per pass (AlleleCounter.run_batch: one
    upload, kernels back to back -- two passes with track_ref_reads), the caller then reads each
    region's counts on its own."""
    if realigned_tables is None:
      realigned_tables = self.realign_tables(tables, regions)
    out = [([], realigned) for realigned in realigned_tables]
    slots, in_region = [], []
    for k, (region, realigned) in enumerate(zip(regions, realigned_tables)):
      rows = np.nonzero((realigned.read_end > region.start) & (region.end > realigned.read_pos.astype(np.int64)))[0]
      if len(rows):
        slots.append(k)
        in_region.append(realigned.take(rows))
    run_batch = getattr(allelecounter.AlleleCounter, 'run_batch', None) or (lambda counters: None)
    positions = [()] * len(slots)
    if self.processor_options.track_ref_reads:
      first_pass = [self._allele_counter(regions[k], t) for k, t in zip(slots, in_region)]
      run_batch(first_pass)
      positions = [self.variant_caller.call_positions_from_allele_counter(c) for c in first_pass]
    counters = [self._allele_counter(regions[k], t, p) for k, t, p in zip(slots, in_region, positions)]
    run_batch(counters)
    for k, counter in zip(slots, counters):
      out[k] = (self.variant_caller.calls_from_allele_counter(counter), realigned_tables[k])
    return out

  def examples_in_region_table(self, region: T.Range, table, stats: Optional[dict] = None, realigned=None,
                               called=None) -> Tuple[List[T.DeepVariantCall], List[bytes]]:
    """`called`: (candidates, realigned table) of this region out of `process_tables`."""
    candidates, realigned = called if called is not None else self.process_table(region, table, realigned)
    if not candidates:
      return candidates, []
    examples, _ = self.generator.encode_region(candidates, [realigned], [0], [0.0], stats if stats is not None else {})
    return candidates, examples

  def call_variants_in_region_table(self, region: T.Range, table, model) -> Tuple[List[T.DeepVariantCall], List[bytes]]:
    candidates, realigned = self.process_table(region, table)
    if not candidates:
      return candidates, []
    return candidates, self.generator.call_variants_in_region(candidates, [realigned], [0], [0.0], model)

  # ---- deferred classification (table path, fused route): regions are drawn on the device one by
  # one, their tensors wait there, and ONE CNN forward classifies a few hundred examples
  def queue_region_table(self, region: T.Range, table, model, realigned=None) -> List[T.DeepVariantCall]:
    candidates, realigned = self.process_table(region, table, realigned)
    return self.queue_region_candidates(candidates, realigned, model)

  def queue_region_candidates(self, candidates, realigned, model) -> List[T.DeepVariantCall]:
    """The drawing half of queue_region_table, for a driver that called the candidates of a batch
    of regions at once (process_tables)."""
    images, plan = (None, [])
    if candidates:
      images, plan = self.generator.encode_region_on_device(candidates, [realigned], [0], [0.0], model.input_shape)
    self._queue.append((candidates, plan, images))
    self.n_queued_examples += len(plan)
    return candidates

  def flush_queue(self, model) -> List[List[bytes]]:
    """-> the CallVariantsOutput records of every queued region, in queue order.  Examples are
    classified independently of their batch (no kernel reduces across examples), so the records
    are the ones region-by-region classification gives, byte for byte."""
    import torch
    from deepvariant_amd import call_variants as cv
    queue, self._queue, self.n_queued_examples = self._queue, [], 0
    tensors = [img for _, plan, img in queue if plan]
    out: List[List[bytes]] = []
    if not tensors:
      return [[] for _ in queue]
    limit = model.max_batch
    stacked = torch.cat(tensors) if len(tensors) > 1 else tensors[0]
    rows = [cv.round_gls_batch(model(stacked[i:i + limit]).cpu().numpy(), 10) for i in range(0, stacked.shape[0], limit)]
    gls = np.concatenate(rows)
    at = 0
    for candidates, plan, _ in queue:
      out.append(self.generator.call_variants_outputs(candidates, plan, gls[at:at + len(plan)]) if plan else [])
      at += len(plan)
    return out

  def examples_in_region(self, region: T.Range, reads: Sequence, stats: Optional[dict] = None
                         ) -> Tuple[List[T.DeepVariantCall], List[bytes]]:
    """-> (candidates, serialised tf.Examples in candidate / alt-combination order)."""
    candidates, realigned = self.process(region, reads)
    if not candidates:
      return candidates, []
    examples, _ = self.generator.encode_region(candidates, [realigned], [0], [0.0], stats if stats is not None else {})
    return candidates, examples

  def call_variants_in_region(self, region: T.Range, reads: Sequence, model) -> Tuple[List[T.DeepVariantCall], List[bytes]]:
    """Fused: -> (candidates, serialised CallVariantsOutput per example); the images never
    leave the device."""
    candidates, realigned = self.process(region, reads)
    if not candidates:
      return candidates, []
    return candidates, self.generator.call_variants_in_region(candidates, [realigned], [0], [0.0], model)
