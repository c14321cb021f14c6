"""The window realigner: select windows with read evidence of variation, assemble candidate
haplotypes per window, re-align each read to its best haplotype and through it to the
reference.  Same names and behaviour as deepvariant/realigner/realigner.py:

  realigner_config / window_selector_config      :276-414   (flag defaults :60-274)
  AssemblyRegion, assign_reads_to_assembled_regions, split_reads   :538-672
  Realigner.call_debruijn_graph / call_fast_pass_aligner / realign_reads / align_to_haplotype
                                                  :675-893
  trim_cigar / trim_read                          :896-1010 (shared with alt_aligned_pileup_lib)

The three compute stages are native: per-position allele counts on the device
(allele_counter.hip, one launch per region), graphs in csrc/debruijn_graph.cpp, alignment in
csrc/fast_pass_aligner.cpp + local_align.cpp.  A region's reads are packed ONCE
(packing.ReadTable) and that table feeds the counter and every window's graph.
"""
from __future__ import annotations

import concurrent.futures
import ctypes as C
import dataclasses
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import alt_aligned_pileup_lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import fast_pass_aligner
from deepvariant_amd import packing
from deepvariant_amd.realigner import debruijn_graph
from deepvariant_amd.realigner import utils
from deepvariant_amd.realigner import window_selector
from deepvariant_amd.realigner.debruijn_graph import DeBruijnGraphOptions
from deepvariant_amd.realigner.window_selector import (
    ALLELE_COUNT_LINEAR, VARIANT_READS, AlleleCountLinearModel, VariantReadsThresholdModel,
    WindowSelectorModel, WindowSelectorOptions)

# Windows of a region are independent: their assembly and alignment calls (native, the GIL is
# released inside ctypes) run on a small thread pool; results are collected in window order.
_THREADS = max(1, int(os.environ.get('DV_REALIGN_THREADS', '4')))
# The table path hands all windows of a batch of regions to one native call (dv_realign_regions),
# which runs them on its own host threads: DV_REALIGN_THREADS, or one per hardware thread up to 16.
_NATIVE_THREADS = max(0, int(os.environ.get('DV_REALIGN_THREADS', '0')))
_pool: Optional[concurrent.futures.ThreadPoolExecutor] = None


def _map_in_order(fn, items):
  global _pool
  items = list(items)
  if _THREADS == 1 or len(items) < 2:
    return [fn(x) for x in items]
  if _pool is None:
    _pool = concurrent.futures.ThreadPoolExecutor(max_workers=_THREADS, thread_name_prefix='dv-realign')
  return list(_pool.map(fn, items))


_UNSET_WS_INT_FLAG = -1
_REF_ALIGN_MARGIN = 20                  # realigner.py:263
_DEFAULT_MIN_SUPPORTING_READS = 2
_DEFAULT_MAX_SUPPORTING_READS = 300
_MIN_SPLIT_LEN = 15
_MIN_ALLELE_SUPPORT = 2

_ALLELE_COUNT_LINEAR_MODEL_DEFAULT = WindowSelectorModel(     # realigner.py:266-278
    model_type=ALLELE_COUNT_LINEAR,
    allele_count_linear_model=AlleleCountLinearModel(
        bias=-0.683379, coeff_soft_clip=2.997000, coeff_substitution=-0.086644, coeff_insertion=2.493585,
        coeff_deletion=1.795914, coeff_reference=-0.059787, decision_boundary=3))


@dataclasses.dataclass
class AlignerOptions:
  """deepvariant/protos/realigner.proto AlignerOptions (the fields FastPassAligner reads)."""
  match: int = 4
  mismatch: int = 6
  gap_open: int = 8
  gap_extend: int = 2
  k: int = 23
  error_rate: float = 0.01
  max_num_of_mismatches: int = 2
  realignment_similarity_threshold: float = 0.16934
  kmer_size: int = 32
  force_alignment: bool = False
  realign_all: bool = False
  read_size: int = 0


@dataclasses.dataclass
class RealignerOptions:
  ws_config: WindowSelectorOptions
  dbg_config: DeBruijnGraphOptions
  aln_config: AlignerOptions
  split_skip_reads: bool = False
  normalize_reads: bool = False


@dataclasses.dataclass
class CandidateHaplotypes:
  """realigner.proto CandidateHaplotypes: a window and the haplotypes assembled for it."""
  span: T.Range
  haplotypes: List[str]


_FLAG_DEFAULTS = dict(                   # realigner.py:60-262
    ws_use_window_selector_model=False, ws_window_selector_model=None,
    ws_min_num_supporting_reads=_UNSET_WS_INT_FLAG, ws_max_num_supporting_reads=_UNSET_WS_INT_FLAG,
    ws_min_mapq=20, ws_min_base_quality=20, ws_min_windows_distance=80, ws_max_window_size=1000,
    realign_all=False, ws_region_expansion_in_bp=20,
    dbg_min_k=10, dbg_max_k=101, dbg_step_k=1, dbg_min_mapq=14, dbg_min_base_quality=15,
    dbg_min_edge_weight=2, dbg_max_num_paths=256, dbg_disable_graph_pruning=False,
    aln_match=4, aln_mismatch=6, aln_gap_open=8, aln_gap_extend=2, aln_k=23, aln_error_rate=0.01,
    max_num_mismatches=2, realignment_similarity_threshold=0.16934, split_skip_reads=False, kmer_size=32,
    keep_legacy_allele_counter_behavior=False, enable_strict_insertion_filter=False, normalize_reads=False)


def _flags(overrides: dict) -> dict:
  unknown = set(overrides) - set(_FLAG_DEFAULTS)
  if unknown:
    raise ValueError('unknown realigner flag(s): %s' % ', '.join(sorted(unknown)))
  flags = dict(_FLAG_DEFAULTS)
  flags.update(overrides)
  return flags


def window_selector_config(**overrides) -> WindowSelectorOptions:
  """window_selector_config(flags) (:276-366); flags by keyword, the reference's defaults.
  `ws_window_selector_model` is a WindowSelectorModel (the reference reads a text proto)."""
  f = _flags(overrides)
  if not f['ws_use_window_selector_model']:
    if f['ws_window_selector_model'] is not None:
      raise ValueError('Cannot specify a ws_window_selector_model if ws_use_window_selector_model is False.')
    lo, hi = f['ws_min_num_supporting_reads'], f['ws_max_num_supporting_reads']
    model = WindowSelectorModel(
        model_type=VARIANT_READS,
        variant_reads_model=VariantReadsThresholdModel(
            min_num_supporting_reads=_DEFAULT_MIN_SUPPORTING_READS if lo == _UNSET_WS_INT_FLAG else lo,
            max_num_supporting_reads=_DEFAULT_MAX_SUPPORTING_READS if hi == _UNSET_WS_INT_FLAG else hi))
  else:
    if f['ws_min_num_supporting_reads'] != _UNSET_WS_INT_FLAG:
      raise ValueError('Cannot use both ws_min_num_supporting_reads and ws_use_window_selector_model flags.')
    if f['ws_max_num_supporting_reads'] != _UNSET_WS_INT_FLAG:
      raise ValueError('Cannot use both ws_max_num_supporting_reads and ws_use_window_selector_model flags.')
    model = f['ws_window_selector_model'] or _ALLELE_COUNT_LINEAR_MODEL_DEFAULT
  if model.model_type == VARIANT_READS:
    m = model.variant_reads_model
    if m.max_num_supporting_reads < m.min_num_supporting_reads:
      raise ValueError('ws_min_supporting_reads should be smaller than ws_max_supporting_reads.')
  return WindowSelectorOptions(
      min_mapq=f['ws_min_mapq'], min_base_quality=f['ws_min_base_quality'],
      min_windows_distance=f['ws_min_windows_distance'], max_window_size=f['ws_max_window_size'],
      region_expansion_in_bp=f['ws_region_expansion_in_bp'], window_selector_model=model,
      keep_legacy_behavior=f['keep_legacy_allele_counter_behavior'], realign_all=f['realign_all'],
      min_allele_support=_MIN_ALLELE_SUPPORT,
      enable_strict_insertion_filter=f['enable_strict_insertion_filter'])


def realigner_config(**overrides) -> RealignerOptions:
  """realigner_config(flags) (:368-414)."""
  f = _flags(overrides)
  return RealignerOptions(
      ws_config=window_selector_config(**overrides),
      dbg_config=DeBruijnGraphOptions(
          min_k=f['dbg_min_k'], max_k=f['dbg_max_k'], step_k=f['dbg_step_k'], min_mapq=f['dbg_min_mapq'],
          min_base_quality=f['dbg_min_base_quality'], min_edge_weight=f['dbg_min_edge_weight'],
          max_num_paths=f['dbg_max_num_paths'], disable_graph_pruning=f['dbg_disable_graph_pruning']),
      aln_config=AlignerOptions(
          match=f['aln_match'], mismatch=f['aln_mismatch'], gap_open=f['aln_gap_open'],
          gap_extend=f['aln_gap_extend'], k=f['aln_k'], error_rate=f['aln_error_rate'],
          max_num_of_mismatches=f['max_num_mismatches'],
          realignment_similarity_threshold=f['realignment_similarity_threshold'], kmer_size=f['kmer_size'],
          force_alignment=False, realign_all=f['realign_all']),
      split_skip_reads=f['split_skip_reads'], normalize_reads=f['normalize_reads'])


class AssemblyRegion:
  """One assembled window: its candidate haplotypes and the reads assigned to it.  Same public
  surface as the reference's class (deepvariant/realigner/realigner.py:538-593: `region`,
  `haplotypes`, `reads`, `read_span`, `add_read`); the span of the assigned reads is kept as a
  running [lowest start, highest end) instead of being recomputed from the read list."""

  __slots__ = ('candidate_haplotypes', 'reads', '_lo', '_hi')

  def __init__(self, candidate_haplotypes: CandidateHaplotypes):
    self.candidate_haplotypes = candidate_haplotypes
    self.reads: List = []
    self._lo: Optional[int] = None
    self._hi: Optional[int] = None

  def __str__(self):
    span = self.read_span
    return 'AssemblyRegion(region={}, span={}) with {} haplotypes and {} reads'.format(
        self.region, span, len(self.haplotypes), len(self.reads))

  @property
  def haplotypes(self) -> List[str]:
    return self.candidate_haplotypes.haplotypes

  @property
  def region(self) -> T.Range:
    return self.candidate_haplotypes.span

  @property
  def read_span(self) -> Optional[T.Range]:
    if self._lo is None:
      return None
    return utils.make_range(self.region.reference_name, self._lo, self._hi)

  def _cover(self, start: int, end: int) -> None:
    self._lo = start if self._lo is None else min(self._lo, start)
    self._hi = end if self._hi is None else max(self._hi, end)

  def add_read(self, read) -> None:
    span = utils.read_range(read)
    self.reads.append(read)
    self._cover(span.start, span.end)

  def add_reads(self, reads: Sequence, starts, ends) -> None:
    """Bulk form for reads whose alignment spans are already known (parallel arrays)."""
    if len(reads):
      self.reads.extend(reads)
      self._cover(int(min(starts)), int(max(ends)))


def _read_spans(reads: Sequence):
  """(contig per read, starts int64, ends int64): ReadRange of every read, as arrays."""
  spans = [utils.read_range(r) for r in reads]
  return ([sp.reference_name for sp in spans], np.array([sp.start for sp in spans], np.int64),
          np.array([sp.end for sp in spans], np.int64))


def assign_reads_to_assembled_regions(assembled_regions: Sequence[AssemblyRegion], reads: Sequence) -> List:
  """Every read joins the window it shares most bases with -- the first such window on ties --
  and the reads that touch no window are returned (the reference's rule,
  deepvariant/realigner/realigner.py:596-619, evaluated for all reads x windows at once)."""
  reads = list(reads)
  if not reads:
    return []
  if not assembled_regions:
    return reads
  contigs, starts, ends = _read_spans(reads)
  w_lo = np.array([ar.region.start for ar in assembled_regions], np.int64)
  w_hi = np.array([ar.region.end for ar in assembled_regions], np.int64)
  shared = np.minimum(ends[:, None], w_hi[None, :]) - np.maximum(starts[:, None], w_lo[None, :])
  same_contig = np.array([[c == ar.region.reference_name for ar in assembled_regions] for c in contigs])
  shared = np.where(same_contig, np.maximum(shared, 0), 0)
  best = shared.argmax(axis=1)                    # argmax returns the FIRST maximum
  claimed = shared[np.arange(len(reads)), best] > 0
  for w, ar in enumerate(assembled_regions):
    mine = np.nonzero(claimed & (best == w))[0]
    ar.add_reads([reads[i] for i in mine], starts[mine], ends[mine])
  return [reads[i] for i in np.nonzero(~claimed)[0]]


def split_reads(reads: Sequence) -> List:
  """Spliced reads (CIGAR `N`) become one read per exon: the operations between two skips, the
  bases and qualities they consume, the position where the part's first reference-consuming
  operation starts, the name `<name>_p<k>` (k counts the parts, dropped ones included).  Parts
  of fewer than 15 bases are dropped; reads without a skip pass through untouched.  Behaviour
  of deepvariant/realigner/realigner.py:622-672, computed from the read's operation arrays."""
  out = []
  for read in reads:
    ops = np.array([c.operation for c in read.alignment.cigar], np.int64)
    if not (ops == utils.SKIP).any():
      out.append(read)
      continue
    lens = np.array([c.operation_length for c in read.alignment.cigar], np.int64)
    on_read = np.isin(ops, list(utils.READ_ADVANCING_OPS))
    on_ref = np.isin(ops, list(utils.REF_ADVANCING_OPS))
    read_after = np.cumsum(np.where(on_read, lens, 0))          # bases consumed up to and including op i
    ref_before = np.cumsum(np.where(on_ref, lens, 0)) - np.where(on_ref, lens, 0)
    cuts = np.nonzero(ops == utils.SKIP)[0].tolist()
    if cuts[-1] != len(ops) - 1:
      cuts.append(len(ops))                      # the stretch after the last skip
    first = 0                                    # first operation of the current part
    for part, cut in enumerate(cuts):
      body = np.arange(first, cut)               # operations of this part (the skip itself excluded)
      seq_lo = int(read_after[first - 1]) if first > 0 else 0
      seq_hi = int(read_after[cut - 1]) if cut > 0 else 0
      if seq_hi - seq_lo >= _MIN_SPLIT_LEN:
        # the reference assigns the position at the first reference-consuming operation, and again
        # at the next one while the value is still 0 ("unset"): the first non-zero start, else 0
        anchors = [int(read.alignment.position.position + ref_before[i]) for i in body if on_ref[i]]
        position = next((a for a in anchors if a), 0)
        src = read.alignment.position
        out.append(T.Read(
            fragment_name='%s_p%d' % (read.fragment_name, part), read_number=read.read_number,
            number_reads=read.number_reads, fragment_length=read.fragment_length,
            proper_placement=read.proper_placement, duplicate_fragment=read.duplicate_fragment,
            failed_vendor_quality_checks=read.failed_vendor_quality_checks,
            secondary_alignment=read.secondary_alignment, supplementary_alignment=read.supplementary_alignment,
            aligned_sequence=read.aligned_sequence[seq_lo:seq_hi], aligned_quality=read.aligned_quality[seq_lo:seq_hi],
            alignment=T.LinearAlignment(
                position=T.Position(src.reference_name, position, src.reverse_strand),
                mapping_quality=read.alignment.mapping_quality,
                cigar=[T.CigarUnit(int(ops[i]), int(lens[i])) for i in body]),
            info=dict(read.info), base_modifications=dict(read.base_modifications)))
      first = cut + 1
  return out


class RealignJob:
  """One dv_realign_regions call over a batch of regions: arguments marshalled by `add`, the native
  call in `start` (optionally on an executor thread), the write-back in `result`."""

  def __init__(self, n_slots: int, want_haplotypes: bool, options: '_lib.DvRealignOptions'):
    self.results: List = [None] * n_slots
    self._want_haplotypes = want_haplotypes
    self._options = options
    self._jobs: List = []          # (slot, table, usable windows)
    self._keep: List = []          # the arrays the region descriptors point into
    self._future = None
    self._output = None            # (handle, DvRealignOutput) once the native call has run

  def add(self, slot: int, table: 'packing.ReadTable', usable: Sequence[T.Range], ref_reader, contig: str,
          n_contig: int) -> None:
    starts = np.ascontiguousarray(table.read_pos, np.int64)
    ends = np.ascontiguousarray(table.read_end, np.int64)
    w_lo = np.array([w.start for w in usable], np.int64)
    w_hi = np.array([w.end for w in usable], np.int64)
    ref_lo = max(0, min(int(starts.min()), int(w_lo.min())) - _REF_ALIGN_MARGIN)
    ref_hi = min(n_contig, max(int(ends.max()), int(w_hi.max())) + _REF_ALIGN_MARGIN)
    ref = ref_reader.get_bases(contig, ref_lo, ref_hi).encode()
    bases = np.ascontiguousarray(table.bases, np.uint8)
    quals = np.ascontiguousarray(table.quals, np.uint8)
    seq_off = np.ascontiguousarray(table.read_seq_off, np.uint32)
    mapq = np.ascontiguousarray(table.read_mapq, np.uint8)
    self._jobs.append((slot, table, list(usable)))
    self._keep.append((bases, quals, seq_off, mapq, starts, ends, w_lo, w_hi, ref, ref_lo, n_contig))

  def _call(self):
    descs = (_lib.DvRealignRegion * len(self._jobs))()
    for d, (slot, table, usable), k in zip(descs, self._jobs, self._keep):
      bases, quals, seq_off, mapq, starts, ends, w_lo, w_hi, ref, ref_lo, n_contig = k
      d.bases, d.quals, d.n_bases = bases.ctypes.data, quals.ctypes.data, len(bases)
      d.read_seq_off, d.read_mapq = seq_off.ctypes.data, mapq.ctypes.data
      d.read_start, d.read_end = starts.ctypes.data, ends.ctypes.data
      d.n_reads, d.n_windows = table.n_reads, len(usable)
      d.window_start, d.window_end = w_lo.ctypes.data, w_hi.ctypes.data
      d.ref, d.ref_start, d.ref_len, d.contig_len = ref, ref_lo, len(ref), n_contig
    handle = C.c_void_p()
    out = _lib.DvRealignOutput()
    _lib.check(_lib.lib().dv_realign_regions(descs, len(self._jobs), C.byref(self._options), C.byref(handle),
                                             C.byref(out)))
    return handle, out

  def start(self, executor=None) -> None:
    if self._jobs and executor is not None:
      self._future = executor.submit(self._call)

  def result(self) -> List:
    """-> [(candidate haplotypes per assembled window, realigned table)] per region of the batch."""
    if not self._jobs:
      return self.results
    if self._output is None:
      self._output = self._future.result() if self._future is not None else self._call()
      self._future = None
      self._write_back()
    return self.results

  def _write_back(self) -> None:
    handle, out = self._output
    jobs, results, want_haplotypes = self._jobs, self.results, self._want_haplotypes
    try:
      n_jobs = len(jobs)
      row_off = np.ctypeslib.as_array(out.region_row_off, shape=(n_jobs + 1,))
      total = int(row_off[-1])
      view = lambda ptr, n, dtype: np.ctypeslib.as_array(ptr, shape=(n,)) if n else np.zeros(0, dtype)   # noqa: E731
      order = view(out.order, total, np.int32)
      status = view(out.status, total, np.int32)
      position = view(out.position, total, np.int64)
      cigar_off = np.ctypeslib.as_array(out.cigar_off, shape=(total + 1,))
      words = view(out.cigar, int(cigar_off[-1]), np.uint32)
      asm_off = np.ctypeslib.as_array(out.region_assembled_off, shape=(n_jobs + 1,))
      n_asm = int(asm_off[-1])
      asm_window = view(out.assembled_window, n_asm, np.int32)
      hap_off = np.ctypeslib.as_array(out.assembled_hap_off, shape=(n_asm + 1,))
      n_haps = int(hap_off[-1])
      text_off = np.ctypeslib.as_array(out.hap_text_off, shape=(n_haps + 1,))
      text = C.string_at(out.hap_text, int(text_off[-1])) if want_haplotypes and n_haps else b''
      for g, (slot, table, usable) in enumerate(jobs):
        a0, a1 = int(asm_off[g]), int(asm_off[g + 1])
        if a0 == a1:                       # no window assembled: the table passes through as it is
          results[slot] = ([], table)
          continue
        haplotypes = []
        if want_haplotypes:
          for a in range(a0, a1):
            haps = [text[int(text_off[h]):int(text_off[h + 1])].decode() for h in range(int(hap_off[a]), int(hap_off[a + 1]))]
            haplotypes.append(CandidateHaplotypes(span=usable[int(asm_window[a])], haplotypes=haps))
        r0, r1 = int(row_off[g]), int(row_off[g + 1])
        changed = np.nonzero(status[r0:r1] == 1)[0]
        realigned = table
        if len(changed):
          c_off = cigar_off[r0:r1 + 1]
          # the changed rows' words are consecutive runs of `words`; rows in between are empty
          realigned = table.with_alignments_csr(
              changed, position[r0:r1][changed],
              np.concatenate([c_off[changed], c_off[changed[-1] + 1:changed[-1] + 2]]), words)
        results[slot] = (haplotypes, realigned.take(order[r0:r1].astype(np.int64)))
    finally:
      _lib.lib().dv_realign_result_free(handle)
      self._keep = []


class Realigner:
  """Realigner(config, ref_reader) (:675-893).  `ref_reader`: n_bases(contig) /
  get_bases(contig, start, end), as everywhere in this package."""

  def __init__(self, config: RealignerOptions, ref_reader, shared_header=None):
    self.config = config
    self.ref_reader = ref_reader
    self.shared_header = shared_header

  # ---- reference access in the reference's terms
  def _is_valid(self, r: T.Range) -> bool:          # GenomeReference::IsValidInterval, reference.cc:95-102
    try:
      n = self.ref_reader.n_bases(r.reference_name)
    except KeyError:
      return False
    return 0 <= r.start <= r.end and r.start < n and r.end <= n

  def _query(self, r: T.Range) -> str:
    if not self._is_valid(r):
      raise ValueError('Invalid interval: %s:%d-%d' % (r.reference_name, r.start, r.end))
    return self.ref_reader.get_bases(r.reference_name, r.start, r.end)

  def call_debruijn_graph(self, windows: Sequence[T.Range], reads: Sequence, table=None) -> List[CandidateHaplotypes]:
    """One graph per window over the reads that overlap it (:703-738)."""
    if table is None:
      table = packing.ReadTable.from_reads(list(reads))
    spans = [utils.read_range(r) for r in reads]
    usable = [w for w in windows
              if w.end - w.start <= self.config.ws_config.max_window_size and self._is_valid(w)]

    def assemble(window):
      ref = self._query(window)
      window_reads = [i for i, s in enumerate(spans) if utils.ranges_overlap(s, window)]
      graph = debruijn_graph.build_from_table(ref, table, window_reads, self.config.dbg_config)
      haplotypes = [ref] if graph is None else graph.candidate_haplotypes()
      if haplotypes and haplotypes != [ref]:
        return CandidateHaplotypes(span=window, haplotypes=haplotypes)
      return None

    return [ch for ch in _map_in_order(assemble, usable) if ch is not None]

  def _aligner(self, read_size: int, force_alignment: bool, prefix_len: int, suffix_len: int):
    a = self.config.aln_config
    return fast_pass_aligner.FastPassAligner(
        match=a.match, mismatch=a.mismatch, gap_open=a.gap_open, gap_extend=a.gap_extend,
        kmer_size=a.kmer_size, read_size=read_size, max_num_of_mismatches=a.max_num_of_mismatches,
        realignment_similarity_threshold=a.realignment_similarity_threshold, force_alignment=force_alignment,
        normalize_reads=self.config.normalize_reads, ref_prefix_len=prefix_len, ref_suffix_len=suffix_len)

  def call_fast_pass_aligner(self, assembled_region: AssemblyRegion) -> List:
    """Realign the region's reads against its haplotypes, each padded with the reference out
    to the reads' span + 20 (:740-793)."""
    if not assembled_region.reads:
      return []
    region = assembled_region.region
    contig = region.reference_name
    span = assembled_region.read_span
    ref_start = max(0, min(span.start, region.start) - _REF_ALIGN_MARGIN)
    ref_end = min(self.ref_reader.n_bases(contig), max(span.end, region.end) + _REF_ALIGN_MARGIN)
    ref_prefix = self._query(utils.make_range(contig, ref_start, region.start))
    ref = self._query(region)
    if ref_end <= region.end:      # no room for a suffix: keep the original alignments
      return assembled_region.reads
    ref_suffix = self._query(utils.make_range(contig, region.end, ref_end))
    aligner = self._aligner(len(assembled_region.reads[0].aligned_sequence), False, len(ref_prefix),
                            len(ref_suffix))
    aligner.set_reference(ref_prefix + ref + ref_suffix, ref_start)
    aligner.set_haplotypes([ref_prefix + target + ref_suffix for target in assembled_region.haplotypes])
    return aligner.realign_reads(assembled_region.reads)

  def realign_reads(self, reads: Sequence, region: T.Range) -> Tuple[List[CandidateHaplotypes], List]:
    """-> (candidate haplotypes per assembled window, all input reads: first the ones no window
    claimed, then window by window; order differs from the input) (:795-855)."""
    if not reads:
      return [], []
    if self.config.split_skip_reads:
      reads = split_reads(reads)
    reads = list(reads)
    table = packing.ReadTable.from_reads(reads)
    windows = window_selector.select_windows(self.config.ws_config, self.ref_reader, reads, region, table=table)
    candidate_haplotypes = self.call_debruijn_graph(windows, reads, table=table)
    assembled_regions = [AssemblyRegion(ch) for ch in candidate_haplotypes]
    realigned = assign_reads_to_assembled_regions(assembled_regions, reads)
    for aligned in _map_in_order(self.call_fast_pass_aligner, assembled_regions):
      realigned.extend(aligned)
    return candidate_haplotypes, realigned

  # ---- the same procedure on packed tables (make_examples' table path): no Read objects, and the
  # assembly + alignment of ALL windows of ALL regions handed over in one native, threaded call
  def realign_table(self, table: 'packing.ReadTable', region: T.Range):
    """`realign_reads` for the reads of a packed table: -> (candidate haplotypes per assembled window,
    the table of ALL input reads -- first the ones no window claimed, then window by window, in
    the order realign_reads returns them -- with the new alignment starts and CIGARs in place)."""
    return self.realign_tables([table], [region])[0]

  def _native_options(self) -> '_lib.DvRealignOptions':
    d, a = self.config.dbg_config, self.config.aln_config
    return _lib.DvRealignOptions(
        _lib.DvDebruijnOptions(d.min_k, d.max_k, d.step_k, d.min_mapq, d.min_base_quality, d.min_edge_weight,
                               d.max_num_paths, int(bool(d.disable_graph_pruning))),
        _lib.DvAlignerOptions(a.match, a.mismatch, a.gap_open, a.gap_extend, a.kmer_size, 0, a.max_num_of_mismatches,
                              a.realignment_similarity_threshold, 0, int(bool(self.config.normalize_reads)), 0, 0),
        _REF_ALIGN_MARGIN, _NATIVE_THREADS)

  def realign_tables(self, tables: Sequence['packing.ReadTable'], regions: Sequence[T.Range],
                     want_haplotypes: bool = True) -> List[Tuple[List[CandidateHaplotypes], 'packing.ReadTable']]:
    """`realign_table` for several calling regions at once: window selection per region (device
    counts from the table, window_selector.py), then ONE dv_realign_regions call in which every
    (region, window) assembly and alignment task of the batch runs on a pool of host threads
    (csrc/region_realigner.cpp), then the new starts and CIGARs written back as array operations.
    Results per region are those of the region-by-region procedure (tests/test_table_path_cpu.py).
    Without `want_haplotypes` the CandidateHaplotypes lists come back empty (make_examples only
    uses the reads)."""
    return self.start_realign_tables(tables, regions, want_haplotypes).result()

  def start_realign_tables(self, tables: Sequence['packing.ReadTable'], regions: Sequence[T.Range],
                           want_haplotypes: bool = True, executor=None) -> 'RealignJob':
    """The same in two steps: the windows of every region are selected now (on the calling thread:
    it launches on the device), the native call runs on `executor` (a concurrent.futures executor;
    None = inside .result()), and `.result()` writes the alignments back.  A region driver starts
    the next batch before it calls candidates and draws pileups for the current one, so the
    assembly and alignment threads work while the main thread is busy elsewhere (the native call
    holds no Python lock)."""
    if self.config.split_skip_reads:
      raise NotImplementedError('split_skip_reads works on Read objects (realign_reads)')
    job = RealignJob(len(tables), want_haplotypes, self._native_options())
    tables, regions = list(tables), list(regions)
    all_windows = window_selector.select_windows_of_tables(self.config.ws_config, self.ref_reader, tables, regions)
    for slot, (table, region, windows) in enumerate(zip(tables, regions, all_windows)):
      if table.n_reads == 0:
        job.results[slot] = ([], table)
        continue
      usable = [w for w in windows
                if w.end - w.start <= self.config.ws_config.max_window_size and self._is_valid(w)]
      if not usable:
        job.results[slot] = ([], table)
        continue
      contig = region.reference_name
      job.add(slot, table, usable, self.ref_reader, contig, self.ref_reader.n_bases(contig))
    job.start(executor)
    return job

  def align_to_haplotype(self, this_haplotype: str, haplotypes: Sequence[str], prefix: str, suffix: str,
                         reads: Sequence, contig: str, ref_start: int) -> List:
    """Reads aligned to a graph of haplotypes, reported in `this_haplotype`'s coordinates
    (:857-893); an entry is None where the reference returns an empty Read."""
    if not reads:
      return []
    margin = min(len(prefix), len(suffix), 100)
    aligner = self._aligner(len(reads[0].aligned_sequence), True, len(prefix) - margin, len(suffix) - margin)
    aligner.set_reference(prefix + this_haplotype + suffix, ref_start)
    aligner.set_haplotypes([prefix + target + suffix for target in haplotypes])
    return aligner.realign_reads(reads)


def trim_cigar(cigar, ref_trim: int, ref_length: int):
  """realigner.trim_cigar (:896-968); the same routine as TrimCigar in alt_aligned_pileup_lib."""
  return alt_aligned_pileup_lib.trim_cigar(cigar, ref_trim, ref_length)


def trim_read(read, region: T.Range):
  """realigner.trim_read (:971-1010)."""
  return alt_aligned_pileup_lib.trim_read(read, region.start, region.end)
