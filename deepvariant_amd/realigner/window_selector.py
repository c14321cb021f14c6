"""Window selection for the realigner: positions whose read evidence (mismatches, indels,
soft clips) asks for local assembly, merged into windows.

  select_windows / _candidates_from_reads / _candidates_to_windows
      deepvariant/realigner/window_selector.py:40-238
  variant_reads_candidates / allele_count_linear_candidates
      deepvariant/realigner/window_selector.cc:62-207

The per-position allele counts come from the device AlleleCounter (one kernel launch per
region, deepvariant_amd/allelecounter.py); turning them into per-position scores is host
numpy.  There is no CPU counting path."""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

from deepvariant_amd import allelecounter
from deepvariant_amd import dv_types as T
from deepvariant_amd.realigner import utils

VARIANT_READS, ALLELE_COUNT_LINEAR = 1, 2     # WindowSelectorModel.ModelType


@dataclasses.dataclass
class VariantReadsThresholdModel:
  min_num_supporting_reads: int = 0
  max_num_supporting_reads: int = 0


@dataclasses.dataclass
class AlleleCountLinearModel:
  bias: float = 0.0
  coeff_soft_clip: float = 0.0
  coeff_substitution: float = 0.0
  coeff_insertion: float = 0.0
  coeff_deletion: float = 0.0
  coeff_reference: float = 0.0
  decision_boundary: float = 0.0


@dataclasses.dataclass
class WindowSelectorModel:
  model_type: int = 0
  variant_reads_model: VariantReadsThresholdModel = dataclasses.field(default_factory=VariantReadsThresholdModel)
  allele_count_linear_model: AlleleCountLinearModel = dataclasses.field(default_factory=AlleleCountLinearModel)


@dataclasses.dataclass
class WindowSelectorOptions:
  """deepvariant/protos/realigner.proto WindowSelectorOptions."""
  min_mapq: int = 0
  min_base_quality: int = 0
  min_windows_distance: int = 0
  max_window_size: int = 0
  region_expansion_in_bp: int = 0
  window_selector_model: WindowSelectorModel = dataclasses.field(default_factory=WindowSelectorModel)
  keep_legacy_behavior: bool = False
  realign_all: bool = False
  min_allele_support: int = 0
  enable_strict_insertion_filter: bool = False


def _update_counts(by, start: int, end: int, counts: np.ndarray):
  """UpdateCounts (window_selector.cc:53-62): [start, end) clipped to the vector."""
  if start > end:
    raise ValueError('Start should be <= end')
  counts[max(start, 0):min(end, len(counts))] += by


def _allele_filter(allele, total_count: int, config: WindowSelectorOptions) -> bool:
  """AlleleFilter (:64-80)."""
  if allele.type == allelecounter.REFERENCE:
    return False
  if allele.count < config.min_allele_support:
    return False
  if config.enable_strict_insertion_filter:
    if allele.type == allelecounter.INSERTION and len(allele.bases) <= 2:
      return float(np.float32(allele.count) / np.float32(total_count)) >= 0.08
  return True


def _positions_with_read_alleles(allele_counter):
  """(offset in the interval, AlleleCount) for the positions that carry read alleles; the others
  add nothing to either model beyond their reference count."""
  sparse = getattr(allele_counter, 'counts_with_read_alleles', None)
  if sparse is None:
    return [(i, ac) for i, ac in enumerate(allele_counter.counts()) if ac.read_alleles]
  counts = sparse()
  if not counts:
    return []
  start = allele_counter.interval_start()
  return [(ac.position.position - start, ac) for ac in counts]


def _ref_supporting_read_counts(allele_counter) -> np.ndarray:
  dense = getattr(allele_counter, 'ref_supporting_read_counts', None)
  if dense is not None:
    return np.asarray(dense())
  return np.array([ac.ref_supporting_read_count for ac in allele_counter.counts()], np.int64)


def variant_reads_candidates_from_allele_counter(allele_counter, config: WindowSelectorOptions) -> List[int]:
  """VariantReadsWindowSelectorCandidates (:101-141)."""
  fast = getattr(allele_counter, 'variant_read_window_counts', None)
  if fast is not None:      # the device counter: footprints straight from its event arrays
    counts = fast(config.min_allele_support, config.enable_strict_insertion_filter)
    if counts is not None:
      return [int(x) for x in counts]
  window_counts = np.zeros(allele_counter.interval_length(), np.int64)
  for i, ac in _positions_with_read_alleles(allele_counter):
    total = allelecounter.total_allele_counts(ac)
    for allele in allelecounter.sum_allele_counts(ac):
      if not _allele_filter(allele, total, config):
        continue
      n = len(allele.bases)
      if allele.type == allelecounter.SUBSTITUTION:
        _update_counts(allele.count, i, i + 1, window_counts)
      elif allele.type in (allelecounter.SOFT_CLIP, allelecounter.INSERTION):
        _update_counts(allele.count, i + 1 - (n - 1), i + n, window_counts)
      elif allele.type == allelecounter.DELETION:
        _update_counts(allele.count, i + 1, i + n, window_counts)
      else:
        raise ValueError('Saw an Allele with an unexpected type %r' % (allele.type,))
  return [int(x) for x in window_counts]


def allele_count_linear_candidates_from_allele_counter(allele_counter, model: AlleleCountLinearModel):
  """AlleleCountLinearWindowSelectorCandidates (:143-207); float32 like the reference."""
  f32 = np.float32
  scores = np.full(allele_counter.interval_length(), f32(model.bias), np.float32)
  # float32 sums depend on their order; the reference walks the positions once and adds, at
  # position i, first ref_supporting_read_count * coeff_reference to scores[i] and then i's read
  # alleles over their footprints.  Same order here: the reference terms of all positions up
  # to i go in right before i's alleles (they touch scores[j] only, so for every j the order
  # is: footprints from positions < j, the reference term of j, footprints from positions >= j).
  ref_terms = _ref_supporting_read_counts(allele_counter).astype(np.float32) * f32(model.coeff_reference)
  done = 0
  coeff = {allelecounter.SUBSTITUTION: f32(model.coeff_substitution),
           allelecounter.SOFT_CLIP: f32(model.coeff_soft_clip),
           allelecounter.INSERTION: f32(model.coeff_insertion),
           allelecounter.DELETION: f32(model.coeff_deletion),
           allelecounter.REFERENCE: f32(model.coeff_reference)}
  for i, ac in _positions_with_read_alleles(allele_counter):
    if i + 1 > done:
      scores[done:i + 1] += ref_terms[done:i + 1]
      done = i + 1
    for allele in ac.read_alleles.values():
      n = len(allele.bases)
      by = f32(allele.count) * coeff[allele.type]
      if allele.type in (allelecounter.SUBSTITUTION, allelecounter.REFERENCE):
        _update_counts(by, i, i + 1, scores)
      elif allele.type in (allelecounter.SOFT_CLIP, allelecounter.INSERTION):
        _update_counts(by, i + 1 - (n - 1), i + n, scores)
      elif allele.type == allelecounter.DELETION:
        _update_counts(by, i + 1, i + n, scores)
      else:
        raise ValueError('Saw an Allele with an unexpected type %r' % (allele.type,))
  scores[done:] += ref_terms[done:]
  return scores


def _make_counter(config: WindowSelectorOptions, ref_reader, reads: Sequence, region: T.Range, table=None):
  """The allele counter of the region expanded by region_expansion_in_bp, with its reads added
  (window_selector.py:40-66).  `table`: the reads already packed (packing.ReadTable), so the counter
  does not pack them again.  The counter is `allelecounter.AlleleCounter`, looked up at call time:
  it counts on the device."""
  expanded = utils.expand(region, config.region_expansion_in_bp, ref_reader.n_bases(region.reference_name))
  counter = allelecounter.AlleleCounter(
      ref_reader, expanded.reference_name, expanded.start, expanded.end,
      min_mapping_quality=config.min_mapq, min_base_quality=config.min_base_quality,
      keep_legacy_behavior=config.keep_legacy_behavior)
  if table is not None:
    counter.add_table(table)
  else:
    for read in reads:
      counter.add(read, 'placeholder_sample_id')
  return counter, expanded


def _candidates_from_counter(config: WindowSelectorOptions, counter, expanded: T.Range) -> List[int]:
  """The positions the model selects, from a counter that holds the region's reads (:68-86)."""
  model_type = config.window_selector_model.model_type
  if model_type == VARIANT_READS:
    conf = config.window_selector_model.variant_reads_model
    counts = variant_reads_candidates_from_allele_counter(counter, config)
    return [expanded.start + i for i, count in enumerate(counts)
            if conf.min_num_supporting_reads <= count <= conf.max_num_supporting_reads]
  if model_type == ALLELE_COUNT_LINEAR:
    conf = config.window_selector_model.allele_count_linear_model
    scores = allele_count_linear_candidates_from_allele_counter(counter, conf)
    return [expanded.start + i for i, score in enumerate(scores) if float(score) > conf.decision_boundary]
  raise ValueError('Unknown enum option "{}" for WindowSelectorModel.model_type'.format(model_type))


def _candidates_from_reads(config: WindowSelectorOptions, ref_reader, reads: Sequence, region: T.Range,
                           table=None) -> List[int]:
  """window_selector._candidates_from_reads (:40-86)."""
  counter, expanded = _make_counter(config, ref_reader, reads, region, table=table)
  return _candidates_from_counter(config, counter, expanded)


def _candidates_to_windows(config: WindowSelectorOptions, candidate_pos: Sequence[int], ref_name: str) -> List[T.Range]:
  """window_selector._candidates_to_windows (:175-212)."""
  windows = []
  d = config.min_windows_distance

  def add(start_pos, end_pos):
    windows.append(utils.make_range(ref_name, start_pos - d, end_pos + d))

  start_pos = end_pos = None
  for pos in sorted(candidate_pos):
    if start_pos is None:
      start_pos = end_pos = pos
    elif pos > end_pos + 2 * d:
      add(start_pos, end_pos)
      start_pos = end_pos = pos
    else:
      end_pos = pos
  if start_pos is not None:
    add(start_pos, end_pos)
  return sorted(windows, key=lambda r: (r.reference_name, r.start, r.end))


def select_windows(config: WindowSelectorOptions, ref_reader, reads: Sequence, region: T.Range,
                   table=None) -> List[T.Range]:
  """window_selector.select_windows (:215-238)."""
  if not reads:
    return []
  if config.realign_all:
    return [region]
  candidates = _candidates_from_reads(config, ref_reader, reads, region, table=table)
  return _candidates_to_windows(config, candidates, region.reference_name)


def select_windows_of_tables(config: WindowSelectorOptions, ref_reader, tables: Sequence, regions: Sequence[T.Range]
                             ) -> List[List[T.Range]]:
  """`select_windows` for the packed read tables of several calling regions: the regions' allele
  counters are filled in ONE device call (AlleleCounter.run_batch: one upload, the kernels back to
  back) before each region's candidates are read off its counter."""
  out: List[List[T.Range]] = [[] for _ in tables]
  if config.realign_all:
    return [[region] if table.n_reads else [] for table, region in zip(tables, regions)]
  jobs = []
  for k, (table, region) in enumerate(zip(tables, regions)):
    if table.n_reads:
      jobs.append((k, region) + _make_counter(config, ref_reader, (), region, table=table))
  run_batch = getattr(allelecounter.AlleleCounter, 'run_batch', None)
  if run_batch is not None:
    run_batch([job[2] for job in jobs])
  for k, region, counter, expanded in jobs:
    out[k] = _candidates_to_windows(config, _candidates_from_counter(config, counter, expanded), region.reference_name)
  return out
