"""Drop-in for `deepvariant.python.make_examples_native` on MI355X.

Same class / method names, argument meaning, return values and error behaviour
as the reference's pybind module
(`deepvariant/python/make_examples_native_pybind.cc:56-108`), i.e. the
`ExamplesGenerator` that `RegionProcessor.writes_examples_in_region` drives
(`deepvariant/make_examples_core.py:1893-2013`).  What the reference does per
candidate on one CPU thread (`deepvariant/make_examples_native.cc:632-736`) is
done here for a whole region in ONE `dv_encode_batch` launch: every
(candidate x alt-allele combination x sample) is one item of a packed batch.

Written in Python because the reference's callers hand over Python protobuf
objects and protoc is unavailable here; everything below the packed batch is
the C ABI (include/dvhip.h).  Not reproduced (explicit errors, never silent):
alt-aligned pileups (`diff_channels` / `base_channels` / `rows` / `single_row`
need the FastPassAligner), `trim_reads_for_pileup`,
`stream_examples`.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from deepvariant_amd import dv_types as T
from deepvariant_amd import genomics_io
from deepvariant_amd import packing
from deepvariant_amd import protowire as pw
from deepvariant_amd import tfrecord
from deepvariant_amd.pileup_image_native import PileupImageEncoderNative, _Encoder

DEEP_VARIANT_VERSION = '1.10.0'  # deepvariant/dv_vcf_constants.py:36

# EncodedVariantType, deepvariant/make_examples_native.h
K_UNKNOWN, K_SNP, K_INDEL = 0, 1, 2


@dataclasses.dataclass
class VariantLabel:
  """make_examples_native.h VariantLabel (is_confident, variant, genotype, is_denovo)."""
  is_confident: bool
  variant: T.Variant
  genotype: List[int]
  is_denovo: bool = False

  def label_for_alt_alleles(self, alt_indices_set) -> int:
    # VariantLabel::LabelForAltAlleles, make_examples_native.cc:814-828
    v = sum(1 for g in self.genotype if g != 0 and (g - 1) in alt_indices_set)
    if not 0 <= v <= 2:
      raise ValueError('label_value out of range')
    return v


def encoded_variant_type(variant) -> int:
  """EncodedVariantType, make_examples_native.cc:301-321."""
  alts = list(variant.alternate_bases)
  if len(variant.reference_bases) == 1 and len(alts) >= 1:
    if all(len(a) == 1 for a in alts):
      return K_SNP
  if len(variant.reference_bases) > 1:
    return K_INDEL
  if any(len(a) > 1 for a in alts):
    return K_INDEL
  return K_UNKNOWN


def alt_allele_combinations(candidate, multi_allelic_mode: int
                            ) -> List[List[str]]:
  """AltAlleleCombinations[FromIndices], make_examples_native.cc:191-267."""
  variant = candidate.variant
  alts = list(variant.alternate_bases)
  explicit = list(getattr(candidate, 'make_examples_alt_allele_indices', []))
  if multi_allelic_mode == T.MultiAllelicMode.UNSPECIFIED:
    raise ValueError('multi_allelic_mode cannot be UNSPECIFIED')
  if explicit:
    out = []
    for idx in explicit:
      indices = list(idx.indices)
      if multi_allelic_mode == T.MultiAllelicMode.NO_HET_ALT_IMAGES:
        if len(indices) == 1:
          out.append([alts[indices[0]]])
      else:
        out.append([alts[i] for i in indices])
    return out
  if multi_allelic_mode == T.MultiAllelicMode.NO_HET_ALT_IMAGES:
    return [[a] for a in alts]
  if multi_allelic_mode != T.MultiAllelicMode.ADD_HET_ALT_IMAGES:
    raise ValueError('Unknown value is specified for PileupImageOptions')
  alleles = [variant.reference_bases] + alts
  out = []
  for i in range(len(alleles)):
    for j in range(i + 1, len(alleles)):
      combo = []
      if i > 0:  # the ref allele is not used in combinations
        combo.append(alleles[i])
      combo.append(alleles[j])
      out.append(combo)
  return out


def encode_alt_alleles(variant, alt_combination) -> Tuple[bytes, set]:
  """EncodeAltAlleles, make_examples_native.cc:350-374."""
  index_of = {}
  for i, alt in enumerate(variant.alternate_bases):
    index_of[alt] = i  # later duplicates overwrite, like the flat_hash_map
  indices = [index_of.get(a, 0) for a in alt_combination]
  return pw.encode_alt_allele_indices(indices), set(indices)


def get_reference_bases_for_pileup(ref_reader, variant, width: int) -> str:
  """GetReferenceBasesForPileup, make_examples_native.cc:514-538: N-padded."""
  half = (width - 1) // 2
  n_bases = ref_reader.n_bases(variant.reference_name)
  start = variant.start - half
  end = start + width
  bases = ref_reader.get_bases(variant.reference_name, max(0, start),
                               min(n_bases, end))
  if start < 0:
    bases = 'N' * (-start) + bases
  if end > n_bases:
    bases = bases + 'N' * (end - n_bases)
  return bases


def calculate_pileup_image_height(options) -> int:
  """CalculatePileupImageHeight, pileup_image_native.cc:220-240."""
  total = 0
  for so in options.sample_options:
    mode = so.alt_aligned_pileup or options.pic_options.alt_aligned_pileup
    mult = 3 if mode == 'rows' else 2 if mode == 'single_row' else 1
    total += so.pileup_height * mult
  return total


# realigner flags' defaults (deepvariant/realigner/realigner.py:168-240), which
# RealignReadsToHaplotype inherits through options.realigner_options.aln_config
DEFAULT_ALN_CONFIG = dict(match=4, mismatch=6, gap_open=8, gap_extend=2, kmer_size=32,
                          max_num_of_mismatches=2, realignment_similarity_threshold=0.16934)


class ExamplesGenerator:
  """`ExamplesGenerator(options: MakeExamplesOptions, example_filenames:
  dict[role, path], test_mode=False)` -- make_examples_native.h:154-278."""

  def __init__(self, options, example_filenames: Dict[str, str],
               test_mode: bool = False, device: int = 0, ref_reader=None):
    self._options = options
    pic = options.pic_options
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    self._alt_mode = aap.get_alt_aligned_pileup(pic.alt_aligned_pileup or aap.NONE)
    if getattr(options, 'stream_examples', False):
      raise NotImplementedError('stream_examples is not supported yet')
    self._encoder_api = PileupImageEncoderNative(pic, device=device)
    self._chan_enums = packing.channel_enums(pic)
    self._half_width = (pic.width - 1) // 2
    self._samples = {so.role: so for so in options.sample_options}
    for so in options.sample_options:
      aap.get_alt_aligned_pileup(so.alt_aligned_pileup or aap.NONE)   # unknown names are fatal
    self._height = calculate_pileup_image_height(options)
    self._labels: List[Optional[VariantLabel]] = []
    self._writers: Dict[str, tfrecord.Writer] = {}
    self._ref = ref_reader
    self._device_encoder: Optional[_Encoder] = None
    self._device = device
    if test_mode:
      return
    if self._ref is None:
      # keep_true_case=false -> upper case (make_examples_native.cc:126-132)
      self._ref = genomics_io.FastaReader(options.reference_filename)
    for role, path in example_filenames.items():
      self._writers[role] = tfrecord.Writer(path)

  # ---------------------------------------------------------------- pybind API
  def append_label(self, label: VariantLabel):
    self._labels.append(label)

  def signal_shard_finished(self):
    for w in self._writers.values():
      w.close()
    self._writers = {}

  def write_examples_in_region(self, candidates: Sequence,
                               reads_per_sample: Sequence[Sequence],
                               sample_order: Sequence[int], role: str,
                               mean_coverage_per_sample: Sequence[float]
                               ) -> Tuple[Dict[str, int], List[int]]:
    """-> (stats, image_shape); make_examples_native.cc:742-793."""
    if self._labels and len(self._labels) != len(candidates):
      raise ValueError('labels_.size() != candidates.size()')  # CHECK, :751
    if role not in self._samples:
      raise ValueError('Role %s not found.' % role)
    if role not in self._writers:
      raise ValueError('Role %s does not have a writer.' % role)
    stats: Dict[str, int] = {}
    examples, image_shape = self.encode_region(
        candidates, reads_per_sample, sample_order, mean_coverage_per_sample,
        stats, role=role)
    writer = self._writers[role]
    for ex in examples:
      writer.write(ex)
    self._labels = []
    return stats, image_shape

  # --------------------------------------------------------------- internals
  def _trim_region_reads(self, candidates, reads_per_sample, sample_order, tables, wanted=None):
    """trim_reads_for_pileup / keep_only_window_spanning_reads
    (make_examples_native.cc:655-685): every candidate gets ITS OWN copies of the reads it
    overlaps, cut to the pileup window (TrimReads), rows still sorted by the untrimmed
    starts.  -> (per-sample read tables of the trimmed copies, {(candidate, sample): index
    range})."""
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    pic = self._options.pic_options
    trimmed = [[] for _ in reads_per_sample]
    starts = [[] for _ in reads_per_sample]
    ranges = {}
    for ci, cand in enumerate(candidates):
      variant = cand.variant
      if wanted is not None and not wanted[ci]:
        continue
      if not get_reference_bases_for_pileup(self._ref, variant, pic.width):
        continue
      q0 = variant.start - pic.read_overlap_buffer_bp
      q1 = variant.end + pic.read_overlap_buffer_bp
      r0, r1 = aap.calculate_alignment_region(variant, self._half_width,
                                              self._ref.n_bases(variant.reference_name))
      for s in sample_order:
        so = self._options.sample_options[s]
        if isinstance(reads_per_sample[s], packing.ReadTable):
          raise NotImplementedError('trim_reads_for_pileup needs Read objects, not a packed table')
        min_overlap = pic.width if so.keep_only_window_spanning_reads \
            else aap.K_DEFAULT_MINIMUM_READ_OVERLAP
        overlapping = [reads_per_sample[s][int(k)] for k in tables[s].query(q0, q1)]
        kept, original = aap.trim_reads(overlapping, r0, r1, min_overlap)
        lo = len(trimmed[s])
        trimmed[s].extend(kept)
        starts[s].extend(original)
        ranges[(ci, s)] = (lo, lo + len(kept))
    new_tables = [self._table_of(trimmed[s], starts[s]) for s in range(len(reads_per_sample))]
    self._trimmed_reads, self._trimmed_starts = trimmed, starts
    return new_tables, ranges

  def _non_uniform(self, cand, table, idx_local, so):
    """SampleOptions.use_non_uniform_downsampling: the reads of one image that stay (every BuildPileupForOneSample
    call of the sample -- reference image and alt-aligned images alike -- samples its own read list,
    pileup_image_native.cc:326-341)."""
    if not so.use_non_uniform_downsampling:
      return idx_local
    pic = self._options.pic_options
    kept = packing.non_uniform_sample(cand, table, idx_local, so.pileup_height - pic.reference_band_height,
                                      so.non_uniform_downsampling_threshold, pic.random_seed)
    return idx_local if kept is None else np.asarray(idx_local)[kept]

  def _table_of(self, reads, sort_positions):
    t = packing.ReadTable.from_reads(
        reads, alignment_positions=sort_positions or None, need_aux=self._encoder_api._need_aux,
        need_seq_aux=self._encoder_api._need_seq_aux)
    if t.read_sort_pos is None:
      t.read_sort_pos = t.read_pos.copy()
    return t

  def _realign_for_alt_images(self, candidates, need_alt, sample_order, trim_ranges):
    """CreateAltAlignedImages' host part (make_examples_native.cc:553-600): per candidate,
    sample and alt allele, the haplotype (reference prefix + alt + suffix) and the trimmed
    reads realigned to it.  -> (per-sample tables of the realigned reads,
    {(candidate, sample, alt): (lo, hi, haplotype window)})."""
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    from deepvariant_amd import fast_pass_aligner as fpa
    pic = self._options.pic_options
    cfg = dict(getattr(self._options, 'aln_config', None) or DEFAULT_ALN_CONFIG)
    n_samples = len(self._trimmed_reads)
    reads_out = [[] for _ in range(n_samples)]
    starts_out = [[] for _ in range(n_samples)]
    ranges = {}
    for ci, cand in enumerate(candidates):
      if not need_alt[ci]:
        continue
      variant = cand.variant
      for s in sample_order:
        if (ci, s) not in trim_ranges or not self._sample_needs_alt(self._options.sample_options[s]):
          continue
        lo, hi = trim_ranges[(ci, s)]
        trimmed = self._trimmed_reads[s][lo:hi]
        starts = self._trimmed_starts[s][lo:hi]
        for alt in variant.alternate_bases:
          haplotype, ref_start, ref_end = aap.create_haplotype(self._ref, variant, alt, self._half_width)
          if len(haplotype) < pic.width:
            continue          # too close to the contig start: no alt-aligned pixels (:575-581)
          realigned = fpa.realign_reads_to_haplotype(haplotype, trimmed, variant.reference_name,
                                                     ref_start, ref_end, self._ref, cfg)
          a = len(reads_out[s])
          for read, start in zip(realigned, starts):
            if read is not None and read.aligned_sequence:
              reads_out[s].append(read)
              starts_out[s].append(start)
          ranges[(ci, s, alt)] = (a, len(reads_out[s]), haplotype[:pic.width])
    return [self._table_of(reads_out[s], starts_out[s]) for s in range(n_samples)], ranges

  @staticmethod
  def _sample_needs_alt(so) -> bool:
    """SampleNeedsAltAlignment (make_examples_native.cc:476-498): a sample that blanks any
    alt-aligned channel gets no alt alignment."""
    return not any(e in (9, 10, 20, 21) for e in so.channels_enum_to_blank)

  def _plan_region(self, candidates, reads_per_sample, sample_order, mean_coverage_per_sample,
                   role=None):
    """-> (PackedBatch, [(candidate index, alt combination)], image_shape): everything
    CreateAndWriteExamplesForCandidate decides before pixels are drawn."""
    pic = self._options.pic_options
    width = pic.width
    n_chan_total = len(pic.channels)  # image_shape[2], make_examples_native.cc:399
    c_enc = len(self._chan_enums)
    image_shape = [self._height, width, n_chan_total]
    row_bytes = width * n_chan_total
    example_bytes = self._height * row_bytes

    # One ReadTable per sample; the encoder batch concatenates them.
    # (a sample may also arrive already packed -- packing.ReadTable.from_bam -- instead of
    # as a list of Read protos)
    tables = []
    for reads in reads_per_sample:
      if isinstance(reads, packing.ReadTable):
        if ((self._encoder_api._need_aux and reads.read_aux is None) or
            any(ch and getattr(reads, 'base_aux%d' % k) is None
                for k, ch in enumerate(self._encoder_api._need_seq_aux))):
          raise ValueError('this channel set needs per-read aux pixels; pack the reads with '
                           'ReadTable.from_reads(need_aux=True)')
        tables.append(reads)
      else:
        tables.append(packing.ReadTable.from_reads(reads, need_aux=self._encoder_api._need_aux,
                                                   need_seq_aux=self._encoder_api._need_seq_aux))
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    role_sample = self._samples.get(role) if role is not None else \
        self._options.sample_options[sample_order[0]]
    # Per candidate (make_examples_native.cc:654-657): trimmed copies of its reads when
    # trim_reads_for_pileup / keep_only_window_spanning_reads is set or the variant gets
    # alt-aligned images.
    need_alt = [self._alt_mode != aap.NONE and aap.need_alt_alignment(pic, c.variant) for c in candidates]
    trim_all = bool(getattr(self._options, 'trim_reads_for_pileup', False) or
                    role_sample.keep_only_window_spanning_reads)
    use_trimmed = [trim_all or na for na in need_alt]
    trim_ranges, alt_ranges = None, {}
    n_samples = len(tables)
    all_tables = list(tables)
    if any(use_trimmed):
      trim_tables, trim_ranges = self._trim_region_reads(candidates, reads_per_sample, sample_order,
                                                         tables, use_trimmed)
      all_tables += trim_tables
      if any(need_alt):
        alt_tables, alt_ranges = self._realign_for_alt_images(candidates, need_alt, sample_order,
                                                              trim_ranges)
        all_tables += alt_tables
      for t in all_tables:
        if t.read_sort_pos is None:
          t.read_sort_pos = t.read_pos.copy()
    # Fast path: one sample, nothing per-item beyond the read list -- the whole region is
    # packed by libdvhip's dv_pack_region (region_packer.cpp) instead of the per-candidate
    # Python below (DV_PY_PACKER=1 keeps the Python path; tests compare the two).
    if (trim_ranges is None and len(sample_order) == 1 and len(tables) == 1 and
        not self._encoder_api._need_list_aux and os.environ.get('DV_PY_PACKER') is None):
      so = self._options.sample_options[sample_order[0]]
      if not so.channels_enum_to_blank and not so.variant_types_to_blank and not so.use_non_uniform_downsampling:
        windows = [get_reference_bases_for_pileup(self._ref, c.variant, width) for c in candidates]
        combos = [list(alt_allele_combinations(c, pic.multi_allelic_mode)) if w else []
                  for c, w in zip(candidates, windows)]
        batch, plan = packing.pack_region_native(
            tables[0], candidates, combos, windows, width, pic.read_overlap_buffer_bp,
            so.pileup_height, example_bytes, use_groups=bool(pic.sort_by_alt_allele_support))
        batch.use_ref_aux = self._encoder_api._need_ref_aux
        mc = float(mean_coverage_per_sample[sample_order[0]])
        batch.item_mean_coverage = [mc] * batch.n_items
        self._alt_plan = []
        return batch, plan, image_shape
    merged, sample_base = _concat_tables(all_tables)
    batch = packing.PackedBatch(table=merged, width=width,
                                use_ref_aux=self._encoder_api._need_ref_aux)
    plan = []  # (candidate index, alt_combination)
    # alt-aligned images that end up in CHANNELS are drawn into scratch rows behind the
    # examples and copied over afterwards (_merge_alt_channels); images that end up as extra
    # ROWS are drawn in place.  alt_plan: (example, first row, rows, [scratch slot of alt 1, of alt 2])
    alt_plan = []
    pending = []   # alt items: (table, idx_local, base, cand, combo, haplotype window, so, mc, vtype, target)
    for ci, cand in enumerate(candidates):
      variant = cand.variant
      ref_bases = get_reference_bases_for_pileup(self._ref, variant, width)
      if not ref_bases:
        continue  # edge of the contig (make_examples_native.cc:650-653)
      ref_idx = batch.add_ref_window(ref_bases)
      q0 = variant.start - pic.read_overlap_buffer_bp
      q1 = variant.end + pic.read_overlap_buffer_bp
      vtype = encoded_variant_type(variant)
      for combo in alt_allele_combinations(cand, pic.multi_allelic_mode):
        out_off = len(plan) * example_bytes
        row0 = 0
        for s in sample_order:
          so = self._options.sample_options[s]
          if use_trimmed[ci]:
            table, base = all_tables[n_samples + s], sample_base[n_samples + s]
            idx_local = np.arange(*trim_ranges[(ci, s)], dtype=np.int64)
          else:
            table, base = tables[s], sample_base[s]
            idx_local = table.query(q0, q1)
          idx_local = self._non_uniform(cand, table, idx_local, so)
          blank = list(so.channels_enum_to_blank)
          if vtype in _types_to_blank(so):
            blank = list(T.DeepVariantChannelEnum)
          mask = packing.blank_mask_for(self._chan_enums, blank)
          mc = float(mean_coverage_per_sample[s])
          batch.add_item(
              variant.start, variant.start - self._half_width, ref_idx,
              idx_local + base,
              packing.support_codes(cand, combo, table, idx_local),
              height=so.pileup_height, out_off=out_off,
              blank_mask=mask, mean_coverage=mc,
              groups=(packing.allele_groups(cand, table, idx_local)
                      if pic.sort_by_alt_allele_support else None),
              list_aux=self._encoder_api._list_aux(cand, combo, table, idx_local))
          sample_mode = aap.get_sample_alt_aligned_pileup(self._alt_mode, so.alt_aligned_pileup or '')
          row_slots = aap.get_alt_image_row_indices(sample_mode, combo)
          if need_alt[ci] and self._sample_needs_alt(so):
            slots = [None, None]
            for k, alt in enumerate(combo[:2]):   # CHECK_LE(alt_combination.size(), 2)
              if (ci, s, alt) not in alt_ranges:
                break                              # haplotype shorter than the window: stop (:575-581)
              lo, hi, hap_window = alt_ranges[(ci, s, alt)]
              slots[k] = len(pending)
              pending.append(dict(
                  table=all_tables[2 * n_samples + s], base=sample_base[2 * n_samples + s],
                  idx=np.arange(lo, hi, dtype=np.int64), cand=cand, combo=combo, window=hap_window,
                  so=so, mc=mc, mask=mask, variant=variant,
                  # rows / single_row: drawn in place, (1 + position) blocks below the reference image
                  in_place=(out_off + (1 + row_slots.index(k)) * so.pileup_height * row_bytes
                            if k in row_slots else None)))
            if self._alt_mode in (aap.BASE_CHANNELS, aap.DIFF_CHANNELS):
              alt_plan.append((len(plan), row0, so.pileup_height, slots))
          block_rows = so.pileup_height * (1 + len(row_slots))
          out_off += block_rows * row_bytes
          row0 += block_rows
        plan.append((ci, combo))
    scratch0 = len(plan) * example_bytes
    n_scratch = 0
    for item in pending:
      so = item['so']
      if item['in_place'] is not None and self._alt_mode not in (aap.BASE_CHANNELS, aap.DIFF_CHANNELS):
        out_off = item['in_place']
        item['scratch'] = None
      else:
        # channel modes read the alt image from scratch; an image that is ALSO a row block
        # (sample-level rows under a global channel mode) is drawn twice, once per place
        out_off = scratch0 + n_scratch * self._max_sample_height() * row_bytes
        item['scratch'] = n_scratch
        n_scratch += 1
      for target in ([out_off] + ([item['in_place']] if item['scratch'] is not None and
                                  item['in_place'] is not None else [])):
        idx_local, table = item['idx'], item['table']
        idx_local = self._non_uniform(item['cand'], table, idx_local, so)
        batch.add_item(
            item['variant'].start, item['variant'].start - self._half_width,
            batch.add_ref_window(item['window']), idx_local + item['base'],
            packing.support_codes(item['cand'], item['combo'], table, idx_local),
            height=so.pileup_height, out_off=target, blank_mask=item['mask'], mean_coverage=item['mc'],
            groups=(packing.allele_groups(item['cand'], table, idx_local)
                    if pic.sort_by_alt_allele_support else None),
            list_aux=self._encoder_api._list_aux(item['cand'], item['combo'], table, idx_local))
    self._alt_plan = [(ex, row0, rows, [None if k is None else pending[k]['scratch'] for k in slots])
                      for ex, row0, rows, slots in alt_plan]
    self._alt_scratch0 = scratch0
    return batch, plan, image_shape

  def _max_sample_height(self) -> int:
    return max(so.pileup_height for so in self._options.sample_options)

  def _merge_alt_channels_device(self, flat, image_shape) -> None:
    """_merge_alt_channels on images that stay in HBM (dv_merge_alt_channels)."""
    import ctypes as C
    import torch
    from deepvariant_amd import _lib
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    h_img, w, c = image_shape
    entries = [(ex, row0, rows, slots) for ex, row0, rows, slots in self._alt_plan if slots[0] is not None]
    if not entries:
      return
    arr = (_lib.DvAltMergeEntry * len(entries))()
    for k, (ex, row0, rows, slots) in enumerate(entries):
      arr[k].example, arr[k].first_row, arr[k].rows = ex, row0, rows
      arr[k].scratch_alt1 = slots[0]
      arr[k].scratch_alt2 = slots[1] if slots[1] is not None else -1
    _lib.check(_lib.lib().dv_merge_alt_channels(
        flat.data_ptr(), self._alt_scratch0, h_img * w * c, self._max_sample_height() * w * c, w, c,
        len(self._chan_enums), 5 if self._alt_mode == aap.DIFF_CHANNELS else 0, arr, len(entries),
        C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)))

  def _merge_alt_channels(self, images: np.ndarray, n_examples: int, image_shape) -> None:
    """FillPileupArray's channel modes (pileup_image_native.h:246-271) on the encoder's
    output: the two trailing channels of every reference-image row block come from
    channel 5 (diff_channels) / 0 (base_channels) of alt image 1 and alt image 2, zero when
    alt 1 is missing, alt 1's again when alt 2 is missing; rows are zipped by index."""
    from deepvariant_amd import alt_aligned_pileup_lib as aap
    if not self._alt_plan:
      return
    h_img, w, c = image_shape
    c_enc = len(self._chan_enums)
    ch = 5 if self._alt_mode == aap.DIFF_CHANNELS else 0
    hs = self._max_sample_height()
    examples = images[:n_examples * h_img * w * c].reshape(n_examples, h_img, w, c)
    scratch = images[self._alt_scratch0:]
    scratch = scratch[:len(scratch) // (hs * w * c) * (hs * w * c)].reshape(-1, hs, w, c)
    for ex, row0, rows, slots in self._alt_plan:
      if slots[0] is None:
        continue
      block = examples[ex, row0:row0 + rows]
      block[:, :, c_enc] = scratch[slots[0], :rows, :, ch]
      block[:, :, c_enc + 1] = scratch[slots[1] if slots[1] is not None else slots[0], :rows, :, ch]

  def encode_region(self, candidates, reads_per_sample, sample_order,
                    mean_coverage_per_sample, stats, role=None) -> Tuple[List[bytes], List[int]]:
    """All examples of one region: one packed batch, one kernel launch."""
    batch, plan, image_shape = self._plan_region(candidates, reads_per_sample, sample_order,
                                                 mean_coverage_per_sample, role)
    pic = self._options.pic_options
    n_chan_total = len(pic.channels)
    example_bytes = self._height * pic.width * n_chan_total
    if not plan:
      return [], image_shape
    if self._device_encoder is None:
      self._device_encoder = _Encoder(pic, pic.width, self._device)
    if n_chan_total < len(self._chan_enums):
      raise ValueError('num_channels smaller than the encoder channel list')
    # (row-block layouts leave the blocks of missing alt images untouched: zero)
    images, _ = self._device_encoder.encode(batch, n_chan_total, min_bytes=len(plan) * example_bytes)
    self._merge_alt_channels(images, len(plan), image_shape)
    examples = []
    for k, (ci, combo) in enumerate(plan):
      label = self._labels[ci] if self._labels else None
      image = images[k * example_bytes:(k + 1) * example_bytes]
      examples.append(self._encode_example(
          candidates[ci].variant, combo, image, image_shape, label, stats))
    return examples, image_shape

  def call_variants_in_region(self, candidates, reads_per_sample, sample_order,
                              mean_coverage_per_sample, model) -> List[bytes]:
    """Fused make_examples -> call_variants for one region (the
    reference's precedent is fast_pipeline / stream_examples.cc:117-176): the pileup
    tensors are encoded straight into device memory and classified there -- no
    tf.Example, no GZIP, no host copy of the images.  Returns serialised
    CallVariantsOutput protos identical to what call_variants writes for the examples
    write_examples_in_region would have produced (same order)."""
    from deepvariant_amd import call_variants as cv
    images, plan = self.encode_region_on_device(candidates, reads_per_sample, sample_order, mean_coverage_per_sample,
                                                model.input_shape)
    if not plan:
      return []
    gls = cv.round_gls_batch(model(images).cpu().numpy(), 10)
    return self.call_variants_outputs(candidates, plan, gls)

  def call_variants_outputs(self, candidates, plan, gls) -> List[bytes]:
    """CallVariantsOutput records of one region's examples from their (rounded) likelihood rows."""
    from deepvariant_amd import call_variants as cv
    out = []
    for (ci, combo), row in zip(plan, gls):
      variant = candidates[ci].variant
      alt_encoded, _ = encode_alt_alleles(variant, combo)
      out.append(cv.create_cvo(pw.encode_variant(variant), row.tolist(), alt_encoded))
    return out

  def encode_region_on_device(self, candidates, reads_per_sample, sample_order, mean_coverage_per_sample,
                              model_shape):
    """The region's pileup tensors drawn on the device and LEFT there: -> (uint8 [n, H, W, C] CUDA
    tensor or None, [(candidate index, alt combination)]).  call_variants_in_region classifies them
    at once; a region driver may collect several regions' tensors and classify them together
    (make_examples' fused route: one CNN forward per few hundred examples instead of one per region)."""
    import torch  # device memory + stream only
    from deepvariant_amd.device_batch import DeviceBatch
    batch, plan, image_shape = self._plan_region(candidates, reads_per_sample, sample_order,
                                                 mean_coverage_per_sample)
    if not plan:
      return None, []
    if list(image_shape) != list(model_shape):
      raise ValueError('example shape %s != model shape %s' % (image_shape, list(model_shape)))
    pic = self._options.pic_options
    if self._device_encoder is None:
      self._device_encoder = _Encoder(pic, pic.width, self._device)
    dev = torch.device('cuda', self._device)
    dbatch = DeviceBatch(batch, dev, pic.reference_band_height)
    example_bytes = int(np.prod(image_shape))
    n_bytes = max(batch.out_bytes(image_shape[2]), len(plan) * example_bytes)
    # alt-aligned layouts: alt images are items of the same launch (rows in place -- missing
    # ones must read as zero -- or scratch images behind the examples for the channel modes)
    alt = self._alt_mode != 'none'
    flat = (torch.zeros if alt else torch.empty)(n_bytes, dtype=torch.uint8, device=dev)
    images = flat[:len(plan) * example_bytes].view([len(plan)] + list(image_shape))
    dbatch.encode(self._device_encoder, image_shape[2], flat)
    if self._alt_plan:
      self._merge_alt_channels_device(flat, image_shape)
    return images, plan

  def _encode_example(self, variant, alt_combination, image: np.ndarray,
                      image_shape, label, stats) -> bytes:
    """EncodeExample, make_examples_native.cc:388-474."""
    alt_encoded, alt_set = encode_alt_alleles(variant, alt_combination)
    vtype = encoded_variant_type(variant)
    locus = '%s:%d-%d' % (variant.reference_name, variant.start + 1, variant.end)
    src_variant = label.variant if label is not None else variant
    features = {
        'locus': [locus.encode()],
        'variant/encoded': [pw.encode_variant(src_variant)],
        'variant_type': [vtype],
        'alt_allele_indices/encoded': [alt_encoded],
        'image/encoded': [image.tobytes()],
        'image/shape': list(image_shape),
        'sequencing_type': [int(self._options.pic_options.sequencing_type)],
    }
    label_value = 0
    if label is not None:
      label_value = label.label_for_alt_alleles(alt_set)
      features['label'] = [label_value]
      if getattr(self._options, 'denovo_regions_filename', ''):
        features['denovo_label'] = [int(label.is_denovo)]
    _update_stats(vtype, label, label_value, stats)
    return pw.encode_example(features)


def _types_to_blank(so) -> set:
  out = set()
  for v in so.variant_types_to_blank:
    if v == 1:  # SampleOptions.VARIANT_TYPE_SNP
      out.add(K_SNP)
    elif v == 2:  # VARIANT_TYPE_INDEL
      out.add(K_INDEL)
  return out


def _update_stats(vtype, label, label_value, stats):
  """UpdateStats, make_examples_native.cc:331-348."""
  stats['n_examples'] = stats.get('n_examples', 0) + 1
  key = 'n_indels' if vtype == K_INDEL else 'n_snps'
  stats[key] = stats.get(key, 0) + 1
  if label is not None:
    for c in (0, 1, 2):
      stats['n_class_%d' % c] = stats.get('n_class_%d' % c, 0) + int(label_value == c)
    stats['n_non_denovo'] = stats.get('n_non_denovo', 0) + int(not label.is_denovo)
    stats['n_denovo'] = stats.get('n_denovo', 0) + int(label.is_denovo)


def _concat_tables(tables: List[packing.ReadTable]):
  """Concatenates per-sample read tables (read indices shift by sample_base)."""
  if len(tables) == 1:
    return tables[0], [0]
  base, n = [], 0
  for t in tables:
    base.append(n)
    n += t.n_reads
  cat = lambda name, dt: np.concatenate([getattr(t, name) for t in tables]).astype(dt)
  seq_off = [np.zeros(1, np.uint32)]
  cig_off = [np.zeros(1, np.uint32)]
  so = co = 0
  for t in tables:
    seq_off.append(t.read_seq_off[1:] + so)
    cig_off.append(t.read_cigar_off[1:] + co)
    so += int(t.read_seq_off[-1])
    co += int(t.read_cigar_off[-1])
  # name ranks must stay comparable across samples: re-rank the union
  keys = [k for t in tables for k in _rank_keys(t)]
  uniq = {k: i for i, k in enumerate(sorted(set(keys)))}
  ranks = np.array([uniq[k] for k in keys], np.uint32)
  def opt(name):
    """Optional arrays: absent everywhere -> None; absent in some tables (an empty table, reads
    without modification tags) -> zeros of the right length there."""
    have = [getattr(t, name) for t in tables]
    if all(h is None for h in have):
      return None
    like = next(h for h in have if h is not None)
    parts = []
    for t, h in zip(tables, have):
      if h is None:
        if name == 'read_sort_pos':
          h = t.read_pos.astype(like.dtype)
        elif name == 'read_aux':
          h = np.zeros((t.n_reads,) + like.shape[1:], like.dtype)
        else:                                  # per-base arrays
          h = np.zeros(len(t.bases), like.dtype)
      parts.append(h)
    return np.concatenate(parts)
  merged = packing.ReadTable(
      n_reads=n, read_pos=cat('read_pos', np.int32), read_sort_pos=opt('read_sort_pos'),
      read_seq_off=np.concatenate(seq_off).astype(np.uint32),
      read_cigar_off=np.concatenate(cig_off).astype(np.uint32),
      read_mapq=cat('read_mapq', np.uint8), read_flags=cat('read_flags', np.uint8),
      read_frag_len=cat('read_frag_len', np.int32), read_hp=cat('read_hp', np.int32),
      read_name_rank=ranks, read_aux=opt('read_aux'),
      bases=cat('bases', np.uint8), quals=cat('quals', np.uint8),
      mod_5mc=opt('mod_5mc'), mod_6ma=opt('mod_6ma'), cigar=cat('cigar', np.uint32),
      keys=[k for t in tables for k in t.keys],
      read_end=cat('read_end', np.int64),
      base_aux0=opt('base_aux0'), base_aux1=opt('base_aux1'), base_aux2=opt('base_aux2'))
  return merged, base


def _rank_keys(table: packing.ReadTable):
  out = []
  for k in table.keys:
    name, num = k.rsplit('/', 1)
    out.append((name.encode(), int(num)))
  return out


def write_example_info_json(examples_path: str, image_shape, channel_names):
  """`<examples>.example_info.json` (make_examples_core.py:3755-3774)."""
  info = {
      'version': DEEP_VARIANT_VERSION,
      'shape': list(image_shape),
      'channels': [T.CHANNEL_NAME_TO_INFO_ENUM[c] for c in channel_names],
  }
  with open(examples_path + '.example_info.json', 'w') as f:
    json.dump(info, f)
  return info
