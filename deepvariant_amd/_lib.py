"""ctypes binding of libdvhip.so (the C ABI in include/dvhip.h).

The library is built in-tree by `deepvariant_amd/csrc/Makefile` (hipcc,
--offload-arch=gfx950) -- see `__graft_entry__.build()`.  There is no CPU
fallback: if the shared object is missing, loading raises, and every compute
entry point returns DV_ERR_NO_DEVICE without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DV_LIB_PATH: A/B runs of two builds (tools/); the product always loads the in-tree library
LIB_PATH = os.environ.get('DV_LIB_PATH') or os.path.join(_HERE, 'libdvhip.so')

DV_MAX_CHANNELS = 16
DV_READ_AUX_STRIDE = 8
DV_MEM_HOST, DV_MEM_DEVICE = 0, 1
DV_HP_NONE = -(1 << 31)

DV_OK = 0
DV_ERR_INVALID_ARGUMENT = -1
DV_ERR_UNSUPPORTED = -2
DV_ERR_NO_DEVICE = -3
DV_ERR_HIP = -4
DV_ERR_OUT_OF_MEMORY = -5
DV_ERR_BAD_INPUT = -6

# Every symbol include/dvhip.h declares (checked by tests/test_abi.py).
ABI_SYMBOLS = [
    'dv_last_error', 'dv_abi_version', 'dv_device_count',
    'dv_encoder_create', 'dv_encoder_destroy', 'dv_encode_batch', 'dv_base_aux_plane', 'dv_flow_channel_pixels',
    'dv_downsample_with_partition_mins',
    'dv_downsample_indices', 'dv_validate_batch', 'dv_query_reads', 'dv_crc32c',
    'dv_model_create', 'dv_model_destroy', 'dv_model_num_params', 'dv_model_conv_macs',
    'dv_model_num_layers', 'dv_model_layer_info', 'dv_model_load_weights', 'dv_model_calibrate', 'dv_model_apply_corrections',
    'dv_model_num_ops', 'dv_model_op_label', 'dv_model_probe_rounding', 'dv_model_set_blank_skip', 'dv_model_blank_thresholds', 'dv_model_is_precise',
    'dv_model_infer', 'dv_model_infer_rows', 'dv_model_graph_stats', 'dv_model_debug_tensor', 'dv_set_profiling', 'dv_profile_ms',
    'dv_last_profile_count',
    'dv_bam_read_region', 'dv_read_table_fill_batch', 'dv_read_table_name',
    'dv_read_table_names', 'dv_read_table_ends', 'dv_read_table_free',
    'dv_cram_read_region', 'dv_cram_header', 'dv_read_table_aux_planes',
    'dv_pack_region', 'dv_packed_region_fill_batch', 'dv_packed_region_items',
    'dv_packed_region_free',
    'dv_aligner_create', 'dv_aligner_destroy', 'dv_aligner_set_reference',
    'dv_aligner_set_haplotypes', 'dv_aligner_set_reads', 'dv_aligner_align_reads',
    'dv_aligner_stage', 'dv_aligner_fast_align', 'dv_aligner_haplotype_info',
    'dv_aligner_read_alignment', 'dv_aligner_merge_alignment', 'dv_aligner_is_normalized',
    'dv_aligner_score_threshold', 'dv_aligner_kmer_occurrences', 'dv_positions_map',
    'dv_merge_cigar_op', 'dv_local_align', 'dv_local_align_many',
    'dv_debruijn_build', 'dv_debruijn_destroy', 'dv_debruijn_kmer_size', 'dv_debruijn_haplotypes',
    'dv_debruijn_graphviz', 'dv_realign_regions', 'dv_realign_result_free', 'dv_phase_reads',
    'dv_count_alleles', 'dv_count_alleles_batch', 'dv_allele_counts_arrays', 'dv_allele_counts_free', 'dv_merge_alt_channels',
]


# dv_ref_fetch_fn (include/dvhip.h): the reference bases a CRAM written against an external FASTA leaves out
REF_FETCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_char),
                           C.POINTER(C.c_int64))


class DvError(RuntimeError):
  """A non-zero dv_status; `.status` carries the code."""

  def __init__(self, status, message):
    super().__init__('libdvhip status %d: %s' % (status, message))
    self.status = status


class DvEncoderOptions(C.Structure):
  _fields_ = [
      ('width', C.c_int32), ('height', C.c_int32),
      ('reference_band_height', C.c_int32), ('n_channels', C.c_int32),
      ('channels', C.c_int32 * DV_MAX_CHANNELS),
      ('base_color_offset_a_and_g', C.c_int32),
      ('base_color_offset_t_and_c', C.c_int32),
      ('base_color_stride', C.c_int32),
      ('allele_supporting_read_alpha', C.c_float),
      ('allele_unsupporting_read_alpha', C.c_float),
      ('other_allele_supporting_read_alpha', C.c_float),
      ('reference_matching_read_alpha', C.c_float),
      ('reference_mismatching_read_alpha', C.c_float),
      ('indel_anchoring_base_char', C.c_int32),
      ('reference_base_quality', C.c_int32),
      ('positive_strand_color', C.c_int32),
      ('negative_strand_color', C.c_int32),
      ('base_quality_cap', C.c_int32), ('mapping_quality_cap', C.c_int32),
      ('min_base_quality', C.c_int32), ('min_mapping_quality', C.c_int32),
      ('random_seed', C.c_uint32),
      ('sort_by_haplotypes', C.c_int32),
      ('hp_tag_for_assembly_polishing', C.c_int32),
      ('sort_by_alt_allele_support', C.c_int32),
      ('min_non_zero_allele_frequency', C.c_float),
  ]


class DvBatch(C.Structure):
  _fields_ = [
      ('memory', C.c_int32),
      ('n_reads', C.c_int32),
      ('read_pos', C.c_void_p), ('read_sort_pos', C.c_void_p),
      ('read_seq_off', C.c_void_p), ('read_cigar_off', C.c_void_p),
      ('read_mapq', C.c_void_p), ('read_flags', C.c_void_p),
      ('read_frag_len', C.c_void_p), ('read_hp', C.c_void_p),
      ('read_name_rank', C.c_void_p), ('read_aux', C.c_void_p),
      ('bases', C.c_void_p), ('quals', C.c_void_p),
      ('mod_5mc', C.c_void_p), ('mod_6ma', C.c_void_p),
      ('cigar', C.c_void_p),
      ('n_bases', C.c_uint32), ('n_cigar', C.c_uint32),
      ('n_items', C.c_int32),
      ('item_variant_start', C.c_void_p), ('item_image_start', C.c_void_p),
      ('item_ref_idx', C.c_void_p), ('item_list_off', C.c_void_p),
      ('item_height', C.c_void_p), ('item_out_off', C.c_void_p),
      ('item_blank_mask', C.c_void_p), ('item_mean_coverage', C.c_void_p),
      ('ref_windows', C.c_void_p), ('n_ref_windows', C.c_uint32),
      ('list_read', C.c_void_p), ('list_code', C.c_void_p),
      ('list_group', C.c_void_p), ('list_aux', C.c_void_p),
      ('n_list', C.c_uint32), ('max_list_len', C.c_uint32),
      ('base_aux0', C.c_void_p), ('base_aux1', C.c_void_p),
      ('ref_aux0', C.c_void_p), ('ref_aux1', C.c_void_p), ('ref_aux2', C.c_void_p),
      ('base_aux2', C.c_void_p),        # ABI v6
      ('max_cigar_ops', C.c_uint32), ('max_item_height', C.c_uint32),   # ABI v7: LDS sizing hints, 0 = unknown
  ]


class DvReadRequirements(C.Structure):
  _fields_ = [('keep_duplicates', C.c_int32),
              ('keep_failed_vendor_quality_checks', C.c_int32),
              ('keep_secondary_alignments', C.c_int32),
              ('keep_supplementary_alignments', C.c_int32),
              ('keep_improperly_placed', C.c_int32),
              ('min_mapping_quality', C.c_int32),
              ('use_original_base_quality_scores', C.c_int32),
              ('parse_base_modifications', C.c_int32),      # ABI v6: MM / ML / MN -> 5mC / 6mA planes
              ('parse_flow_tags', C.c_int32)]               # ABI v6: tp / t0 planes


class DvPackReads(C.Structure):
  _fields_ = [('n_reads', C.c_int32), ('read_pos', C.c_void_p), ('read_end', C.c_void_p),
              ('names', C.c_void_p), ('name_off', C.c_void_p), ('read_number', C.c_void_p)]


class DvPackOptions(C.Structure):
  _fields_ = [('width', C.c_int32), ('read_overlap_buffer_bp', C.c_int32),
              ('pileup_height', C.c_int32), ('n_threads', C.c_int32), ('example_bytes', C.c_uint64)]


class DvPackCandidate(C.Structure):
  _fields_ = [('start', C.c_int64), ('end', C.c_int64), ('n_alts', C.c_int32),
              ('ref_idx', C.c_int32), ('first_combo', C.c_uint32), ('n_combos', C.c_uint32),
              ('first_support', C.c_uint32), ('n_support', C.c_uint32)]


class DvAlignerOptions(C.Structure):
  _fields_ = [('match', C.c_int32), ('mismatch', C.c_int32), ('gap_open', C.c_int32),
              ('gap_extend', C.c_int32), ('kmer_size', C.c_int32), ('read_size', C.c_int32),
              ('max_num_of_mismatches', C.c_int32),
              ('realignment_similarity_threshold', C.c_double),
              ('force_alignment', C.c_int32), ('normalize_reads', C.c_int32),
              ('ref_prefix_len', C.c_int32), ('ref_suffix_len', C.c_int32)]


class DvRealignedRead(C.Structure):
  _fields_ = [('status', C.c_int32), ('n_cigar', C.c_int32), ('position', C.c_int64),
              ('cigar_off', C.c_uint32), ('reserved', C.c_uint32)]


class DvReadAlignment(C.Structure):
  _fields_ = [('position', C.c_int32), ('score', C.c_int32), ('cigar', C.c_char * 120)]


class DvLocalAlignment(C.Structure):
  _fields_ = [('score', C.c_int32), ('ref_begin', C.c_int32), ('ref_end', C.c_int32),
              ('query_begin', C.c_int32), ('query_end', C.c_int32), ('mismatches', C.c_int32),
              ('cigar', C.c_char * 512)]


class DvDebruijnOptions(C.Structure):
  _fields_ = [(n, C.c_int32) for n in ('min_k', 'max_k', 'step_k', 'min_mapq', 'min_base_quality',
                                       'min_edge_weight', 'max_num_paths', 'disable_graph_pruning')]


class DvRealignRegion(C.Structure):
  _fields_ = [('bases', C.c_void_p), ('quals', C.c_void_p), ('n_bases', C.c_int64), ('read_seq_off', C.c_void_p),
              ('read_mapq', C.c_void_p), ('read_start', C.c_void_p), ('read_end', C.c_void_p),
              ('n_reads', C.c_int32), ('n_windows', C.c_int32), ('window_start', C.c_void_p),
              ('window_end', C.c_void_p), ('ref', C.c_char_p), ('ref_start', C.c_int64), ('ref_len', C.c_int64),
              ('contig_len', C.c_int64)]


class DvRealignOptions(C.Structure):
  _fields_ = [('dbg', DvDebruijnOptions), ('aln', DvAlignerOptions), ('ref_align_margin', C.c_int32),
              ('n_threads', C.c_int32)]


class DvRealignOutput(C.Structure):
  _fields_ = [('region_row_off', C.POINTER(C.c_int64)), ('order', C.POINTER(C.c_int32)),
              ('status', C.POINTER(C.c_int32)), ('position', C.POINTER(C.c_int64)),
              ('cigar_off', C.POINTER(C.c_int64)), ('cigar', C.POINTER(C.c_uint32)),
              ('region_assembled_off', C.POINTER(C.c_int32)), ('assembled_window', C.POINTER(C.c_int32)),
              ('assembled_hap_off', C.POINTER(C.c_int32)), ('hap_text_off', C.POINTER(C.c_int64)),
              ('hap_text', C.POINTER(C.c_char))]


class DvPhasingAllele(C.Structure):
  _fields_ = [('bases_off', C.c_int64), ('bases_len', C.c_int32), ('is_ref', C.c_int32),
              ('support_off', C.c_int64), ('n_support', C.c_int32), ('reserved', C.c_int32)]


class DvPhasingCandidate(C.Structure):
  _fields_ = [('start', C.c_int64), ('end', C.c_int64), ('allele_off', C.c_int32), ('n_alleles', C.c_int32)]


class DvAltMergeEntry(C.Structure):
  _fields_ = [('example', C.c_int64), ('first_row', C.c_int32), ('rows', C.c_int32),
              ('scratch_alt1', C.c_int64), ('scratch_alt2', C.c_int64)]


class DvAlleleCounterOptions(C.Structure):
  _fields_ = [('interval_start', C.c_int64), ('interval_end', C.c_int64),
              ('reads_interval_start', C.c_int64), ('reads_interval_end', C.c_int64),
              ('ref_bases', C.c_char_p), ('ref_start', C.c_int64), ('n_ref_bases', C.c_int64),
              ('contig_n_bases', C.c_int64), ('min_mapping_quality', C.c_int32),
              ('min_base_quality', C.c_int32), ('keep_legacy_behavior', C.c_int32),
              ('track_ref_reads', C.c_int32), ('candidate_positions', C.c_void_p),
              ('n_candidate_positions', C.c_int32)]


class DvAlleleEvent(C.Structure):
  # include/dvhip.h dv_allele_event (ABI v3): length_type = length (bits 0-27) | AlleleType (28-30) |
  # is_low_quality (bit 31)
  _fields_ = [('position', C.c_int32), ('read', C.c_uint32), ('read_offset', C.c_uint32),
              ('length_type', C.c_uint32)]


class DvModelDesc(C.Structure):
  _fields_ = [('height', C.c_int32), ('width', C.c_int32),
              ('channels', C.c_int32), ('num_classes', C.c_int32),
              ('max_batch', C.c_int32)]


_lib = None


def build(force: bool = False) -> str:
  """Compiles libdvhip.so for gfx950 with the committed Makefile."""
  args = ['make', '-C', os.path.join(_HERE, 'csrc')]
  if force:
    subprocess.check_call(args + ['clean'], stdout=subprocess.DEVNULL)
  subprocess.check_call(args, stdout=subprocess.DEVNULL)
  return LIB_PATH


def lib():
  """Loads libdvhip.so; raises if it has not been built (no fallback)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
          'deepvariant_amd/libdvhip.so is missing: run '
          '`python -c "import __graft_entry__ as g; g.build()"` '
          '(there is no CPU fallback for the HIP hot path)')
    try:
      # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If /opt/rocm's
      # copies (libdvhip.so's DT_NEEDED) get into the process first, torch later finds
      # "No HIP GPUs"; loaded after torch, libdvhip.so binds to torch's runtime by SONAME
      # and both share one device context.  Without torch this is a plain ROCm library.
      import torch  # noqa: F401  pylint: disable=unused-import,g-import-not-at-top
    except ImportError:
      pass
    l = C.CDLL(LIB_PATH)
    l.dv_last_error.restype = C.c_char_p
    l.dv_crc32c.restype = C.c_uint32
    l.dv_crc32c.argtypes = [C.c_void_p, C.c_size_t]
    l.dv_profile_ms.restype = C.c_double
    l.dv_model_num_params.restype = C.c_int64
    l.dv_model_num_params.argtypes = [C.c_void_p]
    l.dv_model_conv_macs.restype = C.c_int64
    l.dv_model_conv_macs.argtypes = [C.c_void_p]
    l.dv_model_num_layers.argtypes = [C.c_void_p]
    l.dv_model_destroy.argtypes = [C.c_void_p]
    l.dv_encoder_destroy.argtypes = [C.c_void_p]
    l.dv_validate_batch.argtypes = [C.c_void_p, C.c_int32]
    l.dv_encode_batch.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
        C.c_void_p]
    l.dv_model_infer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p]
    l.dv_model_load_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    l.dv_model_calibrate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    l.dv_model_apply_corrections.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    l.dv_model_num_ops.argtypes = [C.c_void_p]
    l.dv_model_infer_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    l.dv_model_set_blank_skip.argtypes = [C.c_void_p, C.c_int]
    l.dv_model_is_precise.argtypes = [C.c_void_p]
    l.dv_model_blank_thresholds.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    l.dv_model_op_label.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    l.dv_model_probe_rounding.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_model_graph_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_model_layer_info.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    l.dv_bam_read_region.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_int, C.c_void_p]
    l.dv_cram_read_region.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, REF_FETCH_FN,
                                      C.c_void_p, C.c_int, C.c_void_p]
    l.dv_cram_header.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
    l.dv_read_table_aux_planes.argtypes = [C.c_void_p] * 6
    l.dv_downsample_with_partition_mins.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                    C.c_uint32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    l.dv_base_aux_plane.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    l.dv_flow_channel_pixels.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p]
    l.dv_read_table_fill_batch.argtypes = [C.c_void_p, C.c_void_p]
    l.dv_read_table_name.restype = C.c_char_p
    l.dv_read_table_name.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    l.dv_read_table_names.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    l.dv_pack_region.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6
    l.dv_packed_region_fill_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    l.dv_packed_region_items.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_packed_region_free.argtypes = [C.c_void_p]
    l.dv_read_table_ends.restype = C.c_void_p
    l.dv_read_table_ends.argtypes = [C.c_void_p]
    l.dv_read_table_free.argtypes = [C.c_void_p]
    l.dv_aligner_create.argtypes = [C.c_void_p, C.c_void_p]
    l.dv_aligner_destroy.argtypes = [C.c_void_p]
    l.dv_aligner_destroy.restype = None
    l.dv_aligner_set_reference.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    l.dv_aligner_set_haplotypes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    l.dv_aligner_set_reads.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    l.dv_aligner_align_reads.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_aligner_stage.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    l.dv_aligner_fast_align.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]
    l.dv_aligner_haplotype_info.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32]
    l.dv_aligner_read_alignment.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    l.dv_aligner_merge_alignment.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p,
                                             C.c_void_p, C.c_int32]
    l.dv_aligner_is_normalized.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p]
    l.dv_aligner_score_threshold.argtypes = [C.c_void_p]
    l.dv_aligner_kmer_occurrences.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]
    l.dv_positions_map.argtypes = [C.c_char_p, C.c_int32, C.c_void_p]
    l.dv_merge_cigar_op.argtypes = [C.c_void_p, C.c_int32, C.c_char, C.c_int32, C.c_int32]
    l.dv_local_align.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int32] * 4 + [C.c_void_p]
    l.dv_local_align_many.argtypes = [C.c_char_p, C.c_int32, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
    l.dv_debruijn_build.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    l.dv_debruijn_destroy.argtypes = [C.c_void_p]
    l.dv_debruijn_destroy.restype = None
    l.dv_debruijn_kmer_size.argtypes = [C.c_void_p]
    l.dv_debruijn_haplotypes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_debruijn_graphviz.argtypes = [C.c_void_p, C.c_void_p]
    l.dv_realign_regions.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_realign_result_free.argtypes = [C.c_void_p]
    l.dv_realign_result_free.restype = None
    l.dv_phase_reads.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int32]
    l.dv_count_alleles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_count_alleles_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.dv_allele_counts_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    l.dv_allele_counts_free.argtypes = [C.c_void_p]
    l.dv_allele_counts_free.restype = None
    l.dv_merge_alt_channels.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    _lib = l
  return _lib


def last_error() -> str:
  return lib().dv_last_error().decode()


def check(status: int):
  if status != DV_OK:
    raise DvError(status, last_error())


def try_crc32c(data: bytes) -> Optional[int]:
  if not os.path.exists(LIB_PATH):
    return None
  buf = bytes(data)
  return int(lib().dv_crc32c(buf, len(buf)))


def device_count() -> int:
  return int(lib().dv_device_count())
