"""ctypes wrapper of the CPU oracle (oracle/libdvoracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by anything under deepvariant_amd/.

Accepts proto-shaped objects (deepvariant_amd.dv_types dataclasses or real
protobuf messages with the same attribute names) and hands them to the C++
restatement in encoder_oracle.cpp.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libdvoracle.so')
# The REFERENCE's own encoder sources, compiled from where they lie by oracle/ref_build/Makefile (only where
# /root/reference exists; the built library travels to the GPU box): same C interface, same symbols.
_REF_LIB_PATH = os.path.join(_HERE, '_ref', 'libdvref.so')
_REFERENCE_ROOT = os.environ.get('DV_REFERENCE_ROOT', '/root/reference')
DVO_MAX_CHANNELS = 32


def build(force: bool = False) -> str:
  """Compiles the oracle with the committed Makefile (g++)."""
  srcs = [os.path.join(_HERE, f) for f in ('encoder_oracle.cpp', 'dvo.h', 'packed_adapter.h')]
  if (force or not os.path.exists(_LIB_PATH) or
      os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)):
    subprocess.check_call(['make', '-C', _HERE, 'libdvoracle.so'],
                          stdout=subprocess.DEVNULL)
  return _LIB_PATH


def build_reference() -> Optional[str]:
  """oracle/_ref/libdvref.so: the reference's own pileup_image_native.cc / pileup_channel_lib.cc / channels/*.cc,
  unmodified, behind dvo.h (oracle/ref_build/: mini_protoc.py + shims + dvref_capi.cc).  Built where the
  reference tree is present (make decides what is stale); elsewhere the prebuilt file is used.  None when
  neither exists."""
  if os.path.isdir(os.path.join(_REFERENCE_ROOT, 'deepvariant', 'channels')):
    subprocess.check_call(['make', '-C', os.path.join(_HERE, 'ref_build'), '-j8', 'REF=' + _REFERENCE_ROOT],
                          stdout=subprocess.DEVNULL)
  return _REF_LIB_PATH if os.path.exists(_REF_LIB_PATH) else None


class DvoOptions(C.Structure):
  _fields_ = [
      ('width', C.c_int32), ('height', C.c_int32),
      ('reference_band_height', C.c_int32), ('n_channels', C.c_int32),
      ('channels', C.c_int32 * DVO_MAX_CHANNELS),
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
      ('use_non_uniform_downsampling', C.c_int32), ('non_uniform_downsampling_threshold', C.c_int32),
  ]


class DvoRead(C.Structure):
  _fields_ = [
      ('fragment_name', C.c_char_p), ('read_number', C.c_int32),
      ('position', C.c_int64), ('mapping_quality', C.c_int32),
      ('reverse_strand', C.c_int32), ('supplementary', C.c_int32),
      ('fragment_length', C.c_int32),
      ('seq', C.c_char_p), ('seq_len', C.c_int32),
      ('qual', C.POINTER(C.c_uint8)), ('qual_len', C.c_int32),
      ('cigar_ops', C.POINTER(C.c_int32)),
      ('cigar_lens', C.POINTER(C.c_int64)), ('n_cigar', C.c_int32),
      ('hp_present', C.c_int32), ('hp_n_values', C.c_int32),
      ('hp_is_int', C.c_int32), ('hp_value', C.c_int32),
      ('mod_5mc', C.POINTER(C.c_uint8)), ('mod_5mc_len', C.c_int32),
      ('mod_6ma', C.POINTER(C.c_uint8)), ('mod_6ma_len', C.c_int32),
      ('tp', C.POINTER(C.c_int8)), ('tp_len', C.c_int32),
      ('t0', C.c_char_p), ('t0_len', C.c_int32),
  ]


class DvoCall(C.Structure):
  _fields_ = [
      ('variant_start', C.c_int64),
      ('n_alts', C.c_int32), ('alts', C.POINTER(C.c_char_p)),
      ('n_support', C.c_int32), ('support_alleles', C.POINTER(C.c_char_p)),
      ('support_offsets', C.POINTER(C.c_int32)),
      ('support_names', C.POINTER(C.c_char_p)),
      ('n_af', C.c_int32), ('af_alleles', C.POINTER(C.c_char_p)),
      ('af_values', C.POINTER(C.c_float)),
      ('n_ref_support', C.c_int32),
      ('reference_bases', C.c_char_p),
      ('ref_support_names', C.POINTER(C.c_char_p)),
      ('alt_ps_present', C.c_int32), ('n_alt_ps', C.c_int32),
      ('alt_ps', C.POINTER(C.c_int32)),
      ('n_rejected_alts', C.c_int32), ('rejected_alts', C.POINTER(C.c_char_p)),
      ('n_rejected_support', C.c_int32),
      ('rejected_support_alleles', C.POINTER(C.c_char_p)),
      ('rejected_support_offsets', C.POINTER(C.c_int32)),
      ('rejected_support_names', C.POINTER(C.c_char_p)),
  ]


class DvoPackedBatch(C.Structure):
  _fields_ = [
      ('n_reads', C.c_int32),
      ('read_pos', C.c_void_p), ('read_sort_pos', C.c_void_p),
      ('read_seq_off', C.c_void_p), ('read_cigar_off', C.c_void_p),
      ('read_mapq', C.c_void_p), ('read_flags', C.c_void_p),
      ('read_frag_len', C.c_void_p), ('read_hp', C.c_void_p),
      ('read_name_rank', C.c_void_p),
      ('bases', C.c_void_p), ('quals', C.c_void_p),
      ('mod_5mc', C.c_void_p), ('mod_6ma', C.c_void_p),
      ('cigar', C.c_void_p),
      ('n_items', C.c_int32),
      ('item_variant_start', C.c_void_p), ('item_image_start', C.c_void_p),
      ('item_ref_idx', C.c_void_p), ('item_list_off', C.c_void_p),
      ('item_height', C.c_void_p), ('item_out_off', C.c_void_p),
      ('ref_windows', C.c_void_p),
      ('list_read', C.c_void_p), ('list_code', C.c_void_p),
      ('list_group', C.c_void_p),
      ('item_blank_mask', C.c_void_p), ('item_mean_coverage', C.c_void_p),
  ]


_lib = None


_ref_lib = None


def _load(path):
  l = C.CDLL(path)
  l.dvo_last_error.restype = C.c_char_p
  l.dvo_channel_str_to_enum.argtypes = [C.c_char_p]
  return l


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = _load(_LIB_PATH)
  return _lib


def reference_available() -> bool:
  return build_reference() is not None


def is_reference_backend() -> bool:
  return hasattr(lib(), 'dvo_is_reference')


@contextlib.contextmanager
def reference_backend():
  """Inside the block every function of this module (encode_read, build_pileup, encode_packed, ...) runs the
  REFERENCE's own encoder (oracle/_ref/libdvref.so) instead of the restatement.  `row_read` results are -2 for
  read rows there (the reference does not say which read a row shows)."""
  global _lib, _ref_lib
  if _ref_lib is None:
    path = build_reference()
    if path is None:
      raise OracleError('oracle/_ref/libdvref.so is not built and %s is not here to build it from' % _REFERENCE_ROOT)
    _ref_lib = _load(path)
  lib()
  saved, _lib = _lib, _ref_lib
  try:
    yield
  finally:
    _lib = saved


class OracleError(RuntimeError):
  pass


def _err():
  return OracleError(lib().dvo_last_error().decode())


def channel_str_to_enum(name: str) -> int:
  v = lib().dvo_channel_str_to_enum(name.encode())
  if v < 0:
    raise OracleError("Channel '%s' should have a corresponding enum" % name)
  return v


def make_options(pic_options) -> DvoOptions:
  """PileupImageOptions -> dvo_options (AllChannelsEnum(""))."""
  o = DvoOptions()
  chans = [channel_str_to_enum(c) for c in pic_options.channels]
  chans = [c for c in chans if c != 0]
  if len(chans) > DVO_MAX_CHANNELS:
    raise OracleError('too many channels')
  o.n_channels = len(chans)
  for i, c in enumerate(chans):
    o.channels[i] = c
  for f in ('width', 'height', 'reference_band_height',
            'base_color_offset_a_and_g', 'base_color_offset_t_and_c',
            'base_color_stride', 'allele_supporting_read_alpha',
            'allele_unsupporting_read_alpha',
            'other_allele_supporting_read_alpha',
            'reference_matching_read_alpha', 'reference_mismatching_read_alpha',
            'reference_base_quality', 'positive_strand_color',
            'negative_strand_color', 'base_quality_cap', 'mapping_quality_cap',
            'random_seed', 'hp_tag_for_assembly_polishing',
            'min_non_zero_allele_frequency'):
    setattr(o, f, getattr(pic_options, f))
  o.sort_by_haplotypes = int(bool(pic_options.sort_by_haplotypes))
  o.sort_by_alt_allele_support = int(
      bool(getattr(pic_options, 'sort_by_alt_allele_support', False)))
  ch = pic_options.indel_anchoring_base_char
  o.indel_anchoring_base_char = ord(ch[0]) if ch else 0
  o.min_base_quality = pic_options.read_requirements.min_base_quality
  o.min_mapping_quality = pic_options.read_requirements.min_mapping_quality
  return o


class _Keep:
  """Keeps ctypes buffers alive for the duration of a call."""

  def __init__(self):
    self.refs = []

  def __call__(self, x):
    self.refs.append(x)
    return x


def _hp_fields(read):
  info = read.info
  present = 'HP' in info
  if not present:
    return 0, 0, 0, 0
  values = info['HP'].values
  n = len(values)
  if n == 0:
    return 1, 0, 0, 0
  v0 = values[0]
  is_int = v0.WhichOneof('kind') == 'int_value'
  return 1, n, int(is_int), int(v0.int_value or 0) if is_int else 0


def _fill_read(dst: DvoRead, read, keep: _Keep):
  dst.fragment_name = keep(read.fragment_name.encode())
  dst.read_number = read.read_number
  dst.position = read.alignment.position.position
  dst.mapping_quality = read.alignment.mapping_quality
  dst.reverse_strand = int(bool(read.alignment.position.reverse_strand))
  dst.supplementary = int(bool(getattr(read, 'supplementary_alignment', False)))
  dst.fragment_length = read.fragment_length
  seq = read.aligned_sequence
  seq_b = keep(seq.encode() if isinstance(seq, str) else bytes(seq))
  dst.seq = seq_b
  dst.seq_len = len(seq_b)
  qual = keep(np.ascontiguousarray(
      np.frombuffer(bytes(bytearray(read.aligned_quality)), dtype=np.uint8)
      if not isinstance(read.aligned_quality, np.ndarray)
      else read.aligned_quality.astype(np.uint8)))
  dst.qual = qual.ctypes.data_as(C.POINTER(C.c_uint8))
  dst.qual_len = len(qual)
  cig = read.alignment.cigar
  ops = keep(np.array([int(c.operation) for c in cig], dtype=np.int32))
  lens = keep(np.array([int(c.operation_length) for c in cig], dtype=np.int64))
  dst.cigar_ops = ops.ctypes.data_as(C.POINTER(C.c_int32))
  dst.cigar_lens = lens.ctypes.data_as(C.POINTER(C.c_int64))
  dst.n_cigar = len(ops)
  (dst.hp_present, dst.hp_n_values, dst.hp_is_int,
   dst.hp_value) = _hp_fields(read)
  mods = getattr(read, 'base_modifications', None) or {}
  for key, pfield, lfield in (('5mC', 'mod_5mc', 'mod_5mc_len'),
                              ('6mA', 'mod_6ma', 'mod_6ma_len')):
    if key in mods:
      arr = keep(np.frombuffer(bytes(mods[key]), dtype=np.uint8).copy())
      setattr(dst, pfield, arr.ctypes.data_as(C.POINTER(C.c_uint8)))
      setattr(dst, lfield, len(arr))
    else:
      setattr(dst, pfield, None)
      setattr(dst, lfield, -1)


  # Ultima's per-base aux tags (read.info["tp"]: int values; read.info["t0"]: one string)
  info = getattr(read, 'info', None) or {}
  dst.tp, dst.tp_len, dst.t0, dst.t0_len = None, 0, None, 0
  if 'tp' in info:
    arr = keep(np.array([int(v.int_value or 0) for v in info['tp'].values] or [0], np.int8))
    dst.tp = arr.ctypes.data_as(C.POINTER(C.c_int8))
    dst.tp_len = len(info['tp'].values)
  if 't0' in info and info['t0'].values:
    raw = keep((info['t0'].values[0].string_value or '').encode('latin-1'))
    dst.t0 = raw
    dst.t0_len = len(raw)


def _str_array(strings: Sequence[str], keep: _Keep):
  arr = (C.c_char_p * max(len(strings), 1))()
  for i, s in enumerate(strings):
    arr[i] = keep(s.encode())
  return keep(arr)


def _make_call(dv_call, keep: _Keep) -> DvoCall:
  c = DvoCall()
  c.variant_start = dv_call.variant.start
  alts = list(dv_call.variant.alternate_bases)
  c.n_alts = len(alts)
  c.alts = _str_array(alts, keep)
  alleles = list(dv_call.allele_support.keys())
  names: List[str] = []
  offs = [0]
  for a in alleles:
    names.extend(dv_call.allele_support[a].read_names)
    offs.append(len(names))
  c.n_support = len(alleles)
  c.support_alleles = _str_array(alleles, keep)
  offs_arr = keep(np.array(offs, dtype=np.int32))
  c.support_offsets = offs_arr.ctypes.data_as(C.POINTER(C.c_int32))
  c.support_names = _str_array(names, keep)
  af = getattr(dv_call, 'allele_frequency', None) or {}
  af_keys = list(af.keys())
  c.n_af = len(af_keys)
  c.af_alleles = _str_array(af_keys, keep)
  af_vals = keep(np.array([af[k] for k in af_keys] or [0], dtype=np.float32))
  c.af_values = af_vals.ctypes.data_as(C.POINTER(C.c_float))
  ref_support = list(getattr(dv_call, 'ref_support', []) or [])
  c.n_ref_support = len(ref_support)
  # read_supports_variant_fuzzy inputs
  c.reference_bases = keep((dv_call.variant.reference_bases or '').encode())
  c.ref_support_names = _str_array(ref_support, keep)
  info = getattr(dv_call.variant, 'info', None) or {}
  if 'ALT_PS' in info:
    vals = [int(v.int_value) for v in info['ALT_PS'].values]
    c.alt_ps_present = 1
    c.n_alt_ps = len(vals)
    ps = keep(np.array(vals or [0], dtype=np.int32))
    c.alt_ps = ps.ctypes.data_as(C.POINTER(C.c_int32))
  rejected = list(getattr(dv_call.variant, 'alternate_bases_rejected', []) or [])
  c.n_rejected_alts = len(rejected)
  c.rejected_alts = _str_array(rejected, keep)
  rsup = getattr(dv_call, 'rejected_allele_support', None) or {}
  ralleles = list(rsup.keys())
  rnames: List[str] = []
  roffs = [0]
  for a in ralleles:
    rnames.extend(rsup[a].read_names)
    roffs.append(len(rnames))
  c.n_rejected_support = len(ralleles)
  c.rejected_support_alleles = _str_array(ralleles, keep)
  roffs_arr = keep(np.array(roffs, dtype=np.int32))
  c.rejected_support_offsets = roffs_arr.ctypes.data_as(C.POINTER(C.c_int32))
  c.rejected_support_names = _str_array(rnames, keep)
  return c


def _blank_array(channels_to_blank, keep: _Keep):
  arr = keep(np.array(list(channels_to_blank or []) or [0], dtype=np.int32))
  return arr.ctypes.data_as(C.POINTER(C.c_int32)), len(channels_to_blank or [])


def encode_reference(pic_options, ref_bases: str) -> np.ndarray:
  """-> uint8 [1, W, C] like the pybind `encode_reference`."""
  o = make_options(pic_options)
  w = len(ref_bases)
  out = np.zeros((1, w, o.n_channels), dtype=np.uint8)
  rc = lib().dvo_encode_reference(C.byref(o), ref_bases.encode(), w,
                                  out.ctypes.data_as(C.c_void_p))
  if rc != 0:
    raise _err()
  return out


def encode_read(pic_options, dv_call, ref_bases: str, read, image_start_pos,
                alt_alleles, channels_to_blank=None) -> Optional[np.ndarray]:
  """-> uint8 [1, W, C] or None (rejected read)."""
  keep = _Keep()
  o = make_options(pic_options)
  w = len(ref_bases)
  out = np.zeros((1, w, o.n_channels), dtype=np.uint8)
  r = DvoRead()
  _fill_read(r, read, keep)
  call = _make_call(dv_call, keep)
  alts = _str_array(list(alt_alleles), keep)
  blank, n_blank = _blank_array(channels_to_blank, keep)
  rc = lib().dvo_encode_read(C.byref(o), C.byref(call), ref_bases.encode(), w,
                             C.byref(r), C.c_int32(image_start_pos), alts,
                             len(alt_alleles), blank, n_blank,
                             out.ctypes.data_as(C.c_void_p))
  if rc < 0:
    raise _err()
  return out if rc == 1 else None


def build_pileup(pic_options, dv_call, ref_bases: str, reads, image_start_pos,
                 alt_alleles, pileup_height=0, mean_coverage=0.0,
                 alignment_positions=None, channels_to_blank=None,
                 return_row_reads=False, non_uniform_downsampling_threshold=None):
  """BuildPileupForOneSample + FillPileupArray -> uint8 [H, W, C].  `non_uniform_downsampling_threshold`: the
  sample's use_non_uniform_downsampling with that many reads kept per allele (pileup_image_native.cc:326-341)."""
  keep = _Keep()
  o = make_options(pic_options)
  if non_uniform_downsampling_threshold is not None:
    o.use_non_uniform_downsampling = 1
    o.non_uniform_downsampling_threshold = int(non_uniform_downsampling_threshold)
  w = len(ref_bases)
  h = pileup_height or pic_options.height
  out = np.zeros((h, w, o.n_channels), dtype=np.uint8)
  row_read = np.full((h,), -1, dtype=np.int32)
  arr = (DvoRead * max(len(reads), 1))()
  for i, rd in enumerate(reads):
    _fill_read(arr[i], rd, keep)
  call = _make_call(dv_call, keep)
  alts = _str_array(list(alt_alleles), keep)
  blank, n_blank = _blank_array(channels_to_blank, keep)
  ap = None
  if alignment_positions is not None and len(alignment_positions):
    ap_arr = keep(np.array(alignment_positions, dtype=np.int64))
    ap = ap_arr.ctypes.data_as(C.POINTER(C.c_int64))
  rc = lib().dvo_build_pileup(
      C.byref(o), C.byref(call), ref_bases.encode(), w, arr, len(reads),
      C.c_int32(image_start_pos), alts, len(alt_alleles), pileup_height,
      C.c_float(mean_coverage), ap, blank, n_blank,
      out.ctypes.data_as(C.c_void_p), row_read.ctypes.data_as(C.c_void_p))
  if rc < 0:
    raise _err()
  if return_row_reads:
    return out, rc, row_read
  return out


def fuzzy_read_supports_alt(dv_call, read, alt_alleles) -> int:
  """ReadSupportsVariantFuzzyChannel::ReadSupportsAlt -> 0 / 1 / 2 / 10 / 9."""
  keep = _Keep()
  c = _make_call(dv_call, keep)
  r = DvoRead()
  _fill_read(r, read, keep)
  alts = _str_array(list(alt_alleles), keep)
  L = lib()
  L.dvo_fuzzy_read_supports_alt.restype = C.c_int
  rc = L.dvo_fuzzy_read_supports_alt(C.byref(c), C.byref(r), alts, len(alt_alleles))
  if rc < 0:
    raise _err()
  return rc


def downsample_indices(n: int, max_reads: int, seed: int) -> np.ndarray:
  out = np.zeros((max(n, 1),), dtype=np.int32)
  lib().dvo_downsample_indices(n, max_reads, C.c_uint32(seed),
                               out.ctypes.data_as(C.c_void_p))
  return out[:n]


def read_overlaps(read, start: int, end: int) -> bool:
  keep = _Keep()
  r = DvoRead()
  _fill_read(r, read, keep)
  return bool(lib().dvo_read_overlaps(C.byref(r), C.c_int64(start),
                                      C.c_int64(end)))


def encode_packed(pic_options, batch, out_channels=None, n_threads=1):
  """Runs the oracle over a packed batch (deepvariant_amd.packing.PackedBatch).

  Returns (images uint8 [total_bytes], rows int32 [n_items]).
  """
  o = make_options(pic_options)
  c_total = out_channels or o.n_channels
  b = DvoPackedBatch()
  keep = []

  def ptr(a, dtype):
    if a is None:
      return None
    a = np.ascontiguousarray(a, dtype=dtype)
    keep.append(a)
    return a.ctypes.data

  b.n_reads = batch.n_reads
  b.read_pos = ptr(batch.read_pos, np.int32)
  b.read_sort_pos = ptr(batch.read_sort_pos, np.int32)
  b.read_seq_off = ptr(batch.read_seq_off, np.uint32)
  b.read_cigar_off = ptr(batch.read_cigar_off, np.uint32)
  b.read_mapq = ptr(batch.read_mapq, np.uint8)
  b.read_flags = ptr(batch.read_flags, np.uint8)
  b.read_frag_len = ptr(batch.read_frag_len, np.int32)
  b.read_hp = ptr(batch.read_hp, np.int32)
  b.read_name_rank = ptr(batch.read_name_rank, np.uint32)
  b.bases = ptr(batch.bases, np.uint8)
  b.quals = ptr(batch.quals, np.uint8)
  b.mod_5mc = ptr(batch.mod_5mc, np.uint8)
  b.mod_6ma = ptr(batch.mod_6ma, np.uint8)
  b.cigar = ptr(batch.cigar, np.uint32)
  b.n_items = batch.n_items
  b.item_variant_start = ptr(batch.item_variant_start, np.int32)
  b.item_image_start = ptr(batch.item_image_start, np.int32)
  b.item_ref_idx = ptr(batch.item_ref_idx, np.uint32)
  b.item_list_off = ptr(batch.item_list_off, np.uint32)
  b.item_height = ptr(batch.item_height, np.uint16)
  b.item_out_off = ptr(batch.item_out_off, np.uint64)
  b.ref_windows = ptr(batch.ref_windows, np.uint8)
  b.list_read = ptr(batch.list_read, np.uint32)
  b.list_code = ptr(batch.list_code, np.uint8)
  b.list_group = ptr(batch.list_group, np.uint8)
  b.item_blank_mask = ptr(batch.item_blank_mask, np.uint32)
  b.item_mean_coverage = ptr(batch.item_mean_coverage, np.float32)
  total = batch.out_bytes(c_total)
  out = np.zeros((total,), dtype=np.uint8)
  rows = np.zeros((max(batch.n_items, 1),), dtype=np.int32)
  rc = lib().dvo_encode_packed(C.byref(o), C.byref(b), c_total,
                               out.ctypes.data_as(C.c_void_p),
                               rows.ctypes.data_as(C.c_void_p), n_threads)
  if rc != 0:
    raise _err()
  return out, rows[:batch.n_items]


# ---------------------------------------------------------------------------------------------------------------
# Candidate generation on the REFERENCE's own code (oracle/_ref/libdvref.so: deepvariant/allelecounter.cc +
# deepvariant/variant_calling_multisample.cc compiled unmodified, one sample): the checker of
# oracle/allelecounter_ref.py and deepvariant_amd/variant_calling.py.  tests/ only.
# ---------------------------------------------------------------------------------------------------------------
class DvrCallingOptions(C.Structure):
  _fields_ = [
      ('partition_size', C.c_int32), ('min_mapping_quality', C.c_int32), ('min_base_quality', C.c_int32),
      ('track_ref_reads', C.c_int32), ('normalize_reads', C.c_int32), ('keep_legacy_behavior', C.c_int32),
      ('call_variants', C.c_int32), ('min_count_snps', C.c_int32), ('min_count_indels', C.c_int32),
      ('min_fraction_snps', C.c_float), ('min_fraction_indels', C.c_float), ('min_fraction_multiplier', C.c_float),
      ('p_error', C.c_float), ('max_gq', C.c_int32), ('gq_resolution', C.c_int32), ('ploidy', C.c_int32),
      ('call_positions_only', C.c_int32),
  ]


def reference_count_and_call(ref_reader, contig: str, start: int, end: int, reads, sample: str = 'sample',
                             candidate_positions=(), full_range=None, min_mapping_quality=0, min_base_quality=0,
                             track_ref_reads=False, normalize_reads=False, keep_legacy_behavior=False,
                             caller: Optional[dict] = None, call_positions_only=False, contig_length: int = 1 << 40,
                             ref_margin: int = 2000):
  """The reference's AlleleCounter over [start, end) of `contig`, fed with `reads` in order, and -- with `caller`
  = dict(min_count_snps=, min_count_indels=, min_fraction_snps=, min_fraction_indels=, ...) -- its multi-sample
  VariantCaller with this one sample as the target.

  -> (counts, calls, positions): counts = {position: (ref_base, ref_supporting_read_count, {read key: (bases,
  type, is_low_quality)})} for the positions that hold anything; calls = list of dicts (start, end,
  reference_bases, alternate_bases, allele_support, ref_support, allele_support_ext, ref_support_ext, info,
  call_set_name, genotype); positions = CallPositionsFromAlleleCounts when asked for."""
  if not reference_available():
    raise OracleError('oracle/_ref/libdvref.so is not available')
  global _ref_lib
  if _ref_lib is None:
    _ref_lib = _load(_REF_LIB_PATH)
  L = _ref_lib
  keep = _Keep()
  arr = (DvoRead * max(len(reads), 1))()
  for i, rd in enumerate(reads):
    _fill_read(arr[i], rd, keep)
  lo = max(0, min(start, full_range[0] if full_range else start) - ref_margin)
  hi = min(contig_length, max(end, full_range[1] if full_range else end) + ref_margin)
  bases = ref_reader.get_bases(contig, lo, hi).encode()
  o = DvrCallingOptions()
  o.partition_size = max(1, end - start)
  o.min_mapping_quality, o.min_base_quality = int(min_mapping_quality), int(min_base_quality)
  o.track_ref_reads, o.normalize_reads = int(bool(track_ref_reads)), int(bool(normalize_reads))
  o.keep_legacy_behavior = int(bool(keep_legacy_behavior))
  o.call_variants = int(caller is not None)
  c = dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06,
           min_fraction_multiplier=1.0, p_error=0.001, max_gq=50, gq_resolution=1, ploidy=2)
  c.update(caller or {})
  for k, v in c.items():
    setattr(o, k, v)
  o.call_positions_only = int(bool(call_positions_only))
  cand = keep(np.array(list(candidate_positions) or [0], dtype=np.int32))
  out, n = C.c_char_p(), C.c_uint64()
  L.dvr_count_and_call.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
  fs, fe = full_range if full_range else (0, 0)
  out_p = C.c_void_p()
  rc = L.dvr_count_and_call(contig.encode(), contig_length, lo, bases, len(bases), start, end, fs, fe, arr, len(reads),
                            sample.encode(), cand.ctypes.data, len(candidate_positions), C.byref(o), C.byref(out_p),
                            C.byref(n))
  if rc != 0:
    raise OracleError(L.dvo_last_error().decode())
  text = C.string_at(out_p.value, n.value).decode()
  L.dvr_free.argtypes = [C.c_void_p]
  L.dvr_free(out_p)
  counts, calls, positions = {}, [], []
  cur = None
  for line in text.split('\n'):
    if not line:
      continue
    f = line.split('\t')
    if f[0] == 'C':
      cur = {}
      counts[int(f[1])] = (f[2], int(f[3]), cur)
    elif f[0] == 'A':
      cur[f[1]] = (f[2], int(f[3]), bool(int(f[4])))
    elif f[0] == 'V':
      calls.append(dict(start=int(f[1]), end=int(f[2]), reference_bases=f[3], alternate_bases=f[4].split(',') if f[4] else [],
                        allele_support={}, ref_support=[], allele_support_ext={}, ref_support_ext=[], info={},
                        call_set_name=None, genotype=None))
    elif f[0] == 'S':
      calls[-1]['allele_support'][f[1]] = f[2].split(',') if f[2] else []
    elif f[0] == 'R':
      calls[-1]['ref_support'] = f[1].split(',') if f[1] else []
    elif f[0] == 'E':
      calls[-1]['allele_support_ext'][f[1]] = [(x.rsplit(':', 1)[0], bool(int(x.rsplit(':', 1)[1]))) for x in f[2].split(',') if x]
    elif f[0] == 'F':
      calls[-1]['ref_support_ext'] = [(x.rsplit(':', 1)[0], bool(int(x.rsplit(':', 1)[1]))) for x in f[1].split(',') if x]
    elif f[0] == 'G':
      calls[-1]['call_set_name'] = f[1]
      calls[-1]['genotype'] = [int(x) for x in f[2].split(',')] if f[2] else []
    elif f[0] == 'I':
      calls[-1]['info'][f[1]] = [float(x) if ('.' in x or 'e' in x or 'n' in x) else int(x) for x in f[2].split(',')] if f[2] else []
    elif f[0] == 'P':
      positions.append(int(f[1]))
  return counts, calls, positions


def reference_window_candidates(ref_reader, contig: str, start: int, end: int, reads, min_mapq=20, min_base_quality=20,
                                keep_legacy_behavior=False, min_allele_support=0, enable_strict_insertion_filter=False,
                                linear_model=None, contig_length: int = 1 << 40, ref_margin: int = 2000):
  """deepvariant/realigner/window_selector.cc on the reference's own AlleleCounter over `reads`:
  -> (reads supporting a variant at every position of [start, end), int32; the linear model's float32 scores, or
  None).  `linear_model` = (bias, coeff_soft_clip, coeff_substitution, coeff_insertion, coeff_deletion,
  coeff_reference, decision_boundary)."""
  if not reference_available():
    raise OracleError('oracle/_ref/libdvref.so is not available')
  global _ref_lib
  if _ref_lib is None:
    _ref_lib = _load(_REF_LIB_PATH)
  L = _ref_lib
  keep = _Keep()
  arr = (DvoRead * max(len(reads), 1))()
  for i, rd in enumerate(reads):
    _fill_read(arr[i], rd, keep)
  lo, hi = max(0, start - ref_margin), min(contig_length, end + ref_margin)
  bases = ref_reader.get_bases(contig, lo, hi).encode()
  counts = np.zeros(end - start, np.int32)
  scores = np.zeros(end - start, np.float32)
  lin = keep(np.array(linear_model, np.float32)) if linear_model is not None else None
  L.dvr_window_candidates.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
  rc = L.dvr_window_candidates(contig.encode(), contig_length, lo, bases, len(bases), start, end, arr, len(reads),
                               int(min_mapq), int(min_base_quality), int(bool(keep_legacy_behavior)), int(min_allele_support),
                               int(bool(enable_strict_insertion_filter)), lin.ctypes.data if lin is not None else None,
                               counts.ctypes.data, scores.ctypes.data)
  if rc < 0:
    raise OracleError(L.dvo_last_error().decode())
  return counts, (scores if linear_model is not None else None)


# ---------------------------------------------------------------------------------------------------------------
# The read realigner and the trimmed-read helpers on the reference's own sources (fast_pass_aligner.cc,
# alt_aligned_pileup_lib.cc in oracle/_ref/libdvref.so; the local aligner under them is the product's libssw
# restatement, see oracle/ref_build/shims/src/ssw_cpp.h).  tests/ only.
# ---------------------------------------------------------------------------------------------------------------
def _ref_text_call(fn_name, argtypes, *args):
  if not reference_available():
    raise OracleError('oracle/_ref/libdvref.so is not available')
  global _ref_lib
  if _ref_lib is None:
    _ref_lib = _load(_REF_LIB_PATH)
  L = _ref_lib
  fn = getattr(L, fn_name)
  fn.argtypes = argtypes + [C.c_void_p, C.c_void_p]
  out_p, n = C.c_void_p(), C.c_uint64()
  rc = fn(*args, C.byref(out_p), C.byref(n))
  if rc != 0:
    raise OracleError(L.dvo_last_error().decode())
  text = C.string_at(out_p.value, n.value).decode('latin-1')
  L.dvr_free.argtypes = [C.c_void_p]
  L.dvr_free(out_p)
  return text


def _parse_read_lines(text):
  """-> list of None (the empty Read) or dict(name, read_number, position, cigar [(op, len)], seq, qual, mapq,
  reverse, mod_5mc, mod_6ma[, original_position])."""
  out = []
  for line in text.split('\n'):
    if not line:
      continue
    f = line.split('\t')
    if f[0] == 'E':
      out.append(None)
    elif f[0] == 'R':
      cigar = [tuple(int(x) for x in u.split(':')) for u in f[4].split(',')] if f[4] else []
      mods = [None if m == '-' else bytes(int(x) for x in m.split(',')) if m else b'' for m in f[9:11]]
      out.append(dict(name=f[1], read_number=int(f[2]), position=int(f[3]), cigar=cigar, seq=f[5],
                      qual=bytes(int(x) for x in f[6].split(',')) if f[6] else b'', mapq=int(f[7]), reverse=bool(int(f[8])),
                      mod_5mc=mods[0], mod_6ma=mods[1]))
    elif f[0] == 'P':
      out[-1]['original_position'] = int(f[1])
  return out


def _aligner_options(cfg):
  keys = ('match', 'mismatch', 'gap_open', 'gap_extend', 'kmer_size', 'read_size', 'max_num_of_mismatches',
          'force_alignment', 'normalize_reads', 'ref_prefix_len', 'ref_suffix_len')
  return (C.c_int32 * len(keys))(*[int(cfg.get(k, 0)) for k in keys]), float(cfg.get('realignment_similarity_threshold', 0.0))


def _read_array(reads, keep):
  arr = (DvoRead * max(len(reads), 1))()
  for i, rd in enumerate(reads):
    _fill_read(arr[i], rd, keep)
  return arr


def reference_align_reads(reference: str, contig: str, ref_start: int, haplotypes, reads, **cfg):
  """FastPassAligner::AlignReads (set_options / set_reference / set_ref_start / set_haplotypes as the window
  realigner and RealignReadsToHaplotype drive it) -> one entry per read (see _parse_read_lines)."""
  keep = _Keep()
  opts, sim = _aligner_options(cfg)
  haps = _str_array(list(haplotypes), keep)
  text = _ref_text_call('dvr_align_reads', [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_double],
                        reference.encode(), contig.encode(), int(ref_start), haps, len(haplotypes), _read_array(reads, keep),
                        len(reads), opts, sim)
  return _parse_read_lines(text)


def reference_trim_reads(reads, contig: str, region_start: int, region_end: int, min_overlap: int = 15):
  """TrimReads (alt_aligned_pileup_lib.cc:231-248) -> kept reads with `original_position`."""
  keep = _Keep()
  text = _ref_text_call('dvr_trim_reads', [C.c_void_p, C.c_int, C.c_char_p, C.c_int64, C.c_int64, C.c_int],
                        _read_array(reads, keep), len(reads), contig.encode(), int(region_start), int(region_end),
                        int(min_overlap))
  return _parse_read_lines(text)


def reference_realign_reads_to_haplotype(haplotype: str, reads, contig: str, ref_start: int, ref_end: int, ref_reader,
                                         contig_length: int, **cfg):
  """RealignReadsToHaplotype (alt_aligned_pileup_lib.cc:278-313)."""
  keep = _Keep()
  opts, sim = _aligner_options(cfg)
  lo, hi = max(0, ref_start - 100), min(contig_length, ref_end + 100)
  bases = ref_reader.get_bases(contig, lo, hi).encode()
  text = _ref_text_call('dvr_realign_reads_to_haplotype',
                        [C.c_char_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_char_p,
                         C.c_int64, C.c_void_p, C.c_double],
                        haplotype.encode(), _read_array(reads, keep), len(reads), contig.encode(), int(ref_start), int(ref_end),
                        int(contig_length), lo, bases, len(bases), opts, sim)
  return _parse_read_lines(text)


def reference_calculate_alignment_region(contig_length: int, variant_start: int, n_reference_bases: int, half_width: int):
  if not reference_available():
    raise OracleError('oracle/_ref/libdvref.so is not available')
  global _ref_lib
  if _ref_lib is None:
    _ref_lib = _load(_REF_LIB_PATH)
  out = (C.c_int64 * 2)()
  _ref_lib.dvr_calculate_alignment_region.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
  if _ref_lib.dvr_calculate_alignment_region(b'contig', int(contig_length), int(variant_start), int(n_reference_bases),
                                             int(half_width), out) != 0:
    raise OracleError(_ref_lib.dvo_last_error().decode())
  return int(out[0]), int(out[1])


# ---------------------------------------------------------------------------------------------------------------
# The top of the hot path on the reference's own code: ExamplesGenerator::WriteExamplesInRegion
# (deepvariant/make_examples_native.cc in oracle/_ref/libdvref.so).  tests/ only.
# ---------------------------------------------------------------------------------------------------------------
def reference_write_examples_in_region(options, ref_reader, contig: str, contig_length: int, candidates,
                                       reads_per_sample, sample_order, role: str, mean_coverage_per_sample,
                                       aln_config: Optional[dict] = None, ref_margin: int = 600):
  """`options`: MakeExamplesOptions-shaped (pic_options, sample_options, trim_reads_for_pileup).
  -> (serialized tf.Examples in the order the reference wrote them, image_shape)."""
  if not reference_available():
    raise OracleError('oracle/_ref/libdvref.so is not available')
  global _ref_lib
  if _ref_lib is None:
    _ref_lib = _load(_REF_LIB_PATH)
  L = _ref_lib
  pic = options.pic_options
  lines = []
  for k in ('width', 'height', 'reference_band_height', 'base_color_offset_a_and_g', 'base_color_offset_t_and_c',
            'base_color_stride', 'allele_supporting_read_alpha', 'allele_unsupporting_read_alpha',
            'other_allele_supporting_read_alpha', 'reference_matching_read_alpha', 'reference_mismatching_read_alpha',
            'indel_anchoring_base_char', 'reference_base_quality', 'positive_strand_color', 'negative_strand_color',
            'base_quality_cap', 'mapping_quality_cap', 'read_overlap_buffer_bp', 'multi_allelic_mode', 'random_seed',
            'num_channels', 'sequencing_type', 'alt_aligned_pileup', 'types_to_alt_align', 'hp_tag_for_assembly_polishing',
            'min_non_zero_allele_frequency'):
    v = getattr(pic, k)
    lines.append('O\tpic.%s\t%s' % (k, repr(float(v)) if isinstance(v, float) else (int(v) if not isinstance(v, str) else v)))
  lines.append('O\tpic.sort_by_haplotypes\t%d' % int(bool(pic.sort_by_haplotypes)))
  lines.append('O\tpic.sort_by_alt_allele_support\t%d' % int(bool(getattr(pic, 'sort_by_alt_allele_support', False))))
  lines.append('O\tpic.min_base_quality\t%d' % pic.read_requirements.min_base_quality)
  lines.append('O\tpic.min_mapping_quality\t%d' % pic.read_requirements.min_mapping_quality)
  lines.append('O\tpic.channels\t%s' % ','.join(pic.channels))
  lines.append('O\ttrim_reads_for_pileup\t%d' % int(bool(options.trim_reads_for_pileup)))
  for k, v in (aln_config or {}).items():
    lines.append('O\taln.%s\t%s' % (k, v))
  for so in options.sample_options:
    lines.append('M\t%s\t%s\t%d\t%s\t%s\t%s\t%d\t%d\t%d' % (
        so.role, so.name, so.pileup_height, ','.join(map(str, so.order)), so.alt_aligned_pileup,
        ','.join(str(int(c)) for c in so.channels_enum_to_blank), int(bool(so.keep_only_window_spanning_reads)),
        int(bool(getattr(so, 'use_non_uniform_downsampling', False))),
        int(getattr(so, 'non_uniform_downsampling_threshold', 0))))
  lo, hi = contig_length, 0
  for c in candidates:
    v = c.variant
    call = v.calls[0] if v.calls else None
    lines.append('C\t%d\t%d\t%s\t%s\t%s\t%s' % (v.start, v.end, v.reference_bases, ','.join(v.alternate_bases),
                                                call.call_set_name if call else '', ','.join(map(str, call.genotype)) if call else ''))
    for allele, support in c.allele_support.items():
      lines.append('S\t%s\t%s' % (allele, ','.join(support.read_names)))
    if getattr(c, 'make_examples_alt_allele_indices', None):
      lines.append('A\t%s' % ';'.join(','.join(map(str, idx.indices if hasattr(idx, 'indices') else idx))
                                      for idx in c.make_examples_alt_allele_indices))
    if call:
      for key, lv in call.info.items():
        vals = lv.values
        if vals and getattr(vals[0], 'number_value', None) not in (None, 0.0) or key == 'VAF':
          lines.append('I\t%s\tfloat\t%s' % (key, ','.join(repr(float(x.number_value)) for x in vals)))
        else:
          lines.append('I\t%s\tint\t%s' % (key, ','.join(str(int(x.int_value)) for x in vals)))
    lo, hi = min(lo, v.start), max(hi, v.end)
  lo, hi = max(0, lo - pic.width - ref_margin), min(contig_length, hi + pic.width + ref_margin)
  bases = ref_reader.get_bases(contig, lo, hi).encode() if hi > lo else b''
  keep = _Keep()
  flat = [r for sample in reads_per_sample for r in sample]
  arr = _read_array(flat, keep)
  n_per = (C.c_int32 * max(len(reads_per_sample), 1))(*[len(s) for s in reads_per_sample])
  order = (C.c_int32 * max(len(sample_order), 1))(*[int(x) for x in sample_order])
  cov = (C.c_float * max(len(reads_per_sample), 1))(*[float(x) for x in mean_coverage_per_sample])
  shape = (C.c_int32 * 3)()
  out_p, n = C.c_void_p(), C.c_uint64()
  L.dvr_write_examples_in_region.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
  rc = L.dvr_write_examples_in_region('\n'.join(lines).encode(), contig.encode(), int(contig_length), lo, bases, len(bases), arr,
                                      n_per, len(reads_per_sample), order, len(sample_order), role.encode(), cov, shape,
                                      C.byref(out_p), C.byref(n))
  if rc != 0:
    raise OracleError(L.dvo_last_error().decode())
  blob = C.string_at(out_p.value, n.value)
  L.dvr_free.argtypes = [C.c_void_p]
  L.dvr_free(out_p)
  examples, at = [], 0
  while at < len(blob):
    ln = int.from_bytes(blob[at:at + 4], 'little')
    examples.append(blob[at + 4:at + 4 + ln])
    at += 4 + ln
  return examples, [int(x) for x in shape]


# ---------------------------------------------------------------------------------------------------------------
# Local assembly and read phasing on the reference's own sources (realigner/debruijn_graph.cc, direct_phasing.cc in
# oracle/_ref/libdvref.so, over the Boost-Graph stand-in of oracle/ref_build/shims/boost/graph/).  tests/ only.
# ---------------------------------------------------------------------------------------------------------------
class ReferenceDeBruijnGraph:
  """What deepvariant_amd.realigner.debruijn_graph.DeBruijnGraph offers, answered by the reference build."""

  def __init__(self, kmer_size, haplotypes, dot):
    self.kmer_size, self._haplotypes, self._dot = kmer_size, haplotypes, dot

  def candidate_haplotypes(self):
    return list(self._haplotypes)

  def graphviz(self):
    return self._dot


def reference_debruijn(ref: str, reads, options):
  """DeBruijnGraph::Build(ref, reads, options) -> ReferenceDeBruijnGraph, or None where the reference returns
  nullptr (no k without a cycle)."""
  keep = _Keep()
  opts = (C.c_int32 * 8)(options.min_k, options.max_k, options.step_k, options.min_mapq, options.min_base_quality,
                         options.min_edge_weight, options.max_num_paths, int(bool(options.disable_graph_pruning)))
  text = _ref_text_call('dvr_debruijn', [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p], ref.encode(), _read_array(reads, keep),
                        len(reads), opts)
  if text.startswith('N'):
    return None
  head, _, dot = text.partition('#GRAPHVIZ\n')
  k, haplotypes = 0, []
  for line in head.split('\n'):
    f = line.split('\t')
    if f[0] == 'K':
      k = int(f[1])
    elif f[0] == 'H':
      haplotypes.append(f[1])
  return ReferenceDeBruijnGraph(k, haplotypes, dot)


class ReferenceDirectPhasing:
  """deepvariant_amd.direct_phasing.DirectPhasing's interface on the reference's DirectPhasing."""

  def __init__(self, min_alleles_to_phase: int = 1):
    self._min = int(min_alleles_to_phase)
    self._variants, self._dot = [], ''

  def phase(self, candidates, reads):
    lines = []
    for c in candidates:
      v = c.variant
      lines.append('C\t%d\t%d\t%s\t%s' % (v.start, v.end, v.reference_bases, ','.join(v.alternate_bases)))
      for allele, infos in c.allele_support_ext.items():
        lines.append('E\t%s\t%s' % (allele, ','.join('%s:%d' % (i.read_name, int(bool(i.is_low_quality))) for i in infos)))
      if c.ref_support_ext:
        lines.append('F\t%s' % ','.join('%s:%d' % (i.read_name, int(bool(i.is_low_quality))) for i in c.ref_support_ext))
    keep = _Keep()
    phases = np.zeros(max(len(reads), 1), np.int32)
    text = _ref_text_call('dvr_phase_reads', [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p], '\n'.join(lines).encode(),
                          _read_array(reads, keep), len(reads), self._min, phases.ctypes.data)
    head, _, self._dot = text.partition('#GRAPHVIZ\n')
    self._variants = []
    for line in head.split('\n'):
      f = line.split('\t')
      if f[0] == 'V':
        self._variants.append((int(f[1]), f[2], f[3], bool(int(f[4]))))
    return [int(p) for p in phases[:len(reads)]]

  def get_phased_variants(self):
    from deepvariant_amd import direct_phasing
    return [direct_phasing.PhasedVariant(*v) for v in self._variants]

  def graphviz(self):
    return self._dot


def reference_normalize_cigars(ref_reader, contig: str, start: int, end: int, reads, contig_length: int = 1 << 40,
                               ref_margin: int = 2000):
  """AlleleCounter::NormalizeAndAdd per read (a counter over [start, end) of `contig`, normalize_reads on)
  -> [(is_modified, read_shift, [(op, length), ...])]."""
  keep = _Keep()
  lo, hi = max(0, start - ref_margin), min(contig_length, end + ref_margin)
  bases = ref_reader.get_bases(contig, lo, hi).encode()
  text = _ref_text_call('dvr_normalize_cigars', [C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_int64, C.c_int64,
                                                 C.c_void_p, C.c_int],
                        contig.encode(), int(contig_length), lo, bases, len(bases), int(start), int(end), _read_array(reads, keep),
                        len(reads))
  out = []
  for line in text.split('\n'):
    f = line.split('\t')
    if f[0] == 'N':
      out.append((bool(int(f[1])), int(f[2]), [tuple(int(x) for x in u.split(':')) for u in f[3].split(',')] if f[3] else []))
  return out
