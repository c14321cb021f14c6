"""Host side of the encoder boundary: proto-shaped inputs -> packed `dv_batch`.

This is where the strings of the reference's inputs stop.  The reference's
encoder does string work per read per candidate
(`ReadSupportsVariantChannel::ReadSupportsAlt`,
deepvariant/channels/read_supports_variant_channel.cc:75-104; the
(fragment_name, read_number) tie-break of `SortImageRows`,
deepvariant/pileup_image_native.cc:97-101).  Here each is resolved ONCE on the
host into a small integer per read / per (item, read); only integers and bytes
cross to the device (include/dvhip.h, `dv_batch`).
"""
from __future__ import annotations

import ctypes as C
import os

import dataclasses
import functools
from typing import Dict, List, Optional, Sequence

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T

DV_READ_REVERSE, DV_READ_SUPPLEMENTARY, DV_READ_HAS_5MC, DV_READ_HAS_6MA = 1, 2, 4, 8

# Channels whose pixel the device computes itself / which need host-computed
# aux bytes (include/dvhip.h DV_CH_*).
_READ_AUX_SLOT = {11: 0, 12: 1, 13: 2, 14: 3, 15: 4}
_SEQ_AUX_CHANNELS = (16, 17, 28, 29, 30)  # is_homopolymer, homopolymer_weighted, the three Ultima flow-space channels: per-base pixels
_FLOW_CHANNELS = (28, 29, 30)       # homopolymer_{insertion,deletion}_quality (tp tag), inter_homopolymer_insertion_quality (t0)
_REF_AUX_CHANNELS = (15, 16, 17)    # ... and gc_content: per-window reference-row pixels
_LIST_AUX_CHANNELS = (8, 25, 27)   # one pixel per (item, read), computed on the host


def channel_enums(pic_options) -> List[int]:
  """AllChannelsEnum("") -- pileup_image_native.cc:125-151."""
  out = []
  for name in pic_options.channels:
    if name not in T.CHANNEL_STR_TO_ENUM:
      raise ValueError(
          "Channel '%s' should have a corresponding enum in "
          'DeepVariantChannelEnum.' % name)
    e = T.CHANNEL_STR_TO_ENUM[name]
    if e != 0:
      out.append(e)
  return out


def make_encoder_options(pic_options, width: Optional[int] = None
                         ) -> _lib.DvEncoderOptions:
  o = _lib.DvEncoderOptions()
  chans = channel_enums(pic_options)
  if len(chans) > _lib.DV_MAX_CHANNELS:
    raise ValueError('at most %d channels' % _lib.DV_MAX_CHANNELS)
  o.n_channels = len(chans)
  for i, c in enumerate(chans):
    o.channels[i] = c
  for f in ('height', 'reference_band_height', 'base_color_offset_a_and_g',
            'base_color_offset_t_and_c', 'base_color_stride',
            'allele_supporting_read_alpha', 'allele_unsupporting_read_alpha',
            'other_allele_supporting_read_alpha',
            'reference_matching_read_alpha', 'reference_mismatching_read_alpha',
            'reference_base_quality', 'positive_strand_color',
            'negative_strand_color', 'base_quality_cap', 'mapping_quality_cap',
            'random_seed', 'hp_tag_for_assembly_polishing',
            'min_non_zero_allele_frequency'):
    setattr(o, f, getattr(pic_options, f))
  o.width = width if width is not None else pic_options.width
  o.sort_by_haplotypes = int(bool(pic_options.sort_by_haplotypes))
  o.sort_by_alt_allele_support = int(
      bool(getattr(pic_options, 'sort_by_alt_allele_support', False)))
  ch = pic_options.indel_anchoring_base_char
  o.indel_anchoring_base_char = ord(ch[0]) if ch else 0
  o.min_base_quality = pic_options.read_requirements.min_base_quality
  o.min_mapping_quality = pic_options.read_requirements.min_mapping_quality
  return o


def _scale_color(value: int, max_val: float) -> int:
  """ScaleColor of channels/*.cc in IEEE fp32."""
  mv = np.float32(max_val)
  if np.float32(value) > mv:
    value = int(mv)
  return int(np.float32(254.0) * (np.float32(value) / mv)) & 0xFF


def is_homopolymer_pixels(seq: bytes) -> np.ndarray:
  """IsHomopolymerChannel::IsHomopolymer + ScaleColorVector(.., 1)
  (channels/is_homopolymer_channel.cc:83-113): runs of >= 3 equal bases -> 254."""
  b = np.frombuffer(seq, np.uint8)
  out = np.zeros(b.size, np.uint8)
  if b.size >= 3:
    trip = (b[2:] == b[1:-1]) & (b[1:-1] == b[:-2])
    out[2:][trip] = 254
    out[1:-1][trip] = 254
    out[:-2][trip] = 254
  return out


_HW_LUT = None


def homopolymer_weighted_pixels(seq: bytes) -> np.ndarray:
  """HomopolymerWeightedChannel::HomopolymerWeighted + ScaleColorVector(.., 30)
  (channels/homopolymer_weighted_channel.cc:85-122).  The reference keeps run lengths
  in a vector<uint8_t>, so runs wrap modulo 256 before the cap at 30."""
  global _HW_LUT
  if _HW_LUT is None:
    _HW_LUT = np.array([_scale_color(v, 30.0) for v in range(256)], np.uint8)
  b = np.frombuffer(seq, np.uint8)
  if b.size == 0:
    return np.zeros(0, np.uint8)
  change = np.flatnonzero(b[1:] != b[:-1]) + 1
  starts = np.concatenate([[0], change])
  lens = np.diff(np.concatenate([starts, [b.size]]))
  return _HW_LUT[np.repeat(lens, lens) & 0xFF]


def seq_aux_planes(enums: Sequence[int]) -> tuple:
  """(channel of base_aux0, of base_aux1, of base_aux2), 0 = plane unused, () if the channel list has no per-base
  host-computed channel: include/dvhip.h's rule, asked of the library (dv_base_aux_plane) so that the host fills the
  plane the device reads."""
  enums = [int(e) for e in enums]
  if not any(e in _SEQ_AUX_CHANNELS for e in enums):
    return ()
  arr = (C.c_int32 * len(enums))(*enums)
  planes = [0, 0, 0]
  for i, e in enumerate(enums):
    if e in _SEQ_AUX_CHANNELS:
      plane = int(_lib.lib().dv_base_aux_plane(arr, len(enums), i))
      if plane < 0:
        _lib.check(plane)
      planes[plane] = e
  return tuple(planes)


def _flow_tags(channel: int, reads: Sequence, seq_off: np.ndarray) -> np.ndarray:
  """The `tags` plane of dv_flow_channel_pixels: tp values (int8, 0 where a read has no tag or a shorter one) /
  t0 characters - 33 (homopolymer_indel_quality_channel.cc:68-84, inter_homopolymer_insertion_quality_channel.cc:
  76-112)."""
  tags = np.zeros(int(seq_off[-1]), np.int8)
  for i, r in enumerate(reads):
    info = getattr(r, 'info', None) or {}
    s0, n = int(seq_off[i]), int(seq_off[i + 1]) - int(seq_off[i])
    if channel in (28, 29):
      if 'tp' in info and info['tp'].values:
        vals = np.array([int(v.int_value or 0) for v in info['tp'].values[:n]], np.int64)
        tags[s0:s0 + len(vals)] = vals.astype(np.int8)
    elif 't0' in info and info['t0'].values:
      raw = np.frombuffer((info['t0'].values[0].string_value or '').encode('latin-1')[:n], np.uint8)
      tags[s0:s0 + len(raw)] = (raw.astype(np.int16) - 33).astype(np.uint8).view(np.int8)
  return tags


def seq_aux_plane(channel: int, reads: Sequence, seqs: Sequence[bytes], bases: np.ndarray, quals: np.ndarray,
                  seq_off: np.ndarray) -> np.ndarray:
  """The per-base pixels of one host-computed channel for the reads of a table."""
  if channel == 16:
    return np.concatenate([is_homopolymer_pixels(x) for x in seqs] or [np.zeros(0, np.uint8)])
  if channel == 17:
    return np.concatenate([homopolymer_weighted_pixels(x) for x in seqs] or [np.zeros(0, np.uint8)])
  if channel not in _FLOW_CHANNELS:
    raise ValueError('channel %d has no per-base plane' % channel)
  tags = _flow_tags(channel, reads, seq_off)
  out = np.zeros(int(seq_off[-1]), np.uint8)
  bases = np.ascontiguousarray(bases, np.uint8)
  quals = np.ascontiguousarray(quals, np.uint8)
  off = np.ascontiguousarray(seq_off, np.uint32)
  _lib.check(_lib.lib().dv_flow_channel_pixels(
      int(channel), bases.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p),
      tags.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(off) - 1,
      out.ctypes.data_as(C.c_void_p)))
  return out


def non_uniform_sample(dv_call, table, idx: np.ndarray, max_reads: int, min_per_allele: int, random_seed: int,
                       forced_draws=None) -> Optional[np.ndarray]:
  """SampleOptions.use_non_uniform_downsampling for one pile-up whose reads are the table rows `idx`
  (DownsampleReadIndicesWithMinsPerAllele, pileup_image_native.cc:242-294): -> positions in `idx` (ascending) of the
  reads that stay, or None where the reference falls back to the uniform shuffle (then the whole list goes to the
  encoder as usual).  Alleles are taken in key order (the reference walks a hash map: for a read listed under two
  alleles its own result depends on hash order).  The sampler is native (dv_downsample_with_partition_mins)."""
  n = len(idx)
  position = {}
  keys = table.keys
  for k, row in enumerate(np.asarray(idx).tolist()):
    position[keys[row]] = k          # a later read replaces an earlier one of the same key (:251-256)
  off, flat = [0], []
  for allele in sorted(dv_call.allele_support):
    for name in dv_call.allele_support[allele].read_names:
      k = position.get(name)
      if k is not None:
        flat.append(k)
    off.append(len(flat))
  listed = set(flat)
  flat += sorted(k for k in position.values() if k not in listed)      # the reads no allele lists: the last element
  off.append(len(flat))
  part_off = np.array(off, np.int32)
  part_idx = np.array(flat or [0], np.int32)
  out = np.zeros(max(n, 1), np.int32)
  n_out = C.c_int32(0)
  forced = None if forced_draws is None else np.ascontiguousarray(forced_draws, np.uint64)
  _lib.check(_lib.lib().dv_downsample_with_partition_mins(
      n, part_off.ctypes.data_as(C.c_void_p), part_idx.ctypes.data_as(C.c_void_p), len(off) - 1, int(max_reads),
      int(min_per_allele), int(random_seed) & 0xFFFFFFFF,
      None if forced is None else forced.ctypes.data_as(C.c_void_p), 0 if forced is None else len(forced),
      out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
  if n_out.value < 0:
    return None
  return out[:n_out.value].astype(np.int64)


def gc_content_pixel(seq: bytes) -> int:
  """GcContentChannel::GcContent + ScaleColor(.., 100) (channels/gc_content_channel.cc:78-101)."""
  if not seq:
    return 0
  b = np.frombuffer(seq, np.uint8)
  gc = int(((b == ord('G')) | (b == ord('C'))).sum())
  pct = int(np.float32(gc) / np.float32(len(seq)) * np.float32(100))
  return _scale_color(pct, 100.0)


def read_key(read) -> str:
  """read_supports_variant_channel.cc:78-79."""
  return '%s/%d' % (read.fragment_name, read.read_number)


def _hp_value(read) -> int:
  """The single integer HP tag, or DV_HP_NONE.

  HaplotypeTagChannel (haplotype_tag_channel.cc:76-100) and GetHapIndex
  (pileup_image_native.cc:449-475) only disagree on reads carrying several HP
  values or a non-integer one; direct phasing never writes those, and the
  packer refuses them rather than guess.
  """
  info = read.info
  if 'HP' not in info:
    return _lib.DV_HP_NONE
  values = info['HP'].values
  if len(values) == 0:
    return _lib.DV_HP_NONE
  if len(values) > 1 or values[0].WhichOneof('kind') != 'int_value':
    raise ValueError('unsupported HP tag shape on read %s' % read.fragment_name)
  return int(values[0].int_value)


class LazyRead(T.Read):
  """A Read made from a row of a packed table (ReadTable.read_factory) whose `alignment` -- a
  LinearAlignment, a Position and one CigarUnit per operation: most of what building a Read costs
  -- is created on first access.  Most reads of a calling region are never asked for it: window
  selection, allele counting and the encoder's batch take the packed row the read carries
  (`_dv_packed`), spans come from it too (realigner.utils.read_range); only the reads the realigner
  assigns to a window are looked at as objects.  Everything else is an ordinary dv_types.Read."""

  def __getattr__(self, name):
    # reached only when normal lookup fails: `alignment` has a default_factory, so the class has no
    # attribute of that name and an instance made without __init__ lands here
    if name == 'alignment':
      d = self.__dict__
      build = d.pop('_dv_alignment', None)
      if build is not None:
        aln = d['alignment'] = build()
        rec = d.get('_dv_packed')
        if rec is not None and rec[0] is None:
          d['_dv_packed'] = (aln,) + rec[1:]     # from now on the record is tied to this object
        return aln
    raise AttributeError(name)

  def __eq__(self, other):
    if not isinstance(other, T.Read):
      return NotImplemented
    return all(getattr(self, f.name) == getattr(other, f.name) for f in dataclasses.fields(T.Read))

  __hash__ = None


def packed_record(read):
  """The read's cached ReadTable._pack_read record if it is still valid, else None: valid while
  `alignment` is the object the record was made from, or -- a LazyRead -- has not been built."""
  rec = getattr(read, '_dv_packed', None)
  if rec is None:
    return None
  d = getattr(read, '__dict__', None)
  aln = d.get('alignment') if d is not None else read.alignment
  if aln is None:
    return rec if rec[0] is None and '_dv_alignment' in d else None
  return rec if rec[0] is aln else None


@dataclasses.dataclass
class ReadTable:
  """Structure-of-arrays image of a list of Read protos (one region)."""
  n_reads: int
  read_pos: np.ndarray
  read_sort_pos: Optional[np.ndarray]
  read_seq_off: np.ndarray
  read_cigar_off: np.ndarray
  read_mapq: np.ndarray
  read_flags: np.ndarray
  read_frag_len: np.ndarray
  read_hp: np.ndarray
  read_name_rank: np.ndarray
  read_aux: Optional[np.ndarray]
  bases: np.ndarray
  quals: np.ndarray
  mod_5mc: Optional[np.ndarray]
  mod_6ma: Optional[np.ndarray]
  cigar: np.ndarray
  keys: List[str]
  read_end: np.ndarray
  base_aux0: Optional[np.ndarray] = None  # is_homopolymer pixel per base
  base_aux1: Optional[np.ndarray] = None  # homopolymer_weighted pixel per base
  base_aux2: Optional[np.ndarray] = None  # third per-base plane (flow-space channels; include/dvhip.h's rule)
  # the Ultima flow-space tags as the native decoders deliver them (parse_flow_tags): tp values / t0 characters - 33
  # per base, and per read bit 0 = has tp, bit 1 = has t0
  flow_tp: Optional[np.ndarray] = None
  flow_t0: Optional[np.ndarray] = None
  flow_present: Optional[np.ndarray] = None

  @staticmethod
  def _pack_read(r, need_aux: bool) -> tuple:
    """Everything from_reads needs from one Read, in packed form: (alignment object, position,
    mapq, flags, fragment length, bases, qualities, 5mC | None, 6mA | None, CIGAR words, end,
    aux row | None, key, sort key).  Kept on the Read (`_dv_packed`): a region's reads are packed
    several times (window selection, candidate calling, the encoder's batch) and most of them
    are the same objects each time -- the realigner builds NEW Read objects for the reads it
    moves, so a cached record is valid as long as `alignment` is still the object it was made
    from (checked by the caller)."""
    aln = r.alignment
    p = aln.position.position
    if not -(1 << 31) <= p < (1 << 31):
      raise ValueError('alignment position does not fit int32')
    mq = aln.mapping_quality
    if not 0 <= mq <= 255:
      raise ValueError('mapping_quality %d outside [0, 255]' % mq)
    f = 0
    if aln.position.reverse_strand:
      f |= DV_READ_REVERSE
    if getattr(r, 'supplementary_alignment', False):
      f |= DV_READ_SUPPLEMENTARY
    seq = r.aligned_sequence
    sb = seq.encode() if isinstance(seq, str) else bytes(seq)
    qb = bytes(bytearray(r.aligned_quality))
    if len(qb) != len(sb):
      raise ValueError('aligned_quality and aligned_sequence differ in length '
                       'for read %s' % r.fragment_name)
    mods = getattr(r, 'base_modifications', None) or {}
    mod_bytes = [None, None]
    for k, (key, bit) in enumerate(((T.K5MC, DV_READ_HAS_5MC), (T.K6MA, DV_READ_HAS_6MA))):
      if key in mods:
        mb = bytes(mods[key])
        if len(mb) != len(sb):
          raise ValueError('base_modifications length mismatch')
        mod_bytes[k] = mb
        f |= bit
    e = p
    qlen = 0
    match_len = gap_len = 0
    cig = []
    for cu in aln.cigar:
      op, ln = int(cu.operation), int(cu.operation_length)
      if not 1 <= op <= 9:
        raise ValueError('Unrecognized CIGAR op')  # reference: LOG(FATAL)
      if not 0 <= ln < (1 << 28):
        raise ValueError('CIGAR operation_length out of range')
      cig.append((ln << 4) | op)
      if op in (1, 8, 9, 3, 4):
        e += ln
      if op in (1, 2, 5, 8, 9):
        qlen += ln
      if op in (1, 8):
        match_len += ln
        gap_len += ln
      elif op == 9:
        gap_len += ln
      elif op in (2, 3):
        gap_len += 1
    if qlen > len(sb):
      raise ValueError('CIGAR consumes more bases than aligned_sequence has')
    aux_row = None
    if need_aux:
      # read_mapping_percent / identity (identical arithmetic),
      # avg_base_quality, gap_compressed_identity -- channels/*.cc.
      aux_row = [0] * _lib.DV_READ_AUX_STRIDE
      f32 = np.float32
      pct = int(f32(match_len) / f32(max(len(sb), 1)) * f32(100)) if sb else 0
      aux_row[0] = _scale_color(pct, 100)
      aux_row[2] = _scale_color(pct, 100)
      if qb:
        if max(qb) > 93:
          raise ValueError('Encountered base quality outside of bounds (0,93)')
        aux_row[1] = _scale_color(int(f32(sum(qb)) / f32(len(qb))), 93)
      if gap_len:
        aux_row[3] = _scale_color(
            int(f32(match_len) / f32(gap_len) * f32(100)), 100)
      aux_row[4] = gc_content_pixel(sb)
    return (aln, p, mq, f, r.fragment_length, sb, qb, mod_bytes[0], mod_bytes[1], cig, e, aux_row,
            read_key(r), (r.fragment_name.encode(), int(r.read_number)))

  @classmethod
  def from_reads(cls, reads: Sequence, alignment_positions=None,
                 need_aux: bool = False, need_seq_aux=False) -> 'ReadTable':
    """`need_seq_aux`: seq_aux_planes(channel enums) -- the channel each base_aux plane carries -- or True for
    is_homopolymer + homopolymer_weighted in planes 0 / 1."""
    n = len(reads)
    planes = (16, 17, 0) if need_seq_aux is True else tuple(need_seq_aux or ())
    recs = []
    for r in reads:
      rec = packed_record(r)
      if rec is None or (need_aux and rec[11] is None):
        rec = cls._pack_read(r, need_aux)
        try:
          r._dv_packed = rec   # pylint: disable=protected-access
        except AttributeError:   # a real protobuf message: no cache
          pass
      recs.append(rec)
    pos = np.array([x[1] for x in recs], np.int32).reshape(n)
    mapq = np.array([x[2] for x in recs], np.uint8).reshape(n)
    flags = np.array([x[3] for x in recs], np.uint8).reshape(n)
    frag = np.array([x[4] for x in recs], np.int32).reshape(n)
    hp = np.array([_hp_value(r) for r in reads], np.int32).reshape(n)
    seqs = [x[5] for x in recs]
    quals = [x[6] for x in recs]
    seq_off = np.zeros(n + 1, np.uint32)
    np.cumsum([len(x) for x in seqs], out=seq_off[1:])
    cig_off = np.zeros(n + 1, np.uint32)
    np.cumsum([len(x[9]) for x in recs], out=cig_off[1:])
    cig = [w for x in recs for w in x[9]]
    read_end = np.array([x[10] for x in recs], np.int64).reshape(n)
    any5 = any(x[7] is not None for x in recs)
    any6 = any(x[8] is not None for x in recs)
    m5 = [x[7] if x[7] is not None else b'\0' * len(x[5]) for x in recs] if any5 else []
    m6 = [x[8] if x[8] is not None else b'\0' * len(x[5]) for x in recs] if any6 else []
    aux = np.array([x[11] for x in recs], np.uint8).reshape(n, _lib.DV_READ_AUX_STRIDE) if need_aux else None
    keys = [x[12] for x in recs]
    sort_keys = [x[13] for x in recs]
    # dense rank under the reference's tuple<string,int> ordering
    uniq = sorted(set(sort_keys))
    rank_of = {k: j for j, k in enumerate(uniq)}
    ranks = np.array([rank_of[k] for k in sort_keys], np.uint32).reshape(n)
    sort_pos = None
    if alignment_positions is not None and len(alignment_positions):
      if len(alignment_positions) != n:
        raise ValueError('alignment_positions must match reads')
      sort_pos = np.array(alignment_positions, np.int64).astype(np.int32)
    bases_arr = np.frombuffer(b''.join(seqs), np.uint8)
    quals_arr = np.frombuffer(b''.join(quals), np.uint8)
    aux_planes = [seq_aux_plane(ch, reads, seqs, bases_arr, quals_arr, seq_off) if ch else None
                  for ch in (planes + (0, 0, 0))[:3]]
    return cls(
        n_reads=n, read_pos=pos, read_sort_pos=sort_pos, read_seq_off=seq_off,
        read_cigar_off=cig_off, read_mapq=mapq, read_flags=flags,
        read_frag_len=frag, read_hp=hp, read_name_rank=ranks, read_aux=aux,
        bases=bases_arr,
        quals=quals_arr,
        mod_5mc=np.frombuffer(b''.join(m5), np.uint8) if any5 else None,
        mod_6ma=np.frombuffer(b''.join(m6), np.uint8) if any6 else None,
        cigar=np.array(cig, np.uint32), keys=keys, read_end=read_end,
        base_aux0=aux_planes[0], base_aux1=aux_planes[1], base_aux2=aux_planes[2])

  @classmethod
  def from_cram(cls, path: str, fetch_reference, contig: Optional[str] = None, start: int = 0,
                end: int = 1 << 62, min_mapping_quality: int = 0, keep_duplicates: bool = False,
                keep_supplementary: bool = False, keep_secondary: bool = False, keep_failed_qc: bool = False,
                keep_improperly_placed: bool = False, use_original_quality_scores: bool = False,
                n_threads: int = 4, parse_base_modifications: bool = False,
                parse_flow_tags: bool = False) -> 'ReadTable':
    """Native CRAM 3.0 -> packed table (dv_cram_read_region, include/dvhip.h;
    deepvariant_amd/csrc/cram_reader.cpp): the same reads, in the same order, `from_bam` yields for
    the BAM of the same alignments.  `fetch_reference(contig, start, end) -> bases` supplies what a
    CRAM written against an external reference leaves out (the reference's --use_ref_for_cram);
    None = embedded references only."""
    import ctypes as C
    lib = _lib.lib()
    req = _lib.DvReadRequirements(int(keep_duplicates), int(keep_failed_qc), int(keep_secondary),
                                  int(keep_supplementary), int(keep_improperly_placed),
                                  int(min_mapping_quality), int(use_original_quality_scores),
                                  int(parse_base_modifications), int(parse_flow_tags))
    raised = []

    def fetch(_ctx, name, lo, hi, out, n_out):
      try:
        bases = fetch_reference(name.decode(), int(lo), int(hi)).encode('latin-1')[:max(0, hi - lo)]
        C.memmove(out, bases, len(bases))
        n_out[0] = len(bases)
        return 0
      except BaseException as e:      # pylint: disable=broad-except   (re-raised below, on the caller's thread)
        raised.append(e)
        return 1
    callback = _lib.REF_FETCH_FN(fetch) if fetch_reference is not None else C.cast(None, _lib.REF_FETCH_FN)
    handle = C.c_void_p()
    status = lib.dv_cram_read_region(
        path.encode(), contig.encode() if contig is not None else None, int(start), int(min(end, (1 << 62))),
        C.byref(req), callback, None, int(n_threads), C.byref(handle))
    if raised:
      raise raised[0]
    if status in (_lib.DV_ERR_BAD_INPUT, _lib.DV_ERR_UNSUPPORTED):      # what the Python decoder raises for the same files
      raise ValueError(_lib.last_error())
    _lib.check(status)
    try:
      return cls._from_native_table(handle)
    finally:
      lib.dv_read_table_free(handle)

  @classmethod
  def from_bam(cls, path: str, contig: Optional[str] = None, start: int = 0,
               end: int = 1 << 62, min_mapping_quality: int = 0,
               keep_duplicates: bool = False, keep_supplementary: bool = False,
               keep_secondary: bool = False, keep_failed_qc: bool = False,
               keep_improperly_placed: bool = False, n_threads: int = 4,
               use_original_quality_scores: bool = False, parse_base_modifications: bool = False,
               parse_flow_tags: bool = False) -> 'ReadTable':
    """Native BAM -> packed table (dv_bam_read_region, include/dvhip.h): the reads of
    `contig` overlapping [start, end) that pass nucleus' ReadRequirements, in file order.
    `parse_base_modifications`: the MM / ML / MN tags become the 5mC / 6mA planes (nucleus' ParseBaseModifications);
    `parse_flow_tags`: the Ultima tp / t0 tags become per-base planes (flow_tp / flow_t0)."""
    import ctypes as C
    lib = _lib.lib()
    req = _lib.DvReadRequirements(int(keep_duplicates), int(keep_failed_qc), int(keep_secondary),
                                  int(keep_supplementary), int(keep_improperly_placed),
                                  int(min_mapping_quality), int(use_original_quality_scores),
                                  int(parse_base_modifications), int(parse_flow_tags))
    handle = C.c_void_p()
    _lib.check(lib.dv_bam_read_region(
        path.encode(), contig.encode() if contig is not None else None, int(start),
        int(min(end, (1 << 62))), C.byref(req), int(n_threads), C.byref(handle)))
    try:
      return cls._from_native_table(handle)
    finally:
      lib.dv_read_table_free(handle)

  @classmethod
  def _from_native_table(cls, handle) -> 'ReadTable':
    """Copies a dv_read_table (what the native BAM / CRAM decoders return) into numpy arrays."""
    import ctypes as C
    lib = _lib.lib()
    b = _lib.DvBatch()
    _lib.check(lib.dv_read_table_fill_batch(handle, C.byref(b)))
    n = b.n_reads

    def arr(ptr, dtype, count):
      if not count:
        return np.zeros(0, dtype)
      buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
      return np.frombuffer(buf, dtype=dtype, count=count).copy()

    ends = arr(lib.dv_read_table_ends(handle), np.int64, n)
    blob, offs, rns, nbytes = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    _lib.check(lib.dv_read_table_names(handle, C.byref(blob), C.byref(offs), C.byref(rns),
                                       C.byref(nbytes)))
    names = bytes(arr(blob.value, np.uint8, nbytes.value)).decode().split('\0')[:n]
    read_numbers = arr(rns.value, np.uint8, n)
    keys = ['%s/%d' % (nm, rn) for nm, rn in zip(names, read_numbers.tolist())]
    m5, m6, tp, t0, present = (C.c_void_p() for _ in range(5))
    _lib.check(lib.dv_read_table_aux_planes(handle, C.byref(m5), C.byref(m6), C.byref(tp), C.byref(t0),
                                            C.byref(present)))
    return cls(
        n_reads=n, read_pos=arr(b.read_pos, np.int32, n), read_sort_pos=None,
        read_seq_off=arr(b.read_seq_off, np.uint32, n + 1),
        read_cigar_off=arr(b.read_cigar_off, np.uint32, n + 1),
        read_mapq=arr(b.read_mapq, np.uint8, n), read_flags=arr(b.read_flags, np.uint8, n),
        read_frag_len=arr(b.read_frag_len, np.int32, n), read_hp=arr(b.read_hp, np.int32, n),
        read_name_rank=arr(b.read_name_rank, np.uint32, n), read_aux=None,
        bases=arr(b.bases, np.uint8, b.n_bases), quals=arr(b.quals, np.uint8, b.n_bases),
        mod_5mc=arr(m5.value, np.uint8, b.n_bases) if m5.value else None,
        mod_6ma=arr(m6.value, np.uint8, b.n_bases) if m6.value else None,
        cigar=arr(b.cigar, np.uint32, b.n_cigar), keys=keys, read_end=ends,
        flow_tp=arr(tp.value, np.int8, b.n_bases) if tp.value else None,
        flow_t0=arr(t0.value, np.uint8, b.n_bases) if t0.value else None,
        flow_present=arr(present.value, np.uint8, n) if present.value else None)

  def to_reads(self, reference_name: str) -> List:
    """Read objects (dv_types.Read) back from the packed table -- what the region chain's host
    stages (realigner glue, phasing) take.  The table does not keep pairing / QC flags (the
    native reader has already applied the read requirements), so number_reads is 2 and those
    flags are False."""
    make = self.read_factory(reference_name)
    return [make(i) for i in range(self.n_reads)]

  def read_factory(self, reference_name: str):
    """-> f(i) = the Read object of table row i.  The table's arrays are unpacked into Python
    lists once, so a caller that needs only some rows (make_examples.RegionReads: the reads of the
    regions of ITS task) pays object construction for those rows only."""
    seq = self.bases.tobytes().decode('latin-1')
    quals = self.quals.tobytes()
    seq_off = self.read_seq_off.tolist()
    cig_off = self.read_cigar_off.tolist()
    ops = (self.cigar & 15).tolist()
    lens = (self.cigar >> 4).tolist()
    pos = self.read_pos.tolist()
    mapq = self.read_mapq.tolist()
    flags = self.read_flags.tolist()
    frag = self.read_frag_len.tolist()
    hp = self.read_hp.tolist()
    keys = self.keys
    raw = self.bases.tobytes()
    words = self.cigar.tolist()
    ends = self.read_end.tolist()
    plain = self.mod_5mc is None and self.mod_6ma is None and self.flow_tp is None
    strand_bits = DV_READ_REVERSE | DV_READ_SUPPLEMENTARY
    m5 = self.mod_5mc.tobytes() if self.mod_5mc is not None else None
    m6 = self.mod_6ma.tobytes() if self.mod_6ma is not None else None
    flow_tp, flow_t0 = self.flow_tp, self.flow_t0
    flow_present = self.flow_present.tolist() if self.flow_present is not None else None

    def make_alignment(i: int):
      return T.LinearAlignment(
          position=T.Position(reference_name, pos[i], bool(flags[i] & DV_READ_REVERSE)),
          mapping_quality=mapq[i],
          cigar=[T.CigarUnit(ops[k], lens[k]) for k in range(cig_off[i], cig_off[i + 1])])

    def make(i: int):
      name, _, number = keys[i].rpartition('/')
      info = {}
      if hp[i] != _lib.DV_HP_NONE:
        info['HP'] = T.ListValue(values=[T.Value(int_value=hp[i])])
      s0, s1 = seq_off[i], seq_off[i + 1]
      if not plain:
        # base modifications / flow-space tags: an ordinary Read carrying what nucleus' reader would have parsed
        # (Read.base_modifications, sam_reader.cc:855-862; info['tp'] a list of ints, info['t0'] one string)
        mods = {}
        if m5 is not None and flags[i] & DV_READ_HAS_5MC:
          mods[T.K5MC] = m5[s0:s1]
        if m6 is not None and flags[i] & DV_READ_HAS_6MA:
          mods[T.K6MA] = m6[s0:s1]
        if flow_present is not None and flow_present[i] & 1:
          info['tp'] = T.ListValue(values=[T.Value(int_value=int(v)) for v in flow_tp[s0:s1]])
        if flow_present is not None and flow_present[i] & 2:
          info['t0'] = T.ListValue(values=[T.Value(string_value=(flow_t0[s0:s1] + 33).tobytes().decode('latin-1'))])
        return T.Read(
            fragment_name=name, read_number=int(number), number_reads=2,
            supplementary_alignment=bool(flags[i] & DV_READ_SUPPLEMENTARY), fragment_length=frag[i],
            aligned_sequence=seq[s0:s1], aligned_quality=quals[s0:s1], alignment=make_alignment(i), info=info,
            base_modifications=mods)
      # the row this Read is made from IS its packed form (ReadTable._pack_read's record, alignment
      # slot empty until the object exists): from_reads on a list that contains it copies the row
      read = object.__new__(LazyRead)
      read.__dict__ = {
          'fragment_name': name, 'read_number': int(number), 'number_reads': 2, 'proper_placement': False,
          'duplicate_fragment': False, 'failed_vendor_quality_checks': False, 'secondary_alignment': False,
          'supplementary_alignment': bool(flags[i] & DV_READ_SUPPLEMENTARY), 'fragment_length': frag[i],
          'aligned_sequence': seq[s0:s1], 'aligned_quality': quals[s0:s1], 'info': info, 'base_modifications': {},
          '_dv_alignment': functools.partial(make_alignment, i), '_dv_contig': reference_name,
          '_dv_packed': (None, pos[i], mapq[i], flags[i] & strand_bits, frag[i], raw[s0:s1], quals[s0:s1], None, None,
                         words[cig_off[i]:cig_off[i + 1]], ends[i], None, keys[i], (name.encode(), int(number)))}
      return read
    return make

  # ---- the table as the unit the region chain passes along (make_examples' table path): rows
  # are selected and realigned as arrays; no Read objects in between
  @staticmethod
  def _segments(offsets: np.ndarray, rows: np.ndarray):
    """Indices of the elements of the variable-length segments `rows` of a flat array with
    `offsets`, concatenated in that order, and the new offsets."""
    off = offsets.astype(np.int64)
    starts, lengths = off[rows], off[rows + 1] - off[rows]
    new_off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lengths, out=new_off[1:])
    total = int(new_off[-1])
    idx = np.arange(total, dtype=np.int64) + np.repeat(starts - new_off[:-1], lengths)
    return idx, new_off

  def take(self, rows) -> 'ReadTable':
    """The table of the given rows, in the given order (a region's reads out of a decoded block,
    the reads the realigner returns in its own order, ...)."""
    rows = np.asarray(rows, np.int64)
    seq_idx, seq_off = self._segments(self.read_seq_off, rows)
    cig_idx, cig_off = self._segments(self.read_cigar_off, rows)
    ranks = np.unique(self.read_name_rank[rows], return_inverse=True)[1].astype(np.uint32) if len(rows) else \
        np.zeros(0, np.uint32)
    per_base = lambda a: None if a is None else a[seq_idx]     # noqa: E731
    keys = self.keys
    return ReadTable(
        n_reads=len(rows), read_pos=self.read_pos[rows],
        read_sort_pos=None if self.read_sort_pos is None else self.read_sort_pos[rows],
        read_seq_off=seq_off.astype(np.uint32), read_cigar_off=cig_off.astype(np.uint32),
        read_mapq=self.read_mapq[rows], read_flags=self.read_flags[rows], read_frag_len=self.read_frag_len[rows],
        read_hp=self.read_hp[rows], read_name_rank=ranks.reshape(len(rows)),
        read_aux=None if self.read_aux is None else self.read_aux[rows],
        bases=self.bases[seq_idx], quals=self.quals[seq_idx], mod_5mc=per_base(self.mod_5mc),
        mod_6ma=per_base(self.mod_6ma), cigar=self.cigar[cig_idx], keys=[keys[i] for i in rows.tolist()],
        read_end=self.read_end[rows], base_aux0=per_base(self.base_aux0), base_aux1=per_base(self.base_aux1),
        base_aux2=per_base(self.base_aux2), flow_tp=per_base(self.flow_tp), flow_t0=per_base(self.flow_t0),
        flow_present=None if self.flow_present is None else self.flow_present[rows])

  def with_alignments(self, rows, positions, cigars: Sequence[np.ndarray]) -> 'ReadTable':
    """A copy in which row rows[k] starts at positions[k] with the CIGAR words cigars[k]
    (what FastPassAligner::RealignReadsToReference changes of a read, fast_pass_aligner.cc:510-590);
    everything else is shared with `self`."""
    if not len(rows):
      return self
    off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum([len(w) for w in cigars], out=off[1:])
    words = np.concatenate([np.asarray(w, np.uint32) for w in cigars]) if off[-1] else np.zeros(0, np.uint32)
    return self.with_alignments_csr(rows, positions, off, words)

  def with_alignments_csr(self, rows, positions, cigar_off, words) -> 'ReadTable':
    """`with_alignments` with the new CIGARs as one array: row rows[k] gets words[cigar_off[k]:cigar_off[k + 1]]
    (the form dv_realign_regions returns them in).  Array operations only."""
    rows = np.asarray(rows, np.int64)
    if not len(rows):
      return self
    cigar_off = np.asarray(cigar_off, np.int64)
    words = np.asarray(words, np.uint32)
    lengths = np.diff(self.read_cigar_off.astype(np.int64))
    in_len = np.diff(cigar_off)
    new_len = lengths.copy()
    new_len[rows] = in_len
    new_off = np.zeros(self.n_reads + 1, np.int64)
    np.cumsum(new_len, out=new_off[1:])
    cigar = np.zeros(int(new_off[-1]), np.uint32)
    # unchanged rows: one segment copy out of the old array; changed rows: one out of `words`
    same = np.ones(self.n_reads, bool)
    same[rows] = False
    same_rows = np.nonzero(same)[0]
    src, _ = self._segments(self.read_cigar_off, same_rows)
    dst = np.arange(len(src), dtype=np.int64) + np.repeat(
        new_off[same_rows] - (np.cumsum(lengths[same_rows]) - lengths[same_rows]), lengths[same_rows])
    cigar[dst] = self.cigar[src]
    total = int(cigar_off[-1] - cigar_off[0])
    src_w = np.arange(total, dtype=np.int64) + cigar_off[0]
    dst_w = np.arange(total, dtype=np.int64) + np.repeat(new_off[rows] - (cigar_off[:-1] - cigar_off[0]), in_len)
    cigar[dst_w] = words[src_w]
    pos = self.read_pos.copy()
    end = self.read_end.copy()
    pos[rows] = np.asarray(positions).astype(pos.dtype)
    # alignment end = start + the reference bases its operations consume (M, =, X, D, N)
    w = words[src_w]
    ops = w & 15
    on_ref = np.where((ops == 1) | (ops == 8) | (ops == 9) | (ops == 3) | (ops == 4), (w >> 4).astype(np.int64), 0)
    span = np.bincount(np.repeat(np.arange(len(rows)), in_len), weights=on_ref, minlength=len(rows)).astype(np.int64)
    end[rows] = (np.asarray(positions, np.int64) + span).astype(end.dtype)
    return dataclasses.replace(self, read_pos=pos, read_end=end, read_cigar_off=new_off.astype(np.uint32), cigar=cigar)

  def query(self, start: int, end: int) -> np.ndarray:
    """InMemoryReader::Query (make_examples_native.cc:802-810): caller order."""
    return np.nonzero((end > self.read_pos) & (start < self.read_end))[0].astype(
        np.uint32)


def concat_tables(tables: Sequence[ReadTable]) -> ReadTable:
  """Rows of several tables of the same kind, one after the other."""
  tables = [t for t in tables if t.n_reads]
  if len(tables) == 1:
    return tables[0]
  if not tables:
    raise ValueError('concat_tables: nothing to concatenate')
  def cat(name):
    parts = [getattr(t, name) for t in tables]
    return None if any(p is None for p in parts) else np.concatenate(parts)
  def offsets(name):
    out = [np.zeros(1, np.int64)]
    base = 0
    for t in tables:
      o = getattr(t, name).astype(np.int64)
      out.append(o[1:] + base)
      base += int(o[-1])
    return np.concatenate(out).astype(np.uint32)
  keys = [k for t in tables for k in t.keys]
  order = {k: i for i, k in enumerate(sorted(set((k.rpartition('/')[0].encode(), int(k.rpartition('/')[2])) for k in keys)))}
  ranks = np.array([order[(k.rpartition('/')[0].encode(), int(k.rpartition('/')[2]))] for k in keys], np.uint32)
  return ReadTable(
      n_reads=len(keys), read_pos=cat('read_pos'), read_sort_pos=cat('read_sort_pos'),
      read_seq_off=offsets('read_seq_off'), read_cigar_off=offsets('read_cigar_off'), read_mapq=cat('read_mapq'),
      read_flags=cat('read_flags'), read_frag_len=cat('read_frag_len'), read_hp=cat('read_hp'), read_name_rank=ranks,
      read_aux=cat('read_aux'), bases=cat('bases'), quals=cat('quals'), mod_5mc=cat('mod_5mc'), mod_6ma=cat('mod_6ma'),
      cigar=cat('cigar'), keys=keys, read_end=cat('read_end'), base_aux0=cat('base_aux0'), base_aux1=cat('base_aux1'),
      base_aux2=cat('base_aux2'), flow_tp=cat('flow_tp'), flow_t0=cat('flow_t0'), flow_present=cat('flow_present'))


def support_codes(dv_call, alt_alleles: Sequence[str], table: ReadTable,
                  read_idx: Sequence[int]) -> np.ndarray:
  """ReadSupportsAlt for each listed read: 0 ref / 1 this alt / 2 other alt."""
  first_alt: Dict[str, str] = {}
  support = dv_call.allele_support
  for alt in dv_call.variant.alternate_bases:
    if alt in support:
      for name in support[alt].read_names:
        first_alt.setdefault(name, alt)
  alts = set(alt_alleles)
  out = np.zeros(len(read_idx), np.uint8)
  for j, r in enumerate(read_idx):
    a = first_alt.get(table.keys[r])
    if a is not None:
      out[j] = 1 if a in alts else 2
  return out


def allele_groups(dv_call, table: ReadTable, read_idx: Sequence[int]
                  ) -> np.ndarray:
  """Allele-support sort group (pileup_image_native.cc:346-393)."""
  alts = list(dv_call.variant.alternate_bases)
  group_of: Dict[str, int] = {}
  for i, alt in enumerate(alts):
    if alt in dv_call.allele_support:
      for name in dv_call.allele_support[alt].read_names:
        group_of[name] = i
  if len(alts) > 255:
    raise ValueError('too many alt alleles')
  return np.array([group_of.get(table.keys[r], len(alts)) for r in read_idx],
                  np.uint8)


def allele_frequency_pixels(pic_options, dv_call, alt_alleles, table, read_idx
                            ) -> np.ndarray:
  """AlleleFrequencyChannel (channels/allele_frequency_channel.cc:53-119)."""
  alts = set(alt_alleles)
  out = np.zeros(len(read_idx), np.uint8)
  support = dv_call.allele_support
  min_af = np.float32(pic_options.min_non_zero_allele_frequency)
  log10_min = np.float32(np.log10(np.float64(min_af)))
  for j, r in enumerate(read_idx):
    key = table.keys[r]
    af = 0.0
    done = False
    for alt in dv_call.variant.alternate_bases:
      if alt in support and not done:
        for name in support[alt].read_names:
          if name == key and alt in alts:
            af = dv_call.allele_frequency.get(alt, 0.0)
            done = True
            break
    af = np.float32(af)
    if af > min_af:
      log10_af = np.float32(np.log10(np.float64(af)))
      out[j] = int(((log10_min - log10_af) / log10_min) * np.float32(254)) & 0xFF
  return out


def allele_sample_probability_pixels(dv_call, table, read_idx) -> np.ndarray:
  """AlleleSampleProbabilityChannel (channels/allele_sample_probability_channel.cc:43-98): the share of the site's
  reads that support the read's own allele (the reference allele for a read no allele lists), square-rooted.  The
  reference counts `total_reads` while it walks allele_support -- a proto map -- up to the read's allele, so for a
  read of any but the first allele its value depends on the map's iteration order, which protobuf leaves
  unspecified.  Here the alleles are walked in KEY order: what the reference's code does when compiled against an
  ordered map (oracle/_ref)."""
  out = np.zeros(len(read_idx), np.uint8)
  support = dv_call.allele_support
  alleles = sorted(support)
  n_ref = len(dv_call.ref_support)
  for j, r in enumerate(read_idx):
    key = table.keys[r]
    total = mine = 0
    found = False
    for allele in alleles:
      names = support[allele].read_names
      total += len(names)
      if key in names:
        mine = len(names)
        found = True
        break
    if not found:
      mine = n_ref
    total += n_ref
    if total == 0:
      continue
    value = np.float32(min(max(np.float32(mine), np.float32(0.0)), np.float32(total)))
    probability = float(np.float32(value) / np.float32(total))       # float division, then double
    out[j] = int(np.float32(254.0) * np.sqrt(np.float64(probability))) & 0xFF
  return out


def fuzzy_read_supports_alt(dv_call, alt_alleles: Sequence[str], key: str, hp_value: int) -> int:
  """ReadSupportsVariantFuzzyChannel::ReadSupportsAlt
  (channels/read_supports_variant_fuzzy_channel.cc:119-288) for the read `key` with HP tag
  `hp_value` (0 = no tag): 1 = supports an alt of the image, 10 / 9 = supports another
  allele of the same phase that is one / two bases longer or shorter than an alt of the
  image, 2 = supports some other alt, 0 = none of these."""
  variant = dv_call.variant
  all_alts = list(variant.alternate_bases)
  image_alts = list(alt_alleles)
  # CalculateAlelePhases: ALT_PS value i + 1 belongs to alt allele i (value 0 is the reference's)
  phases = [0] * len(all_alts)
  info = getattr(variant, 'info', None) or {}
  if 'ALT_PS' in info:
    values = info['ALT_PS'].values
    for i in range(len(all_alts)):
      phases[i] = int(values[i + 1].int_value) if len(values) > i + 1 else 0

  def read_support(allele: str, names) -> int:   # CalculateReadSupport
    if key not in names:
      return 0
    if allele in image_alts:
      return 1
    for image_alt in image_alts:
      g = all_alts.index(image_alt) if image_alt in all_alts else len(all_alts)
      if g >= len(phases):   # CHECK_LT(image_alt_allele_global_index, alt_allele_phases.size())
        raise ValueError('alt allele %r of the image is not an alt of the candidate' % image_alt)
      if phases[g] == 0 or hp_value == 0 or phases[g] == hp_value:
        d = abs(len(image_alt) - len(allele))
        if d == 1:
          return 10
        if d == 2:
          return 9
    return 2

  support = dv_call.allele_support
  for alt in all_alts:
    if alt in support:
      rs = read_support(alt, support[alt].read_names)
      if rs in (1, 10, 9):
        return rs
  rejected = getattr(dv_call, 'rejected_allele_support', None) or {}
  for alt in getattr(variant, 'alternate_bases_rejected', None) or []:
    if alt in rejected:
      rs = read_support(alt, rejected[alt].read_names)
      if rs != 0:
        return rs
  if dv_call.ref_support:
    rs = read_support(variant.reference_bases, dv_call.ref_support)
    if rs in (10, 9):
      return rs
  return 0


def fuzzy_support_color(pic_options, code: int) -> int:
  """ReadSupportsVariantFuzzyChannel::SupportsAltColor (:290-312), fp32 like the reference."""
  if code == 0:
    alpha = np.float32(pic_options.allele_unsupporting_read_alpha)
  elif code == 1:
    alpha = np.float32(pic_options.allele_supporting_read_alpha)
  elif code == 10:
    alpha = np.float32(0.90)
  elif code == 9:
    alpha = np.float32(0.80)
  elif code == 8:
    alpha = np.float32(0.70)
  elif code == 2:
    alpha = np.float32(pic_options.other_allele_supporting_read_alpha)
  else:
    raise ValueError('read_supports_alt can only be 0/1/8/9/10/2')
  return int(np.float32(254.0) * alpha) & 0xFF


def fuzzy_support_pixels(pic_options, dv_call, alt_alleles, table, read_idx) -> np.ndarray:
  """The read_supports_variant_fuzzy pixel of every listed read (constant along the read,
  so it travels in list_aux like the allele-frequency pixel)."""
  out = np.zeros(len(read_idx), np.uint8)
  sets = _FuzzyCall(dv_call)
  for j, r in enumerate(read_idx):
    hp = int(table.read_hp[r])
    code = fuzzy_read_supports_alt(sets, alt_alleles, table.keys[r],
                                   0 if hp == _lib.DV_HP_NONE else hp)
    out[j] = fuzzy_support_color(pic_options, code)
  return out


class _FuzzyCall:
  """A DeepVariantCall view whose read-name lists are sets (membership is all the fuzzy
  channel asks of them), so a deep pile-up does not rescan the lists per read."""

  class _Names:
    def __init__(self, names):
      self.read_names = frozenset(names)

  def __init__(self, dv_call):
    self.variant = dv_call.variant
    self.allele_support = {k: self._Names(v.read_names)
                           for k, v in dv_call.allele_support.items()}
    rej = getattr(dv_call, 'rejected_allele_support', None) or {}
    self.rejected_allele_support = {k: self._Names(v.read_names) for k, v in rej.items()}
    self.ref_support = frozenset(dv_call.ref_support or [])


def _native_names(table: 'ReadTable'):
  """(names blob, offsets, read numbers) of a table's keys, cached on the table."""
  cached = table.__dict__.get('_names_cache')
  if cached is None:
    names, nums = [], np.zeros(table.n_reads, np.uint8)
    for i, k in enumerate(table.keys):
      name, num = k.rsplit('/', 1)
      names.append(name.encode())
      nums[i] = int(num)
    lens = np.array([len(x) + 1 for x in names], np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint32) if names \
        else np.zeros(0, np.uint32)
    blob = np.frombuffer(b'\0'.join(names) + b'\0', np.uint8).copy() if names \
        else np.zeros(1, np.uint8)
    cached = (blob, offs, nums)
    table.__dict__['_names_cache'] = cached
  return cached


def pack_region_native(table: 'ReadTable', candidates: Sequence, combos: Sequence[Sequence[Sequence[str]]],
                       ref_windows: Sequence[Optional[str]], width: int,
                       read_overlap_buffer_bp: int, pileup_height: int, example_bytes: int,
                       use_groups: bool = False):
  """dv_pack_region (include/dvhip.h): the per-candidate Query, the item per alt
  combination and the support codes / allele groups of a whole region in one native call.

  candidates[i] is a DeepVariantCall, combos[i] its alt combinations (lists of alleles,
  make_examples_native.alt_allele_combinations), ref_windows[i] its reference bases for the
  pileup ('' / None = contig edge, candidate skipped).  -> (PackedBatch, [(candidate index,
  alt combination)]) exactly like ExamplesGenerator._plan_region's Python path."""
  import ctypes as C
  lib = _lib.lib()
  blob, offs, nums = _native_names(table)
  read_pos = np.ascontiguousarray(table.read_pos, np.int32)
  read_end = np.ascontiguousarray(table.read_end, np.int64)
  reads = _lib.DvPackReads(table.n_reads, read_pos.ctypes.data, read_end.ctypes.data,
                           blob.ctypes.data, offs.ctypes.data if offs.size else None,
                           nums.ctypes.data if nums.size else None)
  opt = _lib.DvPackOptions(int(width), int(read_overlap_buffer_bp), int(pileup_height),
                           int(os.environ.get('DV_PACK_THREADS', '4')), int(example_bytes))
  n = len(candidates)
  cands = (_lib.DvPackCandidate * max(n, 1))()
  masks: List[int] = []
  keys: List[bytes] = []
  alts_of: List[int] = []
  batch = PackedBatch(table=table, width=width)
  for i, cand in enumerate(candidates):
    v = cand.variant
    alts = list(v.alternate_bases)
    c = cands[i]
    c.start, c.end, c.n_alts = int(v.start), int(v.end), len(alts)
    c.ref_idx = batch.add_ref_window(ref_windows[i]) if ref_windows[i] else -1
    c.first_combo, c.n_combos = len(masks), len(combos[i])
    for combo in combos[i]:
      m = 0
      for a in combo:
        m |= 1 << alts.index(a)
      masks.append(m)
    c.first_support = len(keys)
    for ai, alt in enumerate(alts):
      if alt in cand.allele_support:
        for name in cand.allele_support[alt].read_names:
          keys.append(name.encode())
          alts_of.append(ai)
    c.n_support = len(keys) - c.first_support
  mask_arr = np.array(masks or [0], np.uint32)
  key_lens = np.array([len(k) + 1 for k in keys], np.int64)
  key_off = (np.concatenate([[0], np.cumsum(key_lens)[:-1]]).astype(np.uint32) if keys
             else np.zeros(1, np.uint32))
  key_blob = np.frombuffer(b'\0'.join(keys) + b'\0', np.uint8).copy()
  alt_arr = np.array(alts_of or [0], np.uint8)
  handle = C.c_void_p()
  _lib.check(lib.dv_pack_region(C.byref(reads), C.byref(opt), n, cands, mask_arr.ctypes.data,
                                key_blob.ctypes.data, key_off.ctypes.data, alt_arr.ctypes.data,
                                C.byref(handle)))
  try:
    b = _lib.DvBatch()
    _lib.check(lib.dv_packed_region_fill_batch(handle, int(use_groups), C.byref(b)))
    ic, im = C.c_void_p(), C.c_void_p()
    n_items = lib.dv_packed_region_items(handle, C.byref(ic), C.byref(im))

    def arr(ptr, dtype, count):
      if not count:
        return np.zeros(0, dtype)
      buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
      return np.frombuffer(buf, dtype=dtype, count=count).copy()

    off = arr(b.item_list_off, np.uint32, n_items + 1)
    batch.item_variant_start = arr(b.item_variant_start, np.int32, n_items).tolist()
    batch.item_image_start = arr(b.item_image_start, np.int32, n_items).tolist()
    batch.item_ref_idx = arr(b.item_ref_idx, np.uint32, n_items).tolist()
    batch.item_height = arr(b.item_height, np.uint16, n_items).tolist()
    batch.item_out_off = arr(b.item_out_off, np.uint64, n_items).tolist()
    batch.item_blank_mask = [0] * n_items
    batch.item_mean_coverage = [0.0] * n_items
    batch.item_list_off = off.tolist() if n_items else [0]
    batch.list_read_chunks = [arr(b.list_read, np.uint32, b.n_list)]
    batch.list_code_chunks = [arr(b.list_code, np.uint8, b.n_list)]
    batch.list_group_chunks = [arr(b.list_group, np.uint8, b.n_list) if use_groups
                               else np.zeros(b.n_list, np.uint8)]
    batch.list_aux_chunks = [np.zeros(b.n_list, np.uint8)]
    batch.use_groups = bool(use_groups)
    item_cand = arr(ic.value, np.int32, n_items)
    item_mask = arr(im.value, np.uint32, n_items)
  finally:
    lib.dv_packed_region_free(handle)
  plan = []
  for ci, m in zip(item_cand.tolist(), item_mask.tolist()):
    alts = list(candidates[ci].variant.alternate_bases)
    plan.append((ci, [a for k, a in enumerate(alts) if (m >> k) & 1]))
  return batch, plan


@dataclasses.dataclass
class PackedBatch:
  """Host image of `dv_batch` (numpy arrays), plus the ctypes view."""
  table: ReadTable
  width: int
  item_variant_start: List[int] = dataclasses.field(default_factory=list)
  item_image_start: List[int] = dataclasses.field(default_factory=list)
  item_ref_idx: List[int] = dataclasses.field(default_factory=list)
  item_height: List[int] = dataclasses.field(default_factory=list)
  item_out_off: List[int] = dataclasses.field(default_factory=list)
  item_blank_mask: List[int] = dataclasses.field(default_factory=list)
  item_mean_coverage: List[float] = dataclasses.field(default_factory=list)
  item_list_off: List[int] = dataclasses.field(default_factory=lambda: [0])
  ref_windows_list: List[bytes] = dataclasses.field(default_factory=list)
  list_read_chunks: List[np.ndarray] = dataclasses.field(default_factory=list)
  list_code_chunks: List[np.ndarray] = dataclasses.field(default_factory=list)
  list_group_chunks: List[np.ndarray] = dataclasses.field(default_factory=list)
  list_aux_chunks: List[np.ndarray] = dataclasses.field(default_factory=list)
  use_groups: bool = False
  use_list_aux: bool = False
  use_ref_aux: bool = False   # build ref_aux0..2 (sequence-context channels' reference rows)
  _frozen: Optional[dict] = None

  # ---- building -----------------------------------------------------------
  def add_ref_window(self, ref_bases: str) -> int:
    if len(ref_bases) != self.width:
      raise ValueError('ref_bases.size() != width')  # pileup_image_native.cc:308
    self.ref_windows_list.append(ref_bases.encode() if isinstance(ref_bases, str) else bytes(ref_bases))
    return len(self.ref_windows_list) - 1

  def add_item(self, variant_start: int, image_start: int, ref_idx: int,
               read_idx: np.ndarray, codes: np.ndarray, height: int,
               out_off: int, blank_mask: int = 0, mean_coverage: float = 0.0,
               groups: Optional[np.ndarray] = None,
               list_aux: Optional[np.ndarray] = None):
    self._frozen = None
    self.item_variant_start.append(int(variant_start))
    self.item_image_start.append(int(image_start))
    self.item_ref_idx.append(int(ref_idx))
    self.item_height.append(int(height))
    self.item_out_off.append(int(out_off))
    self.item_blank_mask.append(int(blank_mask))
    self.item_mean_coverage.append(float(mean_coverage))
    n = len(read_idx)
    self.list_read_chunks.append(np.asarray(read_idx, np.uint32))
    self.list_code_chunks.append(np.asarray(codes, np.uint8))
    self.list_group_chunks.append(
        np.asarray(groups, np.uint8) if groups is not None
        else np.zeros(n, np.uint8))
    self.list_aux_chunks.append(
        np.asarray(list_aux, np.uint8) if list_aux is not None
        else np.zeros(n, np.uint8))
    self.use_groups |= groups is not None
    self.use_list_aux |= list_aux is not None
    self.item_list_off.append(self.item_list_off[-1] + n)

  # ---- views (names = dv_batch fields; the oracle adapter reads them too) --
  def _freeze(self) -> dict:
    if self._frozen is None:
      cat = lambda ch, dt: (np.concatenate(ch).astype(dt) if ch
                            else np.zeros(0, dt))
      self._frozen = dict(
          item_variant_start=np.array(self.item_variant_start, np.int32),
          item_image_start=np.array(self.item_image_start, np.int32),
          item_ref_idx=np.array(self.item_ref_idx, np.uint32),
          item_list_off=np.array(self.item_list_off, np.uint32),
          item_height=np.array(self.item_height, np.uint16),
          item_out_off=np.array(self.item_out_off, np.uint64),
          item_blank_mask=np.array(self.item_blank_mask, np.uint32),
          item_mean_coverage=np.array(self.item_mean_coverage, np.float32),
          ref_windows=np.frombuffer(b''.join(self.ref_windows_list), np.uint8),
          list_read=cat(self.list_read_chunks, np.uint32),
          list_code=cat(self.list_code_chunks, np.uint8),
          list_group=cat(self.list_group_chunks, np.uint8),
          list_aux=cat(self.list_aux_chunks, np.uint8))
      if self.use_ref_aux:
        wins = self.ref_windows_list
        plane = lambda fn: (np.concatenate([fn(w) for w in wins]) if wins
                            else np.zeros(0, np.uint8))
        self._frozen['ref_aux0'] = plane(is_homopolymer_pixels)
        self._frozen['ref_aux1'] = plane(homopolymer_weighted_pixels)
        self._frozen['ref_aux2'] = plane(
            lambda w: np.full(len(w), gc_content_pixel(w), np.uint8))
    return self._frozen

  def __getattr__(self, name):
    if name.startswith('_'):
      raise AttributeError(name)
    fz = self._freeze()
    if name in fz:
      if name == 'list_group' and not self.use_groups:
        return None
      if name == 'list_aux' and not self.use_list_aux:
        return None
      return fz[name]
    if name in ('ref_aux0', 'ref_aux1', 'ref_aux2'):
      return None
    t = self.__dict__.get('table')
    if t is not None and hasattr(t, name):
      return getattr(t, name)
    raise AttributeError(name)

  @property
  def n_items(self) -> int:
    return len(self.item_height)

  @property
  def max_list_len(self) -> int:
    off = self.item_list_off
    return max([off[i + 1] - off[i] for i in range(self.n_items)] or [0])

  def out_bytes(self, out_channels: int) -> int:
    row = self.width * out_channels
    return max([o + h * row for o, h in zip(self.item_out_off,
                                            self.item_height)] or [0])

  def to_ctypes(self):
    """-> (DvBatch, keepalive list) with host pointers."""
    fz = self._freeze()
    t = self.table
    b = _lib.DvBatch()
    keep = []

    def ptr(arr, dtype):
      if arr is None:
        return None
      arr = np.ascontiguousarray(arr, dtype=dtype)
      if arr.size == 0:
        arr = np.zeros(1, dtype)
      keep.append(arr)
      return arr.ctypes.data

    b.memory = _lib.DV_MEM_HOST
    b.n_reads = t.n_reads
    b.read_pos = ptr(t.read_pos, np.int32)
    b.read_sort_pos = ptr(t.read_sort_pos, np.int32)
    b.read_seq_off = ptr(t.read_seq_off, np.uint32)
    b.read_cigar_off = ptr(t.read_cigar_off, np.uint32)
    b.read_mapq = ptr(t.read_mapq, np.uint8)
    b.read_flags = ptr(t.read_flags, np.uint8)
    b.read_frag_len = ptr(t.read_frag_len, np.int32)
    b.read_hp = ptr(t.read_hp, np.int32)
    b.read_name_rank = ptr(t.read_name_rank, np.uint32)
    b.read_aux = ptr(t.read_aux, np.uint8)
    b.bases = ptr(t.bases, np.uint8)
    b.quals = ptr(t.quals, np.uint8)
    b.mod_5mc = ptr(t.mod_5mc, np.uint8)
    b.mod_6ma = ptr(t.mod_6ma, np.uint8)
    b.cigar = ptr(t.cigar, np.uint32)
    b.n_bases = int(t.read_seq_off[-1])
    b.n_cigar = int(t.read_cigar_off[-1])
    b.n_items = self.n_items
    b.item_variant_start = ptr(fz['item_variant_start'], np.int32)
    b.item_image_start = ptr(fz['item_image_start'], np.int32)
    b.item_ref_idx = ptr(fz['item_ref_idx'], np.uint32)
    b.item_list_off = ptr(fz['item_list_off'], np.uint32)
    b.item_height = ptr(fz['item_height'], np.uint16)
    b.item_out_off = ptr(fz['item_out_off'], np.uint64)
    b.item_blank_mask = ptr(fz['item_blank_mask'], np.uint32)
    b.item_mean_coverage = ptr(fz['item_mean_coverage'], np.float32)
    b.ref_windows = ptr(fz['ref_windows'], np.uint8)
    b.n_ref_windows = len(self.ref_windows_list)
    b.list_read = ptr(fz['list_read'], np.uint32)
    b.list_code = ptr(fz['list_code'], np.uint8)
    b.list_group = ptr(fz['list_group'], np.uint8) if self.use_groups else None
    b.list_aux = ptr(fz['list_aux'], np.uint8) if self.use_list_aux else None
    b.base_aux0 = ptr(t.base_aux0, np.uint8)
    b.base_aux1 = ptr(t.base_aux1, np.uint8)
    b.base_aux2 = ptr(t.base_aux2, np.uint8)
    for name in ('ref_aux0', 'ref_aux1', 'ref_aux2'):
      setattr(b, name, ptr(fz.get(name), np.uint8))
    b.n_list = int(fz['item_list_off'][-1])
    b.max_list_len = self.max_list_len
    b.max_cigar_ops, b.max_item_height = self.size_hints()
    return b, keep

  def size_hints(self):
    """(max CIGAR operations of any read, max item height): dv_batch's ABI v7 hints."""
    t = self.table
    off = np.asarray(t.read_cigar_off, np.int64)
    ops = int((off[1:] - off[:-1]).max()) if off.size > 1 else 0
    heights = np.asarray(self._freeze()['item_height'])
    return ops, int(heights.max()) if heights.size else 0


def blank_mask_for(chan_enums: Sequence[int], channels_enum_to_blank) -> int:
  mask = 0
  blank = set(int(x) for x in (channels_enum_to_blank or []))
  for i, e in enumerate(chan_enums):
    if e in blank:
      mask |= 1 << i
  return mask
