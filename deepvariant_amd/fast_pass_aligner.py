"""Host mirror of the read realigner over the C ABI (include/dvhip.h, "read realigner").

  FastPassAligner          deepvariant/realigner/fast_pass_aligner.h (class FastPassAligner)
  realign_reads_to_haplotype   deepvariant/alt_aligned_pileup_lib.cc:278-313
  local_align              deepvariant/realigner/ssw.h (Aligner::SetReferenceSequence + Align)

All computation is native (deepvariant_amd/csrc/fast_pass_aligner.cpp, local_align.cpp); this
module only marshals strings and turns results back into dv_types reads.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T

K_REF_ALIGN_MARGIN = 0    # alt_aligned_pileup_lib.cc:62 kRefAlignMargin (the window realigner's is 20)

BUILD_INDEX, INIT_LOCAL_ALIGNER, ALIGN_HAPLOTYPES, POSITION_MAPS, LOCAL_ALIGN_READS, SCORE_THRESHOLD = range(6)


def _strings(items: Sequence[str]):
  arr = (C.c_char_p * max(len(items), 1))()
  for i, s in enumerate(items):
    arr[i] = s.encode() if isinstance(s, str) else bytes(s)
  return arr


class ReadAlignment:
  """ReadAlignment (fast_pass_aligner.h:104-128); position None = kNotAligned."""

  def __init__(self, position=None, cigar='', score=0):
    self.position, self.cigar, self.score = position, cigar, score

  def __eq__(self, other):
    return (self.position, self.cigar, self.score) == (other.position, other.cigar, other.score)

  def __repr__(self):
    return 'ReadAlignment(%r, %r, %r)' % (self.position, self.cigar, self.score)


def _read_alignment(raw: _lib.DvReadAlignment) -> ReadAlignment:
  return ReadAlignment(None if raw.position < 0 else raw.position, raw.cigar.decode(), raw.score)


class FastPassAligner:
  def __init__(self, match=0, mismatch=0, gap_open=0, gap_extend=0, kmer_size=0, read_size=0,
               max_num_of_mismatches=0, realignment_similarity_threshold=0.0, force_alignment=False,
               normalize_reads=False, ref_prefix_len=0, ref_suffix_len=0):
    opt = _lib.DvAlignerOptions(match, mismatch, gap_open, gap_extend, kmer_size, read_size,
                                max_num_of_mismatches, realignment_similarity_threshold,
                                int(force_alignment), int(normalize_reads), ref_prefix_len, ref_suffix_len)
    self._h = C.c_void_p()
    _lib.check(_lib.lib().dv_aligner_create(C.byref(opt), C.byref(self._h)))
    self._n_reads = 0

  def __del__(self):
    if getattr(self, '_h', None):
      _lib.lib().dv_aligner_destroy(self._h)
      self._h = None

  def set_reference(self, reference: str, ref_start: int = 0):
    _lib.check(_lib.lib().dv_aligner_set_reference(self._h, reference.encode(), ref_start))

  def set_haplotypes(self, haplotypes: Sequence[str]):
    _lib.check(_lib.lib().dv_aligner_set_haplotypes(self._h, len(haplotypes), _strings(haplotypes)))

  def set_reads(self, reads: Sequence[str]):
    self._n_reads = len(reads)
    _lib.check(_lib.lib().dv_aligner_set_reads(self._h, len(reads), _strings(reads)))

  def stage(self, which: int, arg: int = 0):
    _lib.check(_lib.lib().dv_aligner_stage(self._h, which, arg))

  def fast_align_reads_to_haplotype(self, haplotype: str, haplotype_score: int = 0):
    out = (_lib.DvReadAlignment * max(self._n_reads, 1))()
    score = C.c_int32(haplotype_score)
    _lib.check(_lib.lib().dv_aligner_fast_align(self._h, haplotype.encode(), C.byref(score), out))
    return score.value, [_read_alignment(out[i]) for i in range(self._n_reads)]

  def haplotype_alignment(self, k: int):
    """-> dict(haplotype_index, haplotype_score, ref_pos, is_reference, cigar)."""
    idx, score, is_ref = C.c_int32(), C.c_int32(), C.c_int32()
    pos = C.c_int64()
    buf = C.create_string_buffer(4096)
    _lib.check(_lib.lib().dv_aligner_haplotype_info(self._h, k, C.byref(idx), C.byref(score), C.byref(pos),
                                                    C.byref(is_ref), buf, len(buf)))
    return dict(haplotype_index=idx.value, haplotype_score=score.value, ref_pos=pos.value,
                is_reference=bool(is_ref.value), cigar=buf.value.decode())

  def read_alignment(self, k: int, read: int) -> ReadAlignment:
    raw = _lib.DvReadAlignment()
    _lib.check(_lib.lib().dv_aligner_read_alignment(self._h, k, read, C.byref(raw)))
    return _read_alignment(raw)

  def calculate_read_to_ref_alignment(self, read_index: int, position: int, read_cigar: str,
                                      haplotype_cigar: str) -> str:
    buf = C.create_string_buffer(4096)
    _lib.check(_lib.lib().dv_aligner_merge_alignment(self._h, read_index, position, read_cigar.encode(),
                                                     haplotype_cigar.encode(), buf, len(buf)))
    return buf.value.decode()

  def is_alignment_normalized(self, cigar: str, ref_offset: int, read: str) -> bool:
    r = _lib.lib().dv_aligner_is_normalized(self._h, cigar.encode(), ref_offset, read.encode())
    if r < 0:
      _lib.check(r)
    return bool(r)

  def score_threshold(self) -> int:
    return _lib.lib().dv_aligner_score_threshold(self._h)

  def kmer_occurrences(self, kmer: str) -> List[Tuple[int, int]]:
    cap = 4096
    reads, offs = (C.c_int32 * cap)(), (C.c_int32 * cap)()
    n = _lib.lib().dv_aligner_kmer_occurrences(self._h, kmer.encode(), cap, reads, offs)
    if n < 0:
      _lib.check(n)
    return [(reads[i], offs[i]) for i in range(min(n, cap))]

  def index_size(self) -> int:
    return _lib.lib().dv_aligner_kmer_occurrences(self._h, b'', 0, None, None)

  def align_reads(self, sequences: Sequence[str]):
    """AlignReads on bare sequences -> [(status, position, [(op, length), ...])]."""
    n = len(sequences)
    self._n_reads += n
    out = (_lib.DvRealignedRead * max(n, 1))()
    words = C.POINTER(C.c_uint32)()
    _lib.check(_lib.lib().dv_aligner_align_reads(self._h, n, _strings(sequences), out, C.byref(words)))
    res = []
    for i in range(n):
      cig = [(words[out[i].cigar_off + k] & 15, words[out[i].cigar_off + k] >> 4) for k in range(out[i].n_cigar)]
      res.append((out[i].status, out[i].position, cig))
    return res

  def align_reads_arrays(self, sequences: Sequence):
    """AlignReads on bare sequences, results as arrays (the table path of the realigner):
    -> (status int32[n], position int64[n], cigar_off int64[n + 1], CIGAR words uint32[...] in the
    packed tables' encoding); status as in `align_reads`."""
    import numpy as np
    n = len(sequences)
    self._n_reads += n
    out = (_lib.DvRealignedRead * max(n, 1))()
    words = C.POINTER(C.c_uint32)()
    _lib.check(_lib.lib().dv_aligner_align_reads(self._h, n, _strings(sequences), out, C.byref(words)))
    rec = np.frombuffer(out, dtype=np.dtype([('status', '<i4'), ('n_cigar', '<i4'), ('position', '<i8'),
                                             ('cigar_off', '<u4'), ('reserved', '<u4')]), count=max(n, 1))[:n]
    off = np.zeros(n + 1, np.int64)
    np.cumsum(rec['n_cigar'], out=off[1:])
    total = int(off[-1])
    w = np.ctypeslib.as_array(words, shape=(total,)).copy() if total else np.zeros(0, np.uint32)
    return rec['status'].copy(), rec['position'].copy(), off, w

  def realign_reads(self, reads: Sequence) -> List[Optional[T.Read]]:
    """FastPassAligner::AlignReads on Read objects (fast_pass_aligner.cc:183-232, :510-590):
    per input read the read with its new alignment, the unchanged read (no better alignment,
    or the merged CIGAR was rejected), or None where the reference returns an empty Read
    (force_alignment and nothing found)."""
    out = []
    for read, (status, position, cigar) in zip(reads, self.align_reads([r.aligned_sequence for r in reads])):
      if status == 2:
        out.append(None)
      elif status == 0:
        out.append(read)
      else:
        out.append(with_alignment(read, position, cigar))
    return out


def with_alignment(read, position: int, cigar) -> T.Read:
  """A copy of `read` whose alignment start and CIGAR are replaced (RealignReadsToReference,
  fast_pass_aligner.cc:510-590: everything else is merged over from the input read)."""
  p = read.alignment.position
  return T.Read(
      fragment_name=read.fragment_name, read_number=read.read_number,
      number_reads=read.number_reads, fragment_length=read.fragment_length,
      proper_placement=read.proper_placement, duplicate_fragment=read.duplicate_fragment,
      failed_vendor_quality_checks=read.failed_vendor_quality_checks,
      secondary_alignment=read.secondary_alignment,
      supplementary_alignment=read.supplementary_alignment,
      aligned_sequence=read.aligned_sequence, aligned_quality=read.aligned_quality,
      alignment=T.LinearAlignment(position=T.Position(p.reference_name, position, p.reverse_strand),
                                  mapping_quality=read.alignment.mapping_quality,
                                  cigar=[T.CigarUnit(op, ln) for op, ln in cigar]),
      info=dict(read.info), base_modifications=dict(read.base_modifications))


def positions_map(cigar: str, haplotype_size: int) -> List[int]:
  out = (C.c_int32 * max(haplotype_size, 1))()
  _lib.check(_lib.lib().dv_positions_map(cigar.encode(), haplotype_size, out))
  return list(out[:haplotype_size])


def merge_cigar_op(cigar: str, op: str, length: int, read_len: int) -> str:
  buf = C.create_string_buffer(cigar.encode(), 4096)
  _lib.check(_lib.lib().dv_merge_cigar_op(buf, len(buf), op.encode(), length, read_len))
  return buf.value.decode()


def local_align(reference: str, query: str, match=2, mismatch=2, gap_open=3, gap_extend=1):
  """One local alignment; defaults are libssw's Aligner() defaults."""
  out = _lib.DvLocalAlignment()
  _lib.check(_lib.lib().dv_local_align(reference.encode(), query.encode(), match, mismatch, gap_open,
                                       gap_extend, C.byref(out)))
  return out


def local_align_many(reference: str, queries: Sequence[str], match=2, mismatch=2, gap_open=3, gap_extend=1):
  """`local_align` of every query against one reference through the 16-lane batch path;
  an entry is None where local_align would raise."""
  n = len(queries)
  out = (_lib.DvLocalAlignment * max(n, 1))()
  _lib.check(_lib.lib().dv_local_align_many(reference.encode(), n, _strings(queries), match, mismatch, gap_open,
                                            gap_extend, out))
  return [None if out[k].score < 0 else out[k] for k in range(n)]


def realign_reads_to_haplotype(haplotype: str, reads: Sequence, contig: str, ref_start: int, ref_end: int,
                               ref_reader, aln_config: Optional[dict] = None) -> List[Optional[T.Read]]:
  """RealignReadsToHaplotype (alt_aligned_pileup_lib.cc:278-313).

  Returns one entry per input read: the read with its new alignment, the unchanged read
  (the merged CIGAR came out empty), or None where the reference returns an empty Read."""
  cfg = dict(aln_config or {})
  cfg['read_size'] = (len(reads[0].aligned_sequence)
                      if reads and len(reads[0].aligned_sequence) > 15 else 200)
  ext_start = max(0, ref_start - K_REF_ALIGN_MARGIN)
  ext_end = min(ref_reader.n_bases(contig), ref_end + K_REF_ALIGN_MARGIN)
  prefix = ref_reader.get_bases(contig, ext_start, ref_start) if ext_start < ref_start else ''
  suffix = ref_reader.get_bases(contig, ref_end, ext_end) if ref_end < ext_end else ''
  target = prefix + haplotype + suffix
  aligner = FastPassAligner(force_alignment=True, ref_prefix_len=ref_start - ext_start,
                            ref_suffix_len=ext_end - ref_end, **cfg)
  aligner.set_reference(target, ext_start)
  aligner.set_haplotypes([target])
  return aligner.realign_reads(reads)
