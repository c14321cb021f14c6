"""Range helpers the realigner uses (third_party/nucleus/util/ranges.py, utils.py and
deepvariant/realigner/utils.py), on dv_types.Range / dv_types.Read."""
from __future__ import annotations

from typing import Optional, Sequence

from deepvariant_amd import dv_types as T

# nucleus CigarUnit::Operation
ALIGNMENT_MATCH, INSERT, DELETE, SKIP, CLIP_SOFT, CLIP_HARD, PAD, SEQUENCE_MATCH, SEQUENCE_MISMATCH = range(1, 10)

CIGAR_ALIGN_OPS = (ALIGNMENT_MATCH, SEQUENCE_MATCH, SEQUENCE_MISMATCH)
CIGAR_INSERT_OPS = (INSERT, CLIP_SOFT)
CIGAR_DELETE_OPS = (DELETE, SKIP)
CIGAR_NO_OPS = (CLIP_HARD,)
CIGAR_OPS = CIGAR_ALIGN_OPS + CIGAR_INSERT_OPS + CIGAR_DELETE_OPS + CIGAR_NO_OPS   # realigner/utils.py:33-43

REF_ADVANCING_OPS = (ALIGNMENT_MATCH, SEQUENCE_MATCH, DELETE, SKIP, SEQUENCE_MISMATCH)   # nucleus util/cigar.py
READ_ADVANCING_OPS = (ALIGNMENT_MATCH, SEQUENCE_MATCH, INSERT, CLIP_SOFT, SEQUENCE_MISMATCH)


def make_range(reference_name: str, start: int, end: int) -> T.Range:
  return T.Range(reference_name, int(start), int(end))


def read_range(read) -> T.Range:
  """utils.read_range: alignment start .. start + reference bases the CIGAR covers."""
  d = getattr(read, '__dict__', None)
  if d is not None and 'alignment' not in d:
    # a packing.LazyRead whose alignment has not been built: the span is in the packed table row
    # it carries (same definition: start + the reference bases of M / = / X / D / N)
    rec = d.get('_dv_packed')
    if rec is not None and rec[0] is None and '_dv_alignment' in d:
      return T.Range(d['_dv_contig'], rec[1], rec[10])
  p = read.alignment.position
  n = sum(c.operation_length for c in read.alignment.cigar if c.operation in REF_ADVANCING_OPS)
  return T.Range(p.reference_name, p.position, p.position + n)


def ranges_overlap(a: T.Range, b: T.Range) -> bool:
  return a.reference_name == b.reference_name and a.end > b.start and b.end > a.start


def overlap_len(a: T.Range, b: T.Range) -> int:
  if a.reference_name != b.reference_name:
    return 0
  return max(0, min(a.end, b.end) - max(a.start, b.start))


def find_max_overlapping(query: T.Range, search: Sequence[T.Range]) -> Optional[int]:
  """ranges.find_max_overlapping (:708-729): the first range with the largest overlap; None if
  nothing overlaps."""
  if not search:
    return None
  overlaps = [overlap_len(query, s) for s in search]
  best = max(range(len(search)), key=lambda i: overlaps[i])
  return None if overlaps[best] == 0 else best


def expand(region: T.Range, n_bp: int, n_contig_bases: Optional[int] = None) -> T.Range:
  """ranges.expand (:732-768)."""
  if n_bp < 0:
    raise ValueError('n_bp must be >= 0 but got {}'.format(n_bp))
  end = region.end + n_bp
  if n_contig_bases is not None:
    end = min(end, n_contig_bases)
  return T.Range(region.reference_name, max(region.start - n_bp, 0), end)
