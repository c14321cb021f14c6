"""TEST INFRASTRUCTURE -- CPU restatement of the reference's AlleleCounter.

Plain Python loops, one function per reference function, each citing
deepvariant/allelecounter.cc.  Only tests/ may import this module; the product path is
deepvariant_amd/allelecounter.py -> libdvhip.so (allele_counter.hip).

Pinned by tests/test_allelecounter_oracle_cpu.py: the vectors of
deepvariant/allelecounter_test.cc over third_party/nucleus/testdata/test.fasta.

Restated: AlleleCounter::Add (:873-979), MakeIndelReadAllele (:402-469), GetPrevBase
(:386-400), CanBasesBeUsed (:206-229), AddReadAlleles (:471-543, incl. track_ref_reads'
REFERENCE read alleles at candidate positions), SumAlleleCounts (:78-117),
TotalAlleleCounts (:165-176).  Not restated: methylation fields, sample_alleles, NormalizeCigar (the product restates that one on the host:
deepvariant_amd/allelecounter.py, pinned by the reference's NormalizeCigar* vectors).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

REFERENCE, SUBSTITUTION, INSERTION, DELETION, SOFT_CLIP = 1, 2, 3, 4, 5   # deepvariant.proto AlleleType
_CANONICAL = frozenset('ACGT')

# nucleus CigarUnit::Operation
_M, _I, _D, _N, _S, _H, _P, _EQ, _X = 1, 2, 3, 4, 5, 6, 7, 8, 9


class Allele:
  def __init__(self, bases, type_, count=1, is_low_quality=False):
    self.bases, self.type, self.count, self.is_low_quality = bases, type_, count, is_low_quality

  def key(self):
    return (self.bases, self.type, self.count)

  def __repr__(self):
    return 'Allele(%r, %d, %d%s)' % (self.bases, self.type, self.count, ', lowq' if self.is_low_quality else '')


class AlleleCount:
  def __init__(self, position, ref_base):
    self.position, self.ref_base = position, ref_base
    self.ref_supporting_read_count = 0
    self.read_alleles: Dict[str, Allele] = {}
    self.track_ref_reads = False


class _ReadAllele:
  INVALID = -1

  def __init__(self, position=-1, bases='', type_=0, low_quality=False):
    self.position, self.bases, self.type, self.low_quality = position, bases, type_, low_quality

  def skip(self):
    return self.position == self.INVALID


def sum_allele_counts(ac: AlleleCount, include_low_quality=False) -> List[Allele]:
  sums: Dict[Tuple[str, int], int] = {}
  for allele in ac.read_alleles.values():
    if include_low_quality or not allele.is_low_quality:
      sums[(allele.bases, allele.type)] = sums.get((allele.bases, allele.type), 0) + 1
  out = [Allele(b, t, n) for (b, t), n in sorted(sums.items())]
  if ac.ref_supporting_read_count > 0 and not ac.track_ref_reads:
    out.append(Allele(ac.ref_base, REFERENCE, ac.ref_supporting_read_count))
  return out


def total_allele_counts(ac: AlleleCount, include_low_quality=False) -> int:
  n = sum(1 for a in ac.read_alleles.values()
          if (not a.is_low_quality or include_low_quality) and a.type != REFERENCE)
  return n + ac.ref_supporting_read_count


class AlleleCounter:
  """AlleleCounter(ref, range, candidate_positions, options) with reads_interval == range, or
  the full_range form (:349-369) when `full_range` is given."""

  def __init__(self, ref_reader, contig: str, start: int, end: int, min_mapping_quality=0,
               min_base_quality=0, keep_legacy_behavior=False, full_range: Optional[Tuple[int, int]] = None,
               candidate_positions=(), track_ref_reads=False):
    self.ref = ref_reader
    self.contig, self.start, self.end = contig, start, end
    r0, r1 = (min(start, full_range[0]), max(end, full_range[1])) if full_range else (start, end)
    self.reads_start, self.reads_end = r0, r1
    self.ref_bases = ref_reader.get_bases(contig, r0, r1)
    self.min_mapq, self.min_bq, self.legacy = min_mapping_quality, min_base_quality, keep_legacy_behavior
    off = max(start - r0, 0)
    self.counts = [AlleleCount(start + i, self.ref_bases[i + off]) for i in range(end - start)]
    self.track_ref_reads = track_ref_reads
    self.candidate_offsets = frozenset(p - start for p in candidate_positions)    # Init, :305-307
    for c in self.counts:
      c.track_ref_reads = track_ref_reads                                         # :318
    self.n_reads_counted = 0

  # ---- helpers
  def _ref_bases(self, rel_start, length):          # RefBases, :371-384
    a = self.reads_start + rel_start
    if a < 0 or a + length > self.ref.n_bases(self.contig):
      return ''
    return self.ref.get_bases(self.contig, a, a + length)

  def _can_bases_be_used(self, read, offset, length):   # CanBasesBeUsed, :206-229 -> (ok, low_quality)
    total = 0
    for i in range(length):
      q = read.aligned_quality[offset + i]
      total += q
      if q < self.min_bq and self.legacy:
        return False, False
      if read.aligned_sequence[offset + i] not in _CANONICAL:
        return False, False
    return True, (not self.legacy and total < self.min_bq * length)

  def _indel(self, read, interval_offset, ref_offset, read_offset, op, op_len):   # MakeIndelReadAllele
    prev = (self._ref_bases(ref_offset - 1, 1) if read_offset == 0
            else read.aligned_sequence[read_offset - 1])
    low = False
    if not prev or prev not in _CANONICAL:
      return _ReadAllele()
    if op != _D:
      ok, low = self._can_bases_be_used(read, read_offset, op_len)
      if not ok:
        return _ReadAllele()
    if op == _D:
      bases = self._ref_bases(ref_offset, op_len)
      if not bases or any(b not in _CANONICAL for b in bases):
        return _ReadAllele()
      type_ = DELETION
    else:
      bases = read.aligned_sequence[read_offset:read_offset + op_len]
      type_ = INSERTION if op == _I else SOFT_CLIP
    return _ReadAllele(interval_offset - 1, prev + bases, type_, low)

  # ---- Add, :873-979
  def add(self, read, sample='sample'):
    aln = read.alignment
    if aln.mapping_quality < self.min_mapq:
      return
    to_add: List[_ReadAllele] = []
    read_offset = 0
    ref_off = aln.position.position - self.reads_start
    int_off = aln.position.position - self.start
    seq = read.aligned_sequence
    for cu in aln.cigar:
      op, n = cu.operation, cu.operation_length
      if op in (_M, _EQ, _X):
        for i in range(n):
          r, b = ref_off + i, read_offset + i
          if 0 <= r < len(self.ref_bases):
            ok, low = self._can_bases_be_used(read, b, 1)
            if ok:
              type_ = REFERENCE if self.ref_bases[r] == seq[b] else SUBSTITUTION
              to_add.append(_ReadAllele(int_off + i, seq[b], type_, low))
        read_offset += n
        ref_off += n
        int_off += n
      elif op in (_S, _I):
        to_add.append(self._indel(read, int_off, ref_off, read_offset, op, n))
        read_offset += n
      elif op == _D:
        to_add.append(self._indel(read, int_off, ref_off, read_offset, op, n))
        ref_off += n
        int_off += n
      elif op in (_P, _N):
        ref_off += n
        int_off += n
    self._add_read_alleles(read, to_add)
    self.n_reads_counted += 1

  def _add_read_alleles(self, read, to_add):   # :471-543
    key = '%s/%d' % (read.fragment_name, read.read_number)
    for i, ra in enumerate(to_add):
      if ra.skip() or not 0 <= ra.position < len(self.counts):
        continue
      if i + 1 < len(to_add) and ra.position == to_add[i + 1].position:
        continue                      # superseded by the indel that VCF places at the same base
      ac = self.counts[ra.position]
      if ra.type == REFERENCE:
        if not ra.low_quality:
          ac.ref_supporting_read_count += 1
        # a REFERENCE read allele exists only at candidate positions, and only when asked (:504-512)
        if not (self.track_ref_reads and ra.position in self.candidate_offsets):
          continue
      ac.read_alleles[key] = Allele(ra.bases, ra.type, 1, ra.low_quality)
