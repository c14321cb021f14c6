"""Host mirror of the reference's AlleleCounter over the C ABI (`dv_count_alleles`).

  AlleleCounter(ref, range, candidate_positions, options)   deepvariant/allelecounter.h:206-518
  .add(read, sample) / .counts() / .summary_counts()        deepvariant/allelecounter.cc:873-1008
  sum_allele_counts / total_allele_counts                    deepvariant/allelecounter.cc:78-203

track_ref_reads + candidate_positions (the two-pass scheme of make_examples_core.py:2880-2932) are
supported: at candidate positions the reference-supporting reads come back by name too.
The reference adds reads one by one on the CPU; here `add` only queues them and the first
call that needs results packs the queue (packing.ReadTable) and counts the whole region in
ONE kernel launch (deepvariant_amd/csrc/allele_counter.hip).  There is no CPU path.
normalize_cigar restates AlleleCounter::NormalizeCigar (:777-845) for --normalize_reads.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import packing

REFERENCE, SUBSTITUTION, INSERTION, DELETION, SOFT_CLIP = 1, 2, 3, 4, 5   # AlleleType
_EVENT_DTYPE = np.dtype([('position', '<i4'), ('read', '<u4'), ('read_offset', '<u4'),
                         ('length_type', '<u4')])        # dv_allele_event: length | type << 28 | low quality << 31


class Allele:
  """deepvariant.proto Allele: bases, type, count, is_low_quality."""

  def __init__(self, bases: str, type_: int, count: int = 1, is_low_quality: bool = False):
    self.bases, self.type, self.count, self.is_low_quality = bases, type_, count, is_low_quality

  def __repr__(self):
    return 'Allele(%r, %d, %d%s)' % (self.bases, self.type, self.count, ', low quality' if self.is_low_quality else '')


class AlleleCount:
  """deepvariant.proto AlleleCount (the fields the candidate caller reads)."""

  def __init__(self, reference_name: str, position: int, ref_base: str):
    self.position = T.Position(reference_name, position, False)
    self.ref_base = ref_base
    self.ref_supporting_read_count = 0
    self.read_alleles: Dict[str, Allele] = {}
    self.track_ref_reads = False


def sum_allele_counts(allele_count: AlleleCount, include_low_quality: bool = False) -> List[Allele]:
  """SumAlleleCounts (:78-117): std::map order = (bases, type) ascending, then the synthetic
  reference allele."""
  sums: Dict[Tuple[str, int], int] = {}
  for allele in allele_count.read_alleles.values():
    if include_low_quality or not allele.is_low_quality:
      k = (allele.bases, allele.type)
      sums[k] = sums.get(k, 0) + 1
  out = [Allele(b, t, n) for (b, t), n in sorted(sums.items())]
  if allele_count.ref_supporting_read_count > 0 and not getattr(allele_count, 'track_ref_reads', False):
    out.append(Allele(allele_count.ref_base, REFERENCE, allele_count.ref_supporting_read_count))
  return out


def total_allele_counts(allele_count: AlleleCount, include_low_quality: bool = False) -> int:
  """TotalAlleleCounts (:165-176)."""
  n = sum(1 for a in allele_count.read_alleles.values()
          if (not a.is_low_quality or include_low_quality) and a.type != REFERENCE)
  return n + allele_count.ref_supporting_read_count


class AlleleCounter:
  def __init__(self, ref_reader, reference_name: str, start: int, end: int,
               candidate_positions: Sequence[int] = (), min_mapping_quality: int = 0,
               min_base_quality: int = 0, keep_legacy_behavior: bool = False,
               full_range: Optional[Tuple[int, int]] = None, track_ref_reads: bool = False):
    self._track_ref_reads = bool(track_ref_reads)
    self._candidate_positions = np.ascontiguousarray(sorted(int(p) for p in candidate_positions), np.int64)
    self._ref = ref_reader
    self._contig, self._start, self._end = reference_name, int(start), int(end)
    self._reads_start = min(self._start, full_range[0]) if full_range else self._start
    self._reads_end = max(self._end, full_range[1]) if full_range else self._end
    self._opt = (int(min_mapping_quality), int(min_base_quality), bool(keep_legacy_behavior))
    self._reads: List = []
    self._table: Optional[packing.ReadTable] = None
    self._counts: Optional[List[AlleleCount]] = None
    self._alleles: Optional[Dict[int, Dict[str, Allele]]] = None
    self._events = None
    self._event_ctx = None
    self._n_counted = 0

  # ---- the reference's interface
  def interval_length(self) -> int:
    return self._end - self._start

  def interval_start(self) -> int:
    return self._start

  def add(self, read, sample: str = ''):
    if self._table is not None:
      raise ValueError('reads were handed over as a packed table; add() cannot be mixed in')
    self._reads.append(read)
    self._counts = self._alleles = self._events = None

  def add_table(self, table: packing.ReadTable):
    """All reads of the region at once, already packed (packing.ReadTable.from_bam / from_reads)."""
    if self._reads:
      raise ValueError('add() was used; add_table() cannot be mixed in')
    self._table = table
    self._counts = self._alleles = self._events = None

  def _ensure(self):
    if self._events is None:
      self._run()

  def _count_at(self, i: int) -> AlleleCount:
    self._build_alleles()
    c = AlleleCount(self._contig, self._start + i, self._interval_ref[i])
    c.ref_supporting_read_count = int(self._ref_counts[i])
    c.track_ref_reads = self._track_ref_reads
    c.read_alleles = self._alleles.get(i, {})
    return c

  def counts(self) -> List[AlleleCount]:
    """Counts(): one AlleleCount per position of the interval."""
    self._ensure()
    if self._counts is None:
      self._counts = [self._count_at(i) for i in range(len(self._ref_counts))]
    return self._counts

  def counts_with_read_alleles(self) -> List[AlleleCount]:
    """Only the positions some read left a non-reference (or tracked reference) allele at, in
    order -- all a candidate caller or window selector has to look at: a position without read
    alleles has no alternate allele to select."""
    self._ensure()
    self._build_alleles()
    return [self._count_at(i) for i in sorted(self._alleles)]

  def counts_with_alt_support(self, min_count: int) -> List[AlleleCount]:
    """The positions at which at least `min_count` reads left a good-quality allele that is
    neither the reference nor a soft clip, in order.  An alternate allele's count is at most that
    number, so a candidate caller whose thresholds ask for `min_count` reads per allele
    (IsGoodAltAllele, variant_calling_multisample.cc:232-238) finds every candidate among them --
    one position in twenty of the ones `counts_with_read_alleles` returns on 30x Illumina data,
    where most read alleles are lone sequencing errors.  The AlleleCounts are complete (all read
    alleles of the position, low-quality ones included); only those positions' Allele objects
    are built."""
    self._ensure()
    ev = self._events
    if not len(ev):
      return []
    packed = ev['length_type']
    kind = (packed >> 28) & 7
    n = self._end - self._start
    good = ((packed >> 31) == 0) & (kind != REFERENCE) & (kind != SOFT_CLIP) & (ev['position'] >= 0) & (ev['position'] < n)
    per_position = np.bincount(ev['position'][good], minlength=n) if good.any() else np.zeros(n, np.int64)
    wanted = np.nonzero(per_position >= max(1, int(min_count)))[0]
    if not len(wanted):
      return []
    if self._alleles is not None:
      return [self._count_at(int(i)) for i in wanted.tolist()]
    alleles = self._alleles_of(np.nonzero(np.isin(ev['position'], wanted))[0])
    out = []
    for i in wanted.tolist():
      c = AlleleCount(self._contig, self._start + i, self._interval_ref[i])
      c.ref_supporting_read_count = int(self._ref_counts[i])
      c.track_ref_reads = self._track_ref_reads
      c.read_alleles = alleles.get(i, {})
      out.append(c)
    return out

  def ref_supporting_read_counts(self) -> np.ndarray:
    self._ensure()
    return self._ref_counts

  def n_counted_reads(self) -> int:
    self._ensure()
    return self._n_counted

  def summary_counts(self, left_padding: int = 0, right_padding: int = 0):
    """SummaryCounts (:986-1008) -> [(reference_name, position, ref_base, ref_supporting_read_count,
    total_read_count)]."""
    counts = self.counts()
    if left_padding < 0 or right_padding < 0 or left_padding + right_padding >= len(counts):
      raise ValueError('Check failed: left_padding + right_padding < counts_.size()')
    return [(self._contig, c.position.position, c.ref_base, c.ref_supporting_read_count, total_allele_counts(c))
            for c in counts[left_padding:len(counts) - right_padding]]

  # ---- the launch
  def _request(self):
    """-> (packed batch, options struct, the objects they point into, (table, window, w0, interval_ref))."""
    table = self._table if self._table is not None else packing.ReadTable.from_reads(self._reads)
    n_contig = self._ref.n_bases(self._contig)
    # reference window: the reads interval plus room for the longest deletion anchored inside it
    ops, lens = table.cigar & 15, table.cigar >> 4
    margin = int(lens[ops == 3].max()) + 1 if (ops == 3).any() else 1
    w0 = max(0, self._reads_start - 1)
    w1 = min(n_contig, self._reads_end + margin)
    window = self._ref.get_bases(self._contig, w0, w1).encode()
    interval_ref = self._ref.get_bases(self._contig, self._start, self._end)
    opt = _lib.DvAlleleCounterOptions(
        self._start, self._end, self._reads_start, self._reads_end, window, w0, len(window), n_contig,
        self._opt[0], self._opt[1], int(self._opt[2]), int(self._track_ref_reads),
        self._candidate_positions.ctypes.data if len(self._candidate_positions) else None,
        len(self._candidate_positions))
    b, keep = packing.PackedBatch(table=table, width=3).to_ctypes()
    return b, opt, keep, (table, window, w0, interval_ref)

  def _take(self, handle, ctx) -> None:
    """Copies the result out of a dv_allele_counts handle and frees it."""
    table, window, w0, interval_ref = ctx
    lib = _lib.lib()
    try:
      refc = C.POINTER(C.c_int32)()
      events = C.POINTER(_lib.DvAlleleEvent)()
      n_events, n_counted = C.c_uint32(), C.c_int32()
      length = lib.dv_allele_counts_arrays(handle, C.byref(refc), C.byref(events), C.byref(n_events),
                                           C.byref(n_counted))
      self._interval_ref = interval_ref
      self._ref_counts = np.ctypeslib.as_array(refc, shape=(length,)).copy() if length else np.zeros(0, np.int32)
      n_ev = int(n_events.value)
      ev = (np.ctypeslib.as_array(C.cast(events, C.POINTER(C.c_uint8)), shape=(n_ev * 16,)).view(_EVENT_DTYPE).copy()
            if n_ev else np.zeros(0, _EVENT_DTYPE))
      # the Allele objects (texts cut out of the reads / the reference) are built on demand: the
      # window selector's default model only needs the events' footprints (variant_read_window_counts)
      self._events, self._event_ctx = ev, (table, window, w0)
      self._alleles, self._counts, self._n_counted = None, None, int(n_counted.value)
    finally:
      lib.dv_allele_counts_free(handle)

  def _run(self):
    b, opt, keep, ctx = self._request()
    handle = C.c_void_p()
    _lib.check(_lib.lib().dv_count_alleles(C.byref(b), C.byref(opt), C.byref(handle), None))
    del keep
    self._take(handle, ctx)

  @staticmethod
  def run_batch(counters: Sequence['AlleleCounter']) -> None:
    """Counts for several counters (a batch of calling regions, each with its reads added) in ONE
    dv_count_alleles_batch call: one upload, kernels back to back, two synchronisations for the
    whole batch.  Afterwards every counter answers as if it had counted alone."""
    todo = [c for c in counters if c._events is None]      # pylint: disable=protected-access
    if not todo:
      return
    if len(todo) == 1:
      todo[0]._run()                                       # pylint: disable=protected-access
      return
    requests = [c._request() for c in todo]                # pylint: disable=protected-access
    n = len(todo)
    batches = (C.c_void_p * n)(*[C.addressof(r[0]) for r in requests])
    options = (C.c_void_p * n)(*[C.addressof(r[1]) for r in requests])
    handles = (C.c_void_p * n)()
    _lib.check(_lib.lib().dv_count_alleles_batch(n, batches, options, handles, None))
    taken = 0
    try:
      for c, r, h in zip(todo, requests, handles):
        taken += 1                                         # _take frees its handle, also when it raises
        c._take(C.c_void_p(h), r[3])                       # pylint: disable=protected-access
    finally:
      for h in list(handles)[taken:]:                      # a failure part-way: the rest is not leaked
        if h:
          _lib.lib().dv_allele_counts_free(C.c_void_p(h))

  def _build_alleles(self):
    if self._alleles is not None:
      return
    self._alleles, self._counts = self._alleles_of(None), None

  def _alleles_of(self, rows) -> Dict[int, Dict[str, Allele]]:
    """{position offset: {read key: Allele}} of the events `rows` (indices, ascending; None = all)."""
    ev = self._events if rows is None else self._events[rows]
    table, window, w0 = self._event_ctx
    seq_off = table.read_seq_off
    bases = table.bases
    keys = table.keys
    s0_all = seq_off[ev['read']].astype(np.int64) + ev['read_offset']
    alleles: Dict[int, Dict[str, Allele]] = {}
    for k, (position, read, read_offset, packed) in enumerate(ev.tolist()):
      length_k, type_k, low = packed & 0x0fffffff, (packed >> 28) & 7, packed >> 31
      s0 = int(s0_all[k])
      if type_k == SUBSTITUTION or type_k == REFERENCE:
        text = chr(bases[s0])
      else:
        anchor = self._start + position - w0              # the base the indel is anchored on
        prev = chr(bases[s0 - 1]) if read_offset > 0 else window[anchor:anchor + 1].decode()
        if type_k == DELETION:
          text = prev + window[anchor + 1:anchor + 1 + length_k].decode()
        else:
          text = prev + bytes(bases[s0:s0 + length_k]).decode()
      # a later read with the same key overwrites (read_alleles is a map keyed by ReadKey)
      alleles.setdefault(position, {})[keys[read]] = Allele(text, type_k, 1, bool(low))
    return alleles

  def variant_read_window_counts(self, min_allele_support: int = 0, strict_insertion_filter: bool = False):
    """VariantReadsWindowSelectorCandidates (window_selector.cc:101-141) straight from the events:
    per position of the interval, the number of reads whose non-reference allele covers it
    (substitution: its base; insertion / soft clip of n bases anchored at i: [i + 1 - n, i + 1 + n);
    deletion: [i + 1, i + 1 + n)).  Every read allele counts once whatever it is grouped with, so
    as long as no allele is filtered by its count (min_allele_support <= 1, no strict insertion
    filter) the sum over alleles of count x footprint is the sum over events of their footprint.
    With min_allele_support > 1 the events are grouped into alleles first -- by (position, type,
    text), the text of a substitution being its base, that of the few indel / clip events cut out
    as `_build_alleles` does -- and the events of alleles seen in fewer reads are dropped.
    -> int64[interval_length], or None when the strict insertion filter needs allele totals."""
    if strict_insertion_filter:
      return None
    self._ensure()
    n = self._end - self._start
    ev = self._events
    out = np.zeros(n + 1, np.int64)
    if not len(ev):
      return out[:n]
    packed = ev['length_type']
    length = (packed & 0x0fffffff).astype(np.int64)
    kind = (packed >> 28) & 7
    low = (packed >> 31).astype(bool)
    pos = ev['position'].astype(np.int64)
    # read_alleles is a map keyed by the read's name/number: of two events of one key at one
    # position the later one stands (supplementary alignments share their key)
    table = self._event_ctx[0]
    if len(set(table.keys)) != len(table.keys):
      key_id = np.unique(np.array(table.keys), return_inverse=True)[1][ev['read']]
      composite = pos * (int(key_id.max()) + 1) + key_id
      last = len(ev) - 1 - np.unique(composite[::-1], return_index=True)[1]
      keep = np.zeros(len(ev), bool)
      keep[last] = True
    else:
      keep = np.ones(len(ev), bool)
    keep &= ~low & (kind != REFERENCE)
    if min_allele_support > 1 and keep.any():
      _, window, w0 = self._event_ctx
      bases, seq_off = table.bases, table.read_seq_off
      s0_all = seq_off[ev['read']].astype(np.int64) + ev['read_offset']
      ident = np.zeros(len(ev), np.int64)
      sub = kind == SUBSTITUTION
      ident[sub] = bases[np.minimum(s0_all[sub], max(len(bases) - 1, 0))] if len(bases) else 0
      texts: Dict[bytes, int] = {}
      for k in np.nonzero(keep & ~sub)[0].tolist():
        s0, ln = int(s0_all[k]), int(length[k])
        anchor = self._start + int(pos[k]) - w0
        prev = bytes(bases[s0 - 1:s0]) if int(ev['read_offset'][k]) > 0 else window[anchor:anchor + 1]
        text = prev + (window[anchor + 1:anchor + 1 + ln] if kind[k] == DELETION else bytes(bases[s0:s0 + ln]))
        ident[k] = 256 + texts.setdefault(text, len(texts))
      m = 257 + len(texts)
      key = ((pos * 8 + kind.astype(np.int64)) * m + ident)[keep]
      _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
      kept_rows = np.nonzero(keep)[0]
      keep = np.zeros(len(ev), bool)
      keep[kept_rows[cnt[inv] >= min_allele_support]] = True
    lo = np.where(kind == SUBSTITUTION, pos, np.where(kind == DELETION, pos + 1, pos + 1 - length))
    hi = np.where(kind == SUBSTITUTION, pos + 1, pos + 1 + length)
    lo, hi = np.clip(lo[keep], 0, n), np.clip(hi[keep], 0, n)
    ok = hi > lo
    np.add.at(out, lo[ok], 1)
    np.add.at(out, hi[ok], -1)
    return np.cumsum(out)[:n]


# ---------------------------------------------------------------- NormalizeCigar (host)
_MATCH_OPS = (1, 8, 9)


def _is_match(op) -> bool:
  return op in _MATCH_OPS


def _merge_operations(cigar: List[List[int]], i: int) -> bool:
  """MergeOperations (:561-590) on cigar[i], cigar[i + 1] = [op, length] pairs."""
  a, b = cigar[i], cigar[i + 1]
  if a[0] == b[0] or (_is_match(a[0]) and _is_match(b[0])):
    a[1] += b[1]
    b[1] = 0
  elif a[0] in (2, 3) and b[0] in (2, 3):
    lo, rest = min(a[1], b[1]), max(a[1], b[1]) - min(a[1], b[1])
    if a[1] > b[1]:
      b[0] = a[0]
    a[0], a[1] = 1, lo
    b[1] = rest
  else:
    return False
  return True


def _swipe_and_merge(cigar: List[List[int]]) -> bool:
  """SwipeAndMerge (:706-730)."""
  modified, merged = False, True
  while merged:
    merged = False
    before = len(cigar)
    cigar[:] = [c for c in cigar if c[1] != 0]
    modified |= len(cigar) < before
    for i in range(len(cigar) - 1):
      if _merge_operations(cigar, i):
        merged = True
        break
    modified |= merged
  return modified


def _handle_heading_indel(cigar: List[List[int]], i: int) -> int:
  """HandleHeadingIndel (:630-648)."""
  if not (i == 0 or (cigar and cigar[0][0] == 5 and i == 1)):
    raise ValueError('Check failed: heading indel position')
  if i >= len(cigar):
    return 0
  if cigar[i][0] == 3:
    shift = cigar[i][1]
    del cigar[i]
    return shift
  if cigar[i][0] == 2:
    shift = -cigar[i][1]
    cigar[i][0] = 1
    return shift
  return 0


def _shift_operation(shift: int, i: int, cigar: List[List[int]]) -> int:
  """ShiftOperation (:654-693)."""
  if i == 0 or (cigar and i == 1 and cigar[0][0] == 5):
    return _handle_heading_indel(cigar, i)
  prev = cigar[i - 1]
  if prev[0] == 5:
    raise ValueError('Check failed: soft clip in the middle of a CIGAR')
  if not _is_match(prev[0]):
    return 0
  if shift > prev[1]:
    raise ValueError('Check failed: shift <= prev_op->operation_length()')
  prev[1] -= shift
  if i + 1 == len(cigar):
    cigar.insert(i + 1, [1, shift])
  else:
    cigar[i + 1][1] += shift
  return 0


def normalize_cigar(read_seq: str, interval_offset: int, cigar: Sequence, ref_bases: str):
  """AlleleCounter::NormalizeCigar (:777-845): left-aligns indels against `ref_bases` (the
  counter's reads-interval reference; `interval_offset` = read start in it).
  -> (is_modified, [CigarUnit], read_shift)."""
  norm = [[c.operation, c.operation_length] for c in cigar]
  read_shift = 0
  if not norm:
    return False, [], 0
  modified = False
  for _ in range(100000000):
    read_offset = 0
    cur = interval_offset + read_shift
    prev_len = norm[0][1]
    shifted = False
    for i, (op, n) in enumerate(norm):
      shift = 0
      if op in (2, 3):
        while prev_len > 0 and (
            (op == 3 and read_offset > 0 and cur + n - 1 < len(ref_bases) and
             read_seq[read_offset - 1] == ref_bases[cur + n - 1]) or
            (op == 2 and cur > 0 and read_offset + n - 1 < len(read_seq) and
             read_seq[read_offset + n - 1] == ref_bases[cur - 1])):
          cur -= 1
          prev_len -= 1
          read_offset -= 1
          shift += 1
        if shift > 0:
          read_shift += _shift_operation(shift, i, norm)
          modified = shifted = True
          break
      prev_len = n
      if op in _MATCH_OPS:
        read_offset += n
        cur += n
      elif op in (5, 2):
        read_offset += n
      elif op in (3, 7, 4):
        cur += n
    merged = _swipe_and_merge(norm)
    modified |= merged
    if not shifted and not merged:
      break
  read_shift += _handle_heading_indel(norm, 0)
  return modified, [T.CigarUnit(op, n) for op, n in norm], read_shift
