"""Minimal BAM (BGZF) and FASTA readers feeding the hot path.

htslib is not available here; zlib is.  Semantics follow the reference's
nucleus readers for the fields the encoder consumes:
  third_party/nucleus/io/sam_reader.cc:734-840  (ConvertToPb): aligned_quality
  is raw phred, read_number = 0 if (unpaired or FREAD1) else 1 (:785),
  fragment_length = isize (:774), info['HP'] from the integer HP aux tag.
  third_party/nucleus/io/reference.cc (IndexedFastaReader, upper-cased unless
  keep_true_case; make_examples_native.cc:128-129 sets keep_true_case=false).

This is "next"-row f1 scaffolding: used by the golden-fixture
generator and by make_examples_native when it is handed file names.
"""
from __future__ import annotations

import gzip
import os
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

from deepvariant_amd import dv_types as T

_SEQ_NT16 = '=ACMGRSVTWYHKDBN'


def _bgzf_blocks(path: str) -> Iterator[bytes]:
  with open(path, 'rb') as f:
    data = f.read()
  pos = 0
  while pos < len(data):
    # gzip header with BC extra subfield carrying the block size.
    xlen = struct.unpack_from('<H', data, pos + 10)[0]
    extra = data[pos + 12:pos + 12 + xlen]
    bsize = None
    p = 0
    while p < xlen:
      si1, si2, slen = extra[p], extra[p + 1], struct.unpack_from('<H', extra, p + 2)[0]
      if si1 == 66 and si2 == 67:
        bsize = struct.unpack_from('<H', extra, p + 4)[0] + 1
      p += 4 + slen
    if bsize is None:
      raise IOError('not a BGZF file: %s' % path)
    cdata = data[pos + 12 + xlen:pos + bsize - 8]
    yield zlib.decompress(cdata, -15)
    pos += bsize


def is_cram(path: str) -> bool:
  with open(path, 'rb') as f:
    return f.read(4) == b'CRAM'


def bam_contig_names(path: str) -> List[str]:
  """The @SQ names of a BAM (or CRAM) header, inflating only the blocks the header spans."""
  if is_cram(path):
    from deepvariant_amd import cram_reader
    return cram_reader.cram_contig_names(path)
  buf = b''
  need = 12
  names: List[str] = []
  for block in _bgzf_blocks(path):
    buf += block
    while True:
      if len(buf) < need:
        break
      if buf[:4] != b'BAM\x01':
        raise IOError('bad BAM magic')
      l_text = struct.unpack_from('<i', buf, 4)[0]
      pos = 8 + l_text
      if len(buf) < pos + 4:
        need = pos + 4
        break
      n_ref = struct.unpack_from('<i', buf, pos)[0]
      pos += 4
      names = []
      complete = True
      for _ in range(n_ref):
        if len(buf) < pos + 4:
          complete = False
          break
        l_name = struct.unpack_from('<i', buf, pos)[0]
        if len(buf) < pos + 4 + l_name + 4:
          complete = False
          break
        names.append(buf[pos + 4:pos + 4 + l_name - 1].decode())
        pos += 4 + l_name + 4
      if complete:
        return names
      need = len(buf) + 1
      break
  raise IOError('truncated BAM header: %s' % path)


def read_bam(path: str, contig: Optional[str] = None, start: int = 0,
             end: int = 1 << 62, use_original_quality_scores: bool = False,
             aux_fields: Sequence[str] = ()) -> Tuple[List[str], List[T.Read]]:
  """Reads a whole (small) BAM; returns (contig names, reads overlapping).  With
  `use_original_quality_scores` the qualities are the OQ:Z tag's characters - 33
  (sam_reader.cc:722-740); a read without the tag is an error here (nucleus leaves its
  aligned_quality empty, which nothing downstream can draw)."""
  buf = b''.join(_bgzf_blocks(path))
  if buf[:4] != b'BAM\x01':
    raise IOError('bad BAM magic')
  l_text = struct.unpack_from('<i', buf, 4)[0]
  pos = 8 + l_text
  n_ref = struct.unpack_from('<i', buf, pos)[0]
  pos += 4
  names = []
  for _ in range(n_ref):
    l_name = struct.unpack_from('<i', buf, pos)[0]
    names.append(buf[pos + 4:pos + 4 + l_name - 1].decode())
    pos += 4 + l_name + 4
  reads: List[T.Read] = []
  n = len(buf)
  while pos < n:
    block_size = struct.unpack_from('<i', buf, pos)[0]
    rec = memoryview(buf)[pos + 4:pos + 4 + block_size]
    pos += 4 + block_size
    (ref_id, rpos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, _next_ref,
     _next_pos, tlen) = struct.unpack_from('<iiBBHHHiiii', rec, 0)
    p = 32
    name = bytes(rec[p:p + l_read_name - 1]).decode()
    p += l_read_name
    cigar = []
    ref_len = 0
    for k in range(n_cigar):
      v = struct.unpack_from('<I', rec, p + 4 * k)[0]
      op, ln = v & 0xF, v >> 4
      cigar.append(T.CigarUnit(T.BAM_OP_TO_NUCLEUS[op], ln))
      if op in (0, 2, 3, 7, 8):
        ref_len += ln
    p += 4 * n_cigar
    seq_bytes = rec[p:p + (l_seq + 1) // 2]
    p += (l_seq + 1) // 2
    qual = bytes(rec[p:p + l_seq])
    p += l_seq
    if (n_cigar == 2 and cigar[0].operation == T.BAM_OP_TO_NUCLEUS[4] and
        cigar[0].operation_length == l_seq and cigar[1].operation == T.BAM_OP_TO_NUCLEUS[3]):
      # more than 65535 operations: the real CIGAR is the CG:B:I tag (SAMv1 4.2.2); htslib
      # swaps it in on read, so nucleus only ever sees the real one
      long_cigar = _find_u32_array_tag(rec[p:], b'CG')
      if long_cigar:
        cigar, ref_len = [], 0
        for v in long_cigar:
          op, ln = v & 0xF, v >> 4
          cigar.append(T.CigarUnit(T.BAM_OP_TO_NUCLEUS[op], ln))
          if op in (0, 2, 3, 7, 8):
            ref_len += ln
    if ref_id < 0 or (flag & 0x4) or (contig is not None and names[ref_id] != contig):
      continue
    if not (end > rpos and start < rpos + max(ref_len, 1)):
      continue
    seq = []
    for i in range(l_seq):
      b = seq_bytes[i >> 1]
      seq.append(_SEQ_NT16[(b >> 4) if (i & 1) == 0 else (b & 0xF)])
    info: Dict[str, T.ListValue] = {}
    aux = rec[p:]
    if use_original_quality_scores:
      oq = _find_string_tag(aux, b'OQ')
      if oq is None:
        raise ValueError('use_original_quality_scores: read %s has no OQ tag' % name)
      if len(oq) != l_seq:
        raise ValueError('OQ tag and sequence are of different length')
      qual = bytes(c - 33 for c in oq)
    hp = _find_int_tag(aux, b'HP')
    if hp is not None:
      info['HP'] = T.ListValue(values=[T.Value(int_value=hp)])
    if aux_fields:      # SamReaderOptions.aux_fields_to_keep (sam_reader.cc:294-470): the listed tags, as info values
      for tag, value in parse_aux_fields(aux).items():
        if tag in aux_fields and tag != 'HP':
          info[tag] = value
    paired = bool(flag & 0x1)
    read_number = 0 if (not paired or (flag & 0x40)) else 1
    reads.append(T.Read(
        fragment_name=name,
        read_number=read_number,
        number_reads=2 if paired else 1,
        proper_placement=bool(flag & 0x2),
        duplicate_fragment=bool(flag & 0x400),
        failed_vendor_quality_checks=bool(flag & 0x200),
        secondary_alignment=bool(flag & 0x100),
        supplementary_alignment=bool(flag & 0x800),
        fragment_length=tlen,
        aligned_sequence=''.join(seq),
        aligned_quality=qual,
        alignment=T.LinearAlignment(
            position=T.Position(reference_name=names[ref_id], position=rpos,
                                reverse_strand=bool(flag & 0x10)),
            mapping_quality=mapq, cigar=cigar),
        info=info))
    if aux_fields:
      reads[-1].base_modifications = parse_base_modifications(reads[-1])      # sam_reader.cc:855-862
    reads[-1]._flag = flag  # pylint: disable=protected-access
    # next_mate_position exists only for a paired read whose mate is mapped with a valid
    # reference id (sam_reader.cc:829-837); without it the read counts as properly placed
    has_mate_pos = paired and not (flag & 0x8) and _next_ref >= 0
    reads[-1]._mate_ok = (not has_mate_pos) or _next_ref == ref_id
  return names, reads


_AUX_SIZES = {b'A': 1, b'c': 1, b'C': 1, b's': 2, b'S': 2, b'i': 4, b'I': 4,
              b'f': 4}
_AUX_FMT = {b'c': '<b', b'C': '<B', b's': '<h', b'S': '<H', b'i': '<i',
            b'I': '<I'}


def parse_aux_fields(aux) -> Dict[str, T.ListValue]:
  """ParseAuxFields (third_party/nucleus/io/sam_reader.cc:294-470) on a BAM record's aux block: every tag as the
  info value nucleus stores -- 'A' and 'Z' strings, integers, floats, B arrays element by element (a B:C array without
  elements is not stored, :394-397); 'H' strings are skipped; a malformed block ends the walk."""
  out: Dict[str, T.ListValue] = {}
  aux = bytes(aux)
  p, n = 0, len(aux)
  while n - p >= 4:
    tag, ty = aux[p:p + 2].decode('latin-1'), aux[p + 2:p + 3]
    p += 3
    if ty == b'A':
      out[tag] = T.ListValue(values=[T.Value(string_value=aux[p:p + 1].decode('latin-1'))])
      p += 1
    elif ty in _AUX_FMT:
      size = _AUX_SIZES[ty]
      if n - p < size:
        break
      out[tag] = T.ListValue(values=[T.Value(int_value=struct.unpack_from(_AUX_FMT[ty], aux, p)[0])])
      p += size
    elif ty == b'f':
      if n - p < 4:
        break
      out[tag] = T.ListValue(values=[T.Value(number_value=struct.unpack_from('<f', aux, p)[0])])
      p += 4
    elif ty in (b'Z', b'H'):
      q = aux.find(b'\0', p)
      if q < 0:
        break
      if ty == b'Z':
        out[tag] = T.ListValue(values=[T.Value(string_value=aux[p:q].decode('latin-1'))])
      p = q + 1
    elif ty == b'B':
      if n - p < 5:
        break
      sub = aux[p:p + 1]
      count = struct.unpack_from('<I', aux, p + 1)[0]
      size = _AUX_SIZES.get(sub, 0) if sub != b'A' else 0
      if not size or n - (p + 5) < count * size:
        break
      p += 5
      if sub == b'f':
        vals = [T.Value(number_value=v) for v in struct.unpack_from('<%df' % count, aux, p)]
      else:
        vals = [T.Value(int_value=v) for v in struct.unpack_from('<%d%s' % (count, _AUX_FMT[sub][1]), aux, p)]
      if not (sub == b'C' and count == 0):
        out[tag] = T.ListValue(values=vals)
      p += count * size
    else:
      break
  return out


_COMPLEMENT = str.maketrans('ACGTacgt', 'TGCAtgca')
_MOD_SPEC = None


def parse_base_modifications(read) -> Dict[str, bytes]:
  """ParseBaseModifications (third_party/nucleus/io/sam_reader.cc:521-719): the read's MM / ML (/ MN) info values ->
  {'5mC': bytes, '6mA': bytes}, one probability per base in the read's aligned orientation.  A second restatement,
  next to the native readers' csrc/aux_planes.h, of the function and of its quirks: unsupported specifications still
  consume their ML values; a specification whose positions run past the read leaves no entry and does not advance
  the ML offset; the two strands of one modification are merged with a SIGNED char maximum; a mismatching MN or an ML
  that is too short drops everything."""
  global _MOD_SPEC
  import re
  if _MOD_SPEC is None:
    _MOD_SPEC = re.compile(r'([ACGTUN])([-+])([a-z]+|[0-9]+)([.?]?)')
  info = read.info
  if 'MM' not in info or 'ML' not in info or not info['MM'].values:
    return {}
  seq = read.aligned_sequence
  mn = info['MN'].values[0].int_value or 0 if 'MN' in info and info['MN'].values else len(seq)
  if mn != len(seq):
    return {}
  reverse = bool(read.alignment.position.reverse_strand)
  if reverse:
    seq = seq[::-1].translate(_COMPLEMENT)
  ml = [v.int_value or 0 for v in info['ML'].values]
  mm = info['MM'].values[0].string_value or ''
  if mm.endswith(';'):
    mm = mm[:-1]
  result: Dict[str, bytes] = {}
  ml_offset = 0
  for mod in mm.split(';'):
    parts = mod.split(',')
    if len(parts) <= 1:
      continue
    m = _MOD_SPEC.fullmatch(parts[0])
    spec = None
    if m:
      base, strand, code = m.group(1), m.group(2), m.group(3)
      if (base, strand, code) == ('C', '+', 'm'):
        spec = T.K5MC
      elif (base, strand, code) in (('A', '+', 'a'), ('T', '-', 'a')):
        spec = T.K6MA
    if spec is None:
      ml_offset += len(parts) - 1
      continue
    deltas = parts[1:]
    plane = bytearray(len(seq))
    idx = base_count = 0
    delta = _stoi(deltas[0])
    for pos, here in enumerate(seq):
      if here != base:
        continue
      if base_count != delta:
        base_count += 1
        continue
      if ml_offset + idx >= len(ml):
        return {}
      plane[pos] = ml[idx + ml_offset] & 0xFF
      base_count = 0
      idx += 1
      if idx >= len(deltas):
        ml_offset += idx
        done = bytes(plane[::-1] if reverse else plane)
        if spec in result:
          signed = lambda b: b - 256 if b > 127 else b      # noqa: E731   (std::max on chars)
          done = bytes(a if signed(a) >= signed(b) else b for a, b in zip(result[spec], done))
        result[spec] = done
        break
      delta = _stoi(deltas[idx])
  return result


def _stoi(text: str) -> int:
  """std::stoi: optional blanks and sign, then digits; what follows is ignored; no digits is an error."""
  import re
  m = re.match(r'\s*([-+]?\d+)', text)
  if not m:
    raise ValueError('MM tag: %r is not a number' % text)
  return int(m.group(1))


def _find_int_tag(aux, tag: bytes) -> Optional[int]:
  p, n = 0, len(aux)
  while p + 3 <= n:
    t = bytes(aux[p:p + 2])
    ty = bytes(aux[p + 2:p + 3])
    p += 3
    if ty in _AUX_SIZES:
      if t == tag and ty in _AUX_FMT:
        return struct.unpack_from(_AUX_FMT[ty], aux, p)[0]
      p += _AUX_SIZES[ty]
    elif ty in (b'Z', b'H'):
      while aux[p] != 0:
        p += 1
      p += 1
    elif ty == b'B':
      sub = bytes(aux[p:p + 1])
      cnt = struct.unpack_from('<i', aux, p + 1)[0]
      p += 5 + cnt * _AUX_SIZES[sub]
    else:
      return None
  return None


def _find_string_tag(aux, tag: bytes) -> Optional[bytes]:
  """The bytes of a `Z` aux tag, or None."""
  p, n = 0, len(aux)
  while p + 3 <= n:
    t = bytes(aux[p:p + 2])
    ty = bytes(aux[p + 2:p + 3])
    p += 3
    if ty in _AUX_SIZES:
      p += _AUX_SIZES[ty]
    elif ty in (b'Z', b'H'):
      q = p
      while aux[q] != 0:
        q += 1
      if t == tag and ty == b'Z':
        return bytes(aux[p:q])
      p = q + 1
    elif ty == b'B':
      sub = bytes(aux[p:p + 1])
      cnt = struct.unpack_from('<i', aux, p + 1)[0]
      p += 5 + cnt * _AUX_SIZES[sub]
    else:
      return None
  return None


def _find_u32_array_tag(aux, tag: bytes) -> Optional[List[int]]:
  """The values of a `B:I` aux tag, or None."""
  p, n = 0, len(aux)
  while p + 3 <= n:
    t = bytes(aux[p:p + 2])
    ty = bytes(aux[p + 2:p + 3])
    p += 3
    if ty in _AUX_SIZES:
      p += _AUX_SIZES[ty]
    elif ty in (b'Z', b'H'):
      while aux[p] != 0:
        p += 1
      p += 1
    elif ty == b'B':
      sub = bytes(aux[p:p + 1])
      cnt = struct.unpack_from('<i', aux, p + 1)[0]
      if t == tag:
        if sub != b'I':
          return None
        return list(struct.unpack_from('<%dI' % cnt, aux, p + 5))
      p += 5 + cnt * _AUX_SIZES[sub]
    else:
      return None
  return None


class FastaReader:
  """FASTA reader (plain, gzip or bgzip); bases upper-cased.  The interface the region chain uses
  of nucleus' IndexedFastaReader (third_party/nucleus/io/reference.h: contig names in file order,
  n_bases, get_bases).

  A plain file with a samtools `.fai` next to it is mapped and read on demand -- a query costs the
  bytes it returns, and a process holds no copy of the genome (R ranks on a GPU share the page
  cache).  Anything else (compressed, no index) is parsed once into memory, record by record with
  bulk byte operations."""

  def __init__(self, path: str):
    self._contigs: Dict[str, str] = {}
    self._index: Dict[str, Tuple[int, int, int, int]] = {}       # name -> (length, offset, line bases, line bytes)
    self._map = None
    with open(path, 'rb') as f:
      compressed = f.read(2) == b'\x1f\x8b'
    fai = path + '.fai'
    if not compressed and os.path.exists(fai):
      import mmap
      with open(fai) as f:
        for line in f:
          parts = line.rstrip('\n').split('\t')
          if len(parts) >= 5:
            self._index[parts[0]] = (int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4]))
      self._file = open(path, 'rb')
      self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
      return
    with (gzip.open if compressed else open)(path, 'rb') as f:
      data = f.read()
    at = data.find(b'>')
    while at >= 0:
      eol = data.find(b'\n', at)
      eol = len(data) if eol < 0 else eol
      nxt = data.find(b'\n>', eol)
      end = len(data) if nxt < 0 else nxt
      header = data[at + 1:eol].split()
      name = header[0].decode() if header else ''
      self._contigs[name] = data[eol + 1:end].translate(None, b'\n\r \t').upper().decode()
      at = -1 if nxt < 0 else nxt + 1

  def n_bases(self, contig: str) -> int:
    if self._map is not None:
      return self._index[contig][0]
    return len(self._contigs[contig])

  def contig_names(self) -> List[str]:
    """Contig names in file order (the order the reference's regions are processed in)."""
    return list(self._index if self._map is not None else self._contigs)

  def get_bases(self, contig: str, start: int, end: int) -> str:
    if self._map is None:
      return self._contigs[contig][start:end]
    length, offset, line_bases, line_bytes = self._index[contig]
    start, end = max(start, 0), min(end, length)
    if end <= start:
      return ''
    first = offset + (start // line_bases) * line_bytes + start % line_bases
    last = offset + (end // line_bases) * line_bytes + end % line_bases
    return self._map[first:last].translate(None, b'\n\r').upper().decode()


def read_satisfies_requirements(read, min_mapping_quality: int = 0,
                                keep_duplicates: bool = False,
                                keep_failed_qc: bool = False,
                                keep_secondary: bool = False,
                                keep_supplementary: bool = False,
                                keep_improperly_placed: bool = False) -> bool:
  """sam_reader_internal::ReadSatisfiesRequirements
  (third_party/nucleus/io/sam_reader.cc:217-247) for aligned reads, with
  IsReadProperlyPlaced (third_party/nucleus/util/utils.cc:261-266)."""
  if read.duplicate_fragment and not keep_duplicates:
    return False
  if read.failed_vendor_quality_checks and not keep_failed_qc:
    return False
  if read.secondary_alignment and not keep_secondary:
    return False
  if read.supplementary_alignment and not keep_supplementary:
    return False
  properly_placed = (read.number_reads < 2 or read.proper_placement or
                     getattr(read, '_mate_ok', True))
  if not properly_placed and not keep_improperly_placed:
    return False
  return read.alignment.mapping_quality >= min_mapping_quality


# ---------------------------------------------------------------- writers (tests, diagnostics)
_NT16_CODE = {c: i for i, c in enumerate(_SEQ_NT16)}


def _bgzf_block(payload: bytes) -> bytes:
  comp = zlib.compressobj(6, zlib.DEFLATED, -15)
  cdata = comp.compress(payload) + comp.flush()
  bsize = len(cdata) + 25
  return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', bsize) + cdata +
          struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload)))


def write_bam(path: str, contigs, reads, sample_name: str = 'sample') -> None:
  """A coordinate-sorted BAM from Read objects (what SamWriter does for the realigner's
  emit_realigned_reads diagnostics, realigner.py:486-497; also the test fixtures' way back to a
  file).  `contigs`: [(name, n_bases)].  Flags are rebuilt from the Read fields this package
  keeps; an integer HP tag is written back."""
  text = '@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % c for c in contigs)
  text += '@RG\tID:rg\tSM:%s\n' % sample_name
  header = b'BAM\x01' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(contigs))
  ref_id = {}
  for i, (name, n) in enumerate(contigs):
    ref_id[name] = i
    header += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', n)
  nucleus_to_bam = {int(v): k for k, v in enumerate(T.BAM_OP_TO_NUCLEUS)}
  records = []
  for r in sorted(reads, key=lambda x: (ref_id[x.alignment.position.reference_name], x.alignment.position.position)):
    aln = r.alignment
    flag = getattr(r, '_flag', None)
    if flag is None:
      flag = ((0x1 if r.number_reads >= 2 else 0) | (0x2 if r.proper_placement else 0) |
              (0x10 if aln.position.reverse_strand else 0) | (0x100 if r.secondary_alignment else 0) |
              (0x200 if r.failed_vendor_quality_checks else 0) | (0x400 if r.duplicate_fragment else 0) |
              (0x800 if r.supplementary_alignment else 0))
      if r.number_reads >= 2:
        flag |= 0x40 if r.read_number == 0 else 0x80
    name = r.fragment_name.encode() + b'\0'
    seq = r.aligned_sequence
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
      packed[i >> 1] |= _NT16_CODE.get(c, 15) << (4 if (i & 1) == 0 else 0)
    cigar = b''.join(struct.pack('<I', (c.operation_length << 4) | nucleus_to_bam[c.operation]) for c in aln.cigar)
    ref_len = sum(c.operation_length for c in aln.cigar if c.operation in (1, 3, 4, 8, 9))
    end = aln.position.position + max(ref_len, 1)
    rid = ref_id[aln.position.reference_name]
    aux = b''
    if 'HP' in r.info and r.info['HP'].values and r.info['HP'].values[0].int_value is not None:
      aux = b'HPi' + struct.pack('<i', r.info['HP'].values[0].int_value)
    # base-modification and flow-space tags, in the types the instruments' files use
    for tag, kind in (('MM', 'Z'), ('ML', 'BC'), ('MN', 'i'), ('tp', 'Bc'), ('t0', 'Z')):
      if tag not in r.info:
        continue
      vals = r.info[tag].values
      if kind == 'Z':
        aux += tag.encode() + b'Z' + (vals[0].string_value or '').encode('latin-1') + b'\0'
      elif kind == 'i':
        aux += tag.encode() + b'i' + struct.pack('<i', int(vals[0].int_value or 0))
      else:
        ints = [int(v.int_value or 0) for v in vals]
        aux += (tag.encode() + b'B' + kind[1].encode() + struct.pack('<I', len(ints)) +
                struct.pack('<%d%s' % (len(ints), kind[1].replace('C', 'B').replace('c', 'b')), *ints))
    aux += getattr(r, '_aux_raw', b'')       # tests: any further BAM-encoded tags, verbatim
    body = (struct.pack('<iiBBHHHiiii', rid, aln.position.position, len(name), aln.mapping_quality,
                        _reg2bin(aln.position.position, end), len(aln.cigar), flag, len(seq), rid if flag & 1 else -1,
                        aln.position.position if flag & 1 else -1, r.fragment_length) +
            name + cigar + bytes(packed) + bytes(bytearray(r.aligned_quality)) + aux)
    records.append(struct.pack('<i', len(body)) + body)
  payload = header + b''.join(records)
  with open(path, 'wb') as f:
    for off in range(0, len(payload), 60000):
      f.write(_bgzf_block(payload[off:off + 60000]))
    f.write(_bgzf_block(b''))         # the BGZF end-of-file marker


def _reg2bin(beg: int, end: int) -> int:
  end -= 1
  for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
    if beg >> shift == end >> shift:
      return base + (beg >> shift)
  return 0


def write_fasta(path: str, contigs, index: bool = False) -> None:
  """contigs: [(name, bases)]; 60 bases a line, gzip if the path ends in .gz; `index` also writes
  the samtools .fai of a plain file (name, length, offset of the first base, 60, 61)."""
  opener = gzip.open if path.endswith('.gz') else open
  fai = []
  at = 0
  with opener(path, 'wb') as f:
    for name, bases in contigs:
      header = ('>%s\n' % name).encode()
      raw = bases.encode() if isinstance(bases, str) else bytes(bases)
      body = b'\n'.join(raw[i:i + 60] for i in range(0, len(raw), 60)) + (b'\n' if raw else b'')
      f.write(header)
      f.write(body)
      fai.append('%s\t%d\t%d\t60\t61\n' % (name, len(raw), at + len(header)))
      at += len(header) + len(body)
  if index and not path.endswith('.gz'):
    with open(path + '.fai', 'w') as f:
      f.writelines(fai)
