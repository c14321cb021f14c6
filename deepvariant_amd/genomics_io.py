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
from typing import Dict, Iterator, List, Optional, Tuple

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
             end: int = 1 << 62, use_original_quality_scores: bool = False
             ) -> Tuple[List[str], List[T.Read]]:
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
