"""CRAM 3.0 -> Read objects (host side, pure Python + zlib / bz2 / lzma).

The packed-table path (make_examples' default: packing.ReadTable.from_cram) goes through the NATIVE
decoder, deepvariant_amd/csrc/cram_reader.cpp (dv_cram_read_region); this module is its Python twin:
it yields Read OBJECTS for the object path (phasing, trimmed / alt-aligned pileups), lists contigs, and
tests/test_cram_native_cpu.py holds the two against each other.

The reference opens CRAM through htslib (third_party/nucleus/io/sam_reader.cc:560-640: `hts_open`,
`--use_ref_for_cram`, `hts_set_opt(CRAM_OPT_REFERENCE)`); htslib is not in this image, so this is a
restatement of the published format (CRAM format specification v3.0, samtools/hts-specs) at the
level make_examples needs: the reads of one contig interval with names, flags, positions, CIGARs,
bases, qualities, template lengths and the HP / OQ tags -- what `genomics_io.read_bam` yields for
a BAM, so that everything downstream (read requirements, packing, the region chain) is shared.

What is decoded: the file definition, container and slice headers, the compression header
(preservation map, data-series and tag encodings), blocks compressed with raw / gzip / bzip2 / lzma /
rANS 4x8 (order 0 and 1), the encodings EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA,
SUBEXP and GAMMA, every read feature of the specification, reference-based sequence
reconstruction (external FASTA or a slice's embedded reference), mate links inside a slice (NF) with
htslib's template-length rule (cram/cram_decode.c cram_decode_slice_xref), and the .crai index
for region queries.  Not supported (raise): CRAM 2.x / 3.1 codecs, GOLOMB / GOLOMB_RICE encodings.

Pinned by tests/test_cram_reader_cpu.py: the reference tree's NA12878 slice as CRAM against the same
slice as BAM (every field of every read), and nucleus' own CRAM test files against their SAM text.
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct
import zlib
from typing import Callable, Dict, List, Optional, Tuple

from deepvariant_amd import dv_types as T

_BAM_OP = {'M': 0, 'I': 1, 'D': 2, 'N': 3, 'S': 4, 'H': 5, 'P': 6, '=': 7, 'X': 8}


# ------------------------------------------------------------------------------ integers
def _itf8(b, i: int) -> Tuple[int, int]:
  v = b[i]
  if v < 0x80:
    return v, i + 1
  if v < 0xC0:
    return ((v & 0x3F) << 8) | b[i + 1], i + 2
  if v < 0xE0:
    return ((v & 0x1F) << 16) | (b[i + 1] << 8) | b[i + 2], i + 3
  if v < 0xF0:
    return ((v & 0x0F) << 24) | (b[i + 1] << 16) | (b[i + 2] << 8) | b[i + 3], i + 4
  x = ((v & 0x0F) << 28) | (b[i + 1] << 20) | (b[i + 2] << 12) | (b[i + 3] << 4) | (b[i + 4] & 0x0F)
  return (x - (1 << 32) if x >= 1 << 31 else x), i + 5


def _ltf8(b, i: int) -> Tuple[int, int]:
  v = b[i]
  n = 0
  while n < 8 and (v << n) & 0x80:
    n += 1
  if n == 0:
    return v, i + 1
  if n == 8:
    x = int.from_bytes(b[i + 1:i + 9], 'big')
  else:
    x = v & (0xFF >> (n + 1))
    for k in range(n):
      x = (x << 8) | b[i + 1 + k]
  if x >= 1 << 63:
    x -= 1 << 64
  return x, i + 1 + n


def _itf8_array(b, i: int) -> Tuple[List[int], int]:
  n, i = _itf8(b, i)
  out = []
  for _ in range(n):
    v, i = _itf8(b, i)
    out.append(v)
  return out, i


# ------------------------------------------------------------------------------ rANS 4x8
_RANS_LOW = 1 << 23
_TF_SHIFT = 12
_TF_MASK = (1 << _TF_SHIFT) - 1


def _rans_freqs(b, i: int) -> Tuple[List[int], List[int], bytes, int]:
  """One order-0 frequency table: -> (freq[256], cumulative[256], symbol of every slot, next i)."""
  freq = [0] * 256
  sym = b[i]
  i += 1
  rle = 0
  last = sym
  while True:
    f = b[i]
    i += 1
    if f >= 0x80:
      f = ((f & 0x7F) << 8) | b[i]
      i += 1
    freq[sym] = f
    if rle:
      rle -= 1
      sym += 1
    else:
      sym = b[i]
      i += 1
      if sym == last + 1:
        rle = b[i]
        i += 1
    last = sym
    if sym == 0:
      break
  cum = [0] * 256
  slots = bytearray(1 << _TF_SHIFT)
  x = 0
  for s in range(256):
    cum[s] = x
    f = freq[s]
    if f:
      slots[x:x + f] = bytes([s]) * f
      x += f
  return freq, cum, bytes(slots), i


def _rans_decode(data: bytes) -> bytes:
  order = data[0]
  out_size = struct.unpack_from('<I', data, 5)[0]
  i = 9
  if out_size == 0:
    return b''
  out = bytearray(out_size)
  if order == 0:
    freq, cum, slots, i = _rans_freqs(data, i)
    r = list(struct.unpack_from('<4I', data, i))
    i += 16
    n = len(data)
    for k in range(out_size):
      j = k & 3
      x = r[j]
      m = x & _TF_MASK
      s = slots[m]
      out[k] = s
      x = freq[s] * (x >> _TF_SHIFT) + m - cum[s]
      while x < _RANS_LOW and i < n:
        x = (x << 8) | data[i]
        i += 1
      r[j] = x
    return bytes(out)
  # order 1: one table per context symbol
  tables: Dict[int, Tuple[List[int], List[int], bytes]] = {}
  ctx = data[i]
  i += 1
  rle = 0
  last = ctx
  while True:
    f, c, s, i = _rans_freqs(data, i)
    tables[ctx] = (f, c, s)
    if rle:
      rle -= 1
      ctx += 1
    else:
      ctx = data[i]
      i += 1
      if ctx == last + 1:
        rle = data[i]
        i += 1
    last = ctx
    if ctx == 0:
      break
  r = list(struct.unpack_from('<4I', data, i))
  i += 16
  n = len(data)
  q = out_size >> 2
  pos = [0, q, 2 * q, 3 * q]
  prev = [0, 0, 0, 0]
  empty = ([0] * 256, [0] * 256, bytes(1 << _TF_SHIFT))
  for _ in range(q):
    for j in range(4):
      f, c, slots = tables.get(prev[j], empty)
      x = r[j]
      m = x & _TF_MASK
      s = slots[m]
      out[pos[j]] = s
      pos[j] += 1
      x = f[s] * (x >> _TF_SHIFT) + m - c[s]
      while x < _RANS_LOW and i < n:
        x = (x << 8) | data[i]
        i += 1
      r[j] = x
      prev[j] = s
  k = pos[3]
  while k < out_size:     # the remainder belongs to the fourth stream
    f, c, slots = tables.get(prev[3], empty)
    x = r[3]
    m = x & _TF_MASK
    s = slots[m]
    out[k] = s
    k += 1
    x = f[s] * (x >> _TF_SHIFT) + m - c[s]
    while x < _RANS_LOW and i < n:
      x = (x << 8) | data[i]
      i += 1
    r[3] = x
    prev[3] = s
  return bytes(out)


def _decompress(method: int, data: bytes, raw_size: int) -> bytes:
  if method == 0:
    return bytes(data)
  if method == 1:
    return zlib.decompress(data, 15 + 32)
  if method == 2:
    return bz2.decompress(data)
  if method == 3:
    return lzma.decompress(data)
  if method == 4:
    return _rans_decode(bytes(data))
  raise ValueError('CRAM block compression method %d is not supported' % method)


# ------------------------------------------------------------------------------ blocks
class _Block:
  __slots__ = ('content_type', 'content_id', 'data', 'pos')

  def __init__(self, content_type: int, content_id: int, data: bytes):
    self.content_type, self.content_id, self.data, self.pos = content_type, content_id, data, 0


def _read_block(b, i: int) -> Tuple[_Block, int]:
  method, ctype = b[i], b[i + 1]
  i += 2
  cid, i = _itf8(b, i)
  csize, i = _itf8(b, i)
  rsize, i = _itf8(b, i)
  data = _decompress(method, b[i:i + csize], rsize)
  if len(data) != rsize:
    raise IOError('CRAM block inflates to %d bytes, header says %d' % (len(data), rsize))
  return _Block(ctype, cid, data), i + csize + 4      # + CRC32


class _Bits:
  """The core data block: bits, most significant first."""

  def __init__(self, data: bytes):
    self.data, self.pos = data, 0     # pos in bits

  def read(self, n: int) -> int:
    v = 0
    for _ in range(n):
      byte = self.data[self.pos >> 3]
      v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
      self.pos += 1
    return v


# ------------------------------------------------------------------------------ encodings
def _parse_encoding(b, i: int):
  codec, i = _itf8(b, i)
  n, i = _itf8(b, i)
  return (codec, bytes(b[i:i + n])), i + n


class _Codecs:
  """Readers for the data series of one slice (its external blocks + core bit stream)."""

  def __init__(self, external: Dict[int, _Block], core: _Bits):
    self.external, self.core = external, core

  def int_reader(self, enc) -> Callable[[], int]:
    codec, p = enc
    if codec == 1:       # EXTERNAL: ITF8 values
      cid, _ = _itf8(p, 0)
      blk = self.external.get(cid)

      def read_ext():
        v, blk.pos = _itf8(blk.data, blk.pos)
        return v
      return read_ext
    if codec == 3:       # HUFFMAN
      alphabet, i = _itf8_array(p, 0)
      lengths, i = _itf8_array(p, i)
      if len(alphabet) == 1 and lengths[0] == 0:
        only = alphabet[0]
        return lambda: only
      # canonical code: symbols sorted by (length, value)
      order = sorted(range(len(alphabet)), key=lambda k: (lengths[k], alphabet[k]))
      codes = {}
      code, prev_len = 0, lengths[order[0]]
      for k in order:
        code <<= lengths[k] - prev_len
        prev_len = lengths[k]
        codes[(lengths[k], code)] = alphabet[k]
        code += 1
      core = self.core

      def read_huff():
        v, n = 0, 0
        while True:
          v = (v << 1) | core.read(1)
          n += 1
          if (n, v) in codes:
            return codes[(n, v)]
          if n > 32:
            raise IOError('bad Huffman code in CRAM core block')
      return read_huff
    if codec == 6:       # BETA
      offset, i = _itf8(p, 0)
      nbits, i = _itf8(p, i)
      core = self.core
      return lambda: core.read(nbits) - offset
    if codec == 7:       # SUBEXP
      offset, i = _itf8(p, 0)
      k, i = _itf8(p, i)
      core = self.core

      def read_subexp():
        n = 0
        while core.read(1):
          n += 1
        if n == 0:
          return core.read(k) - offset
        bits = n + k - 1
        return ((1 << bits) | core.read(bits)) - offset
      return read_subexp
    if codec == 9:       # GAMMA
      offset, _ = _itf8(p, 0)
      core = self.core

      def read_gamma():
        n = 0
        while core.read(1) == 0:
          n += 1
        return ((1 << n) | core.read(n)) - offset
      return read_gamma
    if codec == 0:
      return lambda: 0
    raise ValueError('CRAM integer encoding %d is not supported' % codec)

  def byte_reader(self, enc) -> Callable[[], int]:
    codec, p = enc
    if codec == 1:       # EXTERNAL: raw bytes
      cid, _ = _itf8(p, 0)
      blk = self.external.get(cid)

      def read_byte():
        v = blk.data[blk.pos]
        blk.pos += 1
        return v
      return read_byte
    return self.int_reader(enc)

  def bytes_reader(self, enc) -> Callable[[], bytes]:
    codec, p = enc
    if codec == 5:       # BYTE_ARRAY_STOP
      stop = p[0]
      cid, _ = _itf8(p, 1)
      blk = self.external.get(cid)

      def read_stop():
        end = blk.data.index(stop, blk.pos)
        v = blk.data[blk.pos:end]
        blk.pos = end + 1
        return v
      return read_stop
    if codec == 4:       # BYTE_ARRAY_LEN
      len_enc, i = _parse_encoding(p, 0)
      val_enc, i = _parse_encoding(p, i)
      read_len = self.int_reader(len_enc)
      if val_enc[0] == 1:
        cid, _ = _itf8(val_enc[1], 0)
        blk = self.external.get(cid)

        def read_len_ext():
          n = read_len()
          v = blk.data[blk.pos:blk.pos + n]
          blk.pos += n
          return v
        return read_len_ext
      read_b = self.byte_reader(val_enc)
      return lambda: bytes(read_b() for _ in range(read_len()))
    raise ValueError('CRAM byte-array encoding %d is not supported' % codec)


# ------------------------------------------------------------------------------ headers
class _CompressionHeader:
  def __init__(self, data: bytes):
    i = 0
    _, i = _itf8(data, i)          # map size in bytes
    n, i = _itf8(data, i)
    self.read_names, self.ap_delta, self.ref_required = True, True, True
    self.subst = bytes([0x1B] * 5)
    self.tag_lists: List[List[bytes]] = [[]]
    for _ in range(n):
      key = bytes(data[i:i + 2])
      i += 2
      if key in (b'RN', b'AP', b'RR'):
        v = bool(data[i])
        i += 1
        if key == b'RN':
          self.read_names = v
        elif key == b'AP':
          self.ap_delta = v
        else:
          self.ref_required = v
      elif key == b'SM':
        self.subst = bytes(data[i:i + 5])
        i += 5
      elif key == b'TD':
        ln, i = _itf8(data, i)
        td = bytes(data[i:i + ln])
        i += ln
        self.tag_lists = []
        for entry in td.split(b'\0')[:-1] if td.endswith(b'\0') else td.split(b'\0'):
          self.tag_lists.append([entry[k:k + 3] for k in range(0, len(entry), 3)])
      else:
        raise ValueError('unknown CRAM preservation key %r' % key)
    _, i = _itf8(data, i)
    n, i = _itf8(data, i)
    self.series: Dict[bytes, Tuple[int, bytes]] = {}
    for _ in range(n):
      key = bytes(data[i:i + 2])
      i += 2
      self.series[key], i = _parse_encoding(data, i)
    _, i = _itf8(data, i)
    n, i = _itf8(data, i)
    self.tags: Dict[int, Tuple[int, bytes]] = {}
    for _ in range(n):
      key, i = _itf8(data, i)
      self.tags[key], i = _parse_encoding(data, i)
    # substitution matrix: for reference base R (A C G T N) the code (0..3) of each other base
    self.subst_lookup = {}
    bases = 'ACGTN'
    for r, ref in enumerate(bases):
      others = [x for x in bases if x != ref]
      byte = self.subst[r]
      for k, alt in enumerate(others):
        code = (byte >> (6 - 2 * k)) & 3
        self.subst_lookup[(ref, code)] = alt


def _aux_value(kind: int, raw: bytes):
  """The value bytes of a tag as CRAM stores them (BAM layout without the 3-byte key)."""
  c = chr(kind)
  if c == 'A':
    return chr(raw[0])
  if c in 'cCsSiI':
    return struct.unpack('<' + {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I'}[c], raw[:struct.calcsize(
        {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I'}[c])])[0]
  if c == 'f':
    return struct.unpack('<f', raw[:4])[0]
  if c in 'ZH':
    return raw.rstrip(b'\0').decode()
  if c == 'B':
    sub = chr(raw[0])
    n = struct.unpack_from('<I', raw, 1)[0]
    fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[sub]
    return list(struct.unpack_from('<%d%s' % (n, fmt), raw, 5))
  raise ValueError('unknown tag type %r' % c)


class CramRecord:
  __slots__ = ('flag', 'cram_flags', 'ref_id', 'read_length', 'pos', 'read_group', 'name', 'mate_flags',
               'mate_ref_id', 'mate_pos', 'tlen', 'next_fragment', 'tags', 'mapq', 'seq', 'qual', 'cigar', 'ref_len',
               'mate_line')


# ------------------------------------------------------------------------------ the reader
class CramFile:
  """Containers of one CRAM file; `reads(contig, start, end)` decodes the slices that overlap."""

  def __init__(self, path: str, fetch_reference: Optional[Callable[[str, int, int], str]] = None):
    self.path = path
    # the file is mapped, not read: a whole-genome CRAM is tens of gigabytes, a query touches the
    # header container and the containers its index (or a scan of the container headers) selects
    import mmap
    self._file = open(path, 'rb')
    self.buf = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
    b = self.buf
    if b[:4] != b'CRAM':
      raise IOError('Failed to parse BAM/CRAM file. %s: bad CRAM magic' % path)
    if b[4] != 3 or b[5] != 0:
      raise ValueError('CRAM version %d.%d is not supported (3.0 is)' % (b[4], b[5]))
    self.fetch_reference = fetch_reference
    # the first container holds the SAM header
    _, first_blocks, nxt = self._container(26)
    blk, _ = _read_block(b, first_blocks)
    n = struct.unpack_from('<i', blk.data, 0)[0]
    self.header_text = blk.data[4:4 + n].decode()
    self.contig_names: List[str] = []
    self.contig_lengths: List[int] = []
    for line in self.header_text.split('\n'):
      if line.startswith('@SQ'):
        fields = dict(f.split(':', 1) for f in line.split('\t')[1:] if ':' in f)
        self.contig_names.append(fields['SN'])
        self.contig_lengths.append(int(fields.get('LN', 0)))
    self.first_data_container = nxt

  def _container(self, i: int):
    """-> (header dict, offset of the first block, offset of the next container)."""
    b = self.buf
    length = struct.unpack_from('<i', b, i)[0]
    j = i + 4
    h = {}
    h['ref_id'], j = _itf8(b, j)
    h['start'], j = _itf8(b, j)
    h['span'], j = _itf8(b, j)
    h['n_records'], j = _itf8(b, j)
    h['record_counter'], j = _ltf8(b, j)
    h['bases'], j = _ltf8(b, j)
    h['n_blocks'], j = _itf8(b, j)
    h['landmarks'], j = _itf8_array(b, j)
    j += 4     # CRC32
    return h, j, j + length

  def containers(self):
    i = self.first_data_container
    n = len(self.buf)
    while i < n:
      h, blocks, nxt = self._container(i)
      if h['n_blocks'] > 0 and not (h['ref_id'] == -1 and h['n_records'] == 0 and h['start'] == 4542278):   # EOF marker
        yield h, blocks
      i = nxt

  def _index(self):
    """<path>.crai (gzip text: reference id, start, span, container offset, slice offset, slice
    size per line; CRAM format specification, section 12) -> [(ref_id, start, span, container offset)],
    or None when there is no index next to the file."""
    if not hasattr(self, '_crai'):
      self._crai = None
      import os
      for cand in (self.path + '.crai', self.path[:-5] + '.crai' if self.path.endswith('.cram') else None):
        if cand and os.path.exists(cand):
          rows = []
          with gzip.open(cand, 'rt') as f:
            for line in f:
              parts = line.split('\t')
              if len(parts) >= 4:
                rows.append((int(parts[0]), int(parts[1]), int(parts[2]), int(parts[3])))
          self._crai = rows
          break
    return self._crai

  def containers_for(self, want_ref: int, start: int, end: int):
    """The containers that can hold reads of reference `want_ref` overlapping [start, end): by the
    .crai index when there is one (its offsets point at container headers), else by walking the
    container headers."""
    index = self._index()
    if index is None:
      for h, blocks in self.containers():
        # a single-reference container of another contig or interval, or the unmapped tail (-1), cannot hold
        # a read of the query; multi-reference containers (-2) are looked into
        if h['ref_id'] == -1:
          continue
        if h['ref_id'] >= 0 and (h['ref_id'] != want_ref or h['start'] - 1 >= end or h['start'] - 1 + h['span'] <= start):
          continue
        yield h, blocks
      return
    seen = set()
    for ref_id, s0, span, offset in index:      # (a multi-reference slice has one line per reference)
      if ref_id != want_ref or s0 - 1 >= end or s0 - 1 + span <= start:
        continue
      if offset in seen:
        continue
      seen.add(offset)
      h, blocks, _ = self._container(offset)
      if h['n_blocks'] > 0:
        yield h, blocks

  # ---- one slice
  def _decode_slice(self, ch: _CompressionHeader, i: int, want_ref: Optional[int], lo: int, hi: int) -> List[CramRecord]:
    b = self.buf
    hdr, i = _read_block(b, i)
    d = hdr.data
    k = 0
    s_ref, k = _itf8(d, k)
    s_start, k = _itf8(d, k)
    s_span, k = _itf8(d, k)
    n_records, k = _itf8(d, k)
    record_counter, k = _ltf8(d, k)      # index of the slice's first record in the file
    n_blocks, k = _itf8(d, k)
    _, k = _itf8_array(d, k)
    embedded_id, k = _itf8(d, k)
    ref_md5 = bytes(d[k:k + 16])
    # a slice that cannot overlap the query is skipped BEFORE its blocks are inflated (the rANS decoder is
    # pure Python: decoding and then dropping every non-overlapping slice of a container dominated region queries)
    if want_ref is not None and s_ref >= 0 and (s_ref != want_ref or s_start - 1 >= hi or s_start - 1 + s_span <= lo):
      return []
    external: Dict[int, _Block] = {}
    core = _Bits(b'')
    for _ in range(n_blocks):
      blk, i = _read_block(b, i)
      if blk.content_type == 5:
        core = _Bits(blk.data)
      elif blk.content_type == 4:
        external[blk.content_id] = blk
    codecs = _Codecs(external, core)
    S = ch.series

    def ints(key):
      return codecs.int_reader(S[key]) if key in S else (lambda: 0)

    def byts(key):
      return codecs.byte_reader(S[key]) if key in S else (lambda: 0)

    def arrs(key):
      return codecs.bytes_reader(S[key]) if key in S else (lambda: b'')
    rd = {k2: ints(k2) for k2 in (b'BF', b'CF', b'RI', b'RL', b'AP', b'RG', b'MF', b'NS', b'NP', b'TS', b'NF', b'TL',
                                  b'FN', b'FP', b'DL', b'RS', b'PD', b'HC', b'MQ')}
    rb = {k2: byts(k2) for k2 in (b'FC', b'BS', b'BA', b'QS')}
    ra = {k2: arrs(k2) for k2 in (b'RN', b'IN', b'SC', b'BB', b'QQ')}
    tag_readers: Dict[int, Callable[[], bytes]] = {}
    # reference of the slice
    ref_cache: Dict[int, Tuple[int, str]] = {}
    if embedded_id >= 0 and embedded_id in external:
      ref_cache[s_ref] = (s_start - 1, external[embedded_id].data.decode('latin-1'))

    def ref_bases(ref_id: int, start: int, n: int) -> str:
      if n <= 0:
        return ''
      if ref_id in ref_cache:
        o, text = ref_cache[ref_id]
        if start >= o and start + n <= o + len(text):
          return text[start - o:start - o + n]
      if self.fetch_reference is None:
        raise ValueError('Failed to parse BAM/CRAM file. %s needs a reference (--ref) to be decoded' % self.path)
      name = self.contig_names[ref_id]
      if s_ref >= 0 and ref_id == s_ref:
        o = max(0, s_start - 1)
        text = self.fetch_reference(name, o, o + s_span + 1)
        # the slice header carries the MD5 of the reference stretch it was encoded against (CRAM 3.0
        # section 8.5; all zero = not recorded): htslib refuses a mismatching FASTA, and decoding
        # against the wrong one would silently produce wrong read bases
        if ref_md5 != bytes(16) and len(ref_md5) == 16 and s_span > 0 and len(text) >= s_span:
          import hashlib
          if hashlib.md5(text[:s_span].upper().encode('latin-1')).digest() != ref_md5:
            raise ValueError('Failed to parse BAM/CRAM file. %s: the reference MD5 of the slice at %s:%d does not '
                             'match --ref' % (self.path, name, s_start))
        if start >= o and start + n <= o + len(text):
          ref_cache[ref_id] = (o, text)
          return text[start - o:start - o + n]
      return self.fetch_reference(name, start, start + n)

    recs: List[CramRecord] = []
    prev_pos = s_start
    for _ in range(n_records):
      r = CramRecord()
      r.flag = rd[b'BF']()
      r.cram_flags = rd[b'CF']()
      r.ref_id = rd[b'RI']() if s_ref == -2 else s_ref
      r.read_length = rd[b'RL']()
      ap = rd[b'AP']()
      if ch.ap_delta:
        prev_pos += ap
        r.pos = prev_pos
      else:
        r.pos = ap
      r.read_group = rd[b'RG']()
      r.name = ra[b'RN']() if ch.read_names else b''
      r.mate_flags, r.mate_ref_id, r.mate_pos, r.tlen, r.next_fragment = 0, -1, 0, None, -1
      if r.cram_flags & 0x2:       # detached: mate information stored verbatim
        r.mate_flags = rd[b'MF']()
        if not ch.read_names:
          r.name = ra[b'RN']()
        r.mate_ref_id = rd[b'NS']()
        r.mate_pos = rd[b'NP']()
        r.tlen = rd[b'TS']()
        if r.mate_flags & 0x1:
          r.flag |= 0x20
        if r.mate_flags & 0x2:
          r.flag |= 0x8
      elif r.cram_flags & 0x4:     # mate is a later record of this slice
        r.next_fragment = rd[b'NF']()
      tl = rd[b'TL']()
      r.tags = {}
      for key3 in ch.tag_lists[tl]:
        tid = (key3[0] << 16) | (key3[1] << 8) | key3[2]
        reader = tag_readers.get(tid)
        if reader is None:
          reader = tag_readers[tid] = codecs.bytes_reader(ch.tags[tid])
        raw = reader()
        name2 = key3[:2]
        if name2 in (b'HP', b'OQ', b'CG'):
          r.tags[name2.decode()] = _aux_value(key3[2], raw)
      L = r.read_length
      qual = bytearray(b'\xff' * L)
      if not r.flag & 0x4:
        seq = [''] * L
        cigar: List[Tuple[str, int]] = []

        def push(op, n):
          if n <= 0:
            return
          if cigar and cigar[-1][0] == op:
            cigar[-1] = (op, cigar[-1][1] + n)
          else:
            cigar.append((op, n))
        n_feat = rd[b'FN']()
        rp = 1                    # next read base (1-based) not yet filled
        refp = r.pos - 1          # 0-based reference position of that base
        fpos = 0

        def fill_matches(upto):   # read bases rp .. upto-1 equal the reference
          nonlocal rp, refp
          n = upto - rp
          if n > 0:
            text = ref_bases(r.ref_id, refp, n)
            if len(text) < n:
              text = text + 'N' * (n - len(text))
            seq[rp - 1:upto - 1] = list(text.upper())
            push('M', n)
            rp += n
            refp += n
        for _f in range(n_feat):
          code = chr(rb[b'FC']())
          fpos += rd[b'FP']()
          if code in 'qQ':
            if code == 'Q':
              qual[fpos - 1] = rb[b'QS']()
            else:
              q = ra[b'QQ']()
              qual[fpos - 1:fpos - 1 + len(q)] = q
            continue
          fill_matches(fpos)
          if code == 'B':
            seq[fpos - 1] = chr(rb[b'BA']())
            qual[fpos - 1] = rb[b'QS']()
            push('M', 1)
            rp += 1
            refp += 1
          elif code == 'X':
            c = rb[b'BS']()
            ref_base = ref_bases(r.ref_id, refp, 1).upper() or 'N'
            if ref_base not in 'ACGT':
              ref_base = 'N'
            seq[fpos - 1] = ch.subst_lookup[(ref_base, c)]
            push('M', 1)
            rp += 1
            refp += 1
          elif code == 'I':
            ins = ra[b'IN']()
            seq[fpos - 1:fpos - 1 + len(ins)] = list(ins.decode('latin-1'))
            push('I', len(ins))
            rp += len(ins)
          elif code == 'i':
            seq[fpos - 1] = chr(rb[b'BA']())
            push('I', 1)
            rp += 1
          elif code == 'S':
            clip = ra[b'SC']()
            seq[fpos - 1:fpos - 1 + len(clip)] = list(clip.decode('latin-1'))
            push('S', len(clip))
            rp += len(clip)
          elif code == 'D':
            n = rd[b'DL']()
            push('D', n)
            refp += n
          elif code == 'N':
            n = rd[b'RS']()
            push('N', n)
            refp += n
          elif code == 'H':
            push('H', rd[b'HC']())
          elif code == 'P':
            push('P', rd[b'PD']())
          elif code == 'b':
            bb = ra[b'BB']()
            seq[fpos - 1:fpos - 1 + len(bb)] = list(bb.decode('latin-1'))
            push('M', len(bb))
            rp += len(bb)
            refp += len(bb)
          else:
            raise ValueError('unknown CRAM read feature %r' % code)
        fill_matches(L + 1)
        r.mapq = rd[b'MQ']()
        if r.cram_flags & 0x1:
          for q in range(L):
            qual[q] = rb[b'QS']()
        r.seq = ''.join(seq)
        r.cigar = cigar
        r.ref_len = refp - (r.pos - 1)
      else:
        r.seq = ''.join(chr(rb[b'BA']()) for _b in range(L))
        if r.cram_flags & 0x1:
          for q in range(L):
            qual[q] = rb[b'QS']()
        r.mapq, r.cigar, r.ref_len = 0, [], 0
      r.qual = bytes(qual)
      r.mate_line = -1
      recs.append(r)
    # mates inside the slice: names, mate fields and htslib's template length
    for idx, r in enumerate(recs):
      if r.next_fragment >= 0:
        j = idx + r.next_fragment + 1
        if j < len(recs):
          r.mate_line = j
    for idx, r in enumerate(recs):
      if r.mate_line >= 0 and r.tlen is None:
        chain = [idx]
        j = r.mate_line
        while j >= 0 and j not in chain:
          chain.append(j)
          j = recs[j].mate_line
        same_ref = all(recs[c].ref_id == r.ref_id for c in chain)
        aleft = min(recs[c].pos for c in chain)
        aright = max(recs[c].pos + max(recs[c].ref_len, 1) - 1 for c in chain) if chain else r.pos
        left_cnt = sum(1 for c in chain if recs[c].pos == aleft)
        tlen = aright - aleft + 1 if same_ref else 0
        for c in chain:
          rc = recs[c]
          if not same_ref:
            rc.tlen = 0
          elif rc.pos == aleft and (left_cnt == 1 or rc.flag & 0x40):
            rc.tlen = tlen
          else:
            rc.tlen = -tlen
        # lossy names (RN = false): one generated name for the whole template -- htslib gives both mates
        # of a pair the same generated name, and allele_support matches reads by name + read number
        if not recs[chain[0]].name:
          recs[chain[0]].name = b'%d' % (record_counter + chain[0])
        for a, c in enumerate(chain):      # mate of chain[a] is the next in the chain, the last one's is the first
          m = recs[chain[(a + 1) % len(chain)]]
          rc = recs[c]
          rc.mate_ref_id, rc.mate_pos = m.ref_id, m.pos
          if m.flag & 0x10:
            rc.flag |= 0x20
          if m.flag & 0x4:
            rc.flag |= 0x8
          if not rc.name:
            rc.name = recs[chain[0]].name
    for idx, r in enumerate(recs):
      if r.tlen is None:
        r.tlen = 0
      if not r.name:      # generated from the record's index in the FILE, so names differ across slices
        r.name = b'%d' % (record_counter + idx)
    return recs

  def records(self, contig: Optional[str] = None, start: int = 0, end: int = 1 << 62) -> List[CramRecord]:
    want = self.contig_names.index(contig) if contig is not None else None
    out: List[CramRecord] = []
    for h, blocks in (self.containers() if want is None else self.containers_for(want, start, end)):
      first, _ = _read_block(self.buf, blocks)
      if first.content_type != 1:
        continue
      ch = _CompressionHeader(first.data)
      for lm in h['landmarks']:
        for r in self._decode_slice(ch, blocks + lm, want, start, end):
          if r.flag & 0x4 or r.ref_id < 0:
            continue
          if want is not None and r.ref_id != want:
            continue
          p0 = r.pos - 1
          if not (end > p0 and start < p0 + max(r.ref_len, 1)):
            continue
          out.append(r)
    return out


def cram_contig_names(path: str) -> List[str]:
  return CramFile(path).contig_names


def read_cram(path: str, fetch_reference: Optional[Callable[[str, int, int], str]], contig: Optional[str] = None,
              start: int = 0, end: int = 1 << 62, use_original_quality_scores: bool = False
              ) -> Tuple[List[str], List[T.Read]]:
  """(contig names, reads overlapping [start, end) of `contig`) -- genomics_io.read_bam for a CRAM.
  `fetch_reference(name, start, end) -> bases`: the FASTA the file was written against
  (--ref with --use_ref_for_cram, the reference's default)."""
  f = CramFile(path, fetch_reference)
  reads: List[T.Read] = []
  for r in f.records(contig, start, end):
    flag = r.flag
    qual = r.qual
    if use_original_quality_scores:
      oq = r.tags.get('OQ')
      if oq is None:
        raise ValueError('use_original_quality_scores: read %s has no OQ tag' % r.name.decode())
      if len(oq) != r.read_length:
        raise ValueError('OQ tag and sequence are of different length')
      qual = bytes(ord(c) - 33 for c in oq)
    info: Dict[str, T.ListValue] = {}
    if 'HP' in r.tags:
      info['HP'] = T.ListValue(values=[T.Value(int_value=int(r.tags['HP']))])
    paired = bool(flag & 0x1)
    read = T.Read(
        fragment_name=r.name.decode(), read_number=0 if (not paired or (flag & 0x40)) else 1,
        number_reads=2 if paired else 1, proper_placement=bool(flag & 0x2), duplicate_fragment=bool(flag & 0x400),
        failed_vendor_quality_checks=bool(flag & 0x200), secondary_alignment=bool(flag & 0x100),
        supplementary_alignment=bool(flag & 0x800), fragment_length=int(r.tlen), aligned_sequence=r.seq,
        aligned_quality=qual,
        alignment=T.LinearAlignment(
            position=T.Position(reference_name=f.contig_names[r.ref_id], position=r.pos - 1,
                                reverse_strand=bool(flag & 0x10)),
            mapping_quality=r.mapq,
            cigar=[T.CigarUnit(T.BAM_OP_TO_NUCLEUS[_BAM_OP[op]], n) for op, n in r.cigar]),
        info=info)
    read._flag = flag   # pylint: disable=protected-access
    has_mate_pos = paired and not (flag & 0x8) and r.mate_ref_id >= 0
    read._mate_ok = (not has_mate_pos) or r.mate_ref_id == r.ref_id   # pylint: disable=protected-access
    reads.append(read)
  return f.contig_names, reads
