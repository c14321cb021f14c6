"""A small CRAM 3.0 WRITER for the tests (test infrastructure only; nothing in the product imports it).

The real CRAM files of the reference tree use a handful of the format's options (EXTERNAL / HUFFMAN
series, gzip + rANS blocks, stored names, attached mates).  This writer produces files that take the
other paths of the specification (samtools/hts-specs CRAMv3): every integer / byte / byte-array
encoding (EXTERNAL, HUFFMAN with mixed code lengths, BETA, SUBEXP, GAMMA, BYTE_ARRAY_LEN,
BYTE_ARRAY_STOP), every block codec (raw, gzip, bzip2, lzma, rANS order 0 / 1), every read feature,
lossy names, absolute positions, detached and attached mates, multi-reference slices, embedded
references, several slices per container, a .crai index.  tests/test_cram_native_cpu.py decodes what
it writes with the native decoder (deepvariant_amd/csrc/cram_reader.cpp) and with the Python one
(deepvariant_amd/cram_reader.py) and compares both with the reads that went in.
"""
import bz2
import gzip
import hashlib
import heapq
import lzma
import struct
import zlib
from typing import Dict, List, Optional, Tuple


# ------------------------------------------------------------------------------ integers
def itf8(v: int) -> bytes:
  v &= 0xFFFFFFFF
  if v < 0x80:
    return bytes([v])
  if v < 0x4000:
    return bytes([0x80 | (v >> 8), v & 0xFF])
  if v < 0x200000:
    return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
  if v < 0x10000000:
    return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
  return bytes([0xF0 | ((v >> 28) & 0x0F), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v: int) -> bytes:
  assert 0 <= v < 1 << 56
  if v < 0x80:
    return bytes([v])
  for n in range(1, 8):      # n extra bytes, n leading ones
    if v < 1 << (7 - n + 8 * n):
      head = ((0xFF << (8 - n)) & 0xFF) | (v >> (8 * n))
      return bytes([head]) + (v & ((1 << (8 * n)) - 1)).to_bytes(n, 'big')
  raise AssertionError


def itf8_array(values) -> bytes:
  return itf8(len(values)) + b''.join(itf8(v) for v in values)


class BitWriter:
  def __init__(self):
    self.bits: List[int] = []

  def write(self, value: int, n: int):
    for k in range(n - 1, -1, -1):
      self.bits.append((value >> k) & 1)

  def data(self) -> bytes:
    bits = self.bits + [0] * (-len(self.bits) % 8)
    return bytes(int(''.join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


# ------------------------------------------------------------------------------ block codecs
def compress(method: str, data: bytes, rans_encode=None) -> Tuple[int, bytes]:
  if method == 'raw':
    return 0, data
  if method == 'gzip':
    return 1, gzip.compress(data)
  if method == 'bzip2':
    return 2, bz2.compress(data)
  if method == 'lzma':
    return 3, lzma.compress(data)
  if method in ('rans0', 'rans1'):
    return 4, rans_encode(data, int(method[-1]))
  raise ValueError(method)


def block(method: str, content_type: int, content_id: int, data: bytes, rans_encode=None) -> bytes:
  code, payload = compress(method, data, rans_encode)
  body = bytes([code, content_type]) + itf8(content_id) + itf8(len(payload)) + itf8(len(data)) + payload
  return body + struct.pack('<I', zlib.crc32(body))


# ------------------------------------------------------------------------------ series writers
class Series:
  """One data series: values are queued in record order, then encoded with the configured codec."""

  def __init__(self, key: str, spec):
    self.key, self.spec = key, spec
    self.values: list = []

  def codec(self):
    return self.spec[0]


def _huffman_lengths(freq: Dict[int, int]) -> Dict[int, int]:
  if len(freq) == 1:
    return {next(iter(freq)): 0}
  heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]
  heapq.heapify(heap)
  length = {s: 0 for s in freq}
  n = len(heap)
  while len(heap) > 1:
    fa, _, a = heapq.heappop(heap)
    fb, _, b = heapq.heappop(heap)
    for s in a + b:
      length[s] += 1
    n += 1
    heapq.heappush(heap, (fa + fb, n, a + b))
  return length


def _canonical(lengths: Dict[int, int]) -> Dict[int, Tuple[int, int]]:
  order = sorted(lengths, key=lambda s: (lengths[s], s))
  codes, code, prev = {}, 0, lengths[order[0]]
  for s in order:
    code <<= lengths[s] - prev
    prev = lengths[s]
    codes[s] = (code, lengths[s])
    code += 1
  return codes


class SliceWriter:
  """Collects the (series, value) events of a slice's records in decode order, then lays them out in
  the core bit stream and the external blocks."""

  def __init__(self, config: Dict[str, tuple], block_methods: Dict[int, str], rans_encode):
    self.config, self.block_methods, self.rans_encode = config, block_methods, rans_encode
    self.events: List[Tuple[str, str, object]] = []      # (series key, kind, value)

  def put_int(self, key: str, v: int):
    self.events.append((key, 'int', v))

  def put_byte(self, key: str, v: int):
    self.events.append((key, 'byte', v))

  def put_bytes(self, key: str, v: bytes):
    self.events.append((key, 'bytes', bytes(v)))

  def encodings(self) -> Dict[str, bytes]:
    """series key -> encoding bytes (codec, parameter length, parameters); Huffman alphabets come from the
    queued values."""
    out = {}
    self._huff: Dict[str, Dict[int, Tuple[int, int]]] = {}
    for key, spec in self.config.items():
      out[key] = self._encoding(key, spec)
    return out

  def _values(self, key: str, which=None):
    if which == 'len':
      return [len(v) for k, kind, v in self.events if k == key and kind == 'bytes']
    if which == 'val':
      return [b for k, kind, v in self.events if k == key and kind == 'bytes' for b in v]
    return [v for k, kind, v in self.events if k == key and kind != 'bytes']

  def _int_encoding(self, name: str, spec, values) -> bytes:
    codec = spec[0]
    if codec == 'external':
      p = itf8(spec[1])
      return itf8(1) + itf8(len(p)) + p
    if codec == 'huffman':
      freq: Dict[int, int] = {}
      for v in values:
        freq[v] = freq.get(v, 0) + 1
      if not freq:
        freq = {0: 1}
      lengths = _huffman_lengths(freq)
      self._huff[name] = _canonical(lengths)
      syms = sorted(lengths)
      p = itf8_array(syms) + itf8_array([lengths[s] for s in syms])
      return itf8(3) + itf8(len(p)) + p
    if codec == 'beta':
      p = itf8(spec[1]) + itf8(spec[2])
      return itf8(6) + itf8(len(p)) + p
    if codec == 'subexp':
      p = itf8(spec[1]) + itf8(spec[2])
      return itf8(7) + itf8(len(p)) + p
    if codec == 'gamma':
      p = itf8(spec[1])
      return itf8(9) + itf8(len(p)) + p
    raise ValueError(codec)

  def _encoding(self, key: str, spec) -> bytes:
    codec = spec[0]
    if codec == 'stop':              # ('stop', stop byte, content id)
      p = bytes([spec[1]]) + itf8(spec[2])
      return itf8(5) + itf8(len(p)) + p
    if codec == 'len':               # ('len', length spec, value spec)
      p = (self._int_encoding(key + '/len', spec[1], self._values(key, 'len')) +
           self._int_encoding(key + '/val', spec[2], self._values(key, 'val')))
      return itf8(4) + itf8(len(p)) + p
    return self._int_encoding(key, spec, self._values(key))

  def _emit_int(self, name: str, spec, v: int, core: BitWriter, ext: Dict[int, bytearray], as_byte: bool):
    codec = spec[0]
    if codec == 'external':
      ext.setdefault(spec[1], bytearray())
      ext[spec[1]] += bytes([v & 0xFF]) if as_byte else itf8(v)
    elif codec == 'huffman':
      code, n = self._huff[name][v]
      core.write(code, n)
    elif codec == 'beta':
      assert 0 <= v + spec[1] < 1 << spec[2], (name, v)
      core.write(v + spec[1], spec[2])
    elif codec == 'subexp':
      x, k = v + spec[1], spec[2]
      assert x >= 0
      if x < 1 << k:
        core.write(0, 1)
        core.write(x, k)
      else:
        b = x.bit_length() - 1
        core.write((1 << (b - k + 1)) - 1, b - k + 1)
        core.write(0, 1)
        core.write(x & ((1 << b) - 1), b)
    elif codec == 'gamma':
      x = v + spec[1]
      assert x >= 1
      b = x.bit_length() - 1
      core.write(0, b)
      core.write(1, 1)
      core.write(x & ((1 << b) - 1), b)
    else:
      raise ValueError(codec)

  def blocks(self) -> Tuple[bytes, Dict[int, bytes]]:
    """-> (core block data, {content id: external block data}); call encodings() first."""
    core, ext = BitWriter(), {}
    for key, kind, v in self.events:
      spec = self.config[key]
      if kind == 'bytes':
        if spec[0] == 'stop':
          assert spec[1] not in v
          ext.setdefault(spec[2], bytearray())
          ext[spec[2]] += v + bytes([spec[1]])
        else:
          self._emit_int(key + '/len', spec[1], len(v), core, ext, False)
          for b in v:
            self._emit_int(key + '/val', spec[2], b, core, ext, True)
      else:
        self._emit_int(key, spec, v, core, ext, kind == 'byte')
    return core.data(), {k: bytes(v) for k, v in ext.items()}


# ------------------------------------------------------------------------------ reads -> records
_BASES = 'ACGTN'


def default_subst_matrix() -> bytes:
  return bytes([0x1B] * 5)      # codes 0,1,2,3 for the other bases in ACGTN order


def _subst_code(matrix: bytes, ref: str, alt: str) -> int:
  r = _BASES.index(ref)
  others = [b for b in _BASES if b != ref]
  return (matrix[r] >> (6 - 2 * others.index(alt))) & 3


def features_of(read: dict, ref_text: str, ref_offset: int, matrix: bytes, rng) -> List[tuple]:
  """(code, 1-based read position, payload) in position order.  `ref_text[ref_offset + p]` = base at
  0-based reference position p."""
  feats = []
  rp, refp = 1, read['pos'] - 1
  seq, qual = read['seq'], read['qual']
  for op, n in read['cigar']:
    if op == 'S':
      feats.append(('S', rp, seq[rp - 1:rp - 1 + n].encode()))
      rp += n
    elif op == 'H':
      feats.append(('H', rp, n))
    elif op == 'P':
      feats.append(('P', rp, n))
    elif op == 'D':
      feats.append(('D', rp, n))
      refp += n
    elif op == 'N':
      feats.append(('N', rp, n))
      refp += n
    elif op == 'I':
      if n == 1 and rng.random() < 0.5:
        feats.append(('i', rp, ord(seq[rp - 1])))
      else:
        feats.append(('I', rp, seq[rp - 1:rp - 1 + n].encode()))
      rp += n
    elif op == 'M':
      k = 0
      while k < n:
        rb = ref_text[ref_offset + refp + k].upper() if 0 <= ref_offset + refp + k < len(ref_text) else 'N'
        b = seq[rp - 1 + k]
        if b == rb:
          k += 1
          continue
        choice = rng.random()
        if choice < 0.15 and k + 3 <= n:      # a stretch of bases stored verbatim
          feats.append(('b', rp + k, seq[rp - 1 + k:rp - 1 + k + 3].encode()))
          k += 3
        elif choice < 0.5 or b not in _BASES or rb not in _BASES:
          feats.append(('B', rp + k, (ord(b), qual[rp - 1 + k])))
          k += 1
        else:
          feats.append(('X', rp + k, _subst_code(matrix, rb, b)))
          k += 1
      rp += n
      refp += n
    else:
      raise ValueError(op)
  return feats


class CramWriter:
  """Writes one CRAM 3.0 file.  `config`: series key -> codec spec (see SliceWriter); `block_methods`:
  content id -> 'raw' | 'gzip' | 'bzip2' | 'lzma' | 'rans0' | 'rans1' (0 = the core block)."""

  def __init__(self, contigs: List[Tuple[str, str]], config: Dict[str, tuple], block_methods: Dict[int, str],
               rans_encode, read_names: bool = True, ap_delta: bool = True, matrix: Optional[bytes] = None,
               tag_specs: Optional[Dict[bytes, tuple]] = None):
    self.contigs, self.config, self.block_methods, self.rans_encode = contigs, config, block_methods, rans_encode
    self.read_names, self.ap_delta = read_names, ap_delta
    self.matrix = matrix or default_subst_matrix()
    self.tag_specs = tag_specs or {}
    text = '@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % (n, len(s)) for n, s in contigs)
    header = struct.pack('<i', len(text)) + text.encode()
    self.out = bytearray(b'CRAM' + bytes([3, 0]) + b'dv-amd-test'.ljust(20, b'\0'))
    self._container(0, 0, 0, 0, [block('raw', 0, 0, header)], [])
    self.index_rows: List[Tuple[int, int, int, int, int, int]] = []
    self.counter = 0

  def _container(self, ref_id, start, span, n_records, blocks: List[bytes], landmarks: List[int]) -> int:
    body = b''.join(blocks)
    head = (struct.pack('<i', len(body)) + itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_records) +
            ltf8(getattr(self, 'counter', 0)) + ltf8(0) + itf8(len(blocks)) + itf8_array(landmarks))
    offset = len(self.out)
    self.out += head + struct.pack('<I', zlib.crc32(head)) + body
    return offset

  def _slice(self, reads: List[dict], ref_id: int, embed: bool, with_md5: bool):
    """-> (slice header block, data blocks, encodings, tag encodings, tag list of every read, start, span)."""
    multi = ref_id == -2
    names = dict(enumerate(n for n, _ in self.contigs))
    seqs = dict(enumerate(s for _, s in self.contigs))
    if ref_id >= 0:
      start = min(r['pos'] for r in reads)
      span = max(r['pos'] + _ref_len(r['cigar']) for r in reads) - start
    else:
      start, span = 0, 0
    w = SliceWriter(dict(self.config), self.block_methods, self.rans_encode)
    tag_lists: List[tuple] = []
    tag_cfg: Dict[int, tuple] = {}
    import random
    rng = random.Random(len(reads) * 7919 + ref_id)
    prev = start
    for i, r in enumerate(reads):
      cf = r.get('cf', 0x1)
      w.put_int('BF', r['flag'] & ~(0x20 | 0x8) if not (cf & 0x2) else r['flag'])
      w.put_int('CF', cf)
      if multi:
        w.put_int('RI', r['ref_id'])
      w.put_int('RL', len(r['seq']))
      if self.ap_delta:
        w.put_int('AP', r['pos'] - prev)
        prev = r['pos']
      else:
        w.put_int('AP', r['pos'])
      w.put_int('RG', r.get('rg', -1))
      if self.read_names:
        w.put_bytes('RN', r['name'].encode())
      if cf & 0x2:
        w.put_int('MF', r.get('mf', 0))
        if not self.read_names:
          w.put_bytes('RN', r['name'].encode())
        w.put_int('NS', r.get('mate_ref', -1))
        w.put_int('NP', r.get('mate_pos', 0))
        w.put_int('TS', r.get('tlen', 0))
      elif cf & 0x4:
        w.put_int('NF', r['nf'])
      tags = tuple(r.get('tags', ()))
      key = tuple(t[0] for t in tags)
      if key not in tag_lists:
        tag_lists.append(key)
      w.put_int('TL', tag_lists.index(key))
      for t3, raw in tags:
        tid = (t3[0] << 16) | (t3[1] << 8) | t3[2]
        series = 'tag:%d' % tid
        if series not in w.config:
          w.config[series] = self.tag_specs.get(t3, ('len', ('external', 200 + len(tag_cfg)), ('external', 200 + len(tag_cfg))))
          tag_cfg[tid] = w.config[series]
        w.put_bytes(series, raw)
      if not r['flag'] & 0x4:
        rid = r['ref_id']
        feats = features_of(r, seqs[rid], 0, self.matrix, rng)
        if not cf & 0x1:      # qualities as features: the whole string as one 'q', or scores one by one
          if rng.random() < 0.5:
            feats.append(('q', 1, bytes(r['qual'])))
          else:
            feats += [('Q', k + 1, q) for k, q in enumerate(r['qual'])]
          feats.sort(key=lambda f: f[1])
        w.put_int('FN', len(feats))
        last = 0
        for code, pos, payload in feats:
          w.put_byte('FC', ord(code))
          w.put_int('FP', pos - last)
          last = pos
          if code == 'B':
            w.put_byte('BA', payload[0])
            w.put_byte('QS', payload[1])
          elif code == 'X':
            w.put_byte('BS', payload)
          elif code == 'I':
            w.put_bytes('IN', payload)
          elif code == 'i':
            w.put_byte('BA', payload)
          elif code == 'S':
            w.put_bytes('SC', payload)
          elif code == 'D':
            w.put_int('DL', payload)
          elif code == 'N':
            w.put_int('RS', payload)
          elif code == 'H':
            w.put_int('HC', payload)
          elif code == 'P':
            w.put_int('PD', payload)
          elif code == 'b':
            w.put_bytes('BB', payload)
          elif code == 'q':
            w.put_bytes('QQ', payload)
          elif code == 'Q':
            w.put_byte('QS', payload)
        w.put_int('MQ', r['mapq'])
        if cf & 0x1:
          for q in r['qual']:
            w.put_byte('QS', q)
      else:
        for b in r['seq']:
          w.put_byte('BA', ord(b))
        if cf & 0x1:
          for q in r['qual']:
            w.put_byte('QS', q)
    encodings = w.encodings()
    core, ext = w.blocks()
    embedded_id = -1
    if embed and ref_id >= 0:
      embedded_id = 99
      ext[embedded_id] = seqs[ref_id][start - 1:start - 1 + span].encode()
    md5 = bytes(16)
    if with_md5 and ref_id >= 0:
      md5 = hashlib.md5(seqs[ref_id][start - 1:start - 1 + span].upper().encode()).digest()
    ids = sorted(ext)
    blocks = [block(self.block_methods.get(0, 'raw'), 5, 0, core, self.rans_encode)]
    blocks += [block(self.block_methods.get(cid, 'raw'), 4, cid, ext[cid], self.rans_encode) for cid in ids]
    head = (itf8(ref_id) + itf8(start) + itf8(span) + itf8(len(reads)) + ltf8(self.counter) + itf8(len(blocks)) +
            itf8_array(ids) + itf8(embedded_id) + md5)
    self.counter += len(reads)
    return block('raw', 2, 0, head), blocks, encodings, tag_cfg, tag_lists, start, span

  def add_container(self, slices: List[Tuple[List[dict], int]], embed: bool = False, with_md5: bool = True,
                    ref_required: bool = True):
    """One container of one or more slices: [(reads, slice reference id: >= 0, -1 unmapped, -2 multi)].
    Every slice of a container shares the compression header, so Huffman alphabets are built over all
    of them: the slices are encoded with ONE SliceWriter pass each but the same config."""
    assert len(slices) == 1 or all('huffman' not in str(v) for v in self.config.values()), \
        'several slices per container: use codecs without a per-slice alphabet'
    built = [self._slice(reads, rid, embed, with_md5) for reads, rid in slices]
    encodings, tag_cfg, tag_lists = built[0][2], {}, []
    for b in built:
      tag_cfg.update(b[3])
      for tl in b[4]:
        if tl not in tag_lists:
          tag_lists.append(tl)
    assert all(b[4] == built[0][4] for b in built), 'slices of one container must use the same tag lists here'
    pres = (b'RN' + bytes([int(self.read_names)]) + b'AP' + bytes([int(self.ap_delta)]) + b'RR' + bytes([int(ref_required)]) +
            b'SM' + self.matrix)
    td = b''.join(b''.join(t) + b'\0' for t in tag_lists) if tag_lists else b'\0'
    pres += b'TD' + itf8(len(td)) + td
    pres = itf8(5) + pres
    series = b''.join(k.encode() + v for k, v in encodings.items() if not k.startswith('tag:'))
    series = itf8(sum(1 for k in encodings if not k.startswith('tag:'))) + series
    tags = b''.join(itf8(int(k[4:])) + v for k, v in encodings.items() if k.startswith('tag:'))
    tags = itf8(sum(1 for k in encodings if k.startswith('tag:'))) + tags
    comp = itf8(len(pres)) + pres + itf8(len(series)) + series + itf8(len(tags)) + tags
    comp_block = block('raw', 1, 0, comp)
    blocks, landmarks, at = [comp_block], [], len(comp_block)
    slice_sizes = []
    for head, data, *_ in built:
      landmarks.append(at)
      size = len(head) + sum(len(d) for d in data)
      slice_sizes.append(size)
      at += size
      blocks += [head] + data
    refs = {rid for _, rid in slices}
    c_ref = refs.pop() if len(refs) == 1 else -2
    if c_ref >= 0:
      c_start = min(b[5] for b in built)
      c_span = max(b[5] + b[6] for b in built) - c_start
    else:
      c_start = c_span = 0
    n_records = sum(len(reads) for reads, _ in slices)
    offset = self._container(c_ref, c_start, c_span, n_records, blocks, landmarks)
    for (reads, rid), lm, size, b in zip(slices, landmarks, slice_sizes, built):
      if rid >= 0:
        self.index_rows.append((rid, b[5], b[6], offset, lm, size))
      elif rid == -2:
        for ref in sorted({r['ref_id'] for r in reads if not r['flag'] & 0x4}):
          mine = [r for r in reads if r['ref_id'] == ref and not r['flag'] & 0x4]
          s0 = min(r['pos'] for r in mine)
          self.index_rows.append((ref, s0, max(r['pos'] + _ref_len(r['cigar']) for r in mine) - s0, offset, lm, size))

  def finish(self, path: str, with_index: bool = True):
    eof = bytes.fromhex('0f000000ffffffff0fe0454f4600000000010005bdd94f0001000606010001000100ee63014b')
    with open(path, 'wb') as f:
      f.write(bytes(self.out) + eof)
    if with_index:
      with gzip.open(path + '.crai', 'wt') as f:
        for row in self.index_rows:
          f.write('\t'.join(map(str, row)) + '\n')


def _ref_len(cigar) -> int:
  return sum(n for op, n in cigar if op in 'MDN=X')
