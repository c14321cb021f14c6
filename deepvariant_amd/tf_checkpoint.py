"""Reader for TensorFlow checkpoints ("tensor bundles": `<prefix>.index` +
`<prefix>.data-0000i-of-0000N`) without TensorFlow.

call_variants loads its model with `model.load_weights(checkpoint)`
(deepvariant/call_variants.py:759-762, deepvariant/keras_modeling.py:304-335); the file
format behind that call is third-party (TensorFlow 2.16.1, pinned in
/root/reference/settings.sh; tensorflow/core/util/tensor_bundle/tensor_bundle.cc):

  * `.index` is an SSTable in LevelDB's table format (tensorflow/core/lib/io/table*,
    format.cc): data blocks of prefix-compressed (key, value) entries with a restart array,
    an index block mapping separator keys to block handles, and a 48-byte footer
    (metaindex handle, index handle, magic 0xdb4775248b80fb57).  Every block is followed
    by a 1-byte compression type (0 = none, 1 = snappy) and a masked CRC32C.
  * key "" holds a BundleHeaderProto, every other key a BundleEntryProto
    (tensorflow/core/protobuf/tensor_bundle.proto): dtype, shape, shard_id, offset, size,
    crc32c of the tensor bytes (masked).
  * `.data-*` are the raw little-endian tensor bytes.

Only what a weights import needs is implemented: dense numeric tensors (no slices, no
string/variant tensors -- those entries are listed but not decoded).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {
    1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
    9: np.int64, 10: np.bool_, 14: None,  # bfloat16: handled below
    17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64,
}
DT_STRING = 7


def crc32c(data: bytes) -> int:
  """CRC32C (Castagnoli).  Uses libdvhip's host routine when the library is built."""
  try:
    from deepvariant_amd import _lib
    buf = np.frombuffer(data, np.uint8)
    return int(_lib.lib().dv_crc32c(buf.ctypes.data if buf.size else None, buf.size))
  except Exception:  # pylint: disable=broad-except
    table = _crc_table()
    crc = 0xFFFFFFFF
    for b in data:
      crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_CRC_TABLE: List[int] = []


def _crc_table() -> List[int]:
  if not _CRC_TABLE:
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
      _CRC_TABLE.append(c)
  return _CRC_TABLE


def unmask_crc(masked: int) -> int:
  rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
  return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
  out = shift = 0
  while True:
    b = buf[pos]
    pos += 1
    out |= (b & 0x7F) << shift
    if not b & 0x80:
      return out, pos
    shift += 7
    if shift > 63:
      raise ValueError('varint too long')


def _read_block(f: bytes, offset: int, size: int, verify: bool) -> bytes:
  """Block contents; the trailer is 1 byte compression type + 4 bytes masked CRC32C of
  (contents + type)."""
  if offset + size + 5 > len(f):
    raise ValueError('block handle past the end of the index file')
  body = f[offset:offset + size]
  ctype = f[offset + size]
  if verify:
    want = unmask_crc(struct.unpack_from('<I', f, offset + size + 1)[0])
    if crc32c(f[offset:offset + size + 1]) != want:
      raise ValueError('checkpoint index: block checksum mismatch')
  if ctype == 1:
    return _snappy_uncompress(body)
  if ctype != 0:
    raise ValueError('checkpoint index: unknown block compression %d' % ctype)
  return body


def _snappy_uncompress(src: bytes) -> bytes:
  """Raw snappy (the format description in google/snappy format_description.txt)."""
  n, pos = _varint(src, 0)
  out = bytearray()
  while pos < len(src):
    tag = src[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(src[pos:pos + nb], 'little')
        pos += nb
      ln += 1
      out += src[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = ((tag >> 2) & 7) + 4
      off = ((tag >> 5) << 8) | src[pos]
      pos += 1
    elif kind == 2:
      ln = (tag >> 2) + 1
      off = src[pos] | (src[pos + 1] << 8)
      pos += 2
    else:
      ln = (tag >> 2) + 1
      off = int.from_bytes(src[pos:pos + 4], 'little')
      pos += 4
    if off == 0 or off > len(out):
      raise ValueError('corrupt snappy block')
    for _ in range(ln):          # copies may overlap their own output
      out.append(out[-off])
  if len(out) != n:
    raise ValueError('corrupt snappy block (length)')
  return bytes(out)


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
  if len(block) < 4:
    raise ValueError('checkpoint index: short block')
  n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  end = len(block) - 4 - 4 * n_restarts
  if end < 0:
    raise ValueError('checkpoint index: bad restart array')
  pos, key = 0, b''
  while pos < end:
    shared, pos = _varint(block, pos)
    non_shared, pos = _varint(block, pos)
    vlen, pos = _varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > end:
      raise ValueError('checkpoint index: corrupt entry')
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    yield key, block[pos:pos + vlen]
    pos += vlen


def read_table(index_bytes: bytes, verify: bool = True) -> Dict[bytes, bytes]:
  """All (key, value) pairs of an SSTable."""
  if len(index_bytes) < 48:
    raise ValueError('not a checkpoint index (too short)')
  footer = index_bytes[-48:]
  if struct.unpack_from('<Q', footer, 40)[0] != _TABLE_MAGIC:
    raise ValueError('not a checkpoint index (bad table magic)')
  pos = 0
  _, pos = _varint(footer, pos)      # metaindex handle
  _, pos = _varint(footer, pos)
  idx_off, pos = _varint(footer, pos)
  idx_size, pos = _varint(footer, pos)
  out: Dict[bytes, bytes] = {}
  for _, handle in _block_entries(_read_block(index_bytes, idx_off, idx_size, verify)):
    off, p = _varint(handle, 0)
    size, p = _varint(handle, p)
    for k, v in _block_entries(_read_block(index_bytes, off, size, verify)):
      out[k] = v
  return out


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
  pos = 0
  while pos < len(buf):
    tag, pos = _varint(buf, pos)
    num, wt = tag >> 3, tag & 7
    if wt == 0:
      v, pos = _varint(buf, pos)
    elif wt == 1:
      v = buf[pos:pos + 8]
      pos += 8
    elif wt == 2:
      ln, pos = _varint(buf, pos)
      v = buf[pos:pos + ln]
      pos += ln
    elif wt == 5:
      v = buf[pos:pos + 4]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wt)
    yield num, wt, v


class Entry:
  """BundleEntryProto."""

  def __init__(self, raw: bytes):
    self.dtype = 0
    self.shape: List[int] = []
    self.shard_id = 0
    self.offset = 0
    self.size = 0
    self.crc32c = None
    self.sliced = False
    for num, wt, v in _fields(raw):
      if num == 1:
        self.dtype = v
      elif num == 2:
        for n2, _, dim in _fields(v):          # TensorShapeProto.dim
          if n2 == 2:
            size = 0
            for n3, _, x in _fields(dim):
              if n3 == 1:
                size = x if x < (1 << 63) else x - (1 << 64)
            self.shape.append(size)
      elif num == 3:
        self.shard_id = v
      elif num == 4:
        self.offset = v
      elif num == 5:
        self.size = v
      elif num == 6:
        self.crc32c = struct.unpack('<I', v)[0]
      elif num == 7:
        self.sliced = True


class CheckpointReader:
  """`tf.train.load_checkpoint(prefix)` for dense tensors:
  get_variable_to_shape_map(), get_tensor(name)."""

  def __init__(self, prefix: str, verify: bool = True):
    if prefix.endswith('.index'):
      prefix = prefix[:-len('.index')]
    self.prefix = prefix
    self.verify = verify
    with open(prefix + '.index', 'rb') as f:
      table = read_table(f.read(), verify)
    if b'' not in table:
      raise ValueError('checkpoint index without a bundle header')
    self.num_shards = 1
    for num, _, v in _fields(table[b'']):
      if num == 1:
        self.num_shards = v
      elif num == 2 and v != 0:
        raise ValueError('big-endian checkpoints are not supported')
    self.entries = {k.decode(): Entry(v) for k, v in table.items() if k != b''}
    self._shards: Dict[int, np.memmap] = {}

  def get_variable_to_shape_map(self) -> Dict[str, List[int]]:
    return {k: list(e.shape) for k, e in self.entries.items()}

  def has_tensor(self, name: str) -> bool:
    return name in self.entries

  def _shard(self, i: int):
    if i not in self._shards:
      path = '%s.data-%05d-of-%05d' % (self.prefix, i, self.num_shards)
      self._shards[i] = np.memmap(path, np.uint8, 'r') if os.path.getsize(path) else \
          np.zeros(0, np.uint8)
    return self._shards[i]

  def get_tensor(self, name: str) -> np.ndarray:
    if name not in self.entries:
      raise KeyError('tensor %r is not in the checkpoint' % name)
    e = self.entries[name]
    if e.sliced:
      raise ValueError('%s: partitioned (sliced) tensors are not supported' % name)
    if e.dtype == DT_STRING or e.dtype not in _DTYPES:
      raise ValueError('%s: dtype %d is not a dense numeric tensor' % (name, e.dtype))
    data = self._shard(e.shard_id)
    if e.offset + e.size > data.size:
      raise ValueError('%s: data shard is shorter than the index says' % name)
    raw = bytes(data[e.offset:e.offset + e.size])
    if self.verify and e.crc32c is not None and crc32c(raw) != unmask_crc(e.crc32c):
      raise ValueError('%s: tensor checksum mismatch' % name)
    if e.dtype == 14:                      # bfloat16 -> float32
      arr = (np.frombuffer(raw, np.uint16).astype(np.uint32) << 16).view(np.float32)
    else:
      arr = np.frombuffer(raw, _DTYPES[e.dtype])
    n = int(np.prod(e.shape)) if e.shape else 1
    if arr.size != n:
      raise ValueError('%s: %d values for shape %s' % (name, arr.size, e.shape))
    return arr.reshape(e.shape)
