"""Pure-Python writer of TensorFlow tensor bundles (`<prefix>.index` + one data shard) -- TEST
INFRASTRUCTURE for deepvariant_amd/tf_checkpoint.py and the weights importer.  Follows the
published formats (LevelDB table format; tensorflow/core/protobuf/tensor_bundle.proto):
prefix-compressed data blocks with a restart every 16 keys, uncompressed, masked CRC32C
trailers, one index block, an empty metaindex block, 48-byte footer."""
import struct

import numpy as np

from deepvariant_amd import tf_checkpoint as T

_DT = {np.dtype(np.float32): 1, np.dtype(np.int64): 9, np.dtype(np.int32): 3}


def _varint(v):
  out = bytearray()
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _mask(crc):
  return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def _block(entries, restart_interval=16):
  out = bytearray()
  restarts = []
  prev = b''
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
        shared += 1
    out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
    prev = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _with_trailer(block):
  return block + b'\0' + struct.pack('<I', _mask(T.crc32c(block + b'\0')))


def _shape_proto(shape):
  out = b''
  for d in shape:
    dim = b'\x08' + _varint(d)
    out += b'\x12' + _varint(len(dim)) + dim
  return out


def write_bundle(prefix, tensors, block_entries=40):
  """tensors: {name: ndarray}."""
  data = bytearray()
  items = [(b'', b'\x08\x01' + b'\x1a\x02\x08\x01')]   # num_shards = 1, version {producer: 1}
  for name in sorted(tensors):
    arr = np.ascontiguousarray(tensors[name])
    raw = arr.tobytes()
    shape = _shape_proto(arr.shape)
    entry = (b'\x08' + _varint(_DT[arr.dtype]) + b'\x12' + _varint(len(shape)) + shape +
             (b'\x20' + _varint(len(data)) if len(data) else b'') +
             b'\x28' + _varint(len(raw)) + b'\x35' + struct.pack('<I', _mask(T.crc32c(raw))))
    items.append((name.encode(), entry))
    data += raw
  items.sort()
  out = bytearray()
  index_entries = []
  for i in range(0, len(items), block_entries):
    chunk = items[i:i + block_entries]
    blk = _block(chunk)
    index_entries.append((chunk[-1][0], _varint(len(out)) + _varint(len(blk))))
    out += _with_trailer(blk)
  meta = _block([])
  meta_handle = _varint(len(out)) + _varint(len(meta))
  out += _with_trailer(meta)
  idx = _block(index_entries, restart_interval=1)
  idx_handle = _varint(len(out)) + _varint(len(idx))
  out += _with_trailer(idx)
  footer = meta_handle + idx_handle
  footer += b'\0' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
  out += footer
  with open(prefix + '.index', 'wb') as f:
    f.write(bytes(out))
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))
