"""TFRecord (optionally GZIP) reader/writer, byte-compatible with TensorFlow's.

Replaces the sinks the reference uses on this path:
  nucleus::ExampleWriter::TfRecordImpl::Add  third_party/nucleus/io/example_writer.cc:77-110
  tfrecord.Writer / read_tfrecords          third_party/nucleus/io/tfrecord.py:53-170

Record framing (tensorflow/core/lib/io/record_writer.cc, not in-tree):
  uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
with masked(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8 (mod 2^32); the
whole file is one gzip stream when the name ends in .gz.
"""
from __future__ import annotations

import gzip
import struct
from typing import Iterator, Optional

import numpy as np

_CRC_TABLE = None


def _crc_table():
  global _CRC_TABLE
  if _CRC_TABLE is None:
    poly = 0x82F63B78  # CRC-32C (Castagnoli), reflected
    t = np.zeros((8, 256), dtype=np.uint32)
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ poly if c & 1 else c >> 1
      t[0, i] = c
    for k in range(1, 8):
      for i in range(256):
        c = int(t[k - 1, i])
        t[k, i] = (c >> 8) ^ int(t[0, c & 0xFF])
    _CRC_TABLE = t
  return _CRC_TABLE


def crc32c(data: bytes) -> int:
  """Slicing-by-8 CRC32C; pure Python/numpy (the hot path uses dv_crc32c)."""
  try:
    from deepvariant_amd import _lib
    fast = _lib.try_crc32c(data)
    if fast is not None:
      return fast
  except Exception:  # pylint: disable=broad-except
    pass
  t = _crc_table()
  t0 = [int(x) for x in t[0]]
  crc = 0xFFFFFFFF
  for b in bytes(data):
    crc = (crc >> 8) ^ t0[(crc ^ b) & 0xFF]
  return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
  crc = crc32c(data)
  return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecords(path: str, verify_crc: bool = False,
                   max_records: Optional[int] = None) -> Iterator[bytes]:
  opener = gzip.open if _is_gzip(path) else open
  n = 0
  with opener(path, 'rb') as f:
    while True:
      hdr = f.read(12)
      if len(hdr) < 12:
        return
      length, len_crc = struct.unpack('<QI', hdr)
      data = f.read(length)
      (data_crc,) = struct.unpack('<I', f.read(4))
      if verify_crc:
        if masked_crc32c(hdr[:8]) != len_crc:
          raise IOError('corrupted record length at record %d' % n)
        if masked_crc32c(data) != data_crc:
          raise IOError('corrupted record data at record %d' % n)
      yield data
      n += 1
      if max_records is not None and n >= max_records:
        return


def _is_gzip(path: str) -> bool:
  with open(path, 'rb') as f:
    return f.read(2) == b'\x1f\x8b'


class Writer:
  """`with Writer(path) as w: w.write(bytes)`; GZIP iff path ends with .gz
  (same rule as third_party/nucleus/io/tfrecord.py:94-100)."""

  def __init__(self, path: str, compression_type: Optional[str] = None,
               compresslevel: int = 6):
    if compression_type is None:
      compression_type = 'GZIP' if path.endswith('.gz') else ''
    self._f = (gzip.open(path, 'wb', compresslevel=compresslevel)
               if compression_type == 'GZIP' else open(path, 'wb'))

  def write(self, data: bytes):
    hdr = struct.pack('<Q', len(data))
    self._f.write(hdr)
    self._f.write(struct.pack('<I', masked_crc32c(hdr)))
    self._f.write(data)
    self._f.write(struct.pack('<I', masked_crc32c(data)))

  def close(self):
    self._f.close()

  def __enter__(self):
    return self

  def __exit__(self, *args):
    self.close()
