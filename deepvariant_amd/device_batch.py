"""A `dv_batch` whose arrays live in HBM (torch tensors own the memory).

PyTorch is plumbing here: it allocates device memory and provides the stream;
the encoder is called through the C ABI with raw device pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from deepvariant_amd import _lib
from deepvariant_amd import packing

_FIELDS = [
    ('read_pos', np.int32), ('read_sort_pos', np.int32),
    ('read_seq_off', np.uint32), ('read_cigar_off', np.uint32),
    ('read_mapq', np.uint8), ('read_flags', np.uint8),
    ('read_frag_len', np.int32), ('read_hp', np.int32),
    ('read_name_rank', np.uint32), ('read_aux', np.uint8),
    ('bases', np.uint8), ('quals', np.uint8), ('mod_5mc', np.uint8),
    ('mod_6ma', np.uint8), ('cigar', np.uint32),
    ('item_variant_start', np.int32), ('item_image_start', np.int32),
    ('item_ref_idx', np.uint32), ('item_list_off', np.uint32),
    ('item_height', np.uint16), ('item_out_off', np.uint64),
    ('item_blank_mask', np.uint32), ('item_mean_coverage', np.float32),
    ('ref_windows', np.uint8), ('list_read', np.uint32),
    ('list_code', np.uint8), ('list_group', np.uint8), ('list_aux', np.uint8),
    ('base_aux0', np.uint8), ('base_aux1', np.uint8), ('base_aux2', np.uint8),
    ('ref_aux0', np.uint8), ('ref_aux1', np.uint8), ('ref_aux2', np.uint8),
]


class DeviceBatch:
  """Uploads a PackedBatch once; `encode` launches with device pointers."""

  def __init__(self, batch: packing.PackedBatch, device: torch.device,
               reference_band_height: int = 5):
    # dv_encode_batch cannot inspect a device-resident batch: run its host-side checks
    # (CIGAR ops, CIGAR vs sequence length, index ranges) on the host image before the
    # upload, so that a malformed read fails here instead of reading out of bounds on the GPU
    host_c, keep = batch.to_ctypes()
    _lib.check(_lib.lib().dv_validate_batch(C.byref(host_c), int(reference_band_height)))
    del keep
    self.n_items = batch.n_items
    self.width = batch.width
    self.c = _lib.DvBatch()
    self.c.memory = _lib.DV_MEM_DEVICE
    # every array of the batch at a 16-byte aligned offset of ONE host image, one upload: a 1 kb
    # calling region's batch is ~25 arrays of a few KB each, and 25 small copies cost more than the
    # encoder launch they feed
    parts, at = [], 0
    for name, dtype in _FIELDS:
      arr = getattr(batch, name)
      if arr is None:
        setattr(self.c, name, None)
        continue
      raw = np.ascontiguousarray(arr, dtype=dtype).view(np.uint8).reshape(-1)
      size = max(int(raw.size), 16)
      size += (-size) % 16
      parts.append((name, at, raw))
      at += size
    image = np.zeros(at, np.uint8)
    for name, off, raw in parts:
      image[off:off + raw.size] = raw
    self.storage = torch.from_numpy(image).to(device)
    base = self.storage.data_ptr()
    self.tensors = {}
    for k, (name, off, raw) in enumerate(parts):
      end = parts[k + 1][1] if k + 1 < len(parts) else at
      self.tensors[name] = self.storage[off:end]
      setattr(self.c, name, base + off)
    t = batch.table
    self.c.n_reads = t.n_reads
    self.c.n_bases = int(t.read_seq_off[-1])
    self.c.n_cigar = int(t.read_cigar_off[-1])
    self.c.n_items = batch.n_items
    self.c.n_ref_windows = len(batch.ref_windows_list)
    self.c.n_list = int(batch.item_list_off[-1])
    self.c.max_list_len = batch.max_list_len
    self.c.max_cigar_ops, self.c.max_item_height = host_c.max_cigar_ops, host_c.max_item_height
    self.input_bytes = int(self.storage.numel())

  def encode(self, encoder, out_channels: int, out: torch.Tensor,
             rows: torch.Tensor = None, stream=None):
    if stream is None:
      stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().dv_encode_batch(
        encoder.handle, C.byref(self.c), out_channels, out.data_ptr(),
        rows.data_ptr() if rows is not None else None, _lib.DV_MEM_DEVICE,
        C.c_void_p(stream)))
