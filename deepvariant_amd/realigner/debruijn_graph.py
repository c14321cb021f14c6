"""Host mirror of deepvariant/realigner/python/debruijn_graph (pybind of DeBruijnGraph) over
the C ABI: build(ref, reads, options) -> DeBruijnGraph or None; .kmer_size,
.candidate_haplotypes(), .graphviz().  All graph work is native (csrc/debruijn_graph.cpp)."""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Optional, Sequence

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import packing


@dataclasses.dataclass
class DeBruijnGraphOptions:
  """deepvariant/protos/realigner.proto DeBruijnGraphOptions; defaults = realigner.py's flags."""
  min_k: int = 10
  max_k: int = 101
  step_k: int = 1
  min_mapq: int = 14
  min_base_quality: int = 15
  min_edge_weight: int = 2
  max_num_paths: int = 256
  disable_graph_pruning: bool = False


class DeBruijnGraph:
  def __init__(self, handle):
    self._h = handle

  def __del__(self):
    if getattr(self, '_h', None):
      _lib.lib().dv_debruijn_destroy(self._h)
      self._h = None

  @property
  def kmer_size(self) -> int:
    return int(_lib.lib().dv_debruijn_kmer_size(self._h))

  def candidate_haplotypes(self) -> List[str]:
    n = C.c_int32()
    arr = C.POINTER(C.c_char_p)()
    _lib.check(_lib.lib().dv_debruijn_haplotypes(self._h, C.byref(n), C.byref(arr)))
    return [arr[i].decode() for i in range(n.value)]

  def graphviz(self) -> str:
    text = C.c_char_p()
    _lib.check(_lib.lib().dv_debruijn_graphviz(self._h, C.byref(text)))
    return text.value.decode()


def build_from_table(ref: str, table: packing.ReadTable, read_indices: Sequence[int],
                     options: DeBruijnGraphOptions) -> Optional[DeBruijnGraph]:
  """build() on reads that are already packed: `read_indices` selects the window's reads from
  the region's read table, in the order the reference would add them."""
  opt = _lib.DvDebruijnOptions(options.min_k, options.max_k, options.step_k, options.min_mapq,
                               options.min_base_quality, options.min_edge_weight, options.max_num_paths,
                               int(bool(options.disable_graph_pruning)))
  idx = np.ascontiguousarray(read_indices, np.int32)
  bases = np.ascontiguousarray(table.bases, np.uint8)
  quals = np.ascontiguousarray(table.quals, np.uint8)
  seq_off = np.ascontiguousarray(table.read_seq_off, np.uint32)
  mapq = np.ascontiguousarray(table.read_mapq, np.uint8)
  raw = ref.encode()
  handle = C.c_void_p()
  _lib.check(_lib.lib().dv_debruijn_build(
      raw, len(raw), bases.ctypes.data, quals.ctypes.data, len(bases), seq_off.ctypes.data, mapq.ctypes.data,
      table.n_reads, idx.ctypes.data, len(idx), C.byref(opt), C.byref(handle)))
  return DeBruijnGraph(handle) if handle.value else None


def build(ref: str, reads: Sequence, options: DeBruijnGraphOptions) -> Optional[DeBruijnGraph]:
  """debruijn_graph.build(ref, reads, options)."""
  table = packing.ReadTable.from_reads(list(reads))
  return build_from_table(ref, table, range(len(reads)), options)
