"""The C-ABI library loads and exports every symbol include/dvhip.h declares."""
import os
import re

import pytest

from deepvariant_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'dvhip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(dv_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  l = _lib.lib()
  declared = _declared_symbols()
  assert declared, 'no declarations parsed'
  for sym in declared:
    assert hasattr(l, sym), sym
  assert sorted(_lib.ABI_SYMBOLS) == declared
  assert l.dv_abi_version() == 1


def test_host_helpers_need_no_gpu():
  import ctypes as C
  import numpy as np
  l = _lib.lib()
  assert _lib.try_crc32c(b'123456789') == 0xE3069283
  out = np.zeros(120, np.int32)
  assert l.dv_downsample_indices(120, 95, C.c_uint32(2101079370),
                                 out.ctypes.data_as(C.c_void_p)) == 0
  from oracle import oracle as O
  assert (out == O.downsample_indices(120, 95, 2101079370)).all()


def test_no_cpu_fallback_without_device():
  import ctypes as C
  from deepvariant_amd import packing
  from tests.golden.make_golden import wgs_options
  if _lib.device_count() > 0:
    pytest.skip('GPU present')
  o = packing.make_encoder_options(wgs_options())
  h = C.c_void_p()
  rc = _lib.lib().dv_encoder_create(C.byref(o), 0, C.byref(h))
  assert rc == _lib.DV_ERR_NO_DEVICE


def test_model_create_rejects_oversized_batch_before_touching_a_device():
  """Argument validation comes first: max_batch > 8192 is DV_ERR_INVALID_ARGUMENT with or
  without a GPU (include/dvhip.h dv_model_desc)."""
  import ctypes as C
  from deepvariant_amd import _lib
  lib = _lib.lib()
  desc = _lib.DvModelDesc(100, 221, 7, 3, 8193)
  handle = C.c_void_p()
  rc = lib.dv_model_create(C.byref(desc), 0, C.byref(handle))
  assert rc == _lib.DV_ERR_INVALID_ARGUMENT
  assert b'max_batch' in lib.dv_last_error()
  desc = _lib.DvModelDesc(64, 221, 7, 3, 16)        # H < 75: Inception-v3's minimum
  assert lib.dv_model_create(C.byref(desc), 0, C.byref(handle)) == _lib.DV_ERR_INVALID_ARGUMENT


def test_product_package_never_imports_the_oracle():
  """oracle/ is test infrastructure: nothing under deepvariant_amd/ may import it."""
  import os
  import re
  root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'deepvariant_amd')
  for dirpath, _, files in os.walk(root):
    for f in files:
      if f.endswith(('.py', '.hip', '.cpp', '.h')):
        text = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f
        assert 'libdvoracle' not in text, f


def test_count_alleles_rejects_malformed_host_tables():
  """dv_count_alleles validates a host-resident read table before any kernel could read it
  (no GPU needed: the checks come first)."""
  import ctypes as C
  import numpy as np
  from deepvariant_amd import _lib
  b = _lib.DvBatch()
  pos = np.array([5], np.int32)
  seq_off = np.array([0, 3], np.uint32)
  cig_off = np.array([0, 1], np.uint32)
  mapq = np.array([60], np.uint8)
  bases = np.frombuffer(b'ACG', np.uint8).copy()
  quals = np.array([30, 30, 30], np.uint8)
  b.memory, b.n_reads, b.n_bases, b.n_cigar = _lib.DV_MEM_HOST, 1, 3, 1
  for name, arr in (('read_pos', pos), ('read_seq_off', seq_off), ('read_cigar_off', cig_off),
                    ('read_mapq', mapq), ('bases', bases), ('quals', quals)):
    setattr(b, name, arr.ctypes.data)
  opt = _lib.DvAlleleCounterOptions(0, 20, 0, 20, b'A' * 40, 0, 40, 40, 0, 0, 0)
  h = C.c_void_p()
  for cigar, message in (((10 << 4) | 1, 'CIGAR consumes more bases'), ((3 << 4) | 12, 'Unrecognized CIGAR op')):
    cig = np.array([cigar], np.uint32)
    b.cigar = cig.ctypes.data
    rc = _lib.lib().dv_count_alleles(C.byref(b), C.byref(opt), C.byref(h), None)
    assert rc == _lib.DV_ERR_BAD_INPUT and message in _lib.lib().dv_last_error().decode()
