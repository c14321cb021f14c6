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
  assert l.dv_abi_version() == 8   # 4: dv_realign_regions; 5: dv_cram_read_region; 6: base_aux2 + flow-space channels; 7: dv_model_calibrate; 8: dv_model_probe_rounding, calibration sets


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


def test_realigner_and_phasing_entry_points_reject_malformed_input():
  """dv_debruijn_build / dv_phase_reads / dv_local_align_many validate before they index."""
  import ctypes as C
  import numpy as np
  l = _lib.lib()
  opt = _lib.DvDebruijnOptions(10, 20, 1, 0, 0, 2, 16, 0)
  handle = C.c_void_p()
  bases = np.frombuffer(b'ACGTACGTAC', np.uint8)
  quals = np.full(10, 30, np.uint8)
  off = np.array([0, 10], np.uint32)
  mapq = np.array([60], np.uint8)
  def build(ref=b'ACGTTGCAAGCTTGGATCCA', n_bases=10, reads=(0,), n_table=1, options=opt, seq_off=off):
    idx = np.ascontiguousarray(reads, np.int32)
    return l.dv_debruijn_build(ref, len(ref), bases.ctypes.data, quals.ctypes.data, n_bases, seq_off.ctypes.data,
                               mapq.ctypes.data, n_table, idx.ctypes.data, len(idx), C.byref(options), C.byref(handle))
  assert build() == _lib.DV_OK
  if handle.value:
    l.dv_debruijn_destroy(handle)
  assert build(reads=(1,)) != _lib.DV_OK                       # read index outside the table
  assert build(reads=(-1,)) != _lib.DV_OK
  assert build(n_bases=5) != _lib.DV_OK                        # the read's bases end past the array
  assert build(seq_off=np.array([10, 0], np.uint32)) != _lib.DV_OK
  assert build(options=_lib.DvDebruijnOptions(0, 20, 1, 0, 0, 2, 16, 0)) != _lib.DV_OK     # min_k 0
  assert build(options=_lib.DvDebruijnOptions(10, 20, 0, 0, 0, 2, 16, 0)) != _lib.DV_OK    # step_k 0
  assert 'dv_debruijn_build' in l.dv_last_error().decode()

  cand = (_lib.DvPhasingCandidate * 1)(_lib.DvPhasingCandidate(100, 101, 0, 2))
  alleles = (_lib.DvPhasingAllele * 2)(_lib.DvPhasingAllele(0, 1, 0, 0, 2, 0), _lib.DvPhasingAllele(1, 1, 0, 2, 2, 0))
  support = np.array([0, 1, 2, 3], np.int32)
  lowq = np.zeros(4, np.uint8)
  phases = np.zeros(4, np.int32)
  def phase(n_alleles=2, n_bases=2, n_support=4, n_reads=4, sup=support):
    return l.dv_phase_reads(cand, 1, alleles, n_alleles, b'AC', n_bases, sup.ctypes.data, lowq.ctypes.data, n_support,
                            n_reads, 1, phases.ctypes.data, None, None, None, 0)
  assert phase() == _lib.DV_OK
  assert phase(n_alleles=1) != _lib.DV_OK                      # the candidate's allele range leaves the table
  assert phase(n_bases=1) != _lib.DV_OK                        # an allele's bases leave the string
  assert phase(n_support=3) != _lib.DV_OK                      # ... its support range too
  assert phase(n_reads=3) != _lib.DV_OK                        # a read index >= n_reads
  assert phase(n_reads=-1) != _lib.DV_OK
  assert 'dv_phase_reads' in l.dv_last_error().decode()

  out = (_lib.DvLocalAlignment * 1)()
  assert l.dv_local_align_many(None, 1, None, 2, 2, 3, 1, out) != _lib.DV_OK
  assert l.dv_local_align_many(b'ACGT', -1, None, 2, 2, 3, 1, out) != _lib.DV_OK
