"""The cache of a checkpoint's calibration corrections (deepvariant_amd/inception_v3.py `calibrate_for_checkpoint`)
without a GPU: the device calls are stubbed, the file protocol is real."""
import os

import numpy as np

from deepvariant_amd import calibration_set
from deepvariant_amd.inception_v3 import InceptionV3


class _Stub(InceptionV3):
  def __init__(self, weights, shape=(100, 221, 7)):   # no dv_model_create
    self.input_shape, self.flat_weights, self.device_index = shape, np.asarray(weights, np.float32), 0
    self.measured, self.applied = 0, []
    self._handle = None

  def layer_table(self):
    return [(3, 3, 7, 32, 0), (3, 3, 32, 32, 100), (1, 1, 2048, 3, 200)]

  def calibrate(self, images):
    self.measured += 1
    return np.arange(67, dtype=np.float32) + float(self.flat_weights[0])

  def apply_corrections(self, corr):
    self.applied.append(np.array(corr))


def test_cache_file_protocol(tmp_path, monkeypatch):
  monkeypatch.setattr(calibration_set, 'draw', lambda shape, n, device=0: list(range(n)))
  prefix = str(tmp_path / 'model.ckpt')
  w = np.full(1000, 3.0, np.float32)
  a = _Stub(w)
  corr = a.calibrate_for_checkpoint(256, cache_prefix=prefix)
  files = sorted(os.listdir(tmp_path))
  assert a.measured == 1 and len(files) == 1 and '.dvcal-v%d-' % calibration_set.SET_VERSION in files[0]
  assert files[0].endswith('-100x221x7-n256.f32') and not any('.tmp' in f for f in files)
  b = _Stub(w)
  assert np.array_equal(b.calibrate_for_checkpoint(256, cache_prefix=prefix), corr)
  assert b.measured == 0 and len(b.applied) == 1 and b.calibration['cached'] is True
  # another set size, other weights, another shape: other names, measured again
  for other, n in ((_Stub(w), 128), (_Stub(w + 1), 256), (_Stub(w, (100, 199, 7)), 256)):
    other.calibrate_for_checkpoint(n, cache_prefix=prefix)
    assert other.measured == 1
  assert len(os.listdir(tmp_path)) == 4
  # a truncated / foreign file under the right name is not trusted
  with open(os.path.join(tmp_path, files[0]), 'wb') as f:
    f.write(b'\0' * 12)
  c = _Stub(w)
  assert np.array_equal(c.calibrate_for_checkpoint(256, cache_prefix=prefix), corr) and c.measured == 1
  assert np.fromfile(os.path.join(tmp_path, files[0]), np.float32).size == 67       # and repaired
  # a read-only place, no cache prefix, calibration off, a shape without a set
  d = _Stub(w)
  d.calibrate_for_checkpoint(256, cache_prefix='/proc/nonexistent/ckpt')
  assert d.measured == 1
  e = _Stub(w)
  e.calibrate_for_checkpoint(256)
  assert e.measured == 1 and len(os.listdir(tmp_path)) == 4
  f = _Stub(w)
  assert f.calibrate_for_checkpoint(0, cache_prefix=prefix) is None and f.measured == 0
  g = _Stub(w, (100, 221, 12))
  assert g.calibrate_for_checkpoint(256, cache_prefix=prefix) is None and g.measured == 0 and g.calibration == {'images': 0}


def test_calibration_set_shapes():
  assert calibration_set.supported((100, 221, 7)) and calibration_set.supported((100, 199, 9))
  assert calibration_set.supported((100, 147, 10)) and calibration_set.supported((300, 221, 6))
  assert not calibration_set.supported((100, 221, 12)) and not calibration_set.supported((100, 220, 7))
