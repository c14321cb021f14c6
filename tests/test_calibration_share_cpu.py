"""The calibration hand-over between host ranks that share a GPU (deepvariant_amd/inception_v3.py
`_calibrate_or_share`), without a GPU: the device calls are stubbed, the file protocol is real."""
import glob
import os
import threading

import numpy as np

from deepvariant_amd import inception_v3


class _Stub(inception_v3.InceptionV3):
  """InceptionV3 with the two device calls replaced: `calibrate` "measures" a fixed vector (and counts),
  `apply_corrections` records what it was given."""

  def __init__(self, weights, delay=0.0):     # pylint: disable=super-init-not-called
    self.flat_weights = weights
    self.input_shape = (100, 221, 7)
    self.measured = 0
    self.applied = None
    self._delay = delay

  def __del__(self):
    pass

  def calibrate(self, images):
    import time
    time.sleep(self._delay)
    self.measured += 1
    return np.arange(17, dtype=np.float32) * 0.5 + float(len(images))

  def apply_corrections(self, corrections):
    self.applied = np.array(corrections, np.float32)


def _cleanup(key):
  for f in glob.glob('/dev/shm/dvamd-cal-%s-*' % key) + glob.glob('/tmp/dvamd-cal-%s-*' % key):
    os.remove(f)


def test_one_model_measures_and_the_others_apply():
  key = 'cputest-%d' % os.getpid()
  _cleanup(key)
  try:
    w = np.linspace(-1, 1, 50000).astype(np.float32)
    models = [_Stub(w, delay=0.05) for _ in range(4)]
    threads = [threading.Thread(target=m._calibrate_or_share, args=(list(range(256)), key)) for m in models]   # pylint: disable=protected-access
    for t in threads:
      t.start()
    for t in threads:
      t.join(timeout=30)
    assert sum(m.measured for m in models) == 1                     # exactly one measurement per key
    want = np.arange(17, dtype=np.float32) * 0.5 + 256.0
    for m in models:
      assert m.measured == 1 or np.array_equal(m.applied, want)
    # other weights under the same key do not pick the file up (the name carries a fingerprint of the weights)
    other = _Stub(w * 2.0)
    other._calibrate_or_share(list(range(100)), key)                 # pylint: disable=protected-access
    assert other.measured == 1 and other.applied is None
    # without a key nothing is shared
    alone = _Stub(w)
    alone._calibrate_or_share(list(range(64)), None)                 # pylint: disable=protected-access
    assert alone.measured == 1 and alone.applied is None
  finally:
    _cleanup(key)
