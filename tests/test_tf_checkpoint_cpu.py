"""TensorFlow checkpoint import without TensorFlow (deepvariant_amd/tf_checkpoint.py,
keras_layout.py, call_variants.import_keras_checkpoint): reader vs a bundle written by the
pure-Python writer in tests/, vs the real checkpoint in the reference tree when present, and
the Keras layer ordering that names the variables."""
import os

import numpy as np
import pytest

from deepvariant_amd import call_variants as cv
from deepvariant_amd import keras_layout, tf_checkpoint
from tests import tf_bundle_writer

REF_BUNDLE = '/root/reference/deepvariant/multiallelic_model/variables/variables'


def test_reader_round_trip(tmp_path):
  rng = np.random.default_rng(1)
  tensors = {'layer_with_weights-%d/kernel/.ATTRIBUTES/VARIABLE_VALUE' % i:
             rng.standard_normal((3, 3, 4, 5 + i)).astype(np.float32) for i in range(120)}
  tensors['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(7, np.int64)
  prefix = str(tmp_path / 'ckpt-1')
  tf_bundle_writer.write_bundle(prefix, tensors)
  r = tf_checkpoint.CheckpointReader(prefix)
  assert r.get_variable_to_shape_map().keys() == tensors.keys()
  for k, v in tensors.items():
    np.testing.assert_array_equal(r.get_tensor(k), v)
  # corruption is detected (tensor CRC, block CRC)
  with open(prefix + '.data-00000-of-00001', 'r+b') as f:
    f.seek(100)
    f.write(b'\xff')
  with pytest.raises(ValueError, match='checksum'):
    for k in tensors:
      tf_checkpoint.CheckpointReader(prefix).get_tensor(k)
  with open(prefix + '.index', 'r+b') as f:
    f.seek(50)
    f.write(b'\xff')
  with pytest.raises(ValueError):
    tf_checkpoint.CheckpointReader(prefix)


@pytest.mark.skipif(not os.path.exists(REF_BUNDLE + '.index'), reason='reference tree not present')
def test_reader_on_the_reference_trees_real_checkpoint():
  """deepvariant/multiallelic_model/variables: written by TensorFlow itself, two data shards.
  Pinned facts (shapes, shard, a value) were read with this reader in the build container;
  TensorFlow's own block and tensor checksums verify on every read."""
  r = tf_checkpoint.CheckpointReader(REF_BUNDLE)
  assert r.num_shards == 2 and len(r.entries) == 34
  shapes = r.get_variable_to_shape_map()
  assert shapes['layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'] == [9, 8]
  assert shapes['layer_with_weights-3/bias/.ATTRIBUTES/VARIABLE_VALUE'] == [6]
  assert int(r.get_tensor('optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE')) == 276588
  assert abs(float(r.get_tensor('optimizer/learning_rate/.ATTRIBUTES/VARIABLE_VALUE')) - 0.007) < 1e-6
  for k, e in r.entries.items():
    if e.dtype != tf_checkpoint.DT_STRING:
      assert r.get_tensor(k).shape == tuple(e.shape)


def test_keras_layer_order_facts():
  """What is known about `model.layers` of the reference model without running Keras:
  189 layers own variables (94 conv + 94 BN + Dense); the first is the first convolution
  (deepvariant/keras_modeling.py:176-184), the last the classification head; and the order
  is by depth, so inside an Inception block the LONGEST branch's first conv comes before the
  shorter branches' (mixed0: the 3x3dbl branch's 1x1 -- construction index 8 -- precedes the
  5x5 branch's 1x1 -- index 6 -- and the 1x1 branch -- index 5)."""
  entries, n = keras_layout.variable_layout(7)
  assert n == 21810083
  names = [e[0] for e in entries]
  assert len({nm.split('/')[0] for nm in names}) == 189
  assert entries[0][0].startswith('layer_with_weights-0/kernel') and entries[0][1] == (3, 3, 7, 32)
  assert entries[-1][0].startswith('layer_with_weights-188/bias')
  offs = sorted(e[2] for e in entries)
  sizes = {e[2]: int(np.prod(e[1])) for e in entries}
  assert offs[0] == 0 and all(a + sizes[a] == b for a, b in zip(offs, offs[1:]))   # a partition
  g, out = keras_layout.build_graph(7)
  order = [l.conv_index for l in keras_layout.keras_layer_order(out) if l.kind == 'conv']
  assert order[:5] == [0, 1, 2, 3, 4]
  assert order.index(8) < order.index(6) < order.index(5)
  assert sorted(order) == list(range(94))


def test_import_matches_the_flat_layout(tmp_path):
  """A checkpoint laid out like the reference's (names from the Keras order) imports to
  exactly the flat array the oracle exports; shape mismatches and old layouts are errors."""
  from oracle import inception_ref as R
  ref = R.make_random_model(7, seed=4)
  flat = ref.export_flat()
  entries, n = keras_layout.variable_layout(7)
  assert n == flat.size
  tensors = {name: flat[off:off + int(np.prod(shape))].reshape(shape) for name, shape, off in entries}
  tensors['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(1, np.int64)
  prefix = str(tmp_path / 'model' / 'ckpt')
  os.makedirs(os.path.dirname(prefix))
  tf_bundle_writer.write_bundle(prefix, tensors)
  got = cv.import_keras_checkpoint(prefix, 7)
  np.testing.assert_array_equal(got, flat)
  assert cv.checkpoint_prefix(prefix + '.index') == prefix
  # channel mismatch: error by default; with the flag the common channels are copied
  with pytest.raises(ValueError, match='input channels'):
    cv.import_keras_checkpoint(prefix, 6)
  six = cv.import_keras_checkpoint(prefix, 6, allow_channel_mismatch=True)
  assert six.size == R.InceptionV3(6).num_keras_params()
  np.testing.assert_array_equal(six[:3 * 3 * 6 * 32].reshape(3, 3, 6, 32),
                                flat[:3 * 3 * 7 * 32].reshape(3, 3, 7, 32)[:, :, :6])
  np.testing.assert_array_equal(six[3 * 3 * 6 * 32:], flat[3 * 3 * 7 * 32:])
  # a permuted checkpoint (two kernels of different shape swapped) is rejected by shape
  bad = dict(tensors)
  kernels = [e[0] for e in entries if e[0].endswith('kernel/.ATTRIBUTES/VARIABLE_VALUE')]
  a, b = kernels[1], kernels[4]
  bad[a], bad[b] = bad[b], bad[a]
  tf_bundle_writer.write_bundle(str(tmp_path / 'bad'), bad)
  with pytest.raises(ValueError, match='shape'):
    cv.import_keras_checkpoint(str(tmp_path / 'bad'), 7)
  old = {'layer_with_weights-0/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE':
         np.zeros((3, 3, 7, 32), np.float32)}
  tf_bundle_writer.write_bundle(str(tmp_path / 'old'), old)
  with pytest.raises(ValueError, match='older DeepVariant Keras model'):
    cv.import_keras_checkpoint(str(tmp_path / 'old'), 7)
