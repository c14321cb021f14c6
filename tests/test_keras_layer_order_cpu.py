"""A second, independent derivation of the checkpoint's `layer_with_weights-N` order.

`deepvariant_amd/keras_layout.py` restates tf_keras' `model.layers` ordering on its own graph
objects; nothing in this image can run Keras, and dozens of InceptionV3 convolutions share a
shape, so a shape check alone cannot catch a swap.  This file derives the order a second time
from different ingredients:

  * an explicit NODE LIST of the model by Keras layer NAME (conv2d_17, batch_normalization_17,
    activation_17, average_pooling2d_2, mixed5, mixed9_0, concatenate_1, ...), transcribed from
    tf_keras/applications/inception_v3.py (InceptionV3(include_top=False, pooling='avg')) plus
    DeepVariant's head (deepvariant/keras_modeling.py:113-193,246-336: Dropout -> Dense
    'classification');
  * a direct, recursive transcription of tf_keras engine/functional.py `_build_map_helper` /
    `_map_graph_network` (nodes in post-order, node depth = longest path to an output, layers of
    one depth in first-visit order) -- keras_layout uses an explicit stack and other data
    structures;
  * the layer sequences every public `model.summary()` of Keras' InceptionV3 prints (summary rows
    are `model.layers`), hard-coded below for the stem, mixed0, mixed3, mixed4 and mixed9.

Both derivations must give the same sequence of variable-owning layers, and that sequence must
agree with keras_layout.variable_layout name for name, shape for shape, offset for offset.
"""
from deepvariant_amd import keras_layout


class _Net:
  """Layers by name: kind, input layer names (call order), conv geometry."""

  def __init__(self, channels):
    self.inputs = {}           # name -> [input names]
    self.kind = {}
    self.conv = {}             # conv / bn name -> (kh, kw, cin, cout)
    self.n = {'conv2d': 0, 'max_pooling2d': 0, 'average_pooling2d': 0, 'concatenate': 0}
    self.width = {}            # name -> channels
    self.add('input_1', 'input', [], channels)

  def add(self, name, kind, inputs, width):
    assert name not in self.inputs
    self.inputs[name] = list(inputs)
    self.kind[name] = kind
    self.width[name] = width
    return name

  def _numbered(self, base):
    k = self.n[base]
    self.n[base] += 1
    return base if k == 0 else '%s_%d' % (base, k), k

  def conv2d_bn(self, x, filters, kh, kw):
    name, k = self._numbered('conv2d')
    suffix = '' if k == 0 else '_%d' % k
    geom = (kh, kw, self.width[x], filters)
    self.conv[name] = geom
    self.add(name, 'conv', [x], filters)
    bn = self.add('batch_normalization' + suffix, 'bn', [name], filters)
    self.conv[bn] = geom
    return self.add('activation' + suffix, 'act', [bn], filters)

  def pool(self, base, x):
    name, _ = self._numbered(base)
    return self.add(name, base, [x], self.width[x])

  def concat(self, xs, name=None):
    if name is None:
      name, _ = self._numbered('concatenate')
    return self.add(name, 'concat', xs, sum(self.width[x] for x in xs))


def inception_v3_by_name(channels):
  """tf_keras/applications/inception_v3.py, statement by statement."""
  n = _Net(channels)
  c = n.conv2d_bn
  x = c('input_1', 32, 3, 3)
  x = c(x, 32, 3, 3)
  x = c(x, 64, 3, 3)
  x = n.pool('max_pooling2d', x)
  x = c(x, 80, 1, 1)
  x = c(x, 192, 3, 3)
  x = n.pool('max_pooling2d', x)
  for i, pool_filters in enumerate((32, 64, 64)):           # mixed 0, 1, 2: 35 x 35
    branch1x1 = c(x, 64, 1, 1)
    branch5x5 = c(x, 48, 1, 1)
    branch5x5 = c(branch5x5, 64, 5, 5)
    branch3x3dbl = c(x, 64, 1, 1)
    branch3x3dbl = c(branch3x3dbl, 96, 3, 3)
    branch3x3dbl = c(branch3x3dbl, 96, 3, 3)
    branch_pool = n.pool('average_pooling2d', x)
    branch_pool = c(branch_pool, pool_filters, 1, 1)
    x = n.concat([branch1x1, branch5x5, branch3x3dbl, branch_pool], 'mixed%d' % i)
  branch3x3 = c(x, 384, 3, 3)                               # mixed 3: 17 x 17
  branch3x3dbl = c(x, 64, 1, 1)
  branch3x3dbl = c(branch3x3dbl, 96, 3, 3)
  branch3x3dbl = c(branch3x3dbl, 96, 3, 3)
  branch_pool = n.pool('max_pooling2d', x)
  x = n.concat([branch3x3, branch3x3dbl, branch_pool], 'mixed3')
  for i, c7 in enumerate((128, 160, 160, 192)):              # mixed 4 .. 7
    branch1x1 = c(x, 192, 1, 1)
    branch7x7 = c(x, c7, 1, 1)
    branch7x7 = c(branch7x7, c7, 1, 7)
    branch7x7 = c(branch7x7, 192, 7, 1)
    branch7x7dbl = c(x, c7, 1, 1)
    branch7x7dbl = c(branch7x7dbl, c7, 7, 1)
    branch7x7dbl = c(branch7x7dbl, c7, 1, 7)
    branch7x7dbl = c(branch7x7dbl, c7, 7, 1)
    branch7x7dbl = c(branch7x7dbl, 192, 1, 7)
    branch_pool = n.pool('average_pooling2d', x)
    branch_pool = c(branch_pool, 192, 1, 1)
    x = n.concat([branch1x1, branch7x7, branch7x7dbl, branch_pool], 'mixed%d' % (4 + i))
  branch3x3 = c(x, 192, 1, 1)                               # mixed 8: 8 x 8
  branch3x3 = c(branch3x3, 320, 3, 3)
  branch7x7x3 = c(x, 192, 1, 1)
  branch7x7x3 = c(branch7x7x3, 192, 1, 7)
  branch7x7x3 = c(branch7x7x3, 192, 7, 1)
  branch7x7x3 = c(branch7x7x3, 192, 3, 3)
  branch_pool = n.pool('max_pooling2d', x)
  x = n.concat([branch3x3, branch7x7x3, branch_pool], 'mixed8')
  for i in range(2):                                         # mixed 9, 10
    branch1x1 = c(x, 320, 1, 1)
    branch3x3 = c(x, 384, 1, 1)
    branch3x3_1 = c(branch3x3, 384, 1, 3)
    branch3x3_2 = c(branch3x3, 384, 3, 1)
    branch3x3 = n.concat([branch3x3_1, branch3x3_2], 'mixed9_%d' % i)
    branch3x3dbl = c(x, 448, 1, 1)
    branch3x3dbl = c(branch3x3dbl, 384, 3, 3)
    branch3x3dbl_1 = c(branch3x3dbl, 384, 1, 3)
    branch3x3dbl_2 = c(branch3x3dbl, 384, 3, 1)
    branch3x3dbl = n.concat([branch3x3dbl_1, branch3x3dbl_2])            # 'concatenate', 'concatenate_1'
    branch_pool = n.pool('average_pooling2d', x)
    branch_pool = c(branch_pool, 192, 1, 1)
    x = n.concat([branch1x1, branch3x3, branch3x3dbl, branch_pool], 'mixed%d' % (9 + i))
  x = n.add('global_average_pooling2d', 'gap', [x], n.width[x])           # pooling='avg'
  x = n.add('dropout', 'dropout', [x], n.width[x])                         # keras_modeling.py:283
  n.add('classification', 'dense', [x], 3)                                 # build_classification_head
  return n, 'classification'


def keras_model_layers(net, output):
  """engine/functional.py: _build_map (post-order over keras_inputs, first-visit index per layer)
  then _map_graph_network (depths from the outputs, layers by depth then by first visit).  Every
  layer of this model is called once, so node == layer."""
  finished, in_progress, post_order, first_visit = set(), set(), [], {}

  def build_map_helper(layer):
    if layer in finished:
      return
    assert layer not in in_progress, 'cycle'
    if layer not in first_visit:
      first_visit[layer] = len(first_visit)
    in_progress.add(layer)
    for parent in net.inputs[layer]:
      build_map_helper(parent)
    finished.add(layer)
    in_progress.remove(layer)
    post_order.append(layer)

  import sys
  limit = sys.getrecursionlimit()
  sys.setrecursionlimit(10000)
  try:
    build_map_helper(output)
  finally:
    sys.setrecursionlimit(limit)
  depths = {}
  for layer in reversed(post_order):
    depth = depths.setdefault(layer, 0)
    for parent in net.inputs[layer]:
      depths[parent] = max(depth + 1, depths.get(parent, 0))
  by_depth = {}
  for layer, depth in depths.items():
    by_depth.setdefault(depth, []).append(layer)
  layers = []
  for depth in sorted(by_depth, reverse=True):
    layers.extend(sorted(by_depth[depth], key=lambda l: first_visit[l]))
  return layers


def _triple(ks):
  """conv2d_bn layers of equal depth appear as convs, then BNs, then activations."""
  def nm(base, k):
    return base if k == 0 else '%s_%d' % (base, k)
  return ([nm('conv2d', k) for k in ks] + [nm('batch_normalization', k) for k in ks] +
          [nm('activation', k) for k in ks])


def test_order_matches_the_public_model_summary():
  """Rows of `tf.keras.applications.InceptionV3().summary()` (= model.layers) as published in
  countless notebooks: the stem in construction order; inside a block the deepest branch first."""
  net, out = inception_v3_by_name(3)
  layers = keras_model_layers(net, out)
  assert len(layers) == 311 + 3      # Keras' 311 layers of InceptionV3(include_top=False) + pooling + Dropout + Dense
  stem = (['input_1'] + _triple([0]) + _triple([1]) + _triple([2]) + ['max_pooling2d'] + _triple([3]) +
          _triple([4]) + ['max_pooling2d_1'])
  mixed0 = (_triple([8]) + _triple([6, 9]) + ['average_pooling2d'] + ['conv2d_5', 'conv2d_7', 'conv2d_10', 'conv2d_11'] +
            ['batch_normalization_%d' % k for k in (5, 7, 10, 11)] + ['activation_%d' % k for k in (5, 7, 10, 11)] +
            ['mixed0'])
  assert layers[:len(stem)] == stem
  assert layers[len(stem):len(stem) + len(mixed0)] == mixed0
  i3 = layers.index('conv2d_27')
  mixed3 = (_triple([27]) + _triple([28]) + ['conv2d_26', 'conv2d_29', 'batch_normalization_26',
                                             'batch_normalization_29', 'activation_26', 'activation_29',
                                             'max_pooling2d_2', 'mixed3'])
  assert layers[i3:i3 + len(mixed3)] == mixed3
  mixed4 = (_triple([34]) + _triple([35]) + _triple([31, 36]) + _triple([32, 37]) + ['average_pooling2d_3'] +
            ['conv2d_30', 'conv2d_33', 'conv2d_38', 'conv2d_39'] +
            ['batch_normalization_%d' % k for k in (30, 33, 38, 39)] + ['activation_%d' % k for k in (30, 33, 38, 39)] +
            ['mixed4'])
  i4 = layers.index('conv2d_34')
  assert layers[i4:i4 + len(mixed4)] == mixed4
  mixed9 = (_triple([80]) + _triple([77, 81]) + ['conv2d_78', 'conv2d_79', 'conv2d_82', 'conv2d_83', 'average_pooling2d_7'] +
            ['conv2d_76'] + ['batch_normalization_%d' % k for k in (78, 79, 82, 83)] + ['conv2d_84'] +
            ['batch_normalization_76'] + ['activation_%d' % k for k in (78, 79, 82, 83)] + ['batch_normalization_84'] +
            ['activation_76', 'mixed9_0', 'concatenate', 'activation_84', 'mixed9'])
  i9 = layers.index('conv2d_80')
  assert layers[i9:i9 + len(mixed9)] == mixed9
  assert layers[-4:] == ['mixed10', 'global_average_pooling2d', 'dropout', 'classification']


def test_second_derivation_agrees_with_keras_layout():
  for channels in (3, 6, 7, 10):
    net, out = inception_v3_by_name(channels)
    layers = keras_model_layers(net, out)
    owning = [l for l in layers if net.kind[l] in ('conv', 'bn', 'dense')]
    assert len(owning) == 189
    entries, n_params = keras_layout.variable_layout(channels)
    # construction-order offsets, computed here from the names alone
    conv_names = sorted((l for l in net.kind if net.kind[l] == 'conv'),
                        key=lambda s: int(s.split('_')[1]) if '_' in s else 0)
    assert len(conv_names) == 94
    offset, off = {}, 0
    for name in conv_names:
      kh, kw, ci, co = net.conv[name]
      offset[name] = off
      off += kh * kw * ci * co + 3 * co
    want = []
    for idx, layer in enumerate(owning):
      prefix = 'layer_with_weights-%d/' % idx
      if net.kind[layer] == 'conv':
        want.append((prefix + 'kernel', net.conv[layer], offset[layer]))
      elif net.kind[layer] == 'bn':
        conv = layer.replace('batch_normalization', 'conv2d')
        kh, kw, ci, co = net.conv[conv]
        base = offset[conv] + kh * kw * ci * co
        for k, attr in enumerate(('beta', 'moving_mean', 'moving_variance')):
          want.append((prefix + attr, (co,), base + k * co))
      else:
        want.append((prefix + 'kernel', (2048, 3), off))
        want.append((prefix + 'bias', (3,), off + 2048 * 3))
    assert n_params == off + 2048 * 3 + 3
    got = [(name.replace('/.ATTRIBUTES/VARIABLE_VALUE', ''), tuple(shape), o) for name, shape, o in entries]
    assert got == want


def test_every_batch_norm_follows_its_own_convolution():
  """Pairing checks that catch swaps between layers of different shape: in checkpoint order, the
  BN that owns `layer_with_weights-N` normalises the convolution whose kernel it is stored next
  to in the flat layout; each conv's input width is the width of the tensor it reads."""
  entries, _ = keras_layout.variable_layout(7)
  by_offset = {o: (name, shape) for name, shape, o in entries}
  kernels = [(name, shape, o) for name, shape, o in entries if name.split('/')[1] == 'kernel' and len(shape) == 4]
  assert len(kernels) == 94
  for name, (kh, kw, ci, co), o in kernels:
    end = o + kh * kw * ci * co
    for k, attr in enumerate(('beta', 'moving_mean', 'moving_variance')):
      bn_name, bn_shape = by_offset[end + k * co]
      assert bn_name.split('/')[1] == attr and bn_shape == (co,)
    # the three statistics belong to ONE layer_with_weights index
    assert len({by_offset[end + k * co][0].split('/')[0] for k in range(3)}) == 1
  net, out = inception_v3_by_name(7)
  for layer, (kh, kw, ci, co) in net.conv.items():
    if net.kind[layer] == 'conv':
      assert ci == net.width[net.inputs[layer][0]]
