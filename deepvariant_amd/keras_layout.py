"""Which checkpoint variable holds which layer of the classifier.

call_variants restores `keras_modeling.inceptionv3(...)` with `model.load_weights`
(deepvariant/call_variants.py:759-762, deepvariant/keras_modeling.py:246-336).  That model
is FLAT: `tf.keras.Model(inputs=backbone.input, outputs=head(backbone.output))`, so its
object-graph checkpoint names every layer that owns variables
`layer_with_weights-<N>/<attr>/.ATTRIBUTES/VARIABLE_VALUE`
(deepvariant/keras_modeling.py:176-184 relies on `layer_with_weights-0/kernel` being the
first convolution), where N counts such layers in `model.layers` order.

`model.layers` of a functional Keras model is NOT construction order: tf_keras (2.16,
engine/functional.py `_map_graph_network` / `_build_map`) sorts layers by DEPTH (longest
path to an output, deepest first) and breaks ties by the order a depth-first walk from the
outputs first reaches them.  Parallel Inception branches therefore interleave.  This module
restates that ordering on the InceptionV3 graph (tf_keras applications/inception_v3.py,
SURVEY.md App. B) and maps every variable to its slot in the flat weight layout of
`dv_model_load_weights` (construction order: conv kernel HWIO, BN beta / moving_mean /
moving_variance per conv, then Dense kernel + bias).
"""
from __future__ import annotations

from typing import Dict, List, Tuple


class _Layer:
  def __init__(self, name: str, kind: str, inputs: List['_Layer'], conv_index: int = -1):
    self.name = name
    self.kind = kind              # input | conv | bn | act | maxpool | avgpool | concat | gap | dropout | dense
    self.inputs = inputs
    self.conv_index = conv_index  # construction index of the conv / its BN


class _Graph:
  """Builds the layer graph in tf_keras' construction order."""

  def __init__(self):
    self.layers: List[_Layer] = []
    self.n_conv = 0
    self.conv_shapes: List[Tuple[int, int, int, int]] = []

  def add(self, kind, inputs, conv_index=-1, name=None):
    layer = _Layer(name or '%s_%d' % (kind, len(self.layers)), kind, list(inputs), conv_index)
    self.layers.append(layer)
    return layer

  def conv_bn(self, x, cin, cout, kh, kw):
    i = self.n_conv
    self.n_conv += 1
    self.conv_shapes.append((kh, kw, cin, cout))
    c = self.add('conv', [x], i)
    b = self.add('bn', [c], i)
    return self.add('act', [b])


def build_graph(in_channels: int) -> Tuple[_Graph, _Layer]:
  g = _Graph()
  cb = g.conv_bn
  x = g.add('input', [])
  x = cb(x, in_channels, 32, 3, 3)
  x = cb(x, 32, 32, 3, 3)
  x = cb(x, 32, 64, 3, 3)
  x = g.add('maxpool', [x])
  x = cb(x, 64, 80, 1, 1)
  x = cb(x, 80, 192, 3, 3)
  x = g.add('maxpool', [x])
  cin = 192
  for pool_ch in (32, 64, 64):                      # mixed0..2
    b1 = cb(x, cin, 64, 1, 1)
    b5 = cb(cb(x, cin, 48, 1, 1), 48, 64, 5, 5)
    b3 = cb(cb(cb(x, cin, 64, 1, 1), 64, 96, 3, 3), 96, 96, 3, 3)
    bp = cb(g.add('avgpool', [x]), cin, pool_ch, 1, 1)
    x = g.add('concat', [b1, b5, b3, bp])
    cin = 64 + 64 + 96 + pool_ch
  b3 = cb(x, cin, 384, 3, 3)                         # mixed3
  bd = cb(cb(cb(x, cin, 64, 1, 1), 64, 96, 3, 3), 96, 96, 3, 3)
  x = g.add('concat', [b3, bd, g.add('maxpool', [x])])
  cin = 768
  for c7 in (128, 160, 160, 192):                   # mixed4..7
    b1 = cb(x, cin, 192, 1, 1)
    b7 = cb(cb(cb(x, cin, c7, 1, 1), c7, c7, 1, 7), c7, 192, 7, 1)
    d = cb(x, cin, c7, 1, 1)
    d = cb(d, c7, c7, 7, 1)
    d = cb(d, c7, c7, 1, 7)
    d = cb(d, c7, c7, 7, 1)
    d = cb(d, c7, 192, 1, 7)
    bp = cb(g.add('avgpool', [x]), cin, 192, 1, 1)
    x = g.add('concat', [b1, b7, d, bp])
  b3 = cb(cb(x, cin, 192, 1, 1), 192, 320, 3, 3)     # mixed8
  b7 = cb(cb(cb(cb(x, cin, 192, 1, 1), 192, 192, 1, 7), 192, 192, 7, 1), 192, 192, 3, 3)
  x = g.add('concat', [b3, b7, g.add('maxpool', [x])])
  cin = 1280
  for _ in range(2):                                # mixed9, mixed10
    b1 = cb(x, cin, 320, 1, 1)
    t = cb(x, cin, 384, 1, 1)
    b3 = g.add('concat', [cb(t, 384, 384, 1, 3), cb(t, 384, 384, 3, 1)])
    d = cb(cb(x, cin, 448, 1, 1), 448, 384, 3, 3)
    bd = g.add('concat', [cb(d, 384, 384, 1, 3), cb(d, 384, 384, 3, 1)])
    bp = cb(g.add('avgpool', [x]), cin, 192, 1, 1)
    x = g.add('concat', [b1, b3, bd, bp])
    cin = 2048
  x = g.add('gap', [x])
  x = g.add('dropout', [x])
  out = g.add('dense', [x], name='classification')
  return g, out


def keras_layer_order(output: _Layer) -> List[_Layer]:
  """`model.layers` as tf_keras computes it (every layer is called once here, so a node is
  a layer)."""
  order: Dict[_Layer, int] = {}
  post: List[_Layer] = []
  done = set()

  def walk(layer):                 # _build_map_helper: index on first visit, inputs in order
    if layer in done:
      return
    if layer not in order:
      order[layer] = len(order)
    stack = [(layer, iter(layer.inputs))]
    while stack:
      cur, it = stack[-1]
      nxt = next(it, None)
      if nxt is None:
        stack.pop()
        if cur not in done:
          done.add(cur)
          post.append(cur)
        continue
      if nxt in done:
        continue
      if nxt not in order:
        order[nxt] = len(order)
      stack.append((nxt, iter(nxt.inputs)))

  walk(output)
  depth: Dict[_Layer, int] = {}
  for layer in reversed(post):     # outputs first: depth = longest path to the output
    d = depth.setdefault(layer, 0)
    for parent in layer.inputs:
      depth[parent] = max(d + 1, depth.get(parent, 0))
  return sorted(post, key=lambda l: (-depth[l], order[l]))


def variable_layout(in_channels: int, num_classes: int = 3):
  """-> (entries, n_params): entries = [(checkpoint variable name, shape, offset into the flat
  weight array)], in checkpoint (`layer_with_weights-N`) order."""
  g, out = build_graph(in_channels)
  # offsets in construction order (dv_model_load_weights)
  conv_off = []
  off = 0
  for kh, kw, ci, co in g.conv_shapes:
    conv_off.append(off)
    off += kh * kw * ci * co + 3 * co
  dense_off = off
  n_params = off + 2048 * num_classes + num_classes
  entries = []
  n = 0
  suffix = '/.ATTRIBUTES/VARIABLE_VALUE'
  for layer in keras_layer_order(out):
    if layer.kind == 'conv':
      kh, kw, ci, co = g.conv_shapes[layer.conv_index]
      entries.append(('layer_with_weights-%d/kernel%s' % (n, suffix), (kh, kw, ci, co),
                      conv_off[layer.conv_index]))
      n += 1
    elif layer.kind == 'bn':
      kh, kw, ci, co = g.conv_shapes[layer.conv_index]
      base = conv_off[layer.conv_index] + kh * kw * ci * co
      for k, attr in enumerate(('beta', 'moving_mean', 'moving_variance')):
        entries.append(('layer_with_weights-%d/%s%s' % (n, attr, suffix), (co,), base + k * co))
      n += 1
    elif layer.kind == 'dense':
      entries.append(('layer_with_weights-%d/kernel%s' % (n, suffix), (2048, num_classes), dense_off))
      entries.append(('layer_with_weights-%d/bias%s' % (n, suffix), (num_classes,),
                      dense_off + 2048 * num_classes))
      n += 1
  return entries, n_params
