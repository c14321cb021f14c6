"""CPU fp32 restatement of the call_variants classifier (the CNN oracle).

TEST INFRASTRUCTURE ONLY -- imported by tests/, smoke() and bench.py's
cpu_baseline leg, never by the product.

What it restates
  deepvariant/keras_modeling.py:246-336  `inceptionv3(...)`:
      tf.keras.applications.InceptionV3(include_top=False, weights=None,
      input_shape=(H, W, C), pooling='avg')  ->  Dropout(0.2)  ->
      Dense(3, activation='softmax', dtype=float32, name='classification')
  deepvariant/dv_utils.py:343-366        `preprocess_images`: (x - 128) / 128

The Inception-v3 graph itself is third-party: tf_keras==2.16.0
`applications/inception_v3.py` (pinned in /root/reference/settings.sh:72-80,
run-prereq.sh:205), absent from /root/reference.  Its published architecture
is restated below: every conv is
Conv2D(use_bias=False) -> BatchNormalization(axis=-1, scale=False,
epsilon=1e-3; inference uses the moving statistics) -> ReLU; 'same' is TF SAME
(symmetric for odd kernels at stride 1); AveragePooling2D 'same' divides by the
number of un-padded cells.

PARITY UNPINNED for numerics: no reference test fixes a logit or probability
(deepvariant/call_variants_test.py:109-181 uses random weights and checks
record counts; keras_modeling_test.py:56-76 checks shape/range/sum only), and
no trained checkpoint ships in the tree.  What IS checked offline:
parameter count 21,808,931 at C=3 (= Keras' 21,802,784 backbone + 6,147 head),
21,810,083 at C=7; 94 conv layers; feature-map trace
100x221 -> 49x110 -> 47x108 -> 23x53 -> 21x51 -> 10x25 -> 4x12 -> 1x5 -> 2048.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

BN_EPS = 1e-3


class ConvBN(nn.Module):
  """conv2d_bn of tf_keras inception_v3.py."""

  def __init__(self, cin, cout, kh, kw, stride=1, padding='same'):
    super().__init__()
    pad = ((kh - 1) // 2, (kw - 1) // 2) if padding == 'same' else (0, 0)
    self.conv = nn.Conv2d(cin, cout, (kh, kw), stride=stride, padding=pad,
                          bias=False)
    self.bn = nn.BatchNorm2d(cout, eps=BN_EPS, affine=True)
    # scale=False: gamma is fixed at 1 and not a parameter in Keras.
    self.bn.weight.requires_grad_(False)
    nn.init.ones_(self.bn.weight)

  def forward(self, x):
    if ConvBN.as_gemm:
      return self._forward_gemm(x)
    return F.relu(self.bn(self.conv(x)))

  # The same fp32 arithmetic written as im2col + matmul and an explicit BN: what the large-N
  # tests run on the GPU (tests/cnn_tail.py).  torch's conv2d on ROCm goes through MIOpen,
  # which compiles its solvers on first use on a fresh box (minutes); unfold / matmul / pooling
  # are precompiled ATen + rocBLAS kernels.  Checked against forward() above by
  # tests/test_cnn_oracle_gemm_cpu.py (CPU) and tests/test_hip_cnn_tail.py (GPU vs CPU).
  as_gemm = False

  def _forward_gemm(self, x):
    n, c, h, w = x.shape
    co, ci, kh, kw = self.conv.weight.shape
    sh, sw = self.conv.stride
    ph, pw = self.conv.padding
    oh = (h + 2 * ph - kh) // sh + 1
    ow = (w + 2 * pw - kw) // sw + 1
    if kh == 1 and kw == 1 and sh == 1 and sw == 1:
      cols = x.reshape(n, c, h * w)
    else:
      cols = F.unfold(x, (kh, kw), padding=(ph, pw), stride=(sh, sw))
    y = torch.matmul(self.conv.weight.reshape(co, ci * kh * kw), cols).reshape(n, co, oh, ow)
    inv = 1.0 / torch.sqrt(self.bn.running_var + self.bn.eps)
    y = (y - self.bn.running_mean[None, :, None, None]) * (inv * self.bn.weight)[None, :, None, None] \
        + self.bn.bias[None, :, None, None]
    return F.relu(y)


def _avgpool(x):
  return F.avg_pool2d(x, 3, stride=1, padding=1, count_include_pad=False)


def _maxpool(x):
  return F.max_pool2d(x, 3, stride=2)


class InceptionV3(nn.Module):
  """Layers are registered in construction order = the order
  dv_model_load_weights (include/dvhip.h) expects them."""

  def __init__(self, in_channels: int, num_classes: int = 3):
    super().__init__()
    self.convs = nn.ModuleList()
    c = self._add

    self.stem = [c(in_channels, 32, 3, 3, 2, 'valid'), c(32, 32, 3, 3, 1, 'valid'),
                 c(32, 64, 3, 3), c(64, 80, 1, 1, 1, 'valid'),
                 c(80, 192, 3, 3, 1, 'valid')]
    self.mixed_a = []
    cin = 192
    for pool_ch in (32, 64, 64):  # mixed0..2
      self.mixed_a.append(dict(
          b1=[c(cin, 64, 1, 1)],
          b5=[c(cin, 48, 1, 1), c(48, 64, 5, 5)],
          b3=[c(cin, 64, 1, 1), c(64, 96, 3, 3), c(96, 96, 3, 3)],
          bp=[c(cin, pool_ch, 1, 1)]))
      cin = 64 + 64 + 96 + pool_ch
    self.mixed3 = dict(
        b3=[c(cin, 384, 3, 3, 2, 'valid')],
        b3d=[c(cin, 64, 1, 1), c(64, 96, 3, 3), c(96, 96, 3, 3, 2, 'valid')])
    cin = 384 + 96 + cin  # 768
    self.mixed_b = []
    for c7 in (128, 160, 160, 192):  # mixed4..7
      self.mixed_b.append(dict(
          b1=[c(cin, 192, 1, 1)],
          b7=[c(cin, c7, 1, 1), c(c7, c7, 1, 7), c(c7, 192, 7, 1)],
          b7d=[c(cin, c7, 1, 1), c(c7, c7, 7, 1), c(c7, c7, 1, 7),
               c(c7, c7, 7, 1), c(c7, 192, 1, 7)],
          bp=[c(cin, 192, 1, 1)]))
      cin = 768
    self.mixed8 = dict(
        b3=[c(cin, 192, 1, 1), c(192, 320, 3, 3, 2, 'valid')],
        b7=[c(cin, 192, 1, 1), c(192, 192, 1, 7), c(192, 192, 7, 1),
            c(192, 192, 3, 3, 2, 'valid')])
    cin = 320 + 192 + cin  # 1280
    self.mixed_c = []
    for _ in range(2):  # mixed9, mixed10
      self.mixed_c.append(dict(
          b1=[c(cin, 320, 1, 1)],
          b3=[c(cin, 384, 1, 1), c(384, 384, 1, 3), c(384, 384, 3, 1)],
          b3d=[c(cin, 448, 1, 1), c(448, 384, 3, 3), c(384, 384, 1, 3),
               c(384, 384, 3, 1)],
          bp=[c(cin, 192, 1, 1)]))
      cin = 2048
    self.classification = nn.Linear(2048, num_classes)

  def _add(self, cin, cout, kh, kw, stride=1, padding='same'):
    m = ConvBN(cin, cout, kh, kw, stride, padding)
    self.convs.append(m)
    return m

  @staticmethod
  def _seq(mods, x):
    for m in mods:
      x = m(x)
    return x

  def features(self, x):
    s = self.stem
    x = s[2](s[1](s[0](x)))
    x = _maxpool(x)
    x = s[4](s[3](x))
    x = _maxpool(x)
    for blk in self.mixed_a:
      x = torch.cat([self._seq(blk['b1'], x), self._seq(blk['b5'], x),
                     self._seq(blk['b3'], x),
                     self._seq(blk['bp'], _avgpool(x))], 1)
    x = torch.cat([self._seq(self.mixed3['b3'], x),
                   self._seq(self.mixed3['b3d'], x), _maxpool(x)], 1)
    for blk in self.mixed_b:
      x = torch.cat([self._seq(blk['b1'], x), self._seq(blk['b7'], x),
                     self._seq(blk['b7d'], x),
                     self._seq(blk['bp'], _avgpool(x))], 1)
    x = torch.cat([self._seq(self.mixed8['b3'], x),
                   self._seq(self.mixed8['b7'], x), _maxpool(x)], 1)
    for blk in self.mixed_c:
      b3 = blk['b3'][0](x)
      b3 = torch.cat([blk['b3'][1](b3), blk['b3'][2](b3)], 1)
      b3d = blk['b3d'][1](blk['b3d'][0](x))
      b3d = torch.cat([blk['b3d'][2](b3d), blk['b3d'][3](b3d)], 1)
      x = torch.cat([self._seq(blk['b1'], x), b3, b3d,
                     self._seq(blk['bp'], _avgpool(x))], 1)
    return x.mean(dim=(2, 3))  # GlobalAveragePooling2D

  def forward(self, images_u8_nhwc: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
    """uint8 [N,H,W,C] -> softmax probabilities fp32 [N,3].  `channels_last` only changes
    the memory format torch's CPU kernels work in (a speed knob of the cpu_baseline leg)."""
    x = images_u8_nhwc.to(torch.float32)
    x = (x - 128.0) / 128.0  # dv_utils.preprocess_images
    x = x.permute(0, 3, 1, 2)
    x = x.contiguous(memory_format=torch.channels_last) if channels_last else x.contiguous()
    logits = self.classification(self.features(x))  # Dropout = identity
    return torch.softmax(logits, dim=1)

  # ---- weights in the flat layout of dv_model_load_weights ----------------
  def export_flat(self) -> np.ndarray:
    """conv kernel HWIO, then BN beta, moving_mean, moving_variance per conv;
    finally Dense kernel [2048, classes] and bias."""
    parts = []
    for m in self.convs:
      w = m.conv.weight.detach().permute(2, 3, 1, 0).contiguous()  # OIHW->HWIO
      parts += [w.reshape(-1), m.bn.bias.detach(), m.bn.running_mean,
                m.bn.running_var]
    parts += [self.classification.weight.detach().t().contiguous().reshape(-1),
              self.classification.bias.detach()]
    return torch.cat([p.reshape(-1).float() for p in parts]).numpy()

  def load_flat(self, flat: np.ndarray) -> None:
    """Inverse of export_flat (same layout as dv_model_load_weights)."""
    flat = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32))
    off = 0
    with torch.no_grad():
      for m in self.convs:
        co, ci, kh, kw = m.conv.weight.shape
        n = co * ci * kh * kw
        m.conv.weight.copy_(flat[off:off + n].reshape(kh, kw, ci, co).permute(3, 2, 0, 1))
        off += n
        m.bn.bias.copy_(flat[off:off + co])
        m.bn.running_mean.copy_(flat[off + co:off + 2 * co])
        m.bn.running_var.copy_(flat[off + 2 * co:off + 3 * co])
        off += 3 * co
      k, c = self.classification.weight.shape
      self.classification.weight.copy_(flat[off:off + k * c].reshape(c, k).t())
      off += k * c
      self.classification.bias.copy_(flat[off:off + k])
      off += k
    assert off == flat.numel(), (off, flat.numel())
    self.eval()

  def num_keras_params(self) -> int:
    n = 0
    for m in self.convs:
      n += m.conv.weight.numel() + 3 * m.bn.bias.numel()
    return n + self.classification.weight.numel() + self.classification.bias.numel()


def make_random_model(in_channels: int, seed: int = 0) -> InceptionV3:
  """Seeded weights, like the reference's own call_variants_test.py:109-127
  ("a model with random weights"): He-normal kernels, BN statistics randomised
  with positive variance so the folded scale/shift is non-trivial."""
  g = torch.Generator().manual_seed(seed)
  m = InceptionV3(in_channels)
  with torch.no_grad():
    for cb in m.convs:
      fan_in = cb.conv.weight[0].numel()
      cb.conv.weight.copy_(torch.randn(cb.conv.weight.shape, generator=g) *
                           (2.0 / fan_in) ** 0.5)
      n = cb.bn.bias.numel()
      cb.bn.bias.copy_(torch.randn(n, generator=g) * 0.1)
      cb.bn.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
      cb.bn.running_var.copy_(torch.rand(n, generator=g) + 0.5)
    m.classification.weight.copy_(
        torch.randn(m.classification.weight.shape, generator=g) * 0.05)
    m.classification.bias.copy_(torch.randn(3, generator=g) * 0.1)
  m.eval()
  return m


def conv_layer_table(in_channels: int, h: int, w: int
                     ) -> List[Tuple[int, int, int, int, int, int]]:
  """(kh, kw, cin, cout, oh, ow) per conv, and MACs, for FLOP accounting."""
  m = InceptionV3(in_channels)
  shapes = []
  hooks = []
  for cb in m.convs:
    def hook(mod, inp, out, cb=cb):
      shapes.append((cb.conv.kernel_size[0], cb.conv.kernel_size[1],
                     cb.conv.in_channels, cb.conv.out_channels,
                     out.shape[2], out.shape[3]))
    hooks.append(cb.conv.register_forward_hook(hook))
  m.eval()
  with torch.no_grad():
    m(torch.zeros(1, h, w, in_channels, dtype=torch.uint8))
  for hk in hooks:
    hk.remove()
  return shapes


def macs_per_example(in_channels: int, h: int, w: int) -> int:
  total = 0
  for kh, kw, cin, cout, oh, ow in conv_layer_table(in_channels, h, w):
    total += kh * kw * cin * cout * oh * ow
  return total + 2048 * 3
