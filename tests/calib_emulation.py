"""torch restatement of the product's shift calibration (deepvariant_amd/csrc/calib.hip) -- test infrastructure.

Two walks of the oracle's graph (oracle/inception_ref.py) with the product's rounding points: R in float32 as
given; E with BN-folded weights rounded to fp16 and every stored fp16 activation rounded to fp16 (the pooled
projections in the product's commuted order: raw 1x1 conv, float32 -> average pool -> + shift -> ReLU -> round; the
last block's outputs stay float32 for the global pool -- round 6, BufferDesc::f32 in csrc/model.hip).
corr[layer][c] = mean_E(pre-activation) - mean_R(pre-activation), applied in E before the next layer.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import inception_ref as R


def _folded(cb):
  inv = 1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS)
  return cb.conv.weight * inv[:, None, None, None], cb.bn.bias - cb.bn.running_mean * inv


class _Walk:
  def __init__(self, ref, mode_e, ref_means=None, precise=False):
    self.ref, self.e, self.ref_means = ref, mode_e, ref_means
    self.precise = precise        # precise mode (> 8 input channels): the 17x17 and 8x8 stages' tensors are hi + lo: not rounded
    self.wide = False
    self.index = {id(cb): i for i, cb in enumerate(ref.convs)}
    self.means = [None] * len(ref.convs)
    self.corr = [None] * len(ref.convs)

  def _r(self, x):
    return x.half().float() if self.e else x

  def _act(self, i, z, keep_f32=False):
    self.means[i] = z.double().mean(dim=(0, 2, 3))
    if self.e:
      self.corr[i] = (self.means[i] - self.ref_means[i]).float()
      z = z - self.corr[i][None, :, None, None]
    return F.relu(z if keep_f32 or self.wide else self._r(z))

  def conv(self, cb, x, keep_f32=False):
    w, shift = _folded(cb)
    if self.e:
      w = w.half().float()
    z = F.conv2d(x, w, None, cb.conv.stride, cb.conv.padding) + shift[None, :, None, None]
    return self._act(self.index[id(cb)], z, keep_f32)

  def pooled_projection(self, cb, x, keep_f32=False):
    w, shift = _folded(cb)
    if self.e:
      w = w.half().float()
    raw = F.conv2d(x, w)            # float32 in LDS / in a float32 tensor: never rounded
    z = F.avg_pool2d(raw, 3, stride=1, padding=1, count_include_pad=False) + shift[None, :, None, None]
    return self._act(self.index[id(cb)], z, keep_f32)

  def seq(self, mods, x):
    for m in mods:
      x = self.conv(m, x)
    return x

  def logits(self, images_u8):
    ref, c = self.ref, self.conv
    x = ((images_u8.float() - 128.0) / 128.0).permute(0, 3, 1, 2).contiguous()
    s = ref.stem
    x = c(s[2], c(s[1], c(s[0], x)))
    x = F.max_pool2d(x, 3, stride=2)
    x = c(s[4], c(s[3], x))
    x = F.max_pool2d(x, 3, stride=2)
    for blk in ref.mixed_a:
      x = torch.cat([self.seq(blk['b1'], x), self.seq(blk['b5'], x), self.seq(blk['b3'], x),
                     self.pooled_projection(blk['bp'][0], x)], 1)
    x = torch.cat([self.seq(ref.mixed3['b3'], x), self.seq(ref.mixed3['b3d'], x), F.max_pool2d(x, 3, stride=2)], 1)
    self.wide = self.precise
    for blk in ref.mixed_b:
      x = torch.cat([self.seq(blk['b1'], x), self.seq(blk['b7'], x), self.seq(blk['b7d'], x),
                     self.pooled_projection(blk['bp'][0], x)], 1)
    x = torch.cat([self.seq(ref.mixed8['b3'], x), self.seq(ref.mixed8['b7'], x), F.max_pool2d(x, 3, stride=2)], 1)
    for blk in ref.mixed_c:
      last = blk is ref.mixed_c[-1]       # its outputs feed the global pool in float32
      b3 = c(blk['b3'][0], x)
      b3 = torch.cat([c(blk['b3'][1], b3, last), c(blk['b3'][2], b3, last)], 1)
      b3d = c(blk['b3d'][1], c(blk['b3d'][0], x))
      b3d = torch.cat([c(blk['b3d'][2], b3d, last), c(blk['b3d'][3], b3d, last)], 1)
      x = torch.cat([c(blk['b1'][0], x, last), b3, b3d, self.pooled_projection(blk['bp'][0], x, last)], 1)
    return ref.classification(x.mean(dim=(2, 3)))


def corrections(ref, images_u8, precise=None):
  """-> float32 array: cout corrections per conv layer in layer order, then the 3 logit corrections.
  `precise`: the product's precise mode (default: as dv_model_create decides -- more than 8 input channels)."""
  if precise is None:
    precise = images_u8.shape[-1] > 8
  with torch.no_grad():
    r = _Walk(ref, False)
    lr = r.logits(images_u8)
    e = _Walk(ref, True, r.means, precise=precise)
    le = e.logits(images_u8)
    dl = (le.double().mean(0) - lr.double().mean(0)).float()
  return np.concatenate([c.numpy() for c in e.corr] + [dl.numpy()])
