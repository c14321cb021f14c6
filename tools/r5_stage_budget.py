import sys, time, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from oracle import inception_ref as R
torch.set_num_threads(8)
kind, seed = sys.argv[1], int(sys.argv[2])
x_u8 = torch.from_numpy(np.load('/root/repo/gpurun_out/r5/images_longread.npz')[kind][:96])
C = x_u8.shape[-1]
ref = R.make_random_model(C, seed=seed)

def fold(cb):
  inv = 1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS)
  return cb.conv.weight * inv[:, None, None, None], cb.bn.bias - cb.bn.running_mean * inv

class Walk:
  def __init__(self, round_stages, round_weights=False):
    self.rs, self.rw = set(round_stages), round_weights
    self.stage = None
  def r(self, x):
    return x.half().float() if self.stage in self.rs else x
  def conv(self, cb, x):
    w, s = fold(cb)
    if self.rw: w = w.half().float()
    return F.relu(self.r(F.conv2d(x, w, None, cb.conv.stride, cb.conv.padding) + s[None, :, None, None]))
  def pooled(self, cb, x):
    w, s = fold(cb)
    if self.rw: w = w.half().float()
    raw = self.r(F.conv2d(x, w))
    return F.relu(self.r(F.avg_pool2d(raw, 3, 1, 1, count_include_pad=False) + s[None, :, None, None]))
  def seq(self, mods, x):
    for m in mods: x = self.conv(m, x)
    return x
  def logits(self, u8):
    c = self.conv
    x = ((u8.float() - 128.0) / 128.0).permute(0, 3, 1, 2).contiguous()
    s = ref.stem
    self.stage = 'stem'
    x = c(s[2], c(s[1], c(s[0], x))); x = F.max_pool2d(x, 3, 2); x = c(s[4], c(s[3], x)); x = F.max_pool2d(x, 3, 2)
    self.stage = 'a'
    for b in ref.mixed_a:
      x = torch.cat([self.seq(b['b1'], x), self.seq(b['b5'], x), self.seq(b['b3'], x), self.pooled(b['bp'][0], x)], 1)
    x = torch.cat([self.seq(ref.mixed3['b3'], x), self.seq(ref.mixed3['b3d'], x), F.max_pool2d(x, 3, 2)], 1)
    self.stage = 'b'
    for b in ref.mixed_b:
      x = torch.cat([self.seq(b['b1'], x), self.seq(b['b7'], x), self.seq(b['b7d'], x), self.pooled(b['bp'][0], x)], 1)
    x = torch.cat([self.seq(ref.mixed8['b3'], x), self.seq(ref.mixed8['b7'], x), F.max_pool2d(x, 3, 2)], 1)
    self.stage = 'c'
    for b in ref.mixed_c:
      b3 = c(b['b3'][0], x); b3 = torch.cat([c(b['b3'][1], b3), c(b['b3'][2], b3)], 1)
      b3d = c(b['b3d'][1], c(b['b3d'][0], x)); b3d = torch.cat([c(b['b3d'][2], b3d), c(b['b3d'][3], b3d)], 1)
      x = torch.cat([self.seq(b['b1'], x), b3, b3d, self.pooled(b['bp'][0], x)], 1)
    return ref.classification(x.mean(dim=(2, 3)))

with torch.no_grad():
  t = time.time()
  base = Walk([]).logits(x_u8).double()
  p0 = torch.softmax(base, 1)
  print('%s seed %d: %d images, fp32 walk %.0f s' % (kind, seed, x_u8.shape[0], time.time() - t), flush=True)
  for name, stages, rw in (('stem', ['stem'], False), ('35x35 (mixed0-3)', ['a'], False), ('17x17 (mixed4-8)', ['b'], False),
                           ('8x8 (mixed9-10)', ['c'], False), ('all activations', ['stem', 'a', 'b', 'c'], False),
                           ('weights only', [], True), ('weights + activations', ['stem', 'a', 'b', 'c'], True)):
    lg = Walk(stages, rw).logits(x_u8).double()
    e = lg - base
    ez = e - e.mean(0, keepdim=True)
    dp = (torch.softmax(lg, 1) - p0).abs().max(1).values
    # zero-mean part of dp: remove the mean logit error before the softmax
    dpz = (torch.softmax(lg - e.mean(0, keepdim=True), 1) - p0).abs().max(1).values
    print('%-24s logit err rms %.3e (zero-mean part %.3e)   |dp| mean %.3e max %.3e   zero-mean |dp| mean %.3e max %.3e' % (
        name, e.pow(2).mean().sqrt(), ez.pow(2).mean().sqrt(), dp.mean(), dp.max(), dpz.mean(), dpz.max()), flush=True)
