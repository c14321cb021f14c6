"""Per stored tensor: how much of the classifier's fp16 error enters at each rounding point, and what keeping a set of
tensors wider than fp16 would buy (GPU box; test infrastructure).

  python tools/r6_tensor_budget.py --workload ont --seed 202 --n 1024 > profiles/r06_tensor_budget_ont202.txt

Uses dv_model_probe_rounding (include/dvhip.h): the calibration's own two fp32 pipelines (R exact, E with the MFMA
kernels' rounding points), so the numbers are the PRODUCT's arithmetic, not a CPU emulation.
  1. check: E (product rounding points, product calibration) against the HIP kernels themselves;
  2. attribution: E with fp32 weights and ONE op's output rounded to fp16 -- its share of the zero-mean logit error
     (variances of independent roundings add; the sum is compared with the all-rounded run);
  3. scenarios: sets of tensors kept in fp32 (= what hi + lo storage would give), each with its own calibration on
     OTHER images, |dp| against the fp32 pipeline R on the evaluation images.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from tests import cnn_tail as T   # noqa: E402
from oracle import inception_ref as R   # noqa: E402
from deepvariant_amd import _lib   # noqa: E402
from deepvariant_amd.inception_v3 import InceptionV3   # noqa: E402

SHAPES = {'illumina': (100, 221, 7), 'hifi': (100, 147, 10), 'ont': (100, 199, 9)}


def softmax(z):
  z = z.astype(np.float64)
  z = z - z.max(1, keepdims=True)
  e = np.exp(z)
  return e / e.sum(1, keepdims=True)


class Probe:
  def __init__(self, model, flat):
    self.m, self.flat, self.lib = model, flat, _lib.lib()
    self.n_ops = self.lib.dv_model_num_ops(model._handle)
    self.n_corr = sum(co for _, _, _, co, _ in model.layer_table())
    self.labels = []
    buf = C.create_string_buffer(256)
    for i in range(self.n_ops):
      _lib.check(self.lib.dv_model_op_label(model._handle, i, buf, 256))
      self.labels.append(dict(kv.split('=') for kv in buf.value.decode().split()[1:]) | {'kind': buf.value.decode().split()[0]})

  def run(self, images, keep=None, corr=None, weights_f32=False, want_r=False, measure=False, chunk=1024):
    """-> (logits_r or None, logits_e, corrections or None)"""
    lr, le = [], []
    out_corr = np.zeros(self.n_corr, np.float32) if measure else None
    if measure:
      chunk = images.shape[0]
    for i in range(0, images.shape[0], chunk):
      x = images[i:i + chunk].contiguous()
      n = x.shape[0]
      r = np.zeros((n, 3), np.float32) if (want_r or measure) else None
      e = np.zeros((n, 3), np.float32)
      k = None if keep is None else np.ascontiguousarray(keep, np.uint8)
      c = None if corr is None else np.ascontiguousarray(corr, np.float32)
      torch.cuda.synchronize()
      _lib.check(self.lib.dv_model_probe_rounding(
          self.m._handle, self.flat.ctypes.data, self.flat.size, x.data_ptr(), n,
          None if k is None else k.ctypes.data, (1 if weights_f32 else 0) | (2 if measure else 0),
          None if c is None else c.ctypes.data, 0 if c is None else c.size,
          None if r is None else r.ctypes.data, e.ctypes.data, None if out_corr is None else out_corr.ctypes.data))
      if r is not None:
        lr.append(r)
      le.append(e)
    return (np.concatenate(lr) if lr else None), np.concatenate(le), out_corr


def dp_stats(le, lr):
  d = np.abs(softmax(le) - softmax(lr)).max(1)
  return d


def name_ops(labels):
  """Readable names: block / branch position from the construction order of layers."""
  names = []
  for i, l in enumerate(labels):
    if l['kind'] == 'conv':
      names.append('op%-3d L%-2s conv %sx%s s%s %4s->%-4s @%-6s%s' % (i, l['layer'], *l['k'].split('x'), l['s'], l['cin'], l['cout'],
                                                                    l['out'], ' raw' if l['raw'] == '1' else ''))
    else:
      names.append('op%-3d     %-7s %4s ch @%-6s' % (i, l['kind'], l['cout'], l['out']))
  return names


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='ont', choices=sorted(SHAPES))
  ap.add_argument('--seed', type=int, default=202)
  ap.add_argument('--n', type=int, default=1024)
  ap.add_argument('--ncal', type=int, default=256)
  ap.add_argument('--skip_attribution', action='store_true')
  args = ap.parse_args()
  shape = SHAPES[args.workload]
  t0 = time.time()
  if args.workload == 'illumina':
    x = T.illumina_pileups_gpu(args.n, seed=424242)
    xc = T.illumina_pileups_gpu(args.ncal, seed=990000 + args.seed)
  else:
    x = T.longread_images_gpu(args.workload, args.n)
    xc = T.longread_images_gpu(args.workload, args.ncal, seed=4711 + args.seed)
  ref = R.make_random_model(shape[2], seed=args.seed)
  flat = ref.export_flat()
  m = InceptionV3(shape, max_batch=min(args.n, 2048))
  m.load_flat_weights(flat)
  p = Probe(m, flat)
  names = name_ops(p.labels)
  print('# %s seed %d: %d evaluation images, %d calibration images (other seed), %d graph ops (%.0f s)' % (
      args.workload, args.seed, args.n, args.ncal, p.n_ops, time.time() - t0), flush=True)

  # ---- 1. E against the HIP kernels
  corr = m.calibrate(xc)
  hip = T.hip_probs(m, x, min(args.n, 2048))
  lr, le, _ = p.run(x, corr=corr, want_r=True)
  pe, pr = softmax(le), softmax(lr)
  print('# product (HIP kernels, calibrated) vs R: max |dp| %.3e mean %.3e;  E vs R: max %.3e mean %.3e;  HIP vs E: max %.3e' % (
      np.abs(hip - pr).max(), np.abs(hip - pr).max(1).mean(), np.abs(pe - pr).max(), np.abs(pe - pr).max(1).mean(),
      np.abs(hip - pe).max()), flush=True)
  want = T.oracle_probs_gpu(R.make_random_model(shape[2], seed=args.seed).cuda(), x)
  print('# R vs the torch fp32 oracle: max |dp| %.2e' % np.abs(pr - want).max(), flush=True)

  conv_like = [i for i, l in enumerate(p.labels) if l['kind'] != 'maxpool']
  # ---- 2. attribution (fp32 weights, no corrections: each rounding alone)
  if not args.skip_attribution:
    def zero_mean_var(le_):
      e = (le_ - lr).astype(np.float64)
      e = e - e.mean(0, keepdims=True)
      return float((e ** 2).mean())
    keep_none = np.zeros(p.n_ops, np.uint8)
    _, le_all, _ = p.run(x, keep=keep_none, weights_f32=True)
    v_all = zero_mean_var(le_all)
    _, le_w, _ = p.run(x, keep=np.ones(p.n_ops, np.uint8), weights_f32=False)
    print('# zero-mean logit error variance: all activation roundings %.3e; weights only %.3e' % (v_all, zero_mean_var(le_w)))
    rows = []
    for i in conv_like:
      keep = np.ones(p.n_ops, np.uint8)
      keep[i] = 0
      _, le_i, _ = p.run(x, keep=keep, weights_f32=True)
      rows.append((i, zero_mean_var(le_i)))
      print('  %-58s share of activation variance %6.2f %%' % (names[i], 100.0 * rows[-1][1] / v_all), flush=True)
    print('# sum of single-tensor variances / all-rounded variance = %.3f' % (sum(v for _, v in rows) / v_all))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", 'r6_budget_%s_%d.npy' % (args.workload, args.seed)), np.array(rows))

  # ---- 3. scenarios
  L = p.labels
  def layer(i):
    return int(L[i]['layer']) if L[i]['kind'] == 'conv' else -1
  raws = [i for i in conv_like if L[i]['kind'] == 'conv' and L[i]['raw'] == '1']
  feat_buf = L[-1]['out_buf']
  gap = [i for i in conv_like if L[i]['out_buf'] == feat_buf]
  def stage(i):   # by output map size
    return L[i]['out']
  sizes = []
  for i in conv_like:
    if stage(i) not in sizes:
      sizes.append(stage(i))
  s8 = sizes[-1]
  s17 = sizes[-2]
  st8 = [i for i in conv_like if stage(i) == s8]
  st17 = [i for i in conv_like if stage(i) == s17]
  chains17 = [i for i in st17 if L[i]['kind'] == 'conv' and max(int(v) for v in L[i]['k'].split('x')) == 7]
  heads17 = [i for i in st17 if L[i]['kind'] == 'conv' and L[i]['k'] == '1x1']
  concat17 = [i for i in st17 if int(L[i]['coff']) > 0 or (L[i]['kind'] == 'conv' and int(L[i]['cout']) == 192 and L[i]['k'] in ('1x1',) and L[i]['raw'] == '0')]
  scen = [('product (all fp16)', []),
          ('free: raw pooled fp32 + GAP inputs fp32', raws + gap),
          ('free + 8x8 stage (%s)' % s8, raws + gap + st8),
          ('free + 8x8 + 17x17 chain tensors (7-tap layers)', raws + gap + st8 + chains17),
          ('free + 8x8 + 17x17 1x1 outputs', raws + gap + st8 + heads17),
          ('free + 17x17 stage (%s)' % s17, raws + gap + st17),
          ('free + 8x8 + 17x17 stages', raws + gap + st8 + st17),
          ('everything fp32 (weights fp16)', conv_like)]
  print('# scenarios: tensors kept in fp32 in E, calibration re-measured on the %d other images under the same plan' % args.ncal)
  for label, ops in scen:
    keep = np.zeros(p.n_ops, np.uint8)
    keep[ops] = 1
    _, _, c = p.run(xc, keep=keep, measure=True)
    _, le_s, _ = p.run(x, keep=keep, corr=c)
    d = dp_stats(le_s, lr)
    print('  %-52s max %.3e  p99.9 %.3e  mean %.3e  over 1e-3: %d of %d' % (
        label, d.max(), np.quantile(d, 0.999), d.mean(), int((d > 1e-3).sum()), d.size), flush=True)


if __name__ == '__main__':
  main()
