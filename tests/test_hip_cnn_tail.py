"""The CNN's 1e-3 bar at genome-like sample sizes, off the tuning seeds (GPU).

BASELINE.json `north_star`: "softmax genotype probabilities match within 1e-3" of the reference's fp32
arithmetic (deepvariant/call_variants.py:913-918, deepvariant/dv_utils.py:343-366).  tests/test_hip_precision.py
holds that bar on 2048 pileups for the weight seeds the precision work was tuned on; this file measures the
TAIL: 65,536 encoder-drawn ILLUMINA30 pileups on weight seeds {101, 202, 303} that no sweep ever used, and 2048
examples of each long-read shape (100x147x10, 100x199x9) on the same held-out seeds.  The fp32 oracle runs on the GPU through torch-ROCm
(tests/cnn_tail.py) after being checked against its CPU form on 256 of the same images.

What is asserted is the bar itself, with the product's default model preparation -- plain fp16 weights, shifts
calibrated (dv_model_calibrate) on the checkpoint's fixed synthetic set of 256 OTHER pileups
(InceptionV3.calibrate_for_checkpoint): every one of the 65,536 candidates within 1e-3 on every
held-out seed (profiles/r05_cnn_tail.txt: max |dp| 8.6e-4 / 2.9e-4 / 7.2e-4, none over; without calibration seed 101
has 11 candidates over, with the round-4 split weights 1).  Why the calibration works and what is left of the
fp16 error: HISTORY.md 15 and DESIGN.md 6.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_TAIL = 65536
HELD_OUT_SEEDS = (101, 202, 303)
_cache = {}


def _tail_images():
  from tests import cnn_tail as T
  if 'x' not in _cache:
    _cache['x'] = T.illumina_pileups_gpu(N_TAIL, seed=424242)
  return _cache['x']


def _product_model(shape, weights, max_batch, precise=None):
  """The product's model preparation: weights, then the shift calibration on the checkpoint's fixed synthetic set
  (InceptionV3.calibrate_for_checkpoint -- what call_variants / make_examples do; other pileups than any sample here).
  `precise` None = the product's default for the shape (precise mode for more than 8 input channels), else DV_PRECISE."""
  import os
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.pop('DV_PRECISE', None)
  if precise is not None:
    os.environ['DV_PRECISE'] = '1' if precise else '0'
  try:
    m = InceptionV3(shape, max_batch=max_batch)
  finally:
    os.environ.pop('DV_PRECISE', None)
    if old is not None:
      os.environ['DV_PRECISE'] = old
  m.load_flat_weights(weights)
  m.calibrate_for_checkpoint(256)
  return m


def test_gpu_oracle_equals_cpu_oracle_on_256_pileups():
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  x = _tail_images()
  ref = R.make_random_model(7, seed=HELD_OUT_SEEDS[0])
  ref_gpu = R.make_random_model(7, seed=HELD_OUT_SEEDS[0]).cuda()
  d = T.check_gpu_oracle(ref, ref_gpu, x, n=256, tol=5e-6)
  print('GPU fp32 oracle vs CPU fp32 oracle on 256 pileups: max |dp| %.3g' % d)


@pytest.mark.parametrize('seed', HELD_OUT_SEEDS)
def test_illumina30_tail_on_held_out_weight_seeds(seed):
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  x = _tail_images()
  ref = R.make_random_model(7, seed=seed)
  want = T.oracle_probs_gpu(R.make_random_model(7, seed=seed).cuda(), x)
  model = _product_model((100, 221, 7), ref.export_flat(), 8192)
  got = T.hip_probs(model, x, 8192)
  s = T.tail_stats(got, want)
  print('seed %d: %s' % (seed, T.fmt(s)))
  assert s['prob_spread'] > 5e-2, s                # the random network is not a constant
  assert s['n_over_tol'] == 0, s                   # no candidate of the 65,536 beyond 1e-3
  assert s['max_abs_dp'] <= 1e-3, s


@pytest.mark.parametrize('seed', HELD_OUT_SEEDS)
@pytest.mark.parametrize('kind,shape', [('hifi', (100, 147, 10)), ('ont', (100, 199, 9))])
def test_long_read_shapes_on_2048_examples(kind, shape, seed):
  """The PACBIO / ONT_R104 input shapes (BASELINE configs[3], configs[4]) on bench.py's hifi35 / ont50 images,
  held-out weight seeds, THE PRODUCT'S DEFAULT for these shapes: precise mode (hi + lo activations through the 17x17
  and 8x8 stages, include/dvhip.h dv_model_is_precise) + the checkpoint calibration.  ASSERTED: north_star's bar
  itself on all six (shape, seed) pairs -- no candidate of the 2,048 beyond 1e-3 (profiles/r06_cnn_tail_longread.txt:
  max 2e-5 .. 5.2e-4; ONT / 202 at N = 65,536 in the same file)."""
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  n = 2048
  x = _longread_images(kind, n)
  assert tuple(x.shape[1:]) == shape
  ref = R.make_random_model(shape[2], seed=seed)
  ref_gpu = R.make_random_model(shape[2], seed=seed).cuda()
  T.check_gpu_oracle(ref, ref_gpu, x, n=64, tol=5e-6)
  want = T.oracle_probs_gpu(ref_gpu, x)
  model = _product_model(shape, ref.export_flat(), n)
  assert model.precise
  got = T.hip_probs(model, x, n)
  s = T.tail_stats(got, want)
  print('%s %s seed %d (precise, the default): %s' % (kind, shape, seed, T.fmt(s)))
  assert s['n_over_tol'] == 0, s
  assert s['max_abs_dp'] <= 1e-3, s


@pytest.mark.parametrize('seed', HELD_OUT_SEEDS)
@pytest.mark.parametrize('kind,shape', [('hifi', (100, 147, 10)), ('ont', (100, 199, 9))])
def test_long_read_shapes_in_fast_mode(kind, shape, seed):
  """The same with DV_PRECISE=0 (fp16 activations everywhere: 45 % faster on these shapes).  The bar is asserted;
  a (shape, seed) pair that misses it is reported as XFAIL with its numbers, never as a pass -- ONT_R104 / seed 202
  does (profiles/r06_cnn_tail_longread.txt: 1.15e-3, 5 of 2,048 over; DESIGN.md 6 has the per-tensor budget that says
  why no cheaper set of wide tensors would do).  Hard regression guard underneath: p99.9 <= 1.25e-3, <= 1 % over,
  max <= 1.6e-3."""
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  n = 2048
  x = _longread_images(kind, n)
  ref = R.make_random_model(shape[2], seed=seed)
  want = T.oracle_probs_gpu(R.make_random_model(shape[2], seed=seed).cuda(), x)
  model = _product_model(shape, ref.export_flat(), n, precise=False)
  assert not model.precise
  got = T.hip_probs(model, x, n)
  s = T.tail_stats(got, want)
  print('%s %s seed %d (fast mode): %s' % (kind, shape, seed, T.fmt(s)))
  assert s['p999_abs_dp'] <= 1.25e-3, s
  assert s['n_over_tol'] <= n // 100, s
  assert s['max_abs_dp'] <= 1.6e-3, s
  if s['n_over_tol'] != 0 or s['max_abs_dp'] > 1e-3:
    pytest.xfail('%s seed %d misses the 1e-3 bar in fast mode: %s' % (kind, seed, T.fmt(s)))


def test_illumina30_in_precise_mode_is_opt_in_and_tighter():
  """DV_PRECISE=1 on the short-read shape (off by default there: the bar holds without it): the tail shrinks."""
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  x = _tail_images()[:4096]
  seed = HELD_OUT_SEEDS[0]
  ref = R.make_random_model(7, seed=seed)
  want = T.oracle_probs_gpu(R.make_random_model(7, seed=seed).cuda(), x)
  fast = _product_model((100, 221, 7), ref.export_flat(), 4096)
  prec = _product_model((100, 221, 7), ref.export_flat(), 4096, precise=True)
  assert not fast.precise and prec.precise
  sf = T.tail_stats(T.hip_probs(fast, x, 4096), want)
  sp = T.tail_stats(T.hip_probs(prec, x, 4096), want)
  print('fast: %s\nprecise: %s' % (T.fmt(sf), T.fmt(sp)))
  assert sp['max_abs_dp'] <= 1e-3 and sf['max_abs_dp'] <= 1e-3
  assert sp['mean_abs_dp'] <= 0.8 * sf['mean_abs_dp'], (sf, sp)


def _longread_images(kind, n):
  from tests import cnn_tail as T
  key = ('longread', kind, n)
  if key not in _cache:
    _cache[key] = T.longread_images_gpu(kind, n)
  return _cache[key]
