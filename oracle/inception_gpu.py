"""The CNN oracle (oracle/inception_ref.py) run on the GPU, and the tail statistics of |dp|.

TEST INFRASTRUCTURE ONLY (tests/, tools/, smoke(), bench.py's parity leg) -- never the product.
The fp32 oracle takes minutes per 2048 images on host cores; a genome is 10^6-10^7 candidates and the
largest |dp| grows with the sample.  Here the SAME module runs through torch-ROCm in fp32 in its
im2col + matmul formulation (`ConvBN.as_gemm`: rocBLAS and ATen kernels only -- MIOpen would compile its
solvers for minutes on a fresh box), after `check_gpu_oracle` has compared it with the CPU conv2d form on
some of the same images.
"""
import os

import numpy as np
import torch

from oracle import inception_ref as R


def oracle_probs_gpu(ref_gpu, images, batch=256):
  """fp32 oracle on the GPU: images uint8 [N,H,W,C] (cuda) -> float32 probabilities [N,3] (numpy)."""
  outs = []
  R.ConvBN.as_gemm = True
  try:
    with torch.no_grad():
      for i in range(0, images.shape[0], batch):
        outs.append(ref_gpu(images[i:i + batch]).cpu())
  finally:
    R.ConvBN.as_gemm = False
  return torch.cat(outs).numpy()


def oracle_probs_cpu(ref, images_np, batch=64):
  torch.set_num_threads(min(128, os.cpu_count() or 1))
  with torch.no_grad():
    return torch.cat([ref(torch.from_numpy(images_np[i:i + batch]), channels_last=True)
                      for i in range(0, len(images_np), batch)]).numpy()


def check_gpu_oracle(ref, ref_gpu, images, n=256, tol=5e-6):
  """The GPU run of the oracle against its CPU conv2d form on the first n images; returns max |dp|.
  Both are float32 with different summation orders: measured 0.4e-6 .. 1.6e-6 on 256 pileups
  (profiles/r05_cnn_tail.txt), three orders of magnitude under the bar being checked."""
  x = images[:n]
  got = oracle_probs_gpu(ref_gpu, x)
  want = oracle_probs_cpu(ref, x.cpu().numpy())
  d = float(np.abs(got - want).max())
  assert d <= tol, 'GPU fp32 oracle differs from the CPU oracle by %.3g on %d images' % (d, n)
  return d


def tail_stats(got, want, tol=1e-3):
  e = np.abs(got.astype(np.float64) - want.astype(np.float64)).max(axis=1)
  return {
      'n': int(e.size), 'max_abs_dp': float(e.max()), 'mean_abs_dp': float(e.mean()),
      'p999_abs_dp': float(np.quantile(e, 0.999)), 'p9999_abs_dp': float(np.quantile(e, 0.9999)),
      'n_over_tol': int((e > tol).sum()), 'tol': tol,
      'prob_spread': float((want.max(0) - want.min(0)).max()),
  }


def fmt(s):
  return ('n %d  max %.3e  p99.99 %.3e  p99.9 %.3e  mean %.3e  over %.0e: %d (%.2e of the sample)  spread %.2f' % (
      s['n'], s['max_abs_dp'], s['p9999_abs_dp'], s['p999_abs_dp'], s['mean_abs_dp'], s['tol'], s['n_over_tol'],
      s['n_over_tol'] / s['n'], s['prob_spread']))
