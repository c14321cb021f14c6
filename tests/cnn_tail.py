"""Large-N CNN parity helpers (GPU box; test infrastructure, never imported by the product).

The fp32 oracle (oracle/inception_ref.py) takes minutes per 2048 images on host cores; a genome is
10^6-10^7 candidates and the largest |dp| grows with the sample.  Here the SAME oracle module runs on
the GPU through torch-ROCm in fp32 (its im2col + matmul formulation, `ConvBN.as_gemm`: rocBLAS and
ATen kernels only -- MIOpen would compile its solvers for minutes on a fresh box), after it has been
checked against its own CPU conv2d form on the same images (`check_gpu_oracle`).  Used by
tests/test_hip_cnn_tail.py, tools/r5_cnn_tail.py and bench.py's parity leg.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import inception_ref as R   # noqa: E402  (checker)


def illumina_pileups_gpu(n, seed, chunk=8192, device='cuda'):
  """n encoder-drawn ILLUMINA30 pileups [n,100,221,7] uint8, resident on the GPU (HIP encoder)."""
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  enc = _Encoder(opts, opts.width)
  out = torch.empty((n, 100, 221, 7), dtype=torch.uint8, device=device)
  done = 0
  k = 0
  while done < n:
    m = min(chunk, n - done)
    batch = synth.make_illumina_batch(m, seed=seed + 7919 * k, options=opts, multi_allelic=False)
    img, _ = enc.encode(batch, 7)
    out[done:done + m] = torch.from_numpy(np.ascontiguousarray(img.reshape(-1, 100, 221, 7)[:m])).to(device)
    done += m
    k += 1
  return out


def longread_images_gpu(kind, n, seed=None, device='cuda'):
  """n examples of bench.py's hifi35 / ont50 workload ('hifi' -> [n,100,147,10], 'ont' -> [n,100,199,9]):
  HIP encoder + dv_merge_alt_channels, exactly the bench step's tensor."""
  import ctypes as CT
  import bench
  from deepvariant_amd import _lib
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.pileup_image_native import _Encoder
  opts, batch, with_alt, c_enc, ct = bench.make_longread_workload(kind, n, seed=seed)
  h, w = opts.height, opts.width
  img_bytes = h * w * ct
  entries = (_lib.DvAltMergeEntry * max(len(with_alt), 1))()
  for k, i in enumerate(with_alt):
    entries[k].example, entries[k].first_row, entries[k].rows = i, 0, h
    entries[k].scratch_alt1, entries[k].scratch_alt2 = 2 * k, 2 * k + 1
  dev = torch.device(device)
  dbatch = DeviceBatch(batch, dev)
  enc = _Encoder(opts, w)
  flat = torch.zeros(batch.n_items * img_bytes, dtype=torch.uint8, device=dev)
  rows = torch.empty(batch.n_items, dtype=torch.int32, device=dev)
  stream = torch.cuda.current_stream(dev)
  dbatch.encode(enc, ct, flat, rows)
  _lib.check(_lib.lib().dv_merge_alt_channels(flat.data_ptr(), n * img_bytes, img_bytes, img_bytes, w, ct, c_enc,
                                              5, entries, len(with_alt), CT.c_void_p(stream.cuda_stream)))
  torch.cuda.synchronize(dev)
  return flat[:n * img_bytes].view(n, h, w, ct).clone()


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


def hip_probs(model, images, chunk):
  outs = []
  for i in range(0, images.shape[0], chunk):
    outs.append(model(images[i:i + chunk]).cpu())
  return torch.cat(outs).numpy()


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
