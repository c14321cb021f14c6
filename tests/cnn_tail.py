"""Large-N CNN parity helpers (GPU box; test infrastructure, never imported by the product).

Images for the large-N tests drawn by the HIP encoder and kept on the GPU; the fp32 oracle on the GPU and the
tail statistics live in oracle/inception_gpu.py (re-exported here).  Used by tests/test_hip_cnn_tail.py,
tests/test_hip_calibration.py and tools/r5_cnn_tail.py.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import inception_ref as R   # noqa: E402,F401  (checker)
from oracle.inception_gpu import (check_gpu_oracle, fmt, oracle_probs_cpu, oracle_probs_gpu,   # noqa: E402,F401
                                  tail_stats)


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
  from deepvariant_amd import calibration_set
  dev = torch.device(device)
  return calibration_set.longread_examples(kind, n, seed=seed, device=dev.index or 0)


def hip_probs(model, images, chunk):
  outs = []
  for i in range(0, images.shape[0], chunk):
    outs.append(model(images[i:i + chunk]).cpu())
  return torch.cat(outs).numpy()


