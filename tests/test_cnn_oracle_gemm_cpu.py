"""The oracle's im2col + matmul formulation (oracle/inception_ref.py `ConvBN.as_gemm`, what the
large-N GPU tests run through torch-ROCm without MIOpen) equals its conv2d formulation (CPU)."""
import numpy as np
import pytest
import torch

from oracle import inception_ref as R


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 147, 10), (100, 199, 9)])
def test_gemm_formulation_equals_conv2d(shape):
  h, w, c = shape
  ref = R.make_random_model(c, seed=11)
  rng = np.random.default_rng(3)
  x = torch.from_numpy(rng.integers(0, 256, (3, h, w, c), dtype=np.uint8))
  with torch.no_grad():
    want = ref(x)
    R.ConvBN.as_gemm = True
    try:
      got = ref(x)
    finally:
      R.ConvBN.as_gemm = False
  assert float((got - want).abs().max()) <= 1e-6
  assert got.shape == (3, 3)
