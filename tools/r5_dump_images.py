"""GPU box: a few of the bench's long-read images as a compressed file (for CPU-side emulation of the fp16 roundings)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import cnn_tail as T
out = {}
for kind, n in (('ont', 128), ('hifi', 128)):
  out[kind] = T.longread_images_gpu(kind, n).cpu().numpy()
np.savez_compressed('gpurun_out/r5/images_longread.npz', **out)
