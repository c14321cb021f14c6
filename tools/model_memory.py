"""Device memory one model instance takes (activation buffers + packed weights), by hipMemGetInfo
before / after dv_model_create:  python tools/model_memory.py [max_batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvariant_amd.inception_v3 import InceptionV3   # noqa: E402

for mb in [int(a) for a in sys.argv[1:]] or [256, 1024, 8192]:
  torch.cuda.synchronize()
  free0, _ = torch.cuda.mem_get_info()
  m = InceptionV3((100, 221, 7), max_batch=mb)
  torch.cuda.synchronize()
  free1, _ = torch.cuda.mem_get_info()
  print('max_batch %5d: %8.1f MB  = %.2f MB per example' % (mb, (free0 - free1) / 1e6, (free0 - free1) / 1e6 / mb))
  del m
