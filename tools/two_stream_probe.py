"""Does running two half-batches on two HIP streams (each its own model instance, hence its own
activation buffers and graph) beat one full batch on one stream?  Measures whether kernel tails
and per-kernel drains leave anything for a concurrent stream to pick up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepvariant_amd.inception_v3 import InceptionV3

N, H, W, C = 8104, 100, 221, 7
dev = torch.device('cuda:0')
images = torch.randint(0, 255, (N, H, W, C), dtype=torch.uint8, device=dev)

def timed(fn, steps=12, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e3

one = InceptionV3((H, W, C), max_batch=8192)
one.init_random(1)
s0 = torch.cuda.Stream(device=dev)
def single():
  with torch.cuda.stream(s0):
    one(images)
t1 = timed(single)
print('one stream, %d images: %.2f ms' % (N, t1))
for parts in (2, 3):
  models, streams, chunks = [], [], []
  per = (N + parts - 1) // parts
  for k in range(parts):
    m = InceptionV3((H, W, C), max_batch=per)
    m.init_random(1)
    models.append(m)
    streams.append(torch.cuda.Stream(device=dev))
    chunks.append(images[k * per:(k + 1) * per])
  def multi():
    for m, s, c in zip(models, streams, chunks):
      with torch.cuda.stream(s):
        m(c)
  t = timed(multi)
  print('%d streams x %d images: %.2f ms  (%+.1f %%)' % (parts, per, t, 100 * (t1 - t) / t1))
  del models
