"""The product multi-GPU driver (`make_examples --gpus N`, deepvariant_amd/make_examples.py
distributed_runner) on CPU: world_size 2 over gloo.  Rank r must be task r of N (the reference's
round robin, make_examples_core.py:879-888), the CallVariantsOutput records must cross ranks in one
gather (deepvariant_amd/dist.py gather_records), and rank 0 must write N shard files that are
byte-identical to what two independent `--task r` runs write (scripts/run_deepvariant.py:457-462 +
call_variants.py:934-951 is how the reference gets the same files).  The GPU-only parts (region
processor, classifier) are replaced by a stub with the same interface, so that what is tested is
the driver: sharding, the in-memory sink, the gather and the writers."""
import hashlib
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from deepvariant_amd import genomics_io
from deepvariant_amd import make_examples as me
from deepvariant_amd import tfrecord
from tests import realigner_fixture as RF


class _StubProcessor:
  """call_variants_in_region's interface; one deterministic record per read that STARTS in the
  region (so every record belongs to exactly one region), variable lengths."""

  def call_variants_in_region(self, region, reads, model):
    assert model == 'stub-model'
    records = []
    for r in reads:
      p = r.alignment.position.position
      if region.start <= p < region.end:
        seed = ('%s/%d@%d' % (r.fragment_name, r.read_number, p)).encode()
        records.append(seed + hashlib.sha256(seed).digest() * (1 + p % 5))
    return list(records), records


class _StubHooks(me.RunnerHooks):
  def make_processor(self, options, ref_reader, po, device):
    return _StubProcessor()

  def make_model(self, args, options):
    return 'stub-model'


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _args(tmp, out, extra=()):
  return me.build_arg_parser().parse_args(
      ['--ref', os.path.join(tmp, 'ref.fa'), '--reads', os.path.join(tmp, 'reads.bam'), '--checkpoint', 'random:1',
       '--call_variants_outfile', out, '--regions', 'chr20:10,000,000-10,010,000', '--partition_size', '700'] +
      list(extra))


def _rank(rank, world, port, tmp, flags):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    args = _args(tmp, os.path.join(tmp, 'gathered.cvo.tfrecord@%d.gz' % world), flags)
    me.distributed_runner(args, rank, world, log=open(os.devnull, 'w'), hooks=_StubHooks())
  finally:
    dist.destroy_process_group()


def _fixture(tmp):
  ref, sets = RF.load()
  reads = sorted(sets['wgs'], key=lambda r: r.alignment.position.position)
  genomics_io.write_bam(os.path.join(tmp, 'reads.bam'), [('chr20', 10_020_000)], reads)
  lo = 9_990_000     # the stretch the fixture covers, N before it
  genomics_io.write_fasta(os.path.join(tmp, 'ref.fa'), [('chr20', 'N' * lo + ref.get_bases('chr20', lo, 10_020_000))])
  return len(reads)


# two GPUs with one rank each, or one GPU shared by two host processes: the same two tasks
@pytest.mark.parametrize('flags', [['--gpus', '2'], ['--gpus', '1', '--ranks_per_gpu', '2']], ids=['gpus2', 'ranks_per_gpu2'])
@pytest.mark.timeout(600)
def test_two_ranks_write_what_two_tasks_write(tmp_path, flags):
  tmp = str(tmp_path)
  n_reads = _fixture(tmp)
  world = 2
  # the reference's way: two independent tasks, each writing its own shard
  independent = []
  for task in range(world):
    args = _args(tmp, os.path.join(tmp, 'tasks.cvo.tfrecord@%d.gz' % world), ['--task', str(task)])
    me.make_examples_runner(args, log=open(os.devnull, 'w'), hooks=_StubHooks())
    independent.append(list(tfrecord.read_tfrecords(os.path.join(tmp, 'tasks.cvo.tfrecord-%05d-of-%05d.gz' % (task, world)))))
  assert all(len(x) > 100 for x in independent)
  assert sum(len(x) for x in independent) <= n_reads
  # the node-level driver: two ranks over gloo, one gather, rank 0 writes both shards
  port = _free_port()
  mp.spawn(_rank, args=(world, port, tmp, flags), nprocs=world, join=True)
  for task in range(world):
    got = list(tfrecord.read_tfrecords(os.path.join(tmp, 'gathered.cvo.tfrecord-%05d-of-%05d.gz' % (task, world))))
    assert got == independent[task]
  assert independent[0] != independent[1]


def test_flag_checks():
  ap = me.build_arg_parser()
  with pytest.raises(ValueError, match='fused route'):
    me.check_flags(ap.parse_args(['--ref', 'r', '--reads', 'b', '--examples', 'e@2.gz', '--gpus', '2']))
  args = ap.parse_args(['--ref', 'r', '--reads', 'b', '--call_variants_outfile', 'c.gz', '--checkpoint', 'random:1',
                        '--gpus', '2'])
  me.check_flags(args)
  with pytest.raises(ValueError, match='one shard per rank'):
    me.distributed_runner(args, 0, 2)
  with pytest.raises(ValueError, match='fused route'):
    me.check_flags(ap.parse_args(['--ref', 'r', '--reads', 'b', '--examples', 'e@2.gz', '--ranks_per_gpu', '2']))
  with pytest.raises(ValueError, match='ranks_per_gpu'):
    me.check_flags(ap.parse_args(['--ref', 'r', '--reads', 'b', '--examples', 'e.gz', '--ranks_per_gpu', '0']))


def test_background_model_hands_over_the_model_or_its_error():
  """make_examples._BackgroundModel: the classifier set up on a worker thread -- the input shape is
  known at once, the first use waits for the thread, a failed set-up is raised on the calling
  thread (every time it is asked for), a model of another shape is refused."""
  import threading

  class _Model:
    input_shape = (100, 221, 7)
    max_batch = 64

    def __call__(self, images):
      return ('classified', images)

  started = threading.Event()

  def build():
    started.wait(5)
    return _Model()

  bg = me._BackgroundModel(build, (100, 221, 7))              # pylint: disable=protected-access
  assert bg.input_shape == (100, 221, 7)
  started.set()
  assert bg.max_batch == 64 and bg('x') == ('classified', 'x') and isinstance(bg.get(), _Model)

  def broken():
    raise ValueError('no such checkpoint')
  bad = me._BackgroundModel(broken, (100, 221, 7))            # pylint: disable=protected-access
  for _ in range(2):
    with pytest.raises(ValueError, match='no such checkpoint'):
      bad.get()
  with pytest.raises(ValueError, match='no such checkpoint'):
    bad('x')
  other = me._BackgroundModel(_Model, (100, 147, 10))         # pylint: disable=protected-access
  with pytest.raises(ValueError, match='model shape'):
    other.get()
