"""End to end on the GPU, through the reference's own interfaces:
ExamplesGenerator.write_examples_in_region -> tf.Example TFRecord(GZIP)
-> call_variants -> CallVariantsOutput TFRecord(GZIP).

Input = the golden HG001 chr20:10,000,000-10,010,000 slice (BASELINE.json
configs[0]; tests/golden/illumina_wgs_chr20.npz).
"""
import os

import numpy as np
import pytest
import torch

from deepvariant_amd import dv_types as T
from tests import golden_io
from tests.golden.make_golden import wgs_options
from tests.test_oracle_golden import FIXTURE

pytestmark = pytest.mark.gpu


class _WindowRef:
  """Answers exactly the FASTA queries the generator makes for the fixture."""

  def __init__(self, examples, width):
    hw = (width - 1) // 2
    self._w = {ex['call'].variant.start - hw: ex['ref_window'] for ex in examples}

  def n_bases(self, contig):
    return 1 << 30

  def get_bases(self, contig, start, end):
    return self._w[start][:end - start]


def test_make_examples_then_call_variants(tmp_path):
  from deepvariant_amd import call_variants as cv
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from deepvariant_amd import tfrecord
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import inception_ref as R
  from oracle import oracle as O

  reads, examples, z = golden_io.load(FIXTURE)
  pic = wgs_options()
  hw = (pic.width - 1) // 2
  options = T.MakeExamplesOptions(
      pic_options=pic,
      sample_options=[T.SampleOptions(role='main', name='NA12878',
                                      pileup_height=100)])
  # candidates in example order (multi-allelic sites yield 3 examples each)
  cands, seen = [], set()
  for ex in examples:
    key = (ex['call'].variant.start, tuple(ex['call'].variant.alternate_bases))
    if key not in seen:
      seen.add(key)
      cands.append(ex['call'])
  ex_path = str(tmp_path / 'examples.tfrecord.gz')
  gen = men.ExamplesGenerator(options, {'main': ex_path},
                              ref_reader=_WindowRef(examples, pic.width))
  stats, shape = gen.write_examples_in_region(cands, [reads], [0], 'main', [0.0])
  gen.signal_shard_finished()
  men.write_example_info_json(ex_path, shape, pic.channels)
  assert shape == [100, 221, 7]
  assert stats['n_examples'] == 84
  assert stats['n_snps'] + stats['n_indels'] == 84
  assert cv.example_info_shape(ex_path) == [100, 221, 7]

  got = [pw.decode_example(r) for r in tfrecord.read_tfrecords(ex_path, verify_crc=True)]
  assert len(got) == 84
  n_full = 0
  for i, (ex, rec) in enumerate(zip(examples, got)):
    call = ex['call']
    v = call.variant
    assert rec['locus'][0].decode() == '%s:%d-%d' % (v.reference_name, v.start + 1, v.end)
    assert rec['image/shape'] == [100, 221, 7]
    idx = [call.variant.alternate_bases.index(a) for a in ex['alt_alleles']]
    assert rec['alt_allele_indices/encoded'][0] == pw.encode_alt_allele_indices(idx)
    img = np.frombuffer(rec['image/encoded'][0], np.uint8).reshape(100, 221, 7)
    want = O.build_pileup(pic, call, ex['ref_window'],
                          [reads[k] for k in ex['read_idx']], v.start - hw,
                          ex['alt_alleles'])
    np.testing.assert_array_equal(img, want)
    if z['e_full'][i]:
      np.testing.assert_array_equal(img, ex['image'])  # == the reference's bytes
      n_full += 1
  assert n_full == 7

  ref = R.make_random_model(7, seed=11)
  model = InceptionV3((100, 221, 7), max_batch=32)
  model.load_flat_weights(ref.export_flat())
  out = str(tmp_path / 'cvo.tfrecord.gz')
  n = cv.call_variants(ex_path, out, model, batch_size=32, writer_shards=2)
  assert n == 84
  shards = [str(tmp_path / ('cvo-%05d-of-00002.tfrecord.gz' % i)) for i in range(2)]
  cvos = [pw.decode_call_variants_output(r) for s in shards
          for r in tfrecord.read_tfrecords(s, verify_crc=True)]
  assert len(cvos) == 84
  imgs = np.stack([np.frombuffer(r['image/encoded'][0], np.uint8).reshape(100, 221, 7)
                   for r in got[:6]])
  with torch.no_grad():
    want_p = ref(torch.from_numpy(imgs)).numpy()
  # writer shards interleave records round-robin: record k -> shard k % 2
  by_shard = [[pw.decode_call_variants_output(r) for r in tfrecord.read_tfrecords(s)]
              for s in shards]
  for k in range(6):
    variant, alt, probs = by_shard[k % 2][k // 2]
    assert abs(sum(probs) - 1.0) < 1e-6 and len(probs) == 3
    assert max(abs(p - w) for p, w in zip(probs, want_p[k])) <= 1e-3 + 1e-9
    assert variant.start == examples[k]['call'].variant.start

  # fused region path (no tf.Example / GZIP hop): same CallVariantsOutput bytes, same order
  gen2 = men.ExamplesGenerator(options, {}, test_mode=True,
                               ref_reader=_WindowRef(examples, pic.width))
  fused = gen2.call_variants_in_region(cands, [reads], [0], [0.0], model)
  two_step = [list(tfrecord.read_tfrecords(s)) for s in shards]
  assert len(fused) == 84
  for k, rec in enumerate(fused):
    assert rec == two_step[k % 2][k // 2]

  # the command line (reference flag names) writes the same records
  flat = str(tmp_path / 'weights.npy')
  np.save(flat, ref.export_flat())
  out2 = str(tmp_path / 'cli.tfrecord.gz')
  assert cv.main(['--examples', ex_path, '--outfile', out2, '--checkpoint', flat,
                  '--batch_size', '32', '--writer_threads', '2', '--num_readers', '4']) == 0
  for i in range(2):
    a = list(tfrecord.read_tfrecords(shards[i]))
    b = list(tfrecord.read_tfrecords(str(tmp_path / ('cli-%05d-of-00002.tfrecord.gz' % i))))
    assert a == b


def test_region_from_native_bam_table_equals_region_from_read_protos(tmp_path):
  """f1 plumbing: reads packed natively from a BAM (packing.ReadTable.from_bam) give
  byte-identical examples to the same reads handed over as Read protos."""
  from deepvariant_amd import genomics_io, packing
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import tfrecord
  from tests.test_bam_native_cpu import _write_bam
  bam = str(tmp_path / 's.bam')
  _write_bam(bam, np.random.default_rng(3), n=600)
  _, reads = genomics_io.read_bam(bam, 'chrA', 0, 1 << 40)
  reads = [r for r in reads if genomics_io.read_satisfies_requirements(r, min_mapping_quality=5)]
  table = packing.ReadTable.from_bam(bam, 'chrA', 0, 1 << 40, min_mapping_quality=5)
  assert table.n_reads == len(reads) > 50
  rng = np.random.default_rng(8)
  cands = []
  for k in range(12):
    r = reads[int(rng.integers(0, len(reads)))]
    pos = r.alignment.position.position + 3
    names = ['%s/%d' % (q.fragment_name, q.read_number) for q in reads
             if q.alignment.position.position <= pos < q.alignment.position.position + 30][:5]
    cands.append(T.DeepVariantCall(
        variant=T.Variant('chrA', pos, pos + 1, 'A', ['C', 'G'][:1 + k % 2]),
        allele_support={'C': T.SupportingReads(names)}))
  cands.sort(key=lambda c: c.variant.start)
  pic = wgs_options()
  options = T.MakeExamplesOptions(
      pic_options=pic, sample_options=[T.SampleOptions(role='main', name='s', pileup_height=100)])

  class _Ref:
    def n_bases(self, contig):
      return 100000

    def get_bases(self, contig, start, end):
      return ''.join('ACGT'[(p * 7 + p // 3) % 4] for p in range(start, end))

  outs = []
  for tag, sample in (('protos', reads), ('table', table)):
    path = str(tmp_path / ('%s.tfrecord.gz' % tag))
    gen = men.ExamplesGenerator(options, {'main': path}, ref_reader=_Ref())
    stats, shape = gen.write_examples_in_region(cands, [sample], [0], 'main', [0.0])
    gen.signal_shard_finished()
    outs.append(list(tfrecord.read_tfrecords(path)))
    assert stats['n_examples'] == len(outs[-1]) >= 12
  assert outs[0] == outs[1]
