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
  model.calibrate_for_checkpoint(256)     # the drivers' model preparation (the command line below does the same)
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


def test_reads_to_probabilities_chain():
  """The whole product chain on one synthetic region, no files and no golden inputs in
  between: reads -> device allele counts (dv_count_alleles) -> candidate caller ->
  ExamplesGenerator.call_variants_in_region (dv_encode_batch + dv_model_infer) ->
  CallVariantsOutput.  The oracle chain runs the same steps with the Python allele counter,
  the C++ encoder restatement and the fp32 CNN; candidates and tensors must be identical,
  probabilities within 1e-3."""
  from deepvariant_amd import allelecounter as ac
  from deepvariant_amd import make_examples_native as men
  from deepvariant_amd import protowire as pw
  from deepvariant_amd import variant_calling as vc
  from deepvariant_amd.inception_v3 import InceptionV3
  from oracle import allelecounter_ref as AR
  from oracle import inception_ref as R
  from oracle import oracle as O
  from tests import fuzz_inputs as F
  from tests import test_hip_region_multisample as M

  rng = np.random.default_rng(2024)
  width, height = 221, 100
  pic = F.options(T.PILEUP_DEFAULT_CHANNELS + ['insert_size'], width, height)
  options = T.MakeExamplesOptions(pic_options=pic,
                                  sample_options=[T.SampleOptions(role='main', name='s', pileup_height=height)])
  ref = M._Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000)))
  variants = []
  for pos in range(1200, 4600, 170):
    kind = int(rng.integers(0, 3))
    refb = ref.seq[pos] if kind < 2 else ref.seq[pos:pos + 1 + int(rng.integers(1, 5))]
    altb = ([b for b in 'ACGT' if b != refb][0] if kind == 0 else
            refb + 'GATT'[:int(rng.integers(1, 5))] if kind == 1 else refb[0])
    variants.append((pos, refb, altb))
  reads = M._haplotype_reads(rng, ref.seq, variants, 900, 700, 4700)
  for r in reads:
    r.number_reads, r.fragment_length = 2, int(rng.integers(-600, 600))
  start, end = 1000, 4800
  kw = dict(min_mapping_quality=5, min_base_quality=10)
  counter = ac.AlleleCounter(ref, 'chr1', start, end, **kw)
  oracle_counter = AR.AlleleCounter(ref, 'chr1', start, end, **kw)
  for r in reads:
    counter.add(r)
    oracle_counter.add(r)
  caller = vc.VariantCaller(vc.VariantCallerOptions(2, 2, 0.12, 0.06, sample_name='s'))
  cands = caller.calls_from_allele_counter(counter)
  # the oracle chain's candidates (Python counter -> the same caller) must be the same calls
  want_counts = []
  for c in oracle_counter.counts:
    a = ac.AlleleCount('chr1', c.position, c.ref_base)
    a.ref_supporting_read_count = c.ref_supporting_read_count
    a.read_alleles = {k: ac.Allele(v.bases, v.type, 1, v.is_low_quality) for k, v in c.read_alleles.items()}
    want_counts.append(a)
  want_cands = caller.calls_from_allele_counts(want_counts)
  key = lambda c: (c.variant.start, c.variant.reference_bases, tuple(c.variant.alternate_bases),
                   tuple(sorted((k, tuple(sorted(s.read_names))) for k, s in c.allele_support.items())))
  assert [key(c) for c in cands] == [key(c) for c in want_cands]
  found = {c.variant.start for c in cands}
  assert len(cands) >= 15 and sum(v[0] in found for v in variants) >= 15    # the planted variants are found

  weights = R.make_random_model(7, seed=8)
  model = InceptionV3((height, width, 7), max_batch=64)
  model.load_flat_weights(weights.export_flat())
  gen = men.ExamplesGenerator(options, {}, test_mode=True, ref_reader=ref)
  cvos = gen.call_variants_in_region(cands, [reads], [0], [0.0], model)
  hw = (width - 1) // 2
  k = 0
  worst = 0.0
  for cand in cands:
    v = cand.variant
    window = men.get_reference_bases_for_pileup(ref, v, width)
    overlapping = M._query(reads, v.start - 5, v.end + 5)
    for combo in men.alt_allele_combinations(cand, pic.multi_allelic_mode):
      image = O.build_pileup(pic, cand, window, overlapping, v.start - hw, list(combo), pileup_height=height)
      with torch.no_grad():
        want = weights(torch.from_numpy(image[None])).numpy()[0]
      _, got_alt, got_probs = pw.decode_call_variants_output(cvos[k])
      assert got_alt == [v.alternate_bases.index(a) for a in combo]
      worst = max(worst, float(np.abs(np.array(got_probs) - want).max()))
      k += 1
  assert k == len(cvos) and worst <= 2e-3, worst   # round_gls rounds the emitted values to 1e-10 steps


def test_make_examples_cli_end_to_end(tmp_path):
  """`python -m deepvariant_amd.make_examples` on files: the golden region's reads written back to
  a BAM, the reference stretch to a FASTA, two tasks of a sharded run -> the 84 golden images,
  each in the shard its region belongs to, plus example_info.json; then the fused route
  (--call_variants_outfile + --checkpoint) gives one CallVariantsOutput per example whose
  probabilities equal classifying the written examples with `deepvariant_amd.call_variants`."""
  import json
  import os
  from deepvariant_amd import call_variants as cv
  from deepvariant_amd import genomics_io
  from deepvariant_amd import make_examples as me
  from deepvariant_amd import protowire as pw
  from deepvariant_amd import tfrecord
  from tests import golden_io
  from tests import realigner_fixture as RF
  ref, sets = RF.load()
  stretch_start = 9_995_000
  fasta = str(tmp_path / 'ref.fa.gz')
  genomics_io.write_fasta(fasta, [('chr20', 'N' * stretch_start + ref.get_bases('chr20', stretch_start, 10_100_600))])
  bam = str(tmp_path / 'reads.bam')
  genomics_io.write_bam(bam, [('chr20', 10_100_600)], sets['wgs'], sample_name='NA12878')
  common = ['--ref', fasta, '--reads', bam, '--regions', 'chr20:10,000,000-10,010,000', '--sample_name', 'NA12878',
            '--channel_list', ','.join(T.PILEUP_CHANNELS_WITH_INSERT_SIZE)]
  spec = str(tmp_path / 'examples.tfrecord@2.gz')
  images = {}
  for task in (0, 1):
    assert me.main(common + ['--examples', spec, '--task', str(task)]) == 0
    path = str(tmp_path / ('examples.tfrecord-%05d-of-00002.gz' % task))
    for rec in tfrecord.read_tfrecords(path, verify_crc=True):
      ex = pw.decode_example(rec)
      v = pw.decode_variant(ex['variant/encoded'][0])
      assert ((v.start - 9_999_999) // 1000) % 2 == task                # round-robin over 1 kb regions
      assert v.calls[0].call_set_name == 'NA12878' and 'AD' in v.calls[0].info
      idx = tuple(pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0]))
      images[(v.start, tuple(v.alternate_bases[i] for i in idx))] = np.frombuffer(
          ex['image/encoded'][0], np.uint8).reshape(ex['image/shape'])
  info = json.load(open(str(tmp_path / 'examples.tfrecord-00000-of-00002.gz.example_info.json')))
  assert info['shape'] == [100, 221, 7] and info['channels'] == [1, 2, 3, 4, 5, 6, 19]
  _, golden, _ = golden_io.load(os.path.join(os.path.dirname(__file__), 'golden', 'illumina_wgs_chr20.npz'))
  assert len(images) == len(golden) == 84
  for ex in golden:
    assert np.array_equal(images[(ex['call'].variant.start, tuple(ex['alt_alleles']))], ex['image'])
  # candidate_sweep mode: the reference's golden.candidate_positions, byte for byte
  sweep = str(tmp_path / 'positions.bin')
  assert me.main(common + ['--mode', 'candidate_sweep', '--candidate_positions', sweep]) == 0
  with np.load(RF.FIXTURE) as f:
    assert np.fromfile(sweep, np.int32).tolist() == f['wgs_candidate_positions'].tolist()
  # fused: no tf.Examples
  cvo_path = str(tmp_path / 'cvo.tfrecord.gz')
  assert me.main(common + ['--call_variants_outfile', cvo_path, '--checkpoint', 'random:7']) == 0
  fused = {}
  for rec in tfrecord.read_tfrecords(cvo_path):
    variant, alt, probs = pw.decode_call_variants_output(rec)
    fused[(variant.start, tuple(alt))] = probs
  assert len(fused) == 84
  two_step = str(tmp_path / 'cvo2.tfrecord.gz')
  assert cv.main(['--examples', spec, '--outfile', two_step, '--checkpoint', 'random:7']) == 0
  import glob
  n = 0
  for path in glob.glob(str(tmp_path / 'cvo2*')):          # call_variants shards its output name itself
    for rec in tfrecord.read_tfrecords(path):
      variant, alt, probs = pw.decode_call_variants_output(rec)
      assert np.allclose(fused[(variant.start, tuple(alt))], probs, atol=1e-6)
      n += 1
  assert n == 84


@pytest.mark.timeout(900)
def test_make_examples_two_ranks_sharing_the_gpu(tmp_path):
  """The product multi-rank driver on real hardware: `make_examples --gpus 1 --ranks_per_gpu 2`
  spawns two host processes that share this GPU (own model each, regions i % 2 == r, fused route),
  gathers their CallVariantsOutput records once and lets rank 0 write both shards.  Each shard
  must be byte-identical to what an independent `--task r` run of the same command writes (the
  reference's way of producing them: scripts/run_deepvariant.py:457-462)."""
  from deepvariant_amd import genomics_io
  from deepvariant_amd import make_examples as me
  from deepvariant_amd import tfrecord
  from tests import realigner_fixture as RF
  ref, sets = RF.load()
  stretch_start = 9_995_000
  fasta = str(tmp_path / 'ref.fa')
  genomics_io.write_fasta(fasta, [('chr20', 'N' * stretch_start + ref.get_bases('chr20', stretch_start, 10_100_600))])
  bam = str(tmp_path / 'reads.bam')
  genomics_io.write_bam(bam, [('chr20', 10_100_600)], sets['wgs'], sample_name='NA12878')
  common = ['--ref', fasta, '--reads', bam, '--regions', 'chr20:10,000,000-10,010,000', '--sample_name', 'NA12878',
            '--channel_list', ','.join(T.PILEUP_CHANNELS_WITH_INSERT_SIZE), '--checkpoint', 'random:7']
  tasks = str(tmp_path / 'tasks.cvo.tfrecord@2.gz')
  for task in (0, 1):
    assert me.main(common + ['--call_variants_outfile', tasks, '--task', str(task)]) == 0
  ranks = str(tmp_path / 'ranks.cvo.tfrecord@2.gz')
  assert me.main(common + ['--call_variants_outfile', ranks, '--gpus', '1', '--ranks_per_gpu', '2']) == 0
  total, both = 0, []
  for r in (0, 1):
    want = list(tfrecord.read_tfrecords(str(tmp_path / ('tasks.cvo.tfrecord-%05d-of-00002.gz' % r))))
    got = list(tfrecord.read_tfrecords(str(tmp_path / ('ranks.cvo.tfrecord-%05d-of-00002.gz' % r))))
    assert got == want and len(got) > 10
    total += len(got)
    both += got
  assert total == 84
  # ... and the same records as ONE process writes for the whole region list: a candidate's probabilities do not
  # depend on the rank layout or on which other examples share its batch (the corrections are the checkpoint's)
  one = str(tmp_path / 'one.cvo.tfrecord.gz')
  assert me.main(common + ['--call_variants_outfile', one]) == 0
  assert sorted(tfrecord.read_tfrecords(one)) == sorted(both)


@pytest.mark.timeout(900)
def test_make_examples_from_the_cram_reproduces_the_goldens(tmp_path):
  """deepvariant/make_examples_test.py:330-369 (TestConditions.USE_CRAM): the same golden examples from
  the CRAM form of the NA12878 slice, decoded against --ref (deepvariant_amd/cram_reader.py): all 84
  golden images of chr20:10,000,000-10,010,000, bit for bit."""
  from deepvariant_amd import genomics_io
  from deepvariant_amd import make_examples as me
  from deepvariant_amd import protowire as pw
  from deepvariant_amd import tfrecord
  golden_dir = os.path.join(os.path.dirname(__file__), 'golden')
  with np.load(os.path.join(golden_dir, 'cram.npz')) as z:
    cram = str(tmp_path / 'NA12878_S1.chr20.10_10p1mb.cram')
    with open(cram, 'wb') as f:
      f.write(z['na12878_cram'].tobytes())
  with np.load(os.path.join(golden_dir, 'na12878_100kb.npz')) as z:
    fasta = str(tmp_path / 'ref.fa')
    genomics_io.write_fasta(fasta, [('chr20', 'N' * int(z['ref_start'][0]) + z['ref_bases'].tobytes().decode())])
  out = str(tmp_path / 'examples.tfrecord.gz')
  assert me.main(['--ref', fasta, '--reads', cram, '--regions', 'chr20:10,000,000-10,010,000', '--sample_name', 'NA12878',
                  '--channel_list', ','.join(T.PILEUP_CHANNELS_WITH_INSERT_SIZE), '--examples', out]) == 0
  images = {}
  for rec in tfrecord.read_tfrecords(out, verify_crc=True):
    ex = pw.decode_example(rec)
    v = pw.decode_variant(ex['variant/encoded'][0])
    idx = tuple(pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0]))
    images[(v.start, tuple(v.alternate_bases[i] for i in idx))] = np.frombuffer(
        ex['image/encoded'][0], np.uint8).reshape(ex['image/shape'])
  _, golden, _ = golden_io.load(os.path.join(golden_dir, 'illumina_wgs_chr20.npz'))
  assert len(images) == len(golden) == 84
  for ex in golden:
    assert np.array_equal(images[(ex['call'].variant.start, tuple(ex['alt_alleles']))], ex['image'])


@pytest.mark.timeout(900)
def test_table_path_writes_what_the_object_path_writes(tmp_path, monkeypatch):
  """make_examples keeps a region's reads as a packed table from the BAM decoder to the encoder
  where the configuration allows it (RegionProcessor.table_path_ok); DV_REGION_OBJECTS=1 forces the
  Read-object path.  Both must write byte-identical tf.Examples and CallVariantsOutputs (the goldens
  above pin the table path; this pins the object path to it on a longer stretch)."""
  from deepvariant_amd import genomics_io
  from deepvariant_amd import make_examples as me
  from deepvariant_amd import tfrecord
  golden_dir = os.path.join(os.path.dirname(__file__), 'golden')
  with np.load(os.path.join(golden_dir, 'na12878_100kb.npz')) as z:
    bam = str(tmp_path / 'reads.bam')
    with open(bam, 'wb') as f:
      f.write(z['bam'].tobytes())
    with open(bam + '.bai', 'wb') as f:
      f.write(z['bai'].tobytes())
    fasta = str(tmp_path / 'ref.fa')
    genomics_io.write_fasta(fasta, [('chr20', 'N' * int(z['ref_start'][0]) + z['ref_bases'].tobytes().decode())])
  common = ['--ref', fasta, '--reads', bam, '--regions', 'chr20:10,020,000-10,040,000', '--sample_name', 'NA12878',
            '--channel_list', ','.join(T.PILEUP_CHANNELS_WITH_INSERT_SIZE)]
  outs = {}
  for name, objects in (('tables', False), ('objects', True)):
    if objects:
      monkeypatch.setenv('DV_REGION_OBJECTS', '1')
    else:
      monkeypatch.delenv('DV_REGION_OBJECTS', raising=False)
    ex = str(tmp_path / (name + '.examples.tfrecord.gz'))
    cvo = str(tmp_path / (name + '.cvo.tfrecord.gz'))
    assert me.main(common + ['--examples', ex]) == 0
    # calibration ON (the default): the two routes classify in different batch sizes -- 256 at a time against one
    # region at a time -- and still agree bit for bit, because the corrections come from the checkpoint's fixed
    # calibration set, not from the run's first examples
    assert me.main(common + ['--call_variants_outfile', cvo, '--checkpoint', 'random:7']) == 0
    outs[name] = (list(tfrecord.read_tfrecords(ex)), list(tfrecord.read_tfrecords(cvo)))
  assert outs['tables'][0] == outs['objects'][0] and len(outs['tables'][0]) > 40
  assert outs['tables'][1] == outs['objects'][1] and len(outs['tables'][1]) == len(outs['tables'][0])
