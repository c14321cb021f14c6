"""The region chain's table path (no Read objects between the BAM decoder and the encoder):
packing.ReadTable.take / with_alignments / concat_tables and Realigner.realign_table must give,
row for row, the table that packing the object path's reads gives (Realigner.realign_reads on the
same reads, same order)."""
import dataclasses

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import packing
from deepvariant_amd.realigner import realigner as R
from deepvariant_amd.realigner import utils as U
from tests import realigner_fixture as RF


def _same(a: packing.ReadTable, b: packing.ReadTable, ranks_as_order=True):
  for f in dataclasses.fields(packing.ReadTable):
    x, y = getattr(a, f.name), getattr(b, f.name)
    if f.name == 'read_name_rank' and ranks_as_order:
      # ranks are sort keys: equal up to an order-preserving relabelling
      assert np.array_equal(np.argsort(x, kind='stable'), np.argsort(y, kind='stable'))
      assert np.array_equal(np.unique(x, return_inverse=True)[1], np.unique(y, return_inverse=True)[1])
    elif isinstance(x, np.ndarray):
      assert isinstance(y, np.ndarray) and np.array_equal(x.astype(np.int64), y.astype(np.int64)), f.name
    else:
      assert x == y, f.name


def test_take_and_concat_equal_packing_the_same_reads():
  _, sets = RF.load()
  reads = sets['wgs'][:500]
  table = packing.ReadTable.from_reads(reads)
  rng = np.random.default_rng(3)
  rows = rng.permutation(500)[:137]
  _same(table.take(rows), packing.ReadTable.from_reads([reads[i] for i in rows.tolist()]))
  _same(table.take(np.zeros(0, np.int64)), packing.ReadTable.from_reads([]))
  a, b = np.arange(0, 200), np.arange(350, 500)
  _same(packing.concat_tables([table.take(a), table.take(b)]),
        packing.ReadTable.from_reads([reads[i] for i in a.tolist() + b.tolist()]))


def test_with_alignments_equals_repacking_moved_reads():
  from deepvariant_amd import fast_pass_aligner as F
  _, sets = RF.load()
  reads = sets['wgs'][:60]
  table = packing.ReadTable.from_reads(reads)
  rows, positions, cigars, moved = [3, 17, 59], [], [], list(reads)
  for k, r in enumerate(rows):
    n = len(reads[r].aligned_sequence)
    cigar = [(1, n - 10 - k), (2, 4), (3, 2 + k), (1, 6 + k)]        # M I D M
    pos = reads[r].alignment.position.position + 5 + k
    moved[r] = F.with_alignment(reads[r], pos, cigar)
    positions.append(pos)
    cigars.append(np.array([(ln << 4) | op for op, ln in cigar], np.uint32))
  _same(table.with_alignments(rows, positions, cigars), packing.ReadTable.from_reads(moved))
  assert table.with_alignments([], [], []) is table


@pytest.mark.timeout(900)
@RF.with_oracle_counter
def test_realign_table_equals_realign_reads():
  """The realigner on a table against the realigner on objects, per 1000-base region of the golden
  slice (the goldens themselves pin the object path): same windows, same reads moved, same order."""
  ref, sets = RF.load()
  rl = R.Realigner(R.realigner_config(), ref)
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  n_moved = 0
  for start in range(9_999_999, 10_010_000, 1000):
    region = T.Range('chr20', start, min(start + 1000, 10_010_000))
    in_reads = [r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)]
    table = packing.ReadTable.from_reads(in_reads)
    ch_a, realigned = rl.realign_reads(in_reads, region)
    ch_b, realigned_table = rl.realign_table(table, region)
    assert [(c.span, c.haplotypes) for c in ch_a] == [(c.span, c.haplotypes) for c in ch_b]
    want = packing.ReadTable.from_reads(realigned)
    _same(realigned_table, want)
    n_moved += int((np.sort(want.read_pos) != np.sort(table.read_pos)).sum())
  assert n_moved > 50


@pytest.mark.timeout(900)
@RF.with_oracle_counter
def test_realign_tables_batch_equals_region_by_region():
  """One dv_realign_regions call over every region of the golden slice (the make_examples runner's
  form: all windows of the batch on the native thread pool) against one call per region, with one
  and with several threads: same assembled windows, same haplotypes, same tables."""
  ref, sets = RF.load()
  rl = R.Realigner(R.realigner_config(), ref)
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  regions, tables = [], []
  for start in range(9_999_999, 10_010_000, 1000):
    region = T.Range('chr20', start, min(start + 1000, 10_010_000))
    regions.append(region)
    tables.append(packing.ReadTable.from_reads([r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)]))
  regions.append(T.Range('chr20', 10_020_000, 10_021_000))          # a region without reads rides along
  tables.append(packing.ReadTable.from_reads([]))
  one_by_one = [rl.realign_table(t, r) for t, r in zip(tables, regions)]
  assert sum(len(ch) for ch, _ in one_by_one) > 5
  for threads in (1, 5):
    old, R._NATIVE_THREADS = R._NATIVE_THREADS, threads
    try:
      batch = rl.realign_tables(tables, regions)
      bare = rl.realign_tables(tables, regions, want_haplotypes=False)
    finally:
      R._NATIVE_THREADS = old
    assert len(batch) == len(one_by_one)
    for (ch_a, t_a), (ch_b, t_b), (ch_c, t_c) in zip(one_by_one, batch, bare):
      assert [(c.span, c.haplotypes) for c in ch_a] == [(c.span, c.haplotypes) for c in ch_b]
      _same(t_a, t_b, ranks_as_order=False)
      _same(t_a, t_c, ranks_as_order=False)
      assert ch_c == []


def test_with_alignments_csr_handles_runs_and_empty_input():
  _, sets = RF.load()
  reads = sets['wgs'][:40]
  table = packing.ReadTable.from_reads(reads)
  assert table.with_alignments_csr([], [], np.zeros(1, np.int64), np.zeros(0, np.uint32)) is table
  rows = np.array([0, 1, 39])
  words = np.array([(50 << 4) | 1, (3 << 4) | 3, (51 << 4) | 1, (101 << 4) | 1, (7 << 4) | 5, (94 << 4) | 1], np.uint32)
  off = np.array([0, 3, 4, 6])
  positions = np.array([100, 200, 300])
  got = table.with_alignments_csr(rows, positions, off, words)
  want = table.with_alignments(rows.tolist(), positions.tolist(), [words[0:3], words[3:4], words[4:6]])
  _same(got, want, ranks_as_order=False)
  assert got.read_end[0] == 100 + 104 and got.read_end[1] == 200 + 101 and got.read_end[39] == 300 + 94
  assert np.array_equal(got.cigar[got.read_cigar_off[39]:got.read_cigar_off[40]], words[4:6])
  assert np.array_equal(got.cigar[got.read_cigar_off[2]:got.read_cigar_off[3]],
                        table.cigar[table.read_cigar_off[2]:table.read_cigar_off[3]])


@pytest.mark.timeout(900)
@RF.with_oracle_counter
def test_realign_job_on_an_executor_thread_gives_the_same_tables():
  """start_realign_tables with an executor (the make_examples runner's form: the native call of the
  next batch runs on a worker thread while the current batch is processed) against the blocking call;
  two jobs in flight at once."""
  import concurrent.futures
  ref, sets = RF.load()
  rl = R.Realigner(R.realigner_config(), ref)
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  regions = [T.Range('chr20', s, s + 1000) for s in range(9_999_999, 10_005_999, 1000)]
  tables = [packing.ReadTable.from_reads([r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)])
            for region in regions]
  want = rl.realign_tables(tables, regions)
  with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
    first = rl.start_realign_tables(tables[:3], regions[:3], executor=pool)
    second = rl.start_realign_tables(tables[3:], regions[3:], executor=pool)
    got = first.result() + second.result()
    assert first.result() is first.results                    # a second call does not run anything again
  assert len(got) == len(want)
  for (ch_a, t_a), (ch_b, t_b) in zip(want, got):
    assert [(c.span, c.haplotypes) for c in ch_a] == [(c.span, c.haplotypes) for c in ch_b]
    _same(t_a, t_b, ranks_as_order=False)


@pytest.mark.timeout(900)
@RF.with_oracle_counter
def test_process_tables_equals_process_table_region_by_region():
  """RegionProcessor.process_tables (the runner's form: a batch of regions realigned in one native call,
  their candidates called from counters filled together) against process_table per region."""
  from deepvariant_amd import make_examples_core as mec
  from tests.golden.make_golden import wgs_options
  ref, sets = RF.load()
  options = T.MakeExamplesOptions(pic_options=wgs_options(), sample_options=[T.SampleOptions(name='s')])
  po = mec.RegionProcessorOptions()

  class _NoGenerator(mec.RegionProcessor):        # the candidate half only: no encoder, no device
    def __init__(self):
      from deepvariant_amd import variant_calling
      from deepvariant_amd.realigner import realigner as realigner_module
      self.options, self.ref_reader, self.processor_options = options, ref, po
      self.realigner = realigner_module.Realigner(realigner_module.realigner_config(), ref)
      self.variant_caller = variant_calling.VariantCaller(variant_calling.VariantCallerOptions(
          po.vsc_min_count_snps, po.vsc_min_count_indels, po.vsc_min_fraction_snps, po.vsc_min_fraction_indels,
          sample_name='s', track_ref_reads=po.track_ref_reads))

  proc = _NoGenerator()
  reads = sets['wgs']
  spans = [U.read_range(r) for r in reads]
  regions = [T.Range('chr20', s, s + 1000) for s in range(9_999_999, 10_004_999, 1000)]
  tables = [packing.ReadTable.from_reads([r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)])
            for region in regions]
  together = proc.process_tables(regions, tables)
  n_candidates = 0
  for region, table, (candidates, realigned) in zip(regions, tables, together):
    alone_candidates, alone_table = proc.process_table(region, table)
    assert candidates == alone_candidates
    _same(realigned, alone_table, ranks_as_order=False)
    n_candidates += len(candidates)
  assert n_candidates > 10
