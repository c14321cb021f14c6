"""The device AlleleCounter (dv_count_alleles, allele_counter.hip) against the oracle (GPU):
the reference's own vectors (deepvariant/allelecounter_test.cc via tests/allelecounter_vectors.py)
and a seeded fuzz over every CIGAR op, qualities around the threshold, N bases, reads
hanging off both ends of the interval and of the contig, duplicate read keys."""
import numpy as np
import pytest

from deepvariant_amd import allelecounter as A
from deepvariant_amd import dv_types as T
from oracle import allelecounter_ref as R
from tests import allelecounter_vectors as V

pytestmark = pytest.mark.gpu
CASES = V.cases()


def _compare(got_counts, oracle):
  assert len(got_counts) == len(oracle.counts)
  for i, (g, w) in enumerate(zip(got_counts, oracle.counts)):
    assert g.ref_base == w.ref_base and g.position.position == w.position, i
    assert g.ref_supporting_read_count == w.ref_supporting_read_count, i
    got = {k: (a.bases, a.type, a.is_low_quality) for k, a in g.read_alleles.items()}
    want = {k: (a.bases, a.type, a.is_low_quality) for k, a in w.read_alleles.items()}
    assert got == want, i


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_reference_vectors(case):
  name, contig, start, end, reads, expected = case
  counter = A.AlleleCounter(V.TestRef(), contig, start, end, min_base_quality=V.MIN_BQ)
  oracle = R.AlleleCounter(V.TestRef(), contig, start, end, min_base_quality=V.MIN_BQ)
  for r in reads:
    counter.add(r)
    oracle.add(r)
  assert counter.n_counted_reads() == len(reads)
  for i, (ac, want) in enumerate(zip(counter.counts(), expected)):
    got = sorted((a.bases, a.type, a.count) for a in A.sum_allele_counts(ac))
    assert got == sorted(want), (name, i)
    assert A.total_allele_counts(ac) == sum(n for _, _, n in want)
  _compare(counter.counts(), oracle)


def test_low_mapq_reads_are_ignored():
  counter = A.AlleleCounter(V.TestRef(), 'chr1', 0, 4, min_mapping_quality=10)
  counter.add(V.make_read('chr1', 0, 'ACGT', ['4M'], mapq=0))
  assert counter.n_counted_reads() == 0
  assert all(A.total_allele_counts(ac) == 0 for ac in counter.counts())


class _Ref:
  def __init__(self, seq):
    self.seq = seq

  def n_bases(self, contig):
    return len(self.seq)

  def get_bases(self, contig, start, end):
    return self.seq[start:end]


def _fuzz_reads(rng, ref, n, lo, hi, long_reads):
  reads = []
  for i in range(n):
    cigar, qlen = [], 0
    for _ in range(int(rng.integers(1, 40 if long_reads else 7))):
      op = int(rng.choice([1, 1, 1, 8, 9, 2, 3, 4, 5, 6, 7]))
      ln = int(rng.integers(1, 200 if long_reads and op in (1, 8, 9) else 9))
      if cigar and cigar[-1].operation == op:
        continue
      cigar.append(T.CigarUnit(op, ln))
      qlen += ln if op in (1, 2, 5, 8, 9) else 0
    if qlen == 0:
      cigar.append(T.CigarUnit(1, 3))
      qlen += 3
    start = int(rng.integers(lo, hi))
    # bases: mostly the reference at the implied position, some substitutions, a few N
    seq, pos = [], start
    for cu in cigar:
      if cu.operation in (1, 8, 9):
        for k in range(cu.operation_length):
          rb = ref.seq[pos + k] if 0 <= pos + k < len(ref.seq) else 'A'
          u = rng.random()
          seq.append(rb if u < 0.9 and rb in 'ACGT' else 'N' if u > 0.985 else 'ACGT'[int(rng.integers(0, 4))])
        pos += cu.operation_length
      elif cu.operation in (2, 5):
        seq += ['ACGTN'[int(rng.integers(0, 5 if rng.random() < 0.1 else 4))] for _ in range(cu.operation_length)]
      elif cu.operation in (3, 4, 7):
        pos += cu.operation_length
    quals = rng.integers(5, 45, size=qlen).astype(np.uint8)
    dup = rng.random() < 0.08 and reads
    reads.append(T.Read(
        fragment_name=reads[int(rng.integers(0, len(reads)))].fragment_name if dup else 'f%d' % i,
        read_number=int(rng.integers(0, 2)), number_reads=2, aligned_sequence=''.join(seq),
        aligned_quality=bytes(quals),
        alignment=T.LinearAlignment(position=T.Position('c', start, bool(rng.integers(0, 2))),
                                    mapping_quality=int(rng.integers(0, 61)), cigar=cigar)))
  return reads


@pytest.mark.parametrize('seed,long_reads,legacy', [(1, False, False), (2, False, True), (3, True, False),
                                                    (4, True, True)])
def test_fuzz_against_the_oracle(seed, long_reads, legacy):
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  seq = seq[:700] + 'NN' + seq[702:2500] + 'N' + seq[2501:]
  ref = _Ref(seq)
  for (start, end), (lo, hi) in (((1000, 2000), (700, 2100)), ((0, 400), (0, 420)), ((5600, 6000), (5300, 5990))):
    reads = _fuzz_reads(rng, ref, 700 if not long_reads else 200, lo, hi, long_reads)
    kw = dict(min_mapping_quality=10, min_base_quality=20, keep_legacy_behavior=legacy)
    counter = A.AlleleCounter(ref, 'c', start, end, **kw)
    oracle = R.AlleleCounter(ref, 'c', start, end, **kw)
    for r in reads:
      counter.add(r)
      oracle.add(r)
    assert counter.n_counted_reads() == oracle.n_reads_counted
    _compare(counter.counts(), oracle)
    n_alleles = sum(len(c.read_alleles) for c in oracle.counts)
    assert n_alleles > 150 and sum(c.ref_supporting_read_count for c in oracle.counts) > 400


@pytest.mark.parametrize('seed,long_reads,legacy', [(11, False, False), (12, True, False), (13, False, True)])
def test_track_ref_reads_against_the_oracle(seed, long_reads, legacy):
  """track_ref_reads with candidate positions (allelecounter.cc:504-512, the second pass of
  make_examples_core.py:2880-2932): at the marked positions reference-supporting reads are kept
  by name as REFERENCE read alleles -- including low-quality ones -- and nowhere else; the
  synthetic reference allele disappears from the sums."""
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=4000))
  ref = _Ref(seq[:1500] + 'N' + seq[1501:])
  start, end = 1000, 2000
  reads = _fuzz_reads(rng, ref, 600 if not long_reads else 150, 700, 2100, long_reads)
  candidates = sorted(set(int(p) for p in rng.integers(start - 20, end + 20, size=160)))   # some outside
  kw = dict(min_mapping_quality=10, min_base_quality=20, keep_legacy_behavior=legacy, track_ref_reads=True,
            candidate_positions=candidates)
  counter = A.AlleleCounter(ref, 'c', start, end, **kw)
  oracle = R.AlleleCounter(ref, 'c', start, end, **kw)
  for r in reads:
    counter.add(r)
    oracle.add(r)
  _compare(counter.counts(), oracle)
  marked = {p - start for p in candidates}
  n_ref = n_low = 0
  for i, c in enumerate(counter.counts()):
    refs = [a for a in c.read_alleles.values() if a.type == A.REFERENCE]
    assert not refs or i in marked
    n_ref += len(refs)
    n_low += sum(a.is_low_quality for a in refs)
    assert all(a.bases == c.ref_base for a in refs)
    assert all(a.type != A.REFERENCE or a.count == sum(not r.is_low_quality for r in refs)
               for a in A.sum_allele_counts(c))                                     # no synthetic one
    assert A.total_allele_counts(c) == c.ref_supporting_read_count + sum(
        1 for a in c.read_alleles.values() if a.type != A.REFERENCE and not a.is_low_quality)
  assert n_ref > 150 and (legacy or n_low > 50)
  # without candidate positions nothing is kept by name, the counts are the same
  plain = A.AlleleCounter(ref, 'c', start, end, min_mapping_quality=10, min_base_quality=20,
                          keep_legacy_behavior=legacy, track_ref_reads=True)
  for r in reads:
    plain.add(r)
  for i, (a, b) in enumerate(zip(plain.counts(), counter.counts())):
    assert a.ref_supporting_read_count == b.ref_supporting_read_count
    kept = {k: v.bases for k, v in b.read_alleles.items() if v.type != A.REFERENCE}
    if i in marked:      # a later read with a duplicate key may replace a substitution by its REFERENCE allele
      assert kept.items() <= {k: v.bases for k, v in a.read_alleles.items()}.items()
    else:
      assert kept == {k: v.bases for k, v in a.read_alleles.items()}


def test_full_range_form():
  """The constructor with full_range (allelecounter.cc:349-369): bases are valid over the
  wider reads interval, counts are reported for the inner interval only."""
  rng = np.random.default_rng(9)
  ref = _Ref(''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=3000)))
  reads = _fuzz_reads(rng, ref, 300, 800, 1700, False)
  counter = A.AlleleCounter(ref, 'c', 1000, 1500, min_base_quality=15, full_range=(900, 1600))
  oracle = R.AlleleCounter(ref, 'c', 1000, 1500, min_base_quality=15, full_range=(900, 1600))
  for r in reads:
    counter.add(r)
    oracle.add(r)
  _compare(counter.counts(), oracle)


def test_golden_region_counts_and_candidates():
  """The Illumina golden region (2,737 raw NA12878 reads around the 78 golden candidates): the
  device counter equals the oracle at every one of the ~9.8 K positions, and the candidate
  caller on top of it reproduces the golden DeepVariantCalls exactly as far as the raw reads
  determine them (72 of 78 calls, 47 with identical allele_support -- the reference realigns
  reads first; tests/test_oracle_golden.py pins the same numbers through the oracle)."""
  from tests import golden_io
  from tests import test_oracle_golden as G
  reads, examples, _ = golden_io.load(G.FIXTURE)
  ref = G._WindowRef(examples)
  lo = min(ex['call'].variant.start for ex in examples)
  hi = max(ex['call'].variant.end for ex in examples)
  kw = dict(min_mapping_quality=5, min_base_quality=10)
  counter = A.AlleleCounter(ref, 'chr20', lo, hi, **kw)
  oracle = R.AlleleCounter(ref, 'chr20', lo, hi, **kw)
  for r in reads:
    counter.add(r)
    oracle.add(r)
  assert counter.n_counted_reads() == oracle.n_reads_counted == len(reads) > 2500
  _compare(counter.counts(), oracle)
  counts = counter.counts()
  assert G.golden_candidate_agreement(examples, lambda pos: counts[pos - lo]) == (78, 72, 47)


def test_operations_longer_than_16_bits():
  """A HiFi / ONT soft clip, insertion or deletion longer than 65,535 bases keeps its full length
  (dv_allele_event.length_type has the 28 bits of a BAM CIGAR length): the allele strings equal the
  oracle's, base for base."""
  rng = np.random.default_rng(9)
  n = 200_000
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=n))
  ref = _Ref(seq)

  def bases(k):
    return ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=k))
  long_clip = V.make_read('c', 1000, bases(70_001) + seq[1000:1060], ['70001S', '60M'], name='clip')
  long_ins = V.make_read('c', 1010, seq[1010:1020] + bases(66_000) + seq[1020:1050], ['10M', '66000I', '30M'],
                         name='ins')
  long_del = V.make_read('c', 1005, seq[1005:1015] + seq[1015 + 67_000:1015 + 67_040], ['10M', '67000D', '40M'],
                         name='del')
  counter = A.AlleleCounter(ref, 'c', 900, 1200)
  oracle = R.AlleleCounter(ref, 'c', 900, 1200)
  for r in (long_clip, long_ins, long_del):
    counter.add(r)
    oracle.add(r)
  _compare(counter.counts(), oracle)
  lengths = sorted(len(a.bases) for c in counter.counts() for a in c.read_alleles.values() if len(a.bases) > 60_000)
  assert lengths == [66_001, 67_001, 70_002]


@pytest.mark.parametrize('seed,long_reads', [(21, False), (22, True), (23, False)])
def test_window_counts_from_events_equal_the_allele_walk(seed, long_reads):
  """AlleleCounter.variant_read_window_counts (the window selector's read-support profile computed
  from the device's event arrays: footprints of the events of alleles seen in >= min_allele_support
  reads, grouped by position / type / text) against the walk over Allele objects it replaces
  (window_selector.variant_reads_candidates_from_allele_counter's loop, window_selector.cc:101-141)
  -- on the fuzz reads: every CIGAR op, low-quality alleles, duplicate read keys, clipped ends."""
  from deepvariant_amd.realigner import window_selector as W
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  ref = _Ref(seq)

  class _ObjectsOnly:        # the same counter without the fast method: forces the walk
    def __init__(self, counter):
      self._c = counter
    def __getattr__(self, name):
      if name == 'variant_read_window_counts':
        raise AttributeError(name)
      return getattr(self._c, name)

  for (start, end), (lo, hi) in (((1000, 2000), (700, 2100)), ((0, 400), (0, 420)), ((5600, 6000), (5300, 5990))):
    reads = _fuzz_reads(rng, ref, 700 if not long_reads else 200, lo, hi, long_reads)
    reads += reads[:40]                                   # the same keys again: later events overwrite
    for support in (0, 1, 2, 3):
      counter = A.AlleleCounter(ref, 'c', start, end, min_mapping_quality=10, min_base_quality=20)
      for r in reads:
        counter.add(r)
      config = W.WindowSelectorOptions(min_allele_support=support)
      fast = W.variant_reads_candidates_from_allele_counter(counter, config)
      slow = W.variant_reads_candidates_from_allele_counter(_ObjectsOnly(counter), config)
      assert fast == slow, support
      assert support > 1 or sum(slow) > 0
    assert counter.variant_read_window_counts(2, True) is None      # the strict filter needs allele totals


@pytest.mark.parametrize('seed,long_reads,track', [(31, False, False), (32, True, False), (33, False, True)])
def test_narrowed_visit_gives_the_same_candidates(seed, long_reads, track):
  """VariantCaller on AlleleCounter.counts_with_alt_support (the positions where enough reads carry
  a good non-reference allele for ANY allele to reach the caller's count threshold, taken from the
  event arrays) against the caller on every position with read alleles: same calls, same allele
  support, on the fuzz reads (duplicate keys, low-quality alleles, soft clips, tracked reference reads)."""
  from deepvariant_amd import variant_calling as V
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  ref = _Ref(seq)

  class _Unnarrowed:
    def __init__(self, counter):
      self._c = counter
    def __getattr__(self, name):
      if name == 'counts_with_alt_support':
        raise AttributeError(name)
      return getattr(self._c, name)

  n_calls = 0
  for (start, end), (lo, hi) in (((1000, 2000), (700, 2100)), ((0, 400), (0, 420)), ((5600, 6000), (5300, 5990))):
    reads = _fuzz_reads(rng, ref, 700 if not long_reads else 200, lo, hi, long_reads)
    reads += reads[:40]
    for min_snps, min_indels in ((2, 2), (3, 2), (2, 4)):
      caller = V.VariantCaller(V.VariantCallerOptions(min_snps, min_indels, 0.12, 0.06, sample_name='s', track_ref_reads=track))

      def counter_for(positions=()):
        c = A.AlleleCounter(ref, 'c', start, end, candidate_positions=positions, min_mapping_quality=10,
                            min_base_quality=20, track_ref_reads=track)
        for r in reads:
          c.add(r)
        return c

      positions = caller.call_positions_from_allele_counter(counter_for())
      assert positions == caller.call_positions_from_allele_counter(_Unnarrowed(counter_for()))
      marked = positions if track else ()
      fast = caller.calls_from_allele_counter(counter_for(marked))
      slow = caller.calls_from_allele_counter(_Unnarrowed(counter_for(marked)))
      assert fast == slow                    # dataclasses: variant, allele_support, *_ext, field by field
      assert [c.variant.start for c in fast] == positions
      narrowed = counter_for().counts_with_alt_support(min(min_snps, min_indels))
      assert len(narrowed) < len(counter_for().counts_with_read_alleles())
      n_calls += len(fast)
  assert n_calls > 20 or long_reads      # (random long reads rarely agree on an allele: the equalities above are the test)


@pytest.mark.parametrize('seed,long_reads', [(41, False), (42, True)])
def test_batch_call_equals_one_call_per_region(seed, long_reads):
  """AlleleCounter.run_batch (dv_count_alleles_batch: the regions' arrays in one staging image, kernels
  back to back, two synchronisations for the batch) against every counter counting alone: the same
  reference-supporting counts, the same events in the same order, the same number of counted reads --
  regions of different sizes, one without reads, one with tracked reference reads at candidate
  positions, one counter that has already run."""
  rng = np.random.default_rng(seed)
  seq = ''.join('ACGT'[int(i)] for i in rng.integers(0, 4, size=6000))
  ref = _Ref(seq[:700] + 'NN' + seq[702:])
  spans = [((1000, 2000), (700, 2100)), ((0, 400), (0, 420)), ((5600, 6000), (5300, 5990)), ((2500, 2600), (2300, 2650)),
           ((3000, 3500), (2900, 3600))]
  read_sets = [_fuzz_reads(rng, ref, 500 if not long_reads else 150, lo, hi, long_reads) for _, (lo, hi) in spans]
  read_sets[3] = []                                                   # a region without reads

  def make(k, **kw):
    (start, end), _ = spans[k]
    c = A.AlleleCounter(ref, 'c', start, end, min_mapping_quality=10, min_base_quality=20, **kw)
    for r in read_sets[k]:
      c.add(r)
    return c

  marked = [1100, 1101, 1500, 1999]
  specs = [dict(), dict(keep_legacy_behavior=True), dict(), dict(), dict(track_ref_reads=True, candidate_positions=marked)]
  specs[0] = dict(track_ref_reads=True, candidate_positions=marked)
  alone = [make(k, **specs[k]) for k in range(len(spans))]
  together = [make(k, **specs[k]) for k in range(len(spans))]
  together[2].n_counted_reads()                                       # already ran: run_batch leaves it alone
  A.AlleleCounter.run_batch(together)
  for a, b in zip(alone, together):
    assert a.n_counted_reads() == b.n_counted_reads()
    assert np.array_equal(a.ref_supporting_read_counts(), b.ref_supporting_read_counts())
    assert np.array_equal(a._events, b._events)                       # pylint: disable=protected-access
    assert a.interval_length() == b.interval_length()
  assert sum(len(c._events) for c in together) > 300                  # pylint: disable=protected-access
  assert (((together[0]._events['length_type'] >> 28) & 7) == A.REFERENCE).any()   # pylint: disable=protected-access
  A.AlleleCounter.run_batch([])                                       # nothing to do
  A.AlleleCounter.run_batch(together)                                 # everything has run
