"""The candidate caller (deepvariant_amd/variant_calling.py) against the vectors of
deepvariant/variant_calling_test.cc (CallVariant cases; same inputs and expectations)."""
import pytest

from deepvariant_amd import allelecounter as ac
from deepvariant_amd import variant_calling as vc

REF, SUB, INS, DEL, SOFT = 1, 2, 3, 4, 5
SAMPLE, CHR, START = 'MySampleName', 'chr1', 10


def _allele_count(ref, alleles):
  """VariantCallingTest::ConstructAlleleCount (:287-307)."""
  a = ac.AlleleCount(CHR, START, ref)
  counter = 0
  for bases, type_, count in alleles:
    if type_ == REF:
      a.ref_supporting_read_count += count
      counter += count
    else:
      for _ in range(count):
        counter += 1
        a.read_alleles['read_%d' % counter] = ac.Allele(bases, type_, 1)
  return a


def _caller(min_count=0, min_fraction=0.0, **kw):
  return vc.VariantCaller(vc.VariantCallerOptions(
      min_count_snps=kw.get('snps', min_count), min_count_indels=kw.get('indels', min_count),
      min_fraction_snps=kw.get('fsnps', min_fraction), min_fraction_indels=kw.get('findels', min_fraction),
      sample_name=SAMPLE))


def _check(ref, caller, alleles, want_ref=None, want_alts=None, ad=None, dp=None):
  call = caller.call_variant(_allele_count(ref, alleles))
  if want_alts is None:
    assert call is None
    return None
  v = call.variant
  assert (v.reference_name, v.start, v.end) == (CHR, START, START + len(want_ref))
  assert v.reference_bases == want_ref and v.alternate_bases == want_alts
  assert v.calls[0].call_set_name == SAMPLE and v.calls[0].genotype == [-1, -1]
  if ad is not None:
    dp = sum(ad) if dp is None else dp
    info = v.calls[0].info
    assert [x.int_value for x in info['AD'].values] == ad
    assert info['DP'].values[0].int_value == dp
    assert [x.number_value for x in info['VAF'].values] == pytest.approx([1.0 * n / dp for n in ad[1:]])
  return call


def test_no_variant():
  for count in (0, 1, 10, 100):
    for ref in 'ACGT':
      _check(ref, _caller(3), [(ref, REF, count)])
    _check('A', _caller(3), [('ACCCCC', SOFT, count)])


def test_snp():
  for count in (10, 100):
    for ref in 'ACGT':
      for alt in 'ACGT':
        if alt != ref:
          _check(ref, _caller(3), [(alt, SUB, count)], ref, [alt], [0, count])
          _check(ref, _caller(3), [(alt, SUB, count), (ref, REF, count)], ref, [alt], [count, count])


def test_non_canonical_reference_base():
  _check('A', _caller(3), [('C', SUB, 100)], 'A', ['C'], [0, 100])
  _check('N', _caller(3), [('C', SUB, 100)])
  _check('R', _caller(3), [('C', SUB, 100)])


def test_min_count():
  n = 10
  _check('A', _caller(n + 1), [('C', SUB, n)])
  _check('A', _caller(n), [('C', SUB, n)], 'A', ['C'], [0, n])
  _check('A', _caller(n - 1), [('C', SUB, n)], 'A', ['C'], [0, n])
  _check('A', _caller(n), [('C', SUB, n), ('G', SUB, n - 1)], 'A', ['C'], [0, n], 2 * n - 1)
  _check('A', _caller(n), [('C', SUB, n), ('G', SUB, n)], 'A', ['C', 'G'], [0, n, n])
  _check('A', _caller(n), [('C', SUB, n - 1), ('G', SUB, n - 1)])


def test_min_fraction():
  n = 10
  c = _caller(n, 0.1)
  _check('A', c, [('C', SUB, n)], 'A', ['C'], [0, n])
  _check('A', c, [('A', REF, n), ('C', SUB, n)], 'A', ['C'], [n, n])
  _check('A', c, [('A', REF, n * 100), ('C', SUB, n)])
  _check('A', c, [('A', REF, n), ('C', SUB, n * 100)], 'A', ['C'], [n, 100 * n])
  _check('A', c, [('C', SUB, n), ('G', SUB, n)], 'A', ['C', 'G'], [0, n, n])
  _check('A', c, [('C', SUB, n * 100), ('G', SUB, n)], 'A', ['C'], [0, n * 100], n * 101)
  _check('A', c, [('C', SUB, n), ('G', SUB, n * 100)], 'A', ['G'], [0, n * 100], n * 101)
  _check('A', c, [('A', REF, n * 100), ('C', SUB, n), ('G', SUB, n)])


def test_min_fraction_is_a_float32_threshold():
  """VariantCallerOptions.min_fraction_* are proto `float`s: the reference compares the double ratio with the
  threshold rounded to float32 (variant_calling_multisample.cc:250-254).  float32(0.1) > 0.1, so exactly 10 % is
  rejected; float32(0.12) < 0.12, so exactly 12 % is kept.  (Found by running the reference's own caller:
  tests/test_reference_calling_cpu.py.)"""
  _check('A', _caller(2, 0.1), [('A', REF, 27), ('C', SUB, 3)])                                # 3 / 30 == 0.1
  _check('A', _caller(2, 0.1), [('A', REF, 26), ('C', SUB, 3)], 'A', ['C'], [26, 3])           # 3 / 29 > 0.1
  _check('A', _caller(2, 0.12), [('A', REF, 22), ('C', SUB, 3)], 'A', ['C'], [22, 3])          # 3 / 25 == 0.12
  _check('A', _caller(2, findels=0.1), [('A', REF, 45), ('AC', INS, 5)])                       # 5 / 50 == 0.1


def test_min_snp_indel_separately():
  c = _caller(snps=5, indels=10, fsnps=0.1, findels=0.5)
  _check('A', c, [('A', REF, 8), ('C', SUB, 8)], 'A', ['C'], [8, 8])
  _check('A', c, [('A', REF, 8), ('AC', INS, 8)])
  _check('A', c, [('A', REF, 8), ('AC', INS, 10)], 'A', ['AC'], [8, 10])
  _check('A', c, [('A', REF, 8), ('AC', DEL, 8)])
  _check('A', c, [('A', REF, 8), ('AC', DEL, 10)], 'AC', ['A'], [8, 10])
  _check('A', c, [('A', REF, 80), ('C', SUB, 20)], 'A', ['C'], [80, 20])
  _check('A', c, [('A', REF, 80), ('AC', INS, 20)])
  _check('A', c, [('A', REF, 80), ('AC', INS, 80)], 'A', ['AC'], [80, 80])
  _check('A', c, [('A', REF, 80), ('AC', DEL, 20)])
  _check('A', c, [('A', REF, 80), ('AC', DEL, 80)], 'AC', ['A'], [80, 80])


@pytest.mark.parametrize('ref,alleles,want_ref,want_alts,ad', [
    ('A', [('C', SUB, 10), ('G', SUB, 10)], 'A', ['C', 'G'], [0, 10, 10]),
    ('A', [('AC', DEL, 10)], 'AC', ['A'], [0, 10]),
    ('A', [('ACCC', DEL, 10)], 'ACCC', ['A'], [0, 10]),
    ('A', [('ACCCCCCCCC', DEL, 10)], 'ACCCCCCCCC', ['A'], [0, 10]),
    ('A', [('AC', INS, 10)], 'A', ['AC'], [0, 10]),
    ('A', [('ACCC', INS, 10)], 'A', ['ACCC'], [0, 10]),
    ('A', [('ACCC', INS, 10), ('ATGC', DEL, 11)], 'ATGC', ['A', 'ACCCTGC'], [0, 11, 10]),
    ('A', [('AT', DEL, 10), ('ATGC', DEL, 11)], 'ATGC', ['A', 'AGC'], [0, 11, 10]),
    ('A', [('AT', INS, 10), ('ATGC', INS, 11)], 'A', ['AT', 'ATGC'], [0, 10, 11]),
    ('A', [('C', SUB, 10), ('ATGC', DEL, 11)], 'ATGC', ['A', 'CTGC'], [0, 11, 10]),
    ('T', [('AA', DEL, 10)], 'TA', ['A'], [0, 10]),
    ('T', [('AA', INS, 10)], 'T', ['AA'], [0, 10]),
    ('T', [('AA', DEL, 10), ('TA', DEL, 11)], 'TA', ['A', 'T'], [0, 10, 11]),
    ('A', [('C', SUB, 10), ('ATGC', INS, 11)], 'A', ['ATGC', 'C'], [0, 11, 10]),
    ('A', [('C', SUB, 10), ('AA', INS, 11), ('ACAC', INS, 12), ('ATGC', DEL, 13), ('AT', DEL, 14)],
     'ATGC', ['A', 'AATGC', 'ACACTGC', 'AGC', 'CTGC'], [0, 13, 11, 12, 14, 10]),
])
def test_allele_combinations(ref, alleles, want_ref, want_alts, ad):
  _check(ref, _caller(10), alleles, want_ref, want_alts, ad)


def test_read_support():
  n = 5
  call = _check('A', _caller(n, 0.1), [('A', REF, n), ('ACT', INS, n), ('ATG', DEL, n + 1), ('G', SUB, n - 1)],
                'ATG', ['A', 'ACTTG'], [n, n + 1, n], 4 * n)
  assert sorted(call.allele_support) == sorted(['A', 'ACTTG', vc.K_SUPPORTING_UNCALLED_ALLELE])
  assert sorted(call.allele_support['A'].read_names) == sorted('read_%d' % i for i in range(11, 17))
  assert sorted(call.allele_support['ACTTG'].read_names) == sorted('read_%d' % i for i in range(6, 11))
  assert sorted(call.allele_support[vc.K_SUPPORTING_UNCALLED_ALLELE].read_names) == \
      sorted('read_%d' % i for i in range(17, 21))


def test_calls_from_allele_counts_keeps_position_order():
  counts = [_allele_count('A', [('A', REF, 10)]), _allele_count('A', [('C', SUB, 10)]),
            _allele_count('N', [('C', SUB, 10)]), _allele_count('G', [('GT', INS, 10), ('G', REF, 2)])]
  for i, c in enumerate(counts):
    c.position.position = START + i
  calls = _caller(3).calls_from_allele_counts(counts)
  assert [(c.variant.start, c.variant.reference_bases, c.variant.alternate_bases) for c in calls] == \
      [(START + 1, 'A', ['C']), (START + 3, 'G', ['GT'])]


def test_negative_thresholds_are_fatal():
  with pytest.raises(ValueError):
    vc.VariantCaller(vc.VariantCallerOptions(min_count_snps=-1))
