"""Vectors of deepvariant/realigner/window_selector_test.py, shared by the CPU tests (allele
counts from the oracle) and the GPU tests (allele counts from the device kernel)."""
from deepvariant_amd import dv_types as T
from deepvariant_amd.realigner import window_selector as ws
from tests import realigner_fixture as RF


def mk(bases, start, cigar, quals=None, mapq=50):
  return T.make_read(bases, start=start, cigar=cigar, quals=quals or [64] * len(bases), mapq=mapq)


def linear_config():        # AlleleCountLinearWindowSelectorTest.setUp, :40-61
  return ws.WindowSelectorOptions(
      min_mapq=20, min_base_quality=20, min_windows_distance=4, region_expansion_in_bp=20,
      window_selector_model=ws.WindowSelectorModel(
          model_type=ws.ALLELE_COUNT_LINEAR,
          allele_count_linear_model=ws.AlleleCountLinearModel(
              bias=0, coeff_soft_clip=0, coeff_substitution=-0.5, coeff_insertion=1, coeff_deletion=1,
              coeff_reference=-0.5, decision_boundary=0)),
      min_allele_support=1)


def threshold_config():     # WindowSelectorTest.setUp, :177-191
  return ws.WindowSelectorOptions(
      min_mapq=20, min_base_quality=20, min_windows_distance=4, region_expansion_in_bp=20,
      window_selector_model=ws.WindowSelectorModel(
          model_type=ws.VARIANT_READS,
          variant_reads_model=ws.VariantReadsThresholdModel(min_num_supporting_reads=1,
                                                            max_num_supporting_reads=10)))


def candidates(config, reads, start=0, end=20, ref=None, counter_cls=None):
  """assertCandidatesFromReadsEquals' left-hand side (:63-84).  `counter_cls` = the oracle's counter
  (CPU tests): swapped in for allelecounter.AlleleCounter while the call runs; None = the device counter."""
  import contextlib
  chrom = reads[0].alignment.position.reference_name
  ref = ref if ref is not None else 'A' * (end - start + 512)
  with (RF.oracle_allele_counter() if counter_cls is not None else contextlib.nullcontext()):
    return ws._candidates_from_reads(config, RF.StringRef(chrom, ref), reads, T.Range(chrom, start, end))


# (config name, reads as argument tuples of mk, expected, kwargs)
CASES = []


def _case(cfg, reads, expected, **kw):
  CASES.append((cfg, reads, expected, kw))


# ---- linear model, :86-170
_case('linear', [('AAGA', 10, '4M')], [])
_case('linear', [('AAGTA', 10, '2M2I1M')], [10, 11, 12, 13])
_case('linear', [('AAA', 10, '2M2D1M')], [12, 13])
_case('linear', [('TGATAC', 10, '2S3M1S')], [])
_case('linear', [('AAGA', 10, '2M1X1M')], [])
_case('linear', [('AAGA', 10, '4M'), ('AAAA', 10, '4M')], [])
_case('linear', [('AAAA', 10, '4M'), ('AAA', 10, '3M1D')], [13])
_case('linear', [('AAGA', 10, '4M'), ('AAA', 10, '3M1D')], [13])
_case('linear', [('AAAA', 10, '4M'), ('AAAAT', 10, '4M1I')], [13, 14])
_case('linear', [('AAAT', 10, '3M1S'), ('AAAAT', 10, '4M1I')], [13, 14])
# ---- read-threshold model, one read, :214-284
_case('threshold', [('AAGA', 10, '4M')], [12])
_case('threshold', [('AAGTA', 10, '2M2I1M')], [10, 11, 12, 13])
_case('threshold', [('AAA', 10, '2M2D1M')], [12, 13])
_case('threshold', [('TGATAC', 10, '2S3M1S')], [8, 9, 10, 11, 12, 13])
_case('threshold', [('AAGA', 10, '2M1X1M')], [12])
_case('threshold', [('AAGA', 10, '4M', [64, 64, 10, 30])], [])
_case('threshold', [('AAGTA', 10, '2M2I1M', [64, 64, 10, 21, 64])], [])
_case('threshold', [('TGATAC', 10, '2S3M1S', [21, 10, 64, 64, 64, 64])], [11, 12, 13])
_case('threshold', [('TGATAC', 10, '2S3M1S', [64, 64, 64, 64, 64, 10])], [8, 9, 10, 11])
_case('threshold', [('AAGA', 10, '2M1X1M', [64, 64, 30, 10])], [12])
# ---- every CIGAR operation, :290-349
for _bases, _cigar, _expected in [
    ('A', '1M', []), ('C', '1M', [10]), ('A', '1X', []), ('C', '1X', [10]), ('A', '1=', []), ('C', '1=', [10]),
    ('A', '1M1D', [11]), ('A', '1M2D', [11, 12]), ('A', '1M3D', [11, 12, 13]), ('A', '1M4D', [11, 12, 13, 14]),
    ('AA', '1M1I', [10, 11]), ('AAA', '1M2I', [9, 10, 11, 12]), ('AAAA', '1M3I', [8, 9, 10, 11, 12, 13]),
    ('AA', '1M1S', [10, 11]), ('AAA', '1M2S', [9, 10, 11, 12]), ('AAAA', '1M3S', [8, 9, 10, 11, 12, 13]),
    ('AA', '1S1M', [9, 10]), ('AAA', '2S1M', [8, 9, 10, 11]), ('AAAA', '3S1M', [7, 8, 9, 10, 11, 12]),
    ('AA', '1M1N1M', []), ('AA', '1M2N1M', []), ('A', '1M1H', []), ('A', '1H1M', []),
    ('AA', '1M1P1M', []), ('AA', '1M2P1M', [])]:       # the C++ counter walks over PADs
  _case('threshold', [(_bases, 10, _cigar)], _expected)
# ---- position invariance, :351-371
for _region_start in range(10):
  for _read_start in range(_region_start, 10):
    _case('threshold', [('AGA', _read_start, '3M')], [_read_start + 1], start=_region_start,
          end=_region_start + 100)
# ---- region 5-8 expands by 20 to 0-28, :373-412
for _start in range(10):
  _case('threshold', [('G', _start, '1M')], [_start] if 0 <= _start < 28 else [], start=5, end=8)
  _case('threshold', [('AA', _start, '1M4D1M')],
        [p for p in range(_start + 1, _start + 5) if 0 <= p < 28], start=5, end=8, ref='A' * 100)
# ---- overlapping events are counted twice, :414-423
_case('threshold', [('AAGACCAAA', 0, '4M2I3M')], [2, 3, 4, 5])


def run_case(case, counter_cls=None, min_mapq=None):
  cfg_name, reads, expected, kw = case
  config = linear_config() if cfg_name == 'linear' else threshold_config()
  if min_mapq is not None:
    config.min_mapq = min_mapq
  got = candidates(config, [mk(*r) for r in reads], counter_cls=counter_cls, **kw)
  assert got == expected, (case, got)
