"""Host mirror of the reference's threshold candidate caller, single-sample form.

  VariantCaller(options)                       deepvariant/variant_calling.h:98-240
  .call_variant(allele_count)                  deepvariant/variant_calling.cc:622-671 (CallVariant)
  .calls_from_allele_counts(allele_counts)     :370-382
  .calls_from_allele_counter(counter)          :365-368
  select_alt_alleles / is_good_alt_allele      :232-258
  calc_ref_bases / make_alt_allele / build_allele_map / add_read_depths / add_supporting_reads
                                               :165-230, :260-340, :673-713

It consumes the AlleleCounts the device counter produces (deepvariant_amd/allelecounter.py) and
emits DeepVariantCall objects with the allele_support read-name lists the pileup encoder's
support channel is computed from -- the candidates of make_examples' "calling" mode for one
sample.  With one sample, no complex-allele creation and no methylation-aware options the
multi-sample caller the reference runs in production (variant_calling_multisample.cc) reduces
to exactly these rules (AlleleFilter :264-311 falls back to IsGoodAltAllele when there is no
other sample).  Not restated: multi-sample filtering, complex alleles, gVCF / reference-site
sampling (fraction_reference_sites_to_emit), methylation statistics, CallsFromVcf.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

from deepvariant_amd import allelecounter as ac
from deepvariant_amd import dv_types as T

K_SUPPORTING_UNCALLED_ALLELE = 'UNCALLED_ALLELE'
K_NO_ALT_ALLELE = '.'
_CANONICAL = frozenset('ACGT')


class VariantCallerOptions:
  def __init__(self, min_count_snps=0, min_count_indels=0, min_fraction_snps=0.0, min_fraction_indels=0.0,
               sample_name='', fraction_reference_sites_to_emit=0.0, track_ref_reads=False):
    self.min_count_snps, self.min_count_indels = min_count_snps, min_count_indels
    self.min_fraction_snps, self.min_fraction_indels = min_fraction_snps, min_fraction_indels
    self.sample_name = sample_name
    self.fraction_reference_sites_to_emit = fraction_reference_sites_to_emit
    self.track_ref_reads = track_ref_reads


def _deletion_size(allele) -> int:
  return len(allele.bases) if allele.type == ac.DELETION else -1


def calc_ref_bases(ref_bases: str, alt_alleles: Sequence) -> str:
  """CalcRefBases (:165-191): the longest deletion decides the variant's reference bases."""
  if not alt_alleles:
    return ref_bases
  longest = alt_alleles[0]
  for a in alt_alleles[1:]:               # std::max_element: the FIRST of equal maxima
    if _deletion_size(longest) < _deletion_size(a):
      longest = a
  if longest.type != ac.DELETION:
    return ref_bases
  if len(longest.bases) <= 1:
    raise ValueError('Saw invalid deletion allele with too few bases')
  return ref_bases + longest.bases[1:]


def make_alt_allele(prefix: str, variant_ref: str, from_: int) -> str:
  """MakeAltAllele (:224-229)."""
  return prefix + (variant_ref[from_:] if from_ < len(variant_ref) else '')


def _allele_order(allele):
  return (allele.type, allele.bases)        # OrderAllele, variant_calling.h:83-94


def build_allele_map(alt_alleles: Sequence, ref_bases: str) -> Dict:
  """BuildAlleleMap (:260-296) -> {(type, bases): variant allele}, iterated in OrderAllele order."""
  out = {}
  for a in sorted(alt_alleles, key=_allele_order):
    if a.type in (ac.SUBSTITUTION, ac.INSERTION):
      out[_allele_order(a)] = make_alt_allele(a.bases, ref_bases, 1)
    elif a.type == ac.DELETION:
      if len(a.bases) <= 1:
        raise ValueError('Saw invalid deletion allele with too few bases')
      out[_allele_order(a)] = make_alt_allele(a.bases[:1], ref_bases, len(a.bases))
    elif a.type == ac.SOFT_CLIP:
      continue
    else:
      raise ValueError('Unexpected alt allele')
  return out


def _simplify_ref_alt(ref: str, alt: str) -> str:
  """nucleus SimplifyRefAlt: shared suffix removed (keeping one base), as "ref->alt"."""
  shortest = min(len(ref), len(alt))
  n = 0
  while n < shortest - 1 and ref[len(ref) - 1 - n] == alt[len(alt) - 1 - n]:
    n += 1
  return '%s->%s' % (ref[:len(ref) - n], alt[:len(alt) - n])


def _worth_looking_at(allele_counter, min_count: int = 0):
  """The AlleleCounts a caller has to visit: without read alleles there is no alternate allele,
  and with fewer than `min_count` reads carrying one no allele reaches the caller's count
  threshold (the device counter answers both from its event arrays)."""
  narrowed = getattr(allele_counter, 'counts_with_alt_support', None)
  if narrowed is not None and min_count > 1:
    return narrowed(min_count)
  sparse = getattr(allele_counter, 'counts_with_read_alleles', None)
  return sparse() if sparse is not None else allele_counter.counts()


class VariantCaller:
  def __init__(self, options: VariantCallerOptions):
    for name in ('min_count_snps', 'min_count_indels', 'min_fraction_snps', 'min_fraction_indels',
                 'fraction_reference_sites_to_emit'):
      if getattr(options, name) < 0:
        raise ValueError('%s must be >= 0' % name)           # CHECK_GE in the constructor
    if options.fraction_reference_sites_to_emit > 0:
      raise NotImplementedError('fraction_reference_sites_to_emit (reference-site sampling)')
    self._options = options
    import numpy as np
    self._min_fraction_f32 = (float(np.float32(options.min_fraction_indels)), float(np.float32(options.min_fraction_snps)))

  # ---- thresholds
  def _min_count(self, allele) -> int:
    return self._options.min_count_snps if allele.type == ac.SUBSTITUTION else self._options.min_count_indels

  def _min_fraction(self, allele) -> float:
    # VariantCallerOptions.min_fraction_* are proto `float` fields: the reference compares the double ratio
    # count / total with the threshold ROUNDED TO FLOAT32 (variant_calling_multisample.cc:250-254).  float32(0.1) is
    # above 0.1, so an allele at exactly 10 % is rejected there; comparing with the Python double would keep it
    # (found by tests/test_reference_calling_cpu.py, which runs the reference's own caller).
    return self._min_fraction_f32[allele.type == ac.SUBSTITUTION]

  def is_good_alt_allele(self, allele, total_count: int) -> bool:
    """IsGoodAltAllele (:232-238)."""
    return (allele.type not in (ac.REFERENCE, ac.SOFT_CLIP) and allele.count >= self._min_count(allele) and
            (1.0 * allele.count) / total_count >= self._min_fraction(allele))

  def select_alt_alleles(self, allele_count) -> List:
    """SelectAltAlleles (:244-258)."""
    total = ac.total_allele_counts(allele_count)
    return [a for a in ac.sum_allele_counts(allele_count) if self.is_good_alt_allele(a, total)]

  # ---- one position
  def call_variant(self, allele_count) -> Optional[T.DeepVariantCall]:
    """CallVariant (:622-671)."""
    if not allele_count.ref_base or any(b not in _CANONICAL for b in allele_count.ref_base):
      return None
    alt_alleles = self.select_alt_alleles(allele_count)
    if not alt_alleles:
      return None                              # (KeepReferenceSite is not restated)
    refbases = calc_ref_bases(allele_count.ref_base, alt_alleles)
    allele_map = build_allele_map(alt_alleles, refbases)
    alternate_bases = sorted(allele_map.values())
    pos = allele_count.position
    variant = T.Variant(reference_name=pos.reference_name, start=pos.position,
                        end=pos.position + len(refbases), reference_bases=refbases,
                        alternate_bases=alternate_bases,
                        calls=[T.VariantCall(call_set_name=self._options.sample_name, genotype=[-1, -1])])
    call = T.DeepVariantCall(variant=variant)
    self._add_read_depths(allele_count, alt_alleles, allele_map, refbases, variant)
    self._add_supporting_reads(allele_count.read_alleles, allele_map, refbases, call)
    return call

  def calls_from_allele_counts(self, allele_counts: Sequence) -> List[T.DeepVariantCall]:
    out = []
    for allele_count in allele_counts:
      call = self.call_variant(allele_count)
      if call is not None:
        out.append(call)
    return out

  def _least_allele_count(self) -> int:
    return min(self._options.min_count_snps, self._options.min_count_indels)

  def calls_from_allele_counter(self, allele_counter) -> List[T.DeepVariantCall]:
    return self.calls_from_allele_counts(_worth_looking_at(allele_counter, self._least_allele_count()))

  def call_positions_from_allele_counter(self, allele_counter) -> List[int]:
    return self.call_positions_from_allele_counts(_worth_looking_at(allele_counter, self._least_allele_count()))

  def call_positions_from_allele_counts(self, allele_counts: Sequence) -> List[int]:
    """CallPositionsFromAlleleCounts / CallVariantPosition (variant_calling_multisample.cc:940-1004):
    the positions that WILL become candidates -- the first pass of track_ref_reads, which tells
    the allele counter where to keep the reference-supporting reads by name."""
    out = []
    for allele_count in allele_counts:
      if not allele_count.ref_base or any(b not in _CANONICAL for b in allele_count.ref_base):
        continue
      if self.select_alt_alleles(allele_count):
        out.append(allele_count.position.position)
    return out

  # ---- annotations
  def _add_read_depths(self, allele_count, alt_alleles, allele_map, refbases, variant):
    """AddReadDepths (:298-351): DP, AD, VAF on the first call."""
    info = variant.calls[0].info
    dp = ac.total_allele_counts(allele_count)
    info['DP'] = T.ListValue(values=[T.Value(int_value=dp)])
    by_simplified = {}
    counts = {_allele_order(a): a.count for a in alt_alleles}
    for key, alt in allele_map.items():
      by_simplified[_simplify_ref_alt(refbases, alt)] = counts[key]
    if len(by_simplified) != len(allele_map):
      raise ValueError('Non-unique alternative alleles!')
    ad = [allele_count.ref_supporting_read_count]
    vaf = []
    for alt in variant.alternate_bases:
      n = by_simplified.get(_simplify_ref_alt(variant.reference_bases, alt), 0)
      ad.append(n)
      vaf.append(1.0 * n / dp if dp > 0 else 0.0)
    info['AD'] = T.ListValue(values=[T.Value(int_value=v) for v in ad])
    info['VAF'] = T.ListValue(values=[T.Value(number_value=v) for v in vaf])

  def _add_supporting_reads(self, read_alleles, allele_map, refbases, call):
    """AddSupportingReads (:673-713)."""
    suffix = ''
    if len(call.variant.reference_bases) > len(refbases):
      suffix = call.variant.reference_bases[len(refbases):]
    for read_name, allele in read_alleles.items():
      if allele.type != ac.REFERENCE:
        alt = allele_map.get(_allele_order(allele))
        key = K_SUPPORTING_UNCALLED_ALLELE if alt is None else alt + suffix
        call.allele_support.setdefault(key, T.SupportingReads()).read_names.append(read_name)
        call.allele_support_ext.setdefault(key, []).append(T.ReadSupport(read_name, bool(allele.is_low_quality)))
      elif self._options.track_ref_reads:
        # REFERENCE read alleles are kept by name only under track_ref_reads, at candidate
        # positions (variant_calling.cc:706, variant_calling_multisample.cc:1231-1247)
        call.ref_support.append(read_name)
        call.ref_support_ext.append(T.ReadSupport(read_name, bool(allele.is_low_quality)))
