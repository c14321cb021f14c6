"""Host mirror of deepvariant/python/direct_phasing (pybind of DirectPhasing) over the C ABI
(`dv_phase_reads`, csrc/direct_phasing.cpp):

  DirectPhasing(min_alleles_to_phase).phase(candidates, reads) -> [0 | 1 | 2 per read]
  .get_phased_variants(), .graphviz()          direct_phasing.h:114-160

Candidates are DeepVariantCalls with allele_support_ext / ref_support_ext (variant_calling.py
fills them); reads are matched by "<fragment_name>/<read_number>"."""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Sequence

import numpy as np

from deepvariant_amd import _lib

K_UNCALLED_ALLELE = 'UNCALLED_ALLELE'


@dataclasses.dataclass
class PhasedVariant:
  position: int
  phase_1_bases: str
  phase_2_bases: str
  is_first_in_block: bool = False


def read_key(read) -> str:
  return '%s/%d' % (read.fragment_name, read.read_number)


class DirectPhasing:
  def __init__(self, min_alleles_to_phase: int = 1):
    self._min_alleles_to_phase = int(min_alleles_to_phase)
    self._last = None

  def phase(self, candidates: Sequence, reads: Sequence) -> List[int]:
    """PhaseReads: the phase of every read; candidates must be strictly ordered by start."""
    index = {}
    for i, read in enumerate(reads):
      index[read_key(read)] = i                     # InitializeReadMaps: a later duplicate wins
    cands = (_lib.DvPhasingCandidate * max(len(candidates), 1))()
    alleles, bases, support, low_quality, names = [], [], [], [], []
    n_bases = 0
    for i, cand in enumerate(candidates):
      entries = []
      if cand.ref_support_ext:
        entries.append(('', 1, cand.ref_support_ext))
      for allele, infos in cand.allele_support_ext.items():
        if allele != K_UNCALLED_ALLELE:
          entries.append((allele, 0, infos))
      cands[i] = _lib.DvPhasingCandidate(cand.variant.start, cand.variant.end, len(alleles), len(entries))
      for allele, is_ref, infos in entries:
        raw = allele.encode()
        alleles.append(_lib.DvPhasingAllele(n_bases, len(raw), is_ref, len(support), len(infos), 0))
        names.append((i, allele, is_ref))
        bases.append(raw)
        n_bases += len(raw)
        for info in infos:
          support.append(index.get(info.read_name, -1))
          low_quality.append(1 if info.is_low_quality else 0)
    table = (_lib.DvPhasingAllele * max(len(alleles), 1))(*alleles)
    sup = np.ascontiguousarray(support, np.int32)
    lq = np.ascontiguousarray(low_quality, np.uint8)
    phases = np.zeros(max(len(reads), 1), np.int32)
    allele_phases = np.zeros(max(len(alleles), 1), np.int32)
    allele_flags = np.zeros(max(len(alleles), 1), np.uint8)
    dot = C.create_string_buffer(1 << 20)
    _lib.check(_lib.lib().dv_phase_reads(
        cands, len(candidates), table, len(alleles), b''.join(bases), n_bases, sup.ctypes.data, lq.ctypes.data,
        len(support), len(reads), self._min_alleles_to_phase, phases.ctypes.data, allele_phases.ctypes.data,
        allele_flags.ctypes.data, dot, len(dot)))
    self._last = (list(candidates), names, allele_phases, allele_flags, dot.value.decode())
    return [int(p) for p in phases[:len(reads)]]

  def get_phased_variants(self) -> List[PhasedVariant]:
    """GetPhasedVariants (direct_phasing.cc:326-360): sites where both phases got an allele."""
    if self._last is None:
      return []
    candidates, names, allele_phases, allele_flags, _ = self._last
    out = []
    per_candidate = {}
    for k, (i, allele, is_ref) in enumerate(names):
      if allele_phases[k] >= 0:
        per_candidate.setdefault(i, []).append((k, 'REF' if is_ref else allele))
    for i in sorted(per_candidate):
      # vertex order within a site: the reference vertex, then alleles by bases (AddCandidate)
      verts = sorted(per_candidate[i], key=lambda t: (0, '') if names[t[0]][2] else (1, t[1]))
      picked = ['', '']
      first = False
      for k, text in verts:
        if allele_phases[k] == 1:
          picked[0] = text
        elif allele_phases[k] == 2:
          picked[1] = text
        first = bool(allele_flags[k])
      if picked[0] and picked[1]:
        out.append(PhasedVariant(candidates[i].variant.start, picked[0], picked[1], first))
    return out

  def graphviz(self) -> str:
    return self._last[4] if self._last else ''
