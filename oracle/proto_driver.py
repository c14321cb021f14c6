"""Drives the oracle's PROTO-SHAPED interface from a region in the form make_examples hands over -- test
infrastructure (callers: bench.py's parity leg, __graft_entry__.smoke(), tests/).

`region` = (read table with names, DeepVariantCall-shaped candidates, alt-allele combinations per site, reference
windows per site), as `deepvariant_amd.synth.region_inputs_from_batch` builds it.  For the sites asked for, the
reads are queried the way InMemoryReader::Query does (make_examples_native.cc:802-810: overlapping reads in input
order), turned into Read objects with names, and handed with the candidate (allele_support read-NAME lists) and
the reference window to `oracle.build_pileup` -- which does the reference's string matching, read order and name
sorting itself, so nothing the product packed (support codes, name ranks, read lists) can cancel out of a
comparison.
"""
import numpy as np

from deepvariant_amd import dv_types as T
from oracle import oracle as O


def pileups_of_sites(opts, region, sites):
  """-> (item indices in the product's batch order, [pileup image per item]) for the given site indices."""
  table, cands, combos, windows = region
  first_item = np.concatenate([[0], np.cumsum([len(c) for c in combos])])
  hw = (opts.width - 1) // 2
  pos, end = np.asarray(table.read_pos, np.int64), np.asarray(table.read_end, np.int64)
  seq_off, cig_off = table.read_seq_off, table.read_cigar_off
  bases, quals, cigar = table.bases, table.quals, table.cigar

  def read_object(j):
    j = int(j)
    name, _, number = table.keys[j].rpartition('/')
    s0, s1 = int(seq_off[j]), int(seq_off[j + 1])
    words = cigar[int(cig_off[j]):int(cig_off[j + 1])]
    return T.Read(
        fragment_name=name, read_number=int(number), number_reads=2, fragment_length=int(table.read_frag_len[j]),
        aligned_sequence=bytes(bases[s0:s1]).decode(), aligned_quality=bytes(quals[s0:s1]),
        alignment=T.LinearAlignment(
            position=T.Position('chr1', int(pos[j]), bool(table.read_flags[j] & 1)),
            mapping_quality=int(table.read_mapq[j]),
            cigar=[T.CigarUnit(int(w) & 15, int(w) >> 4) for w in words]))

  items, images = [], []
  for ci in sites:
    v = cands[ci].variant
    lo, hi = v.start - opts.read_overlap_buffer_bp, v.end + opts.read_overlap_buffer_bp
    reads = [read_object(j) for j in np.nonzero((hi > pos) & (lo < end))[0]]
    for k, combo in enumerate(combos[ci]):
      items.append(int(first_item[ci]) + k)
      window = windows[ci] if isinstance(windows[ci], str) else bytes(windows[ci]).decode()
      images.append(O.build_pileup(opts, cands[ci], window, reads, v.start - hw, list(combo)))
  return items, images
