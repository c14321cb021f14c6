"""The product's host stages against the REFERENCE's own code (oracle/_ref/libdvref.so) on REAL reads: the BAMs the
reference ships under deepvariant/testdata/input, region by region as make_examples walks them.  Runs only where
/root/reference exists (this container); TEST INFRASTRUCTURE.  Results: profiles/r04_reference_real_data.txt.

  python tools/ref_real_data.py [illumina|hg002|pacbio ...]

Per 1000-base calling region:
  counts     the allele counts of oracle/allelecounter_ref.py (the device kernel's checker) == AlleleCounter's, at
             every position (reference base, reference-supporting reads, every read allele)
  calls      the product's candidate caller on those counts == the reference's multi-sample caller with one sample
             (alleles, allele_support read lists, AD / DP / VAF)
  windows    the product's window selector == window_selector.cc (reads supporting a variant per position)
  realigner  the product's window realigner as shipped (native de Bruijn assembly, native FastPassAligner, one call per
             region) == the SAME glue driving the reference's DeBruijnGraph::Build and FastPassAligner::AlignReads:
             candidate haplotypes per window, then every read's new position and CIGAR (Illumina only: the long-read
             presets run without the realigner, as the reference's do)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])

from deepvariant_amd import dv_types as T                        # noqa: E402
from deepvariant_amd import genomics_io                          # noqa: E402
from deepvariant_amd import packing                              # noqa: E402
from deepvariant_amd import variant_calling as vc                # noqa: E402
from deepvariant_amd.realigner import realigner as RL            # noqa: E402
from deepvariant_amd.realigner import utils as U                 # noqa: E402
from deepvariant_amd.realigner import window_selector as WS      # noqa: E402
from oracle import oracle as O                                   # noqa: E402
from tests import realigner_fixture as RF                        # noqa: E402
from tests import test_reference_calling_cpu as TC               # noqa: E402

TESTDATA = '/root/reference/deepvariant/testdata/input/'
DATASETS = {
    # name: (bam, fasta, contig, first base, last base, min mapq, realigner, caller thresholds)
    'illumina': ('NA12878_S1.chr20.10_10p1mb.bam', 'ucsc.hg19.chr20.unittest.fasta.gz', 'chr20', 9_999_999, 10_100_000, 5, True,
                 dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06)),
    # (GRCh37 names its contigs 1..22: the hg19 slice of the same chromosome serves, renamed)
    'hg002': ('HG002_NIST_150bp_downsampled_30x.chr20.10_10p1mb.bam', 'ucsc.hg19.chr20.unittest.fasta.gz:chr20=20', '20', 0, 0, 5, True,
              dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.06)),
    'pacbio': ('test_pacbio.chr20_100kbp_at_9mb.bam', 'grch38.chr20_and_21_10M.fa.gz', 'chr20', 9_000_000, 9_100_000, 1, False,
               dict(min_count_snps=2, min_count_indels=2, min_fraction_snps=0.12, min_fraction_indels=0.12)),
}


class Renamed:
  """A FastaReader whose contig `old` answers to `new`."""

  def __init__(self, reader, old, new):
    self._r, self._old, self._new = reader, old, new

  def n_bases(self, contig):
    return self._r.n_bases(self._old if contig == self._new else contig)

  def get_bases(self, contig, start, end):
    return self._r.get_bases(self._old if contig == self._new else contig, start, end)


class ReferenceBackedRealigner(RL.Realigner):
  """The product's realigner glue (window selection, read assignment, reference padding) with the REFERENCE's
  assembly and aligner underneath."""

  def call_debruijn_graph(self, windows, reads, table=None):
    spans = [U.read_range(r) for r in reads]
    out = []
    for w in windows:
      if w.end - w.start > self.config.ws_config.max_window_size or not self._is_valid(w):
        continue
      ref = self._query(w)
      window_reads = [r for r, s in zip(reads, spans) if U.ranges_overlap(s, w)]
      graph = O.reference_debruijn(ref, window_reads, self.config.dbg_config)
      haplotypes = [ref] if graph is None else graph.candidate_haplotypes()
      if haplotypes and haplotypes != [ref]:
        out.append(RL.CandidateHaplotypes(span=w, haplotypes=haplotypes))
    return out

  def call_fast_pass_aligner(self, assembled_region):
    if not assembled_region.reads:
      return []
    region = assembled_region.region
    contig = region.reference_name
    span = assembled_region.read_span
    ref_start = max(0, min(span.start, region.start) - RL._REF_ALIGN_MARGIN)                  # pylint: disable=protected-access
    ref_end = min(self.ref_reader.n_bases(contig), max(span.end, region.end) + RL._REF_ALIGN_MARGIN)   # pylint: disable=protected-access
    prefix = self._query(U.make_range(contig, ref_start, region.start))
    ref = self._query(region)
    if ref_end <= region.end:
      return assembled_region.reads
    suffix = self._query(U.make_range(contig, region.end, ref_end))
    a = self.config.aln_config
    cfg = dict(match=a.match, mismatch=a.mismatch, gap_open=a.gap_open, gap_extend=a.gap_extend, kmer_size=a.kmer_size,
               read_size=len(assembled_region.reads[0].aligned_sequence), max_num_of_mismatches=a.max_num_of_mismatches,
               realignment_similarity_threshold=a.realignment_similarity_threshold, force_alignment=False,
               ref_prefix_len=len(prefix), ref_suffix_len=len(suffix))
    return O.reference_align_reads(prefix + ref + suffix, contig, ref_start,
                                   [prefix + h + suffix for h in assembled_region.haplotypes], assembled_region.reads, **cfg)


def _facts(r):
  if isinstance(r, dict):
    return (r['name'], r['read_number'], r['position'], tuple(map(tuple, r['cigar'])))
  return (r.fragment_name, r.read_number, r.alignment.position.position,
          tuple((c.operation, c.operation_length) for c in r.alignment.cigar))


def run(name):
  bam, fasta, contig, lo, hi, min_mapq, with_realigner, caller_kw = DATASETS[name]
  fasta, _, rename = fasta.partition(':')
  ref = genomics_io.FastaReader(TESTDATA + fasta)
  if rename:
    ref = Renamed(ref, *rename.split('='))
  margin = 2000 if with_realigner else 60000      # long reads hang far over a 1000-base region
  table = packing.ReadTable.from_bam(TESTDATA + bam, contig, 0, 1 << 40, min_mapping_quality=min_mapq)
  if not hi:
    lo, hi = int(table.read_pos.min()), int(table.read_end.max())
  reads = table.to_reads(contig)
  starts, ends = table.read_pos.astype(np.int64), table.read_end.astype(np.int64)
  caller = vc.VariantCaller(vc.VariantCallerOptions(sample_name='s', **caller_kw))
  shipped = RL.Realigner(RL.realigner_config(), ref)
  checked = ReferenceBackedRealigner(RL.realigner_config(), ref)
  n = dict(regions=0, reads=0, alleles=0, calls=0, window_positions=0, windows=0, realigned=0, moved=0)
  bad = []
  t0 = time.time()
  for start in range(lo, hi, 1000):
    region = T.Range(contig, start, min(start + 1000, hi))
    in_reads = [reads[i] for i in np.nonzero((starts < region.end) & (ends > region.start))[0]]
    if not in_reads:
      continue
    n['regions'] += 1
    n['reads'] += len(in_reads)
    try:
      counts, calls, _ = O.reference_count_and_call(ref, contig, region.start, region.end, in_reads, 's',
                                                    contig_length=ref.n_bases(contig), min_mapping_quality=min_mapq,
                                                    min_base_quality=10, caller=caller_kw, ref_margin=margin)
      counter = TC._oracle_counts(ref, contig, region.start, region.end, in_reads, min_mapping_quality=min_mapq,      # pylint: disable=protected-access
                                  min_base_quality=10)
      n['alleles'] += TC._check_counts(counter, region.start, counts)                                              # pylint: disable=protected-access
      TC._check_calls(caller.calls_from_allele_counts(TC._product_counts(counter, contig)), calls, 's')           # pylint: disable=protected-access
      n['calls'] += len(calls)
      if with_realigner:
        ws = shipped.config.ws_config
        theirs, _ = O.reference_window_candidates(ref, contig, region.start, region.end, in_reads, min_mapq=ws.min_mapq,
                                                  min_base_quality=ws.min_base_quality,
                                                  keep_legacy_behavior=ws.keep_legacy_behavior,
                                                  min_allele_support=ws.min_allele_support,
                                                  enable_strict_insertion_filter=ws.enable_strict_insertion_filter,
                                                  contig_length=ref.n_bases(contig))
        wcounter = RF.OracleAlleleCounter(ref, contig, region.start, region.end, min_mapping_quality=ws.min_mapq,
                                          min_base_quality=ws.min_base_quality)
        for r in in_reads:
          wcounter.add(r)
        mine = WS.variant_reads_candidates_from_allele_counter(wcounter, ws)
        assert list(mine) == theirs.tolist(), 'window selector counts'
        n['window_positions'] += int((theirs > 0).sum())
        with RF.oracle_allele_counter():      # (no GPU here: the selector's counts come from the oracle counter)
          hap_a, out_a = shipped.realign_reads(in_reads, region)
          hap_b, out_b = checked.realign_reads(in_reads, region)
        assert [(h.span.start, h.span.end, list(h.haplotypes)) for h in hap_a] == \
               [(h.span.start, h.span.end, list(h.haplotypes)) for h in hap_b], 'candidate haplotypes'
        fa, fb = [_facts(r) for r in out_a], [_facts(r) for r in out_b]
        assert fa == fb, 'realigned reads: %s' % [(x, y) for x, y in zip(fa, fb) if x != y][:2]
        before = {(r.fragment_name, r.read_number): _facts(r) for r in in_reads}
        n['windows'] += len(hap_a)
        n['realigned'] += len(fa)
        n['moved'] += sum(before[(f[0], f[1])] != f for f in fa)
    except Exception as e:      # pylint: disable=broad-except
      bad.append((start, '%s: %s' % (type(e).__name__, str(e)[:300])))
  print('%-9s %s [%d, %d): %d regions, %d read-region pairs | %d read alleles and %d candidates equal | '
        '%d window-selector positions equal | %d assembled windows, %d realigned reads equal (%d moved) | %d regions differ, %.0f s'
        % (name, bam.split('.')[0], lo, hi, n['regions'], n['reads'], n['alleles'], n['calls'], n['window_positions'],
           n['windows'], n['realigned'], n['moved'], len(bad), time.time() - t0))
  for start, msg in bad[:10]:
    print('   region %d: %s' % (start, msg))
  return len(bad)


if __name__ == '__main__':
  if not os.path.isdir(TESTDATA):
    sys.exit('the reference tree is not here')
  names = sys.argv[1:] or list(DATASETS)
  sys.exit(1 if sum(run(nm) for nm in names) else 0)
