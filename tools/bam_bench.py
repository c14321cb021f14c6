#!/usr/bin/env python3
"""Times BAM -> packed read table: native (dv_bam_read_region) vs the Python restatement.
usage: tools/bam_bench.py file.bam contig start end [min_mapq]"""
import sys
import time

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
from deepvariant_amd import genomics_io, packing  # noqa: E402


def main(path, contig, start, end, mapq=0):
  packing.ReadTable.from_bam(path, contig, start, end, min_mapping_quality=mapq)  # warm (loads libs)
  for nt in (1, 4, 8):
    t = time.perf_counter()
    for _ in range(5):
      tab = packing.ReadTable.from_bam(path, contig, start, end, min_mapping_quality=mapq, n_threads=nt)
    dt = (time.perf_counter() - t) / 5
    print('native  %d threads: %.1f ms  (%d reads kept, %.2f M bases)' %
          (nt, dt * 1e3, tab.n_reads, len(tab.bases) / 1e6))
  t = time.perf_counter()
  _, reads = genomics_io.read_bam(path, contig, start, end)
  reads = [r for r in reads if genomics_io.read_satisfies_requirements(r, min_mapping_quality=mapq)]
  packing.ReadTable.from_reads(reads)
  print('python reader + packing: %.1f ms' % ((time.perf_counter() - t) * 1e3))


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]),
       int(sys.argv[5]) if len(sys.argv) > 5 else 0)
