"""Synthetic 30x-Illumina-shaped pileup inputs ("ILLUMINA30").

BASELINE.json's metric is quoted on synthetic 30x Illumina read pileups: there
is no network for real genomes here, so bench.py, smoke() and the full-size
parity tests draw their inputs from this generator.  It emits the packed
`dv_batch` layout directly (deepvariant_amd.packing.PackedBatch).

Per candidate: 221-base window of iid ACGT; depth ~ Poisson(30*160/150) clipped
to [4, 200], plus 2 % deep sites with 96-200 reads (more than the 95 read rows
-> exercises the reference's shuffle-and-truncate path); reads
of length 150 whose start makes them overlap [pos-5, pos+6); 0.3 % base
substitutions; 85 % SNP / 10 % 1-5 bp DEL / 5 % 1-5 bp INS; het:hom 2:1 with
carriers tagged support code 1, 1 % of other reads code 2; 2 % of reads carry a
5-20 bp soft clip; base qualities {2, 11, 25, 37} with p {.01, .04, .10, .85};
MAPQ 60 (92 %), 20-59 (5 %), 0-4 (3 %, rejected at min_mapping_quality 5);
strand 1/2; fragment_length round(N(400, 80)) with random sign; 3 % of SNP
sites are tri-allelic (3 items sharing one read list).
"""
from __future__ import annotations

import numpy as np

from deepvariant_amd import _lib
from deepvariant_amd import dv_types as T
from deepvariant_amd import packing

READ_LEN = 150
_SEG = 640       # reference bases generated per candidate
_SEG_POS = 320   # index of the variant start inside the segment
_ACGT = np.frombuffer(b'ACGT', np.uint8)
SEED = 2101079370  # same constant as pic_options.random_seed


def illumina_options(channels: int = 7, height: int = 100, width: int = 221
                     ) -> T.PileupImageOptions:
  """make_examples defaults for WGS calling (min_mapq 5, min_bq 10)."""
  rr = T.ReadRequirements(min_mapping_quality=5, min_base_quality=10,
                          min_base_quality_mode=1)
  o = T.default_options(rr)
  o.height, o.width = height, width
  if channels == 6:
    o.channels = list(T.PILEUP_DEFAULT_CHANNELS)
  elif channels == 7:
    o.channels = list(T.PILEUP_CHANNELS_WITH_INSERT_SIZE)
  else:
    raise ValueError('channels must be 6 or 7')
  o.num_channels = len(o.channels)
  return o


def make_illumina_batch(n_candidates: int, seed: int = SEED, options=None,
                        mean_depth: float = 32.0, multi_allelic: bool = True,
                        out_channels=None) -> packing.PackedBatch:
  opts = options or illumina_options()
  W, H = opts.width, opts.height
  hw = (W - 1) // 2
  C = out_channels or len(packing.channel_enums(opts))
  rng = np.random.Generator(np.random.PCG64(seed))
  n = n_candidates
  pos = (1000 + 1000 * np.arange(n)).astype(np.int64)
  seg = _ACGT[rng.integers(0, 4, size=(n, _SEG))]
  flat_ref = seg.reshape(-1)

  u = rng.random(n)
  vtype = np.where(u < 0.85, 0, np.where(u < 0.95, 1, 2))  # 0 snp 1 del 2 ins
  vlen = rng.integers(1, 6, size=n)
  hom = rng.random(n) < (1.0 / 3.0)
  tri = (rng.random(n) < 0.03) & (vtype == 0) & multi_allelic
  ref_base_idx = np.searchsorted(_ACGT, seg[:, _SEG_POS])  # ACGT sorted
  alt1 = (ref_base_idx + rng.integers(1, 4, size=n)) % 4
  alt2 = (ref_base_idx + 1 + (alt1 - ref_base_idx - 1 + rng.integers(1, 3, size=n)) % 3) % 4
  ins_bases = _ACGT[rng.integers(0, 4, size=(n, 5))]

  depth = np.clip(rng.poisson(mean_depth, size=n), 4, 200)
  deep = rng.random(n) < 0.02  # pile-ups deeper than the 95 read rows
  depth = np.where(deep, rng.integers(96, 201, size=n), depth)
  R = int(depth.sum())
  cand = np.repeat(np.arange(n), depth)
  first = np.concatenate([[0], np.cumsum(depth)])
  start_rel = rng.integers(-154, 6, size=R)            # relative to pos
  rev = rng.random(R) < 0.5
  um = rng.random(R)
  mapq = np.where(um < 0.92, 60,
                  np.where(um < 0.97, rng.integers(20, 60, size=R),
                           rng.integers(0, 5, size=R))).astype(np.uint8)
  frag = np.rint(rng.normal(400, 80, size=R)).astype(np.int32)
  frag *= np.where(rng.random(R) < 0.5, 1, -1).astype(np.int32)
  # allele carried by each read: 0 ref, 1 alt1, 2 alt2
  ua = rng.random(R)
  frac = np.where(hom[cand], 1.0, 0.5)
  allele = np.where(tri[cand], (ua * 3).astype(np.int64) % 3,
                    (ua < frac).astype(np.int64))
  other = (allele == 0) & (rng.random(R) < 0.01) & ~tri[cand]
  clip = rng.random(R) < 0.02
  clip_len = rng.integers(5, 21, size=R)
  clip_front = rng.random(R) < 0.5

  # plain reads: gather 150 reference bases
  seg_start = _SEG_POS + start_rel                      # index in the segment
  gather = (cand * _SEG + seg_start)[:, None] + np.arange(READ_LEN)[None, :]
  bases = flat_ref[gather].copy()
  # substitutions
  sub = rng.random((R, READ_LEN)) < 0.003
  bases[sub] = _ACGT[rng.integers(0, 4, size=int(sub.sum()))]
  quals = np.array([2, 11, 25, 37], np.uint8)[
      rng.choice(4, size=(R, READ_LEN), p=[0.01, 0.04, 0.10, 0.85])]

  read_pos = (pos[cand] + start_rel).astype(np.int64)
  cig_lists = [None] * R
  vt = vtype[cand]
  vl = vlen[cand]
  site = -start_rel                                     # read index of the site
  covers = (site >= 0) & (site < READ_LEN)
  # SNP carriers
  snp = (vt == 0) & (allele > 0) & covers
  alt_idx = np.where(allele == 1, alt1[cand], alt2[cand])
  bases[np.nonzero(snp)[0], site[snp]] = _ACGT[alt_idx[snp]]
  carrier = snp.copy()
  # indel carriers: rebuilt one by one (about 7 % of reads)
  for r in np.nonzero((vt != 0) & (allele > 0))[0]:
    a = int(site[r]) + 1                                # M bases incl. anchor
    L = int(vl[r])
    ci = int(cand[r])
    if vt[r] == 1:                                      # deletion
      c = READ_LEN - a
      if a < 1 or c < 1:
        continue
      s0 = ci * _SEG + seg_start[r]
      tail0 = ci * _SEG + _SEG_POS + 1 + L
      bases[r, :a] = flat_ref[s0:s0 + a]
      bases[r, a:] = flat_ref[tail0:tail0 + c]
      cig_lists[r] = [(a, 1), (L, 3), (c, 1)]
    else:                                               # insertion
      c = READ_LEN - a - L
      if a < 1 or c < 1:
        continue
      s0 = ci * _SEG + seg_start[r]
      tail0 = ci * _SEG + _SEG_POS + 1
      bases[r, :a] = flat_ref[s0:s0 + a]
      bases[r, a:a + L] = ins_bases[ci, :L]
      bases[r, a + L:] = flat_ref[tail0:tail0 + c]
      cig_lists[r] = [(a, 1), (L, 2), (c, 1)]
    carrier[r] = True
  # soft clips on reads that are still plain 150M
  for r in np.nonzero(clip)[0]:
    if cig_lists[r] is not None:
      continue
    s = int(clip_len[r])
    if clip_front[r]:
      cig_lists[r] = [(s, 5), (READ_LEN - s, 1)]
      bases[r, :s] = _ACGT[rng.integers(0, 4, size=s)]
      read_pos[r] += s
    else:
      cig_lists[r] = [(READ_LEN - s, 1), (s, 5)]
      bases[r, READ_LEN - s:] = _ACGT[rng.integers(0, 4, size=s)]

  n_ops = np.array([1 if c is None else len(c) for c in cig_lists], np.int64)
  cig_off = np.concatenate([[0], np.cumsum(n_ops)]).astype(np.uint32)
  cigar = np.full(int(cig_off[-1]), (READ_LEN << 4) | 1, np.uint32)
  read_end = read_pos + READ_LEN
  for r in np.nonzero(n_ops > 1)[0]:
    o = int(cig_off[r])
    e = int(read_pos[r])
    for k, (ln, op) in enumerate(cig_lists[r]):
      cigar[o + k] = (ln << 4) | op
      if op in (1, 3):
        e += ln
    read_end[r] = e

  flags = rev.astype(np.uint8) * packing.DV_READ_REVERSE
  table = packing.ReadTable(
      n_reads=R, read_pos=read_pos.astype(np.int32), read_sort_pos=None,
      read_seq_off=(np.arange(R + 1, dtype=np.int64) * READ_LEN).astype(np.uint32),
      read_cigar_off=cig_off, read_mapq=mapq, read_flags=flags,
      read_frag_len=frag, read_hp=np.full(R, _lib.DV_HP_NONE, np.int32),
      read_name_rank=rng.permutation(R).astype(np.uint32), read_aux=None,
      bases=bases.reshape(-1), quals=quals.reshape(-1), mod_5mc=None,
      mod_6ma=None, cigar=cigar, keys=[], read_end=read_end)

  batch = packing.PackedBatch(table=table, width=W)
  win0 = _SEG_POS - hw
  img_bytes = H * W * C
  k = 0
  for i in range(n):
    ref_idx = batch.add_ref_window(bytes(seg[i, win0:win0 + W]).decode())
    lo, hi = int(first[i]), int(first[i + 1])
    idx = np.arange(lo, hi, dtype=np.uint32)
    al = allele[lo:hi]
    car = carrier[lo:hi]
    if tri[i]:
      combos = ([1], [2], [1, 2])
    else:
      combos = ([1],)
    for combo in combos:
      codes = np.zeros(hi - lo, np.uint8)
      for a_id in (1, 2):
        m = car & (al == a_id)
        codes[m] = 1 if a_id in combo else 2
      codes[other[lo:hi]] = 2
      batch.add_item(int(pos[i]), int(pos[i]) - hw, ref_idx, idx, codes,
                     height=H, out_off=k * img_bytes)
      k += 1
  return batch
