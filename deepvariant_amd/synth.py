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


# --------------------------------------------------------------------------
# Long-read shapes (SURVEY.md 8d "HIFI35" / "ONT50"; BASELINE.json configs 4, 5)
# --------------------------------------------------------------------------
HIFI_CHANNELS = list(T.PILEUP_DEFAULT_CHANNELS) + ['haplotype', 'base_methylation']
ONT_CHANNELS = list(T.PILEUP_DEFAULT_CHANNELS) + ['haplotype']


def longread_options(kind: str = 'hifi') -> T.PileupImageOptions:
  """PACBIO (W=147, 6+haplotype+methylation) / ONT_R104 (W=199, 6+haplotype) encoder
  options: min_mapping_quality 1, sort_by_haplotypes (make_examples_options.py long-read
  model defaults).  The two alt-aligned diff channels of the released models come from
  the realigner and are not generated here."""
  rr = T.ReadRequirements(min_mapping_quality=1, min_base_quality=10,
                          min_base_quality_mode=1)
  o = T.default_options(rr)
  o.height = 100
  o.width = 147 if kind == 'hifi' else 199
  o.channels = list(HIFI_CHANNELS if kind == 'hifi' else ONT_CHANNELS)
  o.num_channels = len(o.channels)
  o.sort_by_haplotypes = True
  return o


def make_longread_batch(n_candidates: int, kind: str = 'hifi', seed: int = SEED,
                        options=None) -> packing.PackedBatch:
  """HIFI35: depth~Poisson(35), BQ 20-93, MAPQ 60, HP {0,1,2} w.p. {.2,.4,.4}, 5mC byte
  per base, sparse indels, '='/'X' CIGARs.  ONT50: depth~Poisson(50) plus 10 % sites
  with 96-160 reads (shuffle path), ~3 % indel events per base (dozens of CIGAR ops
  per read), BQ 5-40.  Reads are window-trimmed (TrimReads) to <= W+40 bases."""
  opts = options or longread_options(kind)
  W, H = opts.width, opts.height
  hw = (W - 1) // 2
  C = len(packing.channel_enums(opts))
  rng = np.random.Generator(np.random.PCG64(seed ^ (0x9e37 if kind == 'hifi' else 0x79b9)))
  hifi = kind == 'hifi'
  indel_rate = 0.002 if hifi else 0.03
  n = n_candidates
  pos = 5000 + 2000 * np.arange(n, dtype=np.int64)
  seg = _ACGT[rng.integers(0, 4, size=(n, 2 * W + 200))]
  seg0 = pos - hw - W // 2 - 50                      # genomic coordinate of seg[:, 0]
  depth = np.clip(rng.poisson(35 if hifi else 50, size=n), 2, 95 if hifi else 94)
  if not hifi:
    deep = rng.random(n) < 0.10
    depth = np.where(deep, rng.integers(96, 161, size=n), depth)
  p_pos, p_mapq, p_flags, p_hp, p_rank = [], [], [], [], []
  seq_chunks, qual_chunks, mod_chunks, cig_all = [], [], [], []
  seq_off, cig_off, read_end = [0], [0], []
  first = [0]
  allele_of = []
  for i in range(n):
    d = int(depth[i])
    alt = _ACGT[(np.searchsorted(_ACGT, seg[i, pos[i] - seg0[i]]) + 1) % 4]
    for _ in range(d):
      start = int(pos[i]) - hw - int(rng.integers(-20, 21))
      span = W + int(rng.integers(0, 41))            # reference bases covered
      g = start - int(seg0[i])
      ref = seg[i, g:g + span]
      carries = rng.random() < 0.5
      ops, bases = [], []
      j = 0
      while j < span:
        run = int(rng.geometric(indel_rate)) if indel_rate > 0 else span
        run = min(run, span - j)
        chunk = ref[j:j + run].copy()
        site = int(pos[i]) - start
        if carries and j <= site < j + run:
          chunk[site - j] = alt
        if hifi:                                     # '=' / 'X' ops
          k0 = 0
          mism = np.nonzero(chunk != ref[j:j + run])[0]
          for m in mism:
            if m > k0:
              ops.append((int(m - k0), 8))
            ops.append((1, 9))
            k0 = int(m) + 1
          if run > k0:
            ops.append((run - k0, 8))
        else:
          ops.append((run, 1))
        bases.append(chunk)
        j += run
        if j >= span:
          break
        ln = int(rng.integers(1, 4))
        if rng.random() < 0.5:                       # deletion
          ln = min(ln, span - j - 1)
          if ln > 0:
            ops.append((ln, 3))
            j += ln
        else:
          ops.append((ln, 2))
          bases.append(_ACGT[rng.integers(0, 4, size=ln)])
      merged = []
      for ln, op in ops:                             # merge neighbours of one kind
        if merged and merged[-1][1] == op:
          merged[-1] = (merged[-1][0] + ln, op)
        else:
          merged.append((ln, op))
      b = np.concatenate(bases)
      q = (rng.integers(20, 94, size=b.size) if hifi
           else rng.integers(5, 41, size=b.size)).astype(np.uint8)
      seq_chunks.append(b)
      qual_chunks.append(q)
      mod_chunks.append(np.where(rng.random(b.size) < 0.1,
                                 rng.integers(0, 256, size=b.size), 0).astype(np.uint8))
      seq_off.append(seq_off[-1] + b.size)
      cig_all.extend((ln << 4) | op for ln, op in merged)
      cig_off.append(cig_off[-1] + len(merged))
      p_pos.append(start)
      read_end.append(start + sum(ln for ln, op in merged if op in (1, 3, 8, 9)))
      p_mapq.append(60 if rng.random() < 0.95 else int(rng.integers(0, 60)))
      fl = packing.DV_READ_REVERSE if rng.random() < 0.5 else 0
      if hifi:
        fl |= packing.DV_READ_HAS_5MC
      p_flags.append(fl)
      u = rng.random()
      p_hp.append(0 if u < 0.2 else (1 if u < 0.6 else 2))
      allele_of.append(1 if carries else 0)
    first.append(first[-1] + d)
  R = len(p_pos)
  table = packing.ReadTable(
      n_reads=R, read_pos=np.array(p_pos, np.int32), read_sort_pos=None,
      read_seq_off=np.array(seq_off, np.uint32), read_cigar_off=np.array(cig_off, np.uint32),
      read_mapq=np.array(p_mapq, np.uint8), read_flags=np.array(p_flags, np.uint8),
      read_frag_len=np.zeros(R, np.int32), read_hp=np.array(p_hp, np.int32),
      read_name_rank=rng.permutation(R).astype(np.uint32), read_aux=None,
      bases=np.concatenate(seq_chunks), quals=np.concatenate(qual_chunks),
      mod_5mc=np.concatenate(mod_chunks) if hifi else None, mod_6ma=None,
      cigar=np.array(cig_all, np.uint32), keys=[], read_end=np.array(read_end, np.int64))
  batch = packing.PackedBatch(table=table, width=W)
  allele_of = np.array(allele_of, np.uint8)
  for i in range(n):
    w0 = int(pos[i]) - hw - int(seg0[i])
    ref_idx = batch.add_ref_window(bytes(seg[i, w0:w0 + W]).decode())
    lo, hi = first[i], first[i + 1]
    batch.add_item(int(pos[i]), int(pos[i]) - hw, ref_idx,
                   np.arange(lo, hi, dtype=np.uint32), allele_of[lo:hi], height=H,
                   out_off=i * H * W * C)
  return batch


def make_longread_workload(kind: str, n: int, seed=None, options=None):
  """(options, packed batch, candidates with alt images, drawn channels, total channels) of the long-read model
  inputs (PACBIO 100x147x10, ONT_R104 100x199x9): items 0..n-1 are the reference-aligned pileups (example i at
  i * H*W*Ct), items n + 2k, n + 2k + 1 the two alt-aligned images of candidate with_alt[k] (every third
  candidate -- the indel share of the PacBio golden, 131 of 401) in scratch space behind the examples;
  dv_merge_alt_channels then fills the two trailing diff channels (FillPileupArray's channel mode,
  deepvariant/pileup_image_native.h:246-271).  bench.py's hifi35 / ont50 step and the calibration set of the
  long-read shapes (calibration_set.py) are drawn from it."""
  opts = options or longread_options(kind)
  H, W = opts.height, opts.width
  c_enc = len(packing.channel_enums(opts))
  Ct = c_enc + 2                                   # + the two alt-aligned diff channels
  gen = make_longread_batch(n, kind, seed=SEED if seed is None else seed, options=opts)
  img_bytes = H * W * Ct
  batch = packing.PackedBatch(table=gen.table, width=W)
  batch.ref_windows_list = gen.ref_windows_list
  off = np.asarray(gen.item_list_off)
  lr, lc = np.asarray(gen.list_read), np.asarray(gen.list_code)
  for i in range(n):
    a, b = off[i], off[i + 1]
    batch.add_item(gen.item_variant_start[i], gen.item_image_start[i], gen.item_ref_idx[i], lr[a:b], lc[a:b],
                   height=H, out_off=i * img_bytes)
  with_alt = list(range(0, n, 3))
  for k, i in enumerate(with_alt):
    a, b = off[i], off[i + 1]
    for j in range(2):
      batch.add_item(gen.item_variant_start[i], gen.item_image_start[i], gen.item_ref_idx[i], lr[a:b], lc[a:b],
                     height=H, out_off=n * img_bytes + (2 * k + j) * img_bytes)
  return opts, batch, with_alt, c_enc, Ct


def region_inputs_from_batch(batch: packing.PackedBatch, options):
  """The same synthetic workload in REGION form -- a read table with read names plus
  DeepVariantCall-shaped candidates with `allele_support` read-name lists -- i.e. what
  make_examples hands to ExamplesGenerator before any packing.  Used by the host-inclusive
  bench mode and the native packer's tests.  (Reads the generator tagged "other alt" at
  bi-allelic sites become plain non-supporting reads: one alt allele cannot list them.)
  -> (table, candidates, alt combinations per candidate, reference windows per candidate)"""
  t = batch.table
  table = packing.ReadTable(**{f.name: getattr(t, f.name) for f in
                               __import__('dataclasses').fields(packing.ReadTable)})
  # names whose tuple<string, int> order (the reference's row tie-break, pileup_image_native.cc:75-102)
  # IS the table's read_name_rank, so the proto-shaped form and the packed form describe one workload
  table.keys = ['r%09d/0' % int(rank) for rank in t.read_name_rank]
  starts = np.asarray(batch.item_variant_start)
  off = np.asarray(batch.item_list_off)
  lr, lc = np.asarray(batch.list_read), np.asarray(batch.list_code)
  wins = batch.ref_windows_list
  cands, combos, windows = [], [], []
  i, n = 0, batch.n_items
  while i < n:
    k = 1
    while i + k < n and starts[i + k] == starts[i]:
      k += 1
    reads = lr[off[i]:off[i + 1]]
    codes = lc[off[i]:off[i + 1]]
    alts = ['C', 'G'] if k == 3 else ['C']
    support = {'C': T.SupportingReads([table.keys[int(r)] for r in reads[codes == 1]])}
    if k == 3:
      support['G'] = T.SupportingReads([table.keys[int(r)] for r in reads[codes == 2]])
    pos = int(starts[i])
    cands.append(T.DeepVariantCall(variant=T.Variant('chr1', pos, pos + 1, 'A', alts),
                                   allele_support=support))
    combos.append([['C'], ['G'], ['C', 'G']] if k == 3 else [['C']])
    windows.append(wins[batch.item_ref_idx[i]])
    i += k
  return table, cands, combos, windows
