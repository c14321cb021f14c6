#!/usr/bin/env python3
"""bench.py -- candidate pileups/sec (encode + CNN) on N MI355X of one node.

A "step" is one pass of the hot path over one batch of synthetic ILLUMINA30
input that is already resident in HBM:
    dv_encode_batch  (packed reads -> uint8 [B,100,221,7] pileup tensors)
 -> dv_model_infer   (Inception-v3, fp16 MFMA -> fp32 softmax [B,3])
 -> (N > 1) RCCL all-gather of the per-rank [B,3] probabilities + candidate ids
both through the C ABI of libdvhip.so.  Every rank processes its own shard of
candidates (weak scaling, no data-path collective besides the final gather).

Prints ONE JSON line on rank 0 (see the repo task contract), including
  roofline          conv kernels (MFMA bound): achieved TFLOP/s over the timed
                    region, measured with HIP events around every launch
  roofline_encoder  encoder kernel (HBM bound): algorithmic GB/s
  cpu_baseline      the CPU oracle (C++ encoder restatement + fp32 torch
                    Inception) on a bounded sample, on this host's cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (2:1 sparsity excluded)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--batch', type=int, default=7700,
                  help='candidate sites per step per GPU; ~5 %% more pileups (multi-allelic '
                       'sites give 3) -- 7700 sites fill one 8192-example forward')
  ap.add_argument('--channels', type=int, default=7, choices=[6, 7])
  ap.add_argument('--workload', choices=['illumina30', 'hifi35', 'ont50'], default='illumina30',
                  help="SURVEY.md 8(d) workloads.  'illumina30' (default) is the contract's metric: "
                       "100x221x7, BASELINE configs[1].  'hifi35' (PACBIO model shape 100x147x10) and 'ont50' "
                       "(ONT_R104 shape 100x199x9, pile-ups deeper than the image, ~14 CIGAR ops per read) "
                       'print their own line: long-read channel sets with the two alt-aligned diff channels '
                       '(a third of the candidates carry two alt-aligned images, merged on the device); 1 GPU')
  ap.add_argument('--mode', choices=['resident', 'host', 'alleles', 'bam'], default='resident',
                  help="'resident' (default, the contract's metric): inputs already in HBM. "
                       "'host': host-inclusive -- every step packs the region's candidates and "
                       'reads natively (dv_pack_region), uploads them over PCIe and runs the GPU '
                       'path, double buffered (deepvariant_amd/host_pipeline.py); 1 GPU only')
  ap.add_argument('--procs', type=int, default=1,
                  help="--mode bam: host processes sharing the GPU (make_examples --ranks_per_gpu)")
  ap.add_argument('--repeat', type=int, default=1,
                  help='--mode bam: every calling region of the slice this many times (10 = the work of 1 Mb); the '
                       'decoded BAM block is reused between passes')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--calibration-images', type=int, default=256,
                  help='examples (of another synthetic seed) dv_model_calibrate sees before the timed region; 0 = off')
  ap.add_argument('--no-workloads', action='store_true',
                  help='skip the hifi35 / ont50 lines the default 1-GPU run attaches as "workloads"')
  ap.add_argument('--no-dense', action='store_true',
                  help='skip the second timing with blank-row skipping off (value_dense): profiler runs use it so that '
                       'their per-kernel statistics describe one configuration')
  ap.add_argument('--dense-only', action='store_true',
                  help='blank-row skipping off for the whole run (profiler runs of the dense configuration)')
  ap.add_argument('--parity-sites', type=int, default=4096,
                  help='sites of the timed batch checked against the oracle after the timed region')
  ap.add_argument('--cpu-sample', type=int, default=0,
                  help='candidates in the CPU-baseline sample (0 = auto)')
  return ap.parse_args()


def algorithmic_bytes_per_item(batch, out_channels):
  """H*W*C written + per read (bases + quals + 8*n_cigar + 24)
  + W reference bases + 1 support code per listed read."""
  t = batch.table
  off = np.asarray(batch.item_list_off, np.int64)
  seq_len = (t.read_seq_off[1:].astype(np.int64) - t.read_seq_off[:-1])
  n_cig = (t.read_cigar_off[1:].astype(np.int64) - t.read_cigar_off[:-1])
  per_read = 2 * seq_len + 8 * n_cig + 24 + 1
  lr = np.asarray(batch.list_read, np.int64)
  csum = np.concatenate([[0], np.cumsum(per_read[lr])])
  in_bytes = csum[off[1:]] - csum[off[:-1]]
  heights = np.asarray(batch.item_height, np.int64)
  out_bytes = heights * batch.width * out_channels
  return float((in_bytes + out_bytes + batch.width).mean())


def main():
  if len(sys.argv) == 3 and sys.argv[1] == '--cnn-worker':
    _cnn_worker_main(sys.argv[2])
    return
  args = parse_args()
  if args.mode == 'bam':
    if args.gpus != 1:
      raise SystemExit('--mode bam runs one host process on one GPU')
    bam_mode(args)
    return
  if 'WORLD_SIZE' in os.environ:       # launched by torch.distributed.run: one rank per process
    run_rank(args, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
             int(os.environ['WORLD_SIZE']))
    return
  if args.gpus <= 1:
    run_rank(args, 0, 0, 1)
    return
  # --gpus N without a launcher: spawn the N ranks here, one process per GPU
  # (scripts/run_deepvariant.py:457-462 starts its N make_examples shards the same way).
  import socket
  import torch.multiprocessing as mp
  if torch.cuda.device_count() < args.gpus:
    raise SystemExit('--gpus %d but only %d GPU(s) are visible' %
                     (args.gpus, torch.cuda.device_count()))
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  mp.spawn(_spawned_rank, args=(args, port), nprocs=args.gpus, join=True)


def _spawned_rank(rank, args, port):
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  run_rank(args, rank, rank, args.gpus)


def timed_steps(step, sync_all, warmup, steps):
  """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by
  barrier + synchronize on both sides.  Returns (seconds, last step's result)."""
  for _ in range(warmup):
    step()
  sync_all()
  t0 = time.perf_counter()
  out = None
  for _ in range(steps):
    out = step()
  sync_all()
  return time.perf_counter() - t0, out


def make_gather_step(local_step, ids, world, device):
  """local_step() -> probs [n, 3] of this rank's shard.  For world > 1 the returned step
  also runs the product's gather (deepvariant_amd/dist.gather_call_outputs: every rank
  receives every rank's probabilities + candidate ids) inside the timed step."""
  if world == 1:
    return local_step, None
  import torch.distributed as dist  # pylint: disable=g-import-not-at-top
  from deepvariant_amd import dist as dvd
  counts = dvd.exchange_counts(int(ids.shape[0]), device)   # shard sizes are fixed: once
  state = {}

  def step():
    probs = local_step()
    state['all'] = dvd.gather_call_outputs(probs, ids, counts=counts)
    return probs
  return step, state


def reduce_elapsed(elapsed, n_items, world, device):
  """MAX over ranks of the elapsed time, SUM of the per-rank items."""
  if world == 1:
    return elapsed, float(n_items)
  import torch.distributed as dist  # pylint: disable=g-import-not-at-top
  t = torch.tensor([elapsed], dtype=torch.float64, device=device)
  total = torch.tensor([n_items], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  dist.all_reduce(total, op=dist.ReduceOp.SUM)
  return float(t.item()), float(total.item())


def run_rank(args, rank, local_rank, world):
  if world != args.gpus and world > 1:
    raise SystemExit('--gpus must equal WORLD_SIZE')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a GPU: the HIP hot path has no CPU fallback')
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist  # pylint: disable=g-import-not-at-top
    dist.init_process_group('nccl', device_id=dev)

  from deepvariant_amd import _lib, synth
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.inception_v3 import InceptionV3
  from deepvariant_amd.pileup_image_native import _Encoder

  if args.workload != 'illumina30':
    if world > 1 or args.mode != 'resident':
      raise SystemExit('--workload %s runs in resident mode on one GPU' % args.workload)
    return longread_bench(args, dev, local_rank)

  C = args.channels
  opts = synth.illumina_options(C)
  H, W = opts.height, opts.width
  # Each rank owns a different shard of candidates (seeded by rank).
  generated = synth.make_illumina_batch(args.batch, seed=synth.SEED + rank, options=opts)
  # The timed batch is the PRODUCT's packing (dv_pack_region: per-candidate read query, support
  # codes from allele_support read-name lists) of the workload in the form make_examples hands
  # over -- DeepVariantCall-shaped candidates + a read table with names -- so that the parity
  # sample can drive the oracle with those proto-shaped inputs end to end.
  from deepvariant_amd import packing
  region = synth.region_inputs_from_batch(generated, opts)
  host_batch, _ = packing.pack_region_native(region[0], region[1], region[2], region[3], W,
                                             opts.read_overlap_buffer_bp, H, H * W * C)
  n_items = host_batch.n_items
  dbatch = DeviceBatch(host_batch, dev)
  enc = _Encoder(opts, W, device=local_rank)
  model = InceptionV3((H, W, C), max_batch=min(n_items, 8192),
                      device=local_rank)
  model.init_random(seed=1234)          # same weights on every rank
  calibration = calibrate_model(model, args)
  if args.dense_only:
    model.set_blank_skip(False)
  images = torch.empty((n_items, H, W, C), dtype=torch.uint8, device=dev)
  rows = torch.empty(n_items, dtype=torch.int32, device=dev)
  ids = (torch.arange(n_items, device=dev, dtype=torch.int64) +
         rank * (1 << 24))

  # A real (non-default) HIP stream: dv_model_infer replays the forward as one hipGraph
  # on it (the legacy default stream cannot be captured and runs eagerly).
  work = torch.cuda.Stream(device=dev)
  torch.cuda.set_stream(work)

  if args.mode == 'alleles':
    if world > 1:
      raise SystemExit('--mode alleles runs on one GPU')
    return allele_counting(args, host_batch, dev)
  if args.mode == 'host':
    if world != 1:
      raise SystemExit('--mode host runs on one GPU')
    host_inclusive(args, region, opts, C, enc, model, dev)
    return

  # (Measured and removed in round 5: the next step's encoder on a second stream, two pileup buffers, so that it
  # runs under this step's classifier -- 444 K against 455 K candidates/s same box: the encoder's waves take
  # issue slots and L2 from the MFMA kernels for longer than its own 0.55 ms.)
  # The encoder states how many read rows it drew per image (`rows`); with the reference band on top everything below
  # is zero, which is what the classifier's blank-row skipping needs to know (dv_model_infer_rows: no scan of the images)
  band = int(opts.reference_band_height)

  def local_step():
    dbatch.encode(enc, C, images, rows)
    return model(images, rows_used=rows, rows_add=band)

  step, gathered = make_gather_step(local_step, ids, world, dev)

  def sync_all():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize(dev)

  lib = _lib.lib()
  # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
  elapsed, probs = timed_steps(step, sync_all, args.warmup, args.steps)
  # ---- instrumented pass: the SAME K steps again with a HIP event pair around
  # every kernel launch, on the launch stream.  Event profiling forces eager
  # launches (the timed region replays the forward as one hipGraph), so it is
  # kept out of `value`; the per-kernel durations feed `roofline` only and must
  # agree with profiles/*kernel_stats*.
  lib.dv_set_profiling(1)
  for _ in range(args.steps):
    step()
  sync_all()
  enc_ms = lib.dv_profile_ms(0)
  enc_launches = lib.dv_last_profile_count()
  conv_ms = lib.dv_profile_ms(1)
  conv_launches = lib.dv_last_profile_count()
  other_ms = lib.dv_profile_ms(2)
  lib.dv_set_profiling(0)

  # ---- the same K steps with blank-row skipping OFF (dv_model_set_blank_skip; identical probabilities, bit for
  # bit): `value_dense` / `roofline.dense`, so that the headline can be read with and without the data-dependent part
  thr = model.blank_thresholds(n_items) if n_items <= model.max_batch else None
  dense = None
  if thr is not None and not args.no_dense:
    model.set_blank_skip(False)
    elapsed_d, probs_d = timed_steps(step, sync_all, min(args.warmup, 2), args.steps)
    assert torch.equal(probs_d, probs), 'blank-row skipping changed a probability'
    lib.dv_set_profiling(1)
    for _ in range(args.steps):
      step()
    sync_all()
    lib.dv_profile_ms(0)
    conv_ms_d = lib.dv_profile_ms(1)
    lib.dv_set_profiling(0)
    model.set_blank_skip(True)
    elapsed_d, _ = reduce_elapsed(elapsed_d, n_items, world, dev)
    dense = (elapsed_d, conv_ms_d)

  elapsed, items_per_step = reduce_elapsed(elapsed, n_items, world, dev)
  assert torch.isfinite(probs).all()
  if gathered is not None:   # every rank holds every rank's results
    all_p, all_i = gathered['all']
    assert all_p.shape[0] == int(items_per_step) and all_i.unique().numel() == all_p.shape[0]

  if rank == 0:
    value = items_per_step * args.steps / elapsed
    conv_flops_per_item = 2.0 * model.conv_macs_per_example
    executed = executed_conv_flops(model, (H, W, C), thr, conv_flops_per_item) if thr is not None else None
    exec_flops_per_item = executed['executed_flops_per_candidate'] if executed else conv_flops_per_item
    conv_tflops = (exec_flops_per_item * n_items * args.steps /
                   (conv_ms * 1e-3) / 1e12) if conv_ms > 0 else 0.0
    bytes_per_item = algorithmic_bytes_per_item(host_batch, C)
    enc_gbs = (bytes_per_item * n_items * args.steps / (enc_ms * 1e-3) / 1e9
               if enc_ms > 0 else 0.0)
    conv_traffic, enc_traffic, traffic_note = pmc_traffic(args, n_items) if world == 1 else (None, None, None)
    out = {
        'metric': 'candidate pileups/sec (encode+CNN)',
        'value': value,
        'unit': 'candidates/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8 (encoder) / f16 MFMA, f32 accumulate (CNN)',
        'data': 'synthetic',
        'config': {
            'workload': 'configs[1] shape: synthetic 30x Illumina pileups, '
                        'WGS 100x%dx%d, HIP encode + Inception-v3 MFMA on 1 '
                        'MI355X per rank, random-init weights' % (W, C),
            'candidates_per_step_per_gpu': n_items,
            'reads_per_step_per_gpu': int(host_batch.table.n_reads),
            'parallelism': 'interval shards x%d, all-gather of probs' % world,
            'collective_world_size': (dist.get_world_size() if world > 1 else 1),
            'collective_backend': (dist.get_backend() if world > 1 else None),
        },
        'roofline': {
            'kernel': 'conv kernels: stem_a/stem_b (fused stem), chain_kernel (fused 1x7/7x1 chains), '
                      'imgconv_kernel, conv_mfma_kernel<NB,PT>, conv_resident_kernel (all 94 conv layers)',
            'bound': 'mfma',
            'achieved': conv_tflops,
            'peak': MFMA_F16_PEAK_TFLOPS,
            'unit': 'TFLOP/s',
            'frac': conv_tflops / MFMA_F16_PEAK_TFLOPS,
            'traffic': conv_traffic,
            'traffic_note': traffic_note,
            'flops_per_candidate': conv_flops_per_item,
            'executed_flops_per_candidate': exec_flops_per_item,
            'achieved_note': '`achieved` / `frac` count EXECUTED MFMA FLOPs: the nominal 2 x MACs of the 94 layers minus '
                             'the output tiles blank-row skipping copied instead of computing (mirrored on the host from '
                             "the kernels' tile rules and the per-image thresholds of the timed batch); `dense` = the same "
                             'K steps with the skipping switched off, nominal FLOPs',
            'avg_launch_ms': conv_ms / max(conv_launches, 1),
            'launches': conv_launches,
            'ms_per_step': conv_ms / args.steps,
            'timing': 'HIP event pair per launch, instrumented pass of the same K steps',
        },
        'roofline_encoder': {
            'kernel': 'encode_items_kernel',
            'bound': 'hbm',
            'achieved': enc_gbs,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': enc_gbs / HBM_PEAK_GBS,
            'traffic': enc_traffic,
            'bytes_per_candidate': bytes_per_item,
            'avg_launch_ms': enc_ms / max(enc_launches, 1),
            'launches': enc_launches,
            'candidates_per_s': (n_items * args.steps / (enc_ms * 1e-3)
                                 if enc_ms > 0 else 0.0),
        },
        'other_kernels_ms_per_step': other_ms / args.steps,
    }
    out['calibration'] = calibration
    if dense is not None:
      elapsed_d, conv_ms_d = dense
      dense_tflops = conv_flops_per_item * n_items * args.steps / (conv_ms_d * 1e-3) / 1e12 if conv_ms_d > 0 else 0.0
      out['value_dense'] = items_per_step * args.steps / elapsed_d
      out['ms_per_step_dense'] = 1e3 * elapsed_d / args.steps
      out['roofline']['dense'] = {'achieved': dense_tflops, 'frac': dense_tflops / MFMA_F16_PEAK_TFLOPS,
                                  'ms_per_step': conv_ms_d / args.steps, 'flops_per_candidate': conv_flops_per_item}
      out['blank_row_skipping'] = dict(executed, what=(
          'stem tiles whose receptive field holds only the zero rows below the pile-up are copied from the all-blank '
          "image's response (bit-identical; include/dvhip.h dv_model_set_blank_skip); `value` is the product's default, "
          '`value_dense` the same steps with it off'))
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(host_batch, opts, C, args.cpu_sample)
      out['parity'] = parity_sample(region, opts, C, model, images, probs, n=args.parity_sites)
      out['parity'].update(cnn_tail_parity(model, C, images, probs))
    if world == 1 and not args.no_workloads:
      # BASELINE configs[3], configs[4]: the long-read shapes, timed after the headline with the same
      # K / W, attached to the ONE line (the headline `value` / `config` stay ILLUMINA30)
      out['workloads'] = {}
      for w in ('hifi35', 'ont50'):
        sub = argparse.Namespace(**vars(args))
        sub.workload = w
        full = longread_bench(sub, dev, local_rank, emit=False)
        out['workloads'][w] = {k: full[k] for k in ('metric', 'value', 'value_dense', 'unit', 'ms_per_step',
                                                    'ms_per_step_dense', 'config', 'roofline', 'roofline_encoder',
                                                    'other_kernels_ms_per_step', 'calibration', 'blank_row_skipping',
                                                    'parity', 'precise', 'fast') if k in full}
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


def host_inclusive(args, region, opts, C, enc, model, dev):
  """Candidates/s INCLUDING the host: per step the region's candidates + read table are
  packed natively (dv_pack_region: per-candidate read query, support codes from read-name
  lists), staged into pinned memory, uploaded over PCIe (reads included -- a new region
  brings new reads) and encoded + classified on the GPU; packing/upload of step k+1 overlap
  the GPU work of step k.  Not the contract's `value` (inputs resident in HBM)."""
  from deepvariant_amd import host_pipeline as hp, synth
  H, W = opts.height, opts.width
  table, cands, combos, windows = region
  inputs = hp.RegionInputs(table, cands, combos, windows, W, opts.read_overlap_buffer_bp, H, H * W * C)
  pipe = hp.HostPipeline(inputs, enc, model, C, dev, (H, W, C))
  pipe.run(max(args.warmup, 2))
  torch.cuda.synchronize(dev)
  pipe.pack_seconds = pipe.stage_seconds = 0.0
  t0 = time.perf_counter()
  probs = pipe.run(args.steps)
  torch.cuda.synchronize(dev)
  elapsed = time.perf_counter() - t0
  n_items = pipe.slots[0].n_items
  assert torch.isfinite(probs).all() and probs.shape[0] == n_items
  upload_bytes = sum(int(x.numel()) for x in pipe.slots[0].pinned.values())
  print(json.dumps({
      'metric': 'candidate pileups/sec (host packing + PCIe + encode + CNN)',
      'value': n_items * args.steps / elapsed,
      'unit': 'candidates/s',
      'n_gpus': 1,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps,
      'higher_is_better': True,
      'data': 'synthetic',
      'config': {
          'workload': 'the resident-mode workload (synthetic 30x Illumina, 100x%dx%d) handed over '
                      'in region form: read table with names + candidates with allele_support '
                      'read-name lists' % (W, C),
          'candidates_per_step': n_items,
          'reads_per_step': int(table.n_reads),
          'host_threads': 'one packer thread (dv_pack_region on 8 host threads) + the launching thread',
      },
      'pack_ms_per_step': 1e3 * pipe.pack_seconds / args.steps,
      'staging_ms_per_step': 1e3 * pipe.stage_seconds / args.steps,
      'upload_bytes_per_step': upload_bytes,
  }))


def allele_counting(args, host_batch, dev):
  """SURVEY 8f row f2 (the step in front of the path): AlleleCounter::Add over the bench
  workload's reads with dv_count_alleles -- one wave per read, read table resident in HBM.
  A second metric line, never the contract's `value`.  The reference for the reads is
  rebuilt from the reads themselves (the synthetic batch only carries per-candidate
  windows), so most bases are reference matches, as in real data."""
  import ctypes as C
  from deepvariant_amd import _lib
  from deepvariant_amd.device_batch import DeviceBatch
  t = host_batch.table
  n = t.n_reads
  lo = int(t.read_pos.min())
  hi = int(t.read_end.max()) + 64
  rng = np.random.default_rng(3)
  ref = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, size=hi - lo)].copy()
  ops, lens = t.cigar & 15, t.cigar >> 4
  for r in range(n):                      # untimed setup: lay the reads' matched runs onto the reference
    p, q = int(t.read_pos[r]) - lo, int(t.read_seq_off[r])
    for c in range(int(t.read_cigar_off[r]), int(t.read_cigar_off[r + 1])):
      op, ln = int(ops[c]), int(lens[c])
      if op in (1, 8, 9):
        ref[p:p + ln] = t.bases[q:q + ln]
        p += ln
        q += ln
      elif op in (2, 5):
        q += ln
      elif op in (3, 4, 7):
        p += ln
  ref_bytes = ref.tobytes()
  dbatch = DeviceBatch(host_batch, dev)
  opt = _lib.DvAlleleCounterOptions(lo, hi - 64, lo, hi - 64, ref_bytes, lo, len(ref_bytes), hi, 5, 10, 0)
  lib = _lib.lib()
  stream = torch.cuda.current_stream(dev).cuda_stream

  def step():
    h = C.c_void_p()
    _lib.check(lib.dv_count_alleles(C.byref(dbatch.c), C.byref(opt), C.byref(h), C.c_void_p(stream)))
    return h

  def summary(h):
    refc = C.POINTER(C.c_int32)()
    n_ev, n_cnt = C.c_uint32(), C.c_int32()
    length = lib.dv_allele_counts_arrays(h, C.byref(refc), None, C.byref(n_ev), C.byref(n_cnt))
    total = int(np.ctypeslib.as_array(refc, shape=(length,)).sum())
    lib.dv_allele_counts_free(h)
    return total, int(n_ev.value), int(n_cnt.value)

  for _ in range(max(args.warmup, 1)):
    first = summary(step())
  torch.cuda.synchronize(dev)
  lib.dv_set_profiling(1)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    last = summary(step())
  torch.cuda.synchronize(dev)
  elapsed = time.perf_counter() - t0
  kernel_ms = lib.dv_profile_ms(2)
  launches = lib.dv_last_profile_count()
  lib.dv_set_profiling(0)
  assert last == first                      # atomics only add: the counts are run-to-run identical
  n_bases = int(t.read_seq_off[-1])
  # what the kernel has to touch per read: bases + qualities, CIGAR words, position / offsets /
  # mapq, the reference under every matched base, one 4-byte atomic per reference match
  alg = 2 * n_bases + 4 * int(t.read_cigar_off[-1]) + 13 * n + n_bases + 4 * last[0]
  gbs = alg * launches / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
  print(json.dumps({
      'metric': 'reads/sec (allele counting, dv_count_alleles)',
      'value': n * args.steps / elapsed,
      'unit': 'reads/s',
      'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps,
      'higher_is_better': True, 'data': 'synthetic', 'dtype': 'u8 / int32 atomics',
      'config': {'workload': 'the reads of the resident-mode batch (synthetic 30x Illumina)',
                 'reads_per_step': n, 'bases_per_step': n_bases, 'interval_bases': hi - 64 - lo,
                 'ref_supporting_reads_counted': last[0], 'non_reference_events': last[1],
                 'reads_counted': last[2],
                 'includes': 'kernel + result download + host sort of the events (whole call)'},
      'roofline': {'kernel': 'count_alleles_kernel', 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                   'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': None,
                   'bytes_per_launch': alg, 'avg_launch_ms': kernel_ms / max(launches, 1),
                   'bases_per_s': n_bases * launches / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0},
  }))


def _bam_fixture(tmp):
  """The bundled NA12878 slice as files: (bam path, fasta path)."""
  from deepvariant_amd import genomics_io
  fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'na12878_100kb.npz')
  with np.load(fixture) as z:
    bam = os.path.join(tmp, 'NA12878_S1.chr20.10_10p1mb.bam')
    with open(bam, 'wb') as f:
      f.write(z['bam'].tobytes())
    with open(bam + '.bai', 'wb') as f:
      f.write(z['bai'].tobytes())
    fasta = os.path.join(tmp, 'ref.fa')
    lo = int(z['ref_start'][0])
    genomics_io.write_fasta(fasta, [('chr20', 'N' * lo + z['ref_bases'].tobytes().decode())], index=True)
  return bam, fasta


_BAM_DATA = 'BASELINE.json configs[0]: NA12878 30x Illumina, chr20:10,000,000-10,100,000 (reference testdata, bundled)'


def _bam_args(me, tmp, bam, fasta, regions, out, extra=()):
  return me.build_arg_parser().parse_args([
      '--ref', fasta, '--reads', bam, '--checkpoint', 'random:1234', '--sample_name', 'NA12878',
      '--channel_list', 'BASE_CHANNELS,insert_size', '--regions', regions, '--call_variants_outfile',
      os.path.join(tmp, out)] + list(extra))


def _repeat_calling_regions(me, k):
  """Every calling region of the run k times over (a rank's share, in order, pass after pass): the work of a
  k-times longer interval from the bundled 100 kb slice.  Bench-only: wraps the product's region list."""
  if k <= 1:
    return
  original = me.calling_regions

  def repeated(*a, **kw):
    return list(original(*a, **kw)) * k
  me.calling_regions = repeated


def _bam_rank(rank, world, port, tmp, bam, fasta, repeat=1):
  """One of `--procs R` host processes sharing GPU 0 (make_examples --ranks_per_gpu R): warm up
  alone, then the product's distributed runner over the whole slice between two barriers."""
  import torch.distributed as dist
  from deepvariant_amd import make_examples as me, tfrecord
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  devnull = open(os.devnull, 'w')
  try:
    weights = os.path.join(tmp, 'weights.rank%d.f32' % rank)

    class WarmHooks(me.RunnerHooks):
      def make_model(self, args, options):      # the timed run reads its weights from a file, as a real run does
        model = super().make_model(args, options)   # ('random:1234' draws 24 M numbers on the host: seconds, and
        model.flat_weights.tofile(weights)          # eight ranks doing it at once was most of an earlier wall time)
        # the checkpoint's calibration corrections are cached next to it by the first run that loads it
        # (InceptionV3.calibrate_for_checkpoint); the timed run finds them there, as every later run does
        model.calibrate_for_checkpoint(getattr(args, 'calibration_examples', 256), cache_prefix=weights)
        return model

    me.make_examples_runner(_bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,003,000', 'warm%d.cvo.tfrecord.gz' % rank),
                            log=devnull, hooks=WarmHooks())
    spec = 'cvo.tfrecord@%d.gz' % world
    timed_args = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,100,000', spec,
                           ['--gpus', '1', '--ranks_per_gpu', str(world)])
    timed_args.checkpoint = weights
    _repeat_calling_regions(me, repeat)
    # gloo opens its pairwise connections on the first all-gather (5.6 s for 8 ranks, 19 s for 16, measured): a
    # once-per-job cost of the backend, paid here before the timed region like the kernels' first load
    from deepvariant_amd import dist as dvd
    dvd.gather_records([b'warm-up'], device=None)
    dist.barrier()
    t0 = time.perf_counter()
    stats = me.distributed_runner(timed_args, rank, world, log=devnull)
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0, float(stats['n_regions']), float(stats['n_reads']),
                            float(stats['n_candidates']), float(stats['n_examples'])], dtype=torch.float64)
    wall = torch.tensor([elapsed[0], stats['loop_s'], stats['setup_s'], stats['runner_s'], stats['gather_s'],
                         stats['write_s']], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    dist.all_reduce(elapsed, op=dist.ReduceOp.SUM)
    # the same run once more with a HIP event pair around every kernel launch of every rank: how long the GPU
    # computes (summed over the processes that share it) against the wall time of that run
    from deepvariant_amd import _lib
    lib = _lib.lib()
    lib.dv_set_profiling(1)
    again = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,100,000', 'again.' + spec,
                      ['--gpus', '1', '--ranks_per_gpu', str(world)])
    again.checkpoint = weights
    dist.barrier()
    t1 = time.perf_counter()
    me.distributed_runner(again, rank, world, log=devnull)
    torch.cuda.synchronize()
    dist.barrier()
    wall2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
    kernel = torch.tensor([lib.dv_profile_ms(0), lib.dv_profile_ms(1), lib.dv_profile_ms(2)], dtype=torch.float64)
    lib.dv_set_profiling(0)
    dist.all_reduce(wall2, op=dist.ReduceOp.MAX)
    dist.all_reduce(kernel, op=dist.ReduceOp.SUM)
    if rank == 0:
      from deepvariant_amd import sharded_file_utils
      n_written = sum(sum(1 for _ in tfrecord.read_tfrecords(sharded_file_utils.sharded_filename(os.path.join(tmp, spec), r)))
                      for r in range(world))
      assert n_written == int(elapsed[4]) == stats['n_gathered']
      print(json.dumps({
          'metric': 'examples/sec, BAM + FASTA -> CallVariantsOutput (make_examples fused route), %d host '
                    'processes sharing one GPU (make_examples --gpus 1 --ranks_per_gpu %d)' % (world, world),
          'value': int(elapsed[4]) / float(wall[0]), 'unit': 'examples/s', 'n_gpus': 1, 'host_processes': world,
          'data': _BAM_DATA, 'wall_s': float(wall[0]), 'region_loop_s_max_over_ranks': float(wall[1]),
          'setup_s_max_over_ranks': float(wall[2]), 'runner_s_max_over_ranks': float(wall[3]),
          'record_exchange_s_max_over_ranks': float(wall[4]), 'rank0_shard_files_s': float(wall[5]),
          'examples_per_s_region_loop_only': int(elapsed[4]) / float(wall[1]),
          'regions': int(elapsed[1]), 'reads': int(elapsed[2]),
          'candidates': int(elapsed[3]), 'examples': int(elapsed[4]), 'host_cores': os.cpu_count(),
          'passes_over_the_slice': repeat,
          'gpu_kernel_ms_all_ranks': {'encoder': float(kernel[0]), 'cnn': float(kernel[1]),
                                      'other (allele counts, pools, head)': float(kernel[2])},
          'gpu_busy_frac': float(kernel.sum()) / (1e3 * float(wall2[0])),
          'instrumented_wall_s': float(wall2[0]),
          'note': 'not the contract metric: inputs start in files on the host; wall = max over ranks between two '
                  'barriers, includes every rank building its model and loading weights (setup_s: a fixed cost), '
                  'the final gather of the records and rank 0 writing the %d shard files; gpu_busy_frac = kernel '
                  'time of an instrumented repeat summed over the ranks / wall time of that repeat' % world,
      }), flush=True)
  finally:
    dist.destroy_process_group()


def bam_mode(args, log=sys.stderr):
  """BASELINE.json configs[0] end to end on FILES: the reference tree's NA12878 30x Illumina slice
  chr20:10,000,000-10,100,000 (BAM + .bai + FASTA, bundled by tests/golden/make_golden.py
  na12878_100kb) -> make_examples' fused route -> CallVariantsOutput TFRecord on one GPU.  With
  --procs 1 one host process (per-stage milliseconds in the line); with --procs R the product's
  `--ranks_per_gpu R` driver: R processes, regions i % R == r, one GPU.  Prints its own JSON line;
  never the contract's `value`, whose inputs are resident in HBM."""
  import tempfile
  from deepvariant_amd import make_examples as me, tfrecord
  if args.procs > 1:
    import socket
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
      bam, fasta = _bam_fixture(tmp)
      with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
      mp.spawn(_bam_rank, args=(args.procs, port, tmp, bam, fasta, args.repeat), nprocs=args.procs, join=True)
    return
  stage = {}
  setup = {}            # the model's set-up runs on a worker thread, beside the region loop

  def timed(label, fn, store=None):
    store = stage if store is None else store

    def wrapper(*a, **k):
      t0 = time.perf_counter()
      try:
        return fn(*a, **k)
      finally:
        store[label] = store.get(label, 0.0) + time.perf_counter() - t0
    return wrapper

  class Hooks(me.RunnerHooks):
    def make_processor(self, options, ref_reader, po, device):
      proc = super().make_processor(options, ref_reader, po, device)
      proc.realign_reads = timed('realign (window selection on the device, assembly, alignment)', proc.realign_reads)
      proc.candidates_in_region = timed('allele counts (device) + candidate caller', proc.candidates_in_region)
      proc.generator.call_variants_in_region = timed(
          'pack + encode + classify (device) + CallVariantsOutput protos', proc.generator.call_variants_in_region)
      start_batch = proc.start_realign_tables

      def start_realign_tables(*a, **k):      # the native call of batch k + 1 runs while batch k is called and drawn
        finish = timed('realign: window selection (device counts) + marshalling', start_batch)(*a, **k)
        return timed('realign: wait for the assembly / alignment threads + write-back', finish)
      proc.start_realign_tables = start_realign_tables
      proc.generator.encode_region_on_device = timed('pack + encode (device)', proc.generator.encode_region_on_device)
      proc.flush_queue = timed('classify (device, several regions per forward) + CallVariantsOutput protos',
                               proc.flush_queue)
      if os.environ.get('DV_REGION_OBJECTS') is None:   # the table path: a batch's counts in one device call
        proc.process_tables = timed('allele counts (device, one call per batch of regions) + candidate caller',
                                    proc.process_tables)
      return proc

  with tempfile.TemporaryDirectory() as tmp:
    bam, fasta = _bam_fixture(tmp)
    me.RegionReads.__call__ = timed('BAM decode (native) + reads of the region', me.RegionReads.__call__)
    me.RegionReads.table = timed('BAM decode (native) + rows of the region', me.RegionReads.table)
    tfrecord.Writer.write = timed('TFRecord(GZIP) write', tfrecord.Writer.write)
    from deepvariant_amd import call_variants as cv_mod
    from deepvariant_amd.inception_v3 import InceptionV3
    InceptionV3.__init__ = timed('dv_model_create (activations for max_batch examples)', InceptionV3.__init__, setup)
    InceptionV3.load_flat_weights = timed('dv_model_load_weights (BN fold, fp16 pack, upload)', InceptionV3.load_flat_weights, setup)
    cv_mod.load_flat_checkpoint = timed('checkpoint read + load', cv_mod.load_flat_checkpoint, setup)
    weights = os.path.join(tmp, 'weights.f32')

    class WarmHooks(Hooks):
      def make_model(self, args, options):      # the timed run reads its weights from a file, as a real run does
        model = super().make_model(args, options)
        model.flat_weights.tofile(weights)
        # the checkpoint's calibration corrections are cached next to it by the first run that loads it; the timed
        # run finds them there, as every later run does
        model.calibrate_for_checkpoint(getattr(args, 'calibration_examples', 256), cache_prefix=weights)
        return model

    warm = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,003,000', 'warm.cvo.tfrecord.gz')
    me.make_examples_runner(warm, log=open(os.devnull, 'w'), hooks=WarmHooks())     # kernels loaded, graphs captured
    stage.clear()
    setup.clear()
    _repeat_calling_regions(me, args.repeat)
    timed_args = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,100,000', 'cvo.tfrecord.gz')
    timed_args.checkpoint = weights
    t0 = time.perf_counter()
    stats = me.make_examples_runner(timed_args, log=open(os.devnull, 'w'), hooks=Hooks())
    elapsed = time.perf_counter() - t0
    n_written = sum(1 for _ in tfrecord.read_tfrecords(os.path.join(tmp, 'cvo.tfrecord.gz')))
    if os.environ.get('DV_BAM_CPROFILE'):       # where the host time of the loop goes: one more run under cProfile
      import cProfile
      import io
      import pstats
      prof_args = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,100,000', 'prof.cvo.tfrecord.gz')
      prof_args.checkpoint = weights
      pr = cProfile.Profile()
      pr.enable()
      me.make_examples_runner(prof_args, log=open(os.devnull, 'w'), hooks=me.RunnerHooks())
      pr.disable()
      text = io.StringIO()
      pstats.Stats(pr, stream=text).sort_stats('tottime').print_stats(45)
      pstats.Stats(pr, stream=text).sort_stats('cumulative').print_stats(60)
      with open(os.environ['DV_BAM_CPROFILE'], 'w') as f:
        f.write(text.getvalue())
    # the same run once more with a HIP event pair around every kernel launch (eager launches):
    # how long the GPU computes for this slice, against the wall time of the timed run
    timed_stage, timed_setup = dict(stage), dict(setup)
    from deepvariant_amd import _lib
    lib = _lib.lib()
    lib.dv_set_profiling(1)
    again = _bam_args(me, tmp, bam, fasta, 'chr20:10,000,000-10,100,000', 'again.cvo.tfrecord.gz')
    again.checkpoint = weights
    me.make_examples_runner(again, log=open(os.devnull, 'w'), hooks=Hooks())
    torch.cuda.synchronize()
    kernel_ms = {'encoder': lib.dv_profile_ms(0), 'cnn': lib.dv_profile_ms(1), 'other (allele counts, pools, head)': lib.dv_profile_ms(2)}
    lib.dv_set_profiling(0)
    stage.clear()
    stage.update(timed_stage)
  assert n_written == stats['n_examples']
  print(json.dumps({
      'metric': 'examples/sec, BAM + FASTA -> CallVariantsOutput (make_examples fused route), one host process',
      'value': stats['n_examples'] / elapsed,
      'unit': 'examples/s',
      'n_gpus': 1,
      'data': _BAM_DATA,
      'wall_s': elapsed, 'region_loop_s': stats['loop_s'], 'setup_s': stats['setup_s'],
      'examples_per_s_region_loop_only': stats['n_examples'] / stats['loop_s'],
      'regions': stats['n_regions'], 'reads': stats['n_reads'], 'candidates': stats['n_candidates'],
      'examples': stats['n_examples'], 'table_path': stats.get('table_path'), 'passes_over_the_slice': args.repeat,
      'stage_ms': {k: 1e3 * v for k, v in sorted(stage.items(), key=lambda kv: -kv[1])},
      'unaccounted_ms': 1e3 * (elapsed - sum(stage.values())),
      'main_thread_wait_for_realigned_batches_ms': 1e3 * stats.get('wait_for_prepared_batches_s', 0.0),
      'model_setup_ms_on_worker_thread': {k: 1e3 * v for k, v in timed_setup.items()},
      'gpu_kernel_ms': kernel_ms, 'gpu_busy_frac': sum(kernel_ms.values()) / (1e3 * elapsed),
      'host_cores': os.cpu_count(), 'realigner_threads': os.environ.get('DV_REALIGN_THREADS', 'auto (<= 16)'),
      'note': 'not the contract metric: inputs start in files on the host, one Python process drives the region loop '
              '(the realigner of a batch of regions on native host threads, the model set up on a worker thread); '
              'gpu_busy_frac = kernel time of an instrumented repeat of the run / wall time of the timed run',
  }))


def make_longread_workload(kind, n, seed=None):
  """deepvariant_amd.synth.make_longread_workload (the product draws its calibration set with it too)."""
  from deepvariant_amd import synth
  return synth.make_longread_workload(kind, n, seed=seed)


def longread_bench(args, dev, local_rank, emit=True):
  """`--workload hifi35 | ont50`: the long-read shapes of SURVEY.md 8(d) / BASELINE configs[3], [4] through
  the same two entry points, one JSON line each (never the contract's `value`).

  A step = dv_encode_batch over every image of the batch (the reference-aligned pileup of each candidate
  plus, for every third candidate -- the indel share of the PacBio golden, 131 of 401 -- two alt-aligned
  images of the same reads in scratch space behind the examples) -> dv_merge_alt_channels (the two
  trailing diff channels, FillPileupArray's channel mode, deepvariant/pileup_image_native.h:246-271) ->
  dv_model_infer at the released model's input shape.  Reference context: docs/metrics.md:77-86."""
  import ctypes as CT
  from deepvariant_amd import _lib, packing, synth
  from deepvariant_amd.device_batch import DeviceBatch
  from deepvariant_amd.inception_v3 import InceptionV3
  from deepvariant_amd.pileup_image_native import _Encoder
  kind = 'hifi' if args.workload == 'hifi35' else 'ont'
  n = min(args.batch if args.batch != 7700 else 8192, 8192)
  opts, batch, with_alt, c_enc, Ct = make_longread_workload(kind, n)
  H, W = opts.height, opts.width
  img_bytes = H * W * Ct
  scratch0 = n * img_bytes
  entries = (_lib.DvAltMergeEntry * len(with_alt))()
  for k, i in enumerate(with_alt):
    entries[k].example, entries[k].first_row, entries[k].rows = i, 0, H
    entries[k].scratch_alt1, entries[k].scratch_alt2 = 2 * k, 2 * k + 1
  n_images = batch.n_items
  dbatch = DeviceBatch(batch, dev)
  enc = _Encoder(opts, W, device=local_rank)
  model = InceptionV3((H, W, Ct), max_batch=n, device=local_rank)
  model.init_random(seed=1234)
  calibration = calibrate_model(model, args)
  if args.dense_only:
    model.set_blank_skip(False)
  flat = torch.zeros(n_images * img_bytes, dtype=torch.uint8, device=dev)
  images = flat[:n * img_bytes].view(n, H, W, Ct)
  rows = torch.empty(n_images, dtype=torch.int32, device=dev)
  work = torch.cuda.Stream(device=dev)
  torch.cuda.set_stream(work)
  lib = _lib.lib()

  def step():
    dbatch.encode(enc, Ct, flat, rows)
    _lib.check(lib.dv_merge_alt_channels(flat.data_ptr(), scratch0, img_bytes, img_bytes, W, Ct, c_enc, 5,
                                         entries, len(with_alt), CT.c_void_p(work.cuda_stream)))
    return model(images)

  def sync_all():
    torch.cuda.synchronize(dev)

  elapsed, probs = timed_steps(step, sync_all, args.warmup, args.steps)
  lib.dv_set_profiling(1)
  for _ in range(args.steps):
    step()
  sync_all()
  enc_ms = lib.dv_profile_ms(0)
  enc_launches = lib.dv_last_profile_count()
  conv_ms = lib.dv_profile_ms(1)
  conv_launches = lib.dv_last_profile_count()
  other_ms = lib.dv_profile_ms(2)
  lib.dv_set_profiling(0)
  assert torch.isfinite(probs).all()
  thr = model.blank_thresholds(n)
  dense = None
  if thr is not None and not args.no_dense:      # the same K steps with blank-row skipping off (identical probabilities)
    model.set_blank_skip(False)
    elapsed_d, probs_d = timed_steps(step, sync_all, min(args.warmup, 2), args.steps)
    assert torch.equal(probs_d, probs), 'blank-row skipping changed a probability'
    lib.dv_set_profiling(1)
    for _ in range(args.steps):
      step()
    sync_all()
    lib.dv_profile_ms(0)
    conv_ms_d = lib.dv_profile_ms(1)
    lib.dv_set_profiling(0)
    model.set_blank_skip(True)
    dense = (elapsed_d, conv_ms_d)
  conv_flops = 2.0 * model.conv_macs_per_example
  executed = executed_conv_flops(model, (H, W, Ct), thr, conv_flops) if thr is not None else None
  exec_flops = executed['executed_flops_per_candidate'] if executed else conv_flops
  conv_tflops = exec_flops * n * args.steps / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
  # algorithmic bytes: every image's packed reads once per alignment drawn + the n example tensors
  t = batch.table
  seq_len = t.read_seq_off[1:].astype(np.int64) - t.read_seq_off[:-1]
  n_cig = t.read_cigar_off[1:].astype(np.int64) - t.read_cigar_off[:-1]
  per_read = 2 * seq_len + 8 * n_cig + 24 + 1
  in_bytes = float(per_read[np.asarray(batch.list_read, np.int64)].sum()) + n_images * W
  alg_bytes = in_bytes + float(n) * img_bytes
  enc_gbs = alg_bytes * args.steps / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
  out = {
      'metric': 'candidate pileups/sec (encode+CNN), %s' % args.workload,
      'value': n * args.steps / elapsed, 'unit': 'candidates/s', 'n_gpus': 1, 'steps': args.steps,
      'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8 (encoder) / f16 MFMA, f32 accumulate (CNN)',
      'data': 'synthetic',
      'config': {
          'workload': ('BASELINE configs[3] shape: synthetic PacBio HiFi 35x, PACBIO model input 100x147x10'
                       if kind == 'hifi' else
                       'BASELINE configs[4] shape: synthetic ONT R10.4 50x (10 % of the sites 96-160 reads deep, '
                       '~3 % indel events per base), ONT_R104 model input 100x199x9') +
                      '; channels = %d drawn + 2 alt-aligned diff channels; random-init weights' % c_enc,
          'candidates_per_step': n, 'images_drawn_per_step': n_images,
          'reads_per_step': int(t.n_reads), 'cigar_ops_per_read': float(n_cig.mean()),
          'reads_listed_per_image': float(len(batch.list_read)) / n_images,
      },
      'roofline': {
          'kernel': 'conv kernels (all 94 conv layers; inputs with more than 8 channels take preprocess_kernel + '
                    'the per-layer stem instead of the fused uint8 stem)',
          'bound': 'mfma', 'achieved': conv_tflops, 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
          'frac': conv_tflops / MFMA_F16_PEAK_TFLOPS, 'traffic': None, 'flops_per_candidate': conv_flops,
          'executed_flops_per_candidate': exec_flops,
          'avg_launch_ms': conv_ms / max(conv_launches, 1), 'launches': conv_launches,
          'ms_per_step': conv_ms / args.steps,
      },
      'roofline_encoder': {
          'kernel': 'encode_items_kernel', 'bound': 'hbm', 'achieved': enc_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
          'frac': enc_gbs / HBM_PEAK_GBS, 'traffic': None, 'bytes_per_candidate': alg_bytes / n,
          'avg_launch_ms': enc_ms / max(enc_launches, 1), 'launches': enc_launches,
          'images_per_s': n_images * args.steps / (enc_ms * 1e-3) if enc_ms > 0 else 0.0,
      },
      'other_kernels_ms_per_step': other_ms / args.steps,
  }
  out['calibration'] = calibration
  out['precise'] = bool(model.precise)
  out['config']['classifier_mode'] = (
      'precise (the default for inputs of more than 8 channels): every fp16 tensor of the 17x17 and 8x8 stages as hi + lo '
      'pieces, their consumers at twice the K -- what holds 1e-3 on every held-out weight seed; `fast` below = DV_PRECISE=0'
      if model.precise else 'fast (DV_PRECISE=0): fp16 activations everywhere')
  if dense is not None:
    elapsed_d, conv_ms_d = dense
    dense_tflops = conv_flops * n * args.steps / (conv_ms_d * 1e-3) / 1e12 if conv_ms_d > 0 else 0.0
    out['value_dense'] = n * args.steps / elapsed_d
    out['ms_per_step_dense'] = 1e3 * elapsed_d / args.steps
    out['roofline']['dense'] = {'achieved': dense_tflops, 'frac': dense_tflops / MFMA_F16_PEAK_TFLOPS,
                                'ms_per_step': conv_ms_d / args.steps, 'flops_per_candidate': conv_flops}
    out['blank_row_skipping'] = executed
  if not args.no_cpu_baseline:
    out['parity'] = longread_parity(opts, batch, n, with_alt, Ct, c_enc, model, images, probs)
    out['parity'].update(cnn_tail_parity(model, Ct, images, probs))
  if model.precise and not args.no_dense:
    # the same steps in fast mode (DV_PRECISE=0: fp16 activations everywhere) on a second model of the same weights
    flat_weights = model.flat_weights
    del model
    torch.cuda.empty_cache()
    old = os.environ.get('DV_PRECISE')
    os.environ['DV_PRECISE'] = '0'
    try:
      model = InceptionV3((H, W, Ct), max_batch=n, device=local_rank)
    finally:
      if old is None:
        os.environ.pop('DV_PRECISE', None)
      else:
        os.environ['DV_PRECISE'] = old
    model.load_flat_weights(flat_weights)
    calibrate_model(model, args)
    elapsed_f, probs_f = timed_steps(step, sync_all, min(args.warmup, 2), args.steps)
    lib.dv_set_profiling(1)
    for _ in range(args.steps):
      step()
    sync_all()
    lib.dv_profile_ms(0)
    conv_ms_f = lib.dv_profile_ms(1)
    lib.dv_set_profiling(0)
    fast = {'value': n * args.steps / elapsed_f, 'ms_per_step': 1e3 * elapsed_f / args.steps,
            'conv_ms_per_step': conv_ms_f / args.steps}
    if not args.no_cpu_baseline:
      fast['parity'] = cnn_tail_parity(model, Ct, images, probs_f)
    out['fast'] = fast
  if emit:
    print(json.dumps(out))
  return out


def longread_parity(opts, batch, n, with_alt, Ct, c_enc, model, images, probs, sample=192):
  """The first `sample` examples of the timed batch against the oracle: tensors (the oracle draws the
  reference-aligned and the two alt-aligned images, the diff channels are merged here in numpy as
  FillPileupArray does) bit-exact, softmax within 1e-3 of the fp32 restatement."""
  from deepvariant_amd import packing
  from oracle import inception_ref, oracle as O
  H, W = opts.height, opts.width
  sample = min(sample, n)
  img_bytes = H * W * Ct
  alt_k = {i: k for k, i in enumerate(with_alt)}
  off = np.asarray(batch.item_list_off)
  lr, lc = np.asarray(batch.list_read), np.asarray(batch.list_code)
  sub = packing.PackedBatch(table=batch.table, width=W)
  sub.ref_windows_list = batch.ref_windows_list
  slots = []
  for i in range(sample):
    for item in [i] + ([n + 2 * alt_k[i], n + 2 * alt_k[i] + 1] if i in alt_k else []):
      a, b = off[item], off[item + 1]
      sub.add_item(batch.item_variant_start[item], batch.item_image_start[item], batch.item_ref_idx[item],
                   lr[a:b], lc[a:b], height=H, out_off=len(slots) * img_bytes)
      slots.append((i, item))
  raw, _ = O.encode_packed(opts, sub, Ct, n_threads=min(64, os.cpu_count() or 1))
  drawn = raw.reshape(len(slots), H, W, Ct)
  want = np.zeros((sample, H, W, Ct), np.uint8)
  k = 0
  for i in range(sample):
    want[i] = drawn[k]
    if i in alt_k:
      want[i, :, :, c_enc] = drawn[k + 1][:, :, 5]
      want[i, :, :, c_enc + 1] = drawn[k + 2][:, :, 5]
      k += 2
    k += 1
  got = images[:sample].cpu().numpy()
  ref = inception_ref.InceptionV3(Ct)
  ref.load_flat(model.flat_weights)
  torch.set_num_threads(min(64, os.cpu_count() or 1))
  with torch.no_grad():
    wp = torch.cat([ref(torch.from_numpy(got[i:i + 64]), channels_last=True) for i in range(0, sample, 64)])
  dp = (probs[:sample].cpu() - wp).abs().max(1).values
  return {'candidates': sample, 'pileup_tensors_bit_exact': bool((got == want).all()),
          'max_abs_dp': float(dp.max()), 'mean_abs_dp': float(dp.mean()), 'tolerance': 1e-3,
          'ok': bool(dp.max() <= 1e-3),
          'oracle_prob_spread': float((wp.max(0).values - wp.min(0).values).max())}


def parity_sample(region, opts, C, model, images, probs, n=512):
  """`n` SITES strided across the whole TIMED batch against the oracle, after the timed region.

  The oracle is driven through its PROTO-SHAPED interface (oracle.build_pileup: DeepVariantCall
  with allele_support read-NAME lists, Read objects with names, the reference window) -- it does
  the reference's string matching, read query order and name sorting itself, so nothing the
  product packed (support codes, name ranks, read lists) can cancel out.  Pileup tensors must be
  bit-exact; the softmax is checked against the fp32 torch restatement loaded with the same
  weights (bar: 1e-3, BASELINE.json)."""
  from oracle import inception_ref, proto_driver
  n_sites = len(region[1])
  picks = sorted(set(int(i) for i in np.linspace(0, n_sites - 1, num=min(n, n_sites))))
  H, W = opts.height, opts.width
  items, want_imgs = proto_driver.pileups_of_sites(opts, region, picks)
  want_img = np.stack(want_imgs).reshape(len(items), -1)
  idx = torch.tensor(items, dtype=torch.long, device=images.device)
  got_img = images.index_select(0, idx).cpu().numpy().reshape(len(items), -1)
  ref = inception_ref.InceptionV3(C)
  ref.load_flat(model.flat_weights)
  torch.set_num_threads(min(64, os.cpu_count() or 1))
  with torch.no_grad():
    x = torch.from_numpy(got_img.reshape(len(items), H, W, C))
    want = torch.cat([ref(x[i:i + 64], channels_last=True) for i in range(0, len(items), 64)])
  dp = (probs.index_select(0, idx).cpu() - want).abs().max(1).values
  err = float(dp.max())
  return {
      'candidates': len(items),
      'sites': len(picks),
      'sampling': 'sites strided over the whole timed batch; oracle driven with proto-shaped inputs '
                  '(DeepVariantCall + named reads), not with the packed batch',
      'pileup_tensors_bit_exact': bool((got_img == want_img).all()),
      'max_abs_dp': err,
      'mean_abs_dp': float(dp.mean()),
      'tolerance': 1e-3,
      'ok': bool(err <= 1e-3),
      # the check means something only if the oracle's answers differ between candidates
      'oracle_prob_spread': float((want.max(0).values - want.min(0).values).max()),
  }


def executed_conv_flops(model, shape, thr, nominal_flops_per_item):
  """Nominal conv FLOPs minus the output tiles the stem kernels copied (blank-row skipping), mirrored from their
  tile rules (csrc/stem.hip next_tile: a 7 x 54 conv2 tile / a 6 x 9 pooled tile starting at or below the example's
  threshold; model.hip conv_pool_resident_kernel: a 30-position fragment walks down to the deepest threshold among
  its lanes; conv_mfma_kernel: a wave tile inside one example from its threshold row on) and the per-example
  thresholds of the last forward (dv_model_blank_thresholds).  -> dict for the JSON line."""
  h, w, c = shape
  n = thr.shape[1]
  oh1, ow1 = (h - 3) // 2 + 1, (w - 3) // 2 + 1
  oh2, ow2 = oh1 - 2, ow1 - 2
  ph, pw = (oh2 - 3) // 2 + 1, (ow2 - 3) // 2 + 1
  oh4, ow4 = ph - 2, pw - 2
  p4 = (oh4 - 3) // 2 + 1
  t2, t4, t5 = thr[1].astype(np.int64), thr[2].astype(np.int64), thr[4].astype(np.int64)
  mac1, mac2, mac3 = 9 * c * 32, 9 * 32 * 32, 9 * 32 * 64
  mac1x1, mac4 = 64 * 80, 9 * 80 * 192
  skipped = {}
  if c <= 12 and os.environ.get('DV_NO_STEM_A_WIDE') is None or c <= 8:      # fused stem_a: conv2 tiles of 7 rows; conv1 rows that only skipped tiles read
    rows2 = np.minimum(oh2, 7 * ((np.minimum(t2, oh2) + 6) // 7))
    rows1 = np.where(rows2 > 0, np.minimum(oh1, rows2 + 2), 0)
    skipped['stem_a'] = float(((oh2 - rows2) * ow2 * mac2 + (oh1 - rows1) * ow1 * mac1).sum())
  else:           # per-layer conv2 (conv_mfma_kernel<1,4>: 128-pixel wave tiles of the flattened (example, row, column) index)
    px = 128
    first = np.arange(0, n * oh2 * ow2, px, dtype=np.int64)
    last = np.minimum(first + px, n * oh2 * ow2) - 1
    ex_f, ex_l = first // (oh2 * ow2), last // (oh2 * ow2)
    row_f = (first % (oh2 * ow2)) // ow2
    blank = (ex_f == ex_l) & (row_f >= t2[ex_f])
    skipped['conv2'] = float(((last - first + 1) * blank).sum() * mac2)
  rows_p = np.minimum(ph, 6 * ((np.minimum(t4, ph) + 5) // 6))
  rows3 = np.where(rows_p > 0, np.minimum(oh2, 2 * rows_p + 1), 0)
  skipped['stem_b'] = float(((oh2 - rows3) * ow2 * mac3 + (ph - rows_p) * pw * mac1x1).sum())
  pos = n * ow4
  frag0 = np.arange(0, pos, 30, dtype=np.int64)
  lanes = np.minimum(frag0[:, None] + np.arange(32)[None, :], pos - 1)
  s_end = np.minimum(t5[lanes // ow4], p4).max(axis=1)
  rows4 = np.where(s_end > 0, np.minimum(oh4, 2 * s_end + 1), 0)
  new_pos = np.minimum(30, pos - frag0)
  skipped['conv3x3_80_192'] = float(((oh4 - rows4) * new_pos).sum() * mac4)
  total_skipped = 2.0 * sum(skipped.values()) / n
  return {
      'executed_flops_per_candidate': nominal_flops_per_item - total_skipped,
      'skipped_flops_per_candidate': total_skipped,
      'skipped_share_of_nominal': total_skipped / nominal_flops_per_item,
      'skipped_gflop_per_candidate_by_kernel': {k: 2.0 * v / n / 1e9 for k, v in skipped.items()},
      'mean_rows_used': float(thr[0].mean()),
  }


def calibrate_model(model, args):
  """What the product does when it loads a checkpoint (InceptionV3.calibrate_for_checkpoint): dv_model_calibrate on
  the fixed synthetic calibration set of the model's input shape -- other pileups than the timed batch (another
  seed), the same on every rank; model preparation before the timed region.  --calibration-images 0 = the
  uncalibrated fp16 model."""
  from deepvariant_amd import calibration_set
  n = args.calibration_images
  if n <= 0:
    return {'images': 0}
  t0 = time.perf_counter()
  corr = model.calibrate_for_checkpoint(n)
  return {'images': n, 'set': 'deepvariant_amd/calibration_set.py v%d, seed %d' % (calibration_set.SET_VERSION,
                                                                                 calibration_set.SET_SEED),
          'seconds': time.perf_counter() - t0, 'max_abs_shift_correction': float(np.abs(corr).max()),
          'what': 'per-channel mean of the fp16 pipeline error moved into the fp32 shifts (dv_model_calibrate), '
                  'measured on a fixed synthetic set per input shape: a property of the checkpoint, not of the run'}


def cnn_tail_parity(model, C, images, probs, check=64):
  """|dp| over EVERY candidate of the timed batch: the fp32 oracle run on the GPU through torch-ROCm (its im2col +
  matmul form, oracle/inception_gpu.py), after that run has been compared with the CPU oracle on `check` images."""
  from oracle import inception_gpu as G, inception_ref
  ref = inception_ref.InceptionV3(C)
  ref.load_flat(model.flat_weights)
  ref_gpu = inception_ref.InceptionV3(C)
  ref_gpu.load_flat(model.flat_weights)
  ref_gpu = ref_gpu.to(images.device)
  agree = G.check_gpu_oracle(ref, ref_gpu, images, n=check)
  st = G.tail_stats(probs.cpu().numpy(), G.oracle_probs_gpu(ref_gpu, images))
  return {'all_candidates': st['n'], 'max_abs_dp_all': st['max_abs_dp'], 'mean_abs_dp_all': st['mean_abs_dp'],
          'p999_abs_dp': st['p999_abs_dp'], 'p9999_abs_dp': st['p9999_abs_dp'], 'n_over_tol': st['n_over_tol'],
          'gpu_oracle_vs_cpu_oracle_max_abs_dp': agree, 'gpu_oracle_checked_on': check,
          'ok_all': bool(st['max_abs_dp'] <= 1e-3)}


CONV_KERNELS = ('conv_mfma_kernel', 'conv_resident_kernel', 'conv_pool1x1_kernel', 'conv_first_u8_kernel',
                'stem_a_kernel', 'stem_b_kernel', 'imgconv_kernel', 'chain_kernel')


def _pmc_pass(counter, args, timeout_s=300):
  """One `rocprofv3 --kernel-trace --pmc <counter>` pass over a short run of THIS bench (child
  process, 3 forward passes) -> {kernel name: (launches, bytes)}, forward passes.  Counters are in
  KB (MI355X_MICROARCH.md, HBM section)."""
  import glob
  import shutil
  import sqlite3
  import subprocess
  import tempfile
  out = tempfile.mkdtemp(prefix='dvbench_pmc_', dir='/tmp')
  try:
    env = dict(os.environ, TMPDIR='/tmp', DV_BENCH_NO_PMC='1')
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-d', out, '--', sys.executable,
           os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-workloads', '--no-dense',
           '--calibration-images', '0', '--batch', str(args.batch), '--channels', str(args.channels)]
    subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout_s, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dbs = glob.glob(os.path.join(out, '**', '*.db'), recursive=True)
    cur = sqlite3.connect(dbs[0]).cursor()
    suf = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like "
                                     "'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
    rows = cur.execute(
        f'select k.kernel_name, p.value, d.id from rocpd_pmc_event{suf} p '
        f'join rocpd_info_pmc{suf} i on p.pmc_id=i.id '
        f'join rocpd_kernel_dispatch{suf} d on p.event_id=d.event_id '
        f'join rocpd_info_kernel_symbol{suf} k on d.kernel_id=k.id where i.name=?', (counter,))
    agg = {}
    for name, value, did in rows:
      ids, total = agg.setdefault(name, (set(), [0.0]))
      ids.add(did)
      total[0] += value * 1024.0
    return {k: (len(ids), total[0]) for k, (ids, total) in agg.items()}
  finally:
    shutil.rmtree(out, ignore_errors=True)


def pmc_traffic(args, n_items):
  """HBM bytes per launch of the conv kernels and of the encoder: FETCH_SIZE (doubled: gfx950
  tallies a 128-byte request as 64 bytes) + WRITE_SIZE, measured IN THIS RUN by two separate
  rocprofv3 --pmc passes over a short child run of this bench (same kernels, same batch), as the
  guide's HBM section prescribes.  Falls back to the committed passes of round 2 (stale for kernels
  changed since) when rocprofv3 is not on PATH or a pass fails; DV_BENCH_NO_PMC skips both."""
  import shutil
  if os.environ.get('DV_BENCH_NO_PMC') is None and shutil.which('rocprofv3') is not None:
    try:
      fetch = _pmc_pass('FETCH_SIZE', args)
      write = _pmc_pass('WRITE_SIZE', args)
      enc = [k for k in fetch if 'encode_items_kernel' in k][0]
      passes = fetch[enc][0]
      conv = [k for k in fetch if any(c in k for c in CONV_KERNELS)]
      launches = sum(fetch[k][0] for k in conv)
      conv_bytes = (2.0 * sum(fetch[k][1] for k in conv) + sum(write[k][1] for k in conv if k in write)) / launches
      enc_bytes = (2.0 * fetch[enc][1] + write[enc][1]) / passes
      if os.environ.get('DV_BENCH_PMC_SAVE'):   # per-kernel table of the two passes, for profiles/
        with open(os.environ['DV_BENCH_PMC_SAVE'], 'w') as f:
          f.write('# bench.py same-run counter passes (%d forward passes of %d candidates): launches, FETCH_SIZE x2 '
                  '(gfx950 correction) and WRITE_SIZE in MB per forward pass\n' % (passes, n_items))
          for k in sorted(fetch, key=lambda k: -fetch[k][1]):
            f.write('%-90s %5d %12.1f %12.1f\n' % (k[:90], fetch[k][0] // passes, 2.0 * fetch[k][1] / passes / 1e6,
                                                  write.get(k, (0, 0.0))[1] / passes / 1e6))
      return conv_bytes, enc_bytes, (
          'bytes per launch measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE (x2, gfx950) and '
          '--pmc WRITE_SIZE, separate passes over a child run of %d forward passes, %d conv launches each' % (
              passes, launches // max(passes, 1)))
    except Exception as e:     # pylint: disable=broad-except
      sys.stderr.write('bench: counter passes failed (%s: %s); using the committed ones\n' % (type(e).__name__, e))
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                      'r02_pmc_traffic.json')
  try:
    with open(path) as f:
      t = json.load(f)
  except (OSError, ValueError):
    return None, None, None
  if t.get('candidates_per_step') != n_items:
    return None, None, 'PMC passes were taken at %s candidates/step' % t.get('candidates_per_step')
  c, e = t['conv'], t['encoder']
  conv = (c['fetch_bytes_per_pass_x2'] + c['write_bytes_per_pass']) / c['launches_per_pass']
  enc = e['fetch_bytes_per_launch_x2'] + e['write_bytes_per_launch']
  return conv, enc, 'STALE (round-2 kernels) bytes per launch, separate rocprofv3 --pmc passes: ' + t['source']


def usable_cores():
  """Cores this process may actually use: the scheduler affinity mask, cut by the cgroup CPU quota
  (a container on a 256-core host is often given far fewer)."""
  n = os.cpu_count() or 1
  try:
    n = min(n, len(os.sched_getaffinity(0)))
  except (AttributeError, OSError):
    pass
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      with open(path) as f:
        fields = f.read().split()
      if path.endswith('cpu.max'):
        if fields[0] != 'max':
          n = min(n, max(1, int(float(fields[0]) / float(fields[1]) + 0.5)))
      else:
        quota = int(fields[0])
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
          period = int(f.read().split()[0])
        if quota > 0:
          n = min(n, max(1, int(quota / period + 0.5)))
    except (OSError, ValueError, IndexError):
      continue
  return n


def cpu_baseline(host_batch, opts, C, sample):
  """The oracle (a port: C++ encoder restatement + fp32 torch Inception)
  timed on this host's cores on a bounded sample of the same workload."""
  from oracle import inception_ref, oracle as O
  cores = usable_cores()
  n_enc = sample or min(host_batch.n_items, 2048)
  # sub-batch = the first n_enc items (lists index the shared read table)
  t0 = time.perf_counter()
  sub = _first_items(host_batch, n_enc)
  imgs, _ = O.encode_packed(opts, sub, C, n_threads=cores)
  t_enc = t_enc_port = time.perf_counter() - t0
  # The encoder leg on the REFERENCE's own code where its build travelled with the repo (oracle/_ref/libdvref.so:
  # deepvariant/pileup_image_native.cc, pileup_channel_lib.cc, channels/*.cc compiled unmodified, oracle/ref_build/):
  # same packed sample re-expanded into Read / DeepVariantCall objects, BuildPileupForOneSample per candidate, on
  # the same threads.  Its pixels must equal the oracle's; its time is the encoder time of the baseline.
  encoder_kind, ref_note = 'port', 'reference encoder build not present (oracle/_ref/libdvref.so)'
  try:
    if O.reference_available():
      with O.reference_backend():
        t0 = time.perf_counter()
        ref_imgs, _ = O.encode_packed(opts, sub, C, n_threads=cores)
        t_ref = time.perf_counter() - t0
      if not np.array_equal(ref_imgs, imgs):
        raise RuntimeError('the reference build and the oracle restatement drew different pileups')
      t_enc, encoder_kind = t_ref, 'reference'
      ref_note = 'pixels equal the oracle restatement\'s on all %d candidates' % n_enc
  except Exception as e:      # pylint: disable=broad-except   (the baseline is context: never fail the bench on it)
    ref_note = 'reference encoder leg failed: %s' % e
  n_cnn = min(n_enc, 256)
  ref = inception_ref.make_random_model(C, seed=1)
  x = torch.from_numpy(imgs.reshape(-1, opts.height, opts.width, C)[:n_cnn])
  # torch's CPU convs stop scaling long before a 256-core host is full: time a few thread
  # counts on a slice and keep the fastest; then best of two passes over the whole sample.
  best_threads, best_cl, best_rate = 1, False, 0.0
  with torch.no_grad():
    for th in sorted({min(cores, t) for t in (16, 32, 64, 128)}):
      torch.set_num_threads(th)
      for cl in (False, True):
        ref(x[:8], channels_last=cl)
        t0 = time.perf_counter()
        ref(x[:32], channels_last=cl)
        rate = 32 / (time.perf_counter() - t0)
        if rate > best_rate:
          best_threads, best_cl, best_rate = th, cl, rate
    torch.set_num_threads(best_threads)
    t_cnn = float('inf')
    for _ in range(2):
      t0 = time.perf_counter()
      ref(x, channels_last=best_cl)
      t_cnn = min(t_cnn, time.perf_counter() - t0)
  # One torch process leaves most of a 256-core host idle (16 threads won the sweep in round 3); the
  # reference fills a node with N independent shard processes (scripts/run_deepvariant.py:457-462).
  # Same here: P = cores / best_threads classifier processes run concurrently, each on its own copy of
  # the sample, and the node rate is what they finish together.
  procs = max(1, cores // best_threads)
  node_rate, used = _cnn_node_rate(C, imgs.reshape(-1, opts.height, opts.width, C)[:n_cnn], best_threads,
                                   best_cl, procs)
  one_rate = n_cnn / t_cnn
  cnn_rate = max(node_rate, one_rate)
  per_item = t_enc / n_enc + 1.0 / cnn_rate
  return {
      'value': 1.0 / per_item,
      'unit': 'candidates/s',
      'cores': cores,
      'host_logical_cpus': os.cpu_count(),
      'cores_used': used * best_threads if node_rate >= one_rate else best_threads,
      'kind': 'port',
      'sample': '%d candidates encoded by %s on %d threads (%.2f s; %s) '
                '+ fp32 torch-CPU Inception-v3 (a port: tf_keras cannot run here): %d processes x %d threads, each '
                'classifying the same %d candidates in one batch, concurrently (node rate %.0f/s; one process alone '
                '%.0f/s)' %
                (n_enc, "the reference's own encoder sources (oracle/_ref)" if encoder_kind == 'reference' else
                 'the C++ oracle restatement', cores, t_enc, ref_note, used, best_threads, n_cnn, node_rate, one_rate),
      'encoder_kind': encoder_kind,
      'encoder_candidates_per_s_port': n_enc / t_enc_port,
      'cnn_threads': best_threads,
      'cnn_processes': used,
      'cnn_channels_last': best_cl,
      'encoder_candidates_per_s': n_enc / t_enc,
      'cnn_candidates_per_s': cnn_rate,
      'cnn_candidates_per_s_one_process': one_rate,
  }


def _cnn_worker_main(spec):
  """`bench.py --cnn-worker C,H,W,N,threads,channels_last`: one classifier process of the cpu_baseline
  leg.  Prints READY after a warm-up, waits for a line on stdin, classifies N synthetic images in one
  batch and prints the wall-clock span."""
  C, H, W, n, threads, cl = (int(v) for v in spec.split(','))
  from oracle import inception_ref
  torch.set_num_threads(threads)
  ref = inception_ref.make_random_model(C, seed=1)
  x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (n, H, W, C), dtype=np.uint8))
  with torch.no_grad():
    ref(x[:8], channels_last=bool(cl))
    print('READY', flush=True)
    sys.stdin.readline()
    t0 = time.time()
    ref(x, channels_last=bool(cl))
    print('SPAN %.6f %.6f' % (t0, time.time()), flush=True)


def _cnn_node_rate(C, x, threads, channels_last, procs, timeout_s=240):
  """`procs` classifier processes (own interpreters; `threads` torch threads each) classify a batch of
  x's shape at the same time -> (candidates/s of the node, processes that finished).  Bounded by one
  overall deadline; 0 when nothing finished."""
  import select
  import subprocess
  n, H, W, _ = x.shape
  spec = '%d,%d,%d,%d,%d,%d' % (C, H, W, n, threads, int(channels_last))
  ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cnn-worker', spec],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        for _ in range(procs)]
  deadline = time.time() + timeout_s

  def lines(prefix):
    got = {}
    while len(got) < len(ps) and time.time() < deadline:
      waiting = [p.stdout for i, p in enumerate(ps) if i not in got and p.poll() is None]
      if not waiting:
        break
      r, _, _ = select.select(waiting, [], [], max(0.0, min(5.0, deadline - time.time())))
      for f in r:
        i = [q.stdout for q in ps].index(f)
        line = f.readline()
        if line.startswith(prefix):
          got[i] = line
    return got
  spans = []
  try:
    ready = lines('READY')
    for i in ready:
      ps[i].stdin.write('go\n')
      ps[i].stdin.flush()
    ps_all, ps[:] = ps[:], [ps[i] for i in ready]
    for line in lines('SPAN').values():
      _, b, e = line.split()
      spans.append((float(b), float(e)))
    ps[:] = ps_all
  finally:
    for p in ps:
      if p.poll() is None:
        p.kill()
      p.wait()
  if not spans:
    return 0.0, 0
  wall = max(e for _, e in spans) - min(b for b, _ in spans)
  return len(spans) * n / wall, len(spans)


def _first_items(batch, n):
  from deepvariant_amd import packing
  sub = packing.PackedBatch(table=batch.table, width=batch.width)
  sub.ref_windows_list = batch.ref_windows_list
  off = batch.item_list_off
  lr, lc = np.asarray(batch.list_read), np.asarray(batch.list_code)
  for i in range(n):
    a, b = off[i], off[i + 1]
    sub.add_item(batch.item_variant_start[i], batch.item_image_start[i],
                 batch.item_ref_idx[i], lr[a:b], lc[a:b],
                 height=batch.item_height[i], out_off=batch.item_out_off[i])
  return sub


if __name__ == '__main__':
  main()
