"""Mutation fuzz of the host-only alignment-file decoders (csrc/bam_reader.cpp, csrc/cram_reader.cpp, csrc/aux_planes.h)
under AddressSanitizer + UndefinedBehaviorSanitizer.  TEST / DEVELOPMENT TOOL, not part of the product.

  python tools/decoder_fuzz/fuzz.py [work_dir]

Builds tools/decoder_fuzz/harness.cc + the two decoders with clang -fsanitize=address,undefined (host only, no HIP
runtime needed), writes seed files with the test suite's own writers (a BAM whose records carry MM / ML / MN / tp / t0
tags of every shape; CRAMs with embedded references, raw and compressed blocks, tag series), mutates them (bytes, 32-bit
words set to boundary values, deletions -- BGZF members are re-compressed, so the damage reaches the record parser), and
reads every mutant with all plane parsing on.  A sanitizer report or a crash fails the run; decoding errors are the
expected outcome.  Last result: profiles/r04_decoder_fuzz.txt.
"""
import os, sys, struct, subprocess, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from deepvariant_amd import genomics_io as gio
from tests import test_aux_planes_cpu as TA
from tests import cram_writer as W
from tests.test_cram_native_cpu import EXTERNAL_CODECS, BIT_CODECS, _rans_encode, _reads_for, _contigs

work = sys.argv[1] if len(sys.argv) > 1 else '/tmp/dv_decoder_fuzz'
out = work + '/files'
os.makedirs(out, exist_ok=True)
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
FLAGS = ['-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-D__HIP_PLATFORM_AMD__',
         '-I/opt/rocm/include', '-I' + ROOT + '/include', '-I' + ROOT + '/deepvariant_amd/csrc', '-x', 'c++']
objs = []
for src in (ROOT + '/tools/decoder_fuzz/harness.cc', ROOT + '/deepvariant_amd/csrc/bam_reader.cpp',
            ROOT + '/deepvariant_amd/csrc/cram_reader.cpp'):
    obj = work + '/' + os.path.basename(src) + '.o'
    subprocess.check_call([CLANG] + FLAGS + ['-c', src, '-o', obj])
    objs.append(obj)
subprocess.check_call([CLANG, '-fsanitize=address,undefined'] + objs + ['-o', work + '/harness', '-lz', '-ldl', '-lpthread'])
rng = np.random.default_rng(5)
# --- seeds: a BAM with aux tags of every kind, CRAMs with embedded references (raw and compressed blocks)
reads = [TA._random_mod_read(rng, k) for k in range(60)] + TA._case_reads()
for k, r in enumerate(reads):
    r.alignment.position.position = 5 + 70 * k
    if k % 3 == 0:
        n = len(r.aligned_sequence)
        r.info['tp'] = gio.T.ListValue(values=[gio.T.Value(int_value=int(v)) for v in rng.integers(-2, 3, size=n)])
        r.info['t0'] = gio.T.ListValue(values=[gio.T.Value(string_value='5' * n)])
gio.write_bam(out + '/seed.bam', [('chr1', 20000)], reads)
pr = random.Random(3)
contigs = _contigs(pr)
for name, codecs, methods in (('raw', EXTERNAL_CODECS, {}), ('bits', BIT_CODECS, {0: 'gzip', 11: 'rans1', 12: 'bzip2', 14: 'lzma', 15: 'rans0'})):
    w = W.CramWriter(contigs, codecs, methods, _rans_encode)
    g = _reads_for(pr, contigs, 0, 40, 1, 900, 'a')
    for r in g:
        r.pop('tags', None); r.pop('hp', None)
        r['tags'] = [(b'MMZ', b'C+m?,0,1;\0'), (b'MLB', b'C' + struct.pack('<I', 2) + b'\x05\xf0'), (b'tpB', b'c' + struct.pack('<I', 3) + b'\x01\xff\x02'), (b't0Z', b'555\0')]
    w.add_container([(g, 0)], embed=True)
    w.finish(out + '/seed_%s.cram' % name)

def bgzf_payload(path):
    return b''.join(gio._bgzf_blocks(path))

def write_bgzf(path, payload):
    with open(path, 'wb') as f:
        for off in range(0, len(payload), 60000):
            f.write(gio._bgzf_block(payload[off:off + 60000]))
        f.write(gio._bgzf_block(b''))

payload = bytearray(bgzf_payload(out + '/seed.bam'))
hdr = 8 + struct.unpack_from('<i', payload, 4)[0]
files = []
for i in range(300):
    p = bytearray(payload)
    for _ in range(int(rng.integers(1, 6))):
        at = int(rng.integers(hdr, len(p)))
        kind = rng.random()
        if kind < 0.6:
            p[at] = int(rng.integers(0, 256))
        elif kind < 0.8:
            p[at:at + 4] = struct.pack('<I', int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000, 65536])))
        else:
            del p[at:at + int(rng.integers(1, 9))]
    path = out + '/m%03d.bam' % i
    write_bgzf(path, bytes(p))
    files.append(path)
for name in ('raw', 'bits'):
    data = bytearray(open(out + '/seed_%s.cram' % name, 'rb').read())
    for i in range(300):
        p = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(26, len(p) - 40))
            if rng.random() < 0.7:
                p[at] = int(rng.integers(0, 256))
            else:
                p[at:at + 4] = struct.pack('<I', int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000])))
        path = out + '/m_%s%03d.cram' % (name, i)
        open(path, 'wb').write(bytes(p))
        files.append(path)
files = [out + '/seed.bam', out + '/seed_raw.cram', out + '/seed_bits.cram'] + files
crashes = 0
for at in range(0, len(files), 50):
    chunk = files[at:at + 50]
    r = subprocess.run([work + '/harness'] + chunk, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    text = r.stdout + r.stderr
    if 'AddressSanitizer' in text or 'runtime error' in text or r.returncode != 0:
        crashes += 1
        done = text.count(' rc=')
        print('== chunk at', at, 'rc', r.returncode, 'files done', done, 'culprit', chunk[min(done, len(chunk) - 1)])
        print('\n'.join([l for l in text.split('\n') if 'ERROR' in l or 'runtime error' in l or l.strip().startswith('#')][:14]))
outcomes = {}
for at in range(0, len(files), 100):
    r = subprocess.run([work + '/harness'] + files[at:at + 100], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))
    for line in r.stdout.split('\n'):
        if ' rc=' in line:
            msg = line.split(' rc=', 1)[1]
            key = 'decoded' if msg.startswith('0') else ' '.join(w for w in msg.split(' ', 1)[1].split(' ') if '/' not in w)[:60]
            outcomes[key] = outcomes.get(key, 0) + 1
print('%d files (3 seeds + %d mutants); sanitizer reports / crashes: %d' % (len(files), len(files) - 3, crashes))
for k, v in sorted(outcomes.items(), key=lambda kv: -kv[1]):
    print('  %5d  %s' % (v, k))
sys.exit(1 if crashes else 0)
