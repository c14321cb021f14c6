mkdir -p gpurun_out/r2w
DV_NB6=1 timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -x > gpurun_out/r2w/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2w/pytest.log
tail -5 gpurun_out/r2w/pytest.log
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2w/trace_base.txt
DV_NB6=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2w/trace_nb6.txt
python - <<'PY'
import re
def load(f):
  lines=[l.rstrip() for l in open(f) if l.startswith('[dv-op]')]
  idx=[i for i,l in enumerate(lines) if 'total' in l]
  out=[]
  for l in lines[idx[-2]+1:idx[-1]]:
    m=re.search(r'([\d.]+) us',l); out.append((l[8:l.find(m.group(0))].strip(), float(m.group(1))))
  return out
a=load('gpurun_out/r2w/trace_base.txt'); b=load('gpurun_out/r2w/trace_nb6.txt')
print(len(a),len(b))
for (n1,t1),(n2,t2) in zip(a,b):
  if n1!=n2: print('%-60s %8.1f -> %-40s %8.1f'%(n1[:60],t1,n2[-40:],t2))
print('total',sum(t for _,t in a),sum(t for _,t in b))
PY
for v in 0 1 0 1; do DV_NB6=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH nb6=$v', d['value'], d['ms_per_step'])"; done
