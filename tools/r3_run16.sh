# round 3, GPU run 16: tools/store_align.hip -- do stores that leave lines partially written get filled from HBM?
set -x
O=gpurun_out/r3q
mkdir -p $O
R=$PWD
./tools/store_align.bin | tee $O/store_align.txt
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/$O/$C -- $R/tools/store_align.bin > /dev/null 2>&1
python $R/profiles/summarize_pmc.py $(find $R/$O/$C -name '*.db' | head -1) > $R/$O/pmc_$C.txt 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob('$R/$O/$C/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
suf = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = cur.execute(f"select d.id, sum(p.value) from rocpd_pmc_event{suf} p join rocpd_info_pmc{suf} i on p.pmc_id=i.id join rocpd_kernel_dispatch{suf} d on p.event_id=d.event_id where i.name='$C' group by d.id order by d.id").fetchall()
print('$C per dispatch (KB):', [round(v) for _, v in rows])
PY
rm -rf $R/$O/$C
done
