R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mb -- $R/tools/microbench/${MB:-floor} > $R/gpurun_out/mb_stdout.txt 2>&1
echo exit=$?
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/mb/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//2:]
acc = collections.OrderedDict()
prev = None
for r in rows:
    wg = int(r['Workgroup_Size_X']); blocks = int(r['Grid_Size_X']) // wg
    k = (r['Kernel_Name'][:28], blocks, wg)
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    gap = int(r['Start_Timestamp']) - int(prev['End_Timestamp']) if prev else 0
    a = acc.setdefault(k, [0, 0, 0]); a[0] += d; a[1] += gap; a[2] += 1
    prev = r
with open('gpurun_out/mb_table.txt', 'w') as o:
    for k, a in acc.items():
        o.write('%-30s blocks %5d x %4d  dur %6.2f us  gap %5.2f us\n' % (k[0], k[1], k[2], a[0]/a[2]/1e3, a[1]/a[2]/1e3))
PY
rm -rf gpurun_out/mb
# build:  for f in floor floor2 cufetch atomics gridbar; do hipcc -O3 --offload-arch=gfx950 -o tools/microbench/$f tools/microbench/$f.hip; done
