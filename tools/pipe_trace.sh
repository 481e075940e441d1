# rocprofv3 kernel trace of pipelined frames: how much do the two phases overlap on the device?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/bench.py --steps 60 --warmup 10 --pipeline on --no-cpu-baseline --no-also ${ARGS} > $R/gpurun_out/prof_pipe_stdout.txt 2>&1
echo exit=$?
cd $R
f=$(ls gpurun_out/prof_pipe/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
print("columns:", list(rows[0].keys()))
# the timed region of the headline: find a run of frames in the middle
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'downsample' in n]
mid = idx[len(idx) // 3: len(idx) // 3 + 12]
q = collections.Counter(r.get('Queue_Id', '?') for r in rows[mid[0]:mid[-1]])
print("queues in window:", dict(q))
t0 = int(rows[mid[0]]['Start_Timestamp'])
for i in range(mid[0], mid[3]):
    r = rows[i]
    print("%8.1f %8.1f  q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:50]))
per = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in zip(mid, mid[1:])]
print("downsample-to-downsample us:", ["%.0f" % p for p in per])
PY
rm -rf gpurun_out/prof_pipe
