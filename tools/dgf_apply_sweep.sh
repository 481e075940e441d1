cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 python -u tools/dgf_apply_times.py > gpurun_out/q_dgf.txt 2>&1 || { tail -30 gpurun_out/q_dgf.txt; exit 1; }
tail -1 gpurun_out/q_dgf.txt
for cfg in "hd --batch 8 --steps 60" "4k --steps 100" "hd --steps 200"; do
set -- $cfg; wl=$1; shift
timeout -k 10 300 python -u bench.py --workload $wl --no-cpu-baseline --no-also "$@" --dump-kernels gpurun_out/q_k.json > gpurun_out/q_b.txt 2>&1 || { tail -30 gpurun_out/q_b.txt; exit 1; }
python - "$wl$2$3" <<'PY'
import json, sys
line = open("gpurun_out/q_b.txt").read().strip().split("\n")[-1]
j = json.loads(line)
k = json.load(open("gpurun_out/q_k.json"))
d = {e["name"]: e["ms"] * 1e3 for e in k}
print(sys.argv[1], "fps %.1f" % j["value"], " dgf_apply %.2f us" % d.get("dgf_apply_fullres", 0))
PY
done
