cd /root/repo; REPO=$(pwd)
python tools/kernel_times.py synthetic1M > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_syn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_syn -o syn -- python $REPO/tools/kernel_times.py synthetic1M > /dev/null 2>&1
F=$(find /tmp/prof_syn -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
