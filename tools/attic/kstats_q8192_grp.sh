cd /root/repo; export TMPDIR=/tmp; o=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $o/tmpg -o x -- python3 tools/run_q8192.py 8192 grp.lasso > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tmpg/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "oemgpu" in r["Name"] and int(r["Calls"])>100:
        print("  %-45s calls %5s avg %9.1f ns min %7s max %7s" % (r["Name"].split("(")[0][-45:], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
rm -rf $o/tmpg
