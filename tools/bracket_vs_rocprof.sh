R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/agree; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload train --no-cpu-baseline 2>/dev/null | grep "^{" > $O/train.json
python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves 2>/dev/null | grep "^{" > $O/hps.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train -- python3 $R/bench.py --workload train --no-cpu-baseline > $O/p_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_hps -- python3 $R/bench.py --workload hpsearch --steps 10 --warmup 3 --no-cpu-baseline --no-loss-curves > $O/p_hps.log 2>&1
python3 - <<PY
import json,glob,csv
for w,f in (("train","$O/train.json"),("hps","$O/hps.json")):
    d=json.load(open(f)); r=d["roofline"]; print(w,"un-profiled: step",d["ms_per_step"],"kernel",r["kernel"],r["kernel_ms"],r.get("measured"))
for w,log,pat in (("train","$O/p_train.log","bn_bwd_pw_wgrad_kernel<2, 2, 4>"),("hps","$O/p_hps.log","bn_bwd_pw_wgrad_h_kernelILi2ELi2")):
    line=[l for l in open(log) if l.startswith("{")][-1]; d=json.loads(line); r=d["roofline"]
    st=glob.glob("$O/p_"+w+"/**/*kernel_stats.csv",recursive=True)[0]
    row=[x for x in csv.DictReader(open(st)) if pat in x["Name"]][0]
    print(w,"profiled: step",d["ms_per_step"],"bracket",r["kernel_ms"],"rocprofv3 average",float(row["AverageNs"])/1e6,"calls",row["Calls"])
PY
find $O -name "*kernel_trace.csv" -delete
