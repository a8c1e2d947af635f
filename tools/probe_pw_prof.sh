R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pwprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --secondary= --full-model= --no-roofline --regime trained --trained-steps 100 --no-render --workload mixed16384_neuradar_full_fp16 > $O/log.txt 2>&1
cd $R; find $O -name "out_kernel_trace.csv" -delete
python - <<'PY'
import csv,re,os
f=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pwprof/stats/out_kernel_stats.csv'
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(anonymous namespace\)::|void ","",r['Name'])
    if 'pw_' in n or 'conv7' in n or 'bn_' in n: print(n[:100], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
