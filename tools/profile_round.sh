#!/bin/bash
# tools/profile_round.sh TAG: on the GPU box -- kernel-trace stats of the default bench + separate PMC passes for HBM traffic.
# Outputs under gpurun_out/prof_TAG/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.json 2> $O/write.err
cd $R
python3 - $O <<'PY'
import csv, glob, sys, json, collections
O = sys.argv[1]
out = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for f in glob.glob(O + "/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[name] = {k: {"n": len(v), "max": max(v), "sum": sum(v)} for k, v in agg.items()}
json.dump(out, open(O + "/pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
head -12 $O/kernel_stats.csv
