#!/bin/bash
# rocprofv3 kernel trace of the inter-RAHT leg alone (after the rate sum's rewrite)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/r05_kt_inter
mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -o kt -- bash -c "cd $R && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile --legs raht_inter" > $R/$OUT/kt.log 2>&1 )
find $OUT/kt -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_inter.csv \;
rm -rf $OUT/kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_inter.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:8.1f}")
PY
