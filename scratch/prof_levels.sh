cd /tmp && export TMPDIR=/tmp
for k in lidar dense; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$k -o t -- python /root/repo/scratch/one.py $k > /tmp/log_$k.txt 2>&1 || tail -5 /tmp/log_$k.txt
  f=$(find /tmp/prof_$k -name "*kernel_trace.csv" | head -1)
  echo "file: $f"
  python /root/repo/scratch/parse_trace.py "$f" $k
done
