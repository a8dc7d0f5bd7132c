R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in inline nowait samestream; do
rm -rf /tmp/th
if [ $v = inline ]; then export HOTRACK_PREFETCH_GEOMETRY=0; else export HOTRACK_PREFETCH_GEOMETRY=1 HOTRACK_GEO_PROBE=$v; fi
rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/th -o t -- python $R/scripts/bench_train.py --graph --steps 10 --warmup 4 > /tmp/th.log 2>&1
echo "== $v $(grep -o '"ms_per_step": [0-9.]*' /tmp/th.log)"
python $R/scripts/trace_host_vs_device.py /tmp/th > $O/host_vs_device_$v.txt 2>&1; tail -45 $O/host_vs_device_$v.txt | grep -v copyBuffer
done
