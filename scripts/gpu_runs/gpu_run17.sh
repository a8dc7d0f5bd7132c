O=gpurun_out/r05q; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 900 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for v in "" wgmreg1 wgmreg2 wgmnoload; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/hotrack_amd/libpn2_hip.$v.so"; fi
  echo "variant ${v:-dma}: $(PN2_LIB_PATH=$lib python scripts/probes/wgrad_bench.py 2>/dev/null)"
done | tee $O/variants.txt
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.wgmreg1.so python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/train_reg1.json; cut -c1-330 $O/train_reg1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tw -o t -- python $R/scripts/probes/wgrad_bench.py > /dev/null 2>&1
grep -i "wgm" $(find /tmp/tw -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee $R/$O/kernel_stats_wgm.txt
