O=gpurun_out/r05o; mkdir -p $O
R=$PWD
for v in "" wgm2 wgmrc64 wgmnoload wgmnobar wgmnoload_nobar; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/hotrack_amd/libpn2_hip.$v.so"; fi
  echo "variant ${v:-base}: $(PN2_LIB_PATH=$lib python scripts/probes/wgrad_bench.py 2>/dev/null)"
done | tee $O/variants.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tw -o t -- python $R/scripts/probes/wgrad_bench.py > /dev/null 2>&1
grep -i "wgm" $(find /tmp/tw -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee $R/$O/kernel_stats_wgm.txt
