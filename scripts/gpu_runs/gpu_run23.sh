for v in "" wgmrc16; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/hotrack_amd/libpn2_hip.$v.so"; fi
  echo "variant ${v:-base}: $(PN2_LIB_PATH=$lib python scripts/probes/wgrad_bench.py 2>/dev/null)"
done
