# builds sa_fused.hip with each flag set ON THE GPU BOX and prints the kernel-only SA timings (scripts/probes/sa_bench.py)
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch $R/hotrack_amd/csrc/sa_fused.hip
  (cd $R && PN2_EXTRA_HIPCC_FLAGS="$cfg" python -c "from hotrack_amd import _build; _build.build()" > /dev/null 2>&1)
  echo "== ${cfg:-default}"
  (cd $R && python scripts/probes/sa_bench.py 2>&1 | cut -c1-72)
done
