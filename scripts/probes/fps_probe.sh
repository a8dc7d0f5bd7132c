# timing probes of the FPS iteration (results are WRONG with a probe on; timing only): which link of the chain costs what
R=$GRAFT_REPO_ROOT
for v in 0 1 2 4; do
  touch $R/hotrack_amd/csrc/fps.hip
  (cd $R && PN2_EXTRA_HIPCC_FLAGS="-DPN2_FPS_PROBE=$v" python -c "from hotrack_amd import _build; _build.build()" > /dev/null 2>&1)
  echo "== probe $v"
  (cd $R && python scripts/bench_ops.py 2>/dev/null | grep '"fps"' | grep '"threads": "0"' | cut -c1-140)
done
touch $R/hotrack_amd/csrc/fps.hip
