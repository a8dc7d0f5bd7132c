# timing probes of the SA kernel's LOAD role on the GPU box (results are wrong by construction for SA_PROBE != 0)
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch $R/hotrack_amd/csrc/sa_fused.hip
  (cd $R && PN2_EXTRA_HIPCC_FLAGS="-DSA_TRACE=1 $cfg" python -c "from hotrack_amd import _build; _build.build()" > /dev/null 2>&1)
  echo "== ${cfg:-default}"
  (cd $R && python scripts/probes/sa_trace.py 64 | sed -n "3,4p;10,11p")
done
