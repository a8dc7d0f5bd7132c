PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.satrace.so python scripts/probes/sa_trace.py 64 2>/dev/null
