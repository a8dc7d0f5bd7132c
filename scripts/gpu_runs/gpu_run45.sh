PN2_LIB_PATH=hotrack_amd/libpn2_hip.satrace.so python scripts/probes/sa1_trace.py
