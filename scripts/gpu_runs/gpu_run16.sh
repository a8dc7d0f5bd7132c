for w in 2048 1024 512 256 128; do echo "wgs $w: $(PN2_SA1_WGS=$w python scripts/probes/sa_layer1_bench.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:(v['sa_layer1_us'],v['sa_layer1_stats_us'],v['bn_stats_us']) for k,v in d.items()})")"; done
