echo "library: $(HOTRACK_LINEAR_K128_MIN_ROWS=0 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | cut -c60-110)"
for w in 256 224 192 160; do echo "k128 wgs $w: $(PN2_LK_WGS=$w python bench.py --no-cpu-baseline --no-legs 2>/dev/null | cut -c60-110)"; done
echo "library: $(HOTRACK_LINEAR_K128_MIN_ROWS=0 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | cut -c60-110)"
