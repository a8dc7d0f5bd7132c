# headline against batches in flight / hardware queues / SA grid size (is 4 in flight on 8 queues still the best point?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { python bench.py --no-cpu-baseline --no-legs --min-time 3 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(r['value']), r['ms_per_step'])"; }
echo "inflight 4 (default): $(one)"
echo "inflight 3: $(one --inflight 3)"
echo "inflight 5: $(one --inflight 5)"
echo "inflight 6: $(one --inflight 6)"
echo "inflight 8: $(one --inflight 8)"
echo "inflight 6, 16 queues: $(GPU_MAX_HW_QUEUES=16 one --inflight 6)"
echo "inflight 4, sa-cus 240: $(one --sa-cus 240)"
echo "inflight 4 again: $(one)"
