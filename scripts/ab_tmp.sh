for rep in 1 2; do
for v in 0 384 256; do
  DM_WGRAD_PW=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PW=$v', d['ms_per_step'])"
done; done
