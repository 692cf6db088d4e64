for rep in 1 2; do
for T in "" "f1_blocks=512" "f1_blocks=768" "f1_blocks=1536" "f1_fwd_blocks=2048" "f1_fwd_blocks=8192"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras ${T:+--tune $T} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %.4f ms' % ('${T:-default}', d['ms_per_step']))"
done; done
