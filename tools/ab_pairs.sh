# step time per pair vs batch size: does a working set nearer the 256 MB Infinity Cache run faster per window?
for p in 16 32 64 128 256; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --pairs $p 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pairs $p  ms %.4f  us/pair %.2f  value %.0f' % (d['ms_per_step'], d['ms_per_step']*1e3/$p, d['value']))"
done
