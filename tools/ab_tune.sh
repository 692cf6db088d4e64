for t in nt_p8=2 nt_p8=1 nt_p8=2 nt_p8=1; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t', round(d['ms_per_step'],4), {k:round(v,4) for k,v in r['family_ms_per_step'].items()})"
done
