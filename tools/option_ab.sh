for o in "" "--option remap_px=2" "--option fb_chain=0" "--option fb_chain=1" "--option fb_segs=2" "--option fb_segs=3" ""; do
  echo "=== $o"
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-gate --no-alone $o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'])"
done
