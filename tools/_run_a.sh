for o in ""; do
  timeout 400 python bench.py --steps 24 --warmup 5 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$o', d['value'], d['ms_per_step'], d.get('host_cpu_seconds_per_step'), d['roofline_align'].get('ms_per_step'), d['roofline_align'].get('isolated'), d.get('gpu_kernel_ms_per_step'))
        print(d.get('kernels_single_sample'))
"
done
