#!/bin/bash
# usage: tools/tune.sh "VAR=val VAR2=val" ...   -- runs bench (philox, with per-kernel table) per setting
for setting in "$@"; do
  echo "=== $setting"
  env $setting timeout -k 10 100 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --sampler philox 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'])
print('  '+' | '.join('%s[%d] %.1f'%(k['kernel'],k['batch'],k['avg_us']) for k in sorted(d['kernels'],key=lambda k:(k['batch'],k['kernel']))))
"
done
