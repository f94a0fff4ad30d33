for cfg in "1 640" "2 320" "4 160" "8 80" "2 512" "3 256"; do set -- $cfg; echo "sessions=$1 capacity=$2"; TTX_POOL_SESSIONS=$1 TTX_POOL_CAPACITY=$2 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  value', round(d['value'],1), 'device steps', d.get('device_model_calls'), 'ms', round(d['ms_per_step']*d['steps'],1))"; done
