python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16" 2>&1 | tail -4
for v in "" "CHIRON_NO_STREAM16=1"; do
env $v python tools/bench_configs.py f16 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: print(l.strip()); continue
    print('$v', j['config'], j['ms_per_batch'], j['kernels_ms'].get('conv_dma'), j['kernels_ms'].get('lstm_recurrence'))"
done
