python -m pytest tests/test_gpu_parity.py -q -x -k "wide_recurrence or paired_recurrence or saturated or trained_like or dna_logits or edge_cases or full_batch_1100 or randomized" 2>&1 | tail -15
echo "--- rec_probe wide"; python tools/rec_probe.py 1100 1024 512
echo "--- rec_probe narrow"; CHIRON_LSTM_NARROW=1 python tools/rec_probe.py 1100
echo "--- bench wide"; python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
echo "--- bench narrow"; CHIRON_LSTM_NARROW=1 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
