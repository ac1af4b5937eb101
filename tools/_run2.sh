cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for form in narrow wide; do
  if [ $form = narrow ]; then export CHIRON_LSTM_NARROW=1; else unset CHIRON_LSTM_NARROW; fi
  rocprofv3 --kernel-trace --output-format csv -d $O/trace3_$form -o slots3 -- python bench.py --slots 3 --steps 12 --rounds 1 --warmup 3 --no-f16 > $O/trace3_$form.json 2> $O/trace3_$form.err
  f=$(find $O/trace3_$form -name '*kernel_trace.csv' | head -1)
  echo "=== $form $f"; python tools/overlap_trace.py $f 0.5 | tee $O/overlap_slots3_$form.txt
done
