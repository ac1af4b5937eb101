echo base; python tools/rec_probe.py 1100
for v in 1 2 3; do echo variant $v; CHIRON_AMD_LIB=$PWD/build/libchiron_lstm_CHIRON_W32_VARIANT_$v.so python tools/rec_probe.py 1100; done
