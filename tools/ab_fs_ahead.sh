cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2; do for v in d ah3 ah4; do cp rustcv_amd/csrc/build/variants/lib_fs_$v.so rustcv_amd/librustcv_hip.so; RCV_GAUSS_ROWS=0 python tools/bench_ops.py --steps 20 --warmup 5 --only "sigma=1.5" 2>&1 | grep "sigma" | sed "s/^/$v (one-row kernel) /" | cut -c1-150; done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
