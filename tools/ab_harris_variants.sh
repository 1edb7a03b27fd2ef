cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2; do for v in 2_1 d; do cp rustcv_amd/csrc/build/variants/lib_hf_$v.so rustcv_amd/librustcv_hip.so; python bench.py --config 5 --steps 30 --warmup 5 --no-cpu --no-verify 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"; done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
