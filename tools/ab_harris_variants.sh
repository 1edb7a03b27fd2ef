# A/B of product-library variants on BASELINE config 5 (64 x 4K Harris pipeline): bash tools/ab_harris_variants.sh <variant> ...   (variants: rustcv_amd/csrc/build/variants/lib_hf_<v>.so; "d" = the default build)
cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2 3; do for v in "$@"; do cp rustcv_amd/csrc/build/variants/lib_hf_$v.so rustcv_amd/librustcv_hip.so; python bench.py --config 5 --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['verified'])"; done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
