# How the variants are built (from rustcv_amd/csrc, after `make`):  mkdir -p build/variants;  OBJS=$(ls build/*.o | grep -v "membench\|_bench.o\|<file>.o");
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -D<MACRO>=<value> -c <file>.hip -o build/variants/x.o;
#   hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_<prefix>_<name>.so $OBJS build/variants/x.o      (the default build: cp ../librustcv_hip.so build/variants/lib_<prefix>_d.so)
# A/B of product-library variants on BASELINE config 5 (64 x 4K Harris pipeline): bash tools/ab_harris_variants.sh <variant> ...   (variants: rustcv_amd/csrc/build/variants/lib_hf_<v>.so; "d" = the default build)
cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2 3; do for v in "$@"; do cp rustcv_amd/csrc/build/variants/lib_hf_$v.so rustcv_amd/librustcv_hip.so; python bench.py --config 5 --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['verified'])"; done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
