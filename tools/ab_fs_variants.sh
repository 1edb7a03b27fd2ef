# How the variants are built (from rustcv_amd/csrc, after `make`):  mkdir -p build/variants;  OBJS=$(ls build/*.o | grep -v "membench\|_bench.o\|<file>.o");
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -D<MACRO>=<value> -c <file>.hip -o build/variants/x.o;
#   hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_<prefix>_<name>.so $OBJS build/variants/x.o      (the default build: cp ../librustcv_hip.so build/variants/lib_<prefix>_d.so)
# same-call A/B of product-library variants of rcv_filter_f32_stream.hip (rustcv_amd/csrc/build/variants/lib_fs_<v>.so) on the f32 / sigma rows of tools/bench_ops.py and tools/ab_gauss_sigma.py
cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2 3; do for v in "$@"; do cp rustcv_amd/csrc/build/variants/lib_fs_$v.so rustcv_amd/librustcv_hip.so
  python tools/bench_ops.py --steps 20 --warmup 5 --only "f32" 2>&1 | grep "filter2D 7x7 f32" | sed "s/^/$v /" | cut -c1-120
  python tools/ab_gauss_sigma.py 2>&1 | head -8 | sed "s/^/$v /" | cut -c1-120
done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
