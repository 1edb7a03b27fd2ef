# (round 5) rows in flight of the separable f32 stream kernel: variants lib_fs_ah3 / lib_fs_ah4 (-DRCV_FS_AHEAD=3 / 4 of rcv_filter_f32_stream.hip) against the default (2): no difference
# How the variants are built (from rustcv_amd/csrc, after `make`):  mkdir -p build/variants;  OBJS=$(ls build/*.o | grep -v "membench\|_bench.o\|<file>.o");
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -D<MACRO>=<value> -c <file>.hip -o build/variants/x.o;
#   hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_<prefix>_<name>.so $OBJS build/variants/x.o      (the default build: cp ../librustcv_hip.so build/variants/lib_<prefix>_d.so)
cp rustcv_amd/librustcv_hip.so /tmp/orig.so
for r in 1 2; do for v in d ah3 ah4; do cp rustcv_amd/csrc/build/variants/lib_fs_$v.so rustcv_amd/librustcv_hip.so; RCV_GAUSS_ROWS=0 python tools/bench_ops.py --steps 20 --warmup 5 --only "sigma=1.5" 2>&1 | grep "sigma" | sed "s/^/$v (one-row kernel) /" | cut -c1-150; done; done
cp /tmp/orig.so rustcv_amd/librustcv_hip.so
