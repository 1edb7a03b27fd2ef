"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol include/rustcv_hip.h
declares (and ctypes agrees with the C struct layouts), host-only helpers work, and compute entry points
fail loudly -- not silently fall back -- when there is no GPU."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import rustcv_amd
from rustcv_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rustcv_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    assert sorted(_ffi.SIGNATURES) == syms  # the ctypes table covers exactly the header
    L = _ffi.lib()
    assert L.rcv_abi_version() == 1


def test_measurement_kernels_live_outside_the_product_library():
    """plain copies / store patterns / clock probe (bench.py's copy ceiling, tools/) are librustcv_hip_bench.so; the product
    library exports the header's entry points and the rcv__debug_* test hooks only"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    extra = sorted(e for e in exported if e.startswith("rcv_") and e not in _ffi.SIGNATURES and e not in _ffi.DEBUG_SIGNATURES)
    assert not extra, extra
    assert not any(name in exported for name in _ffi.BENCH_SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.BENCH_LIB_PATH], text=True)
    bench = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(_ffi.BENCH_SIGNATURES) <= bench
    _ffi.bench_lib()


def test_shard_range_is_the_partition_rule_of_the_harness():
    """rcv_shard_range (pure host arithmetic, no device) == rustcv_amd.shard.frame_range for every (n, world, rank) tried: contiguous,
    complete, sizes differ by at most one; bad arguments are refused"""
    import ctypes as C
    from rustcv_amd import shard
    L = _ffi.lib()
    a, b = C.c_int64(), C.c_int64()
    for n in list(range(0, 40)) + [64, 255, 256, 512, 1000003, 2 ** 40 + 7]:
        for world in (1, 2, 3, 4, 7, 8, 64):
            end = 0
            sizes = []
            for r in range(world):
                assert L.rcv_shard_range(n, r, world, C.byref(a), C.byref(b)) == 0
                assert (a.value, b.value) == shard.frame_range(n, r, world)
                assert a.value == end
                end = b.value
                sizes.append(b.value - a.value)
            assert end == n and max(sizes) - min(sizes) <= 1
    for bad in ((5, -1, 2), (5, 2, 2), (5, 0, 0), (-1, 0, 1)):
        assert L.rcv_shard_range(*bad, C.byref(a), C.byref(b)) == _ffi.RCV_ERR_ARG
    assert L.rcv_shard_range(5, 0, 1, None, C.byref(b)) == _ffi.RCV_ERR_ARG
    assert L.rcv_group_size(None) == _ffi.RCV_ERR_ARG and not L.rcv_group_ctx(None, 0)
    L.rcv_group_destroy(None)
    h = C.c_void_p()
    assert L.rcv_group_create(None, 0, C.byref(h)) == _ffi.RCV_ERR_ARG and not h


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rustcv_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(rcv_mat),offsetof(rcv_mat,cap),offsetof(rcv_mat,step),offsetof(rcv_mat,rows),offsetof(rcv_mat,channels),'
                   'offsetof(rcv_mat,device),sizeof(rcv_batch),offsetof(rcv_batch,frame_stride),offsetof(rcv_batch,n));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)], text=True).split()))
    m, b = _ffi.rcv_mat, _ffi.rcv_batch
    want = [C.sizeof(m), m.cap.offset, m.step.offset, m.rows.offset, m.channels.offset, m.device.offset,
            C.sizeof(b), b.frame_stride.offset, b.n.offset]
    assert got == want


def test_no_gpu_means_loud_failure_not_fallback():
    if rustcv_amd.device_count() > 0:
        pytest.skip("a GPU is present; this test is about the GPU-less container")
    with pytest.raises(rustcv_amd.RcvError) as e:
        rustcv_amd.Context(0)
    assert e.value.code == _ffi.RCV_ERR_DEVICE
    L = _ffi.lib()
    m = rustcv_amd.Mat(4, 4, 3)._as_rcv()
    assert L.rcv_rectangle(None, C.byref(m), 0, 0, 2, 2, 1, 2, 3, 1) == _ffi.RCV_ERR_ARG  # no ctx -> error, never a CPU path


def test_product_never_references_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "rustcv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                bad = re.search(r"liboracle|pyoracle|from\s+oracle|import\s+oracle|#include[^\n]*oracle|dlopen", txt)
                assert not bad, (os.path.join(base, f), bad.group(0))
    deps = subprocess.check_output(["readelf", "-d", _ffi.LIB_PATH], text=True)
    assert "oracle" not in deps


def test_fourcc_dispatch_mirrors_the_reference():
    from rustcv_amd import videoio
    L = _ffi.lib()
    code = C.c_int(-1)
    assert videoio.YUYV == 0x56595559  # 'Y','U','Y','V' little-endian (pixel_format.rs:10-12,39)
    for fcc, want in ((videoio.YUYV, _ffi.RCV_YUYV2BGR), (videoio.BGRA, _ffi.RCV_BGRA2BGR), (videoio.BGR4, _ffi.RCV_BGRA2BGR),
                      (videoio.RGB3, _ffi.RCV_RGB2BGR)):
        assert L.rcv_fourcc_to_code(fcc, C.byref(code)) == 0 and code.value == want
    assert L.rcv_fourcc_to_code(videoio.MJPEG, C.byref(code)) == _ffi.RCV_ERR_UNSUPPORTED
    assert L.rcv_fourcc_to_code(0xDEADBEEF, C.byref(code)) == _ffi.RCV_ERR_UNSUPPORTED


def test_gaussian_taps_helper_matches_oracle(oracle):
    L = _ffi.lib()
    for ks, sg in ((3, 0.8), (7, 1.5), (15, 3.3), (31, 6.0)):
        t = (C.c_float * ks)()
        assert L.rcv_gaussian_taps_f32(ks, sg, t) == 0
        assert np.array_equal(np.array(t[:], np.float32).view(np.uint32), oracle.gaussian_taps_f32(ks, sg).view(np.uint32))
    assert L.rcv_gaussian_taps_f32(4, 1.0, (C.c_float * 4)()) == _ffi.RCV_ERR_ARG
    assert L.rcv_strerror(-3).decode().startswith("buffer too small")


def test_cpp_facade_compiles_and_links(tmp_path):
    """include/rustcv.hpp + tests/cpp/facade_test.cpp (the reference's tests, C++ spelling) build against the .so."""
    exe = tmp_path / "facade_test"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "rustcv_amd"), "-lrustcv_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "rustcv_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)


# ---- the Rust side of the boundary (rust/rustcv-backend-hip): never compiled here (no rustc), so it is held against the C
# ---- header mechanically: names, arities, parameter kinds, struct fields in order, constants with their values ---------------
RUST_FFI = os.path.join(ROOT, "rust", "rustcv-backend-hip", "src", "ffi.rs")


def _c_kind(ctype):
    """coarse, language-neutral description of a C type: (pointer depth, const-ness of the pointee, base)"""
    t = " ".join(re.sub(r"/\*.*?\*/", "", ctype).split())
    const = t.startswith("const ")
    t = t[6:] if const else t
    base = t.replace("*", "").strip()
    base = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "int8_t": "i8", "size_t": "usize",
            "float": "f32", "double": "f64", "char": "c_char", "void": "void"}.get(base, base)
    return (t.count("*"), const and t.count("*") > 0, base)


def _rust_kind(rtype):
    t = rtype.strip()
    depth, const = 0, False
    while t.startswith("*"):
        depth += 1
        if t.startswith("*const "):
            t, const = t[7:], True
        else:
            t = t[5:]
    base = {"c_int": "i32", "c_void": "void"}.get(t.strip(), t.strip())
    return (depth, const, base)


def _parse_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    funcs = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(rcv_\w+)\s*\(([^;{]*?)\)\s*;", src, re.M | re.S):
        params = [] if m.group(3).strip() in ("", "void") else [" ".join(p.split()) for p in m.group(3).split(",")]
        funcs[m.group(2)] = (_c_kind(m.group(1)), [_c_kind(re.match(r"(.+?)\w+$", p).group(1)) for p in params])
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if decl:
                mm = re.match(r"((?:const\s+)?\w+\s*\**)\s*(.+)$", decl)
                fields += [(n.strip().lstrip("*"), _c_kind(mm.group(1) + "*" * n.count("*"))) for n in mm.group(2).split(",")]
        structs[m.group(2)] = fields
    consts = {m.group(1): int(m.group(2).strip("() ")) for m in re.finditer(r"^#define\s+(RCV_\w+)\s+(\(?-?\d+\)?)\s*$", src, re.M)}
    return funcs, structs, consts


def _parse_rust():
    src = open(RUST_FFI).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", block):
        params = [p.split(":", 1)[1] for p in m.group(2).split(", ")] if m.group(2).strip() else []
        funcs[m.group(1)] = (_rust_kind(m.group(3)) if m.group(3) else (0, False, "void"), [_rust_kind(p) for p in params])
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\n(?:#\[derive\([^)]*\)\]\n)?pub struct (\w+) \{(.*?)\n\}", src, re.S):
        fields = [(f.group(1).rstrip("_"), _rust_kind(f.group(2))) for f in re.finditer(r"pub (\w+): ([^,]+),", m.group(2))]
        structs[m.group(1)] = fields
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (RCV_\w+): c_int = (-?\d+);", src)}
    return funcs, structs, consts


def test_rust_ffi_declares_the_whole_header():
    """every function (name, arity, pointer depth / const-ness / base type of the result and of each parameter), every struct (fields
    in order) and every constant of include/rustcv_hip.h is declared identically in the Rust crate's extern "C" block -- the
    precedent declares the whole bridge (rustcv-camera/src/backend/macos/mod.rs:52-79 vs bridge.h:36-65)"""
    hf, hs, hc = _parse_header()
    rf, rs, rc = _parse_rust()
    assert sorted(hf) == declared_symbols()                 # the parser sees every prototype of the header
    assert sorted(rf) == sorted(hf), sorted(set(hf) ^ set(rf))
    for name in hf:
        assert rf[name] == hf[name], (name, rf[name], hf[name])
    opaque = {k for k, v in rs.items() if not v}             # zero-sized handles (only a private field)
    assert opaque == {"rcv_ctx", "rcv_ring", "rcv_import", "rcv_group"}
    for name, fields in hs.items():
        assert rs[name] == fields, (name, rs[name], fields)
    assert rc == hc and len(hc) >= 29


def test_rust_ffi_is_up_to_date():
    """rust/rustcv-backend-hip/src/ffi.rs is what tools/gen_rust_ffi.py derives from the header today"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(RUST_FFI).read() == gen.generate(), "stale: run python tools/gen_rust_ffi.py"
    lib_rs = open(os.path.join(ROOT, "rust", "rustcv-backend-hip", "src", "lib.rs")).read()
    assert "pub mod ffi;" in lib_rs and 'extern "C" {' not in lib_rs      # one declaration of the ABI, the generated one
    assert "as i32, channels: 1" not in lib_rs                              # (flat-buffer views must not truncate lengths)


def _rust_calls(src):
    """every `rcv_xxx(...)` CALL in a Rust source (not a declaration): name -> list of argument counts"""
    calls = {}
    for m in re.finditer(r"(?<![\w:])(rcv_\w+)\s*\(", src):
        if src[max(0, m.start() - 3):m.start()].endswith("fn "):
            continue
        i, depth, args, cur = m.end(), 1, 0, ""
        while i < len(src) and depth:
            c = src[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                args += 1 if cur.strip() else 0
                cur = ""
                i += 1
                continue
            if depth:
                cur += c
            i += 1
        args += 1 if cur.strip() else 0
        calls.setdefault(m.group(1), []).append(args)
    return calls


def test_rust_facade_wraps_every_entry_point():
    """round 3: the Rust crate is at parity with include/rustcv.hpp -- lib.rs + imgproc.rs together call EVERY
    non-debug function of include/rustcv_hip.h from a safe wrapper, each call with the header's number of arguments; the facade has
    the Mat-shaped borrows (rustcv/src/core/mat.rs:6-15), the owning DeviceBatch / StagingRing handles with Drop, and
    rgb_to_bgr (rustcv-camera/src/decode.rs:213)"""
    hf, _, _ = _parse_header()
    base = os.path.join(ROOT, "rust", "rustcv-backend-hip", "src")
    lib_rs, img_rs = open(os.path.join(base, "lib.rs")).read(), open(os.path.join(base, "imgproc.rs")).read()
    calls = _rust_calls(lib_rs + "\n" + img_rs)
    missing = sorted(n for n in hf if n not in calls)
    assert not missing, f"no Rust wrapper calls {missing}"
    for name, (_, params) in hf.items():
        for got in calls[name]:
            assert got == len(params), (name, got, len(params))
    assert not [n for n in calls if n not in hf and not n.startswith(("rcv_mat", "rcv_batch", "rcv_glyph", "rcv_ring_op"))], "call of an undeclared function"
    assert "pub mod imgproc;" in lib_rs
    for item in ("pub struct MatRef<", "pub struct MatMut<", "pub struct DeviceBatch<", "pub struct StagingRing<", "pub enum HipError",
                 "pub fn rgb_to_bgr(", "pub fn harris_pipeline_batch(", "pub fn filter2d_i8_batch(", "pub fn warp_affine_resize_batch("):
        assert item in img_rs, item
    assert "pub struct DeviceGroup" in lib_rs and "impl Drop for DeviceGroup" in lib_rs and "pub fn frame_range(" in lib_rs
    for handle in ("DeviceBatch", "StagingRing"):
        assert re.search(r"impl<'c> Drop for %s<'c>" % handle, img_rs), handle
    # every safe wrapper of a compute entry point exists under the C name minus its prefix
    for name in hf:
        if name.startswith(("rcv_ring_", "rcv_import_", "rcv_ctx_", "rcv_timer_", "rcv_group_")) or name in (
                "rcv_malloc", "rcv_free", "rcv_upload", "rcv_download", "rcv_memset", "rcv_sync", "rcv_strerror", "rcv_synth_batch", "rcv_shard_range"):
            continue
        assert re.search(r"pub fn %s\(" % name[4:], lib_rs + img_rs), name
    assert img_rs.count("{") == img_rs.count("}") and img_rs.count("(") == img_rs.count(")")   # (no compiler here: at least balanced)
    assert lib_rs.count("{") == lib_rs.count("}") and lib_rs.count("(") == lib_rs.count(")")


def test_product_library_is_slim():
    """round 4: the product library reads fourteen environment knobs -- thirteen dispatch overrides for the tests and (round 6)
    one fault injection for the chained kernel's completion check, no tuning parameter -- exports no measurement entry (those live in librustcv_hip_bench.so) and carries no launch-graph API any more"""
    import re
    import subprocess
    src = open(os.path.join(ROOT, "rustcv_amd", "csrc", "rcv_ctx.hip")).read()
    knobs = set(re.findall(r'"(RCV_[A-Z0-9_]+)"', src[src.index("static void load_knobs()"):src.index("const RcvKnobs& rcv_knobs()")]))
    assert len(knobs) == 14, sorted(knobs)
    others = set()
    for f in os.listdir(os.path.join(ROOT, "rustcv_amd", "csrc")):
        if f.endswith((".hip", ".h")) and f not in ("rcv_ctx.hip", "rcv_membench.hip"):
            others |= set(re.findall(r'getenv\("(RCV_[A-Z0-9_]+)"\)', open(os.path.join(ROOT, "rustcv_amd", "csrc", f)).read()))
    assert not others, others                      # (no file reads the environment behind the table's back)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for k in knobs:
        assert k in design, f"{k} is not documented in DESIGN.md"
    syms = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (rcv_\w+)", syms))
    assert {s for s in exported if s.startswith("rcv__")} == {"rcv__debug_kernels", "rcv__debug_kernels_reset", "rcv__debug_occupancy", "rcv__debug_reload_knobs"}
    assert not [s for s in exported if "graph" in s or "bench" in s or "stripwalk" in s]
    bench = subprocess.run(["nm", "-D", "--defined-only", _ffi.BENCH_LIB_PATH], capture_output=True, text=True).stdout
    assert {"rcv__filter_rows_bench", "rcv__warp_resize_bench", "rcv__membench", "rcv__stripwalk", "rcv__storebench", "rcv__clock_probe"} <= set(re.findall(r" T (rcv_\w+)", bench))


def test_design_md_is_the_short_current_state_document():
    """DESIGN.md = current state in <= 200 lines of <= 160 characters; the per-round studies live in DESIGN_HISTORY.md"""
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")
    assert len(lines) <= 200, len(lines)
    assert max(len(l) for l in lines) <= 160, [i + 1 for i, l in enumerate(lines) if len(l) > 160]
    assert os.path.exists(os.path.join(ROOT, "DESIGN_HISTORY.md"))
    text = "\n".join(lines)
    for must in ("k_filter_rows_chain", "k_warp_resize_stage", "k_harris_fused", "parity unpinned", "no collective", "Out of scope"):
        assert must in text, must
