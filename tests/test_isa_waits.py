"""The wait in front of the staging writes of the LDS-staged warp kernels, read from the compiler's output (CPU test: hipcc
cross-compiles gfx950 without a GPU).

gfx9 counts loads and stores in one in-order counter; the frame loops of these kernels are arranged (rcv_geom.hip: `run`) so that for
tiles inside the destination the compiler can leave the stores of the frame just computed outstanding -- `s_waitcnt vmcnt(N)` with
N = the number of those stores -- instead of `vmcnt(0)`.  That property is worth 5-25 % of the kernels' time
(profiles/r04_warp_store_wait_ablation.txt) and silently depends on control flow the next edit or compiler may change: pin it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def geom_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rcv_geom.s"
    flags = open(os.path.join(ROOT, "rustcv_amd", "csrc", "build", "flags.stamp")).read().split() if os.path.exists(
        os.path.join(ROOT, "rustcv_amd", "csrc", "build", "flags.stamp")) else "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
    flags = [f for f in flags if f not in ("-fPIC",)]
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "rustcv_amd", "csrc", "rcv_geom.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def _body(asm, mangled_part):
    i0 = next(i for i, l in enumerate(asm) if re.match(r"^_Z\S*" + re.escape(mangled_part) + r"\S*:", l))
    i1 = next(i for i in range(i0, len(asm)) if asm[i].startswith(".Lfunc_end"))   # (a kernel has several s_endpgm)
    return [l.strip().split(";")[0].strip() for l in asm[i0:i1]]


@pytest.mark.parametrize("kernel,store,nstores", [("17k_warp_affine_ldsILi3ELb0E", "global_store_dwordx3", 2),
                                                  ("16k_warp_gray_lds4ILi4E", "global_store_dword", 8),
                                                  ("14k_warp_f32_lds", "global_store_dword", 8)])
def test_staging_wait_leaves_the_stores_outstanding(geom_asm, kernel, store, nstores):
    body = [l for l in _body(geom_asm, kernel) if l.startswith(("global_", "s_waitcnt vmcnt", "ds_write", "s_barrier", "s_cbranch", "s_and_saveexec"))]   # (with the branches: a store behind a bounds test is not part of a run)
    # a run of exactly `nstores` unconditional stores, then the wait, then the LDS staging writes of the next frame
    hits = 0
    flow = ("s_cbranch", "s_and_saveexec")
    i = 1
    while i < len(body) - nstores:
        if not body[i].startswith(store + " ") or body[i - 1].startswith("global_store"):
            i += 1
            continue
        # a run of unconditional stores (a store behind a bounds test has a branch in front of it and ends the run), the waits
        # inside / behind it, then the staging writes of the next frame: every wait must leave the stores issued so far outstanding
        j, seen, waits = i, 0, []
        while j < len(body) and (body[j].startswith((store + " ", "s_waitcnt vmcnt(") + flow)):
            if body[j].startswith(store + " "):
                if seen and body[j - 1].startswith(flow):
                    break
                seen += 1
            elif body[j].startswith("s_waitcnt"):
                waits.append((seen, body[j]))
            j += 1
        if seen == nstores and waits and j < len(body) and body[j].startswith("ds_write_b128"):
            assert all(w == "s_waitcnt vmcnt(%d)" % k for k, w in waits), (kernel, waits)
            hits += 1
        i = max(j, i + 1)
    assert hits >= 2, (kernel, hits)   # both halves of the loop unrolled by two


@pytest.fixture(scope="module")
def warp_resize_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rcv_warp_resize.s"
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "rustcv_amd", "csrc", "rcv_warp_resize.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def test_staged_warp_resize_loop_head_leaves_the_store_in_flight(warp_resize_asm):
    """round 5, k_warp_resize_stage<4>: in the frame loop the four (two + two conditional ones) direct-to-LDS loads of the next frame are
    followed by the frame's arithmetic and ONE store; the wait in front of the loop's barrier must be vmcnt(1) -- the store, the youngest
    operation, stays in flight (vmcnt(0) there cost 4-7 % of the launch) -- and nothing between a frame's loads and that barrier may
    wait for vmcnt(0); the taps are read as ds_read2_b32 + ds_read_b32 (a wider read that is not naturally aligned is served one lane
    per cycle: profiles/r05_ubench_lds_taps.txt)"""
    body = _body(warp_resize_asm, "19k_warp_resize_stageILi4ELi0ELi5E")
    keep = [l for l in body if l.startswith(("buffer_load_dwordx4", "buffer_store_dword", "s_waitcnt vmcnt", "s_barrier", "ds_read"))]
    lds_loads = [i for i, l in enumerate(keep) if l.startswith("buffer_load_dwordx4") and l.endswith("lds")]
    assert len(lds_loads) == 12, keep         # prologue + two unrolled frames, four loads each (the fourth: the 64 slots past 768, wave 0)
    first_loop_load = lds_loads[4]
    loop = keep[first_loop_load:]
    # per unrolled frame: loads, tap reads, one store, then `s_waitcnt vmcnt(1)` directly in front of the barrier
    barriers = [i for i, l in enumerate(loop) if l == "s_barrier"]
    assert len(barriers) >= 1
    for b in barriers:
        assert loop[b - 1] == "s_waitcnt vmcnt(1)", loop[max(0, b - 4):b + 1]
    assert "s_waitcnt vmcnt(0)" not in loop, [l for l in loop if "vmcnt" in l]
    assert sum(l.startswith("buffer_store_dword") for l in loop) == 2
    # ... and inside a loop half every staged load is OLDER than the half's store (a load scheduled below the store would still
    # be in flight across the next barrier; the kernel pins the order with scheduling barriers): between two barriers, no load after the store
    segs, cur = [], []
    for l in loop:
        if l == "s_barrier":
            segs.append(cur)
            cur = []
        else:
            cur.append(l)
    for seg in segs:
        st = [i for i, l in enumerate(seg) if l.startswith("buffer_store_dword")]
        if st:
            assert not [l for l in seg[st[0]:] if l.startswith("buffer_load_dwordx4") and l.endswith("lds")], seg
    reads = [l for l in loop if l.startswith("ds_read")]
    assert reads and all(l.startswith(("ds_read2_b32", "ds_read_b32")) for l in reads), sorted(set(l.split()[0] for l in reads))
    # ... and the wait in front of the FIRST loop barrier (entry edge: loads of the first frame + the shaping store) is vmcnt(1) as well
    pre = keep[:first_loop_load]
    pb = max(i for i, l in enumerate(pre) if l == "s_barrier")
    assert pre[pb - 1] == "s_waitcnt vmcnt(1)" and pre[pb - 2].startswith("buffer_store_dword"), pre[pb - 4:pb + 1]


def test_opsel_pack_blocks_carry_their_own_mfma_wait(tmp_path):
    """round 6: rcv_ashr_sat_pk12_mfma packs twelve MFMA accumulators with six v_ashr_pk_u8_i32 in inline asm (the second of a pair with op_sel[3]).  The
    hazard recognizer does not look into inline asm, so the block must open with the wait the compiler itself puts between v_mfma_i32_16x16x64_i8 and
    a dependent VALU instruction on gfx950 -- `s_nop 7`, eight wait states -- wherever the scheduler places it (many blocks sit directly behind an MFMA).
    Checked in the ISA of the row filter: every op_sel block is s_nop 7 + three plain + three op_sel instructions, and the compiler's own figure is read
    from a two-instruction probe compiled next to it."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
    probe = tmp_path / "probe.hip"
    probe.write_text("""#include <hip/hip_runtime.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i* in, unsigned* out, int sh) {
    v4i a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128];
    v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    out[threadIdx.x] = (unsigned)(unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(acc[0], acc[1], sh);
}
""")
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(tmp_path / "probe.s"), str(probe)], stderr=subprocess.DEVNULL)
    pl = [l.strip() for l in open(tmp_path / "probe.s") if l.strip() and not l.strip().startswith(";")]
    i = next(i for i, l in enumerate(pl) if l.startswith("v_mfma_i32_16x16x64_i8"))
    assert pl[i + 1] == "s_nop 7" and pl[i + 2].startswith("v_ashr_pk_u8_i32"), pl[i:i + 3]      # the compiler's own wait: the helper must carry no less
    out = tmp_path / "rows.s"
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "rustcv_amd", "csrc", "rcv_filter_rows_mfma.hip")],
                          stderr=subprocess.DEVNULL)
    lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith(";")]
    blocks = 0
    for i, l in enumerate(lines):
        if l.startswith("v_ashr_pk_u8_i32") and "op_sel" in l and not (lines[i - 1].startswith("v_ashr_pk_u8_i32") and "op_sel" in lines[i - 1]):
            # first op_sel instruction of a block: three plain ones and the wait in front of it
            assert all(lines[i - j].startswith("v_ashr_pk_u8_i32") and "op_sel" not in lines[i - j] for j in (1, 2, 3)), lines[i - 4:i + 3]
            assert lines[i - 4] == "s_nop 7", lines[i - 5:i + 1]
            assert all(lines[i + j].startswith("v_ashr_pk_u8_i32") and "op_sel" in lines[i + j] for j in (1, 2)), lines[i:i + 3]
            blocks += 1
    assert blocks >= 100, blocks


@pytest.fixture(scope="module")
def harris_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rcv_harris_fused.s"
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "rustcv_amd", "csrc", "rcv_harris_fused.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def test_harris_fused_row_loop_has_no_vector_address_math_and_no_pair_copies(harris_asm):
    """round 6, read from the compiler's output of the mask-only BGR instantiation k_harris_fused<false, 0, true, false>: the source rows
    come through BUFFER loads whose row offset is a scalar register (no v_lshl_add_u64 / v_mad_u64_u32 per row), the strip-end neighbour pairs are
    formed by IN-PLACE DPP moves (v_mov_b32_dpp vN, vN: five per row -- two in the Sobel stage, three in the box stage), the swapped pairs are read
    through op_sel, three waves per SIMD without scratch.  What this pins is worth 8 % of config 5 (profiles/r06_harris_timeline.txt, boxes D / H)."""
    body = _body(harris_asm, "14k_harris_fusedILb0ELi0ELb1ELb0E")
    i0 = next(i for i, l in enumerate(body) if l.startswith("buffer_load_dwordx4"))
    loop = body[i0:]
    loads = [l for l in loop if l.startswith("buffer_load_dword")]
    assert loads and all(re.search(r"s\[\d+:\d+\], s\d+ offen", l) for l in loads), loads[:4]
    assert not [l for l in loop if l.startswith(("global_load", "global_store", "flat_"))]
    assert not [l for l in loop if l.startswith(("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32"))]
    inplace = [l for l in loop if re.match(r"v_mov_b32_dpp (v\d+), \1 ", l)]
    swapped = len([l for l in loop if "v_pk_mul_f32" in l and "op_sel:[1,1] op_sel_hi:[0,0]" in l])   # the three products of pair 3, per row
    feeds = swapped // 3                                                                              # rows per trip of the unrolled loop (2 x kAhead)
    assert feeds == 4 and swapped == 3 * feeds, (swapped, feeds)
    assert len(inplace) == 5 * feeds, (len(inplace), feeds)
    meta = "\n".join(harris_asm)
    k = meta[meta.index("14k_harris_fusedILb0ELi0ELb1ELb0E"):]
    assert int(re.search(r"; NumVgprs: (\d+)", k).group(1)) <= 168 and int(re.search(r"; ScratchSize: (\d+)", k).group(1)) == 0
