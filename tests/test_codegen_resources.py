"""Guards against silent codegen drift in the hand-scheduled kernels (no GPU needed: hipcc cross-compiles gfx950).

The filter kernels keep their stationary operand in AGPRs through inline asm at __launch_bounds__(..., 1 or 2), count
their own vmcnt, hard-code s[88:95] in the sibling rendezvous and rely on explicit wait states: a hipcc bump that
spills, renumbers or drops occupancy would still pass the parity tests on small inputs while losing the performance
(or, with scratch, the timing assumptions).  `make -C nano-vectordb_amd asm` dumps the per-kernel resource usage
(-Rpass-analysis=kernel-resource-usage) and the ISA; this test reads both."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nano-vectordb_amd")


@pytest.fixture(scope="module")
def codegen():
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-C", PKG, "asm"], stdout=subprocess.DEVNULL)
    usage, cur = {}, None
    for line in open(os.path.join(PKG, "build", "resource_usage.txt")):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            usage[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            usage[cur][m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"] + list(usage), capture_output=True, text=True, check=True).stdout.splitlines()
    by_name = {}
    for mangled, pretty in zip(usage, names):
        short = re.sub(r"\(.*", "", pretty).replace("void nvdbhip::", "").replace("nvdbhip::", "")
        u = usage[mangled]
        by_name[short if short != mangled else mangled] = dict(
            mangled=mangled, sgpr=int(u["TotalSGPRs"]), vgpr=int(u["VGPRs"]), agpr=int(u["AGPRs"]), scratch=int(u["ScratchSize"]),
            occ=int(u["Occupancy"]), sspill=int(u["SGPRs Spill"]), vspill=int(u["VGPRs Spill"]))
    asm = open(os.path.join(PKG, "build", "nvdb_hip.s")).read()
    return by_name, asm


def _body(asm, mangled):
    a = asm.index(f"\n{mangled}:")
    return asm[a:asm.index("s_endpgm", a)]


def _find(by_name, *parts):
    hits = [k for k in by_name if all(p in k or p in by_name[k]["mangled"] for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return by_name[hits[0]]


def test_no_kernel_uses_scratch_or_spills(codegen):
    by_name, _ = codegen
    assert len(by_name) > 100
    bad = {k: v for k, v in by_name.items() if v["scratch"] or v["vspill"]}
    assert not bad, bad
    # SGPRs parked in VGPR lanes cost nothing in memory; only the exact fp32 kernels at 8 queries per group do it
    parked = {k for k, v in by_name.items() if v["sspill"]}
    # i8p: the 4-slot deferred queue is uniform state; i8w (the A/B alternative, not the default): the XCD-balance stamp and label
    assert all(k.startswith(("scan_exact_kernel<", "filter_i8p_kernel<", "filter_i8w_kernel<", "filter_i8s_kernel<768, true, true")) for k in parked), parked   # (i8s: its stamped developer build only)


def test_register_budgets_of_the_production_kernels(codegen):
    by_name, _ = codegen
    # headline kernel: 8 waves per workgroup, two per SIMD -> 128 VGPRs + 128 AGPRs, no more
    k = _find(by_name, "filter_f16_m16_kernelILi768ELi4ELb1ELb0ELi0ELi2ELi2ELi8E")
    assert k["vgpr"] <= 128 and k["agpr"] <= 128 and k["occ"] == 2, k
    # one-wave-per-SIMD builds: the whole 512-entry file, never beyond
    for key, v in by_name.items():
        if "filter_" in key or "filter_" in v["mangled"]:
            assert v["vgpr"] + v["agpr"] <= 512 and v["occ"] >= 1, (key, v)
    k = _find(by_name, "filter_i8w_kernel<768, 2, 6, true, 2, false, 0>")
    assert k["agpr"] >= 192 and k["occ"] == 1, k                      # the hi plane of 64 queries stays resident in AGPRs
    k = _find(by_name, "filter_i8p_kernel<768, true, false, 6, 0, 4, false>")
    assert k["agpr"] >= 192 and k["occ"] == 1 and k["sspill"] == 0, k  # the pipelined build (default: first-stage survivors logged): same residency, no scratch (checked above)
    k = _find(by_name, "filter_i8p_kernel<768, true, false, 6, 0, 4, true>")
    assert k["agpr"] >= 192 and k["occ"] == 1, k                      # ... with the in-loop second stage (option i8_defer)
    k = _find(by_name, "filter_i8p_kernel<768, true, false, 6, 0, 8, true>")
    assert k["vgpr"] <= 128 and k["agpr"] <= 128 and k["agpr"] >= 96 and k["occ"] == 2, k   # its 8-wave variant (option i8_waves8): 32 queries per wave, two waves per SIMD
    k = _find(by_name, "filter_i8s_kernel<768, true, false, 6, 0, 4>")
    assert k["agpr"] >= 192 and k["occ"] == 1 and k["sspill"] == 0, k  # the product's int8 batch-1024 kernel (16x16x64 MFMA): hi plane of 64 queries in AGPRs
    # exact fp32-order scores on the fp32 matrix cores: 16 queries x 768 floats stationary (192 registers) + a tile of raw rows in flight
    for name in ("scan_exact_mfma_kernel<2, 768>", "scan_exact_mfma_kernel<1, 768>", "scan_exact_mfma_kernel<3, 768>", "scores_exact_mfma_kernel<2, 768>"):
        k = _find(by_name, name)
        assert k["vgpr"] + k["agpr"] <= 512 and k["occ"] == 1, (name, k)
    k = _find(by_name, "filter_f16_kernelILi768ELi1ELi0ELi6E")
    assert k["agpr"] >= 192, k
    # refine v3: two workgroups of three waves per CU -> at most 256 registers per lane
    k = _find(by_name, "refine_l2_rows_kernel<768>")
    assert k["vgpr"] + k["agpr"] <= 256 and k["occ"] >= 2, k


def test_rendezvous_registers_and_hand_placed_instructions(codegen):
    by_name, asm = codegen
    n_sync = 0
    for key, v in by_name.items():
        body = _body(asm, v["mangled"])
        if "s_load_dwordx8 s[88:95]" in body:
            n_sync += 1
            assert v["sgpr"] >= 96, (key, v)                             # s88..s95 are inside the kernel's allocation
            # the block is intact: the load, its wait, then the eight moves out of s88..s95 (nothing in between)
            for blk in re.finditer(r"s_load_dwordx8 s\[88:95\][^\n]*\n\s*s_waitcnt lgkmcnt\(0\)\n((?:\s*s_mov_b32 (?:s\d+|vcc_lo|vcc_hi), s(?:8[89]|9[0-5])\n){8})", body):
                assert sorted(re.findall(r", s(\d+)\n", blk.group(1))) == [str(x) for x in range(88, 96)]
            assert len(re.findall(r"s_load_dwordx8 s\[88:95\]", body)) == len(re.findall(r"s_load_dwordx8 s\[88:95\][^\n]*\n\s*s_waitcnt lgkmcnt\(0\)\n(?:\s*s_mov_b32 (?:s\d+|vcc_lo|vcc_hi), s(?:8[89]|9[0-5])\n){8}", body))
    assert n_sync >= 8
    # the compiler keeps nothing live in s88..s95 across the block because the asm statement declares them clobbered
    src = open(os.path.join(PKG, "csrc", "kernels_filter.h")).read()
    blk = src[src.index('asm volatile("s_load_dwordx8 s[88:95]'):]
    blk = blk[:blk.index(");") + 2]
    assert all(f'"s{r}"' in blk for r in range(88, 96)), "clobber list of the sibling rendezvous"
    # headline kernel: 192 MFMAs per tile and wave pair -> 96 in the 8-wave build's loop body, fed from AGPRs; the
    # compiler must not have copied fragments into VGPRs (v_accvgpr_read inside the loop)
    k = _find(by_name, "filter_f16_m16_kernelILi768ELi4ELb1ELb0ELi0ELi2ELi2ELi8E")
    body = _body(asm, k["mangled"])
    assert body.count("v_mfma_f32_16x16x32_f16") >= 96
    assert body.count("global_load_lds_dwordx4") >= 6 and "s_nop 15" in body
    loop = body[body.index("s_barrier"):]
    assert loop.count("v_accvgpr_read") <= 8, loop.count("v_accvgpr_read")
    # exact MFMA scan, fp16 rows: 192 fp32 MFMAs per tile, every refill load behind a COUNTED wait (one tile of prefetch
    # distance: vmcnt(23) per step, never a drain inside the tile loop)
    k = _find(by_name, "scan_exact_mfma_kernel<2, 768>")
    body = _body(asm, k["mangled"])
    assert body.count("v_mfma_f32_16x16x4_f32") == 192 and body.count("s_waitcnt vmcnt(23)") >= 24, (body.count("v_mfma_f32_16x16x4_f32"), body.count("s_waitcnt vmcnt(23)"))
    k = _find(by_name, "filter_i8s_kernel<768, true, false, 6, 0, 4>")
    body = _body(asm, k["mangled"])
    assert body.count("v_mfma_i32_16x16x64_i8") == 192 and "v_mfma_i32_32x32x32_i8" not in body
    # refine v3: q - half as ONE v_fma_mix_f32 per element (hipcc folds the C++ form back into cvt + sub)
    k = _find(by_name, "refine_l2_rows_kernel<768>")
    body = _body(asm, k["mangled"])
    assert body.count("v_fma_mix_f32") == 192 and body.count("v_cvt_f32_f16") == 0 and body.count("global_load_lds_dwordx4") == 24


def test_exact_image_kernel_keeps_its_prefetch_out_of_the_compilers_hands(codegen):
    """exact_mfma_img_kernel (csrc/kernels_exact_mfma.h): the raw tile lands in LDS slots through direct-to-LDS loads counted by hand.
    The build that kept it in registers lost 18 % to ONE compiler-made `s_waitcnt vmcnt(0)` behind the loads of every tile
    (profiles/r04_exact_img_ablation.txt); this pins what the tile loop may contain."""
    by_name, asm = codegen
    for name, pieces in (("exact_mfma_img_kernel<2, 768, false>", 6), ("exact_mfma_img_kernel<3, 768, false>", 3)):
        k = _find(by_name, name)
        assert k["vgpr"] + k["agpr"] <= 512 and k["occ"] == 1, (name, k)
        start = asm.index(f"\n{k['mangled']}:")
        body = asm[start:asm.index(".Lfunc_end", start)]                                     # (_body stops at the early-exit s_endpgm)
        tag = "BB" + re.search(r"\.LBB(\d+_\d+):\s*; =>This Loop Header: Depth=1", body).group(1)
        # the tile loop = every basic block whose label comment names that header (the loop is laid out rotated: its tail precedes it)
        loop = "\n".join(b for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body) if tag in "\n".join(b.split("\n")[:6]))
        assert loop.count("v_mfma_f32_16x16x4_f32") == 192, (name, loop.count("v_mfma_f32_16x16x4_f32"))
        lpt = pieces + (1 if name.startswith("exact_mfma_img_kernel<3") else 0)               # int8: + the row scales
        assert loop.count("global_load_lds_dwordx4") == pieces and loop.count("s_barrier") == 1, name
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)
        assert waits and all(int(w) == lpt - 1 for w in waits), (name, waits)                 # the hand-counted ones, nothing else
        assert "global_load_dwordx4" not in loop.replace("global_load_lds_dwordx4", ""), name  # no register-bound row loads in the loop

