"""Static check of the generated gfx950 ISA: the beam kernel issues its LDS gathers and its
metadata loads from inline asm with hand-counted s_waitcnt, which the compiler cannot see.  No
instruction may touch a register that such a load still has in flight (tools/check_inflight.py).
Runs on CPU: hipcc cross-compiles bp.hip to an assembly listing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_inflight  # noqa: E402


def test_checker_flags_copy_of_register_in_flight():
    listing = """
k1:
	ds_read_b64 v[4:5], v2 offset:0
	ds_read_b64 v[6:7], v2 offset:512
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v8, v4, v5
	v_mov_b32_e32 v9, v6
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v9, v7
	s_load_dwordx4 s[8:11], s[0:1], 0x0
	s_mov_b32 s9, 0
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s12, s9
	s_endpgm
"""
    kernels = check_inflight.split_kernels(listing)
    bad = check_inflight.check_kernel(kernels["k1"])
    assert [b[1].split()[0] for b in bad] == ["v_mov_b32_e32", "s_mov_b32"]
    assert bad[0][2] == [6] and bad[1][3] == [9]


def test_checker_follows_branches():
    listing = """
k2:
	ds_read_b32 v1, v0
	s_cbranch_scc1 .LBB0_2
	s_waitcnt lgkmcnt(0)
.LBB0_2:
	v_mov_b32_e32 v2, v1
	s_endpgm
"""
    bad = check_inflight.check_kernel(check_inflight.split_kernels(listing)["k2"])
    assert len(bad) == 1 and bad[0][2] == [1]


def test_checker_tracks_hand_issued_vector_loads():
    """vmcnt: loads return in order; only inline-asm loads have their destinations tracked, the
    compiler's own loads just take a slot of the queue; a loop entered with an empty queue keeps
    the steady-state order."""
    listing = """
k3:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v0, s[2:3] offset:8
	;;#ASMEND
	global_load_dword v9, v1, s[2:3]
	;;#ASMSTART
	global_load_dwordx2 v[10:11], v0, s[2:3] offset:0
	;;#ASMEND
	v_add_u32_e32 v12, v9, v9
	s_waitcnt vmcnt(1)
	v_add_u32_e32 v12, v4, v5
	v_add_u32_e32 v12, v10, v4
	s_waitcnt vmcnt(0)
	v_add_u32_e32 v12, v10, v4
.LBB2_1:
	v_add_u32_e32 v13, v4, v4
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v0, s[2:3] offset:8
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[20:23], v0, s[2:3] offset:24
	;;#ASMEND
	s_waitcnt vmcnt(1)
	s_cbranch_scc1 .LBB2_1
	s_waitcnt vmcnt(0)
	s_endpgm
"""
    bad = check_inflight.check_kernel(check_inflight.split_kernels(listing)["k3"])
    # (1) v10 read under vmcnt(1) while its load is the youngest; (2) in the loop the second load
    # re-targets v[20:23] while the previous trip's load of the same registers is still in flight
    assert [(b[1].split()[0], b[2]) for b in bad] == [("v_add_u32_e32", [10]),
                                                      ("global_load_dwordx4", [20, 21, 22, 23])]


@pytest.mark.timeout(900)
def test_beam_kernels_touch_no_register_in_flight(tmp_path):
    from seismic_bpmf_amd.build import ARCH, CSRC, find_hipcc
    out = tmp_path / "bp.s"
    cmd = [find_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "bp.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    kernels = check_inflight.split_kernels(out.read_text())
    names = [n for n in kernels if "bp_beam_wps2_kernel" in n]
    assert len(names) >= 16
    for n in names:
        bad = check_inflight.check_kernel(kernels[n])
        assert not bad, f"{n}: {bad[:4]}"


@pytest.mark.timeout(900)
def test_fast_beam_kernels_touch_no_register_in_flight(tmp_path):
    """bp_fast.hip: the interior-tile kernels (uniform-weight and per-station records, tiles of
    512 / 256 / 128 samples)."""
    from seismic_bpmf_amd.build import ARCH, CSRC, find_hipcc
    out = tmp_path / "bp_fast.s"
    cmd = [find_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "bp_fast.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    kernels = check_inflight.split_kernels(out.read_text())
    names = [n for n in kernels if "bp_beam_fast_kernel" in n]
    assert len(names) == 8
    for n in names:
        body = kernels[n]
        assert sum(t.startswith("ds_read_b64") for t in body) > 100       # the unrolled gathers are there
        # no spills -- except in the two-residency kernels (..., 4, true>), where the compiler keeps part
        # of the carried partial beams in scratch: a reload per source at the boundary of its records,
        # none between the gathers (the checker counts those loads in the in-order vector-memory queue)
        if "ELi4ELb1E" not in n:
            assert not any("scratch_" in t for t in body)
        bad = check_inflight.check_kernel(body)
        assert not bad, f"{n}: {bad[:4]}"


@pytest.mark.timeout(900)
def test_mf_kernels_touch_no_register_in_flight(tmp_path):
    """mf.hip: the MFMA kernels read their operands with inline-asm ds_read_b32 and counted
    lgkmcnt(5) waits while a compiler-issued scalar load (the channel record two channels ahead)
    may be in flight: the same hazard class as the beam kernels."""
    from seismic_bpmf_amd.build import ARCH, CSRC, find_hipcc
    out = tmp_path / "mf.s"
    cmd = [find_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "mf.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    kernels = check_inflight.split_kernels(out.read_text())
    names = [n for n in kernels if "mf_mfma_wave_kernel" in n or "mf_mfma_kernel" in n]
    assert len(names) >= 8
    for n in names:
        body = kernels[n]
        assert any(t.startswith("v_mfma_f32_16x16x4") for t in body)
        bad = check_inflight.check_kernel(body)
        assert not bad, f"{n}: {bad[:4]}"
    # the headline kernel (network sum, step 1, 4 tiles per wave, default epilogue): 4 waves per SIMD -- at most
    # 128 VGPRs -- and nothing in scratch; 16 MFMAs per trip of its K loop
    import re
    text = out.read_text()
    meta = text[text.index("amdhsa.kernels"):]
    head = [b for b in meta.split("  - .agpr_count")[1:]
            if re.search(r"mf_mfma_wave_kernelILb1ELi20ELi5ELb1ELi4ELb0ELb0E", b)]
    assert len(head) == 1
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", head[0]).group(1)) <= 128
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", head[0]).group(1)) == 0
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", head[0]).group(1)) == 0
    hname = [n for n in names if "mf_mfma_wave_kernelILb1ELi20ELi5ELb1ELi4ELb0ELb0E" in n][0]
    assert sum(t.startswith("v_mfma_f32_16x16x4") for t in kernels[hname]) == 16
    # the small problems' variants (1 / 2 tiles per wave) store their staging registers without jump tables:
    # MAXR + MAXT ds_write_b32 in the channel loop and no scratch either
    for key, n_writes in (("ILb1ELi8ELi5ELb1ELi1ELb0ELb0E", 13), ("ILb1ELi12ELi5ELb1ELi2ELb0ELb0E", 17)):
        kn = [n for n in names if "mf_mfma_wave_kernel" + key in n][0]
        stored = sum(2 if t.startswith("ds_write2") else 1 for t in kernels[kn] if t.startswith("ds_write"))
        assert stored == n_writes, (kn, stored)       # (the compiler pairs some into ds_write2st64_b32)
        assert not any("scratch_" in t for t in kernels[kn])


@pytest.mark.timeout(900)
def test_split_kernel_touches_no_register_in_flight(tmp_path):
    """mf_split.hip (option mf.split16, round 6): operands by inline-asm ds_read2_b32 / ds_read_b128 behind counted
    lgkmcnt(5 / 6 / 7 / 8 / 8) waits, one k-step ahead.  Every instantiation: no register touched while its read is in
    flight, nothing in scratch, at most 256 VGPRs (two waves per SIMD), 12 MFMAs per unrolled pair of k-steps + 6 in
    the odd tail."""
    import re
    from seismic_bpmf_amd.build import ARCH, CSRC, find_hipcc
    out = tmp_path / "mf_split.s"
    cmd = [find_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "mf_split.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = out.read_text()
    kernels = check_inflight.split_kernels(text)
    names = [n for n in kernels if "mf_split_kernel" in n]
    assert len(names) == 8                       # network sum x step 1 x segmented templates
    for n in names:
        body = kernels[n]
        assert sum(t.startswith("v_mfma_f32_32x32x16_f16") for t in body) == 18
        assert sum(t.startswith("ds_read_b128") for t in body) == 12 and sum(t.startswith("ds_read2_b32") for t in body) == 12
        # the network-sum variants (the hot ones) keep everything in registers; the per-channel-output variants (the
        # inter-template CC: tiny problems) spill a few address registers around their scattered stores
        if "mf_split_kernelILb1E" in n:
            assert not any("scratch_" in t for t in body)
        bad = check_inflight.check_kernel(body)
        assert not bad, f"{n}: {bad[:4]}"
    meta = text[text.index("amdhsa.kernels"):]
    for b in meta.split("  - .agpr_count")[1:]:
        if "mf_split_kernel" in b:
            assert int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)) <= 256
            if "mf_split_kernelILb1E" in b:
                assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)) == 0
            else:
                assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)) <= 128
