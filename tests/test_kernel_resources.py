"""Build-time resource invariants of the hot kernels, read from the code object that ships (CPU suite).

The fused integrators sit exactly at their register cap (128 VGPRs: four waves per SIMD), and a spill inside their stage sequence costs
5 - 10 % (every wave of a workgroup waits for the reload in front of its barrier).  Round 4 found out the hard way that an innocent
change in the shared point evaluation (one more wave-uniform value carried across the exchange barrier) tipped rk45_attempt_kernel from
0 to 28 bytes of scratch per lane and rk45_single from 1.64e10 to 1.47e10 grid-point-steps/s - visible in no test, only in the
kernel's resource record.  This test pins those records: scratch (.private_segment_fixed_size), LDS and the register count of the
kernels whose speed depends on them, as the compiler wrote them into the shipped libmarl_hip.so's amdhsa metadata.
"""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from check_stream_isa import DEFAULT_SO, LLVM  # noqa: E402


def kernel_records():
    """mangled kernel name -> {private_segment_fixed_size, group_segment_fixed_size, vgpr_count, sgpr_count, ...} of the main code object"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", DEFAULT_SO], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    recs = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            recs[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|group_segment_fixed_size|vgpr_count|sgpr_count|vgpr_spill_count):\s+(\d+)", blk)}
    return recs


@pytest.fixture(scope="module")
def recs():
    if not os.path.exists(DEFAULT_SO):
        pytest.skip("libmarl_hip.so has not been built")
    r = kernel_records()
    assert len(r) > 100, "could not read the kernels' metadata"
    return r


def pick(recs, *parts):
    hits = {k: v for k, v in recs.items() if all(p in k for p in parts)}
    assert hits, parts
    return hits


def test_fused_integrators_sit_at_four_waves_per_simd_without_scratch_where_it_matters(recs):
    # the launch-per-attempt Dormand-Prince kernel, both layouts (not the dPhi_variable set: a model variant, allowed to spill)
    for name, r in pick(recs, "rk45_attempt_kernelILi256ELi1E", "Lb0EEE").items():
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_count"] <= 128 and r["group_segment_fixed_size"] == 37888, (name, r)
    # the per-level fused RK4 kernels of the headline family (tiled layout) and the one-workgroup RK4 sweep
    for name, r in pick(recs, "rk4_fused_kernelILi256ELi1ELi1ELi", "Lb0EEE").items():
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_count"] <= 128, (name, r)
    for name, r in pick(recs, "rk4_sweep_kernelILi1024ELi1ELb0EEE").items():
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_count"] <= 128, (name, r)


def test_persistent_loops_keep_their_occupancy_and_bounded_scratch(recs):
    # streamed RK4: <= 36 B (four 64-bit values of the item prologue, none inside the stage loop: DESIGN.md 4); dPhi_variable set excluded
    for name, r in pick(recs, "rk4_stream_kernelILi256E", "Lb0EEE").items():
        assert r["private_segment_fixed_size"] <= 36 and r["vgpr_count"] <= 128, (name, r)
    # persistent RK45: LDS must leave room for four workgroups per CU (4 x 40 960 = 160 KB), scratch only outside the six evaluations
    for name, r in pick(recs, "rk45_stream_kernelILi256E", "Lb0ELb0EEE").items():
        assert r["group_segment_fixed_size"] <= 40960 and r["vgpr_count"] <= 128 and r["private_segment_fixed_size"] <= 128, (name, r)
    # the 1024-thread adaptive sweep kernel: capped at 128 VGPRs by its shape; its scratch is on record (184 B), not allowed to grow
    for name, r in pick(recs, "rk45_sweep_kernelILi1024ELi1ELb0EEE").items():
        assert r["private_segment_fixed_size"] <= 184 and r["vgpr_count"] <= 128, (name, r)
