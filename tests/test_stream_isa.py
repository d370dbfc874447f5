"""Build-time invariants of the streamed RK4 loop's machine code (tools/check_stream_isa.py): runs on CPU - hipcc cross-compiles
gfx950 here, llvm-objdump reads the code object that ships inside libmarl_hip.so.  VERDICT r2 item 6: the loop's shape guards two
compiler accidents; these tests fail if a future hipcc (or an edit) re-introduces either."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm", "--cuda-device-only", "-c"]   # csrc/Makefile's code generation flags
PROBE = os.path.join(ROOT, "tools", "lab_src", "stream_isa_probe.hip")

needs_toolchain = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs the ROCm toolchain (build container / GPU box image)")


@needs_toolchain
def test_every_shipped_stream_kernel_instantiation_keeps_the_invariants():
    import check_stream_isa as chk
    from marlpde_amd import _abi
    res = chk.check_text(chk.disassemble_so(_abi.LIB_PATH))
    # what marl_api.hip's rk4_stream() can launch: depths 1, 2, 4, 8, 16 in both layouts + the dPhi_variable set (depths 1 and 4)
    assert len(res) == 14, sorted(res)
    assert {n: bad for n, bad in res.items() if bad} == {}


@needs_toolchain
def test_the_checker_rejects_the_round2_loop_shape_and_accepts_the_shipped_one(tmp_path):
    import check_stream_isa as chk
    good, broken = str(tmp_path / "good.co"), str(tmp_path / "broken.co")
    subprocess.run([HIPCC, *FLAGS, "-o", good, PROBE], check=True)
    subprocess.run([HIPCC, *FLAGS, "-DMARL_LAB_BROKEN_STREAM_LATCH", "-o", broken, PROBE], check=True)
    ok = chk.check_text(chk.disassemble_so(good))
    assert len(ok) == 2 and all(not bad for bad in ok.values()), ok
    res = chk.check_text(chk.disassemble_so(broken))
    assert len(res) == 2
    for name, bad in res.items():
        # the item grab in the loop latch: hipcc emits the loop-header barrier without `s_waitcnt lgkmcnt(0)` after the ds_write of
        # s_item - waves read the previous item (profiles/r02_lab_rk4_stream.log, 15 - 200 of 200 runs with stale tiles)
        assert any(b.startswith("I1:") for b in bad), (name, bad)


def test_checker_dataflow_on_a_handwritten_listing():
    """The data-flow itself, without a toolchain: a barrier reached over a back-edge with a pending ds_write must be flagged."""
    import check_stream_isa as chk

    def listing(wait_in_latch):
        rows = [("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("v_mov_b32", "v0, v1"), ("ds_write_b32", "v0, v1")]
        if wait_in_latch:
            rows.append(("s_waitcnt", "lgkmcnt(0)"))
        rows += [("s_cbranch_scc1", "x"), ("s_endpgm", "")]
        text = ["0000000000001000 <k_rk4_stream_kernel_test>:"]
        for i, (op, args) in enumerate(rows):
            tgt = " <k_rk4_stream_kernel_test+0x4>" if op.startswith("s_cbranch") else ""   # back-edge to the s_barrier (address base + 4)
            text.append(f"\t{op} {args}    // {0x1000 + 4 * i:012X}: 00000000{tgt}")
        return "\n".join(text)
    bad = chk.check_function(chk.functions(listing(False), "rk4_stream_kernel")["k_rk4_stream_kernel_test"])
    assert any(b.startswith("I1:") for b in bad)
    good = chk.check_function(chk.functions(listing(True), "rk4_stream_kernel")["k_rk4_stream_kernel_test"])
    assert not any(b.startswith("I1:") for b in good)


@needs_toolchain
def test_every_shipped_persistent_rk45_instantiation_keeps_its_invariants():
    """rk45_stream_kernel: the decision record crosses XCDs without a fence, like the streamed RK4 tiles - J1-J5 of the checker."""
    import check_stream_isa as chk
    from marlpde_amd import _abi
    res = chk.check_rk45_text(chk.disassemble_so(_abi.LIB_PATH))
    # both layouts x constant / time-varying porosity diffusion + the one-attempt-per-launch (DD) pair on field-major slabs
    assert len(res) == 6 and sum("Lb1EEEv" in n for n in res) == 2, sorted(res)
    assert {n: bad for n, bad in res.items() if bad} == {}


def test_rk45_checker_flags_a_record_published_before_the_stores_are_acknowledged():
    import check_stream_isa as chk

    def listing(wait, sc1=True):
        rows = [("global_load_dwordx2", "v[0:1], v2, s[0:1]"),                      # prologue: plain is fine
                ("s_barrier", ""),                                                   # loop head (address base + 4)
                ("global_load_dwordx2", "v[0:1], v2, s[0:1] sc1" if sc1 else "v[0:1], v2, s[0:1]"),
                ("global_store_dwordx2", "v2, v[0:1], s[0:1] sc1"),
                ("global_atomic_min_f64", "v2, v[0:1], s[2:3] sc1")]
        if wait:
            rows.append(("s_waitcnt", "vmcnt(0)"))
        rows += [("global_atomic_add", "v3, v2, v4, s[4:5] sc0 sc1"), ("s_waitcnt", "vmcnt(0)" if wait else "lgkmcnt(0)"), ("s_barrier", ""),
                 ("global_store_dwordx4", "v2, v[4:7], s[6:7] sc1"), ("global_atomic_add", "v3, v2, v4, s[8:9] sc0 sc1"),
                 ("s_cbranch_scc1", "x"), ("s_endpgm", "")]
        text = ["0000000000001000 <k_rk45_stream_kernel_test>:"]
        for i, (op, args) in enumerate(rows):
            tgt = " <k_rk45_stream_kernel_test+0x4>" if op.startswith("s_cbranch") else ""
            text.append(f"\t{op} {args}    // {0x1000 + 4 * i:012X}: 00000000{tgt}")
        return chk.functions("\n".join(text), "rk45_stream_kernel")["k_rk45_stream_kernel_test"]
    assert chk.check_rk45_function(listing(True)) == []
    assert any(b.startswith("J3:") for b in chk.check_rk45_function(listing(False)))
    assert any(b.startswith("J2:") for b in chk.check_rk45_function(listing(True, sc1=False)))
