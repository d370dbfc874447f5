#!/usr/bin/env python3
"""Build-time invariants of rk4_stream_kernel, checked on the gfx950 machine code that ships.

The streamed fixed-step loop (csrc/marl_kernels.h, rk4_stream_kernel) passes tiles between workgroups on different XCDs
without a release / acquire fence (a fence at agent scope writes back the whole L2: 40x slower, measured).  Its correctness
rests on properties of the GENERATED code, two of which the compiler broke in round 2 before the loop got its present shape
(profiles/r02_lab_rk4_stream.log).  This module turns them into checks that fail the build's test-suite
(tests/test_stream_isa.py) if a future hipcc - or an edit of the loop - loses them:

  I1  LDS publication: on EVERY path, no LDS operation (ds_*) is outstanding at an s_barrier - an `s_waitcnt lgkmcnt(0)`
      lies between the last ds_* and the barrier.  (Round 2: with the item grab in the loop latch the barrier at the loop
      header came without it and waves read the previous s_item.)  Checked by a forward data-flow over the kernel's CFG.
  I2  State traffic bypasses the non-coherent L2: the five field loads and the five field stores of a tile
      (global_load_dwordx2 / global_store_dwordx2 in the item loop) carry sc1, as do the done[] / sticky words.
  I3  A tile is published only after its stores are acknowledged: on every path to a `global_store_dword ... sc1` (done[tile],
      sticky) every state store has been followed by `s_waitcnt vmcnt(0)` AND THEN an s_barrier (all waves' stores).
  I4  Barrier census: exactly 9 static s_barrier per instantiation (tables 1, item grab 2, wait 1, stages 4, publish 1) - a
      duplicated barrier is how the jump-threaded private loop of lane 0 showed (hang at NSTEPS = 1).
  I5  No indirect control flow (s_setpc / s_swappc) - the CFG above is complete - and the queue grab is ONE atomic add.

Which guarantees are architectural and which are measured is stated in DESIGN.md ("streamed loop: what is guaranteed").

    python tools/check_stream_isa.py [path/to/libmarl_hip.so | file.s]      exit code 0 = every instantiation passes
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SO = os.path.join(ROOT, "integrating-diagenetic-equations-using-python_amd", "csrc", "libmarl_hip.so")
EXPECTED_BARRIERS = 9
PUBLISH_STORES = 1   # rk45_stream_kernel: one global_store_dwordx4 site (a loop over the replicas of the record)


def disassemble_so(so_path):
    """gfx950 code object of a HIP shared library (or of a `hipcc --cuda-device-only -c` offload bundle) -> llvm-objdump -d text."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        if open(so_path, "rb").read(24).startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
            fat = so_path
        else:
            subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def functions(text, pattern):
    """name -> [(address, opcode, operand text, branch target address or None)] for functions whose name contains `pattern`."""
    out, cur, base = {}, None, 0
    for ln in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:$", ln)
        if m:
            cur = m.group(2) if pattern in m.group(2) else None
            base = int(m.group(1), 16)
            if cur:
                out[cur] = []
            continue
        if cur and ln.startswith("\t"):
            body, _, comment = ln.strip().partition("//")
            am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
            if not am:
                continue
            parts = body.split(None, 1)
            tgt = None
            tm = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>", comment)
            if parts[0].startswith(("s_branch", "s_cbranch")):
                tgt = base + int(tm.group(1), 16) if tm else base
            out[cur].append((int(am.group(1), 16), parts[0], parts[1] if len(parts) > 1 else "", tgt))
    return out


def check_function(ins):
    """Returns the list of violated invariants (strings) of one kernel."""
    bad = []
    index = {a: i for i, (a, *_rest) in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (a, op, args, tgt) in enumerate(ins):
        if op in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            bad.append(f"I5: indirect control flow ({op}) at {a:#x}")
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(index[tgt])
            continue
        if op.startswith("s_cbranch"):
            succ[i].append(index[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)

    # forward data-flow, state bits: L = an LDS operation may be outstanding; V = a state store may be unacknowledged;
    # B = a state store has not yet been followed by (wait vmcnt(0), then s_barrier)
    L, V, B = 1, 2, 4
    state_in = [None] * n
    state_in[0] = 0
    work = [0]
    while work:
        i = work.pop()
        s = state_in[i]
        a, op, args, _ = ins[i]
        if op.startswith("ds_"):
            s |= L
        elif op == "global_store_dwordx2":
            s |= V | B
        elif op == "s_waitcnt":
            if "lgkmcnt(0)" in args:
                s &= ~L
            if "vmcnt(0)" in args:
                s &= ~V
        elif op == "s_barrier":
            if not s & V:
                s &= ~B
        for j in succ[i]:
            merged = s if state_in[j] is None else (state_in[j] | s)
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)

    barriers = atomics = 0
    sc1_loads = sc1_stores = plain_stores = flag_stores = flag_loads = 0
    for i, (a, op, args, _) in enumerate(ins):
        s = state_in[i]
        if s is None:
            continue   # unreachable padding
        if op == "s_barrier":
            barriers += 1
            if s & L:
                bad.append(f"I1: s_barrier at {a:#x} can be reached with an LDS operation outstanding (no s_waitcnt lgkmcnt(0) since the last ds_*)")
        elif op == "global_store_dword" and "sc1" in args:
            flag_stores += 1
            if s & V:
                bad.append(f"I3: flag store at {a:#x} can be reached with state stores unacknowledged (no s_waitcnt vmcnt(0))")
            elif s & B:
                bad.append(f"I3: flag store at {a:#x} can be reached without an s_barrier after the stores were acknowledged")
        elif op == "global_store_dwordx2":
            if "sc1" in args:
                sc1_stores += 1
            else:
                plain_stores += 1
        elif op == "global_load_dwordx2" and "sc1" in args:
            sc1_loads += 1
        elif op == "global_load_dword" and "sc1" in args:
            flag_loads += 1
        elif op.startswith("global_atomic"):
            atomics += 1
    if barriers != EXPECTED_BARRIERS:
        bad.append(f"I4: {barriers} static s_barrier, expected {EXPECTED_BARRIERS}")
    if sc1_stores != 5 or plain_stores:
        bad.append(f"I2: {sc1_stores} sc1 + {plain_stores} plain global_store_dwordx2, expected the five field stores of a tile, all sc1")
    if sc1_loads != 5:
        bad.append(f"I2: {sc1_loads} global_load_dwordx2 with sc1, expected the five field loads of a tile")
    if flag_loads < 3 or flag_stores != 2:
        bad.append(f"I2: {flag_loads} sc1 dword loads / {flag_stores} sc1 dword stores, expected >= 3 (done[t-1..t+1], sticky) / 2 (done[tile], sticky)")
    if atomics != 1:
        bad.append(f"I5: {atomics} global atomics, expected the one queue grab")
    return bad


def check_rk45_function(ins, dd=False):
    """Invariants of rk45_stream_kernel (the persistent Dormand-Prince loop; csrc/marl_kernels.h), on the same data-flow:
      J1  no LDS operation outstanding at any s_barrier (the decision, the grabbed tile index, the 'last workgroup' word travel through LDS);
      J2  every 64-bit / 128-bit global load or store inside a loop (state, per-workgroup sums, monitors, the 16-byte record) carries
          sc1; plain ones only in the prologue, outside every loop (constants and the controller as the host or the previous launch
          left them: a kernel boundary makes those visible).  The one-attempt-per-launch instantiation (DD) also stores its message
          plainly - the launch's end publishes it - and has no record;
      J3  the decision record (global_store_dwordx4 ... sc1) is stored only where every state store / extremum atomic of this workgroup
          has been acknowledged (s_waitcnt vmcnt(0)) and then followed by an s_barrier (its ticket lies in between);
      J4  at least one publish store (none in DD), one ticket / grab atomic add each, extremum atomics present (global_atomic_min_f64);
      J5  no indirect control flow."""
    bad = []
    back_edges = [(tgt, a) for a, op, _args, tgt in ins if tgt is not None and tgt <= a]
    in_loop = lambda a: any(lo <= a <= hi for lo, hi in back_edges)  # noqa: E731
    index = {a: i for i, (a, *_rest) in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (a, op, args, tgt) in enumerate(ins):
        if op in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            bad.append(f"J5: indirect control flow ({op}) at {a:#x}")
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i].append(index[tgt])
            continue
        if op.startswith("s_cbranch"):
            succ[i].append(index[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)
    L, V, B = 1, 2, 4
    state_in = [None] * n
    state_in[0] = 0
    work = [0]
    while work:
        i = work.pop()
        s = state_in[i]
        a, op, args, _ = ins[i]
        if op.startswith("ds_"):
            s |= L
        elif op == "global_store_dwordx2" or op == "global_atomic_min_f64":
            s |= V | B
        elif op == "s_waitcnt":
            if "lgkmcnt(0)" in args:
                s &= ~L
            if "vmcnt(0)" in args:
                s &= ~V
        elif op == "s_barrier":
            if not s & V:
                s &= ~B
        for j in succ[i]:
            merged = s if state_in[j] is None else (state_in[j] | s)
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)
    publish = adds = mins = 0
    for i, (a, op, args, _) in enumerate(ins):
        s = state_in[i]
        if s is None:
            continue
        if op == "s_barrier" and s & L:
            bad.append(f"J1: s_barrier at {a:#x} can be reached with an LDS operation outstanding")
        elif op in ("global_load_dwordx2", "global_store_dwordx2", "global_load_dwordx4", "global_store_dwordx4") and "sc1" not in args:
            if op.startswith("global_load") and not in_loop(a):
                pass
            elif op == "global_store_dwordx2" and dd:
                pass
            else:
                bad.append(f"J2: {op} at {a:#x} without sc1")
        if op == "global_store_dwordx4":
            publish += 1
            if s & V:
                bad.append(f"J3: decision record stored at {a:#x} with state stores / extremum atomics possibly unacknowledged")
            elif s & B:
                bad.append(f"J3: decision record stored at {a:#x} without an s_barrier after the acknowledgement of the workgroup's stores")
        elif op == "global_atomic_add":
            adds += 1
        elif op == "global_atomic_min_f64":
            mins += 1
    if publish != (0 if dd else PUBLISH_STORES) or adds < 2 or mins < 1:
        bad.append(f"J4: {publish} publish stores, {adds} atomic adds (ticket, tile grab), {mins} fp64 atomic minima")
    return bad


def check_text(text, pattern="rk4_stream_kernel"):
    fs = functions(text, pattern)
    return {name: check_function(ins) for name, ins in fs.items()}


def check_rk45_text(text):
    """name -> violations for every rk45_stream_kernel instantiation (the last template argument, Lb1E before the parameter list, is DD)."""
    fs = functions(text, "rk45_stream_kernel")
    return {name: check_rk45_function(ins, dd="Lb1EEEv" in name) for name, ins in fs.items()}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SO
    text = open(src).read() if src.endswith(".s") else disassemble_so(src)
    res = check_text(text)
    if not res:
        print("no rk4_stream_kernel instantiation found")
        return 2
    res.update(check_rk45_text(text))
    rc = 0
    for name, bad in res.items():
        print(("FAIL " if bad else "ok   ") + name)
        for b in bad:
            print("     " + b)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
