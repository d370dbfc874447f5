"""Timeline of the persistent RK45 loop (lab build with -DMARL_LAB_CLOCK45): per attempt, when the workgroups finish their tiles,
take their tickets and see the decision.  MARL_HIP_LIBRARY=<lab .so> python3 tools/rk45_stream_clock.py [N]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dataclasses import asdict
from marlpde_amd import _abi
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
from marlpde_amd.parameters import Map_Scenario

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p = asdict(Map_Scenario()) | {"N": N}
eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
L = p["max_depth"] / p["Xstar"]
x = (np.arange(N) + 0.5) * (L / N)
y = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) * (1.0 + 0.01 * np.sin(2 * np.pi * 8 * x / L))
yd = torch.from_numpy(y.ravel()).cuda()
buf = torch.zeros(eq.state_doubles(1), dtype=torch.float64, device="cuda")
eq.convert_layout_device(yd.data_ptr(), buf.data_ptr(), 0, 1)
dx2 = (L / N) ** 2
for n in (200, 64):   # warm run, then the measured one (stamps of the first 64 attempts of the LAST launch survive)
    b = buf.clone()
    r = eq.integrate_rk45_device(b.data_ptr(), (0.0, 1e9), 0.5 * dx2, 1e-3, 1e-3, 1, max_attempts=n)
lib = _abi.load()
lib.marl_lab_read_clock45.argtypes = [C.c_void_p, C.c_size_t]
clk = np.zeros(64 * 2048 * 4 + 64 * 8, dtype=np.uint64)
assert lib.marl_lab_read_clock45(clk.ctypes.data, clk.size) == 0
last = clk[64 * 2048 * 4:].reshape(64, 8).astype(np.int64) * 0.01
clk = clk[:64 * 2048 * 4].reshape(64, 2048, 4).astype(np.int64)
G = int((clk[1, :, 0] > 0).sum())
print("N", N, "G", G, "accepted", r.n_accepted, "rejected", r.n_rejected)
c = clk[:, :G, :] * 0.01   # us
for a in range(8, 20):
    t0 = c[a, :, 0].min()
    s = c[a] - t0
    print(f"attempt {a}: start {s[:,0].min():6.2f}..{s[:,0].max():6.2f}  tiles done {np.percentile(s[:,1],5):6.2f} / {np.median(s[:,1]):6.2f} / {s[:,1].max():6.2f}"
          f"  ticket {s[:,2].max():6.2f}  seen {s[:,3].min():6.2f}..{s[:,3].max():6.2f}  next start {c[a+1,:,0].min()-t0:6.2f}")
per = np.diff(c[8:60, :, 0].min(axis=1))
print("attempt period us: median %.2f  min %.2f max %.2f" % (np.median(per), per.min(), per.max()))
tiles_span = (c[8:60, :, 1].max(axis=1) - c[8:60, :, 0].min(axis=1))
print("first start -> last tiles done: median %.2f" % np.median(tiles_span))
print("last tiles done -> first seen: median %.2f" % np.median(c[8:60, :, 3].min(axis=1) - c[8:60, :, 1].max(axis=1)))
print("per-workgroup tile phase (done - start): median %.2f  p5 %.2f p95 %.2f" % tuple(np.percentile((c[8:60, :, 1] - c[8:60, :, 0]).ravel(), [50, 5, 95])))

for a in range(8, 14):
    t_done = c[a, :, 1].max()
    print(f"attempt {a}: last tiles done +0  last ticket {c[a,:,2].max()-t_done:5.2f}  last-arriver: enter {last[a,0]-t_done:5.2f} loaded {last[a,1]-t_done:5.2f} reduced {last[a,2]-t_done:5.2f} decided {last[a,3]-t_done:5.2f} published {last[a,4]-t_done:5.2f}  first seen {c[a,:,3].min()-t_done:5.2f} last seen {c[a,:,3].max()-t_done:5.2f}")

lib.marl_lab_read_clock45t.argtypes = [C.c_void_p, C.c_size_t]
ct = np.zeros(2048 * 8 * 8, dtype=np.uint64)
assert lib.marl_lab_read_clock45t(ct.ctypes.data, ct.size) == 0
ck = ct[-4:].astype(np.int64)
print("shader clock over attempts 8..56: %.1f MHz" % ((ck[2] - ck[0]) / ((ck[3] - ck[1]) * 0.01)))
ct = ct.reshape(2048, 8, 8).astype(np.int64)[:G] * 0.01
for ti in range(5):
    m = ct[:, ti, 0] > 0
    if m.sum() == 0:
        continue
    d = ct[m, ti, :]
    print(f"tile {ti} ({m.sum()} workgroups): barrier+loads {np.median(d[:,1]-d[:,0]):5.2f}  six evaluations {np.median(d[:,2]-d[:,1]):5.2f}  err+prefetch+stores {np.median(d[:,3]-d[:,2]):5.2f}  reduce {np.median(d[:,4]-d[:,3]):5.2f}  total {np.median(d[:,4]-d[:,0]):5.2f} us")
