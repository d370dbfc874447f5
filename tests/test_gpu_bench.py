"""bench.py as the driver runs it, rehearsed on the one-GPU box: the N > 1 path (two rank processes started by bench.py
itself, gloo collectives, both ranks on cuda:0) must print ONE JSON line with n_gpus = 2 and the multi-rank extras - so
that a mistake in the rank launch, the sharding or the domain-decomposed driver fails HERE and not on the 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p, lines


def test_bench_two_ranks_rehearsal_on_one_gpu():
    p, lines = _run(["--gpus", "2", "--steps", "8", "--warmup", "2"],
                    {"MARL_BENCH_BACKEND": "gloo", "MARL_BENCH_ONE_DEVICE": "1", "MARL_BENCH_EXTRAS_DEADLINE": "420", "MARL_BENCH_TIMEOUT_RC": "7"}, 900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines                 # exactly one line on stdout, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 8 and j["warmup"] == 2 and j["scaling"] == "weak" and j["extras_timed_out"] is False
    assert j["value"] > 1e9 and j["roofline"]["bound"] == "fp64_valu" and j["cpu_baseline"]["kind"] == "port"
    x = j["extra"]
    assert not [k for k in x if k.endswith("_error")], {k: v for k, v in x.items() if k.endswith("_error")}
    sw = x["BASELINE_configs2_3_sweep_rk45"]
    assert sw["instances_total"] == 8192 and sw["n_ranks"] == 2 and sw["value"] > 1e9
    dd = x["BASELINE_configs4_dd_rk45"]
    assert dd["n_ranks"] == 2 and dd["N"] == 1 << 22 and "all_gather" in dd["transport"] and dd["accepted_steps"] + dd["rejected_steps"] == 500
    assert x["rk4_N1048576_no_reuse"]["n_ranks"] == 2 and j["roofline"]["frac_reuse_off"] > 0


def test_bench_five_ranks_rehearsal_on_one_gpu():
    """The rank launch, port, deadline, sharding and gather logic at a rank count near the node's 8: FIVE ranks on cuda:0 (a gpurun box
    admits six GPU processes at once, and this test process is one of them), gloo collectives, reduced extras
    (MARL_BENCH_LIGHT_EXTRAS).  Five does not divide the grid or the sweep: uneven slabs, uneven halo owners."""
    p, lines = _run(["--gpus", "5", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                    {"MARL_BENCH_BACKEND": "gloo", "MARL_BENCH_ONE_DEVICE": "1", "MARL_BENCH_EXTRAS_DEADLINE": "600", "MARL_BENCH_LIGHT_EXTRAS": "1"}, 1000)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 5 and j["scaling"] == "weak" and j["extras_timed_out"] is False and j["value"] > 1e9
    x = j["extra"]
    assert not [k for k in x if k.endswith("_error")], {k: v for k, v in x.items() if k.endswith("_error")}
    assert x["BASELINE_configs2_3_sweep_rk45"]["instances_total"] == 5 * 256 and x["BASELINE_configs2_3_sweep_rk45"]["n_ranks"] == 5
    dd = x["BASELINE_configs4_dd_rk45"]
    assert dd["n_ranks"] == 5 and "all_gather" in dd["transport"] and dd["accepted_steps"] + dd["rejected_steps"] == 60


def test_bench_watchdog_marks_the_line_and_can_exit_nonzero():
    """A deadline of 0 s fires while the first extra runs: the headline line must come out once, marked at top level, and the
    exit code is the one MARL_BENCH_TIMEOUT_RC asks for (0 by default: the headline was measured before the extras started)."""
    p, lines = _run(["--steps", "8", "--warmup", "2", "--no-cpu-baseline"], {"MARL_BENCH_EXTRAS_DEADLINE": "0.001", "MARL_BENCH_TIMEOUT_RC": "7"}, 600)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["extras_timed_out"] is True and j["n_gpus"] == 1 and j["value"] > 1e9


def test_slab_comm_init_rejects_a_bad_id_without_entering_rccl():
    """world = 2 with an id ncclGetUniqueId cannot have made: an error at once, not a rank waiting in ncclCommInitRank."""
    import ctypes as C
    import time
    from dataclasses import asdict
    from marlpde_amd import _abi
    from marlpde_amd.domain import HipSlabEngine
    from marlpde_amd.parameters import Map_Scenario
    e = HipSlabEngine(asdict(Map_Scenario()), 4096, 0, 2048, 0)
    t0 = time.time()
    with pytest.raises(_abi.MarlError, match="all zero"):
        e.comm_init(b"\x00" * 128, 0, 2)
    with pytest.raises(_abi.MarlError, match="unique id is required"):
        e.comm_init(None, 0, 2)
    with pytest.raises(_abi.MarlError, match="invalid rank"):
        e.comm_init(b"\x01" * 128, 2, 2)
    assert time.time() - t0 < 5.0
    e.comm_probe()            # the box has PyTorch's librccl: loadable, entry points present
    e.close()
