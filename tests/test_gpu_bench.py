"""bench.py end to end on the test GPU: the contract line, the N > 1 code path minus the peers, the radius mode.

The driver runs ``bench.py --gpus N`` on a whole node at round end; nothing but these tests has executed that path's
code before then: an nccl process group, the communicator id broadcast, the C-ABI rank handle (AbiShardEngine), the
exchange forced at world size 1 (PN_BENCH_EXCHANGE=1), the parity leg and the JSON line with ``rccl_world_size``,
``shard_ms`` and ``exchange_ms``.  bench.py is started as a child process (a process group is global state)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, env_extra=None, timeout=600):
    r = None
    for port in (29561, 29567, 29573):  # a busy port only costs a retry
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(env_extra or {})
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                           text=True, timeout=timeout, cwd=ROOT)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_contract_line_on_one_gpu():
    d = _bench(["--config", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["verified"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_ms_per_step"] <= d["ms_per_step"]  # the kernel's hipEvent time fits inside the step


def test_distributed_path_at_world_size_one():
    """The N > 1 branch of bench.py minus the peers: nccl group, id broadcast, rank handle, forced exchange."""
    d = _bench(["--config", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
               {"PN_BENCH_DIST": "1", "PN_BENCH_EXCHANGE": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert d["verified"] is True and d["config"]["rccl_world_size"] == 1 and d["n_gpus"] == 1
    assert d["shard_ms"] > 0.0 and d["exchange_ms"] > 0.0
    assert d["shard_ms"] + d["exchange_ms"] <= d["ms_per_step"] * 1.05


def test_radius_mode():
    d = _bench(["--config", "tiny", "--mode", "radius", "--radius", "nn", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline"])
    assert d["verified"] is True, d["verify"]
    assert 0.2 < d["results_per_query"] < 3.0  # the median nearest-neighbour distance: about half the lists non-empty
    assert "radius" in d["metric"] and d["roofline"]["kernel_ms_per_step"] > 0.0
    e = _bench(["--config", "tiny", "--mode", "radius", "--radius", "0.5", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline"])
    assert e["verified"] is True and e["results_per_query"] == 0.0  # BASELINE configs[2]'s radius: nothing that close
