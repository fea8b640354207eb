"""bench.py's N > 1 entry (CPU, no GPU touched): `python bench.py --gpus N` with no launcher in the environment must start
the N ranks itself -- or refuse -- and can never print an `n_gpus: 1` line for `--gpus N` (missing item 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LTXMI_BENCH_REHEARSAL")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{"):
            out.append(json.loads(line))
    return out


def test_bench_starts_its_own_ranks_when_no_launcher_is_present():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert lines == [{"selftest_launch": True, "n_gpus": 2}], (r.stdout, r.stderr[-2000:])


def test_bench_refuses_more_ranks_than_gpus_instead_of_running_one():
    # this container shows no GPU: `--gpus 8` must exit non-zero and print NO JSON line (round 2 printed n_gpus: 1)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert "--gpus 8" in r.stderr


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and _json_lines(r.stdout) == []
    assert "WORLD_SIZE=4" in r.stderr


@pytest.mark.gpu
def test_bench_watchdog_prints_the_replicas_line_when_the_collective_legs_hang():
    """The legs that run collectives (Ulysses, tile-parallel decode) have only ever run at world size 1 on hardware: a hang in
    them must not cost the line its `value`.  With a deadline no leg can meet, the bench still prints ONE line -- the replicas
    measurement with error entries for those legs -- and exits 0; with the default deadline the same command fills them in."""
    cmd = [sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--no-extras", "--rehearse-both"]
    r = subprocess.run(cmd + ["--collective-timeout", "0.05"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "watchdog" in line["ulysses"]["error"] and "watchdog" in line["ulysses_config3"]["error"]
    assert "watchdog" in r.stderr
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_lines(r.stdout)[0]
    assert line["ulysses"]["value"] > 0 and line["ulysses_config3"]["value"] > 0
