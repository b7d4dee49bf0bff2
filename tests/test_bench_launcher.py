"""`python bench.py --gpus N` must start by itself (a driver that runs it the way it runs `--gpus 1` got SystemExit).
The launcher, the rendezvous, the barrier / max-over-ranks timing and the all-gather run here on the CPU under gloo with bench.py's
stub step (`--stub-step`: a test hook whose JSON line says "stub" in `data`); the GPU step itself is covered by the -m gpu tests."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_gpus_2_without_a_launcher_prints_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--images", "3", "--masks", "7", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert "stub" in j["data"] and j["value"] > 0 and j["ms_per_step"] > 0
    assert j["config"]["workload"].startswith("stub step, 3 x 7 indices per rank (x2 ranks)")


def test_torchrun_style_environment_still_works():
    """The contract's form: the launcher exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; bench.py must not spawn again."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--images", "2", "--masks", "5", "--steps", "1", "--warmup", "0"],
                              env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert len(_json_lines(outs[0][0])) == 1 and _json_lines(outs[1][0]) == []         # rank 0 alone prints
    assert _json_lines(outs[0][0])[0]["n_gpus"] == 2


def test_a_failing_rank_fails_the_launcher_and_stops_the_others():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--images", "2", "--masks", "5"],
                       env=_env(MPX_BENCH_STUB_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and _json_lines(r.stdout) == []
    assert "rank 1 exited with 3" in r.stderr
    assert time.time() - t0 < 120          # rank 0 was terminated instead of waiting out the rendezvous timeout


def test_a_rank_that_ignores_sigterm_is_killed():
    """A rank blocked in a collective or a kernel does not act on SIGTERM: the launcher escalates to SIGKILL after its grace period, returns the
    failing rank's code and leaves no child behind (it used to poll forever)."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--images", "2", "--masks", "5"],
                       env=_env(MPX_BENCH_STUB_FAIL_RANK="1", MPX_BENCH_STUB_DEAF_RANK="0", MPX_BENCH_KILL_GRACE_S="2"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and _json_lines(r.stdout) == []
    assert "rank 1 exited with 3" in r.stderr and "rank 0 ignored SIGTERM" in r.stderr
    assert time.time() - t0 < 60
    import psutil
    alive = [c for c in psutil.Process().children(recursive=True)
             if c.is_running() and c.status() != psutil.STATUS_ZOMBIE and "bench.py" in " ".join(c.cmdline())]
    assert not alive, alive


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
