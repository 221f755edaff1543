"""bench.py --gpus N from a bare shell: the ranks are started as a child process (torch.distributed.run), the arguments
are forwarded, the child's line and exit code come back, and an attempt that ends without a line is followed by the plain
protocol.  No GPU needed: the child is replaced by a stub that prints what it was given."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = r"""
import json, os, sys
argv = sys.argv[1:]
mode = os.environ.get("STUB_MODE", "ok")
if mode == "fail_default" and "--host-offsets" not in argv:
    sys.exit(7)
if mode == "hang_default" and "--host-offsets" not in argv:
    import time
    time.sleep(600)
if mode == "fail_always":
    sys.exit(9)
print("some chatter before the line")
print(json.dumps({"metric": "stub", "argv": argv, "n_gpus": int(argv[argv.index("--gpus") + 1])}))
if mode == "line_then_fail":
    sys.exit(5)
"""


def _args(**kw):
    d = dict(gpus=2, launch_timeout=30, no_launch_fallback=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture
def stub(tmp_path, monkeypatch):
    path = tmp_path / "stub.py"
    path.write_text(STUB)
    seen = []
    genuine = bench.launcher_command

    def fake(n, argv, port):
        real = genuine(n, argv, port)
        seen.append(real)
        return [sys.executable, str(path)] + list(argv)
    monkeypatch.setattr(bench, "launcher_command", fake)
    monkeypatch.setattr(bench, "LAUNCH_LADDER", list(bench.LAUNCH_LADDER))
    return seen


def _lines(capfd):
    out = capfd.readouterr().out
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_launcher_command_is_the_contract_line():
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_arguments_are_forwarded_and_the_line_comes_back(stub, capfd, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "ok")
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "sphere", "--size", "512"]
    assert bench.self_launch(_args(), argv) == 0
    (line,) = _lines(capfd)
    assert line["argv"] == argv and line["n_gpus"] == 2
    assert len(stub) == 1 and stub[0][stub[0].index("--nproc-per-node") + 1] == "2"


def test_a_failed_default_step_is_followed_by_the_plain_protocol(stub, capfd, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "fail_default")
    assert bench.self_launch(_args(), ["--gpus", "2", "--steps", "3"]) == 0
    (line,) = _lines(capfd)
    a = line["argv"]
    assert a[:4] == ["--gpus", "2", "--steps", "3"]
    assert "--host-offsets" in a and "--full-halo" in a and a[a.index("--partition") + 1] == "uniform"
    assert "exit code 7" in a[a.index("--fallback-note") + 1]
    assert len(stub) == 2 and stub[0][stub[0].index("--master-port") + 1] != "0"


def test_a_hung_default_step_is_ended_and_followed_by_the_plain_protocol(stub, capfd, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "hang_default")
    assert bench.self_launch(_args(launch_timeout=2), ["--gpus", "2"]) == 0
    (line,) = _lines(capfd)
    assert "no end within 2 s" in line["argv"][line["argv"].index("--fallback-note") + 1]


def test_exit_code_comes_back_when_nothing_works(stub, capfd, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "fail_always")
    assert bench.self_launch(_args(), ["--gpus", "4"]) == 9
    assert _lines(capfd) == [] and len(stub) == 2
    stub.clear()
    assert bench.self_launch(_args(no_launch_fallback=True), ["--gpus", "4"]) == 9
    assert len(stub) == 1


def test_a_line_followed_by_a_failure_is_not_retried(stub, capfd, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "line_then_fail")
    assert bench.self_launch(_args(), ["--gpus", "2"]) == 5
    assert len(_lines(capfd)) == 1 and len(stub) == 1


def test_bare_shell_run_without_gpus_ends_with_a_message_not_a_hang():
    """The real thing in this container (no GPU): `python bench.py --gpus 2` starts its own ranks through
    torch.distributed.run; they cannot find a device, so the run must END -- non-zero, saying so -- within the guard."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "150", "--no-launch-fallback"], env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode != 0
    assert "starting 2 ranks" in p.stderr and "torch.distributed.run" in p.stderr
    assert "ended without a result" in p.stderr


def test_watchdog_ends_a_stretch_that_does_not_finish():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(1, 'a test stretch'):\n    time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "a test stretch did not finish within 1 s" in p.stderr
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(5, 'short'):\n    pass\nprint('through')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "through" in p.stdout
