"""Host logic of bench.py's data-parallel supervisor (supervise(): every launcher-started rank runs the measurement in a worker child and
retries a failed attempt in a more conservative exchange mode).  No GPU: the worker is replaced by a stub that plays scripted outcomes."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _run(monkeypatch, capsys, script, rank=0, dp_safe=False):
    """script: attempt index -> (returncode, finished, stdout).  Returns (status, relayed stdout lines, attempts the stub saw)."""
    seen = []

    def fake_run(cmd, env=None, stdout=None, text=None):
        att = json.loads(env["UNIVL_BENCH_ATTEMPT"])
        seen.append((att["index"], [a for a in cmd if a.startswith("--dp-safe") or a == "--no-graph"], env["MASTER_PORT"], list(att["earlier"])))
        rc, finished, out = script[att["index"]]
        if finished:
            open(env["UNIVL_BENCH_DONE_FILE"], "w").close()
        return types.SimpleNamespace(returncode=rc, stdout=out if stdout is not None else None)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setenv("RANK", str(rank))
    monkeypatch.setenv("MASTER_PORT", "29700")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    status = bench.supervise(argparse.Namespace(dp_safe=dp_safe))
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    return status, out, seen


GOOD = json.dumps(dict(metric="m", value=1234.5, n_gpus=8))
WATCHDOG = json.dumps(dict(metric="m", value=None, error="watchdog", phase="x"))


def test_first_attempt_succeeds_one_line(monkeypatch, capsys):
    status, out, seen = _run(monkeypatch, capsys, {0: (0, True, "RCCL banner\n" + GOOD + "\n")})
    assert status == 0 and out == [GOOD] and [s[0] for s in seen] == [0]
    assert seen[0][1] == [] and seen[0][2] == "29700"


def test_crash_then_hang_then_eager(monkeypatch, capsys):
    script = {0: (139, False, ""), 1: (3, False, WATCHDOG + "\n"), 2: (0, True, GOOD + "\n")}
    status, out, seen = _run(monkeypatch, capsys, script)
    assert status == 0 and out == [GOOD]                      # exactly one line, the one that carries a value
    assert [s[0] for s in seen] == [0, 1, 2]
    assert seen[1][1] == ["--dp-safe"] and seen[2][1] == ["--dp-safe", "--no-graph"]
    assert [s[2] for s in seen] == ["29700", "29717", "29734"]          # a fresh rendezvous port per attempt
    assert [e["returncode"] for e in seen[2][3]] == [139, 3]            # the worker's line reports what happened before


def test_every_attempt_fails_relays_the_last_error_line(monkeypatch, capsys):
    script = {0: (139, False, ""), 1: (3, False, WATCHDOG + "\n"), 2: (3, False, WATCHDOG + "\n")}
    status, out, _ = _run(monkeypatch, capsys, script)
    assert status == 3 and out == [WATCHDOG]


def test_teardown_failure_after_the_last_barrier_is_not_a_retry(monkeypatch, capsys):
    status, out, seen = _run(monkeypatch, capsys, {0: (1, True, GOOD + "\n")})       # non-zero status AFTER every rank measured
    assert status == 0 and out == [GOOD] and len(seen) == 1
    status, out, seen = _run(monkeypatch, capsys, {0: (1, True, None)}, rank=3)          # other ranks print nothing and do not retry either
    assert status == 0 and out == [] and len(seen) == 1


def test_dp_safe_starts_at_the_conservative_mode(monkeypatch, capsys):
    status, out, seen = _run(monkeypatch, capsys, {1: (0, True, GOOD + "\n")}, dp_safe=True)
    assert status == 0 and [s[0] for s in seen] == [1]
