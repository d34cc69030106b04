"""`python bench.py --gpus N` without a launcher's environment starts its own N ranks (no GPU needed: the ranks run in stub
mode and only report how they were started).  The contract: children, never an exec; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
as torch.distributed.run would set them; one launch nonce shared by all ranks; rank 0's line relayed; worst status returned."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(n, extra_env=None, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "RANDOMFIELD_LAUNCH_NONCE")}
    env.update({"RANDOMFIELD_BENCH_STUB_CHILD": "1", "RANDOMFIELD_COLLECTIVE_TIMEOUT": "1"})
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "7", "--warmup", "2"], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_starts_its_own_ranks_and_relays_rank_zero():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert len(lines) == 1, "exactly rank 0's line is relayed on stdout: %r" % r.stdout       # (the other ranks' stdout goes to stderr)
    assert '"rank": 1' in r.stderr
    line = lines[0]
    assert line["stub"] and line["rank"] == 0 and line["local_rank"] == 0 and line["world"] == 2
    assert line["gpus"] == 2 and line["steps"] == 7 and line["warmup"] == 2                    # the command line travels unchanged
    assert line["master"][0] == "127.0.0.1" and 1024 < int(line["master"][1]) < 65536
    assert line["nonce"].startswith("bench-") and line["ipc_legacy"] == "0"


def test_every_rank_gets_its_own_rank_and_the_same_nonce(tmp_path):
    """All N ranks start with their own RANK / LOCAL_RANK, one MASTER_PORT and one nonce; the worst exit status is returned."""
    r = _run(3, {"RANDOMFIELD_BENCH_STUB_RC_RANK1": "41", "RANDOMFIELD_BENCH_STUB_RC_RANK2": "42"})
    assert r.returncode == 42, (r.returncode, r.stderr[-500:])                                  # the worst status wins
    assert "[0, 41, 42]" in r.stderr
    zero = json.loads(r.stdout.strip().splitlines()[-1])
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert zero["world"] == 3 and sorted(o["rank"] for o in others) == [1, 2] and all(o["local_rank"] == o["rank"] for o in others)
    assert {o["nonce"] for o in others} == {zero["nonce"]} and {tuple(o["master"]) for o in others} == {tuple(zero["master"])}


def test_under_a_launcher_the_environment_wins():
    """With WORLD_SIZE in the environment (torch.distributed.run) bench.py is a rank, not a launcher: it starts nothing."""
    env = dict(os.environ, RANDOMFIELD_BENCH_STUB_CHILD="1", RANK="1", LOCAL_RANK="1", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29555")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=60, env=env, cwd=ROOT)
    assert r.returncode == 0
    line = json.loads(r.stdout.strip())
    assert line["rank"] == 1 and line["world"] == 4 and line["master"] == ["127.0.0.1", "29555"] and line["nonce"] is None


def test_the_launcher_never_touches_hip():
    """The parent must not load the HIP runtime (a forked / spawned rank of a process that has initialised the GPU is what takes
    GPU boxes down): the launcher path runs before `randomfield_amd._hip` is imported and never imports it."""
    src = open(BENCH).read()
    launch = src[src.index("def launch_ranks"):src.index("def main():")]
    assert "import randomfield_amd" not in launch and "from randomfield_amd" not in launch and "os.exec" not in launch
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args.gpus)") < main.index("from randomfield_amd import _hip")
