"""bench.py's multi-process path (N > 1): rendezvous, barrier, max-over-ranks timing and the whole-job
aggregate -- world_size 2 over gloo on CPU, with the stub engine of `--dry-run` (no GPU here)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd):
    env = dict(os.environ, WZ_BENCH_VERBOSE="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


def test_single_process_dry_run():
    out = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "40", "--warmup", "4", "--dry-run"])
    assert out["n_gpus"] == 1 and out["steps"] == 40 and out["scaling"] == "weak"
    assert 1500 < out["value"] < 8100                 # 8 frames per 2 ms step, two stub lanes in flight


def test_two_ranks_over_gloo():
    port = 29500 + os.getpid() % 400
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "40",
               "--warmup", "4", "--dry-run"])
    assert out["n_gpus"] == 2
    assert out["camera_seeds"] == [1234, 2234]        # every rank serves its own camera (cameras are the shard)
    # the host-side legs run on EVERY rank at the same time when N > 1 and come back per rank and summed (stubbed in a dry run)
    leg = out["legs_all_ranks_concurrently"]["host_frames_pinned_b8"]
    assert leg["per_rank"] == [1000.0, 1001.0] and leg["sum"] == 2001.0 and leg["min"] == 1000.0
    single = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "40", "--warmup", "4", "--dry-run"])
    # whole-job aggregate = frames of all ranks / max over ranks of each rank's OWN clock (stopped after its device synchronize,
    # before the gloo barrier): two replicas are worth two, within 10 %
    # (1.6 .. 2.4 rather than 1.8 .. 2.2: the stub steps are 2 ms sleeps, and on a host that is busy otherwise -- the build container while a snapshot is
    #  being packed -- one of the two runs stretches; a clock that included the barrier or counted one rank only would read 1.0 or 4.0)
    assert 1.6 < out["value"] / single["value"] < 2.4


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py starts one process per device ordinal (the reference's
    one-detector-process-per-device scheme, watsor/detection/detector.py:34-50) and still prints ONE line, n_gpus = 2."""
    env_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        out = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "4", "--dry-run"])
    finally:
        os.environ.update(saved)
    assert out["n_gpus"] == 2 and out["camera_seeds"] == [1234, 2234] and out["rounds"] >= 3
    assert 3000 < out["value"] < 16200                # two replicas of the single-process figure


def test_eight_ranks_dry_run():
    """The shape the driver's scaling run has on an 8-GPU node: `--gpus 8` without a launcher -> eight rank processes, one line,
    every rank with its own camera, the host-side legs gathered from all eight.  (Timing is not asserted: eight stub ranks share
    this container's cores.)"""
    env_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        out = run([sys.executable, "bench.py", "--gpus", "8", "--steps", "10", "--warmup", "2", "--dry-run"])
    finally:
        os.environ.update(saved)
    assert out["n_gpus"] == 8 and out["camera_seeds"] == [1234 + 1000 * r for r in range(8)]
    leg = out["legs_all_ranks_concurrently"]["host_frames_pinned_b8"]
    assert leg["per_rank"] == [1000.0 + r for r in range(8)] and leg["sum"] == sum(1000.0 + r for r in range(8))
    assert out["value"] > 0 and out["scaling"] == "weak"


def test_a_rank_that_dies_stops_the_run_at_once():
    """`spawn_ranks` watches its children: a rank that exits non-zero before the rendezvous (no such device, engine creation failed)
    must end the run with an error within seconds -- not leave the other ranks in gloo's rendezvous / barrier until ITS timeout."""
    import time
    env_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")
    env = {k: v for k, v in os.environ.items() if k not in env_keys}
    env.update(WZ_BENCH_VERBOSE="0", MASTER_ADDR="127.0.0.1")
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--steps", "10", "--warmup", "2", "--dry-run", "--fail-rank", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=100)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert "rank 2 exited with code 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]      # no line is better than a line from a broken job
    assert time.time() - t0 < 60
