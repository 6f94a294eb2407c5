"""`python bench.py --gpus N` must really run N ranks: without a torch.distributed environment bench.py spawns them
itself (a child torch.distributed.run), rank 0 prints ONE JSON line with n_gpus = N and the world size the process
group saw.  MSLAM_BENCH_DRY=1 runs the launcher, the rendezvous (gloo) and the reductions without any GPU work, which
is what can be checked on a CPU-only host; on the GPU box the same code path drives the real step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(MSLAM_BENCH_DRY="1", **kw)
    return e


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dist"]["world_size"] == 2 and j["dist"]["launched_by"] == "bench.py"
    assert j["steps"] == 3 and j["warmup"] == 1
    assert j["t_max"] == 2.0 and j["units"] == 200.0          # MAX over ranks of (1 + rank), SUM of 100 per rank
    # the loop-candidate exchange ran over the process group: ONE collective per batch, as the first real RCCL run must show
    assert j["exchange"]["world_size"] == 2 and j["exchange"]["collectives_per_batch"] == 1.0
    assert j["exchange"]["bytes_per_rank_per_batch"] == 4 * 8 * (2 * 2048 + 1)
    # the cfg4 leg that the real run adds at world > 1 (every rank its own stream, the exchange inside the step, cross scores
    # checked against the oracle on the vectors as transmitted): rehearsed on host tensors over the same process group
    assert j["cfg4"]["n_gpus"] == 2 and j["cfg4"]["oracle_check"] is True
    assert j["cfg4"]["exchange"]["world_size"] == 2 and j["cfg4"]["exchange"]["collectives_per_step"] == 1.0


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["dist"]["world_size"] == 1


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
