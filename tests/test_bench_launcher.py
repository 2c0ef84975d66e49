"""bench.py's launcher logic: `python bench.py --gpus N` must be a complete command - under a launcher it is a rank, without
one it re-runs itself under torch.distributed.run, and it refuses (rc 2, a message) instead of asserting when the RCCL backend cannot have
one device per rank.  CPU: the decision function.  GPU (tests/test_gpu_dist.py): the spawned 2-rank gloo job prints one parsable line."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_single_gpu_runs_in_process():
    b = _bench()
    assert b.launch_plan(1, {}, 1, ["--gpus", "1"]) == ("run", None)
    assert b.launch_plan(1, {}, 0, []) == ("run", None)                       # (the run itself then fails loudly: no device)


def test_under_a_launcher_it_is_a_rank():
    b = _bench()
    assert b.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, 8, ["--gpus", "8"]) == ("run", None)
    what, msg = b.launch_plan(8, {"WORLD_SIZE": "4"}, 8, ["--gpus", "8"])
    assert what == "refuse" and "WORLD_SIZE=4" in msg
    what, msg = b.launch_plan(8, {"WORLD_SIZE": "8"}, 1, ["--gpus", "8"])
    assert what == "refuse" and "8 visible devices" in msg
    assert b.launch_plan(8, {"WORLD_SIZE": "8", "NERFART_BENCH_BACKEND": "gloo"}, 1, ["--gpus", "8"]) == ("run", None)


def test_without_a_launcher_it_spawns_one_process_per_gpu():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    what, cmd = b.launch_plan(8, {}, 8, argv, port=29517)
    assert what == "spawn"
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-len(argv) - 1] == os.path.join(REPO, "bench.py") and cmd[-len(argv):] == argv
    what, cmd = b.launch_plan(2, {}, 2, ["--gpus", "2"])                       # a free port is picked when none is given
    assert what == "spawn" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536


def test_fewer_devices_than_ranks_is_refused_not_asserted():
    b = _bench()
    what, msg = b.launch_plan(8, {}, 1, ["--gpus", "8"])
    assert what == "refuse" and "8 visible devices" in msg and "shows 1" in msg
    what, cmd = b.launch_plan(2, {"NERFART_BENCH_BACKEND": "gloo"}, 1, ["--gpus", "2"])
    assert what == "spawn"                                                      # the functional gloo run shares devices
