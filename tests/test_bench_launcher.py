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


def test_the_multi_gpu_line_explains_itself():
    """at N > 1 the line carries `strong` (one frame over N GPUs: north_star's ray-parallel scaling) and `weak` at top level
    whichever is primary, per-rank step times, and an explicit cpu_baseline marker with the cached N = 1 figure; at N = 1 nothing is added."""
    b = _bench()
    assert b.scale_fields(1, False, 250000.0, 518.0, {}, [518.0], None, None) == {}
    n1 = {"value": 250000.0, "n_gpus": 1, "cpu_baseline": {"value": 119.4, "unit": "rays/s", "cores": 32, "kind": "port", "sample": "2048 rays"}}
    sec = {"strong_tiles": {"value": 1.9e6, "unit": "rays/s", "ms_per_step": 68.2, "scaling": "strong"}}
    f = b.scale_fields(8, False, 1.99e6, 521.0, sec, [520.0, 515.0, 519.0, 521.0, 500.0, 510.0, 505.0, 518.0], "profiles/r08_bench_line.json", n1)
    assert f["weak"]["is_primary"] and not f["strong"]["is_primary"]
    assert f["strong"]["value"] == 1.9e6 and f["strong"]["ms_per_step"] == 68.2 and f["weak"]["value"] == 1.99e6
    assert f["strong"]["speedup_vs_cached_n1"] == 7.6 and f["strong"]["efficiency_vs_n1_hint"] == 0.95
    assert f["rank_step_ms"]["max"] == 521.0 and f["rank_step_ms"]["min"] == 500.0 and f["rank_step_ms"]["imbalance_max_over_mean"] > 1.0
    assert f["cpu_baseline"]["n/a at N>1"] and f["cpu_baseline"]["cached_n1"]["value"] == 119.4 and f["n1_reference"]["from"].startswith("profiles/")
    g = b.scale_fields(2, True, 4.6e5, 281.0, {"weak_views": {"value": 4.9e5, "unit": "rays/s", "ms_per_step": 529.0}}, [281.0, 279.0], None, None)
    assert g["strong"]["is_primary"] and g["strong"]["value"] == 4.6e5 and g["weak"]["value"] == 4.9e5
    assert g["strong"]["speedup_vs_cached_n1"] is None and g["cpu_baseline"]["cached_n1"] is None
    import json
    json.dumps(f); json.dumps(g)
    name, line = b.cached_n1_line()
    assert name is None or (line["n_gpus"] == 1 and line["value"] > 0)
