"""tools/predict_scaling.py (multi-GPU readiness without a second GPU): the committed prediction regenerates from the committed, measured
iter_usage maps, and the sharding it models is the one nerfart_amd.dist deals."""
import json
import os
import sys

import numpy as np

from conftest import REPO


def test_committed_prediction_regenerates_from_the_measured_maps(tmp_path, capsys):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import predict_scaling as ps
    maps = os.path.join(REPO, "profiles", "r05b_iter_usage_maps.npz")
    out = tmp_path / "pred.json"
    ps.predict(maps, str(out))
    capsys.readouterr()
    got = json.load(open(out))
    want = json.load(open(os.path.join(REPO, "profiles", "r05_predicted_scaling.json")))
    assert got == want
    row = got["strong_tiles"]["480x270_beta0.01"]
    assert row["rays"] == 480 * 270 and 0.9 <= row["N"]["8"]["2048"]["efficiency"] <= 1.0
    assert row["N"]["8"]["512"]["efficiency"] >= row["N"]["8"]["2048"]["efficiency"]          # finer interleave balances better
    assert got["weak_views"]["480x270_beta0.01"]["N"]["8"]["efficiency_mean_over_steps"] > 0.98


def test_the_model_deals_tiles_as_dist_does():
    from nerfart_amd import dist as nd
    n, tile, world = 129600, 2048, 8
    owner = (np.arange(n) // tile) % world
    for q in range(world):
        mine = nd.my_ray_indices(n, tile, q, world).numpy()
        assert np.array_equal(mine, np.nonzero(owner == q)[0])
