"""Config loader keeps the reference's schema / CLI override semantics (utils/io_util.py:194-340) and the
reference's four YAMLs load unchanged when the reference tree is available."""
import os

import pytest

from nerfart_amd import config, scene, frameworks

REF_CFG = "/root/reference/configs"


def test_attribute_dict_semantics():
    c = config.ConfigDict({"a": {"b": 1, "c": [1, {"d": 2}]}, "x": None})
    assert c.a.b == 1 and c.a.c[1].d == 2
    with pytest.raises(KeyError):
        _ = c.a.zzz
    assert c.a.setdefault("e", 5) == 5 and c.a.e == 5
    assert c.to_dict() == {"a": {"b": 1, "c": [1, {"d": 2}], "e": 5}, "x": None}


def test_cli_overrides_are_typed_by_existing_value():
    c = config.ConfigDict({"data": {"downscale": 1, "pin_memory": True, "cam_file": None}, "expname": "x"})
    config.update_config(c, ["--data:downscale", "2", "--data:pin_memory", "false", "--data:cam_file", "cams.npz", "--expname", "y"])
    assert c.data.downscale == 2 and c.data.pin_memory is False and c.data.cam_file == "cams.npz" and c.expname == "y"


def test_synthetic_configs_build_both_frameworks():
    for fw, n in (("VolSDF", 796347), ("NeuS", 802491)):
        m, _, rk_train, rk_test, fn = frameworks.get_model(scene.synthetic_config(fw))
        assert sum(p.numel() for p in m.parameters()) == n
        assert rk_test["perturb"] is False and callable(fn)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present (GPU box)")
def test_reference_yamls_load_unchanged():
    for name, fw in (("volsdf_fangzhou_nature.yaml", "VolSDF"), ("volsdf_fangzhou_vangogh.yaml", "VolSDF"),
                     ("neus_fangzhou.yaml", "NeuS"), ("neus_fangzhou_vangogh.yaml", "NeuS")):
        c = config.load_yaml(os.path.join(REF_CFG, name))
        assert c.model.framework == fw
        m, _, _, rk_test, _ = frameworks.get_model(c)
        assert type(m).__name__ == fw
        if c.get("finetune"):
            assert c.finetune.target_text.startswith("painting")


def test_base_yaml_merges_nested_sections(tmp_path):
    """load_yaml(path, default_path): the user file overrides the base file key by key INSIDE nested sections (addict's
    recursive update in the reference, io_util.py:201-212) - a shallow update would drop the base's other training keys."""
    from nerfart_amd import config
    (tmp_path / "base.yaml").write_text("expname: base\ntraining:\n  lr: 1.0e-3\n  num_iters: 100\n  scheduler:\n    type: multistep\n    gamma: 0.5\nmodel:\n  W: 256\n")
    (tmp_path / "user.yaml").write_text("expname: mine\ntraining:\n  lr: 5.0e-4\n  scheduler:\n    gamma: 0.1\ndata:\n  downscale: 2\n")
    c = config.load_yaml(str(tmp_path / "user.yaml"), default_path=str(tmp_path / "base.yaml"))
    assert c.expname == "mine" and c.training.lr == 5.0e-4 and c.training.num_iters == 100
    assert c.training.scheduler.type == "multistep" and c.training.scheduler.gamma == 0.1
    assert c.model.W == 256 and c.data.downscale == 2


def test_get_model_configures_but_does_not_load_the_style_losses():
    """An `is_finetune: True` YAML (volsdf_fangzhou_vangogh.yaml's flag): get_model hands the trainer the config its losses are built
    from (volsdf.py:638-645) without touching CLIP / VGG - render.py with such a YAML must not need the checkpoints."""
    from nerfart_amd import scene, frameworks
    for fw in ("VolSDF", "NeuS"):
        cfg = scene.synthetic_config(fw)
        cfg.training.is_finetune = True
        cfg.finetune = {"src_text": "photo", "target_text": "painting", "clip_checkpoint": "/nonexistent/ViT-B-32.pt"}
        model, trainer, _, _, render_fn = frameworks.get_model(cfg, [480, 270])
        assert trainer._style_cfg[1] == (480, 270) and getattr(trainer, "style_loss", None) is None
        import pytest
        with pytest.raises(FileNotFoundError, match="ViT-B-32"):
            trainer._ensure_style_loss()
        cfg.training.is_finetune = False
        _, trainer2, _, _, _ = frameworks.get_model(cfg, [480, 270])
        assert trainer2._style_cfg is None


def test_get_model_ships_the_mixed_mode_and_the_yaml_can_override_it():
    """`model, trainer, ... = get_model(args); trainer(...)` must train without a further call: get_model sets the kernels' arithmetic -
    'mixed' by default (split-bf16 for every value that reaches a pixel or carries a gradient, VolSDF's no-gradient Algorithm-1 sampler on
    the 2-MFMA fp16 kernels; NeuS has no such sampler: split-bf16), `model.precision` in the YAML / `--model:precision` overrides it."""
    from nerfart_amd import scene, frameworks
    m, tr, rk_train, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    assert (m.precision, m.sampler_precision, m.mode) == ("bf16x3", "fp16x2", "mixed") and tr.native is True
    assert rk_train["perturb"] is True                      # the reference's default render_kwargs_train (volsdf.py:982) ...
    assert tr.resamples(rk_train) and not tr.resamples(dict(rk_train, perturb=False))      # ... under which pass 2 draws its own samples
    n, _, _, _, _ = frameworks.get_model(scene.synthetic_config("NeuS"))
    assert (n.precision, n.sampler_precision, n.mode) == ("bf16x3", None, "bf16x3")
    for name, want in (("bf16x3", ("bf16x3", None)), ("fp32", ("fp32", None)), ("mixed", ("bf16x3", "fp16x2"))):
        cfg = scene.synthetic_config("VolSDF")
        cfg.model.precision = name
        m2, tr2, _, _, _ = frameworks.get_model(cfg)
        assert (m2.precision, m2.sampler_precision) == want
        if name == "fp32":
            with pytest.raises(RuntimeError, match="split-bf16"):
                tr2.native
    # the rendering form of `mixed` (second session of round 6): a method, or a key of the YAML / of render.py's command line; the calibration itself is lazy
    assert m.sampler_late_round == 3 and m.calibrate_sampler() is m and (m.sampler_precision, m.sampler_guard, m.sampler_late_round) == ("fp16x1c", 0.005, 3)
    assert m.mode == "mixed (calibrated sampler)" and n.calibrate_sampler() is n and n.mode == "bf16x3"
    cfg = scene.synthetic_config("VolSDF")
    cfg.model.calibrate_sampler = True
    m3, tr3, _, _, _ = frameworks.get_model(cfg)
    assert m3.mode == "mixed (calibrated sampler)" and tr3.native is True
    with pytest.warns(UserWarning, match="calibrated for rendering"):
        tr3._training_sampler()                             # what every step entry point of a Trainer does first: weights are about to change
    assert m3.mode == "mixed"
    m.set_precision("mixed")
    m.set_sampler_precision("fp32")
    assert m.mode == "bf16x3+fp32 sampler (guard 0.005)"    # any sampler that is not the model's own arithmetic is guarded by default
    m.set_precision("bf16x3")                               # a precision is the whole mode: it resets the sampler's
    assert m.sampler_precision is None
    with pytest.raises(ValueError):
        m.set_precision("tf32")


def test_trainer_two_pass_knobs():
    from nerfart_amd import scene, frameworks
    from nerfart_amd.trainer import Trainer
    m, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    assert Trainer(m, reuse_pass1_samples=True).resamples({"perturb": True}) is False
    assert Trainer(m, resample_pass2=True).resamples({"perturb": False}) is True
    assert Trainer(m, resample_pass2=False).resamples({"perturb": True}) is False
    with pytest.raises(ValueError, match="contradicts"):
        Trainer(m, reuse_pass1_samples=True, resample_pass2=True)
    tr = Trainer(m)
    tr.uniform_source = lambda p, first, count, n, dev: __import__("torch").zeros(count + 1, n)
    with pytest.raises(ValueError, match="uniform_source returned"):
        tr._uniform(1, 0, 4, 64, "cpu")
