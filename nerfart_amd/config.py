"""YAML config loading with the reference's schema and CLI override syntax, without ``addict``.

Mirrors utils/io_util.py:194-340: a nested attribute dict that raises KeyError on missing keys
(``ForceKeyErrorDict``), ``load_yaml(path, default_path)``, ``--k1:k2 value`` / ``--k value`` overrides typed
by the existing value (``update_config``), precedence CLI > yaml > base.  The four ``configs/*.yaml`` of the
reference load unchanged (tests/test_config.py).
"""
from __future__ import annotations

import argparse
import copy
import os

import yaml


class ConfigDict(dict):
    """dict with attribute access, recursive wrapping, KeyError on missing keys."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for d in args:
            if d is not None:
                for k, v in dict(d).items():
                    self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise KeyError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def update(self, *args, **kwargs):
        for d in args:
            for k, v in dict(d).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, ConfigDict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)

    def __deepcopy__(self, memo):
        return ConfigDict(copy.deepcopy(self.to_dict(), memo))


def load_yaml(path, default_path=None) -> ConfigDict:
    with open(path, encoding="utf8") as f:
        config = ConfigDict(**yaml.load(f, Loader=yaml.FullLoader))
    if default_path is not None and path != default_path:
        with open(default_path, encoding="utf8") as f:
            main = ConfigDict(**yaml.load(f, Loader=yaml.FullLoader))
        config = merge_nested(main, config)
    return config


def merge_nested(base: ConfigDict, override) -> ConfigDict:
    """`base` updated by `override` key by key, recursing where both sides hold a mapping (what addict's Dict.update does in
    the reference, io_util.py:201-212): a user yaml that sets `training: {lr: ...}` keeps the base file's other training keys."""
    for k, v in dict(override).items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            merge_nested(base[k], v)
        else:
            base[k] = v
    return base


def save_config(datadict: ConfigDict, path: str):
    d = copy.deepcopy(datadict)
    d.training.ckpt_file = None
    d.training.pop("exp_dir", None)
    with open(path, "w", encoding="utf8") as f:
        yaml.dump(d.to_dict(), f, default_flow_style=False)


def update_config(config: ConfigDict, unknown):
    """``--k1:k2 v`` and ``--k v`` overrides, value typed by the existing entry (io_util.py:234-257)."""
    for idx, arg in enumerate(unknown):
        if not arg.startswith("--"):
            continue
        if ":" in arg:
            k1, k2 = arg.replace("--", "").split(":")
            cur = config[k1][k2]
            if type(cur) == bool:
                v = unknown[idx + 1].lower() == "true"
            elif cur is not None:
                v = type(cur)(unknown[idx + 1])
            else:
                v = unknown[idx + 1]
            config[k1][k2] = v
        else:
            config[arg.replace("--", "")] = unknown[idx + 1]
    return config


def create_args_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default=None, help="Path to config file.")
    parser.add_argument("--resume_dir", type=str, default=None, help="Directory of experiment to load.")
    return parser


def load_config(args, unknown, base_config_path=None) -> ConfigDict:
    """command line --over--> args.config --over--> base yaml (io_util.py:268-340)."""
    assert (args.config is not None) != (args.resume_dir is not None), \
        "you must specify ONLY one in 'config' or 'resume_dir' "
    unknown = [u for u in unknown if "local_rank" not in u]
    if args.resume_dir is not None:
        config = update_config(load_yaml(os.path.join(args.resume_dir, "config.yaml")), unknown)
        config.training.exp_dir = args.resume_dir
    else:
        config = update_config(load_yaml(args.config, default_path=base_config_path), unknown)
        if "exp_dir" not in config.training:
            config.training.exp_dir = os.path.join(config.training.log_root_dir, config.expname)
    other = dict(vars(args))
    other.pop("config", None)
    other.pop("resume_dir", None)
    config.update(other)
    # device_ids: the reference spreads one process over GPUs with nn.DataParallel; here every process
    # owns exactly one GPU (LOCAL_RANK) and rays are sharded across processes (nerfart_amd/dist.py).
    config.device_ids = [int(os.environ.get("LOCAL_RANK", 0))]
    return config
