"""Minimal attribute-style config (the reference passes OmegaConf nodes: `opt.key`, `opt.get(key, default)`).

OmegaConf/hydra are not hard requirements of the hot path; any object with attribute access and `.get` works
(an OmegaConf DictConfig can be passed to Multiply(...) unchanged).  `load_config` reads the same YAML layout
as the reference's confs/model/*.yaml.
"""
import os
import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default


def to_config(obj):
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


def load_config(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(__file__), "confs", "model.yaml")
    with open(path) as f:
        return to_config(yaml.safe_load(f))
