"""YAML config loader standing in for ``OmegaConf.load`` (sample/sample.py:136; omegaconf is not
available offline).  The reference YAMLs (configs/*/*_sample.yaml) load unchanged; keys are
attributes that can also be assigned (``conf.ckpt = args.ckpt``, sample.py:137-138)."""
import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict):
            return Config({k: Config._wrap(x) for k, x in v.items()})
        if isinstance(v, list):
            return [Config._wrap(x) for x in v]
        return v


def load_config(path) -> Config:
    with open(path) as f:
        return Config._wrap(yaml.safe_load(f) or {})
