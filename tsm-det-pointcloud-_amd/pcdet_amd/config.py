"""YAML config with attribute access, `_BASE_CONFIG_` inheritance and `--set K V` overrides.

Same behaviour as the reference's pcdet/config.py:16-85 (EasyDict based), without the easydict dependency:
modules receive their sub-dict as `model_cfg` and use both `cfg.KEY` and `cfg.get('KEY', default)`.
"""
import ast
import os
from pathlib import Path

import yaml


class AttrDict(dict):
    """dict whose keys are also attributes; nested dicts (also inside lists) are converted recursively."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kw):
        for k, v in dict(other or {}, **kw).items():
            self[k] = v


def _load_yaml(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def merge_new_config(config, new_config, base_dir=None):
    if "_BASE_CONFIG_" in new_config:
        base = new_config["_BASE_CONFIG_"]
        if not os.path.isabs(base) and not os.path.exists(base) and base_dir is not None:
            base = os.path.join(base_dir, base)
        config.update(AttrDict(_load_yaml(base)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = AttrDict()
        merge_new_config(config[key], val, base_dir)
    return config


def cfg_from_yaml_file(cfg_file, config):
    new_config = _load_yaml(cfg_file)
    # _BASE_CONFIG_ paths in the reference are relative to tools/ (the cwd of train.py); also try <cfg dir>/../..
    base_dir = str(Path(cfg_file).resolve().parent.parent.parent)
    merge_new_config(config=config, new_config=new_config, base_dir=base_dir)
    return config


def cfg_from_list(cfg_list, config):
    """`--set A.B value ...` overrides with type checks (reference config.py:16-48)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split(".")
        d = config
        for sub in keys[:-1]:
            assert sub in d, "NotFoundKey: %s" % sub
            d = d[sub]
        sub = keys[-1]
        assert sub in d, "NotFoundKey: %s" % sub
        try:
            value = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            value = v
        cur = d[sub]
        if type(value) != type(cur) and isinstance(cur, dict):
            for item in value.split(","):
                ck, cv = item.split(":")
                cur[ck] = type(cur[ck])(cv)
        elif type(value) != type(cur) and isinstance(cur, list):
            # "a,b" stays a string after literal_eval, "3,3" becomes a tuple: accept both
            parts = value.split(",") if isinstance(value, str) else list(value)
            d[sub] = [type(cur[0])(x) for x in parts]
        else:
            assert type(value) == type(cur), "type {} does not match original type {}".format(type(value), type(cur))
            d[sub] = value


def log_config_to_file(cfg, pre="cfg", logger=None):
    for key, val in cfg.items():
        if isinstance(val, AttrDict):
            logger.info("\n%s.%s = edict()" % (pre, key))
            log_config_to_file(val, pre=pre + "." + key, logger=logger)
            continue
        logger.info("%s.%s: %s" % (pre, key, val))


cfg = AttrDict()
cfg.ROOT_DIR = str((Path(__file__).resolve().parent / "../").resolve())
cfg.LOCAL_RANK = 0
