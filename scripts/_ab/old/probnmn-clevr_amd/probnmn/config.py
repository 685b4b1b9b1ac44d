"""``Config`` -- the reference's package-wide configuration object (reference: probnmn/config.py:6-272)
without ``yacs``: nested keys as attributes, defaults for every phase, overridden first by a YAML file
and then by a flat ``[key, value, key, value, ...]`` list (dotted keys), frozen afterwards.
Key names and default values are the reference's (probnmn/config.py:48-216); they are the contract the
models' ``from_config`` classmethods read."""
import copy
from typing import Any, Dict, List, Optional

import yaml


class _Node(dict):
    """dict with attribute access; immutable once frozen."""

    def __init__(self, mapping: Optional[Dict[str, Any]] = None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (mapping or {}).items():
            self[k] = _Node(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError("Config is immutable; override values through the YAML file or the override list")
        self[name] = value

    def __setitem__(self, key, value):
        if getattr(self, "_frozen", False):
            raise AttributeError("Config is immutable; override values through the YAML file or the override list")
        super().__setitem__(key, value)

    def freeze(self):
        for v in self.values():
            if isinstance(v, _Node):
                v.freeze()
        object.__setattr__(self, "_frozen", True)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _Node) else copy.deepcopy(v)) for k, v in self.items()}


_DEFAULTS: Dict[str, Any] = {
    "RANDOM_SEED": 0,
    "PHASE": "joint_training",
    "SUPERVISION": 1000,
    "SUPERVISION_QUESTION_MAX_LENGTH": 40,
    "OBJECTIVE": "ours",
    "DATA": {
        "VOCABULARY": "data/clevr_vocabulary",
        "TRAIN": {}, "VAL": {}, "TEST": {},
        "TRAIN_TOKENS": "data/clevr_train_tokens.h5",
        "TRAIN_FEATURES": "data/clevr_train_features.h5",
        "VAL_TOKENS": "data/clevr_val_tokens.h5",
        "VAL_FEATURES": "data/clevr_val_features.h5",
        "TEST_TOKENS": "data/clevr_test_tokens.h5",
        "TEST_FEATURES": "data/clevr_test_features.h5",
    },
    "PROGRAM_PRIOR": {"INPUT_SIZE": 256, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "DROPOUT": 0.0},
    "PROGRAM_GENERATOR": {"INPUT_SIZE": 256, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "DROPOUT": 0.0},
    "QUESTION_RECONSTRUCTOR": {"INPUT_SIZE": 256, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "DROPOUT": 0.0},
    "NMN": {"IMAGE_FEATURE_SIZE": [1024, 14, 14], "MODULE_CHANNELS": 128, "CLASS_PROJECTION_CHANNELS": 1024,
            "CLASSIFIER_LINEAR_SIZE": 1024},
    "ALPHA": 100.0,
    "BETA": 0.1,
    "GAMMA": 1.0,
    "DELTA": 0.99,
    "OPTIM": {"BATCH_SIZE": 256, "NUM_ITERATIONS": 20000, "WEIGHT_DECAY": 0.0, "LR_INITIAL": 0.00001,
              "LR_GAMMA": 0.5, "LR_PATIENCE": 3},
    "CHECKPOINTS": {
        "PROGRAM_PRIOR": "checkpoints/program_prior_best.pth",
        "QUESTION_CODING": "checkpoints/question_coding_1000_ours_best.pth",
        "MODULE_TRAINING": "checkpoints/module_training_1000_ours_best.pth",
    },
}


def _merge(dst: _Node, src: Dict[str, Any], path: str = "") -> None:
    for k, v in src.items():
        if k not in dst:
            raise KeyError("unknown config key: %s%s" % (path, k))
        if isinstance(dst[k], _Node):
            if not isinstance(v, dict):
                raise ValueError("config key %s%s is a section" % (path, k))
            _merge(dst[k], v, path + k + ".")
        else:
            dst[k] = type(dst[k])(v) if isinstance(dst[k], (int, float)) and not isinstance(dst[k], bool) else v


class Config:
    def __init__(self, config_yaml: Optional[str] = None, config_override: Optional[List[Any]] = None):
        root = _Node(copy.deepcopy(_DEFAULTS))
        if config_yaml:
            with open(config_yaml) as f:
                _merge(root, yaml.safe_load(f) or {})
        override = list(config_override or [])
        if len(override) % 2:
            raise ValueError("config_override must be [key, value, key, value, ...]")
        for key, value in zip(override[0::2], override[1::2]):
            node = root
            parts = key.split(".")
            for part in parts[:-1]:
                node = node[part]
            if parts[-1] not in node:
                raise KeyError("unknown config key: %s" % key)
            old = node[parts[-1]]
            if isinstance(value, str) and not isinstance(old, str):
                value = yaml.safe_load(value)
            node[parts[-1]] = value
        for section in ("PROGRAM_PRIOR", "PROGRAM_GENERATOR", "QUESTION_RECONSTRUCTOR"):
            if float(root[section]["DROPOUT"]) != 0.0:  # (fails here, with the key named, not deep inside a constructor)
                raise NotImplementedError("%s.DROPOUT = %s: the gfx950 recurrent kernels are built without dropout "
                                          "(no reference config sets it)" % (section, root[section]["DROPOUT"]))
        root.freeze()
        object.__setattr__(self, "_root", root)

    def __setattr__(self, name, value):
        raise AttributeError("Config is immutable; override values through the YAML file or the override list")

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_root"), name)

    def dump(self, file_path: str) -> None:
        with open(file_path, "w") as f:
            yaml.safe_dump(self._root.to_dict(), f)

    def __str__(self):
        return yaml.safe_dump(self._root.to_dict())
