"""Attribute-style config node and YAML loader reading the reference's keys
(reference nemo/utils/configuration.py: ConfigNode :13-99, include expansion :221-238).
Only what the solver path reads is needed: cfg.inference.{optimizer, scheduler, epochs, inter_mode,
blur_radius, init_mode, pre_render, realign, batch_size, *_sample}, cfg.get('batch_size_no_grad'),
cfg.get('optimize_translation'), cfg.model.down_sample_rate."""
import os

import yaml


class ConfigNode(dict):
    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = ConfigNode(v) if isinstance(v, dict) else v

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


def load_config(path, root=None, overrides=None):
    """Load a YAML file; string values ending in .yaml are replaced by the file they name (relative to `root`)."""
    root = root or os.path.dirname(os.path.abspath(path))

    def expand(node):
        if isinstance(node, dict):
            return {k: expand(v) for k, v in node.items()}
        if isinstance(node, str) and node.endswith(".yaml"):
            cand = node if os.path.isabs(node) else os.path.join(root, node)
            if os.path.exists(cand):
                with open(cand) as fh:
                    return expand(yaml.safe_load(fh))
        return node

    with open(path) as fh:
        cfg = ConfigNode(expand(yaml.safe_load(fh)))
    for key, value in (overrides or {}).items():
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, ConfigNode())
        node[parts[-1]] = value
    return cfg


def default_pose3d_config():
    """The inference block of config/omni_nemo_pose_3d.yaml:15-65 + config/models/nemo_runtime.yaml (values only)."""
    return ConfigNode({
        "task": "3d_pose_estimation",
        "model": {"down_sample_rate": 8, "num_noise": 5, "max_group": 512},
        "inference": {
            "init_mode": "3d_batch", "batch_size": 20, "classification": False,
            "search_translation": False, "optimize_translation": False,
            "px_sample": {"min": 0.0, "max": 0.0, "num": 1}, "py_sample": {"min": 0.0, "max": 0.0, "num": 1},
            "azim_sample": {"min_pi": 0.0, "max_pi": 2.0, "num": 12},
            "elev_sample": {"min_pi": -0.1667, "max_pi": 0.3333, "num": 4},
            "theta_sample": {"min_pi": -0.1667, "max_pi": 0.1667, "num": 3},
            "dist_sample": {"min": 5.0, "max": 5.0, "num": 1},
            "min_distance": 2, "pre_rendering_thr": 80, "blur_radius": 0.0, "inter_mode": "bilinear", "epochs": 300,
            "translation_scale": 0.1,
            "optimizer": {"class_name": "torch.optim.Adam", "lr": 0.05, "betas": [0.4, 0.6]},
            "scheduler": {"class_name": "torch.optim.lr_scheduler.ExponentialLR", "gamma": 0.2},
        },
    })
