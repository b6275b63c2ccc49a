"""Pose-grid helpers (reference nemo/utils/general.py: tensor_linspace :11-40, get_param_samples :106-145)
and normalize_features (reference nemo/utils/features.py:4-5)."""
import numpy as np
import torch


def tensor_linspace(start, end, steps=10):
    """linspace between two equally shaped tensors along a new trailing dimension."""
    assert start.size() == end.size()
    shape = (1,) * start.dim() + (steps,)
    w0 = torch.linspace(1, 0, steps=steps).to(start).view(shape)
    w1 = torch.linspace(0, 1, steps=steps).to(start).view(shape)
    return w0 * start.contiguous().unsqueeze(-1) + w1 * end.contiguous().unsqueeze(-1)


def _span(node, scale=1.0, endpoint=True):
    lo = node["min_pi"] if "min_pi" in node else node["min"]
    hi = node["max_pi"] if "max_pi" in node else node["max"]
    return np.linspace(lo * scale, hi * scale, node["num"], endpoint=endpoint)


def get_param_samples(cfg):
    """(azimuth, elevation, theta, distance, px, py) sample axes, float64, from cfg.inference.*_sample.
    The azimuth axis excludes its end point; angles are the YAML decimals times pi."""
    inf = cfg["inference"] if not hasattr(cfg, "inference") else cfg.inference
    return (_span(inf["azim_sample"], np.pi, endpoint=False), _span(inf["elev_sample"], np.pi),
            _span(inf["theta_sample"], np.pi), _span(inf["dist_sample"]), _span(inf["px_sample"]),
            _span(inf["py_sample"]))


def normalize_features(x, dim=0):
    return x / torch.sum(x ** 2, dim=dim, keepdim=True)[0] ** 0.5
