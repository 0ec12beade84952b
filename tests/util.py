"""Shared helpers for the test-suite: golden loading, config access, comparison utilities."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from paintmind_amd.config import ver2cfg  # noqa: E402


def load_golden(name):
    """-> (params dict keyed by state_dict name, dict of the remaining arrays)"""
    z = np.load(os.path.join(GOLDEN, name))
    params = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    data = {k: z[k] for k in z.files if not k.startswith("w:")}
    return params, data


def api_facts():
    return json.load(open(os.path.join(GOLDEN, "api.json")))


def vq_cfg(version):
    return ver2cfg[version]


def s2_cfg(version):
    return ver2cfg[version]


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def to_torch_sd(params):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
