"""Normalisation scales (ml/dataset/normalization.py:6-22): the `normalization_scales` section of _field_stats.yaml."""
from __future__ import annotations

from pathlib import Path
from typing import Dict

import yaml


def load_normalization_scales(stats_path: Path) -> Dict[str, float]:
    stats_path = Path(stats_path)
    if not stats_path.exists():
        raise FileNotFoundError(f"Stats file not found: {stats_path}")
    with open(stats_path) as f:
        data = yaml.safe_load(f)
    if "normalization_scales" not in data:
        raise KeyError(f"'normalization_scales' section not found in {stats_path}")
    s = data["normalization_scales"]
    return {"S_density": float(s["S_density"]), "S_velx": float(s["S_velx"]), "S_velz": float(s["S_velz"])}
