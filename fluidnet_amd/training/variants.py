"""Variant configurations with the reference's VariantManager interface (ml/scripts/variant_manager.py:7-208): YAML
variants discovered under <config_root>/variants, merged base_config -> model architecture -> variant chain (root to leaf),
parent chains for the K1 -> K2 -> ... curriculum, warm-start checkpoint paths and dependency ordering.  Reads the
reference's own config tree unchanged (ml/config/)."""
from __future__ import annotations

from copy import deepcopy
from pathlib import Path
from typing import Dict, List, Optional

import yaml


class VariantManager:
    def __init__(self, config_root: Path, checkpoints_dir: Path) -> None:
        self.config_root = Path(config_root)
        self.base_config_path = self.config_root / "base_config.yaml"
        self.architectures_dir = self.config_root / "model_architectures"
        self.variants_dir = self.config_root / "variants"
        self.checkpoints_dir = Path(checkpoints_dir)
        self._variants_cache = self.discover_variants()

    # ---- files
    def load_yaml(self, path: Path) -> dict:
        with open(path) as f:
            return yaml.safe_load(f) or {}

    def discover_variants(self) -> Dict[str, Path]:
        found: Dict[str, Path] = {}
        for path in sorted(self.variants_dir.rglob("*.yaml")):
            try:
                name = self.load_yaml(path).get("variant_name")
            except Exception as e:          # an unreadable file is reported and skipped, as in the reference
                print(f"Warning: Failed to load {path}: {e}")
                continue
            if not name:
                continue
            if name in found:
                raise ValueError(f"Duplicate variant name '{name}' found:\n  {found[name]}\n  {path}")
            found[name] = path
        return found

    def get_variant_yaml_path(self, variant_name: str) -> Path:
        if variant_name not in self._variants_cache:
            raise FileNotFoundError(f"Variant '{variant_name}' not found in {self.variants_dir}\n"
                                    f"Available variants: {', '.join(sorted(self._variants_cache))}")
        return self._variants_cache[variant_name]

    def get_variant_relative_dir(self, variant_name: str) -> Path:
        return self.get_variant_yaml_path(variant_name).parent.relative_to(self.variants_dir)

    # ---- inheritance
    def deep_merge(self, base: dict, override: dict) -> dict:
        out = deepcopy(base)
        for key, value in override.items():
            if isinstance(out.get(key), dict) and isinstance(value, dict):
                out[key] = self.deep_merge(out[key], value)
            else:
                out[key] = deepcopy(value)
        return out

    def get_parent_variant(self, variant_name: str) -> Optional[str]:
        return self.load_yaml(self.get_variant_yaml_path(variant_name)).get("parent_variant")

    def resolve_dependency_chain(self, variant_name: str) -> List[str]:
        """[root, ..., variant_name] following parent_variant links."""
        chain: List[str] = []
        cur: Optional[str] = variant_name
        while cur:
            if cur in chain:
                raise ValueError(f"Circular dependency: {cur} in chain {chain}")
            chain.append(cur)
            cur = self.get_parent_variant(cur)
        return chain[::-1]

    def _architecture_of(self, chain: List[str]) -> str:
        for v in chain:
            arch = self.load_yaml(self.get_variant_yaml_path(v)).get("model_architecture")
            if arch:
                return arch
        raise ValueError(f"Could not find model_architecture in chain for {chain[-1]}")

    def build_config_with_inheritance(self, variant_name: str) -> dict:
        chain = self.resolve_dependency_chain(variant_name)
        arch = self.load_yaml(self.get_variant_yaml_path(chain[0])).get("model_architecture")
        if not arch:
            raise ValueError(f"Variant {chain[0]} missing 'model_architecture' field")
        arch_path = self.architectures_dir / f"{arch}.yaml"
        if not arch_path.exists():
            raise FileNotFoundError(f"Architecture '{arch}' not found at {arch_path}")
        merged = self.deep_merge(self.load_yaml(self.base_config_path), self.load_yaml(arch_path))
        for v in chain:
            data = self.load_yaml(self.get_variant_yaml_path(v))
            va = data.get("model_architecture")
            if va is not None and va != arch:
                raise ValueError(f"Architecture mismatch: {chain[0]} uses {arch}, but {v} uses {va}")
            merged = self.deep_merge(merged, data)
        merged["_variant_metadata"] = {
            "variant_name": variant_name, "model_architecture_name": arch, "full_model_name": f"{arch}_{variant_name}",
            "parent_variant": self.get_parent_variant(variant_name), "inheritance_chain": chain,
            "relative_dir": self.get_variant_relative_dir(variant_name)}
        return merged

    # ---- checkpoints
    def _checkpoint_path(self, variant_name: str) -> Path:
        arch = self._architecture_of(self.resolve_dependency_chain(variant_name))
        return self.checkpoints_dir / self.get_variant_relative_dir(variant_name) / f"{arch}_{variant_name}" / "best_model.pth"

    def get_warmstart_checkpoint(self, variant_name: str) -> Optional[Path]:
        parent = self.get_parent_variant(variant_name)
        if not parent:
            return None
        path = self._checkpoint_path(parent)
        if not path.exists():
            raise FileNotFoundError(f"Parent checkpoint not found: {path}\nTrain {parent} before training {variant_name}")
        return path

    def check_checkpoint_exists(self, variant_name: str) -> bool:
        return self._checkpoint_path(variant_name).exists()

    # ---- ordering
    def topological_sort(self, variants: List[str]) -> List[str]:
        """Requested variants plus all their ancestors, every parent before its children."""
        nodes = set()
        for v in variants:
            nodes.update(self.resolve_dependency_chain(v))
        parent = {v: self.get_parent_variant(v) for v in nodes}
        done: List[str] = []
        pending = sorted(nodes)
        while pending:
            ready = [v for v in pending if parent[v] is None or parent[v] in done]
            if not ready:
                raise ValueError("Circular dependency detected")
            done.append(ready[0])
            pending.remove(ready[0])
        return done
