"""`hallmd.utils.load_thruster` (src/hallmd/utils.py:24-84): read a device description for `hallthruster_jl`."""
import json
from pathlib import Path

import yaml


def load_thruster(thruster_dir: str | Path, thruster_filename: str = 'thruster.yml') -> dict:
    """Load `thruster_dir/thruster_filename` (.yml or .json: name, geometry, magnetic_field, shielded, ...) and make every
    file it refers to absolute: a string value that names a file under `thruster_dir` -- by its path relative to the
    directory or by its bare file name -- is replaced by that file's resolved POSIX path, so that the solver can open
    e.g. the magnetic-field table from any working directory."""
    root = Path(thruster_dir)
    spec = root / thruster_filename
    with open(spec, 'r', encoding='utf-8') as fd:
        if spec.suffix == '.yml':
            device = yaml.safe_load(fd)
        elif spec.suffix == '.json':
            device = json.load(fd)
        else:
            raise ValueError(f'Unsupported file type "{spec.suffix}". Only .yml and .json files are supported.')
    known = {}
    for f in sorted(p for p in root.rglob('*') if p.is_file() and p != spec):
        known.setdefault(f.relative_to(root).as_posix(), f)
    for f in list(known.values()):
        known.setdefault(f.name, f)

    def absolute(node):
        if isinstance(node, dict):
            return {k: absolute(v) for k, v in node.items()}
        if isinstance(node, list):
            return [absolute(v) for v in node]
        if isinstance(node, str) and node in known:
            return known[node].resolve().as_posix()
        return node
    return absolute(device)
