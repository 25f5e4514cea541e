"""`hallmd.utils.load_thruster` (src/hallmd/utils.py:24-84): read a device description for `hallthruster_jl`."""
import json
import os
from pathlib import Path

import yaml


def _first_reference(node, value):
    """(dict, key) of the first entry equal to `value`, visiting nested DICTS in key order, depth first -- what the reference's
    `_path_in_dict` finds (src/hallmd/utils.py:12-21).  Lists are not searched: the reference never looks inside one."""
    if isinstance(node, dict):
        for key, child in node.items():
            if isinstance(child, dict):
                hit = _first_reference(child, value)
                if hit is not None:
                    return hit
            elif child == value:
                return node, key
    return None


def load_thruster(thruster_dir: str | Path, thruster_filename: str = 'thruster.yml') -> dict:
    """Load `thruster_dir/thruster_filename` (.yml or .json: name, geometry, magnetic_field, shielded, ...) and make the files
    it refers to absolute, so that the solver can open e.g. the magnetic-field table from any working directory.

    As in the reference (src/hallmd/utils.py:67-85; pinned by tests/golden/hallthruster_jl.json `device_cases`): every file under
    `thruster_dir` is looked for once -- by its path relative to the directory, and only if that is not mentioned by its bare
    name -- and the FIRST mention (nested dicts in key order, depth first) is replaced by the file's resolved POSIX path; a
    second mention of the same file and names inside lists are left as they are.
    One deviation: a mention three or more dicts deep is replaced where it stands.  The reference restarts its walk from the
    top-level dict at every key (`d = config[key]`, utils.py:82) and ends in a KeyError there."""
    root = Path(thruster_dir)
    spec = root / thruster_filename
    with open(spec, 'r', encoding='utf-8') as fd:
        if spec.suffix == '.yml':
            device = yaml.safe_load(fd)
        elif spec.suffix == '.json':
            device = json.load(fd)
        else:
            raise ValueError(f'Unsupported file type "{spec.suffix}". Only .yml and .json files are supported.')
    for folder, _, names in os.walk(root):
        for name in names:
            path = Path(folder) / name
            hit = _first_reference(device, path.relative_to(root).as_posix()) or _first_reference(device, name)
            if hit is not None:
                holder, key = hit
                holder[key] = path.resolve().as_posix()
    return device
