"""`hallmd.utils.load_thruster` (src/hallmd/utils.py:24-84): read a device description for `hallthruster_jl`."""
import json
import os
from pathlib import Path

import yaml

_PARSERS = {'.yml': yaml.safe_load, '.json': json.load}


def _mentions(tree: dict, wanted: str):
    """Every (holder, key) with holder[key] == wanted, nested DICTS visited in key order, depth first; lists are leaves that
    never match -- the order in which the reference's `_path_in_dict` (src/hallmd/utils.py:12-21) would meet them."""
    for key, child in tree.items():
        if isinstance(child, dict):
            yield from _mentions(child, wanted)
        elif child == wanted:
            yield tree, key


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
    parse = _PARSERS.get(spec.suffix)
    if parse is None:
        raise ValueError(f'Unsupported file type "{spec.suffix}". Only .yml and .json files are supported.')
    device = parse(spec.read_text(encoding='utf-8')) if parse is yaml.safe_load else json.loads(spec.read_text(encoding='utf-8'))
    # the files in the order the reference meets them (os.walk): it matters when two of them compete for one mention
    present = [Path(folder, name) for folder, _, names in os.walk(root) for name in names]
    for found in present:
        hit = next(_mentions(device, found.relative_to(root).as_posix()), None) or next(_mentions(device, found.name), None)
        if hit:
            hit[0][hit[1]] = found.resolve().as_posix()
    return device
