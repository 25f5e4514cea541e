"""Host-side logic of bench.py that needs no GPU."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_counter_traffic_is_replayed_only_for_the_kernel_sources_in_the_tree(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 --pmc measurement (profiles/traffic_r*.json).  A record is replayed only if
    it was taken at the bench's launch size, for its input layout, and carries the digest of the kernel sources in the tree
    (`kernel_srchash`): after a kernel edit the old ratio is dropped with a reason, not quoted.  CPU only."""
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    (tmp_path / 'profiles').mkdir()
    monkeypatch.setattr(bench, 'ROOT', tmp_path)
    monkeypatch.setattr(bench, 'kernel_source_hash', lambda: h)
    rec = {'samples_per_launch': 1_250_000, 'hbm_bytes_per_launch': 1.08e9, 'layout': 'tile', 'kernel_srchash': h}
    assert bench.read_committed_traffic(1_250_000, 'tile')[:2] == (None, None)                    # nothing committed
    (tmp_path / 'profiles' / 'traffic_r01a.json').write_text(json.dumps({**rec, 'kernel_srchash': 'feedfacefeedface', 'hbm_bytes_per_launch': 1.0}))
    got = bench.read_committed_traffic(1_250_000, 'tile')
    assert got[0] is None and 'other kernel sources' in got[2]                                    # stale: dropped, and says why
    (tmp_path / 'profiles' / 'traffic_r02a.json').write_text(json.dumps({k: v for k, v in rec.items() if k != 'kernel_srchash'}))
    assert bench.read_committed_traffic(1_250_000, 'tile')[0] is None                             # a record without a digest (rounds 1-2)
    (tmp_path / 'profiles' / 'traffic_r03a.json').write_text(json.dumps(rec))
    assert bench.read_committed_traffic(1_250_000, 'tile')[:2] == (1.08e9, 'traffic_r03a.json')
    assert bench.read_committed_traffic(1_250_000, 'soa')[0] is None                              # measured with the other input layout
    assert bench.read_committed_traffic(1_000_000, 'tile')[0] is None                             # another launch size
    # the committed record of this round, if its digest still matches the tree, is what the driver's bench will replay
    monkeypatch.undo()
    real = bench.read_committed_traffic(1_250_000, 'tile')
    assert real[0] is None or 0.95 * 872 * 1_250_000 < real[0] < 1.1 * 872 * 1_250_000
