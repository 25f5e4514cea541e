#!/usr/bin/env python3
"""Runs the Python snippet of README.md as it stands there (in a scratch directory): the drop-in calls a new user makes first."""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
text = (ROOT / 'README.md').read_text()
start = text.index('```python') + len('```python')
code = text[start:text.index('```', start)]
os.chdir(tempfile.mkdtemp(prefix='readme_'))
ns = {}
exec(compile(code, 'README.md', 'exec'), ns)
print('V_cc =', ns['v'])
print('Sobol inputs:', ns['s']['inputs'])
print('generate_data keys:', sorted(ns['data']))
