#!/usr/bin/env python3
"""profiles/fsolve_traffic.json (one entry per configuration) from a PMC summary written by scripts/pmc_fsolve.sh.

    python scripts/make_traffic_json.py profiles/r03_pmc_fsolve.txt [c3|c5]

HBM bytes per F-solve launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB): FETCH_SIZE under-counts wide
coalesced streaming reads by exactly 2x on gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is exact for this kernel's store pattern (profiles/r02_write_size_calibration.txt: a known 19.2 MB
written as 16-, 12- and 4-byte-per-lane stores reads 18750.0 KB each).  The digest of the kernel sources is
recorded so that bench.py reports `roofline.traffic` only for the tree the counters were collected on."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import fsolve_source_digest   # noqa: E402

src = sys.argv[1]
vals = {}
for line in open(src):
    m = re.match(r'\s+(\w+)\s+([\d.]+)\s+\(n=', line)
    if m:
        vals[m.group(1)] = float(m.group(2))
fetch_kb, write_kb = vals['FETCH_SIZE'], vals['WRITE_SIZE']
try:
    commit = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ''
config = sys.argv[2] if len(sys.argv) > 2 else 'c3'
nnz, n, T, k, s, kernel = {'c3': (9950287, 100000, 10000, 40, 4, 'fsolve_quad_kernel<3,40>'),
                           'c5': (49974995, 1000000, 50000, 64, 8, 'fsolve_mfma_kernel<4,64>')}[config]
for line in open(src):                       # the summary records the entry count of the run it was taken on
    m = re.search(r'nnz[= ](\d+)', line)
    if m:
        nnz = int(m.group(1))
out = {
    'note': __doc__.split('\n\n')[2].replace('\n', ' '),
    'source': os.path.relpath(src, ROOT), 'source_commit': commit, 'kernel_source_sha256': fsolve_source_digest(),
    'config': config, 'kernel': kernel,
    'fetch_size_kb': fetch_kb, 'write_size_kb': write_kb,
    'traffic_bytes': int(round((2 * fetch_kb + write_kb) * 1024)),
    'tcc_hit_rate': vals['TCC_HIT'] / (vals['TCC_HIT'] + vals['TCC_MISS']) if 'TCC_HIT' in vals else None,
    'algorithmic_bytes': nnz * (4 + s + k * s) + (n + 1) * 8 + n * k * s,
    'compulsory_bytes': nnz * (4 + s) + (n + 1) * 8 + T * k * s + n * k * s,
}
path = os.path.join(ROOT, 'profiles', 'fsolve_traffic.json')
try:
    table = json.load(open(path))
    if 'traffic_bytes' in table:             # round-2 format: a single config-3 entry
        table = {table.get('config', 'c3'): table}
except (OSError, ValueError):
    table = {}
table[config] = out
json.dump(table, open(path, 'w'), indent=2)
print(json.dumps(out, indent=2))
