#!/usr/bin/env python3
"""Build-time check: no hot kernel may spill silently.

    python scripts/check_spills.py exp-trmf-nips16_amd/build/remarks_float.log [...]

Reads the -Rpass-analysis=kernel-resource-usage remarks hipcc printed while building a library (the Makefile's `lib` target
captures them) and fails if any kernel of the solver's hot path -- F-solve, X-side Gram, the CG kernels -- uses scratch
memory (ScratchSize > 0 bytes per lane = spilled registers), printing the offenders with their register counts.  Kernels that are
known to spill and are NOT on a path any supported problem takes by default are listed in ALLOW with the reason."""
import re
import subprocess
import sys

HOT = ('fsolve_', 'gram_x_kernel', 'hv_tile_kernel', 'cg_persist_kernel', 'apply_kernel', 'ar_tile_kernel', 'loss_kernel', 'apply_shared_mfma_kernel',
       'theta_', 'dense_tn_mfma_kernel', 'small_gram_mfma_kernel', 'cg_close_kernel')
ALLOW = {}


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main(paths):
    bad, seen = [], 0
    for path in paths:
        cur = {}
        rows = []
        for line in open(path, errors='replace'):
            m = re.search(r'remark: .*?Function Name: (\S+)', line)
            if m:
                cur = {'name': m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r'remark: .*?\s(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)', line)
            if m and cur is not None:
                cur[m.group(1)] = int(m.group(2))
        names = demangle([r['name'] for r in rows])
        for r in rows:
            full = names.get(r['name'], r['name'])
            short = re.sub(r'^void trmf::', '', full).split('(')[0]
            if not any(h in short for h in HOT):
                continue
            seen += 1
            if r.get('ScratchSize [bytes/lane]', 0) > 0:
                if any(a in short for a in ALLOW):
                    continue
                bad.append('%s: %s  scratch %d B/lane, %d VGPRs, %d spilled' % (path, short, r['ScratchSize [bytes/lane]'], r.get('VGPRs', -1), r.get('VGPRs Spill', -1)))
    if seen == 0:
        print('check_spills: no kernel-resource-usage remarks found in', paths)
        return 2
    if bad:
        print('check_spills: hot kernels with scratch (spilled registers):')
        for b in bad:
            print('   ', b)
        return 1
    print('check_spills: %d hot-path kernel instantiations, none uses scratch' % seen)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
