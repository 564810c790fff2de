#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_stats.csv) as a table."""
import glob, os, sqlite3, sys

def from_db(path):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, "
         "max(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6 from {} d join {} s on d.kernel_id=s.id "
         "group by s.kernel_name order by 6 desc").format(kd, ks)
    return list(cur.execute(q))

def main():
    rows = []
    for path in sys.argv[1:]:
        for f in ([path] if path.endswith('.db') else glob.glob(os.path.join(path, '**', '*.db'), recursive=True)):
            rows += from_db(f)
    total = sum(r[5] for r in rows)
    print('{:<100s} {:>6s} {:>10s} {:>10s} {:>10s} {:>10s} {:>6s}'.format('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', '%'))
    for r in rows:
        print('{:<100s} {:>6d} {:>10.2f} {:>10.2f} {:>10.2f} {:>10.3f} {:>6.1f}'.format(r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[5] / total))

if __name__ == '__main__':
    main()
