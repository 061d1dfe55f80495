#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output dir (csv format) into a short text summary for profiles/."""
import csv
import glob
import os
import sys


def main(d, out, title):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        raise SystemExit("no *_kernel_stats.csv under " + d)
    f.sort(key=os.path.getmtime, reverse=True)   # newest run if the directory holds several
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats --output-format csv ; file {os.path.basename(f[0])}\n")
        o.write(f"# total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches\n")
        o.write(f"{'total_ms':>10} {'pct':>7} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel\n")
        for r in rows[:25]:
            name = r["Name"]
            if len(name) > 110:
                name = name[:107] + "..."
            o.write(f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):7.2f} {int(r['Calls']):6d} "
                    f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f}  {name}\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
