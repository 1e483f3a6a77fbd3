"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats CSV as microseconds per step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total {tot/1e3/steps:.1f} us/step over {steps:.0f} steps, {sum(int(r['Calls']) for r in rows)/steps:.0f} launches/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:70]:70s} n/step {int(r['Calls'])/steps:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} us/step {float(r['TotalDurationNs'])/1e3/steps:8.1f} {float(r['Percentage']):5.1f}%")
