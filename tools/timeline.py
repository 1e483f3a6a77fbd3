"""One steady-state step from a rocprofv3 --kernel-trace CSV as a per-queue timeline (development aid).
usage: python tools/timeline.py <dir with *kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys
csv.field_size_limit(1 << 30)
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("k_plan_count", "k_plan_build", "k_plan_single"))]
s, e = idx[-3], idx[-2]
t0 = int(rows[s]["Start_Timestamp"])
print("step wall (plan to plan): %.1f us" % ((int(rows[e]["Start_Timestamp"]) - t0) / 1e3))
qs = sorted(set(r["Queue_Id"] for r in rows[s:e]))
busy = {q: 0.0 for q in qs}
for r in rows[s:e]:
    st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = qs.index(r["Queue_Id"])
    busy[r["Queue_Id"]] += en - st
    print(f"{st:8.1f} {en - st:6.1f}  q{q}  {'      ' * q}{r['Kernel_Name'].replace('void ', '')[:40]}")
print({f"q{qs.index(q)}": round(v, 1) for q, v in busy.items()})
