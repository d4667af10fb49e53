"""Per layer kind average kernel durations from a rocprofv3 kernel trace of tools/dev_v2.py --time (calls are in program
order B|A, T|F, F|T; 23 calls each per kernel instance)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "coupling_rqs" in n:
        by[n.split("(")[1].split(")")[-1] + n.split("::")[1].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    g = len(v) // 3
    if g == 0:
        continue
    parts = [v[i * g:(i + 1) * g] for i in range(3)]
    print(k, " | ".join(f"{name} avg {sum(p[3:]) / len(p[3:]):7.1f} min {min(p):7.1f} us" for name, p in zip(("B|A", "T|F", "F|T"), parts)))
