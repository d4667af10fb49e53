"""Combine two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units) into per-kernel HBM bytes per launch.
gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streams -> x 2 (MI355X_MICROARCH.md, HBM section).
usage: python tools/traffic_json.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, re, sys, collections


def means(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {}
for k, (f, n) in fetch.items():
    w = write.get(k, (0.0, 0))[0]
    flat = k.replace("(anonymous namespace)::", "").replace("void ", "")
    short = re.match(r"[\w:]+", flat).group(0)
    if short in out:   # template instances of one kernel: launch-weighted mean
        o = out[short]
        tot = o["launches"] + n
        for key, val in (("fetch_kib", f), ("write_kib", w)):
            o[key] = (o[key] * o["launches"] + val * n) / tot
        o["launches"] = tot
    else:
        out[short] = dict(fetch_kib=f, write_kib=w, launches=n)
for o in out.values():
    o["hbm_bytes_per_launch"] = 1024.0 * (2.0 * o["fetch_kib"] + o["write_kib"])
    o["rule"] = "2 x FETCH_SIZE + WRITE_SIZE (KiB), averaged over the launches of one bench run"
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, o in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:60]:60s} launches {o['launches']:4d}  {o['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")
