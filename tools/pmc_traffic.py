"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: the counters are KiB; FETCH_SIZE is doubled (gfx950 reports
half of a wide streaming read).  Output: JSON {kernel name: {"launches", "write_bytes", "fetch_bytes",
"hbm_bytes_per_launch"}} for every tmf:: kernel.

usage: python tools/pmc_traffic.py <write_counter_collection.csv> <fetch_counter_collection.csv> <out.json>"""
import csv, hashlib, json, os, sys, collections


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter or "tmf::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


w, f = load(sys.argv[1], "WRITE_SIZE"), load(sys.argv[2], "FETCH_SIZE")
out = {}
for k in sorted(set(w) | set(f)):
    n = max(w.get(k, [0])[0], f.get(k, [0])[0], 1)
    wb = w.get(k, [0, 0.0])[1] * 1024 / n
    fb = 2.0 * f.get(k, [0, 0.0])[1] * 1024 / n
    out[k] = {"launches": n, "write_bytes": round(wb), "fetch_bytes": round(fb), "hbm_bytes_per_launch": round(wb + fb)}
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "temfpy_amd", "csrc")
stamp = {f: hashlib.sha1(open(os.path.join(csrc, f), "rb").read()).hexdigest() for f in sorted(os.listdir(csrc))
         if f.endswith((".hip", ".hpp", ".cpp"))}
json.dump({"note": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes of `tools/profile_sweep.py 1024 512 3` "
                   "(three conversions of the benchmark workload); per-launch averages; FETCH_SIZE doubled per the gfx950 "
                   "correction (MI355X_MICROARCH.md, HBM section); bench.py refuses this file when source_sha1 does not "
                   "match the kernel sources it runs",
           "source_sha1": stamp, "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if "det" in k}, indent=1))
