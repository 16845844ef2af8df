"""Copies the rocprofv3 summaries worth judging from gpurun_out/prof_<tag>/ into profiles/ and derives the
per-launch HBM traffic of the dominant kernel (k_tower32) from the PMC passes:

    bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. exactly half of a
wide coalesced read, so it is doubled (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B stores.
The tower's weight reads are served by L2 / Infinity Cache after the first toucher, so the fabric-side counters
are an upper bound of true HBM traffic (Infinity-Cache hits are counted, same section)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(counter_csv, counter):
    acc = {}
    with open(counter_csv) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main(tag, ch, key=None):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)   # a re-run leaves older files beside the new ones
    st = newest(os.path.join(src, f"stats_c{ch}", "*", "*kernel_stats.csv"))
    shutil.copyfile(st, os.path.join(dst, f"{tag}_c{ch}_kernel_stats.csv"))
    b = os.path.join(src, f"bench_stats_c{ch}.json")
    if os.path.exists(b):
        shutil.copyfile(b, os.path.join(dst, f"{tag}_c{ch}_bench_under_rocprof.json"))
    fetch = per_kernel(newest(os.path.join(src, f"pmc_fetch_c{ch}", "*", "*counter_collection.csv")), "FETCH_SIZE")
    write = per_kernel(newest(os.path.join(src, f"pmc_write_c{ch}", "*", "*counter_collection.csv")), "WRITE_SIZE")
    rows = []
    traffic = {}
    tpath = os.path.join(dst, "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k, (0, 0)), write.get(k, (0, 0))
        by = fk[0] * 1024 * 2 + wk[0] * 1024
        rows.append((k, fk[1], fk[0], wk[0], by))
        if (key or "k_tower").split("<")[0] in k:   # the kernel the key names (k_tower32 by default, k_step for the fused pipeline)
            traffic[key or f"k_tower32<{ch}>"] = round(by)
    with open(os.path.join(dst, f"{tag}_c{ch}_pmc_hbm.csv"), "w") as f:
        f.write("kernel,launches,avg_FETCH_SIZE_KiB,avg_WRITE_SIZE_KiB,hbm_bytes_per_launch(2*FETCH+WRITE)*1024\n")
        for r in rows:
            f.write('"%s",%d,%.3f,%.3f,%.0f\n' % r)
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(open(os.path.join(dst, f"{tag}_c{ch}_pmc_hbm.csv")).read())
    print(open(os.path.join(dst, f"{tag}_c{ch}_kernel_stats.csv")).read())


if __name__ == "__main__":
    # optional 3rd argument: key of the tower kernel in profiles/traffic.json (default k_tower32<ch>; the fp8 passes use
    # k_tower32<fp8,ch>, which is what bench.py looks up for an fp8 run)
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", sys.argv[2] if len(sys.argv) > 2 else "128", sys.argv[3] if len(sys.argv) > 3 else None)
