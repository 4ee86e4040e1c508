"""Fold gpurun_out/prof (scripts/profile_gpu.sh) into profiles/: kernel-stats CSVs and
rNN_traffic.json with HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950
correction of MI355X_MICROARCH.md, HBM section).  usage: python scripts/pmc_to_json.py r01"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = "gpurun_out/prof"
def counter(task, which, name):
    tot, n = {}, {}
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (src, task, which), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    return {k: tot[k] / n[k] for k in tot}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3` "
               "(blackhole B=64; mri B=64, y batch 1); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the "
               "gfx950 correction of MI355X_MICROARCH.md section HBM", "kernels": {}}
for task in ("blackhole", "mri"):
    fe, wr = counter(task, "fetch", "FETCH_SIZE"), counter(task, "write", "WRITE_SIZE")
    for k in sorted(set(fe) & set(wr)):
        if not k.startswith("k_"):
            continue
        out["kernels"][k] = dict(FETCH_SIZE_KB=fe[k], WRITE_SIZE_KB=wr[k],
                                 hbm_bytes_per_launch=(2 * fe[k] + wr[k]) * 1024)
    for f in glob.glob("%s/%s_stats/**/*kernel_stats.csv" % (src, task), recursive=True):
        shutil.copy(f, "profiles/%s_%s_kernel_stats.csv" % (tag, task))
json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
