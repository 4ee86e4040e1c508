"""Fold gpurun_out/prof (scripts/profile_gpu.sh) into profiles/: kernel-stats CSVs and
rNN_traffic.json with HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950
correction of MI355X_MICROARCH.md, HBM section).  usage: python scripts/pmc_to_json.py r01"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = "gpurun_out/prof"
def counter(task, which, name):
    # per kernel, the LARGEST dispatch: the benches also launch a kernel on one chain when they make
    # the measurement (op(x_gt)), and an average over dispatches would mix that launch in
    tot, n = {}, {}
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (src, task, which), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            tot[k] = max(tot.get(k, 0.0), float(row["Counter_Value"]))
            n[k] = 1
    return {k: tot[k] / n[k] for k in tot}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3` "
               "(blackhole B=64; mri B=256, y batch 1); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the "
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
# matrix-pipe counters (per launch, summed over the chip as rocprofv3 reports them)
mf = {}
for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_BF16", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
             "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
    for k, v in counter("blackhole", "mfma", name).items():
        if k.startswith("k_dft"):
            mf.setdefault(k, {})[name] = v
if mf:
    out["mfma"] = mf
    out["mfma_note"] = ("rocprofv3 --pmc pass over `bench.py --steps 3` (B=64): counter sums of a full-size launch; "
                        "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = share of the busy time in which a matrix "
                        "instruction is in the pipe (the kernels are HBM-bound: the matrix work hides under the stream)")
json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
