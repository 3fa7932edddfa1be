#!/usr/bin/env python3
"""Summarise tools/pmc_passes.sh output: per-launch counter values of the LDPC kernel, and for every other kernel of the library (the
front-end and P1 kernels live in an anonymous namespace: named by their function) the per-launch average and the launch count."""
import csv, glob, json, os, re, sys
from collections import defaultdict
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("# rocprofv3 --pmc passes (tools/pmc_passes.sh %s): bench.py --no-cpu-baseline --no-clamped-variant --steps 2 --warmup 1" % tag)
print("# FETCH_SIZE / WRITE_SIZE in KiB as reported; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts wide coalesced reads at 1/2")
ldpc_avg = {}
for name in ("sq", "fetch", "write"):
    files = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("# pass %s: no output" % name)
        continue
    per = defaultdict(lambda: defaultdict(float))        # (kernel, dispatch) -> counter -> value
    dur = {}
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        key = (k, int(row["Dispatch_Id"]))
        per[key][row["Counter_Name"]] += float(row["Counter_Value"])
        dur[key] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    ldpc = sorted(k for k in per if "ldpc_decode" in k[0])
    print("## pass %s: %s, one column per launch (%d launches), then duration in us" % (name, ldpc[0][0].split("(")[0].split("::")[-1] if ldpc else "ldpc", len(ldpc)))
    counters = sorted({c for k in ldpc for c in per[k]})
    for c in counters:
        print("%-24s %s" % (c, " ".join("%.6g" % per[k][c] for k in ldpc)))
        if ldpc:
            ldpc_avg[c] = sum(per[k][c] for k in ldpc) / len(ldpc)
    print("%-24s %s" % ("dur_us", " ".join("%.1f" % dur[k] for k in ldpc)))
    other = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    tdur = defaultdict(float)
    for (k, d), cs in per.items():
        if "ldpc_decode" in k:
            continue
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:44]
        launches[short] += 1
        tdur[short] += dur[(k, d)]
        for c, v in cs.items():
            other[short][c] += v
    print("## pass %s: other kernels, AVERAGE PER LAUNCH (launches, avg us)" % name)
    for k in sorted(other):
        n = launches[k]
        print("%-46s n=%-3d %9.1f us  %s" % (k, n, tdur[k] / n, " ".join("%s=%.6g" % (c, v / n) for c, v in sorted(other[k].items()))))

# per-launch averages of the dominant kernel for bench.py (copy to profiles/ldpc_counters.json): frames / sweeps of the profiled workload from
# the bench line of the same run
try:
    line = [l for l in open(os.path.join(root, "gpurun_out", "pmc_%s_sq.json" % tag)) if l.startswith("{")][-1]
    d = json.loads(line)
    frames = int(re.search(r"(\d+) FEC frames", d["config"]["workload"]).group(1))
    sys.path.insert(0, root)
    import bench
    out = {"config": 3, "frames": frames, "sweeps": 25, "kernel": "ldpc_decode2_kernel", "kernel_sources_sha16": bench.ldpc_kernel_hash(), "source": "tools/pmc_passes.sh %s: separate rocprofv3 --pmc passes of bench.py --no-cpu-baseline --no-extra-legs, "
           "ldpc_decode2_kernel averaged over its launches" % tag,
           "SQ_INSTS_VALU": ldpc_avg["SQ_INSTS_VALU"], "FETCH_SIZE_KiB": ldpc_avg["FETCH_SIZE"], "WRITE_SIZE_KiB": ldpc_avg["WRITE_SIZE"],
           "SQ_LDS_IDX_ACTIVE": ldpc_avg["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": ldpc_avg["SQ_LDS_BANK_CONFLICT"], "SQ_WAIT_ANY": ldpc_avg["SQ_WAIT_ANY"],
           "SQ_WAVE_CYCLES": ldpc_avg["SQ_WAVE_CYCLES"], "GRBM_GUI_ACTIVE": ldpc_avg["GRBM_GUI_ACTIVE"], "SQ_INSTS_LDS": ldpc_avg["SQ_INSTS_LDS"],
           "SQ_INSTS_SALU": ldpc_avg["SQ_INSTS_SALU"]}
    mix = os.path.join(root, "profiles", "ldpc_isa_mix.json")            # tools/ldpc_isa_mix.py: static op mix x the ubench's per-class cycles
    if os.path.exists(mix):
        out.update(json.load(open(mix)))
    json.dump(out, open(os.path.join(root, "gpurun_out", "ldpc_counters_%s.json" % tag), "w"), indent=1)
except Exception as e:                                   # noqa: BLE001 -- a summary tool: say why and go on
    print("# ldpc_counters json not written:", e)
