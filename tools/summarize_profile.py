#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into committed summaries under profiles/ and refresh
profiles/traffic.json (HBM bytes per launch of the attention kernel, FETCH_SIZE doubled per the gfx950 correction)."""
import collections, csv, glob, json, os, shutil, sys

tag, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c2")
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)  # gpurun merges runs: older files of the same tag may linger
ks = newest(f"{src}/kt/*/*_kernel_stats.csv")
shutil.copy(ks, f"profiles/{tag}_{wl}_kernel_stats.csv")


def is_lowbit_attention(k):
    """The low-bit attention kernel of a workload: attn_fwd16_kernel<D, QT = 3 (int8 codes), ...> or the fp8-PV attn_fwd_kernel<...>
    (QT = 0 / 1 instances of attn_fwd16_kernel are the un-quantised fp16 / bf16 kernel bench.py times as a comparison point)."""
    if "attn_fwd16_kernel<" in k:
        return k.split("<", 1)[1].split(",")[1].strip() == "3"
    return "attn_fwd_kernel<" in k


out = {}
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    fs = glob.glob(f"{src}/{d}/*/*_counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0]
        if "lbfa" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_dur_ns_" + d].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for f in ("VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
                acc[k][f] = [float(r[f])]
    for k, v in acc.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
for k, v in out.items():
    if is_lowbit_attention(k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            v["clock_GHz_est"] = cyc / v["_dur_ns_pmc3"]
            v["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        tj = "profiles/traffic.json"
        t = json.load(open(tj)) if os.path.exists(tj) else {}
        t[wl] = round(v["hbm_bytes_per_launch"])
        json.dump(t, open(tj, "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_{wl}_pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    if is_lowbit_attention(k):
        print(k, json.dumps(v, indent=1))
print(open(f"profiles/{tag}_{wl}_kernel_stats.csv").read()[:900])
