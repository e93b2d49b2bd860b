#!/usr/bin/env python
"""Turn the rocprofv3 CSVs a GPU run left under gpurun_out/<run>/ into the small, committed summaries
under profiles/ (kernel-trace stats, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes).

    python tools/summarize_profiles.py gpurun_out/r2_cfg2 r02 cfg2      (third argument: configuration key)

HBM traffic follows MI355X_MICROARCH.md §HBM: the two counters are collected in separate --pmc passes;
both are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide (16 B/lane) coalesced
reads — which is what every kernel here issues — so read bytes = 2 * FETCH_SIZE * 1024."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"(?:void )?capf::igemm_f32_kernel<(\d+), (\d+), (\d+), \d+, \d+, \d+, (\d+),", name)
    if m:
        nw, bm, bn, mode = m.groups()
        return f"igemm_f32<w{nw},{bm}x{bn},{'conv' if mode == '1' else 'rows'}>"
    m = re.match(r"(?:void )?capf::igemm_bf16_kernel<(\d+), (\d+),", name)
    if m:
        return f"igemm_bf16<w4,{m.group(1)}x{m.group(2)},conv>"
    m = re.match(r"(?:void )?capf::igemm_wino43_group(_db)?_kernel", name)
    if m:
        return "igemm_wino43_group"
    m = re.match(r"(?:void )?capf::igemm_wino_group_kernel", name)
    if m:
        return "igemm_wino_group"
    m = re.match(r"(?:void )?capf::igemm_wino_kernel", name)
    if m:
        return "igemm_wino<w4,F(2,3)>"
    m = re.match(r"(?:void )?capf::igemm_f32_pw_kernel", name)
    if m:
        return "igemm_f32_pw<w4,128x64>"
    m = re.match(r"(?:void )?capf::igemm_f32_group_kernel", name)
    if m:
        return "igemm_f32_group"
    m = re.match(r"(?:void )?capf::igemm_bf16_group_(pp|rh|ws)_kernel", name)
    if m:
        return "igemm_bf16_group_" + m.group(1)
    m = re.match(r"(?:void )?capf::igemm_bf16_rh_kernel", name)
    if m:
        return "igemm_bf16_rh<w4,126x64,conv>"
    m = re.match(r"(?:void )?capf::igemm_bf16_stem_stream_kernel", name)
    if m:
        return "igemm_bf16_stem_stream<w4,64x64>"
    m = re.match(r"(?:void )?capf::igemm_bf16_stem_kernel", name)
    if m:
        return "igemm_bf16_stem<w4,128x64>"
    m = re.match(r"(?:void )?capf::igemm_bf16_smallc_kernel", name)
    if m:
        return "igemm_bf16_smallc<w4,128x64>"
    m = re.match(r"(?:void )?capf::igemm_bf16_group_kernel", name)
    if m:
        return "igemm_bf16_group"
    m = re.match(r"(?:void )?capf::igemm_bf16_pwchain_kernel", name)
    if m:
        return "igemm_bf16_pwchain<64,256,64>"
    m = re.match(r"(?:void )?capf::igemm_f32_pwchain_kernel", name)
    if m:
        return "igemm_f32_pwchain<64,256,64>"
    m = re.match(r"(?:void )?capf::igemm_f32_stem_stream_kernel", name)
    if m:
        return "igemm_f32_stem_stream<w4,64x64>"
    m = re.match(r"(?:void )?capf::igemm_f32_smallc_kernel", name)
    if m:
        return "igemm_f32_smallc<w4,128x64>"
    m = re.match(r"(?:void )?capf::igemm_f32h2g_kernel<(\d+),", name)
    if m:
        return "igemm_f32h2g<conv>" if m.group(1) == "1" else "igemm_f32h2g<rows>"
    m = re.match(r"(?:void )?capf::(\w+?)(?:_kernel)?[<(]", name)
    return m.group(1) if m else name[:60]


def csrc_sha():
    """sha256 over the kernel sources the library is built from (sorted names + contents of csrc/*.hip, *.h, *.cpp, Makefile and include/capf.h):
    stored beside the HBM traffic of a round so that bench.py can tell whether the counters were collected on the sources it is timing."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "contextaware-poseformer_amd", "csrc")
    files = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile")
    for f in files + [os.path.join("..", "..", "include", "capf.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else None
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    for sub in sorted(os.listdir(src)):
        stats = [f for f in os.listdir(os.path.join(src, sub)) if f.endswith("kernel_stats.csv")] \
            if os.path.isdir(os.path.join(src, sub)) else []
        for f in stats:
            rows = list(csv.DictReader(open(os.path.join(src, sub, f))))
            dst = os.path.join(out, f"{tag}_{key + '_' if key else ''}{sub}_kernel_stats.csv")
            with open(dst, "w") as fo:
                fo.write("kernel,short,calls,total_ms,avg_us,percent,min_us,max_us\n")
                for r in rows:
                    if "capf" not in r["Name"]:
                        continue
                    fo.write(f"\"{r['Name']}\",\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},"
                             f"{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']},{float(r['MinNs']) / 1e3:.2f},"
                             f"{float(r['MaxNs']) / 1e3:.2f}\n")
            print("wrote", dst)
    traffic = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        d = os.path.join(src, sub)
        if not os.path.isdir(d):
            continue
        for f in os.listdir(d):
            if not f.endswith("counter_collection.csv"):
                continue
            n = collections.Counter()
            for r in csv.DictReader(open(os.path.join(d, f))):
                if r["Counter_Name"] != ctr or "capf" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                traffic[k][ctr + "_KiB"] += float(r["Counter_Value"])
                n[k] += 1
            for k, v in n.items():
                traffic[k]["launches"] = v
    if traffic:
        for k, t in traffic.items():
            n = max(1, t["launches"])
            t["read_bytes_per_launch"] = 2.0 * t["FETCH_SIZE_KiB"] * 1024 / n      # gfx950 correction, see docstring
            t["write_bytes_per_launch"] = t["WRITE_SIZE_KiB"] * 1024 / n
            t["hbm_bytes_per_launch"] = t["read_bytes_per_launch"] + t["write_bytes_per_launch"]
        # bytes per FORWARD: the stem kernel runs once per forward in every configuration, so its launch count is the number of forwards the
        # counters saw -- bench.py divides by ITS count of grouped launches per forward (a level of the two-piece tile is two kernel launches
        # at batch 512, which made cfg3's per-kernel-launch average read as half a level in round 5)
        fw = [t["launches"] for k, t in traffic.items() if "stem_stream" in k]
        if fw and fw[0] > 0:
            for k, t in traffic.items():
                t["hbm_bytes_per_forward"] = (2.0 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024 / fw[0]
        dst = os.path.join(out, f"{tag}_hbm_traffic.json")
        if key:           # one file per round, one entry per configuration
            allcfg = json.load(open(dst)) if os.path.exists(dst) else {}
            allcfg[key] = traffic
            allcfg[key]["_csrc_sha"] = csrc_sha()          # (the sources these counters were collected on; bench.py nulls `traffic` on a mismatch)
            json.dump(allcfg, open(dst, "w"), indent=1, sort_keys=True)
        else:
            json.dump(traffic, open(dst, "w"), indent=1, sort_keys=True)
        print("wrote", dst)
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj) and key:
        lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
        if lines:
            dst = os.path.join(out, f"{tag}_{key}_bench.json")
            open(dst, "w").write(lines[-1] + "\n")
            print("wrote", dst)
    be = os.path.join(src, "bench.err")
    if os.path.exists(be) and key:
        rows = [l for l in open(be).read().splitlines() if "launches/step" in l]
        if rows:
            dst = os.path.join(out, f"{tag}_{key}_kernel_table.txt")
            open(dst, "w").write("\n".join(rows) + "\n")
            print("wrote", dst)


if __name__ == "__main__":
    main()
