"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) into profiles/<round>_pmc_hbm.json.
usage: python tools/pmc_summary.py gpurun_out/pmc_fetch/fetch_counter_collection.csv \
                                   gpurun_out/pmc_write/write_counter_collection.csv profiles/r01_pmc_hbm.json [views clouds]"""
import collections
import csv
import json
import sys

KEEP = ("blend_kernel", "preprocess_kernel", "traverse_kernel", "ds_scatter_kernel", "ds_count_kernel", "ds_scan_kernel",
        "tile_scatter_kernel", "tile_count_kernel", "seg_scan_kernel", "coarse_kernel", "fine_kernel", "bbox_partial_kernel",
        "fused_kernel", "rpe_attention_kernel", "geo_embedding", "fps_multi_kernel", "kpconv", "tq_kernel", "tq_expand_kernel")


def short(name):
    name = name.replace("gr::(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("rocprim"):
        return name[:80]
    return name.split("(")[0]


def collect(path, counter):
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if not any(t in k for t in KEEP):
            continue
        acc[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


def main():
    fetch, write, out = sys.argv[1:4]
    views = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    clouds = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    f = collect(fetch, "FETCH_SIZE")
    w = collect(write, "WRITE_SIZE")
    kernels = []
    for k in sorted(set(f) | set(w)):
        kernels.append({"kernel": k, "launches": f.get(k, w.get(k))[1], "fetch_size_kb_avg": round(f.get(k, (0, 0))[0], 1),
                        "write_size_kb_avg": round(w.get(k, (0, 0))[0], 1)})
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/pmc.sh), command: python bench.py "
                   "--steps 3 --warmup 1 --no-cpu-baseline --pairs 0 --no-extras --no-single-view; values in KB per launch as reported. Per MI355X_MICROARCH.md (HBM "
                   "section) FETCH_SIZE on gfx950 counts wide coalesced reads at 1/2 -> hbm_bytes ~= (2*FETCH_SIZE + WRITE_SIZE)*1024.",
           "config": {"raster": f"1M Gaussians, 640x480, {views} views/launch", "radius": f"{clouds} x 200k-pt clouds/launch, r=0.0625"},
           "units_per_launch": {"raster_blend": views, "radius_fill": clouds, "radius_count": clouds, "radius_fused": clouds,
                                "radius_search": clouds, "radius_tq_limited": clouds},
           "kernels": kernels, "traffic_bytes_per_launch": {}}
    tr = doc["traffic_bytes_per_launch"]
    for k in kernels:  # the kernels bench.py prices against the HBM roofline
        hbm = int((2 * k["fetch_size_kb_avg"] + k["write_size_kb_avg"]) * 1024)
        if k["kernel"].startswith("tq_kernel<32, false") or k["kernel"].startswith("tq_expand_kernel"):
            tr["radius_search"] = tr.get("radius_search", 0) + hbm   # the bare search: compact rows + their expansion
        if k["kernel"].startswith("tq_kernel<32, true"):
            tr["radius_tq_limited"] = hbm
        if k["kernel"].startswith("blend_kernel"):
            doc["traffic_bytes_per_launch"]["raster_blend"] = hbm
        if k["kernel"].startswith("traverse_kernel<128, true"):
            doc["traffic_bytes_per_launch"]["radius_fill"] = hbm
        if k["kernel"].startswith("traverse_kernel<128, false"):
            doc["traffic_bytes_per_launch"]["radius_count"] = hbm
        if k["kernel"].startswith("fused_kernel"):
            doc["traffic_bytes_per_launch"]["radius_fused"] = hbm
    json.dump(doc, open(out, "w"), indent=1)
    for k in kernels:
        print(k)


if __name__ == "__main__":
    main()
