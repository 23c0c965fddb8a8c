#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

    python3 tools/profiling/traffic_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
    python3 tools/profiling/traffic_json.py --stamp <out.json> profiles/traffic.json      (where the git history is)

Per leg of the benchmark the dominant kernel is picked by its name and its grid; the figure recorded is the median
counter value per dispatch, in bytes (the counters are in KiB), FETCH_SIZE doubled: on gfx950 it tallies the 128-byte
requests of wide coalesced reads at 64 bytes.  bench.py reports these figures only while the kernel sources still hash to
what is recorded here."""
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

# leg key of bench.recorded_traffic -> (regex on the kernel name, workgroups of the launch, threads per workgroup)
LEGS = {
    "training_fwd:fp32_split2:65536": (r"gemm_split_mfma<0, 0, 2, true, 2, 1, 0>", 512 * 8, 256),
    "training_fwd_pairs:fp32_split2:65536": (r"gemm_pq256<2, 2, 0>", 256 * 4, 512),
    "training_dgrad_gstats:fp32_split2:65536": (r"gemm_split_mfma<0, 1, 7, true, 2, 1, 0>", 512 * 8, 256),
    "training_dgrad_bnbwd:fp32_split2:65536": (r"gemm_split_mfma<0, 1, 6, true, 2, 1, 0>", 512 * 8, 256),
    "scoring_fwd_pairs:fp32_split2:819200": (r"gemm_pq256<3, 2, 0>", 3200 * 4, 512),
    "imputation_fwd:fp32_split2:25000": (r"gemm_split_mfma<0, 0, 3, true, 2, 1, 0>", 196 * 8, 256),
    "wide_hidden_layer:bf16:65536": (r"gemm_pq256<2, 2, 1>", 256 * 4, 512),
    "wide_hidden_dgrad_gstats:bf16:65536": (r"gemm_pq256<7, 1, 1>", 512 * 4, 256),
    "wide_first_layer_k552:bf16:65536": (r"gemm_bf16_mfma<0, 0, 2, false, false>", 512 * 8, 256),
}


def medians(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter]
    out = {}
    for key, (pattern, groups, threads) in LEGS.items():
        vals = [float(r["Counter_Value"]) for r in rows
                if re.search(pattern, r["Kernel_Name"]) and int(r["Grid_Size"]) == groups * threads]
        if vals:
            out[key] = (statistics.median(vals) * 1024.0, len(vals))
    return out


def stamp(src, dst):
    """Copy a record made on the GPU box (which has no git history) into profiles/, filling in the commit whose kernel
    sources it was measured on; refuses a record whose hash is not that of the working tree's sources."""
    import bench
    rec = json.load(open(src))
    if rec["kernel_sources_hash"] != bench.kernel_sources_hash():
        raise SystemExit(f"{src} was measured on other kernel sources ({rec['kernel_sources_hash']} != {bench.kernel_sources_hash()})")
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "behavio_temporal_vae_amd/csrc"], capture_output=True, text=True).stdout.strip()
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    rec["commit"] = head + (" + uncommitted kernel changes" if dirty else "")
    json.dump(rec, open(dst, "w"), indent=1)
    print("wrote", dst, "for commit", rec["commit"])


def main():
    import bench
    if sys.argv[1] == "--stamp":
        return stamp(sys.argv[2], sys.argv[3])
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    fetch, write = medians(fetch_dir, "FETCH_SIZE"), medians(write_dir, "WRITE_SIZE")
    legs = {}
    for key in LEGS:
        if key in fetch and key in write:
            legs[key] = {"fetch_x2_bytes": 2.0 * fetch[key][0], "write_bytes": write[key][0], "dispatches": [fetch[key][1], write[key][1]]}
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    except OSError:
        commit = "unknown"
    rec = {"kernel_sources_hash": bench.kernel_sources_hash(), "commit": os.environ.get("BTVAE_COMMIT", commit),
           "command": "bash tools/profiling/pmc_r3.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py --steps 2 --warmup 1 --wide-steps 3)",
           "units": "bytes per dispatch, median over the pass's dispatches of that kernel and grid; FETCH_SIZE x 2 (gfx950)",
           "legs": legs}
    json.dump(rec, open(out_path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
