import csv, sys, glob, statistics, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_legs_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print(c, "no file"); continue
    rows = list(csv.DictReader(open(f[0])))
    by = {}
    for r in rows:
        n = r["Kernel_Name"]
        if "gemm_split_mfma" not in n: continue
        key = (re.sub(r"btvae_gemm::|void ", "", n)[:60], r["Grid_Size"])
        by.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(c, k, "n", len(v), "median KB", statistics.median(v))
