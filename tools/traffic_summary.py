"""HBM-side bytes per launch from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; batch-1 launches):
median over the dispatches, KB * 1024, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced
reads on gfx950 (exact for 16 B/lane loads, uncalibrated for the channel kernel's 4 B/lane source loads)."""
import csv, glob, json, re, statistics, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            name = "cascade_tile_kernel" if "wb_casc_jit" in row["Kernel_Name"] else (m.group(1) if m else None)   # (the model-specialised build of the tile kernel)
            if name:
                vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"_note": __doc__.strip()}
for k, c in vals.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fr, wr = statistics.median(c["FETCH_SIZE"]), statistics.median(c["WRITE_SIZE"])
    out[k] = {"fetch_size_kb_raw": fr, "fetch_bytes_corrected": fr * 1024 * 2, "write_bytes": wr * 1024,
              "traffic_bytes_per_launch_b1": fr * 1024 * 2 + wr * 1024, "dispatches": len(c["FETCH_SIZE"])}
print(json.dumps(out, indent=1))
