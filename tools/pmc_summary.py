"""Sum rocprofv3 --pmc counter_collection csv per kernel and counter (averages per launch); --json also writes
{kernel: {COUNTER_per_image: value}} (launch average / --batch) for bench.py's issue_bound object."""
import argparse, csv, collections, glob, json, re
ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--json")
args = ap.parse_args()
only = {'channels_kernel', 'cascade_tile_kernel', 'octaves_block_kernel', 'channels_u1_kernel', 'alive_reduce_kernel', 'octaves_tail_kernel', 'det_pack_kernel'}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for path in args.dirs:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"][:50]
            if "wb_casc_jit" in row["Kernel_Name"]:
                k = "cascade_tile_kernel"                    # (the model-specialised build of the tile kernel)
            if only and k not in only:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k][row["Counter_Name"]] += 1
out = {}
for k in acc:
    print(k)
    out[k] = {}
    for c in sorted(acc[k]):
        v = acc[k][c] / n[k][c]
        out[k][c + "_per_image"] = v / args.batch
        print(f"   {c:28s} {v:16.1f}  (x{n[k][c]})")
    w = acc[k].get("SQ_WAVES", 0) / max(n[k].get("SQ_WAVES", 1), 1)
    if w:
        g = lambda c: acc[k][c] / n[k][c] if c in acc[k] else float("nan")
        print(f"   per wave: VALU {g('SQ_INSTS_VALU') / w:.0f}  SALU {g('SQ_INSTS_SALU') / w:.0f}  LDS {g('SQ_INSTS_LDS') / w:.0f}  "
              f"wave quad-cycles {g('SQ_WAVE_CYCLES') / w:.0f}  parked {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.2f}  "
              f"LDS conflict share {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.2f}")
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
