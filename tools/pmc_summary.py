"""Sum rocprofv3 --pmc counter_collection csv per kernel and counter (averages per launch)."""
import csv, sys, collections, glob, re
only = {'channels_kernel', 'cascade_tile_kernel', 'octaves_block_kernel', 'channels_u1_kernel', 'alive_reduce_kernel', 'octaves_tail_kernel'}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"][:50]
            if only and k not in only:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:28s} {acc[k][c] / n[k][c]:16.1f}  (x{n[k][c]})")
