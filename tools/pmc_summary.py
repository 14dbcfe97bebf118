"""Mean per-dispatch PMC values of the fused kernel from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(list)
meta = {}
for row in csv.DictReader(open(sys.argv[1])):
    if "fused_kernel" not in row["Kernel_Name"]:
        continue
    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    meta = {k: row[k] for k in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Grid_Size")}
print(meta)
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g} max={max(v):.4g}")
