"""Per (kernel, grid) summary of a rocprofv3 --kernel-trace CSV: python tools/trace_summary.py <dir or *_kernel_trace.csv> [out.txt]
rocprofv3's own --stats table averages over every dispatch of a kernel name; since r04 a handle's range calibration launches the
encoder / decoder kernels on a small batch at creation, so the full-size dispatches are listed separately here (by grid size)."""
import collections, csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines = [f"# {src}", f"# {'kernel':70s} {'grid':>10s} {'calls':>6s} {'avg ms':>10s} {'min ms':>10s} {'max ms':>10s} {'total ms':>10s}"]
for (name, grid), ts in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"{name[:72]:72s} {grid:>10s} {len(ts):6d} {sum(ts)/len(ts):10.4f} {min(ts):10.4f} {max(ts):10.4f} {sum(ts):10.3f}")
txt = "\n".join(lines)
print("\n".join(lines[:14]))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
