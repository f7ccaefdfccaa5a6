"""Two routing tables (tools/routing_table.py) cell by cell: usage compare_routing_tables.py OLD.txt NEW.txt [threshold]
Prints the cells whose time changed by more than the threshold (default 10 %) and a summary."""
import sys
def load(path):
    rows = {}
    for line in open(path):
        p = line.split()
        if len(p) >= 7 and p[0] in ("uniform300", "lognormal", "bimodal100_3000") and p[1] in ("sw", "nw", "hw", "ov"):
            rows[(p[0], p[1], p[2], int(p[3]), int(p[4]))] = (float(p[5]), float(p[6]), " ".join(p[7:]))
    return rows
old, new = load(sys.argv[1]), load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.10
faster = slower = same = 0
for key in sorted(set(old) & set(new)):
    o, n = old[key], new[key]
    if o[0] < 0.15 and n[0] < 0.15:     # (searches of a handful of targets: launch latencies, box to box)
        continue
    ratio = n[0] / o[0]
    if ratio < 1 - thr:
        faster += 1
        tag = "FASTER"
    elif ratio > 1 + thr:
        slower += 1
        tag = "SLOWER"
    else:
        same += 1
        continue
    print(f"{tag} {key[0]:16s} {key[1]} {key[2]:5s} N={key[3]:8d} Q={key[4]:5d}: {o[0]:9.3f} -> {n[0]:9.3f} ms ({o[1]:6.2f} -> {n[1]:6.2f} TCUPS) routing {o[2]} -> {n[2]}")
print(f"# {faster} cells more than {thr:.0%} faster, {slower} slower, {same} within, of {len(set(old) & set(new))} common cells")
