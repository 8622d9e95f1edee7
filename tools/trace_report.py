"""Reads gpurun_out/r06_frametrace/tail.csv (tools/frame_trace.sh): per ordinary frame the GPU busy time of the mapping part
by kernel, and one local-optimisation iteration kernel by kernel."""
import sys
from collections import Counter
rows = [l.rstrip("\n").split(",", 2) for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_frametrace/tail.csv")]
rows = [(int(a), int(b), n) for a, b, n in rows]
short = lambda n: n.split("(")[0].replace("void ", "").replace("at::native::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::").replace("rocprim::ROCPRIM_400001_NS::detail::", "rocprim::")[:44]
marks = [i for i, r in enumerate(rows) if "icp_vertex3" in r[2]]
done = 0
for k in range(1, len(marks)):
    a, b = marks[k - 1], marks[k]
    if not (250 < b - a < 500):
        continue
    seg = rows[a:b]
    tm = [i for i, r in enumerate(seg) if "transform_map" in r[2]]
    fl = [i for i, r in enumerate(seg) if "icp_fill" in r[2]]
    if len(tm) < 2 or not fl:
        continue
    mp = seg[tm[1] + 1:fl[0] + 1]
    c, cnt = Counter(), Counter()
    for s, e, n in mp:
        c[short(n)] += (e - s) / 1e3
        cnt[short(n)] += 1
    print("ordinary frame: %d mapping kernels, GPU busy %.0f us" % (len(mp), sum(c.values())))
    for name, v in c.most_common(16):
        print("   %7.1f us  x%-3d %s" % (v, cnt[name], name))
    done += 1
    if done >= 2:
        break
tails = [i for i, r in enumerate(rows) if "map_fused_tail" in r[2]]
pairs = [(tails[k], tails[k + 1]) for k in range(len(tails) - 1) if 8 <= tails[k + 1] - tails[k] <= 16]
if pairs:
    a, b = pairs[len(pairs) // 2]
    t0 = rows[a][1]; prev = t0
    print("one iteration: %d kernels, %.1f us" % (b - a, (rows[b][1] - t0) / 1e3))
    for s, e, n in rows[a + 1:b + 1]:
        print("   %7.1f  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, short(n)))
        prev = e
