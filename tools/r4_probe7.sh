cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r4e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for b in inverse classic; do
  CFMM_BACKSUB=$b timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$b -o t -- python $R/tools/profile_newton.py --solves 2 > $O/tr_$b.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for b in ("inverse", "classic"):
    f = glob.glob(f"gpurun_out/r4e/tr_{b}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "chol_step" in r["Kernel_Name"]]
    ch.sort()
    # group into factorisations of 32 launches; take the last one
    d = [(e - s) / 1e3 for s, e in ch]
    gaps = [(ch[i + 1][0] - ch[i][1]) / 1e3 for i in range(len(ch) - 1)]
    last = d[-32:]
    print(b, "launches", len(d), "mean %.2f" % (sum(d) / len(d)), "last factorisation per launch:", " ".join("%.1f" % x for x in last))
    print("   gaps (last 31): mean %.2f" % (sum(gaps[-31:]) / 31), "span of last factorisation %.1f us" % ((ch[-1][1] - ch[-32][0]) / 1e3))
    other = {}
    for r in rows:
        k = r["Kernel_Name"][:40]
        other.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("   %-42s n %4d mean %.2f total %.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
