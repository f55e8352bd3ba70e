# rocprofv3 kernel stats of one of the widened flavours (configs 4 / 5), every kernel incl. the ones PyTorch launches:
#   bash tools/profile_flavour.sh <tag> <bench_flavours.py arguments...>      (on the GPU box; outputs under gpurun_out/prof_<tag>/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/prof_$T
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/bench_flavours.py "$@" > $O/flavour.json 2> $O/flavour.err
F=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp "$F" $O/${T}_kernel_stats.csv
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"]) / tot:6.3f} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -rf $O/stats                          # raw trace
