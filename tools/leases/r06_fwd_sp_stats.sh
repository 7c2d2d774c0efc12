#!/bin/bash
# rocprofv3 kernel stats of the training step alone, forward on the NHWC engine vs on the SP engine -> gpurun_out/r06/fwd_{nhwc,sp}_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in nhwc sp; do
  rm -rf /tmp/rp_$m
  DISCONET_FWD_MATH=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$m -o t -- python $R/tools/train_step_probe.py --dgrad sp --wgrad sp --steps 8 > $O/fwd_${m}_probe.log 2>&1
  f=$(find /tmp/rp_$m -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/fwd_${m}_stats.csv
  python3 - "$O/fwd_${m}_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], "total ms over the run", round(tot / 1e6, 2))
def grp(name):
    n = name
    for k in ("conv_mfma_kernel", "conv_sp_kernel", "conv_spq_kernel", "conv_pre_pair", "conv_wgrad_sp_s2", "conv_wgrad_sp", "conv_wgrad64", "conv_wgrad_kernel", "wgrad_reduce",
              "bn_apply", "bn_bwd_apply", "bn_bwd_reduce", "bn_stats", "fold_partials", "channel_sum", "sp_from_nhwc", "sp_pack", "pack", "copyBuffer", "fillBuffer"):
        if k in n:
            return k
    return n.split("(")[0][-40:]
g = {}
for r in rows:
    k = grp(r["Name"]); g.setdefault(k, [0, 0.0]); g[k][0] += int(r["Calls"]); g[k][1] += float(r["TotalDurationNs"])
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:22]:
    print("  %-28s calls %5d  ms %8.2f  (per step of 10: %.3f)" % (k, c, t / 1e6, t / 1e6 / 10))
PY
done
