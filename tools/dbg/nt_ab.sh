set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ntab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for nt in 0 64; do
  LR2_GEMM_ABLATE=$nt python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-online --serial-streams > $O/plain_$nt.json 2>/dev/null
  LR2_GEMM_ABLATE=$nt rocprofv3 --kernel-trace --output-format csv -d $O/kt_$nt -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-online --serial-streams > $O/prof_$nt.json 2>/dev/null
  python3 $R/tools/rocprof_summary.py kernel-trace $O/kt_$nt --top 3 --md $O/kt_$nt.md
  rm -rf $O/kt_$nt
done
for nt in 0 64; do for k in plain prof; do python3 -c "
import json
d=json.loads(open('$O/${k}_$nt.json').read().strip().splitlines()[-1])
print('ablate=$nt', '$k', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done; sed -n 5p $O/kt_$nt.md | cut -c1-200; done
