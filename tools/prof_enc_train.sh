#!/bin/bash
# rocprofv3 kernel trace of the dual-encoder forward + backward at the PPO step's shapes (bench.py's dual_encoder_train), summarised
# per dispatch signature.  gpurun -- bash tools/prof_enc_train.sh r03
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/dbg/enc_train_prof.py --iters 2 --detail > $O/${TAG}_encoder_train_events.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_train_kt -- python3 $R/tools/dbg/enc_train_prof.py --iters 3 > $O/encoder_train_under_trace.txt 2> $O/enc_train_kt.err || exit 1
python3 $R/tools/rocprof_summary.py kernel-trace $O/enc_train_kt --title "rocprofv3 --kernel-trace --stats -- python3 tools/dbg/enc_train_prof.py --iters 3 ($TAG; 1 warm-up + 3 timed forward+backward passes of ViT-B/16 over 512 frames + RoBERTa-base over 64 sequences)" --md $O/${TAG}_encoder_train_kernel_trace.md --json $O/${TAG}_encoder_train_kernel_trace.json || exit 1
rm -rf $O/enc_train_kt
cat $O/${TAG}_encoder_train_events.txt
